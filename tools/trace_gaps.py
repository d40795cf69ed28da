"""GPU idle time inside train steps out of a rocprofv3 --kernel-trace run of bench.py (default schedule): union of the kernel
intervals vs wall time between the first and last kernel of the timed steps, plus the largest gaps and what follows them.
usage: python tools/trace_gaps.py <trace_dir> [skip_fraction]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in csv.DictReader(open(f))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]                      # the later part of the run: steady-state steps
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end, gaps = 0, rows[0][0], []
conc = 0
for s, e, n, q in rows:
    if s > cur_end:
        gaps.append((s - cur_end, n, cur_end - t0)); cur_end = s
    if e > cur_end:
        busy += e - cur_end; cur_end = e
tot = sum(e - s for s, e, _, _ in rows)
print('window %.3f ms, union busy %.3f ms (%.1f %%), sum of kernel durations %.3f ms (avg concurrency %.2f)' % ((t1 - t0) / 1e6, busy / 1e6, 100 * busy / (t1 - t0), tot / 1e6, tot / busy))
gaps.sort(reverse=True)
print('idle total %.3f ms in %d gaps; largest:' % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in gaps[:15]:
    print('  %8.1f us at +%.3f ms before %s' % (g[0] / 1e3, g[2] / 1e6, g[1]))
