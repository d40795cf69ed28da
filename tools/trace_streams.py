"""Per-queue view of a rocprofv3 --kernel-trace run of bench.py (default schedule): busy time, span and the top kernels of every
HIP stream inside the steady-state steps -- which lane is the long pole.   usage: python tools/trace_streams.py <trace_dir> [skip_fraction]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:50], r[key]) for r in rows]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print('columns available: %s; window %.3f ms' % (key, (t1 - t0) / 1e6))
per = collections.defaultdict(list)
for r in rows:
    per[r[3]].append(r)
for q, rs in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    busy = sum(e - s for s, e, _, _ in rs)
    # union within the queue (kernels of one queue may overlap slightly)
    cur, uni = rs[0][0], 0
    for s, e, _, _ in rs:
        if e > cur:
            uni += e - max(s, cur); cur = e
    agg = collections.defaultdict(float)
    for s, e, n, _ in rs:
        agg[n] += e - s
    top = sorted(agg.items(), key=lambda kv: -kv[1])[:6]
    print('queue %-6s n %5d  busy %8.3f ms (%.0f %% of window)  span %.3f..%.3f ms' % (q, len(rs), busy / 1e6, 100 * uni / (t1 - t0), (rs[0][0] - t0) / 1e6, (max(r[1] for r in rs) - t0) / 1e6))
    print('      ' + ', '.join('%s %.2f' % (n, v / 1e6) for n, v in top))
print()
for q, rs in per.items():
    gaps = []
    for (s0, e0, n0, _), (s1, e1, n1, _) in zip(rs, rs[1:]):
        if s1 - e0 > 30000:
            gaps.append((s1 - e0, (e0 - t0) / 1e6, n0, n1))
    tot = sum(g[0] for g in gaps)
    print('queue %s: %d idle gaps > 30 us, %.2f ms in total; largest:' % (q, len(gaps), tot / 1e6))
    for g in sorted(gaps, reverse=True)[:12]:
        print('   %8.1f us at +%8.3f ms   after %-44s before %s' % (g[0] / 1e3, g[1], g[2][:44], g[3][:44]))
