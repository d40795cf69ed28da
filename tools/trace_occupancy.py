"""How full is the chip over a train step?  From a rocprofv3 --kernel-trace run of bench.py (default schedule): every kernel is given the
fraction of the 256 CUs its grid can occupy -- min(1, workgroups / 256) (a workgroup per CU is the floor of what fills a CU; most kernels here
run 1-2 workgroups of 256 threads per CU) -- and the step's wall time is binned by the SUM of those fractions over the kernels running
at that instant.  Small-grid phases (the 8^3 / 16^3 levels) show up as wall time at a low fill even though "a kernel is running"
100 % of the time.   usage: python tools/trace_occupancy.py <trace_dir> [skip_fraction=0.5]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for r in csv.DictReader(open(f)):
    gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(1, int(r.get('Grid_Size_Y', 1) or 1)) * max(1, int(r.get('Grid_Size_Z', 1) or 1))
    wx = max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 256)) or 256)) * max(1, int(r.get('Workgroup_Size_Y', 1) or 1)) * max(1, int(r.get('Workgroup_Size_Z', 1) or 1))
    wgs = max(1, gx // wx)
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), min(1.0, wgs / 256.0), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:60], wgs))
rows.sort()
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, fr, n, w in rows:
    ev.append((s, fr, n)); ev.append((e, -fr, n))
ev.sort()
bins = collections.OrderedDict((k, 0) for k in ('idle', '< 0.25', '0.25-0.5', '0.5-1', '1-2', '>= 2'))
low_by = collections.Counter()
fill, last, live = 0.0, t0, collections.Counter()
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        k = 'idle' if fill < 1e-9 else '< 0.25' if fill < 0.25 else '0.25-0.5' if fill < 0.5 else '0.5-1' if fill < 1 else '1-2' if fill < 2 else '>= 2'
        bins[k] += dt
        if 1e-9 < fill < 0.5:
            for nm, c in live.items():
                if c > 0:
                    low_by[nm] += dt
    fill += d; last = t
    live[n] += 1 if d > 0 else -1
tot = t1 - t0
print('window %.3f ms; wall time by chip fill (sum over running kernels of min(1, workgroups / 256)):' % (tot / 1e6))
for k, v in bins.items():
    print('   fill %-9s %8.3f ms  %5.1f %%' % (k, v / 1e6, 100.0 * v / tot))
print('kernels running while the fill is below 0.5 (wall time they were live in such intervals; intervals overlap):')
for n, v in low_by.most_common(18):
    print('   %-60s %8.3f ms' % (n, v / 1e6))
