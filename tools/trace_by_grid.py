"""Per-(kernel, grid) launch durations out of a rocprofv3 --kernel-trace run: which launches of a template are the slow ones.
usage: python tools/trace_by_grid.py <trace_dir> [name-substring ...]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
subs = sys.argv[2:]
agg = collections.defaultdict(lambda: [0, 0.0, 1e18, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    if subs and not any(s in n for s in subs):
        continue
    k = (n.split('(')[0][:70], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[k]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print('total %.3f ms' % (tot / 1e3))
for k, v in rows[:60]:
    print('%-70s grid %8s x%-3s wg %4s  n %5d  avg %8.1f us  min %8.1f  max %8.1f  sum %8.3f ms' % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0], v[2], v[3], v[1] / 1e3))
