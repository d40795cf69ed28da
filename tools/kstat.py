"""Per-kernel averages out of a rocprofv3 --kernel-trace --stats --output-format csv directory, filtered by substrings:
    python tools/kstat.py <dir> <steps> substr [substr ...]      (development aid)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2])
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r['Name']
    tot += float(r['TotalDurationNs'])
    if any(k in n for k in sys.argv[3:]):
        print('%-90s calls/step %6.1f avg %8.1f us  total/step %7.3f ms' % (n[:90], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
print('all kernels: %.3f ms per step' % (tot / 1e6 / steps))
