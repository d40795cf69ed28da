"""Launches per STEADY-STATE train step by kernel name out of a rocprofv3 --kernel-trace run of bench.py: the trace is cut at the last
adam_kernel of every step (`adam_per_step` launches: 4 networks, 6 with the generators' early suffix update), the first steps (start-up
copies, fills, warm-up) are dropped and the rest averaged.  Round 5's
"68 __amd_rocclr_copyBuffer per step" was the whole trace divided by its steps: the start-up weight loads were in it.
usage: python tools/launches_per_step.py <trace_dir> [steps_to_skip=3] [adam_per_step=6]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
APS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:70])
              for r in csv.DictReader(open(f)))
cuts, n_adam = [], 0
for i, (s, e, n) in enumerate(rows):
    if n.startswith('adam_kernel'):
        n_adam += 1
        if n_adam % APS == 0:
            cuts.append(i + 1)
steps = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])][skip:]
if not steps:
    raise SystemExit('not enough steps in the trace')
cnt, dur = collections.Counter(), collections.Counter()
for st in steps:
    for s, e, n in st:
        cnt[n] += 1; dur[n] += e - s
ns = len(steps)
print('%d steady-state steps; launches per step %.1f, kernel time per step %.3f ms' % (ns, sum(cnt.values()) / ns, sum(dur.values()) / ns / 1e6))
for n, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print('%-70s %6.1f launches  %8.3f ms' % (n, c / ns, dur[n] / ns / 1e6))
