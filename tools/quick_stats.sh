#!/bin/bash
# rocprofv3 kernel-trace statistics of 7 serial-schedule train steps -> gpurun_out/<tag>_kernel_stats.csv (development aid; collect_profiles.sh is the full set)
tag=${1:-quick}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qs_$tag; mkdir -p $O
export VG_NO_REBUILD=1
cd /tmp; export TMPDIR=/tmp
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $O
cd $R; python3 - <<P
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per step %.2f' % (tot/7e6))
for r in rows[:60]:
    print('%-100s %5s %8.3f %8.1f' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/7e6, float(r['AverageNs'])/1e3))
P
