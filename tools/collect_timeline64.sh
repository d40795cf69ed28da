#!/bin/bash
# the same views as tools/collect_timeline.sh (2)-(4) for BASELINE config 2 (64^3, batch 2):  tools/collect_timeline64.sh <tag>
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline64_$tag; mkdir -p $O
export VG_NO_REBUILD=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --size 64 --batch 2 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/trace.log 2>&1
cd $R
{
  echo "# tools/collect_timeline64.sh $tag: 64^3 batch 2 train step (BASELINE config 2), default schedule"
  python3 tools/trace_streams.py $O/trace 0.5 | head -14
  echo; python3 tools/trace_gaps.py $O/trace 0.5 | head -8
  echo; python3 tools/trace_occupancy.py $O/trace 0.5
  echo; python3 tools/launches_per_step.py $O/trace | head -3
} > $R/gpurun_out/timeline64_$tag.txt 2>&1
rm -rf $O/trace
cat $R/gpurun_out/timeline64_$tag.txt | cut -c1-200
