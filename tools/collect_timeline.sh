#!/bin/bash
# Where the 128^3 step's wall time goes, lane by lane (VERDICT r5 ask #6):  tools/collect_timeline.sh <tag>
#   (1) HIP-event milestones of both lanes (tools/timeline.py), (2) a rocprofv3 kernel trace of the DEFAULT schedule (two lanes, side
#   streams, optimizer stream) summarised per queue: busy time, idle gaps and what each gap waits for (tools/trace_streams.py,
#   tools/trace_gaps.py), (3) launches per steady-state step by kernel name.  Output: profiles/<tag>_lane_timeline.txt
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline_$tag; mkdir -p $O
export VG_NO_REBUILD=1
cd $R
python3 tools/timeline.py > $O/milestones.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/trace.log 2>&1
cd $R
{
  echo "# tools/collect_timeline.sh $tag: 128^3 batch 1 train step, default schedule"
  echo "## (1) lane milestones from HIP events (tools/timeline.py; last of two steps)"
  awk 'BEGIN{RS=""} {last=$0} END{print last}' $O/milestones.txt
  echo; echo "## (2) per-queue busy time and idle gaps, steady-state steps of a rocprofv3 --kernel-trace run (tools/trace_streams.py)"
  python3 tools/trace_streams.py $O/trace 0.5
  echo; echo "## (3) whole-GPU idle time (tools/trace_gaps.py)"
  python3 tools/trace_gaps.py $O/trace 0.5
  echo; echo "## (3b) how full the chip is over the step (tools/trace_occupancy.py)"
  python3 tools/trace_occupancy.py $O/trace 0.5
  echo; echo "## (4) launches per steady-state step, by kernel (tools/launches_per_step.py: steps delimited by adam_kernel groups)"
  python3 tools/launches_per_step.py $O/trace
} > $R/profiles/${tag}_lane_timeline.txt 2>&1
rm -rf $O/trace
cp $R/profiles/${tag}_lane_timeline.txt $O/
tail -40 $R/profiles/${tag}_lane_timeline.txt | cut -c1-200
