#!/bin/bash
# BASELINE config 3 (128x128x64, batch 2, bf16, clDice on) under rocprofv3 on the GPU box:  tools/collect_config3.sh <tag>
#   kernel-trace stats of the serial schedule, then (separate runs) the MFMA-busy and the HBM FETCH_SIZE / WRITE_SIZE counters, summarised by
#   the same tools as the 128^3 workload.  Output: gpurun_out/prof_<tag>_c3/ and profiles/<tag>_config3_*.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${tag}_c3; mkdir -p $O
export VG_NO_REBUILD=1
# --no-replay: every rocprof pass must see exactly warmup + steps train steps (tools/hbm_pmc.py and tools/mfma_pmc.py divide by them;
# round 5 ran the replay leg too and reported twice the launches and bytes "per step")
A="--dims 128 128 64 --batch 2 --no-cpu-baseline --no-roofline --no-synced --no-replay"
cd /tmp; export TMPDIR=/tmp
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 $A > $O/stats.log 2>&1
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 $A > $O/fetch.log 2>&1
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 $A > $O/write.log 2>&1
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 1 --warmup 1 $A > $O/mfma.log 2>&1
cd $R
python3 tools/hbm_pmc.py $O/fetch $O/write $O/hbm_pmc.json > $O/hbm_pmc.log 2>&1
python3 tools/mfma_pmc.py $O/mfma $O/mfma_pmc.json > $O/mfma_pmc.log 2>&1
python3 $R/bench.py --steps 20 --warmup 5 $A > $O/bench_line.json 2> $O/bench.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write $O/mfma
cp $O/kernel_stats.csv $R/profiles/${tag}_config3_kernel_stats.csv; cp $O/hbm_pmc.json $R/profiles/${tag}_config3_hbm_pmc.json
cp $O/mfma_pmc.json $R/profiles/${tag}_config3_mfma_pmc.json; cp $O/bench_line.json $R/profiles/${tag}_config3_line.json
python3 - <<PY
import json
for f in ('$R/profiles/${tag}_config3_hbm_pmc.json', '$R/profiles/${tag}_config3_mfma_pmc.json'):
    j = json.load(open(f))
    j['workload'] = 'BASELINE config 3: VanGan.train_step, 128x128x64 volumes, batch 2, bf16, clDice on (bench.py --dims 128 128 64 --batch 2); read that for the "(128^3, batch 1)" of the note'
    json.dump(j, open(f, 'w'), indent=1)
PY
cp $R/profiles/${tag}_config3_* $O/ 2>/dev/null
cut -c1-400 $O/bench_line.json
