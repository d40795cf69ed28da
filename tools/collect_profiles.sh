#!/bin/bash
# Collects the round's profile evidence on the GPU box into gpurun_out/prof_<tag>/ :  tools/collect_profiles.sh <tag>
#   bench line (default schedule), per-kernel table of the timing step, rocprofv3 kernel-trace stats (serial schedule),
#   FETCH_SIZE / WRITE_SIZE / MFMA PMC passes (separate runs, --kernel-trace only), joined roofline table.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
export VG_NO_REBUILD=1
cd /tmp; export TMPDIR=/tmp
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/stats.log 2>&1     # 7 train steps in the trace
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/fetch.log 2>&1
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/write.log 2>&1
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/mfma.log 2>&1
cd $R
python3 tools/hbm_pmc.py $O/fetch $O/write $O/hbm_pmc.json > $O/hbm_pmc.log 2>&1
cp $O/hbm_pmc.json $R/profiles/${tag}_hbm_pmc.json     # the bench line below quotes it (same kernel sources: hash-stamped)
python3 $R/bench.py --steps 20 --warmup 5 --dump-kernels $O/bench_kernels.json > $O/bench_line.json 2> $O/bench.err
python3 tools/mfma_pmc.py $O/mfma $O/mfma_pmc.json > $O/mfma_pmc.log 2>&1
python3 tools/roofline_by_kernel.py $O/bench_kernels.json $O/stats 7 $O/fetch $O/write 2 $O/roofline_by_kernel.json > $O/roofline_by_kernel.txt 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write $O/mfma          # raw traces are large; the summaries above are what gets committed
cp $O/roofline_by_kernel.json $R/profiles/${tag}_roofline_by_kernel.json; cp $O/roofline_by_kernel.txt $R/profiles/${tag}_roofline_by_kernel.txt
cp $O/kernel_stats.csv $R/profiles/${tag}_bench128_kernel_stats.csv; cp $O/mfma_pmc.json $R/profiles/${tag}_mfma_pmc.json; cp $O/bench_line.json $R/profiles/${tag}_bench128_line.json
cp $R/profiles/${tag}_*.* $O/ 2>/dev/null
tail -3 $O/roofline_by_kernel.txt; cut -c1-300 $O/bench_line.json
