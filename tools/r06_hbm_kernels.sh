#!/bin/bash
# per-(kernel, grid) durations of the kernels that do no matrix work (serial schedule, 7 steps): which launches are the big ones and what
# HBM rate they reach -- tools/r06_hbm_kernels.sh <tag>  ->  gpurun_out/hbm_kernels_<tag>.txt
tag=${1:-x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hk_$tag; mkdir -p $O
export VG_NO_REBUILD=1
cd /tmp; export TMPDIR=/tmp
VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/log 2>&1
cd $R
python3 tools/trace_by_grid.py $O/t actnorm pw_ skel erode materialize reduce_partials pack_weights adam sqnorm mm_ ssim bce mse dot_sums cldice stem_short c1m cells tanh axpby minmax randn dropout fill copy > gpurun_out/hbm_kernels_$tag.txt 2>&1
rm -rf $O
head -50 gpurun_out/hbm_kernels_$tag.txt | cut -c1-200
