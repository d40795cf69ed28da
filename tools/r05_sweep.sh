#!/bin/bash
# same-box sweep of schedule / grid knobs on the 128^3 step (2 alternating rounds per setting): tools/r05_sweep.sh
out=gpurun_out/r05_sweep.txt
: > $out
run() { env $@ VG_NO_REBUILD=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-infer --no-configs --no-cpu-baseline --no-ddp-path --no-roofline --no-synced --no-replay 2>>gpurun_out/r05_sweep.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
for cfg in "X=0" "VG_WGRAD_INLINE=1" "VG_WGRAD_INLINE=0" "VG_WGRAD_INLINE=-1" "VG_WGRAD_INLINE=1 VG_BFIRST=7" "VG_WGRAD_INLINE=1 VG_BFIRST=9" "VG_WGRAD_INLINE=0 VG_BFIRST=7"; do
  echo "$cfg : $(run $cfg)" >> $out
done; done
sort $out
