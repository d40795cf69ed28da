#!/bin/bash
# Round 5: clDice kernels old vs new in the step (same box, alternating)
out=gpurun_out/r05_skel_ab.txt
: > $out
for rep in 1 2; do
for cfg in "VG_SKEL_MULTI=1 VG_SKEL_AUX=0" "VG_SKEL_MULTI=2 VG_SKEL_AUX=0" "VG_SKEL_MULTI=1 VG_SKEL_AUX=1" "VG_SKEL_MULTI=2 VG_SKEL_AUX=1"; do
  echo "== $cfg" >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-infer --no-cpu-baseline --no-ddp-path --no-roofline --no-synced 2>>gpurun_out/r05_skel_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done; done
cat $out
