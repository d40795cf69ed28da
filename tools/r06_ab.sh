#!/bin/bash
# same-box A/B of environment settings: the 128^3 train step (ms per step) and the config-5 inference (ms per volume, fp16 / bf16), three
# alternating rounds:   tools/r06_ab.sh "VG_STEM_FUSED=1" "VG_STEM_FUSED=0"   ->  gpurun_out/r06_ab.txt
out=gpurun_out/r06_ab.txt
: > $out
for rep in 1 2 3; do
for cfg in "$@"; do
  echo -n "$cfg : step " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-infer --no-configs --no-cpu-baseline --no-ddp-path --no-roofline --no-synced --no-replay 2>>gpurun_out/r06_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'], end='')" >> $out
  if [ -z "$AB_NO_INFER" ]; then
  echo -n "  infer fp16/bf16 " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --infer --steps 5 --warmup 2 2>>gpurun_out/r06_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f / %.2f' % (d['ms_per_step'], d['bf16']['ms_per_volume']), end='')" >> $out
  fi
  echo >> $out
done; done
cat $out
