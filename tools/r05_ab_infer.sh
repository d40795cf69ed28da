#!/bin/bash
out=gpurun_out/r05_ab_infer.txt
: > $out
for rep in 1 2 3; do
for cfg in "$@"; do
  echo -n "$cfg : " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --infer --steps 4 --warmup 2 2>>gpurun_out/r05_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('fp16 %.2f bf16 %.2f' % (d['ms_per_step'], d['bf16']['ms_per_volume']))" >> $out
done; done
cat $out
