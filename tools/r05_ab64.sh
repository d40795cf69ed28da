#!/bin/bash
# same-box A/B on BASELINE config 2 (64^3, batch 2): tools/r05_ab64.sh "A=1" "A=0" ...
out=gpurun_out/r05_ab64.txt
: > $out
for rep in 1 2 3 4 5; do
for cfg in "$@"; do
  echo -n "$cfg : " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --size 64 --batch 2 --steps 30 --warmup 5 --no-infer --no-configs --no-cpu-baseline --no-ddp-path --no-roofline --no-synced --no-replay 2>>gpurun_out/r05_ab64.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])" >> $out
done; done
cat $out
