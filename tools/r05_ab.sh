#!/bin/bash
# generic same-box A/B of environment settings on the 128^3 step: tools/r05_ab.sh "A=1" "A=0 B=2" ...   (3 alternating rounds)
out=gpurun_out/r05_ab.txt
: > $out
for rep in 1 2 3; do
for cfg in "$@"; do
  echo -n "$cfg : " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-infer --no-configs --no-cpu-baseline --no-ddp-path --no-roofline --no-synced --no-replay 2>>gpurun_out/r05_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])" >> $out
done; done
cat $out
