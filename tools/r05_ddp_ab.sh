#!/bin/bash
# data-parallel schedule on one GPU (VG_FAKE_AR): delta against the plain step for settings given as arguments
out=gpurun_out/r05_ddp_ab.txt
: > $out
for rep in 1 2; do
for cfg in "$@"; do
  echo -n "$cfg : " >> $out
  env $cfg VG_NO_REBUILD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-configs --no-cpu-baseline --no-roofline --no-synced --no-replay 2>>gpurun_out/r05_ddp_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['ddp_path']; print('plain %.3f  ddp %.3f  delta %.3f' % (d['ms_per_step'], p['ms_per_step'], p['delta_ms']))" >> $out
done; done
cat $out
