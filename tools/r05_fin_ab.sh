#!/bin/bash
# Round 5: InstanceNorm finalisation by the producing launch (VG_FIN_TAIL) on / off, same box, alternating
out=gpurun_out/r05_fin_ab.txt
: > $out
for rep in 1 2 3; do
for cfg in "VG_FIN_TAIL=0" "VG_FIN_TAIL=1"; do
  echo "== $cfg" >> $out
  env $cfg VG_NO_REBUILD=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-ddp-path --no-roofline --no-synced --no-replay 2>>gpurun_out/r05_fin_ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('128^3 %.3f | ' % d['ms_per_step'] + ' | '.join('%.3f' % c['eager_ms_per_step'] for c in d['configs']))" >> $out
done; done
cat $out
