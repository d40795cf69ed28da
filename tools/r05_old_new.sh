#!/bin/bash
# same-box comparison of older trees with this one: inference (config 5); trees given as arguments (default: _wt_mid .)
out=gpurun_out/r05_old_new.txt
: > $out
trees="${@:-_wt_mid .}"
for rep in 1 2 3; do
for t in $trees; do
  echo -n "$t : " >> $out
  (cd $t; VG_NO_REBUILD=1 timeout 300 python bench.py --infer --steps 4 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('infer fp16 %.2f bf16 %.2f' % (d['ms_per_step'], d['bf16']['ms_per_volume']))") >> $out
done; done
cat $out
