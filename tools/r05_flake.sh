#!/bin/bash
# how often does the teacher-forced 32^3 batch-2 test fail, and with what stem.short.w error, under the settings given as arguments (16 runs each)
for cfg in "$@"; do
  f=0; vals=""
  for i in $(seq 1 16); do
    o=$(env $cfg python -m pytest tests/test_gpu_teacher.py -x -q -s -k "train_step_32_b2 and not unfused" -p no:cacheprovider 2>&1)
    echo "$o" | grep -q "1 passed" || f=$((f+1))
    v=$(echo "$o" | grep "stem.short.w" | grep "rel=" | sed 's/.*rel=\([0-9.e+-]*\).*/\1/' | sort -g | tail -1)
    vals="$vals $v"
  done
  echo "$cfg : failures $f / 16 ; worst stem.short.w rel per run:$vals"
done
