#!/bin/bash
# wgrad_thin: grid sizes (workgroups in all) on the two thin layers, batch 2, then in the step
for w in 256 384 512; do
  echo "== WGRAD_THIN_WGS=$w"; VG_WGRAD_THIN_WGS=$w python tools/bench_layers.py --only wgrad --layers "stem.cb,dec0.cb1" --batch 2 2>&1 | grep -v amdgpu | tail -3
done
echo "== old kernels"; VG_WGRAD_THIN=0 python tools/bench_layers.py --only wgrad --layers "stem.cb,dec0.cb1" --batch 2 2>&1 | grep -v amdgpu | tail -3
