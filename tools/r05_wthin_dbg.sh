#!/bin/bash
# needs a development build of the library: VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE python -m van_gan_amd.build --force (the production library refuses the knob)
export VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE
for dbg in 0 1 2 3 4 5 7; do
  echo -n "WT_DBG=$dbg (1 no K loop, 2 no commit, 4 no loads after the first): "; VG_WT_DBG=$dbg VG_WGRAD_THIN_WGS=512 python tools/bench_layers.py --only wgrad --layers "stem.cb" --batch 2 2>&1 | grep "stem.cb" | cut -c40-60
done
