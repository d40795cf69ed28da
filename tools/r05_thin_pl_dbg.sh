#!/bin/bash
# needs a development build of the library: VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE python -m van_gan_amd.build --force (the production library refuses the knob)
export VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE
# timing ablations of the panel-loop thin data gradient (conv_thin_kernel<0,..,true,1,3>): VG_THIN_DBG bits 1 no statistics flush,
# 2 no pre-norm loads, 4 no MFMA loop, 8 no epilogue, 16 staging only for a workgroup's first tile (results are wrong: timing only)
for dbg in "$@"; do
  VG_THIN_DBG=$dbg bash tools/quick_stats.sh pl$dbg 2>/dev/null | grep "true, 1, 3>" | awk -v d=$dbg '{print "dbg", d, $(NF-2), $(NF-1), $NF}'
done
