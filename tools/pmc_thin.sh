#!/bin/bash
# Instruction counters of the thin-channel specialist inside the train step, per kernel template:  tools/pmc_thin.sh <tag> [VG_KEY=value ...]
# (one rocprofv3 --pmc pass over 3 serial-schedule steps; VALU / MFMA / LDS / VMEM instructions per wave and the busy cycles).
# Written for VERDICT r4 item 4: VALU instructions per MFMA of conv_thin<...,bs> before / after the panel loop (VG_CONV_THIN_PL=0 / 1).
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_thin_$tag; mkdir -p $O
export VG_NO_REBUILD=1 VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay > $O/run.log 2>&1
cd $R; python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_thin_kernel' not in n: continue
        key = re.sub(r'\(.*', '', n).replace('void ', '')
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
lines = ['%s: instructions per wave (all dispatches of 3 steps) and per MFMA' % tag]
for k, d in sorted(agg.items()):
    w, mf = max(d['SQ_WAVES'], 1), max(d['SQ_INSTS_MFMA'], 1)
    lines.append('%-62s dispatches %3d  valu/wave %7.0f  mfma/wave %6.0f  lds/wave %6.0f  vmem_rd/wave %5.0f  vmem_wr/wave %5.0f  salu/wave %6.0f  VALU per MFMA %.2f' % (
        k, cnt[(k, 'SQ_WAVES')], d['SQ_INSTS_VALU'] / w, d['SQ_INSTS_MFMA'] / w, d['SQ_INSTS_LDS'] / w, d['SQ_INSTS_VMEM_RD'] / w, d['SQ_INSTS_VMEM_WR'] / w, d['SQ_INSTS_SALU'] / w, d['SQ_INSTS_VALU'] / mf))
open('gpurun_out/pmc_thin_%s.txt' % tag, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
rm -rf $O/p
