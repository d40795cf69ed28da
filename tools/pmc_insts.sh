#!/bin/bash
# Instruction counters of EVERY kernel template of the train step that issues MFMAs:  tools/pmc_insts.sh <tag> [VG_KEY=value ...]
# (one rocprofv3 --pmc pass over 3 serial-schedule steps; VALU / MFMA / LDS / VMEM / SALU instructions per wave, VALU and SALU per MFMA).
# VERDICT r5 ask #6: SQ_INSTS_VALU / SQ_INSTS_MFMA per template beside the MFMA-busy fraction.  Output: gpurun_out/pmc_insts_<tag>.txt
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_insts_$tag; mkdir -p $O
export VG_NO_REBUILD=1 VG_LANES=0 VG_SIDE_STREAM=0 VG_OPT_STREAM=0
A=${PMC_ARGS:-}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-infer --no-configs --no-synced --no-ddp-path --no-replay $A > $O/run.log 2>&1
cd $R; python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        key = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
lines = ['%s: instructions per wave (all dispatches of 3 serial-schedule steps) and per MFMA, kernel templates that issue MFMAs' % tag]
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_INSTS_MFMA']):
    if d['SQ_INSTS_MFMA'] <= 0: continue
    w, mf = max(d['SQ_WAVES'], 1), d['SQ_INSTS_MFMA']
    lines.append('%-72s dispatches %3d  valu/wave %7.0f  mfma/wave %6.0f  lds/wave %6.0f  vmem_rd/wave %5.0f  vmem_wr/wave %5.0f  salu/wave %6.0f  VALU/MFMA %5.2f  SALU/MFMA %5.2f' % (
        k[:72], cnt[(k, 'SQ_WAVES')], d['SQ_INSTS_VALU'] / w, mf / w, d['SQ_INSTS_LDS'] / w, d['SQ_INSTS_VMEM_RD'] / w, d['SQ_INSTS_VMEM_WR'] / w, d['SQ_INSTS_SALU'] / w, d['SQ_INSTS_VALU'] / mf, d['SQ_INSTS_SALU'] / mf))
open('gpurun_out/pmc_insts_%s.txt' % tag, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
rm -rf $O/p
