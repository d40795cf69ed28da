#!/bin/bash
# PMC passes over one layer (tools/run_layer.py): usage  tools/pmc_layer.sh <outdir> <layer> <mode>   (env selects the kernel flavour)
out=$1; layer=$2; mode=$3
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_layer.py $layer 3 $mode > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv' not in n and 'wgrad' not in n: continue
        key = n.split('(')[0][-70:]
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        print('   %-28s %16.0f  (per dispatch %14.0f)' % (c, d[c], d[c] / max(cnt[(k, c)], 1)))
PY
