#!/bin/bash
# Vector-memory path counters of one layer (tools/run_layer.py): tools/pmc_cache.sh <outdir> <layer> <mode>
out=$1; layer=$2; mode=$3
cd /tmp; export TMPDIR=/tmp
P1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/c$i -- python3 $GRAFT_REPO_ROOT/tools/run_layer.py $layer 3 $mode > $out/c$i.log 2>&1
  tail -2 $out/c$i.log
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/c*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv' not in n and 'wgrad' not in n: continue
        key = n.split('(')[0][-70:]
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        print('   %-40s per dispatch %16.0f' % (c, d[c] / max(cnt[(k, c)], 1)))
PY
