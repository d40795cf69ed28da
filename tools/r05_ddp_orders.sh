#!/bin/bash
# Round 5: the data-parallel schedule on one GPU (bench.py --ddp-fake) under different stream placements of the communication.
# HIP serves the streams of a process from 4 hardware queues; which stream the communication aliases is decided here.
out=gpurun_out/r05_ddp_orders.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" VG_NO_REBUILD=1 timeout 300 python bench.py --ddp-fake --steps 20 --warmup 5 2>>gpurun_out/r05_ddp_orders.err | grep ddp_fake | python -c "
import sys,json
d=json.loads(sys.stdin.read())['ddp_fake']
print('plain %.2f' % d['plain_ms_per_step'], ' '.join('%s: %.2f (+%.2f)' % (l['ring_bus_GBps_emulated'], l['ms_per_step'], l['delta_ms']) for l in d['legs']), d['legs'][0]['comm_stream'])" >> $out
}
run VG_COMM_ON_OPT=1
run VG_COMM_ON_OPT=1 VG_FAKE_AR_NULL=1
run VG_COMM_ON_OPT=1 VG_FAKE_AR_WG=8
run VG_COMM_ON_OPT=1 VG_FAKE_AR_WG=64
run VG_COMM_ON_OPT=0
run VG_COMM_ON_OPT=1 VG_FAKE_AR_INNER=1
run VG_COMM_ON_OPT=1 VG_XSTEP=0
run VG_COMM_ON_OPT=1 VG_AR_SPLIT=0
run VG_COMM_ON_OPT=1 VG_AR_SPLIT=0 VG_XSTEP=0
cat $out
