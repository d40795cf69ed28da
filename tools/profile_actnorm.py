"""Per-call timing of the (IN -> act) backward launches of one 128^3 train step (development aid)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan, ops
from van_gan_amd.synth import synth_volumes
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = VanGan((size,) * 3, batch_size=1, device='cuda:0')
rI, rS = synth_volumes(1, size, size, size, seed=1)
rI, rS = rI.cuda(), rS.cuda()
eng.train_step(rI, rS)
rec = []
_orig = ops.actnorm_run
def wrapped(d, stats_done=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = _orig(d, stats_done); e1.record()
    key = ((d.N, d.D, d.H, d.W), d.C, bool(d.g_padded), bool(d.norm), bool(d.accumulate), bool(d.x1), 'apply only' if stats_done else 'stats+apply')
    rec.append((key, e0, e1)); return r
ops.actnorm_run = wrapped
import van_gan_amd.nets as nets
nets.ops.actnorm_run = wrapped
eng.train_step(rI, rS)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print('total IN backward (statistics where not carried by the data gradient, + apply) %.2f ms in %d calls' % (tot, len(rec)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    N, D, H, W = k[0]
    gb = N * D * H * W * k[1] * (10 if k[3] else 6) / 1e9
    print('dims %-18s C=%3d pad=%d norm=%d acc=%d cat=%d cs=%s  n=%3d  %7.3f ms  (%.0f GB/s algorithmic)' % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1], gb * v[0] / v[1] * 1e3))
