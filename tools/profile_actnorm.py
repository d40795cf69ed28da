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
_orig = ops.actnorm_bwd
def wrapped(g, g_padded, x, dims, C_, dx, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = _orig(g, g_padded, x, dims, C_, dx, **kw); e1.record()
    key = (tuple(dims), C_, bool(g_padded), bool(kw.get('norm')), bool(kw.get('accumulate')), kw.get('x1') is not None, kw.get('dx_cstride', 0))
    rec.append((key, e0, e1)); return r
ops.actnorm_bwd = wrapped
import van_gan_amd.nets as nets
nets.ops.actnorm_bwd = wrapped
eng.train_step(rI, rS)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print('total actnorm_bwd (stats+apply) %.2f ms in %d calls' % (tot, len(rec)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    N, D, H, W = k[0]
    gb = N * D * H * W * k[1] * (10 if k[3] else 6) / 1e9
    print('dims %-18s C=%3d pad=%d norm=%d acc=%d cat=%d cs=%s  n=%3d  %7.3f ms  (%.0f GB/s algorithmic)' % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1], gb * v[0] / v[1] * 1e3))
