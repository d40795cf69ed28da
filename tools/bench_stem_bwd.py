"""vg_stem_short_bwd alone at the 128^3 paired shape (N = 2, 16 channels): microseconds per launch and the HBM rate, over grid caps.
usage: python tools/bench_stem_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from van_gan_amd import ops  # noqa: E402
from van_gan_amd.ops import Arena  # noqa: E402

dev = torch.device('cuda:0')
ops.set_device(0)
N, D, C = 2, 128, 16
g = torch.randn(N, D, D, D, C, device=dev).to(torch.bfloat16)
x = torch.rand(N, D, D, D, 1, device=dev) * 2 - 1
w = torch.randn(C, device=dev) * 0.3
gamma = torch.ones(C, device=dev)
dw, dg, db = (torch.zeros(C, device=dev) for _ in range(3))
ar = Arena(64 << 20, dev)
nbytes = g.numel() * 2 + x.numel() * 4
for cap in (255, 509, 1021, 2045):
    ops._lib.lib.vg_set_tuning(b'STEM_BWD_GRID', cap, 0)
    for _ in range(3):
        ar.reset(); ops.stem_short_bwd(ar, g, x, N, C, w, gamma, dw, dg, db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ar.reset()
    e0.record()
    for _ in range(20):
        ops.stem_short_bwd(ar, g, x, N, C, w, gamma, dw, dg, db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('grid cap %5d: %7.1f us per launch, %.2f TB/s' % (cap, us, nbytes / us / 1e6))
# reference value in float64
gd, xd = g.double().reshape(N, -1, C), x.double().reshape(N, -1, 1)
xc = xd - xd.mean(1, keepdim=True)
wq = w.to(torch.bfloat16).double()
rs = (wq[None] ** 2 * (xc ** 2).mean(1) + 1e-3).rsqrt()
ref = (1e-3 * gamma.double() * (rs ** 3 * (gd * xc).sum(1)).sum(0))
dw.zero_(); ar.reset(); ops.stem_short_bwd(ar, g, x, N, C, w, gamma, dw, dg, db); torch.cuda.synchronize()
print('dw vs float64: rel %.2e' % float((dw.double() - ref).norm() / ref.norm()))
