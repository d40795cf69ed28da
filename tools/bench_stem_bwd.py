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

lib = ops._lib.lib
S = D * D * D
for cap in (255, 509, 767, 1021, 2045):
    lib.vg_set_tuning(b'STEM_BWD_GRID', cap, 0)
    G = int(lib.vg_stem_short_bwd_workgroups(N, S, 16))
    part = torch.empty(N * G * 34, dtype=torch.float64, device=dev)
    ticket = torch.zeros(4, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = (g.data_ptr(), 0, x.data_ptr(), N, S, 16, w.data_ptr(), gamma.data_ptr(), 1e-3, 1, dw.data_ptr(), dg.data_ptr(), db.data_ptr(), part.data_ptr(), G,
            ticket.data_ptr(), st)
    for _ in range(5):
        assert lib.vg_stem_short_bwd(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.vg_stem_short_bwd(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print('grid cap %5d (G = %4d per sample): %7.1f us per launch, %.2f TB/s' % (cap, G, us, nbytes / us / 1e6))
lib.vg_set_tuning(b'STEM_BWD_GRID', 0, 1)
# reference value in float64
gd, xd = g.double().reshape(N, -1, 16), x.double().reshape(N, -1, 1)
xc = xd - xd.mean(1, keepdim=True)
wq = w.to(torch.bfloat16).double()
rs = (wq[None] ** 2 * (xc ** 2).mean(1) + 1e-3).rsqrt()
ref = (1e-3 * gamma.double() * (rs ** 3 * (gd * xc).sum(1)).sum(0))
dw.zero_(); ar.reset(); ops.stem_short_bwd(ar, g, x, N, C, w, gamma, dw, dg, db); torch.cuda.synchronize()
print('dw vs float64: rel %.2e' % float((dw.double() - ref).norm() / ref.norm()))
