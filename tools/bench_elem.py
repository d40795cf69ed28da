"""Micro-benchmark of the (IN -> act) backward kernels and conv epilogue statistics (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import ops
from van_gan_amd._lib import ActNormBwdDesc, lib, check
import ctypes as C

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device('cuda:0')
import sys
Ws = [int(a) for a in sys.argv[1:]] or [0]
for (S, Cc, padded) in [(128, 16, True), (128, 48, True), (64, 32, True), (128, 16, False)] + [(w, 16, True) for w in Ws if w]:
    N = 1
    dims = (N, S, S, S)
    gshape = (N, S + 2, S + 2, S + 2, Cc) if padded else (N, S, S, S, Cc)
    g = torch.randn(gshape, device=dev).to(torch.bfloat16)
    x = torch.randn(N, S, S, S, Cc, device=dev).to(torch.bfloat16)
    sc, sh, mean, rstd = [torch.rand(N, Cc, device=dev) + 0.5 for _ in range(4)]
    gamma = torch.ones(Cc, device=dev)
    red = torch.zeros(8, N, Cc, 2, device=dev)
    dx = torch.zeros(N, S, S, S, Cc, dtype=torch.bfloat16, device=dev)
    d = ActNormBwdDesc()
    d.g, d.g_padded, d.x = g.data_ptr(), int(padded), x.data_ptr()
    d.N, d.D, d.H, d.W, d.C = N, S, S, S, Cc
    d.scale, d.shift, d.act, d.norm = sc.data_ptr(), sh.data_ptr(), 1, 1
    d.gamma, d.mean, d.rstd, d.red = gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), red.data_ptr()
    d.dx, d.accumulate = dx.data_ptr(), 0
    st = ops.stream()
    t_s = timeit(lambda: check(lib.vg_actnorm_bwd_stats(C.byref(d), st)))
    t_a = timeit(lambda: check(lib.vg_actnorm_bwd_apply(C.byref(d), st)))
    gb = (g.numel() + x.numel()) * 2 / 1e9
    print('S=%d C=%d padded=%d: stats %.3f ms (%.0f GB/s)  apply %.3f ms (%.0f GB/s)' % (S, Cc, padded, t_s, gb / t_s * 1e3, t_a, (gb + dx.numel() * 2 / 1e9) / t_a * 1e3))
