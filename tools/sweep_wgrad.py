"""Sweep of the materialised-operand weight gradient (vg_wgrad_dma.hip) over its plan knobs, layer by layer at the shapes of
one train step (development aid; run on the GPU box):
    python tools/sweep_wgrad.py [--size 128] [--layers bridge,down2] [--quick]
For every layer: the default plan's time, then every (CO, PL, BX, BM) combination that runs; prints the best five."""
import argparse
import ctypes as C
import itertools
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from van_gan_amd import _lib, ops  # noqa: E402
from van_gan_amd.nets import ParamStore  # noqa: E402
from van_gan_amd.ops import ConvLayer, Src  # noqa: E402


def tune(key, val):
    _lib.lib.vg_set_tuning(key.encode(), int(val), 0 if val is not None else 1)


def untune(key):
    _lib.lib.vg_set_tuning(key.encode(), 0, 1)


def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (name, k, cin, cout, stride, pad, level, concat split, N)
def layers(S):
    L = [('stem.cb 16->16', 3, 16, 16, 1, 'reflect', 0, None, 1), ('enc1.cb1 16->32 s2', 3, 16, 32, 2, 'reflect', 0, None, 1),
         ('enc1.cb2 32->32', 3, 32, 32, 1, 'reflect', 1, None, 1), ('enc2.cb1 32->64 s2', 3, 32, 64, 2, 'reflect', 1, None, 1),
         ('enc2.cb2 64->64', 3, 64, 64, 1, 'reflect', 2, None, 1), ('enc3.cb1 64->128 s2', 3, 64, 128, 2, 'reflect', 2, None, 1),
         ('enc3.cb2 128->128', 3, 128, 128, 1, 'reflect', 3, None, 1), ('enc4.cb1 128->256 s2', 3, 128, 256, 2, 'reflect', 3, None, 1),
         ('bridge 256->256', 3, 256, 256, 1, 'reflect', 4, None, 1),
         ('dec3.cb1 384->128', 3, 384, 128, 1, 'reflect', 3, (256, 128), 1), ('dec2.cb1 192->64', 3, 192, 64, 1, 'reflect', 2, (128, 64), 1),
         ('dec1.cb1 96->32', 3, 96, 32, 1, 'reflect', 1, (64, 32), 1), ('dec0.cb1 48->16', 3, 48, 16, 1, 'reflect', 0, (32, 16), 1),
         ('D.down0 64->128 k4s2', 4, 64, 128, 2, 'reflect', 1, None, 2), ('D.down1 128->256 k4s2', 4, 128, 256, 2, 'reflect', 2, None, 2),
         ('D.down2 256->512 k4s1', 4, 256, 512, 1, 'same', 3, None, 2)]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--layers', default='')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--gen-batch', type=int, default=2, help='samples of the generator layers (2: the paired backward sweep at batch 1)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    S = args.size
    for name, k, cin, cout, stride, pad, lv, cat, N in layers(S):
        if not name.startswith('D.'):
            N = args.gen_batch
        if args.layers and not any(t in name for t in args.layers.split(',')):
            continue
        dims = (S >> lv,) * 3
        st = ParamStore([('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
        lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims, need_dgrad=False)
        scale = torch.rand(N, cin, device=dev) + 0.5
        shift = torch.randn(N, cin, device=dev) * 0.1
        if cat:
            low = torch.randn(N, dims[0] // 2, dims[1] // 2, dims[2] // 2, cat[0], device=dev).to(torch.bfloat16)
            skip = torch.randn(N, *dims, cat[1], device=dev).to(torch.bfloat16)
            src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=scale, shift=shift, act=ops.ACT_RELU)
        else:
            x = torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16)
            nz = None
            if name.startswith('D.'):
                npad = 1 if pad == 'reflect' else 0
                nz = (torch.randn(N, *[d_ + 2 * npad for d_ in dims], cin, device=dev) * 0.1).to(torch.bfloat16)
            src = Src(x, (N,) + dims, cin, scale=scale, shift=shift, act=ops.ACT_LRELU if nz is not None else ops.ACT_RELU,
                      noise=nz, noise_pad=1 if pad == 'reflect' else 0)
        dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
        flops = 2.0 * N * math.prod(lay.out_dims) * cout * cin * k ** 3
        fn = lambda: lay._wgrad(src, dy)

        def variant():
            d = lay._fwd_desc(src)
            vb = C.create_string_buffer(512)
            sc = ops.WGRAD_SCRATCH[(dy.device, ops.stream())]
            _lib.lib.vg_conv3d_wgrad_variant(C.byref(d), 0, lay.f_idx_host, lay.f_T, sc.numel() * 4, vb, 512)
            return vb.value.decode()
        for key in ('WGRAD_DMA', 'WGRAD_DMA_CO', 'WGRAD_DMA_PL', 'WGRAD_DMA_BX', 'WGRAD_DMA_BM', 'WGRAD_DMA_NBUF', 'WGRAD_DMA_2CU'):
            untune(key)
        t_def = timeit(fn)
        v_def = variant()
        tune('WGRAD_DMA_2CU', 0)
        t_nb2 = timeit(fn)
        untune('WGRAD_DMA_2CU')
        print('%-24s default %7.1f us %6.0f TF/s  %s   | one WG per CU %7.1f us' % (name, t_def * 1e3, flops / t_def / 1e9, v_def, t_nb2 * 1e3), flush=True)
        npl = cin // 16
        res = []
        cos = [c for c in (64, 32, 16) if c <= cout]
        pls = [p for p in range(1, npl + 1) if npl % p == 0 and p <= 6]
        bxs = [1, 2, 4, 8, 16, 32, 64, 128, 256] if not args.quick else [1, 4, 16, 64, 256]
        bms = [512, 256, 128, 64] if not args.quick else [512, 128]
        seen = set()
        for co, pl, bx, bm in itertools.product(cos, pls, bxs, bms):
            tune('WGRAD_DMA_CO', co); tune('WGRAD_DMA_PL', pl); tune('WGRAD_DMA_BX', bx); tune('WGRAD_DMA_BM', bm)
            v = variant()
            if not v.startswith("wgrad_dma") or (v, bx) in seen:
                continue
            seen.add((v, bx))
            cols = (npl // pl) * (cout // co)
            if cols * bx > 1024:
                continue
            try:
                t = timeit(fn, 3)
            except Exception as e:          # noqa: BLE001
                print('   failed', co, pl, bx, bm, e); continue
            res.append((t, co, pl, bx, bm, v))
        res.sort()
        for t, co, pl, bx, bm, v in res[:6]:
            print('      %7.1f us %6.0f TF/s  co%d pl%d bx%d bm%d  %s' % (t * 1e3, flops / t / 1e9, co, pl, bx, bm, v))
        for key in ('WGRAD_DMA_CO', 'WGRAD_DMA_PL', 'WGRAD_DMA_BX', 'WGRAD_DMA_BM'):
            untune(key)


if __name__ == '__main__':
    main()
