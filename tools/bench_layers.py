"""Per-layer micro-benchmark of the gather-convolution kernels at the shapes of the 128^3 train step
(development aid; run on the GPU box:  python tools/bench_layers.py [--size 128] [--only wgrad]).
Prints algorithmic TFLOP/s per layer for forward / data-gradient / weight-gradient."""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from van_gan_amd import ops  # noqa: E402
from van_gan_amd.nets import ParamStore  # noqa: E402
from van_gan_amd.ops import ConvLayer, Src  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--only', default='')
    ap.add_argument('--layers', default='')
    ap.add_argument('--batch', type=int, default=1)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    S = args.size
    # (name, k, cin, cout, stride, pad, level, concat split)
    L = [('stem.cb 16->16', 3, 16, 16, 1, 'reflect', 0, None), ('dec0.cb1 48->16', 3, 48, 16, 1, 'reflect', 0, (32, 16)),
         ('dec0.short 48->16 k1', 1, 48, 16, 1, 'same', 0, (32, 16)), ('dec1.short 96->32 k1', 1, 96, 32, 1, 'same', 1, (64, 32)),
         ('p.short 16->16 k1 64^3', 1, 16, 16, 1, 'same', 1, None), ('p.short 16->16 k1 32^3', 1, 16, 16, 1, 'same', 2, None),
         ('enc1.short 16->32 k1s2', 1, 16, 32, 2, 'same', 0, None), ('enc2.short 32->64 k1s2', 1, 32, 64, 2, 'same', 1, None), ('enc1.cb1 16->32 s2', 3, 16, 32, 2, 'reflect', 0, None),
         ('enc1.cb2 32->32', 3, 32, 32, 1, 'reflect', 1, None), ('dec1.cb1 96->32', 3, 96, 32, 1, 'reflect', 1, (64, 32)),
         ('enc2.cb2 64->64', 3, 64, 64, 1, 'reflect', 2, None), ('dec2.cb1 192->64', 3, 192, 64, 1, 'reflect', 2, (128, 64)),
         ('enc3.cb2 128->128', 3, 128, 128, 1, 'reflect', 3, None), ('dec3.cb1 384->128', 3, 384, 128, 1, 'reflect', 3, (256, 128)),
         ('bridge 256->256', 3, 256, 256, 1, 'reflect', 4, None),
         ('D.down0 64->128 k4s2', 4, 64, 128, 2, 'reflect', 1, None), ('D.down1 128->256 k4s2', 4, 128, 256, 2, 'reflect', 2, None),
         ('D.down2 256->512 k4s1', 4, 256, 512, 1, 'same', 3, None)]
    print('%-26s %10s %10s %10s   (ms | TFLOP/s)' % ('layer', 'fwd', 'dgrad', 'wgrad'))
    for name, k, cin, cout, stride, pad, lv, cat in L:
        if args.layers and not any(t in name for t in args.layers.split(',')):
            continue
        dims = (S >> lv,) * 3
        specs = [('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')]
        st = ParamStore(specs, dev)
        st.param('c.w').normal_(0, 0.05)
        lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims)
        lay.pack()
        N = args.batch
        scale = torch.rand(N, cin, device=dev) + 0.5
        shift = torch.randn(N, cin, device=dev) * 0.1
        raw = 'short' in name                 # the shortcut convolutions read the block input as stored (no IN / activation)
        if raw:
            scale = shift = None
        if cat:
            low = torch.randn(N, dims[0] // 2, dims[1] // 2, dims[2] // 2, cat[0], device=dev).to(torch.bfloat16)
            skip = torch.randn(N, *dims, cat[1], device=dev).to(torch.bfloat16)
            src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=scale, shift=shift, act=ops.ACT_NONE if raw else ops.ACT_RELU)
        else:
            x = torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16)
            nz = None
            if name.startswith('D.'):
                npad = 1 if pad == 'reflect' else 0
                nz = (torch.randn(N, *[d_ + 2 * npad for d_ in dims], cin, device=dev) * 0.1).to(torch.bfloat16)
            src = Src(x, (N,) + dims, cin, scale=scale, shift=shift, act=ops.ACT_LRELU if nz is not None else (ops.ACT_NONE if raw else ops.ACT_RELU),
                      noise=nz, noise_pad=1 if pad == 'reflect' else 0)
        out = torch.zeros(N, *lay.out_dims, cout, dtype=torch.bfloat16, device=dev)
        sums = torch.zeros(8, N, cout, 2, device=dev)
        dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
        dp = torch.zeros(N, *lay.buf_dims, cin, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * N * math.prod(lay.out_dims) * cout * cin * k ** 3
        res = []
        for kind, fn in (('fwd', lambda: lay.forward(src, out, sums=None if os.environ.get('NOSUMS') else sums)), ('dgrad', lambda: lay.dgrad(dy, N, dp, raw)),
                         ('wgrad', lambda: lay.wgrad(src, dy))):
            if args.only and kind != args.only:
                res.append('        -')
                continue
            ms = timeit(fn)
            res.append('%6.3f|%5.0f' % (ms, flops / ms / 1e9))
        print('%-26s %12s %12s %12s   ck f=%d d=%s' % (name, res[0], res[1], res[2], lay.f_ck, lay.d_classes[0]['ck']))


if __name__ == '__main__':
    main()
