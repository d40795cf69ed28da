"""In-kernel phase stamps of conv_kernel for one layer (diagnostic): cycles spent in staging / barrier wait / MFMA loop /
epilogue per tile, median over workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from van_gan_amd import ops
from van_gan_amd._lib import lib
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0')
cases = {'stem': (3, 16, 16, 1, 'reflect', 128, None), 'dec0': (3, 48, 16, 1, 'reflect', 128, (32, 16)), 'enc2': (3, 64, 64, 1, 'reflect', 32, None),
         'down2': (4, 256, 512, 1, 'same', 16, None), 'bridge': (3, 256, 256, 1, 'reflect', 8, None), 'enc3': (3, 128, 128, 1, 'reflect', 16, None),
         'dec3': (3, 384, 128, 1, 'reflect', 16, (256, 128))}
for name in (sys.argv[1:] or list(cases)):
    k, cin, cout, stride, pad, S, cat = cases[name]
    dims = (S,) * 3
    st = ParamStore([('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
    st.param('c.w').normal_(0, 0.05)
    lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims); lay.pack()
    N = 1
    sc, sh = torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.1
    if cat:
        low = torch.randn(N, S // 2, S // 2, S // 2, cat[0], device=dev).to(torch.bfloat16)
        skip = torch.randn(N, *dims, cat[1], device=dev).to(torch.bfloat16)
        src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=sc, shift=sh, act=ops.ACT_RELU)
    else:
        src = Src(torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16), (N,) + dims, cin, scale=sc, shift=sh, act=ops.ACT_RELU)
    out = torch.zeros(N, *lay.out_dims, cout, dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(8, N, cout, 2, device=dev)
    lay.forward(src, out, sums=sums); torch.cuda.synchronize()
    buf = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
    lib.vg_set_stamp_buffer(buf.data_ptr())
    lay.forward(src, out, sums=sums); torch.cuda.synchronize()
    lib.vg_set_stamp_buffer(None)
    b = buf.cpu().numpy().reshape(-1, 8, 8)
    used = b[:, 0, 0] > 0
    b = b[used]
    print('%s: %d workgroups stamped' % (name, len(b)))
    for it in range(min(4, 8)):
        ok = b[:, it, 4] > 0
        if not ok.any(): break
        x = b[ok, it].astype(np.float64)
        stage, wait, mfma, epi = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3]
        print('  tile %d: stage %7.0f  barrier %7.0f  mfma %7.0f  epilogue %7.0f   (median cycles, %d wgs)' % (
            it, np.median(stage), np.median(wait), np.median(mfma), np.median(epi), ok.sum()))
    span = (b[:, :, 4].max(axis=1) - b[:, 0, 0])
    print('  first-8-tiles span per wg: median %.0f cycles' % np.median(span))
