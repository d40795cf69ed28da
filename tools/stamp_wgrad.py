"""In-kernel phase stamps of wgrad_kernel (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from van_gan_amd import ops
from van_gan_amd._lib import lib
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0')
cases = {'stem': (3, 16, 16, 1, 'reflect', 128, None), 'dec0': (3, 48, 16, 1, 'reflect', 128, (32, 16)), 'enc1': (3, 32, 32, 1, 'reflect', 64, None),
         'enc2': (3, 64, 64, 1, 'reflect', 32, None), 'down2': (4, 256, 512, 1, 'same', 16, None),
         'down0n': (4, 64, 128, 2, 'reflect', 64, 'noise'), 'down1n': (4, 128, 256, 2, 'reflect', 32, 'noise'), 'down2n': (4, 256, 512, 1, 'same', 16, 'noise'),
         'down0': (4, 64, 128, 2, 'reflect', 64, None)}
for name in (sys.argv[1:] or list(cases)):
    k, cin, cout, stride, pad, S, cat = cases[name]
    dims = (S,) * 3
    st = ParamStore([('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
    st.param('c.w').normal_(0, 0.05)
    lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims); lay.pack()
    N = 2 if name.startswith('down') else 1
    sc, sh = torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.1
    if cat == 'noise':
        npad = 1 if pad == 'reflect' else 0
        nz = (torch.randn(N, S + 2 * npad, S + 2 * npad, S + 2 * npad, cin, device=dev) * 0.1).to(torch.bfloat16)
        src = Src(torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16), (N,) + dims, cin, scale=sc, shift=sh, act=ops.ACT_LRELU, noise=nz, noise_pad=npad)
    elif cat:
        low = torch.randn(N, S // 2, S // 2, S // 2, cat[0], device=dev).to(torch.bfloat16)
        skip = torch.randn(N, *dims, cat[1], device=dev).to(torch.bfloat16)
        src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=sc, shift=sh, act=ops.ACT_RELU)
    else:
        src = Src(torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16), (N,) + dims, cin, scale=sc, shift=sh, act=ops.ACT_RELU)
    dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
    lay.wgrad(src, dy); torch.cuda.synchronize()
    buf = torch.zeros(8192 * 64, dtype=torch.int64, device=dev)
    lib.vg_set_stamp_buffer(buf.data_ptr())
    lay.wgrad(src, dy); torch.cuda.synchronize()
    lib.vg_set_stamp_buffer(None)
    b = buf.cpu().numpy().reshape(-1, 8, 8)
    b = b[b[:, 0, 0] > 0]
    print('%s: %d workgroups stamped' % (name, len(b)))
    for it in range(3):
        ok = b[:, it, 3] > 0
        if not ok.any(): break
        x = b[ok, it].astype(np.float64)
        print('  tile %d: stage %7.0f  barrier %7.0f  mfma %7.0f   (median cycles, %d wgs)' % (it, np.median(x[:, 1] - x[:, 0]), np.median(x[:, 2] - x[:, 1]), np.median(x[:, 3] - x[:, 2]), ok.sum()))
    print('  whole workgroup (first stamp -> slab write): median %.0f cycles' % np.median(b[:, 7, 7] - b[:, 0, 0]))
