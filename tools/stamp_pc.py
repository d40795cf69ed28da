"""In-kernel phase stamps of conv_pc_kernel for one layer (diagnostic; VG_CONV_PC=1): cycles per stage spent by the
producer waves (staging) and by the consumer waves (MFMA loop, epilogue) and waiting at the stage barrier, median over
workgroups.  usage: VG_CONV_PC=1 python tools/stamp_pc.py [stem dec0 enc1 ...] [--dgrad]"""
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
         'dec1': (3, 96, 32, 1, 'reflect', 64, (64, 32)), 'enc2': (3, 64, 64, 1, 'reflect', 32, None)}
dgrad = '--dgrad' in sys.argv
for name in ([a for a in sys.argv[1:] if not a.startswith('--')] or list(cases)):
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
    dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
    dp = torch.zeros(N, *lay.buf_dims, cin, dtype=torch.bfloat16, device=dev)
    run = (lambda: lay.dgrad(dy, N, dp, False)) if dgrad else (lambda: lay.forward(src, out, sums=sums))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf = torch.zeros(8192 * 64, dtype=torch.int64, device=dev)
    lib.vg_set_stamp_buffer(buf.data_ptr())
    run(); torch.cuda.synchronize()
    lib.vg_set_stamp_buffer(None)
    b = buf.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
    b = b[b[:, 0, 2] > 0]
    print('%s %s: %.1f us, %d workgroups stamped' % (name, 'dgrad' if dgrad else 'fwd', e0.elapsed_time(e1) * 1e3, len(b)))
    k = b[:, :5, 7]
    okk = (k > 0).all(axis=1)
    if okk.any():
        kk = k[okk]
        print('  kernel phases (median cycles): tables+weights %6.0f | first stage staged %6.0f | stage loop %7.0f | tail %6.0f | total %7.0f ; first wg start -> last wg end %7.0f' % (
            np.median(kk[:, 1] - kk[:, 0]), np.median(kk[:, 2] - kk[:, 1]), np.median(kk[:, 3] - kk[:, 2]), np.median(kk[:, 4] - kk[:, 3]),
            np.median(kk[:, 4] - kk[:, 0]), kk[:, 4].max() - kk[:, 0].min()))
    for it in range(3):
        ok = (b[:, it, 6] > 0) & (b[:, it, 5] > 0)
        if not ok.any():
            break
        x = b[ok, it]
        med = lambda v: np.median(v)
        print('  stage %d: producer work %6.0f wait %6.0f | consumer mfma %6.0f epilogue %6.0f wait %6.0f | stage %6.0f cycles' % (
            it, med(x[:, 1] - x[:, 0]), med(x[:, 5] - x[:, 1]), med(x[:, 3] - x[:, 2]), med(x[:, 4] - x[:, 3]), med(x[:, 6] - x[:, 4]),
            med(x[:, 6] - x[:, 2])))
