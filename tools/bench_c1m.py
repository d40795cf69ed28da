"""Development aid: the single-channel layers (stem 1 -> 16 3x3x3 with its 1x1x1 shortcut, D.conv0 1 -> 64 4x4x4 s2) alone, HIP events, over
the persistent-grid sizes of vg_c1k3.hip (C1M_FWD_WGS / C1M_WGRAD_WGS workgroups per CU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from van_gan_amd import ops, _lib
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src

dev = torch.device('cuda:0')
ops.set_device(0)
dims = (128, 128, 128)


def timeit(fn, R=30):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3


for (k, cout, stride, N) in ((3, 16, 1, 1), (4, 64, 2, 2), (1, 16, 1, 1)):
    st = ParamStore([('c.w', (k, k, k, 1, cout), 'he_normal'), ('c.b', (cout,), 'zeros')], dev)
    st.param('c.w').normal_(0, 0.05)
    lay = ConvLayer(st, 'c', k, 1, cout, stride, 'reflect' if k > 1 else 'same', True, dims)
    lay.pack()
    x = torch.randn(N, *dims, 1, device=dev)
    nz = (torch.randn(N, *[n + 2 for n in dims], 1, device=dev) * 0.1).to(torch.bfloat16) if k == 4 else None
    src = Src(x, (N,) + dims, 1, f32=True, noise=nz, noise_pad=1 if k == 4 else 0)
    out = torch.empty(N, *lay.out_dims, cout, dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(8, N, cout, 2, device=dev)
    dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
    for wgs in (1, 2, 3, 4, 6, 8):
        _lib.lib.vg_set_tuning(b'C1M_FWD_WGS', wgs, 0); _lib.lib.vg_set_tuning(b'C1M_WGRAD_WGS', wgs, 0)
        tf = timeit(lambda: lay.forward(src, out, sums=sums))
        tw = timeit(lambda: lay.wgrad(src, dy))
        print('k%d 1->%d s%d N=%d  wgs/CU %d  fwd %7.1f us   wgrad %7.1f us   %s' % (k, cout, stride, N, wgs, tf, tw, ops.conv_variant(lay._fwd_desc(src)) if False else ''))
        if k == 1:
            break
