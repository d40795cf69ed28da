"""In-kernel phase sums of wgrad_dma_kernel (diagnostic): python tools/stamp_wgrad_dma.py [stem dec0 ...] [KEY=VAL tuning ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from van_gan_amd import ops, _lib
from van_gan_amd._lib import lib
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0')
cases = {'stem': (3, 16, 16, 1, 'reflect', 128, None), 'dec0': (3, 48, 16, 1, 'reflect', 128, (32, 16)), 'enc1': (3, 32, 32, 1, 'reflect', 64, None),
         'enc2': (3, 64, 64, 1, 'reflect', 32, None), 'bridge': (3, 256, 256, 1, 'reflect', 8, None), 'dec1': (3, 96, 32, 1, 'reflect', 64, (64, 32)),
         'down0': (4, 64, 128, 2, 'reflect', 64, 'noise'), 'down1': (4, 128, 256, 2, 'reflect', 32, 'noise'), 'down2': (4, 256, 512, 1, 'same', 16, 'noise')}
names = [a for a in sys.argv[1:] if '=' not in a] or list(cases)
for a in sys.argv[1:]:
    if '=' in a:
        k_, v_ = a.split('=')
        lib.vg_set_tuning(k_.encode(), int(v_), 0)
for name in names:
    k, cin, cout, stride, pad, S, cat = cases[name]
    dims = (S,) * 3
    st = ParamStore([('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
    lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims, need_dgrad=False)
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
    for _ in range(3):
        lay._wgrad(src, dy)
    torch.cuda.synchronize()
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.vg_set_stamp_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lay._wgrad(src, dy); e1.record(); torch.cuda.synchronize()
    lib.vg_set_stamp_buffer(None)
    import ctypes as C
    d = lay._fwd_desc(src); vb = C.create_string_buffer(512)
    scr = ops.WGRAD_SCRATCH[(dy.device, ops.stream())]
    lib.vg_conv3d_wgrad_variant(C.byref(d), 0, lay.f_idx_host, lay.f_T, scr.numel() * 4, vb, 512)
    b = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    b = b[b[:, 5] > 0]
    rt = b[:, 7]
    print('%s: %s  call %.1f us; %d workgroups; span of end stamps %.1f us' % (name, vb.value.decode(), e0.elapsed_time(e1) * 1e3, len(b), (rt.max() - rt.min()) / 100.0))
    med = np.median(b, axis=0)
    print('   median cycles: prologue %6.0f | per tile: wait %6.0f issue %5.0f k-loop %6.0f (x %d tiles) | slab+db %6.0f | total %7.0f' % (
        med[0], med[1] / med[6], med[2] / med[6], med[3] / med[6], med[6], med[4], med[5]))
