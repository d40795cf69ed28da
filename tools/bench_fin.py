"""Development aid: one forward convolution launch with and without the InstanceNorm-finalisation tail (vg_fin_desc), HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from van_gan_amd import ops
from van_gan_amd.nets import ParamStore, Norm
from van_gan_amd.ops import ConvLayer, Src, Arena

dev = torch.device('cuda:0')
ops.set_device(0)
for (cin, cout, dims, N) in ((32, 32, (64, 64, 64), 2), (16, 16, (128, 128, 128), 1), (128, 128, (16, 16, 16), 2), (256, 256, (8, 8, 8), 2)):
    st = ParamStore([('c.w', (3, 3, 3, cin, cout), 'he_normal'), ('c.b', (cout,), 'zeros'), ('n.gamma', (cout,), 'ones'), ('n.beta', (cout,), 'zeros')], dev)
    st.param('c.w').normal_(0, 0.05)
    lay = ConvLayer(st, 'c', 3, cin, cout, 1, 'reflect', True, dims, need_dgrad=False)
    lay.pack()
    nrm = Norm(st, 'n', cout)
    ar = Arena(1 << 30, dev)
    x = torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16)
    src = Src(x, (N,) + dims, cin)
    out = torch.empty(N, *dims, cout, dtype=torch.bfloat16, device=dev)
    R = 40
    for mode in ('plain', 'tail', 'plain+finalize'):
        ar.reset()
        sums = [ar.alloc((8, N, cout, 2), torch.float32, zero=True) for _ in range(R + 2)]
        stt = nrm.state(ar, N)
        fins = [ops.fin_desc(ar, float(dims[0] * dims[1] * dims[2]), [nrm.job(stt)]) for _ in range(R + 2)]
        def go(i):
            lay.forward(src, out, sums=sums[i], fin=fins[i] if mode == 'tail' else None)
            if mode == 'plain+finalize':
                ops.in_finalize(sums[i], cout, float(dims[0] * dims[1] * dims[2]), nrm.gamma, nrm.beta, N, stt['scale'], stt['shift'], stt['mean'], stt['rstd'])
        go(R); go(R + 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(R):
            go(i)
        e1.record(); torch.cuda.synchronize()
        print('%3d->%3d %s N=%d  %-15s %7.1f us per launch   %s' % (cin, cout, dims, N, mode, e0.elapsed_time(e1) / R * 1e3, ops.conv_variant(lay._fwd_desc(src)) if mode == 'plain' else ''))
