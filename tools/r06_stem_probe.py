"""Where does the error of the stem shortcut's kernel gradient come from?  One generator application (bf16 product kernels), then from the
tensors the engine STORED (d_out = gradient of the stem output, sc = shortcut output, x) the same gradient in float64 two ways -- the
closed form of vg_in_scale_invariant_wgrad and the explicit InstanceNorm backward + 1x1x1 weight gradient -- against the HIP value and the
teacher-forced oracle's.  usage: python tools/r06_stem_probe.py [D H W] [runs]"""
import os
import sys

os.environ['VG_STEM_FUSED'] = '0'        # the probe reads the STORED shortcut tensor (round 5's forward; the default no longer materialises it)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import vangan_oracle as O  # noqa: E402
from test_gpu_nets import perturb  # noqa: E402
from test_gpu_teacher import _gen_keys  # noqa: E402
from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs  # noqa: E402
from van_gan_amd.ops import Arena  # noqa: E402

dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 32, 32)
runs = int(sys.argv[4]) if len(sys.argv) >= 5 else 3
N = 2
dev = torch.device('cuda:0')
P = perturb(O.init_params(O.gen_param_specs(), 11), 12)
x, _ = O.synth_volumes(N, *dims, seed=5)
g = torch.Generator().manual_seed(3)
gy = torch.randn(N, *dims, 1, generator=g) / (N * dims[0] * dims[1] * dims[2])
EPS = 1e-3


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


for run in range(runs):
    st = ParamStore(gen_param_specs(), dev)
    st.load(P)
    net = ResUNet(st, dims, torch.bfloat16)
    net.pack()
    ar = Arena(int(N * dims[0] * dims[1] * dims[2] * 6000) + (1 << 30), dev)
    y = torch.zeros(N, *dims, 1, device=dev)
    ctx = net.forward(ar, x.to(dev), y)
    st.g.zero_()
    net.backward(ar, ctx, gy.to(dev))
    torch.cuda.synchronize()
    hip = st.export(st.g)['stem.short.w'].flatten().double()
    s = ctx['stem']
    d_out = s['out'].grad.double().reshape(N, -1, 16)            # [N, S, C]
    sc = s['sc'].data.double().reshape(N, -1, 16)
    xs = x.to(dev).double().reshape(N, -1, 1)
    w = st.param('stem.short.w').flatten().double()
    gamma = st.param('stem.short.in.gamma').double()
    wq = w.float().to(torch.bfloat16).double()
    mean = sc.mean(1, keepdim=True)
    var = ((sc - mean) ** 2).mean(1, keepdim=True)
    rstd = (var + EPS).rsqrt()
    xh = (sc - mean) * rstd
    r1 = (d_out * xh).sum(1)                  # [N, C]
    closed = (EPS * gamma * (rstd[:, 0] ** 2 * r1).sum(0) / wq)
    dn = d_out * gamma
    dsc = rstd * (dn - dn.mean(1, keepdim=True) - xh * (dn * xh).mean(1, keepdim=True))
    explicit = (dsc * xs).sum((0, 1))
    # the same closed form with xhat from x itself (sc = w x + b exactly): what float64 autograd sees without the storage rounding of sc
    mx = xs.mean(1, keepdim=True)
    vx = ((xs - mx) ** 2).mean(1, keepdim=True)
    rs_x = (wq ** 2 * vx + EPS).rsqrt()                        # [N,1,C]
    xh_x = (xs - mx) * wq * rs_x
    r1x = (d_out * xh_x).sum(1)
    closed_x = (EPS * gamma * (rs_x[:, 0] ** 2 * r1x).sum(0) / wq)
    # oracle, teacher-forced
    T = {key: O.to_ncdhw(ctx[blk][field].data.float().cpu()) for key, (blk, field) in _gen_keys()}
    T['y'] = O.to_ncdhw(y.float().cpu())
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    O.TEACHER = lambda key, t: T[key]
    try:
        yr = O.resunet_forward(Pr, x, q=O.bf16_round)
    finally:
        O.TEACHER = None
    (yr * gy).sum().backward()
    ref = Pr['stem.short.w'].grad.flatten().double()
    hip, closed, explicit, closed_x = hip.cpu(), closed.cpu(), explicit.cpu(), closed_x.cpu()
    print('run %d  dims %s' % (run, dims,))
    print('  hip      vs oracle %.4f   hip vs closed64 %.5f' % (rel(hip, ref), rel(hip, closed)))
    print('  closed64 vs oracle %.4f   (float64 sums over the stored bf16 d_out and sc)' % rel(closed, ref))
    print('  explicit64 vs oracle %.4f (IN backward + 1x1x1 weight gradient in float64 from the same stored tensors)' % rel(explicit, ref))
    print('  closed64 with xhat from x vs oracle %.4f' % rel(closed_x, ref))
    if run == 0:
        for c in range(16):
            print('    c%02d w % .4f  oracle % .4e hip % .4e closed % .4e explicit % .4e closed_x % .4e' % (c, float(w[c]), float(ref[c]), float(hip[c]), float(closed[c]),
                                                                                                float(explicit[c]), float(closed_x[c])))
    del ar, ctx
