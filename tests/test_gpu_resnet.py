"""SURVEY 8(f)4, first part: the ResNet generator (generator.py:7-73, as vangan.py:88-97,127-134 configures it) as a forward network on
the HIP kernels (van_gan_amd.nets.ResNetGenerator) against the oracle's restatement (oracle.vangan_oracle.resnet_forward).

Tolerances: exact-parity mode (fp32 storage): every stored tensor and the output relative L2 <= 2e-3 (measured ~1e-5: fp32 summation
order); bf16 product mode against the oracle with bf16 storage points: stored tensors <= 3e-2, output <= 4e-2 (the bf16 noise floor of
a 20-convolution chain, as for the ResUNet in test_gpu_nets.py).  Covered on the way: a 7^3 single-channel convolution (W-packed, 49
taps), stride-2 convolutions on odd grids (28 -> 14 -> 7 -> 4), the virtual UpSampling3D without a concat partner under a 4^3 'same'
convolution, vg_affine_add with a pending InstanceNorm + ReLU + dropout on its first operand, and the 7^3 head as seven accumulating
49-tap chunks with the tanh in the last one."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vangan_oracle as O  # noqa: E402
from test_gpu_nets import perturb, rel_l2  # noqa: E402


def _run(dims, N, dtype, with_drop):
    from van_gan_amd.nets import ParamStore, ResNetGenerator, resnet_param_specs
    from van_gan_amd.ops import Arena
    dev = torch.device('cuda:0')
    P = perturb(O.init_params(O.resnet_param_specs(), 21), 22)
    st = ParamStore(resnet_param_specs(), dev)
    st.load(P)
    net = ResNetGenerator(st, dims, dtype)
    net.pack()
    S = dims[0] * dims[1] * dims[2]
    ar = Arena(int(N * S * 4000) + (256 << 20), dev)
    x, _ = O.synth_volumes(N, *dims, seed=9)
    g = torch.Generator().manual_seed(4)
    drop = None
    if with_drop:
        drop = {'c7': (torch.rand(N, 32, generator=g) >= 0.5).float() / 0.5}
        for i, c in enumerate((64, 128, 256)):
            drop['down%d' % i] = (torch.rand(N, c, generator=g) >= 0.2).float() / 0.8
    y = torch.zeros(N, *dims, 1, device=dev)
    taps = net.forward(ar, x.to(dev), y, None if drop is None else {k: v.to(dev) for k, v in drop.items()})
    torch.cuda.synchronize()
    rt = {}
    if dtype == torch.float32:
        yr = O.resnet_forward({k: v.double() for k, v in P.items()}, x.double(), drop=None if drop is None else {k: v.double() for k, v in drop.items()}, taps=rt)
        tol_t, tol_y = 2e-3, 2e-3
    else:
        yr = O.resnet_forward(P, x, q=O.bf16_round, drop=drop, taps=rt)
        tol_t, tol_y = 3e-2, 4e-2
    worst = 0.0
    for k, ref in rt.items():
        got = taps[k].data if hasattr(taps[k], 'data') and not isinstance(taps[k], torch.Tensor) else taps[k]
        e = rel_l2(got.float(), O.to_ndhwc(ref))
        worst = max(worst, e)
        assert e < tol_t, (k, e)
    e = rel_l2(y, yr)
    print('ResNet generator %s %s N=%d drop=%s: worst stored tensor rel %.2e, output rel %.2e max abs %.2e'
          % ('fp32' if dtype == torch.float32 else 'bf16', dims, N, with_drop, worst, e, float((y.cpu().double() - yr.double()).abs().max())))
    assert e < tol_y and y.shape == x.shape and bool(torch.isfinite(y).all())


def test_resnet_generator_fp32_32():
    _run((32, 32, 32), 2, torch.float32, with_drop=True)


def test_resnet_generator_bf16_32():
    _run((32, 32, 32), 2, torch.bfloat16, with_drop=False)


def test_resnet_generator_bf16_64_kernels_of_larger_grids():
    """64^3: the layers leave the small-grid kernel variants (conv32_kernel / wider tiles), as they do at 128^3."""
    _run((64, 64, 64), 1, torch.bfloat16, with_drop=True)


def _run_backward(dims, N, dtype, with_drop):
    """Parameter gradients of sum(y * gy) through ResNetGenerator.backward against autograd through the oracle's restatement."""
    from van_gan_amd.nets import ParamStore, ResNetGenerator, resnet_param_specs
    from van_gan_amd.ops import Arena
    from test_gpu_nets import grad_report
    dev = torch.device('cuda:0')
    P = perturb(O.init_params(O.resnet_param_specs(), 31), 32)
    st = ParamStore(resnet_param_specs(), dev)
    st.load(P)
    net = ResNetGenerator(st, dims, dtype)
    net.pack()
    S = dims[0] * dims[1] * dims[2]
    ar = Arena(int(N * S * 9000) + (512 << 20), dev)
    x, _ = O.synth_volumes(N, *dims, seed=13)
    g = torch.Generator().manual_seed(6)
    drop = None
    if with_drop:
        drop = {'c7': (torch.rand(N, 32, generator=g) >= 0.5).float() / 0.5}
        for i, c in enumerate((64, 128, 256)):
            drop['down%d' % i] = (torch.rand(N, c, generator=g) >= 0.2).float() / 0.8
    y = torch.zeros(N, *dims, 1, device=dev)
    taps = net.forward(ar, x.to(dev), y, None if drop is None else {k: v.to(dev) for k, v in drop.items()})
    gy = torch.randn(y.shape, generator=g) / y.numel() ** 0.5
    st.g.zero_()
    net.backward(ar, taps, gy.to(dev))
    torch.cuda.synchronize()
    if dtype == torch.float32:
        Pr = {k: v.double().requires_grad_(True) for k, v in P.items()}
        yr = O.resnet_forward(Pr, x.double(), drop=None if drop is None else {k: v.double() for k, v in drop.items()})
        (yr * gy.double()).sum().backward()
    else:
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        yr = O.resnet_forward(Pr, x, q=O.bf16_round, drop=drop)
        (yr * gy).sum().backward()
    assert rel_l2(y, yr.detach()) < (2e-3 if dtype == torch.float32 else 4e-2)
    got = st.export(st.g)
    ref = {k: v.grad for k, v in Pr.items()}
    assert all(bool(torch.isfinite(v).all()) for v in got.values())
    return grad_report(got, ref, 'ResNet generator %s %s' % ('fp32' if dtype == torch.float32 else 'bf16', dims),
                       rel_tol=2e-2 if dtype == torch.float32 else 1.0, cos_tol=0.9995 if dtype == torch.float32 else 0.0,
                       abs_tol=1e-3 if dtype == torch.float32 else 1.0, check=dtype == torch.float32)


def test_resnet_generator_backward_fp32_32():
    """Exact-parity mode (fp32 storage): every parameter gradient tensor rel <= 2e-2, cos >= 0.9995 against float64 autograd (measured:
    worst 5.6e-3 -- fp32 accumulation over 32^3 voxels in another order, as in test_gpu_fp32.py; the whole-network cosine > 0.99999),
    with SpatialDropout3D multipliers on c7 / down0..2 -- the backward of (f)4's first variant: tanh, the 7^3 head's seven chunks (weight
    gradient per chunk into its taps of the DHWIO tensor, data gradients accumulating), UpSampling3D's sum-pool, the residual Add's two
    branches, stride-2 data gradients on odd grids, the W-packed 7^3 stem's weight gradient."""
    cos = _run_backward((32, 32, 32), 2, torch.float32, with_drop=True)
    assert cos > 0.99999


def _teacher_backward(dims, N, with_drop=True):
    """Teacher-forced backward parity of the ResNet generator's bf16 PRODUCT kernels (the method of tests/test_gpu_teacher.py): every
    tensor the HIP forward stored -- the 7^3 stem, the three stride-2 stages, both convolutions and the sum of each of the six residual
    blocks, the three upsampling convolutions, the output -- replaces the oracle's value at the same storage point (gradient
    straight-through), so InstanceNorm statistics and ReLU masks are those of the HIP forward state and autograd through the oracle
    yields, tensor by tensor, what the HIP backward has to produce for THAT state."""
    from van_gan_amd.nets import ParamStore, ResNetGenerator, resnet_param_specs
    from van_gan_amd.ops import Arena
    from test_gpu_nets import grad_report
    dev = torch.device('cuda:0')
    P = perturb(O.init_params(O.resnet_param_specs(), 31), 32)
    st = ParamStore(resnet_param_specs(), dev)
    st.load(P)
    net = ResNetGenerator(st, dims, torch.bfloat16)
    net.pack()
    S = dims[0] * dims[1] * dims[2]
    ar = Arena(int(N * S * 9000) + (512 << 20), dev)
    x, _ = O.synth_volumes(N, *dims, seed=13)
    g = torch.Generator().manual_seed(6)
    drop = None
    if with_drop:
        drop = {'c7': (torch.rand(N, 32, generator=g) >= 0.5).float() / 0.5}
        for i, c in enumerate((64, 128, 256)):
            drop['down%d' % i] = (torch.rand(N, c, generator=g) >= 0.2).float() / 0.8
    y = torch.zeros(N, *dims, 1, device=dev)
    taps = net.forward(ar, x.to(dev), y, None if drop is None else {k: v.to(dev) for k, v in drop.items()})
    gy = torch.randn(y.shape, generator=g) / y.numel() ** 0.5
    st.g.zero_()
    net.backward(ar, taps, gy.to(dev))
    torch.cuda.synchronize()
    val = lambda t: O.to_ncdhw((t.data if not isinstance(t, torch.Tensor) else t).float().cpu())
    T = {k: val(v) for k, v in taps.items() if k != '_ctx'}
    for (k, _s1, r1, _n1, _s2, r2, _n2) in taps['_ctx']['res']:
        T[k + '.c1'], T[k + '.c2'] = val(r1), val(r2)
    T['y'] = O.to_ncdhw(y.float().cpu())
    used, drift = set(), {}

    def teacher(key, t):
        used.add(key)
        drift[key] = float((T[key].double() - t.detach().double()).norm() / (T[key].double().norm() + 1e-30))
        return T[key]

    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    O.TEACHER = teacher
    try:
        yr = O.resnet_forward(Pr, x, q=O.bf16_round, drop=drop)
    finally:
        O.TEACHER = None
    assert used == set(T), sorted(set(T) ^ used)
    worst = sorted(drift.items(), key=lambda kv: -kv[1])[:4]
    print('teacher-forced ResNet forward: worst per-tensor drift (oracle layer on HIP inputs vs HIP stored)', worst)
    assert worst[0][1] < 2e-2, worst
    (yr * gy).sum().backward()
    got = st.export(st.g)
    assert all(bool(torch.isfinite(v).all()) for v in got.values())
    return grad_report(got, {k: v.grad for k, v in Pr.items()}, 'ResNet generator bf16 %s (teacher-forced)' % (dims,), rel_tol=8e-2,
                       cos_tol=0.997, abs_tol=4e-2)


def test_resnet_generator_backward_bf16_64_teacher_forced():
    """Product precision at 64^3 (the kernels of the larger grids: conv_dma data gradients, wgrad_dma, conv32, the thin-channel head in
    seven 49-tap chunks): all 38 parameter-gradient tensors rel <= 8e-2 / cos >= 0.997 (the tolerances of tests/test_gpu_teacher.py),
    whole-network cosine >= 0.9995.  Replaces round 4's free-running cos > 0.9 floor."""
    cos = _teacher_backward((64, 64, 64), 1)
    assert cos >= 0.9995, cos


def test_resnet_generator_backward_bf16_128_teacher_forced():
    """The generator at BASELINE's 128^3 patch (its launches there select seven kernel variants no smaller grid does:
    tests/test_variant_coverage.py::test_resnet_generator_walk): same tolerances.  ~1 minute of CPU oracle."""
    cos = _teacher_backward((128, 128, 128), 1)
    assert cos >= 0.9995, cos


def test_resnet_generator_backward_bf16_32_teacher_forced_batch2():
    cos = _teacher_backward((32, 32, 32), 2)
    assert cos >= 0.9995, cos


def _gen_masks(B, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for app in ('G_IS.a', 'G_SI.a', 'G_IS.b', 'G_SI.b'):
        d = {'c7': (torch.rand(B, 32, generator=g) >= 0.5).float() / 0.5}
        for i, c in enumerate((64, 128, 256)):
            d['down%d' % i] = (torch.rand(B, c, generator=g) >= 0.2).float() / 0.8
        out[app] = d
    return out


def test_train_step_with_resnet_generators_fp32_32():
    """VanGan(generator='resnet') -- the generator pair vangan.py:88-97,127-134 builds for gen_i2s = gen_s2i = 'resnet' -- one full
    train_step in exact-parity mode against the oracle's train_step with the same generators and the same SpatialDropout3D masks on all
    four applications: outputs and the ten result scalars rel <= 2e-3, every parameter gradient tensor rel <= 5e-2 / cos >= 0.999,
    whole-network cosine >= 0.9995 (the tolerances of test_gpu_fp32.py), Adam applied."""
    from van_gan_amd import VanGan
    dev = torch.device('cuda:0')
    dims, B = (32, 32, 32), 1
    eng = VanGan(dims, batch_size=B, n_devices=1, device='cuda:0', seed=0, layer_noise=0.0, dropout_rate=0.0, precision='fp32',
                 generator='resnet')
    rs, ds = O.resnet_param_specs(), O.disc_param_specs()
    P = {'gen_IS': O.init_params(rs, 50), 'gen_SI': O.init_params(rs, 51), 'disc_I': O.init_params(ds, 52), 'disc_S': O.init_params(ds, 53)}
    P = {k: perturb(v, 60 + i) for i, (k, v) in enumerate(P.items())}
    eng.load_weights(P)
    rI, rS = O.synth_volumes(B, *dims, seed=77)
    masks = _gen_masks(B, 5)
    res = eng.train_step(rI.to(dev), rS.to(dev), noise={}, drop={k: {n: t.to(dev) for n, t in v.items()} for k, v in masks.items()})
    Pd = {k: {n: t.double() for n, t in v.items()} for k, v in P.items()}
    ref, grads, aux = O.train_step(Pd, {}, rI.double(), rS.double(), O.Cfg(B, 1),
                                   drop={k: {n: t.double() for n, t in v.items()} for k, v in masks.items()})
    for k in ('fake_S', 'fake_I', 'cycled_S', 'cycled_I'):
        r = rel_l2(eng._aux[k], aux[k])
        print('   %-10s rel l2 %.3e' % (k, r))
        assert r < 2e-3, k
    for k in O.RESULT_KEYS:
        print('   %-24s hip %.6f  oracle %.6f' % (k, res[k], ref[k]))
        assert abs(res[k] - ref[k]) <= 2e-3 * abs(ref[k]) + 1e-6, k
    from test_gpu_nets import grad_report
    got = eng.export_grads()
    for net in ('disc_I', 'disc_S', 'gen_IS', 'gen_SI'):
        cos = grad_report(got[net], grads[net], net + ' (resnet generators) fp32', rel_tol=5e-2, cos_tol=0.999)
        assert cos > 0.9995, (net, cos)
    W = eng.export_weights()
    nbad = ntot = 0
    for net in W:
        for n in W[net]:
            d = (W[net][n].double() - Pd[net][n]).abs()
            nbad += int((d > 1e-3).sum()); ntot += d.numel()
    assert nbad == 0, (nbad, ntot)                       # one Adam step moves a weight by <= ~lr = 2e-4 ... 6.3e-4


def test_train_steps_with_resnet_generators_bf16_64():
    """Product precision, 64^3, batch 1, the engine's own noise / dropout streams (discriminator AND generator masks): three steps run,
    every result scalar is finite, the weights move, test_step (no dropout) works on the same engine."""
    from van_gan_amd import VanGan
    dev = torch.device('cuda:0')
    dims, B = (64, 64, 64), 1
    eng = VanGan(dims, batch_size=B, device='cuda:0', seed=1, generator='resnet')
    w0 = {k: s.w.clone() for k, s in eng.stores.items()}
    rI, rS = O.synth_volumes(B, *dims, seed=3)
    for _ in range(3):
        res = eng.train_step(rI.to(dev), rS.to(dev))
        assert all(v == v and abs(v) < 1e6 for v in res.values()), res
    t = eng.test_step(rI.to(dev), rS.to(dev))
    assert all(v == v for v in t.values())
    for k, s in eng.stores.items():
        d = float((s.w - w0[k]).abs().max())
        assert 1e-5 < d < 5e-3, (k, d)
