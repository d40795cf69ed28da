"""Teacher-forced backward parity of the bf16 PRODUCT path, end to end.

The bf16 engine's gradients cannot be compared with a free-running oracle beyond cos ~0.9: thirty convolutions with bf16 storage
make the forward chaotic (ReLU masks, min-max arg-extrema and InstanceNorm statistics flip on one rounding; the oracle's OWN gradient
moves by cos 0.93-0.96 under a 1e-6 jitter, tests/test_oracle_kat.py::test_bf16_noise_floor), and the exact-parity mode (fp32 storage)
runs conv_kernel<float> / wgrad_kernel<float>, not the kernels the benchmark times (conv_thin, conv32, conv_dma, wgrad_dma, pw_gemm,
the fused shortcut / concat / InstanceNorm-backward launches, bstat epilogues, paired 2B-sample sweeps).

Here the oracle is TEACHER-FORCED: the HIP engine runs its real train_step (product schedule: two lanes, side streams, paired sweeps,
every fusion on, discriminator noise and channel dropout ON with explicit tensors); every tensor it stored in the forward pass -- all
120 generator and 16 discriminator convolution outputs, the four generated volumes, the logits -- is copied into the oracle's forward
(oracle.vangan_oracle.TEACHER: value replaced, gradient straight-through), so both sides differentiate the SAME forward state; torch
autograd through the oracle then gives the gradients the HIP backward has to produce, tensor by tensor.  What still differs: bf16
storage of the HIP activation gradients (one rounding per layer on the way down), fp32 summation order, InstanceNorm statistics from
(sum, sum of squares) instead of two passes.

Stated tolerance (measured values are printed): every parameter tensor of all four networks relative L2 <= 8e-2 and cosine >= 0.997
(measured: all but the generators' stem tensors <= 3e-2 / >= 0.999).  Round 5 had loosened it for the 16-element stem.short.w to 2e-1
(3e-2 .. 1.2e-1 run to run, blamed on the order of the float atomics); round 6 found the cause -- the closed form took xhat from the
STORED shortcut output, whose 16-bit rounding its 1/w amplified for channels with a small kernel weight (float64 sums over the stored
tensors reproduced the old kernel to 1e-5: it was never the summation) -- and removed it (vg_stem_short_bwd: xhat from the fp32 volume,
deterministic).  What is left for that tensor is the bf16 backward sweep above it, and the test now says so in two assertions: the kernel
against float64 sums over the stored tensors at 2e-3, and the end-to-end bound STEM_SHORT_E2E below with its measured distribution;
tensors whose gradient is analytically ~0 (biases in front of an InstanceNorm) absolutely, <= 5e-3 of the network's largest gradient
norm; whole-network cosine >= 0.9995 (measured 0.99998-1.00000, rel 4e-4 discriminators / 3e-3 - 7e-3 generators).  A dropped term in a fused launch (first-writer bits, a missing accumulate, the wrong half of a paired
tensor) moves single tensors by O(1) and fails this."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vangan_oracle as O  # noqa: E402
from test_gpu_nets import grad_report, perturb, rel_l2  # noqa: E402


def _gen_keys():
    ks = [('stem.conv1', ('stem', 'c1')), ('stem', ('stem', 'out'))]          # (the stem's shortcut is not stored: nets._STEM_FUSED)
    for b in ['enc%d' % e for e in range(1, 5)] + ['dec%d' % d for d in (3, 2, 1, 0)]:
        ks += [(b + '.cb1', (b, 'r')), (b + '.short', (b, 'sc')), (b, (b, 'out'))]
    ks += [('bridge.cb1', ('bridge', 'b1')), ('bridge.cb2', ('bridge', 'b2'))]
    return ks


def _teacher_from_engine(eng, B):
    """key -> NCDHW float tensor on the host, from the engine's stored forward tensors."""
    T = {}
    ctxs = eng._fwd_ctx
    for app in ('G_IS.a', 'G_SI.a', 'G_IS.b', 'G_SI.b'):
        c = ctxs[app]
        for key, (blk, field) in _gen_keys():
            T['%s/%s' % (app, key)] = O.to_ncdhw(c[blk][field].data.float().cpu())
        T['%s/y' % app] = O.to_ncdhw(c['y'].float().cpu())
    for d, logits in (('D_S', eng._aux['logits_S']), ('D_I', eng._aux['logits_I'])):
        c = ctxs[d]
        for i, k in enumerate(('conv0', 'down0', 'down1', 'down2')):
            a = O.to_ncdhw(c['acts'][i].data.float().cpu())
            T['%s.real/%s' % (d, k)], T['%s.fake/%s' % (d, k)] = a[:B], a[B:]
        lg = O.to_ncdhw(logits.float().cpu())
        T['%s.real/logits' % d], T['%s.fake/logits' % d] = lg[:B], lg[B:]
    return T


def _run(dims, B, seed, env=None):
    from van_gan_amd import VanGan
    dev = torch.device('cuda:0')
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        import importlib
        import van_gan_amd.ops as ops_mod
        import van_gan_amd.vangan as vg_mod
        if env:                                   # the schedule switches are read at import
            importlib.reload(ops_mod); importlib.reload(vg_mod)
        eng = vg_mod.VanGan(dims, batch_size=B, n_devices=1, device='cuda:0', seed=0, layer_noise=0.1, dropout_rate=0.2)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    P = {k: perturb(v, 40 + i) for i, (k, v) in enumerate(O.make_models(0).items())}
    eng.load_weights(P)
    rI, rS = O.synth_volumes(B, *dims, seed=seed)
    # explicit stochastic tensors: the same GaussianNoise / SpatialDropout3D draws on both sides
    g = torch.Generator().manual_seed(seed + 1)
    noise_h, drop_h, noise_o, drop_o = {}, {}, {}, {}
    for d, disc in (('S', eng.disc_S), ('I', eng.disc_I)):
        nz = {k: (torch.randn(shp, generator=g) * 0.1).to(torch.bfloat16) for k, shp in disc.noise_shapes(2 * B).items()}
        dp = {k: (torch.rand(2 * B, c, generator=g) >= 0.2).float() / 0.8 for k, c in (('down0', 128), ('down1', 256), ('down2', 512))}
        noise_h[d] = {k: t.to(dev) for k, t in nz.items()}
        drop_h[d] = {k: t.to(dev) for k, t in dp.items()}
        for half, sl in (('real', slice(0, B)), ('fake', slice(B, 2 * B))):
            noise_o['%s_%s' % (d, half)] = {k: t[sl].float() for k, t in nz.items()}
            drop_o['%s_%s' % (d, half)] = {k: t[sl] for k, t in dp.items()}
    res = eng.train_step(rI.to(dev), rS.to(dev), noise=noise_h, drop=drop_h, apply=False)
    torch.cuda.synchronize()
    got = eng.export_grads()
    T = _teacher_from_engine(eng, B)
    used, drift = set(), {}

    def teacher(key, x):
        t = T[key]                                # a missing key is a test bug: fail loudly
        assert t.shape == x.shape, (key, tuple(t.shape), tuple(x.shape))
        used.add(key)
        drift[key] = float((t.double() - x.detach().double()).norm() / (t.double().norm() + 1e-30))
        return t

    O.TEACHER = teacher
    try:
        ref, grads, aux = O.train_step(P, {}, rI, rS, O.Cfg(B, 1), noise=noise_o, drop=drop_o, q=O.bf16_round, apply=False)
    finally:
        O.TEACHER = None
    assert used == set(T), sorted(set(T) - used)[:5]
    # the oracle, fed the HIP tensors layer by layer, must reproduce each NEXT stored tensor to one rounding: the forward wiring
    worst = sorted(drift.items(), key=lambda kv: -kv[1])[:5]
    print('teacher-forced forward: worst per-tensor drift (oracle layer on HIP inputs vs HIP stored)', worst)
    assert worst[0][1] < 2e-2, worst
    for k in O.RESULT_KEYS:
        print('   %-24s hip %.6f  oracle(teacher-forced) %.6f' % (k, res[k], ref[k]))
        assert abs(res[k] - ref[k]) <= 2e-3 * abs(ref[k]) + 1e-5, k
    _run.eng = eng                      # the engine of the last run (its stored tensors: stem_short_from_stored)
    return got, grads


def stem_short_from_stored(eng, net):
    """d loss / d stem.short.w in float64 from the tensors the engine STORED (the gradient of the stem output as the backward sweep left
    it, the input volume): gamma * eps * rs^3 * sum_v d_out * (x - mean x) per channel, summed over the sweep's contexts (one 2B-sample
    context of a paired sweep, or the two applications').  What vg_stem_short_bwd has to reproduce to rounding."""
    return _stem_short_f64(eng.stores[net], eng._bwd_ctx[net])


def _stem_short_f64(st, ctxs):
    w = st.param('stem.short.w').flatten().float().to(torch.bfloat16).double()
    gamma = st.param('stem.short.in.gamma').double()
    tot = torch.zeros_like(w)
    for c in ctxs:
        ctx = c['stem']
        g = ctx['out'].grad.double()
        N, C = g.shape[0], g.shape[-1]
        g = g.reshape(N, -1, C)
        x = ctx['sx'].x0.double().reshape(N, -1, 1)
        xc = x - x.mean(1, keepdim=True)
        rs = (w[None, :] ** 2 * (xc ** 2).mean(1) + 1e-3).rsqrt()      # [N, C]
        tot += 1e-3 * gamma * (rs ** 3 * (g * xc).sum(1)).sum(0)
    return tot.cpu()


# The ONE tensor with a bound of its own against the oracle, and why it is not a loosened check: d loss / d stem.short.w is
# gamma * eps * rs^3 * sum_v d_out * (x - mean x), a sum whose terms cancel to ~1e-4 of their size in a train step, over a gradient d_out
# that reaches the stem after a whole backward sweep in bf16 storage (the oracle differentiates in fp32).  (a) The KERNEL is held to
# float64 sums over the very tensors the engine stored, at 2e-3 (measured 1e-5 .. 1e-7; deterministic) -- that pins its arithmetic;
# (b) against the oracle the error is that of the bf16 sweep above it, not of this kernel: float64 explicit InstanceNorm backward +
# weight gradient from the same stored tensors differs from the oracle by as much (tools/r06_stem_probe.py), 100 samples measured
# min 0.2 %, median 2.2 %, max 9.4 % (tools/r06_flake.py, profiles/r06_teacher_flake_25runs.txt) -- bound 1.5e-1 / cos 0.985.
STEM_SHORT_E2E = (1.5e-1, 0.985)


def _check(got, grads, label, eng=None):
    for net in ('disc_I', 'disc_S', 'gen_IS', 'gen_SI'):
        special = None
        if net.startswith('gen'):
            special = {'stem.short.w': STEM_SHORT_E2E}
            if eng is not None and getattr(eng, '_bwd_ctx', None):
                ref64 = stem_short_from_stored(eng, net)
                err = float((got[net]['stem.short.w'].double().flatten() - ref64).norm() / ref64.norm())
                print('%s stem.short.w: kernel vs float64 sums over the stored tensors: rel %.2e' % (net, err))
                assert err <= 2e-3, (net, err)
        cos = grad_report(got[net], grads[net], '%s %s (teacher-forced)' % (label, net), rel_tol=8e-2, cos_tol=0.997, abs_tol=5e-3, special=special)
        assert cos >= 0.9995, (net, cos)


def test_teacher_forced_train_step_32_b2():
    got, grads = _run((32, 32, 32), 2, seed=1234)
    _check(got, grads, 'product schedule', _run.eng)


def test_unfused_schedule_teacher_forced_32_b2():
    """The same check for the engine with the fused launches, the paired sweeps and the LDS-DMA data gradients OFF
    (VG_FUSE_CONCAT=0 VG_FUSE_CONCAT_NORM=0 VG_BSTAT=0 VG_PAIR_BWD=0 VG_CONV_DMA=0): both schedules are held to the oracle at their OWN
    forward state.  (They cannot be compared with each other directly: the InstanceNorm sums are float atomics, two runs of the bf16
    forward differ in a last bit somewhere and the generator amplifies that to cos ~0.98 between their gradients -- measured.)"""
    try:
        got_b, grads_b = _run((32, 32, 32), 2, seed=77, env={'VG_FUSE_CONCAT': '0', 'VG_FUSE_CONCAT_NORM': '0', 'VG_BSTAT': '0', 'VG_PAIR_BWD': '0',
                                                            'VG_CONV_DMA': '0'})
    finally:
        import importlib
        import van_gan_amd.ops as ops_mod
        import van_gan_amd.vangan as vg_mod
        importlib.reload(ops_mod); importlib.reload(vg_mod)          # back to the defaults for the tests that follow
    _check(got_b, grads_b, 'unfused schedule', _run.eng)


def test_teacher_forced_generator_128x128x64():
    """One generator application at BASELINE config 3's patch size, bf16 product kernels (conv_thin / conv32 / conv_dma / wgrad_dma /
    pw_gemm at their true launch shapes): all 116 parameter gradients against autograd through the teacher-forced oracle."""
    from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs
    from van_gan_amd.ops import Arena
    dev = torch.device('cuda:0')
    dims, N = (128, 128, 64), 1
    P = perturb(O.init_params(O.gen_param_specs(), 11), 12)
    st = ParamStore(gen_param_specs(), dev)
    st.load(P)
    net = ResUNet(st, dims, torch.bfloat16)
    net.pack()
    S = dims[0] * dims[1] * dims[2]
    ar = Arena(int(N * S * 6000) + (1 << 30), dev)
    x, _ = O.synth_volumes(N, *dims, seed=5)
    y = torch.zeros(N, *dims, 1, device=dev)
    ctx = net.forward(ar, x.to(dev), y)
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(y.shape, generator=g) / y.numel()
    st.g.zero_()
    net.backward(ar, ctx, gy.to(dev))
    torch.cuda.synchronize()
    ref64 = _stem_short_f64(st, [ctx])
    err64 = float((st.export(st.g)['stem.short.w'].double().flatten() - ref64).norm() / ref64.norm())
    print('stem.short.w: kernel vs float64 sums over the stored tensors: rel %.2e' % err64)
    assert err64 <= 2e-3, err64
    T = {key: O.to_ncdhw(ctx[blk][field].data.float().cpu()) for key, (blk, field) in _gen_keys()}
    T['y'] = O.to_ncdhw(y.float().cpu())
    used = set()

    def teacher(key, t):
        used.add(key)
        return T[key]

    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    O.TEACHER = teacher
    try:
        yr = O.resunet_forward(Pr, x, q=O.bf16_round)
    finally:
        O.TEACHER = None
    assert used == set(T)
    (yr * gy).sum().backward()
    # abs_tol: the biases in front of an InstanceNorm have an analytically zero gradient -- the sum of the layer's output gradient over
    # 1 M voxels; with that gradient STORED in bf16 the rounding errors do not cancel as the exact values do (measured 1.7e-2 of the
    # largest tensor norm on stem.conv1.b, which sums the full-resolution 16-channel gradient)
    cos = grad_report(st.export(st.g), {k: v.grad for k, v in Pr.items()}, 'generator 128x128x64 bf16 (teacher-forced)', rel_tol=8e-2, cos_tol=0.997,
                      abs_tol=4e-2, special={'stem.short.w': STEM_SHORT_E2E})
    assert cos >= 0.9995, cos
