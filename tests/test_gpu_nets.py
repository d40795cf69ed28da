"""GPU parity of the full networks and of VanGan.train_step against the CPU oracle (bf16 storage emulated by the
oracle's ``q`` hook; gradients flow straight through the rounding).

Stated tolerances (bf16 operands, fp32 accumulation, bf16 gradient tensors):
  forward volumes / logits : relative L2 <= 4e-2 and max abs error <= 0.3 on the tanh output (range [-1,1]).  This is
                             the measured bf16 noise floor of the 30-conv generator: re-running the ORACLE with bf16
                             storage and a 1e-6 relative jitter before each rounding moves its own output by
                             rel 1.8e-2 / max 0.14 at 32^3 (tests/test_oracle_kat.py::test_bf16_noise_floor);
                             fp32 vs bf16 storage differ by rel 2.9e-2.  Per-kernel exactness is pinned separately in
                             test_gpu_ops.py on identical operands.
  losses                   : relative error <= 3e-2
  parameter gradients      : discriminator (5 convs): per tensor relative L2 <= 1e-1 and cosine >= 0.99 (biases in front of
                             an InstanceNorm have an analytically ZERO gradient and are checked absolutely).
                             generator (30 convs, 28 InstanceNorms, ReLU masks): bf16 rounding makes the gradient itself
                             chaotic -- the oracle's OWN gradient moves by rel 0.30 / cos 0.95 under the 1e-6 jitter --
                             (and, through two generators in series, gen_SI's by rel 0.6-0.9 / cos 0.61-0.83), so only the
                             whole-network cosine is asserted against that measured floor; per-tensor numbers are printed.
                             The exact backward arithmetic of every kernel is pinned in test_gpu_ops.py, and the
                             end-to-end wiring at fp32 tolerance in test_gpu_fp32.py (fp32 storage mode).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vangan_oracle as O  # noqa: E402


def _dev():
    return torch.device('cuda:0')


def rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def perturb(P, seed):
    g = torch.Generator().manual_seed(seed)
    for k, v in P.items():
        if k.endswith('.b') or k.endswith('.beta'):
            v.add_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith('.gamma'):
            v.mul_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
    return P


def grad_report(got: dict, ref: dict, label: str, rel_tol=1e-1, cos_tol=0.99, check=True, abs_tol=2e-3, special=None):
    """special: {tensor name: (rel_tol, cos_tol)} for tensors with a stated tolerance of their own."""
    gmax = max(float(v.double().norm()) for v in ref.values())
    rows, bad = [], []
    for k in ref:
        a, b = got[k].double().cpu().flatten(), ref[k].double().flatten()
        nb = float(b.norm())
        err = float((a - b).norm())
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        rows.append((k, nb, err / (nb + 1e-30), cos))
        if nb < 1e-2 * gmax:         # small tensors: absolute check against the largest gradient norm
            if err > abs_tol * gmax:
                bad.append((k, 'abs', err, gmax))
        elif err / nb > (special or {}).get(k, (rel_tol, cos_tol))[0] or cos < (special or {}).get(k, (rel_tol, cos_tol))[1]:
            bad.append((k, 'rel', err / nb, cos))
    worst = sorted(rows, key=lambda r: -r[2])[:8]
    print('\n[%s] worst relative gradient errors:' % label)
    for r in worst:
        print('   %-28s |ref|=%.3e rel=%.3e cos=%.5f' % r)
    tot_a = torch.cat([got[k].double().cpu().flatten() for k in ref]); tot_b = torch.cat([ref[k].double().flatten() for k in ref])
    print('   whole-network gradient: rel %.3e cos %.5f' % (float((tot_a - tot_b).norm() / tot_b.norm()),
                                                             float(tot_a @ tot_b / (tot_a.norm() * tot_b.norm()))))
    cos_all = float(tot_a @ tot_b / (tot_a.norm() * tot_b.norm()))
    if check:
        assert not bad, '%s: gradient mismatch %s' % (label, bad[:6])
    return cos_all


def test_generator_forward_backward_32():
    from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs
    from van_gan_amd.ops import Arena
    dev = _dev()
    dims, N = (32, 32, 32), 1
    P = perturb(O.init_params(O.gen_param_specs(), 11), 12)
    st = ParamStore(gen_param_specs(), dev)
    st.load(P)
    net = ResUNet(st, dims)
    net.pack()
    ar = Arena(1 << 30, dev)
    x, _ = O.synth_volumes(N, *dims, seed=5)
    y = torch.zeros(N, *dims, 1, device=dev)
    ctx = net.forward(ar, x.to(dev), y)
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    taps = {}
    yr = O.resunet_forward(Pr, x, q=O.bf16_round, taps=taps)
    # intermediate taps localise a failure
    for name, key in (('stem.conv1', None), ('stem', 'stem'), ('enc1', 'enc1'), ('enc4', 'enc4'), ('dec0', 'dec0')):
        if key is None:
            got = ctx['stem']['c1'].data
        else:
            got = ctx[key]['out'].data
        r = O.to_ndhwc(taps[name]).detach()
        e = rel_l2(got.float(), r)
        print('tap %-10s rel_l2 %.3e' % (name, e))
        assert e < 3e-2, name
    err = (y.cpu() - yr.detach()).abs().max()
    print('generator output: max abs err %.3e, rel l2 %.3e' % (float(err), rel_l2(y, yr.detach())))
    assert err < 0.3 and rel_l2(y, yr.detach()) < 4e-2
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(y.shape, generator=g) / y.numel()
    (yr * gy).sum().backward()
    st.g.zero_()
    net.backward(ar, ctx, gy.to(dev))
    torch.cuda.synchronize()
    cos = grad_report(st.export(st.g), {k: v.grad for k, v in Pr.items()}, 'generator', check=False)
    assert cos > 0.93        # oracle-vs-jittered-oracle floor on this input: 0.95 (test_oracle_kat.py)


def test_discriminator_forward_backward_32():
    from van_gan_amd.nets import ParamStore, PatchGAN, disc_param_specs
    from van_gan_amd.ops import Arena
    dev = _dev()
    dims, N = (32, 32, 32), 2
    P = perturb(O.init_params(O.disc_param_specs(), 21), 22)
    st = ParamStore(disc_param_specs(), dev)
    st.load(P)
    net = PatchGAN(st, dims)
    net.pack()
    ar = Arena(1 << 30, dev)
    x, _ = O.synth_volumes(N, *dims, seed=6)
    g = torch.Generator().manual_seed(4)
    noise = {k: (torch.randn(shp, generator=g) * 0.1).to(torch.bfloat16) for k, shp in net.noise_shapes(N).items()}
    drop = {k: ((torch.rand(N, c, generator=g) > 0.2).float() / 0.8) for k, c in (('down0', 128), ('down1', 256), ('down2', 512))}
    logits = torch.zeros(N, 4, 4, 4, 1, device=dev)
    ctx = net.forward(ar, x.to(dev), logits, {k: v.to(dev) for k, v in noise.items()}, {k: v.to(dev) for k, v in drop.items()})
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    lr_ = O.disc_forward(Pr, xr, {k: v.float() for k, v in noise.items()}, drop, q=O.bf16_round)
    print('disc logits rel l2 %.3e' % rel_l2(logits, lr_.detach()))
    assert rel_l2(logits, lr_.detach()) < 2e-2
    gl = torch.randn(logits.shape, generator=g)
    (lr_ * gl).sum().backward()
    st.g.zero_()
    dx = torch.zeros(N, *dims, 1, device=dev)
    net.backward(ar, ctx, gl.to(dev), 0, N, wgrad=True, dx=dx)
    torch.cuda.synchronize()
    grad_report(st.export(st.g), {k: v.grad for k, v in Pr.items()}, 'discriminator')
    e = rel_l2(dx, xr.grad)
    print('disc input gradient rel l2 %.3e' % e)
    assert e < 8e-2
    # sub-batch sweep (generator-loss path): samples [1,2) only, no weight gradients
    before = st.g.clone()
    dx1 = torch.zeros(1, *dims, 1, device=dev)
    net.backward(ar, ctx, gl[1:].to(dev), 1, 2, wgrad=False, dx=dx1)
    torch.cuda.synchronize()
    assert torch.equal(before, st.g)
    assert rel_l2(dx1, xr.grad[1:]) < 8e-2


def _engine_vs_oracle(dims, B, steps=1):
    from van_gan_amd import VanGan
    dev = _dev()
    eng = VanGan(dims, batch_size=B, n_devices=1, device='cuda:0', seed=0, layer_noise=0.0, dropout_rate=0.0)
    P = {k: perturb(v, 40 + i) for i, (k, v) in enumerate(O.make_models(0).items())}
    eng.load_weights(P)
    cfg = O.Cfg(B, 1)
    rI, rS = O.synth_volumes(B, *dims, seed=1234)
    state = {}
    for s in range(steps):
        res = eng.train_step(rI.to(dev), rS.to(dev), noise={}, drop={})
        ref, grads, aux = O.train_step(P, state, rI, rS, cfg, q=O.bf16_round)
        print('\nstep %d' % s)
        for k in O.RESULT_KEYS:
            print('   %-24s hip %.6f  oracle %.6f' % (k, res[k], ref[k]))
        for k in ('fake_S', 'fake_I', 'cycled_S', 'cycled_I'):
            e = (eng._aux[k].cpu() - aux[k]).abs().max()
            r = rel_l2(eng._aux[k], aux[k])
            print('   %-10s max abs err %.3e  rel l2 %.3e' % (k, float(e), r))
            # cycled_* went through two generators; from the 2nd step on the two runs also hold slightly different weights
            # (Adam's first step is ~lr*sign(g): elements with ~0 gradient flip), which the chaotic forward amplifies
            assert r < ((4e-2 if k.startswith('fake') else 2.5e-1) if s == 0 else 0.6), k
        for k in O.RESULT_KEYS:
            assert abs(res[k] - ref[k]) <= (3e-2 if s == 0 else 1e-1) * abs(ref[k]) + 1e-4, k
        got = eng.export_grads()
        for net in ('disc_I', 'disc_S', 'gen_IS', 'gen_SI'):
            cos = grad_report(got[net], grads[net], net, check=False)
            # oracle-vs-jittered-oracle floor at 32^3 (measured): gen_IS cos 0.93-0.96, gen_SI 0.61-0.83
            floor = {'disc_I': 0.99, 'disc_S': 0.99, 'gen_IS': 0.85, 'gen_SI': 0.5}[net]
            if s == 0:
                assert cos > floor, (net, cos)
        W = eng.export_weights()
        for net in W:
            for n in W[net]:
                d = (W[net][n] - P[net][n]).abs().max()
                assert d < 3 * 2e-4 * (s + 1) + 1e-6, (net, n, float(d))     # |Adam step| <= ~lr per step
    return eng


def test_train_step_32_b1():
    _engine_vs_oracle((32, 32, 32), 1)


def test_train_step_32_b2_two_steps():
    _engine_vs_oracle((32, 32, 32), 2, steps=2)


def test_test_step_matches_forward_losses():
    from van_gan_amd import VanGan
    dev = _dev()
    dims, B = (32, 32, 32), 1
    eng = VanGan(dims, batch_size=B, device='cuda:0', seed=3)
    P = eng.export_weights()
    rI, rS = O.synth_volumes(B, *dims, seed=99)
    res = eng.test_step(rI.to(dev), rS.to(dev))
    ref = O.test_step(P, rI, rS, O.Cfg(B, 1), q=O.bf16_round)
    for k in O.RESULT_KEYS:
        assert abs(res[k] - ref[k]) <= 3e-2 * abs(ref[k]) + 1e-4, (k, res[k], ref[k])
