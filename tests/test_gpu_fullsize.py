"""GPU vs ORACLE at the FULL sizes of BASELINE configs 3 and 4 (the round-2 verdict: above 64^3 the engine was only compared
with itself).  The oracle's generator forward at 128 x 128 x 64 (75 GMAC) and discriminator forward at 128^3 (61 GMAC) run on
the host in float32 (tens of seconds on the box's cores); the engine runs the same networks in exact-parity mode (fp32
storage, f32 MFMA: the kernels / tiles / schedules of the product path) and, for the generator, in the bf16 product mode
against the oracle with bf16 storage points (q = bf16_round).

Tolerances: fp32 mode relative L2 <= 2e-3 (measured ~1e-5: only summation orders differ); bf16 mode <= 4e-2 (the bf16 noise
floor of 30 chained convolutions, tests/test_oracle_kat.py::test_bf16_noise_floor)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vangan_oracle as O  # noqa: E402
from test_gpu_nets import perturb, rel_l2  # noqa: E402


def _dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_generator_forward_128x128x64_vs_oracle(precision):
    """ResUNet (resunet_model.py:185-249) at the patch size of BASELINE config 3."""
    from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs
    from van_gan_amd.ops import Arena
    dev = _dev()
    dims, N = (128, 128, 64), 1
    dt = torch.float32 if precision == 'fp32' else torch.bfloat16
    P = perturb(O.init_params(O.gen_param_specs(), 21), 22)
    st = ParamStore(gen_param_specs(), dev)
    st.load(P)
    net = ResUNet(st, dims, dt)
    net.pack()
    ar = Arena((6 << 30) if precision == 'fp32' else (3 << 30), dev)
    x, _ = O.synth_volumes(N, *dims, seed=9)
    y = torch.zeros(N, *dims, 1, device=dev)
    net.forward(ar, x.to(dev), y, save=False)
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        yr = O.resunet_forward(P, x, q=None if precision == 'fp32' else O.bf16_round)
    e = rel_l2(y, yr)
    print('generator %s forward at %s: rel L2 vs oracle %.3e, max abs %.3e' % (precision, dims, e, float((y.cpu() - yr).abs().max())))
    assert torch.isfinite(y).all()
    assert e < (2e-3 if precision == 'fp32' else 4e-2)


def test_generator_backward_128x128x64_fp32_vs_oracle_autograd():
    """All 116 parameter gradients of one generator application at the config-3 patch size, exact-parity mode, against
    torch autograd through the oracle (float32 on the host): whole-network cosine >= 0.9995, relative L2 <= 2e-2."""
    from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs
    from van_gan_amd.ops import Arena
    dev = _dev()
    dims, N = (128, 128, 64), 1
    P = perturb(O.init_params(O.gen_param_specs(), 23), 24)
    st = ParamStore(gen_param_specs(), dev)
    st.load(P)
    net = ResUNet(st, dims, torch.float32)
    net.pack()
    ar = Arena(24 << 30, dev)
    x, _ = O.synth_volumes(N, *dims, seed=11)
    y = torch.zeros(N, *dims, 1, device=dev)
    ctx = net.forward(ar, x.to(dev), y)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)) / y.numel()
    st.g.zero_()
    net.backward(ar, ctx, gy.to(dev))
    torch.cuda.synchronize()
    got = st.export(st.g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yr = O.resunet_forward(Pr, x)
    (yr * gy).sum().backward()
    a = torch.cat([got[k].double().flatten() for k in Pr])
    b = torch.cat([Pr[k].grad.double().flatten() for k in Pr])
    cos = float((a @ b) / (a.norm() * b.norm()))
    rel = float((a - b).norm() / b.norm())
    print('generator fp32 gradients at %s: cosine %.6f, rel L2 %.3e' % (dims, cos, rel))
    assert cos > 0.9995 and rel < 2e-2


def test_discriminator_forward_128cubed_vs_oracle():
    """PatchGAN (discriminator.py:7-124) at 128^3 (BASELINE config 4's per-GPU patch), exact-parity mode, noise and dropout off."""
    from van_gan_amd.nets import ParamStore, PatchGAN, disc_param_specs
    from van_gan_amd.ops import Arena
    dev = _dev()
    dims, N = (128, 128, 128), 1
    P = perturb(O.init_params(O.disc_param_specs(), 31), 32)
    st = ParamStore(disc_param_specs(), dev)
    st.load(P)
    net = PatchGAN(st, dims, torch.float32)
    net.pack()
    ar = Arena(4 << 30, dev)
    x, _ = O.synth_volumes(N, *dims, seed=10)
    logits = torch.zeros(N, 16, 16, 16, 1, device=dev)
    net.forward(ar, x.to(dev), logits)
    torch.cuda.synchronize()
    with torch.no_grad():
        lr = O.disc_forward(P, x)
    e = rel_l2(logits, lr)
    print('discriminator fp32 forward at %s: rel L2 vs oracle %.3e' % (dims, e))
    assert e < 2e-3
