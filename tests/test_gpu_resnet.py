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
