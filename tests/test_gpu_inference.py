"""Sliding-window inference (SURVEY 8 a26 / BASELINE config 5): GPU stitch vs the numpy restatement of
GanMonitor.stitch_subvolumes driven by the oracle generator.  fp32 storage mode => tolerance 0.05 on the 0..255 output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import stitch_oracle as S  # noqa: E402
from oracle import vangan_oracle as O  # noqa: E402


@pytest.mark.parametrize('process_img', [False, True])
def test_stitch_subvolumes_matches_oracle(process_img):
    from van_gan_amd import VanGan
    k = (32, 32, 32)
    eng = VanGan(k, batch_size=4, device='cuda:0', seed=5, precision='fp32')
    P = eng.export_weights()['gen_IS']
    g = torch.Generator().manual_seed(3)
    vol = torch.rand(56, 48, 40, 1, generator=g) * 2 - 1

    def gen(a):
        with torch.no_grad():
            return O.resunet_forward(P, torch.from_numpy(np.ascontiguousarray(a)).float()).numpy()

    ref = S.stitch_subvolumes(gen, vol.numpy(), (1,) + k + (1,), stride=(20, 20, 16), complete=True, padFactor=0.25,
                              process_img=process_img)
    got = eng.stitch_subvolumes('gen_IS', vol, k, stride=(20, 20, 16), complete=True, padFactor=0.25,
                                process_img=process_img, window_batch=3).cpu().numpy()
    assert got.shape == ref.shape == (56, 48, 40, 1)
    assert not np.isnan(ref).any()
    err = np.abs(got - ref).max()
    print('stitch max abs err (0..255 scale): %.4f' % err)
    assert err < 0.05


def test_fp16_generator_matches_oracle_with_fp16_storage_points():
    """BASELINE config 5 names fp16: the generator's forward in libvangan_hip_h.so (IEEE half storage, fp32 accumulation) against
    the oracle with every stored tensor and the weights rounded to fp16 at the same points (q hook); and against the unrounded
    fp32 oracle, where fp16 (11 significand bits) must sit well below the bf16 path's 1.8e-2 noise floor."""
    from van_gan_amd import VanGan
    k = (32, 32, 32)
    eng = VanGan(k, batch_size=2, device='cuda:0', seed=7)
    P = eng.export_weights()['gen_IS']
    x, _ = O.synth_volumes(2, *k, seed=21)
    net = eng.fp16_generator('gen_IS')
    ar = eng.arena
    ar.reset()
    y = ar.alloc((2,) + k + (1,), torch.float32)
    xin = ar.alloc((2,) + k + (1,), torch.float32)
    xin.copy_(x)
    from van_gan_amd import ops
    with ops.Fp16():
        net.forward(ar, xin, y, save=False)
    torch.cuda.synchronize()
    assert net.L['stem.cb'].f_wp.dtype == torch.float16

    def q16(t):
        return t.half().float()
    with torch.no_grad():
        ref16 = O.resunet_forward(P, x, q=q16)
        ref32 = O.resunet_forward(P, x)
    got = y.cpu()
    e16 = float((got - ref16).norm() / ref16.norm())
    e32 = float((got - ref32).norm() / ref32.norm())
    # the same forward on the engine's bf16 path: fp16 keeps three more significand bits and must come out clearly closer
    yb = ar.alloc((2,) + k + (1,), torch.float32)
    eng.gen_IS.forward(ar, xin, yb, save=False)
    torch.cuda.synchronize()
    eb = float((yb.cpu() - ref32).norm() / ref32.norm())
    print('fp16 generator: rel L2 vs fp16-rounded oracle %.3e, vs fp32 oracle %.3e (bf16 path vs fp32 oracle: %.3e)' % (e16, e32, eb))
    assert e16 < 1e-2 and e32 < 1e-2 and e32 < 0.6 * eb


def test_window_origins_follow_reference_loop():
    from van_gan_amd.inference import window_origins
    # 256x256x128 volume, padFactor 0.1, stride 50 (post_training.py:38-39): 306x306x152 padded -> 5 x 5 x 2 windows
    assert window_origins(306, 128, 50) == [0, 50, 100, 150, 178] and window_origins(152, 128, 50) == [0, 24]
    # clamped duplicates are kept (they are counted twice by pix_tracker, as in the reference)
    assert window_origins(128, 128, 25) == [0, 0]


def test_stitch_subvolumes_with_the_resnet_generator():
    """The same sliding-window path over an engine built with the ResNet generators."""
    from van_gan_amd import VanGan
    k = (32, 32, 32)
    eng = VanGan(k, batch_size=4, device='cuda:0', seed=6, precision='fp32', generator='resnet')
    P = eng.export_weights()['gen_IS']
    g = torch.Generator().manual_seed(4)
    vol = torch.rand(56, 48, 40, 1, generator=g) * 2 - 1

    def gen(a):
        with torch.no_grad():
            return O.resnet_forward(P, torch.from_numpy(np.ascontiguousarray(a)).float()).numpy()

    ref = S.stitch_subvolumes(gen, vol.numpy(), (1,) + k + (1,), stride=(20, 20, 16), complete=True, padFactor=0.25, process_img=True)
    got = eng.stitch_subvolumes('gen_IS', vol, k, stride=(20, 20, 16), complete=True, padFactor=0.25, process_img=True,
                                window_batch=3).cpu().numpy()
    err = np.abs(got - ref).max()
    print('stitch (resnet generator) max abs err (0..255 scale): %.4f' % err)
    assert got.shape == ref.shape and err < 0.05


def test_fp16_stitch_with_the_resnet_generator():
    """fp16 storage build (libvangan_hip_h.so) over the ResNet generator: finite, and within fp16 rounding of the engine's fp32 result."""
    from van_gan_amd import VanGan
    k = (32, 32, 32)
    eng = VanGan(k, batch_size=4, device='cuda:0', seed=7, precision='fp32', generator='resnet')
    g = torch.Generator().manual_seed(5)
    vol = torch.rand(48, 40, 32, 1, generator=g) * 2 - 1
    kw = dict(stride=(16, 16, 16), complete=True, padFactor=0.25, process_img=True, window_batch=3)
    ref = eng.stitch_subvolumes('gen_IS', vol, k, **kw).cpu().numpy()
    got = eng.stitch_subvolumes('gen_IS', vol, k, precision='fp16', **kw).cpu().numpy()
    err = np.abs(got - ref).max()
    print('fp16 stitch (resnet generator) vs fp32: max abs err %.3f on the 0..255 scale' % err)
    assert np.isfinite(got).all() and got.shape == ref.shape and err < 4.0        # measured 1.9 (0.7 % of the range: 26 fp16-stored layers, min-max stretched)
