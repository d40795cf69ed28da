"""CPU: the third-party ("TP") pieces of the oracle cross-checked against INDEPENDENT implementations that ship in this image.

The oracle (oracle/vangan_oracle.py) restates TensorFlow / TensorFlow-Addons semantics from their documentation; TensorFlow
itself cannot be installed here and the reference has no golden vectors, so parity with the reference stays "unpinned"
(DESIGN section 4).  What CAN be done is to take the single author out of the loop: every TP building block below is compared
with a library implementation written by someone else -- torch.nn.functional (instance_norm, max_pool3d, conv3d, pad,
binary_cross_entropy, leaky_relu), torch autograd (amax / amin tie splitting), torch.optim.Adam, scipy.ndimage (grey
erosion / dilation with an explicit footprint).  Reference lines each block stands for are cited per test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ref as R
from oracle import vangan_oracle as O


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)


def test_instance_norm_vs_torch_functional():
    """tfa.layers.InstanceNormalization (resunet_model.py:36, building_blocks.py:190): per-(n,c) biased variance, eps 1e-3."""
    x = _rand(2, 5, 6, 7, 8, seed=1) * 3 + 0.7          # NCDHW
    gamma, beta = _rand(5, seed=2), _rand(5, seed=3)
    got = O.instance_norm(x, gamma, beta)
    ref = F.instance_norm(x, weight=gamma, bias=beta, eps=1e-3)
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-12)
    # the numpy restatement is channels-last
    got_np = R.instance_norm(x.permute(0, 2, 3, 4, 1).numpy(), gamma.numpy(), beta.numpy())
    assert np.allclose(got_np, ref.permute(0, 2, 3, 4, 1).numpy(), rtol=1e-12, atol=1e-12)


def test_reflection_pad_and_same_padding_vs_torch():
    """ReflectionPadding3D (building_blocks.py:30-39) = mirror without the edge; Keras 'same' for k4 s1 pads (1, 2), for k3 s1
    (1, 1), for k1 s2 on an even extent (0, 0), for k4 s2 (1, 1)."""
    x = _rand(1, 2, 4, 5, 6, seed=4)
    assert torch.equal(O.reflect_pad1(x), F.pad(x, (1, 1, 1, 1, 1, 1), mode='reflect'))
    assert np.array_equal(R.reflect_pad1(x.permute(0, 2, 3, 4, 1).numpy()), F.pad(x, (1, 1, 1, 1, 1, 1), mode='reflect').permute(0, 2, 3, 4, 1).numpy())
    for (n, k, s), want in (((16, 4, 1), (1, 2)), ((16, 3, 1), (1, 1)), ((16, 1, 2), (0, 0)), ((16, 4, 2), (1, 1)), ((7, 3, 2), (1, 1))):
        assert O._same_pads(n, k, s) == want and R.same_pads(n, k, s) == want
    # conv3d 'same' with k4: an explicitly padded F.conv3d (cross-correlation, un-flipped kernel)
    w = _rand(4, 4, 4, 3, 5, seed=5)                  # DHWIO
    xi = _rand(1, 3, 6, 6, 6, seed=6)
    got = O.conv3d(xi, w, None, 1, 'same')
    ref = F.conv3d(F.pad(xi, (1, 2, 1, 2, 1, 2)), w.permute(4, 3, 0, 1, 2))
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-12)


def _footprint19():
    fp = np.zeros((3, 3, 3), bool)
    fp[1, :, :] = True; fp[:, 1, :] = True; fp[:, :, 1] = True
    return fp


def test_soft_erode_dilate_vs_max_pool3d_and_scipy():
    """soft_erode, 3-D branch (clDice_func.py:23-26): min of three MaxPool3D(-x) with pool sizes (3,3,1), (3,1,3), (1,3,3),
    'same' (out-of-range taps ignored) = grey erosion with the 19-voxel union of the three axis-aligned 3x3 planes;
    soft_dilate (clDice_func.py:41-42) = MaxPool3D 3x3x3."""
    import scipy.ndimage as ndi
    x = torch.rand(2, 7, 8, 9, generator=torch.Generator().manual_seed(7), dtype=torch.float64)        # [B, D, H, W]
    x5 = x[:, None]                                                                                     # torch pools NCDHW
    e = O.soft_erode(x)
    pools = [-F.max_pool3d(-x5, ks, 1, pd) for ks, pd in (((3, 3, 1), (1, 1, 0)), ((3, 1, 3), (1, 0, 1)), ((1, 3, 3), (0, 1, 1)))]
    assert torch.equal(e, torch.minimum(torch.minimum(pools[0], pools[1]), pools[2])[:, 0])
    assert int(_footprint19().sum()) == 19
    for b in range(2):
        ref = ndi.grey_erosion(x[b].numpy(), footprint=_footprint19(), mode='constant', cval=np.inf)
        assert np.array_equal(e[b].numpy(), ref)
        assert np.array_equal(R.soft_erode(x[b:b + 1].numpy())[0], ref)
    d = O.soft_dilate(x)
    assert torch.equal(d, F.max_pool3d(x5, 3, 1, 1)[:, 0])
    for b in range(2):
        assert np.array_equal(d[b].numpy(), ndi.grey_dilation(x[b].numpy(), size=(3, 3, 3), mode='constant', cval=-np.inf))


def test_pooling_gradient_vs_torch_max_pool3d_autograd():
    """MaxPool3D backward routes each window's gradient to ONE arg-max (TP); on tie-free data torch's max_pool3d autograd is the
    same function, so the oracle's hand-written pooling backward (vangan_oracle._Pool3) must equal it for the dilation and,
    through min = -max(-x) and the three-way minimum, for the erosion."""
    x = torch.rand(1, 6, 7, 8, generator=torch.Generator().manual_seed(8), dtype=torch.float64)
    gy = _rand(1, 6, 7, 8, seed=9)
    xa = x.clone().requires_grad_(True)
    (O.soft_dilate(xa) * gy).sum().backward()
    xb = x.clone().requires_grad_(True)
    (F.max_pool3d(xb[:, None], 3, 1, 1)[:, 0] * gy).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=0, atol=1e-14)
    xa = x.clone().requires_grad_(True)
    (O.soft_erode(xa) * gy).sum().backward()
    xb = x.clone().requires_grad_(True)
    pools = [-F.max_pool3d(-xb[:, None], ks, 1, pd) for ks, pd in (((3, 3, 1), (1, 1, 0)), ((3, 1, 3), (1, 0, 1)), ((1, 3, 3), (0, 1, 1)))]
    (torch.minimum(torch.minimum(pools[0], pools[1]), pools[2])[:, 0] * gy).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=0, atol=1e-14)


def test_keras_bce_vs_torch_bce_away_from_the_clip():
    """Keras BinaryCrossentropy(from_logits=False) (loss_functions.py:185-190): clip p to [1e-7, 1 - 1e-7] and add 1e-7 inside
    both logs.  Away from the clip this is torch's binary_cross_entropy up to 1e-7 / p."""
    g = torch.Generator().manual_seed(10)
    p = torch.rand(4, 5, 6, 7, 1, generator=g, dtype=torch.float64) * 0.9 + 0.05
    t = (torch.rand(4, 5, 6, 7, 1, generator=g, dtype=torch.float64) > 0.5).double()
    got = O.keras_bce(t, p)                          # mean over the channel axis (size 1)
    ref = F.binary_cross_entropy(p, t, reduction='none').mean(-1)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=0, atol=3e-6)
    assert np.allclose(R.keras_bce(t.numpy(), p.numpy()), ref.numpy(), rtol=0, atol=3e-6)
    # at the clip: p = 0 and p = 1 give finite values of about -log(2e-7) / -log(1 + ...) -- torch clamps its log at -100 instead
    edge = O.keras_bce(torch.tensor([[1.0], [0.0]], dtype=torch.float64), torch.tensor([[0.0], [1.0]], dtype=torch.float64))
    assert torch.allclose(edge, torch.full_like(edge, -math.log(2e-7)), rtol=1e-9)


def test_min_max_norm_tie_gradients_vs_torch_amax_amin():
    """min_max_norm_tf (utils.py:27-48): gradient through reduce_max / reduce_min is split evenly among ties (TP:
    math_grad._MinOrMaxGrad) -- torch.amax / torch.amin document the same rule."""
    x = torch.tensor([[0.5, 2.0, 2.0, -1.0, -1.0, -1.0, 0.25, 2.0]], dtype=torch.float64).reshape(1, 2, 2, 2, 1)
    gy = _rand(1, 2, 2, 2, 1, seed=11)
    xa = x.clone().requires_grad_(True)
    (O.min_max_norm(xa) * gy).sum().backward()
    xb = x.clone().requires_grad_(True)
    mx, mn = torch.amax(xb, dim=(1, 2, 3, 4), keepdim=True), torch.amin(xb, dim=(1, 2, 3, 4), keepdim=True)
    (((xb - mn) / (mx - mn)) * gy).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-12, atol=1e-14)


def test_leaky_relu_and_lsgan_terms():
    """LeakyReLU(0.2) (building_blocks.py:193-195) and the LSGAN terms on constant logits (loss_functions.py:273-274, 306-308)."""
    x = _rand(3, 4, seed=12)
    assert np.allclose(R.lrelu(x.numpy()), F.leaky_relu(x, 0.2).numpy())
    logits = torch.full((2, 3, 3, 3, 1), 0.25, dtype=torch.float64)
    assert math.isclose(float(O.mse(torch.ones_like(logits), logits, 2.0)), 2 * 0.75 ** 2 / 2.0, rel_tol=1e-12)


def test_adam_with_per_variable_clipnorm_vs_torch_optim():
    """tf.keras Adam(2e-4, 0.5, 0.9, clipnorm=100) (vangan.py:220-235): per-VARIABLE clip_by_norm, then the standard Adam moment
    recursion with epsilon outside the bias-corrected root (lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); w -= lr_t m / (sqrt(v) + eps)).
    torch.optim.Adam corrects v before the root, i.e. its eps sits at eps_torch * sqrt(1 - b2^t) in Keras' form: driven with
    eps_torch = 1e-7 / sqrt(1 - b2^t) per step it must reproduce the Keras update to rounding; with a fixed 1e-7 the two differ by
    O(lr * eps / |g|) (both checked)."""
    shapes = {'a': (7, 5), 'b': (11,), 'c': (3, 3, 3, 2, 4)}

    def run(eps_of_step):
        g0 = torch.Generator().manual_seed(13)
        P = {k: torch.randn(sh, generator=g0, dtype=torch.float64) for k, sh in shapes.items()}
        tp = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        opt = torch.optim.Adam(list(tp.values()), lr=2e-4, betas=(0.5, 0.9), eps=1e-7)
        state, mine, worst = {}, {k: v.clone() for k, v in P.items()}, 0.0
        for step in range(3):
            grads = {k: torch.randn(shapes[k], generator=g0, dtype=torch.float64) * (40.0 if k == 'a' else 1.0) for k in shapes}
            assert float(grads['a'].norm()) > 100.0 > float(grads['b'].norm())          # 'a' is clipped, 'b' is not
            O.adam_step(mine, grads, state)
            for k in shapes:
                n = float(grads[k].norm())
                tp[k].grad = grads[k] * (100.0 / n) if n > 100.0 else grads[k].clone()
            for grp in opt.param_groups:
                grp['eps'] = eps_of_step(step + 1)
            opt.step()
            worst = max(worst, max(float((mine[k] - tp[k].detach()).abs().max()) for k in shapes))
        return worst

    assert run(lambda t: 1e-7 / math.sqrt(1.0 - 0.9 ** t)) <= 1e-14        # Keras' epsilon placement: identical to rounding
    w_fixed = run(lambda t: 1e-7)                                          # torch's own placement: O(lr * eps / |g|) apart
    assert 1e-12 < w_fixed < 1e-6
