"""The oracle's hand-written and third-party-semantics pieces against float64 finite differences and against a second,
independent implementation (oracle/np_ref.py) -- CPU only.

oracle/vangan_oracle.py is PARITY UNPINNED (TensorFlow cannot run here, the reference has no vectors): what can be
removed is the self-referential part.  Everything differentiable is checked by central differences on <= 8^3 volumes;
the non-differentiable tie rules (TP, quoted from memory of the TF 2.10 / Keras sources) are checked as SUBGRADIENT
properties that any valid rule must satisfy plus the documented choice itself:
  * MaxPool3D gradient (TF core/kernels/maxpooling_op.cc, SpatialMaxPoolWithArgMaxHelper): the whole gradient of a
    window goes to ONE arg-max, the first in scan order;  tf.minimum(x, y) gradient goes to x where x <= y
    (math_grad.py _MinimumGrad: xmask = x <= y), which makes soft_erode prefer the earlier pooling plane;
  * reduce_min / reduce_max gradient (math_grad.py _MinOrMaxGrad): split EQUALLY among the ties;
  * clip_by_norm per variable AFTER the all-reduce, Adam epsilon 1e-7 outside the square root (optimizer_v2/adam.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ref as R
from oracle import vangan_oracle as O

torch.set_default_dtype(torch.float32)


def _fd(f, x, eps=1e-6):
    """central-difference gradient of scalar f at x (float64 tensor), element by element"""
    g = torch.zeros_like(x)
    xf, gf = x.view(-1), g.view(-1)
    for i in range(xf.numel()):
        old = float(xf[i])
        xf[i] = old + eps; fp = float(f(x))
        xf[i] = old - eps; fm = float(f(x))
        xf[i] = old
        gf[i] = (fp - fm) / (2 * eps)
    return g


def _rand(shape, seed, lo=0.05, hi=0.95):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * (hi - lo) + lo


@pytest.mark.parametrize('fn', ['erode', 'dilate', 'open'])
def test_pooling_backward_matches_finite_differences_on_continuous_data(fn):
    """_Pool3.backward (hand-written) on tie-free input: min/max are differentiable there."""
    f = {'erode': O.soft_erode, 'dilate': O.soft_dilate, 'open': O.soft_open}[fn]
    x = _rand((1, 5, 6, 4), 1).requires_grad_(True)
    w = _rand((1, 5, 6, 4), 2)
    (f(x) * w).sum().backward()
    num = _fd(lambda t: (f(t) * w).sum(), x.detach().clone())
    assert torch.allclose(x.grad, num, atol=1e-7), float((x.grad - num).abs().max())


def test_soft_skeleton_and_cldice_match_finite_differences():
    """soft_skel (31 erosions / 16 dilations at iters=15; 4 here) and the full Dice + clDice loss, continuous data."""
    p = _rand((1, 6, 6, 6, 1), 3).requires_grad_(True)
    t = (_rand((1, 6, 6, 6, 1), 4) > 0.6).double()
    loss = lambda q: O.soft_dice_cldice(t, q, iters=4)
    loss(p).backward()
    num = _fd(loss, p.detach().clone())
    assert torch.allclose(p.grad, num, atol=2e-7), float((p.grad - num).abs().max())
    w = _rand((1, 6, 6, 6), 5)
    x = _rand((1, 6, 6, 6), 6).requires_grad_(True)
    (O.soft_skel(x, 3) * w).sum().backward()
    num = _fd(lambda q: (O.soft_skel(q, 3) * w).sum(), x.detach().clone())
    assert torch.allclose(x.grad, num, atol=2e-7)


def test_pooling_backward_on_binary_ties_is_a_valid_one_winner_subgradient():
    """Binary labels tie everywhere.  Whatever the tie rule, a one-arg-max routing must (1) conserve the gradient mass of
    every window, (2) deliver only to voxels that attain the extremum, (3) be reproduced by the plain-Python restatement of
    the documented rule: first attaining candidate in the order p1 (3,3,1) window, p2 (3,1,3), p3 (1,3,3), raster inside."""
    x = (_rand((1, 5, 5, 5), 7) > 0.5).double().requires_grad_(True)
    g = _rand((1, 5, 5, 5), 8)
    for f, offs, is_min in ((O.soft_erode, O._erode_offsets(), True), (O.soft_dilate, O._dilate_offsets(), False)):
        x.grad = None
        y = f(x)
        yd = y.detach()
        (y * g).sum().backward()
        assert math.isclose(float(x.grad.sum()), float(g.sum()), rel_tol=1e-12)
        exp = torch.zeros_like(x)
        xd = x.detach()
        _, D, H, W = xd.shape
        for d in range(D):
            for h in range(H):
                for w in range(W):
                    for (a, b, c) in offs:                      # first candidate that attains the window's extremum wins
                        dd, hh, ww = d + a, h + b, w + c
                        if 0 <= dd < D and 0 <= hh < H and 0 <= ww < W and float(xd[0, dd, hh, ww]) == float(yd[0, d, h, w]):
                            exp[0, dd, hh, ww] += g[0, d, h, w]
                            break
        assert torch.allclose(x.grad, exp, rtol=1e-13, atol=1e-13), float((x.grad - exp).abs().max())   # summation order differs
    # forward values agree with the independent numpy pooling (sorted offsets there: the VALUE does not depend on order)
    assert np.array_equal(O.soft_erode(x.detach()).numpy(), R.soft_erode(x.detach().numpy()))


def test_min_max_norm_gradient_with_ties_splits_equally():
    x = _rand((2, 3, 4, 4, 1), 9)
    x[0, 0, 0, 0, 0] = x[0, 1, 1, 1, 0] = x[0, 2, 3, 3, 0] = 2.0         # three tied maxima in sample 0
    x[1, 0, 0, 1, 0] = x[1, 2, 2, 2, 0] = -1.0                           # two tied minima in sample 1
    w = _rand(x.shape, 10)
    xr = x.clone().requires_grad_(True)
    (O.min_max_norm(xr) * w).sum().backward()
    # non-tied elements: ordinary derivative
    num = _fd(lambda t: (O.min_max_norm(t) * w).sum(), x.clone())
    tied = torch.zeros_like(x, dtype=torch.bool)
    for idx in ((0, 0, 0, 0, 0), (0, 1, 1, 1, 0), (0, 2, 3, 3, 0), (1, 0, 0, 1, 0), (1, 2, 2, 2, 0)):
        tied[idx] = True
    assert torch.allclose(xr.grad[~tied], num[~tied], atol=1e-7)

    # tied elements: shifting ALL ties of one extremum together is differentiable; its derivative is the SUM of their
    # gradients -- and the rule splits the extremum's part equally, so within a tie group the gradients differ only by the
    # elements' own direct term
    def shifted(t, group, e):
        t = t.clone()
        for idx in group:
            t[idx] += e
        return (O.min_max_norm(t) * w).sum()
    for group in (((0, 0, 0, 0, 0), (0, 1, 1, 1, 0), (0, 2, 3, 3, 0)), ((1, 0, 0, 1, 0), (1, 2, 2, 2, 0))):
        d = (float(shifted(x, group, 1e-6)) - float(shifted(x, group, -1e-6))) / 2e-6
        assert math.isclose(d, float(sum(xr.grad[i] for i in group)), rel_tol=1e-6, abs_tol=1e-8)
        n, k = group[0][0], len(group)
        mn, mx = float(x[n].min()), float(x[n].max())
        direct = [float(w[i]) / (mx - mn) for i in group]                # d/dx_i of the (x_i - min)/(max - min) numerator
        shared = [float(xr.grad[i]) - di for i, di in zip(group, direct)]
        assert max(shared) - min(shared) < 1e-9, shared                   # the extremum's share is the same for every tie


def test_ssim_and_instance_norm_match_finite_differences():
    t, p = _rand((1, 5, 5, 5, 1), 11), _rand((1, 5, 5, 5, 1), 12).requires_grad_(True)
    O.ssim_loss_3d(t, p).mean().backward()
    num = _fd(lambda q: O.ssim_loss_3d(t, q).mean(), p.detach().clone())
    assert torch.allclose(p.grad, num, atol=1e-8)
    x = (_rand((2, 3, 4, 4, 4), 13) * 4 - 2).requires_grad_(True)
    gm, bt = _rand((3,), 14, 0.5, 1.5), _rand((3,), 15, -0.5, 0.5)
    w = _rand(x.shape, 16)
    (O.instance_norm(x, gm, bt) * w).sum().backward()
    num = _fd(lambda q: (O.instance_norm(q, gm, bt) * w).sum(), x.detach().clone())
    assert torch.allclose(x.grad, num, atol=1e-7)
    # closed form of the independent implementation
    dx, dg, db = R.instance_norm_backward(x.detach().permute(0, 2, 3, 4, 1).numpy(), gm.numpy(), w.permute(0, 2, 3, 4, 1).numpy())
    assert np.allclose(dx, x.grad.permute(0, 2, 3, 4, 1).numpy(), atol=1e-10)


@pytest.mark.parametrize('stride', [1, 2])
def test_residual_block_backward_numpy_vs_autograd(stride):
    """Hand-derived numpy backward of one residual block (conv / reflect-pad transpose / InstanceNorm / ReLU / shortcut)
    against torch autograd through the oracle's _res_block."""
    rng = np.random.default_rng(3)
    ci, co, S = 8, 16, 8
    name = 'blk'
    shapes = {name + '.cb1.in.gamma': (ci,), name + '.cb1.in.beta': (ci,), name + '.cb1.conv.w': (3, 3, 3, ci, co),
              name + '.cb1.conv.b': (co,), name + '.cb2.in.gamma': (co,), name + '.cb2.in.beta': (co,),
              name + '.cb2.conv.w': (3, 3, 3, co, co), name + '.cb2.conv.b': (co,), name + '.short.w': (1, 1, 1, ci, co),
              name + '.short.b': (co,), name + '.short.in.gamma': (co,), name + '.short.in.beta': (co,)}
    pn = {k: rng.standard_normal(s) * (0.2 if k.endswith('.w') else 0.5) + (1.0 if k.endswith('gamma') else 0.0) for k, s in shapes.items()}
    x = rng.standard_normal((2, S, S, S, ci))
    so = S // stride
    gout = rng.standard_normal((2, so, so, so, co))
    dx, grads = R.res_block_backward(pn, name, x, stride, gout)
    pt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in pn.items()}
    xt = torch.from_numpy(x).requires_grad_(True)
    out = O._res_block(pt, name, O.to_ncdhw(xt), stride, None)
    assert np.allclose(O.to_ndhwc(out).detach().numpy(), R._res(pn, name, x, stride), atol=1e-10)
    (O.to_ndhwc(out) * torch.from_numpy(gout)).sum().backward()
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-9), float(np.abs(dx - xt.grad.numpy()).max())
    for k in shapes:
        assert np.allclose(grads[k], pt[k].grad.numpy(), atol=1e-8), k


def test_adam_step_against_scalar_restatement():
    """optimizer_v2 Adam with clipnorm per variable: an element-by-element Python loop over three steps, one tensor whose
    gradient norm (200) is clipped to 100 and one that is not."""
    rng = np.random.default_rng(5)
    w = {'a': torch.tensor(rng.standard_normal(6), dtype=torch.float64), 'b': torch.tensor(rng.standard_normal(4), dtype=torch.float64)}
    w0 = {k: v.clone().numpy() for k, v in w.items()}
    st = {}
    m = {k: np.zeros_like(v) for k, v in w0.items()}
    v2 = {k: np.zeros_like(v) for k, v in w0.items()}
    ww = {k: v.copy() for k, v in w0.items()}
    lr, b1, b2, eps, clip = 2e-4, 0.5, 0.9, 1e-7, 100.0
    for t in range(1, 4):
        ga = rng.standard_normal(6); ga *= 200.0 / np.linalg.norm(ga)
        gb = rng.standard_normal(4) * 0.01
        O.adam_step(w, {'a': torch.tensor(ga), 'b': torch.tensor(gb)}, st, lr, b1, b2, clip)
        for k, g in (('a', ga), ('b', gb)):
            nrm = math.sqrt(sum(float(e) ** 2 for e in g))
            gg = [float(e) * (clip / nrm) if nrm > clip else float(e) for e in g]
            lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            for i in range(len(gg)):
                m[k][i] = b1 * m[k][i] + (1 - b1) * gg[i]
                v2[k][i] = b2 * v2[k][i] + (1 - b2) * gg[i] * gg[i]
                ww[k][i] -= lr_t * m[k][i] / (math.sqrt(v2[k][i]) + eps)
    for k in w:
        assert np.allclose(w[k].numpy(), ww[k], rtol=1e-12, atol=1e-15), k
