"""Independent float64 numpy restatement of the forward arithmetic of the VAN-GAN hot path -- TEST
INFRASTRUCTURE ONLY (second, independent implementation used to cross-check oracle/vangan_oracle.py; only
tests/ may import it).  PARITY UNPINNED against TensorFlow (see vangan_oracle.py header).

Nothing here calls torch: convolutions are explicit sums over taps, pooling is explicit shifting, so an error
in the torch restatement's use of F.conv3d / F.pad / autograd helpers shows up as a disagreement.
All arrays are NDHWC like the reference; conv kernels DHWIO.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

IN_EPS = 1e-3
BCE_EPS = 1e-7


def reflect_pad1(x):
    """building_blocks.py:30-39 (tf.pad REFLECT)."""
    return np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)), mode='reflect')


def same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv3d(x, w, b=None, stride=1, padding='valid'):
    """Keras Conv3D (cross-correlation), explicit tap loop.  x: [N,D,H,W,Ci], w: [kd,kh,kw,Ci,Co]."""
    kd, kh, kw, ci, co = w.shape
    if padding == 'same':
        pads = [same_pads(x.shape[1 + a], (kd, kh, kw)[a], stride) for a in range(3)]
        x = np.pad(x, ((0, 0), pads[0], pads[1], pads[2], (0, 0)))
    N, D, H, W, _ = x.shape
    od, oh, ow = (D - kd) // stride + 1, (H - kh) // stride + 1, (W - kw) // stride + 1
    y = np.zeros((N, od, oh, ow, co), dtype=np.float64)
    for a in range(kd):
        for bb in range(kh):
            for c in range(kw):
                xs = x[:, a:a + (od - 1) * stride + 1:stride, bb:bb + (oh - 1) * stride + 1:stride,
                       c:c + (ow - 1) * stride + 1:stride, :]
                y += xs @ w[a, bb, c]
    if b is not None:
        y = y + b
    return y


def instance_norm(x, gamma, beta):
    mu = x.mean(axis=(1, 2, 3), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(1, 2, 3), keepdims=True)
    return (x - mu) / np.sqrt(var + IN_EPS) * gamma + beta


def relu(x):
    return np.maximum(x, 0.0)


def lrelu(x):
    return np.where(x > 0, x, 0.2 * x)


def _cb(p, name, x, stride):
    a = relu(instance_norm(x, p[name + '.in.gamma'], p[name + '.in.beta']))
    return conv3d(reflect_pad1(a), p[name + '.conv.w'], p[name + '.conv.b'], stride, 'valid')


def _res(p, name, x, stride):
    r = _cb(p, name + '.cb1', x, stride)
    r = _cb(p, name + '.cb2', r, 1)
    sc = conv3d(x, p[name + '.short.w'], p[name + '.short.b'], stride, 'same')
    return instance_norm(sc, p[name + '.short.in.gamma'], p[name + '.short.in.beta']) + r


def resunet_forward(p: Dict[str, np.ndarray], x, taps: Optional[dict] = None):
    """resunet_model.py:185-249.  taps: filled with the block outputs (stem, enc1..enc4, bridge, dec3..dec0), NDHWC."""
    c = conv3d(reflect_pad1(x), p['stem.conv1.w'], p['stem.conv1.b'], 1, 'valid')
    c = _cb(p, 'stem.cb', c, 1)
    sc = conv3d(x, p['stem.short.w'], p['stem.short.b'], 1, 'same')
    h = c + instance_norm(sc, p['stem.short.in.gamma'], p['stem.short.in.beta'])
    skips = [h]
    if taps is not None:
        taps['stem'] = h
    for e in range(1, 5):
        h = _res(p, 'enc%d' % e, h, 2)
        skips.append(h)
        if taps is not None:
            taps['enc%d' % e] = h
    h = _cb(p, 'bridge.cb1', h, 1)
    h = _cb(p, 'bridge.cb2', h, 1)
    if taps is not None:
        taps['bridge'] = h
    for d in (3, 2, 1, 0):
        up = h.repeat(2, axis=1).repeat(2, axis=2).repeat(2, axis=3)
        h = _res(p, 'dec%d' % d, np.concatenate([up, skips[d]], axis=-1), 1)
        if taps is not None:
            taps['dec%d' % d] = h
    return np.tanh(conv3d(h, p['out.w'], p['out.b'], 1, 'same'))


def disc_forward(p, x, noise: Optional[dict] = None, drop: Optional[dict] = None):
    """discriminator.py:7-124."""
    noise, drop = noise or {}, drop or {}

    def nz(k, t):
        return t + noise[k] if noise.get(k) is not None else t

    def act(k, t, g, b):
        t = lrelu(instance_norm(t, g, b))
        if drop.get(k) is not None:
            t = t * drop[k][:, None, None, None, :]
        return t

    h = conv3d(nz('conv0', reflect_pad1(x)), p['conv0.w'], p['conv0.b'], 2, 'valid')
    h = act(None, h, p['conv0.in.gamma'], p['conv0.in.beta'])
    for i in range(3):
        k = 'down%d' % i
        if i < 2:
            h = conv3d(nz(k, reflect_pad1(h)), p[k + '.w'], None, 2, 'valid')
        else:
            h = conv3d(nz(k, h), p[k + '.w'], None, 1, 'same')
        h = act(k, h, p[k + '.in.gamma'], p[k + '.in.beta'])
    return conv3d(nz('out', h), p['out.w'], p['out.b'], 1, 'same')


def min_max_norm(x):
    mn = x.min(axis=(1, 2, 3, 4), keepdims=True)
    mx = x.max(axis=(1, 2, 3, 4), keepdims=True)
    return (x - mn) / (mx - mn)


def keras_bce(t, p):
    pc = np.clip(p, BCE_EPS, 1 - BCE_EPS)
    return (-(t * np.log(pc + BCE_EPS) + (1 - t) * np.log(1 - pc + BCE_EPS))).mean(axis=-1)


def gaussian_taps(size=3, sigma=1.5):
    grid = np.arange(-size // 2 + 1, size // 2 + 1, dtype=np.float64)
    g = np.exp(-0.5 * (grid / sigma) ** 2) / (sigma * math.sqrt(2 * math.pi))
    return g / g.sum()


def _filt(v):
    g = gaussian_taps()
    vp = np.pad(v, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    N, D, H, W, _ = v.shape
    out = np.zeros_like(v)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                out += g[a] * g[b] * g[c] * vp[:, a:a + D, b:b + H, c:c + W, :]
    return out


def ssim_loss_3d(t, p, max_val=1.0, k1=0.01, k2=0.03):
    mt, mp = _filt(t), _filt(p)
    stt, spp, stp = _filt(t * t) - mt * mt, _filt(p * p) - mp * mp, _filt(t * p) - mt * mp
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    return 1.0 - (2 * mt * mp + c1) * (2 * stp + c2) / ((mt * mt + mp * mp + c1) * (stt + spp + c2))


def _pool(x, offsets, is_min):
    """x: [B,D,H,W]; min/max over offsets, out-of-range ignored ('same' padding of MaxPool3D)."""
    fill = np.inf if is_min else -np.inf
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1)), constant_values=fill)
    B, D, H, W = x.shape
    out = np.full_like(x, fill)
    for a, b, c in offsets:
        v = xp[:, 1 + a:1 + a + D, 1 + b:1 + b + H, 1 + c:1 + c + W]
        out = np.minimum(out, v) if is_min else np.maximum(out, v)
    return out


def erode_offsets():
    """Union of the (3,3,1), (3,1,3), (1,3,3) windows of clDice_func.py:23-26: 19 voxels."""
    s = set()
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            s.add((a, b, 0)); s.add((a, 0, b)); s.add((0, a, b))
    return sorted(s)


def soft_erode(x):
    return _pool(x, erode_offsets(), True)


def soft_dilate(x):
    return _pool(x, [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], False)


def soft_skel(img, iters):
    img1 = soft_dilate(soft_erode(img))
    skel = relu(img - img1)
    for _ in range(iters):
        img = soft_erode(img)
        img1 = soft_dilate(soft_erode(img))
        delta = relu(img - img1)
        skel = skel + relu(delta - skel * delta)
    return skel


def soft_dice_cldice(t, p, iters=15, alpha=0.5):
    t, p = t[..., 0], p[..., 0]
    sp, st = soft_skel(p, iters), soft_skel(t, iters)
    pres = ((sp * t).sum() + 1.0) / (sp.sum() + 1.0)
    rec = ((st * p).sum() + 1.0) / (st.sum() + 1.0)
    cl = 1.0 - 2.0 * pres * rec / (pres + rec)
    dice = 1.0 - (2.0 * (t * p).sum() + 1.0) / (t.sum() + p.sum() + 1.0)
    return (1 - alpha) * dice + alpha * cl


def reduce_mean(x, gbs, axis=None):
    m = x.mean() if axis is None else x.mean(axis=axis)
    return np.sum(m) / gbs


def mse(a, b, gbs):
    return reduce_mean((a - b) ** 2, gbs, axis=tuple(range(1, a.ndim)))


def compute_losses(P, real_I, real_S, gbs, n_devices=1, lc=10.0, lr=5.0, lt=5.0, iters=15):
    """vangan.py:270-353 (training=False: no noise/dropout)."""
    fake_S = resunet_forward(P['gen_IS'], real_I)
    fake_I = resunet_forward(P['gen_SI'], real_S)
    cyc_S = resunet_forward(P['gen_IS'], fake_I)
    cyc_I = resunet_forward(P['gen_SI'], fake_S)
    rS, cS = min_max_norm(real_S), min_max_norm(cyc_S)
    cycle_I = reduce_mean(keras_bce(rS, cS), gbs) * lc
    seg = soft_dice_cldice(rS, cS, iters) * (lt / n_devices)
    cycle_S = mse(real_I, cyc_I, gbs) * lc
    rec = reduce_mean(ssim_loss_3d(min_max_norm(real_I), min_max_norm(cyc_I)), gbs) * lr
    dRS, dFS = disc_forward(P['disc_S'], real_S), disc_forward(P['disc_S'], fake_S)
    dRI, dFI = disc_forward(P['disc_I'], real_I), disc_forward(P['disc_I'], fake_I)
    gIS, gSI = mse(np.ones_like(dFS), dFS, gbs), mse(np.ones_like(dFI), dFI, gbs)
    dI = 0.5 * (mse(np.ones_like(dRI), dRI, gbs) + mse(np.zeros_like(dFI), dFI, gbs))
    dS = 0.5 * (mse(np.ones_like(dRS), dRS, gbs) + mse(np.zeros_like(dFS), dFS, gbs))
    keys = ['total_IS_loss', 'total_SI_loss', 'D_I_loss', 'D_S_loss', 'gen_IS_loss', 'gen_SI_loss',
            'cycle_gen_SIS_loss', 'cycle_gen_ISI_loss', 'seg_loss', 'reconstruction_loss_I']
    vals = [gIS + cycle_I + seg, gSI + cycle_S + rec, dI, dS, gIS, gSI, cycle_I, cycle_S, seg, rec]
    return dict(zip(keys, [float(v) for v in vals]))


def adam_first_step(w, g, lr=2e-4, b1=0.5, b2=0.9, eps=1e-7, clipnorm=100.0):
    """Closed form of the first Keras Adam step with per-variable clipnorm."""
    n = math.sqrt(float((g ** 2).sum()))
    if n > clipnorm:
        g = g * (clipnorm / n)
    m, v = (1 - b1) * g, (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2) / (1 - b1)
    return w - lr_t * m / (np.sqrt(v) + eps)


# ------------------------------------------------------------------------------------------------------
# Hand-derived BACKWARD of one residual block (resunet_model.py:103-143), closed forms in numpy: the second,
# independent implementation of what torch autograd computes in oracle/vangan_oracle.py::_res_block.
# ------------------------------------------------------------------------------------------------------
def conv3d_backward(x, w, gy, stride=1, padding='valid'):
    """-> (dx, dw, db) of conv3d above: explicit tap loop (dx scatters gy @ w^T back to the strided input slice)."""
    kd, kh, kw, ci, co = w.shape
    pads = [(0, 0)] * 3
    if padding == 'same':
        pads = [same_pads(x.shape[1 + a], (kd, kh, kw)[a], stride) for a in range(3)]
    xp = np.pad(x, ((0, 0), pads[0], pads[1], pads[2], (0, 0)))
    N, od, oh, ow, _ = gy.shape
    dxp = np.zeros_like(xp, dtype=np.float64)
    dw = np.zeros_like(w, dtype=np.float64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                sl = (slice(None), slice(a, a + (od - 1) * stride + 1, stride), slice(b, b + (oh - 1) * stride + 1, stride),
                      slice(c, c + (ow - 1) * stride + 1, stride), slice(None))
                dw[a, b, c] = np.einsum('ndhwi,ndhwo->io', xp[sl], gy)
                dxp[sl] += gy @ w[a, b, c].T
    D, H, W = x.shape[1:4]
    dx = dxp[:, pads[0][0]:pads[0][0] + D, pads[1][0]:pads[1][0] + H, pads[2][0]:pads[2][0] + W, :]
    return dx, dw, gy.sum(axis=(0, 1, 2, 3))


def reflect_pad1_backward(gp):
    """Transpose of reflect_pad1: every padded position adds its gradient to the interior voxel it mirrors."""
    g = gp.copy()
    for ax in (1, 2, 3):
        n = g.shape[ax] - 2
        idx = np.arange(-1, n + 1)
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
        out = np.zeros(g.shape[:ax] + (n,) + g.shape[ax + 1:], dtype=np.float64)
        np.add.at(out, (slice(None),) * ax + (idx,), g)
        g = out
    return g


def instance_norm_backward(x, gamma, gy):
    """-> (dx, dgamma, dbeta) of instance_norm above (biased variance, eps inside the square root):
    dx = gamma*rstd * (gy - mean(gy) - xhat*mean(gy*xhat))."""
    mu = x.mean(axis=(1, 2, 3), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(1, 2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(var + IN_EPS)
    xh = (x - mu) * rstd
    dgamma, dbeta = (gy * xh).sum(axis=(0, 1, 2, 3)), gy.sum(axis=(0, 1, 2, 3))
    m1 = gy.mean(axis=(1, 2, 3), keepdims=True)
    m2 = (gy * xh).mean(axis=(1, 2, 3), keepdims=True)
    return gamma * rstd * (gy - m1 - xh * m2), dgamma, dbeta


def _cb_backward(p, name, x, stride, gy):
    n = instance_norm(x, p[name + '.in.gamma'], p[name + '.in.beta'])
    a = relu(n)
    gap, dw, db = conv3d_backward(reflect_pad1(a), p[name + '.conv.w'], gy, stride, 'valid')
    gn = reflect_pad1_backward(gap) * (n > 0)
    dx, dg, dbt = instance_norm_backward(x, p[name + '.in.gamma'], gn)
    return dx, {name + '.conv.w': dw, name + '.conv.b': db, name + '.in.gamma': dg, name + '.in.beta': dbt}


def res_block_backward(p, name, x, stride, gout):
    """d(sum(out * gout)) / d(x, parameters) for out = _res(p, name, x, stride)."""
    r = _cb(p, name + '.cb1', x, stride)
    sc = conv3d(x, p[name + '.short.w'], p[name + '.short.b'], stride, 'same')
    grads = {}
    gr, g2 = _cb_backward(p, name + '.cb2', r, 1, gout)
    grads.update(g2)
    gsc, dg, dbt = instance_norm_backward(sc, p[name + '.short.in.gamma'], gout)
    grads[name + '.short.in.gamma'], grads[name + '.short.in.beta'] = dg, dbt
    dx_s, dw, db = conv3d_backward(x, p[name + '.short.w'], gsc, stride, 'same')
    grads[name + '.short.w'], grads[name + '.short.b'] = dw, db
    dx_1, g1 = _cb_backward(p, name + '.cb1', x, stride, gr)
    grads.update(g1)
    return dx_1 + dx_s, grads


# ------------------------------------------------------------------------------------------------------
# Hand-derived BACKWARD of the whole generator and discriminator (resunet_model.py:185-249, discriminator.py:7-124) from the closed
# forms above: d(sum(net(x) * g_out)) / d(parameters, x).  Second author of what torch autograd computes through
# oracle/vangan_oracle.py::resunet_forward / disc_forward (tests/golden/make_golden_np.py checks the two against each other and files
# the results as fixtures).
# ------------------------------------------------------------------------------------------------------
def resunet_backward(p, x, gy):
    """-> (dx, {parameter name: gradient}) for y = resunet_forward(p, x), upstream gradient gy = dL/dy (NDHWC)."""
    w1, b1 = p['stem.conv1.w'], p['stem.conv1.b']
    c1 = conv3d(reflect_pad1(x), w1, b1, 1, 'valid')
    cbo = _cb(p, 'stem.cb', c1, 1)
    sc = conv3d(x, p['stem.short.w'], p['stem.short.b'], 1, 'same')
    h = cbo + instance_norm(sc, p['stem.short.in.gamma'], p['stem.short.in.beta'])
    skips, enc_in = [h], {}
    for e in range(1, 5):
        enc_in[e] = h
        h = _res(p, 'enc%d' % e, h, 2)
        skips.append(h)
    br_in = h
    b1o = _cb(p, 'bridge.cb1', br_in, 1)
    h = _cb(p, 'bridge.cb2', b1o, 1)
    dec_in = {}
    for d in (3, 2, 1, 0):
        up = h.repeat(2, axis=1).repeat(2, axis=2).repeat(2, axis=3)
        dec_in[d] = np.concatenate([up, skips[d]], axis=-1)
        h = _res(p, 'dec%d' % d, dec_in[d], 1)
    y = np.tanh(conv3d(h, p['out.w'], p['out.b'], 1, 'same'))
    grads = {}
    g = gy * (1.0 - y * y)
    dh, grads['out.w'], grads['out.b'] = conv3d_backward(h, p['out.w'], g, 1, 'same')
    gskip = [0.0] * 5
    for d in (0, 1, 2, 3):
        dcat, gr = res_block_backward(p, 'dec%d' % d, dec_in[d], 1, dh)
        grads.update(gr)
        cu = dcat.shape[-1] - skips[d].shape[-1]
        gskip[d] = gskip[d] + dcat[..., cu:]
        du = dcat[..., :cu]
        N, D, H, W, C = du.shape
        dh = du.reshape(N, D // 2, 2, H // 2, 2, W // 2, 2, C).sum(axis=(2, 4, 6))       # transpose of UpSampling3D(2)
    dh, g2 = _cb_backward(p, 'bridge.cb2', b1o, 1, dh)
    grads.update(g2)
    dh, g1 = _cb_backward(p, 'bridge.cb1', br_in, 1, dh)
    grads.update(g1)
    for e in (4, 3, 2, 1):
        gout = dh + (gskip[e] if e < 4 else 0.0)          # skips[e] feeds dec(e) for e = 1..3; enc4's output only the bridge
        dh, gr = res_block_backward(p, 'enc%d' % e, enc_in[e], 2, gout)
        grads.update(gr)
    g0 = dh + gskip[0]
    gsc, grads['stem.short.in.gamma'], grads['stem.short.in.beta'] = instance_norm_backward(sc, p['stem.short.in.gamma'], g0)
    dx_s, grads['stem.short.w'], grads['stem.short.b'] = conv3d_backward(x, p['stem.short.w'], gsc, 1, 'same')
    dc1, gcb = _cb_backward(p, 'stem.cb', c1, 1, g0)
    grads.update(gcb)
    dxp, grads['stem.conv1.w'], grads['stem.conv1.b'] = conv3d_backward(reflect_pad1(x), w1, dc1, 1, 'valid')
    return reflect_pad1_backward(dxp) + dx_s, grads


def disc_backward(p, x, glogits):
    """-> (dx, {parameter name: gradient}) for logits = disc_forward(p, x) (no noise, no dropout: training=False wiring)."""
    layers = []                                   # (name, conv input as fed to conv3d, pre-norm conv output, reflect padded?, stride, padding)
    xp = reflect_pad1(x)
    c = conv3d(xp, p['conv0.w'], p['conv0.b'], 2, 'valid')
    layers.append(('conv0', xp, c, True, 2, 'valid'))
    h = lrelu(instance_norm(c, p['conv0.in.gamma'], p['conv0.in.beta']))
    for i in range(3):
        k = 'down%d' % i
        if i < 2:
            xp = reflect_pad1(h)
            c = conv3d(xp, p[k + '.w'], None, 2, 'valid')
            layers.append((k, xp, c, True, 2, 'valid'))
        else:
            c = conv3d(h, p[k + '.w'], None, 1, 'same')
            layers.append((k, h, c, False, 1, 'same'))
        h = lrelu(instance_norm(c, p[k + '.in.gamma'], p[k + '.in.beta']))
    grads = {}
    dh, grads['out.w'], grads['out.b'] = conv3d_backward(h, p['out.w'], glogits, 1, 'same')
    for k, xin, c, padded, stride, padding in reversed(layers):
        n = instance_norm(c, p[k + '.in.gamma'], p[k + '.in.beta'])
        gn = dh * np.where(n > 0, 1.0, 0.2)                     # TP: LeakyReLU gradient uses x > 0
        dc, grads[k + '.in.gamma'], grads[k + '.in.beta'] = instance_norm_backward(c, p[k + '.in.gamma'], gn)
        dxin, grads[k + '.w'], db = conv3d_backward(xin, p[k + '.w'], dc, stride, padding)
        if k == 'conv0':
            grads['conv0.b'] = db
        dh = reflect_pad1_backward(dxin) if padded else dxin
    return dh, grads
