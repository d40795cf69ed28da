"""GPU parity tests of the individual libvangan_hip.so entry points against the CPU oracle.

Tolerances: the HIP kernels take bf16 operands (inputs and weights rounded to bf16 exactly as the oracle's
``q`` hook does) and accumulate in fp32, so against a float64 oracle on the SAME rounded operands the only
differences are fp32 accumulation order (~1e-6 relative) plus one bf16 rounding of the stored output
(2^-9 relative).  Checks therefore use  |hip - ref| <= 1.2e-2*|ref| + 2e-3*max|ref|  for bf16 outputs and
1e-4 relative (L2) for fp32 outputs/gradients unless stated otherwise.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vangan_oracle as O  # noqa: E402


def _dev():
    return torch.device('cuda:0')


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def close_bf16(got, ref, name=''):
    got, ref = got.double().cpu(), ref.double().cpu()
    tol = 1.2e-2 * ref.abs() + 2e-3 * ref.abs().max() + 1e-30
    bad = (got - ref).abs() > tol
    assert not bad.any(), '%s: %d/%d outside tolerance, max err %.3e (max ref %.3e)' % (
        name, int(bad.sum()), bad.numel(), float((got - ref).abs().max()), float(ref.abs().max()))


def rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def make_layer(k, cin, cout, stride, pad, dims, bias=True, seed=0):
    from van_gan_amd.nets import ParamStore
    from van_gan_amd.ops import ConvLayer
    specs = [('c.w', (k, k, k, cin, cout), 'x')] + ([('c.b', (cout,), 'x')] if bias else [])
    st = ParamStore(specs, _dev())
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(k, k, k, cin, cout, generator=g) / math.sqrt(k ** 3 * cin)
    st.param('c.w').copy_(w)
    if bias:
        st.param('c.b').copy_(torch.randn(cout, generator=g) * 0.1)
    lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, bias, dims)
    lay.pack()
    return st, lay


def ref_conv(x_ncdhw, w_dhwio, b, stride, pad):
    """float64 reference on already-rounded operands."""
    xx = O.reflect_pad1(x_ncdhw) if pad == 'reflect' else x_ncdhw
    return O.conv3d(xx, w_dhwio, b, stride, 'valid' if pad == 'reflect' else 'same')


CONV_CASES = [
    # k, cin, cout, stride, pad, dims
    (3, 16, 16, 1, 'reflect', (8, 8, 16)),
    (3, 32, 32, 1, 'reflect', (4, 8, 8)),
    (3, 16, 32, 2, 'reflect', (8, 8, 16)),
    (1, 16, 32, 2, 'same', (8, 8, 16)),
    (1, 48, 16, 1, 'same', (4, 4, 16)),
    (3, 64, 64, 1, 'reflect', (4, 4, 4)),
    (3, 128, 128, 1, 'reflect', (2, 2, 2)),
    (3, 1, 16, 1, 'reflect', (8, 8, 16)),
    (1, 16, 1, 1, 'same', (8, 8, 16)),
    (4, 1, 64, 2, 'reflect', (8, 8, 16)),
    (4, 64, 128, 2, 'reflect', (4, 8, 8)),
    (4, 32, 64, 1, 'same', (4, 4, 4)),
    (3, 64, 1, 1, 'same', (4, 4, 4)),
    (3, 96, 32, 1, 'reflect', (4, 4, 8)),
    # wide layers big enough for the 32x32x16-MFMA flavour (conv32_kernel): 128-wide panel forward, 64-wide data gradient
    (3, 64, 128, 1, 'reflect', (16, 16, 32)),
    (4, 64, 128, 2, 'reflect', (32, 32, 32)),
    # strided layers whose dY needs several channel chunks: class-parallel data gradient (one launch, workgroup -> (class,
    # tile)); classes of 1/2/4/8 taps (k3), ragged class extents (odd dims), the 32x32x16 flavour (D.down1-like)
    (3, 64, 128, 2, 'reflect', (8, 8, 16)),
    (3, 128, 256, 2, 'reflect', (4, 8, 8)),
    (3, 16, 128, 2, 'reflect', (5, 7, 9)),
    (4, 128, 256, 2, 'reflect', (16, 16, 16)),
    # single-channel sources run W-packed (k taps along W -> k pseudo-channels); besides the stem / D.conv0 shapes above:
    # zero padding, ragged extents, stride 2 with 'same' padding
    (3, 1, 16, 1, 'same', (6, 10, 12)),
    (4, 1, 32, 2, 'same', (7, 9, 11)),
    # 1x1x1 with one channel on one side: the HBM-bound VALU kernels of vg_pointwise.hip (with (1, 16, 1) above: forward
    # C->1 / 1->C with statistics, both weight gradients, both data gradients)
    (1, 1, 16, 1, 'same', (8, 8, 16)),
    (1, 32, 1, 1, 'same', (5, 6, 7)),
    # 3x3x3 single-channel stem convolution on its VALU kernels: ragged W (quads of 4 voxels with a masked tail)
    (3, 1, 16, 1, 'reflect', (5, 6, 10)),
]


@pytest.mark.parametrize('k,cin,cout,stride,pad,dims', CONV_CASES)
def test_conv_forward_dgrad_wgrad(k, cin, cout, stride, pad, dims):
    """Conv3D forward (+ on-read IN/ReLU, bias, statistics), data gradient and weight gradient vs autograd."""
    from van_gan_amd import ops
    from van_gan_amd.ops import Src
    dev = _dev()
    N = 2
    st, lay = make_layer(k, cin, cout, stride, pad, dims)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, *dims, cin, generator=g)
    scale = torch.rand(N, cin, generator=g) + 0.5
    shift = torch.randn(N, cin, generator=g) * 0.3
    use_norm = cin > 1
    f32_src = cin == 1
    xs = x if f32_src else x.to(torch.bfloat16)
    xd = xs.to(dev)
    src = Src(xd, (N,) + dims, cin, f32=f32_src, scale=scale.to(dev) if use_norm else None,
              shift=shift.to(dev) if use_norm else None, act=ops.ACT_RELU if use_norm else ops.ACT_NONE)
    out_f32 = cout == 1
    out = torch.zeros(N, *lay.out_dims, cout, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    sums = torch.zeros(8, N, cout, 2, device=dev)
    lay.forward(src, out, sums=sums)
    torch.cuda.synchronize()
    sums = sums.sum(0)
    # reference
    xr = xs.double()
    a = xr
    if use_norm:
        a = F.relu(xr * scale.double().view(N, 1, 1, 1, cin) + shift.double().view(N, 1, 1, 1, cin))
    a = bf(a).requires_grad_(True)
    w = bf(st.param('c.w').cpu()).requires_grad_(True)
    b = st.param('c.b').cpu().double().requires_grad_(True)
    y = ref_conv(O.to_ncdhw(a), w, b, stride, pad)
    y_ndhwc = O.to_ndhwc(y)
    if out_f32:
        assert rel_l2(out, y_ndhwc) < 1e-4
    else:
        close_bf16(out, y_ndhwc, 'forward')
    ref_sums = torch.stack([y_ndhwc.sum(dim=(1, 2, 3)), (y_ndhwc ** 2).sum(dim=(1, 2, 3))], dim=-1)
    assert rel_l2(sums, ref_sums) < 2e-2
    # backward
    dy = torch.randn(y_ndhwc.shape, generator=g)
    dys = dy if out_f32 else dy.to(torch.bfloat16)
    (y_ndhwc * bf(dys)).sum().backward()        # the kernels round dY to bf16 when staging it
    dyd = dys.to(dev)
    lay.wgrad(src, dyd)
    torch.cuda.synchronize()
    assert rel_l2(st.grad('c.w'), w.grad) < 2e-3, 'wgrad'
    assert rel_l2(st.grad('c.b'), b.grad) < 2e-3, 'bgrad'
    # data gradient: padded grid for reflect (fold on host for the check), plain for 'same'
    dp = torch.zeros(N, *lay.buf_dims, cin, dtype=torch.bfloat16, device=dev)
    lay.dgrad(dyd, N, dp, accumulate=False)
    dxg = torch.zeros(N, *dims, cin, dtype=torch.float32 if cin == 1 else torch.bfloat16, device=dev)
    ops.actnorm_bwd(dp, pad == 'reflect', None, (N,) + dims, cin, dxg, act=ops.ACT_NONE, norm=False, accumulate=False)
    torch.cuda.synchronize()
    ref_dx = a.grad
    got = dxg.double().cpu()
    tol = 2.5e-2 * ref_dx.abs() + 6e-3 * ref_dx.abs().max()
    assert ((got - ref_dx).abs() <= tol).all(), 'dgrad max err %.3e' % float((got - ref_dx).abs().max())


def test_conv_virtual_concat_residual_noise_tanh():
    """Virtual upsample+concat input, residual*scale+shift epilogue, noise on the padded grid, tanh/f32 output."""
    from van_gan_amd import ops
    from van_gan_amd.ops import Src
    dev = _dev()
    N, dims, cu, cs, cout = 2, (4, 8, 8), 32, 16, 16
    st, lay = make_layer(3, cu + cs, cout, 1, 'reflect', dims)
    g = torch.Generator().manual_seed(2)
    low = torch.randn(N, 2, 4, 4, cu, generator=g).to(torch.bfloat16)
    skip = torch.randn(N, *dims, cs, generator=g).to(torch.bfloat16)
    scale = torch.rand(N, cu + cs, generator=g) + 0.5
    shift = torch.randn(N, cu + cs, generator=g) * 0.3
    noise = (torch.randn(N, 6, 10, 10, cu + cs, generator=g) * 0.1).to(torch.bfloat16)
    res = torch.randn(N, *dims, cout, generator=g).to(torch.bfloat16)
    rs, rb = torch.rand(N, cout, generator=g) + 0.5, torch.randn(N, cout, generator=g)
    src = Src(low.to(dev), (N,) + dims, cu, skip.to(dev), cs, shift0=1, scale=scale.to(dev), shift=shift.to(dev),
              act=ops.ACT_LRELU, noise=noise.to(dev), noise_pad=1)
    out = torch.zeros(N, *dims, cout, dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(8, N, cout, 2, device=dev)
    lay.forward(src, out, sums=sums, res=res.to(dev), res_scale=rs.to(dev), res_shift=rb.to(dev))
    torch.cuda.synchronize()
    up = low.double().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    cat = torch.cat([up, skip.double()], dim=-1)
    a = F.leaky_relu(cat * scale.double().view(N, 1, 1, 1, -1) + shift.double().view(N, 1, 1, 1, -1), 0.2)
    ap = O.to_ndhwc(O.reflect_pad1(O.to_ncdhw(a))) + noise.double()
    y = O.conv3d(O.to_ncdhw(bf(ap)), bf(st.param('c.w').cpu()), st.param('c.b').cpu().double(), 1, 'valid')
    y = O.to_ndhwc(y) + res.double() * rs.double().view(N, 1, 1, 1, -1) + rb.double().view(N, 1, 1, 1, -1)
    close_bf16(out, y, 'concat/res/noise forward')
    # weight gradient through the same virtual operand
    dy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
    lay.wgrad(src, dy.to(dev))
    torch.cuda.synchronize()
    apq = bf(ap).requires_grad_(False)
    w = bf(st.param('c.w').cpu()).requires_grad_(True)
    yy = O.to_ndhwc(O.conv3d(O.to_ncdhw(apq), w, None, 1, 'valid'))
    (yy * dy.double()).sum().backward()
    assert rel_l2(st.grad('c.w'), w.grad) < 2e-3
    # tanh + f32 output on a 16->1 conv
    st2, lay2 = make_layer(1, 16, 1, 1, 'same', dims, seed=3)
    x = torch.randn(N, *dims, 16, generator=g).to(torch.bfloat16)
    o2 = torch.zeros(N, *dims, 1, device=dev)
    lay2.forward(Src(x.to(dev), (N,) + dims, 16), o2, tanh=True)
    torch.cuda.synchronize()
    y2 = torch.tanh(O.to_ndhwc(O.conv3d(O.to_ncdhw(x.double()), bf(st2.param('c.w').cpu()), st2.param('c.b').cpu().double(), 1, 'same')))
    assert rel_l2(o2, y2) < 1e-4


@pytest.mark.parametrize('cin,filters,dims,N', [(32, 16, (8, 8, 8), 2), (128, 64, (4, 6, 8), 1), (256, 128, (4, 4, 4), 2), (48, 16, (16, 16, 16), 1)])
def test_conv3d_transpose_k2s2(cin, filters, dims, N):
    """the k2 s2 Conv3DTranspose of the 'deconv' decoder (resunet_model.py:168-174, vnet_model.py:244-245) as a recipe over
    the strided data gradient (ops.ConvTranspose3dK2S2): forward + bias, data gradient, kernel and bias gradients against
    F.conv_transpose3d / autograd through the oracle restatement, on identical bf16-rounded operands."""
    from van_gan_amd import ops
    from van_gan_amd.nets import ParamStore
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    st = ParamStore([('up.w', (2, 2, 2, filters, cin), 'x'), ('up.b', (filters,), 'x')], dev)
    w = torch.randn(2, 2, 2, filters, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(filters, generator=g) * 0.1
    st.load({'up.w': w, 'up.b': b})
    lay = ops.ConvTranspose3dK2S2(st, 'up', cin, filters, dims)
    lay.pack()
    x = torch.randn(N, *dims, cin, generator=g).to(torch.bfloat16)
    odims = tuple(2 * n for n in dims)
    y = torch.full((N,) + odims + (filters,), 7.0, dtype=torch.bfloat16, device=dev)
    lay.forward(x.to(dev), N, y)
    dy = torch.randn(N, *odims, filters, generator=g).to(torch.bfloat16)
    dx = torch.full((N,) + dims + (cin,), 7.0, dtype=torch.bfloat16, device=dev)
    st.g.zero_()
    lay.backward(x.to(dev), dy.to(dev), N, dx)
    ops.side_join()
    torch.cuda.synchronize()
    xr = O.to_ncdhw(x.double()).requires_grad_(True)
    wr = bf(w.double()).requires_grad_(True)
    br = b.double().requires_grad_(True)
    ref = O.conv3d_transpose_k2s2(xr, wr, br)
    close_bf16(y, O.to_ndhwc(ref.detach()), 'Conv3DTranspose forward')
    (ref * O.to_ncdhw(dy.double())).sum().backward()
    close_bf16(dx, O.to_ndhwc(xr.grad), 'Conv3DTranspose data gradient')
    assert rel_l2(st.grad('up.w'), wr.grad) < 2e-3
    assert rel_l2(st.grad('up.b'), br.grad) < 2e-3


def test_in_finalize_and_actnorm_bwd():
    """InstanceNorm scale/shift + (IN -> ReLU) backward with the reflect-pad transpose, vs autograd."""
    from van_gan_amd import ops
    dev = _dev()
    N, dims, Cc = 2, (4, 6, 8), 16
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(N, *dims, Cc, generator=g) * 2 + 0.5).to(torch.bfloat16)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2
    xd = x.double()
    sums = torch.zeros(8, N, Cc, 2)
    sums[3] = torch.stack([xd.sum(dim=(1, 2, 3)), (xd ** 2).sum(dim=(1, 2, 3))], dim=-1).float() * 0.25
    sums[5] = sums[3] * 3.0          # striped partial sums
    sums = sums.to(dev)
    scale, shift, mean, rstd = [torch.zeros(N, Cc, device=dev) for _ in range(4)]
    S = dims[0] * dims[1] * dims[2]
    ops.in_finalize(sums, Cc, S, gamma.to(dev), beta.to(dev), N, scale, shift, mean, rstd)
    xr = xd.clone().requires_grad_(True)
    gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    nrm = O.to_ndhwc(O.instance_norm(O.to_ncdhw(xr), gm, bt))
    torch.cuda.synchronize()
    got_n = xd * scale.cpu().double().view(N, 1, 1, 1, Cc) + shift.cpu().double().view(N, 1, 1, 1, Cc)
    assert rel_l2(got_n, nrm.detach()) < 1e-4
    a = F.relu(nrm)
    ap = O.to_ndhwc(O.reflect_pad1(O.to_ncdhw(a)))
    gp = torch.randn(ap.shape, generator=g).to(torch.bfloat16)
    (ap * gp.double()).sum().backward()
    red = torch.zeros(8, N, Cc, 2, device=dev)
    dx = torch.zeros(N, *dims, Cc, dtype=torch.bfloat16, device=dev)
    dgam, dbet = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ops.actnorm_bwd(gp.to(dev), True, x.to(dev), (N,) + dims, Cc, dx, scale=scale, shift=shift, act=ops.ACT_RELU,
                    norm=True, gamma=gamma.to(dev), mean=mean, rstd=rstd, red=red, accumulate=False, dgamma=dgam, dbeta=dbet)
    torch.cuda.synchronize()
    assert rel_l2(dx, xr.grad) < 1e-2
    assert rel_l2(dgam, gm.grad) < 1e-3 and rel_l2(dbet, bt.grad) < 1e-3


def test_concat_bwd_and_tanh_bwd():
    from van_gan_amd import ops
    dev = _dev()
    N, D, H, W, Cu, Cs = 2, 4, 4, 8, 16, 8
    g = torch.Generator().manual_seed(5)
    gcat = torch.randn(N, D, H, W, Cu + Cs, generator=g).to(torch.bfloat16)
    dlow = torch.randn(N, D // 2, H // 2, W // 2, Cu, generator=g).to(torch.bfloat16)
    dskip = torch.randn(N, D, H, W, Cs, generator=g).to(torch.bfloat16)
    dl, ds = dlow.to(dev).clone(), dskip.to(dev).clone()
    ops.concat_bwd(gcat.to(dev), (N, D, H, W), Cu, Cs, dl, ds)
    torch.cuda.synchronize()
    gu = gcat[..., :Cu].double().view(N, D // 2, 2, H // 2, 2, W // 2, 2, Cu).sum(dim=(2, 4, 6))
    close_bf16(dl, dlow.double() + gu, 'dlow')
    close_bf16(ds, dskip.double() + gcat[..., Cu:].double(), 'dskip')
    y = torch.tanh(torch.randn(1000, generator=g)); dy = torch.randn(1000, generator=g)
    dp = torch.zeros(1000, device=dev)
    ops.tanh_bwd(dy.to(dev), y.to(dev), dp)
    assert rel_l2(dp, dy * (1 - y * y)) < 1e-6


def test_losses_vs_oracle():
    """min-max (fwd/bwd incl. ties), BCE, MSE, LSGAN terms, SSIM fwd/bwd vs the oracle's autograd."""
    from van_gan_amd import ops
    dev = _dev()
    B, D, H, W = 2, 6, 8, 10
    S = D * H * W
    g = torch.Generator().manual_seed(6)
    x = torch.tanh(torch.randn(B, D, H, W, 1, generator=g) * 2)
    x[0, 0, 0, 0, 0] = x[0].max(); x[0, 1, 1, 1, 0] = x[0].max()           # tie at the max
    t = (torch.rand(B, D, H, W, 1, generator=g) > 0.7).float()
    xd, td = x.to(dev), t.to(dev)
    mm, y = torch.zeros(B, 4, device=dev), torch.zeros(B, D, H, W, 1, device=dev)
    ops.minmax(xd, B, S, mm); ops.minmax_apply(xd, mm, B, S, y)
    xr = x.double().requires_grad_(True)
    yr = O.min_max_norm(xr)
    assert rel_l2(y, yr.detach()) < 1e-6
    assert float(mm[0, 3]) == 3.0 and float(mm[1, 3]) == 1.0
    # BCE
    acc = torch.zeros(4, device=dev)
    gy = torch.zeros_like(y)
    ops.bce(td, y, acc[0:1], 0.37, gy)
    lb = O.keras_bce(t.double(), yr).sum()
    (0.37 * lb).backward()
    assert abs(float(acc[0]) - float(lb)) < 1e-3 * abs(float(lb))
    dx = torch.zeros_like(y); tmp2 = torch.zeros(B, 2, device=dev)
    ops.minmax_bwd(xd, y, gy, mm, B, S, tmp2, dx)
    torch.cuda.synchronize()
    assert rel_l2(dx, xr.grad) < 5e-3       # fp32 sums with heavy cancellation (1/(p+eps) BCE gradients) vs float64
    # MSE and LSGAN constants
    a, b = torch.randn(B, S, generator=g), torch.randn(B, S, generator=g)
    gb = torch.zeros(B, S, device=dev)
    ops.mse(a.to(dev), b.to(dev), acc[1:2], 0.5, gb)
    assert abs(float(acc[1]) - float(((a - b) ** 2).sum())) < 1e-3 * float(((a - b) ** 2).sum())
    assert rel_l2(gb, 0.5 * 2 * (b - a)) < 1e-5
    gx = torch.zeros(B, S, device=dev)
    ops.mse_const(b.to(dev).to(torch.bfloat16), 1.0, acc[2:3], 2.0, gx)
    bb = b.to(torch.bfloat16).float()
    assert abs(float(acc[2]) - float(((bb - 1) ** 2).sum())) < 1e-3 * float(((bb - 1) ** 2).sum())
    assert rel_l2(gx, 4 * (bb - 1)) < 1e-5
    # SSIM
    p = torch.rand(B, D, H, W, 1, generator=g); tt = torch.rand(B, D, H, W, 1, generator=g)
    pr = p.double().requires_grad_(True)
    ls = O.ssim_loss_3d(tt.double(), pr).sum()
    ls.backward()
    part = torch.zeros(3, B, D, H, W, 1, device=dev); gp = torch.zeros(B, D, H, W, 1, device=dev)
    ops.ssim_fwd(tt.to(dev), p.to(dev), (B, D, H, W), acc[3:4], part)
    ops.ssim_bwd(tt.to(dev), p.to(dev), part, (B, D, H, W), 1.0, gp)
    torch.cuda.synchronize()
    assert abs(float(acc[3]) - float(ls)) < 1e-3 * abs(float(ls))
    assert rel_l2(gp, pr.grad) < 1e-3


def test_soft_skeleton_chain_kernel_is_bitwise_the_per_step_kernels():
    """The single-launch skeleton chain (all steps of one soft_skel, running skeleton in registers) against the one-launch-per-step
    kernels it replaces: every intermediate skeleton bit for bit, on a grid with ragged tiles."""
    from van_gan_amd import ops
    from van_gan_amd._lib import lib
    dev = _dev()
    B, D, H, W, it = 2, 11, 13, 37, 6
    g = torch.Generator().manual_seed(17)
    p = torch.rand(B, D, H, W, 1, generator=g).to(dev)
    vol = (B, D, H, W, 1)
    out = []
    for chain in (1, 0):
        lib.vg_set_tuning(b'SKEL_CHAIN', chain, 0)
        imgs, skels = torch.zeros((it + 2,) + vol, device=dev), torch.full((it + 1,) + vol, 3.0, device=dev)
        ops.soft_skel_fwd(p, (B, D, H, W), it, imgs, skels)
        torch.cuda.synchronize()
        out.append((imgs.cpu(), skels.cpu()))
    lib.vg_set_tuning(b'SKEL_CHAIN', 0, 1)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_multi_erosion_launches_are_bitwise_the_per_step_kernel():
    """Round 5: four (two, one) erosions per launch on LDS tiles with a 4- (2-, 1-) voxel halo against one launch per erosion: the
    whole stored chain img_0 .. img_{iters+1} and every skeleton bit for bit, ragged tiles, 7 erosions = one launch of each kind."""
    from van_gan_amd import ops
    from van_gan_amd._lib import lib
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    for (B, D, H, W, it) in ((2, 11, 13, 37, 6), (1, 32, 16, 64, 15), (1, 5, 3, 9, 3)):
        p = torch.rand(B, D, H, W, 1, generator=g).to(dev)
        vol = (B, D, H, W, 1)
        out = []
        for multi in (4, 2, 1):
            lib.vg_set_tuning(b'SKEL_MULTI', multi, 0)
            imgs, skels = torch.full((it + 2,) + vol, -5.0, device=dev), torch.full((it + 1,) + vol, 3.0, device=dev)
            ops.soft_skel_fwd(p, (B, D, H, W), it, imgs, skels)
            torch.cuda.synchronize()
            out.append((imgs.cpu(), skels.cpu()))
        lib.vg_set_tuning(b'SKEL_MULTI', 0, 1)
        assert torch.equal(out[0][0][0], p.cpu())                          # img_0 filed by the first launch
        for o in out[:2]:
            assert torch.equal(o[0], out[2][0]) and torch.equal(o[1], out[2][1]), (B, D, H, W, it)


def test_skeleton_backward_from_filed_codes_matches_the_scan_kernels():
    """Round 5: with aux the forward pass files delta_j and the FIRST arg-max / arg-min code of every pooling window, and the backward
    pass is iters + 2 streaming launches that route by table lookup.  Against the scan kernels (which recompute the arg-extrema from the
    stored chain, two launches per step): same chain and skeletons bit for bit, every code equal to a brute-force first-candidate scan in
    the reference's order (continuous data AND binary data full of ties), gradients equal up to the order of the float atomics."""
    from van_gan_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(29)
    for (B, D, H, W, it, binary) in ((2, 11, 13, 37, 6, False), (1, 16, 24, 40, 15, False), (1, 12, 10, 33, 5, True), (1, 5, 3, 9, 3, False)):
        p = torch.rand(B, D, H, W, 1, generator=g)
        if binary:
            p = (F.avg_pool3d(p.permute(0, 4, 1, 2, 3), 3, 1, 1) > 0.5).float().permute(0, 2, 3, 4, 1).contiguous()
        p = p.to(dev)
        vol = (B, D, H, W, 1)
        dims = (B, D, H, W)
        n = B * D * H * W
        imgs, skels = torch.zeros((it + 2,) + vol, device=dev), torch.zeros((it + 1,) + vol, device=dev)
        ops.soft_skel_fwd(p, dims, it, imgs, skels)
        imgs2, skels2 = torch.zeros((it + 2,) + vol, device=dev), torch.zeros((it + 1,) + vol, device=dev)
        aux = torch.zeros(ops.skel_aux_bytes(dims, it), dtype=torch.uint8, device=dev)
        ops.soft_skel_fwd(p, dims, it, imgs2, skels2, aux)
        torch.cuda.synchronize()
        assert torch.equal(imgs, imgs2) and torch.equal(skels, skels2)
        # brute-force codes from the stored chain: first extremum in scan order (erode: the three 3x3 planes one after the other)
        ic = imgs2.cpu()[..., 0]
        codeM = aux[(it + 1) * n * 4:(it + 1) * n * 5].cpu().view(it + 1, B, D, H, W)
        codeN = aux[(it + 1) * n * 5:].cpu().view(it + 1, B, D, H, W)
        delta = aux[:(it + 1) * n * 4].view(torch.float32).cpu().view(it + 1, B, D, H, W)
        er_order = [(a, b, 0) for a in (-1, 0, 1) for b in (-1, 0, 1)] + [(a, 0, c) for a in (-1, 0, 1) for c in (-1, 0, 1)] + \
                   [(0, b, c) for b in (-1, 0, 1) for c in (-1, 0, 1)]
        di_order = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]

        def first_arg(x, order, sign):
            """x: [B,D,H,W]; code of the first candidate (in `order`) attaining the extremum; sign=+1: max, -1: min."""
            pad = F.pad(x * sign, (1, 1, 1, 1, 1, 1), value=float('-inf'))
            best = torch.full_like(x, float('-inf')); code = torch.zeros_like(x, dtype=torch.int64)
            for (a, b, c) in order:
                v = pad[:, 1 + a:1 + a + D, 1 + b:1 + b + H, 1 + c:1 + c + W]
                better = v > best
                best = torch.where(better, v, best)
                code = torch.where(better, torch.full_like(code, ((a + 1) * 3 + (b + 1)) * 3 + (c + 1)), code)
            return code, best * sign
        for j in range(it + 1):
            cn, mn = first_arg(ic[j], er_order, -1)
            assert torch.equal(mn, ic[j + 1]) and torch.equal(cn, codeN[j].long()), ('arg-min codes', j)
            cm, mx = first_arg(ic[j + 1], di_order, +1)
            dl = torch.relu(ic[j] - mx)
            assert torch.equal(dl, delta[j]), ('delta', j)
            assert torch.equal(cm, codeM[j].long()), ('arg-max codes', j)
        gskel = torch.randn(vol, generator=g).to(dev)
        res = []
        for use_aux in (True, False):
            gp = torch.full(vol, 0.25, device=dev)                      # accumulated into
            work = torch.full((4,) + vol, float('nan'), device=dev) if use_aux else torch.zeros((3,) + vol, device=dev)
            ops.soft_skel_bwd(imgs2, skels2, gskel, dims, it, work, gp, aux if use_aux else None)
            torch.cuda.synchronize()
            res.append(gp.cpu())
        assert bool(torch.isfinite(res[0]).all())
        err = float((res[0] - res[1]).abs().max()), float(res[1].abs().max())
        assert err[0] <= 2e-5 * max(err[1], 1.0), (B, D, H, W, it, binary, err)


def test_soft_skeleton_and_cldice():
    """soft_skel forward/backward and the Dice+clDice combination vs the oracle (continuous data: no ties)."""
    from van_gan_amd import ops
    dev = _dev()
    B, D, H, W, it = 2, 8, 10, 12, 4
    g = torch.Generator().manual_seed(7)
    p = torch.rand(B, D, H, W, 1, generator=g)
    t = (F.avg_pool3d(torch.rand(B, 1, D, H, W, generator=g), 3, 1, 1) > 0.52).float().permute(0, 2, 3, 4, 1).contiguous()
    vol = (B, D, H, W, 1)
    imgs_p, skels_p = torch.zeros((it + 2,) + vol, device=dev), torch.zeros((it + 1,) + vol, device=dev)
    imgs_t, skels_t = torch.zeros((it + 2,) + vol, device=dev), torch.zeros((it + 1,) + vol, device=dev)
    pd, td = p.to(dev), t.to(dev)
    aux = torch.zeros(ops.skel_aux_bytes((B, D, H, W), it), dtype=torch.uint8, device=dev)
    ops.soft_skel_fwd(pd, (B, D, H, W), it, imgs_p, skels_p, aux)
    ops.soft_skel_fwd(td, (B, D, H, W), it, imgs_t, skels_t)
    pr = p.double().requires_grad_(True)
    sk_p = O.soft_skel(pr[..., 0], it)
    sk_t = O.soft_skel(t.double()[..., 0], it)
    torch.cuda.synchronize()
    assert rel_l2(skels_p[it][..., 0], sk_p.detach()) < 1e-6
    assert rel_l2(skels_t[it][..., 0], sk_t) < 1e-6
    w = 5.0
    loss = O.soft_dice_cldice(t.double(), pr, it) * w
    loss.backward()
    sums, coef = torch.zeros(9, device=dev), torch.zeros(8, device=dev)
    ops.dot_sums(skels_p[it], td, sums[0:3]); ops.dot_sums(skels_t[it], pd, sums[3:6]); ops.dot_sums(td, pd, sums[6:9])
    ops.cldice_coef(sums, w, 0.5, coef)
    gskel, gp = torch.zeros(vol, device=dev), torch.zeros(vol, device=dev)
    ops.cldice_grads(td, skels_t[it], coef, gskel, gp)
    work = torch.full((4,) + vol, float('nan'), device=dev)          # scratch: needs no initialisation
    gp0 = gp.clone()
    ops.soft_skel_bwd(imgs_p, skels_p, gskel, (B, D, H, W), it, work, gp, aux)
    torch.cuda.synchronize()
    assert abs(float(coef[5]) - float(loss)) < 1e-4 * abs(float(loss))
    assert rel_l2(gp, pr.grad) < 1e-3
    # the scan path (no aux: arg-extrema recomputed from the stored chain) against the same oracle gradient
    work3 = torch.zeros((3,) + vol, device=dev)
    ops.soft_skel_bwd(imgs_p, skels_p, gskel, (B, D, H, W), it, work3, gp0)
    torch.cuda.synchronize()
    assert rel_l2(gp0, pr.grad) < 1e-3


def test_adam_clip():
    from van_gan_amd import ops
    from van_gan_amd.nets import ParamStore
    specs = [('a', (1000,), 'x'), ('b', (3, 3, 3, 16, 16), 'x'), ('c', (7,), 'x'), ('d', (33, 65), 'x'), ('e', (21001,), 'x'), ('f', (5,), 'x')]
    st = ParamStore(specs, _dev())
    g = torch.Generator().manual_seed(8)
    P = {n: torch.randn(sh, generator=g) for n, sh, _ in specs}
    G = {n: torch.randn(sh, generator=g) for n, sh, _ in specs}
    G['b'] = G['b'] * 50.0                  # norm >> 100: clipped
    G['e'] = G['e'] * 3.0                   # a clipped tensor spread over seven 4096-element blocks
    st.load(P)
    state = {}
    for step in range(1, 3):
        for n in P:
            st.grad(n).copy_(G[n])
        lr_t = 2e-4 * math.sqrt(1 - 0.9 ** step) / (1 - 0.5 ** step)
        ops.adam_clip(st.w, st.g, st.m, st.v, st.seg_off, st.T, st.norms, lr_t, 0.5, 0.9, 1e-7, 100.0)
        Pd = {n: v.double() for n, v in P.items()} if step == 1 else Pd
        O.adam_step(Pd, {n: v.double() for n, v in G.items()}, state)
    torch.cuda.synchronize()
    out = st.export()
    for n in P:
        assert (out[n].double() - Pd[n]).abs().max() < 1e-6, n      # fp32 ulp at |w|~4 is 4.8e-7
    assert abs(float(st.norms[1]) - float((G['b'].double() ** 2).sum())) < 1e-3 * float((G['b'].double() ** 2).sum())
    assert abs(float(st.norms[4]) - float((G['e'].double() ** 2).sum())) < 1e-3 * float((G['e'].double() ** 2).sum())
    # the clip factor is reproducible bit for bit (fixed summation order: replicas must not drift apart)
    w_first = st.w.clone()
    for rep in range(20):
        st.load(P); st.m.zero_(); st.v.zero_()
        for step in range(1, 3):
            for n in P:
                st.grad(n).copy_(G[n])
            lr_t = 2e-4 * math.sqrt(1 - 0.9 ** step) / (1 - 0.5 ** step)
            ops.adam_clip(st.w, st.g, st.m, st.v, st.seg_off, st.T, st.norms, lr_t, 0.5, 0.9, 1e-7, 100.0)
        assert torch.equal(st.w, w_first), rep


def test_rng_statistics():
    from van_gan_amd import ops
    dev = _dev()
    z = torch.zeros(1 << 20, dtype=torch.bfloat16, device=dev)
    ops.randn_bf16(z, 0.1, 1234, 0)
    zf = z.float()
    assert abs(float(zf.mean())) < 1e-3 and abs(float(zf.std()) - 0.1) < 2e-3
    m = torch.zeros(1 << 16, device=dev)
    ops.dropout_mask(m, 0.2, 99, 0)
    keep = float((m > 0).float().mean())
    assert abs(keep - 0.8) < 0.01 and abs(float(m.max()) - 1.25) < 1e-6


@pytest.mark.gpu
def test_conv_lds_dma_staging_subprocess():
    """The optional LDS-DMA (global_load_lds) staging path of conv_kernel is selected per process by VG_CONV_DMA=1:
    run the forward/data-gradient conv cases again in a child process with it enabled."""
    import subprocess, sys
    env = dict(os.environ, VG_CONV_DMA='1', VG_NO_REBUILD='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu',
                        '-k', 'test_conv_forward_dgrad_wgrad'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('net', ['gen', 'disc'])
def test_table_repack_equals_per_layer_pack(net):
    """vg_pack_weights_multi (one launch, blocks shared out by operand size) writes exactly what the per-operand
    vg_pack_weights writes, for every forward and data-gradient operand of a network."""
    from van_gan_amd.nets import ParamStore, PatchGAN, ResUNet, disc_param_specs, gen_param_specs
    dev = _dev()
    st = ParamStore(gen_param_specs() if net == 'gen' else disc_param_specs(), dev)
    st.w.copy_(torch.randn(st.w.shape, generator=torch.Generator().manual_seed(3)).to(dev))
    m = ResUNet(st, (32, 32, 32)) if net == 'gen' else PatchGAN(st, (32, 32, 32))
    m.pack()
    torch.cuda.synchronize()
    got = [[it[2].clone() for it in l.pack_items()] for l in m.L.values()]
    for l in m.L.values():
        for it in l.pack_items():
            it[2].fill_(7.0)
        l.pack()
    torch.cuda.synchronize()
    n = 0
    for l, g in zip(m.L.values(), got):
        for it, a in zip(l.pack_items(), g):
            assert torch.equal(it[2], a), l.name
            n += 1
    assert n >= (30 if net == 'gen' else 5)


@pytest.mark.parametrize('c_low,c_skip,cout,dims', [(32, 16, 16, (16, 16, 32)), (64, 32, 32, (8, 16, 16)), (64, 32, 64, (8, 8, 16)),
                                                    (32, 16, 16, (6, 10, 12))])
@pytest.mark.parametrize('acc_low,acc_skip', [(False, False), (True, True), (True, False)])
def test_shortcut_dgrad_concat_fused(c_low, c_skip, cout, dims, acc_low, acc_skip):
    """vg_shortcut_dgrad_concat (the decoder shortcut's data gradient + the backward of UpSampling3D + concatenate in one launch,
    resunet_model.py:126-131,175-181) against float64 on the same bf16 operands: dskip (+)= (dcat + dy W^T)[..., c_low:],
    dlow (+)= sum over every 2x2x2 block of (dcat + dy W^T)[..., :c_low]; and dcat itself must stay untouched."""
    from van_gan_amd import ops
    dev = _dev()
    N, cin = 2, c_low + c_skip
    st, lay = make_layer(1, cin, cout, 1, 'same', dims, bias=True, seed=3)
    g = torch.Generator().manual_seed(11)
    D, H, W = dims
    dy = torch.randn(N, D, H, W, cout, generator=g).to(torch.bfloat16)
    dcat = torch.randn(N, D, H, W, cin, generator=g).to(torch.bfloat16)
    dlow0 = torch.randn(N, D // 2, H // 2, W // 2, c_low, generator=g).to(torch.bfloat16)
    dskip0 = torch.randn(N, D, H, W, c_skip, generator=g).to(torch.bfloat16)
    dcat_d, dlow, dskip = dcat.to(dev), dlow0.to(dev).clone(), dskip0.to(dev).clone()
    assert ops.FUSE_CONCAT
    lay.dgrad_concat(dy.to(dev), N, dcat_d, c_low, dlow, dskip, acc_low=acc_low, acc_skip=acc_skip)
    torch.cuda.synchronize()
    assert torch.equal(dcat_d.cpu(), dcat), 'the concat gradient is read-only in the fused launch'
    w = bf(st.param('c.w').cpu()).view(cin, cout)
    full = dcat.double() + dy.double() @ w.t()
    low = full[..., :c_low].view(N, D // 2, 2, H // 2, 2, W // 2, 2, c_low).sum(dim=(2, 4, 6))
    skip = full[..., c_low:]
    if acc_low:
        low = low + dlow0.double()
    if acc_skip:
        skip = skip + dskip0.double()
    close_bf16(dlow, low, 'dlow')
    close_bf16(dskip, skip, 'dskip')
    # and the two-step path it replaces gives the same within the bf16 rounding of the intermediate
    os.environ['VG_PW_SPLIT'] = '0'
    from van_gan_amd import _lib
    _lib.lib.vg_set_tuning(b'PW_SPLIT', 0, 0)
    try:
        dcat2, dlow2, dskip2 = dcat.to(dev), dlow0.to(dev).clone(), dskip0.to(dev).clone()
        lay.dgrad_concat(dy.to(dev), N, dcat2, c_low, dlow2, dskip2, acc_low=acc_low, acc_skip=acc_skip)
        torch.cuda.synchronize()
    finally:
        _lib.lib.vg_set_tuning(b'PW_SPLIT', 0, 1)
        del os.environ['VG_PW_SPLIT']
    assert not torch.equal(dcat2.cpu(), dcat)            # the fallback did accumulate into the concat gradient
    assert rel_l2(dlow2, low) < 1e-2 and rel_l2(dskip2, skip) < 1e-2


@pytest.mark.parametrize('c_low,c_skip,cout,dims', [(32, 16, 16, (16, 16, 32)), (64, 32, 32, (8, 16, 16)), (32, 16, 16, (6, 10, 12)),
                                                    (64, 32, 64, (4, 4, 8))])
@pytest.mark.parametrize('acc', [False, True])
def test_shortcut_dgrad_concat_norm_fused(c_low, c_skip, cout, dims, acc):
    """vg_shortcut_dgrad_concat_norm: the decoder block's input gradient in ONE launch -- (InstanceNorm -> ReLU) backward of the first
    convolution's input from its reflection-padded data gradient (pad transposed on read), + the shortcut's data gradient, split and
    2x2x2 sum-pooled into the gradients of the two concat sources (resunet_model.py:103-143,175-181) -- against float64 autograd
    semantics written out by hand on the same bf16 operands."""
    from van_gan_amd import ops
    dev = _dev()
    N, C_ = 2, c_low + c_skip
    D, H, W = dims
    S = D * H * W
    st, lay = make_layer(1, C_, cout, 1, 'same', dims, bias=True, seed=5)
    g = torch.Generator().manual_seed(13)
    dy = torch.randn(N, D, H, W, cout, generator=g).to(torch.bfloat16)
    dp1 = torch.randn(N, D + 2, H + 2, W + 2, C_, generator=g).to(torch.bfloat16)
    low = torch.randn(N, D // 2, H // 2, W // 2, c_low, generator=g).to(torch.bfloat16)
    skip = torch.randn(N, D, H, W, c_skip, generator=g).to(torch.bfloat16)
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    mean, rstd = torch.randn(N, C_, generator=g) * 0.3, torch.rand(N, C_, generator=g) + 0.5
    scale = gamma.view(1, C_) * rstd
    shift = beta.view(1, C_) - mean * scale
    dlow0 = torch.randn(N, D // 2, H // 2, W // 2, c_low, generator=g).to(torch.bfloat16)
    dskip0 = torch.randn(N, D, H, W, c_skip, generator=g).to(torch.bfloat16)
    dgamma, dbeta = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    red = torch.zeros(ops.STRIPES * N * C_ * 2 + 4, device=dev)
    dlow, dskip = dlow0.to(dev).clone(), dskip0.to(dev).clone()
    nd = ops.actnorm_desc(dp1.to(dev), True, low.to(dev), (N, D, H, W), C_, None, scale=scale.to(dev), shift=shift.to(dev), act=ops.ACT_RELU,
                          norm=True, gamma=gamma.to(dev), mean=mean.to(dev), rstd=rstd.to(dev), red=red, accumulate=False,
                          x1=skip.to(dev), c_x0=c_low, x0_shift=1, dgamma=dgamma, dbeta=dbeta)
    ops.actnorm_stats(nd)
    served = lay.dgrad_concat_norm(dy.to(dev), N, nd, c_low, dlow, dskip, acc_low=acc, acc_skip=acc)
    torch.cuda.synchronize()
    assert served
    # reference
    x = torch.cat([low.double().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3), skip.double()], dim=-1)
    z = torch.zeros(N, C_, D, H, W, dtype=torch.float64, requires_grad=True)
    (O.reflect_pad1(z) * dp1.double().permute(0, 4, 1, 2, 3)).sum().backward()
    gf = z.grad.permute(0, 2, 3, 4, 1)                                    # transpose of the reflection pad
    v = lambda t: t.double().view(N, 1, 1, 1, C_)
    dn = gf * ((x * v(scale) + v(shift)) > 0)
    xh = (x - v(mean)) * v(rstd)
    r0, r1 = dn.sum(dim=(1, 2, 3), keepdim=True), (dn * xh).sum(dim=(1, 2, 3), keepdim=True)
    dx = gamma.double().view(1, 1, 1, 1, C_) * v(rstd) * (dn - r0 / S - xh * r1 / S)
    w = bf(st.param('c.w').cpu()).view(C_, cout)
    full = dx + dy.double() @ w.t()
    rlow = full[..., :c_low].view(N, D // 2, 2, H // 2, 2, W // 2, 2, c_low).sum(dim=(2, 4, 6))
    rskip = full[..., c_low:]
    if acc:
        rlow, rskip = rlow + dlow0.double(), rskip + dskip0.double()
    close_bf16(dlow, rlow, 'dlow')
    close_bf16(dskip, rskip, 'dskip')
    assert rel_l2(dbeta, r0.sum(0).flatten()) < 1e-4 and rel_l2(dgamma, r1.sum(0).flatten()) < 1e-4


@pytest.mark.parametrize('k,pad,dims,src_f32,transform,noise', [
    (3, 'reflect', (12, 24, 48), True, False, False),      # several tiles per axis, whole tiles
    (3, 'reflect', (9, 19, 37), False, True, False),       # ragged in every axis, bf16 source with the on-read affine + LeakyReLU
    (3, 'same', (6, 10, 12), True, True, False),           # zero padding
    (3, 'reflect', (2, 2, 2), True, False, False),         # the smallest grid reflection allows
    (4, 'reflect', (16, 32, 64), True, False, True),       # the discriminator's first layer: 4x4x4 stride 2 over reflect pad + noise
    (4, 'reflect', (10, 14, 44), True, False, True),       # ragged tiles
    (4, 'reflect', (8, 8, 16), False, True, False),        # bf16 source, on-read transform, no noise
])
def test_single_channel_convolutions_on_the_matrix_pipe(k, pad, dims, src_f32, transform, noise):
    """vg_c1k3.hip (1 -> 16, 3x3x3: the stem's first convolution, resunet_model.py:44-60; 1 -> 64, 4x4x4 stride 2 over the reflect-padded
    volume + noise: the discriminator's, discriminator.py:50-60): forward with bias, statistics and the finalisation tail, weight and
    bias gradient -- against a float64 reference on the same rounded operands, and against the kernels they replace in the 16-bit
    builds (VG_C1M=0: the VALU kernels of vg_pointwise.hip / the W-packed generic MFMA kernels), which stay the exact-parity mode's.
    Tolerances: bf16 output rounding for the forward (close_bf16), fp32 summation order for the gradients (2e-3 relative L2
    against float64; 1e-4 between the two kernel families)."""
    from van_gan_amd import ops, _lib
    from van_gan_amd.ops import Src
    dev = _dev()
    N, cout, stride = 2, (16 if k == 3 else 64), k - 2
    st, lay = make_layer(k, 1, cout, stride, pad, dims, seed=3)
    odims = tuple(lay.out_dims)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, *dims, 1, generator=g)
    xs = x if src_f32 else x.to(torch.bfloat16)
    scale = (torch.rand(N, 1, generator=g) + 0.5) if transform else None
    shift = (torch.randn(N, 1, generator=g) * 0.3) if transform else None
    nz = (torch.randn(N, *[n + 2 for n in dims], 1, generator=g) * 0.1).to(torch.bfloat16) if noise else None
    src = Src(xs.to(dev), (N,) + dims, 1, f32=src_f32, scale=None if scale is None else scale.to(dev),
              shift=None if shift is None else shift.to(dev), act=ops.ACT_LRELU if transform else ops.ACT_NONE,
              noise=None if nz is None else nz.to(dev), noise_pad=1 if noise else 0)
    with ops.DryRun() as dry:                       # the launches below are the MFMA kernels' (then, with VG_C1M=0, the older ones')
        lay.forward(src, torch.zeros(N, *odims, cout, dtype=torch.bfloat16, device=dev), sums=torch.zeros(8, N, cout, 2, device=dev))
        lay.wgrad(src, torch.zeros(N, *odims, cout, dtype=torch.bfloat16, device=dev))
    assert [v.split('<')[0] for _, _, v in dry.records] == ['c1m_fwd', 'c1m_wgrad'], dry.records
    res = {}
    for mode in (1, 0):
        _lib.lib.vg_set_tuning(b'C1M', mode, 0)
        try:
            out = torch.zeros(N, *odims, cout, dtype=torch.bfloat16, device=dev)
            sums = torch.zeros(8, N, cout, 2, device=dev)
            lay.forward(src, out, sums=sums)
            st.g.zero_()
            dy = torch.randn(N, *odims, cout, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16)
            lay.wgrad(src, dy.to(dev))
            torch.cuda.synchronize()
            res[mode] = (out.float().cpu(), sums.sum(0).cpu(), st.grad('c.w').clone().cpu(), st.grad('c.b').clone().cpu())
        finally:
            _lib.lib.vg_set_tuning(b'C1M', 0, 1)
    a = xs.double()
    if transform:
        a = F.leaky_relu(a * scale.double().view(N, 1, 1, 1, 1) + shift.double().view(N, 1, 1, 1, 1), 0.2)
    a = O.to_ncdhw(a)
    if pad == 'reflect':
        a = O.reflect_pad1(a)
    if noise:
        a = a + O.to_ncdhw(nz.double())
    a = bf(a)
    w = bf(st.param('c.w').cpu()).requires_grad_(True)
    b = st.param('c.b').cpu().double().requires_grad_(True)
    y = O.to_ndhwc(O.conv3d(a, w, b, stride, 'valid' if pad == 'reflect' else 'same'))
    (y * bf(dy)).sum().backward()
    out, sums, gw, gb = res[1]
    close_bf16(out, y, 'forward')
    yq = bf(y.detach())
    ref_sums = torch.stack([yq.sum(dim=(1, 2, 3)), (yq ** 2).sum(dim=(1, 2, 3))], dim=-1)
    assert rel_l2(sums, ref_sums) < 1e-4
    assert rel_l2(gw, w.grad) < 2e-3 and rel_l2(gb, b.grad) < 2e-3
    # the two kernel families multiply the same rounded operands: only the fp32 summation order differs
    o0, s0, gw0, gb0 = res[0]
    assert float((out - o0).abs().max()) <= 2 ** -7 * float(o0.abs().max())           # at most one bf16 ulp of the largest value
    assert rel_l2(out, o0) < 2e-3 and rel_l2(sums, s0) < 1e-4
    assert rel_l2(gw, gw0) < 1e-4 and rel_l2(gb, gb0) < 1e-4


@pytest.mark.parametrize('dims,N', [((8, 12, 36), 2), ((4, 4, 4), 1), ((128, 128, 128), 1)])
def test_strided_single_channel_data_gradient_in_cell_form(dims, N):
    """ConvLayer.dgrad_input of the discriminators' first layer (4x4x4 stride 2 over the reflect-padded single-channel volume,
    discriminator.py:50-60; the generator loss reaches the generators through it, vangan.py:433-438): a stride-1 convolution of dY
    over 2x2x2 cells of the padded grid on the thin-channel specialist + vg_cells_fold, against (a) float64 autograd through
    reflection pad + convolution on the same bf16-rounded operands (small grids) and (b) the generic form -- 8 output-parity classes
    onto the padded grid + the fold of vg_actnorm_bwd -- at every size incl. the 128^3 patch of BASELINE config 1.  Both forms round
    each padded-grid value to bf16 once and add up to 8 of them in fp32: they differ by fp32 summation order inside the MFMA only."""
    from van_gan_amd import ops
    from van_gan_amd.ops import Arena
    dev = _dev()
    cout = 64
    st, lay = make_layer(4, 1, cout, 2, 'reflect', dims, seed=5)
    assert lay.cell is not None
    odims = tuple(lay.out_dims)
    dy = torch.randn(N, *odims, cout, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16)
    ar = Arena(1 << 28, dev)
    dx = torch.full((N,) + dims + (1,), 7.0, device=dev)
    lay.dgrad_input(ar, dy.to(dev), N, dx)
    # the generic form
    dp = torch.zeros((N,) + tuple(lay.buf_dims) + (1,), dtype=torch.bfloat16, device=dev)
    lay.dgrad(dy.to(dev), N, dp, accumulate=False)
    dx0 = torch.full((N,) + dims + (1,), 7.0, device=dev)
    ops.actnorm_bwd(dp, True, None, (N,) + dims, 1, dx0, act=ops.ACT_NONE, norm=False, accumulate=False)
    torch.cuda.synchronize()
    assert rel_l2(dx, dx0) < 2e-3
    assert float((dx - dx0).abs().max()) <= 2e-2 * float(dx0.abs().max())
    if math.prod(dims) <= 1 << 16:
        x = torch.zeros(N, 1, *dims, dtype=torch.float64, requires_grad=True)
        w = bf(st.param('c.w').cpu())
        (O.conv3d(O.reflect_pad1(x), w, None, 2, 'valid') * O.to_ncdhw(bf(dy))).sum().backward()
        ref = O.to_ndhwc(x.grad)
        got = dx.double().cpu()
        tol = 2.5e-2 * ref.abs() + 8e-3 * ref.abs().max()
        assert ((got - ref).abs() <= tol).all(), 'max err %.3e of %.3e' % (float((got - ref).abs().max()), float(ref.abs().max()))


@pytest.mark.parametrize('cin,cat,dims', [(16, None, (64, 64, 64)), (48, (32, 16), (64, 64, 64)), (16, None, (42, 70, 40)), (48, (32, 16), (34, 66, 40))])
def test_thin_layer_weight_gradient_families_agree_and_are_deterministic(cin, cat, dims):
    """wgrad_thin_kernel (vg_conv_thin.hip: the waves split the voxels and hold the whole 27 x 16 x 16 slab; operand staged by the forward
    kernel's routine; slabs added in a fixed two-level order) against wgrad_dma_kernel<.,1,DIRECT> (VG_WGRAD_THIN=0) on a 64^3 layer
    with the decoder's virtual upsample + concat source, IN affine + ReLU on read: the same rounded operands, fp32 sums in a different
    order (rel 1e-4); two launches of the new kernel give bitwise the same gradient (no float atomics between workgroups); grids that
    are no multiple of the 16 x 8 x 4 tile in any axis (masked dY, clamped staging).  Parity with the float64 oracle at the true layer
    shapes: tests/test_gpu_layers.py (every variant of the BASELINE configurations)."""
    from van_gan_amd import ops, _lib
    from van_gan_amd.ops import Src
    dev = _dev()
    N, cout = 2, 16
    st, lay = make_layer(3, cin, cout, 1, 'reflect', dims, seed=9)
    g = torch.Generator().manual_seed(4)
    scale, shift = (torch.rand(N, cin, generator=g) + 0.5).to(dev), (torch.randn(N, cin, generator=g) * 0.2).to(dev)
    if cat:
        low = torch.randn(N, *[d // 2 for d in dims], cat[0], generator=g).to(torch.bfloat16).to(dev)
        skip = torch.randn(N, *dims, cat[1], generator=g).to(torch.bfloat16).to(dev)
        src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=scale, shift=shift, act=ops.ACT_RELU)
    else:
        src = Src(torch.randn(N, *dims, cin, generator=g).to(torch.bfloat16).to(dev), (N,) + dims, cin, scale=scale, shift=shift, act=ops.ACT_RELU)
    dy = torch.randn(N, *dims, cout, generator=g).to(torch.bfloat16).to(dev)
    with ops.DryRun() as dry:
        lay.wgrad(src, dy)
    assert dry.records[0][2].startswith('wgrad_thin<m1>'), dry.records
    res = []
    for mode in (1, 1, 0):
        _lib.lib.vg_set_tuning(b'WGRAD_THIN', mode, 0)
        try:
            st.g.zero_()
            lay.wgrad(src, dy)
            torch.cuda.synchronize()
            res.append((st.grad('c.w').clone(), st.grad('c.b').clone()))
        finally:
            _lib.lib.vg_set_tuning(b'WGRAD_THIN', 0, 1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[0][0].abs().max()) > 0
    assert rel_l2(res[0][0], res[2][0]) < 1e-4 and rel_l2(res[0][1], res[2][1]) < 1e-4


@pytest.mark.parametrize('dims,N,f32', [((32, 32, 32), 2, False), ((16, 24, 40), 1, False), ((12, 10, 20), 2, False), ((16, 16, 32), 2, True)],
                         ids=['32^3 thin specialist', '16x24x40 thin, ragged tiles', '12x10x20 generic kernel (masked)', 'fp32 generic kernel'])
def test_stem_shortcut_as_an_affine_function_of_the_volume(dims, N, f32):
    """The stem's shortcut (resunet_model.py:96-99: Conv3D(16, 1x1x1)(x) -> InstanceNorm on the single-channel volume) without its tensor:
    vg_stem_short_fwd turns the volume's mean / variance into scale * x + shift per (sample, channel), and the block's second convolution
    adds that in its epilogue from the fp32 volume (vg_conv_desc::res_c1) -- in the 16-channel specialist and in the generic kernel.
    Against float64: conv(relu(IN(c1))) + IN(conv1x1(x)), computed the reference's way."""
    from van_gan_amd import ops
    from van_gan_amd.ops import Arena, Src
    dev = _dev()
    dt = torch.float32 if f32 else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    from van_gan_amd.nets import ParamStore
    from van_gan_amd.ops import ConvLayer
    specs = [('c.w', (3, 3, 3, 16, 16), 'x'), ('c.b', (16,), 'x')]
    st = ParamStore(specs, dev)
    st.param('c.w').copy_(torch.randn(3, 3, 3, 16, 16, generator=g) / math.sqrt(27 * 16))
    st.param('c.b').copy_(torch.randn(16, generator=g) * 0.1)
    lay = ConvLayer(st, 'c', 3, 16, 16, 1, 'reflect', True, dims, dtype=dt)
    lay.pack()
    x = torch.rand(N, *dims, 1, generator=g) * 2 - 1
    x[1 % N] = x[1 % N] * 0.3 + 0.5                            # samples with different mean / variance
    c1 = torch.randn(N, *dims, 16, generator=g).to(dt)
    scale, shift = torch.rand(N, 16, generator=g) + 0.5, torch.randn(N, 16, generator=g) * 0.3
    ws = torch.randn(16, generator=g) * 0.4
    ws[3] = 0.01                                               # a channel whose stored 16-bit branch would have had a handful of levels
    bs, gam, bet = torch.randn(16, generator=g) * 0.2, torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    ar = Arena(64 << 20, dev)
    sc_scale, sc_shift = torch.zeros(N, 16, device=dev), torch.zeros(N, 16, device=dev)
    xd = x.to(dev)
    ops.stem_short_fwd(ar, xd, N, 16, ws.to(dev), gam.to(dev), bet.to(dev), sc_scale, sc_shift, round16=not f32)
    src = Src(c1.to(dev), (N,) + dims, 16, scale=scale.to(dev), shift=shift.to(dev), act=ops.ACT_RELU)
    out = torch.zeros(N, *dims, 16, dtype=dt, device=dev)
    sums = torch.zeros(8, N, 16, 2, device=dev)
    lay.forward(src, out, sums=sums, res=xd, res_scale=sc_scale, res_shift=sc_shift, res_c1=True)
    torch.cuda.synchronize()
    q = (lambda t: t.double()) if f32 else bf
    wq = q(ws)
    sc = x.double() * wq.view(1, 1, 1, 1, -1) + bs.double().view(1, 1, 1, 1, -1)                  # the branch as the reference computes it
    scn = O.to_ndhwc(O.instance_norm(O.to_ncdhw(sc), gam.double(), bet.double()))
    # scale / shift themselves
    mu, var = x.double().mean((1, 2, 3, 4)), x.double().var((1, 2, 3, 4), unbiased=False)
    rs = (wq[None] ** 2 * var[:, None] + 1e-3).rsqrt()
    assert rel_l2(sc_scale, gam.double()[None] * wq[None] * rs) < 1e-5
    assert rel_l2(sc_shift, bet.double()[None] - gam.double()[None] * wq[None] * rs * mu[:, None]) < 1e-5
    a = F.relu(c1.double() * scale.double().view(N, 1, 1, 1, -1) + shift.double().view(N, 1, 1, 1, -1))
    y = O.to_ndhwc(ref_conv(O.to_ncdhw(q(a)), q(st.param('c.w').cpu()), st.param('c.b').cpu().double(), 1, 'reflect')) + scn
    if f32:
        assert rel_l2(out, y) < 1e-4
    else:
        close_bf16(out, y, 'stem.cb + shortcut on read')
    # the output statistics the launch accumulated are those of the stored values
    o = out.double().cpu()
    assert rel_l2(sums.sum(0)[..., 0], o.sum((1, 2, 3))) < 1e-4 and rel_l2(sums.sum(0)[..., 1], (o ** 2).sum((1, 2, 3))) < 1e-4


def test_collapsed_upsampled_chunks_of_the_decoder_convolution():
    """conv_thin_kernel<..., UP> (off by default, VG_CONV_THIN_UP=1): the upsampled channel chunks of a decoder block's first convolution
    contracted over the half-resolution image with the D / H taps collapsed (UpSampling3D -> concatenate -> IN -> ReLU -> reflect pad ->
    3x3x3 convolution, resunet_model.py:175-181, 42-66): against the plain 27-tap form of the SAME launch (bitwise-close: only the single
    rounding of the summed weights differs) and against the float64 oracle on the virtual concat, incl. the grid's border tiles where the
    reflection pad turns into edge replication at half resolution."""
    from van_gan_amd import ops
    from van_gan_amd.ops import Src
    dev = _dev()
    N, dims, cu, cs, cout = 1, (64, 64, 64), 32, 16, 16             # (the layer planner hands this shape to the specialist with 16-channel chunks)
    old = os.environ.get('VG_CONV_THIN_UP')
    outs = {}
    try:
        for up in (0, 1):
            os.environ['VG_CONV_THIN_UP'] = str(up)
            ops._lib.lib.vg_set_tuning(b'CONV_THIN_UP', up, 0)
            st, lay = make_layer(3, cu + cs, cout, 1, 'reflect', dims, seed=5)
            lay.enable_up(cu)
            lay.pack()
            assert (lay.wp_up is not None) == bool(up)
            g = torch.Generator().manual_seed(2)
            low = torch.randn(N, *(n // 2 for n in dims), cu, generator=g).to(torch.bfloat16)
            skip = torch.randn(N, *dims, cs, generator=g).to(torch.bfloat16)
            scale = torch.rand(N, cu + cs, generator=g) + 0.5
            shift = torch.randn(N, cu + cs, generator=g) * 0.3
            src = Src(low.to(dev), (N,) + dims, cu, skip.to(dev), cs, shift0=1, scale=scale.to(dev), shift=shift.to(dev), act=ops.ACT_RELU)
            out = torch.zeros(N, *dims, cout, dtype=torch.bfloat16, device=dev)
            sums = torch.zeros(8, N, cout, 2, device=dev)
            with ops.DryRun() as dr:
                lay.forward(src, out, sums=sums)
            assert (',up>' in dr.records[0][2]) == bool(up), dr.records
            lay.forward(src, out, sums=sums)
            torch.cuda.synchronize()
            outs[up] = out.double().cpu()
    finally:
        ops._lib.lib.vg_set_tuning(b'CONV_THIN_UP', 0, 1)
        if old is None:
            os.environ.pop('VG_CONV_THIN_UP', None)
        else:
            os.environ['VG_CONV_THIN_UP'] = old
    upx = low.double().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    cat = torch.cat([upx, skip.double()], dim=-1)
    a = F.relu(cat * scale.double().view(N, 1, 1, 1, -1) + shift.double().view(N, 1, 1, 1, -1))
    y = O.to_ndhwc(ref_conv(O.to_ncdhw(bf(a)), bf(st.param('c.w').cpu()), st.param('c.b').cpu().double(), 1, 'reflect'))
    close_bf16(outs[0], y, '27-tap form')
    close_bf16(outs[1], y, 'collapsed form')
    assert rel_l2(outs[1], outs[0]) < 6e-3          # two bf16 results of the same sums, one with the weights summed before rounding
