"""Which kernel variants do the BASELINE configurations run, and a parity case at a TRUE LAYER SHAPE for each of them.

The schedules of van_gan_amd/nets.py (generator forward + backward, discriminator forward + both backward sweeps) are walked
on the CPU in `ops.DryRun` mode: every vg_conv3d / vg_conv3d_wgrad call runs its complete host-side dispatch without
launching and reports the kernel variant it selected (template arguments + run-time regime, include/vangan_hip.h:
vg_conv3d_variant), together with a shape-level recipe of the call.  For every variant the cheapest call that selects it
becomes its representative; tests/test_gpu_layers.py replays the representative on the GPU with random contents and
compares it with the oracle's convolution (autograd for the two gradients), tests/test_variant_coverage.py (CPU) proves the
representatives cover every variant of BASELINE configs 2, 3 and 4.

Helper module (no tests in here)."""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

# name -> (patch dims, per-replica batch).  The first three are BASELINE.json configs 2, 3 and 4 (per-GPU workload of the
# 8-GPU run); 32^3 (config 1) only donates cheaper representatives of variants it shares with them.
CONFIGS = {
    '128^3 B1': ((128, 128, 128), 1),
    '64^3 B2': ((64, 64, 64), 2),
    '128x128x64 B2': ((128, 128, 64), 2),
    '32^3 B1': ((32, 32, 32), 1),
}
NEEDED = ('128^3 B1', '64^3 B2', '128x128x64 B2')


def enumerate_config(dims, B, precision='bf16'):
    """-> [(kind, layer name, variant, recipe)] of one train step's conv launches (one generator: a forward application and the
    backward sweep over both applications; one discriminator with its two backward sweeps: the others repeat the same calls)."""
    from van_gan_amd import ops
    from van_gan_amd.nets import ParamStore, PatchGAN, ResUNet, disc_param_specs, gen_param_specs
    dt = torch.bfloat16 if precision == 'bf16' else torch.float32
    S = dims[0] * dims[1] * dims[2]
    G = ResUNet(ParamStore(gen_param_specs(), 'cpu'), dims, dt)
    D = PatchGAN(ParamStore(disc_param_specs(), 'cpu'), dims, dt)
    ar = ops.Arena(int(B * S * 5200 * 2) + (512 << 20), 'cpu')          # never touched: dry runs do not write
    with ops.DryRun() as dry:
        # as in VanGan._losses_and_backward: the two applications of a generator are B-sample forward passes into paired 2B-sample
        # tensors and ONE 2B-sample backward sweep (the second forward repeats the first one's calls: not recorded twice)
        from van_gan_amd.nets import pair_ctx
        x2 = ar.alloc((2 * B,) + dims + (1,), torch.float32)
        y, yb = ar.alloc((B,) + dims + (1,), torch.float32), ar.alloc((B,) + dims + (1,), torch.float32)
        ar.pair_begin('g', 0)
        ctx = G.forward(ar, x2[:B], y)
        ar.pair_end()
        n0 = len(dry.records)
        ar.pair_begin('g', 1)
        G.forward(ar, x2[B:], yb)
        ar.pair_end()
        del dry.records[n0:]; del dry.recipes[n0:]
        G.backward(ar, pair_ctx(ar, ctx, x2, (y, yb), G.lv[0]), x2)
        lg = ar.alloc((2 * B,) + tuple(n // 8 for n in dims) + (1,), torch.float32)
        noise = {k: torch.empty(shp, dtype=torch.bfloat16) for k, shp in D.noise_shapes(2 * B).items()}
        drop = {k: torch.empty(2 * B, c) for k, c in (('down0', 128), ('down1', 256), ('down2', 512))}
        dctx = D.forward(ar, x2, lg, noise, drop)
        # the step's discriminator backward: ONE sweep over [d critic loss (2B); d generator loss (B)] (PatchGAN.backward_both)
        lg3 = ar.alloc((3 * B,) + tuple(n // 8 for n in dims) + (1,), torch.float32)
        D.backward_both(ar, dctx, lg3, B, ar.alloc((B,) + dims + (1,), torch.float32))
    return [(k, n, v, r) for (k, n, v), r in zip(dry.records, dry.recipes)]


def enumerate_resnet(dims, N):
    """[(kind, layer, variant)] of one application (forward + backward) of the ResNet generator in dry-run mode."""
    from van_gan_amd import ops
    from van_gan_amd.nets import ParamStore, ResNetGenerator, resnet_param_specs
    G = ResNetGenerator(ParamStore(resnet_param_specs(), 'cpu'), dims, torch.bfloat16)
    ar = ops.Arena(int(N * dims[0] * dims[1] * dims[2] * 9000) + (512 << 20), 'cpu')
    with ops.DryRun() as dry:
        x, y = torch.empty((N,) + dims + (1,)), torch.empty((N,) + dims + (1,))
        drop = {k: torch.empty(N, c) for k, c in ResNetGenerator.DROP_CH.items()}
        G.backward(ar, G.forward(ar, x, y, drop), y)
    return list(dry.records)


def recipe_macs(r) -> float:
    L = r['layer']
    out = [(-(-n // L['stride'])) for n in L['in_dims']]
    N = r['src']['N'] if 'src' in r else r['N']
    return float(N) * math.prod(out) * L['cin'] * L['cout'] * L['k'] ** 3


@functools.lru_cache(maxsize=None)
def all_records(precision='bf16'):
    return {name: enumerate_config(dims, B, precision) for name, (dims, B) in CONFIGS.items()}


@functools.lru_cache(maxsize=None)
def representatives(precision='bf16') -> Dict[Tuple[str, str], dict]:
    """(kind, variant) -> {'recipe', 'config', 'layer', 'macs'}: the cheapest call that selects the variant."""
    best: Dict[Tuple[str, str], dict] = {}
    for cfg, recs in all_records(precision).items():
        for kind, name, variant, recipe in recs:
            key, m = (kind, variant), recipe_macs(recipe)
            if key not in best or m < best[key]['macs']:
                best[key] = dict(recipe=recipe, config=cfg, layer=name, macs=m)
    return best


def needed_variants(precision='bf16'):
    recs = all_records(precision)
    return sorted({(k, v) for cfg in NEEDED for (k, _, v, _) in recs[cfg]})


# ----------------------------------------------------------------------------------------------------------------------
# replaying a recipe on the GPU against the CPU oracle
# ----------------------------------------------------------------------------------------------------------------------
def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def _ref_dtype(macs):
    return torch.float64 if macs < 3e9 else torch.float32      # float32 CPU accumulation error (~1e-6) << one bf16 rounding


def make_layer_from(ctor, dev, seed=0, dtype=torch.bfloat16):
    from van_gan_amd.nets import ParamStore
    from van_gan_amd.ops import ConvLayer
    k, cin, cout = ctor['k'], ctor['cin'], ctor['cout']
    specs = [('c.w', (k, k, k, cin, cout), 'x')] + ([('c.b', (cout,), 'x')] if ctor['bias'] else [])
    st = ParamStore(specs, dev)
    g = torch.Generator().manual_seed(seed)
    st.param('c.w').copy_(torch.randn(k, k, k, cin, cout, generator=g) / math.sqrt(k ** 3 * cin))
    if ctor['bias']:
        st.param('c.b').copy_(torch.randn(cout, generator=g) * 0.1)
    lay = ConvLayer(st, 'c', k, cin, cout, ctor['stride'], ctor['pad'], ctor['bias'], ctor['in_dims'],
                    need_dgrad=ctor['need_dgrad'], dtype=dtype)
    if ctor.get('c_up'):
        lay.enable_up(ctor['c_up'])
    return st, lay


def make_operand(sr, pad, dev, g):
    """Random contents for a Src recipe -> (Src on dev, host tensors)."""
    from van_gan_amd.ops import Src
    N, dims, c0, c1, sh = sr['N'], tuple(sr['dims']), sr['c0'], sr['c1'], sr['shift0']
    C_ = c0 + c1
    low = tuple(n >> sh for n in dims)
    x0 = torch.randn(N, *low, c0, generator=g)
    x0 = x0 if sr['f32'] else x0.to(torch.bfloat16)
    x1 = torch.randn(N, *dims, c1, generator=g).to(torch.bfloat16) if c1 else None
    scale = (torch.rand(N, C_, generator=g) + 0.5) if sr['affine'] else None
    shift = (torch.randn(N, C_, generator=g) * 0.3) if sr['affine'] else None
    noise = None
    if sr['noise']:
        npd = sr['noise_pad']
        noise = (torch.randn(N, *[n + 2 * npd for n in dims], C_, generator=g) * 0.1).to(torch.bfloat16)
    d = lambda t: None if t is None else t.to(dev)
    src = Src(d(x0), (N,) + dims, c0, d(x1), c1, shift0=sh, f32=sr['f32'], scale=d(scale), shift=d(shift), act=sr['act'],
              noise=d(noise), noise_pad=sr['noise_pad'])
    return src, dict(x0=x0, x1=x1, scale=scale, shift=shift, noise=noise)


def ref_operand(sr, host, pad, dt):
    """The convolution's input as the kernels stage it: act(IN-affine(virtual upsample + concat)) [-> reflection pad]
    + noise, rounded to bf16.  NCDHW, on the padded grid for 'reflect'."""
    from oracle import vangan_oracle as O
    from van_gan_amd import ops
    x = host['x0'].to(dt)
    if sr['shift0']:
        x = x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)       # UpSampling3D(2), nearest
    if host['x1'] is not None:
        x = torch.cat([x, host['x1'].to(dt)], dim=-1)                                      # concatenate([x_up, skip])
    N, C_ = x.shape[0], x.shape[-1]
    if host['scale'] is not None:
        x = x * host['scale'].to(dt).view(N, 1, 1, 1, C_) + host['shift'].to(dt).view(N, 1, 1, 1, C_)
    if sr['act'] == ops.ACT_RELU:
        x = F.relu(x)
    elif sr['act'] == ops.ACT_LRELU:
        x = F.leaky_relu(x, 0.2)
    a = O.to_ncdhw(x)
    if pad == 'reflect':
        a = O.reflect_pad1(a)
    if host['noise'] is not None:
        a = a + O.to_ncdhw(host['noise'].to(dt))
    return bf(a)


def close_bf16(got, ref, name='', rel=1.2e-2, floor=2e-3):
    got, ref = got.double().cpu(), ref.double().cpu()
    tol = rel * ref.abs() + floor * ref.abs().max() + 1e-30
    bad = (got - ref).abs() > tol
    assert not bad.any(), '%s: %d/%d outside tolerance, max err %.3e (max ref %.3e)' % (
        name, int(bad.sum()), bad.numel(), float((got - ref).abs().max()), float(ref.abs().max()))


def rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def dry_variants(fn):
    """Variant strings `fn()` would launch (fn issues ConvLayer calls)."""
    from van_gan_amd import ops
    with ops.DryRun() as dry:
        fn()
    return [v for (_, _, v) in dry.records]


def run_recipe(recipe, expect_variant, dev, seed=1, precision='bf16'):
    """Replay one recorded call on `dev` with random contents and compare with the oracle.  Tolerances as in
    tests/test_gpu_ops.py: identical bf16-rounded operands into the reference, so only fp32 accumulation order and the one
    rounding of the stored output differ."""
    from oracle import vangan_oracle as O
    L = recipe['layer']
    kind, pad, stride = recipe['kind'], L['pad'], L['stride']
    dt = _ref_dtype(recipe_macs(recipe))
    g = torch.Generator().manual_seed(seed)
    st, lay = make_layer_from(L, dev)
    lay.pack()
    w_ref = bf(st.param('c.w').cpu().to(dt))
    b_ref = st.param('c.b').cpu().to(dt) if L['bias'] else None
    conv_pad = 'valid' if pad == 'reflect' else 'same'
    if kind in ('fwd', 'wgrad'):
        sr = recipe['src']
        N = sr['N']
        src, host = make_operand(sr, pad, dev, g)
        ap = ref_operand(sr, host, pad, dt)
    if kind == 'fwd':
        odt = torch.float32 if recipe['out_f32'] else torch.bfloat16
        out = torch.zeros(N, *lay.out_dims, L['cout'], dtype=odt, device=dev)
        sums = torch.zeros(8, N, L['cout'], 2, device=dev) if recipe['sums'] else None
        res = rs = rb = None
        if recipe['res']:
            res = torch.randn(N, *lay.out_dims, L['cout'], generator=g).to(torch.bfloat16)
            rs, rb = torch.rand(N, L['cout'], generator=g) + 0.5, torch.randn(N, L['cout'], generator=g)
        call = lambda: lay.forward(src, out, sums=sums, res=None if res is None else res.to(dev),
                                   res_scale=None if rs is None else rs.to(dev), res_shift=None if rb is None else rb.to(dev),
                                   tanh=recipe['tanh'])
        assert dry_variants(call) == [expect_variant]
        call()
        torch.cuda.synchronize()
        y = O.to_ndhwc(O.conv3d(ap, w_ref, b_ref, stride, conv_pad))
        if res is not None:
            y = y + res.to(dt) * rs.to(dt).view(N, 1, 1, 1, -1) + rb.to(dt).view(N, 1, 1, 1, -1)
        if recipe['tanh']:
            y = torch.tanh(y)
        if recipe['out_f32']:
            assert rel_l2(out, y) < 1e-4, 'forward (f32 output) rel %.2e' % rel_l2(out, y)
        else:
            close_bf16(out, y, 'forward')
        if sums is not None:
            yq = bf(y) if not recipe['out_f32'] else y
            ref_sums = torch.stack([yq.sum(dim=(1, 2, 3)), (yq ** 2).sum(dim=(1, 2, 3))], dim=-1)
            assert rel_l2(sums.sum(0), ref_sums) < 2e-2, 'IN statistics'
        return
    odims = lay.out_dims
    dy = torch.randn(recipe['src']['N'] if kind == 'wgrad' else recipe['N'], *odims, L['cout'], generator=g)
    dys = dy if recipe['dy_f32'] else dy.to(torch.bfloat16)
    dy_ref = O.to_ncdhw(bf(dys.to(dt)))                 # the kernels round dY to bf16 when they stage it
    if kind == 'wgrad':
        call = lambda: lay.wgrad(src, dys.to(dev))
        assert dry_variants(call) == [expect_variant]
        st.g.zero_()
        call()
        torch.cuda.synchronize()
        w = w_ref.clone().requires_grad_(True)
        (O.conv3d(ap, w, None, stride, conv_pad) * dy_ref).sum().backward()
        assert rel_l2(st.grad('c.w'), w.grad) < 2e-3, 'weight gradient rel %.2e' % rel_l2(st.grad('c.w'), w.grad)
        if L['bias']:
            assert rel_l2(st.grad('c.b'), dy_ref.sum(dim=(0, 2, 3, 4))) < 2e-3, 'bias gradient'
        return
    # data gradient: on the reflect-PADDED grid for 'reflect' convs (the fold is a separate kernel), plain grid for 'same'
    N = recipe['N']
    odt = torch.float32 if recipe['out_f32'] else torch.bfloat16
    prior = None
    if recipe['accumulate']:
        prior = torch.randn(N, *lay.buf_dims, L['cin'], generator=g).to(odt)
        dp = prior.to(dev).clone()
    else:
        dp = torch.full((N,) + tuple(lay.buf_dims) + (L['cin'],), 7.0, dtype=odt, device=dev)       # must be overwritten everywhere
    bs = recipe.get('bstat')
    bdesc = None
    if bs:
        # the IN backward that consumes this data gradient: its statistics ride on the launch (ConvLayer.dgrad(bstat=...))
        from van_gan_amd import ops
        Cc, dims_in = L['cin'], tuple(lay.in_dims)
        c0 = bs['c_x0'] if bs['cat'] else Cc
        sh = 1 if bs['cat'] else 0
        bx0 = torch.randn(N, *[n >> sh for n in dims_in], c0, generator=g).to(torch.bfloat16)
        bx1 = torch.randn(N, *dims_in, Cc - c0, generator=g).to(torch.bfloat16) if bs['cat'] else None
        bsc, bsf = torch.rand(N, Cc, generator=g) + 0.5, torch.randn(N, Cc, generator=g) * 0.3
        bmu, brs = torch.randn(N, Cc, generator=g) * 0.2, torch.rand(N, Cc, generator=g) + 0.5
        red = torch.zeros(ops.STRIPES * N * Cc * 2 + 4, device=dev)
        dgam, dbet = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        dxo = torch.zeros(N, *dims_in, Cc, dtype=torch.bfloat16, device=dev)
        tod = lambda t: None if t is None else t.to(dev)
        bdesc = ops.actnorm_desc(dp, bs['pad'], tod(bx0), (N,) + dims_in, Cc, dxo, scale=tod(bsc), shift=tod(bsf), act=bs['act'], norm=True,
                                 gamma=torch.ones(Cc, device=dev), mean=tod(bmu), rstd=tod(brs), red=red, accumulate=False, x1=tod(bx1),
                                 c_x0=c0 if bs['cat'] else 0, x0_shift=sh, dgamma=dgam, dbeta=dbet)
    done = []
    call = lambda: done.append(lay.dgrad(dys.to(dev), N, dp, accumulate=recipe['accumulate'], bstat=bdesc))
    got_variants = dry_variants(call)
    assert expect_variant in got_variants and len(set(got_variants)) == 1, (got_variants, expect_variant)
    call()
    torch.cuda.synchronize()
    if bs:
        assert done[-1] is True
        # reference statistics from the data gradient the kernel stored: transpose of the reflection pad, dn = g * act'(pre)
        gp = O.to_ncdhw(dp.double().cpu())
        a0 = torch.zeros(N, L['cin'], *lay.in_dims, dtype=torch.float64, requires_grad=True)
        ((O.reflect_pad1(a0) if bs['pad'] else a0) * gp).sum().backward()
        gf = O.to_ndhwc(a0.grad)
        xf = bx0.double()
        if bs['cat']:
            xf = torch.cat([xf.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3), bx1.double()], dim=-1)
        v = lambda t: t.double().view(N, 1, 1, 1, -1)
        pre = xf * v(bsc) + v(bsf)
        slope = 0.0 if bs['act'] == ops.ACT_RELU else (0.2 if bs['act'] == ops.ACT_LRELU else 1.0)
        dn = gf * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
        xh = (xf - v(bmu)) * v(brs)
        ref_red = torch.stack([dn.sum(dim=(1, 2, 3)), (dn * xh).sum(dim=(1, 2, 3))], dim=-1)
        got_red = red[:ops.STRIPES * N * L['cin'] * 2].view(ops.STRIPES, N, L['cin'], 2).sum(0)       # striped: the apply pass adds them up
        assert rel_l2(got_red, ref_red) < 2e-3, 'fused IN-backward statistics rel %.2e' % rel_l2(got_red, ref_red)
        ops.actnorm_run(bdesc, stats_done=True)                                                     # apply: dx and the gamma / beta gradients
        torch.cuda.synchronize()
        assert rel_l2(dbet, ref_red[..., 0].sum(0)) < 2e-3 and rel_l2(dgam, ref_red[..., 1].sum(0)) < 2e-3
        ref_dx = v(brs) * (dn - ref_red[..., 0].view(N, 1, 1, 1, -1) / dn[0, ..., 0].numel() - xh * ref_red[..., 1].view(N, 1, 1, 1, -1) / dn[0, ..., 0].numel())
        assert rel_l2(dxo, ref_dx) < 1e-2, 'IN backward after fused statistics rel %.2e' % rel_l2(dxo, ref_dx)
    xin = torch.zeros(N, L['cin'], *lay.buf_dims, dtype=dt, requires_grad=True)
    (O.conv3d(xin, w_ref, None, stride, conv_pad) * dy_ref).sum().backward()
    ref = O.to_ndhwc(xin.grad)
    if prior is not None:
        ref = ref + prior.to(dt)
    close_bf16(dp, ref, 'data gradient', rel=2.5e-2, floor=6e-3)


def stress_recipe(recipe, expect_variant, dev, launches=200, seed=3):
    """K-split exchange under load: the recorded call `launches` times back to back, alternating between two streams (each with its
    own output buffer; the library's per-stream scratch -- partial tiles + arrival counters -- is reused by every launch of a
    stream), every output compared BITWISE with the first launch's.  The slices' partial tiles are added in slice order, so any
    difference is a partial tile read before it was complete, or a counter that did not return to zero."""
    L = recipe['layer']
    kind = recipe['kind']
    g = torch.Generator().manual_seed(seed)
    st, lay = make_layer_from(L, dev)
    lay.pack()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    if kind == 'fwd':
        sr = recipe['src']
        src, _ = make_operand(sr, L['pad'], dev, g)
        odt = torch.float32 if recipe['out_f32'] else torch.bfloat16
        outs = [torch.zeros(sr['N'], *lay.out_dims, L['cout'], dtype=odt, device=dev) for _ in range(3)]
        call = lambda o: lay.forward(src, o, tanh=recipe['tanh'])
    else:
        assert kind == 'dgrad' and not recipe['accumulate'] and not recipe.get('bstat')
        N = recipe['N']
        dy = torch.randn(N, *lay.out_dims, L['cout'], generator=g)
        dys = (dy if recipe['dy_f32'] else dy.to(torch.bfloat16)).to(dev)
        odt = torch.float32 if recipe['out_f32'] else torch.bfloat16
        outs = [torch.zeros((N,) + tuple(lay.buf_dims) + (L['cin'],), dtype=odt, device=dev) for _ in range(3)]
        call = lambda o: lay.dgrad(dys, N, o, accumulate=False)
    got = dry_variants(lambda: call(outs[2]))
    assert got == [expect_variant], (got, expect_variant)
    call(outs[2])
    torch.cuda.synchronize()
    ref = outs[2].clone()
    assert bool(torch.isfinite(ref.float()).all()) and float(ref.float().abs().max()) > 0
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    bad = 0
    for i in range(launches):
        with torch.cuda.stream(streams[i & 1]):
            outs[i & 1].fill_(7.0)
            call(outs[i & 1])
            bad = bad + (outs[i & 1] != ref).sum()             # device-side count: no host sync between the launches
    torch.cuda.synchronize()
    bad = int(bad)
    assert bad == 0, '%d elements differ from the first launch over %d launches on two streams' % (bad, launches)
