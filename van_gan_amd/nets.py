"""ResUNet generator and 3-D PatchGAN discriminator of VAN-GAN as explicit forward/backward schedules of
libvangan_hip.so kernels (no autograd, no torch arithmetic).

Reference: resunet_model.py:185-249 as configured at vangan.py:112-122,151-162 (filters=16, num_layers=4,
upsample_mode='simple', tanh) and discriminator.py:7-124 as configured at vangan.py:167-192.
Parameter names/order/layout (Keras DHWIO) follow oracle/vangan_oracle.py::gen_param_specs/disc_param_specs
so that weights can be exchanged with the oracle 1:1.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, Arena, ConvLayer, Src

GEN_F = [16, 32, 64, 128, 256]
_STEM_AUX = os.environ.get('VG_STEM_AUX', '1') != '0'      # the stem shortcut's kernel gradient in closed form from the statistics pass (backward_iter)
# The stem's shortcut is never materialised: Conv3D(16, 1x1x1)(x) -> InstanceNorm of the single-channel volume is an affine function of x per
# (sample, channel) -- vg_stem_short_fwd gives scale / shift from the volume's mean and variance, stem.cb's epilogue adds scale * x + shift
# (vg_conv_desc::res_c1).  No 1 -> 16 launch, no 16-channel tensor written and read back (64 B per voxel and application), and the branch
# is exact where the stored tensor was rounded to 16 bits.  0: the materialised branch (needed by the explicit backward, VG_STEM_AUX=0).
_STEM_FUSED = os.environ.get('VG_STEM_FUSED', '1') != '0'


def gen_param_specs() -> List[Tuple[str, Tuple[int, ...], str]]:
    f = GEN_F
    s: List[Tuple[str, Tuple[int, ...], str]] = []

    def conv(name, k, ci, co, init):
        s.append((name + '.w', (k, k, k, ci, co), init)); s.append((name + '.b', (co,), 'zeros'))

    def inorm(name, c):
        s.append((name + '.gamma', (c,), 'ones')); s.append((name + '.beta', (c,), 'zeros'))

    conv('stem.conv1', 3, 1, f[0], 'glorot_uniform')
    inorm('stem.cb.in', f[0]); conv('stem.cb.conv', 3, f[0], f[0], 'he_normal')
    conv('stem.short', 1, 1, f[0], 'glorot_uniform'); inorm('stem.short.in', f[0])

    def resblock(name, ci, co):
        inorm(name + '.cb1.in', ci); conv(name + '.cb1.conv', 3, ci, co, 'he_normal')
        inorm(name + '.cb2.in', co); conv(name + '.cb2.conv', 3, co, co, 'he_normal')
        conv(name + '.short', 1, ci, co, 'he_normal'); inorm(name + '.short.in', co)

    for e in range(1, 5):
        resblock('enc%d' % e, f[e - 1], f[e])
    inorm('bridge.cb1.in', f[4]); conv('bridge.cb1.conv', 3, f[4], f[4], 'he_normal')
    inorm('bridge.cb2.in', f[4]); conv('bridge.cb2.conv', 3, f[4], f[4], 'he_normal')
    for d in (3, 2, 1, 0):
        resblock('dec%d' % d, f[d + 1] + f[d], f[d])
    conv('out', 1, f[0], 1, 'glorot_uniform')
    return s


def disc_param_specs(wasserstein_patches: int = 0) -> List[Tuple[str, Tuple[int, ...], str]]:
    """wasserstein_patches = n > 0: + the Flatten -> Dropout(0.2) -> Dense(1) head of discriminator.py:116-119 over the n patch logits."""
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    s.append(('conv0.w', (4, 4, 4, 1, 64), 'he_normal')); s.append(('conv0.b', (64,), 'zeros'))
    s.append(('conv0.in.gamma', (64,), 'glorot_vec')); s.append(('conv0.in.beta', (64,), 'zeros'))
    ci = 64
    for i in range(3):
        co = ci * 2
        s.append(('down%d.w' % i, (4, 4, 4, ci, co), 'he_normal'))
        s.append(('down%d.in.gamma' % i, (co,), 'glorot_vec')); s.append(('down%d.in.beta' % i, (co,), 'zeros'))
        ci = co
    s.append(('out.w', (3, 3, 3, 512, 1), 'he_normal')); s.append(('out.b', (1,), 'zeros'))
    if wasserstein_patches:
        s.append(('dense.w', (wasserstein_patches, 1), 'glorot_dense')); s.append(('dense.b', (1,), 'zeros'))
    return s


class ParamStore:
    """One network's parameters as flat fp32 buffers (w, grad, Adam m/v) + per-tensor views.
    The flat gradient buffer is the RCCL all-reduce bucket of that network."""

    def __init__(self, specs, device):
        self.specs = specs
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        bounds = [0]
        for name, shape, _ in specs:
            n = int(math.prod(shape))
            self.offsets[name] = (off, tuple(shape))
            off += n
            bounds.append(off)
        self.total = off
        self.T = len(specs)
        self.w = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.seg_off = torch.tensor(bounds, dtype=torch.int64, device=device)
        # T squared norms + two partial sums per 4096-element block (vg_adam_clip adds the norms in a fixed order)
        self.norms = torch.zeros(self.T + 2 * ((self.total + 4095) // 4096), dtype=torch.float32, device=device)
        self.step = 0

    def _view(self, buf, name):
        off, shape = self.offsets[name]
        return buf[off:off + int(math.prod(shape))].view(*shape)

    def param(self, name):
        return self._view(self.w, name)

    def grad(self, name):
        return self._view(self.g, name)

    def load(self, tensors: Dict[str, torch.Tensor]):
        for name, (off, shape) in self.offsets.items():
            self.param(name).copy_(tensors[name].to(torch.float32).reshape(shape))

    def export(self, buf=None) -> Dict[str, torch.Tensor]:
        buf = self.w if buf is None else buf
        return {name: self._view(buf, name).detach().cpu().clone() for name in self.offsets}


def init_reference(store: ParamStore, seed: int):
    """Reference initialisers on the host (setup only): he_normal / glorot_uniform / zeros / ones
    (resunet_model.py:47,85-95,246; building_blocks.py:129; discriminator.py:11)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, init in store.specs:
        if init == 'zeros':
            t = torch.zeros(shape)
        elif init == 'ones':
            t = torch.ones(shape)
        elif init == 'glorot_vec':
            t = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(3.0 / shape[0])
        elif init == 'glorot_dense':        # Dense kernel [in, out]
            t = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / (shape[0] + shape[1]))
        elif init == 'he_vec':              # he_normal over a 1-D shape (C,): Keras' _compute_fans gives fan_in = fan_out = C
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=g)
            t = t * (math.sqrt(2.0 / shape[0]) / 0.87962566103423978)
        else:
            rf = shape[0] * shape[1] * shape[2]
            fan_in, fan_out = rf * shape[3], rf * shape[4]
            if init == 'glorot_uniform':
                t = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / (fan_in + fan_out))
            else:
                t = torch.empty(shape)
                torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=g)
                t = t * (math.sqrt(2.0 / fan_in) / 0.87962566103423978)
        out[name] = t
    store.load(out)


class Act:
    """A stored activation: bf16 data + per-(n,c) (sum, sumsq) accumulated by its producer's epilogue."""

    def __init__(self, arena: Arena, N, dims, C_, dtype=torch.bfloat16, want_sums=True):
        self.N, self.dims, self.C = N, tuple(dims), C_
        self.data = arena.alloc((N,) + tuple(dims) + (C_,), dtype)
        self.sums = arena.alloc((ops.STRIPES, N, C_, 2), torch.float32, zero=True) if want_sums else None
        self.count = float(dims[0] * dims[1] * dims[2])
        self.grad = None
        self.grad_init = False

    def alloc_grad(self, arena: Arena):
        """Gradient buffer, NOT zeroed: its first writer overwrites (first_write() says whether a writer is the first), every later
        one accumulates.  (The ten buffers of a generator sweep were 185 MB of memsets per application at 128^3.)"""
        if self.grad is None:
            self.grad = arena.alloc((self.N,) + self.dims + (self.C,), self.data.dtype)
            self.grad_init = False
        return self.grad

    def first_write(self) -> bool:
        """True exactly once after alloc_grad: the caller's launch must then OVERWRITE the buffer."""
        if self.grad_init:
            return False
        self.grad_init = True
        return True


class Norm:
    """InstanceNorm parameters + the per-application scale/shift/mean/rstd it produces."""

    def __init__(self, store: ParamStore, name: str, C_: int):
        self.name, self.C = name, C_
        self.gamma, self.beta = store.param(name + '.gamma'), store.param(name + '.beta')
        self.dgamma, self.dbeta = store.grad(name + '.gamma'), store.grad(name + '.beta')

    def state(self, arena: Arena, N: int, mult=None) -> dict:
        """The per-application arrays of this norm, to be filled by the launch(es) that produce its input (ops.fin_desc: the last
        workgroup of the producing convolution finalises the statistics -- no vg_in_finalize launch between producer and consumer)."""
        st = {k: arena.alloc((N, self.C), torch.float32) for k in ('scale', 'shift', 'mean', 'rstd')}
        st['mult'] = mult
        return st

    def job(self, st: dict, c_off: int = 0):
        """One consumer entry of ops.fin_desc: the producer's channels land at [c_off, c_off + its Cout) of this norm's arrays."""
        return (self.gamma, self.beta, st['mult'], st, c_off, self.C)

    def finalize(self, arena: Arena, a0: Act, a1: Optional[Act] = None, mult=None):
        N = a0.N
        # (four allocations with the sample index leading: the paired arena mode doubles the leading dimension)
        st = {k: arena.alloc((N, self.C), torch.float32) for k in ('scale', 'shift', 'mean', 'rstd')}
        ops.in_finalize(a0.sums, a0.C, a0.count, self.gamma, self.beta, N, st['scale'], st['shift'], st['mean'],
                        st['rstd'], sums1=None if a1 is None else a1.sums, c1=0 if a1 is None else a1.C,
                        count1=1.0 if a1 is None else a1.count, mult=mult)
        st['mult'] = mult
        return st


def run_to_end(it):
    """Exhaust a generator and return its value (the *_iter methods of the networks are generators so that the engine can enqueue
    two of them alternately; called directly they run in one go)."""
    try:
        while True:
            next(it)
    except StopIteration as e:
        return e.value


def pair_ctx(ar: Arena, ctx: dict, x_full: torch.Tensor, ys, lv0) -> dict:
    """The backward context of BOTH applications of a generator whose forward passes ran in the arena's paired mode
    (Arena.pair_begin): ctx is the first application's context; every stored tensor is replaced by the 2N-sample tensor it is the
    first half of (object identity between the entries is preserved: the gradient buffers hang on the Act objects).  x_full: the
    [2N, D, H, W, 1] input volumes of the two applications, ys: their two output volumes."""
    import copy
    memo = {}

    def full(t):
        f = ar.full_of(t)
        return t if f is None else f

    def m(o):
        if o is None or isinstance(o, (int, float, str, bool)):
            return o
        k = id(o)
        if k in memo:
            return memo[k]
        if isinstance(o, torch.Tensor):
            r = full(o)
        elif isinstance(o, Act):
            assert ar.full_of(o.data) is not None, 'activation was not allocated in paired mode'
            r = copy.copy(o)
            r.data = full(o.data); r.N = r.data.shape[0]; r.sums = None; r.grad = None; r.grad_init = False
        elif isinstance(o, Src):
            r = copy.copy(o)
            r.x0, r.x1, r.scale, r.shift = full(o.x0), full(o.x1), full(o.scale), full(o.shift)
            r.N = r.x0.shape[0]
        elif isinstance(o, dict):
            r = {}
            memo[k] = r
            for kk, v in o.items():
                r[kk] = m(v)
            return r
        elif isinstance(o, (tuple, list)):
            r = type(o)(m(v) for v in o)
        else:
            r = o
        memo[k] = r
        return r

    out = m(ctx)
    N2 = 2 * ctx['N']
    out['N'], out['x'], out['y'] = N2, x_full, list(ys)
    out['stem']['sx'] = Src(x_full, (N2,) + tuple(lv0), 1, f32=True)
    return out


# ======================================================================================================
# Generator
# ======================================================================================================
class ResUNet:
    def __init__(self, store: ParamStore, dims: Tuple[int, int, int], dtype: torch.dtype = torch.bfloat16,
                 upsample_mode: str = 'simple'):
        # resunet_model.py:185-249 as vangan.py:112-122,151-162 configures it: upsample_mode='simple' (UpSampling3D).  The
        # reference's other mode cannot be built by the reference itself: 'deconv' reflect-pads before its k2 s2
        # Conv3DTranspose (resunet_model.py:168-174, padding='valid' is never overridden at :241), which yields 2(S+2) voxels
        # per axis against the skip tensor's 2S, and Keras' concatenate raises on the mismatch -- so does this constructor.
        if upsample_mode == 'deconv':
            raise ValueError("upsample_mode='deconv': the reference builds ReflectionPadding3D + Conv3DTranspose(k2,s2,'valid') "
                             'with 2(S+2) outputs per axis and concatenates it with a 2S skip tensor (resunet_model.py:168-181): '
                             'a Concatenate shape error in Keras; only upsample_mode=\'simple\' is a working configuration')
        if upsample_mode != 'simple':
            raise ValueError("upsample_mode must be 'simple' (UpSampling3D(2), vangan.py:114,153)")
        self.dtype = dtype
        D, H, W = dims
        if any(n % 16 or n < 32 for n in dims):
            raise ValueError('spatial dims must be multiples of 16 and >= 32 (4 stride-2 stages + reflect pad)')
        self.store, self.dims = store, tuple(dims)
        f = GEN_F
        lv = [tuple(n >> i for n in dims) for i in range(5)]
        self.lv = lv
        L = self.L = {}
        Nn = self.Nn = {}
        L['stem.conv1'] = ConvLayer(store, 'stem.conv1', 3, 1, f[0], 1, 'reflect', True, lv[0], need_dgrad=False, dtype=self.dtype)
        L['stem.cb'] = ConvLayer(store, 'stem.cb.conv', 3, f[0], f[0], 1, 'reflect', True, lv[0], dtype=self.dtype)
        L['stem.short'] = ConvLayer(store, 'stem.short', 1, 1, f[0], 1, 'same', True, lv[0], need_dgrad=False, dtype=self.dtype)
        Nn['stem.cb'] = Norm(store, 'stem.cb.in', f[0]); Nn['stem.short'] = Norm(store, 'stem.short.in', f[0])

        def resblock(name, ci, co, stride, in_dims, out_dims):
            L[name + '.cb1'] = ConvLayer(store, name + '.cb1.conv', 3, ci, co, stride, 'reflect', True, in_dims, dtype=self.dtype)
            L[name + '.cb2'] = ConvLayer(store, name + '.cb2.conv', 3, co, co, 1, 'reflect', True, out_dims, dtype=self.dtype)
            L[name + '.short'] = ConvLayer(store, name + '.short', 1, ci, co, stride, 'same', True, in_dims, dtype=self.dtype)
            Nn[name + '.cb1'] = Norm(store, name + '.cb1.in', ci)
            Nn[name + '.cb2'] = Norm(store, name + '.cb2.in', co)
            Nn[name + '.short'] = Norm(store, name + '.short.in', co)

        for e in range(1, 5):
            resblock('enc%d' % e, f[e - 1], f[e], 2, lv[e - 1], lv[e])
        for b in ('bridge.cb1', 'bridge.cb2'):
            L[b] = ConvLayer(store, b + '.conv', 3, f[4], f[4], 1, 'reflect', True, lv[4], dtype=self.dtype)
            Nn[b] = Norm(store, b + '.in', f[4])
        for d in (3, 2, 1, 0):
            resblock('dec%d' % d, f[d + 1] + f[d], f[d], 1, lv[d], lv[d])
            L['dec%d.cb1' % d].enable_up(f[d + 1])            # its first f[d+1] input channels are the upsampled low-resolution tensor
        L['out'] = ConvLayer(store, 'out', 1, f[0], 1, 1, 'same', True, lv[0], dtype=self.dtype)

    def pack(self):
        if getattr(self, '_ptab', None) is None:
            self._ptab = ops.PackTable(list(self.L.values()), self.store.w.device)
        self._ptab.run()

    # ---------------------------------------------------------------------------------------------
    def _block_fwd(self, ar: Arena, name: str, N: int, src_raw: Src, nrm_inputs, out_dims, co, ctx, save: bool = True, n1=None,
                   out_jobs=None):
        """residual_block (resunet_model.py:103-143): out = conv2(relu(IN(conv1(relu(IN(x)))))) + IN(short(x)).
        save=False (inference): the block output is allocated first and everything else the block allocates (r, the
        shortcut, the InstanceNorm scale/shift vectors) is handed back to the arena once the block's kernels are queued.
        n1 / out_jobs (ops.FIN_TAIL): the state of the block's first norm, already filled by the launches that produced its input, and
        the consumer entries (Norm.job) of the norms that read the block's OUTPUT -- finalised by the block's last convolution."""
        L, Nn = self.L, self.Nn
        out = mk = None
        if not save:
            out = Act(ar, N, out_dims, co, dtype=self.dtype)
            mk = ar.mark()
        tail = n1 is not None
        if not tail:
            n1 = Nn[name + '.cb1'].finalize(ar, *nrm_inputs)
        s1 = Src(src_raw.x0, (N,) + tuple(L[name + '.cb1'].in_dims), src_raw.c0, src_raw.x1, src_raw.c1, src_raw.shift0,
                 scale=n1['scale'], shift=n1['shift'], act=ACT_RELU)
        r = Act(ar, N, out_dims, co, dtype=self.dtype)
        sc = Act(ar, N, out_dims, co, dtype=self.dtype)
        # the shortcut branch (1x1x1 convolution of the raw block input + its IN finalisation) does not depend on conv1: on the
        # lane's side stream (idle in the forward pass) it leaves the dependent chain, which on the deep levels is all latency
        # (28.45 -> 28.30 ms per step, inference 49.6 -> 48.4 ms per volume).  The same for the shortcut's IN backward on an
        # auxiliary stream per lane was measured slower (29.3 ms): the backward already runs four streams.
        if tail:
            ns, n2 = Nn[name + '.short'].state(ar, N), Nn[name + '.cb2'].state(ar, N)
            f_s = ops.fin_desc(ar, sc.count, [Nn[name + '.short'].job(ns)])
            f_1 = ops.fin_desc(ar, r.count, [Nn[name + '.cb2'].job(n2)])
        fork = ops.fork_side()
        with fork:
            L[name + '.short'].forward(src_raw, sc.data, sums=sc.sums, fin=f_s if tail else None)
            if not tail:
                ns = Nn[name + '.short'].finalize(ar, sc)
        L[name + '.cb1'].forward(s1, r.data, sums=r.sums, fin=f_1 if tail else None)
        if not tail:
            n2 = Nn[name + '.cb2'].finalize(ar, r)
        fork.join()
        s2 = Src(r.data, (N,) + tuple(out_dims), co, scale=n2['scale'], shift=n2['shift'], act=ACT_RELU)
        if out is None:
            out = Act(ar, N, out_dims, co, dtype=self.dtype)
        f_o = ops.fin_desc(ar, out.count, out_jobs) if (tail and out_jobs) else None
        L[name + '.cb2'].forward(s2, out.data, sums=out.sums, res=sc.data, res_scale=ns['scale'], res_shift=ns['shift'], fin=f_o)
        if save:
            ctx[name] = dict(n1=n1, s1=s1, r=r, sc=sc, ns=ns, n2=n2, s2=s2, out=out, src_raw=src_raw)
        else:
            ar.release(mk)                      # stream-ordered: later allocations are written by later kernels
        return out

    def forward(self, ar: Arena, x: torch.Tensor, y: torch.Tensor, save: bool = True) -> dict:
        """x: fp32 [N,D,H,W,1]; y: fp32 [N,D,H,W,1] output buffer (tanh).  Returns the context for backward
        (save=True) or a stub (save=False: forward-only, block temporaries are recycled -- sliding-window inference)."""
        return run_to_end(self.forward_iter(ar, x, y, save))

    def forward_iter(self, ar: Arena, x: torch.Tensor, y: torch.Tensor, save: bool = True):
        """forward() as a generator that yields between blocks: the host enqueues ~60 launches per application, and two
        applications on two stream lanes enqueued one after the other leave the second lane idle for the first one's whole enqueue
        time; the engine steps two of these alternately (vangan.interleave)."""
        N = x.shape[0]
        f, lv, L, Nn = GEN_F, self.lv, self.L, self.Nn
        ctx = {'N': N, 'x': x, 'y': y}
        # ops.FIN_TAIL: every InstanceNorm's scale / shift / mean / rstd is written by the last workgroup of the launch that produces
        # the norm's input (vg_fin_desc) -- 30 vg_in_finalize launches per application leave the lane's dependent chain.  A block's
        # FIRST norm reads the previous block's output (a decoder block's: [upsampled low-resolution output; encoder skip], two
        # producers, two channel ranges of one array), so those states exist before the first launch.
        tail = ops.FIN_TAIL
        pre, jobs = {}, {}
        if tail:
            pre['stem.cb'], pre['stem.short'] = Nn['stem.cb'].state(ar, N), Nn['stem.short'].state(ar, N)
            for b in ['enc%d' % e for e in range(1, 5)] + ['dec%d' % d for d in (3, 2, 1, 0)]:
                pre[b] = Nn[b + '.cb1'].state(ar, N)
            pre['bridge.cb1'], pre['bridge.cb2'] = Nn['bridge.cb1'].state(ar, N), Nn['bridge.cb2'].state(ar, N)
            # consumers of every block output: skips[d] (stem, enc1..enc3) feeds enc(d+1).cb1 and, behind the f[d+1] upsampled
            # channels, dec(d).cb1; enc4 feeds the bridge; bridge / dec outputs feed the next decoder block's low half
            jobs['stem'] = [Nn['enc1.cb1'].job(pre['enc1']), Nn['dec0.cb1'].job(pre['dec0'], f[1])]
            for e in range(1, 4):
                jobs['enc%d' % e] = [Nn['enc%d.cb1' % (e + 1)].job(pre['enc%d' % (e + 1)]), Nn['dec%d.cb1' % e].job(pre['dec%d' % e], f[e + 1])]
            jobs['enc4'] = [Nn['bridge.cb1'].job(pre['bridge.cb1'])]
            jobs['bridge.cb1'] = [Nn['bridge.cb2'].job(pre['bridge.cb2'])]
            jobs['bridge.cb2'] = [Nn['dec3.cb1'].job(pre['dec3'])]
            for d in (3, 2, 1):
                jobs['dec%d' % d] = [Nn['dec%d.cb1' % (d - 1)].job(pre['dec%d' % (d - 1)])]
            jobs['dec0'] = None
        sx = Src(x, (N,) + lv[0], 1, f32=True)
        c1 = Act(ar, N, lv[0], f[0], dtype=self.dtype)
        L['stem.conv1'].forward(sx, c1.data, sums=c1.sums, fin=ops.fin_desc(ar, c1.count, [Nn['stem.cb'].job(pre['stem.cb'])]) if tail else None)
        fused = _STEM_FUSED and _STEM_AUX
        if fused:
            sc = None
            ns = dict(scale=ar.alloc((N, f[0]), torch.float32), shift=ar.alloc((N, f[0]), torch.float32), mean=None, rstd=None, mult=None)
            ops.stem_short_fwd(ar, x, N, f[0], L['stem.short'].w, Nn['stem.short'].gamma, Nn['stem.short'].beta, ns['scale'], ns['shift'],
                               round16=self.dtype != torch.float32)
        else:
            sc = Act(ar, N, lv[0], f[0], dtype=self.dtype)
            L['stem.short'].forward(sx, sc.data, sums=sc.sums, fin=ops.fin_desc(ar, sc.count, [Nn['stem.short'].job(pre['stem.short'])]) if tail else None)
            ns = pre['stem.short'] if tail else Nn['stem.short'].finalize(ar, sc)
        n1 = pre['stem.cb'] if tail else Nn['stem.cb'].finalize(ar, c1)
        s1 = Src(c1.data, (N,) + lv[0], f[0], scale=n1['scale'], shift=n1['shift'], act=ACT_RELU)
        h = Act(ar, N, lv[0], f[0], dtype=self.dtype)
        L['stem.cb'].forward(s1, h.data, sums=h.sums, res=x if fused else sc.data, res_scale=ns['scale'], res_shift=ns['shift'],
                             fin=ops.fin_desc(ar, h.count, jobs['stem']) if tail else None, res_c1=fused)
        ctx['stem'] = dict(sx=sx, c1=c1, sc=sc, ns=ns, n1=n1, s1=s1, out=h)
        yield
        skips = [h]
        for e in range(1, 5):
            raw = Src(h.data, (N,) + lv[e - 1], f[e - 1])
            b = 'enc%d' % e
            h = self._block_fwd(ar, b, N, raw, (h,), lv[e], f[e], ctx, save, n1=pre.get(b), out_jobs=jobs.get(b))
            if save:
                ctx['enc%d' % e]['inp'] = (skips[-1],)
            skips.append(h)
            yield
        nb1 = pre['bridge.cb1'] if tail else Nn['bridge.cb1'].finalize(ar, h)
        sb1 = Src(h.data, (N,) + lv[4], f[4], scale=nb1['scale'], shift=nb1['shift'], act=ACT_RELU)
        b1 = Act(ar, N, lv[4], f[4], dtype=self.dtype)
        L['bridge.cb1'].forward(sb1, b1.data, sums=b1.sums, fin=ops.fin_desc(ar, b1.count, jobs['bridge.cb1']) if tail else None)
        nb2 = pre['bridge.cb2'] if tail else Nn['bridge.cb2'].finalize(ar, b1)
        sb2 = Src(b1.data, (N,) + lv[4], f[4], scale=nb2['scale'], shift=nb2['shift'], act=ACT_RELU)
        b2 = Act(ar, N, lv[4], f[4], dtype=self.dtype)
        L['bridge.cb2'].forward(sb2, b2.data, sums=b2.sums, fin=ops.fin_desc(ar, b2.count, jobs['bridge.cb2']) if tail else None)
        ctx['bridge'] = dict(inp=h, nb1=nb1, sb1=sb1, b1=b1, nb2=nb2, sb2=sb2, b2=b2)
        h = b2
        for d in (3, 2, 1, 0):
            skip = skips[d]
            raw = Src(h.data, (N,) + lv[d], h.C, skip.data, skip.C, shift0=1)       # virtual upsample + concat
            low = h
            yield
            b = 'dec%d' % d
            h = self._block_fwd(ar, b, N, raw, (low, skip), lv[d], f[d], ctx, save, n1=pre.get(b), out_jobs=jobs.get(b))
            if save:
                ctx['dec%d' % d]['inp'] = (low, skip)
        so = Src(h.data, (N,) + lv[0], f[0])
        L['out'].forward(so, y, tanh=True)
        ctx['out'] = dict(so=so, inp=h)
        return ctx

    # ---------------------------------------------------------------------------------------------
    def _norm_desc(self, ar, g, g_padded, src: Src, st, norm: Norm, dx, act, dx_cstride=0, accumulate=True):
        """Descriptor of the (IN -> act) backward of the operand described by `src` with statistics `st` (ops.actnorm_desc)."""
        N = src.N
        red = ops.alloc_red(ar, N, src.C)
        return ops.actnorm_desc(g, g_padded, src.x0, (N, src.D, src.H, src.W), src.C, dx, scale=st['scale'], shift=st['shift'],
                                act=act, norm=True, gamma=norm.gamma, mean=st['mean'], rstd=st['rstd'], red=red,
                                accumulate=accumulate, x1=src.x1, c_x0=src.c0 if src.x1 is not None else 0, x0_shift=src.shift0,
                                dx_cstride=dx_cstride, dgamma=norm.dgamma, dbeta=norm.dbeta)

    def _norm_bwd(self, ar, g, g_padded, src: Src, st, norm: Norm, dx, act, dx_cstride=0, accumulate=True):
        ops.actnorm_run(self._norm_desc(ar, g, g_padded, src, st, norm, dx, act, dx_cstride, accumulate))

    def _dgrad_norm_bwd(self, ar, lay, dy, N, dp, src: Src, st, norm: Norm, dx, act, accumulate):
        """Data gradient of `lay` into the padded grid dp, then the (IN -> act) backward of its input operand `src` into dx; the
        statistics pass rides on the data-gradient launch where the kernel carries it (ConvLayer.dgrad(bstat=...))."""
        dsc = self._norm_desc(ar, dp, True, src, st, norm, dx, act, accumulate=accumulate)
        ops.actnorm_run(dsc, stats_done=lay.dgrad(dy, N, dp, accumulate=False, bstat=dsc))

    def _block_bwd(self, ar: Arena, name: str, c: dict, N: int, inline: bool = False):
        """Backward of one residual block given the complete gradient of its output in c['out'].grad.
        inline: the block's weight gradients run on the lane itself (ConvLayer.wgrad)."""
        L, Nn = self.L, self.Nn
        out, r, sc = c['out'], c['r'], c['sc']
        d_out = out.grad
        cb1, cb2, short = L[name + '.cb1'], L[name + '.cb2'], L[name + '.short']
        mk = ar.mark()
        # shortcut InstanceNorm (no activation): d_sc.  Its statistics pass goes first (it needs only d_out); its apply pass shares a
        # launch with the apply pass of the norm in front of conv2 (ops.actnorm_apply2: nothing orders the two outputs, and both sets
        # of statistics are complete once conv2's data gradient has run) -- one launch fewer on the lane's chain per block
        d_sc = ar.alloc(sc.data.shape, self.dtype)
        ssc = Src(sc.data, (N,) + sc.dims, sc.C)
        dsc1 = self._norm_desc(ar, d_out, False, ssc, c['ns'], Nn[name + '.short'], d_sc, ACT_NONE, accumulate=False)
        ops.actnorm_stats(dsc1)
        # conv2: weights + data gradient on the padded grid, folded through relu(IN(r))
        cb2.wgrad(c['s2'], d_out, inline)
        dp = ar.alloc((N,) + cb2.buf_dims + (r.C,), self.dtype)
        d_r = ar.alloc(r.data.shape, self.dtype)
        dsc2 = self._norm_desc(ar, dp, True, c['s2'], c['n2'], Nn[name + '.cb2'], d_r, ACT_RELU, accumulate=False)
        if not cb2.dgrad(d_out, N, dp, accumulate=False, bstat=dsc2):
            ops.actnorm_stats(dsc2)
        ops.actnorm_apply2(dsc1, dsc2)
        # conv1 and shortcut conv read the block input (possibly the virtual concat)
        s1, raw = c['s1'], c['src_raw']
        cb1.wgrad(s1, d_r, inline)
        short.wgrad(raw, d_sc, inline)
        dp1 = ar.alloc((N,) + cb1.buf_dims + (s1.C,), self.dtype)
        inp = c['inp']
        if len(inp) == 1:            # encoder block: accumulate into the input's gradient
            gin = inp[0].alloc_grad(ar) if inp[0].grad is None else inp[0].grad
            assert inp[0].grad_init, 'encoder input gradient must have had its first (overwriting) writer: the decoder skip path'
            self._dgrad_norm_bwd(ar, cb1, d_r, N, dp1, s1, c['n1'], Nn[name + '.cb1'], gin, ACT_RELU, accumulate=True)
            short.dgrad(d_sc, N, gin, accumulate=True)
        else:                        # decoder block: gradient of the virtual concat, then split / sum-pool
            low, skip = inp
            # conv branch: data gradient of cb1 on the padded grid (+ the statistics of its input's IN backward); then ONE launch adds
            # the shortcut's data gradient, applies the IN backward and splits / sum-pools into the gradients of the two concat sources
            # (ops.ConvLayer.dgrad_concat_norm).  Where that launch does not serve the shape: apply pass into a concat-gradient
            # buffer, then the shortcut's data gradient fused with the concat backward (or, failing that too, three launches).
            dsc = self._norm_desc(ar, dp1, True, s1, c['n1'], Nn[name + '.cb1'], None, ACT_RELU, accumulate=False)
            stats_done = cb1.dgrad(d_r, N, dp1, accumulate=False, bstat=dsc)
            if not stats_done:
                ops.actnorm_stats(dsc)
            a_low, a_skip = not low.first_write(), not skip.first_write()
            if not short.dgrad_concat_norm(d_sc, N, dsc, low.C, low.grad, skip.grad, acc_low=a_low, acc_skip=a_skip):
                dcat = ar.alloc((N,) + tuple(cb1.in_dims) + (s1.C,), self.dtype)
                ops.actnorm_set_dx(dsc, dcat)
                ops.actnorm_run(dsc, stats_done=True)
                short.dgrad_concat(d_sc, N, dcat, low.C, low.grad, skip.grad, acc_low=a_low, acc_skip=a_skip)
        ar.release(mk, defer=True)

    def backward(self, ar: Arena, ctx: dict, gy: torch.Tensor, inline_from: int = -1):
        run_to_end(self.backward_iter(ar, ctx, gy, inline_from))

    def grad_suffix_offset(self, first_param: str = 'enc4.cb1.in.gamma') -> int:
        """Offset of `first_param` in the flat parameter / gradient buffer: everything from there to the end (enc4, bridge, decoder,
        output head: 34 of the 38 MB) is complete once the backward sweep has finished block enc4."""
        return self.store.offsets[first_param][0]

    def backward_iter(self, ar: Arena, ctx: dict, gy: torch.Tensor, inline_from: int = -1, on_suffix_done=None):
        """gy: fp32 [N,D,H,W,1] gradient w.r.t. the tanh output.  Adds parameter gradients into store.g.
        inline_from (the LAST sweep of a lane): the weight gradients of encoder blocks <= inline_from and of the stem are launched on
        the lane itself instead of its side stream -- at the end of a step the side stream is a couple of milliseconds behind the
        lanes, which have nothing left to do."""
        N = ctx['N']
        L, Nn = self.L, self.Nn
        # gradient buffers of every tensor that has more than one consumer / is read across blocks
        acts = [ctx['stem']['out']] + [ctx['enc%d' % e]['out'] for e in range(1, 5)] + \
               [ctx['bridge']['b2']] + [ctx['dec%d' % d]['out'] for d in (3, 2, 1, 0)]
        for a in acts:
            a.grad = None
            a.alloc_grad(ar)
        # output conv + tanh
        dpre = ar.alloc(gy.shape, torch.float32)
        if isinstance(ctx['y'], list):               # paired context: the two applications' outputs are separate volumes
            nb = N // len(ctx['y'])
            for i, yi in enumerate(ctx['y']):
                ops.tanh_bwd(gy[i * nb:(i + 1) * nb], yi, dpre[i * nb:(i + 1) * nb])
        else:
            ops.tanh_bwd(gy, ctx['y'], dpre)
        h = ctx['out']['inp']
        L['out'].wgrad(ctx['out']['so'], dpre)
        L['out'].dgrad(dpre, N, h.grad, accumulate=not h.first_write())
        for d in (0, 1, 2, 3):
            self._block_bwd(ar, 'dec%d' % d, ctx['dec%d' % d], N)
            yield
        # bridge
        b = ctx['bridge']
        mk = ar.mark()
        cb2, cb1 = L['bridge.cb2'], L['bridge.cb1']
        cb2.wgrad(b['sb2'], b['b2'].grad)
        dp = ar.alloc((N,) + cb2.buf_dims + (b['b1'].C,), self.dtype)
        d_b1 = ar.alloc(b['b1'].data.shape, self.dtype)
        self._dgrad_norm_bwd(ar, cb2, b['b2'].grad, N, dp, b['sb2'], b['nb2'], Nn['bridge.cb2'], d_b1, ACT_RELU, accumulate=False)
        cb1.wgrad(b['sb1'], d_b1)
        dp = ar.alloc((N,) + cb1.buf_dims + (b['inp'].C,), self.dtype)
        self._dgrad_norm_bwd(ar, cb1, d_b1, N, dp, b['sb1'], b['nb1'], Nn['bridge.cb1'], b['inp'].grad, ACT_RELU, accumulate=not b['inp'].first_write())
        ar.release(mk, defer=True)
        for e in (4, 3, 2, 1):
            yield
            self._block_bwd(ar, 'enc%d' % e, ctx['enc%d' % e], N, inline=e <= inline_from)   # the sweep's last weight gradients on the lane itself
            if e == 4 and on_suffix_done is not None:
                on_suffix_done()                 # gradients of enc4 ... output head are complete (data parallel: reduce them now)
        yield
        inl = inline_from >= 0
        # stem
        s = ctx['stem']
        d_out = s['out'].grad
        stem_aux = _STEM_AUX or s['sc'] is None             # (a fused forward has no stored shortcut tensor to run the explicit path on)
        ssc = None if s['sc'] is None else Src(s['sc'].data, (N,) + s['sc'].dims, s['sc'].C)
        if stem_aux:
            # the shortcut reads the single-channel volume and nobody needs its data gradient; its output normalises to
            # w*rs*(x - mean x), so the loss sees w only through eps: dL/dw = eps*gamma*rs^3 * sum d_out*(x - mean x), dL/db = 0 -- a closed
            # form in two moments of d_out against the volume itself (vg_stem_short_bwd).  No apply pass, no gradient tensor, no
            # weight-gradient launch, no read of the stored shortcut tensor: two full-resolution 16-channel passes per sweep less, and a
            # well-conditioned, deterministic number where the explicit path sums a million cancelling terms.
            nrm = Nn['stem.short']
            ops.stem_short_bwd(ar, d_out, s['sx'].x0, N, GEN_F[0], L['stem.short'].w, nrm.gamma, L['stem.short'].gw,
                               dgamma=nrm.dgamma, dbeta=nrm.dbeta, round16=self.dtype != torch.float32)
        else:
            d_sc = ar.alloc(s['sc'].data.shape, self.dtype)
            self._norm_bwd(ar, d_out, False, ssc, s['ns'], Nn['stem.short'], d_sc, ACT_NONE, accumulate=False)
        cb = L['stem.cb']
        cb.wgrad(s['s1'], d_out, inl)
        dp = ar.alloc((N,) + cb.buf_dims + (s['c1'].C,), self.dtype)
        d_c1 = ar.alloc(s['c1'].data.shape, self.dtype)
        self._dgrad_norm_bwd(ar, cb, d_out, N, dp, s['s1'], s['n1'], Nn['stem.cb'], d_c1, ACT_RELU, accumulate=False)
        L['stem.conv1'].wgrad(s['sx'], d_c1, inl)
        if not stem_aux:
            L['stem.short'].wgrad(s['sx'], d_sc, inl)


# ======================================================================================================
# (f)4: the ResNet generator (generator.py:7-73) -- forward only (inference / test_step-style use)
# ======================================================================================================
RESNET_F, RESNET_DOWN, RESNET_RES, RESNET_UP = 32, 3, 6, 3


def resnet_param_specs() -> List[Tuple[str, Tuple[int, ...], str]]:
    """Names / order / layouts of oracle.vangan_oracle.resnet_param_specs (weights exchange 1:1).  Every InstanceNorm of this
    generator is created with gamma_initializer='he_normal' (generator.py:14,40,49,55,62; building_blocks.py:107,121,190,277)."""
    s: List[Tuple[str, Tuple[int, ...], str]] = []

    def inorm(name, c):
        s.append((name + '.gamma', (c,), 'he_vec')); s.append((name + '.beta', (c,), 'zeros'))

    f = RESNET_F
    s.append(('c7.w', (7, 7, 7, 1, f), 'he_normal')); inorm('c7.in', f)
    for i in range(RESNET_DOWN):
        s.append(('down%d.w' % i, (3, 3, 3, f, 2 * f), 'he_normal')); inorm('down%d.in' % i, 2 * f)
        f *= 2
    for j in range(RESNET_RES):
        for c in ('c1', 'c2'):
            s.append(('res%d.%s.w' % (j, c), (3, 3, 3, f, f), 'he_normal')); inorm('res%d.%s.in' % (j, c), f)
    for i in range(RESNET_UP):
        s.append(('up%d.w' % i, (4, 4, 4, f, f // 2), 'he_normal')); inorm('up%d.in' % i, f // 2)
        f //= 2
    s.append(('out.w', (7, 7, 7, f, 1), 'glorot_uniform')); s.append(('out.b', (1,), 'zeros'))
    return s


class ResNetGenerator:
    """get_resnet_generator (generator.py:7-73) as vangan.py:88-97,127-134 configures it (filters 32, three stride-2 stages, six
    residual blocks, three UpSampling3D + 4^3 'same' stages, 7^3 head with tanh): the non-default generator of SURVEY 8(f)4, forward
    and backward (parameter gradients).  Built from the same kernels as the ResUNet:
      * the 7^3 stem reads the single-channel volume W-packed (49 (d, h) taps x 7 pseudo-channels: the path of D.conv0);
      * every InstanceNorm + ReLU (+ SpatialDropout3D multipliers in training mode) is applied on read by the next convolution;
      * UpSampling3D is virtual (the gather reads the low-resolution tensor at idx >> 1), the 4^3 'same' convolution is D.down2's;
      * the residual Add -- input + InstanceNorm(conv2), whose statistics exist only after conv2 has finished -- is vg_affine_add;
      * the 7^3 head (343 taps; a launch takes 64) is a chain of seven 49-tap chunks accumulating into the fp32 output, tanh in the
        last one (vg_conv_desc::tanh_out with accumulate).
    Spatial sizes: n -> n - 4 -> three times floor((m - 1) / 2) + 1 -> x 8; the output has the input's size when n is a multiple of 16."""

    def __init__(self, store: ParamStore, dims: Tuple[int, int, int], dtype: torch.dtype = torch.bfloat16):
        if any(n % 16 or n < 32 for n in dims):
            raise ValueError('spatial dims must be multiples of 16 and >= 32')
        self.dtype, self.store, self.dims = dtype, store, tuple(dims)
        L, Nn = {}, {}
        f = RESNET_F
        d = tuple(dims)
        L['c7'] = ConvLayer(store, 'c7', 7, 1, f, 1, 'reflect', False, d, need_dgrad=False, dtype=dtype)
        Nn['c7'] = Norm(store, 'c7.in', f)
        d = L['c7'].out_dims
        for i in range(RESNET_DOWN):
            k = 'down%d' % i
            L[k] = ConvLayer(store, k, 3, f, 2 * f, 2, 'reflect', False, d, need_dgrad=True, dtype=dtype)
            Nn[k] = Norm(store, k + '.in', 2 * f)
            d, f = L[k].out_dims, 2 * f
        for j in range(RESNET_RES):
            for c in ('c1', 'c2'):
                k = 'res%d.%s' % (j, c)
                L[k] = ConvLayer(store, k, 3, f, f, 1, 'reflect', False, d, need_dgrad=True, dtype=dtype)
                Nn[k] = Norm(store, k + '.in', f)
        for i in range(RESNET_UP):
            k = 'up%d' % i
            d = tuple(2 * n for n in d)
            L[k] = ConvLayer(store, k, 4, f, f // 2, 1, 'same', False, d, need_dgrad=True, dtype=dtype)
            Nn[k] = Norm(store, k + '.in', f // 2)
            f //= 2
        assert d == tuple(dims), (d, dims)
        # the head: one chunk per kernel depth slice (49 taps); only the first adds the bias
        self.head = [ConvLayer(store, 'out', 7, f, 1, 1, 'same', a == 0, d, need_dgrad=True, dtype=dtype,
                               tap_subset=list(range(a * 49, (a + 1) * 49))) for a in range(7)]
        self.L, self.Nn = L, Nn

    def pack(self):
        if getattr(self, '_ptab', None) is None:
            self._ptab = ops.PackTable(list(self.L.values()) + self.head, self.store.w.device)
        self._ptab.run()

    def forward(self, ar: Arena, x: torch.Tensor, y: torch.Tensor, drop: Optional[dict] = None, save: bool = True) -> dict:
        """x: fp32 [N,D,H,W,1]; y: fp32 [N,D,H,W,1] output (tanh).  drop: SpatialDropout3D multipliers [N,C] for 'c7' and
        'down0..2' (training=True behaviour), None = inference.  Returns the stored tensors (taps) for the parity tests."""
        drop = drop or {}
        L, Nn = self.L, self.Nn
        N = x.shape[0]
        taps = {}
        ctx = {'N': N, 'y': y, 'chain': [], 'res': [], 'ups': []}      # what backward() needs: every convolution's source, output, statistics
        h = Act(ar, N, L['c7'].out_dims, L['c7'].cout, dtype=self.dtype)
        src = Src(x, (N,) + self.dims, 1, f32=True)
        L['c7'].forward(src, h.data, sums=h.sums)
        taps['c7'] = h
        key = 'c7'
        ctx['c7'] = src
        for i in range(RESNET_DOWN):
            k = 'down%d' % i
            st = Nn[key].finalize(ar, h, mult=drop.get(key))
            src = Src(h.data, (N,) + h.dims, h.C, scale=st['scale'], shift=st['shift'], act=ACT_RELU)
            a = Act(ar, N, L[k].out_dims, L[k].cout, dtype=self.dtype)
            L[k].forward(src, a.data, sums=a.sums)
            taps[k] = a
            ctx['chain'].append((k, src, h, st, key))                   # (conv, its source, the source's raw tensor, its statistics, its norm)
            h, key = a, k
        st = Nn[key].finalize(ar, h, mult=drop.get(key))
        ctx['trunk_in'] = (h, st, key)
        cur, sc, sf, act = h.data, st['scale'], st['shift'], ACT_RELU          # the block input with its pending on-read transform
        dims, C_ = h.dims, h.C
        S = dims[0] * dims[1] * dims[2]
        for j in range(RESNET_RES):
            k = 'res%d' % j
            r1 = Act(ar, N, dims, C_, dtype=self.dtype)
            s1 = Src(cur, (N,) + dims, C_, scale=sc, shift=sf, act=act)
            L[k + '.c1'].forward(s1, r1.data, sums=r1.sums)
            n1 = Nn[k + '.c1'].finalize(ar, r1)
            r2 = Act(ar, N, dims, C_, dtype=self.dtype)
            s2 = Src(r1.data, (N,) + dims, C_, scale=n1['scale'], shift=n1['shift'], act=ACT_RELU)
            L[k + '.c2'].forward(s2, r2.data, sums=r2.sums)
            n2 = Nn[k + '.c2'].finalize(ar, r2)
            out = ar.alloc((N,) + dims + (C_,), self.dtype)
            ops.affine_add(cur, sc, sf, act, r2.data, n2['scale'], n2['shift'], N, S, C_, out)
            taps[k] = out
            ctx['res'].append((k, s1, r1, n1, s2, r2, n2))
            cur, sc, sf, act = out, None, None, ACT_NONE
        for i in range(RESNET_UP):
            k = 'up%d' % i
            dims = tuple(2 * n for n in dims)
            a = Act(ar, N, dims, L[k].cout, dtype=self.dtype)
            su = Src(cur, (N,) + dims, C_, shift0=1, scale=sc, shift=sf, act=act)
            L[k].forward(su, a.data, sums=a.sums)
            taps[k] = a
            st = Nn[k].finalize(ar, a)
            ctx['ups'].append((k, su, a, st))
            cur, sc, sf, act, C_ = a.data, st['scale'], st['shift'], ACT_RELU, L[k].cout
        src = Src(cur, (N,) + dims, C_, scale=sc, shift=sf, act=act)
        for a_, lay in enumerate(self.head):
            lay.forward(src, y, tanh=(a_ == len(self.head) - 1), accumulate=(a_ > 0))
        ctx['head'] = src
        taps['_ctx'] = ctx
        return taps

    DROP_RATES = {'c7': 0.5, 'down0': 0.2, 'down1': 0.2, 'down2': 0.2}      # generator.py:44, building_blocks.py downsample()
    DROP_CH = {'c7': RESNET_F, 'down0': 2 * RESNET_F, 'down1': 4 * RESNET_F, 'down2': 8 * RESNET_F}

    def forward_iter(self, ar: Arena, x: torch.Tensor, y: torch.Tensor, drop: Optional[dict] = None):
        """The engine's resumable-enqueue protocol (ResUNet.forward_iter): this network is enqueued in one piece."""
        return self.forward(ar, x, y, drop() if callable(drop) else drop)
        yield

    def _in_bwd(self, ar: Arena, g, g_padded: bool, raw: 'Act', st: dict, norm: 'Norm', act: int):
        """(InstanceNorm -> act -> dropout) backward of the tensor raw.data given the gradient g of its transformed value (on the
        reflect-padded grid when g_padded): returns the gradient of raw.data; gamma / beta gradients are added."""
        N = raw.N
        dx = ar.alloc((N,) + raw.dims + (raw.C,), self.dtype)
        red = ops.alloc_red(ar, N, raw.C)
        ops.actnorm_bwd(g, g_padded, raw.data, (N,) + raw.dims, raw.C, dx, scale=st['scale'], shift=st['shift'], mult=st.get('mult'),
                        act=act, norm=True, gamma=norm.gamma, mean=st['mean'], rstd=st['rstd'], red=red, accumulate=False,
                        dgamma=norm.dgamma, dbeta=norm.dbeta)
        return dx

    def backward(self, ar: Arena, taps: dict, gy: torch.Tensor, inline_from: int = -1):
        """gy: fp32 [N,D,H,W,1], gradient w.r.t. the tanh output of forward() (whose return value is `taps`).  Adds the parameter
        gradients of every layer into store.g (generator.py:7-73 under tf.GradientTape, vangan.py:426-438).  The generator's input
        gradient is not needed: each generator's loss reaches only its own applications (vangan.py:321-353).
        Every step is an existing launch: tanh backward; per convolution the weight gradient from its on-read source, the data gradient
        (all 49-tap chunks of the head accumulate into one buffer), the transpose of the reflection pad folded into the IN backward of
        the source; UpSampling3D's backward is the 2x2x2 sum-pool of vg_concat_bwd without a skip half; the residual Add hands its
        gradient to both operands -- the convolution branch's IN backward reads it, the identity branch accumulates the folded data
        gradient of the block's first convolution into it."""
        ctx = taps['_ctx']
        L, Nn, N = self.L, self.Nn, ctx['N']
        mk = ar.mark()
        dpre = ar.alloc(gy.shape, torch.float32)
        ops.tanh_bwd(gy, ctx['y'], dpre)
        # ---- head: seven 49-tap chunks over one source
        src = ctx['head']
        k_up, su, a_up, st_up = ctx['ups'][-1]
        dp = ar.alloc((N,) + tuple(self.head[0].buf_dims) + (a_up.C,), self.dtype)
        for i, lay in enumerate(self.head):
            lay.wgrad(src, dpre)
            lay.dgrad(dpre, N, dp, accumulate=(i > 0))
        g = self._in_bwd(ar, dp, False, a_up, st_up, Nn[k_up], ACT_RELU)                 # gradient of up2's raw output
        # ---- up stages: 4^3 'same' over the virtual UpSampling3D of the previous tensor
        for i in range(RESNET_UP - 1, -1, -1):
            k, su, a, st = ctx['ups'][i]
            lay = L[k]
            lay.wgrad(su, g)
            fine = tuple(lay.in_dims)
            dpf = ar.alloc((N,) + fine + (lay.cin,), self.dtype)
            lay.dgrad(g, N, dpf, accumulate=False)
            low = tuple(n // 2 for n in fine)
            gl = ar.alloc((N,) + low + (lay.cin,), self.dtype)
            ops.concat_bwd(dpf, (N,) + fine, lay.cin, 0, gl, None, acc_low=False, acc_skip=False)
            if i > 0:
                kp, _, ap, stp = ctx['ups'][i - 1]
                g = self._in_bwd(ar, gl, False, ap, stp, Nn[kp], ACT_RELU)
            else:
                g = gl                                                                  # gradient of the trunk's output (no pending transform)
        # ---- residual blocks
        h_in, st_in, key_in = ctx['trunk_in']
        for j in range(RESNET_RES - 1, -1, -1):
            k, s1, r1, n1, s2, r2, n2 = ctx['res'][j]
            c1, c2 = L[k + '.c1'], L[k + '.c2']
            d_r2 = self._in_bwd(ar, g, False, r2, n2, Nn[k + '.c2'], ACT_NONE)
            c2.wgrad(s2, d_r2)
            dp2 = ar.alloc((N,) + tuple(c2.buf_dims) + (c2.cin,), self.dtype)
            c2.dgrad(d_r2, N, dp2, accumulate=False)
            d_r1 = self._in_bwd(ar, dp2, True, r1, n1, Nn[k + '.c1'], ACT_RELU)
            c1.wgrad(s1, d_r1)
            dp1 = ar.alloc((N,) + tuple(c1.buf_dims) + (c1.cin,), self.dtype)
            c1.dgrad(d_r1, N, dp1, accumulate=False)
            # identity branch: g (the Add's gradient) + the folded data gradient of c1 = gradient of the block input's transformed value
            ops.actnorm_bwd(dp1, True, None, (N,) + tuple(c1.in_dims), c1.cin, g, act=ACT_NONE, norm=False, accumulate=True)
        g = self._in_bwd(ar, g, False, h_in, st_in, Nn[key_in], ACT_RELU)                # through down2's IN + ReLU + dropout
        # ---- stride-2 stages, then the stem (weights only)
        for (k, src, h_prev, st_prev, key_prev) in reversed(ctx['chain']):
            lay = L[k]
            lay.wgrad(src, g)
            dp = ar.alloc((N,) + tuple(lay.buf_dims) + (lay.cin,), self.dtype)
            lay.dgrad(g, N, dp, accumulate=False)
            g = self._in_bwd(ar, dp, True, h_prev, st_prev, Nn[key_prev], ACT_RELU)
        L['c7'].wgrad(ctx['c7'], g)
        ar.release(mk, defer=True)


# ======================================================================================================
# Discriminator
# ======================================================================================================
class PatchGAN:
    """get_discriminator (discriminator.py:7-124): reflect-pad -> noise -> Conv(64,k4,s2,bias) -> IN -> LReLU ->
    2 x downsample(k4,s2,'valid' after reflect pad) -> downsample(k4,s1,'same') -> noise -> Conv(1,k3,'same')."""

    NAMES = ['conv0', 'down0', 'down1', 'down2', 'out']

    def __init__(self, store: ParamStore, dims: Tuple[int, int, int], dtype: torch.dtype = torch.bfloat16):
        self.dtype = dtype
        self.store, self.dims = store, tuple(dims)
        lv = [tuple(n >> i for n in dims) for i in range(4)]
        self.lv = lv
        self.ch = [1, 64, 128, 256, 512]
        L = self.L = {}
        L['conv0'] = ConvLayer(store, 'conv0', 4, 1, 64, 2, 'reflect', True, lv[0], dtype=self.dtype)
        L['down0'] = ConvLayer(store, 'down0', 4, 64, 128, 2, 'reflect', False, lv[1], dtype=self.dtype)
        L['down1'] = ConvLayer(store, 'down1', 4, 128, 256, 2, 'reflect', False, lv[2], dtype=self.dtype)
        L['down2'] = ConvLayer(store, 'down2', 4, 256, 512, 1, 'same', False, lv[3], dtype=self.dtype)
        L['out'] = ConvLayer(store, 'out', 3, 512, 1, 1, 'same', True, lv[3], dtype=self.dtype)
        self.Nn = {k: Norm(store, k + '.in', c) for k, c in zip(self.NAMES[:4], self.ch[1:])}
        # wasserstein=True (discriminator.py:116-119): Flatten -> Dropout(0.2) -> Dense(1) over the patch logits; present when the store has it
        self.n_patch = lv[3][0] * lv[3][1] * lv[3][2]
        self.dense = 'dense.w' in store.offsets
        if self.dense:
            self.dw_, self.db_ = store.param('dense.w'), store.param('dense.b')
            self.gdw, self.gdb = store.grad('dense.w'), store.grad('dense.b')

    def head_forward(self, logits: torch.Tensor, mask: Optional[torch.Tensor], z: torch.Tensor):
        """z[N] = Dense(Dropout(Flatten(logits))): mask [N, n_patch] dropout multipliers (training) or None."""
        ops.dense_head_fwd(logits, mask, self.dw_, self.db_, logits.shape[0], self.n_patch, z)

    def head_backward(self, logits: torch.Tensor, mask: Optional[torch.Tensor], gz: torch.Tensor, dlogits: Optional[torch.Tensor], wgrad: bool):
        """gz[N] = d loss / d z; dlogits (same shape as logits) is overwritten; wgrad: the Dense kernel / bias gradients are added."""
        ops.dense_head_bwd(logits, mask, self.dw_, gz, logits.shape[0], self.n_patch, dx=dlogits, dw=self.gdw if wgrad else None,
                           db=self.gdb if wgrad else None)

    def pack(self):
        if getattr(self, '_ptab', None) is None:
            self._ptab = ops.PackTable(list(self.L.values()), self.store.w.device)
        self._ptab.run()

    def noise_shapes(self, N: int):
        lv = self.lv
        return {'conv0': (N,) + tuple(n + 2 for n in lv[0]) + (1,), 'down0': (N,) + tuple(n + 2 for n in lv[1]) + (64,),
                'down1': (N,) + tuple(n + 2 for n in lv[2]) + (128,), 'down2': (N,) + lv[3] + (256,),
                'out': (N,) + lv[3] + (512,)}

    def forward(self, ar: Arena, x: torch.Tensor, logits: torch.Tensor, noise: Optional[dict] = None,
                drop: Optional[dict] = None) -> dict:
        """x: fp32 [N,D,H,W,1]; logits: fp32 [N,D/8,H/8,W/8,1].  noise[k]: bf16 tensors (noise_shapes) or None;
        drop[k]: fp32 [N,C] channel multipliers for down0/1/2 or None."""
        noise, drop = noise or {}, drop or {}
        N = x.shape[0]
        lv, L, Nn = self.lv, self.L, self.Nn
        ctx = {'N': N, 'x': x}
        src = Src(x, (N,) + lv[0], 1, f32=True, noise=noise.get('conv0'), noise_pad=1)
        acts, srcs, sts = [], [src], []
        tail = ops.FIN_TAIL
        h = Act(ar, N, lv[1], 64, dtype=self.dtype)
        # (ops.FIN_TAIL: the norm behind a convolution -- with that layer's channel-dropout multipliers folded in -- is finalised by the
        # convolution's own launch: four vg_in_finalize launches per application leave the chain)
        nxt = Nn['conv0'].state(ar, N, mult=None) if tail else None
        L['conv0'].forward(src, h.data, sums=h.sums, fin=ops.fin_desc(ar, h.count, [Nn['conv0'].job(nxt)]) if tail else None)
        acts.append(h)
        prev_drop = None
        for i, k in enumerate(['down0', 'down1', 'down2', 'out']):
            st = nxt if tail else Nn[self.NAMES[i]].finalize(ar, h, mult=prev_drop)
            sts.append(st)
            lay = L[k]
            src = Src(h.data, (N,) + tuple(lay.in_dims), h.C, scale=st['scale'], shift=st['shift'], act=ACT_LRELU,
                      noise=noise.get(k), noise_pad=1 if lay.pad == 'reflect' else 0)
            srcs.append(src)
            if k == 'out':
                lay.forward(src, logits)
            else:
                h = Act(ar, N, lay.out_dims, lay.cout, dtype=self.dtype)
                prev_drop = drop.get(k)
                nxt = Nn[k].state(ar, N, mult=prev_drop) if tail else None
                lay.forward(src, h.data, sums=h.sums, fin=ops.fin_desc(ar, h.count, [Nn[k].job(nxt)]) if tail else None)
                acts.append(h)
        ctx.update(acts=acts, srcs=srcs, sts=sts)
        return ctx

    def backward_both(self, ar: Arena, ctx: dict, g3: torch.Tensor, B: int, dx: torch.Tensor):
        """The two backward sweeps of a train step (vangan.py:426-438: the critic loss over [real; fake] with parameter gradients, the
        generator loss through the fake half down to the input volume without) as ONE sweep over 3B gradient samples [d critic / d logits
        (2B); d generator loss / d logits (B)]: the data gradients need no forward tensor, the InstanceNorm backward reads the fake
        half's activations for the third group (vg_actnorm_bwd_desc::alias_n0), weight gradients take the first 2B samples, only the
        generator-loss group reaches dx (fp32 [B,D,H,W,1]).  Half the launches of the two sweeps, 1.5 x the work per launch on the
        latency-bound 16^3 / 32^3 levels."""
        L, Nn = self.L, self.Nn
        N3, N2 = 3 * B, 2 * B
        mk = ar.mark()
        g = g3
        for j, k in enumerate(['out', 'down2', 'down1', 'down0']):
            lay = L[k]
            li = 4 - j
            lay.wgrad(ctx['srcs'][li], g[:N2])
            a = ctx['acts'][li - 1]
            dp = ar.alloc((N3,) + tuple(lay.buf_dims) + (lay.cin,), self.dtype)
            st = ctx['sts'][li - 1]
            nrm = Nn[self.NAMES[li - 1]]
            red = ops.alloc_red(ar, N3, a.C)
            dxa = ar.alloc((N3,) + a.dims + (a.C,), self.dtype)
            dsc = ops.actnorm_desc(dp, lay.pad == 'reflect', a.data, (N3,) + a.dims, a.C, dxa, scale=st['scale'], shift=st['shift'],
                                   mult=st['mult'], act=ACT_LRELU, norm=True, gamma=nrm.gamma, mean=st['mean'], rstd=st['rstd'], red=red,
                                   accumulate=False, dgamma=nrm.dgamma, dbeta=nrm.dbeta, alias_n0=N2, alias_shift=B, pgrad_n=N2)
            # (the statistics pass rides in the data gradient's epilogue where the LDS-DMA family serves the layer)
            ops.actnorm_run(dsc, stats_done=lay.dgrad(g, N3, dp, accumulate=False, bstat=dsc))
            g = dxa
        lay = L['conv0']
        lay.wgrad(ctx['srcs'][0], g[:N2])
        lay.dgrad_input(ar, g[N2:], B, dx)
        ar.release(mk, defer=True)

    def backward(self, ar: Arena, ctx: dict, dlogits: torch.Tensor, n0: int, n1: int, wgrad: bool,
                 dx: Optional[torch.Tensor] = None):
        """Backward for samples [n0,n1) with upstream dlogits (fp32 [n1-n0,...,1]).  wgrad: accumulate parameter
        gradients; dx: if given (fp32 [n1-n0,D,H,W,1]) receives the gradient w.r.t. the input volume."""
        L, Nn = self.L, self.Nn
        N = n1 - n0
        mk = ar.mark()

        def sl(t):
            return None if t is None else t[n0:n1]

        def sub(src: Src) -> Src:
            return Src(sl(src.x0), (N, src.D, src.H, src.W), src.c0, f32=src.f32, scale=sl(src.scale), shift=sl(src.shift),
                       act=src.act, noise=sl(src.noise), noise_pad=src.noise_pad)

        g = dlogits
        names = ['out', 'down2', 'down1', 'down0']
        for j, k in enumerate(names):
            lay = L[k]
            li = 4 - j                     # index into srcs; acts[li-1] is the input tensor of this conv
            src = sub(ctx['srcs'][li])
            if wgrad:
                lay.wgrad(src, g)
            a = ctx['acts'][li - 1]
            dp = ar.alloc((N,) + tuple(lay.buf_dims) + (lay.cin,), self.dtype)
            lay.dgrad(g, N, dp, accumulate=False)
            st = ctx['sts'][li - 1]
            nrm = Nn[self.NAMES[li - 1]]
            red = ops.alloc_red(ar, N, a.C)
            dxa = ar.alloc((N,) + a.dims + (a.C,), self.dtype)
            ops.actnorm_bwd(dp, lay.pad == 'reflect', sl(a.data), (N,) + a.dims, a.C, dxa, scale=sl(st['scale']),
                            shift=sl(st['shift']), mult=sl(st['mult']), act=ACT_LRELU, norm=True, gamma=nrm.gamma,
                            mean=sl(st['mean']), rstd=sl(st['rstd']), red=red, accumulate=False,
                            dgamma=nrm.dgamma if wgrad else None, dbeta=nrm.dbeta if wgrad else None)
            g = dxa
        lay = L['conv0']
        src = sub(ctx['srcs'][0])
        if wgrad:
            lay.wgrad(src, g)
        if dx is not None:
            lay.dgrad_input(ar, g, N, dx)
        ar.release(mk, defer=True)
