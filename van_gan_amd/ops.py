"""Thin host-side wrappers over the C ABI (include/vangan_hip.h): every function here only marshals
torch-owned device buffers into libvangan_hip.so calls on torch's current HIP stream.  No arithmetic on
the data path is done by torch."""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, PAD_REFLECT, PAD_ZERO, ActNormBwdDesc, ConvDesc, FinDesc, check, lib

IN_EPS = 1e-3          # tfa InstanceNormalization default epsilon (resunet_model.py:36)
STRIPES = 8            # VG_STRIPES of include/vangan_hip.h


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def conv_variant(d) -> str:
    """Name of the kernel variant vg_conv3d would launch for this descriptor (host-only dry run of the dispatch)."""
    buf = C.create_string_buffer(512)
    _lib.lib.vg_conv3d_variant(C.byref(d), buf, 512)
    return buf.value.decode()


_DEV: Optional[int] = None          # device index of the engine (set_device): lets stream() take the fast path
_STREAM_OBJ = {}                    # (device index, raw handle) -> torch.cuda.Stream object of that handle


def set_device(index: int):
    """The engine's device.  torch.cuda.current_stream() costs ~8 us of Python per call (device-index plumbing) and the step
    asks for the current stream ~1300 times: with the index known, the raw handle comes from one C call instead."""
    global _DEV
    _DEV = int(index)


def stream() -> int:
    if DRY is not None:
        return 0
    if _DEV is not None:
        return torch._C._cuda_getCurrentRawStream(_DEV)
    return torch.cuda.current_stream().cuda_stream


def current_stream_obj() -> torch.cuda.Stream:
    """torch.cuda.Stream object of the current stream, cached per raw handle."""
    if _DEV is None:
        return torch.cuda.current_stream()
    key = (_DEV, torch._C._cuda_getCurrentRawStream(_DEV))
    so = _STREAM_OBJ.get(key)
    if so is None:
        so = _STREAM_OBJ[key] = torch.cuda.current_stream(_DEV)
    return so


class DryRun:
    """Context manager: every libvangan_hip.so call made through this module is replaced by a no-op, except that
    vg_conv3d / vg_conv3d_wgrad run their complete host-side dispatch WITHOUT launching and report the kernel variant they
    selected (vg_conv3d_variant / vg_conv3d_wgrad_variant).  Works on CPU tensors: the schedules of nets.py can be walked on
    a box without a GPU to list the kernels a configuration would run (tests/test_variant_coverage.py).
    records: [(kind, layer name, variant string)], kind in fwd / dgrad / wgrad."""

    _PASS = ('vg_conv3d_dma_bn', 'vg_conv3d_scratch_bytes', 'vg_conv3d_thin_np', 'vg_conv3d_plan', 'vg_packed_ktot', 'vg_packed_rows', 'vg_conv3d_lds_bytes', 'vg_status_string', 'vg_set_tuning',
             'vg_stem_short_bwd_workgroups', 'vg_stem_short_fwd_workgroups')

    def __init__(self):
        self.records = []
        self.tag = ('?', '?')
        self.recipe = None          # how to rebuild the call that is being recorded (ConvLayer fills it in)
        self.recipes = []           # parallel to records

    def __getattr__(self, name):            # stands in for `lib`
        real = _lib.lib
        if name in self._PASS:
            return getattr(real, name)
        if name == 'vg_conv3d':
            def conv(dref, _stream):
                buf = C.create_string_buffer(512)
                rc = real.vg_conv3d_variant(dref, buf, 512)
                self.records.append(self.tag + (buf.value.decode(),)); self.recipes.append(self.recipe)
                return rc
            return conv
        if name == 'vg_conv3d_wgrad':
            def wgrad(dref, _dy, dy_f32, tap_idx, T, _gw, _gb, _sc, sc_bytes, _stream):
                buf = C.create_string_buffer(512)
                rc = real.vg_conv3d_wgrad_variant(dref, dy_f32, tap_idx, T, sc_bytes, buf, 512)
                self.records.append(self.tag + (buf.value.decode(),)); self.recipes.append(self.recipe)
                return rc
            return wgrad
        return lambda *a: 0

    def variants(self):
        return sorted({(k, v) for k, _, v in self.records})

    def __enter__(self):
        global DRY, lib
        self._saved = lib
        DRY, lib = self, self
        return self

    def __exit__(self, *exc):
        global DRY, lib
        DRY, lib = None, self._saved
        return False


DRY: Optional[DryRun] = None


class Recorder:
    """Context manager: every libvangan_hip.so call made through this module is executed AND appended to a launch list
    (function, marshalled arguments -- device addresses, descriptors, the raw stream handle), together with the stream
    dependencies issued through wait_stream / wait_event / record_event below.  A train step's list is a pure function of the
    engine's static workspace, so replaying it (VanGan.train_step_replay) re-enqueues the step -- same kernels, same streams, same
    dependencies -- without running the ~8 ms of Python that builds ~900 descriptors; the per-step scalars come from the device
    parameter block (the *_dev entry points).  Unlike a HIP graph of the same step (VanGan.capture_train_step: correct, but
    hipGraphLaunch serialises most of the lanes' overlap on ROCm 7.2 and costs ~4 ms per step) the replay keeps the eager schedule."""

    _PURE = DryRun._PASS + ('vg_conv3d_variant', 'vg_conv3d_wgrad_variant', 'vg_abi_sizeof', 'vg_version', 'vg_storage16')

    def __init__(self):
        self.cmds = []           # (callable, args)
        self.keep = []           # objects whose addresses are baked into descriptors (must outlive the list)
        self._wrapped = {}

    def __getattr__(self, name):
        real = getattr(_lib.lib, name)
        if name in self._PURE:
            return real
        w = self._wrapped.get(name)
        if w is None:
            cmds = self.cmds

            def w(*a, _f=real):
                rc = _f(*a)
                cmds.append((_f, a))
                return rc
            self._wrapped[name] = w
        return w

    def __enter__(self):
        global REC, lib
        self._saved = lib
        REC, lib = self, self
        return self

    def __exit__(self, *exc):
        global REC, lib
        REC, lib = None, self._saved
        return False


REC: Optional[Recorder] = None


def _host(fn, *a):
    """A host-side stream operation: done now and, while a step is being recorded, kept for the replay."""
    fn(*a)
    if REC is not None:
        REC.cmds.append((fn, a))


def wait_stream(waiter: torch.cuda.Stream, other: torch.cuda.Stream):
    if REC is None:
        waiter.wait_stream(other)
        return
    ev = torch.cuda.Event()              # a replay re-records the SAME event object in the same program order
    _host(ev.record, other); _host(waiter.wait_event, ev)


def wait_event(waiter: torch.cuda.Stream, ev):
    _host(waiter.wait_event, ev)


def record_event(stream: torch.cuda.Stream):
    if REC is None:
        return stream.record_event()
    ev = torch.cuda.Event()
    _host(ev.record, stream)
    return ev


def zero_fill(t: torch.Tensor):
    """t.zero_() on the current stream (recorded as vg_memset_zero with that stream's handle)."""
    if REC is None:
        t.zero_()
    else:
        check(lib.vg_memset_zero(_p(t), t.numel() * t.element_size(), stream()), 'vg_memset_zero')


def copy(dst: torch.Tensor, src: torch.Tensor):
    if REC is None or not (dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()):
        assert REC is None, 'recorded copies must be contiguous and type-preserving'
        dst.copy_(src)
    else:
        check(lib.vg_copy_bytes(_p(dst), _p(src), dst.numel() * dst.element_size(), stream()), 'vg_copy_bytes')


class Fp16:
    """Context manager: every library call made through this module goes to libvangan_hip_h.so, the build whose 16-bit buffers
    hold IEEE half precision (fp16 sliding-window inference, BASELINE config 5).  Tensors handed to the calls inside must be
    torch.float16 where the bf16 path has torch.bfloat16."""

    def __enter__(self):
        global lib
        self._saved = lib
        lib = _lib.lib_fp16()
        return self

    def __exit__(self, *exc):
        global lib
        lib = self._saved
        return False


class KernelProfile:
    """Optional per-launch HIP-event timing of the MFMA kernels on the stream they are launched on (torch's current
    stream), with the algorithmic FLOPs of every launch.  Used by bench.py for the roofline object."""

    def __init__(self):
        self.rows = {}          # kind -> [launches, flops, [(ev0, ev1), ...], bytes]
        self.vrows = {}         # (kind, kernel variant) -> the same: the roofline of every kernel template by itself
        self.lrows = {}         # (kind, layer name, kernel variant) -> the same: where the step's time goes, layer by layer

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, kind: str, flops: float, e0, nbytes: float = 0.0, variant: Optional[str] = None, layer: Optional[str] = None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        r = self.rows.setdefault(kind, [0, 0.0, [], 0.0])
        r[0] += 1; r[1] += flops; r[2].append((e0, e1)); r[3] += nbytes
        if variant:
            # one row per kernel TEMPLATE, as rocprofv3 names kernels: the run-time regimes behind '|' and the bs1 / bs2 flavour of the
            # thin specialist's statistics epilogue (one template, a kernel argument) are summed
            tmpl = variant.split('|')[0].replace(',bs1>', ',bs>').replace(',bs2>', ',bs>').replace(',bs1,pl>', ',bs,pl>').replace(',bs2,pl>', ',bs,pl>')
            v = self.vrows.setdefault((kind, tmpl), [0, 0.0, [], 0.0])
            v[0] += 1; v[1] += flops; v[2].append((e0, e1)); v[3] += nbytes
            if layer:
                v = self.lrows.setdefault((kind, layer, tmpl), [0, 0.0, [], 0.0])
                v[0] += 1; v[1] += flops; v[2].append((e0, e1)); v[3] += nbytes

    def by_layer(self):
        """[{kind, layer, kernel, launches, ms, gflop, tflops}] sorted by time."""
        torch.cuda.synchronize()
        rows = []
        for (kind, layer, var), (n, fl, evs, nb) in self.lrows.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            rows.append(dict(kind=kind, layer=layer, kernel=var, launches=n, ms=ms, gflop=fl / 1e9,
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else None, algorithmic_bytes=nb))
        return sorted(rows, key=lambda r: -r['ms'])

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for k, (n, fl, evs, nb) in self.rows.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            out[k] = dict(launches=n, flops=fl, ms=ms, bytes=nb)
        return out

    def by_variant(self):
        """[{kind, kernel, launches, ms, gflop, tflops, bytes}] sorted by time: one row per kernel template."""
        torch.cuda.synchronize()
        rows = []
        for (kind, var), (n, fl, evs, nb) in self.vrows.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            rows.append(dict(kind=kind, kernel=var, launches=n, ms=ms, gflop=fl / 1e9, tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                             algorithmic_bytes=nb))
        return sorted(rows, key=lambda r: -r['ms'])


PROF: Optional[KernelProfile] = None

# Weight-gradient launches go to a second HIP stream: within a layer the weight gradient (needs dY and the stored input)
# and the data-gradient chain (dY -> dX -> IN backward -> next layer) are independent, and on the small deep layers either
# alone leaves most CUs idle.  The side stream waits for the producer of dY; the main stream waits for the side stream
# before arena memory is recycled (Arena.release) and before the gradients are consumed (side_join).
LAZY_RELEASE = os.environ.get('VG_LAZY_RELEASE', '1') != '0'
FORK_SHORT = os.environ.get('VG_FORK_SHORT', '1') != '0'   # forward shortcut branches on the lane's side stream
FUSE_CONCAT_NORM = os.environ.get('VG_FUSE_CONCAT_NORM', '1') != '0'   # ... and the conv branch's IN backward apply in the same launch
FUSE_CONCAT = os.environ.get('VG_FUSE_CONCAT', '1') != '0'       # decoder shortcut data gradient + concat backward in one launch (ConvLayer.dgrad_concat)
BSTAT = os.environ.get('VG_BSTAT', '1') != '0'       # IN-backward statistics with the data-gradient launch (ConvLayer.dgrad(bstat=...))
# ... also in the epilogue of the LDS-DMA family for its strided / dropout-carrying / sample-aliased uses (the encoder's stride-2 layers, the
# discriminators).  Off: measured, the epilogue's gathered loads of the pre-norm tensor cost the thin units of those launches more than the
# statistics pass they replace (D.down0: +65 us against 37); the stride-1 wide layers of the generators carry them either way (neutral in
# time, 18 launches fewer per step)
BSTAT_DMA = os.environ.get('VG_BSTAT_DMA', '0') != '0'
SIDE: Optional[bool] = None                   # truthy: weight-gradient side streams enabled (one per issuing stream)
_SIDE_OF = {}                                  # (device index, issuing stream handle) -> its side stream.  The default stream
#                                                has handle 0 on every device, hence the device index in the key; entries
#                                                are never dropped, so several engines in one process share them safely


def side_enable(device, on: bool = True):
    global SIDE
    SIDE = True if on else None


def _side_key(cur: torch.cuda.Stream):
    return (cur.device.index, cur.cuda_stream)


def _side_of(cur: torch.cuda.Stream) -> torch.cuda.Stream:
    sd = _SIDE_OF.get(_side_key(cur))
    if sd is None:
        sd = _SIDE_OF[_side_key(cur)] = torch.cuda.Stream(device=cur.device)
    return sd


class fork_side:
    """with fork_side() as f: ...launches...; f.join() -- the block runs on the current stream's side stream behind everything the
    current stream has issued so far; join() makes the current stream wait for it.  Serial (a no-op) when side streams are off,
    in the per-launch timing pass and in dry runs."""

    def __init__(self):
        self.on = SIDE is not None and PROF is None and DRY is None and FORK_SHORT
        self.cur = self.sd = self.ctx = None

    def __enter__(self):
        if self.on:
            self.cur = current_stream_obj()
            self.sd = _side_of(self.cur)
            wait_stream(self.sd, self.cur)
            self.ctx = torch.cuda.stream(self.sd)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            wait_stream(self.cur, self.sd)


def side_event():
    """An event behind everything the current stream has handed to its side stream so far (None: no side stream in use)."""
    if SIDE is None:
        return None
    sd = _SIDE_OF.get(_side_key(current_stream_obj()))
    return None if sd is None else record_event(sd)


def side_join():
    """The current stream waits for every weight-gradient launch it handed to its side stream."""
    if SIDE is not None:
        cur = current_stream_obj()
        sd = _SIDE_OF.get(_side_key(cur))
        if sd is not None:
            wait_stream(cur, sd)


CONV_SCRATCH = {}      # (device index, stream) -> uint8 workspace of vg_conv3d launches on that stream (vg_conv_desc::scratch)
CONV_SCRATCH_BYTES = _lib.SCRATCH_CTR_BYTES + (int(os.environ.get('VG_CONV_SCRATCH_MB', '160')) << 20)
_SCRATCH_RETIRED = []   # outgrown workspaces (see conv_scratch)


def conv_scratch(d: ConvDesc, s_: int, device=None):
    """Hand the issuing stream's workspace to a vg_conv3d descriptor: arrival counters (zeroed here, once; every launch leaves them
    at zero) + fp32 partial tiles of K-split launches + the materialised operand of the LDS-DMA convolution.  One buffer per
    (device, stream): launches of one stream are ordered, the lanes and their side streams run concurrently.  The library itself
    never allocates (include/vangan_hip.h)."""
    # a call with weights in the LDS-DMA block layout has no other kernel, and its operand grows with the batch: the workspace is
    # grown to what THIS call needs (vg_conv3d_scratch_bytes) instead of failing with VG_EINVAL on a larger batch or patch
    need = CONV_SCRATCH_BYTES
    if d.wlayout:
        need = max(need, int(lib.vg_conv3d_scratch_bytes(C.byref(d))))
    if DRY is not None:                       # dry runs plan exactly as the real launch does: same (dummy) workspace size
        d.scratch, d.scratch_bytes = 1 << 20, need
        return
    key = (_DEV if device is None else device, s_)
    sc = CONV_SCRATCH.get(key)
    if sc is None or sc.numel() < need:
        dev = torch.device('cuda', key[0]) if key[0] is not None else torch.device('cuda')
        if sc is not None:
            _SCRATCH_RETIRED.append(sc)       # launches in flight on the stream may still use it: never handed back to the allocator
        sc = torch.empty(need, dtype=torch.uint8, device=dev)
        # s_ is torch's current stream at every call site (ops.stream()): the memset precedes every launch that will use the buffer
        sc[:_lib.SCRATCH_CTR_BYTES].zero_()
        CONV_SCRATCH[key] = sc
    d.scratch, d.scratch_bytes = sc.data_ptr(), sc.numel()


WGRAD_SCRATCH = {}     # (device, stream) -> fp32 scratch of the weight gradients: materialised operand + partial slabs (384 MB)
WGRAD_SCRATCH_ELEMS = 96 << 20


# ------------------------------------------------------------------------------------------------------
# workspace arena: deterministic addresses, one allocation per engine
# ------------------------------------------------------------------------------------------------------
class Arena:
    ZPOOL = 8 << 20         # small zero-initialised allocations come from one pool cleared by a single memset per step

    def __init__(self, nbytes: int, device):
        self.buf = torch.empty(nbytes + self.ZPOOL, dtype=torch.uint8, device=device)
        self.zpool = self.buf[nbytes:]
        self.buf = self.buf[:nbytes]
        self.off = 0
        self.zoff = 0
        self.peak = 0
        self.lazy_ok = True         # False: this arena is shared by sweeps of both lanes (VG_LANES=0), so release() always recycles
        self.zpool.zero_()
        self._pair, self._pairs, self._pair_pos, self._full = None, {}, {}, {}
        self._cache, self._cpos, self._halves = [], 0, {}

    # Paired allocation (the two applications of one generator in a train step): while pair_begin(key, 0) is active every
    # non-zeroed allocation [N, ...] reserves [2N, ...] and returns the first half; pair_begin(key, 1) replays the SAME allocation
    # sequence and returns the second halves.  The forward passes stay ordinary N-sample launches on contiguous sample slices, the
    # backward sweep runs ONCE over the 2N-sample tensors (full_of maps a first-half view to its whole).
    def pair_begin(self, key, slot: int):
        self._pair = (key, slot)
        self._pair_pos[key] = 0
        if slot == 0:
            self._pairs[key] = []

    def pair_resume(self, key, slot: int):
        """Continue the sequence of (key, slot) after other allocations (two forward passes enqueued alternately)."""
        self._pair = (key, slot)

    def pair_end(self):
        self._pair = None

    def full_of(self, t):
        return None if t is None else self._full.get(t.data_ptr())

    def reset(self):
        self._pair, self._pairs, self._pair_pos, self._full = None, {}, {}, {}
        self._cpos = 0
        self.off = 0
        self.zmax = max(getattr(self, 'zmax', 0), self.zoff)
        # a RECORDED reset must clear what every later replay will have used, not what the steps before the recording happened to use
        # (a test_step / generate / inference call in front of the first train_step_replay leaves a small non-zero high-water mark, and
        # a replayed train step would then run on stale tickets and accumulators): always the whole pool -- 8 MiB, a few microseconds
        nz = self.zoff if REC is None else self.ZPOOL
        if nz:
            zero_fill(self.zpool[:nz])
        self.zoff = 0

    _ESZ = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.int32: 4, torch.int64: 8, torch.uint8: 1, torch.float64: 8}

    def alloc(self, shape: Sequence[int], dtype: torch.dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(shape)
        if self._pair is not None and not zero:
            key, slot = self._pair
            if slot == 0:
                self._pair = None
                full = self.alloc((2 * shape[0],) + shape[1:], dtype)
                self._pair = (key, slot)
                self._pairs[key].append(full)
                self._full[full.data_ptr()] = full
                h = self._halves.get(full.data_ptr())            # (whole, first half, second half): views made once, not once per step
                if h is None or h[0] is not full:
                    h = self._halves[full.data_ptr()] = (full, full[:shape[0]], full[shape[0]:])
                return h[1]
            full = self._pairs[key][self._pair_pos[key]]
            self._pair_pos[key] += 1
            assert tuple(full.shape) == (2 * shape[0],) + shape[1:] and full.dtype == dtype, 'paired allocation sequences differ'
            return self._halves[full.data_ptr()][2]
        # A train step asks for the same ~1300 blocks in the same order every time, and the three torch view operations behind a block
        # cost ~3 us: a quarter of the host's enqueue time per step (cProfile, tools/host_profile.py).  The sequence is cached -- entry
        # i is reused while the request (offsets, shape, type) repeats, and rebuilt from the first difference on.
        i, cache = self._cpos, self._cache
        if i < len(cache):
            e = cache[i]
            if e[0] == self.off and e[1] == self.zoff and e[2] == shape and e[3] is dtype and e[4] == zero:
                self._cpos = i + 1
                self.off, self.zoff = e[5], e[6]
                if self.off > self.peak:
                    self.peak = self.off
                if e[8] and DRY is None:
                    zero_fill(e[7])
                return e[7]
        off0, zoff0 = self.off, self.zoff
        t, big_zero = self._carve(shape, dtype, zero)
        entry = (off0, zoff0, shape, dtype, zero, self.off, self.zoff, t, big_zero)
        if i < len(cache):
            cache[i] = entry
            del cache[i + 1:]
        else:
            cache.append(entry)
        self._cpos = i + 1
        return t

    def _carve(self, shape, dtype, zero):
        n = int(math.prod(shape))
        nbytes = n * self._ESZ[dtype]
        if zero and nbytes <= 65536:
            zs = (self.zoff + 255) // 256 * 256
            if zs + nbytes <= self.ZPOOL:
                self.zoff = zs + nbytes
                return self.zpool[zs:zs + nbytes].view(dtype).view(*shape), False
        start = (self.off + 255) // 256 * 256
        if start + nbytes > self.buf.numel():
            raise MemoryError('arena exhausted: need %d more bytes' % (start + nbytes - self.buf.numel()))
        self.off = start + nbytes
        self.peak = max(self.peak, self.off)
        t = self.buf[start:start + nbytes].view(dtype).view(*shape)
        if zero and DRY is None:
            zero_fill(t)
        return t, bool(zero)

    def mark(self) -> int:
        return self.off

    def release(self, mark: int, defer: bool = False):
        """Hand the allocations since `mark` back.  defer=True (the backward sweeps): when weight-gradient launches may still be
        reading them on a side stream, the memory is simply NOT recycled before the next reset() -- joining the side stream here
        stalled the data-gradient chain behind every block's weight gradients (each lane idle 45 % of a step in the kernel trace);
        the workspace is sized for it (a few GB more at 128^3, of 288)."""
        if defer and LAZY_RELEASE and self.lazy_ok and SIDE is not None and PROF is None and DRY is None:
            return
        side_join()                 # weight gradients on the side stream may still read the buffers being recycled
        self.off = mark


# ------------------------------------------------------------------------------------------------------
# convolution geometry
# ------------------------------------------------------------------------------------------------------
def same_pad_before(n: int, k: int, s: int) -> int:
    """TF 'SAME' (TP): total=max((ceil(n/s)-1)*s+k-n,0), before=total//2."""
    out = -(-n // s)
    return max((out - 1) * s + k - n, 0) // 2


class Src:
    """Input operand of a gather-convolution: (virtual concat of) tensor(s) + on-read transform."""

    def __init__(self, x0: torch.Tensor, dims: Tuple[int, int, int, int], c0: int, x1: Optional[torch.Tensor] = None,
                 c1: int = 0, shift0: int = 0, f32: bool = False, scale=None, shift=None, act: int = ACT_NONE,
                 noise=None, noise_pad: int = 0):
        self.x0, self.x1, self.c0, self.c1, self.shift0, self.f32 = x0, x1, c0, c1, shift0, f32
        self.N, self.D, self.H, self.W = dims
        self.scale, self.shift, self.act, self.noise, self.noise_pad = scale, shift, act, noise, noise_pad

    @property
    def C(self):
        return self.c0 + self.c1

    def recipe(self) -> dict:
        """Shape-level description (no data): enough to rebuild an equivalent operand with random contents."""
        return dict(N=self.N, dims=(self.D, self.H, self.W), c0=self.c0, c1=self.c1, shift0=self.shift0, f32=bool(self.f32),
                    affine=self.scale is not None, act=self.act, noise=self.noise is not None, noise_pad=self.noise_pad)

    def fill(self, d: ConvDesc):
        d.src0, d.src1 = _p(self.x0), _p(self.x1)
        d.c_src0, d.c_src1, d.src0_shift, d.src_f32 = self.c0, self.c1, self.shift0, int(self.f32)
        d.N, d.D, d.H, d.W = self.N, self.D, self.H, self.W
        d.in_scale, d.in_shift, d.act = _p(self.scale), _p(self.shift), self.act
        d.noise, d.noise_pad = _p(self.noise), self.noise_pad


def _set_taps(d: ConvDesc, taps: List[Tuple[int, int, int]]):
    d.ntaps = len(taps)
    for i, (a, b, c) in enumerate(taps):
        d.tap_d[i], d.tap_h[i], d.tap_w[i] = a, b, c


def _ck_candidates(C_: int) -> List[int]:
    if C_ == 1:
        return [16]
    return [c for c in (64, 48, 32, 16) if C_ % c == 0]


class ConvLayer:
    """One Conv3D of the reference (k^3, stride, 'reflect' = ReflectionPadding3D+valid, or 'same'), with its
    packed bf16 weights for the forward and for every output-parity class of the data gradient."""

    def __init__(self, store, name: str, k: int, cin: int, cout: int, stride: int, pad: str, bias: bool,
                 in_dims: Tuple[int, int, int], need_dgrad: bool = True, dtype: torch.dtype = torch.bfloat16,
                 tap_subset: Optional[Sequence[int]] = None):
        """tap_subset (forward only): this object multiplies only the listed taps (indices into the k^3 raster order of the DHWIO
        kernel) -- a 7^3 convolution (343 taps, VG_MAX_TAPS is 64) is a chain of such chunks accumulating into one output
        (nets.ResNetGenerator's head, generator.py:68)."""
        assert tap_subset is None or stride == 1            # (a chunk's data gradient: one output-parity class with the chunk's taps)
        self.dtype, self.f32 = dtype, int(dtype == torch.float32)
        self.ctor = dict(k=k, cin=cin, cout=cout, stride=stride, pad=pad, bias=bias, in_dims=tuple(in_dims), need_dgrad=need_dgrad)
        self.name, self.k, self.cin, self.cout, self.stride, self.pad, self.has_bias = name, k, cin, cout, stride, pad, bias
        self.w, self.gw = store.param(name + '.w'), store.grad(name + '.w')
        self.b, self.gb = (store.param(name + '.b'), store.grad(name + '.b')) if bias else (None, None)
        dev = self.w.device
        D, H, W = in_dims
        self.in_dims = in_dims
        if pad == 'reflect':
            self.pb = (1, 1, 1)
            self.out_dims = tuple((n + 2 - k) // stride + 1 for n in in_dims)
            self.pad_mode = PAD_REFLECT
        else:
            self.pb = tuple(same_pad_before(n, k, stride) for n in in_dims)
            self.out_dims = tuple(-(-n // stride) for n in in_dims)
            self.pad_mode = PAD_ZERO
        T = k * k * k
        # ---- forward ----
        # single-channel source with k > 1: W-packed (include/vangan_hip.h: wpack) -- the k taps along W become k
        # pseudo-channels, k*k (d, h) taps remain; the DHWIO kernel is read as [k*k][k][cout] without moving a byte
        self.wpack = k if (cin == 1 and 1 < k <= 8 and tap_subset is None and os.environ.get('VG_WPACK', '1') != '0') else 0
        if tap_subset is not None:
            full = [(a - self.pb[0], b - self.pb[1], c - self.pb[2]) for a in range(k) for b in range(k) for c in range(k)]
            self.f_taps = [full[i] for i in tap_subset]
            self.f_T, self.f_cin = len(self.f_taps), cin
        elif self.wpack:
            self.f_taps = [(a - self.pb[0], b - self.pb[1], 0) for a in range(k) for b in range(k)]
            self.f_T, self.f_cin = k * k, k
        else:
            self.f_taps = [(a - self.pb[0], b - self.pb[1], c - self.pb[2]) for a in range(k) for b in range(k) for c in range(k)]
            self.f_T, self.f_cin = T, cin
        T = self.f_T
        sub = list(range(T)) if tap_subset is None else [int(i) for i in tap_subset]
        # weight gradient of a tap chunk: the chunk's taps are a contiguous run of the DHWIO tensor [tap][ci][co], so its dW is a
        # contiguous slice of the gradient buffer (the partial-slab path sums whole [T][ci][co] slabs: it must not see foreign taps)
        self.gw_w, self.w_idx_host = self.gw, None
        if tap_subset is not None:
            assert sub == list(range(sub[0], sub[0] + len(sub))), 'tap_subset must be a contiguous run of taps'
            per_tap = cin * cout
            self.gw_w = self.gw.view(-1)[sub[0] * per_tap:(sub[0] + len(sub)) * per_tap]
            self.w_idx_host = (C.c_int32 * len(sub))(*range(len(sub)))
        self.f_idx_host = (C.c_int32 * T)(*sub)
        self.f_idx = torch.tensor(sub, dtype=torch.int32, device=dev)
        # the wide layers run on the LDS-DMA family (vg_conv_dma.hip): the library says which, and with which channel panel; their
        # weights are packed in that kernel's block layout (vg_pack_weights_dma)
        self.f_bn = self._dma_bn(cin, cout, self.f_taps, stride, in_dims, self.out_dims) if (not self.wpack and tap_subset is None) else 0
        if self.f_bn:
            self.f_ck = 16
            self.f_wp = torch.zeros(cout * cin * T, dtype=dtype, device=dev)
        else:
            self.f_ck = self._pick_ck(cin, self.f_taps, stride, in_dims, self.out_dims, cout, wpack=self.wpack)
            self.f_ktot = check(lib.vg_packed_ktot(T, self.f_cin, self.f_ck), 'vg_packed_ktot')
            self.f_wp = torch.zeros(lib.vg_packed_rows(cout), self.f_ktot, dtype=dtype, device=dev)
        # ---- data gradient: one class per output parity; buffer = padded grid for 'reflect' ----
        self.d_classes = []
        self.d_bn, self.d_fused = 0, False
        if need_dgrad:
            padded = pad == 'reflect'
            self.buf_dims = tuple(n + 2 for n in in_dims) if padded else tuple(in_dims)
            per_dim = []
            for ax in range(3):
                pbe = 0 if padded else self.pb[ax]
                lst = []
                for pc in range(stride):
                    taps = [(t, (pc + pbe - t) // stride) for t in range(k) if (pc + pbe - t) % stride == 0]
                    cnt = -(-(self.buf_dims[ax] - pc) // stride)
                    lst.append((pc, taps, cnt))
                per_dim.append(lst)
            for (pd, td, nd) in per_dim[0]:
                for (ph, th, nh) in per_dim[1]:
                    for (pw, tw, nw) in per_dim[2]:
                        if not td or not th or not tw or min(nd, nh, nw) < 1:
                            continue
                        taps, idx = [], []
                        for (a, oa) in td:
                            for (b, ob) in th:
                                for (c, oc) in tw:
                                    if tap_subset is not None and ((a * k + b) * k + c) not in sub:
                                        continue
                                    taps.append((oa, ob, oc)); idx.append((a * k + b) * k + c)
                        self.d_classes.append(dict(off=(pd, ph, pw), iters=(nd, nh, nw), taps=taps, idx_list=idx))
            # the wide layers: ONE class-parallel launch of the LDS-DMA family over the zero-padded dY (every class a stride-1 walk)
            self.d_fused = False
            self.d_bn = self._dma_bn_dgrad()
            if self.d_bn:
                for c in self.d_classes:
                    c['ck'] = 16
                self.d_fused = True
            # strided convs: ONE fused launch computes all output-parity classes from a dY halo tile staged once, when
            # dY fits one channel chunk (<= 64 channels) and the plan is LDS-feasible; else one launch per class
            if not self.d_bn and len(self.d_classes) > 1 and cout <= 64 and not self.f32 and sum(len(c['taps']) for c in self.d_classes) <= _lib.VG_MAX_TAPS:
                ck_f = -(-cout // 16) * 16
                for c in self.d_classes:
                    c['ck'] = ck_f
                self.d_fused = self._fused_desc(None, 1, None, False, probe=True) is not None
            # otherwise (several channel chunks of dY, or exact-parity mode): still ONE launch, class-parallel -- every
            # workgroup serves one class, so the launch carries all classes' tiles at once (8 launches of 8-64 workgroups
            # each left most of the chip idle on enc3/enc4/D.down1).  All classes share the chunk size.
            if (not self.d_bn and not self.d_fused and len(self.d_classes) > 1 and os.environ.get('VG_CLS_PAR', '1') != '0'
                    and sum(len(c['taps']) for c in self.d_classes) <= _lib.VG_MAX_TAPS):
                best, best_key = None, None
                for ck in _ck_candidates(cout):
                    for c in self.d_classes:
                        c['ck'] = ck
                    if self._fused_desc(None, 1, None, False, probe=True) is None:
                        continue
                    plan = self._last_plan
                    key = (plan[0] * plan[1], 1 if plan[2] <= 80 * 1024 else 0, ck)
                    if ck < 32 and best is not None and best_key[0] >= key[0]:
                        continue
                    if best_key is None or key > best_key:
                        best, best_key = ck, key
                if best is not None:
                    for c in self.d_classes:
                        c['ck'] = best
                    self.d_fused = True
            for c in self.d_classes:
                c['idx'] = torch.tensor(c['idx_list'], dtype=torch.int32, device=dev)
                if self.d_bn:
                    c['wp'] = torch.zeros(cin * cout * len(c['taps']), dtype=dtype, device=dev)
                    continue
                if not self.d_fused:
                    c['ck'] = self._pick_ck(cout, c['taps'], 1, self.out_dims, c['iters'], cin)
                c['ktot'] = check(lib.vg_packed_ktot(len(c['taps']), cout, c['ck']), 'vg_packed_ktot')
                c['wp'] = torch.zeros(lib.vg_packed_rows(cin), c['ktot'], dtype=dtype, device=dev)
        # 4x4x4 stride-2 layer over the reflect-padded SINGLE-channel volume (the discriminators' first layer): its data gradient as a
        # stride-1 convolution over 2x2x2 cells of the padded grid on the thin-channel specialist + a fold (vg_pack_cell_weights,
        # vg_cells_fold; dgrad_input below) instead of 8 output-parity classes with one live MFMA column each
        self.cell = None
        if (need_dgrad and cin == 1 and k == 4 and stride == 2 and pad == 'reflect' and not self.f32 and cout % 16 == 0 and tap_subset is None
                and all(n % 2 == 0 and n >= 4 for n in in_dims) and os.environ.get('VG_CELL_DGRAD', '1') != '0'):
            ktot = check(lib.vg_packed_ktot(27, cout, 16), 'vg_packed_ktot')
            self.cell = dict(dims=tuple(n // 2 + 1 for n in in_dims), ktot=ktot,
                             wp=torch.zeros(lib.vg_packed_rows(16), ktot, dtype=dtype, device=dev),
                             taps=[(a - 1, b - 1, c - 1) for a in range(3) for b in range(3) for c in range(3)])

    def _dma_bn(self, cin, cout, taps, istr, in_dims, out_dims) -> int:
        """Channel-panel width with which the LDS-DMA family serves this forward convolution (0: the gather kernels do).
        Policy (measured layer by layer and in the step): the family is bound by LDS read bandwidth at ~50 % of the MFMA rate
        of its CUs.  Solo it beats the gather kernels only where their staging is expensive (the decoder's first convolutions over the
        virtual upsample + concat, >= 192 input channels: 'cat') and loses to conv32_kernel's register-streamed weights on the plain wide
        layers; in the concurrent step the forward gains nothing (20.66 vs 20.51 ms), so forwards stay on the gather kernels ('none')."""
        if self.f32 or os.environ.get('VG_CONV_DMA', '1') == '0' or len(taps) > _lib.VG_MAX_TAPS:
            return 0
        pol = os.environ.get('VG_CONV_DMA_FWD', 'none')
        if pol == 'none' or (pol == 'cat' and cin < 192) or math.prod(out_dims) < int(os.environ.get('VG_CONV_DMA_MINVOX', '4096')):
            return 0
        d = ConvDesc()
        d.c_src0, d.c_src1, d.N = cin, 0, 1
        d.D, d.H, d.W = in_dims
        d.istr, d.pad_mode, d.ostr = istr, self.pad_mode, 1
        _set_taps(d, taps)
        d.OD, d.OH, d.OW = out_dims
        d.BD, d.BH, d.BW = out_dims
        d.Cout = cout
        return max(0, lib.vg_conv3d_dma_bn(C.byref(d)))

    def _dma_bn_dgrad(self) -> int:
        """The same for the data gradient: all output-parity classes in one class-parallel launch."""
        if self.f32 or os.environ.get('VG_CONV_DMA', '1') == '0' or not self.d_classes:
            return 0
        # data gradients iterate over padded / parity-class grids (17^3, 18^3, 33^3, 34^3 ...): the family's linear tiles fill those
        # to 85-97 % where box tiles reach 43-68 %.  Not at the 8^3 level (a handful of tiles: the K split's exchange costs more)
        pol = os.environ.get('VG_CONV_DMA_DGRAD', 'all')
        if pol == 'none' or math.prod(self.out_dims) < int(os.environ.get('VG_CONV_DMA_MINVOX', '4096')):
            return 0
        if pol == 'big' and self.cin * self.cout < 128 * 64:
            return 0
        if sum(len(c['taps']) for c in self.d_classes) > _lib.VG_MAX_TAPS or len(self.d_classes) > 8:
            return 0
        for c in self.d_classes:
            c.setdefault('ck', 16)
        d = self._fused_desc(None, 1, None, False, probe=True, plan=False)
        return max(0, lib.vg_conv3d_dma_bn(C.byref(d)))

    def _pick_ck(self, C_, taps, istr, in_dims, iters, rows, wpack=0) -> int:
        d = ConvDesc()
        d.wpack, d.wpack_wmin = wpack, -self.pb[2]
        d.src0 = d.out = d.wpacked = 1 << 20
        d.c_src0, d.c_src1, d.N = C_, 0, 1
        d.D, d.H, d.W = in_dims
        d.istr, d.pad_mode, d.ostr = istr, PAD_ZERO, 1
        _set_taps(d, taps)
        d.OD, d.OH, d.OW = iters
        d.BD, d.BH, d.BW = iters
        d.Cout = rows
        d.f32 = self.f32
        # plan of every feasible chunk size; prefer the biggest tile (voxels x channel panel: fewer halo voxels staged
        # per output and more MFMAs per barrier), then two resident workgroups per CU (<= 80 KiB), then the largest
        # chunk (every further chunk is another staging pass per tile).  Measured on D.down0 (64->128, k4 s2, 64^3):
        # CK 32 / 128-voxel tile 0.128 ms, CK 16 / 128 voxels 0.196 ms, CK 64 / 64 voxels 0.247 ms.
        # the 32-channel layers at 64^3: two-panel instance of the thin-channel specialist, which wants 16-channel chunks
        if not wpack and not self.f32 and os.environ.get('VG_CONV_THIN2', '1') != '0' and lib.vg_conv3d_thin_np(C.byref(d)) == 2:
            return 16
        best, best_key = None, None
        plan = (C.c_int32 * 4)()
        for ck in _ck_candidates(C_):
            d.CK = ck
            if lib.vg_conv3d_plan(C.byref(d), plan) != 0:
                continue
            key = (plan[0] * plan[1], 1 if plan[2] <= 80 * 1024 else 0, ck)
            if ck < 32 and best is not None and best_key[0] >= key[0]:
                continue                       # do not trade chunk size below 32 channels for occupancy alone
            if best_key is None or key > best_key:
                best, best_key = ck, key
        if best is None:
            raise _lib.VgError('no LDS-feasible tile for %s' % self.name)
        return best

    def pack_items(self):
        """(w, tap_idx, out, Cin, Cout, ntaps, transpose, CK, f32, bn) of every packed operand, for PackTable (bn > 0: the block layout
        of the LDS-DMA family with that panel width)."""
        T = self.f_T
        items = [(self.w, self.f_idx, self.f_wp, self.f_cin, self.cout, T, 0, self.f_ck, self.f32, self.f_bn)]
        for c in self.d_classes:
            items.append((self.w, c['idx'], c['wp'], self.cin, self.cout, len(c['taps']), 1, c['ck'], self.f32, self.d_bn))
        return items

    def enable_up(self, c_up: int):
        """This layer's first c_up input channels are the virtually upsampled half-resolution tensor of a decoder block (UpSampling3D ->
        concatenate -> IN -> ReLU -> reflect pad -> 3x3x3 convolution, resunet_model.py:175-181, 42-66): the 16-channel specialist then
        contracts those chunks over the half-resolution image with the D / H taps collapsed (vg_conv_desc::wpacked_up; 12 taps instead of
        27, 40 % of the staging).  A no-op for layers the specialist does not serve.
        OFF by default (VG_CONV_THIN_UP=1 enables it): measured at 128^3, the launch with 38 % fewer MFMAs and 40 % less staging is 3 % faster
        (117.8 vs 121 us) and the step and the sliding-window inference 0.5-2 % SLOWER with the four extra repack launches -- a chunk
        pass of this kernel costs ~5 us whatever it multiplies (load round trip + commit + three barriers against 0.3-0.75 us of MFMAs),
        so removing work inside a pass buys nothing; removing PASSES would."""
        self.wp_up, self.c_up = None, 0
        if (self.f32 or self.k != 3 or self.stride != 1 or self.pad_mode != PAD_REFLECT or c_up <= 0 or c_up % 16 or c_up >= self.cin
                or self.f_bn or self.f_ck != 16 or self.wpack or os.environ.get('VG_CONV_THIN_UP', '0') == '0'):
            return
        self.c_up = c_up
        self.ctor = dict(self.ctor, c_up=c_up)
        self.wp_up = torch.zeros((c_up // 16) * 4 * self.cout * 192, dtype=self.f_wp.dtype, device=self.f_wp.device)

    def pack_up(self):
        if getattr(self, 'wp_up', None) is not None:
            check(lib.vg_pack_up_weights(_p(self.w), self.cin, self.cout, self.c_up, _p(self.wp_up), stream()), 'vg_pack_up_weights ' + self.name)

    def pack_cell(self):
        """The packed operand of the cell form of the data gradient (dgrad_input) from the fp32 master weights."""
        if getattr(self, 'cell', None) is not None:
            check(lib.vg_pack_cell_weights(_p(self.w), self.cout, _p(self.cell['wp']), stream()), 'vg_pack_cell_weights ' + self.name)

    def pack(self):
        """fp32 master weights -> bf16 packed operands (after every optimizer step)."""
        T = self.f_T
        s = stream()
        self.pack_cell()
        self.pack_up()
        if self.f_bn:
            check(lib.vg_pack_weights_dma(_p(self.w), T, self.f_cin, self.cout, _p(self.f_idx), T, 0, self.f_bn, _p(self.f_wp), s), 'pack')
        else:
            check(lib.vg_pack_weights(_p(self.w), max(T, self.k ** 3 if not self.wpack else T), self.f_cin, self.cout, _p(self.f_idx), T, 0,
                                      self.f_ck, _p(self.f_wp), self.f32, s), 'pack')
        for c in self.d_classes:
            if self.d_bn:
                check(lib.vg_pack_weights_dma(_p(self.w), self.k ** 3, self.cin, self.cout, _p(c['idx']), len(c['taps']), 1, self.d_bn,
                                              _p(c['wp']), s), 'pack')
            else:
                check(lib.vg_pack_weights(_p(self.w), self.k ** 3, self.cin, self.cout, _p(c['idx']), len(c['taps']), 1, c['ck'],
                                          _p(c['wp']), self.f32, s), 'pack')

    def _fwd_desc(self, src: Src) -> ConvDesc:
        # the static part (taps, geometry, packed weights) is built once and block-copied: filling ~100 ctypes fields from
        # Python cost ~9 us per call, 780 calls per train step
        t = getattr(self, '_fwd_tmpl', None)
        if t is None:
            t = ConvDesc()
            t.istr, t.pad_mode = self.stride, self.pad_mode
            t.wpack, t.wpack_wmin = self.wpack, -self.pb[2]
            _set_taps(t, self.f_taps)
            t.OD, t.OH, t.OW = self.out_dims
            t.ostr, t.ooff_d, t.ooff_h, t.ooff_w = 1, 0, 0, 0
            t.BD, t.BH, t.BW = self.out_dims
            t.Cout, t.wpacked, t.CK = self.cout, _p(self.f_wp), self.f_ck
            t.f32 = self.f32
            t.wlayout = self.f_bn
            self._fwd_tmpl = t
        d = ConvDesc()
        C.memmove(C.byref(d), C.byref(t), C.sizeof(ConvDesc))
        src.fill(d)
        if getattr(self, 'wp_up', None) is not None and d.src0_shift and d.c_src0 == self.c_up:
            d.wpacked_up = _p(self.wp_up)
        return d

    def forward(self, src: Src, out: torch.Tensor, sums=None, res=None, res_scale=None, res_shift=None,
                tanh: bool = False, accumulate: bool = False, fin: Optional[FinDesc] = None, res_c1: bool = False):
        """fin (ops.fin_desc, with sums): the launch also finalises the InstanceNorm statistics of `out` for its consuming norm(s).
        res_c1: `res` is a single-channel fp32 volume [N, D, H, W, 1] broadcast over the output channels (vg_conv_desc::res_c1)."""
        assert src.C == self.cin and (src.D, src.H, src.W) == tuple(self.in_dims)
        d = self._fwd_desc(src)
        if fin is not None:
            assert sums is not None
            d.fin = C.addressof(fin)
            d._fin_keep = fin
            if REC is not None:
                REC.keep.append(fin)
        d.bias = _p(self.b)
        d.res, d.res_scale, d.res_shift = _p(res), _p(res_scale), _p(res_shift)
        d.res_c1 = int(bool(res_c1) and res is not None)
        assert not d.res_c1 or res.dtype == torch.float32
        d.tanh_out = int(tanh)
        d.out, d.out_f32, d.accumulate = _p(out), int(out.dtype == torch.float32), int(bool(accumulate))
        d.out_sums = _p(sums)
        s_ = stream()
        conv_scratch(d, s_, out.device.index)
        if DRY is not None:
            DRY.tag = ('fwd', self.name)
            DRY.recipe = dict(kind='fwd', layer=self.ctor, src=src.recipe(), res=res is not None, res_c1=bool(res_c1), tanh=bool(tanh),
                              sums=sums is not None, out_f32=out.dtype == torch.float32)
        e0 = PROF.begin() if PROF is not None else None
        check(lib.vg_conv3d(C.byref(d), s_), 'vg_conv3d ' + self.name)
        if e0 is not None:
            esz = 4 if self.f32 else 2
            PROF.end('conv_fwd', 2.0 * src.N * math.prod(self.out_dims) * self.cout * self.cin * self.k ** 3, e0,
                     src.N * esz * (math.prod(self.in_dims) * self.cin + math.prod(self.out_dims) * self.cout), conv_variant(d), self.name)

    def wgrad(self, src: Src, dy: torch.Tensor, inline: bool = False):
        """inline: launch on the issuing stream itself instead of its side stream (the tail of a lane's last sweep, where the side
        stream is the one that is behind); a per-call argument, so that two sweeps enqueued alternately cannot see each other's."""
        if SIDE is not None and PROF is None and DRY is None and not inline:   # the per-launch timing pass serialises (attributable durations)
            cur = current_stream_obj()
            sd = _side_of(cur)
            wait_stream(sd, cur)                                 # dY (and everything before it) is ready
            self._wgrad(src, dy, sd.cuda_stream)                 # launched on the side stream by handle: torch's current
        else:                                                    # stream is not switched (the context manager cost ~10 us)
            self._wgrad(src, dy)

    def _wgrad(self, src: Src, dy: torch.Tensor, on_stream: Optional[int] = None):
        d = self._fwd_desc(src)
        s_ = stream() if on_stream is None else on_stream
        if DRY is not None:
            DRY.tag = ('wgrad', self.name)
            DRY.recipe = dict(kind='wgrad', layer=self.ctor, src=src.recipe(), dy_f32=dy.dtype == torch.float32)
        e0 = PROF.begin() if PROF is not None else None
        # one partial-slab scratch per (device, stream): launches on one stream reuse it in order, the two lanes of the
        # engine issue weight gradients concurrently and must not share it
        key = (dy.device, s_)
        sc = WGRAD_SCRATCH.get(key)
        if sc is None:
            sc = WGRAD_SCRATCH[key] = torch.empty(WGRAD_SCRATCH_ELEMS, dtype=torch.float32, device=dy.device)
        widx = self.f_idx_host if self.w_idx_host is None else self.w_idx_host
        check(lib.vg_conv3d_wgrad(C.byref(d), _p(dy), int(dy.dtype == torch.float32), widx, self.f_T,
                                  _p(self.gw_w), _p(self.gb), _p(sc), sc.numel() * 4, s_), 'vg_conv3d_wgrad ' + self.name)
        if e0 is not None:
            esz = 4 if self.f32 else 2
            vb = C.create_string_buffer(512)
            _lib.lib.vg_conv3d_wgrad_variant(C.byref(d), int(dy.dtype == torch.float32), widx, self.f_T, sc.numel() * 4, vb, 512)
            PROF.end('conv_wgrad', 2.0 * src.N * math.prod(self.out_dims) * self.cout * self.cin * self.k ** 3, e0,
                     src.N * esz * (math.prod(self.in_dims) * self.cin + math.prod(self.out_dims) * self.cout), vb.value.decode(), self.name)

    def _fused_desc(self, dy, N, out, accumulate, probe=False, plan=True):
        """Descriptor of the fused all-classes data gradient (probe=True: dummy pointers, returns None if infeasible; plan=False:
        the bare descriptor, no feasibility check)."""
        d = ConvDesc()
        dummy = 1 << 20
        d.src0, d.src1 = (dummy if probe else _p(dy)), None
        d.c_src0, d.c_src1, d.src0_shift = self.cout, 0, 0
        d.src_f32 = 0 if probe else int(dy.dtype == torch.float32 and self.cout == 1)
        d.N = N
        d.D, d.H, d.W = self.out_dims
        d.act, d.istr, d.pad_mode = ACT_NONE, 1, PAD_ZERO
        taps = [t for c in self.d_classes for t in c['taps']]
        _set_taps(d, taps)
        d.OD, d.OH, d.OW = [max(c['iters'][a] for c in self.d_classes) for a in range(3)]
        d.ostr = self.stride
        d.ooff_d, d.ooff_h, d.ooff_w = self.d_classes[0]['off']
        d.BD, d.BH, d.BW = self.buf_dims
        d.Cout, d.CK = self.cin, self.d_classes[0]['ck']
        d.f32 = self.f32
        d.nclass = len(self.d_classes)
        t0 = 0
        for i, c in enumerate(self.d_classes):
            d.cls_tap0[i] = t0
            t0 += len(c['taps'])
            d.cls_w[i] = dummy if probe else c['wp'].data_ptr()
            for a in range(3):
                d.cls_ooff[i][a] = c['off'][a]
                d.cls_iters[i][a] = c['iters'][a]
        d.cls_tap0[len(self.d_classes)] = t0
        d.wpacked = d.cls_w[0]
        d.wlayout = getattr(self, 'd_bn', 0)
        if probe and not plan:
            return d
        if probe:
            d.out = dummy
            plan = (C.c_int32 * 4)()
            ok = lib.vg_conv3d_plan(C.byref(d), plan) == 0
            self._last_plan = tuple(plan)
            return d if ok else None
        d.out, d.out_f32, d.accumulate = _p(out), int(out.dtype == torch.float32), int(accumulate)
        return d

    def dgrad(self, dy: torch.Tensor, N: int, out: torch.Tensor, accumulate: bool, bstat=None, bias=None) -> bool:
        """d/d input: writes the reflect-PADDED grid for 'reflect' convs (fold it with actnorm_bwd), the plain
        input grid for 'same' convs.  out: [N, *buf_dims, cin] bf16 (or f32 when cin==1).
        bstat: the actnorm_desc() of the IN backward that consumes `out` (bstat.g is out): when the data gradient is ONE launch,
        its statistics pass is done with it (in the kernel's epilogue where the kernel can, else right behind it) and True is
        returned -- the caller then runs actnorm_run(bstat, stats_done=True)."""
        # only where the 16-channel specialist (which carries the statistics in its epilogue) is the expected kernel: elsewhere the
        # separate statistics launch stays where it was
        # ... or the LDS-DMA family, whose epilogue goes through LDS (any class structure, channel-dropout multipliers, sample aliasing)
        use_bs = bstat is not None and BSTAT and not accumulate and not self.f32 and out.dtype in (torch.bfloat16, torch.float16) and bstat.norm \
            and ((len(self.d_classes) == 1 and self.stride == 1 and self.k == 3 and self.pad == 'reflect'
                  and self.d_classes[0]['ck'] == 16 and self.cin % 16 == 0 and self.cout % 16 == 0 and not bstat.mult)
                 or (bool(self.d_bn) and BSTAT_DMA))
        if DRY is not None:
            DRY.tag = ('dgrad', self.name)
            DRY.recipe = dict(kind='dgrad', layer=self.ctor, N=N, accumulate=bool(accumulate), dy_f32=dy.dtype == torch.float32,
                              out_f32=out.dtype == torch.float32, bstat=None if not use_bs else dict(
                                  cat=bool(bstat.x1), c_x0=bstat.c_x0, act=bstat.act, pad=bool(bstat.g_padded), mult=bool(bstat.mult),
                                  alias_n0=bstat.alias_n0, alias_shift=bstat.alias_shift))
        if REC is not None and use_bs:
            REC.keep.append(bstat)               # its address is baked into the recorded descriptor
        if self.d_fused:
            t = getattr(self, '_dg_tmpl', None)
            if t is None:                            # static part once (taps / weights / offsets of every class), then block copies
                t = self._dg_tmpl = self._fused_desc(dy, N, out, accumulate)
            d = ConvDesc()
            C.memmove(C.byref(d), C.byref(t), C.sizeof(ConvDesc))
            d.src0, d.N = _p(dy), N
            d.src_f32 = int(dy.dtype == torch.float32 and self.cout == 1)
            d.out, d.out_f32, d.accumulate = _p(out), int(out.dtype == torch.float32), int(accumulate)
            d.bstat = C.addressof(bstat) if use_bs else None
            d.bias = _p(bias)                 # (only the transposed-convolution use of this launch has one: ConvTranspose3dK2S2)
            s_ = stream()
            conv_scratch(d, s_, out.device.index)
            e0 = PROF.begin() if PROF is not None else None
            check(lib.vg_conv3d(C.byref(d), s_), 'vg_conv3d(dgrad, fused classes) ' + self.name)
            if e0 is not None:
                # algorithmic FLOPs = the forward's (every (output voxel, tap) pair once); the padded-grid iteration space of the
                # launch also multiplies dY's zero border, which is not counted
                PROF.end('conv_dgrad', 2.0 * N * math.prod(self.out_dims) * self.cin * self.cout * self.k ** 3, e0,
                         N * 2 * (math.prod(self.out_dims) * self.cout + math.prod(self.buf_dims) * self.cin), conv_variant(d), self.name)
            return use_bs
        for c in self.d_classes:
            t = c.get('tmpl')
            if t is None:
                t = ConvDesc()
                t.src1 = None
                t.c_src0, t.c_src1, t.src0_shift = self.cout, 0, 0
                t.D, t.H, t.W = self.out_dims
                t.act, t.istr, t.pad_mode = ACT_NONE, 1, PAD_ZERO
                _set_taps(t, c['taps'])
                t.OD, t.OH, t.OW = c['iters']
                t.ostr = self.stride
                t.ooff_d, t.ooff_h, t.ooff_w = c['off']
                t.BD, t.BH, t.BW = self.buf_dims
                t.Cout, t.wpacked, t.CK = self.cin, _p(c['wp']), c['ck']
                t.f32 = self.f32
                c['tmpl'] = t
            d = ConvDesc()
            C.memmove(C.byref(d), C.byref(t), C.sizeof(ConvDesc))
            d.src0, d.N = _p(dy), N
            d.src_f32 = int(dy.dtype == torch.float32 and self.cout == 1)
            d.out, d.out_f32, d.accumulate = _p(out), int(out.dtype == torch.float32), int(accumulate)
            d.bstat = C.addressof(bstat) if use_bs else None
            d.bias = _p(bias)
            s_ = stream()
            conv_scratch(d, s_, out.device.index)
            e0 = PROF.begin() if PROF is not None else None
            check(lib.vg_conv3d(C.byref(d), s_), 'vg_conv3d(dgrad) ' + self.name)
            if e0 is not None:      # algorithmic FLOPs of this parity class: its taps only
                PROF.end('conv_dgrad', 2.0 * N * math.prod(self.out_dims) * self.cin * self.cout * len(c['taps']), e0,
                         N * (4 if self.f32 else 2) * (math.prod(self.out_dims) * self.cout * len(c['taps']) / self.k ** 3
                                                       + math.prod(c['iters']) * self.cin), conv_variant(d), self.name)
        return use_bs


    def _sc_desc(self, dy, N):
        c = self.d_classes[0]
        d = ConvDesc()
        d.src1 = None
        d.c_src0, d.c_src1, d.src0_shift = self.cout, 0, 0
        d.D, d.H, d.W = self.out_dims
        d.act, d.istr, d.pad_mode = ACT_NONE, 1, PAD_ZERO
        _set_taps(d, c['taps'])
        d.OD, d.OH, d.OW = c['iters']
        d.ostr = self.stride
        d.ooff_d, d.ooff_h, d.ooff_w = c['off']
        d.BD, d.BH, d.BW = self.buf_dims
        d.Cout, d.wpacked, d.CK = self.cin, _p(c['wp']), c['ck']
        d.f32 = self.f32
        d.src0, d.N = _p(dy), N
        return d

    def dgrad_input(self, ar: 'Arena', dy: torch.Tensor, N: int, dx: torch.Tensor):
        """d/d input of a layer that reads a single-channel volume, folded through the reflection pad: dx fp32 [N, D, H, W, 1] is
        overwritten.  The 4x4x4 stride-2 layer runs in its cell form (see __init__; not in the dry-run walks, which enumerate the
        generic launch -- tests/test_gpu_ops.py compares the two forms at the true shape); everything else is dgrad onto the
        padded grid + the fold of vg_actnorm_bwd."""
        assert self.cin == 1 and self.pad == 'reflect'
        if self.cell is None or DRY is not None or dy.dtype not in (torch.bfloat16, torch.float16):
            dp = ar.alloc((N,) + tuple(self.buf_dims) + (1,), self.dtype)
            self.dgrad(dy, N, dp, accumulate=False)
            actnorm_bwd(dp, True, None, (N,) + tuple(self.in_dims), 1, dx, act=ACT_NONE, norm=False, accumulate=False)
            return
        c = self.cell
        cells = ar.alloc((N,) + c['dims'] + (16,), self.dtype)
        t = c.get('tmpl')
        if t is None:
            t = c['tmpl'] = ConvDesc()
            t.src1 = None
            t.c_src0, t.c_src1, t.src0_shift = self.cout, 0, 0
            t.D, t.H, t.W = self.out_dims
            t.act, t.istr, t.pad_mode, t.ostr = ACT_NONE, 1, PAD_ZERO, 1
            _set_taps(t, c['taps'])
            t.OD, t.OH, t.OW = c['dims']
            t.BD, t.BH, t.BW = c['dims']
            t.Cout, t.wpacked, t.CK = 16, _p(c['wp']), 16
            t.f32 = 0
        d = ConvDesc()
        C.memmove(C.byref(d), C.byref(t), C.sizeof(ConvDesc))
        d.src0, d.N, d.out = _p(dy), N, _p(cells)
        s_ = stream()
        conv_scratch(d, s_, dy.device.index)
        e0 = PROF.begin() if PROF is not None else None
        check(lib.vg_conv3d(C.byref(d), s_), 'vg_conv3d(dgrad, cells) ' + self.name)
        if e0 is not None:
            PROF.end('conv_dgrad', 2.0 * N * math.prod(self.out_dims) * self.cin * self.cout * self.k ** 3, e0,
                     N * 2 * (math.prod(self.out_dims) * self.cout + math.prod(c['dims']) * 16), conv_variant(d), self.name)
        D_, H_, W_ = self.in_dims
        check(lib.vg_cells_fold(_p(cells), N, D_, H_, W_, _p(dx), s_), 'vg_cells_fold ' + self.name)

    def dgrad_concat_norm(self, dy: torch.Tensor, N: int, nd: 'ActNormBwdDesc', c_low: int, dlow: torch.Tensor, dskip: torch.Tensor,
                          acc_low: bool, acc_skip: bool) -> bool:
        """dgrad_concat with the conv branch folded in (vg_shortcut_dgrad_concat_norm): nd is the actnorm_desc of the block's first
        convolution's input (statistics already in nd.red); nothing of the concat gradient is stored.  False: not served -- the
        caller runs the apply pass into a concat-gradient buffer and dgrad_concat."""
        if not (FUSE_CONCAT and FUSE_CONCAT_NORM and DRY is None and not self.f32 and self.k == 1 and self.stride == 1
                and len(self.d_classes) == 1 and not self.d_fused):
            return False
        c = self.d_classes[0]
        d = self._sc_desc(dy, N)
        e0 = PROF.begin() if PROF is not None else None
        rc = lib.vg_shortcut_dgrad_concat_norm(C.byref(d), C.byref(nd), _p(dlow), _p(dskip), c_low,
                                               int(bool(acc_low)) | (int(bool(acc_skip)) << 1), stream())
        if rc < 0:
            check(rc, 'vg_shortcut_dgrad_concat_norm ' + self.name)
        if rc == 0 and e0 is not None:
            PROF.end('conv_dgrad', 2.0 * N * math.prod(c['iters']) * self.cin * self.cout, e0,
                     N * 2 * (math.prod(self.out_dims) * self.cout + math.prod(self.buf_dims) * self.cin),
                     'pw_gemm_split<%d,%d,n1>' % ((self.cout + 31) // 32, self.cin // 16), self.name)
        return rc == 0

    def dgrad_concat(self, dy: torch.Tensor, N: int, dcat: torch.Tensor, c_low: int, dlow: torch.Tensor, dskip: torch.Tensor,
                     acc_low: bool, acc_skip: bool):
        """Decoder shortcut (1x1x1, stride 1): the data gradient added to the concat gradient `dcat` (which already holds the conv
        branch's part) and the backward of UpSampling3D + concatenate in one launch (vg_shortcut_dgrad_concat): the sums are written
        to dlow / dskip, dcat is only read.  Falls back to dgrad(accumulate) + concat_bwd where the fused kernel does not serve the
        shape (and in the dry-run / per-launch timing passes, which enumerate the convolution calls)."""
        fused = (FUSE_CONCAT and DRY is None and not self.f32 and self.k == 1 and self.stride == 1 and len(self.d_classes) == 1
                 and not self.d_fused and dcat.dtype == torch.bfloat16)
        if fused:
            c = self.d_classes[0]
            d = ConvDesc()
            d.src1 = None
            d.c_src0, d.c_src1, d.src0_shift = self.cout, 0, 0
            d.D, d.H, d.W = self.out_dims
            d.act, d.istr, d.pad_mode = ACT_NONE, 1, PAD_ZERO
            _set_taps(d, c['taps'])
            d.OD, d.OH, d.OW = c['iters']
            d.ostr = self.stride
            d.ooff_d, d.ooff_h, d.ooff_w = c['off']
            d.BD, d.BH, d.BW = self.buf_dims
            d.Cout, d.wpacked, d.CK = self.cin, _p(c['wp']), c['ck']
            d.f32 = self.f32
            d.src0, d.N = _p(dy), N
            d.out, d.out_f32, d.accumulate = _p(dcat), 0, 1
            e0 = PROF.begin() if PROF is not None else None
            rc = lib.vg_shortcut_dgrad_concat(C.byref(d), _p(dlow), _p(dskip), c_low, int(bool(acc_low)) | (int(bool(acc_skip)) << 1), stream())
            if rc == 0:
                if e0 is not None:
                    PROF.end('conv_dgrad', 2.0 * N * math.prod(c['iters']) * self.cin * self.cout, e0,
                             N * 2 * (math.prod(self.out_dims) * self.cout + math.prod(self.buf_dims) * self.cin),
                             'pw_gemm_split<%d,%d,n0>' % ((self.cout + 31) // 32, self.cin // 16), self.name)
                return
            if rc < 0:
                check(rc, 'vg_shortcut_dgrad_concat ' + self.name)
        self.dgrad(dy, N, dcat, accumulate=True)
        concat_bwd(dcat, (N,) + tuple(self.in_dims), c_low, self.cin - c_low, dlow, dskip, acc_low=acc_low, acc_skip=acc_skip)


class ConvTranspose3dK2S2:
    """``Conv3DTranspose(filters, (2, 2, 2), strides=(2, 2, 2))`` of the reference's 'deconv' decoder (resunet_model.py:168-174 with
    padding='valid', vnet_model.py:244-245 with 'same' -- the same thing for k = s) as a recipe over the existing entry points
    (SURVEY 8(f)4).  TF defines conv_transpose as the gradient of conv w.r.t. its input with the kernel stored [kd][kh][kw][out][in]:
    byte for byte the DHWIO kernel of a k2 s2 Conv3D from `filters` to `cin` channels, so
      forward            = that Conv3D's strided DATA GRADIENT (8 output-parity classes of one tap each in one launch) + bias,
      data gradient      = that Conv3D's FORWARD over the upstream gradient,
      kernel gradient    = that Conv3D's WEIGHT gradient with the roles exchanged (source: upstream gradient, "dy": the layer's input),
      bias gradient      = vg_bias_grad.
    store parameters: name + '.w' [2,2,2,filters,cin] (Keras layout), name + '.b' [filters].  in_dims: the layer's INPUT grid."""

    def __init__(self, store, name: str, cin: int, filters: int, in_dims: Tuple[int, int, int], dtype: torch.dtype = torch.bfloat16):
        self.cin, self.filters, self.in_dims, self.out_dims = cin, filters, tuple(in_dims), tuple(2 * n for n in in_dims)
        self.conv = ConvLayer(store, name, 2, filters, cin, 2, 'same', False, self.out_dims, need_dgrad=True, dtype=dtype)
        self.b, self.gb = store.param(name + '.b'), store.grad(name + '.b')

    def pack(self):
        self.conv.pack()

    def forward(self, x: torch.Tensor, N: int, out: torch.Tensor):
        """x [N, *in_dims, cin] -> out [N, *2 in_dims, filters] (both in the network's 16-bit / exact-parity storage type)."""
        assert tuple(x.shape) == (N,) + self.in_dims + (self.cin,) and tuple(out.shape) == (N,) + self.out_dims + (self.filters,)
        self.conv.dgrad(x, N, out, accumulate=False, bias=self.b)

    def backward(self, x: torch.Tensor, dy: torch.Tensor, N: int, dx: Optional[torch.Tensor]):
        """dy [N, *2 in_dims, filters]: adds the kernel / bias gradients to the store's gradient buffers; dx [N, *in_dims, cin] (or None)."""
        src = Src(dy, (N,) + self.out_dims, self.filters, f32=False)
        if dx is not None:
            self.conv.forward(src, dx)
        self.conv.wgrad(src, x)
        check(lib.vg_bias_grad(_p(dy), int(dy.dtype == torch.float32), N * math.prod(self.out_dims), self.filters, _p(self.gb), stream()),
              'vg_bias_grad')


class PackTable:
    """All packed operands of a network repacked by ONE kernel launch (vg_pack_weights_multi)."""

    def __init__(self, layers, device):
        items = [it for l in layers for it in l.pack_items()]
        arr = (_lib.PackItem * len(items))()
        blk = 0
        per_block = int(os.environ.get('VG_PACK_ELEMS', 8192))     # packed elements per block: big operands get many blocks
        for a, (w, idx, out, cin, cout, ntaps, tr, ck, f32, bn) in zip(arr, items):
            a.w, a.tap_idx, a.out = w.data_ptr(), idx.data_ptr(), out.data_ptr()
            a.Cin, a.Cout, a.ntaps, a.transpose, a.CK, a.out_f32, a.bn = cin, cout, ntaps, tr, ck, f32, bn
            a.blk0, a.nblk = blk, max(1, min(4096, -(-out.numel() // per_block)))
            blk += a.nblk
        self.total_blocks = blk
        raw = bytes(arr)
        self.keep = items
        self.n = len(items)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.cells = [l for l in layers if getattr(l, 'cell', None) is not None]
        self.ups = [l for l in layers if getattr(l, 'wp_up', None) is not None]

    def run(self):
        check(lib.vg_pack_weights_multi(_p(self.table), self.n, self.total_blocks, stream()), 'vg_pack_weights_multi')
        for l in self.cells:
            l.pack_cell()
        for l in self.ups:
            l.pack_up()


# ------------------------------------------------------------------------------------------------------
# InstanceNorm helpers
# ------------------------------------------------------------------------------------------------------
def in_finalize(sums0, c0, count0, gamma, beta, N, scale, shift, mean=None, rstd=None, sums1=None, c1=0, count1=1.0,
                mult=None):
    check(lib.vg_in_finalize(_p(sums0), c0, float(count0), _p(sums1), c1, float(count1), _p(gamma), _p(beta), _p(mult),
                             N, IN_EPS, _p(scale), _p(shift), _p(mean), _p(rstd), stream()), 'vg_in_finalize')


FIN_TAIL = os.environ.get('VG_FIN_TAIL', '1') != '0'     # InstanceNorm finalisation by the producing launch's last workgroup (vg_fin_desc)


def fin_desc(ar: 'Arena', count: float, jobs) -> FinDesc:
    """vg_fin_desc of a producing launch: jobs = [(gamma, beta, mult, state, c_off, c_tot)], state = {'scale', 'shift', 'mean', 'rstd'}
    tensors [N, c_tot] of the consuming norm.  The ticket word comes from the arena's zero pool (cleared once per step)."""
    f = FinDesc()
    tk = ar.alloc((1,), torch.int32, zero=True)
    f.ticket, f.count, f.eps, f.njobs = _p(tk), float(count), IN_EPS, len(jobs)
    keep = [tk]
    for j, (gamma, beta, mult, st, c_off, c_tot) in enumerate(jobs):
        q = f.job[j]
        q.gamma, q.beta, q.mult = _p(gamma), _p(beta), _p(mult)
        q.scale, q.shift, q.mean, q.rstd = _p(st['scale']), _p(st['shift']), _p(st['mean']), _p(st['rstd'])
        q.c_off, q.c_tot = c_off, c_tot
        keep += [gamma, beta, mult, st]
    f._keep = keep
    return f


def alloc_red(ar: 'Arena', N: int, C_: int) -> torch.Tensor:
    """Striped reduction buffer of the InstanceNorm backward [STRIPES][N][C][2] followed by 4 words, the first of which is
    the ticket of its last-workgroup fold (zero-initialised with the rest)."""
    return ar.alloc((STRIPES * N * C_ * 2 + 4,), torch.float32, zero=True)


def actnorm_desc(g, g_padded, x, dims, C_, dx, *, scale=None, shift=None, mult=None, act=ACT_NONE, norm=False,
                 gamma=None, mean=None, rstd=None, red=None, accumulate=False, x1=None, c_x0=0, x0_shift=0,
                 dx_cstride=0, dx_coff=0, dgamma=None, dbeta=None, alias_n0=0, alias_shift=0, pgrad_n=0) -> ActNormBwdDesc:
    """Descriptor of the (InstanceNorm -> act -> dropout) backward: statistics + apply (+ parameter gradients).  `red` from
    alloc_red() (with ticket word: one launch fewer) or a plain zeroed [STRIPES, N, C, 2] tensor.  The tensors are kept
    alive on the descriptor (it may be handed to ConvLayer.dgrad(bstat=...) before actnorm_run)."""
    d = ActNormBwdDesc()
    d.g, d.g_padded = _p(g), int(g_padded)
    d.x, d.x_f32 = _p(x), int(x is not None and x.dtype == torch.float32)
    d.x1, d.c_x0, d.x0_shift = _p(x1), c_x0, x0_shift
    d.N, d.D, d.H, d.W = dims
    d.C = C_
    d.scale, d.shift, d.mult, d.act, d.norm = _p(scale), _p(shift), _p(mult), act, int(norm)
    d.gamma, d.mean, d.rstd, d.red = _p(gamma), _p(mean), _p(rstd), _p(red)
    d.dx, d.dx_f32, d.accumulate = _p(dx), int(dx is not None and dx.dtype == torch.float32), int(accumulate)       # dx may be set later
    d.dx_cstride, d.dx_coff = dx_cstride, dx_coff
    d.f32 = int(g.dtype == torch.float32)
    d.dgamma, d.dbeta = _p(dgamma), _p(dbeta)      # parameter gradients come out of the statistics pass
    d.alias_n0, d.alias_shift, d.pgrad_n = alias_n0, alias_shift, pgrad_n
    nred = STRIPES * dims[0] * C_ * 2
    d.ticket = (red.data_ptr() + 4 * nred) if (red is not None and red.dim() == 1 and red.numel() == nred + 4) else None
    d._keep = (g, x, dx, scale, shift, mult, gamma, mean, rstd, red, x1, dgamma, dbeta)
    return d


def actnorm_stats(d: ActNormBwdDesc):
    check(lib.vg_actnorm_bwd_stats(C.byref(d), stream()), 'vg_actnorm_bwd_stats')


def stem_short_fwd(ar: 'Arena', x, N, C_, w, gamma, beta, scale, shift, round16=True):
    """The stem's shortcut (1x1x1 convolution of the single-channel volume + InstanceNorm) as the affine  scale * x + shift  per (sample,
    channel), from the mean and variance of the volume (vg_stem_short_fwd): the branch is never materialised, the block's second
    convolution adds it in its epilogue (ConvLayer.forward(res=x, res_c1=True))."""
    S = x.numel() // N
    G = int(lib.vg_stem_short_fwd_workgroups(N, S))
    if G < 1:
        raise _lib.VgError('vg_stem_short_fwd: unsupported shape N=%d S=%d' % (N, S))
    part = ar.alloc((N, G, 2), torch.float64)
    ticket = ar.alloc((4,), torch.int32, zero=True)
    check(lib.vg_stem_short_fwd(_p(x), N, S, C_, _p(w), _p(gamma), _p(beta), IN_EPS, int(round16), _p(scale), _p(shift), _p(part), G, _p(ticket),
                                stream()), 'vg_stem_short_fwd')


def stem_short_bwd(ar: 'Arena', g, x, N, C_, w, gamma, dw, dgamma=None, dbeta=None, round16=True):
    """Kernel gradient (+ gamma / beta gradients) of a single-channel 1x1x1 convolution in front of an InstanceNorm from two moments of
    the block-output gradient g [N, D, H, W, C] against the volume x [N, D, H, W, 1] itself (vg_stem_short_bwd): no apply pass, no
    gradient tensor, no weight-gradient launch; deterministic (partial table + fixed-order final sum)."""
    S = g.numel() // (N * C_)
    G = int(lib.vg_stem_short_bwd_workgroups(N, S, C_))
    if G < 1:
        raise _lib.VgError('vg_stem_short_bwd: unsupported shape N=%d S=%d C=%d' % (N, S, C_))
    part = ar.alloc((N, G, 2 * C_ + 2), torch.float64)
    ticket = ar.alloc((4,), torch.int32, zero=True)
    check(lib.vg_stem_short_bwd(_p(g), int(g.dtype == torch.float32), _p(x), N, S, C_, _p(w), _p(gamma), IN_EPS, int(round16), _p(dw), _p(dgamma),
                                _p(dbeta), _p(part), G, _p(ticket), stream()), 'vg_stem_short_bwd')


def actnorm_set_dx(d: ActNormBwdDesc, dx: torch.Tensor):
    d.dx, d.dx_f32 = _p(dx), int(dx.dtype == torch.float32)
    d._keep = d._keep + (dx,)


def actnorm_run(d: ActNormBwdDesc, stats_done: bool = False):
    """stats_done: the statistics were produced with the data gradient (ConvLayer.dgrad(bstat=d) returned True)."""
    if stats_done:
        check(lib.vg_actnorm_bwd_apply(C.byref(d), stream()), 'vg_actnorm_bwd_apply')
    else:
        check(lib.vg_actnorm_bwd(C.byref(d), stream()), 'vg_actnorm_bwd')       # statistics (when norm) + apply behind one C call


def actnorm_apply2(d1: ActNormBwdDesc, d2: ActNormBwdDesc):
    """The apply passes of two independent norms (statistics of both done) in one launch (vg_actnorm_bwd_apply2)."""
    check(lib.vg_actnorm_bwd_apply2(C.byref(d1), C.byref(d2), stream()), 'vg_actnorm_bwd_apply2')


def actnorm_bwd(g, g_padded, x, dims, C_, dx, **kw):
    """stats + apply (+ parameter gradients) of the (InstanceNorm -> act -> dropout) backward (see actnorm_desc)."""
    actnorm_run(actnorm_desc(g, g_padded, x, dims, C_, dx, **kw))


def concat_bwd(g, dims, Cu, Cs, dlow, dskip, acc_low: bool = True, acc_skip: bool = True):
    N, D, H, W = dims
    check(lib.vg_concat_bwd(_p(g), N, D, H, W, Cu, Cs, _p(dlow), _p(dskip), int(g.dtype == torch.float32),
                            int(bool(acc_low)) | (int(bool(acc_skip)) << 1), stream()), 'vg_concat_bwd')


def affine_add(a, a_scale, a_shift, a_act, b, b_scale, b_shift, N, S, C_, out):
    """out = act(a * a_scale + a_shift) + (b * b_scale + b_shift)   (vg_affine_add: the residual Add of the ResNet generator)"""
    check(lib.vg_affine_add(_p(a), _p(a_scale), _p(a_shift), a_act, _p(b), _p(b_scale), _p(b_shift), N, S, C_, _p(out),
                            int(out.dtype == torch.float32), stream()), 'vg_affine_add')


def tanh_bwd(dy, y, dpre):
    check(lib.vg_tanh_bwd(_p(dy), _p(y), _p(dpre), dy.numel(), stream()), 'vg_tanh_bwd')


# ------------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------------
def minmax(x, B, S, mm4):
    check(lib.vg_minmax(_p(x), B, S, _p(mm4), stream()), 'vg_minmax')


def minmax_apply(x, mm4, B, S, y):
    check(lib.vg_minmax_apply(_p(x), _p(mm4), B, S, _p(y), stream()), 'vg_minmax_apply')


def minmax_bwd(x, y, gy, mm4, B, S, tmp2, dx):
    check(lib.vg_minmax_bwd(_p(x), _p(y), _p(gy), _p(mm4), B, S, _p(tmp2), _p(dx), stream()), 'vg_minmax_bwd')


def bce(t, p, acc, gscale=0.0, gp=None, accumulate=False):
    check(lib.vg_bce(_p(t), _p(p), t.numel(), _p(acc), gscale, _p(gp), int(accumulate), stream()), 'vg_bce')


def mse(a, b, acc, gscale=0.0, gb=None, accumulate=False):
    check(lib.vg_mse(_p(a), _p(b), a.numel(), _p(acc), gscale, _p(gb), int(accumulate), stream()), 'vg_mse')


def mse_const(x, target, acc, gscale=0.0, gx=None, accumulate=False):
    check(lib.vg_mse_const(_p(x), int(x.dtype == torch.float32), target, x.numel(), _p(acc), gscale, _p(gx),
                           int(accumulate), stream()), 'vg_mse_const')


def ssim_fwd(t, p, dims, acc, part):
    B, D, H, W = dims
    check(lib.vg_ssim_fwd(_p(t), _p(p), B, D, H, W, _p(acc), _p(part), stream()), 'vg_ssim_fwd')


def ssim_bwd(t, p, part, dims, gscale, gp, accumulate=False):
    B, D, H, W = dims
    check(lib.vg_ssim_bwd(_p(t), _p(p), _p(part), B, D, H, W, gscale, _p(gp), int(accumulate), stream()), 'vg_ssim_bwd')


def skel_aux_bytes(dims, iters) -> int:
    """Size of the aux buffer of soft_skel_fwd / soft_skel_bwd (include/vangan_hip.h): delta (fp32) + two code bytes per voxel and step."""
    B, D, H, W = dims
    return (iters + 1) * B * D * H * W * 6


def soft_skel_fwd(img, dims, iters, imgs, skels, aux=None):
    B, D, H, W = dims
    assert aux is None or aux.numel() * aux.element_size() >= skel_aux_bytes(dims, iters)
    check(lib.vg_soft_skel_fwd(_p(img), B, D, H, W, iters, _p(imgs), _p(skels), _p(aux), stream()), 'vg_soft_skel_fwd')


def soft_skel_bwd(imgs, skels, gskel, dims, iters, work, gimg, aux=None):
    B, D, H, W = dims
    assert work.numel() >= (4 if aux is not None else 3) * B * D * H * W
    check(lib.vg_soft_skel_bwd(_p(imgs), _p(skels), _p(gskel), B, D, H, W, iters, _p(work), _p(gimg), _p(aux), stream()),
          'vg_soft_skel_bwd')


def cldice_coef(sums7, w, alpha, coef6):
    check(lib.vg_cldice_coef(_p(sums7), w, alpha, _p(coef6), stream()), 'vg_cldice_coef')


def cldice_grads(t, skel_t, coef6, gskel_p, gp, accumulate=False):
    check(lib.vg_cldice_grads(_p(t), _p(skel_t), _p(coef6), t.numel(), _p(gskel_p), _p(gp), int(accumulate), stream()),
          'vg_cldice_grads')


def dot_sums(a, b, sums3):
    check(lib.vg_dot_sums(_p(a), _p(b), a.numel(), _p(sums3), stream()), 'vg_dot_sums')


def dense_head_fwd(x, mask, w, b, N, n, z):
    check(lib.vg_dense_head_fwd(_p(x), _p(mask), _p(w), _p(b), N, n, _p(z), stream()), 'vg_dense_head_fwd')


def dense_head_bwd(x, mask, w, gz, N, n, dx=None, dw=None, db=None):
    check(lib.vg_dense_head_bwd(_p(x), _p(mask), _p(w), _p(gz), N, n, _p(dx), _p(dw), _p(db), stream()), 'vg_dense_head_bwd')


def wasserstein_terms(z, B, inv, acc2, gz_d=None, gz_g=None):
    check(lib.vg_wasserstein_terms(_p(z), B, inv, _p(acc2), _p(gz_d), _p(gz_g), stream()), 'vg_wasserstein_terms')


def axpby(a, alpha, b, beta, y, accumulate=False):
    check(lib.vg_axpby(_p(a), alpha, _p(b), beta, a.numel(), _p(y), int(accumulate), stream()), 'vg_axpby')


def adam_clip(w, g, m, v, seg_off, T, norms, lr_t, beta1, beta2, eps, clipnorm, grad_scale=1.0):
    assert norms.numel() >= T + 2 * ((w.numel() + 4095) // 4096), 'norms: T squared norms + 2 partial sums per 4096-element block'
    check(lib.vg_adam_clip(_p(w), _p(g), _p(m), _p(v), _p(seg_off), T, w.numel(), _p(norms), lr_t, beta1, beta2, eps,
                           clipnorm, grad_scale, stream()), 'vg_adam_clip')


def adam_clip_dev(w, g, m, v, seg_off, T, norms, lr_t_dev: int, beta1, beta2, eps, clipnorm, grad_scale=1.0):
    """adam_clip with the bias-corrected rate read from device memory (lr_t_dev: device address of one float): replayable graphs."""
    check(lib.vg_adam_clip_dev(_p(w), _p(g), _p(m), _p(v), _p(seg_off), T, w.numel(), _p(norms), lr_t_dev, beta1, beta2, eps,
                               clipnorm, grad_scale, stream()), 'vg_adam_clip_dev')


def randn_bf16_dev(out, std_dev: int, seed, offset_dev: int, offset_add: int):
    check(lib.vg_randn_bf16_dev(_p(out), out.numel(), std_dev, seed, offset_dev, offset_add, stream()), 'vg_randn_bf16_dev')


def dropout_mask_dev(out, rate, seed, offset_dev: int, offset_add: int):
    check(lib.vg_dropout_mask_dev(_p(out), out.numel(), rate, seed, offset_dev, offset_add, stream()), 'vg_dropout_mask_dev')


def randn_bf16(out, std, seed, offset):
    check(lib.vg_randn_bf16(_p(out), out.numel(), std, seed, offset, stream()), 'vg_randn_bf16')


def dropout_mask(out, rate, seed, offset):
    check(lib.vg_dropout_mask(_p(out), out.numel(), rate, seed, offset, stream()), 'vg_dropout_mask')
