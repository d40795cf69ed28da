"""Sliding-window generator inference on MI355X: the counterpart of GanMonitor.stitch_subvolumes
(custom_callback.py:47-223) / run_mapping (:466-509) / post_training.epoch_sweep (post_training.py:22-39) without the
TIFF I/O.  Windows are batched through the HIP generator; overlap-add, coverage count, division, un-padding and the
final 255*min-max run on the GPU."""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch

from . import ops
from ._lib import check, lib
from .ops import _p, stream


def window_origins(n: int, k: int, s: int):
    """The reference's loop (custom_callback.py:140-190): dim_out+1 iterations, start += stride, clamped to n-k (so the
    last windows may coincide - they are then counted twice, as in the reference)."""
    dim_out = int(math.floor((n - k) / s + 1))
    out, start = [], 0
    for _ in range(dim_out + 1):
        if start > n - k:
            start = n - k
        out.append(start)
        start += s
    return out


def stitch_subvolumes(engine, gen: str, img: torch.Tensor, subvol_size: Sequence[int], stride=(25, 25, 128), complete=True,
                      padFactor: float = 0.25, border_removal: bool = True, process_img: bool = False,
                      window_batch: int = 4, precision: str = None) -> torch.Tensor:
    """img: [X,Y,Z,1] fp32 (host or device).  gen: 'gen_IS' or 'gen_SI'.  Returns 255*minmax(pred) as fp32 [X,Y,Z,1]
    on the device (custom_callback.py:202).  subvol_size is (kX,kY,kZ).
    precision: None = the engine's training precision (bf16 / fp32 storage); 'fp16' = IEEE half-precision storage with fp32
    accumulation (BASELINE config 5; the reference's inference runs whatever policy TF was given, post_training.py:38-39): the
    generator's forward runs in libvangan_hip_h.so (van_gan_amd.ops.Fp16) on weights repacked to fp16 at the start of the call."""
    import contextlib
    dev = engine.device
    ops.set_device(dev.index)
    if precision not in (None, 'fp16', engine.precision):
        raise ValueError("precision must be None, 'fp16' or the engine's own precision")
    half = precision == 'fp16'
    net = engine.fp16_generator(gen) if half else engine.nets[gen]
    kx, ky, kz = subvol_size
    if tuple(net.dims) != (kx, ky, kz):
        raise ValueError('generator was built for windows %s' % (net.dims,))
    v = img.to(dev, torch.float32)[..., 0]
    ox, oy, oz = v.shape
    sx = sy = sz = 0
    if complete:                       # np.pad(..., 'symmetric'): mirror INCLUDING the edge voxel (host-side indexing)
        sx, sy = int(padFactor * ox), int(padFactor * oy)
        sz = 0 if stride[2] == 1 else int(padFactor * oz)

        def sym(n, p):
            idx = torch.arange(-p, n + p, device=dev)
            idx = torch.where(idx < 0, -idx - 1, idx)
            return torch.where(idx >= n, 2 * n - 1 - idx, idx)
        v = v[sym(ox, sx)][:, sym(oy, sy)][:, :, sym(oz, sz)].contiguous()
    X, Y, Z = v.shape
    if not complete or not border_removal:
        px = py = pz = 0
    else:
        px, py, pz = int(0.1 * kx), int(0.1 * ky), int(0.1 * kz)
        if kz == Z:
            pz = 0
    pred = torch.zeros(X, Y, Z, device=dev)
    cnt = torch.zeros(X, Y, Z, device=dev)
    origins = [(a, b, c) for a in window_origins(X, kx, stride[0]) for b in window_origins(Y, ky, stride[1])
               for c in window_origins(Z, kz, stride[2])]
    S = kx * ky * kz
    # two lanes: consecutive window batches alternate between two streams, each with its own workspace, so that the
    # low-occupancy deep layers of one batch overlap the full-resolution layers of the other
    main = torch.cuda.current_stream()
    lane_b = getattr(engine, '_lane_b', None)
    lanes = [(main, engine.arena)]
    if lane_b is not None and getattr(engine, 'arena_b', None) is not None:
        lane_b.wait_stream(main)                             # v, pred, cnt are ready
        lanes.append((lane_b, engine.arena_b))
    for bi, i0 in enumerate(range(0, len(origins), window_batch)):
        chunk = origins[i0:i0 + window_batch]
        B = len(chunk)
        strm, ar = lanes[bi % len(lanes)]
        with torch.cuda.stream(strm):
            ar.reset()
            xin = ar.alloc((B, kx, ky, kz, 1), torch.float32)
            yout = ar.alloc((B, kx, ky, kz, 1), torch.float32)
            for b, (a, bb, c) in enumerate(chunk):
                xin[b, ..., 0].copy_(v[a:a + kx, bb:bb + ky, c:c + kz])
            if process_img:            # process_imaging_otf with axis=None (main.py:169-177): per-window min-max to [-1,1]
                mm = ar.alloc((B, 4), torch.float32)
                tmp = ar.alloc((B, kx, ky, kz, 1), torch.float32)
                ops.minmax(xin, B, S, mm)
                ops.minmax_apply(xin, mm, B, S, tmp)
                ones = ar.alloc((B, kx, ky, kz, 1), torch.float32)
                ones.fill_(1.0)                              # memset-style fill (plumbing)
                ops.axpby(tmp, 2.0, ones, -1.0, xin)         # 2*n - 1
            with (ops.Fp16() if half else contextlib.nullcontext()):
                net.forward(ar, xin, yout, save=False)
            for b, (a, bb, c) in enumerate(chunk):
                check(lib.vg_overlap_add(_p(yout[b]), kx, ky, kz, px, py, pz, a, bb, c, X, Y, Z, _p(pred), _p(cnt), stream()),
                      'vg_overlap_add')
    if len(lanes) > 1:
        main.wait_stream(lane_b)
    out = torch.zeros(ox, oy, oz, device=dev)
    check(lib.vg_divide_crop(_p(pred), _p(cnt), X, Y, Z, sx, sy, sz, ox, oy, oz, _p(out), stream()), 'vg_divide_crop')
    mm = torch.zeros(1, 4, device=dev)
    nrm = torch.zeros_like(out)
    ops.minmax(out, 1, out.numel(), mm)
    ops.minmax_apply(out, mm, 1, out.numel(), nrm)
    res = torch.zeros_like(out)
    ops.axpby(nrm, 255.0, None, 0.0, res)
    return res[..., None]
