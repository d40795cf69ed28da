"""Data-parallel synchronisation of the VAN-GAN step: one process per GPU, RCCL (torch.distributed 'nccl') SUM
all-reduce of the four flat gradient buckets on a side stream, SUM of the 10 result scalars, rank-0 weight broadcast.

Counterpart of tf.distribute.MirroredStrategy in the reference (main.py:22; implicit all-reduce inside
optimizer.minimize, vangan.py:426-438; strategy.reduce(SUM) of the result dict, vangan.py:472-473).  Losses are already
pre-divided by the GLOBAL batch size on every replica (loss_functions.py:21-22) and the clDice term by n_devices
(loss_functions.py:226), so the reduction is a plain SUM and gradients are NOT averaged afterwards.

Device-agnostic on purpose: the same class runs over gloo on CPU tensors (tests/test_ddp_gloo.py)."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

RING_WORLD = 8          # the node size the stand-in's hold time is priced for (BASELINE config 4: 8 x MI355X)


class GradSync:
    """`fake` (default: VG_FAKE_AR=1, only with one process on a GPU): the data-parallel schedule runs for real with world == 1 --
    communication stream, per-bucket events, pieces, the engine's cross-step overlap -- and every all-reduce is replaced by
    vg_local_exchange (include/vangan_hip.h): a device-local kernel ON THE COMMUNICATION STREAM (torch >= 2.7 launches a
    synchronous collective -- async_op=False, what start() issues -- on the caller's current stream; ProcessGroupNCCL's internal
    stream only serves async_op=True) that moves the bucket's bytes with VG_FAKE_AR_WG workgroups and holds them for the time a
    ring over RING_WORLD GPUs would need at VG_FAKE_AR_GBPS GB/s of bus bandwidth (0: no hold).  VG_FAKE_AR_INNER=1 puts the
    kernel on a second stream behind the communication stream instead (older torch); VG_FAKE_AR_NULL=1 launches nothing (the
    schedule alone: events, waits, cross-step mode).
    `stream`: run the communication on this stream instead of a new one (the engine passes its optimizer stream: the bucket's
    optimizer step is the only consumer, and HIP has 4 hardware queues for the streams of a process)."""

    def __init__(self, buckets: Dict[str, torch.Tensor], process_group=None, weights: Optional[Dict[str, torch.Tensor]] = None,
                 fake: Optional[bool] = None, stream=None):
        self.buckets, self.weights, self.pg = buckets, weights, process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        dev = next(iter(buckets.values())).device
        self.cuda = dev.type == 'cuda'
        if fake is None:
            fake = os.environ.get('VG_FAKE_AR', '0') == '1'
        self.fake = bool(fake) and self.world == 1 and self.cuda
        # VG_DDP_FORCE=1: a process group of ONE rank still runs every collective (what a 1-GPU box can execute of the RCCL path:
        # communicator set-up, ncclAllReduce / ncclBroadcast calls on the engine's streams -- tests/test_gpu_ddp.py)
        self.forced = process_group is not None and self.world == 1 and not self.fake and os.environ.get('VG_DDP_FORCE', '0') == '1'
        self.active = self.world > 1 or self.fake or self.forced
        self.stream = (stream if stream is not None else torch.cuda.Stream(device=dev)) if (self.cuda and self.active) else None
        self.pending: Dict[str, list] = {}          # bucket name -> CUDA events on the comm stream / async work handles (one per piece)
        self.rank = dist.get_rank(process_group) if process_group is not None else 0
        self.moved_bytes = 0                        # fake mode: bytes handed to the stand-in so far
        self._inner = self._scratch = None
        if self.fake:
            self._wg = int(os.environ.get('VG_FAKE_AR_WG', '32'))
            self._gbps = float(os.environ.get('VG_FAKE_AR_GBPS', '0'))
            self._scratch = torch.empty(max(b.numel() for b in buckets.values()), dtype=torch.float32, device=dev)

    def _reduce(self, t: torch.Tensor):
        """SUM all-reduce of a flat fp32 piece on the current (communication) stream."""
        if not self.fake:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            return
        from . import _lib
        # only whole 16-byte units strictly inside the piece: the elements next to it may still be written by the backward sweep
        a = t.data_ptr()
        lo = (-(a // 4)) % 4
        n = (t.numel() - lo) // 4 * 4
        if n <= 0:
            return
        nbytes = 4 * n
        self.moved_bytes += nbytes
        if os.environ.get('VG_FAKE_AR_NULL', '0') == '1':
            return
        cur = torch.cuda.current_stream()
        on = cur
        if os.environ.get('VG_FAKE_AR_INNER', '0') == '1':
            if self._inner is None:
                self._inner = torch.cuda.Stream(device=t.device)
            on = self._inner
            on.wait_stream(cur)
        hold_us = int(nbytes * 2.0 * (RING_WORLD - 1) / RING_WORLD / (self._gbps * 1e3)) if self._gbps > 0 else 0
        _lib.check(_lib.lib.vg_local_exchange(a + 4 * lo, self._scratch.data_ptr(), n, self._wg, min(hold_us, 100000),
                                              on.cuda_stream), 'vg_local_exchange')
        if on is not cur:
            cur.wait_stream(on)

    def start(self, names: Iterable[str], also=None, lo: int = 0, hi: Optional[int] = None):
        """Issue the all-reduce of these buckets; on GPU it runs on the side stream behind everything already queued
        on the current stream (and behind the event `also`: the weight gradients a lane handed to ITS side stream), so the
        remaining backward sweeps overlap with it.
        lo / hi: only the elements [lo, hi) of the flat bucket (one name).  A backward sweep completes a generator's gradients from
        the END of the flat buffer towards its start (output head, decoder, bridge, encoder, stem -- the reverse of the parameter
        order), so the engine reduces a finished suffix while the sweep is still running; finish(name) waits for every piece."""
        if not self.active:
            return
        names = list(names)
        assert (lo == 0 and hi is None) or len(names) == 1
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            self.stream.wait_event(ev)
            if also is not None:
                self.stream.wait_event(also)
            with torch.cuda.stream(self.stream):
                for n in names:
                    self._reduce(self.buckets[n][lo:hi])
                    done = torch.cuda.Event()
                    done.record()
                    self.pending.setdefault(n, []).append(done)
        else:
            for n in names:
                self.pending.setdefault(n, []).append(dist.all_reduce(self.buckets[n][lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self, names: Optional[Iterable[str]] = None):
        """The current stream (GPU) / the host (CPU tensors) waits until the reduced buckets `names` (default: all that
        are in flight) have landed.  Per bucket, so that a network's optimizer step can run as soon as ITS bucket is
        there while the other networks' backward sweeps and all-reduces are still going."""
        if not self.active:
            return
        for n in (list(self.pending) if names is None else list(names)):
            for h in self.pending.pop(n, []):
                if self.stream is not None:
                    torch.cuda.current_stream().wait_event(h)
                else:
                    h.wait()

    def reduce_dict(self, d: Dict[str, float], keys: List[str]) -> Dict[str, float]:
        """vangan.py:472-473: strategy.reduce(SUM) of every result scalar."""
        if self.world == 1 and not self.forced:
            return d
        dev = next(iter(self.buckets.values())).device
        t = torch.tensor([d[k] for k in keys], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return dict(zip(keys, t.cpu().tolist()))

    def broadcast_weights(self, src: int = 0):
        if (self.world == 1 and not self.forced) or self.weights is None:
            return
        for w in self.weights.values():
            dist.broadcast(w, src=src, group=self.pg)
