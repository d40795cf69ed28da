"""Data-parallel synchronisation of the VAN-GAN step: one process per GPU, RCCL (torch.distributed 'nccl') SUM
all-reduce of the four flat gradient buckets on a side stream, SUM of the 10 result scalars, rank-0 weight broadcast.

Counterpart of tf.distribute.MirroredStrategy in the reference (main.py:22; implicit all-reduce inside
optimizer.minimize, vangan.py:426-438; strategy.reduce(SUM) of the result dict, vangan.py:472-473).  Losses are already
pre-divided by the GLOBAL batch size on every replica (loss_functions.py:21-22) and the clDice term by n_devices
(loss_functions.py:226), so the reduction is a plain SUM and gradients are NOT averaged afterwards.

Device-agnostic on purpose: the same class runs over gloo on CPU tensors (tests/test_ddp_gloo.py)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, buckets: Dict[str, torch.Tensor], process_group=None, weights: Optional[Dict[str, torch.Tensor]] = None):
        self.buckets, self.weights, self.pg = buckets, weights, process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        dev = next(iter(buckets.values())).device
        self.cuda = dev.type == 'cuda'
        self.stream = torch.cuda.Stream(device=dev) if (self.cuda and self.world > 1) else None
        self.pending: Dict[str, list] = {}          # bucket name -> CUDA events on the comm stream / async work handles (one per piece)
        self.rank = dist.get_rank(process_group) if process_group is not None else 0

    def start(self, names: Iterable[str], also=None, lo: int = 0, hi: Optional[int] = None):
        """Issue the all-reduce of these buckets; on GPU it runs on the side stream behind everything already queued
        on the current stream (and behind the event `also`: the weight gradients a lane handed to ITS side stream), so the
        remaining backward sweeps overlap with it.
        lo / hi: only the elements [lo, hi) of the flat bucket (one name).  A backward sweep completes a generator's gradients from
        the END of the flat buffer towards its start (output head, decoder, bridge, encoder, stem -- the reverse of the parameter
        order), so the engine reduces a finished suffix while the sweep is still running; finish(name) waits for every piece."""
        if self.world == 1:
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
                    dist.all_reduce(self.buckets[n][lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
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
        if self.world == 1:
            return
        for n in (list(self.pending) if names is None else list(names)):
            for h in self.pending.pop(n, []):
                if self.stream is not None:
                    torch.cuda.current_stream().wait_event(h)
                else:
                    h.wait()

    def reduce_dict(self, d: Dict[str, float], keys: List[str]) -> Dict[str, float]:
        """vangan.py:472-473: strategy.reduce(SUM) of every result scalar."""
        if self.world == 1:
            return d
        dev = next(iter(self.buckets.values())).device
        t = torch.tensor([d[k] for k in keys], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return dict(zip(keys, t.cpu().tolist()))

    def broadcast_weights(self, src: int = 0):
        if self.world == 1 or self.weights is None:
            return
        for w in self.weights.values():
            dist.broadcast(w, src=src, group=self.pg)
