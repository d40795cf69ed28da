"""VanGan training engine for MI355X: the host-side mirror of the reference's ``VanGan`` class
(vangan.py:20-550) for the default non-Wasserstein ResUNet path, scheduling libvangan_hip.so kernels.

Same surface as the reference for the hot path: ``train_step(real_I, real_S) -> dict`` with the 10 result keys
of vangan.py:338-351, ``test_step``, ``distributed_train_step`` (one process per GPU, RCCL SUM all-reduce of the
four flat gradient buckets - the counterpart of MirroredStrategy's implicit all-reduce inside
optimizer.minimize, vangan.py:426-438,475-490), ``save_checkpoint`` / ``load_checkpoint``, and the public
attributes ``gen_IS, gen_SI, disc_I, disc_S, layer_noise, current_epoch``.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from . import ops, roctx
from .dist import GradSync
from .nets import (pair_ctx, PatchGAN, ParamStore, ResNetGenerator, ResUNet, disc_param_specs, gen_param_specs, init_reference,
                   resnet_param_specs)
from .ops import Arena

RESULT_KEYS = ['total_IS_loss', 'total_SI_loss', 'D_I_loss', 'D_S_loss', 'gen_IS_loss', 'gen_SI_loss',
               'cycle_gen_SIS_loss', 'cycle_gen_ISI_loss', 'seg_loss', 'reconstruction_loss_I']
_BFIRST = int(os.environ.get('VG_BFIRST', '5'))       # re-swept with the paired sweeps: 0: 22.27, 4: 21.98, 5: 21.83 ms
_NOJOIN = os.environ.get('VG_NOJOIN', '1') != '0'
_LAZY_AR = os.environ.get('VG_LAZY_AR', '1') != '0'
_INLINE_ENV = os.environ.get('VG_WGRAD_INLINE')
_INLINE = int(_INLINE_ENV) if _INLINE_ENV is not None else 0      # encoder blocks <= this and the stem; round 3 (DMA weight gradients): off 24.96, 1: 24.92, 2: 24.74, 3: 24.59, 4: 24.62 ms; round 4, final kernels, alternating: 2: 18.85 / 18.88 / 18.87, 3: 18.99 / 18.91 / 19.01; round 5 (wgrad_thin: the side streams are no longer behind at the end of a sweep), same box: -1: 17.78 / 17.79 / 17.94, 0: 17.84 / 17.89 / 17.89, 1: 17.82 / 17.92 / 17.96, 2: 17.93 / 17.97 / 18.08, 3: 17.95 / 18.00, 4: 18.07 / 18.08; the data-parallel schedule (all-reduce of a bucket behind the side stream's weight gradients) wants 1: 18.08 / 18.08 against 18.21 / 18.26 for 0 and 18.09 / 18.11 for 2 -- VanGan.__init__ picks 1 when it synchronises gradients
_PAIR_BWD = os.environ.get('VG_PAIR_BWD', '1') != '0'       # one 2B-sample backward sweep per generator (both applications) instead of two
_AR_SPLIT = os.environ.get('VG_AR_SPLIT', '1') != '0'       # world > 1: a generator's finished gradient suffix is all-reduced while its sweep still runs
_SKEL_BWD_A = int(os.environ.get('VG_SKEL_BWD_A', '0'))      # clDice backward on lane A: 1 before its discriminator sweeps, 2 right before its generator sweep
_D_ONE_SWEEP = os.environ.get('VG_D_ONE_SWEEP', '1') != '0'    # one 3B-sample backward sweep per discriminator (PatchGAN.backward_both) instead of a 2B and a B sweep
_SKEL_FWD_A = os.environ.get('VG_SKEL_FWD_A', '0') != '0'    # the predicted skeleton's forward pass on lane A (behind lane B's min-max of cycled_S)
_SKEL_AUX = os.environ.get('VG_SKEL_AUX', '0') != '0'        # clDice backward from codes filed by the forward pass (streaming launches) instead of re-scanning
_EARLY_ADAM_DDP = os.environ.get('VG_EARLY_ADAM_DDP', '1') != '0'     # ... under gradient synchronisation too, behind the suffix's early all-reduce
_EARLY_ADAM = os.environ.get('VG_EARLY_ADAM', '1') != '0'     # a generator's finished parameter suffix (enc4 ... output head, 91 %) is updated and repacked while its sweep still runs
_INTERLEAVE = os.environ.get('VG_INTERLEAVE', '0') != '0'   # the two lanes' enqueue sequences alternate block by block on the host: measured neutral (21.69 vs 21.63 ms), off


def interleave(*seqs, on=True):
    """seqs: (stream context factory, generator).  Steps the generators alternately, each inside its stream context, until all
    are exhausted; returns their values.  on=False: one after the other."""
    vals = [None] * len(seqs)
    live = list(range(len(seqs)))
    while live:
        for i in list(live):
            ctxf, gen = seqs[i]
            with ctxf():
                try:
                    while True:
                        next(gen)
                        if on:
                            break
                except StopIteration as e:
                    vals[i] = e.value
                    live.remove(i)
    return vals
NETS = ['gen_IS', 'gen_SI', 'disc_I', 'disc_S']
# roctx range (VG_ROCTX=1) opened when a milestone of _mark() has been enqueued: the name of what the host enqueues NEXT
_PHASE_AFTER = {'A start': 'enqueue: generators, first application (both lanes)', 'A G1 fwd': 'enqueue: lane B event', 'B G1 fwd': 'enqueue: generators, cycle application',
                'A G2 fwd': 'enqueue: lane B mark', 'B G2 fwd': 'enqueue: target skeleton', 'A target skeleton': 'enqueue: lane B BCE + clDice (+ backward)',
                'B clDice': 'enqueue: lane A MSE + SSIM (+ backward)', 'A cycle losses': 'enqueue: D_S forward + LSGAN', 'A D_S fwd': 'enqueue: D_I forward + LSGAN',
                'A D fwd': 'enqueue: discriminator backward sweeps', 'A D bwd': 'enqueue: next sweep', 'B D bwd': 'enqueue: next sweep',
                'A G adv bwd': 'enqueue: next sweep', 'B G adv bwd': 'enqueue: next sweep', 'A G cyc bwd': 'enqueue: all-reduce + optimizer (gen_IS)',
                'B G cyc bwd': 'enqueue: all-reduce + optimizer (gen_SI)', 'A all joined': None}


_ENGINE_STREAMS: Dict = {}


def _engine_stream(device, role: str):
    """The lane / optimizer stream of every engine of this process on `device`.  One stream object per role, shared: HIP maps streams
    onto a few hardware queues in creation order, and a second engine's fresh streams landed on queues its other lane already used --
    its two lanes then ran one after the other (64^3 batch 2: 11.4 ms per step behind a 128^3 engine in the same process, 9.45 ms in a
    fresh one).  Engines that share the streams are ordered against each other by them, which is what sequential use needs."""
    key = (device.index, role)
    s = _ENGINE_STREAMS.get(key)
    if s is None:
        s = _ENGINE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


class _StepParams:
    """Device block of the per-step host scalars of a captured train step (include/vangan_hip.h: the *_dev entry points): bytes
    [0, 8) the Philox counter base, [8, 12) the discriminator noise standard deviation, [16, 32) lr_t of the four networks.  It is
    rewritten before every replay by a one-thread launch that carries the values as kernel ARGUMENTS (vg_set_step_params): bound at
    enqueue time, so a host that runs ahead of the device (sync=False) cannot hand step N the scalars of step N+1 -- a pinned mirror
    + asynchronous copy could."""

    def __init__(self, device):
        self.dev = torch.zeros(32, dtype=torch.uint8, device=device)
        self.base = 0                   # Philox counter at the start of the captured step: launches carry offsets relative to it
        self.per_step = 0               # counter advance of one step
        p = self.dev.data_ptr()
        self.off_ptr, self.std_ptr = p, p + 8

    def lr_ptr(self, name: str) -> int:
        return self.dev.data_ptr() + 16 + 4 * NETS.index(name)

    def refresh(self, offset: int, std: float, lr_t):
        """On torch's current stream (the stream the replay / graph launch is issued on, in front of it)."""
        lr = [float(v) for v in lr_t]
        ops.check(ops._lib.lib.vg_set_step_params(self.dev.data_ptr(), int(offset), float(std), lr[0], lr[1], lr[2], lr[3], ops.stream()),
                  'vg_set_step_params')


class VanGan:
    def __init__(self, subvol_patch_size=(128, 128, 128), batch_size: int = 1, global_batch_size: Optional[int] = None,
                 n_devices: int = 1, device: str = 'cuda:0', seed: int = 0, lambda_cycle: float = 10.0,
                 lambda_reconstruction: float = 5.0, lambda_topology: float = 5.0, lr: float = 2e-4,
                 beta_1: float = 0.5, beta_2: float = 0.9, clipnorm: float = 100.0, layer_noise: float = 0.1,
                 dropout_rate: float = 0.2, skel_iters: int = 15, output_dir: Optional[str] = None,
                 process_group=None, arena_bytes: Optional[int] = None, precision: str = 'bf16', generator: str = 'resUnet',
                 wasserstein: bool = False):
        """wasserstein=True: what the reference's wasserstein=True trains once its step is traced: Wasserstein critic /
        generator losses (loss_functions.py:325-355), discriminators with the Flatten -> Dropout(0.2) -> Dense(1) head
        (discriminator.py:116-119), generators updated every step, no gradient penalty (the reference's never reaches a weight).  Pass the
        optimizers of vangan.py:195-203 explicitly: lr=1e-4, beta_1=0.0, beta_2=0.9, clipnorm=0 (compat.VanGan does)."""
        if not torch.cuda.is_available():
            raise RuntimeError('VanGan engine needs an MI355X (HIP device); there is no CPU fallback')
        if precision not in ('bf16', 'fp32'):
            raise ValueError("precision must be 'bf16' (product path) or 'fp32' (exact-parity mode)")
        if generator not in ('resUnet', 'resnet'):
            raise ValueError("generator must be 'resUnet' (default, vangan.py:113-123) or 'resnet' (vangan.py:88-97)")
        self.generator = generator                # both generators of one engine have the same architecture
        self.wasserstein = bool(wasserstein)
        self.precision = precision
        self.dtype = torch.bfloat16 if precision == 'bf16' else torch.float32
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        ops.set_device(self.device.index)
        self.dims = tuple(subvol_patch_size)
        self.batch_size = batch_size
        self.n_devices = n_devices
        self.global_batch_size = global_batch_size if global_batch_size is not None else batch_size * n_devices
        self.lambda_cycle, self.lambda_reconstruction, self.lambda_topology = lambda_cycle, lambda_reconstruction, lambda_topology
        self.lr, self.beta_1, self.beta_2, self.clipnorm = lr, beta_1, beta_2, clipnorm
        # per-network learning rate: None = self.lr; a float; or a schedule callable(step) evaluated at the network's optimizer
        # iteration count when its Adam step is enqueued (the reference assigns a PolynomialDecay to each optimizer's .lr,
        # custom_callback.py:343-365; Keras evaluates it at optimizer.iterations)
        self.lrs: Dict[str, object] = {}
        self.adam_eps = 1e-7                      # TP: tf.keras Adam default epsilon
        self.layer_noise = layer_noise            # vangan.py:77 (GanMonitor decays it per epoch)
        self.dropout_rate = dropout_rate
        self.skel_iters = skel_iters
        self.current_epoch = 0
        self.checkpoint_loaded = False            # vangan.py:77 (set by the caller after load_checkpoint; read by GanMonitor)
        self.pg = process_group
        self.seed = seed
        # Philox keys of the discriminator noise / dropout streams.  `seed` controls the initialisation (identical on every
        # replica, and rank 0 broadcasts anyway); the stochastic layers draw INDEPENDENT streams per replica, as the
        # MirroredStrategy replicas of the reference do (discriminator.py:52,108): the rank is folded into the key.
        import torch.distributed as _dist
        self.rank = _dist.get_rank(process_group) if process_group is not None else 0
        self.noise_key = seed + 7919 + self.rank * 1000003
        self.drop_key = seed + 104729 + self.rank * 1000003
        self.rng_offset = 0                       # Philox counter; persisted in the checkpoint
        self.stores: Dict[str, ParamStore] = {}
        for i, name in enumerate(NETS):
            gspecs = gen_param_specs() if generator == 'resUnet' else resnet_param_specs()
            n_patch = (self.dims[0] // 8) * (self.dims[1] // 8) * (self.dims[2] // 8) if self.wasserstein else 0
            st = ParamStore(gspecs if name.startswith('gen') else disc_param_specs(n_patch), self.device)
            init_reference(st, seed + i)
            self.stores[name] = st
        GenNet = ResUNet if generator == 'resUnet' else ResNetGenerator
        self.gen_IS = GenNet(self.stores['gen_IS'], self.dims, self.dtype)
        self.gen_SI = GenNet(self.stores['gen_SI'], self.dims, self.dtype)
        self.disc_I = PatchGAN(self.stores['disc_I'], self.dims, self.dtype)
        self.disc_S = PatchGAN(self.stores['disc_S'], self.dims, self.dtype)
        self.nets = {'gen_IS': self.gen_IS, 'gen_SI': self.gen_SI, 'disc_I': self.disc_I, 'disc_S': self.disc_S}
        S = self.dims[0] * self.dims[1] * self.dims[2]
        if arena_bytes is None:
            # bytes per voxel: measured peak of one train step with the deferred release of backward temporaries (ops.Arena.release)
            arena_bytes = int(batch_size * S * (8000 if ops.LAZY_RELEASE else 5200) * (2 if precision == 'fp32' else 1)) + (512 << 20)
        self.arena = Arena(arena_bytes, self.device)
        # weight gradients go to a side stream of the stream that issues them (one per lane): within a layer they are
        # independent of the data-gradient chain.  36.9 vs 37.6 ms/step; a single side stream shared by both lanes cost 2 ms.
        ops.side_enable(self.device, os.environ.get('VG_SIDE_STREAM', '1') != '0')
        # Streams, in a DELIBERATE creation order (VG_STREAM_ORDER): HIP serves the streams of a process from GPU_MAX_HW_QUEUES (4)
        # hardware queues handed out in creation order, and streams that share a queue run one after the other.
        # Roles: lane_b (the second forward / backward lane; the first is the caller's current stream), opt (optimizer stream),
        # side_a / side_b (weight-gradient side streams of the two lanes), comm (gradient all-reduce, data-parallel runs only).  The
        # first four are the single-GPU set; the communication of a data-parallel run comes LAST so that it never displaces one of
        # them -- and by default it is not a stream of its own at all: the all-reduce of a bucket is queued on the optimizer stream
        # (VG_COMM_ON_OPT=0: own stream), in front of the optimizer step that is its only consumer; RCCL's internal stream (created by
        # the first collective, the rank-0 weight broadcast) then is the fifth and only extra one.
        lanes = os.environ.get('VG_LANES', '1') != '0'
        self._lane_b = self._opt = None
        self.ddp = self.pg is not None or (os.environ.get('VG_FAKE_AR', '0') == '1')
        comm = None
        for role in os.environ.get('VG_STREAM_ORDER', 'lane_b,opt,side_b,side_a,comm').split(','):
            if role == 'lane_b' and lanes:
                # forward lanes: the I->S->I chain (G_IS(real_I), G_SI(fake_S), D_S, cycle losses on cycled_I) and the S->I->S chain
                # are independent until the backward sweeps, so they run on two streams and fill each other's low-occupancy layers
                self._lane_b = _engine_stream(self.device, 'lane_b')
            elif role == 'opt' and os.environ.get('VG_OPT_STREAM', '1') != '0':
                # optimizer stream: a network's clip + Adam + weight repack is queued here as soon as ITS backward sweeps are issued
                # and waits only for ITS gradient bucket (all-reduce event), while the other networks' backward sweeps still run
                self._opt = _engine_stream(self.device, 'opt')
            elif role == 'side_a' and ops.SIDE is not None:
                ops._side_of(torch.cuda.current_stream(self.device))
            elif role == 'side_b' and ops.SIDE is not None and self._lane_b is not None:
                ops._side_of(self._lane_b)
            elif role == 'comm' and self.ddp and not (self._opt is not None and os.environ.get('VG_COMM_ON_OPT', '1') != '0'):
                comm = _engine_stream(self.device, 'comm')
        if self.ddp and comm is None:
            comm = self._opt                 # None (VG_OPT_STREAM=0): GradSync makes its own
        # second workspace for lane B's backward temporaries (bump allocators cannot interleave mark/release)
        self.arena_b = Arena(arena_bytes // 2, self.device) if self._lane_b is not None else None
        # one arena for every backward sweep (VG_LANES=0): the workspace is sized for the two-lane layout, so backward temporaries
        # are recycled there (joining release) instead of being kept until the next reset
        self.arena.lazy_ok = self._lane_b is not None
        self.sync = GradSync({k: s.g for k, s in self.stores.items()}, self.pg, {k: s.w for k, s in self.stores.items()}, stream=comm)
        self.ddp = self.sync.active
        self._inline = _INLINE if (_INLINE_ENV is not None or not self.ddp) else 1      # (see _INLINE)
        self._tl = [] if os.environ.get('VG_TIMELINE') == '1' else None
        self._side_ev = {}
        self._early_tab, self._early_done = {}, {}
        # world > 1: the step does not end with a join of the optimizer stream -- the last buckets' all-reduce + Adam + repack run
        # under the head of the NEXT step, whose consumers wait for the update event of the network they read (VG_XSTEP=0: join)
        self._xstep = (self.ddp or os.environ.get('VG_XSTEP_SINGLE', '0') == '1') and self._opt is not None and os.environ.get('VG_XSTEP', '1') != '0'
        self._upd_ev = {}
        self._cap = None                 # _StepParams while a train step is being captured into a HIP graph (capture_train_step)
        self._graph = None
        self._fp16_nets = {}
        self.checkpoint_dir = None
        if output_dir is not None:
            self.checkpoint_dir = os.path.join(output_dir, 'checkpoints')
            os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.repack()

    # ------------------------------------------------------------------------------------------------
    def repack(self):
        ops.set_device(self.device.index)
        self._join_updates()
        for n in self.nets.values():
            n.pack()

    def load_weights(self, P: Dict[str, Dict[str, torch.Tensor]]):
        ops.set_device(self.device.index)
        self._join_updates()            # cross-step mode: a queued Adam step must not land on top of the loaded weights
        for k in NETS:
            self.stores[k].load({n: t.to(self.device) for n, t in P[k].items()})
        self.repack()

    def export_weights(self):
        self._join_updates()
        return {k: self.stores[k].export() for k in NETS}

    def export_grads(self):
        self._join_updates()
        return {k: self.stores[k].export(self.stores[k].g) for k in NETS}

    # ------------------------------------------------------------------------------------------------
    def _make_noise(self, disc: PatchGAN, N: int, ar: Arena):
        if self.layer_noise <= 0 and self.dropout_rate <= 0:
            return None, None
        noise, drop = {}, {}
        if self.layer_noise > 0:
            # the five noise tensors of an application are views of ONE buffer filled by ONE launch (they share sigma; five launches of
            # ~15 us sat on the lane's chain in front of every discriminator application); sub-tensors start on 16-byte boundaries
            shapes = disc.noise_shapes(N)
            sizes = {k: int(math.prod(shp)) for k, shp in shapes.items()}
            flat = ar.alloc((sum((n + 7) // 8 * 8 for n in sizes.values()),), torch.bfloat16)
            if self._cap is not None:       # under graph capture: counter base and standard deviation come from the device block
                ops.randn_bf16_dev(flat, self._cap.std_ptr, self.noise_key, self._cap.off_ptr, self.rng_offset - self._cap.base)
            else:
                ops.randn_bf16(flat, self.layer_noise, self.noise_key, self.rng_offset)
            self.rng_offset += (flat.numel() + 3) // 4
            o = 0
            for k, shp in shapes.items():
                noise[k] = flat[o:o + sizes[k]].view(shp)
                o += (sizes[k] + 7) // 8 * 8
        if self.dropout_rate > 0:
            chans = (('down0', 128), ('down1', 256), ('down2', 512))
            dflat = ar.alloc((N * sum(c for _, c in chans),), torch.float32)          # likewise: one launch for the three masks
            if self._cap is not None:
                ops.dropout_mask_dev(dflat, self.dropout_rate, self.drop_key, self._cap.off_ptr, self.rng_offset - self._cap.base)
            else:
                ops.dropout_mask(dflat, self.dropout_rate, self.drop_key, self.rng_offset)
            self.rng_offset += dflat.numel()
            o = 0
            for k, c in chans:
                drop[k] = dflat[o:o + N * c].view(N, c)
                o += N * c
            if self.wasserstein:                 # Dropout(0.2) in front of the Dense head: elementwise over the flattened patch logits
                t = ar.alloc((N, disc.n_patch), torch.float32)
                if self._cap is not None:
                    ops.dropout_mask_dev(t, self.dropout_rate, self.drop_key + 77, self._cap.off_ptr, self.rng_offset - self._cap.base)
                else:
                    ops.dropout_mask(t, self.dropout_rate, self.drop_key + 77, self.rng_offset)
                self.rng_offset += t.numel()
                drop['head'] = t
        return noise, drop

    def _make_gen_drop(self, N: int, ar: Arena):
        """Channel multipliers of the ResNet generator's SpatialDropout3D layers for one application (ResNetGenerator.DROP_RATES)."""
        out = {}
        for k, rate in ResNetGenerator.DROP_RATES.items():
            t = ar.alloc((N, ResNetGenerator.DROP_CH[k]), torch.float32)
            if self._cap is not None:
                ops.dropout_mask_dev(t, rate, self.drop_key + 31, self._cap.off_ptr, self.rng_offset - self._cap.base)
            else:
                ops.dropout_mask(t, rate, self.drop_key + 31, self.rng_offset)
            self.rng_offset += t.numel()
            out[k] = t
        return out

    def _join_updates(self):
        """Cross-step mode: the current stream waits for every optimizer step still queued (consumers outside train_step)."""
        if self._upd_ev and self._opt is not None:
            torch.cuda.current_stream().wait_stream(self._opt)
            self._upd_ev = {}

    def _need(self, *names):
        """The current stream is about to read the weights of these networks: wait for their pending update (cross-step mode)."""
        if self._upd_ev:
            cur = torch.cuda.current_stream()
            for n in names:
                ev = self._upd_ev.get(n)
                if ev is not None:
                    cur.wait_event(ev)

    def _mark(self, name: str):
        """Development aid (VG_TIMELINE=1): an event on the current stream, printed by timeline() -- where the lanes wait for each
        other, without a profiler's launch overhead."""
        if self._tl is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._tl.append((name, e))
        if roctx.ON:
            roctx.phase(_PHASE_AFTER.get(name, 'after ' + name))

    def timeline(self):
        torch.cuda.synchronize(self.device)
        t0 = self._tl[0][1]
        out = [(n, t0.elapsed_time(e)) for n, e in self._tl]
        self._tl = []
        return out

    def _losses_and_backward(self, real_I, real_S, training: bool, noise, drop, do_backward: bool, apply: bool = False):
        ar = self.arena
        ar.reset()
        if self._tl is not None:
            self._tl = []
        self._mark('A start')
        ops.set_device(self.device.index)          # module-level fast path of ops.stream(): this engine's device
        B = real_I.shape[0]
        D, H, W = self.dims
        S = D * H * W
        gbs = float(self.global_batch_size)
        vol = (B, D, H, W, 1)
        f32 = torch.float32
        acc = ar.alloc((16,), f32, zero=True)
        bufS, bufI = ar.alloc((2 * B, D, H, W, 1), f32), ar.alloc((2 * B, D, H, W, 1), f32)
        ops.copy(bufS[:B], real_S); ops.copy(bufI[:B], real_I)
        rI, rS, fake_S, fake_I = bufI[:B], bufS[:B], bufS[B:], bufI[B:]
        cyc_S, cyc_I = ar.alloc(vol, f32), ar.alloc(vol, f32)
        import contextlib
        main = torch.cuda.current_stream()
        lane_b = self._lane_b if ops.PROF is None else None        # the per-launch timing pass stays serial
        if lane_b is not None:
            ops.wait_stream(lane_b, main)                                 # inputs copied, arena reset
        def laneB():
            return torch.cuda.stream(lane_b) if lane_b is not None else contextlib.nullcontext()
        # Lane balance (timeline from HIP events, tools/timeline.py): lane B carries the clDice skeletons, so lane A used to idle
        # 3.2 ms at the forward join.  Everything of the S-side losses that depends on real_S only (its min-max normalisation and
        # the TARGET skeleton) and the forward of D_I (needs fake_I only) therefore run on lane A, behind events.
        it = self.skel_iters
        dims4 = (B, D, H, W)
        mmS, nS = ar.alloc((B, 4), f32), ar.alloc(vol, f32)
        ops.minmax(rS, B, S, mmS); ops.minmax_apply(rS, mmS, B, S, nS)                   # lane A, first thing: lane B needs nS at 5.8 ms
        ev_nS = ops.record_event(main) if lane_b is not None else None
        # Both applications of a generator share 2B-sample tensors (Arena paired mode): the forward passes are B-sample launches on
        # the two sample halves, the backward runs ONE 2B-sample sweep per generator instead of two B-sample sweeps (half the
        # launches, twice the work per launch on the latency-bound deep levels, the weight gradients' slab writes once for both).
        pair = do_backward and _PAIR_BWD and self.generator == 'resUnet'       # (the ResNet generator: one B-sample sweep per application)
        gdrop = (drop or {}) if self.generator == 'resnet' else {}
        def fwd(gen, key, slot, x, y):
            """A generator application as a resumable enqueue sequence; its allocations go to the paired slot (key, slot)."""
            if self.generator == 'resnet':
                # SpatialDropout3D of generator.py:44 / downsample(): per-application channel masks (training only); a test may hand
                # them in as drop['G_IS.a'] ... (application names of oracle.compute_losses)
                app = 'G_%s.%s' % (key[4:], 'ab'[slot])
                if app in gdrop:
                    gd = gdrop[app]
                elif training and noise is None and self.dropout_rate > 0:
                    gd = lambda: self._make_gen_drop(B, ar)           # drawn when the application is enqueued, on ITS lane's stream
                else:
                    gd = None
                it = gen.forward_iter(ar, x, y, gd)
            else:
                it = gen.forward_iter(ar, x, y)
            if pair:
                ar.pair_begin(key, slot); ar.pair_end()
            def steps():
                while True:
                    if pair:
                        ar.pair_resume(key, slot)
                    try:
                        next(it)
                    except StopIteration as e:
                        return e.value
                    finally:
                        ar.pair_end()
                    yield
            return steps()
        # (VG_INTERLEAVE=1 enqueues the two lanes' sequences ALTERNATELY, block by block -- the idea: enqueued one after the other, the
        # second lane's stream sits empty for the first one's enqueue time.  Measured neutral: the host runs far enough ahead.)
        self._need('gen_IS'); self._need('gen_SI')
        with laneB():
            self._need('gen_SI'); self._need('gen_IS')
        c1, c2 = interleave((contextlib.nullcontext, fwd(self.gen_IS, 'gen_IS', 0, rI, fake_S)),       # vangan.py:295   (lane A)
                            (laneB, fwd(self.gen_SI, 'gen_SI', 0, rS, fake_I)), on=_INTERLEAVE)           # :297            (lane B)
        self._mark('A G1 fwd')
        with laneB():
            ev_fakeI = ops.record_event(lane_b) if lane_b is not None else None
            self._mark('B G1 fwd')
        c4, c3 = interleave((contextlib.nullcontext, fwd(self.gen_SI, 'gen_SI', 1, fake_S, cyc_I)),     # :305            (lane A)
                            (laneB, fwd(self.gen_IS, 'gen_IS', 1, fake_I, cyc_S)), on=_INTERLEAVE)        # :300            (lane B)
        self._mark('A G2 fwd')
        with laneB():
            self._mark('B G2 fwd')
        # upstream gradients of the two applications, adjacent: [adversarial (through the discriminator); cycle]
        gS2 = ar.alloc((2 * B, D, H, W, 1), f32) if do_backward else None
        gI2 = ar.alloc((2 * B, D, H, W, 1), f32) if do_backward else None

        # ---- cycle / segmentation losses on cycled_S (loss_functions.py:185-190, 211-226): lane B; target skeleton: lane A ----
        imgs_t, skels_t = ar.alloc((it + 2,) + vol, f32), ar.alloc((it + 1,) + vol, f32)
        # VG_SKEL_FWD_A: the PREDICTED skeleton's forward pass on lane A as well (behind lane B's min-max normalisation of cycled_S) -- lane B
        # is the longer lane; its clDice then starts from the finished skeletons
        skel_fwd_on_a = _SKEL_FWD_A and lane_b is not None
        ev_ncS = None
        if skel_fwd_on_a:
            with laneB():
                ops.wait_event(lane_b, ev_nS)
                mmcS = ar.alloc((B, 4), f32)
                ncS = ar.alloc(vol, f32)
                ops.minmax(cyc_S, B, S, mmcS); ops.minmax_apply(cyc_S, mmcS, B, S, ncS)
                ev_ncS = ops.record_event(lane_b)
        ops.soft_skel_fwd(nS, dims4, it, imgs_t, skels_t)                                # lane A
        if skel_fwd_on_a:
            imgs_p, skels_p = ar.alloc((it + 2,) + vol, f32), ar.alloc((it + 1,) + vol, f32)
            aux_p = ar.alloc((ops.skel_aux_bytes(dims4, it),), torch.uint8) if (do_backward and _SKEL_AUX) else None
            ops.wait_event(main, ev_ncS)
            ops.soft_skel_fwd(ncS, dims4, it, imgs_p, skels_p, aux_p)                    # lane A
        ev_skel_t = ops.record_event(main) if lane_b is not None else None
        self._mark('A target skeleton')
        with laneB():
            if not skel_fwd_on_a:
                if lane_b is not None:
                    ops.wait_event(lane_b, ev_nS)
                mmcS = ar.alloc((B, 4), f32)
                ncS = ar.alloc(vol, f32)
                ops.minmax(cyc_S, B, S, mmcS); ops.minmax_apply(cyc_S, mmcS, B, S, ncS)
            g_ncS = ar.alloc(vol, f32) if do_backward else None
            ops.bce(nS, ncS, acc[0:1], self.lambda_cycle / (B * S * gbs), g_ncS, accumulate=False)
            if not skel_fwd_on_a:
                imgs_p, skels_p = ar.alloc((it + 2,) + vol, f32), ar.alloc((it + 1,) + vol, f32)
                # the predicted skeleton is differentiated: its forward pass files delta and the pooling arg-extrema codes (6 B per voxel and step)
                aux_p = ar.alloc((ops.skel_aux_bytes(dims4, it),), torch.uint8) if (do_backward and _SKEL_AUX) else None
                ops.soft_skel_fwd(ncS, dims4, it, imgs_p, skels_p, aux_p)
            if lane_b is not None:
                ops.wait_event(lane_b, ev_skel_t)
            skel_p, skel_t = skels_p[it], skels_t[it]
            sums = ar.alloc((9,), f32, zero=True)
            coef = ar.alloc((8,), f32, zero=True)
            ops.dot_sums(skel_p, nS, sums[0:3]); ops.dot_sums(skel_t, ncS, sums[3:6]); ops.dot_sums(nS, ncS, sums[6:9])
            ops.cldice_coef(sums, self.lambda_topology / self.n_devices, 0.5, coef)
            def cldice_backward():
                gskel = ar.alloc(vol, f32)
                ops.cldice_grads(nS, skel_t, coef, gskel, g_ncS, accumulate=True)
                work = ar.alloc((4,) + vol, f32)
                ops.soft_skel_bwd(imgs_p, skels_p, gskel, dims4, it, work, g_ncS, aux_p)
                tmp2 = ar.alloc((B, 2), f32, zero=True)
                ops.minmax_bwd(cyc_S, ncS, g_ncS, mmcS, B, S, tmp2, gS2[B:])
            g_cS = gS2[B:] if do_backward else None
            # VG_SKEL_BWD_A: the skeleton's backward (34 launches whose result only lane A's generator sweep consumes) on lane A, behind
            # an event of lane B's clDice forward -- lane B is the longer lane (tools/timeline.py: lane A ends 1.2 ms before it)
            skel_bwd_on_a = do_backward and _SKEL_BWD_A and lane_b is not None
            ev_cldice = ops.record_event(lane_b) if skel_bwd_on_a else None
            if do_backward and not skel_bwd_on_a:
                cldice_backward()
            self._mark('B clDice')

        # ---- cycle MSE + SSIM reconstruction on cycled_I (loss_functions.py:179-180, 193-208) ----
        g_cI = gI2[B:] if do_backward else None
        ops.mse(rI, cyc_I, acc[1:2], self.lambda_cycle / (S * gbs), g_cI, accumulate=False)
        mmI, mmcI = ar.alloc((B, 4), f32), ar.alloc((B, 4), f32)
        nI, ncI = ar.alloc(vol, f32), ar.alloc(vol, f32)
        ops.minmax(rI, B, S, mmI); ops.minmax_apply(rI, mmI, B, S, nI)
        ops.minmax(cyc_I, B, S, mmcI); ops.minmax_apply(cyc_I, mmcI, B, S, ncI)
        part = ar.alloc((3,) + vol, f32) if do_backward else None
        ops.ssim_fwd(nI, ncI, dims4, acc[2:3], part)
        if do_backward:
            g_ncI = ar.alloc(vol, f32)
            ops.ssim_bwd(nI, ncI, part, dims4, self.lambda_reconstruction / (B * S * gbs), g_ncI, accumulate=False)
            g_tmp = ar.alloc(vol, f32)
            tmp2b = ar.alloc((B, 2), f32, zero=True)
            ops.minmax_bwd(cyc_I, ncI, g_ncI, mmcI, B, S, tmp2b, g_tmp)
            ops.axpby(g_tmp, 1.0, None, 0.0, g_cI, accumulate=True)
        self._mark('A cycle losses')

        # ---- discriminators on [real; fake] (vangan.py:315-319) and LSGAN losses (:329-332) ----
        ld = tuple(n // 8 for n in self.dims)
        nps = ld[0] * ld[1] * ld[2]
        logS, logI = ar.alloc((2 * B,) + ld + (1,), f32), ar.alloc((2 * B,) + ld + (1,), f32)
        if training and noise is None:
            nzS, dpS = self._make_noise(self.disc_S, 2 * B, ar)
            nzI, dpI = self._make_noise(self.disc_I, 2 * B, ar)
        else:
            noise, drop = noise or {}, drop or {}
            nzS, dpS, nzI, dpI = noise.get('S'), drop.get('S'), noise.get('I'), drop.get('I')
        self._need('disc_S', 'disc_I')
        dS = self.disc_S.forward(ar, bufS, logS, nzS, dpS)                                  # lane A: needs fake_S
        gd = 1.0 / (nps * gbs)
        # upstream gradients at the patch logits, adjacent: [d critic loss (2B: real, fake); d generator loss (B: fake)] -- one 3B-sample
        # backward sweep per discriminator reads them as one tensor (VG_D_ONE_SWEEP)
        gS3 = ar.alloc((3 * B,) + tuple(logS.shape[1:]), f32) if do_backward else None
        gI3 = ar.alloc((3 * B,) + tuple(logI.shape[1:]), f32) if do_backward else None
        gS_D, gI_D = (gS3[:2 * B], gI3[:2 * B]) if do_backward else (None, None)
        gS_G, gI_G = (gS3[2 * B:], gI3[2 * B:]) if do_backward else (None, None)
        wz = {}
        def w_terms(tag, disc, logits, dp, a0):
            """Wasserstein mode: Dense head over [real; fake] patch logits, sum z_real / sum z_fake into acc[a0:a0+2], d loss / d z of the
            critic loss (2B) and of the generator loss (fake half) -- the head's backward runs with the sweeps (after the gradient buffers are cleared)."""
            z = ar.alloc((2 * B,), f32)
            mask = (dp or {}).get('head')
            disc.head_forward(logits, mask, z)
            gzd, gzg = (ar.alloc((2 * B,), f32), ar.alloc((B,), f32)) if do_backward else (None, None)
            ops.wasserstein_terms(z, B, 1.0 / (B * gbs), acc[a0:a0 + 2], gzd, gzg)
            wz[tag] = (mask, gzd, gzg)
        if self.wasserstein:
            w_terms('S', self.disc_S, logS, dpS, 9)
        else:
            ops.mse_const(logS[B:], 1.0, acc[3:4], gd, gS_G)                                  # gen_IS_loss
            ops.mse_const(logS[:B], 1.0, acc[5:6], 0.5 * gd, None if gS_D is None else gS_D[:B])
            ops.mse_const(logS[B:], 0.0, acc[6:7], 0.5 * gd, None if gS_D is None else gS_D[B:])
        self._mark('A D_S fwd')
        if lane_b is not None:
            ops.wait_event(main, ev_fakeI)                                                       # fake_I comes from lane B's first generator
        dI = self.disc_I.forward(ar, bufI, logI, nzI, dpI)                                  # lane A as well: lane B is the longer one
        if self.wasserstein:
            w_terms('I', self.disc_I, logI, dpI, 11)
        else:
            ops.mse_const(logI[B:], 1.0, acc[4:5], gd, gI_G)                                  # gen_SI_loss
            ops.mse_const(logI[:B], 1.0, acc[7:8], 0.5 * gd, None if gI_D is None else gI_D[:B])
            ops.mse_const(logI[B:], 0.0, acc[8:9], 0.5 * gd, None if gI_D is None else gI_D[B:])
        self._mark('A D fwd')
        if skel_bwd_on_a and _SKEL_BWD_A == 1:
            ops.wait_event(main, ev_cldice)
            cldice_backward()
        # No full join before the backward sweeps (VG_NOJOIN): lane A's discriminator sweeps and its adversarial generator sweep need
        # nothing of lane B; only its cycle sweep (c3 ran on lane B, g_cS comes out of lane B's clDice) waits for lane B's forward.
        nojoin = lane_b is not None and do_backward and _NOJOIN
        ev_bfwd = None
        if lane_b is not None:
            if nojoin:
                ev_bfwd = ops.record_event(lane_b)
            else:
                ops.wait_stream(main, lane_b)                                                    # lanes join before the backward sweeps

        self._upd_ev = {}                # every stream that reads weights in this step has queued its waits (main: all four networks)
        if do_backward:
            for st in self.stores.values():
                ops.zero_fill(st.g)
            # Backward lanes: lane A = D_S sweeps + both gen_IS applications, lane B = D_I sweeps + both gen_SI applications
            # (total_loss_I only reaches gen_IS, total_loss_S only gen_SI; each lane accumulates into its own networks'
            # gradient buffers).  Per network: D loss over [real;fake] (weights), generator loss through the fake half.
            g_fS, g_fI = gS2[:B], gI2[:B]
            arB = ar
            if lane_b is not None:
                arB = self.arena_b
                arB.reset()
                ops.wait_stream(lane_b, main)
            def w_head_bwd(tag, disc, logits, g_D, g_G):
                mask, gzd, gzg = wz[tag]
                disc.head_backward(logits, mask, gzd, g_D, wgrad=True)                                   # critic loss: [real; fake], Dense gradients
                disc.head_backward(logits[B:], None if mask is None else mask[B:], gzg, g_G, wgrad=False)  # generator loss through the fake half

            def a_disc():
                if self.wasserstein:
                    w_head_bwd('S', self.disc_S, logS, gS_D, gS_G)
                if _D_ONE_SWEEP:
                    self.disc_S.backward_both(ar, dS, gS3, B, g_fS)
                    self._start_allreduce(['disc_S'], lazy=apply)
                else:
                    self.disc_S.backward(ar, dS, gS_D, 0, 2 * B, wgrad=True)
                    self._start_allreduce(['disc_S'], lazy=apply)
                    self.disc_S.backward(ar, dS, gS_G, B, 2 * B, wgrad=False, dx=g_fS)       # still reads D_S's packed weights
                if apply:
                    self._schedule_update('disc_S')
                self._mark('A D bwd')

            def b_disc():
                # (Moving D_I's D-loss sweep to lane A, whose sweeps finish 3.7 ms before lane B's, was measured: the main lanes then end
                # at 26.8 / 23.4 ms but lane A's weight-gradient side stream becomes the tail -- 29.9 vs 29.4 ms per step.)
                with laneB():
                    if self.wasserstein:
                        w_head_bwd('I', self.disc_I, logI, gI_D, gI_G)
                    if _D_ONE_SWEEP:
                        self.disc_I.backward_both(arB, dI, gI3, B, g_fI)
                        self._start_allreduce(['disc_I'], lazy=apply)
                    else:
                        self.disc_I.backward(arB, dI, gI_D, 0, 2 * B, wgrad=True)
                        self._start_allreduce(['disc_I'], lazy=apply)
                        self.disc_I.backward(arB, dI, gI_G, B, 2 * B, wgrad=False, dx=g_fI)
                    if apply:
                        self._schedule_update('disc_I')
                    self._mark('B D bwd')

            mk = ar.mark()
            mkb = arB.mark()
            self._bwd_ctx = {'gen_IS': [c1, c3], 'gen_SI': [c2, c4]}        # what the sweeps store (gradient buffers hang on the Act objects): for tests

            def a_adv():
                if pair:
                    return
                self.gen_IS.backward(ar, c1, g_fS); ar.release(mk, defer=True)        # adversarial application
                self._mark('A G adv bwd')

            def b_adv():
                if pair:
                    return
                with laneB():
                    self.gen_SI.backward(arB, c2, g_fI); arB.release(mkb, defer=True)
                    self._mark('B G adv bwd')

            def a_cyc():
                if ev_bfwd is not None:
                    ops.wait_event(main, ev_bfwd)                                          # c3 and g_cS are lane B's
                self.gen_IS.backward(ar, c3, g_cS, inline_from=self._inline); ar.release(mk, defer=True)    # cycle application
                self._mark('A G cyc bwd')
                self._start_allreduce(['gen_IS'], lazy=apply)
                if apply:
                    self._schedule_update('gen_IS')

            def b_cyc():
                with laneB():
                    self.gen_SI.backward(arB, c4, g_cI, inline_from=self._inline); arB.release(mkb, defer=True)
                    self._mark('B G cyc bwd')
                    self._start_allreduce(['gen_SI'], lazy=apply)
                    if apply:
                        self._schedule_update('gen_SI')

            # host enqueue order per stage (bit i of VG_BFIRST: lane B's sweep of stage i is enqueued before lane A's)
            stages = ((a_disc, b_disc), (a_adv, b_adv)) if pair else ((a_disc, b_disc), (a_adv, b_adv), (a_cyc, b_cyc))
            for i, (fa, fb) in enumerate(stages):
                if (_BFIRST >> i) & 1:
                    fb(); fa()
                else:
                    fa(); fb()
            if pair:
                # the generators' 2B-sample sweeps over both applications ([adversarial; cycle]), enqueued alternately block by block
                if skel_bwd_on_a and _SKEL_BWD_A == 2:
                    ops.wait_event(main, ev_cldice)
                    cldice_backward()
                if ev_bfwd is not None:
                    ops.wait_event(main, ev_bfwd)                                          # c3 and g_cS are lane B's
                ccA = pair_ctx(ar, c1, bufI, (fake_S, cyc_S), self.gen_IS.lv[0])
                ccB = pair_ctx(ar, c2, bufS, (fake_I, cyc_I), self.gen_SI.lv[0])
                self._bwd_ctx = {'gen_IS': [ccA], 'gen_SI': [ccB]}
                # data parallel: the finished suffix of a generator's gradient bucket (enc4 ... output head, 34 of 38 MB) goes to the
                # all-reduce when the sweep has passed enc4 -- with ~40 % of the sweep still ahead; only the last 4 MB wait for its end
                split = self.ddp and apply and _AR_SPLIT
                early_adam = (_EARLY_ADAM and apply and self._opt is not None and self._cap is None and ops.PROF is None
                              and ops.DRY is None and ops.REC is None and (not self.ddp or _EARLY_ADAM_DDP))
                def early(name, gen):
                    ea = early_adam and hasattr(gen, 'grad_suffix_offset')
                    if split:
                        def f():
                            self._start_allreduce([name], lazy=apply, lo=gen.grad_suffix_offset())
                            if ea:
                                self._early_update(name)         # behind the suffix's all-reduce (queued on / awaited by the optimizer stream)
                        return f
                    if ea and not self.ddp:
                        return lambda: self._early_update(name)
                    return None
                order = ((laneB, self.gen_SI.backward_iter(arB, ccB, gI2, inline_from=self._inline, on_suffix_done=early('gen_SI', self.gen_SI))),
                         (contextlib.nullcontext, self.gen_IS.backward_iter(ar, ccA, gS2, inline_from=self._inline, on_suffix_done=early('gen_IS', self.gen_IS))))
                interleave(*(order if (_BFIRST >> 2) & 1 else order[::-1]), on=_INTERLEAVE)
                hiA = self.gen_IS.grad_suffix_offset() if split else None
                hiB = self.gen_SI.grad_suffix_offset() if split else None
                ar.release(mk, defer=True)
                self._mark('A G cyc bwd')
                self._start_allreduce(['gen_IS'], lazy=apply, hi=hiA)
                if apply:
                    self._schedule_update('gen_IS')
                with laneB():
                    arB.release(mkb, defer=True)
                    self._mark('B G cyc bwd')
                    self._start_allreduce(['gen_SI'], lazy=apply, hi=hiB)
                    if apply:
                        self._schedule_update('gen_SI')
            if lane_b is not None:
                with laneB():
                    ops.side_join()                 # lane B's weight gradients (its lane no longer waits for them on the way)
                ops.wait_stream(main, lane_b)
            ops.side_join()
            if apply and self._opt is not None and not (self._xstep and ops.PROF is None):
                ops.wait_stream(main, self._opt)
            self._mark('A all joined')
        self._acc, self._coef = acc, coef
        self._aux = dict(fake_S=fake_S, fake_I=fake_I, cycled_S=cyc_S, cycled_I=cyc_I, logits_S=logS, logits_I=logI)
        # forward contexts of this step (views into the arena, valid until the next step resets it): what the engine STORED, for the
        # teacher-forced parity test (tests/test_gpu_teacher.py).  G_IS.a = G_IS(real_I), G_SI.a = G_SI(real_S), G_IS.b = G_IS(fake_I),
        # G_SI.b = G_SI(fake_S); the discriminators ran on [real; fake] batches
        self._fwd_ctx = {'G_IS.a': c1, 'G_SI.a': c2, 'G_IS.b': c3, 'G_SI.b': c4, 'D_S': dS, 'D_I': dI}
        return B, S, nps

    def _results(self, B, S, nps) -> Dict[str, float]:
        a = self._acc.cpu().tolist()        # the only host sync of the step
        seg = float(self._coef[5].item())
        gbs = float(self.global_batch_size)
        cyc_I = a[0] / (B * S * gbs) * self.lambda_cycle
        cyc_S = a[1] / (S * gbs) * self.lambda_cycle
        rec = a[2] / (B * S * gbs) * self.lambda_reconstruction
        if self.wasserstein:            # -reduce_mean(D(fake)), -reduce_mean(D(real) - D(fake)): acc[9:11] = (sum z_real, sum z_fake) of D_S, [11:13] of D_I
            gIS, gSI = -a[10] / (B * gbs), -a[12] / (B * gbs)
            dS, dI = -(a[9] - a[10]) / (B * gbs), -(a[11] - a[12]) / (B * gbs)
        else:
            gIS, gSI = a[3] / (nps * gbs), a[4] / (nps * gbs)
            dS = 0.5 * (a[5] + a[6]) / (nps * gbs)
            dI = 0.5 * (a[7] + a[8]) / (nps * gbs)
        vals = [gIS + cyc_I + seg, gSI + cyc_S + rec, dI, dS, gIS, gSI, cyc_I, cyc_S, seg, rec]
        return dict(zip(RESULT_KEYS, vals))

    # ------------------------------------------------------------------------------------------------
    def _start_allreduce(self, names, lazy: bool = False, lo: int = 0, hi=None):
        """The gradient buckets `names` are complete once the current lane AND its weight-gradient side stream have run what was
        issued so far.  lazy (an optimizer step follows on the optimizer stream): the lane itself does not wait for its side
        stream -- the all-reduce stream and the optimizer stream do (it used to stall, e.g. between the two discriminator sweeps,
        until the discriminator's weight gradients had finished)."""
        if lazy and self._opt is not None and ops.PROF is None and _LAZY_AR:
            ev = ops.side_event()
            for n in names:
                self._side_ev[n] = ev
            self.sync.start(names, also=ev, lo=lo, hi=hi)
            return
        ops.side_join()                     # the weight gradients of these networks were issued on the side stream
        self.sync.start(names, lo=lo, hi=hi)

    def _finish_allreduce(self):
        self.sync.finish()

    def _adam(self, name: str):
        """a22: per-variable clip-by-norm + Adam on the (reduced) flat gradient bucket of one network, then its bf16 repack."""
        st = self.stores[name]
        early = self._early_done.pop(name, None)
        if early is not None:                  # the suffix was updated and repacked when the sweep passed enc4 (_early_update): the prefix is left
            ep, lr_t = early
            off = ep['off']
            ops.adam_clip(st.w[:off], st.g[:off], st.m[:off], st.v[:off], ep['seg_lo'], ep['T_lo'], ep['norms_lo'], lr_t, self.beta_1, self.beta_2,
                          self.adam_eps, self.clipnorm, 1.0)
            ep['ptab_lo'].run()
            return
        st.step += 1
        t = st.step
        if self._cap is not None:              # under graph capture: lr_t of the replayed step comes from the device block
            ops.adam_clip_dev(st.w, st.g, st.m, st.v, st.seg_off, st.T, st.norms, self._cap.lr_ptr(name), self.beta_1, self.beta_2,
                              self.adam_eps, self.clipnorm, 1.0)
        else:
            ops.adam_clip(st.w, st.g, st.m, st.v, st.seg_off, st.T, st.norms, self._lr_t(name, t), self.beta_1, self.beta_2,
                          self.adam_eps, self.clipnorm, 1.0)
        self.nets[name].pack()

    def _early_parts(self, name: str):
        """Split of a ResUNet generator's flat parameter buffer at the first parameter of enc4 (nets.ResUNet.grad_suffix_offset): segment
        tables, norm scratch and repack tables of the two parts.  A backward sweep completes the gradients from the END of the buffer
        (output head, decoder, bridge, enc4: 8.65 of 9.54 M parameters) with ~40 % of the sweep still ahead; clip-by-norm is per variable,
        so the suffix can be updated then (vangan.py:426-438 applies the optimizers after the tape: the same update, earlier)."""
        ep = self._early_tab.get(name)
        if ep is None:
            net, st = self.nets[name], self.stores[name]
            off = net.grad_suffix_offset()
            bounds = st.seg_off.cpu().tolist()
            k = bounds.index(off)
            dev = st.w.device
            hi_names = ('enc4', 'bridge', 'dec3', 'dec2', 'dec1', 'dec0', 'out')
            lay_hi = [l for n_, l in net.L.items() if n_.split('.')[0] in hi_names]
            lay_lo = [l for n_, l in net.L.items() if n_.split('.')[0] not in hi_names]
            assert off % 4 == 0 and len(lay_hi) + len(lay_lo) == len(net.L)
            nrm = lambda n_el, T: torch.zeros(T + 2 * ((n_el + 4095) // 4096), dtype=torch.float32, device=dev)
            ep = self._early_tab[name] = dict(
                off=off, T_lo=k, T_hi=st.T - k,
                seg_lo=torch.tensor(bounds[:k + 1], dtype=torch.int64, device=dev),
                seg_hi=torch.tensor([b - off for b in bounds[k:]], dtype=torch.int64, device=dev),
                norms_lo=nrm(off, k), norms_hi=nrm(st.total - off, st.T - k),
                ptab_lo=ops.PackTable(lay_lo, dev), ptab_hi=ops.PackTable(lay_hi, dev))
        return ep

    def _early_update(self, name: str):
        """Called on a generator's lane when its (last) backward sweep has passed block enc4: clip + Adam + repack of the finished suffix
        on the optimizer stream, behind the lane and the weight gradients it has handed to its side stream so far."""
        st, ep = self.stores[name], self._early_parts(name)
        st.step += 1
        lr_t = self._lr_t(name, st.step)
        ops.wait_stream(self._opt, ops.current_stream_obj())
        sev = ops.side_event()
        if sev is not None:
            ops.wait_event(self._opt, sev)
        off = ep['off']
        with torch.cuda.stream(self._opt):
            self.sync.finish([name])                     # data parallel: the suffix's all-reduce (no-op otherwise)
            ops.adam_clip(st.w[off:], st.g[off:], st.m[off:], st.v[off:], ep['seg_hi'], ep['T_hi'], ep['norms_hi'], lr_t, self.beta_1, self.beta_2,
                          self.adam_eps, self.clipnorm, 1.0)
            ep['ptab_hi'].run()
        self._early_done[name] = (ep, lr_t)

    def _lr_t(self, name: str, t: int) -> float:
        """Adam's bias-corrected rate of network `name` at its step t (1-based); the base rate is self.lr, a per-network float or a
        schedule evaluated at the iteration count BEFORE this step, as Keras does."""
        lr = self.lrs.get(name)
        lr = self.lr if lr is None else (float(lr(t - 1)) if callable(lr) else lr)
        return lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def _schedule_update(self, name: str):
        """Queue clip + Adam + repack of one network behind (a) everything the current lane has issued for it and (b) the
        all-reduce of ITS bucket only -- on the optimizer stream, so the lane carries on with the next backward sweep.
        Called after the last kernel that reads the network's packed weights in this step."""
        if self._opt is None or ops.PROF is not None:          # serial mode (per-launch timing pass, VG_OPT_STREAM=0)
            self.sync.finish([name])
            self._adam(name)
            return
        ops.wait_stream(self._opt, ops.current_stream_obj())
        sev = self._side_ev.pop(name, None)
        if sev is not None:
            ops.wait_event(self._opt, sev)      # ... and the weight gradients of this network on the lane's side stream
        with torch.cuda.stream(self._opt):
            self.sync.finish([name])
            self._adam(name)
            if self._xstep:
                self._upd_ev[name] = self._opt.record_event()

    def _apply_adam(self):
        ops.set_device(self.device.index)
        for name in NETS:
            self._adam(name)

    # ------------------------------------------------------------------------------------------------
    def train_step(self, real_I: torch.Tensor, real_S: torch.Tensor, noise=None, drop=None, apply: bool = True,
                   sync: bool = True):
        """vangan.py:380-440 (non-Wasserstein branch).  real_*: fp32 [B,D,H,W,1] on the device."""
        B, S, nps = self._losses_and_backward(real_I, real_S, True, noise, drop, True, apply=apply)
        self._finish_allreduce()                 # apply=False (tests): the reduced gradients stay in the buckets
        return self._results(B, S, nps) if sync else None

    # ------------------------------------------------------------------------------------------------
    def capture_train_step(self, warmup: int = 2):
        """Capture one train step (forward, losses, the backward sweeps, 4 x Adam, weight repack: everything train_step enqueues, on
        all of its streams) into a HIP graph.  train_step_graph() then replays it: the host refreshes a 32-byte parameter block
        (Philox counter, noise standard deviation, the four lr_t) and the two input volumes instead of enqueueing ~900 launches --
        for the configurations whose step the host cannot enqueue fast enough (64^3: 8.4 ms of Python for a 9.3 ms step).
        Single process only (the all-reduce stays outside graphs); shapes, dropout rate, loss weights and the noise on/off decision
        are those of the engine at capture time.  The library has been capture-legal since it stopped allocating (vg_conv_desc::scratch)."""
        if self.sync.active:
            raise NotImplementedError('graph capture covers the single-process step (the gradient all-reduce is not captured)')
        ops.set_device(self.device.index)
        B, (D, H, W) = self.batch_size, self.dims
        self._g_in = (torch.zeros(B, D, H, W, 1, device=self.device), torch.zeros(B, D, H, W, 1, device=self.device))
        self._g_stream = _engine_stream(self.device, 'graph')
        cur = torch.cuda.current_stream()
        self._g_stream.wait_stream(cur)
        # eager steps ON the capture stream first: per-stream workspaces, side stream, arena sequence, kernel attributes -- nothing is
        # allocated or configured while capturing.  (They are real steps: synthetic inputs would train on zeros, so the caller's
        # first batch is used -- hand it in through warm_inputs, else the weights are restored afterwards.)
        snap = {k: (s.w.clone(), s.m.clone(), s.v.clone(), s.step) for k, s in self.stores.items()}
        rng0 = self.rng_offset
        # The captured step runs WITHOUT the weight-gradient side streams: ROCm 7.2's hipStreamEndCapture takes the process down
        # (SIGSEGV inside the runtime) on a capture that forks the two lanes AND a side stream per lane; lanes + optimizer stream, or
        # side streams without the second lane, capture and replay fine (tools/graph_bisect.py).  The graph's branches still overlap.
        side_saved, ops.SIDE = ops.SIDE, None
        try:
            return self._capture(cur, snap, rng0, warmup)
        finally:
            ops.SIDE = side_saved

    def _capture(self, cur, snap, rng0, warmup):
        with torch.cuda.stream(self._g_stream):
            self._g_in[0].normal_(); self._g_in[1].normal_()
            for _ in range(max(1, warmup)):
                self.train_step(self._g_in[0], self._g_in[1], sync=False)
        torch.cuda.synchronize(self.device)
        for k, s in self.stores.items():
            s.w.copy_(snap[k][0]); s.m.copy_(snap[k][1]); s.v.copy_(snap[k][2]); s.step = snap[k][3]
        self.rng_offset = rng0
        self.repack()
        torch.cuda.synchronize(self.device)
        cap = _StepParams(self.device)
        cap.base = self.rng_offset
        steps0 = {k: s.step for k, s in self.stores.items()}
        g = torch.cuda.CUDAGraph()
        self._cap = cap
        try:
            with torch.cuda.graph(g, stream=self._g_stream):
                self._g_shape = self._losses_and_backward(self._g_in[0], self._g_in[1], True, None, None, True, apply=True)
        finally:
            self._cap = None
        cap.per_step = self.rng_offset - cap.base
        self.rng_offset = cap.base                      # nothing ran while capturing
        for k, s in self.stores.items():
            s.step = steps0[k]
        self._graph, self._gcap = g, cap
        cur.wait_stream(self._g_stream)
        return g

    def train_step_graph(self, real_I: torch.Tensor, real_S: torch.Tensor, sync: bool = True):
        """One train step by replaying the captured graph (capture_train_step): same arithmetic, same streams' dependencies."""
        if self._graph is None:
            self.capture_train_step()
        ops.set_device(self.device.index)
        self._g_in[0].copy_(real_I); self._g_in[1].copy_(real_S)
        lr_t = [self._lr_t(n, self.stores[n].step + 1) for n in NETS]
        self._gcap.refresh(self.rng_offset, max(self.layer_noise, 0.0), lr_t)
        self._graph.replay()
        self.rng_offset += self._gcap.per_step
        for n in NETS:
            self.stores[n].step += 1
        return self._results(*self._g_shape) if sync else None

    def train_step_replay(self, real_I: torch.Tensor, real_S: torch.Tensor, sync: bool = True):
        """train_step without the Python: the first call runs the step eagerly while recording every library launch and stream
        dependency it enqueues (ops.Recorder; the per-step scalars go through the device parameter block, _StepParams); later calls
        refresh the block and the two input volumes and re-issue the list -- ~1 000 C calls with baked arguments instead of ~8 ms of
        descriptor building, on the same streams with the same dependencies as the eager step.  For the configurations whose step the
        host cannot enqueue fast enough (64^3: 8.4 ms of enqueue for a 9.3 ms step).  Single process; shapes, dropout rate, loss
        weights and the noise on/off decision are those of the first call."""
        if self.sync.active:
            raise NotImplementedError('the launch list covers the single-process step (no all-reduce)')
        ops.set_device(self.device.index)
        if getattr(self, '_rlist', None) is None:
            B, (D, H, W) = self.batch_size, self.dims
            self._r_in = (torch.zeros(B, D, H, W, 1, device=self.device), torch.zeros(B, D, H, W, 1, device=self.device))
            self._rcap = cap = _StepParams(self.device)
            self._r_main = torch.cuda.current_stream()
        cap = self._rcap
        if torch.cuda.current_stream() != self._r_main:
            raise RuntimeError('train_step_replay must be called on the stream it was recorded on')
        self._r_in[0].copy_(real_I); self._r_in[1].copy_(real_S)
        lr_t = [self._lr_t(n, self.stores[n].step + 1) for n in NETS]
        cap.base = self.rng_offset                      # recorded offsets are relative to the step's base counter
        cap.refresh(self.rng_offset, max(self.layer_noise, 0.0), lr_t)
        if getattr(self, '_rlist', None) is None:
            self._cap = cap
            try:
                with ops.Recorder() as rec:
                    self._r_shape = self._losses_and_backward(self._r_in[0], self._r_in[1], True, None, None, True, apply=True)
            finally:
                self._cap = None
            cap.per_step = self.rng_offset - cap.base
            self._rlist, self._rkeep = rec.cmds, rec.keep
        else:
            for f, a in self._rlist:
                rc = f(*a)
                if rc is not None and rc < 0:
                    raise ops._lib.VgError('replayed launch %s failed: %s (%d)' % (getattr(f, '__name__', f), ops._lib.lib.vg_status_string(rc).decode(), rc))
            self.rng_offset += cap.per_step
            for n in NETS:
                self.stores[n].step += 1
        return self._results(*self._r_shape) if sync else None

    def test_step(self, real_I: torch.Tensor, real_S: torch.Tensor):
        """vangan.py:442-457: training=False => no noise, no dropout, no backward."""
        B, S, nps = self._losses_and_backward(real_I, real_S, False, {}, {}, False)
        return self._results(B, S, nps)

    def distributed_train_step(self, x, y):
        """vangan.py:475-490: per-replica step + SUM of the result dict over replicas."""
        res = self.train_step(x, y)
        return self.reduce_dict(res)

    def distributed_test_step(self, x, y):
        return self.reduce_dict(self.test_step(x, y))

    def reduce_dict(self, d: Dict[str, float]) -> Dict[str, float]:
        return self.sync.reduce_dict(d, RESULT_KEYS)

    def broadcast_weights(self, src: int = 0):
        ops.set_device(self.device.index)
        self._join_updates()            # (see load_weights)
        self.sync.broadcast_weights(src)
        self.repack()

    def generate(self, gen: str, x: torch.Tensor) -> torch.Tensor:
        """gen(x, training=False) of the reference's callers (custom_callback.py:174-175): generator forward on fp32 [B,D,H,W,1]
        device volumes at the engine's patch size, in chunks of the engine's batch size; returns fp32 [B,D,H,W,1] on the device."""
        if gen not in ('gen_IS', 'gen_SI'):
            raise ValueError('gen must be gen_IS or gen_SI')
        if x.dim() != 5 or tuple(x.shape[1:]) != self.dims + (1,):
            raise ValueError('expected [B, %d, %d, %d, 1] volumes, got %s' % (self.dims + (tuple(x.shape),)))
        ops.set_device(self.device.index)
        self._join_updates()
        net = self.nets[gen]
        out = torch.empty(x.shape, dtype=torch.float32, device=self.device)
        nb = max(1, self.batch_size)
        for i in range(0, x.shape[0], nb):
            self.arena.reset()
            xin = x[i:i + nb].to(self.device, torch.float32).contiguous()
            net.forward(self.arena, xin, out[i:i + nb], save=False)
        return out

    def fp16_generator(self, gen: str):
        """The generator `gen` as an fp16-storage network over the SAME fp32 master weights (forward only; built on first use),
        freshly repacked: 16-bit buffers of libvangan_hip_h.so hold IEEE half precision (include/vangan_hip.h: vg_storage16)."""
        if gen not in ('gen_IS', 'gen_SI'):
            raise ValueError('gen must be gen_IS or gen_SI')
        ops.set_device(self.device.index)
        self._join_updates()
        with ops.Fp16():
            net = self._fp16_nets.get(gen)
            if net is None:
                GenNet = ResUNet if self.generator == 'resUnet' else ResNetGenerator
                net = self._fp16_nets[gen] = GenNet(self.stores[gen], self.dims, torch.float16)
            net.pack()
        return net

    def stitch_subvolumes(self, gen: str, img, subvol_size=None, **kw):
        """GanMonitor.stitch_subvolumes (custom_callback.py:47-223) on the GPU; see van_gan_amd/inference.py."""
        from .inference import stitch_subvolumes as _st
        self._join_updates()
        return _st(self, gen, img, tuple(subvol_size) if subvol_size is not None else self.dims, **kw)

    # ------------------------------------------------------------------------------------------------
    def save_checkpoint(self, epoch: int):
        """vangan.py:247-250 (own format: the TF tensor-bundle format is not readable without TF).  The replicas hold identical
        weights and optimizer slots, so rank 0 writes (atomically: temp file + rename) and every rank meets at a barrier;
        the Philox counter of the noise / dropout streams is saved per file too (each rank resumes its own key at the
        common counter)."""
        if self.checkpoint_dir is None:
            raise ValueError('save_checkpoint needs the engine to be built with output_dir=...')
        path = os.path.join(self.checkpoint_dir, 'checkpoint_e%d.pt' % (epoch + 1))
        self._join_updates()
        if self.rank == 0:
            torch.cuda.synchronize(self.device)
            blob = {k: dict(w=s.w.cpu(), m=s.m.cpu(), v=s.v.cpu(), step=s.step) for k, s in self.stores.items()}
            blob['_rng_offset'] = int(self.rng_offset)
            tmp = path + '.tmp.%d' % os.getpid()
            torch.save(blob, tmp)
            os.replace(tmp, path)
        if self.pg is not None:
            import torch.distributed as dist
            dist.barrier(group=self.pg)
        return path

    def load_checkpoint(self, epoch: Optional[int], newpath: Optional[str] = None) -> bool:
        ops.set_device(self.device.index)
        d = newpath if newpath is not None else self.checkpoint_dir
        path = os.path.join(d, 'checkpoint_e%d.pt' % epoch) if (d is not None and epoch is not None) else ''
        if not d or not path or not os.path.exists(path):
            print('Error: Checkpoint not found!')                  # vangan.py:267-268: prints, does not raise
            return False
        ck = torch.load(path, map_location='cpu')
        self._join_updates()            # (see load_weights): the optimizer stream may still hold the last step's Adam + repack
        for k, s in self.stores.items():
            s.w.copy_(ck[k]['w']); s.m.copy_(ck[k]['m']); s.v.copy_(ck[k]['v']); s.step = ck[k]['step']
        self.rng_offset = int(ck.get('_rng_offset', 0))
        self.repack()
        return True
