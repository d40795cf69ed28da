"""Training data pipeline on HBM-resident volumes + per-epoch schedules: host mirror of the reference's DatasetGen
(dataset.py:205-251), process_imaging_otf (main.py:169-177) and GanMonitor.set_learning_rate /
updateDiscriminatorNoise (custom_callback.py:326-424).  SURVEY section 8 rows (f)2 and (f)3.

The volumes stay in HBM; a training batch is produced by one gather kernel per sample (crop + flips + rot90,
vg_crop_augment), the label-crop rejection test by vg_crop_max, the on-the-fly imaging normalisation by the min-max
kernels of the loss path.  Random draws come from a numpy Generator on the host (they are a few integers per sample) and
can be passed explicitly, which is how the parity tests compare with the oracle bit for bit."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import check, lib

SEG_THRESH = 0.8            # dataset.py:49
MAX_RECROPS = 200           # dataset.py:236


def _p(t):
    return None if t is None else t.data_ptr()


def crop_augment(vol: torch.Tensor, origin, patch, flip_lr: bool, flip_ud: bool, rot_k: int, out: torch.Tensor):
    """out[px,py,pz,C] = rot90_k(flip_ud?(flip_lr?(vol[origin : origin + patch])))   (tf.image semantics on [X,Y,Z,C])."""
    X, Y, Z, Cc = vol.shape
    check(lib.vg_crop_augment(_p(vol), X, Y, Z, Cc, origin[0], origin[1], origin[2], patch[0], patch[1], patch[2],
                              int(flip_lr), int(flip_ud), int(rot_k), _p(out), ops.stream()), 'vg_crop_augment')


def crop_max(vol: torch.Tensor, origin, patch, out: torch.Tensor):
    X, Y, Z, Cc = vol.shape
    check(lib.vg_crop_max(_p(vol), X, Y, Z, Cc, origin[0], origin[1], origin[2], patch[0], patch[1], patch[2], _p(out),
                          ops.stream()), 'vg_crop_max')


def rot_k_from_uniform(u: float, max_rotation_angle: float = 180.0) -> int:
    """dataset.py:215-218 (TP: the angle is in radians when it is floor-divided by 90, so k is -1 or 0)."""
    angle = (-max_rotation_angle + u * 2 * max_rotation_angle) * (math.pi / 180.0)
    return int(angle // 90)


class DataPipeline:
    """imaging / segmentation volumes: lists of fp32 [X,Y,Z,C] tensors on the device (C = 1 on the hot path)."""

    def __init__(self, imaging: Sequence[torch.Tensor], segmentation: Sequence[torch.Tensor], patch: Tuple[int, int, int],
                 batch_size: int, seed: int = 0, otf_imaging: bool = True):
        if not imaging or not segmentation:
            raise ValueError('empty dataset')
        self.imaging, self.segmentation = list(imaging), list(segmentation)
        self.patch, self.B = tuple(patch), batch_size
        self.rng = np.random.default_rng(seed)
        self.otf = otf_imaging
        dev = self.imaging[0].device
        self._mx = torch.zeros(1, device=dev)
        self._order_i: List[int] = []
        self._order_s: List[int] = []

    # -- draws (host): the reference shuffles the file lists once per pass (dataset.py:137-190) --
    def _next(self, order: List[int], n: int) -> int:
        if not order:
            order.extend(self.rng.permutation(n).tolist())
        return order.pop()

    def _origin(self, vol) -> Tuple[int, int, int]:
        return tuple(int(self.rng.integers(0, vol.shape[a] - self.patch[a] + 1)) for a in range(3))

    def seg_origin(self, vol) -> Tuple[Tuple[int, int, int], int]:
        """Label-crop rejection (dataset.py:229-249): re-draw while max(crop) < 0.8, at most 200 times."""
        o = self._origin(vol)
        tries = 0
        while tries < MAX_RECROPS:
            crop_max(vol, o, self.patch, self._mx)
            if float(self._mx.item()) >= SEG_THRESH:        # one scalar read-back per candidate (rare beyond the first)
                break
            o = self._origin(vol)
            tries += 1
        return o, tries

    def next_batch(self):
        """(real_I, real_S): fp32 [B, *patch, C] on the device, imaging normalised to [-1, 1] per sample when otf."""
        B, P = self.B, self.patch
        dev = self.imaging[0].device
        ci, cs = self.imaging[0].shape[3], self.segmentation[0].shape[3]
        rI = torch.empty((B,) + P + (ci,), device=dev)
        rS = torch.empty((B,) + P + (cs,), device=dev)
        for b in range(B):
            vi = self.imaging[self._next(self._order_i, len(self.imaging))]
            crop_augment(vi, self._origin(vi), P, self.rng.random() > 0.5, self.rng.random() > 0.5, 0, rI[b])
            vs = self.segmentation[self._next(self._order_s, len(self.segmentation))]
            o, _ = self.seg_origin(vs)
            crop_augment(vs, o, P, self.rng.random() > 0.5, self.rng.random() > 0.5, rot_k_from_uniform(self.rng.random()), rS[b])
        if self.otf:
            rI = imaging_otf(rI)
        return rI, rS


def imaging_otf(batch: torch.Tensor) -> torch.Tensor:
    """process_imaging_otf (main.py:169-177): 2*(x-min)/(max-min) - 1 per sample, on the GPU."""
    B = batch.shape[0]
    S = batch[0].numel()
    mm = torch.zeros(B, 4, device=batch.device)
    n01 = torch.empty_like(batch)
    ops.minmax(batch, B, S, mm)
    ops.minmax_apply(batch, mm, B, S, n01)
    key = (batch.device, batch.numel())
    ones = _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones(batch.numel(), device=batch.device)
    out = torch.empty_like(batch)
    ops.axpby(n01, 2.0, ones, -1.0, out)        # 2*n01 is exact, so this equals 2*(x-min)/(max-min) - 1 in fp32
    return out


_ONES = {}


# ------------------------------------------------------------------------------------------------------
# schedules (custom_callback.py:326-424): per-step scalars for VanGan.lr / VanGan.layer_noise
# ------------------------------------------------------------------------------------------------------
def learning_rate(initial_lr: float, epoch: int, step_in_epoch: int, epochs: int, initiate_decay: int, train_steps: int,
                  schedule_step: str = 'global_iterations', iterations: Optional[int] = None) -> float:
    """set_learning_rate (custom_callback.py:326-365): constant until epoch INITIATE_LR_DECAY, then
    PolynomialDecay(INITIAL_LR -> 0, decay_steps = (EPOCHS - INITIATE_LR_DECAY) * train_steps, power 1) assigned to
    ``optimizer.lr``.

    schedule_step selects the step the decay is evaluated at:
      'global_iterations' (default, reference-exact): TP -- Keras optimizer_v2._decayed_lr calls the schedule with the
          optimizer's own ``iterations``, which has counted every step since step 0.  With the reference defaults
          (EPOCHS 200, INITIATE_LR_DECAY 100) iterations == decay_steps at the swap, so the rate is 0 for the whole second half.
          ``iterations`` = optimizer steps taken so far (default: epoch * train_steps + step_in_epoch).
      'since_install': the presumably intended linear decay -- steps counted from the epoch the schedule was installed."""
    if epoch < initiate_decay:
        return initial_lr
    decay_steps = (epochs - initiate_decay) * train_steps
    if schedule_step == 'global_iterations':
        it = epoch * train_steps + step_in_epoch if iterations is None else iterations
    elif schedule_step == 'since_install':
        it = (epoch - initiate_decay) * train_steps + step_in_epoch
    else:
        raise ValueError("schedule_step must be 'global_iterations' or 'since_install'")
    step = min(it, decay_steps)
    lr = initial_lr * (1.0 - step / decay_steps)
    if lr <= 0.0 and epoch < epochs - 1:
        _warn_once('lr0', 'learning rate is 0 from epoch %d of %d on: the reference schedule evaluated at the optimizer\'s global '
                   'iteration count (custom_callback.py:326-365) has used up its decay window; schedule_step=\'since_install\' '
                   'gives the linear decay instead' % (epoch, epochs))
    return lr


def learning_rate_resumed(initial_lr: float, resume_epoch: int, epoch: int, step_in_epoch: int, epochs: int,
                          initiate_decay: int, train_steps: int, schedule_step: str = 'global_iterations',
                          iterations: Optional[int] = None) -> float:
    """set_learning_rate's resume branch (custom_callback.py:365-397): a new PolynomialDecay from
    INITIAL_LR / (EPOCHS - INITIATE) * (EPOCHS - resume_epoch) over (EPOCHS - INITIATE - resume_epoch) * train_steps steps
    (the reference subtracts both, kept as is; a NEGATIVE window gives 0 as TF's PolynomialDecay does, an empty one -- 0/0 in TF -- is refused)."""
    start = initial_lr / (epochs - initiate_decay) * (epochs - resume_epoch)
    decay_steps = (epochs - initiate_decay - resume_epoch) * train_steps
    if decay_steps < 0 and schedule_step == 'global_iterations':
        # TP: tf PolynomialDecay clamps the step to decay_steps (min(step, negative) / negative = 1) and returns end_learning_rate:
        # the reference carries on with a rate of 0
        _warn_once('lr0r', 'resumed past EPOCHS - INITIATE_LR_DECAY: the reference schedule yields a learning rate of 0')
        return 0.0
    if decay_steps <= 0:
        raise ValueError('schedule undefined: decay_steps <= 0 (resumed at or past EPOCHS - INITIATE_LR_DECAY; TF divides 0 by 0)')
    if schedule_step == 'global_iterations':
        it = epoch * train_steps + step_in_epoch if iterations is None else iterations      # restored optimizer.iterations
    elif schedule_step == 'since_install':
        it = (epoch - resume_epoch) * train_steps + step_in_epoch
    else:
        raise ValueError("schedule_step must be 'global_iterations' or 'since_install'")
    return start * (1.0 - min(it, decay_steps) / decay_steps)


_WARNED = set()


def _warn_once(key: str, msg: str):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def discriminator_noise(init_noise: float, epoch: int, no_noise_epoch: int) -> float:
    decay_rate = 1.0 if no_noise_epoch == 0 else epoch / no_noise_epoch
    return max(init_noise * (1.0 - decay_rate), 0.0)
