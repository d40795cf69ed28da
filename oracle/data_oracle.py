"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference's training data pipeline and per-epoch schedules.

Only tests/ may import this module.  PARITY UNPINNED against TensorFlow (not installable here; the reference has no
tests for these paths): the restatement follows the reference line by line, TP marks semantics that depend on TensorFlow
behaviour, and the random draws are explicit arguments so that the HIP path can be compared bit for bit.

Follows:
  DatasetGen.process_imaging_domain / process_seg_domain / random_spatial_augmentation   dataset.py:205-251
  process_imaging_otf                                                                    main.py:169-177
  GanMonitor.set_learning_rate / updateDiscriminatorNoise                                custom_callback.py:326-424
"""
import math

import numpy as np

SEG_THRESH = 0.8            # dataset.py:49
MAX_RECROPS = 200           # dataset.py:236


def crop(vol, origin, patch):
    """tf.image.random_crop(image, size) with the offset drawn by the caller (dataset.py:226,238).  vol [X,Y,Z,C]."""
    x0, y0, z0 = origin
    px, py, pz = patch
    return vol[x0:x0 + px, y0:y0 + py, z0:z0 + pz, :]


def spatial_augmentation(arr, flip_lr, flip_ud, rot_k):
    """random_spatial_augmentation (dataset.py:205-220) for given draws.  TP: on a 4-D tensor tf.image treats axis 0 (X) as
    batch, axis 1 (Y) as height and axis 2 (Z) as width: flip_left_right reverses Z, flip_up_down reverses Y, rot90
    turns the (Y, Z) plane counter-clockwise."""
    if flip_lr:
        arr = arr[:, :, ::-1, :]
    if flip_ud:
        arr = arr[:, ::-1, :, :]
    return np.rot90(arr, k=rot_k, axes=(1, 2))


def rot_k_from_uniform(u, max_rotation_angle=180):
    """dataset.py:215-218.  TP: the angle is converted to RADIANS before `// 90`, so k = floor(angle_rad / 90) is -1 for
    negative angles and 0 otherwise (|angle_rad| <= pi < 90): the segmentation volumes are turned by 0 or -90 degrees.
    u in [0,1) is the uniform draw."""
    angle = (-max_rotation_angle + u * 2 * max_rotation_angle) * (math.pi / 180.0)
    return int(angle // 90)


def process_imaging_domain(vol, origin, patch, flip_lr, flip_ud):
    """dataset.py:221-227: crop, flips only (preserve_depth_orientation=True)."""
    return spatial_augmentation(crop(vol, origin, patch), flip_lr, flip_ud, 0)


def process_seg_domain(vol, origins, patch, flip_lr, flip_ud, rot_k):
    """dataset.py:229-251: take the first candidate crop whose maximum reaches SEG_THRESH; after MAX_RECROPS re-draws the
    last candidate is used whatever it holds.  origins: the sequence of candidate offsets (>= 1 entries).  Returns the
    augmented patch and the index of the candidate used."""
    used = 0
    arr = crop(vol, origins[0], patch)
    i = 0
    while i < MAX_RECROPS and arr.max() < SEG_THRESH:
        used = min(i + 1, len(origins) - 1)
        arr = crop(vol, origins[used], patch)
        i += 1
    return spatial_augmentation(arr, flip_lr, flip_ud, rot_k), used


def process_imaging_otf(batch):
    """main.py:169-177: per-sample min-max over axes (1,2,3,4) to [-1, 1].  batch [B,D,H,W,C]."""
    mx = batch.max(axis=(1, 2, 3, 4), keepdims=True)
    mn = batch.min(axis=(1, 2, 3, 4), keepdims=True)
    return 2.0 * (batch - mn) / (mx - mn) - 1.0


def learning_rate(initial_lr, epoch, step_in_epoch, epochs, initiate_decay, train_steps, schedule_step='global_iterations'):
    """set_learning_rate (custom_callback.py:326-365): constant until epoch == INITIATE_LR_DECAY, then Keras
    PolynomialDecay(power=1, end=0, cycle=False) over decay_steps = (EPOCHS - INITIATE_LR_DECAY) * train_steps:
        lr(step) = initial * (1 - min(step, decay_steps) / decay_steps)
    TP (tf.keras 2.10 optimizer_v2._decayed_lr, restated from memory: ``local_step = cast(self.iterations); lr_t =
    lr_t(local_step)``): the schedule's step is the optimizer's GLOBAL iteration count, which already equals
    INITIATE_LR_DECAY * train_steps when the schedule is installed -- with the reference defaults (200 / 100) that is
    decay_steps, so the learning rate is 0 from epoch 100 on.  schedule_step='since_install' is the
    linear decay the authors presumably intended; it is NOT what the reference computes."""
    if epoch < initiate_decay:
        return initial_lr
    decay_steps = (epochs - initiate_decay) * train_steps
    if schedule_step == 'global_iterations':
        it = epoch * train_steps + step_in_epoch
    else:
        it = (epoch - initiate_decay) * train_steps + step_in_epoch
    return initial_lr * (1.0 - min(it, decay_steps) / decay_steps)


def discriminator_noise(init_noise, epoch, no_noise_epoch):
    """updateDiscriminatorNoise (custom_callback.py:399-424): linear decay to zero at epoch NO_NOISE; NO_NOISE == 0
    switches the noise off from the first epoch."""
    decay_rate = 1.0 if no_noise_epoch == 0 else epoch / no_noise_epoch
    return max(init_noise * (1.0 - decay_rate), 0.0)


def learning_rate_resumed(initial_lr, resume_epoch, epoch, step_in_epoch, epochs, initiate_decay, train_steps,
                          schedule_step='global_iterations'):
    """set_learning_rate's second branch (custom_callback.py:365-397), taken once when a checkpoint was loaded at an epoch
    past INITIATE_LR_DECAY: the decay restarts from INITIAL_LR / (EPOCHS - INITIATE) * (EPOCHS - resume_epoch) and runs
    over (EPOCHS - INITIATE - resume_epoch) * train_steps steps -- the reference subtracts BOTH, so the window is shorter
    than the epochs left (and empty or negative for resume_epoch >= EPOCHS - INITIATE; TP: Keras' PolynomialDecay then
    divides by decay_steps: a negative window yields end_learning_rate = 0, an empty one 0/0, which this restatement refuses).  Step: as in learning_rate (TP: the restored
    optimizer.iterations, i.e. the global count)."""
    start = initial_lr / (epochs - initiate_decay) * (epochs - resume_epoch)
    decay_steps = (epochs - initiate_decay - resume_epoch) * train_steps
    if decay_steps < 0 and schedule_step == 'global_iterations':
        return 0.0          # TP: PolynomialDecay: min(step, decay_steps) / decay_steps = 1 for a negative window => end_learning_rate
    if decay_steps <= 0:
        raise ValueError('reference schedule undefined: decay_steps <= 0')
    if schedule_step == 'global_iterations':
        it = epoch * train_steps + step_in_epoch
    else:
        it = (epoch - resume_epoch) * train_steps + step_in_epoch
    return start * (1.0 - min(it, decay_steps) / decay_steps)
