"""CPU tests of the data-pipeline / schedule oracle (oracle/data_oracle.py) and of the host mirror's scalar schedules
(van_gan_amd/data.py).  No GPU, no reference import: the properties below are the ones the reference's code implies
(dataset.py:205-251, custom_callback.py:326-424)."""
import math

import numpy as np
import pytest

from oracle import data_oracle as do


def _vol(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_flips_and_rot_follow_tf_image_axes():
    v = _vol((3, 4, 5, 2))
    np.testing.assert_array_equal(do.spatial_augmentation(v, True, False, 0), v[:, :, ::-1])
    np.testing.assert_array_equal(do.spatial_augmentation(v, False, True, 0), v[:, ::-1])
    sq = _vol((2, 4, 4, 1))
    r = do.spatial_augmentation(sq, False, False, 1)
    # counter-clockwise quarter turn of the (Y,Z) plane: out[y,z] = in[z, Z-1-y]
    for y in range(4):
        for z in range(4):
            np.testing.assert_array_equal(r[:, y, z], sq[:, z, 3 - y])
    # four quarter turns are the identity, -1 == 3
    np.testing.assert_array_equal(do.spatial_augmentation(sq, False, False, -1), do.spatial_augmentation(sq, False, False, 3))
    x = sq
    for _ in range(4):
        x = do.spatial_augmentation(x, False, False, 1)
    np.testing.assert_array_equal(x, sq)


def test_rot_k_is_minus_one_or_zero():
    ks = {do.rot_k_from_uniform(u) for u in np.linspace(0, 0.999, 101)}
    assert ks == {-1, 0}
    assert do.rot_k_from_uniform(0.25) == -1 and do.rot_k_from_uniform(0.75) == 0 and do.rot_k_from_uniform(0.5) == 0


def test_seg_rejection_takes_first_bright_candidate():
    v = np.zeros((8, 8, 8, 1), np.float32)
    v[6, 6, 6, 0] = 1.0
    origins = [(0, 0, 0), (1, 1, 1), (4, 4, 4), (0, 0, 0)]
    out, used = do.process_seg_domain(v, origins, (4, 4, 4), False, False, 0)
    assert used == 2 and out.max() == 1.0
    # nothing bright anywhere: the loop runs out and the last candidate is used
    out, used = do.process_seg_domain(np.zeros_like(v), origins, (4, 4, 4), False, False, 0)
    assert used == len(origins) - 1 and out.max() == 0.0
    # threshold is inclusive (tf.less)
    v2 = np.full((4, 4, 4, 1), 0.8, np.float32)
    _, used = do.process_seg_domain(v2, [(0, 0, 0), (0, 0, 0)], (4, 4, 4), False, False, 0)
    assert used == 0


def test_otf_range():
    b = _vol((2, 4, 4, 4, 1), 3)
    y = do.process_imaging_otf(b)
    assert y.shape == b.shape
    np.testing.assert_allclose(y.min(axis=(1, 2, 3, 4)), -1.0)
    np.testing.assert_allclose(y.max(axis=(1, 2, 3, 4)), 1.0)


@pytest.mark.parametrize('mod', ['oracle', 'host'])
def test_schedules_known_answers(mod):
    if mod == 'oracle':
        lr, noise = do.learning_rate, do.discriminator_noise
    else:
        from van_gan_amd import data
        lr, noise = data.learning_rate, data.discriminator_noise
    # defaults of the reference: EPOCHS 200, INITIATE_LR_DECAY 100, lr 2e-4
    assert lr(2e-4, 0, 0, 200, 100, 50) == 2e-4
    assert lr(2e-4, 99, 49, 200, 100, 50) == 2e-4
    # reference-exact (default): Keras evaluates PolynomialDecay at the optimizer's global iteration count, which equals
    # decay_steps at the swap when EPOCHS - INITIATE == INITIATE  =>  the rate is 0 from epoch 100 on
    assert lr(2e-4, 100, 0, 200, 100, 50) == 0.0
    assert lr(2e-4, 150, 7, 200, 100, 50) == 0.0
    # ... and a partial decay when the decay window is longer than the constant phase: EPOCHS 10, INITIATE 2, 5 steps
    assert math.isclose(lr(2e-4, 2, 0, 10, 2, 5), 2e-4 * (1 - 10 / 40), rel_tol=1e-12)
    assert math.isclose(lr(2e-4, 5, 3, 10, 2, 5), 2e-4 * (1 - 28 / 40), rel_tol=1e-12)
    assert lr(2e-4, 8, 0, 10, 2, 5) == 0.0
    # 'since_install': the linear decay the authors presumably intended
    si = dict(schedule_step='since_install')
    assert lr(2e-4, 100, 0, 200, 100, 50, **si) == 2e-4
    assert math.isclose(lr(2e-4, 150, 0, 200, 100, 50, **si), 1e-4, rel_tol=1e-12)
    assert math.isclose(lr(2e-4, 199, 49, 200, 100, 50, **si), 2e-4 / 5000, rel_tol=1e-9)
    assert lr(2e-4, 200, 0, 200, 100, 50, **si) == 0.0
    assert noise(0.9, 0, 100) == 0.9
    assert math.isclose(noise(0.9, 50, 100), 0.45)
    assert noise(0.9, 100, 100) == 0.0 and noise(0.9, 150, 100) == 0.0
    assert noise(0.9, 0, 0) == 0.0


def test_host_rot_k_matches_oracle():
    from van_gan_amd import data
    for u in np.linspace(0, 0.999, 57):
        assert data.rot_k_from_uniform(u) == do.rot_k_from_uniform(u)


class _FakeGan:
    """Stands in for VanGan in the host-loop tests: records the scalars the monitor wrote before each step."""

    def __init__(self):
        self.lr, self.layer_noise, self.current_epoch, self.checkpoint_loaded = 2e-4, 0.9, 0, False
        self.seen, self.saved = [], []

    def distributed_train_step(self, x, y):
        self.seen.append(('train', self.current_epoch, self.lr, self.layer_noise))
        return {'gen_IS_loss': float(x), 'disc_S_loss': float(y)}

    def distributed_test_step(self, x, y):
        self.seen.append(('test', self.current_epoch, self.lr, self.layer_noise))
        return {'gen_IS_loss': 0.0}

    def save_checkpoint(self, epoch):
        self.saved.append(epoch)


@pytest.mark.parametrize('mode', ['global_iterations', 'since_install'])
def test_fit_loop_applies_schedules_per_step_and_checkpoints(mode):
    from van_gan_amd.train import GanMonitor, fit
    gan = _FakeGan()
    E, I, T = 6, 2, 3
    mon = GanMonitor(EPOCHS=E, INITIATE_LR_DECAY=I, INITIAL_LR=2e-4, train_steps=T, NO_NOISE=4, schedule_step=mode)
    batches = [(1.0, 2.0)] * 1000
    hist = fit(gan, batches, mon, val_ds=batches, val_steps=2)
    tr = [s for s in gan.seen if s[0] == 'train']
    assert len(tr) == E * T and len(gan.seen) == E * (T + 2)
    k = 0
    for epoch in range(E):
        for step in range(T):
            _, ep, lr, nz = tr[k]; k += 1
            assert ep == epoch
            assert lr == do.learning_rate(2e-4, epoch, step, E, I, T, mode)
            assert nz == do.discriminator_noise(0.9, epoch, 4)
    assert gan.saved == [1, 3, 5]                       # epoch % 2 == 1 or the last epoch (main.py:230)
    assert hist[-1]['train']['gen_IS_loss'] == 1.0 and len(hist) == E


@pytest.mark.parametrize('mode', ['global_iterations', 'since_install'])
def test_resumed_schedule_matches_oracle(mode):
    from van_gan_amd.train import GanMonitor
    gan = _FakeGan()
    gan.checkpoint_loaded = True
    E, I, T, R = 20, 4, 5, 10
    mon = GanMonitor(EPOCHS=E, INITIATE_LR_DECAY=I, INITIAL_LR=2e-4, train_steps=T, NO_NOISE=4, schedule_step=mode)
    for epoch in range(R, R + 3):
        for step in range(T):
            got = mon.set_learning_rate(gan, epoch, step)
            assert got == do.learning_rate_resumed(2e-4, R, epoch, step, E, I, T, mode)
    assert gan.checkpoint_loaded is False
    assert math.isclose(do.learning_rate_resumed(2e-4, R, R, 0, E, I, T, 'since_install'), 2e-4 / 16 * 10)
    # global count: the restored optimizer.iterations (R * T = 50) already exceeds the 30-step window => 0
    assert do.learning_rate_resumed(2e-4, R, R, 0, E, I, T) == 0.0
    with pytest.raises(ValueError):
        do.learning_rate_resumed(2e-4, 16, 16, 0, E, I, T)
    gan2 = _FakeGan(); gan2.checkpoint_loaded = True
    if mode == 'global_iterations':      # negative window: TF's PolynomialDecay returns end_learning_rate (0), the run carries on
        with pytest.warns(RuntimeWarning):
            assert GanMonitor(E, I, 2e-4, T, 4).set_learning_rate(gan2, 17, 0) == 0.0
        assert do.learning_rate_resumed(2e-4, 17, 17, 0, E, I, T) == 0.0


def test_product_synth_generator_equals_the_oracles():
    """bench.py / smoke inputs come from van_gan_amd.synth (the product must not import oracle/); the parity tests draw
    theirs from the oracle's copy: same volumes, bit for bit."""
    import torch
    from oracle import vangan_oracle as O
    from van_gan_amd.synth import synth_volumes
    for B, dims, seed in ((1, (32, 32, 32), 1234), (2, (16, 32, 48), 7)):
        a, b = synth_volumes(B, *dims, seed=seed), O.synth_volumes(B, *dims, seed=seed)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert a[0].shape == (B,) + dims + (1,) and float(a[0].min()) == -1.0 and float(a[0].max()) == 1.0
        assert set(a[1].unique().tolist()) == {-1.0, 1.0}
