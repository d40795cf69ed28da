"""GPU parity of the resident-volume data pipeline with oracle/data_oracle.py.
The crop/flip/rot90 gather and the crop maximum move fp32 values without arithmetic: the bar is BIT-EXACT.
The on-the-fly imaging normalisation is fp32 arithmetic with a correctly rounded division: tolerance 1 ulp-scale
(abs 2.4e-7 on values in [-1, 1])."""
import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as do  # noqa: E402


def _dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('shape,patch,C', [((20, 24, 28), (8, 12, 12), 1), ((9, 17, 17), (9, 17, 17), 1),
                                           ((12, 16, 16), (5, 7, 7), 3), ((66, 70, 70), (64, 64, 64), 1)])
def test_crop_augment_bit_exact_all_draws(shape, patch, C):
    from van_gan_amd import data
    rng = np.random.default_rng(sum(shape))
    vol = rng.standard_normal(shape + (C,)).astype(np.float32)
    dv = torch.from_numpy(vol).to(_dev())
    out = torch.empty(patch + (C,), device=_dev())
    for lr, ud, k in itertools.product((False, True), (False, True), (-1, 0, 1, 2, 3)):
        o = tuple(int(rng.integers(0, shape[a] - patch[a] + 1)) for a in range(3))
        out.fill_(float('nan'))
        data.crop_augment(dv, o, patch, lr, ud, k, out)
        want = do.spatial_augmentation(do.crop(vol, o, patch), lr, ud, k)
        got = out.cpu().numpy()
        assert got.tobytes() == np.ascontiguousarray(want).tobytes(), (lr, ud, k, o)


def test_crop_augment_rejects_bad_arguments():
    from van_gan_amd import data
    dv = torch.zeros(8, 8, 10, 1, device=_dev())
    out = torch.empty(4, 4, 6, 1, device=_dev())
    with pytest.raises(RuntimeError):
        data.crop_augment(dv, (5, 0, 0), (4, 4, 6), False, False, 0, out)       # crop leaves the volume
    with pytest.raises(RuntimeError):
        data.crop_augment(dv, (0, 0, 0), (4, 4, 6), False, False, 1, out)       # quarter turn of a non-square plane


def test_crop_max_and_seg_rejection():
    from van_gan_amd import data
    rng = np.random.default_rng(5)
    vol = (rng.random((40, 40, 40, 1)) * 0.5).astype(np.float32)                # everything below the 0.8 threshold
    vol[30:34, 30:34, 30:34, 0] = 1.0                                           # one bright vessel blob
    dv = torch.from_numpy(vol).to(_dev())
    mx = torch.zeros(1, device=_dev())
    for _ in range(20):
        o = tuple(int(rng.integers(0, 40 - 16 + 1)) for _ in range(3))
        data.crop_max(dv, o, (16, 16, 16), mx)
        assert mx.item() == do.crop(vol, o, (16, 16, 16)).max()
    pipe = data.DataPipeline([dv], [dv], (16, 16, 16), 1, seed=3)
    for _ in range(10):
        o, tries = pipe.seg_origin(dv)
        assert do.crop(vol, o, (16, 16, 16)).max() >= do.SEG_THRESH
    # a volume with nothing bright: the loop gives up after MAX_RECROPS and still returns an in-range origin
    dark = torch.zeros(20, 20, 20, 1, device=_dev())
    pipe2 = data.DataPipeline([dark], [dark], (16, 16, 16), 1, seed=4)
    o, tries = pipe2.seg_origin(dark)
    assert tries == do.MAX_RECROPS and all(0 <= o[a] <= 4 for a in range(3))


def test_imaging_otf_matches_oracle():
    from van_gan_amd import data
    b = np.random.default_rng(9).standard_normal((2, 16, 16, 16, 1)).astype(np.float32) * 3 + 1
    got = data.imaging_otf(torch.from_numpy(b).to(_dev())).cpu().numpy()
    want = do.process_imaging_otf(b)
    np.testing.assert_allclose(got, want, rtol=0, atol=2.4e-7)
    assert got.min() == -1.0 and got.max() == 1.0


def test_next_batch_shapes_and_ranges():
    from van_gan_amd import data
    rng = np.random.default_rng(1)
    img = [torch.from_numpy(rng.standard_normal((24, 40, 40, 1)).astype(np.float32)).to(_dev()) for _ in range(2)]
    seg = [torch.from_numpy((rng.random((24, 40, 40, 1)) > 0.9).astype(np.float32)).to(_dev()) for _ in range(3)]
    pipe = data.DataPipeline(img, seg, (16, 32, 32), 2, seed=0)
    for _ in range(3):
        rI, rS = pipe.next_batch()
        assert rI.shape == (2, 16, 32, 32, 1) and rS.shape == (2, 16, 32, 32, 1)
        assert rI.amin(dim=(1, 2, 3, 4)).eq(-1).all() and rI.amax(dim=(1, 2, 3, 4)).eq(1).all()
        assert set(rS.unique().tolist()) <= {0.0, 1.0} and rS.amax() == 1.0


def test_fit_end_to_end_small(tmp_path):
    """Resident volumes -> DataPipeline -> VanGan.train_step under the GanMonitor schedules, 3 epochs x 2 steps at 32^3,
    then a checkpoint round trip: finite losses, the scalars follow the oracle, restored weights are identical."""
    import math
    from van_gan_amd import data
    from van_gan_amd.train import GanMonitor, fit
    from van_gan_amd.vangan import VanGan
    rng = np.random.default_rng(2)
    img = [torch.from_numpy(rng.standard_normal((40, 48, 48, 1)).astype(np.float32)).to(_dev())]
    seg = [torch.from_numpy((rng.random((40, 48, 48, 1)) > 0.95).astype(np.float32)).to(_dev())]
    dims = (32, 32, 32)
    gan = VanGan(dims, batch_size=1, device='cuda:0', seed=1, layer_noise=0.9, output_dir=str(tmp_path))
    pipe = data.DataPipeline(img, seg, dims, 1, seed=0)
    mon = GanMonitor(EPOCHS=3, INITIATE_LR_DECAY=1, INITIAL_LR=2e-4, train_steps=2, NO_NOISE=2)
    hist = fit(gan, pipe, mon, val_ds=pipe, val_steps=1)
    assert len(hist) == 3
    for h in hist:
        assert all(math.isfinite(v) for v in h['train'].values()) and all(math.isfinite(v) for v in h['val'].values())
        assert h['noise'] == do.discriminator_noise(0.9, h['epoch'], 2)
        assert h['lr'] == do.learning_rate(2e-4, h['epoch'], 1, 3, 1, 2)       # default: reference-exact global step
    w0 = {k: s.w.clone() for k, s in gan.stores.items()}
    # a resumed run: same seed (hence the same noise / dropout keys), state from the checkpoint of the last epoch
    gan2 = VanGan(dims, batch_size=1, device='cuda:0', seed=1, layer_noise=0.9, output_dir=str(tmp_path))
    assert not torch.equal(gan2.stores['gen_IS'].w, w0['gen_IS'])
    assert gan2.load_checkpoint(3, newpath=gan.checkpoint_dir)
    for k, s in gan2.stores.items():
        assert torch.equal(s.w, w0[k]) and s.step == gan.stores[k].step
        assert torch.equal(s.m, gan.stores[k].m) and torch.equal(s.v, gan.stores[k].v), k       # Adam slots
    assert gan2.rng_offset == gan.rng_offset > 0                  # the noise stream continues, it does not replay from 0
    # the next step of the resumed engine is the next step of the original one (same weights, slots, noise draws, rate):
    # only the order of float atomics differs between two runs, which the bf16 pipeline amplifies to its noise floor
    gan2.layer_noise, gan2.lr = gan.layer_noise, gan.lr
    rI, rS = pipe.next_batch()
    ra, rb = gan.train_step(rI, rS), gan2.train_step(rI, rS)
    for k in ra:
        assert abs(ra[k] - rb[k]) <= 3e-2 * abs(ra[k]) + 1e-6, (k, ra[k], rb[k])
    with pytest.raises(ValueError):
        VanGan(dims, batch_size=1, device='cuda:0').save_checkpoint(0)       # built without output_dir
    assert VanGan(dims, batch_size=1, device='cuda:0').load_checkpoint(1) is False
