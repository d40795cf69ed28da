"""Reference-shaped constructor (van_gan_amd/compat.py, vangan.py:20-245): argument translation and error behaviour on
the CPU with an injected engine factory; one real step through it on the GPU."""
import argparse

import numpy as np
import pytest
import torch


def _args(**over):
    a = argparse.Namespace(N_DEVICES=4, INPUT_IMG_SIZE=(1, 512, 512, 140, 1), CHANNELS=1, GLOBAL_BATCH_SIZE=12, DIMENSIONS=3,
                           SUBVOL_PATCH_SIZE=(128, 128, 128), train_steps=50, BATCH_SIZE=3, output_dir=None)
    for k, v in over.items():
        setattr(a, k, v)
    return a


class _FakeStore:
    step = 0


class _FakeEngine:
    def __init__(self, **kw):
        self.kw = kw
        self.device = 'cpu'
        self.layer_noise, self.lr, self.current_epoch, self.checkpoint_loaded = 0.1, 2e-4, 0, False
        self.gen_IS = self.gen_SI = self.disc_I = self.disc_S = object()
        self.calls = []
        self.lrs = {}
        self.dims = tuple(kw.get('subvol_patch_size', (4, 4, 4)))
        self.stores = {k: _FakeStore() for k in ('gen_IS', 'gen_SI', 'disc_I', 'disc_S')}
        self.generated = []

    def generate(self, key, x):
        self.generated.append((key, tuple(x.shape), x.dtype))
        return torch.tanh(x)

    def export_weights(self):
        return {k: {'stem.conv1.w': torch.ones(2, 2)} for k in self.stores}

    def distributed_train_step(self, x, y):
        self.calls.append((x, y))
        return {'gen_IS_loss': 1.0}


def test_reference_arguments_are_translated():
    from van_gan_amd.compat import VanGan
    g = VanGan(_args(), None, gen_i2s='resUnet', gen_s2i='resUnet', engine_factory=_FakeEngine)
    kw = g.eng.kw
    assert kw['subvol_patch_size'] == (128, 128, 128) and kw['batch_size'] == 3 and kw['global_batch_size'] == 12
    assert kw['n_devices'] == 4 and kw['lambda_cycle'] == 10.0 and kw['lambda_topology'] == 5.0
    assert g.subvol_patch_size == (128, 128, 128, 1) and g.train_steps == 50 and g.icritic == 1 and g.ncritic == 5
    # GanMonitor writes these through the wrapper (custom_callback.py:343-365, 422-424)
    g.layer_noise = 0.05; g.current_epoch = 7; g.checkpoint_loaded = True
    assert g.eng.layer_noise == 0.05 and g.eng.current_epoch == 7 and g.eng.checkpoint_loaded is True
    # numpy / tf-like inputs become fp32 torch tensors on the engine's device
    x = np.zeros((1, 4, 4, 4, 1), np.float64)

    class TfLike:
        def numpy(self):
            return x
    g.distributed_train_step(x, TfLike())
    a, b = g.eng.calls[0]
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and tuple(b.shape) == (1, 4, 4, 4, 1)


def test_cpu_box_n_devices_zero_means_one_replica():
    """main.py on a box without visible GPUs: N_DEVICES = len(GPUs) = 0 and GLOBAL_BATCH_SIZE = 0."""
    from van_gan_amd.compat import engine_kwargs_from_args
    kw = engine_kwargs_from_args(_args(N_DEVICES=0, GLOBAL_BATCH_SIZE=0, BATCH_SIZE=2), gen_i2s='resUnet', gen_s2i='resUnet')
    assert kw['n_devices'] == 1 and kw['global_batch_size'] == 2


def test_error_behaviour_follows_the_reference():
    from van_gan_amd.compat import VanGan, engine_kwargs_from_args
    with pytest.raises(ValueError, match='IS Generator type not recognised'):          # vangan.py:124
        VanGan(_args(), None, gen_i2s='unet++', gen_s2i='resUnet', engine_factory=_FakeEngine)
    with pytest.raises(ValueError, match='SI Generator type not recognised'):          # vangan.py:164
        VanGan(_args(), None, gen_i2s='resUnet', gen_s2i='nope', engine_factory=_FakeEngine)
    # the constructor defaults of the reference select 'resnet' (main.py overrides them): built since round 4
    assert engine_kwargs_from_args(_args())['generator'] == 'resnet'
    assert engine_kwargs_from_args(_args(), gen_i2s='resUnet', gen_s2i='resUnet')['generator'] == 'resUnet'
    with pytest.raises(NotImplementedError):
        VanGan(_args(), None, gen_i2s='vnet', gen_s2i='vnet', engine_factory=_FakeEngine)
    with pytest.raises(NotImplementedError):
        VanGan(_args(), None, gen_i2s='resnet', gen_s2i='resUnet', engine_factory=_FakeEngine)
    # wasserstein=True selects what the reference trains once its step is traced: Wasserstein losses, Dense head,
    # the optimizers of vangan.py:195-203; the gradient penalty / n-critic arguments are accepted and inert
    kw = engine_kwargs_from_args(_args(), gen_i2s='resUnet', gen_s2i='resUnet', wasserstein=True, ncritic=5, gp_weight=10.0)
    assert kw['wasserstein'] is True and kw['lr'] == 1e-4 and kw['beta_1'] == 0.0 and kw['beta_2'] == 0.9 and kw['clipnorm'] == 0.0
    with pytest.raises(NotImplementedError):
        engine_kwargs_from_args(_args(DIMENSIONS=2), gen_i2s='resUnet', gen_s2i='resUnet')


@pytest.mark.gpu
def test_real_engine_through_the_reference_constructor(tmp_path):
    from van_gan_amd.compat import VanGan
    from van_gan_amd.vangan import RESULT_KEYS
    from van_gan_amd.synth import synth_volumes
    a = _args(N_DEVICES=0, GLOBAL_BATCH_SIZE=0, BATCH_SIZE=1, SUBVOL_PATCH_SIZE=(32, 32, 32), output_dir=str(tmp_path))
    g = VanGan(a, None, gen_i2s='resUnet', gen_s2i='resUnet')
    rI, rS = synth_volumes(1, 32, 32, 32, seed=3)
    res = g.distributed_train_step(rI.numpy(), rS.numpy())
    assert list(res) == RESULT_KEYS and all(np.isfinite(v) for v in res.values())
    g.save_checkpoint(0)
    assert g.load_checkpoint(1)


def test_unbuildable_reference_variants_raise_like_the_reference():
    """ResUNet(upsample_mode='deconv') is a shape error in the reference itself (resunet_model.py:168-181: reflect pad + k2 s2
    Conv3DTranspose gives 2(S+2) voxels, the skip tensor has 2S): the constructor raises, as Keras' concatenate does."""
    from van_gan_amd.nets import ParamStore, ResUNet, gen_param_specs
    st = ParamStore(gen_param_specs(), 'cpu')
    with pytest.raises(ValueError, match='deconv'):
        ResUNet(st, (32, 32, 32), upsample_mode='deconv')
    with pytest.raises(ValueError):
        ResUNet(st, (32, 32, 32), upsample_mode='bilinear')
    with pytest.raises(ValueError):
        ResUNet(st, (48, 40, 32))                       # spatial dims must be multiples of 16 (4 stride-2 stages)


def test_resnet_generator_specs_match_the_oracle():
    """The non-default ResNet generator exchanges weights with its oracle restatement 1:1: same names, order, shapes,
    25 176 897 parameters (generator.py:7-73 with filters=32, 3 + 6 + 3 blocks as vangan.py:127-134 configures it)."""
    from oracle import vangan_oracle as O
    from van_gan_amd.nets import resnet_param_specs
    a, b = resnet_param_specs(), O.resnet_param_specs()
    assert [(n, tuple(s)) for n, s, _ in a] == [(n, tuple(s)) for n, s, _ in b]
    assert O.n_params(b) == 25176897


def test_surface_the_reference_ganmonitor_touches(tmp_path, monkeypatch):
    """Replays, attribute access by attribute access, what the reference's GanMonitor does to the model
    (custom_callback.py:42-45 save_model, :174-175 gen(arr[None], training=False)[0], :343-365 optimizer.lr = PolynomialDecay,
    :413-424 GaussianNoise.stddev over disc.layers, :441-444 on_epoch_start) against compat.VanGan over a fake engine -- with a
    stand-in `tensorflow` module in sys.modules, since the reference tests `isinstance(layer, tf.keras.layers.GaussianNoise)`."""
    import sys
    import types
    from van_gan_amd.compat import VanGan

    class GaussianNoise:                       # tf.keras.layers.GaussianNoise
        pass

    class PolynomialDecay:                     # tf.keras.optimizers.schedules.PolynomialDecay(power=1, end 0)
        def __init__(self, initial_learning_rate, decay_steps, end_learning_rate, power):
            self.lr0, self.n, self.end = initial_learning_rate, decay_steps, end_learning_rate

        def __call__(self, step):
            return (self.lr0 - self.end) * (1 - min(step, self.n) / self.n) + self.end

    tf = types.SimpleNamespace(keras=types.SimpleNamespace(
        layers=types.SimpleNamespace(GaussianNoise=GaussianNoise),
        optimizers=types.SimpleNamespace(schedules=types.SimpleNamespace(PolynomialDecay=PolynomialDecay))))
    monkeypatch.setitem(sys.modules, 'tensorflow', tf)
    args = _args(EPOCHS=200, INITIATE_LR_DECAY=100, INITIAL_LR=2e-4, NO_NOISE=200, SUBVOL_PATCH_SIZE=(4, 4, 4), output_dir=str(tmp_path))
    model = VanGan(args, None, gen_i2s='resUnet', gen_s2i='resUnet', engine_factory=_FakeEngine)

    # --- set_learning_rate, the epoch the decay is installed (custom_callback.py:342-365)
    for opt in (model.gen_I_optimizer, model.gen_S_optimizer, model.disc_I_optimizer, model.disc_S_optimizer):
        assert opt.lr == 2e-4 and opt.iterations == 0
        opt.lr = tf.keras.optimizers.schedules.PolynomialDecay(initial_learning_rate=args.INITIAL_LR,
                                                               decay_steps=(args.EPOCHS - args.INITIATE_LR_DECAY) * args.train_steps,
                                                               end_learning_rate=0, power=1)
    assert set(model.eng.lrs) == {'gen_IS', 'gen_SI', 'disc_I', 'disc_S'} and all(callable(v) for v in model.eng.lrs.values())
    assert model.eng.lrs['gen_IS'](2500) == pytest.approx(1e-4)
    model.eng.stores['disc_S'].step = 17
    assert model.disc_S_optimizer.iterations == 17
    model.gen_I_optimizer.learning_rate = 1e-5                       # Keras' other spelling; a plain float
    assert model.eng.lrs['gen_IS'] == 1e-5

    # --- updateDiscriminatorNoise over model.disc_I / model.disc_S (custom_callback.py:413-424, called from :443-444)
    def update_noise(m, init_noise, epoch):
        noise = max(init_noise * (1. - epoch / args.NO_NOISE), 0.0)
        hits = 0
        for layer in m.layers:
            if isinstance(layer, tf.keras.layers.GaussianNoise):
                layer.stddev = noise
                hits += 1
        return hits
    for epoch in (0, 50, 100):
        assert update_noise(model.disc_I, model.layer_noise, epoch) == 5          # discriminator.py:52,108 + one per downsample block
        assert update_noise(model.disc_S, model.layer_noise, epoch) == 5
        assert model.eng.layer_noise == pytest.approx(0.1 * (1 - epoch / 200))
        assert model.layer_noise == 0.1            # vangan.py:77: the constructor constant, not the decayed value (no compounding)
    assert update_noise(model.gen_IS, 0.1, 0) == 0

    # --- stitch_subvolumes' generator call (custom_callback.py:174-175)
    arr = np.linspace(-1, 1, 64, dtype=np.float64).reshape(4, 4, 4, 1)
    out = model.gen_IS(np.expand_dims(arr, axis=0), training=False)[0]
    assert isinstance(out, np.ndarray) and out.shape == (4, 4, 4, 1) and out.dtype == np.float32
    np.testing.assert_allclose(out, np.tanh(arr), rtol=1e-6)
    assert model.eng.generated == [('gen_IS', (1, 4, 4, 4, 1), torch.float32)]
    assert isinstance(model.gen_SI(torch.zeros(2, 4, 4, 4, 1)), torch.Tensor)
    with pytest.raises(NotImplementedError):
        model.disc_I(np.zeros((1, 4, 4, 4, 1)))

    # --- save_model (custom_callback.py:42-45)
    import os
    for m, tag in ((model.gen_IS, 'genAB'), (model.gen_SI, 'genBA'), (model.disc_I, 'discA'), (model.disc_S, 'discB')):
        path = m.save(os.path.join(str(tmp_path), 'checkpoints/e{epoch}_{tag}'.format(epoch=1, tag=tag)))
        assert os.path.exists(path) and 'stem.conv1.w' in torch.load(path)

    # --- the build's own monitor still works through the same object: a model-wide rate replaces the per-optimizer schedules
    model.lr = 5e-5
    assert model.eng.lr == 5e-5 and model.eng.lrs == {} and model.disc_I_optimizer.lr == 5e-5
    model.layer_noise = 0.05
    assert model.eng.layer_noise == 0.05 and model.layer_noise == 0.05
