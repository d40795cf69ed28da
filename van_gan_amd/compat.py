"""Reference-shaped constructor: ``VanGan(args, strategy, ..., gen_i2s=..., gen_s2i=...)`` as main.py:196-200 calls it
(vangan.py:20-245), over the MI355X engine (``van_gan_amd.vangan.VanGan``).

A maintainer of the reference swaps ``from vangan import VanGan`` for ``from van_gan_amd.compat import VanGan`` and keeps
the rest of main.py: the dataset, ``train(ds, gan, ...)`` (vangan.py:510-550) and ``GanMonitor`` only touch the surface
mirrored here.  What is read from ``args`` is exactly what the reference reads (vangan.py:37-58,80): ``N_DEVICES,
INPUT_IMG_SIZE, CHANNELS, GLOBAL_BATCH_SIZE, DIMENSIONS, SUBVOL_PATCH_SIZE, train_steps, BATCH_SIZE, output_dir``.

Differences that are deliberate and loud:
  * built: 3-D, single channel, ``gen_i2s == gen_s2i == 'resUnet'`` (the default path) or ``== 'resnet'``, ``wasserstein`` False
    or True (True = what the reference trains once its step is traced: Wasserstein losses, Dense head, the optimizers of
    vangan.py:195-203; its gradient penalty never reaches a weight and ``ncritic`` is frozen at trace time -- DESIGN.md section 8),
    not semi-supervised.  Unknown generator names raise the reference's own ``ValueError`` (vangan.py:124,164); known but
    unbuilt variants ('vnet', mixed generator pairs, 2-D) raise ``NotImplementedError`` naming SURVEY section 8(f)4;
  * ``N_DEVICES == 0`` (what ``len(GPUs)`` gives on a box TensorFlow sees no GPU on, main.py:62-105) is read as 1: the
    reference would divide by zero in cycle_seg_loss (loss_functions.py:226) and build with GLOBAL_BATCH_SIZE 0;
  * ``strategy`` is accepted and ignored (None is fine): data parallelism is one process per GPU with a
    ``torch.distributed`` process group (``process_group=``), not an in-process MirroredStrategy.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

KNOWN_GENERATORS = ('resnet', 'resUnet', 'vnet')       # vangan.py:88-124, 127-164


def engine_kwargs_from_args(args, lambda_cycle=10.0, lambda_identity=5, lambda_reconstruction=5, lambda_topology=5,
                            gen_i2s='resnet', gen_s2i='resnet', semi_supervised=False, wasserstein=False,
                            ncritic=5, gp_weight=10.0) -> Dict:
    """Validate the reference's constructor arguments and translate them into the engine's (pure host logic)."""
    if gen_i2s not in KNOWN_GENERATORS:
        raise ValueError('IS Generator type not recognised')          # vangan.py:124
    if gen_s2i not in KNOWN_GENERATORS:
        raise ValueError('SI Generator type not recognised')          # vangan.py:164
    if gen_i2s != gen_s2i or gen_i2s == 'vnet':
        raise NotImplementedError("train_step is built for gen_i2s == gen_s2i == 'resUnet' (the default, main.py:196-200) and, SURVEY "
                                  "section 8(f)4, for gen_i2s == gen_s2i == 'resnet' (generator.py:7-73: van_gan_amd.nets.ResNetGenerator, "
                                  "tests/test_gpu_resnet.py); 'vnet' and mixed pairs are not built")
    if semi_supervised:
        raise NotImplementedError('semi_supervised is never enabled by main.py and is not built')
    if int(args.DIMENSIONS) != 3:
        raise NotImplementedError('DIMENSIONS must be 3 (the 2-D branch of the reference is not on the hot path)')
    if int(args.CHANNELS) != 1:
        raise NotImplementedError('CHANNELS must be 1 (single-channel imaging and label volumes)')
    patch = tuple(int(v) for v in args.SUBVOL_PATCH_SIZE)[:3]
    if len(patch) != 3:
        raise ValueError('SUBVOL_PATCH_SIZE must have three entries')
    n_dev = int(args.N_DEVICES)
    if n_dev < 0:
        raise ValueError('N_DEVICES must be >= 0')
    n_dev = max(n_dev, 1)                                             # 0 on a box without visible GPUs -> one replica
    batch = int(args.BATCH_SIZE)
    gbs = int(args.GLOBAL_BATCH_SIZE)
    if gbs <= 0:
        gbs = batch * n_dev                                           # N_DEVICES 0 made it 0 in main.py:70-71
    return dict(subvol_patch_size=patch, batch_size=batch, global_batch_size=gbs, n_devices=n_dev,
                lambda_cycle=float(lambda_cycle), lambda_reconstruction=float(lambda_reconstruction),
                lambda_topology=float(lambda_topology), output_dir=getattr(args, 'output_dir', None), generator=gen_i2s,
                # wasserstein=True: what the reference trains once distributed_train_step is traced -- Wasserstein losses, the Dense head,
                # the optimizers of vangan.py:195-203, generators every step; its gradient penalty never reaches a weight and n-critic is
                # frozen at trace time: ncritic / gp_weight are accepted and inert
                **(dict(wasserstein=True, lr=1e-4, beta_1=0.0, beta_2=0.9, clipnorm=0.0) if wasserstein else {}))


def to_device_volume(t, device) -> torch.Tensor:
    """tf.Tensor / numpy array / torch tensor [B,D,H,W,1] -> fp32 torch tensor in HBM."""
    if isinstance(t, torch.Tensor):
        return t.to(device=device, dtype=torch.float32)
    if hasattr(t, 'numpy') and not isinstance(t, np.ndarray):
        t = t.numpy()
    return torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(device)


def _tf_layer_class(name: str):
    """tf.keras.layers.<name> when the CALLER has imported TensorFlow (the reference's GanMonitor tests
    `isinstance(layer, tf.keras.layers.GaussianNoise)`, custom_callback.py:423); None otherwise.  Never imports TensorFlow."""
    import sys
    tf = sys.modules.get('tensorflow')
    try:
        return getattr(tf.keras.layers, name) if tf is not None else None
    except AttributeError:
        return None


class LayerShim:
    """One entry of `model.layers`: only its kind is visible (`name`)."""

    def __init__(self, name: str):
        self.name = name


class GaussianNoiseShim(LayerShim):
    """A GaussianNoise layer of a discriminator (discriminator.py:52,108; building_blocks.py:180) as GanMonitor sees it
    (custom_callback.py:413-424): `layer.stddev = noise` sets the standard deviation the NEXT step draws its noise with.  The
    engine has one value for all noise layers of both discriminators (the reference writes the same number into every one of them).
    `isinstance(layer, tf.keras.layers.GaussianNoise)` holds when the caller has TensorFlow loaded: `__class__` reports that type."""

    def __init__(self, eng, name: str):
        super().__init__(name)
        self._eng = eng

    stddev = property(lambda self: self._eng.layer_noise, lambda self, v: setattr(self._eng, 'layer_noise', max(float(v), 0.0)))

    @property
    def __class__(self):
        return _tf_layer_class('GaussianNoise') or GaussianNoiseShim


class ModelShim:
    """A network of the engine behind the tf.keras.Model surface the reference's callers use: `gen(x, training=False)`
    (custom_callback.py:174-175, 229-262), `model.layers` (:422-424), `model.save(path)` (:42-45).  Every other attribute is the
    engine network's own (van_gan_amd.nets.ResUNet / ResNetGenerator / PatchGAN)."""

    def __init__(self, eng, key: str):
        self._eng, self._key = eng, key
        self._net = getattr(eng, key)

    def __getattr__(self, name):
        return getattr(self._net, name)

    @property
    def layers(self):
        if not self._key.startswith('disc'):
            return [LayerShim('conv3d')]
        # discriminator.py:50-117 in layer order: pad, input noise, conv0, IN, LeakyReLU, 3 x downsample (each with its GaussianNoise,
        # building_blocks.py:180), noise, output conv
        out = [LayerShim('reflection_padding3d'), GaussianNoiseShim(self._eng, 'gaussian_noise'), LayerShim('conv3d'),
               LayerShim('instance_normalization'), LayerShim('leaky_re_lu')]
        for i in range(3):
            out += [GaussianNoiseShim(self._eng, 'gaussian_noise_%d' % (i + 1)), LayerShim('conv3d_%d' % (i + 1))]
        return out + [GaussianNoiseShim(self._eng, 'gaussian_noise_4'), LayerShim('conv3d_4')]

    def __call__(self, x, training: bool = False):
        """Generator forward on [B,D,H,W,1] (numpy / tf.Tensor / torch); returns the same kind of array the caller gave (tf.Tensor
        -> numpy, which is what GanMonitor indexes and adds into its numpy accumulator)."""
        if not self._key.startswith('gen'):
            raise NotImplementedError('only the generators are callable through the compat surface (GanMonitor calls gen(x, training=False))')
        if training and getattr(self._eng, 'generator', 'resUnet') != 'resUnet':
            raise NotImplementedError("training=True applies SpatialDropout3D in the ResNet generator: use train_step")
        was_torch = isinstance(x, torch.Tensor)
        y = self._eng.generate(self._key, to_device_volume(x, self._eng.device))
        return y if was_torch else y.cpu().numpy()

    def save(self, path: str, **_):
        """model.save(path) of custom_callback.py:42-45: the network's weights by parameter name (Keras layouts, fp32) as
        `<path>.pt` -- a TF SavedModel cannot be written without TensorFlow; VanGan.load_weights({net: torch.load(...)}) reads it."""
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
        w = self._eng.export_weights()[self._key]
        torch.save({k: v.cpu() for k, v in w.items()}, path + '.pt')
        return path + '.pt'


class OptimizerShim:
    """tf.keras.optimizers.Adam as GanMonitor.set_learning_rate uses it (custom_callback.py:343-397): `.lr` (alias
    `.learning_rate`) takes a float or a schedule -- any callable of the step, e.g. a PolynomialDecay -- which the engine evaluates at
    this optimizer's `.iterations` when it enqueues the network's Adam step, as Keras' `_decayed_lr` does."""

    def __init__(self, eng, key: str):
        self._eng, self._key = eng, key

    def _get(self):
        v = self._eng.lrs.get(self._key)
        return self._eng.lr if v is None else v

    def _set(self, v):
        self._eng.lrs[self._key] = v if callable(v) else float(v)

    lr = property(_get, _set)
    learning_rate = property(_get, _set)

    @property
    def iterations(self) -> int:
        stores = getattr(self._eng, 'stores', None)
        return int(stores[self._key].step) if stores else 0


class VanGan:
    """vangan.py:20-550 surface used by main.py, train() and GanMonitor: constructor, train_step / test_step,
    distributed_train_step / distributed_test_step, reduce_dict, save_checkpoint / load_checkpoint, gen_IS / gen_SI /
    disc_I / disc_S, layer_noise, current_epoch, checkpoint_loaded, and the inert n-critic fields."""

    def __init__(self, args, strategy=None, lambda_cycle=10.0, lambda_identity=5, lambda_reconstruction=5,
                 lambda_topology=5, gen_i2s='resnet', gen_s2i='resnet', semi_supervised=False, wasserstein=False,
                 ncritic=5, gp_weight=10.0, *, device: str = 'cuda:0', process_group=None, seed: int = 0,
                 engine_factory: Optional[Callable] = None, **engine_kw):
        kw = engine_kwargs_from_args(args, lambda_cycle, lambda_identity, lambda_reconstruction, lambda_topology, gen_i2s,
                                     gen_s2i, semi_supervised, wasserstein, ncritic, gp_weight)
        kw.update(engine_kw)
        if engine_factory is None:
            from .vangan import VanGan as engine_factory              # needs an MI355X; raises without one
        self.strategy = strategy
        self.n_devices, self.global_batch_size, self.batch_size = kw['n_devices'], kw['global_batch_size'], kw['batch_size']
        self.img_size = getattr(args, 'INPUT_IMG_SIZE', None)
        self.channels, self.dims = int(args.CHANNELS), int(args.DIMENSIONS)
        self.subvol_patch_size = kw['subvol_patch_size'] + (self.channels,)
        self.seg_subvol_patch_size = kw['subvol_patch_size'] + (1,)
        self.train_steps = getattr(args, 'train_steps', None)
        self.lambda_cycle, self.lambda_identity = lambda_cycle, lambda_identity
        self.lambda_reconstruction, self.lambda_topology = lambda_reconstruction, lambda_topology
        self.gen_i2s_typ, self.gen_s2i_typ = gen_i2s, gen_s2i
        self.semi_supervised, self.wasserstein = semi_supervised, wasserstein
        self.ncritic, self.icritic, self.initModel, self.updateGen, self.gp_weight = ncritic, 1, True, True, gp_weight
        self.eng = engine_factory(device=device, process_group=process_group, seed=seed, **kw)
        # the networks and optimizers as the reference's GanMonitor touches them (custom_callback.py:42-45,174-175,343-365,413-424)
        self.gen_IS, self.gen_SI = ModelShim(self.eng, 'gen_IS'), ModelShim(self.eng, 'gen_SI')
        self.disc_I, self.disc_S = ModelShim(self.eng, 'disc_I'), ModelShim(self.eng, 'disc_S')
        if not hasattr(self.eng, 'lrs'):
            self.eng.lrs = {}
        self.gen_I_optimizer, self.gen_S_optimizer = OptimizerShim(self.eng, 'gen_IS'), OptimizerShim(self.eng, 'gen_SI')     # vangan.py:425-430
        self.disc_I_optimizer, self.disc_S_optimizer = OptimizerShim(self.eng, 'disc_I'), OptimizerShim(self.eng, 'disc_S')   # :433-438
        self.checkpoint_dir = getattr(self.eng, 'checkpoint_dir', None)
        # vangan.py:77: a constructor constant in the reference -- GanMonitor passes it as the INITIAL noise of its decay every epoch
        # (custom_callback.py:443-444) and writes the decayed value into the layers, never back into this attribute
        self._layer_noise0 = float(self.eng.layer_noise)

    def _set_layer_noise(self, v):
        """The build's own train.GanMonitor writes the current value here; both the constant and the engine follow."""
        self._layer_noise0 = float(v)
        self.eng.layer_noise = float(v)

    def _set_lr(self, v):
        self.eng.lr = float(v)
        self.eng.lrs = {}                 # a rate written for the whole model replaces per-optimizer schedules

    # scalars GanMonitor reads and writes (custom_callback.py:343-365,422-424,441-444)
    layer_noise = property(lambda self: self._layer_noise0, _set_layer_noise)
    lr = property(lambda self: self.eng.lr, _set_lr)
    current_epoch = property(lambda self: self.eng.current_epoch, lambda self, v: setattr(self.eng, 'current_epoch', int(v)))
    checkpoint_loaded = property(lambda self: self.eng.checkpoint_loaded,
                                 lambda self, v: setattr(self.eng, 'checkpoint_loaded', bool(v)))
    stores = property(lambda self: getattr(self.eng, 'stores', None))

    def _dev(self, t):
        return to_device_volume(t, self.eng.device)

    def train_step(self, real_I, real_S):
        return self.eng.train_step(self._dev(real_I), self._dev(real_S))

    def test_step(self, real_I, real_S):
        return self.eng.test_step(self._dev(real_I), self._dev(real_S))

    def distributed_train_step(self, x, y):
        return self.eng.distributed_train_step(self._dev(x), self._dev(y))

    def distributed_test_step(self, x, y):
        return self.eng.distributed_test_step(self._dev(x), self._dev(y))

    def reduce_dict(self, d):
        return self.eng.reduce_dict(d)

    def save_checkpoint(self, epoch):
        return self.eng.save_checkpoint(epoch)

    def load_checkpoint(self, epoch=None, expect_partial=False, newpath=None):
        return self.eng.load_checkpoint(epoch, newpath)
