"""van_gan_amd -- MI355X-native engine for the VAN-GAN ``train_step`` hot path (psweens/VAN-GAN vangan.py).

The package directory is spelled with underscores (``van_gan_amd``) because a hyphen is not importable in
Python.  Importing it builds/loads ``libvangan_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/vangan_hip.h``); there is no CPU or eager-PyTorch fallback for the data path.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library cannot be built/loaded)
from .vangan import NETS, RESULT_KEYS, VanGan  # noqa: F401
