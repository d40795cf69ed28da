"""Generates the fixtures of BASELINE configs 2 and 3 at sizes the CPU oracle finishes in minutes:
    tests/golden/train_step_64_b2.npz        64^3, batch 2            (config 2)
    tests/golden/train_step_64x64x32_b2.npz  64x64x32, batch 2        (config 3's non-cubic shape, scaled down 2x per axis)
from oracle/vangan_oracle.py in float32 with float64 loss accumulation (the reference itself cannot be run here: no
TensorFlow).  Run from the repo root in the BUILD container:  python tests/golden/make_golden_configs.py
Data only: seeds, the 10 result scalars of one train step, five selected gradients, the first generated volume."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vangan_oracle as O  # noqa: E402

SEED, DATA_SEED = 0, 4321
GRADS = ('gen_IS/stem.conv1.w', 'gen_IS/out.w', 'gen_SI/dec0.cb1.conv.w', 'disc_I/conv0.w', 'disc_S/out.w')
for name, dims, B in (('train_step_64_b2', (64, 64, 64), 2), ('train_step_64x64x32_b2', (64, 64, 32), 2)):
    t0 = time.time()
    P = O.make_models(SEED)
    rI, rS = O.synth_volumes(B, *dims, seed=DATA_SEED)
    res, grads, aux = O.train_step(P, {}, rI, rS, O.Cfg(B, 1), apply=False)
    out = {'seed': SEED, 'data_seed': DATA_SEED, 'dims': np.array(dims), 'batch': B,
           'losses': np.array([res[k] for k in O.RESULT_KEYS], dtype=np.float64),
           'fake_S0': aux['fake_S'][0].numpy().astype(np.float32), 'cycled_I0_mean': float(aux['cycled_I'][0].mean())}
    for key in GRADS:
        net, n = key.split('/')
        out['grad:' + key] = grads[net][n].numpy().astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes in %.0f s' % (time.time() - t0), {k: round(float(v), 5) for k, v in res.items()})
