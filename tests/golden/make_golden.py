"""Generates tests/golden/train_step_32_b1.npz from the oracle in float64 (the reference cannot be run here: no
TensorFlow; see oracle/vangan_oracle.py).  Run from the repo root:  python tests/golden/make_golden.py
The fixture holds data only: inputs, 10 losses for 2 consecutive steps, selected gradients, one generated volume,
selected weights after the two Adam steps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vangan_oracle as O  # noqa: E402

SEED, DATA_SEED = 0, 1234
P = O.make_models(SEED, dtype=torch.float64)
rI, rS = O.synth_volumes(1, 32, 32, 32, seed=DATA_SEED)
out = {'seed': SEED, 'data_seed': DATA_SEED, 'real_I': rI.numpy(), 'real_S': rS.numpy()}
losses = []
state = {}
for step in range(2):
    res, grads, aux = O.train_step(P, state, rI.double(), rS.double(), O.Cfg(1, 1))
    losses.append([res[k] for k in O.RESULT_KEYS])
    if step == 0:
        for key in ('gen_IS/stem.conv1.w', 'gen_IS/out.w', 'gen_SI/dec0.cb1.conv.w', 'disc_I/conv0.w', 'disc_S/out.w'):
            net, name = key.split('/')
            out['grad:' + key] = grads[net][name].numpy().astype(np.float32)
        out['fake_S'] = aux['fake_S'].numpy().astype(np.float32)
out['losses'] = np.array(losses)
for key in ('gen_IS/out.w', 'disc_I/out.w'):
    net, name = key.split('/')
    out['w2:' + key] = P[net][name].numpy().astype(np.float32)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_step_32_b1.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes')
