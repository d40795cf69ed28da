"""GPU parity at TRUE LAYER SHAPES: one case per kernel variant that BASELINE configs 2, 3 and 4 run (and config 1's extra
ones), each a real layer call of the train step -- forward with its on-read InstanceNorm/activation, virtual upsample+concat,
noise, residual and statistics epilogue; data gradient on the padded grid; weight + bias gradient -- replayed with random
contents and compared with the oracle's convolution / autograd on the same bf16-rounded operands
(tests/layer_recipes.py; tolerances of tests/test_gpu_ops.py)."""
import pytest
import torch

import layer_recipes as LR

pytestmark = pytest.mark.gpu

_REPS = LR.representatives()
_KEYS = sorted(_REPS, key=lambda kv: _REPS[kv]['macs'])


def _id(kv):
    r = _REPS[kv]
    return '%s %s [%s %s]' % (kv[0], kv[1], r['config'], r['layer'])


@pytest.mark.parametrize('kv', _KEYS, ids=[_id(kv) for kv in _KEYS])
def test_layer_variant_matches_oracle(kv):
    r = _REPS[kv]
    LR.run_recipe(r['recipe'], kv[1], torch.device('cuda:0'))


def test_producer_consumer_flavour_subprocess():
    """The opt-in producer/consumer flavour of the convolution (vg_conv_pc.hip, VG_CONV_PC=1) changes which kernels the
    configurations select: run this file again in a child process with it enabled -- the parametrisation is re-derived there,
    so every conv_pc variant the BASELINE configurations would run gets its true-shape parity case too."""
    import os, subprocess, sys
    if os.environ.get('VG_CONV_PC') == '1':
        pytest.skip('already inside the child run')
    env = dict(os.environ, VG_CONV_PC='1', VG_NO_REBUILD='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu'], env=env,
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'conv_pc<' in r.stdout or ' passed' in r.stdout
