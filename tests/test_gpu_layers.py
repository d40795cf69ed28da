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

