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



_KSPLIT = [kv for kv in _KEYS if (kv[1].startswith('conv_dma') and '|ks1|' in kv[1] and kv[0] == 'dgrad' and not _REPS[kv]['recipe']['accumulate']
                                  and not _REPS[kv]['recipe'].get('bstat'))
           or (kv[1].startswith('conv<') and kv[1].rsplit('|ks', 1)[-1] not in ('0', '1') and kv[0] == 'fwd')]


def test_ksplit_cases_exist():
    assert any(kv[1].startswith('conv_dma') for kv in _KSPLIT) and any(kv[1].startswith('conv<') for kv in _KSPLIT)


@pytest.mark.parametrize('kv', _KSPLIT, ids=[_id(kv) for kv in _KSPLIT])
def test_ksplit_exchange_is_bitwise_stable_under_load(kv):
    """the K-split exchange (relaxed agent-scope stores + vmcnt(0) + workgroup barrier + relaxed ticket; the
    last arriver reads with agent-scope loads and resets the counter; DESIGN 3.1 has the hardware argument) has no fence.  200
    launches of every K-split variant of both families, alternating between two streams, must reproduce the first launch bit for bit."""
    r = _REPS[kv]
    LR.stress_recipe(r['recipe'], kv[1], torch.device('cuda:0'), launches=200)
