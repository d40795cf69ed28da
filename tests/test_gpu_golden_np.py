"""The HIP engine (exact-parity mode) against the TWO-AUTHOR fixtures of tests/golden/make_golden_np.py: one train step at
32^3 with batch 1 and with batch 2 / global batch 2 -- 10 losses, the generated volumes, every block output of G_IS(real_I) (per-channel
moments + 128 sampled voxels), and the gradient of EVERY one of the 262 parameter tensors through its norm and its projection on a seeded
random direction (the InstanceNorm gamma / beta gradients and the first / last convolution of each network in full).  The fixture values
are float64 numpy (explicit loops, hand-derived backward) that agreed with torch autograd to 1e-10 when they were written."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from oracle import vangan_oracle as O  # noqa: E402
from make_golden_np import BLOCKS, direction  # noqa: E402


@pytest.mark.parametrize('B', [1, 2])
def test_train_step_matches_the_two_author_fixture(B):
    from van_gan_amd import VanGan
    f = np.load(os.path.join(HERE, 'golden', 'train_step_32_b%d_np.npz' % B))
    dims = tuple(int(v) for v in f['dims'])
    eng = VanGan(dims, batch_size=B, global_batch_size=int(f['global_batch']), n_devices=1, device='cuda:0', seed=0, layer_noise=0.0,
                 dropout_rate=0.0, precision='fp32')
    eng.load_weights(O.make_models(int(f['seed'])))
    rI, rS = torch.from_numpy(f['real_I']), torch.from_numpy(f['real_S'])
    res = eng.train_step(rI.cuda(), rS.cuda(), noise={}, drop={}, apply=False)
    for i, k in enumerate(O.RESULT_KEYS):
        ref = float(f['losses'][i])
        print('   %-24s hip %.6f  fixture %.6f' % (k, res[k], ref))
        assert abs(res[k] - ref) <= 1e-4 * abs(ref) + 1e-6, k
    for name in ('fake_S', 'fake_I'):
        got = eng._aux[name].float().cpu().numpy()
        assert np.linalg.norm(got - f[name]) / np.linalg.norm(f[name]) < 1e-4, name
    # block by block: what the engine STORED in the forward pass of G_IS(real_I)
    ctx = eng._fwd_ctx['G_IS.a']
    for b in BLOCKS:
        t = (ctx['bridge']['b2'] if b == 'bridge' else ctx[b]['out']).data.double().cpu().numpy()
        mean, msq = t.mean(axis=(1, 2, 3)), (t ** 2).mean(axis=(1, 2, 3))
        rms = np.sqrt(f['blk:%s:msq' % b])
        assert np.abs(mean - f['blk:%s:mean' % b]).max() <= 1e-4 * rms.max(), b
        assert np.abs(msq - f['blk:%s:msq' % b]).max() <= 2e-4 * f['blk:%s:msq' % b].max(), b
        val = t.reshape(t.shape[0], -1, t.shape[-1])[:, f['blk:%s:idx' % b], :]
        assert np.abs(val - f['blk:%s:val' % b]).max() <= 2e-4 * rms.max() + 1e-6, b
    # every gradient tensor: norm, projection on the fixture's seeded direction, the small tensors in full
    grads = eng.export_grads()
    worst = ('', 0.0)
    for net in ('gen_IS', 'gen_SI', 'disc_I', 'disc_S'):
        names, norms, projs = [str(n) for n in f['gnames:' + net]], f['gnorm:' + net], f['gproj:' + net]
        gmax = float(norms.max())
        for n, nr, pr in zip(names, norms, projs):
            g = grads[net][n].double().numpy()
            if nr < 1e-6 * gmax:                 # analytically zero (a bias in front of an InstanceNorm): absolute
                assert np.linalg.norm(g) <= 2e-3 * gmax, (net, n)
                continue
            e_n = abs(np.linalg.norm(g) - nr) / nr
            e_p = abs(float((g * direction(net, n, g.shape)).sum()) - pr) / nr          # |<dg, unit direction>| <= |dg|
            if max(e_n, e_p) > worst[1]:
                worst = (net + '/' + n, max(e_n, e_p))
            # 5e-2 per tensor: the bound of the other exact-parity gradient checks (tests/test_gpu_fp32.py) -- at 32^3 the deep levels have 8-512
            # voxels, and one ReLU mask that flips between fp32 and float64 moves a 64-element beta gradient by a percent or two
            assert e_n <= 5e-2 and e_p <= 5e-2, (net, n, e_n, e_p)
            key = 'grad:%s/%s' % (net, n)
            if key in f.files:
                r = f[key].astype(np.float64)
                assert np.linalg.norm(g - r) <= 5e-2 * nr, key
    print('   worst gradient deviation (norm or projection, relative to the tensor norm): %.2e at %s' % (worst[1], worst[0]))
