"""world_size-2 gloo test (CPU) of the data-parallel host logic: flat gradient buckets, SUM all-reduce issued per
network, SUM of the result dict, weight broadcast, identical Adam update on every rank.

Semantics under test (reference: MirroredStrategy, vangan.py:426-438,472-490; loss_functions.py:21-22,226): every replica
computes its losses on its LOCAL batch with the GLOBAL batch size in the denominators and lambda_topology/n_devices on the
clDice term; gradients and result scalars are SUMMED over replicas.  The compute here is the CPU oracle (tests may use it);
the synchronisation code is the product's van_gan_amd.dist.GradSync on the product's ParamStore layout."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vangan_oracle as O

DIMS = (32, 32, 32)
NETS = ['gen_IS', 'gen_SI', 'disc_I', 'disc_S']


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_grads(rank, world):
    P = O.make_models(0)
    rI, rS = O.synth_volumes(1, *DIMS, seed=1234 + rank)
    res, grads, _ = O.train_step(P, {}, rI, rS, O.Cfg(world, world), apply=False)
    return P, res, grads


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(4)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from van_gan_amd.dist import GradSync
    from van_gan_amd.nets import ParamStore, disc_param_specs, gen_param_specs
    stores = {k: ParamStore(gen_param_specs() if k.startswith('gen') else disc_param_specs(), 'cpu') for k in NETS}
    P, res, grads = _rank_grads(rank, world)
    for k in NETS:
        stores[k].load(P[k] if rank == 0 else {n: torch.zeros_like(t) for n, t in P[k].items()})
        for n in grads[k]:
            stores[k].grad(n).copy_(grads[k][n])
    sync = GradSync({k: s.g for k, s in stores.items()}, dist.group.WORLD, {k: s.w for k, s in stores.items()})
    sync.broadcast_weights(0)
    # gen_IS in two pieces, as the engine does it: the suffix a backward sweep finishes first (enc4 ... output head), then the rest
    off = stores['gen_IS'].offsets['enc4.cb1.in.gamma'][0]
    assert 0 < off < stores['gen_IS'].total and off % 4 == 0
    sync.start(['disc_I', 'disc_S']); sync.start(['gen_IS'], lo=off); sync.start(['gen_IS'], hi=off); sync.start(['gen_SI'])
    assert len(sync.pending['gen_IS']) == 2
    # per-bucket completion: a network's optimizer step waits for ITS bucket only
    sync.finish(['disc_S'])
    assert 'disc_S' not in sync.pending and {'disc_I', 'gen_IS', 'gen_SI'} <= set(sync.pending)
    sync.finish(['gen_SI', 'never_started'])
    sync.finish()
    assert not sync.pending
    red = sync.reduce_dict(res, O.RESULT_KEYS)
    # identical Adam update from the reduced buckets on every rank
    state = {}
    for k in NETS:
        Pk = stores[k].export()
        O.adam_step(Pk, stores[k].export(stores[k].g), state.setdefault(k, {}))
        stores[k].load(Pk)
    torch.save({'g': {k: stores[k].g.clone() for k in NETS}, 'w': {k: stores[k].w.clone() for k in NETS}, 'res': red,
                'local': res}, out % rank)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_gradient_sum_and_weight_sync(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / 'rank%d.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = [torch.load(out % i) for i in range(world)]
    torch.set_num_threads(4)          # same reduction order as the workers (the fp32 oracle is order-sensitive)
    # serial restatement of the same semantics
    from van_gan_amd.nets import ParamStore, disc_param_specs, gen_param_specs
    tot = {k: None for k in NETS}
    res_sum = {k: 0.0 for k in O.RESULT_KEYS}
    for rank in range(world):
        _, res, grads = _rank_grads(rank, world)
        for k in NETS:
            st = ParamStore(gen_param_specs() if k.startswith('gen') else disc_param_specs(), 'cpu')
            for n in grads[k]:
                st.grad(n).copy_(grads[k][n])
            tot[k] = st.g.clone() if tot[k] is None else tot[k] + st.g
        for k2 in res_sum:
            res_sum[k2] += res[k2]
    for k in NETS:
        for i in range(world):
            rel = float((r[i]['g'][k] - tot[k]).norm() / tot[k].norm())
            assert rel < 1e-4, (k, rel)
        assert torch.equal(r[0]['w'][k], r[1]['w'][k]), 'replicas diverged: ' + k
    for k2 in O.RESULT_KEYS:
        assert r[0]['res'][k2] == pytest.approx(res_sum[k2], rel=1e-5)
        assert r[0]['res'][k2] == pytest.approx(r[1]['res'][k2], rel=1e-6)
    # the clDice term is scaled by 1/n_devices per replica (loss_functions.py:226): summed, it is O(single-device value)
    assert 0.5 * r[0]['local']['seg_loss'] < r[0]['res']['seg_loss'] / world < 2.0 * r[0]['local']['seg_loss']
