"""HIP-graph capture of a whole train step (VanGan.capture_train_step / train_step_graph): a replayed step must be the eager step --
same losses, same gradients, same weights after Adam, with the per-step scalars (Philox counter, noise standard deviation, lr_t)
coming from the device parameter block."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(precision, dims=(32, 32, 32), B=1, **kw):
    from van_gan_amd import VanGan
    from van_gan_amd.synth import synth_volumes
    a = VanGan(dims, batch_size=B, device='cuda:0', seed=3, precision=precision, **kw)
    b = VanGan(dims, batch_size=B, device='cuda:0', seed=3, precision=precision, **kw)
    b.load_weights(a.export_weights())
    rI, rS = synth_volumes(B, *dims, seed=7)
    return a, b, rI.cuda(), rS.cuda()


def _cos(x, y):
    x, y = x.double().flatten(), y.double().flatten()
    return float((x @ y) / (x.norm() * y.norm() + 1e-300))


def _grads_agree(eager, other, tag):
    from van_gan_amd.vangan import NETS
    ge, gg = eager.export_grads(), other.export_grads()
    for n in NETS:
        fe = torch.cat([t.flatten() for t in ge[n].values()]); fg = torch.cat([t.flatten() for t in gg[n].values()])
        assert _cos(fe, fg) > 0.9999, (tag, n, _cos(fe, fg))            # two eager engines measure 0.99997 (float atomics)
        assert float((fe - fg).norm() / fe.norm()) < 1e-2, (tag, n)          # two eager runs differ by ~2e-3 (order of the float atomics)


def test_replayed_step_is_the_eager_step_fp32():
    """Exact-parity storage (differences: only the order of float atomics), noise and dropout ON with the same Philox keys.  With the
    learning rate at 0 the weights stay put, so the SECOND step -- a pure graph replay -- must reproduce the eager engine's losses and
    every gradient; then two training steps each way (Adam's sign-like first updates amplify the atomics' last bit: losses only)."""
    from van_gan_amd.vangan import NETS, RESULT_KEYS
    eager, graph, rI, rS = _pair('fp32')
    graph.capture_train_step()
    assert graph.rng_offset == eager.rng_offset and all(graph.stores[n].step == 0 for n in NETS)      # capturing ran nothing
    for e in (eager, graph):
        e.lr = 0.0
    for step in range(2):
        re = eager.train_step(rI, rS)
        rg = graph.train_step_graph(rI, rS)
        for k in RESULT_KEYS:
            assert abs(re[k] - rg[k]) <= 2e-5 * abs(re[k]) + 1e-7, (step, k, re[k], rg[k])
        assert graph.rng_offset == eager.rng_offset
        _grads_agree(eager, graph, 'graph step %d' % step)
    for e in (eager, graph):
        e.lr = 2e-4
    for step in range(2):
        re = eager.train_step(rI, rS)
        rg = graph.train_step_graph(rI, rS)
        for k in RESULT_KEYS:
            assert abs(re[k] - rg[k]) <= 1e-2 * abs(re[k]) + 1e-6, (step, k, re[k], rg[k])
    for n in NETS:
        assert graph.stores[n].step == eager.stores[n].step == 4


def test_replay_follows_the_host_schedules_bf16():
    """Product precision: the replay honours a changed learning rate and noise level (read from the parameter block, not baked in),
    and stays finite and close to the eager engine over a few steps."""
    from van_gan_amd.vangan import RESULT_KEYS
    eager, graph, rI, rS = _pair('bf16')
    graph.capture_train_step()
    for step in range(3):
        if step == 2:
            for e in (eager, graph):
                e.lr, e.layer_noise = 0.0, 0.0                     # GanMonitor's end state: no update, no noise
            w_before = {n: graph.stores[n].w.clone() for n in graph.stores}
        re = eager.train_step(rI, rS)
        rg = graph.train_step_graph(rI, rS)
        for k in RESULT_KEYS:
            assert rg[k] == rg[k] and abs(re[k] - rg[k]) <= 5e-2 * abs(re[k]) + 1e-3, (step, k, re[k], rg[k])
    torch.cuda.synchronize()
    for n in graph.stores:
        assert torch.equal(graph.stores[n].w, w_before[n]), 'lr = 0 must leave the weights alone: lr_t is read per replay'


def test_launch_list_replay_is_the_eager_step_fp32():
    """VanGan.train_step_replay: the first call records the step's launches and stream dependencies while running it, later calls
    re-issue the list with refreshed inputs and parameter block.  Learning rate 0 for three steps (weights stay put): the recorded
    step and two pure replays -- with OTHER input volumes than the recorded ones -- reproduce the eager engine's losses and gradients."""
    from van_gan_amd.vangan import NETS, RESULT_KEYS
    eager, rep, rI, rS = _pair('fp32')
    for e in (eager, rep):
        e.lr = 0.0
    for step in range(3):
        x, y = (rI, rS) if step % 2 == 0 else (rI.flip(1).contiguous(), rS.flip(2).contiguous())     # the inputs are not baked in
        re = eager.train_step(x, y)
        rr = rep.train_step_replay(x, y)
        for k in RESULT_KEYS:
            assert abs(re[k] - rr[k]) <= 2e-5 * abs(re[k]) + 1e-7, (step, k, re[k], rr[k])
        assert rep.rng_offset == eager.rng_offset
        _grads_agree(eager, rep, 'replay step %d' % step)
    assert len(rep._rlist) > 500
    for e in (eager, rep):
        e.lr = 2e-4
    for step in range(2):
        re = eager.train_step(rI, rS)
        rr = rep.train_step_replay(rI, rS)
        for k in RESULT_KEYS:
            assert abs(re[k] - rr[k]) <= 1e-2 * abs(re[k]) + 1e-6, (step, k, re[k], rr[k])
    for n in NETS:
        assert rep.stores[n].step == eager.stores[n].step == 5


def test_launch_list_replay_follows_the_host_schedules_bf16():
    from van_gan_amd.vangan import RESULT_KEYS
    eager, rep, rI, rS = _pair('bf16', dims=(64, 32, 32), B=2)
    for step in range(4):
        if step == 3:
            for e in (eager, rep):
                e.lr, e.layer_noise = 0.0, 0.0
            w_before = {n: rep.stores[n].w.clone() for n in rep.stores}
        re = eager.train_step(rI, rS)
        rr = rep.train_step_replay(rI, rS)
        for k in RESULT_KEYS:
            assert rr[k] == rr[k] and abs(re[k] - rr[k]) <= 5e-2 * abs(re[k]) + 1e-3, (step, k, re[k], rr[k])
    torch.cuda.synchronize()
    for n in rep.stores:
        assert torch.equal(rep.stores[n].w, w_before[n])


def test_unsynchronised_replays_keep_their_own_step_scalars():
    """with sync=False the host runs ahead of the device, so the per-step scalars must be bound when a replay is ENQUEUED
    (vg_set_step_params carries them as kernel arguments).  Learning-rate sequence 0, 0, 2e-4, 0, 0 without a single synchronisation:
    the one non-zero rate must reach exactly its own step -- a parameter block refreshed through a pinned mirror + asynchronous copy
    hands step 2 the rate of step 3 (0: the weights would not move at all)."""
    from van_gan_amd.vangan import NETS
    eager, rep, rI, rS = _pair('fp32')
    w0 = {n: rep.stores[n].w.clone() for n in NETS}
    seq = (0.0, 0.0, 2e-4, 0.0, 0.0)
    for lr in seq:
        eager.lr = lr
        eager.train_step(rI, rS, sync=False)
    for lr in seq:
        rep.lr = lr
        rep.train_step_replay(rI, rS, sync=False)
    torch.cuda.synchronize()
    assert rep.rng_offset == eager.rng_offset
    for n in NETS:
        moved = float((rep.stores[n].w - w0[n]).abs().max())
        assert moved > 1e-5, (n, 'the step with lr = 2e-4 did not update the weights', moved)
        # one Adam step from zero moments moves every weight by ~ +-lr (the sign of its gradient): the two engines' updates agree except
        # where a near-zero gradient changes sign with the order of the float atomics
        ue, ur = (eager.stores[n].w - w0[n]).double(), (rep.stores[n].w - w0[n]).double()
        d = float((ur - ue).norm() / ue.norm())
        assert d < 0.2, (n, d)
    # the same through the HIP graph
    eager2, graph, rI, rS = _pair('fp32')
    graph.capture_train_step()
    w0 = {n: graph.stores[n].w.clone() for n in NETS}
    for lr in seq:
        graph.lr = lr
        graph.train_step_graph(rI, rS, sync=False)
    torch.cuda.synchronize()
    for n in NETS:
        assert float((graph.stores[n].w - w0[n]).abs().max()) > 1e-5, n


def test_replay_recorded_after_forward_only_calls():
    """test_step / generate before the FIRST train_step_replay leave a small high-water mark in the arenas' zero pools; the
    recorded reset must still clear everything a train step takes from the pool (tickets of the InstanceNorm finalisation tails, loss
    accumulators), or every replay after the recorded step runs on stale tickets.  Three steps with lr = 0 against an eager engine."""
    from van_gan_amd.vangan import RESULT_KEYS
    eager, rep, rI, rS = _pair('fp32')
    for e in (eager, rep):
        e.lr = 0.0
        e.test_step(rI, rS)
        e.generate('gen_IS', rI)
    for step in range(3):
        re = eager.train_step(rI, rS)
        rr = rep.train_step_replay(rI, rS)
        for k in RESULT_KEYS:
            assert abs(re[k] - rr[k]) <= 2e-5 * abs(re[k]) + 1e-7, (step, k, re[k], rr[k])
        _grads_agree(eager, rep, 'replay after test_step, step %d' % step)
