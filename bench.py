"""bench.py -- VAN-GAN train_step throughput on MI355X (BASELINE.json metric: train-steps/s + Mvoxels/s at 128^3 bf16).

A "step" is one full VanGan.train_step (4 generator forwards, 4 discriminator forwards on [real;fake], all losses
incl. the clDice soft skeleton, the four backward sweeps, 4x Adam, weight repack) on one batch of synthetic volumes
that are already resident in HBM.  N=1: 128^3, batch 1 (the per-GPU workload of BASELINE config 4).  N>1: one
process per GPU (torch.distributed.run), batch 1 per GPU, RCCL SUM all-reduce of the four gradient buckets on a side
stream => weak scaling.

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     : the dominant kernel family (the bf16 MFMA gather-convolution: forward + data-gradient + weight-gradient
                 launches of one step), algorithmic conv FLOPs of those launches / their summed HIP-event durations,
                 against the 2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline : the oracle (fp32 torch-CPU restatement of the reference graph -- NOT TensorFlow, which is not
                 installable here) timed on this box's host cores on a bounded 32^3 sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PMC_SUMMARY = 'r06_hbm_pmc.json'   # tools/hbm_pmc.py output for the current kernels (stamped with their source hash)
CONV_FLOP_PER_VOXEL = 2530548.0    # 12 F_G + 14 F_D per voxel-sample-step


def synth_on_device(B, dims, seed, device):
    """SURVEY 8d synthetic volumes, generated once on the host (setup, untimed) and moved to HBM."""
    from van_gan_amd.synth import synth_volumes        # same generator family as the parity tests (tests/test_data_oracle.py)
    rI, rS = synth_volumes(B, *dims, seed=seed)
    return rI.to(device), rS.to(device)


def cpu_baseline(budget_s=12.0):
    import torch
    from oracle import vangan_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))   # 32^3 tensors: more threads only add synchronisation
    dims, B = (32, 32, 32), 1
    P = O.make_models(0)
    rI, rS = O.synth_volumes(B, *dims, seed=1234)
    cfg = O.Cfg(B, 1)
    state = {}
    O.train_step(P, state, rI, rS, cfg)                 # warm-up (allocator, thread pool)
    n, t0 = 0, time.time()
    while True:
        O.train_step(P, state, rI, rS, cfg)
        n += 1
        el = time.time() - t0
        if el >= budget_s or n >= 8:
            break
    sps = n / el
    return {'value': sps * B * dims[0] * dims[1] * dims[2] / 1e6, 'unit': 'Mvoxels/s', 'steps_per_s': sps,
            'cores': torch.get_num_threads(), 'host_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': 'oracle.train_step (fp32 torch-CPU restatement of the reference graph, not TensorFlow) at '
                      '32x32x32 batch 1, %d steps in %.1f s on %d torch threads (torch.get_num_threads) of %s host CPUs (os.cpu_count)'
                      % (n, el, torch.get_num_threads(), os.cpu_count())}


def time_infer(eng, device, steps, warmup, precision):
    """Config 5: one 256x256x128 volume, 128^3 windows, stride 50, symmetric pad 0.1, 10 % border crop
    (post_training.py:38-39; custom_callback.py:47-223): 50 windows through the generator, overlap-add on the GPU.
    precision 'fp16' (what BASELINE.json names: IEEE half storage, libvangan_hip_h.so) or None (the engine's bf16)."""
    import torch
    vol = (torch.rand(256, 256, 128, 1, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(device)
    kw = dict(stride=(50, 50, 50), complete=True, padFactor=0.1, process_img=True, window_batch=2, precision=precision)
    out = None
    for _ in range(max(warmup, 2)):                     # >= 2: the first volume of a precision loads its library / repacks / grows the arena
        out = eng.stitch_subvolumes('gen_IS', vol, (128, 128, 128), **kw)
    torch.cuda.synchronize()
    steps = max(steps, 5)
    per = []
    for _ in range(steps):                              # volume by volume: min and median are reported beside the mean
        t0 = time.perf_counter()
        out = eng.stitch_subvolumes('gen_IS', vol, (128, 128, 128), **kw)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
    el = sum(per) / steps
    med = sorted(per)[steps // 2]
    return {'workload': 'GanMonitor.stitch_subvolumes 256x256x128, 50 windows of 128^3, stride 50, pad 0.1 (BASELINE config 5)',
            'dtype': precision or 'bf16', 'ms_per_volume': el * 1e3, 'ms_per_volume_min': min(per) * 1e3, 'ms_per_volume_median': med * 1e3,
            'volumes_per_sec': 1.0 / el, 'Mvoxels_per_sec': 256 * 256 * 128 / el / 1e6,
            'generator_tflops': 50 * 2 * 149.65e9 / el / 1e12, 'windows': 50, 'steps': steps, 'warmup': max(warmup, 2),
            'finite': bool(torch.isfinite(out).all())}


def time_config(dims, B, device, steps=20, warmup=5, graph=True, generator='resUnet'):
    """BASELINE configs 2 and 3 beside the headline: a full train_step at another patch size / batch on a fresh engine
    (same kernels, same schedule, noise + dropout + clDice on), timed like the headline loop -- enqueued launch by launch
    (`eager_ms_per_step`) and as replays of the step's recorded launch list (`replay_ms_per_step`, VanGan.train_step_replay: the same
    kernels, streams and dependencies re-issued without the Python that builds them, the host refreshing a 32-byte parameter block per
    step).  `ms_per_step` is the faster of the two paths and `path` names it.  (A HIP graph of the step -- VanGan.train_step_graph --
    is correct and slower than either on ROCm 7.2: DESIGN 6.19.)"""
    import torch
    from van_gan_amd import VanGan
    eng = VanGan(dims, batch_size=B, device=device, seed=0, generator=generator)
    rI, rS = synth_on_device(B, dims, 1234, device)
    for _ in range(warmup):
        eng.train_step(rI, rS, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for i in range(steps):
        res = eng.train_step(rI, rS, sync=(i == steps - 1))
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    eager_ms, g_ms, g_err = el * 1e3, None, None
    finite = all(v == v and abs(v) != float('inf') for v in res.values())
    if graph:
        try:
            for _ in range(warmup):
                eng.train_step_replay(rI, rS, sync=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                resg = eng.train_step_replay(rI, rS, sync=(i == steps - 1))
            torch.cuda.synchronize()
            g_ms = (time.perf_counter() - t0) / steps * 1e3
            finite = finite and all(v == v and abs(v) != float('inf') for v in resg.values())
            if g_ms < eager_ms:
                el = g_ms * 1e-3
        except Exception as e:              # the eager figure stands; the reason is on the line
            g_err = '%s: %s' % (type(e).__name__, str(e)[:300])
    S = dims[0] * dims[1] * dims[2]
    out = {'workload': 'VanGan.train_step, %dx%dx%d volumes, batch %d, bf16, clDice on, disc noise+dropout on' % (dims + (B,))
                       + ('' if generator == 'resUnet' else ", generator='%s' (the non-default pair of vangan.py:88-97, SURVEY 8(f)4; its conv FLOPs are ~3x the ResUNet's: whole_step_conv_tflops does not apply)" % generator),
           'ms_per_step': el * 1e3, 'path': 'launch-list replay' if (g_ms is not None and g_ms < eager_ms) else 'eager enqueue',
           'eager_ms_per_step': eager_ms, 'replay_ms_per_step': g_ms, 'replay_error': g_err,
           'train_steps_per_sec': 1.0 / el, 'Mvoxels_per_sec': B * S / el / 1e6,
           'whole_step_conv_tflops': (B * S * CONV_FLOP_PER_VOXEL / el / 1e12) if generator == 'resUnet' else None, 'steps': steps, 'warmup': warmup,
           'finite': finite, 'arena_peak_gb': eng.arena.peak / 1e9}
    del eng
    torch.cuda.empty_cache()
    return out


def _timed_steps(eng, rI, rS, steps, warmup):
    import torch
    for _ in range(warmup):
        eng.train_step(rI, rS, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(rI, rS, sync=(i == steps - 1))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def ddp_fake_child(args):
    """Child process of the N=1 line (`--ddp-fake`): BASELINE config 4's per-GPU workload with the DATA-PARALLEL schedule switched on
    for one process (VG_FAKE_AR=1: communication on its stream, per-bucket events, early suffix pieces, cross-step optimizer overlap;
    every all-reduce replaced by vg_local_exchange moving the same bytes, optionally held for a ring's duration) next to the plain
    single-GPU schedule of the SAME process.  Prints {"ddp_fake": ...}."""
    import torch
    from van_gan_amd import VanGan
    device = 'cuda:0'
    torch.cuda.set_device(0)
    dims, B = (args.size,) * 3, args.batch
    rI, rS = synth_on_device(B, dims, 1234, device)
    out = {'workload': 'VanGan.train_step %dx%dx%d batch %d, one process' % (dims + (B,)), 'steps': args.steps, 'warmup': args.warmup, 'legs': []}
    os.environ['VG_FAKE_AR'] = '0'
    eng = VanGan(dims, batch_size=B, device=device, seed=0)
    plain = _timed_steps(eng, rI, rS, args.steps, args.warmup)
    del eng
    torch.cuda.empty_cache()
    os.environ['VG_FAKE_AR'] = '1'
    for gbps in (0.0, 150.0):
        os.environ['VG_FAKE_AR_GBPS'] = str(gbps)
        eng = VanGan(dims, batch_size=B, device=device, seed=0)
        eng.broadcast_weights(0)
        ms = _timed_steps(eng, rI, rS, args.steps, args.warmup)
        n = args.steps + args.warmup
        out['legs'].append({'ring_bus_GBps_emulated': gbps or None, 'ms_per_step': ms, 'delta_ms': ms - plain,
                            'moved_MB_per_step': eng.sync.moved_bytes / n / 1e6, 'workgroups': eng.sync._wg,
                            'comm_stream': 'optimizer stream' if eng.sync.stream is eng._opt else 'own stream',
                            'xstep': bool(eng._xstep)})
        del eng
        torch.cuda.empty_cache()
    out['plain_ms_per_step'] = plain
    out['ms_per_step'] = out['legs'][-1]['ms_per_step']
    out['delta_ms'] = out['legs'][-1]['delta_ms']
    out['note'] = ('world == 1: GradSync runs the real data-parallel schedule with vg_local_exchange in place of ncclAllReduce (include/vangan_hip.h); '
                   'ms_per_step / delta_ms are the leg whose exchange kernels hold their stream for the duration of an 8-GPU ring at 150 GB/s bus bandwidth; '
                   'RCCL itself has not run')
    print(json.dumps({'ddp_fake': out}))


def ddp_fake(args):
    """Run ddp_fake_child in a fresh process (its streams are created in the data-parallel order) and return its object."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--ddp-fake', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--size', str(args.size), '--batch', str(args.batch)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    except subprocess.TimeoutExpired:
        return {'error': 'timeout'}
    for line in r.stdout.decode(errors='replace').splitlines():
        if line.startswith('{"ddp_fake"'):
            return json.loads(line)['ddp_fake']
    return {'error': 'rc %d: %s' % (r.returncode, r.stderr.decode(errors='replace')[-800:])}


def bench_infer(args, device):
    import torch  # noqa: F401
    from van_gan_amd import VanGan
    eng = VanGan((128, 128, 128), batch_size=2, device=device, seed=0)
    rb = time_infer(eng, device, args.steps, args.warmup, None)
    r = time_infer(eng, device, args.steps, args.warmup, 'fp16')
    print(json.dumps({'metric': 'sliding-window inference Mvoxels/s (256x256x128 volume, 50 windows of 128^3, fp16)',
                      'value': r['Mvoxels_per_sec'], 'unit': 'Mvoxels/s', 'volumes_per_sec': r['volumes_per_sec'], 'n_gpus': 1,
                      'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_volume'], 'higher_is_better': True,
                      'dtype': 'fp16', 'data': 'synthetic', 'windows': 50, 'generator_tflops': r['generator_tflops'],
                      'config': {'workload': r['workload']}, 'finite': r['finite'], 'bf16': rb}))


def _self_launch(args):
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sys.stderr.write(r.stderr.decode(errors='replace')[-6000:])
    for line in r.stdout.decode(errors='replace').splitlines():
        if line.startswith('{"metric"'):
            print(line)
    if r.returncode != 0:
        raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--dims', type=int, nargs=3, default=None, help='D H W of a non-cubic workload (BASELINE config 3: --dims 128 128 64 --batch 2); overrides --size')
    ap.add_argument('--batch', type=int, default=1, help='per-GPU batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dump-kernels', default='', help='write the per-kernel-template table of the timing step (JSON) here')
    ap.add_argument('--infer', action='store_true', help='BASELINE config 5 instead: 256x256x128 sliding-window generator inference')
    ap.add_argument('--no-configs', action='store_true', help='skip the configs array (BASELINE configs 2 and 3) of the default N=1 line')
    ap.add_argument('--no-synced', action='store_true', help='skip the second timed loop that reads the 10 result scalars every step')
    ap.add_argument('--no-infer', action='store_true', help='skip the inference object (config 5) of the default N=1 line')
    ap.add_argument('--no-replay', action='store_true', help='skip the launch-list replay figure (replay_ms_per_step) of the N=1 line')
    ap.add_argument('--graph', action='store_true', help='also time the headline workload as HIP-graph replays (reported as graph_ms_per_step; the headline value stays the eager loop)')
    ap.add_argument('--no-ddp-path', action='store_true', help='skip the ddp_path object (the data-parallel schedule with a stand-in all-reduce, in a child process) of the default N=1 line')
    ap.add_argument('--ddp-fake', action='store_true', help='(child mode of ddp_path) time the data-parallel schedule on one GPU with vg_local_exchange in place of the all-reduce')
    args = ap.parse_args()
    if args.ddp_fake:
        return ddp_fake_child(args)

    import torch
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' in os.environ and world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d in the environment: a %d-rank figure must not be recorded as the '
                         '%d-GPU scaling point (launch with --nproc-per-node %d, or unset WORLD_SIZE)' % (args.gpus, world, world, args.gpus, args.gpus))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (the reference's main.py:22 MirroredStrategy is single-command too): start the
        # one-process-per-GPU job as a CHILD, before anything here has touched the GPU (never exec from a GPU process), and relay
        # its JSON line and return code
        return _self_launch(args)
    # VG_BENCH_ONE_DEVICE=1 (development aid): every rank on cuda:0 over gloo, so that the N > 1 code path of this file can be
    # exercised on a 1-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    one_dev = os.environ.get('VG_BENCH_ONE_DEVICE', '0') == '1'
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    device = 'cuda:%d' % local
    pg = None
    if world > 1:
        if one_dev:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(device))
        pg = dist.group.WORLD

    from van_gan_amd import VanGan, ops
    if args.infer:
        return bench_infer(args, device)
    dims = tuple(args.dims) if args.dims else (args.size,) * 3
    if args.dims:
        args.size = 0                      # not the 128^3 headline workload: no configs / inference objects, no PMC traffic of that workload
    B = args.batch
    eng = VanGan(dims, batch_size=B, n_devices=world, device=device, seed=0, process_group=pg)
    eng.broadcast_weights(0)
    rI, rS = synth_on_device(B, dims, 1234 + rank, device)

    for _ in range(args.warmup):
        eng.train_step(rI, rS, sync=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for i in range(args.steps):
        res = eng.train_step(rI, rS, sync=(i == args.steps - 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = el / args.steps * 1e3
    S = dims[0] * dims[1] * dims[2]
    gbatch = B * world
    steps_per_s = args.steps / el
    mvox = steps_per_s * gbatch * S / 1e6

    # the reference loop (vangan.py:540-543) and train.train() consume the 10 result scalars EVERY step: the same loop with the host
    # reading them each step (one device->host copy + sync per step), reported beside the headline
    ms_synced = None
    if not args.no_synced:
        ns = max(3, min(10, args.steps))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(ns):
            eng.train_step(rI, rS, sync=True)
        torch.cuda.synchronize()
        ms_synced = (time.perf_counter() - t1) / ns * 1e3
    graph_ms = replay_ms = None
    if rank == 0 and world == 1 and not args.no_replay:
        # the same step re-issued from its recorded launch list (VanGan.train_step_replay): what the host-bound configurations gain
        for _ in range(args.warmup):
            eng.train_step_replay(rI, rS, sync=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eng.train_step_replay(rI, rS, sync=(i == args.steps - 1))
        torch.cuda.synchronize()
        replay_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if args.graph and world == 1:
        eng.capture_train_step()
        for _ in range(args.warmup):
            eng.train_step_graph(rI, rS, sync=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eng.train_step_graph(rI, rS, sync=(i == args.steps - 1))
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t1) / args.steps * 1e3
    configs = None
    if rank == 0 and world == 1 and args.size == 128 and not args.no_configs:
        configs = [time_config((64, 64, 64), 2, device), time_config((128, 128, 64), 2, device),
                   time_config((128, 128, 128), 1, device, steps=10, warmup=3, graph=False, generator='resnet')]
    infer = None
    if rank == 0 and world == 1 and args.size == 128 and not args.no_infer:
        # BASELINE config 5 beside the headline: the same engine's gen_IS, fp16 storage (and the bf16 figure next to it)
        # bf16 first (the engine's own library is warm), then the fp16 build; 2 warm-up volumes + 5 timed ones each
        inf_b = time_infer(eng, device, 5, 2, None)
        infer = time_infer(eng, device, 5, 2, 'fp16')
        infer['bf16_ms_per_volume'] = inf_b['ms_per_volume']
        infer['bf16_ms_per_volume_median'] = inf_b['ms_per_volume_median']
    roof = None
    summ = None
    byvar = None
    if not args.no_roofline:
        # EVERY rank runs the per-launch timing step: it contains the gradient all-reduces, which must be matched on all ranks
        # two passes, the second is reported: the first serial-schedule step after the two-lane timing loop runs its kernels with
        # cold instruction caches and a clock that has not settled (its family figure scattered by 10 % from run to run)
        for _ in range(2):
            ops.PROF = ops.KernelProfile()
            eng.train_step(rI, rS, sync=True)
        summ = ops.PROF.summary()
        byvar = ops.PROF.by_variant()
        bylayer = ops.PROF.by_layer()
        ops.PROF = None
    if rank == 0 and summ is not None:
        tot_fl = sum(v['flops'] for v in summ.values())
        tot_ms = sum(v['ms'] for v in summ.values())
        n = sum(v['launches'] for v in summ.values())
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM traffic of the same kernel family: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) cannot run inside this
        # process, so the per-launch figure comes from the committed summary of those passes on this command
        # it is only quoted while it describes THIS build: the summary is stamped with the hash of the kernel sources it was
        # measured on (van_gan_amd/build.py::_src_hash) and dropped (null) when the sources have changed since
        traffic, traffic_src = None, None
        try:
            from van_gan_amd.build import _src_hash
            with open(os.path.join(ROOT, 'profiles', PMC_SUMMARY)) as f:
                pm = json.load(f)
            if args.size == 128 and B == 1 and pm.get('csrc_hash') == _src_hash():
                traffic = pm['hbm_bytes_per_launch']
                traffic_src = 'profiles/%s (csrc %s)' % (PMC_SUMMARY, pm['csrc_hash'][:12])
            elif args.size == 128 and B == 1:
                traffic_src = 'stale: profiles/%s was measured on csrc %s, this build is %s' % (
                    PMC_SUMMARY, str(pm.get('csrc_hash'))[:12], _src_hash()[:12])
        except (OSError, KeyError, ValueError):
            pass
        alg_bytes = sum(v.get('bytes', 0.0) for v in summ.values())
        roof = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (read + write), PMC', 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': alg_bytes / n,
                'kernel': 'conv_kernel / conv32_kernel / conv_thin_kernel / conv_dma_kernel (+ its materialize pass) / wgrad_dma_kernel (+ its materialize and reduce_partials passes) / wgrad_thin_kernel (+ its slab pass) / wgrad_kernel<bf16,*> + the pointwise pw_* kernels of the 1x1x1 layers and the c1m_* kernels of the single-channel layers (every vg_conv3d / vg_conv3d_wgrad call of one step)',
                'launches_per_step': n, 'avg_launch_ms': tot_ms / n, 'kernel_ms_per_step': tot_ms,
                'algorithmic_gflop_per_step': tot_fl / 1e9,
                'by_kind': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else None} for k, v in summ.items()},
                'whole_step_conv_tflops': steps_per_s * gbatch / world * S * CONV_FLOP_PER_VOXEL / 1e12}
        if byvar:
            # the single kernel template with the most time in the step, by itself (the family figure above averages ~70
            # templates): algorithmic FLOPs of ITS launches / ITS summed HIP-event durations
            dom = byvar[0]                 # rows are keyed by the rocprof kernel template (run-time regimes of one template summed)
            roof['dominant_kernel'] = {'kernel': dom['kernel'], 'kind': dom['kind'], 'launches_per_step': dom['launches'],
                                       'ms_per_step': dom['ms'], 'avg_launch_ms': dom['ms'] / dom['launches'],
                                       'achieved': dom['tflops'], 'frac': dom['tflops'] / PEAK_BF16_TFLOPS,
                                       'algorithmic_bytes_per_launch': dom['algorithmic_bytes'] / dom['launches']}
            roof['by_kernel'] = [{'kernel': r['kernel'], 'kind': r['kind'], 'launches': r['launches'], 'ms': round(r['ms'], 4),
                                  'tflops': None if r['tflops'] is None else round(r['tflops'], 1)} for r in byvar[:12]]
            if args.dump_kernels:
                with open(args.dump_kernels, 'w') as f:
                    json.dump(byvar, f, indent=1)
            if os.environ.get('VG_DUMP_LAYERS'):           # development aid: where the family's time goes, layer by layer
                with open(os.environ['VG_DUMP_LAYERS'], 'w') as f:
                    for r in bylayer:
                        f.write('%-11s %-34s %-44s n %3d  %8.3f ms  %7.1f TF/s\n' % (r['kind'], r['layer'], r['kernel'], r['launches'], r['ms'], r['tflops'] or 0))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    ddp = None
    if rank == 0 and world == 1 and args.size == 128 and not args.no_ddp_path:
        ddp = ddp_fake(args)

    dev_names = None
    if world > 1:
        mine = '%d:cuda:%d:%s' % (rank, local, torch.cuda.get_device_name(local))
        dev_names = [None] * world
        dist.all_gather_object(dev_names, mine)
        dist.barrier()
    if rank == 0:
        out = {
            'metric': 'train Mvoxels/s (VanGan.train_step, 128^3 bf16)' if args.size == 128 else 'train Mvoxels/s (VanGan.train_step, %dx%dx%d bf16)' % dims,
            'value': mvox, 'unit': 'Mvoxels/s', 'train_steps_per_sec': steps_per_s,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'VanGan.train_step, %dx%dx%d volumes, batch %d per GPU (global %d), clDice on, disc noise+dropout on'
                                   % (dims + (B, gbatch)), 'parallelism': 'dp%d' % world},
            'ms_per_step_synced': ms_synced, 'replay_ms_per_step': replay_ms, 'graph_ms_per_step': graph_ms, 'configs': configs,
            'losses': res, 'roofline': roof, 'cpu_baseline': cpu, 'inference': infer,
            'arena_peak_gb': eng.arena.peak / 1e9, 'ddp_path': ddp,
        }
        if world > 1:
            # what the process group actually was (the driver can check it against --gpus): backend, ranks, devices
            out['distributed'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'device_count': torch.cuda.device_count(),
                                  'devices': dev_names, 'one_device_debug': one_dev}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
