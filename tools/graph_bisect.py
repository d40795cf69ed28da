"""Development aid: which part of a train step can be captured into a HIP graph on this ROCm?  Every probe runs in its own process
(a failing capture may take the process down): python tools/graph_bisect.py            -> runs all probes
                                              python tools/graph_bisect.py <probe>    -> one probe in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBES = {
    'fill_only': {},
    'one_conv': {},
    'fork_join': {},
    'test_step_serial': dict(VG_LANES='0', VG_SIDE_STREAM='0', VG_OPT_STREAM='0'),
    'train_noapply_serial': dict(VG_LANES='0', VG_SIDE_STREAM='0', VG_OPT_STREAM='0'),
    'train_serial': dict(VG_LANES='0', VG_SIDE_STREAM='0', VG_OPT_STREAM='0'),
    'train_side': dict(VG_LANES='0', VG_SIDE_STREAM='1', VG_OPT_STREAM='0'),
    'train_opt': dict(VG_LANES='0', VG_SIDE_STREAM='0', VG_OPT_STREAM='1'),
    'train_lanes': dict(VG_LANES='1', VG_SIDE_STREAM='0', VG_OPT_STREAM='0'),
    'train_lanes_side': dict(VG_LANES='1', VG_SIDE_STREAM='1', VG_OPT_STREAM='0'),
    'train_lanes_opt': dict(VG_LANES='1', VG_SIDE_STREAM='0', VG_OPT_STREAM='1'),
    'train_side_opt': dict(VG_LANES='0', VG_SIDE_STREAM='1', VG_OPT_STREAM='1'),
    'train_full_nolazy': dict(VG_LAZY_AR='0'),
    'train_full_noforkshort': dict(VG_FORK_SHORT='0'),
    'train_full_nojoin0': dict(VG_NOJOIN='0'),
    'train_full': {},
}


def probe(name):
    import torch
    from van_gan_amd import VanGan, ops
    from van_gan_amd.synth import synth_volumes
    dev = torch.device('cuda:0')
    if name in ('fill_only', 'one_conv', 'fork_join'):
        s = torch.cuda.Stream()
        x = torch.zeros(1 << 20, device=dev)
        y = torch.zeros(1 << 20, device=dev)
        g = torch.cuda.CUDAGraph()
        ops.set_device(0)
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.axpby(x, 2.0, None, 0.0, y)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            x.fill_(1.0)
            if name != 'fill_only':
                ops.axpby(x, 2.0, None, 0.0, y)
            if name == 'fork_join':
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    ops.axpby(x, 3.0, None, 0.0, y, accumulate=True)
                torch.cuda.current_stream().wait_stream(s2)
        g.replay(); torch.cuda.synchronize()
        print(name, 'OK', float(y[0]))
        return
    dims, B = (32, 32, 32), 1
    eng = VanGan(dims, batch_size=B, device='cuda:0', seed=0)
    rI, rS = synth_volumes(B, *dims, seed=1)
    rI, rS = rI.cuda(), rS.cuda()
    s = torch.cuda.Stream()
    do_bwd = not name.startswith('test_step')
    apply = do_bwd and 'noapply' not in name
    with torch.cuda.stream(s):
        for _ in range(2):
            eng._losses_and_backward(rI, rS, True, None, None, do_bwd, apply=apply)
    torch.cuda.synchronize()
    from van_gan_amd.vangan import _StepParams
    cap = _StepParams(eng.device); cap.base = eng.rng_offset
    eng._cap = cap
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng._losses_and_backward(rI, rS, True, None, None, do_bwd, apply=apply)
    eng._cap = None
    cap.refresh(0, 0.1, [1e-4] * 4)
    g.replay(); torch.cuda.synchronize()
    print(name, 'OK', eng._acc.cpu().tolist()[:4])


if __name__ == '__main__':
    if len(sys.argv) > 1:
        probe(sys.argv[1])
    else:
        for name, env in PROBES.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, VG_NO_REBUILD='1', **env),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            out = r.stdout.decode().strip().splitlines()
            err = [l for l in r.stderr.decode().splitlines() if 'Error' in l or 'error' in l or 'Segmentation' in l or 'HIP' in l]
            print('%-22s rc %4d  %s  %s' % (name, r.returncode, out[-1] if out else '', ' | '.join(err[-3:])[:300]), flush=True)
