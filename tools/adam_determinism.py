"""Is clip + Adam bit-reproducible on a fixed gradient?  (Replicas must apply bit-identical updates to the all-reduced buckets;
development aid, run on the GPU box.)   python tools/adam_determinism.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan
from van_gan_amd.synth import synth_volumes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = VanGan((32,) * 3, batch_size=1, n_devices=2, device='cuda:0', seed=0, layer_noise=0.0, dropout_rate=0.0, precision='fp32')
rI, rS = synth_volumes(1, 32, 32, 32, seed=5)
eng.train_step(rI.cuda(), rS.cuda(), apply=False)
torch.cuda.synchronize()
state = {k: (s.w.clone(), s.m.clone(), s.v.clone(), s.g.clone(), s.step) for k, s in eng.stores.items()}
first = None
for r in range(reps):
    for k, s in eng.stores.items():
        w, m, v, g, st = state[k]
        s.w.copy_(w); s.m.copy_(m); s.v.copy_(v); s.g.copy_(g); s.step = st
    eng._apply_adam()
    torch.cuda.synchronize()
    cur = {k: s.w.clone() for k, s in eng.stores.items()}
    nrm = {k: s.norms.clone() for k, s in eng.stores.items()}
    if first is None:
        first, nfirst = cur, nrm
        for k, s in eng.stores.items():
            n = s.norms[:s.T].sqrt()
            print('%-8s tensors %3d, clipped (norm > 100): %d, max norm %.3e' % (k, s.T, int((n > 100).sum()), float(n.max())))
    else:
        for k in cur:
            dw = int((cur[k] != first[k]).sum()); dn = int((nrm[k] != nfirst[k]).sum())
            if dw or dn:
                print('rep %d %-8s: %d weights differ, %d norms differ' % (r, k, dw, dn))
print('done')
