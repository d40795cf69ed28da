"""How long does the host need to ENQUEUE one train step vs how long the GPU needs to run it? (development aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan
from van_gan_amd.synth import synth_volumes
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = VanGan((size,) * 3, batch_size=1, device='cuda:0')
rI, rS = synth_volumes(1, size, size, size, seed=1)
rI, rS = rI.cuda(), rS.cuda()
for _ in range(2):
    eng.train_step(rI, rS, sync=False)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    eng.train_step(rI, rS, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.1f ms, total %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
