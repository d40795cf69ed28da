"""Lane timeline of one 128^3 train step from HIP events (VG_TIMELINE=1): when each lane reaches its milestones."""
import os, sys
os.environ['VG_TIMELINE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan
from van_gan_amd.synth import synth_volumes
eng = VanGan((128,) * 3, batch_size=1, device='cuda:0')
rI, rS = synth_volumes(1, 128, 128, 128, seed=1); rI, rS = rI.cuda(), rS.cuda()
for _ in range(4):
    eng.train_step(rI, rS, sync=False)
eng.timeline()
for rep in range(2):
    eng.train_step(rI, rS, sync=False)
    for n, t in sorted(eng.timeline(), key=lambda x: x[1]):
        print('%8.2f ms  %s' % (t, n))
    print()
