import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from van_gan_amd import VanGan
from van_gan_amd.synth import synth_volumes
eng = VanGan((128,)*3, batch_size=1, device='cuda:0')
rI, rS = synth_volumes(1, 128, 128, 128, seed=1); rI, rS = rI.cuda(), rS.cuda()
for _ in range(3): eng.train_step(rI, rS)
torch.cuda.synchronize(); t=time.time()
for _ in range(10): eng.train_step(rI, rS)
torch.cuda.synchronize(); print('ms/step %.2f' % ((time.time()-t)*100), 'peak A %.2f GB of %.2f, B %.2f GB of %.2f' % (eng.arena.peak/2**30, eng.arena.buf.numel()/2**30, eng.arena_b.peak/2**30, eng.arena_b.buf.numel()/2**30))
