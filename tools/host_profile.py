"""cProfile of the host side of one train step (development aid): where the enqueue time goes.  usage: host_profile.py [size] [batch]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan
from van_gan_amd.synth import synth_volumes
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = VanGan((size,) * 3, batch_size=B, device='cuda:0')
rI, rS = synth_volumes(B, size, size, size, seed=1)
rI, rS = rI.cuda(), rS.cuda()
for _ in range(3):
    eng.train_step(rI, rS, sync=False)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); eng.train_step(rI, rS, sync=False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('enqueue %.2f ms, total %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eng.train_step(rI, rS, sync=False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
