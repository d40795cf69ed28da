"""Soak run (development aid): resident synthetic volumes -> DataPipeline -> VanGan.train_step under the GanMonitor schedules for a
few hundred steps; prints the epoch means and checks that everything stays finite and the arena does not grow.
usage: python tools/soak.py [size=64] [epochs=6] [steps=40]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from van_gan_amd import data  # noqa: E402
from van_gan_amd.train import GanMonitor, fit  # noqa: E402
from van_gan_amd.vangan import VanGan  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
V = size + 32


def tubes(n):
    """a few random bright tubes in a dark volume (segmentation-like), and a blurred noisy copy (imaging-like)"""
    seg = np.zeros((V, V, V, 1), np.float32)
    for _ in range(n):
        p = rng.integers(4, V - 4, 3).astype(np.float64); d = rng.standard_normal(3); d /= np.linalg.norm(d)
        for t in range(3 * V):
            q = np.round(p + d * (t - 1.5 * V)).astype(int)
            if ((q >= 1) & (q < V - 1)).all():
                seg[q[0] - 1:q[0] + 2, q[1] - 1:q[1] + 2, q[2] - 1:q[2] + 2, 0] = 1.0
    img = seg + 0.3 * rng.standard_normal(seg.shape).astype(np.float32)
    return torch.from_numpy(img).to(dev), torch.from_numpy(seg).to(dev)


vols = [tubes(12) for _ in range(3)]
pipe = data.DataPipeline([v[0] for v in vols], [v[1] for v in vols], (size,) * 3, 1, seed=1)
gan = VanGan((size,) * 3, batch_size=1, device='cuda:0', seed=0, layer_noise=0.9)
mon = GanMonitor(EPOCHS=epochs, INITIATE_LR_DECAY=epochs // 2, INITIAL_LR=2e-4, train_steps=steps, NO_NOISE=epochs - 1)
peak0 = None
hist = fit(gan, pipe, mon, val_ds=pipe, val_steps=2, save=False)
for h in hist:
    tr = h['train']
    assert all(math.isfinite(v) for v in tr.values()), h
    print('epoch %2d  lr %.2e  noise %.3f  total_IS %.3f  total_SI %.3f  cyc_SIS %.3f  cyc_ISI %.3f  D_I %.3f  D_S %.3f' % (
        h['epoch'], h['lr'], h['noise'], tr['total_IS_loss'], tr['total_SI_loss'], tr['cycle_gen_SIS_loss'],
        tr['cycle_gen_ISI_loss'], tr['D_I_loss'], tr['D_S_loss']))
print('arena peak GB %.2f' % (gan.arena.peak / 1e9), ' torch allocated GB %.2f' % (torch.cuda.memory_allocated() / 1e9))
print('soak OK: %d steps' % (epochs * steps))
