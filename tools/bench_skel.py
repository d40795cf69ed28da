"""Development aid: soft-skeleton forward / backward launch times by themselves (HIP events), at a power-of-two volume and at one whose
per-step stride is not a power of two (HBM channel aliasing between the per-step slabs shows as the difference)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from van_gan_amd import ops

dev = torch.device('cuda:0')
it = 15
for dims in ((1, 128, 128, 128), (1, 120, 128, 128), (1, 128, 128, 136)):
    B, D, H, W = dims
    vol = dims + (1,)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(vol, generator=g)
    x = torch.nn.functional.avg_pool3d(x.permute(0, 4, 1, 2, 3), 5, 1, 2).permute(0, 2, 3, 4, 1).contiguous().to(dev)
    imgs, skels = torch.zeros((it + 2,) + vol, device=dev), torch.zeros((it + 1,) + vol, device=dev)
    aux = torch.zeros(ops.skel_aux_bytes(dims, it), dtype=torch.uint8, device=dev)
    gskel = torch.randn(vol, generator=g).to(dev)
    gp = torch.zeros(vol, device=dev)
    work = torch.zeros((4,) + vol, device=dev)

    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_f0 = timeit(lambda: ops.soft_skel_fwd(x, dims, it, imgs, skels))
    t_f1 = timeit(lambda: ops.soft_skel_fwd(x, dims, it, imgs, skels, aux))
    t_b1 = timeit(lambda: ops.soft_skel_bwd(imgs, skels, gskel, dims, it, work, gp, aux))
    t_b0 = timeit(lambda: ops.soft_skel_bwd(imgs, skels, gskel, dims, it, work, gp))
    nz = float((aux[:(it + 1) * B * D * H * W * 4].view(torch.float32) > 0).float().mean())
    print('%s  fwd %.0f us  fwd+aux %.0f us  bwd(aux) %.0f us  bwd(scan) %.0f us   per Mvoxel: %.0f %.0f %.0f %.0f   delta>0: %.3f'
          % (dims, t_f0, t_f1, t_b1, t_b0, *[t / (B * D * H * W / 1e6) for t in (t_f0, t_f1, t_b1, t_b0)], nz))
