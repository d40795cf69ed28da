"""Development aid: phase cycle counts of wgrad_thin_kernel (VG_WT_DBG=8: s_memtime sums by thread 0 of every workgroup: wait at the tile's
first barrier / commit / wait at the second barrier / issue + K loop), stem.cb-like layer at 128^3, two volumes."""
import os, sys
os.environ['VG_WT_DBG'] = '8'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from van_gan_amd import ops
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0'); ops.set_device(0)
dims, N = (128, 128, 128), 2
for cin in (16, 48):
    st = ParamStore([('c.w', (3, 3, 3, cin, 16), 'he_normal'), ('c.b', (16,), 'zeros')], dev)
    lay = ConvLayer(st, 'c', 3, cin, 16, 1, 'reflect', True, dims); lay.pack()
    x = torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16)
    src = Src(x, (N,) + dims, cin, scale=torch.rand(N, cin, device=dev) + 0.5, shift=torch.randn(N, cin, device=dev) * 0.1, act=ops.ACT_RELU)
    dy = torch.randn(N, *dims, 16, device=dev).to(torch.bfloat16)
    for _ in range(3):
        lay.wgrad(src, dy)
    torch.cuda.synchronize()
    sc = list(ops.WGRAD_SCRATCH.values())[0]
    nch = cin // 16
    bx = (512 // (nch * N)) & ~7
    nslab = bx * N
    off = nch * nslab * 7168 + nch * 28 * 8 * 256 + 4096          # floats: slabs | part2 | tickets(+4096 words)
    raw = sc[off:off + nch * nslab * 8].view(torch.int64).view(-1, 4).cpu().double()
    tiles = 4096 / bx
    print('%d -> 16: workgroups %d, tiles per workgroup %.1f; s_memtime ticks per tile: wait0 %.0f  commit %.0f  wait1 %.0f  issue+K %.0f'
          % (cin, raw.shape[0], tiles, *(raw.mean(0) / tiles).tolist()))
