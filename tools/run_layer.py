"""Run one conv layer configuration (forward, optionally dgrad/wgrad) a few times -- target for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import ops
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0')
cases = {'stem': (3, 16, 16, 1, 'reflect', 128, None), 'dec0': (3, 48, 16, 1, 'reflect', 128, (32, 16)), 'enc2': (3, 64, 64, 1, 'reflect', 32, None),
         'enc1': (3, 32, 32, 1, 'reflect', 64, None), 'down2': (4, 256, 512, 1, 'same', 16, None), 'down0': (4, 64, 128, 2, 'reflect', 64, None), 'bridge': (3, 256, 256, 1, 'reflect', 8, None), 'enc3': (3, 128, 128, 1, 'reflect', 16, None), 'down1': (4, 128, 256, 2, 'reflect', 32, None)}
name = sys.argv[1] if len(sys.argv) > 1 else 'stem'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else 'fwd'
k, cin, cout, stride, pad, S, cat = cases[name]
dims = (S,) * 3
st = ParamStore([('c.w', (k, k, k, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
st.param('c.w').normal_(0, 0.05)
lay = ConvLayer(st, 'c', k, cin, cout, stride, pad, True, dims); lay.pack()
N = 2 if name.startswith('down') else 1
sc, sh = torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.1
if cat:
    low = torch.randn(N, S // 2, S // 2, S // 2, cat[0], device=dev).to(torch.bfloat16)
    skip = torch.randn(N, *dims, cat[1], device=dev).to(torch.bfloat16)
    src = Src(low, (N,) + dims, cat[0], skip, cat[1], shift0=1, scale=sc, shift=sh, act=ops.ACT_RELU)
else:
    src = Src(torch.randn(N, *dims, cin, device=dev).to(torch.bfloat16), (N,) + dims, cin, scale=sc, shift=sh, act=ops.ACT_RELU)
out = torch.zeros(N, *lay.out_dims, cout, dtype=torch.bfloat16, device=dev)
sums = torch.zeros(8, N, cout, 2, device=dev)
dy = torch.randn(N, *lay.out_dims, cout, device=dev).to(torch.bfloat16)
dp = torch.zeros(N, *lay.buf_dims, cin, dtype=torch.bfloat16, device=dev)
for _ in range(reps):
    if mode == 'fwd':
        lay.forward(src, out, sums=sums)
    elif mode == 'dgrad':
        lay.dgrad(dy, N, dp, False)
    else:
        lay.wgrad(src, dy)
torch.cuda.synchronize()
print('done', name)
