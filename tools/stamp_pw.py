"""Wave start / end times (s_memrealtime, 100 MHz) of one pw_gemm launch: launch ramp vs in-kernel latency (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from van_gan_amd import ops
from van_gan_amd._lib import lib
from van_gan_amd.nets import ParamStore
from van_gan_amd.ops import ConvLayer, Src
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cin = cout = 16
st = ParamStore([('c.w', (1, 1, 1, cin, cout), 'x'), ('c.b', (cout,), 'x')], dev)
st.param('c.w').normal_(0, 0.05)
lay = ConvLayer(st, 'c', 1, cin, cout, 1, 'same', True, (S,) * 3); lay.pack()
dy = torch.randn(1, S, S, S, cout, device=dev).to(torch.bfloat16)
dp = torch.zeros(1, S, S, S, cin, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    lay.dgrad(dy, 1, dp, True)
torch.cuda.synchronize()
buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
lib.vg_set_stamp_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lay.dgrad(dy, 1, dp, True); e1.record(); torch.cuda.synchronize()
lib.vg_set_stamp_buffer(None)
b = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
nb = len(b) // 4
print('%d waves; event time %.1f us' % (len(b), e0.elapsed_time(e1) * 1e3))
names = ['start', 'weights in', 'x in', 'old in', '-', '-', '-', 'stores acked']
for x in range(8):
    print('XCD %d: ' % x + '  '.join('%s %.1f' % (names[k], (np.median(b[x::8 * 1][:, k].reshape(-1)) - t0) / 100) for k in (0, 1, 3, 2, 7)) if False else
          'XCD %d: ' % x + '  '.join('%s %.1f' % (names[k], (np.median(b.reshape(nb, 4, 8)[x::8, :, k]) - t0) / 100) for k in (0, 1, 3, 2, 7)))
