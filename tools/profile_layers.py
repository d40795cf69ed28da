"""Per-layer breakdown of the gather-convolution launches of one full 128^3 train step (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from van_gan_amd import VanGan, ops
from van_gan_amd.synth import synth_volumes

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = VanGan((size,) * 3, batch_size=1, device='cuda:0')
rI, rS = synth_volumes(1, size, size, size, seed=1)
rI, rS = rI.cuda(), rS.cuda()
eng.train_step(rI, rS)
orig_end = ops.KernelProfile.end
ops.PROF = ops.KernelProfile()
import van_gan_amd.ops as O
# tag rows by layer name
def fwd(self, src, out, **kw):
    e0 = ops.PROF.begin(); r = _f(self, src, out, **kw); ops.PROF.end('fwd  ' + self.name + ' %s' % (tuple(self.in_dims),), 1.0, e0); return r
def wg(self, src, dy):
    e0 = ops.PROF.begin(); r = _w(self, src, dy); ops.PROF.end('wgrad ' + self.name + ' %s' % (tuple(self.in_dims),), 1.0, e0); return r
def dg(self, dy, N, out, accumulate):
    e0 = ops.PROF.begin(); r = _d(self, dy, N, out, accumulate); ops.PROF.end('dgrad ' + self.name + ' %s' % (tuple(self.in_dims),), 1.0, e0); return r
_f, _w, _d = O.ConvLayer.forward, O.ConvLayer.wgrad, O.ConvLayer.dgrad
P = ops.PROF
ops.PROF = None
O.ConvLayer.forward, O.ConvLayer.wgrad, O.ConvLayer.dgrad = fwd, wg, dg
ops.PROF = None
class Shim:
    def begin(self): return P.begin()
    def end(self, k, f, e): P.end(k, f, e)
# run one step with tagging (inner PROF disabled)
ops.PROF = None
import types
ops_PROF_backup = None
# monkeypatch: our wrappers use ops.PROF inside; set to P only inside wrappers
def fwd2(self, src, out, **kw):
    e0 = P.begin(); r = _f(self, src, out, **kw); P.end('fwd   %-18s %s' % (self.name, tuple(self.in_dims)), 1.0, e0); return r
def wg2(self, src, dy):
    e0 = P.begin(); r = _w(self, src, dy); P.end('wgrad %-18s %s' % (self.name, tuple(self.in_dims)), 1.0, e0); return r
def dg2(self, dy, N, out, accumulate):
    e0 = P.begin(); r = _d(self, dy, N, out, accumulate); P.end('dgrad %-18s %s' % (self.name, tuple(self.in_dims)), 1.0, e0); return r
O.ConvLayer.forward, O.ConvLayer.wgrad, O.ConvLayer.dgrad = fwd2, wg2, dg2
eng.train_step(rI, rS)
s = P.summary()
tot = sum(v['ms'] for v in s.values())
print('total conv ms %.2f' % tot)
for k, v in sorted(s.items(), key=lambda kv: -kv[1]['ms'])[:45]:
    print('%-44s n=%3d  %7.3f ms  %5.1f%%' % (k, v['launches'], v['ms'], 100 * v['ms'] / tot))
