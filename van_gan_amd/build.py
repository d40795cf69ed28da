"""Build libvangan_hip.so (gfx950 only) in-tree with hipcc.  No fallback: a missing compiler is an error."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvangan_hip.so')
SOURCES = ['vg_conv.hip', 'vg_wgrad.hip', 'vg_elem.hip', 'vg_loss.hip', 'vg_adam.hip']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libvangan_hip.so cannot be built')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'vangan_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (s, out.decode()))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode())
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
