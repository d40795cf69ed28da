"""Build libvangan_hip.so and libvangan_hip_h.so (gfx950 only) in-tree with hipcc.  No fallback: a missing compiler is an error.
The second library is the SAME sources compiled with -DVG_FP16: its 16-bit buffers hold IEEE half precision instead of bfloat16
(fp16 sliding-window inference, BASELINE config 5)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvangan_hip.so')
LIB_H = os.path.join(HERE, 'libvangan_hip_h.so')
SOURCES = ['vg_conv.hip', 'vg_conv_thin.hip', 'vg_conv_dma.hip', 'vg_wgrad.hip', 'vg_wgrad_dma.hip', 'vg_pointwise.hip', 'vg_c1k3.hip', 'vg_elem.hip', 'vg_loss.hip', 'vg_adam.hip']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libvangan_hip.so cannot be built')


HASHFILE = LIB + '.srchash'
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17']


def _extra_defs():
    """Development builds only, e.g. VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE (the timing-ablation knobs of vg_conv_thin.hip, whose results are
    wrong by design, exist only in such a build)."""
    return os.environ.get('VG_EXTRA_DEFS', '').split()


_CC_ID = None


def _compiler_id() -> str:
    """`hipcc --version`: objects of another compiler are never linked with fresh ones."""
    global _CC_ID
    if _CC_ID is None:
        try:
            _CC_ID = subprocess.run([_hipcc(), '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
        except OSError:
            _CC_ID = 'unknown'
    return _CC_ID


def _src_hash() -> str:
    """Content hash of every source that goes into the library (mtimes do not survive the copy to the GPU box)."""
    import hashlib
    h = hashlib.sha256(' '.join(_extra_defs()).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, '..', 'include', 'vangan_hip.h')]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB_H) or not os.path.exists(HASHFILE):
        return True
    with open(HASHFILE) as f:
        return f.read().strip() != _src_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Serialised by a file lock: under torch.distributed.run every rank imports the package at the same time."""
    if not force and not needs_build():
        return LIB
    import fcntl
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    with open(os.path.join(HERE, 'build', '.lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # another rank built it while this one waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _obj_hash(src: str, defs) -> str:
    """What an object file depends on: its source, every header / include file of csrc/, the public header, the whole command line
    and the compiler's identity."""
    import hashlib
    h = hashlib.sha256((' '.join(FLAGS + list(defs)) + '\n' + _compiler_id()).encode())
    deps = [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith('.hip'))
    deps.append(os.path.join(HERE, '..', 'include', 'vangan_hip.h'))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _build_locked(verbose: bool) -> str:
    hipcc = _hipcc()
    procs = []
    variants = (('', LIB, []), ('h_', LIB_H, ['-DVG_FP16']))
    objs = {LIB: [], LIB_H: []}
    for tag, lib, defs in variants:
        for s in SOURCES:
            o = os.path.join(HERE, 'build', tag + s.replace('.hip', '.o'))
            objs[lib].append(o)
            defs = defs + _extra_defs()
            oh = _obj_hash(s, defs)
            try:                                  # incremental: an object whose inputs did not change is kept
                with open(o + '.hash') as f:
                    if os.path.exists(o) and f.read().strip() == oh:
                        continue
            except OSError:
                pass
            # the object is written beside its final name and renamed on success, and its old hash goes first: a compile that is
            # killed half-way leaves neither a truncated object under the final name nor a hash that vouches for one
            try:
                os.remove(o + '.hash')
            except OSError:
                pass
            otmp = o + '.tmp.%d' % os.getpid()
            cmd = [hipcc] + FLAGS + defs + ['-c', os.path.join(CSRC, s), '-o', otmp]
            if verbose:
                print(' '.join(cmd))
            procs.append((tag + s, o, otmp, oh, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, o, otmp, oh, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (s, out.decode()))
        os.replace(otmp, o)
        with open(o + '.hash', 'w') as f:
            f.write(oh)
    for _, lib, _ in variants:
        tmp = lib + '.tmp.%d' % os.getpid()
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs[lib]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stdout.decode())
        os.replace(tmp, lib)          # atomic: a process that already mapped the old library keeps its inode
    with open(HASHFILE, 'w') as f:
        f.write(_src_hash())
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
