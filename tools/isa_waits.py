#!/usr/bin/env python3
"""Where does a kernel wait for memory?  Compiles one csrc file to gfx950 assembly and prints, per kernel (or for the kernels whose
mangled name contains --kernel), a compact trace of its control flow: labels, branches, barriers, global / scratch / LDS-DMA memory
instructions and every s_waitcnt, with the vector-ALU and MFMA instructions between them counted.  Flags the two patterns that cost the
thin specialist 10-40 % of its launches in round 5:

  COPY   an `s_waitcnt vmcnt(n)` directly followed by v_mov copies -- a loop-carried (or branch-merged) loaded value that the register
         allocator gave fresh registers and copies home: the whole round trip is in the open at that point;
  STORE  a global load issued after a global store in the same basic block run -- the compiler cannot move it above the store (may
         alias), and vmcnt counts in order, so its use waits for the store as well.

Development aid (no GPU needed):  python tools/isa_waits.py vg_conv_thin.hip --kernel conv_thin_kernelILi0ELb0ELb0ELb0ELb1ELi1ELi3
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'van_gan_amd', 'csrc')


def assemble(src):
    out = tempfile.NamedTemporaryFile(suffix='.s', delete=False).name
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only',
           '-I', os.path.join(ROOT, 'include'), os.path.join(CSRC, src), '-o', out]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        sys.exit(r.stdout.decode()[-3000:])
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    for m in re.finditer(r'^(_Z\w+):\s*; @\1$', text, re.M):
        end = text.find('.Lfunc_end', m.end())
        yield m.group(1), text[m.end():end].split('\n')


def trace(body, full):
    rows, valu, mfma, store_seen = [], 0, 0, False
    flags = {'COPY': 0, 'STORE': 0}
    for k, line in enumerate(body):
        t = line.strip()
        if not t or t.startswith(';'):
            continue
        if t.startswith('v_mfma'):
            mfma += 1
            continue
        if t.startswith('v_'):
            valu += 1
            continue
        if not re.match(r'(global_|buffer_|scratch_|s_waitcnt|s_barrier|s_cbranch|s_branch|s_endpgm|\.LBB)', t):
            continue
        tag = ''
        if t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch') or 's_barrier' in t:
            store_seen = False
        if t.startswith('global_store') or t.startswith('buffer_store'):
            store_seen = True
        if (t.startswith('global_load') and not t.startswith('global_load_lds')) and store_seen:
            tag = 'STORE'
        if t.startswith('s_waitcnt') and 'vmcnt' in t:
            nxt = [x.strip() for x in body[k + 1:k + 4]]
            if any(x.startswith('v_mov') or x.startswith('v_accvgpr') for x in nxt):
                tag = 'COPY'
        if tag:
            flags[tag] += 1
        if full or tag or 'vmcnt' in t or 's_barrier' in t or t.startswith('scratch_'):
            if mfma or valu:                                   # instructions since the previous printed row
                rows.append('        [%d mfma, %d valu]' % (mfma, valu))
                mfma = valu = 0
            rows.append('%6d  %-70s %s' % (k, t[:70], tag))
    return rows, flags


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('source', help='file under van_gan_amd/csrc, e.g. vg_conv_thin.hip')
    ap.add_argument('--kernel', default='', help='substring of the mangled kernel name (default: a summary line per kernel)')
    ap.add_argument('--full', action='store_true', help='with --kernel: every label, branch and memory instruction, not only the waits')
    a = ap.parse_args()
    text = assemble(a.source)
    for name, body in kernels(text):
        if a.kernel and a.kernel not in name:
            continue
        rows, flags = trace(body, a.full)
        nv = re.search(r'; NumVgprs: (\d+)', text[text.find('.Lfunc_end', text.find(name + ':')):][:4000])
        spill = sum(1 for l in body if 'scratch_' in l)
        print('%s  lines %d  vgprs %s  scratch ops %d  vmcnt(0) %d  COPY %d  STORE %d' % (
            name, len(body), nv.group(1) if nv else '?', spill, sum(1 for l in body if 'vmcnt(0)' in l), flags['COPY'], flags['STORE']))
        if a.kernel:
            print('\n'.join(rows))


if __name__ == '__main__':
    main()
