"""CPU tests: the C-ABI library builds/loads and exports every symbol declared in include/vangan_hip.h (no compute
calls: there is no GPU here), host-side geometry (taps, output-parity classes of the data gradient) against a brute-
force index enumeration, descriptor validation, flat parameter store layout."""
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from van_gan_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'vangan_hip.h')).read()
    declared = set(re.findall(r'^(?:int|int64_t|const char\*)\s+(vg_[a-z0-9_]+)\s*\(', hdr, flags=re.M))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), 'libvangan_hip.so does not export %s' % name
    assert set(_lib.EXPORTS) == declared, (set(_lib.EXPORTS) ^ declared)
    assert _lib.lib.vg_version() == 3 and _lib.lib.vg_storage16() == 0
    assert _lib.lib.vg_status_string(-2).decode().startswith('tile')
    # the fp16-storage build of the same sources (libvangan_hip_h.so): same export set, same descriptor layout, reports itself
    h = _lib.lib_fp16()
    for name in sorted(declared):
        assert hasattr(h, name), 'libvangan_hip_h.so does not export %s' % name
    assert h.vg_storage16() == 1 and h.vg_version() == 3
    assert [h.vg_abi_sizeof(i) for i in range(4)] == [_lib.lib.vg_abi_sizeof(i) for i in range(4)]


def test_descriptor_struct_layout_matches_header():
    """ctypes mirror of vg_conv_desc must have the C layout (pointers 8-byte aligned, int8 tap arrays of 64)."""
    from van_gan_amd._lib import ActNormBwdDesc, ConvDesc
    assert C.sizeof(ConvDesc) % 8 == 0 and ConvDesc.tap_d.size == 64 and ConvDesc.tap_h.offset == ConvDesc.tap_d.offset + 64
    assert ConvDesc.src1.offset == 8 and ConvDesc.in_scale.offset % 8 == 0 and ConvDesc.out_sums.offset % 8 == 0
    assert ActNormBwdDesc.x.offset % 8 == 0 and C.sizeof(ActNormBwdDesc) % 8 == 0


def test_invalid_descriptors_are_rejected_without_gpu():
    from van_gan_amd._lib import ConvDesc, lib
    d = ConvDesc()
    assert lib.vg_conv3d_lds_bytes(C.byref(d)) == -1                       # null pointers
    d.src0 = d.out = d.wpacked = 1 << 20
    d.c_src0, d.N, d.D, d.H, d.W = 12, 1, 8, 8, 8                           # 12 channels: not a multiple of 8
    d.istr, d.ostr, d.ntaps, d.OD, d.OH, d.OW, d.BD, d.BH, d.BW, d.Cout, d.CK = 1, 1, 1, 8, 8, 8, 8, 8, 8, 16, 16
    assert lib.vg_conv3d_lds_bytes(C.byref(d)) == -1
    d.c_src0 = 16
    assert lib.vg_conv3d_lds_bytes(C.byref(d)) > 0
    d.CK = 24
    assert lib.vg_conv3d_lds_bytes(C.byref(d)) == -1
    assert lib.vg_packed_ktot(27, 48, 48) == 1312 and lib.vg_packed_ktot(27, 16, 16) == 448 and lib.vg_packed_rows(1) == 64


@pytest.mark.parametrize('k,stride,pad,n', [(3, 1, 'reflect', 8), (3, 2, 'reflect', 8), (4, 2, 'reflect', 8), (4, 1, 'same', 6),
                                            (1, 2, 'same', 8), (3, 1, 'same', 6), (1, 1, 'same', 4)])
def test_dgrad_classes_cover_the_transposed_convolution(k, stride, pad, n):
    """The output-parity classes + tap offsets of ConvLayer enumerate exactly the (input position, tap, output position)
    triples of the forward convolution (1-D check per axis; the 3-D classes are their cartesian product)."""
    from van_gan_amd.nets import ParamStore
    from van_gan_amd.ops import ConvLayer
    st = ParamStore([('c.w', (k, k, k, 16, 16), 'x'), ('c.b', (16,), 'x')], 'cpu')
    lay = ConvLayer(st, 'c', k, 16, 16, stride, pad, True, (n, n, n))
    no = lay.out_dims[0]
    pb = 1 if pad == 'reflect' else lay.pb[0]
    # forward triples on the buffer grid (padded coordinates for 'reflect'): o*s + t - pb_eff = i
    pbe = 0 if pad == 'reflect' else pb
    want = {(o * stride + t - pbe, t, o) for o in range(no) for t in range(k) if 0 <= o * stride + t - pbe < lay.buf_dims[0]}
    got = set()
    for c in lay.d_classes:
        taps1d = sorted({(td, i // (k * k)) for (td, _, _), i in zip(c['taps'], c['idx'].tolist())})
        for j in range(c['iters'][0]):
            i = j * stride + c['off'][0]
            for (off, t) in taps1d:
                o = j + off
                if 0 <= o < no:
                    got.add((i, t, o))
    assert got == want
    # forward taps: input offset = t - pb
    assert sorted({t[0] for t in lay.f_taps}) == [t - pb for t in range(k)]
    assert lay.out_dims[0] == ((n + 2 - k) // stride + 1 if pad == 'reflect' else -(-n // stride))


def test_param_store_layout_and_roundtrip():
    from van_gan_amd.nets import ParamStore, disc_param_specs, gen_param_specs, init_reference
    from oracle import vangan_oracle as O
    assert [(n, tuple(s)) for n, s, _ in gen_param_specs()] == [(n, tuple(s)) for n, s, _ in O.gen_param_specs()]
    assert [(n, tuple(s)) for n, s, _ in disc_param_specs()] == [(n, tuple(s)) for n, s, _ in O.disc_param_specs()]
    st = ParamStore(gen_param_specs(), 'cpu')
    assert st.total == 9538929 and st.T == 116 and int(st.seg_off[-1]) == st.total
    init_reference(st, 3)
    P = st.export()
    st2 = ParamStore(gen_param_specs(), 'cpu')
    st2.load(P)
    assert torch.equal(st.w, st2.w)
    w = P['enc1.cb1.conv.w']
    fan_in = 27 * 16
    assert abs(float(w.std()) - (2.0 / fan_in) ** 0.5) < 0.1 * (2.0 / fan_in) ** 0.5       # he_normal
    assert float(P['stem.cb.in.gamma'].min()) == 1.0 and float(P['out.b'].abs().max()) == 0.0


def test_engine_refuses_to_run_without_gpu():
    from van_gan_amd import VanGan
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        VanGan((32, 32, 32))


def test_library_never_allocates_device_memory():
    """Boundary contract (include/vangan_hip.h, SURVEY 8b 'Ownership'): the library is enqueue-only on caller-owned buffers.  Neither
    build may even IMPORT an allocating / freeing / synchronising HIP entry point (undefined dynamic symbols of the built .so)."""
    import subprocess
    from van_gan_amd import build
    banned = re.compile(r'\b(hipMalloc\w*|hipFree\w*|hipHostMalloc|hipMallocAsync|hipDeviceSynchronize|hipStreamSynchronize|hipMemcpy)\b')
    for lib in (build.LIB, build.LIB_H):
        out = subprocess.run(['nm', '-D', '--undefined-only', lib], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert 'hipLaunchKernel' in out or 'hipModuleLaunchKernel' in out or '__hipPushCallConfiguration' in out      # sanity: nm sees the HIP imports
        hits = sorted(set(banned.findall(out)))
        assert not hits, '%s imports %s: the library must not allocate, free or synchronise' % (os.path.basename(lib), hits)


def test_conv_desc_scratch_fields_trail_the_struct():
    from van_gan_amd._lib import ConvDesc, SCRATCH_CTR_BYTES
    assert ConvDesc.scratch.offset == ConvDesc.bstat.offset + 8 and ConvDesc.scratch_bytes.offset == ConvDesc.scratch.offset + 8
    hdr = open(os.path.join(ROOT, 'include', 'vangan_hip.h')).read()
    assert int(re.search(r'#define VG_SCRATCH_CTR_BYTES (\d+)', hdr).group(1)) == SCRATCH_CTR_BYTES


def test_dma_layers_plan_at_any_batch():
    """the LDS-DMA layout is chosen per layer from the shape alone, its operand grows with the batch, and block-layout
    weights have no other kernel -- with a fixed 160 MB workspace the PatchGAN's data gradients at 128^3 planned for batch 4 and
    returned VG_EINVAL for batch 8.  The host now sizes the workspace per call from vg_conv3d_scratch_bytes: every batch plans."""
    import ctypes as C
    from van_gan_amd import ops
    from van_gan_amd._lib import lib
    from van_gan_amd.nets import ParamStore, PatchGAN, disc_param_specs
    dims = (128, 128, 128)
    D = PatchGAN(ParamStore(disc_param_specs(), 'cpu'), dims, torch.bfloat16)
    for B in (1, 8):                                                  # 2B samples through the discriminator ([real; fake])
        ar = ops.Arena(24 << 30, 'cpu')                              # never touched (dry runs do not write): virtual memory only
        with ops.DryRun() as dry:
            x2 = torch.empty((2 * B,) + dims + (1,))
            lg = torch.empty((2 * B,) + tuple(n // 8 for n in dims) + (1,))
            noise = {k: torch.empty(shp, dtype=torch.bfloat16) for k, shp in D.noise_shapes(2 * B).items()}
            drop = {k: torch.empty(2 * B, c) for k, c in (('down0', 128), ('down1', 256), ('down2', 512))}
            ctx = D.forward(ar, x2, lg, noise, drop)
            D.backward(ar, ctx, lg, 0, 2 * B, wgrad=False, dx=torch.empty((2 * B,) + dims + (1,)))
        dma = [v for k, _, v in dry.records if k == 'dgrad' and v.startswith('conv_dma')]
        assert len(dma) == 3, dry.records
    # and the query itself: grows with N, zero for classic-layout descriptors
    lay = D.L['down2']
    assert lay.d_bn
    need = []
    for N in (2, 16):
        d = lay._fused_desc(None, N, None, False, probe=True, plan=False)
        need.append(int(lib.vg_conv3d_scratch_bytes(C.byref(d))))
    assert need[1] > need[0] > 16384
    d.wlayout = 0
    assert lib.vg_conv3d_scratch_bytes(C.byref(d)) == 0


def test_tuning_registry_holds_every_key():
    """vg_set_tuning / vg_tune share one table of switches keyed by name (the VG_* environment variables without the prefix).  In round 5
    the library passed 64 distinct keys and the table was full: vg_set_tuning of a key first seen late in a process returned VG_EINVAL
    (unchecked by the callers) and the switch kept its default -- a test that forces a kernel family then silently tests the other one.
    The table holds 256 now; a rejected key is an error the caller can see."""
    from van_gan_amd import _lib
    for i in range(150):
        assert _lib.lib.vg_set_tuning(b'TEST_KEY_%03d' % i, i, 0) == 0
    for i in range(150):
        assert _lib.lib.vg_set_tuning(b'TEST_KEY_%03d' % i, 0, 1) == 0          # reset: back to environment / default
    assert _lib.lib.vg_set_tuning(b'X' * 40, 1, 0) < 0                           # over-long key: refused, not truncated
