"""ctypes binding of libvangan_hip.so (include/vangan_hip.h).  There is NO fallback: if the library cannot be
built or loaded, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the CDLL below: the process must use the HIP runtime torch ships, otherwise
#                         libvangan_hip.so binds /opt/rocm's libamdhip64 first and torch then runs on a second copy

from . import build as _build

VG_MAX_TAPS = 64
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1
SCRATCH_CTR_BYTES = 16384          # VG_SCRATCH_CTR_BYTES

c_void_p, c_int, c_float, c_i64, c_u64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_uint64


class ConvDesc(C.Structure):
    _fields_ = [
        ('src0', c_void_p), ('src1', c_void_p), ('c_src0', c_int), ('c_src1', c_int), ('src0_shift', c_int),
        ('src_f32', c_int), ('N', c_int), ('D', c_int), ('H', c_int), ('W', c_int),
        ('in_scale', c_void_p), ('in_shift', c_void_p), ('act', c_int), ('noise', c_void_p), ('noise_pad', c_int),
        ('istr', c_int), ('pad_mode', c_int), ('ntaps', c_int),
        ('tap_d', C.c_int8 * VG_MAX_TAPS), ('tap_h', C.c_int8 * VG_MAX_TAPS), ('tap_w', C.c_int8 * VG_MAX_TAPS),
        ('OD', c_int), ('OH', c_int), ('OW', c_int), ('ostr', c_int), ('ooff_d', c_int), ('ooff_h', c_int),
        ('ooff_w', c_int), ('BD', c_int), ('BH', c_int), ('BW', c_int), ('Cout', c_int),
        ('wpacked', c_void_p), ('CK', c_int), ('bias', c_void_p),
        ('res', c_void_p), ('res_scale', c_void_p), ('res_shift', c_void_p), ('tanh_out', c_int),
        ('out', c_void_p), ('out_f32', c_int), ('accumulate', c_int), ('out_sums', c_void_p), ('f32', c_int),
        ('nclass', c_int), ('cls_tap0', c_int * 9), ('cls_w', c_void_p * 8), ('cls_ooff', (c_int * 3) * 8),
        ('cls_iters', (c_int * 3) * 8), ('wpack', c_int), ('wpack_wmin', c_int), ('bstat', c_void_p),
        ('scratch', c_void_p), ('scratch_bytes', c_i64), ('wlayout', c_int), ('fin', c_void_p), ('res_c1', c_int), ('wpacked_up', c_void_p),
    ]


class FinJob(C.Structure):
    _fields_ = [('gamma', c_void_p), ('beta', c_void_p), ('mult', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
                ('mean', c_void_p), ('rstd', c_void_p), ('c_off', c_int), ('c_tot', c_int)]


class FinDesc(C.Structure):
    _fields_ = [('ticket', c_void_p), ('count', c_float), ('eps', c_float), ('njobs', c_int), ('job', FinJob * 2)]


class PackItem(C.Structure):
    _fields_ = [('w', c_void_p), ('tap_idx', c_void_p), ('out', c_void_p), ('Cin', c_int), ('Cout', c_int), ('ntaps', c_int),
                ('transpose', c_int), ('CK', c_int), ('out_f32', c_int), ('blk0', c_int), ('nblk', c_int), ('bn', c_int), ('pad_', c_int)]


class ActNormBwdDesc(C.Structure):
    _fields_ = [
        ('g', c_void_p), ('g_padded', c_int), ('x', c_void_p), ('x_f32', c_int),
        ('x1', c_void_p), ('c_x0', c_int), ('x0_shift', c_int),
        ('N', c_int), ('D', c_int), ('H', c_int), ('W', c_int), ('C', c_int),
        ('scale', c_void_p), ('shift', c_void_p), ('mult', c_void_p), ('act', c_int), ('norm', c_int),
        ('gamma', c_void_p), ('mean', c_void_p), ('rstd', c_void_p), ('red', c_void_p),
        ('dx', c_void_p), ('dx_f32', c_int), ('accumulate', c_int), ('dx_cstride', c_int), ('dx_coff', c_int),
        ('f32', c_int), ('dgamma', c_void_p), ('dbeta', c_void_p), ('ticket', c_void_p),
        ('alias_n0', c_int), ('alias_shift', c_int), ('pgrad_n', c_int), ('pad_', c_int),
    ]


_SIGS = {
    'vg_version': ([], c_int),
    'vg_storage16': ([], c_int),
    'vg_abi_sizeof': ([c_int], c_int),
    'vg_set_stamp_buffer': ([c_void_p], c_int),
    'vg_conv3d': ([C.POINTER(ConvDesc), c_void_p], c_int),
    'vg_conv3d_lds_bytes': ([C.POINTER(ConvDesc)], c_int),
    'vg_conv3d_plan': ([C.POINTER(ConvDesc), C.POINTER(C.c_int32)], c_int),
    'vg_conv3d_variant': ([C.POINTER(ConvDesc), C.c_char_p, c_int], c_int),
    'vg_conv3d_thin_np': ([C.POINTER(ConvDesc)], c_int),
    'vg_set_tuning': ([C.c_char_p, c_int, c_int], c_int),
    'vg_conv3d_wgrad_variant': ([C.POINTER(ConvDesc), c_int, C.POINTER(c_int), c_int, c_i64, C.c_char_p, c_int], c_int),
    'vg_pack_weights': ([c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p], c_int),
    'vg_pack_weights_multi': ([c_void_p, c_int, c_int, c_void_p], c_int),
    'vg_crop_augment': ([c_void_p] + [c_int] * 13 + [c_void_p, c_void_p], c_int),
    'vg_crop_max': ([c_void_p] + [c_int] * 10 + [c_void_p, c_void_p], c_int),
    'vg_conv3d_dma_bn': ([C.POINTER(ConvDesc)], c_int),
    'vg_conv3d_scratch_bytes': ([C.POINTER(ConvDesc)], c_i64),
    'vg_pack_weights_dma': ([c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    'vg_packed_ktot': ([c_int, c_int, c_int], c_int),
    'vg_packed_rows': ([c_int], c_int),
    'vg_conv3d_wgrad': ([C.POINTER(ConvDesc), c_void_p, c_int, C.POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p, c_i64,
                         c_void_p], c_int),
    'vg_in_finalize': ([c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int,
                        c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    'vg_actnorm_bwd_stats': ([C.POINTER(ActNormBwdDesc), c_void_p], c_int),
    'vg_actnorm_bwd_apply': ([C.POINTER(ActNormBwdDesc), c_void_p], c_int),
    'vg_actnorm_bwd_apply2': ([C.POINTER(ActNormBwdDesc), C.POINTER(ActNormBwdDesc), c_void_p], c_int),
    'vg_actnorm_bwd': ([C.POINTER(ActNormBwdDesc), c_void_p], c_int),
    'vg_stem_short_fwd_workgroups': ([c_int, c_i64], c_int),
    'vg_stem_short_fwd': ([c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p, c_void_p, C.c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                          c_void_p], c_int),
    'vg_stem_short_bwd_workgroups': ([c_int, c_i64, c_int], c_int),
    'vg_stem_short_bwd': ([c_void_p, c_int, c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p, C.c_float, c_int, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_int, c_void_p, c_void_p], c_int),
    'vg_in_param_grads': ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    'vg_concat_bwd': ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p], c_int),
    'vg_shortcut_dgrad_concat': ([C.POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_int, c_void_p], c_int),
    'vg_shortcut_dgrad_concat_norm': ([C.POINTER(ConvDesc), C.POINTER(ActNormBwdDesc), c_void_p, c_void_p, c_int, c_int, c_void_p], c_int),
    'vg_affine_add': ([c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p, c_int, c_void_p], c_int),
    'vg_bias_grad': ([c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p], c_int),
    'vg_pack_cell_weights': ([c_void_p, c_int, c_void_p, c_void_p], c_int),
    'vg_cells_fold': ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    'vg_tanh_bwd': ([c_void_p, c_void_p, c_void_p, c_i64, c_void_p], c_int),
    'vg_minmax': ([c_void_p, c_int, c_i64, c_void_p, c_void_p], c_int),
    'vg_minmax_apply': ([c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p], c_int),
    'vg_minmax_bwd': ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p], c_int),
    'vg_bce': ([c_void_p, c_void_p, c_i64, c_void_p, c_float, c_void_p, c_int, c_void_p], c_int),
    'vg_mse': ([c_void_p, c_void_p, c_i64, c_void_p, c_float, c_void_p, c_int, c_void_p], c_int),
    'vg_mse_const': ([c_void_p, c_int, c_float, c_i64, c_void_p, c_float, c_void_p, c_int, c_void_p], c_int),
    'vg_ssim_fwd': ([c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    'vg_ssim_bwd': ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p], c_int),
    'vg_soft_skel_fwd': ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    'vg_soft_skel_bwd': ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    'vg_cldice_coef': ([c_void_p, c_float, c_float, c_void_p, c_void_p], c_int),
    'vg_cldice_grads': ([c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p], c_int),
    'vg_dense_head_fwd': ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p], c_int),
    'vg_dense_head_bwd': ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    'vg_wasserstein_terms': ([c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    'vg_dot_sums': ([c_void_p, c_void_p, c_i64, c_void_p, c_void_p], c_int),
    'vg_overlap_add': ([c_void_p] + [c_int] * 12 + [c_void_p, c_void_p, c_void_p], c_int),
    'vg_divide_crop': ([c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p], c_int),
    'vg_axpby': ([c_void_p, c_float, c_void_p, c_float, c_i64, c_void_p, c_int, c_void_p], c_int),
    'vg_adam_clip': ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_float, c_float,
                      c_float, c_float, c_float, c_float, c_void_p], c_int),
    'vg_local_exchange': ([c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p], c_int),
    'vg_randn_bf16': ([c_void_p, c_i64, c_float, c_u64, c_u64, c_void_p], c_int),
    'vg_dropout_mask': ([c_void_p, c_i64, c_float, c_u64, c_u64, c_void_p], c_int),
    'vg_randn_bf16_dev': ([c_void_p, c_i64, c_void_p, c_u64, c_void_p, c_u64, c_void_p], c_int),
    'vg_dropout_mask_dev': ([c_void_p, c_i64, c_float, c_u64, c_void_p, c_u64, c_void_p], c_int),
    'vg_adam_clip_dev': ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_float,
                          c_float, c_float, c_float, c_float, c_void_p], c_int),
    'vg_set_step_params': ([c_void_p, c_u64, c_float, c_float, c_float, c_float, c_float, c_void_p], c_int),
    'vg_pack_up_weights': ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    'vg_memset_zero': ([c_void_p, c_i64, c_void_p], c_int),
    'vg_copy_bytes': ([c_void_p, c_void_p, c_i64, c_void_p], c_int),
    'vg_f32_to_bf16': ([c_void_p, c_void_p, c_i64, c_void_p], c_int),
    'vg_bf16_to_f32': ([c_void_p, c_void_p, c_i64, c_void_p], c_int),
}

EXPORTS = sorted(list(_SIGS.keys()) + ['vg_status_string'])


def _load(path=None, storage16=0):
    """Build if the sources changed, then dlopen.  A stale library is only accepted on explicit request (VG_NO_REBUILD=1, used
    on the GPU box where the prebuilt .so travels with the snapshot): the descriptor structs are mirrored by hand in this
    file, so a library built from other sources could corrupt memory instead of failing.  Either way the struct sizes the
    library was compiled with are checked against the ctypes mirrors below."""
    path = _build.LIB if path is None else path
    if not os.path.exists(path) or (_build.needs_build() and os.environ.get('VG_NO_REBUILD') != '1'):
        _build.build()                               # raises on failure: no silent fallback to an old binary
    lib = C.CDLL(path)
    for name, (args, ret) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ret
    lib.vg_status_string.argtypes = [c_int]
    lib.vg_status_string.restype = C.c_char_p
    for which, mirror in ((0, ConvDesc), (1, ActNormBwdDesc), (2, PackItem), (3, FinDesc)):
        got = lib.vg_abi_sizeof(which)
        if got != C.sizeof(mirror):
            raise ImportError('libvangan_hip.so ABI mismatch: sizeof(%s) is %d in the library, %d in van_gan_amd/_lib.py '
                              '(stale build? remove %s and rebuild)' % (mirror.__name__, got, C.sizeof(mirror), path))
    if lib.vg_storage16() != storage16:
        raise ImportError('%s stores %s in its 16-bit buffers, expected %s' % (path, ('bf16', 'fp16')[lib.vg_storage16()], ('bf16', 'fp16')[storage16]))
    return lib


lib = _load()
_lib_h = None


def lib_fp16():
    """libvangan_hip_h.so: the same entry points with IEEE half precision in every 16-bit buffer (inference only; loaded on
    first use, RTLD_LOCAL: its symbols do not meet the bf16 library's)."""
    global _lib_h
    if _lib_h is None:
        _lib_h = _load(_build.LIB_H, 1)
    return _lib_h


class VgError(RuntimeError):
    pass


def check(rc: int, what: str = '') -> int:
    if rc < 0:
        raise VgError('%s failed: %s (%d)' % (what or 'libvangan_hip call', lib.vg_status_string(rc).decode(), rc))
    return rc
