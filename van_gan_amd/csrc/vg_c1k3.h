// vg_c1k3.h -- launch parameters of the convolutions that read a SINGLE-channel volume (the stem's 3x3x3 1 -> 16, resunet_model.py:44-60;
// the discriminator's first 4x4x4 stride-2 1 -> 64, discriminator.py:50-60), shared by the VALU kernels of vg_pointwise.hip (3x3x3, any
// C <= 32, both storage types) and the MFMA kernels of vg_c1k3.hip (16-bit storage).
#pragma once
#include "vg_common.h"

struct C1K3 {
    const void* x; int x_f32; float sc, sf; int act, pad_mode;
    int D, H, W, C, W4;                     // C = Cout, W4 = quads per row
    int td0, th0, tw0;                      // offset of the first tap per axis (-pad_before)
    const void* w; int Ktot, CK;            // packed [Cout][Ktot], k = tap * CK + j  (W-packed layout)
    const float* bias; void* out; float* sums;
    const void* dy; float* dw; float* db;
    const float* scale; const float* shift;
    VgFin fin;
};
// position p of an axis of length n under the pad mode: the source index (clamped to a valid one) and whether the tap reads data
__device__ __forceinline__ int c1_resolve(int p, int n, int reflect, bool& ok) {
    ok = true;
    if (reflect) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; return p < 0 ? 0 : (p >= n ? n - 1 : p); }
    ok = p >= 0 && p < n;
    return ok ? p : 0;
}

// the MFMA family (vg_c1k3.hip): kernel size KS in {3, 4} with stride ST = KS - 2, C = 16 (KS 3) or 64 (KS 4) output channels
struct C1M {
    const void* x; int x_f32; float sc, sf; int act, pad_mode;
    const float* scale; const float* shift;         // per-sample on-read affine or NULL (then sc, sf)
    const void* noise;                              // bf16 [N][D+2][H+2][W+2] on the reflect-padded grid, added after the activation, or NULL
    int D, H, W, OD, OH, OW, C;
    int td0, th0, tw0;
    const void* w; int Ktot, CK;
    const float* bias; void* out; float* sums;
    const void* dy; float* dw; float* db; float* part;   // part: [workgroups][KS^3 * C] partial slabs (then summed by reduce_partials), or NULL: atomics
    VgFin fin;
};
// VG_OK when the launch was done (or recorded by a dry run), 1 when the shape is not served (the caller continues), < 0 on error
int c1m_fwd(const vg_conv_desc* d, hipStream_t s);
int c1m_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch, int64_t scratch_bytes, hipStream_t s);
