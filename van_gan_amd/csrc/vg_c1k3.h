// vg_c1k3.h -- launch parameters of the single-channel 3x3x3 stem convolution (1 -> C, resunet_model.py:44-60), shared by the VALU
// kernels of vg_pointwise.hip (any C <= 32, both storage types) and the MFMA kernels of vg_c1k3.hip (C = 16, 16-bit storage).
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
// vg_c1k3.hip: VG_OK when the launch was done, 1 when the shape is not served there (C != 16, fp32 storage, switched off)
int c1k3m_fwd(const C1K3& c, int N, bool src_f32, hipStream_t s);
int c1k3m_wgrad(const C1K3& c, int N, bool src_f32, hipStream_t s);
