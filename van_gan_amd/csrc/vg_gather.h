// vg_gather.h -- the input side shared by the forward/data-gradient kernel (vg_conv.hip) and the
// weight-gradient kernel (vg_wgrad.hip): description of the gathered operand and the routine that
// stages one halo tile of it into LDS (normalised, activated, noised, rounded to bf16).
#pragma once
#include "vg_common.h"
#include <stdlib.h>

struct GatherIn {
    const void* src0; const void* src1;
    int c0, c1, shift0, src_f32;
    int N, D, H, W, Cin;
    const float* in_scale; const float* in_shift; int act;
    const bf16_t* noise; int npad;
    int istr, pad_mode, ntaps;
    int8_t td[VG_MAX_TAPS], th[VG_MAX_TAPS], tw[VG_MAX_TAPS];
    int tmin_d, tmin_h, tmin_w, HD, HH, HW, RS, CK;
    int tdl, thl, twl, tiles_d, tiles_h, tiles_w;
    int f32;        // storage type of multi-channel tensors / LDS tile: 0 bf16, 1 f32
    int dbg;        // development ablation flags (VG_DEBUG env): 1 skip halo staging, 2 skip dY staging, 4 skip MFMA
    unsigned long long* stamps;   // diagnostic build only (vg_set_stamp_buffer): s_memtime stamps per phase, else NULL
};

extern unsigned long long* g_vg_stamps;

__device__ __forceinline__ bool resolve_pos(int& p, int n, int mode) {
    if (mode == VG_PAD_REFLECT) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
        p = p < 0 ? 0 : (p >= n ? n - 1 : p);   // tile overhang only (those outputs are masked)
        return true;
    }
    return p >= 0 && p < n;
}

// scs: LDS floats [2*CK] (scale then shift) for channels chunk*CK .. +CK of sample n.
__device__ __forceinline__ void stage_scale_shift(const GatherIn& g, float* scs, int n, int chunk, int tid) {
    if (tid < g.CK) {
        const int c = chunk * g.CK + tid;
        float sc = 1.f, sh = 0.f;
        if (g.in_scale && c < g.Cin) { sc = g.in_scale[n * g.Cin + c]; sh = g.in_shift[n * g.Cin + c]; }
        scs[tid] = sc; scs[g.CK + tid] = sh;
    }
}

// raw 8-channel vector as loaded from global memory
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { bf16x8 v; };
template <> struct Raw8<float> { f32x4 a, b; };
__device__ __forceinline__ void raw_load(Raw8<bf16_t>& r, const bf16_t* p) { r.v = *(const bf16x8*)p; }
__device__ __forceinline__ void raw_load(Raw8<float>& r, const float* p) { r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 4); }
__device__ __forceinline__ void raw_unpack(const Raw8<bf16_t>& r, float* o) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)r.v[j]);
}
__device__ __forceinline__ void raw_unpack(const Raw8<float>& r, float* o) {
    o[0] = r.a[0]; o[1] = r.a[1]; o[2] = r.a[2]; o[3] = r.a[3]; o[4] = r.b[0]; o[5] = r.b[1]; o[6] = r.b[2]; o[7] = r.b[3];
}


// ------------------------------------------------------------------------------------------------------------------
// Halo staging: ONE path for interior and border tiles.  Padding/reflection is separable per axis, so each tile first
// resolves its HD+HH+HW halo coordinates into three tiny LDS tables (resolved source coordinate, or -1 = zero fill);
// every unit then computes its address with a few integer ops and no branches, and the global loads of UB units are
// issued before the first is consumed -- border tiles (the majority on 32^3 and smaller grids) no longer serialise one
// load per unit.  rtab: LDS ints [HD + HH + HW]; must be filled (stage_resolve_axes) and synchronised before use.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_resolve_axes(const GatherIn& g, int* rtab, int od0, int oh0, int ow0, int tid) {
    const int pd0 = od0 * g.istr + g.tmin_d, ph0 = oh0 * g.istr + g.tmin_h, pw0 = ow0 * g.istr + g.tmin_w;
    for (int i = tid; i < g.HD + g.HH + g.HW; i += 256) {       // HD alone can exceed the workgroup on thin deep tiles
        int p, ext;
        if (i < g.HD) { p = pd0 + i; ext = g.D; }
        else if (i < g.HD + g.HH) { p = ph0 + i - g.HD; ext = g.H; }
        else { p = pw0 + i - g.HD - g.HH; ext = g.W; }
        rtab[i] = resolve_pos(p, ext, g.pad_mode) ? p : -1;
    }
}

template <typename T, bool NOISE, int UB = 4>
__device__ __forceinline__ void stage_halo_tile(const GatherIn& g, char* halo, const float* scs, const int* vtab, const int* rtab,
                                              int n, int od0, int oh0, int ow0, int chunk, int tid) {
    const int gpc = g.CK >> 3;
    const int vstride = 256 / gpc;
    if (tid >= vstride * gpc) return;
    const int cg = tid % gpc, vl = tid / gpc;
    const int nvox = g.HD * g.HH * g.HW;
    const int c = chunk * g.CK + cg * 8;
    const int ND = g.D + 2 * g.npad, NH = g.H + 2 * g.npad, NW = g.W + 2 * g.npad;
    const int qd0 = od0 * g.istr + g.tmin_d + g.npad, qh0 = oh0 * g.istr + g.tmin_h + g.npad, qw0 = ow0 * g.istr + g.tmin_w + g.npad;
    if (c >= g.Cin) {                                   // channel padding of the last chunk / second group of a 1-channel source
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int hv = vl; hv < nvox; hv += vstride) store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, z);
        return;
    }
    if (g.Cin == 1) {                                   // single-channel source (fp32 or bf16 volume): one scalar per voxel
        const float sc0 = scs[0], sf0 = scs[g.CK];
        const size_t nb = (size_t)n * g.D * g.H * g.W;
        for (int hv = vl; hv < nvox; hv += vstride) {
            const int e = vtab[hv];
            const int hd = e & 2047, hh = (e >> 11) & 2047, hw = e >> 22;
            const int rd = rtab[hd], rh = rtab[g.HD + hh], rw = rtab[g.HD + g.HH + hw];
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((rd | rh | rw) >= 0) {
                const size_t idx = nb + ((size_t)rd * g.H + rh) * g.W + rw;
                const float x = g.src_f32 ? ((const float*)g.src0)[idx] : bf2f(((const bf16_t*)g.src0)[idx]);
                float y = vg_act(x * sc0 + sf0, g.act);
                if (NOISE) {
                    const int qd = qd0 + hd, qh = qh0 + hh, qw = qw0 + hw;
                    if (g.noise && qd >= 0 && qd < ND && qh >= 0 && qh < NH && qw >= 0 && qw < NW)
                        y += bf2f(g.noise[((size_t)(n * ND + qd) * NH + qh) * NW + qw]);
                }
                v[0] = y;
            }
            store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, v);
        }
        return;
    }
    const bool plain = !g.in_scale && g.act == VG_ACT_NONE && !NOISE;     // data-gradient operand: pure copy
    const int sh = g.shift0;
    const bool from0 = c < g.c0;
    const int Hs = from0 ? (g.H >> sh) : g.H, Ws = from0 ? (g.W >> sh) : g.W, Dsz = from0 ? (g.D >> sh) : g.D;
    const int cs = from0 ? g.c0 : g.c1;
    const int ssh = from0 ? sh : 0;
    const T* base = from0 ? (const T*)g.src0 + (size_t)n * Dsz * Hs * Ws * cs + c
                          : (const T*)g.src1 + (size_t)n * Dsz * Hs * Ws * cs + (c - g.c0);
    float sc[8], sf[8];
    if (!plain) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = scs[cg * 8 + j]; sf[j] = scs[g.CK + cg * 8 + j]; }
    }
    for (int hv0 = vl; hv0 < nvox; hv0 += vstride * UB) {
        Raw8<T> raw[UB];
        Raw8<bf16_t> nz[NOISE ? UB : 1];
        int st[UB];                 // -1: beyond the tile, 0: zero fill, 1: data, 3: data + noise
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int hv = hv0 + k * vstride;
            st[k] = -1;
            if (hv < nvox) {
                const int e = vtab[hv];
                const int hd = e & 2047, hh = (e >> 11) & 2047, hw = e >> 22;
                const int rd = rtab[hd], rh = rtab[g.HD + hh], rw = rtab[g.HD + g.HH + hw];
                st[k] = ((rd | rh | rw) >= 0) ? 1 : 0;
                if (st[k]) {
                    const int idx = (((rd >> ssh) * Hs + (rh >> ssh)) * Ws + (rw >> ssh)) * cs;      // < 2^31 elements per sample
                    raw_load(raw[k], base + idx);
                    if (NOISE) {
                        const int qd = qd0 + hd, qh = qh0 + hh, qw = qw0 + hw;
                        if (g.noise && qd >= 0 && qd < ND && qh >= 0 && qh < NH && qw >= 0 && qw < NW) {
                            raw_load(nz[k], g.noise + (((size_t)(n * ND + qd) * NH + qh) * NW + qw) * g.Cin + c);
                            st[k] = 3;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (st[k] < 0) continue;
            const int hv = hv0 + k * vstride;
            T* dst = (T*)(halo + (size_t)hv * g.RS) + cg * 8;
            if (plain && st[k] > 0) { *(Raw8<T>*)dst = raw[k]; continue; }
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (st[k] > 0) {
                float x[8], z[8];
                raw_unpack(raw[k], x);
                if (NOISE && st[k] == 3) raw_unpack(nz[k], z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = vg_act(x[j] * sc[j] + sf[j], g.act);
                    if (NOISE && st[k] == 3) y += z[j];
                    v[j] = y;
                }
            }
            store8<T>(dst, v);
        }
    }
}

__device__ __forceinline__ void build_voxel_table(const GatherIn& g, int* vtab, int tid, int nthreads) {
    const int nvox = g.HD * g.HH * g.HW;
    for (int hv = tid; hv < nvox; hv += nthreads) {
        const int hw = hv % g.HW; const int t2 = hv / g.HW;
        vtab[hv] = (t2 / g.HH) | ((t2 % g.HH) << 11) | (hw << 22);
    }
}

// ---- host: validate the input side of a descriptor and derive the tile geometry for BM voxels ----
static inline int fill_gather(const vg_conv_desc* d, GatherIn& g, int CK, int BM) {
    if (!d || !d->src0) return VG_EINVAL;
    g.f32 = d->f32 ? 1 : 0;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("VG_DEBUG"); dbg = e ? atoi(e) : 0; } g.dbg = dbg; }
    g.stamps = g_vg_stamps;
    const int Cin = d->c_src0 + d->c_src1;
    if (Cin < 1 || d->ntaps < 1 || d->ntaps > VG_MAX_TAPS) return VG_EINVAL;
    if (Cin != 1 && ((d->c_src0 % 8) || (d->c_src1 % 8))) return VG_EINVAL;
    if (Cin == 1 && (d->src1 || d->src0_shift)) return VG_EINVAL;
    if (d->src_f32 && Cin != 1) return VG_EINVAL;
    if (d->c_src1 > 0 && !d->src1) return VG_EINVAL;
    if (CK < 16 || (CK % 16) || CK > 128) return VG_EINVAL;
    if (d->istr < 1 || d->istr > 2) return VG_EINVAL;
    if (d->src0_shift && ((d->D | d->H | d->W) & 1)) return VG_EINVAL;
    if (d->pad_mode == VG_PAD_REFLECT && (d->D < 2 || d->H < 2 || d->W < 2)) return VG_EINVAL;
    if (d->N < 1 || d->OD < 1 || d->OH < 1 || d->OW < 1) return VG_EINVAL;
    g.src0 = d->src0; g.src1 = d->src1; g.c0 = d->c_src0; g.c1 = d->c_src1; g.shift0 = d->src0_shift ? 1 : 0;
    g.src_f32 = d->src_f32; g.N = d->N; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = Cin;
    g.in_scale = d->in_scale; g.in_shift = d->in_shift; g.act = d->act;
    g.noise = (const bf16_t*)d->noise; g.npad = d->noise ? d->noise_pad : 0;
    g.istr = d->istr; g.pad_mode = d->pad_mode; g.ntaps = d->ntaps; g.CK = CK;
    int mn[3] = {127, 127, 127}, mx[3] = {-128, -128, -128};
    for (int i = 0; i < d->ntaps; ++i) {
        g.td[i] = d->tap_d[i]; g.th[i] = d->tap_h[i]; g.tw[i] = d->tap_w[i];
        const int v[3] = {d->tap_d[i], d->tap_h[i], d->tap_w[i]};
        for (int a = 0; a < 3; ++a) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
    }
    g.tmin_d = mn[0]; g.tmin_h = mn[1]; g.tmin_w = mn[2];
    g.RS = CK * (d->f32 ? 4 : 2) + 16;
    // tile shape: powers of two with product BM that minimise the halo volume (staging work and L2 traffic scale with
    // it); the innermost extent stays >= 8 voxels where the grid allows so that rows remain long contiguous runs
    int TW = 1, TH = 1, TD = 1;
    {
        const int ex[3] = {mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1};
        const int capd = pow2_ceil(d->OD), caph = pow2_ceil(d->OH), capw = pow2_ceil(d->OW) < 16 ? pow2_ceil(d->OW) : 16;
        long best = -1;
        for (int tw = 1; tw <= capw && tw <= BM; tw <<= 1) {
            if (tw < 8 && tw < capw) continue;
            for (int th = 1; th <= caph && tw * th <= BM; th <<= 1) {
                const int td = BM / (tw * th);
                long vol = (long)((td - 1) * d->istr + ex[0]) * ((th - 1) * d->istr + ex[1]) * ((tw - 1) * d->istr + ex[2]);
                if (td > capd) vol *= 4;          // overhang in D wastes whole planes: only when nothing else fits
                if (best < 0 || vol < best || (vol == best && tw > TW)) { best = vol; TW = tw; TH = th; TD = td; }
            }
        }
    }
    g.twl = ilog2_exact(TW); g.thl = ilog2_exact(TH); g.tdl = ilog2_exact(TD);
    g.tiles_w = (d->OW + TW - 1) / TW; g.tiles_h = (d->OH + TH - 1) / TH; g.tiles_d = (d->OD + TD - 1) / TD;
    g.HD = (TD - 1) * d->istr + (mx[0] - mn[0]) + 1;
    g.HH = (TH - 1) * d->istr + (mx[1] - mn[1]) + 1;
    g.HW = (TW - 1) * d->istr + (mx[2] - mn[2]) + 1;
    return VG_OK;
}
static inline int halo_bytes(const GatherIn& g) { return g.HD * g.HH * g.HW * g.RS; }
