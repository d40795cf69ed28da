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

// Stage the halo tile whose output-tile origin is (od0,oh0,ow0): units of (halo voxel, 8 channels) = 16 B.
template <typename T>
__device__ __forceinline__ void stage_halo(const GatherIn& g, char* halo, const float* scs, int n, int od0, int oh0,
                                           int ow0, int chunk, int tid, int nthreads) {
    const int gpc = g.CK >> 3;
    const int units = g.HD * g.HH * g.HW * gpc;
    const int Ds = g.D >> g.shift0, Hs = g.H >> g.shift0, Ws = g.W >> g.shift0;
    const int ND = g.D + 2 * g.npad, NH = g.H + 2 * g.npad, NW = g.W + 2 * g.npad;
    for (int u = tid; u < units; u += nthreads) {
        const int hv = u / gpc, cg = u - hv * gpc;
        const int hw = hv % g.HW; const int t2 = hv / g.HW;
        const int hh = t2 % g.HH, hd = t2 / g.HH;
        int pd = od0 * g.istr + g.tmin_d + hd, ph = oh0 * g.istr + g.tmin_h + hh, pw = ow0 * g.istr + g.tmin_w + hw;
        const int qd = pd + g.npad, qh = ph + g.npad, qw = pw + g.npad;        // position on the noise grid
        bool valid = resolve_pos(pd, g.D, g.pad_mode);
        valid &= resolve_pos(ph, g.H, g.pad_mode);
        valid &= resolve_pos(pw, g.W, g.pad_mode);
        const int c = chunk * g.CK + cg * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid && c < g.Cin) {
            float x[8];
            int nval = 8;
            if (g.Cin == 1) {
                nval = 1;
                const size_t idx = ((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw;
                x[0] = g.src_f32 ? ((const float*)g.src0)[idx] : bf2f(((const bf16_t*)g.src0)[idx]);
            } else {
                if (c < g.c0) {
                    const size_t idx = (((size_t)(n * Ds + (pd >> g.shift0)) * Hs + (ph >> g.shift0)) * Ws + (pw >> g.shift0)) * g.c0 + c;
                    load8<T>((const T*)g.src0 + idx, x);
                } else {
                    const size_t idx = (((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw) * g.c1 + (c - g.c0);
                    load8<T>((const T*)g.src1 + idx, x);
                }
            }
            const bool has_noise = g.noise && qd >= 0 && qd < ND && qh >= 0 && qh < NH && qw >= 0 && qw < NW;
            const size_t nidx = has_noise ? ((((size_t)(n * ND + qd) * NH + qh) * NW + qw) * g.Cin + c) : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nval) {
                    float y = vg_act(x[j] * scs[cg * 8 + j] + scs[g.CK + cg * 8 + j], g.act);
                    if (has_noise) y += bf2f(g.noise[nidx + j]);
                    v[j] = y;
                }
            }
        }
        store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, v);
    }
}

// Unit table: the (halo voxel, channel group) decomposition of a tile is the same for every tile, so a persistent
// workgroup computes it once (no integer divisions in the per-tile loop).  utab[u] = hd | hh<<8 | hw<<16 | cg<<24.
__device__ __forceinline__ void build_unit_table(const GatherIn& g, int* utab, int tid, int nthreads) {
    const int gpc = g.CK >> 3;
    const int units = g.HD * g.HH * g.HW * gpc;
    for (int u = tid; u < units; u += nthreads) {
        const int hv = u / gpc, cg = u - hv * gpc;
        const int hw = hv % g.HW; const int t2 = hv / g.HW;
        const int hh = t2 % g.HH, hd = t2 / g.HH;
        utab[u] = hd | (hh << 8) | (hw << 16) | (cg << 24);
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

// Table-driven staging, UB units per thread per batch: all global loads of a batch are issued before the first one is
// consumed (the per-unit load->wait->transform chain of the simple loop exposes one HBM latency per unit).
template <typename T, int UB = 4>
__device__ __forceinline__ void stage_halo_tab(const GatherIn& g, char* halo, const float* scs, const int* utab, int n,
                                               int od0, int oh0, int ow0, int chunk, int tid, int nthreads) {
    const int gpc = g.CK >> 3;
    const int units = g.HD * g.HH * g.HW * gpc;
    const int Ds = g.D >> g.shift0, Hs = g.H >> g.shift0, Ws = g.W >> g.shift0;
    const int ND = g.D + 2 * g.npad, NH = g.H + 2 * g.npad, NW = g.W + 2 * g.npad;
    const int bd = od0 * g.istr + g.tmin_d, bh = oh0 * g.istr + g.tmin_h, bw = ow0 * g.istr + g.tmin_w;
    const bool c1mode = g.Cin == 1;
    for (int u0 = tid; u0 < units; u0 += nthreads * UB) {
        Raw8<T> raw[UB];
        float x1[UB];
        int meta[UB];              // LDS element offset (in 8-channel groups) | flags
        bool ok[UB];
        size_t nidx[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int u = u0 + k * nthreads;
            ok[k] = false; meta[k] = -1; nidx[k] = ~(size_t)0; x1[k] = 0.f;
            if (u < units) {
                const int e = utab[u];
                const int hd = e & 255, hh = (e >> 8) & 255, hw = (e >> 16) & 255, cg = e >> 24;
                int pd = bd + hd, ph = bh + hh, pw = bw + hw;
                const int qd = pd + g.npad, qh = ph + g.npad, qw = pw + g.npad;
                bool valid = resolve_pos(pd, g.D, g.pad_mode);
                valid &= resolve_pos(ph, g.H, g.pad_mode);
                valid &= resolve_pos(pw, g.W, g.pad_mode);
                const int c = chunk * g.CK + cg * 8;
                meta[k] = (((hd * g.HH + hh) * g.HW + hw) << 4) | cg;
                ok[k] = valid && c < g.Cin;
                if (ok[k]) {
                    if (c1mode) {
                        const size_t idx = ((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw;
                        x1[k] = g.src_f32 ? ((const float*)g.src0)[idx] : bf2f(((const bf16_t*)g.src0)[idx]);
                    } else if (c < g.c0) {
                        const size_t idx = (((size_t)(n * Ds + (pd >> g.shift0)) * Hs + (ph >> g.shift0)) * Ws + (pw >> g.shift0)) * g.c0 + c;
                        raw_load(raw[k], (const T*)g.src0 + idx);
                    } else {
                        const size_t idx = (((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw) * g.c1 + (c - g.c0);
                        raw_load(raw[k], (const T*)g.src1 + idx);
                    }
                    if (g.noise && qd >= 0 && qd < ND && qh >= 0 && qh < NH && qw >= 0 && qw < NW)
                        nidx[k] = (((size_t)(n * ND + qd) * NH + qh) * NW + qw) * g.Cin + c;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (meta[k] < 0) continue;
            const int cg = meta[k] & 15, hv = meta[k] >> 4;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ok[k]) {
                float x[8];
                if (c1mode) { x[0] = x1[k]; } else raw_unpack(raw[k], x);
                const int nval = c1mode ? 1 : 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < nval) {
                        float y = vg_act(x[j] * scs[cg * 8 + j] + scs[g.CK + cg * 8 + j], g.act);
                        if (nidx[k] != ~(size_t)0) y += bf2f(g.noise[nidx[k] + j]);
                        v[j] = y;
                    }
                }
            }
            store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Staging v3: per-halo-voxel table + one fixed 8-channel group per thread.
//   vtab[hv] = { hd | hh<<8 | hw<<16, element offset into src0 (half-res aware), into src1, into the noise grid }
// Thread t owns channel group cg = t % gpc for the whole chunk (its 8 scale/shift pairs sit in registers) and walks the
// halo voxels vl, vl+vstride, ...  Interior tiles (no padding / reflection inside the halo, no noise, multi-channel
// source) take the fast path: address = tile base + table offset, transform, one 16-byte LDS store.  Everything else
// (border tiles, single-channel sources, noise) takes the general path with the same thread mapping.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void build_voxel_table(const GatherIn& g, int* vtab, int tid, int nthreads) {
    const int nvox = g.HD * g.HH * g.HW;
    const int sh = g.shift0;
    const int Hs = g.H >> sh, Ws = g.W >> sh;
    const int par_d = g.tmin_d & sh, par_h = g.tmin_h & sh, par_w = g.tmin_w & sh;    // parity of the tile base
    for (int hv = tid; hv < nvox; hv += nthreads) {
        const int hw = hv % g.HW; const int t2 = hv / g.HW;
        const int hh = t2 % g.HH, hd = t2 / g.HH;
        vtab[hv * 4] = hd | (hh << 8) | (hw << 16);
        vtab[hv * 4 + 1] = ((((hd + par_d) >> sh) * Hs + ((hh + par_h) >> sh)) * Ws + ((hw + par_w) >> sh)) * g.c0;
        vtab[hv * 4 + 2] = ((hd * g.H + hh) * g.W + hw) * g.c1;
        vtab[hv * 4 + 3] = ((hd * (g.H + 2 * g.npad) + hh) * (g.W + 2 * g.npad) + hw) * g.Cin;
    }
}

template <typename T, bool NOISE, int UB = 4>
__device__ __forceinline__ void stage_halo_v3(const GatherIn& g, char* halo, const float* scs, const int* vtab, int n,
                                              int od0, int oh0, int ow0, int chunk, int tid) {
    const int gpc = g.CK >> 3;
    const int vstride = 256 / gpc;                      // voxels handled in parallel
    if (tid >= vstride * gpc) return;
    const int cg = tid % gpc, vl = tid / gpc;
    const int nvox = g.HD * g.HH * g.HW;
    const int c = chunk * g.CK + cg * 8;
    const int pd0 = od0 * g.istr + g.tmin_d, ph0 = oh0 * g.istr + g.tmin_h, pw0 = ow0 * g.istr + g.tmin_w;
    const bool interior = pd0 >= 0 && ph0 >= 0 && pw0 >= 0 && pd0 + g.HD <= g.D && ph0 + g.HH <= g.H && pw0 + g.HW <= g.W;
    const bool tile_even = ((g.tdl > 0) || !(g.istr & 1)) && ((g.thl > 0) || !(g.istr & 1)) && ((g.twl > 0) || !(g.istr & 1));
    const bool fast = interior && g.Cin != 1 && (g.shift0 == 0 || tile_even);
    const bool plain = !g.in_scale && g.act == VG_ACT_NONE && !NOISE;          // data-gradient operand: pure copy
    T* dst0 = (T*)halo + cg * 8;
    if (fast) {
        if (c >= g.Cin) {                               // channel padding of the last chunk
            const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int hv = vl; hv < nvox; hv += vstride) store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, z);
            return;
        }
        const bool from0 = c < g.c0;
        const int sh = g.shift0;
        const T* base;
        if (from0) base = (const T*)g.src0 + ((((size_t)n * (g.D >> sh) + (pd0 >> sh)) * (g.H >> sh) + (ph0 >> sh)) * (g.W >> sh) + (pw0 >> sh)) * g.c0 + c;
        else base = (const T*)g.src1 + ((((size_t)n * g.D + pd0) * g.H + ph0) * g.W + pw0) * g.c1 + (c - g.c0);
        const int sel = from0 ? 1 : 2;
        // noise lives on the (D+2np)^3 grid: an interior halo never leaves it
        const bf16_t* nbase = NOISE ? g.noise + ((((size_t)n * (g.D + 2 * g.npad) + pd0 + g.npad) * (g.H + 2 * g.npad) + ph0 + g.npad) * (g.W + 2 * g.npad) + pw0 + g.npad) * g.Cin + c : nullptr;
        float sc[8], sf[8];
        if (!plain) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = scs[cg * 8 + j]; sf[j] = scs[g.CK + cg * 8 + j]; }
        }
        for (int hv0 = vl; hv0 < nvox; hv0 += vstride * UB) {
            Raw8<T> raw[UB];
            Raw8<bf16_t> nz[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int hv = hv0 + k * vstride;
                if (hv < nvox) {
                    raw_load(raw[k], base + vtab[hv * 4 + sel]);
                    if (NOISE) raw_load(nz[k], nbase + vtab[hv * 4 + 3]);
                }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int hv = hv0 + k * vstride;
                if (hv < nvox) {
                    T* dst = (T*)(halo + (size_t)hv * g.RS) + cg * 8;
                    if (plain) { *(Raw8<T>*)dst = raw[k]; }
                    else {
                        float x[8];
                        raw_unpack(raw[k], x);
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = vg_act(x[j] * sc[j] + sf[j], g.act);
                        if (NOISE) {
                            float z[8];
                            raw_unpack(nz[k], z);
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] += z[j];
                        }
                        store8<T>(dst, x);
                    }
                }
            }
        }
        return;
    }
    // ---- general path (border tiles, single-channel sources, noise): same thread mapping, one unit at a time (batching
    // these loads as well was measured to cost more in registers/occupancy than it gained) ----
    const int Ds = g.D >> g.shift0, Hs = g.H >> g.shift0, Ws = g.W >> g.shift0;
    const int ND = g.D + 2 * g.npad, NH = g.H + 2 * g.npad, NW = g.W + 2 * g.npad;
    const bool c1mode = g.Cin == 1;
    (void)dst0;
    for (int hv = vl; hv < nvox; hv += vstride) {
        const int e = vtab[hv * 4];
        const int hd = e & 255, hh = (e >> 8) & 255, hw = e >> 16;
        int pd = pd0 + hd, ph = ph0 + hh, pw = pw0 + hw;
        const int qd = pd + g.npad, qh = ph + g.npad, qw = pw + g.npad;
        bool valid = resolve_pos(pd, g.D, g.pad_mode);
        valid &= resolve_pos(ph, g.H, g.pad_mode);
        valid &= resolve_pos(pw, g.W, g.pad_mode);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid && c < g.Cin) {
            float x[8];
            int nval = 8;
            if (c1mode) {
                nval = 1;
                const size_t idx = ((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw;
                x[0] = g.src_f32 ? ((const float*)g.src0)[idx] : bf2f(((const bf16_t*)g.src0)[idx]);
            } else if (c < g.c0) {
                const size_t idx = (((size_t)(n * Ds + (pd >> g.shift0)) * Hs + (ph >> g.shift0)) * Ws + (pw >> g.shift0)) * g.c0 + c;
                load8<T>((const T*)g.src0 + idx, x);
            } else {
                const size_t idx = (((size_t)(n * g.D + pd) * g.H + ph) * g.W + pw) * g.c1 + (c - g.c0);
                load8<T>((const T*)g.src1 + idx, x);
            }
            const bool has_noise = NOISE && g.noise && qd >= 0 && qd < ND && qh >= 0 && qh < NH && qw >= 0 && qw < NW;
            const size_t nidx = has_noise ? ((((size_t)(n * ND + qd) * NH + qh) * NW + qw) * g.Cin + c) : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nval) {
                    float y = vg_act(x[j] * scs[cg * 8 + j] + scs[g.CK + cg * 8 + j], g.act);
                    if (has_noise) y += bf2f(g.noise[nidx + j]);
                    v[j] = y;
                }
            }
        }
        store8<T>((T*)(halo + (size_t)hv * g.RS) + cg * 8, v);
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
