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
    int tmin_d, tmin_h, tmin_w, HD, HH, HW, CK;
    int HHp, HWp, US, PSB;   // LDS halo image, pitches HHp/HWp in voxels, unit = 8 channels = US bytes.  Two forms, both
    int DS;                  // bytes of one D-slice of the image (HHp*HWp*VS, rounded to 64 units for LDS-DMA staging)
    int VS, CS, planar;      // addressed as voxel * VS + channel_group * CS:  planar [group][HD][HHp][HWp] (VS = US,
                             // CS = PSB: bank-conflict-free fragment reads, voxel-fastest staging) for thin chunks, and
                             // row-major [HD][HHp][HWp][CK + pad] (VS = row bytes, CS = US: full-line global loads) for wide ones
    int HWh;                 // istr 2: W positions are stored de-interleaved (even columns, then odd columns from HWh on) so
                             // that the stride-2 voxels of a sub-tile are consecutive units; 0 for istr 1
    int tdl, thl, twl, tiles_d, tiles_h, tiles_w;
    int f32;        // storage type of multi-channel tensors / LDS tile: 0 bf16, 1 f32
    // W-packed single-channel source (Cin == 1, k > 1): the kernel sees the pseudo-input P[d][h][ow][j] = x[d][h][ow*istr + wmin + j],
    // j < wpack = k, i.e. a k-channel source convolved with the k*k (d, h) taps only -- K shrinks from 16 padded channels
    // per tap to 16 per (d, h) tap PAIR OF k taps, and the DHWIO kernel [k][k][k][1][Cout] is, unchanged in memory, the
    // [k*k taps][k channels][Cout] kernel of that pseudo-convolution.  The halo image has no W halo (HW = tile width, unit
    // W step for either stride); HWx is the true input W extent of a tile (axis-table length), Cw the weight-side Cin.
    int wpack, HWx, wmin, Cw;
    int lean;       // >= 0: lean staging mode (VG_STAGE_*) of a multi-channel bf16 source; -1: original column staging
    int dbg;        // development ablation flags (VG_DEBUG env, -DVG_ABLATE builds only; else 0): 1 skip halo staging, 2 skip dY staging,
                    // 4 skip MFMA, 64 skip the epilogue's vector store
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

// ------------------------------------------------------------------------------------------------------------------
// Halo staging by COLUMNS.  A column is one (hh, hw, 8-channel group) of the halo tile; a thread owns whole columns and
// walks them along D.  Everything per-lane is resolved once per column (H/W reflection or zero padding, source of the
// virtual concat, upsample shift, noise position: separable per axis, kept in small LDS axis tables); the D axis is
// wave-uniform and costs scalar instructions only.  The loop body is straight-line: addresses of invalid units are
// clamped and their data zeroed afterwards, so the loads of UB units are really in flight together (a conditional load
// makes the compiler drain vmcnt before the next one) and the per-unit VALU cost is the transform itself.
//   ctab: LDS ints [2 * ncols]: {hh | hw<<10 | cg<<20, byte offset of the column inside one halo D-plane}; built once.
//   rtab: LDS ints [3][HH + HW]: element offsets (or -1) along H and W for src0, src1 and the noise tensor; per tile.
// ------------------------------------------------------------------------------------------------------------------
// storage column of halo column hw
__host__ __device__ __forceinline__ int halo_pos_w(const GatherIn& g, int hw) { return g.HWh ? (hw >> 1) + (hw & 1) * g.HWh : hw; }
__host__ __device__ __forceinline__ int stage_ncols(const GatherIn& g) { return g.HH * g.HW * (g.Cin == 1 ? 1 : (g.CK >> 3)); }
// length of the per-tile axis tables (H entries, then W entries)
// (C1 = 0: the kernel variant never sees a single-channel source -- the W-packed form is compiled out; the hot multi-channel
// kernels are instruction-issue bound and every extra scalar field they touch shows up in their run time)
template <int C1 = 2>
__host__ __device__ __forceinline__ int stage_axis_len(const GatherIn& g) { return g.HH + ((C1 != 0 && C1 != 3 && g.wpack) ? g.HWx : g.HW); }

template <int NT = 256>
__device__ __forceinline__ void build_column_table(const GatherIn& g, int* ctab, int tid) {
    const int gpc = g.Cin == 1 ? 1 : (g.CK >> 3);
    const int ncols = g.HH * g.HW * gpc;
    const int nv = g.HH * g.HW;
    for (int col = tid; col < ncols; col += NT) {
        // planar: voxel-fastest, 8 consecutive lanes write 8 consecutive units of one plane (conflict-free ds_write_b128);
        // row-major: channel-group-fastest, consecutive lanes read one voxel's contiguous channels
        const int cg = g.planar ? col / nv : col % gpc, v = g.planar ? col - cg * nv : col / gpc;
        const int hh = v / g.HW, hw = v - hh * g.HW;
        ctab[2 * col] = hh | (hw << 10) | (cg << 20);
        ctab[2 * col + 1] = cg * g.CS + (hh * g.HWp + halo_pos_w(g, hw)) * g.VS;
    }
}

template <int C1 = 2>
__device__ __forceinline__ void stage_resolve_axes(const GatherIn& g, int* rtab, int oh0, int ow0, int tid) {
    const int L = stage_axis_len<C1>(g);
    const int ph0 = oh0 * g.istr + g.tmin_h, pw0 = ow0 * g.istr + ((C1 != 0 && C1 != 3 && g.wpack) ? g.wmin : g.tmin_w);
    const int cs0 = g.Cin == 1 ? 1 : g.c0;
    // wave s resolves table s (src0, src1, noise): the three small jobs run side by side, off wave 0's critical path
    const int set = tid >> 6;
    if (set < 3 && !(set == 1 && g.c1 == 0) && !(set == 2 && !g.noise)) {
        for (int j = tid & 63; j < L; j += 64) {
            const bool isH = j < g.HH;
            int p = isH ? ph0 + j : pw0 + j - g.HH;
            const int q = p + g.npad;
            bool valid = resolve_pos(p, isH ? g.H : g.W, g.pad_mode);
            int off;
            if (set == 0) { const int ps = p >> g.shift0; off = isH ? ps * (g.W >> g.shift0) * cs0 : ps * cs0; }
            else if (set == 1) off = isH ? p * g.W * g.c1 : p * g.c1;
            else {
                const int next = (isH ? g.H : g.W) + 2 * g.npad;
                valid = valid && q >= 0 && q < next;
                off = isH ? q * (g.W + 2 * g.npad) * g.Cin : q * g.Cin;
            }
            rtab[set * L + j] = valid ? off : -1;
        }
    }
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// y = act(x * sc + sf) on 8 channels, branch-free: act(y) = max(y, slope * y) with slope 1 (none), 0 (ReLU), 0.2 (leaky)
__device__ __forceinline__ void stage_affine_act(float* x, const f32x2* sc, const f32x2* sf, float slope) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 v = {x[2 * j], x[2 * j + 1]};
        v = v * sc[j] + sf[j];
        const f32x2 t = v * slope;
        x[2 * j] = fmaxf(v[0], t[0]); x[2 * j + 1] = fmaxf(v[1], t[1]);
    }
}

// Work split of a tile: few columns (thin tiles) are additionally split into D segments so that all 256 threads carry
// loads; otherwise threads take whole columns round-robin.  Returns this thread's first column, column stride, and D range.
struct StageSplit { int col0, cstride, hd_lo, hd_hi; };
template <int NT = 256>
__device__ __forceinline__ StageSplit stage_split(int ncols, int HD, int tid) {
    StageSplit sp = {tid, NT, 0, HD};
    if (ncols <= NT / 2) {
        int nseg = NT / ncols; if (nseg > HD) nseg = HD;
        const int seglen = (HD + nseg - 1) / nseg;
        const int seg = tid / ncols;
        sp.col0 = seg < nseg ? tid - seg * ncols : ncols;         // surplus threads idle
        sp.hd_lo = seg * seglen; sp.hd_hi = min(HD, sp.hd_lo + seglen);
    }
    return sp;
}

// single-channel source of element type S (the fp32 input volumes, bf16 logits/gradients)
template <typename T, typename S, bool NOISE, int UB, bool WP = true>
__device__ __forceinline__ void stage_halo_c1(const GatherIn& g, char* halo, const float* scs, const int* ctab, const int* rtab,
                                              int n, int pd0, int tid) {
    const int L = stage_axis_len<(WP ? 2 : 0)>(g);
    const int ncols = g.HH * g.HW;
    const int plane = g.DS;
    const float slope = g.act == VG_ACT_RELU ? 0.f : (g.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    const int ND = g.D + 2 * g.npad;
    const int nplane = (g.H + 2 * g.npad) * (g.W + 2 * g.npad);
    const float sc0 = scs[0], sf0 = scs[g.CK];
    const int splane = g.H * g.W;
    const S* sbase = (const S*)g.src0 + (size_t)n * g.D * splane;
    const bf16_t* nbase = NOISE ? g.noise + (size_t)n * ND * nplane : nullptr;
    const int ngrp = g.CK >> 3;
    const StageSplit sp = stage_split(ncols, g.HD, tid);
    if (WP && g.wpack) {
        // W-packed: channel j of halo voxel (hd, hh, hw) is the source at W position hw*istr + j of the tile's input window
        for (int col = sp.col0; col < ncols; col += sp.cstride) {
            const int e = ctab[2 * col], hoff = ctab[2 * col + 1];
            const int hh = e & 1023, hw = (e >> 10) & 1023;
            const int oh = rtab[hh];
            int ow[8], nw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int wi = min(hw * g.istr + j, g.HWx - 1);
                ow[j] = j < g.wpack ? rtab[g.HH + wi] : -1;
                nw[j] = (NOISE && j < g.wpack) ? rtab[2 * L + g.HH + wi] : -1;
            }
            const int nh = NOISE ? rtab[2 * L + hh] : -1;
            for (int hd = sp.hd_lo; hd < sp.hd_hi; ++hd) {
                int rd = pd0 + hd;
                const int qd = rd + g.npad;
                const bool dvalid = resolve_pos(rd, g.D, g.pad_mode);
                const bool nd = NOISE && dvalid && qd >= 0 && qd < ND && nh >= 0;
                const S* srow = sbase + (dvalid ? rd * splane : 0) + (oh >= 0 ? oh : 0);
                const bf16_t* nrow = NOISE ? nbase + (nd ? qd * nplane + nh : 0) : nullptr;
                float xv[8], zv[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {                      // straight-line clamped loads, masked afterwards
                    xv[j] = ld_global(srow + (ow[j] >= 0 ? ow[j] : 0));
                    zv[j] = NOISE ? ld_global(nrow + (nw[j] >= 0 ? nw[j] : 0)) : 0.f;
                }
#pragma unroll
                for (int j = 4; j < 8; ++j) { xv[j] = 0.f; zv[j] = 0.f; }
                if (g.wpack > 4) {
#pragma unroll
                    for (int j = 4; j < 8; ++j) {
                        xv[j] = ld_global(srow + (ow[j] >= 0 ? ow[j] : 0));
                        zv[j] = NOISE ? ld_global(nrow + (nw[j] >= 0 ? nw[j] : 0)) : 0.f;
                    }
                }
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = xv[j] * sc0 + sf0;
                    y = fmaxf(y, y * slope) + ((nd && nw[j] >= 0) ? zv[j] : 0.f);
                    v[j] = (dvalid && oh >= 0 && ow[j] >= 0) ? y : 0.f;
                }
                const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                T* row = (T*)(halo + (size_t)hd * plane + hoff);
                store8<T>(row, v);
                for (int b = 1; b < ngrp; ++b) store8<T>((T*)((char*)row + (size_t)b * g.CS), z8);
            }
        }
        return;
    }
    for (int col = sp.col0; col < ncols; col += sp.cstride) {
        const int e = ctab[2 * col], hoff = ctab[2 * col + 1];
        const int hh = e & 1023, hw = (e >> 10) & 1023;
        const int oh = rtab[hh], ow = rtab[g.HH + hw];
        const bool cvalid = (oh | ow) >= 0;
        const int coff = cvalid ? oh + ow : 0;
        int noff = 0; bool nvalid = false;
        if (NOISE) { const int nh = rtab[2 * L + hh], nw = rtab[2 * L + g.HH + hw]; nvalid = (nh | nw) >= 0; noff = nvalid ? nh + nw : 0; }
        for (int hd0 = sp.hd_lo; hd0 < sp.hd_hi; hd0 += UB) {
            float xv[UB], zv[UB]; bool ok[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int hd = hd0 + k < sp.hd_hi ? hd0 + k : sp.hd_hi - 1;
                int rd = pd0 + hd;
                const int qd = rd + g.npad;
                const bool dvalid = resolve_pos(rd, g.D, g.pad_mode);
                xv[k] = ld_global(sbase + (dvalid ? rd * splane : 0) + coff);
                ok[k] = dvalid && cvalid;
                zv[k] = 0.f;
                if (NOISE) {
                    const bool nd = dvalid && qd >= 0 && qd < ND;
                    const float z = ld_global(nbase + (nd ? qd * nplane : 0) + noff);
                    zv[k] = (nd && nvalid) ? z : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                if (hd0 + k < sp.hd_hi) {
                    float y = xv[k] * sc0 + sf0;
                    y = fmaxf(y, y * slope) + zv[k];
                    const float v[8] = {ok[k] ? y : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    T* row = (T*)(halo + (size_t)(hd0 + k) * plane + hoff);
                    store8<T>(row, v);
                    for (int b = 1; b < ngrp; ++b) store8<T>((T*)((char*)row + (size_t)b * g.CS), z8);
                }
            }
        }
    }
}

// C1: 0 = the source is never single-channel (that path is compiled out), 1 = always single-channel, 2 = decided at run
// time (weight-gradient kernel), 3 = decided at run time, W-packed form not supported.  The multi-channel conv_kernel
// variants use 3 although they never see a single-channel source: they sit at their register cap, and both compiling the
// path out (0) and the W-packed version of it (2) made the allocator spill 4 more VGPRs in the 8-sub-tile variants
// (+17 % run time on the 16->16 layers at 128^3; -Rpass-analysis=kernel-resource-usage: 10 -> 14 spilled VGPRs).
template <typename T, bool NOISE, int UB = 4, int C1 = 2>
__device__ __forceinline__ void stage_halo_tile(const GatherIn& g, char* halo, const float* scs, const int* ctab, const int* rtab,
                                                int n, int od0, int chunk, int tid) {
    const int L = stage_axis_len<C1>(g);
    const int ncols = stage_ncols(g);
    const int plane = g.DS;                                    // bytes of one D-slice of the halo image
    const int pd0 = od0 * g.istr + g.tmin_d;
    const float slope = g.act == VG_ACT_RELU ? 0.f : (g.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    const bool zero_mode = g.pad_mode != VG_PAD_REFLECT;
    const int ND = g.D + 2 * g.npad;
    const int nplane = (g.H + 2 * g.npad) * (g.W + 2 * g.npad) * g.Cin;       // noise elements per D-plane

    if constexpr (C1 != 0) {
        if (C1 == 1 || g.Cin == 1) {
            // single-channel source (fp32 or bf16 volume): one scalar per voxel -> channel 0 of an otherwise zero row
            if (g.src_f32) stage_halo_c1<T, float, NOISE, UB, C1 != 3>(g, halo, scs, ctab, rtab, n, pd0, tid);
            else stage_halo_c1<T, bf16_t, NOISE, UB, C1 != 3>(g, halo, scs, ctab, rtab, n, pd0, tid);
            return;
        }
    }

    const bool plain = !g.in_scale && g.act == VG_ACT_NONE && !NOISE;     // data-gradient operand: pure copy
    const int sh = g.shift0;
    const int dpl0 = (g.H >> sh) * (g.W >> sh) * g.c0, dpl1 = g.H * g.W * g.c1;      // source elements per D-plane
    const T* b0 = (const T*)g.src0 + (size_t)n * (g.D >> sh) * dpl0;
    const T* b1 = (const T*)g.src1 + (size_t)n * g.D * dpl1;
    const bf16_t* nb = NOISE ? g.noise + (size_t)n * ND * nplane : nullptr;
    const StageSplit sp = stage_split(ncols, g.HD, tid);
    for (int col = sp.col0; col < ncols; col += sp.cstride) {
        const int e = ctab[2 * col], hoff = ctab[2 * col + 1];
        const int hh = e & 1023, hw = (e >> 10) & 1023, cg = e >> 20;
        const int c = chunk * g.CK + cg * 8;
        const bool from0 = c < g.c0;
        const int set = from0 ? 0 : L;
        const int oh = rtab[set + hh], ow = rtab[set + g.HH + hw];
        const bool cvalid = c < g.Cin && (oh | ow) >= 0;
        const T* pc = cvalid ? (from0 ? b0 + c : b1 + (c - g.c0)) + (oh + ow) : b0;      // invalid columns read a dummy, then zero
        const bf16_t* pn = nullptr; bool nvalid = false;
        if (NOISE) {
            const int nh = rtab[2 * L + hh], nw = rtab[2 * L + g.HH + hw];
            nvalid = cvalid && (nh | nw) >= 0;
            pn = nb + (nvalid ? nh + nw + c : 0);
        }
        f32x2 sc[4], sf[4];
        if (!plain) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[j] = (f32x2){scs[cg * 8 + 2 * j], scs[cg * 8 + 2 * j + 1]};
                sf[j] = (f32x2){scs[g.CK + cg * 8 + 2 * j], scs[g.CK + cg * 8 + 2 * j + 1]};
            }
        }
        for (int hd0 = sp.hd_lo; hd0 < sp.hd_hi; hd0 += UB) {
            Raw8<T> raw[UB];
            Raw8<bf16_t> nz[NOISE ? UB : 1];
            bool ok[UB], nok[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int hd = hd0 + k < sp.hd_hi ? hd0 + k : sp.hd_hi - 1;
                int rd = pd0 + hd;
                const int qd = rd + g.npad;
                const bool dvalid = resolve_pos(rd, g.D, g.pad_mode);          // wave-uniform
                const int s0 = dvalid ? (rd >> sh) * dpl0 : 0, s1 = dvalid ? rd * dpl1 : 0;
                raw_load(raw[k], pc + (from0 ? s0 : s1));
                ok[k] = dvalid && cvalid;
                nok[k] = false;
                if (NOISE) {
                    const bool nd = dvalid && qd >= 0 && qd < ND;
                    raw_load(nz[k], pn + (nd ? qd * nplane : 0));
                    nok[k] = nd && nvalid;
                }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                if (hd0 + k < sp.hd_hi) {
                    T* dst = (T*)(halo + (size_t)(hd0 + k) * plane + hoff);
                    if (plain) {
                        Raw8<T> r = raw[k];
                        if (zero_mode || g.Cin % g.CK) raw_mask(r, ok[k]);
                        *(Raw8<T>*)dst = r;
                    } else {
                        float x[8];
                        raw_unpack(raw[k], x);
                        stage_affine_act(x, sc, sf, slope);
                        if (NOISE) {
                            float z[8];
                            raw_unpack(nz[k], z);
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] += nok[k] ? z[j] : 0.f;
                        }
                        if (zero_mode || g.Cin % g.CK) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = ok[k] ? x[j] : 0.f;
                        }
                        store8<T>(dst, x);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LEAN staging (multi-channel sources; used by the producer/consumer kernels).  Same column walk as stage_halo_tile, but
//   * the D axis has a per-tile table too: rt[set][HH + HW + HD + VG_DPAD] holds BYTE offsets along H, W and D (or -1
//     where the position is zero padding), so a unit's address is  column pointer + one LDS word  -- the first version
//     resolved reflection / padding and multiplied out the plane offset per unit in vector ALU code (~35 of its ~70
//     instructions per unit; the staging waves are VALU-issue bound, in-kernel stamps: 9.8 k cycles per 2000-unit tile);
//     the D entries are repeated VG_DPAD times past the end so that a batch reads dt[hd0 + k] with immediate offsets;
//   * the on-read transform is a template parameter (plain copy / affine + ReLU / affine + LeakyReLU [+ noise]): no
//     run-time slope, no dead masking code in reflect mode (MASK = zero padding or a ragged last channel chunk).
// ------------------------------------------------------------------------------------------------------------------
#define VG_DPAD 16
#define VG_STAGE_PLAIN 0      // no scale/shift, no activation (data-gradient operand, raw shortcut sources)
#define VG_STAGE_RELU 1       // act(x * scale + shift), ReLU
#define VG_STAGE_LRELU 2      // ... LeakyReLU(0.2)
#define VG_STAGE_LRELU_NOISE 3
#define VG_STAGE_LRELU_M 4    // the same two with masking (zero padding: D.down2 / D.out)
#define VG_STAGE_LRELU_NOISE_M 5
#define VG_STAGE_GENERIC 6    // run-time everything (scale/shift optional, any activation, noise if present, masked): rare combinations

__host__ __device__ __forceinline__ int stage_axis_len3(const GatherIn& g) { return g.HH + g.HW + g.HD + VG_DPAD; }

__device__ __forceinline__ void stage_resolve_axes3(const GatherIn& g, int* rt, int od0, int oh0, int ow0, int tid) {
    const int L = stage_axis_len3(g);
    const int esz = g.f32 ? 4 : 2;
    const int pd0 = od0 * g.istr + g.tmin_d, ph0 = oh0 * g.istr + g.tmin_h, pw0 = ow0 * g.istr + g.tmin_w;
    const int sh = g.shift0;
    const int set = tid >> 6;            // wave s resolves table s (src0, src1, noise)
    if (set < 3 && !(set == 1 && g.c1 == 0) && !(set == 2 && !g.noise)) {
        for (int j = tid & 63; j < L; j += 64) {
            const int axis = j < g.HH ? 1 : (j < g.HH + g.HW ? 2 : 0);                    // 1 H, 2 W, 0 D
            int jj = axis == 1 ? j : (axis == 2 ? j - g.HH : j - g.HH - g.HW);
            if (axis == 0 && jj >= g.HD) jj = g.HD - 1;                                 // padding entries repeat the last plane
            const int n_ax = axis == 1 ? g.H : (axis == 2 ? g.W : g.D);
            int p = (axis == 1 ? ph0 : (axis == 2 ? pw0 : pd0)) + jj;
            const int q = p + g.npad;
            bool valid = resolve_pos(p, n_ax, g.pad_mode);
            long off;
            if (set == 0) {
                const int ps = p >> sh, Ws = g.W >> sh, Hs = g.H >> sh;
                off = axis == 1 ? (long)ps * Ws * g.c0 : (axis == 2 ? (long)ps * g.c0 : (long)ps * Hs * Ws * g.c0);
                off *= esz;
            } else if (set == 1) {
                off = axis == 1 ? (long)p * g.W * g.c1 : (axis == 2 ? (long)p * g.c1 : (long)p * g.H * g.W * g.c1);
                off *= esz;
            } else {
                const int NW = g.W + 2 * g.npad, NH = g.H + 2 * g.npad;
                valid = valid && q >= 0 && q < n_ax + 2 * g.npad;
                off = axis == 1 ? (long)q * NW * g.Cin : (axis == 2 ? (long)q * g.Cin : (long)q * NH * NW * g.Cin);
                off *= 2;
            }
            rt[set * L + j] = valid ? (int)off : -1;
        }
    }
}

// MODE >= 0: compile-time transform (producer/consumer kernels); MODE == -1: taken from g.lean at run time (wave-uniform
// branches per unit: the scalar unit has slack, the vector ALU does not) -- NOISE then says whether the kernel variant
// carries the noise operand at all.
// SEG > 0: the tile's columns are cut into SEG segments along D and the (column, segment) items dealt round-robin (for tiles
// whose column count is not a multiple of the thread count: 360 columns x 2 segments over 256 threads is balanced, 360 whole
// columns are not); SEG == 0: stage_split decides.
template <typename T, int MODE, int UB, int NT = 256, bool NOISE_RT = false, int SEG = 0, int HDC = 0>
__device__ __forceinline__ void stage_halo_lean(const GatherIn& g, char* halo, const float* scs, const int* ctab, const int* rt,
                                                int n, int chunk, int tid) {
    const int mode = MODE >= 0 ? MODE : g.lean;
    constexpr bool NOISE = MODE >= 0 ? (MODE == VG_STAGE_LRELU_NOISE || MODE == VG_STAGE_LRELU_NOISE_M) : NOISE_RT;
    const bool noise_on = NOISE && (MODE >= 0 || mode == VG_STAGE_LRELU_NOISE || mode == VG_STAGE_LRELU_NOISE_M ||
                                    (mode == VG_STAGE_GENERIC && g.noise != nullptr));
    const bool MASK = mode == VG_STAGE_PLAIN || mode >= VG_STAGE_LRELU_M;    // zero padding / ragged last chunk possible
    const bool plain = mode == VG_STAGE_PLAIN, relu = mode == VG_STAGE_RELU;
    const bool generic = MODE < 0 && mode == VG_STAGE_GENERIC;
    const float gslope = g.act == VG_ACT_RELU ? 0.f : (g.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    constexpr int esz = (int)sizeof(T);
    const int L = stage_axis_len3(g);
    const int ncols = stage_ncols(g);
    const int plane = g.DS;
    const int sh = g.shift0;
    const char* b0 = (const char*)g.src0 + (size_t)n * (g.D >> sh) * (g.H >> sh) * (g.W >> sh) * g.c0 * esz;
    const char* b1 = (const char*)g.src1 + (size_t)n * g.D * g.H * g.W * g.c1 * esz;
    const char* nb = NOISE ? (const char*)g.noise + (size_t)n * (g.D + 2 * g.npad) * (g.H + 2 * g.npad) * (g.W + 2 * g.npad) * g.Cin * 2 : nullptr;
    StageSplit sp = stage_split<NT>(ncols, g.HD, tid);
    const int HDv = HDC > 0 ? HDC : g.HD;                        // (compile-time halo depth where the kernel fixes its tile)
    const int seglen = SEG > 0 ? (HDv + SEG - 1) / SEG : 0;
    const int nitems = SEG > 0 ? ncols * SEG : ncols;
    int sgi = SEG > 0 ? tid / ncols : 0, scol = SEG > 0 ? tid - sgi * ncols : 0;      // (segment, column) of the first item
    for (int item = (SEG > 0 ? tid : sp.col0); item < nitems; item += (SEG > 0 ? NT : sp.cstride)) {
        int col = item;
        if (SEG > 0) {
            col = scol; sp.hd_lo = sgi * seglen; sp.hd_hi = min(HDv, sp.hd_lo + seglen);
            scol += NT; if (scol >= ncols) { scol -= ncols; ++sgi; }                 // next item of this thread (NT <= ncols)
        }
        const int e = ctab[2 * col], hoff = ctab[2 * col + 1];
        const int hh = e & 1023, hw = (e >> 10) & 1023, cg = e >> 20;
        const int c = chunk * g.CK + cg * 8;
        const bool from0 = c < g.c0;
        const int* rs = rt + (from0 ? 0 : L);
        const int oh = rs[hh], ow = rs[g.HH + hw];
        const bool cvalid = !MASK || (c < g.Cin && (oh | ow) >= 0);
        const char* pc = (from0 ? b0 + (size_t)c * esz : b1 + (size_t)(c - g.c0) * esz) + (cvalid ? oh + ow : 0);
        if (!cvalid) pc = b0;                                              // invalid columns read a dummy, then zero
        const int* dt = rs + g.HH + g.HW + sp.hd_lo;
        const char* pn = nullptr; const int* ndt = nullptr; bool nvalid = false;
        if (NOISE) {
            if (noise_on) {
                const int nh = rt[2 * L + hh], nw = rt[2 * L + g.HH + hw];
                nvalid = cvalid && (nh | nw) >= 0;
                pn = nb + (nvalid ? nh + nw + c * 2 : 0);
                ndt = rt + 2 * L + g.HH + g.HW + sp.hd_lo;
            }
        }
        f32x2 sc[4], sf[4];
        if (!plain) {                       // (scs holds 1 / 0 when the source has no scale / shift: stage_scale_shift)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[j] = (f32x2){scs[cg * 8 + 2 * j], scs[cg * 8 + 2 * j + 1]};
                sf[j] = (f32x2){scs[g.CK + cg * 8 + 2 * j], scs[g.CK + cg * 8 + 2 * j + 1]};
            }
        }
        char* dcol = halo + hoff + (size_t)sp.hd_lo * plane;
        const int nd = sp.hd_hi - sp.hd_lo;
        for (int h0 = 0; h0 < nd; h0 += UB) {
            Raw8<T> raw[UB];
            Raw8<bf16_t> nz[NOISE ? UB : 1];
            int od[UB], nod[NOISE ? UB : 1];
#pragma unroll
            for (int k = 0; k < UB; ++k) {                                  // entries past the segment are real table words
                od[k] = dt[h0 + k];
                raw_load(raw[k], (const T*)(pc + (unsigned)max(od[k], 0)));
                if (NOISE) { if (noise_on) { nod[k] = ndt[h0 + k]; raw_load(nz[k], (const bf16_t*)(pn + (unsigned)max(nod[k], 0))); } }
            }
            // keep every load of the batch ahead of the first use: a load whose only use sits under `h0 + k < nd` is
            // otherwise sunk into that branch by the compiler and waited for with vmcnt(0), which drains the whole batch
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                raw_pin(raw[k]);
                if (NOISE) { if (noise_on) raw_pin(nz[k]); }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                if (h0 + k < nd) {
                    T* dst = (T*)(dcol + (size_t)(h0 + k) * plane);
                    const bool ok = cvalid && od[k] >= 0;
                    if (plain) {
                        Raw8<T> r = raw[k];
                        raw_mask(r, ok);
                        *(Raw8<T>*)dst = r;
                    } else {
                        float x[8];
                        raw_unpack(raw[k], x);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x2 v = {x[2 * j], x[2 * j + 1]};
                            v = v * sc[j] + sf[j];
                            x[2 * j] = v[0]; x[2 * j + 1] = v[1];
                        }
                        if (relu) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                        } else if (generic) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], x[j] * gslope);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], x[j] * VG_LRELU);
                        }
                        if (NOISE) {
                            if (noise_on) {
                                float z[8];
                                raw_unpack(nz[k], z);
                                const bool nok = nvalid && nod[k] >= 0;
#pragma unroll
                                for (int j = 0; j < 8; ++j) x[j] += nok ? z[j] : 0.f;
                            }
                        }
                        if (MASK) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = ok ? x[j] : 0.f;
                        }
                        store8<T>(dst, x);
                    }
                }
            }
        }
    }
}

// (Measured and dropped: the lean routine inside conv_kernel / conv32_kernel / wgrad_kernel -- both as a second code path
// and as the only path of the bf16 multi-channel instantiations.  Forward unchanged, data gradients 5-10 % faster, weight
// gradients 10-30 % SLOWER (the noise-carrying ones spill), train step 33.6 -> 35.0 ms: those kernels sit at their register
// caps and are bound by vector-instruction issue; a routine with fewer instructions does not help when the allocator
// answers it with spills.  DESIGN 6.14.)

// ---- host: validate the input side of a descriptor and derive the tile geometry for BM voxels ----
// padded pitches of the LDS halo image: the 16 voxels of an MFMA sub-tile (TW x 16/TW rows [x planes]) must fall into 16
// different 16-byte bank groups, i.e. be distinct modulo 16 units
// Padding is taken only while it costs <= 25 % of the image (thin tiles on small grids keep a 2-way conflict instead of a
// 2-3x larger LDS footprint, which would cost a resident workgroup); strided gathers (istr 2) cannot be made conflict-free
// by pitch alone and are not padded.
static inline void pad_pitches(int hh, int hw, int tw, int th, int istr, int& hhp, int& hwp) {
    const int hw0 = istr == 2 ? 2 * ((hw + 1) / 2) : hw;          // de-interleaved rows hold two halves of ceil(hw/2)
    hhp = hh; hwp = hw0;
    if (tw >= 16) return;
    int x = hw0; while (((istr * x) % (2 * tw)) != tw) ++x;       // rows of a sub-tile land tw units apart (odd multiples)
    int y = hh;
    if (tw * th < 16) { while (((istr * y * x) % (2 * tw * th)) != tw * th) ++y; }
    if ((long)x * y * 4 <= (long)hw0 * hh * 5) { hwp = x; hhp = y; }
}
// skew: extra bytes on the plane stride (0: consecutive planes share banks -- right for the forward kernel whose lane
// groups mix two planes over disjoint voxel sets; 64: the weight-gradient kernel's transposed reads take both planes of
// the same 8 voxels)
static inline int fill_gather(const vg_conv_desc* d, GatherIn& g, int CK, int BM, int skew = 0, int dma = 0) {
    if (!d || !d->src0) return VG_EINVAL;
    g.f32 = d->f32 ? 1 : 0;
#ifdef VG_ABLATE
    g.dbg = vg_tune("DEBUG", 0);        // diagnostic builds only (-DVG_ABLATE): phase ablations skip staging / MFMA / stores
#else
    g.dbg = 0;
#endif
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
    g.wpack = d->wpack; g.wmin = d->wpack_wmin; g.Cw = d->wpack ? d->wpack : Cin; g.HWx = 0;
    {   // lean staging mode (see stage_halo_lean); LEAN=0 switches it off everywhere (A/B and fallback)
        g.lean = -1;
        const bool zero = d->pad_mode != VG_PAD_REFLECT || (Cin % CK) != 0;
        if (Cin != 1 && !d->f32) {
            if (!d->in_scale && d->act == VG_ACT_NONE && !d->noise) g.lean = VG_STAGE_PLAIN;
            else if (d->in_scale && d->act == VG_ACT_RELU && !d->noise && !zero) g.lean = VG_STAGE_RELU;
            else if (d->in_scale && d->act == VG_ACT_LRELU)
                g.lean = d->noise ? (zero ? VG_STAGE_LRELU_NOISE_M : VG_STAGE_LRELU_NOISE) : (zero ? VG_STAGE_LRELU_M : VG_STAGE_LRELU);
            else g.lean = VG_STAGE_GENERIC;          // any other combination: affine (if given) + run-time activation + noise + mask
        }
    }
    if (d->wpack) {
        if (Cin != 1 || d->wpack < 2 || d->wpack > 8) return VG_EINVAL;
        for (int i = 0; i < d->ntaps; ++i) if (d->tap_w[i] != 0) return VG_EINVAL;
    }
    int mn[3] = {127, 127, 127}, mx[3] = {-128, -128, -128};
    for (int i = 0; i < d->ntaps; ++i) {
        g.td[i] = d->tap_d[i]; g.th[i] = d->tap_h[i]; g.tw[i] = d->tap_w[i];
        const int v[3] = {d->tap_d[i], d->tap_h[i], d->tap_w[i]};
        for (int a = 0; a < 3; ++a) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
    }
    g.tmin_d = mn[0]; g.tmin_h = mn[1]; g.tmin_w = mn[2];
    g.US = d->f32 ? 32 : 16;
    { const int pl = vg_tune("PLANAR", -1); g.planar = pl >= 0 ? pl : (CK <= 48 && d->istr == 1); }
    // tile shape: powers of two with product BM that minimise the halo volume (staging work and L2 traffic scale with
    // it); the innermost extent stays >= 8 voxels where the grid allows so that rows remain long contiguous runs
    int TW = 1, TH = 1, TD = 1;
    const int w16 = vg_tune("TILE_W16", 1);
    {
        const int ex[3] = {mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1};
        const int capd = pow2_ceil(d->OD), caph = pow2_ceil(d->OH), capw = pow2_ceil(d->OW) < 16 ? pow2_ceil(d->OW) : 16;
        long best = -1;
        for (int tw = 1; tw <= capw && tw <= BM; tw <<= 1) {
            if (tw < 8 && tw < capw) continue;
            for (int th = 1; th <= caph && tw * th <= BM; th <<= 1) {
                const int td = BM / (tw * th);
                const int hw_ = d->wpack ? tw : (tw - 1) * d->istr + ex[2], hh_ = (th - 1) * d->istr + ex[1];
                int hwp_ = (d->istr == 2 && !d->wpack) ? 2 * ((hw_ + 1) / 2) : hw_, hhp_ = hh_;
                if (g.planar) pad_pitches(hh_, hw_, tw, th, d->istr, hhp_, hwp_);
                long vol = (long)((td - 1) * d->istr + ex[0]) * hhp_ * hwp_;       // LDS image incl. pitch padding
                if (td > capd) vol *= 4;          // overhang in D wastes whole planes: only when nothing else fits
                // Planar images whose 16-voxel MFMA sub-tile spans two rows of 8 read with 2-way bank conflicts unless the row
                // pitch was padded (PMC on the 16->16 layers at 128^3: 43 % of the LDS cycles were conflict cycles with the
                // 10-unit pitch of the 8x8x8 tile).  A 16-wide tile has each sub-tile in ONE row -- consecutive 16-byte units,
                // conflict-free at any pitch -- for 8 % more halo (18x10x6 vs 10x10x10): charge the conflicting shapes 25 %.
                if (w16 && g.planar && skew == 0 && tw < 16 && hwp_ == hw_ && d->istr == 1) vol += vol / 4;
                if (best < 0 || vol < best || (vol == best && (tw > TW || (tw == TW && th > TH)))) { best = vol; TW = tw; TH = th; TD = td; }   // ties: wider, then taller (D is the walked axis)
            }
        }
    }
    g.twl = ilog2_exact(TW); g.thl = ilog2_exact(TH); g.tdl = ilog2_exact(TD);
    g.tiles_w = (d->OW + TW - 1) / TW; g.tiles_h = (d->OH + TH - 1) / TH; g.tiles_d = (d->OD + TD - 1) / TD;
    g.HD = (TD - 1) * d->istr + (mx[0] - mn[0]) + 1;
    g.HH = (TH - 1) * d->istr + (mx[1] - mn[1]) + 1;
    g.HW = (TW - 1) * d->istr + (mx[2] - mn[2]) + 1;
    if (d->wpack) { g.HWx = (TW - 1) * d->istr + d->wpack; g.HW = TW; }
    g.HWh = (d->istr == 2 && !d->wpack) ? (g.HW + 1) / 2 : 0;
    g.HHp = g.HH; g.HWp = g.HWh ? 2 * g.HWh : g.HW;
    if (g.planar) pad_pitches(g.HH, g.HW, TW, TH, d->istr, g.HHp, g.HWp);
    if (g.planar) {
        g.VS = g.US; g.DS = g.HHp * g.HWp * g.US;
        if (dma) g.DS = (g.DS + 1023) & ~1023;          // whole 64-unit pieces per D-slice (one LDS-DMA wave-instruction each)
        g.PSB = ((g.HD * g.DS + 255) & ~255) + skew; g.CS = g.PSB;
    } else { g.VS = CK * (d->f32 ? 4 : 2) + 16; g.CS = g.US; g.DS = g.HHp * g.HWp * g.VS; g.PSB = 0; }
    return VG_OK;
}
static inline int halo_bytes(const GatherIn& g) { return g.planar ? (g.CK >> 3) * g.PSB : g.HD * g.DS; }
// LDS ints of the staging tables (column table + per-tile axis tables)
static inline int stage_table_ints(const GatherIn& g) { return 2 * stage_ncols(g) + 6 * stage_axis_len(g); }   // column table + two axis-table buffers
static inline int stage_table_ints3(const GatherIn& g) {            // axis tables incl. the D axis (lean staging)
    const int a = stage_axis_len(g), b = stage_axis_len3(g);
    return 2 * stage_ncols(g) + 6 * (a > b ? a : b);
}
