// vg_conv_thin.hip -- specialist gather-convolution for the thin-channel 3x3x3 stride-1 layers of the two finest levels
// (stem.cb, dec0.cb1/cb2 at full resolution and their data gradients: the largest single share of the train step).
//
// Why a specialist: PMC on conv_kernel<16,8> shows the SIMDs 59 % busy issuing ordinary vector instructions
// (5 850 per 512-voxel tile) against 18 % for the matrix pipe -- the layer is bound by instruction count, and most of those
// instructions exist only because the generic kernel keeps its geometry and its epilogue options in run-time variables.
// Here the tile is FIXED at 16 (W) x 8 (H) x 4 (D) output voxels, 16 output channels per workgroup, 16 input channels per
// chunk, 27 taps: halo image 18 x 10 x 6 voxels, planar [channel group 2][6][10][18] x 16 bytes.  Consequences:
//   * a wave owns one D-plane of the tile = 8 sub-tiles of 16 consecutive voxels (one W row each): the B fragment of
//     sub-tile j is at  lane_base[K-step] + j * 288 bytes -- an immediate of ds_read_b128; lane_base[14] lives in registers
//     for the whole kernel, so the MFMA loop issues NO address arithmetic (the generic loop: 8 adds + a table read per K-step);
//   * the weight fragment of K-step s is at  lane_row + chunk_base + s * 64 -- an immediate as well;
//   * the loop is fully unrolled (14 K-steps) with the fragments of two K-steps ahead in flight;
//   * the epilogue options are template flags (bias, residual, statistics), the arithmetic is packed f32, two sub-tiles are
//     exchanged across the 16-lane rows with v_permlane16_swap so that every lane stores 16 contiguous bytes;
//   * staging is the lean routine (vg_gather.h) with a compile-time transform, its 360 columns x 2 D-segments dealt out evenly.
// Everything else (persistent workgroups, tables, InstanceNorm statistics carried in registers) follows conv_kernel.
#include "vg_conv_common.h"
#include <stdio.h>

// Timing-ablation knobs (VG_THIN_DBG, VG_WT_DBG: skip the MFMA loop / the commit / the statistics flush -- the RESULTS ARE WRONG, timing
// only, tools/r05_*_dbg.sh).  They act only in a development build (VG_EXTRA_DEFS=-DVG_DEBUG_ABLATE, van_gan_amd/build.py); the
// production library refuses the call (VG_EINVAL, loudly) when one of them is set, so that a stray environment variable cannot corrupt
// the gradients of dec0.cb1 / stem.cb silently.
static int vg_ablate_knob(const char* key) {
    const int v = vg_tune(key, 0);
#ifdef VG_DEBUG_ABLATE
    return v;
#else
    if (v != 0) { fprintf(stderr, "libvangan_hip: VG_%s=%d is a timing-ablation knob of a -DVG_DEBUG_ABLATE build; this library refuses it\n", key, v); return -1; }
    return 0;
#endif
}
#include "vg_dma_common.h"
#include <type_traits>
#include <algorithm>
#include <cstdio>

#ifndef VG_THIN_WPE
#define VG_THIN_WPE 2      // waves per SIMD the register allocation allows (3 would fit the LDS, 53.5 KB per workgroup, but spills 40-120 registers)
#endif
#ifndef VG_THIN_PF2
#define VG_THIN_PF2 0     // prefetch of the next halo under the MFMA loop in the two-panel forward instances too: spills 14-58 registers there
#endif
#ifndef VG_THIN_PD
#define VG_THIN_PD 1      // K-steps of fragments in flight ahead of the MFMAs (2 measured the same; 1 leaves the registers for the staged loads)
#endif
namespace {
constexpr int TW = 16, TH = 8, TD = 4;                 // output tile
constexpr int HW = 18, HH = 10, HD = 6;                // halo (3x3x3, stride 1)
constexpr int UNIT = 16;                               // bytes of one 8-channel bf16 unit
constexpr int ROWB = HW * UNIT;                        // 288: one halo row
constexpr int DSB = HH * ROWB;                         // 2880: one halo D-plane
constexpr int PSB = ((HD * DSB + 255) / 256) * 256;    // 17408: one channel-group plane (what fill_gather computes)
constexpr int HALO = 2 * PSB;                          // two channel groups (16 channels)
constexpr int KSTEPS = 14;                             // ceil(27 taps * 2 groups / 4)
constexpr int KCPAD = 448;                             // packed K of one 16-channel chunk (27 * 16 rounded to 32)
}

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Staging of the specialist: stage_halo_lean<T, MODE, 3, 256, false, 2, HD> cut in two.  thin_issue() starts all nine 16-byte loads
// of a thread's three (column, D-segment) items (720 items = 360 columns x 2 segments of 3 planes over 256 threads: items tid,
// tid + 256, tid + 512); thin_commit() transforms and writes them into the halo image.  The kernel issues the loads of the NEXT
// (tile, chunk) right before the MFMA loop of the current one and commits them after it, so the global round trip runs under the
// matrix work.  (The lean routine takes its items one after the other -- three round trips per tile and chunk, in the open: with two
// workgroups per CU there is little else to run under them.  Skipping the staging altogether took 27 % off the 16 -> 16 layer;
// skipping the MFMAs nothing.)  Only the raw data stay in registers across the loop: commit re-reads the table words it needs.
struct ThinItem { int hoff, cg; bool cv; const char* pc; const int* dt; };
// Axis tables of the WHOLE grid, built once per workgroup (thin_axis_tables): per source (src0, then src1) the byte offsets (or -1) of
// every H, W and D position a tile of this launch can touch -- entry (o0 + j) of an axis is halo position j of the tile with origin o0.
// (The per-tile tables of the lean routine cost 157 vector instructions per tile on three of the four waves: a sixth of a tile's vector
// work in a kernel that is bound by exactly that)
struct ThinTab { const int* tab; int AL, NH, NW, od0, oh0, ow0; };
__device__ __forceinline__ void thin_axis_tables(const GatherIn& g, int* tab, int NH, int NW, int ND, int tid) {
    const int AL = NH + NW + ND, sh = g.shift0;
    for (int i = tid; i < 2 * AL; i += 256) {
        const int set = i >= AL ? 1 : 0, j = i - set * AL;
        if (set == 1 && g.c1 == 0) { tab[i] = -1; continue; }
        const int axis = j < NH ? 1 : (j < NH + NW ? 2 : 0);                              // 1 H, 2 W, 0 D
        const int jj = axis == 1 ? j : (axis == 2 ? j - NH : j - NH - NW);
        const int n_ax = axis == 1 ? g.H : (axis == 2 ? g.W : g.D);
        int p = (axis == 1 ? g.tmin_h : (axis == 2 ? g.tmin_w : g.tmin_d)) + jj;
        const bool valid = resolve_pos(p, n_ax, g.pad_mode);
        long off;
        if (set == 0) {
            const int ps = p >> sh, Ws = g.W >> sh, Hs = g.H >> sh;
            off = axis == 1 ? (long)ps * Ws * g.c0 : (axis == 2 ? (long)ps * g.c0 : (long)ps * Hs * Ws * g.c0);
        } else off = axis == 1 ? (long)p * g.W * g.c1 : (axis == 2 ? (long)p * g.c1 : (long)p * g.H * g.W * g.c1);
        tab[i] = valid ? (int)(off * 2) : -1;
    }
}
template <bool PLAIN>
__device__ __forceinline__ ThinItem thin_item(const GatherIn& g, const int* ctab, const ThinTab& tt, const char* b0, const char* b1, int chunk, int item) {
    constexpr int NCOLS = HH * HW * 2, SEGL = HD / 2;
    ThinItem t;
    const int seg = item >= NCOLS ? 1 : 0, col = item - seg * NCOLS;
    const int e = ctab[2 * col];
    t.hoff = ctab[2 * col + 1] + seg * SEGL * DSB;
    const int hh = e & 1023, hw = (e >> 10) & 1023;
    t.cg = e >> 20;
    const int c = chunk * 16 + t.cg * 8;
    const bool from0 = c < g.c0;
    const int* rs = tt.tab + (from0 ? 0 : tt.AL);
    const int oh = rs[tt.oh0 + hh], ow = rs[tt.NH + tt.ow0 + hw];
    t.cv = !PLAIN || (c < g.Cin && (oh | ow) >= 0);
    t.pc = t.cv ? (from0 ? b0 + (size_t)c * 2 : b1 + (size_t)(c - g.c0) * 2) + (oh + ow) : b0;          // invalid columns read a dummy, then zero
    t.dt = rs + tt.NH + tt.NW + tt.od0 + seg * SEGL;
    return t;
}
template <int MODE>
__device__ __forceinline__ void thin_issue(const GatherIn& g, const int* ctab, const ThinTab& rt, int n, int chunk, int tid, Raw8<bf16_t> (&raw)[3][HD / 2]) {
    typedef bf16_t T;
    constexpr int NITEMS = 4 * HH * HW, SEGL = HD / 2;
    const int sh = g.shift0;
    const char* b0 = (const char*)g.src0 + (size_t)n * (g.D >> sh) * (g.H >> sh) * (g.W >> sh) * g.c0 * 2;
    const char* b1 = (const char*)g.src1 + (size_t)n * g.D * g.H * g.W * g.c1 * 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int item = tid + 256 * i;
        if (i < 2 || item < NITEMS) {
            const ThinItem t = thin_item<MODE == VG_STAGE_PLAIN>(g, ctab, rt, b0, b1, chunk, item);
#pragma unroll
            for (int k = 0; k < SEGL; ++k) raw_load(raw[i][k], (const T*)(t.pc + (unsigned)max(t.dt[k], 0)));
        }
    }
}
template <int MODE>
__device__ __forceinline__ void thin_commit(const GatherIn& g, char* halo, const float* scs, const int* ctab, const ThinTab& rt, int chunk, int tid,
                                            Raw8<bf16_t> (&raw)[3][HD / 2]) {
    typedef bf16_t T;
    constexpr bool plain = MODE == VG_STAGE_PLAIN;
    constexpr int NITEMS = 4 * HH * HW, SEGL = HD / 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int item = tid + 256 * i;
        if (i < 2 || item < NITEMS) {
            const ThinItem t = thin_item<plain>(g, ctab, rt, nullptr, nullptr, chunk, item);
            f32x2 sc[4], sf[4];
            if (!plain) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sc[j] = (f32x2){scs[t.cg * 8 + 2 * j], scs[t.cg * 8 + 2 * j + 1]};
                    sf[j] = (f32x2){scs[16 + t.cg * 8 + 2 * j], scs[16 + t.cg * 8 + 2 * j + 1]};
                }
            }
#pragma unroll
            for (int k = 0; k < SEGL; ++k) {
                T* dst = (T*)(halo + t.hoff + k * DSB);
                if (plain) {
                    Raw8<T> r = raw[i][k];
                    raw_mask(r, t.cv && t.dt[k] >= 0);
                    *(Raw8<T>*)dst = r;
                } else {
                    float x[8];
                    raw_unpack(raw[i][k], x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x2 v = {x[2 * j], x[2 * j + 1]};
                        v = v * sc[j] + sf[j];
                        x[2 * j] = fmaxf(v[0], 0.f); x[2 * j + 1] = fmaxf(v[1], 0.f);
                    }
                    store8<T>(dst, x);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// UP: channel chunks that come from the virtually UPSAMPLED half-resolution tensor of a decoder block's first convolution
// (resunet_model.py:175-181: UpSampling3D(2) -> concatenate -> IN -> ReLU -> reflect pad -> 3^3 convolution).  IN and ReLU act per
// voxel and channel, so they commute with the nearest-neighbour upsampling, and along D and H the three taps of a fine output voxel of
// parity p read only TWO half-resolution voxels: p = 0 -> (i - 1) with weight W[-1] and i with W[0] + W[1]; p = 1 -> i with
// W[-1] + W[0] and (i + 1) with W[1]; the reflection pad at fine resolution is edge replication at half resolution.  A sub-tile of this
// kernel is one fine row (fixed d, h): its 16 columns share the (D, H) parity class, so the accumulator layout and the whole epilogue
// stay as they are -- only the operand changes: the halo image holds 4 half-resolution planes x 6 half-resolution rows x 18 FINE
// columns (W is not collapsed: the columns of a sub-tile alternate in W parity), 13.8 instead of 34.6 KB staged and transformed per
// chunk, and the contraction is 12 collapsed taps x 16 channels = 6 K-steps instead of 14, with one weight panel per class (vg_pack_up_weights).
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int UPD = 4, UPH = 6;                         // half-resolution planes / rows of a tile's halo
constexpr int UPCOLS = UPH * HW * 2;                   // (row, column, channel group) columns of one plane: 216
constexpr int UPITEMS = UPCOLS * 2;                    // x 2 plane pairs
constexpr int KUP = 192;                               // 12 collapsed taps x 16 channels
constexpr int KSUP = 6;                                // K-steps of an upsampled chunk
constexpr int WRSU = KUP * 2 + 16;                     // LDS row stride of a class panel
}
// column table of the half-resolution image (same plane geometry as the fine one: row stride ROWB, plane stride DSB)
__device__ __forceinline__ void build_up_table(int* utu, int tid) {
    for (int col = tid; col < UPCOLS; col += 256) {
        const int cg = col / (UPH * HW), v = col - cg * (UPH * HW);
        const int lr = v / HW, hw = v - lr * HW;
        utu[2 * col] = lr | (hw << 10) | (cg << 20);
        utu[2 * col + 1] = cg * PSB + (lr * HW + hw) * UNIT;
    }
}
// half-resolution axis tables of src0 for the whole grid: entry j of an axis is half-resolution position (j - 1), edge-replicated
__device__ __forceinline__ void thin_up_axis_tables(const GatherIn& g, int* lowtab, int NHl, int NDl, int tid) {
    const int Hs = g.H >> 1, Ws = g.W >> 1, Ds = g.D >> 1;
    for (int i = tid; i < NHl + NDl; i += 256) {
        const bool isH = i < NHl;
        int L = (isH ? i : i - NHl) - 1;
        const int n_ax = isH ? Hs : Ds;
        L = L < 0 ? 0 : (L >= n_ax ? n_ax - 1 : L);
        const long off = isH ? (long)L * Ws * g.c0 : (long)L * Hs * Ws * g.c0;
        lowtab[i] = (int)(off * 2);
    }
}
struct ThinUpTab { const int* wtab; const int* lowH; const int* lowD; };          // wtab: the fine W axis table of src0 at this tile's origin
template <int MODE>
__device__ __forceinline__ void thin_issue_up(const GatherIn& g, const int* utu, const ThinUpTab& t, int n, int chunk, int tid, Raw8<bf16_t> (&raw)[3][HD / 2]) {
    const char* b0 = (const char*)g.src0 + (size_t)n * (g.D >> 1) * (g.H >> 1) * (g.W >> 1) * g.c0 * 2;
    asm volatile("" : "+v"(tid));             // (nothing of the per-thread item decode is to live across the tile loop)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = tid + 256 * i;
        if (i < 1 || item < UPITEMS) {
            const int seg = item >= UPCOLS ? 1 : 0, col = item - seg * UPCOLS;
            const int e = utu[2 * col];
            const int lr = e & 1023, hw = (e >> 10) & 1023, cg = e >> 20;
            const char* pc = b0 + (size_t)(chunk * 16 + cg * 8) * 2 + (t.lowH[lr] + max(t.wtab[hw], 0));
#pragma unroll
            for (int k = 0; k < 2; ++k) raw_load(raw[i][k], (const bf16_t*)(pc + (unsigned)t.lowD[seg * 2 + k]));
        }
    }
}
__device__ __forceinline__ void thin_commit_up(char* halo, const float* scs, const int* utu, int tid, Raw8<bf16_t> (&raw)[3][HD / 2]) {
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = tid + 256 * i;
        if (i < 1 || item < UPITEMS) {
            const int seg = item >= UPCOLS ? 1 : 0, col = item - seg * UPCOLS;
            const int cg = utu[2 * col] >> 20, hoff = utu[2 * col + 1] + seg * 2 * DSB;
            f32x2 sc[4], sf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[j] = (f32x2){scs[cg * 8 + 2 * j], scs[cg * 8 + 2 * j + 1]};
                sf[j] = (f32x2){scs[16 + cg * 8 + 2 * j], scs[16 + cg * 8 + 2 * j + 1]};
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float x[8];
                raw_unpack(raw[i][k], x);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 v = {x[2 * j], x[2 * j + 1]};
                    v = v * sc[j] + sf[j];
                    x[2 * j] = fmaxf(v[0], 0.f); x[2 * j + 1] = fmaxf(v[1], 0.f);
                }
                store8<bf16_t>((bf16_t*)(halo + hoff + k * DSB), x);
            }
        }
    }
}

// MODE: VG_STAGE_PLAIN (data-gradient operand, zero padded) or VG_STAGE_RELU (forward: IN affine + ReLU, reflect padded)
// BSTAT (data gradient): the epilogue also accumulates the statistics of the IN backward that consumes this output --
// sum dn and sum dn * xhat with dn = g * mult * act'(x * scale + shift) taken at the reflect-folded position of the pre-norm
// tensor x -- into p.sums (same striped layout as the forward statistics), so that the statistics pass need not re-read g.
// PL (data gradient of a 16-channel tensor into 16 PL channels -- dec0.cb1's 16 -> 48): the PL output panels are LOOPED over one staged
// halo image instead of being separate workgroups that each stage it; the 16 x 448 weight panel of the next use arrives by LDS-DMA
// (two buffers) under the MFMA loop of the current one, the statistics of a panel are flushed to LDS per tile (registers).
template <int MODE, bool BIAS, bool RES, bool STATS, bool BSTAT = false, int NP = 1, int PL = 1, bool UP = false>
__global__ __launch_bounds__(256, VG_THIN_WPE) void conv_thin_kernel(const GatherIn g, const ConvOut p) {
    static_assert(PL == 1 || (NP == 1 && MODE == VG_STAGE_PLAIN && !BIAS && !RES && !STATS), "panel loop: plain data gradient only");
    static_assert(!UP || (MODE == VG_STAGE_RELU && !RES && !BSTAT && PL == 1 && NP == 1), "collapsed upsampled chunks: the decoder's first convolution, forward, one panel");
    // NP: 16-channel output panels per workgroup.  NP = 2 (the 32-channel layers at 64^3, conv_thin2 in the variant names): every B
    // fragment feeds two MFMAs -- 10 fragment reads per 16 MFMAs instead of 9 per 8 -- and the halo is staged once per 32 channels.
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    const int n = blockIdx.z, ntile = blockIdx.y;
    // ---- LDS: [halo][scale/shift 2*16 floats][statistics 32*NP floats][tap offsets 32 ints][column table][axis tables x 2]
    //           [weights of ONE 16-channel chunk: 16*NP rows x (448 + 8) bf16]
    char* halo = smem;
    float* scs = (float*)(smem + HALO);
    float* stat = scs + 32;
    float* bsc = stat + 32 * NP * PL;                    // BSTAT: [scale | shift | rstd | -mean * rstd][16 NP PL channels of this workgroup]
    float* ebs = bsc + 64 * NP * PL;                     // two panels: [bias | residual scale | residual shift][32 channels] (read in the epilogue)
    int* tapb = (int*)(ebs + (NP > 1 ? 48 * NP : 0));
    int* utab = tapb + 32;
    constexpr int NCOLS = HH * HW * 2;
    int* xtab = utab + 2 * NCOLS;                        // axis tables of the whole grid (thin_axis_tables)
    const int NHt = g.tiles_h * TH + 2, NWt = g.tiles_w * TW + 2, NDt = g.tiles_d * TD + 2;
    // UP: half-resolution axis tables of src0 [NHl + NDl] and the column table of the half-resolution image
    const int NHl = UP ? g.tiles_h * (TH / 2) + 2 : 0, NDl = UP ? g.tiles_d * (TD / 2) + 2 : 0;
    int* lowtab = xtab + 2 * (NHt + NWt + NDt);
    int* utu = lowtab + NHl + NDl;
    char* wlds = (char*)(utu + (UP ? 2 * UPCOLS + 16 : 0));
    // WDMA (one-panel forward instances with several channel chunks): the next pass's 16 x 448 weight panel arrives by LDS-DMA in a second
    // buffer under the MFMA loop of the current pass, instead of a global -> register -> LDS copy in the open at the head of every pass
    const bool wdma = MODE == VG_STAGE_RELU && NP == 1 && PL == 1 && !UP && p.wdma;
    wlds = (char*)(((size_t)wlds + ((PL > 1 || wdma) ? 1023 : 15)) & ~(size_t)((PL > 1 || wdma) ? 1023 : 15));
    const int Ktot = p.Ktot, nchunks = p.nchunks;
    constexpr int WRS = KCPAD * 2 + 16;                  // LDS row stride of the chunk panel (16 bytes of padding: bank spread)
    const int cop = ntile * 16 * NP * PL;                // first output channel of this workgroup

    build_column_table(g, utab, tid);
    if (tid < 32 * NP * PL) stat[tid] = 0.f;
    if constexpr (NP > 1 && (BIAS || RES)) {
        if (tid < 16 * NP) {
            const int c = cop + tid;
            ebs[tid] = BIAS ? p.bias[c] : 0.f;
            if (RES) { ebs[16 * NP + tid] = p.rs[n * p.Cout + c]; ebs[32 * NP + tid] = p.rb[n * p.Cout + c]; }
        }
    }
    if constexpr (BSTAT) {
        // per-channel constants of the statistics: from LDS in the epilogue (fetched from memory there, every tile waited an L2 round trip for them)
        constexpr int CW = 16 * NP * PL;
        if (tid < CW) {
            const int nc = n * p.Cout + cop + tid;
            const float rs = p.bs_rs[nc];
            bsc[tid] = p.bs_sc[nc]; bsc[CW + tid] = p.bs_sf[nc]; bsc[2 * CW + tid] = rs; bsc[3 * CW + tid] = -p.bs_mu[nc] * rs;
        }
    }
    if (tid < 27) tapb[tid] = (g.td[tid] - g.tmin_d) * DSB + (g.th[tid] - g.tmin_h) * ROWB + (g.tw[tid] - g.tmin_w) * UNIT;
    // the (16 NP) x 448 weight panel of one chunk -> LDS: 16 NP rows x 56 units of 16 bytes, 3.5 NP per thread
    auto load_weights = [&](int chunk) {
        const char* src = (const char*)p.wp + ((size_t)cop * Ktot + (size_t)chunk * KCPAD) * 2;
        constexpr int NU = 16 * NP * 56, NK = (NU + 255) / 256;
        f32x4 v[NK];
        int tid = threadIdx.x;
        if constexpr (UP) asm volatile("" : "+v"(tid));          // (see load_weights_up: no hoisting of the per-thread addresses in the UP instances)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int u = min(tid + k * 256, NU - 1);
            const int r = u / 56, c = u - r * 56;
            v[k] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(src + (size_t)r * Ktot * 2 + c * 16);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int u = tid + k * 256;
            if (u < NU) { const int r = u / 56, c = u - r * 56; *(f32x4*)(wlds + r * WRS + c * 16) = v[k]; }
        }
    };
    // PL: one use's panel as 15 LDS-DMA pieces of 64 x 16 bytes, dealt to the waves; unit u = row * 57 + c (c == 56: the row's padding)
    constexpr int WBUF = 15 * 1024;
    const unsigned wlds_a = (unsigned)(uintptr_t)(lds_void_d*)wlds;
    auto dma_weights = [&](int panel, int buf, int chunk_ = 0) {
        const char* src = (const char*)p.wp + ((size_t)(cop + 16 * panel) * Ktot + (size_t)chunk_ * KCPAD) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = wave + 4 * k;
            if (piece < 15) {
                const int u = piece * 64 + lane, r = min(u / 57, 15), c = min(u - (u / 57) * 57, 55);
                glds16(src, r * Ktot * 2 + c * 16, wlds_a + buf * WBUF + piece * 1024);
            }
        }
    };
    if constexpr (PL > 1) dma_weights(0, 0); else if constexpr (!UP) { if (wdma) dma_weights(0, 0, 0); else load_weights(0); }        // (UP: chunk 0 is an upsampled one: loaded in the chunk loop, the lambda needs the tables)
    // ---- per-lane constants of the MFMA loop: B-fragment base of every K-step (tap and channel group of this lane's k-group)
    const int wbase = li * WRS + kg * 16;                                // A fragment: row li of panel 0, k-group kg (panel q: + 16 q rows)
    const int wbase_up = li * WRSU + kg * 16;                            // ... of a class panel of an upsampled chunk (UP)
    const int co0 = cop + 4 * kg;                                        // this lane's 4 output channels of panel 0 (panel q: + 16 q)
    float s1[NP * PL][4], s2[NP * PL][4];              // PL: [0] is the panel in work, [1], [2] the next two (rotated after every panel)
    f32x2 e_b[NP][2], e_rs[NP][2], e_rb[NP][2];
#pragma unroll
    for (int q = 0; q < NP * PL; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[q][r] = 0.f; s2[q][r] = 0.f; }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int c = co0 + 16 * q;
        e_b[q][0] = (f32x2){0.f, 0.f}; e_b[q][1] = e_b[q][0];
        if (BIAS && NP == 1) { e_b[q][0] = (f32x2){p.bias[c], p.bias[c + 1]}; e_b[q][1] = (f32x2){p.bias[c + 2], p.bias[c + 3]}; }
        if (RES && NP == 1) {
            e_rs[q][0] = (f32x2){p.rs[n * p.Cout + c], p.rs[n * p.Cout + c + 1]}; e_rs[q][1] = (f32x2){p.rs[n * p.Cout + c + 2], p.rs[n * p.Cout + c + 3]};
            e_rb[q][0] = (f32x2){p.rb[n * p.Cout + c], p.rb[n * p.Cout + c + 1]}; e_rb[q][1] = (f32x2){p.rb[n * p.Cout + c + 2], p.rb[n * p.Cout + c + 3]};
        }
    }
    // BSTAT: the source tensor of each 16-channel panel (the per-channel constants are fetched in the epilogue:
    // held across the MFMA loop they pushed this instance over its register cap)
    const T* b_x[NP]; int b_cs[NP], b_sh[NP]; float b_slope = 1.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) { b_x[q] = nullptr; b_cs[q] = 0; b_sh[q] = 0; }
    auto bstat_panel = [&](const int pn16) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int c = co0 + 16 * q + pn16;
            const bool lo = c < p.bs_c0;                                 // panel-uniform (bs_c0 is a multiple of 16)
            b_sh[q] = lo ? p.bs_sh : 0; b_cs[q] = lo ? p.bs_c0 : p.Cout - p.bs_c0;
            b_x[q] = (lo ? (const T*)p.bs_x0 + c : (const T*)p.bs_x1 + (c - p.bs_c0))
                     + (size_t)n * (p.bs_D >> b_sh[q]) * (p.bs_H >> b_sh[q]) * (p.bs_W >> b_sh[q]) * b_cs[q];
        }
    };
    if (BSTAT) {
        if constexpr (PL == 1) bstat_panel(0);
        b_slope = p.bs_act == VG_ACT_RELU ? 0.f : (p.bs_act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    }
    // 16-byte stores: after the row swap an even k-group lane holds channels 8*(kg/2)..+7 of sub-tile j, an odd one of j+1
    const int cst = cop + 8 * (kg >> 1);
    const int jodd = kg & 1;

    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    // tile walk.  p.xw (gridDim.x a multiple of 8): the workgroups that share an XCD (blockIdx.x & 7: round-robin placement; speed only)
    // walk ONE contiguous slab of the tile sequence (w fastest, then h, then d) side by side, so the halo voxels that neighbouring tiles
    // share -- 2.1 x the tile with this 16 x 8 x 4 shape -- are found in that XCD's L2 instead of being fetched once per XCD
    int t0 = blockIdx.x, tstep = gridDim.x, tend = tiles_per_n;
    if (p.xw) {
        const int slab = (tiles_per_n + 7) >> 3, xcd = blockIdx.x & 7;
        t0 = xcd * slab + ((int)blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = min(tiles_per_n, (xcd + 1) * slab);
    }
    int gs_w, gs_h, gs_d;
    { int t = tstep; gs_w = t % g.tiles_w; t /= g.tiles_w; gs_h = t % g.tiles_h; gs_d = t / g.tiles_h; }
    int ti_w, ti_h, ti_d;
    { int t = t0; ti_w = t % g.tiles_w; t /= g.tiles_w; ti_h = t % g.tiles_h; ti_d = t / g.tiles_h; }
    if (nchunks == 1) stage_scale_shift(g, scs, n, 0, tid);
    thin_axis_tables(g, xtab, NHt, NWt, NDt, tid);
    if constexpr (UP) { build_up_table(utu, tid); thin_up_axis_tables(g, lowtab, NHl, NDl, tid); }
    lds_only_barrier();
    auto tabat = [&](int od0_, int oh0_, int ow0_) { return ThinTab{xtab, NHt + NWt + NDt, NHt, NWt, od0_, oh0_, ow0_}; };
    int boff[KSTEPS];
    {
        const int rowbase = wave * DSB + li * UNIT;                      // sub-tile 0 of this wave's plane, this lane's voxel
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            int G = 4 * s + kg; if (G > 53) G = 53;                       // padded K (weights are zero there)
            boff[s] = rowbase + tapb[G >> 1] + (G & 1) * PSB;
        }
    }

    // UP: B-fragment base of the 6 K-steps of an upsampled chunk: k-group G = 4 s + kg -> collapsed tap G / 2 = (td * 2 + th) * 3 + tw
    // of the 2 x 2 x 3 stencil on the half-resolution image, channel group G & 1; this wave's fine plane `wave` reads half-resolution
    // planes ((wave + 1) >> 1) + td, sub-tile (fine row) j rows ((j + 1) >> 1) + th (the row part is an immediate of the read)
    // (not kept in registers across the tile loop -- the instance has none to spare: the six bases are rebuilt from a 12-entry LDS table
    //  at the head of every upsampled chunk's MFMA loop)
    int* tapu = utu + 2 * UPCOLS;
    if constexpr (UP) { if (tid < 12) { const int cdh = tid / 3, cw = tid - cdh * 3; tapu[tid] = (cdh >> 1) * DSB + (cdh & 1) * ROWB + cw * UNIT; } }
    const int pd_cls = UP ? (wave & 1) * 2 : 0;                          // class = pd * 2 + ph; this wave's plane parity
    constexpr int CLSB = 16 * NP * WRSU;                                 // bytes of one class panel in LDS
    // the four class panels of one upsampled chunk -> LDS: 64 NP rows x 24 units of 16 bytes
    auto load_weights_up = [&](int chunk) {
        const char* src = (const char*)p.wp_up + ((size_t)chunk * 4 * p.Cout + cop) * KUP * 2;
        constexpr int NU = 4 * 16 * NP * 24, NK = (NU + 255) / 256;
        f32x4 v[NK];
        int tl = tid;
        asm volatile("" : "+v"(tl));          // opaque: the per-thread source / destination addresses are rebuilt here, not hoisted out of the tile loop
                                              // (hoisted they are 6 NP 64-bit + 32-bit values per thread held across everything: 60-190 spilled registers)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int u = min(tl + k * 256, NU - 1);
            const int r = u / 24, c = u - r * 24, cls = r / (16 * NP), rr = r - cls * 16 * NP;
            v[k] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(src + ((size_t)cls * p.Cout + rr) * KUP * 2 + c * 16);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int u = tl + k * 256;
            if (u < NU) { const int r = u / 24, c = u - r * 24; *(f32x4*)(wlds + r * WRSU + c * 16) = v[k]; }
        }
    };
    auto uptab = [&](int od0_, int oh0_, int ow0_) { return ThinUpTab{xtab + NHt + ow0_, lowtab + (oh0_ >> 1), lowtab + NHl + (od0_ >> 1)}; };
    typedef const __attribute__((address_space(3))) bf16x8 lds_frag;
    int it = 0;
    // PF: the loads of the NEXT (tile, chunk) are issued ahead of the MFMA loop and stay in registers across it.  Forward instances
    // only (16 -> 16: 68 -> 64 us, 48 -> 16: 148 -> 142 us); the data-gradient instances got SLOWER with it (16 -> 48: 178 -> 195 us;
    // the BSTAT one sits at the register cap and spilled the nine units): they load, then commit, back to back.
    constexpr bool PF = MODE == VG_STAGE_RELU && (NP == 1 || VG_THIN_PF2);
    Raw8<T> raw[3][HD / 2];
    if (PF && t0 < tend) {
        if constexpr (UP) thin_issue_up<MODE>(g, utu, uptab(ti_d * TD, ti_h * TH, ti_w * TW), n, 0, tid, raw);
        else thin_issue<MODE>(g, utab, tabat(ti_d * TD, ti_h * TH, ti_w * TW), n, 0, tid, raw);
    }
    bf16x4 bxp[(BSTAT && NP == 1) ? 8 : 1];
    bf16x4 bxn[(BSTAT && PL > 1) ? 8 : 1];                                 // (PL: the rows requested a panel ahead)
    for (int tile = t0; tile < tend; tile += tstep, ++it) {
        const int od0 = ti_d * TD, oh0 = ti_h * TH, ow0 = ti_w * TW;
        ti_w += gs_w; if (ti_w >= g.tiles_w) { ti_w -= g.tiles_w; ++ti_h; }
        ti_h += gs_h; if (ti_h >= g.tiles_h) { ti_h -= g.tiles_h; ++ti_d; }
        ti_d += gs_d;
        const bool more = tile + tstep < tend;
        f32x4 acc[NP][8];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // BSTAT (one panel): the pre-norm values of this lane's 8 output rows are requested HERE, ahead of the tile's staging and MFMA loop,
        // and consumed in the epilogue -- issued there (round 3) every tile waited a full HBM round trip for them with nothing left to run
        auto issue_bxp = [&]() {
            auto fold = [&](int qq, int nn) { int i = qq - p.bs_pad; i = i < 0 ? -i : i; i = i >= nn ? 2 * nn - 2 - i : i; return min(max(i, 0), nn - 1); };
            const int XH = p.bs_H >> b_sh[0], XW = p.bs_W >> b_sh[0];
            const int id = fold(od0 + wave + p.ood, p.bs_D) >> b_sh[0], iw = fold(ow0 + li + p.oow, p.bs_W) >> b_sh[0];
            const T* xcol = b_x[0] + ((size_t)id * XH * XW + iw) * b_cs[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ih = fold(oh0 + j + p.ooh, p.bs_H) >> b_sh[0];
                bxp[j] = *(const __attribute__((address_space(1))) bf16x4*)(uintptr_t)(xcol + (size_t)ih * XW * b_cs[0]);
            }
        };
        if constexpr (BSTAT && NP == 1 && PL == 1) issue_bxp();
        // PL: the pre-norm rows of the NEXT panel (or the next tile's first) are requested row pair by row pair inside the epilogue, as soon as
        // the pair's registers are free -- a whole MFMA loop ahead of their use (requested in front of the loop they were waited for: 12 %)
        const T* n_xcol = nullptr; int n_rowp = 0, n_sh = 0, n_oh0 = 0;
        auto bx_prepare = [&](const int pn16, const int od0_, const int oh0_, const int ow0_) {
            auto fold = [&](int qq, int nn) { int i = qq - p.bs_pad; i = i < 0 ? -i : i; i = i >= nn ? 2 * nn - 2 - i : i; return min(max(i, 0), nn - 1); };
            bstat_panel(pn16);
            const int XH = p.bs_H >> b_sh[0], XW = p.bs_W >> b_sh[0];
            const int id = fold(od0_ + wave + p.ood, p.bs_D) >> b_sh[0], iw = fold(ow0_ + li + p.oow, p.bs_W) >> b_sh[0];
            n_xcol = b_x[0] + ((size_t)id * XH * XW + iw) * b_cs[0];
            n_rowp = XW * b_cs[0]; n_sh = b_sh[0]; n_oh0 = oh0_;
        };
        auto bx_addr = [&](const int j) {
            auto fold = [&](int qq, int nn) { int i = qq - p.bs_pad; i = i < 0 ? -i : i; i = i >= nn ? 2 * nn - 2 - i : i; return min(max(i, 0), nn - 1); };
            const int ih = fold(n_oh0 + j + p.ooh, p.bs_H) >> n_sh;
            return n_xcol + (size_t)ih * n_rowp;
        };
        auto bx_row = [&](const int j) { return *(const __attribute__((address_space(1))) bf16x4*)(uintptr_t)bx_addr(j); };
        // the MFMA loop of one panel: 14 K-steps x 8 sub-tiles, every address an immediate, the fragments of the next K-step in flight
        auto mfma_panel = [&](const char* wb) {
            constexpr int PD = VG_THIN_PD, NB = PD + 1;
            bf16x8 a[NB], b[NB][8];
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                a[u] = *(lds_frag*)(wb + u * 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) b[u][j] = *(lds_frag*)(halo + boff[u] + j * ROWB);
            }
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (s + PD < KSTEPS) {
                    a[(s + PD) % NB] = *(lds_frag*)(wb + (s + PD) * 64);
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[(s + PD) % NB][j] = *(lds_frag*)(halo + boff[s + PD] + j * ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[0][j] = VG_MFMA16(a[s % NB], b[s % NB][j], acc[0][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (PL == 1)
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool upc = UP && chunk < p.nup;                          // (workgroup-uniform) this chunk comes from the upsampled half-resolution tensor
            if (it | chunk) lds_only_barrier();                            // previous readers of the halo image / weight panel are done
            if (nchunks > 1) {
                stage_scale_shift(g, scs, n, chunk, tid);
                if constexpr (UP) { if (upc) load_weights_up(chunk); else load_weights(chunk); }
                else if ((it | chunk) && !wdma) load_weights(chunk);
                lds_only_barrier();
            }
            if constexpr (UP) {
                if (upc) { if (!PF) thin_issue_up<MODE>(g, utu, uptab(od0, oh0, ow0), n, chunk, tid, raw); thin_commit_up(halo, scs, utu, tid, raw); }
                else { if (!PF) thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, chunk, tid, raw); thin_commit<MODE>(g, halo, scs, utab, tabat(od0, oh0, ow0), chunk, tid, raw); }
            } else {
            if (!PF) thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, chunk, tid, raw);
            thin_commit<MODE>(g, halo, scs, utab, tabat(od0, oh0, ow0), chunk, tid, raw);
            }
            if (wdma) __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's pieces of THIS pass's panel (requested a pass ago) have landed
            lds_only_barrier();
            // the next (tile, chunk)'s loads go out now and land under the MFMA loop
            if (PF) {
                if constexpr (UP) {
                    if (chunk + 1 < nchunks) {
                        if (chunk + 1 < p.nup) thin_issue_up<MODE>(g, utu, uptab(od0, oh0, ow0), n, chunk + 1, tid, raw);
                        else thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, chunk + 1, tid, raw);
                    } else if (more) thin_issue_up<MODE>(g, utu, uptab(ti_d * TD, ti_h * TH, ti_w * TW), n, 0, tid, raw);
                } else {
                if (chunk + 1 < nchunks) thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, chunk + 1, tid, raw);
                else if (more) thin_issue<MODE>(g, utab, tabat(ti_d * TD, ti_h * TH, ti_w * TW), n, 0, tid, raw);
                }
            }
            const int wuse = it * nchunks + chunk;                         // pass counter: its panel sits in buffer wuse & 1
            if (wdma && (chunk + 1 < nchunks || more)) dma_weights(0, (wuse + 1) & 1, chunk + 1 < nchunks ? chunk + 1 : 0);
            if constexpr (UP) {
                if (upc) {
                    // ---- MFMA loop of an upsampled chunk: 6 K-steps; per K-step two weight fragments per panel (the two H-parity classes of
                    //      this wave's plane) and FIVE image fragments -- fine rows 2 r - 1 and 2 r read the same half-resolution rows
                    const char* wbu = wlds + pd_cls * CLSB + wbase_up;
                    int boff_up[KSUP];
                    {
                        const int upbase = ((wave + 1) >> 1) * DSB + li * UNIT;
#pragma unroll
                        for (int s_ = 0; s_ < KSUP; ++s_) { const int G = 4 * s_ + kg; boff_up[s_] = upbase + tapu[G >> 1] + (G & 1) * PSB; }
                    }
                    bf16x8 ae[2][NP], ao[2][NP], bu[2][5];
#pragma unroll
                    for (int q = 0; q < NP; ++q) { ae[0][q] = *(lds_frag*)(wbu + q * 16 * WRSU); ao[0][q] = *(lds_frag*)(wbu + CLSB + q * 16 * WRSU); }
#pragma unroll
                    for (int r = 0; r < 5; ++r) bu[0][r] = *(lds_frag*)(halo + boff_up[0] + r * ROWB);
#pragma unroll
                    for (int s_ = 0; s_ < KSUP; ++s_) {
                        if (s_ + 1 < KSUP) {
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                ae[(s_ + 1) & 1][q] = *(lds_frag*)(wbu + q * 16 * WRSU + (s_ + 1) * 64);
                                ao[(s_ + 1) & 1][q] = *(lds_frag*)(wbu + CLSB + q * 16 * WRSU + (s_ + 1) * 64);
                            }
#pragma unroll
                            for (int r = 0; r < 5; ++r) bu[(s_ + 1) & 1][r] = *(lds_frag*)(halo + boff_up[s_ + 1] + r * ROWB);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < NP; ++q)
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[q][j] = VG_MFMA16((j & 1) ? ao[s_ & 1][q] : ae[s_ & 1][q], bu[s_ & 1][(j + 1) >> 1], acc[q][j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    continue;
                }
            }
            // ---- MFMA loop: 14 K-steps x 8 sub-tiles x NP panels, every address an immediate, the fragments of the next K-step in flight
            const char* wb = wlds + wbase + (wdma ? (wuse & 1) * WBUF : 0);
            constexpr int PD = VG_THIN_PD, NB = PD + 1;                    // K-steps of fragments in flight ahead of the MFMAs
            if constexpr (NP == 1) {
                bf16x8 a[NB][NP], b[NB][8];
#pragma unroll
                for (int u = 0; u < PD; ++u) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) a[u][q] = *(lds_frag*)(wb + q * 16 * WRS + u * 64);
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[u][j] = *(lds_frag*)(halo + boff[u] + j * ROWB);
                }
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    if (s + PD < KSTEPS) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[(s + PD) % NB][q] = *(lds_frag*)(wb + q * 16 * WRS + (s + PD) * 64);
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[(s + PD) % NB][j] = *(lds_frag*)(halo + boff[s + PD] + j * ROWB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NP; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[q][j] = VG_MFMA16(a[s % NB][q], b[s % NB][j], acc[q][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // two panels: a K-step is 16 MFMAs (256 cycles), so the B fragments are double-buffered per HALF step (4 sub-tiles:
                // 8 MFMAs cover the LDS round trip of the next four) -- 32 registers of fragments instead of 64; the accumulators take 64
                bf16x8 a[2][NP], b[2][4];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[0][q] = *(lds_frag*)(wb + q * 16 * WRS);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[0][j] = *(lds_frag*)(halo + boff[0] + j * ROWB);
#pragma unroll
                for (int hs = 0; hs < 2 * KSTEPS; ++hs) {
                    const int s = hs >> 1, h = hs & 1;
                    if (hs + 1 < 2 * KSTEPS) {
                        const int s2 = (hs + 1) >> 1, h2 = (hs + 1) & 1;
                        if (h2 == 0) {
#pragma unroll
                            for (int q = 0; q < NP; ++q) a[s2 & 1][q] = *(lds_frag*)(wb + q * 16 * WRS + s2 * 64);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) b[(hs + 1) & 1][j] = *(lds_frag*)(halo + boff[s2] + (4 * h2 + j) * ROWB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NP; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q][4 * h + j] = VG_MFMA16(a[s & 1][q], b[hs & 1][j], acc[q][4 * h + j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue: wave = D-plane `wave` of the tile, sub-tile j = H row j, lane li = voxel W, lanes own 4 channels;
        //      pairs of sub-tiles are exchanged across the 16-lane rows so that every lane stores 8 channels = 16 bytes.
        //      Tiles that lie wholly inside the output (all of them for the forward layers, ~80 % for the data gradients
        //      on the padded grid) take the predicate-free instance.
        const int od = od0 + wave, ow = ow0 + li;
        const size_t rowpitch = (size_t)p.BW * p.Cout;                                       // elements per output H row (ostr == 1)
        const size_t obase = (((size_t)(n * p.BD + od + p.ood) * p.BH + oh0 + p.ooh) * p.BW + ow + p.oow) * p.Cout;
        const bool full = od0 + TD <= p.OD && oh0 + TH <= p.OH && ow0 + TW <= p.OW;
        auto epilogue = [&](auto masked_tag, const int pn16) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const bool dw_ok = !MASKED || (od < p.OD && ow < p.OW);
            const int nrow = MASKED ? p.OH - oh0 : TH;                                        // valid H rows of this tile
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
            T* const optr = (T*)p.out + obase + cst + 16 * q + pn16;
            const T* const rptr = RES ? (const T*)p.res + obase + co0 + 16 * q : nullptr;
            // (vg_conv_desc::res_c1: the residual is a single-channel fp32 volume -- one value per voxel row, kept in the first word of rr[j])
            const float* const r1ptr = RES ? (const float*)p.res + (((size_t)(n * p.BD + od + p.ood) * p.BH + oh0 + p.ooh) * p.BW + ow + p.oow) : nullptr;
            f32x2 eb[2], ers[2], erb[2];
            if constexpr (NP == 1) { eb[0] = e_b[0][0]; eb[1] = e_b[0][1]; if (RES) { ers[0] = e_rs[0][0]; ers[1] = e_rs[0][1]; erb[0] = e_rb[0][0]; erb[1] = e_rb[0][1]; } }
            else {              // (two panels: fetched here, from LDS -- held across the MFMA loop they cost 24 registers the accumulators need;
                                //  fetched from memory every tile waited an L2 round trip for them)
                const float* eq = ebs + 4 * kg + 16 * q;
                if (BIAS) { const f32x4 t = *(const f32x4*)eq; eb[0] = (f32x2){t[0], t[1]}; eb[1] = (f32x2){t[2], t[3]}; }
                if (RES) {
                    const f32x4 t = *(const f32x4*)(eq + 16 * NP), u = *(const f32x4*)(eq + 32 * NP);
                    ers[0] = (f32x2){t[0], t[1]}; ers[1] = (f32x2){t[2], t[3]}; erb[0] = (f32x2){u[0], u[1]}; erb[1] = (f32x2){u[2], u[3]};
                }
            }
            bf16x4 bx[BSTAT ? 8 : 1];
            f32x2 b_sc[2], b_sf[2], b_rs[2], b_nm[2];
            if (BSTAT) {
                constexpr int CW = 16 * NP * PL;
                const float* bq = bsc + 4 * kg + 16 * q + pn16;
                const f32x4 c_sc = *(const f32x4*)bq, c_sf = *(const f32x4*)(bq + CW), c_rs = *(const f32x4*)(bq + 2 * CW), c_nm = *(const f32x4*)(bq + 3 * CW);
                b_sc[0] = (f32x2){c_sc[0], c_sc[1]}; b_sc[1] = (f32x2){c_sc[2], c_sc[3]}; b_sf[0] = (f32x2){c_sf[0], c_sf[1]}; b_sf[1] = (f32x2){c_sf[2], c_sf[3]};
                b_rs[0] = (f32x2){c_rs[0], c_rs[1]}; b_rs[1] = (f32x2){c_rs[2], c_rs[3]};
                b_nm[0] = (f32x2){c_nm[0], c_nm[1]}; b_nm[1] = (f32x2){c_nm[2], c_nm[3]};
                // padded output coordinate -> interior coordinate -> transpose of the reflection pad (-1 -> 1, n -> n-2); rows of a
                // masked tile that lie outside are clamped (their contribution is zeroed below).  All 8 loads are issued up front.
                auto fold = [&](int qq, int nn) { int i = qq - p.bs_pad; i = i < 0 ? -i : i; i = i >= nn ? 2 * nn - 2 - i : i; return min(max(i, 0), nn - 1); };
                const int XH = p.bs_H >> b_sh[q], XW = p.bs_W >> b_sh[q];
                const int id = fold(od + p.ood, p.bs_D) >> b_sh[q], iw = fold(ow + p.oow, p.bs_W) >> b_sh[q];
                const T* xcol = b_x[q] + ((size_t)id * XH * XW + iw) * b_cs[q];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (NP == 1) bx[j] = bxp[j];
                    else {
                        const int ih = fold(oh0 + j + p.ooh, p.bs_H) >> b_sh[q];
                        bx[j] = *(const __attribute__((address_space(1))) bf16x4*)(uintptr_t)(xcol + (size_t)ih * XW * b_cs[q]);
                    }
                }
            }
            // RES: all eight rows of the residual are requested up front.  (Loaded where they are used, the compiler could not move them above
            // the stores of the rows before -- p.out and p.res may alias for all it knows -- and every row pair waited its own round trip.)
            // (Two panels: four rows at a time -- eight more registers spill there.)
            constexpr int RRN = NP == 1 ? 8 : 4;
            bf16x4 rr[RES ? 8 : 1];
#pragma unroll
            for (int jp = 0; jp < 8; jp += 2) {
                if constexpr (RES) {
                    if (jp % RRN == 0) {
#pragma unroll
                        for (int j = jp; j < jp + RRN; ++j) {
                            const bool ok = !MASKED || (dw_ok && j < nrow);
                            if (NP == 1 && p.res1) {                    // (one-panel instances only: the 16-channel stem; the two-panel epilogue has no register to spare)
                                const float xv = ld_global(ok ? r1ptr + (size_t)j * p.BW : r1ptr);
                                rr[j] = __builtin_bit_cast(bf16x4, (u32x2){__float_as_uint(xv), 0u});
                            } else
                            rr[j] = *(const __attribute__((address_space(1))) bf16x4*)(uintptr_t)(ok ? rptr + j * rowpitch : rptr);
                        }
                    }
                }
                bf16x4 pk[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = jp + e;
                    const bool ok = !MASKED || (dw_ok && j < nrow);
                    f32x2 v0 = {acc[q][j][0], acc[q][j][1]}, v1 = {acc[q][j][2], acc[q][j][3]};
                    if (BIAS) { v0 += eb[0]; v1 += eb[1]; }
                    if (RES) {
                        const bf16x4 r = rr[RES ? j : 0];
                        f32x2 r0 = {bf2f((bf16_t)r[0]), bf2f((bf16_t)r[1])}, r1 = {bf2f((bf16_t)r[2]), bf2f((bf16_t)r[3])};
                        if (NP == 1 && p.res1) { const float xv = __uint_as_float(__builtin_bit_cast(u32x2, r)[0]); r0 = (f32x2){xv, xv}; r1 = r0; }
                        v0 += r0 * ers[0] + erb[0]; v1 += r1 * ers[1] + erb[1];
                    }
                    pk[e] = (bf16x4){(short)f2bf(v0[0]), (short)f2bf(v0[1]), (short)f2bf(v1[0]), (short)f2bf(v1[1])};
                    if (STATS) {
                        f32x2 q0 = {bf2f((bf16_t)pk[e][0]), bf2f((bf16_t)pk[e][1])}, q1 = {bf2f((bf16_t)pk[e][2]), bf2f((bf16_t)pk[e][3])};
                        if (MASKED && !ok) { q0 = (f32x2){0.f, 0.f}; q1 = q0; }
                        f32x2 a0 = {s1[q][0], s1[q][1]}, a1 = {s1[q][2], s1[q][3]}, c0 = {s2[q][0], s2[q][1]}, c1 = {s2[q][2], s2[q][3]};
                        a0 += q0; a1 += q1; c0 += q0 * q0; c1 += q1 * q1;
                        s1[q][0] = a0[0]; s1[q][1] = a0[1]; s1[q][2] = a1[0]; s1[q][3] = a1[1]; s2[q][0] = c0[0]; s2[q][1] = c0[1]; s2[q][2] = c1[0]; s2[q][3] = c1[1];
                    }
                    if (BSTAT) {
                        f32x2 q0 = {bf2f((bf16_t)pk[e][0]), bf2f((bf16_t)pk[e][1])}, q1 = {bf2f((bf16_t)pk[e][2]), bf2f((bf16_t)pk[e][3])};
                        if (MASKED && !ok) { q0 = (f32x2){0.f, 0.f}; q1 = q0; }
                        const bf16x4 xr = bx[BSTAT ? j : 0];
                        const f32x2 x0 = {bf2f((bf16_t)xr[0]), bf2f((bf16_t)xr[1])}, x1 = {bf2f((bf16_t)xr[2]), bf2f((bf16_t)xr[3])};
                        const f32x2 pre0 = x0 * b_sc[0] + b_sf[0], pre1 = x1 * b_sc[1] + b_sf[1];
                        const f32x2 d0 = {pre0[0] > 0.f ? 1.f : b_slope, pre0[1] > 0.f ? 1.f : b_slope};     // TP: the activation gradient uses pre > 0
                        const f32x2 d1 = {pre1[0] > 0.f ? 1.f : b_slope, pre1[1] > 0.f ? 1.f : b_slope};
                        const f32x2 dn0 = q0 * d0, dn1 = q1 * d1;                                          // (no dropout multiplier on this path)
                        const f32x2 xh0 = x0 * b_rs[0] + b_nm[0], xh1 = x1 * b_rs[1] + b_nm[1];
                        f32x2 a0 = {s1[q][0], s1[q][1]}, a1 = {s1[q][2], s1[q][3]}, c0 = {s2[q][0], s2[q][1]}, c1 = {s2[q][2], s2[q][3]};
                        a0 += dn0; a1 += dn1; c0 += dn0 * xh0; c1 += dn1 * xh1;
                        s1[q][0] = a0[0]; s1[q][1] = a0[1]; s1[q][2] = a1[0]; s1[q][3] = a1[1]; s2[q][0] = c0[0]; s2[q][1] = c0[1]; s2[q][2] = c1[0]; s2[q][3] = c1[1];
                    }
                }
                // rows (16-lane groups) 1,3 of pk[0] <-> rows 0,2 of pk[1]: even rows end with sub-tile jp channels [4kg..4kg+7],
                // odd rows with sub-tile jp+1 channels [4(kg-1)..4kg+3]
                const u32x2 wa = __builtin_bit_cast(u32x2, pk[0]), wb2 = __builtin_bit_cast(u32x2, pk[1]);
                const u32x2 x0 = __builtin_amdgcn_permlane16_swap(wa[0], wb2[0], false, false);
                const u32x2 x1 = __builtin_amdgcn_permlane16_swap(wa[1], wb2[1], false, false);
                const u32x4 outv = {x0[0], x1[0], x0[1], x1[1]};
                const int j = jp + jodd;
                if (!MASKED || (dw_ok && j < nrow)) *(u32x4*)(optr + j * rowpitch) = outv;
                if constexpr (BSTAT && PL > 1) { __builtin_amdgcn_sched_barrier(0); bxn[jp] = bx_row(jp); bxn[jp + 1] = bx_row(jp + 1); __builtin_amdgcn_sched_barrier(0); }   // the next panel's rows (bx_prepare)
            }
            }
        };
        if constexpr (PL == 1) { if (full) epilogue(std::false_type{}, 0); else epilogue(std::true_type{}, 0); }
        else {
            if (it) lds_only_barrier();                                    // the previous tile's last MFMA loop is done with the halo image
            // The pre-norm rows of the NEXT panel (or of the next tile's first) are requested inside the epilogue, row pair by row pair, into
            // bxn -- a whole MFMA loop ahead of their use -- and move to bxp behind the vmcnt(0) that follows that loop.  (Requested straight
            // into bxp, the loop-carried rows got fresh registers and were copied home at the END of the epilogue behind an s_waitcnt
            // vmcnt(0): the whole round trip in the open, 163 of 399 us.  Written in place by an asm load, the compiler copied registers
            // with the load still pending.  With the three panels unrolled -- no loop-carried rows at all -- 83 registers spilled.)
            if constexpr (BSTAT) {
                if (it == 0) {
                    bx_prepare(0, od0, oh0, ow0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) bxn[j] = bx_row(j);
                }
            }
            if (!(p.dbg & 16) || it == 0) {
            thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, 0, tid, raw);
            thin_commit<MODE>(g, halo, scs, utab, tabat(od0, oh0, ow0), 0, tid, raw);
            }
#pragma unroll 1
            for (int pn = 0; pn < PL; ++pn) {
                const int use = it * PL + pn;
                // every wave has waited for its pieces of this use's panel (below / ahead of the first tile) and is done with the other buffer
                if (use == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_only_barrier();
                if (pn + 1 < PL || more) dma_weights(pn + 1 < PL ? pn + 1 : 0, (use + 1) & 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(p.dbg & 4)) mfma_panel(wlds + (use & 1) * WBUF + wbase);
                __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), as an instruction the compiler's own counting sees: the next use's
                asm volatile("" ::: "memory");                             // pieces and this panel's pre-norm rows have landed
                if constexpr (BSTAT) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_mov_b64 %0, %1" : "=&v"(bxp[j]) : "v"(bxn[j]));   // a copy the compiler cannot fold: bxn's registers must be free for the requests below
                    // (the last panel of the last tile requests its own rows again: no branch around the requests)
                    if (pn + 1 < PL) bx_prepare(16 * (pn + 1), od0, oh0, ow0); else if (more) bx_prepare(0, ti_d * TD, ti_h * TH, ti_w * TW);
                }
                if (full) epilogue(std::false_type{}, 16 * pn); else epilogue(std::true_type{}, 16 * pn);
                if constexpr (BSTAT) {                                      // the running sums of the next panel move into place (back where they were after PL panels)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = s1[0][r], b = s2[0][r];
#pragma unroll
                        for (int q = 0; q + 1 < PL; ++q) { s1[q][r] = s1[q + 1][r]; s2[q][r] = s2[q + 1][r]; }
                        s1[PL - 1][r] = a; s2[PL - 1][r] = b;
                    }
                }
            }
        }
    }
    if constexpr (PL > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (a workgroup without tiles still has its first panel in flight)
    if ((STATS || BSTAT) && p.sums) {
#pragma unroll
        for (int q = 0; q < NP * PL; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[q][r], b = s2[q][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (li == 0) { atomicAdd(&stat[(16 * q + 4 * kg + r) * 2], a); atomicAdd(&stat[(16 * q + 4 * kg + r) * 2 + 1], b); }
            }
        __syncthreads();
        if (tid < 32 * NP * PL) {
            const int co = cop + (tid >> 1);
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            if (co < p.Cout) atomicAdd(&p.sums[(((size_t)stripe * gridDim.z + n) * p.Cout + co) * 2 + (tid & 1)], stat[tid]);
        }
        if constexpr (STATS && (NP == 2 || !RES)) { if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.z, p.Cout, gridDim.x * gridDim.y * gridDim.z, (int*)stat); }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int vg_conv_thin_lds_bytes(const GatherIn& g, int np, int pl, bool up, bool wdma) {
    const int AL = (g.tiles_h * TH + 2) + (g.tiles_w * TW + 2) + (g.tiles_d * TD + 2);          // axis tables of the whole grid, two sources
    int head = HALO + (32 + 32 * np * pl + 64 * np * pl + (np > 1 ? 48 * np : 0) + 32) * 4 + (2 * HH * HW * 2 + 2 * AL) * 4;
    if (pl > 1 || wdma) return head + 1024 + 2 * 15 * 1024;                                     // two LDS-DMA buffers of one 16-row panel
    if (up) {           // half-resolution axis tables + column table; the weight area holds the four class panels of an upsampled chunk
        head += (g.tiles_h * (TH / 2) + 2 + g.tiles_d * (TD / 2) + 2 + 2 * UPCOLS + 16) * 4;
        return head + 16 + std::max(16 * np * (KCPAD * 2 + 16), 4 * 16 * np * WRSU);
    }
    return head + 16 + 16 * np * (KCPAD * 2 + 16);
}

// Does this launch have the one shape the specialist serves?  (g from fill_gather for a 512-voxel tile; np: 16-channel panels per workgroup)
bool vg_conv_thin_ok(const vg_conv_desc* d, const GatherIn& g, const ConvOut& k, const ConvCls& q, int np) {
    if (!vg_tune("CONV_THIN", 1)) return false;
    if (d->f32 || d->noise || q.ncls != 1 || d->istr != 1 || d->ostr != 1 || d->CK != 16 || d->ntaps != 27 || d->wpack) return false;
    if ((d->c_src0 + d->c_src1) % 16 || (d->Cout % (16 * np)) || d->tanh_out || d->accumulate || d->out_f32) return false;
    if (g.lean != VG_STAGE_PLAIN && g.lean != VG_STAGE_RELU) return false;
    if (!g.planar || g.HW != HW || g.HH != HH || g.HD != HD || g.HWp != HW || g.HHp != HH || g.DS != DSB || g.PSB != PSB) return false;
    if ((1 << g.twl) != TW || (1 << g.thl) != TH || (1 << g.tdl) != TD) return false;
    if (d->res && (d->bias == nullptr)) return false;                  // (instantiated combinations only)
    if (d->res_c1 && np != 1) return false;                           // (the single-channel residual lives in the one-panel epilogue)
    if (k.Ktot != k.nchunks * KCPAD) return false;
    for (int i = 0; i < 27; ++i)                                       // a full 3x3x3 stencil (any order)
        if (d->tap_d[i] - g.tmin_d > 2 || d->tap_h[i] - g.tmin_h > 2 || d->tap_w[i] - g.tmin_w > 2) return false;
    if (np > 1) {
        // two panels per workgroup halve the workgroup count: not below ~100 workgroups (threshold swept on BASELINE configs 2-4: 384 / 192 /
        // 96 -> 64^3 batch 2: 9.53 / 9.47 / 9.40 ms per step, 128x128x64 batch 2: 21.32 / 21.20 / 21.22, 128^3: 19.42 / 19.52 / 19.35)
        const long wgs = (long)g.tiles_d * g.tiles_h * g.tiles_w * (d->Cout / (16 * np)) * d->N;
        if (!vg_tune("CONV_THIN2", 1) || wgs < vg_tune("CONV_THIN2_MINWG", 96)) return false;
    }
    return true;
}

template <int MODE, bool BIAS, bool RES, bool STATS, bool BSTAT, int NP, int PL = 1, bool UP = false>
static int launch_thin_np(const GatherIn& g, const ConvOut& k, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_thin_kernel<MODE, BIAS, RES, STATS, BSTAT, NP, PL, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    int per_cu = 2;
    if (lds > 0 && VG_LDS_LIMIT / lds < per_cu) per_cu = VG_LDS_LIMIT / lds;
    if (per_cu < 1) per_cu = 1;
    const int tiles = g.tiles_d * g.tiles_h * g.tiles_w;
    const int ny = k.Cout / (16 * NP * PL);
    const int wg = vg_tune("CONV_THIN_WGS", 0) > 0 ? vg_tune("CONV_THIN_WGS", 0) : 256 * per_cu;
    int bx = wg / (ny * g.N); if (bx < 1) bx = 1; if (bx > tiles) bx = tiles;
    ConvOut k2 = k; k2.xw = 0; k2.dbg = PL > 1 ? vg_ablate_knob("THIN_DBG") : 0;
    if (k2.dbg < 0) return VG_EINVAL;
    if (vg_tune("CONV_THIN_XCD", 1) && bx >= 16 && tiles >= 4 * bx) { bx &= ~7; k2.xw = 1; }
    // bs1 / bs2: IN-backward statistics in the epilogue, of a plain / a virtually concatenated (half-resolution + skip) pre-norm tensor
    char name[96];
    snprintf(name, sizeof name, "%s<m%%d,b%%d,r%%d,s%%d%s%s%s>|walk%%d|ch%%d", NP == 2 ? "conv_thin2" : "conv_thin", BSTAT ? (k.bs_x1 ? ",bs2" : ",bs1") : "", PL > 1 ? ",pl" : "", UP ? ",up" : "");
    if (vg_dry(name, MODE, (int)BIAS, (int)RES, (int)STATS, tiles > bx ? 1 : 0, k.nchunks > 1 ? (k.wdma ? 2 : 1) : 0)) return VG_OK;      // ch2: chunk panels by LDS-DMA
    // the two-panel instances (32-channel layers) finalise the InstanceNorm statistics of their output in the launch (last workgroup)
    const bool fin_here = STATS && k2.fin.ticket && k2.sums && ((NP == 2 && vg_tune("CONV_THIN2_FIN", 1)) || (NP == 1 && !RES && vg_tune("CONV_THIN1_FIN", 1)));
    if (!fin_here) k2.fin.ticket = nullptr;
    if constexpr (!UP) k2.nup = 0;
    hipLaunchKernelGGL((conv_thin_kernel<MODE, BIAS, RES, STATS, BSTAT, NP, PL, UP>), dim3(bx, ny, g.N), dim3(256), lds, s, g, k2);
    if (fin_here) vg_fin_done = true;
    // (no finalisation tail in the one-panel instances: with it the 16 -> 16 residual instance ran 17 % slower even when the tail was not taken --
    // code placement, not registers: the allocation was unchanged -- and the sliding-window inference lost 1.5 ms per volume; vg_conv3d
    // runs the small finalisation kernel behind these launches instead)
    return vg_check_launch();
}
template <int MODE, bool BIAS, bool RES, bool STATS, bool BSTAT = false>
static int launch_thin(const GatherIn& g, const ConvOut& k, int np, hipStream_t s) {
    if constexpr (BSTAT && MODE == VG_STAGE_PLAIN) {
        // 16 -> 48 (dec0.cb1's data gradient): the three output panels looped over one staged halo
        const int lds3 = vg_conv_thin_lds_bytes(g, 1, 3);
        if (np == 1 && k.Cout == 48 && k.nchunks == 1 && 2 * lds3 <= VG_LDS_LIMIT && vg_tune("CONV_THIN_PL", 1))
            return launch_thin_np<MODE, BIAS, RES, STATS, BSTAT, 1, 3>(g, k, lds3, s);
    }
    if constexpr (MODE == VG_STAGE_RELU && BIAS && !RES && !BSTAT) {
        // the decoder's first convolution: its upsampled chunks contracted over the half-resolution image (conv_thin_kernel<..., UP>)
        if (k.nup > 0 && k.nup < k.nchunks && vg_tune("CONV_THIN_UP", 0)) {        // (off by default: see ops.ConvLayer.enable_up for the measurement)
            const int ldsu = vg_conv_thin_lds_bytes(g, np, 1, true);
            if (2 * ldsu <= VG_LDS_LIMIT) {
                // (one-panel instance only: the two-panel form needs 93 KB of LDS per workgroup and spills 60-90 registers)
                if (np == 1) return launch_thin_np<MODE, BIAS, RES, STATS, BSTAT, 1, 1, true>(g, k, ldsu, s);
            }
        }
    }
    if constexpr (MODE == VG_STAGE_RELU && !BSTAT) {
        // one-panel forward with several channel chunks (dec0.cb1: 48 -> 16): the weight panels double-buffered by LDS-DMA
        const int ldsw = vg_conv_thin_lds_bytes(g, 1, 1, false, true);
        if (np == 1 && k.nchunks > 1 && 2 * ldsw <= VG_LDS_LIMIT && vg_tune("CONV_THIN_WDMA", 1)) {
            ConvOut kw = k; kw.wdma = 1;
            return launch_thin_np<MODE, BIAS, RES, STATS, BSTAT, 1>(g, kw, ldsw, s);
        }
    }
    const int lds = vg_conv_thin_lds_bytes(g, np);
    if (lds > VG_LDS_LIMIT) return VG_ELDS;
    if constexpr (!BSTAT) { if (np == 2) return launch_thin_np<MODE, BIAS, RES, STATS, BSTAT, 2>(g, k, lds, s); }
    return np == 1 ? launch_thin_np<MODE, BIAS, RES, STATS, BSTAT, 1>(g, k, lds, s) : VG_EINVAL;
}

int vg_launch_conv_thin(const GatherIn& g, const ConvOut& k, int np, hipStream_t s, float* red, bool& did_stats) {
    const bool st = k.sums != nullptr;
    if (g.lean == VG_STAGE_PLAIN) {
        // data gradient (no bias / residual / statistics) and raw-source forward convolutions.  (The two-panel instance with the
        // IN-backward statistics in its epilogue spills 86 registers: those launches leave the statistics to the pass behind them.)
        if (np == 1 && red && k.bs_x0 && !k.bs_ml && k.bs_sc && k.bias == nullptr && k.res == nullptr && !st && (k.bs_c0 % 16) == 0 && (k.bs_c0 == k.Cout || k.bs_x1)
            && k.OD == k.BD && k.OH == k.BH && k.OW == k.BW && !k.ood && !k.ooh && !k.oow && vg_tune("CONV_BSTAT", 1)) {
            ConvOut k2 = k; k2.sums = red;                 // the IN-backward statistics take the place of the forward ones
            const int rc = launch_thin<VG_STAGE_PLAIN, false, false, false, true>(g, k2, np, s);
            did_stats = rc == VG_OK && !vg_dry_on();
            return rc;
        }
        if (k.bias == nullptr && k.res == nullptr && !st) return launch_thin<VG_STAGE_PLAIN, false, false, false>(g, k, np, s);
        if (k.bias != nullptr && k.res == nullptr) return st ? launch_thin<VG_STAGE_PLAIN, true, false, true>(g, k, np, s)
                                                             : launch_thin<VG_STAGE_PLAIN, true, false, false>(g, k, np, s);
        return 1;
    }
    if (k.bias == nullptr) return 1;
    if (k.res != nullptr) return st ? launch_thin<VG_STAGE_RELU, true, true, true>(g, k, np, s) : launch_thin<VG_STAGE_RELU, true, true, false>(g, k, np, s);
    return st ? launch_thin<VG_STAGE_RELU, true, false, true>(g, k, np, s) : launch_thin<VG_STAGE_RELU, true, false, false>(g, k, np, s);
}

// vg_pack_up_weights (include/vangan_hip.h): the class panels of the collapsed upsampled chunks.  One thread per packed element.
__global__ void pack_up_weights_kernel(const float* __restrict__ w, int Cin, int Cout, int nup, bf16_t* __restrict__ out) {
    const int total = nup * 4 * Cout * KUP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int t = i;
        const int kk = t % KUP; t /= KUP;
        const int co = t % Cout; t /= Cout;
        const int cls = t & 3, chunk = t >> 2;
        const int t12 = kk >> 4, cl = kk & 15, cdh = t12 / 3, cw = t12 - cdh * 3;
        const int cd = cdh >> 1, ch = cdh & 1, pd = cls >> 1, ph = cls & 1;
        // original taps (index 0..2 = offset -1..+1) summed into collapsed tap c for parity p: p = 0: {0}, {1, 2}; p = 1: {0, 1}, {2}
        const int d0 = pd == 0 ? (cd == 0 ? 0 : 1) : (cd == 0 ? 0 : 2), d1 = pd == 0 ? (cd == 0 ? 0 : 2) : (cd == 0 ? 1 : 2);
        const int h0 = ph == 0 ? (ch == 0 ? 0 : 1) : (ch == 0 ? 0 : 2), h1 = ph == 0 ? (ch == 0 ? 0 : 2) : (ch == 0 ? 1 : 2);
        const int ci = chunk * 16 + cl;
        float a = 0.f;
        for (int a_ = d0; a_ <= d1; ++a_)
            for (int b_ = h0; b_ <= h1; ++b_) a += w[((size_t)((a_ * 3 + b_) * 3 + cw) * Cin + ci) * Cout + co];
        out[i] = f2bf(a);
    }
}
extern "C" int vg_pack_up_weights(const float* w, int Cin, int Cout, int c_up, void* out, vg_stream_t stream) {
    vg_begin();
    if (!w || !out || Cin < 16 || Cout < 1 || c_up < 16 || (c_up % 16) || c_up > Cin) return VG_EINVAL;
    const int nup = c_up / 16, total = nup * 4 * Cout * KUP;
    hipLaunchKernelGGL(pack_up_weights_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, nup, (bf16_t*)out);
    return vg_check_launch();
}

// ================================================================================================================================
// Weight gradient of the same layers (3x3x3, stride 1, 16-channel chunks, 16 output channels): wgrad_thin_kernel
// ================================================================================================================================
// dW[tap][ci][co] = sum over voxels X[voxel + tap][ci] * dY[voxel][co] with X the operand exactly as the forward kernel stages it (same
// routine: thin_issue / thin_commit, same transform, same single rounding).  wgrad_dma_kernel<R,1,DIRECT> gives each of its 8 waves a few
// (tap, plane) ROWS and walks all voxels of a tile with them: 10 LDS fragment reads per 4 MFMAs, every K-step a dependent LDS round trip
// (290 cycles per K-step for 64 cycles of MFMA).  Here the waves split the VOXELS (wave = D-plane of the tile) and every wave
// holds the WHOLE slab -- 27 taps x 16 ci x 16 co = 27 accumulator tiles (+ 1 for the bias gradient) -- so a K-step of 32 voxels is 2
// fragment reads of dY and 27 x 2 of X feeding 28 independent MFMAs: nothing in the loop waits for anything but LDS bandwidth.
//   * both operands are needed voxel-major per lane while X lies [channel group][voxel][8 ch] and dY [voxel][16 ch] in LDS:
//     ds_read_b64_tr_b16 hands a 16-lane group a 4 (voxels) x 16 (channels) block transposed; each lane supplies its own row address, so
//     the two channel-group planes of the halo image are one block, and the tap is an immediate offset (a DSB + b ROWB + c UNIT) on one
//     base register -- no address arithmetic in the loop;
//   * bank conflicts: the plane stride carries a 64-byte skew (fill_gather's skew) so that the two planes of the same 4 voxels take
//     different banks; the dY rows are padded by 128 bytes per 8 voxels for the same reason;
//   * a workgroup keeps its 28 accumulator tiles over all its tiles, the four waves' copies are added in wave order through LDS, and the
//     slab goes to a partial buffer that wgrad_thin_reduce_kernel sums in a fixed order (16 loads of 16 bytes in flight per lane).
struct WgThin { const void* dy; float* part; float* dw; float* db; int Cin, nslab, dbg; unsigned* tickets; int ntickets, xw; };
namespace {
constexpr int WT_DYROW = TW * 32 + 128;                 // 640: 16 voxels x 32 bytes + 128 bytes behind the first 8
constexpr int WT_DYB = TD * TH * WT_DYROW;              // 20480
constexpr int WT_SLAB = 28 * 256;                       // floats: [tap 0..26 | bias][ci 16][co 16]
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_thin_kernel(const GatherIn g, const WgThin p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.y, n = blockIdx.z;
    const int PSBs = g.PSB;                                               // plane stride incl. the skew
    char* halo = smem;
    float* scs = (float*)(smem + 2 * PSBs);
    int* utab = (int*)(scs + 32);
    constexpr int NCOLS = HH * HW * 2;
    int* xtab = utab + 2 * NCOLS;
    const int NHt = g.tiles_h * TH + 2, NWt = g.tiles_w * TW + 2, NDt = g.tiles_d * TD + 2;
    char* dyt = (char*)(xtab + 2 * (NHt + NWt + NDt));
    dyt = (char*)(((size_t)dyt + 15) & ~(size_t)15);
    build_column_table(g, utab, tid);
    stage_scale_shift(g, scs, n, chunk, tid);
    thin_axis_tables(g, xtab, NHt, NWt, NDt, tid);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < p.ntickets) p.tickets[tid] = 0u;      // the slab pass's tickets (it runs behind this launch)
    __syncthreads();                                                         // the tables are read by the first tile's loads
    auto tabat = [&](int od0_, int oh0_, int ow0_) { return ThinTab{xtab, NHt + NWt + NDt, NHt, NWt, od0_, oh0_, ow0_}; };
    // transposed-read addresses: lane 4 q + pp of 16-lane group gq supplies voxel row q (+ 4 for the second read), channels 4 pp .. + 3;
    // the K-step's 32 voxels: tile row y = 2 s + (gq >> 1), column w = 8 (gq & 1) + q (+ 4)
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int wq = 8 * (gq & 1) + q, yq = gq >> 1;
    const int abase = (pp >> 1) * PSBs + ((wave * HH + yq) * HW + wq) * UNIT + (pp & 1) * 8;
    const int bbase = (wave * TH + yq) * WT_DYROW + wq * 32 + (wq >> 3) * 128 + pp * 8;
    f32x4 acc[28];
#pragma unroll
    for (int t = 0; t < 28; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
    { const short o = (short)f2bf(1.f); ones = (bf16x8){o, o, o, o, o, o, o, o}; }
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    const T* dyn = (const T*)p.dy + (size_t)n * g.D * g.H * g.W * 16;
    // the loads of a tile (9 + 4 units of 16 bytes per thread) are issued ahead of the previous tile's K loop and land under it
    Raw8<T> raw[3][HD / 2];
    f32x4 dv[4];
    auto origin = [&](int tile, int& od0, int& oh0, int& ow0) {
        const int ti_w = tile % g.tiles_w, t2 = tile / g.tiles_w, ti_h = t2 % g.tiles_h, ti_d = t2 / g.tiles_h;
        od0 = ti_d * TD; oh0 = ti_h * TH; ow0 = ti_w * TW;
    };
    auto issue = [&](int od0, int oh0, int ow0) {
        thin_issue<MODE>(g, utab, tabat(od0, oh0, ow0), n, chunk, tid, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                       // dY of the tile: 1024 units of 16 bytes (out of range: zeros)
            const int u = tid + 256 * i, vox = u >> 1, w = vox & 15, y = (vox >> 4) & 7, z = vox >> 7;
            const bool ok = od0 + z < g.D && oh0 + y < g.H && ow0 + w < g.W;
            const T* src = dyn + (((size_t)min(od0 + z, g.D - 1) * g.H + min(oh0 + y, g.H - 1)) * g.W + min(ow0 + w, g.W - 1)) * 16 + (u & 1) * 8;
            dv[i] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)src;
            if (!ok) dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // tile walk as the forward kernel's (p.xw: gridDim.x a multiple of 8): the workgroups that share an XCD (blockIdx.x & 7) walk ONE contiguous
    // eighth of the tile sequence side by side, so the halo voxels neighbouring tiles share (2.1 x the tile) are found in that XCD's L2
    int t0 = blockIdx.x, tstep = gridDim.x, tend = tiles_per_n;
    if (p.xw) {
        const int sl = (tiles_per_n + 7) >> 3, xcd = blockIdx.x & 7;
        t0 = xcd * sl + ((int)blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = min(tiles_per_n, (xcd + 1) * sl);
    }
    int od0, oh0, ow0;
    if (t0 < tend) { origin(t0, od0, oh0, ow0); issue(od0, oh0, ow0); }
    unsigned long long c_wait0 = 0, c_commit = 0, c_wait1 = 0, c_k = 0, t_a = 0, t_b = 0;
    for (int tile = t0; tile < tend; tile += tstep) {
        if (p.dbg & 8) t_a = __builtin_amdgcn_s_memtime();
        __syncthreads();                                                     // the previous tile's fragment reads are done
        if (p.dbg & 8) { t_b = __builtin_amdgcn_s_memtime(); c_wait0 += t_b - t_a; }
        if (!(p.dbg & 2)) thin_commit<MODE>(g, halo, scs, utab, tabat(od0, oh0, ow0), chunk, tid, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 256 * i, vox = u >> 1, w = vox & 15;
            *(f32x4*)(dyt + (vox >> 4) * WT_DYROW + w * 32 + (w >> 3) * 128 + (u & 1) * 16) = dv[i];
        }
        if (p.dbg & 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_a = __builtin_amdgcn_s_memtime(); c_commit += t_a - t_b; }
        __syncthreads();
        if (p.dbg & 8) { t_b = __builtin_amdgcn_s_memtime(); c_wait1 += t_b - t_a; }
        if (!(p.dbg & 4) && tile + tstep < tend) { origin(tile + tstep, od0, oh0, ow0); issue(od0, oh0, ow0); }
        if (p.dbg & 1) continue;
        // the tile's 4 x 27 (K-step, tap) products in 18 groups of six; the fragments of group i + 1 are read while the MFMAs of group i issue
        // (two register sets of 6 x 4).  Without the explicit order the scheduler hoists all 54 reads of a K-step and spills the prefetched
        // tile; nine per group spill 40 registers.
        constexpr int WG_ = 6, NG_ = 4 * 27 / WG_;
        bf16x8 afr[2][WG_], bfr[2];
        auto load_b = [&](int s) {
            const s16x4 bl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + bbase + s * 2 * WT_DYROW));
            const s16x4 bh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + bbase + s * 2 * WT_DYROW + 128));
            bfr[s & 1] = (bf16x8){bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
        };
        auto load_group = [&](int i, int set) {
#pragma unroll
            for (int k = 0; k < WG_; ++k) {
                const int e = i * WG_ + k, s = e / 27, t = e - 27 * s;
                if (t == 0) load_b(s);
                const int o = abase + (t / 9) * DSB + ((t / 3) % 3 + 2 * s) * ROWB + (t % 3) * UNIT;
                const s16x4 al = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(halo + o));
                const s16x4 ah = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(halo + o + 4 * UNIT));
                afr[set][k] = (bf16x8){al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
            }
        };
        load_group(0, 0);
#pragma unroll
        for (int i = 0; i < NG_; ++i) {
            if (i + 1 < NG_) load_group(i + 1, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < WG_; ++k) {
                const int e = i * WG_ + k, s = e / 27, t = e - 27 * s;
                acc[t] = VG_MFMA16(afr[i & 1][k], bfr[s & 1], acc[t]);
                if (t == 26) acc[27] = VG_MFMA16(ones, bfr[s & 1], acc[27]);   // every row: the column sums of dY (bias gradient)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.dbg & 8) { t_a = __builtin_amdgcn_s_memtime(); c_k += t_a - t_b; }
    }
    if ((p.dbg & 8) && tid == 0) {
        unsigned long long* o = (unsigned long long*)(p.tickets + 4096) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
        o[0] = c_wait0; o[1] = c_commit; o[2] = c_wait1; o[3] = c_k;
    }
    // the four waves' slabs added in wave order (deterministic), then one store per element into this workgroup's partial slab:
    // lane (kg, li) of tile t holds [ci = 4 kg + e][co = li]
    float* red = (float*)halo;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 28; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* r = red + (t * 4 + e) * 64 + lane;
                    *r = w == 0 ? acc[t][e] : *r + acc[t][e];
                }
        }
    }
    __syncthreads();
    float* slab = p.part + ((size_t)chunk * p.nslab + (size_t)n * gridDim.x + blockIdx.x) * WT_SLAB;
    for (int i = tid; i < 28 * 256; i += 256) {
        const int l = i & 63, e = (i >> 6) & 3, t = i >> 8;
        slab[t * 256 + (4 * (l >> 4) + e) * 16 + (l & 15)] = red[i];
    }
}

// dw[(t Cin + 16 chunk + ci) 16 + co] += sum over the chunk's slabs, in a fixed order whatever the timing: block (column of 256 elements,
// chunk, group gr of WT_RG) -- wave w adds the slabs 4 gr + w, + 4 WT_RG, .. (16 loads of 16 bytes in flight per lane), the four waves are
// added as ((0 + 1) + 2) + 3, the group's partial goes to part2, and the group that takes the column's last ticket adds the WT_RG partials in
// group order and hands the result to dw (the bias row, t = 27, ci = 0 of chunk 0, to db).  28 x chunks x WT_RG blocks instead of 28 x chunks:
// the slab pass of a 512-workgroup launch is a 2-deep chain of loads, not a 16-deep one.
constexpr int WT_RG = 8;
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* part, int nslab, int Cin, float* dw, float* db, float* part2, unsigned* tickets) {
    __shared__ f32x4 sm[4][64];
    __shared__ int last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, chunk = blockIdx.y, gr = blockIdx.z;
    const int j = (blockIdx.x * 64 + lane) * 4;                          // slab-local element of this lane's 4
    const float* base = part + (size_t)chunk * nslab * WT_SLAB + j;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = 4 * gr + w;
    for (; b + 60 * WT_RG < nslab; b += 64 * WT_RG) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *(const f32x4*)(base + (size_t)(b + 4 * WT_RG * k) * WT_SLAB);
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; b < nslab; b += 4 * WT_RG) s += *(const f32x4*)(base + (size_t)b * WT_SLAB);
    sm[w][lane] = s;
    __syncthreads();
    float* col = part2 + ((size_t)(chunk * 28 + blockIdx.x) * WT_RG) * 256;           // [group][256 elements] of this column
    if (w == 0) {
        const f32x4 r = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
        float* o = col + gr * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) __hip_atomic_store(o + k, r[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the partial is written through before the ticket (the K split's exchange)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* tk = tickets + chunk * 28 + blockIdx.x;
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == WT_RG - 1;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last || w != 0) return;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g2 = 0; g2 < WT_RG; ++g2)
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] += __hip_atomic_load(col + g2 * 256 + lane * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int t = j >> 8, ci = (j >> 4) & 15, co = j & 15;              // 4 consecutive co
    if (t < 27) {
        float* o = dw + ((size_t)t * Cin + 16 * chunk + ci) * 16 + co;
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(o + k, r[k]);              // (atomic: two applications of a network may add into one gradient buffer)
    } else if (ci == 0 && chunk == 0 && db) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(db + co + k, r[k]);
    }
}

// VG_OK when served (or recorded by a dry run), 1 when the shape is not one of this kernel's, < 0 on error
int vg_wgrad_thin(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host, int T_total, float* dw, float* db,
                  float* scratch, int64_t scratch_bytes, hipStream_t s) {
    if (!vg_tune("WGRAD_THIN", 1) || d->f32 || dy_f32 || d->src_f32 || d->wpack || d->noise || d->nclass > 1) return 1;
    const int Cin = d->c_src0 + d->c_src1;
    if (d->ntaps != 27 || T_total != 27 || d->istr != 1 || d->Cout != 16 || (Cin % 16) || Cin < 16 || Cin > 64 || d->CK != 16) return 1;
    if (d->OD != d->D || d->OH != d->H || d->OW != d->W) return 1;
    if ((int64_t)d->D * d->H * d->W < vg_tune("WGRAD_THIN_MINVOX", 65536)) return 1;     // small grids: the slab costs more than the K loop
    int tmin[3] = {d->tap_d[0], d->tap_h[0], d->tap_w[0]};
    for (int t = 0; t < 27; ++t)                                         // raster order: the tap is an immediate offset in the loop
        if (tap_idx_host[t] != t || d->tap_d[t] != tmin[0] + t / 9 || d->tap_h[t] != tmin[1] + (t / 3) % 3 || d->tap_w[t] != tmin[2] + t % 3) return 1;
    GatherIn g;
    if (fill_gather(d, g, 16, 512) != VG_OK) return 1;                   // (the forward kernel's tile: 16 x 8 x 4; a skew argument would change the shape search)
    if (g.lean != VG_STAGE_PLAIN && g.lean != VG_STAGE_RELU) return 1;
    if (!g.planar || g.HW != HW || g.HH != HH || g.HD != HD || g.HWp != HW || g.HHp != HH || g.DS != DSB || g.PSB != PSB) return 1;
    g.PSB += 64; g.CS = g.PSB;                                           // plane stride + 64 bytes: the two planes of a voxel on different banks
    if ((1 << g.twl) != TW || (1 << g.thl) != TH || (1 << g.tdl) != TD) return 1;
    const int AL = (g.tiles_h * TH + 2) + (g.tiles_w * TW + 2) + (g.tiles_d * TD + 2);
    const int lds = 2 * g.PSB + 32 * 4 + (2 * HH * HW * 2 + 2 * AL) * 4 + 16 + WT_DYB;
    if (lds > VG_LDS_LIMIT || 2 * g.PSB < WT_SLAB * 4) return 1;
    const int nchunks = Cin / 16, tiles = g.tiles_d * g.tiles_h * g.tiles_w;
    // persistent grid: the kernel is bound by latency -- loads, commit, K loop and the barriers between them run one after the other inside
    // a workgroup, two workgroups per CU at most (LDS) -- so alone it wants every slot (16 -> 16 at 128^3, two volumes: 173 / 152 / 128 us for
    // 256 / 384 / 512 workgroups), but it runs on a side stream next to its lane: in the step 384 is as fast as 512 or faster (17.79-17.86 vs 17.86-17.99 ms)
    int bx = vg_tune("WGRAD_THIN_WGS", 512) / (nchunks * d->N);
    if (bx < 1) bx = 1; if (bx > tiles) bx = tiles;
    int xw = 0;
    if (vg_tune("WGRAD_THIN_XCD", 1) && bx >= 16 && tiles >= 4 * bx) { bx &= ~7; xw = 1; }
    const int nslab = bx * d->N;
    // scratch: [chunks][slabs][WT_SLAB] partial slabs | [chunks][28][WT_RG][256] group partials of the slab pass | its tickets
    const int64_t slab_f = (int64_t)nchunks * nslab * WT_SLAB, part2_f = (int64_t)nchunks * 28 * WT_RG * 256;
    if (!scratch || (slab_f + part2_f + nchunks * 28 + 64 + 4096 + 8 * 4096) * 4 > scratch_bytes) return 1;
    if (vg_dry("wgrad_thin<m%d>|ch%d|walk%d", g.lean, nchunks > 1 ? 1 : 0, tiles > bx ? 1 : 0)) return VG_OK;
    float* part2 = scratch + slab_f;
    unsigned* tickets = (unsigned*)(part2 + part2_f);
    const int wt_dbg = vg_ablate_knob("WT_DBG");
    if (wt_dbg < 0) return VG_EINVAL;
    WgThin p = {dy, scratch, dw, db, Cin, nslab, wt_dbg, tickets, nchunks * 28, xw};
    const dim3 grid(bx, nchunks, d->N);
    if (g.lean == VG_STAGE_PLAIN) {
        static bool a0 = false;
        if (!a0) { (void)hipFuncSetAttribute((const void*)wgrad_thin_kernel<VG_STAGE_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT); a0 = true; }
        hipLaunchKernelGGL((wgrad_thin_kernel<VG_STAGE_PLAIN>), grid, dim3(256), lds, s, g, p);
    } else {
        static bool a1 = false;
        if (!a1) { (void)hipFuncSetAttribute((const void*)wgrad_thin_kernel<VG_STAGE_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT); a1 = true; }
        hipLaunchKernelGGL((wgrad_thin_kernel<VG_STAGE_RELU>), grid, dim3(256), lds, s, g, p);
    }
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(WT_SLAB / 256, nchunks, WT_RG), dim3(256), 0, s, (const float*)scratch, nslab, Cin, dw, db, part2, tickets);
    return vg_check_launch();
}
