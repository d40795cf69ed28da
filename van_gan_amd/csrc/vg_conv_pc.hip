// vg_conv_pc.hip -- producer/consumer flavour of the gather-convolution (forward and data gradient, bf16).
//
// Same implicit GEMM, LDS halo image, tables, MFMA loop and epilogue as conv_kernel (vg_conv.hip), but the two halves of a
// tile's work run CONCURRENTLY on one CU instead of one after the other: a 512-thread workgroup has
//   waves 0-3  consumers: MFMA loop over the staged halo image + epilogue (bias / residual / tanh / statistics / stores)
//   waves 4-7  producers: global loads of the NEXT stage's halo, on-read InstanceNorm / activation / noise, bf16 rounding,
//                         LDS writes into the other of two halo buffers, plus the per-tile axis tables and scale/shift
// with ONE workgroup barrier per stage (a stage = one channel chunk of one tile).  conv_kernel's phases were each latency
// bound and added up (stage -> barrier -> MFMA -> epilogue -> barrier, ~16 k cycles per 512-voxel tile per CU
// against ~2 k cycles of MFMA work); here the matrix pipe of a SIMD (consumer wave) and its vector ALU / memory path
// (producer wave) work side by side, and the barrier orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier), so the
// consumers' epilogue stores drain in the background instead of being waited for by every wave (__syncthreads() waits
// for vmcnt(0)).  The producers carry no accumulators, so they can keep a whole column of loads in flight (UB = 10).
#include "vg_conv_common.h"

#ifndef VG_PC_UB
#define VG_PC_UB 10
#endif

#define VG_PSTAMP(who, slot) do { if (g.stamps && tid == (who) && s < 8) g.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + s * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)

// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct TileIt {
    int w, h, d;
    __device__ __forceinline__ void advance(const GatherIn& g, int sw, int sh, int sd) {
        w += sw; if (w >= g.tiles_w) { w -= g.tiles_w; ++h; }
        h += sh; if (h >= g.tiles_h) { h -= g.tiles_h; ++d; }
        d += sd;
    }
};

// MC: 0 = one class; 2 = class-parallel (the workgroup serves the class blockIdx.x % ncls), as in conv_kernel
// MODE: the on-read transform of the lean staging (VG_STAGE_*, vg_gather.h); single-channel sources (C1) keep the staging
// of conv_kernel, for them MODE is 0 (no noise) or 1 (noise)
// NCW: consumer waves = producer waves (4: 512-thread workgroups, 2 waves per SIMD for the big tiles; 8: 1024-thread
// workgroups, 4 waves per SIMD under a 128-register cap -- the same tile shared out over twice the waves: in-kernel stamps
// showed both roles latency/dependency bound at ~45 % of the vector issue rate with one consumer and one producer wave
// per SIMD)
template <typename T, int BN, int MSUB, int MODE, bool WL, int MC, bool C1, int NCW>
__global__ __launch_bounds__(128 * NCW, (NCW == 8 ? 4 : ((BN / 16) * MSUB >= 4 ? 2 : 4))) void conv_pc_kernel(const GatherIn g, const ConvOut p, const ConvCls q) {
    static_assert(sizeof(T) == 2, "the producer/consumer flavour is bf16 only");
    static_assert(MC == 0 || MC == 2, "fused classes stay on conv_kernel");
    constexpr int NT = 64 * NCW;                                   // threads per role
    constexpr int WN = BN / 16, WM = NCW / WN, MW = 4 * MSUB / WM;
    static_assert(WM * WN == NCW && MW >= 1 && MW * WM == 4 * MSUB, "tile does not divide over the consumer waves");
    static_assert(!C1 || NCW == 4, "single-channel staging assumes 256 producer threads");
    constexpr int SC1 = 1;                                         // C1 kernels only: always single-channel
    constexpr bool NOISE = C1 ? (MODE == 1) : (MODE == VG_STAGE_LRELU_NOISE || MODE == VG_STAGE_LRELU_NOISE_M);
    constexpr int UB = NOISE ? VG_PC_UB / 2 : VG_PC_UB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#define VG_KSTAMP(k) do { if (g.stamps && tid == 0) g.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + (k) * 8 + 7] = __builtin_readcyclecounter(); } while (0)
    VG_KSTAMP(0);
    const bool producer = wave >= NCW;
    const int ptid = tid - NT;
    const int wave_n = (wave % NCW) % WN, wave_m = (wave % NCW) / WN;
    const int n = blockIdx.z, ntile = blockIdx.y;
    const int TWm = (1 << g.twl) - 1, THm = (1 << g.thl) - 1;
    int bx = blockIdx.x, gx = gridDim.x, t0 = 0, nt = g.ntaps;
    int Ktot = p.Ktot, kc_pad = p.kc_pad, WRS = p.WRS;
    const void* wsrc = p.wp;
    int c_od = p.ood, c_oh = p.ooh, c_ow = p.oow, c_OD = p.OD, c_OH = p.OH, c_OW = p.OW;
    if constexpr (MC == 2) {
        const int cls = __builtin_amdgcn_readfirstlane(bx % q.ncls);
        bx /= q.ncls; gx /= q.ncls;
        t0 = q.tap0[cls]; nt = q.tap0[cls + 1] - t0;
        Ktot = q.ktot[cls]; kc_pad = Ktot / p.nchunks; WRS = Ktot * (int)sizeof(T) + 16;
        wsrc = q.wp[cls];
        c_od = q.off[cls][0]; c_oh = q.off[cls][1]; c_ow = q.off[cls][2];
        c_OD = q.it[cls][0]; c_OH = q.it[cls][1]; c_OW = q.it[cls][2];
    }
    // ---- LDS: [halo 0][halo 1][tap offsets][scale/shift x 2][statistics][column table][axis tables x 2][K-step offsets][weights]
    const int hbytes = g.planar ? (g.CK >> 3) * g.PSB : g.HD * g.DS;
    char* halo = smem;
    int* tapoff = (int*)(smem + 2 * hbytes);
    float* scs = (float*)(smem + 2 * hbytes + 256);
    float* stat = scs + 4 * g.CK;
    int* utab = (int*)(stat + BN * 2);
    const int ncols = stage_ncols(g);
    const int RTN = C1 ? 3 * stage_axis_len<SC1>(g) : 3 * stage_axis_len3(g);
    int* rtab = utab + 2 * ncols;
    const int gpc = g.CK >> 3;
    const int ngroups = nt * gpc;
    const int ksteps = (ngroups + 3) >> 2;
    int* koff = rtab + 2 * RTN;
    char* wlds = (char*)(koff + ksteps * 4);
    wlds = (char*)(((size_t)wlds + 15) & ~(size_t)15);

    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    const int nloc = bx < tiles_per_n ? (tiles_per_n - bx + gx - 1) / gx : 0;       // tiles this workgroup visits
    const int nchunks = p.nchunks;
    const int nstages = nloc * nchunks;
    int gs_w, gs_h, gs_d;
    { int t = gx; gs_w = t % g.tiles_w; t /= g.tiles_w; gs_h = t % g.tiles_h; gs_d = t / g.tiles_h; }
    TileIt first;
    { int t = bx; first.w = t % g.tiles_w; t /= g.tiles_w; first.h = t % g.tiles_h; first.d = t / g.tiles_h; }

    // ---- per-workgroup tables (all 512 threads) ----
    if (tid < nt)
        tapoff[tid] = (g.td[t0 + tid] - g.tmin_d) * g.DS + ((g.th[t0 + tid] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[t0 + tid] - g.tmin_w)) * g.VS;
    if (tid < BN * 2) stat[tid] = 0.f;
    for (int i = tid; i < ksteps * 4; i += 2 * NT) {
        int G = i; if (G >= ngroups) G = ngroups - 1;                       // padded K: weights are zero there
        int tp = G / gpc; const int cgq = G - tp * gpc; tp += t0;
        koff[i] = (g.td[tp] - g.tmin_d) * g.DS + ((g.th[tp] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[tp] - g.tmin_w)) * g.VS + cgq * g.CS;
    }
    if (WL) {               // weight panel -> LDS, 16 B per thread per step, four loads in flight per thread
        const char* src = (const char*)wsrc;
        const int per_row = (Ktot * (int)sizeof(T)) >> 4;
        for (int u0 = tid; u0 < BN * per_row; u0 += 8 * NT) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = min(u0 + k * 2 * NT, BN * per_row - 1);
                const int r = u / per_row, c = u - r * per_row;
                v[k] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(src + ((size_t)(ntile * BN + r) * Ktot) * sizeof(T) + c * 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * 2 * NT;
                if (u < BN * per_row) { const int r = u / per_row, c = u - r * per_row; *(f32x4*)(wlds + (size_t)r * WRS + c * 16) = v[k]; }
            }
        }
    }
    auto resolve = [&](int* rt, const TileIt& t) {
        if constexpr (C1) stage_resolve_axes<SC1>(g, rt, t.h << g.thl, t.w << g.twl, ptid);
        else stage_resolve_axes3(g, rt, t.d << g.tdl, t.h << g.thl, t.w << g.twl, ptid);
    };
    auto stage = [&](char* hb, const float* sc, const int* rt, const TileIt& t, int chunk) {
        if constexpr (C1) stage_halo_tile<T, NOISE, UB, SC1>(g, hb, sc, utab, rt, n, t.d << g.tdl, chunk, ptid);
        else stage_halo_lean<T, MODE, (NCW == 8 ? UB / 2 : UB), NT>(g, hb, sc, utab, rt, n, chunk, ptid);
    };
    if (producer) {
        build_column_table<NT>(g, utab, ptid);
        if (nstages > 0) {
            stage_scale_shift(g, scs, n, 0, ptid);
            if (nchunks == 1) stage_scale_shift(g, scs + 2 * g.CK, n, 0, ptid);       // one chunk: both buffers hold it for good
            resolve(rtab, first);
        }
    }
    lds_barrier();
    VG_KSTAMP(1);

    // ---- consumer constants ----
    int rowbase[MW], ooff[MW], dhw[MW];
    const int co0 = ntile * BN + wave_n * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = (wave_m * MW + i) * 16 + (lane & 15);
        const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
        rowbase[i] = d * g.istr * g.DS + (h * g.istr * g.HWp + w) * g.VS;
        ooff[i] = ((d * p.ostr * p.BH + h * p.ostr) * p.BW + w * p.ostr) * p.Cout + co0;
        dhw[i] = d | (h << 10) | (w << 20);
    }
    const lds_ptr<T> wrow_l = (lds_ptr<T>)(wlds + (size_t)(wave_n * 16 + (lane & 15)) * WRS) + 8 * (lane >> 4);
    const glb_ptr<T> wrow_g = (glb_ptr<T>)wsrc + (size_t)(ntile * BN + wave_n * 16 + (lane & 15)) * Ktot + 8 * (lane >> 4);
    float s1[4], s2[4], e_bias[4], e_rs[4], e_rb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s1[r] = 0.f; s2[r] = 0.f;
        const int co = co0 + r;
        e_bias[r] = (!producer && p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        e_rs[r] = (!producer && p.res && co < p.Cout) ? p.rs[n * p.Cout + co] : 0.f;
        e_rb[r] = (!producer && p.res && co < p.Cout) ? p.rb[n * p.Cout + co] : 0.f;
    }
    const bool vec_epi = (p.Cout & 3) == 0 && !p.tanh_out;        // every lane owns 4 whole channels

    // ---- prologue: stage 0 into buffer 0, tables of stage 1 ----
    // producer cursor: (pj, pc) = (local tile, chunk) of the stage it stages next; pcur / pnxt = coordinates of tile pj / pj + 1
    int pj = 0, pc = 0;
    TileIt pcur = first, pnxt = first;
    pnxt.advance(g, gs_w, gs_h, gs_d);
    if (producer && nstages > 0) {
        stage(halo, scs, rtab, pcur, 0);
        if (++pc == nchunks) { pc = 0; ++pj; pcur = pnxt; pnxt.advance(g, gs_w, gs_h, gs_d); }
        if (nstages > 1) {
            if (pc == 0) resolve(rtab + (pj & 1) * RTN, pcur);
            if (nchunks > 1) stage_scale_shift(g, scs + 2 * g.CK, n, pc, ptid);
        }
    }
    lds_barrier();
    VG_KSTAMP(2);

    // Two role loops with the same number of barriers (one per stage).  They are separate loops, not two branches of one
    // loop body, so that the consumers' loop-carried state (accumulators, per-lane epilogue constants: ~80 registers) is not
    // live in the producers' loop and vice versa: the register allocation is the maximum of the two roles, not the sum.
    if (producer) {
        for (int s = 0; s < nstages; ++s) {
            VG_PSTAMP(NT, 0);
            if (s + 1 < nstages && !(g.dbg & 1)) {
                const float* sc_cur = scs + (nchunks > 1 ? ((s + 1) & 1) * 2 * g.CK : 0);
                stage(halo + ((s + 1) & 1) * hbytes, sc_cur, rtab + (pj & 1) * RTN, pcur, pc);
                if (++pc == nchunks) { pc = 0; ++pj; pcur = pnxt; pnxt.advance(g, gs_w, gs_h, gs_d); }
                if (s + 2 < nstages) {
                    if (pc == 0) resolve(rtab + (pj & 1) * RTN, pcur);
                    if (nchunks > 1) stage_scale_shift(g, scs + (s & 1) * 2 * g.CK, n, pc, ptid);
                }
            }
            VG_PSTAMP(NT, 1);
            lds_barrier();
            VG_PSTAMP(NT, 5);
        }
    } else {
        int cc = 0;
        TileIt ccur = first;
        f32x4 acc[MW];
        for (int s = 0; s < nstages; ++s) {
            VG_PSTAMP(0, 2);
            if (cc == 0) {
#pragma unroll
                for (int i = 0; i < MW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const char* hb = halo + (s & 1) * hbytes;
            const size_t kbase = (size_t)cc * kc_pad;
            if (!(g.dbg & 4)) {
                if constexpr (WL) conv_mfma_chunk<T, MW>(acc, wrow_l + kbase, hb, rowbase, tapoff, koff, ksteps, nt, g.CK, g.CS, lane);
                else conv_mfma_chunk<T, MW>(acc, wrow_g + kbase, hb, rowbase, tapoff, koff, ksteps, nt, g.CK, g.CS, lane);
            }
            VG_PSTAMP(0, 3);
            if (++cc == nchunks) {
                cc = 0;
                const int od0 = ccur.d << g.tdl, oh0 = ccur.h << g.thl, ow0 = ccur.w << g.twl;
                ccur.advance(g, gs_w, gs_h, gs_d);
                if (!(g.dbg & 8)) {
#include "vg_conv_epilogue.inc"
                }
            }
            VG_PSTAMP(0, 4);
            lds_barrier();
            VG_PSTAMP(0, 6);
        }
    }
    VG_KSTAMP(3);
    if (p.sums) {
        if (!producer) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[r], b = s2[r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if ((lane & 15) == 0) {
                    const int cl = wave_n * 16 + 4 * (lane >> 4) + r;
                    atomicAdd(&stat[cl * 2], a);
                    atomicAdd(&stat[cl * 2 + 1], b);
                }
            }
        }
        lds_barrier();
        if (tid < BN * 2) {
            const int co = ntile * BN + (tid >> 1);
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            if (co < p.Cout) atomicAdd(&p.sums[(((size_t)stripe * gridDim.z + n) * p.Cout + co) * 2 + (tid & 1)], stat[tid]);
        }
    }
    VG_KSTAMP(4);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int vg_conv_pc_lds_bytes(const GatherIn& g, int BN, int CK, int wbytes, int ksteps_total) {
    const int ksteps = ksteps_total > 0 ? ksteps_total : (g.ntaps * (CK >> 3) + 3) >> 2;
    return 2 * halo_bytes(g) + 256 + 4 * CK * 4 + BN * 2 * 4 + stage_table_ints3(g) * 4 + ksteps * 16 + 16 + wbytes;
}

template <typename T, int BN, int MSUB, int MODE, bool WL, int MC, bool C1, int NCW>
static int launch_pc4(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_pc_kernel<T, BN, MSUB, MODE, WL, MC, C1, NCW>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    // persistent grid = resident capacity: the variants with >= 4 sub-tiles per wave are compiled for 2 waves per SIMD (one
    // 512-thread workgroup per CU; under the 128-register cap of 4 waves they spilled 24-38 VGPRs), the others for 4 (two
    // workgroups per CU when LDS allows)
    int per_cu = (NCW == 8 || (BN / 16) * MSUB >= 4) ? 1 : 2;
    if (lds > 0 && VG_LDS_LIMIT / lds < per_cu) per_cu = VG_LDS_LIMIT / lds;
    if (per_cu < 1) per_cu = 1;
    const int wg_env = vg_tune("CONV_PC_WGS", 0);
    const int wg_target = wg_env > 0 ? wg_env : 256 * per_cu;
    const int tiles = g.tiles_d * g.tiles_h * g.tiles_w;
    const int ny = (k.Cout + BN - 1) / BN;
    const int ncp = MC == 2 ? q.ncls : 1;
    int bx = wg_target / (ny * g.N * ncp); if (bx < 1) bx = 1; if (bx > tiles) bx = tiles;
    if (vg_dry("conv_pc<bf16,%d,%d,m%d,wl%d,mc%d,c1%d,w%d>|walk%d|ch%d", BN, MSUB, MODE, (int)WL, MC, (int)C1, NCW, tiles > bx ? 1 : 0,
               k.nchunks > 1 ? 1 : 0)) return VG_OK;
    hipLaunchKernelGGL((conv_pc_kernel<T, BN, MSUB, MODE, WL, MC, C1, NCW>), dim3(bx * ncp, ny, g.N), dim3(128 * NCW), lds, s, g, k, q);
    return vg_check_launch();
}
template <typename T, int BN, int MSUB, int MODE, bool WL, int MC, bool C1>
static int launch_pc3(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    // tiles of >= 4 sub-tiles per consumer wave can be shared out over 8 + 8 waves instead (CONV_PC_W8, default on)
#ifdef VG_PC_W8      // measured slower (stem.cb forward 0.101 -> 0.137 ms): twice the waves issue the same vector work plus their own
                     // per-wave overheads; the roles are bound by vector-instruction issue, not by latency.  Kept for experiments.
    if constexpr (!C1 && (BN / 16) * MSUB >= 4) {
        if (vg_tune("CONV_PC_W8", 0)) return launch_pc4<T, BN, MSUB, MODE, WL, MC, C1, 8>(g, k, q, lds, s);
    }
#endif
    return launch_pc4<T, BN, MSUB, MODE, WL, MC, C1, 4>(g, k, q, lds, s);
}
// staging mode of a multi-channel source, or -1 when the combination has no lean variant (the caller then stays on conv_kernel)
int vg_conv_pc_mode(const GatherIn& g) {
    if (g.Cin == 1) return g.noise ? 1 : 0;
    return (g.lean >= 0 && g.lean != VG_STAGE_GENERIC) ? g.lean : -1;
}
template <int BN, int MSUB, bool WL>
static int launch_pc2(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    typedef bf16_t T;
    const int mode = vg_conv_pc_mode(g);
    if (mode < 0) return VG_EINVAL;
    if (q.par) {                                    // class-parallel data gradient: plain multi-channel sources
        if (mode != VG_STAGE_PLAIN || g.Cin == 1) return VG_EINVAL;
        return launch_pc3<T, BN, MSUB, VG_STAGE_PLAIN, WL, 2, false>(g, k, q, lds, s);
    }
    if (g.Cin == 1) return mode ? launch_pc3<T, BN, MSUB, 1, WL, 0, true>(g, k, q, lds, s) : launch_pc3<T, BN, MSUB, 0, WL, 0, true>(g, k, q, lds, s);
    switch (mode) {
        case VG_STAGE_PLAIN: return launch_pc3<T, BN, MSUB, VG_STAGE_PLAIN, WL, 0, false>(g, k, q, lds, s);
        case VG_STAGE_RELU: return launch_pc3<T, BN, MSUB, VG_STAGE_RELU, WL, 0, false>(g, k, q, lds, s);
        case VG_STAGE_LRELU: return launch_pc3<T, BN, MSUB, VG_STAGE_LRELU, WL, 0, false>(g, k, q, lds, s);
        case VG_STAGE_LRELU_NOISE: return launch_pc3<T, BN, MSUB, VG_STAGE_LRELU_NOISE, WL, 0, false>(g, k, q, lds, s);
        case VG_STAGE_LRELU_M: return launch_pc3<T, BN, MSUB, VG_STAGE_LRELU_M, WL, 0, false>(g, k, q, lds, s);
        default: return launch_pc3<T, BN, MSUB, VG_STAGE_LRELU_NOISE_M, WL, 0, false>(g, k, q, lds, s);
    }
}
template <int BN, int MSUB>
static int launch_pc1(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    return k.w_lds ? launch_pc2<BN, MSUB, true>(g, k, q, lds, s) : launch_pc2<BN, MSUB, false>(g, k, q, lds, s);
}
int vg_launch_conv_pc(const GatherIn& g, const ConvOut& k, const ConvCls& q, int BN, int MSUB, int lds, hipStream_t s) {
    if (BN == 16) {
        switch (MSUB) {
            case 8: return launch_pc1<16, 8>(g, k, q, lds, s);
            case 4: return launch_pc1<16, 4>(g, k, q, lds, s);
            case 2: return launch_pc1<16, 2>(g, k, q, lds, s);
            default: return launch_pc1<16, 1>(g, k, q, lds, s);
        }
    }
    if (BN == 32) return MSUB == 4 ? launch_pc1<32, 4>(g, k, q, lds, s) : (MSUB == 2 ? launch_pc1<32, 2>(g, k, q, lds, s) : launch_pc1<32, 1>(g, k, q, lds, s));
    return MSUB == 2 ? launch_pc1<64, 2>(g, k, q, lds, s) : launch_pc1<64, 1>(g, k, q, lds, s);
}
