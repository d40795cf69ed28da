// vg_wgrad.hip -- weight gradient of the gather-convolution on bf16 MFMA (gfx950).
//
// dW[tap][ci][co] += sum_{n,o} P[n, bnd(o*istr+tap), ci] * dY[n,o,co],   db[co] += sum dY
// (what tf.GradientTape returns for the Conv3D kernels/biases on the path, vangan.py:426-438), with
// P = the same on-read transformed operand the forward uses (vg_gather.h).
//
// GEMM view: M = ci, N = co, K = voxels.  Both operands are stored [voxel][channel] (channel innermost),
// i.e. K is the SLOW axis of both, so the MFMA fragments (8 consecutive k per lane) are fetched with the
// gfx950 transposing LDS read ds_read_b64_tr_b16: 4 voxel rows x 16 channels per 16-lane group, delivered
// column-major.
//
// A persistent workgroup owns a slab of dW = (tap group) x (ci block) x (co block of 16*Q channels) and walks a strided
// set of voxel tiles.  A wave owns "rows" r = (tap, 16 ci) of the slab (r = wave, wave+4, ...) times all Q co-blocks:
// per 32-voxel K-step it reads the Q dY fragments once and one P fragment per row, so LDS traffic per MFMA is
// (2R+2Q)/(RQ) transposing reads instead of 4.  The slab lives in accumulators for the whole walk and is added to dW
// with fp32 atomics once at the end.
#include "vg_gather.h"

struct WgradK {
    const void* dy; int dy_f32; int Cout;
    int OD, OH, OW;
    int CIB, COB, ncib, ncob, tpg, ntg;     // ci/co block sizes, counts, taps per group, tap groups
    int tap_src[VG_MAX_TAPS];               // packed tap -> source tap index in dW
    float* dw; float* db;
    int total_tiles, DYS;
    float* part; int dw_elems;              // per-workgroup-column partial slabs (no atomics) or NULL
};

// halo units a thread has in flight per staging batch (each a 16-byte load, plus one of the noise tensor)
#ifndef VG_WGRAD_UB
#define VG_WGRAD_UB (NOISE ? VG_WGRAD_UBN : (RMAX * Q > 4 ? 2 : 4))
#endif
#ifndef VG_WGRAD_UBN
#define VG_WGRAD_UBN 4
#endif
#ifndef VG_WGRAD_2W
#define VG_WGRAD_2W 16     // accumulator fragments per wave from which the register cap is 2 waves per SIMD
#endif
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#define VG_WSTAMP(slot) do { if (g.stamps && tid == 0 && it < 8) g.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + it * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ bf16x8 tr_frag(const char* base0, const char* base1) {
    // two transposed 4x16 block reads -> 8 consecutive k for this lane's column
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)base0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)base1);
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <typename T, int RMAX, int Q, bool NOISE>
__global__ __launch_bounds__(256, (RMAX * Q >= VG_WGRAD_2W ? 2 : 3)) void wgrad_kernel(const GatherIn g, const WgradK p) {
    constexpr bool F32 = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, li = lane & 15;
    int by = blockIdx.y;
    const int cob = by % p.ncob; by /= p.ncob;
    const int cib = by % p.ncib; const int tg = by / p.ncib;
    const int tap0 = tg * p.tpg;
    const int ntap_here = min(p.tpg, g.ntaps - tap0);
    const int tci = p.CIB >> 4;
    const int nrows = ntap_here * tci;

    const int BM = 1 << (g.tdl + g.thl + g.twl);
    char* halo = smem;
    const int hbytes = g.planar ? (g.CK >> 3) * g.PSB : g.HD * g.DS;
    char* dyt = smem + hbytes;
    const int dybytes = BM * p.DYS;
    int* tapoff = (int*)(dyt + dybytes);
    int* tapsrc = tapoff + 64;
    float* scs = (float*)((char*)tapoff + 512);
    int* utab = (int*)(scs + 2 * g.CK);
    int* rtab = utab + 2 * stage_ncols(g);
    const int RTN = 3 * stage_axis_len(g);               // two axis-table buffers: tile t+1 is resolved while tile t is staged
    int* ktab = rtab + 2 * RTN;                      // [BM/32][64 lanes]{r0, r1}: halo byte offsets of the transposed reads

    if (tid < g.ntaps)
        tapoff[tid] = (g.td[tid] - g.tmin_d) * g.DS + ((g.th[tid] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[tid] - g.tmin_w)) * g.VS;
    if (tid < VG_MAX_TAPS) tapsrc[tid] = p.tap_src[tid];
    build_column_table(g, utab, tid);
    if ((int)blockIdx.x < p.total_tiles) {           // axis tables of the first tile
        int t2 = blockIdx.x % (g.tiles_d * g.tiles_h * g.tiles_w);
        const int fw_i = t2 % g.tiles_w; t2 /= g.tiles_w;
        const int fh_i = t2 % g.tiles_h;
        stage_resolve_axes(g, rtab, fh_i << g.thl, fw_i << g.twl, tid);
    }
    if constexpr (!F32) {
        const int TWm_ = (1 << g.twl) - 1, THm_ = (1 << g.thl) - 1;
        for (int e = tid; e < (BM / 32) * 64; e += 256) {
            // lane (lg2, li2) of K-step ks supplies voxel row m0 (and m0 + 4) for the transposed block reads
            const int ks = e >> 6, l2 = e & 63, lg2 = l2 >> 4, li2 = l2 & 15;
            const int m0 = ks * 32 + 8 * lg2 + (li2 >> 2), m1 = m0 + 4;
            const int w0 = m0 & TWm_, h0 = (m0 >> g.twl) & THm_, d0 = m0 >> (g.twl + g.thl);
            const int w1 = m1 & TWm_, h1 = (m1 >> g.twl) & THm_, d1 = m1 >> (g.twl + g.thl);
            const int lo = ((li2 & 3) >> 1) * g.CS + 8 * (li2 & 1);
            ktab[2 * e] = d0 * g.istr * g.DS + (h0 * g.istr * g.HWp + w0) * g.VS + lo;
            ktab[2 * e + 1] = d1 * g.istr * g.DS + (h1 * g.istr * g.HWp + w1) * g.VS + lo;
        }
    }
    __syncthreads();

    // rows of this wave: r = wave + 4*j -> (tap, ci16); byte offset of the row's P fragment inside the halo tile
    int aoff[RMAX];
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        const int r = wave + 4 * j;
        const int rr = r < nrows ? r : 0;
        aoff[j] = tapoff[tap0 + rr / tci] + (rr % tci) * 2 * g.CS;          // a 16-channel row block = two 8-channel groups
    }
    f32x4 acc[RMAX][Q];
#pragma unroll
    for (int j = 0; j < RMAX; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbsum = 0.f;
    int cur_n = -1;

    const int TWm = (1 << g.twl) - 1, THm = (1 << g.thl) - 1;
    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    constexpr int gcol = (Q * 16) >> 3, gcol_l = Q == 1 ? 1 : (Q == 2 ? 2 : 3);     // 8-channel groups per dY row
    const bool do_db = p.db && cib == 0 && tg == 0;

    int it = -1;
    // (sample, tile) coordinates advance incrementally by the grid stride: no per-tile divisions
    int gs_w, gs_h, gs_d, gs_n;
    { int t = gridDim.x; gs_w = t % g.tiles_w; t /= g.tiles_w; gs_h = t % g.tiles_h; t /= g.tiles_h; gs_d = t % g.tiles_d; gs_n = t / g.tiles_d; }
    int ti_w, ti_h, ti_d, ti_n;
    { int t = blockIdx.x; ti_w = t % g.tiles_w; t /= g.tiles_w; ti_h = t % g.tiles_h; t /= g.tiles_h; ti_d = t % g.tiles_d; ti_n = t / g.tiles_d; }
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        ++it;
        VG_WSTAMP(0);
        const int n = ti_n;
        const int od0 = ti_d << g.tdl, oh0 = ti_h << g.thl, ow0 = ti_w << g.twl;
        ti_w += gs_w; if (ti_w >= g.tiles_w) { ti_w -= g.tiles_w; ++ti_h; }
        ti_h += gs_h; if (ti_h >= g.tiles_h) { ti_h -= g.tiles_h; ++ti_d; }
        ti_d += gs_d; if (ti_d >= g.tiles_d) { ti_d -= g.tiles_d; ++ti_n; }
        ti_n += gs_n;
        __syncthreads();                       // previous tile consumed; this tile's axis tables (built during it) visible
        if (n != cur_n) {                      // block-uniform: the on-read affine depends on the sample only
            stage_scale_shift(g, scs, n, cib, tid);
            cur_n = n;
            __syncthreads();
        }
        int* rt = rtab + (it & 1) * RTN;
        // ---- dY tile [BM][16*Q] (zero outside the grid / beyond Cout): its loads are issued first and stay in flight
        // while the halo tile is staged.  Straight-line: out-of-range units load a clamped address and are zeroed ----
        Raw8<T> yraw[4]; float y1[4]; bool yok[4];
        const bool dy_on = !(g.dbg & 2) || tile == (int)blockIdx.x;
        if (dy_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = min(tid + k * 256, BM * gcol - 1);
                const int m = u >> gcol_l, cg = u & (gcol - 1);
                const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
                const int od = od0 + d, oh = oh0 + h, ow = ow0 + w;
                const int c = cob * p.COB + cg * 8;
                yok[k] = od < p.OD && oh < p.OH && ow < p.OW && c < p.Cout;
                const size_t vox = yok[k] ? ((size_t)(n * p.OD + od) * p.OH + oh) * p.OW + ow : 0;
                if (p.Cout == 1) y1[k] = p.dy_f32 ? ld_global((const float*)p.dy + vox) : ld_global((const bf16_t*)p.dy + vox);
                else raw_load(yraw[k], (const T*)p.dy + vox * p.Cout + (yok[k] ? c : 0));
            }
        }
        if (!(g.dbg & 1) || tile == (int)blockIdx.x) stage_halo_tile<T, NOISE, VG_WGRAD_UB>(g, halo, scs, utab, rt, n, od0, cib, tid);
        if (tile + (int)gridDim.x < p.total_tiles) {      // axis tables of the next tile into the other buffer
            stage_resolve_axes(g, rtab + ((it + 1) & 1) * RTN, ti_h << g.thl, ti_w << g.twl, tid);
        }
        if (dy_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = tid + k * 256;
                if (u < BM * gcol) {
                    const int m = u >> gcol_l, cg = u & (gcol - 1);
                    T* dst = (T*)(dyt + (size_t)m * p.DYS) + cg * 8;
                    if (p.Cout == 1) {
                        const float v[8] = {yok[k] ? y1[k] : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        store8<T>(dst, v);
                    } else {
                        raw_mask(yraw[k], yok[k]);
                        *(Raw8<T>*)dst = yraw[k];
                    }
                }
            }
        }
        VG_WSTAMP(1);
        __syncthreads();
        VG_WSTAMP(2);
        if (do_db) {      // thread (row group, channel): partial column sums of the dY tile
            const int nrg = 256 / p.COB;
            if (tid < nrg * p.COB) {
                const int c = tid % p.COB;
                float s = 0.f;
                for (int m = tid / p.COB; m < BM; m += nrg) s += ld1<T>((const T*)(dyt + (size_t)m * p.DYS) + c);
                dbsum += s;
            }
        }
        if (g.dbg & 4) continue;
        if constexpr (F32) {
            // exact-parity mode: v_mfma_f32_16x16x4_f32, k = 4 voxels; lane (lg, li): A[ci=li][k=lg], B[k=lg][co=li]
            for (int s = 0; s < BM / 4; ++s) {
                const int m0 = s * 4 + lg;
                const int w0 = m0 & TWm, h0 = (m0 >> g.twl) & THm, d0 = m0 >> (g.twl + g.thl);
                const int r0 = d0 * g.istr * g.DS + (h0 * g.istr * g.HWp + w0) * g.VS + (li >> 3) * g.CS + (li & 7) * 4;
                const char* y0 = dyt + (size_t)m0 * p.DYS + li * 4;
                float b[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) b[q] = *(const float*)(y0 + q * 64);
#pragma unroll
                for (int j = 0; j < RMAX; ++j) {
                    const float a = *(const float*)(halo + r0 + aoff[j]);
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[q], acc[j][q], 0, 0, 0);
                }
            }
        } else {
            // ---- K loop over voxels, 32 per MFMA.  Rows are processed in chunks of RC; the operand fragments of the
            // next chunk (or of the next K-step: its dY fragments too) are fetched before the MFMAs of the current one.
            // No MFMA or fetch is conditional: rows beyond nrows re-read row 0 and are dropped at the slab write ----
            constexpr int RC = RMAX <= 8 ? RMAX : 6;
            static_assert(RMAX % RC == 0, "rows per wave must be a multiple of the chunk");
            constexpr int NCH = RMAX / RC;
            const int nks = BM / 32;                                   // even (BM is 64, 128 or 256)
            // per-lane halo offsets of K-step ks come from the table built once per workgroup (kaddr_fill); the dY
            // offsets advance by 32 rows per K-step.  Two register sets per operand, selected by compile-time parity, so
            // the software pipeline needs no register moves.
            const int2* kt = (const int2*)ktab + lane;
            const char* ybase = dyt + (size_t)(8 * lg + (li >> 2)) * p.DYS + 8 * (li & 3);
            const int ystep = 32 * p.DYS, y4 = 4 * p.DYS;
            bf16x8 A[2][RC], B[2][Q];
            {
                const int2 r = kt[0];
#pragma unroll
                for (int q = 0; q < Q; ++q) B[0][q] = tr_frag(ybase + q * 32, ybase + y4 + q * 32);          // B[k=voxel][co]
#pragma unroll
                for (int j = 0; j < RC; ++j) A[0][j] = tr_frag(halo + r.x + aoff[j], halo + r.y + aoff[j]);   // A[ci][k=voxel]
            }
            for (int ks = 0; ks < nks; ks += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int2 rc = kt[(ks + u) * 64];
                    const int kn = min(ks + u + 1, nks - 1);
                    const int2 rn = kt[kn * 64];
                    const char* yn = ybase + (size_t)kn * ystep;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        constexpr int dummy = 0; (void)dummy;
                        const int s = (u * NCH + c) & 1;
                        if (c + 1 < NCH) {
#pragma unroll
                            for (int j = 0; j < RC; ++j) A[s ^ 1][j] = tr_frag(halo + rc.x + aoff[(c + 1) * RC + j], halo + rc.y + aoff[(c + 1) * RC + j]);
                        } else {
#pragma unroll
                            for (int q = 0; q < Q; ++q) B[u ^ 1][q] = tr_frag(yn + q * 32, yn + y4 + q * 32);
#pragma unroll
                            for (int j = 0; j < RC; ++j) A[s ^ 1][j] = tr_frag(halo + rn.x + aoff[j], halo + rn.y + aoff[j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < RC; ++j)
#pragma unroll
                            for (int q = 0; q < Q; ++q)
                                acc[c * RC + j][q] = VG_MFMA16(A[s][j], B[u][q], acc[c * RC + j][q]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        VG_WSTAMP(3);
    }
    // ---- add the slab: lane holds dW rows ci = 4*lg + r, column co = li ----
    { const int it = 7; VG_WSTAMP(7); }
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        const int r = wave + 4 * j;
        if (r >= nrows) continue;
        const int tap = tap0 + r / tci, ci0 = cib * p.CIB + (r % tci) * 16 + 4 * lg;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int co = cob * p.COB + q * 16 + li;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ci0 + e < g.Cw) {
                    const size_t i = ((size_t)tapsrc[tap] * g.Cw + ci0 + e) * p.Cout + co;
                    if (p.part) p.part[(size_t)blockIdx.x * p.dw_elems + i] = acc[j][q][e];
                    else atomicAdd(&p.dw[i], acc[j][q][e]);
                }
        }
    }
    if (do_db) {        // block-reduce the bias partials in LDS first: one contiguous atomic per channel per workgroup
        __syncthreads();
        float* red = (float*)halo;
        if (tid < p.COB) red[tid] = 0.f;
        __syncthreads();
        if (tid < (256 / p.COB) * p.COB) atomicAdd(&red[tid % p.COB], dbsum);
        __syncthreads();
        if (tid < p.COB) {
            const int co = cob * p.COB + tid;
            if (co < p.Cout) atomicAdd(&p.db[co], red[tid]);
        }
    }
}

// dw[i] += sum_b part[b][i]  (fixed order per element: bitwise reproducible, unlike the atomic path).
// Block = 4 waves x 64 consecutive elements; wave w sums slabs b = w, w+4, ... with 8 independent loads in flight.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* part, int nb, int n, float* dw) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < n) {
        int b = w;
        for (; b + 28 < nb; b += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(b + 4 * k) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < nb; b += 4) s += part[(size_t)b * n + i];
    }
    sm[w][lane] = s;
    __syncthreads();
    // atomic: two applications of one network may run their backward sweeps concurrently (four-lane schedule) and both add into
    // the network's gradient buffer; the SUM over this launch's slabs above stays in a fixed order
    if (w == 0 && i < n) atomicAdd(&dw[i], sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
}

void vg_launch_reduce_partials(const float* part, int nb, int n, float* dw, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 63) / 64), dim3(256), 0, s, part, nb, n, dw);
}

template <typename T, int RMAX, int Q, bool NOISE>
static void launch_wgrad2(const GatherIn& g, const WgradK& k, dim3 grid, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, RMAX, Q, NOISE>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, RMAX, Q, NOISE>), grid, dim3(256), lds, s, g, k);
}
template <typename T, int RMAX, int Q>
static void launch_wgrad(const GatherIn& g, const WgradK& k, dim3 grid, int lds, hipStream_t s) {
    if (g.noise) launch_wgrad2<T, RMAX, Q, true>(g, k, grid, lds, s); else launch_wgrad2<T, RMAX, Q, false>(g, k, grid, lds, s);
}

extern "C" int vg_conv3d_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host,
                               int T_total, float* dw, float* db, float* scratch, int64_t scratch_bytes,
                               vg_stream_t stream) {
    vg_begin();
    if (!d || !dy || !dw || !tap_idx_host) return VG_EINVAL;
    if (d->Cout < 1 || (d->Cout != 1 && (d->Cout % 8))) return VG_EINVAL;
    if (dy_f32 && d->Cout != 1 && !d->f32) return VG_EINVAL;
    if (d->src0 && tap_idx_host[0] == 0) {
        const int wrc = vg_wgrad_pw_dma(d, dy, dy_f32, T_total, dw, db, scratch, scratch_bytes, (hipStream_t)stream);      // 1x1x1 shortcuts
        if (wrc <= 0) return wrc;
        const int prc = vg_pointwise_wgrad(d, dy, dy_f32, T_total, dw, db, scratch, scratch_bytes, (hipStream_t)stream);
        if (prc <= 0) return prc;
    }
    {   // the thin full-resolution layers (16-channel chunks, 16 output channels): voxel-split waves holding the whole slab (vg_conv_thin.hip)
        const int trc = vg_wgrad_thin(d, dy, dy_f32, tap_idx_host, T_total, dw, db, scratch, scratch_bytes, (hipStream_t)stream);
        if (trc <= 0) return trc;
    }
    {   // materialised operand + LDS-DMA staging (vg_wgrad_dma.hip) where the shape is one of its
        const int drc = vg_wgrad_dma(d, dy, dy_f32, tap_idx_host, T_total, dw, db, scratch, scratch_bytes, (hipStream_t)stream);
        if (drc <= 0) return drc;
    }
    const int Cin = d->c_src0 + d->c_src1;
    const int Cinp = ((Cin + 15) / 16) * 16, Coutp = ((d->Cout + 15) / 16) * 16;
    const int COB = Coutp >= 64 ? 64 : (Coutp >= 32 ? 32 : 16);
    if (Coutp % COB) return VG_EINVAL;
    const int Q = COB / 16;
    int RMAX = Q == 1 ? 24 : (Q == 2 ? 12 : 6);
    const int esz = d->f32 ? 4 : 2;
    GatherIn g; WgradK k;
    int max_bm = vg_tune("WGRAD_BM", 256), max_cib = vg_tune("WGRAD_CIB", 64);
    // 4x4x4 kernels (64 taps): 16-channel chunks.  A workgroup then owns (24 taps x 16 ci) x 64 co instead of (6 taps x 64 ci)
    // x 64 co -- the same accumulators and MFMAs per tile, a quarter of the halo bytes staged for them, and the input is
    // re-staged by ntg x ncob = 3 x 8 workgroup columns instead of 11 x 8 (D.down2: 0.351 -> 0.269 ms; HBM traffic of the
    // noise-carrying weight gradients was 12.8x algorithmic, profiles/r02_roofline_by_kernel.json)
    if (d->ntaps >= 64 && max_cib > 16 && vg_tune("WGRAD_CIB16_K4", 1)) max_cib = 16;
    // 3x3x3 kernels with >= 32 output channels: 16-channel chunks as well, so that ONE workgroup column carries all 27 taps of
    // its 16 input channels (27 rows: the <8,2> variant, or <7,4> below): the input halo is staged once per (ci chunk, co
    // block) instead of once per tap group as well, and no row of MFMAs is spent on padding rows
    // (sweep: dec1.cb1 0.155 -> 0.138 ms, dec3.cb1 0.076 -> 0.067, bridge 0.045 -> 0.039, enc3.cb2 0.054 -> 0.048)
    const bool k3_all_taps = d->ntaps == 27 && Coutp >= 32 && !d->f32 && vg_tune("WGRAD_CIB16", 1);
    if (k3_all_taps && max_cib > 16) max_cib = 16;
    // candidate (BM, CIB) in order of preference: large tile + all channels, LDS <= 80 KiB so that two workgroups fit a CU
    int best_bm = 0, best_cib = 0, best_lds = 0;
    for (int pass = 0; pass < 2 && !best_bm; ++pass) {
        const int limit = pass == 0 ? vg_tune("WGRAD_LDS0", 80 * 1024) : VG_LDS_LIMIT;
        for (int bm = 256; bm >= 64 && !best_bm; bm = bm == 256 ? 128 : bm - 64)
            for (int c = max_cib; c >= 16; c -= 16) {
                if (Cinp % c) continue;
                if (bm == 256 && (Q > 2 || max_bm < 256)) continue;     // dY tile staging holds <= 4 units per thread
                int rc = fill_gather(d, g, c, bm, d->f32 ? 0 : 64);
                if (rc != VG_OK) return rc;
                const int lds = halo_bytes(g) + bm * (COB * esz + 16) + 512 + 2 * c * 4 + stage_table_ints(g) * 4 + (bm / 32) * 512;
                if (lds <= limit) { best_bm = bm; best_cib = c; best_lds = lds; break; }
            }
    }
    if (!best_bm) return VG_ELDS;
    const int CIB = best_cib, lds = best_lds;
    int rc = fill_gather(d, g, CIB, best_bm, d->f32 ? 0 : 64);
    if (rc != VG_OK) return rc;
    k.DYS = COB * esz + 16;
    k.dy = dy; k.dy_f32 = dy_f32; k.Cout = d->Cout; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW;
    k.CIB = CIB; k.COB = COB; k.ncib = Cinp / CIB; k.ncob = Coutp / COB;
    const int rows_per_tap = CIB / 16;
    if (Q == 4 && k3_all_taps && CIB == 16) RMAX = 7;          // 28 rows >= 27 taps: the <7,4> variant
    k.tpg = (4 * RMAX) / rows_per_tap; if (k.tpg < 1) return VG_EINVAL;
    if (k.tpg > d->ntaps) k.tpg = d->ntaps;
    k.ntg = (d->ntaps + k.tpg - 1) / k.tpg;
    k.tpg = (d->ntaps + k.ntg - 1) / k.ntg;                  // balance the tap groups
    for (int i = 0; i < VG_MAX_TAPS; ++i) k.tap_src[i] = i < d->ntaps ? tap_idx_host[i] : 0;
    k.dw = dw; k.db = db;
    k.total_tiles = d->N * g.tiles_d * g.tiles_h * g.tiles_w;
    const int by = k.ntg * k.ncib * k.ncob;
    // persistent grid = resident capacity (2 workgroups per CU for the big-slab variants, 3 for the small ones; LDS)
    const int wg_env = vg_tune("WGRAD_WGS", 0);
    const int rw_ = (k.tpg * rows_per_tap + 3) / 4;
    const int rmax_sel = d->f32 ? RMAX : (rw_ <= 2 ? 2 : (rw_ <= 8 && Q <= 2 ? 8 : (Q == 4 && rw_ == 7 ? 7 : RMAX)));
    int per_cu = (rmax_sel * Q >= VG_WGRAD_2W) ? 2 : 3;
    if (VG_LDS_LIMIT / lds < per_cu) per_cu = VG_LDS_LIMIT / lds;
    if (per_cu < 1) per_cu = 1;
    // 384 persistent workgroups (1.5 per CU), not the resident capacity (512 / 768): the weight gradients run on side streams next
    // to the data-gradient chain of their lane and the other lane's kernels, and a launch that occupies every slot starves those
    // (128^3 train step: 256 -> 30.44 ms, 320 -> 30.52, 384 -> 30.08..30.21, 448 -> 30.56, 512 -> 30.75, 640 -> 30.99)
    const int wg_target = wg_env > 0 ? wg_env : (256 * per_cu < 384 ? 256 * per_cu : 384);
    int bx = wg_target / by; if (bx < 1) bx = 1; if (bx > k.total_tiles) bx = k.total_tiles;
    // many workgroups per dW element: float atomics on a few-KB dW serialise (measured 0.7 ms on a 27 KB dW from 1024
    // workgroups), so each workgroup column stores its slab to a private partial buffer that a second kernel sums
    k.dw_elems = T_total * (d->wpack ? d->wpack : Cin) * d->Cout;     // W-packed single-channel source: T_total = k*k taps of k pseudo-channels
    k.part = nullptr;
    if (bx > 8 && scratch && (int64_t)bx * k.dw_elems * 4 <= scratch_bytes) k.part = scratch;
    const dim3 grid(bx, by, 1);
    hipStream_t s = (hipStream_t)stream;
    // rows per wave actually needed: small slabs (27 taps x 16 channels = 7 rows per wave) take the kernel variant with
    // few accumulators (3 workgroups per CU instead of 2)
    const int rw = (k.tpg * rows_per_tap + 3) / 4;
    {
        const int rsel = d->f32 ? RMAX : (rw <= 2 ? 2 : ((rw <= 8 && Q <= 2) ? 8 : (Q == 4 && rw == 7 ? 7 : RMAX)));
        if (vg_dry("wgrad<%s,%d,%d,n%d>|bm%d|cib%d|part%d|walk%d", d->f32 ? "f32" : "bf16", rsel, Q, g.noise ? 1 : 0, best_bm, CIB,
                   k.part ? 1 : 0, k.total_tiles > bx ? 1 : 0)) return VG_OK;
    }
    if (d->f32) {
        if (Q == 1) launch_wgrad<float, 24, 1>(g, k, grid, lds, s);
        else if (Q == 2) launch_wgrad<float, 12, 2>(g, k, grid, lds, s);
        else launch_wgrad<float, 6, 4>(g, k, grid, lds, s);
    } else if (Q == 1) {
        if (rw <= 2) launch_wgrad<bf16_t, 2, 1>(g, k, grid, lds, s);
        else if (rw <= 8) launch_wgrad<bf16_t, 8, 1>(g, k, grid, lds, s);
        else launch_wgrad<bf16_t, 24, 1>(g, k, grid, lds, s);
    } else if (Q == 2) {
        if (rw <= 2) launch_wgrad<bf16_t, 2, 2>(g, k, grid, lds, s);
        else if (rw <= 8) launch_wgrad<bf16_t, 8, 2>(g, k, grid, lds, s);
        else launch_wgrad<bf16_t, 12, 2>(g, k, grid, lds, s);
    } else {
        if (rw <= 2) launch_wgrad<bf16_t, 2, 4>(g, k, grid, lds, s);
        else if (rw == 7) launch_wgrad<bf16_t, 7, 4>(g, k, grid, lds, s);
        else launch_wgrad<bf16_t, 6, 4>(g, k, grid, lds, s);
    }
    if (k.part) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((k.dw_elems + 63) / 64), dim3(256), 0, s, k.part, bx, k.dw_elems, dw);
    }
    return vg_check_launch();
}

extern "C" int vg_conv3d_wgrad_variant(const vg_conv_desc* d, int dy_f32, const int32_t* tap_idx_host, int T_total,
                                       int64_t scratch_bytes, char* buf, int buflen) {
    if (!buf || buflen < 64) return VG_EINVAL;
    vg_dry_begin(buf, buflen);
    // dummy non-null operands: nothing is launched or dereferenced in a dry run
    const int rc = vg_conv3d_wgrad(d, (const void*)(uintptr_t)0x1000, dy_f32, tap_idx_host, T_total, (float*)(uintptr_t)0x1000,
                                   (float*)(uintptr_t)0x1000, scratch_bytes > 0 ? (float*)(uintptr_t)0x1000 : nullptr, scratch_bytes, nullptr);
    vg_dry_end();
    return rc;
}
