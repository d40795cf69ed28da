// vg_conv_dma.hip -- forward / data-gradient convolution of the WIDE layers with BOTH MFMA operands staged by LDS-DMA (gfx950).
//
//   out[n, o*ostr + ooff, co] (+)= sum_taps sum_ci P[n, o + tap, ci] * W[tap][ci][co]          (stride-1 walk over a padded operand)
//
// Replaces conv32_kernel for Conv3D layers with >= 64 input and a multiple of 64 output channels (discriminator.py:64-117 down0/1/2,
// resunet_model.py:42-66 / 103-143 at the 32^3 .. 8^3 levels) and for their data gradients (tf.GradientTape d/d input,
// vangan.py:426-438): forward of the stride-1 layers, data gradient of the stride-1 AND stride-2 layers (per output-parity class a
// stride-1 walk over dY; all classes in one class-parallel launch).  conv32_kernel staged the activations with ~70 vector
// instructions per 16 bytes at one wave per SIMD and streamed its weight fragments through L1, every wave re-reading every
// fragment (profiles/r03_pmc_down2_*.txt: 11 issue slots per MFMA, 60 % LDS bank-conflict cycles, 9x the algorithmic HBM bytes).
// Here, as in vg_wgrad_dma.hip:
//   * the activation operand is MATERIALISED once per call by materialize_kernel (padding, virtual upsample + concat, InstanceNorm
//     apply, activation, dropout mask, noise resolved there):  P[n][ci / 16][Dp][Hp][Wp][16 ch]  bf16;
//   * the weights are packed (vg_pack_weights_dma) in exactly the order the kernel consumes them:
//         Wd[co / BN][ci / 16][tap][half: ci & 8][BN rows][8 ci]      one (panel, plane, tap group) block is contiguous;
//   * 512 threads = 8 waves (2 per SIMD): 4 along the voxels x 2 along the channels, tile BM = 256 voxels x BN = 128 / 64 channels,
//     v_mfma_f32_32x32x16_bf16, every fragment one conflict-free ds_read_b128; both operands arrive by global_load_lds_dwordx4
//     (no VGPR, no staging VALU), the weight blocks through a ring of 2-3 buffers, the activation image double-buffered per plane.
// Tiles are q-LINEAR: TD consecutive D planes x QT consecutive positions of the row-major (h, w) plane (whole rows in the LDS
// image).  The data gradients iterate over padded / parity-class grids of 17^3, 18^3, 33^3, 34^3 ... positions, which box tiles of
// 8 x 8 x 4 fill to 43-68 %; linear runs fill them to 85-97 %.
// K (= channels x taps) is split over workgroups where a launch has fewer tiles than CUs (deep levels): the slices leave fp32
// partial tiles in the caller's scratch and the slice that arrives last (ticket counter) adds them up in slice order -- the
// exchange of conv_kernel's K split (device-scope relaxed stores / loads, no fence: an agent-scope fence invalidates the XCD's L2
// under the kernels of the other streams).
// Epilogue through LDS: the fp32 tile is laid down [voxel][channel], then every thread handles 8 consecutive channels of a voxel:
// bias, residual * scale + shift, accumulate, ONE rounding to bf16, a 16-byte store (256-byte runs per voxel), and the
// per-(sample, channel) sums of the stored values for the next InstanceNorm.
#include "vg_dma_common.h"

#define VG_CD_MAXA 6          // A (activation image) DMA pieces per wave and plane (8 waves x 6 KiB = 48 KiB per buffer)
#define VG_CD_MAXW 5          // W (weight block) DMA pieces per wave and stage (8 x 5 KiB = 40 KiB per buffer)

typedef __attribute__((ext_vector_type(16))) float f32x16_d;

struct CdCls { const char* w; int tap0, nt, GT, G; int ood, ooh, oow; };
struct CdK {
    const char* P; long n_bytes; int plane_bytes; int HpA, WpA, NPL;
    int N, OD, OH, OW;                   // iteration space (the same for every class)
    int ostr, BD, BH, BW, Cout;
    int ncls; CdCls cls[8];
    int tapoff[VG_MAX_TAPS];             // byte offset of a tap inside one half of the LDS image
    int TD, QT, qtl, cptl, tiles_q, tiles;
    int HHb, HWb, IMG, nvox, nA, abuf;   // LDS image: rows / width of the halo box, 16-byte units per half (multiple of 64), pieces, bytes per buffer
    int wblk, nwb, woff, miscoff;        // bytes reserved per weight buffer, ring depth, LDS offsets
    int ncob, ks, ppk;                   // channel panels, K slices, planes per slice
    int U, upx;                          // units (class, panel, slice, sample, tile), units per XCD label
    const float* bias; const char* res; const float* rs; const float* rb; int accumulate; float* sums;
    char* out;
    float* ks_part; unsigned* ks_cnt;
    unsigned m_ow, m_hw, m_hhw;          // fast_div magics: OW; HWb; HHb * HWb
};

__global__ __launch_bounds__(256) void pack_weights_dma_kernel(const float* __restrict__ w, const int* __restrict__ tap_idx, bf16_t* __restrict__ out,
                                                               int Cin, int Cout, int ntaps, int transpose, int bn) {
    vg_pack_dma_units(w, tap_idx, out, Cin, Cout, ntaps, transpose, bn, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
extern "C" int vg_pack_weights_dma(const float* w, int T, int Cin, int Cout, const int32_t* tap_idx_dev, int ntaps, int transpose, int bn,
                                   void* out, vg_stream_t stream) {
    vg_begin();
    if (!w || !tap_idx_dev || !out || ntaps < 1 || ntaps > T || (bn != 64 && bn != 128)) return VG_EINVAL;
    const int NR = transpose ? Cin : Cout, C = transpose ? Cout : Cin;
    if ((NR % bn) || (C % 16)) return VG_EINVAL;
    const long units = (long)NR * (C / 16) * ntaps * 2;
    int blocks = (int)((units + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_dma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (const int*)tap_idx_dev, (bf16_t*)out, Cin, Cout,
                       ntaps, transpose, bn);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// the kernel: NW = 32-channel blocks per wave (BN = 64 * NW), MW = 32-voxel sub-tiles per wave (BM = 128 * MW)
//   acc layout (32x32x16, A = weights, B = activations): lane l holds voxel (l & 31), channels 8*jj + 4*(l >> 5) + r  (acc[4*jj + r])
// ------------------------------------------------------------------------------------------------------------------
template <int NW, int MW>
__global__ __launch_bounds__(512, 2) void conv_dma_kernel(const CdK p) {
    constexpr int BN = 64 * NW, BM = 128 * MW;
    constexpr int PITCH = BN * 4 + 16;                     // bytes of one voxel row of the fp32 epilogue tile (16-byte skew: conflict-free 16-byte stores)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lv = lane & 31, lk = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_d*)smem;
    int* vtab = (int*)(smem + p.miscoff);                  // [BM]: voxel index of a tile voxel in the output buffer, -1 outside the grid
    float* stat = (float*)(vtab + BM);                     // [BN][2]: this workgroup's sums of the current unit
    int* flag = (int*)(stat + BN * 2);
    int* tapL = flag + 16;                                 // [VG_MAX_TAPS]: p.tapoff in LDS (a scalar load per tap would be waited for in front of
                                                           // every fragment read; LDS returns in order, so an offset read issued one step ahead is free)
    if (tid < VG_MAX_TAPS) tapL[tid] = p.tapoff[tid];      // published by the first unit's first barrier

    // ---- per-lane source offsets of the activation image's DMA pieces: the halo box has the same shape for every tile ----
    int aoffs[VG_CD_MAXA];
    const int pph = p.IMG >> 6;                            // pieces per half
#pragma unroll
    for (int k = 0; k < VG_CD_MAXA; ++k) {
        const int piece = wave + 8 * k;
        const int half = piece >= pph ? 1 : 0;
        int idx = (piece - half * pph) * 64 + lane;
        if (idx >= p.nvox) idx = 0;
        const int hd = fast_div(idx, p.m_hhw), rem = idx - hd * (p.HHb * p.HWb);
        const int hh = fast_div(rem, p.m_hw), hw = rem - hh * p.HWb;
        aoffs[k] = ((hd * p.HpA + hh) * p.WpA + hw) * 32 + half * 16;
    }
    const int n_a = wave < p.nA ? (p.nA - wave + 7) >> 3 : 0;               // this wave's A pieces per image
    const int wlane = (lk * BN + wn * NW * 32 + lv) * 16;                     // this lane's weight fragment inside a tap's [half][row] block
    const int CPT = 1 << p.cptl;                                              // 32-voxel chunks per D plane of a tile

    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, JX = gridDim.x >> 3;
    const int u_lo = xcd * p.upx, u_hi = min(u_lo + p.upx, p.U);
    for (int u = u_lo + jx; u < u_hi; u += JX) {
        // ---- unit -> (class, panel, slice, sample, tile) ----
        int t = u;
        const int tile = t % p.tiles; t /= p.tiles;
        const int n = t % p.N; t /= p.N;
        const int slice = t % p.ks; t /= p.ks;
        const int cob = t % p.ncob; const int ci = t / p.ncob;
        const CdCls& c = p.cls[ci];
        const int td_i = tile / p.tiles_q, qr = tile - td_i * p.tiles_q;
        const int d0 = td_i * p.TD, q0 = qr << p.qtl, h0 = fast_div(q0, p.m_ow);
        const int p_lo = slice * p.ppk, npl = p.ppk;
        const int GT = c.GT, G = c.G, nst = npl * G;
        const int nWp = (GT * BN) >> 5;                                       // 1-KiB pieces of one weight block
        const int n_w = wave < nWp ? (nWp - wave + 7) >> 3 : 0;
        const int wtap = 2 * BN * 16;                                         // bytes of one tap of a weight block
        const char* pbase = p.P + (size_t)n * p.n_bytes + (size_t)((d0 * p.HpA + h0) * p.WpA) * 32;
        const char* wbase = c.w + (size_t)cob * p.NPL * c.nt * wtap;

        // ---- lane geometry of this tile; output table ----
        int abase[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int sidx = wm * MW + i, dl = sidx >> p.cptl, ch = sidx & (CPT - 1);
            const int q = q0 + ch * 32 + lv;
            int h = fast_div(q, p.m_ow), w = q - h * p.OW;
            if (q >= p.OH * p.OW) { h = h0; w = 0; }
            abase[i] = ((dl * p.HHb + (h - h0)) * p.HWb + w) * 16 + lk * p.IMG * 16;
        }
        __syncthreads();                                   // the previous unit's epilogue has read vtab / stat / the LDS tile
        if (tid < BM) {
            const int dl = tid >> p.qtl, q = q0 + (tid & (p.QT - 1)), d = d0 + dl;
            const int h = fast_div(q, p.m_ow), w = q - h * p.OW;
            const bool ok = d < p.OD && q < p.OH * p.OW;
            vtab[tid] = ok ? ((n * p.BD + d * p.ostr + c.ood) * p.BH + h * p.ostr + c.ooh) * p.BW + w * p.ostr + c.oow : -1;
        }
        if (tid < BN * 2) stat[tid] = 0.f;

        auto issueA = [&](int pl, int buf) {
            const char* b = pbase + (size_t)pl * p.plane_bytes;
#pragma unroll
            for (int k = 0; k < VG_CD_MAXA; ++k) {
                const int piece = wave + 8 * k;
                if (piece < p.nA) glds16(b, aoffs[k], lds0 + buf * p.abuf + piece * 1024);
            }
        };
        auto issueW = [&](int st, int buf) {
            const int pl = st / G, g = st - pl * G;
            const char* b = wbase + ((size_t)(p_lo + pl) * c.nt + g * GT) * wtap;
#pragma unroll
            for (int k = 0; k < VG_CD_MAXW; ++k) {
                const int piece = wave + 8 * k;
                if (piece < nWp) glds16(b, piece * 1024 + lane * 16, lds0 + p.woff + buf * p.wblk + piece * 1024);
            }
        };
        const int ahead = p.nwb - 1;
        issueA(p_lo, 0);
        for (int s2 = 0; s2 < ahead && s2 < nst; ++s2) issueW(s2, s2);

        f32x16_d acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

        // ---- stages: (plane, tap group).  A stage issues the weight block `ahead` stages on (and, in the first stage of a plane, the
        // next plane's image) right after its barrier, then multiplies.  Copies complete in issue order; `allow` = this wave's copies
        // that were issued after the ones this stage needs. ----
        int pl = 0, g = 0, wb = 0;
        for (int st = 0; st < nst; ++st) {
            // Copies complete in issue order.  Issue order: prologue A(plane 0), W(0) .. W(ahead - 1); stage j: [A(next plane) in the first
            // stage of a plane], then W(j + ahead).  This stage needs W(st) and the image of its plane; `allow` = this wave's copies issued
            // after the younger of the two (walk back over the stages that issued since).
            int allow = 0;
            {
                bool stopped = false;
                int pj = pl, gj = g;
                for (int j2 = st - 1; j2 > st - ahead && j2 >= 0; --j2) {
                    if (gj == 0) { gj = G - 1; --pj; } else --gj;                 // (plane, group) of stage j2
                    const bool a_j = gj == 0 && pj + 1 < npl, w_j = j2 + ahead < nst;
                    if (a_j && pj + 1 == pl) { allow += w_j ? n_w : 0; stopped = true; break; }      // that image is this stage's: it must have landed
                    allow += (a_j ? n_a : 0) + (w_j ? n_w : 0);
                }
                if (!stopped && st < ahead)
                    for (int w2 = st + 1; w2 < ahead; ++w2) if (w2 < nst) allow += n_w;             // the prologue's later weight blocks
            }
            wait_vmcnt(allow);
            __syncthreads();                                                  // everybody's copies have landed; the buffers rewritten below are no longer read
            if (g == 0 && pl + 1 < npl) issueA(p_lo + pl + 1, (pl + 1) & 1);
            if (st + ahead < nst) { int nb = wb + ahead; if (nb >= p.nwb) nb -= p.nwb; issueW(st + ahead, nb); }
            const char* ab = smem + (pl & 1) * p.abuf;
            const char* wq = smem + p.woff + wb * p.wblk + wlane;
            const int* tl = tapL + c.tap0 + g * GT;
            {
                bf16x8 A[2][MW], W[2][NW];
                const int last = GT - 1;
                int o1 = tl[min(1, last)];
                {
                    const int to = tl[0];
#pragma unroll
                    for (int i = 0; i < MW; ++i) A[0][i] = *(const bf16x8*)(ab + abase[i] + to);
#pragma unroll
                    for (int jn = 0; jn < NW; ++jn) W[0][jn] = *(const bf16x8*)(wq + jn * 512);
                }
                for (int tt = 0; tt < GT; tt += 2) {
                    const int o2 = tl[min(tt + 2, last)];                     // offsets run one step ahead of the fragments that use them
                    {
                        const int t1 = min(tt + 1, last);
#pragma unroll
                        for (int i = 0; i < MW; ++i) A[1][i] = *(const bf16x8*)(ab + abase[i] + o1);
#pragma unroll
                        for (int jn = 0; jn < NW; ++jn) W[1][jn] = *(const bf16x8*)(wq + t1 * wtap + jn * 512);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int jn = 0; jn < NW; ++jn) acc[i][jn] = VG_MFMA32(W[0][jn], A[0][i], acc[i][jn]);
                    __builtin_amdgcn_sched_barrier(0);
                    o1 = tl[min(tt + 3, last)];
                    {
                        const int t2 = min(tt + 2, last);
#pragma unroll
                        for (int i = 0; i < MW; ++i) A[0][i] = *(const bf16x8*)(ab + abase[i] + o2);
#pragma unroll
                        for (int jn = 0; jn < NW; ++jn) W[0][jn] = *(const bf16x8*)(wq + t2 * wtap + jn * 512);
                    }
                    if (tt + 1 >= GT) {                                       // odd tap count: the phantom step adds zero
#pragma unroll
                        for (int jn = 0; jn < NW; ++jn) W[1][jn] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int jn = 0; jn < NW; ++jn) acc[i][jn] = VG_MFMA32(W[1][jn], A[1][i], acc[i][jn]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (++g == G) { g = 0; ++pl; }
            if (++wb == p.nwb) wb = 0;
        }

        // ---- K split: leave the partial tile in scratch; the slice that arrives last adds all of them in slice order ----
        if (p.ks > 1) {
            constexpr int SLOT = 512 * MW * NW * 16;
            const size_t cell = ((size_t)(ci * p.ncob + cob) * p.N + n) * p.tiles + tile;
            float* slot = p.ks_part + (cell * p.ks + slice) * SLOT;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __hip_atomic_store(slot + ((i * NW + jn) * 16 + r) * 512 + tid, acc[i][jn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this thread's partial is written through ...
            __syncthreads();                                                  // ... every thread's, before the ticket
            if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.ks_cnt + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == p.ks - 1;
            if (!last) continue;
            const float* base = p.ks_part + cell * p.ks * SLOT;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][jn][r] = __hip_atomic_load(base + ((i * NW + jn) * 16 + r) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int sl = 1; sl < p.ks; ++sl)
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][jn][r] += __hip_atomic_load(base + (size_t)sl * SLOT + ((i * NW + jn) * 16 + r) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(p.ks_cnt + cell, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // ready for the next launch on this stream
        }

        // ---- epilogue: fp32 tile -> LDS [voxel][channel]; then 8 channels of one voxel per thread ----
        __syncthreads();                                   // the last stage's fragments have been read (the tile overlays the operand buffers)
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int v = (wm * MW + i) * 32 + lv;
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ch = (wn * NW + jn) * 32 + 8 * jj + 4 * lk;
                    *(f32x4*)(smem + v * PITCH + ch * 4) = (f32x4){acc[i][jn][4 * jj], acc[i][jn][4 * jj + 1], acc[i][jn][4 * jj + 2], acc[i][jn][4 * jj + 3]};
                }
        }
        __syncthreads();
        {
            constexpr int NCG = BN / 8, NVS = 512 / NCG;                     // channel groups of 8; voxel slots
            const int cg = tid % NCG, vs = tid / NCG;
            const int co = cob * BN + cg * 8;
            float b8[8], rs8[8], rb8[8], s1[8], s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                b8[e] = p.bias ? p.bias[co + e] : 0.f;
                rs8[e] = p.res ? p.rs[n * p.Cout + co + e] : 0.f;
                rb8[e] = p.res ? p.rb[n * p.Cout + co + e] : 0.f;
                s1[e] = 0.f; s2[e] = 0.f;
            }
            for (int v = vs; v < BM; v += NVS) {
                const int idx = vtab[v];
                if (idx < 0) continue;
                const f32x4 x0 = *(const f32x4*)(smem + v * PITCH + cg * 32), x1 = *(const f32x4*)(smem + v * PITCH + cg * 32 + 16);
                float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                const size_t o = (size_t)idx * p.Cout + co;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += b8[e];
                if (p.res) {
                    float rr[8]; load8<bf16_t>((const bf16_t*)p.res + o, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += rr[e] * rs8[e] + rb8[e];
                }
                if (p.accumulate) {
                    float oo[8]; load8<bf16_t>((const bf16_t*)p.out + o, oo);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += oo[e];
                }
                bf16x8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = (short)f2bf(x[e]);
                *(bf16x8*)((bf16_t*)p.out + o) = pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float y = bf2f((bf16_t)pk[e]); s1[e] += y; s2[e] += y * y; }
            }
            if (p.sums) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { atomicAdd(&stat[(cg * 8 + e) * 2], s1[e]); atomicAdd(&stat[(cg * 8 + e) * 2 + 1], s2[e]); }
                __syncthreads();
                if (tid < BN * 2) {
                    const int stripe = blockIdx.x & (VG_STRIPES - 1);
                    atomicAdd(&p.sums[(((size_t)stripe * p.N + n) * p.Cout + cob * BN + (tid >> 1)) * 2 + (tid & 1)], stat[tid]);
                }
            }
        }
    }
}

template <int NW, int MW>
static void launch_cd(const CdK& k, int grid, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<NW, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_dma_kernel<NW, MW>), dim3(grid), dim3(512), lds, s, k);
}

// ------------------------------------------------------------------------------------------------------------------
// host side: eligibility, plan, operand pass, launch
// ------------------------------------------------------------------------------------------------------------------
struct CdPlan { int BN, TD, QT, HD, HHb, HWb, IMG, nvox, nA, abuf, wblk, nwb, lds, tiles_d, tiles_q, DpA, HpA, WpA, mn[3], ex[3]; int ncls; int nt[8], GT[8]; };

// Shape-only part of the decision (no pointers looked at): fills the plan, returns the channel panel BN (64 / 128) or 0.
static int cd_plan(const vg_conv_desc* d, CdPlan& pl) {
    if (!d || !vg_tune("CONV_DMA_FAMILY", 1)) return 0;
    const int Cin = d->c_src0 + d->c_src1;
    if (d->f32 || d->src_f32 || d->wpack || d->tanh_out || d->out_f32 || d->istr != 1) return 0;
    if (Cin < vg_tune("CONV_DMA_MINCIN", 64) || (Cin % 16) || d->Cout < 64 || (d->Cout % 64)) return 0;
    if (d->c_src1 > 0 && ((d->c_src0 % 8) || (d->c_src1 % 8))) return 0;
    if (d->ntaps < 1 || d->ntaps > VG_MAX_TAPS || d->N < 1 || d->OD < 1 || d->OH < 1 || d->OW < 1) return 0;
    if (d->ostr < 1 || d->ostr > 2) return 0;
    if (d->src0_shift && ((d->D | d->H | d->W) & 1)) return 0;
    const int BN = (d->Cout % 128 == 0 && vg_tune("CONV_DMA_BN", 128) >= 128) ? 128 : 64;
    pl.BN = BN;
    pl.ncls = d->nclass > 1 ? d->nclass : 1;
    if (pl.ncls > 8) return 0;
    if (pl.ncls > 1) {
        if (d->cls_tap0[0] != 0 || d->cls_tap0[pl.ncls] != d->ntaps) return 0;
        for (int c = 0; c < pl.ncls; ++c) {
            pl.nt[c] = d->cls_tap0[c + 1] - d->cls_tap0[c];
            if (pl.nt[c] < 1) return 0;
            for (int a = 0; a < 3; ++a) if (d->cls_iters[c][a] != (a == 0 ? d->OD : (a == 1 ? d->OH : d->OW))) return 0;     // one tile grid for all classes
        }
    } else pl.nt[0] = d->ntaps;
    if (d->ntaps < 8) return 0;                            // 1x1x1 shortcuts stay on the HBM-bound pointwise kernels
    for (int a = 0; a < 3; ++a) { pl.mn[a] = 127; pl.ex[a] = -128; }
    for (int i = 0; i < d->ntaps; ++i) {
        const int v[3] = {d->tap_d[i], d->tap_h[i], d->tap_w[i]};
        for (int a = 0; a < 3; ++a) { if (v[a] < pl.mn[a]) pl.mn[a] = v[a]; if (v[a] > pl.ex[a]) pl.ex[a] = v[a]; }
    }
    for (int a = 0; a < 3; ++a) pl.ex[a] = pl.ex[a] - pl.mn[a] + 1;
    if (d->pad_mode == VG_PAD_REFLECT) {         // one reflection only (materialize_kernel)
        const int od[3] = {d->OD, d->OH, d->OW}, nn[3] = {d->D, d->H, d->W};
        for (int a = 0; a < 3; ++a) if (pl.mn[a] < -(nn[a] - 1) || (od[a] - 1) + pl.mn[a] + pl.ex[a] - 1 > 2 * nn[a] - 2) return 0;
    }
    // tile: TD planes x QT positions of the (h, w) plane, TD * QT = 256; fewest tiles, then the smallest image.  Taps per stage GT
    // (a divisor of the tap count, <= 9; a class of <= 9 taps is one stage per plane) and the depth of the weight ring: the ring must
    // cover the copies' latency, (nwb - 1) * GT >= ~12 taps of MFMA time, with stages as long as the LDS allows.
    const int BM = 256, plane = d->OH * d->OW;
    const int tile_bytes = BM * (BN * 4 + 16), misc = BM * 4 + BN * 8 + 64 + VG_MAX_TAPS * 4;
    const int wtap = 2 * BN * 16;
    int ntmax = 0; bool small = true;
    for (int c = 0; c < pl.ncls; ++c) { if (pl.nt[c] > ntmax) ntmax = pl.nt[c]; if (pl.nt[c] > 9) small = false; }
    if (!small && pl.ncls > 1) {                           // several classes with more than 9 taps: they must agree on GT
        for (int c = 1; c < pl.ncls; ++c) if (pl.nt[c] != pl.nt[0]) return 0;
    }
    long best = -1;
    for (int TD = 1; TD <= 8; TD <<= 1) {
        const int QT = BM / TD;
        if (vg_tune("CONV_DMA_TD", 0) && TD != vg_tune("CONV_DMA_TD", 0)) continue;
        int R = (QT - 1 + d->OW - 1) / d->OW + 1; if (R > d->OH) R = d->OH;
        const int HD = TD + pl.ex[0] - 1, HHb = R + pl.ex[1] - 1, HWb = d->OW + pl.ex[2] - 1;
        const int nvox = HD * HHb * HWb, IMG = ((nvox + 63) / 64) * 64;
        const int nA = 2 * IMG / 64;
        if (nA > 8 * VG_CD_MAXA) continue;
        const int abuf = 2 * IMG * 16;
        int bGT = 0, bnwb = 0; long bq = -1;
        for (int GT = small ? ntmax : 9; GT >= 1; --GT) {
            if (!small && (ntmax % GT)) continue;
            if (GT * BN / 32 > 8 * VG_CD_MAXW) continue;
            int nwb = (VG_LDS_LIMIT - misc - 2 * abuf) / (GT * wtap); if (nwb > 8) nwb = 8;
            if (vg_tune("CONV_DMA_NWB", 0) && nwb > vg_tune("CONV_DMA_NWB", 0)) nwb = vg_tune("CONV_DMA_NWB", 0);
            if (nwb < 2) { if (small) break; continue; }
            const int cover = (nwb - 1) * GT;
            const long q = (cover >= 12 ? 1000 : 0) + (cover >= 12 ? GT * 10 : cover * 10) + (nwb <= 4 ? 1 : 0);
            if (q > bq) { bq = q; bGT = GT; bnwb = nwb; }
            if (small) break;                               // the classes' own tap counts are the stages
        }
        if (!bGT) continue;
        if (bnwb > 4 && (bnwb - 1) * bGT > 24) bnwb = 24 / bGT + 1;          // no deeper than useful
        if (bnwb < 2) bnwb = 2;
        int body = 2 * abuf + bnwb * bGT * wtap; if (body < tile_bytes) body = tile_bytes;
        if (body + misc > VG_LDS_LIMIT) continue;
        const int tiles_d = (d->OD + TD - 1) / TD, tiles_q = (plane + QT - 1) / QT;
        const long score = (long)tiles_d * tiles_q * 100000 + nvox + ((bnwb - 1) * bGT < 12 ? 40000 : 0);
        if (best < 0 || score < best) {
            best = score;
            pl.TD = TD; pl.QT = QT; pl.HD = HD; pl.HHb = HHb; pl.HWb = HWb; pl.IMG = IMG; pl.nvox = nvox; pl.nA = nA; pl.abuf = abuf;
            pl.wblk = bGT * wtap; pl.nwb = bnwb; pl.lds = body + misc; pl.tiles_d = tiles_d; pl.tiles_q = tiles_q;
            for (int c = 0; c < pl.ncls; ++c) pl.GT[c] = small ? pl.nt[c] : bGT;
            pl.DpA = tiles_d * TD + pl.ex[0] - 1;
            pl.HpA = ((tiles_q - 1) * QT) / d->OW + HHb; if (pl.HpA < d->OH + pl.ex[1] - 1) pl.HpA = d->OH + pl.ex[1] - 1;
            pl.WpA = HWb;
        }
    }
    if (best < 0) return 0;
    if ((int64_t)pl.DpA * pl.HpA * pl.WpA * 32 >= (1LL << 31)) return 0;
    return BN;
}

extern "C" int vg_conv3d_dma_bn(const vg_conv_desc* d) {
    vg_begin();
    CdPlan pl;
    return cd_plan(d, pl);
}

// VG_OK: served; 1: not one of this family's shapes (only possible when d->wlayout == 0); < 0: error
int vg_conv_dma(const vg_conv_desc* d, hipStream_t s) {
    CdPlan pl;
    const int BN = cd_plan(d, pl);
    if (!BN) return d->wlayout ? VG_EINVAL : 1;
    if (!d->wlayout) return 1;                             // weights in the classic layout: the caller packed for the other kernels
    if (d->wlayout != BN) return VG_EINVAL;               // packed for another panel width
    if (!d->out || !d->src0 || !d->scratch) return VG_EINVAL;
    if (d->res && (!d->res_scale || !d->res_shift)) return VG_EINVAL;
    if (pl.ncls == 1 && !d->wpacked) return VG_EINVAL;
    const int Cin = d->c_src0 + d->c_src1, NPL = Cin / 16, ncob = d->Cout / BN;
    const int tiles = pl.tiles_d * pl.tiles_q;
    const int64_t plane_bytes = (int64_t)pl.DpA * pl.HpA * pl.WpA * 32;
    const int64_t p_bytes = ((plane_bytes * NPL * d->N + 255) / 256) * 256;
    // K split: aim at ~CONV_DMA_WGS workgroups (the other lane and the weight-gradient streams fill the rest of the chip)
    const long units0 = (long)pl.ncls * ncob * d->N * tiles;
    const long target = vg_tune("CONV_DMA_WGS", 128);
    const int64_t slot_bytes = 512LL * 2 * (BN / 64) * 16 * 4;
    int ks = 1;
    const int f_ks = vg_tune("CONV_DMA_KS", 0);
    for (int c = 1; c <= NPL; ++c) {
        if (NPL % c) continue;
        if (c > 1 && (units0 > VG_SCRATCH_CTR_BYTES / 4 || VG_SCRATCH_CTR_BYTES + p_bytes + units0 * c * slot_bytes > d->scratch_bytes)) break;
        ks = c;
        if (f_ks ? c >= f_ks : units0 * c >= target) break;
    }
    if (VG_SCRATCH_CTR_BYTES + p_bytes > d->scratch_bytes) return VG_EINVAL;
    CdK k;
    k.P = (const char*)d->scratch + VG_SCRATCH_CTR_BYTES; k.n_bytes = plane_bytes * NPL; k.plane_bytes = (int)plane_bytes;
    k.HpA = pl.HpA; k.WpA = pl.WpA; k.NPL = NPL;
    k.N = d->N; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW;
    k.ostr = d->ostr; k.BD = d->BD; k.BH = d->BH; k.BW = d->BW; k.Cout = d->Cout;
    k.ncls = pl.ncls;
    for (int c = 0; c < pl.ncls; ++c) {
        CdCls& q = k.cls[c];
        q.w = (const char*)(pl.ncls > 1 ? d->cls_w[c] : d->wpacked);
        if (!q.w) return VG_EINVAL;
        q.tap0 = pl.ncls > 1 ? d->cls_tap0[c] : 0; q.nt = pl.nt[c]; q.GT = pl.GT[c]; q.G = pl.nt[c] / pl.GT[c];
        q.ood = pl.ncls > 1 ? d->cls_ooff[c][0] : d->ooff_d; q.ooh = pl.ncls > 1 ? d->cls_ooff[c][1] : d->ooff_h; q.oow = pl.ncls > 1 ? d->cls_ooff[c][2] : d->ooff_w;
    }
    for (int c = pl.ncls; c < 8; ++c) k.cls[c] = k.cls[0];
    for (int i = 0; i < VG_MAX_TAPS; ++i)
        k.tapoff[i] = i < d->ntaps ? (((d->tap_d[i] - pl.mn[0]) * pl.HHb + (d->tap_h[i] - pl.mn[1])) * pl.HWb + (d->tap_w[i] - pl.mn[2])) * 16 : 0;
    k.TD = pl.TD; k.QT = pl.QT; k.qtl = ilog2_exact(pl.QT); k.cptl = ilog2_exact(pl.QT / 32); k.tiles_q = pl.tiles_q; k.tiles = tiles;
    k.HHb = pl.HHb; k.HWb = pl.HWb; k.IMG = pl.IMG; k.nvox = pl.nvox; k.nA = pl.nA; k.abuf = pl.abuf;
    k.wblk = pl.wblk; k.nwb = pl.nwb; k.woff = 2 * pl.abuf;
    k.miscoff = pl.lds - (256 * 4 + BN * 8 + 64 + VG_MAX_TAPS * 4);
    k.ncob = ncob; k.ks = ks; k.ppk = NPL / ks;
    const long U = units0 * ks;
    if (U >= (1L << 30)) return VG_EINVAL;
    k.U = (int)U; k.upx = (int)((U + 7) / 8);
    k.bias = d->bias; k.res = (const char*)d->res; k.rs = d->res_scale; k.rb = d->res_shift; k.accumulate = d->accumulate; k.sums = d->out_sums;
    k.out = (char*)d->out;
    k.ks_cnt = (unsigned*)d->scratch; k.ks_part = (float*)((char*)d->scratch + VG_SCRATCH_CTR_BYTES + p_bytes);
    auto magic = [](int dd) { return dd <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)dd + 1ULL); };
    k.m_ow = magic(d->OW); k.m_hw = magic(pl.HWb); k.m_hhw = magic(pl.HHb * pl.HWb);
    int per_x = vg_tune("CONV_DMA_GRID", 256) / 8; if (per_x < 1) per_x = 1; if (per_x > k.upx) per_x = k.upx;
    const int grid = 8 * per_x;
    if (vg_dry("conv_dma<%d,%d>|td%d|gt%d|nwb%d|ks%d|cls%d|walk%d", BN, 256, pl.TD, pl.GT[0], pl.nwb, ks > 1 ? 1 : 0, pl.ncls > 1 ? 1 : 0, U > grid ? 1 : 0)) return VG_OK;
    MatK m;
    m.src0 = d->src0; m.src1 = d->src1; m.c0 = d->c_src0; m.c1 = d->c_src1; m.shift0 = d->src0_shift ? 1 : 0;
    m.N = d->N; m.D = d->D; m.H = d->H; m.W = d->W; m.Cin = Cin;
    m.in_scale = d->in_scale; m.in_shift = d->in_shift; m.act = d->act;
    m.noise = (const bf16_t*)d->noise; m.npad = d->noise ? d->noise_pad : 0; m.pad_mode = d->pad_mode;
    m.pmin_d = pl.mn[0]; m.pmin_h = pl.mn[1]; m.pmin_w = pl.mn[2]; m.Dp = pl.DpA; m.Hp = pl.HpA; m.Wp = pl.WpA;
    m.deint = 0; m.WE = 0; m.Wps = pl.WpA; m.out = (bf16_t*)k.P;
    vg_launch_materialize(m, s);
    if (BN == 128) launch_cd<2, 2>(k, grid, pl.lds, s); else launch_cd<1, 2>(k, grid, pl.lds, s);
    return vg_check_launch();
}
