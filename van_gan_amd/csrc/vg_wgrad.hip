// vg_wgrad.hip -- weight gradient of the gather-convolution on bf16 MFMA (gfx950).
//
// dW[tap][ci][co] += sum_{n,o} P[n, bnd(o*istr+tap), ci] * dY[n,o,co],   db[co] += sum dY
// (what tf.GradientTape returns for the Conv3D kernels/biases on the path, vangan.py:426-438), with
// P = the same on-read transformed operand the forward uses (vg_gather.h).
//
// GEMM view: M = ci, N = co, K = voxels.  Both operands are stored [voxel][channel] (channel innermost),
// i.e. K is the SLOW axis of both, so the MFMA fragments (8 consecutive k per lane) are fetched with the
// gfx950 transposing LDS read ds_read_b64_tr_b16: 4 voxel rows x 16 channels per 16-lane group, delivered
// column-major.  A workgroup owns a (tap-group, ci-block, co-block) slab of dW and walks a strided set of
// voxel tiles, keeping the slab in accumulators; it adds the slab to dW with fp32 atomics once at the end.
#include "vg_gather.h"

struct WgradK {
    const void* dy; int dy_f32; int Cout;
    int OD, OH, OW;
    int CIB, COB, ncib, ncob, tpg, ntg;     // ci/co block sizes, counts, taps per group, tap groups
    int tap_src[VG_MAX_TAPS];               // packed tap -> source tap index in dW
    float* dw; float* db;
    int total_tiles, DYS;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const char* base0, const char* base1) {
    // two transposed 4x16 block reads -> 8 consecutive k for this lane's column
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)base0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)base1);
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <typename T, int MAXI>
__global__ __launch_bounds__(256) void wgrad_kernel(const GatherIn g, const WgradK p) {
    constexpr bool F32 = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, li = lane & 15;
    // blockIdx.y -> (tap group, ci block, co block)
    int by = blockIdx.y;
    const int cob = by % p.ncob; by /= p.ncob;
    const int cib = by % p.ncib; const int tg = by / p.ncib;
    const int tap0 = tg * p.tpg;
    const int ntap_here = min(p.tpg, g.ntaps - tap0);
    const int tci = p.CIB >> 4, tco = p.COB >> 4;
    const int nitems = ntap_here * tci * tco;

    const int BM = 1 << (g.tdl + g.thl + g.twl);
    char* halo = smem;
    const int hbytes = g.HD * g.HH * g.HW * g.RS;
    char* dyt = smem + hbytes;
    const int dybytes = BM * p.DYS;
    int* tapoff = (int*)(dyt + dybytes);
    int* tapsrc = tapoff + 64;
    float* scs = (float*)((char*)tapoff + 512);

    if (tid < g.ntaps)
        tapoff[tid] = (((g.td[tid] - g.tmin_d) * g.HH + (g.th[tid] - g.tmin_h)) * g.HW + (g.tw[tid] - g.tmin_w)) * g.RS;
    if (tid < VG_MAX_TAPS) tapsrc[tid] = p.tap_src[tid];

    // per-wave items: item = wave + 4*j -> (tap, ci16, co16)
    int it_tap[MAXI], it_ci[MAXI], it_co[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        int item = wave + 4 * j;
        if (item >= nitems) item = nitems - 1;         // idle slot: computed but not written
        it_co[j] = item % tco; item /= tco;
        it_ci[j] = item % tci; it_tap[j] = tap0 + item / tci;
    }
    f32x4 acc[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbsum = 0.f;

    const int TWm = (1 << g.twl) - 1, THm = (1 << g.thl) - 1;
    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    const int gco = p.COB >> 3;                 // 8-channel groups per dY row
    const bool do_db = p.db && cib == 0 && tg == 0;

    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n; int t = tile - n * tiles_per_n;
        const int tw_i = t % g.tiles_w; t /= g.tiles_w;
        const int th_i = t % g.tiles_h; const int td_i = t / g.tiles_h;
        const int od0 = td_i << g.tdl, oh0 = th_i << g.thl, ow0 = tw_i << g.twl;
        __syncthreads();
        stage_scale_shift(g, scs, n, cib, tid);
        __syncthreads();
        stage_halo<T>(g, halo, scs, n, od0, oh0, ow0, cib, tid, 256);
        // ---- stage dY tile [BM][COB] (zero outside the grid / beyond Cout) ----
        for (int u = tid; u < BM * gco; u += 256) {
            const int m = u / gco, cg = u - m * gco;
            const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
            const int od = od0 + d, oh = oh0 + h, ow = ow0 + w;
            const int c = cob * p.COB + cg * 8;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (od < p.OD && oh < p.OH && ow < p.OW && c < p.Cout) {
                const size_t vox = ((size_t)(n * p.OD + od) * p.OH + oh) * p.OW + ow;
                if (p.Cout == 1) v[0] = p.dy_f32 ? ((const float*)p.dy)[vox] : bf2f(((const bf16_t*)p.dy)[vox]);
                else load8<T>((const T*)p.dy + vox * p.Cout + c, v);
            }
            store8<T>((T*)(dyt + (size_t)m * p.DYS) + cg * 8, v);
        }
        __syncthreads();
        if (do_db) {      // thread (row group, channel): partial column sums of the dY tile
            const int nrg = 256 / p.COB;
            if (tid < nrg * p.COB) {
                const int c = tid % p.COB;
                float s = 0.f;
                for (int m = tid / p.COB; m < BM; m += nrg) s += ld1<T>((const T*)(dyt + (size_t)m * p.DYS) + c);
                dbsum += s;
            }
        }
        if constexpr (F32) {
            // exact-parity mode: v_mfma_f32_16x16x4_f32, k = 4 voxels; lane (lg, li): A[ci=li][k=lg], B[k=lg][co=li]
            for (int s = 0; s < BM / 4; ++s) {
                const int m0 = s * 4 + lg;
                const int w0 = m0 & TWm, h0 = (m0 >> g.twl) & THm, d0 = m0 >> (g.twl + g.thl);
                const int r0 = ((d0 * g.istr * g.HH + h0 * g.istr) * g.HW + w0 * g.istr) * g.RS;
                const char* y0 = dyt + (size_t)m0 * p.DYS;
#pragma unroll
                for (int j = 0; j < MAXI; ++j) {
                    const float a = *(const float*)(halo + r0 + tapoff[it_tap[j]] + (it_ci[j] * 16 + li) * 4);
                    const float b = *(const float*)(y0 + (it_co[j] * 16 + li) * 4);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                }
            }
            continue;
        }
        // ---- K loop over voxels, 32 per MFMA ----
        for (int s = 0; s < BM / 32; ++s) {
            // this lane supplies the address of voxel row m0 (and m0+4) for the transposed block reads
            const int m0 = s * 32 + 8 * lg + (li >> 2);
            const int m1 = m0 + 4;
            const int w0 = m0 & TWm, h0 = (m0 >> g.twl) & THm, d0 = m0 >> (g.twl + g.thl);
            const int w1 = m1 & TWm, h1 = (m1 >> g.twl) & THm, d1 = m1 >> (g.twl + g.thl);
            const int r0 = ((d0 * g.istr * g.HH + h0 * g.istr) * g.HW + w0 * g.istr) * g.RS + 8 * (li & 3);
            const int r1 = ((d1 * g.istr * g.HH + h1 * g.istr) * g.HW + w1 * g.istr) * g.RS + 8 * (li & 3);
            const char* y0 = dyt + (size_t)m0 * p.DYS + 8 * (li & 3);
            const char* y1 = dyt + (size_t)m1 * p.DYS + 8 * (li & 3);
#pragma unroll
            for (int j = 0; j < MAXI; ++j) {
                const int to = tapoff[it_tap[j]] + it_ci[j] * 32;
                const bf16x8 a = tr_frag(halo + r0 + to, halo + r1 + to);          // A[ci][k=voxel]
                const bf16x8 b = tr_frag(y0 + it_co[j] * 32, y1 + it_co[j] * 32);  // B[k=voxel][co]
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    // ---- add the slab: lane holds dW rows ci = 4*lg + r, column co = li ----
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        if (wave + 4 * j >= nitems) continue;
        const int co = cob * p.COB + it_co[j] * 16 + li;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = cib * p.CIB + it_ci[j] * 16 + 4 * lg + r;
            if (ci < g.Cin)
                atomicAdd(&p.dw[((size_t)tapsrc[it_tap[j]] * g.Cin + ci) * p.Cout + co], acc[j][r]);
        }
    }
    if (do_db && tid < (256 / p.COB) * p.COB) {
        const int co = cob * p.COB + tid % p.COB;
        if (co < p.Cout) atomicAdd(&p.db[co], dbsum);
    }
}

extern "C" int vg_conv3d_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host,
                               float* dw, float* db, vg_stream_t stream) {
    if (!d || !dy || !dw || !tap_idx_host) return VG_EINVAL;
    if (d->Cout < 1 || (d->Cout != 1 && (d->Cout % 8))) return VG_EINVAL;
    if (dy_f32 && d->Cout != 1 && !d->f32) return VG_EINVAL;
    const int Cin = d->c_src0 + d->c_src1;
    const int Cinp = ((Cin + 15) / 16) * 16, Coutp = ((d->Cout + 15) / 16) * 16;
    int CIB = 16;
    for (int c = 64; c >= 16; c -= 16) if (Cinp % c == 0) { CIB = c; break; }
    int COB = 16;
    for (int c = 64; c >= 16; c -= 16) if (Coutp % c == 0) { COB = c; break; }
    constexpr int MAXI = 16;
    GatherIn g; WgradK k;
    int BM = 128, lds = 0, rc;
    for (;;) {
        rc = fill_gather(d, g, CIB, BM);
        if (rc != VG_OK) return rc;
        k.DYS = COB * (d->f32 ? 4 : 2) + 16;
        lds = halo_bytes(g) + BM * k.DYS + 512 + 2 * CIB * 4;
        if (lds <= VG_LDS_LIMIT) break;
        if (BM > 64) BM = 64;
        else if (CIB > 16) CIB = (CIB == 48) ? 16 : CIB / 2;
        else return VG_ELDS;
    }
    k.dy = dy; k.dy_f32 = dy_f32; k.Cout = d->Cout; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW;
    k.CIB = CIB; k.COB = COB; k.ncib = Cinp / CIB; k.ncob = Coutp / COB;
    const int per_tap = (CIB / 16) * (COB / 16);
    k.tpg = (MAXI * 4) / per_tap; if (k.tpg < 1) return VG_EINVAL;
    if (k.tpg > d->ntaps) k.tpg = d->ntaps;
    k.ntg = (d->ntaps + k.tpg - 1) / k.tpg;
    for (int i = 0; i < VG_MAX_TAPS; ++i) k.tap_src[i] = i < d->ntaps ? tap_idx_host[i] : 0;
    k.dw = dw; k.db = db;
    k.total_tiles = d->N * g.tiles_d * g.tiles_h * g.tiles_w;
    const int by = k.ntg * k.ncib * k.ncob;
    int bx = 2048 / by; if (bx < 1) bx = 1; if (bx > k.total_tiles) bx = k.total_tiles;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<bf16_t, MAXI>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<float, MAXI>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    if (d->f32) hipLaunchKernelGGL((wgrad_kernel<float, MAXI>), dim3(bx, by, 1), dim3(256), lds, (hipStream_t)stream, g, k);
    else hipLaunchKernelGGL((wgrad_kernel<bf16_t, MAXI>), dim3(bx, by, 1), dim3(256), lds, (hipStream_t)stream, g, k);
    return vg_check_launch();
}
