// vg_pointwise.hip -- the 1x1x1 convolutions with ONE channel on one side (gfx950), behind vg_conv3d / vg_conv3d_wgrad.
//
// resunet_model.py:240-249 output head  Conv3D(1, 1, activation='tanh')  (16 -> 1), its data gradient (1 -> 16) and weight
// gradient, and the 1 -> 16 shortcut convolution of the stem (resunet_model.py:103-143 with a 1-channel input) have 16..32
// FLOP per voxel: they are HBM-bound byte work.  On the MFMA path they used 1 of 16 rows or columns of every tile and ran at
// ~3x their memory time (86 us for 75 MB at 128^3).  Here each is one pass: every byte read once, written once, 16-byte
// accesses, reductions through LDS and one atomic per (block, value).
//
// Same arithmetic contract as the MFMA kernels: operands are the bf16 (or, in exact-parity mode, fp32) stored values and the
// packed weights, on-read transform y = act(x * scale + shift) rounded to the storage type before the product (as the LDS
// halo image is), dY rounded to the storage type, fp32 accumulation.
#include "vg_common.h"
#include "vg_c1k3.h"
#include <stdlib.h>

namespace {

struct PW {
    const void* x; int x_f32;              // source: [N][S][C] of T, or single channel [N][S] of float / bf16
    const float* scale; const float* shift; int act;
    const void* w; int kc_pad, CK;         // packed weights (row-major [rows][Ktot])
    const float* bias;
    void* out; int out_f32, accumulate, tanh_out;
    float* sums;                           // [VG_STRIPES][N][C][2] or NULL
    const void* dy; int dy_f32;            // weight gradient
    float* dw; float* db;
    int N, C; int64_t S;
    VgFin fin;                             // InstanceNorm finalisation of the output by the last workgroup (pw_1toc)
};

__device__ __forceinline__ float pw_slope(int act) { return act == VG_ACT_RELU ? 0.f : (act == VG_ACT_LRELU ? VG_LRELU : 1.f); }
__device__ __forceinline__ float ld_single(const void* p, int f32, int64_t i) {
    return f32 ? ld_global((const float*)p + i) : ld_global((const bf16_t*)p + i);
}

// ---- C -> 1 forward: one thread per voxel, UB voxels in flight -------------------------------------------------------
template <typename T, int G>       // G = C / 8 channel groups
__global__ __launch_bounds__(256) void pw_cto1_kernel(const PW p) {
    const int n = blockIdx.y;
    float wv[G * 8], sc[G * 8], sf[G * 8];
#pragma unroll
    for (int c = 0; c < G * 8; ++c) {
        wv[c] = ld1<T>((const T*)p.w + (c / p.CK) * p.kc_pad + (c % p.CK));
        sc[c] = p.scale ? p.scale[n * p.C + c] : 1.f;
        sf[c] = p.scale ? p.shift[n * p.C + c] : 0.f;
    }
    const float slope = pw_slope(p.act), b = p.bias ? p.bias[0] : 0.f;
    const T* xb = (const T*)p.x + (size_t)n * p.S * p.C;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.S; v += 2 * stride) {
        Raw8<T> r[2][G];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t vv = v + k * stride < p.S ? v + k * stride : v;
#pragma unroll
            for (int g = 0; g < G; ++g) raw_load(r[k][g], xb + vv * p.C + g * 8);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (v + k * stride >= p.S) continue;
            float a = b;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float x[8]; raw_unpack(r[k][g], x);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = x[j] * sc[g * 8 + j] + sf[g * 8 + j];
                    y = fmaxf(y, y * slope);
                    a += rnd<T>(y) * wv[g * 8 + j];
                }
            }
            const size_t o = (size_t)n * p.S + v + k * stride;
            if (p.tanh_out && !p.accumulate) a = tanhf(a);
            if (p.out_f32) { float* q = (float*)p.out + o; a = p.accumulate ? *q + a : a; *q = a; }
            else { bf16_t* q = (bf16_t*)p.out + o; const bf16_t h = f2bf(p.accumulate ? bf2f(*q) + a : a); *q = h; a = bf2f(h); }
            s1 += a; s2 += a * a;
        }
    }
    if (p.sums) {                         // InstanceNorm statistics of the stored values
        __shared__ float red[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = s1; red[(threadIdx.x >> 6) * 2 + 1] = s2; }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            atomicAdd(&p.sums[((size_t)stripe * gridDim.y + n) * 2 + threadIdx.x],
                      red[threadIdx.x] + red[2 + threadIdx.x] + red[4 + threadIdx.x] + red[6 + threadIdx.x]);
        }
    }
}

// ---- 1 -> C forward / data gradient: thread = (voxel, 8-channel group) ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pw_1toc_kernel(const PW p) {
    __shared__ float part[16 * 256];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int gpc = p.C >> 3, vpb = 256 / gpc;
    const int cg = tid % gpc, vl = tid / gpc;
    const bool live = tid < gpc * vpb;
    float wv[8], bv[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wv[j] = live ? ld1<T>((const T*)p.w + (size_t)(cg * 8 + j) * p.kc_pad) : 0.f;
        bv[j] = (live && p.bias) ? p.bias[cg * 8 + j] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }
    const float sc = p.scale ? p.scale[n] : 1.f, sf = p.scale ? p.shift[n] : 0.f, slope = pw_slope(p.act);
    const int64_t xoff = (int64_t)n * p.S;
    T* ob = (T*)p.out + (size_t)n * p.S * p.C + cg * 8;
    const int64_t stride = (int64_t)gridDim.x * vpb;
    if (live)
    for (int64_t v = (int64_t)blockIdx.x * vpb + vl; v < p.S; v += 4 * stride) {
        float xs[4]; Raw8<T> old[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t vv = v + k * stride < p.S ? v + k * stride : v;
            xs[k] = ld_single(p.x, p.x_f32, xoff + vv);
            if (p.accumulate) raw_load(old[k], ob + vv * p.C);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (v + k * stride >= p.S) continue;
            float y = xs[k] * sc + sf;
            y = rnd<T>(fmaxf(y, y * slope));
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = y * wv[j] + bv[j];
            if (p.accumulate) {
                float q[8]; raw_unpack(old[k], q);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += q[j];
            }
            store8<T>(ob + (v + k * stride) * p.C, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float r = rnd<T>(o[j]); s1[j] += r; s2[j] += r * r; }
        }
    }
    if (!p.sums) return;
    // block reduction (value-major LDS, fixed order), one atomic per (channel, moment) and block
#pragma unroll
    for (int j = 0; j < 8; ++j) { part[(2 * j) * 256 + tid] = live ? s1[j] : 0.f; part[(2 * j + 1) * 256 + tid] = live ? s2[j] : 0.f; }
    __syncthreads();
    const int stripe = blockIdx.x & (VG_STRIPES - 1);
    float* dst = p.sums + ((size_t)stripe * gridDim.y + n) * p.C * 2;
    for (int o = tid; o < p.C * 2; o += 256) {
        const int ch = o >> 1, mom = o & 1, g8 = ch >> 3, j = ch & 7;
        const float* src = part + (2 * j + mom) * 256 + g8;
        float a = 0.f;
        for (int t = 0; t < vpb; ++t) a += src[t * gpc];
        atomicAdd(&dst[o], a);
    }
    if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.y, p.C, gridDim.x * gridDim.y, (int*)part);
}

// ---- Cin (8 or 16) -> C, plain source (the data gradient of a 1x1x1 shortcut convolution, decoder level 0: 16 -> 48 at 128^3,
// accumulated into the concat gradient): thread = (voxel, 8-channel output group), its Cin x 8 weights live in registers
template <typename T, int GI>
__global__ __launch_bounds__(256) void pw_ctoc_kernel(const PW p, int Cin, int Ktot) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const int gpc = p.C >> 3, vpb = 256 / gpc;
    const int cg = tid % gpc, vl = tid / gpc;
    if (tid >= gpc * vpb) return;
    float wv[GI * 8][8], bv[8];
#pragma unroll
    for (int k = 0; k < GI * 8; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            wv[k][j] = ld1<T>((const T*)p.w + (size_t)(cg * 8 + j) * Ktot + (k / p.CK) * p.kc_pad + (k % p.CK));
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = p.bias ? p.bias[cg * 8 + j] : 0.f;
    const T* xb = (const T*)p.x + (size_t)n * p.S * Cin;
    T* ob = (T*)p.out + (size_t)n * p.S * p.C + cg * 8;
    const int64_t stride = (int64_t)gridDim.x * vpb;
    for (int64_t v = (int64_t)blockIdx.x * vpb + vl; v < p.S; v += 2 * stride) {
        Raw8<T> xr[2][GI], old[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t vv = v + k * stride < p.S ? v + k * stride : v;
#pragma unroll
            for (int g = 0; g < GI; ++g) raw_load(xr[k][g], xb + vv * Cin + g * 8);
            if (p.accumulate) raw_load(old[k], ob + vv * p.C);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (v + k * stride >= p.S) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = bv[j];
#pragma unroll
            for (int g = 0; g < GI; ++g) {
                float x[8]; raw_unpack(xr[k][g], x);
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += x[e] * wv[g * 8 + e][j];
            }
            if (p.accumulate) {
                float q[8]; raw_unpack(old[k], q);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += q[j];
            }
            store8<T>(ob + (v + k * stride) * p.C, o);
        }
    }
}

// ---- Cin -> Cout, both multi-channel, plain source (the 1x1x1 shortcut convolutions of the residual blocks,
// resunet_model.py:103-143: decoder level 0 reads the virtual upsample + concat 48 -> 16 at 128^3, the encoders 16 -> 32 with
// stride 2, ...; and their data gradients Cout -> Cin accumulated into the block-input gradient).  A few dozen FLOP per byte:
// HBM-bound.  The generic gather kernels stage a halo image through LDS for ONE tap and ran these at 1.1-1.6 TB/s.
// Here there is no LDS on the data path at all: with the voxel as the N index of v_mfma_f32_16x16x32_bf16, lane (kg, r) supplies
// B[k = 8kg..8kg+7][n = r] = 8 consecutive channels of voxel r -- exactly one 16-byte global load from the [voxel][channel]
// layout -- the weights A[m = co][k] sit in registers for the whole kernel, and the result D[m = 4kg..4kg+3][n = r] is 4
// consecutive output channels of voxel r; two sub-tiles are exchanged with v_permlane16_swap so that every lane stores 16
// contiguous bytes.  A wave handles MS sub-tiles of 16 consecutive output voxels per iteration (all their loads in flight
// together); InstanceNorm statistics of the stored values are carried in registers.
struct PWG {
    const bf16_t* x0; const bf16_t* x1; int c0, c1, Cin, sh;
    const bf16_t* w; int Ktot, CK, kc_pad;
    const float* bias; bf16_t* out; float* sums;
    int N, ID, IH, IW, OD, OH, OW, istr;
    int BD, BH, BW, ostr, od0, oh0, ow0;
    int Cout; int SO;
    unsigned long long* stamps;          // diagnostic (vg_set_stamp_buffer): s_memrealtime at wave start / end, else NULL
    // SPLIT (data gradient of a decoder shortcut, fused with the backward of UpSampling3D + concatenate): `out` is only READ (the
    // gradient of the virtual concat left there by the conv branch); the sum goes to dskip (channels >= sc0, per voxel) and, summed
    // over every 2x2x2 block, to dlow (channels < sc0, half resolution).  acc bit 0: dlow accumulates, bit 1: dskip accumulates.
    bf16_t* dlow; bf16_t* dskip; int sc0, sacc;
    // SPLIT == 2: the conv branch's part of the concat gradient is not stored at all -- it is the (InstanceNorm -> act) backward of
    // the block's first convolution, computed here from that convolution's padded data gradient `ng` (reflection pad folded on
    // read), the forward operands nx0 (low resolution, channels < sc0) / nx1 and the per-(sample, channel) constants
    const bf16_t* ng; const bf16_t* nx0; const bf16_t* nx1;
    const float* n_scale; const float* n_shift; const float* n_mean; const float* n_rstd; const float* n_gamma; const float* n_mult;
    VgFin fin;                           // STATS launches: InstanceNorm finalisation of the output by the last workgroup
    const float* n_red; float* n_dgamma; float* n_dbeta; int n_act;
};
}
extern unsigned long long* g_vg_stamps;
namespace {
typedef __attribute__((ext_vector_type(2))) float pw_f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned pw_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned pw_u32x4;

// g[0..7] += the mirrored copies of voxel (dk, hk, wk) on a reflection-padded grid [D+2][H+2][W+2][C] (gp: sample base + channel):
// every combination of {own, mirrored} per axis except the all-own one (the caller loaded it); index 1 mirrors to padded 0, index
// n - 2 to padded n + 1 (extents >= 4: an index has at most one mirror per axis).  The seven loads go out together on clamped
// addresses (a serial loop made the 30 % of waves that hold a border voxel wait for up to seven round trips).
__device__ __forceinline__ void vg_reflect_fold8(const bf16_t* gp, int C, int D, int H, int W, int dk, int hk, int wk, float* g) {
    const int PH = H + 2, PW_ = W + 2;
    const int md = dk == 1 ? 0 : (dk == D - 2 ? D + 1 : -1), mh = hk == 1 ? 0 : (hk == H - 2 ? H + 1 : -1), mw = wk == 1 ? 0 : (wk == W - 2 ? W + 1 : -1);
    Raw8<bf16_t> rr[7];
    bool val[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        const int qd = (t & 4) ? md : dk + 1, qh = (t & 2) ? mh : hk + 1, qw = (t & 1) ? mw : wk + 1;
        val[t - 1] = (qd | qh | qw) >= 0;
        raw_load(rr[t - 1], gp + (val[t - 1] ? ((size_t)(qd * PH + qh) * PW_ + qw) * C : ((size_t)((dk + 1) * PH + hk + 1) * PW_ + wk + 1) * C));
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        float r[8]; raw_unpack(rr[t], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += val[t] ? r[j] : 0.f;
    }
}

template <int KS, int NB, bool GEO, bool ACC, bool STATS, int SPLIT = 0>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const PWG p) {
    static_assert(!SPLIT || (ACC && !GEO && !STATS), "SPLIT: accumulating, same-grid launches");
    constexpr int MS = (KS * NB <= 2) ? 4 : 2;
    __shared__ float stat[NB * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    if (p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 8] = __builtin_amdgcn_s_memrealtime();
    // SPLIT == 2: dx = A * m * g - B - Cc * x per channel of sample n, m = 1 where x * scale + shift > 0 else the activation's slope
    // (the IN backward k0 * dn - k1 - k2 * xhat of vg_elem.hip with dn = g * mult * m and xhat = (x - mean) * rstd, constants
    // folded); the 8 statistics stripes are added up here, and workgroup 0 of the sample adds the gamma / beta gradients
    float* ntab = nullptr;
    if constexpr (SPLIT == 2) {
        __shared__ float ntab_s[5 * NB * 16];
        ntab = ntab_s;
        constexpr int C = NB * 16;
        if (tid < C && tid < p.Cout) {
            const int nc = n * p.Cout + tid;
            const size_t total = (size_t)p.N * p.Cout * 2;
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int t = 0; t < VG_STRIPES; ++t) { r0 += p.n_red[t * total + (size_t)nc * 2]; r1 += p.n_red[t * total + (size_t)nc * 2 + 1]; }
            if (p.n_dgamma && blockIdx.x == 0) { atomicAdd(&p.n_dbeta[tid], r0); atomicAdd(&p.n_dgamma[tid], r1); }
            const float cnt = (float)p.SO, rs = p.n_rstd[nc], mu = p.n_mean[nc], gr = p.n_gamma[tid] * rs;
            const float k1 = gr * r0 / cnt, k2 = gr * r1 / cnt, cc = k2 * rs;
            ntab_s[tid] = p.n_scale ? p.n_scale[nc] : 1.f;
            ntab_s[C + tid] = p.n_scale ? p.n_shift[nc] : 0.f;
            ntab_s[2 * C + tid] = gr * (p.n_mult ? p.n_mult[nc] : 1.f);
            ntab_s[3 * C + tid] = k1 - cc * mu;
            ntab_s[4 * C + tid] = cc;
        }
        __syncthreads();
    }
    bf16x8 wA[NB][KS];
    pw_f32x2 e_b[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 32 + kg * 8;
            wA[nb][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (co < p.Cout && c < p.Cin) wA[nb][ks] = *(const bf16x8*)(p.w + (size_t)co * p.Ktot + (c / p.CK) * p.kc_pad + (c % p.CK));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cb = nb * 16 + kg * 4 + i;
            e_b[nb][i >> 1][i & 1] = (p.bias && cb < p.Cout) ? p.bias[cb] : 0.f;
        }
    }
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 1] = __builtin_amdgcn_s_memrealtime(); }
    float s1[NB][4], s2[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1[nb][i] = 0.f; s2[nb][i] = 0.f; }
    if (STATS) { if (tid < NB * 32) stat[tid] = 0.f; __syncthreads(); }
    const int OHW = p.OH * p.OW;
    const int ID2 = p.ID >> p.sh, IH2 = p.IH >> p.sh, IW2 = p.IW >> p.sh;
    const int nwt = (p.SO + MS * 16 - 1) / (MS * 16);
    const int jodd = kg & 1, chof = (kg & ~1) * 4;
    // SPLIT: the lane holds channels nb*16 + chof .. + 7 of one voxel of sub-tile jp + jodd.  Low-resolution source: sum over the
    // 2x2x2 block = lanes r, r^1, r^2, r^4 of the 16-lane row, one lane of eight stores; skip source: plain store per voxel.
    auto split_store = [&](int wt, int jp, int nb, float* o, bool ok, size_t vox) {
        if (nb * 16 < p.sc0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { o[i] += __shfl_xor(o[i], 1); o[i] += __shfl_xor(o[i], 2); o[i] += __shfl_xor(o[i], 4); }
            if (ok && (r & 7) == 0) {
                const size_t cidx = (size_t)n * (p.SO >> 3) + (size_t)wt * (MS * 2) + (jp + jodd) * 2 + (r >> 3);
                bf16_t* dst = p.dlow + cidx * p.sc0 + nb * 16 + chof;
                if (p.sacc & 1) { float old[8]; load8<bf16_t>(dst, old); for (int i = 0; i < 8; ++i) o[i] += old[i]; }
                store8<bf16_t>(dst, o);
            }
        } else if (ok) {
            bf16_t* dst = p.dskip + vox * (p.Cout - p.sc0) + (nb * 16 - p.sc0) + chof;
            if (p.sacc & 2) { float old[8]; load8<bf16_t>(dst, old); for (int i = 0; i < 8; ++i) o[i] += old[i]; }
            store8<bf16_t>(dst, o);
        }
    };
    for (int wt = blockIdx.x * 4 + wave; wt < nwt; wt += gridDim.x * 4) {
        bf16x8 xb[MS][KS];
        bool vok[MS], oks[MS / 2];
        size_t i0s[MS], i1s[MS], ovs[MS / 2];           // ovs / oks: the sub-tile this lane STORES of each pair (jp + jodd)
        size_t ovf[MS / 2];                             // SPLIT: its voxel index (sample included)
        int fdhw[MS / 2];                               // SPLIT == 2: its coordinates, packed
#pragma unroll
        for (int jp = 0; jp < MS; jp += 2) {
            size_t ovp[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = jp + e;
                const int v = wt * (MS * 16) + i * 16 + r;
                vok[i] = v < p.SO;
                const int vc = vok[i] ? v : p.SO - 1;
                if (GEO) {
                    const int od = vc / OHW, rem = vc - od * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
                    const int id = od * p.istr, ih = oh * p.istr, iw = ow * p.istr;
                    i0s[i] = (((size_t)n * ID2 + (id >> p.sh)) * IH2 + (ih >> p.sh)) * IW2 + (iw >> p.sh);
                    i1s[i] = (((size_t)n * p.ID + id) * p.IH + ih) * p.IW + iw;
                    ovp[e] = (((size_t)n * p.BD + od * p.ostr + p.od0) * p.BH + oh * p.ostr + p.oh0) * p.BW + ow * p.ostr + p.ow0;
                } else if (SPLIT) {
                    // voxels enumerated block-major: 8 consecutive indices are one 2x2x2 block, so that the 16 voxels of a sub-tile
                    // are two whole blocks and the pooling is a reduction over 8 neighbouring lanes
                    const int cb = vc >> 3, j = vc & 7;
                    const int CW = p.OW >> 1, CHW = (p.OH >> 1) * CW;
                    const int cd = cb / CHW, rem = cb - cd * CHW, ch = rem / CW, cw = rem - ch * CW;
                    const int fd = 2 * cd + (j >> 2), fh = 2 * ch + ((j >> 1) & 1), fw = 2 * cw + (j & 1);
                    const int vf = (fd * p.OH + fh) * p.OW + fw;
                    i0s[i] = i1s[i] = ovp[e] = (size_t)n * p.SO + vf;
                    if (e == jodd) fdhw[jp >> 1] = fd | (fh << 10) | (fw << 20);
                } else {
                    i0s[i] = i1s[i] = ovp[e] = (size_t)n * p.SO + vc;
                }
            }
            ovf[jp >> 1] = jodd ? ovp[1] : ovp[0];
            ovs[jp >> 1] = (jodd ? ovp[1] : ovp[0]) * p.Cout + chof;
            oks[jp >> 1] = jodd ? vok[jp + 1] : vok[jp];
        }
        // accumulate mode: the old values are fetched together with the operands (out-of-range voxels were clamped to the last
        // one: a valid address)
        Raw8<bf16_t> oldv[ACC ? MS / 2 : 1][ACC ? NB : 1];
        if (ACC && SPLIT != 2) {
#pragma unroll
            for (int jh = 0; jh < MS / 2; ++jh)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) raw_load(oldv[jh][nb], p.out + ovs[jh] + (nb * 16 < p.Cout ? nb * 16 : 0));
            if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 3] = __builtin_amdgcn_s_memrealtime(); }
        }
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // lanes beyond Cin (K padding of the last step) re-read a valid channel group of their own voxel and are zeroed: a
                // shared dummy address would send half of every wave of the launch to ONE cache line
                const int c = ks * 32 + kg * 8;
                const bool cok = c < p.Cin;
                const int ce = cok ? c : c % p.Cin;
                const bf16_t* q = ce < p.c0 ? p.x0 + i0s[i] * p.c0 + ce : p.x1 + i1s[i] * p.c1 + (ce - p.c0);
                Raw8<bf16_t> t;
                raw_load(t, q);
                raw_mask(t, cok);
                xb[i][ks] = t.v;
            }
        if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 2] = __builtin_amdgcn_s_memrealtime(); }
        if constexpr (SPLIT == 2) {
            // One channel block at a time -- loads of the padded data gradient (interior position) and the forward operand, the
            // block's MFMAs, the IN backward, split / pool / store -- so that the live state is one block's, not all NB blocks'
            // (all at once: 216-305 VGPRs, one or two waves per SIMD on an HBM-bound kernel: 171 us where 72 + 115 were to be beaten).
            const int PH = p.OH + 2, PW_ = p.OW + 2;
            size_t gi[MS / 2], li[MS / 2], si[MS / 2];
            bool bord[MS / 2];
#pragma unroll
            for (int jh = 0; jh < MS / 2; ++jh) {
                const int fd = fdhw[jh] & 1023, fh = (fdhw[jh] >> 10) & 1023, fw = fdhw[jh] >> 20;
                gi[jh] = ((((size_t)n * (p.OD + 2) + fd + 1) * PH + fh + 1) * PW_ + fw + 1) * p.Cout + chof;
                li[jh] = ((((size_t)n * (p.OD >> 1) + (fd >> 1)) * (p.OH >> 1) + (fh >> 1)) * (p.OW >> 1) + (fw >> 1)) * p.sc0 + chof;
                si[jh] = ovf[jh] * (p.Cout - p.sc0) + chof;
                bord[jh] = fd == 1 || fd == p.OD - 2 || fh == 1 || fh == p.OH - 2 || fw == 1 || fw == p.OW - 2;
            }
            constexpr int C = NB * 16;
            const float slope = pw_slope(p.n_act);
            // every load of the wave-tile goes out first (7 KB in flight per wave: the kernel is HBM-bound), the blocks are then
            // finished one by one
            Raw8<bf16_t> gq[NB][MS / 2], xq[NB][MS / 2];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int jh = 0; jh < MS / 2; ++jh) {
                    raw_load(gq[nb][jh], p.ng + gi[jh] + nb * 16);
                    raw_load(xq[nb][jh], nb * 16 < p.sc0 ? p.nx0 + li[jh] + nb * 16 : p.nx1 + si[jh] + (nb * 16 - p.sc0));
                }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 a[MS];
#pragma unroll
                for (int i = 0; i < MS; ++i) {
                    a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) a[i] = VG_MFMA16(wA[nb][ks], xb[i][ks], a[i]);
                }
                const float* tb = ntab + nb * 16 + chof;
#pragma unroll
                for (int jp = 0; jp < MS; jp += 2) {
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const pw_u32x2 x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[jp][i]), __float_as_uint(a[jp + 1][i]), false, false);
                        o[i] = __uint_as_float(x[0]); o[4 + i] = __uint_as_float(x[1]);
                    }
                    float q[8], xv[8];
                    raw_unpack(gq[nb][jp >> 1], q); raw_unpack(xq[nb][jp >> 1], xv);
                    // transpose of the reflection pad: voxels next to a face also own the mirrored positions of the padded grid
                    if (bord[jp >> 1]) {
                        const int fd = fdhw[jp >> 1] & 1023, fh = (fdhw[jp >> 1] >> 10) & 1023, fw = fdhw[jp >> 1] >> 20;
                        vg_reflect_fold8(p.ng + (size_t)n * (p.OD + 2) * PH * PW_ * p.Cout + nb * 16 + chof, p.Cout, p.OD, p.OH, p.OW, fd, fh, fw, q);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float pre = xv[i] * tb[i] + tb[C + i];
                        const float m = (pre > 0.f || p.n_act == VG_ACT_NONE) ? 1.f : slope;
                        o[i] += tb[2 * C + i] * m * q[i] - tb[3 * C + i] - tb[4 * C + i] * xv[i];
                    }
                    split_store(wt, jp, nb, o, oks[jp >> 1], ovf[jp >> 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        f32x4 acc[MS][NB];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[i][nb] = VG_MFMA16(wA[nb][ks], xb[i][ks], acc[i][nb]);
            }
        // ---- epilogue: sub-tiles (jp, jp+1) exchanged across the 16-lane rows: even rows end with sub-tile jp channels
        // [4kg..4kg+7], odd rows with sub-tile jp+1 channels [4(kg-1)..4kg+3]
#pragma unroll
        for (int jp = 0; jp < MS; jp += 2) {
            const bool ok = oks[jp >> 1];
            const size_t ob = ovs[jp >> 1];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (nb * 16 >= p.Cout) continue;
                bf16_t* optr = p.out + ob + nb * 16;
                pw_u32x4 outv;
                if (ACC) {
                    // single rounding of (old + new): the exchange is done on the f32 values
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const pw_u32x2 x = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[jp][nb][i] + e_b[nb][i >> 1][i & 1]),
                                                                            __float_as_uint(acc[jp + 1][nb][i] + e_b[nb][i >> 1][i & 1]), false, false);
                        o[i] = __uint_as_float(x[0]); o[4 + i] = __uint_as_float(x[1]);
                    }
                    // the bias of the received half belongs to the neighbouring row's channels: e_b was added before the swap
                    float q[8]; raw_unpack(oldv[ACC ? jp >> 1 : 0][ACC ? nb : 0], q);
                    if (SPLIT) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] += q[i];
                        split_store(wt, jp, nb, o, ok, ovf[jp >> 1]);
                        continue;
                    }
                    bf16x8 pk;
#pragma unroll
                    for (int i = 0; i < 8; ++i) pk[i] = (short)f2bf(o[i] + q[i]);
                    outv = __builtin_bit_cast(pw_u32x4, pk);
                } else {
                    bf16x4 pk[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        pw_f32x2 v0 = {acc[jp + e][nb][0], acc[jp + e][nb][1]}, v1 = {acc[jp + e][nb][2], acc[jp + e][nb][3]};
                        v0 += e_b[nb][0]; v1 += e_b[nb][1];
                        pk[e] = (bf16x4){(short)f2bf(v0[0]), (short)f2bf(v0[1]), (short)f2bf(v1[0]), (short)f2bf(v1[1])};
                        if (STATS) {
                            const bool okv = vok[jp + e];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float qv = okv ? bf2f((bf16_t)pk[e][i]) : 0.f;
                                s1[nb][i] += qv; s2[nb][i] += qv * qv;
                            }
                        }
                    }
                    const pw_u32x2 wa = __builtin_bit_cast(pw_u32x2, pk[0]), wb = __builtin_bit_cast(pw_u32x2, pk[1]);
                    const pw_u32x2 x0 = __builtin_amdgcn_permlane16_swap(wa[0], wb[0], false, false);
                    const pw_u32x2 x1 = __builtin_amdgcn_permlane16_swap(wa[1], wb[1], false, false);
                    outv = (pw_u32x4){x0[0], x1[0], x0[1], x1[1]};
                }
                if (ok) *(pw_u32x4*)optr = outv;
            }
        }
    }
    if (p.stamps && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); }
    if (STATS && p.sums) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = s1[nb][i], b = s2[nb][i];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (r == 0) { atomicAdd(&stat[(nb * 16 + 4 * kg + i) * 2], a); atomicAdd(&stat[(nb * 16 + 4 * kg + i) * 2 + 1], b); }
            }
        __syncthreads();
        if (tid < NB * 32) {
            const int co = tid >> 1;
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            if (co < p.Cout) atomicAdd(&p.sums[(((size_t)stripe * gridDim.y + n) * p.Cout + co) * 2 + (tid & 1)], stat[tid]);
        }
        if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.y, p.Cout, gridDim.x * gridDim.y, (int*)stat);
    }
}

// ---- weight gradient of the multi-channel 1x1x1 shortcut convolutions: dW[ci][co] = sum_v x[v][ci] * dY[v][co], db = sum dY.
// 2 * Cin * Cout FLOP against 2 * (Cin + Cout) bytes per voxel: HBM-bound, and with K = voxels the MFMA operands would need
// transposed (2-byte strided) global reads.  VALU instead: thread role = (8 input channels, 16 output channels) keeps its
// 8 x 16 block of dW in registers (packed f32 FMAs), reads 16 bytes of x and 32 bytes of dY per voxel, walks the voxels with an
// incrementally updated (d, h, w) (virtual upsample / concat / stride of the source), and the block's roles are reduced
// through LDS into one partial slab per workgroup (summed in a fixed order by reduce_partials_kernel).
struct PWW {
    const bf16_t* x0; const bf16_t* x1; int c0, c1, Cin, sh;
    const bf16_t* dy; int Cout;
    int N, ID, IH, IW, OD, OH, OW, istr, SO;
    int gin, roles, rp, vpb;             // rp = roles rounded up to a power of two (threads role >= roles idle), vpb = 256 / rp
    float* dw; float* db; float* part; int dw_elems;
};

__global__ __launch_bounds__(256) void pw_wgrad_cc_kernel(const PWW p) {
    __shared__ float red[4 * 64 * 16];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int role = tid & (p.rp - 1), vl = tid / p.rp;
    const bool live = role < p.roles;
    const int rl = live ? role : 0;
    const int cg = rl % p.gin, cob = rl / p.gin, c = cg * 8;
    const bool from0 = c < p.c0;
    const float bmask = cg == 0 ? 1.f : 0.f;
    pw_f32x2 acc[8][8], bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bs[i] = (pw_f32x2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (pw_f32x2){0.f, 0.f};
    }
    const int ID2 = p.ID >> p.sh, IH2 = p.IH >> p.sh, IW2 = p.IW >> p.sh;
    const int step = gridDim.x * p.vpb;
    // (od, oh, ow) of this thread's first voxel and of the grid stride: advanced with carries, no per-voxel divisions
    int v = blockIdx.x * p.vpb + vl;
    const int OHW = p.OH * p.OW;
    int od = v / OHW, oh = (v - od * OHW) / p.OW, ow = v - od * OHW - oh * p.OW;
    const int sd = step / OHW, sh_ = (step - sd * OHW) / p.OW, sw = step - sd * OHW - sh_ * p.OW;
    const bf16_t* xb = from0 ? p.x0 + c : p.x1 + (c - p.c0);
    const int cs = from0 ? p.c0 : p.c1, shx = from0 ? p.sh : 0;
    const int XD = from0 ? ID2 : p.ID, XH = from0 ? IH2 : p.IH, XW = from0 ? IW2 : p.IW;
    const bf16_t* yb = p.dy + (size_t)n * p.SO * p.Cout + cob * 16;
    if (live)
    for (; v < p.SO; v += 2 * step) {
        Raw8<bf16_t> xr[2], yr[2][2]; bool okv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int vk = v + k * step;
            okv[k] = vk < p.SO;
            const int dq = okv[k] ? od : 0, hq = okv[k] ? oh : 0, wq = okv[k] ? ow : 0, vq = okv[k] ? vk : 0;
            const size_t xi = (((size_t)n * XD + ((dq * p.istr) >> shx)) * XH + ((hq * p.istr) >> shx)) * XW + ((wq * p.istr) >> shx);
            raw_load(xr[k], xb + xi * cs);
            raw_load(yr[k][0], yb + (size_t)vq * p.Cout);
            raw_load(yr[k][1], yb + (size_t)vq * p.Cout + 8);
            ow += sw; if (ow >= p.OW) { ow -= p.OW; ++oh; }
            oh += sh_; if (oh >= p.OH) { oh -= p.OH; ++od; }
            od += sd;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float x[8], y[16];
            raw_unpack(xr[k], x); raw_unpack(yr[k][0], y); raw_unpack(yr[k][1], y + 8);
            pw_f32x2 y2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y2[j] = okv[k] ? (pw_f32x2){y[2 * j], y[2 * j + 1]} : (pw_f32x2){0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const pw_f32x2 xi2 = {x[i], x[i]};
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] += xi2 * y2[j];
            }
            const pw_f32x2 bm = {bmask, bmask};
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[j] += y2[j] * bm;
        }
    }
    // ---- reduction.  Roles are dealt out modulo rp (a power of two): the lanes of a wave that share a role are rp apart and
    // are summed with xor shuffles; one lane per (wave, role) then holds 144 values (8 rows of 16 + 16 bias sums), which go
    // through LDS in 9 chunks of 16 and are added over the 4 waves into the workgroup's partial slab ----
    for (int off = 32; off >= p.rp; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bs[j][0] += __shfl_xor(bs[j][0], off); bs[j][1] += __shfl_xor(bs[j][1], off);
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc[i][j][0] += __shfl_xor(acc[i][j][0], off); acc[i][j][1] += __shfl_xor(acc[i][j][1], off); }
        }
    }
    float* slab = p.part ? p.part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * p.dw_elems : nullptr;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int ch = 0; ch < 9; ++ch) {
        __syncthreads();
        if (lane < p.rp) {
            float* dst = red + (size_t)(wave * p.rp + lane) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const pw_f32x2 val = ch < 8 ? acc[ch < 8 ? ch : 0][j] : bs[j];
                dst[2 * j] = val[0]; dst[2 * j + 1] = val[1];
            }
        }
        __syncthreads();
        for (int o = tid; o < p.rp * 16; o += 256) {
            const int ro = o >> 4, i = o & 15;
            if (ro >= p.roles) continue;
            const float sacc = red[o] + red[p.rp * 16 + o] + red[2 * p.rp * 16 + o] + red[3 * p.rp * 16 + o];
            const int rcg = ro % p.gin, rcob = ro / p.gin, co = rcob * 16 + i;
            if (ch < 8) {
                const size_t idx = (size_t)(rcg * 8 + ch) * p.Cout + co;
                if (slab) slab[idx] = sacc; else atomicAdd(&p.dw[idx], sacc);
            } else if (rcg == 0 && p.db) atomicAdd(&p.db[co], sacc);
        }
    }
}

// ---- weight gradients: dW[c] = sum_v P[v,(c)] * dY[v,(c)], db = sum dY; one side single-channel ------------------------
// MULTI_X: true = C -> 1 layer (x has C channels, dY one), false = 1 -> C layer (x one channel, dY C channels)
template <typename T, bool MULTI_X>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const PW p) {
    __shared__ float part[16 * 256];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int gpc = p.C >> 3, vpb = 256 / gpc;
    const int cg = tid % gpc, vl = tid / gpc;
    const bool live = tid < gpc * vpb;
    float a[8], bsum[8], sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = 0.f; bsum[j] = 0.f;
        const int ci = MULTI_X ? n * p.C + cg * 8 + j : n;
        sc[j] = (p.scale && live) ? p.scale[ci] : 1.f; sf[j] = (p.scale && live) ? p.shift[ci] : 0.f;
    }
    const float slope = pw_slope(p.act);
    const T* mb = (const T*)(MULTI_X ? p.x : p.dy) + (size_t)n * p.S * p.C + cg * 8;       // the multi-channel operand
    const void* sb = MULTI_X ? p.dy : p.x;                                                  // the single-channel operand
    const int sb_f32 = MULTI_X ? p.dy_f32 : p.x_f32;
    const int64_t soff = (int64_t)n * p.S;
    const int64_t stride = (int64_t)gridDim.x * vpb;
    if (live)
    for (int64_t v = (int64_t)blockIdx.x * vpb + vl; v < p.S; v += 4 * stride) {
        Raw8<T> m[4]; float sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t vv = v + k * stride < p.S ? v + k * stride : v;
            raw_load(m[k], mb + vv * p.C);
            sv[k] = ld_single(sb, sb_f32, soff + vv);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (v + k * stride >= p.S) continue;
            float q[8]; raw_unpack(m[k], q);
            if (MULTI_X) {
                const float g = rnd<T>(sv[k]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float y = q[j] * sc[j] + sf[j]; y = rnd<T>(fmaxf(y, y * slope)); a[j] += y * g; }
                bsum[0] += g;
            } else {
                float y = sv[k] * sc[0] + sf[0]; y = rnd<T>(fmaxf(y, y * slope));
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] += y * q[j]; bsum[j] += q[j]; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { part[(2 * j) * 256 + tid] = live ? a[j] : 0.f; part[(2 * j + 1) * 256 + tid] = live ? bsum[j] : 0.f; }
    __syncthreads();
    for (int o = tid; o < p.C * 2; o += 256) {
        const int ch = o >> 1, mom = o & 1, g8 = ch >> 3, j = ch & 7;
        const float* src = part + (2 * j + mom) * 256 + g8;
        float s = 0.f;
        for (int t = 0; t < vpb; ++t) s += src[t * gpc];
        if (mom == 0) atomicAdd(&p.dw[ch], s);
        else if (p.db) {
            if (!MULTI_X) atomicAdd(&p.db[ch], s);
            else if (ch == 0) {
                // C -> 1: the bias gradient is the sum of dY; every channel group of the block counted its own voxels' dY in
                // slot j = 0, so summing group 0's column alone covers each voxel once
                atomicAdd(&p.db[0], s);
            }
        }
    }
}

// ---- single-channel source, 3x3x3 taps (the stem convolution 1 -> 16, resunet_model.py:44-60): 27 x 16 MACs per voxel are
// VALU work next to 4 + 32 bytes of traffic.  A thread owns 4 consecutive voxels along W and 8 output channels; the 3x3x6
// source window lives in registers, the weights come from LDS as broadcast reads shared by the 4 voxels.
// (struct C1K3, c1_resolve: vg_c1k3.h -- shared with the MFMA kernels of vg_c1k3.hip)
// loads the rows (a, b) x 6 columns of the source window of quad (d, h, w0): NA d-offsets starting at a0
// S = element type of the single-channel source (a run-time switch here would put every load under a branch, and the compiler
// drains vmcnt at each of them)
template <typename T, typename S, int NA>
__device__ __forceinline__ void c1_window(const C1K3& p, int n, int d, int h, int w0, int a0, float (&xv)[NA][3][6]) {
    const bool refl = p.pad_mode == VG_PAD_REFLECT;
    const float slope = pw_slope(p.act);
    int cw[6]; bool okw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) cw[i] = c1_resolve(w0 + p.tw0 + i, p.W, refl, okw[i]);
    const int64_t base = (int64_t)n * p.D * p.H * p.W;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        bool okd; const int rd = c1_resolve(d + p.td0 + a0 + a, p.D, refl, okd);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            bool okh; const int rh = c1_resolve(h + p.th0 + b, p.H, refl, okh);
            const int64_t row = base + ((int64_t)rd * p.H + rh) * p.W;
#pragma unroll
            for (int i = 0; i < 6; ++i) xv[a][b][i] = ld_global((const S*)p.x + row + cw[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float y = xv[a][b][i] * p.sc + p.sf;
                y = rnd<T>(fmaxf(y, y * slope));
                xv[a][b][i] = (okd && okh && okw[i]) ? y : 0.f;
            }
        }
    }
}

template <typename T, typename S>
__global__ __launch_bounds__(256, 3) void c1k3_fwd_kernel(C1K3 p) {
    __shared__ float wl[27 * 32];                 // [tap(a,b)][j][co], Cout <= 32
    __shared__ float part[16 * 256];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int gpc = p.C >> 3, qpb = 256 / gpc;
    const int cg = tid % gpc, ql = tid / gpc;
    const bool live = tid < gpc * qpb;
    if (p.scale) { p.sc = p.scale[n]; p.sf = p.shift[n]; }
    for (int i = tid; i < 27 * p.C; i += 256) {
        const int co = i % p.C, tj = i / p.C, t = tj / 3, j = tj - t * 3;
        wl[i] = ld1<T>((const T*)p.w + (size_t)co * p.Ktot + t * p.CK + j);
    }
    __syncthreads();
    float bv[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bv[j] = (live && p.bias) ? p.bias[cg * 8 + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    const int64_t nq = (int64_t)p.D * p.H * p.W4;
    if (live)
    for (int64_t q = (int64_t)blockIdx.x * qpb + ql; q < nq; q += (int64_t)gridDim.x * qpb) {
        const int wq = (int)(q % p.W4); const int64_t r = q / p.W4;
        const int h = (int)(r % p.H), d = (int)(r / p.H), w0 = wq * 4;
        float acc[4][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[v][j] = bv[j];
        // one d-offset at a time (not unrolled): the fully unrolled form hoisted all 54 weight reads and 54 window loads and
        // needed 256 VGPRs (one wave per SIMD) or, capped, 1 KB of scratch per lane
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
            float xv[1][3][6];
            c1_window<T, S, 1>(p, n, d, h, w0, a, xv);
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* wp = wl + ((a * 3 + b) * 3 + j) * p.C + cg * 8;
                    const f32x4 w0v = *(const f32x4*)wp, w1v = *(const f32x4*)(wp + 4);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float x = xv[0][b][v + j];
                        acc[v][0] += x * w0v[0]; acc[v][1] += x * w0v[1]; acc[v][2] += x * w0v[2]; acc[v][3] += x * w0v[3];
                        acc[v][4] += x * w1v[0]; acc[v][5] += x * w1v[1]; acc[v][6] += x * w1v[2]; acc[v][7] += x * w1v[3];
                    }
                }
        }
        T* ob = (T*)p.out + (((size_t)n * p.D + d) * p.H + h) * p.W * p.C + cg * 8;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (w0 + v >= p.W) continue;
            store8<T>(ob + (size_t)(w0 + v) * p.C, acc[v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float rr = rnd<T>(acc[v][j]); s1[j] += rr; s2[j] += rr * rr; }
        }
    }
    if (!p.sums) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) { part[(2 * j) * 256 + tid] = live ? s1[j] : 0.f; part[(2 * j + 1) * 256 + tid] = live ? s2[j] : 0.f; }
    __syncthreads();
    const int stripe = blockIdx.x & (VG_STRIPES - 1);
    float* dst = p.sums + ((size_t)stripe * gridDim.y + n) * p.C * 2;
    for (int o = tid; o < p.C * 2; o += 256) {
        const int ch = o >> 1, mom = o & 1, g8 = ch >> 3, j = ch & 7;
        const float* src = part + (2 * j + mom) * 256 + g8;
        float a = 0.f;
        for (int t = 0; t < qpb; ++t) a += src[t * gpc];
        atomicAdd(&dst[o], a);
    }
    if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.y, p.C, gridDim.x * gridDim.y, (int*)part);
}

// weight gradient: thread = (quad, 8-channel group, d-offset a): 3 x 3 x 8 accumulators dW[a][b][j][c] (+ bias gradient in the
// a = 0 role), reduced over the block through LDS in chunks of 20 values, one atomic per (block, value)
template <typename T, typename S>
__global__ __launch_bounds__(256) void c1k3_wgrad_kernel(C1K3 p) {
    __shared__ float part[20 * 256];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int gpc = p.C >> 3, roles = 3 * gpc, qpb = 256 / roles;
    const int role = tid % roles, cg = role % gpc, a = role / gpc, ql = tid / roles;
    const bool live = tid < roles * qpb;
    if (p.scale) { p.sc = p.scale[n]; p.sf = p.shift[n]; }
    float acc[80];                                  // [b][j][c] = 72, then 8 bias sums
#pragma unroll
    for (int i = 0; i < 80; ++i) acc[i] = 0.f;
    const int64_t nq = (int64_t)p.D * p.H * p.W4;
    if (live)
    for (int64_t q = (int64_t)blockIdx.x * qpb + ql; q < nq; q += (int64_t)gridDim.x * qpb) {
        const int wq = (int)(q % p.W4); const int64_t r = q / p.W4;
        const int h = (int)(r % p.H), d = (int)(r / p.H), w0 = wq * 4;
        float xv[1][3][6];
        c1_window<T, S, 1>(p, n, d, h, w0, a, xv);
        const T* yb = (const T*)p.dy + (((size_t)n * p.D + d) * p.H + h) * p.W * p.C + cg * 8;
        Raw8<T> yr[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) raw_load(yr[v], yb + (size_t)min(w0 + v, p.W - 1) * p.C);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float g[8]; raw_unpack(yr[v], g);
            const bool okv = w0 + v < p.W;
#pragma unroll
            for (int c = 0; c < 8; ++c) g[c] = okv ? g[c] : 0.f;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float x = xv[0][b][v + j];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[(b * 3 + j) * 8 + c] += x * g[c];
                }
            if (a == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[72 + c] += g[c];
            }
        }
    }
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {          // unrolled: the accumulators must stay in registers (static indices)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 20; ++i) part[i * 256 + tid] = live ? acc[chunk * 20 + i] : 0.f;
        __syncthreads();
        // outputs of this chunk: (role, i) -> sum over the block's quads
        for (int o = tid; o < roles * 20; o += 256) {
            const int i = o / roles, ro = o - i * roles;
            const float* src = part + i * 256 + ro;
            float sacc = 0.f;
            for (int t = 0; t < qpb; ++t) sacc += src[t * roles];
            const int val = chunk * 20 + i, rcg = ro % gpc, ra = ro / gpc;
            if (val < 72) {
                const int b = val / 24, j = (val / 8) % 3, c = val & 7;
                atomicAdd(&p.dw[(size_t)(((ra * 3 + b) * 3) + j) * p.C + rcg * 8 + c], sacc);
            } else if (ra == 0 && p.db) atomicAdd(&p.db[rcg * 8 + (val - 72)], sacc);
        }
    }
}

bool c1k3_fill(const vg_conv_desc* d, C1K3& p) {
    if (d->c_src0 != 1 || d->src1 || d->c_src1 || d->wpack != 3 || d->ntaps != 9 || d->istr != 1 || d->noise || d->nclass > 1) return false;
    if (d->Cout < 8 || d->Cout > 32 || (d->Cout % 8)) return false;
    if (d->OD != d->D || d->OH != d->H || d->OW != d->W) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tap_d[t] != d->tap_d[0] + t / 3 || d->tap_h[t] != d->tap_h[0] + t % 3 || d->tap_w[t] != 0) return false;
    p.x = d->src0; p.x_f32 = d->src_f32; p.sc = 1.f; p.sf = 0.f; p.scale = d->in_scale; p.shift = d->in_shift; p.act = d->act;
    p.pad_mode = d->pad_mode; p.D = d->D; p.H = d->H; p.W = d->W; p.C = d->Cout; p.W4 = (d->W + 3) / 4;
    p.td0 = d->tap_d[0]; p.th0 = d->tap_h[0]; p.tw0 = d->wpack_wmin;
    p.CK = d->CK; p.Ktot = ((9 * d->CK + 31) / 32) * 32;
    return true;
}

bool pw_enabled() { return vg_tune("PW", 1) != 0; }
// common shape test: one centre tap, unit strides, whole grid, plain single source
bool pw_shape_ok(const vg_conv_desc* d) {
    if (!pw_enabled() || d->ntaps != 1 || d->tap_d[0] || d->tap_h[0] || d->tap_w[0]) return false;
    if (d->istr != 1 || d->src1 || d->c_src1 || d->src0_shift || d->noise || d->wpack || d->nclass > 1) return false;
    if (d->OD != d->D || d->OH != d->H || d->OW != d->W) return false;
    return true;
}
int pw_blocks(int64_t work_items, int N) {
    int64_t b = (work_items + 255) / 256;
    const int64_t cap = 2047 / (N > 0 ? N : 1);          // odd cap: see anb_grid (HBM channel aliasing of power-of-two strides)
    if (b > cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

template <bool GEO, bool ACC, bool STATS>
bool pwg_launch(int KS, int NB, dim3 grid, hipStream_t s, const PWG& p) {
#define PWG_CASE(ks, nb) if (KS == ks && NB == nb) { hipLaunchKernelGGL((pw_gemm_kernel<ks, nb, GEO, ACC, STATS>), grid, dim3(256), 0, s, p); return true; }
    PWG_CASE(1, 1) PWG_CASE(1, 2) PWG_CASE(1, 3) PWG_CASE(1, 4) PWG_CASE(1, 6) PWG_CASE(2, 1) PWG_CASE(2, 2) PWG_CASE(3, 2)
    PWG_CASE(4, 4) PWG_CASE(6, 4) PWG_CASE(2, 8)
#undef PWG_CASE
    return false;
}
template <int SPLIT>
bool pwg_launch_split(int KS, int NB, dim3 grid, hipStream_t s, const PWG& p) {
#define PWG_CASE(ks, nb) if (KS == ks && NB == nb) { hipLaunchKernelGGL((pw_gemm_kernel<ks, nb, false, true, false, SPLIT>), grid, dim3(256), 0, s, p); return true; }
    PWG_CASE(1, 3) PWG_CASE(1, 6) PWG_CASE(2, 6)
#undef PWG_CASE
    return false;
}
bool pwg_case_ok(int KS, int NB) {
    static const int tab[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 6}, {2, 1}, {2, 2}, {3, 2}, {4, 4}, {6, 4}, {2, 8}};
    for (auto& t : tab) if (t[0] == KS && t[1] == NB) return true;
    return false;
}
// the multi-channel 1x1x1 case: VG_OK when launched, 1 when the shape is not served
int pw_gemm_conv(const vg_conv_desc* d, hipStream_t s) {
    if (!pw_enabled() || !vg_tune("PW_GEMM", 1)) return 1;
    if (d->ntaps != 1 || d->tap_d[0] || d->tap_h[0] || d->tap_w[0] || d->noise || d->wpack || d->nclass > 1) return 1;
    if (d->f32 || d->src_f32 || d->out_f32 || d->in_scale || d->act != VG_ACT_NONE || d->res || d->tanh_out) return 1;
    const int Cin = d->c_src0 + d->c_src1;
    if (Cin < 8 || (Cin % 8) || (d->c_src1 && (d->c_src0 % 8)) || d->Cout < 16 || (d->Cout % 16) || (d->CK % 8)) return 1;
    if (d->c_src1 && !d->src1) return 1;
    if (d->src0_shift && ((d->D | d->H | d->W) & 1)) return 1;
    if (d->istr < 1 || d->ostr < 1 || (d->OD - 1) * d->istr >= d->D || (d->OH - 1) * d->istr >= d->H || (d->OW - 1) * d->istr >= d->W) return 1;
    if (d->accumulate && d->out_sums) return 1;
    const int64_t SO = (int64_t)d->OD * d->OH * d->OW;
    if (SO < 1 || SO > (1 << 30)) return 1;
    const int KS = (Cin + 31) / 32, NB = d->Cout / 16;
    if (!pwg_case_ok(KS, NB)) return 1;
    const bool geo = d->src0_shift || d->istr != 1 || d->ostr != 1 || d->ooff_d || d->ooff_h || d->ooff_w || d->BD != d->OD || d->BH != d->OH
                     || d->BW != d->OW || d->OD != d->D || d->OH != d->H || d->OW != d->W;
    PWG p = {};
    p.x0 = (const bf16_t*)d->src0; p.x1 = (const bf16_t*)(d->c_src1 ? d->src1 : d->src0); p.c0 = d->c_src0; p.c1 = d->c_src1 ? d->c_src1 : 1;
    p.Cin = Cin; p.sh = d->src0_shift ? 1 : 0;
    p.w = (const bf16_t*)d->wpacked; p.CK = d->CK; p.kc_pad = ((d->CK + 31) / 32) * 32; p.Ktot = ((Cin + d->CK - 1) / d->CK) * p.kc_pad;
    p.bias = d->bias; p.out = (bf16_t*)d->out; p.sums = d->out_sums;
    p.N = d->N; p.ID = d->D; p.IH = d->H; p.IW = d->W; p.OD = d->OD; p.OH = d->OH; p.OW = d->OW; p.istr = d->istr;
    p.BD = d->BD; p.BH = d->BH; p.BW = d->BW; p.ostr = d->ostr; p.od0 = d->ooff_d; p.oh0 = d->ooff_h; p.ow0 = d->ooff_w;
    p.Cout = d->Cout; p.SO = (int)SO; p.stamps = g_vg_stamps;
    p.fin = vg_fin_of(d);
    const int MS = (KS * NB <= 2) ? 4 : 2;
    const int64_t nwt = (SO + MS * 16 - 1) / (MS * 16);
    int64_t b = (nwt + 3) / 4;
    const int capt = vg_tune("PW_GEMM_CAP", 2047);
    const int64_t cap = (capt / d->N) > 0 ? (capt / d->N) : 1;
    if (b > cap) b = cap;
    const dim3 grid((int)b, d->N);
    if (vg_dry("pw_gemm<%d,%d,g%d,a%d>", KS, NB, (geo || !d->accumulate) ? 1 : 0, d->accumulate ? 1 : 0)) return VG_OK;
    bool ok;
    if (!d->accumulate) { ok = pwg_launch<true, false, true>(KS, NB, grid, s, p); if (ok && p.sums && p.fin.ticket) vg_fin_done = true; }
    else ok = geo ? pwg_launch<true, true, false>(KS, NB, grid, s, p) : pwg_launch<false, true, false>(KS, NB, grid, s, p);
    return ok ? vg_check_launch() : 1;
}

// Data gradient of a decoder shortcut fused with the backward of UpSampling3D + concatenate: d describes the accumulating launch
// (src0 = gradient of the shortcut's output, out = gradient of the virtual concat with the conv branch's part already in it);
// instead of adding into `out` and leaving the split / 2x2x2 sum to vg_concat_bwd, the sums go to dskip / dlow directly.
int pw_gemm_split(const vg_conv_desc* d, const vg_actnorm_bwd_desc* nb_, void* dlow, void* dskip, int c_low, int acc, hipStream_t s) {
    if (!pw_enabled() || !vg_tune("PW_GEMM", 1) || !vg_tune("PW_SPLIT", 1)) return 1;
    if (d->ntaps != 1 || d->tap_d[0] || d->tap_h[0] || d->tap_w[0] || d->noise || d->wpack || d->nclass > 1) return 1;
    if (d->f32 || d->src_f32 || d->out_f32 || d->in_scale || d->act != VG_ACT_NONE || d->res || d->tanh_out || d->out_sums || d->bias) return 1;
    const int Cin = d->c_src0;
    if (d->c_src1 || d->src0_shift || Cin < 8 || (Cin % 8) || d->Cout < 32 || (d->Cout % 16) || (d->CK % 8) || (!d->accumulate && !nb_)) return 1;
    if (nb_) {          // the conv branch's (InstanceNorm -> act) backward computed in the launch (SPLIT == 2)
        if (!nb_->norm || !nb_->g_padded || nb_->f32 || nb_->x_f32 || !nb_->g || !nb_->x || !nb_->x1 || !nb_->x0_shift || nb_->c_x0 != c_low) return 1;
        if (nb_->C != d->Cout || nb_->N != d->N || nb_->D != d->D || nb_->H != d->H || nb_->W != d->W || !nb_->gamma || !nb_->mean || !nb_->rstd || !nb_->red) return 1;
        if (d->D < 4 || d->H < 4 || d->W < 4 || d->D > 1023 || d->H > 1023 || d->W > 1023) return 1;
    }
    if (c_low < 16 || (c_low % 16) || c_low >= d->Cout) return 1;
    if (d->istr != 1 || d->ostr != 1 || d->ooff_d || d->ooff_h || d->ooff_w || d->BD != d->OD || d->BH != d->OH || d->BW != d->OW
        || d->OD != d->D || d->OH != d->H || d->OW != d->W || ((d->D | d->H | d->W) & 1)) return 1;
    const int64_t SO = (int64_t)d->OD * d->OH * d->OW;
    if (SO < 8 || SO > (1 << 30)) return 1;
    const int KS = (Cin + 31) / 32, NB = d->Cout / 16;
    if (!((KS == 1 && (NB == 3 || NB == 6)) || (KS == 2 && NB == 6))) return 1;
    PWG p = {};
    p.x0 = (const bf16_t*)d->src0; p.x1 = p.x0; p.c0 = Cin; p.c1 = 1; p.Cin = Cin; p.sh = 0;
    p.w = (const bf16_t*)d->wpacked; p.CK = d->CK; p.kc_pad = ((d->CK + 31) / 32) * 32; p.Ktot = ((Cin + d->CK - 1) / d->CK) * p.kc_pad;
    p.bias = nullptr; p.out = (bf16_t*)d->out; p.sums = nullptr;
    p.N = d->N; p.ID = d->D; p.IH = d->H; p.IW = d->W; p.OD = d->OD; p.OH = d->OH; p.OW = d->OW; p.istr = 1;
    p.BD = d->BD; p.BH = d->BH; p.BW = d->BW; p.ostr = 1;
    p.Cout = d->Cout; p.SO = (int)SO; p.stamps = g_vg_stamps;
    p.dlow = (bf16_t*)dlow; p.dskip = (bf16_t*)dskip; p.sc0 = c_low; p.sacc = acc;
    if (nb_) {
        p.ng = (const bf16_t*)nb_->g; p.nx0 = (const bf16_t*)nb_->x; p.nx1 = (const bf16_t*)nb_->x1;
        p.n_scale = nb_->scale; p.n_shift = nb_->shift; p.n_mean = nb_->mean; p.n_rstd = nb_->rstd; p.n_gamma = nb_->gamma; p.n_mult = nb_->mult;
        p.n_red = nb_->red; p.n_dgamma = (nb_->dgamma && nb_->dbeta) ? nb_->dgamma : nullptr; p.n_dbeta = nb_->dbeta; p.n_act = nb_->act;
    }
    const int MS = (KS * NB <= 2) ? 4 : 2;
    const int64_t nwt = (SO + MS * 16 - 1) / (MS * 16);
    int64_t b = (nwt + 3) / 4;
    const int capt = vg_tune("PW_GEMM_CAP", 2047);
    const int64_t cap = (capt / d->N) > 0 ? (capt / d->N) : 1;
    if (b > cap) b = cap;
    if (vg_dry("pw_gemm_split<%d,%d,n%d>", KS, NB, nb_ ? 1 : 0)) return VG_OK;
    const bool ok = nb_ ? pwg_launch_split<2>(KS, NB, dim3((int)b, d->N), s, p) : pwg_launch_split<1>(KS, NB, dim3((int)b, d->N), s, p);
    return ok ? vg_check_launch() : 1;
}

int pw_wgrad_cc(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch,
                int64_t scratch_bytes, hipStream_t s) {
    if (!pw_enabled() || !vg_tune("PW_WGRAD_CC", 1)) return 1;
    if (T_total != 1 || d->ntaps != 1 || d->tap_d[0] || d->tap_h[0] || d->tap_w[0] || d->noise || d->wpack || d->nclass > 1) return 1;
    if (d->f32 || d->src_f32 || dy_f32 || d->in_scale || d->act != VG_ACT_NONE) return 1;
    const int Cin = d->c_src0 + d->c_src1;
    if (Cin < 8 || (Cin % 8) || (d->c_src1 && (d->c_src0 % 8)) || d->Cout < 16 || (d->Cout % 16)) return 1;
    if (d->c_src1 && !d->src1) return 1;
    if (d->src0_shift && ((d->D | d->H | d->W) & 1)) return 1;
    if (d->istr < 1 || (d->OD - 1) * d->istr >= d->D || (d->OH - 1) * d->istr >= d->H || (d->OW - 1) * d->istr >= d->W) return 1;
    const int64_t SO = (int64_t)d->OD * d->OH * d->OW;
    if (SO < 1 || SO > (1 << 30)) return 1;
    PWW p = {};
    p.gin = Cin / 8; p.roles = p.gin * (d->Cout / 16);
    // measured against the MFMA weight-gradient kernel (kernel + slab reduction, us): 6 roles (48 -> 16 at 128^3) 123 vs 164,
    // 4 roles (16 -> 32 s2) 30 vs 41, 16 roles (32 -> 64 s2) 23 vs 29, but 24 roles (96 -> 32 at 64^3) 71 vs 55: the packed-f32
    // FMAs (150 instructions per voxel and role, 4 cycles each on a 16-lane SIMD) become the bound as the channel product grows
    if (p.roles > vg_tune("PW_WGRAD_ROLES", 16)) return 1;
    p.rp = pow2_ceil(p.roles); p.vpb = 256 / p.rp;
    p.x0 = (const bf16_t*)d->src0; p.x1 = (const bf16_t*)(d->c_src1 ? d->src1 : d->src0); p.c0 = d->c_src0; p.c1 = d->c_src1 ? d->c_src1 : 1;
    p.Cin = Cin; p.sh = d->src0_shift ? 1 : 0; p.dy = (const bf16_t*)dy; p.Cout = d->Cout;
    p.N = d->N; p.ID = d->D; p.IH = d->H; p.IW = d->W; p.OD = d->OD; p.OH = d->OH; p.OW = d->OW; p.istr = d->istr; p.SO = (int)SO;
    p.dw = dw; p.db = db; p.dw_elems = Cin * d->Cout;
    int64_t b = (SO + (int64_t)p.vpb * 8 - 1) / ((int64_t)p.vpb * 8);
    const int capt = vg_tune("PW_WGRAD_CAP", 511);
    const int64_t cap = (capt / d->N) > 0 ? (capt / d->N) : 1;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    const int nslab = (int)b * d->N;
    p.part = (nslab > 4 && scratch && (int64_t)nslab * p.dw_elems * 4 <= scratch_bytes) ? scratch : nullptr;
    if (vg_dry("pw_wgrad_cc<r%d>|part%d", p.roles, p.part ? 1 : 0)) return VG_OK;
    hipLaunchKernelGGL(pw_wgrad_cc_kernel, dim3((int)b, d->N), dim3(256), 0, s, p);
    if (p.part) vg_launch_reduce_partials(p.part, nslab, p.dw_elems, dw, s);
    return vg_check_launch();
}

}  // namespace

extern "C" int vg_shortcut_dgrad_concat(const vg_conv_desc* d, void* dlow, void* dskip, int c_low, int accumulate, vg_stream_t stream) {
    vg_begin();
    if (!d || !d->src0 || !d->out || !d->wpacked || !dlow || !dskip) return VG_EINVAL;
    return pw_gemm_split(d, nullptr, dlow, dskip, c_low, accumulate, (hipStream_t)stream);
}
extern "C" int vg_shortcut_dgrad_concat_norm(const vg_conv_desc* d, const vg_actnorm_bwd_desc* b, void* dlow, void* dskip, int c_low,
                                             int accumulate, vg_stream_t stream) {
    vg_begin();
    if (!d || !b || !d->src0 || !d->wpacked || !dlow || !dskip) return VG_EINVAL;
    if (!vg_tune("PW_SPLIT_NORM", 1)) return 1;
    return pw_gemm_split(d, b, dlow, dskip, c_low, accumulate, (hipStream_t)stream);
}

// returns VG_OK when the launch was done here, 1 when the shape is not one of the pointwise cases (caller continues), < 0 on error
int vg_pointwise_conv(const vg_conv_desc* d, hipStream_t s) {
    if (pw_enabled() && !d->res && !d->tanh_out && !d->accumulate && d->ostr == 1 && !d->ooff_d && !d->ooff_h && !d->ooff_w
        && d->BD == d->OD && d->BH == d->OH && d->BW == d->OW && !(d->out_f32 && !d->f32)) {
        { const int mrc = c1m_fwd(d, s); if (mrc <= 0) return mrc; }            // 16-bit storage: the MFMA kernels of vg_c1k3.hip
        C1K3 c;
        if (c1k3_fill(d, c)) {
            c.w = d->wpacked; c.bias = d->bias; c.out = d->out; c.sums = d->out_sums;
            c.fin = vg_fin_of(d);
            const int qpb = 256 / (c.C >> 3);
            int64_t b = ((int64_t)c.D * c.H * c.W4 + qpb - 1) / qpb;
            const int64_t cap = (2047 / d->N) > 0 ? (2047 / d->N) : 1;
            if (b > cap) b = cap;
            const dim3 grid((int)b, d->N);
            if (vg_dry("c1k3_fwd<%s,%s>", d->f32 ? "f32" : "bf16", d->src_f32 ? "f32" : "bf16")) return VG_OK;
            if (d->f32) {
                if (d->src_f32) hipLaunchKernelGGL((c1k3_fwd_kernel<float, float>), grid, dim3(256), 0, s, c);
                else hipLaunchKernelGGL((c1k3_fwd_kernel<float, bf16_t>), grid, dim3(256), 0, s, c);
            } else {
                if (d->src_f32) hipLaunchKernelGGL((c1k3_fwd_kernel<bf16_t, float>), grid, dim3(256), 0, s, c);
                else hipLaunchKernelGGL((c1k3_fwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, c);
            }
            if (c.sums && c.fin.ticket) vg_fin_done = true;
            return vg_check_launch();
        }
    }
    { const int grc = pw_gemm_conv(d, s); if (grc <= 0) return grc; }
    if (!pw_shape_ok(d) || d->res) return 1;
    if (d->ostr != 1 || d->ooff_d || d->ooff_h || d->ooff_w || d->BD != d->OD || d->BH != d->OH || d->BW != d->OW) return 1;
    // tanh_out with accumulate means out = tanh(out + value) (vg_conv_desc): these one-pass kernels accumulate AFTER the activation
    // slot, so the combination goes to the MFMA path, whose epilogue implements it -- never a silently un-activated result
    if (d->tanh_out && d->accumulate) return 1;
    const int Cin = d->c_src0;
    PW p = {};
    p.x = d->src0; p.x_f32 = d->src_f32; p.scale = d->in_scale; p.shift = d->in_shift; p.act = d->act;
    p.w = d->wpacked; p.CK = d->CK; p.kc_pad = ((d->CK + 31) / 32) * 32;
    p.bias = d->bias; p.out = d->out; p.out_f32 = (d->out_f32 || d->f32) ? 1 : 0; p.accumulate = d->accumulate; p.tanh_out = d->tanh_out;
    p.sums = d->out_sums; p.N = d->N; p.S = (int64_t)d->D * d->H * d->W;
    if (d->Cout == 1 && Cin >= 8 && Cin <= 32 && (Cin % 8) == 0) {
        p.C = Cin;
        const dim3 grid(pw_blocks((p.S + 1) / 2, d->N), d->N);
        if (vg_dry("pw_cto1<%s,%d>", d->f32 ? "f32" : "bf16", Cin / 8 > 4 ? 4 : Cin / 8)) return VG_OK;
#define PW_CTO1(T)                                                                                                      \
        switch (Cin / 8) {                                                                                              \
            case 1: hipLaunchKernelGGL((pw_cto1_kernel<T, 1>), grid, dim3(256), 0, s, p); break;                        \
            case 2: hipLaunchKernelGGL((pw_cto1_kernel<T, 2>), grid, dim3(256), 0, s, p); break;                        \
            case 3: hipLaunchKernelGGL((pw_cto1_kernel<T, 3>), grid, dim3(256), 0, s, p); break;                        \
            default: hipLaunchKernelGGL((pw_cto1_kernel<T, 4>), grid, dim3(256), 0, s, p); break;                       \
        }
        if (d->f32) { PW_CTO1(float) } else { PW_CTO1(bf16_t) }
#undef PW_CTO1
        return vg_check_launch();
    }
    if (Cin == 1 && d->Cout >= 8 && d->Cout <= 256 && (d->Cout % 8) == 0 && !d->tanh_out && !(d->out_f32 && !d->f32)) {
        p.C = d->Cout;
        p.kc_pad = ((d->CK + 31) / 32) * 32;                 // one tap, one chunk: row stride of the packed operand
        const int vpb = 256 / (p.C >> 3);
        const dim3 grid(pw_blocks((p.S + 3) / 4 * (256 / vpb), d->N), d->N);
        if (vg_dry("pw_1toc<%s>", d->f32 ? "f32" : "bf16")) return VG_OK;
        p.fin = vg_fin_of(d);
        if (d->f32) hipLaunchKernelGGL((pw_1toc_kernel<float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((pw_1toc_kernel<bf16_t>), grid, dim3(256), 0, s, p);
        if (p.sums && p.fin.ticket) vg_fin_done = true;
        return vg_check_launch();
    }
    if ((Cin == 8 || Cin == 16) && d->Cout >= 16 && d->Cout <= 128 && (d->Cout % 8) == 0 && !d->in_scale && d->act == VG_ACT_NONE
        && !d->tanh_out && !d->out_sums && !d->src_f32 && !(d->out_f32 && !d->f32)) {
        p.C = d->Cout;
        const int nchunks = (Cin + d->CK - 1) / d->CK;
        const int Ktot = nchunks * p.kc_pad;
        const int vpb = 256 / (p.C >> 3);
        const dim3 grid(pw_blocks((p.S + 1) / 2 * (256 / vpb), d->N), d->N);
        if (vg_dry("pw_ctoc<%s,%d>", d->f32 ? "f32" : "bf16", Cin / 8)) return VG_OK;
        if (d->f32) {
            if (Cin == 8) hipLaunchKernelGGL((pw_ctoc_kernel<float, 1>), grid, dim3(256), 0, s, p, Cin, Ktot);
            else hipLaunchKernelGGL((pw_ctoc_kernel<float, 2>), grid, dim3(256), 0, s, p, Cin, Ktot);
        } else {
            if (Cin == 8) hipLaunchKernelGGL((pw_ctoc_kernel<bf16_t, 1>), grid, dim3(256), 0, s, p, Cin, Ktot);
            else hipLaunchKernelGGL((pw_ctoc_kernel<bf16_t, 2>), grid, dim3(256), 0, s, p, Cin, Ktot);
        }
        return vg_check_launch();
    }
    return 1;
}

int vg_pointwise_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch,
                       int64_t scratch_bytes, hipStream_t s) {
    if (pw_enabled()) { const int mrc = c1m_wgrad(d, dy, dy_f32, T_total, dw, db, scratch, scratch_bytes, s); if (mrc <= 0) return mrc; }
    if (pw_enabled() && T_total == 9 && !(dy_f32 && !d->f32)) {
        C1K3 c;
        if (c1k3_fill(d, c)) {
            c.dy = dy; c.dw = dw; c.db = db;
            const int qpb = 256 / (3 * (c.C >> 3));
            int64_t b = ((int64_t)c.D * c.H * c.W4 + (int64_t)qpb * 8 - 1) / ((int64_t)qpb * 8);
            const int64_t cap = (767 / d->N) > 0 ? (767 / d->N) : 1;
            if (b > cap) b = cap;
            if (b < 1) b = 1;
            const dim3 grid((int)b, d->N);
            if (vg_dry("c1k3_wgrad<%s,%s>", d->f32 ? "f32" : "bf16", d->src_f32 ? "f32" : "bf16")) return VG_OK;
            if (d->f32) {
                if (d->src_f32) hipLaunchKernelGGL((c1k3_wgrad_kernel<float, float>), grid, dim3(256), 0, s, c);
                else hipLaunchKernelGGL((c1k3_wgrad_kernel<float, bf16_t>), grid, dim3(256), 0, s, c);
            } else {
                if (d->src_f32) hipLaunchKernelGGL((c1k3_wgrad_kernel<bf16_t, float>), grid, dim3(256), 0, s, c);
                else hipLaunchKernelGGL((c1k3_wgrad_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, c);
            }
            return vg_check_launch();
        }
    }
    { const int wrc = pw_wgrad_cc(d, dy, dy_f32, T_total, dw, db, scratch, scratch_bytes, s); if (wrc <= 0) return wrc; }
    if (!pw_shape_ok(d) || T_total != 1) return 1;
    const int Cin = d->c_src0;
    PW p = {};
    p.x = d->src0; p.x_f32 = d->src_f32; p.scale = d->in_scale; p.shift = d->in_shift; p.act = d->act;
    p.dy = dy; p.dy_f32 = dy_f32; p.dw = dw; p.db = db; p.N = d->N; p.S = (int64_t)d->D * d->H * d->W;
    const bool c_to_1 = d->Cout == 1 && Cin >= 8 && Cin <= 256 && (Cin % 8) == 0;
    const bool one_to_c = Cin == 1 && d->Cout >= 8 && d->Cout <= 256 && (d->Cout % 8) == 0 && !(dy_f32 && !d->f32);
    if (!c_to_1 && !one_to_c) return 1;
    p.C = c_to_1 ? Cin : d->Cout;
    const int vpb = 256 / (p.C >> 3);
    // few blocks: every block ends with 2*C same-address atomics (the reason the MFMA weight gradient uses partial slabs)
    int64_t b = (p.S + (int64_t)vpb * 16 - 1) / ((int64_t)vpb * 16);
    const int64_t cap = (511 / d->N) > 0 ? (511 / d->N) : 1;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    const dim3 grid((int)b, d->N);
    if (vg_dry("pw_wgrad<%s,%s>", d->f32 ? "f32" : "bf16", c_to_1 ? "cto1" : "1toc")) return VG_OK;
    if (c_to_1) {
        if (d->f32) hipLaunchKernelGGL((pw_wgrad_kernel<float, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, true>), grid, dim3(256), 0, s, p);
    } else {
        if (d->f32) hipLaunchKernelGGL((pw_wgrad_kernel<float, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, false>), grid, dim3(256), 0, s, p);
    }
    return vg_check_launch();
}
