// vg_elem.hip -- HBM-bound elementwise / reduction kernels around the convolutions:
// InstanceNorm scale/shift finalisation, the backward of (InstanceNorm -> activation -> dropout) with the
// transpose of ReflectionPadding3D folded into the read, the backward of the virtual upsample+concat,
// tanh backward, dtype copies and the counter-based RNG for GaussianNoise / SpatialDropout3D.
#include "vg_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// InstanceNorm finalise (tfa InstanceNormalization: biased variance, eps inside rsqrt)
// ------------------------------------------------------------------------------------------------
__global__ void in_finalize_kernel(const float* sums0, int c0, float cnt0, const float* sums1, int c1, float cnt1,
                                   const float* gamma, const float* beta, const float* mult, int N, float eps,
                                   float* scale, float* shift, float* mean_o, float* rstd_o) {
    const int C = c0 + c1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    float s = 0.f, ss = 0.f, cnt;
    if (c < c0) {
        for (int t = 0; t < VG_STRIPES; ++t) { s += sums0[(((size_t)t * N + n) * c0 + c) * 2]; ss += sums0[(((size_t)t * N + n) * c0 + c) * 2 + 1]; }
        cnt = cnt0;
    } else {
        for (int t = 0; t < VG_STRIPES; ++t) { s += sums1[(((size_t)t * N + n) * c1 + (c - c0)) * 2]; ss += sums1[(((size_t)t * N + n) * c1 + (c - c0)) * 2 + 1]; }
        cnt = cnt1;
    }
    const float mean = s / cnt;
    float var = ss / cnt - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
    float sc = (gamma ? gamma[c] : 1.f) * rstd;
    float sh = (beta ? beta[c] : 0.f) - mean * sc;
    if (mult) { const float m = mult[i]; sc *= m; sh *= m; }
    scale[i] = sc; shift[i] = sh;
    if (mean_o) mean_o[i] = mean;
    if (rstd_o) rstd_o[i] = rstd;
}

extern "C" int vg_in_finalize(const float* sums0, int c0, float count0, const float* sums1, int c1, float count1,
                              const float* gamma, const float* beta, const float* mult, int N, float eps,
                              float* scale, float* shift, float* mean, float* rstd, vg_stream_t stream) {
    vg_begin();
    if (!sums0 || c0 < 1 || c1 < 0 || (c1 > 0 && !sums1) || !scale || !shift || N < 1) return VG_EINVAL;
    const int total = N * (c0 + c1);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums0, c0,
                       count0, sums1, c1, count1, gamma, beta, mult, N, eps, scale, shift, mean, rstd);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// backward of a = mult*act(x*scale+shift): stats pass and apply pass
// ------------------------------------------------------------------------------------------------
struct ANB {
    const void* g; int g_padded;
    const void* x; int x_f32;
    const void* x1; int c_x0, x0_shift;
    int N, D, H, W, C;
    const float* scale; const float* shift; const float* mult;
    int act, norm;
    const float* gamma; const float* mean; const float* rstd;
    float* red;
    int* ticket; float* dgamma; float* dbeta;      // fold by the last workgroup (stats kernel)
    void* dx; int dx_f32, accumulate, dx_cstride, dx_coff;
    int gpc, vpb;      // channel groups per voxel, voxels per block-iteration
    int alias_n0, alias_sh, pgrad_n;      // samples >= alias_n0 read x and the per-(n, c) constants of sample n - alias_sh; parameter gradients from samples < pgrad_n (0: all)
};
__device__ __forceinline__ int anb_nx(const ANB& p, int n) { return (p.alias_n0 > 0 && n >= p.alias_n0) ? n - p.alias_sh : n; }

// per-thread channel constants (the thread owns channels c..c+VEC-1 of sample n for its whole walk)
template <int VEC> struct ChanK { float sc[VEC], sh[VEC], mu[VEC], rs[VEC], ml[VEC]; };
__device__ __forceinline__ void ldvec(const float* p, float* o, int n) {       // n = 8 (two 16-byte loads) or 1
    if (n == 8) {
        const __attribute__((address_space(1))) f32x4* q = (const __attribute__((address_space(1))) f32x4*)(uintptr_t)p;
        const f32x4 a = q[0], b = q[1];
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    } else o[0] = *p;
}
template <int VEC>
__device__ __forceinline__ void load_chank(const ANB& p, int n, int c, ChanK<VEC>& k) {
    const int nc = anb_nx(p, n) * p.C + c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { k.sc[j] = 1.f; k.sh[j] = 0.f; k.mu[j] = 0.f; k.rs[j] = 0.f; k.ml[j] = 1.f; }
    if (p.scale) { ldvec(p.scale + nc, k.sc, VEC); ldvec(p.shift + nc, k.sh, VEC); }
    if (p.norm) { ldvec(p.mean + nc, k.mu, VEC); ldvec(p.rstd + nc, k.rs, VEC); }
    if (p.mult) ldvec(p.mult + nc, k.ml, VEC);
}

template <typename T, int VEC> struct RawV;
template <typename T> struct RawV<T, 8> { Raw8<T> r; };
template <typename T> struct RawV<T, 1> { float r; };
template <typename T> __device__ __forceinline__ void rawv_load(RawV<T, 8>& o, const T* p) { raw_load(o.r, p); }
template <typename T> __device__ __forceinline__ void rawv_load(RawV<T, 1>& o, const T* p) { o.r = ld_global(p); }
template <typename T> __device__ __forceinline__ void rawv_unpack(const RawV<T, 8>& o, float* v) { raw_unpack(o.r, v); }
template <typename T> __device__ __forceinline__ void rawv_unpack(const RawV<T, 1>& o, float* v) { v[0] = o.r; }

// The walk shared by the statistics and the apply pass.  A thread owns VEC channels and visits voxels
// v0, v0 + stride, ...; UB voxels are handled per iteration with all their loads (upstream gradient at the interior
// position of the padded grid, pre-activation input) issued back to back on clamped addresses -- a conditional load
// makes the compiler drain the memory counter before the next one, which is what made the first version of these
// kernels latency-bound at ~1 TB/s.  (d, h, w) advance incrementally (no per-voxel division).  The transpose of the
// reflection pad only adds terms on the two planes next to each face: a rare divergent tail after the main loads.
template <typename T, int VEC, int UB, typename F>
__device__ __forceinline__ void anb_walk_from(const ANB& p, int n, int c, int v, const int stride, const ChanK<VEC>& ck, F&& consume) {
    const int S = p.D * p.H * p.W;
    if (v >= S) return;
    int w = v % p.W, t0 = v / p.W; int h = t0 % p.H, d = t0 / p.H;
    const int sw = stride % p.W, t1 = stride / p.W; const int sh_ = t1 % p.H, sd = t1 / p.H;
    const int PH = p.H + 2, PW = p.W + 2;
    const T* gp = (const T*)p.g + (p.g_padded ? (size_t)n * (p.D + 2) * PH * PW * p.C : (size_t)n * S * p.C) + c;
    const bool need_x = p.act != VG_ACT_NONE || p.norm;
    const bool cat0 = p.x1 && c < p.c_x0;                 // virtual concat: low-resolution source (nearest upsample)
    const int xs = p.x1 ? p.x0_shift : 0;
    const int Hs = p.H >> xs, Ws = p.W >> xs;
    const int cx = p.x1 ? (cat0 ? p.c_x0 : p.C - p.c_x0) : p.C;
    const T* xp = nullptr;
    const int nx = anb_nx(p, n);              // the forward tensors' sample (a second gradient of the same sample: vg_actnorm_bwd_desc::alias_n0)
    if (VEC == 8) xp = p.x1 ? (cat0 ? (const T*)p.x + (size_t)nx * (p.D >> xs) * Hs * Ws * cx + c : (const T*)p.x1 + (size_t)nx * S * cx + (c - p.c_x0))
                            : (const T*)p.x + (size_t)nx * S * cx + c;
    const void* x1p = (const char*)p.x + (size_t)nx * S * (p.x_f32 ? 4 : 2);      // VEC == 1: f32 or bf16 volume
    for (; v < S; v += UB * stride) {
        RawV<T, VEC> gr[UB], xr[UB];
        float x1v[UB];
        int vv[UB], dd[UB], hh[UB], ww[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const bool ok = v + k * stride < S;
            vv[k] = ok ? v + k * stride : -1;
            dd[k] = d; hh[k] = h; ww[k] = w;
            const int vc = ok ? v + k * stride : v;                       // clamped: re-reads voxel 0 of the batch
            const int dc = ok ? d : dd[0], hc = ok ? h : hh[0], wc = ok ? w : ww[0];
            const size_t gi = p.g_padded ? ((size_t)((dc + 1) * PH + hc + 1) * PW + wc + 1) * p.C : (size_t)vc * p.C;
            rawv_load(gr[k], gp + gi);
            if (need_x) {
                if (VEC == 8) {
                    const size_t xi = cat0 ? ((size_t)((dc >> xs) * Hs + (hc >> xs)) * Ws + (wc >> xs)) * cx : (size_t)vc * cx;
                    rawv_load(xr[k], xp + xi);
                } else {
                    x1v[k] = p.x_f32 ? ld_global((const float*)x1p + vc) : ld_global((const bf16_t*)x1p + vc);
                }
            }
            // advance (d, h, w) by the stride
            w += sw; if (w >= p.W) { w -= p.W; ++h; }
            h += sh_; if (h >= p.H) { h -= p.H; ++d; }
            d += sd;
        }
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (vv[k] < 0) continue;
            float g[VEC], x[VEC];
            rawv_unpack(gr[k], g);
            if (need_x) { if (VEC == 8) rawv_unpack(xr[k], x); else x[0] = x1v[k]; }
            if (p.g_padded) {
                const int dk = dd[k], hk = hh[k], wk = ww[k];
                const bool dhb = dk == 1 || dk == p.D - 2 || hk == 1 || hk == p.H - 2;
                const bool wb = wk == 1 || wk == p.W - 2;
                if (dhb) {
                    // reflected copies: every combination of {own, mirrored} per axis except the all-own one (already
                    // loaded).  An axis has at most one mirrored plane here (index 1 -> padded 0, index n-2 -> padded n+1;
                    // both only when n == 3, which the descriptor check excludes for padded grids... handled: second below)
                    const int md = dk == 1 ? 0 : (dk == p.D - 2 ? p.D + 1 : -1);
                    const int mh = hk == 1 ? 0 : (hk == p.H - 2 ? p.H + 1 : -1);
                    const int mw = wk == 1 ? 0 : (wk == p.W - 2 ? p.W + 1 : -1);
                    // n == 3: index 1 is next to both faces -> second mirror
                    const int md2 = (dk == 1 && dk == p.D - 2) ? p.D + 1 : -1;
                    const int mh2 = (hk == 1 && hk == p.H - 2) ? p.H + 1 : -1;
                    const int mw2 = (wk == 1 && wk == p.W - 2) ? p.W + 1 : -1;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b)
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
                                if ((a | b | e) == 0) continue;
                                const int qd = a == 0 ? dk + 1 : (a == 1 ? md : md2);
                                const int qh = b == 0 ? hk + 1 : (b == 1 ? mh : mh2);
                                const int qw = e == 0 ? wk + 1 : (e == 1 ? mw : mw2);
                                if ((qd | qh | qw) < 0) continue;
                                const size_t idx = ((size_t)(qd * PH + qh) * PW + qw) * p.C;
                                if (VEC == 8) {
                                    float r[8]; load8<T>(gp + idx, r);
#pragma unroll
                                    for (int j = 0; j < 8; ++j) g[j] += r[j];
                                } else g[0] += ld1<T>(gp + idx);
                            }
                } else if (wb) {
                    // the common border case (two voxels of every row): one or two mirrored copies along W only
                    const size_t rowi = (size_t)((dk + 1) * PH + hk + 1) * PW;
                    float r[VEC];
                    if (wk == 1) {
                        if (VEC == 8) load8<T>(gp + rowi * p.C, r); else r[0] = ld1<T>(gp + rowi * p.C);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) g[j] += r[j];
                    }
                    if (wk == p.W - 2) {
                        if (VEC == 8) load8<T>(gp + (rowi + p.W + 1) * p.C, r); else r[0] = ld1<T>(gp + (rowi + p.W + 1) * p.C);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) g[j] += r[j];
                    }
                }
            }
            float dn[VEC], xh[VEC];
            const float slope = p.act == VG_ACT_RELU ? 0.f : (p.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float gv = g[j] * ck.ml[j];
                if (need_x) {
                    const float pre = x[j] * ck.sc[j] + ck.sh[j];
                    gv *= (pre > 0.f || p.act == VG_ACT_NONE) ? 1.f : slope;      // TP: LeakyRelu/ReLU grad uses pre > 0
                    xh[j] = (x[j] - ck.mu[j]) * ck.rs[j];
                } else xh[j] = 0.f;
                dn[j] = gv;
            }
            consume(vv[k], dn, xh);
        }
    }
}

template <typename T, int VEC, int UB, typename F>
__device__ __forceinline__ void anb_walk(const ANB& p, int n, int c, int vl, const ChanK<VEC>& ck, F&& consume) {
    anb_walk_from<T, VEC, UB>(p, n, c, (int)(blockIdx.x * p.vpb + vl), (int)(gridDim.x * p.vpb), ck, consume);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void actnorm_stats_kernel(const ANB p) {
    __shared__ float part[16 * 256];         // [value j][thread]: this block's per-thread partial sums
    __shared__ float red[512 * 2];           // [channel][2] block result (C <= 512)
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int nthr = p.gpc * p.vpb;
    const int cg = tid % p.gpc, vl = tid / p.gpc;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (tid < nthr) {
        ChanK<VEC> ck;
        load_chank<VEC>(p, n, cg * VEC, ck);
        anb_walk<T, VEC, 4>(p, n, cg * VEC, vl, ck, [&](int, const float* dn, const float* xh) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s0[j] += dn[j]; s1[j] += dn[j] * xh[j]; }
        });
    }
    // block reduction without atomics: partials to LDS (value-major: conflict-free), then thread (channel, moment)
    // adds the vpb partials of its channel group in a fixed order; ONE contiguous global atomic per (channel, moment)
#pragma unroll
    for (int j = 0; j < VEC; ++j) { part[(2 * j) * 256 + tid] = tid < nthr ? s0[j] : 0.f; part[(2 * j + 1) * 256 + tid] = tid < nthr ? s1[j] : 0.f; }
    __syncthreads();
    for (int o = tid; o < p.C * 2; o += 256) {
        const int ch = o >> 1, mom = o & 1;
        const int g8 = ch / VEC, j = ch - g8 * VEC;
        const float* src = part + (2 * j + mom) * 256 + g8;
        float a = 0.f;
        for (int t = 0; t < p.vpb; ++t) a += src[t * p.gpc];
        red[o] = a;
    }
    __syncthreads();
    const int stripe = blockIdx.x & (VG_STRIPES - 1);
    float* dst = p.red + ((size_t)stripe * p.N + n) * p.C * 2;
    for (int i = tid; i < p.C * 2; i += 256) atomicAdd(&dst[i], red[i]);
}

// bx / gx: this workgroup's index and the number of workgroups of ITS job (a launch may carry two jobs: actnorm_apply2_kernel)
template <typename T, int VEC>
__device__ __forceinline__ void actnorm_apply_body(const ANB& p, const int bx, const int gx) {
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int nthr = p.gpc * p.vpb;
    if (tid >= nthr) return;
    const int cg = tid % p.gpc, vl = tid / p.gpc;
    const int S = p.D * p.H * p.W;
    const int c = cg * VEC;
    // coefficients of dx = k0*dn - k1 - k2*xhat
    float k0[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { k0[j] = 1.f; k1[j] = 0.f; k2[j] = 0.f; }
    if (p.norm) {
        // the striped sums of the statistics pass (or of the data-gradient epilogue that carried it) are added up HERE, by every
        // thread for its own channels -- a fold at the end of the statistics kernel (last-workgroup ticket: fence, returning
        // atomic, acquire, fold loop) or a fold launch cost more than these 8 x 2 vector loads per thread
        const int nc = n * p.C + c;
        const size_t total = (size_t)p.N * p.C * 2;
        float gm[VEC], rs[VEC], r[2 * VEC];
        ldvec(p.gamma + c, gm, VEC); ldvec(p.rstd + anb_nx(p, n) * p.C + c, rs, VEC);
#pragma unroll
        for (int j = 0; j < 2 * VEC; ++j) r[j] = 0.f;
#pragma unroll
        for (int t = 0; t < VG_STRIPES; ++t) {
            float q[2 * VEC];
            ldvec(p.red + t * total + (size_t)nc * 2, q, VEC);
            if (VEC == 8) ldvec(p.red + t * total + (size_t)nc * 2 + 8, q + 8, VEC); else q[1] = p.red[t * total + (size_t)nc * 2 + 1];
#pragma unroll
            for (int j = 0; j < 2 * VEC; ++j) r[j] += q[j];
        }
        if (p.dgamma && bx == 0 && vl == 0 && (p.pgrad_n <= 0 || n < p.pgrad_n)) {          // sum(dn) is d/d beta, sum(dn * xhat) is d/d gamma (summed over samples)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { atomicAdd(&p.dbeta[c + j], r[2 * j]); atomicAdd(&p.dgamma[c + j], r[2 * j + 1]); }
        }
        const float cnt = (float)S;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float gr = gm[j] * rs[j]; k0[j] = gr; k1[j] = gr * r[2 * j] / cnt; k2[j] = gr * r[2 * j + 1] / cnt; }
    }
    ChanK<VEC> ck;
    load_chank<VEC>(p, n, c, ck);
    anb_walk_from<T, VEC, 4>(p, n, c, bx * p.vpb + vl, gx * p.vpb, ck, [&](int v, const float* dn, const float* xh) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = k0[j] * dn[j] - k1[j] - k2[j] * xh[j];
        const size_t oidx = ((size_t)n * S + v) * p.dx_cstride + p.dx_coff + c;
        if (p.dx_f32) {
            float* q = (float*)p.dx + oidx;
#pragma unroll
            for (int j = 0; j < VEC; ++j) q[j] = p.accumulate ? q[j] + o[j] : o[j];
        } else if (VEC == 8) {
            bf16_t* q = (bf16_t*)p.dx + oidx;
            if (p.accumulate) {
                float old[8]; load8<bf16_t>(q, old);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += old[j];
            }
            store8<bf16_t>(q, o);
        } else {
            bf16_t* q = (bf16_t*)p.dx + oidx;
            q[0] = f2bf(p.accumulate ? bf2f(q[0]) + o[0] : o[0]);
        }
    });
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void actnorm_apply_kernel(const ANB p) { actnorm_apply_body<T, VEC>(p, (int)blockIdx.x, (int)gridDim.x); }
// two independent apply passes of one batch in ONE launch (a residual block's shortcut norm and the norm in front of its second convolution:
// their statistics are complete at the same point of the backward sweep and nothing orders their outputs): workgroups [0, g1) serve p1
template <typename T, int VEC>
__global__ __launch_bounds__(256) void actnorm_apply2_kernel(const ANB p1, const ANB p2, const int g1) {
    if ((int)blockIdx.x < g1) actnorm_apply_body<T, VEC>(p1, (int)blockIdx.x, g1);
    else actnorm_apply_body<T, VEC>(p2, (int)blockIdx.x - g1, (int)gridDim.x - g1);
}

static int fill_anb(const vg_actnorm_bwd_desc* d, ANB& p, bool apply) {
    if (!d || !d->g || d->N < 1 || d->C < 1) return VG_EINVAL;
    if (d->C != 1 && (d->C % 8)) return VG_EINVAL;
    if (d->x_f32 && d->C != 1 && !d->f32) return VG_EINVAL;
    if ((d->act != VG_ACT_NONE || d->norm) && !d->x) return VG_EINVAL;
    if (d->norm && (!d->mean || !d->rstd || !d->red)) return VG_EINVAL;
    if (apply && (!d->dx || (d->norm && !d->gamma))) return VG_EINVAL;
    if (d->g_padded && (d->D < 2 || d->H < 2 || d->W < 2)) return VG_EINVAL;
    if (d->x1 && (d->C == 1 || (d->c_x0 % 8) || d->c_x0 < 8 || d->c_x0 >= d->C)) return VG_EINVAL;
    p.g = d->g; p.g_padded = d->g_padded; p.x = d->x; p.x_f32 = (d->x_f32 || d->f32) ? 1 : 0;
    p.x1 = d->x1; p.c_x0 = d->c_x0; p.x0_shift = d->x1 ? (d->x0_shift ? 1 : 0) : 0;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.C = d->C;
    p.scale = d->scale; p.shift = d->shift; p.mult = d->mult; p.act = d->act; p.norm = d->norm;
    p.gamma = d->gamma; p.mean = d->mean; p.rstd = d->rstd; p.red = d->red;
    p.ticket = d->ticket; p.dgamma = (d->dgamma && d->dbeta) ? d->dgamma : nullptr; p.dbeta = d->dbeta;
    p.dx = d->dx; p.dx_f32 = (d->dx_f32 || d->f32) ? 1 : 0; p.accumulate = d->accumulate;
    p.dx_cstride = d->dx_cstride > 0 ? d->dx_cstride : d->C; p.dx_coff = d->dx_coff;
    p.alias_n0 = d->alias_n0; p.alias_sh = d->alias_shift; p.pgrad_n = d->pgrad_n;
    if (d->alias_n0 < 0 || (d->alias_n0 > 0 && (d->alias_shift < 1 || d->alias_shift > d->alias_n0 || d->alias_n0 >= d->N)) || d->pgrad_n < 0) return VG_EINVAL;
    p.gpc = d->C == 1 ? 1 : d->C / 8;
    if (p.gpc > 256 || d->C > 512) return VG_EINVAL;
    p.vpb = 256 / p.gpc;
    if (!p.dx_f32 && d->C != 1 && ((p.dx_cstride % 8) || (p.dx_coff % 8))) return VG_EINVAL;
    return VG_OK;
}
static dim3 anb_grid(const ANB& p, bool stats = false) {
    const int S = p.D * p.H * p.W;
    int bx = (S + p.vpb - 1) / p.vpb;
    const int cap_total = vg_tune("ANB_GRID_CAP", 767), cap_stats = vg_tune("ANB_STATS_CAP", 511);
    // whole launch resident at once (the 8-channel kernels hold ~150 VGPRs: 3 blocks per CU = 768 slots; re-swept at the end of
    // round 1: 767 / 511 beat the earlier 2047 / 1023 by ~0.5 % of the step): one block more than that runs alone afterwards.
    // The caps are odd on purpose: the UB voxels a thread has in flight are gridDim.x*vpb voxels apart, and with a
    // power-of-two grid on a power-of-two volume that is a power-of-two byte stride -- every in-flight load of the chip
    // then falls on the same HBM channels (measured at 128^3 x 16: stats 0.061 -> 0.046 ms, apply 0.053 -> 0.043 ms)
    int cap = (stats ? cap_stats : cap_total) / (p.N > 0 ? p.N : 1);
    if (cap < 1) cap = 1;
    if (bx > cap) bx = cap;
    return dim3(bx, p.N);
}

extern "C" int vg_actnorm_bwd_stats(const vg_actnorm_bwd_desc* d, vg_stream_t stream) {
    vg_begin();
    ANB p; int rc = fill_anb(d, p, false);
    if (rc != VG_OK) return rc;
    if (!p.red) return VG_EINVAL;
    if (d->f32) {
        if (p.C == 1) hipLaunchKernelGGL((actnorm_stats_kernel<float, 1>), anb_grid(p, true), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((actnorm_stats_kernel<float, 8>), anb_grid(p, true), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (p.C == 1) hipLaunchKernelGGL((actnorm_stats_kernel<bf16_t, 1>), anb_grid(p, true), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((actnorm_stats_kernel<bf16_t, 8>), anb_grid(p, true), dim3(256), 0, (hipStream_t)stream, p);
    }
    return vg_check_launch();          // red stays striped: the apply pass adds the stripes up (and the gamma / beta gradients)
}
extern "C" int vg_actnorm_bwd_apply(const vg_actnorm_bwd_desc* d, vg_stream_t stream) {
    vg_begin();
    ANB p; int rc = fill_anb(d, p, true);
    if (rc != VG_OK) return rc;
    if (d->f32) {
        if (p.C == 1) hipLaunchKernelGGL((actnorm_apply_kernel<float, 1>), anb_grid(p), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((actnorm_apply_kernel<float, 8>), anb_grid(p), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (p.C == 1) hipLaunchKernelGGL((actnorm_apply_kernel<bf16_t, 1>), anb_grid(p), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((actnorm_apply_kernel<bf16_t, 8>), anb_grid(p), dim3(256), 0, (hipStream_t)stream, p);
    }
    return vg_check_launch();
}

extern "C" int vg_actnorm_bwd_apply2(const vg_actnorm_bwd_desc* d1, const vg_actnorm_bwd_desc* d2, vg_stream_t stream) {
    vg_begin();
    ANB p1, p2;
    int rc = fill_anb(d1, p1, true); if (rc != VG_OK) return rc;
    rc = fill_anb(d2, p2, true); if (rc != VG_OK) return rc;
    // one launch serves both only when they share the kernel instance and the sample axis; else two launches, in order
    if (!vg_tune("ANB_APPLY2", 1) || d1->N != d2->N || (d1->f32 != 0) != (d2->f32 != 0) || d1->C == 1 || d2->C == 1) {
        rc = vg_actnorm_bwd_apply(d1, stream);
        return rc != VG_OK ? rc : vg_actnorm_bwd_apply(d2, stream);
    }
    const dim3 g1 = anb_grid(p1), g2 = anb_grid(p2);
    const dim3 grid(g1.x + g2.x, g1.y);
    if (d1->f32) hipLaunchKernelGGL((actnorm_apply2_kernel<float, 8>), grid, dim3(256), 0, (hipStream_t)stream, p1, p2, (int)g1.x);
    else hipLaunchKernelGGL((actnorm_apply2_kernel<bf16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, p1, p2, (int)g1.x);
    return vg_check_launch();
}

// statistics (when d->norm) + apply in one call.  (A single-launch form for small grids -- one workgroup per (sample, 8-channel
// group) plane, no atomics or ticket -- was measured and dropped: at 16^3 x 128 it took 50 us against 23 us for the two launches,
// because only C/8 = 16 workgroups carry the whole walk incl. its reflect-pad tails; at 8^3 x 256 and 16^3 x 512 it tied.)
extern "C" int vg_actnorm_bwd(const vg_actnorm_bwd_desc* d, vg_stream_t stream) {
    if (d && d->norm) { const int rc = vg_actnorm_bwd_stats(d, stream); if (rc != VG_OK) return rc; }
    return vg_actnorm_bwd_apply(d, stream);
}

__global__ void in_param_grads_kernel(const float* red, int N, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N * VG_STRIPES; ++n) { a += red[((size_t)n * C + c) * 2]; b += red[((size_t)n * C + c) * 2 + 1]; }
    dbeta[c] += a; dgamma[c] += b;
}
// Kernel gradient (+ the norm's gamma / beta gradients) of the stem's shortcut: a 1x1x1 convolution of the SINGLE-channel volume x in
// front of an InstanceNorm.  Its output w*x + b normalises to xhat[c] = w[c]*rs[c]*(x - mean x), rs[c] = (w[c]^2 var x + eps)^-1/2, so
// everything the backward needs is two per-channel moments of the output gradient g against the volume itself,
//     R0[n][c] = sum_v g[n][v][c],       T[n][c] = sum_v g[n][v][c] * x[n][v],       A = T - mean(x) * R0
//     dL/dw[c] += sum_n eps * gamma[c] * rs^3 * A,    dL/dgamma[c] += sum_n w[c] * rs * A,    dL/dbeta[c] += sum_n R0,    dL/db = 0.
// Round 5 took xhat from the STORED shortcut tensor instead; for a channel with a small kernel weight that tensor is b + w*x rounded to
// 16 bits -- a handful of levels -- and the closed form, which has 1/w in it, amplified the quantisation to 6-13 % of the gradient's norm
// (tools/r06_stem_probe.py: float64 sums over the stored tensors reproduce the old kernel to 1e-5, so it was never the summation order).
// x is exact (fp32), 8x fewer bytes than the stored tensor, and the result is deterministic: per-workgroup partial sums (fp32 inside a wave,
// double from there on), added up in a fixed order by the workgroup that draws the last ticket.
template <typename T>
__global__ __launch_bounds__(256) void stem_short_bwd_kernel(const T* g, const float* x, int N, int64_t S, int C, const float* w, const float* gamma,
                                                             float eps, int round16, float* dw, float* dgamma, float* dbeta, double* part,
                                                             unsigned* ticket) {
    __shared__ float sm[4][8][18];
    __shared__ double fin[256][4];
    __shared__ int last;
    const int gpc = C >> 3, vpb = 256 / gpc, tid = threadIdx.x, cg = tid % gpc, vl = tid / gpc, n = blockIdx.y, G = gridDim.x, NV = 2 * C + 2;
    float r0[8], t[8], sx = 0.f, sxx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { r0[j] = 0.f; t[j] = 0.f; }
    const T* gn = g + (size_t)n * S * C + cg * 8;
    const float* xn = x + (size_t)n * S;
    // eight voxels per trip, their sixteen loads issued before the first use (one load in flight per thread kept HBM at a third of its rate)
    constexpr int UN = 8;
    const int64_t vstep = (int64_t)G * vpb;
    int64_t v = (int64_t)blockIdx.x * vpb + vl;
    for (; v + (UN - 1) * vstep < S; v += UN * vstep) {
        Raw8<T> raw[UN]; float xv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { raw_load(raw[u], gn + (v + u * vstep) * C); xv[u] = ld_global(xn + v + u * vstep); }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float gv[8];
            raw_unpack(raw[u], gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { r0[j] += gv[j]; t[j] = fmaf(gv[j], xv[u], t[j]); }
            sx += xv[u]; sxx = fmaf(xv[u], xv[u], sxx);
        }
    }
    for (; v < S; v += vstep) {
        float gv[8];
        load8<T>(gn + v * C, gv);
        const float xv = xn[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) { r0[j] += gv[j]; t[j] = fmaf(gv[j], xv, t[j]); }
        sx += xv; sxx = fmaf(xv, xv, sxx);
    }
    // Inside a wave the partial sums are added in fp32 (a wave's share is ~1 000 terms: far from the cancellation of the whole volume, which
    // is what needs care -- fp32 here costs ~5e-7 of the result); across waves and workgroups in double, in a fixed order.
    float d[18];
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = r0[j]; d[8 + j] = t[j]; }
    d[16] = sx; d[17] = sxx;
    for (int o = 32; o >= gpc; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 18; ++k) d[k] += __shfl_xor(d[k], o);
    }
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < gpc) {
#pragma unroll
        for (int k = 0; k < 18; ++k) sm[wv][lane][k] = d[k];
    }
    __syncthreads();
    if (tid < NV) {
        int cgi, k;
        if (tid < C) { cgi = tid >> 3; k = tid & 7; }
        else if (tid < 2 * C) { cgi = (tid - C) >> 3; k = 8 + ((tid - C) & 7); }
        else { cgi = 0; k = 16 + tid - 2 * C; }
        // agent-scope store: the partial goes to the level the XCDs share (no __threadfence(): its L2 write-back / invalidate per workgroup cost
        // more than the kernel's whole stream -- 50 -> 174 us from 254 to 2 044 workgroups)
        __hip_atomic_store(part + ((size_t)n * G + blockIdx.x) * NV + tid,
                           (((double)sm[0][cgi][k] + (double)sm[1][cgi][k]) + (double)sm[2][cgi][k]) + (double)sm[3][cgi][k], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // the exchange of vg_fin_tail (vg_common.h): stores acknowledged (vmcnt), workgroup barrier, one relaxed agent-scope ticket; the
    // workgroup that draws the last one reads every partial with agent-scope loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(N * G - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    auto ald = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const int P = 256 / C, c = tid % C, pk = tid / C;
    double adw = 0., adg = 0., adb = 0.;
    for (int nn = 0; nn < N; ++nn) {
        double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
        const int g0 = (int)((int64_t)pk * G / P), g1 = (int)((int64_t)(pk + 1) * G / P);
        const double* base = part + (size_t)nn * G * NV;
        int gi = g0;
        for (; gi + 8 <= g1; gi += 8) {
            double q0[8], q1[8], q2[8], q3[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const double* q = base + (size_t)(gi + u) * NV; q0[u] = ald(q + c); q1[u] = ald(q + C + c); q2[u] = ald(q + 2 * C); q3[u] = ald(q + 2 * C + 1); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 += q0[u]; a1 += q1[u]; a2 += q2[u]; a3 += q3[u]; }
        }
        for (; gi < g1; ++gi) { const double* q = base + (size_t)gi * NV; a0 += ald(q + c); a1 += ald(q + C + c); a2 += ald(q + 2 * C); a3 += ald(q + 2 * C + 1); }
        fin[tid][0] = a0; fin[tid][1] = a1; fin[tid][2] = a2; fin[tid][3] = a3;
        __syncthreads();
        if (tid < C) {
            double R0 = 0., Tt = 0., Sx = 0., Sxx = 0.;
            for (int k = 0; k < P; ++k) { const double* f = fin[k * C + c]; R0 += f[0]; Tt += f[1]; Sx += f[2]; Sxx += f[3]; }
            const double mu = Sx / (double)S;
            double var = Sxx / (double)S - mu * mu;
            var = var < 0. ? 0. : var;
            const double wc = round16 ? (double)bf2f(f2bf(w[c])) : (double)w[c];
            const double rs = 1. / sqrt(wc * wc * var + (double)eps);
            const double A = Tt - mu * R0;
            adw += (double)eps * (double)gamma[c] * rs * rs * rs * A;
            adg += wc * rs * A;
            adb += R0;
        }
        __syncthreads();
    }
    if (tid < C) {
        dw[c] += (float)adw;
        if (dgamma) { dgamma[c] += (float)adg; dbeta[c] += (float)adb; }
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // left at zero for the next launch that is handed this word
}
// The stem's shortcut in the forward pass, never materialised (include/vangan_hip.h: vg_stem_short_fwd): mean and variance of the volume ->
// the per-(sample, channel) affine  scale * x + shift  that the block's second convolution adds in its epilogue (vg_conv_desc::res_c1).
__global__ __launch_bounds__(256) void stem_short_fwd_kernel(const float* x, int N, int64_t S, int C, const float* w, const float* gamma,
                                                             const float* beta, float eps, int round16, float* scale, float* shift, double* part,
                                                             unsigned* ticket) {
    __shared__ float sm[4][2];
    __shared__ double fin[256][2];
    __shared__ double mom[2];
    __shared__ int last;
    const int tid = threadIdx.x, n = blockIdx.y, G = gridDim.x;
    const float* xn = x + (size_t)n * S;
    float sx = 0.f, sxx = 0.f;
    const int64_t S4 = (((uintptr_t)xn & 15) == 0) ? S >> 2 : 0;
    const int64_t step = (int64_t)G * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + tid;
    for (; i + 3 * step < S4; i += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(xn + 4 * (i + u * step));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sx += v[u][j]; sxx = fmaf(v[u][j], v[u][j], sxx); }
    }
    for (; i < S4; i += step) {
        const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(xn + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sx += v[j]; sxx = fmaf(v[j], v[j], sxx); }
    }
    for (int64_t k = 4 * S4 + (int64_t)blockIdx.x * 256 + tid; k < S; k += step) { const float v = xn[k]; sx += v; sxx = fmaf(v, v, sxx); }
    sx = wave_sum(sx); sxx = wave_sum(sxx);
    if ((tid & 63) == 0) { sm[tid >> 6][0] = sx; sm[tid >> 6][1] = sxx; }
    __syncthreads();
    if (tid < 2)
        __hip_atomic_store(part + ((size_t)n * G + blockIdx.x) * 2 + tid, (((double)sm[0][tid] + (double)sm[1][tid]) + (double)sm[2][tid]) + (double)sm[3][tid],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(N * G - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    for (int nn = 0; nn < N; ++nn) {
        double a0 = 0., a1 = 0.;
        for (int gi = tid; gi < G; gi += 256) {
            a0 += __hip_atomic_load(part + ((size_t)nn * G + gi) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a1 += __hip_atomic_load(part + ((size_t)nn * G + gi) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        fin[tid][0] = a0; fin[tid][1] = a1;
        __syncthreads();
        if (tid < 2) { double a = 0.; for (int k = 0; k < 256; ++k) a += fin[k][tid]; mom[tid] = a; }       // fixed order
        __syncthreads();
        const double mu = mom[0] / (double)S;
        double var = mom[1] / (double)S - mu * mu;
        var = var < 0. ? 0. : var;
        for (int c = tid; c < C; c += 256) {
            const double wc = round16 ? (double)bf2f(f2bf(w[c])) : (double)w[c];
            const double sc = (double)gamma[c] * wc / sqrt(wc * wc * var + (double)eps);
            scale[nn * C + c] = (float)sc;
            shift[nn * C + c] = (float)((double)beta[c] - sc * mu);
        }
        __syncthreads();
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int vg_stem_short_fwd_workgroups(int N, int64_t S) {
    if (N < 1 || S < 1) return VG_EINVAL;
    int64_t g = S / 32768;                  // 32 float4 per thread
    if (g < 1) g = 1;
    if (g > 128) g = 128;
    return (int)g;
}
extern "C" int vg_stem_short_fwd(const float* x, int N, int64_t S, int C, const float* w, const float* gamma, const float* beta, float eps, int round16,
                                 float* scale, float* shift, double* part, int G, unsigned* ticket, vg_stream_t stream) {
    vg_begin();
    if (!x || !w || !gamma || !beta || !scale || !shift || !part || !ticket || C < 1) return VG_EINVAL;
    if (G < 1 || G != vg_stem_short_fwd_workgroups(N, S)) return VG_EINVAL;
    hipLaunchKernelGGL(stem_short_fwd_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, x, N, S, C, w, gamma, beta, eps, round16, scale, shift, part, ticket);
    return vg_check_launch();
}
extern "C" int vg_stem_short_bwd_workgroups(int N, int64_t S, int C) {
    if (N < 1 || S < 1 || C < 8 || C > 64 || (256 % C)) return VG_EINVAL;
    const int vpb = 256 / (C >> 3);
    int64_t gx = (S + vpb - 1) / vpb;
    // one workgroup per CU, an ODD count (the voxels a thread has in flight are G*vpb apart: fill_anb's remark on HBM channels); more
    // workgroups only lengthen the last one's fixed-order sum (128^3 x 16, two samples: 36 us at 254 workgroups = 4.2 TB/s, 40 at 508, 50 at 1 020)
    int cap = vg_tune("STEM_BWD_GRID", 255) / N;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    return (int)gx;
}
extern "C" int vg_stem_short_bwd(const void* g, int g_f32, const float* x, int N, int64_t S, int C, const float* w, const float* gamma, float eps,
                                 int round16, float* dw, float* dgamma, float* dbeta, double* part, int G, unsigned* ticket, vg_stream_t stream) {
    vg_begin();
    if (!g || !x || !w || !gamma || !dw || !part || !ticket || (dgamma && !dbeta)) return VG_EINVAL;
    if (G != vg_stem_short_bwd_workgroups(N, S, C) || G < 1) return VG_EINVAL;
    if (g_f32) hipLaunchKernelGGL((stem_short_bwd_kernel<float>), dim3(G, N), dim3(256), 0, (hipStream_t)stream, (const float*)g, x, N, S, C, w, gamma,
                                  eps, round16, dw, dgamma, dbeta, part, ticket);
    else hipLaunchKernelGGL((stem_short_bwd_kernel<bf16_t>), dim3(G, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, x, N, S, C, w, gamma,
                            eps, round16, dw, dgamma, dbeta, part, ticket);
    return vg_check_launch();
}
extern "C" int vg_in_param_grads(const float* red, int N, int C, float* dgamma, float* dbeta, vg_stream_t stream) {
    vg_begin();
    if (!red || !dgamma || !dbeta || N < 1 || C < 1) return VG_EINVAL;
    hipLaunchKernelGGL(in_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, red, N, C, dgamma, dbeta);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// backward of UpSampling3D(2) + concatenate (resunet_model.py:175-181)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void concat_bwd_kernel(const T* g, int N, int D, int H, int W, int Cu, int Cs, T* dlow, T* dskip, int acc) {
    const int C = Cu + Cs;
    const int gu = Cu / 8, gs = Cs / 8;
    const size_t nlow = (size_t)N * (D / 2) * (H / 2) * (W / 2) * gu;
    const size_t nskip = (size_t)N * D * H * W * gs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlow + nskip; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nlow) {
            const int cg = (int)(i % gu); size_t v = i / gu;
            const int w = (int)(v % (W / 2)); v /= (W / 2);
            const int h = (int)(v % (H / 2)); v /= (H / 2);
            const int d = (int)(v % (D / 2)); const int n = (int)(v / (D / 2));
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 2; ++e) {
                const size_t idx = ((((size_t)n * D + 2 * d + a) * H + 2 * h + b) * W + 2 * w + e) * C + cg * 8;
                float r[8]; load8<T>(g + idx, r);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += r[j];
            }
            if (acc & 1) {
                float old[8]; load8<T>(dlow + i * 8, old);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += old[j];
            }
            store8<T>(dlow + i * 8, s);
        } else {
            const size_t k = i - nlow;
            const int cg = (int)(k % gs); const size_t v = k / gs;
            float a[8];
            load8<T>(g + v * C + Cu + cg * 8, a);
            if (acc & 2) {
                float old[8]; load8<T>(dskip + k * 8, old);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += old[j];
            }
            store8<T>(dskip + k * 8, a);
        }
    }
}
extern "C" int vg_concat_bwd(const void* g, int N, int D, int H, int W, int Cu, int Cs, void* dlow, void* dskip,
                             int f32, int accumulate, vg_stream_t stream) {
    vg_begin();
    // Cs == 0 (dskip ignored): the backward of a bare UpSampling3D -- the 2x2x2 sum-pool (ResNet generator, generator.py:58-66)
    if (!g || !dlow || (Cs && !dskip) || (Cu % 8) || (Cs % 8) || Cu < 8 || (Cs && Cs < 8) || ((D | H | W) & 1)) return VG_EINVAL;
    const size_t total = (size_t)N * (D / 2) * (H / 2) * (W / 2) * (Cu / 8) + (size_t)N * D * H * W * (Cs / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 8191) blocks = 8191;
    if (f32) hipLaunchKernelGGL(concat_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)g, N, D, H, W, Cu,
                                Cs, (float*)dlow, (float*)dskip, accumulate);
    else hipLaunchKernelGGL(concat_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, N, D, H, W, Cu,
                            Cs, (bf16_t*)dlow, (bf16_t*)dskip, accumulate);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// small elementwise helpers
// ------------------------------------------------------------------------------------------------
__global__ void tanh_bwd_kernel(const float* dy, const float* y, float* dpre, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dpre[i] = dy[i] * (1.f - y[i] * y[i]);
}
__global__ void axpby_kernel(const float* a, float alpha, const float* b, float beta, int64_t n, float* y, int acc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = alpha * a[i] + (b ? beta * b[i] : 0.f);
        y[i] = acc ? y[i] + v : v;
    }
}
__global__ void f32_to_bf16_kernel(const float* x, bf16_t* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* x, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}
static inline int ew_blocks(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// db[c] += sum over rows of dy[row][c] (bias gradient of a layer whose output gradient is dy: the k2 s2 Conv3DTranspose).  A block
// walks a strided set of rows with C-contiguous lanes, sums in registers, then per channel through LDS; one atomic per (block, channel).
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dy, int64_t rows, int C, float* db) {
    __shared__ float sm[256];
    const int lanes_per_row = C < 256 ? C : 256;                   // C <= 256: 256 / C rows in flight per block iteration
    const int rpb = 256 / lanes_per_row;
    const int c = threadIdx.x % lanes_per_row, r0 = threadIdx.x / lanes_per_row;
    for (int cb = 0; cb < C; cb += lanes_per_row) {
        float a = 0.f;
        if (r0 < rpb && cb + c < C)
            for (int64_t r = (int64_t)blockIdx.x * rpb + r0; r < rows; r += (int64_t)gridDim.x * rpb) a += ld1<T>(dy + r * C + cb + c);
        sm[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < lanes_per_row && cb + threadIdx.x < C) {
            float t = 0.f;
            for (int k = 0; k < rpb; ++k) t += sm[k * lanes_per_row + threadIdx.x];
            atomicAdd(&db[cb + threadIdx.x], t);
        }
        __syncthreads();
    }
}
extern "C" int vg_bias_grad(const void* dy, int f32, int64_t rows, int C, float* db, vg_stream_t stream) {
    vg_begin();
    if (!dy || !db || rows < 1 || C < 1) return VG_EINVAL;
    const int rpb = 256 / (C < 256 ? C : 256);
    int64_t b = (rows + rpb - 1) / rpb; if (b > 1023) b = 1023;
    if (f32) hipLaunchKernelGGL(bias_grad_kernel<float>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, (const float*)dy, rows, C, db);
    else hipLaunchKernelGGL(bias_grad_kernel<bf16_t>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, rows, C, db);
    return vg_check_launch();
}

extern "C" int vg_tanh_bwd(const float* dy, const float* y, float* dpre, int64_t n, vg_stream_t stream) {
    vg_begin();
    if (!dy || !y || !dpre || n < 0) return VG_EINVAL;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dpre, n);
    return vg_check_launch();
}
extern "C" int vg_axpby(const float* a, float alpha, const float* b, float beta, int64_t n, float* y, int accumulate,
                        vg_stream_t stream) {
    vg_begin();
    if (!a || !y || n < 0) return VG_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta, n, y, accumulate);
    return vg_check_launch();
}
extern "C" int vg_f32_to_bf16(const float* x, void* y, int64_t n, vg_stream_t stream) {
    vg_begin();
    if (!x || !y || n < 0) return VG_EINVAL;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    return vg_check_launch();
}
extern "C" int vg_bf16_to_f32(const void* x, float* y, int64_t n, vg_stream_t stream) {
    vg_begin();
    if (!x || !y || n < 0) return VG_EINVAL;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
    return vg_check_launch();
}

// out = act_a(a * sa + ha) + (b * sb + hb): the Add of the ResNet generator's residual block (building_blocks.py:68-123:
// layers.add([input_tensor, InstanceNorm(conv2)])), both operands read with their pending on-read affine -- the block input may
// still be the un-normalised output of the convolution in front of it.  8 channels per thread.
template <typename T>
__global__ __launch_bounds__(256) void affine_add_kernel(const T* __restrict__ a, const float* __restrict__ sa, const float* __restrict__ ha, int act_a,
                                                         const T* __restrict__ b, const float* __restrict__ sb, const float* __restrict__ hb,
                                                         int64_t S, int C, int64_t units, T* __restrict__ out) {
    const int cg = C >> 3;
    const float slope = act_a == VG_ACT_RELU ? 0.f : (act_a == VG_ACT_LRELU ? VG_LRELU : 1.f);
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int c = (int)(u % cg) * 8;
        const int64_t v = u / cg;
        const int n = (int)(v / S);
        float x[8], y[8];
        load8<T>(a + v * C + c, x); load8<T>(b + v * C + c, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float p = sa ? x[e] * sa[n * C + c + e] + ha[n * C + c + e] : x[e];
            p = fmaxf(p, p * slope);
            const float q = sb ? y[e] * sb[n * C + c + e] + hb[n * C + c + e] : y[e];
            x[e] = p + q;
        }
        store8<T>(out + v * C + c, x);
    }
}
extern "C" int vg_affine_add(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b, const float* b_scale,
                             const float* b_shift, int N, int64_t S, int C, void* out, int f32, vg_stream_t stream) {
    vg_begin();
    if (!a || !b || !out || N < 1 || S < 1 || C < 8 || (C % 8) || (a_scale && !a_shift) || (b_scale && !b_shift)) return VG_EINVAL;
    const int64_t units = (int64_t)N * S * (C / 8);
    if (f32) hipLaunchKernelGGL(affine_add_kernel<float>, dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, (const float*)a, a_scale, a_shift,
                                a_act, (const float*)b, b_scale, b_shift, S, C, units, (float*)out);
    else hipLaunchKernelGGL(affine_add_kernel<bf16_t>, dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, a_scale, a_shift,
                            a_act, (const bf16_t*)b, b_scale, b_shift, S, C, units, (bf16_t*)out);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG (Philox-4x32-10) for GaussianNoise / SpatialDropout3D
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void randn_bf16_kernel(bf16_t* out, int64_t n, float std, uint64_t seed, uint64_t offset, const float* std_dev, const uint64_t* off_dev) {
    if (std_dev) std = *std_dev;                  // per-step scalars from device memory: a captured graph of the step stays replayable
    if (off_dev) offset += *off_dev;
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x5eedu, 0u};
        philox4(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const float r0 = sqrtf(-2.f * __logf(u01(c[0]))), r1 = sqrtf(-2.f * __logf(u01(c[2])));
        const float t0 = 6.28318530718f * u01(c[1]), t1 = 6.28318530718f * u01(c[3]);
        const float z[4] = {r0 * __cosf(t0), r0 * __sinf(t0), r1 * __cosf(t1), r1 * __sinf(t1)};
        for (int j = 0; j < 4; ++j) if (q * 4 + j < n) out[q * 4 + j] = f2bf(std * z[j]);
    }
}
__global__ void dropout_mask_kernel(float* out, int64_t n, float rate, uint64_t seed, uint64_t offset, const uint64_t* off_dev) {
    if (off_dev) offset += *off_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)i;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0xd60bu, 0u};
        philox4(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        out[i] = u01(c[0]) < rate ? 0.f : 1.f / (1.f - rate);
    }
}
extern "C" int vg_randn_bf16(void* out, int64_t n, float std, uint64_t seed, uint64_t offset, vg_stream_t stream) {
    vg_begin();
    if (!out || n < 0) return VG_EINVAL;
    hipLaunchKernelGGL(randn_bf16_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, n, std, seed, offset,
                       (const float*)nullptr, (const uint64_t*)nullptr);
    return vg_check_launch();
}
extern "C" int vg_randn_bf16_dev(void* out, int64_t n, const float* std_dev, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_add,
                                 vg_stream_t stream) {
    vg_begin();
    if (!out || n < 0 || !std_dev || !offset_dev) return VG_EINVAL;
    hipLaunchKernelGGL(randn_bf16_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, n, 0.f, seed, offset_add,
                       std_dev, offset_dev);
    return vg_check_launch();
}
extern "C" int vg_dropout_mask(float* out, int64_t n, float rate, uint64_t seed, uint64_t offset, vg_stream_t stream) {
    vg_begin();
    if (!out || n < 0 || rate < 0.f || rate >= 1.f) return VG_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, n, rate, seed, offset, (const uint64_t*)nullptr);
    return vg_check_launch();
}
extern "C" int vg_dropout_mask_dev(float* out, int64_t n, float rate, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_add,
                                   vg_stream_t stream) {
    vg_begin();
    if (!out || n < 0 || rate < 0.f || rate >= 1.f || !offset_dev) return VG_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, n, rate, seed, offset_add, offset_dev);
    return vg_check_launch();
}

// The 32-byte per-step parameter block of a replayed / graph-launched train step, written by a kernel whose ARGUMENTS carry the values:
// they are bound when the launch is enqueued, so a host that runs ahead (sync=False) cannot overwrite step N's scalars with step N+1's
// before step N's consumers have read them (a pinned host mirror + asynchronous copy could: the copy reads the mirror when it EXECUTES).
__global__ void set_step_params_kernel(unsigned char* block, uint64_t offset, float std, float lr0, float lr1, float lr2, float lr3) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *(uint64_t*)block = offset;
        *(float*)(block + 8) = std;
        *(float*)(block + 12) = 0.f;
        float* lr = (float*)(block + 16);
        lr[0] = lr0; lr[1] = lr1; lr[2] = lr2; lr[3] = lr3;
    }
}
extern "C" int vg_set_step_params(void* block, uint64_t offset, float std, float lr0, float lr1, float lr2, float lr3, vg_stream_t stream) {
    vg_begin();
    if (!block || ((uintptr_t)block & 7)) return VG_EINVAL;
    hipLaunchKernelGGL(set_step_params_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned char*)block, offset, std, lr0, lr1, lr2, lr3);
    return vg_check_launch();
}

// Plumbing with an explicit stream (a recorded launch list replays these instead of torch's zero_() / copy_(), which go to torch's
// CURRENT stream): enqueue-only, caller-owned memory.
extern "C" int vg_memset_zero(void* p, int64_t nbytes, vg_stream_t stream) {
    vg_begin();
    if (!p || nbytes < 0) return VG_EINVAL;
    if (nbytes == 0) return VG_OK;
    return hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream) == hipSuccess ? VG_OK : VG_ELAUNCH;
}
extern "C" int vg_copy_bytes(void* dst, const void* src, int64_t nbytes, vg_stream_t stream) {
    vg_begin();
    if (!dst || !src || nbytes < 0) return VG_EINVAL;
    if (nbytes == 0) return VG_OK;
    return hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? VG_OK : VG_ELAUNCH;
}

extern "C" const char* vg_status_string(int code) {
    switch (code) {
        case VG_OK: return "ok";
        case VG_EINVAL: return "invalid argument or unsupported shape";
        case VG_ELDS: return "tile does not fit in LDS";
        case VG_ELAUNCH: return "HIP launch failure";
        default: return "unknown status";
    }
}
// ------------------------------------------------------------------------------------------------
// tuning registry and dry-run recorder (vg_common.h)
// ------------------------------------------------------------------------------------------------
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
namespace {
struct TuneEntry { char key[32]; int value; int state; };      // state: 0 unset, 1 from env/default (cached), 2 forced
TuneEntry g_tune[256];          // (64 overflowed in round 5: a full table silently turned vg_set_tuning into a no-op for new keys)
int g_ntune = 0;
std::mutex g_tune_mu;
thread_local char* t_dry_buf = nullptr;
thread_local int t_dry_len = 0;
TuneEntry* tune_find(const char* key, bool create) {
    for (int i = 0; i < g_ntune; ++i) if (!strcmp(g_tune[i].key, key)) return &g_tune[i];
    if (!create || g_ntune >= 256 || strlen(key) >= sizeof(g_tune[0].key)) return nullptr;
    TuneEntry* e = &g_tune[g_ntune++];
    strcpy(e->key, key); e->value = 0; e->state = 0;
    return e;
}
}
int vg_tune(const char* key, int dflt) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    TuneEntry* e = tune_find(key, true);
    if (!e) return dflt;
    if (e->state == 0) {
        char name[48]; snprintf(name, sizeof name, "VG_%s", key);
        const char* v = getenv(name);
        e->value = v ? atoi(v) : dflt; e->state = 1;
    }
    return e->value;
}
extern "C" int vg_set_tuning(const char* key, int value, int reset) {
    if (!key) return VG_EINVAL;
#ifndef VG_ABLATE
    if (!strcmp(key, "DEBUG")) return VG_EINVAL;        // phase ablations (silent no-store kernels) exist in diagnostic builds only
#endif
    std::lock_guard<std::mutex> lk(g_tune_mu);
    TuneEntry* e = tune_find(key, true);
    if (!e) return VG_EINVAL;
    if (reset) e->state = 0; else { e->value = value; e->state = 2; }
    return VG_OK;
}
bool vg_dry(const char* fmt, ...) {
    if (!t_dry_buf) return false;
    va_list ap; va_start(ap, fmt);
    const int used = (int)strlen(t_dry_buf);
    if (used < t_dry_len - 2) {
        if (used) { t_dry_buf[used] = ';'; t_dry_buf[used + 1] = 0; }
        const int u2 = (int)strlen(t_dry_buf);
        vsnprintf(t_dry_buf + u2, t_dry_len - u2, fmt, ap);
    }
    va_end(ap);
    return true;
}
void vg_dry_begin(char* buf, int n) { t_dry_buf = buf; t_dry_len = n; if (buf && n > 0) buf[0] = 0; }
void vg_dry_end() { t_dry_buf = nullptr; t_dry_len = 0; }
bool vg_dry_on() { return t_dry_buf != nullptr; }

extern "C" int vg_version(void) {
    vg_begin(); return 3; }
extern "C" int vg_storage16(void) {
#ifdef VG_FP16
    return 1;
#else
    return 0;
#endif
}
extern "C" int vg_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(vg_conv_desc);
        case 1: return (int)sizeof(vg_actnorm_bwd_desc);
        case 2: return (int)sizeof(vg_pack_item);
        case 3: return (int)sizeof(vg_fin_desc);
        default: return VG_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// Training data pipeline on resident volumes (dataset.py:205-251): crop + flips + rot90 as ONE gather.
// out[a][i][j][c] (a along X, i along Y = "height", j along Z = "width"):
//   rot90 k (counter-clockwise, tf.image.rot90): k=1: r[i][j] = f[j][P-1-i]; k=2: r[i][j] = f[P-1-i][Q-1-j];
//   k=3: r[i][j] = f[P-1-j][i]   (P x Q = rows x cols of f; odd k needs P == Q)
//   f = flip_up_down?(flip_left_right?(crop)): f[i][j] = crop[ud ? P-1-i : i][lr ? Q-1-j : j]
// ------------------------------------------------------------------------------------------------
__global__ void crop_augment_kernel(const float* __restrict__ vol, int Y, int Z, int C, int x0, int y0, int z0,
                                    int px, int py, int pz, int lr, int ud, int k, float* __restrict__ out) {
    const int64_t total = (int64_t)px * py * pz * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C); int64_t r = e / C;
        const int j = (int)(r % pz); r /= pz;
        const int i = (int)(r % py); const int a = (int)(r / py);
        int fi, fj;                                    // position in f (after the flips, before the rotation)
        if (k == 0) { fi = i; fj = j; }
        else if (k == 1) { fi = j; fj = pz - 1 - i; }
        else if (k == 2) { fi = py - 1 - i; fj = pz - 1 - j; }
        else { fi = py - 1 - j; fj = i; }
        const int ci = ud ? py - 1 - fi : fi, cj = lr ? pz - 1 - fj : fj;
        out[e] = vol[(((int64_t)(x0 + a) * Y + (y0 + ci)) * Z + (z0 + cj)) * C + c];
    }
}
extern "C" int vg_crop_augment(const float* vol, int X, int Y, int Z, int C, int x0, int y0, int z0, int px, int py, int pz,
                               int flip_lr, int flip_ud, int rot_k, float* out, vg_stream_t stream) {
    vg_begin();
    if (!vol || !out || C < 1 || px < 1 || py < 1 || pz < 1) return VG_EINVAL;
    if (x0 < 0 || y0 < 0 || z0 < 0 || x0 + px > X || y0 + py > Y || z0 + pz > Z) return VG_EINVAL;
    const int k = ((rot_k % 4) + 4) % 4;
    if ((k & 1) && py != pz) return VG_EINVAL;
    const int64_t total = (int64_t)px * py * pz * C;
    int64_t b = (total + 255) / 256; if (b > 8191) b = 8191;
    hipLaunchKernelGGL(crop_augment_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, vol, Y, Z, C, x0, y0, z0, px, py, pz,
                       flip_lr ? 1 : 0, flip_ud ? 1 : 0, k, out);
    return vg_check_launch();
}
__global__ void crop_max_kernel(const float* __restrict__ vol, int Y, int Z, int C, int x0, int y0, int z0, int px, int py, int pz,
                                float* out) {
    __shared__ float red[4];
    const int64_t total = (int64_t)px * py * pz * C;
    float m = -INFINITY;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C); int64_t r = e / C;
        const int j = (int)(r % pz); r /= pz;
        const int i = (int)(r % py); const int a = (int)(r / py);
        m = fmaxf(m, vol[(((int64_t)(x0 + a) * Y + (y0 + i)) * Z + (z0 + j)) * C + c]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // float max through the integer atomics (values may be negative): order-preserving key
        int* io = (int*)out;
        int old = *io, assumed;
        do { assumed = old; if (__int_as_float(assumed) >= m) break; old = atomicCAS(io, assumed, __float_as_int(m)); } while (old != assumed);
    }
}
extern "C" int vg_crop_max(const float* vol, int X, int Y, int Z, int C, int x0, int y0, int z0, int px, int py, int pz,
                           float* out, vg_stream_t stream) {
    vg_begin();
    if (!vol || !out || C < 1 || px < 1 || py < 1 || pz < 1) return VG_EINVAL;
    if (x0 < 0 || y0 < 0 || z0 < 0 || x0 + px > X || y0 + py > Y || z0 + pz > Z) return VG_EINVAL;
    if (hipMemsetD32Async((hipDeviceptr_t)out, (int)0xFF800000u, 1, (hipStream_t)stream) != hipSuccess) return VG_ELAUNCH;   // -inf
    const int64_t total = (int64_t)px * py * pz * C;
    int64_t b = (total + 1023) / 1024; if (b > 1023) b = 1023; if (b < 1) b = 1;
    hipLaunchKernelGGL(crop_max_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, vol, Y, Z, C, x0, y0, z0, px, py, pz, out);
    return vg_check_launch();
}
