// vg_loss.hip -- HBM-bound fp32 loss kernels on single-channel volumes [B][S]:
// min_max_norm_tf (utils.py:27-48) fwd/bwd, Keras BCE (loss_functions.py:185-190), MSE / LSGAN terms
// (loss_functions.py:56-68,273-274,306-308), 3-D SSIM (loss_functions.py:86-117) fwd/bwd, and the clDice
// soft skeleton (clDice_func.py:8-149) fwd/bwd.
#include "vg_common.h"

static inline int lblocks(int64_t n, int per = 256) { int64_t b = (n + per - 1) / per; return (int)(b > 4095 ? 4095 : (b < 1 ? 1 : b)); }   // odd cap: no power-of-two grid stride (HBM channel aliasing)

__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
    return r;
}

// ------------------------------------------------------------------------------------------------
// min-max normalisation
// ------------------------------------------------------------------------------------------------
__global__ void mm_init_kernel(float* mm4, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { mm4[b * 4] = INFINITY; mm4[b * 4 + 1] = -INFINITY; mm4[b * 4 + 2] = 0.f; mm4[b * 4 + 3] = 0.f; }
}
__device__ __forceinline__ void atomic_min_f(float* a, float v) {
    unsigned* ua = (unsigned*)a; unsigned old = *ua;
    while (v < __uint_as_float(old)) { const unsigned prev = atomicCAS(ua, old, __float_as_uint(v)); if (prev == old) break; old = prev; }
}
__device__ __forceinline__ void atomic_max_f(float* a, float v) {
    unsigned* ua = (unsigned*)a; unsigned old = *ua;
    while (v > __uint_as_float(old)) { const unsigned prev = atomicCAS(ua, old, __float_as_uint(v)); if (prev == old) break; old = prev; }
}
__global__ void mm_reduce_kernel(const float* x, int64_t S, float* mm4) {
    __shared__ float smn[4], smx[4];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * S;
    float mn = INFINITY, mx = -INFINITY;
    // 16-byte loads where the sample is aligned (a thread's walk is a chain of dependent-latency iterations: 4 instead of 16 at 128^3)
    const int64_t S4 = (((uintptr_t)xb & 15) == 0) ? S >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)xb)[i];
        mn = fminf(fminf(mn, fminf(v[0], v[1])), fminf(v[2], v[3])); mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    for (int64_t i = 4 * S4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = xb[i]; mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        atomic_min_f(&mm4[b * 4], mn); atomic_max_f(&mm4[b * 4 + 1], mx);
    }
}
__global__ void mm_count_kernel(const float* x, int64_t S, float* mm4) {
    __shared__ float sm[4];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * S;
    const float mn = mm4[b * 4], mx = mm4[b * 4 + 1];
    float c0 = 0.f, c1 = 0.f;
    const int64_t S4 = (((uintptr_t)xb & 15) == 0) ? S >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)xb)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0 += (v[j] == mn) ? 1.f : 0.f; c1 += (v[j] == mx) ? 1.f : 0.f; }
    }
    for (int64_t i = 4 * S4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = xb[i]; c0 += (v == mn) ? 1.f : 0.f; c1 += (v == mx) ? 1.f : 0.f;
    }
    c0 = block_sum(c0, sm); c1 = block_sum(c1, sm);
    if (threadIdx.x == 0) { if (c0 != 0.f) atomicAdd(&mm4[b * 4 + 2], c0); if (c1 != 0.f) atomicAdd(&mm4[b * 4 + 3], c1); }
}
extern "C" int vg_minmax(const float* x, int B, int64_t S, float* mm4, vg_stream_t stream) {
    vg_begin();
    if (!x || !mm4 || B < 1 || S < 1) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mm_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, mm4, B);
    const int bx = lblocks(S, 1024) > 512 ? 512 : lblocks(S, 1024);
    hipLaunchKernelGGL(mm_reduce_kernel, dim3(bx, B), dim3(256), 0, s, x, S, mm4);
    hipLaunchKernelGGL(mm_count_kernel, dim3(bx, B), dim3(256), 0, s, x, S, mm4);
    return vg_check_launch();
}
__global__ void mm_apply_kernel(const float* x, const float* mm4, int64_t S, float* y) {
    const int b = blockIdx.y;
    const float mn = mm4[b * 4], r = mm4[b * 4 + 1] - mn;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x)
        y[(size_t)b * S + i] = (x[(size_t)b * S + i] - mn) / r;
}
extern "C" int vg_minmax_apply(const float* x, const float* mm4, int B, int64_t S, float* y, vg_stream_t stream) {
    vg_begin();
    if (!x || !mm4 || !y || B < 1 || S < 1) return VG_EINVAL;
    hipLaunchKernelGGL(mm_apply_kernel, dim3(lblocks(S), B), dim3(256), 0, (hipStream_t)stream, x, mm4, S, y);
    return vg_check_launch();
}
__global__ void mm_bwd_sums_kernel(const float* y, const float* gy, int64_t S, float* tmp2) {
    __shared__ float sm[4];
    const int b = blockIdx.y;
    float a = 0.f, c = 0.f;
    const float* gb = gy + (size_t)b * S; const float* yb = y + (size_t)b * S;
    const int64_t S4 = ((((uintptr_t)gb | (uintptr_t)yb) & 15) == 0) ? S >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 g = ((const f32x4*)gb)[i], v = ((const f32x4*)yb)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a += g[j] * (v[j] - 1.f); c += g[j] * v[j]; }
    }
    for (int64_t i = 4 * S4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = gb[i], v = yb[i];
        a += g * (v - 1.f); c += g * v;
    }
    a = block_sum(a, sm); c = block_sum(c, sm);
    if (threadIdx.x == 0) { atomicAdd(&tmp2[b * 2], a); atomicAdd(&tmp2[b * 2 + 1], c); }
}
__global__ void mm_bwd_apply_kernel(const float* x, const float* gy, const float* mm4, const float* tmp2, int64_t S, float* dx) {
    const int b = blockIdx.y;
    const float mn = mm4[b * 4], mx = mm4[b * 4 + 1], r = mx - mn;
    const float gmn = tmp2[b * 2] / (r * mm4[b * 4 + 2]), gmx = -tmp2[b * 2 + 1] / (r * mm4[b * 4 + 3]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[(size_t)b * S + i];
        float g = gy[(size_t)b * S + i] / r;
        if (v == mn) g += gmn;
        if (v == mx) g += gmx;
        dx[(size_t)b * S + i] = g;
    }
}
extern "C" int vg_minmax_bwd(const float* x, const float* y, const float* gy, const float* mm4, int B, int64_t S,
                             float* tmp2, float* dx, vg_stream_t stream) {
    vg_begin();
    if (!x || !y || !gy || !mm4 || !tmp2 || !dx || B < 1 || S < 1) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int bx = lblocks(S, 1024) > 512 ? 512 : lblocks(S, 1024);
    hipLaunchKernelGGL(mm_bwd_sums_kernel, dim3(bx, B), dim3(256), 0, s, y, gy, S, tmp2);
    hipLaunchKernelGGL(mm_bwd_apply_kernel, dim3(lblocks(S), B), dim3(256), 0, s, x, gy, mm4, tmp2, S, dx);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// BCE / MSE
// ------------------------------------------------------------------------------------------------
#define VG_BCE_EPS 1e-7f
__global__ void bce_kernel(const float* t, const float* p, int64_t n, float* acc, float gscale, float* gp, int accum) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float y = t[i], pr = p[i];
        const float pc = fminf(fmaxf(pr, VG_BCE_EPS), 1.f - VG_BCE_EPS);
        const float a = pc + VG_BCE_EPS, c = 1.f - pc + VG_BCE_EPS;
        s += -(y * logf(a) + (1.f - y) * logf(c));
        if (gp) {
            const bool pass = pr >= VG_BCE_EPS && pr <= 1.f - VG_BCE_EPS;
            const float g = pass ? gscale * (-(y / a) + (1.f - y) / c) : 0.f;
            gp[i] = accum ? gp[i] + g : g;
        }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
extern "C" int vg_bce(const float* t, const float* p, int64_t n, float* acc, float gscale, float* gp, int accumulate,
                      vg_stream_t stream) {
    vg_begin();
    if (!t || !p || !acc || n < 1) return VG_EINVAL;
    hipLaunchKernelGGL(bce_kernel, dim3(lblocks(n, 1024) > 511 ? 511 : lblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, t, p, n, acc, gscale, gp,
                       accumulate);          // (one float atomic per block on one address: <= 511 blocks)
    return vg_check_launch();
}
__global__ void mse_kernel(const float* a, const float* b, int64_t n, float* acc, float gscale, float* gb, int accum) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = b[i] - a[i];
        s += d * d;
        if (gb) { const float g = gscale * 2.f * d; gb[i] = accum ? gb[i] + g : g; }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
extern "C" int vg_mse(const float* a, const float* b, int64_t n, float* acc, float gscale, float* gb, int accumulate,
                      vg_stream_t stream) {
    vg_begin();
    if (!a || !b || !acc || n < 1) return VG_EINVAL;
    hipLaunchKernelGGL(mse_kernel, dim3(lblocks(n, 1024) > 511 ? 511 : lblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, n, acc, gscale, gb,
                       accumulate);
    return vg_check_launch();
}
__global__ void mse_const_kernel(const void* x, int x_f32, float target, int64_t n, float* acc, float gscale, float* gx, int accum) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x_f32 ? ((const float*)x)[i] : bf2f(((const bf16_t*)x)[i]);
        const float d = v - target;
        s += d * d;
        if (gx) { const float g = gscale * 2.f * d; gx[i] = accum ? gx[i] + g : g; }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
extern "C" int vg_mse_const(const void* x, int x_f32, float target, int64_t n, float* acc, float gscale, float* gx,
                            int accumulate, vg_stream_t stream) {
    vg_begin();
    if (!x || !acc || n < 1) return VG_EINVAL;
    hipLaunchKernelGGL(mse_const_kernel, dim3(lblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, x_f32, target, n, acc,
                       gscale, gx, accumulate);
    return vg_check_launch();
}
// (First version: 2 048 blocks of scalar loads, three float atomics per block on ONE cache line -- 82 us for two 8-MB volumes, all of it
// the serialised atomics.  Now <= 255 blocks, 16-byte loads.)
__global__ void dot_sums_kernel(const float* a, const float* b, int64_t n, float* sums3) {
    __shared__ float sm[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int64_t n4 = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) ? n >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 x = ((const f32x4*)a)[i], y = ((const f32x4*)b)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0 += x[j] * y[j]; s1 += x[j]; s2 += y[j]; }
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[i], y = b[i]; s0 += x * y; s1 += x; s2 += y;
    }
    s0 = block_sum(s0, sm); s1 = block_sum(s1, sm); s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) { atomicAdd(&sums3[0], s0); atomicAdd(&sums3[1], s1); atomicAdd(&sums3[2], s2); }
}
extern "C" int vg_dot_sums(const float* a, const float* b, int64_t n, float* sums3, vg_stream_t stream) {
    vg_begin();
    if (!a || !b || !sums3 || n < 1) return VG_EINVAL;
    const int blocks = lblocks(n, 4096) > 255 ? 255 : lblocks(n, 4096);
    hipLaunchKernelGGL(dot_sums_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, n, sums3);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// SSIM with a 3^3 Gaussian (sigma 1.5), zero 'SAME' padding
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gauss3(float* g) {
    const float e = expf(-0.5f * (1.f / 1.5f) * (1.f / 1.5f));       // loss_functions.py:88-91
    const float s = 1.f + 2.f * e;
    g[0] = e / s; g[1] = 1.f / s; g[2] = e / s;
}
#define SSIM_C1 1e-4f
#define SSIM_C2 9e-4f
__global__ void ssim_fwd_kernel(const float* t, const float* p, int B, int D, int H, int W, float* acc, float* part) {
    __shared__ float sm[4];
    float g[3]; gauss3(g);
    const int64_t S = (int64_t)D * H * W, total = (int64_t)B * S;
    float loss = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W); int64_t r = i / W; const int h = (int)(r % H); r /= H; const int d = (int)(r % D); const int b = (int)(r / D);
        float mt = 0.f, mp = 0.f, ett = 0.f, epp = 0.f, etp = 0.f;
        for (int a = -1; a <= 1; ++a) { const int dd = d + a; if (dd < 0 || dd >= D) continue;
            for (int c = -1; c <= 1; ++c) { const int hh = h + c; if (hh < 0 || hh >= H) continue;
                for (int e = -1; e <= 1; ++e) { const int ww = w + e; if (ww < 0 || ww >= W) continue;
                    const float wt = g[a + 1] * g[c + 1] * g[e + 1];
                    const size_t j = (size_t)b * S + ((size_t)dd * H + hh) * W + ww;
                    const float tv = t[j], pv = p[j];
                    mt += wt * tv; mp += wt * pv; ett += wt * tv * tv; epp += wt * pv * pv; etp += wt * tv * pv;
                } } }
        const float stt = ett - mt * mt, spp = epp - mp * mp, stp = etp - mt * mp;
        const float A = 2.f * mt * mp + SSIM_C1, Bq = 2.f * stp + SSIM_C2;
        const float Cq = mt * mt + mp * mp + SSIM_C1, Dq = stt + spp + SSIM_C2;
        const float inv = 1.f / (Cq * Dq);
        const float ssim = A * Bq * inv;
        loss += 1.f - ssim;
        if (part) {
            const float dmu = (2.f * mt * Bq - 2.f * mt * A) * inv - ssim * (2.f * mp * Dq - 2.f * mp * Cq) * inv;
            const float depp = -ssim / Dq;
            const float detp = 2.f * A * inv;
            part[i] = -dmu; part[total + i] = -depp; part[2 * total + i] = -detp;
        }
    }
    loss = block_sum(loss, sm);
    if (threadIdx.x == 0) atomicAdd(acc, loss);
}
extern "C" int vg_ssim_fwd(const float* t, const float* p, int B, int D, int H, int W, float* acc, float* part,
                           vg_stream_t stream) {
    vg_begin();
    if (!t || !p || !acc || B < 1 || D < 1 || H < 1 || W < 1) return VG_EINVAL;
    const int blocks = lblocks((int64_t)B * D * H * W) > 1023 ? 1023 : lblocks((int64_t)B * D * H * W);          // one float atomic per block on one address
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, B, D, H, W, acc, part);
    return vg_check_launch();
}
__global__ void ssim_bwd_kernel(const float* t, const float* p, const float* part, int B, int D, int H, int W, float gscale,
                                float* gp, int accum) {
    float g[3]; gauss3(g);
    const int64_t S = (int64_t)D * H * W, total = (int64_t)B * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W); int64_t r = i / W; const int h = (int)(r % H); r /= H; const int d = (int)(r % D); const int b = (int)(r / D);
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
        for (int a = -1; a <= 1; ++a) { const int dd = d + a; if (dd < 0 || dd >= D) continue;
            for (int c = -1; c <= 1; ++c) { const int hh = h + c; if (hh < 0 || hh >= H) continue;
                for (int e = -1; e <= 1; ++e) { const int ww = w + e; if (ww < 0 || ww >= W) continue;
                    const float wt = g[a + 1] * g[c + 1] * g[e + 1];
                    const size_t j = (size_t)b * S + ((size_t)dd * H + hh) * W + ww;
                    f0 += wt * part[j]; f1 += wt * part[total + j]; f2 += wt * part[2 * total + j];
                } } }
        const float v = gscale * (f0 + 2.f * p[i] * f1 + t[i] * f2);
        gp[i] = accum ? gp[i] + v : v;
    }
}
extern "C" int vg_ssim_bwd(const float* t, const float* p, const float* part, int B, int D, int H, int W, float gscale,
                           float* gp, int accumulate, vg_stream_t stream) {
    vg_begin();
    if (!t || !p || !part || !gp || B < 1) return VG_EINVAL;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(lblocks((int64_t)B * D * H * W)), dim3(256), 0, (hipStream_t)stream, t, p, part, B, D, H, W,
                       gscale, gp, accumulate);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// clDice soft skeleton.  Candidate order = TF evaluation order (see oracle/_erode_offsets): the windows
// (3,3,1), (3,1,3), (1,3,3) for soft_erode (19-voxel union), raster order for the 27-voxel soft_dilate.
// TP: the gradient of each pooling goes to the FIRST candidate attaining the extremum.
// ------------------------------------------------------------------------------------------------
// Forward soft-skeleton steps, LDS-tiled.  A block computes a 32 x 8 x 8 (W x H x D) tile: the 34 x 10 x 10 halo goes to
// LDS once (outside the volume: +inf for the erosion, -inf for the dilation = "neighbour skipped"), then a thread walks
// one (h, w) column along D with a sliding window of per-slice partial results -- 9 LDS reads and ~12 min/max per
// output instead of 19/27 bounds-checked global loads.  Values only (min/max are order-independent), so the result is
// bitwise the same as the scan; the backward kernels keep the first-candidate scan because they need the argmin/argmax.
//   ERODE : out = min over the 19-voxel neighbourhood (3x3x3 minus the 8 corners) of `in`
//   !ERODE: dil = max over 3x3x3 of `in` (= img_{j+1}); delta = relu(imgj - dil);
//           out = prev ? prev + relu(delta - prev*delta) : delta            (clDice_func.py:8-31)
#define SK_TW 32
#define SK_TH 8
#define SK_TD 8
template <bool ERODE>
__global__ __launch_bounds__(256) void skel_tile_kernel(const float* __restrict__ in, const float* __restrict__ imgj,
                                                        const float* __restrict__ prev, int D, int H, int W, float* __restrict__ out) {
    __shared__ float t[SK_TD + 2][SK_TH + 2][SK_TW + 2];
    const int tid = threadIdx.x;
    const int tiles_w = (W + SK_TW - 1) / SK_TW, tiles_h = (H + SK_TH - 1) / SK_TH;
    int bt = blockIdx.x;
    const int tw = bt % tiles_w; bt /= tiles_w;
    const int th = bt % tiles_h; const int td = bt / tiles_h;
    const int w0 = tw * SK_TW, h0 = th * SK_TH, d0 = td * SK_TD;
    const size_t vol = (size_t)blockIdx.y * D * H * W;
    const float fill = ERODE ? INFINITY : -INFINITY;
    constexpr int NH = (SK_TD + 2) * (SK_TH + 2) * (SK_TW + 2);
    for (int i = tid; i < NH; i += 256) {
        const int x = i % (SK_TW + 2); const int r = i / (SK_TW + 2);
        const int y = r % (SK_TH + 2), z = r / (SK_TH + 2);
        const int gw = w0 + x - 1, gh = h0 + y - 1, gd = d0 + z - 1;
        float v = fill;
        if (gw >= 0 && gw < W && gh >= 0 && gh < H && gd >= 0 && gd < D) v = in[vol + ((size_t)gd * H + gh) * W + gw];
        (&t[0][0][0])[i] = v;
    }
    __syncthreads();
    const int tx = tid & (SK_TW - 1), ty = tid >> 5;
    const int gw = w0 + tx, gh = h0 + ty;
    float p9[3], p5[3];                       // per-slice partials of slices s-2, s-1, s (ring)
#pragma unroll
    for (int sl = 0; sl < SK_TD + 2; ++sl) {
        const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
        const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
        const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
        float plus, full;
        if (ERODE) {
            plus = fminf(fminf(fminf(a01, a21), fminf(a10, a12)), a11);
            full = fminf(plus, fminf(fminf(a00, a02), fminf(a20, a22)));
        } else {
            plus = fmaxf(fmaxf(fmaxf(a01, a21), fmaxf(a10, a12)), a11);
            full = fmaxf(plus, fmaxf(fmaxf(a00, a02), fmaxf(a20, a22)));
        }
        p9[sl % 3] = full; p5[sl % 3] = plus;
        if (sl >= 2) {
            const int gd = d0 + sl - 2;           // output slice (centre = slice sl-1 of the halo)
            const int c = (sl - 1) % 3, lo = (sl - 2) % 3, hi = sl % 3;
            const float nb = ERODE ? fminf(p9[c], fminf(p5[lo], p5[hi])) : fmaxf(p9[c], fmaxf(p9[lo], p9[hi]));
            if (gw < W && gh < H && gd < D) {
                const size_t o = vol + ((size_t)gd * H + gh) * W + gw;
                if (ERODE) out[o] = nb;
                else {
                    const float delta = fmaxf(imgj[o] - nb, 0.f);
                    if (prev) { const float sp = prev[o]; out[o] = sp + fmaxf(delta - sp * delta, 0.f); }
                    else out[o] = delta;
                }
            }
        }
    }
}
// All skeleton steps of one soft_skel in ONE launch.  skel_j = skel_{j-1} + relu(delta_j - skel_{j-1} * delta_j) with
// delta_j = relu(img_j - dilate(img_{j+1})) is a recursion in j at every voxel whose only neighbourhood access is the dilation of
// the STORED erosion chain, so a block keeps the running skeleton and img_j of its 32 x 8 x 8 tile in registers and only loads the
// tile of img_{j+1} (with halo) per step: 20 instead of 36 bytes per voxel and step, and iters fewer launches per skeleton.  Same
// arithmetic in the same order as skel_tile_kernel<false>: bitwise the same skeletons.
// AUX (round 5, the skeleton that will be differentiated): the launch also files, per step j and voxel, delta_j and the code of the FIRST
// arg-max of the dilation window (0..26 = ((a+1)*3 + (b+1))*3 + (c+1), offsets along D, H, W; the raster order of the reference's
// scan) -- the backward pass then routes gradients by table lookup instead of re-scanning 27 + 19 neighbours per voxel and step.
// The first maximum of each 3 x 3 slice window, then the first of the three slices: the raster order is slice-major, so this IS
// the first candidate of the full scan.
template <bool AUX>
__global__ __launch_bounds__(256) void skel_chain_kernel(const float* __restrict__ imgs, int64_t n, int iters, int D, int H, int W,
                                                         float* __restrict__ skels, float* __restrict__ deltas, unsigned char* __restrict__ codes) {
    __shared__ float t[SK_TD + 2][SK_TH + 2][SK_TW + 2];
    const int tid = threadIdx.x;
    const int tiles_w = (W + SK_TW - 1) / SK_TW, tiles_h = (H + SK_TH - 1) / SK_TH;
    int bt = blockIdx.x;
    const int tw = bt % tiles_w; bt /= tiles_w;
    const int th = bt % tiles_h; const int td = bt / tiles_h;
    const int w0 = tw * SK_TW, h0 = th * SK_TH, d0 = td * SK_TD;
    const size_t vol = (size_t)blockIdx.y * D * H * W;
    constexpr int NH = (SK_TD + 2) * (SK_TH + 2) * (SK_TW + 2);
    const int tx = tid & (SK_TW - 1), ty = tid >> 5;
    const int gw = w0 + tx, gh = h0 + ty;
    const bool col_ok = gw < W && gh < H;
    float cen[SK_TD], sk[SK_TD];
#pragma unroll
    for (int k = 0; k < SK_TD; ++k) {
        const int gd = d0 + k;
        cen[k] = (col_ok && gd < D) ? imgs[vol + ((size_t)gd * H + gh) * W + gw] : 0.f;       // img_0
        sk[k] = 0.f;
    }
    for (int j = 0; j <= iters; ++j) {
        const float* __restrict__ in = imgs + (size_t)(j + 1) * n;
        __syncthreads();                                  // the previous step's readers are done with the tile
        for (int i = tid; i < NH; i += 256) {
            const int x = i % (SK_TW + 2); const int r = i / (SK_TW + 2);
            const int y = r % (SK_TH + 2), z = r / (SK_TH + 2);
            const int qw = w0 + x - 1, qh = h0 + y - 1, qd = d0 + z - 1;
            float v = -INFINITY;
            if (qw >= 0 && qw < W && qh >= 0 && qh < H && qd >= 0 && qd < D) v = in[vol + ((size_t)qd * H + qh) * W + qw];
            (&t[0][0][0])[i] = v;
        }
        __syncthreads();
        float* __restrict__ out = skels + (size_t)j * n;
        float p9[3]; int k9[3];
#pragma unroll
        for (int sl = 0; sl < SK_TD + 2; ++sl) {
            const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
            const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
            const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
            if (AUX) {
                float m = a00; int k = 0;                     // first maximum of the slice window, raster order (b, c), strict >
                if (a01 > m) { m = a01; k = 1; } if (a02 > m) { m = a02; k = 2; }
                if (a10 > m) { m = a10; k = 3; } if (a11 > m) { m = a11; k = 4; } if (a12 > m) { m = a12; k = 5; }
                if (a20 > m) { m = a20; k = 6; } if (a21 > m) { m = a21; k = 7; } if (a22 > m) { m = a22; k = 8; }
                p9[sl % 3] = m; k9[sl % 3] = k;
            } else {
                const float plus = fmaxf(fmaxf(fmaxf(a01, a21), fmaxf(a10, a12)), a11);
                p9[sl % 3] = fmaxf(plus, fmaxf(fmaxf(a00, a02), fmaxf(a20, a22)));
            }
            if (sl >= 2) {
                const int k = sl - 2, gd = d0 + k;
                const float nb = fmaxf(p9[(sl - 1) % 3], fmaxf(p9[(sl - 2) % 3], p9[sl % 3]));
                const float delta = fmaxf(cen[k] - nb, 0.f);
                const float sp = sk[k];
                sk[k] = j ? sp + fmaxf(delta - sp * delta, 0.f) : delta;
                if (col_ok && gd < D) {
                    const size_t o = vol + ((size_t)gd * H + gh) * W + gw;
                    out[o] = sk[k];
                    if (AUX) {
                        float best = p9[(sl - 2) % 3]; int code = k9[(sl - 2) % 3];
                        if (p9[(sl - 1) % 3] > best) { best = p9[(sl - 1) % 3]; code = 9 + k9[(sl - 1) % 3]; }
                        if (p9[sl % 3] > best) { best = p9[sl % 3]; code = 18 + k9[sl % 3]; }
                        deltas[(size_t)j * n + o] = delta; codes[(size_t)j * n + o] = (unsigned char)code;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < SK_TD; ++k) cen[k] = t[k + 1][ty + 1][tx + 1];       // img_{j+1} is the next step's img_j
    }
}
__device__ __forceinline__ int64_t sk_off(int code, int H, int W) {
    const int a = code / 9, r = code - a * 9, b = r / 3, c = r - b * 3;
    return ((int64_t)(a - 1) * H + (b - 1)) * W + (c - 1);
}
#define SK_TILE_ORIGIN()                                                                                    \
    const int tiles_w = (W + SK_TW - 1) / SK_TW, tiles_h = (H + SK_TH - 1) / SK_TH;                           \
    int bt_ = blockIdx.x; const int tw_ = bt_ % tiles_w; bt_ /= tiles_w;                                      \
    const int th_ = bt_ % tiles_h, td_ = bt_ / tiles_h;                                                       \
    const int w0 = tw_ * SK_TW, h0 = th_ * SK_TH, d0 = td_ * SK_TD;                                           \
    const size_t vol = (size_t)blockIdx.y * D * H * W;                                                        \
    const int tx = tid & (SK_TW - 1), ty = tid >> 5; const int gw = w0 + tx, gh = h0 + ty;

// ---- K erosions per launch (round 5).  img_{j+1} = soft_erode(img_j) has a dependency radius of one voxel per step, so a block that
// holds its 32 x 8 x 8 tile with a halo of K voxels in LDS can run K erosions back to back: stage s reads the region with halo
// K - s + 1 from one LDS buffer and leaves the region with halo K - s in the other (voxels outside the volume stay +inf =
// "neighbour skipped", exactly as the per-step kernel pads), writing the tile's own voxels of img_{j+s} to HBM on the way -- every
// intermediate volume is still stored (the skeleton chain and the backward pass read all of them), but the chain is READ once per
// K steps instead of once per step, and a skeleton costs ceil((iters + 1) / K) launches instead of iters + 1.  Same column walk
// with per-slice partial minima as skel_tile_kernel<true>; min is order-independent, so the volumes are bitwise the same.
template <int E>
__device__ __forceinline__ void erode_stage(const float* __restrict__ src, float* __restrict__ dst, int w0, int h0, int d0, int D, int H, int W,
                                            size_t vol, float* __restrict__ out, int tid) {
    constexpr int DD = SK_TD + 2 * E, DH = SK_TH + 2 * E, DW = SK_TW + 2 * E;       // destination region (halo E)
    constexpr int SH = DH + 2, SW = DW + 2;                                          // source region (halo E + 1)
    for (int col = tid; col < DH * DW; col += 256) {
        const int y = col / DW, x = col - y * DW;
        const int gh = h0 - E + y, gw = w0 - E + x;
        const bool col_in = gh >= 0 && gh < H && gw >= 0 && gw < W;
        const bool col_tile = y >= E && y < E + SK_TH && x >= E && x < E + SK_TW;
        float p9[3], p5[3];
#pragma unroll
        for (int sl = 0; sl < DD + 2; ++sl) {
            const float* r0 = src + (sl * SH + y) * SW + x;
            const float* r1 = r0 + SW; const float* r2 = r1 + SW;
            const float a00 = r0[0], a01 = r0[1], a02 = r0[2], a10 = r1[0], a11 = r1[1], a12 = r1[2], a20 = r2[0], a21 = r2[1], a22 = r2[2];
            const float plus = fminf(fminf(fminf(a01, a21), fminf(a10, a12)), a11);
            p9[sl % 3] = fminf(plus, fminf(fminf(a00, a02), fminf(a20, a22))); p5[sl % 3] = plus;
            if (sl >= 2) {
                const int z = sl - 2, gd = d0 - E + z;
                const float nb = fminf(p9[(sl - 1) % 3], fminf(p5[(sl - 2) % 3], p5[sl % 3]));
                const bool in = col_in && gd >= 0 && gd < D;
                if (E > 0) dst[(z * DH + y) * DW + x] = in ? nb : INFINITY;
                if (in && col_tile && z >= E && z < E + SK_TD) out[vol + ((size_t)gd * H + gh) * W + gw] = nb;
            }
        }
    }
}
template <int K>
__global__ __launch_bounds__(256) void erode_multi_kernel(const float* __restrict__ in, int D, int H, int W, int64_t n, float* __restrict__ outs,
                                                          float* __restrict__ copy0) {
    extern __shared__ float sk_lds[];
    constexpr int XD = SK_TD + 2 * K, XH = SK_TH + 2 * K, XW = SK_TW + 2 * K;
    float* A = sk_lds; float* Bf = sk_lds + XD * XH * XW;
    const int tid = threadIdx.x;
    SK_TILE_ORIGIN()
    (void)tx; (void)ty; (void)gw; (void)gh;
    for (int i = tid; i < XD * XH * XW; i += 256) {
        const int x = i % XW; const int r = i / XW; const int y = r % XH, z = r / XH;
        const int qw = w0 - K + x, qh = h0 - K + y, qd = d0 - K + z;
        float v = INFINITY;
        if (qw >= 0 && qw < W && qh >= 0 && qh < H && qd >= 0 && qd < D) {
            const size_t o = vol + ((size_t)qd * H + qh) * W + qw;
            v = in[o];
            // the first launch of a skeleton also files its input as img_0 (the chain and the backward pass index the stored chain)
            if (copy0 && x >= K && x < K + SK_TW && y >= K && y < K + SK_TH && z >= K && z < K + SK_TD) copy0[o] = v;
        }
        A[i] = v;
    }
    __syncthreads();
    erode_stage<K - 1>(A, Bf, w0, h0, d0, D, H, W, vol, outs, tid);
    if constexpr (K >= 2) { __syncthreads(); erode_stage<K - 2>(Bf, A, w0, h0, d0, D, H, W, vol, outs + n, tid); }
    if constexpr (K >= 3) { __syncthreads(); erode_stage<K - 3>(A, Bf, w0, h0, d0, D, H, W, vol, outs + 2 * n, tid); }
    if constexpr (K >= 4) { __syncthreads(); erode_stage<K - 4>(Bf, A, w0, h0, d0, D, H, W, vol, outs + 3 * n, tid); }
}
template <int K>
static void launch_erode_multi(const float* in, int D, int H, int W, int64_t n, float* outs, float* copy0, dim3 grid, hipStream_t s) {
    constexpr int XD = SK_TD + 2 * K, XH = SK_TH + 2 * K, XW = SK_TW + 2 * K;
    constexpr int lds = (XD * XH * XW + (XD - 2) * (XH - 2) * (XW - 2)) * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)erode_multi_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT); attr = true; }
    hipLaunchKernelGGL(erode_multi_kernel<K>, grid, dim3(256), lds, s, in, D, H, W, n, outs, copy0);
}
static dim3 skel_grid(int B, int D, int H, int W) {
    return dim3(((W + SK_TW - 1) / SK_TW) * ((H + SK_TH - 1) / SK_TH) * ((D + SK_TD - 1) / SK_TD), B);
}
// ---- backward steps, LDS-tiled like the forward ones.  The pooling gradients go to the FIRST arg-min / arg-max in the
// reference's scan order (ties: TP, tf max_pool3d / -max_pool3d(-x) gradients), so the candidates are visited in exactly
// that order; outside the volume the tile holds +-inf, which a strict comparison never selects.
__device__ __forceinline__ void sk_load_tile(float (&t)[SK_TD + 2][SK_TH + 2][SK_TW + 2], const float* __restrict__ in, size_t vol,
                                             int w0, int h0, int d0, int D, int H, int W, float fill, int tid) {
    constexpr int NH = (SK_TD + 2) * (SK_TH + 2) * (SK_TW + 2);
    for (int i = tid; i < NH; i += 256) {
        const int x = i % (SK_TW + 2); const int r = i / (SK_TW + 2);
        const int y = r % (SK_TH + 2), z = r / (SK_TH + 2);
        const int gw = w0 + x - 1, gh = h0 + y - 1, gd = d0 + z - 1;
        float v = fill;
        if (gw >= 0 && gw < W && gh >= 0 && gh < H && gd >= 0 && gd < D) v = in[vol + ((size_t)gd * H + gh) * W + gw];
        (&t[0][0][0])[i] = v;
    }
}
// backward step j, part 1: local gradients of the skeleton update
__global__ __launch_bounds__(256) void skel_bwd_local_kernel(const float* __restrict__ imgj, const float* __restrict__ imgj1,
                                                             const float* __restrict__ prev, int D, int H, int W,
                                                             float* gs, float* dimgj, float* dimgj1) {
    __shared__ float t[SK_TD + 2][SK_TH + 2][SK_TW + 2];
    const int tid = threadIdx.x;
    SK_TILE_ORIGIN()
    sk_load_tile(t, imgj1, vol, w0, h0, d0, D, H, W, -INFINITY, tid);
    __syncthreads();
    if (gw >= W || gh >= H) return;
    // dilation value and its FIRST arg-max (scan order a (D), b (H), c (W) ascending) by the forward kernels' column walk: the first
    // maximum of each 3 x 3 slice window, then the first of the three slices (9 LDS reads per voxel instead of 27)
    float m9[3]; int k9[3];
#pragma unroll
    for (int sl = 0; sl < SK_TD + 2; ++sl) {
        const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
        const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
        const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
        float m = a00; int k = 0;
        if (a01 > m) { m = a01; k = 1; } if (a02 > m) { m = a02; k = 2; }
        if (a10 > m) { m = a10; k = 3; } if (a11 > m) { m = a11; k = 4; } if (a12 > m) { m = a12; k = 5; }
        if (a20 > m) { m = a20; k = 6; } if (a21 > m) { m = a21; k = 7; } if (a22 > m) { m = a22; k = 8; }
        m9[sl % 3] = m; k9[sl % 3] = k;
        if (sl < 2) continue;
        const int gd = d0 + sl - 2;
        if (gd >= D) continue;
        const size_t i = vol + ((size_t)gd * H + gh) * W + gw;
        float dil = m9[(sl - 2) % 3]; int code = k9[(sl - 2) % 3];
        if (m9[(sl - 1) % 3] > dil) { dil = m9[(sl - 1) % 3]; code = 9 + k9[(sl - 1) % 3]; }
        if (m9[sl % 3] > dil) { dil = m9[sl % 3]; code = 18 + k9[sl % 3]; }
        const float raw = imgj[i] - dil;
        const float delta = fmaxf(raw, 0.f);
        const float g = gs[i];
        float ddelta;
        if (prev) {
            const float sp = prev[i];
            const float u = delta - sp * delta;
            const float dr = u > 0.f ? g : 0.f;
            ddelta = dr * (1.f - sp);
            gs[i] = g - dr * delta;             // d skel_{j-1}
        } else ddelta = g;
        const float e = raw > 0.f ? ddelta : 0.f;
        if (e != 0.f) { dimgj[i] += e; atomicAdd(&dimgj1[(int64_t)i + sk_off(code, H, W)], -e); }
    }
}
// part 2: d img_j += erode^T(d img_{j+1})
// (dimgj1 is consumed: every element is read by exactly one thread, which leaves a zero behind -- the buffer is the NEXT iteration's
// accumulator, and the memset launch between two iterations sat on lane B's dependent chain)
__global__ __launch_bounds__(256) void erode_bwd_kernel(const float* __restrict__ imgj, float* __restrict__ dimgj1,
                                                        int D, int H, int W, float* dimgj) {
    __shared__ float t[SK_TD + 2][SK_TH + 2][SK_TW + 2];
    const int tid = threadIdx.x;
    SK_TILE_ORIGIN()
    sk_load_tile(t, imgj, vol, w0, h0, d0, D, H, W, INFINITY, tid);
    __syncthreads();
    if (gw >= W || gh >= H) return;
    // first arg-min of the 19-voxel footprint in the reference's scan order, by the column walk of skel_erode_code_kernel
    float vC[3], vR[3], vF[3]; int iC[3], iR[3], iF[3];
#pragma unroll
    for (int sl = 0; sl < SK_TD + 2; ++sl) {
        const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
        const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
        const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
        { float m = a01; int k = 0; if (a11 < m) { m = a11; k = 1; } if (a21 < m) { m = a21; k = 2; } vC[sl % 3] = m; iC[sl % 3] = k; }
        { float m = a10; int k = 0; if (a11 < m) { m = a11; k = 1; } if (a12 < m) { m = a12; k = 2; } vR[sl % 3] = m; iR[sl % 3] = k; }
        { float m = a00; int k = 0;
          if (a01 < m) { m = a01; k = 1; } if (a02 < m) { m = a02; k = 2; }
          if (a10 < m) { m = a10; k = 3; } if (a11 < m) { m = a11; k = 4; } if (a12 < m) { m = a12; k = 5; }
          if (a20 < m) { m = a20; k = 6; } if (a21 < m) { m = a21; k = 7; } if (a22 < m) { m = a22; k = 8; }
          vF[sl % 3] = m; iF[sl % 3] = k; }
        if (sl < 2) continue;
        const int gd = d0 + sl - 2;
        if (gd >= D) continue;
        const size_t i = vol + ((size_t)gd * H + gh) * W + gw;
        const float g = dimgj1[i];
        if (g == 0.f) continue;
        dimgj1[i] = 0.f;
        const int lo = (sl - 2) % 3, c = (sl - 1) % 3, hi = sl % 3;
        float best = vC[lo]; int code = iC[lo] * 3 + 1;
        if (vC[c] < best) { best = vC[c]; code = 9 + iC[c] * 3 + 1; }
        if (vC[hi] < best) { best = vC[hi]; code = 18 + iC[hi] * 3 + 1; }
        if (vR[lo] < best) { best = vR[lo]; code = 3 + iR[lo]; }
        if (vR[c] < best) { best = vR[c]; code = 12 + iR[c]; }
        if (vR[hi] < best) { best = vR[hi]; code = 21 + iR[hi]; }
        if (vF[c] < best) { best = vF[c]; code = 9 + iF[c]; }
        atomicAdd(&dimgj[(int64_t)i + sk_off(code, H, W)], g);
    }
}

// soft_erode with the code of the FIRST arg-min in the reference's scan order (TP: the windows (3,3,1), (3,1,3), (1,3,3) one after
// the other, raster order inside each, strict <): per slice the first minimum of the H-column (b; c = 0), of the W-row (c; b = 0) and
// of the whole 3 x 3 window; then set 1 = the three slices' columns in slice order, set 2 = the rows, set 3 = the centre slice's
// window.  Values are those of skel_tile_kernel<true> (min is order-independent): the stored chain is bitwise the same.
__global__ __launch_bounds__(256) void skel_erode_code_kernel(const float* __restrict__ in, int D, int H, int W, float* __restrict__ out,
                                                              unsigned char* __restrict__ codes) {
    __shared__ float t[SK_TD + 2][SK_TH + 2][SK_TW + 2];
    const int tid = threadIdx.x;
    SK_TILE_ORIGIN()
    sk_load_tile(t, in, vol, w0, h0, d0, D, H, W, INFINITY, tid);
    __syncthreads();
    float vC[3], vR[3], vF[3]; int iC[3], iR[3], iF[3];
#pragma unroll
    for (int sl = 0; sl < SK_TD + 2; ++sl) {
        const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
        const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
        const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
        { float m = a01; int k = 0; if (a11 < m) { m = a11; k = 1; } if (a21 < m) { m = a21; k = 2; } vC[sl % 3] = m; iC[sl % 3] = k; }      // (b, 0)
        { float m = a10; int k = 0; if (a11 < m) { m = a11; k = 1; } if (a12 < m) { m = a12; k = 2; } vR[sl % 3] = m; iR[sl % 3] = k; }      // (0, c)
        { float m = a00; int k = 0;
          if (a01 < m) { m = a01; k = 1; } if (a02 < m) { m = a02; k = 2; }
          if (a10 < m) { m = a10; k = 3; } if (a11 < m) { m = a11; k = 4; } if (a12 < m) { m = a12; k = 5; }
          if (a20 < m) { m = a20; k = 6; } if (a21 < m) { m = a21; k = 7; } if (a22 < m) { m = a22; k = 8; }
          vF[sl % 3] = m; iF[sl % 3] = k; }
        if (sl >= 2) {
            const int gd = d0 + sl - 2;
            const int lo = (sl - 2) % 3, c = (sl - 1) % 3, hi = sl % 3;
            float best = vC[lo]; int code = iC[lo] * 3 + 1;                                  // set 1: (a, b, 0)
            if (vC[c] < best) { best = vC[c]; code = 9 + iC[c] * 3 + 1; }
            if (vC[hi] < best) { best = vC[hi]; code = 18 + iC[hi] * 3 + 1; }
            if (vR[lo] < best) { best = vR[lo]; code = 3 + iR[lo]; }                         // set 2: (a, 0, c)
            if (vR[c] < best) { best = vR[c]; code = 12 + iR[c]; }
            if (vR[hi] < best) { best = vR[hi]; code = 21 + iR[hi]; }
            if (vF[c] < best) { best = vF[c]; code = 9 + iF[c]; }                            // set 3: (0, b, c)
            if (gw < W && gh < H && gd < D) {
                const size_t o = vol + ((size_t)gd * H + gh) * W + gw;
                out[o] = best; codes[o] = (unsigned char)code;
            }
        }
    }
}

// ---- backward from the stored codes (round 5): pure streaming launches, no neighbourhood scans ----
// Launch F_k (k = iters + 1 ... 0) runs two independent pieces that both accumulate into d img_k:
//   A (k <= iters): erosion transpose of step k,   d img_k[i + off(argmin code_k[i])] += d img_{k+1}[i]      (d img_{k+1} is complete)
//   B (k >= 1):     local gradient of step j = k-1, e = e_j(i) from (delta_j, skel_{j-1}, gs); gs <- d skel_{j-1} in place;
//                   d img_j[i] = e (first writer: plain store);   d img_k[i + off(argmax code_j[i])] -= e
// d img_{k+1} gets its last contribution in F_{k+1}, so the launches only need stream order.  iters + 2 launches per skeleton instead of
// 2 (iters + 1) + memset + final add; the atomics are the sparse ones (e != 0 only on thin structures).
template <int V> __device__ __forceinline__ void sk_ldf(const float* p, int64_t q, float* o) {
    if constexpr (V == 4) { const f32x4 v = ((const f32x4*)p)[q]; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; } else o[0] = p[q];
}
template <int V> __device__ __forceinline__ void sk_stf(float* p, int64_t q, const float* o) {
    if constexpr (V == 4) ((f32x4*)p)[q] = (f32x4){o[0], o[1], o[2], o[3]}; else p[q] = o[0];
}
template <int V> __device__ __forceinline__ void sk_ldc(const unsigned char* p, int64_t q, int* o) {
    if constexpr (V == 4) { const unsigned w = ((const unsigned*)p)[q]; o[0] = w & 255; o[1] = (w >> 8) & 255; o[2] = (w >> 16) & 255; o[3] = w >> 24; }
    else o[0] = p[q];
}
template <int V>
__global__ __launch_bounds__(256) void skel_bwd_stream_kernel(const float* __restrict__ d_in, const unsigned char* __restrict__ codeN,
                                                              float* d_acc, float* d_out, int out_accumulate,
                                                              const float* __restrict__ delta, const unsigned char* __restrict__ codeM,
                                                              const float* __restrict__ prev, const float* gs_in, float* gs_out,
                                                              int64_t n, int H, int W) {
    // V voxels per thread and iteration (V = 4: 16-byte loads / stores, 4-byte code loads; the caller guarantees n % 4 == 0)
    const int64_t nv = n / V;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = q * V;
        float g_in[V], dl[V], sp[V], g[V]; int cn[V], cm[V];
        if (d_in) { sk_ldf<V>(d_in, q, g_in); sk_ldc<V>(codeN, q, cn); }
        if (delta) { sk_ldf<V>(delta, q, dl); sk_ldf<V>(gs_in, q, g); sk_ldc<V>(codeM, q, cm); if (prev) sk_ldf<V>(prev, q, sp); }
        if (d_in) {                                               // piece A
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (g_in[v] != 0.f) atomicAdd(&d_acc[i0 + v + sk_off(cn[v], H, W)], g_in[v]);
        }
        if (delta) {                                              // piece B
            float e[V], gn[V];
            bool changed = false;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float ddelta; gn[v] = g[v];
                if (prev) {
                    const float u = dl[v] - sp[v] * dl[v];
                    const float dr = u > 0.f ? g[v] : 0.f;
                    ddelta = dr * (1.f - sp[v]);
                    gn[v] = g[v] - dr * dl[v];                    // d skel_{j-1}
                } else ddelta = g[v];
                e[v] = dl[v] > 0.f ? ddelta : 0.f;                // delta > 0 <=> raw > 0
                changed |= gn[v] != g[v];
                if (e[v] != 0.f) atomicAdd(&d_acc[i0 + v + sk_off(cm[v], H, W)], -e[v]);
            }
            // gs only moves where the skeleton update was active (thin structures): elsewhere the line is not rewritten -- unless this
            // launch is the one that copies the caller's gradient into the private buffer
            if (prev && (changed || gs_in != gs_out)) sk_stf<V>(gs_out, q, gn);
            if (out_accumulate) {
                float o[V]; sk_ldf<V>(d_out, q, o);
#pragma unroll
                for (int v = 0; v < V; ++v) e[v] += o[v];
            }
            sk_stf<V>(d_out, q, e);
        }
    }
}
extern "C" int vg_soft_skel_fwd(const float* img, int B, int D, int H, int W, int iters, float* imgs, float* skels, void* aux,
                                vg_stream_t stream) {
    vg_begin();
    if (!img || !imgs || !skels || B < 1 || iters < 0) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * D * H * W;
    const dim3 grid = skel_grid(B, D, H, W);
    const int multi = vg_tune("SKEL_MULTI", 2);
    // aux (the skeleton that will be differentiated): [iters+1][n] float delta | [iters+1][n] u8 arg-max codes | [iters+1][n] u8 arg-min codes
    float* deltas = (float*)aux;
    unsigned char* codeM = aux ? (unsigned char*)(deltas + (size_t)(iters + 1) * n) : nullptr;
    unsigned char* codeN = aux ? codeM + (size_t)(iters + 1) * n : nullptr;
    if (aux) {
        if (hipMemcpyAsync(imgs, img, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VG_ELAUNCH;
        for (int j = 0; j <= iters; ++j)
            hipLaunchKernelGGL(skel_erode_code_kernel, grid, dim3(256), 0, s, (const float*)(imgs + j * n), D, H, W, imgs + (j + 1) * n, codeN + (size_t)j * n);
    } else if (multi > 1) {
        // K erosions per launch; the first launch files its input as img_0 (no copy launch in front)
        for (int j = 0; j <= iters;) {
            const int left = iters + 1 - j;
            const float* in = j ? imgs + j * n : img;
            float* c0 = j ? nullptr : imgs;
            if (left >= 4 && multi >= 4) { launch_erode_multi<4>(in, D, H, W, n, imgs + (j + 1) * n, c0, grid, s); j += 4; }
            else if (left >= 2) { launch_erode_multi<2>(in, D, H, W, n, imgs + (j + 1) * n, c0, grid, s); j += 2; }
            else { launch_erode_multi<1>(in, D, H, W, n, imgs + (j + 1) * n, c0, grid, s); j += 1; }
        }
    } else {
        if (hipMemcpyAsync(imgs, img, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VG_ELAUNCH;
        for (int j = 0; j <= iters; ++j)
            hipLaunchKernelGGL(skel_tile_kernel<true>, grid, dim3(256), 0, s, imgs + j * n, (const float*)nullptr, (const float*)nullptr,
                               D, H, W, imgs + (j + 1) * n);
    }
    if (aux) {
        hipLaunchKernelGGL(skel_chain_kernel<true>, grid, dim3(256), 0, s, (const float*)imgs, n, iters, D, H, W, skels, deltas, codeM);
        return vg_check_launch();
    }
    if (vg_tune("SKEL_CHAIN", 1)) {
        hipLaunchKernelGGL(skel_chain_kernel<false>, grid, dim3(256), 0, s, (const float*)imgs, n, iters, D, H, W, skels, (float*)nullptr, (unsigned char*)nullptr);
        return vg_check_launch();
    }
    for (int j = 0; j <= iters; ++j)
        hipLaunchKernelGGL(skel_tile_kernel<false>, grid, dim3(256), 0, s, imgs + (j + 1) * n, imgs + j * n,
                           j ? skels + (j - 1) * n : (const float*)nullptr, D, H, W, skels + j * n);
    return vg_check_launch();
}

extern "C" int vg_soft_skel_bwd(const float* imgs, const float* skels, const float* gskel, int B, int D, int H, int W,
                                int iters, float* work, float* gimg, const void* aux, vg_stream_t stream) {
    vg_begin();
    if (!imgs || !skels || !gskel || !work || !gimg || B < 1 || iters < 0) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * D * H * W;
    const int blocks = lblocks(n);
    (void)blocks;
    const dim3 grid = skel_grid(B, D, H, W);
    if (aux) {
        // streaming launches F_{iters+1} ... F_0 over the codes the forward pass filed (see skel_bwd_stream_kernel)
        const float* deltas = (const float*)aux;
        const unsigned char* codeM = (const unsigned char*)(deltas + (size_t)(iters + 1) * n);
        const unsigned char* codeN = codeM + (size_t)(iters + 1) * n;
        float* gs = work;
        float* d[3] = {work + n, work + 2 * n, work + 3 * n};
        if (hipMemsetAsync(d[(iters + 1) % 3], 0, n * sizeof(float), s) != hipSuccess) return VG_ELAUNCH;       // d img_{iters+1}: only scattered into
        const bool vec = (n % 4) == 0 && ((((uintptr_t)work | (uintptr_t)gimg | (uintptr_t)gskel | (uintptr_t)skels | (uintptr_t)aux) & 15) == 0);
        const int blocks = lblocks(vec ? n / 4 : n);
        for (int k = iters + 1; k >= 0; --k) {
            const int j = k - 1;                                                  // piece B's step
            const bool A = k <= iters, Bp = k >= 1;
            float* d_acc = k == 0 ? gimg : d[k % 3];                              // d img_k (k = 0: straight into the caller's gradient)
            float* d_out = j == 0 ? gimg : (Bp ? d[j % 3] : nullptr);             // d img_j
#define SK_BWD_ARGS A ? (const float*)d[(k + 1) % 3] : (const float*)nullptr, A ? codeN + (size_t)k * n : (const unsigned char*)nullptr, \
                    d_acc, d_out, j == 0 ? 1 : 0, Bp ? deltas + (size_t)j * n : (const float*)nullptr,                                \
                    Bp ? codeM + (size_t)j * n : (const unsigned char*)nullptr,                                                        \
                    (Bp && j > 0) ? skels + (size_t)(j - 1) * n : (const float*)nullptr, k == iters + 1 ? gskel : (const float*)gs, gs, n, H, W
            if (vec) hipLaunchKernelGGL(skel_bwd_stream_kernel<4>, dim3(blocks), dim3(256), 0, s, SK_BWD_ARGS);
            else hipLaunchKernelGGL(skel_bwd_stream_kernel<1>, dim3(blocks), dim3(256), 0, s, SK_BWD_ARGS);
#undef SK_BWD_ARGS
        }
        return vg_check_launch();
    }
    float* gs = work; float* bufA = work + n; float* bufB = work + 2 * n;     // bufA = d img_{j+1}, bufB = d img_j
    if (hipMemcpyAsync(gs, gskel, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VG_ELAUNCH;
    if (hipMemsetAsync(bufA, 0, 2 * n * sizeof(float), s) != hipSuccess) return VG_ELAUNCH;
    for (int j = iters; j >= 0; --j) {
        hipLaunchKernelGGL(skel_bwd_local_kernel, grid, dim3(256), 0, s, imgs + j * n, imgs + (j + 1) * n,
                           j ? skels + (j - 1) * n : (const float*)nullptr, D, H, W, gs, bufB, bufA);
        hipLaunchKernelGGL(erode_bwd_kernel, grid, dim3(256), 0, s, imgs + j * n, bufA, D, H, W, bufB);
        if (j > 0) { float* t = bufA; bufA = bufB; bufB = t; }          // (bufA was zeroed by erode_bwd_kernel as it was read)
    }
    // bufB = d img_0
    return vg_axpby(bufB, 1.f, nullptr, 0.f, n, gimg, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// clDice / Dice combination (clDice_func.py:83-149) -- coefficients on device, no host sync
// sums: [0..2] = (sum skel_p*t, sum skel_p, sum t) ; [3..5] = (sum skel_t*p, sum skel_t, sum p) ; [6] = sum t*p
// coef: gskel_p = c0*t - c1 ;  gp += c2*t + c3 + c4*skel_t ;  coef[5] = loss value (already * w)
// ------------------------------------------------------------------------------------------------
__global__ void cldice_coef_kernel(const float* sums, float w, float alpha, float* coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float A = sums[0], Bp = sums[1], T = sums[2], Cc = sums[3], Dt = sums[4], P = sums[5], I = sums[6];
    const float pres = (A + 1.f) / (Bp + 1.f), rec = (Cc + 1.f) / (Dt + 1.f);
    const float cl = 1.f - 2.f * pres * rec / (pres + rec);
    const float den = T + P + 1.f;
    const float dice = 1.f - (2.f * I + 1.f) / den;
    const float q = (pres + rec) * (pres + rec);
    const float dcl_dpres = -2.f * rec * rec / q, dcl_drec = -2.f * pres * pres / q;
    coef[0] = w * alpha * dcl_dpres / (Bp + 1.f);
    coef[1] = w * alpha * dcl_dpres * (A + 1.f) / ((Bp + 1.f) * (Bp + 1.f));
    coef[2] = -w * (1.f - alpha) * 2.f / den;
    coef[3] = w * (1.f - alpha) * (2.f * I + 1.f) / (den * den);
    coef[4] = w * alpha * dcl_drec / (Dt + 1.f);
    coef[5] = w * ((1.f - alpha) * dice + alpha * cl);
}
__global__ void cldice_grads_kernel(const float* t, const float* skel_t, const float* coef, int64_t n, float* gskel_p, float* gp, int accum) {
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float tv = t[i];
        gskel_p[i] = c0 * tv - c1;
        const float g = c2 * tv + c3 + c4 * skel_t[i];
        gp[i] = accum ? gp[i] + g : g;
    }
}
extern "C" int vg_cldice_coef(const float* sums7, float w, float alpha, float* coef6, vg_stream_t stream) {
    vg_begin();
    if (!sums7 || !coef6) return VG_EINVAL;
    hipLaunchKernelGGL(cldice_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums7, w, alpha, coef6);
    return vg_check_launch();
}
extern "C" int vg_cldice_grads(const float* t, const float* skel_t, const float* coef6, int64_t n, float* gskel_p, float* gp,
                               int accumulate, vg_stream_t stream) {
    vg_begin();
    if (!t || !skel_t || !coef6 || !gskel_p || !gp || n < 1) return VG_EINVAL;
    hipLaunchKernelGGL(cldice_grads_kernel, dim3(lblocks(n)), dim3(256), 0, (hipStream_t)stream, t, skel_t, coef6, n, gskel_p, gp, accumulate);
    return vg_check_launch();
}


// ------------------------------------------------------------------------------------------------
// Wasserstein mode (what `wasserstein=True` of the reference actually trains): the discriminator's
// Flatten -> Dropout(0.2) -> Dense(1) head over the patch logits (discriminator.py:116-119) and the critic / generator loss terms
// (loss_functions.py:325-355: -reduce_mean(D(real) - D(fake)), -reduce_mean(D(fake))).  A few thousand values per sample: one workgroup each.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_head_fwd_kernel(const float* x, const float* mask, const float* w, const float* b, int n, float* z) {
    __shared__ float sm[4];
    const int s = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += x[(size_t)s * n + i] * (mask ? mask[(size_t)s * n + i] : 1.f) * w[i];
    a = block_sum(a, sm);
    if (threadIdx.x == 0) z[s] = a + (b ? b[0] : 0.f);
}
// dx[s][i] = gz[s] * m[s][i] * w[i];  dw[i] += sum_s gz[s] * m[s][i] * x[s][i];  db += sum_s gz[s]     (grid: ceil(n / 256) blocks)
__global__ __launch_bounds__(256) void dense_head_bwd_kernel(const float* x, const float* mask, const float* w, const float* gz, int N, int n,
                                                             float* dx, float* dw, float* db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float acc = 0.f;
        const float wi = w[i];
        for (int s = 0; s < N; ++s) {
            const float m = mask ? mask[(size_t)s * n + i] : 1.f, g = gz[s];
            if (dx) dx[(size_t)s * n + i] = g * m * wi;
            acc += g * m * x[(size_t)s * n + i];
        }
        if (dw) dw[i] += acc;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) { float t = 0.f; for (int s = 0; s < N; ++s) t += gz[s]; db[0] += t; }
}
// z: [2B] head outputs of [real; fake].  acc[0] += sum z_real, acc[1] += sum z_fake; gz_d[2B] = d(D loss)/dz = (-inv ... , +inv ...),
// gz_g[B] = d(G loss)/dz_fake = -inv, inv = 1 / (B * global batch size).
__global__ void wasserstein_terms_kernel(const float* z, int B, float inv, float* acc, float* gz_d, float* gz_g) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float r = 0.f, f = 0.f;
    for (int s = 0; s < B; ++s) { r += z[s]; f += z[B + s]; if (gz_d) { gz_d[s] = -inv; gz_d[B + s] = inv; } if (gz_g) gz_g[s] = -inv; }
    acc[0] += r; acc[1] += f;
}
extern "C" int vg_dense_head_fwd(const float* x, const float* mask, const float* w, const float* b, int N, int n, float* z, vg_stream_t stream) {
    vg_begin();
    if (!x || !w || !z || N < 1 || n < 1) return VG_EINVAL;
    hipLaunchKernelGGL(dense_head_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, mask, w, b, n, z);
    return vg_check_launch();
}
extern "C" int vg_dense_head_bwd(const float* x, const float* mask, const float* w, const float* gz, int N, int n, float* dx, float* dw, float* db,
                                 vg_stream_t stream) {
    vg_begin();
    if (!x || !w || !gz || N < 1 || n < 1 || (!dx && !dw && !db)) return VG_EINVAL;
    hipLaunchKernelGGL(dense_head_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, mask, w, gz, N, n, dx, dw, db);
    return vg_check_launch();
}
extern "C" int vg_wasserstein_terms(const float* z, int B, float inv, float* acc2, float* gz_d, float* gz_g, vg_stream_t stream) {
    vg_begin();
    if (!z || !acc2 || B < 1) return VG_EINVAL;
    hipLaunchKernelGGL(wasserstein_terms_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, z, B, inv, acc2, gz_d, gz_g);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Sliding-window inference (custom_callback.py:47-223, stitch_subvolumes): overlap-add of border-cropped window
// predictions with a coverage counter, then division.  Volumes are [X][Y][Z] fp32 (single channel).
// ------------------------------------------------------------------------------------------------
__global__ void overlap_add_kernel(const float* win, int kx, int ky, int kz, int px, int py, int pz, int x0, int y0, int z0,
                                   int Y, int Z, float* pred, float* cnt) {
    const int cx = kx - 2 * px, cy = ky - 2 * py, cz = kz - 2 * pz;
    const int64_t total = (int64_t)cx * cy * cz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % cz); int64_t r = i / cz; const int y = (int)(r % cy); const int x = (int)(r / cy);
        const float v = win[((size_t)(x + px) * ky + (y + py)) * kz + (z + pz)];
        const size_t o = ((size_t)(x0 + px + x) * Y + (y0 + py + y)) * Z + (z0 + pz + z);
        atomicAdd(&pred[o], v);       // overlapping windows may be in flight on two streams (inference lanes)
        atomicAdd(&cnt[o], 1.f);
    }
}
extern "C" int vg_overlap_add(const float* win, int kx, int ky, int kz, int px, int py, int pz, int x0, int y0, int z0,
                              int X, int Y, int Z, float* pred, float* cnt, vg_stream_t stream) {
    vg_begin();
    if (!win || !pred || !cnt || kx - 2 * px < 1 || ky - 2 * py < 1 || kz - 2 * pz < 1) return VG_EINVAL;
    if (x0 < 0 || y0 < 0 || z0 < 0 || x0 + kx > X || y0 + ky > Y || z0 + kz > Z) return VG_EINVAL;
    const int64_t total = (int64_t)(kx - 2 * px) * (ky - 2 * py) * (kz - 2 * pz);
    hipLaunchKernelGGL(overlap_add_kernel, dim3(lblocks(total)), dim3(256), 0, (hipStream_t)stream, win, kx, ky, kz, px, py, pz,
                       x0, y0, z0, Y, Z, pred, cnt);
    return vg_check_launch();
}
// out[x][y][z] = pred/cnt on the un-padded sub-box (np.true_divide: 0/0 -> NaN, as the reference)
__global__ void divide_crop_kernel(const float* pred, const float* cnt, int Y, int Z, int sx, int sy, int sz, int ox, int oy, int oz,
                                   float* out) {
    const int64_t total = (int64_t)ox * oy * oz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % oz); int64_t r = i / oz; const int y = (int)(r % oy); const int x = (int)(r / oy);
        const size_t o = ((size_t)(x + sx) * Y + (y + sy)) * Z + (z + sz);
        out[i] = pred[o] / cnt[o];
    }
}
extern "C" int vg_divide_crop(const float* pred, const float* cnt, int X, int Y, int Z, int sx, int sy, int sz, int ox, int oy,
                              int oz, float* out, vg_stream_t stream) {
    vg_begin();
    if (!pred || !cnt || !out || sx + ox > X || sy + oy > Y || sz + oz > Z || ox < 1 || oy < 1 || oz < 1) return VG_EINVAL;
    hipLaunchKernelGGL(divide_crop_kernel, dim3(lblocks((int64_t)ox * oy * oz)), dim3(256), 0, (hipStream_t)stream, pred, cnt, Y, Z,
                       sx, sy, sz, ox, oy, oz, out);
    return vg_check_launch();
}
