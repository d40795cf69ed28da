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
__global__ __launch_bounds__(256) void skel_chain_kernel(const float* __restrict__ imgs, int64_t n, int iters, int D, int H, int W,
                                                         float* __restrict__ skels) {
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
        float p9[3];
#pragma unroll
        for (int sl = 0; sl < SK_TD + 2; ++sl) {
            const float a00 = t[sl][ty][tx], a01 = t[sl][ty][tx + 1], a02 = t[sl][ty][tx + 2];
            const float a10 = t[sl][ty + 1][tx], a11 = t[sl][ty + 1][tx + 1], a12 = t[sl][ty + 1][tx + 2];
            const float a20 = t[sl][ty + 2][tx], a21 = t[sl][ty + 2][tx + 1], a22 = t[sl][ty + 2][tx + 2];
            const float plus = fmaxf(fmaxf(fmaxf(a01, a21), fmaxf(a10, a12)), a11);
            p9[sl % 3] = fmaxf(plus, fmaxf(fmaxf(a00, a02), fmaxf(a20, a22)));
            if (sl >= 2) {
                const int k = sl - 2, gd = d0 + k;
                const float nb = fmaxf(p9[(sl - 1) % 3], fmaxf(p9[(sl - 2) % 3], p9[sl % 3]));
                const float delta = fmaxf(cen[k] - nb, 0.f);
                const float sp = sk[k];
                sk[k] = j ? sp + fmaxf(delta - sp * delta, 0.f) : delta;
                if (col_ok && gd < D) out[vol + ((size_t)gd * H + gh) * W + gw] = sk[k];
            }
        }
#pragma unroll
        for (int k = 0; k < SK_TD; ++k) cen[k] = t[k + 1][ty + 1][tx + 1];       // img_{j+1} is the next step's img_j
    }
}
static dim3 skel_grid(int B, int D, int H, int W) {
    return dim3(((W + SK_TW - 1) / SK_TW) * ((H + SK_TH - 1) / SK_TH) * ((D + SK_TD - 1) / SK_TD), B);
}
extern "C" int vg_soft_skel_fwd(const float* img, int B, int D, int H, int W, int iters, float* imgs, float* skels,
                                vg_stream_t stream) {
    vg_begin();
    if (!img || !imgs || !skels || B < 1 || iters < 0) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * D * H * W;
    const int blocks = lblocks(n);
    if (hipMemcpyAsync(imgs, img, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VG_ELAUNCH;
    (void)blocks;
    const dim3 grid = skel_grid(B, D, H, W);
    for (int j = 0; j <= iters; ++j)
        hipLaunchKernelGGL(skel_tile_kernel<true>, grid, dim3(256), 0, s, imgs + j * n, (const float*)nullptr, (const float*)nullptr,
                           D, H, W, imgs + (j + 1) * n);
    if (vg_tune("SKEL_CHAIN", 1)) {
        hipLaunchKernelGGL(skel_chain_kernel, grid, dim3(256), 0, s, (const float*)imgs, n, iters, D, H, W, skels);
        return vg_check_launch();
    }
    for (int j = 0; j <= iters; ++j)
        hipLaunchKernelGGL(skel_tile_kernel<false>, grid, dim3(256), 0, s, imgs + (j + 1) * n, imgs + j * n,
                           j ? skels + (j - 1) * n : (const float*)nullptr, D, H, W, skels + j * n);
    return vg_check_launch();
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
#define SK_TILE_ORIGIN()                                                                                    \
    const int tiles_w = (W + SK_TW - 1) / SK_TW, tiles_h = (H + SK_TH - 1) / SK_TH;                           \
    int bt_ = blockIdx.x; const int tw_ = bt_ % tiles_w; bt_ /= tiles_w;                                      \
    const int th_ = bt_ % tiles_h, td_ = bt_ / tiles_h;                                                       \
    const int w0 = tw_ * SK_TW, h0 = th_ * SK_TH, d0 = td_ * SK_TD;                                           \
    const size_t vol = (size_t)blockIdx.y * D * H * W;                                                        \
    const int tx = tid & (SK_TW - 1), ty = tid >> 5; const int gw = w0 + tx, gh = h0 + ty;

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
    for (int z = 0; z < SK_TD; ++z) {
        const int gd = d0 + z;
        if (gd >= D) break;
        const size_t i = vol + ((size_t)gd * H + gh) * W + gw;
        // dilation value and first arg-max, scan order a (D), b (H), c (W) ascending
        float dil = -INFINITY; int o = 0;
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int b = -1; b <= 1; ++b)
#pragma unroll
                for (int c = -1; c <= 1; ++c) {
                    const float v = t[z + 1 + a][ty + 1 + b][tx + 1 + c];
                    if (v > dil) { dil = v; o = (a * H + b) * W + c; }
                }
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
        if (e != 0.f) { dimgj[i] += e; atomicAdd(&dimgj1[(int64_t)i + o], -e); }
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
    for (int z = 0; z < SK_TD; ++z) {
        const int gd = d0 + z;
        if (gd >= D) break;
        const size_t i = vol + ((size_t)gd * H + gh) * W + gw;
        const float g = dimgj1[i];
        if (g == 0.f) continue;
        dimgj1[i] = 0.f;
        float best = INFINITY; int o = 0;
#define SK_ER(a, b, c) { const float v = t[z + 1 + (a)][ty + 1 + (b)][tx + 1 + (c)]; if (v < best) { best = v; o = ((a) * H + (b)) * W + (c); } }
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int b = -1; b <= 1; ++b) SK_ER(a, b, 0)
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int c = -1; c <= 1; ++c) SK_ER(a, 0, c)
#pragma unroll
        for (int b = -1; b <= 1; ++b)
#pragma unroll
            for (int c = -1; c <= 1; ++c) SK_ER(0, b, c)
#undef SK_ER
        atomicAdd(&dimgj[(int64_t)i + o], g);
    }
}
extern "C" int vg_soft_skel_bwd(const float* imgs, const float* skels, const float* gskel, int B, int D, int H, int W,
                                int iters, float* work, float* gimg, vg_stream_t stream) {
    vg_begin();
    if (!imgs || !skels || !gskel || !work || !gimg || B < 1 || iters < 0) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * D * H * W;
    const int blocks = lblocks(n);
    (void)blocks;
    const dim3 grid = skel_grid(B, D, H, W);
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
