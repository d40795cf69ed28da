// vg_adam.hip -- multi-tensor Adam with per-tensor clip-by-norm over one flat fp32 buffer per network.
// Restates tf.keras.optimizers.Adam(2e-4, beta_1=0.5, beta_2=0.9, clipnorm=100) (vangan.py:220-235) as applied
// by optimizer.minimize (vangan.py:426-438): TP (TF 2.10 optimizer_v2) the gradient of EACH variable is
// clipped to norm<=clipnorm, then m,v are updated and w -= lr_t*m/(sqrt(v)+eps) with
// lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.
#include "vg_common.h"

// segment id of flat element i by binary search over seg_off[T+1]
__device__ __forceinline__ int find_seg(const int64_t* off, int T, int64_t i) {
    int lo = 0, hi = T;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// every block covers a contiguous span of 256*ELEMS elements; spans may straddle tensors
#define ADAM_ELEMS 8
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, const int64_t* off, int T, int64_t total, float scale, float* norms) {
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ADAM_ELEMS;
    if (base >= total) return;
    int seg = find_seg(off, T, base);
    float s = 0.f;
    for (int e = 0; e < ADAM_ELEMS; ++e) {
        const int64_t i = base + e;
        if (i >= total) break;
        if (i >= off[seg + 1]) { atomicAdd(&norms[seg], s); s = 0.f; while (i >= off[seg + 1]) ++seg; }
        const float v = g[i] * scale; s += v * v;
    }
    // wave-level combine when the whole wave sits in one segment
    const int seg0 = __shfl(seg, 0);
    const int64_t first = __shfl(base, 0);
    const bool uniform = __all(seg == seg0) && find_seg(off, T, first) == seg0;
    if (uniform) { s = wave_sum(s); if ((threadIdx.x & 63) == 0) atomicAdd(&norms[seg0], s); }
    else if (s != 0.f) atomicAdd(&norms[seg], s);
}
__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, const int64_t* off, int T,
                                                   int64_t total, const float* norms, float lr_t, float b1, float b2, float eps,
                                                   float clip, float scale) {
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ADAM_ELEMS;
    if (base >= total) return;
    int seg = find_seg(off, T, base);
    float nrm = sqrtf(norms[seg]);
    float f = (clip > 0.f && nrm > clip) ? clip / nrm : 1.f;
    for (int e = 0; e < ADAM_ELEMS; ++e) {
        const int64_t i = base + e;
        if (i >= total) break;
        if (i >= off[seg + 1]) { while (i >= off[seg + 1]) ++seg; nrm = sqrtf(norms[seg]); f = (clip > 0.f && nrm > clip) ? clip / nrm : 1.f; }
        const float gi = g[i] * scale * f;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
extern "C" int vg_adam_clip(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total,
                            float* norms, float lr_t, float beta1, float beta2, float eps, float clipnorm, float grad_scale,
                            vg_stream_t stream) {
    vg_begin();
    if (!w || !g || !m || !v || !seg_off_dev || !norms || T < 1 || total < 1) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(norms, 0, T * sizeof(float), s) != hipSuccess) return VG_ELAUNCH;
    const int blocks = (int)((total + 256 * ADAM_ELEMS - 1) / (256 * ADAM_ELEMS));
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, s, g, seg_off_dev, T, total, grad_scale, norms);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, w, g, m, v, seg_off_dev, T, total, norms, lr_t, beta1, beta2, eps,
                       clipnorm, grad_scale);
    return vg_check_launch();
}
