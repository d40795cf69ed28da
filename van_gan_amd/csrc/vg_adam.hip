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

// Every block owns a contiguous span of ADAM_SPAN elements of the flat buffer.  A span inside one tensor (all but ~one per
// tensor) is streamed with 16-byte loads/stores, consecutive lanes on consecutive float4s; a span that straddles tensors
// walks its tensors one after the other.
//
// The squared norms are summed in a FIXED order (no float atomics): data-parallel replicas apply the clip factor to bit-identical
// reduced buckets and must come out with bit-identical weights, or they drift apart step by step.  A tensor that lies inside one
// span is summed by that block straight into norms[seg]; a tensor spread over several spans leaves one partial per block in
// bp[2 * block + slot] (slot 0: the piece starts at the block's first element, slot 1: it starts inside the block -- only a tensor's
// first piece can) and adam_kernel adds the partials of its tensor in block order.
#define ADAM_SPAN 4096
typedef const __attribute__((address_space(1))) f32x4* gp_f32x4;

__device__ __forceinline__ float block_sum(float s, float* red) {
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, const int64_t* off, int T, int64_t total, float scale, float* norms, float* bp) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * ADAM_SPAN;
    const int64_t b1 = b0 + ADAM_SPAN < total ? b0 + ADAM_SPAN : total;
    int seg = find_seg(off, T, b0);
    if (tid < 2) bp[2 * blockIdx.x + tid] = 0.f;
    if (off[seg + 1] >= b1 && b1 - b0 == ADAM_SPAN) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ADAM_SPAN / 1024; ++j) {
            const f32x4 v = *(gp_f32x4)(uintptr_t)(g + b0 + (int64_t)(j * 256 + tid) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = v[e] * scale; s += x * x; }
        }
        s = block_sum(s, red);
        if (tid == 0) {
            if (off[seg] == b0 && off[seg + 1] == b1) norms[seg] = s;          // the tensor is exactly this span
            else bp[2 * blockIdx.x] = s;
        }
        return;
    }
    for (; seg < T && off[seg] < b1; ++seg) {                      // block-uniform loop over the tensors of the span
        const int64_t lo = off[seg] > b0 ? off[seg] : b0, hi = off[seg + 1] < b1 ? off[seg + 1] : b1;
        float s = 0.f;
        for (int64_t i = lo + tid; i < hi; i += 256) { const float x = g[i] * scale; s += x * x; }
        s = block_sum(s, red);
        if (tid == 0 && hi > lo) {
            if (off[seg] >= b0 && off[seg + 1] <= b1) norms[seg] = s;          // whole tensor inside the span
            else bp[2 * blockIdx.x + (lo == b0 ? 0 : 1)] = s;
        }
    }
}

// squared norm of tensor seg: norms[seg] when one block summed it, else its per-block partials in block order (block-uniform call)
__device__ __forceinline__ float seg_sqnorm(float* norms, const float* bp, const int64_t* off, int seg, float* red) {
    const int64_t lo = off[seg], hi = off[seg + 1];
    const int64_t bl = lo / ADAM_SPAN, bh = (hi - 1) / ADAM_SPAN;
    if (bl == bh) return norms[seg];
    float s = 0.f;
    for (int64_t b = bl + threadIdx.x; b <= bh; b += 256) s += bp[2 * b + ((b == bl && lo != bl * ADAM_SPAN) ? 1 : 0)];
    s = block_sum(s, red);
    if (threadIdx.x == 0 && (int64_t)blockIdx.x == bl) norms[seg] = s;      // for the caller's diagnostics; nobody reads it back
    return s;
}

__device__ __forceinline__ void adam_elem(float& w, float g, float& m, float& v, float f, float lr_t, float b1, float b2, float eps) {
    const float gi = g * f;
    m = b1 * m + (1.f - b1) * gi;
    v = b2 * v + (1.f - b2) * gi * gi;
    w -= lr_t * m / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, const int64_t* off, int T,
                                                   int64_t total, float* norms, const float* bp, float lr_t, float b1, float b2, float eps,
                                                   float clip, float scale, const float* lr_dev) {
    __shared__ float red[4];
    if (lr_dev) lr_t = *lr_dev;                    // the bias-corrected rate of THIS step from device memory (replayable graphs)
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * ADAM_SPAN;
    const int64_t s1 = s0 + ADAM_SPAN < total ? s0 + ADAM_SPAN : total;
    int seg = find_seg(off, T, s0);
    if (off[seg + 1] >= s1 && s1 - s0 == ADAM_SPAN) {
        const float nrm = sqrtf(seg_sqnorm(norms, bp, off, seg, red));
        const float f = ((clip > 0.f && nrm > clip) ? clip / nrm : 1.f);
        f32x4 gv[ADAM_SPAN / 1024], mv[ADAM_SPAN / 1024], vv[ADAM_SPAN / 1024], wv[ADAM_SPAN / 1024];
#pragma unroll
        for (int j = 0; j < ADAM_SPAN / 1024; ++j) {
            const int64_t i = s0 + (int64_t)(j * 256 + tid) * 4;
            gv[j] = *(gp_f32x4)(uintptr_t)(g + i); mv[j] = *(gp_f32x4)(uintptr_t)(m + i);
            vv[j] = *(gp_f32x4)(uintptr_t)(v + i); wv[j] = *(gp_f32x4)(uintptr_t)(w + i);
        }
#pragma unroll
        for (int j = 0; j < ADAM_SPAN / 1024; ++j) {
            const int64_t i = s0 + (int64_t)(j * 256 + tid) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float we = wv[j][e], me = mv[j][e], ve = vv[j][e];
                adam_elem(we, gv[j][e] * scale, me, ve, f, lr_t, b1, b2, eps);
                wv[j][e] = we; mv[j][e] = me; vv[j][e] = ve;
            }
            *(f32x4*)(m + i) = mv[j]; *(f32x4*)(v + i) = vv[j]; *(f32x4*)(w + i) = wv[j];
        }
        return;
    }
    for (; seg < T && off[seg] < s1; ++seg) {
        const int64_t lo = off[seg] > s0 ? off[seg] : s0, hi = off[seg + 1] < s1 ? off[seg + 1] : s1;
        const float nrm = sqrtf(seg_sqnorm(norms, bp, off, seg, red));
        const float f = ((clip > 0.f && nrm > clip) ? clip / nrm : 1.f);
        for (int64_t i = lo + tid; i < hi; i += 256) adam_elem(w[i], g[i] * scale, m[i], v[i], f, lr_t, b1, b2, eps);
    }
}
static int adam_launch(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total, float* norms, float lr_t,
                       const float* lr_dev, float beta1, float beta2, float eps, float clipnorm, float grad_scale, vg_stream_t stream);
extern "C" int vg_adam_clip(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total,
                            float* norms, float lr_t, float beta1, float beta2, float eps, float clipnorm, float grad_scale,
                            vg_stream_t stream) {
    return adam_launch(w, g, m, v, seg_off_dev, T, total, norms, lr_t, nullptr, beta1, beta2, eps, clipnorm, grad_scale, stream);
}
extern "C" int vg_adam_clip_dev(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total,
                                float* norms, const float* lr_t_dev, float beta1, float beta2, float eps, float clipnorm, float grad_scale,
                                vg_stream_t stream) {
    if (!lr_t_dev) return VG_EINVAL;
    return adam_launch(w, g, m, v, seg_off_dev, T, total, norms, 0.f, lr_t_dev, beta1, beta2, eps, clipnorm, grad_scale, stream);
}
static int adam_launch(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total, float* norms, float lr_t,
                       const float* lr_dev, float beta1, float beta2, float eps, float clipnorm, float grad_scale, vg_stream_t stream) {
    vg_begin();
    if (!w || !g || !m || !v || !seg_off_dev || !norms || T < 1 || total < 1) return VG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((total + ADAM_SPAN - 1) / ADAM_SPAN);
    float* bp = norms + T;                       // [blocks][2] per-block partials of the tensors that span several blocks
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, s, g, seg_off_dev, T, total, grad_scale, norms, bp);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, w, g, m, v, seg_off_dev, T, total, norms, bp, lr_t, beta1, beta2, eps,
                       clipnorm, grad_scale, lr_dev);
    return vg_check_launch();
}

// ---- stand-in for the gradient all-reduce on a box with ONE GPU (vg_local_exchange, include/vangan_hip.h) ----
// A workgroup owns a contiguous slice (as an RCCL channel does) and moves it twice: buf -> scratch, then scratch -> buf; the second
// pass re-reads what the same lanes wrote, so no workgroup depends on another and buf ends bit-identical.  Afterwards every workgroup
// holds its CU until min_ticks of the 100 MHz wall clock have passed since it started: the time a ring over xGMI would occupy the
// communication stream and its channels' CUs for a message of that size.
__global__ __launch_bounds__(256) void local_exchange_kernel(float* buf, float* scratch, int64_t n4, int64_t per_wg, long long min_ticks) {
    const long long t0 = wall_clock64();
    const int64_t lo = (int64_t)blockIdx.x * per_wg, hi = lo + per_wg < n4 ? lo + per_wg : n4;
    f32x4* b4 = (f32x4*)buf; f32x4* s4 = (f32x4*)scratch;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s4[i] = b4[i];
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) b4[i] = s4[i];
    while (wall_clock64() - t0 < min_ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int vg_local_exchange(float* buf, float* scratch, int64_t n, int workgroups, int min_us, vg_stream_t stream) {
    vg_begin();
    if (!buf || !scratch || n < 0 || (n & 3) || workgroups < 1 || workgroups > 1024 || min_us < 0 || min_us > 100000
        || (((uintptr_t)buf | (uintptr_t)scratch) & 15)) return VG_EINVAL;
    if (n == 0) return VG_OK;
    const int64_t n4 = n / 4, per_wg = (n4 + workgroups - 1) / workgroups;
    hipLaunchKernelGGL(local_exchange_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, buf, scratch, n4, per_wg,
                       (long long)min_us * 100);
    return vg_check_launch();
}
