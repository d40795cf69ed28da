// vg_common.h -- shared device helpers for the gfx950 kernels of libvangan_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vangan_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 16 B = one MFMA operand fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

#define VG_LDS_LIMIT 163840
#define VG_LRELU 0.2f

// The 16-bit storage format is a property of the BUILD: libvangan_hip.so stores bf16 (training and inference), the same sources
// compiled with -DVG_FP16 give libvangan_hip_h.so, whose 16-bit buffers hold IEEE half precision (BASELINE config 5: fp16
// sliding-window inference, post_training.py:38-39).  Only these two conversions and the MFMA opcode differ; "bf16" in type and
// kernel names then reads "the build's 16-bit format".
#ifdef VG_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 vg_h8;
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }      // v_cvt_f16_f32: RNE
#define VG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vg_h8, a), __builtin_bit_cast(vg_h8, b), c, 0, 0, 0)
#define VG_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vg_h8, a), __builtin_bit_cast(vg_h8, b), c, 0, 0, 0)
#define VG_STORAGE16 "fp16"
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
#define VG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define VG_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VG_STORAGE16 "bf16"
#endif
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

// ---- storage-type helpers: activations/gradients/weights are bf16 (product) or f32 (exact-parity mode) ----
template <typename T> __device__ __forceinline__ void load8(const T* p, float* o);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* o) {
    const bf16x8 r = *(const bf16x8*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)r[j]);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* o) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(v[j]);
    *(bf16x8*)p = r;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <typename T> __device__ __forceinline__ float rnd(float v);          // value as it will be stored
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bfround(v); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

__device__ __forceinline__ float vg_act(float x, int act) {
    if (act == VG_ACT_RELU) return fmaxf(x, 0.f);
    if (act == VG_ACT_LRELU) return x > 0.f ? x : VG_LRELU * x;
    return x;
}
__device__ __forceinline__ float vg_act_grad(float pre, int act) {
    if (act == VG_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    if (act == VG_ACT_LRELU) return pre > 0.f ? 1.f : VG_LRELU;   // TP: LeakyRelu grad uses x>0
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- InstanceNorm finalisation as the tail of the producing launch (vg_fin_desc, include/vangan_hip.h) ----
struct VgFin { unsigned* ticket; float count, eps; int njobs; vg_fin_job job[2]; };
static inline VgFin vg_fin_of(const vg_conv_desc* d) {
    VgFin f; f.ticket = nullptr; f.count = 1.f; f.eps = 0.f; f.njobs = 0;
    if (d->fin && d->out_sums) { f.ticket = d->fin->ticket; f.count = d->fin->count; f.eps = d->fin->eps; f.njobs = d->fin->njobs;
                                 f.job[0] = d->fin->job[0]; f.job[1] = d->fin->job[d->fin->njobs > 1 ? 1 : 0]; }
    return f;
}
// scale / shift / mean / rstd of sample n, channel c from the 8 striped (sum, sum of squares) pairs: in_finalize_kernel's arithmetic
__device__ __forceinline__ void vg_fin_one(const VgFin& f, float s, float ss, int n, int c) {
    const float mean = s / f.count;
    float var = ss / f.count - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + f.eps);
    for (int j = 0; j < f.njobs; ++j) {
        const vg_fin_job& q = f.job[j];
        const int cc = q.c_off + c, i = n * q.c_tot + cc;
        float sc = (q.gamma ? q.gamma[cc] : 1.f) * rstd;
        float sh = (q.beta ? q.beta[cc] : 0.f) - mean * sc;
        if (q.mult) { const float m = q.mult[i]; sc *= m; sh *= m; }
        q.scale[i] = sc; q.shift[i] = sh;
        if (q.mean) q.mean[i] = mean;
        if (q.rstd) q.rstd[i] = rstd;
    }
}
// Called by EVERY workgroup of the launch after its last atomicAdd into the striped sums.  The adds are device-scope atomics; a
// workgroup takes its ticket once its own have been acknowledged (s_waitcnt vmcnt(0) per thread, then the workgroup barrier), so the
// workgroup that draws the last ticket finds every contribution at the level the XCDs share and reads it there (agent-scope loads) --
// the exchange of the K split, no __threadfence().
// flag: one word of the workgroup's LDS that nobody uses any more (no static __shared__ here: the kernels raise their dynamic LDS limit
// to all 160 KiB, which a single static byte makes an invalid request).
__device__ __forceinline__ void vg_fin_tail(const VgFin& f, const float* sums, int N, int C, unsigned nblocks, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *flag = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u ? 1 : 0;
    __syncthreads();
    if (!*flag) return;
    for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
        const int n = i / C, c = i - n * C;
        float s = 0.f, ss = 0.f;
        for (int t = 0; t < VG_STRIPES; ++t) {
            const float* p = sums + (((size_t)t * N + n) * C + c) * 2;
            s += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ss += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        vg_fin_one(f, s, ss, n, c);
    }
}
// the same as a launch of its own, for the kernel families without the tail (vg_conv.hip: vg_conv3d runs it right behind them)
int vg_launch_fin(const vg_conv_desc* d, hipStream_t s);
extern thread_local bool vg_fin_done;      // set by a launch function whose kernel carries the tail

// ---- tuning knobs: vg_tune("CONV_BN", 0) reads the override set by vg_set_tuning, else the environment variable VG_CONV_BN
// (once), else the default.  One registry for every host-side heuristic switch, so that tests can force kernel variants in
// process (vg_set_tuning) and sweeps can use the environment.
int vg_tune(const char* key, int dflt);
// ---- dry run (vg_conv3d_variant / vg_conv3d_wgrad_variant): the dispatch code runs unchanged, and at the launch site the
// name of the kernel variant it selected is recorded instead of launching.  Returns true when recording.
bool vg_dry(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void vg_dry_begin(char* buf, int n);
void vg_dry_end();

bool vg_dry_on();
// a sticky error left by an earlier, unrelated HIP call in this thread must not be blamed on our launch
static inline void vg_begin() { (void)hipGetLastError(); }
static inline int vg_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VG_OK : VG_ELAUNCH;
}
// vg_pointwise.hip: 1x1x1 convolutions with one channel on one side (HBM-bound VALU kernels).  Return VG_OK when the call
// was served, 1 when the shape is not one of theirs (the caller continues on the MFMA path), < 0 on error.
int vg_pointwise_conv(const vg_conv_desc* d, hipStream_t s);
int vg_pointwise_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch,
                       int64_t scratch_bytes, hipStream_t s);
// vg_wgrad_dma.hip: weight gradient from a materialised operand (VG_OK served, 1 not one of its shapes, < 0 error)
int vg_wgrad_dma(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host, int T_total, float* dw, float* db,
                 float* scratch, int64_t scratch_bytes, hipStream_t s);
int vg_wgrad_pw_dma(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch,
                    int64_t scratch_bytes, hipStream_t s);
// vg_conv_thin.hip: weight gradient of the thin full-resolution layers (VG_OK served, 1 not one of its shapes, < 0 error)
int vg_wgrad_thin(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host, int T_total, float* dw, float* db,
                  float* scratch, int64_t scratch_bytes, hipStream_t s);
// vg_conv_dma.hip: forward / data gradient with both operands staged by LDS-DMA (VG_OK served, 1 not one of its shapes, < 0 error)
int vg_conv_dma(const vg_conv_desc* d, hipStream_t s, bool* did_stats = nullptr);
// vg_wgrad.hip: dw[i] += sum_b part[b][i] in a fixed order (partial slabs written by weight-gradient workgroups)
void vg_launch_reduce_partials(const float* part, int nb, int n, float* dw, hipStream_t s);
static inline int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// raw 8-channel vector as loaded from global memory
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { bf16x8 v; };
template <> struct Raw8<float> { f32x4 a, b; };
// global address space stated explicitly: a source pointer selected per lane (virtual concat) would otherwise be generic
// and the load a flat_load (slower, and it ties up the LDS counter as well)
__device__ __forceinline__ void raw_load(Raw8<bf16_t>& r, const bf16_t* p) { r.v = *(const __attribute__((address_space(1))) bf16x8*)(uintptr_t)p; }
__device__ __forceinline__ void raw_load(Raw8<float>& r, const float* p) {
    const __attribute__((address_space(1))) f32x4* q = (const __attribute__((address_space(1))) f32x4*)(uintptr_t)p;
    r.a = q[0]; r.b = q[1];
}
__device__ __forceinline__ void raw_unpack(const Raw8<bf16_t>& r, float* o) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)r.v[j]);
}
__device__ __forceinline__ void raw_unpack(const Raw8<float>& r, float* o) {
    o[0] = r.a[0]; o[1] = r.a[1]; o[2] = r.a[2]; o[3] = r.a[3]; o[4] = r.b[0]; o[5] = r.b[1]; o[6] = r.b[2]; o[7] = r.b[3];
}


// opaque use + redefinition: the value must exist at this point of the program (no sinking of its load below it)
__device__ __forceinline__ void raw_pin(Raw8<bf16_t>& r) { asm volatile("" : "+v"(r.v)); }
__device__ __forceinline__ void raw_pin(Raw8<float>& r) { asm volatile("" : "+v"(r.a), "+v"(r.b)); }
__device__ __forceinline__ void raw_mask(Raw8<bf16_t>& r, bool keep) { if (!keep) r.v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ void raw_mask(Raw8<float>& r, bool keep) { if (!keep) { r.a = (f32x4){0.f, 0.f, 0.f, 0.f}; r.b = r.a; } }

__device__ __forceinline__ float ld_global(const float* p) { return *(const __attribute__((address_space(1))) float*)(uintptr_t)p; }
__device__ __forceinline__ float ld_global(const bf16_t* p) { return bf2f(*(const __attribute__((address_space(1))) bf16_t*)(uintptr_t)p); }

