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

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

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

static inline int vg_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VG_OK : VG_ELAUNCH;
}
static inline int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
