// vg_dma_common.h -- shared by the LDS-DMA kernels (vg_wgrad_dma.hip: weight gradients; vg_conv_dma.hip: forward / data gradient):
// the materialised operand P and its elementwise producer, the LDS-DMA copy and the counted vmcnt wait.
#pragma once
#include "vg_gather.h"

typedef __attribute__((address_space(3))) void lds_void_d;
typedef const __attribute__((address_space(1))) void glb_void_d;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_d;

// ------------------------------------------------------------------------------------------------------------------
// operand materialisation
// ------------------------------------------------------------------------------------------------------------------
struct MatK {
    const void* src0; const void* src1; int c0, c1, shift0;
    int N, D, H, W, Cin;
    const float* in_scale; const float* in_shift; int act;
    const bf16_t* noise; int npad;
    int pad_mode;
    int pmin_d, pmin_h, pmin_w;      // input position of padded index 0
    int Dp, Hp, Wp;                  // padded extents (positions)
    int deint, WE, Wps;              // W stored de-interleaved: WE even positions first; Wps = stored row length (voxels)
    bf16_t* out;
};

// launches materialize_kernel (vg_wgrad_dma.hip) on s
void vg_launch_materialize(const MatK& m, hipStream_t s);

// n / d for 0 <= n < 2^32 / d by one multiply-high, m = floor(2^32 / d) + 1 from the host (the integer divisions of the
// prologue -- 13 DMA pieces x 3 divisions x ~40 instructions -- were 6 us of every workgroup's life)
__device__ __forceinline__ int fast_div(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }       // m == 0: d == 1

// One LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS bytes [lds_addr, lds_addr + 1024).  Inline asm on
// purpose: for the builtin hipcc tracks the copy as a pending LDS write and drains it (s_waitcnt vmcnt(0)) in front of the next
// ds_read -- the copy of tile t+1 must stay in flight under the MFMA loop of tile t.  The kernel counts it itself (vmcnt(0)
// at the top of the tile loop, where nothing else is outstanding).  M0 = LDS base of the piece, restored afterwards.
__device__ __forceinline__ void glds16(const char* sbase, int voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): all but this wave's n youngest copies
// have landed.  n beyond the table waits for more than asked (a smaller count is always safe).  A computed jump into a table of
// (s_waitcnt vmcnt(i); s_branch end) pairs, 8 bytes each: hipcc lowers a 41-way switch over inline-asm cases into a chain of
// ~100 scalar compare / branch instructions, which the LDS-DMA kernels paid once per stage.
#define VG_VM1(i) "s_waitcnt vmcnt(" #i ")\n\ts_branch 2f\n\t"
#define VG_VM8(a, b, c, d, e, f, g, h) VG_VM1(a) VG_VM1(b) VG_VM1(c) VG_VM1(d) VG_VM1(e) VG_VM1(f) VG_VM1(g) VG_VM1(h)
__device__ __forceinline__ void wait_vmcnt(int n_) {
    const int n = __builtin_amdgcn_readfirstlane(n_);          // wave-uniform by contract; pins it to a scalar register
    asm volatile(
        "s_min_u32 s42, %0, 40\n\t"
        "s_lshl_b32 s42, s42, 3\n\t"
        "s_getpc_b64 s[40:41]\n"
        "0:\n\t"
        "s_add_u32 s40, s40, s42\n\t"
        "s_addc_u32 s41, s41, 0\n\t"
        "s_add_u32 s40, s40, 1f-0b\n\t"
        "s_addc_u32 s41, s41, 0\n\t"
        "s_setpc_b64 s[40:41]\n"
        "1:\n\t"
        VG_VM8(0, 1, 2, 3, 4, 5, 6, 7) VG_VM8(8, 9, 10, 11, 12, 13, 14, 15) VG_VM8(16, 17, 18, 19, 20, 21, 22, 23)
        VG_VM8(24, 25, 26, 27, 28, 29, 30, 31) VG_VM8(32, 33, 34, 35, 36, 37, 38, 39) VG_VM1(40)
        "2:\n\t"
        : : "s"(n) : "s40", "s41", "s42", "scc", "memory");
}

// Kernel of vg_pack_weights_dma / the bn > 0 items of vg_pack_weights_multi: fp32 DHWIO [T][Cin][Cout] -> Wd (see the header).
// transpose 0: rows = output channels, contraction = input channels (forward); 1: rows = input channels, contraction = output
// channels (data gradient).  One thread per 16-byte unit (8 contraction channels of one row).
__device__ __forceinline__ void vg_pack_dma_units(const float* __restrict__ w, const int* __restrict__ tap_idx, bf16_t* __restrict__ out, int Cin, int Cout,
                                  int ntaps, int transpose, int bn, int u0, int ustride) {
    const int NR = transpose ? Cin : Cout, C = transpose ? Cout : Cin;
    const int npl = C >> 4;
    // 32-bit index arithmetic (units = NR * npl * ntaps * 2 <= 2^22 for the widest layer; the 64-bit divisions of the first version
    // were most of this kernel's time) and, for the data-gradient layout, the unit's 8 contraction channels as two 16-byte loads
    const unsigned units = (unsigned)NR * (unsigned)npl * (unsigned)ntaps * 2u;
    const bool vec = transpose && (((uintptr_t)w & 15) == 0) && (Cout % 8) == 0;
    for (unsigned u = (unsigned)u0; u < units; u += (unsigned)ustride) {
        unsigned t = u;
        const int row = (int)(t % (unsigned)bn); t /= (unsigned)bn;
        const int half = (int)(t & 1u); t >>= 1;
        const int tap = (int)(t % (unsigned)ntaps); t /= (unsigned)ntaps;
        const int plane = (int)(t % (unsigned)npl); const int cob = (int)(t / (unsigned)npl);
        const int r = cob * bn + row, c0 = plane * 16 + half * 8, ts = tap_idx[tap];
        float v[8];
        if (vec) load8<float>(w + ((size_t)ts * Cin + r) * Cout + c0, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = transpose ? w[((size_t)ts * Cin + r) * Cout + c0 + e] : w[((size_t)ts * Cin + c0 + e) * Cout + r];
        }
        store8<bf16_t>(out + (size_t)u * 8, v);
    }
}
