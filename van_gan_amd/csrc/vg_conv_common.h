// vg_conv_common.h -- definitions shared by the translation units of the gather-convolution: vg_conv.hip (conv_kernel,
// conv32_kernel, host-side tile choice and dispatch) and vg_conv_thin.hip (the 16-channel specialist).
#pragma once
#include "vg_gather.h"

struct ConvOut {
    int OD, OH, OW, ostr, ood, ooh, oow, BD, BH, BW, Cout;
    const void* wp; int Ktot, nchunks, kc_pad;
    const float* bias; const void* res; const float* rs; const float* rb; int tanh_out;
    int res1;           // vg_conv_desc::res_c1: res is a single-channel fp32 volume broadcast over the output channels
    int wdma;           // conv_thin_kernel, one-panel forward with several chunks: weight panels double-buffered by LDS-DMA
    const void* wp_up; int nup;   // conv_thin_kernel<..., UP>: class panels of the collapsed upsampled chunks (vg_conv_desc::wpacked_up), their number
    void* out; int out_f32, accumulate; float* sums;
    int w_lds;          // 1: the BN x Ktot weight panel of this workgroup is copied to LDS once (row stride WRS bytes)
    int dma;            // 1: LDS-DMA double-buffered staging (planar bf16 image, weights in LDS)
    int xw;             // conv_thin_kernel: XCD-aware tile walk (set by its launcher)
    int dbg;            // conv_thin_kernel, panel-loop instance: timing ablations (VG_THIN_DBG; 0 in production)
    int WRS;
    // IN-backward statistics of the output fused into the epilogue (conv_thin_kernel<..., BSTAT>; vg_conv_desc::bstat): the
    // pre-norm tensor(s) of the layer whose gradient this launch produces and its per-(sample, channel) constants
    const void* bs_x0; const void* bs_x1; int bs_c0, bs_sh, bs_act, bs_pad, bs_D, bs_H, bs_W;
    const float* bs_sc; const float* bs_sf; const float* bs_mu; const float* bs_rs; const float* bs_ml;
    // K split over workgroups (conv_kernel, small grids): slices per tile, fp32 partial tiles, arrival counters per (sample, panel, tile)
    int ks; float* ks_part; unsigned* ks_cnt;
    char* scratch; long scratch_bytes;          // vg_conv_desc::scratch (host-side planning only)
    VgFin fin;          // InstanceNorm finalisation of the output by the last workgroup (vg_conv_desc::fin); ticket == nullptr: off
};
// output-parity classes fused into one launch (data gradient of a strided conv): a separate kernel argument that only the
// multi-class kernel variants read
struct ConvCls {
    int ncls;
    int par;                        // 1: class-parallel launch (blockIdx.x = walker * ncls + class; any number of channel chunks)
    int tap0[9];                    // taps of class c: [tap0[c], tap0[c+1])
    const void* wp[8];              // packed weights of class c, ktot[c] elements per row
    int ktot[8], woff[8];           // ... and the byte offset of its panel inside the LDS weight area
    int ks0[9];                     // first K-step of class c in the koff table
    int off[8][3], it[8][3];        // output offset / number of outputs per axis of class c
};

// 4 consecutive channels as stored (epilogue operands)
template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> { bf16x4 v; };
template <> struct Vec4<float> { f32x4 v; };
__device__ __forceinline__ void vec4_load(Vec4<bf16_t>& r, const bf16_t* p) { r.v = *(const __attribute__((address_space(1))) bf16x4*)(uintptr_t)p; }
__device__ __forceinline__ void vec4_load(Vec4<float>& r, const float* p) { r.v = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)p; }
__device__ __forceinline__ void vec4_unpack(const Vec4<bf16_t>& r, float* o) { for (int j = 0; j < 4; ++j) o[j] = bf2f((bf16_t)r.v[j]); }
__device__ __forceinline__ void vec4_unpack(const Vec4<float>& r, float* o) { for (int j = 0; j < 4; ++j) o[j] = r.v[j]; }

#ifndef VG_CONV_MW2
#define VG_CONV_MW2 4      // sub-tiles per wave from which a variant is compiled for 2 waves per SIMD (256 VGPRs)
#endif
#ifndef VG_CONV_WAVES
#define VG_CONV_WAVES 3      // waves per SIMD the register allocation must allow (3 workgroups per CU)
#endif
template <typename T> using lds_ptr = const __attribute__((address_space(3))) T*;
template <typename T> using glb_ptr = const __attribute__((address_space(1))) T*;

// MFMA over the (tap, channel-group) pairs of one channel chunk.  w points at this lane's fragment of K-step 0 (LDS
// panel or global row: WP carries the address space).  bf16: K-steps are processed KU at a time, all operand fetches of
// a group issued before its first MFMA, the halo offsets of the NEXT group fetched meanwhile; the remainder steps run
// one by one afterwards so that no MFMA sits under a condition (conditional MFMAs made the compiler shuttle the
// accumulators between AGPRs and VGPRs around every group).
template <typename T, int MW, typename WP>
__device__ __forceinline__ void conv_mfma_chunk(f32x4 (&acc)[MW], WP w, const char* halo, const int (&rowbase)[MW],
                                                const int* tapoff, const int* koff, int ksteps, int ntaps, int CK, int CS, int lane) {
    if constexpr (sizeof(T) == 4) {
        // exact-parity mode: f32 operands, v_mfma_f32_16x16x4_f32 (k = 4 consecutive channels of one tap)
        int tap = 0, ch0 = 0;
        const int nk4 = (ntaps * CK) >> 2;
        for (int s = 0; s < nk4; ++s) {
            const int chn = ch0 + (lane >> 4);
            const int off = tapoff[tap] + (chn >> 3) * CS + (chn & 7) * 4;
            float b[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) b[i] = *(const float*)(halo + rowbase[i] + off);
            const float a = w[s * 4];
#pragma unroll
            for (int i = 0; i < MW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], acc[i], 0, 0, 0);
            ch0 += 4;
            if (ch0 >= CK) { ch0 = 0; ++tap; }
        }
    } else {
        // explicit two-stage software pipeline: while the MFMAs of K-step s run, the operand fragments of step s+1 are
        // already on their way from LDS (and the halo offset of step s+2 is being fetched); sched_barriers keep the
        // compiler from sinking the prefetch below the MFMAs again to save registers.
        typedef const __attribute__((address_space(3))) bf16x8 lds_frag;
        typedef const __attribute__((address_space(1))) bf16x8 glb_frag;
        auto wfrag = [&](int step) -> bf16x8 {
            if constexpr (__is_same(WP, lds_ptr<T>)) return *(lds_frag*)(w + step * 32);
            else return *(glb_frag*)(w + step * 32);
        };
        const int kg = lane >> 4;
        const int last = ksteps - 1;
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (!__is_same(WP, lds_ptr<T>)) {
            // weights straight from L2 (panels too big for LDS: the wide layers): a ring of four fragments, i.e. the
            // fetch for K-step s+4 is issued when step s has been multiplied -- one step of MFMAs (64-128 cycles) does not
            // cover an L2 round trip.  Halo fragments ping-pong between two sets as in the LDS-weight loop below.
            bf16x8 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = wfrag(min(u, last));
            bf16x8 bb[2][MW];
            {
                const int o0 = koff[kg];
#pragma unroll
                for (int i = 0; i < MW; ++i) bb[0][i] = *(const bf16x8*)(halo + rowbase[i] + o0);
            }
            int on1 = koff[min(1, last) * 4 + kg];
            for (int s = 0; s < ksteps; s += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int on2 = koff[min(s + u + 2, last) * 4 + kg];
#pragma unroll
                    for (int i = 0; i < MW; ++i) bb[(u + 1) & 1][i] = *(const bf16x8*)(halo + rowbase[i] + on1);
                    if (s + u >= ksteps) a[u] = zero8;                 // phantom steps of the last group add zero
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MW; ++i) acc[i] = VG_MFMA16(a[u], bb[u & 1][i], acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    a[u] = wfrag(min(s + u + 4, last));
                    on1 = on2;
                }
            }
            return;
        }
        bf16x8 a0 = wfrag(0), a1;
        bf16x8 b0[MW], b1[MW];
        {
            const int o0 = koff[kg];
#pragma unroll
            for (int i = 0; i < MW; ++i) b0[i] = *(const bf16x8*)(halo + rowbase[i] + o0);
        }
        int o1 = koff[min(1, last) * 4 + kg];
        for (int s = 0; s < ksteps; s += 2) {
            const int o2 = koff[min(s + 2, last) * 4 + kg];
            a1 = wfrag(min(s + 1, last));
#pragma unroll
            for (int i = 0; i < MW; ++i) b1[i] = *(const bf16x8*)(halo + rowbase[i] + o1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MW; ++i) acc[i] = VG_MFMA16(a0, b0[i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            o1 = koff[min(s + 3, last) * 4 + kg];
            a0 = wfrag(min(s + 2, last));
#pragma unroll
            for (int i = 0; i < MW; ++i) b0[i] = *(const bf16x8*)(halo + rowbase[i] + o2);
            if (s + 1 >= ksteps) a1 = zero8;                       // odd K-step count: the phantom step adds zero
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MW; ++i) acc[i] = VG_MFMA16(a1, b1[i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}


// vg_conv_thin.hip: specialist for the 16-channel-chunk 3x3x3 stride-1 layers with a fixed 16x8x4 tile
bool vg_conv_thin_ok(const vg_conv_desc* d, const GatherIn& g, const ConvOut& k, const ConvCls& q, int np);       // np: 16-channel panels per workgroup (1 / 2)
int vg_conv_thin_lds_bytes(const GatherIn& g, int np, int pl = 1, bool up = false, bool wdma = false);       // pl: output panels looped inside a workgroup (1 / 3); up: collapsed upsampled chunks
// red != NULL: accumulate the IN-backward statistics (ConvOut::bs_*) into red in the epilogue when the instance exists (did_stats)
int vg_launch_conv_thin(const GatherIn& g, const ConvOut& k, int np, hipStream_t s, float* red, bool& did_stats);       // VG_OK, < 0 on error, 1: not one of its combinations
