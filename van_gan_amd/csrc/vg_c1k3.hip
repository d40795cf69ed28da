// vg_c1k3.hip -- the single-channel 3x3x3 stem convolution (1 -> 16, resunet_model.py:44-60) and its weight gradient on the matrix
// pipe (gfx950).  16-bit storage builds only; the VALU kernels of vg_pointwise.hip keep the exact-parity mode and other widths.
//
// Why: the layer is 4 + 32 bytes of traffic per voxel next to 27 x 16 MACs.  As VALU work (c1k3_fwd_kernel: a thread owns 4 voxels x
// 8 channels, 864 FMAs + ~400 instructions of window loads and transforms per iteration) it ran at 120 us per 128^3 volume where
// the 67 MB it writes take 15; the weight gradient at 153 us for two volumes (134 MB of dY: 27 us).  Both are chip-filling
// launches at the head / tail of every generator application.
//
// How: the 3 taps along W become 4 K-slots of a GEMM (the 4th multiplies a zero weight), the 9 (d, h) taps its rows:
//   k = 4 * (3 * a + b) + j,   a, b = tap offset along D, H,   j = 0..3 along W            -- 36 slots.
// The LDS image of a 16 x 8 x 4 output tile holds, per (halo plane, halo row, output column w), ONE 8-byte entry
// {x[w-1], x[w], x[w+1], x[w+2]} of transformed, rounded source values: slot group (a, b) of output voxel (z, y, w) is the entry at
// (z + a, y + b, w) -- an aligned ds_read_b64, no gather, 8 bytes of LDS per 4 K-slots.
//   forward   D[co][voxel] = W[co][k] * X[k][voxel]: two v_mfma_f32_16x16x32 per 16 voxels (k 0..31, 32..35 + zeros), the B fragment of a
//             lane = entries of rows 2 kg, 2 kg + 1 (then row 8); epilogue as conv_thin's (bias, 16-byte stores after a row swap,
//             InstanceNorm statistics of the stored values, finalisation by the last workgroup).
//   gradient  dW[k][co] = sum over voxels X[k][voxel] * dY[voxel][co]: voxels are the K index, so both operands are needed
//             voxel-major per lane while the image (slot-major per voxel) and dY ([voxel][16 channels]) are stored the other way round:
//             ds_read_b64_tr_b16 hands a 16-lane group a 4 (voxels) x 16 (slots | channels) block transposed, and since every lane
//             supplies its own row address, the "row" of the A block is assembled on the fly from the entries of 4 different (a, b)
//             rows -- no im2col image.  Three M tiles (rows 0-3, 4-7, 8 + a row of ones whose product is the bias gradient).
#include "vg_c1k3.h"

namespace {
constexpr int TW = 16, TH = 8, TD = 4, HH = TH + 2, HD = TD + 2;
constexpr int ROWB = TW * 8 + 8;                       // bytes of one image row: 16 entries + 8 (bank spread of the staging writes)
constexpr int IMGB = HD * HH * ROWB;                   // 8160
constexpr int DYROW = TW * 32;                         // bytes of one dY row of the tile (16 voxels x 16 channels)
constexpr int DYB = TD * TH * DYROW;                   // 16384

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float c1_slope(int act) { return act == VG_ACT_RELU ? 0.f : (act == VG_ACT_LRELU ? VG_LRELU : 1.f); }

// the image of the tile with origin (od0, oh0, ow0): thread = (halo row 0..59, quarter of the 16 columns); 7 source values give 4 entries
template <typename S>
__device__ __forceinline__ void c1_stage_image(const C1K3& p, char* img, int n, int od0, int oh0, int ow0, float sc, float sf, float slope, int tid) {
    const int row = tid >> 2, q = tid & 3;
    if (row >= HD * HH) return;
    const bool refl = p.pad_mode == VG_PAD_REFLECT;
    const int zz = row / HH, yy = row - zz * HH;
    bool okd, okh;
    const int rd = c1_resolve(od0 + p.td0 + zz, p.D, refl, okd), rh = c1_resolve(oh0 + p.th0 + yy, p.H, refl, okh);
    const S* src = (const S*)p.x + (((int64_t)n * p.D + rd) * p.H + rh) * p.W;
    float v[7]; bool ok[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int cw = c1_resolve(ow0 + p.tw0 + 4 * q + i, p.W, refl, ok[i]); v[i] = ld_global(src + cw); }
    unsigned short h[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float y = v[i] * sc + sf;
        y = fmaxf(y, y * slope);
        h[i] = (okd && okh && ok[i]) ? f2bf(y) : (unsigned short)0;
    }
    char* dst = img + row * ROWB + q * 32;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        u32x4 o;
        o[0] = h[e] | ((unsigned)h[e + 1] << 16); o[1] = h[e + 2] | ((unsigned)h[e + 3] << 16);
        o[2] = h[e + 1] | ((unsigned)h[e + 2] << 16); o[3] = h[e + 3] | ((unsigned)h[e + 4] << 16);
        *(u32x2*)(dst + e * 8) = (u32x2){o[0], o[1]};
        *(u32x2*)(dst + e * 8 + 8) = (u32x2){o[2], o[3]};
    }
}

// the weights of K-step s as the lane's 16x16x32 fragment: row li (output channel), k = 32 s + 8 kg + i -> (tap row, slot)
__device__ __forceinline__ bf16x8 c1_weight_frag(const C1K3& p, int li, int kg, int s) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 32 * s + 8 * kg + i, t = k >> 2, j = k & 3;
        r[i] = (t < 9 && j < 3) ? (short)((const bf16_t*)p.w)[(size_t)li * p.Ktot + t * p.CK + j] : (short)0;
    }
    return r;
}

template <typename S>
__global__ __launch_bounds__(256, 2) void c1k3m_fwd_kernel(const C1K3 p) {
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) char img[IMGB];
    __shared__ float stat[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    const float sc = p.scale ? p.scale[n] : p.sc, sf = p.scale ? p.shift[n] : p.sf, slope = c1_slope(p.act);
    const bf16x8 wa0 = c1_weight_frag(p, li, kg, 0), wa1 = c1_weight_frag(p, li, kg, 1);
    f32x2 eb[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.bias) { eb[0] = (f32x2){p.bias[4 * kg], p.bias[4 * kg + 1]}; eb[1] = (f32x2){p.bias[4 * kg + 2], p.bias[4 * kg + 3]}; }
    if (tid < 32) stat[tid] = 0.f;
    // B-fragment offsets of sub-tile 0 of this wave's plane: rows 2 kg and 2 kg + 1 of the (a, b) taps, then row 8
    const int r0 = 2 * kg, r1 = 2 * kg + 1;
    const int off0 = ((wave + r0 / 3) * HH + r0 % 3) * ROWB + li * 8, off1 = ((wave + r1 / 3) * HH + r1 % 3) * ROWB + li * 8;
    const int off2 = ((wave + 2) * HH + 2) * ROWB + li * 8;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int ntiles = tiles_w * tiles_h * tiles_d;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int cst = 8 * (kg >> 1), jodd = kg & 1;
    const size_t rowpitch = (size_t)p.W * 16;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, t2 = tile / tiles_w, th = t2 % tiles_h, td = t2 / tiles_h;
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
        __syncthreads();                                                   // the previous tile's fragment reads are done
        c1_stage_image<S>(p, img, n, od0, oh0, ow0, sc, sf, slope, tid);
        __syncthreads();
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x2 lo = *(const u32x2*)(img + off0 + j * ROWB), hi = *(const u32x2*)(img + off1 + j * ROWB), tl = *(const u32x2*)(img + off2 + j * ROWB);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, ((u32x4){lo[0], lo[1], hi[0], hi[1]}));
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, ((u32x4){tl[0], tl[1], tl[0], tl[1]}));      // (slots 36.. multiply zero weights)
            acc[j] = VG_MFMA16(wa0, b0, ((f32x4){0.f, 0.f, 0.f, 0.f}));
            acc[j] = VG_MFMA16(wa1, b1, acc[j]);
        }
        // epilogue: lane = voxel (od, oh0 + j, ow) x channels 4 kg .. + 3; pairs of sub-tiles exchange 16-lane rows -> 16-byte stores
        const int od = od0 + wave, ow = ow0 + li;
        const bool dw_ok = od < p.D && ow < p.W;
        T* const optr = (T*)p.out + ((((size_t)n * p.D + min(od, p.D - 1)) * p.H + oh0) * p.W + min(ow, p.W - 1)) * 16 + cst;
#pragma unroll
        for (int jp = 0; jp < 8; jp += 2) {
            bf16x4 pk[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = jp + e;
                const bool ok = dw_ok && oh0 + j < p.H;
                f32x2 v0 = {acc[j][0], acc[j][1]}, v1 = {acc[j][2], acc[j][3]};
                v0 += eb[0]; v1 += eb[1];
                pk[e] = (bf16x4){(short)f2bf(v0[0]), (short)f2bf(v0[1]), (short)f2bf(v1[0]), (short)f2bf(v1[1])};
                f32x2 q0 = {bf2f((bf16_t)pk[e][0]), bf2f((bf16_t)pk[e][1])}, q1 = {bf2f((bf16_t)pk[e][2]), bf2f((bf16_t)pk[e][3])};
                if (!ok) { q0 = (f32x2){0.f, 0.f}; q1 = q0; }
                s1[0] += q0[0]; s1[1] += q0[1]; s1[2] += q1[0]; s1[3] += q1[1];
                s2[0] += q0[0] * q0[0]; s2[1] += q0[1] * q0[1]; s2[2] += q1[0] * q1[0]; s2[3] += q1[1] * q1[1];
            }
            const u32x2 wa = __builtin_bit_cast(u32x2, pk[0]), wb = __builtin_bit_cast(u32x2, pk[1]);
            const u32x2 x0 = __builtin_amdgcn_permlane16_swap(wa[0], wb[0], false, false);
            const u32x2 x1 = __builtin_amdgcn_permlane16_swap(wa[1], wb[1], false, false);
            const u32x4 outv = {x0[0], x1[0], x0[1], x1[1]};
            const int j = jp + jodd;
            if (dw_ok && oh0 + j < p.H) *(u32x4*)(optr + j * rowpitch) = outv;
        }
    }
    if (!p.sums) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = s1[r], b = s2[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (li == 0) { atomicAdd(&stat[(4 * kg + r) * 2], a); atomicAdd(&stat[(4 * kg + r) * 2 + 1], b); }
    }
    __syncthreads();
    if (tid < 32) {
        const int stripe = blockIdx.x & (VG_STRIPES - 1);
        atomicAdd(&p.sums[((size_t)stripe * gridDim.y + n) * 32 + tid], stat[tid]);
    }
    if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.y, 16, gridDim.x * gridDim.y, (int*)img);
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256, 2) void c1k3m_wgrad_kernel(const C1K3 p) {
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) char img[IMGB + 16];          // + {1,1,1,1}, {0,0,0,0}: the bias-gradient row and the unused rows
    __shared__ __attribute__((aligned(16))) char dyt[DYB];
    __shared__ float red[3 * 4 * 64];                                     // [M tile][value][lane]: the four waves' accumulators summed
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    const float sc = p.scale ? p.scale[n] : p.sc, sf = p.scale ? p.shift[n] : p.sf, slope = c1_slope(p.act);
    if (tid == 0) {
        const unsigned one = f2bf(1.f);
        *(u32x4*)(img + IMGB) = (u32x4){one | (one << 16), one | (one << 16), 0u, 0u};
    }
    for (int i = tid; i < 3 * 4 * 64; i += 256) red[i] = 0.f;
    // transposed-read addresses (ds_read_b64_tr_b16: lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. 4 pp + 3 of its block):
    //   rows = voxels 8 g + q (+ 4 for the second read) of the K-step's 32 voxels -- image row y = 2 s + (g >> 1), column w = 8 (g & 1) + q (+ 4);
    //   A columns = the four slots of tap row 4 mt + pp;   B columns = channels 4 pp .. 4 pp + 3 of dY
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int wq = 8 * (g & 1) + q, yq = g >> 1;
    int aoff[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int r = 4 * mt + pp;
        aoff[mt] = r < 9 ? ((wave + r / 3) * HH + yq + r % 3) * ROWB + wq * 8 : (r == 9 ? IMGB : IMGB + 8);
    }
    const int amove[3] = {1, 1, pp == 0 ? 1 : 0};                          // the constant rows do not move with the voxel
    const int boff = ((wave * TH + yq) * TW + wq) * 32 + pp * 8;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int ntiles = tiles_w * tiles_h * tiles_d;
    f32x4 acc[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, t2 = tile / tiles_w, th = t2 % tiles_h, td = t2 / tiles_h;
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
        // dY of the tile: 32 rows of 512 bytes, 4 units of 16 bytes per thread (out-of-range voxels: zeros -- they then count for nothing)
        f32x4 dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 256 * i, row = u >> 5, c16 = u & 31;
            const int z = row >> 3, y = row & 7, w = c16 >> 1;
            const bool ok = od0 + z < p.D && oh0 + y < p.H && ow0 + w < p.W;
            const T* src = (const T*)p.dy + ((((size_t)n * p.D + min(od0 + z, p.D - 1)) * p.H + min(oh0 + y, p.H - 1)) * p.W + min(ow0 + w, p.W - 1)) * 16 + (c16 & 1) * 8;
            dv[i] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)src;
            if (!ok) dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                                   // the previous tile's reads are done
        c1_stage_image<S>(p, img, n, od0, oh0, ow0, sc, sf, slope, tid);
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(dyt + (tid + 256 * i) * 16) = dv[i];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                      // K-step: rows 2 s, 2 s + 1 of this wave's plane
            const s16x4 b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + boff + s * 2 * DYROW));
            const s16x4 b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + boff + s * 2 * DYROW + 4 * 32));
            const bf16x8 b = {b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const int o = aoff[mt] + amove[mt] * (s * 2 * ROWB);
                const s16x4 a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + o));
                const s16x4 a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + o + amove[mt] * 32));
                const bf16x8 a = {a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
                acc[mt] = VG_MFMA16(a, b, acc[mt]);
            }
        }
    }
    // lane (kg, li) of M tile mt holds dW[tap row 4 mt + kg][slot e = 0..3][channel li]: sum the four waves through LDS, then one atomic
    // per (workgroup, value): 27 x 16 weights + 16 bias sums (the row of ones: tap row 9, slot 0)
    const int li = lane & 15, kg = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&red[(mt * 4 + e) * 64 + lane], acc[mt][e]);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int r = 4 * mt + kg;
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (r < 9) atomicAdd(&p.dw[(size_t)(r * 3 + e) * 16 + li], red[(mt * 4 + e) * 64 + lane]);
            if (r == 9 && p.db) atomicAdd(&p.db[li], red[(mt * 4) * 64 + lane]);
        }
    }
}
}  // namespace

static bool c1k3m_ok(const C1K3& c) {
    if (!vg_tune("C1K3M", 1) || c.C != 16) return false;
    if (c.D < 2 || c.H < 2 || c.W < 2) return false;
    return true;
}
static dim3 c1k3m_grid(const C1K3& c, int N, int per_cu) {
    const int64_t tiles = (int64_t)((c.D + TD - 1) / TD) * ((c.H + TH - 1) / TH) * ((c.W + TW - 1) / TW);
    int64_t b = (int64_t)256 * per_cu / (N > 0 ? N : 1);
    if (b > tiles) b = tiles;
    if (b < 1) b = 1;
    return dim3((unsigned)b, (unsigned)N);
}

int c1k3m_fwd(const C1K3& c, int N, bool src_f32, hipStream_t s) {
    if (!c1k3m_ok(c)) return 1;
    if (vg_dry("c1k3m_fwd<%s>", src_f32 ? "f32" : "bf16")) return VG_OK;
    const dim3 grid = c1k3m_grid(c, N, vg_tune("C1K3M_FWD_WGS", 4));
    if (src_f32) hipLaunchKernelGGL((c1k3m_fwd_kernel<float>), grid, dim3(256), 0, s, c);
    else hipLaunchKernelGGL((c1k3m_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, c);
    return vg_check_launch();
}

int c1k3m_wgrad(const C1K3& c, int N, bool src_f32, hipStream_t s) {
    if (!c1k3m_ok(c) || !vg_tune("C1K3M_WGRAD", 1)) return 1;
    if (vg_dry("c1k3m_wgrad<%s>", src_f32 ? "f32" : "bf16")) return VG_OK;
    const dim3 grid = c1k3m_grid(c, N, vg_tune("C1K3M_WGRAD_WGS", 2));
    if (src_f32) hipLaunchKernelGGL((c1k3m_wgrad_kernel<float>), grid, dim3(256), 0, s, c);
    else hipLaunchKernelGGL((c1k3m_wgrad_kernel<bf16_t>), grid, dim3(256), 0, s, c);
    return vg_check_launch();
}
