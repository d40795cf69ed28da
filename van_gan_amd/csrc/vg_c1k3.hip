// vg_c1k3.hip -- the convolutions that read a SINGLE-channel volume, and their weight gradients, on the matrix pipe (gfx950):
//   the stem's 3x3x3 convolution 1 -> 16 (resunet_model.py:44-60) and the discriminator's first layer, 4x4x4 stride 2, 1 -> 64 over the
//   reflect-padded volume + Gaussian noise (discriminator.py:50-60).
// 16-bit storage builds only; the exact-parity mode stays on the VALU / generic kernels (vg_pointwise.hip, vg_conv.hip).
//
// Why: these layers move 4 + 32 (or 4 + 16) bytes per voxel next to a few hundred MACs.  As VALU work (c1k3_fwd_kernel: a thread owns
// 4 voxels x 8 channels, 864 FMAs + ~400 instructions of window loads and transforms per iteration) the stem ran at 120 us per 128^3
// volume where the 67 MB it writes take 15, its weight gradient at 153 us for two volumes (134 MB of dY: 27 us); on the generic MFMA
// kernels (W-packed: 16-channel chunks with 3 or 4 live channels) the discriminator's layer took 110 / 128 us.  All are chip-filling
// launches at the head / tail of every network application.
//
// How: the KS taps along W become 4 K-slots of a GEMM (KS = 3: the 4th multiplies a zero weight), the KS^2 (d, h) taps its rows:
//   k = 4 * (KS * a + b) + j,   a, b = tap offset along D, H,   j = 0..3 along W            -- 36 or 64 slots, two K-steps of 32.
// The LDS image of an output tile (16 x TH x 4) holds, per (source plane, source row, OUTPUT column w), ONE 8-byte entry
// {x[ST w + j - pad], j = 0..3} of transformed, rounded source values: slot group (a, b) of output voxel (z, y, w) is the entry at
// (ST z + a, ST y + b, w) -- an aligned ds_read_b64, no gather, 8 bytes of LDS per 4 K-slots.
//   forward   D[co][voxel] = W[co][k] * X[k][voxel]: two v_mfma_f32_16x16x32 per 16 voxels and 16 channels, the B fragment of a lane =
//             the entries of rows 8 s + 2 kg, + 1; epilogue as conv_thin's (bias, 16-byte stores after a row swap, InstanceNorm
//             statistics of the stored values, finalisation by the last workgroup).
//   gradient  dW[k][co] = sum over voxels X[k][voxel] * dY[voxel][co]: voxels are the K index, so both operands are needed
//             voxel-major per lane while the image (slot-major per voxel) and dY ([voxel][channels]) are stored the other way round:
//             ds_read_b64_tr_b16 hands a 16-lane group a 4 (voxels) x 16 (slots | channels) block transposed, and since every lane
//             supplies its own row address, the "row" of the A block is assembled on the fly from the entries of 4 different (a, b)
//             rows -- no im2col image.  M tiles of 4 tap rows; one extra row of ones whose product is the bias gradient.  A workgroup
//             keeps its accumulators across all its tiles and hands in one partial slab (summed by reduce_partials) or atomics.
#include "vg_c1k3.h"

namespace {
constexpr int TW = 16, TD = 4;
constexpr int ROWB = TW * 8 + 8;                       // bytes of one image row: 16 entries + 8 (bank spread of the staging writes)

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KS, int ST, int TH> struct Geo {
    static constexpr int IH = ST * TH + KS - ST, ID = ST * TD + KS - ST;        // source rows / planes under the tile
    static constexpr int NROWS = ID * IH, IMGB = NROWS * ROWB;
    static constexpr int NI = 3 * ST + 4;                                      // source values of one (row, quarter of the columns) item
    static constexpr int NRT = KS * KS;                                        // tap rows
};

__device__ __forceinline__ float c1_slope(int act) { return act == VG_ACT_RELU ? 0.f : (act == VG_ACT_LRELU ? VG_LRELU : 1.f); }

// the image of the tile with output origin (od0, oh0, ow0): item = (source row, quarter of the 16 columns); NI source values give 4 entries
template <typename S, int KS, int ST, int TH, bool NOISE>
__device__ __forceinline__ void c1_stage_image(const C1M& p, char* img, int n, int od0, int oh0, int ow0, float sc, float sf, float slope, int tid) {
    typedef Geo<KS, ST, TH> G;
    const bool refl = p.pad_mode == VG_PAD_REFLECT;
    for (int item = tid; item < G::NROWS * 4; item += 256) {
        const int row = item >> 2, q = item & 3;
        const int zz = row / G::IH, yy = row - zz * G::IH;
        const int pd = ST * od0 + p.td0 + zz, ph = ST * oh0 + p.th0 + yy, pw0 = ST * (ow0 + 4 * q) + p.tw0;
        bool okd, okh;
        const int rd = c1_resolve(pd, p.D, refl, okd), rh = c1_resolve(ph, p.H, refl, okh);
        const S* src = (const S*)p.x + (((int64_t)n * p.D + rd) * p.H + rh) * p.W;
        const bf16_t* nz = nullptr;
        if (NOISE) nz = (const bf16_t*)p.noise + (((int64_t)n * (p.D + 2) + min(max(pd + 1, 0), p.D + 1)) * (p.H + 2) + min(max(ph + 1, 0), p.H + 1)) * (p.W + 2);
        float v[G::NI], nv[G::NI]; bool ok[G::NI];
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            const int cw = c1_resolve(pw0 + i, p.W, refl, ok[i]);
            v[i] = ld_global(src + cw);
            nv[i] = NOISE ? ld_global(nz + min(max(pw0 + i + 1, 0), p.W + 1)) : 0.f;
        }
        unsigned short h[G::NI];
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            float y = v[i] * sc + sf;
            y = fmaxf(y, y * slope);
            y = (okd && okh && ok[i]) ? y : 0.f;
            h[i] = f2bf(y + nv[i]);
        }
        char* dst = img + row * ROWB + q * 32;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *(u32x2*)(dst + e * 8) = (u32x2){h[ST * e] | ((unsigned)h[ST * e + 1] << 16), h[ST * e + 2] | ((unsigned)h[ST * e + 3] << 16)};
    }
}

// the weights of output channel co, K-step s as the lane's 16x16x32 fragment: k = 32 s + 8 kg + i -> (tap row, slot)
template <int KS>
__device__ __forceinline__ bf16x8 c1_weight_frag(const C1M& p, int co, int kg, int s) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 32 * s + 8 * kg + i, t = k >> 2, j = k & 3;
        r[i] = (t < KS * KS && j < KS) ? (short)((const bf16_t*)p.w)[(size_t)co * p.Ktot + t * p.CK + j] : (short)0;
    }
    return r;
}

template <typename S, int KS, int ST, int MT, bool NOISE>
__global__ __launch_bounds__(256, 2) void c1m_fwd_kernel(const C1M p) {
    typedef bf16_t T;
    constexpr int TH = 8, C = 16 * MT;
    typedef Geo<KS, ST, TH> G;
    __shared__ __attribute__((aligned(16))) char img[G::IMGB];
    __shared__ float stat[32 * MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    const float sc = p.scale ? p.scale[n] : p.sc, sf = p.scale ? p.shift[n] : p.sf, slope = c1_slope(p.act);
    bf16x8 wa[MT][2];
    f32x2 eb[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        wa[mt][0] = c1_weight_frag<KS>(p, 16 * mt + li, kg, 0); wa[mt][1] = c1_weight_frag<KS>(p, 16 * mt + li, kg, 1);
        const int c = 16 * mt + 4 * kg;
        eb[mt][0] = (f32x2){0.f, 0.f}; eb[mt][1] = eb[mt][0];
        if (p.bias) { eb[mt][0] = (f32x2){p.bias[c], p.bias[c + 1]}; eb[mt][1] = (f32x2){p.bias[c + 2], p.bias[c + 3]}; }
    }
    if (tid < 32 * MT) stat[tid] = 0.f;
    // B-fragment offsets of sub-tile 0 of this wave's plane: tap rows 8 s + 2 kg and + 1 (rows past the last multiply zero weights:
    // they read the last row's entry, which is finite)
    int boff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = min(8 * s + 2 * kg + h, G::NRT - 1), a = t / KS, b = t - a * KS;
            boff[s][h] = ((ST * wave + a) * G::IH + b) * ROWB + li * 8;
        }
    const int tiles_w = (p.OW + TW - 1) / TW, tiles_h = (p.OH + TH - 1) / TH, tiles_d = (p.OD + TD - 1) / TD;
    const int ntiles = tiles_w * tiles_h * tiles_d;
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }
    const int cst = 8 * (kg >> 1), jodd = kg & 1;
    const size_t rowpitch = (size_t)p.OW * C;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, t2 = tile / tiles_w, th = t2 % tiles_h, td = t2 / tiles_h;
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
        __syncthreads();                                                   // the previous tile's fragment reads are done
        c1_stage_image<S, KS, ST, TH, NOISE>(p, img, n, od0, oh0, ow0, sc, sf, slope, tid);
        __syncthreads();
        // lane = voxel (od, oh0 + j, ow) x channels 16 mt + 4 kg .. + 3; pairs of sub-tiles exchange 16-lane rows -> 16-byte stores
        const int od = od0 + wave, ow = ow0 + li;
        const bool dw_ok = od < p.OD && ow < p.OW;
        T* const obase = (T*)p.out + ((((size_t)n * p.OD + min(od, p.OD - 1)) * p.OH + oh0) * p.OW + min(ow, p.OW - 1)) * C + cst;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jo = ST * j * ROWB;
                const u32x2 a0 = *(const u32x2*)(img + boff[0][0] + jo), a1 = *(const u32x2*)(img + boff[0][1] + jo);
                const u32x2 c0 = *(const u32x2*)(img + boff[1][0] + jo), c1 = *(const u32x2*)(img + boff[1][1] + jo);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, ((u32x4){a0[0], a0[1], a1[0], a1[1]}));
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, ((u32x4){c0[0], c0[1], c1[0], c1[1]}));
                acc[j] = VG_MFMA16(wa[mt][0], b0, ((f32x4){0.f, 0.f, 0.f, 0.f}));
                acc[j] = VG_MFMA16(wa[mt][1], b1, acc[j]);
            }
            T* const optr = obase + 16 * mt;
#pragma unroll
            for (int jp = 0; jp < 8; jp += 2) {
                bf16x4 pk[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = jp + e;
                    const bool ok = dw_ok && oh0 + j < p.OH;
                    f32x2 v0 = {acc[j][0], acc[j][1]}, v1 = {acc[j][2], acc[j][3]};
                    v0 += eb[mt][0]; v1 += eb[mt][1];
                    pk[e] = (bf16x4){(short)f2bf(v0[0]), (short)f2bf(v0[1]), (short)f2bf(v1[0]), (short)f2bf(v1[1])};
                    f32x2 q0 = {bf2f((bf16_t)pk[e][0]), bf2f((bf16_t)pk[e][1])}, q1 = {bf2f((bf16_t)pk[e][2]), bf2f((bf16_t)pk[e][3])};
                    if (!ok) { q0 = (f32x2){0.f, 0.f}; q1 = q0; }
                    s1[mt][0] += q0[0]; s1[mt][1] += q0[1]; s1[mt][2] += q1[0]; s1[mt][3] += q1[1];
                    s2[mt][0] += q0[0] * q0[0]; s2[mt][1] += q0[1] * q0[1]; s2[mt][2] += q1[0] * q1[0]; s2[mt][3] += q1[1] * q1[1];
                }
                const u32x2 wa_ = __builtin_bit_cast(u32x2, pk[0]), wb_ = __builtin_bit_cast(u32x2, pk[1]);
                const u32x2 x0 = __builtin_amdgcn_permlane16_swap(wa_[0], wb_[0], false, false);
                const u32x2 x1 = __builtin_amdgcn_permlane16_swap(wa_[1], wb_[1], false, false);
                const u32x4 outv = {x0[0], x1[0], x0[1], x1[1]};
                const int j = jp + jodd;
                if (dw_ok && oh0 + j < p.OH) *(u32x4*)(optr + j * rowpitch) = outv;
            }
        }
    }
    if (!p.sums) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = s1[mt][r], b = s2[mt][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (li == 0) { atomicAdd(&stat[(16 * mt + 4 * kg + r) * 2], a); atomicAdd(&stat[(16 * mt + 4 * kg + r) * 2 + 1], b); }
        }
    __syncthreads();
    if (tid < 32 * MT) {
        const int stripe = blockIdx.x & (VG_STRIPES - 1);
        atomicAdd(&p.sums[((size_t)stripe * gridDim.y + n) * (2 * C) + tid], stat[tid]);
    }
    if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.y, C, gridDim.x * gridDim.y, (int*)img);
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------
template <typename S, int KS, int ST, int TH, int NT, bool NOISE>
__global__ __launch_bounds__(256, 2) void c1m_wgrad_kernel(const C1M p) {
    typedef bf16_t T;
    typedef Geo<KS, ST, TH> G;
    constexpr int C = 16 * NT, VB = 2 * C;                                 // bytes of one dY voxel
    constexpr int DYROW = TW * VB, DYB = TD * TH * DYROW;
    constexpr int MTS = (G::NRT + 1 + 3) / 4;                              // M tiles: the tap rows + the row of ones, 4 rows per tile
    constexpr int UPT = DYB / 16 / 256;                                    // 16-byte units of dY per thread and tile
    static_assert(MTS * NT * 1024 <= DYB, "the accumulator exchange reuses the dY tile");
    __shared__ __attribute__((aligned(16))) char img[G::IMGB + 16];       // + {1,1,1,1}, {0,0,0,0}: the bias-gradient row and the unused rows
    __shared__ __attribute__((aligned(16))) char dyt[DYB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    const float sc = p.scale ? p.scale[n] : p.sc, sf = p.scale ? p.shift[n] : p.sf, slope = c1_slope(p.act);
    if (tid == 0) {
        const unsigned one = f2bf(1.f);
        *(u32x4*)(img + G::IMGB) = (u32x4){one | (one << 16), one | (one << 16), 0u, 0u};
    }
    // transposed-read addresses (ds_read_b64_tr_b16: lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. 4 pp + 3 of its block):
    //   rows = voxels 8 g + q (+ 4 for the second read) of the K-step's 32 voxels -- tile row y = 2 s + (g >> 1), column w = 8 (g & 1) + q (+ 4);
    //   A columns = the four slots of tap row 4 mt + pp;   B columns = channels 16 nt + 4 pp .. + 3 of dY
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int wq = 8 * (g & 1) + q, yq = g >> 1;
    int aoff[MTS], amove[MTS];
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt) {
        const int r = 4 * mt + pp, a = r / KS, b = r - a * KS;
        amove[mt] = r < G::NRT ? 1 : 0;                                    // the constant rows do not move with the voxel
        aoff[mt] = r < G::NRT ? ((ST * wave + a) * G::IH + ST * yq + b) * ROWB + wq * 8 : (r == G::NRT ? G::IMGB : G::IMGB + 8);
    }
    const int boff = ((wave * TH + yq) * TW + wq) * VB + pp * 8;
    const int tiles_w = (p.OW + TW - 1) / TW, tiles_h = (p.OH + TH - 1) / TH, tiles_d = (p.OD + TD - 1) / TD;
    const int ntiles = tiles_w * tiles_h * tiles_d;
    f32x4 acc[MTS][NT];
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, t2 = tile / tiles_w, th = t2 % tiles_h, td = t2 / tiles_h;
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
        // dY of the tile, 16-byte units (out-of-range voxels: zeros -- they then count for nothing)
        f32x4 dv[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + 256 * i, vox = u / (2 * NT), c16 = u - vox * (2 * NT);
            const int w = vox & 15, y = (vox >> 4) % TH, z = vox / (16 * TH);
            const bool ok = od0 + z < p.OD && oh0 + y < p.OH && ow0 + w < p.OW;
            const T* src = (const T*)p.dy + ((((size_t)n * p.OD + min(od0 + z, p.OD - 1)) * p.OH + min(oh0 + y, p.OH - 1)) * p.OW + min(ow0 + w, p.OW - 1)) * C + c16 * 8;
            dv[i] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)src;
            if (!ok) dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                                   // the previous tile's reads are done
        c1_stage_image<S, KS, ST, TH, NOISE>(p, img, n, od0, oh0, ow0, sc, sf, slope, tid);
#pragma unroll
        for (int i = 0; i < UPT; ++i) *(f32x4*)(dyt + (tid + 256 * i) * 16) = dv[i];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TH / 2; ++s) {                                 // K-step: rows 2 s, 2 s + 1 of this wave's plane
            bf16x8 a[MTS];
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
                const int o = aoff[mt] + amove[mt] * (s * 2 * ST * ROWB);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + o));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + o + amove[mt] * 32));
                a[mt] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + boff + s * 2 * DYROW + nt * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dyt + boff + s * 2 * DYROW + nt * 32 + 4 * VB));
                const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt) acc[mt][nt] = VG_MFMA16(a[mt], b, acc[mt][nt]);
            }
        }
    }
    // lane (kg, li) of tile (mt, nt) holds dW[tap row 4 mt + kg][slot e = 0..3][channel 16 nt + li]: the four waves' sums through LDS
    // (the dY tile's memory), then one value per (workgroup, element): the workgroup's slab, or an atomic
    float* red = (float*)dyt;
    __syncthreads();
    for (int i = tid; i < MTS * NT * 256; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&red[((mt * NT + nt) * 4 + e) * 64 + lane], acc[mt][nt][e]);
    __syncthreads();
    float* slab = p.part ? p.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (KS * KS * KS * C) : nullptr;
    for (int i = tid; i < MTS * NT * 256; i += 256) {
        const int l = i & 63, e = (i >> 6) & 3, t = i >> 8, nt = t % NT, mt = t / NT;
        const int r = 4 * mt + (l >> 4), co = 16 * nt + (l & 15);
        const float v = red[i];
        if (r < G::NRT && e < KS) {
            const int idx = (r * KS + e) * C + co;
            if (slab) slab[idx] = v; else atomicAdd(&p.dw[idx], v);
        } else if (r == G::NRT && e == 0 && p.db) atomicAdd(&p.db[co], v);
    }
}
}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------------
static bool c1m_fill(const vg_conv_desc* d, C1M& p) {
    if (!vg_tune("C1M", 1) || d->f32 || d->out_f32 || d->c_src0 != 1 || d->src1 || d->c_src1 || d->nclass > 1) return false;
    const int KS = d->wpack;
    if (KS != 3 && KS != 4) return false;
    if (d->istr != KS - 2 || d->ntaps != KS * KS || d->Cout != (KS == 3 ? 16 : 64)) return false;
    for (int t = 0; t < KS * KS; ++t)
        if (d->tap_d[t] != d->tap_d[0] + t / KS || d->tap_h[t] != d->tap_h[0] + t % KS || d->tap_w[t] != 0) return false;
    if (d->noise && !(d->noise_pad == 1 && d->pad_mode == VG_PAD_REFLECT)) return false;
    if (d->D < 2 || d->H < 2 || d->W < 2 || d->OD < 1 || d->OH < 1 || d->OW < 1) return false;
    p = C1M{};
    p.x = d->src0; p.x_f32 = d->src_f32; p.sc = 1.f; p.sf = 0.f; p.scale = d->in_scale; p.shift = d->in_shift; p.act = d->act;
    p.pad_mode = d->pad_mode; p.noise = d->noise;
    p.D = d->D; p.H = d->H; p.W = d->W; p.OD = d->OD; p.OH = d->OH; p.OW = d->OW; p.C = d->Cout;
    p.td0 = d->tap_d[0]; p.th0 = d->tap_h[0]; p.tw0 = d->wpack_wmin;
    p.w = d->wpacked; p.CK = d->CK; p.Ktot = ((KS * KS * d->CK + 31) / 32) * 32;
    return true;
}
static dim3 c1m_grid(const C1M& c, int N, int th, int per_cu) {
    const int64_t tiles = (int64_t)((c.OD + TD - 1) / TD) * ((c.OH + th - 1) / th) * ((c.OW + TW - 1) / TW);
    int64_t b = (int64_t)256 * per_cu / (N > 0 ? N : 1);
    if (b > tiles) b = tiles;
    if (b < 1) b = 1;
    return dim3((unsigned)b, (unsigned)N);
}

template <typename S, int KS, int MT>
static void c1m_launch_fwd(const C1M& c, dim3 grid, hipStream_t s) {
    if (c.noise) hipLaunchKernelGGL((c1m_fwd_kernel<S, KS, KS - 2, MT, true>), grid, dim3(256), 0, s, c);
    else hipLaunchKernelGGL((c1m_fwd_kernel<S, KS, KS - 2, MT, false>), grid, dim3(256), 0, s, c);
}
int c1m_fwd(const vg_conv_desc* d, hipStream_t s) {
    C1M c;
    if (!c1m_fill(d, c)) return 1;
    c.bias = d->bias; c.out = d->out; c.sums = d->out_sums; c.fin = vg_fin_of(d);
    const int KS = d->wpack;
    if (vg_dry("c1m_fwd<%s,%d,%d,%d,n%d>", d->src_f32 ? "f32" : "bf16", KS, KS - 2, c.C / 16, c.noise ? 1 : 0)) return VG_OK;
    const dim3 grid = c1m_grid(c, d->N, 8, vg_tune("C1M_FWD_WGS", KS == 3 ? 4 : 2));
    if (KS == 3) { if (d->src_f32) c1m_launch_fwd<float, 3, 1>(c, grid, s); else c1m_launch_fwd<bf16_t, 3, 1>(c, grid, s); }
    else { if (d->src_f32) c1m_launch_fwd<float, 4, 4>(c, grid, s); else c1m_launch_fwd<bf16_t, 4, 4>(c, grid, s); }
    if (c.sums && c.fin.ticket) vg_fin_done = true;
    return vg_check_launch();
}

template <typename S, int KS, int TH, int NT>
static void c1m_launch_wgrad(const C1M& c, dim3 grid, hipStream_t s) {
    if (c.noise) hipLaunchKernelGGL((c1m_wgrad_kernel<S, KS, KS - 2, TH, NT, true>), grid, dim3(256), 0, s, c);
    else hipLaunchKernelGGL((c1m_wgrad_kernel<S, KS, KS - 2, TH, NT, false>), grid, dim3(256), 0, s, c);
}
int c1m_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch, int64_t scratch_bytes, hipStream_t s) {
    C1M c;
    if (dy_f32 || !vg_tune("C1M_WGRAD", 1) || !c1m_fill(d, c)) return 1;
    const int KS = d->wpack;
    if (T_total != KS * KS) return 1;
    c.dy = dy; c.dw = dw; c.db = db;
    const int th = KS == 3 ? 8 : 4;
    // workgroups per CU: every workgroup hands in a slab of KS^3 * C sums, and the slab pass is a dependent chain over the slabs (measured at
    // 128^3: 3x3x3 1 -> 16: 47 / 41 / 48 / 52 us for 1 / 2 / 3 / 4 per CU; 4x4x4 1 -> 64, two volumes: 84 / 101 / 152 / 158)
    const dim3 grid = c1m_grid(c, d->N, th, vg_tune("C1M_WGRAD_WGS", KS == 3 ? 2 : 1));
    const int nslab = (int)(grid.x * grid.y), dw_elems = KS * KS * KS * c.C;
    c.part = (nslab > 4 && scratch && (int64_t)nslab * dw_elems * 4 <= scratch_bytes && vg_tune("C1M_WGRAD_PART", 1)) ? scratch : nullptr;
    if (vg_dry("c1m_wgrad<%s,%d,%d,%d,n%d>|part%d", d->src_f32 ? "f32" : "bf16", KS, KS - 2, c.C / 16, c.noise ? 1 : 0, c.part ? 1 : 0)) return VG_OK;
    if (KS == 3) { if (d->src_f32) c1m_launch_wgrad<float, 3, 8, 1>(c, grid, s); else c1m_launch_wgrad<bf16_t, 3, 8, 1>(c, grid, s); }
    else { if (d->src_f32) c1m_launch_wgrad<float, 4, 4, 4>(c, grid, s); else c1m_launch_wgrad<bf16_t, 4, 4, 4>(c, grid, s); }
    if (c.part) vg_launch_reduce_partials(c.part, nslab, dw_elems, dw, s);
    return vg_check_launch();
}

// ---- data gradient of the 4x4x4 stride-2 layer w.r.t. its single-channel input ------------------------------------------------------
// Position pp of the reflect-padded grid receives dY[q] * W[t] for 2 q + t = pp: two taps per axis.  Cell c = the 2 x 2 x 2 block of
// padded positions 2 c + r: all eight draw from the same dY voxels q = c - 1 + n, n in {0, 1} per axis, with tap t = r + 2 - 2 n.  So the
// data gradient is a stride-1 convolution of dY (C channels) to 8 "channels" (the block's positions) over the grid of cells -- a shape
// the thin-channel specialist serves (3x3x3 stencil with the n = 2 taps zero, 16 output channels of which 8 are used) -- followed by a
// fold of the cells into the volume (depth-to-space + the transpose of ReflectionPadding3D(1)).  The generic path multiplied, per
// output-parity class, 8 taps x C channels into a 16-column MFMA tile with ONE live column: 153 us per 128^3 volume; this one ~30.
namespace {
__global__ __launch_bounds__(256) void cell_pack_kernel(const float* __restrict__ w, int C, int Ktot, bf16_t* __restrict__ out) {
    // packed [64 rows][Ktot], k = chunk * 448 + tap * 16 + ch % 16 (vg_pack_weights with 27 taps, CK = 16); row = 4 rd + 2 rh + rw
    const size_t total = (size_t)64 * Ktot;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / Ktot), k = (int)(i % Ktot);
        const int chunk = k / 448, kl = k % 448, tap = kl / 16, ch = chunk * 16 + (kl & 15);
        float v = 0.f;
        if (row < 8 && tap < 27 && ch < C) {
            const int nd = tap / 9, nh = (tap / 3) % 3, nw = tap % 3;
            const int td = ((row >> 2) & 1) + 2 - 2 * nd, th = ((row >> 1) & 1) + 2 - 2 * nh, tw = (row & 1) + 2 - 2 * nw;
            if (td >= 0 && th >= 0 && tw >= 0) v = w[(size_t)((td * 4 + th) * 4 + tw) * C + ch];          // (t <= 3 always)
        }
        out[i] = f2bf(v);
    }
}
__global__ __launch_bounds__(256) void cells_fold_kernel(const bf16_t* __restrict__ cells, int D, int H, int W, float* __restrict__ dx) {
    const int n = blockIdx.y;
    const int CD = D / 2 + 1, CH = H / 2 + 1, CW = W / 2 + 1;
    const bf16_t* cb = cells + (size_t)n * CD * CH * CW * 16;
    const int64_t total = (int64_t)D * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int iw = (int)(i % W); const int64_t r = i / W; const int ih = (int)(r % H), id = (int)(r / H);
        // padded positions that reflect onto (id, ih, iw): i + 1, and 0 for i == 1, and n + 1 for i == n - 2
        int pd[2], ph[2], pw[2], nd = 1, nh = 1, nw = 1;
        pd[0] = id + 1; ph[0] = ih + 1; pw[0] = iw + 1;
        if (id == 1) pd[nd++] = 0; else if (id == D - 2) pd[nd++] = D + 1;          // (n >= 4: the two never share an index)
        if (ih == 1) ph[nh++] = 0; else if (ih == H - 2) ph[nh++] = H + 1;
        if (iw == 1) pw[nw++] = 0; else if (iw == W - 2) pw[nw++] = W + 1;
        float s = 0.f;
        for (int a = 0; a < nd; ++a)
            for (int b = 0; b < nh; ++b)
                for (int c = 0; c < nw; ++c) {
                    const int qd = pd[a], qh = ph[b], qw = pw[c];
                    s += bf2f(cb[(((size_t)(qd >> 1) * CH + (qh >> 1)) * CW + (qw >> 1)) * 16 + ((qd & 1) * 4 + (qh & 1) * 2 + (qw & 1))]);
                }
        dx[(size_t)n * total + i] = s;
    }
}
}  // namespace

extern "C" int vg_pack_cell_weights(const float* w, int C, void* out, vg_stream_t stream) {
    vg_begin();
    if (!w || !out || C < 16 || (C % 16)) return VG_EINVAL;
    const int Ktot = (C / 16) * 448;
    hipLaunchKernelGGL(cell_pack_kernel, dim3((64 * Ktot + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, C, Ktot, (bf16_t*)out);
    return vg_check_launch();
}
extern "C" int vg_cells_fold(const void* cells, int N, int D, int H, int W, float* dx, vg_stream_t stream) {
    vg_begin();
    // (an axis of length 3 folds both border positions, 0 and n + 1, onto index 1 == n - 2: the kernel keeps one extra slot per axis, so n >= 4)
    if (!cells || !dx || N < 1 || D < 4 || H < 4 || W < 4 || ((D | H | W) & 1)) return VG_EINVAL;
    int64_t b = ((int64_t)D * H * W + 255) / 256;
    if (b > 2047 / N) b = 2047 / N > 0 ? 2047 / N : 1;
    hipLaunchKernelGGL(cells_fold_kernel, dim3((unsigned)b, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cells, D, H, W, dx);
    return vg_check_launch();
}
