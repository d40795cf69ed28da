// vg_conv.hip -- gather-convolution as an implicit GEMM on bf16 MFMA (gfx950).
//
// Replaces, for the VAN-GAN hot path: ReflectionPadding3D (building_blocks.py:30-39) + InstanceNorm
// apply + ReLU/LeakyReLU (resunet_model.py:23-39, building_blocks.py:190-195) + UpSampling3D/concatenate
// (resunet_model.py:175-181) + GaussianNoise/SpatialDropout3D + Conv3D (resunet_model.py:42-143,
// discriminator.py:50-117) + Add/tanh, and the data gradient of those convolutions.
//
// Structure per workgroup (256 threads = 4 waves, one output tile of TDxTHxTW voxels x BN channels):
//   for each chunk of CK contraction channels:
//     stage the input HALO tile once into LDS, already normalised/activated/noised and rounded
//     to bf16 (every input element is transformed once, not once per tap);
//     for each K-step of 32 (tap, channel-group) pairs: MFMA 16x16x32 bf16 with
//       A = packed weights (rows = output channels, straight from L2) and
//       B = halo rows (cols = voxels, ds_read_b128 of 8 channels of one voxel),
//     so the accumulator holds 4 consecutive output channels of one voxel per lane and the
//     epilogue stores 8 B per lane, 512 B contiguous per 16-voxel subtile.
//   epilogue: bias, residual*scale+shift, tanh, bf16 round, per-(n,c) sum / sum-of-squares of
//   the stored values (for the next InstanceNorm), store or accumulate.
#include "vg_conv_common.h"
#include "vg_dma_common.h"

unsigned long long* g_vg_stamps = nullptr;
extern "C" int vg_set_stamp_buffer(void* p) { g_vg_stamps = (unsigned long long*)p; return VG_OK; }
#define VG_STAMP(slot) do { if (g.stamps && tid == 0 && it < 8) g.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + it * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA staging (planar bf16 images, multi-channel sources without noise).  The halo tile of the NEXT stage is copied
// global -> LDS by global_load_lds_dwordx4 while the MFMAs and the epilogue of the current stage run: no VGPR, no wait
// until the next stage starts.  One wave-instruction fills 64 consecutive 16-byte units (one piece of a D-slice of one
// channel-group plane; slices are padded to whole pieces), each lane supplying its own source address: reflection, zero
// padding (-> a zero page), the virtual concat and the nearest upsample are all in that address.  The on-read transform
// act(x*scale+shift) then runs LDS -> LDS when the stage is consumed.
//   cq  : LDS ints [SU]      static: hh | hw << 10 of image column q, or -1 for pitch/piece padding
//   tcol: LDS ints [2][SU]   per tile: byte offset of column q inside one D-plane of src0 / src1, or -1
// ------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned int vg_zero16[4];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void dma_build_cq(const GatherIn& g, int* cq, int tid) {
    const int SU = g.DS >> 4;
    for (int q = tid; q < SU; q += 256) {
        const int hh = q / g.HWp, hw = q - hh * g.HWp;
        cq[q] = (hh < g.HH && hw < g.HW) ? (hh | (hw << 10)) : -1;
    }
}
__device__ __forceinline__ void dma_build_tcol(const GatherIn& g, const int* cq, int* tcol, int oh0, int ow0, int tid) {
    const int SU = g.DS >> 4;
    const int ph0 = oh0 * g.istr + g.tmin_h, pw0 = ow0 * g.istr + g.tmin_w;
    const int sh = g.shift0, Ws0 = g.W >> sh;
    for (int q = tid; q < SU; q += 256) {
        const int e = cq[q];
        int o0 = -1, o1 = -1;
        if (e >= 0) {
            int ph = ph0 + (e & 1023), pw = pw0 + (e >> 10);
            const bool ok = resolve_pos(ph, g.H, g.pad_mode) & resolve_pos(pw, g.W, g.pad_mode);
            if (ok) { o0 = (((ph >> sh) * Ws0 + (pw >> sh)) * g.c0) * 2; o1 = ((ph * g.W + pw) * g.c1) * 2; }
        }
        tcol[q] = o0; tcol[SU + q] = o1;
    }
}
// issue the copies of one stage (tile origin od0, channel chunk `chunk`) into the image at hb
__device__ __forceinline__ void dma_issue(const GatherIn& g, char* hb, const int* tcol, int n, int od0, int chunk, int wave, int lane) {
    const int SU = g.DS >> 4, J = SU >> 6, gpc = g.CK >> 3;
    const int nitems = gpc * g.HD * J;
    const int sh = g.shift0;
    const size_t pl0 = (size_t)(g.H >> sh) * (g.W >> sh) * g.c0 * 2, pl1 = (size_t)g.H * g.W * g.c1 * 2;     // bytes per source D-plane
    const char* zp = (const char*)vg_zero16;
    // items (cg, hd, j) in j-fastest order, wave w takes items w, w+4, ...: decoded incrementally (no divisions)
    int j = wave % J, t2 = wave / J; int hd = t2 % g.HD, cg = t2 / g.HD;
    const int dj = 4 % J, dhd = (4 / J) % g.HD, dcg = (4 / J) / g.HD;
    for (int it = wave; it < nitems; it += 4) {
        const int c = chunk * g.CK + cg * 8;
        int rd = od0 * g.istr + g.tmin_d + hd;
        const bool dvalid = resolve_pos(rd, g.D, g.pad_mode) && c < g.Cin;
        const bool from0 = c < g.c0;
        const char* base = from0 ? (const char*)g.src0 + ((size_t)n * (g.D >> sh) + (rd >> sh)) * pl0 + (size_t)c * 2
                                 : (const char*)g.src1 + ((size_t)n * g.D + rd) * pl1 + (size_t)(c - g.c0) * 2;
        const int toff = tcol[(from0 ? 0 : SU) + j * 64 + lane];
        const char* src = (dvalid && toff >= 0) ? base + toff : zp;
        char* dst = hb + (size_t)cg * g.PSB + (size_t)hd * g.DS + j * 1024;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(uintptr_t)src, (lds_void_t*)dst, 16, 0, 0);
        j += dj; if (j >= J) { j -= J; ++hd; }
        hd += dhd; if (hd >= g.HD) { hd -= g.HD; ++cg; }
        cg += dcg;
    }
}
// in-place on-read transform of a landed stage: y = act(x*scale+shift), 0 where the source position is padding
__device__ __forceinline__ void dma_transform(const GatherIn& g, char* hb, const int* tcol, const float* scs, int od0, int chunk,
                                              int wave, int lane) {
    if (!g.in_scale && g.act == VG_ACT_NONE) return;                // data-gradient operand: pure copy, zero page did the padding
    const int SU = g.DS >> 4, J = SU >> 6, gpc = g.CK >> 3;
    const int nitems = gpc * g.HD * J;
    const float slope = g.act == VG_ACT_RELU ? 0.f : (g.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    const bool zero_mode = g.pad_mode != VG_PAD_REFLECT;
    int j = wave % J, t2 = wave / J; int hd = t2 % g.HD, cg = t2 / g.HD;
    const int dj = 4 % J, dhd = (4 / J) % g.HD, dcg = (4 / J) / g.HD;
    for (int it = wave; it < nitems; it += 4, j += dj, hd += (j >= J ? 1 : 0) + dhd, j -= (j >= J ? J : 0), cg += (hd >= g.HD ? 1 : 0) + dcg, hd -= (hd >= g.HD ? g.HD : 0)) {
        const int c = chunk * g.CK + cg * 8;
        if (c >= g.Cin) continue;                                   // channel padding stays zero
        int rd = od0 * g.istr + g.tmin_d + hd;
        const bool dvalid = resolve_pos(rd, g.D, g.pad_mode);
        bf16x8* u = (bf16x8*)(hb + (size_t)cg * g.PSB + (size_t)hd * g.DS + j * 1024) + lane;
        if (zero_mode && !dvalid) continue;                         // whole slice came from the zero page
        f32x2 sc[4], sf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = (f32x2){scs[cg * 8 + 2 * e], scs[cg * 8 + 2 * e + 1]};
            sf[e] = (f32x2){scs[g.CK + cg * 8 + 2 * e], scs[g.CK + cg * 8 + 2 * e + 1]};
        }
        const bf16x8 raw = *u;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = bf2f((bf16_t)raw[e]);
        stage_affine_act(x, sc, sf, slope);
        if (zero_mode) {
            const bool ok = tcol[(c < g.c0 ? 0 : SU) + j * 64 + lane] >= 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(x[e]);
        *u = o;
    }
}

// Persistent workgroup: blockIdx.z = sample, blockIdx.y = BN-channel panel, blockIdx.x walks the output tiles of the
// sample.  Per workgroup ONCE: tap offsets, the (halo voxel, channel group) unit table, the weight panel (if it fits in
// LDS); per tile: stage halo (batched global loads), MFMA loop, epilogue; InstanceNorm statistics are carried in
// registers across tiles and flushed once.
// MC: 0 = one class (plain launch); 1 = all output-parity classes looped over one staged halo tile (single channel chunk);
// 2 = class-parallel: the workgroup serves the ONE class blockIdx.x % ncls (own taps / weights / output sub-lattice) and
// walks the tiles with the remaining part of blockIdx.x -- the launch carries ncls times the workgroups of a per-class launch
template <typename T, int BN, int MSUB, bool NOISE, bool WL, bool DMA, int MC, bool C1>
__global__ __launch_bounds__(256, ((BN / 16) * MSUB >= VG_CONV_MW2 ? 2 : VG_CONV_WAVES)) void conv_kernel(const GatherIn g, const ConvOut p, const ConvCls q) {
    static_assert(!(MC && DMA), "fused classes use the synchronous staging path");
    constexpr bool F32 = sizeof(T) == 4;
    // wave decomposition: WN waves along the channel panel (one 16-channel sub-tile each, so a weight fragment is
    // fetched by exactly one wave: L1 delivers 64 B/clk, LDS 256 B/clk), WM waves along the voxels
    constexpr int WN = BN / 16, WM = 4 / WN, MW = MSUB * WN;     // MW voxel sub-tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave % WN, wave_m = wave / WN;
    const int n = blockIdx.z, ntile = blockIdx.y;
    const int TWm = (1 << g.twl) - 1, THm = (1 << g.thl) - 1;
    int bx = blockIdx.x, gx = gridDim.x, t0 = 0, nt = g.ntaps;
    int Ktot = p.Ktot, kc_pad = p.kc_pad, WRS = p.WRS;
    const void* wsrc = p.wp;
    int c_od = p.ood, c_oh = p.ooh, c_ow = p.oow, c_OD = p.OD, c_OH = p.OH, c_OW = p.OW;
    // K split over workgroups (small deep-level grids: 8^3 / 16^3 voxels, hundreds of channels -- a tile's dependent chain of
    // hundreds of K-steps on a handful of workgroups is all latency): blockIdx.x = walker * ks + slice, slice s multiplies the
    // channel chunks [s * nchunks / ks, (s + 1) * nchunks / ks) of every tile and leaves its fp32 partial tile in p.ks_part; the
    // slice that arrives last (p.ks_cnt) adds the partials in slice order and runs the epilogue.
    int ks_slice = 0, cps = p.nchunks;
    if constexpr (MC == 0 && !DMA) {
        if (p.ks > 1) { ks_slice = __builtin_amdgcn_readfirstlane(bx % p.ks); bx /= p.ks; gx /= p.ks; cps = p.nchunks / p.ks; }
    }
    const int c_lo = ks_slice * cps;
    if constexpr (MC == 2) {
        const int cls = __builtin_amdgcn_readfirstlane(bx % q.ncls);
        bx /= q.ncls; gx /= q.ncls;
        t0 = q.tap0[cls]; nt = q.tap0[cls + 1] - t0;
        Ktot = q.ktot[cls]; kc_pad = Ktot / p.nchunks; WRS = Ktot * (int)sizeof(T) + 16;
        wsrc = q.wp[cls];
        c_od = q.off[cls][0]; c_oh = q.off[cls][1]; c_ow = q.off[cls][2];
        c_OD = q.it[cls][0]; c_OH = q.it[cls][1]; c_OW = q.it[cls][2];
    }

    char* halo = smem;
    const int hbytes = g.planar ? (g.CK >> 3) * g.PSB : g.HD * g.DS;
    const int nhalo = DMA ? 2 : 1;                   // LDS-DMA staging double-buffers the image
    int* tapoff = (int*)(smem + nhalo * hbytes);
    float* scs = (float*)(smem + nhalo * hbytes + 256);
    float* stat = scs + (DMA ? 4 : 2) * g.CK;        // DMA: two scale/shift buffers (next chunk is prepared ahead)
    int* utab = (int*)(stat + BN * 2);
    const int ncols = stage_ncols(g);
    const int SU = g.DS >> 4;                        // DMA: units per D-slice (multiple of 64)
    const int RTN = 3 * stage_axis_len<(C1 ? 1 : 3)>(g);               // one buffer of per-tile axis tables; two buffers: the tables of tile t+1
                                                     // are resolved while tile t is staged (no barrier, off the critical path)
    const int nunits = DMA ? 5 * SU : 2 * ncols + 2 * RTN;               // staging tables (see conv_lds_bytes)
    int* rtab = utab + 2 * ncols;
    int* cq = utab; int* tcolb = utab + SU;          // DMA: static column table, tcol[2 tiles][2 sources][SU]
    const int gpc = g.CK >> 3;
    const int ngroups = nt * gpc;
    const int ksteps = MC == 1 ? q.ks0[q.ncls] : (ngroups + 3) >> 2;      // MC: K-steps of all classes, each padded to whole steps
    int* koff = utab + nunits;                       // byte offset of the B fragment of (K-step, lane>>4) inside the halo tile
    char* wlds = (char*)(koff + ksteps * 4);
    wlds = (char*)(((size_t)wlds + 15) & ~(size_t)15);

    if (tid < (MC == 1 ? g.ntaps : nt))
        tapoff[tid] = (g.td[t0 + tid] - g.tmin_d) * g.DS + ((g.th[t0 + tid] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[t0 + tid] - g.tmin_w)) * g.VS;
    if (tid < BN * 2) stat[tid] = 0.f;
    if constexpr (DMA) dma_build_cq(g, cq, tid); else build_column_table(g, utab, tid);
    for (int i = tid; i < ksteps * 4; i += 256) {
        int tp, cgq;
        if constexpr (MC == 1) {
            int cls = 0; while (cls + 1 < q.ncls && (i >> 2) >= q.ks0[cls + 1]) ++cls;
            const int ng = (q.tap0[cls + 1] - q.tap0[cls]) * gpc;
            int G = i - q.ks0[cls] * 4; if (G >= ng) G = ng - 1;                // padded K: weights are zero there
            tp = q.tap0[cls] + G / gpc; cgq = G % gpc;
        } else {
            int G = i; if (G >= ngroups) G = ngroups - 1;                       // padded K: weights are zero there
            tp = G / gpc; cgq = G - tp * gpc; tp += t0;
        }
        koff[i] = (g.td[tp] - g.tmin_d) * g.DS + ((g.th[tp] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[tp] - g.tmin_w)) * g.VS + cgq * g.CS;
    }
    if (WL) {               // weight panel(s) -> LDS, 16 B per thread per step
        const int npan = MC == 1 ? q.ncls : 1;
        for (int cls = 0; cls < npan; ++cls) {
            const int ktot = MC == 1 ? q.ktot[cls] : Ktot;
            const char* src = (const char*)(MC == 1 ? q.wp[cls] : wsrc);
            char* dstp = wlds + (MC == 1 ? q.woff[cls] : 0);
            const int per_row = (ktot * (int)sizeof(T)) >> 4, wrs = ktot * (int)sizeof(T) + 16;
            // four loads in flight per thread: a load -> store loop with an unknown trip count is not pipelined by the
            // compiler and cost one L2 round trip per 16 bytes (several microseconds of every launch's prologue)
            for (int u0 = tid; u0 < BN * per_row; u0 += 1024) {
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = min(u0 + k * 256, BN * per_row - 1);
                    const int r = u / per_row, c = u - r * per_row;
                    v[k] = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(src + ((size_t)(ntile * BN + r) * ktot) * sizeof(T) + c * 16);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = u0 + k * 256;
                    if (u < BN * per_row) { const int r = u / per_row, c = u - r * per_row; *(f32x4*)(dstp + (size_t)r * wrs + c * 16) = v[k]; }
                }
            }
        }
    }

    int rowbase[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = (wave_m * MW + i) * 16 + (lane & 15);
        const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
        // W step is one unit for either stride: stride-2 gathers read the de-interleaved image (halo_pos_w)
        rowbase[i] = d * g.istr * g.DS + (h * g.istr * g.HWp + w) * g.VS;
    }
    // weight fragment source: LDS panel or global (L2) rows -- kept as two address-space-typed pointers (a pointer
    // selected between the two becomes generic and every fragment fetch a flat_load)
    const lds_ptr<T> wrow_l = (lds_ptr<T>)(wlds + (size_t)(wave_n * 16 + (lane & 15)) * WRS) + (F32 ? 1 : 8) * (lane >> 4);
    const glb_ptr<T> wrow_g = (glb_ptr<T>)wsrc + (size_t)(ntile * BN + wave_n * 16 + (lane & 15)) * Ktot + (F32 ? 1 : 8) * (lane >> 4);
    float s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    // per-lane epilogue constants: this lane always produces channels co0..co0+3
    const int co0 = ntile * BN + wave_n * 16 + 4 * (lane >> 4);
    float e_bias[4], e_rs[4], e_rb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + r;
        e_bias[r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        e_rs[r] = (p.res && co < p.Cout) ? p.rs[n * p.Cout + co] : 0.f;
        e_rb[r] = (p.res && co < p.Cout) ? p.rb[n * p.Cout + co] : 0.f;
    }
    // output position of (sub-tile i, this lane) relative to the tile origin: element offset and packed (d, h, w)
    int ooff[MW], dhw[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = (wave_m * MW + i) * 16 + (lane & 15);
        const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
        ooff[i] = ((d * p.ostr * p.BH + h * p.ostr) * p.BW + w * p.ostr) * p.Cout + co0;
        dhw[i] = d | (h << 10) | (w << 20);
    }
    const bool vec_epi = (p.Cout & 3) == 0 && !p.tanh_out;        // every lane owns 4 whole channels: vector loads/stores
    if (p.nchunks == 1) stage_scale_shift(g, scs, n, 0, tid);
    if constexpr (!DMA) {
        if (bx < g.tiles_d * g.tiles_h * g.tiles_w) {
            int t = bx;
            const int tw_i = t % g.tiles_w; t /= g.tiles_w;
            const int th_i = t % g.tiles_h;
            stage_resolve_axes<(C1 ? 1 : 3)>(g, rtab, th_i << g.thl, tw_i << g.twl, tid);
        }
    }
    __syncthreads();

    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    int it = -1;
    int sidx = 0;                                    // DMA: running stage index (tile visit x channel chunk)
    auto tile_origin = [&](int tile, int& od0, int& oh0, int& ow0) {
        int t = tile;
        const int tw_i = t % g.tiles_w; t /= g.tiles_w;
        const int th_i = t % g.tiles_h; const int td_i = t / g.tiles_h;
        od0 = td_i << g.tdl; oh0 = th_i << g.thl; ow0 = tw_i << g.twl;
    };
    if constexpr (DMA) {
        // prologue: tables and copies of the first stage
        if (bx < tiles_per_n) {
            int od0, oh0, ow0; tile_origin(bx, od0, oh0, ow0);
            dma_build_tcol(g, cq, tcolb, oh0, ow0, tid);
            if (p.nchunks > 1) stage_scale_shift(g, scs, n, 0, tid);
            __syncthreads();
            dma_issue(g, halo, tcolb, n, od0, 0, wave, lane);
        }
    }
    // tile coordinates advance incrementally by the grid stride (decomposed once): the per-tile divisions by run-time tile
    // counts were ~300 scalar instructions of the ~500-instruction empty tile iteration
    int gs_w, gs_h, gs_d;
    { int t = gx; gs_w = t % g.tiles_w; t /= g.tiles_w; gs_h = t % g.tiles_h; gs_d = t / g.tiles_h; }
    int ti_w, ti_h, ti_d;
    { int t = bx; ti_w = t % g.tiles_w; t /= g.tiles_w; ti_h = t % g.tiles_h; ti_d = t / g.tiles_h; }
    for (int tile = bx; tile < tiles_per_n; tile += gx) {
        ++it;
        VG_STAMP(0);
        const int od0 = ti_d << g.tdl, oh0 = ti_h << g.thl, ow0 = ti_w << g.twl;
        // coordinates of the next tile of this workgroup (valid if tile + gridDim.x < tiles_per_n)
        ti_w += gs_w; if (ti_w >= g.tiles_w) { ti_w -= g.tiles_w; ++ti_h; }
        ti_h += gs_h; if (ti_h >= g.tiles_h) { ti_h -= g.tiles_h; ++ti_d; }
        ti_d += gs_d;
        const int nx_d0 = ti_d << g.tdl, nx_h0 = ti_h << g.thl, nx_w0 = ti_w << g.twl;
        f32x4 acc[MW];
        if constexpr (MC == 1) {
            // fused output-parity classes: the dY halo tile is staged once; every class runs its own taps / weights /
            // accumulators over it and writes its own output sub-lattice
            __syncthreads();                       // previous readers of the halo tile are done; this tile's axis tables visible
            if (!(g.dbg & 1)) stage_halo_tile<T, NOISE, (NOISE ? 4 : 6), (C1 ? 1 : 3)>(g, halo, scs, utab, rtab + (it & 1) * RTN, n, od0, 0, tid);
            if (tile + gx < tiles_per_n) {
                stage_resolve_axes<(C1 ? 1 : 3)>(g, rtab + ((it + 1) & 1) * RTN, nx_h0, nx_w0, tid);
            }
            VG_STAMP(1);
            __syncthreads();
            VG_STAMP(2);
            for (int cls = 0; cls < q.ncls; ++cls) {
#pragma unroll
                for (int i = 0; i < MW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(g.dbg & 4)) {
                    const int ktot = q.ktot[cls];
                    const int ks = q.ks0[cls + 1] - q.ks0[cls], nt = q.tap0[cls + 1] - q.tap0[cls];
                    if constexpr (WL) {
                        const lds_ptr<T> w = (lds_ptr<T>)(wlds + q.woff[cls] + (size_t)(wave_n * 16 + (lane & 15)) * (ktot * (int)sizeof(T) + 16)) + (F32 ? 1 : 8) * (lane >> 4);
                        conv_mfma_chunk<T, MW>(acc, w, halo, rowbase, tapoff + q.tap0[cls], koff + q.ks0[cls] * 4, ks, nt, g.CK, g.CS, lane);
                    } else {
                        const glb_ptr<T> w = (glb_ptr<T>)q.wp[cls] + (size_t)(ntile * BN + wave_n * 16 + (lane & 15)) * ktot + (F32 ? 1 : 8) * (lane >> 4);
                        conv_mfma_chunk<T, MW>(acc, w, halo, rowbase, tapoff + q.tap0[cls], koff + q.ks0[cls] * 4, ks, nt, g.CK, g.CS, lane);
                    }
                }
                if (!(g.dbg & 8)) {
                    c_od = q.off[cls][0]; c_oh = q.off[cls][1]; c_ow = q.off[cls][2];
                    c_OD = q.it[cls][0]; c_OH = q.it[cls][1]; c_OW = q.it[cls][2];
#include "vg_conv_epilogue.inc"
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < MW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int chunk = c_lo; chunk < c_lo + cps; ++chunk) {
            char* hb = halo;
            if constexpr (DMA) {
                hb = halo + (sidx & 1) * hbytes;
                int* tc = tcolb + (it & 1) * 2 * SU;
                const float* sc_cur = scs + (p.nchunks > 1 ? (sidx & 1) * 2 * g.CK : 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this stage's copies have landed (this wave's share)
                __syncthreads();                                         // ... everyone's; previous stage fully consumed
                // identity of the next stage; its tables / scale-shift go to the other buffers while this stage is transformed
                const bool next_same = chunk + 1 < p.nchunks;
                const int ntile2 = next_same ? tile : tile + gx;
                const bool has_next = ntile2 < tiles_per_n;
                int nd0 = od0, nh0 = oh0, nw0 = ow0;
                if (has_next && !next_same) { nd0 = nx_d0; nh0 = nx_h0; nw0 = nx_w0; dma_build_tcol(g, cq, tcolb + ((it + 1) & 1) * 2 * SU, nh0, nw0, tid); }
                if (has_next && p.nchunks > 1) stage_scale_shift(g, scs + ((sidx + 1) & 1) * 2 * g.CK, n, next_same ? chunk + 1 : 0, tid);
                if (!(g.dbg & 1)) dma_transform(g, hb, tc, sc_cur, od0, chunk, wave, lane);
                if (chunk == 0) VG_STAMP(1);
                __syncthreads();
                if (chunk == 0) VG_STAMP(2);
                if (has_next)
                    dma_issue(g, halo + ((sidx + 1) & 1) * hbytes, tcolb + ((next_same ? it : it + 1) & 1) * 2 * SU, n, nd0, next_same ? chunk + 1 : 0, wave, lane);
                ++sidx;
            } else {
                __syncthreads();                       // previous readers of the halo tile are done; this tile's axis tables visible
                if (p.nchunks > 1) { stage_scale_shift(g, scs, n, chunk, tid); __syncthreads(); }
                if (!(g.dbg & 1)) stage_halo_tile<T, NOISE, (NOISE ? 4 : 6), (C1 ? 1 : 3)>(g, halo, scs, utab, rtab + (it & 1) * RTN, n, od0, chunk, tid);
                if (chunk == c_lo && tile + gx < tiles_per_n) {       // axis tables of the next tile, other buffer
                    stage_resolve_axes<(C1 ? 1 : 3)>(g, rtab + ((it + 1) & 1) * RTN, nx_h0, nx_w0, tid);
                }
                if (chunk == c_lo) VG_STAMP(1);
                __syncthreads();
                if (chunk == c_lo) VG_STAMP(2);
            }
            const size_t kbase = (size_t)chunk * kc_pad;
            if (g.dbg & 4) continue;
            if constexpr (WL) conv_mfma_chunk<T, MW>(acc, wrow_l + kbase, hb, rowbase, tapoff, koff, ksteps, nt, g.CK, g.CS, lane);
            else conv_mfma_chunk<T, MW>(acc, wrow_g + kbase, hb, rowbase, tapoff, koff, ksteps, nt, g.CK, g.CS, lane);
        }
        VG_STAMP(3);
        bool ks_last = true;
        if constexpr (MC == 0 && !DMA) {
            if (p.ks > 1) {
                // The exchange uses device-scope relaxed atomic stores / loads (write-through to, resp. read from, the level the
                // XCDs share) instead of __threadfence(): an agent-scope fence is buffer_wbl2 + buffer_inv -- it writes back and
                // invalidates this XCD's whole L2 under every kernel of the other streams (measured: 24.7 -> 30.2 ms per step).
                constexpr int SLOT = 256 * MW * 4;                       // floats of one partial tile: lane-linear accumulator dump
                const size_t cell = (size_t)(n * gridDim.y + ntile) * tiles_per_n + tile;
                float* slot = p.ks_part + (cell * p.ks + ks_slice) * SLOT;
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(slot + (i * 4 + r) * 256 + tid, acc[i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this thread's partial has been written through ...
                __syncthreads();                                         // ... every thread's, before the ticket
                int* flag = tapoff + (VG_MAX_TAPS - 1);                  // (the host keeps ntaps < VG_MAX_TAPS for split launches)
                if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.ks_cnt + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                ks_last = *flag == p.ks - 1;
                if (ks_last) {
                    const float* base = p.ks_part + cell * p.ks * SLOT;
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][r] = __hip_atomic_load(base + (i * 4 + r) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int sl = 1; sl < p.ks; ++sl)
#pragma unroll
                        for (int i = 0; i < MW; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[i][r] += __hip_atomic_load(base + (size_t)sl * SLOT + (i * 4 + r) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tid == 0) __hip_atomic_store(p.ks_cnt + cell, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
                }
            }
        }
        if (ks_last && !(g.dbg & 8)) {
#include "vg_conv_epilogue.inc"
        }
        }
        VG_STAMP(4);
    }
    if (p.sums) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = s1[r], b = s2[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if ((lane & 15) == 0) {
                const int cl = wave_n * 16 + 4 * (lane >> 4) + r;
                atomicAdd(&stat[cl * 2], a);
                atomicAdd(&stat[cl * 2 + 1], b);
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int co = ntile * BN + (tid >> 1);
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            if (co < p.Cout) atomicAdd(&p.sums[(((size_t)stripe * gridDim.z + n) * p.Cout + co) * 2 + (tid & 1)], stat[tid]);
        }
        if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.z, p.Cout, gridDim.x * gridDim.y * gridDim.z, (int*)stat);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv32_kernel: the wide-layer flavour (Cin >= 64, Cout a multiple of 64; bf16; weights from L2).  Same staging, tables and
// persistent-tile structure as conv_kernel, but v_mfma_f32_32x32x16_bf16: a wave owns 32 output channels x 32-voxel
// sub-tiles, so one A fragment (32 weight rows x 16 k) and one B fragment (16 k x 32 voxels) feed twice the MACs of the
// 16x16x32 form -- half the LDS traffic and ~40 % fewer instructions per MAC -- and the channel panel is 128 (4 waves along
// the channels) or 64 wide, which halves how often the same input tile is re-staged for another panel.  PMC/ablation on
// D.down0/1/2 before: MFMA phase 40-70 % of the time at 10-19 % MFMA utilisation, staging up to 40 %.
//   acc layout (32x32): lane l holds voxel (l & 31), channels 8*j + 4*(l >> 5) + r  for j = 0..3, r = 0..3  (acc[4*j + r])
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// CP: class-parallel launch of the output-parity classes of a strided data gradient (see conv_kernel, MC == 2)
template <int BN, int MSUB, bool NOISE, bool CP>
__global__ __launch_bounds__(256, 2) void conv32_kernel(const GatherIn g, const ConvOut p, const ConvCls q) {
    typedef bf16_t T;
    constexpr int WN = BN / 32, WM = 4 / WN;            // waves along the channel panel / along the voxels
    constexpr int NSUB = (64 * MSUB) / 32;              // 32-voxel sub-tiles per tile
    static_assert(NSUB % WM == 0, "sub-tiles must divide over the voxel waves");
    constexpr int MW = NSUB / WM;                       // sub-tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave % WN, wave_m = wave / WN;
    const int n = blockIdx.z, ntile = blockIdx.y;
    const int TWm = (1 << g.twl) - 1, THm = (1 << g.thl) - 1;
    const int lv = lane & 31, lk = lane >> 5;
    int bx = blockIdx.x, gx = gridDim.x, t0 = 0, nt = g.ntaps;
    int Ktot = p.Ktot, kc_pad = p.kc_pad;
    const void* wsrc = p.wp;
    int c_od = p.ood, c_oh = p.ooh, c_ow = p.oow, c_OD = p.OD, c_OH = p.OH, c_OW = p.OW;
    if constexpr (CP) {
        const int cls = __builtin_amdgcn_readfirstlane(bx % q.ncls);
        bx /= q.ncls; gx /= q.ncls;
        t0 = q.tap0[cls]; nt = q.tap0[cls + 1] - t0;
        Ktot = q.ktot[cls]; kc_pad = Ktot / p.nchunks;
        wsrc = q.wp[cls];
        c_od = q.off[cls][0]; c_oh = q.off[cls][1]; c_ow = q.off[cls][2];
        c_OD = q.it[cls][0]; c_OH = q.it[cls][1]; c_OW = q.it[cls][2];
    }

    char* halo = smem;
    const int hbytes = g.planar ? (g.CK >> 3) * g.PSB : g.HD * g.DS;
    int* tapoff = (int*)(smem + hbytes);
    float* scs = (float*)(smem + hbytes + 256);
    float* stat = scs + 2 * g.CK;
    int* utab = (int*)(stat + BN * 2);
    const int ncols = stage_ncols(g);
    const int RTN = 3 * stage_axis_len<3>(g);
    int* rtab = utab + 2 * ncols;
    const int c16 = g.CK >> 4;                           // 16-channel K-steps per tap
    const int ksteps = nt * c16;
    int* koff = rtab + 2 * RTN;                          // [K-step][k-group 0/1]: byte offset of the B fragment in the halo image

    if (tid < nt)
        tapoff[tid] = (g.td[t0 + tid] - g.tmin_d) * g.DS + ((g.th[t0 + tid] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[t0 + tid] - g.tmin_w)) * g.VS;
    if (tid < BN * 2) stat[tid] = 0.f;
    build_column_table(g, utab, tid);
    for (int i = tid; i < ksteps * 2; i += 256) {
        const int st = i >> 1, kg = i & 1;
        const int tq = st / c16, cgq = (st - tq * c16) * 2 + kg, tp = t0 + tq;
        koff[i] = (g.td[tp] - g.tmin_d) * g.DS + ((g.th[tp] - g.tmin_h) * g.HWp + halo_pos_w(g, g.tw[tp] - g.tmin_w)) * g.VS + cgq * g.CS;
    }
    int rowbase[MW], ooff[MW], dhw[MW];
    const int co_lane = ntile * BN + wave_n * 32 + 4 * lk;        // first channel of this lane's group 0 (groups are 8 apart)
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = (wave_m * MW + i) * 32 + lv;
        const int w = m & TWm, h = (m >> g.twl) & THm, d = m >> (g.twl + g.thl);
        rowbase[i] = d * g.istr * g.DS + (h * g.istr * g.HWp + w) * g.VS;
        ooff[i] = ((d * p.ostr * p.BH + h * p.ostr) * p.BW + w * p.ostr) * p.Cout + co_lane;
        dhw[i] = d | (h << 10) | (w << 20);
    }
    const glb_ptr<T> wrow = (glb_ptr<T>)wsrc + (size_t)(ntile * BN + wave_n * 32 + lv) * Ktot + 8 * lk;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    if (p.nchunks == 1) stage_scale_shift(g, scs, n, 0, tid);
    const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
    if (bx < tiles_per_n) {
        int t = bx;
        const int tw_i = t % g.tiles_w; t /= g.tiles_w;
        const int th_i = t % g.tiles_h;
        stage_resolve_axes<3>(g, rtab, th_i << g.thl, tw_i << g.twl, tid);
    }
    __syncthreads();

    int gs_w, gs_h, gs_d;
    { int t = gx; gs_w = t % g.tiles_w; t /= g.tiles_w; gs_h = t % g.tiles_h; gs_d = t / g.tiles_h; }
    int ti_w, ti_h, ti_d;
    { int t = bx; ti_w = t % g.tiles_w; t /= g.tiles_w; ti_h = t % g.tiles_h; ti_d = t / g.tiles_h; }
    typedef const __attribute__((address_space(1))) bf16x8 glb_frag;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    int it = -1;
    for (int tile = bx; tile < tiles_per_n; tile += gx) {
        ++it;
        const int od0 = ti_d << g.tdl, oh0 = ti_h << g.thl, ow0 = ti_w << g.twl;
        ti_w += gs_w; if (ti_w >= g.tiles_w) { ti_w -= g.tiles_w; ++ti_h; }
        ti_h += gs_h; if (ti_h >= g.tiles_h) { ti_h -= g.tiles_h; ++ti_d; }
        ti_d += gs_d;
        f32x16 acc[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            __syncthreads();                       // previous readers of the halo tile are done; this tile's axis tables visible
            if (p.nchunks > 1) { stage_scale_shift(g, scs, n, chunk, tid); __syncthreads(); }
            stage_halo_tile<T, NOISE, 4, 3>(g, halo, scs, utab, rtab + (it & 1) * RTN, n, od0, chunk, tid);
            if (chunk == 0 && tile + gx < tiles_per_n)
                stage_resolve_axes<3>(g, rtab + ((it + 1) & 1) * RTN, ti_h << g.thl, ti_w << g.twl, tid);
            __syncthreads();
            // ---- K loop: ring of RD weight fragments (L2 latency), halo fragments ping-pong ----
            const glb_ptr<T> w = wrow + (size_t)chunk * kc_pad;
            const int last = ksteps - 1;
            constexpr int RD = 8;                  // weight-fragment ring depth: one workgroup per CU has no other wave to hide L2 latency
            bf16x8 a[RD];
#pragma unroll
            for (int u = 0; u < RD; ++u) a[u] = *(glb_frag*)(w + min(u, last) * 16);
            bf16x8 bb[2][MW];
            {
                const int o0 = koff[lk];
#pragma unroll
                for (int i = 0; i < MW; ++i) bb[0][i] = *(const bf16x8*)(halo + rowbase[i] + o0);
            }
            int on1 = koff[min(1, last) * 2 + lk];
            for (int s = 0; s < ksteps; s += RD) {
#pragma unroll
                for (int u = 0; u < RD; ++u) {
                    const int on2 = koff[min(s + u + 2, last) * 2 + lk];
#pragma unroll
                    for (int i = 0; i < MW; ++i) bb[(u + 1) & 1][i] = *(const bf16x8*)(halo + rowbase[i] + on1);
                    if (s + u >= ksteps) a[u] = zero8;                 // phantom steps of the last group add zero
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MW; ++i) acc[i] = VG_MFMA32(a[u], bb[u & 1][i], acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    a[u] = *(glb_frag*)(w + min(s + u + RD, last) * 16);
                    on1 = on2;
                }
            }
        }
        // ---- epilogue: four groups of 4 consecutive channels per lane and sub-tile ----
        const size_t tbase = (((size_t)(n * p.BD + od0 * p.ostr + c_od) * p.BH + oh0 * p.ostr + c_oh) * p.BW + ow0 * p.ostr + c_ow) * p.Cout;
        const int remd = c_OD - od0, remh = c_OH - oh0, remw = c_OW - ow0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co_lane + 8 * j;
            if (co >= p.Cout) continue;
            float bias4[4] = {0.f, 0.f, 0.f, 0.f}, rs4[4], rb4[4];
            if (p.bias) { const f32x4 b4 = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(p.bias + co); bias4[0] = b4[0]; bias4[1] = b4[1]; bias4[2] = b4[2]; bias4[3] = b4[3]; }
            if (p.res) {
                const f32x4 r4 = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(p.rs + n * p.Cout + co);
                const f32x4 q4 = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(p.rb + n * p.Cout + co);
                for (int r = 0; r < 4; ++r) { rs4[r] = r4[r]; rb4[r] = q4[r]; }
            }
            // the residual / accumulate operands of all the group's sub-tiles are requested up front: loaded where they are used, every sub-tile
            // waited its own round trip (the loads cannot move above the stores of the sub-tiles before: p.out and p.res may alias)
            bool inr_[MW];
            float addv[MW][4];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                inr_[i] = (dhw[i] & 1023) < remd && ((dhw[i] >> 10) & 1023) < remh && (dhw[i] >> 20) < remw;
#pragma unroll
                for (int r = 0; r < 4; ++r) addv[i][r] = bias4[r];
            }
            if (p.res) {
                Vec4<T> rv[MW];
#pragma unroll
                for (int i = 0; i < MW; ++i) vec4_load(rv[i], (const T*)p.res + tbase + (inr_[i] ? ooff[i] : 0) + 8 * j);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    float x[4]; vec4_unpack(rv[i], x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) addv[i][r] += x[r] * rs4[r] + rb4[r];
                }
            }
            if (p.accumulate) {
                if (p.out_f32) {
                    Vec4<float> ov[MW];
#pragma unroll
                    for (int i = 0; i < MW; ++i) vec4_load(ov[i], (const float*)p.out + tbase + (inr_[i] ? ooff[i] : 0) + 8 * j);
#pragma unroll
                    for (int i = 0; i < MW; ++i) { float x[4]; vec4_unpack(ov[i], x); for (int r = 0; r < 4; ++r) addv[i][r] += x[r]; }
                } else {
                    Vec4<bf16_t> ov[MW];
#pragma unroll
                    for (int i = 0; i < MW; ++i) vec4_load(ov[i], (const bf16_t*)p.out + tbase + (inr_[i] ? ooff[i] : 0) + 8 * j);
#pragma unroll
                    for (int i = 0; i < MW; ++i) { float x[4]; vec4_unpack(ov[i], x); for (int r = 0; r < 4; ++r) addv[i][r] += x[r]; }
                }
            }
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                if (!inr_[i]) continue;
                const size_t o = tbase + ooff[i] + 8 * j;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][4 * j + r] + addv[i][r];
                if (p.out_f32) {
                    *(f32x4*)((float*)p.out + o) = (f32x4){v[0], v[1], v[2], v[3]};
                } else {
                    const bf16x4 pk = {(short)f2bf(v[0]), (short)f2bf(v[1]), (short)f2bf(v[2]), (short)f2bf(v[3])};
                    *(bf16x4*)((bf16_t*)p.out + o) = pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = bf2f((bf16_t)pk[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[4 * j + r] += v[r]; s2[4 * j + r] += v[r] * v[r]; }
            }
        }
    }
    if (p.sums) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = s1[r], b = s2[r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (lv == 0) {
                const int cl = wave_n * 32 + 8 * (r >> 2) + 4 * lk + (r & 3);
                atomicAdd(&stat[cl * 2], a);
                atomicAdd(&stat[cl * 2 + 1], b);
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int co = ntile * BN + (tid >> 1);
            const int stripe = blockIdx.x & (VG_STRIPES - 1);
            if (co < p.Cout) atomicAdd(&p.sums[(((size_t)stripe * gridDim.z + n) * p.Cout + co) * 2 + (tid & 1)], stat[tid]);
        }
        if (p.fin.ticket) vg_fin_tail(p.fin, p.sums, gridDim.z, p.Cout, gridDim.x * gridDim.y * gridDim.z, (int*)stat);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// K-split scratch: the caller's vg_conv_desc::scratch of the issuing stream (launches of one stream are ordered, launches of
// different streams may overlap and must not share it): [VG_SCRATCH_CTR_BYTES of arrival counters][fp32 partial tiles].  The
// caller zeroes the counters once; every launch leaves them at zero.  No scratch (or too small a one): the launch runs unsplit.
#define VG_KS_CELLS (VG_SCRATCH_CTR_BYTES / 4)

static int conv_lds_bytes(const GatherIn& g, int BN, int CK, int wbytes, int dma = 0, int ksteps_total = 0) {
    const int nunits = dma ? 5 * (g.DS >> 4) : stage_table_ints(g);
    const int ksteps = ksteps_total > 0 ? ksteps_total : (g.ntaps * (CK >> 3) + 3) >> 2;
    return (dma ? 2 : 1) * halo_bytes(g) + 256 + (dma ? 4 : 2) * CK * 4 + BN * 2 * 4 + nunits * 4 + ksteps * 16 + 16 + wbytes;
}

static int fill_conv(const vg_conv_desc* d, GatherIn& g, ConvOut& k, ConvCls& q, int& BN, int& MSUB, int& lds) {
    if (!d || !d->out || !d->wpacked) return VG_EINVAL;
    if (d->Cout < 1 || (d->Cout != 1 && (d->Cout % 4))) return VG_EINVAL;
    if (d->ostr < 1 || d->ostr > 2) return VG_EINVAL;
    if (d->res && (!d->res_scale || !d->res_shift)) return VG_EINVAL;
    const int Cin = d->c_src0 + d->c_src1;
    const int Cpad = ((Cin + d->CK - 1) / d->CK) * d->CK;
    k.nchunks = Cpad / d->CK;
    k.kc_pad = ((d->ntaps * d->CK + 31) / 32) * 32;
    k.Ktot = k.nchunks * k.kc_pad;
    const int esz = d->f32 ? 4 : 2;
    k.WRS = k.Ktot * esz + 16;
    // output-parity classes
    const int gpc_ = d->CK >> 3;
    q.ncls = d->nclass > 1 ? d->nclass : 1;
    if (q.ncls > 8) return VG_EINVAL;
    // several channel chunks (or fp32 / noise sources): the classes cannot share one staged halo tile and its accumulators;
    // they run class-parallel instead -- one launch, workgroup -> (class, tile walker)
    q.par = (q.ncls > 1 && (k.nchunks != 1 || d->f32 || d->noise)) ? 1 : 0;
    int wrow_bytes = 0;                 // LDS bytes of one row of every class panel
    int ksteps_par = 0;
    if (q.par) {
        if (d->cls_tap0[0] != 0 || d->cls_tap0[q.ncls] != d->ntaps) return VG_EINVAL;
        for (int c = 0; c < q.ncls; ++c) {
            const int nt = d->cls_tap0[c + 1] - d->cls_tap0[c];
            if (nt < 1 || !d->cls_w[c]) return VG_EINVAL;
            q.tap0[c] = d->cls_tap0[c]; q.wp[c] = d->cls_w[c];
            q.ktot[c] = k.nchunks * (((nt * d->CK + 31) / 32) * 32);
            q.woff[c] = 0; q.ks0[c] = 0;
            for (int a = 0; a < 3; ++a) { q.off[c][a] = d->cls_ooff[c][a]; q.it[c][a] = d->cls_iters[c][a]; }
            if (d->cls_iters[c][0] > d->OD || d->cls_iters[c][1] > d->OH || d->cls_iters[c][2] > d->OW) return VG_EINVAL;
            const int rb = q.ktot[c] * esz + 16, ks = (nt * gpc_ + 3) >> 2;
            if (rb > wrow_bytes) wrow_bytes = rb;
            if (ks > ksteps_par) ksteps_par = ks;
        }
        q.tap0[q.ncls] = d->ntaps; q.ks0[q.ncls] = ksteps_par;
    } else if (q.ncls == 1) {
        q.tap0[0] = 0; q.tap0[1] = d->ntaps; q.wp[0] = d->wpacked; q.ktot[0] = k.Ktot; q.woff[0] = 0;
        q.ks0[0] = 0; q.ks0[1] = (d->ntaps * gpc_ + 3) >> 2;
        q.off[0][0] = d->ooff_d; q.off[0][1] = d->ooff_h; q.off[0][2] = d->ooff_w;
        q.it[0][0] = d->OD; q.it[0][1] = d->OH; q.it[0][2] = d->OW;
        wrow_bytes = k.WRS;
    } else {
        if (k.nchunks != 1 || d->cls_tap0[0] != 0 || d->cls_tap0[q.ncls] != d->ntaps) return VG_EINVAL;
        if (d->f32 || d->noise) return VG_EINVAL;              // fused classes: bf16, noise-free sources only
        q.ks0[0] = 0;
        for (int c = 0; c < q.ncls; ++c) {
            const int nt = d->cls_tap0[c + 1] - d->cls_tap0[c];
            if (nt < 1 || !d->cls_w[c]) return VG_EINVAL;
            q.tap0[c] = d->cls_tap0[c]; q.wp[c] = d->cls_w[c];
            q.ktot[c] = ((nt * d->CK + 31) / 32) * 32;
            q.ks0[c + 1] = q.ks0[c] + ((nt * gpc_ + 3) >> 2);
            for (int a = 0; a < 3; ++a) { q.off[c][a] = d->cls_ooff[c][a]; q.it[c][a] = d->cls_iters[c][a]; }
            if (d->cls_iters[c][0] > d->OD || d->cls_iters[c][1] > d->OH || d->cls_iters[c][2] > d->OW) return VG_EINVAL;
            wrow_bytes += q.ktot[c] * esz + 16;
        }
        q.tap0[q.ncls] = d->ntaps;
    }
    const int ksteps_total = q.ks0[q.ncls];
    // Tile choice.  Efficiency wants a wide channel panel (BN) and a big voxel tile (MSUB*64: fewer halo voxels per
    // output voxel, more MFMAs per staged byte); the chip wants >= ~2 workgroups per CU.  Small grids (16^3, 8^3 levels)
    // therefore take narrow panels / small tiles.  Within a class, prefer <= 80 KiB of LDS (two workgroups per CU).
    const int max_ms16 = vg_tune("CONV_MS16", 8);
    const int use_dma = vg_tune("CONV_DMA", 0);     // measured neutral (51.5 vs 51.7 ms/step): these kernels are issue-bound, not latency-bound
    const int force_msub = vg_tune("CONV_MSUB", 0), no_wlds = vg_tune("CONV_NOWLDS", 0), force_bn = vg_tune("CONV_BN", 0);
    const int bn_max = d->Cout <= 16 ? 16 : (d->Cout <= 32 ? 32 : 64);
    const long fill_t = vg_tune("CONV_FILL", 128);
    int found = 0, rc = VG_ELDS;
    long best_score = -1;
    int best_bn = 0, best_ms = 0, best_wl = 0, best_lds = 0, best_dma = 0;
    // LDS-DMA staging: bf16 planar image of a multi-channel, noise-free source, weights resident in LDS
    const bool dma_ok = use_dma && !d->f32 && Cin != 1 && !d->noise && d->istr == 1 && d->CK <= 48 && q.ncls == 1;
    for (int bn = bn_max; bn >= 16; bn >>= 1) {
        if (force_bn && bn != force_bn && bn != bn_max) continue;
        for (int ms = (bn == 16 ? max_ms16 : (bn == 32 ? 4 : 2)); ms >= 1; ms >>= 1) {
            if (force_msub && ms != force_msub) continue;
            if (ms == 8 && d->f32) continue;
            // fused classes keep a second set of per-class state live: their 8-sub-tile variants spill ~80 VGPRs (324 B of
            // scratch per lane) and lose to the next smaller tile (enc1.cb1 data gradient at 128^3: 0.144 -> 0.084 ms)
            if (q.ncls > 1 && !q.par && (bn / 16) * ms >= 8) continue;
            rc = fill_gather(d, g, d->CK, 64 * ms);
            if (rc != VG_OK) return rc;
            const long wgs = (long)g.tiles_d * g.tiles_h * g.tiles_w * ((d->Cout + bn - 1) / bn) * d->N * (q.par ? q.ncls : 1);
            const int wbytes = bn * wrow_bytes;
            int wl = (wbytes <= 56 * 1024 && !no_wlds && !d->f32) ? 1 : 0;       // exact-parity mode reads weights from L2
            int need = conv_lds_bytes(g, bn, d->CK, wl ? wbytes : 0, 0, ksteps_total);
            if (need > 80 * 1024 && wl) { const int n2 = conv_lds_bytes(g, bn, d->CK, 0, 0, ksteps_total); if (n2 <= 80 * 1024 || need > VG_LDS_LIMIT) { wl = 0; need = n2; } }
            if (need > VG_LDS_LIMIT) continue;
            if (ms == 8 && need > 80 * 1024) continue;     // the 512-voxel tile only pays while two workgroups stay resident
            int dma = 0;
            if (dma_ok && wl) {
                GatherIn g2; rc = fill_gather(d, g2, d->CK, 64 * ms, 0, 1);
                if (rc != VG_OK) return rc;
                const int need2 = conv_lds_bytes(g2, bn, d->CK, wbytes, 1, ksteps_total);
                if (g2.planar && need2 <= 80 * 1024) { dma = 1; need = need2; }
            }
            // score: reaching 512 workgroups dominates, then work per workgroup-tile (bn*ms), then small LDS
            // sweep of the 128^3 train step: 512 -> 31.5 ms, 256 -> 30.9, 128 -> 30.6, 64 -> 30.7 (the other lane and the side streams fill the chip)
            const long fill = wgs >= fill_t ? fill_t : wgs;
            // (useful channels per panel, not the panel width: a 64-wide panel on 48 or 96 output channels multiplies zeros in a
            // quarter of its MFMAs -- there the 32-wide panel with the twice larger voxel tile wins: 16->48 data gradient at 128^3
            // 0.36 -> 0.32 ms)
            const int ny_ = (d->Cout + bn - 1) / bn;
            const long useful = (long)((d->Cout + ny_ - 1) / ny_) * ms;
            const long score = fill * 100000 + useful * 100 + (need <= 80 * 1024 ? 50 : 0);
            if (score > best_score) { best_score = score; best_bn = bn; best_ms = ms; best_wl = wl; best_lds = need; best_dma = dma; found = 1; }
        }
    }
    if (found) { BN = best_bn; MSUB = best_ms; k.w_lds = best_wl; k.dma = best_dma; lds = best_lds; }
    if (!found) return VG_ELDS;
    rc = fill_gather(d, g, d->CK, 64 * MSUB, 0, k.dma);
    if (rc != VG_OK) return rc;
    if (q.ncls > 1 && !q.par) { int off = 0; for (int c = 0; c < q.ncls; ++c) { q.woff[c] = off; off += BN * (q.ktot[c] * esz + 16); } }
    k.OD = d->OD; k.OH = d->OH; k.OW = d->OW; k.ostr = d->ostr; k.ood = d->ooff_d; k.ooh = d->ooff_h; k.oow = d->ooff_w;
    k.BD = d->BD; k.BH = d->BH; k.BW = d->BW; k.Cout = d->Cout;
    k.wp = d->wpacked;
    k.bias = d->bias; k.res = d->res; k.rs = d->res_scale; k.rb = d->res_shift; k.tanh_out = d->tanh_out;
    k.res1 = (d->res && d->res_c1) ? 1 : 0;
    k.wdma = 0;
    k.wp_up = d->wpacked_up; k.nup = (d->wpacked_up && d->src0_shift && d->pad_mode == VG_PAD_REFLECT && d->c_src0 > 0 && (d->c_src0 % 16) == 0) ? d->c_src0 / 16 : 0;
    if (d->res_c1 && (!d->res || d->ostr != 1 || q.ncls > 1)) return VG_EINVAL;
    k.out = d->out; k.out_f32 = (d->out_f32 || d->f32) ? 1 : 0; k.accumulate = d->accumulate; k.sums = d->out_sums;
    k.bs_x0 = nullptr; k.xw = 0;
    k.ks = 1; k.ks_part = nullptr; k.ks_cnt = nullptr;
    k.fin = vg_fin_of(d);
    k.scratch = (char*)d->scratch; k.scratch_bytes = d->scratch ? d->scratch_bytes : 0;
    if (const vg_actnorm_bwd_desc* b = d->bstat) {       // validated by vg_conv3d
        k.bs_x0 = b->x; k.bs_x1 = b->x1; k.bs_c0 = b->x1 ? b->c_x0 : b->C; k.bs_sh = b->x1 ? (b->x0_shift ? 1 : 0) : 0;
        k.bs_act = b->act; k.bs_pad = b->g_padded ? 1 : 0; k.bs_D = b->D; k.bs_H = b->H; k.bs_W = b->W;
        k.bs_sc = b->scale; k.bs_sf = b->shift; k.bs_mu = b->mean; k.bs_rs = b->rstd; k.bs_ml = b->mult;
    }
    return VG_OK;
}

// ---- wide-layer flavour (conv32_kernel): eligibility and tile plan ----
static int plan_conv32(const vg_conv_desc* d, const ConvOut& k, const ConvCls& q, GatherIn& g, int& BN, int& MSUB, int& lds) {
    const int use32 = vg_tune("CONV32", 1);
    const int Cin = d->c_src0 + d->c_src1;
    if (!use32 || d->f32 || d->res_c1 || (q.ncls != 1 && !q.par) || (d->Cout % 64) || Cin < 64 || Cin == 1 || d->tanh_out || (d->CK % 16)) return VG_EINVAL;
    const int ncp = q.par ? q.ncls : 1;
    if ((long)d->OD * d->OH * d->OW * d->N * (d->Cout / 64) * ncp < 256 * 64) return VG_EINVAL;     // too small to fill the chip with 64-wide panels
    BN = (d->Cout % 128 == 0) ? 128 : 64;
    int tmax = d->ntaps;
    if (q.par) { tmax = 0; for (int c = 0; c < q.ncls; ++c) tmax = std::max(tmax, q.tap0[c + 1] - q.tap0[c]); }
    const int ksteps = tmax * (d->CK >> 4);
    long best = -1; int best_ms = 0, best_lds = 0;
    const long fill32 = vg_tune("CONV32_FILL", 256);
    for (int ms = (BN == 128 ? 2 : 4); ms >= (BN == 128 ? 1 : 2); ms >>= 1) {
        int rc = fill_gather(d, g, d->CK, 64 * ms);
        if (rc != VG_OK) return rc;
        const int need = halo_bytes(g) + 256 + 2 * d->CK * 4 + BN * 2 * 4 + stage_table_ints(g) * 4 + ksteps * 8 + 16;
        if (need > VG_LDS_LIMIT) continue;
        const long wgs = (long)g.tiles_d * g.tiles_h * g.tiles_w * (d->Cout / BN) * d->N * ncp;
        // every tile streams its BN x K weight panel from L2 (64 B/clk per CU): 64 voxels per tile give exactly the
        // 64 FLOP/B that the MFMA rate needs, 128 voxels give headroom -- so one workgroup per CU with the big tile beats
        // two with the small one
        const long fill = wgs >= fill32 ? fill32 : wgs;
        const long score = fill * 1000 + ms * 10 + (need <= 80 * 1024 ? 5 : 0);
        if (score > best) { best = score; best_ms = ms; best_lds = need; }
    }
    if (best < 0) return VG_ELDS;
    MSUB = best_ms; lds = best_lds;
    (void)k;
    return fill_gather(d, g, d->CK, 64 * MSUB);
}
template <int BN, int MSUB, bool NOISE, bool CP>
static int launch_conv32b(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv32_kernel<BN, MSUB, NOISE, CP>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    int per_cu = 2;
    if (lds > 0 && VG_LDS_LIMIT / lds < per_cu) per_cu = VG_LDS_LIMIT / lds;
    if (per_cu < 1) per_cu = 1;
    const int tiles = g.tiles_d * g.tiles_h * g.tiles_w;
    const int ny = k.Cout / BN;
    const int ncp = CP ? q.ncls : 1;
    int bx = 256 * per_cu / (ny * g.N * ncp); if (bx < 1) bx = 1; if (bx > tiles) bx = tiles;
    // variant name for the coverage tests: template arguments + the two run-time regimes that change which code runs
    // (walk: a workgroup visits more than one tile; ch: several channel chunks per tile)
    if (vg_dry("conv32<%d,%d,n%d,cp%d>|walk%d|ch%d", BN, MSUB, (int)NOISE, (int)CP, tiles > bx ? 1 : 0, k.nchunks > 1 ? 1 : 0)) return VG_OK;
    hipLaunchKernelGGL((conv32_kernel<BN, MSUB, NOISE, CP>), dim3(bx * ncp, ny, g.N), dim3(256), lds, s, g, k, q);
    if (k.fin.ticket && k.sums) vg_fin_done = true;
    return vg_check_launch();
}
static int launch_conv32(const GatherIn& g, const ConvOut& k, const ConvCls& q, int BN, int MSUB, int lds, hipStream_t s) {
    const bool nz = g.noise != nullptr;
    if (q.par) {                                  // data gradients: noise-free sources
        if (nz) return VG_EINVAL;
        if (BN == 128) return MSUB == 2 ? launch_conv32b<128, 2, false, true>(g, k, q, lds, s) : launch_conv32b<128, 1, false, true>(g, k, q, lds, s);
        return MSUB == 4 ? launch_conv32b<64, 4, false, true>(g, k, q, lds, s) : launch_conv32b<64, 2, false, true>(g, k, q, lds, s);
    }
    if (BN == 128) {
        if (MSUB == 2) return nz ? launch_conv32b<128, 2, true, false>(g, k, q, lds, s) : launch_conv32b<128, 2, false, false>(g, k, q, lds, s);
        return nz ? launch_conv32b<128, 1, true, false>(g, k, q, lds, s) : launch_conv32b<128, 1, false, false>(g, k, q, lds, s);
    }
    if (MSUB == 4) return nz ? launch_conv32b<64, 4, true, false>(g, k, q, lds, s) : launch_conv32b<64, 4, false, false>(g, k, q, lds, s);
    return nz ? launch_conv32b<64, 2, true, false>(g, k, q, lds, s) : launch_conv32b<64, 2, false, false>(g, k, q, lds, s);
}

extern "C" int vg_conv3d_plan(const vg_conv_desc* d, int32_t* plan4) {
    vg_begin();
    if (!plan4) return VG_EINVAL;
    GatherIn g; ConvOut k; ConvCls q; int BN, MSUB, lds;
    int rc = fill_conv(d, g, k, q, BN, MSUB, lds);
    if (rc != VG_OK) return rc;
    { GatherIn g2; int bn2, ms2, lds2; if (plan_conv32(d, k, q, g2, bn2, ms2, lds2) == VG_OK) { g = g2; BN = bn2; MSUB = ms2; lds = lds2; } }
    plan4[0] = BN; plan4[1] = 64 * MSUB; plan4[2] = lds;
    plan4[3] = g.tiles_d * g.tiles_h * g.tiles_w * ((d->Cout + BN - 1) / BN) * d->N;
    return VG_OK;
}

extern "C" int vg_conv3d_variant(const vg_conv_desc* d, char* buf, int buflen) {
    if (!buf || buflen < 64) return VG_EINVAL;
    vg_dry_begin(buf, buflen);
    const int rc = vg_conv3d(d, nullptr);
    vg_dry_end();
    return rc;
}

extern "C" int vg_conv3d_lds_bytes(const vg_conv_desc* d) {
    vg_begin();
    GatherIn g; ConvOut k; ConvCls q; int BN, MSUB, lds;
    int rc = fill_conv(d, g, k, q, BN, MSUB, lds);
    if (rc == VG_OK) { GatherIn g2; int bn2, ms2, lds2; if (plan_conv32(d, k, q, g2, bn2, ms2, lds2) == VG_OK) lds = lds2; }
    return rc == VG_OK ? lds : rc;
}

template <typename T, int BN, int MSUB, bool NOISE, bool WL, bool DMA, int MC, bool C1 = false>
static int launch_conv3(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_kernel<T, BN, MSUB, NOISE, WL, DMA, MC, C1>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    // persistent grid = what is resident at once (register cap: 3 workgroups per CU, 2 for the 8-sub-tile variants; LDS):
    // every further workgroup would repeat the per-workgroup prologue (weight panel, tables) for fewer tiles each
    const int wg_env = vg_tune("CONV_WGS", 0);
    int per_cu = ((BN / 16) * MSUB >= VG_CONV_MW2) ? 2 : VG_CONV_WAVES;
    if (lds > 0 && VG_LDS_LIMIT / lds < per_cu) per_cu = VG_LDS_LIMIT / lds;
    if (per_cu < 1) per_cu = 1;
    const int wg_target = wg_env > 0 ? wg_env : 256 * per_cu;
    const int tiles = g.tiles_d * g.tiles_h * g.tiles_w;
    const int ny = (k.Cout + BN - 1) / BN;
    const int ncp = MC == 2 ? q.ncls : 1;
    int bx = wg_target / (ny * g.N * ncp); if (bx < 1) bx = 1; if (bx > tiles) bx = tiles;
    // K split (see the kernel): grids that leave most of the chip empty and have several channel chunks per tile.  Not with the weight
    // panel in LDS (every slice would copy the whole panel: dec3.short 29 -> 44 us).  Solo times at 128^3 (rocprofv3, serial schedule):
    // 256->256 at 8^3 50.8 -> 27.1 us, the 16^3 forward layers 42.9 -> 33.0, D.out 99.8 -> 60.1, the 8^3 data gradients 52.1 -> 48.1:
    // -0.7 ms of kernel time per step -- and no change of the concurrent step (24.2 ms either way): these launches occupy a
    // fraction of the chip and their latency is covered by the other lane and the weight-gradient streams.
    int ks = 1;
    const long cells = (long)tiles * ny * g.N;
    const long ks_part_bytes = k.scratch_bytes - VG_SCRATCH_CTR_BYTES;          // <= 0: no scratch from the caller, no split
    if (MC == 0 && !DMA && !WL && k.nchunks > 1 && g.ntaps < VG_MAX_TAPS && bx == tiles && cells <= vg_tune("CONV_KSPLIT_CELLS", 256) && cells <= VG_KS_CELLS) {
        const int cap = vg_tune("CONV_KSPLIT", 8);
        constexpr long SLOTB = 256L * (MSUB * (BN / 16)) * 16;
        for (int c = k.nchunks; c >= 2; --c)
            if (k.nchunks % c == 0 && c <= cap && cells * c <= 1024 && cells * c * SLOTB <= ks_part_bytes) { ks = c; break; }
    }
    if (vg_dry("conv<%s,%d,%d,n%d,wl%d,dma%d,mc%d,c1%d>|walk%d|ch%d|ks%d", sizeof(T) == 4 ? "f32" : "bf16", BN, MSUB, (int)NOISE, (int)WL, (int)DMA,
               MC, (int)C1, tiles > bx ? 1 : 0, k.nchunks > 1 ? 1 : 0, ks)) return VG_OK;
    ConvOut k2 = k;
    k2.ks = ks; k2.ks_part = nullptr; k2.ks_cnt = nullptr;
    if (ks > 1) { k2.ks_cnt = (unsigned*)k.scratch; k2.ks_part = (float*)(k.scratch + VG_SCRATCH_CTR_BYTES); }
    dim3 grid(bx * ncp * ks, ny, g.N);
    hipLaunchKernelGGL((conv_kernel<T, BN, MSUB, NOISE, WL, DMA, MC, C1>), grid, dim3(256), lds, s, g, k2, q);
    if (k2.fin.ticket && k2.sums) vg_fin_done = true;
    return vg_check_launch();
}
template <typename T, int BN, int MSUB, bool NOISE>
static int launch_conv2(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    if (g.Cin == 1) {                                     // single-channel source: its staging path is a kernel variant of its own
        if (q.ncls > 1) return VG_EINVAL;
        if constexpr (sizeof(T) == 2) { if (k.w_lds) return launch_conv3<T, BN, MSUB, NOISE, true, false, 0, true>(g, k, q, lds, s); }
        return launch_conv3<T, BN, MSUB, NOISE, false, false, 0, true>(g, k, q, lds, s);
    }
    if constexpr (!NOISE) {
        if (q.par) {                                      // class-parallel data gradient (noise-free sources)
            if constexpr (sizeof(T) == 2) { if (k.w_lds) return launch_conv3<T, BN, MSUB, NOISE, true, false, 2>(g, k, q, lds, s); }
            return launch_conv3<T, BN, MSUB, NOISE, false, false, 2>(g, k, q, lds, s);
        }
    }
    if constexpr (sizeof(T) == 2 && !NOISE) {
        if (q.ncls > 1) return k.w_lds ? launch_conv3<T, BN, MSUB, NOISE, true, false, 1>(g, k, q, lds, s)
                                       : launch_conv3<T, BN, MSUB, NOISE, false, false, 1>(g, k, q, lds, s);
        if (k.w_lds && k.dma) return launch_conv3<T, BN, MSUB, NOISE, true, true, 0>(g, k, q, lds, s);
    }
    if (q.ncls > 1) return VG_EINVAL;                     // several classes: noise-free sources only
    if constexpr (sizeof(T) == 2) { if (k.w_lds) return launch_conv3<T, BN, MSUB, NOISE, true, false, 0>(g, k, q, lds, s); }
    return launch_conv3<T, BN, MSUB, NOISE, false, false, 0>(g, k, q, lds, s);
}
template <typename T, int BN, int MSUB>
static int launch_conv(const GatherIn& g, const ConvOut& k, const ConvCls& q, int lds, hipStream_t s) {
    return g.noise ? launch_conv2<T, BN, MSUB, true>(g, k, q, lds, s) : launch_conv2<T, BN, MSUB, false>(g, k, q, lds, s);
}
template <typename T>
static int dispatch_conv(const GatherIn& g, const ConvOut& k, const ConvCls& q, int BN, int MSUB, int lds, hipStream_t s) {
    if (BN == 16 && MSUB == 8) { if constexpr (sizeof(T) == 2) return launch_conv<T, 16, 8>(g, k, q, lds, s); else return VG_EINVAL; }
    if (BN == 16) return MSUB == 4 ? launch_conv<T, 16, 4>(g, k, q, lds, s) : (MSUB == 2 ? launch_conv<T, 16, 2>(g, k, q, lds, s) : launch_conv<T, 16, 1>(g, k, q, lds, s));
    if (BN == 32) return MSUB == 4 ? launch_conv<T, 32, 4>(g, k, q, lds, s) : (MSUB == 2 ? launch_conv<T, 32, 2>(g, k, q, lds, s) : launch_conv<T, 32, 1>(g, k, q, lds, s));
    return MSUB == 2 ? launch_conv<T, 64, 2>(g, k, q, lds, s) : launch_conv<T, 64, 1>(g, k, q, lds, s);
}

static inline bool thin2_shape(const vg_conv_desc* d, const ConvOut& k, const ConvCls& q) {
    return !d->f32 && d->CK == 16 && d->Cout >= 32 && (d->Cout % 32) == 0 && q.ncls == 1 && d->ntaps == 27;
}

// Shape-only: 2 when the two-panel instance of the thin-channel specialist (conv_thin_kernel<..., NP = 2>) serves this convolution
// with 16-channel chunks (d->CK is ignored) -- the caller then packs the weights with CK = 16; else 0.
extern "C" int vg_conv3d_thin_np(const vg_conv_desc* d0) {
    vg_begin();
    if (!d0) return 0;
    vg_conv_desc d = *d0; d.CK = 16;
    GatherIn g, g2; ConvOut k; ConvCls q; int BN, MSUB, lds;
    if (fill_conv(&d, g, k, q, BN, MSUB, lds) != VG_OK || !thin2_shape(&d, k, q)) return 0;
    if (fill_gather(&d, g2, 16, 512) != VG_OK || !vg_conv_thin_ok(&d, g2, k, q, 2)) return 0;
    return vg_conv_thin_lds_bytes(g2, 2) <= VG_LDS_LIMIT ? 2 : 0;
}

// did_stats: set when the launched kernel accumulated the IN-backward statistics of d->bstat itself (striped, unfolded)
static int conv3d_impl(const vg_conv_desc* d, vg_stream_t stream, bool& did_stats) {
    if (d && d->wlayout) return vg_conv_dma(d, (hipStream_t)stream, &did_stats);   // LDS-DMA family (weights in its block layout): served there or an error
    if (d && d->out && d->wpacked && d->src0 && !d->res_c1) {           // 1x1x1 with a single channel on one side: HBM-bound VALU kernels
        const int prc = vg_pointwise_conv(d, (hipStream_t)stream);
        if (prc <= 0) return prc;
    }
    GatherIn g; ConvOut k; ConvCls q; int BN, MSUB, lds;
    int rc = fill_conv(d, g, k, q, BN, MSUB, lds);
    if (rc != VG_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!d->res_c1 && thin2_shape(d, k, q)) {        // 32-channel panels of the thin-channel specialist (weights packed with 16-channel chunks for it)
        GatherIn g2;
        if (fill_gather(d, g2, 16, 512) == VG_OK && vg_conv_thin_ok(d, g2, k, q, 2)) {
            const int trc = vg_launch_conv_thin(g2, k, 2, s, d->bstat ? d->bstat->red : nullptr, did_stats);
            if (trc <= 0 && trc != VG_ELDS) return trc;      // (its whole-grid axis tables do not fit the LDS on an enormous grid: the gather kernels serve it)
        }
    }
    { GatherIn g2; int bn2, ms2, lds2; if (plan_conv32(d, k, q, g2, bn2, ms2, lds2) == VG_OK) return launch_conv32(g2, k, q, bn2, ms2, lds2, s); }
    if (!d->f32 && MSUB == 8 && BN == 16 && vg_conv_thin_ok(d, g, k, q, 1)) {
        const int trc = vg_launch_conv_thin(g, k, 1, s, d->bstat ? d->bstat->red : nullptr, did_stats);
        if (trc <= 0 && trc != VG_ELDS) return trc;
    }
    return d->f32 ? dispatch_conv<float>(g, k, q, BN, MSUB, lds, s) : dispatch_conv<bf16_t>(g, k, q, BN, MSUB, lds, s);
}

thread_local bool vg_fin_done = false;
__global__ void fin_kernel(VgFin f, const float* sums, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f, ss = 0.f;
    for (int t = 0; t < VG_STRIPES; ++t) { const float* p = sums + (((size_t)t * N + n) * C + c) * 2; s += p[0]; ss += p[1]; }
    vg_fin_one(f, s, ss, n, c);
}
int vg_launch_fin(const vg_conv_desc* d, hipStream_t s) {
    const VgFin f = vg_fin_of(d);
    if (!f.njobs) return VG_OK;
    const int total = d->N * d->Cout;
    hipLaunchKernelGGL(fin_kernel, dim3((total + 255) / 256), dim3(256), 0, s, f, (const float*)d->out_sums, d->N, d->Cout);
    return vg_check_launch();
}
static int fin_check(const vg_conv_desc* d) {
    const vg_fin_desc* f = d->fin;
    if (!f) return VG_OK;
    if (!d->out_sums || !f->ticket || f->njobs < 1 || f->njobs > 2 || !(f->count > 0.f)) return VG_EINVAL;
    for (int j = 0; j < f->njobs; ++j) {
        const vg_fin_job& q = f->job[j];
        if (!q.scale || !q.shift || q.c_off < 0 || q.c_off + d->Cout > q.c_tot) return VG_EINVAL;
    }
    return VG_OK;
}

extern "C" int vg_conv3d(const vg_conv_desc* d, vg_stream_t stream) {
    vg_begin();
    if (!d) return VG_EINVAL;
    if (fin_check(d) != VG_OK) return VG_EINVAL;
    vg_fin_done = false;
    const vg_actnorm_bwd_desc* b = d->bstat;
    if (b) {
        // the statistics are those of THIS launch's complete output: same tensor, same grid, plain bf16 stores
        const int pd = b->g_padded ? 2 : 0;
        if (b->g != d->out || !b->norm || !b->red || !b->x || !b->mean || !b->rstd || b->f32 || d->f32 || d->out_f32 || d->accumulate
            || b->N != d->N || b->C != d->Cout || b->D + pd != d->BD || b->H + pd != d->BH || b->W + pd != d->BW)
            return VG_EINVAL;
    }
    bool did_stats = false;
    int rc = conv3d_impl(d, stream, did_stats);
    if (rc == VG_OK && d->fin && !vg_fin_done && !vg_dry_on()) rc = vg_launch_fin(d, (hipStream_t)stream);
    if (rc != VG_OK || !b || vg_dry_on()) return rc;
    if (did_stats) return rc;                       // the epilogue left the striped sums in b->red; the apply pass adds them up
    return vg_actnorm_bwd_stats(b, stream);
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 DHWIO [T][Cin][Cout] -> bf16 [rows_pad][Ktot], k = chunk*kc_pad + tap*CK + ch
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cin, int Cout, const int* __restrict__ tap_idx,
                                    int ntaps, int transpose, int CK, int kc_pad, int Ktot, int rows_pad,
                                    void* __restrict__ out, int out_f32) {
    const size_t total = (size_t)rows_pad * Ktot;
    const int NR = transpose ? Cin : Cout;     // logical rows
    const int C = transpose ? Cout : Cin;      // contraction channels
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / Ktot), k = (int)(i % Ktot);
        const int chunk = k / kc_pad, kl = k % kc_pad;
        const int tap = kl / CK, ch = chunk * CK + kl % CK;
        float v = 0.f;
        if (row < NR && tap < ntaps && ch < C) {
            const int ts = tap_idx[tap];
            v = transpose ? w[((size_t)ts * Cin + row) * Cout + ch] : w[((size_t)ts * Cin + ch) * Cout + row];
        }
        if (out_f32) ((float*)out)[i] = v; else ((bf16_t*)out)[i] = f2bf(v);
    }
}

// One launch repacks every operand of a network.  The host gives item i the blocks [blk0, blk0 + nblk) of a 1-D grid in
// proportion to its size (the first version gave every item 128 blocks: the two 8.4 M-element operands of D.down2 then ran
// on half the chip long after the ~90 small items had finished: 0.9 ms per step for 0.5 GB of traffic).  Forward
// operands ([Cout rows][K]) transpose the DHWIO kernel, so they go through a 64x64 LDS tile: reads run along Cout
// (contiguous in w), writes along K (contiguous in the packed row); data-gradient operands ([Cin rows][K = (tap, co)])
// are contiguous on both sides already.
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const vg_pack_item* __restrict__ items, int n) {
    __shared__ float tile[64][65];
    __shared__ int sel;
    const int tid = threadIdx.x;
    for (int t = tid; t < n; t += 256) {
        const int b0 = items[t].blk0;
        if ((int)blockIdx.x >= b0 && (int)blockIdx.x < b0 + items[t].nblk) sel = t;
    }
    __syncthreads();
    const vg_pack_item it = items[sel];
    const int bx = blockIdx.x - it.blk0, gx = it.nblk;
    if (it.bn > 0) { vg_pack_dma_units(it.w, it.tap_idx, (bf16_t*)it.out, it.Cin, it.Cout, it.ntaps, it.transpose, it.bn, bx * 256 + tid, gx * 256); return; }
    const int C = it.transpose ? it.Cout : it.Cin, NR = it.transpose ? it.Cin : it.Cout;
    const int nchunks = (C + it.CK - 1) / it.CK;
    const int kc_pad = ((it.ntaps * it.CK + 31) / 32) * 32;
    const int Ktot = nchunks * kc_pad;
    const int rows_pad = ((NR + 63) / 64) * 64;
    if (it.transpose) {
        // data-gradient operand: a packed row is contiguous in w along the contraction channel, so a thread takes 8 consecutive k (one
        // (chunk, tap): CK is a multiple of 16, kc_pad of 32): one index decode, two 16-byte loads, one 16-byte store per 8 elements
        // (the per-element form ran at 1.4 TB/s on integer divisions: 0.39 ms per step).  A block takes whole rows while there are
        // enough of them, else (few long rows) a slice of every row.
        const int kspl = rows_pad >= gx ? 1 : (gx + rows_pad - 1) / rows_pad;
        const int rb = bx / kspl, ks = bx - rb * kspl, nrb = (gx + kspl - 1) / kspl;
        const bool vec = !it.out_f32 && (C % 8) == 0 && (((uintptr_t)it.w & 15) == 0) && (((uintptr_t)it.out & 15) == 0);
        if (vec) {
            const int K8 = Ktot >> 3;
            for (int row = rb; row < rows_pad; row += nrb)
                for (int k8 = ks * 256 + tid; k8 < K8; k8 += 256 * kspl) {
                    const int k = k8 << 3;
                    const int chunk = k / kc_pad, kl = k - chunk * kc_pad;
                    const int tap = kl / it.CK, ch = chunk * it.CK + (kl - tap * it.CK);
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (row < NR && tap < it.ntaps && ch < C) load8<float>(it.w + ((size_t)it.tap_idx[tap] * it.Cin + row) * it.Cout + ch, v);
                    store8<bf16_t>((bf16_t*)it.out + (size_t)row * Ktot + k, v);
                }
            return;
        }
        for (int row = rb; row < rows_pad; row += nrb)
            for (int k = ks * 256 + tid; k < Ktot; k += 256 * kspl) {
                const int chunk = k / kc_pad, kl = k - chunk * kc_pad;
                const int tap = kl / it.CK, ch = chunk * it.CK + (kl - tap * it.CK);
                float v = 0.f;
                if (row < NR && tap < it.ntaps && ch < C) v = it.w[((size_t)it.tap_idx[tap] * it.Cin + row) * it.Cout + ch];
                const size_t o = (size_t)row * Ktot + k;
                if (it.out_f32) ((float*)it.out)[o] = v; else ((bf16_t*)it.out)[o] = f2bf(v);
            }
        return;
    }
    // forward operand: w is contiguous along the packed ROW (output channel), the packed tensor along k -- 64 x 64 tiles through LDS.
    // Load phase: thread (kk = tid / 4, quarter = tid % 4) decodes ONE k and fetches 16 rows of it as four 16-byte loads; store phase:
    // thread (row = tid / 4, quarter) converts 16 consecutive k of its row and writes two 16-byte stores.
    const int tk = (Ktot + 63) >> 6, tr = rows_pad >> 6;
    const bool vecf = !it.out_f32 && (NR % 4) == 0 && (((uintptr_t)it.w & 15) == 0) && (((uintptr_t)it.out & 15) == 0) && (Ktot % 8) == 0;
    for (int t = bx; t < tk * tr; t += gx) {
        const int r0 = (t / tk) << 6, k0 = (t % tk) << 6;
        if (vecf) {
            {
                const int kk = tid >> 2, rq = (tid & 3) << 4, k = k0 + kk;
                const float* src = nullptr;
                if (k < Ktot) {
                    const int chunk = k / kc_pad, kl = k - chunk * kc_pad;
                    const int tap = kl / it.CK, ch = chunk * it.CK + (kl - tap * it.CK);
                    if (tap < it.ntaps && ch < C) src = it.w + ((size_t)it.tap_idx[tap] * it.Cin + ch) * it.Cout + r0 + rq;
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (src && r0 + rq + 4 * q4 < NR) v = *(const __attribute__((address_space(1))) f32x4*)(uintptr_t)(src + 4 * q4);      // NR % 4 == 0: all four or none
#pragma unroll
                    for (int e = 0; e < 4; ++e) tile[kk][rq + 4 * q4 + e] = v[e];
                }
            }
            __syncthreads();
            {
                const int rr = tid >> 2, kq = (tid & 3) << 4;
                bf16_t* dst = (bf16_t*)it.out + (size_t)(r0 + rr) * Ktot + k0 + kq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (k0 + kq + 8 * h < Ktot) {                                  // Ktot % 8 == 0: all eight or none
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = tile[kq + 8 * h + e][rr];
                        store8<bf16_t>(dst + 8 * h, v);
                    }
                }
            }
            __syncthreads();
            continue;
        }
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int kk = j * 4 + (tid >> 6), rr = tid & 63;
            const int k = k0 + kk, row = r0 + rr;
            float v = 0.f;
            if (k < Ktot) {
                const int chunk = k / kc_pad, kl = k - chunk * kc_pad;
                const int tap = kl / it.CK, ch = chunk * it.CK + (kl - tap * it.CK);
                if (row < NR && tap < it.ntaps && ch < C) v = it.w[((size_t)it.tap_idx[tap] * it.Cin + ch) * it.Cout + row];
            }
            tile[kk][rr] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int rr = j * 4 + (tid >> 6), kk = tid & 63;
            if (k0 + kk < Ktot) {
                const size_t o = (size_t)(r0 + rr) * Ktot + k0 + kk;
                if (it.out_f32) ((float*)it.out)[o] = tile[kk][rr]; else ((bf16_t*)it.out)[o] = f2bf(tile[kk][rr]);
            }
        }
        __syncthreads();
    }
}
extern "C" int vg_pack_weights_multi(const vg_pack_item* items_dev, int n, int total_blocks, vg_stream_t stream) {
    vg_begin();
    if (!items_dev || n < 1 || total_blocks < n) return VG_EINVAL;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n);
    return vg_check_launch();
}

extern "C" int vg_packed_ktot(int ntaps, int C, int CK) {
    vg_begin();
    if (ntaps < 1 || C < 1 || CK < 16 || (CK % 16)) return VG_EINVAL;
    const int nchunks = (C + CK - 1) / CK;
    return nchunks * (((ntaps * CK + 31) / 32) * 32);
}
extern "C" int vg_packed_rows(int N) {
    vg_begin(); return ((N + 63) / 64) * 64; }

extern "C" int vg_pack_weights(const float* w, int T, int Cin, int Cout, const int32_t* tap_idx_dev, int ntaps,
                               int transpose, int CK, void* out, int out_f32, vg_stream_t stream) {
    vg_begin();
    if (!w || !tap_idx_dev || !out || ntaps < 1 || ntaps > T) return VG_EINVAL;
    const int C = transpose ? Cout : Cin, NR = transpose ? Cin : Cout;
    const int Ktot = vg_packed_ktot(ntaps, C, CK);
    if (Ktot < 0) return Ktot;
    const int kc_pad = ((ntaps * CK + 31) / 32) * 32;
    const int rows_pad = vg_packed_rows(NR);
    const size_t total = (size_t)rows_pad * Ktot;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout,
                       tap_idx_dev, ntaps, transpose, CK, kc_pad, Ktot, rows_pad, out, out_f32);
    int rc = vg_check_launch();
    return rc == VG_OK ? Ktot : rc;
}
