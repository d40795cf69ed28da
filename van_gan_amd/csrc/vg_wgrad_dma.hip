// vg_wgrad_dma.hip -- weight gradient of the gather-convolution from a MATERIALISED operand, staged by LDS-DMA (gfx950).
//
// dW[tap][ci][co] += sum_{n,o} P[n, o*istr + tap, ci] * dY[n,o,co]      (tf.GradientTape d/dW of Conv3D, vangan.py:426-438)
//
// vg_wgrad.hip transforms the operand P = noise + mask * act(IN(x)) on the fly, once per (tap group x output block) column,
// with ~70 vector instructions per 16 bytes: its weight gradients are bound by vector-instruction issue (21 VALU per MFMA on
// the discriminators' 4x4x4 layers, profiles/r02_pmc_down0_wgrad.txt).  Here P is written ONCE per layer by an elementwise
// kernel (materialize_kernel: reflection / zero padding, virtual upsample + concat, InstanceNorm apply, activation, dropout
// mask, noise all resolved there) in the layout this kernel wants:
//     P[n][ci / 16][Dp][Hp][Wp'][16 channels]     bf16, padded grid, 16-channel planes; for stride-2 layers the W axis is
//                                                 stored de-interleaved (even positions, then odd positions)
// so that a workgroup's halo box of one plane is a set of long contiguous runs and the staging is nothing but
// global_load_lds_dwordx4 (no VGPR, no transform, ~3 instructions per KiB), double-buffered against the MFMA loop.
//
// Workgroup = 512 threads (8 waves, 2 per SIMD), persistent over a strided set of output tiles.  It owns the slab
// dW[all taps][PL planes of 16 ci][CO = 16*Q co]: wave w holds rows r = w, w+8, ... (row = (tap, plane)) x Q column blocks in
// accumulators, so a halo tile is staged ONCE for all taps (the 4x4x4 layers re-staged it per tap group: 3.8-9.6x the
// algorithmic HBM bytes, profiles/r02_roofline_by_kernel.txt).  GEMM view M = ci, N = co, K = voxels: K is the slow axis of
// both operands in memory, the fragments come from ds_read_b64_tr_b16 (as in vg_wgrad.hip).  LDS images:
//     A (halo): [plane][hd][hh][hw'][16 ch]  32 B per voxel; the 8 voxels a half-wave reads are consecutive along W
//               (256 contiguous bytes = all 64 banks once; stride 2: de-interleaved W keeps them consecutive)
//     B (dY)  : [voxel][CO ch] with the 32-byte blocks of a voxel XOR-swizzled by voxel bits (conflict-free transposed
//               reads for CO = 32 / 64); the swizzle is applied on the DMA's per-lane SOURCE address, LDS stays linear.
// K order inside a 32-voxel step: lane group lg reads voxels 4*lg..4*lg+3 and 16+4*lg..: any bijection is valid for a
// contraction as long as both operands use it, and this one makes each half-wave read 8 consecutive voxels.
#include "vg_dma_common.h"

__global__ __launch_bounds__(256) void materialize_kernel(const MatK p) {
    const int tid = threadIdx.x;
    int row = blockIdx.x;
    const int hp = row % p.Hp; row /= p.Hp;
    const int dp = row % p.Dp; const int n = row / p.Dp;
    int pd = dp + p.pmin_d, ph = hp + p.pmin_h;
    const int qd = pd + p.npad, qh = ph + p.npad;
    const bool vd = resolve_pos(pd, p.D, p.pad_mode), vh = resolve_pos(ph, p.H, p.pad_mode);
    const int npl = p.Cin >> 4;
    const int units = npl * p.Wps * 2;
    const int sh = p.shift0;
    const float slope = p.act == VG_ACT_RELU ? 0.f : (p.act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    const int ND = p.D + 2 * p.npad, NH = p.H + 2 * p.npad, NW = p.W + 2 * p.npad;
    const bool nrow = p.noise && vd && vh && qd >= 0 && qd < ND && qh >= 0 && qh < NH;
    const size_t s0row = (((size_t)n * (p.D >> sh) + (pd >> sh)) * (p.H >> sh) + (ph >> sh)) * (p.W >> sh);
    const size_t s1row = (((size_t)n * p.D + pd) * p.H + ph) * p.W;
    const size_t nzrow = (((size_t)n * ND + (nrow ? qd : 0)) * NH + (nrow ? qh : 0)) * NW;
    for (int u = tid; u < units; u += 256) {
        const int plane = u / (p.Wps * 2), r = u - plane * p.Wps * 2;
        const int ws = r >> 1, half = r & 1;
        const int j = p.deint ? (ws < p.WE ? 2 * ws : 2 * (ws - p.WE) + 1) : ws;
        int pw = j + p.pmin_w;
        const int qw = pw + p.npad;
        const bool vw = j < p.Wp && resolve_pos(pw, p.W, p.pad_mode);
        const bool valid = vd && vh && vw;
        const int c = plane * 16 + half * 8;
        float x[8];
        {
            const bool from0 = c < p.c0;
            const bf16_t* src = from0 ? (const bf16_t*)p.src0 + (s0row + (valid ? (pw >> sh) : 0)) * p.c0 + c
                                      : (const bf16_t*)p.src1 + (s1row + (valid ? pw : 0)) * p.c1 + (c - p.c0);
            if (!valid) src = (const bf16_t*)p.src0;
            Raw8<bf16_t> raw; raw_load(raw, src);
            raw_unpack(raw, x);
        }
        if (p.in_scale) {
            const f32x4* sc = (const f32x4*)(p.in_scale + n * p.Cin + c);
            const f32x4* sf = (const f32x4*)(p.in_shift + n * p.Cin + c);
            const f32x4 s0 = sc[0], s1 = sc[1], f0 = sf[0], f1 = sf[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = x[e] * s0[e] + f0[e]; x[4 + e] = x[4 + e] * s1[e] + f1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], x[e] * slope);
        if (p.noise) {
            const bool nok = nrow && valid && qw >= 0 && qw < NW;
            Raw8<bf16_t> nz; raw_load(nz, p.noise + (nzrow + (nok ? qw : 0)) * p.Cin + c);
            float z[8]; raw_unpack(nz, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += nok ? z[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = valid ? x[e] : 0.f;
        bf16_t* dst = p.out + ((((size_t)(n * npl + plane) * p.Dp + dp) * p.Hp + hp) * p.Wps + ws) * 16 + half * 8;
        store8<bf16_t>(dst, x);
    }
}

void vg_launch_materialize(const MatK& m, hipStream_t s) { hipLaunchKernelGGL(materialize_kernel, dim3(m.N * m.Dp * m.Hp), dim3(256), 0, s, m); }

// ------------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------------
#define VG_WD_MAXA 9          // A (halo) DMA pieces per wave and tile (8 waves x 9 KiB = 72 KiB of halo planes per buffer)
#define VG_WD_MAXB 4          // B (dY) pieces per wave and tile (32 KiB)
#define VG_WD_TAB 64          // DIRECT: entries of one per-tile source offset table (HD + HH + HW)

struct WgdK {
    const char* P; const char* dy;
    int N, OD, OH, OW, Cout, Cin;
    int plane_bytes;                  // one P plane: Dp*Hp*Wps*32
    int Hp, Wps, WEP;                 // P row geometry; WEP: voxel offset of the odd half of a row (de-interleaved), else 0
    int istr, deint;
    int ntaps; int tap_src[VG_MAX_TAPS]; int8_t td[VG_MAX_TAPS], th[VG_MAX_TAPS], tw[VG_MAX_TAPS];     // 0-based tap offsets
    int PL, NPL, ncob, CO2;           // planes per workgroup, planes of P, output blocks, bytes of one dY row in LDS (CO * 2)
    int tdl, thl, twl, tiles_d, tiles_h, tiles_w, total_tiles;
    int HD, HH, HW, HWE;              // halo box (voxels); HWE: even part of a de-interleaved halo row
    int imgp, nA, nB, bufb, nbuf;     // 1-KiB pieces per plane image, A pieces, B pieces per tile; bytes per buffer; buffers (2..5)
    float* dw; float* db; float* part; int dw_elems;
    unsigned m_imgp, m_hhw, m_hw, m_pl, m_tpn, m_tw, m_th, m_ncob;     // floor(2^32 / d) + 1 of the run-time divisors (fast_div)
    int co2l;                         // log2(CO2)
    // DIRECT (no materialised operand): the stored sources ([voxel][C] bf16; x0 optionally at half resolution = virtual nearest
    // upsample, x1 the concat partner), the input grid, the input position of padded index 0, the on-read transform
    const char* x0; const char* x1; int c0, c1, sh0;
    int D, H, W, pmin_d, pmin_h, pmin_w;
    const float* in_scale; const float* in_shift; float slope;
    unsigned long long* stamps;       // diagnostic (vg_set_stamp_buffer): per workgroup 8 words of phase cycle sums, else NULL
};

__device__ __forceinline__ bf16x8 tr_frag_d(const char* base0, const char* base1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_d*)base0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_d*)base1);
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// DIRECT: the halo planes are copied from the STORED tensors (reflection padding, virtual upsample + concat and the stride-2
// de-interleave all live in the per-lane source address, separable per axis: three small per-tile tables) and transformed in place
// in LDS (InstanceNorm affine + activation, each lane on the 16 bytes it copied itself -- between its own vmcnt wait and the tile's
// barrier, so no extra synchronisation).  For the thin full-resolution layers (Cin <= 48) the operand pass was as large as the weight
// gradient itself: 38 us to rewrite 67 MB for a 67 us kernel.
template <int R, int Q, bool DIRECT>
__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(const WgdK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, li = lane & 15;
    const int cib = fast_div(blockIdx.y, p.m_ncob), cob = blockIdx.y - cib * p.ncob;
    const int BM = 1 << (p.tdl + p.thl + p.twl);
    const int nks = BM >> 5;
    const int TWm = (1 << p.twl) - 1, THm = (1 << p.thl) - 1;
    int* tapoff = (int*)(smem + p.nbuf * p.bufb);
    const int CO = p.CO2 >> 1;
    const int nrows = p.ntaps * p.PL;
    const bool stamp = p.stamps != nullptr;
    unsigned long long t_begin = 0, t_loop = 0, t_a = 0, s_wait = 0, s_issue = 0, s_k = 0;
    if (stamp) t_begin = __builtin_readcyclecounter();

    // ---- per-lane DMA source offsets (the halo box has the same shape for every tile: P is padded, tiles divide the grid) ----
    int aoffs[VG_WD_MAXA], boffs[VG_WD_MAXB];
    int acst[DIRECT ? VG_WD_MAXA : 1];
    {
        const int nvox = p.HD * p.HH * p.HW;
#pragma unroll
        for (int k = 0; k < VG_WD_MAXA; ++k) {
            const int piece = wave + 8 * k;
            const int pl = fast_div(piece, p.m_imgp), pk = piece - pl * p.imgp;
            int iv = pk * 32 + (lane >> 1);
            if (iv >= nvox) iv = 0;
            const int hd = fast_div(iv, p.m_hhw), rem = iv - hd * (p.HH * p.HW);
            const int hh = fast_div(rem, p.m_hw), ws = rem - hh * p.HW;
            if constexpr (DIRECT) {
                // packed table indices (D, H, W axis entry of this lane's voxel) + source select; acst: channel bytes inside the source row
                const int j = p.deint ? (ws < p.HWE ? 2 * ws : 2 * (ws - p.HWE) + 1) : ws;
                const int c = (cib * p.PL + pl) * 16 + (lane & 1) * 8;
                const int src1 = c >= p.c0 ? 1 : 0;
                aoffs[k] = hd | ((p.HD + hh) << 8) | ((p.HD + p.HH + j) << 16) | (src1 << 24);
                acst[k] = (src1 ? c - p.c0 : c) * 2;
            } else {
                const int gw = p.deint ? (ws < p.HWE ? ws : p.WEP + ws - p.HWE) : ws;
                aoffs[k] = pl * p.plane_bytes + ((hd * p.Hp + hh) * p.Wps + gw) * 32 + (lane & 1) * 16;
            }
        }
#pragma unroll
        for (int k = 0; k < VG_WD_MAXB; ++k) {
            const int byte = (wave + 8 * k) * 1024 + lane * 16;
            int m = byte >> p.co2l; const int s = (byte - (m << p.co2l)) >> 4;
            if (m >= BM) m = 0;
            const int f = p.CO2 == 128 ? (m >> 1) & 3 : (p.CO2 == 64 ? (m >> 2) & 1 : 0);
            const int blk = (s >> 1) ^ f;
            const int w = m & TWm, h = (m >> p.twl) & THm, d = m >> (p.twl + p.thl);
            boffs[k] = (((d * p.OH + h) * p.OW + w) * p.Cout + cob * CO + blk * 16 + (s & 1) * 8) * 2;
        }
    }
    const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_d*)smem;
    // DIRECT: per-tile source offset tables, [2 buffers][2 sources][VG_WD_TAB] bytes-from-sample-base per axis entry (D entries, then
    // H, then W); built by the first 2 * (HD + HH + HW) threads for the tile about to be requested, published by the tile barrier
    int* otab = tapoff + VG_MAX_TAPS;
    float* scs = (float*)(otab + 4 * VG_WD_TAB);                 // [N][2][PL * 16]: scale, shift of this workgroup's channels
    auto build_tab = [&](int tile, int tb) {
        const int L = p.HD + p.HH + p.HW;
        if (tid >= 2 * L) return;
        int t = tile;
        const int n = fast_div(t, p.m_tpn); t -= n * tiles_per_n;
        const int t1 = fast_div(t, p.m_tw), ti_w = t - t1 * p.tiles_w;
        const int ti_d = fast_div(t1, p.m_th), ti_h = t1 - ti_d * p.tiles_h;
        const int src1 = tid >= L ? 1 : 0, e = tid - src1 * L;
        const int sh = src1 ? 0 : p.sh0, C2 = (src1 ? p.c1 : p.c0) * 2;
        const int Hs = p.H >> sh, Ws = p.W >> sh;
        int pos, nn, mul;
        if (e < p.HD) { pos = (ti_d << p.tdl) * p.istr + p.pmin_d + e; nn = p.D; mul = Hs * Ws * C2; }
        else if (e < p.HD + p.HH) { pos = (ti_h << p.thl) * p.istr + p.pmin_h + e - p.HD; nn = p.H; mul = Ws * C2; }
        else { pos = (ti_w << p.twl) * p.istr + p.pmin_w + e - p.HD - p.HH; nn = p.W; mul = C2; }
        pos = pos < 0 ? -pos : pos; pos = pos >= nn ? 2 * nn - 2 - pos : pos;              // reflection (the host admits single reflections only)
        pos = min(max(pos, 0), nn - 1);
        otab[(tb * 2 + src1) * VG_WD_TAB + e] = (pos >> sh) * mul;
    };
    auto issue = [&](int tile, int bufoff, int tb) {
        int t = tile;
        const int n = fast_div(t, p.m_tpn); t -= n * tiles_per_n;
        const int t1 = fast_div(t, p.m_tw), ti_w = t - t1 * p.tiles_w;
        const int ti_d = fast_div(t1, p.m_th), ti_h = t1 - ti_d * p.tiles_h;
        const int od0 = ti_d << p.tdl, oh0 = ti_h << p.thl, ow0 = ti_w << p.twl;
        const char* bbase = p.dy + ((((size_t)n * p.OD + od0) * p.OH + oh0) * p.OW + ow0) * p.Cout * 2;
        if constexpr (DIRECT) {
            const char* a0 = p.x0 + (size_t)n * (p.D >> p.sh0) * (p.H >> p.sh0) * (p.W >> p.sh0) * p.c0 * 2;
            const char* a1 = p.x1 + (size_t)n * p.D * p.H * p.W * p.c1 * 2;
            const int* tb0 = otab + tb * 2 * VG_WD_TAB;
#pragma unroll
            for (int k = 0; k < VG_WD_MAXA; ++k) {
                const int piece = wave + 8 * k;
                if (piece < p.nA) {
                    const int cr = aoffs[k];
                    const int* tq = tb0 + (cr >> 24) * VG_WD_TAB;
                    const int off = tq[cr & 255] + tq[(cr >> 8) & 255] + tq[(cr >> 16) & 255] + acst[k];
                    const bool lo = (fast_div(piece, p.m_imgp) + cib * p.PL) * 16 < p.c0;      // wave-uniform: a piece is one plane
                    glds16(lo ? a0 : a1, off, lds0 + bufoff + piece * 1024);
                }
            }
        } else {
            const char* abase = p.P + (size_t)(n * p.NPL + cib * p.PL) * p.plane_bytes
                                + (size_t)((od0 * p.istr * p.Hp + oh0 * p.istr) * p.Wps + (p.deint ? ow0 : ow0 * p.istr)) * 32;
#pragma unroll
            for (int k = 0; k < VG_WD_MAXA; ++k) {
                const int piece = wave + 8 * k;
                if (piece < p.nA) glds16(abase, aoffs[k], lds0 + bufoff + piece * 1024);
            }
        }
#pragma unroll
        for (int k = 0; k < VG_WD_MAXB; ++k) {
            const int piece = wave + 8 * k;
            if (piece < p.nB) glds16(bbase, boffs[k], lds0 + bufoff + (p.nA + piece) * 1024);
        }
    };
    // this workgroup's tiles: blockIdx.x, + gridDim.x, ...; nbuf - 1 of them are kept in flight ahead of the one being multiplied
    const int ntw = ((int)blockIdx.x < p.total_tiles) ? (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int n_w = (wave < p.nA ? (p.nA - wave + 7) >> 3 : 0) + (wave < p.nB ? (p.nB - wave + 7) >> 3 : 0);     // this wave's copies per tile
    if constexpr (DIRECT) {
        for (int i = tid; i < p.N * 2 * p.PL * 16; i += 512) {
            const int n = i / (2 * p.PL * 16), r = i - n * 2 * p.PL * 16, sf = r >= p.PL * 16 ? 1 : 0, c = r - sf * p.PL * 16;
            const float* src = sf ? p.in_shift : p.in_scale;
            scs[i] = src ? src[n * p.Cin + cib * p.PL * 16 + c] : (sf ? 0.f : 1.f);
        }
        for (int j = 0; j < p.nbuf - 1 && j < ntw; ++j) {
            build_tab(blockIdx.x + j * gridDim.x, j & 1);
            __syncthreads();
            issue(blockIdx.x + j * gridDim.x, j * p.bufb, j & 1);
        }
    } else {
        for (int j = 0; j < p.nbuf - 1 && j < ntw; ++j) issue(blockIdx.x + j * gridDim.x, j * p.bufb, 0);
    }

    // ---- table: tap offsets inside the halo image ----
    if (tid < p.ntaps)
        tapoff[tid] = ((p.td[tid] * p.HH + p.th[tid]) * p.HW + (p.deint ? (p.tw[tid] & 1) * p.HWE + (p.tw[tid] >> 1) : p.tw[tid])) * 32;
    __syncthreads();

    // rows of this wave: r = wave + 8*j -> (tap, plane); rows beyond the slab re-read row 0 and are dropped at the write
    int aoff[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int r = wave + 8 * j, rr = r < nrows ? r : 0;
        const int tp = fast_div(rr, p.m_pl);
        aoff[j] = (rr - tp * p.PL) * p.imgp * 1024 + tapoff[tp];
    }
    f32x4 acc[R][Q];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments: voxel row m0 = 4*lg + (li >> 2) of the K-step, swizzled 32-byte block q ^ f(m0)
    const int m0l = 4 * lg + (li >> 2);
    const int fsw = p.CO2 == 128 ? (m0l >> 1) & 3 : (p.CO2 == 64 ? (m0l >> 2) & 1 : 0);
    int yq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) yq[q] = m0l * p.CO2 + 32 * (q ^ fsw) + 8 * (li & 3);
    const int ystep = 32 * p.CO2, y16 = 16 * p.CO2;
    // A fragments: halo offset of a K-step's voxel = lane part (voxel m0l resp. m0l + 16 of the step: w, and the row inside the
    // step's 32 / TW rows) + a wave-uniform part (the step's first row and plane: scalar arithmetic, no table, no LDS read in the
    // dependent chain).  The host keeps TH >= 32 / TW, so lane row + step row never carries into the next plane.
    const int rps_l = 5 - p.twl;                                   // log2 of the rows one K-step spans
    const int wsx = p.deint ? 1 : p.istr;
    const int lx = (((m0l >> p.twl) * p.istr) * p.HW + (m0l & TWm) * wsx) * 32 + 8 * (li & 3);
    const int ly = ((((m0l + 16) >> p.twl) * p.istr) * p.HW + ((m0l + 16) & TWm) * wsx) * 32 + 8 * (li & 3);
    auto soff = [&](int ks) { const int q32 = ks << rps_l; return (((q32 >> p.thl) * p.istr * p.HH + (q32 & THm) * p.istr) * p.HW) << 5; };
    const bool do_db = p.db && cib == 0;
    float dbs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dbs[e] = 0.f;
    const int nslots = CO >> 3;                  // 16-byte slots per dY row
    const int db_s = tid % nslots, db_v0 = tid / nslots, db_vs = 512 / nslots;

    constexpr int RC = R <= 4 ? R : 4;
    constexpr int NCH = (R + RC - 1) / RC;
    int cur = 0, nxt = p.nbuf - 1;                          // buffer of the tile being multiplied / of the next tile to request
    if (stamp) t_loop = __builtin_readcyclecounter();
    for (int it = 0; it < ntw; ++it) {
        char* hb = smem + cur * p.bufb;
        if (stamp) t_a = __builtin_readcyclecounter();
        const int newer = min(p.nbuf - 2, ntw - 1 - it);         // tiles requested after this one
        const int itn = it + p.nbuf - 1;                         // the tile requested in this iteration
        if constexpr (DIRECT) { if (itn < ntw) build_tab(blockIdx.x + itn * gridDim.x, itn & 1); }
        wait_vmcnt(newer * n_w);                             // this wave's share of the tile has landed ...
        if constexpr (DIRECT) {
            // ... and is transformed where it lies: P = act(x * scale + shift), rounded to bf16 exactly as materialize_kernel does
            const int n_cur = fast_div((int)blockIdx.x + it * (int)gridDim.x, p.m_tpn);
            const float* sc0 = scs + n_cur * 2 * p.PL * 16 + (lane & 1) * 8;
#pragma unroll
            for (int k = 0; k < VG_WD_MAXA; ++k) {
                const int piece = wave + 8 * k;
                if (piece < p.nA) {
                    const int pl = fast_div(piece, p.m_imgp);
                    bf16x8* u = (bf16x8*)(hb + piece * 1024 + lane * 16);
                    const f32x4 s0 = *(const f32x4*)(sc0 + pl * 16), s1 = *(const f32x4*)(sc0 + pl * 16 + 4);
                    const f32x4 f0 = *(const f32x4*)(sc0 + (p.PL + pl) * 16), f1 = *(const f32x4*)(sc0 + (p.PL + pl) * 16 + 4);
                    const bf16x8 raw = *u;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = bf2f((bf16_t)raw[e]) * s0[e] + f0[e], b = bf2f((bf16_t)raw[4 + e]) * s1[e] + f1[e];
                        a = fmaxf(a, a * p.slope); b = fmaxf(b, b * p.slope);
                        o[e] = (short)f2bf(a); o[4 + e] = (short)f2bf(b);
                    }
                    *u = o;
                    // (Spreading the NEXT tile's transform over the K-steps of the current one -- its copies landed a tile ago with three
                    // buffers -- was measured slower, 98 vs 89 us on the 16->16 layers: LDS returns in order, so waiting for the
                    // transform's read drains the K loop's prefetched fragments.  Packed arithmetic -- v_pk_fma_f32, ReLU as v_pk_max_i16 on
                    // the rounded pair, the plane's constants loaded once per tile -- was slower as well: 95 vs 90 us.)
                }
            }
        }
        __syncthreads();                                     // ... everybody's has; and the buffer multiplied last is no longer being read
        if (stamp) { const unsigned long long t = __builtin_readcyclecounter(); s_wait += t - t_a; t_a = t; }
        if (itn < ntw) issue(blockIdx.x + itn * gridDim.x, nxt * p.bufb, itn & 1);
        if (stamp) { const unsigned long long t = __builtin_readcyclecounter(); s_issue += t - t_a; t_a = t; }
        cur = cur + 1 == p.nbuf ? 0 : cur + 1; nxt = nxt + 1 == p.nbuf ? 0 : nxt + 1;
        const char* yb = hb + p.nA * 1024;
        if (do_db) {
            for (int v = db_v0; v < BM; v += db_vs) {
                const bf16x8 r = *(const bf16x8*)(yb + v * p.CO2 + db_s * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) dbs[e] += bf2f((bf16_t)r[e]);
            }
        }
        // ---- K loop over the tile's voxels, 32 per step; rows in chunks of RC with the next chunk's (or next step's) operand
        // fragments fetched before the MFMAs of the current one.  No MFMA or fetch is conditional. ----
        {
            bf16x8 A[2][RC], B[2][Q];
            {
#pragma unroll
                for (int q = 0; q < Q; ++q) B[0][q] = tr_frag_d(yb + yq[q], yb + yq[q] + y16);
#pragma unroll
                for (int j = 0; j < RC; ++j) A[0][j] = tr_frag_d(hb + lx + aoff[j], hb + ly + aoff[j]);
            }
            for (int ks = 0; ks < nks; ks += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kn = min(ks + u + 1, nks - 1);
                    const char* hc = hb + soff(ks + u);
                    const char* hn = hb + soff(kn);
                    const char* yn = yb + kn * ystep;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int s = (u * NCH + c) & 1;
                        if (c + 1 < NCH) {
#pragma unroll
                            for (int j = 0; j < RC; ++j) {
                                const int jj = (c + 1) * RC + j < R ? (c + 1) * RC + j : R - 1;
                                A[s ^ 1][j] = tr_frag_d(hc + lx + aoff[jj], hc + ly + aoff[jj]);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < Q; ++q) B[u ^ 1][q] = tr_frag_d(yn + yq[q], yn + yq[q] + y16);
#pragma unroll
                            for (int j = 0; j < RC; ++j) A[s ^ 1][j] = tr_frag_d(hn + lx + aoff[j], hn + ly + aoff[j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < RC; ++j)
                            if (c * RC + j < R) {
#pragma unroll
                                for (int q = 0; q < Q; ++q)
                                    acc[c * RC + j][q] = VG_MFMA16(A[s][j], B[u][q], acc[c * RC + j][q]);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (stamp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s_k += __builtin_readcyclecounter() - t_a; }
    }
    unsigned long long t_slab = 0;
    if (stamp) t_slab = __builtin_readcyclecounter();
    // ---- the slab: lane holds dW rows ci = 4*lg + e, column co = li of every (row, column block) ----
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int r = wave + 8 * j;
        if (r >= nrows) continue;
        const int tap = fast_div(r, p.m_pl), ci0 = (cib * p.PL + r - tap * p.PL) * 16 + 4 * lg;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int co = cob * CO + q * 16 + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t i = ((size_t)p.tap_src[tap] * p.Cin + ci0 + e) * p.Cout + co;
                if (p.part) p.part[(size_t)blockIdx.x * p.dw_elems + i] = acc[j][q][e];
                else atomicAdd(&p.dw[i], acc[j][q][e]);
            }
        }
    }
    if (do_db) {        // block-reduce the bias partials in LDS, then one atomic per channel per workgroup
        __syncthreads();
        float* red = (float*)smem;
        if (tid < CO) red[tid] = 0.f;
        __syncthreads();
        if (db_v0 < BM) {
            const int fv = p.CO2 == 128 ? (db_v0 >> 1) & 3 : (p.CO2 == 64 ? (db_v0 >> 2) & 1 : 0);
            const int ch = 16 * ((db_s >> 1) ^ fv) + 8 * (db_s & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&red[ch + e], dbs[e]);
        }
        __syncthreads();
        if (tid < CO) atomicAdd(&p.db[cob * CO + tid], red[tid]);
    }
    if (stamp && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = t_loop - t_begin; o[1] = s_wait; o[2] = s_issue; o[3] = s_k; o[4] = __builtin_readcyclecounter() - t_slab;
        o[5] = __builtin_readcyclecounter() - t_begin; o[6] = ntw; o[7] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int R, int Q, bool DIRECT>
static void launch_wd2(const WgdK& k, dim3 grid, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_dma_kernel<R, Q, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_dma_kernel<R, Q, DIRECT>), grid, dim3(512), lds, s, k);
}
template <int R, int Q>
static void launch_wd(const WgdK& k, bool direct, dim3 grid, int lds, hipStream_t s) {
    if (direct) launch_wd2<R, Q, true>(k, grid, lds, s); else launch_wd2<R, Q, false>(k, grid, lds, s);
}

// Serve vg_conv3d_wgrad through the materialised-operand path.  Returns VG_OK when served, 1 when the call is not one of its
// shapes (the caller continues with wgrad_kernel), < 0 on error.
int vg_wgrad_dma(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host, int T_total, float* dw, float* db,
                 float* scratch, int64_t scratch_bytes, hipStream_t s) {
    if (!vg_tune("WGRAD_DMA", 1)) return 1;
    const int Cin = d->c_src0 + d->c_src1;
    if (d->f32 || dy_f32 || d->src_f32 || d->wpack || Cin < 16 || (Cin % 16) || (d->Cout % 16) || d->ntaps < 8 || d->ntaps > VG_MAX_TAPS) return 1;
    if (d->istr < 1 || d->istr > 2 || !scratch || T_total < 1) return 1;
    if ((d->OW % 8) || d->OD < 2 || d->OH < 2) return 1;
    if (d->c_src1 > 0 && ((d->c_src0 % 8) || (d->c_src1 % 8))) return 1;
    // taps: extents and 0-based offsets
    int mn[3] = {127, 127, 127}, mx[3] = {-128, -128, -128};
    for (int i = 0; i < d->ntaps; ++i) {
        const int v[3] = {d->tap_d[i], d->tap_h[i], d->tap_w[i]};
        for (int a = 0; a < 3; ++a) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
    }
    const int ex[3] = {mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1};
    if (d->pad_mode == VG_PAD_REFLECT) {         // one reflection only: the padded grid must stay within [-(n-1), 2n-2]
        const int od[3] = {d->OD, d->OH, d->OW}, nn[3] = {d->D, d->H, d->W};
        for (int a = 0; a < 3; ++a) if (mn[a] < -(nn[a] - 1) || (od[a] - 1) * d->istr + mx[a] > 2 * nn[a] - 2) return 1;
    }
    const int NPL = Cin / 16;
    const int deint = d->istr == 2 ? 1 : 0;
    const int Dp = (d->OD - 1) * d->istr + ex[0], Hp = (d->OH - 1) * d->istr + ex[1], Wp = (d->OW - 1) * d->istr + ex[2];
    const int WE = (Wp + 1) / 2, Wps = deint ? 2 * WE : Wp;
    const int64_t plane_bytes = (int64_t)Dp * Hp * Wps * 32;
    // thin reflect-padded noise-free layers: no operand pass, the kernel copies the stored tensors (wgrad_dma_kernel<.., DIRECT>)
    const bool direct = vg_tune("WGRAD_DMA_DIRECT", 1) && d->pad_mode == VG_PAD_REFLECT && !d->noise && Cin <= vg_tune("WGRAD_DMA_DIRECT_CIN", 48)
                        && (d->c_src1 == 0 || (d->c_src0 % 16) == 0) && (!d->src0_shift || (!(d->D & 1) && !(d->H & 1) && !(d->W & 1)))
                        && (int64_t)d->D * d->H * d->W * (d->c_src0 > d->c_src1 ? d->c_src0 : d->c_src1) * 2 < (1LL << 31);
    const int64_t p_bytes = direct ? 0 : ((plane_bytes * NPL * d->N + 255) / 256) * 256;
    if (p_bytes >= (1LL << 31)) return 1;
    const int64_t dw_elems = (int64_t)T_total * Cin * d->Cout;
    // ---- plan: column block CO (16 * Q channels), planes per workgroup PL, tile, K split bx.  Every candidate (CO, PL) with an
    // instantiated (rows per wave, Q) variant is priced with a small model of one workgroup's life (cycles): tile loop =
    // max(MFMA issue, staging at the per-CU LDS-DMA rate) per tile + one DMA latency, the slab write, and -- for bx > 1 -- the
    // partial slabs' round trip through HBM plus the reduce launch.  Small deep layers (8^3, 16^3: few voxels, megabytes of dW)
    // come out with narrow columns and no K split (slab added to dW with atomics once), the big thin layers with wide slabs
    // and 256 K slices.  WGRAD_DMA_CO / _PL / _BX / _BM override (sweeps: tools/sweep_wgrad.py).
    struct Plan { int CO, PL, R, TD, TH, TW, lds, columns, tiles, bx, nbuf; double cost; };
    Plan best; best.cost = -1;
    const int f_co = vg_tune("WGRAD_DMA_CO", 0), f_pl = vg_tune("WGRAD_DMA_PL", 0), f_bx = vg_tune("WGRAD_DMA_BX", 0);
    const int bm_cap = vg_tune("WGRAD_DMA_BM", 512), nbuf_cap = vg_tune("WGRAD_DMA_NBUF", 5), wg_target = vg_tune("WGRAD_DMA_WGS", 256);
    static const int RQ4[] = {8, 0}, RQ2[] = {4, 8, 0}, RQ1[] = {4, 8, 12, 0};       // the instantiated (rows per wave, Q) variants
    for (int CO = 64; CO >= 16; CO >>= 1) {
        if (CO > d->Cout || (d->Cout % CO) || (f_co && CO != f_co)) continue;
        const int Q = CO / 16;
        const int* ravail = Q == 4 ? RQ4 : (Q == 2 ? RQ2 : RQ1);
        for (int PL = NPL; PL >= 1; --PL) {
            if ((NPL % PL) || (f_pl && PL != f_pl) || plane_bytes * PL >= (1LL << 30)) continue;
            const int rneed = (d->ntaps * PL + 7) / 8;
            int R = 0; for (int i = 0; ravail[i]; ++i) if (ravail[i] >= rneed) { R = ravail[i]; break; }
            if (!R) continue;
            const int columns = (NPL / PL) * (d->Cout / CO);
            // thin slabs (<= 8 accumulator fragments per wave, <= 128 VGPRs): two workgroups per CU -- their MFMA phases are too
            // short to cover an LDS round trip with two waves per SIMD
            const int per_cu = 1;          // (two workgroups per CU for the thin slabs: measured slower, 132 vs 115 us on the 16->16 layers at 128^3)
            const int lds_cap = VG_LDS_LIMIT / per_cu;
            for (int bm = 512; bm >= 64; bm >>= 1) {
                if (bm > bm_cap) continue;
                // tile: TW in {8, 16}, powers of two dividing the grid, two buffers within the LDS, least halo volume
                long bvol = -1; int t3[5] = {0, 0, 0, 0, 0}, bufb = 0;
                for (int tw = 8; tw <= 16; tw <<= 1) {
                    if (d->OW % tw) continue;
                    for (int th = 1; th <= d->OH && tw * th <= bm; th <<= 1) {
                        if ((d->OH % th) || tw * th < 32) continue;       // a K-step's 32 voxels stay inside one plane of the tile
                        const int td = bm / (tw * th);
                        if (td > d->OD || (d->OD % td)) continue;
                        const int HD = (td - 1) * d->istr + ex[0], HH = (th - 1) * d->istr + ex[1], HW = (tw - 1) * d->istr + ex[2];
                        const int imgp = (HD * HH * HW * 32 + 1023) / 1024;
                        const int nA = PL * imgp, nB = bm * CO * 2 / 1024;
                        if (nA > 8 * VG_WD_MAXA || nB > 8 * VG_WD_MAXB || nB < 1) continue;
                        if (direct && (HD + HH + HW > VG_WD_TAB || d->N * PL > 16)) continue;
                        const int tabs = VG_MAX_TAPS * 4 + (direct ? 4 * VG_WD_TAB * 4 + d->N * 2 * PL * 16 * 4 : 0);
                        if (2 * (nA + nB) * 1024 + tabs > lds_cap) continue;
                        int nb_ = (lds_cap - tabs) / ((nA + nB) * 1024); if (nb_ > nbuf_cap) nb_ = nbuf_cap; if (nb_ < 2) nb_ = 2;
                        const int lds = nb_ * (nA + nB) * 1024 + tabs;
                        const long vol = (long)HD * HH * HW;
                        if (bvol < 0 || vol < bvol) { bvol = vol; t3[0] = td; t3[1] = th; t3[2] = tw; t3[3] = lds; t3[4] = nb_; bufb = (nA + nB) * 1024; }
                    }
                }
                if (bvol < 0) continue;
                const int tiles = d->N * (d->OD / t3[0]) * (d->OH / t3[1]) * (d->OW / t3[2]);
                // fast_div(n, m) is exact while n * d < 2^32: the largest division of the kernel is tile index / tiles per sample
                if ((int64_t)tiles * (tiles / d->N) >= (1LL << 32)) continue;
                static const int BXC[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 0};
                for (int ib = 0; ib < 17; ++ib) {
                    int bxx = BXC[ib] ? BXC[ib] : wg_target * per_cu / columns;        // last candidate: exactly the target grid
                    if (bxx < 1) bxx = 1;
                    if (bxx > tiles) bxx = tiles;
                    if (f_bx) bxx = f_bx < tiles ? f_bx : tiles;
                    else if ((long)bxx * columns > (long)wg_target * per_cu && bxx > 1) continue;
                    const int64_t part_bytes = bxx > 1 ? (int64_t)bxx * dw_elems * 4 : 0;
                    if (p_bytes + part_bytes > scratch_bytes) { if (f_bx) break; continue; }
                    const double ntile = (double)((tiles + bxx - 1) / bxx);
                    const double rounds = (double)(((long)bxx * columns + 256 * per_cu - 1) / (256 * per_cu));          // workgroups beyond the CUs queue up
                    const double mfma = (bm / 32) * R * Q * 16.0 * 2.0 * per_cu, stage = bufb / 14.0 * per_cu;
                    // a copy takes ~4500 cycles to land: with nbuf - 1 tiles in flight a tile costs at least latency / (nbuf - 1)
                    const double lat = 4500.0 / (t3[4] - 1);
                    double per = mfma > stage ? mfma : stage; if (lat > per) per = lat;
                    const double loop = ntile * per + 5000.0;
                    const double slab = (double)dw_elems / columns * 4.0 / 16.0;                  // one workgroup's slab at ~16 B/clk
                    double tail;
                    if (bxx > 1) tail = ((double)(bxx + 1) * dw_elems * 4.0 + dw_elems * 4.0 * 3.0) / 2500.0 + 12000.0;   // partials back in, atomics out (bytes per clk, chip), reduce launch
                    else tail = (double)dw_elems * 4.0 * 3.0 / 2500.0;                                                     // atomics: ~1/3 of the store rate
                    const double cost = rounds * (loop + slab) + tail + 6000.0;
                    if (best.cost < 0 || cost < best.cost) best = Plan{CO, PL, R, t3[0], t3[1], t3[2], t3[3], columns, tiles, bxx, t3[4], cost};
                    if (f_bx) break;
                }
            }
        }
    }
    if (best.cost < 0) return 1;
    const int CO = best.CO, Q = CO / 16, PL = best.PL, Rsel = best.R;
    const int TD = best.TD, TH = best.TH, TW = best.TW, lds = best.lds, columns = best.columns, ncob = d->Cout / CO;
    const int BM = TD * TH * TW;
    WgdK k;
    k.N = d->N; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout; k.Cin = Cin;
    k.plane_bytes = (int)plane_bytes; k.Hp = Hp; k.Wps = Wps; k.WEP = deint ? WE : 0; k.istr = d->istr; k.deint = deint;
    k.ntaps = d->ntaps;
    for (int i = 0; i < VG_MAX_TAPS; ++i) {
        k.tap_src[i] = i < d->ntaps ? tap_idx_host[i] : 0;
        k.td[i] = i < d->ntaps ? d->tap_d[i] - mn[0] : 0; k.th[i] = i < d->ntaps ? d->tap_h[i] - mn[1] : 0; k.tw[i] = i < d->ntaps ? d->tap_w[i] - mn[2] : 0;
    }
    k.PL = PL; k.NPL = NPL; k.ncob = ncob; k.CO2 = CO * 2;
    k.tdl = ilog2_exact(TD); k.thl = ilog2_exact(TH); k.twl = ilog2_exact(TW);
    k.tiles_d = d->OD / TD; k.tiles_h = d->OH / TH; k.tiles_w = d->OW / TW;
    k.total_tiles = d->N * k.tiles_d * k.tiles_h * k.tiles_w;
    k.HD = (TD - 1) * d->istr + ex[0]; k.HH = (TH - 1) * d->istr + ex[1]; k.HW = (TW - 1) * d->istr + ex[2];
    k.HWE = deint ? (k.HW + 1) / 2 : 0;
    k.imgp = (k.HD * k.HH * k.HW * 32 + 1023) / 1024; k.nA = PL * k.imgp; k.nB = BM * CO * 2 / 1024; k.bufb = (k.nA + k.nB) * 1024; k.nbuf = best.nbuf;
    k.dw = dw; k.db = db; k.dw_elems = (int)dw_elems; k.stamps = g_vg_stamps;
    auto magic = [](int dd) { return dd <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)dd + 1ULL); };
    k.m_imgp = magic(k.imgp); k.m_hhw = magic(k.HH * k.HW); k.m_hw = magic(k.HW); k.m_pl = magic(PL);
    k.m_tpn = magic(k.tiles_d * k.tiles_h * k.tiles_w); k.m_tw = magic(k.tiles_w); k.m_th = magic(k.tiles_h); k.m_ncob = magic(ncob);
    k.co2l = ilog2_exact(CO * 2);
    const int bx = best.bx;
    k.P = (const char*)scratch; k.dy = (const char*)dy;
    k.part = bx > 1 ? (float*)((char*)scratch + p_bytes) : nullptr;
    k.x0 = (const char*)d->src0; k.x1 = (const char*)d->src1; k.c0 = d->c_src0; k.c1 = d->c_src1; k.sh0 = d->src0_shift ? 1 : 0;
    k.D = d->D; k.H = d->H; k.W = d->W; k.pmin_d = mn[0]; k.pmin_h = mn[1]; k.pmin_w = mn[2];
    k.in_scale = d->in_scale; k.in_shift = d->in_shift;
    k.slope = d->act == VG_ACT_RELU ? 0.f : (d->act == VG_ACT_LRELU ? VG_LRELU : 1.f);
    if (vg_dry("wgrad_dma<%d,%d,d%d>|bm%d|pl%d|s%d|nb%d|part%d|walk%d", Rsel, Q, direct ? 1 : 0, BM, PL, d->istr, k.nbuf, k.part ? 1 : 0, k.total_tiles > bx ? 1 : 0)) return VG_OK;
    const dim3 grid(bx, columns, 1);
    if (!direct) {
    MatK m;
    m.src0 = d->src0; m.src1 = d->src1; m.c0 = d->c_src0; m.c1 = d->c_src1; m.shift0 = d->src0_shift ? 1 : 0;
    m.N = d->N; m.D = d->D; m.H = d->H; m.W = d->W; m.Cin = Cin;
    m.in_scale = d->in_scale; m.in_shift = d->in_shift; m.act = d->act;
    m.noise = (const bf16_t*)d->noise; m.npad = d->noise ? d->noise_pad : 0; m.pad_mode = d->pad_mode;
    m.pmin_d = mn[0]; m.pmin_h = mn[1]; m.pmin_w = mn[2]; m.Dp = Dp; m.Hp = Hp; m.Wp = Wp;
    m.deint = deint; m.WE = WE; m.Wps = Wps; m.out = (bf16_t*)scratch;
    vg_launch_materialize(m, s);
    }
    if (Q == 4) launch_wd<8, 4>(k, direct, grid, lds, s);
    else if (Q == 2) { if (Rsel == 4) launch_wd<4, 2>(k, direct, grid, lds, s); else launch_wd<8, 2>(k, direct, grid, lds, s); }
    else { if (Rsel == 4) launch_wd<4, 1>(k, direct, grid, lds, s); else if (Rsel == 8) launch_wd<8, 1>(k, direct, grid, lds, s); else launch_wd<12, 1>(k, direct, grid, lds, s); }
    if (k.part) vg_launch_reduce_partials(k.part, bx, k.dw_elems, dw, s);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1x1 convolutions (the residual blocks' shortcuts, resunet_model.py:126-131: raw block input, stride 1 or 2, virtual
// upsample + concat in the decoder): dW[ci][co] = sum_v X[v*istr][ci] * dY[v][co].  No taps, no halo, no on-read transform: both
// tiles are copied by LDS-DMA straight from the stored tensors -- the upsample shift, the concat source and the stride live in the
// per-lane source offsets, which are the same for every tile (even tile origins).  The slab is tiny (1 x 1 ... 24 x 8 fragments of
// 16 x 16) and K = all voxels, so the eight waves split K as well as rows: wave (kgrp, rgrp) takes K-steps ks = kgrp mod KS and the
// 16-channel row blocks r = rgrp mod RS (x all Q column blocks); the KS partial slabs of a workgroup are summed through LDS at the end.
// These layers ran on the on-the-fly kernels at 4-10x their HBM time (dec1.short: 69 us for 38 MB).
// ------------------------------------------------------------------------------------------------------------------
struct WpwK {
    const char* x0; const char* x1; const char* dy;
    int c0, c1, sh0, istr;
    int N, D, H, W, OD, OH, OW, Cin, Cout;
    int FA, RS, KS, ksl, ncob, CO2, co2l;
    int tdl, thl, twl, tiles_d, tiles_h, tiles_w, total_tiles;
    int ppp, nA, nB, bufb, nbuf;          // 1-KiB pieces per 16-channel plane of the X tile (BM / 32), A / B pieces, buffer bytes, buffers
    unsigned m_tpn, m_tw, m_th, m_ppp, m_ncob;
    float* dw; float* db; float* part; int dw_elems;
};

template <int R, int Q>
__global__ __launch_bounds__(512, 2) void wgrad_pw_dma_kernel(const WpwK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, li = lane & 15;
    const int rgrp0 = fast_div(blockIdx.y, p.m_ncob), cob = blockIdx.y - rgrp0 * p.ncob;      // (row-block group of the workgroup: always 0 here)
    (void)rgrp0;
    const int kgrp = wave & (p.KS - 1), rgrp = wave >> p.ksl;
    const int BM = 1 << (p.tdl + p.thl + p.twl);
    const int nks = BM >> 5;
    const int TWm = (1 << p.twl) - 1, THm = (1 << p.thl) - 1;
    const int CO = p.CO2 >> 1;
    const int Hs = p.H >> p.sh0, Ws = p.W >> p.sh0;
    int aoffs[VG_WD_MAXA], boffs[VG_WD_MAXB];
#pragma unroll
    for (int k = 0; k < VG_WD_MAXA; ++k) {
        const int piece = wave + 8 * k;
        const int plane = fast_div(piece, p.m_ppp), pk = piece - plane * p.ppp;
        const int m = pk * 32 + (lane >> 1);
        const int c = plane * 16 + (lane & 1) * 8;
        const int w = (m & TWm) * p.istr, h = ((m >> p.twl) & THm) * p.istr, d = (m >> (p.twl + p.thl)) * p.istr;
        aoffs[k] = c < p.c0 ? ((((d >> p.sh0) * Hs + (h >> p.sh0)) * Ws + (w >> p.sh0)) * p.c0 + c) * 2
                            : (((d * p.H + h) * p.W + w) * p.c1 + (c - p.c0)) * 2;
    }
#pragma unroll
    for (int k = 0; k < VG_WD_MAXB; ++k) {
        const int byte = (wave + 8 * k) * 1024 + lane * 16;
        int m = byte >> p.co2l; const int s = (byte - (m << p.co2l)) >> 4;
        if (m >= BM) m = 0;
        const int f = p.CO2 == 128 ? (m >> 1) & 3 : (p.CO2 == 64 ? (m >> 2) & 1 : 0);
        const int blk = (s >> 1) ^ f;
        const int w = m & TWm, h = (m >> p.twl) & THm, d = m >> (p.twl + p.thl);
        boffs[k] = (((d * p.OH + h) * p.OW + w) * p.Cout + cob * CO + blk * 16 + (s & 1) * 8) * 2;
    }
    const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_d*)smem;
    auto issue = [&](int tile, int bufoff) {
        int t = tile;
        const int n = fast_div(t, p.m_tpn); t -= n * tiles_per_n;
        const int t1 = fast_div(t, p.m_tw), ti_w = t - t1 * p.tiles_w;
        const int ti_d = fast_div(t1, p.m_th), ti_h = t1 - ti_d * p.tiles_h;
        const int od0 = ti_d << p.tdl, oh0 = ti_h << p.thl, ow0 = ti_w << p.twl;
        const int id0 = od0 * p.istr, ih0 = oh0 * p.istr, iw0 = ow0 * p.istr;
        const char* a0 = p.x0 + ((((size_t)n * (p.D >> p.sh0) + (id0 >> p.sh0)) * Hs + (ih0 >> p.sh0)) * Ws + (iw0 >> p.sh0)) * p.c0 * 2;
        const char* a1 = p.x1 + ((((size_t)n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.c1 * 2;
        const char* bbase = p.dy + ((((size_t)n * p.OD + od0) * p.OH + oh0) * p.OW + ow0) * p.Cout * 2;
#pragma unroll
        for (int k = 0; k < VG_WD_MAXA; ++k) {
            const int piece = wave + 8 * k;
            if (piece < p.nA) {
                const bool lo = fast_div(piece, p.m_ppp) * 16 < p.c0;          // wave-uniform: a piece is one plane
                glds16(lo ? a0 : a1, aoffs[k], lds0 + bufoff + piece * 1024);
            }
        }
#pragma unroll
        for (int k = 0; k < VG_WD_MAXB; ++k) {
            const int piece = wave + 8 * k;
            if (piece < p.nB) glds16(bbase, boffs[k], lds0 + bufoff + (p.nA + piece) * 1024);
        }
    };
    const int ntw = ((int)blockIdx.x < p.total_tiles) ? (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int n_w = (wave < p.nA ? (p.nA - wave + 7) >> 3 : 0) + (wave < p.nB ? (p.nB - wave + 7) >> 3 : 0);
    for (int j = 0; j < p.nbuf - 1 && j < ntw; ++j) issue(blockIdx.x + j * gridDim.x, j * p.bufb);

    // row blocks of this wave: plane = rgrp + RS * j (beyond FA: plane 0, dropped at the write)
    int poff[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const int pl = rgrp + p.RS * j; poff[j] = (pl < p.FA ? pl : 0) * (BM * 32); }
    f32x4 acc[R][Q];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m0l = 4 * lg + (li >> 2);
    const int fsw = p.CO2 == 128 ? (m0l >> 1) & 3 : (p.CO2 == 64 ? (m0l >> 2) & 1 : 0);
    int yq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) yq[q] = m0l * p.CO2 + 32 * (q ^ fsw) + 8 * (li & 3);
    const int lx = m0l * 32 + 8 * (li & 3);
    const int ystep = 32 * p.CO2, y16 = 16 * p.CO2;
    const bool do_db = p.db != nullptr;
    float dbs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dbs[e] = 0.f;
    const int nslots = CO >> 3;
    const int db_s = tid % nslots, db_v0 = tid / nslots, db_vs = 512 / nslots;

    int cur = 0, nxt = p.nbuf - 1;
    for (int it = 0; it < ntw; ++it) {
        const char* hb = smem + cur * p.bufb;
        const int newer = min(p.nbuf - 2, ntw - 1 - it);
        wait_vmcnt(newer * n_w);
        __syncthreads();
        if (it + p.nbuf - 1 < ntw) issue(blockIdx.x + (it + p.nbuf - 1) * gridDim.x, nxt * p.bufb);
        cur = cur + 1 == p.nbuf ? 0 : cur + 1; nxt = nxt + 1 == p.nbuf ? 0 : nxt + 1;
        const char* yb = hb + p.nA * 1024;
        if (do_db) {
            for (int v = db_v0; v < BM; v += db_vs) {
                const bf16x8 r = *(const bf16x8*)(yb + v * p.CO2 + db_s * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) dbs[e] += bf2f((bf16_t)r[e]);
            }
        }
        for (int ks = kgrp; ks < nks; ks += p.KS) {
            bf16x8 A[R], B[Q];
            const char* hk = hb + ks * 1024 + lx;
            const char* yk = yb + ks * ystep;
#pragma unroll
            for (int q = 0; q < Q; ++q) B[q] = tr_frag_d(yk + yq[q], yk + yq[q] + y16);
#pragma unroll
            for (int j = 0; j < R; ++j) A[j] = tr_frag_d(hk + poff[j], hk + poff[j] + 512);
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[j][q] = VG_MFMA16(A[j], B[q], acc[j][q]);
        }
    }
    // ---- sum the KS partial slabs of the workgroup through LDS (pairwise rounds), then the kgrp == 0 waves write ----
    float* red = (float*)smem;
    int sl = 1;                                               // log2(2 * step)
    for (int step = 1; step < p.KS; step <<= 1, ++sl) {
        // receivers of this round: kgrp a multiple of 2 * step; slot = (row group, kgrp / (2 * step)): at most 4 slots of R * Q KiB
        const int slot = rgrp * (p.KS >> sl) + (kgrp >> sl);
        __syncthreads();
        if ((kgrp & (2 * step - 1)) == step) {                  // sender -> the receiver step waves below it (same slot index)
            float* dst = red + (size_t)(slot * R * Q) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int q = 0; q < Q; ++q) *(f32x4*)(dst + (j * Q + q) * 256) = acc[j][q];
        }
        __syncthreads();
        if ((kgrp & (2 * step - 1)) == 0) {
            const float* src = red + (size_t)(slot * R * Q) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[j][q] += *(const f32x4*)(src + (j * Q + q) * 256);
        }
    }
    if (kgrp == 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int pl = rgrp + p.RS * j;
            if (pl >= p.FA) continue;
            const int ci0 = pl * 16 + 4 * lg;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int co = cob * CO + q * 16 + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t i = (size_t)(ci0 + e) * p.Cout + co;
                    if (p.part) p.part[(size_t)blockIdx.x * p.dw_elems + i] = acc[j][q][e];
                    else atomicAdd(&p.dw[i], acc[j][q][e]);
                }
            }
        }
    }
    if (do_db) {
        __syncthreads();
        if (tid < CO) red[tid] = 0.f;
        __syncthreads();
        if (db_v0 < BM) {
            const int fv = p.CO2 == 128 ? (db_v0 >> 1) & 3 : (p.CO2 == 64 ? (db_v0 >> 2) & 1 : 0);
            const int ch = 16 * ((db_s >> 1) ^ fv) + 8 * (db_s & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&red[ch + e], dbs[e]);
        }
        __syncthreads();
        if (tid < CO) atomicAdd(&p.db[cob * CO + tid], red[tid]);
    }
}

template <int R, int Q>
static void launch_wpw(const WpwK& k, dim3 grid, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_pw_dma_kernel<R, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_pw_dma_kernel<R, Q>), grid, dim3(512), lds, s, k);
}

// Serve a 1x1x1 weight gradient (raw multi-channel source).  VG_OK served, 1 not one of its shapes, < 0 error.
int vg_wgrad_pw_dma(const vg_conv_desc* d, const void* dy, int dy_f32, int T_total, float* dw, float* db, float* scratch,
                    int64_t scratch_bytes, hipStream_t s) {
    if (!vg_tune("WGRAD_PW_DMA", 1)) return 1;
    const int Cin = d->c_src0 + d->c_src1;
    if (d->f32 || dy_f32 || d->src_f32 || d->wpack || d->ntaps != 1 || T_total != 1 || d->tap_d[0] || d->tap_h[0] || d->tap_w[0]) return 1;
    if (d->in_scale || d->act != VG_ACT_NONE || d->noise || !scratch) return 1;
    if (Cin < 16 || (Cin % 16) || (d->Cout % 16) || (d->c_src0 % 16) || (d->c_src1 % 16)) return 1;
    if (d->istr < 1 || d->istr > 2 || d->OD * d->istr != d->D || d->OH * d->istr != d->H || d->OW * d->istr != d->W) return 1;
    if (d->src0_shift && d->istr != 1) return 1;
    const int FA = Cin / 16;
    const int CO = d->Cout >= 64 ? 64 : d->Cout;
    if ((d->Cout % CO) || FA > 24) return 1;
    const int Q = CO / 16, ncob = d->Cout / CO;
    // rows per wave / K split: the instantiated (R, Q); RS row groups x KS K groups = 8 waves
    int R = 0, RS = 1;
    if (Q == 1) { R = 3; } else if (Q == 2) { R = 6; } else { R = FA <= 4 ? 4 : 6; }
    while (RS < 8 && RS * R < FA) RS <<= 1;
    if (RS * R < FA) return 1;
    const int KS = 8 / RS;
    // tile: even origins (upsample shift), TW in {8, 16}, TW * TH >= 32; two to four buffers
    int best[5] = {0, 0, 0, 0, 0};
    for (int bm = 512; bm >= 64 && !best[0]; bm >>= 1) {
        const int nA = FA * (bm / 32), nB = bm * CO * 2 / 1024;
        if (nA > 8 * VG_WD_MAXA || nB > 8 * VG_WD_MAXB || nB < 1) continue;
        const int bufb = (nA + nB) * 1024;
        if (2 * bufb > VG_LDS_LIMIT || (bm / 32) < KS) continue;
        if (8 * R * Q * 1024 / 2 > VG_LDS_LIMIT) continue;         // the reduction's largest round: half the waves' slabs
        for (int tw = 16; tw >= 8 && !best[0]; tw >>= 1) {
            if (d->OW % tw) continue;
            for (int th = 2; th <= d->OH && tw * th <= bm; th <<= 1) {
                if ((d->OH % th) || tw * th < 32) continue;
                const int td = bm / (tw * th);
                if (td > d->OD || (d->OD % td) || (d->src0_shift && (td & 1))) continue;
                int nb = VG_LDS_LIMIT / bufb; if (nb > 4) nb = 4;
                best[0] = td; best[1] = th; best[2] = tw; best[3] = bufb; best[4] = nb;
                break;
            }
        }
    }
    if (!best[0]) return 1;
    const int TD = best[0], TH = best[1], TW = best[2], BM = TD * TH * TW;
    if (d->src0_shift && ((TD | TH | TW) & 1)) return 1;
    WpwK k;
    k.x0 = (const char*)d->src0; k.x1 = (const char*)d->src1; k.dy = (const char*)dy;
    k.c0 = d->c_src0; k.c1 = d->c_src1; k.sh0 = d->src0_shift ? 1 : 0; k.istr = d->istr;
    k.N = d->N; k.D = d->D; k.H = d->H; k.W = d->W; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW; k.Cin = Cin; k.Cout = d->Cout;
    k.FA = FA; k.RS = RS; k.KS = KS; k.ksl = ilog2_exact(KS); k.ncob = ncob; k.CO2 = CO * 2; k.co2l = ilog2_exact(CO * 2);
    k.tdl = ilog2_exact(TD); k.thl = ilog2_exact(TH); k.twl = ilog2_exact(TW);
    k.tiles_d = d->OD / TD; k.tiles_h = d->OH / TH; k.tiles_w = d->OW / TW;
    k.total_tiles = d->N * k.tiles_d * k.tiles_h * k.tiles_w;
    k.ppp = BM / 32; k.nA = FA * k.ppp; k.nB = BM * CO * 2 / 1024; k.bufb = best[3]; k.nbuf = best[4];
    auto magic = [](int dd) { return dd <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)dd + 1ULL); };
    k.m_tpn = magic(k.tiles_d * k.tiles_h * k.tiles_w); k.m_tw = magic(k.tiles_w); k.m_th = magic(k.tiles_h); k.m_ppp = magic(k.ppp); k.m_ncob = magic(ncob);
    k.dw = dw; k.db = db; k.dw_elems = Cin * d->Cout;
    if ((int64_t)k.total_tiles * (k.total_tiles / d->N) >= (1LL << 32)) return 1;      // beyond the exact range of fast_div (n * d < 2^32)
    if (k.total_tiles * ncob < 32) return 1;          // a handful of tiles (8^3 level): two launches cost more than the on-the-fly kernel (22 vs 18 us)
    int bx = vg_tune("WGRAD_PW_WGS", 256) / ncob; if (bx < 1) bx = 1; if (bx > k.total_tiles) bx = k.total_tiles;
    const int64_t part_bytes = (int64_t)bx * k.dw_elems * 4;
    if (part_bytes > scratch_bytes) return 1;
    k.part = (float*)scratch;
    int lds = k.nbuf * k.bufb; { const int rl = 4 * R * Q * 1024; if (rl > lds) lds = rl; }
    if (vg_dry("wgrad_pw_dma<%d,%d>|bm%d|rs%d|s%d|u%d|nb%d|walk%d", R, Q, BM, RS, d->istr, k.sh0, k.nbuf, k.total_tiles > bx ? 1 : 0)) return VG_OK;
    const dim3 grid(bx, ncob, 1);
    if (Q == 1) launch_wpw<3, 1>(k, grid, lds, s);
    else if (Q == 2) launch_wpw<6, 2>(k, grid, lds, s);
    else if (R == 4) launch_wpw<4, 4>(k, grid, lds, s);
    else launch_wpw<6, 4>(k, grid, lds, s);
    vg_launch_reduce_partials(k.part, bx, k.dw_elems, dw, s);
    return vg_check_launch();
}
