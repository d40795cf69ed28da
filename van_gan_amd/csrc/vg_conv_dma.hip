// vg_conv_dma.hip -- forward / data-gradient convolution of the WIDE layers with BOTH MFMA operands staged by LDS-DMA (gfx950).
//
//   out[n, o*ostr + ooff, co] (+)= sum_taps sum_ci P[n, o + tap, ci] * W[tap][ci][co]          (stride-1 walk over a padded operand)
//
// Replaces conv32_kernel for Conv3D layers with >= 64 input and a multiple of 64 output channels (discriminator.py:64-117 down0/1/2,
// resunet_model.py:42-66 / 103-143 at the 32^3 .. 8^3 levels) and for their data gradients (tf.GradientTape d/d input,
// vangan.py:426-438): forward of the stride-1 layers, data gradient of the stride-1 AND stride-2 layers (per output-parity class a
// stride-1 walk over dY; all classes in one class-parallel launch).  conv32_kernel staged the activations with ~70 vector
// instructions per 16 bytes at one wave per SIMD and streamed its weight fragments through L1, every wave re-reading every
// fragment (profiles/r03_pmc_down2_*.txt: 11 issue slots per MFMA, 60 % LDS bank-conflict cycles, 9x the algorithmic HBM bytes).
// Here, as in vg_wgrad_dma.hip:
//   * the activation operand is MATERIALISED once per call by materialize_kernel (padding, virtual upsample + concat, InstanceNorm
//     apply, activation, dropout mask, noise resolved there):  P[n][ci / 16][Dp][Hp][Wp][16 ch]  bf16;
//   * the weights are packed (vg_pack_weights_dma) in exactly the order the kernel consumes them:
//         Wd[co / BN][ci / 16][tap][half: ci & 8][BN rows][8 ci]      one (panel, plane, tap group) block is contiguous;
//   * 512 threads = 8 waves (2 per SIMD): 4 along the voxels x 2 along the channels, tile BM = 256 voxels x BN = 128 / 64 channels,
//     v_mfma_f32_32x32x16_bf16, every fragment one conflict-free ds_read_b128; both operands arrive by global_load_lds_dwordx4
//     (no VGPR, no staging VALU), the weight blocks through a ring of 2-3 buffers, the activation image double-buffered per plane.
// Tiles are q-LINEAR: TD consecutive D planes x QT consecutive positions of the row-major (h, w) plane (whole rows in the LDS
// image).  The data gradients iterate over padded / parity-class grids of 17^3, 18^3, 33^3, 34^3 ... positions, which box tiles of
// 8 x 8 x 4 fill to 43-68 %; linear runs fill them to 85-97 %.
// K (= channels x taps) is split over workgroups where a launch has fewer tiles than CUs (deep levels): the slices leave fp32
// partial tiles in the caller's scratch and the slice that arrives last (ticket counter) adds them up in slice order -- the
// exchange of conv_kernel's K split (device-scope relaxed stores / loads, no fence: an agent-scope fence invalidates the XCD's L2
// under the kernels of the other streams).
// Epilogue through LDS: the fp32 tile is laid down [voxel][channel], then every thread handles 8 consecutive channels of a voxel:
// bias, residual * scale + shift, accumulate, ONE rounding to bf16, a 16-byte store (256-byte runs per voxel), and the
// per-(sample, channel) sums of the stored values for the next InstanceNorm.
#include "vg_dma_common.h"

#define VG_CD_MAXA 6          // A (activation image) DMA pieces per wave and plane (8 waves x 6 KiB = 48 KiB per buffer)
#define VG_CD_MAXW 5          // W (weight block) DMA pieces per wave and stage (8 x 5 KiB = 40 KiB per buffer)

typedef __attribute__((ext_vector_type(16))) float f32x16_d;

struct CdCls { const char* w; int tap0, nt, GT, G; int ood, ooh, oow; };
struct CdK {
    const char* P; long n_bytes; int plane_bytes; int HpA, WpA, NPL;
    int N, OD, OH, OW;                   // iteration space (the same for every class)
    int ostr, BD, BH, BW, Cout;
    int ncls; CdCls cls[8];
    int tapoff[VG_MAX_TAPS];             // byte offset of a tap inside one half of the LDS image
    int TD, QT, qtl, cptl, tiles_q, tiles;
    int HHb, HWb, IMG, nvox, nA, abuf;   // LDS image: rows / width of the halo box, 16-byte units per half (multiple of 64), pieces, bytes per buffer
    int wblk, nwb, woff, miscoff;        // bytes reserved per weight buffer, ring depth, LDS offsets
    int ncob, ks, ppk;                   // channel panels, K slices, planes per slice
    int U, upx;                          // units (class, panel, slice, sample, tile), units per XCD label
    const float* bias; const char* res; const float* rs; const float* rb; int accumulate; float* sums;
    char* out;
    float* ks_part; unsigned* ks_cnt;
    unsigned m_ow, m_hw, m_hhw;          // fast_div magics: OW; HWb; HHb * HWb
    int dbg;                             // development ablations (VG_CONV_DMA_DBG): 1 no copy waits, 2 no stage barriers, 4 no weight copies after the prologue
    // BSTAT (data gradients): the statistics pass of the IN backward that consumes this launch's output, in the epilogue -- sum dn and
    // sum dn * xhat with dn = g * mult * act'(x * scale + shift) at the reflect-folded position of the pre-norm tensor x (ConvOut::bs_* of
    // the thin-channel specialist; here also the channel-dropout multipliers and the sample aliasing of vg_actnorm_bwd_desc)
    const char* bs_x0; const char* bs_x1; int bs_c0, bs_sh, bs_act, bs_pad, bs_D, bs_H, bs_W, bs_an0, bs_ash;
    const float* bs_sc; const float* bs_sf; const float* bs_mu; const float* bs_rs; const float* bs_ml; float* bs_red;
    unsigned m_bw, m_bh;                 // fast_div magics: BW, BH
};

__global__ __launch_bounds__(256) void pack_weights_dma_kernel(const float* __restrict__ w, const int* __restrict__ tap_idx, bf16_t* __restrict__ out,
                                                               int Cin, int Cout, int ntaps, int transpose, int bn) {
    vg_pack_dma_units(w, tap_idx, out, Cin, Cout, ntaps, transpose, bn, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
extern "C" int vg_pack_weights_dma(const float* w, int T, int Cin, int Cout, const int32_t* tap_idx_dev, int ntaps, int transpose, int bn,
                                   void* out, vg_stream_t stream) {
    vg_begin();
    if (!w || !tap_idx_dev || !out || ntaps < 1 || ntaps > T || (bn != 64 && bn != 128)) return VG_EINVAL;
    const int NR = transpose ? Cin : Cout, C = transpose ? Cout : Cin;
    if ((NR % bn) || (C % 16)) return VG_EINVAL;
    const long units = (long)NR * (C / 16) * ntaps * 2;
    int blocks = (int)((units + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_dma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (const int*)tap_idx_dev, (bf16_t*)out, Cin, Cout,
                       ntaps, transpose, bn);
    return vg_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// the kernel: NW = 32-channel blocks per wave (BN = 64 * NW), MW = 32-voxel sub-tiles per wave (BM = 128 * MW)
//   acc layout (32x32x16, A = weights, B = activations): lane l holds voxel (l & 31), channels 8*jj + 4*(l >> 5) + r  (acc[4*jj + r])
// ------------------------------------------------------------------------------------------------------------------
template <int NW, int MW, int GT, bool BSTAT = false>
__global__ __launch_bounds__(512, 2) void conv_dma_kernel(const CdK p) {
    constexpr int BN = 64 * NW, BM = 128 * MW;
    constexpr int PITCH = BN * 4 + 16;                     // bytes of one voxel row of the fp32 epilogue tile (16-byte skew: conflict-free 16-byte stores)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lv = lane & 31, lk = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_d*)smem;
    int* vtab = (int*)(smem + p.miscoff);                  // [BM]: voxel index of a tile voxel in the output buffer, -1 outside the grid
    float* stat = (float*)(vtab + BM);                     // [BN][2]: this workgroup's sums of the current unit
    int* flag = (int*)(stat + BN * 2);
    // [VG_MAX_TAPS + 16]: the tap offsets in LDS (a scalar load per tap would be waited for in front of every fragment read; LDS returns in
    // order, so an offset read issued one step ahead is free).  Per class: its taps followed by a copy of its first two -- the read
    // pointer runs two taps ahead of the multiplication and wraps at the end of a plane.
    int* tapL = flag + 16;
    if (tid < VG_MAX_TAPS + 16) {
        int e = tid, ci = 0;
        while (ci + 1 < p.ncls && e >= p.cls[ci].nt + 2) { e -= p.cls[ci].nt + 2; ++ci; }
        const int nt = p.cls[ci].nt;
        tapL[tid] = e < nt + 2 ? p.tapoff[p.cls[ci].tap0 + (e < nt ? e : e - nt)] : 0;      // published by the first unit's first barrier
    }

    // ---- per-lane source offsets of the activation image's DMA pieces: the halo box has the same shape for every tile ----
    int aoffs[VG_CD_MAXA];
    const int pph = p.IMG >> 6;                            // pieces per half
#pragma unroll
    for (int k = 0; k < VG_CD_MAXA; ++k) {
        const int piece = wave + 8 * k;
        const int half = piece >= pph ? 1 : 0;
        int idx = (piece - half * pph) * 64 + lane;
        if (idx >= p.nvox) idx = 0;
        const int hd = fast_div(idx, p.m_hhw), rem = idx - hd * (p.HHb * p.HWb);
        const int hh = fast_div(rem, p.m_hw), hw = rem - hh * p.HWb;
        aoffs[k] = ((hd * p.HpA + hh) * p.WpA + hw) * 32 + half * 16;
    }
    const int n_a = wave < p.nA ? (p.nA - wave + 7) >> 3 : 0;               // this wave's A pieces per image
    const int wlane = (lk * BN + wn * NW * 32 + lv) * 16;                     // this lane's weight fragment inside a tap's [half][row] block
    const int CPT = 1 << p.cptl;                                              // 32-voxel chunks per D plane of a tile

    // one unit per workgroup.  blockIdx & 7 labels the workgroups that share an XCD (round-robin placement; speed only): label x takes
    // the units [x * upx, (x + 1) * upx), which are consecutive in (class, panel, slice) -- its workgroups stream the same weights
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int u = xcd * p.upx + jx;
    if (u >= p.U || jx >= p.upx) return;
    {
        // ---- unit -> (class, panel, slice, sample, tile) ----
        int t = u;
        const int tile = t % p.tiles; t /= p.tiles;
        const int n = t % p.N; t /= p.N;
        const int slice = t % p.ks; t /= p.ks;
        const int cob = t % p.ncob; const int ci = t / p.ncob;
        const CdCls& c = p.cls[ci];
        const int td_i = tile / p.tiles_q, qr = tile - td_i * p.tiles_q;
        const int d0 = td_i * p.TD, q0 = qr << p.qtl, h0 = fast_div(q0, p.m_ow);
        const int p_lo = slice * p.ppk, npl = p.ppk;
        constexpr int gt = GT;                                                // taps per stage (every class: the host checks)
        const int G = c.G, nst = npl * G;
        const int nWp = (gt * BN) >> 5;                                       // 1-KiB pieces of one weight block
        const int n_w = wave < nWp ? (nWp - wave + 7) >> 3 : 0;
        const int wtap = 2 * BN * 16;                                         // bytes of one tap of a weight block
        const char* pbase = p.P + (size_t)n * p.n_bytes + (size_t)((d0 * p.HpA + h0) * p.WpA) * 32;
        const char* wbase = c.w + (size_t)cob * p.NPL * c.nt * wtap;

        // ---- lane geometry of this tile; output table ----
        int abase[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int sidx = wm * MW + i, dl = sidx >> p.cptl, ch = sidx & (CPT - 1);
            const int q = q0 + ch * 32 + lv;
            int h = fast_div(q, p.m_ow), w = q - h * p.OW;
            if (q >= p.OH * p.OW) { h = h0; w = 0; }
            abase[i] = ((dl * p.HHb + (h - h0)) * p.HWb + w) * 16 + lk * p.IMG * 16;
        }
        __syncthreads();                                   // the previous unit's epilogue has read vtab / stat / the LDS tile
        if (tid < BM) {
            const int dl = tid >> p.qtl, q = q0 + (tid & (p.QT - 1)), d = d0 + dl;
            const int h = fast_div(q, p.m_ow), w = q - h * p.OW;
            const bool ok = d < p.OD && q < p.OH * p.OW;
            vtab[tid] = ok ? ((n * p.BD + d * p.ostr + c.ood) * p.BH + h * p.ostr + c.ooh) * p.BW + w * p.ostr + c.oow : -1;
        }
        if (tid < BN * 2) stat[tid] = 0.f;

        // ---- K loop: ONE continuous fragment pipeline over all (plane, tap group, tap) of the unit.  Stage = (plane, tap group): its
        // weight block sits in ring slot stage % nwb, its image in buffer plane & 1.  The only synchronisation is the EVENT in the last
        // tap of a stage (that tap's fragments are in registers by then): counted vmcnt wait for the NEXT stage's copies + barrier --
        // after it nobody reads this stage's weight slot any more (nor, in a plane's last stage, its image buffer), so the block nwb
        // stages on and the image two planes on are requested into them, behind the MFMAs of the step.  Copies complete in issue
        // order (prologue: image 0, image 1, blocks 0 .. nwb - 1; event j: [image], block j + nwb); `allow` = this wave's copies
        // issued after the younger of the two things the next stage needs.  The weight blocks of consecutive stages are consecutive
        // in memory (also across planes), the images one plane apart: two running source pointers. ----
        const int nwb = p.nwb;
        const char* asrc = pbase + (size_t)p_lo * p.plane_bytes;             // image of the next plane to request
        const char* wsrc = wbase + (size_t)p_lo * c.nt * wtap;               // weight block of the next stage to request
        const int wstage = gt * wtap;
        auto issueA = [&](int buf) {
#pragma unroll
            for (int k = 0; k < VG_CD_MAXA; ++k) {
                const int piece = wave + 8 * k;
                if (piece < p.nA) glds16(asrc, aoffs[k], lds0 + buf * p.abuf + piece * 1024);
            }
            asrc += p.plane_bytes;
        };
        const int wl0 = wave * 1024 + lane * 16;
        auto issueW = [&](int buf) {
            const unsigned dst = lds0 + p.woff + buf * p.wblk + wave * 1024;
            for (int k = 0; k < n_w; ++k) glds16(wsrc, wl0 + k * 8192, dst + k * 8192);
            wsrc += wstage;
        };
        issueA(0);
        if (npl > 1) issueA(1);
        for (int s2 = 0; s2 < nwb && s2 < nst; ++s2) issueW(s2);

        f32x16_d acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

        wait_vmcnt((min(nwb, nst) - 1) * n_w);
        __syncthreads();                                                      // block 0 and image 0 have landed; vtab / stat are written
        {
            const int tl0 = (int)((const char*)tapL - smem) + (c.tap0 + 2 * ci) * 4;         // smem offset of this class's tap offsets
            bf16x8 A[2][MW], W[2][NW];
            {
                const int to = *(const int*)(smem + tl0);
#pragma unroll
                for (int i = 0; i < MW; ++i) A[0][i] = *(const bf16x8*)(smem + abase[i] + to);
#pragma unroll
                for (int jn = 0; jn < NW; ++jn) W[0][jn] = *(const bf16x8*)(smem + p.woff + wlane + jn * 512);
            }
            // The stage loop.  Its body is straight-line: GT - 1 plain steps (multiply set t & 1, fetch tap t + 1 into the other set;
            // fragment addresses are stage base + compile-time immediates) and the event step.  Every MFMA of the K loop sits in this
            // one loop body: loop nests with MFMAs in several blocks made hipcc rename the 64 accumulator registers per block and
            // copy them at every back-edge (7 vector instructions per MFMA), a per-tap loop with run-time tap bookkeeping cost 10
            // scalar instructions per MFMA and several taken branches per tap.
            int tlp = tl0;                                                    // smem offset of the table entry of the stage's first tap
            int pl = 0, g = 0, wb = 0;
            int aoff = 0, wrun = p.woff + wlane;                              // image buffer of the current plane; this lane's fragment of the stage's first tap
#define VG_CD_LOAD(os, ao, wo)                                                                                                     \
            {                                                                                                                      \
                _Pragma("unroll") for (int i = 0; i < MW; ++i) A[os][i] = *(const bf16x8*)(smem + (abase[i] + (ao)));              \
                _Pragma("unroll") for (int jn = 0; jn < NW; ++jn) W[os][jn] = *(const bf16x8*)(smem + (wo) + jn * 512);            \
            }
#define VG_CD_MFMA(cs)                                                                                                             \
            {                                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
                _Pragma("unroll") for (int i = 0; i < MW; ++i)                                                                     \
                    _Pragma("unroll") for (int jn = 0; jn < NW; ++jn) acc[i][jn] = VG_MFMA32(W[cs][jn], A[cs][i], acc[i][jn]);     \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
            }
            for (int st = 0; st < nst; ++st) {
                int o[GT];
#pragma unroll
                for (int t = 1; t < GT; ++t) o[t] = *(const int*)(smem + tlp + 4 * t);
                const int onext = *(const int*)(smem + tlp + 4 * gt);         // first tap of the next stage (the table repeats a class's first taps behind its last)
#pragma unroll
                for (int t = 0; t < GT - 1; ++t) {
                    VG_CD_LOAD((t + 1) & 1, aoff + o[t + 1], wrun + (t + 1) * wtap)
                    VG_CD_MFMA(t & 1)
                }
                // ---- event step: the stage's last tap (set (gt - 1) & 1) ----
                const bool more = st + 1 < nst;
                int nwbuf = wb, npl_ = pl, ng = g, ntlp = tlp, naoff = aoff, nwrun = wrun, on = 0;
                if (more) {
                    int allow;
                    if (G >= nwb - 1 && st >= nwb - 1) {
                        // steady state: the events st - nwb + 2 .. st - 1 issued after the needed block; at most one of them an image
                        int nw_ = min(st - 1, nst - nwb - 1) - (st + 2 - nwb) + 1; nw_ = nw_ < 0 ? 0 : nw_;
                        allow = nw_ * n_w + ((pl >= 1 && g <= nwb - 3 && pl + 1 < npl) ? n_a : 0);
                    } else {
                        allow = 0;
                        int j2 = st - 1, pj = pl, gj = g;
                        for (; j2 >= 0; --j2) {
                            if (gj == 0) { gj = G - 1; --pj; } else --gj;
                            const bool a_j = gj == G - 1 && pj + 2 < npl, w_j = j2 + nwb < nst;
                            if (w_j && j2 + nwb == st + 1) break;
                            if (a_j && g == G - 1 && pj + 2 == pl + 1) { allow += w_j ? n_w : 0; break; }
                            allow += (a_j ? n_a : 0) + (w_j ? n_w : 0);
                        }
                        if (j2 < 0 && st + 1 < nwb) allow += (min(nwb, nst) - 2 - st) * n_w;
                    }
                    if (!(p.dbg & 1)) wait_vmcnt(allow);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (!(p.dbg & 2)) __builtin_amdgcn_s_barrier();
                    nwbuf = wb + 1 == nwb ? 0 : wb + 1;
                    if (++ng == G) { ng = 0; ++npl_; }
                    ntlp = ng == 0 ? tl0 : tlp + 4 * gt;
                    naoff = (npl_ & 1) * p.abuf;
                    nwrun = p.woff + nwbuf * p.wblk + wlane;
                    on = onext;
                }
                if constexpr ((GT - 1) & 1) {                                 // last tap in set 1: the next stage's first tap goes to set 0 ahead of the MFMAs
                    VG_CD_LOAD(0, naoff + on, nwrun)
                    VG_CD_MFMA(1)
                } else {                                                      // last tap in set 0: multiply, then refill set 0 (an LDS round trip per stage in the open)
                    VG_CD_MFMA(0)
                    VG_CD_LOAD(0, naoff + on, nwrun)
                }
                if (more) {
                    if (g == G - 1 && pl + 2 < npl) issueA(pl & 1);
                    if (st + nwb < nst && !(p.dbg & 4)) issueW(wb);
                }
                wb = nwbuf; pl = npl_; g = ng; tlp = ntlp; aoff = naoff; wrun = nwrun;
            }
#undef VG_CD_LOAD
#undef VG_CD_MFMA
        }

        // ---- K split: leave the partial tile in scratch; the slice that arrives last adds all of them in slice order ----
        if (p.ks > 1) {
            constexpr int SLOT = 512 * MW * NW * 16;
            const size_t cell = ((size_t)(ci * p.ncob + cob) * p.N + n) * p.tiles + tile;
            float* slot = p.ks_part + (cell * p.ks + slice) * SLOT;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __hip_atomic_store(slot + ((i * NW + jn) * 16 + r) * 512 + tid, acc[i][jn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this thread's partial is written through ...
            __syncthreads();                                                  // ... every thread's, before the ticket
            if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.ks_cnt + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == p.ks - 1;
            if (!last) return;
            const float* base = p.ks_part + cell * p.ks * SLOT;
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][jn][r] = __hip_atomic_load(base + ((i * NW + jn) * 16 + r) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int sl = 1; sl < p.ks; ++sl)
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][jn][r] += __hip_atomic_load(base + (size_t)sl * SLOT + ((i * NW + jn) * 16 + r) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(p.ks_cnt + cell, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // ready for the next launch on this stream
        }

        // ---- epilogue: fp32 tile -> LDS [voxel][channel]; then 8 channels of one voxel per thread ----
        __syncthreads();                                   // the last stage's fragments have been read (the tile overlays the operand buffers)
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int v = (wm * MW + i) * 32 + lv;
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ch = (wn * NW + jn) * 32 + 8 * jj + 4 * lk;
                    *(f32x4*)(smem + v * PITCH + ch * 4) = (f32x4){acc[i][jn][4 * jj], acc[i][jn][4 * jj + 1], acc[i][jn][4 * jj + 2], acc[i][jn][4 * jj + 3]};
                }
        }
        __syncthreads();
        {
            constexpr int NCG = BN / 8, NVS = 512 / NCG;                     // channel groups of 8; voxel slots
            const int cg = tid % NCG, vs = tid / NCG;
            const int co = cob * BN + cg * 8;
            float b8[8], rs8[8], rb8[8], s1[8], s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                b8[e] = p.bias ? p.bias[co + e] : 0.f;
                rs8[e] = p.res ? p.rs[n * p.Cout + co + e] : 0.f;
                rb8[e] = p.res ? p.rb[n * p.Cout + co + e] : 0.f;
                s1[e] = 0.f; s2[e] = 0.f;
            }
            if constexpr (BSTAT) {
                // data gradient with the consuming IN backward's statistics (no bias / residual / accumulate on this path: the host checks).
                // The pre-norm values x of ALL this thread's voxels are requested first (KV loads of 16 bytes in flight: one latency per unit,
                // not one per voxel), then each voxel is rounded, stored and counted: dn = g * mult * act'(x * scale + shift), xhat = (x - mean) * rstd
                // at the reflect-folded interior position of the buffer position.
                constexpr int KV = BM / NVS;
                const int nx = (p.bs_an0 > 0 && n >= p.bs_an0) ? n - p.bs_ash : n;       // forward tensors' sample (vg_actnorm_bwd_desc::alias_n0)
                float q_sc[8], q_sf[8], q_rs[8], q_nm[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = nx * p.Cout + co + e;
                    q_sc[e] = p.bs_sc ? p.bs_sc[i] : 1.f; q_sf[e] = p.bs_sc ? p.bs_sf[i] : 0.f;
                    q_rs[e] = p.bs_rs[i]; q_nm[e] = -p.bs_mu[i] * q_rs[e];
                }
                const bool lo = !p.bs_x1 || co < p.bs_c0;                                // group-uniform (bs_c0 is a multiple of 8)
                const int q_sh = (p.bs_x1 && lo) ? p.bs_sh : 0, q_cs = p.bs_x1 ? (lo ? p.bs_c0 : p.Cout - p.bs_c0) : p.Cout;
                const bf16_t* q_x = (const bf16_t*)(lo ? p.bs_x0 : p.bs_x1) + (size_t)nx * (p.bs_D >> q_sh) * (p.bs_H >> q_sh) * (p.bs_W >> q_sh) * q_cs + (lo ? co : co - p.bs_c0);
                const float q_slope = p.bs_act == VG_ACT_RELU ? 0.f : (p.bs_act == VG_ACT_LRELU ? VG_LRELU : 1.f);
                const bool q_noact = p.bs_act == VG_ACT_NONE;
                Raw8<bf16_t> xr[KV]; int idxs[KV];
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    const int idx = vtab[vs + k * NVS];
                    idxs[k] = idx;
                    // buffer position -> interior position -> transpose of the reflection pad (-1 -> 1, n -> n - 2); outside voxels read voxel 0
                    const int r = max(idx, n * (p.BD * p.BH * p.BW)) - n * (p.BD * p.BH * p.BW);
                    const int t1 = fast_div(r, p.m_bw), bw = r - t1 * p.BW, bd = fast_div(t1, p.m_bh), bh = t1 - bd * p.BH;
                    auto fold = [&](int q, int nn) { int i = q - p.bs_pad; i = i < 0 ? -i : i; i = i >= nn ? 2 * nn - 2 - i : i; return min(max(i, 0), nn - 1); };
                    const int fd = fold(bd, p.bs_D) >> q_sh, fh = fold(bh, p.bs_H) >> q_sh, fw = fold(bw, p.bs_W) >> q_sh;
                    raw_load(xr[k], q_x + ((size_t)(fd * (p.bs_H >> q_sh) + fh) * (p.bs_W >> q_sh) + fw) * q_cs);
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    const int idx = idxs[k], v = vs + k * NVS;
                    if (idx < 0) continue;
                    const f32x4 x0 = *(const f32x4*)(smem + v * PITCH + cg * 32), x1 = *(const f32x4*)(smem + v * PITCH + cg * 32 + 16);
                    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk[e] = (short)f2bf(x[e]);
                    *(bf16x8*)((bf16_t*)p.out + (size_t)idx * p.Cout + co) = pk;
                    float xv[8];
                    raw_unpack(xr[k], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float pre = xv[e] * q_sc[e] + q_sf[e];
                        const float gv = bf2f((bf16_t)pk[e]) * ((pre > 0.f || q_noact) ? 1.f : q_slope);
                        s1[e] += gv; s2[e] += gv * (xv[e] * q_rs[e] + q_nm[e]);
                    }
                }
                if (p.bs_ml) {                                                           // the channel-dropout multiplier is a per-channel constant: applied to the sums
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float m = p.bs_ml[nx * p.Cout + co + e]; s1[e] *= m; s2[e] *= m; }
                }
            } else {
            for (int v = vs; v < BM; v += NVS) {
                const int idx = vtab[v];
                if (idx < 0) continue;
                const f32x4 x0 = *(const f32x4*)(smem + v * PITCH + cg * 32), x1 = *(const f32x4*)(smem + v * PITCH + cg * 32 + 16);
                float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                const size_t o = (size_t)idx * p.Cout + co;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += b8[e];
                if (p.res) {
                    float rr[8]; load8<bf16_t>((const bf16_t*)p.res + o, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += rr[e] * rs8[e] + rb8[e];
                }
                if (p.accumulate) {
                    float oo[8]; load8<bf16_t>((const bf16_t*)p.out + o, oo);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += oo[e];
                }
                bf16x8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = (short)f2bf(x[e]);
                *(bf16x8*)((bf16_t*)p.out + o) = pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float y = bf2f((bf16_t)pk[e]); s1[e] += y; s2[e] += y * y; }
            }
            }
            float* const sums_dst = BSTAT ? p.bs_red : p.sums;
            if (sums_dst) {
                // the workgroup's sums without LDS atomics (NVS threads per channel group on the same 16 words serialise: with the statistics
                // in a data gradient's epilogue that was as long as a thin unit's K loop): partials [value][thread] over the fp32 tile's
                // memory, then thread (channel, moment) adds the NVS partials of its channel group in a fixed order
                constexpr int PR = 512 + 1;                                   // row pitch in floats (bank spread of the column sums)
                float* part = (float*)smem;
                __syncthreads();                                             // every thread is done with the tile
#pragma unroll
                for (int e = 0; e < 8; ++e) { part[(2 * e) * PR + tid] = s1[e]; part[(2 * e + 1) * PR + tid] = s2[e]; }
                __syncthreads();
                if (tid < BN * 2) {
                    const int ch = tid >> 1, mom = tid & 1;
                    const float* src = part + (2 * (ch & 7) + mom) * PR + (ch >> 3);
                    float a = 0.f;
                    for (int t = 0; t < NVS; ++t) a += src[t * NCG];
                    const int stripe = blockIdx.x & (VG_STRIPES - 1);
                    atomicAdd(&sums_dst[(((size_t)stripe * p.N + n) * p.Cout + cob * BN + ch) * 2 + mom], a);
                }
            }
        }
    }
}

template <int NW, int MW, int GT, bool BSTAT>
static void launch_cd3(const CdK& k, int grid, int lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<NW, MW, GT, BSTAT>, hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_dma_kernel<NW, MW, GT, BSTAT>), dim3(grid), dim3(512), lds, s, k);
}
// GT (template): the stage body is unrolled for this many taps; classes with fewer taps per stage skip the steps they do not have
template <int NW, int MW, bool BSTAT>
static int launch_cd2(const CdK& k, int gtmax, int grid, int lds, hipStream_t s) {
    if (gtmax == 9) launch_cd3<NW, MW, 9, BSTAT>(k, grid, lds, s);
    else if (gtmax == 8) launch_cd3<NW, MW, 8, BSTAT>(k, grid, lds, s);
    else if (gtmax == 3) launch_cd3<NW, MW, 3, BSTAT>(k, grid, lds, s);
    else if (gtmax == 4) launch_cd3<NW, MW, 4, BSTAT>(k, grid, lds, s);
    else return VG_EINVAL;
    return VG_OK;
}
template <int NW, int MW>
static int launch_cd(const CdK& k, int gtmax, int grid, int lds, hipStream_t s) {
    return k.bs_x0 ? launch_cd2<NW, MW, true>(k, gtmax, grid, lds, s) : launch_cd2<NW, MW, false>(k, gtmax, grid, lds, s);
}

// ------------------------------------------------------------------------------------------------------------------
// host side: eligibility, plan, operand pass, launch
// ------------------------------------------------------------------------------------------------------------------
struct CdPlan { int BN, TD, QT, HD, HHb, HWb, IMG, nvox, nA, abuf, wblk, nwb, lds, tiles_d, tiles_q, DpA, HpA, WpA, mn[3], ex[3]; int ncls; int nt[8], GT[8]; };

// Shape-only part of the decision (no pointers looked at): fills the plan, returns the channel panel BN (64 / 128) or 0.
static int cd_plan(const vg_conv_desc* d, CdPlan& pl) {
    if (!d || !vg_tune("CONV_DMA_FAMILY", 1)) return 0;
    const int Cin = d->c_src0 + d->c_src1;
    if (d->f32 || d->src_f32 || d->wpack || d->tanh_out || d->out_f32 || d->istr != 1) return 0;
    if (Cin < vg_tune("CONV_DMA_MINCIN", 64) || (Cin % 16) || d->Cout < 64 || (d->Cout % 64)) return 0;
    if (d->c_src1 > 0 && ((d->c_src0 % 8) || (d->c_src1 % 8))) return 0;
    if (d->ntaps < 1 || d->ntaps > VG_MAX_TAPS || d->N < 1 || d->OD < 1 || d->OH < 1 || d->OW < 1) return 0;
    if (d->ostr < 1 || d->ostr > 2) return 0;
    if (d->src0_shift && ((d->D | d->H | d->W) & 1)) return 0;
    const int BN = (d->Cout % 128 == 0 && vg_tune("CONV_DMA_BN", 128) >= 128) ? 128 : 64;
    pl.BN = BN;
    pl.ncls = d->nclass > 1 ? d->nclass : 1;
    if (pl.ncls > 8) return 0;
    if (pl.ncls > 1) {
        if (d->cls_tap0[0] != 0 || d->cls_tap0[pl.ncls] != d->ntaps) return 0;
        for (int c = 0; c < pl.ncls; ++c) {
            pl.nt[c] = d->cls_tap0[c + 1] - d->cls_tap0[c];
            if (pl.nt[c] < 1) return 0;
            for (int a = 0; a < 3; ++a) if (d->cls_iters[c][a] != (a == 0 ? d->OD : (a == 1 ? d->OH : d->OW))) return 0;     // one tile grid for all classes
        }
    } else pl.nt[0] = d->ntaps;
    if (d->ntaps < 8) return 0;                            // 1x1x1 shortcuts stay on the HBM-bound pointwise kernels
    for (int a = 0; a < 3; ++a) { pl.mn[a] = 127; pl.ex[a] = -128; }
    for (int i = 0; i < d->ntaps; ++i) {
        const int v[3] = {d->tap_d[i], d->tap_h[i], d->tap_w[i]};
        for (int a = 0; a < 3; ++a) { if (v[a] < pl.mn[a]) pl.mn[a] = v[a]; if (v[a] > pl.ex[a]) pl.ex[a] = v[a]; }
    }
    for (int a = 0; a < 3; ++a) pl.ex[a] = pl.ex[a] - pl.mn[a] + 1;
    if (d->pad_mode == VG_PAD_REFLECT) {         // one reflection only (materialize_kernel)
        const int od[3] = {d->OD, d->OH, d->OW}, nn[3] = {d->D, d->H, d->W};
        for (int a = 0; a < 3; ++a) if (pl.mn[a] < -(nn[a] - 1) || (od[a] - 1) + pl.mn[a] + pl.ex[a] - 1 > 2 * nn[a] - 2) return 0;
    }
    // tile: TD planes x QT positions of the (h, w) plane, TD * QT = 256; fewest tiles, then the smallest image.  Taps per stage GT
    // (a divisor of the tap count, <= 9; a class of <= 9 taps is one stage per plane) and the depth of the weight ring: the ring must
    // cover the copies' latency, (nwb - 1) * GT >= ~12 taps of MFMA time, with stages as long as the LDS allows.
    const int BM = 256, plane = d->OH * d->OW;
    const int tile_bytes = BM * (BN * 4 + 16), misc = BM * 4 + BN * 8 + 64 + (VG_MAX_TAPS + 16) * 4;
    const int wtap = 2 * BN * 16;
    const int ntmax = pl.nt[0];
    for (int c = 1; c < pl.ncls; ++c) if (pl.nt[c] != ntmax) return 0;       // the kernel's stage body is unrolled for ONE tap count per stage
    const bool small = false;
    long best = -1;
    for (int TD = 1; TD <= 8; TD <<= 1) {
        const int QT = BM / TD;
        if (vg_tune("CONV_DMA_TD", 0) && TD != vg_tune("CONV_DMA_TD", 0)) continue;
        int R = (QT - 1 + d->OW - 1) / d->OW + 1; if (R > d->OH) R = d->OH;
        const int HD = TD + pl.ex[0] - 1, HHb = R + pl.ex[1] - 1, HWb = d->OW + pl.ex[2] - 1;
        const int nvox = HD * HHb * HWb, IMG = ((nvox + 63) / 64) * 64;
        const int nA = 2 * IMG / 64;
        if (nA > 8 * VG_CD_MAXA) continue;
        const int abuf = 2 * IMG * 16;
        int bGT = 0, bnwb = 0; long bq = -1;
        for (int GT = small ? ntmax : 9; GT >= 1; --GT) {
            if ((ntmax % GT) || (GT != 9 && GT != 8 && GT != 4 && GT != 3)) continue;
            if (!small && vg_tune("CONV_DMA_GT", 0) && GT != vg_tune("CONV_DMA_GT", 0) && (ntmax % vg_tune("CONV_DMA_GT", 0)) == 0) continue;
            if (GT * BN / 32 > 8 * VG_CD_MAXW) continue;
            int nwb = (VG_LDS_LIMIT - misc - 2 * abuf) / (GT * wtap); if (nwb > 8) nwb = 8;
            if (vg_tune("CONV_DMA_NWB", 0) && nwb > vg_tune("CONV_DMA_NWB", 0)) nwb = vg_tune("CONV_DMA_NWB", 0);
            if (nwb < 2) { if (small) break; continue; }
            const int cover = (nwb - 1) * GT, need = vg_tune("CONV_DMA_COVER", 8);
            const long q = (cover >= need ? 1000 : 0) + (cover >= need ? GT * 10 : cover * 10) + (nwb <= 4 ? 1 : 0);
            if (q > bq) { bq = q; bGT = GT; bnwb = nwb; }
            if (small) break;                               // the classes' own tap counts are the stages
        }
        if (!bGT) continue;
        if (bnwb > 4 && (bnwb - 1) * bGT > 24) bnwb = 24 / bGT + 1;          // no deeper than useful
        if (bnwb < 2) bnwb = 2;
        int body = 2 * abuf + bnwb * bGT * wtap; if (body < tile_bytes) body = tile_bytes;
        if (body + misc > VG_LDS_LIMIT) continue;
        const int tiles_d = (d->OD + TD - 1) / TD, tiles_q = (plane + QT - 1) / QT;
        const long score = (long)tiles_d * tiles_q * 100000 + nvox + ((bnwb - 1) * bGT < vg_tune("CONV_DMA_COVER", 8) ? 40000 : 0);
        if (best < 0 || score < best) {
            best = score;
            pl.TD = TD; pl.QT = QT; pl.HD = HD; pl.HHb = HHb; pl.HWb = HWb; pl.IMG = IMG; pl.nvox = nvox; pl.nA = nA; pl.abuf = abuf;
            pl.wblk = bGT * wtap; pl.nwb = bnwb; pl.lds = body + misc; pl.tiles_d = tiles_d; pl.tiles_q = tiles_q;
            for (int c = 0; c < pl.ncls; ++c) pl.GT[c] = small ? pl.nt[c] : bGT;
            pl.DpA = tiles_d * TD + pl.ex[0] - 1;
            pl.HpA = ((tiles_q - 1) * QT) / d->OW + HHb; if (pl.HpA < d->OH + pl.ex[1] - 1) pl.HpA = d->OH + pl.ex[1] - 1;
            pl.WpA = HWb;
        }
    }
    if (best < 0) return 0;
    if ((int64_t)pl.DpA * pl.HpA * pl.WpA * 32 >= (1LL << 31)) return 0;
    return BN;
}

extern "C" int vg_conv3d_dma_bn(const vg_conv_desc* d) {
    vg_begin();
    CdPlan pl;
    return cd_plan(d, pl);
}

// Workspace with which this family runs d at its preferred plan: counters + materialised operand (a function of d->N, which the
// shape-only vg_conv3d_dma_bn cannot see) + the partial tiles of the K split it would choose given room.  0: not one of its shapes.
extern "C" int64_t vg_conv3d_scratch_bytes(const vg_conv_desc* d) {
    vg_begin();
    if (!d || !d->wlayout || d->N < 1) return 0;
    CdPlan pl;
    const int BN = cd_plan(d, pl);
    if (!BN || BN != d->wlayout) return 0;
    const int Cin = d->c_src0 + d->c_src1, NPL = Cin / 16, ncob = d->Cout / BN;
    const int64_t plane_bytes = (int64_t)pl.DpA * pl.HpA * pl.WpA * 32;
    const int64_t p_bytes = ((plane_bytes * NPL * d->N + 255) / 256) * 256;
    const long units0 = (long)pl.ncls * ncob * d->N * pl.tiles_d * pl.tiles_q;
    const long target = vg_tune("CONV_DMA_WGS", 128);
    const int64_t slot_bytes = 512LL * 2 * (BN / 64) * 16 * 4;
    int ks = 1;
    for (int c = 1; c <= NPL; ++c) {
        if (NPL % c) continue;
        if (c > 1 && units0 > VG_SCRATCH_CTR_BYTES / 4) break;
        ks = c;
        if (units0 * c >= target) break;
    }
    return VG_SCRATCH_CTR_BYTES + p_bytes + (ks > 1 ? units0 * ks * slot_bytes : 0);
}

// VG_OK: served; 1: not one of this family's shapes (only possible when d->wlayout == 0); < 0: error
int vg_conv_dma(const vg_conv_desc* d, hipStream_t s, bool* did_stats) {
    if (d->res_c1) return d->wlayout ? VG_EINVAL : 1;      // (single-channel residual: the 16-channel specialist / the generic kernel)
    CdPlan pl;
    const int BN = cd_plan(d, pl);
    if (!BN) return d->wlayout ? VG_EINVAL : 1;
    if (!d->wlayout) return 1;                             // weights in the classic layout: the caller packed for the other kernels
    if (d->wlayout != BN) return VG_EINVAL;               // packed for another panel width
    if (!d->out || !d->src0 || !d->scratch) return VG_EINVAL;
    if (d->res && (!d->res_scale || !d->res_shift)) return VG_EINVAL;
    if (pl.ncls == 1 && !d->wpacked) return VG_EINVAL;
    const int Cin = d->c_src0 + d->c_src1, NPL = Cin / 16, ncob = d->Cout / BN;
    const int tiles = pl.tiles_d * pl.tiles_q;
    const int64_t plane_bytes = (int64_t)pl.DpA * pl.HpA * pl.WpA * 32;
    const int64_t p_bytes = ((plane_bytes * NPL * d->N + 255) / 256) * 256;
    // K split: aim at ~CONV_DMA_WGS workgroups (the other lane and the weight-gradient streams fill the rest of the chip)
    const long units0 = (long)pl.ncls * ncob * d->N * tiles;
    const long target = vg_tune("CONV_DMA_WGS", 128);
    const int64_t slot_bytes = 512LL * 2 * (BN / 64) * 16 * 4;
    int ks = 1;
    const int f_ks = vg_tune("CONV_DMA_KS", 0);
    for (int c = 1; c <= NPL; ++c) {
        if (NPL % c) continue;
        if (c > 1 && (units0 > VG_SCRATCH_CTR_BYTES / 4 || VG_SCRATCH_CTR_BYTES + p_bytes + units0 * c * slot_bytes > d->scratch_bytes)) break;
        ks = c;
        if (f_ks ? c >= f_ks : units0 * c >= target) break;
    }
    if (VG_SCRATCH_CTR_BYTES + p_bytes > d->scratch_bytes) return VG_EINVAL;
    CdK k;
    k.P = (const char*)d->scratch + VG_SCRATCH_CTR_BYTES; k.n_bytes = plane_bytes * NPL; k.plane_bytes = (int)plane_bytes;
    k.HpA = pl.HpA; k.WpA = pl.WpA; k.NPL = NPL;
    k.N = d->N; k.OD = d->OD; k.OH = d->OH; k.OW = d->OW;
    k.ostr = d->ostr; k.BD = d->BD; k.BH = d->BH; k.BW = d->BW; k.Cout = d->Cout;
    k.ncls = pl.ncls;
    for (int c = 0; c < pl.ncls; ++c) {
        CdCls& q = k.cls[c];
        q.w = (const char*)(pl.ncls > 1 ? d->cls_w[c] : d->wpacked);
        if (!q.w) return VG_EINVAL;
        q.tap0 = pl.ncls > 1 ? d->cls_tap0[c] : 0; q.nt = pl.nt[c]; q.GT = pl.GT[c]; q.G = pl.nt[c] / pl.GT[c];
        q.ood = pl.ncls > 1 ? d->cls_ooff[c][0] : d->ooff_d; q.ooh = pl.ncls > 1 ? d->cls_ooff[c][1] : d->ooff_h; q.oow = pl.ncls > 1 ? d->cls_ooff[c][2] : d->ooff_w;
    }
    for (int c = pl.ncls; c < 8; ++c) k.cls[c] = k.cls[0];
    for (int i = 0; i < VG_MAX_TAPS; ++i)
        k.tapoff[i] = i < d->ntaps ? (((d->tap_d[i] - pl.mn[0]) * pl.HHb + (d->tap_h[i] - pl.mn[1])) * pl.HWb + (d->tap_w[i] - pl.mn[2])) * 16 : 0;
    k.TD = pl.TD; k.QT = pl.QT; k.qtl = ilog2_exact(pl.QT); k.cptl = ilog2_exact(pl.QT / 32); k.tiles_q = pl.tiles_q; k.tiles = tiles;
    k.HHb = pl.HHb; k.HWb = pl.HWb; k.IMG = pl.IMG; k.nvox = pl.nvox; k.nA = pl.nA; k.abuf = pl.abuf;
    k.wblk = pl.wblk; k.nwb = pl.nwb; k.woff = 2 * pl.abuf;
    k.miscoff = pl.lds - (256 * 4 + BN * 8 + 64 + (VG_MAX_TAPS + 16) * 4);
    k.ncob = ncob; k.ks = ks; k.ppk = NPL / ks;
    const long U = units0 * ks;
    if (U >= (1L << 30)) return VG_EINVAL;
    k.U = (int)U; k.upx = (int)((U + 7) / 8);
    k.bias = d->bias; k.res = (const char*)d->res; k.rs = d->res_scale; k.rb = d->res_shift; k.accumulate = d->accumulate; k.sums = d->out_sums;
    k.out = (char*)d->out;
    k.ks_cnt = (unsigned*)d->scratch; k.ks_part = (float*)((char*)d->scratch + VG_SCRATCH_CTR_BYTES + p_bytes);
    auto magic = [](int dd) { return dd <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)dd + 1ULL); };
    k.dbg = vg_tune("CONV_DMA_DBG", 0);
    k.m_ow = magic(d->OW); k.m_hw = magic(pl.HWb); k.m_hhw = magic(pl.HHb * pl.HWb);
    const int grid = 8 * k.upx;                            // one unit per workgroup
    // the statistics of the consuming IN backward in the epilogue (vg_conv3d has checked that b describes exactly this launch's output)
    k.bs_x0 = nullptr; k.bs_red = nullptr;
    const vg_actnorm_bwd_desc* b = d->bstat;
    if (b && vg_tune("CONV_DMA_BSTAT", 1) && b->norm && !b->f32 && !b->x_f32 && b->x && b->mean && b->rstd && b->red && b->C == d->Cout
        && !d->accumulate && !d->out_sums && !d->res && !d->bias && (!b->x1 || (b->c_x0 > 0 && b->c_x0 < b->C && (b->c_x0 % 8) == 0))
        && (!b->x1 || !b->x0_shift || !((b->D | b->H | b->W) & 1))) {
        k.bs_x0 = (const char*)b->x; k.bs_x1 = (const char*)b->x1; k.bs_c0 = b->x1 ? b->c_x0 : b->C; k.bs_sh = (b->x1 && b->x0_shift) ? 1 : 0;
        k.bs_act = b->act; k.bs_pad = b->g_padded ? 1 : 0; k.bs_D = b->D; k.bs_H = b->H; k.bs_W = b->W;
        k.bs_an0 = b->alias_n0; k.bs_ash = b->alias_shift;
        k.bs_sc = b->scale; k.bs_sf = b->shift; k.bs_mu = b->mean; k.bs_rs = b->rstd; k.bs_ml = b->mult; k.bs_red = b->red;
        k.m_bw = magic(d->BW); k.m_bh = magic(d->BH);
    }
    if (vg_dry("conv_dma<%d,%d>|td%d|gt%d|nwb%d|ks%d|cls%d|walk%d|bs%d", BN, 256, pl.TD, pl.GT[0], pl.nwb, ks > 1 ? 1 : 0, pl.ncls > 1 ? 1 : 0, U > 256 ? 1 : 0, k.bs_x0 ? 1 : 0)) return VG_OK;
    if (k.bs_x0 && did_stats) *did_stats = true;
    MatK m;
    m.src0 = d->src0; m.src1 = d->src1; m.c0 = d->c_src0; m.c1 = d->c_src1; m.shift0 = d->src0_shift ? 1 : 0;
    m.N = d->N; m.D = d->D; m.H = d->H; m.W = d->W; m.Cin = Cin;
    m.in_scale = d->in_scale; m.in_shift = d->in_shift; m.act = d->act;
    m.noise = (const bf16_t*)d->noise; m.npad = d->noise ? d->noise_pad : 0; m.pad_mode = d->pad_mode;
    m.pmin_d = pl.mn[0]; m.pmin_h = pl.mn[1]; m.pmin_w = pl.mn[2]; m.Dp = pl.DpA; m.Hp = pl.HpA; m.Wp = pl.WpA;
    m.deint = 0; m.WE = 0; m.Wps = pl.WpA; m.out = (bf16_t*)k.P;
    vg_launch_materialize(m, s);
    int gtmax = 0; for (int c = 0; c < pl.ncls; ++c) if (pl.GT[c] > gtmax) gtmax = pl.GT[c];
    const int lrc = BN == 128 ? launch_cd<2, 2>(k, gtmax, grid, pl.lds, s) : launch_cd<1, 2>(k, gtmax, grid, pl.lds, s);
    if (lrc != VG_OK) return lrc;
    return vg_check_launch();
}
