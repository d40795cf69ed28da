/* vangan_hip.h -- C ABI of libvangan_hip.so (MI355X / gfx950 only).
 *
 * The reference (psweens/VAN-GAN) has NO native plugin/FFI interface: its hot path sits behind
 * the Python class VanGan (vangan.py:20-550) and executes inside TensorFlow.  These entry points
 * are what a ctypes binding on the reference side would call in place of the TensorFlow ops of
 * that path; each one cites the reference lines whose arithmetic it replaces.
 *
 * Rules of the boundary:
 *   - every pointer is a DEVICE pointer owned by the caller, valid on `stream` for the call;
 *   - the library never allocates/frees device memory and never synchronises: enqueue-only (launch plans that need
 *     device workspace take it from the caller: vg_conv_desc::scratch, the scratch argument of vg_conv3d_wgrad);
 *   - every entry returns 0 on success, <0 on error (vg_status_string); no exception or abort
 *     crosses the ABI;
 *   - activations are NDHWC (channel innermost). "bf16" buffers hold 16-bit brain floats.
 */
#ifndef VANGAN_HIP_H
#define VANGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vg_stream_t;          /* hipStream_t */

#define VG_OK 0
#define VG_EINVAL (-1)              /* bad argument / unsupported shape */
#define VG_ELDS (-2)                /* tile does not fit the 160 KiB LDS */
#define VG_ELAUNCH (-3)             /* HIP launch failure */

#define VG_ACT_NONE 0
#define VG_ACT_RELU 1
#define VG_ACT_LRELU 2              /* LeakyReLU(0.2), discriminator.py:75 */

#define VG_PAD_ZERO 0               /* Keras 'same' zero padding */
#define VG_PAD_REFLECT 1            /* ReflectionPadding3D, building_blocks.py:30-39 */

#define VG_MAX_TAPS 64

/* Per-(n,c) reduction buffers (InstanceNorm sums, backward reductions) are STRIPED: [VG_STRIPES][N][C][2]; a
 * workgroup adds into stripe (its index mod VG_STRIPES) so that float atomics from 10^4 workgroups do not serialise
 * on one 128-byte line; consumers sum the stripes. */
#define VG_STRIPES 8

const char* vg_status_string(int code);
/* diagnostic builds only: device buffer receiving s_memtime stamps of the conv kernel phases (NULL = off) */
int vg_set_stamp_buffer(void* dev_u64);
int vg_version(void);
/* The 16-bit storage format of THIS build: 0 = bfloat16 (libvangan_hip.so: training and inference), 1 = IEEE half precision
 * (libvangan_hip_h.so: the same sources compiled with -DVG_FP16 -- the fp16 sliding-window inference of BASELINE config 5,
 * post_training.py:38-39 / custom_callback.py:174-175).  Every "bf16" buffer of this header holds that format. */
int vg_storage16(void);
/* sizeof() of the descriptor structs as THIS library was compiled (which: 0 vg_conv_desc, 1 vg_actnorm_bwd_desc,
 * 2 vg_pack_item, 3 vg_fin_desc; else VG_EINVAL): a binding that mirrors the structs by hand checks its layout at load time. */
int vg_abi_sizeof(int which);

/* ---------------------------------------------------------------------------------------------
 * Gather-convolution (implicit GEMM on bf16 MFMA).  One descriptor covers
 *   - Conv3D forward for every variant on the path (resunet_model.py:42-143,
 *     building_blocks.py:126-196, discriminator.py:50-117): k3/k1/k4, stride 1/2, reflect or zero
 *     padding folded into the index math, InstanceNorm-apply + ReLU/LeakyReLU (+ channel-dropout,
 *     + GaussianNoise) applied on read, nearest-2x upsample + concat read virtually
 *     (resunet_model.py:175-181), bias / residual-add / tanh / next-InstanceNorm statistics in
 *     the epilogue;
 *   - its data gradient (what tf.GradientTape computes for d/d input of Conv3D): the same kernel
 *     on dY with transposed packed weights, per output-parity class for stride 2.
 * out[n, o*ostr+ooff, co] (+)= sum_taps sum_ci f(src[n, bnd(o*istr + tap), ci]) * W[tap][ci][co]
 * --------------------------------------------------------------------------------------------- */
struct vg_actnorm_bwd_desc_s;
/* InstanceNorm finalisation of a convolution's OUTPUT by the launch that produces it (resunet_model.py:23-39 / building_blocks.py:190:
 * tfa InstanceNormalization = per-(n, c) mean and biased variance over the volume, y = (x - mean) * rsqrt(var + eps) * gamma + beta).
 * The producing kernel already accumulates (sum, sum of squares) of what it stores (out_sums); with a vg_fin_desc the workgroup that
 * finishes LAST (ticket) turns them into the on-read affine of up to two consuming norms -- scale = gamma * rstd (* mult),
 * shift = (beta - mean * gamma * rstd) (* mult), and mean / rstd for the backward pass -- instead of a vg_in_finalize launch
 * between producer and consumer (120 launches of a train step, each on a lane's dependent chain).  A tensor may feed two norms
 * (an encoder output: the next block's first norm and, as the skip half of a virtual concat, a decoder block's): job j writes
 * channels [c_off, c_off + Cout) of arrays with c_tot channels per sample.  Kernel families without the epilogue are followed by a
 * small kernel inside vg_conv3d: either way the arrays are complete, in stream order, when the call's launches have run. */
typedef struct {
    const float* gamma; const float* beta;    /* [c_tot] of the consuming norm (NULL: 1 / 0) */
    const float* mult;                        /* [N][c_tot] SpatialDropout3D multipliers folded into scale / shift, or NULL */
    float* scale; float* shift;               /* [N][c_tot] */
    float* mean; float* rstd;                 /* [N][c_tot] or NULL */
    int32_t c_off, c_tot;
} vg_fin_job;
typedef struct {
    uint32_t* ticket;                         /* one zeroed word per launch (left non-zero) */
    float count;                              /* voxels per (n, c) of the produced tensor */
    float eps;
    int32_t njobs;                            /* 1 or 2 */
    vg_fin_job job[2];
} vg_fin_desc;
typedef struct {
    /* input: virtual concat of src0 (c_src0 channels) and src1 (c_src1 channels, may be 0) */
    const void* src0;
    const void* src1;
    int32_t c_src0, c_src1;
    int32_t src0_shift;      /* 1: src0 is stored at half resolution, read at (d>>1,h>>1,w>>1) */
    int32_t src_f32;         /* 1: sources are float32 (only with a single input channel) */
    int32_t N, D, H, W;      /* logical input grid (full resolution) */
    const float* in_scale;   /* [N][Cin] on-read affine (InstanceNorm apply) or NULL */
    const float* in_shift;
    int32_t act;             /* VG_ACT_* applied after the affine */
    const void* noise;       /* bf16 [N][D+2np][H+2np][W+2np][Cin] added after act, or NULL */
    int32_t noise_pad;       /* np: 1 when the noise lives on the reflect-padded grid */
    /* geometry */
    int32_t istr;            /* input stride per output step */
    int32_t pad_mode;        /* VG_PAD_* for out-of-range input positions */
    int32_t ntaps;
    int8_t tap_d[VG_MAX_TAPS], tap_h[VG_MAX_TAPS], tap_w[VG_MAX_TAPS];  /* input offset of tap */
    int32_t OD, OH, OW;      /* iteration space: output positions per dim */
    int32_t ostr, ooff_d, ooff_h, ooff_w;   /* buffer position = o*ostr + ooff */
    int32_t BD, BH, BW;      /* output buffer grid */
    int32_t Cout;
    /* weights: bf16 [round_up(Cout,64)][Ktot], packed by vg_pack_weights with the same CK/taps */
    const void* wpacked;
    int32_t CK;              /* contraction channels staged per chunk (multiple of 16) */
    const float* bias;       /* [Cout] or NULL */
    /* epilogue */
    const void* res;         /* bf16 [N][BD][BH][BW][Cout] residual operand or NULL */
    const float* res_scale;  /* [N][Cout] affine on the residual (InstanceNorm of the shortcut) */
    const float* res_shift;
    int32_t tanh_out;        /* tanh after everything else; with accumulate: out = tanh(out + value) (last tap chunk of a 7^3 head) */
    void* out;               /* bf16 (or f32 when out_f32) [N][BD][BH][BW][Cout] */
    int32_t out_f32;
    int32_t accumulate;      /* out += value (data-gradient accumulation) */
    float* out_sums;         /* [VG_STRIPES][N][Cout][2] += (sum, sum of squares) of the stored values, or NULL */
    int32_t f32;             /* exact-parity mode: every "bf16" buffer of this call (multi-channel sources, res, out,
                                packed weights, and dy/dgrad operands) is float32 and the MFMA is the f32 16x16x4 form */
    /* Fused output-parity classes (data gradient of a strided Conv3D, vangan.py:426-438 via tf.GradientTape): with
       nclass > 1 ONE launch stages each dY halo tile once and produces the outputs of all classes.  Class c uses the
       taps tap_*[cls_tap0[c] .. cls_tap0[c+1]), its own packed weights cls_w[c] (packed with those taps and CK), the
       output offset cls_ooff[c] (replaces ooff_*) and cls_iters[c] outputs per axis (replaces OD/OH/OW, which must
       hold the per-axis maximum).  With one channel chunk (c_src0 + c_src1 <= CK, bf16, no noise) the classes share the
       staged tile; otherwise the launch is class-parallel (workgroup -> one class), any number of chunks, all classes
       packed with the same CK.  nclass 0 or 1: fields unused. */
    int32_t nclass;
    int32_t cls_tap0[9];
    const void* cls_w[8];
    int32_t cls_ooff[8][3];
    int32_t cls_iters[8][3];
    /* W-packed single-channel source (the Conv3D layers that read a 1-channel volume: resunet_model.py:44-60 stem,
       discriminator.py:50-60 first conv).  wpack = k > 1: the k taps along W become k pseudo-channels, the descriptor lists
       only the k*k (d, h) taps (tap_w = 0), wpack_wmin is the W offset of the first tap (-pad_before).  The packed weights
       are the same DHWIO kernel read as [k*k taps][k channels][Cout] (vg_pack_weights with ntaps = k*k, Cin = k), and
       vg_conv3d_wgrad returns dw in that layout, i.e. the unchanged DHWIO tensor (T_total = k*k).  0: off. */
    int32_t wpack;
    int32_t wpack_wmin;
    /* Optional (data-gradient launches): the STATISTICS pass of the IN backward that consumes this launch's output (g == out,
       bf16, one launch covering the whole grid, no accumulate).  The 16-channel specialist accumulates sum dn and sum dn*xhat
       in its epilogue -- the upstream gradient is not re-read from HBM -- other kernels are followed by vg_actnorm_bwd_stats.
       Either way the striped sums in red are complete when the call returns; the caller then runs vg_actnorm_bwd_apply (which adds
       the stripes up and produces dgamma / dbeta). */
    const struct vg_actnorm_bwd_desc_s* bstat;
    /* Optional caller-owned device workspace for launches on THIS stream (the library never allocates).  Bytes
       [0, VG_SCRATCH_CTR_BYTES) are arrival counters: zeroed ONCE by the caller when it allocates the buffer, left at zero by every
       launch.  The rest carries no state between launches (fp32 partial tiles of K-split launches, the materialised operand of the
       LDS-DMA convolution).  Two launches may share a scratch only if they are ordered (same stream).  NULL or too small: the
       library chooses launch plans that need none. */
    void* scratch;
    int64_t scratch_bytes;
    /* Layout of wpacked / cls_w: 0 = rows x Ktot (vg_pack_weights, the gather kernels); 64 / 128 = the block layout of the LDS-DMA
       convolution with that channel-panel width (vg_pack_weights_dma; the caller asks vg_conv3d_dma_bn which layers take it and needs
       a scratch big enough for the materialised operand).  A call whose weights are in the block layout is served by that kernel
       family or fails with VG_EINVAL -- never by a silent fallback. */
    int32_t wlayout;
    /* Optional (forward launches with out_sums): finalise the InstanceNorm statistics of this launch's output for its consumers
       (see vg_fin_desc). */
    const vg_fin_desc* fin;
    /* res_c1 != 0: `res` is a SINGLE-channel fp32 volume [N][BD][BH][BW] (same grid as the output, ostr == 1) whose value is broadcast
       over the output channels: out += res[v] * res_scale[n][c] + res_shift[n][c].  The stem's shortcut Conv3D(16, 1x1x1)(x) ->
       InstanceNorm (resunet_model.py:96-99) is an affine function of the input volume per channel, so the block's residual add reads
       the 4-byte volume instead of a stored 16-channel tensor (vg_stem_short_fwd produces the scale / shift).  Served by the 16-channel
       specialist and the generic kernel; other families return VG_EINVAL. */
    int32_t res_c1;
    /* Optional, forward launches whose src0 is the virtually upsampled half-resolution tensor (src0_shift != 0, c_src0 a multiple of 16,
       reflection pad, 3x3x3): the class panels of vg_pack_up_weights for the c_src0 / 16 upsampled channel chunks.  The 16-channel
       specialist then contracts those chunks over the half-resolution image with the D / H taps collapsed (12 instead of 27 taps per
       chunk; UpSampling3D + IN + ReLU + reflect pad + Conv3D of resunet_model.py:175-181 and :42-66 -- IN and ReLU commute with the
       upsampling, the reflection pad becomes edge replication at half resolution).  NULL: the plain 27-tap form. */
    const void* wpacked_up;
} vg_conv_desc;
#define VG_SCRATCH_CTR_BYTES 16384

int vg_conv3d(const vg_conv_desc* d, vg_stream_t stream);

/* bytes of dynamic LDS the chosen tile needs, or <0 */
int vg_conv3d_lds_bytes(const vg_conv_desc* d);
/* the launch plan vg_conv3d would use for this descriptor: plan[0] = channel panel BN, plan[1] = voxels per tile,
   plan[2] = LDS bytes, plan[3] = workgroups; lets the host compare channel-chunk sizes (CK) before packing weights */
int vg_conv3d_plan(const vg_conv_desc* d, int32_t* plan4);
/* Shape-only: 2 when the two-panel instance of the thin-channel specialist (3x3x3, stride 1, 16-channel chunks, 32 output channels
   per workgroup: the 32-channel layers at 64^3) would serve d -- the host then packs those weights with CK = 16; else 0.  d->CK is
   ignored.  (Replaces nothing in the reference: tile planning of resunet_model.py:36-43,82-95's Conv3D calls.) */
int vg_conv3d_thin_np(const vg_conv_desc* d);
/* Dry run of vg_conv3d: the whole dispatch runs, nothing is launched, and buf receives the name of the kernel variant the
   call would launch, e.g. "conv<bf16,16,8,n0,wl1,dma0,mc0,c10>|walk1|ch0" (template arguments, then walk = a workgroup
   visits more than one tile, ch = several channel chunks per tile) or "pw_cto1<bf16,2>".  The parity tests use it to prove
   that every variant the BASELINE configurations run is compared with the oracle (tests/test_variant_coverage.py). */
int vg_conv3d_variant(const vg_conv_desc* d, char* buf, int buflen);
/* Host-side heuristic switches (forced tile shapes, persistent grid sizes, ...): key without the VG_ prefix of the
   environment variable that sets the same switch, e.g. ("CONV_MSUB", 4).  reset != 0: back to environment / default.
   Testing and tuning aid; the defaults are what the benchmarks run. */
int vg_set_tuning(const char* key, int value, int reset);

/* Class panels of the collapsed upsampled chunks (vg_conv_desc::wpacked_up) from the fp32 DHWIO kernel w[3][3][3][Cin][Cout] of a decoder
 * block's first convolution whose first c_up input channels (a multiple of 16) are the upsampled tensor:
 *   out[chunk][class = pd*2 + ph][co][(td*2 + th)*3 + tw][16 channels of the chunk]   (16-bit storage format of the build), where for an
 *   output voxel of parity p along an axis the collapsed tap t = 0 sums the original taps {-1} (p = 0) or {-1, 0} (p = 1) and t = 1 sums
 *   {0, +1} (p = 0) or {+1} (p = 1); sums in fp32, one rounding. */
int vg_pack_up_weights(const float* w, int Cin, int Cout, int c_up, void* out, vg_stream_t stream);

/* Pack fp32 Keras DHWIO weights [T][Cin][Cout] to the bf16 layout vg_conv3d reads.
 * transpose=0: rows = Cout, contraction = Cin (forward); transpose=1: rows = Cin, contraction =
 * Cout (data gradient).  tap_idx[i] selects the source tap of packed tap i.  Returns Ktot>0. */
int vg_pack_weights(const float* w, int T, int Cin, int Cout, const int32_t* tap_idx_dev, int ntaps,
                    int transpose, int CK, void* out, int out_f32, vg_stream_t stream);
/* Table-driven repack of every packed operand of a network in ONE launch: items_dev is a device array of n
 * vg_pack_item (all pointers device pointers).  Item i is served by the nblk >= 1 blocks starting at blk0 of a 1-D grid of
 * total_blocks = sum(nblk) blocks (the caller sizes nblk in proportion to the operand, the ranges must tile the grid). */
typedef struct {
    const float* w; const int32_t* tap_idx; void* out;
    int32_t Cin, Cout, ntaps, transpose, CK, out_f32;
    int32_t blk0, nblk;
    int32_t bn;              /* > 0: the block layout of vg_pack_weights_dma with this panel width (CK / out_f32 unused) */
    int32_t pad_;
} vg_pack_item;
int vg_pack_weights_multi(const vg_pack_item* items_dev, int n, int total_blocks, vg_stream_t stream);
/* The LDS-DMA convolution (vg_conv_dma.hip: both MFMA operands staged by global_load_lds, the wide layers of discriminator.py:64-117
 * and resunet_model.py:103-143 and their data gradients).  vg_conv3d_dma_bn: shape-only query -- the channel-panel width (64 / 128)
 * with which vg_conv3d would serve this descriptor through that family, 0 if it would not; no pointer of d is looked at.
 * vg_pack_weights_dma: fp32 DHWIO [T][Cin][Cout] -> bf16 [rows / bn][contraction / 16][tap][8-channel half][bn rows][8], rows =
 * output channels (transpose 0) or input channels (transpose 1, data gradient); rows * contraction * ntaps elements. */
int vg_conv3d_dma_bn(const vg_conv_desc* d);
/* Bytes of vg_conv_desc::scratch with which a call whose weights are in the block layout (wlayout != 0) runs at its preferred plan:
 * counters + the materialised operand -- which grows with d->N, the one thing the shape-only query above cannot see -- + the partial
 * tiles of its K split.  A smaller scratch drops the K split first; one that cannot hold the operand makes vg_conv3d return VG_EINVAL
 * (block-layout weights have no other kernel).  0 for every other descriptor.  The caller sizes / grows its workspace with this. */
int64_t vg_conv3d_scratch_bytes(const vg_conv_desc* d);
int vg_pack_weights_dma(const float* w, int T, int Cin, int Cout, const int32_t* tap_idx_dev, int ntaps, int transpose, int bn,
                        void* out, vg_stream_t stream);
int vg_packed_ktot(int ntaps, int C, int CK);
int vg_packed_rows(int N);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of the same gather-convolution (tf.GradientTape d/dW of Conv3D, vangan.py:426-438):
 * dW[tap][ci][co] += sum_{n,o} f(src[n, bnd(o*istr+tap), ci]) * dY[n,o,co];  db[co] += sum dY.
 * Uses the input fields of vg_conv_desc (src*, transform, taps, OD/OH/OW); `dy` is bf16 (or f32 when
 * dy_f32) [N][OD][OH][OW][Cout]; dw is fp32 DHWIO [T_total][Cin][Cout] indexed by tap_idx_host[i].
 * --------------------------------------------------------------------------------------------- */
int vg_conv3d_wgrad(const vg_conv_desc* d, const void* dy, int dy_f32, const int32_t* tap_idx_host,
                    int T_total, float* dw, float* db, float* scratch, int64_t scratch_bytes, vg_stream_t stream);
/* dry run of vg_conv3d_wgrad (see vg_conv3d_variant): "wgrad<bf16,8,1,n0>|bm256|cib16|part1|walk1" */
int vg_conv3d_wgrad_variant(const vg_conv_desc* d, int dy_f32, const int32_t* tap_idx_host, int T_total,
                            int64_t scratch_bytes, char* buf, int buflen);
/* scratch (optional, device): when many workgroups share one dW element their slabs are stored to
 * scratch[workgroup column][T_total*Cin*Cout] and summed in a fixed order by a second kernel instead of float atomics. */

/* ---------------------------------------------------------------------------------------------
 * InstanceNorm helpers (tfa InstanceNormalization, resunet_model.py:36, building_blocks.py:190)
 * --------------------------------------------------------------------------------------------- */
/* scale/shift[n][c] for on-read normalisation from accumulated striped (sum,sumsq) [VG_STRIPES][N][c][2]; channels
 * [0,c0) come from sums0 (count0 voxels), [c0,c0+c1) from sums1.  mult[n][c] (SpatialDropout3D mask, >=0) optional.
 * Also writes mean/rstd [N][C] when non-NULL (needed by the backward). */
int vg_in_finalize(const float* sums0, int c0, float count0, const float* sums1, int c1, float count1,
                   const float* gamma, const float* beta, const float* mult, int N, float eps,
                   float* scale, float* shift, float* mean, float* rstd, vg_stream_t stream);

/* Backward of  a = mult * act(x*scale+shift)  [InstanceNorm -> activation -> dropout], with the
 * upstream gradient g read either plain ([N][D][H][W][C]) or folded from the reflect-padded grid
 * ([N][D+2][H+2][W+2][C], transpose of ReflectionPadding3D).
 * pass 1 (stats):  red[stripe][n][c][0] += sum dn,  red[stripe][n][c][1] += sum dn*xhat,   dn = g*mult*act'(.)   (striped partial
 *                  sums; the apply pass adds the stripes up and produces dgamma / dbeta)
 * pass 2 (apply):  dx (+)= gamma*rstd*(dn - mean(dn) - xhat*mean(dn*xhat))      (norm=1)
 *                  dx (+)= dn                                                     (norm=0)
 * x is bf16 (x_f32=0) ; g is bf16; dx is bf16 unless dx_f32. */
typedef struct vg_actnorm_bwd_desc_s {
    const void* g; int32_t g_padded;
    const void* x; int32_t x_f32;            /* forward input of the norm (channels [0,c_x0) when x1 is set) */
    const void* x1; int32_t c_x0; int32_t x0_shift;   /* virtual upsample+concat: x = [up(x), x1] */
    int32_t N, D, H, W, C;
    const float* scale; const float* shift;   /* on-read affine of the forward, [N][C], or NULL */
    const float* mult;                        /* [N][C] or NULL */
    int32_t act; int32_t norm;
    const float* gamma; const float* mean; const float* rstd;   /* norm=1 */
    float* red;                               /* [VG_STRIPES][N][C][2], zeroed by the caller */
    void* dx; int32_t dx_f32; int32_t accumulate;
    int32_t dx_cstride, dx_coff;              /* dx channel stride / offset (write into a slice) */
    int32_t f32;                              /* exact-parity mode: g, x, x1 and dx are float32 */
    float* dgamma;                            /* optional [C]: the APPLY pass adds d/d gamma and d/d beta of the InstanceNorm */
    float* dbeta;                             /* (summed over samples and stripes) instead of a separate vg_in_param_grads */
    int32_t* ticket;        /* unused (the stripes are added up by the apply pass); kept for layout stability */
    /* Several upstream gradients of ONE forward sample in one launch (the discriminator's two backward sweeps -- the critic loss over
       [real; fake] and the generator loss through the fake half, vangan.py:426-438 -- as one 3B-sample sweep): samples n >= alias_n0 of g /
       dx / red read x, x1 and every per-(sample, channel) array (scale, shift, mean, rstd, mult) of sample n - alias_shift.  0: off.
       pgrad_n > 0: only samples < pgrad_n add to dgamma / dbeta (the generator-loss gradient updates no discriminator parameter). */
    int32_t alias_n0, alias_shift, pgrad_n, pad_;
} vg_actnorm_bwd_desc;
int vg_actnorm_bwd_stats(const vg_actnorm_bwd_desc* d, vg_stream_t stream);
int vg_actnorm_bwd_apply(const vg_actnorm_bwd_desc* d, vg_stream_t stream);
/* the apply passes of two independent norms of one batch (same N; statistics of both complete) in one launch -- a residual block's shortcut
 * norm and the norm in front of its second convolution (resunet_model.py:103-143 under the tape); two launches where the pair does not share
 * a kernel instance */
int vg_actnorm_bwd_apply2(const vg_actnorm_bwd_desc* d1, const vg_actnorm_bwd_desc* d2, vg_stream_t stream);
/* both passes in one call (statistics only when d->norm) */
int vg_actnorm_bwd(const vg_actnorm_bwd_desc* d, vg_stream_t stream);
/* The stem's shortcut in the forward pass (resunet_model.py:96-99: Conv3D(C, 1x1x1)(x) -> InstanceNorm on the single-channel volume x [N][S],
 * fp32) WITHOUT materialising its output: w[c]*x + b[c] normalises to gamma[c]*w[c]*rs*(x - mean x) + beta[c] with
 * rs = (w[c]^2 var x + eps)^-1/2 (the convolution's bias cancels), i.e. the branch is  scale[n][c] * x + shift[n][c]  with
 *     scale = gamma*w*rs,   shift = beta - scale * mean x.
 * One launch: per-workgroup partial sums of x and x^2 in part[N][G][2] (doubles), fixed-order sum and the 2*N*C results by the
 * workgroup that draws the last of the N*G tickets (*ticket zero on entry, left at zero).  G = vg_stem_short_fwd_workgroups(N, S).
 * The consuming convolution adds the branch through vg_conv_desc::res_c1.  round16 as in vg_stem_short_bwd. */
int vg_stem_short_fwd_workgroups(int N, int64_t S);
int vg_stem_short_fwd(const float* x, int N, int64_t S, int C, const float* w, const float* gamma, const float* beta, float eps, int round16,
                      float* scale, float* shift, double* part, int G, unsigned* ticket, vg_stream_t stream);
/* The stem's shortcut in the backward pass (resunet_model.py:96-99: Conv3D(16, 1x1x1)(x) -> InstanceNorm, no activation; its input is the
 * single-channel volume, so no data gradient exists).  The branch's output w[c]*x + b[c] normalises to
 * xhat[c] = w[c]*rs[c]*(x - mean x), rs[c] = (w[c]^2 var x + eps)^-1/2: the loss sees w[c] only through eps, and with the two moments
 * R0 = sum_v g, T = sum_v g*x of the gradient g of the block output [N][S][C] (16-bit storage or fp32) against x [N][S] (fp32),
 *     dL/dw[c] += sum_n eps*gamma[c]*rs^3*(T - mean(x) R0),   dgamma[c] += sum_n w[c]*rs*(T - mean(x) R0),   dbeta[c] += sum_n R0,
 *     dL/db[c] = 0 identically
 * -- no apply pass, no gradient tensor, no weight-gradient launch, and no read of the stored branch output (whose 16-bit rounding the
 * 1/w of the older statistics-based form amplified to 6-13 % of this gradient).  Deterministic: part[N][G][2C+2] doubles receive one
 * partial sum per workgroup (fp32 inside a wave, double across waves), the workgroup drawing the last of the N*G tickets (*ticket zeroed by the caller) adds them in a fixed
 * order.  G = vg_stem_short_bwd_workgroups(N, S, C); C in {8, 16, 32, 64}.  w: the C kernel weights as the forward used them
 * (round16 != 0: rounded to the library's 16-bit storage format first). */
int vg_stem_short_bwd_workgroups(int N, int64_t S, int C);
int vg_stem_short_bwd(const void* g, int g_f32, const float* x, int N, int64_t S, int C, const float* w, const float* gamma, float eps,
                      int round16, float* dw, float* dgamma, float* dbeta, double* part, int G, unsigned* ticket, vg_stream_t stream);
/* dgamma[c] += sum_{stripes,n} red[.][n][c][1], dbeta[c] += sum red[.][n][c][0] */
int vg_in_param_grads(const float* red, int N, int C, float* dgamma, float* dbeta, vg_stream_t stream);

/* Backward of the virtual upsample+concat (resunet_model.py:175-181): g is bf16 [N][D][H][W][Cu+Cs];
 * dlow[N][D/2][H/2][W/2][Cu] (+)= sum of the 8 children, dskip[N][D][H][W][Cs] (+)= g[..., Cu:].
 * accumulate: bit 0 -- add to dlow (else overwrite), bit 1 -- add to dskip: the first writer of a gradient buffer overwrites,
 * so the buffers need no memset.  Cs == 0 (dskip ignored): the backward of a bare UpSampling3D (generator.py:58-66), a 2x2x2 sum-pool. */
int vg_concat_bwd(const void* g, int N, int D, int H, int W, int Cu, int Cs, void* dlow, void* dskip,
                  int f32, int accumulate, vg_stream_t stream);

/* The data gradient of a decoder block's 1x1x1 shortcut convolution (resunet_model.py:126-131 over the concat of :175-181) FUSED with
 * vg_concat_bwd: d is the descriptor of the accumulating data-gradient launch (src0 = gradient of the shortcut's output, out = the
 * gradient of the virtual concat [N][D][H][W][c_low + Cs] holding the convolution branch's part; it is only READ); the sums go to
 * dskip / (over every 2x2x2 block) dlow with vg_concat_bwd's accumulate bits -- the concat gradient is neither rewritten nor re-read
 * (804 -> 486 MB per application at 128^3).  Returns VG_OK when launched, 1 when the shape is not served (the caller then runs
 * vg_conv3d(d) + vg_concat_bwd), < 0 on error. */
int vg_shortcut_dgrad_concat(const vg_conv_desc* d, void* dlow, void* dskip, int c_low, int accumulate, vg_stream_t stream);
/* The same with the block's first convolution branch folded in: the concat gradient is never stored.  b describes the (InstanceNorm ->
 * act) backward of that convolution's input (g = its data gradient on the reflection-padded grid, x / x1 = the virtual concat's
 * sources, statistics already in b->red: vg_actnorm_bwd_stats or the data-gradient epilogue); the launch computes
 * dx = gamma * rstd * (dn - mean(dn) - xhat * mean(dn * xhat)) per voxel where vg_actnorm_bwd_apply would have written it, adds the
 * shortcut's data gradient, and writes dskip / dlow as above; d->out is not used; b->dgamma / b->dbeta are added as the apply pass does.
 * 838 -> 436 MB per application at 128^3.  Returns 1 when the shape is not served (caller: vg_actnorm_bwd_apply + the call above). */
int vg_shortcut_dgrad_concat_norm(const vg_conv_desc* d, const vg_actnorm_bwd_desc* b, void* dlow, void* dskip, int c_low,
                                  int accumulate, vg_stream_t stream);

/* out = act(a * a_scale + a_shift) + (b * b_scale + b_shift): layers.add([input_tensor, InstanceNorm(conv2)]) of the ResNet generator's
 * residual block (building_blocks.py:68-123; generator.py:52-56), both operands [N][S][C] bf16 (f32: float) with their pending on-read
 * affine [N][C] (NULL: none); a_act a VG_ACT_* applied to the first operand after its affine. */
int vg_affine_add(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b, const float* b_scale,
                  const float* b_shift, int N, int64_t S, int C, void* out, int f32, vg_stream_t stream);

/* db[c] += sum_rows dy[row][c]: the bias gradient of a layer from its output gradient dy [rows][C] (bf16, or float when f32).
 * Conv3D's comes out of vg_conv3d_wgrad; this entry serves the k2 s2 Conv3DTranspose of the 'deconv' decoder (resunet_model.py:168-174,
 * vnet_model.py:244-245), whose forward IS the strided data gradient of vg_conv3d (output-parity classes, one tap each, bias in the
 * epilogue), whose data gradient is the forward convolution with the same packed weights and whose kernel gradient is
 * vg_conv3d_wgrad with the roles of input and output gradient exchanged (van_gan_amd.ops.ConvTranspose3dK2S2 is the recipe). */
int vg_bias_grad(const void* dy, int f32, int64_t rows, int C, float* db, vg_stream_t stream);

/* Data gradient of a 4x4x4 stride-2 Conv3D over ReflectionPadding3D(1) of a SINGLE-channel volume (discriminator.py:50-60 under the tape,
 * vangan.py:433-438: the generator loss reaches the generators through it) as a stride-1 convolution over CELLS: cell c = the 2x2x2 block
 * of reflect-padded positions 2c + r, all of which draw from dY voxels c - 1 + n, n in {0,1} per axis, with tap r + 2 - 2n.
 * vg_pack_cell_weights: fp32 DHWIO kernel w [4][4][4][1][C] -> the packed operand (16-bit, [64][(C/16)*448]: rows 4rd+2rh+rw < 8, 27 taps
 * n - 1 in -1..1 of which the n = 2 ones are zero, CK = 16) of a vg_conv3d call with src0 = dY [N][D/2][H/2][W/2][C], zero padding,
 * OD/OH/OW = D/2+1 .., Cout = 16, out = cells.  vg_cells_fold: cells (16-bit [N][D/2+1][H/2+1][W/2+1][16]) -> dx fp32 [N][D][H][W]: depth to
 * space and the transpose of the reflection pad (positions 0 and n+1 fold onto 1 and n-2).  D, H, W even and >= 4. */
int vg_pack_cell_weights(const float* w, int C, void* out, vg_stream_t stream);
int vg_cells_fold(const void* cells, int N, int D, int H, int W, float* dx, vg_stream_t stream);

/* d_pre = dy * (1 - y*y)   (tanh output activation, resunet_model.py:245), all fp32 */
int vg_tanh_bwd(const float* dy, const float* y, float* dpre, int64_t n, vg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss kernels (fp32 single-channel volumes [B][S])
 * --------------------------------------------------------------------------------------------- */
/* mm[b] = (min, max, n_min, n_max) -- utils.py:27-48 */
int vg_minmax(const float* x, int B, int64_t S, float* mm4, vg_stream_t stream);
int vg_minmax_apply(const float* x, const float* mm4, int B, int64_t S, float* y, vg_stream_t stream);
/* dx = gy/(max-min) + [x==min] * sum(gy*(y-1))/(R n_min) - [x==max] * sum(gy*y)/(R n_max);
 * tmp2[b][2] is scratch (zeroed by the caller) */
int vg_minmax_bwd(const float* x, const float* y, const float* gy, const float* mm4, int B, int64_t S,
                  float* tmp2, float* dx, vg_stream_t stream);
/* BCE of loss_functions.py:185-190 on normalised volumes: acc[0] += sum bce; gp (+)= gscale*dbce/dp */
int vg_bce(const float* t, const float* p, int64_t n, float* acc, float gscale, float* gp, int accumulate,
           vg_stream_t stream);
/* acc[0] += sum (a-b)^2 ; gb (+)= gscale * 2 (b-a) -- loss_functions.py:56-68 */
int vg_mse(const float* a, const float* b, int64_t n, float* acc, float gscale, float* gb, int accumulate,
           vg_stream_t stream);
/* LSGAN terms on patch logits (loss_functions.py:273-274,306-308):
 * acc[0] += sum (target - x)^2 ; gx (+)= gscale * 2 (x - target); x bf16 or f32 */
int vg_mse_const(const void* x, int x_f32, float target, int64_t n, float* acc, float gscale, float* gx,
                 int accumulate, vg_stream_t stream);
/* SSIM (loss_functions.py:86-117): acc[0] += sum (1-ssim); part[3][B][S] = dL/d(mu_p, E[pp], E[tp]) */
int vg_ssim_fwd(const float* t, const float* p, int B, int D, int H, int W, float* acc, float* part,
                vg_stream_t stream);
int vg_ssim_bwd(const float* t, const float* p, const float* part, int B, int D, int H, int W, float gscale,
                float* gp, int accumulate, vg_stream_t stream);

/* clDice soft skeleton (clDice_func.py:8-80).  imgs: [iters+2][B][S] receives img_0..img_{iters+1}
 * (img_{j+1} = soft_erode(img_j)); skels: [iters+1][B][S] receives the skeleton after every step, the
 * last slab is soft_skel(img, iters).
 * aux == NULL (a skeleton that is not differentiated: the target's): the erosion chain runs two steps per launch on tiles held in LDS
 * with a 2-voxel halo, the skeleton recursion is one launch over the stored chain.
 * aux != NULL ((iters+1) * B*S * 6 bytes: [iters+1][B][S] float delta_j, then [iters+1][B][S] uint8 arg-max codes of the dilation in
 * delta_j = relu(img_j - dilate(img_{j+1})), then [iters+1][B][S] uint8 arg-min codes of the erosion img_{j+1} = erode(img_j); a code
 * is ((a+1)*3 + (b+1))*3 + (c+1) for the offset (a, b, c) along (D, H, W) of the FIRST extremum in TensorFlow's scan order, TP): the
 * forward pass files where every pooling gradient will go, and vg_soft_skel_bwd with the same aux routes by table lookup. */
int vg_soft_skel_fwd(const float* img, int B, int D, int H, int W, int iters, float* imgs, float* skels, void* aux,
                     vg_stream_t stream);
/* gimg += (d skel / d img)^T gskel.  aux != NULL (what vg_soft_skel_fwd filed): iters + 2 streaming launches, work = [4][B][S] scratch
 * (no initialisation needed); aux == NULL: two scan launches per step that recompute the arg-extrema from imgs, work = [3][B][S]. */
int vg_soft_skel_bwd(const float* imgs, const float* skels, const float* gskel, int B, int D, int H, int W,
                     int iters, float* work, float* gimg, const void* aux, vg_stream_t stream);
/* Dice + clDice combination (clDice_func.py:83-149, loss_functions.py:223-226) without a host sync.
 * sums7 = (sum skel_p*t, sum skel_p, sum t, sum skel_t*p, sum skel_t, sum p, sum t*p);
 * coef6: gskel_p = c0*t - c1;  gp += c2*t + c3 + c4*skel_t;  c5 = w*((1-alpha)*dice + alpha*clDice). */
int vg_cldice_coef(const float* sums7, float w, float alpha, float* coef6, vg_stream_t stream);
int vg_cldice_grads(const float* t, const float* skel_t, const float* coef6, int64_t n, float* gskel_p, float* gp,
                    int accumulate, vg_stream_t stream);
/* sums[k] += (sum a*b, sum a, sum b) over n elements */
int vg_dot_sums(const float* a, const float* b, int64_t n, float* sums3, vg_stream_t stream);
/* y (+)= alpha*a + beta*b (b may be NULL) */
int vg_axpby(const float* a, float alpha, const float* b, float beta, int64_t n, float* y, int accumulate,
             vg_stream_t stream);

/* Wasserstein mode -- what `wasserstein=True` of the reference trains once traced (its gradient penalty never reaches a weight: DESIGN.md
 * section 8): the discriminator's Flatten -> Dropout(0.2) -> Dense(1) head over the patch logits (discriminator.py:116-119),
 *   z[s] = b + sum_i w[i] * mask[s][i] * x[s][i]          (mask: dropout multipliers {0, 1/(1-rate)} or NULL),
 * its backward  dx[s][i] = gz[s] * mask[s][i] * w[i],  dw[i] += sum_s gz[s] * mask[s][i] * x[s][i],  db += sum_s gz[s]  (dx / dw / db
 * optional), and the loss terms of loss_functions.py:325-355 on z = [real (B); fake (B)]: acc2[0] += sum z_real, acc2[1] += sum z_fake,
 * gz_d[2B] = d(-reduce_mean(D(real) - D(fake)))/dz = (-inv ..., +inv ...), gz_g[B] = d(-reduce_mean(D(fake)))/dz_fake = -inv,
 * inv = 1 / (B * global batch size) (reduce_mean(axis=None) averages over the batch too, loss_functions.py:21-22). */
int vg_dense_head_fwd(const float* x, const float* mask, const float* w, const float* b, int N, int n, float* z, vg_stream_t stream);
int vg_dense_head_bwd(const float* x, const float* mask, const float* w, const float* gz, int N, int n, float* dx, float* dw, float* db,
                      vg_stream_t stream);
int vg_wasserstein_terms(const float* z, int B, float inv, float* acc2, float* gz_d, float* gz_g, vg_stream_t stream);

/* Sliding-window inference (GanMonitor.stitch_subvolumes, custom_callback.py:47-223): pred/cnt [X][Y][Z] fp32.
 * vg_overlap_add: pred[box] += window[crop], cnt[box] += 1 for the border-cropped box of one k^3 window at (x0,y0,z0)
 * (custom_callback.py:165-183); vg_divide_crop: out = pred/cnt on the un-padded sub-box (:192-200; 0/0 = NaN as numpy). */
int vg_overlap_add(const float* win, int kx, int ky, int kz, int px, int py, int pz, int x0, int y0, int z0,
                   int X, int Y, int Z, float* pred, float* cnt, vg_stream_t stream);
int vg_divide_crop(const float* pred, const float* cnt, int X, int Y, int Z, int sx, int sy, int sz, int ox, int oy,
                   int oz, float* out, vg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-tensor Adam with per-tensor clip-by-norm (tf.keras.optimizers.Adam(2e-4, 0.5, 0.9,
 * clipnorm=100), vangan.py:220-235, applied by optimizer.minimize at vangan.py:426-438).
 * w,g,m,v: flat fp32 buffers of `total` elements; seg_off[T+1] tensor boundaries (device int64);
 * norms[T + 2 * ceil(total / 4096)] scratch (T squared norms, then per-block partial sums: the norms
 * are added in a fixed order, so replicas holding the same reduced gradients apply bit-identical
 * updates).  grad_scale multiplies g before clipping (1/world for mean, 1 for the reference's SUM
 * all-reduce).
 * --------------------------------------------------------------------------------------------- */
int vg_adam_clip(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T,
                 int64_t total, float* norms, float lr_t, float beta1, float beta2, float eps,
                 float clipnorm, float grad_scale, vg_stream_t stream);

/* The three entry points whose per-step host scalars change from step to step -- the Philox counter, the noise standard deviation
 * (GanMonitor decays it per epoch, custom_callback.py:413-424) and Adam's bias-corrected rate lr_t -- with those scalars read from
 * DEVICE memory: a HIP graph captured over one VanGan.train_step (vangan.py:380-440) is then replayable, the host refreshing a
 * 32-byte parameter block before each replay instead of re-enqueueing ~900 launches.  offset = *offset_dev + offset_add. */
int vg_randn_bf16_dev(void* out, int64_t n, const float* std_dev, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_add,
                      vg_stream_t stream);
int vg_dropout_mask_dev(float* out, int64_t n, float rate, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_add,
                        vg_stream_t stream);
int vg_adam_clip_dev(float* w, const float* g, float* m, float* v, const int64_t* seg_off_dev, int T, int64_t total, float* norms,
                     const float* lr_t_dev, float beta1, float beta2, float eps, float clipnorm, float grad_scale, vg_stream_t stream);

/* Writes that 32-byte parameter block: bytes [0, 8) the Philox counter base, [8, 12) the noise standard deviation, [16, 32) lr_t of
 * gen_IS, gen_SI, disc_I, disc_S (the reference's four optimizers, vangan.py:220-235).  The values travel as KERNEL ARGUMENTS, i.e. they
 * are bound at enqueue time: a host that enqueues step N+1 before step N has executed cannot disturb step N (block: 8-byte aligned). */
int vg_set_step_params(void* block, uint64_t offset, float std, float lr0, float lr1, float lr2, float lr3, vg_stream_t stream);

/* Stand-in for the RCCL SUM all-reduce of a gradient bucket (the implicit all-reduce of optimizer.minimize under MirroredStrategy,
 * vangan.py:426-438; main.py:22) on a box with ONE GPU, so that the data-parallel schedule -- communication stream, per-bucket
 * events, early suffix pieces, cross-step optimizer overlap -- can be timed without a second device: `workgroups` workgroups (RCCL
 * channels) move buf[0, n) to scratch and back (2 x 4n bytes read + written, buf ends bit-identical; n a multiple of 4, both
 * pointers 16-byte aligned) and then hold their CUs until min_us microseconds of wall clock have passed since they started (the
 * time a ring over xGMI would take for the message; 0: none).  Never part of a real multi-GPU run. */
int vg_local_exchange(float* buf, float* scratch, int64_t n, int workgroups, int min_us, vg_stream_t stream);

/* N(0,std) noise as bf16 and SpatialDropout3D channel masks {0,1/(1-rate)} (fp32), counter-based RNG */
int vg_randn_bf16(void* out, int64_t n, float std, uint64_t seed, uint64_t offset, vg_stream_t stream);
int vg_dropout_mask(float* out, int64_t n, float rate, uint64_t seed, uint64_t offset, vg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training data pipeline on resident volumes (DatasetGen.process_imaging_domain / process_seg_domain /
 * random_spatial_augmentation, dataset.py:205-251).  vol: fp32 [X][Y][Z][C]; out: fp32 [px][py][pz][C].
 * vg_crop_augment: out = rot90_k( flip_up_down?( flip_left_right?( vol[x0:x0+px, y0:y0+py, z0:z0+pz] ))) with the
 *   tf.image semantics the reference gets on 4-D tensors: X is a batch axis, "height" = Y, "width" = Z, i.e.
 *   flip_left_right reverses Z, flip_up_down reverses Y, rot90 turns the (Y,Z) plane counter-clockwise k times
 *   (k is taken modulo 4; odd k needs py == pz).  The random draws stay with the caller.
 * vg_crop_max: out[0] = max over the crop box (the seg-domain rejection test reduce_max(arr) < 0.8, dataset.py:242).
 * --------------------------------------------------------------------------------------------- */
int vg_crop_augment(const float* vol, int X, int Y, int Z, int C, int x0, int y0, int z0, int px, int py, int pz,
                    int flip_lr, int flip_ud, int rot_k, float* out, vg_stream_t stream);
int vg_crop_max(const float* vol, int X, int Y, int Z, int C, int x0, int y0, int z0, int px, int py, int pz,
                float* out, vg_stream_t stream);

/* Device memset-to-zero / device-to-device copy on an explicit stream (hipMemsetAsync / hipMemcpyAsync): what a recorded launch list
 * (van_gan_amd.VanGan.record_train_step) replays in place of torch's zero_() / copy_(), which would go to torch's current stream. */
int vg_memset_zero(void* p, int64_t nbytes, vg_stream_t stream);
int vg_copy_bytes(void* dst, const void* src, int64_t nbytes, vg_stream_t stream);

/* f32 <-> bf16 copies */
int vg_f32_to_bf16(const float* x, void* y, int64_t n, vg_stream_t stream);
int vg_bf16_to_f32(const void* x, float* y, int64_t n, vg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
