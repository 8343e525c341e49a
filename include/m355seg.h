/*
 * m355seg.h — C ABI of libm355seg.so: the MI355X (gfx950) 3D U-Net hot path.
 *
 * This is the drop-in boundary for the efirdc/Segmentation-Pipeline hot path
 * (SURVEY.md §8b).  The reference has no native code: every entry point below
 * replaces a stock torch.nn op the reference calls from Python, and the comment
 * on each one cites the reference call site (paths relative to the reference
 * repo root) it stands in for.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - All tensors are float32 (M355_F32), NCDHW, spatial dims dense
 *    (stride of W == 1, H == W, D == H*W, C == D*H*W).  The batch stride is
 *    explicit (in elements) wherever a tensor may be a channel slice of a
 *    larger concat buffer; 0 means dense (C*D*H*W).
 *  - Caller owns every buffer, including workspace.  The library never
 *    frees device memory and keeps no pointer to a tensor or workspace after a
 *    call returns.  The ONE piece of device state it keeps is the work-queue
 *    pool of the queue-driven conv kernels (256 KB per device: 4096 slots of
 *    eight ticket counters + an exit counter, all zero between launches).  A
 *    caller that hands that pool over with m355_queue_pool_set() before its
 *    first conv launch on a device -- the Python host does, from torch's
 *    allocator -- gets a library that makes no device allocation at all;
 *    otherwise the pool is hipMalloc'ed + hipMemset on first use (never inside
 *    a stream capture) and lives until the process exits.
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t passed as
 *    void*); no implicit synchronisation.
 *  - Return 0 (M355_OK) or a negative status; never throws, never aborts.
 *    m355_last_error() returns a thread-local message for the last failure.
 */
#ifndef M355SEG_H
#define M355SEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: m355_conv3d_desc.reserved became `flags` (packed weights, fused softmax); c8 / 16-bit entry points; ensembles;
 * padded sliding window; weight standardisation
 * 3: c8-only training flow (norm / pool / conv-transpose / trilinear / dropout / space-to-depth backward on c8 tensors,
 *    loss-scaled pack / unpack), device-side sampler, one-pass grid aggregation, caller-provided work-queue pool
 *    (m355_queue_pool_*), fused norm-backward tail */
#define M355_ABI_VERSION 3

enum {
  M355_OK = 0,
  M355_EINVALID_ARG = -1,
  M355_EUNSUPPORTED = -2,
  M355_ELAUNCH = -3,
  M355_EWORKSPACE = -4
};

enum { M355_F32 = 0 };

/* arithmetic of the 3x3x3 convolutions:
 *   M355_COMPUTE_F32  exact fp32: v_mfma_f32_32x32x2_f32, a k-ordered fp32 fma chain (default)
 *   M355_COMPUTE_BF16 operands rounded to bf16 (round-to-nearest-even), v_mfma_f32_32x32x16_bf16 with
 *                     fp32 accumulation (BASELINE cfg3).
 *   M355_COMPUTE_F16  the same with IEEE fp16 operands, v_mfma_f32_32x32x16_f16 (BASELINE cfg5:
 *                     "mixed fp16 with MFMA channel-GEMM path"); values beyond +-65504 become inf.
 *   M355_COMPUTE_F32X3 fp32 tensors in and out, fp32 results, bf16 matrix pipe: every operand is split EXACTLY into
 *                     three bf16 values (x = hi + mid + lo, 8 + 8 + 8 significant bits; each step exact in fp32) and
 *                     six of the nine plane products -- all but the three below 2^-24 of the product -- run on
 *                     v_mfma_f32_32x32x16_bf16 with fp32 accumulation (conv3d_f32x3.hip): 6 MFMAs of K = 16 stand for
 *                     8 fp32 MFMAs of K = 2 at a sixteenth of the cost per K.  Measured against fp64 the results are
 *                     as accurate as fp32 arithmetic is: max error 2e-7 .. 1.5e-6 of max |result| on the BASELINE
 *                     layers, within 2.5x of the fp32 MFMA kernels' (profiles/r04_f32x3_accuracy.txt; both carry the
 *                     error of an fp32 accumulation).  Applies to the 3x3x3 / stride 1 / pad 1 forward and data gradient of
 *                     layers with >= 8 K-channels and > 4 M-channels, and to the weight gradient of layers with > 4
 *                     channels on both sides, at least two z planes and < 2^24 voxels per sample, and to the forward of
 *                     the k = 2 / stride 2 conv-transpose (m355_conv_transpose3d_fwd); every other descriptor and entry
 *                     point treats it as M355_COMPUTE_F32.  This is what the host side's
 *                     default precision "fp32" passes (ops.py); "fp32_mfma" passes M355_COMPUTE_F32.
 * The 16-bit modes apply to conv3d fwd, bwd_data and bwd_weight (the edge layers' weight gradient runs in exact fp32).
 * In the 16-bit modes the convolution kernels read their input in the "c8" layout
 *     x16[n][cb][voxel][8]   cb = channel block of 8 (zero-padded past C), 16-bit elements,
 * i.e. the 8 channels of a voxel are one aligned 16-byte item = one MFMA operand fragment (h16.hpp).
 * The *_h16 entry points take such tensors directly (the normalisation / pooling passes of the model
 * path write them, m355_norm_act_fwd_h16 ...); the plain entry points accept fp32 NCDHW and convert
 * into their workspace first. */
enum { M355_COMPUTE_F32 = 0, M355_COMPUTE_BF16 = 1, M355_COMPUTE_F16 = 2, M355_COMPUTE_F32X3 = 3 };

/* activation fused into the normalise pass (components.py:26,54-55) */
enum { M355_ACT_NONE = 0, M355_ACT_RELU = 1, M355_ACT_LEAKY_RELU = 2 };

int m355_version(void);
const char* m355_last_error(void);
/* The M355_* tuning overrides (test / sweep hooks: M355_CONV_NTW, M355_CONV_KSPLIT, M355_CONV_SLOTS, ...) are
 * read from the environment once, when the library is loaded; call this after changing them. */
void m355_reload_tuning(void);
/* Work-queue pool (see "Conventions"): `zeroed_device_buffer` = m355_queue_pool_bytes() bytes of zero-filled device
 * memory on `device`, 64-byte aligned, owned by the caller and kept alive (and otherwise untouched) for as long as the
 * library is used on that device.  Call before the first conv launch on the device; M355_EUNSUPPORTED once a different
 * pool is in use there.  Every queue-driven launch takes a 64-byte slot keyed by (device, stream, capture id): launches
 * on different streams, and launch sequences captured into different hipGraphs, never share one. */
size_t m355_queue_pool_bytes(void);
int m355_queue_pool_set(void* zeroed_device_buffer, size_t bytes, int32_t device);
/* fp16 training flow (loss-scaled activation gradients, "grad_scale" below): `device_word` = a zeroed 4-byte device word
 * owned by the caller (NULL: off).  From then on, on that device, every kernel that rounds a loss-scaled gradient to fp16
 * ORs bit 0 into it when it had to clamp a value to +-65504 (the gradient was clipped: the scale is too large), and every
 * epilogue that removes the loss scale from a parameter gradient (grad_unscale != 1) ORs bit 1 when its result is not
 * finite.  The host reads and clears the word after a backward pass and, when it is non-zero, skips the optimizer step and
 * lowers the scale (segmentation_pipeline_amd.ops.fp16_overflow(); the reference has no reduced-precision path). */
int m355_overflow_flag_set(void* device_word, int32_t device);

/* ------------------------------------------------------------------ conv3d
 * Replaces nn.Conv3d as used by Block3d (models/components.py:36,42,51) and the
 * out conv (models/modular_unet.py:83,99), and the strided F.conv3d of
 * BlurConv3d (components.py:119).  Cubic kernel k, isotropic stride / zero pad.
 *   y[n,o,z,y,x] = bias[o] + add[n,o,z,y,x]
 *                + sum_{c,dz,dy,dx} w[o,c,dz,dy,dx] * x[n,c,z*s+dz-p, ...]
 * `bias` and `add` may be NULL.  `add` has y's layout (residual branch fusion,
 * components.py:67-68).  Weight layout [Cout,Cin,k,k,k] (torch).
 */
typedef struct m355_conv3d_desc {
  int32_t N, Cin, Cout;
  int32_t D, H, W;          /* INPUT spatial size of the forward op */
  int32_t k, stride, pad;
  int32_t out_pad;          /* conv-transpose only */
  int64_t x_batch_stride;   /* elements; 0 = dense */
  int64_t y_batch_stride;   /* elements; 0 = dense */
  int32_t compute;          /* M355_COMPUTE_* (3x3x3 / s1 / p1; M355_COMPUTE_F32X3 also k2 s2 conv-transpose fwd; ignored elsewhere) */
  int32_t flags;            /* M355_CONV_* bits, 0 by default */
} m355_conv3d_desc;

/* desc->flags */
enum {
  /* the `w` argument of m355_conv3d_fwd(_stats|_h16) / m355_conv3d_bwd_data(_h16) points at weights packed by
   * m355_conv3d_pack for THIS descriptor (same shapes, compute mode and direction) instead of the torch-layout
   * filter: the per-launch repacking kernel is skipped.  Weights only change at optimizer.step, so a caller
   * packs once per parameter version.  The buffer is read-only for the conv kernels: any number of launches, on any
   * streams, may share it concurrently (the work-queue state of the queue-driven kernels lives in a per-(stream,
   * capture) slot of the work-queue pool, see "Conventions" above and m355_queue_pool_set). */
  M355_CONV_W_PACKED = 1,
  /* forward only: nn.Softmax(dim=1) over the Cout output channels applied in the conv epilogue (the out conv +
   * hypothesis of ModularUNet, models/modular_unet.py:99-100); allowed when m355_conv3d_fuses_softmax(desc) != 0
   * (the fp32 Cout <= 4 kernel, where a voxel's channels sit in one thread's registers).  Same bits as
   * m355_conv3d_fwd followed by m355_softmax_fwd. */
  M355_CONV_SOFTMAX = 2
};
int32_t m355_conv3d_fuses_softmax(const m355_conv3d_desc* d);
/* which: 0 = forward, 1 = data gradient.  0 bytes = this descriptor has no packed form (generic direct kernels). */
size_t m355_conv3d_packed_bytes(const m355_conv3d_desc* d, int32_t which);
int m355_conv3d_pack(const m355_conv3d_desc* d, int32_t which, const float* w, void* packed, void* stream);
/* The same for many weights in one launch: after optimizer.step (segmentation_trainer.py:259) every conv weight of a
 * model needs its forward and data-gradient forms again -- ~37 packs of 5-15 us each, back to back on the critical
 * path of a cfg2 train step.  Each item is packed exactly as m355_conv3d_pack(&item.desc, item.which, item.w,
 * item.packed) would (identical bytes); `packed` must hold m355_conv3d_packed_bytes(&item.desc, item.which). */
typedef struct m355_pack_item {
  m355_conv3d_desc desc;
  int32_t which;
  const float* w;
  void* packed;
} m355_pack_item;
int m355_conv3d_pack_batch(const m355_pack_item* items, int32_t n, void* stream);

size_t m355_conv3d_fwd_workspace(const m355_conv3d_desc* d);
int m355_conv3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w,
                    const float* bias, const float* add, float* y,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Forward with the statistics of the following normalisation fused into the epilogue
 * (Block3d: conv -> norm, models/components.py:51-53): besides y the kernel writes, per sample n,
 * slot p and output channel o, the (sum, sum of squares) of the y values one wave stored:
 *   stat_partials[((n * P + p) * Cout + o) * 2 + {0,1}],   P = m355_conv3d_stats_slots(desc)
 * (every slot of every channel is written; slots of waves that lie outside the volume hold zeros).
 * m355_norm_stats_from_partials() turns them into mean / rstd without reading y again.
 * P == 0 means this descriptor has no fused statistics (not 3x3x3 s1 p1, the fp32 Cout <= 4 kernel, or a
 * split-K plan): call m355_conv3d_fwd + m355_norm_stats instead. */
int64_t m355_conv3d_stats_slots(const m355_conv3d_desc* d);
int m355_conv3d_fwd_stats(const m355_conv3d_desc* d, const float* x, const float* w,
                          const float* bias, const float* add, float* y, float* stat_partials,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- c8 tensors (16-bit compute modes) ----
 * bytes of a dense c8 tensor; conversion fp32 NCDHW <-> c8 (round-to-nearest-even; batch strides in
 * ELEMENTS of the respective tensor, 0 = dense). */
size_t m355_act16_bytes(int32_t N, int32_t C, int64_t S);
int m355_act16_pack(const float* x, void* x16, int32_t N, int32_t C, int64_t S, int64_t x_batch_stride,
                    int64_t x16_batch_stride, int32_t compute, void* stream);
int m355_act16_unpack(const void* x16, float* x, int32_t N, int32_t C, int64_t S, int64_t x16_batch_stride,
                      int64_t x_batch_stride, int32_t compute, void* stream);
/* conv3d forward / data gradient (3x3x3 s1 p1, desc->compute = BF16 | F16) on a c8 input: x16 holds
 * desc->Cin channels (fwd), dy16 desc->Cout channels (bwd_data); outputs are fp32 NCDHW exactly as in
 * m355_conv3d_fwd(_stats) / m355_conv3d_bwd_data (stat_partials may be NULL).  Workspace:
 * m355_conv3d_h16_workspace(desc, which) bytes, which = 0 forward, 1 data gradient. */
size_t m355_conv3d_h16_workspace(const m355_conv3d_desc* d, int32_t which);
int m355_conv3d_fwd_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride, const float* w,
                        const float* bias, const float* add, float* y, float* stat_partials, void* workspace,
                        size_t workspace_bytes, void* stream);
/* forward whose OUTPUT is c8 as well (written by the epilogue: lanes exchange channel halves and store whole
 * 16-byte items): the pre-norm tensor of conv -> norm/act -> conv never exists in fp32.  No fused `add`. */
/* statistics slots of m355_conv3d_fwd_h16_c8 (differs from m355_conv3d_stats_slots for split-K plans, whose
 * reduction pass emits the partials: one slot per block of that pass) */
int64_t m355_conv3d_stats_slots_c8(const m355_conv3d_desc* d);
int m355_conv3d_fwd_h16_c8(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride, const float* w,
                           const float* bias, void* y16, int64_t y16_batch_stride, float* stat_partials,
                           void* workspace, size_t workspace_bytes, void* stream);
int m355_conv3d_bwd_data_h16(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride, const float* w,
                             float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* weight gradient of the 3x3x3 / stride 1 / padding 1 conv with BOTH operands in c8: x16 = the packed conv input the
 * forward pass consumed, dy16 = the packed output gradient the data gradient consumes (so the 16-bit training flow
 * converts each tensor once).  dw fp32 [Cout][Cin][3][3][3]; dbias (may be NULL) is reduced from the fp32 `dy`
 * (NCDHW, desc->y_batch_stride; may be NULL when dbias is).  Replaces the reference's autograd conv weight
 * gradient under `torch.cuda.amp.autocast` (segmentation_pipeline/segmentation_trainer.py:203-227).  Volumes
 * below 2^25 voxels per sample. */
size_t m355_conv3d_bwd_weight_h16_workspace(const m355_conv3d_desc* d);
int m355_conv3d_bwd_weight_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                               int64_t dy16_batch_stride, const float* dy, float* dw, float* dbias, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Introspection for profiling: which kernel variant a 3x3x3 conv dispatches to.
 * which: 0 = forward, 1 = data gradient, 2 = weight gradient (of m355_conv3d_bwd_weight).  out[0] = kernel family:
 * 0 generic direct kernel, 1 MFMA implicit GEMM (one output tile per workgroup), 3 the same as a persistent kernel
 * (workgroups walk several tiles), 2 z-Toeplitz small-Cout kernel, 4 the 16-bit operand kernel (persistent), 5 its
 * 8-wave double-buffered variant, 6 its one-item-per-workgroup variant, 7 the split kernel of M355_COMPUTE_F32X3
 * (conv3_f32x3_kernel); weight gradient: 8 conv3_bww_x3_kernel, 9 conv3_mfma_bww2(c)_kernel, 10 the edge-layer kernel
 * conv3_mfma_bww_small_kernel, 11 the c8 kernel behind an operand pack (16-bit modes).  out[1] = voxel groups per wave
 * (NTW), out[2] = lanes along x per group (GX; weight gradient: tile width), out[3] = split-K factor (weight gradient:
 * voxel-range splits).  Pure host function. */
int m355_conv3d_plan(const m355_conv3d_desc* d, int32_t which, int32_t* out4);

/* dx = conv-transpose of dy with w (autograd of the op above w.r.t. x).
 * desc describes the FORWARD op; dx has x's shape and x_batch_stride,
 * dy has y's shape and y_batch_stride. */
size_t m355_conv3d_bwd_data_workspace(const m355_conv3d_desc* d);
int m355_conv3d_bwd_data(const m355_conv3d_desc* d, const float* dy, const float* w,
                         float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* dw[o,c,taps] = sum_{n,voxels} dy * shifted x;  dbias[o] = sum dy (NULL to skip).
 * Deterministic (fixed-order two-stage reduction). */
size_t m355_conv3d_bwd_weight_workspace(const m355_conv3d_desc* d);
int m355_conv3d_bwd_weight(const m355_conv3d_desc* d, const float* x, const float* dy,
                           float* dw, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------- conv-transpose3d
 * Replaces nn.ConvTranspose3d(kernel_size=2, stride=2) reached through the
 * upsample_class hook (models/modular_unet.py:20-21,72-81,96) and the
 * F.conv_transpose3d of BlurConvTranspose3d (components.py:152).
 * desc: N, Cin, Cout, D/H/W = INPUT spatial, k, stride, pad, out_pad.
 * Weight layout [Cin,Cout,k,k,k] (torch).  Output spatial =
 * (D-1)*stride - 2*pad + k + out_pad.
 */
size_t m355_conv_transpose3d_workspace(const m355_conv3d_desc* d);
int m355_conv_transpose3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w,
                              const float* bias, float* y,
                              void* workspace, size_t workspace_bytes, void* stream);
int m355_conv_transpose3d_bwd_data(const m355_conv3d_desc* d, const float* dy, const float* w,
                                   float* dx, void* workspace, size_t workspace_bytes,
                                   void* stream);
int m355_conv_transpose3d_bwd_weight(const m355_conv3d_desc* d, const float* x, const float* dy,
                                     float* dw, float* dbias,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* nn.ConvTranspose3d(kernel_size=2, stride=2) c8 -> c8 for the 16-bit modes (fp32 weights and arithmetic: the op
 * is HBM-bound): the output lands directly in its slot of the decoder's c8 concat buffer
 * (models/modular_unet.py:96-97).  Other geometries: M355_EUNSUPPORTED (unpack / fp32 / pack instead). */
int m355_conv_transpose3d_fwd_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                  const float* w, const float* bias, void* y16, int64_t y16_batch_stride,
                                  int32_t compute, void* stream);

/* --------------------------------------------------- normalisation (+ act)
 * Replaces normalization_class(out_channels) + activation_class() inside
 * Block3d (components.py:52-55): nn.BatchNorm3d (default) or nn.GroupNorm via
 * functools.partial.  One descriptor covers both:
 *   groups == 0 : BatchNorm — one statistic per channel over (N, voxels)
 *   groups  > 0 : GroupNorm — one statistic per (n, group) over (C/groups, voxels)
 * y = act((x - mean) * rstd * gamma[c] + beta[c]) (+ add, the residual branch).
 */
typedef struct m355_norm_desc {
  int32_t N, C;
  int64_t S;                /* voxels per channel (D*H*W) */
  int32_t groups;           /* 0 = batch norm */
  int32_t act;              /* M355_ACT_* */
  float eps;
  float act_slope;          /* leaky relu negative slope */
  int64_t x_batch_stride;   /* elements; 0 = dense */
  int64_t y_batch_stride;
  int64_t add_batch_stride; /* batch stride of the fused residual `add` (0 = dense C*S) */
} m355_norm_desc;

/* number of statistics: C for BN, N*groups for GN */
int64_t m355_norm_num_stats(const m355_norm_desc* d);
size_t m355_norm_workspace(const m355_norm_desc* d);

/* Pass 1: mean[s], rstd[s] (biased variance, 1/sqrt(var+eps)).  For BN in
 * training mode running_mean/running_var (may be NULL) are updated in place
 * with `momentum` and the UNBIASED variance, as torch does. */
int m355_norm_stats(const m355_norm_desc* d, const float* x, float* mean, float* rstd,
                    float* running_mean, float* running_var, float momentum,
                    void* workspace, size_t workspace_bytes, void* stream);
/* The same statistics from the partials of m355_conv3d_fwd_stats (x is not read).  `slots` = P of
 * the producing conv; desc->C must be that conv's Cout and desc->S its output voxel count.
 * Workspace: m355_norm_workspace(desc) bytes.  Fixed summation order (double), bit-reproducible. */
int m355_norm_stats_from_partials(const m355_norm_desc* d, const float* stat_partials, int64_t slots,
                                  float* mean, float* rstd, float* running_mean, float* running_var,
                                  float momentum, void* workspace, size_t workspace_bytes, void* stream);
/* BN eval mode: derive mean/rstd from the running statistics. */
int m355_norm_stats_from_running(const m355_norm_desc* d, const float* running_mean,
                                 const float* running_var, float* mean, float* rstd,
                                 void* stream);
/* Pass 2: normalise + affine + activation (+ add).  gamma/beta/add may be NULL. */
int m355_norm_act_fwd(const m355_norm_desc* d, const float* x, const float* mean,
                      const float* rstd, const float* gamma, const float* beta,
                      const float* add, float* y, void* stream);
/* Backward.  x is the SAVED PRE-NORM input; the activation mask and x_hat are
 * recomputed from it.  Produces dx (x's layout), dgamma[C], dbeta[C] (either may
 * be NULL when gamma is NULL).  `training` = 0 for BN eval mode (statistics are
 * constants, no mean-subtraction terms). */
/* normalise + activation with nn.AvgPool3d(2, 2) of the result as a second output (an encoder block's output
 * continues into the skip connection and, pooled, into the next level: models/modular_unet.py:90-92).  y as in
 * m355_norm_act_fwd (no residual add); pooled: fp32 [N, C, D/2, H/2, W/2] with its own batch stride (0 = dense).
 * D * H * W must equal desc->S, all even.  Bit-identical to m355_norm_act_fwd followed by m355_avgpool3d_2x_fwd. */
int m355_norm_act_pool_fwd(const m355_norm_desc* d, const float* x, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, float* y, float* pooled, int64_t pooled_batch_stride,
                           int32_t D, int32_t H, int32_t W, void* stream);
int m355_norm_act_bwd(const m355_norm_desc* d, const float* x, const float* dy,
                      const float* mean, const float* rstd, const float* gamma,
                      const float* beta, float* dx, float* dgamma, float* dbeta,
                      int training, void* workspace, size_t workspace_bytes, void* stream);
/* The same with dx ALSO written as a c8 tensor (h16 layout above) by the second pass: in the 16-bit training flow dx
 * is the output gradient of the convolution in front of the normalisation, and that convolution's data- and
 * weight-gradient kernels read it as c8 (m355_conv3d_bwd_data_h16 / m355_conv3d_bwd_weight_h16). */
int m355_norm_act_bwd_h16(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, float* dx, float* dgamma,
                          float* dbeta, int training, void* dx16, int64_t dx16_batch_stride, int32_t compute,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Synchronised batch norm: BatchNorm3d statistics over the batch of ALL ranks of a data-parallel job, so that a model
 * trained with its batch sharded over GPUs normalises exactly as the reference's single-process run of the whole batch
 * (models/nested_residual_unet.py:19-23 is BatchNorm; segmentation_trainer.py:189-262 is one process).  The library
 * provides the local halves, the caller all-reduces (SUM, RCCL) between them:
 *   forward : m355_norm_sums -> all-reduce(sums) -> m355_norm_stats_from_sums
 *   backward: m355_norm_act_bwd_reduce(total_count = &sums[2C]) -> all-reduce(stat_m) -> m355_norm_act_bwd_apply
 * sums: double[2*C + 1] = {sum x, sum x^2} per channel, then the element count (N*S) that went into them; from x or from
 * the producing conv's epilogue partials (stat_partials != NULL, x unused).  desc->groups must be 0.
 * m355_norm_act_bwd_reduce divides by *total_count (device pointer; NULL = the local count), so the SUM over ranks of
 * stat_m[C][2] is the global mean; dgamma / dbeta stay local sums (the gradient all-reduce averages them like any
 * other parameter gradient).  m355_norm_act_bwd_apply: dx (and, dx16 != NULL, its c8 twin as m355_norm_act_bwd_h16). */
int m355_norm_sums(const m355_norm_desc* d, const float* x, const float* stat_partials, int64_t slots, double* sums,
                   void* workspace, size_t workspace_bytes, void* stream);
int m355_norm_stats_from_sums(const m355_norm_desc* d, const double* sums, float* mean, float* rstd,
                              float* running_mean, float* running_var, float momentum, void* stream);
int m355_norm_act_bwd_reduce(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                             int training, const double* total_count, float* stat_m, void* workspace,
                             size_t workspace_bytes, void* stream);
int m355_norm_act_bwd_apply(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, const float* stat_m, float* dx,
                            void* dx16, int64_t dx16_batch_stride, int32_t compute, void* stream);

/* The passes that WRITE c8 tensors in the 16-bit modes (so that conv -> norm/act -> conv never converts):
 * m355_norm_act_fwd with a c8 output y16 (the layout transposer: reads the fp32 NCDHW conv output) and,
 * when y != NULL, the usual fp32 NCDHW output as well (desc->y_batch_stride) for non-conv consumers;
 * nn.AvgPool3d(2, 2) (models/modular_unet.py:22,41,64,92) c8 -> c8 (fp32 sums, one rounding). */
int m355_norm_act_fwd_h16(const m355_norm_desc* d, const float* x, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, const float* add, float* y, void* y16,
                          int64_t y16_batch_stride, int32_t compute, void* stream);
/* normalise + activation (+ residual add16) c8 -> c8, for a pre-norm tensor produced by m355_conv3d_fwd_h16_c8;
 * statistics of such a tensor when its conv had none fused (split-K plans): per-channel (sum, sum of squares)
 * partials [N][P][C][2], P = m355_act16_partials_slots(S), fed to m355_norm_stats_from_partials. */
int m355_norm_act_fwd_c8(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, const void* add16,
                         int64_t add16_batch_stride, void* y16, int64_t y16_batch_stride, int32_t compute,
                         void* stream);
int64_t m355_act16_partials_slots(int64_t S);
int m355_act16_channel_partials(const void* x16, int64_t x16_batch_stride, int32_t N, int32_t C, int64_t S,
                                int32_t compute, float* stat_partials, void* stream);
int m355_avgpool3d_2x_fwd_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                              int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream);

/* ---- the c8-only TRAINING flow of the 16-bit modes (round 3) ----
 * With autograd on, activations and activation gradients of the conv -> norm/act -> conv -> pool / conv-transpose ->
 * concat chain exist only as c8 tensors, as they already do under no_grad; these are the backward halves.  They replace
 * the reference's autograd under `torch.cuda.amp.autocast` (segmentation_pipeline/segmentation_trainer.py:203-227) for
 * Block3d (models/components.py:62-73), nn.AvgPool3d and nn.ConvTranspose3d(2, 2) (models/modular_unet.py:64,72-81,92,96).
 *
 * grad_scale / grad_unscale: the fp16 mode carries activation gradients multiplied by a power of two (at full size the
 * gradient of a mean over ~2e6 voxels is ~1e-7, below the fp16 normal range).  The scale is applied where a gradient
 * first becomes c8 (m355_act16_pack_scaled; values past the fp16 range saturate) and removed in the fp32 epilogue of
 * every PARAMETER gradient (`grad_unscale` = 1 / scale); activation gradients keep it.  bf16: 1. */
int m355_act16_pack_scaled(const float* x, void* x16, int32_t N, int32_t C, int64_t S, int64_t x_batch_stride,
                           int64_t x16_batch_stride, int32_t compute, float scale, void* stream);
int m355_act16_unpack_scaled(const void* x16, float* x, int32_t N, int32_t C, int64_t S, int64_t x16_batch_stride,
                             int64_t x_batch_stride, int32_t compute, float scale, void* stream);
/* data gradient of the 3x3x3 conv, c8 in -> c8 out (dx16 holds desc->Cin channels); workspace:
 * m355_conv3d_h16_workspace(desc, 1).  Same arithmetic as m355_conv3d_bwd_data_h16 followed by one rounding. */
int m355_conv3d_bwd_data_h16_c8(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride, const float* w,
                                void* dx16, int64_t dx16_batch_stride, void* workspace, size_t workspace_bytes,
                                void* stream);
/* weight gradient from c8 operands as m355_conv3d_bwd_weight_h16, with the bias gradient (may be NULL) reduced from
 * the c8 dy16 itself and both multiplied by grad_unscale in their fp32 epilogues. */
size_t m355_conv3d_bwd_weight_c8_workspace(const m355_conv3d_desc* d);
int m355_conv3d_bwd_weight_c8(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                              int64_t dy16_batch_stride, float* dw, float* dbias, float grad_unscale, void* workspace,
                              size_t workspace_bytes, void* stream);
/* normalisation + activation backward on c8 operands: x16 = the saved PRE-NORM tensor, the incoming gradient is
 * dy16 (may be NULL when dpool16 is given) + un-pooled dpool16 (may be NULL; then D, H, W are ignored): an encoder
 * block's output continues into the skip connection AND, through nn.AvgPool3d(2, 2), into the next level, and the sum
 * of the two gradients is formed inside this pass.  dx16: c8, same shape as x16.  dgamma / dbeta (may be NULL) fp32,
 * multiplied by grad_unscale.  Workspace: m355_norm_workspace(desc).  Expressions as m355_norm_act_bwd. */
int m355_norm_act_bwd_c8(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                         int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride, int32_t D,
                         int32_t H, int32_t W, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, void* dx16, int64_t dx16_batch_stride, float* dgamma, float* dbeta,
                         int training, float grad_unscale, int32_t compute, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The same backward as its two halves around an exchange (synchronised BatchNorm on the c8 flow -- the reference's
 * dmri_hippo model is BatchNorm, models/nested_residual_unet.py:19-23, and a batch sharded over ranks must normalise
 * with the statistics of the whole batch): _reduce = first pass + per-statistic sums divided by `total_count[0]` (a
 * device double: the element count per statistic over ALL ranks) -> stat_m[2 * num_stats] (this rank's share of the
 * two gradient means) and the local dgamma / dbeta; the host all-reduces stat_m (SUM); _apply = second pass with the
 * reduced stat_m.  With total_count = this rank's own count the pair equals m355_norm_act_bwd_c8 bit for bit. */
int m355_norm_act_bwd_c8_reduce(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                                int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride, int32_t D,
                                int32_t H, int32_t W, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, float* dgamma, float* dbeta, int training, const double* total_count,
                                float grad_unscale, float* stat_m, int32_t compute, void* workspace, size_t workspace_bytes,
                                void* stream);
int m355_norm_act_bwd_c8_apply(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                               int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride, int32_t D,
                               int32_t H, int32_t W, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* stat_m, void* dx16, int64_t dx16_batch_stride,
                               int32_t compute, void* stream);
/* nn.AvgPool3d(2, 2) backward c8 -> c8, optionally adding the gradient of the un-pooled tensor's other consumer
 * (dskip16, may be NULL): dx16[v] = dskip16[v] + dpool16[v / 2] / 8.  D, H, W: the UN-pooled size (even). */
int m355_avgpool3d_2x_bwd_h16(const void* dpool16, const void* dskip16, void* dx16, int32_t N, int32_t C, int32_t D,
                              int32_t H, int32_t W, int64_t dpool16_batch_stride, int64_t dskip16_batch_stride,
                              int64_t dx16_batch_stride, int32_t compute, void* stream);
/* nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True) c8 -> c8 and its (gather-form, deterministic)
 * backward: the `up` of NestedResUNet (segmentation_pipeline/models/nested_residual_unet.py:74,92-101) in the c8
 * flow.  D, H, W: the LOW-resolution size; fp32 interpolation with ATen's align_corners weights, one rounding. */
int m355_upsample_trilinear2x_fwd_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                      int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream);
int m355_upsample_trilinear2x_bwd_h16(const void* dy16, void* dx16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                      int64_t dy16_batch_stride, int64_t dx16_batch_stride, int32_t compute,
                                      void* stream);
/* space-to-depth / depth-to-space by 2 on c8 activations: the rearrangement around the stride-1 3x3x3 form of
 * BlurConv3d / BlurConvTranspose3d (segmentation_pipeline/models/components.py:91-154) in the 16-bit flows.  N, C, D, H,
 * W describe the FULL-resolution tensor (C channels, even sizes); the packed tensor has 8 * C channels (channel
 * c * 8 + pz * 4 + py * 2 + px = element of c8 block c) at half the resolution.  Each is the other's backward. */
int m355_space_to_depth2_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                             int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream);
int m355_depth_to_space2_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                             int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream);
/* y16[n][c][s] = x16[n][c][s] * scale[n * C + c]: nn.Dropout3d on a c8 activation (nested_residual_unet.py:43-44,
 * components.py:70-71) and its backward (the same call on the gradient); saturating for fp16. */
int m355_act16_channel_scale(const void* x16, const float* scale, void* y16, int32_t N, int32_t C, int64_t S,
                             int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream);
/* nn.ConvTranspose3d(kernel_size=2, stride=2) backward from c8 operands on the 16-bit matrix core (operands rounded to
 * the 16-bit type, fp32 accumulate): dx16 (desc->Cin channels, c8) from dy16 (desc->Cout channels at twice the
 * resolution); dw fp32 [Cin][Cout][2][2][2] and dbias (may be NULL), both multiplied by grad_unscale.
 * m355_conv_transpose3d_h16_bwd_supported(desc) == 0: unpack / m355_conv_transpose3d_bwd_* / pack instead (other
 * geometries, Cout > 128). */
int32_t m355_conv_transpose3d_h16_bwd_supported(const m355_conv3d_desc* d);
size_t m355_conv_transpose3d_h16_bwd_workspace(const m355_conv3d_desc* d);
int m355_conv_transpose3d_bwd_data_h16(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride,
                                       const float* w, void* dx16, int64_t dx16_batch_stride, int32_t compute,
                                       void* stream);
int m355_conv_transpose3d_bwd_weight_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                         const void* dy16, int64_t dy16_batch_stride, float* dw, float* dbias,
                                         float grad_unscale, int32_t compute, void* workspace, size_t workspace_bytes,
                                         void* stream);

/* --------------------------------------------------------------- pooling
 * nn.AvgPool3d(kernel_size=2, stride=2, count_include_pad=False)
 * (models/modular_unet.py:22,41,64,92).  Input D,H,W must be even.
 */
int m355_avgpool3d_2x_fwd(const float* x, float* y, int32_t N, int32_t C,
                          int32_t D, int32_t H, int32_t W,
                          int64_t x_batch_stride, int64_t y_batch_stride, void* stream);
int m355_avgpool3d_2x_bwd(const float* dy, float* dx, int32_t N, int32_t C,
                          int32_t D, int32_t H, int32_t W, /* INPUT size of fwd */
                          int64_t dy_batch_stride, int64_t dx_batch_stride, void* stream);

/* dx = add + pool-backward(dy): the encoder block's output feeds the pool AND the skip connection
 * (models/modular_unet.py:90-92), so its gradient is the sum of two; `add` (dx's shape, own batch stride) is the
 * skip gradient -- one pass instead of a pool-backward pass plus a separate elementwise add. */
int m355_avgpool3d_2x_bwd_add(const float* dy, const float* add, float* dx, int32_t N, int32_t C,
                              int32_t D, int32_t H, int32_t W, /* INPUT size of fwd */
                              int64_t dy_batch_stride, int64_t add_batch_stride, int64_t dx_batch_stride,
                              void* stream);

/* ------------------------------------------------------------- upsampling
 * nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)
 * (models/modular_unet.py:20,39,80,96).  D,H,W = input size; output = 2x.
 */
int m355_upsample_trilinear2x_fwd(const float* x, float* y, int32_t N, int32_t C,
                                  int32_t D, int32_t H, int32_t W,
                                  int64_t x_batch_stride, int64_t y_batch_stride, void* stream);
int m355_upsample_trilinear2x_bwd(const float* dy, float* dx, int32_t N, int32_t C,
                                  int32_t D, int32_t H, int32_t W,
                                  int64_t dy_batch_stride, int64_t dx_batch_stride, void* stream);

/* ---------------------------------------------------------------- softmax
 * nn.Softmax(dim=1) (models/modular_unet.py:26,46,84,100) and the softmax of
 * StochasticMatrix (components.py:170-185): x viewed as [N, C, inner, S] with
 * softmax over C; inner = 1 for the plain case, C for the stochastic matrix.
 * diag_bias is added to x[n, i, i, :] first when inner == C (0 to disable).
 */
int m355_softmax_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t inner,
                     int64_t S, float diag_bias, void* stream);
int m355_softmax_bwd(const float* y, const float* dy, float* dx, int32_t N, int32_t C,
                     int32_t inner, int64_t S, void* stream);

/* ------------------------------------------------------- hybrid dice loss
 * HybridLogisticDiceLoss.forward (criterions/hybrid_logistic_dice_loss.py:13-43).
 * prediction p and one-hot target t: dense [N, C, S].
 * out[0] = loss, out[1] = dice_loss, out[2] = logistic_loss.
 * sums (workspace kept for backward): [N*C*4] = {sum p*t, sum_p, sum_t, sum t*log p_safe}
 * where sum_p/sum_t are of squares when square_dice != 0.
 */
size_t m355_hybrid_loss_workspace(int32_t N, int32_t C, int64_t S);
int m355_hybrid_loss_fwd(const float* p, const float* t, int32_t N, int32_t C, int64_t S,
                         float dice_weight, const float* class_weights /* [C] or NULL */,
                         int32_t square_dice, float* out3, float* sums,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dp = dloss * d(loss)/dp, closed form from the saved sums. */
int m355_hybrid_loss_bwd(const float* p, const float* t, const float* sums,
                         const float* dloss /* device scalar */, int32_t N, int32_t C, int64_t S,
                         float dice_weight, const float* class_weights, int32_t square_dice,
                         float* dp, void* stream);

/* ------------------------------------------------------------ elementwise
 * torch.cat([x_up, x_skip], dim=1) (models/modular_unet.py:97) when the
 * producer could not write into the concat buffer directly: strided copy of
 * [N, C, S] (src_batch_stride) into a channel slice (dst_batch_stride).
 * Dropout3d (components.py:58-60,70-71): y = x * scale[n*C + c].
 */
int m355_copy_channels(const float* src, float* dst, int32_t N, int32_t C, int64_t S,
                       int64_t src_batch_stride, int64_t dst_batch_stride, void* stream);
int m355_channel_scale(const float* x, const float* scale, float* y, int32_t N, int32_t C,
                       int64_t S, void* stream);
/* y = a + b (dense, n elements): residual add when not fused. */
int m355_add(const float* a, const float* b, float* y, int64_t n, void* stream);

/* Space-to-depth / depth-to-space by 2 along D, H, W (each other's inverse and adjoint):
 *   s2d: y[n, c*8 + p, z, y, x] = x[n, c, 2z+pz, 2y+py, 2x+px],  p = pz*4 + py*2 + px   (input D,H,W even)
 *   d2s: the inverse; D,H,W below are always those of the FULL-resolution tensor.
 * They turn the reference's strided Blur convolutions (components.py:91-154: effective 4x4x4 kernel,
 * stride 2, padding 1) into stride-1 3x3x3 convolutions that run on the MFMA path:
 *   BlurConv3d:          conv3(s2d(x), W') ;   BlurConvTranspose3d: d2s(conv3(x, W'')). */
int m355_space_to_depth2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                         int64_t x_batch_stride, int64_t y_batch_stride, void* stream);
int m355_depth_to_space2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                         int64_t x_batch_stride, int64_t y_batch_stride, void* stream);

/* Weight transform of BlurConv3d / BlurConvTranspose3d (reference models/components.py:112-119,
 * 145-152) for 3x3x3 filters: optional per-filter standardisation (unbiased std, eps 1e-5), the
 * depthwise 2x2x2 box blur with padding 1 (scale[b] = value of the module's `kernel` buffer for
 * dim-1 channel b) giving a 4x4x4 filter, and its rearrangement into the sparse 3x3x3 filter of the
 * space-to-depth formulation, in one pass.  w: [A][B][27] (the module's weight: A = Cout, B = Cin for
 * the strided conv; A = Cin, B = Cout for the transposed conv).
 *   transposed == 0:  wexp [A][8B][27]   (conv3d weight over the space-to-depth input)
 *   transposed == 1:  wexp [8B][A][27]   (conv3d weight producing the 8 output parities)
 * mean_std: [A][2] saved statistics (written by fwd, read by bwd; may be NULL when standardize == 0).
 * bwd: dw = d(loss)/d(w) from dwexp. */
int m355_blur_weight_fwd(const float* w, const float* scale, float* wexp, float* mean_std, int32_t A, int32_t B,
                         int32_t standardize, int32_t transposed, void* stream);
int m355_blur_weight_bwd(const float* dwexp, const float* w, const float* scale, const float* mean_std, float* dw,
                         int32_t A, int32_t B, int32_t standardize, int32_t transposed, void* stream);

/* WSConv3d weight standardisation (reference models/components.py:81-88; also the optional first step of
 * the Blur convolutions, folded into m355_blur_weight_* above): per dim-0 filter a of n = Cin*k^3 entries,
 *   wn = (w - mean_a) / (std_a + 1e-5)   with torch.std's unbiased estimator.
 * mean_std [A][2] is written by fwd and read by bwd (dw from dwn). */
int m355_weight_standardize_fwd(const float* w, float* wn, float* mean_std, int32_t A, int32_t n, void* stream);
int m355_weight_standardize_bwd(const float* dwn, const float* w, const float* mean_std, float* dw, int32_t A,
                                int32_t n, void* stream);

/* Device-side weighted patch sampling.  Replaces tio.WeightedSampler(patch_size,
 * probability_map='patch_probability') of the reference's patch loader (data_loader_factory.py:36-54,
 * research/msseg2/msseg2.py:148-149; the map comes from ImageFromLabels, transforms/image_from_labels.py:11-57): the
 * patch CENTRE is drawn with probability proportional to the map (negative entries count as 0) restricted to centres
 * whose patch lies inside the volume; the returned location is the patch CORNER.  m355_sampler_build: a two-level
 * cumulative table of the map ((ceil(V / 1024) + 1) doubles, once per map); table[last] is the total weight (the caller
 * checks it is positive).  m355_sampler_draw: locations[p] = corner of the patch whose centre is the first voxel with
 * cumulative weight > u[p] * total, u in [0, 1) -- one wave per patch, one launch per batch; feed the locations to
 * m355_patch_gather.  fp64 sums in a fixed order: a draw is a pure function of (map, u). */
size_t m355_sampler_table_bytes(int32_t V0, int32_t V1, int32_t V2);
int m355_sampler_build(const float* prob, int32_t V0, int32_t V1, int32_t V2, int32_t p0, int32_t p1, int32_t p2,
                       double* table, void* stream);
int m355_sampler_draw(const float* prob, const double* table, int32_t V0, int32_t V1, int32_t V2, int32_t p0, int32_t p1,
                      int32_t p2, const double* u, int32_t P, int32_t* locations, void* stream);

/* The whole aggregation of a sliding window in one pass (torchio GridAggregator('average') as driven by
 * PatchPredict.predict, prediction.py:124-152): tiles [n0*n1*n2, C, ps0, ps1, ps2] in GridSampler order (tile index =
 * (a * n1 + b) * n2 + c over the per-axis start lists starts[0..n0), [n0..n0+n1), [n0+n1..n0+n1+n2), device int32, in
 * coordinates of the padded volume) -> out [C, V0, V1, V2] = per-voxel mean of the covering tiles, where voxel v of the
 * output is voxel v + border of the padded volume (border = 0: no padding).  Same bits as m355_patch_accumulate over
 * all tiles in order followed by m355_patch_finalize(_crop).  Every output voxel must be covered by a tile. */
int m355_patch_aggregate_grid(const float* tiles, const int32_t* starts, int32_t n0, int32_t n1, int32_t n2, float* out,
                              int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                              int32_t b0, int32_t b1, int32_t b2, void* stream);

/* ------------------------------------------------- sliding-window patches
 * PatchPredict (prediction.py:124-152) delegates tiling/aggregation to torchio
 * 0.18.45 GridSampler / GridAggregator(overlap_mode='average').  These entry
 * points are the tensor-level part: gather P patches [P, C, ps0, ps1, ps2] from
 * a volume [C, V0, V1, V2] at integer corner locations loc[P,3] (i0,j0,k0), and
 * the aggregation out[c,v] = (sum over covering patches, in patch order) /
 * (number of covering patches).
 */
int m355_patch_gather(const float* volume, const int32_t* loc, float* patches,
                      int32_t P, int32_t C, int32_t V0, int32_t V1, int32_t V2,
                      int32_t ps0, int32_t ps1, int32_t ps2, void* stream);
int m355_patch_accumulate(const float* patches, const int32_t* loc, float* accum /* [C,V] */,
                          float* count /* [V] */, int32_t P, int32_t C,
                          int32_t V0, int32_t V1, int32_t V2,
                          int32_t ps0, int32_t ps1, int32_t ps2, void* stream);
int m355_patch_finalize(const float* accum, const float* count, float* out,
                        int32_t C, int64_t V, void* stream);

/* padding_mode != None (prediction.py:114,132; research/msseg2/competition/ms-inference.py:35 uses "edge"):
 * torchio pads the volume by b = patch_overlap // 2 per side (numpy.pad) before tiling and crops the aggregate.
 * The padded volume is never materialised: loc is in PADDED coordinates, the gather maps indices back
 * (mode: 0 constant `value`, 1 edge, 2 reflect, 3 symmetric, 4 wrap -- numpy.pad's definitions), the accumulators
 * have the padded size [P0,P1,P2] = V + 2b, and the finalize writes the cropped [C, V0,V1,V2] average. */
int m355_patch_gather_padded(const float* volume, const int32_t* loc, float* patches, int32_t P, int32_t C,
                             int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                             int32_t b0, int32_t b1, int32_t b2, int32_t mode, float value, void* stream);
int m355_patch_finalize_crop(const float* accum, const float* count, float* out, int32_t C, int32_t P0, int32_t P1,
                             int32_t P2, int32_t b0, int32_t b1, int32_t b2, void* stream);

/* ----------------------------------------------------- evaluation counts
 * CustomArgMax (transforms/custom_label_transforms.py:267) + the TP/FP/FN/TN
 * sums of SegmentationEvaluator (evaluators/segmentation_evaluator.py:69-86):
 * argmax over C of prob [N,C,S] (first max wins, as torch.argmax) against an
 * integer label map [N,S]; counts[n, c, 4] = {TP, FP, FN, TN} as int64.
 * argmax_out (int32 [N,S]) may be NULL.
 */
int m355_argmax_confusion(const float* prob, const int32_t* target, int32_t* argmax_out,
                          int64_t* counts, int32_t N, int32_t C, int64_t S, void* stream);

/* ------------------------------------------------- test-time ensembles
 * EnsembleFlips / EnsembleOrientations / EnsembleModels (models/ensemble.py:38-103): a member predicts on
 * x.permute(0, 1, *perm).flip(f) and the reference maps the prediction back with .flip(f).permute(inverse) before
 * apply_strategy (:16-35) stacks all members and takes the mean, or argmax -> mode -> one_hot.
 * size3 = canonical spatial size; perm3[j] in {0,1,2} = canonical axis of member axis j; flip_mask bit j = member
 * axis j is reversed (identity: perm3 = {0,1,2}, flip_mask = 0).
 *   m355_flip_permute        member input [N,C,size3[perm]] from x [N,C,size3], one gather pass
 *   m355_ensemble_accumulate reads a member prediction THROUGH the inverse transform (it is never mapped back or
 *                            stacked): mode 0: acc[N,C,S] (+)= pred; mode 1: votes[N,C,S] (int32) += 1 at the
 *                            member's argmax class (first maximum, as torch.argmax); first != 0 initialises
 *   m355_ensemble_finalize   mode 0: mean_out = acc / members; mode 1: onehot_out[N,C,S] (int64) of the class with
 *                            the most votes, ties -> the smallest class index (torch.mode on the CPU) */
int m355_flip_permute(const float* x, float* member, int32_t N, int32_t C, const int32_t* size3, const int32_t* perm3,
                      int32_t flip_mask, void* stream);
int m355_ensemble_accumulate(const float* pred, float* acc, int32_t* votes, int32_t N, int32_t C, const int32_t* size3,
                             const int32_t* perm3, int32_t flip_mask, int32_t mode, int32_t first, void* stream);
int m355_ensemble_finalize(const float* acc, const int32_t* votes, float* mean_out, int64_t* onehot_out, int32_t N,
                           int32_t C, int64_t S, int32_t members, int32_t mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M355SEG_H */
