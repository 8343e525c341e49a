// Backward passes of the 16-bit TRAINING flow that read AND write the c8 activation layout (h16.hpp).
//
// Round 2 kept every activation and every activation gradient of a training step as an fp32 NCDHW tensor next to the
// c8 twin the convolution kernels read; the 16-bit step was then bound by those fp32 tensors (norm backward 1.7 ms,
// conversions 0.6 ms of a 10.5 ms cfg2 step).  Here the gradient of an activation exists ONLY in c8 between
// conv data-gradient -> norm/act backward -> conv data-gradient (-> pool backward), exactly as the activations do in
// the forward direction:
//   * norm_bwd_*_c8: the two passes of the normalisation + activation backward (norm.hip: norm_bwd_partial /
//     norm_bwd_apply) on c8 operands -- x16 = the saved pre-norm conv output, dy16 = the gradient of the activated
//     output -- 2 + 2 bytes per element read twice and 2 written, instead of 4 + 4 twice and 4 + 2;
//     optionally the incoming gradient is dy16 + un-pool(dpool16): an encoder block's output feeds the skip connection
//     and, through nn.AvgPool3d(2, 2), the next level (models/modular_unet.py:90-92), and the sum of the two gradients
//     is formed in the registers of this pass instead of in a tensor of its own;
//   * avgpool2_bwd_c8: the stand-alone form of that sum (blocks whose last pass is not a normalisation);
//   * pack / unpack with a power-of-two scale: the fp16 mode carries activation gradients multiplied by a loss scale
//     (fp16 has 5 exponent bits: at full size the gradient of a mean over ~2e6 voxels is ~1e-7, below the fp16
//     normal range); the scale enters where a gradient first becomes c8 and leaves in the fp32 epilogues of the
//     parameter gradients (`grad_unscale`).  bf16 keeps scale 1.
// Reference ops replaced: autograd of normalization_class + activation_class inside Block3d
// (models/components.py:52-55,62-73) and of nn.AvgPool3d (models/modular_unet.py:64,92) under
// `torch.cuda.amp.autocast` (segmentation_trainer.py:203-227).
#include "h16.hpp"

namespace m355 {

__device__ __forceinline__ float act16_grad_t(float pre, int act, float slope) {
  if (act == M355_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return pre > 0.f ? 1.f : slope;
  return 1.f;
}

// ---------------------------------------------------------------- scaled layout conversion
template <typename HT>
__global__ __launch_bounds__(256) void pack_act16_scaled_kernel(const float* __restrict__ x, HT* __restrict__ x16, int C,
                                                                int64_t S, int64_t xbs, int64_t x16bs, float scale, int* __restrict__ oflag) {
  using hx8 = typename H16<HT>::x8;
  const int cb = blockIdx.y, n = blockIdx.z;
  const float* xn = x + (int64_t)n * xbs + (int64_t)cb * 8 * S;
  hx8* dst = reinterpret_cast<hx8*>(x16 + (int64_t)n * x16bs) + (int64_t)cb * S;
  const int nc = min(8, C - cb * 8);
  bool sat = false;
  for (int64_t s = blockIdx.x * 256ll + threadIdx.x; s < S; s += gridDim.x * 256ll) {
    hx8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < nc ? to_h16_sat<HT>(xn[(int64_t)j * S + s] * scale, sat) : (HT)0.f;
    dst[s] = v;
  }
  report_saturation(sat, oflag);
}

template <typename HT>
__global__ __launch_bounds__(256) void unpack_act16_scaled_kernel(const HT* __restrict__ x16, float* __restrict__ x, int C,
                                                                  int64_t S, int64_t x16bs, int64_t xbs, float scale) {
  using hx8 = typename H16<HT>::x8;
  const int cb = blockIdx.y, n = blockIdx.z;
  float* xn = x + (int64_t)n * xbs + (int64_t)cb * 8 * S;
  const hx8* src = reinterpret_cast<const hx8*>(x16 + (int64_t)n * x16bs) + (int64_t)cb * S;
  const int nc = min(8, C - cb * 8);
  for (int64_t s = blockIdx.x * 256ll + threadIdx.x; s < S; s += gridDim.x * 256ll) {
    const hx8 v = src[s];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < nc) xn[(int64_t)j * S + s] = (float)v[j] * scale;
  }
}

// ---------------------------------------------------------------- incoming gradient of a (skip, pooled) pair
// g[v] = dy16[v] (when present) + 0.125 * dpool16[v / 2 per axis]: autograd of y -> (y, AvgPool3d(2, 2)(y)).
// A thread walks voxels v0, v0 + step, v0 + 2 step, ...: the (z, y, x) of the pooled source advance incrementally
// (one division pair at the start instead of two per item -- the pass is HBM-bound only if the ALU keeps up).
template <typename HT, bool POOL>
struct GradSrc {
  using hx8 = typename H16<HT>::x8;
  const hx8* dy;   // may be null when POOL
  const hx8* dp;   // pooled gradient (POOL only)
  int H, W, OH, OW;
  int x, y, z, dx, dyq;   // position of the current voxel; step = dyq rows + dx columns
  __device__ __forceinline__ void start(int64_t v, int64_t step) {
    if constexpr (POOL) {
      x = (int)(v % W);
      const int64_t r = v / W;
      y = (int)(r % H);
      z = (int)(r / H);
      dx = (int)(step % W);
      dyq = (int)(step / W);
    }
  }
  __device__ __forceinline__ void advance() {
    if constexpr (POOL) {
      x += dx;
      y += dyq;
      if (x >= W) { x -= W; ++y; }
      while (y >= H) { y -= H; ++z; }
    }
  }
  // the gradient at voxel v == the current position (clamped loads past the end are the caller's business)
  __device__ __forceinline__ void load(int64_t v, float (&g)[8]) const {
    if constexpr (POOL) {
      const hx8 p = dp[((int64_t)(z >> 1) * OH + (y >> 1)) * OW + (x >> 1)];
      if (dy) {
        const hx8 d = dy[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = fmaf((float)p[j], 0.125f, (float)d[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (float)p[j] * 0.125f;
      }
    } else {
      const hx8 d = dy[v];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = (float)d[j];
    }
  }
};

// ---------------------------------------------------------------- normalisation backward, pass 1
// partial[((n*C + c)*nblk + b)*2 + {0,1}] = (sum g', sum g' * xhat) over chunk b of channel (n, c), g' = g * act'(pre):
// the layout of norm_bwd_partial_kernel, so norm_bwd_reduce_kernel finalizes both.  grid (nblk, CB, N).
template <typename HT, bool POOL>
__global__ __launch_bounds__(256) void norm_bwd_partial_c8_kernel(
    const HT* __restrict__ x16, const HT* __restrict__ dy16, const HT* __restrict__ dp16, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    double* __restrict__ partial, int C, int64_t S, int groups, int act, float slope, int64_t xbs16, int64_t ybs16,
    int64_t pbs16, int H, int W, int nblk) {
  using hx8 = typename H16<HT>::x8;
  __shared__ float red[4][16];
  const int b = blockIdx.x, cb = blockIdx.y, n = blockIdx.z;
  const int c0 = cb * 8;
  float m[8], r[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(c0 + j, C - 1);
    const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
    m[j] = mean[s];
    r[j] = rstd[s];
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    sc[j] = r[j] * g;
    sh[j] = bt - m[j] * sc[j];   // same expressions as the forward pass
  }
  const hx8* xp = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16) + (int64_t)cb * S;
  GradSrc<HT, POOL> src;
  src.dy = dy16 ? reinterpret_cast<const hx8*>(dy16 + (int64_t)n * ybs16) + (int64_t)cb * S : nullptr;
  src.H = H; src.W = W; src.OH = H >> 1; src.OW = W >> 1;
  src.dp = POOL ? reinterpret_cast<const hx8*>(dp16 + (int64_t)n * pbs16) + (int64_t)cb * (S >> 3) : nullptr;
  const int64_t begin = (int64_t)b * NORM_CHUNK_C8, end = min(S, begin + NORM_CHUNK_C8);
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  constexpr int U = 4;     // items in flight per thread (2 U 16-byte loads): the chunk is 16 items per thread
  src.start(min(begin + threadIdx.x, S - 1), 256);
  for (int64_t i0 = begin + threadIdx.x; i0 < end; i0 += 256 * U) {
    hx8 xv[U];
    float g[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + 256 * u;
      const bool ok = i < end;
      if (ok) {
        xv[u] = xp[i];
        src.load(i, g[u]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[u][j] = (HT)0.f; g[u][j] = 0.f; }
      }
      src.advance();
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xf = (float)xv[u][j];
        const float xh = (xf - m[j]) * r[j];
        const float pre = fmaf(xf, sc[j], sh[j]);
        const float gg = g[u][j] * act16_grad_t(pre, act, slope);
        a1[j] += gg;
        a2[j] = fmaf(gg, xh, a2[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a1[j] = wave_sum(a1[j]);
    a2[j] = wave_sum(a2[j]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[w][j] = a1[j];
      red[w][8 + j] = a2[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x & 7, k = threadIdx.x >> 3;
    const double t = (((double)red[0][k * 8 + j] + (double)red[1][k * 8 + j]) + (double)red[2][k * 8 + j]) +
                     (double)red[3][k * 8 + j];
    const int c = c0 + j;
    if (c < C) partial[(((int64_t)n * C + c) * nblk + b) * 2 + k] = t;
  }
}

// ---------------------------------------------------------------- normalisation backward, pass 2 (c8 -> c8)
// dx = rstd * (g' * gamma - m1 - xhat * m2), written as c8 only.  grid (chunks, CB, N); U items per thread in flight.
template <typename HT, bool POOL>
__global__ __launch_bounds__(256) void norm_bwd_apply_c8c8_kernel(
    const HT* __restrict__ x16, const HT* __restrict__ dy16, const HT* __restrict__ dp16, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stat_m, HT* __restrict__ dx16, int C, int64_t S, int groups, int act, float slope,
    int64_t xbs16, int64_t ybs16, int64_t pbs16, int64_t dxbs16, int H, int W, int* __restrict__ oflag) {
  using hx8 = typename H16<HT>::x8;
  constexpr int U = 2;
  bool sat = false;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int c0 = cb * 8, nc = min(8, C - c0);
  float m[8], r[8], gm[8], sc[8], sh[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(c0 + j, C - 1);
    const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
    m[j] = mean[s];
    r[j] = rstd[s];
    gm[j] = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    sc[j] = r[j] * gm[j];
    sh[j] = bt - m[j] * sc[j];
    m1[j] = stat_m[s * 2];
    m2[j] = stat_m[s * 2 + 1];
  }
  const hx8* xp = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16) + (int64_t)cb * S;
  GradSrc<HT, POOL> src;
  src.dy = dy16 ? reinterpret_cast<const hx8*>(dy16 + (int64_t)n * ybs16) + (int64_t)cb * S : nullptr;
  src.H = H; src.W = W; src.OH = H >> 1; src.OW = W >> 1;
  src.dp = POOL ? reinterpret_cast<const hx8*>(dp16 + (int64_t)n * pbs16) + (int64_t)cb * (S >> 3) : nullptr;
  hx8* dst = reinterpret_cast<hx8*>(dx16 + (int64_t)n * dxbs16) + (int64_t)cb * S;
  const int64_t stride = gridDim.x * 256ll;
  src.start(min<int64_t>(blockIdx.x * 256ll + threadIdx.x, S - 1), stride);
  for (int64_t i0 = blockIdx.x * 256ll + threadIdx.x; i0 < S; i0 += stride * U) {
    hx8 xv[U];
    float g[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < S) {
        xv[u] = xp[i];
        src.load(i, g[u]);
      }
      src.advance();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= S) break;
      hx8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xf = (float)xv[u][j];
        const float xh = (xf - m[j]) * r[j];
        const float pre = fmaf(xf, sc[j], sh[j]);
        const float gg = g[u][j] * act16_grad_t(pre, act, slope) * gm[j];
        const float v = r[j] * (gg - m1[j] - xh * m2[j]);
        o[j] = j < nc ? to_h16_sat<HT>(v, sat) : (HT)0.f;
      }
      dst[i] = o;
    }
  }
  report_saturation(sat, oflag);
}

// ---------------------------------------------------------------- AvgPool3d(2, 2) backward (+ skip gradient), c8 -> c8
template <typename HT>
__global__ __launch_bounds__(256) void avgpool2_bwd_c8_kernel(const HT* __restrict__ dp16, const HT* __restrict__ dskip16,
                                                              HT* __restrict__ dx16, int CB, int D, int H, int W,
                                                              int64_t pbs16, int64_t sbs16, int64_t xbs16, int N,
                                                              int* __restrict__ oflag) {
  using hx8 = typename H16<HT>::x8;
  const int64_t S = (int64_t)D * H * W;
  const int64_t total = (int64_t)N * CB * S;
  bool sat = false;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t v = i % S;
    const int64_t q = i / S;
    const int cb = (int)(q % CB), n = (int)(q / CB);
    GradSrc<HT, true> src;
    src.dy = dskip16 ? reinterpret_cast<const hx8*>(dskip16 + (int64_t)n * sbs16) + (int64_t)cb * S : nullptr;
    src.dp = reinterpret_cast<const hx8*>(dp16 + (int64_t)n * pbs16) + (int64_t)cb * (S >> 3);
    src.H = H; src.W = W; src.OH = H >> 1; src.OW = W >> 1;
    src.start(v, 0);
    float g[8];
    src.load(v, g);
    hx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = to_h16_sat<HT>(g[j], sat);
    (reinterpret_cast<hx8*>(dx16 + (int64_t)n * xbs16) + (int64_t)cb * S)[v] = o;
  }
  report_saturation(sat, oflag);
}

// ---------------------------------------------------------------- bias gradient from a c8 gradient
// dbias[c] = unscale * sum_{n, slot} part[n][slot][c][0] (the per-channel sums of m355_act16_channel_partials)
__global__ __launch_bounds__(64) void dbias_from_partials_kernel(const float* __restrict__ part, float* __restrict__ dbias,
                                                                 int N, int C, int slots, float unscale,
                                                                 int* __restrict__ oflag) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double a = 0.0;
  const int64_t items = (int64_t)N * slots;
  for (int64_t i = lane; i < items; i += 64) a += (double)part[(i * C + c) * 2];
  a = wave_sum(a);
  if (lane == 0) {
    const float r = (float)(a * (double)unscale);
    dbias[c] = r;
    report_nonfinite(r, oflag);
  }
}

int launch_dbias_c8(const void* dy16, int64_t dybs16, float* dbias, int N, int C, int64_t S, int compute, float unscale,
                    void* ws, hipStream_t st) {
  const int slots = (int)m355_act16_partials_slots(S);
  float* part = (float*)ws;
  if (int rc = m355_act16_channel_partials(dy16, dybs16, N, C, S, compute, part, st)) return rc;
  hipLaunchKernelGGL(dbias_from_partials_kernel, dim3((unsigned)C), dim3(64), 0, st, part, dbias, N, C, slots, unscale,
                     overflow_flag());
  return check_launch("dbias_c8");
}

size_t dbias_c8_ws_bytes(int N, int C, int64_t S) {
  return (size_t)round_up((int64_t)N * m355_act16_partials_slots(S) * C * 2 * 4, 256);
}

}  // namespace m355

using namespace m355;

static int check_h16(const char* who, int32_t compute) {
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "%s: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16", who);
  return M355_OK;
}

extern "C" int m355_act16_pack_scaled(const float* x, void* x16, int32_t N, int32_t C, int64_t S, int64_t x_batch_stride,
                                      int64_t x16_batch_stride, int32_t compute, float scale, void* stream) {
  if (int rc = check_h16("act16_pack_scaled", compute)) return rc;
  M355_REQUIRE(x && x16, M355_EINVALID_ARG, "act16_pack_scaled: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0 && N <= 65535 && c8_blocks(C) <= 65535, M355_EINVALID_ARG, "act16_pack_scaled: bad shape");
  M355_REQUIRE(((uintptr_t)x16 & 15) == 0 && x16_batch_stride % 8 == 0, M355_EINVALID_ARG, "act16_pack_scaled: c8 tensor not 16B aligned");
  const int64_t xbs = dense_or(x_batch_stride, (int64_t)C * S), x16bs = dense_or(x16_batch_stride, c8_blocks(C) * S * 8);
  dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 256 * 4), 4096)), (unsigned)c8_blocks(C), (unsigned)N);
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(pack_act16_scaled_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, (__bf16*)x16, C, S, xbs,
                       x16bs, scale, overflow_flag());
  else
    hipLaunchKernelGGL(pack_act16_scaled_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, x, (_Float16*)x16, C, S,
                       xbs, x16bs, scale, overflow_flag());
  return check_launch("act16_pack_scaled");
}

extern "C" int m355_act16_unpack_scaled(const void* x16, float* x, int32_t N, int32_t C, int64_t S, int64_t x16_batch_stride,
                                        int64_t x_batch_stride, int32_t compute, float scale, void* stream) {
  if (int rc = check_h16("act16_unpack_scaled", compute)) return rc;
  M355_REQUIRE(x && x16, M355_EINVALID_ARG, "act16_unpack_scaled: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0 && N <= 65535 && c8_blocks(C) <= 65535, M355_EINVALID_ARG, "act16_unpack_scaled: bad shape");
  M355_REQUIRE(((uintptr_t)x16 & 15) == 0 && x16_batch_stride % 8 == 0, M355_EINVALID_ARG, "act16_unpack_scaled: c8 tensor not 16B aligned");
  const int64_t xbs = dense_or(x_batch_stride, (int64_t)C * S), x16bs = dense_or(x16_batch_stride, c8_blocks(C) * S * 8);
  dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 256 * 4), 4096)), (unsigned)c8_blocks(C), (unsigned)N);
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(unpack_act16_scaled_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x16, x, C,
                       S, x16bs, xbs, scale);
  else
    hipLaunchKernelGGL(unpack_act16_scaled_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x16, x,
                       C, S, x16bs, xbs, scale);
  return check_launch("act16_unpack_scaled");
}

// The two halves of the c8 normalisation backward.  Fused (m355_norm_act_bwd_c8): pass 1 -> reduce -> pass 2 in one call.
// Split (synchronised BatchNorm on the c8 flow, round 4): ..._c8_reduce = pass 1 + the finalize stage dividing by the
// element count over ALL ranks (stat_m = this rank's share of the two gradient means), the host all-reduces stat_m (SUM),
// ..._c8_apply = pass 2 -- as m355_norm_act_bwd_reduce / _apply do for fp32 tensors.
static int norm_bwd_c8_check(const char* who, const m355_norm_desc* d, const void* x16, const void* dy16, const void* dpool16,
                             const void* dx16_or_ws, int32_t D, int32_t H, int32_t W, int32_t compute) {
  if (int rc = check_h16(who, compute)) return rc;
  M355_REQUIRE(d && x16 && (dy16 || dpool16) && dx16_or_ws, M355_EINVALID_ARG, "%s: null pointer", who);
  M355_REQUIRE(d->N > 0 && d->C > 0 && d->S > 0 && d->N <= 65535 && c8_blocks(d->C) <= 65535, M355_EINVALID_ARG, "%s: bad shape", who);
  M355_REQUIRE(d->groups >= 0 && (d->groups == 0 || d->C % d->groups == 0), M355_EINVALID_ARG,
               "%s: C=%d not divisible by groups=%d", who, d->C, d->groups);
  M355_REQUIRE(d->act >= M355_ACT_NONE && d->act <= M355_ACT_LEAKY_RELU, M355_EINVALID_ARG, "%s: bad activation", who);
  M355_REQUIRE(!dpool16 || (D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && (int64_t)D * H * W == d->S),
               M355_EINVALID_ARG, "%s: a pooled gradient needs even D, H, W with D*H*W == S", who);
  return M355_OK;
}

struct NormBwdC8Geom {
  int64_t xbs, ybs, dxbs, pbs;
  int nblk;
  double* partial;
  float* stat_m;
};

static int norm_bwd_c8_geom(const char* who, const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                            int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride, const void* dx16,
                            int64_t dx16_batch_stride, void* workspace, NormBwdC8Geom* g) {
  const int64_t dense = c8_blocks(d->C) * d->S * 8;
  g->xbs = dense_or(x16_batch_stride, dense);
  g->ybs = dense_or(dy16_batch_stride, dense);
  g->dxbs = dense_or(dx16_batch_stride, dense);
  g->pbs = dpool16 ? dense_or(dpool16_batch_stride, c8_blocks(d->C) * (d->S / 8) * 8) : 0;   // (pooled: S / 8 voxels)
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)dy16 | (uintptr_t)dpool16 | (uintptr_t)dx16) & 15) == 0 && g->xbs % 8 == 0 &&
                   g->ybs % 8 == 0 && g->dxbs % 8 == 0 && g->pbs % 8 == 0, M355_EINVALID_ARG, "%s: c8 tensor not 16B aligned", who);
  g->nblk = (int)ceil_div(d->S, NORM_CHUNK_C8);
  g->partial = (double*)workspace;
  g->stat_m = workspace ? (float*)((char*)workspace + round_up((int64_t)d->N * d->C * g->nblk * 2 * sizeof(double), 256)) : nullptr;
  return M355_OK;
}

static void norm_bwd_c8_pass1(const m355_norm_desc* d, const NormBwdC8Geom& g, const void* x16, const void* dy16, const void* dpool16,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, int H, int W,
                              int32_t compute, hipStream_t st) {
  const dim3 g1((unsigned)g.nblk, (unsigned)c8_blocks(d->C), (unsigned)d->N);
#define M355_NB1(HT, POOL)                                                                                                 \
  hipLaunchKernelGGL((norm_bwd_partial_c8_kernel<HT, POOL>), g1, dim3(256), 0, st, (const HT*)x16, (const HT*)dy16,        \
                     (const HT*)dpool16, mean, rstd, gamma, beta, g.partial, d->C, d->S, d->groups, d->act, d->act_slope, g.xbs, \
                     g.ybs, g.pbs, H, W, g.nblk)
  if (compute == M355_COMPUTE_BF16) { if (dpool16) M355_NB1(__bf16, true); else M355_NB1(__bf16, false); }
  else { if (dpool16) M355_NB1(_Float16, true); else M355_NB1(_Float16, false); }
#undef M355_NB1
}

static void norm_bwd_c8_pass2(const m355_norm_desc* d, const NormBwdC8Geom& g, const void* x16, const void* dy16, const void* dpool16,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, const float* stat_m,
                              void* dx16, int H, int W, int32_t compute, hipStream_t st) {
  const dim3 g2((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(d->S, 256 * 2), 1024)), (unsigned)c8_blocks(d->C),
                (unsigned)d->N);
#define M355_NB2(HT, POOL)                                                                                                 \
  hipLaunchKernelGGL((norm_bwd_apply_c8c8_kernel<HT, POOL>), g2, dim3(256), 0, st, (const HT*)x16, (const HT*)dy16,        \
                     (const HT*)dpool16, mean, rstd, gamma, beta, stat_m, (HT*)dx16, d->C, d->S, d->groups, d->act,         \
                     d->act_slope, g.xbs, g.ybs, g.pbs, g.dxbs, H, W, overflow_flag())
  if (compute == M355_COMPUTE_BF16) { if (dpool16) M355_NB2(__bf16, true); else M355_NB2(__bf16, false); }
  else { if (dpool16) M355_NB2(_Float16, true); else M355_NB2(_Float16, false); }
#undef M355_NB2
}

extern "C" int m355_norm_act_bwd_c8(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                                    int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride,
                                    int32_t D, int32_t H, int32_t W, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, void* dx16, int64_t dx16_batch_stride,
                                    float* dgamma, float* dbeta, int training, float grad_unscale, int32_t compute,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = norm_bwd_c8_check("norm_act_bwd_c8", d, x16, dy16, dpool16, dx16, D, H, W, compute)) return rc;
  M355_REQUIRE(mean && rstd && workspace, M355_EINVALID_ARG, "norm_act_bwd_c8: null pointer");
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE, "norm_act_bwd_c8: workspace too small");
  NormBwdC8Geom g;
  if (int rc = norm_bwd_c8_geom("norm_act_bwd_c8", d, x16, x16_batch_stride, dy16, dy16_batch_stride, dpool16,
                                dpool16_batch_stride, dx16, dx16_batch_stride, workspace, &g))
    return rc;
  hipStream_t st = (hipStream_t)stream;
  norm_bwd_c8_pass1(d, g, x16, dy16, dpool16, mean, rstd, gamma, beta, H, W, compute, st);
  if (int rc = launch_norm_bwd_reduce(g.partial, gamma, dgamma, dbeta, g.stat_m, d->N, d->C, d->groups, d->S, training,
                                      grad_unscale, st))
    return rc;
  norm_bwd_c8_pass2(d, g, x16, dy16, dpool16, mean, rstd, gamma, beta, g.stat_m, dx16, H, W, compute, st);
  return check_launch("norm_act_bwd_c8");
}

extern "C" int m355_norm_act_bwd_c8_reduce(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                                           int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride,
                                           int32_t D, int32_t H, int32_t W, const float* mean, const float* rstd,
                                           const float* gamma, const float* beta, float* dgamma, float* dbeta, int training,
                                           const double* total_count, float grad_unscale, float* stat_m, int32_t compute,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = norm_bwd_c8_check("norm_act_bwd_c8_reduce", d, x16, dy16, dpool16, workspace, D, H, W, compute)) return rc;
  M355_REQUIRE(mean && rstd && stat_m, M355_EINVALID_ARG, "norm_act_bwd_c8_reduce: null pointer");
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE, "norm_act_bwd_c8_reduce: workspace too small");
  NormBwdC8Geom g;
  if (int rc = norm_bwd_c8_geom("norm_act_bwd_c8_reduce", d, x16, x16_batch_stride, dy16, dy16_batch_stride, dpool16,
                                dpool16_batch_stride, nullptr, 0, workspace, &g))
    return rc;
  hipStream_t st = (hipStream_t)stream;
  norm_bwd_c8_pass1(d, g, x16, dy16, dpool16, mean, rstd, gamma, beta, H, W, compute, st);
  return launch_norm_bwd_reduce(g.partial, gamma, dgamma, dbeta, stat_m, d->N, d->C, d->groups, d->S, training, grad_unscale, st,
                                total_count);
}

extern "C" int m355_norm_act_bwd_c8_apply(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride, const void* dy16,
                                          int64_t dy16_batch_stride, const void* dpool16, int64_t dpool16_batch_stride,
                                          int32_t D, int32_t H, int32_t W, const float* mean, const float* rstd,
                                          const float* gamma, const float* beta, const float* stat_m, void* dx16,
                                          int64_t dx16_batch_stride, int32_t compute, void* stream) {
  if (int rc = norm_bwd_c8_check("norm_act_bwd_c8_apply", d, x16, dy16, dpool16, dx16, D, H, W, compute)) return rc;
  M355_REQUIRE(mean && rstd && stat_m, M355_EINVALID_ARG, "norm_act_bwd_c8_apply: null pointer");
  NormBwdC8Geom g;
  if (int rc = norm_bwd_c8_geom("norm_act_bwd_c8_apply", d, x16, x16_batch_stride, dy16, dy16_batch_stride, dpool16,
                                dpool16_batch_stride, dx16, dx16_batch_stride, nullptr, &g))
    return rc;
  norm_bwd_c8_pass2(d, g, x16, dy16, dpool16, mean, rstd, gamma, beta, stat_m, dx16, H, W, compute, (hipStream_t)stream);
  return check_launch("norm_act_bwd_c8_apply");
}

extern "C" int m355_avgpool3d_2x_bwd_h16(const void* dpool16, const void* dskip16, void* dx16, int32_t N, int32_t C, int32_t D,
                                         int32_t H, int32_t W, int64_t dpool16_batch_stride, int64_t dskip16_batch_stride,
                                         int64_t dx16_batch_stride, int32_t compute, void* stream) {
  if (int rc = check_h16("avgpool3d_2x_bwd_h16", compute)) return rc;
  M355_REQUIRE(dpool16 && dx16, M355_EINVALID_ARG, "avgpool3d_2x_bwd_h16: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, M355_EINVALID_ARG, "avgpool3d_2x_bwd_h16: bad shape");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED, "avgpool3d_2x_bwd_h16: odd spatial size (%d,%d,%d)", D, H, W);
  const int CB = (int)c8_blocks(C);
  const int64_t S = (int64_t)D * H * W;
  const int64_t pbs = dense_or(dpool16_batch_stride, CB * (S / 8) * 8), sbs = dense_or(dskip16_batch_stride, CB * S * 8);
  const int64_t xbs = dense_or(dx16_batch_stride, CB * S * 8);
  M355_REQUIRE((((uintptr_t)dpool16 | (uintptr_t)dskip16 | (uintptr_t)dx16) & 15) == 0 && pbs % 8 == 0 && sbs % 8 == 0 && xbs % 8 == 0,
               M355_EINVALID_ARG, "avgpool3d_2x_bwd_h16: c8 tensor not 16B aligned");
  const int64_t total = (int64_t)N * CB * S;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 256), 16384));
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(avgpool2_bwd_c8_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dpool16,
                       (const __bf16*)dskip16, (__bf16*)dx16, CB, D, H, W, pbs, sbs, xbs, N, overflow_flag());
  else
    hipLaunchKernelGGL(avgpool2_bwd_c8_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)dpool16, (const _Float16*)dskip16, (_Float16*)dx16, CB, D, H, W, pbs, sbs, xbs, N,
                       overflow_flag());
  return check_launch("avgpool3d_2x_bwd_h16");
}
