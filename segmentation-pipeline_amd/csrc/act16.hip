// HBM-bound passes of the 16-bit compute modes that WRITE the c8 activation layout (h16.hpp) the
// convolution kernels read: the normalise + activation pass after a convolution is the layout
// transposer (it already owns one full read + write of the tensor), the 2x2x2 average pool works c8 -> c8.
//
// Reference ops replaced: normalization_class + activation_class inside Block3d
// (models/components.py:52-55, + the residual sum :67-68) and nn.AvgPool3d(2, 2)
// (models/modular_unet.py:22,41,64,92), for the data flow conv -> norm/act -> conv under BASELINE cfg3 / cfg5.
#include "h16.hpp"

namespace m355 {

__device__ __forceinline__ float act16_fwd(float v, int act, float slope) {
  if (act == M355_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return v > 0.f ? v : v * slope;
  return v;
}

// grid: (chunks over S, channel blocks, N).  x: fp32 NCDHW conv output; y16: c8; y32 (may be null): fp32
// NCDHW copy for consumers that are not convolutions.
template <typename HT, bool VEC>
__global__ __launch_bounds__(256) void norm_act_fwd_c8_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ add,
    float* __restrict__ y32, HT* __restrict__ y16, int C, int64_t S, int groups, int act, float slope, int64_t xbs,
    int64_t y32bs, int64_t abs_, int64_t y16bs) {
  using hx8 = typename H16<HT>::x8;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int c0 = cb * 8, nc = min(8, C - c0);
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(c0 + j, C - 1);
    const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
    const float m = mean[s], r = rstd[s];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    sc[j] = r * g;
    sh[j] = b - m * sc[j];  // same expression as norm_act_fwd_kernel
  }
  const float* xp = x + (int64_t)n * xbs + (int64_t)c0 * S;
  const float* ap = add ? add + (int64_t)n * abs_ + (int64_t)c0 * S : nullptr;
  float* yp = y32 ? y32 + (int64_t)n * y32bs + (int64_t)c0 * S : nullptr;
  hx8* dst = reinterpret_cast<hx8*>(y16 + (int64_t)n * y16bs) + (int64_t)cb * S;
  // Each lane reads one voxel of each of the 8 planes (a wave: 256 contiguous bytes per plane) and writes one
  // 16-byte item (a wave: 1 KiB contiguous).  U voxels per thread, a whole grid stride apart, are in flight
  // together (8*U independent loads).  (Four ADJACENT voxels per thread -- 16-byte plane loads -- was tried:
  // its item stores are 64 bytes apart across lanes and the pass ran 1.9x slower.)
  constexpr int U = VEC ? 4 : 1;
  const int64_t stride = gridDim.x * 256ll;
  for (int64_t i0 = blockIdx.x * 256ll + threadIdx.x; i0 < S; i0 += stride * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = min(i0 + u * stride, S - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[u][j] = j < nc ? xp[(int64_t)j * S + i] : 0.f;
    }
    if (ap) {
      float a[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = min(i0 + u * stride, S - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[u][j] = j < nc ? ap[(int64_t)j * S + i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = act16_fwd(fmaf(v[u][j], sc[j], sh[j]), act, slope) + a[u][j];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = act16_fwd(fmaf(v[u][j], sc[j], sh[j]), act, slope);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= S) break;
      hx8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (HT)(j < nc ? v[u][j] : 0.f);
      dst[i] = o;
      if (yp) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nc) yp[(int64_t)j * S + i] = v[u][j];
      }
    }
  }
}

// Backward twin of the pass above (16-bit TRAINING flow): the second pass of the normalisation backward
// (norm_bwd_apply_kernel in norm.hip: dx = rstd * (g * gamma - m1 - xhat * m2), g = dy * act'(pre)) over 8 channel
// planes per thread, writing dx as fp32 NCDHW (same expressions, bit-identical to the fp32 pass) AND as c8 items --
// dx is the output gradient of the convolution in front of the norm, whose data- and weight-gradient kernels read
// it as c8; emitting the twin here replaces a separate conversion pass (read 4 + write 2 bytes per element) by
// 2 bytes per element written.
__device__ __forceinline__ float act16_grad(float pre, int act, float slope) {
  if (act == M355_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return pre > 0.f ? 1.f : slope;
  return 1.f;
}

template <typename HT>
__global__ __launch_bounds__(256) void norm_bwd_apply_c8_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stat_m, float* __restrict__ dx, HT* __restrict__ dx16, int C, int64_t S, int groups,
    int act, float slope, int64_t xbs, int64_t ybs, int64_t dx16bs) {
  using hx8 = typename H16<HT>::x8;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int c0 = cb * 8, nc = min(8, C - c0);
  float m[8], r[8], g[8], sc[8], sh[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(c0 + j, C - 1);
    const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
    m[j] = mean[s];
    r[j] = rstd[s];
    g[j] = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    sc[j] = r[j] * g[j];
    sh[j] = bt - m[j] * sc[j];  // same expressions as norm_bwd_apply_kernel
    m1[j] = stat_m[s * 2];
    m2[j] = stat_m[s * 2 + 1];
  }
  const float* xp = x + (int64_t)n * xbs + (int64_t)c0 * S;
  const float* dp = dy + (int64_t)n * ybs + (int64_t)c0 * S;
  float* op = dx + (int64_t)n * xbs + (int64_t)c0 * S;
  hx8* dst = reinterpret_cast<hx8*>(dx16 + (int64_t)n * dx16bs) + (int64_t)cb * S;
  constexpr int U = 2;
  const int64_t stride = gridDim.x * 256ll;
  for (int64_t i0 = blockIdx.x * 256ll + threadIdx.x; i0 < S; i0 += stride * U) {
    float xv[U][8], dv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = min(i0 + u * stride, S - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xv[u][j] = j < nc ? xp[(int64_t)j * S + i] : 0.f;
        dv[u][j] = j < nc ? dp[(int64_t)j * S + i] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= S) break;
      hx8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[u][j] - m[j]) * r[j];
        const float pre = fmaf(xv[u][j], sc[j], sh[j]);
        const float gg = dv[u][j] * act16_grad(pre, act, slope) * g[j];
        const float v = r[j] * (gg - m1[j] - xh * m2[j]);
        o[j] = (HT)(j < nc ? v : 0.f);
        if (j < nc) op[(int64_t)j * S + i] = v;
      }
      dst[i] = o;
    }
  }
}

int launch_norm_bwd_apply_c8(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, const float* stat_m, float* dx, void* dx16, int N, int C, int64_t S,
                             int groups, int act, float slope, int64_t xbs, int64_t ybs, int64_t dx16bs, int compute,
                             hipStream_t st) {
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 256 * 2), 1024));
  dim3 grid(bx, (unsigned)c8_blocks(C), (unsigned)N);
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(norm_bwd_apply_c8_kernel<__bf16>, grid, dim3(256), 0, st, x, dy, mean, rstd, gamma, beta, stat_m, dx,
                       (__bf16*)dx16, C, S, groups, act, slope, xbs, ybs, dx16bs);
  else
    hipLaunchKernelGGL(norm_bwd_apply_c8_kernel<_Float16>, grid, dim3(256), 0, st, x, dy, mean, rstd, gamma, beta, stat_m,
                       dx, (_Float16*)dx16, C, S, groups, act, slope, xbs, ybs, dx16bs);
  return check_launch("norm_bwd_apply_c8");
}

// c8 -> c8 average pool: one thread per output voxel and channel block; the 2x2x2 window is four 32-byte
// runs (two x-adjacent items each).  Sums in fp32 in torch's (z, y, x) order, one rounding at the end.
template <typename HT>
__global__ __launch_bounds__(256) void avgpool2_c8_kernel(const HT* __restrict__ x16, HT* __restrict__ y16, int CB,
                                                          int D, int H, int W, int64_t xbs, int64_t ybs, int N) {
  using hx8 = typename H16<HT>::x8;
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int64_t OS = (int64_t)OD * OH * OW, S = (int64_t)D * H * W;
  const int64_t total = (int64_t)N * CB * OS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t ov = i % OS;
    const int64_t r = i / OS;
    const int cb = (int)(r % CB), n = (int)(r / CB);
    const int ox = (int)(ov % OW), oy = (int)((ov / OW) % OH), oz = (int)(ov / ((int64_t)OW * OH));
    const hx8* src = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs) + (int64_t)cb * S;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const hx8 v = src[((int64_t)(2 * oz + (q >> 2)) * H + 2 * oy + ((q >> 1) & 1)) * W + 2 * ox + (q & 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = q == 0 ? (float)v[j] : acc[j] + (float)v[j];
    }
    hx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (HT)(acc[j] * 0.125f);
    (reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs) + (int64_t)cb * OS)[ov] = o;
  }
}

// c8 -> c8 normalise + activation (+ residual, also c8): the pre-norm tensor was written by the conv epilogue as
// c8 (m355_conv3d_fwd_h16_c8), so the pass is purely elementwise on 16-byte items: 2 + 2 bytes per element
// instead of 4 + 4 for the fp32 pass.  U items per thread, a grid stride apart, in flight together.
template <typename HT>
__global__ __launch_bounds__(256) void norm_act_c8c8_kernel(
    const HT* __restrict__ x16, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const HT* __restrict__ add16,
    HT* __restrict__ y16, int C, int64_t S, int groups, int act, float slope, int64_t xbs16, int64_t abs16,
    int64_t ybs16) {
  using hx8 = typename H16<HT>::x8;
  constexpr int U = 4;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int c0 = cb * 8, nc = min(8, C - c0);
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(c0 + j, C - 1);
    const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
    const float m = mean[s], r = rstd[s];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    sc[j] = r * g;
    sh[j] = b - m * sc[j];
  }
  const hx8* src = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16) + (int64_t)cb * S;
  const hx8* asrc = add16 ? reinterpret_cast<const hx8*>(add16 + (int64_t)n * abs16) + (int64_t)cb * S : nullptr;
  hx8* dst = reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs16) + (int64_t)cb * S;
  const int64_t stride = gridDim.x * 256ll;
  for (int64_t i0 = blockIdx.x * 256ll + threadIdx.x; i0 < S; i0 += stride * U) {
    hx8 v[U], a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[min(i0 + u * stride, S - 1)];
    if (asrc) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = asrc[min(i0 + u * stride, S - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= S) break;
      hx8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = act16_fwd(fmaf((float)v[u][j], sc[j], sh[j]), act, slope);
        if (asrc) t += (float)a[u][j];
        o[j] = (HT)(j < nc ? t : 0.f);
      }
      dst[i] = o;
    }
  }
}

// (sum, sum of squares) per channel of a c8 tensor, in the format of the conv epilogue partials
// part[n][slot][C][2] (consumed by m355_norm_stats_from_partials): the statistics of a pre-norm tensor whose
// producing conv had no fused statistics (split-K plans on the small levels).  grid (slots, CB, N).
template <typename HT>
__global__ __launch_bounds__(256) void act16_channel_partials_kernel(const HT* __restrict__ x16,
                                                                     float* __restrict__ part, int C, int64_t S,
                                                                     int64_t xbs16, int slots) {
  using hx8 = typename H16<HT>::x8;
  __shared__ float red[4][16];
  const int slot = blockIdx.x, cb = blockIdx.y, n = blockIdx.z;
  const int64_t chunk = (S + slots - 1) / slots;
  const int64_t begin = (int64_t)slot * chunk, end = min(S, begin + chunk);
  const hx8* src = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16) + (int64_t)cb * S;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
    const hx8 v = src[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = (float)v[j];
      s1[j] += t;
      s2[j] = fmaf(t, t, s2[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = wave_sum(s1[j]);
    s2[j] = wave_sum(s2[j]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[w][j] = s1[j];
      red[w][8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x & 7, k = threadIdx.x >> 3;
    const float t = ((red[0][k * 8 + j] + red[1][k * 8 + j]) + red[2][k * 8 + j]) + red[3][k * 8 + j];
    const int c = cb * 8 + j;
    if (c < C) part[(((int64_t)n * slots + slot) * C + c) * 2 + k] = t;
  }
}

}  // namespace m355

using namespace m355;

extern "C" int m355_norm_act_fwd_h16(const m355_norm_desc* d, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, const float* add, float* y,
                                     void* y16, int64_t y16_batch_stride, int32_t compute, void* stream) {
  M355_REQUIRE(d && x && mean && rstd && y16, M355_EINVALID_ARG, "norm_act_fwd_h16: null pointer");
  M355_REQUIRE(d->N > 0 && d->C > 0 && d->S > 0 && d->N <= 65535 && c8_blocks(d->C) <= 65535, M355_EINVALID_ARG,
               "norm_act_fwd_h16: bad shape");
  M355_REQUIRE(d->groups >= 0 && (d->groups == 0 || d->C % d->groups == 0), M355_EINVALID_ARG,
               "norm_act_fwd_h16: C=%d not divisible by groups=%d", d->C, d->groups);
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "norm_act_fwd_h16: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  const int64_t CS = (int64_t)d->C * d->S;
  const int64_t xbs = dense_or(d->x_batch_stride, CS), ybs = dense_or(d->y_batch_stride, CS);
  const int64_t abs_ = dense_or(d->add_batch_stride, CS);
  const int64_t y16bs = dense_or(y16_batch_stride, c8_blocks(d->C) * d->S * 8);
  M355_REQUIRE(((uintptr_t)y16 & 15) == 0 && y16bs % 8 == 0, M355_EINVALID_ARG, "norm_act_fwd_h16: c8 tensor not 16B aligned");
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  (void)al16;
  const bool vec = d->S >= 4096;   // large tensors: four voxels per thread in flight
  dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(d->S, 256 * (vec ? 4 : 1)), 2048)),
            (unsigned)c8_blocks(d->C), (unsigned)d->N);
#define M355_NA16(HT, V)                                                                                             \
  hipLaunchKernelGGL((norm_act_fwd_c8_kernel<HT, V>), grid, dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, \
                     beta, add, y, (HT*)y16, d->C, d->S, d->groups, d->act, d->act_slope, xbs, ybs, abs_, y16bs)
  if (compute == M355_COMPUTE_BF16) {
    if (vec) M355_NA16(__bf16, true); else M355_NA16(__bf16, false);
  } else {
    if (vec) M355_NA16(_Float16, true); else M355_NA16(_Float16, false);
  }
#undef M355_NA16
  return check_launch("norm_act_fwd_h16");
}

extern "C" int m355_avgpool3d_2x_fwd_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H,
                                         int32_t W, int64_t x16_batch_stride, int64_t y16_batch_stride,
                                         int32_t compute, void* stream) {
  M355_REQUIRE(x16 && y16, M355_EINVALID_ARG, "avgpool3d_2x_fwd_h16: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, M355_EINVALID_ARG, "avgpool3d_2x_fwd_h16: bad shape");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED,
               "avgpool3d_2x_fwd_h16: odd spatial size (%d,%d,%d)", D, H, W);
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "avgpool3d_2x_fwd_h16: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  const int CB = (int)c8_blocks(C);
  const int64_t S = (int64_t)D * H * W;
  const int64_t xbs = dense_or(x16_batch_stride, CB * S * 8), ybs = dense_or(y16_batch_stride, CB * (S / 8) * 8);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)y16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "avgpool3d_2x_fwd_h16: c8 tensor not 16B aligned");
  const int64_t total = (int64_t)N * CB * (S / 8);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 256), 8192));
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(avgpool2_c8_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x16,
                       (__bf16*)y16, CB, D, H, W, xbs, ybs, N);
  else
    hipLaunchKernelGGL(avgpool2_c8_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x16, (_Float16*)y16, CB, D, H, W, xbs, ybs, N);
  return check_launch("avgpool3d_2x_fwd_h16");
}

static int check_c8_args(const char* who, const m355_norm_desc* d, int32_t compute) {
  M355_REQUIRE(d, M355_EINVALID_ARG, "%s: null descriptor", who);
  M355_REQUIRE(d->N > 0 && d->C > 0 && d->S > 0 && d->N <= 65535 && c8_blocks(d->C) <= 65535, M355_EINVALID_ARG,
               "%s: bad shape", who);
  M355_REQUIRE(d->groups >= 0 && (d->groups == 0 || d->C % d->groups == 0), M355_EINVALID_ARG,
               "%s: C=%d not divisible by groups=%d", who, d->C, d->groups);
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "%s: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16", who);
  return M355_OK;
}

extern "C" int m355_norm_act_fwd_c8(const m355_norm_desc* d, const void* x16, int64_t x16_batch_stride,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    const void* add16, int64_t add16_batch_stride, void* y16,
                                    int64_t y16_batch_stride, int32_t compute, void* stream) {
  if (int rc = check_c8_args("norm_act_fwd_c8", d, compute)) return rc;
  M355_REQUIRE(x16 && mean && rstd && y16, M355_EINVALID_ARG, "norm_act_fwd_c8: null pointer");
  const int64_t dense = c8_blocks(d->C) * d->S * 8;
  const int64_t xbs = dense_or(x16_batch_stride, dense), abs_ = dense_or(add16_batch_stride, dense);
  const int64_t ybs = dense_or(y16_batch_stride, dense);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)y16 | (uintptr_t)add16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0 &&
                   abs_ % 8 == 0, M355_EINVALID_ARG, "norm_act_fwd_c8: c8 tensor not 16B aligned");
  dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(d->S, 256 * 4), 2048)), (unsigned)c8_blocks(d->C),
            (unsigned)d->N);
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(norm_act_c8c8_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x16, mean,
                       rstd, gamma, beta, (const __bf16*)add16, (__bf16*)y16, d->C, d->S, d->groups, d->act,
                       d->act_slope, xbs, abs_, ybs);
  else
    hipLaunchKernelGGL(norm_act_c8c8_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x16,
                       mean, rstd, gamma, beta, (const _Float16*)add16, (_Float16*)y16, d->C, d->S, d->groups, d->act,
                       d->act_slope, xbs, abs_, ybs);
  return check_launch("norm_act_fwd_c8");
}

extern "C" int64_t m355_act16_partials_slots(int64_t S) {
  return S <= 0 ? 0 : std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 4096), 1024));
}

extern "C" int m355_act16_channel_partials(const void* x16, int64_t x16_batch_stride, int32_t N, int32_t C, int64_t S,
                                           int32_t compute, float* stat_partials, void* stream) {
  M355_REQUIRE(x16 && stat_partials, M355_EINVALID_ARG, "act16_channel_partials: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0 && N <= 65535 && c8_blocks(C) <= 65535, M355_EINVALID_ARG,
               "act16_channel_partials: bad shape");
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "act16_channel_partials: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  const int64_t xbs = dense_or(x16_batch_stride, c8_blocks(C) * S * 8);
  M355_REQUIRE(((uintptr_t)x16 & 15) == 0 && xbs % 8 == 0, M355_EINVALID_ARG, "act16_channel_partials: c8 tensor not 16B aligned");
  const int slots = (int)m355_act16_partials_slots(S);
  dim3 grid((unsigned)slots, (unsigned)c8_blocks(C), (unsigned)N);
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(act16_channel_partials_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x16, stat_partials, C, S, xbs, slots);
  else
    hipLaunchKernelGGL(act16_channel_partials_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x16, stat_partials, C, S, xbs, slots);
  return check_launch("act16_channel_partials");
}
