// GroupNorm / BatchNorm3d + activation, forward and backward (HBM-bound).
//
// Reference ops replaced: normalization_class(out_channels) and
// activation_class() inside Block3d (models/components.py:52-55): nn.BatchNorm3d
// by default, nn.GroupNorm through functools.partial (the north-star config).
//
// A "statistic" s is a mean/rstd pair:
//   BN : s = channel c,          data = N runs of S contiguous floats
//   GN : s = n*groups + g,       data = 1 run of (C/groups)*S contiguous floats
// Pass 1 (stats): blocks reduce chunks to (sum, sumsq) in fp32 per thread over
// short runs, combined in double -> partial[s][blk][2]; a finalize kernel sums
// the partials in fixed order.  Pass 2 (apply): float4 streaming, one (n,c)
// channel chunk per block so gamma/beta/mean/rstd are block-uniform.
#include "common.hpp"
#include "h16.hpp"

namespace m355 {

struct NormGeom {
  int64_t nstats;  // number of statistics
  int64_t runs;    // runs per statistic
  int64_t len;     // elements per run
  int64_t count;   // runs * len
  int nblk;        // blocks per statistic
};

static NormGeom geom(const m355_norm_desc* d) {
  NormGeom g;
  if (d->groups == 0) {
    g.nstats = d->C;
    g.runs = d->N;
    g.len = d->S;
  } else {
    g.nstats = (int64_t)d->N * d->groups;
    g.runs = 1;
    g.len = (int64_t)(d->C / d->groups) * d->S;
  }
  g.count = g.runs * g.len;
  g.nblk = (int)ceil_div(g.count, NORM_CHUNK);
  return g;
}

__device__ __forceinline__ int64_t stat_base(int64_t s, int64_t run, int groups, int C, int64_t S,
                                             int64_t xbs, int64_t len) {
  if (groups == 0) return run * xbs + s * S;  // BN: s = channel, run = n
  const int64_t n = s / groups, g = s % groups;
  return n * xbs + g * len;
}

// partial[(s*nblk + b)*2 + {0,1}] = (sum, sumsq) of chunk b of statistic s
// VEC: len % 4 == 0 and 16-byte aligned runs, so a float4 never straddles a run.
template <bool VEC>
__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ x,
                                                           double* __restrict__ partial,
                                                           int groups, int C, int64_t S,
                                                           int64_t xbs, int64_t runs, int64_t len,
                                                           int nblk) {
  __shared__ double scratch[4];
  const int64_t s = blockIdx.y;
  const int b = blockIdx.x;
  const int64_t count = runs * len;
  const int64_t begin = (int64_t)b * NORM_CHUNK;
  const int64_t end = min(count, begin + NORM_CHUNK);
  constexpr int STEP = VEC ? 4 : 1;
  float s1 = 0.f, s2 = 0.f;
  // one division per thread, then incremental (run, off) bookkeeping
  int64_t i = begin + (int64_t)threadIdx.x * STEP;
  int64_t run = i / len, off = i - run * len;
  for (; i < end; i += 256 * STEP) {
    const float* p = x + stat_base(s, run, groups, C, S, xbs, len) + off;
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 = fmaf(v.x, v.x, s2); s2 = fmaf(v.y, v.y, s2);
      s2 = fmaf(v.z, v.z, s2); s2 = fmaf(v.w, v.w, s2);
    } else {
      const float v = *p;
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
    off += 256 * STEP;
    while (off >= len) { off -= len; ++run; }
  }
  const double t1 = block_sum<double, 256>((double)s1, scratch);
  const double t2 = block_sum<double, 256>((double)s2, scratch);
  if (threadIdx.x == 0) {
    partial[((int64_t)s * nblk + b) * 2 + 0] = t1;
    partial[((int64_t)s * nblk + b) * 2 + 1] = t2;
  }
}

// one wave (64 lanes) per statistic: lanes stride over the partials, fixed-order shuffle tree
__global__ __launch_bounds__(64) void norm_finalize_kernel(const double* __restrict__ partial, float* __restrict__ mean,
                                     float* __restrict__ rstd, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, float momentum, float eps,
                                     int64_t nstats, int nblk, int64_t count, const double* __restrict__ count_ptr) {
  const int64_t s = blockIdx.x;
  if (count_ptr) count = (int64_t)count_ptr[0];   // synchronised batch norm: the element count summed over the ranks
  double t1 = 0.0, t2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    t1 += partial[(s * nblk + b) * 2 + 0];
    t2 += partial[(s * nblk + b) * 2 + 1];
  }
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if (threadIdx.x != 0) return;
  const double m = t1 / (double)count;
  double var = t2 / (double)count - m * m;
  if (var < 0.0) var = 0.0;
  mean[s] = (float)m;
  rstd[s] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[s] = (1.f - momentum) * running_mean[s] + momentum * (float)m;
  if (running_var) {
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_var[s] = (1.f - momentum) * running_var[s] + momentum * (float)unbiased;
  }
}

// Synchronised batch norm, local half: sums[s][2] = (sum x, sum x^2) of statistic s in double (one wave per statistic,
// fixed order), sums[nstats*2] = the local element count.  The ranks all-reduce this buffer (SUM) and finalize from it.
__global__ __launch_bounds__(64) void norm_sums_kernel(const double* __restrict__ partial, double* __restrict__ sums,
                                                        int64_t nstats, int nblk, int64_t count) {
  const int64_t s = blockIdx.x;
  double t1 = 0.0, t2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    t1 += partial[(s * nblk + b) * 2 + 0];
    t2 += partial[(s * nblk + b) * 2 + 1];
  }
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if (threadIdx.x != 0) return;
  sums[s * 2 + 0] = t1;
  sums[s * 2 + 1] = t2;
  if (s == 0) sums[nstats * 2] = (double)count;
}

// statistics from the conv epilogue partials part[n][slot][C][2] (float), stage 1: block (b, s) sums a
// contiguous share of the (sample, slot, channel-in-group) items of statistic s in a fixed order
// (double) -> partial[(s*nblk + b)*2 + {0,1}]; norm_finalize_kernel then combines the nblk partials.
__global__ __launch_bounds__(256) void norm_from_partials_kernel(
    const float* __restrict__ part, double* __restrict__ partial, int groups, int N, int C, int64_t slots,
    int nblk) {
  __shared__ double scratch[4];
  const int64_t s = blockIdx.y;
  const int b = blockIdx.x;
  // GN: s = n*groups + g -> sample n, channels [g*cg, (g+1)*cg); BN: s = channel, all samples
  const int cg = groups == 0 ? 1 : C / groups;
  const int c_begin = groups == 0 ? (int)s : (int)(s % groups) * cg;
  const int64_t n_begin = groups == 0 ? 0 : s / groups, runs = groups == 0 ? N : 1;
  const int64_t per_n = slots * cg, items = runs * per_n;
  const int64_t share = (items + nblk - 1) / nblk;
  const int64_t begin = (int64_t)b * share, end = min(items, begin + share);
  double t1 = 0.0, t2 = 0.0;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
    const int64_t run = i / per_n, j = i - run * per_n;
    const int64_t slot = j / cg;
    const int c = c_begin + (int)(j - slot * cg);
    const float2 v = *reinterpret_cast<const float2*>(part + (((n_begin + run) * slots + slot) * C + c) * 2);
    t1 += (double)v.x;
    t2 += (double)v.y;
  }
  t1 = block_sum<double, 256>(t1, scratch);
  t2 = block_sum<double, 256>(t2, scratch);
  if (threadIdx.x == 0) {
    partial[((int64_t)s * nblk + b) * 2 + 0] = t1;
    partial[((int64_t)s * nblk + b) * 2 + 1] = t2;
  }
}

// The same for statistics with few partial items (<= 8192: every level below 64^3 of a U-Net), finalize included:
// one block per statistic sums ALL its items and writes mean / rstd (and the BatchNorm running statistics) itself,
// so the pair of latency-bound launches (~8 + ~5 us) is one.
__global__ __launch_bounds__(1024) void norm_from_partials_final_kernel(
    const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, int groups, int N, int C, int64_t slots, int64_t count) {
  __shared__ double scratch[16];   // 1024 threads: the pass is a latency chain, 8 items per thread keep it short
  const int64_t s = blockIdx.x;
  const int cg = groups == 0 ? 1 : C / groups;
  const int c_begin = groups == 0 ? (int)s : (int)(s % groups) * cg;
  const int64_t n_begin = groups == 0 ? 0 : s / groups, runs = groups == 0 ? N : 1;
  const int64_t per_n = slots * cg, items = runs * per_n;
  double t1 = 0.0, t2 = 0.0;
  for (int64_t i = threadIdx.x; i < items; i += 1024) {
    const int64_t run = i / per_n, j = i - run * per_n;
    const int64_t slot = j / cg;
    const int c = c_begin + (int)(j - slot * cg);
    const float2 v = *reinterpret_cast<const float2*>(part + (((n_begin + run) * slots + slot) * C + c) * 2);
    t1 += (double)v.x;
    t2 += (double)v.y;
  }
  t1 = block_sum<double, 1024>(t1, scratch);
  t2 = block_sum<double, 1024>(t2, scratch);
  if (threadIdx.x != 0) return;
  const double m = t1 / (double)count;
  double var = t2 / (double)count - m * m;
  if (var < 0.0) var = 0.0;
  mean[s] = (float)m;
  rstd[s] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[s] = (1.f - momentum) * running_mean[s] + momentum * (float)m;
  if (running_var) {
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_var[s] = (1.f - momentum) * running_var[s] + momentum * (float)unbiased;
  }
}

__global__ void norm_from_running_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                         float* __restrict__ mean, float* __restrict__ rstd,
                                         float eps, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  rstd[c] = 1.f / sqrtf(rv[c] + eps);
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
  if (act == M355_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return v > 0.f ? v : v * slope;
  return v;
}
__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
  if (act == M355_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return pre > 0.f ? 1.f : slope;
  return 1.f;
}

// grid: (chunks over S, C, N).  y = act((x-mean)*rstd*gamma + beta) + add
template <bool VEC>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ add,
    float* __restrict__ y, int C, int64_t S, int groups, int act, float slope, int64_t xbs,
    int64_t ybs, int64_t abs_) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
  const float m = mean[s], r = rstd[s];
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = r * g;
  const float sh = b - m * sc;
  const float* xp = x + (int64_t)n * xbs + (int64_t)c * S;
  float* yp = y + (int64_t)n * ybs + (int64_t)c * S;
  const float* ap = add ? add + (int64_t)n * abs_ + (int64_t)c * S : nullptr;
  if (VEC) {
    const int64_t S4 = S >> 2;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S4; i += gridDim.x * 256ll) {
      float4 v = reinterpret_cast<const float4*>(xp)[i];
      v.x = act_fwd(fmaf(v.x, sc, sh), act, slope);
      v.y = act_fwd(fmaf(v.y, sc, sh), act, slope);
      v.z = act_fwd(fmaf(v.z, sc, sh), act, slope);
      v.w = act_fwd(fmaf(v.w, sc, sh), act, slope);
      if (ap) {
        const float4 a = reinterpret_cast<const float4*>(ap)[i];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S; i += gridDim.x * 256ll) {
      float v = act_fwd(fmaf(xp[i], sc, sh), act, slope);
      if (ap) v += ap[i];
      yp[i] = v;
    }
  }
}

// The same pass with nn.AvgPool3d(2, 2) of the result as a SECOND output (models/modular_unet.py:90-92: an encoder
// block's output continues both into the skip connection and, pooled, into the next level): one thread per pooled
// voxel normalises + activates its 2x2x2 window (four 8-byte row pairs), stores the eight values and their mean --
// the pool's own pass over the activated tensor (a full re-read of the level) disappears.  Summation order and the
// normalise expression are those of avgpool2_fwd_kernel / norm_act_fwd_kernel: bit-identical to the two-pass result.
// grid: (chunks over pooled voxels, C, N)
__global__ __launch_bounds__(256) void norm_act_pool_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ pooled,
    int C, int D, int H, int W, int groups, int act, float slope, int64_t xbs, int64_t ybs, int64_t pbs) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
  const float m = mean[s], r = rstd[s];
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = r * g;
  const float sh = b - m * sc;
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int64_t S = (int64_t)D * H * W, OS = (int64_t)OD * OH * OW;
  const float* xp = x + (int64_t)n * xbs + (int64_t)c * S;
  float* yp = y + (int64_t)n * ybs + (int64_t)c * S;
  float* pp = pooled + (int64_t)n * pbs + (int64_t)c * OS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < OS; i += gridDim.x * 256ll) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH), oz = (int)(t / OH);
    const int64_t r00 = ((int64_t)(2 * oz) * H + 2 * oy) * W + 2 * ox;
    const int64_t off[4] = {r00, r00 + W, r00 + (int64_t)H * W, r00 + (int64_t)H * W + W};
    float2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float2*>(xp + off[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q].x = act_fwd(fmaf(v[q].x, sc, sh), act, slope);
      v[q].y = act_fwd(fmaf(v[q].y, sc, sh), act, slope);
      *reinterpret_cast<float2*>(yp + off[q]) = v[q];
    }
    pp[i] = (((((((v[0].x + v[0].y) + v[1].x) + v[1].y) + v[2].x) + v[2].y) + v[3].x) + v[3].y) * 0.125f;
  }
}

// Backward pass 1: per (n,c) partial sums A = sum g, B = sum g*xhat with
// g = dy * act'(pre).  partial[((n*C + c)*nblk + b)*2 + {0,1}]
template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    double* __restrict__ partial, int C, int64_t S, int groups, int act, float slope, int64_t xbs,
    int64_t ybs, int nblk) {
  __shared__ double scratch[4];
  const int b = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
  const float m = mean[s], r = rstd[s];
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float* xp = x + (int64_t)n * xbs + (int64_t)c * S;
  const float* dp = dy + (int64_t)n * ybs + (int64_t)c * S;
  const int64_t begin = (int64_t)b * NORM_CHUNK, end = min(S, begin + NORM_CHUNK);
  float a1 = 0.f, a2 = 0.f;
  const float sc = r * g, sh = bt - m * sc;  // same expression as the forward pass
  auto acc = [&](float xv, float dv) {
    const float xh = (xv - m) * r;
    const float pre = fmaf(xv, sc, sh);
    const float gg = dv * act_grad(pre, act, slope);
    a1 += gg;
    a2 = fmaf(gg, xh, a2);
  };
  if (VEC) {
    for (int64_t i = begin + threadIdx.x * 4; i < end; i += 1024) {
      const float4 xv = *reinterpret_cast<const float4*>(xp + i);
      const float4 dv = *reinterpret_cast<const float4*>(dp + i);
      acc(xv.x, dv.x); acc(xv.y, dv.y); acc(xv.z, dv.z); acc(xv.w, dv.w);
    }
  } else {
    for (int64_t i = begin + threadIdx.x; i < end; i += 256) acc(xp[i], dp[i]);
  }
  const double t1 = block_sum<double, 256>((double)a1, scratch);
  const double t2 = block_sum<double, 256>((double)a2, scratch);
  if (threadIdx.x == 0) {
    const int64_t o = (((int64_t)n * C + c) * nblk + b) * 2;
    partial[o] = t1;
    partial[o + 1] = t2;
  }
}

// Both finalize stages in one launch (one wave per block index s; every sum in a fixed lane-strided order + the
// fixed shuffle tree of wave_sum): for s < nstats the per-statistic means m1 = mean(dxhat), m2 = mean(dxhat * xhat)
// straight from the per-block partials, for s < C the parameter gradients dgamma[c] = sum_n B, dbeta[c] = sum_n A.
// These launches are latency-bound (~5 us each between two bandwidth-bound passes), so one instead of two is
// 108 launches less per cfg2 train step.
__global__ __launch_bounds__(64) void norm_bwd_reduce_kernel(const double* __restrict__ partial,
                                                              const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ stat_m, int N,
                                                              int C, int groups, int nblk, int64_t count, int training,
                                                              const double* __restrict__ count_ptr, float grad_unscale,
                                                              int* __restrict__ oflag) {
  // synchronised batch norm: divide by the element count over all ranks; the SUM all-reduce of stat_m is then the mean
  if (count_ptr) count = (int64_t)count_ptr[0];
  const int64_t nstats = groups == 0 ? C : (int64_t)N * groups;
  const int64_t s = blockIdx.x;
  const int lane = threadIdx.x;
  if (s < nstats) {
    double m1 = 0.0, m2 = 0.0;
    if (training) {
      if (groups == 0) {
        const int c = (int)s;
        const int64_t items = (int64_t)N * nblk;
        for (int64_t i = lane; i < items; i += 64) {
          const int64_t n = i / nblk, k = i - n * nblk;
          m1 += partial[((n * C + c) * nblk + k) * 2];
          m2 += partial[((n * C + c) * nblk + k) * 2 + 1];
        }
        const double g = gamma ? (double)gamma[c] : 1.0;
        m1 *= g;
        m2 *= g;
      } else {
        const int cpg = C / groups;
        const int64_t n = s / groups;
        const int c0 = (int)(s % groups) * cpg;
        const int64_t items = (int64_t)cpg * nblk;
        for (int64_t i = lane; i < items; i += 64) {
          const int cc = (int)(i / nblk);
          const int64_t k = i - (int64_t)cc * nblk;
          const double g = gamma ? (double)gamma[c0 + cc] : 1.0;
          m1 += g * partial[((n * C + c0 + cc) * nblk + k) * 2];
          m2 += g * partial[((n * C + c0 + cc) * nblk + k) * 2 + 1];
        }
      }
      m1 = wave_sum(m1);
      m2 = wave_sum(m2);
    }
    if (lane == 0) {
      stat_m[s * 2 + 0] = (float)(m1 / (double)count);
      stat_m[s * 2 + 1] = (float)(m2 / (double)count);
    }
  }
  if (s < C && (dgamma || dbeta)) {
    const int c = (int)s;
    double a = 0.0, bb = 0.0;
    const int64_t items = (int64_t)N * nblk;
    for (int64_t i = lane; i < items; i += 64) {
      const int64_t n = i / nblk, k = i - n * nblk;
      a += partial[((n * C + c) * nblk + k) * 2];
      bb += partial[((n * C + c) * nblk + k) * 2 + 1];
    }
    a = wave_sum(a);
    bb = wave_sum(bb);
    if (lane == 0) {
      // grad_unscale: 1 except in the fp16 training flow, whose activation gradients travel multiplied by a power of
      // two (loss scaling); the parameter gradients leave in true units
      if (dbeta) dbeta[c] = (float)(a * (double)grad_unscale);
      if (dgamma) dgamma[c] = (float)(bb * (double)grad_unscale);
      report_nonfinite((float)(a * (double)grad_unscale), oflag);
      report_nonfinite((float)(bb * (double)grad_unscale), oflag);
    }
  }
}

// Backward pass 2: dx = rstd * (g*gamma - m1 - xhat*m2)
template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stat_m, float* __restrict__ dx, int C, int64_t S, int groups, int act,
    float slope, int64_t xbs, int64_t ybs) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t s = groups == 0 ? c : (int64_t)n * groups + c / (C / groups);
  const float m = mean[s], r = rstd[s];
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float m1 = stat_m[s * 2], m2 = stat_m[s * 2 + 1];
  const float* xp = x + (int64_t)n * xbs + (int64_t)c * S;
  const float* dp = dy + (int64_t)n * ybs + (int64_t)c * S;
  float* op = dx + (int64_t)n * xbs + (int64_t)c * S;
  const float sc = r * g, sh = bt - m * sc;  // same expression as the forward pass
  auto f = [&](float xv, float dv) {
    const float xh = (xv - m) * r;
    const float pre = fmaf(xv, sc, sh);
    const float gg = dv * act_grad(pre, act, slope) * g;
    return r * (gg - m1 - xh * m2);
  };
  if (VEC) {
    const int64_t S4 = S >> 2;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S4; i += gridDim.x * 256ll) {
      const float4 xv = reinterpret_cast<const float4*>(xp)[i];
      const float4 dv = reinterpret_cast<const float4*>(dp)[i];
      float4 o;
      o.x = f(xv.x, dv.x); o.y = f(xv.y, dv.y); o.z = f(xv.z, dv.z); o.w = f(xv.w, dv.w);
      reinterpret_cast<float4*>(op)[i] = o;
    }
  } else {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S; i += gridDim.x * 256ll)
      op[i] = f(xp[i], dp[i]);
  }
}

static int validate_norm(const m355_norm_desc* d, const char* who) {
  M355_REQUIRE(d != nullptr, M355_EINVALID_ARG, "%s: null descriptor", who);
  M355_REQUIRE(d->N > 0 && d->C > 0 && d->S > 0, M355_EINVALID_ARG, "%s: non-positive size", who);
  M355_REQUIRE(d->groups >= 0 && (d->groups == 0 || d->C % d->groups == 0), M355_EINVALID_ARG,
               "%s: C=%d not divisible by groups=%d", who, d->C, d->groups);
  M355_REQUIRE(d->act >= M355_ACT_NONE && d->act <= M355_ACT_LEAKY_RELU, M355_EINVALID_ARG,
               "%s: bad activation %d", who, d->act);
  M355_REQUIRE(d->N <= 65535 && d->C <= 65535, M355_EUNSUPPORTED, "%s: N or C > 65535", who);
  return M355_OK;
}

static bool vec_ok(const m355_norm_desc* d, const void* a, const void* b, const void* c) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  const int64_t abs_ = dense_or(d->add_batch_stride, (int64_t)d->C * d->S);
  return (d->S % 4 == 0) && (xbs % 4 == 0) && (ybs % 4 == 0) && (abs_ % 4 == 0) && al(a) && al(b) && (!c || al(c));
}

}  // namespace m355

using namespace m355;

extern "C" int64_t m355_norm_num_stats(const m355_norm_desc* d) {
  if (!d) return 0;
  return d->groups == 0 ? d->C : (int64_t)d->N * d->groups;
}

extern "C" size_t m355_norm_workspace(const m355_norm_desc* d) {
  if (!d || d->N <= 0 || d->C <= 0 || d->S <= 0) return 0;
  const NormGeom g = geom(d);
  const size_t fwd = (size_t)g.nstats * g.nblk * 2 * sizeof(double);
  const int nblk_c = (int)ceil_div(d->S, NORM_CHUNK_C8);   // (the c8 backward's chunks: the larger of the two layouts)
  const size_t bwd = (size_t)round_up((int64_t)d->N * d->C * nblk_c * 2 * sizeof(double), 256) +
                     (size_t)round_up((int64_t)d->N * d->C * 2 * sizeof(double), 256) +
                     (size_t)g.nstats * 2 * sizeof(float) + 256;
  return std::max(fwd, bwd) + 256;
}

extern "C" int m355_norm_stats(const m355_norm_desc* d, const float* x, float* mean, float* rstd,
                               float* running_mean, float* running_var, float momentum,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_norm(d, "norm_stats")) return rc;
  M355_REQUIRE(x && mean && rstd && workspace, M355_EINVALID_ARG, "norm_stats: null pointer");
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE,
               "norm_stats: workspace too small");
  M355_REQUIRE(d->groups == 0 || (!running_mean && !running_var), M355_EINVALID_ARG,
               "norm_stats: running statistics are only defined for batch norm");
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d);
  M355_REQUIRE(g.nstats <= 65535, M355_EUNSUPPORTED, "norm_stats: too many statistics");
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  double* partial = (double*)workspace;
  const bool vec = (g.len % 4 == 0) && (d->S % 4 == 0) && (xbs % 4 == 0) && ((uintptr_t)x & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(norm_partial_kernel<true>, dim3((unsigned)g.nblk, (unsigned)g.nstats),
                       dim3(256), 0, st, x, partial, d->groups, d->C, d->S, xbs, g.runs, g.len,
                       g.nblk);
  else
    hipLaunchKernelGGL(norm_partial_kernel<false>, dim3((unsigned)g.nblk, (unsigned)g.nstats),
                       dim3(256), 0, st, x, partial, d->groups, d->C, d->S, xbs, g.runs, g.len,
                       g.nblk);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((unsigned)g.nstats), dim3(64), 0, st, partial, mean,
                     rstd, running_mean, running_var, momentum, d->eps, g.nstats, g.nblk, g.count, nullptr);
  return check_launch("norm_stats");
}

extern "C" int m355_norm_stats_from_partials(const m355_norm_desc* d, const float* stat_partials, int64_t slots,
                                             float* mean, float* rstd, float* running_mean,
                                             float* running_var, float momentum, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  if (int rc = validate_norm(d, "norm_stats_from_partials")) return rc;
  M355_REQUIRE(stat_partials && mean && rstd && workspace && slots > 0, M355_EINVALID_ARG,
               "norm_stats_from_partials: null pointer / no slots");
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE,
               "norm_stats_from_partials: workspace too small");
  M355_REQUIRE(d->groups == 0 || (!running_mean && !running_var), M355_EINVALID_ARG,
               "norm_stats_from_partials: running statistics are only defined for batch norm");
  const NormGeom g = geom(d);
  M355_REQUIRE(g.nstats <= 65535, M355_EUNSUPPORTED, "norm_stats_from_partials: too many statistics");
  hipStream_t st = (hipStream_t)stream;
  // blocks per statistic: ~2048 partial items each, never more than the workspace of norm_stats holds
  const int64_t items = (d->groups == 0 ? (int64_t)d->N : 1) * slots * (d->groups == 0 ? 1 : d->C / d->groups);
  if (items <= 8192) {
    hipLaunchKernelGGL(norm_from_partials_final_kernel, dim3((unsigned)g.nstats), dim3(1024), 0, st, stat_partials, mean,
                       rstd, running_mean, running_var, momentum, d->eps, d->groups, d->N, d->C, slots, g.count);
    return check_launch("norm_stats_from_partials");
  }
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(g.nblk, ceil_div(items, 2048)));
  double* partial = (double*)workspace;
  hipLaunchKernelGGL(norm_from_partials_kernel, dim3((unsigned)nblk, (unsigned)g.nstats), dim3(256), 0, st,
                     stat_partials, partial, d->groups, d->N, d->C, slots, nblk);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((unsigned)g.nstats), dim3(64), 0, st, partial, mean, rstd,
                     running_mean, running_var, momentum, d->eps, g.nstats, nblk, g.count, nullptr);
  return check_launch("norm_stats_from_partials");
}

extern "C" int m355_norm_stats_from_running(const m355_norm_desc* d, const float* running_mean,
                                            const float* running_var, float* mean, float* rstd,
                                            void* stream) {
  if (int rc = validate_norm(d, "norm_stats_from_running")) return rc;
  M355_REQUIRE(d->groups == 0, M355_EINVALID_ARG, "norm_stats_from_running: batch norm only");
  M355_REQUIRE(running_mean && running_var && mean && rstd, M355_EINVALID_ARG,
               "norm_stats_from_running: null pointer");
  hipLaunchKernelGGL(norm_from_running_kernel, dim3((unsigned)ceil_div(d->C, 64)), dim3(64), 0,
                     (hipStream_t)stream, running_mean, running_var, mean, rstd, d->eps, d->C);
  return check_launch("norm_stats_from_running");
}

extern "C" int m355_norm_act_fwd(const m355_norm_desc* d, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta,
                                 const float* add, float* y, void* stream) {
  if (int rc = validate_norm(d, "norm_act_fwd")) return rc;
  M355_REQUIRE(x && mean && rstd && y, M355_EINVALID_ARG, "norm_act_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  const int64_t abs_ = dense_or(d->add_batch_stride, (int64_t)d->C * d->S);
  const bool vec = vec_ok(d, x, y, add);
  const int64_t work = vec ? d->S / 4 : d->S;
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work, 256 * 4), 1024));
  dim3 grid(bx, (unsigned)d->C, (unsigned)d->N);
  if (vec)
    hipLaunchKernelGGL(norm_act_fwd_kernel<true>, grid, dim3(256), 0, st, x, mean, rstd, gamma,
                       beta, add, y, d->C, d->S, d->groups, d->act, d->act_slope, xbs, ybs, abs_);
  else
    hipLaunchKernelGGL(norm_act_fwd_kernel<false>, grid, dim3(256), 0, st, x, mean, rstd, gamma,
                       beta, add, y, d->C, d->S, d->groups, d->act, d->act_slope, xbs, ybs, abs_);
  return check_launch("norm_act_fwd");
}

// first half of the normalisation backward: partial sums + reduction -> stat_m[s][2] (means of dxhat and dxhat*xhat),
// dgamma, dbeta.  stat_m may be the caller's buffer (synchronised batch norm: all-reduced before the second half).
static int norm_act_bwd_reduce_impl(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                    int training, const double* count_ptr, float* stat_m, void* workspace,
                                    size_t workspace_bytes, hipStream_t st, const char* who) {
  if (int rc = validate_norm(d, who)) return rc;
  M355_REQUIRE(x && dy && mean && rstd && workspace && stat_m, M355_EINVALID_ARG, "%s: null pointer", who);
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE, "%s: workspace too small", who);
  const NormGeom g = geom(d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  const int nblk_c = (int)ceil_div(d->S, NORM_CHUNK);
  double* partial = (double*)workspace;
  const bool vec = (d->S % 4 == 0) && (xbs % 4 == 0) && (ybs % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(norm_bwd_partial_kernel<true>,
                       dim3((unsigned)nblk_c, (unsigned)d->C, (unsigned)d->N), dim3(256), 0, st, x,
                       dy, mean, rstd, gamma, beta, partial, d->C, d->S, d->groups, d->act,
                       d->act_slope, xbs, ybs, nblk_c);
  else
    hipLaunchKernelGGL(norm_bwd_partial_kernel<false>,
                       dim3((unsigned)nblk_c, (unsigned)d->C, (unsigned)d->N), dim3(256), 0, st, x,
                       dy, mean, rstd, gamma, beta, partial, d->C, d->S, d->groups, d->act,
                       d->act_slope, xbs, ybs, nblk_c);
  const int64_t nthreads = std::max<int64_t>(g.nstats, d->C);
  hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3((unsigned)nthreads), dim3(64), 0, st, partial, gamma, dgamma, dbeta,
                     stat_m, d->N, d->C, d->groups, nblk_c, g.count, training, count_ptr, 1.f, nullptr);
  return check_launch(who);
}

// the finalize stage for the c8 backward (train16.hip), whose first pass writes the same partial layout
int m355::launch_norm_bwd_reduce(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* stat_m, int N,
                           int C, int groups, int64_t S, int training, float grad_unscale, hipStream_t st,
                           const double* count_ptr) {
  const int nblk_c = (int)ceil_div(S, NORM_CHUNK_C8);
  const int64_t nstats = groups == 0 ? C : (int64_t)N * groups;
  const int64_t count = groups == 0 ? (int64_t)N * S : (int64_t)(C / groups) * S;
  hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3((unsigned)std::max<int64_t>(nstats, C)), dim3(64), 0, st, partial, gamma,
                     dgamma, dbeta, stat_m, N, C, groups, nblk_c, count, training, count_ptr, grad_unscale,
                     grad_unscale != 1.f ? overflow_flag() : nullptr);
  return check_launch("norm_bwd_reduce");
}

// second half: dx (and its c8 twin) from x, dy and the per-statistic means
static int norm_act_bwd_apply_impl(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, const float* stat_m, float* dx,
                                   hipStream_t st, void* dx16, int64_t dx16_batch_stride, int compute, const char* who) {
  if (int rc = validate_norm(d, who)) return rc;
  M355_REQUIRE(x && dy && mean && rstd && dx && stat_m, M355_EINVALID_ARG, "%s: null pointer", who);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  if (dx16)   // 16-bit training flow: dx as fp32 and as c8 in one pass
    return launch_norm_bwd_apply_c8(x, dy, mean, rstd, gamma, beta, stat_m, dx, dx16, d->N, d->C, d->S, d->groups, d->act,
                                    d->act_slope, xbs, ybs, dense_or(dx16_batch_stride, c8_blocks(d->C) * d->S * 8), compute,
                                    st);
  const bool vec = (d->S % 4 == 0) && (xbs % 4 == 0) && (ybs % 4 == 0) &&
                   (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  const int64_t work = vec ? d->S / 4 : d->S;
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work, 256 * 4), 1024));
  dim3 grid(bx, (unsigned)d->C, (unsigned)d->N);
  if (vec)
    hipLaunchKernelGGL(norm_bwd_apply_kernel<true>, grid, dim3(256), 0, st, x, dy, mean, rstd,
                       gamma, beta, stat_m, dx, d->C, d->S, d->groups, d->act, d->act_slope, xbs,
                       ybs);
  else
    hipLaunchKernelGGL(norm_bwd_apply_kernel<false>, grid, dim3(256), 0, st, x, dy, mean, rstd,
                       gamma, beta, stat_m, dx, d->C, d->S, d->groups, d->act, d->act_slope, xbs,
                       ybs);
  return check_launch(who);
}

static int norm_act_bwd_impl(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, float* dx, float* dgamma,
                             float* dbeta, int training, void* workspace, size_t workspace_bytes, void* stream,
                             void* dx16, int64_t dx16_batch_stride, int compute) {
  M355_REQUIRE(d && workspace && dx, M355_EINVALID_ARG, "norm_act_bwd: null pointer");
  const int nblk_c = (int)ceil_div(d->S, NORM_CHUNK);
  float* stat_m = (float*)((char*)workspace + round_up((int64_t)d->N * d->C * nblk_c * 2 * sizeof(double), 256));
  if (int rc = norm_act_bwd_reduce_impl(d, x, dy, mean, rstd, gamma, beta, dgamma, dbeta, training, nullptr, stat_m,
                                        workspace, workspace_bytes, (hipStream_t)stream, "norm_act_bwd"))
    return rc;
  return norm_act_bwd_apply_impl(d, x, dy, mean, rstd, gamma, beta, stat_m, dx, (hipStream_t)stream, dx16,
                                 dx16_batch_stride, compute, "norm_act_bwd");
}

extern "C" int m355_norm_act_pool_fwd(const m355_norm_desc* d, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, float* y, float* pooled,
                                      int64_t pooled_batch_stride, int32_t D, int32_t H, int32_t W, void* stream) {
  if (int rc = validate_norm(d, "norm_act_pool_fwd")) return rc;
  M355_REQUIRE(x && mean && rstd && y && pooled, M355_EINVALID_ARG, "norm_act_pool_fwd: null pointer");
  M355_REQUIRE(D > 0 && H > 0 && W > 0 && (int64_t)D * H * W == d->S, M355_EINVALID_ARG,
               "norm_act_pool_fwd: D*H*W != desc->S");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED, "norm_act_pool_fwd: odd spatial size (%d,%d,%d)",
               D, H, W);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  const int64_t pbs = dense_or(pooled_batch_stride, (int64_t)d->C * (d->S / 8));
  M355_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 7) == 0 && xbs % 2 == 0 && ybs % 2 == 0, M355_EINVALID_ARG,
               "norm_act_pool_fwd: x / y not 8B aligned");
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(d->S / 8, 256 * 2), 1024));
  dim3 grid(bx, (unsigned)d->C, (unsigned)d->N);
  hipLaunchKernelGGL(norm_act_pool_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, pooled,
                     d->C, D, H, W, d->groups, d->act, d->act_slope, xbs, ybs, pbs);
  return check_launch("norm_act_pool_fwd");
}

extern "C" int m355_norm_act_bwd(const m355_norm_desc* d, const float* x, const float* dy,
                                 const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, float* dx, float* dgamma, float* dbeta,
                                 int training, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return norm_act_bwd_impl(d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, training, workspace, workspace_bytes,
                           stream, nullptr, 0, 0);
}

extern "C" int m355_norm_act_bwd_h16(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta, float* dx,
                                     float* dgamma, float* dbeta, int training, void* dx16, int64_t dx16_batch_stride,
                                     int32_t compute, void* workspace, size_t workspace_bytes, void* stream) {
  M355_REQUIRE(dx16 && (compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16) && ((uintptr_t)dx16 & 15) == 0 &&
                   dx16_batch_stride % 8 == 0,
               M355_EINVALID_ARG, "norm_act_bwd_h16: needs an aligned c8 destination and a 16-bit compute type");
  return norm_act_bwd_impl(d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, training, workspace, workspace_bytes,
                           stream, dx16, dx16_batch_stride, compute);
}

// ---- synchronised batch norm (statistics over the batch of ALL ranks): the local halves around the all-reduce ----

extern "C" int m355_norm_sums(const m355_norm_desc* d, const float* x, const float* stat_partials, int64_t slots,
                              double* sums, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_norm(d, "norm_sums")) return rc;
  M355_REQUIRE((x || stat_partials) && sums && workspace, M355_EINVALID_ARG, "norm_sums: null pointer");
  M355_REQUIRE(!stat_partials || slots > 0, M355_EINVALID_ARG, "norm_sums: partials without slots");
  M355_REQUIRE(d->groups == 0, M355_EINVALID_ARG, "norm_sums: batch norm only (group statistics never cross samples)");
  M355_REQUIRE(workspace_bytes >= m355_norm_workspace(d), M355_EWORKSPACE, "norm_sums: workspace too small");
  const NormGeom g = geom(d);
  M355_REQUIRE(g.nstats <= 65535, M355_EUNSUPPORTED, "norm_sums: too many statistics");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace;
  int nblk = g.nblk;
  if (stat_partials) {   // the producing conv's epilogue partials [N][slots][C][2]
    const int64_t items = (int64_t)d->N * slots;
    nblk = (int)std::max<int64_t>(1, std::min<int64_t>(g.nblk, ceil_div(items, 2048)));
    hipLaunchKernelGGL(norm_from_partials_kernel, dim3((unsigned)nblk, (unsigned)g.nstats), dim3(256), 0, st,
                       stat_partials, partial, d->groups, d->N, d->C, slots, nblk);
  } else {
    const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
    const bool vec = (g.len % 4 == 0) && (d->S % 4 == 0) && (xbs % 4 == 0) && ((uintptr_t)x & 15) == 0;
    if (vec)
      hipLaunchKernelGGL(norm_partial_kernel<true>, dim3((unsigned)g.nblk, (unsigned)g.nstats), dim3(256), 0, st, x,
                         partial, d->groups, d->C, d->S, xbs, g.runs, g.len, g.nblk);
    else
      hipLaunchKernelGGL(norm_partial_kernel<false>, dim3((unsigned)g.nblk, (unsigned)g.nstats), dim3(256), 0, st, x,
                         partial, d->groups, d->C, d->S, xbs, g.runs, g.len, g.nblk);
  }
  hipLaunchKernelGGL(norm_sums_kernel, dim3((unsigned)g.nstats), dim3(64), 0, st, partial, sums, g.nstats, nblk, g.count);
  return check_launch("norm_sums");
}

extern "C" int m355_norm_stats_from_sums(const m355_norm_desc* d, const double* sums, float* mean, float* rstd,
                                         float* running_mean, float* running_var, float momentum, void* stream) {
  if (int rc = validate_norm(d, "norm_stats_from_sums")) return rc;
  M355_REQUIRE(sums && mean && rstd, M355_EINVALID_ARG, "norm_stats_from_sums: null pointer");
  M355_REQUIRE(d->groups == 0, M355_EINVALID_ARG, "norm_stats_from_sums: batch norm only");
  const NormGeom g = geom(d);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((unsigned)g.nstats), dim3(64), 0, (hipStream_t)stream, sums, mean, rstd,
                     running_mean, running_var, momentum, d->eps, g.nstats, 1, g.count, sums + g.nstats * 2);
  return check_launch("norm_stats_from_sums");
}

extern "C" int m355_norm_act_bwd_reduce(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta, float* dgamma,
                                        float* dbeta, int training, const double* total_count, float* stat_m,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  return norm_act_bwd_reduce_impl(d, x, dy, mean, rstd, gamma, beta, dgamma, dbeta, training, total_count, stat_m,
                                  workspace, workspace_bytes, (hipStream_t)stream, "norm_act_bwd_reduce");
}

extern "C" int m355_norm_act_bwd_apply(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, const float* stat_m,
                                       float* dx, void* dx16, int64_t dx16_batch_stride, int32_t compute, void* stream) {
  M355_REQUIRE(!dx16 || ((compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16) && ((uintptr_t)dx16 & 15) == 0 &&
                         dx16_batch_stride % 8 == 0),
               M355_EINVALID_ARG, "norm_act_bwd_apply: the c8 destination needs alignment and a 16-bit compute type");
  return norm_act_bwd_apply_impl(d, x, dy, mean, rstd, gamma, beta, stat_m, dx, (hipStream_t)stream, dx16,
                                 dx16_batch_stride, compute, "norm_act_bwd_apply");
}
