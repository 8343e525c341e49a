// Hybrid logistic + Dice loss (fwd/bwd), sliding-window patch gather /
// aggregation, and argmax + confusion counts.  All HBM-bound single-pass kernels.
//
// Reference code replaced:
//   HybridLogisticDiceLoss.forward  criterions/hybrid_logistic_dice_loss.py:13-43
//   PatchPredict tiling/aggregation prediction.py:132-143 (torchio GridSampler /
//     GridAggregator(overlap_mode='average'))
//   CustomArgMax + SegmentationEvaluator counts
//     transforms/custom_label_transforms.py:267, evaluators/segmentation_evaluator.py:69-86
#include "common.hpp"

namespace m355 {

constexpr int LOSS_CHUNK = 16384;

// partial[((n*C + c)*nblk + b)*4 + k], k: 0 = sum p*t, 1 = sum p(^2), 2 = sum t(^2),
// 3 = sum t*log(p_safe)
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ p,
                                                           const float* __restrict__ t,
                                                           double* __restrict__ partial, int64_t S,
                                                           int square_dice, int nblk) {
  __shared__ double scratch[4];
  const int b = blockIdx.x;
  const int64_t nc = blockIdx.y;
  const float* pp = p + nc * S;
  const float* tp = t + nc * S;
  const int64_t begin = (int64_t)b * LOSS_CHUNK, end = min(S, begin + LOSS_CHUNK);
  // reference: eps = 1e-8; prediction_safe = (prediction + eps) / (1 + eps), evaluated by
  // torch in fp32 with the python scalars rounded to fp32 (1 + 1e-8 == 1.0f)
  const float eps = 1e-8f;
  const float denom = (float)(1.0 + 1e-8);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
    const float pv = pp[i], tv = tp[i];
    s0 = fmaf(pv, tv, s0);
    if (square_dice) {
      s1 = fmaf(pv, pv, s1);
      s2 = fmaf(tv, tv, s2);
    } else {
      s1 += pv;
      s2 += tv;
    }
    const float ps = (pv + eps) / denom;
    s3 = fmaf(tv, logf(ps), s3);
  }
  const double t0 = block_sum<double, 256>((double)s0, scratch);
  const double t1 = block_sum<double, 256>((double)s1, scratch);
  const double t2 = block_sum<double, 256>((double)s2, scratch);
  const double t3 = block_sum<double, 256>((double)s3, scratch);
  if (threadIdx.x == 0) {
    double* o = partial + (nc * nblk + b) * 4;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
  }
}

// single block: sums[nc*4+k], out3 = {loss, dice_loss, logistic_loss}
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* __restrict__ partial,
                                                            const float* __restrict__ class_w,
                                                            float* __restrict__ sums,
                                                            float* __restrict__ out3, int N, int C,
                                                            int64_t S, int nblk, float dice_weight) {
  __shared__ double scratch[4];
  const int NC = N * C;
  double dice_acc = 0.0, log_acc = 0.0;
  for (int nc = threadIdx.x; nc < NC; nc += 256) {
    double v[4] = {0, 0, 0, 0};
    for (int b = 0; b < nblk; ++b)
      for (int k = 0; k < 4; ++k) v[k] += partial[((int64_t)nc * nblk + b) * 4 + k];
    for (int k = 0; k < 4; ++k) sums[nc * 4 + k] = (float)v[k];
    // dice_coeffs = 2*overlap / (total + eps)   (fp32 in the reference)
    const float overlap = (float)v[0], total = (float)v[1] + (float)v[2];
    const float dice = 2.f * overlap / (total + 1e-8f);
    dice_acc += (double)(1.f - dice);
    float logistic = (float)(v[3] / (double)S);  // torch.mean over spatial dims
    if (class_w) logistic *= class_w[nc % C];
    log_acc += (double)(-logistic);
  }
  const double d = block_sum<double, 256>(dice_acc, scratch);
  const double l = block_sum<double, 256>(log_acc, scratch);
  if (threadIdx.x == 0) {
    const float dice_loss = (float)(d / NC), logistic_loss = (float)(l / NC);
    out3[0] = (1.f - dice_weight) * logistic_loss + dice_weight * dice_loss;
    out3[1] = dice_loss;
    out3[2] = logistic_loss;
  }
}

// dp = dloss * [ (1-w) * d(logistic_loss)/dp + w * d(dice_loss)/dp ]
//   d logistic_loss / dp = -cw[c] * t / ((p + eps) ) / (N*C*S)          (denominator 1+eps == 1)
//   d dice_loss / dp     = -(1/(N*C)) * (2 t / (T+eps) - 2 O * dT/dp / (T+eps)^2)
//                          dT/dp = 2p (square_dice) or 1
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ p,
                                                       const float* __restrict__ t,
                                                       const float* __restrict__ sums,
                                                       const float* __restrict__ dloss,
                                                       const float* __restrict__ class_w,
                                                       float* __restrict__ dp, int N, int C,
                                                       int64_t S, float dice_weight,
                                                       int square_dice) {
  const int64_t nc = blockIdx.y;
  const float g = dloss[0];
  const float O = sums[nc * 4 + 0];
  const float T = sums[nc * 4 + 1] + sums[nc * 4 + 2] + 1e-8f;
  const float inv_nc = 1.f / (float)(N * C);
  const float cw = class_w ? class_w[nc % C] : 1.f;
  const float klog = -(1.f - dice_weight) * cw * inv_nc / (float)S * g;
  const float kd1 = -dice_weight * inv_nc * 2.f / T * g;          // * t
  const float kd2 = dice_weight * inv_nc * 2.f * O / (T * T) * g;  // * dT/dp
  const float denom = (float)(1.0 + 1e-8);
  const float* pp = p + nc * S;
  const float* tp = t + nc * S;
  float* op = dp + nc * S;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < S; i += gridDim.x * 256ll) {
    const float pv = pp[i], tv = tp[i];
    const float ps = (pv + 1e-8f) / denom;
    float d = klog * tv / (ps * denom);
    d = fmaf(kd1, tv, d);
    d = fmaf(kd2, square_dice ? 2.f * pv : 1.f, d);
    op[i] = d;
  }
}

// ------------------------------------------------------------------ patches
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ vol,
                                                           const int32_t* __restrict__ loc,
                                                           float* __restrict__ patches, int P,
                                                           int C, int V0, int V1, int V2, int ps0,
                                                           int ps1, int ps2) {
  const int64_t PS = (int64_t)ps0 * ps1 * ps2;
  const int64_t total = (int64_t)P * C * PS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = (int)(i % ps2);
    int64_t r = i / ps2;
    const int j = (int)(r % ps1);
    r /= ps1;
    const int ii = (int)(r % ps0);
    r /= ps0;
    const int c = (int)(r % C);
    const int pidx = (int)(r / C);
    const int i0 = loc[pidx * 3], j0 = loc[pidx * 3 + 1], k0 = loc[pidx * 3 + 2];
    patches[i] = vol[(((int64_t)c * V0 + (i0 + ii)) * V1 + (j0 + j)) * V2 + (k0 + k)];
  }
}

// Padded variant: GridSampler(padding_mode=...) (prediction.py:114,132; torchio 0.18.45 pads the volume by
// patch_overlap // 2 per side with numpy.pad before tiling and GridAggregator crops it off again).  The padded
// volume is never materialised: `loc` is in PADDED coordinates and the source index is mapped back.
// mode: 0 constant (value), 1 edge, 2 reflect (mirror without repeating the border voxel), 3 symmetric, 4 wrap
__device__ __forceinline__ int pad_map(int p, int V, int mode) {  // p in [-b, V + b) -> source index, or -1
  if (p >= 0 && p < V) return p;
  switch (mode) {
    case 1: return min(max(p, 0), V - 1);
    case 2: {
      if (V == 1) return 0;
      const int period = 2 * V - 2;
      int q = p % period;
      if (q < 0) q += period;
      return q < V ? q : period - q;
    }
    case 3: {
      const int period = 2 * V;
      int q = p % period;
      if (q < 0) q += period;
      return q < V ? q : period - 1 - q;
    }
    case 4: {
      int q = p % V;
      return q < 0 ? q + V : q;
    }
    default: return -1;
  }
}

__global__ __launch_bounds__(256) void patch_gather_padded_kernel(const float* __restrict__ vol,
                                                                  const int32_t* __restrict__ loc,
                                                                  float* __restrict__ patches, int P, int C, int V0,
                                                                  int V1, int V2, int ps0, int ps1, int ps2, int b0,
                                                                  int b1, int b2, int mode, float value) {
  const int64_t PS = (int64_t)ps0 * ps1 * ps2;
  const int64_t total = (int64_t)P * C * PS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int k = (int)(i % ps2);
    int64_t r = i / ps2;
    const int j = (int)(r % ps1);
    r /= ps1;
    const int ii = (int)(r % ps0);
    r /= ps0;
    const int c = (int)(r % C);
    const int pidx = (int)(r / C);
    const int s0 = pad_map(loc[pidx * 3] + ii - b0, V0, mode), s1 = pad_map(loc[pidx * 3 + 1] + j - b1, V1, mode),
              s2 = pad_map(loc[pidx * 3 + 2] + k - b2, V2, mode);
    patches[i] = (s0 < 0 || s1 < 0 || s2 < 0) ? value : vol[(((int64_t)c * V0 + s0) * V1 + s1) * V2 + s2];
  }
}

// out[c, v] = accum[c, v + b] / count[v + b]: the average over covering patches with the padding cropped off
__global__ __launch_bounds__(256) void patch_finalize_crop_kernel(const float* __restrict__ accum,
                                                                  const float* __restrict__ count,
                                                                  float* __restrict__ out, int C, int P0, int P1, int P2,
                                                                  int b0, int b1, int b2) {
  const int V0 = P0 - 2 * b0, V1 = P1 - 2 * b1, V2 = P2 - 2 * b2;
  const int64_t V = (int64_t)V0 * V1 * V2, PV = (int64_t)P0 * P1 * P2;
  const int64_t total = (int64_t)C * V;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t v = i % V;
    const int c = (int)(i / V);
    const int k = (int)(v % V2), j = (int)((v / V2) % V1), ii = (int)(v / ((int64_t)V2 * V1));
    const int64_t pv = ((int64_t)(ii + b0) * P1 + (j + b1)) * P2 + (k + b2);
    out[i] = accum[(int64_t)c * PV + pv] / count[pv];
  }
}

// gather-form, deterministic: each output voxel sums the covering patches in patch order
// (== torchio's sequential `output[...] += patch` over the batch) and counts them.
__global__ __launch_bounds__(256) void patch_accumulate_kernel(const float* __restrict__ patches,
                                                               const int32_t* __restrict__ loc,
                                                               float* __restrict__ accum,
                                                               float* __restrict__ count, int P,
                                                               int C, int V0, int V1, int V2,
                                                               int ps0, int ps1, int ps2) {
  const int64_t V = (int64_t)V0 * V1 * V2;
  const int64_t PS = (int64_t)ps0 * ps1 * ps2;
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < V; v += gridDim.x * 256ll) {
    const int k = (int)(v % V2);
    const int j = (int)((v / V2) % V1);
    const int i = (int)(v / ((int64_t)V2 * V1));
    float cnt = count[v];
    for (int pidx = 0; pidx < P; ++pidx) {
      const int i0 = loc[pidx * 3], j0 = loc[pidx * 3 + 1], k0 = loc[pidx * 3 + 2];
      const int di = i - i0, dj = j - j0, dk = k - k0;
      if (di < 0 || di >= ps0 || dj < 0 || dj >= ps1 || dk < 0 || dk >= ps2) continue;
      const int64_t off = ((int64_t)di * ps1 + dj) * ps2 + dk;
      for (int c = 0; c < C; ++c)
        accum[(int64_t)c * V + v] += patches[((int64_t)pidx * C + c) * PS + off];
      cnt += 1.f;
    }
    count[v] = cnt;
  }
}

__global__ __launch_bounds__(256) void patch_finalize_kernel(const float* __restrict__ accum,
                                                             const float* __restrict__ count,
                                                             float* __restrict__ out, int C,
                                                             int64_t V) {
  const int64_t total = (int64_t)C * V;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll)
    out[i] = accum[i] / count[i % V];
}

// Aggregation of a whole tile grid in ONE pass (GridAggregator('average'), prediction.py:124-152): for every output
// voxel the tiles covering it are summed in tile order -- the order patch_accumulate adds them batch after batch, so
// the bits are identical -- and divided by their number.  The tiles of torchio's GridSampler lie on the product of three
// per-axis start lists (tile index = (a * n1 + b) * n2 + c); a voxel is covered by a contiguous run of starts per axis.
// No accumulator / count volumes, no zero-fill, no read-modify-write: tiles read once, output written once
// (cfg4: 1.3 GB of traffic in four phases -> 0.4 GB in one).  `border`: the output is the padded volume's interior.
__global__ __launch_bounds__(256) void patch_aggregate_grid_kernel(const float* __restrict__ tiles,
                                                                   const int32_t* __restrict__ starts, int n0, int n1, int n2,
                                                                   float* __restrict__ out, int C, int V0, int V1, int V2,
                                                                   int ps0, int ps1, int ps2, int b0, int b1, int b2) {
  const int64_t V = (int64_t)V0 * V1 * V2;
  const int64_t PS = (int64_t)ps0 * ps1 * ps2;
  const int32_t* s0 = starts;
  const int32_t* s1 = starts + n0;
  const int32_t* s2 = starts + n0 + n1;
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < V; v += gridDim.x * 256ll) {
    const int k = (int)(v % V2) + b2;                       // coordinates in the padded volume
    const int j = (int)((v / V2) % V1) + b1;
    const int i = (int)(v / ((int64_t)V2 * V1)) + b0;
    for (int c = 0; c < C; ++c) {
      float sum = 0.f;
      int cnt = 0;
      for (int a = 0; a < n0; ++a) {
        const int di = i - s0[a];
        if (di < 0 || di >= ps0) continue;
        for (int b = 0; b < n1; ++b) {
          const int dj = j - s1[b];
          if (dj < 0 || dj >= ps1) continue;
          for (int q = 0; q < n2; ++q) {
            const int dk = k - s2[q];
            if (dk < 0 || dk >= ps2) continue;
            const int64_t p = ((int64_t)a * n1 + b) * n2 + q;
            const float t = tiles[(p * C + c) * PS + ((int64_t)di * ps1 + dj) * ps2 + dk];
            sum = cnt == 0 ? t : sum + t;
            ++cnt;
          }
        }
      }
      out[(int64_t)c * V + v] = sum / (float)cnt;
    }
  }
}

// ------------------------------------------------------- argmax + confusion
// counts[(n*C + c)*4 + {TP, FP, FN, TN}]; block-level reduction then one atomic per
// (block, class, stat) -- integer adds, so the result is order-independent.
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ prob,
                                                               const int32_t* __restrict__ target,
                                                               int32_t* __restrict__ argmax_out,
                                                               unsigned long long* __restrict__ counts,
                                                               int C, int64_t S) {
  extern __shared__ unsigned int sh[];  // C*3: TP, FP, FN per class
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < C * 3; i += 256) sh[i] = 0;
  __syncthreads();
  const float* pn = prob + (int64_t)n * C * S;
  for (int64_t s = blockIdx.x * 256ll + threadIdx.x; s < S; s += gridDim.x * 256ll) {
    int best = 0;
    float bv = pn[s];
    for (int c = 1; c < C; ++c) {
      const float v = pn[(int64_t)c * S + s];
      if (v > bv) { bv = v; best = c; }  // first maximum wins (torch.argmax)
    }
    if (argmax_out) argmax_out[(int64_t)n * S + s] = best;
    const int tg = target[(int64_t)n * S + s];
    if (best == tg) {
      atomicAdd(&sh[best * 3 + 0], 1u);
    } else {
      atomicAdd(&sh[best * 3 + 1], 1u);
      if (tg >= 0 && tg < C) atomicAdd(&sh[tg * 3 + 2], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 3; i += 256) {
    if (sh[i]) atomicAdd(&counts[((int64_t)n * C + i / 3) * 4 + (i % 3)], (unsigned long long)sh[i]);
  }
}

// TN = S - TP - FP - FN
__global__ void confusion_tn_kernel(unsigned long long* __restrict__ counts, int NC, int64_t S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NC)
    counts[i * 4 + 3] = (unsigned long long)S - counts[i * 4] - counts[i * 4 + 1] - counts[i * 4 + 2];
}


// ------------------------------------------------------------------ device-side weighted patch sampling (N2)
// tio.WeightedSampler(patch_size, probability_map) (data_loader_factory.py:36-54, research/msseg2/msseg2.py:148-149;
// the map is ImageFromLabels' "brain 1, lesion 100", transforms/image_from_labels.py:11-57): a patch CENTRE is drawn with
// probability proportional to the map, restricted to centres whose patch fits in the volume.  Round 2 built a float64
// cumulative sum of the whole volume per CALL (134 MB for 256^3).  Here a two-level table is built once per map --
// table[b] = sum of the (clamped, border-masked) weights of all voxels before block b of 1024 voxels, table[nb] = the
// total -- and a draw is a binary search over the table plus a scan of ONE block: one wave per patch, one launch per
// batch.  All sums are fp64 in a fixed order, so a draw is a pure function of (map, u).
constexpr int SAMPLER_BLOCK = 1024;

struct SamplerGeom {
  int V0, V1, V2, lo0, lo1, lo2, hi0, hi1, hi2;   // a centre i is valid iff lo <= i < V - hi (per axis)
};
__device__ __forceinline__ double sampler_weight(const float* __restrict__ prob, int64_t idx, int64_t total,
                                                 const SamplerGeom& g) {
  if (idx >= total) return 0.0;
  const int k = (int)(idx % g.V2);
  const int64_t r = idx / g.V2;
  const int j = (int)(r % g.V1), i = (int)(r / g.V1);
  if (i < g.lo0 || i >= g.V0 - g.hi0 || j < g.lo1 || j >= g.V1 - g.hi1 || k < g.lo2 || k >= g.V2 - g.hi2) return 0.0;
  const float w = prob[idx];
  return w > 0.f ? (double)w : 0.0;      // negative weights count as 0, NaN as 0
}

// table[b + 1] = weight of block b (thread t: voxels 4t .. 4t+3 in order; then the fixed tree of block_sum)
__global__ __launch_bounds__(256) void sampler_block_sums_kernel(const float* __restrict__ prob, double* __restrict__ table,
                                                                 SamplerGeom g) {
  __shared__ double scratch[4];
  const int64_t total = (int64_t)g.V0 * g.V1 * g.V2;
  const int64_t base = (int64_t)blockIdx.x * SAMPLER_BLOCK + threadIdx.x * 4;
  double v = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) v += sampler_weight(prob, base + q, total, g);
  const double t = block_sum<double, 256>(v, scratch);
  if (threadIdx.x == 0) table[blockIdx.x + 1] = t;
}

// in place: table[1..nb] block weights -> table[b] = sum of the blocks before b (table[0] = 0, table[nb] = total)
__global__ __launch_bounds__(256) void sampler_prefix_kernel(double* __restrict__ table, int nb) {
  __shared__ double part[256];
  const int t = threadIdx.x;
  const int per = (nb + 255) / 256;
  const int b0 = min(nb, t * per), b1 = min(nb, b0 + per);
  double s = 0.0;
  for (int b = b0; b < b1; ++b) s += table[b + 1];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    double run = 0.0;
    for (int q = 0; q < 256; ++q) {
      const double x = part[q];
      part[q] = run;
      run += x;
    }
    table[0] = 0.0;
  }
  __syncthreads();
  double run = part[t];
  for (int b = b0; b < b1; ++b) {      // table[b + 1] <- inclusive prefix; read before write, each entry owned by one thread
    run += table[b + 1];
    table[b + 1] = run;
  }
}

// one wave per patch: corner location of the patch whose centre is the first voxel with cumulative weight > u * total
__global__ __launch_bounds__(64) void sampler_draw_kernel(const float* __restrict__ prob, const double* __restrict__ table,
                                                          int nb, SamplerGeom g, int p0, int p1, int p2,
                                                          const double* __restrict__ u, int P, int32_t* __restrict__ loc) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const int64_t total = (int64_t)g.V0 * g.V1 * g.V2;
  const double sum = table[nb];
  double target = u[p] * sum;
  if (!(target < sum)) target = sum * (1.0 - 1.1102230246251565e-16);   // u == 1 (rounding): the last weighted voxel
  if (!(target >= 0.0)) target = 0.0;
  int lo = 0, hi = nb;                       // largest b with table[b] <= target (wave-uniform)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (table[mid] <= target) lo = mid; else hi = mid;
  }
  const int b = lo;
  const double r = target - table[b];
  const int64_t base = (int64_t)b * SAMPLER_BLOCK + lane * 16;
  double w[16], s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    w[q] = sampler_weight(prob, base + q, total, g);
    s += w[q];
  }
  double incl = s;                            // inclusive scan over the lanes, fixed order
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const double excl = incl - s;
  // the lane whose range holds the target; rounding can leave no lane (r a hair above the block weight): the last
  // lane with positive weight then
  const unsigned long long hit = __ballot(s > 0.0 && incl > r);
  const unsigned long long pos = __ballot(s > 0.0);
  if (pos == 0ull) {                          // (an all-zero map: m355_sampler_build's caller checks table[nb] > 0)
    if (lane == 0) loc[p * 3] = loc[p * 3 + 1] = loc[p * 3 + 2] = 0;
    return;
  }
  const int L = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)pos);
  if (lane == L) {
    double run = excl;
    int pick = -1, last = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (w[q] > 0.0) {
        last = q;
        run += w[q];
        if (pick < 0 && run > r) pick = q;
      }
    }
    if (pick < 0) pick = last;
    const int64_t idx = base + pick;
    const int k = (int)(idx % g.V2);
    const int64_t rr = idx / g.V2;
    const int j = (int)(rr % g.V1), i = (int)(rr / g.V1);
    loc[p * 3 + 0] = min(max(i - p0 / 2, 0), g.V0 - p0);
    loc[p * 3 + 1] = min(max(j - p1 / 2, 0), g.V1 - p1);
    loc[p * 3 + 2] = min(max(k - p2 / 2, 0), g.V2 - p2);
  }
}

}  // namespace m355

using namespace m355;

extern "C" size_t m355_hybrid_loss_workspace(int32_t N, int32_t C, int64_t S) {
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  return (size_t)N * C * ceil_div(S, LOSS_CHUNK) * 4 * sizeof(double) + 256;
}

extern "C" int m355_hybrid_loss_fwd(const float* p, const float* t, int32_t N, int32_t C, int64_t S,
                                    float dice_weight, const float* class_weights,
                                    int32_t square_dice, float* out3, float* sums, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  M355_REQUIRE(p && t && out3 && sums && workspace, M355_EINVALID_ARG, "hybrid_loss_fwd: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0, M355_EINVALID_ARG, "hybrid_loss_fwd: non-positive size");
  M355_REQUIRE((int64_t)N * C <= 65535, M355_EUNSUPPORTED, "hybrid_loss_fwd: N*C > 65535");
  M355_REQUIRE(workspace_bytes >= m355_hybrid_loss_workspace(N, C, S), M355_EWORKSPACE,
               "hybrid_loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)ceil_div(S, LOSS_CHUNK);
  double* partial = (double*)workspace;
  hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)nblk, (unsigned)(N * C)), dim3(256), 0, st,
                     p, t, partial, S, square_dice, nblk);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, class_weights, sums,
                     out3, N, C, S, nblk, dice_weight);
  return check_launch("hybrid_loss_fwd");
}

extern "C" int m355_hybrid_loss_bwd(const float* p, const float* t, const float* sums,
                                    const float* dloss, int32_t N, int32_t C, int64_t S,
                                    float dice_weight, const float* class_weights,
                                    int32_t square_dice, float* dp, void* stream) {
  M355_REQUIRE(p && t && sums && dloss && dp, M355_EINVALID_ARG, "hybrid_loss_bwd: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0, M355_EINVALID_ARG, "hybrid_loss_bwd: non-positive size");
  M355_REQUIRE((int64_t)N * C <= 65535, M355_EUNSUPPORTED, "hybrid_loss_bwd: N*C > 65535");
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 1024), 2048));
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(bx, (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream,
                     p, t, sums, dloss, class_weights, dp, N, C, S, dice_weight, square_dice);
  return check_launch("hybrid_loss_bwd");
}

static int check_patch_args(int32_t P, int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0,
                            int32_t ps1, int32_t ps2, const char* who) {
  M355_REQUIRE(P > 0 && C > 0 && V0 > 0 && V1 > 0 && V2 > 0 && ps0 > 0 && ps1 > 0 && ps2 > 0,
               M355_EINVALID_ARG, "%s: non-positive size", who);
  M355_REQUIRE(ps0 <= V0 && ps1 <= V1 && ps2 <= V2, M355_EINVALID_ARG,
               "%s: patch (%d,%d,%d) larger than volume (%d,%d,%d)", who, ps0, ps1, ps2, V0, V1, V2);
  return M355_OK;
}

static int sampler_geom(const char* who, int V0, int V1, int V2, int p0, int p1, int p2, SamplerGeom* g) {
  M355_REQUIRE(V0 > 0 && V1 > 0 && V2 > 0 && p0 > 0 && p1 > 0 && p2 > 0, M355_EINVALID_ARG, "%s: non-positive size", who);
  M355_REQUIRE(p0 <= V0 && p1 <= V1 && p2 <= V2, M355_EINVALID_ARG, "%s: patch (%d,%d,%d) exceeds the volume (%d,%d,%d)", who, p0,
               p1, p2, V0, V1, V2);
  M355_REQUIRE(ceil_div((int64_t)V0 * V1 * V2, SAMPLER_BLOCK) < (1ll << 30), M355_EUNSUPPORTED, "%s: volume too large", who);
  // centre index inside the patch: p // 2; voxels after the centre: p - p // 2 - 1
  *g = SamplerGeom{V0, V1, V2, p0 / 2, p1 / 2, p2 / 2, p0 - p0 / 2 - 1, p1 - p1 / 2 - 1, p2 - p2 / 2 - 1};
  return M355_OK;
}

extern "C" size_t m355_sampler_table_bytes(int32_t V0, int32_t V1, int32_t V2) {
  if (V0 <= 0 || V1 <= 0 || V2 <= 0) return 0;
  return (size_t)(ceil_div((int64_t)V0 * V1 * V2, SAMPLER_BLOCK) + 1) * sizeof(double);
}

extern "C" int m355_sampler_build(const float* prob, int32_t V0, int32_t V1, int32_t V2, int32_t p0, int32_t p1, int32_t p2,
                                  double* table, void* stream) {
  SamplerGeom g;
  if (int rc = sampler_geom("sampler_build", V0, V1, V2, p0, p1, p2, &g)) return rc;
  M355_REQUIRE(prob && table, M355_EINVALID_ARG, "sampler_build: null pointer");
  const int nb = (int)ceil_div((int64_t)V0 * V1 * V2, SAMPLER_BLOCK);
  hipLaunchKernelGGL(sampler_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, prob, table, g);
  hipLaunchKernelGGL(sampler_prefix_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, table, nb);
  return check_launch("sampler_build");
}

extern "C" int m355_sampler_draw(const float* prob, const double* table, int32_t V0, int32_t V1, int32_t V2, int32_t p0,
                                 int32_t p1, int32_t p2, const double* u, int32_t P, int32_t* locations, void* stream) {
  SamplerGeom g;
  if (int rc = sampler_geom("sampler_draw", V0, V1, V2, p0, p1, p2, &g)) return rc;
  M355_REQUIRE(prob && table && u && locations && P > 0, M355_EINVALID_ARG, "sampler_draw: null pointer / no patches");
  const int nb = (int)ceil_div((int64_t)V0 * V1 * V2, SAMPLER_BLOCK);
  hipLaunchKernelGGL(sampler_draw_kernel, dim3((unsigned)P), dim3(64), 0, (hipStream_t)stream, prob, table, nb, g, p0, p1, p2,
                     u, P, locations);
  return check_launch("sampler_draw");
}

extern "C" int m355_patch_gather(const float* volume, const int32_t* loc, float* patches, int32_t P,
                                 int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0,
                                 int32_t ps1, int32_t ps2, void* stream) {
  if (int rc = check_patch_args(P, C, V0, V1, V2, ps0, ps1, ps2, "patch_gather")) return rc;
  M355_REQUIRE(volume && loc && patches, M355_EINVALID_ARG, "patch_gather: null pointer");
  const int64_t total = (int64_t)P * C * ps0 * ps1 * ps2;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 16384);
  hipLaunchKernelGGL(patch_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, volume,
                     loc, patches, P, C, V0, V1, V2, ps0, ps1, ps2);
  return check_launch("patch_gather");
}

extern "C" int m355_patch_gather_padded(const float* volume, const int32_t* loc, float* patches, int32_t P, int32_t C,
                                        int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                                        int32_t b0, int32_t b1, int32_t b2, int32_t mode, float value, void* stream) {
  if (int rc = check_patch_args(P, C, V0 + 2 * b0, V1 + 2 * b1, V2 + 2 * b2, ps0, ps1, ps2, "patch_gather_padded")) return rc;
  M355_REQUIRE(volume && loc && patches, M355_EINVALID_ARG, "patch_gather_padded: null pointer");
  M355_REQUIRE(b0 >= 0 && b1 >= 0 && b2 >= 0 && mode >= 0 && mode <= 4, M355_EINVALID_ARG,
               "patch_gather_padded: bad border / mode");
  const int64_t total = (int64_t)P * C * ps0 * ps1 * ps2;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 16384);
  hipLaunchKernelGGL(patch_gather_padded_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, volume, loc, patches,
                     P, C, V0, V1, V2, ps0, ps1, ps2, b0, b1, b2, mode, value);
  return check_launch("patch_gather_padded");
}

extern "C" int m355_patch_finalize_crop(const float* accum, const float* count, float* out, int32_t C, int32_t P0,
                                        int32_t P1, int32_t P2, int32_t b0, int32_t b1, int32_t b2, void* stream) {
  M355_REQUIRE(accum && count && out, M355_EINVALID_ARG, "patch_finalize_crop: null pointer");
  M355_REQUIRE(C > 0 && b0 >= 0 && b1 >= 0 && b2 >= 0 && P0 > 2 * b0 && P1 > 2 * b1 && P2 > 2 * b2, M355_EINVALID_ARG,
               "patch_finalize_crop: bad shape");
  const int64_t total = (int64_t)C * (P0 - 2 * b0) * (P1 - 2 * b1) * (P2 - 2 * b2);
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 16384);
  hipLaunchKernelGGL(patch_finalize_crop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, accum, count, out, C,
                     P0, P1, P2, b0, b1, b2);
  return check_launch("patch_finalize_crop");
}

extern "C" int m355_patch_aggregate_grid(const float* tiles, const int32_t* starts, int32_t n0, int32_t n1, int32_t n2,
                                         float* out, int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1,
                                         int32_t ps2, int32_t b0, int32_t b1, int32_t b2, void* stream) {
  M355_REQUIRE(tiles && starts && out, M355_EINVALID_ARG, "patch_aggregate_grid: null pointer");
  M355_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && C > 0 && V0 > 0 && V1 > 0 && V2 > 0 && ps0 > 0 && ps1 > 0 && ps2 > 0 && b0 >= 0 &&
                   b1 >= 0 && b2 >= 0, M355_EINVALID_ARG, "patch_aggregate_grid: bad shape");
  const int64_t V = (int64_t)V0 * V1 * V2;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(V, 256), 32768);
  hipLaunchKernelGGL(patch_aggregate_grid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tiles, starts, n0, n1, n2,
                     out, C, V0, V1, V2, ps0, ps1, ps2, b0, b1, b2);
  return check_launch("patch_aggregate_grid");
}

extern "C" int m355_patch_accumulate(const float* patches, const int32_t* loc, float* accum,
                                     float* count, int32_t P, int32_t C, int32_t V0, int32_t V1,
                                     int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                                     void* stream) {
  if (int rc = check_patch_args(P, C, V0, V1, V2, ps0, ps1, ps2, "patch_accumulate")) return rc;
  M355_REQUIRE(patches && loc && accum && count, M355_EINVALID_ARG, "patch_accumulate: null pointer");
  const int64_t V = (int64_t)V0 * V1 * V2;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(V, 256), 16384);
  hipLaunchKernelGGL(patch_accumulate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     patches, loc, accum, count, P, C, V0, V1, V2, ps0, ps1, ps2);
  return check_launch("patch_accumulate");
}

extern "C" int m355_patch_finalize(const float* accum, const float* count, float* out, int32_t C,
                                   int64_t V, void* stream) {
  M355_REQUIRE(accum && count && out, M355_EINVALID_ARG, "patch_finalize: null pointer");
  M355_REQUIRE(C > 0 && V > 0, M355_EINVALID_ARG, "patch_finalize: non-positive size");
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div((int64_t)C * V, 256), 16384);
  hipLaunchKernelGGL(patch_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, accum,
                     count, out, C, V);
  return check_launch("patch_finalize");
}

extern "C" int m355_argmax_confusion(const float* prob, const int32_t* target, int32_t* argmax_out,
                                     int64_t* counts, int32_t N, int32_t C, int64_t S,
                                     void* stream) {
  M355_REQUIRE(prob && target && counts, M355_EINVALID_ARG, "argmax_confusion: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0, M355_EINVALID_ARG, "argmax_confusion: non-positive size");
  M355_REQUIRE(N <= 65535 && C <= 4096, M355_EUNSUPPORTED, "argmax_confusion: N > 65535 or C > 4096");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)N * C * 4 * sizeof(int64_t), st);
  M355_REQUIRE(e == hipSuccess, M355_ELAUNCH, "argmax_confusion: memset failed: %s",
               hipGetErrorString(e));
  // each block handles <= 2^31 voxels, so the 32-bit LDS counters cannot overflow
  const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 2048), 1024));
  hipLaunchKernelGGL(argmax_confusion_kernel, dim3(bx, (unsigned)N), dim3(256),
                     (size_t)C * 3 * sizeof(unsigned int), st, prob, target, argmax_out,
                     (unsigned long long*)counts, C, S);
  hipLaunchKernelGGL(confusion_tn_kernel, dim3((unsigned)ceil_div((int64_t)N * C, 64)), dim3(64), 0,
                     st, (unsigned long long*)counts, N * C, S);
  return check_launch("argmax_confusion");
}
