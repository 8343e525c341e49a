// Weight transform of the reference's Blur convolutions, forward and backward, in one kernel each.
//
// Reference ops replaced (models/components.py): BlurConv3d.forward :112-119 and
// BlurConvTranspose3d.forward :145-152 -- optional weight standardisation
//   w <- (w - mean_a) / (std_a + 1e-5)      (per dim-0 filter a, unbiased std, :114-116 / :147-149)
// followed by F.conv3d(w, kernel, padding=1, groups=C) with the all-equal 2x2x2 `kernel` buffer
// (:118 / :151): a 3x3x3 filter becomes a 4x4x4 one, out[i] = scale_b * sum_{d in {0,1}^3} w[i+d-1].
// The 4x4x4 / stride-2 (transposed) convolution is then run as a stride-1 3x3x3 convolution over the
// space-to-depth tensor (elementwise.hip), whose sparse filter is a gather of the 4x4x4 one:
//   strided conv      wexp[a][b*8 + p][t] = blur[a][b][d],  d = 2t - 1 + p  per axis (valid 0..3)
//   transposed conv   wexp[b*8 + p][a][t] = blur[a][b][d],  d = 3 - 2t + p  per axis
// (p = parity bits pz*4+py*2+px, t = tap tz*9+ty*3+tx); invalid (p, t) pairs are zero.
// The weights are tiny (<= a few hundred KB): one workgroup per dim-0 filter, nothing to tune.
#include "common.hpp"

namespace m355 {

// per-axis gather index of the 4-tap blurred filter for (parity, tap); < 0 or > 3: structurally zero
__device__ __forceinline__ int blur_d(int par, int tap, int transposed) {
  return transposed ? 3 - 2 * tap + par : 2 * tap - 1 + par;
}
// inverse: the (parity, tap) that reads blurred tap d
__device__ __forceinline__ void blur_pt(int d, int transposed, int* par, int* tap) {
  *par = (d + 1) & 1;
  *tap = transposed ? (3 + *par - d) / 2 : (d + 1 - *par) / 2;
}

// mean / std statistics of filter a (double accumulation), broadcast to the block
__device__ __forceinline__ void filter_stats(const float* __restrict__ wa, int n, double* scratch, float* mean,
                                             float* stdv) {
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = wa[i];
    s1 += v;
    s2 += v * v;
  }
  s1 = block_sum<double, 256, true>(s1, scratch);
  s2 = block_sum<double, 256, true>(s2, scratch);
  const double m = s1 / n;
  double var = n > 1 ? (s2 - n * m * m) / (n - 1) : 0.0;  // torch.std: unbiased
  if (var < 0.0) var = 0.0;
  *mean = (float)m;
  *stdv = (float)sqrt(var);
}

__global__ __launch_bounds__(256) void blur_weight_fwd_kernel(const float* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              float* __restrict__ wexp, float* __restrict__ mean_std,
                                                              int A, int B, int standardize, int transposed) {
  __shared__ double scratch[4];
  const int a = blockIdx.x;
  const float* wa = w + (int64_t)a * B * 27;
  float m = 0.f, inv = 1.f;
  if (standardize) {
    float sd;
    filter_stats(wa, B * 27, scratch, &m, &sd);
    inv = 1.f / (sd + 1e-5f);
    if (threadIdx.x == 0 && blockIdx.y == 0) {
      mean_std[a * 2 + 0] = m;
      mean_std[a * 2 + 1] = sd;
    }
  }
  // blockIdx.y splits the B*216 outputs of a filter into chunks (the statistics above are recomputed per
  // chunk: a few KB of cached reads) so that small filter counts still fill the chip
  const int chunk = (B * 216 + (int)gridDim.y - 1) / (int)gridDim.y;
  const int e_end = min(B * 216, ((int)blockIdx.y + 1) * chunk);
  for (int e = (int)blockIdx.y * chunk + threadIdx.x; e < e_end; e += 256) {
    const int b = e / 216, r = e - b * 216;
    const int p = r / 27, t = r - p * 27;
    const int dz = blur_d(p >> 2, t / 9, transposed), dy = blur_d((p >> 1) & 1, (t / 3) % 3, transposed),
              dx = blur_d(p & 1, t % 3, transposed);
    float v = 0.f;
    if (dz >= 0 && dz <= 3 && dy >= 0 && dy <= 3 && dx >= 0 && dx <= 3) {
      // 8 shifted adds in (z, y, x) order, zeros included: the summation order of the torch restatement
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int z = dz + (q >> 2) - 1, y = dy + ((q >> 1) & 1) - 1, x = dx + (q & 1) - 1;
        const bool in = z >= 0 && z <= 2 && y >= 0 && y <= 2 && x >= 0 && x <= 2;
        const float wv = in ? (wa[b * 27 + (z * 3 + y) * 3 + x] - m) * inv : 0.f;
        acc = q == 0 ? wv : acc + wv;
      }
      v = acc * scale[b];
    }
    const int64_t o = transposed ? (((int64_t)b * 8 + p) * A + a) * 27 + t : (((int64_t)a * B + b) * 8 + p) * 27 + t;
    wexp[o] = v;
  }
}

// gradient of the (not yet de-standardised) filter element (b, x) of filter a
__device__ __forceinline__ float blur_grad_elem(const float* __restrict__ dwexp, const float* __restrict__ scale,
                                                int a, int b, int x, int A, int B, int transposed) {
  const int xz = x / 9, xy = (x / 3) % 3, xx = x % 3;
  float g = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {  // w[x] feeds the blurred taps x + {0,1}^3
    const int dz = xz + (q >> 2), dy = xy + ((q >> 1) & 1), dx = xx + (q & 1);
    int pz, tz, py, ty, px, tx;
    blur_pt(dz, transposed, &pz, &tz);
    blur_pt(dy, transposed, &py, &ty);
    blur_pt(dx, transposed, &px, &tx);
    const int p = pz * 4 + py * 2 + px, t = tz * 9 + ty * 3 + tx;
    const int64_t o = transposed ? (((int64_t)b * 8 + p) * A + a) * 27 + t : (((int64_t)a * B + b) * 8 + p) * 27 + t;
    g += dwexp[o];
  }
  return g * scale[b];
}

__global__ __launch_bounds__(256) void blur_weight_bwd_kernel(const float* __restrict__ dwexp,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ mean_std,
                                                              float* __restrict__ dw, int A, int B, int standardize,
                                                              int transposed) {
  __shared__ double scratch[4];
  const int a = blockIdx.x;
  const int n = B * 27;
  const float* wa = w + (int64_t)a * n;
  float* dwa = dw + (int64_t)a * n;
  if (!standardize) {
    for (int i = threadIdx.x; i < n; i += 256) dwa[i] = blur_grad_elem(dwexp, scale, a, i / 27, i % 27, A, B, transposed);
    return;
  }
  // wn_i = (w_i - m) / (s + eps), s = unbiased std:
  //   dw_k = (g_k - mean(g)) / (s + eps) - (w_k - m) * sum_i g_i (w_i - m) / ((s + eps)^2 (n - 1) s)
  const float m = mean_std[a * 2 + 0], sd = mean_std[a * 2 + 1];
  double sg = 0.0, sgw = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double g = blur_grad_elem(dwexp, scale, a, i / 27, i % 27, A, B, transposed);
    sg += g;
    sgw += g * ((double)wa[i] - m);
  }
  sg = block_sum<double, 256, true>(sg, scratch);
  sgw = block_sum<double, 256, true>(sgw, scratch);
  const double se = (double)sd + 1e-5;
  const double c1 = sg / n;
  const double c2 = (n > 1 && sd > 0.f) ? sgw / (se * se * (n - 1) * (double)sd) : 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double g = blur_grad_elem(dwexp, scale, a, i / 27, i % 27, A, B, transposed);
    dwa[i] = (float)((g - c1) / se - ((double)wa[i] - m) * c2);
  }
}

// Plain weight standardisation of WSConv3d (models/components.py:81-88):
//   wn[a][i] = (w[a][i] - mean_a) / (std_a + 1e-5),  unbiased std over the n = Cin*k^3 entries of filter a.
__global__ __launch_bounds__(256) void weight_standardize_fwd_kernel(const float* __restrict__ w, float* __restrict__ wn,
                                                                     float* __restrict__ mean_std, int n) {
  __shared__ double scratch[4];
  const int a = blockIdx.x;
  const float* wa = w + (int64_t)a * n;
  float m, sd;
  filter_stats(wa, n, scratch, &m, &sd);
  const float inv = 1.f / (sd + 1e-5f);
  if (threadIdx.x == 0) {
    mean_std[a * 2 + 0] = m;
    mean_std[a * 2 + 1] = sd;
  }
  for (int i = threadIdx.x; i < n; i += 256) wn[(int64_t)a * n + i] = (wa[i] - m) * inv;
}

// dw_k = (g_k - mean(g)) / (s + eps) - (w_k - m) * sum_i g_i (w_i - m) / ((s + eps)^2 (n - 1) s)
__global__ __launch_bounds__(256) void weight_standardize_bwd_kernel(const float* __restrict__ dwn,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ mean_std,
                                                                     float* __restrict__ dw, int n) {
  __shared__ double scratch[4];
  const int a = blockIdx.x;
  const float* wa = w + (int64_t)a * n;
  const float* ga = dwn + (int64_t)a * n;
  const float m = mean_std[a * 2 + 0], sd = mean_std[a * 2 + 1];
  double sg = 0.0, sgw = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double g = ga[i];
    sg += g;
    sgw += g * ((double)wa[i] - m);
  }
  sg = block_sum<double, 256, true>(sg, scratch);
  sgw = block_sum<double, 256, true>(sgw, scratch);
  const double se = (double)sd + 1e-5;
  const double c1 = sg / n;
  const double c2 = (n > 1 && sd > 0.f) ? sgw / (se * se * (n - 1) * (double)sd) : 0.0;
  for (int i = threadIdx.x; i < n; i += 256)
    dw[(int64_t)a * n + i] = (float)(((double)ga[i] - c1) / se - ((double)wa[i] - m) * c2);
}

}  // namespace m355

using namespace m355;

extern "C" int m355_weight_standardize_fwd(const float* w, float* wn, float* mean_std, int32_t A, int32_t n,
                                           void* stream) {
  M355_REQUIRE(w && wn && mean_std, M355_EINVALID_ARG, "weight_standardize_fwd: null pointer");
  M355_REQUIRE(A > 0 && n > 0, M355_EINVALID_ARG, "weight_standardize_fwd: bad filter shape (%d, %d)", A, n);
  hipLaunchKernelGGL(weight_standardize_fwd_kernel, dim3((unsigned)A), dim3(256), 0, (hipStream_t)stream, w, wn,
                     mean_std, n);
  return check_launch("weight_standardize_fwd");
}

extern "C" int m355_weight_standardize_bwd(const float* dwn, const float* w, const float* mean_std, float* dw,
                                           int32_t A, int32_t n, void* stream) {
  M355_REQUIRE(dwn && w && mean_std && dw, M355_EINVALID_ARG, "weight_standardize_bwd: null pointer");
  M355_REQUIRE(A > 0 && n > 0, M355_EINVALID_ARG, "weight_standardize_bwd: bad filter shape (%d, %d)", A, n);
  hipLaunchKernelGGL(weight_standardize_bwd_kernel, dim3((unsigned)A), dim3(256), 0, (hipStream_t)stream, dwn, w,
                     mean_std, dw, n);
  return check_launch("weight_standardize_bwd");
}

extern "C" int m355_blur_weight_fwd(const float* w, const float* scale, float* wexp, float* mean_std, int32_t A,
                                    int32_t B, int32_t standardize, int32_t transposed, void* stream) {
  M355_REQUIRE(w && scale && wexp && (mean_std || !standardize), M355_EINVALID_ARG, "blur_weight_fwd: null pointer");
  M355_REQUIRE(A > 0 && B > 0 && A <= 65535 && (int64_t)B * 216 < (1ll << 31), M355_EINVALID_ARG,
               "blur_weight_fwd: bad filter count (%d, %d)", A, B);
  // ~2048 outputs per block
  const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div((int64_t)B * 216, 2048), 64));
  hipLaunchKernelGGL(blur_weight_fwd_kernel, dim3((unsigned)A, chunks), dim3(256), 0, (hipStream_t)stream, w, scale,
                     wexp, mean_std, A, B, standardize, transposed);
  return check_launch("blur_weight_fwd");
}

extern "C" int m355_blur_weight_bwd(const float* dwexp, const float* w, const float* scale, const float* mean_std,
                                    float* dw, int32_t A, int32_t B, int32_t standardize, int32_t transposed,
                                    void* stream) {
  M355_REQUIRE(dwexp && w && scale && dw && (mean_std || !standardize), M355_EINVALID_ARG,
               "blur_weight_bwd: null pointer");
  M355_REQUIRE(A > 0 && B > 0 && A <= 65535 && (int64_t)B * 216 < (1ll << 31), M355_EINVALID_ARG,
               "blur_weight_bwd: bad filter count (%d, %d)", A, B);
  hipLaunchKernelGGL(blur_weight_bwd_kernel, dim3((unsigned)A), dim3(256), 0, (hipStream_t)stream, dwexp, w, scale,
                     mean_std, dw, A, B, standardize, transposed);
  return check_launch("blur_weight_bwd");
}
