// ConvTranspose3d for gfx950.
//
// Reference ops replaced: nn.ConvTranspose3d(kernel_size=2, stride=2) reached
// through ModularUNet's upsample_class hook (models/modular_unet.py:20-21,72-81,96)
// and the F.conv_transpose3d of BlurConvTranspose3d (models/components.py:152,
// effective k=4, s=2, p=1).
//
// k=2,s=2,p=0 is a non-overlapping scatter: every input voxel produces its own
// 2x2x2 output block, i.e. a [8*Cout x Cin] x [Cin x voxels] GEMM with AI ~ 28
// flop/B in fp32 -> HBM-bound.  The fast path keeps one
// input voxel per lane (coalesced x reads, float2-coalesced y writes) and reads
// the weights through the scalar cache (block-uniform addresses).
#include "common.hpp"

namespace m355 {

constexpr int CT_OT = 4;  // output channels per thread in the k2s2 forward (32 accumulators)

// grid: (voxel blocks, Cout/CT_OT, N).  w: [Cin, Cout, 2,2,2]
__global__ __launch_bounds__(256) void convt_k2s2_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs) {
  const int64_t S = (int64_t)D * H * W;
  const int64_t v = blockIdx.x * 256ll + threadIdx.x;
  const int o0 = blockIdx.y * CT_OT;
  const int n = blockIdx.z;
  const bool active = v < S;
  const float* xp = x + (int64_t)n * xbs + (active ? v : 0);
  float acc[CT_OT][8];
#pragma unroll
  for (int j = 0; j < CT_OT; ++j) {
    const float b = (bias && o0 + j < Cout) ? bias[o0 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[j][t] = b;
  }
  const int no = min(CT_OT, Cout - o0);
  for (int c = 0; c < Cin; ++c) {
    const float xv = active ? xp[(int64_t)c * S] : 0.f;
    const float* wc = w + ((int64_t)c * Cout + o0) * 8;  // block-uniform -> scalar loads
#pragma unroll
    for (int j = 0; j < CT_OT; ++j) {
      if (j < no) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[j][t] = fmaf(xv, wc[j * 8 + t], acc[j][t]);
      }
    }
  }
  if (!active) return;
  const int ix = (int)(v % W);
  const int iy = (int)((v / W) % H);
  const int iz = (int)(v / ((int64_t)W * H));
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = S * 8;
#pragma unroll
  for (int j = 0; j < CT_OT; ++j) {
    if (j < no) {
      float* yo = y + (int64_t)n * ybs + (int64_t)(o0 + j) * OS;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float2 val = make_float2(acc[j][a * 4 + b * 2], acc[j][a * 4 + b * 2 + 1]);
          *reinterpret_cast<float2*>(yo + ((int64_t)(2 * iz + a) * OH + (2 * iy + b)) * OW + 2 * ix) =
              val;
        }
    }
  }
}

constexpr int CT_CT = 16;  // input channels per thread in the k2s2 data gradient

// dx[n,c,v] = sum_{o,t} dy[n,o,2v+t] * w[c,o,t]; grid: (voxel blocks, Cin/CT_CT, N)
__global__ __launch_bounds__(256) void convt_k2s2_bwd_data_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int Cin,
    int Cout, int D, int H, int W, int64_t xbs, int64_t ybs) {
  const int64_t S = (int64_t)D * H * W;
  const int64_t v = blockIdx.x * 256ll + threadIdx.x;
  const int c0 = blockIdx.y * CT_CT;
  const int n = blockIdx.z;
  if (v >= S) return;
  const int ix = (int)(v % W);
  const int iy = (int)((v / W) % H);
  const int iz = (int)(v / ((int64_t)W * H));
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = S * 8;
  const int nc = min(CT_CT, Cin - c0);
  float acc[CT_CT];
#pragma unroll
  for (int j = 0; j < CT_CT; ++j) acc[j] = 0.f;
  const float* dp = dy + (int64_t)n * ybs + ((int64_t)(2 * iz) * OH + 2 * iy) * OW + 2 * ix;
  for (int o = 0; o < Cout; ++o) {
    const float* q = dp + (int64_t)o * OS;
    float g[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float2 t = *reinterpret_cast<const float2*>(q + ((int64_t)a * OH + b) * OW);
        g[a * 4 + b * 2] = t.x;
        g[a * 4 + b * 2 + 1] = t.y;
      }
#pragma unroll
    for (int j = 0; j < CT_CT; ++j) {
      if (j < nc) {
        const float* wc = w + ((int64_t)(c0 + j) * Cout + o) * 8;  // block-uniform
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[j] = fmaf(g[t], wc[t], acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CT_CT; ++j)
    if (j < nc) dx[(int64_t)n * xbs + (int64_t)(c0 + j) * S + v] = acc[j];
}

// dw[c,o,t] = sum_{n,v} x[n,c,v] * dy[n,o,2v+t].
// grid: (splits, ceil(Cout/8), ceil(Cin/32)).  A block owns a 32 c x 8 o tile and walks its
// slice of voxels 64 at a time through LDS: xs[c][v] (stride 65: conflict-free per c) and
// dys[o][v][8 taps] (block-broadcast b128 reads).  Thread (c = tid&31, o = tid>>5) keeps the
// 8 taps of its (c,o) pair; fp32 over 64 voxels, flushed to double every step.
__global__ __launch_bounds__(256) void convt_k2s2_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N,
    int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs, int nsplit) {
  __shared__ float xs[32 * 65];
  __shared__ __attribute__((aligned(16))) float dys[8 * 64 * 8];
  const int tid = threadIdx.x;
  const int split = blockIdx.x, o0 = blockIdx.y * 8, c0 = blockIdx.z * 32;
  const int64_t S = (int64_t)D * H * W;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = S * 8;
  const int64_t total = (int64_t)N * S;
  const int64_t nsteps = (total + 63) / 64;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t step_begin = split * per, step_end = min(nsteps, step_begin + per);
  const int tc = tid & 31, to = tid >> 5;
  double dacc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) dacc[t] = 0.0;
  for (int64_t step = step_begin; step < step_end; ++step) {
    const int64_t g0 = step * 64;
    __syncthreads();
    // stage x: 32 channels x 64 voxels
    for (int i = tid; i < 32 * 64; i += 256) {
      const int c = i >> 6, vv = i & 63;
      const int64_t g = g0 + vv;
      float val = 0.f;
      if (g < total && c0 + c < Cin) {
        const int64_t n = g / S, v = g - n * S;
        val = x[n * xbs + (int64_t)(c0 + c) * S + v];
      }
      xs[c * 65 + vv] = val;
    }
    // stage dy: 8 o x 64 voxels x (2x2) float2
    for (int i = tid; i < 8 * 4 * 64; i += 256) {
      const int vv = i & 63, ab = (i >> 6) & 3, o = i >> 8;
      const int64_t g = g0 + vv;
      float2 val = make_float2(0.f, 0.f);
      if (g < total && o0 + o < Cout) {
        const int64_t n = g / S, v = g - n * S;
        const int ix = (int)(v % W);
        const int iy = (int)((v / W) % H);
        const int iz = (int)(v / ((int64_t)W * H));
        val = *reinterpret_cast<const float2*>(
            dy + n * ybs + (int64_t)(o0 + o) * OS +
            ((int64_t)(2 * iz + (ab >> 1)) * OH + (2 * iy + (ab & 1))) * OW + 2 * ix);
      }
      *reinterpret_cast<float2*>(dys + (o * 64 + vv) * 8 + ab * 2) = val;
    }
    __syncthreads();
    float part[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) part[t] = 0.f;
#pragma unroll 8
    for (int vv = 0; vv < 64; ++vv) {
      const float xv = xs[tc * 65 + vv];
      const float4 d0 = *reinterpret_cast<const float4*>(dys + (to * 64 + vv) * 8);
      const float4 d1 = *reinterpret_cast<const float4*>(dys + (to * 64 + vv) * 8 + 4);
      part[0] = fmaf(xv, d0.x, part[0]); part[1] = fmaf(xv, d0.y, part[1]);
      part[2] = fmaf(xv, d0.z, part[2]); part[3] = fmaf(xv, d0.w, part[3]);
      part[4] = fmaf(xv, d1.x, part[4]); part[5] = fmaf(xv, d1.y, part[5]);
      part[6] = fmaf(xv, d1.z, part[6]); part[7] = fmaf(xv, d1.w, part[7]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) dacc[t] += (double)part[t];
  }
  if (c0 + tc < Cin && o0 + to < Cout) {
    float* out = slab + (((int64_t)split * Cin + (c0 + tc)) * Cout + (o0 + to)) * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) out[t] = (float)dacc[t];
  }
}

__global__ void convt_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                         int64_t total, int nsplit) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    double v = 0.0;
    for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * total + i];
    out[i] = (float)v;
  }
}

// ----------------------------------------------------- generic direct kernels
// y[n,o,oz,oy,ox] = bias[o] + sum_{c, taps: (o + pad - d) % stride == 0} x[n,c,(o+pad-d)/stride] * w[c,o,d]
__global__ void convt_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                        const float* __restrict__ bias, float* __restrict__ y,
                                        int N, int Cin, int Cout, int D, int H, int W, int OD, int OH,
                                        int OW, int k, int stride, int pad, int64_t xbs,
                                        int64_t ybs) {
  const int64_t OS = (int64_t)OD * OH * OW;
  const int64_t total = (int64_t)N * Cout * OS;
  const int k3 = k * k * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    r /= OD;
    const int o = (int)(r % Cout);
    const int n = (int)(r / Cout);
    float acc = bias ? bias[o] : 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float* xc = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
      const float* wc = w + ((int64_t)c * Cout + o) * k3;
      for (int dz = 0; dz < k; ++dz) {
        const int tz = oz + pad - dz;
        if (tz < 0 || tz % stride) continue;
        const int iz = tz / stride;
        if (iz >= D) continue;
        for (int dy = 0; dy < k; ++dy) {
          const int ty = oy + pad - dy;
          if (ty < 0 || ty % stride) continue;
          const int iy = ty / stride;
          if (iy >= H) continue;
          for (int dx = 0; dx < k; ++dx) {
            const int tx = ox + pad - dx;
            if (tx < 0 || tx % stride) continue;
            const int ix = tx / stride;
            if (ix >= W) continue;
            acc = fmaf(xc[((int64_t)iz * H + iy) * W + ix], wc[(dz * k + dy) * k + dx], acc);
          }
        }
      }
    }
    y[(int64_t)n * ybs + (int64_t)o * OS + ((int64_t)oz * OH + oy) * OW + ox] = acc;
  }
}

// dx[n,c,iz,iy,ix] = sum_{o,d} dy[n,o,i*stride + d - pad] * w[c,o,d]
__global__ void convt_direct_bwd_data_kernel(const float* __restrict__ dy,
                                             const float* __restrict__ w, float* __restrict__ dx,
                                             int N, int Cin, int Cout, int D, int H, int W, int OD,
                                             int OH, int OW, int k, int stride, int pad,
                                             int64_t xbs, int64_t ybs) {
  const int64_t S = (int64_t)D * H * W;
  const int64_t OS = (int64_t)OD * OH * OW;
  const int64_t total = (int64_t)N * Cin * S;
  const int k3 = k * k * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    int64_t r = i / W;
    const int iy = (int)(r % H);
    r /= H;
    const int iz = (int)(r % D);
    r /= D;
    const int c = (int)(r % Cin);
    const int n = (int)(r / Cin);
    float acc = 0.f;
    for (int o = 0; o < Cout; ++o) {
      const float* dyo = dy + (int64_t)n * ybs + (int64_t)o * OS;
      const float* wc = w + ((int64_t)c * Cout + o) * k3;
      for (int dz = 0; dz < k; ++dz) {
        const int oz = iz * stride + dz - pad;
        if (oz < 0 || oz >= OD) continue;
        for (int dyy = 0; dyy < k; ++dyy) {
          const int oy = iy * stride + dyy - pad;
          if (oy < 0 || oy >= OH) continue;
          for (int dxx = 0; dxx < k; ++dxx) {
            const int ox = ix * stride + dxx - pad;
            if (ox < 0 || ox >= OW) continue;
            acc = fmaf(dyo[((int64_t)oz * OH + oy) * OW + ox], wc[(dz * k + dyy) * k + dxx], acc);
          }
        }
      }
    }
    dx[(int64_t)n * xbs + (int64_t)c * S + ((int64_t)iz * H + iy) * W + ix] = acc;
  }
}

// dw[c,o,d] = sum_{n,iv} x[n,c,iv] * dy[n,o,iv*stride + d - pad]; one block per (c,o,tap)
__global__ __launch_bounds__(256) void convt_direct_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N, int Cin,
    int Cout, int D, int H, int W, int OD, int OH, int OW, int k, int stride, int pad, int64_t xbs,
    int64_t ybs) {
  __shared__ double scratch[4];
  const int k3 = k * k * k;
  int b = blockIdx.x;
  const int tap = b % k3;
  b /= k3;
  const int o = b % Cout;
  const int c = b / Cout;
  const int dz = tap / (k * k), dyy = (tap / k) % k, dxx = tap % k;
  const int64_t S = (int64_t)D * H * W;
  const int64_t OS = (int64_t)OD * OH * OW;
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* xc = x + (int64_t)n * xbs + (int64_t)c * S;
    const float* dyo = dy + (int64_t)n * ybs + (int64_t)o * OS;
    float part = 0.f;
    int cnt = 0;
    for (int64_t s = threadIdx.x; s < S; s += 256) {
      const int ix = (int)(s % W);
      const int iy = (int)((s / W) % H);
      const int iz = (int)(s / ((int64_t)W * H));
      const int oz = iz * stride + dz - pad, oy = iy * stride + dyy - pad,
                ox = ix * stride + dxx - pad;
      if (oz >= 0 && oz < OD && oy >= 0 && oy < OH && ox >= 0 && ox < OW)
        part = fmaf(xc[s], dyo[((int64_t)oz * OH + oy) * OW + ox], part);
      if (++cnt == 64) { acc += part; part = 0.f; cnt = 0; }
    }
    acc += part;
  }
  const double tot = block_sum<double, 256>(acc, scratch);
  if (threadIdx.x == 0) dw[((int64_t)c * Cout + o) * k3 + tap] = (float)tot;
}

static bool is_k2s2(const m355_conv3d_desc* d) {
  return d->k == 2 && d->stride == 2 && d->pad == 0 && d->out_pad == 0;
}
static int convt_out(int in, const m355_conv3d_desc* d) {
  return (in - 1) * d->stride - 2 * d->pad + d->k + d->out_pad;
}
static int convt_nsplit(const m355_conv3d_desc* d) {
  const int64_t tiles = ceil_div(d->Cin, 32) * ceil_div(d->Cout, 8);
  const int64_t nsteps = ceil_div((int64_t)d->N * d->D * d->H * d->W, 64);
  int64_t ns = std::max<int64_t>(1, 1024 / tiles);
  ns = std::min<int64_t>(ns, nsteps);
  return (int)ns;
}

static int validate_convt(const m355_conv3d_desc* d, const char* who) {
  M355_REQUIRE(d != nullptr, M355_EINVALID_ARG, "%s: null descriptor", who);
  M355_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->D > 0 && d->H > 0 && d->W > 0,
               M355_EINVALID_ARG, "%s: non-positive dimension", who);
  M355_REQUIRE(d->k >= 1 && d->k <= 7 && d->stride >= 1 && d->pad >= 0 && d->out_pad >= 0 &&
                   d->out_pad < d->stride,
               M355_EINVALID_ARG, "%s: bad k/stride/pad/out_pad", who);
  M355_REQUIRE(d->N <= 65535 && d->Cin <= 65535 && d->Cout <= 65535, M355_EUNSUPPORTED,
               "%s: N/Cin/Cout > 65535", who);
  return M355_OK;
}

}  // namespace m355

using namespace m355;

// shared with conv3d.hip
int launch_dbias(const float* dy, float* dbias, int N, int Cout, int64_t S, int64_t ybs, void* ws,
                 hipStream_t st);
static size_t convt_slab_bytes(const m355_conv3d_desc* d) {
  return is_k2s2(d) ? (size_t)round_up((int64_t)convt_nsplit(d) * d->Cin * d->Cout * 8 * 4, 256) : 0;
}
static size_t convt_dbias_bytes(const m355_conv3d_desc* d) {
  const int64_t OS = (int64_t)convt_out(d->D, d) * convt_out(d->H, d) * convt_out(d->W, d);
  return (size_t)round_up((int64_t)d->Cout * ceil_div(OS, 32768) * 8, 256);
}

extern "C" size_t m355_conv_transpose3d_workspace(const m355_conv3d_desc* d) {
  if (!d) return 0;
  return convt_slab_bytes(d) + convt_dbias_bytes(d);
}

extern "C" int m355_conv_transpose3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w,
                                         const float* bias, float* y, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_fwd")) return rc;
  M355_REQUIRE(x && w && y, M355_EINVALID_ARG, "conv_transpose3d_fwd: null pointer");
  (void)workspace; (void)workspace_bytes;
  hipStream_t st = (hipStream_t)stream;
  const int OD = convt_out(d->D, d), OH = convt_out(d->H, d), OW = convt_out(d->W, d);
  M355_REQUIRE(OD > 0 && OH > 0 && OW > 0, M355_EINVALID_ARG, "conv_transpose3d_fwd: empty output");
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)y & 7) == 0) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    dim3 grid((unsigned)ceil_div(S, 256), (unsigned)ceil_div(d->Cout, CT_OT), (unsigned)d->N);
    hipLaunchKernelGGL(convt_k2s2_fwd_kernel, grid, dim3(256), 0, st, x, w, bias, y, d->Cin, d->Cout,
                       d->D, d->H, d->W, xbs, ybs);
    return check_launch("convt_k2s2_fwd");
  }
  const int64_t total = (int64_t)d->N * d->Cout * OD * OH * OW;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 65535);
  hipLaunchKernelGGL(convt_direct_fwd_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, y, d->N,
                     d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride, d->pad, xbs, ybs);
  return check_launch("convt_direct_fwd");
}

extern "C" int m355_conv_transpose3d_bwd_data(const m355_conv3d_desc* d, const float* dy,
                                              const float* w, float* dx, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_bwd_data")) return rc;
  M355_REQUIRE(dy && w && dx, M355_EINVALID_ARG, "conv_transpose3d_bwd_data: null pointer");
  (void)workspace; (void)workspace_bytes;
  hipStream_t st = (hipStream_t)stream;
  const int OD = convt_out(d->D, d), OH = convt_out(d->H, d), OW = convt_out(d->W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)dy & 7) == 0) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    dim3 grid((unsigned)ceil_div(S, 256), (unsigned)ceil_div(d->Cin, CT_CT), (unsigned)d->N);
    hipLaunchKernelGGL(convt_k2s2_bwd_data_kernel, grid, dim3(256), 0, st, dy, w, dx, d->Cin,
                       d->Cout, d->D, d->H, d->W, xbs, ybs);
    return check_launch("convt_k2s2_bwd_data");
  }
  const int64_t total = (int64_t)d->N * d->Cin * d->D * d->H * d->W;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 65535);
  hipLaunchKernelGGL(convt_direct_bwd_data_kernel, dim3(blocks), dim3(256), 0, st, dy, w, dx, d->N,
                     d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride, d->pad, xbs, ybs);
  return check_launch("convt_direct_bwd_data");
}

extern "C" int m355_conv_transpose3d_bwd_weight(const m355_conv3d_desc* d, const float* x,
                                                const float* dy, float* dw, float* dbias,
                                                void* workspace, size_t workspace_bytes,
                                                void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_bwd_weight")) return rc;
  M355_REQUIRE(x && dy && dw, M355_EINVALID_ARG, "conv_transpose3d_bwd_weight: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OD = convt_out(d->D, d), OH = convt_out(d->H, d), OW = convt_out(d->W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)dy & 7) == 0) {
    const int nsplit = convt_nsplit(d);
    const size_t need = convt_slab_bytes(d);
    M355_REQUIRE(workspace && workspace_bytes >= need, M355_EWORKSPACE,
                 "conv_transpose3d_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes,
                 need);
    float* slab = (float*)workspace;
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(d->Cout, 8), (unsigned)ceil_div(d->Cin, 32));
    hipLaunchKernelGGL(convt_k2s2_bwd_weight_kernel, grid, dim3(256), 0, st, x, dy, slab, d->N,
                       d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, nsplit);
    const int64_t total = (int64_t)d->Cin * d->Cout * 8;
    hipLaunchKernelGGL(convt_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 1024)),
                       dim3(256), 0, st, slab, dw, total, nsplit);
  } else {
    const int k3 = d->k * d->k * d->k;
    const int64_t nblk = (int64_t)d->Cin * d->Cout * k3;
    M355_REQUIRE(nblk < (1ll << 31), M355_EUNSUPPORTED, "conv_transpose3d_bwd_weight: grid too large");
    hipLaunchKernelGGL(convt_direct_bwd_weight_kernel, dim3((unsigned)nblk), dim3(256), 0, st, x, dy,
                       dw, d->N, d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride,
                       d->pad, xbs, ybs);
  }
  if (dbias) {
    M355_REQUIRE(workspace && workspace_bytes >= convt_slab_bytes(d) + convt_dbias_bytes(d), M355_EWORKSPACE,
                 "conv_transpose3d_bwd_weight: workspace too small for the bias gradient");
    launch_dbias(dy, dbias, d->N, d->Cout, (int64_t)OD * OH * OW, ybs, (char*)workspace + convt_slab_bytes(d), st);
  }
  return check_launch("conv_transpose3d_bwd_weight");
}
