// ConvTranspose3d for gfx950.
//
// Reference ops replaced: nn.ConvTranspose3d(kernel_size=2, stride=2) reached
// through ModularUNet's upsample_class hook (models/modular_unet.py:20-21,72-81,96)
// and the F.conv_transpose3d of BlurConvTranspose3d (models/components.py:152,
// effective k=4, s=2, p=1).
//
// k=2,s=2,p=0 is a non-overlapping scatter: every input voxel produces its own
// 2x2x2 output block, i.e. a [8*Cout x Cin] x [Cin x voxels] GEMM with AI ~ 28
// flop/B in fp32 -> HBM-bound.  The three k2s2 ops run as
// fp32-MFMA GEMMs with the voxel on the lane (coalesced x reads, float2-coalesced
// y / dy accesses); any other geometry takes the generic direct kernels below.
#include "common.hpp"

namespace m355 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =============================== k2s2 as fp32-MFMA GEMMs ===============================
// Every input voxel owns its 2x2x2 output block, so with t = a*4 + b*2 + c (the position
// inside the block) and m = o*8 + t:
//   fwd        Y[m, v]  = bias[o] + sum_ci W[ci, m] * X[ci, v]          M = 8*Cout, K = Cin
//   bwd-data   dX[ci,v] =           sum_m  W[ci, m] * dYr[m, v]         M = Cin,    K = 8*Cout
//   bwd-weight dW[ci,m] =           sum_v  X[ci, v] * dYr[m, v]         M = Cin, N = 8*Cout, K = voxels
// where dYr[m, v] = dY[o, 2z+a, 2y+b, 2x+c] is the output block gathered per input voxel and
// the torch weight layout [Cin][Cout][2][2][2] is already W[ci][m] with m contiguous.
// All three are HBM-bound (AI ~ 28 flop/B); the MFMA only has to keep up with the stream.
// Small accumulator footprints -> several workgroups per CU hide the synchronous staging.

// ---- forward: workgroup = 256 consecutive input voxels (wave w: 2 groups of 32), loops over
// the 32-row m-tiles (4 output channels each).  The x tile [KC ci][256] is staged once when
// Cin <= KC.  C/D layout puts t = (r&3) + 4*half on the lane's registers, so (r, r+1) is the
// (c=0, c=1) pair of one (o, a, b): float2 stores, 256 B contiguous per 32 lanes.
constexpr int CTF_KC = 64;
constexpr int CTF_MTG = 4;  // m-tiles (of 4 output channels) per workgroup
__global__ __launch_bounds__(256) void convt_k2s2_fwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs) {
  constexpr int NVT = 256, KC = CTF_KC;
  __shared__ float xs[KC * NVT];
  __shared__ float ws[KC * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int n = blockIdx.z;
  const float* xn = x + (int64_t)n * xbs;
  float* yn = y + (int64_t)n * ybs;
  const int nchunks = (Cin + KC - 1) / KC;
  const int mtiles = (Cout + 3) / 4;
  // blockIdx.y owns CTF_MTG consecutive m-tiles (16 output channels): more workgroups in flight,
  // and concurrent workgroups write different output planes
  const int mt_begin = blockIdx.y * CTF_MTG, mt_end = min(mtiles, mt_begin + CTF_MTG);
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;

  // output coordinates of this lane's two voxels
  int64_t obase[2];
  bool vok[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int v = v0 + (wave * 2 + g) * 32 + l32;
    vok[g] = v < S;
    const int vv = vok[g] ? v : 0;
    const int ix = vv % W, iy = (vv / W) % H, iz = vv / (W * H);
    obase[g] = ((int64_t)(2 * iz + half) * OH + 2 * iy) * OW + 2 * ix;  // a = half
  }

  for (int mt = mt_begin; mt < mt_end; ++mt) {
    const int o0 = mt * 4;
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch * KC;
      const int kc = min(KC, Cin - c0);
      __syncthreads();
      if (nchunks > 1 || mt == mt_begin) {
        const int v = v0 + tid;
        const bool vin = v < S;
        const float* xp = xn + (int64_t)c0 * S + (vin ? v : 0);
        // unconditional loads (clamped), all KC rows written (rows >= kc are zero)
#pragma unroll 16
        for (int c = 0; c < KC; ++c) {
          const bool ok = vin && c < kc;
          const float val = xp[ok ? (int64_t)c * S : 0];
          xs[c * NVT + tid] = ok ? val : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < KC * 32 / 256; ++j) {
        const int i = tid + 256 * j;
        const int c = i >> 5, m = i & 31;
        const int o = o0 + (m >> 3);
        const bool ok = c < kc && o < Cout;
        const float val = w[ok ? ((int64_t)(c0 + c) * Cout + o) * 8 + (m & 7) : 0];
        ws[i] = ok ? val : 0.f;
      }
      __syncthreads();
      const float* wb = ws + half * 32 + l32;
      const float* xb = xs + half * NVT + wave * 64 + l32;
      const int kce = (kc + 7) & ~7;  // rows up to KC are zero-filled
#pragma unroll 1
      for (int k = 0; k < kce; k += 8) {
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
          const float a = wb[(k + kk) * 32];
#pragma unroll
          for (int g = 0; g < 2; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[(k + kk) * NVT + g * 32], acc[g], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (!vok[g]) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // q = output channel inside the tile (r >> 2)
        const int o = o0 + q;
        if (o >= Cout) continue;
        const float bv = bias ? bias[o] : 0.f;
        float* yo = yn + (int64_t)o * OS + obase[g];
#pragma unroll
        for (int b = 0; b < 2; ++b)
          *reinterpret_cast<float2*>(yo + (int64_t)b * OW) =
              make_float2(acc[g][q * 4 + b * 2] + bv, acc[g][q * 4 + b * 2 + 1] + bv);
      }
    }
  }
}

// ---- data gradient: workgroup = 256 input voxels x 64 input channels (2 m-tiles, so dY is read
// once for Cin <= 64); K is walked 4 output channels (32 k) at a time.  dY is de-interleaved
// while staging: dys[(o,t)][v], the k-pair is (c=0, c=1) of one (o,a,b).
__global__ __launch_bounds__(256) void convt_k2s2_bwd_data_mfma_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int Cin,
    int Cout, int D, int H, int W, int64_t xbs, int64_t ybs) {
  constexpr int NVT = 256, KO = 4, KK = KO * 8;
  __shared__ float dys[KK * NVT];
  __shared__ float ws[KK * 65];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int c0 = blockIdx.y * 64;
  const int n = blockIdx.z;
  const float* dyn = dy + (int64_t)n * ybs;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  // this thread's staging voxel
  const int sv = v0 + tid;
  const bool sok = sv < S;
  const int svv = sok ? sv : 0;
  const int64_t sbase = ((int64_t)(2 * (svv / (W * H))) * OH + 2 * ((svv / W) % H)) * OW + 2 * (svv % W);

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.f;

  for (int o0 = 0; o0 < Cout; o0 += KO) {
    __syncthreads();
#pragma unroll
    for (int o = 0; o < KO; ++o)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        float2 val = make_float2(0.f, 0.f);
        if (sok && o0 + o < Cout)
          val = *reinterpret_cast<const float2*>(dyn + (int64_t)(o0 + o) * OS + sbase +
                                                 (int64_t)(ab >> 1) * OH * OW + (int64_t)(ab & 1) * OW);
        dys[(o * 8 + ab * 2) * NVT + tid] = val.x;
        dys[(o * 8 + ab * 2 + 1) * NVT + tid] = val.y;
      }
    // weights of this k-slab, transposed to ws[k][ci] (row stride 65: conflict-free both ways)
    for (int i = tid; i < 64 * KK; i += 256) {
      const int k = i & (KK - 1), c = i >> 5;  // KK == 32
      const int o = o0 + (k >> 3);
      ws[k * 65 + c] = (c0 + c < Cin && o < Cout) ? w[((int64_t)(c0 + c) * Cout + o) * 8 + (k & 7)] : 0.f;
    }
    __syncthreads();
    const float* wb = ws + half * 65 + l32;
    const float* db = dys + half * NVT + wave * 64 + l32;
#pragma unroll 4
    for (int k = 0; k < KK; k += 2) {
      const float a0 = wb[k * 65], a1 = wb[k * 65 + 32];
      const float b0 = db[k * NVT], b1 = db[k * NVT + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float* dxn = dx + (int64_t)n * xbs;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int v = v0 + (wave * 2 + g) * 32 + l32;
    if (v >= S) continue;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < Cin) dxn[(int64_t)c * S + v] = acc[m][g][r];
      }
  }
}

// ---- weight gradient: workgroup = 64 input channels (2 m-tiles) x 16 output channels (wave w:
// o-group w, N-tile = 4 o x 8 t), persistent over 64-voxel tiles of its split; partial
// dW -> slab[split][Cin][Cout][8] (coalesced: the lane index IS (o,t)), fixed-order reduce.
__global__ __launch_bounds__(256) void convt_k2s2_bwd_weight_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N,
    int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs, int nsplit) {
  constexpr int NV = 64;
  __shared__ float xs[64 * (NV + 1)];
  __shared__ float dys[128 * (NV + 1)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int split = blockIdx.x, o0 = blockIdx.y * 16, c0 = blockIdx.z * 64;
  const int S = D * H * W;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  const int64_t total = (int64_t)N * S;
  const int64_t ntiles = (total + NV - 1) / NV;

  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const int sv = tid & 63, sq = tid >> 6;  // staging: voxel, quarter
  for (int64_t tile = split; tile < ntiles; tile += nsplit) {
    const int64_t g = tile * NV + sv;
    const bool ok = g < total;
    const int64_t nn = ok ? g / S : 0;
    const int v = ok ? (int)(g - nn * S) : 0;
    __syncthreads();
#pragma unroll 4
    for (int c = sq; c < 64; c += 4)
      xs[c * (NV + 1) + sv] = (ok && c0 + c < Cin) ? x[nn * xbs + (int64_t)(c0 + c) * S + v] : 0.f;
    const int64_t sbase = nn * ybs + ((int64_t)(2 * (v / (W * H))) * OH + 2 * ((v / W) % H)) * OW + 2 * (v % W);
#pragma unroll 4
    for (int i = sq; i < 64; i += 4) {  // i = o_local*4 + ab
      const int o = o0 + (i >> 2), ab = i & 3;
      float2 val = make_float2(0.f, 0.f);
      if (ok && o < Cout)
        val = *reinterpret_cast<const float2*>(dy + sbase + (int64_t)o * OS + (int64_t)(ab >> 1) * OH * OW +
                                               (int64_t)(ab & 1) * OW);
      dys[((i >> 2) * 8 + ab * 2) * (NV + 1) + sv] = val.x;
      dys[((i >> 2) * 8 + ab * 2 + 1) * (NV + 1) + sv] = val.y;
    }
    __syncthreads();
    const float* xb = xs + l32 * (NV + 1) + half;
    const float* db = dys + (wave * 32 + l32) * (NV + 1) + half;
#pragma unroll 8
    for (int k = 0; k < NV; k += 2) {
      const float b = db[k];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[k], b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[32 * (NV + 1) + k], b, acc[1], 0, 0, 0);
    }
  }
  // D[i = ci][j = (o,t)]
  const int o = o0 + wave * 4 + (l32 >> 3);
  if (o < Cout) {
    float* sl = slab + (int64_t)split * Cin * Cout * 8;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < Cin) sl[((int64_t)c * Cout + o) * 8 + (l32 & 7)] = acc[m][r];
      }
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
  const int64_t tiles = ceil_div(d->Cin, 64) * ceil_div(d->Cout, 16);
  const int64_t nsteps = ceil_div((int64_t)d->N * d->D * d->H * d->W, 64);
  int64_t ns = std::max<int64_t>(1, 768 / tiles);
  ns = std::min<int64_t>(ns, nsteps);
  return (int)ns;
}
static bool convt_fits_i32(const m355_conv3d_desc* d) {
  return (int64_t)d->D * d->H * d->W * 8 * std::max(d->Cin, d->Cout) < (1ll << 31);
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
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)y & 7) == 0 && convt_fits_i32(d)) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    dim3 grid((unsigned)ceil_div(S, 256), (unsigned)ceil_div(ceil_div(d->Cout, 4), CTF_MTG), (unsigned)d->N);
    hipLaunchKernelGGL(convt_k2s2_fwd_mfma_kernel, grid, dim3(256), 0, st, x, w, bias, y, d->Cin,
                       d->Cout, d->D, d->H, d->W, xbs, ybs);
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
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)dy & 7) == 0 && convt_fits_i32(d)) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    dim3 grid((unsigned)ceil_div(S, 256), (unsigned)ceil_div(d->Cin, 64), (unsigned)d->N);
    hipLaunchKernelGGL(convt_k2s2_bwd_data_mfma_kernel, grid, dim3(256), 0, st, dy, w, dx, d->Cin,
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
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)dy & 7) == 0 && convt_fits_i32(d)) {
    const int nsplit = convt_nsplit(d);
    const size_t need = convt_slab_bytes(d);
    M355_REQUIRE(workspace && workspace_bytes >= need, M355_EWORKSPACE,
                 "conv_transpose3d_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes,
                 need);
    float* slab = (float*)workspace;
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(d->Cout, 16), (unsigned)ceil_div(d->Cin, 64));
    hipLaunchKernelGGL(convt_k2s2_bwd_weight_mfma_kernel, grid, dim3(256), 0, st, x, dy, slab, d->N,
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
