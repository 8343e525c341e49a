// HBM-bound streaming kernels: AvgPool3d(2,2), trilinear x2 upsampling
// (align_corners=True), channel softmax, channel-slice copy (concat), channel
// scale (Dropout3d) and add.
//
// Reference ops replaced: nn.AvgPool3d / nn.Upsample / nn.Softmax defaults of
// ModularUNet (models/modular_unet.py:20-26,39-46,92,96,100), torch.cat
// (modular_unet.py:97), Dropout3d and the residual add of Block3d
// (models/components.py:58-60,67-71), StochasticMatrix softmax (components.py:170-185).
#include "common.hpp"
#include "h16.hpp"

namespace m355 {

static inline unsigned grid_for(int64_t work, int per_block = 256, int64_t cap = 8192) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work, per_block), cap));
}

// ------------------------------------------------------------------ avg pool
// torch accumulates the window in (z, y, x) order and divides by the window size.
template <bool VEC>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int N, int C,
                                                           int D, int H, int W, int64_t xbs,
                                                           int64_t ybs) {
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int OWV = VEC ? OW / 2 : OW;  // VEC: two outputs per thread
  const int64_t total = (int64_t)N * C * OD * OH * OWV;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ox = (int)(i % OWV);
    int64_t r = i / OWV;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    r /= OD;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const float* xp = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
    float* yp = y + (int64_t)n * ybs + (int64_t)c * OD * OH * OW;
    const int64_t r00 = ((int64_t)(2 * oz) * H + 2 * oy) * W;
    const int64_t r01 = r00 + W, r10 = r00 + (int64_t)H * W, r11 = r10 + W;
    if (VEC) {
      const float4 a = *reinterpret_cast<const float4*>(xp + r00 + 4 * ox);
      const float4 b = *reinterpret_cast<const float4*>(xp + r01 + 4 * ox);
      const float4 cc = *reinterpret_cast<const float4*>(xp + r10 + 4 * ox);
      const float4 d = *reinterpret_cast<const float4*>(xp + r11 + 4 * ox);
      float2 o;
      o.x = (((((((a.x + a.y) + b.x) + b.y) + cc.x) + cc.y) + d.x) + d.y) * 0.125f;
      o.y = (((((((a.z + a.w) + b.z) + b.w) + cc.z) + cc.w) + d.z) + d.w) * 0.125f;
      *reinterpret_cast<float2*>(yp + ((int64_t)oz * OH + oy) * OW + 2 * ox) = o;
    } else {
      const int xi = 2 * ox;
      const float s = ((((((xp[r00 + xi] + xp[r00 + xi + 1]) + xp[r01 + xi]) + xp[r01 + xi + 1]) +
                         xp[r10 + xi]) + xp[r10 + xi + 1]) + xp[r11 + xi]) + xp[r11 + xi + 1];
      yp[((int64_t)oz * OH + oy) * OW + ox] = s * 0.125f;
    }
  }
}

// add (may be null): a second gradient of the pooled tensor's INPUT, summed in the same pass -- the skip
// connection's gradient (models/modular_unet.py:90-92: the block output feeds both the pool and the concat)
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int C,
                                                           int D, int H, int W, int64_t dybs,
                                                           int64_t dxbs, const float* __restrict__ add, int64_t abs_) {
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int W2 = W / 2;  // one thread per x pair
  const int64_t total = (int64_t)N * C * D * H * W2;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int xp = (int)(i % W2);
    int64_t r = i / W2;
    const int iy = (int)(r % H);
    r /= H;
    const int iz = (int)(r % D);
    r /= D;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const float g = dy[(int64_t)n * dybs + (int64_t)c * OD * OH * OW +
                       ((int64_t)(iz / 2) * OH + iy / 2) * OW + xp] * 0.125f;
    const int64_t sp = (int64_t)c * D * H * W + ((int64_t)iz * H + iy) * W + 2 * xp;
    float* o = dx + (int64_t)n * dxbs + sp;
    if (add) {
      const float* a = add + (int64_t)n * abs_ + sp;
      o[0] = a[0] + g;
      o[1] = a[1] + g;
    } else {
      o[0] = g;
      o[1] = g;
    }
  }
}

// --------------------------------------------------------- trilinear upsample
// aten/src/ATen/native/UpSample.h semantics for align_corners=True:
//   ratio = (in-1)/(out-1) (float), src = ratio*o, i0 = min(floor(src), in-1),
//   l1 = clamp(src - i0, 0, 1), l0 = 1 - l1, i1 = i0 + (i0 < in-1)
struct Lin {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ float lin_ratio(int in, int out) {
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
}
// `ratio` = lin_ratio(in, out): a correctly rounded division is ~12 instructions, and the backward evaluates 24
// coordinates per element
__device__ __forceinline__ Lin lin_coord(int o, int in, int out, float ratio) {
  Lin r;
  if (in == out) {
    r.i0 = r.i1 = o; r.l0 = 1.f; r.l1 = 0.f;
    return r;
  }
  // rounded product, then rounded difference, as ATen computes them: contracted into one fma the weight differs by up
  // to half an ulp of src (4e-6 at index 64) and the interpolated value by that times the local slope
  // (HIP's __fmul_rn / __fsub_rn are plain operators and contract like any other: the pragma is what keeps them apart)
  float src, frac;
  {
#pragma clang fp contract(off)
    src = ratio * (float)o;
    r.i0 = min((int)floorf(src), in - 1);
    frac = src - (float)r.i0;
  }
  r.l1 = fminf(fmaxf(frac, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  return r;
}
__device__ __forceinline__ Lin lin_coord(int o, int in, int out) { return lin_coord(o, in, out, lin_ratio(in, out)); }

__global__ __launch_bounds__(256) void trilinear2_fwd_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, int N, int C,
                                                             int D, int H, int W, int64_t xbs,
                                                             int64_t ybs) {
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t total = (int64_t)N * C * OD * OH * OW;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    r /= OD;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const Lin lz = lin_coord(oz, D, OD), ly = lin_coord(oy, H, OH), lx = lin_coord(ox, W, OW);
    const float* xp = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
    auto row = [&](int z, int yy) {
      const float* p = xp + ((int64_t)z * H + yy) * W;
      return lx.l0 * p[lx.i0] + lx.l1 * p[lx.i1];
    };
    const float v0 = ly.l0 * row(lz.i0, ly.i0) + ly.l1 * row(lz.i0, ly.i1);
    const float v1 = ly.l0 * row(lz.i1, ly.i0) + ly.l1 * row(lz.i1, ly.i1);
    y[(int64_t)n * ybs + (int64_t)c * OD * OH * OW + ((int64_t)oz * OH + oy) * OW + ox] =
        lz.l0 * v0 + lz.l1 * v1;
  }
}

// Four consecutive outputs of one output row per thread: index decomposition, z / y coordinates and the four row
// pointers are shared, the result leaves as one float4.  The same expressions per output as trilinear2_fwd_kernel.
// (the one-output-per-thread kernel spent ~130 lane-cycles per output on div / mod chains and scalar stores: 1.2 TB/s
// on the upsampling levels of NestedResUNet, models/nested_residual_unet.py:58)
__global__ __launch_bounds__(256) void trilinear2_fwd_q_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, int N, int C,
                                                               int D, int H, int W, int64_t xbs,
                                                               int64_t ybs) {
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W, OW4 = OW >> 2;
  const int64_t total = (int64_t)N * C * OD * OH * OW4;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int q = (int)(i % OW4);
    unsigned r = (unsigned)(i / OW4);            // (host: N*C*OD*OH < 2^31)
    const int oy = (int)(r % (unsigned)OH);
    r /= (unsigned)OH;
    const int oz = (int)(r % (unsigned)OD);
    r /= (unsigned)OD;
    const int c = (int)(r % (unsigned)C);
    const int n = (int)(r / (unsigned)C);
    const Lin lz = lin_coord(oz, D, OD), ly = lin_coord(oy, H, OH);
    const float* xp = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
    const float* p00 = xp + ((int64_t)lz.i0 * H + ly.i0) * W;
    const float* p01 = xp + ((int64_t)lz.i0 * H + ly.i1) * W;
    const float* p10 = xp + ((int64_t)lz.i1 * H + ly.i0) * W;
    const float* p11 = xp + ((int64_t)lz.i1 * H + ly.i1) * W;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Lin lx = lin_coord(4 * q + k, W, OW);
      const float r00 = lx.l0 * p00[lx.i0] + lx.l1 * p00[lx.i1];
      const float r01 = lx.l0 * p01[lx.i0] + lx.l1 * p01[lx.i1];
      const float r10 = lx.l0 * p10[lx.i0] + lx.l1 * p10[lx.i1];
      const float r11 = lx.l0 * p11[lx.i0] + lx.l1 * p11[lx.i1];
      const float v0 = ly.l0 * r00 + ly.l1 * r01;
      const float v1 = ly.l0 * r10 + ly.l1 * r11;
      o[k] = lz.l0 * v0 + lz.l1 * v1;
    }
    *reinterpret_cast<float4*>(y + (int64_t)n * ybs + (int64_t)c * OD * OH * OW + ((int64_t)oz * OH + oy) * OW + 4 * q) =
        make_float4(o[0], o[1], o[2], o[3]);
  }
}

// The same through LDS: a workgroup owns 4 output planes x 16 output rows x all columns of one (n, c) volume, stages the
// <= 4 x 10 input rows they read (coalesced) and takes its eight taps per output from LDS instead of eight scattered
// global loads -- the quad kernel above is load-instruction-bound (32 L1 loads per thread, 1.9 TB/s of output).
// Same expressions per output as the other two forward kernels.
constexpr int TRI_TZ = 4, TRI_TY = 16, TRI_PZ = 4, TRI_PY = 10;
__global__ __launch_bounds__(256) void trilinear2_fwd_lds_kernel(const float* __restrict__ x,
                                                                 float* __restrict__ y, int C, int D, int H,
                                                                 int W, int64_t xbs, int64_t ybs) {
  extern __shared__ float tri_lds[];   // [TRI_PZ][TRI_PY][W]
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W, OW4 = OW >> 2;
  const int nc = blockIdx.z, n = nc / C, c = nc % C;
  const int oz0 = blockIdx.y * TRI_TZ, oy0 = blockIdx.x * TRI_TY;
  const int zlo = lin_coord(oz0, D, OD).i0, ylo = lin_coord(oy0, H, OH).i0;
  const float* xp = x + (int64_t)n * xbs + (int64_t)c * D * H * W;
  // rows past the volume are clamped duplicates (never selected: the coordinates clamp the same way)
  for (int i = threadIdx.x; i < TRI_PZ * TRI_PY * W; i += 256) {
    const int xx = i % W, r = i / W;
    const int py = r % TRI_PY, pz = r / TRI_PY;
    tri_lds[i] = xp[((int64_t)min(zlo + pz, D - 1) * H + min(ylo + py, H - 1)) * W + xx];
  }
  __syncthreads();
  float* yp = y + (int64_t)n * ybs + (int64_t)c * OD * OH * OW;
  for (int i = threadIdx.x; i < TRI_TZ * TRI_TY * OW4; i += 256) {
    const int q = i % OW4, r = i / OW4;
    const int oy = oy0 + r % TRI_TY, oz = oz0 + r / TRI_TY;
    if (oy >= OH || oz >= OD) continue;
    const Lin lz = lin_coord(oz, D, OD), ly = lin_coord(oy, H, OH);
    const float* p00 = tri_lds + ((lz.i0 - zlo) * TRI_PY + (ly.i0 - ylo)) * W;
    const float* p01 = tri_lds + ((lz.i0 - zlo) * TRI_PY + (ly.i1 - ylo)) * W;
    const float* p10 = tri_lds + ((lz.i1 - zlo) * TRI_PY + (ly.i0 - ylo)) * W;
    const float* p11 = tri_lds + ((lz.i1 - zlo) * TRI_PY + (ly.i1 - ylo)) * W;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Lin lx = lin_coord(4 * q + k, W, OW);
      const float r00 = lx.l0 * p00[lx.i0] + lx.l1 * p00[lx.i1];
      const float r01 = lx.l0 * p01[lx.i0] + lx.l1 * p01[lx.i1];
      const float r10 = lx.l0 * p10[lx.i0] + lx.l1 * p10[lx.i1];
      const float r11 = lx.l0 * p11[lx.i0] + lx.l1 * p11[lx.i1];
      const float v0 = ly.l0 * r00 + ly.l1 * r01;
      const float v1 = ly.l0 * r10 + ly.l1 * r11;
      o[k] = lz.l0 * v0 + lz.l1 * v1;
    }
    *reinterpret_cast<float4*>(yp + ((int64_t)oz * OH + oy) * OW + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// weight with which output o (size out) reads input i (size in)
__device__ __forceinline__ float lin_weight(int o, int i, int in, int out, float ratio) {
  if (o < 0 || o >= out) return 0.f;
  const Lin l = lin_coord(o, in, out, ratio);
  return (l.i0 == i ? l.l0 : 0.f) + (l.i1 == i ? l.l1 : 0.f);
}

// gather-form backward (deterministic): every output index that can read input i
// lies in [2i-3, 2i+4] for out = 2*in (trilinear backward).
__global__ __launch_bounds__(256) void trilinear2_bwd_kernel(const float* __restrict__ dy,
                                                             float* __restrict__ dx, int N, int C,
                                                             int D, int H, int W, int64_t dybs,
                                                             int64_t dxbs) {
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t total = (int64_t)N * C * D * H * W;
  const float rz = lin_ratio(D, OD), ry = lin_ratio(H, OH), rx = lin_ratio(W, OW);
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ix = (int)(i % W);
    int64_t r = i / W;
    const int iy = (int)(r % H);
    r /= H;
    const int iz = (int)(r % D);
    r /= D;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const float* dp = dy + (int64_t)n * dybs + (int64_t)c * OD * OH * OW;
    float wx[8], wy[8];   // (the y weights once per element, not once per z tap: 24 weight evaluations instead of 80)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wx[k] = lin_weight(2 * ix - 3 + k, ix, W, OW, rx);
      wy[k] = lin_weight(2 * iy - 3 + k, iy, H, OH, ry);
    }
    float acc = 0.f;
    for (int kz = 0; kz < 8; ++kz) {
      const int oz = 2 * iz - 3 + kz;
      const float wz = lin_weight(oz, iz, D, OD, rz);
      if (wz == 0.f) continue;
#pragma unroll
      for (int ky = 0; ky < 8; ++ky) {
        if (wy[ky] == 0.f) continue;
        const int oy = 2 * iy - 3 + ky;
        const float* row = dp + ((int64_t)oz * OH + oy) * OW;
        float racc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ox = 2 * ix - 3 + k;
          if (wx[k] != 0.f) racc = fmaf(wx[k], row[ox], racc);
        }
        acc = fmaf(wz * wy[ky], racc, acc);
      }
    }
    dx[(int64_t)n * dxbs + (int64_t)c * D * H * W + ((int64_t)iz * H + iy) * W + ix] = acc;
  }
}

// -------------------------------------------------------------------- softmax
// x viewed as [N, C, inner, S]; softmax over C.  One thread per (n, inner, s).
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x,
                                                          float* __restrict__ y, int N, int C,
                                                          int inner, int64_t S, float diag_bias) {
  const int64_t total = (int64_t)N * inner * S;
  const int64_t cstride = (int64_t)inner * S;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t s = i % S;
    const int64_t r = i / S;
    const int in = (int)(r % inner);
    const int n = (int)(r / inner);
    const int64_t base = ((int64_t)n * C * inner + in) * S + s;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
      float v = x[base + c * cstride];
      if (inner > 1 && c == in) v += diag_bias;
      mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
      float v = x[base + c * cstride];
      if (inner > 1 && c == in) v += diag_bias;
      sum += expf(v - mx);
    }
    const float inv = 1.f / sum;
    for (int c = 0; c < C; ++c) {
      float v = x[base + c * cstride];
      if (inner > 1 && c == in) v += diag_bias;
      y[base + c * cstride] = expf(v - mx) * inv;
    }
  }
}

// dx_c = y_c * (dy_c - sum_j y_j dy_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dx, int N, int C,
                                                          int inner, int64_t S) {
  const int64_t total = (int64_t)N * inner * S;
  const int64_t cstride = (int64_t)inner * S;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t s = i % S;
    const int64_t r = i / S;
    const int in = (int)(r % inner);
    const int n = (int)(r / inner);
    const int64_t base = ((int64_t)n * C * inner + in) * S + s;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot = fmaf(y[base + c * cstride], dy[base + c * cstride], dot);
    for (int c = 0; c < C; ++c)
      dx[base + c * cstride] = y[base + c * cstride] * (dy[base + c * cstride] - dot);
  }
}

// ------------------------------------------------------- copy / scale / add
template <bool VEC>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src,
                                                            float* __restrict__ dst, int N,
                                                            int64_t CS, int64_t sbs, int64_t dbs) {
  // CS = C*S contiguous floats per sample
  const int64_t per = VEC ? CS / 4 : CS;
  const int64_t total = (int64_t)N * per;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t n = i / per, k = i - n * per;
    if (VEC)
      reinterpret_cast<float4*>(dst + n * dbs)[k] = reinterpret_cast<const float4*>(src + n * sbs)[k];
    else
      dst[n * dbs + k] = src[n * sbs + k];
  }
}

__global__ __launch_bounds__(256) void channel_scale_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ scale,
                                                            float* __restrict__ y, int64_t NC,
                                                            int64_t S) {
  const int64_t total = NC * S;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll)
    y[i] = x[i] * scale[i / S];
}

template <bool VEC>
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a,
                                                  const float* __restrict__ b,
                                                  float* __restrict__ y, int64_t n) {
  const int64_t total = VEC ? n / 4 : n;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    if (VEC) {
      const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
      reinterpret_cast<float4*>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    } else {
      y[i] = a[i] + b[i];
    }
  }
}

// ---------------------------------------------- space-to-depth / depth-to-space (factor 2)
// TO_DEPTH: full[n,c,2z+pz,2y+py,2x+px] -> packed[n,c*8+p,z,y,x]; otherwise the inverse.
// One thread per (row pair of the full tensor): reads/writes float2 on the full-resolution side.
template <bool TO_DEPTH>
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                                  int C, int D, int H, int W, int64_t fbs, int64_t pbs) {
  // D,H,W: full resolution.  fbs / pbs: batch strides of the full / packed tensors.
  const int HD = D / 2, HH = H / 2, HW2 = W / 2;
  const int64_t total = (int64_t)N * C * D * H * HW2;  // one float2 of the full tensor per thread
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int xh = (int)(i % HW2);
    int64_t r = i / HW2;
    const int fy = (int)(r % H);
    r /= H;
    const int fz = (int)(r % D);
    r /= D;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const int64_t fidx = (int64_t)n * fbs + (((int64_t)c * D + fz) * H + fy) * W + 2 * xh;
    const int p0 = ((fz & 1) * 4 + (fy & 1) * 2);
    const int64_t pidx = (int64_t)n * pbs + ((((int64_t)c * 8 + p0) * HD + (fz >> 1)) * HH + (fy >> 1)) * HW2 + xh;
    const int64_t pstep = (int64_t)HD * HH * HW2;  // px = 1 is the next packed channel
    if (TO_DEPTH) {
      const float2 v = *reinterpret_cast<const float2*>(src + fidx);
      dst[pidx] = v.x;
      dst[pidx + pstep] = v.y;
    } else {
      *reinterpret_cast<float2*>(dst + fidx) = make_float2(src[pidx], src[pidx + pstep]);
    }
  }
}

// ------------------------------------------------- trilinear upsample / channel scale on the c8 layout (h16.hpp)
// NestedResUNet (models/nested_residual_unet.py:74,92-101) in a 16-bit precision mode: activations and their gradients
// only exist as c8 items (8 channels of a voxel = 16 bytes), so a thread computes ONE output voxel for 8 channels --
// index decomposition and interpolation weights are shared by the 8 -- in fp32 with the same expressions as
// trilinear2_fwd_kernel, and rounds once.  grid-stride over N * CB * output voxels.
template <typename HT>
__global__ __launch_bounds__(256) void trilinear2_fwd_c8_kernel(const HT* __restrict__ x16, HT* __restrict__ y16, int N,
                                                                int CB, int D, int H, int W, int64_t xbs, int64_t ybs) {
  using hx8 = typename H16<HT>::x8;
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t S = (int64_t)D * H * W, OS = S * 8;
  const int64_t total = (int64_t)N * CB * OS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    r /= OD;
    const int cb = (int)(r % CB);
    const int n = (int)(r / CB);
    const Lin lz = lin_coord(oz, D, OD), ly = lin_coord(oy, H, OH), lx = lin_coord(ox, W, OW);
    const hx8* xp = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs) + (int64_t)cb * S;
    float row[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const hx8* p = xp + ((int64_t)((q & 2) ? lz.i1 : lz.i0) * H + ((q & 1) ? ly.i1 : ly.i0)) * W;
      const hx8 a = p[lx.i0], b = p[lx.i1];
#pragma unroll
      for (int j = 0; j < 8; ++j) row[q][j] = lx.l0 * (float)a[j] + lx.l1 * (float)b[j];
    }
    hx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v0 = ly.l0 * row[0][j] + ly.l1 * row[1][j];
      const float v1 = ly.l0 * row[2][j] + ly.l1 * row[3][j];
      o[j] = (HT)(lz.l0 * v0 + lz.l1 * v1);
    }
    (reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs) + (int64_t)cb * OS)[((int64_t)oz * OH + oy) * OW + ox] = o;
  }
}

// gather-form backward on c8 gradients (deterministic; window as in trilinear2_bwd_kernel): fp32 sums, one rounding
// (saturating for fp16, whose gradients carry the loss scale).
template <typename HT>
__global__ __launch_bounds__(256) void trilinear2_bwd_c8_kernel(const HT* __restrict__ dy16, HT* __restrict__ dx16, int N,
                                                                int CB, int D, int H, int W, int64_t dybs, int64_t dxbs,
                                                                int* __restrict__ oflag) {
  using hx8 = typename H16<HT>::x8;
  bool sat = false;
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t S = (int64_t)D * H * W, OS = S * 8;
  const int64_t total = (int64_t)N * CB * S;
  const float rz = lin_ratio(D, OD), ry = lin_ratio(H, OH), rx = lin_ratio(W, OW);
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ix = (int)(i % W);
    int64_t r = i / W;
    const int iy = (int)(r % H);
    r /= H;
    const int iz = (int)(r % D);
    r /= D;
    const int cb = (int)(r % CB);
    const int n = (int)(r / CB);
    const hx8* dp = reinterpret_cast<const hx8*>(dy16 + (int64_t)n * dybs) + (int64_t)cb * OS;
    float wx[8], wy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wx[k] = lin_weight(2 * ix - 3 + k, ix, W, OW, rx);
      wy[k] = lin_weight(2 * iy - 3 + k, iy, H, OH, ry);
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kz = 0; kz < 8; ++kz) {
      const int oz = 2 * iz - 3 + kz;
      const float wz = lin_weight(oz, iz, D, OD, rz);
      if (wz == 0.f) continue;
#pragma unroll
      for (int ky = 0; ky < 8; ++ky) {
        if (wy[ky] == 0.f) continue;
        const int oy = 2 * iy - 3 + ky;
        const hx8* row = dp + ((int64_t)oz * OH + oy) * OW;
        const float wzy = wz * wy[ky];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (wx[k] == 0.f) continue;
          const hx8 g = row[2 * ix - 3 + k];
          const float w = wzy * wx[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, (float)g[j], acc[j]);
        }
      }
    }
    hx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = to_h16_sat<HT>(acc[j], sat);
    (reinterpret_cast<hx8*>(dx16 + (int64_t)n * dxbs) + (int64_t)cb * S)[((int64_t)iz * H + iy) * W + ix] = o;
  }
  report_saturation(sat, oflag);
}

// y16[n][c][s] = x16[n][c][s] * scale[n * C + c]   (nn.Dropout3d on a c8 activation, and its backward)
template <typename HT>
__global__ __launch_bounds__(256) void channel_scale_c8_kernel(const HT* __restrict__ x16, const float* __restrict__ scale,
                                                               HT* __restrict__ y16, int N, int C, int CB, int64_t S,
                                                               int64_t xbs, int64_t ybs, int* __restrict__ oflag) {
  using hx8 = typename H16<HT>::x8;
  const int64_t total = (int64_t)N * CB * S;
  bool sat = false;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int64_t s = i % S;
    const int64_t r = i / S;
    const int cb = (int)(r % CB);
    const int n = (int)(r / CB);
    const hx8 v = (reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs) + (int64_t)cb * S)[s];
    hx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      o[j] = to_h16_sat<HT>(c < C ? (float)v[j] * scale[(int64_t)n * C + c] : 0.f, sat);
    }
    (reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs) + (int64_t)cb * S)[s] = o;
  }
  report_saturation(sat, oflag);
}

// space-to-depth / depth-to-space by 2 on c8 activations (the Blur convolutions of the msseg2 family in the 16-bit
// flows: models/components.py:91-154 run as a stride-1 3x3x3 convolution over the space-to-depth input, resp. producing
// the 8 output parities).  Packed channel c * 8 + p (p = pz * 4 + py * 2 + px) IS element p of c8 block c of the packed
// tensor, so the operation is an 8 x 8 transpose of 16-bit values per (half-resolution voxel, block of 8 channels):
//   TO_DEPTH: 8 items of the full tensor (the 2 x 2 x 2 voxels, 8 channels each) -> 8 items of the packed tensor (8 packed
//   blocks = the 8 channels, 8 parities each); otherwise the inverse.  C = channels of the FULL tensor.
template <typename HT, bool TO_DEPTH>
__global__ __launch_bounds__(256) void s2d_c8_kernel(const HT* __restrict__ src, HT* __restrict__ dst, int N, int C, int D,
                                                     int H, int W, int64_t fbs, int64_t pbs) {
  // D, H, W: full resolution; fbs / pbs: batch strides (elements) of the full / packed c8 tensors
  using hx8 = typename H16<HT>::x8;
  const int HD = D / 2, HH = H / 2, HW = W / 2;
  const int CB = (C + 7) / 8;
  const int64_t S = (int64_t)D * H * W, PS = (int64_t)HD * HH * HW;
  const int64_t total = (int64_t)N * CB * PS;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int x = (int)(i % HW);
    int64_t r = i / HW;
    const int y = (int)(r % HH);
    r /= HH;
    const int z = (int)(r % HD);
    r /= HD;
    const int cb = (int)(r % CB);
    const int n = (int)(r / CB);
    const int64_t pv = ((int64_t)z * HH + y) * HW + x;
    const int64_t fv0 = ((int64_t)(2 * z) * H + 2 * y) * W + 2 * x;
    if (TO_DEPTH) {
      const hx8* fin = reinterpret_cast<const hx8*>(src + (int64_t)n * fbs) + (int64_t)cb * S;
      hx8 it[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) it[p] = fin[fv0 + ((int64_t)(p >> 2) * H + ((p >> 1) & 1)) * W + (p & 1)];
      hx8* pout = reinterpret_cast<hx8*>(dst + (int64_t)n * pbs);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (cb * 8 + j >= C) break;
        hx8 o;
#pragma unroll
        for (int p = 0; p < 8; ++p) o[p] = it[p][j];
        pout[(int64_t)(cb * 8 + j) * PS + pv] = o;
      }
    } else {
      const hx8* pin = reinterpret_cast<const hx8*>(src + (int64_t)n * pbs);
      hx8 it[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const hx8 zero = {};
        it[j] = cb * 8 + j < C ? pin[(int64_t)(cb * 8 + j) * PS + pv] : zero;
      }
      hx8* fout = reinterpret_cast<hx8*>(dst + (int64_t)n * fbs) + (int64_t)cb * S;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        hx8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = it[j][p];
        fout[fv0 + ((int64_t)(p >> 2) * H + ((p >> 1) & 1)) * W + (p & 1)] = o;
      }
    }
  }
}

}  // namespace m355

using namespace m355;

static int s2d_common(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W, int64_t xbs_,
                      int64_t ybs_, void* stream, bool to_depth, const char* who) {
  M355_REQUIRE(x && y, M355_EINVALID_ARG, "%s: null pointer", who);
  M355_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, M355_EINVALID_ARG, "%s: non-positive dimension", who);
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED, "%s: odd spatial size (%d,%d,%d)", who,
               D, H, W);
  const int64_t dense = (int64_t)C * D * H * W;
  // to_depth: x is the full tensor; otherwise y is
  const int64_t fbs = dense_or(to_depth ? xbs_ : ybs_, dense), pbs = dense_or(to_depth ? ybs_ : xbs_, dense);
  const float* full = to_depth ? x : y;
  M355_REQUIRE(((uintptr_t)full & 7) == 0 && fbs % 2 == 0, M355_EUNSUPPORTED, "%s: full tensor not 8-byte aligned", who);
  const int64_t total = (int64_t)N * C * D * H * (W / 2);
  const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 256), 16384));
  if (to_depth)
    hipLaunchKernelGGL(s2d_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, C, D, H, W, fbs, pbs);
  else
    hipLaunchKernelGGL(s2d_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, C, D, H, W, fbs,
                       pbs);
  return check_launch(who);
}

extern "C" int m355_space_to_depth2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                    int64_t x_batch_stride, int64_t y_batch_stride, void* stream) {
  return s2d_common(x, y, N, C, D, H, W, x_batch_stride, y_batch_stride, stream, true, "space_to_depth2");
}
extern "C" int m355_depth_to_space2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                    int64_t x_batch_stride, int64_t y_batch_stride, void* stream) {
  return s2d_common(x, y, N, C, D, H, W, x_batch_stride, y_batch_stride, stream, false, "depth_to_space2");
}

static int check_ncdhw(int N, int C, int D, int H, int W, const char* who) {
  M355_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, M355_EINVALID_ARG,
               "%s: non-positive dimension", who);
  return M355_OK;
}

extern "C" int m355_avgpool3d_2x_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t D,
                                     int32_t H, int32_t W, int64_t x_batch_stride,
                                     int64_t y_batch_stride, void* stream) {
  if (int rc = check_ncdhw(N, C, D, H, W, "avgpool3d_2x_fwd")) return rc;
  M355_REQUIRE(x && y, M355_EINVALID_ARG, "avgpool3d_2x_fwd: null pointer");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED,
               "avgpool3d_2x_fwd: odd spatial size (%d,%d,%d)", D, H, W);
  const int64_t xbs = dense_or(x_batch_stride, (int64_t)C * D * H * W);
  const int64_t ybs = dense_or(y_batch_stride, (int64_t)C * (D / 2) * (H / 2) * (W / 2));
  const bool vec = (W % 4 == 0) && (xbs % 4 == 0) && (ybs % 2 == 0) && ((uintptr_t)x & 15) == 0 &&
                   ((uintptr_t)y & 7) == 0;
  const int64_t total = (int64_t)N * C * (D / 2) * (H / 2) * (vec ? W / 4 : W / 2);
  if (vec)
    hipLaunchKernelGGL(avgpool2_fwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, C, D, H, W, xbs, ybs);
  else
    hipLaunchKernelGGL(avgpool2_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, C, D, H, W, xbs, ybs);
  return check_launch("avgpool3d_2x_fwd");
}

extern "C" int m355_avgpool3d_2x_bwd(const float* dy, float* dx, int32_t N, int32_t C, int32_t D,
                                     int32_t H, int32_t W, int64_t dy_batch_stride,
                                     int64_t dx_batch_stride, void* stream) {
  if (int rc = check_ncdhw(N, C, D, H, W, "avgpool3d_2x_bwd")) return rc;
  M355_REQUIRE(dy && dx, M355_EINVALID_ARG, "avgpool3d_2x_bwd: null pointer");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED,
               "avgpool3d_2x_bwd: odd spatial size (%d,%d,%d)", D, H, W);
  const int64_t dxbs = dense_or(dx_batch_stride, (int64_t)C * D * H * W);
  const int64_t dybs = dense_or(dy_batch_stride, (int64_t)C * (D / 2) * (H / 2) * (W / 2));
  const int64_t total = (int64_t)N * C * D * H * (W / 2);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     dy, dx, N, C, D, H, W, dybs, dxbs, (const float*)nullptr, (int64_t)0);
  return check_launch("avgpool3d_2x_bwd");
}

extern "C" int m355_avgpool3d_2x_bwd_add(const float* dy, const float* add, float* dx, int32_t N, int32_t C, int32_t D,
                                         int32_t H, int32_t W, int64_t dy_batch_stride, int64_t add_batch_stride,
                                         int64_t dx_batch_stride, void* stream) {
  if (int rc = check_ncdhw(N, C, D, H, W, "avgpool3d_2x_bwd_add")) return rc;
  M355_REQUIRE(dy && add && dx, M355_EINVALID_ARG, "avgpool3d_2x_bwd_add: null pointer");
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED,
               "avgpool3d_2x_bwd_add: odd spatial size (%d,%d,%d)", D, H, W);
  const int64_t dense = (int64_t)C * D * H * W;
  const int64_t dxbs = dense_or(dx_batch_stride, dense), abs_ = dense_or(add_batch_stride, dense);
  const int64_t dybs = dense_or(dy_batch_stride, (int64_t)C * (D / 2) * (H / 2) * (W / 2));
  const int64_t total = (int64_t)N * C * D * H * (W / 2);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     dy, dx, N, C, D, H, W, dybs, dxbs, add, abs_);
  return check_launch("avgpool3d_2x_bwd_add");
}

extern "C" int m355_upsample_trilinear2x_fwd(const float* x, float* y, int32_t N, int32_t C,
                                             int32_t D, int32_t H, int32_t W,
                                             int64_t x_batch_stride, int64_t y_batch_stride,
                                             void* stream) {
  if (int rc = check_ncdhw(N, C, D, H, W, "upsample_trilinear2x_fwd")) return rc;
  M355_REQUIRE(x && y, M355_EINVALID_ARG, "upsample_trilinear2x_fwd: null pointer");
  const int64_t xbs = dense_or(x_batch_stride, (int64_t)C * D * H * W);
  const int64_t ybs = dense_or(y_batch_stride, (int64_t)C * D * H * W * 8);
  const int64_t total = (int64_t)N * C * D * H * W * 8;
  const bool quads = W % 2 == 0 && ybs % 4 == 0 && ((uintptr_t)y & 15) == 0;
  const size_t lds = (size_t)TRI_PZ * TRI_PY * W * sizeof(float);
  if (quads && D >= 2 && H >= 2 && lds <= 48 * 1024 && (int64_t)N * C <= 65535 && ceil_div(2 * D, TRI_TZ) <= 65535) {
    dim3 grid((unsigned)ceil_div(2 * H, TRI_TY), (unsigned)ceil_div(2 * D, TRI_TZ), (unsigned)(N * C));
    hipLaunchKernelGGL(trilinear2_fwd_lds_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, y, C, D, H, W, xbs, ybs);
  } else if (quads && (int64_t)N * C * D * H * 4 < (1ll << 31))
    hipLaunchKernelGGL(trilinear2_fwd_q_kernel, dim3(grid_for(total / 4, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, C, D, H, W, xbs, ybs);
  else
    hipLaunchKernelGGL(trilinear2_fwd_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, C, D, H, W, xbs, ybs);
  return check_launch("upsample_trilinear2x_fwd");
}

extern "C" int m355_upsample_trilinear2x_bwd(const float* dy, float* dx, int32_t N, int32_t C,
                                             int32_t D, int32_t H, int32_t W,
                                             int64_t dy_batch_stride, int64_t dx_batch_stride,
                                             void* stream) {
  if (int rc = check_ncdhw(N, C, D, H, W, "upsample_trilinear2x_bwd")) return rc;
  M355_REQUIRE(dy && dx, M355_EINVALID_ARG, "upsample_trilinear2x_bwd: null pointer");
  const int64_t dxbs = dense_or(dx_batch_stride, (int64_t)C * D * H * W);
  const int64_t dybs = dense_or(dy_batch_stride, (int64_t)C * D * H * W * 8);
  const int64_t total = (int64_t)N * C * D * H * W;
  hipLaunchKernelGGL(trilinear2_bwd_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, N, C, D, H, W, dybs, dxbs);
  return check_launch("upsample_trilinear2x_bwd");
}

static int check_c8_op(const char* who, const void* a, const void* b, int N, int C, int D, int H, int W, int compute,
                       int64_t abs16, int64_t bbs16) {
  M355_REQUIRE(a && b, M355_EINVALID_ARG, "%s: null pointer", who);
  M355_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, M355_EINVALID_ARG, "%s: non-positive dimension", who);
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "%s: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16", who);
  M355_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0 && abs16 % 8 == 0 && bbs16 % 8 == 0, M355_EINVALID_ARG,
               "%s: c8 tensor not 16B aligned", who);
  return M355_OK;
}

extern "C" int m355_upsample_trilinear2x_fwd_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H,
                                                 int32_t W, int64_t x16_batch_stride, int64_t y16_batch_stride,
                                                 int32_t compute, void* stream) {
  const int CB = (int)c8_blocks(C);
  const int64_t S = (int64_t)D * H * W;
  const int64_t xbs = dense_or(x16_batch_stride, CB * S * 8), ybs = dense_or(y16_batch_stride, CB * S * 64);
  if (int rc = check_c8_op("upsample_trilinear2x_fwd_h16", x16, y16, N, C, D, H, W, compute, xbs, ybs)) return rc;
  const int64_t total = (int64_t)N * CB * S * 8;
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(trilinear2_fwd_c8_kernel<__bf16>, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)x16, (__bf16*)y16, N, CB, D, H, W, xbs, ybs);
  else
    hipLaunchKernelGGL(trilinear2_fwd_c8_kernel<_Float16>, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)x16, (_Float16*)y16, N, CB, D, H, W, xbs, ybs);
  return check_launch("upsample_trilinear2x_fwd_h16");
}

extern "C" int m355_upsample_trilinear2x_bwd_h16(const void* dy16, void* dx16, int32_t N, int32_t C, int32_t D, int32_t H,
                                                 int32_t W, int64_t dy16_batch_stride, int64_t dx16_batch_stride,
                                                 int32_t compute, void* stream) {
  const int CB = (int)c8_blocks(C);
  const int64_t S = (int64_t)D * H * W;
  const int64_t dybs = dense_or(dy16_batch_stride, CB * S * 64), dxbs = dense_or(dx16_batch_stride, CB * S * 8);
  if (int rc = check_c8_op("upsample_trilinear2x_bwd_h16", dy16, dx16, N, C, D, H, W, compute, dybs, dxbs)) return rc;
  const int64_t total = (int64_t)N * CB * S;
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(trilinear2_bwd_c8_kernel<__bf16>, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)dy16, (__bf16*)dx16, N, CB, D, H, W, dybs, dxbs, overflow_flag());
  else
    hipLaunchKernelGGL(trilinear2_bwd_c8_kernel<_Float16>, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)dy16, (_Float16*)dx16, N, CB, D, H, W, dybs, dxbs,
                       overflow_flag());
  return check_launch("upsample_trilinear2x_bwd_h16");
}

extern "C" int m355_act16_channel_scale(const void* x16, const float* scale, void* y16, int32_t N, int32_t C, int64_t S,
                                        int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute,
                                        void* stream) {
  const int CB = (int)c8_blocks(C);
  const int64_t xbs = dense_or(x16_batch_stride, CB * S * 8), ybs = dense_or(y16_batch_stride, CB * S * 8);
  M355_REQUIRE(scale && S > 0, M355_EINVALID_ARG, "act16_channel_scale: null scale or empty tensor");
  if (int rc = check_c8_op("act16_channel_scale", x16, y16, N, C, 1, 1, 1, compute, xbs, ybs)) return rc;
  const int64_t total = (int64_t)N * CB * S;
  if (compute == M355_COMPUTE_BF16)
    hipLaunchKernelGGL(channel_scale_c8_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x16, scale, (__bf16*)y16, N, C, CB, S, xbs, ybs, overflow_flag());
  else
    hipLaunchKernelGGL(channel_scale_c8_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x16, scale, (_Float16*)y16, N, C, CB, S, xbs, ybs, overflow_flag());
  return check_launch("act16_channel_scale");
}

static int s2d_c8_common(const void* x16, void* y16, int N, int C, int D, int H, int W, int64_t xbs_, int64_t ybs_,
                         int compute, void* stream, bool to_depth, const char* who) {
  M355_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, M355_EUNSUPPORTED, "%s: odd spatial size (%d,%d,%d)", who, D, H, W);
  const int64_t S = (int64_t)D * H * W;
  const int64_t fdense = c8_blocks(C) * S * 8, pdense = (int64_t)C * (S / 8) * 8;   // packed: C blocks of 8 parities
  const int64_t fbs = dense_or(to_depth ? xbs_ : ybs_, fdense), pbs = dense_or(to_depth ? ybs_ : xbs_, pdense);
  if (int rc = check_c8_op(who, x16, y16, N, C, D, H, W, compute, fbs, pbs)) return rc;
  const int64_t total = (int64_t)N * c8_blocks(C) * (S / 8);
  const dim3 grid(grid_for(total, 256, 65536));
#define M355_S2D(HT, TD) \
  hipLaunchKernelGGL((s2d_c8_kernel<HT, TD>), grid, dim3(256), 0, (hipStream_t)stream, (const HT*)x16, (HT*)y16, N, C, D, H, W, fbs, pbs)
  if (compute == M355_COMPUTE_BF16) {
    if (to_depth) M355_S2D(__bf16, true); else M355_S2D(__bf16, false);
  } else {
    if (to_depth) M355_S2D(_Float16, true); else M355_S2D(_Float16, false);
  }
#undef M355_S2D
  return check_launch(who);
}

extern "C" int m355_space_to_depth2_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                        int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream) {
  return s2d_c8_common(x16, y16, N, C, D, H, W, x16_batch_stride, y16_batch_stride, compute, stream, true,
                       "space_to_depth2_h16");
}
extern "C" int m355_depth_to_space2_h16(const void* x16, void* y16, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                                        int64_t x16_batch_stride, int64_t y16_batch_stride, int32_t compute, void* stream) {
  return s2d_c8_common(x16, y16, N, C, D, H, W, x16_batch_stride, y16_batch_stride, compute, stream, false,
                       "depth_to_space2_h16");
}

extern "C" int m355_softmax_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t inner,
                                int64_t S, float diag_bias, void* stream) {
  M355_REQUIRE(x && y, M355_EINVALID_ARG, "softmax_fwd: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && inner > 0 && S > 0, M355_EINVALID_ARG,
               "softmax_fwd: non-positive size");
  M355_REQUIRE(inner == 1 || inner == C, M355_EINVALID_ARG,
               "softmax_fwd: inner must be 1 or C (stochastic matrix)");
  const int64_t total = (int64_t)N * inner * S;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x,
                     y, N, C, inner, S, diag_bias);
  return check_launch("softmax_fwd");
}

extern "C" int m355_softmax_bwd(const float* y, const float* dy, float* dx, int32_t N, int32_t C,
                                int32_t inner, int64_t S, void* stream) {
  M355_REQUIRE(y && dy && dx, M355_EINVALID_ARG, "softmax_bwd: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && inner > 0 && S > 0, M355_EINVALID_ARG,
               "softmax_bwd: non-positive size");
  const int64_t total = (int64_t)N * inner * S;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y,
                     dy, dx, N, C, inner, S);
  return check_launch("softmax_bwd");
}

extern "C" int m355_copy_channels(const float* src, float* dst, int32_t N, int32_t C, int64_t S,
                                  int64_t src_batch_stride, int64_t dst_batch_stride,
                                  void* stream) {
  M355_REQUIRE(src && dst, M355_EINVALID_ARG, "copy_channels: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0, M355_EINVALID_ARG, "copy_channels: non-positive size");
  const int64_t CS = (int64_t)C * S;
  const int64_t sbs = dense_or(src_batch_stride, CS), dbs = dense_or(dst_batch_stride, CS);
  const bool vec = (CS % 4 == 0) && (sbs % 4 == 0) && (dbs % 4 == 0) &&
                   (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  const int64_t total = (int64_t)N * (vec ? CS / 4 : CS);
  if (vec)
    hipLaunchKernelGGL(copy_channels_kernel<true>, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, CS, sbs, dbs);
  else
    hipLaunchKernelGGL(copy_channels_kernel<false>, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, CS, sbs, dbs);
  return check_launch("copy_channels");
}

extern "C" int m355_channel_scale(const float* x, const float* scale, float* y, int32_t N, int32_t C,
                                  int64_t S, void* stream) {
  M355_REQUIRE(x && scale && y, M355_EINVALID_ARG, "channel_scale: null pointer");
  M355_REQUIRE(N > 0 && C > 0 && S > 0, M355_EINVALID_ARG, "channel_scale: non-positive size");
  const int64_t total = (int64_t)N * C * S;
  hipLaunchKernelGGL(channel_scale_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     x, scale, y, (int64_t)N * C, S);
  return check_launch("channel_scale");
}

extern "C" int m355_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  M355_REQUIRE(a && b && y, M355_EINVALID_ARG, "add: null pointer");
  M355_REQUIRE(n > 0, M355_EINVALID_ARG, "add: non-positive size");
  const bool vec = (n % 4 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(add_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a,
                       b, y, n);
  else
    hipLaunchKernelGGL(add_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a,
                       b, y, n);
  return check_launch("add");
}
