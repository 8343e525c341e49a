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
// y / dy accesses); any other geometry takes the generic direct kernels below.  (M355_COMPUTE_F32X3: the forward runs
// on the bf16 matrix pipe through the exact three-way operand split -- convt_k2s2_fwd_x3_kernel, conv3d_f32x3.hip.)
#include "h16.hpp"

namespace m355 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the k2s2 forward of M355_COMPUTE_F32X3 (conv3d_f32x3.hip)
int convt_fwd_x3_nvt(int Cin);
void launch_convt_fwd_x3(int nvt, dim3 grid, const float* x, const float* w, const float* bias, float* y, int Cin, int Cout,
                         int D, int H, int W, int64_t xbs, int64_t ybs, int mt_per_wg, hipStream_t st);

// =============================== k2s2 as fp32-MFMA GEMMs ===============================
// Every input voxel owns its 2x2x2 output block, so with t = a*4 + b*2 + c (the position
// inside the block) and m = o*8 + t:
//   fwd        Y[m, v]  = bias[o] + sum_ci W[ci, m] * X[ci, v]          M = 8*Cout, K = Cin
//   bwd-data   dX[ci,v] =           sum_m  W[ci, m] * dYr[m, v]         M = Cin,    K = 8*Cout
//   bwd-weight dW[ci,m] =           sum_v  X[ci, v] * dYr[m, v]         M = Cin, N = 8*Cout, K = voxels
// where dYr[m, v] = dY[o, 2z+a, 2y+b, 2x+c] is the output block gathered per input voxel and
// the torch weight layout [Cin][Cout][2][2][2] is already W[ci][m] with m contiguous.
// AI ~ 28 flop/B: right at the fp32-MFMA / HBM ridge, so each kernel reads its large operand
// exactly once and overlaps the stream with the MFMAs (register prefetch of the next K slab while
// the current one is multiplied out of LDS; several workgroups per CU).

// ---- forward: workgroup = NVT consecutive input voxels, the whole [Cin][NVT] x tile resident in
// LDS (read once); the four waves walk different 32-row m-tiles (4 output channels each) and read
// their weights straight from L2 as the MFMA A operand.  C/D layout puts t = (r&3) + 4*half on the
// lane's registers, so (r, r+1) is the (c=0, c=1) pair of one (o, a, b): float2 stores, 256 B
// contiguous per 32 lanes.
template <int NVT>
__global__ __launch_bounds__(256) void convt_k2s2_fwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs, int mt_per_wg) {
  extern __shared__ float xs[];  // [CinR][NVT], CinR = Cin rounded up to a whole chunk (rows >= Cin are zero)
  constexpr int NG = NVT / 32, P = 256 / NVT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int n = blockIdx.z;
  const float* xn = x + (int64_t)n * xbs;
  float* yn = y + (int64_t)n * ybs;
  constexpr int KC = 8;  // k-pairs per weight chunk
  const int Cin2 = (Cin + 2 * KC - 1) / (2 * KC) * (2 * KC);
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  {
    const int sv = tid % NVT, part = tid / NVT;
    const bool vin = v0 + sv < S;
    const float* xp = xn + (vin ? v0 + sv : 0);
#pragma unroll 16
    for (int c = part; c < Cin2; c += P) {
      const bool ok = vin && c < Cin;
      const float val = xp[ok ? (int64_t)c * S : 0];  // unconditional load from a clamped address
      xs[c * NVT + sv] = ok ? val : 0.f;
    }
  }
  __syncthreads();
  const int mtiles = (Cout + 3) / 4;
  const int mt_begin = blockIdx.y * mt_per_wg, mt_end = min(mtiles, mt_begin + mt_per_wg);
  const int wrow = Cout * 8;  // weight offsets fit 32 bits (host: convt_fits_i32)
  // output offsets of this lane's voxels (a = half)
  int64_t obase[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int v = min(v0 + g * 32 + l32, S - 1);
    const int ix = v % W, iy = (v / W) % H, iz = v / (W * H);
    obase[g] = ((int64_t)(2 * iz + half) * OH + 2 * iy) * OW + 2 * ix;
  }
  // weights: KC k-pairs per chunk in registers; the next chunk (possibly of the next m-tile) is in
  // flight during the MFMAs of the current one
  float a_cur[KC], a_nxt[KC];
  // raw loads from clamped addresses; the zero mask is applied when the chunk becomes current
  // (a select right after the load would make the wave wait for it before the MFMAs)
  auto wload = [&](float* a, int mt, int k0) {
    const int col = min(mt * 32 + l32, Cout * 8 - 1);  // always in bounds: no select, no branch
#pragma unroll
    for (int j = 0; j < KC; ++j) a[j] = w[min(k0 + 2 * j + half, Cin - 1) * wrow + col];
  };
  auto wmask = [&](float* dst, const float* src, int mt, int k0) {
    const bool ook = mt * 4 + (l32 >> 3) < Cout;
#pragma unroll
    for (int j = 0; j < KC; ++j) dst[j] = (ook && k0 + 2 * j + half < Cin) ? src[j] : 0.f;
  };
  const float* xb = xs + half * NVT + l32;
  int mt = mt_begin + wave;
  if (mt < mt_end) {
    wload(a_nxt, mt, 0);
    wmask(a_cur, a_nxt, mt, 0);
  }
  while (mt < mt_end) {
    const int o0 = mt * 4;
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    for (int k0 = 0; k0 < Cin2; k0 += 2 * KC) {
      const bool last = k0 + 2 * KC >= Cin2;
      const int nmt = last ? mt + 4 : mt, nk0 = last ? 0 : k0 + 2 * KC;
      if (nmt < mt_end) wload(a_nxt, nmt, nk0);
      // branch-free chunk: the LDS reads of step j+1 are issued before the MFMAs of step j
      float bq[2][NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) bq[0][g] = xb[k0 * NVT + g * 32];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        if (j + 1 < KC) {
#pragma unroll
          for (int g = 0; g < NG; ++g) bq[(j + 1) & 1][g] = xb[(k0 + 2 * j + 2) * NVT + g * 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks each read to just before its MFMA
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], bq[j & 1][g], acc[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      wmask(a_cur, a_nxt, nmt, nk0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // q = output channel inside the tile (r >> 2)
      const int o = o0 + q;
      if (o >= Cout) continue;
      const float bv = bias ? bias[o] : 0.f;
      float* yo = yn + (int64_t)o * OS;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (v0 + g * 32 + l32 >= S) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
          *reinterpret_cast<float2*>(yo + obase[g] + (int64_t)b * OW) =
              make_float2(acc[g][q * 4 + b * 2] + bv, acc[g][q * 4 + b * 2 + 1] + bv);
      }
    }
    mt += 4;
  }
}

// ---- c8 -> c8 variant (16-bit precision modes under no_grad, ops.Act16): the same fp32-MFMA GEMM (the op is
// HBM-bound, the weights stay exact fp32), but x arrives as c8 items (16 B = 8 channels of a voxel, converted to
// fp32 while staged) and the output is written as c8 straight into its slot of the decoder's concat buffer:
// a wave owns TWO adjacent m-tiles (8 output channels), so a lane holds the 8 channels of each of its output
// voxels (a = lane half, (b, c) on the registers) and stores them as one 16-byte item.  No fp32 round trip
// (unpack -> conv-transpose -> pack cost 3 extra passes over the largest decoder tensors).
template <int NVT, typename HT>
__global__ __launch_bounds__(256) void convt_k2s2_fwd_c8_kernel(
    const HT* __restrict__ x16, const float* __restrict__ w, const float* __restrict__ bias, HT* __restrict__ y16,
    int Cin, int Cout, int D, int H, int W, int64_t xbs16, int64_t ybs16, int mp_per_wg) {
  using hx8 = typename H16<HT>::x8;
  extern __shared__ float xs[];  // [Cin2][NVT] fp32, rows >= Cin are zero
  constexpr int NG = NVT / 32, P = 256 / NVT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int n = blockIdx.z;
  constexpr int KC = 8;
  const int Cin2 = (Cin + 2 * KC - 1) / (2 * KC) * (2 * KC);
  const int CBin = (Cin + 7) / 8;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  {
    const int sv = tid % NVT, part = tid / NVT;
    const bool vin = v0 + sv < S;
    const hx8* xp = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16) + (vin ? v0 + sv : 0);
    for (int cb = part; cb < Cin2 / 8; cb += P) {
      const bool ok = vin && cb < CBin;
      const hx8 v = xp[ok ? (int64_t)cb * S : 0];
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[(cb * 8 + j) * NVT + sv] = (ok && cb * 8 + j < Cin) ? (float)v[j] : 0.f;
    }
  }
  __syncthreads();
  const int mtiles = (Cout + 3) / 4, mpairs = (mtiles + 1) / 2;
  const int mp_begin = blockIdx.y * mp_per_wg, mp_end = min(mpairs, mp_begin + mp_per_wg);
  const int wrow = Cout * 8;
  int64_t obase[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int v = min(v0 + g * 32 + l32, S - 1);
    const int ix = v % W, iy = (v / W) % H, iz = v / (W * H);
    obase[g] = ((int64_t)(2 * iz + half) * OH + 2 * iy) * OW + 2 * ix;
  }
  float a_cur[2][KC], a_nxt[2][KC];
  auto wload = [&](int mp, int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = min((2 * mp + t) * 32 + l32, Cout * 8 - 1);
#pragma unroll
      for (int j = 0; j < KC; ++j) a_nxt[t][j] = w[min(k0 + 2 * j + half, Cin - 1) * wrow + col];
    }
  };
  auto wmask = [&](int mp, int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bool ook = (2 * mp + t) * 4 + (l32 >> 3) < Cout;
#pragma unroll
      for (int j = 0; j < KC; ++j) a_cur[t][j] = (ook && k0 + 2 * j + half < Cin) ? a_nxt[t][j] : 0.f;
    }
  };
  const float* xb = xs + half * NVT + l32;
  int mp = mp_begin + wave;
  if (mp < mp_end) {
    wload(mp, 0);
    wmask(mp, 0);
  }
  hx8* yn = reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs16);
  while (mp < mp_end) {
    f32x16 acc[2][NG];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][g][r] = 0.f;
    for (int k0 = 0; k0 < Cin2; k0 += 2 * KC) {
      const bool last = k0 + 2 * KC >= Cin2;
      const int nmp = last ? mp + 4 : mp, nk0 = last ? 0 : k0 + 2 * KC;
      if (nmp < mp_end) wload(nmp, nk0);
      float bq[2][NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) bq[0][g] = xb[k0 * NVT + g * 32];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        if (j + 1 < KC) {
#pragma unroll
          for (int g = 0; g < NG; ++g) bq[(j + 1) & 1][g] = xb[(k0 + 2 * j + 2) * NVT + g * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[t][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][j], bq[j & 1][g], acc[t][g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      wmask(nmp, nk0);
    }
    // 8 output channels o0 .. o0+7 = channel block mp of the c8 output
    const int o0 = mp * 8;
    float bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = (bias && o0 + q < Cout) ? bias[o0 + q] : 0.f;
    hx8* yo = yn + (int64_t)mp * OS;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (v0 + g * 32 + l32 >= S) continue;
#pragma unroll
      for (int bc = 0; bc < 4; ++bc) {  // (b, c) = output y / x parity
        hx8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = (HT)(acc[0][g][q * 4 + bc] + bv[q]);
          o[4 + q] = (HT)(acc[1][g][q * 4 + bc] + bv[4 + q]);
        }
        yo[obase[g] + (int64_t)(bc >> 1) * OW + (bc & 1)] = o;
      }
    }
    mp += 4;
  }
}

// ---- c8 -> c8 on the 16-bit matrix core (large levels).  The kernel above multiplies in fp32 (157 TFLOP/s peak)
// and lands its 16-byte items 32 bytes apart; for the two full-resolution decoder levels that is 3-6x the time the
// bytes need.  Here the GEMM [8 Cout x Cin] x [Cin x voxels] runs on v_mfma_f32_32x32x16_{bf16,f16}: a c8 item
// (8 channels of a voxel) IS the B fragment of a lane, loaded from global memory straight into registers and kept
// for all the m-tiles; the weights are rounded to the 16-bit type (the operand rounding of the mode, as in the
// 3x3x3 kernels) and staged ONCE per workgroup in LDS in A-fragment order, then the workgroup walks voxel tiles.
// An m-tile is (channel block cb, z parity a); its 32 rows are ordered so that a lane ends up with the 8 channels
// of two output voxels -- row m: channel (m & 3) + 4 * ((m >> 3) & 1), x parity c = (m >> 2) & 1 = the lane half,
// y parity b = m >> 4 -- and one store instruction of the wave covers 64 consecutive 16-byte items of an output row.
template <typename HT, int KS, int NG>
__global__ __launch_bounds__(256) void convt_k2s2_fwd_h16_kernel(
    const HT* __restrict__ x16, const float* __restrict__ w, const float* __restrict__ bias, HT* __restrict__ y16,
    int Cin, int Cout, int D, int H, int W, int64_t xbs16, int64_t ybs16, int mt_per_wg) {
  using hx8 = typename H16<HT>::x8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  hx8* afrag = reinterpret_cast<hx8*>(lds_raw);  // [m-tile of this workgroup][k-step][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int ks_n = (Cin + 15) / 16, CBin = (Cin + 7) / 8;
  const int mtiles = 2 * ((Cout + 7) / 8);
  const int mt0 = blockIdx.y * mt_per_wg, nmt = min(mtiles, mt0 + mt_per_wg) - mt0;
  {
    // staging in the MEMORY order of w ([Cin][Cout][a][b][c]: (o, b, c) runs of one (k, a) are 16-byte pieces 32 bytes
    // apart) with 2-byte scatter writes into fragment order -- every element of the fragment space is written exactly
    // once, zeros past Cin / Cout.  (Gathering each fragment with 8 loads Cout * 32 bytes apart made this staging,
    // repeated by every workgroup, longer than the voxel tiles it serves.)
    HT* af16 = reinterpret_cast<HT*>(lds_raw);
    const int kpad = ks_n * 16;
    const int total4 = nmt * kpad * 8;                     // float4 pieces: the (b, c) quad of one (k, o, a)
    for (int e0 = tid; e0 < total4; e0 += 256 * 8) {       // 8 x 16-byte loads in flight per thread
      float4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        const int o_in = e & 7, k = (e >> 3) % kpad, tl = (e >> 3) / kpad;
        const int mt = mt0 + tl, cb = mt >> 1, a = mt & 1, o = cb * 8 + o_in;
        const bool ok = e < total4 && k < Cin && o < Cout;
        v[u] = ok ? *reinterpret_cast<const float4*>(w + (((int64_t)k * Cout + o) * 2 + a) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        // row m = (o_in & 3) + 4 c + 8 (o_in >> 2) + 16 b of the m-tile, element k & 7 of lane half (k >> 3) & 1
        dst[u] = e < total4 ? ((tl * ks_n + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (o_in & 3) + 8 * (o_in >> 2)) * 8 + (k & 7) : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) {
          af16[dst[u]] = (HT)v[u].x;               // (b, c) = (0, 0)
          af16[dst[u] + 4 * 8] = (HT)v[u].y;       // c = 1: row + 4
          af16[dst[u] + 16 * 8] = (HT)v[u].z;      // b = 1: row + 16
          af16[dst[u] + 20 * 8] = (HT)v[u].w;
        }
    }
  }
  __syncthreads();
  const int n = blockIdx.z;
  const hx8* xin = reinterpret_cast<const hx8*>(x16 + (int64_t)n * xbs16);
  hx8* yn = reinterpret_cast<hx8*>(y16 + (int64_t)n * ybs16);
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  const hx8 zero = {};
  for (int vt = blockIdx.x; (int64_t)vt * (128 * NG) < S; vt += gridDim.x) {
    const int vb = vt * (128 * NG) + wave * (32 * NG);
    if (vb >= S) continue;
    hx8 bfr[KS][NG];
    int64_t obase[NG];
    bool vok[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int v = vb + g * 32 + l32;
      vok[g] = v < S;
      const int vc = min(v, S - 1);
      const int ix = vc % W, iy = (vc / W) % H, iz = vc / (W * H);
      obase[g] = ((int64_t)(2 * iz) * OH + 2 * iy) * OW + 2 * ix + half;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int cbk = 2 * s + half;
        bfr[s][g] = (s < ks_n && vok[g] && cbk < CBin) ? xin[(int64_t)cbk * S + vc] : zero;
      }
    }
    for (int t = 0; t < nmt; ++t) {
      const int mt = mt0 + t, cb = mt >> 1, a = mt & 1;
      f32x16 acc[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
      const hx8* af = afrag + (int64_t)t * ks_n * 64 + lane;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        if (s < ks_n) {
          const hx8 av = af[s * 64];
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g] = H16<HT>::mfma(av, bfr[s][g], acc[g]);
        }
      float bv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) bv[q] = (bias && cb * 8 + q < Cout) ? bias[cb * 8 + q] : 0.f;
      hx8* yo = yn + (int64_t)cb * OS + (int64_t)a * OH * OW;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (!vok[g]) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          hx8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (HT)(acc[g][b * 8 + q] + bv[q]);
          yo[obase[g] + (int64_t)b * OW] = o;
        }
      }
    }
  }
}

// ---- data gradient: workgroup = NVT input voxels x 32*MT input channels; K is walked 4 output
// channels (32 k) at a time.  dY is de-interleaved while staging (dys[(o,t)][v]; the k-pair is
// (c=0, c=1) of one (o,a,b)); the next slab's global loads are in flight while the current one is
// multiplied.  Waves tile [MT m-tiles] x [NVT/32 n-tiles] as WM x WN.
template <int NVT, int MT>
__global__ __launch_bounds__(256) void convt_k2s2_bwd_data_mfma_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int Cin,
    int Cout_all, int D, int H, int W, int64_t xbs, int64_t ybs, int ksplit, int o_per_split,
    float* __restrict__ slab) {
  // split-K: blockIdx.z = n * ksplit + ks; this workgroup contracts output channels [o_lo, Cout) only
  // and writes its partial dX to slab[ks] (dense [N][Cin][S]); convt_dx_reduce_kernel sums the splits
  constexpr int KO = 4, KK = KO * 8;
  constexpr int P = 256 / NVT, ROWS = 16 / P;        // float2 rows staged per thread
  constexpr int CT = 32 * MT, WSS = CT + 1, WPT = CT * KK / 256;
  constexpr int WN = (NVT / 32 < 4) ? NVT / 32 : 4, WM = 4 / WN;
  constexpr int NGW = NVT / 32 / WN, MTW = MT / WM;
  static_assert(MT % WM == 0 && NGW >= 1, "wave tiling");
  __shared__ float dys[KK * NVT];
  __shared__ float ws[KK * WSS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_n = wave % WN, wave_m = wave / WN;
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int c0 = blockIdx.y * CT;
  const int n = (int)blockIdx.z / ksplit, ks = (int)blockIdx.z % ksplit;
  const int o_lo = ks * o_per_split;
  const int Cout = min(Cout_all, o_lo + o_per_split);  // end of this split's channel range
  const float* dyn = dy + (int64_t)n * ybs;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  // this thread's staging voxel and rows
  const int svl = tid % NVT, part = tid / NVT;
  const int sv = v0 + svl;
  const bool sok = sv < S;
  const int svv = sok ? sv : 0;
  const int64_t sbase = ((int64_t)(2 * (svv / (W * H))) * OH + 2 * ((svv / W) % H)) * OW + 2 * (svv % W);

  f32x16 acc[MTW][NGW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int g = 0; g < NGW; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.f;

  float2 pf[ROWS];
  float wpf[WPT];
  auto fetch = [&](int o0) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int idx = part + P * j, o = o0 + (idx >> 2), ab = idx & 3;
      const bool ok = sok && o < Cout;
      pf[j] = *reinterpret_cast<const float2*>(
          dyn + (ok ? (int64_t)o * OS + sbase + (int64_t)(ab >> 1) * OH * OW + (int64_t)(ab & 1) * OW : 0));
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int i = tid + 256 * j;
      const int k = i & (KK - 1), c = i >> 5;  // KK == 32
      const int o = o0 + (k >> 3);
      const bool ok = c0 + c < Cin && o < Cout;
      wpf[j] = w[ok ? ((int64_t)(c0 + c) * Cout_all + o) * 8 + (k & 7) : 0];  // row stride: ALL output channels
    }
  };
  auto commit = [&](int o0) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int idx = part + P * j, o = o0 + (idx >> 2);
      const bool ok = sok && o < Cout;
      dys[(idx * 2) * NVT + svl] = ok ? pf[j].x : 0.f;
      dys[(idx * 2 + 1) * NVT + svl] = ok ? pf[j].y : 0.f;
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int i = tid + 256 * j;
      const int k = i & (KK - 1), c = i >> 5;
      const bool ok = c0 + c < Cin && o0 + (k >> 3) < Cout;
      ws[k * WSS + c] = ok ? wpf[j] : 0.f;  // transposed: row stride WSS (odd) is conflict-free both ways
    }
  };

  fetch(o_lo);
  for (int o0 = o_lo; o0 < Cout; o0 += KO) {
    __syncthreads();  // previous slab fully consumed
    commit(o0);
    __syncthreads();
    if (o0 + KO < Cout) fetch(o0 + KO);  // in flight during the MFMAs below
    const float* wb = ws + half * WSS + wave_m * MTW * 32 + l32;
    const float* db = dys + half * NVT + wave_n * NGW * 32 + l32;
    // software-pipelined over the 16 k-pairs: the LDS reads of pair s+1 are issued before the MFMAs of
    // pair s (sched_barrier: hipcc otherwise sinks every read to just before its MFMA and waits on it)
    float a[2][MTW], b[2][NGW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) a[0][m] = wb[m * 32];
#pragma unroll
    for (int g = 0; g < NGW; ++g) b[0][g] = db[g * 32];
#pragma unroll
    for (int s = 0; s < KK / 2; ++s) {
      if (s + 1 < KK / 2) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[(s + 1) & 1][m] = wb[(2 * s + 2) * WSS + m * 32];
#pragma unroll
        for (int g = 0; g < NGW; ++g) b[(s + 1) & 1][g] = db[(2 * s + 2) * NVT + g * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int g = 0; g < NGW; ++g)
          acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][m], b[s & 1][g], acc[m][g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* dxn = ksplit == 1 ? dx + (int64_t)n * xbs : slab + ((int64_t)ks * gridDim.z / ksplit + n) * Cin * S;
#pragma unroll
  for (int g = 0; g < NGW; ++g) {
    const int v = v0 + (wave_n * NGW + g) * 32 + l32;
    if (v >= S) continue;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + (wave_m * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < Cin) dxn[(int64_t)c * S + v] = acc[m][g][r];
      }
  }
}

// ---- weight gradient: workgroup = 32*MT input channels x 16 output channels (wave w: 4 o x 8 t
// = one N-tile), persistent over the 64-voxel tiles of its split with the next tile's loads in
// flight during the MFMAs.  Partial dW -> slab[split][Cin][Cout][8] (coalesced: the lane index IS
// (o,t)), fixed-order reduce.  The bias gradient falls out of the dY values already in registers:
// per-thread partial sums per output channel, reduced at the end -> bslab[split][Cout] (double).
template <int MT>
__global__ __launch_bounds__(256) void convt_k2s2_bwd_weight_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab,
    double* __restrict__ bslab, int N, int Cin, int Cout, int D, int H, int W, int64_t xbs, int64_t ybs,
    int nsplit) {
  constexpr int NV = 64, CT = 32 * MT, XR = CT / 4;
  __shared__ float xs[CT * (NV + 1)];
  __shared__ float dys[128 * (NV + 1)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int split = blockIdx.x, o0 = blockIdx.y * 16, c0 = blockIdx.z * CT;
  const int S = D * H * W;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  const int64_t total = (int64_t)N * S;
  const int64_t ntiles = (total + NV - 1) / NV;
  const bool want_bias = bslab != nullptr && blockIdx.z == 0;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float bsum[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bsum[j] = 0.f;

  const int sv = tid & 63, sq = wave;  // staging: voxel, quarter (rows sq, sq+4, ...; ab == sq for dY)
  float xpf[XR];
  float2 dpf[16];
  bool tok = false;
  auto fetch = [&](int64_t tile) {
    const int64_t g = tile * NV + sv;
    tok = g < total;
    const int64_t nn = tok ? g / S : 0;
    const int v = tok ? (int)(g - nn * S) : 0;
    const float* xp = x + nn * xbs + v;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int c = c0 + sq + 4 * j;
      xpf[j] = xp[(tok && c < Cin) ? (int64_t)c * S : 0];
    }
    const float* dp = dy + nn * ybs + ((int64_t)(2 * (v / (W * H)) + (sq >> 1)) * OH + 2 * ((v / W) % H) + (sq & 1)) * OW +
                      2 * (v % W);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = o0 + j;
      dpf[j] = *reinterpret_cast<const float2*>(dp + ((tok && o < Cout) ? (int64_t)o * OS : 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int c = sq + 4 * j;
      xs[c * (NV + 1) + sv] = (tok && c0 + c < Cin) ? xpf[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const bool ok = tok && o0 + j < Cout;
      const float vx = ok ? dpf[j].x : 0.f, vy = ok ? dpf[j].y : 0.f;
      dys[(j * 8 + sq * 2) * (NV + 1) + sv] = vx;
      dys[(j * 8 + sq * 2 + 1) * (NV + 1) + sv] = vy;
      bsum[j] += vx + vy;
    }
  };

  if (split < ntiles) fetch(split);
  for (int64_t tile = split; tile < ntiles; tile += nsplit) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + nsplit < ntiles) fetch(tile + nsplit);
    const float* xb = xs + l32 * (NV + 1) + half;
    const float* db = dys + (wave * 32 + l32) * (NV + 1) + half;
    float a[2][MT], b[2];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[0][m] = xb[m * 32 * (NV + 1)];
    b[0] = db[0];
#pragma unroll
    for (int s = 0; s < NV / 2; ++s) {
      if (s + 1 < NV / 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[(s + 1) & 1][m] = xb[m * 32 * (NV + 1) + 2 * s + 2];
        b[(s + 1) & 1] = db[2 * s + 2];
      }
      __builtin_amdgcn_sched_barrier(0);  // reads of pair s+1 in flight during the MFMAs of pair s
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][m], b[s & 1], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D[i = ci][j = (o,t)]
  const int o = o0 + wave * 4 + (l32 >> 3);
  if (o < Cout) {
    float* sl = slab + (int64_t)split * Cin * Cout * 8;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < Cin) sl[((int64_t)c * Cout + o) * 8 + (l32 & 7)] = acc[m][r];
      }
  }
  if (want_bias) {
    __syncthreads();
    float* red = dys;  // [16 o][4 waves]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float s = wave_sum(bsum[j]);
      if (lane == 0) red[j * 4 + wave] = s;
    }
    __syncthreads();
    if (tid < 16 && o0 + tid < Cout)
      bslab[(int64_t)split * Cout + o0 + tid] =
          ((double)red[tid * 4] + (double)red[tid * 4 + 1]) + ((double)red[tid * 4 + 2] + (double)red[tid * 4 + 3]);
  }
}

// dW = sum over splits (fixed order, double); dbias likewise from the per-split bias partials
__global__ void convt_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t total,
                                         int nsplit, const double* __restrict__ bslab, float* __restrict__ dbias,
                                         int Cout) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    double v = 0.0;
    for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * total + i];
    out[i] = (float)v;
  }
  if (dbias && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) {
      double v = 0.0;
      for (int s = 0; s < nsplit; ++s) v += bslab[(int64_t)s * Cout + o];
      dbias[o] = (float)v;
    }
}

// dx[n, c, v] = sum over splits of slab[ks][n][c][v] (fixed order)
__global__ void convt_dx_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dx, int N, int64_t CS,
                                       int64_t xbs, int ksplit) {
  const int64_t total = (int64_t)N * CS;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = slab[i];
    for (int k = 1; k < ksplit; ++k) v += slab[(int64_t)k * total + i];
    dx[(i / CS) * xbs + i % CS] = v;
  }
}

// =============================== k2s2 backward on c8 operands (16-bit training flow) ===============================
// The activation gradient of the decoder travels as c8 only (train16.hip): dy16 is the c8 gradient of the up-sampled
// slot of a concat buffer (written by the data gradient of the decoder block's first convolution), x16 the c8 input the
// forward consumed.  Both kernels below are HBM-bound streams with the GEMM on v_mfma_f32_32x32x16_{bf16,f16}.
//
// ---- data gradient  dX[c, v] = sum_{t, o} W[c, o, t] * dY[o, 2v + t]:  M = Cin, K = 8 * Cout, N = voxels.
// A k-step of 16 is (t, two channel blocks): the c8 item of output voxel 2v + t in block ob IS the B fragment of lane
// (voxel v, k-half ob & 1) -- global -> register -> MFMA, dY is read exactly once per m-tile group.  The weights
// (rounded to the 16-bit type) are staged once per workgroup in A-fragment order.  The rows of an m-tile are ordered
// so that a lane ends up with whole channel blocks: row m = 8i + 4h + j holds channel 8 * (2 * (i >> 1) + h) +
// 4 * (i & 1) + j of the tile, i.e. accumulators 0..7 / 8..15 of lane half h are the channel blocks h / 2 + h -- two
// 16-byte stores per lane, 512 contiguous bytes per half-wave.
template <typename HT, int MTW>
__global__ __launch_bounds__(256) void convt_k2s2_bwd_data_h16_kernel(
    const HT* __restrict__ dy16, const float* __restrict__ w, HT* __restrict__ dx16, int Cin, int Cout, int D, int H, int W,
    int64_t ybs16, int64_t xbs16, int mt_per_wg) {
  using hx8 = typename H16<HT>::x8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  hx8* afrag = reinterpret_cast<hx8*>(lds_raw);  // [m-tile of this workgroup][k-step][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int CBout = (Cout + 7) / 8, CBin = (Cin + 7) / 8;
  const int npair = (CBout + 1) / 2, nks = 8 * npair;        // k-step s = pair * 8 + t
  const int mtiles = (Cin + 31) / 32;
  const int mt0 = blockIdx.y * mt_per_wg, nmt = min(mtiles, mt0 + mt_per_wg) - mt0;
  {
    // staging in the memory order of w ([Cin][Cout][t]: the (o, t) plane of one input channel is contiguous) with
    // 2-byte scatter writes into fragment order; every element written once, zeros past Cin / Cout
    HT* af16 = reinterpret_cast<HT*>(lds_raw);
    const int opad = npair * 16;
    const int total4 = nmt * 32 * opad * 2;                // float4 pieces: t = 0..3 / 4..7 of one (c, o)
    for (int e0 = tid; e0 < total4; e0 += 256 * 8) {       // 8 x 16-byte loads in flight per thread
      float4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        const int th = e & 1, o = (e >> 1) % opad;
        const int m = ((e >> 1) / opad) & 31, q = ((e >> 1) / opad) >> 5;
        const int i = m >> 3, hh = (m >> 2) & 1, j4 = m & 3;
        const int c = (mt0 + q) * 32 + 8 * (2 * (i >> 1) + hh) + 4 * (i & 1) + j4;
        const bool ok = e < total4 && c < Cin && o < Cout;
        v[u] = ok ? *reinterpret_cast<const float4*>(w + ((int64_t)c * Cout + o) * 8 + 4 * th) : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[u] = e < total4 ? ((q * nks + (o >> 4) * 8 + 4 * th) * 64 + ((o >> 3) & 1) * 32 + m) * 8 + (o & 7) : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) {                       // k-step = pair * 8 + t: consecutive t are 64 fragments apart
          af16[dst[u]] = (HT)v[u].x;
          af16[dst[u] + 64 * 8] = (HT)v[u].y;
          af16[dst[u] + 2 * 64 * 8] = (HT)v[u].z;
          af16[dst[u] + 3 * 64 * 8] = (HT)v[u].w;
        }
    }
  }
  __syncthreads();
  const int n = blockIdx.z;
  const hx8* yin = reinterpret_cast<const hx8*>(dy16 + (int64_t)n * ybs16);
  hx8* xo = reinterpret_cast<hx8*>(dx16 + (int64_t)n * xbs16);
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  const hx8 zero = {};
  for (int vt = blockIdx.x; (int64_t)vt * 128 < S; vt += gridDim.x) {
    const int v = vt * 128 + wave * 32 + l32;
    const bool vok = v < S;
    const int vc = min(v, S - 1);
    const int ix = vc % W, iy = (vc / W) % H, iz = vc / (W * H);
    const int64_t obase = ((int64_t)(2 * iz) * OH + 2 * iy) * OW + 2 * ix;
    f32x16 acc[MTW];
#pragma unroll
    for (int q = 0; q < MTW; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // two statically named register sets in ping-pong (an array indexed by `pair & 1` lives in scratch memory)
    hx8 b0[8], b1[8];
    auto fetch = [&](int pair, hx8 (&dst)[8]) {
      const int ob = 2 * pair + half;
      const bool ok = vok && ob < CBout;
      const hx8* src = yin + (int64_t)(ok ? ob : 0) * OS + obase;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        dst[t] = ok ? src[((int64_t)(t >> 2) * OH + ((t >> 1) & 1)) * OW + (t & 1)] : zero;
    };
    auto multiply = [&](int pair, const hx8 (&b)[8]) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int q = 0; q < MTW; ++q)
          if (q < nmt) acc[q] = H16<HT>::mfma(afrag[((int64_t)q * nks + pair * 8 + t) * 64 + lane], b[t], acc[q]);
      }
    };
    fetch(0, b0);
    for (int pair = 0; pair < npair; pair += 2) {
      if (pair + 1 < npair) fetch(pair + 1, b1);
      multiply(pair, b0);
      if (pair + 1 < npair) {
        if (pair + 2 < npair) fetch(pair + 2, b0);
        multiply(pair + 1, b1);
      }
    }
    if (vok) {
#pragma unroll
      for (int q = 0; q < MTW; ++q) {
        if (q >= nmt) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int cb = (mt0 + q) * 4 + 2 * u + half;
          if (cb < CBin) {
            hx8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (HT)(cb * 8 + j < Cin ? acc[q][u * 8 + j] : 0.f);
            xo[(int64_t)cb * S + v] = o;
          }
        }
      }
    }
  }
}

// ---- weight gradient  dW[c, o, t] = sum_v X[c, v] * dY[o, 2v + t]:  M = Cin, N = Cout (per t), K = voxels.
// Both operands need k = 16 x-adjacent voxels of ONE channel per lane while c8 keeps the 8 channels of a voxel
// together: as in conv3_bww_c8_kernel the tiles sit in LDS voxel-major (64-byte rows of 32 channels) and the fragments
// come out of `ds_read_b64_tr_b16`.  A workgroup = (32-channel o-tile, CT 32-channel c-tiles, voxel split); a tile is
// 2 x 32 input voxels (x rows) and their 2 x 4 x 64 output voxels; wave w owns t = 2w, 2w + 1 (z-parity a = w >> 1,
// y-parity b = w & 1, both x-parities).  dY is read once per c-tile group, X once per o-tile.
template <typename HT>
__device__ __forceinline__ typename H16<HT>::x8 tr_frag_rs(const unsigned char* p, int rowbytes4) {
  typedef short s16x4_ __attribute__((ext_vector_type(4)));
  typedef short s16x8_ __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4_ lds_s16x4_;
  const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(p));
  const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(p + rowbytes4));
  const s16x8_ v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename H16<HT>::x8, v);
}

template <typename HT, int CT>
__global__ __launch_bounds__(256) void convt_k2s2_bww_c8_kernel(
    const HT* __restrict__ x16, const HT* __restrict__ dy16, float* __restrict__ slab, int N, int Cin, int Cout, int D, int H,
    int W, int64_t xbs16, int64_t ybs16, int nsplit, int cgroups) {
  constexpr int TY = 2, TX = 32, NV = TY * TX;                 // 64 input voxels per tile (one z)
  constexpr int XROW = CT * 64;                                // bytes per voxel row of the x tile
  constexpr int XI = NV * CT * 4, DI = (2 * 2 * TY * 2 * TX) * 4;   // 16-byte items: x tile, dy tile (512 voxels)
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) uint4 xs[XI];
  __shared__ __attribute__((aligned(16))) uint4 ds[DI];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int split = blockIdx.x, otile = blockIdx.y, cg = blockIdx.z;
  const int CBin = (Cin + 7) / 8, CBout = (Cout + 7) / 8;
  const int S = D * H * W, OH = 2 * H, OW = 2 * W, OS = S * 8;
  const int ty_tiles = (H + TY - 1) / TY, tx_tiles = (W + TX - 1) / TX;
  const int tiles_per_n = D * ty_tiles * tx_tiles, ntiles = N * tiles_per_n;
  const int cb0 = cg * CT * 4;                                  // first input channel block of this workgroup
  const int nbx = max(0, min(CT * 4, CBin - cb0)), nbd = max(0, min(4, CBout - 4 * otile));

  // transposed-read lane bases (see conv3_bww_c8_kernel): lane 4q + p of a 16-lane group addresses voxel row q,
  // channels 4p .. 4p+3 of the group's 16-channel half; groups 2, 3 take the voxels 8 .. 15 of the k-step
  const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int xl = (8 * (grp >> 1) + tq) * XROW + (grp & 1) * 32 + tp * 8;
  const int dl = (8 * (grp >> 1) + tq) * 128 + (grp & 1) * 32 + tp * 8;   // dy rows of one x-parity are 128 B apart
  const int a = wave >> 1, b = wave & 1;

  f32x16 acc[2][CT];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < CT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;

  // the next tile travels global -> registers while the current one is multiplied out of LDS
  constexpr int XPER = XI / 256, DPER = DI / 256;   // CT and 8 items per thread
  uint4 xr[XPER], dr[DPER];
  auto fetch = [&](int tile, bool live) {
    int t = tile;
    const int n = t / tiles_per_n;
    t -= n * tiles_per_n;
    const int txt = t % tx_tiles;
    t /= tx_tiles;
    const int tyt = t % ty_tiles, z0 = t / ty_tiles;
    const int y0 = tyt * TY, x0 = txt * TX;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x16 + (int64_t)n * xbs16 + (int64_t)cb0 * S * 8), 0, live ? nbx * S * 16 : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dy16 + (int64_t)n * ybs16 + (int64_t)(4 * otile) * OS * 8), 0, live ? nbd * OS * 16 : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < XPER; ++k) {           // item e = (voxel e / (4 CT), channel block e % (4 CT))
      const int e = tid + 256 * k;
      const int vx = e / (CT * 4), cbl = e - vx * (CT * 4);
      const int yy = vx / TX, xx = vx - yy * TX;
      const bool ok = y0 + yy < H && x0 + xx < W && cbl < nbx;
      xr[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
          rx, ok ? (unsigned)(cbl * S + (z0 * H + y0 + yy) * W + x0 + xx) * 16u : OOB, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < DPER; ++k) {           // dy tile voxel-major: [az][oy][ox][4 blocks], oy < 2 TY, ox < 2 TX
      const int e = tid + 256 * k;
      const int vo = e >> 2, cbl = e & 3;
      const int ox = vo % (2 * TX), oy = (vo / (2 * TX)) % (2 * TY), az = vo / (4 * TX * TY);
      const int gy = 2 * y0 + oy, gx = 2 * x0 + ox;
      const bool ok = gy < OH && gx < OW && cbl < nbd;
      dr[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
          rd, ok ? (unsigned)(cbl * OS + ((2 * z0 + az) * OH + gy) * OW + gx) * 16u : OOB, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < XPER; ++k) xs[tid + 256 * k] = xr[k];
#pragma unroll
    for (int k = 0; k < DPER; ++k) ds[tid + 256 * k] = dr[k];
  };
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(xs) + xl;
  const unsigned char* db = reinterpret_cast<const unsigned char*>(ds) + dl;
  if (split < ntiles) {
    fetch(split, true);
    commit();
  }
  __syncthreads();
  for (int tile = split; tile < ntiles; tile += nsplit) {
    const bool more = tile + nsplit < ntiles;
    fetch(more ? tile + nsplit : tile, more);   // (zero-sized descriptors after the last tile)
#pragma unroll
    for (int yy = 0; yy < TY; ++yy)
#pragma unroll
      for (int xk = 0; xk < 2; ++xk) {
        typename H16<HT>::x8 af[CT];
#pragma unroll
        for (int q = 0; q < CT; ++q) af[q] = tr_frag_rs<HT>(xb + (yy * TX + 16 * xk) * XROW + q * 64, 4 * XROW);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          // output voxel (az = a, oy = 2 yy + b, ox = 2 (16 xk + k) + c), 64 bytes per voxel row
          const typename H16<HT>::x8 bf =
              tr_frag_rs<HT>(db + (((a * 2 * TY + 2 * yy + b) * 2 * TX) + 2 * 16 * xk + c) * 64, 4 * 128);
#pragma unroll
          for (int q = 0; q < CT; ++q) acc[c][q] = H16<HT>::mfma(af[q], bf, acc[c][q]);
        }
      }
    __syncthreads();   // every wave is done reading this tile
    if (more) commit();
    __syncthreads();
  }
  // partial dW -> slab[split][t][c][o] (lane = output channel: 32 consecutive floats per store)
  float* sl = slab + (int64_t)split * 8 * Cin * Cout;
  const int o = otile * 32 + l32;
  if (o < Cout) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int t = a * 4 + b * 2 + c;
#pragma unroll
      for (int q = 0; q < CT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (cg * CT + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (ci < Cin) sl[((int64_t)t * Cin + ci) * Cout + o] = acc[c][q][r];
        }
    }
  }
}

// dW[c][o][t] = unscale * sum over splits of slab[split][t][c][o], in a fixed order: a block owns 64 consecutive slab
// positions; its 4 waves take the splits s = wave, wave + 4, ... (coalesced 256-byte reads), then the four partial
// sums are added in wave order
__global__ __launch_bounds__(256) void convt_slab_reduce_t_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                  int Cin, int Cout, int nsplit, float unscale,
                                                                  int* __restrict__ oflag) {
  __shared__ double part[4][64];
  const int64_t plane = (int64_t)Cin * Cout * 8, cc = (int64_t)Cin * Cout;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t j = blockIdx.x * 64ll + lane;              // slab position t * Cin * Cout + (c * Cout + o)
  double v = 0.0;
  if (j < plane) {
    int s = wv;
    for (; s + 12 < nsplit; s += 16) {                     // four loads in flight
      const float a0 = slab[(int64_t)s * plane + j], a1 = slab[(int64_t)(s + 4) * plane + j];
      const float a2 = slab[(int64_t)(s + 8) * plane + j], a3 = slab[(int64_t)(s + 12) * plane + j];
      v += (double)a0; v += (double)a1; v += (double)a2; v += (double)a3;
    }
    for (; s < nsplit; s += 4) v += (double)slab[(int64_t)s * plane + j];
  }
  part[wv][lane] = v;
  __syncthreads();
  if (wv == 0 && j < plane) {
    const double tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    const int t = (int)(j / cc);
    const float r = (float)(tot * (double)unscale);
    dw[(j - (int64_t)t * cc) * 8 + t] = r;
    report_nonfinite(r, oflag);
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
static int convt_bww_mt(const m355_conv3d_desc* d) { return d->Cin > 64 ? 4 : 2; }
static int convt_nsplit(const m355_conv3d_desc* d) {
  const int64_t tiles = ceil_div(d->Cin, 32 * convt_bww_mt(d)) * ceil_div(d->Cout, 16);
  const int64_t nsteps = ceil_div((int64_t)d->N * d->D * d->H * d->W, 64);
  int64_t ns = std::max<int64_t>(1, 2 * num_cus() / tiles);  // persistent: 2 workgroups per CU, one round
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
  const int64_t generic = (int64_t)d->Cout * ceil_div(OS, DBIAS_CHUNK) * 8;              // launch_dbias partials
  const int64_t fused = is_k2s2(d) ? (int64_t)convt_nsplit(d) * d->Cout * 8 : 0;        // bslab[split][Cout]
  return (size_t)round_up(std::max(generic, fused), 256);
}

// forward tile: the largest voxel tile whose [Cin][NVT] x slab fits 64 KB of LDS (0: none does)
static int convt_fwd_nvt(const m355_conv3d_desc* d) {
  const int64_t cin2 = round_up(d->Cin, 16);  // whole weight chunks (KC = 8 k-pairs)
  for (int nvt : {128, 64, 32})  // 256 needs > 256 registers: one workgroup per CU, no overlap
    if (cin2 * nvt * 4 <= 65536) return nvt;
  return 0;
}

// data-gradient plan: voxel tile, channel tile, and a split of the K = 8*Cout contraction when the level has
// too few voxels to fill the chip (deep levels: few, long, latency-bound workgroups otherwise)
struct ConvtBwdPlan {
  int mt, nvt, ksplit, o_per_split;
  size_t slab_bytes;
};
static ConvtBwdPlan plan_convt_bwd(const m355_conv3d_desc* d) {
  ConvtBwdPlan p{};
  const int64_t S = (int64_t)d->D * d->H * d->W;
  p.mt = d->Cin > 64 ? 4 : 2;  // all input channels of a standard level in one pass over dY
  const int64_t cblocks = ceil_div(d->Cin, 32 * p.mt);
  p.nvt = 256;
  while (p.nvt > 64 && ceil_div(S, p.nvt) * cblocks * d->N < 512) p.nvt >>= 1;
  const int64_t wgs = ceil_div(S, p.nvt) * cblocks * d->N;
  int64_t ks = 1;
  if (wgs < 256) ks = std::min<int64_t>(std::min<int64_t>(8, ceil_div(512, wgs)), std::max(1, d->Cout / 16));
  p.o_per_split = (int)round_up(ceil_div(d->Cout, ks), 4);
  p.ksplit = (int)ceil_div(d->Cout, p.o_per_split);
  p.slab_bytes = p.ksplit > 1 ? (size_t)round_up((int64_t)p.ksplit * d->N * d->Cin * S * 4, 256) : 0;
  return p;
}

extern "C" size_t m355_conv_transpose3d_workspace(const m355_conv3d_desc* d) {
  if (!d) return 0;
  const size_t bwd = is_k2s2(d) ? plan_convt_bwd(d).slab_bytes : 0;
  return std::max(convt_slab_bytes(d) + convt_dbias_bytes(d), bwd);
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
  // M355_COMPUTE_F32X3: the same GEMM on the bf16 matrix pipe through the exact three-way operand split (conv3d_f32x3.hip)
  const bool x3 = d->compute == M355_COMPUTE_F32X3 && tuning().f32x3 && tuning().f32x3_convt && convt_fwd_x3_nvt(d->Cin) != 0;
  const int nvt = x3 ? convt_fwd_x3_nvt(d->Cin) : convt_fwd_nvt(d);
  if (is_k2s2(d) && nvt && (ybs % 2 == 0) && ((uintptr_t)y & 7) == 0 && convt_fits_i32(d)) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    const int64_t vox_tiles = ceil_div(S, nvt) * d->N;
    const int mtiles = (int)ceil_div(d->Cout, 4);
    // x is re-read once per m-tile group: split M only as far as needed to fill the chip
    const int64_t groups = std::max<int64_t>(1, std::min<int64_t>(ceil_div(768, vox_tiles), ceil_div(mtiles, 4)));
    const int mt_per_wg = (int)round_up(ceil_div(mtiles, groups), 4);
    dim3 grid((unsigned)ceil_div(S, nvt), (unsigned)ceil_div(mtiles, mt_per_wg), (unsigned)d->N);
    if (x3) {
      launch_convt_fwd_x3(nvt, grid, x, w, bias, y, d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, mt_per_wg, st);
      return check_launch("convt_k2s2_fwd_x3");
    }
    const size_t lds = (size_t)round_up(d->Cin, 16) * nvt * 4;
#define M355_CONVT_FWD(NVT)                                                                                   \
  hipLaunchKernelGGL(convt_k2s2_fwd_mfma_kernel<NVT>, grid, dim3(256), lds, st, x, w, bias, y, d->Cin, d->Cout, \
                     d->D, d->H, d->W, xbs, ybs, mt_per_wg)
    switch (nvt) {
      case 128: M355_CONVT_FWD(128); break;
      case 64: M355_CONVT_FWD(64); break;
      default: M355_CONVT_FWD(32); break;
    }
#undef M355_CONVT_FWD
    return check_launch("convt_k2s2_fwd");
  }
  const int64_t total = (int64_t)d->N * d->Cout * OD * OH * OW;
  const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 65535);
  hipLaunchKernelGGL(convt_direct_fwd_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, y, d->N,
                     d->Cin, d->Cout, d->D, d->H, d->W, OD, OH, OW, d->k, d->stride, d->pad, xbs, ybs);
  return check_launch("convt_direct_fwd");
}

extern "C" int m355_conv_transpose3d_fwd_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                             const float* w, const float* bias, void* y16, int64_t y16_batch_stride,
                                             int32_t compute, void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_fwd_h16")) return rc;
  M355_REQUIRE(x16 && w && y16, M355_EINVALID_ARG, "conv_transpose3d_fwd_h16: null pointer");
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "conv_transpose3d_fwd_h16: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  const int nvt = std::min(convt_fwd_nvt(d), 64);  // two m-tiles per wave: 64 voxels keep two workgroups per CU
  M355_REQUIRE(is_k2s2(d) && nvt && convt_fits_i32(d), M355_EUNSUPPORTED,
               "conv_transpose3d_fwd_h16: only kernel_size 2 / stride 2 / padding 0 has a c8 kernel");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t xbs = dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8);
  const int64_t ybs = dense_or(y16_batch_stride, c8_blocks(d->Cout) * S * 8 * 8);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)y16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "conv_transpose3d_fwd_h16: c8 tensor not 16B aligned");
  hipStream_t st = (hipStream_t)stream;
  {
    // large levels: 16-bit MFMA kernel (weights staged once per workgroup; worth it from ~16k voxels per sample)
    const int ks_n = (int)ceil_div(d->Cin, 16);
    const int mtiles = 2 * (int)c8_blocks(d->Cout);
    if (S >= 16384 && ks_n <= 8 && tuning().convt_h16) {
      const int mt_per_wg = std::max(1, std::min(mtiles, 64 / ks_n));          // <= 64 KB of A fragments
      const int groups = (int)ceil_div(mtiles, mt_per_wg);
      const int ng = (S / 256) * groups * d->N >= 512 ? 2 : 1;
      const int64_t vox_tiles = ceil_div(S, 128 * ng);
      const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(vox_tiles, ceil_div(512, (int64_t)groups * d->N)));
      dim3 grid((unsigned)gx, (unsigned)groups, (unsigned)d->N);
      const size_t lds = (size_t)mt_per_wg * ks_n * 1024;
#define M355_CONVT_H16(HT, KS, NG)                                                                                  \
  hipLaunchKernelGGL((convt_k2s2_fwd_h16_kernel<HT, KS, NG>), grid, dim3(256), lds, st, (const HT*)x16, w, bias,     \
                     (HT*)y16, d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, mt_per_wg)
#define M355_CONVT_H16_T(HT)                                                         \
  if (ks_n <= 4) { if (ng == 2) M355_CONVT_H16(HT, 4, 2); else M355_CONVT_H16(HT, 4, 1); } \
  else { if (ng == 2) M355_CONVT_H16(HT, 8, 2); else M355_CONVT_H16(HT, 8, 1); }
      if (compute == M355_COMPUTE_BF16) { M355_CONVT_H16_T(__bf16) } else { M355_CONVT_H16_T(_Float16) }
#undef M355_CONVT_H16_T
#undef M355_CONVT_H16
      return check_launch("convt_k2s2_fwd_h16");
    }
  }
  const int64_t vox_tiles = ceil_div(S, nvt) * d->N;
  const int mpairs = (int)ceil_div(ceil_div(d->Cout, 4), 2);
  const int64_t groups = std::max<int64_t>(1, std::min<int64_t>(ceil_div(768, vox_tiles), ceil_div(mpairs, 4)));
  const int mp_per_wg = (int)round_up(ceil_div(mpairs, groups), 4);
  dim3 grid((unsigned)ceil_div(S, nvt), (unsigned)ceil_div(mpairs, mp_per_wg), (unsigned)d->N);
  const size_t lds = (size_t)round_up(d->Cin, 16) * nvt * 4;
#define M355_CONVT_C8(NVT, HT)                                                                                       \
  hipLaunchKernelGGL((convt_k2s2_fwd_c8_kernel<NVT, HT>), grid, dim3(256), lds, st, (const HT*)x16, w, bias, (HT*)y16, \
                     d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, mp_per_wg)
  if (compute == M355_COMPUTE_BF16) {
    if (nvt == 64) M355_CONVT_C8(64, __bf16); else M355_CONVT_C8(32, __bf16);
  } else {
    if (nvt == 64) M355_CONVT_C8(64, _Float16); else M355_CONVT_C8(32, _Float16);
  }
#undef M355_CONVT_C8
  return check_launch("convt_k2s2_fwd_c8");
}

extern "C" int m355_conv_transpose3d_bwd_data(const m355_conv3d_desc* d, const float* dy,
                                              const float* w, float* dx, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_bwd_data")) return rc;
  M355_REQUIRE(dy && w && dx, M355_EINVALID_ARG, "conv_transpose3d_bwd_data: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OD = convt_out(d->D, d), OH = convt_out(d->H, d), OW = convt_out(d->W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->Cin * d->D * d->H * d->W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * OD * OH * OW);
  if (is_k2s2(d) && (ybs % 2 == 0) && ((uintptr_t)dy & 7) == 0 && convt_fits_i32(d)) {
    const int64_t S = (int64_t)d->D * d->H * d->W;
    const ConvtBwdPlan bp = plan_convt_bwd(d);
    const int mt = bp.mt, nvt = bp.nvt, ksplit = bp.ksplit;
    float* slab = nullptr;
    if (ksplit > 1) {
      M355_REQUIRE(workspace && workspace_bytes >= bp.slab_bytes, M355_EWORKSPACE,
                   "conv_transpose3d_bwd_data: workspace too small (%zu < %zu)", workspace_bytes, bp.slab_bytes);
      slab = (float*)workspace;
    }
    dim3 grid((unsigned)ceil_div(S, nvt), (unsigned)ceil_div(d->Cin, 32 * mt), (unsigned)(d->N * ksplit));
#define M355_CONVT_BWD(NVT, MT)                                                                                \
  hipLaunchKernelGGL((convt_k2s2_bwd_data_mfma_kernel<NVT, MT>), grid, dim3(256), 0, st, dy, w, dx, d->Cin,    \
                     d->Cout, d->D, d->H, d->W, xbs, ybs, ksplit, bp.o_per_split, slab)
    if (mt == 2) {
      if (nvt == 256) M355_CONVT_BWD(256, 2); else if (nvt == 128) M355_CONVT_BWD(128, 2); else M355_CONVT_BWD(64, 2);
    } else {
      if (nvt == 256) M355_CONVT_BWD(256, 4); else if (nvt == 128) M355_CONVT_BWD(128, 4); else M355_CONVT_BWD(64, 4);
    }
#undef M355_CONVT_BWD
    if (ksplit > 1) {
      const int64_t total = (int64_t)d->N * d->Cin * S;
      hipLaunchKernelGGL(convt_dx_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 2048)), dim3(256),
                         0, st, slab, dx, d->N, (int64_t)d->Cin * S, xbs, ksplit);
    }
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
    double* bslab = nullptr;
    if (dbias) {
      M355_REQUIRE(workspace_bytes >= need + convt_dbias_bytes(d), M355_EWORKSPACE,
                   "conv_transpose3d_bwd_weight: workspace too small for the bias gradient");
      bslab = (double*)((char*)workspace + need);
    }
    const int mt = convt_bww_mt(d);
    dim3 grid((unsigned)nsplit, (unsigned)ceil_div(d->Cout, 16), (unsigned)ceil_div(d->Cin, 32 * mt));
    if (mt == 2)
      hipLaunchKernelGGL(convt_k2s2_bwd_weight_mfma_kernel<2>, grid, dim3(256), 0, st, x, dy, slab, bslab, d->N,
                         d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, nsplit);
    else
      hipLaunchKernelGGL(convt_k2s2_bwd_weight_mfma_kernel<4>, grid, dim3(256), 0, st, x, dy, slab, bslab, d->N,
                         d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, nsplit);
    const int64_t total = (int64_t)d->Cin * d->Cout * 8;
    hipLaunchKernelGGL(convt_slab_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 1024)),
                       dim3(256), 0, st, slab, dw, total, nsplit, bslab, dbias, d->Cout);
    return check_launch("conv_transpose3d_bwd_weight");
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

// ---------------------------------------------------------------- c8 backward entry points (16-bit training flow)
static bool convt_c8_bwd_ok(const m355_conv3d_desc* d) {
  // the MFMA kernels: k2 s2, all weights of an m-tile group within 64 KB of A fragments, 32-bit buffer offsets
  const int64_t S = (int64_t)d->D * d->H * d->W;
  return is_k2s2(d) && 8 * ceil_div(c8_blocks(d->Cout), 2) <= 64 && S * 8 * 16 * 4 < (1ll << 31) &&
         S * 16 * std::min<int64_t>(16, c8_blocks(d->Cin)) < (1ll << 31);
}
static int convt_c8_ct(const m355_conv3d_desc* d) { const int ct = (int)ceil_div(d->Cin, 32); return ct >= 4 ? 4 : (ct >= 2 ? 2 : 1); }
static int convt_c8_nsplit(const m355_conv3d_desc* d) {
  const int64_t ntiles = (int64_t)d->N * d->D * ceil_div(d->H, 2) * ceil_div(d->W, 32);
  const int64_t groups = ceil_div(d->Cout, 32) * ceil_div(ceil_div(d->Cin, 32), convt_c8_ct(d));
  // (every split writes a slab of 8 * Cin * Cout floats that the reduction reads again: one workgroup per CU)
  const int64_t wgs = (int64_t)(tuning().convt_wgs ? tuning().convt_wgs : 1) * num_cus();
  return (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, wgs / groups));
}

extern "C" int32_t m355_conv_transpose3d_h16_bwd_supported(const m355_conv3d_desc* d) {
  return d && d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->D > 0 && d->H > 0 && d->W > 0 && convt_c8_bwd_ok(d) ? 1 : 0;
}

extern "C" size_t m355_conv_transpose3d_h16_bwd_workspace(const m355_conv3d_desc* d) {
  if (!d || !m355_conv_transpose3d_h16_bwd_supported(d)) return 0;
  const int64_t OS = (int64_t)d->D * d->H * d->W * 8;
  return (size_t)round_up((int64_t)convt_c8_nsplit(d) * 8 * d->Cin * d->Cout * 4, 256) + dbias_c8_ws_bytes(d->N, d->Cout, OS);
}

extern "C" int m355_conv_transpose3d_bwd_data_h16(const m355_conv3d_desc* d, const void* dy16, int64_t dy16_batch_stride,
                                                  const float* w, void* dx16, int64_t dx16_batch_stride, int32_t compute,
                                                  void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_bwd_data_h16")) return rc;
  M355_REQUIRE(dy16 && w && dx16, M355_EINVALID_ARG, "conv_transpose3d_bwd_data_h16: null pointer");
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "conv_transpose3d_bwd_data_h16: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  M355_REQUIRE(convt_c8_bwd_ok(d), M355_EUNSUPPORTED,
               "conv_transpose3d_bwd_data_h16: only kernel_size 2 / stride 2 / padding 0 with Cout <= 128 has a c8 kernel "
               "(m355_conv_transpose3d_h16_bwd_supported)");
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t ybs = dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8 * 8);
  const int64_t xbs = dense_or(dx16_batch_stride, c8_blocks(d->Cin) * S * 8);
  M355_REQUIRE((((uintptr_t)dy16 | (uintptr_t)dx16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "conv_transpose3d_bwd_data_h16: c8 tensor not 16B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nks = 8 * (int)ceil_div(c8_blocks(d->Cout), 2);
  const int mtiles = (int)ceil_div(d->Cin, 32);
  int mt_per_wg = std::max(1, std::min(std::min(mtiles, 64 / nks), 4));
  if (mt_per_wg == 3) mt_per_wg = 2;
  const int groups = (int)ceil_div(mtiles, mt_per_wg);
  const int64_t vox_tiles = ceil_div(S, 128);
  // one residency of workgroups (2 per CU), each walking its share of the voxel tiles: the weights are staged once
  const int64_t wgs = (int64_t)(tuning().convt_wgs ? tuning().convt_wgs : 2) * num_cus();
  // (weights are staged per workgroup: at least four voxel tiles each on the small levels)
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(vox_tiles, 4), ceil_div(wgs, (int64_t)groups * d->N)));
  dim3 grid((unsigned)gx, (unsigned)groups, (unsigned)d->N);
  const size_t lds = (size_t)mt_per_wg * nks * 1024;
#define M355_CTBD(HT, MTW)                                                                                            \
  hipLaunchKernelGGL((convt_k2s2_bwd_data_h16_kernel<HT, MTW>), grid, dim3(256), lds, st, (const HT*)dy16, w, (HT*)dx16, \
                     d->Cin, d->Cout, d->D, d->H, d->W, ybs, xbs, mt_per_wg)
#define M355_CTBD_T(HT)                                                  \
  if (mt_per_wg == 1) M355_CTBD(HT, 1); else if (mt_per_wg == 2) M355_CTBD(HT, 2); else M355_CTBD(HT, 4);
  if (compute == M355_COMPUTE_BF16) { M355_CTBD_T(__bf16) } else { M355_CTBD_T(_Float16) }
#undef M355_CTBD_T
#undef M355_CTBD
  return check_launch("convt_k2s2_bwd_data_h16");
}

extern "C" int m355_conv_transpose3d_bwd_weight_h16(const m355_conv3d_desc* d, const void* x16, int64_t x16_batch_stride,
                                                    const void* dy16, int64_t dy16_batch_stride, float* dw, float* dbias,
                                                    float grad_unscale, int32_t compute, void* workspace,
                                                    size_t workspace_bytes, void* stream) {
  if (int rc = validate_convt(d, "conv_transpose3d_bwd_weight_h16")) return rc;
  M355_REQUIRE(x16 && dy16 && dw && workspace, M355_EINVALID_ARG, "conv_transpose3d_bwd_weight_h16: null pointer");
  M355_REQUIRE(compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16, M355_EINVALID_ARG,
               "conv_transpose3d_bwd_weight_h16: compute must be M355_COMPUTE_BF16 or M355_COMPUTE_F16");
  M355_REQUIRE(convt_c8_bwd_ok(d), M355_EUNSUPPORTED,
               "conv_transpose3d_bwd_weight_h16: unsupported geometry (m355_conv_transpose3d_h16_bwd_supported)");
  M355_REQUIRE(workspace_bytes >= m355_conv_transpose3d_h16_bwd_workspace(d), M355_EWORKSPACE,
               "conv_transpose3d_bwd_weight_h16: workspace too small (%zu < %zu)", workspace_bytes,
               m355_conv_transpose3d_h16_bwd_workspace(d));
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t xbs = dense_or(x16_batch_stride, c8_blocks(d->Cin) * S * 8);
  const int64_t ybs = dense_or(dy16_batch_stride, c8_blocks(d->Cout) * S * 8 * 8);
  M355_REQUIRE((((uintptr_t)x16 | (uintptr_t)dy16) & 15) == 0 && xbs % 8 == 0 && ybs % 8 == 0, M355_EINVALID_ARG,
               "conv_transpose3d_bwd_weight_h16: c8 tensor not 16B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int ct = convt_c8_ct(d), nsplit = convt_c8_nsplit(d);
  const int cgroups = (int)ceil_div(ceil_div(d->Cin, 32), ct);
  float* slab = (float*)workspace;
  dim3 grid((unsigned)nsplit, (unsigned)ceil_div(d->Cout, 32), (unsigned)cgroups);
#define M355_CTBW(HT, CT)                                                                                               \
  hipLaunchKernelGGL((convt_k2s2_bww_c8_kernel<HT, CT>), grid, dim3(256), 0, st, (const HT*)x16, (const HT*)dy16, slab,  \
                     d->N, d->Cin, d->Cout, d->D, d->H, d->W, xbs, ybs, nsplit, cgroups)
#define M355_CTBW_T(HT) if (ct == 1) M355_CTBW(HT, 1); else if (ct == 2) M355_CTBW(HT, 2); else M355_CTBW(HT, 4);
  if (compute == M355_COMPUTE_BF16) { M355_CTBW_T(__bf16) } else { M355_CTBW_T(_Float16) }
#undef M355_CTBW_T
#undef M355_CTBW
  const int64_t total = (int64_t)d->Cin * d->Cout * 8;
  hipLaunchKernelGGL(convt_slab_reduce_t_kernel, dim3((unsigned)ceil_div(total, 64)), dim3(256), 0, st, slab, dw, d->Cin,
                     d->Cout, nsplit, grad_unscale, grad_unscale != 1.f ? overflow_flag() : nullptr);
  if (dbias) {
    const size_t slab_b = (size_t)round_up((int64_t)nsplit * 8 * d->Cin * d->Cout * 4, 256);
    if (int rc = launch_dbias_c8(dy16, ybs, dbias, d->N, d->Cout, S * 8, compute, grad_unscale, (char*)workspace + slab_b, st))
      return rc;
  }
  return check_launch("conv_transpose3d_bwd_weight_h16");
}
