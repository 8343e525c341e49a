// Epilogue of the 16-bit convolution kernels that write their output tile in the c8 layout (conv3d_h16.hip).
#pragma once
#include "conv3d_common.hpp"

namespace m355 {

// ---- c8 epilogue: the output tile of one wave as c8 items (the next pass reads the tensor in c8 anyway) ----
// C/D layout: lane = voxel (l32) x channel-half: registers 4q..4q+3 of a lane are channels 8q + 4*half + 0..3 of
// its voxel.  One v_permlane32_swap per dword hands the lower lanes the upper lanes' half of block q and the
// upper lanes the lower lanes' half of block q+1, so every lane owns ONE complete 8-channel item (lower
// lanes: block q, upper lanes: block q+1) and stores it with a single 16-byte instruction: 8 store
// instructions per lane and tile (NTW = 4) instead of the 64 dword stores of the fp32 NCDHW epilogue, which
// was the largest fixed cost of a short-K item (store-issue bound, ~9k cycles of a ~24k-cycle item).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NTW, int GY, typename HT>
__device__ __forceinline__ void store_conv_tile_c8(const f32x16 (&acc)[NTW], HT* __restrict__ dst16,
                                                   const float* __restrict__ bias, int o0, int Cout, int z, int y0,
                                                   int xg, int ly, int half, int H, int W, int64_t S, bool lane_ok,
                                                   float* __restrict__ stat) {
  // Round 4: this tail is pure vector-ALU work (a wave64 instruction costs 4 cycles, and the edge-layer kernel -- 28 MFMAs
  // per tile -- spent most of its time here): bias add, statistics and the out-of-volume mask run on register PAIRS
  // (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: half the instructions, the same roundings), the mask is a multiply by
  // 1 / 0 instead of 16 selects per group.
  using hx4 = typename H16<HT>::x4;
  const int iHW = H * W;
  const int ob = o0 + 4 * half;
  f32x2 bb[8];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = ob + (r & 3) + 8 * (r >> 2);
    bb[r >> 1][r & 1] = (bias && o < Cout) ? bias[o] : 0.f;     // padded channels of the last block stay exactly zero
  }
  f32x2 s1[8], s2[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) s1[r] = s2[r] = f32x2{0.f, 0.f};
  const int CBout = (Cout + 7) >> 3;
  uint4* base = reinterpret_cast<uint4*>(dst16);
#pragma unroll
  for (int g = 0; g < NTW; ++g) {
    const int yg = y0 + g * GY + ly;
    const bool ok = lane_ok && yg < H;
    f32x2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = f32x2{acc[g][2 * r], acc[g][2 * r + 1]} + bb[r];
    if (stat) {
      const float okf = ok ? 1.f : 0.f;
      const f32x2 mk = f32x2{okf, okf};
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f32x2 t = v[r] * mk;          // (finite values: x * 1 = x, x * 0 = 0 -- as the select it replaces)
        s1[r] += t;
        s2[r] = __builtin_elementwise_fma(t, t, s2[r]);
      }
    }
    const int64_t vox = (int64_t)z * iHW + (int64_t)yg * W + xg;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {  // block pairs (0, 1) and (2, 3) of the 32-channel tile
      hx4 lo, hi;                     // this lane's 4 channels of block 2qp and of block 2qp + 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = (HT)v[(8 * qp + j) >> 1][(8 * qp + j) & 1];
        hi[j] = (HT)v[(8 * qp + 4 + j) >> 1][(8 * qp + 4 + j) & 1];
      }
      uint2 X = __builtin_bit_cast(uint2, lo), Y = __builtin_bit_cast(uint2, hi);
      auto r0 = __builtin_amdgcn_permlane32_swap(X.x, Y.x, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(X.y, Y.y, false, false);
      // lower lanes: (own block 2qp ch 0-3 | partner's ch 4-7); upper lanes: (partner's block 2qp+1 ch 0-3 | own ch 4-7)
      const uint4 item = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      const int cb = (o0 >> 3) + 2 * qp + half;
      if (ok && cb < CBout) base[(int64_t)cb * S + vox] = item;
    }
  }
  if (stat) {
    float a[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a[r] = s1[r >> 1][r & 1];
      a[16 + r] = s2[r >> 1][r & 1];
    }
    const int l32 = opaque((int)threadIdx.x) & 31;   // (recomputed, not hoisted out of a persistent item loop: conv3d_common.hpp)
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {
      const bool up = (l32 & h) != 0;
#pragma unroll
      for (int i = 0; i < h; ++i) {
        const float send = up ? a[i] : a[i + h];
        const float keep = up ? a[i + h] : a[i];
        a[i] = keep + __shfl_xor(send, h, 64);
      }
    }
    const int r = l32 & 15, q = l32 >> 4;
    const int o = o0 + 4 * (opaque((int)threadIdx.x >> 5) & 1) + (r & 3) + 8 * (r >> 2);
    if (o < Cout) stat[(int64_t)o * 2 + q] = a[0];
  }
}

}  // namespace m355
