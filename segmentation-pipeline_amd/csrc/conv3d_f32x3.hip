// fp32 convolution on the bf16 matrix pipe: M355_COMPUTE_F32X3 (3x3x3 / stride 1 / pad 1, forward and data gradient of
// nn.Conv3d in Block3d, /root/reference/segmentation_pipeline/models/components.py:48-56).
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate and this chip clocks both down under load
// (DESIGN 4.6), so the fp32 layers sit at the power ceiling of that instruction.  An fp32 number splits EXACTLY into three
// bf16 numbers (8 + 8 + 8 significant bits):
//     x = hi + mid + lo,   hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid            (every step exact in fp32)
// and a product of two bf16 values is exact in the MFMA's fp32 accumulator, so
//     x * w = hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi) + [mid*lo + lo*mid + lo*lo <= 2^-24 |x*w|]
// Six v_mfma_f32_32x32x16_bf16 (K = 16 each) stand for eight fp32 MFMAs (K = 2 each) at a sixteenth of the cost per K:
// 2.7x fewer matrix-core cycles for the same sum, with the dropped terms below half an ulp of each product.  Measured
// against an fp64 convolution the result is as accurate as the fp32 MFMA kernel's (max |err| / max |y| 4e-7 .. 1.2e-6
// for both on the cfg2 layers: the error is the fp32 ACCUMULATION's either way; profiles/r04_f32x3_accuracy.txt).
//
// Data flow: the activations stay fp32 NCDHW in HBM.  A workgroup stages the halo tile of an 8-channel chunk through
// registers (coalesced dword loads along x through a buffer descriptor that returns 0 for padding), splits every
// value in registers (5.5 vector-ALU ops per element, once per staged element) and writes three bf16 planes to LDS in
// the c8 form [plane][voxel][8 channels] -- one 16-byte item per voxel and plane, which IS the B fragment of the MFMA
// for a lane (voxel, k-half).  K = 16 of one MFMA = 8 channels x 2 TAPS: the lower lane half reads tap t0, the upper
// half tap t1 of a pair (27 taps = 14 pairs, the last one half empty), so a chunk is 8 channels and its three planes
// fit LDS twice per CU (58.8 KB per workgroup at NTW = 4).  The weights are split and laid out in fragment order once
// per optimizer step (pack_w3_x3_kernel: [channel tile][chunk][pair][plane][lane] x 16 B) and stream from L2 straight
// into registers, two pairs ahead of their use -- their three planes would not fit LDS next to the activations.
#include "conv3d_common.hpp"

namespace m355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_PAIRS = 14;
// tap (dz*9 + dy*3 + dx) of lane half `half` of pair `pair`; -1 = no tap (zero weights)
//   pairs 0..8   (dz, dy) = (pair / 3, pair % 3): dx = 0 | 1         -> the halves are one voxel apart along x
//   pairs 9..11  dz = pair - 9, dx = 2: dy = 0 | 1                   -> one tile row apart
//   pair 12      dy = 2, dx = 2: dz = 0 | 1                          -> one tile plane apart
//   pair 13      (2, 2, 2) | none
__host__ __device__ constexpr int x3_pair_tap(int pair, int half) {
  return pair < 9 ? pair * 3 + half : (pair < 12 ? (pair - 9) * 9 + half * 3 + 2 : (pair == 12 ? half * 9 + 8 : (half ? -1 : 26)));
}

// x = hi + mid + lo, the three as the HIGH halves of the returned words (lo: the word of the exact remainder, whose low
// half is zero).  Inf / NaN stay in the hi plane only (Inf - Inf would put a NaN next to an Inf).
__device__ __forceinline__ void x3_split(float v, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned u = __float_as_uint(v);
  hi = u & 0xffff0000u;
  const float r1 = (u & 0x7f800000u) == 0x7f800000u ? 0.f : v - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));
}
// (high half of b) : (high half of a)
__device__ __forceinline__ unsigned x3_pack(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- weights: split + fragment order ----
// wq[((((ot * nchunks + ch) * 14 + pair) * 3 + plane) * 64 + lane] = 8 bf16: the channels 8 ch .. 8 ch + 7 of output
// channel 32 ot + (lane & 31) at tap x3_pair_tap(pair, lane >> 5); plane 0 / 1 / 2 = hi / mid / lo.
// transpose (data gradient): the logical filter is W'[m][k][t] = w[k][m][26 - t] (m over Cin_w, k over Cout_w).
__global__ void pack_w3_x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int Cout_w, int Cin_w, int kin,
                                  int mout, int nchunks, int otiles, int transpose) {
  const int64_t total = (int64_t)otiles * nchunks * X3_PAIRS * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int64_t f = i >> 6;
    const int pair = (int)(f % X3_PAIRS), ch = (int)((f / X3_PAIRS) % nchunks);
    const int o = (int)(f / ((int64_t)X3_PAIRS * nchunks)) * 32 + (lane & 31);
    const int tap = x3_pair_tap(pair, lane >> 5);
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      float v = 0.f;
      if (tap >= 0 && o < mout && c < kin)
        v = transpose ? w[((int64_t)c * Cin_w + o) * 27 + (26 - tap)] : w[((int64_t)o * Cin_w + c) * 27 + tap];
      x3_split(v, h[j], m[j], l[j]);
    }
    u32x4* dst = wq + f * 3 * 64 + lane;
    dst[0] = (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
    dst[64] = (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
    dst[128] = (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
  }
}

void launch_pack_w3_x3(const FwdPlan& p, const float* w, void* wp, int Cout_w, int Cin_w, bool transpose, hipStream_t st) {
  const int kin = transpose ? Cout_w : Cin_w, mout = transpose ? Cin_w : Cout_w;
  const int64_t total = (int64_t)p.otiles * p.nchunks * X3_PAIRS * 64;
  hipLaunchKernelGGL(pack_w3_x3_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 2048)), dim3(256), 0, st, w,
                     (u32x4*)wp, Cout_w, Cin_w, kin, mout, p.nchunks, p.otiles, transpose ? 1 : 0);
}

// ---- the kernel: one output tile (4 z x NTW*GY y x GX x voxels, 32 channels) of one split per workgroup ----
// Tile geometry, item order and epilogue are conv3_mfma_fwd_kernel's (conv3d.hip): wave w owns plane z0 + w; the
// accumulator tile of a 32x32 MFMA has the same layout for both operand types, so store_conv_tile (bias, residual,
// fused GroupNorm statistics, split-K slabs) is shared.
template <int NTW, int GX>
__global__ __launch_bounds__(256, 2) void conv3_f32x3_kernel(
    const float* __restrict__ x, const u32x4* __restrict__ wq, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, float* __restrict__ slab, int Cin, int Cout, int D, int H, int W,
    int ty_tiles, int tx_tiles, int nchunks, int ksplit, int64_t xbs, int64_t ybs, int64_t slab_stride,
    float* __restrict__ stat, int otiles, int order) {
  using T = FwdTile<NTW, GX>;
  constexpr int GY = T::GY, TZ = T::TZ, TY = T::TY, TX = T::TX, RS = T::RS, PS = T::PS;
  constexpr int NV = (TZ + 2) * PS;            // voxels of the halo tile
  constexpr int VPER = (NV + 255) / 256;       // voxels staged per thread
  constexpr int NLOAD = VPER * 8;              // dword loads per thread and chunk
  constexpr int LPS = (NLOAD + X3_PAIRS - 1) / X3_PAIRS;   // ... issued per pair step
  static_assert(3 * NV * 16 <= 64 * 1024 && ((2 * PS + (NTW * GY + 2) * RS + 2) * 16 + 2 * NV * 16) < 65536, "LDS offsets");
  __shared__ u32x4 xs[3][NV];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const int ly = l32 / GX, lx = l32 % GX;

  // XCD-aware placement and item order: see conv3_mfma_fwd_kernel
  int bt = blockIdx.x;
  if ((gridDim.x & 7) == 0) bt = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int sp_count = (int)gridDim.x / otiles;
  const int otile = (order & 2) ? bt % otiles : bt / sp_count;
  bt = (order & 2) ? bt / otiles : bt % sp_count;
  const int sp_index = bt;
  const int txt = bt % tx_tiles;
  bt /= tx_tiles;
  const int tyt = bt % ty_tiles;
  const int tzt = bt / ty_tiles;
  const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
  const int o0 = otile * 32;
  const int n = blockIdx.z / ksplit;
  const int ks = blockIdx.z % ksplit;
  const int cps = (nchunks + ksplit - 1) / ksplit;
  const int ch_begin = ks * cps;
  const int ch_end = min(nchunks, ch_begin + cps);

  const float* xn = x + (int64_t)n * xbs;
  const int iHW = H * W;
  const int DHW = iHW * D;
  const unsigned cstride = (unsigned)DHW * 4u;   // < 2^28 (host check): OOB + 7 * cstride does not wrap

  constexpr unsigned OOB = 0x80000000u;
  unsigned goff[VPER];
#pragma unroll
  for (int i = 0; i < VPER; ++i) {
    const int v = tid + 256 * i;
    unsigned off = OOB;
    if (v < NV) {
      const int zz = v / PS, r2 = v - zz * PS;
      const int yy = r2 / RS, xx = r2 - yy * RS;
      const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
      if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        off = (unsigned)(gz * iHW + gy * W + gx) * 4u;
    }
    goff[i] = off;
  }

  f32x16 acc[NTW];
#pragma unroll
  for (int g = 0; g < NTW; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  float xr[VPER][8];
  __amdgpu_buffer_rsrc_t rx;
  auto chunk_setup = [&](int ch, bool live) {   // descriptor over the channels of the chunk that exist: the rest reads 0
    const int c0 = ch * 8;
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (int64_t)c0 * DHW), 0, live ? min(8, Cin - c0) * DHW * 4 : 0,
                                           0x00020000);
  };
  auto fetch = [&](int k) {   // k: compile-time load index
    if (k < NLOAD)
      xr[k >> 3][k & 7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, goff[k >> 3] + (k & 7) * cstride, 0, 0));
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < VPER; ++i) {
      const int v = tid + 256 * i;
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x3_split(xr[i][c], h[c], m[c], l[c]);
      if (v < NV) {
        xs[0][v] = (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
        xs[1][v] = (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
        xs[2][v] = (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
      }
    }
  };

  // the weight fragments of this channel tile are one linear stream of (pair, plane) items over the chunks
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(wq + (int64_t)otile * nchunks * (X3_PAIRS * 3 * 64)), 0, nchunks * (X3_PAIRS * 3 * 1024), 0x00020000);
  const int pg_last = ch_end * X3_PAIRS - 1;
  const unsigned lane16 = (unsigned)lane * 16u;
  u32x4 afr[3][3];   // [slot][plane]; pair p of a chunk sits in slot p % 3
  auto aload = [&](int slot, int pg) {
    const int q = min(pg, pg_last) * 3;   // (uniform; past the end of this split: a harmless re-read)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) afr[slot][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (q + pl) * 1024, 0));
  };

  if (ch_begin < ch_end) {
    chunk_setup(ch_begin, true);
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) fetch(k);
    aload(0, ch_begin * X3_PAIRS);
    aload(1, ch_begin * X3_PAIRS + 1);
    commit();
  }
  __syncthreads();

  const int vb0 = wave * PS + ly * RS + lx;                // lane's voxel in the halo tile for tap (0, 0, 0), row group 0
  const int vbx = vb0 + half, vby = vb0 + half * RS, vbz = vb0 + half * PS;
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const bool more = ch + 1 < ch_end;
    chunk_setup(more ? ch + 1 : ch, more);
    const int pg = ch * X3_PAIRS;
#pragma unroll
    for (int p = 0; p < X3_PAIRS; ++p) {
      const int t0 = x3_pair_tap(p, 0);
      const int off = (t0 / 9) * PS + ((t0 / 3) % 3) * RS + t0 % 3;
      const int vb = (p < 9 ? vbx : (p < 12 ? vby : (p == 12 ? vbz : vb0))) + off;
      aload((p + 2) % 3, pg + p + 2);
      const bf16x8 a_hi = __builtin_bit_cast(bf16x8, afr[p % 3][0]), a_mid = __builtin_bit_cast(bf16x8, afr[p % 3][1]),
                   a_lo = __builtin_bit_cast(bf16x8, afr[p % 3][2]);
      bf16x8 bh[NTW], bm[NTW], bl[NTW];
#pragma unroll
      for (int g = 0; g < NTW; ++g) bh[g] = __builtin_bit_cast(bf16x8, xs[0][vb + g * GY * RS]);
#pragma unroll
      for (int g = 0; g < NTW; ++g) bm[g] = __builtin_bit_cast(bf16x8, xs[1][vb + g * GY * RS]);
#pragma unroll
      for (int g = 0; g < NTW; ++g) bl[g] = __builtin_bit_cast(bf16x8, xs[2][vb + g * GY * RS]);
      // the small terms first
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, bm[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bm[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NTW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bl[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < LPS; ++k) fetch(p * LPS + k);
    }
    // the next chunk's pairs 0 / 1 were loaded into slots 14 % 3 = 2 and 15 % 3 = 0
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4 t = afr[0][pl];
      afr[0][pl] = afr[2][pl];
      afr[1][pl] = t;
    }
    __syncthreads();   // every wave has read its fragments of this chunk
    if (more) {
      commit();
      __syncthreads();
    }
  }

  const int z = z0 + wave;
  const int xg = x0 + lx;
  const bool lane_ok = z < D && xg < W;
  if (ksplit == 1) {
    float* st = stat ? stat + (((int64_t)n * sp_count + sp_index) * 4 + wave) * Cout * 2 : nullptr;
    store_conv_tile<NTW, GY>(acc, y + (int64_t)n * ybs, add ? add + (int64_t)n * ybs : nullptr, bias, o0, Cout, z, y0, xg, ly,
                             half, D, H, W, lane_ok, st);
  } else {
    store_conv_tile<NTW, GY>(acc, slab + (int64_t)ks * slab_stride + (int64_t)n * Cout * D * iHW, nullptr, nullptr, o0, Cout, z,
                             y0, xg, ly, half, D, H, W, lane_ok, nullptr);
  }
}

template <int NTW, int GX>
static void launch_x3(const FwdPlan& p, const float* x, const void* wp, const float* bias, const float* add, float* y,
                      float* slab, int N, int kin, int mout, int D, int H, int W, int64_t xbs, int64_t ybs, hipStream_t st,
                      float* stat) {
  dim3 grid((unsigned)(p.tz_tiles * p.ty_tiles * p.tx_tiles * p.otiles), 1u, (unsigned)(N * p.ksplit));
  hipLaunchKernelGGL((conv3_f32x3_kernel<NTW, GX>), grid, dim3(256), 0, st, x, (const u32x4*)wp, bias, add, y, slab, kin, mout,
                     D, H, W, p.ty_tiles, p.tx_tiles, p.nchunks, p.ksplit, xbs, ybs, (int64_t)N * mout * D * H * W, stat,
                     p.otiles, tuning().conv_cube & 2);
}

// launches the kernel of plan p (p.x3 != 0); the caller (run_mfma_conv) has checked the workspace, packed the weights
// and runs the split-K reduction
int launch_x3_conv(const FwdPlan& p, const float* in, const void* wp, const float* bias, const float* add, float* out,
                   float* slab, int N, int kin, int mout, int D, int H, int W, int64_t in_bs, int64_t out_bs, hipStream_t st,
                   float* stat) {
#define M355_X3_CASE(NTW, GX)                                                                          \
  if (p.ntw == NTW && p.gx == GX) {                                                                    \
    launch_x3<NTW, GX>(p, in, wp, bias, add, out, slab, N, kin, mout, D, H, W, in_bs, out_bs, st, stat); \
    return M355_OK;                                                                                    \
  }
  M355_X3_CASE(4, 32)
  M355_X3_CASE(2, 32)
  M355_X3_CASE(1, 32)
  M355_X3_CASE(4, 16)
  M355_X3_CASE(2, 16)
  M355_X3_CASE(1, 16)
  M355_X3_CASE(4, 8)
  M355_X3_CASE(2, 8)
  M355_X3_CASE(1, 8)
#undef M355_X3_CASE
  set_error("conv3d(f32x3): no kernel for ntw=%d gx=%d", p.ntw, p.gx);
  return M355_EUNSUPPORTED;
}

}  // namespace m355
