// fp32 convolution on the bf16 matrix pipe: M355_COMPUTE_F32X3 -- forward, data gradient and weight gradient of the
// 3x3x3 / stride 1 / pad 1 nn.Conv3d in Block3d (/root/reference/segmentation_pipeline/models/components.py:48-56) and the
// forward of the k2 s2 nn.ConvTranspose3d (models/modular_unet.py:72-81).  In this file: conv3_f32x3_kernel (+ the 16-row
// remainder tile conv3_f32x3_m16_kernel), conv3_bww_x3_kernel / conv3_bww_x3c_kernel, convt_k2s2_fwd_x3_kernel, the weight
// packs and the planners of the weight gradient.
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate and this chip clocks both down under load
// (DESIGN 4.6), so the fp32 layers sit at the power ceiling of that instruction.  An fp32 number splits EXACTLY into three
// bf16 numbers (8 + 8 + 8 significant bits):
//     x = hi + mid + lo,   hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid            (every step exact in fp32)
// and a product of two bf16 values is exact in the MFMA's fp32 accumulator, so
//     x * w = hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi) + [mid*lo + lo*mid + lo*lo <= 2^-24 |x*w|]
// Six v_mfma_f32_32x32x16_bf16 (K = 16 each) stand for eight fp32 MFMAs (K = 2 each) at a sixteenth of the cost per K:
// 2.7x fewer matrix-core cycles for the same sum, with the dropped terms below half an ulp of each product.  Measured
// against an fp64 convolution the result is as accurate as fp32 arithmetic is (max |err| / max |y| 2e-7 .. 1.5e-6 on the
// cfg2 layers, within 2.5x of the fp32 MFMA kernel's: the error is an fp32 ACCUMULATION's either way;
// profiles/r04_f32x3_accuracy.txt).
//
// Data flow: the activations stay fp32 NCDHW in HBM.  A workgroup stages the halo tile of an 8-channel chunk through
// registers (coalesced dword loads along x through a buffer descriptor that returns 0 for padding), splits every
// value in registers (5.5 vector-ALU ops per element, once per staged element) and writes three bf16 planes to LDS in
// the c8 form [plane][voxel][8 channels] -- one 16-byte item per voxel and plane, which IS the B fragment of the MFMA
// for a lane (voxel, k-half).  K = 16 of one MFMA = 8 channels x 2 TAPS: the lower lane half reads tap t0, the upper
// half tap t1 of a pair (27 taps = 14 pairs, the last one half empty), so a chunk is 8 channels and its three planes
// fit LDS twice per CU (58.8 KB per workgroup at NTW = 4).  The weights are split and laid out in fragment order once
// per optimizer step (pack_w3_x3_kernel: [channel tile][chunk][pair][plane][lane] x 16 B) and stream from L2 straight
// into registers, two (NTW = 4) or six (NTW <= 2) pairs ahead of their use -- their three planes would not fit LDS next
// to the activations.
#include "conv3d_common.hpp"

namespace m355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_PAIRS = 14;
constexpr int X3_QUADS = 7;   // 16-row tile (conv3_f32x3_m16_kernel): K = 32 = 8 channels x 4 taps, tap 4 q + (lane >> 4)
// tap (dz*9 + dy*3 + dx) of lane half `half` of pair `pair`; -1 = no tap (zero weights)
//   pairs 0..8   (dz, dy) = (pair / 3, pair % 3): dx = 0 | 1         -> the halves are one voxel apart along x
//   pairs 9..11  dz = pair - 9, dx = 2: dy = 0 | 1                   -> one tile row apart
//   pair 12      dy = 2, dx = 2: dz = 0 | 1                          -> one tile plane apart
//   pair 13      (2, 2, 2) | none
__host__ __device__ constexpr int x3_pair_tap(int pair, int half) {
  return pair < 9 ? pair * 3 + half : (pair < 12 ? (pair - 9) * 9 + half * 3 + 2 : (pair == 12 ? half * 9 + 8 : (half ? -1 : 26)));
}

// x = hi + mid + lo, the three as the HIGH halves of the returned words (lo: the word of the exact remainder, whose low
// half is zero).  Inf / NaN stay in the hi plane only (Inf - Inf would put a NaN next to an Inf); the outputs such an
// operand touches come out non-finite (Inf, or NaN where it meets a zero plane of the other operand), as they must.
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
__device__ __forceinline__ void pack_w3_x3_item(const float* __restrict__ w, u32x4* __restrict__ wq, int64_t i, int Cout_w,
                                                int Cin_w, int kin, int mout, int nchunks, int transpose) {
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

// the 16-row remainder tile (conv3_f32x3_m16_kernel), behind the 32-row tiles' region:
// wq16[((ch * 7 + quad) * 3 + plane) * 64 + lane] = 8 bf16: the channels 8 ch .. 8 ch + 7 of output channel o16 + (lane & 15)
// at tap 4 quad + (lane >> 4) (tap 27: zeros)
__device__ __forceinline__ void pack_w3_x3_item16(const float* __restrict__ w, u32x4* __restrict__ wq16, int64_t i, int Cout_w,
                                                  int Cin_w, int kin, int mout, int o16, int transpose) {
  const int lane = (int)(i & 63);
  const int64_t f = i >> 6;
  const int quad = (int)(f % X3_QUADS), ch = (int)(f / X3_QUADS);
  const int o = o16 + (lane & 15);
  const int tap = 4 * quad + (lane >> 4);
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = ch * 8 + j;
    float v = 0.f;
    if (tap < 27 && o < mout && c < kin)
      v = transpose ? w[((int64_t)c * Cin_w + o) * 27 + (26 - tap)] : w[((int64_t)o * Cin_w + c) * 27 + tap];
    x3_split(v, h[j], m[j], l[j]);
  }
  u32x4* dst = wq16 + f * 3 * 64 + lane;
  dst[0] = (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
  dst[64] = (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
  dst[128] = (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
}

// items [0, n32) are the 32-row tiles', [n32, n32 + n16) the 16-row tile's
__device__ __forceinline__ void pack_w3_x3_any(const float* __restrict__ w, u32x4* __restrict__ wq, int64_t i, int Cout_w, int Cin_w,
                                               int nchunks, int otiles, int tile16, int transpose) {
  const int kin = transpose ? Cout_w : Cin_w, mout = transpose ? Cin_w : Cout_w;
  const int64_t n32 = (int64_t)otiles * nchunks * X3_PAIRS * 64;
  if (i < n32) pack_w3_x3_item(w, wq, i, Cout_w, Cin_w, kin, mout, nchunks, transpose);
  else if (tile16) pack_w3_x3_item16(w, wq + n32 * 3, i - n32, Cout_w, Cin_w, kin, mout, otiles * 32, transpose);
}
__host__ __device__ inline int64_t x3_pack_items(int nchunks, int otiles, int tile16) {
  return (int64_t)otiles * nchunks * X3_PAIRS * 64 + (tile16 ? (int64_t)nchunks * X3_QUADS * 64 : 0);
}

__global__ void pack_w3_x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int Cout_w, int Cin_w, int nchunks,
                                  int otiles, int tile16, int transpose) {
  const int64_t total = x3_pack_items(nchunks, otiles, tile16);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_w3_x3_any(w, wq, i, Cout_w, Cin_w, nchunks, otiles, tile16, transpose);
}

void launch_pack_w3_x3(const FwdPlan& p, const float* w, void* wp, int Cout_w, int Cin_w, bool transpose, hipStream_t st) {
  const int64_t total = x3_pack_items(p.nchunks, p.otiles, p.tile16);
  hipLaunchKernelGGL(pack_w3_x3_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 2048)), dim3(256), 0, st, w,
                     (u32x4*)wp, Cout_w, Cin_w, p.nchunks, p.otiles, p.tile16, transpose ? 1 : 0);
}

// every split-kernel weight form of a model in ONE launch (m355_conv3d_pack_batch: after optimizer.step both forms of
// every conv weight are re-packed -- 34 launches of ~6 us in a cfg2 step otherwise); blocks [blk0, blk0 + nblk) of the
// grid belong to entry k
__global__ __launch_bounds__(256) void pack_w3_x3_batch_kernel(const X3PackBatch b) {
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= b.e[k + 1].blk0) ++k;
  const X3PackEntry& e = b.e[k];
  const int64_t total = x3_pack_items(e.nchunks, e.otiles, e.tile16);
  for (int64_t i = (int64_t)((int)blockIdx.x - e.blk0) * 256 + threadIdx.x; i < total; i += (int64_t)e.nblk * 256)
    pack_w3_x3_any(e.w, (u32x4*)e.wq, i, e.Cout, e.Cin, e.nchunks, e.otiles, e.tile16, e.transpose);
}

void launch_pack_x3_batch(X3PackBatch& b, int n, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    X3PackEntry& e = b.e[i];
    e.blk0 = blocks;
    e.nblk = (int)std::min<int64_t>(ceil_div(x3_pack_items(e.nchunks, e.otiles, e.tile16), 256), 512);
    blocks += e.nblk;
  }
  b.n = n;
  hipLaunchKernelGGL(pack_w3_x3_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
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
  // Weight fragments in flight: a pair step is 6 * NTW MFMAs, so at NTW = 4 two pairs of distance (~1500 cycles) cover
  // the L2 latency; at NTW <= 2 (the 32^3 .. 8^3 levels: 6 or 12 MFMAs per pair) they did not -- six pairs ahead there
  // (a ring of 7 slots divides the 14 pairs of a chunk: no rotation at the chunk boundary).
  constexpr int AR = NTW >= 4 ? 3 : 7, AD = AR - 1;
  u32x4 afr[AR][3];   // [slot][plane]; pair p of a chunk sits in slot p % AR
  auto aload = [&](int slot, int pg) {
    const int q = min(pg, pg_last) * 3;   // (uniform; past the end of this split: a harmless re-read)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) afr[slot][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (q + pl) * 1024, 0));
  };

  if (ch_begin < ch_end) {
    chunk_setup(ch_begin, true);
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) fetch(k);
#pragma unroll
    for (int i = 0; i < AD; ++i) aload(i, ch_begin * X3_PAIRS + i);
    commit();
  }
  __syncthreads();

  const int vb0 = wave * PS + ly * RS + lx;                // lane's voxel in the halo tile for tap (0, 0, 0), row group 0
  const int vbx = vb0 + half, vby = vb0 + half * RS, vbz = vb0 + half * PS;
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const bool more = ch + 1 < ch_end;
    chunk_setup(more ? ch + 1 : ch, more);
    const int pg = ch * X3_PAIRS;
    // A pair step is cut into NH sub-steps of GH voxel groups: while the 6 * GH MFMAs of a sub-step run, the B fragments of
    // the NEXT sub-step (3 planes x GH groups) are read from LDS into the other half of bq -- every read has a sub-step of
    // MFMAs to land, with the registers of one pair step.  (Left to the compiler the reads sat directly in front of their
    // MFMAs and the weight loads directly in front of their use: 60 % MFMA-busy.)  The scheduling barriers pin the
    // interleave: LDS reads behind the first MFMAs, global loads (weights of pair p + AD, activations of the next chunk)
    // behind the rest.
    {
      constexpr int NH = NTW >= 2 ? 2 : 1, GH = NTW / NH, NSUB = X3_PAIRS * NH;
      bf16x8 bq[2][GH][3];
      auto bload = [&](int buf, int p, int h) __attribute__((always_inline)) {
        const int t0 = x3_pair_tap(p, 0);
        const int off = (t0 / 9) * PS + ((t0 / 3) % 3) * RS + t0 % 3;
        const int vb = (p < 9 ? vbx : (p < 12 ? vby : (p == 12 ? vbz : vb0))) + off;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int gi = 0; gi < GH; ++gi) bq[buf][gi][pl] = __builtin_bit_cast(bf16x8, xs[pl][vb + (h * GH + gi) * GY * RS]);
      };
      bload(0, 0, 0);
#pragma unroll
      for (int sb = 0; sb < NSUB; ++sb) {
        const int p = sb / NH, h = sb % NH;
        const bool nb = sb + 1 < NSUB, first = h == 0, last = h == NH - 1;
        if (nb) bload((sb + 1) & 1, (sb + 1) / NH, (sb + 1) % NH);
        if (first) aload((p + AD) % AR, pg + p + AD);
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, afr[p % AR][0]), a_mid = __builtin_bit_cast(bf16x8, afr[p % AR][1]),
                     a_lo = __builtin_bit_cast(bf16x8, afr[p % AR][2]);
        // (the small terms of a product group first; consecutive MFMAs write different accumulators)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          constexpr int PB[6] = {0, 0, 0, 1, 1, 2};
          const bf16x8 a = pr == 0 ? a_lo : (pr == 1 || pr == 3 ? a_mid : a_hi);   // (lo,hi) (mid,hi) (hi,hi) (mid,mid) (hi,mid) (hi,lo)
#pragma unroll
          for (int gi = 0; gi < GH; ++gi)
            acc[h * GH + gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[sb & 1][gi][PB[pr]], acc[h * GH + gi], 0, 0, 0);
        }
        if (last) {
#pragma unroll
          for (int k = 0; k < LPS; ++k) fetch(p * LPS + k);
        }
        constexpr int NM = 6 * GH, NR = 3 * GH;
        const int nv = (first ? 3 : 0) + (last ? LPS : 0);   // global loads of this sub-step
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
          if (nb && i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // a B fragment of the next sub-step
          if (i >= NR && i - NR < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a global load
        }
#pragma unroll
        for (int i = NM - NR; i < 3 + LPS; ++i)
          if (i < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (AR == 3) {   // the next chunk's pairs 0 / 1 were loaded into slots 14 % 3 = 2 and 15 % 3 = 0
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const u32x4 t = afr[0][pl];
        afr[0][pl] = afr[2][pl];
        afr[1][pl] = t;
      }
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

// ---- the 16-row remainder tile of channel counts that are no multiple of 32 (the reference's real widths are 40 / 80 / 120:
// research/msseg2/msseg2.py:87) on v_mfma_f32_16x16x32_bf16: half the matrix-pipe time of a padded 32-row tile.  Staging,
// chunk loop and weight streaming as conv3_f32x3_kernel; C/D layout and epilogue as the fp32 kernels' 16-row tile
// (store_conv_tile16: col = lane & 15 (voxel), row = 4 * (lane >> 4) + reg).
template <int NTW, int GX>
__global__ __launch_bounds__(256, 2) void conv3_f32x3_m16_kernel(
    const float* __restrict__ x, const u32x4* __restrict__ wq, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ y, float* __restrict__ slab, int Cin, int Cout, int D, int H, int W,
    int ty_tiles, int tx_tiles, int nchunks, int ksplit, int64_t xbs, int64_t ybs, int64_t slab_stride,
    float* __restrict__ stat, int o16) {
  using T = FwdTile<NTW, GX>;
  constexpr int GY = T::GY, TZ = T::TZ, TY = T::TY, TX = T::TX, RS = T::RS, PS = T::PS;
  constexpr int NV = (TZ + 2) * PS;            // voxels of the halo tile
  constexpr int VPER = (NV + 255) / 256;       // voxels staged per thread
  constexpr int NLOAD = VPER * 8;              // dword loads per thread and chunk
  constexpr int LPS = (NLOAD + X3_QUADS - 1) / X3_QUADS;   // ... issued per quad step
  static_assert(3 * NV * 16 <= 64 * 1024 && ((2 * PS + (NTW * GY + 2) * RS + 2) * 16 + 2 * NV * 16) < 65536, "LDS offsets");
  __shared__ u32x4 xs[3][NV];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const int ly = l32 / GX, lx = l32 % GX;

  // XCD-aware placement: see conv3_mfma_fwd_kernel (one channel tile: the grid is the spatial tiles)
  int bt = blockIdx.x;
  if ((gridDim.x & 7) == 0) bt = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int sp_count = (int)gridDim.x;
  const int sp_index = bt;
  const int txt = bt % tx_tiles;
  bt /= tx_tiles;
  const int tyt = bt % ty_tiles;
  const int tzt = bt / ty_tiles;
  const int z0 = tzt * TZ, y0 = tyt * TY, x0 = txt * TX;
  const int o0 = o16;
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

  f32x4v acc[NTW][2];   // [voxel group][16-voxel half]: 4 channels of one voxel per lane
#pragma unroll
  for (int g = 0; g < NTW; ++g)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[g][h2][r] = 0.f;

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

  // the weight fragments of the tile are one linear stream of (quad, plane) items over the chunks, six quads ahead of
  // their use (a ring of 7 slots = the quads of a chunk)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, nchunks * (X3_QUADS * 3 * 1024), 0x00020000);
  const int qg_last = ch_end * X3_QUADS - 1;
  const unsigned lane16 = (unsigned)lane * 16u;
  constexpr int AR = 7, AD = AR - 1;
  u32x4 afr[AR][3];   // [slot][plane]; quad q of a chunk sits in slot q
  auto aload = [&](int slot, int qg) {
    const int q = min(qg, qg_last) * 3;   // (uniform; past the end of this split: a harmless re-read)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) afr[slot][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (q + pl) * 1024, 0));
  };

  if (ch_begin < ch_end) {
    chunk_setup(ch_begin, true);
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) fetch(k);
#pragma unroll
    for (int i = 0; i < AD; ++i) aload(i, ch_begin * X3_QUADS + i);
    commit();
  }
  __syncthreads();

  // K = 32 of a 16x16x32 MFMA = 8 channels x 4 TAPS: 16-lane group kg of a fragment holds tap 4 q + kg of quad q (27 taps = 7
  // quads, the last slot empty); a 32-voxel group is two MFMAs (its 16-voxel halves).  Per-lane: the halo voxel of its
  // column for both halves and the tap offsets of its group.
  const int kg = lane >> 4, l16 = lane & 15;
  int toff[X3_QUADS], vb16[2];
#pragma unroll
  for (int q = 0; q < X3_QUADS; ++q) {
    const int tap = min(4 * q + kg, 26);   // (slot 27: zero weights)
    toff[q] = (tap / 9) * PS + ((tap / 3) % 3) * RS + tap % 3;
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int vox = 16 * h2 + l16;
    vb16[h2] = wave * PS + (vox / GX) * RS + vox % GX;
  }
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const bool more = ch + 1 < ch_end;
    chunk_setup(more ? ch + 1 : ch, more);
    const int qg = ch * X3_QUADS;
    // a step = one (quad, voxel group): 12 MFMAs alternating between the group's two halves, the 6 B fragments of the next
    // step read behind the first six of them, the weights of quad q + 6 and the next chunk's activations behind the rest
    {
      constexpr int NSTEP = X3_QUADS * NTW;
      bf16x8 bq[2][2][3];
      auto bload = [&](int buf, int q, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bq[buf][h2][pl] = __builtin_bit_cast(bf16x8, xs[pl][vb16[h2] + toff[q] + g * GY * RS]);
      };
      bload(0, 0, 0);
#pragma unroll
      for (int sb = 0; sb < NSTEP; ++sb) {
        const int q = sb / NTW, g = sb % NTW;
        const bool nb = sb + 1 < NSTEP, first = g == 0, last = g == NTW - 1;
        if (nb) bload((sb + 1) & 1, (sb + 1) / NTW, (sb + 1) % NTW);
        if (first) aload((q + AD) % AR, qg + q + AD);
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, afr[q][0]), a_mid = __builtin_bit_cast(bf16x8, afr[q][1]),
                     a_lo = __builtin_bit_cast(bf16x8, afr[q][2]);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          constexpr int PB[6] = {0, 0, 0, 1, 1, 2};
          const bf16x8 a = pr == 0 ? a_lo : (pr == 1 || pr == 3 ? a_mid : a_hi);   // (lo,hi) (mid,hi) (hi,hi) (mid,mid) (hi,mid) (hi,lo)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
            acc[g][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[sb & 1][h2][PB[pr]], acc[g][h2], 0, 0, 0);
        }
        if (last) {
#pragma unroll
          for (int k = 0; k < LPS; ++k) fetch(q * LPS + k);
        }
        const int nv = (first ? 3 : 0) + (last ? LPS : 0);   // global loads of this step
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
          if (nb && i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // a B fragment of the next step
          if (i >= 6 && i - 6 < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a global load
        }
#pragma unroll
        for (int i = 6; i < 3 + LPS; ++i)
          if (i < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // every wave has read its fragments of this chunk
    if (more) {
      commit();
      __syncthreads();
    }
  }

  const int z = z0 + wave;
  if (ksplit == 1) {
    float* st = stat ? stat + (((int64_t)n * sp_count + sp_index) * 4 + wave) * Cout * 2 : nullptr;
    store_conv_tile16<NTW, GX>(acc, y + (int64_t)n * ybs, add ? add + (int64_t)n * ybs : nullptr, bias, o0, Cout, z, y0, x0, lane,
                               D, H, W, st);
  } else {
    store_conv_tile16<NTW, GX>(acc, slab + (int64_t)ks * slab_stride + (int64_t)n * Cout * D * iHW, nullptr, nullptr, o0, Cout, z,
                               y0, x0, lane, D, H, W, nullptr);
  }
}

template <int NTW, int GX>
static void launch_x3(const FwdPlan& p, const float* x, const void* wp, const float* bias, const float* add, float* y,
                      float* slab, int N, int kin, int mout, int D, int H, int W, int64_t xbs, int64_t ybs, hipStream_t st,
                      float* stat) {
  const unsigned sp = (unsigned)(p.tz_tiles * p.ty_tiles * p.tx_tiles);
  if (p.otiles > 0)
    hipLaunchKernelGGL((conv3_f32x3_kernel<NTW, GX>), dim3(sp * p.otiles, 1u, (unsigned)(N * p.ksplit)), dim3(256), 0, st, x,
                       (const u32x4*)wp, bias, add, y, slab, kin, mout, D, H, W, p.ty_tiles, p.tx_tiles, p.nchunks, p.ksplit, xbs,
                       ybs, (int64_t)N * mout * D * H * W, stat, p.otiles, tuning().conv_cube & 2);
  if (p.tile16)   // the 1..16 remaining channels: their own launch over the spatial tiles (disjoint channels of y / the slabs)
    hipLaunchKernelGGL((conv3_f32x3_m16_kernel<NTW, GX>), dim3(sp, 1u, (unsigned)(N * p.ksplit)), dim3(256), 0, st, x,
                       (const u32x4*)wp + (int64_t)p.otiles * p.nchunks * (X3_PAIRS * 3 * 64), bias, add, y, slab, kin, mout, D, H, W,
                       p.ty_tiles, p.tx_tiles, p.nchunks, p.ksplit, xbs, ybs, (int64_t)N * mout * D * H * W, stat, p.otiles * 32);
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


// ------------------------------------------------------------------------------------------------ weight gradient
// dW[o, c, tap] = sum_v dy[o, v] * x[c, v + off(tap)] on the same three-way split (nn.Conv3d's weight gradient,
// /root/reference/segmentation_pipeline/models/components.py:48-56 under autograd).  MFMA view as in conv3_bww_c8_kernel
// (conv3d_h16.hip): rows = 32 output channels (A = dy), columns = 32 input channels (B = x), K = 16 x-adjacent voxels;
// each wave owns 7 of the 27 taps.  Both operands are activations, so both are split while they are staged: fp32 NCDHW
// -> registers (coalesced dwords along x, zero padding by the buffer descriptor) -> hi / mid / lo -> three voxel-major
// LDS planes [voxel][32 channels] (64-byte rows).  The fragments come out of ds_read_b64_tr_b16 (k = 8 voxels of one
// channel per lane); the three dy planes of a k-step are read once for the wave's 7 taps, the three x planes once per
// tap, and feed six MFMAs: 0.57 fragment reads per MFMA where the 16-bit kernel needs 1.14.
// Tile = ONE z plane of TY x TX = 64 voxels (4 k-steps).  A workgroup walks its tiles along z; the x halo planes sit in
// a RING of four slots (plane p in slot (p + 1) & 3): tile z reads planes z - 1 .. z + 1 while plane z + 2 is split and
// committed to the fourth slot, so a tile costs ONE barrier and one new halo plane ((TY + 2) x (TX + 2) voxels x 32
// channels) + its dy tile (double-buffered) in global loads.
// The workgroup is EIGHT waves in two roles.  Waves 0..3 multiply: fragment reads and MFMAs, nothing else.  Waves 4..7
// stage: they wait for the loads of tile t + 1 (issued a tile ago), split them, write the LDS planes and issue the loads
// of tile t + 2.  With one wave per SIMD doing both (the first version) the ~390 vector-ALU instructions of a tile's
// split and address arithmetic did NOT hide behind that wave's own MFMAs: the kernel without them ran 34 % faster
// (188 -> 252 TF on the cfg2 layers), with them the matrix pipe was 63 % busy.  A staging wave shares its SIMD with one
// multiplying wave and fills exactly the issue slots the MFMAs leave.
// LDS: 4 x 3 x HP x 64 B + 2 x 3 x 4 KB = 127 KB at TX = 32: one workgroup per CU.
typedef short x3_s16x4 __attribute__((ext_vector_type(4)));
typedef short x3_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 x3_tr_frag(const unsigned char* p) {   // two transposed 4-voxel blocks -> 8 k-values
  typedef __attribute__((address_space(3))) x3_s16x4 lds_s16x4;
  const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 256));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Channel counts that are no multiple of 32 (the reference's real widths are 40 / 80 / 120): as in conv3_mfma_bww2c_kernel a
// remainder of 1..16 channels on the o and / or the c side runs on 16-channel sub-tiles (MT x NT of them), here on
// v_mfma_f32_16x16x32_bf16 -- K = 32 voxels = two of the tile's 16-voxel k-steps at once, the four 16-lane groups of the
// transposed read take its four voxel octets -- and the staging waves stage only the channel blocks the sub-tiles read:
// a (32 o, 16 c) pair costs half a full pair, (16, 16) about a third (it is then the staging that sets its tile time).
// C/D layout 16x16: col (c) = lane & 15, row (o) = 4 * (lane >> 4) + reg.
template <int TX>
struct BwX3 {
  static constexpr int TY = 64 / TX, HR = TX + 2, HP = (TY + 2) * HR;
  static constexpr int PLANE_B = HP * 64 + 64, SLOT_B = 3 * PLANE_B, DBUF_B = 3 * 4096;
};

template <int TX, int MT, int NT>
__device__ __forceinline__ void bww_x3_body(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ sl, int N, int Cin, int Cout, int D,
    int H, int W, int ty_tiles, int tx_tiles, int nsplit, int split, int c0, int o0, int64_t xbs, int64_t ybs,
    unsigned char* __restrict__ xs, unsigned char* __restrict__ ds) {
  constexpr bool F = MT == 2 && NT == 2;                         // full pair: 32x32x16 MFMAs, 4 k-steps of 16 voxels
  constexpr int CBX = 2 * NT, CBD = 2 * MT;                      // 8-channel blocks staged of x / of dy
  constexpr int TY = 64 / TX, HR = TX + 2, HP = (TY + 2) * HR;   // tile rows; halo row / plane in voxels
  constexpr int XI = HP * CBX, XPER = (XI + 255) / 256;          // 8-channel items of a halo plane, per staging thread
  constexpr int PLANE_B = HP * 64 + 64, SLOT_B = 3 * PLANE_B, DBUF_B = 3 * 4096;   // + one pad row: where threads without an item write
  constexpr int KH = TX >= 16 ? 8 : HR;                          // voxels 8..15 of a k-step: 8 columns on, or the next row
  constexpr unsigned OOB = 0x80000000u;
  static_assert(4 * SLOT_B + 2 * DBUF_B <= 160 * 1024 && 2 * PLANE_B + 8 * HR * 64 + 512 < 65536, "LDS size / read offsets");
  static_assert(PLANE_B == BwX3<TX>::PLANE_B, "layout");
  // xs: [slot][split plane][halo voxel][32 channels]; ds: [buffer][split plane][voxel][32 channels]

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool stager = wave8 >= 4;
  const int wave = wave8 & 3;                 // of its role
  const int tid = threadIdx.x & 255;          // thread of its role
  const int iHW = H * W, S = D * iHW;

  // tiles: z fastest inside a (sample, y tile, x tile) column; split s owns a contiguous range
  const int ntiles = N * ty_tiles * tx_tiles * D;
  const int per = ntiles / nsplit, rem = ntiles - per * nsplit;
  const int t_begin = split * per + min(split, rem), t_end = t_begin + per + (split < rem ? 1 : 0);
  const int col0 = t_begin < t_end ? t_begin / D : 0;
  int z = t_begin - col0 * D, buf = 0;        // of the tile being multiplied (both roles keep them)

  if (stager) {
    // ------------------------------------------------------------------------------------------------ staging waves
    const unsigned cstride = (unsigned)S * 4u;   // S < 2^24 (host check): 32 channels stay below 2^31 bytes
    // thread-invariant part: item e of a halo plane = (halo voxel e / 4, 8-channel block e % 4).  The channel block runs
    // fastest over the lanes, so a wave's 16-byte LDS writes of an item row are 1 KB contiguous (voxel-fastest they sat
    // 64 bytes apart, 16 lanes on one bank: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE); a global load
    // instruction then covers 16 voxels of 4 channel blocks: four 64-byte runs.
    int xrel[XPER], xdst[XPER];
    unsigned xcode[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
      const int e = tid + 256 * k;
      const int cbl = e % CBX, hv = e / CBX;
      const int yy = hv / HR, xx = hv - yy * HR;
      xrel[k] = cbl * 8 * S + yy * W + xx;
      xdst[k] = e < XI ? hv * 64 + cbl * 16 : HP * 64 + (tid & 3) * 16;
      xcode[k] = e < XI ? (1u << yy) | ((unsigned)xx << 16) : 0xffffu;   // past the end: never valid
    }
    const int dv = (tid / CBD) & 63, dcb = tid % CBD;   // dy item: (voxel of the tile, 8-channel block); threads past 64 CBD: none
    const bool dhas = tid < 64 * CBD;
    const int dvy = dv / TX, dvx = dv - dvy * TX;
    const int ddst = dv * 64 + dcb * 16;

    // staging state of the column that is fetched from
    __amdgpu_buffer_rsrc_t rx, rd;
    unsigned ymask = 0u;
    int colbase = 0, xlim = 0, dbase = 0;
    bool dok = false;
    auto column_setup = [&](int col) __attribute__((always_inline)) {
      const int txt = col % tx_tiles;
      const int c2 = col / tx_tiles;
      const int tyt = c2 % ty_tiles, n = c2 / ty_tiles;
      const int y0 = tyt * TY, x0 = txt * TX;
      const int nbx = min(16 * NT, Cin - c0), nbd = min(16 * MT, Cout - o0);
      rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * xbs + (int64_t)c0 * S), 0, nbx * S * 4, 0x00020000);
      rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * ybs + (int64_t)o0 * S), 0, nbd * S * 4, 0x00020000);
      ymask = 0u;
      for (int yy = 0; yy < TY + 2; ++yy)
        if (y0 + yy - 1 >= 0 && y0 + yy - 1 < H) ymask |= 1u << yy;
      colbase = (y0 - 1) * W + x0 - 1;
      xlim = x0 - 1;                                   // halo column xx is inside the volume iff 0 <= xlim + xx < W
      const int gy = y0 + dvy, gx = x0 + dvx;
      dok = dhas & (gy < H) & (gx < W);
      dbase = dcb * 8 * S + gy * W + gx;
    };
    auto fetch_plane = [&](float (&r)[XPER][8], int p, bool on) __attribute__((always_inline)) {   // halo plane p (absolute z, may lie outside)
      const bool pv = on && p >= 0 && p < D;
#pragma unroll
      for (int k = 0; k < XPER; ++k) {
        const int xx = (int)(xcode[k] >> 16);
        const bool ok = pv & (((xcode[k] & 0xffffu) & ~ymask) == 0u) & ((unsigned)(xlim + xx) < (unsigned)W);
        const unsigned off = ok ? (unsigned)(colbase + p * iHW + xrel[k]) * 4u : OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          r[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + j * cstride, 0, 0));
      }
    };
    auto fetch_dy = [&](float (&r)[8], int zz, bool on) __attribute__((always_inline)) {
      const unsigned off = (on & dok) ? (unsigned)(dbase + zz * iHW) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        r[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, off + j * cstride, 0, 0));
    };
    auto commit8 = [&](const float (&v)[8], unsigned char* dst, int plane_bytes) __attribute__((always_inline)) {   // split + three 16-byte items
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x3_split(v[c], h[c], m[c], l[c]);
      *(u32x4*)(dst) = (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
      *(u32x4*)(dst + plane_bytes) = (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
      *(u32x4*)(dst + 2 * plane_bytes) = (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
    };
    auto commit_plane = [&](const float (&r)[XPER][8], int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < XPER; ++k) commit8(r[k], xs + slot * SLOT_B + xdst[k], PLANE_B);
    };
    auto commit_dy = [&](const float (&r)[8], int b) __attribute__((always_inline)) {
      if (dhas) commit8(r, ds + b * DBUF_B + ddst, 4096);   // (wave-uniform: 64 CBD is a multiple of 64)
    };
    // start of a column / of this split's range: the three planes of tile zz, loaded here and now (once per D tiles)
    auto cold = [&](int zz) __attribute__((always_inline)) {
      float r0[XPER][8], r1[XPER][8], r2[XPER][8];
      fetch_plane(r0, zz - 1, true);
      fetch_plane(r1, zz, true);
      fetch_plane(r2, zz + 1, true);
      commit_plane(r0, zz & 3);
      commit_plane(r1, (zz + 1) & 3);
      commit_plane(r2, (zz + 2) & 3);
    };

    int pz = 0, pcol = col0;   // of the tile the loads are issued for; the staging state is that tile's column
    float xr[XPER][8], dr[8];  // plane z + 2 and the dy tile of tile t + 1, in flight while tile t is multiplied
    if (t_begin < t_end) {
      column_setup(pcol);
      cold(z);
      float d0[8];
      fetch_dy(d0, z, true);
      commit_dy(d0, 0);
      const bool more1 = t_begin + 1 < t_end;
      pz = z + 1;
      if (pz == D) {
        pz = 0;
        pcol += 1;
        if (more1) column_setup(pcol);
      }
      fetch_plane(xr, pz + 1, more1 && pz != 0);   // (the planes of a new column are loaded by cold())
      fetch_dy(dr, pz, more1);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
      const bool more1 = tile + 1 < t_end, more2 = tile + 2 < t_end;
      // the data of tile + 1 (a tile in flight) into the slot no tap of tile z reads and the other dy buffer; unconditional:
      // after the last tile and before a new column the registers hold zeros and their targets are rewritten before use
      commit_plane(xr, (z + 3) & 3);
      commit_dy(dr, buf ^ 1);
      pz += 1;
      if (pz == D) {   // (uniform; once per D tiles)
        pz = 0;
        pcol += 1;
        if (more2) column_setup(pcol);
      }
      fetch_plane(xr, pz + 1, more2 && pz != 0);
      fetch_dy(dr, pz, more2);
      __syncthreads();   // the multiplying waves are done with tile z; tile z + 1 is in LDS
      const bool newcol = z + 1 == D;
      if (more1 && newcol) {   // (uniform) tile + 1 opens a column: its three planes, here and now.  The staging state is
        cold(0);               // already that column's (tile + 2 lies in it as well: D >= 2, host check)
        __syncthreads();
      }
      z = newcol ? 0 : z + 1;
      buf ^= 1;
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- multiplying waves
  const int half = lane >> 5, l32 = lane & 31;
  // transposed-read lane bases: lane 4q + p of a 16-lane group addresses voxel row q, channels 4p .. 4p+3 of its
  // group's 16-channel half; groups 2, 3 (the upper MFMA half) take the voxels 8 .. 15 of the k-step.  Sub-tile form:
  // the four groups take the four voxel octets of a 32-voxel step, all of the same 16 channels.
  const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int oct = TX == 32 ? 8 * grp : (TX == 16 ? (grp >> 1) * HR + 8 * (grp & 1) : grp * HR);
  const int lbA = F ? (8 * (grp >> 1) + tq) * 64 + (grp & 1) * 32 + tp * 8 : (8 * grp + tq) * 64 + tp * 8;
  const int lbB = F ? (KH * (grp >> 1) + tq) * 64 + (grp & 1) * 32 + tp * 8 : (oct + tq) * 64 + tp * 8;
  int tdz[7], tyx[7];   // this wave's taps: plane offset dz, byte offset of (dy, dx) inside a plane
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = min(wave * 7 + t, 26);
    tdz[t] = tap / 9;
    tyx[t] = (((tap / 3) % 3) * HR + tap % 3) * 64 + lbB;
  }

  typedef float f32x4a __attribute__((ext_vector_type(4)));
  f32x16 acc[F ? 7 : 1];
  f32x4a acc16[F ? 1 : 7][MT][NT];
  if constexpr (F) {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  } else {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc16[t][mt][nt][r] = 0.f;
  }

  __syncthreads();   // the first tile is in LDS
  for (int tile = t_begin; tile < t_end; ++tile) {
    // fragment bases of THIS tile: tap plane dz -> ring slot (z + dz) & 3
    const unsigned char* xt[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) xt[t] = xs + tyx[t] + ((z + tdz[t]) & 3) * SLOT_B;
    const unsigned char* da = ds + buf * DBUF_B + lbA;
    if constexpr (F) {
      // 4 k-steps x 7 taps, software-pipelined by one tap: the three x planes of the next (k-step, tap) -- and, once per
      // k-step, the three dy planes of the next k-step -- are requested between the six MFMAs of this one
      auto afrag = [&](int g, int pl) __attribute__((always_inline)) { return x3_tr_frag(da + pl * 4096 + g * 1024); };
      auto bfrag = [&](int g, int t, int pl) __attribute__((always_inline)) {
        const int goff = (TX == 32 ? (g >> 1) * HR + 16 * (g & 1) : (TX == 16 ? g * HR : 2 * g * HR)) * 64;
        return x3_tr_frag(xt[t] + pl * PLANE_B + goff);
      };
      bf16x8 aq[2][3], bq[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) aq[0][pl] = afrag(0, pl);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bq[0][pl] = bfrag(0, 0, pl);
#pragma unroll
      for (int s = 0; s < 28; ++s) {
        const int g = s / 7, t = s % 7;
        const bool nb = s + 1 < 28, na = t == 0 && g + 1 < 4;
        if (nb) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bq[(s + 1) & 1][pl] = bfrag((s + 1) / 7, (s + 1) % 7, pl);
        }
        if (na) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) aq[(g + 1) & 1][pl] = afrag(g + 1, pl);
        }
        const bf16x8 ah = aq[g & 1][0], am = aq[g & 1][1], al = aq[g & 1][2];
        const bf16x8 bh = bq[s & 1][0], bm = bq[s & 1][1], bl = bq[s & 1][2];
        // (the planes in the order their reads were issued; the small terms of a product group first)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        // the reads of the next step go out behind the first three MFMAs: three MFMAs of slack before their use
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // MFMA
          if (nb && i < 3) {
            if (na) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // fragment reads of the next step
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // sub-tile form: 2 steps of 32 voxels x 7 taps; per (step, tap) MT x NT x 6 MFMAs 16x16x32, product by product over
      // the sub-tiles, software-pipelined by one tap
      auto afrag = [&](int ks, int mt, int pl) __attribute__((always_inline)) {
        return x3_tr_frag(da + pl * 4096 + ks * 2048 + mt * 32);
      };
      auto bfrag = [&](int ks, int t, int nt, int pl) __attribute__((always_inline)) {
        const int koff = (TX == 32 ? ks * HR : (TX == 16 ? 2 * ks * HR : 4 * ks * HR)) * 64;
        return x3_tr_frag(xt[t] + pl * PLANE_B + koff + nt * 32);
      };
      bf16x8 aq[2][MT][3], bq[2][NT][3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) aq[0][mt][pl] = afrag(0, mt, pl);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[0][nt][pl] = bfrag(0, 0, nt, pl);
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        const int g = s / 7, t = s % 7;
        const bool nb = s + 1 < 14, na = t == 0 && g + 1 < 2;
        if (nb) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[(s + 1) & 1][nt][pl] = bfrag((s + 1) / 7, (s + 1) % 7, nt, pl);
        }
        if (na) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) aq[(g + 1) & 1][mt][pl] = afrag(g + 1, mt, pl);
        }
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 0, 1, 2, 1, 0};   // (lo,hi) (mid,hi) (mid,mid) (hi,lo) (hi,mid) (hi,hi)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc16[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[g & 1][mt][PA[pr]], bq[s & 1][nt][PB[pr]],
                                                                          acc16[t][mt][nt], 0, 0, 0);
        }
        constexpr int NM = 6 * MT * NT;   // MFMAs of a step; its reads go out behind the first half of them
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // MFMA
          if (nb && i < NM / 2) {
            if (na) __builtin_amdgcn_sched_group_barrier(0x100, (6 * NT + 6 * MT + NM / 2 - 1) / (NM / 2), 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, (6 * NT + NM / 2 - 1) / (NM / 2), 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // every wave is done reading this tile; the next one is in LDS
    const bool newcol = z + 1 == D;
    if (tile + 1 < t_end && newcol) __syncthreads();   // (the staging waves load the three planes of the new column)
    z = newcol ? 0 : z + 1;
    buf ^= 1;
  }

  // partial dW -> slab[split][27][Cout][Cin] (lane = input channel: 32 / 16 consecutive floats per store)
  if constexpr (F) {
    const int c = c0 + l32;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int tap = wave * 7 + t;
      if (tap < 27 && c < Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (o < Cout) sl[((int64_t)tap * Cout + o) * Cin + c] = acc[t][r];
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int tap = wave * 7 + t;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int c = c0 + 16 * nt + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * mt + 4 * grp + r;
            if (tap < 27 && c < Cin && o < Cout) sl[((int64_t)tap * Cout + o) * Cin + c] = acc16[t][mt][nt][r];
          }
        }
    }
  }
}

template <int TX>
__global__ __launch_bounds__(512, 1) void conv3_bww_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N, int Cin, int Cout, int D,
    int H, int W, int ty_tiles, int tx_tiles, int nsplit, int ctiles, int otiles, int64_t xbs, int64_t ybs) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[4 * BwX3<TX>::SLOT_B];
  __shared__ __attribute__((aligned(16))) unsigned char ds[2 * BwX3<TX>::DBUF_B];
  int vid;  // XCD-aware placement, (c-tile, o-tile) pair fastest: see conv3_mfma_bww2_kernel
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int pairs = ctiles * otiles;
  const int pair = vid % pairs, split = vid / pairs;
  bww_x3_body<TX, 2, 2>(x, dy, slab + (int64_t)split * 27 * Cout * Cin, N, Cin, Cout, D, H, W, ty_tiles, tx_tiles, nsplit, split,
                        (pair % ctiles) * 32, (pair / ctiles) * 32, xbs, ybs, xs, ds);
}

// The same with the pairs of a channel remainder on their 16-channel sub-tiles: ONE launch holds all four pair classes, each
// cut into a number of voxel-range splits proportional to its cost (see conv3_mfma_bww2c_kernel).
template <int TX>
__global__ __launch_bounds__(512, 1) void conv3_bww_x3c_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N, int Cin, int Cout, int D,
    int H, int W, int ty_tiles, int tx_tiles, BwwClasses k, int64_t xbs, int64_t ybs) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[4 * BwX3<TX>::SLOT_B];
  __shared__ __attribute__((aligned(16))) unsigned char ds[2 * BwX3<TX>::DBUF_B];
  int vid;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int cls = vid >= k.start[3] ? 3 : (vid >= k.start[2] ? 2 : (vid >= k.start[1] ? 1 : 0));
  const int idx = vid - k.start[cls];
  const int npairs = cls == 0 ? k.of * k.cf : (cls == 1 ? k.of : (cls == 2 ? k.cf : 1));
  const int pair = idx % npairs, split = idx / npairs;
  float* sl = slab + (int64_t)split * 27 * Cout * Cin;
  const int ns = k.ns[cls];
  if (cls == 0) {
    bww_x3_body<TX, 2, 2>(x, dy, sl, N, Cin, Cout, D, H, W, ty_tiles, tx_tiles, ns, split, (pair % k.cf) * 32, (pair / k.cf) * 32,
                          xbs, ybs, xs, ds);
  } else if (cls == 1) {   // c remainder: 32 o x 16 c
    bww_x3_body<TX, 2, 1>(x, dy, sl, N, Cin, Cout, D, H, W, ty_tiles, tx_tiles, ns, split, k.cf * 32, pair * 32, xbs, ybs, xs, ds);
  } else if (cls == 2) {   // o remainder: 16 o x 32 c
    bww_x3_body<TX, 1, 2>(x, dy, sl, N, Cin, Cout, D, H, W, ty_tiles, tx_tiles, ns, split, pair * 32, k.of * 32, xbs, ybs, xs, ds);
  } else {
    bww_x3_body<TX, 1, 1>(x, dy, sl, N, Cin, Cout, D, H, W, ty_tiles, tx_tiles, ns, split, k.cf * 32, k.of * 32, xbs, ybs, xs, ds);
  }
}

// tile width: the padded volume decides (ties: the wider tile, whose rows coalesce better)
static int bww_x3_tx(int H, int W) {
  int best = 32;
  int64_t best_v = -1;
  for (int tx : {32, 16, 8}) {
    const int ty = 64 / tx;
    const int64_t v = round_up(W, tx) * round_up(H, ty);
    if (best_v < 0 || v < best_v) best = tx, best_v = v;
  }
  return best;
}

BwwX3Plan plan_bww_x3(int N, int Cin, int Cout, int D, int H, int W) {
  BwwX3Plan p{};
  p.tx = bww_x3_tx(H, W);
  p.ty_tiles = (int)ceil_div(H, 64 / p.tx);
  p.tx_tiles = (int)ceil_div(W, p.tx);
  p.ctiles = (int)ceil_div(Cin, 32);
  p.otiles = (int)ceil_div(Cout, 32);
  const int64_t ntiles = (int64_t)N * p.ty_tiles * p.tx_tiles * D;
  const auto rem16 = [](int c) { return c % 32 >= 1 && c % 32 <= 16 ? 1 : 0; };
  p.k.orem = tuning().tile16 ? rem16(Cout) : 0;
  p.k.crem = tuning().tile16 ? rem16(Cin) : 0;
  p.k.of = p.otiles - p.k.orem;
  p.k.cf = p.ctiles - p.k.crem;
  p.classes = p.k.orem || p.k.crem;
  // cost of a tile of each pair class in units of a full 32 x 32 pair (the (16, 16) class is bound by its staging)
  const double cost[4] = {1.0, 0.5, 0.5, 0.35};
  const int64_t npairs[4] = {(int64_t)p.k.of * p.k.cf, (int64_t)p.k.of * p.k.crem, (int64_t)p.k.orem * p.k.cf,
                             (int64_t)p.k.orem * p.k.crem};
  double units = 0;
  for (int c = 0; c < 4; ++c) units += cost[c] * (double)npairs[c];
  const int cus = num_cus();
  // one workgroup per CU: time ~ residencies x (tiles per split x ~3.5 us + ~10 us of cold start and slab write) + the
  // slab traffic (written here, read by the reduction)
  const double slab_us = 2.0 * (double)Cout * Cin * 27 * 4 / 4.0e6;   // per split
  int64_t cand[32];
  int nc = 0;
  for (int64_t ns = 1; ns < ntiles && nc < 20; ns *= 2) cand[nc++] = ns;
  for (int r = 1; r <= 6; ++r) cand[nc++] = std::max<int64_t>(1, (int64_t)((double)cus * r / units));
  cand[nc++] = std::max<int64_t>(1, ntiles);
  std::sort(cand, cand + nc);
  double best = 1e30;
  int64_t nsplit = 1;
  for (int i = 0; i < nc; ++i) {
    const int64_t ns = std::min<int64_t>(cand[i], std::max<int64_t>(1, ntiles));
    if (ns * Cout * Cin * 27 * 4 > (256ll << 20) && ns > 1) continue;
    const double rounds = std::ceil(units * (double)ns / (double)cus - 1e-9);
    const double cost_us = rounds * ((double)ceil_div(ntiles, ns) * 4.0 + 10.0) + (double)ns * slab_us;
    if (cost_us < best * 0.97) {
      best = cost_us;
      nsplit = ns;
    }
  }
  if (const int force = tuning().bww_nsplit) nsplit = std::min<int64_t>(force, std::max<int64_t>(1, ntiles));
  p.nsplit = (int)nsplit;
  int64_t max_ns = nsplit;
  for (int c = 0; c < 4; ++c) p.k.ns[c] = p.nsplit;
  if (p.classes) {
    // the split count of a class is proportional to its cost, so that every workgroup of the launch lasts about equally
    // long; the rounding of the per-class counts must not spill a workgroup into another residency
    double base = (double)nsplit;
    const int64_t budget = (int64_t)std::ceil(units * base / (double)cus - 1e-9) * cus;
    for (;;) {
      int wg = 0;
      max_ns = 1;
      for (int c = 0; c < 4; ++c) {
        const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)(base * cost[c] + 0.5)));
        p.k.ns[c] = npairs[c] ? (int)ns : 1;
        p.k.start[c] = wg;
        wg += (int)(npairs[c] * p.k.ns[c]);
        if (npairs[c]) max_ns = std::max<int64_t>(max_ns, ns);
      }
      p.class_wgs = wg;
      if (wg <= budget || base <= 1.0 || tuning().bww_nsplit) break;
      base *= 0.99;
    }
  }
  p.slab_bytes = (size_t)round_up(max_ns * Cout * Cin * 27 * 4, 256);
  return p;
}

int launch_bww_x3(const BwwX3Plan& p, const float* x, const float* dy, float* slab, int N, int Cin, int Cout, int D, int H,
                  int W, int64_t xbs, int64_t ybs, hipStream_t st) {
  if (p.classes) {
    const dim3 grid((unsigned)p.class_wgs);
#define M355_X3_BWWC(TXV)                                                                                               \
  hipLaunchKernelGGL((conv3_bww_x3c_kernel<TXV>), grid, dim3(512), 0, st, x, dy, slab, N, Cin, Cout, D, H, W, p.ty_tiles, \
                     p.tx_tiles, p.k, xbs, ybs);
    if (p.tx == 32) { M355_X3_BWWC(32) } else if (p.tx == 16) { M355_X3_BWWC(16) } else { M355_X3_BWWC(8) }
#undef M355_X3_BWWC
    return M355_OK;
  }
  const dim3 grid((unsigned)(p.ctiles * p.otiles * p.nsplit));
#define M355_X3_BWW(TXV)                                                                                              \
  hipLaunchKernelGGL((conv3_bww_x3_kernel<TXV>), grid, dim3(512), 0, st, x, dy, slab, N, Cin, Cout, D, H, W, p.ty_tiles, \
                     p.tx_tiles, p.nsplit, p.ctiles, p.otiles, xbs, ybs);
  if (p.tx == 32) { M355_X3_BWW(32) } else if (p.tx == 16) { M355_X3_BWW(16) } else { M355_X3_BWW(8) }
#undef M355_X3_BWW
  return M355_OK;
}

// ------------------------------------------------------------------------------------ ConvTranspose3d k2 s2, forward
// Y[m, v] = bias[o] + sum_ci W[ci, m] * X[ci, v], m = o * 8 + t (convt.hip: every input voxel owns its 2x2x2 output block)
// on the same split: the fp32-MFMA kernel (convt_k2s2_fwd_mfma_kernel) spends ~65 us of matrix-pipe time on the 64 -> 32
// level @64^3 -> 128^3 against ~60 us of stores; here the pipe needs a third of that.  The [Cin][NVT] x tile is staged as
// three bf16 planes of c8 items [plane][channel block][voxel] (loads coalesced along the voxels, 16-byte LDS writes and
// fragment reads contiguous over the lanes); a wave walks 32-row m-tiles (4 output channels x 8 taps), splits its
// weight fragment (8 input channels of a weight column, straight from L2) in registers -- once per (m-tile, 16-channel
// chunk), used for 6 x NVT / 32 MFMAs -- and stores float2 pairs exactly as the fp32 kernel does.
template <int NVT>
__global__ __launch_bounds__(256) void convt_k2s2_fwd_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int Cin,
    int Cout, int D, int H, int W, int64_t xbs, int64_t ybs, int mt_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs3_raw[];   // [plane][CB][NVT] x 16 B
  u32x4* xs3 = reinterpret_cast<u32x4*>(xs3_raw);
  constexpr int NG = NVT / 32, P = 256 / NVT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int S = D * H * W;
  const int v0 = blockIdx.x * NVT;
  const int n = blockIdx.z;
  const float* xn = x + (int64_t)n * xbs;
  float* yn = y + (int64_t)n * ybs;
  const int CB = (Cin + 15) / 16 * 2;          // 8-channel blocks, whole 16-channel chunks
  const int OH = 2 * H, OW = 2 * W;
  const int64_t OS = (int64_t)S * 8;
  {
    const int sv = tid % NVT, part = tid / NVT;
    const bool vin = v0 + sv < S;
    const float* xp = xn + (vin ? v0 + sv : 0);
    for (int cb = part; cb < CB; cb += P) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const bool ok = vin && c < Cin;
        const float t = xp[ok ? (int64_t)c * S : 0];   // unconditional load from a clamped address
        v[j] = ok ? t : 0.f;
      }
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x3_split(v[j], h[j], m[j], l[j]);
      xs3[(0 * CB + cb) * NVT + sv] = (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
      xs3[(1 * CB + cb) * NVT + sv] = (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
      xs3[(2 * CB + cb) * NVT + sv] = (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
    }
  }
  __syncthreads();
  const int mtiles = (Cout + 3) / 4;
  const int mt_begin = blockIdx.y * mt_per_wg, mt_end = min(mtiles, mt_begin + mt_per_wg);
  const int wrow = Cout * 8;  // weight offsets fit 32 bits (host: convt_fits_i32)
  int64_t obase[NG];          // output offsets of this lane's voxels (a = half)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int v = min(v0 + g * 32 + l32, S - 1);
    const int ix = v % W, iy = (v / W) % H, iz = v / (W * H);
    obase[g] = ((int64_t)(2 * iz + half) * OH + 2 * iy) * OW + 2 * ix;
  }
  // weights: the 8 channels (k0 + 8 half .. + 7) of weight column mt * 32 + l32, raw; the next chunk (possibly of the next
  // m-tile) is in flight during the MFMAs of the current one.  Loads from clamped addresses, the zero mask at the split.
  float a_nxt[8];
  auto wload = [&](int mt, int k0) __attribute__((always_inline)) {
    const int col = min(mt * 32 + l32, Cout * 8 - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) a_nxt[j] = w[min(k0 + 8 * half + j, Cin - 1) * wrow + col];
  };
  bf16x8 ah, am, al;
  auto wsplit = [&](int mt, int k0) __attribute__((always_inline)) {
    const bool ook = mt * 4 + (l32 >> 3) < Cout;
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x3_split((ook && k0 + 8 * half + j < Cin) ? a_nxt[j] : 0.f, h[j], m[j], l[j]);
    ah = __builtin_bit_cast(bf16x8, (u32x4){x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])});
    am = __builtin_bit_cast(bf16x8, (u32x4){x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])});
    al = __builtin_bit_cast(bf16x8, (u32x4){x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])});
  };
  const int nchunks = CB / 2;
  int mt = mt_begin + wave;
  if (mt < mt_end) {
    wload(mt, 0);
    wsplit(mt, 0);
  }
  while (mt < mt_end) {
    const int o0 = mt * 4;
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const bool last = ch + 1 >= nchunks;
      const int nmt = last ? mt + 4 : mt, nk0 = last ? 0 : (ch + 1) * 16;
      if (nmt < mt_end) wload(nmt, nk0);
      const u32x4* bp = xs3 + (2 * ch + half) * NVT + l32;
      bf16x8 bh[NG], bm[NG], bl[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) bh[g] = __builtin_bit_cast(bf16x8, bp[g * 32]);
#pragma unroll
      for (int g = 0; g < NG; ++g) bm[g] = __builtin_bit_cast(bf16x8, bp[CB * NVT + g * 32]);
#pragma unroll
      for (int g = 0; g < NG; ++g) bl[g] = __builtin_bit_cast(bf16x8, bp[2 * CB * NVT + g * 32]);
      // (the small terms of a product group first; consecutive MFMAs write different accumulators)
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[g], acc[g], 0, 0, 0);
      wsplit(nmt, nk0);
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

// voxels per workgroup: the three planes of the x tile (6 bytes per element) within 64 KB, so that two workgroups share a CU
int convt_fwd_x3_nvt(int Cin) {
  const int64_t cin2 = round_up(Cin, 16);
  for (int nvt : {128, 64, 32})
    if (cin2 * nvt * 6 <= 65536) return nvt;
  return 0;
}

void launch_convt_fwd_x3(int nvt, dim3 grid, const float* x, const float* w, const float* bias, float* y, int Cin, int Cout,
                         int D, int H, int W, int64_t xbs, int64_t ybs, int mt_per_wg, hipStream_t st) {
  const size_t lds = (size_t)round_up(Cin, 16) * nvt * 6;
#define M355_CONVT_X3(NVT) \
  hipLaunchKernelGGL(convt_k2s2_fwd_x3_kernel<NVT>, grid, dim3(256), lds, st, x, w, bias, y, Cin, Cout, D, H, W, xbs, ybs, mt_per_wg)
  switch (nvt) {
    case 128: M355_CONVT_X3(128); break;
    case 64: M355_CONVT_X3(64); break;
    default: M355_CONVT_X3(32); break;
  }
#undef M355_CONVT_X3
}

}  // namespace m355
