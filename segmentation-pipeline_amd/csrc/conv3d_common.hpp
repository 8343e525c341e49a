// Shared pieces of the 3x3x3 convolution kernels (conv3d.hip: fp32 MFMA path, planning, ABI;
// conv3d_h16.hip: bf16 / fp16 operand kernels).
#pragma once
#include "common.hpp"
#include "h16.hpp"

namespace m355 {


// Opaque copy: stops LICM from hoisting per-element address decode out of a loop (which
// would keep hundreds of loop-invariant registers alive and spill).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Work-queue state of the persistent kernels: counter[0..7] = tickets of the eight item regions, counter[8] =
// workgroups that have left.  The LAST workgroup to leave zeroes the state, so a packed-weight buffer (which
// holds it) can be reused launch after launch without a memset; the weight-pack kernels zero it initially.
__device__ __forceinline__ void queue_leave(int* __restrict__ counter) {
  if (threadIdx.x == 0) {
    if (atomicAdd(counter + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int r = 0; r < 9; ++r) counter[r] = 0;
    }
  }
}

// ------------------------------------------------------------ MFMA fwd kernel
template <int NTW, int GX>
struct FwdTile {
  static constexpr int GY = 32 / GX;
  static constexpr int TZ = 4;  // one z slice per wave
  static constexpr int TY = NTW * GY;
  static constexpr int TX = GX;
  static constexpr int RS = TX + 2;
  static constexpr int PS = (TY + 2) * RS;
  static constexpr int CS = (TZ + 2) * PS;
  static constexpr int CC = 4;  // input channels per LDS chunk (even: MFMA k-pair)
  static constexpr int NROWS = CC * (TZ + 2) * (TY + 2);
};

// Output tile of one wave: C/D layout col = lane&31 (voxel), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// The bias / residual values of all 16 rows are loaded as one batch (one wait) and the row offsets
// k*DHW are uniform, so the 16*NTW stores go out back to back.  (Written the obvious way --
// `if (o < Cout) { v = acc; if (bias) v += bias[o]; if (add) v += add[idx]; y[idx] = v; }` per
// element -- hipcc emits a branch, a load and a vmcnt(0) per element: ~10 us per tile with the
// matrix core idle.)
template <int NTW, int GY>
__device__ __forceinline__ void store_conv_tile(const f32x16 (&acc)[NTW], float* __restrict__ dst,
                                                const float* __restrict__ addp, const float* __restrict__ bias,
                                                int o0, int Cout, int z, int y0, int xg, int ly, int half,
                                                int D, int H, int W, bool lane_ok, float* __restrict__ stat) {
  // lane_ok: this lane's (z, x) column lies inside the volume.  Every lane stays active to the end
  // (the statistics below are reduced with cross-lane shuffles).
  // stat (may be null): (sum, sum of squares) of the values this WAVE stores, per output channel ->
  // stat[o*2 + {0,1}]; the normalisation that follows the conv sums these partials instead of
  // reading y again (m355_conv3d_fwd_stats / m355_norm_stats_from_partials).
  const int64_t HW = (int64_t)H * W, DHW = HW * D;
  const int ob = o0 + 4 * half;  // this lane's first output channel; row r is channel ob + (r&3) + 8*(r>>2)
  float bb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bb[r] = 0.f;
  if (bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = bias[min(ob + (r & 3) + 8 * (r >> 2), Cout - 1)];
  }
  float s1[16], s2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
  for (int g = 0; g < NTW; ++g) {
    const int yg = y0 + g * GY + ly;
    const bool ok = lane_ok && yg < H;
    const int64_t base = ok ? (int64_t)ob * DHW + (int64_t)z * HW + (int64_t)yg * W + xg : 0;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[g][r] + bb[r];
    if (addp) {
      float aa[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);
        aa[r] = addp[(ok && ob + k < Cout) ? base + (int64_t)k * DHW : 0];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += aa[r];
    }
    if (stat) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = ok ? v[r] : 0.f;
        s1[r] += t;
        s2[r] = fmaf(t, t, s2[r]);
      }
    }
    if (ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);
        if (ob + k < Cout) dst[base + (int64_t)k * DHW] = v[r];
      }
    }
  }
  if (stat) {
    // reduce-scatter over the 32 lanes of each half (xor < 32 stays inside the half): 32 values
    // (16 rows x {sum, sumsq}) are summed over 32 lanes with 16+8+4+2+1 shuffles; lane l ends up
    // holding value index l & 31 = q*16 + r.  Fixed order -> bit-reproducible.
    float a[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a[r] = s1[r];
      a[16 + r] = s2[r];
    }
    // (opaque: the lane-derived values of this once-per-item tail are RECOMPUTED here -- hoisted out of the persistent
    // kernels' item loop they sat in registers through every MFMA chunk, and conv3_mfma_fwd_p_kernel<4, 32> at the
    // 256-register limit spilled them to scratch: 5 VGPR spills, 16 B of scratch, round-3 review)
    const int l32 = opaque((int)threadIdx.x) & 31;
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


// ------------------------------------------------------------------ planning
struct FwdPlan {
  bool mfma;
  bool persistent;  // more items than resident workgroups: the queue-driven kernel variants
  int gx, ntw;
  int tz_tiles, ty_tiles, tx_tiles;
  int otiles, kin_pad, mout_pad, nchunks, ksplit;   // otiles: 32-row output tiles
  int tile16;       // + one 16-row remainder tile at channel 32 * otiles (fp32 path, mout % 32 in 1..16)
  int nw;           // waves per workgroup = z slices of a tile: 4, or 8 (16-bit kernels, double-buffered variant)
  int oneshot;      // 16-bit kernels: one item per workgroup instead of the work queue (items of 1-2 chunks)
  int x3;           // M355_COMPUTE_F32X3 and the layer qualifies: conv3_f32x3_kernel (conv3d_f32x3.hip), 8-channel chunks
  size_t wp_bytes, slab_bytes;
};

// ---- 16-row output tile (v_mfma_f32_16x16x4_f32) for the remainder of channel counts that are no multiple of
// 32 (the reference's real widths are 40 / 80 / 120: research/msseg2/msseg2.py:87, main_config.py:123-127;
// a 32-row tile for 8 remaining channels is 75 % padding).  Same flop rate as the 32x32x2 form; the K-step of 4
// is exactly one 4-channel LDS chunk at a tap.  C/D layout: col = lane & 15 (voxel), row = 4 * (lane >> 4) + reg.
// A 32-voxel group is two MFMAs (x-halves h2); acc[g][h2] holds 4 channels of one voxel per lane.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NTW, int GX>
__device__ __forceinline__ void store_conv_tile16(const f32x4v (&acc)[NTW][2], float* __restrict__ dst,
                                                  const float* __restrict__ addp, const float* __restrict__ bias,
                                                  int o0, int Cout, int z, int y0, int x0, int lane, int D, int H, int W,
                                                  float* __restrict__ stat) {
  constexpr int GY = 32 / GX;
  const int64_t HW = (int64_t)H * W, DHW = HW * D;
  const int ob = o0 + 4 * (lane >> 4);
  float bb[4], s1[4], s2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bb[r] = (bias && ob + r < Cout) ? bias[ob + r] : 0.f;
    s1[r] = s2[r] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < NTW; ++g)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int vox = 16 * h2 + (lane & 15);
      const int yg = y0 + g * GY + vox / GX, xg = x0 + vox % GX;
      const bool ok = z < D && yg < H && xg < W;
      const int64_t base = ok ? (int64_t)ob * DHW + (int64_t)z * HW + (int64_t)yg * W + xg : 0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[g][h2][r] + bb[r];
      if (addp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += addp[(ok && ob + r < Cout) ? base + (int64_t)r * DHW : 0];
      }
      if (stat) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = ok ? v[r] : 0.f;
          s1[r] += t;
          s2[r] = fmaf(t, t, s2[r]);
        }
      }
      if (ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + r < Cout) dst[base + (int64_t)r * DHW] = v[r];
      }
    }
  if (stat) {  // sum over the 16 lanes (voxels) of each channel quad, fixed order
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        s1[r] += __shfl_xor(s1[r], off, 64);
        s2[r] += __shfl_xor(s2[r], off, 64);
      }
    }
    if ((lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ob + r < Cout) {
          stat[(int64_t)(ob + r) * 2] = s1[r];
          stat[(int64_t)(ob + r) * 2 + 1] = s2[r];
        }
    }
  }
}

int pick_gx(int W);
inline bool is16(int compute) { return compute == M355_COMPUTE_BF16 || compute == M355_COMPUTE_F16; }
// M355_COMPUTE_F32X3 (conv3d_f32x3.hip): split + fragment-ordered weights, and the kernel launch of a plan with x3 != 0
void launch_pack_w3_x3(const FwdPlan& p, const float* w, void* wp, int Cout_w, int Cin_w, bool transpose, hipStream_t st);
int launch_x3_conv(const FwdPlan& p, const float* in, const void* wp, const float* bias, const float* add, float* out,
                   float* slab, int N, int kin, int mout, int D, int H, int W, int64_t in_bs, int64_t out_bs, hipStream_t st,
                   float* stat);
// pair classes of a weight gradient whose channel counts leave a 1..16 channel remainder (conv3_mfma_bww2c_kernel,
// conv3_bww_x3c_kernel)
struct BwwClasses {
  int of, cf, orem, crem;   // full 32-channel tiles per side, and whether a 16-row remainder tile follows them
  int ns[4];                // voxel-range splits of a pair of class (o remainder ? 2 : 0) + (c remainder ? 1 : 0)
  int start[4];             // first workgroup of each class
};
// ... and its weight gradient (conv3_bww_x3_kernel / conv3_bww_x3c_kernel): slab[split][27][Cout][Cin] partials, reduced by
// the caller with k.ns[class] splits per pair class
struct BwwX3Plan {
  int tx, ty_tiles, tx_tiles, ctiles, otiles, nsplit;
  bool classes;     // a 1..16 channel remainder on either side: the class kernel
  BwwClasses k;     // always filled: without remainders one class with ns[*] = nsplit
  int class_wgs;
  size_t slab_bytes;
};
BwwX3Plan plan_bww_x3(int N, int Cin, int Cout, int D, int H, int W);
int launch_bww_x3(const BwwX3Plan& p, const float* x, const float* dy, float* slab, int N, int Cin, int Cout, int D, int H,
                  int W, int64_t xbs, int64_t ybs, hipStream_t st);
// ConvTranspose3d k2 s2 forward on the split (conv3d_f32x3.hip; caller: convt.hip)
int convt_fwd_x3_nvt(int Cin);
void launch_convt_fwd_x3(int nvt, dim3 grid, const float* x, const float* w, const float* bias, float* y, int Cin, int Cout,
                         int D, int H, int W, int64_t xbs, int64_t ybs, int mt_per_wg, hipStream_t st);
FwdPlan plan_mfma(int N, int kin, int mout, int D, int H, int W, int compute = M355_COMPUTE_F32);

// 16-bit operand convolution (conv3d_h16.hip).  in16: c8 layout (h16.hpp) with `in16_bs` ELEMENTS between
// samples; out: fp32 NCDHW.  Workspace: p.wp_bytes (packed weights + work queue) + p.slab_bytes.
int run_h16_conv(const FwdPlan& p, int compute, const void* in16, int64_t in16_bs, const float* w, bool transpose,
                 int Cout_w, int Cin_w, const float* bias, const float* add, float* out, int N, int kin, int mout,
                 int D, int H, int W, int64_t out_bs, void* ws, size_t ws_bytes, hipStream_t st, float* stat,
                 const void* prepacked = nullptr, bool out16 = false, bool softmax = false);
void launch_pack_w3_h16(const FwdPlan& p, int compute, const float* w, void* wp, int Cout_w, int Cin_w, bool transpose,
                        hipStream_t st);

// One entry of a batched weight pack (m355_conv3d_pack_batch): after optimizer.step every conv weight of a model is
// re-packed (forward + data-gradient form), ~37 launches of 5-15 us each on the critical path of a train step; the batch
// kernel does them in one launch, blockIdx.y = entry.  kind: 0 = fp32 MFMA layout, 1 = bf16, 2 = fp16 (c8 chunks).
struct PackEntry {
  const float* w;
  void* wp;
  int* counter;       // work queue in the tail of the packed region, zeroed like the single pack kernels do
  int Cout, Cin;      // of the weight tensor
  int kdim;           // fp32: kin_pad; 16-bit: nchunks
  int mout_pad;
  int transpose, kind;
  int blk0, nblk;     // this entry's blocks [blk0, blk0 + nblk) of the launch: proportional to its element count
};
constexpr int PACK_BATCH = 64;   // 64 x 56 B of kernel arguments (< 4 KB)
struct PackBatch {
  PackEntry e[PACK_BATCH];
  int n;
};
void launch_pack_batch(PackBatch& b, int n, hipStream_t st);
// ... and of the split-kernel forms (M355_COMPUTE_F32X3, conv3d_f32x3.hip)
struct X3PackEntry {
  const float* w;
  void* wq;
  int Cout, Cin;      // of the weight tensor
  int nchunks, otiles, tile16, transpose;
  int blk0, nblk;
};
struct X3PackBatch {
  X3PackEntry e[PACK_BATCH];
  int n;
};
void launch_pack_x3_batch(X3PackBatch& b, int n, hipStream_t st);

// fp32 MFMA weight layout (see pack_w3_kernel): block `bid` of `nblk` of one tensor
__device__ __forceinline__ void pack_w3_body(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                             int kin_pad, int mout_pad, int transpose, int64_t bid, int64_t nblk) {
  const int64_t total = (int64_t)kin_pad * 27 * mout_pad;
  for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
    const int m = (int)(i % mout_pad);
    const int64_t r = i / mout_pad;
    const int tap = (int)(r % 27);
    const int kc = (int)(r / 27);
    float v = 0.f;
    if (!transpose) {
      if (kc < Cin && m < Cout) v = w[((int64_t)m * Cin + kc) * 27 + tap];
    } else {
      if (kc < Cout && m < Cin) v = w[((int64_t)kc * Cin + m) * 27 + (26 - tap)];
    }
    wp[i] = v;
  }
}
// blocks along the voxel axis of splitk_reduce_c8_kernel == statistics slots it emits per sample
inline int64_t splitk_c8_slots(int64_t S) { return std::max<int64_t>(1, std::min<int64_t>(ceil_div(S, 256), 2048)); }

// weight gradient from c8 operands (tile 2 x 4 x 32 voxels); slab[split][27][Cout][Cin], summed by slab_reduce_t_kernel
int launch_bww_c8(int compute, const void* x16, const void* dy16, float* slab, int N, int Cin, int Cout, int D, int H,
                  int W, int nsplit, int64_t xbs16, int64_t ybs16, hipStream_t st);
// the same for an edge layer (Cin <= 4 or Cout <= 4): narrow channel and tap share the MFMA column
int launch_bww_c8_small(int compute, const void* x16, const void* dy16, float* slab, int N, int Cin, int Cout, int D, int H,
                  int W, int nsplit, int64_t xbs16, int64_t ybs16, hipStream_t st);

// y[n,o,s] = bias[o] + add[n,o,s] + sum_ks slab[ks][n,o,s]   (fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                     const float* __restrict__ add, float* __restrict__ y, int N,
                                     int Cout, int64_t S, int ksplit, int64_t slab_stride,
                                     int64_t ybs);

}  // namespace m355
