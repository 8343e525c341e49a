// Shared helpers for libm355seg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/m355seg.h"

namespace m355 {

void set_error(const char* fmt, ...);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// Checks the launch that was just enqueued.  hipGetLastError is cheap and does
// not synchronise.
static inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return M355_ELAUNCH;
  }
  return M355_OK;
}

#define M355_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      m355::set_error(__VA_ARGS__);     \
      return (code);                    \
    }                                   \
  } while (0)

constexpr int WAVE = 64;
constexpr int NORM_CHUNK = 16384;   // elements (c8: voxels of one channel block) per block in the normalisation reduction passes
constexpr int NORM_CHUNK_C8 = 4096; // voxels per block of the c8 backward's first pass (8 channels per thread: smaller chunks fill the chip)
constexpr int DBIAS_CHUNK = 8192;   // voxels per partial sum of the bias gradient (launch_dbias; workspace = Cout * chunks doubles)

// wave-wide sum (64 lanes), result valid in every lane
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); result valid in thread 0
// (and broadcast to all when BCAST).  `scratch` must hold NT/64 elements.
template <typename T, int NT, bool BCAST = false>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect scratch reuse
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  T r = T(0);
  if (BCAST || threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += scratch[i];
  }
  return r;
}

// Tuning overrides (test / sweep hooks).  The M355_* environment variables are read ONCE, when the
// library is first used, and again only on m355_reload_tuning(): the launch paths never call getenv.
// 0 / unset = use the built-in heuristic.
struct Tuning {
  int conv_ntw = 0, conv_ksplit = 0, conv_slots = 0, conv_persistent = 1;
  int no_small = 0, smallcout_valu = 1, bww_nsplit = 0, bww_gen = 2, bww_queue = 1, h16_persistent = 1, tile16 = 1, convt_h16 = 1, h16_w8 = 1, h16_oneshot = 1, h16_xcd = 1;
  int conv_cube = 3, h16_order = 3;   // item order of the conv kernels: bit 0 = (y, z) tiles in 4x4 cubes, bit 1 = channel tile fastest
  int fuse_softmax = 1;
  int f32x3 = 1;         // M355_COMPUTE_F32X3 layers run conv3_f32x3_kernel (0: they run the fp32 MFMA kernels)
  int f32x3_bww = 1;     // ... and conv3_bww_x3_kernel for the weight gradient
  int f32x3_edge = 0;    // 1: the split kernel also for layers with 3..7 K-channels (M355_F32X3_EDGE; see x3_layer)
  int f32x3_convt = 1;   // ... and convt_k2s2_fwd_x3_kernel for the k2 s2 conv-transpose forward
  int convt_wgs = 0;     // c8 conv-transpose kernels: workgroups per CU of the persistent grids (0 = built-in)
  int h16_stagger = 2;   // 16-bit conv kernel: start offset of the odd workgroup of a CU, in units of 1024 cycles
};
const Tuning& tuning();

static inline int64_t dense_or(int64_t stride, int64_t dense) { return stride ? stride : dense; }

// Compute units of the current device (256 on MI355X); the tile planners size their grids in rounds
// over the CUs.  Falls back to 256 when no device is visible (workspace queries on a build host).
int num_cus();

// zero-initialised, self-resetting work-queue state (16 ints) of this (device, stream): abi.cpp
int* queue_state(hipStream_t st);
// fp16 training flow: the caller's overflow word of the current device (m355_overflow_flag_set), or null.  Kernels that
// round a loss-scaled gradient to fp16 OR bit 0 into it when a value had to be clamped to +-65504 (to_h16_sat), the
// epilogues that remove the loss scale from a parameter gradient OR bit 1 when the result is not finite.
int* overflow_flag();

}  // namespace m355
