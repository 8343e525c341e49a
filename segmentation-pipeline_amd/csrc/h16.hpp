// 16-bit operand types of the bf16 / fp16 compute modes (M355_COMPUTE_BF16 / M355_COMPUTE_F16) and the
// "c8" activation layout they read:
//
//   x16[n][cb][s][8]    cb = channel block (channels 8*cb .. 8*cb+7, zero-padded past C), s = voxel (D*H*W)
//
// i.e. the 8 channels of a voxel are one aligned 16-byte item.  That item IS the MFMA operand fragment of
// v_mfma_f32_32x32x16_{bf16,f16} for a lane (voxel, k-half), so the convolution kernels move 16 bytes per
// lane from HBM to LDS to the matrix core without touching them, a tap shift is a whole-item offset, and a
// channel slice of a concat buffer that starts at a multiple of 8 channels is a contiguous run of blocks.
#pragma once
#include "common.hpp"

namespace m355 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename HT>
struct H16;
template <>
struct H16<__bf16> {
  typedef __bf16 x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 x4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct H16<_Float16> {
  typedef _Float16 x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 x4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <typename HT>
__device__ __forceinline__ HT to_h16_sat(float v);
template <>
__device__ __forceinline__ __bf16 to_h16_sat<__bf16>(float v) { return (__bf16)v; }
template <>
__device__ __forceinline__ _Float16 to_h16_sat<_Float16>(float v) {
  // a scaled gradient past the fp16 range saturates instead of becoming inf (which would turn every downstream sum
  // into NaN); NaN stays NaN
  return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
}

// the same, remembering in `sat` that a value was clamped (the kernel ORs bit 0 into the overflow word at its end)
template <typename HT>
__device__ __forceinline__ HT to_h16_sat(float v, bool& sat);
template <>
__device__ __forceinline__ __bf16 to_h16_sat<__bf16>(float v, bool&) { return (__bf16)v; }
template <>
__device__ __forceinline__ _Float16 to_h16_sat<_Float16>(float v, bool& sat) {
  sat = sat || fabsf(v) > 65504.f;
  return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
}
__device__ __forceinline__ void report_saturation(bool sat, int* __restrict__ oflag) {
  if (sat && oflag) atomicOr(oflag, 1);
}
__device__ __forceinline__ void report_nonfinite(float v, int* __restrict__ oflag) {
  if (oflag && !(fabsf(v) <= 3.4028235e38f)) atomicOr(oflag, 2);   // inf or NaN
}

static inline int64_t c8_blocks(int64_t C) { return (C + 7) / 8; }

// fp32 NCDHW -> c8 (round to nearest even) and back; implemented in conv3d_h16.hip
int launch_pack_act16(const float* x, void* x16, int N, int C, int64_t S, int64_t xbs, int64_t x16bs, int compute,
                      hipStream_t st);
int launch_unpack_act16(const void* x16, float* x, int N, int C, int64_t S, int64_t x16bs, int64_t xbs, int compute,
                        hipStream_t st);

// second pass of the normalisation backward that also emits dx as c8 (act16.hip; used by m355_norm_act_bwd_h16)
int launch_norm_bwd_apply_c8(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, const float* stat_m, float* dx, void* dx16, int N, int C, int64_t S,
                             int groups, int act, float slope, int64_t xbs, int64_t ybs, int64_t dx16bs, int compute,
                             hipStream_t st);

// c8-only training flow (train16.hip / norm.hip): finalize stage of the normalisation backward shared by the fp32 and c8
// first passes; bias gradient of a conv from its c8 output gradient
int launch_norm_bwd_reduce(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* stat_m, int N,
                           int C, int groups, int64_t S, int training, float grad_unscale, hipStream_t st,
                           const double* count_ptr = nullptr);
size_t dbias_c8_ws_bytes(int N, int C, int64_t S);
int launch_dbias_c8(const void* dy16, int64_t dybs16, float* dbias, int N, int C, int64_t S, int compute, float unscale,
                    void* ws, hipStream_t st);

}  // namespace m355
