// What can a wave-level MFMA + LDS-fragment loop reach on this box?  (round 4: calibrates the structure of the 16-bit conv
// kernels: accumulator tiles per wave, waves per SIMD, LDS fragment reads per MFMA, random vs zero operands.)
//   MFMA = v_mfma_f32_32x32x2_f32 (4096 flop, 16 passes); every "step" issues NR ds_read_b128 fragment reads for the
//   NEXT step and NM MFMAs on the current fragments (software-pipelined by one step, like the conv kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// NA A-fragments x NB B-fragments per step -> NA*NB MFMAs on NA*NB accumulators (outer-product register tile)
template <int NA, int NB, int WAVES, int REP, int KIND = 0>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const float* __restrict__ src, float* out, int iters,
                                                          unsigned long long* clk) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += WAVES * 64) lds[i] = src[i];   // 64 KB of fragments
  __syncthreads();
  using ACC = typename std::conditional<KIND == 0, f32x16, typename std::conditional<KIND == 1, f32x4, f32x32>::type>::type;
  constexpr int NR_ = KIND == 0 ? 16 : (KIND == 1 ? 4 : 32);
  ACC acc[NA * NB];
#pragma unroll
  for (int i = 0; i < NA * NB; ++i)
#pragma unroll
    for (int r = 0; r < NR_; ++r) acc[i][r] = 0.f;
  const float* base = lds + (tid & 63);
  float fa[2][NA], fb[2][NB];
  auto rd = [&](int slot, int s) {
    const float* p = base + ((s * 8) & 4095);
#pragma unroll
    for (int i = 0; i < NA; ++i) fa[slot][i] = p[i * 64];
#pragma unroll
    for (int j = 0; j < NB; ++j) fb[slot][j] = p[8192 + j * 64];
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  rd(0, 0);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (REP > 0) rd((u + 1) & 1, it + u + 1);
#pragma unroll
      for (int rep = 0; rep < (REP > 0 ? REP : 1); ++rep)
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
        {
          if constexpr (KIND == 0) acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[REP > 0 ? u : 0][i], fb[REP > 0 ? u : 0][j], acc[i * NB + j], 0, 0, 0);
          if constexpr (KIND == 1) acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[REP > 0 ? u : 0][i], fb[REP > 0 ? u : 0][j], acc[i * NB + j], 0, 0, 0);
          if constexpr (KIND == 2) acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x1f32(fa[REP > 0 ? u : 0][i], fb[REP > 0 ? u : 0][j], acc[i * NB + j], 0, 0, 0);
        }
      constexpr int NM = NA * NB * (REP > 0 ? REP : 1), NR = REP > 0 ? NA + NB : 0;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA * NB; ++i)
#pragma unroll
    for (int r = 0; r < NR_; ++r) s += acc[i][r];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NA, int NB, int WAVES, int REP = 1, int KIND = 0>
void run(int blocks, const float* src, const char* what) {
  float* out; hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const int iters = 2000 / (REP > 0 ? REP : 1);
  hipFuncSetAttribute((const void*)loop_kernel<NA, NB, WAVES, REP, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop_kernel<NA, NB, WAVES, REP, KIND><<<blocks, WAVES * 64, 65536>>>(src, out, 200, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop_kernel<NA, NB, WAVES, REP, KIND><<<blocks, WAVES * 64, 65536>>>(src, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)blocks * WAVES * iters * NA * NB * (REP > 0 ? REP : 1);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0) ;   // s_memrealtime ticks at 100 MHz
  const double cyc_per_mfma = (double)h[0] / ((double)iters * NA * NB * (REP > 0 ? REP : 1));
  printf("k%d %-6s tile %dx%d (%2d acc) %d waves/WG x %4d WGs  reads/MFMA %.2f : %7.3f ms %7.1f TFLOP/s  clock %.2f GHz  %5.1f cycles/MFMA per wave\n",
         KIND, what, NA, NB, NA * NB, WAVES, blocks, REP > 0 ? (double)(NA + NB) / (NA * NB * REP) : 0.0, ms, mfmas * (KIND == 1 ? 2048.0 : 4096.0) / ms / 1e9, ghz, cyc_per_mfma);
  hipFree(out); hipFree(clk);
}

int main() {
  float* src[2];
  float* h = (float*)malloc(65536);
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 16384; ++i) h[i] = k ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMalloc(&src[k], 65536); hipMemcpy(src[k], h, 65536, hipMemcpyHostToDevice);
  }
  const char* names[2] = {"random", "zeros"};
  for (int k = 0; k < 2; ++k) {
    run<1, 4, 4>(512, src[k], names[k]);    // the default conv kernel's shape: 4 acc, two workgroups per CU
    run<2, 2, 4>(512, src[k], names[k]);
    run<2, 4, 4>(256, src[k], names[k]);    // 8 acc, one wave per SIMD
    run<2, 4, 4>(512, src[k], names[k]);    // 8 acc, two waves per SIMD
    run<4, 4, 4>(256, src[k], names[k]);    // 16 acc (vendor GEMM shape), one wave per SIMD
    run<2, 4, 8>(256, src[k], names[k]);    // 8 acc, 8 waves in one workgroup
    run<2, 4, 4, 4>(256, src[k], names[k]);   // fragments reused 4x: 0.19 reads/MFMA
    run<2, 4, 4, 10>(256, src[k], names[k]);  // 0.075
    run<2, 4, 4, 0>(256, src[k], names[k]);   // no LDS reads (register-only)
    run<1, 4, 4, 1, 2>(512, src[k], names[k]);   // 32x32x1 (2 blocks): 4 MFMAs = 8 tiles, 1.25 reads/MFMA
    run<2, 4, 4, 1, 2>(256, src[k], names[k]);   // 8 MFMAs = 16 tiles, 256 acc registers
    run<2, 2, 4, 1, 2>(256, src[k], names[k]);
    run<4, 4, 4, 1, 1>(256, src[k], names[k]);   // 16x16x4: 16 tiles of 4 registers
    run<4, 8, 4, 1, 1>(256, src[k], names[k]);
  }
  return 0;
}
