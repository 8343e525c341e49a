// What can a wave-level MFMA + LDS-fragment loop reach on this box?  (round 4: calibrates the structure of the 16-bit conv
// kernels: accumulator tiles per wave, waves per SIMD, LDS fragment reads per MFMA, random vs zero operands.)
//   MFMA = v_mfma_f32_32x32x16_bf16 (32768 flop, 8 passes); every "step" issues NR ds_read_b128 fragment reads for the
//   NEXT step and NM MFMAs on the current fragments (software-pipelined by one step, like the conv kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NA A-fragments x NB B-fragments per step -> NA*NB MFMAs on NA*NB accumulators (outer-product register tile)
template <int NA, int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const f32x4* __restrict__ src, float* out, int iters,
                                                          unsigned long long* clk) {
  extern __shared__ f32x4 lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += WAVES * 64) lds[i] = src[i];   // 64 KB of fragments
  __syncthreads();
  f32x16 acc[NA * NB];
#pragma unroll
  for (int i = 0; i < NA * NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const f32x4* base = lds + (tid & 63);
  f32x4 fa[2][NA], fb[2][NB];
  auto rd = [&](int slot, int s) {
    const f32x4* p = base + ((s * 8) & 1023);
#pragma unroll
    for (int i = 0; i < NA; ++i) fa[slot][i] = p[i * 64];
#pragma unroll
    for (int j = 0; j < NB; ++j) fb[slot][j] = p[2048 + j * 64];
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  rd(0, 0);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rd((u + 1) & 1, it + u + 1);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[u][i]),
                                                                     __builtin_bit_cast(bf16x8, fb[u][j]), acc[i * NB + j], 0, 0, 0);
      constexpr int NM = NA * NB, NR = NA + NB;
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
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NA, int NB, int WAVES>
void run(int blocks, const f32x4* src, const char* what) {
  float* out; hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const int iters = 4000;
  hipFuncSetAttribute((const void*)loop_kernel<NA, NB, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop_kernel<NA, NB, WAVES><<<blocks, WAVES * 64, 65536>>>(src, out, 200, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop_kernel<NA, NB, WAVES><<<blocks, WAVES * 64, 65536>>>(src, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)blocks * WAVES * iters * NA * NB;
  const double ghz = (double)h[0] / ((double)h[1] * 10.0) ;   // s_memrealtime ticks at 100 MHz
  const double cyc_per_mfma = (double)h[0] / ((double)iters * NA * NB);
  printf("%-6s tile %dx%d (%2d acc) %d waves/WG x %4d WGs  reads/MFMA %.2f : %7.3f ms %7.1f TFLOP/s  clock %.2f GHz  %5.1f cycles/MFMA per wave\n",
         what, NA, NB, NA * NB, WAVES, blocks, (double)(NA + NB) / (NA * NB), ms, mfmas * 32768.0 / ms / 1e9, ghz, cyc_per_mfma);
  hipFree(out); hipFree(clk);
}

int main() {
  f32x4* src[2];
  unsigned short* h = (unsigned short*)malloc(65536);
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 32768; ++i) {
      float v = k ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f);
      unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16);
    }
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
  }
  return 0;
}
