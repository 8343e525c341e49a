// Register-only fp32 MFMA loops, random vs zero operands: does the instruction shape change the clock the chip holds?
// (The vendor fp32 GEMM on this box -- MT256x256x32_MI16x16x1 -- reaches 153 TFLOP/s on random data; v_mfma_f32_32x32x2_f32
// loops are held at ~2.1 GHz = 133-137 TFLOP/s.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void loop(const float* src, float* out, int iters, unsigned long long* clk) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[threadIdx.x + 256 * i]; b[i] = src[1024 + threadIdx.x + 256 * i]; }
  f32x16 acc16[NACC]; f32x4 acc4[NACC]; f32x32 acc32[NACC];
  for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; for (int r = 0; r < 32; ++r) acc32[i][r] = 0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3], b[(i >> 2) & 3], acc16[i], 0, 0, 0);
      if constexpr (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc4[i], 0, 0, 0);
      if constexpr (KIND == 2) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[i & 3], b[(i >> 2) & 3], acc16[i], 0, 0, 0);
      if constexpr (KIND == 3) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[i & 3], b[(i >> 2) & 3], acc32[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc16[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; for (int r = 0; r < 32; ++r) s += acc32[i][r]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int KIND, int NACC>
void run(const float* src, const char* data, const char* name, double flop) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<KIND, NACC><<<256, 256>>>(src, out, 500, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop<KIND, NACC><<<256, 256>>>(src, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-6s %-26s %2d acc: %7.3f ms %6.1f TFLOP/s  clock %.2f GHz  %.1f cycles/MFMA\n", data, name, NACC, ms,
         256.0 * 4 * iters * NACC * flop / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / ((double)iters * NACC));
}
int main() {
  float h[2048]; float* src[2];
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 2048; ++i) h[i] = k ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMalloc(&src[k], sizeof(h)); hipMemcpy(src[k], h, sizeof(h), hipMemcpyHostToDevice);
  }
  const char* nm[2] = {"random", "zeros"};
  for (int k = 0; k < 2; ++k) {
    run<0, 8>(src[k], nm[k], "v_mfma_f32_32x32x2_f32", 4096.0);
    run<1, 8>(src[k], nm[k], "v_mfma_f32_16x16x4_f32", 2048.0);
    run<1, 16>(src[k], nm[k], "v_mfma_f32_16x16x4_f32", 2048.0);
    run<2, 8>(src[k], nm[k], "v_mfma_f32_16x16x1_4b_f32", 2048.0);
    run<3, 4>(src[k], nm[k], "v_mfma_f32_32x32x1_2b_f32", 4096.0);
  }
  return 0;
}
