// Register-only v_mfma_f32_32x32x2_f32 loop: the practical fp32-MFMA ceiling on this device.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, 1000, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("%s blocks=%d nacc=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, NACC, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, "1 wave/SIMD");
  run<8>(256, "1 wave/SIMD");
  run<4>(512, "2 waves/SIMD");
  run<4>(1024, "4 waves/SIMD");
  run<1>(256, "1 wave/SIMD dependent chain");
  return 0;
}
