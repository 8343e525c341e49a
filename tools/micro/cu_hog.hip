// Test helper (tools/overlap_probe.py): a kernel that HOLDS `wgs` CUs for `cycles` shader cycles the way a
// collective that overlaps the backward pass does -- 256 threads with a 64 KB LDS allocation and a large
// register footprint, so no one-workgroup-per-CU compute kernel can co-reside with it.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256, 1) void hog_kernel(long long cycles, float* sink) {
  __shared__ float big[16000];
  big[threadIdx.x] = threadIdx.x;
  float keep[192];
#pragma unroll
  for (int i = 0; i < 192; ++i) keep[i] = big[(threadIdx.x + i) & 255] + i;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < 192; ++i) keep[i] = keep[i] * 1.0000001f + 1e-9f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 192; ++i) s += keep[i];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" void cu_hog(int wgs, long long cycles, float* sink, void* stream) {
  hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, cycles, sink);
}
