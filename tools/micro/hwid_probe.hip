// Which HW_REG_HW_ID fields tell two co-resident workgroups of a CU apart?  (2 workgroups/CU by LDS, 256 threads)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out) {
  __shared__ float big[16000];  // 64 KB -> two workgroups per CU
  big[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  for (volatile int i = 0; i < 20000; ++i) big[(threadIdx.x + i) & 15999] += 1.f;  // stay resident for a while
  if (big[threadIdx.x] < 0) out[0] = 0;
}
int main() {
  const int G = 512;
  unsigned* d;
  hipMalloc(&d, G * 4 * 2 * 4);
  hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d);
  std::vector<unsigned> h(G * 8);
  hipMemcpy(h.data(), d, G * 32, hipMemcpyDeviceToHost);
  for (int b = 0; b < 24; ++b) {
    printf("block %3d:", b);
    for (int w = 0; w < 4; ++w) {
      unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
      printf("  hw=%08x wave=%u simd=%u cu=%u sh=%u se=%u xcc=%u |", hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1,
             (hw >> 13) & 7, x & 15);
    }
    printf("\n");
  }
  // co-residency: group blocks by (xcc, se, sh, cu) and print wave ids of wave 0 of each
  std::map<unsigned, std::vector<std::pair<int, unsigned>>> cu;
  for (int b = 0; b < G; ++b) {
    unsigned hw = h[(b * 4) * 2], x = h[(b * 4) * 2 + 1] & 15;
    cu[(x << 16) | (hw & 0xff00)].push_back({b, hw & 15});
  }
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("cu key %06x:", kv.first);
    for (auto& p : kv.second) printf(" block %d wave_id %u;", p.first, p.second);
    printf("\n");
  }
  printf("distinct CUs seen: %zu\n", cu.size());
  return 0;
}
