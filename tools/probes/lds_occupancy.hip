// Probe: how many workgroups of a given LDS footprint gfx950 keeps resident per CU.
// 512 workgroups (2 per CU) that each spin ~100 us: the launch takes ~100 us if two fit, ~200 if one.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void spin(uint32_t* out, uint64_t ticks) {
  extern __shared__ uint8_t dyn[];
  dyn[threadIdx.x] = (uint8_t)threadIdx.x;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = dyn[3];
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 1 << 20);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int threads : {256, 320}) {
    for (uint32_t lds : {32768u, 65536u, 73728u, 77824u, 79872u, 80896u, 81408u, 81664u, 81832u, 81888u, 81920u, 98304u}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(spin, dim3(512), dim3(threads), lds, 0, d, 1000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(512), dim3(threads), lds, 0, d, 200000);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("threads %d lds %6u: %.3f ms  (%s)\n", threads, lds, ms, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
