// Probe: does a CU of gfx950 keep TWO workgroups of T threads resident when each uses 80 KiB of LDS
// and V vector registers per lane?  512 workgroups that each spin ~100 us: the launch takes ~100 us if
// two fit per CU, ~200 us if one.  (The decode kernel wants 9-10 waves per workgroup; rule of thumb
// "waves per SIMD = floor(512 / VGPRs)" says 80 VGPRs allow 6 waves per SIMD = 24 per CU.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int V>
__device__ __forceinline__ void claim();  // claim V vector registers: write the highest one
#define CLAIM(V, R) \
  template <>       \
  __device__ __forceinline__ void claim<V>() { asm volatile("v_mov_b32 " R ", 0" ::: R); }
CLAIM(32, "v31")
CLAIM(64, "v63")
CLAIM(72, "v71")
CLAIM(80, "v79")
CLAIM(96, "v95")
CLAIM(104, "v103")
CLAIM(128, "v127")
template <int V>
__global__ void spin(uint32_t* out, uint64_t ticks) {
  extern __shared__ uint8_t dyn[];
  dyn[threadIdx.x] = (uint8_t)threadIdx.x;
  claim<V>();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = dyn[3];
}
template <int V>
void run(uint32_t* d) {
  hipFuncSetAttribute((const void*)spin<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int threads : {512, 576, 640, 704, 768, 1024}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<V>, dim3(512), dim3(threads), 81920, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<V>, dim3(512), dim3(threads), 81920, 0, d, 200000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("vgprs %3d threads %4d (%2d waves): %.3f ms  -> %s  (%s)\n", V, threads, threads / 64, ms,
           ms < 0.13f ? "two per CU" : "ONE per CU", hipGetErrorString(hipGetLastError()));
  }
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 1 << 20);
  run<32>(d);
  run<64>(d);
  run<72>(d);
  run<80>(d);
  run<96>(d);
  run<104>(d);
  run<128>(d);
  return 0;
}
