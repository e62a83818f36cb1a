// Probe: cycles of one ds_cmpst_rtn_b32 with k lanes on ONE address (k = 64, 32, 16, 8, 4), of a ds_mskor_rtn_b32 on 64
// distinct addresses and of a plain ds_read_b32, each waited for; a wave alone on its CU (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
  __shared__ uint32_t word[80];
  const uint32_t lane = threadIdx.x;
  word[lane] = lane;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&word[64];
  const uint32_t am = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&word[lane];
  int slot = 0;
  for (uint32_t act : {64u, 32u, 16u, 8u, 4u}) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 8; rep++) {
      word[64] = 1;
      __syncthreads();
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      uint32_t ret = 0;
      if (lane < act) asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(a), "v"(lane), "v"(lane + 1) : "memory");
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (t1 - t0 < best) best = t1 - t0;
      out[32 + lane] = ret;
    }
    if (lane == 0) out[slot] = (uint32_t)best;
    slot++;
  }
  {
    unsigned long long best = ~0ull, best2 = ~0ull;
    for (int rep = 0; rep < 8; rep++) {
      uint32_t ret;
      unsigned long long t0 = __builtin_amdgcn_s_memtime();
      asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(am), "v"(0xffffu), "v"(lane) : "memory");
      unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (t1 - t0 < best) best = t1 - t0;
      t0 = __builtin_amdgcn_s_memtime();
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(am) : "memory");
      t1 = __builtin_amdgcn_s_memtime();
      if (t1 - t0 < best2) best2 = t1 - t0;
      out[32 + lane] = ret;
    }
    if (lane == 0) { out[slot] = (uint32_t)best; out[slot + 1] = (uint32_t)best2; }
  }
}
int main() {
  uint32_t* d; uint32_t h[8];
  hipMalloc(&d, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("cmpst, lanes on one address: 64: %u  32: %u  16: %u  8: %u  4: %u cycles;  mskor 64 distinct: %u;  ds_read_b32: %u\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
  return 0;
}
