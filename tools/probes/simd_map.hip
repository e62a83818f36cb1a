// Which SIMD do the waves of a 512-thread workgroup land on, three workgroups per CU (48 KiB of LDS each)?
// hipcc --offload-arch=gfx950 -O3 -o simd_map simd_map.hip && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512) void k(uint32_t* out, int spin) {
  __shared__ uint32_t pad[12000];
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  pad[threadIdx.x] = id;
  __syncthreads();
  // stay resident so that the CU fills up with three workgroups
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = pad[threadIdx.x];
}
int main() {
  const int nwg = 256 * 3;
  uint32_t* d;
  hipMalloc(&d, nwg * 8 * 4);
  k<<<nwg, 512>>>(d, 2000000);
  hipDeviceSynchronize();
  static uint32_t h[nwg * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // group by (se, sh, cu): print the first few CUs
  int shown = 0;
  for (int w = 0; w < nwg && shown < 12; w++) {
    uint32_t id0 = h[w * 8];
    uint32_t cu = (id0 >> 8) & 15, sh = (id0 >> 12) & 1, se = (id0 >> 13) & 7;
    if (!(cu == 0 && sh == 0)) continue;
    printf("wg %4d se %u sh %u cu %u: simd of waves 0..7 =", w, se, sh, cu);
    for (int i = 0; i < 8; i++) printf(" %u", (h[w * 8 + i] >> 4) & 3);
    printf("   wave slots =");
    for (int i = 0; i < 8; i++) printf(" %u", h[w * 8 + i] & 15);
    printf("\n");
    shown++;
  }
  return 0;
}
