// Probe: what ds_mskor_rtn_b32 returns and in which order lanes on one address are served (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mskor_probe tools/probes/mskor_probe.hip && tools/probes/mskor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
  __shared__ uint32_t m[64];
  const uint32_t lane = threadIdx.x;
  m[lane] = 0x11110000u + lane;  // dword i holds 0x1111 in the high half, i in the low half
  __syncthreads();
  // (1) distinct addresses: replace the low half with 0xAA00 + lane
  {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&m[lane];
    uint32_t ret;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(a), "v"(0xffffu), "v"(0xAA00u + lane) : "memory");
    out[lane] = ret;
    __syncthreads();
    out[64 + lane] = m[lane];
  }
  __syncthreads();
  // (2) all lanes of a group of 4 on ONE address (dword 0..15), low half := 0xBB00 + lane
  m[lane] = 0x22220000u + lane;
  __syncthreads();
  {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&m[lane / 4];
    uint32_t ret;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(a), "v"(0xffffu), "v"(0xBB00u + lane) : "memory");
    out[128 + lane] = ret;
    __syncthreads();
    out[192 + lane] = m[lane];
  }
  // (3) lanes 2k and 2k+1 on the two halves of one dword
  __syncthreads();
  m[lane] = 0x33334444u;
  __syncthreads();
  {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&m[lane / 2];
    const uint32_t sh = (lane & 1) * 16;
    uint32_t ret;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(a), "v"(0xffffu << sh), "v"((0xC000u + lane) << sh) : "memory");
    out[256 + lane] = ret;
    __syncthreads();
    out[320 + lane] = m[lane];
  }
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[384];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[6] = {"(1) returned", "(1) memory after", "(2) returned (4 lanes per address)", "(2) memory after", "(3) returned (two halves)", "(3) memory after"};
  for (int s = 0; s < 6; s++) {
    printf("%s:\n", names[s]);
    for (int i = 0; i < 16; i++) printf(" %08x", h[64 * s + i]);
    printf("\n");
  }
  return 0;
}
