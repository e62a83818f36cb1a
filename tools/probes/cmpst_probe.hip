// Probe: ds_cmpst_rtn_b32 with all 64 lanes on ONE address -- operand order and order of service (gfx950).
// A chain of forward pointers nxt[i] > i is walked by one instruction if lanes are served in ascending order:
// lane i swaps in nxt[i] iff the word holds i, and gets back what it found.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, const uint32_t* nxt_in) {
  __shared__ uint32_t word;
  const uint32_t lane = threadIdx.x;
  const uint32_t nxt = nxt_in[lane];
  if (lane == 0) word = 1;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&word;
  uint32_t ret, fin;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("ds_cmpst_rtn_b32 %0, %2, %3, %4\n\tds_read_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(ret), "=&v"(fin) : "v"(a), "v"(lane), "v"(nxt) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = ret;
  out[64 + lane] = fin;
  if (lane == 0) out[128] = (uint32_t)(t1 - t0);
}
int main() {
  uint32_t h_n[64], *d_n, *d_o, h[129];
  // chain 1 -> 5 -> 9 -> 20 -> 33 -> 34 -> 61 -> 200 ; every other lane points somewhere forward too
  for (int i = 0; i < 64; i++) h_n[i] = i + 2;
  h_n[1] = 5; h_n[5] = 9; h_n[9] = 20; h_n[20] = 33; h_n[33] = 34; h_n[34] = 61; h_n[61] = 200;
  hipMalloc(&d_n, 256); hipMalloc(&d_o, 129 * 4);
  hipMemcpy(d_n, h_n, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_o, d_n);
  hipMemcpy(h, d_o, sizeof h, hipMemcpyDeviceToHost);
  printf("lanes that found their own index (on the chain):");
  for (int i = 0; i < 64; i++) if (h[i] == (uint32_t)i) printf(" %d", i);
  printf("\nreturned:");
  for (int i = 0; i < 64; i++) printf(" %u", h[i]);
  printf("\nfinal word (read behind it): %u   cycles for the pair: %u\n", h[64], h[128]);
  return 0;
}
