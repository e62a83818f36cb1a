// Are the lanes of ONE global_atomic_swap instruction that hit the same address served in ascending lane
// order (as ds_mskor_rtn_b32 serves them in LDS), and what does the trip cost a lone wave?
// hipcc --offload-arch=gfx950 -O3 -o gatomic_order gatomic_order.hip && ./gatomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void k(uint32_t* tab, uint32_t* out, unsigned long long* ticks, int groups, int iters) {
  const uint32_t lane = threadIdx.x;
  uint32_t* t = tab + blockIdx.x * 16384;
  // `groups` lanes share each address: lane L -> slot L / groups (times a stride, spread over cache lines)
  const uint32_t slot = (lane / groups) * 37 % 16384;
  uint32_t bad = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t mine = 0x10000u * (it + 1) + lane;
    const uint32_t old = atomicExch(&t[(slot + 101 * it) % 16384], mine);
    // ascending service: the first lane of a group sees the table's value (0 here: each round uses fresh
    // slots), every other lane sees the lane right before it
    const bool first = lane % groups == 0;
    const uint32_t want = first ? 0u : 0x10000u * (it + 1) + lane - 1;
    bad += old != want;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = bad;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  const int nb = 1024, iters = 64;
  uint32_t *tab, *out;
  unsigned long long* ticks;
  hipMalloc(&tab, (size_t)nb * 16384 * 4);
  hipMalloc(&out, nb * 64 * 4);
  hipMalloc(&ticks, nb * 8);
  for (int groups : {1, 2, 4, 8}) {
    hipMemset(tab, 0, (size_t)nb * 16384 * 4);
    hipDeviceSynchronize();
    k<<<nb, 64>>>(tab, out, ticks, groups, iters);
    hipDeviceSynchronize();
    static uint32_t h[1024 * 64];
    static unsigned long long ht[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost);
    unsigned long long bad = 0, tt = 0;
    for (int i = 0; i < nb * 64; i++) bad += h[i];
    for (int i = 0; i < nb; i++) tt += ht[i];
    printf("lanes per address %d: out-of-order returns %llu of %d; %.0f ticks per atomic (dependent, %d waves in flight)\n", groups,
           bad, nb * 64 * iters, (double)tt / nb / iters, nb);
  }
  return 0;
}
