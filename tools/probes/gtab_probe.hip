// Price of a hash-table phase through GLOBAL memory with plain 16-bit accesses (the encoder's tables beyond the four that fit
// a CU's LDS): 64 lanes at scattered slots of a private 32 KiB table,
//   A: one dependent load per trip                 (global_load_ushort sc1)
//   B: load old, store mine, load back -- one trip (what a fresh round's table phase would be)
// for 1, 2, 3, 4 waves per CU (tables of 2..8 MiB per XCD in all), beside nothing else.  Ticks are s_memtime (shader clock).
// hipcc --offload-arch=gfx950 -O3 -o gtab_probe gtab_probe.hip && ./gtab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t ld16_sc1(const uint16_t* p) {
  uint32_t v;
  asm volatile("global_load_ushort %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__global__ __launch_bounds__(64) void k(uint16_t* tab, uint32_t* out, unsigned long long* ticks, int iters) {
  const uint32_t lane = threadIdx.x;
  uint16_t* t = tab + (size_t)blockIdx.x * 16384;
  uint32_t x = lane * 2654435761u + blockIdx.x * 40503u, acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t h = ((x * 0x1e35a7bdu) >> 18) & 16383;
    if (MODE == 0) {
      const uint32_t v = ld16_sc1(t + h);
      acc += v;
      x = x * 1664525u + 1013904223u + v;  // dependent
    } else {
      uint32_t old, chk;
      const uint32_t mine = (it * 64 + lane) & 0xffff;
      asm volatile(
          "global_load_ushort %0, %2, off sc1\n\t"
          "global_store_short %2, %3, off\n\t"
          "global_load_ushort %1, %2, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(old), "=&v"(chk)
          : "v"(t + h), "v"(mine)
          : "memory");
      acc += (chk != mine);  // lost to another lane of the instruction (same slot)
      x = x * 1664525u + 1013904223u + old + chk;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  const int maxw = 256 * 8, iters = 2000;
  uint16_t* tab;
  uint32_t* out;
  unsigned long long* ticks;
  hipMalloc(&tab, (size_t)maxw * 16384 * 2);
  hipMalloc(&out, maxw * 64 * 4);
  hipMalloc(&ticks, maxw * 8);
  hipMemset(tab, 0, (size_t)maxw * 16384 * 2);
  static unsigned long long ht[maxw];
  for (int mode = 0; mode < 2; mode++)
    for (int wpc : {1, 2, 3, 4, 6, 8}) {
      const int nb = 256 * wpc;
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) k<0><<<nb, 64>>>(tab, out, ticks, iters);
        else k<1><<<nb, 64>>>(tab, out, ticks, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(ht, ticks, nb * 8, hipMemcpyDeviceToHost);
      unsigned long long tt = 0, mx = 0;
      for (int i = 0; i < nb; i++) { tt += ht[i]; mx = ht[i] > mx ? ht[i] : mx; }
      printf("%s, %d waves per CU (%4.1f MiB of tables per XCD): %.0f ticks per trip (slowest wave %.0f)\n",
             mode ? "B load+store+load" : "A load", wpc, wpc * 32 * 32 / 1024.0, (double)tt / nb / iters, (double)mx / iters);
    }
  return 0;
}
