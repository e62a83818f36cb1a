// Throughput of the encoder's candidate gather on one CU's vector memory pipeline: every lane loads N bytes at an arbitrary
// byte offset of a 64 KiB block (its wave's own), W waves per CU issuing such loads back to back (dependent: the next
// address comes from the loaded data, as a round's does), with `active` of the 64 lanes taking part.
// hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int BYTES>
__global__ __launch_bounds__(64) void k(const uint8_t* blocks, uint32_t* out, unsigned long long* ticks, int iters, int active, int near) {
  const uint32_t lane = threadIdx.x;
  const uint8_t* b = blocks + (size_t)blockIdx.x * 65536;
  uint32_t x = lane * 2654435761u + blockIdx.x * 40503u, acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    // near: candidates within 2 KiB behind a moving position (text); else anywhere in the block
    const uint32_t pos = (it * 61u) & 0xffffu;
    uint32_t off = near ? ((pos + 65536 - (x % 2048u)) & 0xffffu) : (x & 0xffffu);
    if (off > 65536 - 16) off = 65536 - 16;
    uint32_t v = 0;
    if ((int)lane < active) {
      if (BYTES == 16) {
        uint4 q;
        __builtin_memcpy(&q, b + off, 16);
        v = q.x ^ q.y ^ q.z ^ q.w;
      } else if (BYTES == 8) {
        uint2 q;
        __builtin_memcpy(&q, b + off, 8);
        v = q.x ^ q.y;
      } else {
        __builtin_memcpy(&v, b + off, 4);
      }
    }
    acc += v;
    x = x * 1664525u + 1013904223u + (v & 1);  // dependent
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  const int maxw = 256 * 8, iters = 2000;
  uint8_t* blocks;
  uint32_t* out;
  unsigned long long* ticks;
  hipMalloc(&blocks, (size_t)maxw * 65536);
  hipMalloc(&out, maxw * 64 * 4);
  hipMalloc(&ticks, maxw * 8);
  hipMemset(blocks, 1, (size_t)maxw * 65536);
  static unsigned long long ht[maxw];
  for (int near = 0; near < 2; near++)
    for (int bytes : {16, 8, 4})
      for (int active : {64, 32, 16})
        for (int wpc : {1, 4, 8}) {
          const int nb = 256 * wpc;
          for (int rep = 0; rep < 2; rep++) {
            if (bytes == 16) k<16><<<nb, 64>>>(blocks, out, ticks, iters, active, near);
            else if (bytes == 8) k<8><<<nb, 64>>>(blocks, out, ticks, iters, active, near);
            else k<4><<<nb, 64>>>(blocks, out, ticks, iters, active, near);
            hipDeviceSynchronize();
          }
          hipMemcpy(ht, ticks, nb * 8, hipMemcpyDeviceToHost);
          unsigned long long tt = 0;
          for (int i = 0; i < nb; i++) tt += ht[i];
          const double per = (double)tt / nb / iters;
          printf("%s %2d bytes, %2d lanes, %d waves per CU: %5.0f ticks per trip -> %4.0f ticks of the CU's pipeline per load\n",
                 near ? "near" : "anywhere", bytes, active, wpc, per, per / wpc);
        }
  return 0;
}
