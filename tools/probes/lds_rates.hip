// Probe: LDS pipeline cost (cycles per wave-instruction at saturation) for the access shapes
// the decoder uses (gfx950).  16 waves per CU, 8 independent ops per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <typename P> __device__ __forceinline__ void st32u(P* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
template <typename P> __device__ __forceinline__ uint32_t ld32u(const P* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
template <typename P> __device__ __forceinline__ void st16u(P* p, uint16_t v) { __builtin_memcpy(p, &v, 2); }
constexpr int N = 512;
template <int MODE>
__global__ void k(uint64_t* out, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t s[65536 + 64];
  const uint32_t lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t*>(s)[i] = i;
  __syncthreads();
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t r = threadIdx.x * 2654435761u + seed;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      r = r * 1664525u + 1013904223u;
      const uint32_t rnd = (r >> 8) & 0xfffc;
      const uint32_t base = ((i * 8 + j) & 127) * 512;
      if (MODE == 0) acc[j] += *reinterpret_cast<uint32_t*>(s + base + lane * 4);            // aligned b32 read
      if (MODE == 1) acc[j] += ld32u(s + base + lane * 4 + 1);                                // unaligned +1
      if (MODE == 2) acc[j] += *reinterpret_cast<uint32_t*>(s + rnd);                         // random aligned
      if (MODE == 3) acc[j] += ld32u(s + (rnd | 1));                                          // random unaligned
      if (MODE == 4) *reinterpret_cast<uint32_t*>(s + base + lane * 4) = r;                   // aligned write
      if (MODE == 5) st32u(s + base + lane * 4 + 1, r);                                       // unaligned write
      if (MODE == 6) st32u(s + (rnd | 1), r);                                                 // random unaligned write
      if (MODE == 7) s[base + lane] = (uint8_t)r;                                             // b8 contiguous
      if (MODE == 8) s[rnd + (r & 3)] = (uint8_t)r;                                           // b8 random
      if (MODE == 9) st16u(s + (rnd | (r & 2)), (uint16_t)r);                                 // b16 random aligned
      if (MODE == 10) atomicAnd(reinterpret_cast<uint32_t*>(s + rnd), r);                     // atomic random
      if (MODE == 11) acc[j] += (uint32_t)*reinterpret_cast<uint64_t*>(s + (rnd & 0xfff8));   // b64 random aligned
      if (MODE == 12) acc[j] += s[rnd + (r & 3)];                                             // u8 read random
      if (MODE == 13) *reinterpret_cast<uint32_t*>(s + rnd) = r;                              // random aligned write
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t a = 0;
  for (int j = 0; j < 8; j++) a += acc[j];
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a == 0x12345678) out[0] = a;
}
template <int MODE> void run(uint64_t* d, const char* name) {
  for (int waves : {1, 16}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(256 * waves);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    double per = sum / h.size() / (N * 8);
    printf("%-30s waves/CU %2d: %.1f ticks per instr per wave -> LDS pipe %.1f ticks per instr\n", name, waves, per, per / waves);
  }
}
int main() {
  uint64_t* d; hipMalloc(&d, 8 * 8192);
  run<0>(d, "aligned b32 read");  run<1>(d, "unaligned(+1) b32 read"); run<2>(d, "random aligned b32 read");
  run<3>(d, "random unaligned b32 read"); run<4>(d, "aligned b32 write"); run<5>(d, "unaligned(+1) b32 write");
  run<6>(d, "random unaligned b32 write"); run<7>(d, "b8 write contiguous"); run<8>(d, "b8 write random");
  run<9>(d, "b16 write random aligned"); run<10>(d, "atomic and random"); run<11>(d, "b64 read random aligned");
  run<12>(d, "u8 read random"); run<13>(d, "random aligned b32 write");
  return 0;
}
