// Probe: do sub-dword / unaligned LDS writes from different lanes of ONE instruction to
// different bytes of the same dword all land?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

template <typename P> __device__ __forceinline__ void st32u(P* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
template <typename P> __device__ __forceinline__ uint32_t ld32u(const P* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// mode 0: lane i writes byte i (ds_write_b8)            -> bytes 0..63
// mode 1: lane i writes 4 bytes at 5*i+1 (unaligned b32) -> disjoint bytes, shared dwords
// mode 2: lane i writes 3 bytes at 3*i  (b8+b16 mix)
// mode 3: lane i writes 4 bytes at 4*i+off (all same misalignment, adjacent)
__global__ void probe(uint8_t* out, int mode, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t s[1024];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = 0xEE;
  __syncthreads();
  if (mode == 0) {
    s[lane] = (uint8_t)lane;
  } else if (mode == 1) {
    st32u(s + 5 * lane + 1, 0x01010101u * lane + 0x03020100u);
  } else if (mode == 2) {
    s[3 * lane] = (uint8_t)(lane);
    s[3 * lane + 1] = (uint8_t)(lane + 64);
    s[3 * lane + 2] = (uint8_t)(lane + 128);
  } else {
    st32u(s + 4 * lane + off, 0x01010101u * lane + 0x03020100u);
  }
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = s[i];
}

// mode 4: unaligned READS: lane i reads 4 bytes at 5*i+1
__global__ void probe_read(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t s[1024];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  out[lane] = ld32u(s + 5 * lane + 1);
}

int main() {
  uint8_t* d; hipMalloc(&d, 1024);
  std::vector<uint8_t> h(1024), e(1024);
  int bad_total = 0;
  for (int mode = 0; mode < 4; mode++) {
    for (int off = 0; off < (mode == 3 ? 4 : 1); off++) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, off);
      hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
      std::fill(e.begin(), e.end(), 0xEE);
      for (uint32_t l = 0; l < 64; l++) {
        if (mode == 0) e[l] = l;
        else if (mode == 1) { uint32_t v = 0x01010101u * l + 0x03020100u; memcpy(&e[5 * l + 1], &v, 4); }
        else if (mode == 2) { e[3 * l] = l; e[3 * l + 1] = l + 64; e[3 * l + 2] = l + 128; }
        else { uint32_t v = 0x01010101u * l + 0x03020100u; memcpy(&e[4 * l + off], &v, 4); }
      }
      int bad = 0;
      for (int i = 0; i < 1024; i++) if (h[i] != e[i]) { if (bad < 6) printf("  mode %d off %d byte %d got %02x want %02x\n", mode, off, i, h[i], e[i]); bad++; }
      printf("mode %d off %d: %d wrong bytes\n", mode, off, bad);
      bad_total += bad;
    }
  }
  uint32_t* dr; hipMalloc(&dr, 256);
  hipLaunchKernelGGL(probe_read, dim3(1), dim3(64), 0, 0, dr);
  uint32_t hr[64]; hipMemcpy(hr, dr, 256, hipMemcpyDeviceToHost);
  int badr = 0;
  for (uint32_t l = 0; l < 64; l++) {
    uint32_t v = 0; for (int k = 0; k < 4; k++) v |= (uint32_t)(uint8_t)((5 * l + 1 + k) * 7 + 3) << (8 * k);
    if (v != hr[l]) badr++;
  }
  printf("unaligned reads: %d wrong\n", badr);
  return 0;
}
