// Probe: what single instructions and small idioms cost a wave that is alone on its SIMD (the
// encoder's situation: one wave per block, four blocks per CU).  s_memtime ticks = shader cycles.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
constexpr int N = 4096;
__global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t seed, const uint32_t* tab) {
  __shared__ uint32_t lds[10000];  // ~40 KB: four workgroups per CU
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 10000; i += 64) lds[i] = (i * 7 + seed) % 10000;
  __syncthreads();
  uint64_t res[10];
  uint32_t sink = 0;
  // 0: dependent SALU adds
  {
    uint32_t s = seed;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
    }
    res[0] = __builtin_amdgcn_s_memtime() - t0;
    sink += s;
  }
  // 1: dependent VALU adds
  {
    uint32_t v = seed + lane;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));
    }
    res[1] = __builtin_amdgcn_s_memtime() - t0;
    sink += v;
  }
  // 2: readlane chain: e = readlane(nxt, e)
  {
    uint32_t nxt = (lane * 5 + 3) & 63;
    uint32_t e = seed & 63;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) e = __builtin_amdgcn_readlane(nxt, e);
    }
    res[2] = __builtin_amdgcn_s_memtime() - t0;
    sink += e;
  }
  // 3: tight taken-branch loop, one SALU op per iteration
  {
    uint32_t s = N;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "1:\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n" : "+s"(s) : : "scc");
    res[3] = __builtin_amdgcn_s_memtime() - t0;
    sink += s;
  }
  // 4: ballot -> ctz -> readlane -> compare (VALU -> SALU -> VALU round trip), dependent
  {
    uint32_t v = seed + lane;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const uint64_t b = __ballot((v & 64) == 0);
        const uint32_t j = b ? __builtin_ctzll(b) : 0;
        v += __builtin_amdgcn_readlane(v, j) | 1;
      }
    }
    res[4] = __builtin_amdgcn_s_memtime() - t0;
    sink += v;
  }
  // 5: dependent LDS reads
  {
    uint32_t a = lane;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
      for (int r = 0; r < 8; r++) a = lds[a];
    }
    res[5] = __builtin_amdgcn_s_memtime() - t0;
    sink += a;
  }
  // 6: ds_bpermute chain
  {
    uint32_t a = lane * 4, v = (lane * 13 + 5) & 63;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
      for (int r = 0; r < 8; r++) a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a << 2), (int)v);
    }
    res[6] = __builtin_amdgcn_s_memtime() - t0;
    sink += a;
  }
  // 7: 64-bit scalar shift/or/and chain (4 SALU per step)
  {
    uint64_t m = seed | 1;
    uint32_t sh = seed & 31;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        asm volatile("s_lshl_b64 %0, %0, %1\n s_or_b64 %0, %0, 5\n s_ff1_i32_b64 %1, %0\n s_and_b32 %1, %1, 7"
                     : "+s"(m), "+s"(sh) : : "scc");
      }
    }
    res[7] = __builtin_amdgcn_s_memtime() - t0;
    sink += (uint32_t)m + sh;
  }
  // 8: not-taken branches: s_cmp + s_cbranch (never taken) x N
  {
    uint32_t s = seed | 1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 2\n2:\n" : "+s"(s) : : "scc");
    }
    res[8] = __builtin_amdgcn_s_memtime() - t0;
    sink += s;
  }
  // 9: taken forward branches over one instruction
  {
    uint32_t s = seed | 1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 2\n3:\n s_add_u32 %0, %0, 2\n" : "+s"(s) : : "scc");
    }
    res[9] = __builtin_amdgcn_s_memtime() - t0;
    sink += s;
  }
  if (lane == 0)
    for (int j = 0; j < 10; j++) out[blockIdx.x * 10 + j] = res[j];
  if (sink == 0x12345678) out[0] = sink + tab[0];
}
int main() {
  uint64_t* d;
  uint32_t* t;
  const int wgs = 1024;
  hipMalloc(&d, wgs * 10 * 8);
  hipMalloc(&t, 64);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, d, 12345u, t);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(wgs * 10);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[10] = {"dependent s_add_u32", "dependent v_add_u32", "readlane chain (e = readlane(nxt, e))",
                           "tight loop: s_sub, s_cmp, taken s_cbranch (per iteration)",
                           "ballot -> ctz -> readlane -> add (per step)", "dependent ds_read_b32",
                           "dependent ds_bpermute_b32", "s_lshl_b64, s_or_b64, s_ff1, s_and (per 4)",
                           "s_cmp + s_cbranch not taken + s_add (per 3)", "s_cmp + s_cbranch taken + s_add (per 3)"};
  for (int j = 0; j < 10; j++) {
    double sum = 0;
    for (int b = 0; b < wgs; b++) sum += (double)h[b * 10 + j];
    printf("%-62s %.1f ticks\n", names[j], sum / wgs / N);
  }
  return 0;
}
