// Probe: how fast one SIMD of gfx950 issues dependent VALU work, as a function of the number of
// waves resident on it and of the instruction-level parallelism inside each wave.  Answers
// "is a lone wave per SIMD issue-latency bound, and do 2 waves (or 2 independent chains) fix it".
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
constexpr int N = 2048;
template <int ILP>
__global__ void k(uint64_t* out, uint32_t seed) {
  uint32_t a[ILP];
  for (int j = 0; j < ILP; j++) a[j] = threadIdx.x * 2654435761u + seed + j;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int j = 0; j < ILP; j++) {  // every instruction depends on the previous one of its chain
        a[j] = (a[j] ^ (a[j] >> 7)) + seed;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int j = 0; j < ILP; j++) s += a[j];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (s == 0x12345678) out[0] = s;
}
template <int ILP>
void run(uint64_t* d) {
  for (int waves : {1, 4, 8, 16, 32}) {  // per CU: 0.25, 1, 2, 4, 8 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ILP>, dim3(256), dim3(64 * (waves > 16 ? 16 : waves)), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    const int wgs = waves > 16 ? 256 * (waves / 16) : 256;
    const int wpw = waves > 16 ? 16 : waves;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ILP>, dim3(wgs), dim3(64 * wpw), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(wgs * wpw);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double instrs = (double)N * 8 * ILP * 3;  // xor, shift, add (shift+xor may fuse: see ISA)
    const double per = sum / h.size() / instrs;
    printf("ILP %d waves/CU %2d: %.2f ticks per VALU instr per wave, %.2f per SIMD-instr; kernel %.3f ms "
           "-> %.2f ns per instr per wave\n",
           ILP, waves, per, per / (waves / 4.0 < 1 ? 1 : waves / 4.0), ms, ms * 1e6 / instrs);
  }
}
int main() {
  uint64_t* d;
  hipMalloc(&d, 1 << 20);
  run<1>(d);
  run<2>(d);
  run<4>(d);
  return 0;
}
