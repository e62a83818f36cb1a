// crc_pack_kernels.h -- masked CRC32C and the length-then-data packing pass for gfx950.
#pragma once

#include "common.h"

namespace snappy_hip {

// ============================================================================================
// Masked CRC32C (maskedCrc, snappy/codec.nim:71-75 -> masked_crc32c, snappy/crc32c.c:759-763;
// table algorithm crc32c.c:204-214 / :676-731).
//
// One 256-thread workgroup per unit.  The message is right-aligned in rows of 256 dwords
// (1 KiB); thread t owns column t.  A CRC register is linear over GF(2), so the CRC of the
// message is the XOR of the CRCs of the 256 "column messages" (column t's dwords in place,
// zeros elsewhere).  Walking down a column is one Horner step per row:
//     s <- Z_1024(s ^ dword)              (absorb 4 bytes, then 1020 zero bytes)
// done with four 256-entry LDS tables (the same shape as slicing-by-4, but for a 1 KiB
// stride).  Every row is one fully coalesced 1 KiB load per wave; nothing is staged.
// After its last row a column is advanced over the 4*(256-t) bytes that follow it with one
// GF(2) polynomial multiplication by x^(32*(256-t)) mod P, and the 256 columns are XOR-ed.
// The initial 0xffffffff of the register is the XOR of the first four message bytes with ff.
// ============================================================================================
constexpr uint32_t kCrcThreads = 256;
constexpr uint32_t kCrcPoly = 0x82f63b78u;  // reflected Castagnoli

struct CrcParams {
  const uint8_t* in;
  const uint64_t* off;
  const uint32_t* len;
  uint32_t* crc;
  uint64_t n_units;
  const uint32_t* stride_tab;  // [4][256]: Z_1024(b << 8k)
  const uint32_t* col_mul;     // [256]: x^(32*(256-t)) mod P
  // fixed-size mode (off == nullptr): unit i = in[i*block_len ..], last one short
  uint64_t total_len;
  uint32_t block_len;
  const uint8_t* done;  // != nullptr: units with done[u] != 0 already have their CRC (decode2_kernel.h)
  // != nullptr: the bytes are also COPIED while they are read, unit u to copy_out + copy_off[u] when copy_cap[u]
  // != 0 (a stored chunk of a framed stream: checksummed and delivered in one pass, snappy.nim:244-256)
  uint8_t* copy_out;
  const uint64_t* copy_off;
  const uint32_t* copy_cap;
};

// a(x)*b(x) mod P(x), reflected representation (bit 31 = x^0).
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll 4
  for (int i = 0; i < 32; i++) {
    p ^= (a & 0x80000000u) ? b : 0;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1) ? kCrcPoly : 0);
  }
  return p;
}

__global__ __launch_bounds__(kCrcThreads) void crc32c_units_kernel(CrcParams prm) {
  __shared__ uint32_t s_tab[4][256];
  __shared__ uint32_t s_part[kCrcThreads / 64];
  const uint32_t t = threadIdx.x;
  const uint64_t u = blockIdx.x;
  if (u >= prm.n_units) return;
  if (prm.done && prm.done[u]) return;
  for (uint32_t i = t; i < 1024; i += kCrcThreads) (&s_tab[0][0])[i] = prm.stride_tab[i];
  __syncthreads();

  const uint8_t* msg;
  uint32_t n;
  if (prm.off) {
    msg = prm.in + prm.off[u];
    n = prm.len[u];
  } else {
    const uint64_t pos = u * (uint64_t)prm.block_len;
    msg = prm.in + pos;
    n = (uint32_t)(prm.total_len - pos < prm.block_len ? prm.total_len - pos : prm.block_len);
  }

  uint32_t reg;  // CRC register before the final inversion
  if (n < 4) {   // crc32c.c:204-214 territory: too short for the init trick, do it bitwise
    reg = 0xffffffffu;
    for (uint32_t i = 0; i < n; i++) {
      reg ^= msg[i];
      for (int k = 0; k < 8; k++) reg = (reg >> 1) ^ ((reg & 1) ? kCrcPoly : 0);
    }
    if (t == 0 && prm.copy_out && prm.copy_cap[u])
      for (uint32_t i = 0; i < n; i++) prm.copy_out[prm.copy_off[u] + i] = msg[i];
  } else {
    const uint32_t row_bytes = 4 * kCrcThreads;
    const uint32_t rows = (n + row_bytes - 1) / row_bytes;
    const int64_t pad = (int64_t)rows * row_bytes - n;  // virtual leading zero bytes
    uint32_t s = 0;
    uint8_t* const cp = (prm.copy_out && prm.copy_cap[u]) ? prm.copy_out + prm.copy_off[u] : nullptr;
    auto word = [&](uint32_t r) -> uint32_t {  // my dword of row r
      const int64_t pos = (int64_t)r * row_bytes + 4 * t - pad;
      if (pos >= 4) {
        const uint32_t w = ld32u(msg + pos);
        if (cp) st32u(cp + pos, w);
        return w;
      }
      uint32_t w = 0;  // touches the message start: virtual zero padding and the 0xffffffff init
      for (int k = 0; k < 4; k++) {
        const int64_t j = pos + k;
        if (j >= 0) {
          w |= (uint32_t)(msg[j] ^ (j < 4 ? 0xff : 0)) << (8 * k);
          if (cp) cp[j] = msg[j];
        }
      }
      return w;
    };
    auto step = [&](uint32_t r, uint32_t w) {
      const uint32_t x = s ^ w;
      if (r + 1 < rows) {
        s = s_tab[0][x & 0xff] ^ s_tab[1][(x >> 8) & 0xff] ^ s_tab[2][(x >> 16) & 0xff] ^ s_tab[3][x >> 24];
      } else {
        s = gf2_mulmod(prm.col_mul[t], x);  // the 4*(256-t) bytes from here to the end
      }
    };
    // (the rows' loads do not depend on the register: sixteen go out together -- measured with 4 / 8 / 16 / 32 / 64 rows:
    // the framed 4 GiB stream 8.65 / 8.56 / 8.49 / 8.54 / 8.90 ms --, the chain of table lookups behind them; a row at a
    // time, each waiting for its own load, is bound by the trip to memory)
    // Only rows 0 and 1 can touch the message's start (pad < one row): they go through word(); every later row is a plain
    // dword per thread, and nothing about its load is conditional -- a load under a condition is waited for where the
    // branches join, one trip to HBM after the other.
    uint32_t r = 0;
    for (; r < 2 && r < rows; r++) step(r, word(r));
#ifndef CRC_ROWS_IN_FLIGHT
#define CRC_ROWS_IN_FLIGHT 16
#endif
    constexpr uint32_t kIn = CRC_ROWS_IN_FLIGHT;
    const uint8_t* const mine = msg + 4 * t - pad;  // my dword of row r at mine + r * row_bytes
    for (; r < rows; r += kIn) {  // (a last, shorter batch loads its last row again instead of branching around loads)
      uint32_t w[kIn];
#pragma unroll
      for (uint32_t k = 0; k < kIn; k++) w[k] = ld32u(mine + (uint64_t)(r + k < rows ? r + k : rows - 1) * row_bytes);
      if (cp) {
#pragma unroll
        for (uint32_t k = 0; k < kIn; k++)
          if (r + k < rows) st32u(cp + 4 * t - pad + (uint64_t)(r + k) * row_bytes, w[k]);
      }
#pragma unroll
      for (uint32_t k = 0; k < kIn; k++)
        if (r + k < rows) step(r + k, w[k]);
    }
    for (int d = 32; d >= 1; d >>= 1) s ^= __shfl_xor(s, d, 64);
    if ((t & 63) == 0) s_part[t >> 6] = s;
    __syncthreads();
    reg = s_part[0] ^ s_part[1] ^ s_part[2] ^ s_part[3];
  }
  if (t == 0) {
    const uint32_t crc = ~reg;                                // crc32c.c:761
    prm.crc[u] = ((crc >> 15) | (crc << 17)) + kMaskDelta;    // crc32c.c:762
  }
}

// ============================================================================================
// Packing: the serial `written += ...` of snappy.nim:59-62 / :149-153 as scan + gather.
// ============================================================================================
constexpr uint32_t kScanThreads = 1024;

// offsets[0] = base, offsets[i+1] = offsets[i] + sizes[i].  One workgroup; a pass takes 8 192 sizes (eight
// consecutive ones per thread: their prefix in registers, one DPP scan of the threads' sums per wave, the
// sixteen wave sums through LDS), so 65 536 sizes are eight passes of two barriers each.
__device__ __forceinline__ void scan_sizes_body(const uint32_t* sizes, uint64_t n, uint64_t base, uint64_t* offsets) {
  constexpr uint32_t kPer = 8;
  // (64-bit sums throughout: the sizes of pack and of the frame scans are below 2^17, but the kernel also scans the
  // tile sums of untrusted raw streams and a caller's sizes -- a pass of 8 192 of those can exceed 2^32, and the
  // total this kernel reports is compared with the stream's declared length)
  __shared__ uint64_t s_wave[2][kScanThreads / 64];
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) offsets[0] = base;
  uint64_t carry = base;  // (every thread keeps it: the pass total is read by all)
  uint32_t par = 0;
  auto load = [&](uint64_t c, uint32_t* r) {
    const uint64_t i0 = c + (uint64_t)t * kPer;
    if (i0 + kPer <= n && (((uintptr_t)(sizes + i0)) & 15) == 0) {
      const uint4 a = *reinterpret_cast<const uint4*>(sizes + i0), b = *reinterpret_cast<const uint4*>(sizes + i0 + 4);
      r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    } else {
#pragma unroll
      for (uint32_t k = 0; k < kPer; k++) r[k] = i0 + k < n ? sizes[i0 + k] : 0;
    }
  };
  uint32_t nx[kPer];
  load(0, nx);
  for (uint64_t c = 0; c < n; c += (uint64_t)kScanThreads * kPer, par ^= 1) {
    const uint64_t i0 = c + (uint64_t)t * kPer;
    uint64_t v[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) v[k] = nx[k];
    if (c + (uint64_t)kScanThreads * kPer < n) load(c + (uint64_t)kScanThreads * kPer, nx);  // (the next pass's, under way during this one)
#pragma unroll
    for (uint32_t k = 1; k < kPer; k++) v[k] += v[k - 1];
    uint64_t incl = v[kPer - 1];  // inclusive prefix sum over the wave
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up(incl, d, 64);
      incl += lane >= d ? up : 0;
    }
    const uint64_t before_t = incl - v[kPer - 1];
    if (lane == 63) s_wave[par][wv] = incl;
    __syncthreads();  // (two buffers: the next pass's store cannot overtake this pass's loads)
    uint64_t before_w = 0, pass_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanThreads / 64; k++) {
      const uint64_t w = s_wave[par][k];
      before_w += k < wv ? w : 0;
      pass_total += w;
    }
    const uint64_t at = carry + before_w + before_t;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++)
      if (i0 + k < n) offsets[i0 + k + 1] = at + v[k];
    carry += pass_total;
  }
}

__global__ __launch_bounds__(kScanThreads) void scan_sizes_kernel(const uint32_t* sizes, uint64_t n, uint64_t base,
                                                                  uint64_t* offsets) {
  scan_sizes_body(sizes, n, base, offsets);
}

// Up to four such scans of n sizes each, one workgroup per scan (the framed stream's three chunk-list scans).
struct ScanJobs {
  const uint32_t* sizes[4];
  uint64_t* offsets[4];
  uint64_t n;
};
// gridDim.y > 1: the scan in tiles of one pass (kScanTile sizes) each, a workgroup a tile -- which first adds up every
// size in front of its tile by itself (independent loads, no pass waits for another), then scans its tile from there:
// 65 536 sizes in ~15 us instead of eight dependent passes of one workgroup (62 us).
constexpr uint32_t kScanTile = kScanThreads * 8;  // (= one pass of scan_sizes_body)
__global__ __launch_bounds__(kScanThreads) void scan_sizes_jobs_kernel(ScanJobs j) {
  const uint32_t* const sizes = j.sizes[blockIdx.x];
  uint64_t* const offsets = j.offsets[blockIdx.x];
  if (gridDim.y == 1) {
    scan_sizes_body(sizes, j.n, 0, offsets);
    return;
  }
  const uint64_t t0 = (uint64_t)blockIdx.y * kScanTile;
  if (blockIdx.y && t0 >= j.n) return;  // (tile 0 always runs: offsets[0] is written for n = 0 too)
  __shared__ uint64_t s_part[kScanThreads / 64];
  uint64_t part = 0;
  for (uint64_t i = threadIdx.x; i < t0; i += kScanThreads) part += sizes[i];
#pragma unroll
  for (uint32_t d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
  __syncthreads();
  uint64_t base = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanThreads / 64; k++) base += s_part[k];
  const uint64_t left = j.n - (t0 < j.n ? t0 : j.n);
  scan_sizes_body(sizes + t0, left < kScanTile ? left : kScanTile, base, offsets + t0);
}

// Copy slot i (src = slots + i*stride, sizes[i] bytes) to out + offsets[i].
// 256 threads per slot; destination is made 16-byte aligned so the body is dwordx4 stores.
__global__ __launch_bounds__(256) void gather_slots_kernel(const uint8_t* slots, uint32_t stride,
                                                           const uint32_t* sizes,
                                                           const uint64_t* offsets,
                                                           uint64_t n_blocks, uint8_t* out) {
  const uint64_t b = blockIdx.x;
  if (b >= n_blocks) return;
  const uint8_t* src = slots + b * (uint64_t)stride;
  uint8_t* dst = out + offsets[b];
  const uint32_t n = sizes[b];
  const uint32_t t = threadIdx.x;
  uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
  if (head > n) head = n;
  if (t < head) dst[t] = src[t];
  const uint32_t body = (n - head) & ~15u;
  for (uint32_t i = t * 16; i < body; i += 256 * 16) {
    uint4 v;
    __builtin_memcpy(&v, src + head + i, 16);
    *reinterpret_cast<uint4*>(dst + head + i) = v;
  }
  const uint32_t tail0 = head + body;
  if (tail0 + t < n) dst[tail0 + t] = src[tail0 + t];
}

// ============================================================================================
// Launch order of a decode batch: units of similar compressed length next to each other, longest
// first.  Two workgroups share a CU (and up to 14 index waves), and neighbours of the same kind get
// in each other's way less than a text block next to a run of zeros does: measured on the default
// corpus mix, -9 % decode kernel time and -16 % index pass time against corpus order, and the
// longest units no longer form the tail.  A counting sort by length / 1 KiB; the order inside a
// bucket is whatever the atomics produce (results do not depend on it: every unit has its own
// output range).
// ============================================================================================
constexpr uint32_t kOrderBuckets = 128;
constexpr int kOrderByLength = 0;  // key = compressed length: bucket = length / 1 KiB, longest first
constexpr int kOrderBySketch = 1;  // key = sketch of an input block (below), bucket = key / 4

__device__ __forceinline__ uint32_t order_bucket(uint32_t key, int mode) {
  const uint32_t b = mode == kOrderByLength ? key >> 10 : key >> 2;
  const uint32_t bc = b < kOrderBuckets - 1 ? b : kOrderBuckets - 1;
  return mode == kOrderByLength ? kOrderBuckets - 1 - bc : bc;
}

__global__ __launch_bounds__(256) void order_count_kernel(const uint32_t* keys, uint64_t n, int mode, uint32_t* counts) {
  __shared__ uint32_t s_c[kOrderBuckets];
  if (threadIdx.x < kOrderBuckets) s_c[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&s_c[order_bucket(keys[i], mode)], 1u);
  __syncthreads();
  if (threadIdx.x < kOrderBuckets && s_c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_c[threadIdx.x]);
}

// counts -> first position of every bucket (in place), one wave
__global__ __launch_bounds__(64) void order_scan_kernel(uint32_t* counts) {
  const uint32_t lane = threadIdx.x;
  uint32_t a = counts[2 * lane], b = counts[2 * lane + 1];
  uint32_t x = a + b;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += y;
  }
  const uint32_t before = x - (a + b);
  counts[2 * lane] = before;
  counts[2 * lane + 1] = before + a;
}

// (a workgroup ranks its 256 items per bucket in LDS and reserves each bucket's range with ONE atomic on the
// global cursor: item by item, 65 536 returning atomics on 128 hot words took 0.15 ms)
__global__ __launch_bounds__(256) void order_scatter_kernel(const uint32_t* keys, uint64_t n, int mode, uint32_t* cursor,
                                                            uint32_t* perm) {
  __shared__ uint32_t s_c[kOrderBuckets], s_base[kOrderBuckets];
  static_assert(kOrderBuckets <= 256, "one thread per bucket");
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x; i0 < n; i0 += stride) {  // (uniform trip count)
    if (threadIdx.x < kOrderBuckets) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = i0 + threadIdx.x;
    uint32_t b = 0, rank = 0;
    if (i < n) {
      b = order_bucket(keys[i], mode);
      rank = atomicAdd(&s_c[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kOrderBuckets && s_c[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_c[threadIdx.x]);
    __syncthreads();
    if (i < n) perm[s_base[b] + rank] = (uint32_t)i;
    __syncthreads();
  }
}

// The index pass takes the same units in a SPREAD order: unit i of its launch is entry (i * stride) mod n of the sorted list,
// stride ~ n / golden ratio and coprime to n (a low-discrepancy walk through the list: neighbours in the launch come from
// all over it).  The sorted order runs the corpus's kinds one after the other -- the one-literal units (the longest streams:
// a copy at HBM rate), then text and html (walked: bound by latency, 18 waves a CU), the periods last (64 KiB written
// from a 4 KiB stream) -- so the pass was bound by HBM, then by latency, then by HBM again; spread, the copies and writes run
// beside the walks: index pass 2.08 -> 1.93 ms per 4 GiB.  (The indexed decoder keeps the sorted order: there like next to
// like is what pays.)
__global__ __launch_bounds__(256) void order_spread_kernel(const uint32_t* perm, uint64_t n, uint64_t stride, uint32_t* spread) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) spread[i] = perm[(i * stride) % n];
}

// The encoder's blocks have no length to go by, so blocks that look alike are put next to each
// other: the sketch of a block is the number of distinct values (of 512 possible) among the hashes
// of the 512 aligned dwords of its first 2 KiB -- a handful for runs and short periods, a few hundred
// for text, nearly all 512 for random bytes.  (Text blocks, whose rounds live on L2 hits for their
// candidates, next to blocks that stream literals through the same L2 cost 10-14 % of the encode
// time of the default mix.)  One wave per block.
__global__ __launch_bounds__(64) void encode_sketch_kernel(const uint8_t* in, uint64_t total_len, uint32_t block_len,
                                                          uint64_t n_blocks, uint32_t* sketch) {
  __shared__ uint32_t s_bits[16];
  const uint64_t blk = blockIdx.x;
  if (blk >= n_blocks) return;
  const uint32_t lane = threadIdx.x;
  if (lane < 16) s_bits[lane] = 0;
  wave_fence();
  const uint64_t pos = blk * (uint64_t)block_len;
  const uint8_t* src = in + pos;
  const uint64_t n = total_len - pos < block_len ? total_len - pos : block_len;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t at = 4 * (lane + 64 * k);
    if (at + 4 <= n) {
      const uint32_t h = (ld32u(src + at) * 0x1e35a7bdu) >> 23;  // 9 bits
      atomicOr(&s_bits[h >> 5], 1u << (h & 31));
    }
  }
  wave_fence();
  uint32_t c = lane < 16 ? (uint32_t)__builtin_popcount(s_bits[lane]) : 0;
  for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if (lane == 0) sketch[blk] = c;
}

}  // namespace snappy_hip
