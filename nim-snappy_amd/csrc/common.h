// common.h -- shared definitions for the gfx950 Snappy kernels (wave64, CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snappy_hip {

// ---- format constants (snappy/codec.nim:9-34, :53; snappy/encoder.nim:11-12) ---------------
constexpr uint32_t kMaxBlockLen = 65536;         // codec.nim:14
constexpr uint32_t kInputMargin = 15;            // codec.nim:26
constexpr uint32_t kMinNonLiteral = 17;          // codec.nim:53
constexpr uint32_t kMaxTableBits = 14;           // encoder.nim:11
constexpr uint32_t kMaxTableSize = 1u << kMaxTableBits;
constexpr uint32_t kMaxCompressedBlockLen = 76490;  // codec.nim:217: 32 + 65536 + 65536/6
constexpr uint32_t kMaskDelta = 0xa282ead8u;        // crc32c.c:49

// Status words shared with include/snappy_hip.h (1 + ordinal of the reference's enums).
constexpr uint32_t kOk = 0, kBufferTooSmall = 1, kInvalidInput = 2, kCrcMismatch = 3;
// Internal: the unit does not fit the 64 KiB LDS window, run the whole-stream kernel.
constexpr uint32_t kNeedsStreamKernel = 0x80000000u;
// Internal: the indexed decoder declined the unit (a limit of its fast path), run the one-pass
// block kernel on it.
constexpr uint32_t kNeedsOnePass = 0x80000001u;
// Internal: the indexed decoder's ring-window instantiation passes the unit on to the whole-block one.
constexpr uint32_t kNeedsWindow = 0x80000002u;
// Internal: the index pass has decoded the unit itself (few, long elements: sparse_kernel.h); the decode launches leave it alone,
// the last of them (the one-pass kernel over the declined units) makes it kOk.
constexpr uint32_t kDoneEarly = 0x80000003u;

enum Unit : int { kUnitBody = 0, kUnitRaw = 1, kUnitFrame = 2 };

// Phase ablation (`dbg`) and cycle counters (`stats`) exist only in builds with -DSNAPPY_HIP_DEBUG
// (python nim-snappy_amd/build.py --debug): the shipped library has neither, whatever the
// environment says.
#ifdef SNAPPY_HIP_DEBUG
#define SNAPPY_DBG(prm) ((prm).dbg)
#define SNAPPY_STATS(prm) ((prm).stats)
#else
#define SNAPPY_DBG(prm) 0u
#define SNAPPY_STATS(prm) ((unsigned long long*)nullptr)
#endif

// ---- wave64 helpers --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
// (the builtin on the predicate itself: HIP's __ballot(int) turns the predicate into 0 / 1 and compares that with 0 again --
// two vector instructions per ballot, and the kernels' loops are full of ballots)
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ uint32_t readfirst(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }

// Compiler + hardware ordering point for memory that lanes of ONE wave exchange through LDS
// (LDS operations of a wave execute in issue order; this keeps the compiler from moving them).
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Exclusive prefix sum over the 64 lanes; *total receives the wave sum.  DPP only (row shifts
// inside each row of 16, then row broadcasts), no trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v) {  // lanes without a source read 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t x = v;
  x += dpp_mov0<0x111>(x);  // row_shr:1
  x += dpp_mov0<0x112>(x);  // row_shr:2
  x += dpp_mov0<0x114>(x);  // row_shr:4
  x += dpp_mov0<0x118>(x);  // row_shr:8
  // row_bcast:15 -> lane 15 of a row into the next row (rows 1 and 3 only: the DPP row mask; the others add 0);
  // row_bcast:31 -> lane 31 into lanes 32..63
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
  (void)lane;
  *total = readlane(x, 63);
  return x - v;
}

// Exclusive prefix MAXIMUM over the 64 lanes (0 for lane 0); *total receives the wave maximum.
__device__ __forceinline__ uint32_t wave_excl_scan_max(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t x = v;
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  x = mx(x, dpp_mov0<0x111>(x));  // row_shr:1 (lanes without a source read 0, the identity)
  x = mx(x, dpp_mov0<0x112>(x));
  x = mx(x, dpp_mov0<0x114>(x));
  x = mx(x, dpp_mov0<0x118>(x));
  // (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 only -- the DPP row mask; the other rows get 0, the
  // identity: no per-lane condition, whose lane masks the compiler kept in spilled scalar registers)
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
  (void)lane;
  *total = readlane(x, 63);
  return dpp_mov0<0x138>(x);  // wave_shr:1: the inclusive maximum of the lane before me
}

// Little-endian unaligned accessors.  gfx950 runs with unaligned global / LDS access enabled;
// hipcc lowers these to single global_load_dword / ds_read_b32 style instructions.
template <typename P>
__device__ __forceinline__ uint32_t ld32u(const P* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
template <typename P>
__device__ __forceinline__ void st32u(P* p, uint32_t v) {
  __builtin_memcpy(p, &v, 4);
}

}  // namespace snappy_hip
