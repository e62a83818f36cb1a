// decode2_kernel.h -- pass 2 of the block decoder: build the block in LDS, flush it once.
//
// Semantics: decodeAllTags, snappy/decoder.nim:20-155.  One 512-thread workgroup per unit, two
// workgroups per CU (64 KiB output window + 16 KiB of staging each).  The index written by
// index_units_kernel tells every 16-byte region of the tag stream where its first element
// starts, where that element writes, and how many elements start in the region, so the stream
// is consumed 128 regions (2 KiB) per step with no serial parse:
//
//   waves 0,1  "front end": one region per lane (wave 0 the first KiB of the step, wave 1 the
//           second).  A lane walks its few elements; literal payloads go straight into the output
//           window (literals have no dependencies), and every element leaves its copy offset
//           (0 = literal) and its output position in the step's element list (slot = wave prefix
//           sum of the per-region element counts).  The next 2 KiB of the stream and their index
//           entries are in flight from HBM meanwhile.
//   waves 2-7  "resolvers": the copies of the PREVIOUS step, byte-parallel.  The six waves take the
//           256-byte groups of that step's output in turn, one aligned dword per lane.  A byte
//           finds its element by rank among the element starts of its group (the front end
//           recorded which element covers each 256-byte boundary); a copy byte's source is "own
//           position - offset".  Sources inside the group are followed to a byte that is final
//           (pointer doubling, a few rounds through a 512-byte scratch); sources below the group
//           are final once every group below has published (one frontier, strictly in order): a
//           wave prepares its group ahead of time and gathers, stores and publishes in its turn.
//           The cost does not depend on how long or how deep the copy chains of the data are.
//   all     flush the finished block with 16-byte stores, and -- when the caller wants it -- compute
//           its masked CRC32C from the window.
//
// One workgroup barrier per step separates "list k is complete" from "list k is consumed".
// A step is two chains of dependent LDS round trips side by side (the front end's trips, the resolvers'
// preparations and turns; their instructions run at raised wave priority); with three workgroups on a CU the
// chains hide behind each other and what the kernel is bound by is instructions and LDS operations per output
// byte (ablation in profiles/README.md).  A unit that is one literal is copied HBM to HBM and never gets here.
// Workgroup i takes unit order[i] (units of similar compressed length run next to each other).
//
// The inner loops are written branch-free: a lane that has nothing to store stores to a sink
// slot instead of branching around the store (a taken branch costs more than the store).
#pragma once

#include "common.h"
#include "crc_pack_kernels.h"
#include "index_kernel.h"

namespace snappy_hip {

#ifndef D2_THREADS
#define D2_THREADS 512
#endif
constexpr uint32_t kD2Threads = D2_THREADS;  // waves 0,1: front end; waves 2..: resolvers
constexpr uint32_t kD2Pool = kD2Threads / 64 - 2;
constexpr uint32_t kFeWaves = 2;   // a step's stream (2 KiB) is parsed by the first two waves,
constexpr uint32_t kFeTrips = 4;   // 256 bytes per trip, four trips each
constexpr uint32_t kD2Ring = 4096;
constexpr uint32_t kElemCap = 1024;  // elements per 2 KiB step: the format's maximum (2 bytes each)
constexpr uint32_t kGroup = 256;     // output bytes one resolver wave handles per pass (4 per lane)
// The output window: the whole block (kMaxBlockLen), or a RING of the last kRingWin bytes (see the kernel).
// (Round 4: 16 KiB, FOUR workgroups a CU at 64 registers a lane, instead of 32 KiB and three at 80: the kernel is bound
// by the instructions its resolver waves issue, and 24 of them a CU do a quarter more than 18 -- the sources a smaller ring
// no longer holds are read back from the L2, which was measured to cost next to nothing here)
#ifndef D2_RING_WIN
#define D2_RING_WIN 16384
#endif
constexpr uint32_t kRingWin = D2_RING_WIN;
// The checksumming ring instantiation (RCRC) serves decode_blocks' d_crc only when built with -DD2_FUSED_CRC=1 (a variant
// library of the tests, tests/test_gpu_faults.py).  Shipped: crc32c_units_kernel over the decoded units -- with a ring of
// 16 KiB at 64 registers the checksumming variant spills fifteen of them and costs the ring kernel +0.9 ms per 55 167 chunks,
// the separate pass over their 3.6 GB 0.6 ms less than that and the CRC pass over the index pass's units it replaces
// (framed stream 442 -> 474 GB/s; with the 32 KiB ring of round 3 the fused form was the faster one).
#ifndef D2_FUSED_CRC
#define D2_FUSED_CRC 0
#endif
constexpr bool kD2FusedCrc = D2_FUSED_CRC != 0;
constexpr uint32_t d2_wgs_per_cu(uint32_t win) { return win <= 16384 ? 4 : (win < kMaxBlockLen ? 3 : 1); }  // (by LDS)
#ifndef D2_RING
#define D2_RING 1  // (0: experiments, the whole-block instantiation only)
#endif
constexpr bool kD2RingFirst = D2_RING != 0;
#ifndef D2_RING_CATCHUPS
#define D2_RING_CATCHUPS 32  // (html in a ring of 16 KiB: a catch-up flush every few steps is cheaper than the whole-block window)
#endif
constexpr uint32_t kRingCatchUps = D2_RING_CATCHUPS;  // wide steps (see the kernel) a unit may have in the ring
// dynamic LDS of a launch: the window and 64 scratch dwords behind it, one per lane (no bank conflicts)
constexpr uint32_t out_alloc(uint32_t win) { return win + 256; }
constexpr uint32_t kD2DynWindow = out_alloc(kMaxBlockLen);  // dynamic LDS of the whole-block instantiation's launches
constexpr uint32_t kMaxSteps = kMaxFastIn / kChunk + 2;

struct Decode2Params {
  const uint8_t* in;
  const uint64_t* in_off;
  const uint32_t* in_len;
  uint8_t* out;
  const uint64_t* out_off;
  uint32_t* out_len;  // from the index pass (set to 0 here for a unit this pass rejects)
  uint32_t* status;
  const uint64_t* idx_off;  // nullptr: u * idx_stride
  uint64_t idx_stride;
  const uint32_t* idx;
  uint64_t n_units;
  int unit;
  uint32_t* timeouts;  // [1] turns that were given up on (see the resolvers' wait): never, on a consistent index
  int second;  // the launch after the ring-window one: only the units that one passed on (kNeedsWindow)
  const uint32_t* pass_list;  // ... listed by passed_on_list_kernel; pass_list[-2] = how many
  int dbg;  // timing experiments: 1 no literal payloads, 2 no resolver, 4 no front end, 8 no flush, 16 no read-backs from HBM
  unsigned long long* stats;  // DEBUG counters (nullptr = off)
  // masked CRC32C of every unit's output, computed from the LDS window while it is flushed
  // (nullptr = not wanted); crc_done[u] = 1 where it was written (the units this kernel declines
  // are checksummed by crc32c_units_kernel afterwards)
  uint32_t* crc;
  uint8_t* crc_done;
  const uint32_t* crc_tab;  // CrcParams::stride_tab
  const uint32_t* crc_col;  // CrcParams::col_mul
  uint32_t crc_k32k;        // x^(8 * 32768) mod P: advances a CRC register over 32 KiB
  const uint32_t* order;    // workgroup i takes unit order[i] (nullptr: unit i)
  const uint16_t* tag_lut;  // kTagLut in device memory: what a tag byte says about its element (see the front end)
};

// (keeps a rarely taken condition a BRANCH: the compiler cannot fold what follows an asm statement into scalar selects
// that every pass through the code then executes)
__device__ __forceinline__ bool asm_nop() {
  asm volatile("");
  return true;
}

// Branch-free element decode (decoder.nim:42-109); no validity checks, the index pass did them.
__device__ __forceinline__ void decode_fast(uint32_t tag, uint32_t b14, bool* is_copy, uint32_t* L,
                                            uint32_t* size, uint32_t* hdr, uint32_t* off) {
  const uint32_t t = tag & 3, hi6 = tag >> 2;
  const uint32_t lenlen = hi6 >= 60 ? hi6 - 59 : 0;
  const uint32_t m = lenlen ? (0xffffffffu >> (32 - 8 * lenlen)) : 0;
  const uint32_t Llit = lenlen ? (b14 & m) + 1 : hi6 + 1;
  const uint32_t L1 = 4 + (hi6 & 7), L2 = 1 + hi6;
  const uint32_t off1 = ((tag & 0xe0) << 3) | (b14 & 0xff);
  *is_copy = t != 0;
  *hdr = 1 + lenlen;
  *L = t == 0 ? Llit : (t == 1 ? L1 : L2);
  *off = t == 1 ? off1 : (t == 2 ? (b14 & 0xffff) : b14);
  *size = t == 0 ? 1 + lenlen + Llit : (t == 1 ? 2 : (t == 2 ? 3 : 5));
}

// The copy loops of a run extension (see the resolver), out of line: they are cold for text-like
// data, and keeping them out of the resolver's loop keeps that loop's code and registers tight.
// MASK: window address of output byte x = x & MASK (all ones: the window holds the whole block; the ring
// wraps, and a source's five dwords may lie on both sides of its end: each is addressed on its own).
typedef __attribute__((address_space(3))) uint8_t lds_u8;
template <uint32_t MASK>
__device__ __attribute__((noinline)) void extend_run(lds_u8* out, uint32_t g, uint32_t run_end, uint32_t run_off,
                                                     uint32_t lane) {
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  auto rd = [&](uint32_t a) -> uint32_t { return *reinterpret_cast<const lds_u32*>(out + (a & MASK)); };  // a: 4-aligned
  // W = the smallest multiple of the offset that is >= kGroup: then W - offset < kGroup, i.e. for
  // x >= g + kGroup the source x - W is not below g - offset, the first byte the run's own chain of
  // copies reaches from x (a larger multiple could read bytes from before the run)
  const uint32_t W = run_off * ((kGroup - 1 + run_off) / run_off);
  // ... and once the run is long enough behind x, 1 KiB per trip (16 bytes per lane) with W4, the
  // smallest multiple >= 1024: x - W4 >= g - offset needs x >= g + W4 - offset
  const uint32_t W4 = run_off * ((1023 + run_off) / run_off);
  uint32_t x1 = (g + W4 - run_off + kGroup - 1) & ~(kGroup - 1);  // first group start that may use W4
  x1 = x1 > g + kGroup ? x1 : g + kGroup;
  x1 = x1 < run_end ? x1 : run_end;
  for (uint32_t x = g + kGroup + 4 * lane; x < x1; x += kGroup) {
    const uint32_t src = x - W, sa = src & ~3u;
    *reinterpret_cast<lds_u32*>(out + (x & MASK)) = __funnelshift_r(rd(sa), rd(sa + 4), (src & 3) * 8);
  }
  asm volatile("" ::: "memory");
  for (uint32_t x = x1 + 16 * lane; x < run_end; x += 1024) {  // (x1, run_end: multiples of 256)
    const uint32_t src = x - W4, sa = src & ~3u;
    const uint32_t r0 = rd(sa), r1 = rd(sa + 4), r2 = rd(sa + 8), r3 = rd(sa + 12), r4 = rd(sa + 16);
    const uint32_t sh8 = (src & 3) * 8;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v.x = __funnelshift_r(r0, r1, sh8);
    v.y = __funnelshift_r(r1, r2, sh8);
    v.z = __funnelshift_r(r2, r3, sh8);
    v.w = __funnelshift_r(r3, r4, sh8);
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(out + (x & MASK)) = v;
  }
}

// The whole-block output window is DYNAMIC shared memory (launch with kD2DynWindow bytes): with the whole
// footprint declared statically the compiler derives "at most N waves per SIMD" from it and pads
// the kernel's VGPR allocation to enforce that -- which can leave no room for the second
// workgroup of a CU (measured: one workgroup per CU with 5 waves per workgroup).
extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn_window[];

// WIN = kMaxBlockLen: the window holds the whole unit, two workgroups per CU.
// WIN = kRingWin: the window is a ring of the unit's last WIN output bytes -- out[x] lives at x & (WIN - 1)
// -- so that FOUR workgroups fit a CU (16 KiB and 64 registers a lane; round 3: three, with 32 KiB and 80: the kernel
// is bound by what its resolver waves issue, and a CU hides one block's chain behind the others': profiles/README.md).  Every step starts by writing the bytes that
// became final one step ago to HBM; a copy whose source is older than what this step leaves of the ring
// (ring_lo) reads it back from there -- those bytes were written at least a step earlier (far_lo, with a
// wait for the stores in front of the barrier in between).  The ring is sound while the output of the step
// being parsed, the step being resolved and the one before it fit in WIN bytes; where they do not (a "wide"
// step: much output from little stream) everything final is written and waited for at once, after which two
// steps must fit.  A unit with a step wider than that, with more than kRingCatchUps wide steps, or with an
// output that is not 16-byte aligned is handed to the whole-block instantiation (kNeedsWindow), launched
// second over the same units.
// RCRC (ring only): the unit's masked CRC32C is computed WHILE the ring is flushed -- the column scheme of
// crc32c_units_kernel (the message right-aligned in rows of 1 KiB, one dword column per thread, one Horner
// step `s <- Z1024(s ^ dword)` per row): a row is final, and still in the ring, when its bytes are flushed, so
// threads 256..511 take the rows the flush has just completed, their column registers live across the steps.
// The framed stream (uncompressFramed, snappy.nim:231) then decodes on the ring kernel too.
// (__launch_bounds__' second argument is waves per SIMD: 8 = four workgroups of eight waves a CU.  It also caps the scalar
// registers -- 80 at eight waves a SIMD, the trap handler's 16 set aside: 45 are spilled to the lanes of a vector register;
// at 7 the compiler takes 72 vector registers, and four workgroups no longer fit.  Round 6 measured that those spills are
// ~3 % of what a block executes, profiles/r06_instruction_table.md.)
template <uint32_t WIN, bool RCRC = false>
__global__ __launch_bounds__(kD2Threads, WIN < kMaxBlockLen ? (kD2Threads / 64 * d2_wgs_per_cu(WIN) + 3) / 4 : 1) void decode_indexed_kernel(Decode2Params prm) {
  constexpr bool RING = WIN < kMaxBlockLen;
  static_assert(RING || !RCRC, "the whole-block instantiation checksums its window at the end");
  constexpr uint32_t kOutSink = WIN;
  constexpr uint32_t kSlack = RCRC ? 1024 : 0;
  auto wa = [](uint32_t x) -> uint32_t { return RING ? (x & (WIN - 1)) : x; };  // window address of output byte x
  // (the ring window is a static array: its LDS address is then a compile-time constant that folds into the
  // instructions' offset fields -- the dynamic array's base is added to every address with an instruction;
  // -2.5 % kernel time)
  __shared__ __attribute__((aligned(16))) uint8_t s_static_window[RING ? out_alloc(WIN) : 16];
  uint8_t* const s_out = RING ? s_static_window : s_dyn_window;
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kD2Ring + 16];
  // pointer-doubling / start-mask scratch, one per resolver wave
  constexpr uint32_t kPool = kD2Pool;
  constexpr uint32_t kPoolFirst = kFeWaves;  // first wave that resolves
  __shared__ __attribute__((aligned(16))) uint16_t s_r16[kPool][kGroup];
  // element lists of the current and the previous step, in stream order
  // (one dword per element: low half the copy offset, 0 = literal; high half its first output byte -- the
  // front end writes an entry with one store)
  __shared__ __attribute__((aligned(4))) uint32_t s_el[2][kElemCap];
  __shared__ uint32_t s_sbase[kMaxSteps];   // output position where step k starts
  __shared__ uint32_t s_cnt[2];             // elements in the list
  __shared__ uint32_t s_front;              // every output byte below this position is final
  __shared__ uint32_t s_err;
  // a long literal that covers whole steps: (first step after it) << 16 | (first step that may be skipped)
  __shared__ uint32_t s_skip;
  // list slot of the element covering byte 256 m (eight entries of slack: the resolvers look one round of groups ahead,
  // past the unit's end at its last step -- what they read there is never used)
  __shared__ uint16_t s_gidx[kMaxBlockLen / kGroup + 8];
  // s + 1 where step s (list s & 1) holds a literal with length bytes that covers whole groups: only then do the
  // resolvers of that step look for groups to skip (text has next to none: a look per group saved)
  __shared__ uint32_t s_big[2];
  __shared__ uint32_t s_crc_acc, s_crc_cnt;            // the unit's CRC: XOR of the waves' parts, waves done
  __shared__ uint32_t s_runbad;                          // the unit is not one literal + copies of one offset
  __shared__ uint32_t s_rcrc_tab[RCRC ? 1024 : 1];       // (ring + CRC) the four stride tables of the column scheme
  // the front end's tag table: in LDS where there is room for it.  (Two workgroups of the whole-block instantiation
  // have 32 bytes to spare; three of the checksumming one fit a CU up to 53 760 bytes each -- LDS is handed out in
  // pieces of 1 280 bytes: with the table, 53 976 bytes, only two fit and the kernel takes 9.5 ms instead of 7.9.
  // Those two read the table through the vector cache.)
  constexpr bool kLutInLds = RING && !RCRC;
  __shared__ uint16_t s_lut[kLutInLds ? 256 : 2];

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = readfirst(tid >> 6);  // (scalar register: wave-uniform by construction)
  if (blockIdx.x >= prm.n_units) return;
  // (the launch behind the ring-window one: workgroup i takes the i-th unit on the list of passed-on units --
  // a workgroup with nothing to do leaves after one load, not after a chain of two)
  if (!RING && prm.second && blockIdx.x >= prm.pass_list[-2]) return;
  const uint64_t u = (!RING && prm.second) ? prm.pass_list[blockIdx.x]
                                           : (prm.order ? prm.order[blockIdx.x] : blockIdx.x);  // (launch order, crc_pack_kernels.h)
  // The start of a workgroup is a chain of dependent trips to HBM (which unit -> its parameters ->
  // its first bytes -> its index and stream), and a block's latency is what this kernel is bound by:
  // everything that can go out together does.  First every per-unit parameter ...
  const uint32_t st0 = prm.status[u];
  const uint32_t total = prm.out_len[u];
  const uint64_t in_off = prm.in_off[u];
  const uint32_t n_all = prm.in_len[u];
  const uint64_t out_off = prm.out_off[u];
  const uint64_t idx_base = prm.idx_off ? prm.idx_off[u] : u * prm.idx_stride;
  // (the index pass already decided every other unit; kNeedsWindow: the ring instantiation passed it on)
  if (st0 != (prm.second ? kNeedsWindow : kOk)) return;
  if (total == 0) return;
  if (!RING && st0 == kNeedsWindow && tid == 0) prm.status[u] = kOk;  // (mine now; a failure below overwrites it)

  // ... then, knowing only where the unit lies: its first bytes (the varint of a raw unit, the first
  // tag), the first index entries and the first 4 KiB of the stream.  The ring is laid out from the
  // unit's first byte, header included, so that its fill does not wait for the header's length.
  const uint8_t* const unit = prm.in + in_off;
  uint8_t* gout = prm.out + out_off;
  const uint32_t* idx = prm.idx + idx_base;
  uint32_t fb[10];
#pragma unroll
  for (uint32_t k = 0; k < 10; k++) fb[k] = unit[k < n_all ? k : n_all - 1];
  const uint32_t shift0 = (uint32_t)((uintptr_t)unit & 15);
  const uint8_t* g0 = unit - shift0;
  const uint32_t q_end = (uint32_t)(((uint64_t)shift0 + n_all + 15) & ~15ull);
  const uint32_t n_regions0 = (n_all + kSub - 1) / kSub;  // (>= the regions of the tag stream; all inside the unit's index)
  // The stream ring is kept by the LAST wave: its 256 bytes of a step are the only ones whose look-ahead (an element's
  // bytes, a short literal's payload) reaches into the next step's bytes, which land in the ring at the start of
  // the step -- the wave that reads them is the wave that stored them, and a wave's LDS operations execute in order.
  const bool ringw = wave == kFeWaves - 1;
  uint4 rf[4];  // (ring wave) the ring's first 4 KiB
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t q = (lane + 64 * i) * 16;
    rf[i] = make_uint4(0, 0, 0, 0);
    if (ringw && q < q_end) rf[i] = *reinterpret_cast<const uint4*>(g0 + q);
  }
  const uint32_t ie0 = lane < n_regions0 ? idx[lane] : 0;            // the first step's index entries, regions 0..63
  const uint32_t io0 = 64 + lane < n_regions0 ? idx[64 + lane] : 0;  // ... and 64..127
  const uint32_t sb0 = tid * 128 < n_regions0 ? idx[tid * 128] : 0;  // (kMaxSteps <= the workgroup's threads)
  static_assert(kMaxSteps <= kD2Threads, "one s_sbase entry per thread");

  uint32_t hdr = 0;
  if (prm.unit == kUnitRaw) {  // skip the varint (validated by the index pass: at most 5 bytes)
    while (hdr < 4 && (fb[hdr] & 0x80)) hdr++;
    hdr++;
  }
  const uint8_t* in0 = unit + hdr;
  const uint32_t n = n_all - hdr;

  // ---- the unit's first element, if it is a literal (fb[hdr ..]: the tag and its length bytes;
  // hdr + 1 + 4 <= 10): lit0_L payload bytes behind lit0_h bytes of tag and length ------------------
  uint32_t lit0_L = 0, lit0_h = 0;
  {
    uint32_t tb[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) {
      tb[k] = fb[k];
#pragma unroll
      for (uint32_t h2 = 1; h2 <= 5; h2++) tb[k] = hdr == h2 ? fb[k + h2] : tb[k];
    }
    const uint32_t tag = tb[0];
    const uint32_t hi6 = tag >> 2;
    const uint32_t lenlen = hi6 >= 60 ? hi6 - 59 : 0;
    if ((tag & 3) == 0 && 1 + lenlen <= n && !(SNAPPY_DBG(prm) & 64)) {
      uint32_t L = hi6 + 1;
      if (lenlen) {
        const uint32_t b = tb[1] | (tb[2] << 8) | (tb[3] << 16) | (tb[4] << 24);
        L = (lenlen == 4 ? b : (b & ((1u << (8 * lenlen)) - 1))) + 1;
      }
      lit0_L = L;
      lit0_h = 1 + lenlen;
    }
  }
  // ---- a unit that is ONE literal (what encodeBlock makes of incompressible data, encoder.nim:249-253):
  // nothing to resolve, nothing to stage -- the payload goes straight from HBM to HBM. ----------------
  if (lit0_L == total && n - lit0_h == lit0_L) {  // (the index pass has validated the element and the total)
    const uint8_t* src = in0 + lit0_h;
    if (((uintptr_t)gout & 15) == 0) {
      for (uint32_t i = tid * 16; i < total; i += kD2Threads * 16) {
        if (i + 16 <= total) {
          uint4 v;
          __builtin_memcpy(&v, src + i, 16);  // (unaligned on the load side, where it is free)
          *reinterpret_cast<uint4*>(gout + i) = v;
        } else {
          for (uint32_t k = i; k < total; k++) gout[k] = src[k];
        }
      }
    } else {
      for (uint32_t i = tid * 4; i < total; i += kD2Threads * 4) {
        if (i + 4 <= total) {
          st32u(gout + i, ld32u(src + i));
        } else {
          for (uint32_t k = i; k < total; k++) gout[k] = src[k];
        }
      }
    }
    return;  // (no CRC here: crc_done stays 0 and the CRC kernel takes the unit)
  }

  const uint32_t shift = shift0 + hdr;  // ring position of the tag stream's first byte (<= 20)
  const uint32_t n_chunks = (n + kChunk - 1) / kChunk;   // 2 KiB steps
  const uint32_t n_regions = (n + kSub - 1) / kSub;       // 16-byte index entries

  if (kLutInLds && tid < 256) s_lut[tid] = prm.tag_lut[tid];  // (visible after the barrier below)
  if (tid == 0) {
    s_front = 0;
    s_skip = 0;
    s_err = 0;
    s_cnt[0] = 0;
    s_cnt[1] = 0;
    s_crc_acc = 0;
    s_crc_cnt = 0;
    s_runbad = 0;
    s_big[0] = 0;
    s_big[1] = 0;
  }

  // ring[q & 4095] = stream byte q - shift; at the start of step s it holds q in
  // [2048 s, 2048 s + 4096)
  auto ring_store = [&](uint32_t q, uint4 v) {
    const uint32_t i = q & (kD2Ring - 1);
    *reinterpret_cast<uint4*>(s_ring + i) = v;
    if (i == 0) *reinterpret_cast<uint4*>(s_ring + kD2Ring) = v;
  };
  // Stores that a lane must not perform go to its private sink dword instead of being branched
  // around.  (One shared sink address would serialise the 64 lanes on one LDS bank.)
  const uint32_t sink = kOutSink + lane * 4;
  uint16_t* const sink16 = reinterpret_cast<uint16_t*>(s_out + sink);

  // keeps the compiler from reordering LDS traffic across it; the hardware executes the LDS
  // operations of one wave in issue order, one instruction at a time for the whole CU
  auto cbar = [] { asm volatile("" ::: "memory"); };

  auto fill_ring = [&](uint32_t step) {  // (ring wave) the 4 KiB of the stream that start with `step`
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = step * kChunk + (lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, *reinterpret_cast<const uint4*>(g0 + q));
    }
  };
  if (ringw) {  // land what was fetched at the start
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = (lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, rf[i]);
    }
  }

  // Index entries of the NEXT step (mine and the other front-end wave's) are fetched at the start
  // of a step and consumed at the start of the next one, before anything younger is issued, so
  // the wait for them never also waits for fresh loads.  Entries past the end read as "none".
  const uint32_t none_end = idx_none(total);
  auto idx_at = [&](uint32_t r) -> uint32_t { return r < n_regions ? idx[r] : none_end; };
  uint32_t ie_pref = lane < n_regions ? ie0 : none_end;       // regions 0..63 of the step
  uint32_t io_pref = 64 + lane < n_regions ? io0 : none_end;  // regions 64..127
  if (tid <= n_chunks && tid < kMaxSteps) s_sbase[tid] = tid < n_chunks ? idx_first_dst(sb0) : total;
  __syncthreads();

  // ---- a unit that is one literal followed by copies that all have ONE offset (what encodeBlock
  // makes of a period: zeros, a repeating pattern, a ramp -- every match is found at the same
  // distance and emitCopy cuts it into copy2 elements, encoder.nim:97-120): its output is periodic
  // behind the literal, out[x] = out[x - offset], so it is written straight to HBM from one image of
  // the period in LDS.  The whole tag stream is in the ring; the index pass has validated it as a
  // sequence of elements, so "every third byte behind the literal is a copy2 tag" proves that those
  // are the element starts.  (decoder.nim:112: 1 <= offset <= the literal's length.)
  if (lit0_L && lit0_L < total && ((uintptr_t)gout & 15) == 0 && n + shift <= kD2Ring &&
      lit0_h + lit0_L + 3 <= n && (n - lit0_h - lit0_L) % 3 == 0) {
    const uint32_t q0 = lit0_h + lit0_L;  // the first copy
    const uint32_t nrec = (n - q0) / 3;
    auto rb = [&](uint32_t t) -> uint32_t { return s_ring[t + shift]; };
    const uint32_t roff = rb(q0 + 1) | (rb(q0 + 2) << 8);
    if ((rb(q0) & 3) == 2 && roff >= 1 && roff <= lit0_L) {
      bool okr = true;
      for (uint32_t k = tid; k < nrec; k += kD2Threads) {
        const uint32_t t = q0 + 3 * k;
        okr = okr && (rb(t) & 3) == 2 && (rb(t + 1) | (rb(t + 2) << 8)) == roff;
      }
      if (!okr) s_runbad = 1;
      __syncthreads();
      if (s_runbad == 0) {
        // image of the period: rep[i] = out[L - offset + i mod offset] for i < M + 16, M a multiple
        // of the offset of about 4 KiB
        const uint32_t M = roff * (4096 / roff > 0 ? 4096 / roff : 1);
        const uint32_t pb = lit0_h + lit0_L - roff;  // stream position of the period's first byte
        for (uint32_t i0 = tid * 16; i0 < M + 16; i0 += kD2Threads * 16) {
          uint32_t r = i0 % roff;
#pragma unroll
          for (uint32_t j = 0; j < 16; j++) {
            s_out[i0 + j] = (uint8_t)rb(pb + r);
            r = r + 1 == roff ? 0 : r + 1;
          }
        }
        __syncthreads();
        const uint32_t* const rep32 = reinterpret_cast<const uint32_t*>(s_out);
        for (uint32_t x = tid * 16; x < total; x += kD2Threads * 16) {
          if (x >= lit0_L && x + 16 <= total) {
            const uint32_t ix = (x - lit0_L) % M;
            const uint32_t a = ix >> 2, sh8 = (ix & 3) * 8;
            const uint32_t r0 = rep32[a], r1 = rep32[a + 1], r2 = rep32[a + 2], r3 = rep32[a + 3], r4 = rep32[a + 4];
            *reinterpret_cast<uint4*>(gout + x) =
                make_uint4(__funnelshift_r(r0, r1, sh8), __funnelshift_r(r1, r2, sh8), __funnelshift_r(r2, r3, sh8),
                           __funnelshift_r(r3, r4, sh8));
          } else {
            for (uint32_t k = x; k < x + 16 && k < total; k++)
              gout[k] = k < lit0_L ? (uint8_t)rb(lit0_h + k) : s_out[(k - lit0_L) % M];
          }
        }
        return;  // (no CRC here: crc_done stays 0 and the CRC kernel takes the unit)
      }
    }
  }
  // ---- ring window: is the unit one for it? (see above) ----
  uint32_t flushed = 0;  // (ring) every output byte below this has been written to HBM ...
  uint32_t far_lo = 0;   // ... and below this, a step earlier: visible to the whole workgroup
  uint32_t ring_lo = 0;  // what this step leaves of the ring: positions from here on (<= far_lo)
  if (RING) {
    // (steps that need the catch-up flush in the loop -- three steps do not fit, two do -- cost a barrier and
    // a trip to the L2 each: a unit with more than a few, e.g. repeated long strings, is better off with the
    // whole-block window)
    bool wide = ((uintptr_t)gout & 15) != 0;  // (the ring is flushed in aligned 16-byte pieces)
    bool catch_up = false;
    if (tid <= n_chunks && tid < kMaxSteps) {
      const uint32_t hi = s_sbase[tid + 1 <= n_chunks ? tid + 1 : n_chunks];
      const uint32_t lo = tid >= 1 ? s_sbase[tid - 1] & ~15u : 0;
      const uint32_t lo3 = tid >= 2 ? s_sbase[tid - 2] & ~15u : 0;
      // (RCRC: a row that the flush completes starts up to 1 023 bytes below the previous step's flush
      // mark, and must still be in the ring: a KiB of slack in every ring condition)
      wide = wide || hi + kSlack - lo > WIN;
      catch_up = hi + kSlack - lo3 > WIN;
    }
    const uint32_t n_catch = __syncthreads_count(catch_up);
    if (__syncthreads_or(wide) || n_catch > kRingCatchUps) {
      if (tid == 0) prm.status[u] = kNeedsWindow;
      return;
    }
  }
  // ---- (ring + CRC) the column scheme's state: rows of 1 KiB of the right-aligned message ----------------
  const bool do_crc = RCRC && prm.crc != nullptr && total >= 4;  // (shorter units: the CRC kernel, crc_done stays 0)
  const uint32_t crc_rows_n = (total + 1023) / 1024;
  const uint32_t crc_pad = crc_rows_n * 1024 - total;  // virtual zero bytes in front of the message
  uint32_t crc_row = 0;   // rows done (uniform)
  uint32_t crc_reg = 0;   // my column's register (threads 256..511: column tid - 256)
  if (do_crc) {
    for (uint32_t i = tid; i < 1024; i += kD2Threads) s_rcrc_tab[i] = prm.crc_tab[i];
    // (visible to waves 4-7 after the next barrier; the first rows are taken in step 1 at the earliest)
  }
  // rows that end at or below `to` (a position everything below which is final and still in the ring)
  auto crc_rows_to = [&](uint32_t to) {
    const uint32_t upto_row = (to + crc_pad) / 1024;  // rows [crc_row, upto_row) are complete
    if (tid >= 256 && tid < 512) {
      const uint32_t t = tid - 256;
      const uint32_t sh8 = (total & 3) * 8;  // byte phase of the columns' dwords (0 for whole blocks)
      const uint32_t* const o32 = reinterpret_cast<const uint32_t*>(s_out);
      for (uint32_t r = crc_row; r < upto_row; r++) {
        const int32_t pos = (int32_t)(r * 1024 + 4 * t) - (int32_t)crc_pad;
        uint32_t w;
        if (pos >= 4) {  // aligned dwords + funnel shift; the two may lie on both sides of the ring's end
          const uint32_t a = (uint32_t)pos & ~3u;
          w = __funnelshift_r(o32[wa(a) >> 2], o32[wa(a + 4) >> 2], sh8);
        } else {  // the message's start: virtual zero padding in front of it, the 0xffffffff init on its first 4 bytes
          w = 0;
          for (int k = 0; k < 4; k++) {
            const int32_t j = pos + k;
            if (j >= 0) w |= (uint32_t)(s_out[wa((uint32_t)j)] ^ (j < 4 ? 0xff : 0)) << (8 * k);
          }
        }
        const uint32_t x = crc_reg ^ w;
        if (r + 1 < crc_rows_n) {
          crc_reg = s_rcrc_tab[x & 0xff] ^ s_rcrc_tab[256 + ((x >> 8) & 0xff)] ^ s_rcrc_tab[512 + ((x >> 16) & 0xff)] ^
                    s_rcrc_tab[768 + (x >> 24)];
        } else {
          crc_reg = gf2_mulmod(prm.crc_col[t], x);  // the 4 * (256 - t) bytes from here to the message's end
        }
      }
    }
    crc_row = upto_row > crc_row ? upto_row : crc_row;
  };
  // per-lane constants of the front end
  const uint32_t fe_bp = (lane >> 2) * 4;     // ds_bpermute address of my region's index entry in a half's first trip
  const uint32_t fe_nsh = (lane & 3) * 4;     // my quad's bits of the region's start mask
  const uint32_t fe_q = lane * 4 + shift;     // ring position of my quad, counted from a trip's first byte
  const uint32_t fe_q1 = (lane & 3) >= 1 ? 0xffffffffu : 0u, fe_q2 = (lane & 3) >= 2 ? 0xffffffffu : 0u;
  bool passed_on = false;
#ifdef D2_INJECT_GIVE_UP
  uint32_t d2_inject = 0;
#endif
  uint4 pre[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  uint32_t pq[2] = {0xffffffffu, 0xffffffffu};  // ring data in flight (ring wave)

  unsigned long long tm_work = 0, tm_bar = 0;             // DEBUG timers
  unsigned long long tt_flush = 0, tt_fe = 0, tt_prep = 0, tt_wait = 0, tt_turn = 0;  // DEBUG: where a step's time goes
  auto now = [&]() -> unsigned long long { return SNAPPY_STATS(prm) ? __builtin_amdgcn_s_memtime() : 0ull; };
  uint32_t acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0;    // DEBUG counters, flushed once per wave
  for (uint32_t s = 0; s <= n_chunks; s++) {
    if (s_err) break;  // set before the last barrier: every wave sees it here
    {
      // Steps that lie entirely inside one long literal hold no element and produce no output of
      // their own (the literal was copied when its tag was met): instead of paying a barrier and a
      // prefetch round trip for each of them, go straight to the step where the literal ends.
      const uint32_t sk = readfirst(s_skip);
      const uint32_t to = sk >> 16, from = sk & 0xffffu;
      if (s >= from && s < to) {
        s = to < n_chunks ? to : n_chunks;
        if (ringw && s < n_chunks) fill_ring(s);
        ie_pref = idx_at(s * 128 + lane);
        io_pref = idx_at(s * 128 + 64 + lane);
        pq[0] = pq[1] = 0xffffffffu;  // nothing in flight for the ring
        __syncthreads();
      }
    }
    const unsigned long long tm0 = SNAPPY_STATS(prm) ? __builtin_amdgcn_s_memtime() : 0;
    if (RING) {
      // the last barrier made everything below s_sbase[s - 1] final (the list of step s - 2 is resolved) and
      // the previous step's flush visible
      far_lo = flushed;
      const uint32_t hi = readfirst(s_sbase[s + 1 <= n_chunks ? s + 1 : n_chunks]);
      const uint32_t upto = s >= 1 ? readfirst(s_sbase[s - 1]) & ~15u : 0;
      auto flush_to = [&](uint32_t to) {
        for (uint32_t i = flushed + tid * 16; i < to; i += kD2Threads * 16)
          *reinterpret_cast<uint4*>(gout + i) = *reinterpret_cast<const uint4*>(s_out + wa(i));
        flushed = to > flushed ? to : flushed;
      };
      if (hi + kSlack - far_lo > WIN) {
        // a wide step (a stretch of copies: much output from little stream): what is final now is written
        // and waited for at once, so that the ring has to hold two steps only -- a wait for the stores
        // and a barrier, paid by the steps that produce many bytes
        flush_to(upto);
        if (do_crc) crc_rows_to(upto);  // (before the barrier: the front end may overwrite these bytes after it)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        far_lo = flushed;
        if (hi + kSlack - far_lo > WIN) {  // (after a fast-forward the steps are not consecutive: the check before the loop missed it)
          passed_on = true;
          break;
        }
      }
      ring_lo = hi > WIN ? hi - WIN : 0;  // (nothing this step writes lies at or beyond hi)
      flush_to(upto);
      if (do_crc) crc_rows_to(upto);  // (these rows stay in the ring for the whole step: upto >= far_lo >= ring_lo)
    }
    // ---- prefetch hand-over (all waves, straight-line) ---------------------------------------------
    // everything fetched during the previous step is consumed here, BEFORE new loads are issued
    // (the empty asm pins the wait to this point)
    asm volatile("" ::"v"(ie_pref), "v"(io_pref), "v"(pre[0].x), "v"(pre[0].w), "v"(pre[1].x),
                 "v"(pre[1].w));
    const uint32_t ie = ie_pref, io_cur = io_pref;  // index entries of this step
    if (ringw && s < n_chunks) {
      // the ring slots of the previous step are free now: land the 2 KiB fetched meanwhile
#pragma unroll
      for (int i = 0; i < 2; i++)
        if (pq[i] < q_end) ring_store(pq[i], pre[i]);
    }
    // next step's index entries and the 2 KiB of stream after the ring's contents
    // (the front-end waves only: the resolvers have no use for them)
    if (wave < kFeWaves) {
      ie_pref = idx_at((s + 1) * 128 + lane);
      io_pref = idx_at((s + 1) * 128 + 64 + lane);
    }
    if (ringw) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        pq[i] = s * kChunk + kD2Ring + (lane + 64 * i) * 16;
        const uint32_t qc = pq[i] < q_end ? pq[i] : 0;  // clamped: always a valid address
        pre[i] = *reinterpret_cast<const uint4*>(g0 + qc);
      }
    }

    const unsigned long long tq0 = now();
    tt_flush += tq0 - tm0;
    if (s < n_chunks && wave < kFeWaves && !(SNAPPY_DBG(prm) & 4)) {
      // =================================== front end ===========================================
      // Two waves, four trips each; a trip takes 256 bytes of the step's stream, four per lane ("quad").  The
      // index says where the elements start, so nothing here is a walk and no trip waits for another: a quad
      // holds at most two starts (an element is at least two bytes long), A and B; both are decoded at once (a
      // 256-entry table by tag byte), their output positions are the region's first position plus a prefix
      // sum of lengths over the region's four lanes, their list slots a prefix sum of start counts.  Literal
      // payloads are stored by the lanes that hold the payload BYTES, in three runs per lane: what the last
      // short literal of an earlier lane ("C": prefix maximum + one cross-lane read) leaves of its payload in my
      // quad, A's payload bytes in my quad, B's.  Literals with length bytes (61 bytes and more) are copied by
      // the whole wave straight from HBM.
      // (The step is bound by the resolvers' chain of turns behind one group's preparation -- timers in
      // profiles/README.md; the front end only has to stay off that path: two waves of it take about 2/3 of a step.)
      __builtin_amdgcn_s_setprio(3);
      const uint32_t buf = s & 1;
      uint32_t* const el = s_el[buf];
      // ---- how many elements start in front of a trip's 256 bytes: one inclusive scan over both halves' counts
      // (regions 0..63 in the low half of a dword, 64..127 in the high half) ----
      uint32_t cscan = __builtin_popcount(idx_starts(ie)) | (__builtin_popcount(idx_starts(io_cur)) << 16);
      cscan += dpp_mov0<0x111>(cscan);  // row_shr:1, 2, 4, 8
      cscan += dpp_mov0<0x112>(cscan);
      cscan += dpp_mov0<0x114>(cscan);
      cscan += dpp_mov0<0x118>(cscan);
      cscan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cscan, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
      cscan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cscan, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
      const uint32_t call = readlane(cscan, 63);
      if (wave == 0 && lane == 0) s_cnt[buf] = (call & 0xffffu) + (call >> 16);  // (<= kElemCap by the format)
      bool bad = false, far_off = false;
#pragma unroll 1
      for (uint32_t trip = 0; trip < kFeTrips; trip++) {
      const uint32_t vw = wave * kFeTrips + trip;  // which 256 bytes of the step
      uint32_t base, mine;
      {
        const uint32_t upto = readlane(cscan, trip * 16 + 15);                // regions of my half up to the trip's last
        const uint32_t below = trip ? readlane(cscan, trip * 16 - 1) : 0u;    // ... and below its first
        const uint32_t sh = wave * 16;
        const uint32_t before_half = (below >> sh) & 0xffffu;
        mine = ((upto >> sh) & 0xffffu) - before_half;
        base = before_half + (wave ? (call & 0xffffu) : 0u);
      }
      bool big = false;  // a literal with length bytes is my quad's last element: done below
      uint32_t big_dst = 0, big_len = 0, big_src = 0, big_slot = 0;
      if (mine) {  // (nothing starts in my 256 bytes: the payload of a long literal, or the stream's end)
        acc_a++;
        const uint32_t ent = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(fe_bp + trip * 64), (int)(wave == 0 ? ie : io_cur));
        const uint32_t nib = (idx_starts(ent) >> fe_nsh) & 15u;
        const uint32_t rdst = ent >> 16;
        const uint32_t xr = lane * 4;                        // my first byte, counted from the wave's
        const uint32_t xw = s * kChunk + vw * 256;           // the trip's first byte: stream position
        // my four bytes and the eight behind them (the ring position is not dword-aligned: four aligned
        // dwords, three funnel shifts by a uniform amount; the ring's first 16 bytes are mirrored behind it)
        const uint32_t* const ra = reinterpret_cast<const uint32_t*>(s_ring + ((fe_q + xw) & (kD2Ring - 1) & ~3u));
        const uint32_t a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3];
        const uint32_t sh8 = readfirst((shift & 3) * 8);
        const uint32_t w0 = __funnelshift_r(a0, a1, sh8), w1 = __funnelshift_r(a1, a2, sh8), w2 = __funnelshift_r(a2, a3, sh8);
        // ---- my (up to) two elements: A at byte bA, B at byte bB (4 = none) ----
        const uint32_t nib2 = nib & (nib - 1);
        const bool hasA = nib != 0, hasB = nib2 != 0;
        const uint32_t bA = __builtin_ctz(nib | 16u), bB = __builtin_ctz(nib2 | 16u);
        const uint32_t dA = __funnelshift_r(w0, w1, 8 * bA), dB = __funnelshift_r(w0, w1, 8 * bB);  // from the tag on
        // tag table (decoder.nim:42-109; validity: the index pass): [0:7) length of the forms without length
        // bytes, [8:11) a copy1's offset bits 8..10, bit 11 literal, bit 12 literal with length bytes, bit 13 copy4
        uint32_t tA, tB;
        if (kLutInLds) {
          tA = hasA ? s_lut[dA & 0xffu] : 0u;
          tB = hasB ? s_lut[dB & 0xffu] : 0u;
        } else {
          tA = hasA ? prm.tag_lut[dA & 0xffu] : 0u;
          tB = hasB ? prm.tag_lut[dB & 0xffu] : 0u;
        }
        uint32_t LA = tA & 0x7fu, LB = tB & 0x7fu;
        // offset: 1, 2 or 3 bytes behind the tag (none for a literal), a copy1's high bits from the table
        uint32_t offA = __builtin_amdgcn_ubfe(dA, 8, (dA & 3u) * 8) | (tA & 0x700u);
        uint32_t offB = __builtin_amdgcn_ubfe(dB, 8, (dB & 3u) * 8) | (tB & 0x700u);
        const bool litA = (tA & 0x800u) != 0, litB = (tB & 0x800u) != 0;
        const bool bigA = (tA & 0x1000u) != 0, bigB = (tB & 0x1000u) != 0;
        uint32_t hdr_big = 1;
        if (__builtin_expect(ballot(((tA | tB) & 0x3000u) != 0) != 0, 0)) {
          // literals with length bytes (decoder.nim:54-75); a copy4's fourth offset byte (an offset of 2^24 and
          // more is beyond any block: the element list holds 16 bits, so it is made 0 = invalid here)
          const uint32_t hA = __funnelshift_r(w1, w2, 8 * bA), hB = __funnelshift_r(w1, w2, 8 * bB);  // bytes 4.. behind the tag
          const uint32_t tg = bigB ? dB : dA;
          const uint32_t lenlen = ((tg >> 2) & 63u) - 59;  // 1..4 where it applies
          const uint32_t b14 = ((bigB ? dB : dA) >> 8) | ((bigB ? hB : hA) << 24);
          const uint32_t m = 0xffffffffu >> (32 - 8 * ((lenlen & 7) ? (lenlen & 7) : 4));
          const uint32_t Lb = (b14 & m) + 1;
          LA = bigA ? Lb : LA;
          LB = bigB ? Lb : LB;
          hdr_big = 1 + (lenlen & 7);
          if ((tA & 0x2000u) && ((hA & 0xffu) || (offA >> 16))) offA = 0;
          if ((tB & 0x2000u) && ((hB & 0xffu) || (offB >> 16))) offB = 0;
        }
        // ---- output positions: the region's first one + the lengths in front of me in my region (4 lanes) ----
        uint32_t dstA, dstB;
        {
          const uint32_t T = LA + LB;
          const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T, 0x90 /* quad_perm:[0,0,1,2] */, 0xf, 0xf, false);
          const uint32_t x1 = T + (s1 & fe_q1);
          const uint32_t s2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x1, 0x44 /* quad_perm:[0,1,0,1] */, 0xf, 0xf, false);
          dstA = rdst + x1 + (s2 & fe_q2) - T;  // (x1 + s2 & q2: inclusive over the quad)
          dstB = dstA + LA;
        }
        // ---- list slots: a prefix sum of the start counts ----
        uint32_t slotA;
        {
          const uint32_t cnt = __builtin_popcount(nib);
          uint32_t c = cnt;
          c += dpp_mov0<0x111>(c);
          c += dpp_mov0<0x112>(c);
          c += dpp_mov0<0x114>(c);
          c += dpp_mov0<0x118>(c);
          c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x142, 0xa, 0xf, false);
          c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x143, 0xc, 0xf, false);
          slotA = base + c - cnt;
        }
        const uint32_t slotB = slotA + 1;
        // ---- list entries; the slot at each 256-byte output boundary an element covers ----
        {
          const bool badA = hasA && !litA && offA - 1 >= dstA, badB = hasB && !litB && offB - 1 >= dstB;  // decoder.nim:112
          bad = bad || badA || badB;
          // (0xffff in the list means "literal", and 65 535 is a legal copy offset -- of ONE element: a copy at output
          // position 65 535 of a full block, decoder.nim:112.  A unit that holds it goes to the one-pass kernel.)
          far_off = far_off || (hasA && !litA && !badA && offA == 0xffffu) || (hasB && !litB && !badB && offB == 0xffffu);
          uint32_t* const sink32 = reinterpret_cast<uint32_t*>(sink16);
          *(hasA ? el + slotA : sink32) = ((litA || badA) ? 0xffffu : offA) | (dstA << 16);  // (list value of a literal: 0xffff)
          *(hasB ? el + slotB : sink32) = ((litB || badB) ? 0xffffu : offB) | (dstB << 16);
          // (an element covers a boundary when its last byte and the byte in front of it lie in different groups)
          const uint32_t eA = dstA + LA - 1, eB = dstB + LB - 1;
          *((hasA && ((eA ^ (dstA - 1)) >= kGroup)) ? s_gidx + (eA / kGroup) : sink16) = (uint16_t)slotA;
          *((hasB && ((eB ^ (dstB - 1)) >= kGroup)) ? s_gidx + (eB / kGroup) : sink16) = (uint16_t)slotB;
        }
        // ---- literal payloads ----
        const bool shortA = litA && !bigA, shortB = litB && !bigB;
        // what a lane hands on: its last element, if that is a short literal: (payload end, counted from the wave's
        // first byte) << 16 | (output position - that count of the payload's first byte) & 0xffff; else 0
        const uint32_t info_last = hasB ? (shortB ? ((xr + bB + 1 + LB) << 16) | ((dstB - (xr + bB + 1)) & 0xffffu) : 0u)
                                        : (shortA ? ((xr + bA + 1 + LA) << 16) | ((dstA - (xr + bA + 1)) & 0xffffu) : 0u);
        uint32_t infoC;
        uint32_t last_lit;  // (uniform) 1 + the last lane whose last element is a short literal
        {
          uint32_t k = info_last ? lane + 1 : 0u;
          auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
          k = mx(k, dpp_mov0<0x111>(k));
          k = mx(k, dpp_mov0<0x112>(k));
          k = mx(k, dpp_mov0<0x114>(k));
          k = mx(k, dpp_mov0<0x118>(k));
          k = mx(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x142, 0xa, 0xf, false));
          k = mx(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x143, 0xc, 0xf, false));
          last_lit = readlane(k, 63);
          const uint32_t before = dpp_mov0<0x138>(k);  // wave_shr:1: the inclusive maximum of the lane before me
          const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(before * 4 - 4), (int)info_last);
          infoC = before ? got : 0u;
        }
        if (!(SNAPPY_DBG(prm) & 1)) {
          auto wad = [&](uint32_t a) -> uint32_t { return RING ? (a & (WIN - 1)) : (a & 0xffffu); };
          // C: bytes 0 .. min(bA, what is left of its payload) of my quad
          {
            const uint32_t pe = infoC >> 16;
            const uint32_t left = pe > xr ? pe - xr : 0u;
            const uint32_t nC = left < bA ? left : bA;
            const uint32_t aC = xr + infoC;
            s_out[0 < nC ? wad(aC) : sink] = (uint8_t)w0;
            s_out[1 < nC ? wad(aC + 1) : sink + 1] = (uint8_t)(w0 >> 8);
            s_out[2 < nC ? wad(aC + 2) : sink + 2] = (uint8_t)(w0 >> 16);
            s_out[3 < nC ? wad(aC + 3) : sink + 3] = (uint8_t)(w0 >> 24);
          }
          // A: its payload bytes in my quad: behind its tag, in front of B's tag (or the quad's end)
          {
            const uint32_t room = bB - bA - 1;  // (bA <= 3 here; bB = 4 without a B)
            const uint32_t nA = shortA ? (LA < room ? LA : room) : 0u;
            const uint32_t v = w0 >> (8 * bA + 8);
            s_out[0 < nA ? wad(dstA) : sink] = (uint8_t)v;
            s_out[1 < nA ? wad(dstA + 1) : sink + 1] = (uint8_t)(v >> 8);
            s_out[2 < nA ? wad(dstA + 2) : sink + 2] = (uint8_t)(v >> 16);
          }
          // B: one byte, when it starts at byte 2
          s_out[(shortB && bB == 2) ? wad(dstB) : sink + 3] = (uint8_t)(w0 >> 24);
          // the wave's last short literal may run on behind the wave's 256 bytes (by less than 60): one byte per lane
          const uint32_t tl = last_lit ? readlane(info_last, last_lit - 1) : 0u;
          if ((tl >> 16) > 256) {
            const uint32_t xt = 256 + lane;
            const uint8_t v = s_ring[(xw + xt + shift) & (kD2Ring - 1)];
            s_out[xt < (tl >> 16) ? wad(xt + tl) : sink] = v;
          }
        }
        if (__builtin_expect(ballot((hasA && bigA) || (hasB && bigB)) != 0, 0)) {
          const bool bb = hasB && bigB;
          big = bb || (hasA && bigA);
          big_dst = bb ? dstB : dstA;
          big_len = bb ? LB : LA;
          big_src = xw + xr + (bb ? bB : bA) + hdr_big;
          big_slot = bb ? slotB : slotA;
        }
      }


      // long literals: whole wave, straight from HBM (at most one per lane)
      uint64_t bigs = ballot(big);
      while (bigs) {
        const uint32_t e = ctz64(bigs);
        bigs &= bigs - 1;
        const uint32_t eL = readlane(big_len, e);
        const uint32_t ed = readlane(big_dst, e);
        const uint32_t es = readlane(big_src, e);
        const uint32_t eslot = readlane(big_slot, e);
        {
          // steps s+2 .. to-1 lie inside this literal (s+1 still has to resolve this step's list)
          const uint32_t to = (es + eL) / kChunk;
          if (to > s + 2 && lane == 0) atomicMax(&s_skip, (to << 16) | (s + 2));
        }
        // every 256-byte boundary the literal covers maps to its slot
        if (lane == 0) s_big[buf] = s + 1;
        for (uint32_t m = (ed + kGroup - 1) / kGroup + lane; m * kGroup < ed + eL; m += 64)
          s_gidx[m & (kMaxBlockLen / kGroup - 1)] = (uint16_t)(eslot | 0x8000u);  // flag: inside a long literal
        // destination-aligned: up to 15 head bytes one per lane, then 16-byte pieces whose LDS store
        // is aligned (the unaligned side is the load from HBM, where it costs next to nothing; an
        // unaligned LDS dword store costs 10-20 aligned ones), eight pieces per lane in flight
        const uint32_t head = (16 - (ed & 15)) & 15;
        const uint32_t hd = head < eL ? head : eL;
        if (lane < hd) s_out[wa(ed + lane)] = in0[es + lane];
        const uint32_t body = (eL - hd) & ~15u;
        const uint8_t* src = in0 + es + hd;
        const uint32_t dst0 = ed + hd;  // 16-byte aligned when body > 0 (a piece never straddles the ring's end)
        for (uint32_t i = lane * 16; i < body; i += 8 * 1024) {
          // (loads from clamped addresses instead of guarded ones: no private array, no spills; a
          // clamped piece rewrites piece i with its own bytes)
          uint32_t ix[8];
          uint4 v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) ix[j] = i + j * 1024 < body ? i + j * 1024 : i;
#pragma unroll
          for (int j = 0; j < 8; j++) __builtin_memcpy(&v[j], src + ix[j], 16);
#pragma unroll
          for (int j = 0; j < 8; j++) *reinterpret_cast<uint4*>(s_out + wa(dst0 + ix[j])) = v[j];
        }
        if (hd + body + lane < eL) s_out[wa(ed + hd + body + lane)] = in0[es + hd + body + lane];
      }
      }  // trips
      if (ballot(bad) && lane == 0) atomicOr(&s_err, 1u);
      if (__builtin_expect(ballot(far_off) != 0, 0) && lane == 0) atomicOr(&s_err, 4u);
      __builtin_amdgcn_s_setprio(0);
      acc_b += 1;
    }
    tt_fe += now() - tq0;
    if (s >= 1 && wave >= kPoolFirst && !(SNAPPY_DBG(prm) & 2)) {
      // =================================== resolvers ===============================================
      const uint32_t buf = (s - 1) & 1;
      const uint32_t cb = readfirst(s_sbase[s - 1]), cn = readfirst(s_sbase[s]);
      const uint32_t count = readfirst(s_cnt[buf]);
      const uint16_t* const el16 = reinterpret_cast<const uint16_t*>(s_el[buf]);
      auto o16 = [&](uint32_t e) -> uint32_t { return el16[2 * e]; };      // copy offset, 0xffff = literal
      auto d16 = [&](uint32_t e) -> uint32_t { return el16[2 * e + 1]; };  // first output byte
      uint16_t* const r16 = s_r16[wave - kPoolFirst];
      const uint32_t gfirst = cb & ~(kGroup - 1);
      // a group in the middle of one long literal holds no copy (the front end flags the
      // boundaries such a literal covers); nobody works on it, the group before it publishes it
      auto is_skip = [&](uint32_t gg) -> bool {  // gg > cb (per lane)
        const uint32_t a = s_gidx[gg / kGroup];
        const uint32_t b2 = s_gidx[gg / kGroup + 1];
        return (a & 0x8000u) && gg + kGroup < cn && a == b2;
      };
      uint32_t front = cb;  // what I know of s_front
      constexpr uint32_t B = kGroup / 64;  // bytes per lane: one aligned dword
      static_assert(B == 4, "one dword per lane");
      const uint32_t* const el32 = s_el[buf];
      // (a group's first words -- who covers its first byte, whether the group behind it is inside a long literal --
      // are requested while the group before it is worked on: one round trip less on the wave's path)
      uint32_t g = gfirst + (wave - kPoolFirst) * kGroup;
      const bool has_big = readfirst(s_big[buf]) == s;  // (this list's step is s - 1)
      uint32_t ge_raw = s_gidx[g / kGroup];
      uint32_t gn_raw = has_big ? s_gidx[g / kGroup + 1] : 0u;
      // (an empty step has nothing to resolve -- and after a fast-forward its list is not even its own)
      for (; g < cn && cb < cn; g += kPool * kGroup) {
        const unsigned long long tg0 = now();
        const uint32_t ge = g > cb ? readfirst(ge_raw) : 0;
        const uint32_t gnext = readfirst(gn_raw);
        {
          const uint32_t gm = (g + kPool * kGroup) / kGroup;
          ge_raw = s_gidx[gm];
          if (has_big) gn_raw = s_gidx[gm + 1];
        }
        if ((ge & 0x8000u) && readfirst(is_skip(g) ? 1u : 0u)) continue;  // (flag first: one read for most groups)
        acc_c++;
        const uint32_t p = g + B * lane;
        // (what I saw of the frontier at my last turn; it only grows)
        if (front > (g > cb ? g : cb)) continue;  // a run extension (below) has covered my group
        // E0: the element that covers byte g (the step's first element where the step starts inside the group or
        // at its first byte).  From E0 on, every element that starts below the group's end writes its list value
        // -- copy offset, or 0xffff for a literal -- at its first byte's slot of the scratch, E0 at slot 0; "what
        // covers byte x" is then the last value at or below x: a prefix maximum over (slot + 1) << 16 | value.
        const uint32_t E0 = g > cb ? (ge & 0x7fffu) : 0u;
        uint32_t* const r32 = reinterpret_cast<uint32_t*>(r16 + B * lane);  // my B slots
        r32[0] = 0;
        r32[1] = 0;
        cbar();
        uint32_t nbelow = 0;  // elements from E0 on that start below the group's end
        for (uint32_t e = E0 + lane;; e += 64) {
          const uint32_t v = e < count ? el32[e] : 0xffffffffu;
          const uint32_t d = v >> 16;
          const bool below = e < count && d < g + kGroup;
          const uint32_t slot = d > g ? d - g : 0;  // (E0 may start below g)
          *(below ? r16 + slot : sink16) = (uint16_t)v;
          const uint64_t mb = ballot(below);
          nbelow += (uint32_t)__builtin_popcountll(mb);
          if (mb != ~0ull) break;
        }
        cbar();
        uint32_t kj[B];  // (slot + 1) << 16 | value of the start at each of my bytes, 0 where none is
        {
          const uint32_t r0 = r32[0], r1 = r32[1];
          cbar();
          const uint32_t pos1 = (B * lane + 1) << 16;
          kj[0] = (r0 & 0xffffu) ? (r0 & 0xffffu) | pos1 : 0u;
          kj[1] = (r0 >> 16) ? (r0 >> 16) | (pos1 + (1u << 16)) : 0u;
          kj[2] = (r1 & 0xffffu) ? (r1 & 0xffffu) | (pos1 + (2u << 16)) : 0u;
          kj[3] = (r1 >> 16) ? (r1 >> 16) | (pos1 + (3u << 16)) : 0u;
        }
#pragma unroll
        for (uint32_t j = 1; j < B; j++) kj[j] = kj[j] > kj[j - 1] ? kj[j] : kj[j - 1];
        uint32_t kmax;
        const uint32_t before = wave_excl_scan_max(kj[B - 1], lane, &kmax);
        const uint32_t tot = nbelow;  // (element starts in the group, and E0)
        // my bytes that belong to this step: [lo, hi) of 0..B
        uint32_t rmask = (1u << B) - 1;
        if (g < cb || g + kGroup > cn) {  // only the step's first and last group are partial
          const uint32_t lo = p >= cb ? 0 : (cb - p < B ? cb - p : B);
          const uint32_t hi = p + B <= cn ? B : (cn > p ? cn - p : 0);
          rmask = ((1u << hi) - 1) & ~((1u << lo) - 1);
        }
        uint32_t sp[B], offj[B];
        bool cp[B];
        bool anyc = false, off_differs = false;
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
          const bool in = (rmask >> j) & 1;
          const uint32_t off = (kj[j] > before ? kj[j] : before) & 0xffffu;
          cp[j] = in && off != 0xffffu;
          offj[j] = cp[j] ? off : 0;
          sp[j] = p + j - offj[j];
          anyc = anyc || cp[j];
        }
        // A group that is one run of copies with one offset (how the encoder splits a long match,
        // encoder.nim:97-112): if the run goes on, I will also do the whole groups that follow
        // inside it, one group per trip, once it is my turn.  run_end = first byte after them.
        // (only groups of few, long elements are examined: tot = element starts in the group)
        uint32_t run_off = 0, run_end = 0;
        if (__builtin_expect(tot <= kGroup / 32 + 1, 0) && (asm_nop(), g >= cb && g + kGroup <= cn) && (run_off = readfirst(offj[0])) != 0) {
#pragma unroll
          for (uint32_t j = 0; j < B; j++) off_differs = off_differs || offj[j] != run_off;
          if (ballot(off_differs) == 0) {
            uint32_t R = cn;
            for (uint32_t e = E0 + tot + lane;; e += 64) {  // elements after those of my group
              const uint32_t oo = e < count ? o16(e) : 0;
              const uint64_t mm = ballot(oo != run_off);
              if (mm) {
                const uint32_t ef = readfirst(e) + ctz64(mm);
                if (ef < count) R = readfirst(d16(ef));
                break;
              }
            }
            run_end = R & ~(kGroup - 1);
          }
        }
        const bool work = ballot(anyc) != 0;  // (a group of literals only has nothing to do)
        // ---- sources inside my own group: follow them to a final byte ----------------------------
        bool dep[B];
        bool anydep = false;
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
          dep[j] = cp[j] && sp[j] >= g;
          anydep = anydep || dep[j];
        }
        if (ballot(anydep)) {
          for (uint32_t it = 0; it < 11; it++) {
            acc_d++;
            // every copy byte publishes its pointer, every other byte "I am final" (0xffff, never a
            // source position); a byte takes over the pointer of the copy byte it points to: the
            // chain length halves per round
            cbar();
#pragma unroll
            for (uint32_t k = 0; k < B / 2; k++)
              r32[k] = (cp[2 * k] ? sp[2 * k] : 0xffffu) | ((cp[2 * k + 1] ? sp[2 * k + 1] : 0xffffu) << 16);
            cbar();
            anydep = false;
            // (the four reads go out together: left to itself the compiler reads, waits and evaluates byte by byte --
            // four LDS round trips a round)
            uint32_t tq[B];
#pragma unroll
            for (uint32_t j = 0; j < B; j++) tq[j] = r16[dep[j] ? sp[j] - g : 0];
            asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]));
#pragma unroll
            for (uint32_t j = 0; j < B; j++) {
              const uint32_t t = tq[j];
              const bool fin = t == 0xffffu;  // my source is a final byte of the group
              sp[j] = (dep[j] && !fin) ? t : sp[j];
              dep[j] = dep[j] && !fin && t >= g;
              anydep = anydep || dep[j];
            }
            cbar();
            if (!ballot(anydep)) break;
          }
        }
        cbar();
        // (ring) sources that have left the window: from HBM, where they were written at least a step ago
        // (device-scope loads: past this CU's vector cache, which may hold the line's older state).
        // (Requested ahead of the pointer rounds, so that the trip to the L2 runs beside them: measured slower --
        // four more registers alive across the rounds, and the compiler then waits for each of a round's four
        // LDS reads on its own.)
        uint32_t far_m = 0, far_v = 0;
        if (RING) {
          bool fj[B];
          bool anyfar = false;
#pragma unroll
          for (uint32_t j = 0; j < B; j++) {
            fj[j] = cp[j] && sp[j] < ring_lo && !(SNAPPY_DBG(prm) & 16);  // (DEBUG 16: no read-backs -- wrong bytes, timing only)
            anyfar = anyfar || fj[j];
          }
          if (__builtin_expect(ballot(anyfar) != 0, 0)) {
            // (four loads in flight, every lane: the addresses are selected, not the loads -- the empty asm keeps
            // the compiler from turning the selects into branches around the loads, each with its own wait)
            uint32_t ix[B];
#pragma unroll
            for (uint32_t j = 0; j < B; j++) ix[j] = fj[j] ? sp[j] : 0;
            asm volatile("" : "+v"(ix[0]), "+v"(ix[1]), "+v"(ix[2]), "+v"(ix[3]));
#pragma unroll
            for (uint32_t j = 0; j < B; j++) {
              const uint32_t b = __hip_atomic_load(gout + ix[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              far_m |= fj[j] ? 0xffu << (8 * j) : 0;
              far_v |= fj[j] ? b << (8 * j) : 0;
              sp[j] = fj[j] ? p + j : sp[j];  // (its window read in the turn: anything inside the window)
            }
          }
          // A run is extended inside the ring: its sources must still be there.  (Otherwise group by group,
          // like any other.)
          if (__builtin_expect(run_end != 0, 0)) {  // (a branch, not a dozen scalar selects in every group: the empty asm pins it)
            asm volatile("");
            if (run_end > g + kGroup) {
              const uint32_t lo = g >= run_off + 1024 ? g - run_off - 1024 : 0;  // (below every source it reads)
              if (lo < ring_lo) run_end = 0;
            }
          }
        }
        // how many of the groups after mine are skipped: I publish them with mine
        uint32_t nskip = 0;
        // (only looked into when the next group starts inside a long literal)
        while (g + kGroup < cn && (gnext & 0x8000u)) {
          const uint32_t gg = g + kGroup * (1 + nskip + lane);
          const uint64_t sk = ballot(gg < cn && is_skip(gg));
          const uint32_t c = (~sk) ? ctz64(~sk) : 64;
          nskip += c;
          if (c < 64) break;
        }
        cbar();
        // ---- everything the turn needs is worked out before the wait: the four source addresses, where
        // the dword goes, whether any lane has to store bytewise -- between seeing the frontier and
        // publishing there are the gather, three instructions of assembly and two stores ----
        static_assert(B == 4, "one dword per lane");
        // a dword that is entirely this step's: its other bytes are final (literals) and may be
        // rewritten with their own value; the step's first and last dwords are stored bytewise
        // (the front end may be writing the next step's literals into the same dword right now)
        const bool full = (rmask & 15) == 15;
        const bool any4 = cp[0] || cp[1] || cp[2] || cp[3];
        const bool any_partial = ballot(!full && any4) != 0;
        lds_u8* const wo = (lds_u8*)s_out;
        // (complete LDS addresses: the window does not start at LDS address 0)
        uint32_t a0 = (uint32_t)(uintptr_t)(wo + wa(sp[0])), a1 = (uint32_t)(uintptr_t)(wo + wa(sp[1]));
        uint32_t a2 = (uint32_t)(uintptr_t)(wo + wa(sp[2])), a3 = (uint32_t)(uintptr_t)(wo + wa(sp[3]));
        uint32_t ad = (uint32_t)(uintptr_t)(wo + ((full && any4) ? wa(p) : sink));
        uint32_t front_after = g + kGroup * (1 + nskip);
        front_after = front_after < cn ? front_after : cn;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(ad), "+s"(front_after));
        if (RING) asm volatile("" : "+v"(far_m), "+v"(far_v));  // (the loads have landed before the wait)
        // ---- my turn: every group below mine has published, i.e. everything below g is final ------
        const unsigned long long tg1 = now();
        tt_prep += tg1 - tg0;
        const uint32_t expect = g > cb ? g : cb;
        // (from here to the publish this wave is, or is about to be, on the step's critical path)
        __builtin_amdgcn_s_setprio(3);
#ifndef D2_LOOSE_POLL
        // (the wait is written out.  Round 3: read, wait, readfirstlane, compare, branch + a bounded counter -- eight
        // instructions a look, six of them scalar, where the compiler's loop took eighteen.  Round 6: the comparison is a
        // vector compare into VCC -- every lane reads the same word -- and the counter is stepped once per four looks:
        // four instructions a look, two of them scalar.  A turn is looked for ~15 times, the scalar unit is shared by the
        // CU's 32 waves, and the looks were a third of the scalar instructions of a group.)
#ifdef D2_INJECT_GIVE_UP  // (tests/test_gpu_faults.py: every k-th turn that has to be waited for is given up on at once)
        const uint32_t looks = (front < expect && (++d2_inject % (D2_INJECT_GIVE_UP)) == 0) ? 1 : 256;
#else
        constexpr uint32_t looks = 256;
#endif
        for (uint32_t spin = looks == 256 ? 0 : 400001; front < expect; spin += 1024) {
          uint32_t left = looks, fv;
          const uint32_t fa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_front;
#define D2_LOOK                       \
  "ds_read_b32 %[fv], %[fa]\n"        \
  "s_waitcnt lgkmcnt(0)\n"            \
  "v_cmp_le_u32_e32 vcc, %[ex], %[fv]\n" \
  "s_cbranch_vccnz 2f\n"
          asm volatile(
              "1:\n" D2_LOOK D2_LOOK D2_LOOK D2_LOOK
              "s_sub_u32 %[left], %[left], 1\n"
              "s_cmp_lg_u32 %[left], 0\n"
              "s_cbranch_scc1 1b\n"
              "2:\n"
              "v_readfirstlane_b32 %[fr], %[fv]\n"
              : [fr] "+s"(front), [fv] "=&v"(fv), [left] "+s"(left)
              : [fa] "v"(fa), [ex] "s"(expect)
              : "scc", "vcc", "memory");
#undef D2_LOOK
          if (front >= expect) break;
          if (spin > 400000 || s_err != 0) {
            // cannot happen on a consistent index; never hang the GPU
            if (lane == 0) {
              atomicOr(&s_err, 4u);
              if (spin > 400000 && prm.timeouts) atomicAdd(prm.timeouts, 1u);
            }
            break;
          }
        }
#else
        for (uint32_t spin = 0; front < expect; spin++) {
          if ((spin & 1023) == 1023 && (spin > 400000 || s_err != 0)) {
            // cannot happen on a consistent index; never hang the GPU
            if (lane == 0) {
              atomicOr(&s_err, 4u);
              if (spin > 400000 && prm.timeouts) atomicAdd(prm.timeouts, 1u);
            }
            break;
          }
          front = readfirst(__hip_atomic_load(&s_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          cbar();
        }
#endif
        const unsigned long long tg2 = now();
        tt_wait += tg2 - tg1;
        if (front > expect) {  // covered by a run extension meanwhile
          __builtin_amdgcn_s_setprio(0);
          continue;
        }
        if (work) {
          // every source is final now: gather (most groups would have to fetch again after an early
          // gather anyway)
          auto at = [](uint32_t a) { return (const lds_u8*)(uintptr_t)a; };
          uint32_t v = (uint32_t)*at(a0) | ((uint32_t)*at(a1) << 8) | ((uint32_t)*at(a2) << 16) | ((uint32_t)*at(a3) << 24);
          if (RING) v = (v & ~far_m) | far_v;
          *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)ad = v;
          if (any_partial) {
#pragma unroll
            for (uint32_t j = 0; j < B; j++)
              wo[(!full && cp[j]) ? wa(p + j) : sink + j] = (uint8_t)(v >> (8 * j));
          }
        }
        if (__builtin_expect(run_end > g + kGroup, 0)) {
          // ---- run extension: out[x] = out[x - W]; W >= the group size, so a trip only reads what
          // earlier trips (or earlier groups) wrote ----
          acc_a++;
          static_assert(B == 4, "extend_run copies one dword per lane in its first loop");
          cbar();
          extend_run<RING ? WIN - 1 : 0xffffffffu>((lds_u8*)s_out, g, run_end, run_off, lane);
          cbar();
          nskip = 0;
          for (;;) {
            const uint32_t gg = run_end + kGroup * (nskip + lane);
            const uint64_t sk = ballot(gg < cn && gg > cb && is_skip(gg));
            const uint32_t c = (~sk) ? ctz64(~sk) : 64;
            nskip += c;
            if (c < 64) break;
          }
          front = run_end + kGroup * nskip;
          front = front < cn ? front : cn;
        } else {
          front = front_after;
        }
        cbar();
        if (lane == 0) __hip_atomic_store(&s_front, front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_s_setprio(0);
        tt_turn += now() - tg2;
      }
    }
    const unsigned long long tm1 = SNAPPY_STATS(prm) ? __builtin_amdgcn_s_memtime() : 0;
    // Workgroup barrier for LDS traffic only: __syncthreads() would also drain vmcnt, i.e. wait
    // for the global prefetches that are meant to stay in flight across the barrier.
    // (ring: the flush stores of this step must have reached the L2 before the step after next reads them)
    if (RING) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (SNAPPY_STATS(prm)) {
      tm_work += tm1 - tm0;
      tm_bar += __builtin_amdgcn_s_memtime() - tm1;
    }
  }
  if (SNAPPY_STATS(prm) && lane == 0 && (wave == 0 || wave == 5)) {  // DEBUG
    unsigned long long* st = prm.stats + (wave == 0 ? 0 : 12);
    atomicAdd(&st[6], tt_flush);
    atomicAdd(&st[7], tt_fe);
    atomicAdd(&st[8], tt_prep);
    atomicAdd(&st[9], tt_wait);
    atomicAdd(&st[10], tt_turn);
    atomicAdd(&st[0], (unsigned long long)acc_a);
    atomicAdd(&st[1], (unsigned long long)acc_b);
    atomicAdd(&st[2], (unsigned long long)acc_c);
    atomicAdd(&st[3], (unsigned long long)acc_d);
    atomicAdd(&st[4], tm_work);
    atomicAdd(&st[5], tm_bar);
  }

  // ---- flush ------------------------------------------------------------------------------------
  if (s_err) {
    if (tid == 0) {
      prm.status[u] = (s_err & 1) ? kInvalidInput : kNeedsOnePass;
      if (s_err & 1) prm.out_len[u] = 0;  // (a failed unit reports no bytes, like the one-pass kernel)
    }
    return;
  }
  if (RING) {
    if (passed_on) {  // (uniform: decided from s_sbase; what was flushed so far is rewritten by the other instantiation)
      if (tid == 0) prm.status[u] = kNeedsWindow;
      return;
    }
    for (uint32_t i = flushed + tid * 16; i < total; i += kD2Threads * 16) {
      if (i + 16 <= total) {
        *reinterpret_cast<uint4*>(gout + i) = *reinterpret_cast<const uint4*>(s_out + wa(i));
      } else {
        for (uint32_t k = i; k < total; k++) gout[k] = s_out[wa(k)];
      }
    }
    if (do_crc) {  // the rows the loop has not seen, then the columns together (crc_pack_kernels.h)
      crc_rows_to(total);
      uint32_t part = (tid >= 256 && tid < 512) ? crc_reg : 0;
      for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
      if (lane == 0 && wave >= 4 && wave < 8) {
        atomicXor(&s_crc_acc, part);
        cbar();
        if (atomicAdd(&s_crc_cnt, 1u) == 3) {  // the last of the four waves: every part is in
          const uint32_t crc = ~atomicOr(&s_crc_acc, 0u);         // crc32c.c:761
          prm.crc[u] = ((crc >> 15) | (crc << 17)) + kMaskDelta;    // crc32c.c:762
          prm.crc_done[u] = 1;
        }
      }
    }
    return;  // (without RCRC no CRC from a ring: crc_done stays 0)
  }
  if (SNAPPY_DBG(prm) & 8) return;
  if (((uintptr_t)gout & 15) == 0) {
    for (uint32_t i = tid * 16; i < total; i += kD2Threads * 16) {
      if (i + 16 <= total) {
        *reinterpret_cast<uint4*>(gout + i) = *reinterpret_cast<const uint4*>(s_out + i);
      } else {
        for (uint32_t k = i; k < total; k++) gout[k] = s_out[k];
      }
    }
  } else {
    for (uint32_t i = tid * 4; i < total; i += kD2Threads * 4) {
      if (i + 4 <= total) {
        st32u(gout + i, *reinterpret_cast<const uint32_t*>(s_out + i));
      } else {
        for (uint32_t k = i; k < total; k++) gout[k] = s_out[k];
      }
    }
  }
  if (prm.crc == nullptr) return;

  // ---- masked CRC32C of s_out[0 .. total) (maskedCrc, snappy/codec.nim:71-75), from the window ----
  // The column scheme of crc32c_units_kernel (crc_pack_kernels.h) with the message in LDS, run
  // twice side by side to halve the chain of dependent rows: waves 4-7 take the last 32 KiB,
  // waves 0-3 what lies in front of them, advanced afterwards over those 32 KiB with one
  // multiplication by x^(8 * 32768) mod P.
  uint32_t* const s_ctab = reinterpret_cast<uint32_t*>(s_ring);  // (the ring is free now)
  for (uint32_t i = tid; i < 1024; i += kD2Threads) s_ctab[i] = prm.crc_tab[i];
  __syncthreads();
  constexpr uint32_t kHalf = 32768;
  const uint32_t split = total > kHalf + 4 ? total - kHalf : 0;  // (a front part is never shorter than 4 bytes)
  const bool front = wave < 4;
  const uint32_t base = front ? 0 : split;
  const uint32_t n_part = front ? split : total - split;
  const uint32_t t = tid & 255;
  uint32_t part = 0;  // my column's share of the CRC register
  if (total < 4) {
    if (tid == 0) {
      uint32_t reg = 0xffffffffu;
      for (uint32_t i = 0; i < total; i++) {
        reg ^= s_out[i];
        for (int k = 0; k < 8; k++) reg = (reg >> 1) ^ ((reg & 1) ? kCrcPoly : 0);
      }
      part = reg;
    }
  } else if (n_part) {
    const uint32_t rows = (n_part + 1023) / 1024;
    const int32_t pad = (int32_t)(rows * 1024 - n_part);  // virtual leading zero bytes
    const uint32_t* const o32 = reinterpret_cast<const uint32_t*>(s_out);
    const uint32_t sh8 = ((base + n_part) & 3) * 8;        // byte phase of my dwords (0 for whole blocks)
    uint32_t sreg = 0;
    for (uint32_t r = 0; r < rows; r++) {
      const int32_t pos = (int32_t)(r * 1024 + 4 * t) - pad;
      uint32_t w;
      if (pos >= 4) {  // aligned dwords + funnel shift (an unaligned LDS dword read costs 10-20 aligned ones)
        const uint32_t a = (base + (uint32_t)pos) >> 2;
        w = __funnelshift_r(o32[a], o32[a + 1], sh8);
      } else {  // touches the part's start: virtual zero padding; the message's start: the 0xffffffff init
        w = 0;
        for (int k = 0; k < 4; k++) {
          const int32_t j = pos + k;
          if (j >= 0) w |= (uint32_t)(s_out[base + j] ^ ((base == 0 && j < 4) ? 0xff : 0)) << (8 * k);
        }
      }
      const uint32_t x = sreg ^ w;
      if (r + 1 < rows) {
        sreg = s_ctab[x & 0xff] ^ s_ctab[256 + ((x >> 8) & 0xff)] ^ s_ctab[512 + ((x >> 16) & 0xff)] ^
               s_ctab[768 + (x >> 24)];
      } else {
        sreg = gf2_mulmod(prm.crc_col[t], x);  // the 4*(256-t) bytes from here to the part's end
      }
    }
    part = sreg;
  }
  for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
  if (front && split) part = gf2_mulmod(prm.crc_k32k, part);  // ... and the 32 KiB behind the front part
  if (lane == 0) {
    atomicXor(&s_crc_acc, part);
    cbar();
    if (atomicAdd(&s_crc_cnt, 1u) == kD2Threads / 64 - 1) {  // the last wave: every part is in
      const uint32_t crc = ~atomicOr(&s_crc_acc, 0u);             // crc32c.c:761
      prm.crc[u] = ((crc >> 15) | (crc << 17)) + kMaskDelta;        // crc32c.c:762
      prm.crc_done[u] = 1;
    }
  }
}

// The units the ring-window launch passed on (status kNeedsWindow), in launch order (wave by wave): list[-2] =
// how many.  One thread per unit.
__global__ __launch_bounds__(256) void passed_on_list_kernel(const uint32_t* status, const uint32_t* order, uint64_t n_units,
                                                             uint32_t* list) {
  const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
  const uint32_t u = i < n_units ? (order ? order[i] : (uint32_t)i) : 0;
  const bool mine = i < n_units && status[u] == kNeedsWindow;
  const uint64_t m = ballot(mine);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(list - 2, (uint32_t)__builtin_popcountll(m));
  base = readfirst(base);
  if (mine) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1))] = u;
}

// Region count per unit for the index (bounded: the index pass stops once a unit has produced
// more than 64 KiB, which takes at most 6 stream bytes per output byte).
constexpr uint32_t kMaxRegionsPerUnit = kMaxFastIn / kSub + 8;  // index entries (16 bytes each)

__global__ void region_counts_kernel(const uint32_t* in_len, uint64_t n_units, uint32_t* counts) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n_units) return;
  const uint64_t c = ((uint64_t)in_len[i] + kSub - 1) / kSub + 2;
  counts[i] = c < kMaxRegionsPerUnit ? (uint32_t)c : kMaxRegionsPerUnit;
}

}  // namespace snappy_hip
