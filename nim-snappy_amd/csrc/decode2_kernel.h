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
constexpr uint32_t kD2Ring = 4096;
constexpr uint32_t kElemCap = 1024;  // elements per 2 KiB step: the format's maximum (2 bytes each)
constexpr uint32_t kGroup = 256;     // output bytes one resolver wave handles per pass (4 per lane)
// The output window: the whole block (kMaxBlockLen), or a RING of the last kRingWin bytes (see the kernel).
constexpr uint32_t kRingWin = 32768;
#ifndef D2_RING
#define D2_RING 1  // (0: experiments, the whole-block instantiation only)
#endif
constexpr bool kD2RingFirst = D2_RING != 0;
#ifndef D2_RING_CATCHUPS
#define D2_RING_CATCHUPS 2
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
  int dbg;  // timing experiments: 1 no literal payloads, 2 no resolver, 4 no walk, 8 no flush
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
};

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
// -- so that THREE workgroups fit a CU (the kernel is bound by a block's chain of steps, and a CU hides
// one block's chain behind the others': profiles/README.md).  Every step starts by writing the bytes that
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
template <uint32_t WIN, bool RCRC = false>
__global__ __launch_bounds__(kD2Threads, WIN < kMaxBlockLen ? 6 : 1) void decode_indexed_kernel(Decode2Params prm) {
  constexpr bool RING = WIN < kMaxBlockLen;
  static_assert(RING || !RCRC, "the whole-block instantiation checksums its window at the end");
  constexpr uint32_t kOutSink = WIN;
  constexpr uint32_t kSlack = RCRC ? 1024 : 0;
  auto wa = [](uint32_t x) -> uint32_t { return RING ? (x & (WIN - 1)) : x; };  // window address of output byte x
  // (the ring window is a static array: its LDS address is then a compile-time constant that folds into the
  // instructions' offset fields -- the dynamic array's base is added to every address with an instruction;
  // -2.5 % kernel time)
  __shared__ __attribute__((aligned(16))) uint8_t s_static_window[RING ? out_alloc(kRingWin) : 16];
  uint8_t* const s_out = RING ? s_static_window : s_dyn_window;
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kD2Ring + 16];
  // pointer-doubling / start-mask scratch, one per resolver wave
  __shared__ __attribute__((aligned(16))) uint16_t s_r16[kD2Pool][kGroup];
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
  __shared__ uint16_t s_gidx[kMaxBlockLen / kGroup];  // list slot of the element covering byte 256 m
  __shared__ uint32_t s_crc_acc, s_crc_cnt;            // the unit's CRC: XOR of the waves' parts, waves done
  __shared__ uint32_t s_runbad;                          // the unit is not one literal + copies of one offset
  __shared__ uint32_t s_rcrc_tab[RCRC ? 1024 : 1];       // (ring + CRC) the four stride tables of the column scheme

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
  const bool fe = wave <= 1;                             // front-end waves
  const uint32_t half = wave == 1 ? 1 : 0;               // which 1 KiB of the step is mine
  uint4 rf[4];  // (wave 1) the ring's first 4 KiB
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t q = (lane + 64 * i) * 16;
    rf[i] = make_uint4(0, 0, 0, 0);
    if (wave == 1 && q < q_end) rf[i] = *reinterpret_cast<const uint4*>(g0 + q);
  }
  const uint32_t ie0 = half * 64 + lane < n_regions0 ? idx[half * 64 + lane] : 0;
  const uint32_t io0 = (1 - half) * 64 + lane < n_regions0 ? idx[(1 - half) * 64 + lane] : 0;
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

  if (tid == 0) {
    s_front = 0;
    s_skip = 0;
    s_err = 0;
    s_cnt[0] = 0;
    s_cnt[1] = 0;
    s_crc_acc = 0;
    s_crc_cnt = 0;
    s_runbad = 0;
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

  // Copy L (0 = nothing, <= 64) bytes to s_out[dst..].  rd(k) returns the k-th ALIGNED dword of
  // the source counted from the dword that holds its first byte; sh = source address & 3.
  // Unaligned LDS dword accesses cost ~10-20x an aligned one on gfx950 (tools/probes/
  // lds_rates.hip), so sources are read as aligned dwords and re-aligned with a funnel shift,
  // and the destination is written bytewise.
  auto lean_copy = [&](uint32_t dst, auto rd, uint32_t sh, uint32_t L) {
    const uint32_t sh8 = sh * 8;
    uint32_t s0 = rd(0u), s1 = rd(1u), s2 = rd(2u);
    uint32_t v0 = __funnelshift_r(s0, s1, sh8), v1 = __funnelshift_r(s1, s2, sh8);
    // A byte's address is (it is mine ? the element's window address : my sink) + j: one select per byte, and
    // the + j rides in the store's offset field.  (An element that runs over the ring's end -- one in a few
    // thousand -- takes the form with a wrap per byte.)
    const uint32_t b0 = wa(dst);
    if (RING && __builtin_expect(ballot(L != 0 && b0 + L > WIN) != 0, 0)) {
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) s_out[L > j ? wa(dst + j) : sink + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) s_out[L > 4 + j ? wa(dst + 4 + j) : sink + j] = (uint8_t)(v1 >> (8 * j));
      for (uint32_t k = 8; ballot(L > k); k += 8) {
        s0 = s2;
        s1 = rd(k / 4 + 1);
        s2 = rd(k / 4 + 2);
        v0 = __funnelshift_r(s0, s1, sh8);
        v1 = __funnelshift_r(s1, s2, sh8);
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) s_out[L > k + j ? wa(dst + k + j) : sink + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
          s_out[L > k + 4 + j ? wa(dst + k + 4 + j) : sink + j] = (uint8_t)(v1 >> (8 * j));
      }
      return;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) s_out[(L > j ? b0 : sink) + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) s_out[(L > 4 + j ? b0 : sink - 4) + 4 + j] = (uint8_t)(v1 >> (8 * j));
    for (uint32_t k = 8; ballot(L > k); k += 8) {  // longer elements: 8 more bytes per trip
      s0 = s2;
      s1 = rd(k / 4 + 1);
      s2 = rd(k / 4 + 2);
      v0 = __funnelshift_r(s0, s1, sh8);
      v1 = __funnelshift_r(s1, s2, sh8);
      const uint32_t bk = b0 + k;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) s_out[(L > k + j ? bk : sink) + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) s_out[(L > k + 4 + j ? bk : sink - 4) + 4 + j] = (uint8_t)(v1 >> (8 * j));
    }
  };
  auto ring_al = [&](uint32_t q) -> uint32_t {  // aligned dword that holds stream byte q - shift
    return *reinterpret_cast<const uint32_t*>(s_ring + (q & (kD2Ring - 1) & ~3u));
  };
  // keeps the compiler from reordering LDS traffic across it; the hardware executes the LDS
  // operations of one wave in issue order, one instruction at a time for the whole CU
  auto cbar = [] { asm volatile("" ::: "memory"); };

  auto fill_ring = [&](uint32_t step) {  // (wave 1) the 4 KiB of the stream that start with `step`
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = step * kChunk + (lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, *reinterpret_cast<const uint4*>(g0 + q));
    }
  };
  if (wave == 1) {  // land what was fetched at the start
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = (lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, rf[i]);
    }
  }

  // Index entries of the NEXT step (mine and the other front-end wave's) are fetched at the start
  // of a step and consumed at the start of the next one, before anything younger is issued, so
  // the wait for them never also waits for fresh loads.  Entries past the end read as "none".
  const uint32_t none_end = kIdxNone | (total << 11);
  auto idx_at = [&](uint32_t r) -> uint32_t { return r < n_regions ? idx[r] : none_end; };
  // (every wave executes these loads -- straight-line code keeps the compiler from copying the
  // loaded registers, and thereby waiting for them, at the end of the front-end branch)
  uint32_t ie_pref = half * 64 + lane < n_regions ? ie0 : none_end;
  uint32_t io_pref = (1 - half) * 64 + lane < n_regions ? io0 : none_end;
  if (tid <= n_chunks && tid < kMaxSteps) s_sbase[tid] = tid < n_chunks ? sb0 >> 11 : total;
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
    if (tid >= 256) {
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
  bool passed_on = false;
  uint4 pre[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  uint32_t pq[2] = {0xffffffffu, 0xffffffffu};  // ring data in flight (wave 1)

  unsigned long long tm_work = 0, tm_bar = 0;             // DEBUG timers
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
        if (wave == 1 && s < n_chunks) fill_ring(s);
        ie_pref = idx_at(s * 128 + half * 64 + lane);
        io_pref = idx_at(s * 128 + (1 - half) * 64 + lane);
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
    if (wave == 1 && s < n_chunks) {
      // the ring slots of the previous step are free now: land the 2 KiB fetched meanwhile
#pragma unroll
      for (int i = 0; i < 2; i++)
        if (pq[i] < q_end) ring_store(pq[i], pre[i]);
    }
    // next step's index entries and the 2 KiB of stream after the ring's contents
    // (the front-end waves only: the resolvers have no use for them)
    if (fe) {
      ie_pref = idx_at((s + 1) * 128 + half * 64 + lane);
      io_pref = idx_at((s + 1) * 128 + (1 - half) * 64 + lane);
      if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
          pq[i] = s * kChunk + kD2Ring + (lane + 64 * i) * 16;
          const uint32_t qc = pq[i] < q_end ? pq[i] : 0;  // clamped: always a valid address
          pre[i] = *reinterpret_cast<const uint4*>(g0 + qc);
        }
      }
    }

    if (fe && s < n_chunks) {
      // =================================== front end ===========================================
      // (the front end's trips and the resolvers' turns are the two chains a step waits for: their
      // instructions go first; the preparation of later groups fills the gaps)
      __builtin_amdgcn_s_setprio(3);
      const uint32_t buf = s & 1;
      const uint32_t c0 = s * kChunk + half * (kChunk / 2);
      const uint32_t e_off = ie & 63;
      const bool had = e_off != kIdxNone;
      const uint32_t nel = had ? (ie >> 6) & 31 : 0;  // elements that start in my region
      const uint32_t onel = (io_cur & 63) != kIdxNone ? (io_cur >> 6) & 31 : 0;
      uint32_t dst = ie >> 11;
      uint32_t ctot, otot;
      uint32_t slot = wave_excl_scan(nel, lane, &ctot);
      (void)wave_excl_scan(onel, lane, &otot);
      if (half) slot += otot;  // the first half's elements come first in the list
      ctot += otot;            // elements of the whole step (<= kElemCap by the format)
      if (wave == 0 && lane == 0) s_cnt[buf] = ctot;
      uint32_t* const el = s_el[buf];

      const uint32_t rs = c0 + lane * kSub;
      const uint32_t r_end = rs + kSub < n ? rs + kSub : n;
      uint32_t pos = rs + e_off;
      bool live = had && pos < n && !(SNAPPY_DBG(prm) & 4);
      bool big = false;  // a literal longer than 64 bytes ends my region: done below
      uint32_t big_dst = 0, big_len = 0, big_src = 0, big_slot = 0;
      bool bad = false;
      // the tag and the four bytes after it are fetched one trip ahead: as soon as an element's
      // size is known the next element's bytes are requested, before this one's stores are issued
      uint32_t t0 = ring_al(pos + shift), t1 = ring_al(pos + shift + 4), t2 = ring_al(pos + shift + 8);
      while (ballot(live)) {
        acc_a++;
        const uint32_t q = pos + shift;
        const uint32_t w0 = __funnelshift_r(t0, t1, (q & 3) * 8), w1 = __funnelshift_r(t1, t2, (q & 3) * 8);
        const uint32_t b14 = (w0 >> 8) | (w1 << 24);
        bool is_copy;
        uint32_t L, size, hdr, off;
        decode_fast(w0 & 0xff, b14, &is_copy, &L, &size, &hdr, &off);
        {
          const uint32_t qn = q + (live ? size : 0);
          cbar();
          t0 = ring_al(qn);
          t1 = ring_al(qn + 4);
          t2 = ring_al(qn + 8);
          cbar();
        }
        const bool cpy = live && is_copy;
        const bool lit = live && !is_copy;
        const bool bad_off = cpy && (off == 0 || off > dst);  // decoder.nim:112
        bad = bad || bad_off;
        // ---- the element's list entry, and its slot at the 256-byte boundary it covers (if any) --
        const bool put = live && slot < kElemCap;
        *(put ? el + slot : reinterpret_cast<uint32_t*>(sink16)) = ((cpy && !bad_off) ? (off & 0xffffu) : 0u) | (dst << 16);
        const uint32_t mb = (dst + kGroup - 1) / kGroup;
        const bool covers = live && mb * kGroup < dst + L;
        *(covers ? s_gidx + (mb & (kMaxBlockLen / kGroup - 1)) : sink16) = (uint16_t)slot;
        // ---- literal: payload of up to 8 bytes here, up to 64 in lean_copy's rare loop -----------
        const uint32_t qs = q + hdr;
        const uint32_t Lw = (lit && L <= 64 && !(SNAPPY_DBG(prm) & 1)) ? L : 0;  // bytes this lane writes
        lean_copy(dst, [&](uint32_t k) { return ring_al(qs + 4 * k); }, qs & 3, Lw);
        if (lit && L > 64) {
          big = true;
          big_dst = dst;
          big_len = L;
          big_src = pos + hdr;
          big_slot = slot;
        }
        slot += live ? 1 : 0;
        dst += live ? L : 0;
        pos += live ? size : 0;
        live = live && pos < r_end;
      }
      // long literals: whole wave, straight from HBM (at most one per region)
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
      if (ballot(bad) && lane == 0) s_err = 1;
      __builtin_amdgcn_s_setprio(0);
      acc_b += 1;
    } else if (!fe && s >= 1 && !(SNAPPY_DBG(prm) & 2)) {
      // =================================== resolvers ===============================================
      const uint32_t buf = (s - 1) & 1;
      const uint32_t cb = readfirst(s_sbase[s - 1]), cn = readfirst(s_sbase[s]);
      const uint32_t count = readfirst(s_cnt[buf]);
      const uint16_t* const el16 = reinterpret_cast<const uint16_t*>(s_el[buf]);
      auto o16 = [&](uint32_t e) -> uint32_t { return el16[2 * e]; };      // copy offset, 0 = literal
      auto d16 = [&](uint32_t e) -> uint32_t { return el16[2 * e + 1]; };  // first output byte
      uint16_t* const r16 = s_r16[wave - 2];
      const uint32_t gfirst = cb & ~(kGroup - 1);
      // a group in the middle of one long literal holds no copy (the front end flags the
      // boundaries such a literal covers); nobody works on it, the group before it publishes it
      auto is_skip = [&](uint32_t gg) -> bool {  // gg > cb (per lane)
        const uint32_t a = s_gidx[gg / kGroup];
        const uint32_t b2 = s_gidx[(gg / kGroup + 1) & (kMaxBlockLen / kGroup - 1)];
        return (a & 0x8000u) && gg + kGroup < cn && a == b2;
      };
      uint32_t front = cb;  // what I know of s_front
      constexpr uint32_t B = kGroup / 64;  // bytes per lane (4 or 8): one or two aligned dwords
      // (an empty step has nothing to resolve -- and after a fast-forward its list is not even its own)
      for (uint32_t g = gfirst + (wave - 2) * kGroup; g < cn && cb < cn; g += kD2Pool * kGroup) {
        // (one round trip for the three words a group starts with: who covers its first byte, whether
        // the group behind it is inside a long literal, how far the frontier is)
        const uint32_t ge_raw = s_gidx[g / kGroup];
        const uint32_t gn_raw = s_gidx[(g / kGroup + 1) & (kMaxBlockLen / kGroup - 1)];
        const uint32_t fr_raw = __hip_atomic_load(&s_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cbar();
        const uint32_t ge = g > cb ? readfirst(ge_raw) : 0;
        const uint32_t gnext = readfirst(gn_raw);
        if ((ge & 0x8000u) && readfirst(is_skip(g) ? 1u : 0u)) continue;  // (flag first: one read for most groups)
        acc_c++;
        const uint32_t p = g + B * lane;
        front = readfirst(fr_raw);
        if (front > (g > cb ? g : cb)) continue;  // a run extension (below) has covered my group
        // the element that covers byte g is E0 (none in the step's first group when it starts
        // inside it)
        const uint32_t E0 = g > cb ? (ge & 0x7fffu) : (g == cb ? 0u : 0xffffffffu);
        // every element after E0 that starts inside the group writes its index (1 = E0 + 1, ...) at
        // its first byte's slot of the scratch; "which element covers byte x" is then the largest
        // index at or below x: a prefix maximum.  (Slot 0 cannot hold a start -- E0 covers byte g
        // -- and serves as the sink of the lanes that have nothing to write.)
        uint32_t* const r32 = reinterpret_cast<uint32_t*>(r16 + B * lane);  // my B slots
#pragma unroll
        for (uint32_t k = 0; k < B / 2; k++) r32[k] = 0;
        cbar();
        for (uint32_t e = E0 + 1 + lane;; e += 64) {
          const uint32_t d = e < count ? d16(e) : 0xffffffffu;
          const bool in = d - g < kGroup;  // (d > g: the list is in output order)
          r16[in ? d - g : 0] = (uint16_t)(e - E0);
          if (ballot(in) != ~0ull) break;
        }
        cbar();
        uint32_t li[B];  // index (relative to E0) of the element that covers each of my bytes
#pragma unroll
        for (uint32_t k = 0; k < B / 2; k++) {
          const uint32_t rv = r32[k];
          li[2 * k] = rv & 0xffffu;
          li[2 * k + 1] = rv >> 16;
        }
        cbar();
        li[0] = lane == 0 ? 0 : li[0];
#pragma unroll
        for (uint32_t j = 1; j < B; j++) li[j] = li[j] > li[j - 1] ? li[j] : li[j - 1];
        uint32_t tot;  // elements that start inside the group
        const uint32_t before = wave_excl_scan_max(li[B - 1], lane, &tot);
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
          const uint32_t ei = E0 + (li[j] > before ? li[j] : before);
          const uint32_t off = o16(in ? ei : 0);
          cp[j] = in && off != 0;
          offj[j] = cp[j] ? off : 0;
          sp[j] = p + j - offj[j];
          anyc = anyc || cp[j];
        }
        // A group that is one run of copies with one offset (how the encoder splits a long match,
        // encoder.nim:97-112): if the run goes on, I will also do the whole groups that follow
        // inside it, one group per trip, once it is my turn.  run_end = first byte after them.
        // (only groups of few, long elements are examined: tot = element starts in the group)
        uint32_t run_off = 0, run_end = 0;
        if (__builtin_expect(tot <= kGroup / 32 && g >= cb && g + kGroup <= cn && (run_off = readfirst(offj[0])) != 0, 0)) {
#pragma unroll
          for (uint32_t j = 0; j < B; j++) off_differs = off_differs || offj[j] != run_off;
          if (ballot(off_differs) == 0) {
            uint32_t R = cn;
            for (uint32_t e = E0 + tot + 1 + lane;; e += 64) {  // elements after those of my group
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
#pragma unroll
            for (uint32_t j = 0; j < B; j++) {
              const uint32_t t = r16[dep[j] ? sp[j] - g : 0];
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
        // (device-scope loads: past this CU's vector cache, which may hold the line's older state)
        uint32_t far_m = 0, far_v = 0;
        if (RING) {
          bool fj[B];
          bool anyfar = false;
#pragma unroll
          for (uint32_t j = 0; j < B; j++) {
            fj[j] = cp[j] && sp[j] < ring_lo;
            anyfar = anyfar || fj[j];
          }
          if (__builtin_expect(ballot(anyfar) != 0, 0)) {
#pragma unroll
            for (uint32_t j = 0; j < B; j++) {
              const uint32_t b = __hip_atomic_load(gout + (fj[j] ? sp[j] : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              far_m |= fj[j] ? 0xffu << (8 * j) : 0;
              far_v |= fj[j] ? b << (8 * j) : 0;
              sp[j] = fj[j] ? p + j : sp[j];  // (its window read below: anything inside the window)
            }
          }
          // A run is extended inside the ring: its sources must still be there.  (Otherwise group by group,
          // like any other.)
          if (run_end > g + kGroup) {
            const uint32_t lo = g >= run_off + 1024 ? g - run_off - 1024 : 0;  // (below every source it reads)
            if (lo < ring_lo) run_end = 0;
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
        const uint32_t expect = g > cb ? g : cb;
        // (from here to the publish this wave is, or is about to be, on the step's critical path)
        __builtin_amdgcn_s_setprio(3);
#ifndef D2_LOOSE_POLL
        // (the wait is written out: read, wait, compare -- five instructions a poll; the compiler's form of the
        // loop below took eighteen scalar instructions a poll, and the scalar unit is shared by the CU's waves)
        for (uint32_t spin = 0; front < expect; spin += 1024) {
          uint32_t left = 1024, fv;
          const uint32_t fa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_front;
          asm volatile(
              "1:\n"
              "ds_read_b32 %[fv], %[fa]\n"
              "s_waitcnt lgkmcnt(0)\n"
              "v_readfirstlane_b32 %[fr], %[fv]\n"
              "s_cmp_ge_u32 %[fr], %[ex]\n"
              "s_cbranch_scc1 2f\n"
              "s_sub_u32 %[left], %[left], 1\n"
              "s_cmp_lg_u32 %[left], 0\n"
              "s_cbranch_scc1 1b\n"
              "2:\n"
              : [fr] "+s"(front), [fv] "=&v"(fv), [left] "+s"(left)
              : [fa] "v"(fa), [ex] "s"(expect)
              : "scc", "memory");
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
  if (SNAPPY_STATS(prm) && lane == 0 && (wave == 0 || wave == 2)) {  // DEBUG
    unsigned long long* st = prm.stats + (wave == 0 ? 0 : 8);
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
      uint32_t part = tid >= 256 ? crc_reg : 0;
      for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
      if (lane == 0 && wave >= 4) {
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
