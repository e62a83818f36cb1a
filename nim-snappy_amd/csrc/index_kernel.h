// index_kernel.h -- pass 1 of the v2 block decoder: where do elements start?
//
// A Snappy tag stream (snappy/decoder.nim:39-109) can only be split by walking it: whether a
// byte is a tag or literal payload depends on everything before it.  This kernel breaks that
// serial chain with lane-parallel speculation, at high occupancy (one wave per unit, ~8 KiB of
// LDS, no output staging), and leaves a small index behind so that the decode kernel
// (decode2_kernel.h) can start every 32-byte region of the stream independently:
//
//   1. Right-to-left pass, one 32-byte region per lane: for EVERY byte k of the region compute
//      "if an element started here, where does the element chain leave my region, how many
//      output bytes and how many elements does it hold on the way".  Going right to
//      left, position k only needs its own element and the already finished entry of k+size,
//      so all 64 lanes run 32 independent steps (table in LDS, 33-dword row stride = no bank
//      conflicts).
//   2. Chain across the 64 regions of a 2 KiB chunk as a fixed-point iteration: lane 0 knows
//      its true entry; every other lane guesses; each round every lane looks up where a chain
//      entering at its current guess leaves, and hands that to the next lane.  Lane r is exact
//      after at most r rounds, in practice after a few (chains that start at different bytes
//      merge quickly); the loop stops when nothing changes, which is exactly "all correct".
//   3. A wave prefix sum of the per-region output byte counts gives every region its output
//      position.
//
//   4. Each lane then walks its own region once more, from its entry (the sizes of the elements at its
//      32 positions are in eight registers), and marks every element start on the way.
//
// Index entry per 16 bytes of stream (u32):  [0:16) bit k = an element starts at byte k of them;
// [16:32) output position of the first such element (< 65536: an element writes at least one byte).
// Where no element starts (the payload of a long literal): 0 in the low half, and the output position of
// the next element MINUS ONE in the high half (the unit's total can be 65536; it is never 0 there: the
// stream's first byte is an element start).  With the starts known, the decode kernel's front end is
// position-parallel: no lane has to decode element k to find element k + 1.
//
// All input-side checks of decodeAllTags (truncated elements, the 61-byte rule of
// decoder.nim:54-57, 4-byte length wrap :67-68, length bounds :77-79 / :127-128 through the
// total) are decided here; copy offsets (:112) are checked by the decode kernel.
#pragma once

#include <type_traits>

#include "common.h"
#include "decode_kernel.h"

namespace snappy_hip {

constexpr uint32_t kRegion = 32;                    // stream bytes per lane
constexpr uint32_t kChunk = 64 * kRegion;           // stream bytes per wave step
constexpr uint32_t kRowStride = kRegion + 1;        // LDS row stride in dwords
constexpr uint32_t kExitEnd = 0, kExitErr = 1;      // exit field: chain ended / invalid element
constexpr uint32_t kExitFar = 896;                  // exit field >= 896: far exit = 896 + 32 * (length bytes - 1) + k
constexpr uint32_t kOutSat = 0x1ffff - 1024;        // saturated element length (> 65536 = invalid); leaves
                                                    // room for the <= 16 x 64 bytes a region's chain adds on top
constexpr uint32_t kIdxNone = 32;                    // (region entry offset: no element starts in the region)
// index entries (see above)
__device__ __forceinline__ uint32_t idx_starts(uint32_t e) { return e & 0xffffu; }
__device__ __forceinline__ uint32_t idx_first_dst(uint32_t e) { return (e >> 16) + ((e & 0xffffu) ? 0u : 1u); }  // of the first start at or behind the entry's bytes
__device__ __forceinline__ uint32_t idx_none(uint32_t next_dst) { return (next_dst ? next_dst - 1 : 0u) << 16; }
constexpr uint32_t kSub = 16;                       // stream bytes per index entry (half a region)
}  // namespace snappy_hip
#include "sparse_kernel.h"  // (a unit of few, long elements is decoded by the wave that indexed it)
namespace snappy_hip {
constexpr uint32_t kSizeStride = 36;                // byte stride of a lane's row in the size table
// Longest tag stream the indexed path takes (a valid 64 KiB block needs at most 76 490 bytes,
// snappy/codec.nim:217); longer units go to the one-pass kernel.
constexpr uint32_t kMaxFastIn = 81920;  // >= maxCompressedLen(65536) = 76490: 40 steps of 2 KiB

__device__ __forceinline__ uint32_t t_pack(uint32_t exit_rel, uint32_t nelem, uint32_t outsum) {
  return exit_rel | (nelem << 10) | (outsum << 15);
}
__device__ __forceinline__ uint32_t t_exit(uint32_t t) { return t & 1023; }
__device__ __forceinline__ uint32_t t_nelem(uint32_t t) { return (t >> 10) & 31; }
__device__ __forceinline__ uint32_t t_out(uint32_t t) { return t >> 15; }
// value of lane-1 (0 for lane 0) through DPP, no LDS round trip
__device__ __forceinline__ uint32_t lane_prev_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

struct IndexParams {
  const uint8_t* in;
  const uint64_t* in_off;
  const uint32_t* in_len;
  const uint32_t* out_cap;
  uint32_t* out_len;
  uint32_t* status;
  const uint64_t* idx_off;  // first index entry of each unit (nullptr: u * idx_stride)
  uint64_t idx_stride;
  uint32_t* idx;
  uint64_t n_units;
  int unit;
  uint32_t* blk_in;  // split mode: stream position (after the varint) where output block k starts
  const uint32_t* order;  // workgroup i takes unit order[i] (nullptr: unit i)
  // Units of 2 .. kSparseMax elements whose stream is longer than the indexed decoder's stream ring are DECODED here, by
  // the wave that has just indexed them: few, long elements (long literals, stretches of 64-byte copies) are the
  // business of sparse_kernel.h.  (One element: a literal, copied straight by the indexed decoder; a short stream of many
  // copies: a period, written from one image there.)  sparse = 0: never; out / out_off: where the units' output goes.
  uint32_t sparse;
  uint8_t* out;
  const uint64_t* out_off;
  uint32_t* sparse_counters;  // (may be nullptr) [0..1] 64-bit sum of such units' stream + output bytes, [2] their number
};

// Decode "the element that would start here" from its tag and the four bytes after it.
// rem = stream bytes after the tag.  Returns false if the element is invalid.
__device__ __forceinline__ bool decode_element(uint32_t tag, uint32_t b14, uint32_t rem,
                                               bool* is_copy, uint32_t* L, uint32_t* size,
                                               uint32_t* hdr, uint32_t* off) {
  const uint32_t hi6 = tag >> 2, t = tag & 3;
  bool ok = true;
  *is_copy = t != 0;
  *hdr = 1;
  *off = 0;
  if (t == 0) {  // decoder.nim:42-84
    uint32_t len = hi6 + 1, h = 1;
    if (len >= 61) {
      ok = rem >= 61;  // decoder.nim:54-57
      const uint32_t lenlen = len - 60;
      const uint32_t m = lenlen == 4 ? 0xffffffffu : ((1u << (8 * lenlen)) - 1);
      len = (b14 & m) + 1;
      ok = ok && len != 0;  // decoder.nim:67-68
      h = 1 + lenlen;
    }
    ok = ok && !(rem - (h - 1) < len);  // decoder.nim:78
    *L = len;
    *hdr = h;
    *size = h + len;
  } else if (t == 1) {  // decoder.nim:86-94
    ok = rem >= 1;
    *L = 4 + (hi6 & 7);
    *off = ((tag & 0xe0) << 3) | (b14 & 0xff);
    *size = 2;
  } else if (t == 2) {  // decoder.nim:95-102
    ok = rem >= 2;
    *L = 1 + hi6;
    *off = b14 & 0xffff;
    *size = 3;
  } else {  // decoder.nim:103-109
    ok = rem >= 4;
    *L = 1 + hi6;
    *off = b14;
    *size = 5;
  }
  return ok;
}

// The same without branches (selects only), for the pass that evaluates every byte position.
__device__ __forceinline__ bool decode_element_bf(uint32_t tag, uint32_t b14, uint32_t rem, uint32_t* L,
                                                  uint32_t* size) {
  const uint32_t hi6 = tag >> 2, t = tag & 3;
  // literal (decoder.nim:42-84)
  const bool longlit = hi6 >= 60;
  const uint32_t lenlen = longlit ? hi6 - 59 : 0;
  const uint32_t m = 0xffffffffu >> (32 - 8 * (lenlen ? lenlen : 4));
  const uint32_t llen = longlit ? (b14 & m) + 1 : hi6 + 1;
  const uint32_t h = 1 + lenlen;
  const bool lit_ok = (!longlit || (rem >= 61 && llen != 0)) && !(rem - (h - 1) < llen);  // :54-57, :67-68, :78
  // copies (decoder.nim:86-109)
  const uint32_t csize = t == 1 ? 2 : (t == 2 ? 3 : 5);
  const bool copy_ok = rem >= csize - 1;
  const uint32_t clen = t == 1 ? 4 + (hi6 & 7) : 1 + hi6;
  *L = t == 0 ? llen : clen;
  *size = t == 0 ? h + llen : csize;
  return t == 0 ? lit_ok : copy_ok;
}

// SPLIT = false: the index pass of the block decoder (units of at most 64 KiB output).
// SPLIT = true: one wave walks ONE raw buffer of any length the same way, chunk after chunk, and
// records where every 64 KiB block of its output starts in the stream, so that the blocks can
// then be decoded in parallel like independent units.  That works when no element straddles a
// 64 KiB output boundary (true for every encoder that works in 64 KiB blocks, snappy.nim:49-62);
// otherwise the unit is handed to the whole-stream kernel (kNeedsStreamKernel).
// TEAM = true (round 5): the index pass of a SMALL batch -- kSplitWaves waves share a unit's walk the way SPLIT's do (a
// unit's 32 chunks one after the other by one wave are ~250 us whatever the batch; 1 024 units leave three quarters of
// the GPU's wave slots empty): 245 -> ~95 us for 1 024 units.  Of the units this pass writes itself (4.2) the one-literal
// and one-period ones are written by the wave that concludes the walk; units of few long elements are left to the indexed
// decoder's eight waves here (one wave's sparse decode of a unit is ~200 us: the longest thing in a small batch).
// In SPLIT mode kSplitWaves waves share the walk: the tables of a chunk do not depend on where the
// element chain enters it, so wave w prepares chunks w, w+4, ... ahead of time and only the short
// chain step waits for the previous chunk's result (handed on through an LDS mailbox).
constexpr uint32_t kSplitWaves = 4;
template <bool SPLIT, bool TEAM = SPLIT>
__global__ __launch_bounds__(TEAM ? 64 * kSplitWaves : 64) void index_units_kernel(IndexParams prm) {
  static_assert(TEAM || !SPLIT, "SPLIT is a team's walk");
  constexpr uint32_t W = TEAM ? kSplitWaves : 1;
  __shared__ uint32_t s_tab[W * 64 * kRowStride];
  // TEAM: [chunk & 7] = {chunk + 1, entry_abs, op, state | straddle << 2, -, the first region's output bytes and elements}
  __shared__ uint32_t s_mail[8][8];
  __shared__ uint32_t s_left;  // TEAM, not SPLIT: waves that have run out of chunks (their index entries are written)
  // per tag byte: element length [0:7), stream size [7:14), bit 14 = literal with length bytes
  __shared__ uint16_t s_lut[256];

  const uint32_t lane = lane_id();
  for (uint32_t tg = lane; tg < 256; tg += 64) {  // decoder.nim:42-109 for the forms without length bytes
    const uint32_t hi6 = tg >> 2, ty = tg & 3;
    const uint32_t len = ty == 1 ? 4 + (hi6 & 7) : 1 + hi6;          // literal < 61 and copy2/copy4: 1 + hi6
    const uint32_t sz = ty == 0 ? 1 + len : (ty == 1 ? 2 : (ty == 2 ? 3 : 5));
    const bool longlit = ty == 0 && hi6 >= 60;
    s_lut[tg] = (uint16_t)(longlit ? (1u << 14) : (len | (sz << 7)));
  }
  const uint32_t wave = TEAM ? readfirst(threadIdx.x >> 6) : 0;
  if (TEAM) {
    if (threadIdx.x < 64) (&s_mail[0][0])[threadIdx.x] = 0;
    if (threadIdx.x == 64) s_left = 0;
    __syncthreads();
  } else {
    wave_fence();
  }
  if (blockIdx.x >= prm.n_units) return;
  const uint64_t u = prm.order ? prm.order[blockIdx.x] : blockIdx.x;  // (launch order, crc_pack_kernels.h)
  const uint8_t* in0 = prm.in + prm.in_off[u];
  uint32_t n = prm.in_len[u];
  const uint32_t cap = prm.out_cap[u];
  uint32_t* idx = prm.idx + (prm.idx_off ? prm.idx_off[u] : u * prm.idx_stride);

  auto finish = [&](uint32_t st, uint32_t written) {
    if (lane == 0) {
      prm.status[u] = st;
      prm.out_len[u] = written;
    }
  };

  // ---- header: identical to decode_units_kernel ---------------------------------------------
  uint32_t limit;
  bool exact = false;
  if (prm.unit == kUnitRaw) {
    uint32_t ulen = 0, hdr = 0;
    if (lane == 0) hdr = parse_varint32(in0, n, &ulen);
    hdr = readfirst(hdr);
    ulen = readfirst(ulen);
    if (hdr == 0) return finish(kInvalidInput, 0);
    if (cap < ulen) return finish(kBufferTooSmall, 0);
    if (ulen == 0) return finish(hdr == n ? kOk : kInvalidInput, 0);
    in0 += hdr;
    n -= hdr;
    limit = ulen;
    exact = true;
  } else {
    if (n == 0) return finish(kOk, 0);
    if (cap == 0) return finish(kBufferTooSmall, 0);
    limit = cap;
  }
  if (!SPLIT && exact && limit > kMaxBlockLen) return finish(kNeedsStreamKernel, 0);
  const uint32_t win_limit = SPLIT ? limit : (limit < kMaxBlockLen ? limit : kMaxBlockLen);
  // what an over-long output means: a bigger unit goes to the whole-stream kernel
  const uint32_t too_long = limit > win_limit ? kNeedsStreamKernel : kInvalidInput;

  if (!SPLIT && n > kMaxFastIn) return finish(kNeedsOnePass, 0);  // (SPLIT: the host bounds n and the length)
  bool straddle = false;  // SPLIT: an element crosses a 64 KiB output boundary

  // stream size of the element at each position.  (The block decoder's index keeps the 32 sizes of a region in
  // eight registers: the pass is bound by how many waves fit a CU, and without this table a wave needs 8.75 KiB
  // of LDS: 18 waves instead of 14.)
  __shared__ uint8_t s_sz[SPLIT ? W * 64 * kSizeStride : 16];

  const uint32_t shift = (uint32_t)((uintptr_t)in0 & 15);
  const uint8_t* g0 = in0 - shift;                // 16-byte aligned
  const uint32_t g_end = (shift + n + 15) & ~15u;  // aligned end of the unit

  uint32_t entry_abs = 0;  // stream position of the next real element (uniform)
  uint32_t op = 0;         // output bytes before it (uniform)
  bool ended = false;
  const uint32_t row = (wave * 64 + lane) * kRowStride;
  const uint32_t row8 = SPLIT ? (wave * 64 + lane) * kSizeStride : 0;

  // TEAM: hand the chain over to the wave that owns the next chunk / tell everybody to stop
  uint32_t first_out = 0, first_n = 0;  // (uniform) the first region's chain: output bytes, elements
  auto post = [&](uint32_t ci, uint32_t state, uint32_t e_abs, uint32_t o, bool strad) {
    if (lane == 0) {
      uint32_t* m = s_mail[ci & 7];
      m[1] = e_abs;
      m[2] = o;
      m[3] = state | (strad ? 4u : 0u);
      m[5] = first_out;
      m[6] = first_n;
      asm volatile("" ::: "memory");
      __hip_atomic_store(&m[0], ci + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  bool strad_before = false;  // SPLIT: a straddling element in an earlier chunk
  bool concluded = false;     // TEAM, not SPLIT: this wave holds the walk's final state
  uint32_t n_elem_lane = 0;   // elements that start in my regions, over all chunks (sparse verdict)
  unsigned long long tA = 0, tW = 0, tC = 0, tt0 = 0, tt1 = 0, tt2 = 0;  // DEBUG (SPLIT, prm.idx != nullptr)
  const bool dbgt = SPLIT && prm.idx != nullptr;
  if (TEAM && wave == 0) post(0, 0, 0, 0, false);
#ifndef IDX_NO_EARLY_PERIOD
  // (wave 0 uses ITS OWN rows of s_tab as early_period_unit's scratch -- in a team the other waves are building their
  // tables in the rows behind them meanwhile)
  static_assert(64 * kRowStride * 4 >= kPeriodLdsBytes, "early_period_unit's scratch must fit one wave's table rows");
  if (!SPLIT && prm.sparse && n <= 4096 && wave == 0) {
    // a unit that is one literal + copies of one offset is recognised from its stream and written at once: no walk
    // (sparse_kernel.h, early_period_unit with the total unknown); anything else: the walk below
    // (TEAM: wave 0, in its own tables' LDS; the others build their chunks' tables meanwhile and are told to stop)
    uint32_t tot = 0;
    if (early_period_unit(in0, n, prm.out + prm.out_off[u], kPeriodTotalUnknown, s_tab, limit, exact, &tot)) {
      if (prm.sparse_counters && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(prm.sparse_counters), (unsigned long long)prm.in_len[u] + tot);
        atomicAdd(prm.sparse_counters + 2, 1u);
      }
      finish(kDoneEarly, tot);
      if (TEAM) post(1, 2, 0, 0, false);
      return;
    }
    wave_fence();
  }
#endif

  for (uint32_t c0 = wave * kChunk; c0 < n && !ended; c0 += W * kChunk) {
    const uint32_t rs = c0 + lane * kRegion;  // my region's first stream position
    const uint32_t ci = c0 / kChunk;
    uint32_t entry_off = kIdxNone, out_here = 0, nelem_here = 0;
    uint32_t szp[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (block decoder's index) sizes of the elements at my 32 positions, a byte each
    // a verdict inside the loop: write it and, in SPLIT mode, stop the other waves
    auto bail = [&](uint32_t st) {
      finish(st, 0);
      if (TEAM) post(ci + 1, 2, 0, 0, false);
    };

    if (dbgt) tt0 = __builtin_amdgcn_s_memtime();
    // (the tables of a chunk that a long literal covers entirely are not needed; SPLIT cannot
    // know that yet and builds them anyway)
    if (TEAM || entry_abs < c0 + kChunk) {
      // ---- my 32 region bytes + 8 bytes lookahead, from 16-byte aligned loads ---------------
      uint32_t w[10];
      {
        uint32_t a[16];
        const uint32_t q = rs + shift;
        const uint32_t qa = q & ~15u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (qa + 16 * i < g_end) v = *reinterpret_cast<const uint4*>(g0 + qa + 16 * i);
          a[4 * i] = v.x;
          a[4 * i + 1] = v.y;
          a[4 * i + 2] = v.z;
          a[4 * i + 3] = v.w;
        }
        const uint32_t sh = q & 15;
        const uint32_t dsh = sh >> 2, bsh = (sh & 3) * 8;
        // dword-granular part of the shift with selects, byte part with a funnel shift
        uint32_t b[11];
#pragma unroll
        for (int i = 0; i < 11; i++) {
          b[i] = dsh == 0 ? a[i] : dsh == 1 ? a[i + 1] : dsh == 2 ? a[i + 2] : a[i + 3];
        }
#pragma unroll
        for (int i = 0; i < 10; i++) w[i] = __funnelshift_r(b[i], b[i + 1], bsh);
      }

      // ---- right-to-left pass over my 32 positions --------------------------------------------
      // (interior = the whole step and the longest short element behind it lie inside the stream:
      // the bounds checks of the short forms fold away, which is one instruction in six)
      // The entry of position k needs the finished entry of k + size: a chain of LDS round trips down the region.  Two
      // things keep it short.  The tag table is read for eight positions at a time (those reads depend on nothing).
      // And the table read of position k - 1 is issued BEFORE the entry of k is written: an element is at least two
      // bytes (a tag and a byte of literal or offset), so k - 1 never needs k's entry, only k + 1's and later ones --
      // two positions' round trips are in flight at any time.  (The compiler cannot know that and keeps the program's
      // order of LDS accesses: read, wait, write, read, wait: 64 trips a region before, about 24 now.)
      auto tabulate = [&](auto interior_c) {
      constexpr bool kInterior = decltype(interior_c)::value;
      struct Info {
        uint32_t L, out_code, Ls, szb, ridx;
        bool inside, ok, inreg;
      };
      auto tag_of = [&](int k) -> uint32_t {
        const uint32_t lo = w[k >> 2];
        return (lo >> ((k & 3) * 8)) & 0xff;
      };
      // what position k's own element says (e = its tag's table entry)
      auto make_info = [&](int k, uint32_t e) -> Info {
        Info f;
        const uint32_t p = rs + k;
        f.inside = kInterior || p < n;
        uint32_t L = e & 127, size = (e >> 7) & 127;
        bool ok = kInterior || (f.inside && p + size <= n);  // every short form: the element must end inside the stream
        // what the entry of an element that leaves the region looks like: the short forms (at most 65
        // bytes) exit at nx itself, with their own length and a size that fits a byte
        uint32_t nx = (uint32_t)k + size;
        uint32_t out_code = nx, Ls = L, szb = size;
        if (__builtin_expect(ballot(f.inside && (e >> 14)) != 0, 0)) {  // a literal with length bytes somewhere (rare in text)
          // (selects only: a divergent branch per position costs more than the work it skips)
          const uint32_t lo = w[k >> 2], hi = w[(k >> 2) + 1], hi2 = w[(k >> 2) + 2];
          const uint32_t sh8 = (k & 3) * 8;
          const uint32_t d0 = sh8 ? __funnelshift_r(lo, hi, sh8) : lo;   // bytes k..k+3
          const uint32_t d1 = sh8 ? __funnelshift_r(hi, hi2, sh8) : hi;  // bytes k+4..k+7
          const uint32_t tag = d0 & 0xff;
          const uint32_t b14 = (d0 >> 8) | (d1 << 24);
          const uint32_t rem = f.inside ? n - p - 1 : 0;
          const uint32_t lenlen = (tag >> 2) - 59;  // 1..4 where it applies
          const uint32_t m = 0xffffffffu >> (32 - 8 * (lenlen & 7 ? (lenlen & 7) : 4));
          const uint32_t llen = (b14 & m) + 1;
          const bool lok = rem >= 61 && llen != 0 && !(rem - lenlen < llen);  // decoder.nim:54-57, :67-68, :78
          const bool ll = f.inside && (e >> 14);
          L = ll ? llen : L;
          size = ll ? 1 + lenlen + llen : size;
          ok = ll ? lok : ok;
          nx = (uint32_t)k + size;  // ok => no wrap
          // (only such a literal can exit far away; with the number of its length bytes in the exit code
          // and its length in the entry itself the chain step needs no second look at the stream)
          const uint32_t far_code = kExitFar + ((((tag >> 2) - 60) & 3) << 5) + (uint32_t)k;
          out_code = nx < kExitFar ? nx : far_code;
          Ls = L < kOutSat ? L : kOutSat;
          szb = size < 255 ? size : 255;
        }
        f.L = L, f.out_code = out_code, f.Ls = Ls, f.szb = szb, f.ok = ok;
        f.inreg = ok && nx < kRegion;
        f.ridx = f.inreg ? nx : (uint32_t)k;
        return f;
      };
#pragma unroll
      for (int kb = kRegion - 8; kb >= 0; kb -= 8) {
        uint32_t ee[8];
#pragma unroll
        for (int j = 0; j < 8; j++) ee[j] = s_lut[tag_of(kb + j)];
        Info cur = make_info(kb + 7, ee[7]);
        uint32_t tn = s_tab[row + cur.ridx];
#pragma unroll
        for (int j = 7; j >= 0; j--) {
          const int k = kb + j;
          Info nxt = cur;
          uint32_t tn_nxt = 0;
          if (j > 0) {
            nxt = make_info(k - 1, ee[j - 1]);
            tn_nxt = s_tab[row + nxt.ridx];  // (before k's entry is written: see above)
          }
          // the fields are additive along the chain: same exit, one more element, L more bytes
          // (an error or end entry keeps its exit code; in-region lengths are <= 64, so the 17-bit
          // sum cannot overflow past the saturated terminal element)
          const uint32_t t_in = tn + ((1u << 10) | (cur.L << 15));
          const uint32_t t_out_of = cur.out_code | (1u << 10) | (cur.Ls << 15);
          const uint32_t t = !cur.inside ? t_pack(kExitEnd, 0, 0)
                                         : (!cur.ok ? t_pack(kExitErr, 0, 0) : (cur.inreg ? t_in : t_out_of));
          const uint32_t szb = cur.ok ? cur.szb : 255;
          s_tab[row + k] = t;
          if (SPLIT) s_sz[row8 + k] = (uint8_t)szb;
          else szp[k >> 2] |= szb << (8 * (k & 3));
          cur = nxt;
          tn = tn_nxt;
        }
      }
      };
      if (c0 + kChunk + 64 <= n) tabulate(std::true_type{});
      else tabulate(std::false_type{});
      wave_fence();
    }
    // ---- chain across the regions: fixed point ----------------------------------------------------
    // in_abs = stream position at which the element chain arrives at my region (>= rs).  Lane 0
    // knows it (e_abs), the others start from a guess; every round hands each region's exit to
    // the next lane, and when nothing changes any more all of them are exact.
    uint32_t in_abs = rs, out_abs = rs, tv = 0;
    uint32_t far_pos = 0xffffffffu, far_end = 0;  // the last far exit looked up (per lane)
    bool has = false;
    auto iterate = [&](uint32_t e_abs, uint32_t max_rounds) {
      if (lane == 0) in_abs = e_abs;
      for (uint32_t round = 0; round < max_rounds; round++) {
        has = in_abs < rs + kRegion;
        out_abs = in_abs;
        if (has) {
          tv = s_tab[row + (in_abs - rs)];
          const uint32_t ex = t_exit(tv);
          if (ex == kExitEnd || ex == kExitErr) {
            out_abs = 0xffffffffu;  // nothing follows
          } else if (ex >= kExitFar) {
            // far exit: a long literal at offset kk of my region; its entry holds its own length
            const uint32_t kk = (ex - kExitFar) & 31, lenlen = ((ex - kExitFar) >> 5) + 1;
            const uint32_t fl = t_out(s_tab[row + kk]);
            if (fl < kOutSat) {
              out_abs = rs + kk + 1 + lenlen + fl;
            } else {
              // saturated (longer than any block: the unit is about to be rejected or handed to the
              // whole-stream kernel): the exact end from the stream itself, looked up once
              const uint32_t fp = rs + kk;
              if (fp != far_pos) {
                const uint32_t ftag = in0[fp];
                uint32_t fb = 0;
                for (uint32_t i = 0; i < 4 && fp + 1 + i < n; i++) fb |= (uint32_t)in0[fp + 1 + i] << (8 * i);
                bool c;
                uint32_t L, size, hdr, off;
                decode_element(ftag, fb, 0xffffffffu, &c, &L, &size, &hdr, &off);
                far_pos = fp;
                far_end = fp + size;
              }
              out_abs = far_end;
            }
          } else {
            out_abs = rs + ex;
          }
        }
        // what arrives at my region is the exit of the nearest region in front of me that the chain really enters: a
        // region it passes over (has == false: a long literal's payload) hands on what it got, and lane by lane that
        // takes a round per region passed -- up to 63 for a chunk with a long literal in it -- so those are skipped
        const uint64_t hm = ballot(has);
        uint32_t prev;
        if (hm == ~0ull) {
          prev = lane_prev_u32(out_abs);
        } else {
          const uint64_t below = hm & ((1ull << lane) - 1);
          const uint32_t j = below ? 63 - (uint32_t)__builtin_clzll(below) : 0;
          const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)out_abs);
          prev = below ? v : e_abs;
        }
        // (a position below my region -- the nearest region that is entered exits into one in between that still
        // believes it is passed over -- says nothing about mine: I keep what I have, that region picks it up next round)
        const uint32_t nin = lane == 0 ? e_abs : (prev < rs ? in_abs : prev);
        const bool changed = nin != in_abs;
        in_abs = nin;
        if (!ballot(changed)) break;
      }
    };
    if (TEAM) {
      if (dbgt) tt1 = __builtin_amdgcn_s_memtime();
      // wait for the previous chunk's result
      uint32_t* m = s_mail[ci & 7];
      while (__hip_atomic_load(&m[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ci + 1)
        __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      entry_abs = readfirst(m[1]);
      op = readfirst(m[2]);
      const uint32_t fl = readfirst(m[3]);
      first_out = readfirst(m[5]);
      first_n = readfirst(m[6]);
      strad_before = (fl & 4) != 0;
      if (dbgt) tt2 = __builtin_amdgcn_s_memtime();
      if ((fl & 3) == 2) {  // somebody has written the verdict
        post(ci + 1, 2, 0, 0, false);
        return;
      }
    }
    if (entry_abs < c0 + kChunk) {  // otherwise a long literal covers the whole chunk
      iterate(entry_abs, 0xffffffffu);

      // ---- verdicts and this chunk's contribution ---------------------------------------------------
      const bool bad = has && t_exit(tv) == kExitErr;
      if (ballot(bad)) {
        bail(kInvalidInput);
        return;
      }
      if (has && in_abs < n) {  // in_abs == n is the end of the stream, not an element
        entry_off = in_abs - rs;
        out_here = t_out(tv);
        nelem_here = t_nelem(tv);
      }
      ended = ballot(has && t_exit(tv) == kExitEnd) != 0;
      entry_abs = readlane(out_abs, 63);
    }

    // output positions: saturating counts keep the sum below 2^32 (64 * 0x1ffff)
    uint32_t tot;
    const uint32_t before = wave_excl_scan(out_here, lane, &tot);
    if (op + tot > win_limit) {
      bail(too_long);
      return;
    }
    // SPLIT: a saturated length (an element of more than ~127 KiB) cannot be placed: whole-stream kernel
    if (SPLIT && ballot(out_here >= kOutSat)) {
      bail(kNeedsStreamKernel);
      return;
    }

    const uint32_t pos0 = op + before;
    if (SPLIT) {
      // ---- a 64 KiB output boundary inside my region's elements: which element starts there? ------
      const uint32_t kb = (pos0 + kMaxBlockLen - 1) / kMaxBlockLen;  // first boundary at or after pos0
      const uint32_t B = kb * kMaxBlockLen;
      if (entry_off != kIdxNone && kb >= 1 && B - pos0 < out_here) {
        uint32_t pw = entry_off, dst = pos0;
        bool found = false;
        for (uint32_t it = 0; it < kRegion && pw < kRegion; it++) {
          if (dst >= B) break;
          const uint32_t sz = s_sz[row8 + pw];
          const uint32_t nx = pw + sz;
          const uint32_t rest = t_out(s_tab[row + pw]);                       // bytes from pw to the exit
          const uint32_t after = nx < kRegion ? t_out(s_tab[row + nx]) : 0;  // ... from the next element
          dst += rest - after;
          pw = nx;
        }
        found = dst == B && pw < kRegion && rs + pw < n;
        if (found) prm.blk_in[kb] = rs + pw;
        // not at an element start, or the region's last element runs over the next boundary too
        if (!found && dst != B) straddle = true;
        // (dst == B without an element here: the chain left my region exactly at the boundary, the
        // region it lands in records it)
        if (pos0 + out_here > B + kMaxBlockLen) straddle = true;
      }
    } else {
    // ---- the element starts of my region: walk the chain from my entry, 16 positions at a time ------
    uint32_t bm = 0;
    {
      uint32_t pw = entry_off;  // (kIdxNone = 32: nothing starts here)
      while (ballot(pw < kSub)) {
        const uint32_t wv = pw < 4 ? szp[0] : (pw < 8 ? szp[1] : (pw < 12 ? szp[2] : szp[3]));
        if (pw < kSub) {
          bm |= 1u << pw;
          pw += (wv >> ((pw & 3) * 8)) & 0xffu;
          pw = rs + pw < n ? pw : kRegion;  // (the chain may end exactly at the stream's end: not an element)
        }
      }
      while (ballot(pw < kRegion)) {
        const uint32_t wv = pw < 20 ? szp[4] : (pw < 24 ? szp[5] : (pw < 28 ? szp[6] : szp[7]));
        if (pw < kRegion) {
          bm |= 1u << pw;
          pw += (wv >> ((pw & 3) * 8)) & 0xffu;
          pw = rs + pw < n ? pw : kRegion;
        }
      }
    }
    // the first start of the second half writes at pos0 + (what the starts of the first half produce)
    uint32_t pos1 = pos0 + out_here;  // (none there: the next element's position)
    if (bm >> kSub) pos1 = pos0 + (out_here - t_out(s_tab[row + kSub + ctz64(bm >> kSub)]));
    if (rs < n) {
      idx[2 * (rs / kRegion)] = (bm & 0xffffu) ? (bm & 0xffffu) | (pos0 << 16) : idx_none(pos1);
      idx[2 * (rs / kRegion) + 1] = (bm >> kSub) ? (bm >> kSub) | (pos1 << 16) : idx_none(pos0 + out_here);
    }
    }
    n_elem_lane += nelem_here;
    if (!SPLIT && c0 == 0) {
      first_out = readlane(out_here, 0);
      first_n = readlane(nelem_here, 0);
    }
    op += tot;
    if (TEAM) {
      const bool strad = SPLIT && (strad_before || ballot(straddle) != 0);
      if (ended || c0 + kChunk >= n) {  // the walk is complete: I hold the final state
        uint32_t st = kOk;
        if (!ended && entry_abs != n) st = kInvalidInput;
        else if (exact && op != limit) st = kInvalidInput;  // snappy.nim:107-108
        else if (strad) st = kNeedsStreamKernel;
        if (!SPLIT && st == kOk) {  // (the block decoder's index: the units this pass writes itself, below)
          concluded = true;
          break;
        }
        finish(st, st == kOk ? op : 0);
        post(ci + 1, 2, 0, 0, false);
        return;
      }
      post(ci + 1, 0, entry_abs, op, strad);
      if (dbgt) {
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        tA += tt1 - tt0;
        tW += tt2 - tt1;
        tC += t3 - tt2;
      }
    }
  }
  if (dbgt && lane == 0 && wave == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(prm.idx);
    dbg[0] = tA;
    dbg[1] = tW;
    dbg[2] = tC;
  }
  if (TEAM && !concluded) {  // (my chunks ran out; the wave that owns the last chunk concludes)
    if (!SPLIT) {  // ... and may read the index entries I wrote (sparse_decode_unit): they have arrived when it sees the count
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(&s_left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }
  if (TEAM) {  // the others have left: every chunk in front of my last one was theirs or mine (a wave without a chunk leaves at once)
    while (__hip_atomic_load(&s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != W - 1) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }

  // the chain must consume the stream exactly (every element was bounds-checked against n)
  if (!ended && entry_abs != n) return finish(kInvalidInput, 0);
  if (exact && op != limit) return finish(kInvalidInput, 0);  // snappy.nim:107-108
  if (!SPLIT && prm.sparse) {
    // Three kinds of unit this wave finishes itself, in the LDS of its tables (sparse_kernel.h): one literal; a literal
    // and copies of one offset (a period); few, long elements.  Their status, kDoneEarly, is carried past the indexed
    // decoder's launches.
    static_assert(TEAM || sizeof(s_tab) >= kSparseLds, "the sparse decoder works in the table's LDS");
    static_assert(TEAM || sizeof(s_tab) >= 4096 + 4096 + 32, "... and the period's stream and image");
    uint8_t* const gptr = prm.out + prm.out_off[u];
    auto count_early = [&]() {  // (bench.py: the bytes of these units are this pass's, not the indexed decoder's)
      if (prm.sparse_counters && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(prm.sparse_counters), (unsigned long long)prm.in_len[u] + op);
        atomicAdd(prm.sparse_counters + 2, 1u);
      }
    };
#ifndef IDX_NO_EARLY_LIT
    // The first element writes every byte.  It must BE a literal: this pass validates no copy offsets, and a unit
    // whose one element is a copy (`01 01`, `fe 01 00`: decoder.nim:112 rejects them, op <= offset - 1) would
    // otherwise be "copied" from the bytes behind its stream.  Such a unit falls through to kOk, and the indexed
    // decoder's offset check gives the reference's verdict.
    if (first_n == 1 && first_out == op && (in0[0] & 3) == 0) {
      const uint32_t hi6 = (uint32_t)in0[0] >> 2;
      const uint32_t lenlen = hi6 >= 60 ? hi6 - 59 : 0;
      if (1 + lenlen + op == n) {
        early_literal_unit(in0 + 1 + lenlen, gptr, op);
        count_early();
        return finish(kDoneEarly, op);
      }
    }
#endif
    if (n <= 4096) {
#ifndef IDX_NO_EARLY_PERIOD
      wave_fence();
      if (early_period_unit(in0, n, gptr, op, s_tab)) {
        count_early();
        return finish(kDoneEarly, op);
      }
#endif
    } else {
      uint32_t n_elem;
      (void)wave_excl_scan(n_elem_lane, lane, &n_elem);
      if (!TEAM && n_elem >= 2 && n_elem <= kSparseMax) {
        // (the index entries were written by other lanes of this wave: the stores are waited for -- one wave, one CU)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t st = sparse_decode_unit(in0, n, idx, gptr, op, s_tab);
        if (st == kOk) count_early();
        return finish(st == kOk ? kDoneEarly : st, st == kInvalidInput ? 0 : op);
      }
    }
  }
  finish(kOk, op);
}

// DEBUG: serial check of the index of one unit against a plain walk (one thread per unit).
// report[u*4..]: region of the first mismatch (or ~0), expected entry, got entry.
__global__ void verify_index_kernel(IndexParams prm, uint32_t* report) {
  const uint64_t u = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (u >= prm.n_units) return;
  report[u * 4] = 0xffffffffu;
  if (prm.status[u] != kOk || prm.out_len[u] == 0) return;
  const uint8_t* in0 = prm.in + prm.in_off[u];
  uint32_t n = prm.in_len[u];
  if (prm.unit == kUnitRaw) {
    uint32_t hdr = 0;
    while (in0[hdr] & 0x80) hdr++;
    hdr++;
    in0 += hdr;
    n -= hdr;
  }
  const uint32_t* idx = prm.idx + (prm.idx_off ? prm.idx_off[u] : u * prm.idx_stride);
  uint32_t pos = 0, dst = 0;
  const uint32_t nreg = (n + kSub - 1) / kSub;
  for (uint32_t r = 0; r < nreg; r++) {
    const uint32_t rs = r * kSub;
    uint32_t bm = 0, d0 = dst;
    while (pos < rs + kSub && pos < n) {
      uint32_t b = 0;
      for (uint32_t i = 0; i < 4 && pos + 1 + i < n; i++) b |= (uint32_t)in0[pos + 1 + i] << (8 * i);
      bool c;
      uint32_t L, size, hdr, off;
      decode_element(in0[pos], b, 0xffffffffu, &c, &L, &size, &hdr, &off);
      bm |= 1u << (pos - rs);
      dst += L;
      pos += size;
    }
    const uint32_t want = bm ? bm | (d0 << 16) : idx_none(dst);
    const uint32_t got = idx[r];
    if (got != want) {
      report[u * 4] = r;
      report[u * 4 + 1] = want;
      report[u * 4 + 2] = got;
      return;
    }
  }
}

}  // namespace snappy_hip
