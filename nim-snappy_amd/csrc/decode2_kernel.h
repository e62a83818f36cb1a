// decode2_kernel.h -- pass 2 of the v2 block decoder: build the block in LDS, flush it once.
//
// Semantics: decodeAllTags, snappy/decoder.nim:20-155.  One 256-thread workgroup per unit,
// two workgroups per CU (64 KiB output window + 14 KiB of staging each).  The index written by
// index_units_kernel tells every 32-byte region of the tag stream where its first element
// starts, where that element writes, and how many copy elements start in the region, so the
// stream is consumed 64 regions (2 KiB) per step with no serial parse:
//
//   wave 0  "front end": one region per lane; walks its few elements, writes literal payloads
//           straight into the output window (literals have no dependencies) and appends copy
//           elements, in stream order, to a list in LDS (slot = wave prefix sum of the
//           per-region copy counts).  The next 2 KiB of the stream and its index entries are
//           in flight from HBM while the current ones are processed.
//   waves 2,3 "far copies": a copy of the PREVIOUS step's list whose source ends below that
//           chunk's first output byte depends on nothing unresolved (all earlier lists are
//           done), so these two waves execute all such copies in parallel, alternate batches,
//           and cross them off the list (62 % of the copies of text, 44 % of html).
//   wave 1  "resolver": waits for them, then consumes what is left of the PREVIOUS step's list,
//           64 elements at a time.  Runs of
//           consecutive copies with one offset (how the encoder splits long matches,
//           encoder.nim:97-112) are merged back into one copy.  Copies are then resolved in
//           rounds against a high-water mark: everything below the destination of the first
//           unresolved copy is final, so every copy whose source ends below it can run
//           now, one per lane; the first unresolved one can always run.  Long or
//           self-overlapping copies are done by all 64 lanes.
//   all     flush the finished block with 16-byte stores.
//
// One workgroup barrier per step separates "list k is complete" from "list k is consumed".
//
// The inner loops are written branch-free: a lane that has nothing to store stores to a sink
// slot instead of branching around the store (a taken branch costs more than the store).
#pragma once

#include "common.h"
#include "index_kernel.h"

namespace snappy_hip {

constexpr uint32_t kD2Threads = 512;  // waves 0,1: front end; waves 2-7: resolvers
constexpr uint32_t kD2Pool = kD2Threads / 64 - 2;
// (a workgroup's waves are dealt round-robin to the 4 SIMDs: waves 0 and 4 share one, so the
// two front-end waves must not be 0 and 4)
constexpr uint32_t kD2Ring = 4096;
constexpr uint32_t kListCap = 704;   // elements per 2 KiB step that the wide-step (list) mode takes
constexpr uint32_t kElemCap = 1024;  // elements per 2 KiB step: the format's maximum (2 bytes each)
constexpr uint32_t kGroup = 256;     // output bytes one resolver wave handles per pass (4 per lane)
constexpr uint32_t kPendBits = 16384;  // window of the element-start bitmap (output positions)
constexpr uint32_t kPendWords = kPendBits / 32;
// byte mode keeps kElemCap u16 offsets in a list buffer and two waves' scratch behind them; one
// more wave's scratch goes into the (then unused) length array
static_assert((kListCap + 64) * 4 >= kElemCap * 2 + 2 * kGroup * 2, "list buffer too small for byte mode");
static_assert(kListCap + 64 >= kGroup * 2 && (kListCap + 64) % 8 == 0, "length array too small for scratch");
constexpr uint32_t kOutSink = kMaxBlockLen + 16; // 64 scratch dwords behind the output window,
constexpr uint32_t kOutAlloc = kMaxBlockLen + 16 + 256 + 16;  // one per lane (no bank conflicts)

struct Decode2Params {
  const uint8_t* in;
  const uint64_t* in_off;
  const uint32_t* in_len;
  uint8_t* out;
  const uint64_t* out_off;
  const uint32_t* out_len;  // from the index pass
  uint32_t* status;
  const uint64_t* idx_off;  // nullptr: u * idx_stride
  uint64_t idx_stride;
  const uint32_t* idx;
  uint64_t n_units;
  int unit;
  int dbg;  // timing experiments: 1 no literal payloads, 2 no resolver, 4 no walk, 8 no flush
  unsigned long long* stats;  // DEBUG counters (nullptr = off)
};

// value of lane-1 (0 for lane 0), without a trip through the LDS crossbar
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
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

// The 64 KiB output window is DYNAMIC shared memory (launch with kOutAlloc bytes): with the whole
// footprint declared statically the compiler derives "at most 3 waves per SIMD" from it and pads
// the kernel's VGPR allocation to 129+ to enforce that -- which leaves no room for the second
// workgroup of a CU whenever both put two waves on one SIMD (measured: one workgroup per CU).
extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn_window[];

__global__ __launch_bounds__(kD2Threads) void decode_indexed_kernel(Decode2Params prm) {
  uint8_t* const s_out = s_dyn_window;
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kD2Ring + 16];
  __shared__ __attribute__((aligned(8))) uint32_t s_cp[2][kListCap + 64];  // dst | offset << 16   (+64: sink slots)
  __shared__ __attribute__((aligned(8))) uint8_t s_cl[2][kListCap + 64];   // length 1..64, 0 = skip
  __shared__ uint32_t s_cnt[2];
  __shared__ uint32_t s_xdone[2];  // far-copy waves finished with list k
  __shared__ uint32_t s_near[2][2];  // near copies left in each half of list k after compaction
  __shared__ uint32_t s_mode[2];     // list k is resolved through the pending-byte bitmap
  __shared__ uint32_t s_sbase[kMaxFastIn / kChunk + 2];  // output position where step k starts
  // One bit per output byte (position mod kPendBits): set where an element of a step that is not
  // resolved yet starts.  Together with the per-step list of copy offsets this maps every
  // output byte to its element.
  __shared__ unsigned long long s_pend[kPendWords / 2];
  __shared__ uint32_t s_err;
  // pointer-doubling scratch of resolver waves 2-4; waves 5-7 use the parts of their step's list
  // buffers that byte mode leaves free
  __shared__ __attribute__((aligned(8))) uint16_t s_r16[3][kGroup];
  __shared__ uint32_t s_front;                 // every output byte below this position is final
  __shared__ uint16_t s_gidx[kPendBits / kGroup];  // list slot of the element that covers byte 256 m

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint64_t u = blockIdx.x;
  if (u >= prm.n_units) return;
  if (prm.status[u] != kOk) return;  // the index pass already decided this unit
  const uint32_t total = prm.out_len[u];
  if (total == 0) return;

  const uint8_t* in0 = prm.in + prm.in_off[u];
  uint32_t n = prm.in_len[u];
  if (prm.unit == kUnitRaw) {  // skip the varint (validated by the index pass)
    uint32_t hdr = 0;
    while (in0[hdr] & 0x80) hdr++;
    hdr++;
    in0 += hdr;
    n -= hdr;
  }
  uint8_t* gout = prm.out + prm.out_off[u];
  const uint32_t* idx = prm.idx + (prm.idx_off ? prm.idx_off[u] : u * prm.idx_stride);

  const uint32_t shift = (uint32_t)((uintptr_t)in0 & 15);
  const uint8_t* g0 = in0 - shift;
  const uint32_t q_end = (uint32_t)(((uint64_t)shift + n + 15) & ~15ull);
  const uint32_t n_chunks = (n + kChunk - 1) / kChunk;   // 2 KiB steps
  const uint32_t n_regions = (n + kSub - 1) / kSub;       // 16-byte index entries
  const bool fe = wave <= 1;                             // front-end waves
  const uint32_t half = wave == 1 ? 1 : 0;               // which 1 KiB of the step is mine

  if (tid == 0) {
    s_err = 0;
    s_cnt[0] = 0;
    s_cnt[1] = 0;
    s_xdone[0] = 0;
    s_xdone[1] = 0;
    s_mode[0] = 0;
    s_mode[1] = 0;
    s_front = 0;
  }
  for (uint32_t i = tid; i < kPendWords / 2; i += kD2Threads) s_pend[i] = 0;

  // ring[q & 4095] = stream byte q - shift; at the start of step s it holds q in
  // [2048 s, 2048 s + 4096)
  auto ring_store = [&](uint32_t q, uint4 v) {
    const uint32_t i = q & (kD2Ring - 1);
    *reinterpret_cast<uint4*>(s_ring + i) = v;
    if (i == 0) *reinterpret_cast<uint4*>(s_ring + kD2Ring) = v;
  };
  auto ring32 = [&](uint32_t q) -> uint32_t { return ld32u(s_ring + (q & (kD2Ring - 1))); };
  // Stores that a lane must not perform go to its private sink dword instead of being branched
  // around.  (One shared sink address would serialise the 64 lanes on one LDS bank.)
  const uint32_t sink = kOutSink + lane * 4;
  // store the low nb (0..4) bytes of v at s_out[at..]
  auto out_store_upto4 = [&](uint32_t at, uint32_t v, uint32_t nb) {
    st32u(s_out + (nb == 4 ? at : sink), v);
    const bool part = nb < 4;
    s_out[part && nb > 0 ? at : sink] = (uint8_t)v;
    s_out[part && nb > 1 ? at + 1 : sink + 1] = (uint8_t)(v >> 8);
    s_out[part && nb > 2 ? at + 2 : sink + 2] = (uint8_t)(v >> 16);
  };
  // store the first len (0..16) bytes of v[0..3] at s_out[at..]: 4 dword + 3 byte stores
  auto out_store_upto16 = [&](uint32_t at, const uint32_t* v, uint32_t len) {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) st32u(s_out + (len >= 4 * k + 4 ? at + 4 * k : sink), v[k]);
    const uint32_t t0 = len & 12, r = len & 3;  // tail: r bytes at at+t0 (t0 = 16 needs none)
    const uint32_t tv = t0 == 0 ? v[0] : (t0 == 4 ? v[1] : (t0 == 8 ? v[2] : v[3]));
    const bool tl = len < 16;
    s_out[tl && r > 0 ? at + t0 : sink] = (uint8_t)tv;
    s_out[tl && r > 1 ? at + t0 + 1 : sink + 1] = (uint8_t)(tv >> 8);
    s_out[tl && r > 2 ? at + t0 + 2 : sink + 2] = (uint8_t)(tv >> 16);
  };

  // Copy L (0 = nothing, <= 64) bytes to s_out[dst..].  rd(k) returns the k-th ALIGNED dword of
  // the source counted from the dword that holds its first byte; sh = source address & 3.
  // Unaligned LDS dword accesses cost ~10-20x an aligned one on gfx950 (tools/probes/
  // lds_rates.hip), so sources are read as aligned dwords and re-aligned with a funnel shift,
  // and the destination is written bytewise.
  auto lean_copy_pre = [&](uint32_t dst, uint32_t s0, uint32_t s1, uint32_t s2, auto rd, uint32_t sh,
                           uint32_t L) {
    const uint32_t sh8 = sh * 8;
    uint32_t v0 = __funnelshift_r(s0, s1, sh8), v1 = __funnelshift_r(s1, s2, sh8);
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) s_out[L > j ? dst + j : sink + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) s_out[L > 4 + j ? dst + 4 + j : sink + j] = (uint8_t)(v1 >> (8 * j));
    for (uint32_t k = 8; ballot(L > k); k += 8) {  // longer elements: 8 more bytes per trip
      s0 = s2;
      s1 = rd(k / 4 + 1);
      s2 = rd(k / 4 + 2);
      v0 = __funnelshift_r(s0, s1, sh8);
      v1 = __funnelshift_r(s1, s2, sh8);
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) s_out[L > k + j ? dst + k + j : sink + j] = (uint8_t)(v0 >> (8 * j));
#pragma unroll
      for (uint32_t j = 0; j < 4; j++)
        s_out[L > k + 4 + j ? dst + k + 4 + j : sink + j] = (uint8_t)(v1 >> (8 * j));
    }
  };
  auto lean_copy = [&](uint32_t dst, auto rd, uint32_t sh, uint32_t L) {
    lean_copy_pre(dst, rd(0u), rd(1u), rd(2u), rd, sh, L);
  };
  auto out_al = [&](uint32_t a) -> uint32_t {  // aligned dword that holds s_out[a]
    return *reinterpret_cast<const uint32_t*>(s_out + (a & ~3u));
  };
  auto ring_al = [&](uint32_t q) -> uint32_t {  // aligned dword that holds stream byte q - shift
    return *reinterpret_cast<const uint32_t*>(s_ring + (q & (kD2Ring - 1) & ~3u));
  };
  // pending-byte bitmap as dwords; a run of len bits at pos touches up to three of them
  uint32_t* const pw = reinterpret_cast<uint32_t*>(s_pend);
  auto bits_make = [&](uint32_t pos, uint32_t len, uint32_t* d0, uint32_t* d1, uint32_t* d2) {
    const uint32_t sh = pos & 31;
    const unsigned long long m = len >= 64 ? ~0ull : ((1ull << len) - 1);
    const unsigned long long lo = m << sh;
    *d0 = (uint32_t)lo;
    *d1 = (uint32_t)(lo >> 32);
    *d2 = sh ? (uint32_t)(m >> (64 - sh)) : 0;
  };

  if (wave == 1) {  // prologue: first 4 KiB of the stream
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = (lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, *reinterpret_cast<const uint4*>(g0 + q));
    }
  }
  __syncthreads();

  // Index entries of the NEXT step (mine and the other front-end wave's) are fetched at the start
  // of a step and consumed at the start of the next one, before anything younger is issued, so
  // the wait for them never also waits for fresh loads.  Entries past the end read as "none".
  const uint32_t none_end = kIdxNone | (total << 11);
  auto idx_at = [&](uint32_t r) -> uint32_t { return r < n_regions ? idx[r] : none_end; };
  // (every wave executes these loads -- straight-line code keeps the compiler from copying the
  // loaded registers, and thereby waiting for them, at the end of the front-end branch)
  uint32_t ie_pref = idx_at(half * 64 + lane);
  uint32_t io_pref = idx_at((1 - half) * 64 + lane);
  for (uint32_t k = tid; k <= n_chunks && k < kMaxFastIn / kChunk + 2; k += kD2Threads)
    s_sbase[k] = k < n_chunks ? idx_at(k * 128) >> 11 : total;
  __syncthreads();
  uint32_t cprev = 0;  // output position where the previous step starts
  uint4 pre[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  uint32_t pq[2] = {0xffffffffu, 0xffffffffu};  // ring data in flight (wave 1)

  unsigned long long tm_pre = 0, tm_walk = 0, tm_post = 0, tm_bar = 0;
  uint32_t acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0;  // DEBUG counters, flushed once per wave
  for (uint32_t s = 0; s <= n_chunks; s++) {
    if (s_err) break;  // set before the last barrier: every wave sees it here
    const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
    unsigned long long tm1 = tm0, tm2 = tm0;
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
    ie_pref = idx_at((s + 1) * 128 + half * 64 + lane);
    io_pref = idx_at((s + 1) * 128 + (1 - half) * 64 + lane);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      pq[i] = s * kChunk + kD2Ring + (lane + 64 * i) * 16;
      const uint32_t qc = pq[i] < q_end ? pq[i] : 0;  // clamped: always a valid address
      pre[i] = *reinterpret_cast<const uint4*>(g0 + qc);
    }

    if (fe && s < n_chunks) {
      // =================================== front end ===========================================
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
      ctot += otot;            // elements of the whole step
      // output position where this step starts / the next one starts
      const uint32_t cbase = s_sbase[s], cnext = s_sbase[s + 1];
      // byte mode: this step and the previous one fit the window of the start bitmap
      const bool bm = cnext - cprev <= kPendBits - kGroup;
      // list mode (wide steps = long elements) has a smaller list; denser data than it takes
      // cannot also be wide, but a hostile stream is handed to the one-pass kernel
      const bool dense = !bm && ctot > kListCap;
      if (wave == 0 && lane == 0) {
        s_cnt[buf] = dense ? 0 : ctot;
        s_xdone[buf] = 0;
        s_mode[buf] = bm ? 1 : 0;
        if (dense) s_err = 2;
      }
      if (dense) slot = kListCap;  // all appends of this step go to the sink slots
      cprev = cbase;
      uint16_t* const o16 = reinterpret_cast<uint16_t*>(s_cp[buf]);  // byte mode: offset per element
      uint16_t* const sink16 = reinterpret_cast<uint16_t*>(s_out + sink);

      const uint32_t rs = c0 + lane * kSub;
      const uint32_t r_end = rs + kSub < n ? rs + kSub : n;
      uint32_t pos = rs + e_off;
      bool live = had && pos < n && !(prm.dbg & 4) && !((prm.dbg & 16) && half == 1) && !((prm.dbg & 32) && half == 0);
      bool big = false;  // a literal longer than 64 bytes ends my region: done below
      uint32_t big_dst = 0, big_len = 0, big_src = 0, big_slot = 0;
      bool bad = false;
      uint32_t st_trips = 0;
      tm1 = __builtin_amdgcn_s_memtime();
      while (ballot(live)) {
        st_trips++;
        const uint32_t q = pos + shift;
        const uint32_t t0 = ring_al(q), t1 = ring_al(q + 4), t2 = ring_al(q + 8);
        const uint32_t w0 = __funnelshift_r(t0, t1, (q & 3) * 8), w1 = __funnelshift_r(t1, t2, (q & 3) * 8);
        const uint32_t b14 = (w0 >> 8) | (w1 << 24);
        bool is_copy;
        uint32_t L, size, hdr, off;
        decode_fast(w0 & 0xff, b14, &is_copy, &L, &size, &hdr, &off);
        const bool cpy = live && is_copy;
        const bool lit = live && !is_copy;
        const bool bad_off = cpy && (off == 0 || off > dst);  // decoder.nim:112
        bad = bad || bad_off;
        if (bm) {
          // ---- byte mode: offset (0 = literal) at the element's slot, start bit at its first
          // output byte, and its slot at the 256-byte boundary it covers (if any) ----------------
          *((live && slot < kElemCap) ? o16 + slot : sink16) = (cpy && !bad_off) ? (uint16_t)off : (uint16_t)0;
          atomicOr(pw + ((dst & (kPendBits - 1)) >> 5), live ? 1u << (dst & 31) : 0u);
          const uint32_t mb = (dst + kGroup - 1) / kGroup;
          const bool covers = live && mb * kGroup < dst + L;
          *(covers ? s_gidx + (mb & (kPendBits / kGroup - 1)) : sink16 + 1) = (uint16_t)slot;
        } else {
          // ---- list mode: (dst, offset, length) per element; literals as length 0 ---------------
          const uint32_t sl = (live && slot < kListCap) ? slot : kListCap + lane;
          s_cp[buf][sl] = dst | (off << 16);
          // bit 7: source ends below this chunk's output = independent of every unresolved copy
          const uint32_t far = (dst - off + L <= cbase) ? 0x80u : 0u;
          s_cl[buf][sl] = (cpy && !bad_off) ? (uint8_t)(L | far) : (uint8_t)0;  // 0 = nothing to do
        }
        const uint32_t my_slot = slot;
        slot += live ? 1 : 0;
        // ---- literal: payload of up to 16 bytes here, up to 64 in the rare loop below -----------
        const uint32_t qs = q + hdr;
        const uint32_t Lw = (lit && L <= 64 && !(prm.dbg & 1)) ? L : 0;  // bytes this lane writes
        lean_copy(dst, [&](uint32_t k) { return ring_al(qs + 4 * k); }, qs & 3, Lw);
        if (lit && L > 64) {
          big = true;
          big_dst = dst;
          big_len = L;
          big_src = pos + hdr;
          big_slot = my_slot;
        }
        dst += live ? L : 0;
        pos += live ? size : 0;
        live = live && pos < r_end;
      }
      tm2 = __builtin_amdgcn_s_memtime();
      // long literals: whole wave, straight from HBM (at most one per region)
      uint64_t bigs = ballot(big);
      while (bigs) {
        const uint32_t e = ctz64(bigs);
        bigs &= bigs - 1;
        const uint32_t eL = readlane(big_len, e);
        const uint32_t ed = readlane(big_dst, e);
        const uint32_t es = readlane(big_src, e);
        if (bm) {  // every 256-byte boundary the literal covers maps to its slot
          const uint32_t eslot = readlane(big_slot, e);
          for (uint32_t m = (ed + kGroup - 1) / kGroup + lane; m * kGroup < ed + eL; m += 64)
            s_gidx[m & (kPendBits / kGroup - 1)] = (uint16_t)eslot;
        }
        // 16 bytes per lane and pass, four passes in flight
        const uint32_t body = eL & ~15u;
        for (uint32_t i = lane * 16; i < body; i += 4 * 1024) {
          uint4 v[4];
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (i + j * 1024 < body) __builtin_memcpy(&v[j], in0 + es + i + j * 1024, 16);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (i + j * 1024 < body) {
              st32u(s_out + ed + i + j * 1024, v[j].x);
              st32u(s_out + ed + i + j * 1024 + 4, v[j].y);
              st32u(s_out + ed + i + j * 1024 + 8, v[j].z);
              st32u(s_out + ed + i + j * 1024 + 12, v[j].w);
            }
          }
        }
        if (body + lane < eL) s_out[ed + body + lane] = in0[es + body + lane];
      }
      if (ballot(bad) && lane == 0) s_err = 1;
      acc_a += st_trips;
      acc_b += 1;
      acc_c += bm ? 1 : 0;
    } else if (wave >= 2 && s >= 1 && s_mode[(s - 1) & 1] && !(prm.dbg & 2)) {
      // =================================== byte-mode resolver ======================================
      // The three waves take the 256-byte groups of the previous step's output in turn, one
      // aligned dword per lane.  A byte finds its element by counting start bits from the group's
      // first byte (whose element the front end recorded in s_gidx); a copy byte's source is
      // "own position - offset".  Sources inside the group are followed to a byte that is final
      // (pointer doubling through a small LDS array); sources below the group must lie under the
      // frontier s_front, which the groups advance strictly in order.
      const uint32_t buf = (s - 1) & 1;
      const uint32_t cb = s_sbase[s - 1], cn = s_sbase[s];
      const uint16_t* const o16 = reinterpret_cast<const uint16_t*>(s_cp[buf]);
      uint16_t* const r16 = wave < 5 ? s_r16[wave - 2]
                            : wave < 7 ? reinterpret_cast<uint16_t*>(s_cp[buf]) + kElemCap + (wave - 5) * kGroup
                                       : reinterpret_cast<uint16_t*>(s_cl[buf]);
      auto cbar = [] { asm volatile("" ::: "memory"); };
      uint32_t front = cb;  // what I know of s_front
      for (uint32_t g = (cb & ~(kGroup - 1)) + (wave - 2) * kGroup; g < cn; g += kD2Pool * kGroup) {
        acc_c++;
        const uint32_t p = g + 4 * lane;
        const uint32_t wbits = pw[(p & (kPendBits - 1)) >> 5];
        front = __hip_atomic_load(&s_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // my bytes that belong to this step: [lo, hi) of 0..4
        const uint32_t lo = p >= cb ? 0 : (cb - p < 4 ? cb - p : 4);
        const uint32_t hi = p + 4 <= cn ? 4 : (cn > p ? cn - p : 0);
        const uint32_t rmask = ((1u << hi) - 1) & ~((1u << lo) - 1);
        const uint32_t sb = (wbits >> (p & 31)) & rmask;
        // the element that covers byte g is E0; a start bit AT g is that element itself
        const uint32_t cbits = sb & ~((lane == 0 && g >= cb) ? 1u : 0u);
        const uint32_t E0 = g > cb ? (uint32_t)s_gidx[(g / kGroup) & (kPendBits / kGroup - 1)]
                                   : (g == cb ? 0u : 0xffffffffu);
        uint32_t tot;
        const uint32_t excl = wave_excl_scan((uint32_t)__builtin_popcount(cbits), lane, &tot);
        uint32_t sp[4];
        bool cp[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const bool in = (rmask >> j) & 1;
          const uint32_t idx = E0 + excl + (uint32_t)__builtin_popcount(cbits & ((2u << j) - 1));
          const uint32_t off = o16[in ? idx : 0];
          cp[j] = in && off != 0;
          sp[j] = p + j - (cp[j] ? off : 0);
        }
        // ---- sources inside my own group: follow them to a final byte ----------------------------
        if (ballot((cp[0] && sp[0] >= g) || (cp[1] && sp[1] >= g) || (cp[2] && sp[2] >= g) ||
                   (cp[3] && sp[3] >= g))) {
          for (uint32_t it = 0; it < 10; it++) {
            acc_b++;
            // every byte publishes its pointer (final bytes point to themselves) and takes over
            // the pointer of the byte it points to: the chain length halves
            cbar();
            *reinterpret_cast<uint2*>(r16 + 4 * lane) = make_uint2(sp[0] | (sp[1] << 16), sp[2] | (sp[3] << 16));
            cbar();
            bool changed = false;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
              const bool dep = cp[j] && sp[j] >= g;
              const uint32_t t = r16[dep ? sp[j] - g : 0];
              changed = changed || (dep && t != sp[j]);
              sp[j] = dep ? t : sp[j];
            }
            cbar();
            if (!ballot(changed)) break;
          }
        }
        // ---- my start bits are consumed (every lane's read of them was issued above) -----------------
        if ((lane & 7) == 0) {
          const uint32_t l32 = p >= cb ? 0 : (cb - p < 32 ? cb - p : 32);
          const uint32_t h32 = p + 32 <= cn ? 32 : (cn > p ? cn - p : 0);
          const uint32_t m32 = (uint32_t)(((1ull << h32) - 1) & ~((1ull << l32) - 1));
          atomicAnd(pw + ((p & (kPendBits - 1)) >> 5), ~m32);
        }
        // ---- gather early; bytes whose source was not final yet are fetched again below ------------
        const bool stale = (cp[0] && sp[0] < g && sp[0] >= front) || (cp[1] && sp[1] < g && sp[1] >= front) ||
                           (cp[2] && sp[2] < g && sp[2] >= front) || (cp[3] && sp[3] < g && sp[3] >= front);
        cbar();
        uint32_t v = (uint32_t)s_out[sp[0]] | ((uint32_t)s_out[sp[1]] << 8) | ((uint32_t)s_out[sp[2]] << 16) |
                     ((uint32_t)s_out[sp[3]] << 24);
        cbar();
        // ---- my turn: every group below mine has published, i.e. everything below g is final ------
        const uint32_t expect = g > cb ? g : cb;
        for (uint32_t spin = 0; front != expect; spin++) {
          acc_a++;
          if ((spin & 1023) == 1023 && (spin > 400000 || s_err != 0)) {
            // cannot happen on a consistent index; never hang the GPU
            if (lane == 0) atomicOr(&s_err, 4u);
            break;
          }
          front = __hip_atomic_load(&s_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          cbar();
        }
        if (ballot(stale)) {
          acc_d++;
          v = (uint32_t)s_out[sp[0]] | ((uint32_t)s_out[sp[1]] << 8) | ((uint32_t)s_out[sp[2]] << 16) |
              ((uint32_t)s_out[sp[3]] << 24);
        }
        const bool anyc = cp[0] || cp[1] || cp[2] || cp[3];
        const bool full = rmask == 15;  // the whole dword is this step's: its other bytes are final
        *reinterpret_cast<uint32_t*>(s_out + ((full && anyc) ? p : sink)) = v;
        if (ballot(!full && anyc)) {  // the step's first and last dword: bytewise
#pragma unroll
          for (uint32_t j = 0; j < 4; j++) s_out[(!full && cp[j]) ? p + j : sink + j] = (uint8_t)(v >> (8 * j));
        }
        front = g + kGroup < cn ? g + kGroup : cn;
        cbar();
        if (lane == 0) __hip_atomic_store(&s_front, front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if ((wave == 3 || wave == 4) && s >= 1 && !s_mode[(s - 1) & 1] && !(prm.dbg & 2)) {
      // =================================== far copies ============================================
      const uint32_t buf = (s - 1) & 1;
      const uint32_t count = s_cnt[buf];
      // each of the two waves owns one contiguous half of the list: it executes the far copies
      // and compacts the remaining (near) ones in place, in order, at the front of its half
      const uint32_t half = ((count + 127) / 128) * 64;
      const uint32_t seg0 = (wave - 3) * half;
      const uint32_t seg1 = seg0 + half < count ? seg0 + half : count;
      uint32_t wpos = seg0;
      for (uint32_t b0 = seg0; b0 < seg1; b0 += 64) {
        const uint32_t i = b0 + lane;
        const uint32_t lf = i < seg1 ? s_cl[buf][i] : 0;
        const uint32_t e = s_cp[buf][i];
        const bool far = (lf & 0x80) != 0;
        const bool near = lf != 0 && !far;
        const uint32_t len = lf & 0x7f;
        const uint32_t dst = e & 0xffff;
        const uint32_t src = far ? dst - (e >> 16) : 0;
        lean_copy(dst, [&](uint32_t k) { return out_al(src + 4 * k); }, src & 3, far ? len : 0);
        const uint64_t nm = ballot(near);
        const uint32_t rank = (uint32_t)__builtin_popcountll(nm & ((1ull << lane) - 1));
        wave_fence();  // all reads of this batch are done before its slots are reused
        const uint32_t to = near ? wpos + rank : kListCap + lane;
        s_cp[buf][to] = e;
        s_cl[buf][to] = (uint8_t)len;
        wpos += (uint32_t)__builtin_popcountll(nm);
      }
      if (lane == 0) s_near[buf][wave - 3] = wpos - seg0;
      wave_fence();
      if (lane == 0) atomicAdd(&s_xdone[buf], 1u);
    } else if (wave == 2 && s >= 1 && !s_mode[(s - 1) & 1] && !(prm.dbg & 2)) {
      // =================================== resolver ==============================================
      for (uint32_t spin = 0; __hip_atomic_load(&s_xdone[(s - 1) & 1], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP) < 2 && spin < 400000; spin++)
        __builtin_amdgcn_s_sleep(2);
      wave_fence();
      const uint32_t buf = (s - 1) & 1;
      const uint32_t half = ((s_cnt[buf] + 127) / 128) * 64;
      for (uint32_t seg = 0; seg < 2; seg++)
      for (uint32_t b0 = seg * half, count = seg * half + s_near[buf][seg]; b0 < count; b0 += 64) {
        const uint32_t i = b0 + lane;
        const uint32_t len = i < count ? (s_cl[buf][i] & 0x7fu) : 0;
        const uint32_t e = s_cp[buf][i];
        const bool act = len != 0;
        const uint32_t dst = e & 0xffff, off = e >> 16;
        const uint32_t src = act ? dst - off : 0;
        // merge runs: same offset, destination continues the previous copy
        const uint32_t p_e = lane_prev(e), p_len = lane_prev(len);
        const bool cont = act && p_len != 0 && (p_e >> 16) == off && (p_e & 0xffff) + p_len == dst;
        const uint64_t heads = ballot(act && !cont);
        uint32_t mlen = len;
        if (ballot(cont)) {  // rare on text, the rule on run-like data
          uint32_t tot;
          const uint32_t excl = wave_excl_scan(len, lane, &tot);
          const uint64_t later = lane == 63 ? 0 : heads & ~((2ull << lane) - 1);
          const uint32_t nxt = later ? ctz64(later) : 64;
          const uint32_t nxt_excl = __shfl(excl, nxt & 63, 64);
          mlen = (nxt == 64 ? tot : nxt_excl) - excl;  // merged length (meaningful on heads)
        }
        // per-lane path: short and not self-overlapping
        const bool simple = mlen <= 16 && off >= mlen;
        const uint64_t simple_m = ballot(simple);

        uint64_t pending = heads;
        while (pending) {
          const uint32_t first = ctz64(pending);
          const uint32_t fd = readlane(dst, first);  // everything below fd is final
          if (!((simple_m >> first) & 1)) {
            // ---- long or self-overlapping copy: the whole wave -------------------------------
            const uint32_t fL = readlane(mlen, first);
            const uint32_t foff = readlane(off, first);
            const uint32_t fs = fd - foff;
            if (fL <= 256 && foff >= fL) {  // one pass, no overlap
              const uint32_t at = lane * 4;
              const uint32_t v = ld32u(s_out + fs + (at < fL ? at : 0));
              out_store_upto4(fd + at, v, fL > at ? (fL - at < 4 ? fL - at : 4) : 0);
            } else {
              uint32_t done = 0, period = foff;
              if (foff < 256) {
                // overlap: out[fd+i] = out[fs + i mod foff].  Bytewise for the first 512 bytes;
                // after that a multiple of the period that is >= 256 is the copy distance.
                const uint32_t head_len = fL < 512 ? fL : 512;
                uint32_t j = lane % foff;  // (lane + 64 t) mod foff, kept incrementally
                const uint32_t inc = 64 % foff;
                for (uint32_t i = lane; i < head_len; i += 64) {
                  s_out[fd + i] = s_out[fs + j];
                  j += inc;
                  j = j >= foff ? j - foff : j;
                }
                wave_fence();
                done = head_len;
                period = foff * ((255 + foff) / foff);  // multiple of foff in [256, 511]
              }
              for (uint32_t i = done + lane * 4; i < fL; i += 256) {
                // period >= 256: the 256 bytes of one pass only read bytes of earlier passes
                const uint32_t v = ld32u(s_out + fd + i - period);
                out_store_upto4(fd + i, v, fL - i < 4 ? fL - i : 4);
                wave_fence();
              }
            }
            pending &= pending - 1;
            wave_fence();
            continue;
          }
          // ---- short copies: one per lane, everything whose source is final ------------------
          const bool ready = ((pending >> lane) & 1) && simple && (lane == first || src + mlen <= fd);
          lean_copy(dst, [&](uint32_t k) { return out_al(src + 4 * k); }, src & 3, ready ? mlen : 0);
          pending &= ~ballot(ready);
          wave_fence();
        }
      }
      if (lane == 0) s_front = s_sbase[s];  // what byte mode expects to find
    }
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    // Workgroup barrier for LDS traffic only: __syncthreads() would also drain vmcnt, i.e. wait
    // for the global prefetches that are meant to stay in flight across the barrier.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long tm4 = __builtin_amdgcn_s_memtime();
    if (wave == 0 || wave == 2) {
      tm_pre += tm1 - tm0;
      tm_walk += tm2 - tm1;
      tm_post += tm3 - tm2;
      if (wave == 2 && s >= 1 && !s_mode[(s - 1) & 1]) {  // list-mode share of the resolver's time
        tm_walk += tm3 - tm2;
        tm_pre += 1;
      }
      tm_bar += tm4 - tm3;
    }
  }
  if (prm.stats && lane == 0 && wave >= 2) {
    atomicAdd(&prm.stats[0], (unsigned long long)acc_a);
    atomicAdd(&prm.stats[1], (unsigned long long)acc_b);
    atomicAdd(&prm.stats[2], (unsigned long long)acc_c);
    atomicAdd(&prm.stats[3], (unsigned long long)acc_d);
  }
  if (prm.stats && tid == 0) {
    atomicAdd(&prm.stats[4], (unsigned long long)acc_a);
    atomicAdd(&prm.stats[5], (unsigned long long)acc_b);
    atomicAdd(&prm.stats[6], (unsigned long long)acc_c);
  }
  if (prm.stats && tid == 128) {
    atomicAdd(&prm.stats[11], tm_post);  // pool wave: work
    atomicAdd(&prm.stats[12], tm_bar);   // pool wave: waiting at the step barrier
    atomicAdd(&prm.stats[13], tm_walk);  // pool wave: work in list-mode steps
    atomicAdd(&prm.stats[14], tm_pre);   // list-mode steps
  }
  if (prm.stats && tid == 0) {
    atomicAdd(&prm.stats[7], tm_pre);
    atomicAdd(&prm.stats[8], tm_walk);
    atomicAdd(&prm.stats[9], tm_post);
    atomicAdd(&prm.stats[10], tm_bar);
  }

  // ---- flush ------------------------------------------------------------------------------------
  if (s_err) {
    if (tid == 0) prm.status[u] = (s_err & 1) ? kInvalidInput : kNeedsOnePass;
    return;
  }
  if (prm.dbg & 8) return;
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
