// split_kernels.h -- where do the 64 KiB blocks of ONE raw Snappy buffer start?
//
// uncompress() of a buffer that decodes to more than one block (snappy.nim:84-110): the stream has
// one varint and no block delimiters (snappy.nim:49-62), so block k's first element can only be
// found by walking the tags (decoder.nim:39-109) -- a chain through the whole stream.  The chain is
// broken up by guessing: the stream is cut into segments of kSplitSeg bytes, one LANE per segment
// walks the elements of its segment from where it believes the chain enters it, and tells the
// segments behind it where the chain leaves.  Segment 0's entry is known; every other segment
// starts with the guess "at my first byte".  A walk that starts wrong falls into step with the real
// chain after a few elements (elements are short), so after a few rounds most segments are right,
// and the rounds repeat until no entry changes -- which, since segment 0 is right and every segment
// takes its entry from the EARLIEST segment that claims it, is exactly the state where all are right
// (an element that jumps over segments, a long literal, writes the entry of every segment it
// covers).  Then a prefix sum over the segments' output bytes places them in the output, and a last
// walk records, for every 64 KiB boundary, the stream position of the element that starts there --
// or that an element straddles it (a foreign encoder: the caller falls back to the serial walk).
// All input-side checks of decodeAllTags are made by the walk (decode_element, index_kernel.h).
#pragma once

#include "common.h"
#include "index_kernel.h"

namespace snappy_hip {

constexpr uint32_t kSplitSeg = 256;      // stream bytes per lane
// Segments one element may claim: a literal of a whole 64 KiB block and its length bytes.  (A walk
// that starts wrong reads payload bytes as tags, and one byte in fifty is the tag of a literal with
// explicit length: unbounded, such claims would keep overriding the right ones far downstream.
// Streams with longer literals converge slowly or not at all within the round limit: serial walk.)
constexpr uint32_t kSplitMaxCover = 65536 / kSplitSeg + 1;
constexpr uint32_t kSplitClean = 16;  // native elements in a row that make a walk trusted (see the kernel)
// Follow-through (see the end of split_walk_kernel): literals at least this long, at most so many in a row
constexpr uint32_t kSplitFollowMin = 1024;
constexpr uint32_t kSplitFollowMax = 1u << 16;

struct SplitParams {
  const uint8_t* in;      // the tag stream (behind the varint)
  uint32_t n;             // its length
  uint32_t nseg;
  const unsigned long long* nxt_in;  // [nseg] (writer segment << 32) | entry position; ~0: nobody claimed it
  unsigned long long* nxt_out;
  uint32_t* prev;         // [nseg] the entry used in the previous round (bit 31: it was a claimed one)
  uint32_t* outb;         // [nseg] output bytes of the elements that start in the segment
  uint32_t* memo;         // [nseg] the last walk's exit ([0:31), all ones = invalid element) and trust (bit 31)
  uint32_t* follow;       // [nseg] the entry from which somebody has walked this segment's chain of long literals
  uint32_t* changed;      // entries that changed this round
  uint32_t* flags;        // [1] invalid element met, [2] an element straddles a 64 KiB boundary of the output
  // locate pass
  const uint64_t* out_at; // [nseg + 1] exclusive prefix sum of outb
  uint32_t* blk_in;       // [nblk] stream position where output block k starts
  int locate;
};

// One wave per workgroup: its 64 segments are 64 KiB of stream, staged in LDS with coalesced loads
// before the lanes walk them (64 lanes reading their own KiB byte by byte straight from memory move a
// cache line per element and lane: 1.4 GB of L2 traffic for a round over 32 MiB of stream).
constexpr uint32_t kSplitWg = 64;
constexpr uint32_t kSplitStage = kSplitWg * kSplitSeg + 16;  // + the bytes an element's header may reach over
extern __shared__ __attribute__((aligned(16))) uint8_t s_split_dyn[];

// the element at p (stage: the workgroup's range of the stream, from stream position lo): false = invalid
// (decoder.nim:54-57, :67-68, :77-79, truncated copies)
__device__ __forceinline__ bool split_element(const uint8_t* stage, uint32_t lo, uint32_t n, uint32_t p, uint32_t* L,
                                              uint32_t* size, uint32_t* tag_out) {
  const uint8_t* q = stage + (p - lo);
  uint32_t b[5];
#pragma unroll
  for (uint32_t i = 0; i < 5; i++) b[i] = q[i];  // (staged bytes behind the stream's end are zero)
  *tag_out = b[0];
  return decode_element_bf(b[0], b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24), n - p - 1, L, size);
}

__global__ __launch_bounds__(kSplitWg) void split_walk_kernel(SplitParams p) {
  const uint32_t s = blockIdx.x * kSplitWg + threadIdx.x;
  const uint32_t wg_lo = blockIdx.x * kSplitWg * kSplitSeg;
  uint8_t* const stage = s_split_dyn;
  const bool live = s < p.nseg;  // (lanes behind the last segment only help with the staging)
  const uint32_t seg_lo = s * kSplitSeg;
  const uint32_t seg_hi = seg_lo + kSplitSeg < p.n ? seg_lo + kSplitSeg : p.n;
  uint32_t e = seg_lo;  // the guess
  bool claimed = s == 0;
  if (!live) {
    e = 0xffffffffu;
  } else if (p.locate) {
    e = p.prev[s] & 0x7fffffffu;  // (the entries of the last round)
  } else if (s == 0) {
    e = 0;
  } else {
    const unsigned long long k = p.nxt_in[s];
    if (k != ~0ull) {
      e = (uint32_t)k;
      claimed = true;
    }
  }
  bool same = !live;
  if (!p.locate && live) {
    // (bit 31: the entry was claimed -- a walk from a claimed entry starts trusted, so a guess that
    // turns into a claim of the same position is a different walk)
    const uint32_t pw = e | (claimed ? 0x80000000u : 0u);
    same = pw == p.prev[s];
    const uint64_t ch = __ballot(!same);  // (one atomic per wave, not per lane)
    if (ch && threadIdx.x == (uint32_t)__builtin_ctzll(ch)) atomicAdd(p.changed, (uint32_t)__builtin_popcountll(ch));
    p.prev[s] = pw;
    const_cast<unsigned long long*>(p.nxt_in)[s] = ~0ull;  // (mine to reset: this buffer is written again next round)
  }
  // Trust: a walk that started wrong reads payload bytes as tags, and one payload byte in four looks
  // like a copy4 -- an element no block encoder writes (nor literal tags 62/63: encoder.nim:44-125).
  // A walk may claim segments beyond its neighbour (the target of a long literal) only with kSplitClean
  // "native" elements in a row behind it: its far claims would otherwise keep overriding right ones
  // downstream.  A walk from a claimed entry starts trusted, one from a guess does not.
  uint32_t pos = e, out = 0, clean = claimed ? kSplitClean : 0;
  uint64_t op = (p.locate && live) ? p.out_at[s] : 0;
  bool bad = false;
  if (__ballot(!same && e < seg_hi)) {  // somebody walks: stage the workgroup's 64 KiB (+16) of stream
    for (uint32_t i = threadIdx.x * 16; i < kSplitStage; i += kSplitWg * 16) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint64_t g = (uint64_t)wg_lo + i;
      if (g + 16 <= p.n) {
        __builtin_memcpy(&v, p.in + g, 16);
      } else {
        uint8_t t[16] = {0};
        for (uint32_t k = 0; k < 16 && g + k < p.n; k++) t[k] = p.in[g + k];
        __builtin_memcpy(&v, t, 16);
      }
      *reinterpret_cast<uint4*>(stage + i) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (live && same) {  // same entry as last round: same walk, same result
    const uint32_t m = p.memo[s];
    pos = m & 0x7fffffffu;
    clean = (m >> 31) ? kSplitClean : 0;
    bad = pos == 0x7fffffffu;
  } else if (live) {
    while (pos < seg_hi) {
      uint32_t L, size, tg;
      if (!split_element(stage, wg_lo, p.n, pos, &L, &size, &tg)) {
        bad = true;
        break;
      }
      if (p.locate) {
        if ((op & 0xffffu) == 0) p.blk_in[op >> 16] = pos;
        else if (((op + L - 1) >> 16) != (op >> 16)) p.flags[2] = 1;  // crosses a 64 KiB boundary of the output
        op += L;
      }
      clean = ((tg & 3) == 3 || ((tg & 3) == 0 && (tg >> 2) >= 62)) ? 0 : clean + 1;
      out += L;
      pos += size;
    }
  }
  if (p.locate) {  // (uniform)
    if (live && bad && e < seg_hi) p.flags[1] = 1;
    return;
  }
  if (live && !same) {
    p.outb[s] = e < seg_hi ? out : 0;
    p.memo[s] = bad ? 0x7fffffffu : (pos | (clean >= kSplitClean ? 0x80000000u : 0u));
  }
  // (no lane leaves before the end: the follow-through below is the whole wave's work)
  const bool claims = live && !bad && e < seg_hi;  // (a segment the chain passes over claims nothing)
  if (claims) {
    // the chain leaves at pos: that is the entry of every segment up to the one that holds it
    uint32_t t_hi = pos / kSplitSeg;
    t_hi = t_hi < p.nseg - 1 ? t_hi : p.nseg - 1;
    t_hi = t_hi < s + kSplitMaxCover ? t_hi : s + kSplitMaxCover;
    if (clean < kSplitClean) t_hi = t_hi < s + 1 ? t_hi : s + 1;
    const unsigned long long key = ((unsigned long long)s << 32) | pos;
    for (uint32_t t = s + 1; t <= t_hi; t++) atomicMin(&p.nxt_out[t], key);
  }
  // Follow-through.  Long literals back to back (incompressible blocks: one literal of 64 KiB each) are a
  // chain that would advance ONE literal per round -- the segment a literal ends in learns its entry from
  // the walk of the segment the literal starts in.  So a trusted fresh walk that lands on a long literal
  // does that segment's walk as well (it is that one element), in that segment's name -- the same keys its
  // own lane writes from the next round on -- and goes on while it keeps landing on long literals: one
  // trip to memory for the literal's header per hop, the claims are written by the whole wave.
  // `follow` keeps later walks from doing the same chain again.
  bool fol = claims && !same && clean >= kSplitClean && p.follow[s] != e;
  uint32_t from = s, p0 = pos, hops = 0;
  while (__ballot(fol)) {
    uint32_t T = 0, T1 = 0, p1 = 0;
    bool go = false;
    if (fol) {
      T = p0 / kSplitSeg;
      // (beyond the cover T's entry was not claimed)
      if (p0 < p.n && T > from && T <= from + kSplitMaxCover && hops < kSplitFollowMax) {
        uint32_t b[5];
#pragma unroll
        for (uint32_t i = 0; i < 5; i++) b[i] = p0 + i < p.n ? p.in[p0 + i] : 0;
        uint32_t L, size;
        go = decode_element_bf(b[0], b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24), p.n - p0 - 1, &L, &size) &&
             (b[0] & 3) == 0 && (b[0] >> 2) < 62 && size >= kSplitFollowMin;  // a block encoder's long literal
        p1 = p0 + size;  // (> the end of segment T: T's walk is this element)
        T1 = p1 / kSplitSeg;
        T1 = T1 < p.nseg - 1 ? T1 : p.nseg - 1;
        T1 = T1 < T + kSplitMaxCover ? T1 : T + kSplitMaxCover;
      }
      fol = go;
    }
    for (uint64_t m = __ballot(go); m; m &= m - 1) {
      const uint32_t l = (uint32_t)__builtin_ctzll(m);
      const uint32_t bT = readlane(T, l), bT1 = readlane(T1, l), bp1 = readlane(p1, l);
      const unsigned long long k2 = ((unsigned long long)bT << 32) | bp1;
      for (uint32_t t = bT + 1 + threadIdx.x; t <= bT1; t += kSplitWg) atomicMin(&p.nxt_out[t], k2);
    }
    if (go) {
      p.follow[T] = p0;
      from = T;
      p0 = p1;
      hops++;
    }
  }
}

// Is the state the right one?  Independent of how it was reached: the walked segments must form ONE
// chain -- each walk starts exactly where another one ended (or at the stream's first byte), every walk
// is started by one, the last one ends at the stream's end.  Such a chain IS the sequential parse.
// step 0: every walk marks the segment its exit lies in, and checks that segment's entry;
// step 1: every walk but segment 0's must have been marked.  fail[0] != 0: not (yet) the right state.
__global__ __launch_bounds__(256) void split_check_kernel(SplitParams p, uint32_t* reached, uint32_t* fail, int step) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= p.nseg) return;
  const uint32_t seg_hi = (s + 1) * kSplitSeg < p.n ? (s + 1) * kSplitSeg : p.n;
  const uint32_t e = p.prev[s] & 0x7fffffffu;
  if (e >= seg_hi) return;  // the chain passes over this segment
  if (step == 0) {
    const uint32_t m = p.memo[s] & 0x7fffffffu;
    if (m == 0x7fffffffu || m > p.n) {
      fail[0] = 1;  // an invalid element: for the serial walk to judge
    } else if (m < p.n) {
      const uint32_t t = m / kSplitSeg;
      if ((p.prev[t] & 0x7fffffffu) != m) fail[0] = 1;
      reached[t] = 1;
    }
  } else if (s != 0 && !reached[s]) {
    fail[0] = 1;
  } else if (s == 0 && e != 0) {
    fail[0] = 1;
  }
}

}  // namespace snappy_hip
