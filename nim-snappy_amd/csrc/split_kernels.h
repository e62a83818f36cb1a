// split_kernels.h -- where do the 64 KiB blocks of ONE raw Snappy buffer start?
//
// uncompress() of a buffer that decodes to more than one block (snappy.nim:84-110): the stream has
// one varint and no block delimiters (snappy.nim:49-62), so block k's first element can only be
// found by walking the tags (decoder.nim:39-109) -- a chain through the whole stream.  The chain is
// broken up by guessing: the stream is cut into segments of kSplitSeg bytes, and every segment keeps a
// short list of CANDIDATE entries -- positions at which the element chain may enter it.  It starts
// with the guess "at my first byte"; a walk of a segment from a candidate ends at the first element
// start behind the segment (its exit), and that exit becomes a candidate of the segment it lies in.
// A walk that starts wrong falls into step with the real chain after a few elements (elements are
// short), so after a few rounds of "walk every new candidate" the real entry is among every segment's
// candidates -- also where wrong walks never fall into step (an element stream of period 3 such as
// "fe 0a 00" repeated parses as copy2 elements from its second byte too: two chains side by side).
// Which candidates are the real ones is then a question about a linked list: node = (segment,
// candidate), successor = the node of its exit; the nodes reachable from (segment 0, position 0) ARE
// the sequential parse.  They are marked by pointer doubling in log2(segments) steps, and the chain is
// complete if the root's last pointer is the stream's end.  Then a prefix sum over the marked walks'
// output bytes places the segments in the output, and a last walk records, for every 64 KiB boundary,
// the stream position of the element that starts there -- or that an element straddles it (a foreign
// encoder: the caller falls back to the serial walk).  All input-side checks of decodeAllTags are made
// by the walk (decode_element, index_kernel.h).
//
// Round 5: a launch of the walk kernel is not one round any more.  A wave (64 segments, their 16 KiB in LDS)
// runs LOCAL rounds until none of its segments has a candidate that is not walked -- what a walk hands to a
// segment of the same wave is walked without another launch; only what crosses into another wave's segments
// may need the next launch (that wave's dirty flag; a wave without it leaves after one load).  And a walk stops
// where it falls into step with the lane's first walk of the same launch (which leaves a checkpoint in each of
// the segment's eight 32-byte blocks: its first element start there): it takes over that walk's exit and, by
// walking that walk again up to the meeting point (a few elements), its output bytes -- the second walk of a
// segment costs a few elements instead of a hundred.  tools/split_model.py is this procedure on the CPU.
#pragma once

#include "common.h"
#include "index_kernel.h"

namespace snappy_hip {

constexpr uint32_t kSplitSeg = 256;   // stream bytes per lane
constexpr uint32_t kSplitCand = 6;    // candidate entries per segment behind its first guess (more: the caller falls back)
// A walk that starts wrong reads payload bytes as tags, and one payload byte in four looks like a copy4 --
// an element no block encoder writes (nor literal tags 62/63: encoder.nim:44-125).  Only a walk whose last
// kSplitClean elements were "native" hands its exit on as a candidate (else the lists of incompressible
// stretches would overflow with the exits of walks that lead nowhere); a candidate that was handed on
// starts with that credit (bit 31 of its list entry), a first guess does not.
constexpr uint32_t kSplitClean = 8;
constexpr uint32_t kSplitTrusted = 0x80000000u;
// Follow-through (see the walk kernel): literals at least this long, at most so many in a row
constexpr uint32_t kSplitFollowMin = 1024;
constexpr uint32_t kSplitFollowMax = 1u << 16;
// node = segment * kSplitCand + slot; successor codes that are not nodes:
constexpr uint32_t kSplitPending = 0xffffffffu;  // not walked yet / its exit is not a candidate yet
constexpr uint32_t kSplitEnd = 0xfffffffeu;      // the walk ended at the stream's end
constexpr uint32_t kSplitBad = 0xfffffffdu;      // the walk met an invalid element
constexpr uint32_t kSplitFirstCode = kSplitBad;

struct SplitParams {
  const uint8_t* in;      // the tag stream (behind the varint)
  uint32_t n;             // its length
  uint32_t nseg;
  uint32_t* ent;          // [nseg * kSplitCand] candidate entries (all ones: empty)
  uint32_t* ext;          // [nseg * kSplitCand] exit of the walk from it (kSplitPending: not walked, kSplitBad)
  uint32_t* ob;           // [nseg * kSplitCand] output bytes of the elements it walked
  uint32_t* counters;     // [0] candidates added, [1] a segment's list overflowed
  uint32_t* dirty;        // [waves] a candidate was handed to one of the wave's segments (cleared by the wave when it looks)
  int first;              // the first launch: every segment also walks from its guess, its first byte
  uint32_t local_max;     // local rounds a launch (kSplitLocalMax; DEBUG: fewer)
  // after the marking
  uint32_t* entry;        // [nseg] the entry of the real chain (all ones: it passes over the segment)
  uint32_t* outb;         // [nseg] output bytes of the elements that start in the segment
  uint32_t* flags;        // [1] invalid element met, [2] an element straddles a 64 KiB boundary of the output
  const uint64_t* out_at; // [nseg + 1] exclusive prefix sum of outb
  uint32_t* blk_in;       // [nblk] stream position where output block k starts
  uint32_t nblk;          // (blk_in's length: a total that is not the declared length must not write beyond it)
  uint32_t* bad;          // [0] the split's verdict bits (split_table_kernel), [1] the root's last pointer, [2] overflow
};

// Everything the split starts from, in one launch (eight fills and copies of a few bytes each cost 0.2 ms of
// launch gaps in front of the first walk): candidate lists empty but for the root, nothing walked, counters and
// flags zero, no block start known, no unit length, no verdict.
__global__ __launch_bounds__(256) void split_init_kernel(uint32_t* ent_ext, uint64_t n_ent_ext, uint32_t* counters16,
                                                         uint32_t* dirty, uint32_t n_dirty, uint32_t* blk, uint32_t n_blk,
                                                         uint32_t* out_len, uint32_t n_len, uint32_t* bad3) {
  const uint64_t t = blockIdx.x * 256ull + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = t; i < n_ent_ext; i += stride) ent_ext[i] = i == 0 ? kSplitTrusted : 0xffffffffu;  // (node 0: position 0)
  for (uint64_t i = t; i < n_dirty; i += stride) dirty[i] = 0;
  for (uint64_t i = t; i < n_blk; i += stride) blk[i] = 0xffffffffu;
  for (uint64_t i = t; i < n_len; i += stride) out_len[i] = 0;
  if (t < 16) counters16[t] = 0;
  if (t < 3) bad3[t] = 0;
}

// One wave per workgroup: its 64 segments are 16 KiB of stream, staged in LDS with coalesced loads
// before the lanes walk them (64 lanes reading their own segment byte by byte straight from memory move
// a cache line per element and lane).  A segment's row is its 256 bytes and the next four (an element's
// header may reach over), so rows are 65 dwords apart: the lanes of a wave, each in its own row at about
// the same place, read 32 different banks (rows of 256 bytes: ONE bank, and the walks of a local round all
// start near their rows' first bytes).
constexpr uint32_t kSplitWg = 64;
constexpr uint32_t kSplitRow = kSplitSeg + 4;
constexpr uint32_t kSplitStage = kSplitWg * kSplitRow;  // 16 640 bytes: 13 pieces of LDS, nine waves a CU
constexpr uint32_t kSplitLocalMax = kSplitWg + 2;       // local rounds a launch (a chain handed on segment by segment)
extern __shared__ __attribute__((aligned(16))) uint8_t s_split_dyn[];

// the element at offset `off` of the lane's segment (row: its staged bytes): false = invalid
// (decoder.nim:54-57, :67-68, :77-79, truncated copies).  Two aligned dwords hold the tag and the four bytes behind it.
// *native: an element a block encoder writes (no copy4, no literal with three or four length bytes: encoder.nim:44-125).
// A walk is ONE lane's chain of dependent steps, and a lone wave issues an instruction every four to eight cycles whatever
// it is, the compiler's mask bookkeeping around every branch included: the forms without length bytes, 64 bytes or more
// from the stream's end (nothing to check there: decoder.nim:77-79, :86-109 hold), are a dozen selects; only a wave in
// which some lane has another form goes through decode_element_bf.
__device__ __forceinline__ bool split_element(const uint8_t* row, uint32_t off, uint32_t rem, uint32_t* L, uint32_t* size,
                                              bool* native) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(row + (off & ~3u));
  const uint32_t lo = q[0], hi = q[1];  // (staged bytes behind the stream's end are zero)
  const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, off & 3);
  const uint32_t tag = v & 0xff, t = tag & 3, hi6 = tag >> 2;
  const uint32_t is_lit = (uint32_t)((int32_t)(t - 1) >> 31), is_c1 = (uint32_t)((int32_t)((t ^ 1) - 1) >> 31);  // all ones / zero
  *size = (is_lit & (hi6 + 2)) | (~is_lit & (t + 1 + (t >> 1 & t)));
  *L = (is_c1 & (4 + (hi6 & 7))) | (~is_c1 & (hi6 + 1));
  *native = t != 3;
  bool ok = true;
  const bool other = rem < 64 || (tag & 0xf3) == 0xf0;  // near the end, or a literal with length bytes
  if (__ballot(other)) {
    if (other) {
      const uint32_t b14 = (uint32_t)(((((uint64_t)hi << 32) | lo) >> (8 * (off & 3))) >> 8);
      *native = !(t == 3 || (t == 0 && hi6 >= 62));
      ok = decode_element_bf(tag, b14, rem, L, size);
    }
  }
  return ok;
}

__device__ __forceinline__ void split_stage(const SplitParams& p, uint8_t* stage, uint32_t wg_lo) {
  // pieces of 16 bytes: piece i is row i / 16, column 16 * (i % 16); a row's first dword is also the row before's last
  auto put = [&](uint32_t i, uint4 v) {
    const uint32_t row = i >> 4, col = (i & 15) * 16;
    if (row < kSplitWg) {
      uint32_t* d = reinterpret_cast<uint32_t*>(stage + row * kSplitRow + col);
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
    if (col == 0 && row > 0) *reinterpret_cast<uint32_t*>(stage + (row - 1) * kSplitRow + kSplitSeg) = v.x;
  };
  if ((uint64_t)wg_lo + kSplitWg * kSplitSeg + 16 <= p.n) {
    // all sixteen loads of a lane in flight at once (a load, its wait and its stores sixteen times over is sixteen trips
    // to memory one behind the other: 30 us a wave, three times what the walk of the staged bytes takes)
    uint4 v[16];
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) __builtin_memcpy(&v[t], p.in + wg_lo + (t * kSplitWg + threadIdx.x) * 16, 16);
    uint32_t x = 0;
    if (threadIdx.x == 0) __builtin_memcpy(&x, p.in + wg_lo + kSplitWg * kSplitSeg, 4);
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) put(t * kSplitWg + threadIdx.x, v[t]);
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(stage + (kSplitWg - 1) * kSplitRow + kSplitSeg) = x;
  } else {
    for (uint32_t i = threadIdx.x; i < kSplitWg * 16 + 1; i += kSplitWg) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint64_t g = (uint64_t)wg_lo + (uint64_t)i * 16;
      if (g + 16 <= p.n) {
        __builtin_memcpy(&v, p.in + g, 16);
      } else if (g < p.n) {
        uint8_t t[16] = {0};
        for (uint32_t k = 0; k < 16 && g + k < p.n; k++) t[k] = p.in[g + k];
        __builtin_memcpy(&v, t, 16);
      }
      put(i, v);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}


// `pos` becomes a (trusted) candidate of the segment it lies in, if it is not one already; the slot, or kSplitCand.
// A new one raises its wave's dirty flag -- behind the entry: a wave that sees the flag sees the entry.
__device__ __forceinline__ uint32_t split_add(const SplitParams& p, uint32_t pos, uint32_t* n_added) {
  uint32_t* e = p.ent + (pos / kSplitSeg) * kSplitCand;
  for (uint32_t c = 0; c < kSplitCand; c++) {
    const uint32_t old = atomicCAS(&e[c], 0xffffffffu, pos | kSplitTrusted);
    if (old == 0xffffffffu) {
      __hip_atomic_store(&p.dirty[pos / (kSplitSeg * kSplitWg)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++*n_added;
      return c;
    }
    if (old == (pos | kSplitTrusted)) return c;
  }
  p.counters[1] = 1;
  return kSplitCand;
}

// A walk, from w.pos to the segment's end.  MODE 1: the lane's first walk of this launch leaves CHECKPOINTS -- for each of
// the segment's eight 32-byte blocks, the first element start in it (a byte of cp: offset in the block + 1, 0: none).
// MODE 2: a later walk compares its own first start in every block with the checkpoint: equal = it has fallen into step
// with the first walk (*hit, the walk stops there).  MODE 0: neither.  (A bitmap of every start would find the meeting
// point a few elements sooner, for a register file indexed by position: sixteen selects at every block border.)
struct SplitWalk {
  uint32_t pos, out, clean, last_size;
  bool bad;
};
template <int MODE>
__device__ __forceinline__ void split_walk_loop(const uint8_t* row, uint32_t seg_lo, uint32_t seg_hi, uint32_t n, uint64_t* cp,
                                                SplitWalk* w, bool* hit) {
  uint32_t jprev = 8;
  while (w->pos < seg_hi) {
    const uint32_t off = w->pos - seg_lo;
    if (MODE) {
      const uint32_t j = off >> 5, field = (off & 31) + 1;
      const bool crossing = j != jprev;
      jprev = j;
      if (MODE == 2) {
        if (crossing && ((uint32_t)(*cp >> (8 * j)) & 0xff) == field) {
          *hit = true;
          break;
        }
      } else {
        *cp |= crossing ? (uint64_t)field << (8 * j) : 0;
      }
    }
    uint32_t L, size;
    bool nat;
    if (!split_element(row, off, n - w->pos - 1, &L, &size, &nat)) {
      w->bad = true;
      break;
    }
    w->clean = nat ? w->clean + 1 : 0;
    w->out += L;
    w->pos += size;
    w->last_size = size;
  }
}

// One launch: every wave with something new walks its candidates in local rounds (see the head of this file).
__global__ __launch_bounds__(kSplitWg) void split_walk_kernel(SplitParams p) {
  const uint32_t s = blockIdx.x * kSplitWg + threadIdx.x;
  const uint32_t wg_lo = blockIdx.x * kSplitWg * kSplitSeg;
  uint8_t* const stage = s_split_dyn;
  if (!p.first) {  // (uniform: one load)
    if (!__hip_atomic_load(&p.dirty[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
  // (cleared BEFORE the lists are read: a candidate that arrives behind the look leaves the flag up for the next launch)
  if (threadIdx.x == 0) __hip_atomic_store(&p.dirty[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool live = s < p.nseg;
  const uint32_t seg_lo = s * kSplitSeg;
  const uint32_t seg_hi = (s + 1) * kSplitSeg < p.n ? (s + 1) * kSplitSeg : p.n;
  const uint8_t* const row = stage + threadIdx.x * kSplitRow;
  uint32_t* const my_ent = p.ent + (size_t)s * kSplitCand;
  uint32_t* const my_ext = p.ext + (size_t)s * kSplitCand;
  uint32_t* const my_ob = p.ob + (size_t)s * kSplitCand;
  split_stage(p, stage, wg_lo);
  // the lane's first walk of this launch (in the first launch: the guess): its checkpoints, where it started, its exit
  // code and output bytes, whether it handed its exit on
  uint64_t cp = 0;
  bool have_first = false, first_handed = false;
  uint32_t first_entry = 0, first_code = kSplitPending, first_ob = 0;
  uint32_t done = 0, n_added = 0;  // (done: the slots walked in this launch)
  for (uint32_t it = 0; it < p.local_max; it++) {
    uint32_t todo = 0;
    uint32_t ent[kSplitCand + 1];
    ent[kSplitCand] = seg_lo;  // bit kSplitCand: the guess, which is nobody's successor and has no slot
    if (p.first && it == 0) {  // (nothing is listed yet but the root)
      if (live && s != 0) todo |= 1u << kSplitCand;
#pragma unroll
      for (uint32_t c = 0; c < kSplitCand; c++) ent[c] = 0xffffffffu;
      if (s == 0) ent[0] = kSplitTrusted, todo |= 1u;
    } else {
      // (all twelve loads in flight at once: one after the other they are twelve trips to the L2 a round)
      uint32_t x[kSplitCand];
#pragma unroll
      for (uint32_t c = 0; c < kSplitCand; c++)
        ent[c] = live ? __hip_atomic_load(&my_ent[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
#pragma unroll
      for (uint32_t c = 0; c < kSplitCand; c++)
        x[c] = live ? __hip_atomic_load(&my_ext[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
      for (uint32_t c = 0; c < kSplitCand; c++)
        if (ent[c] != 0xffffffffu && !((done >> c) & 1) && x[c] == kSplitPending) todo |= 1u << c;
    }
    if (!__ballot(todo != 0)) break;
    for (uint32_t cc = 0; cc <= kSplitCand; cc++) {
      const uint32_t c = cc == 0 ? kSplitCand : cc - 1;  // (the guess first)
      if (!__ballot((todo >> c) & 1)) continue;
      if ((todo >> c) & 1) {
        uint32_t e = ent[kSplitCand];
#pragma unroll
        for (uint32_t k = 0; k < kSplitCand; k++) e = k == c ? ent[k] : e;
        SplitWalk w{e & ~kSplitTrusted, 0, (e & kSplitTrusted) ? kSplitClean : 0, 0, false};
        const uint32_t entry = w.pos;
        uint32_t code = kSplitPending;  // (pending: the walk reached the segment's end on its own)
        bool took_over = false, hit = false;
        if (!have_first) {
          split_walk_loop<1>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit);
        } else {
          split_walk_loop<2>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit);
          if (hit) {
            // In step with the lane's first walk from here on: its exit, and what it put out behind this point (found by
            // walking it again up to here, a few elements).  An exit that walk kept to itself -- it ended without its
            // credit of native elements -- is no use: this walk goes on to the end on its own, and hands on what it finds.
            if (first_code >= kSplitFirstCode || first_handed) {
              SplitWalk r{first_entry, 0, 0, 0, false};
              bool h2 = false;
              split_walk_loop<0>(row, seg_lo, w.pos, p.n, &cp, &r, &h2);
              code = first_code;
              w.out += first_ob - r.out;
              took_over = true;
            } else {
              split_walk_loop<0>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit);
            }
          }
        }
        if (w.bad) code = kSplitBad;
        const bool own_exit = code == kSplitPending;
        if (own_exit) code = w.pos == p.n ? kSplitEnd : w.pos;
        if (c < kSplitCand) {
          __hip_atomic_store(&my_ob[c], w.out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&my_ext[c], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        done |= 1u << c;
        bool handed_on = took_over;
        if (own_exit && w.pos < p.n && w.clean >= kSplitClean) {
          uint32_t pos = w.pos;
          uint32_t slot = split_add(p, pos, &n_added);
          handed_on = slot < kSplitCand;
          // Follow-through.  Long literals back to back (incompressible blocks: one literal of 64 KiB each) are
          // a chain that would be discovered ONE literal per round -- the segment a literal ends in learns the
          // candidate from the walk of the segment the literal starts in.  So a walk that leaves its segment
          // with a long literal and lands on another one does that segment's walk as well (it is that one
          // element), and goes on while it keeps landing on long literals: one trip to memory per literal.
          // (A walk of payload bytes lands on the tag of a long literal one time in a hundred: such chains die.)
          if (w.last_size >= kSplitFollowMin) {
            for (uint32_t hop = 0; hop < kSplitFollowMax && slot < kSplitCand; hop++) {
              const uint32_t node = (pos / kSplitSeg) * kSplitCand + slot;
              if (__hip_atomic_load(&p.ext[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kSplitPending) break;  // somebody has been here
              uint32_t b[5];
#pragma unroll
              for (uint32_t i = 0; i < 5; i++) b[i] = pos + i < p.n ? p.in[pos + i] : 0;
              uint32_t L, size;
              if (!decode_element_bf(b[0], b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24), p.n - pos - 1, &L, &size)) break;
              // (a block encoder's long literal; size >= the segment: the walk is this one element)
              if ((b[0] & 3) != 0 || (b[0] >> 2) >= 62 || size < kSplitFollowMin) break;
              const uint32_t p1 = pos + size;
              __hip_atomic_store(&p.ob[node], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&p.ext[node], p1 == p.n ? kSplitEnd : p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (p1 >= p.n) break;
              pos = p1;
              slot = split_add(p, pos, &n_added);
            }
          }
        }
        if (!have_first) {
          have_first = true;
          first_entry = entry;
          first_code = code;
          first_ob = w.out;
          first_handed = handed_on;
        }
      }
    }
    // (the lists are read again: the lanes' entries and exits of this round have arrived at the L2)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  for (int d = 32; d >= 1; d >>= 1) n_added += __shfl_xor(n_added, d, 64);
  if (threadIdx.x == 0 && n_added) atomicAdd(&p.counters[0], n_added);
}

// successor of every node, and the root's mark
__global__ __launch_bounds__(256) void split_succ_kernel(SplitParams p, uint32_t* jump, uint8_t* reach) {
  const uint32_t node = blockIdx.x * 256 + threadIdx.x;
  if (node >= p.nseg * kSplitCand) return;
  uint32_t j = kSplitPending;
  const uint32_t x = p.ext[node];
  if (p.ent[node] == 0xffffffffu) {
    j = kSplitPending;
  } else if (x >= kSplitFirstCode) {
    j = x;
  } else {
    const uint32_t t = x / kSplitSeg;
    for (uint32_t c = 0; c < kSplitCand; c++)
      if (p.ent[t * kSplitCand + c] == (x | kSplitTrusted)) j = t * kSplitCand + c;  // (what is handed on is trusted)
  }
  jump[node] = j;
  reach[node] = node == 0 ? 1 : 0;
}

// one step of the pointer jumping, four-fold (half the launches of doubling): a marked node marks the three
// nodes its pointer leads to in one, two and three hops, every pointer then shows four times as far.  (Before
// step k the marked nodes are those less than 4^k hops from the root; they mark 4^k, 2*4^k, 3*4^k further.)
__global__ __launch_bounds__(256) void split_double_kernel(uint32_t n_nodes, const uint32_t* jump_in, uint32_t* jump_out,
                                                           uint8_t* reach) {
  const uint32_t node = blockIdx.x * 256 + threadIdx.x;
  if (node >= n_nodes) return;
  uint32_t j = jump_in[node];
  const bool mark = j < kSplitFirstCode && reach[node];
#pragma unroll
  for (int hop = 0; hop < 3 && j < kSplitFirstCode; hop++) {
    if (mark) reach[j] = 1;
    j = jump_in[j];
  }
  jump_out[node] = j;
}

// the real chain's entry and output bytes of every segment
// (root: the root's last pointer -- the chain is complete if that is the stream's end; if not, no segment gets an
// entry, bit 8 of the verdict says so, and the host, which looks once behind the decode, starts again with looks)
__global__ __launch_bounds__(256) void split_select_kernel(SplitParams p, const uint8_t* reach, const uint32_t* root) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  const uint32_t r = *root;
  if (s == 0) {
    p.bad[1] = r;
    p.bad[2] = p.counters[1];
    if (r != kSplitEnd) atomicOr(&p.bad[0], 8u);
  }
  if (s >= p.nseg) return;
  uint32_t e = 0xffffffffu, o = 0;
  for (uint32_t c = 0; c < kSplitCand && r == kSplitEnd; c++)
    if (reach[s * kSplitCand + c]) {
      e = p.ent[s * kSplitCand + c] & ~kSplitTrusted;
      o = p.ob[s * kSplitCand + c];
    }
  p.entry[s] = e;
  p.outb[s] = o;
}

// Exclusive prefix sum of outb over millions of segments (one workgroup running through them takes 3 ms for a
// GiB of stream): tiles of 4096 are summed, the few hundred tile sums are scanned by scan_sizes_kernel, and
// every tile is scanned again from its base.
constexpr uint32_t kSplitTile = 4096;
__global__ __launch_bounds__(256) void split_tile_sums_kernel(const uint32_t* v, uint32_t n, uint32_t* tile_sum) {
  __shared__ uint32_t s_w[4];
  const uint32_t lo = blockIdx.x * kSplitTile;
  uint32_t x = 0;
  for (uint32_t i = lo + threadIdx.x; i < lo + kSplitTile && i < n; i += 256) x += v[i];
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];  // (a tile's output: < 2^32)
}
__global__ __launch_bounds__(256) void split_tile_scan_kernel(const uint32_t* v, uint32_t n, const uint64_t* tile_base,
                                                               uint64_t* offsets) {
  __shared__ uint64_t s_w[4];
  __shared__ uint64_t s_carry;
  const uint32_t lo = blockIdx.x * kSplitTile, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_carry = tile_base[blockIdx.x];
    if (blockIdx.x == 0) offsets[0] = 0;
  }
  __syncthreads();
  for (uint32_t c = lo; c < lo + kSplitTile; c += 256) {
    const uint32_t i = c + threadIdx.x;
    uint64_t x = i < n ? v[i] : 0;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t y = __shfl_up(x, d, 64);
      if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    uint64_t before = s_carry;
    for (uint32_t k = 0; k < wv; k++) before += s_w[k];
    if (i < n) offsets[i + 1] = before + x;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = before + x;
    __syncthreads();
  }
}

// the last walk, along the real chain: which element starts at each 64 KiB boundary of the output
__global__ __launch_bounds__(kSplitWg) void split_locate_kernel(SplitParams p) {
  const uint32_t s = blockIdx.x * kSplitWg + threadIdx.x;
  const uint32_t wg_lo = blockIdx.x * kSplitWg * kSplitSeg;
  uint8_t* const stage = s_split_dyn;
  const bool live = s < p.nseg;
  const uint32_t seg_hi = (s + 1) * kSplitSeg < p.n ? (s + 1) * kSplitSeg : p.n;
  uint32_t pos = live ? p.entry[s] : 0xffffffffu;
  if (!__ballot(pos < seg_hi)) return;
  split_stage(p, stage, wg_lo);
  if (!live || pos >= seg_hi) return;
  const uint8_t* const row = stage + threadIdx.x * kSplitRow;
  uint64_t op = p.out_at[s];
  while (pos < seg_hi) {
    uint32_t L, size;
    bool nat;
    if (!split_element(row, pos - s * kSplitSeg, p.n - pos - 1, &L, &size, &nat)) {
      p.flags[1] = 1;
      return;
    }
    if ((op & 0xffffu) == 0 && (op >> 16) < p.nblk) p.blk_in[op >> 16] = pos;
    else if (((op + L - 1) >> 16) != (op >> 16)) p.flags[2] = 1;  // crosses a 64 KiB boundary of the output
    op += L;
    pos += size;
  }
}

// the blocks as units of the block decoder, from where they start in the stream (blk_in[k], k >= 1; block 0 at 0)
// *bad: 1 = a block without a start (the caller falls back to the serial walk), 2 = the stream is invalid (its
// elements do not produce the declared length, snappy.nim:107-108, or the last walk met an invalid one), 4 = an
// element straddles a 64 KiB boundary (a foreign encoder: fall back) -- the speculative split's verdicts, looked
// at by the host once, behind the decode
__global__ __launch_bounds__(256) void split_table_kernel(const uint32_t* blk_in, uint32_t nblk, uint32_t n_tags,
                                                          uint32_t hdr, uint64_t len, uint64_t* in_off, uint32_t* in_len,
                                                          uint64_t* out_off, uint32_t* out_cap, uint32_t* bad,
                                                          const uint64_t* total, const uint32_t* flags) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k == 0 && total) {
    uint32_t v = 0;
    if (*total != len || flags[1]) v |= 2;
    if (flags[2]) v |= 4;
    if (v) atomicOr(bad, v);
  }
  if (k >= nblk) return;
  const uint32_t b0 = k == 0 ? 0 : blk_in[k], b1 = k + 1 == nblk ? n_tags : blk_in[k + 1];
  const bool ok = b0 != 0xffffffffu && b1 != 0xffffffffu && b1 >= b0 && b1 <= n_tags;
  if (!ok) atomicOr(bad, 1u);  // (a block nobody recorded the start of: the caller falls back; the unit is made empty)
  const uint64_t oo = (uint64_t)k * kMaxBlockLen;
  in_off[k] = (uint64_t)hdr + (ok ? b0 : 0);
  in_len[k] = ok ? b1 - b0 : 0;
  out_off[k] = oo;
  out_cap[k] = (uint32_t)(len - oo < kMaxBlockLen ? len - oo : kMaxBlockLen);
}

// Behind the decode: every block decoded to its full length?  (bit 16 of the verdict: e.g. a copy that reaches into an
// earlier block -- a foreign encoder; the caller falls back.)  The host reads the verdict's three words, once.
__global__ __launch_bounds__(256) void split_verdict_kernel(const uint32_t* status, const uint32_t* out_len, uint32_t nblk,
                                                            uint64_t len, uint32_t* bad) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nblk) return;
  const uint64_t oo = (uint64_t)k * kMaxBlockLen;
  const uint32_t oc = (uint32_t)(len - oo < kMaxBlockLen ? len - oo : kMaxBlockLen);
  if (status[k] != kOk || out_len[k] != oc) atomicOr(bad, 16u);
}

}  // namespace snappy_hip
