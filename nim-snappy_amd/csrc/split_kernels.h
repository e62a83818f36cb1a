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
// Round 5: the rounds are not launches over every segment any more.  ONE launch (split_bulk_kernel: a wave per 64
// segments, their 16 KiB staged in LDS) walks every segment from its guess -- the segment's FIRST walk, which leaves a
// checkpoint in each of the segment's eight 32-byte blocks (its first element start there) and a summary (entry, exit,
// output bytes, whether the exit was handed on) -- and then, in a few local rounds, what those walks hand to segments
// of the same wave: a later walk stops where it enters a block at the first walk's checkpoint (it has fallen into
// step with it), and takes over that walk's exit and, by walking that walk again up to the meeting point, its output
// bytes.  Most second walks meet the first within a dozen elements; the few of a wave that do not would hold up its
// other 63 lanes for the length of a whole walk each, so a local walk has a budget of elements, and what is not done by
// then -- and what is handed to another wave's segment -- goes on a QUEUE of nodes.  The later rounds are launches of
// split_tail_kernel over that queue, a LANE per node whatever its segment (the lane stages its own segment), each
// round's new candidates on the next round's queue: the stragglers of all waves side by side, as many lanes as there
// are stragglers.  tools/split_model.py is this procedure on the CPU.
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
  uint32_t* counters;     // [0] candidates added, [1] a segment's list (or the queue) overflowed, [2 + r] round r's queue length
  // the segment's first walk (the bulk launch): checkpoints, entry, exit code, output bytes | handed on << 31
  uint64_t* cp;           // [nseg]
  uint32_t* f_entry;      // [nseg]
  uint32_t* f_code;       // [nseg]
  uint32_t* f_ob;         // [nseg]
  // the queue of nodes to walk: this round's (tail launch) and the next one's
  const uint32_t* q_in;
  const uint32_t* q_in_count;
  uint32_t* q_out;
  uint32_t* q_out_count;
  uint32_t q_cap;
  uint32_t budget, hops;  // kSplitBudget, kSplitHops (DEBUG builds: others, for measurements)
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
__global__ __launch_bounds__(256) void split_init_kernel(uint32_t* ent_ext, uint64_t n_ent_ext, uint32_t* counters24,
                                                         uint32_t* blk, uint32_t n_blk, uint32_t* out_len, uint32_t n_len,
                                                         uint32_t* bad3) {
  const uint64_t t = blockIdx.x * 256ull + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = t; i < n_ent_ext; i += stride) ent_ext[i] = i == 0 ? kSplitTrusted : 0xffffffffu;  // (node 0: position 0)
  for (uint64_t i = t; i < n_blk; i += stride) blk[i] = 0xffffffffu;
  for (uint64_t i = t; i < n_len; i += stride) out_len[i] = 0;
  if (t < 24) counters24[t] = 0;  // (the counters' sixteen words and the flags' eight behind them)
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
constexpr uint32_t kSplitBudget = 64;     // elements a second walk of the bulk launch may take (most meet the first walk within a dozen)
constexpr uint32_t kSplitMaxRounds = 13;  // tail launches at most (their queue lengths live in counters[2 .. 15])
constexpr uint32_t kSplitNoNode = 0xffffffffu;
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
// Slot 0 is not given out here: it is the segment's own -- what the first walk of the segment before hands to it inside a
// wave of the bulk launch, written by the lane of the segment itself without an atomic (segment 0: the root).  If that
// entry is this position it is found (behind the bulk launch always; during it, perhaps not yet: then the position is
// listed twice, and both nodes are walked -- the marking takes either).
__device__ __forceinline__ uint32_t split_add(const SplitParams& p, uint32_t pos, bool* is_new, uint32_t* n_added) {
  uint32_t* e = p.ent + (pos / kSplitSeg) * kSplitCand;
  *is_new = false;
  // (the look at slot 0 and the first compare-and-swap travel together: a walk along a chain is a chain of trips to
  // memory, and this is one of them.  If slot 0 turns out to hold the position, the swap has listed it twice -- see above)
  const uint32_t e0 = __hip_atomic_load(&e[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t c = 1; c < kSplitCand; c++) {
    const uint32_t old = atomicCAS(&e[c], 0xffffffffu, pos | kSplitTrusted);
    if (old == 0xffffffffu) {
      *is_new = true;
      ++*n_added;
      return c;
    }
    if (old == (pos | kSplitTrusted)) return c;
    if (e0 == (pos | kSplitTrusted)) return 0;  // (slots 1 .. are somebody else's: the own slot holds it)
  }
  p.counters[1] = 1;
  return kSplitCand;
}

// The nodes the lanes of a wave want walked in the next round, onto its queue: one atomic a wave.  (Wave-uniform call.)
__device__ __forceinline__ void split_push(const SplitParams& p, uint32_t node) {
  const uint64_t m = __ballot(node != kSplitNoNode);
  if (!m) return;
  uint32_t base = 0;
  const uint32_t lane = threadIdx.x & 63;
  if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(p.q_out_count, (uint32_t)__builtin_popcountll(m));
  base = __shfl(base, __builtin_ctzll(m), 64);
  if (node != kSplitNoNode) {
    const uint32_t i = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
    if (i < p.q_cap) p.q_out[i] = node;
    else p.counters[1] = 1;  // (what does not fit is lost: the chain may be incomplete, like with a list that overflowed)
  }
}

__device__ __forceinline__ void split_push2(const SplitParams& p, uint32_t a, uint32_t b) {
  const uint64_t ma = __ballot(a != kSplitNoNode), mb = __ballot(b != kSplitNoNode);
  if (!(ma | mb)) return;
  const uint32_t na = (uint32_t)__builtin_popcountll(ma), nb = (uint32_t)__builtin_popcountll(mb);
  uint32_t base = 0;
  const uint32_t lane = threadIdx.x & 63;
  if (lane == 0) base = atomicAdd(p.q_out_count, na + nb);
  base = __shfl(base, 0, 64);
  const uint64_t below = (1ull << lane) - 1;
  if (a != kSplitNoNode) {
    const uint32_t i = base + (uint32_t)__builtin_popcountll(ma & below);
    if (i < p.q_cap) p.q_out[i] = a;
    else p.counters[1] = 1;
  }
  if (b != kSplitNoNode) {
    const uint32_t i = base + na + (uint32_t)__builtin_popcountll(mb & below);
    if (i < p.q_cap) p.q_out[i] = b;
    else p.counters[1] = 1;
  }
}

// A walk, from w.pos to the segment's end (at most `budget` elements).  MODE 1: the segment's first walk leaves
// CHECKPOINTS -- for each of the segment's eight 32-byte blocks, the first element start in it (a byte of cp: offset in
// the block + 1, 0: none).  MODE 2: a later walk compares its own first start in every block with the checkpoint:
// equal = it has fallen into step with the first walk (*hit, the walk stops there).  MODE 0: neither.  (A bitmap of
// every start would find the meeting point a few elements sooner, for a register file indexed by position: sixteen
// selects at every block border.)
struct SplitWalk {
  uint32_t pos, out, clean, last_size;
  bool bad;
};
template <int MODE>
__device__ __forceinline__ void split_walk_loop(const uint8_t* row, uint32_t seg_lo, uint32_t seg_hi, uint32_t n, uint64_t* cp,
                                                SplitWalk* w, bool* hit, uint32_t budget) {
  uint32_t jprev = 8;
  while (w->pos < seg_hi && budget-- != 0) {
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

// A walk that reached its segment's end on its own hands its exit on -- if its last kSplitClean elements were native.
// *handed: the exit is a candidate now (or was one).  Returns the node that still has to be walked (the exit's, or the
// last one of the follow-through), kSplitNoNode if there is none or it is a segment of the caller's own wave (my_wg),
// whose owner will see it in its list.
__device__ __forceinline__ uint32_t split_hand_on(const SplitParams& p, const SplitWalk& w, uint32_t my_wg, bool* handed,
                                                  uint32_t* n_added) {
  *handed = false;
  if (w.pos >= p.n || w.clean < kSplitClean) return kSplitNoNode;
  uint32_t pos = w.pos;
  bool is_new;
  uint32_t slot = split_add(p, pos, &is_new, n_added);
  *handed = slot < kSplitCand;
  uint32_t open = is_new ? (pos / kSplitSeg) * kSplitCand + slot : kSplitNoNode;
  // Follow-through.  Long literals back to back (incompressible blocks: one literal of 64 KiB each) are
  // a chain that would be discovered ONE literal per round -- the segment a literal ends in learns the
  // candidate from the walk of the segment the literal starts in.  So a walk that leaves its segment
  // with a long literal and lands on another one does that segment's walk as well (it is that one
  // element), and goes on while it keeps landing on long literals: one trip to memory per literal.
  // (A walk of payload bytes lands on the tag of a long literal one time in a hundred: such chains die.)
  if (w.last_size >= kSplitFollowMin) {
    for (uint32_t hop = 0; hop < kSplitFollowMax && slot < kSplitCand; hop++) {
      const uint32_t node = (pos / kSplitSeg) * kSplitCand + slot;
      if (__hip_atomic_load(&p.ext[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kSplitPending) {  // somebody has been here
        open = kSplitNoNode;
        break;
      }
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
      open = kSplitNoNode;
      if (p1 >= p.n) break;
      pos = p1;
      slot = split_add(p, pos, &is_new, n_added);
      if (is_new) open = (pos / kSplitSeg) * kSplitCand + slot;
    }
  }
  if (open != kSplitNoNode && open / (kSplitCand * kSplitWg) == my_wg) open = kSplitNoNode;
  return open;
}

// The walk of candidate (segment t, slot c, list entry e) against the segment's first walk (cp, f_*).  Returns the node
// for the next round's queue: its exit's, or ITS OWN if it was not done within the budget (it is walked from its start
// again there, without one), or kSplitNoNode.
__device__ __forceinline__ uint32_t split_candidate(const SplitParams& p, const uint8_t* row, uint32_t t, uint32_t c, uint32_t e,
                                                    uint64_t cp, uint32_t f_entry, uint32_t f_code, uint32_t f_obh,
                                                    uint32_t budget, uint32_t my_wg, uint32_t* n_added) {
  const uint32_t seg_lo = t * kSplitSeg;
  const uint32_t seg_hi = seg_lo + kSplitSeg < p.n ? seg_lo + kSplitSeg : p.n;
  const uint32_t node = t * kSplitCand + c;
  SplitWalk w{e & ~kSplitTrusted, 0, (e & kSplitTrusted) ? kSplitClean : 0, 0, false};
  uint32_t code = kSplitPending;  // (pending: the walk reached the segment's end on its own)
  bool hit = false;
  split_walk_loop<2>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit, budget);
  if (hit) {
    // In step with the segment's first walk from here on: its exit, and what it put out behind this point (found by
    // walking it again up to here, a few elements).  An exit that walk kept to itself -- it ended without its
    // credit of native elements -- is no use: this walk goes on to the end on its own, and hands on what it finds.
    if (f_code >= kSplitFirstCode || (f_obh >> 31)) {
      SplitWalk r{f_entry, 0, 0, 0, false};
      bool h2 = false;
      split_walk_loop<0>(row, seg_lo, w.pos, p.n, &cp, &r, &h2, 0xffffffffu);
      code = f_code;
      w.out += (f_obh & 0x7fffffffu) - r.out;
    } else {
      if (budget != 0xffffffffu) return node;
      split_walk_loop<0>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit, 0xffffffffu);
    }
  } else if (!w.bad && w.pos < seg_hi) {
    return node;  // (the budget)
  }
  if (w.bad) code = kSplitBad;
  const bool own_exit = code == kSplitPending;
  if (own_exit) code = w.pos == p.n ? kSplitEnd : w.pos;
  __hip_atomic_store(&p.ob[node], w.out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p.ext[node], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool handed;
  return own_exit ? split_hand_on(p, w, my_wg, &handed, n_added) : kSplitNoNode;
}

#ifdef SNAPPY_HIP_DEBUG  // phase timers of the bulk launch (100 MHz ticks summed over the waves): flags[3 .. 7]
#define SPLIT_TICK(k)                                                                  \
  do {                                                                                 \
    const uint64_t now_ = wall_clock64();                                              \
    if (threadIdx.x == 0) atomicAdd(&p.flags[3 + (k)], (uint32_t)(now_ - tick_));      \
    tick_ = now_;                                                                      \
  } while (0)
#else
#define SPLIT_TICK(k) do { } while (0)
#endif

// The bulk launch: every segment's first walk, and the walk of what the first walk of the segment before hands to it
// (see the head of this file).  A wave's trips to memory are what it takes: its stream bytes, one atomic for the exits
// that leave the wave or skip a segment, one for the exits of second walks that did not meet the first, one for the
// queue -- the hand-over from a segment to the next inside the wave is a shuffle, and that candidate is slot 0 of the
// segment's list, written by the segment's own lane.  (With every hand-over a compare-and-swap and every look at a
// list a load, a wave made ten trips one behind the other: 85 us, 24 of them walking.)
__global__ __launch_bounds__(kSplitWg) void split_bulk_kernel(SplitParams p) {
#ifdef SNAPPY_HIP_DEBUG
  uint64_t tick_ = wall_clock64();
#endif
  const uint32_t s = blockIdx.x * kSplitWg + threadIdx.x;
  const uint32_t wg_lo = blockIdx.x * kSplitWg * kSplitSeg;
  uint8_t* const stage = s_split_dyn;
  const bool live = s < p.nseg;
  const uint32_t seg_lo = s * kSplitSeg;
  const uint32_t seg_hi = (s + 1) * kSplitSeg < p.n ? (s + 1) * kSplitSeg : p.n;
  const uint8_t* const row = stage + threadIdx.x * kSplitRow;
  split_stage(p, stage, wg_lo);
  SPLIT_TICK(0);
  uint32_t n_added = 0;
  // the first walk: from the segment's first byte, a guess, which is nobody's successor and has no slot (segment 0: the
  // root, slot 0, and no guess)
  uint64_t cp = 0;
  uint32_t f_code = kSplitBad, f_obh = 0, push_a = kSplitNoNode, push_b = kSplitNoNode;
  uint32_t direct = 0xffffffffu;  // the exit, if it goes to the next segment by shuffle
  if (live) {
    SplitWalk w{seg_lo, 0, s == 0 ? kSplitClean : 0, 0, false};
    bool hit = false, handed = false;
    split_walk_loop<1>(row, seg_lo, seg_hi, p.n, &cp, &w, &hit, 0xffffffffu);
    SPLIT_TICK(1);
    f_code = w.bad ? kSplitBad : (w.pos == p.n ? kSplitEnd : w.pos);
    if (s == 0) {
      __hip_atomic_store(&p.ob[0], w.out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p.ext[0], f_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!w.bad && w.pos < p.n && w.clean >= kSplitClean) {
      if (threadIdx.x + 1 < kSplitWg && w.pos < seg_hi + kSplitSeg && w.last_size < kSplitFollowMin) {
        direct = w.pos;
        handed = true;
      } else {
        push_a = split_hand_on(p, w, 0xffffffffu, &handed, &n_added);
      }
    }
    f_obh = w.out | (handed ? 0x80000000u : 0u);
    p.cp[s] = cp;
    p.f_entry[s] = seg_lo;
    p.f_code[s] = f_code;
    p.f_ob[s] = f_obh;
  }
  SPLIT_TICK(2);
  // the second walk: what the segment before handed over
  const uint32_t cand = __shfl_up(direct, 1, 64);
  if (live && threadIdx.x != 0 && cand != 0xffffffffu) {
    __hip_atomic_store(&p.ent[(size_t)s * kSplitCand], cand | kSplitTrusted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    push_b = split_candidate(p, row, s, 0, cand | kSplitTrusted, cp, seg_lo, f_code, f_obh, p.budget, 0xffffffffu, &n_added);
  }
  SPLIT_TICK(3);
  split_push2(p, push_a, push_b);
  SPLIT_TICK(4);
  for (int d = 32; d >= 1; d >>= 1) n_added += __shfl_xor(n_added, d, 64);
  if (threadIdx.x == 0 && n_added) atomicAdd(&p.counters[0], n_added);
}

// A lane stages segment t by itself (row: its own 260 bytes of LDS), all its loads in flight at once.
__device__ __forceinline__ void split_stage_row(const SplitParams& p, uint8_t* row, uint32_t t) {
  const uint64_t g0 = (uint64_t)t * kSplitSeg;
  uint32_t* const d = reinterpret_cast<uint32_t*>(row);
  if (g0 + kSplitSeg + 16 <= p.n) {
    uint4 v[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) __builtin_memcpy(&v[k], p.in + g0 + k * 16, 16);
    uint32_t x;
    __builtin_memcpy(&x, p.in + g0 + kSplitSeg, 4);
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) d[4 * k] = v[k].x, d[4 * k + 1] = v[k].y, d[4 * k + 2] = v[k].z, d[4 * k + 3] = v[k].w;
    d[kSplitSeg / 4] = x;
  } else {
    for (uint32_t k = 0; k <= kSplitSeg / 4; k++) {
      uint32_t x = 0;
      for (uint32_t b = 0; b < 4; b++)
        if (g0 + 4 * k + b < p.n) x |= (uint32_t)p.in[g0 + 4 * k + b] << (8 * b);
      d[k] = x;
    }
  }
}

// A later round: the queue's nodes, a lane each; the lane stages its node's segment itself, and goes on along what the
// walk hands on (the next node of ITS chain: repeated strings are a literal of kilobytes, a stretch of copies, a
// literal ... -- a hop each, and nothing but the chain itself finds them), kSplitHops nodes at most.
constexpr uint32_t kSplitHops = 16;
__global__ __launch_bounds__(kSplitWg) void split_tail_kernel(SplitParams p) {
  uint8_t* const row = s_split_dyn + threadIdx.x * kSplitRow;
  const uint32_t n_items = min(__hip_atomic_load(p.q_in_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), p.q_cap);
  uint32_t n_added = 0;
  for (uint32_t base = blockIdx.x * kSplitWg; base < n_items; base += gridDim.x * kSplitWg) {
    const uint32_t i = base + threadIdx.x;
    uint32_t node = i < n_items ? p.q_in[i] : kSplitNoNode;
    for (uint32_t hop = 0; hop < p.hops && __ballot(node != kSplitNoNode); hop++) {
      uint32_t next = kSplitNoNode;
      if (node != kSplitNoNode) {
        const uint32_t t = node / kSplitCand, c = node % kSplitCand;
        // (the node's list entry, its segment's bytes and the first walk's summary in ONE trip: a hop is a few trips to
        // memory and a walk; whether the node has been walked meanwhile -- by a follow-through, by the bulk launch, as a
        // duplicate of the queue -- is looked at behind it)
        const uint32_t e = __hip_atomic_load(&p.ent[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t x = __hip_atomic_load(&p.ext[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t cp = p.cp[t];
        const uint32_t fe = p.f_entry[t], fc = p.f_code[t], fo = p.f_ob[t];
        split_stage_row(p, row, t);
        if (x == kSplitPending && e != 0xffffffffu)
          next = split_candidate(p, row, t, c, e, cp, fe, fc, fo, 0xffffffffu, 0xffffffffu, &n_added);
      }
      node = next;
    }
    split_push(p, node);
  }
  for (int d = 32; d >= 1; d >>= 1) n_added += __shfl_xor(n_added, d, 64);
  if (threadIdx.x == 0 && n_added) atomicAdd(&p.counters[0], n_added);
}

// successor of every node, and the root's mark.  The marking's arrays are SLOT-major (id = slot * nseg + segment: the
// root is id 0): most segments list one or two candidates, so the ids of slots 2 .. 5 are nearly all empty, whole
// 256-id blocks of them -- any[block] says which blocks hold a node at all, and a step of the pointer jumping leaves a
// block without one after one load (segment-major, every step streamed all six slots of every segment: 0.45 ms of the
// 1 GiB buffer's 0.5 ms look).
__global__ __launch_bounds__(256) void split_succ_kernel(SplitParams p, uint32_t* jump, uint8_t* reach, uint32_t* any) {
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  bool listed = false;
  if (id < p.nseg * kSplitCand) {
    const uint32_t c = id / p.nseg, s = id - c * p.nseg, node = s * kSplitCand + c;
    uint32_t j = kSplitPending;
    const uint32_t x = p.ext[node];
    listed = p.ent[node] != 0xffffffffu;
    if (!listed) {
      j = kSplitPending;
    } else if (x >= kSplitFirstCode) {
      j = x;
    } else {
      const uint32_t t = x / kSplitSeg;
      for (uint32_t k = 0; k < kSplitCand; k++)
        if (p.ent[t * kSplitCand + k] == (x | kSplitTrusted)) j = k * p.nseg + t;  // (what is handed on is trusted)
    }
    jump[id] = j;
    reach[id] = id == 0 ? 1 : 0;
  }
  const int some = __syncthreads_or(listed);
  if (threadIdx.x == 0) any[blockIdx.x] = (uint32_t)some;
}

// one step of the pointer jumping, kSplitJump-fold (a third of the launches of doubling): a marked node marks the
// nodes its pointer leads to in one, two ... kSplitJump - 1 hops, every pointer then shows kSplitJump times as far.
// (Before step k the marked nodes are those less than J^k hops from the root; they mark J^k, 2 J^k ... further.)
constexpr int kSplitJump = 8;
// (A block without a node is never read: only a node is anybody's successor.)
__global__ __launch_bounds__(256) void split_double_kernel(uint32_t n_nodes, const uint32_t* jump_in, uint32_t* jump_out,
                                                           uint8_t* reach, const uint32_t* any) {
  if (!any[blockIdx.x]) return;
  const uint32_t node = blockIdx.x * 256 + threadIdx.x;
  if (node >= n_nodes) return;
  uint32_t j = jump_in[node];
  const bool mark = j < kSplitFirstCode && reach[node];
#pragma unroll
  for (int hop = 0; hop < kSplitJump - 1 && j < kSplitFirstCode; hop++) {
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
    if (reach[c * p.nseg + s]) {  // (slot-major: split_succ_kernel)
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

// the last walk, along the real chain: which element starts at each 64 KiB boundary of the output.  One boundary in a
// hundred and fifty segments: a LANE per boundary finds the segment whose elements cover it (out_at is sorted: the last
// segment that starts at or below the boundary), stages that one segment and walks it from the real chain's entry.
// (Round 4 walked every segment again, a wave per 64: 0.5 ms for a GiB.)
__global__ __launch_bounds__(kSplitWg) void split_locate_kernel(SplitParams p) {
  uint8_t* const row = s_split_dyn + threadIdx.x * kSplitRow;
  const uint32_t k = blockIdx.x * kSplitWg + threadIdx.x + 1;  // block 0 starts at the stream's first byte
  if (k >= p.nblk) return;
  const uint64_t want = (uint64_t)k << 16;
  if (want >= p.out_at[p.nseg]) return;  // (the elements do not produce that much: the table kernel says so)
  uint32_t lo = 0, hi = p.nseg;  // the last s with out_at[s] <= want (out_at[0] = 0; such an s puts out bytes: out_at[s + 1] > want)
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (p.out_at[mid] <= want) lo = mid;
    else hi = mid;
  }
  const uint32_t s = lo;
  uint32_t pos = p.entry[s];
  const uint32_t seg_lo = s * kSplitSeg;
  const uint32_t seg_hi = seg_lo + kSplitSeg < p.n ? seg_lo + kSplitSeg : p.n;
  if (pos >= seg_hi) {  // (cannot be: a segment that puts out bytes has an entry -- an inconsistency of the split's own
    p.flags[2] = 1;     // machinery is no verdict on the stream: the caller falls back to the one-workgroup walk)
    return;
  }
  split_stage_row(p, row, s);
  uint64_t op = p.out_at[s];
  while (pos < seg_hi && op < want) {
    uint32_t L, size;
    bool nat;
    if (!split_element(row, pos - seg_lo, p.n - pos - 1, &L, &size, &nat)) {
      p.flags[2] = 1;  // (cannot be either: the marked chain's walks decoded every one of these elements -- fall back)
      return;
    }
    op += L;
    pos += size;
  }
  if (op == want) p.blk_in[k] = pos;  // (the element that starts the block -- in this segment or behind it)
  else p.flags[2] = 1;                // an element crosses the boundary: a foreign encoder (or op < want: cannot be)
}

// the blocks as units of the block decoder, from where they start in the stream (blk_in[k], k >= 1; block 0 at 0)
// *bad: 1 = a block without a start (the caller falls back to the serial walk), 2 = the stream is invalid (its
// elements do not produce the declared length, snappy.nim:107-108), 4 = an element straddles a 64 KiB boundary (a
// foreign encoder) or the last walk contradicts the marking (cannot be): fall back -- the speculative split's verdicts, looked
// at by the host once, behind the decode
__global__ __launch_bounds__(256) void split_table_kernel(const uint32_t* blk_in, uint32_t nblk, uint32_t n_tags,
                                                          uint32_t hdr, uint64_t len, uint64_t* in_off, uint32_t* in_len,
                                                          uint64_t* out_off, uint32_t* out_cap, uint32_t* bad,
                                                          const uint64_t* total, const uint32_t* flags) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k == 0 && total) {
    uint32_t v = 0;
    if (*total != len) v |= 2;
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
