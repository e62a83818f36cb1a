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
  int first;              // the first round: every segment also walks from its guess, its first byte
  // after the marking
  uint32_t* entry;        // [nseg] the entry of the real chain (all ones: it passes over the segment)
  uint32_t* outb;         // [nseg] output bytes of the elements that start in the segment
  uint32_t* flags;        // [1] invalid element met, [2] an element straddles a 64 KiB boundary of the output
  const uint64_t* out_at; // [nseg + 1] exclusive prefix sum of outb
  uint32_t* blk_in;       // [nblk] stream position where output block k starts
  uint32_t nblk;          // (blk_in's length: a total that is not the declared length must not write beyond it)
};

// Everything the split starts from, in one launch (eight fills and copies of a few bytes each cost 0.2 ms of
// launch gaps in front of the first walk): candidate lists empty but for the root, nothing walked, counters and
// flags zero, no block start known, no unit length, no verdict.
__global__ __launch_bounds__(256) void split_init_kernel(uint32_t* ent_ext, uint64_t n_ent_ext, uint32_t* counters16,
                                                         uint32_t* blk, uint32_t n_blk, uint32_t* out_len, uint32_t n_len,
                                                         uint32_t* bad) {
  const uint64_t t = blockIdx.x * 256ull + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = t; i < n_ent_ext; i += stride) ent_ext[i] = i == 0 ? kSplitTrusted : 0xffffffffu;  // (node 0: position 0)
  for (uint64_t i = t; i < n_blk; i += stride) blk[i] = 0xffffffffu;
  for (uint64_t i = t; i < n_len; i += stride) out_len[i] = 0;
  if (t < 16) counters16[t] = 0;
  if (t == 0) *bad = 0;
}

// One wave per workgroup: its 64 segments are 16 KiB of stream, staged in LDS with coalesced loads
// before the lanes walk them (64 lanes reading their own segment byte by byte straight from memory move
// a cache line per element and lane).
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

__device__ __forceinline__ void split_stage(const SplitParams& p, uint8_t* stage, uint32_t wg_lo) {
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


// `pos` becomes a (trusted) candidate of the segment it lies in, if it is not one already; the slot, or kSplitCand
__device__ __forceinline__ uint32_t split_add(const SplitParams& p, uint32_t pos) {
  uint32_t* e = p.ent + (pos / kSplitSeg) * kSplitCand;
  for (uint32_t c = 0; c < kSplitCand; c++) {
    const uint32_t old = atomicCAS(&e[c], 0xffffffffu, pos | kSplitTrusted);
    if (old == 0xffffffffu) {
      atomicAdd(&p.counters[0], 1u);
      return c;
    }
    if (old == (pos | kSplitTrusted)) return c;
  }
  p.counters[1] = 1;
  return kSplitCand;
}

// One round: every candidate that has not been walked yet is walked, its exit becomes a candidate.
__global__ __launch_bounds__(kSplitWg) void split_walk_kernel(SplitParams p) {
  const uint32_t s = blockIdx.x * kSplitWg + threadIdx.x;
  const uint32_t wg_lo = blockIdx.x * kSplitWg * kSplitSeg;
  uint8_t* const stage = s_split_dyn;
  const bool live = s < p.nseg;
  const uint32_t seg_hi = (s + 1) * kSplitSeg < p.n ? (s + 1) * kSplitSeg : p.n;
  uint32_t todo = 0;  // slots to walk (bit kSplitCand: the guess, which is nobody's successor and has no slot)
  uint32_t ent[kSplitCand + 1];
  ent[kSplitCand] = s * kSplitSeg;
  if (p.first && live && s != 0) todo |= 1u << kSplitCand;
#pragma unroll
  for (uint32_t c = 0; c < kSplitCand; c++) {
    ent[c] = live ? __hip_atomic_load(&p.ent[s * kSplitCand + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    if (ent[c] != 0xffffffffu &&
        __hip_atomic_load(&p.ext[s * kSplitCand + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kSplitPending)
      todo |= 1u << c;
  }
  if (!__ballot(todo != 0)) return;
  split_stage(p, stage, wg_lo);
#pragma unroll
  for (uint32_t c = 0; c <= kSplitCand; c++) {
    if (!((todo >> c) & 1)) continue;
    uint32_t pos = ent[c] & ~kSplitTrusted, out = 0, last_size = 0;
    uint32_t clean = (ent[c] & kSplitTrusted) ? kSplitClean : 0;
    bool bad = false;
    while (pos < seg_hi) {
      uint32_t L, size, tg;
      if (!split_element(stage, wg_lo, p.n, pos, &L, &size, &tg)) {
        bad = true;
        break;
      }
      clean = ((tg & 3) == 3 || ((tg & 3) == 0 && (tg >> 2) >= 62)) ? 0 : clean + 1;
      out += L;
      pos += size;
      last_size = size;
    }
    if (c < kSplitCand) {
      p.ob[s * kSplitCand + c] = out;
      __hip_atomic_store(&p.ext[s * kSplitCand + c], bad ? kSplitBad : (pos == p.n ? kSplitEnd : pos), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bad || pos >= p.n || clean < kSplitClean) continue;
    uint32_t slot = split_add(p, pos);
    // Follow-through.  Long literals back to back (incompressible blocks: one literal of 64 KiB each) are
    // a chain that would be discovered ONE literal per round -- the segment a literal ends in learns the
    // candidate from the walk of the segment the literal starts in.  So a walk that leaves its segment
    // with a long literal and lands on another one does that segment's walk as well (it is that one
    // element), and goes on while it keeps landing on long literals: one trip to memory per literal.
    // (A walk of payload bytes lands on the tag of a long literal one time in a hundred: such chains die.)
    if (last_size < kSplitFollowMin) continue;
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
      p.ob[node] = L;
      __hip_atomic_store(&p.ext[node], p1 == p.n ? kSplitEnd : p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p1 >= p.n) break;
      pos = p1;
      slot = split_add(p, pos);
    }
  }
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
__global__ __launch_bounds__(256) void split_select_kernel(SplitParams p, const uint8_t* reach) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= p.nseg) return;
  uint32_t e = 0xffffffffu, o = 0;
  for (uint32_t c = 0; c < kSplitCand; c++)
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
  uint64_t op = p.out_at[s];
  while (pos < seg_hi) {
    uint32_t L, size, tg;
    if (!split_element(stage, wg_lo, p.n, pos, &L, &size, &tg)) {
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

}  // namespace snappy_hip
