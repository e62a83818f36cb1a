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
//   wave 1  "resolver": consumes the PREVIOUS step's copy list 64 elements at a time.  Runs of
//           consecutive copies with one offset (how the encoder splits long matches,
//           encoder.nim:97-112) are merged back into one copy.  Copies are then resolved in
//           rounds against a high-water mark: everything below the destination of the first
//           unresolved copy is final, so every copy whose source ends below it can run
//           now, one per lane; the first unresolved one can always run.  Long (merged) copies
//           are done by all 64 lanes, overlapping ones by widening the period first.
//   all     flush the finished block with 16-byte stores.
//
// One workgroup barrier per step separates "list k is complete" from "list k is consumed".
#pragma once

#include "common.h"
#include "index_kernel.h"

namespace snappy_hip {

constexpr uint32_t kD2Threads = 256;
constexpr uint32_t kD2Ring = 4096;
constexpr uint32_t kListCap = 1024;  // a 2 KiB chunk holds at most 1024 copy elements

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
};

__global__ __launch_bounds__(kD2Threads) void decode_indexed_kernel(Decode2Params prm) {
  __shared__ __attribute__((aligned(16))) uint8_t s_out[kMaxBlockLen + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kD2Ring + 16];
  __shared__ uint32_t s_cp[2][kListCap];  // dst | offset << 16
  __shared__ uint8_t s_cl[2][kListCap];   // length 1..64
  __shared__ uint32_t s_cnt[2];
  __shared__ uint32_t s_err;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint64_t u = blockIdx.x;
  if (u >= prm.n_units) return;
  if (prm.status[u] != kOk) return;       // the index pass already decided this unit
  const uint32_t total = prm.out_len[u];
  if (total == 0) return;

  const uint8_t* in0 = prm.in + prm.in_off[u];
  uint32_t n = prm.in_len[u];
  if (prm.unit == kUnitRaw) {             // skip the varint (validated by the index pass)
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
  const uint64_t q_end = ((uint64_t)shift + n + 15) & ~15ull;
  const uint32_t n_chunks = (n + kChunk - 1) / kChunk;
  const uint32_t n_regions = (n + kRegion - 1) / kRegion;

  if (tid == 0) {
    s_err = 0;
    s_cnt[0] = 0;
    s_cnt[1] = 0;
  }

  // ring[q & 4095] = stream byte q - shift; at the start of step s it holds q in
  // [2048 s, 2048 s + 4096)
  auto ring_store = [&](uint64_t q, uint4 v) {
    const uint32_t i = (uint32_t)q & (kD2Ring - 1);
    *reinterpret_cast<uint4*>(s_ring + i) = v;
    if (i == 0) *reinterpret_cast<uint4*>(s_ring + kD2Ring) = v;
  };
  auto ring32 = [&](uint64_t q) -> uint32_t { return ld32u(s_ring + ((uint32_t)q & (kD2Ring - 1))); };

  if (wave == 0) {  // prologue: first 4 KiB of the stream
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint64_t q = (uint64_t)(lane + 64 * i) * 16;
      if (q < q_end) ring_store(q, *reinterpret_cast<const uint4*>(g0 + q));
    }
  }
  __syncthreads();

  for (uint32_t s = 0; s <= n_chunks; s++) {
    if (wave == 0 && s < n_chunks) {
      // =================================== front end ===========================================
      const uint32_t buf = s & 1;
      const uint64_t c0 = (uint64_t)s * kChunk;
      // loads for the step after next, consumed at the end of this step
      uint4 pre[2];
      uint64_t pq[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        pq[i] = c0 + kD2Ring + (uint64_t)(lane + 64 * i) * 16;
        pre[i] = make_uint4(0, 0, 0, 0);
        if (pq[i] < q_end) pre[i] = *reinterpret_cast<const uint4*>(g0 + pq[i]);
      }
      const uint32_t r = s * 64 + lane;
      const uint32_t ie = r < n_regions ? idx[r] : kIdxNone;
      const uint32_t e_off = ie & 63;
      const uint32_t ncopy = (ie >> 6) & 31;
      uint32_t dst = ie >> 11;
      uint32_t ctot;
      uint32_t slot = wave_excl_scan(e_off == kIdxNone ? 0 : ncopy, lane, &ctot);
      if (lane == 0) s_cnt[buf] = ctot;
      const uint32_t slot_end = slot + (e_off == kIdxNone ? 0 : ncopy);
      const uint32_t dst0 = dst;

      const uint64_t rs = c0 + (uint64_t)lane * kRegion;
      uint64_t pos = rs + e_off;
      bool live = e_off != kIdxNone && pos < n;
      bool big = false;          // a literal longer than 64 bytes ends my region: done below
      uint32_t big_dst = 0, big_len = 0;
      uint64_t big_src = 0;
      bool bad = false;
      while (ballot(live)) {
        if (live) {
          const uint64_t q = pos + shift;
          const uint32_t w0 = ring32(q), w1 = ring32(q + 4);
          const uint32_t tag = w0 & 0xff;
          const uint32_t b14 = (w0 >> 8) | (w1 << 24);
          bool is_copy;
          uint32_t L, size, hdr, off;
          decode_element(tag, b14, 0xffffffffu, &is_copy, &L, &size, &hdr, &off);
          if (is_copy) {
            const bool bad_off = off == 0 || off > dst;  // decoder.nim:112
            bad = bad || bad_off;
            s_cp[buf][slot] = bad_off ? (dst | (1u << 16)) : (dst | (off << 16));
            s_cl[buf][slot] = bad_off ? 0 : (uint8_t)L;  // length 0 = skipped by the resolver
            slot++;
          } else if (L <= 64) {
            const uint64_t qs = q + hdr;
#pragma unroll
            for (uint32_t k = 0; k < 16; k += 4) {
              if (k < L) {
                const uint32_t v = ring32(qs + k);
                const uint32_t nb = L - k < 4 ? L - k : 4;
                if (nb == 4) {
                  st32u(s_out + dst + k, v);
                } else {
                  s_out[dst + k] = (uint8_t)v;
                  if (nb > 1) s_out[dst + k + 1] = (uint8_t)(v >> 8);
                  if (nb > 2) s_out[dst + k + 2] = (uint8_t)(v >> 16);
                }
              }
            }
            for (uint32_t k = 16; k < L; k += 4) {
              const uint32_t v = ring32(qs + k);
              const uint32_t nb = L - k < 4 ? L - k : 4;
              if (nb == 4) {
                st32u(s_out + dst + k, v);
              } else {
                s_out[dst + k] = (uint8_t)v;
                if (nb > 1) s_out[dst + k + 1] = (uint8_t)(v >> 8);
                if (nb > 2) s_out[dst + k + 2] = (uint8_t)(v >> 16);
              }
            }
          } else {
            big = true;
            big_dst = dst;
            big_len = L;
            big_src = pos + hdr;
          }
          dst += L;
          pos += size;
          live = pos < rs + kRegion && pos < n;
        }
      }
      // long literals: whole wave, straight from HBM (at most one per region)
      uint64_t bigs = ballot(big);
      while (bigs) {
        const uint32_t e = ctz64(bigs);
        bigs &= bigs - 1;
        const uint32_t eL = readlane(big_len, e);
        const uint32_t ed = readlane(big_dst, e);
        const uint64_t es = ((uint64_t)readlane((uint32_t)(big_src >> 32), e) << 32) |
                            readlane((uint32_t)big_src, e);
        for (uint32_t i = lane * 4; i < eL; i += 256) {
          if (i + 4 <= eL) {
            st32u(s_out + ed + i, ld32u(in0 + es + i));
          } else {
            for (uint32_t k = i; k < eL; k++) s_out[ed + k] = in0[es + k];
          }
        }
      }
      if (ballot(bad) && lane == 0) s_err = 1;
      {  // DEBUG consistency checks of the index against the walk
        const bool had = e_off != kIdxNone;
        if (ballot(had && slot != slot_end) && lane == 0) atomicOr(&s_err, 2u);
        // my final dst must be the first dst of the next region that has an entry
        const uint64_t hm = ballot(had);
        const uint64_t later = lane == 63 ? 0 : hm & ~((2ull << lane) - 1);
        const uint32_t nx = later ? ctz64(later) : 64;
        const uint32_t nd = __shfl(dst0, nx & 63, 64);
        if (ballot(had && nx != 64 && nd != dst) && lane == 0) atomicOr(&s_err, 4u);
      }
      // the ring slots of this chunk are free now: land the prefetched 2 KiB
#pragma unroll
      for (int i = 0; i < 2; i++)
        if (pq[i] < q_end) ring_store(pq[i], pre[i]);
    } else if (wave == 1 && s >= 1) {
      // =================================== resolver ==============================================
      const uint32_t buf = (s - 1) & 1;
      const uint32_t count = s_cnt[buf];
      for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        const uint32_t i = b0 + lane;
        const uint32_t len = i < count ? s_cl[buf][i] : 0;
        const bool act = len != 0;
        const uint32_t e = act ? s_cp[buf][i] : 0;
        const uint32_t dst = e & 0xffff, off = e >> 16;
        // merge runs: same offset, destination continues the previous copy
        const uint32_t p_e = __shfl_up(e, 1, 64), p_len = __shfl_up(len, 1, 64);
        const bool cont = act && lane > 0 && (p_e >> 16) == off && (p_e & 0xffff) + p_len == dst;
        const uint64_t heads = ballot(act && !cont);
        uint32_t tot;
        const uint32_t excl = wave_excl_scan(len, lane, &tot);
        const uint64_t later = lane == 63 ? 0 : heads & ~((2ull << lane) - 1);
        const uint32_t nxt = later ? ctz64(later) : 64;
        const uint32_t nxt_excl = __shfl(excl, nxt & 63, 64);
        const uint32_t mlen = (nxt == 64 ? tot : nxt_excl) - excl;  // merged length (heads only)
        const uint32_t src = dst - off;

        uint64_t pending = heads;
        while (pending) {
          const uint32_t first = ctz64(pending);
          const uint32_t fL = readlane(mlen, first);
          const uint32_t fd = readlane(dst, first);
          if (fL > 16) {
            // ---- long copy by the whole wave -------------------------------------------------
            const uint32_t foff = readlane(off, first);
            const uint32_t fs = fd - foff;
            uint32_t done = 0;
            uint32_t period = foff;
            if (foff < 256) {
              // overlap: out[fd+i] = out[fs + i mod foff]; write up to 512 bytes bytewise, after
              // which a multiple of the period that is >= 256 serves as the copy distance
              const uint32_t rcp = 65536u / foff + 1;
              const uint32_t head_len = fL < 512 ? fL : 512;
              for (uint32_t i = lane; i < head_len; i += 64) {
                uint32_t j = i;
                if (foff <= i) {
                  // i < 512, foff < 256: (i*rcp)>>16 == i/foff whenever i*foff < 65536... use
                  // the safe form for the upper half
                  uint32_t qd = (i * rcp) >> 16;
                  if (qd * foff > i) qd--;
                  j = i - qd * foff;
                  if (j >= foff) j -= foff;
                }
                s_out[fd + i] = s_out[fs + j];
              }
              wave_fence();
              done = head_len;
              period = foff * ((255 + foff) / foff);  // multiple of foff in [256, 511]
            }
            for (uint32_t i = done + lane * 4; i < fL; i += 256) {
              // period >= 256: the 256 bytes of one pass only read bytes of earlier passes
              const uint32_t v = ld32u(s_out + fd + i - period);
              const uint32_t nb = fL - i < 4 ? fL - i : 4;
              if (nb == 4) {
                st32u(s_out + fd + i, v);
              } else {
                s_out[fd + i] = (uint8_t)v;
                if (nb > 1) s_out[fd + i + 1] = (uint8_t)(v >> 8);
                if (nb > 2) s_out[fd + i + 2] = (uint8_t)(v >> 16);
              }
              wave_fence();
            }
            pending &= pending - 1;
            continue;
          }
          // ---- short copies: one per lane, everything whose source is final ------------------
          const bool pend_me = (pending >> lane) & 1;
          const bool ready = pend_me && mlen <= 16 && (lane == first || src + mlen <= fd);
          if (ready) {
            if (off >= mlen) {
              uint32_t v[4];
#pragma unroll
              for (uint32_t k = 0; k < 4; k++)
                if (4 * k < mlen) v[k] = ld32u(s_out + src + 4 * k);
#pragma unroll
              for (uint32_t k = 0; k < 4; k++) {
                if (4 * k < mlen) {
                  const uint32_t nb = mlen - 4 * k < 4 ? mlen - 4 * k : 4;
                  if (nb == 4) {
                    st32u(s_out + dst + 4 * k, v[k]);
                  } else {
                    s_out[dst + 4 * k] = (uint8_t)v[k];
                    if (nb > 1) s_out[dst + 4 * k + 1] = (uint8_t)(v[k] >> 8);
                    if (nb > 2) s_out[dst + 4 * k + 2] = (uint8_t)(v[k] >> 16);
                  }
                }
              }
            } else {  // overlapping short copy: pattern of `off` bytes
              uint32_t j = 0;
              uint32_t v[16];
#pragma unroll
              for (uint32_t k = 0; k < 16; k++) {
                if (k < mlen) v[k] = s_out[src + j];
                j = j + 1 == off ? 0 : j + 1;
              }
#pragma unroll
              for (uint32_t k = 0; k < 16; k++)
                if (k < mlen) s_out[dst + k] = (uint8_t)v[k];
            }
          }
          pending &= ~ballot(ready);
          wave_fence();
        }
      }
    }
    __syncthreads();
  }

  // ---- flush ------------------------------------------------------------------------------------
  if (s_err) {
    if (tid == 0) prm.status[u] = s_err == 1 ? kInvalidInput : 1000 + s_err;
    if (s_err == 1) return;
  }
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
constexpr uint32_t kMaxRegionsPerUnit = (6 * kMaxBlockLen + 2 * kChunk) / kRegion + 4;

__global__ void region_counts_kernel(const uint32_t* in_len, uint64_t n_units, uint32_t* counts) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n_units) return;
  const uint64_t c = ((uint64_t)in_len[i] + kRegion - 1) / kRegion + 1;
  counts[i] = c < kMaxRegionsPerUnit ? (uint32_t)c : kMaxRegionsPerUnit;
}

}  // namespace snappy_hip
