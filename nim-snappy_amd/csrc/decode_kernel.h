// decode_kernel.h -- Snappy block decoder for gfx950 (wave64).
//
// Semantics: decodeAllTags, snappy/decoder.nim:20-155, and the header handling of
// uncompress, snappy.nim:92-110.  One wave decodes one independent unit.
//
//   * the uncompressed block (<= 64 KiB) lives in LDS while it is being built, so that copy
//     elements read their source at LDS latency; it is flushed to HBM once, with 16-byte
//     coalesced stores;
//   * the compressed stream is pulled through an 8 KiB LDS ring with 16-byte coalesced loads;
//   * 64 lanes parse 64 consecutive stream bytes speculatively (each lane decodes "the element
//     that would start at my byte"), a scalar walk keeps the lanes that really are element
//     starts, a wave prefix sum gives every element its output position;
//   * literals are independent and copied first; copy elements are resolved in rounds against
//     a high-water mark (everything below it is final), short ones one element per lane,
//     long ones cooperatively by the whole wave.
//
// Unlike the reference's CPU loops the kernel never writes outside [dst, dst+len) of an
// element: neighbouring units are decoded concurrently.
#pragma once

#include "common.h"

namespace snappy_hip {

constexpr uint32_t kRing = 8192;  // bytes of compressed stream kept in LDS (power of two)
constexpr uint32_t kRingMask = kRing - 1;
constexpr uint32_t kRingAhead = 320;  // bytes that must be resident past ip for one window
constexpr uint32_t kLenClamp = 1u << 25;

struct DecodeParams {
  const uint8_t* in;
  const uint64_t* in_off;
  const uint32_t* in_len;
  uint8_t* out;
  const uint64_t* out_off;
  const uint32_t* out_cap;
  uint32_t* out_len;
  uint32_t* status;
  const uint8_t* kind;  // optional per-unit Unit, overrides `unit`
  uint64_t n_units;
  int unit;
  int dbg;  // timing experiments only: 1 skip literals, 2 skip copies, 4 skip flush
  uint32_t only_status;  // block kernel: if non-zero, handle only units in this state
  const uint32_t* list;  // optional: the units to take (list[-2] = how many); nullptr: unit = workgroup
};

constexpr int kUnitStored = 3;  // internal: verbatim bytes (uncompressed framed chunk)

// Parse the uint32 LEB128 length header of a raw Snappy buffer (snappy.nim:92; stew/leb128).
// Returns bytes consumed, or 0 on truncation / overflow.
__device__ __forceinline__ uint32_t parse_varint32(const uint8_t* p, uint32_t n, uint32_t* val) {
  uint32_t v = 0;
  for (uint32_t i = 0; i < 5 && i < n; i++) {
    uint32_t b = p[i];
    if (i == 4 && (b >> 4)) return 0;
    v |= (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) {
      *val = v;
      return i + 1;
    }
  }
  return 0;
}

// OUT_GLOBAL = false: block kernel, output staged in LDS (units that decode to <= 64 KiB).
// OUT_GLOBAL = true : whole-stream kernel for anything larger; output and back-references go
//                     through global memory (one wave, serial over the stream: SURVEY.md 8e).
template <bool OUT_GLOBAL>
__device__ __forceinline__ void decode_units_body(const DecodeParams& prm, uint64_t unit_idx, uint8_t* s_ring, uint8_t* s_out) {
  const uint32_t lane = lane_id();
  if (unit_idx >= prm.n_units) return;
  const int unit = prm.kind ? (int)prm.kind[unit_idx] : prm.unit;

  if (OUT_GLOBAL) {
    // only units the block kernel handed over
    if (prm.status[unit_idx] != kNeedsStreamKernel) return;
  } else if (prm.only_status) {
    const uint32_t st_now = prm.status[unit_idx];
    // (a unit the index pass decoded itself carried kDoneEarly past the indexed decoder's launches: this launch is the
    // last to look at every unit)
    if (st_now == kDoneEarly && lane == 0) prm.status[unit_idx] = kOk;
    if (st_now != prm.only_status) return;
  }

  const uint8_t* in0 = prm.in + prm.in_off[unit_idx];
  uint32_t n = prm.in_len[unit_idx];
  uint8_t* gout = prm.out + prm.out_off[unit_idx];
  const uint32_t cap = prm.out_cap[unit_idx];

  auto finish = [&](uint32_t st, uint32_t written) {
    if (lane == 0) {
      prm.status[unit_idx] = st;
      prm.out_len[unit_idx] = written;
    }
  };

  // ---- verbatim unit: cooperative copy -----------------------------------------------------
  if (unit == kUnitStored) {
    if (OUT_GLOBAL) return;
    if (n > cap) {
      finish(kBufferTooSmall, 0);
      return;
    }
    for (uint32_t i = lane * 4; i < n; i += 256) {
      if (i + 4 <= n) {
        st32u(gout + i, ld32u(in0 + i));
      } else {
        for (uint32_t k = i; k < n; k++) gout[k] = in0[k];
      }
    }
    finish(kOk, n);
    return;
  }

  // ---- header (snappy.nim:92-102) ----------------------------------------------------------
  uint32_t limit;       // bytes the stream may produce
  bool exact = false;   // RAW: must produce exactly `limit`
  if (unit == kUnitRaw) {
    uint32_t ulen = 0, hdr = 0;
    if (lane == 0) hdr = parse_varint32(in0, n, &ulen);
    hdr = readfirst(hdr);
    ulen = readfirst(ulen);
    if (hdr == 0) {
      finish(kInvalidInput, 0);
      return;
    }
    if (cap < ulen) {
      finish(kBufferTooSmall, 0);
      return;
    }
    if (ulen == 0) {
      finish(hdr == n ? kOk : kInvalidInput, 0);
      return;
    }
    in0 += hdr;
    n -= hdr;
    limit = ulen;
    exact = true;
  } else {
    if (n == 0) {  // decoder.nim:26-27
      finish(kOk, 0);
      return;
    }
    if (cap == 0) {  // decoder.nim:29-30
      finish(kBufferTooSmall, 0);
      return;
    }
    limit = cap;
  }
  if (!OUT_GLOBAL && exact && limit > kMaxBlockLen) {
    finish(kNeedsStreamKernel, 0);
    return;
  }
  const uint32_t win_limit = OUT_GLOBAL ? limit : (limit < kMaxBlockLen ? limit : kMaxBlockLen);

  // ---- stream ring -------------------------------------------------------------------------
  // q-space = stream position + shift, so that 16-byte ring chunks are 16-byte aligned in HBM.
  const uint32_t shift = (uint32_t)((uintptr_t)in0 & 15);
  const uint8_t* g0 = in0 - shift;
  const uint64_t q_end = ((uint64_t)shift + n + 15) & ~15ull;  // exclusive, 16-aligned
  uint64_t q_filled = 0;                                       // ring holds [.., q_filled)

  auto ring_ensure = [&](uint64_t q_ip) {
    uint64_t want = q_ip + kRingAhead;
    if (want > q_end) want = q_end;
    if (q_filled >= want) return;
    uint64_t from = q_filled;
    uint64_t lo = q_ip & ~15ull;
    if (from < lo) from = lo;  // jumped ahead (long literal): restart at ip
    uint64_t to = lo + kRing;
    if (to > q_end) to = q_end;
    wave_fence();
    for (uint64_t q = from + (uint64_t)lane * 16; q < to; q += 64 * 16) {
      uint4 v = *reinterpret_cast<const uint4*>(g0 + q);
      uint32_t idx = (uint32_t)q & kRingMask;
      *reinterpret_cast<uint4*>(s_ring + idx) = v;
      if (idx == 0) *reinterpret_cast<uint4*>(s_ring + kRing) = v;  // mirror for wrap reads
    }
    q_filled = to;
    wave_fence();
  };
  // 4 bytes at q (any alignment) out of the ring; the mirror makes index+3 always valid.
  auto ring32 = [&](uint64_t q) -> uint32_t { return ld32u(s_ring + ((uint32_t)q & kRingMask)); };

  // ---- output accessors --------------------------------------------------------------------
  auto out_ld8 = [&](uint64_t i) -> uint32_t { return OUT_GLOBAL ? gout[i] : s_out[i]; };
  auto out_st8 = [&](uint64_t i, uint32_t v) {
    if (OUT_GLOBAL) gout[i] = (uint8_t)v; else s_out[i] = (uint8_t)v;
  };
  auto out_ld32 = [&](uint64_t i) -> uint32_t { return OUT_GLOBAL ? ld32u(gout + i) : ld32u(s_out + i); };
  auto out_st32 = [&](uint64_t i, uint32_t v) {
    if (OUT_GLOBAL) st32u(gout + i, v); else st32u(s_out + i, v);
  };
  auto out_store_n = [&](uint64_t i, uint32_t v, uint32_t nb) {  // nb in 1..4
    if (nb == 4) {
      out_st32(i, v);
    } else {
      out_st8(i, v & 0xff);
      if (nb > 1) out_st8(i + 1, (v >> 8) & 0xff);
      if (nb > 2) out_st8(i + 2, (v >> 16) & 0xff);
    }
  };
  // Makes the wave's earlier output stores visible to its later loads.
  auto out_fence = [&]() {
    if (OUT_GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    wave_fence();
  };

  uint64_t ip = 0;  // stream position of the next element (uniform)
  uint64_t op = 0;  // output bytes produced (uniform)

  while (ip < n) {
    ring_ensure(ip + shift);

    // ---- speculative per-lane parse: "the element that would start at byte ip+lane" ---------
    const uint64_t p = ip + lane;
    const bool in_range = p < n;
    const uint64_t rem64 = in_range ? (uint64_t)n - p - 1 : 0;  // bytes after the tag
    const uint32_t rem = rem64 > 0xffffffffull ? 0xffffffffu : (uint32_t)rem64;
    const uint64_t q = p + shift;
    const uint32_t w0 = ring32(q), w1 = ring32(q + 4);
    const uint32_t tag = w0 & 0xff;
    const uint32_t b14 = (w0 >> 8) | (w1 << 24);  // stream bytes 1..4 after the tag
    const uint32_t hi6 = tag >> 2;
    const uint32_t t = tag & 3;
    bool err = false;
    bool is_copy = t != 0;
    uint32_t L, size, srcv;  // output length, stream bytes of the element, src pos / offset
    if (t == 0) {            // literal, decoder.nim:42-84
      L = hi6 + 1;
      uint32_t hdr = 1;
      if (L >= 61) {
        if (rem < 61) err = true;  // decoder.nim:54-57
        uint32_t lenlen = L - 60;
        uint32_t m = lenlen == 4 ? 0xffffffffu : ((1u << (8 * lenlen)) - 1);
        L = (b14 & m) + 1;
        if (L == 0) err = true;  // decoder.nim:67-68
        hdr = 1 + lenlen;
      }
      if (!err && rem - (hdr - 1) < L) err = true;  // decoder.nim:78
      srcv = (uint32_t)p + hdr;
      size = hdr + L;
    } else if (t == 1) {  // decoder.nim:86-94
      if (rem < 1) err = true;
      L = 4 + (hi6 & 7);
      srcv = ((tag & 0xe0) << 3) | (b14 & 0xff);
      size = 2;
    } else if (t == 2) {  // decoder.nim:95-102
      if (rem < 2) err = true;
      L = 1 + hi6;
      srcv = b14 & 0xffff;
      size = 3;
    } else {  // decoder.nim:103-109
      if (rem < 4) err = true;
      L = 1 + hi6;
      srcv = b14;
      size = 5;
    }

    // ---- scalar walk: which lanes are real element starts ----------------------------------
    const uint64_t errmask = ballot(err);
    uint64_t chain = 0;
    uint64_t pos = 0;
    bool bad = false;
    while (pos < 64 && ip + pos < n) {
      chain |= 1ull << pos;
      if ((errmask >> pos) & 1) {
        bad = true;
        break;
      }
      pos += readlane(size, (uint32_t)pos);
    }
    if (bad) {
      finish(kInvalidInput, 0);
      return;
    }
    const bool mine = (chain >> lane) & 1;

    // ---- output positions -------------------------------------------------------------------
    uint32_t tot;
    const uint32_t Lc = mine ? (L < kLenClamp ? L : kLenClamp) : 0;
    const uint64_t dst = op + wave_excl_scan(Lc, lane, &tot);
    bool bad_off = mine && is_copy && (srcv == 0 || (uint64_t)srcv > dst);  // decoder.nim:112
    bool bad_room = mine && (dst + L > win_limit);                          // :77-79, :127-128
    const uint64_t fails = ballot(bad_off || bad_room);
    if (fails) {
      uint32_t f = ctz64(fails);
      bool f_off = (ballot(bad_off) >> f) & 1;
      // an element past the 64 KiB LDS window of a bigger unit is not an error: hand over
      finish((!OUT_GLOBAL && !f_off && limit > win_limit) ? kNeedsStreamKernel : kInvalidInput, 0);
      return;
    }
    const uint32_t last = 63 - (uint32_t)__builtin_clzll(chain);
    const uint64_t op_next = (uint64_t)__shfl((uint32_t)(dst - op), last, 64) + op + readlane(L, last);

    // ---- literals: no dependencies ----------------------------------------------------------
    if (!(SNAPPY_DBG(prm) & 1)) {
      const bool lit = mine && !is_copy;
      if (lit && L <= 16) {  // one element per lane, source in the ring
        const uint64_t qs = (uint64_t)srcv + shift;
#pragma unroll
        for (uint32_t k = 0; k < 16; k += 4) {
          if (k < L) {
            uint32_t v = ring32(qs + k);
            uint32_t nb = L - k < 4 ? L - k : 4;
            out_store_n(dst + k, v, nb);
          }
        }
      }
      uint64_t longs = ballot(lit && L > 16);
      while (longs) {  // whole wave per element, source straight from HBM
        const uint32_t e = ctz64(longs);
        longs &= longs - 1;
        const uint32_t eL = readlane(L, e);
        const uint64_t es = readlane(srcv, e);
        const uint64_t ed = op + readlane((uint32_t)(dst - op), e);
        for (uint64_t i = lane * 4; i < eL; i += 256) {
          if (i + 4 <= eL) {
            out_st32(ed + i, ld32u(in0 + es + i));
          } else {
            for (uint64_t k = i; k < eL; k++) out_st8(ed + k, in0[es + k]);
          }
        }
      }
    }
    out_fence();

    // ---- copies: rounds against the high-water mark ----------------------------------------
    if (!(SNAPPY_DBG(prm) & 2)) {
      uint64_t pending = ballot(mine && is_copy);
      const uint64_t src = dst - srcv;  // valid for copy lanes (offset <= dst checked above)
      while (pending) {
        const uint32_t first = ctz64(pending);
        const uint32_t fL = readlane(L, first);
        if (fL > 16) {
          // long copy: all lanes, one byte each; overlap (offset < length) replicates the
          // pattern of the `offset` bytes before dst (decoder.nim:130-151)
          const uint32_t foff = readlane(srcv, first);
          const uint64_t fd = op + readlane((uint32_t)(dst - op), first);
          const uint64_t fs = fd - foff;
          if (lane < fL) {
            uint32_t j = lane;
            if (foff < fL) {
              const uint32_t rcp = 65536u / foff + 1;  // exact floor(j / foff) for j < 64
              j = lane - ((lane * rcp) >> 16) * foff;
            }
            uint32_t v = out_ld8(fs + j);
            out_st8(fd + lane, v);
          }
          pending &= pending - 1;
          out_fence();
          continue;
        }
        const uint64_t hwm = op + readlane((uint32_t)(dst - op), first);  // all below is final
        const bool pend_me = (pending >> lane) & 1;
        const bool ready = pend_me && L <= 16 && (lane == first || src + L <= hwm);
        if (ready) {
          if (!OUT_GLOBAL && srcv >= L) {  // source entirely below dst (s_out is padded)
            uint32_t v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
              if (4 * k < L) v[k] = out_ld32(src + 4 * k);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
              if (4 * k < L) out_store_n(dst + 4 * k, v[k], L - 4 * k < 4 ? L - 4 * k : 4);
          } else {  // byte path: overlapping short copy (offset < length replicates), and
                    // every short copy of the whole-stream kernel (no reads past the buffer)
            uint32_t j = 0;
            uint32_t v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
              if (k < L) v[k] = out_ld8(src + j);
              j = j + 1 == srcv ? 0 : j + 1;
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; k++)
              if (k < L) out_st8(dst + k, v[k]);
          }
        }
        pending &= ~ballot(ready);
        out_fence();
      }
    }

    op = op_next;
    ip += pos;
  }

  if (exact && op != limit) {  // snappy.nim:107-108
    finish(kInvalidInput, 0);
    return;
  }

  // ---- flush the finished block to HBM, 16 bytes per lane -----------------------------------
  if (!OUT_GLOBAL && !(SNAPPY_DBG(prm) & 4)) {
    wave_fence();
    const uint32_t total = (uint32_t)op;
    if (((uintptr_t)gout & 15) == 0) {  // block-aligned output (the batch layouts): b128 both sides
      for (uint32_t i = lane * 16; i < total; i += 64 * 16) {
        if (i + 16 <= total) {
          *reinterpret_cast<uint4*>(gout + i) = *reinterpret_cast<const uint4*>(s_out + i);
        } else {
          for (uint32_t k = i; k < total; k++) gout[k] = s_out[k];
        }
      }
    } else {
      for (uint32_t i = lane * 4; i < total; i += 64 * 4) {
        if (i + 4 <= total) {
          st32u(gout + i, *reinterpret_cast<const uint32_t*>(s_out + i));
        } else {
          for (uint32_t k = i; k < total; k++) gout[k] = s_out[k];
        }
      }
    }
  }
  finish(kOk, (uint32_t)op);
}

// prm.list == nullptr: workgroup i takes unit i.  Otherwise (the block decoder's fallback behind the indexed decoder,
// round 5): the units are those of a list a small kernel has made (list[-2] = how many; decode_finish_kernel), taken by
// however many workgroups were launched -- 65 536 workgroups of this kernel's 68 KB of LDS that look at a status and
// leave cost 51 us a step; a list that is empty costs a load.
template <bool OUT_GLOBAL>
__global__ __launch_bounds__(64) void decode_units_kernel(DecodeParams prm) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_out[OUT_GLOBAL ? 16 : kMaxBlockLen + 16];
  if (prm.list) {
    const uint32_t n_list = __hip_atomic_load(prm.list - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t i = blockIdx.x; i < n_list; i += gridDim.x) {
      decode_units_body<OUT_GLOBAL>(prm, prm.list[i], s_ring, s_out);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (one wave: its LDS of this unit is done with)
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    decode_units_body<OUT_GLOBAL>(prm, blockIdx.x, s_ring, s_out);
  }
}

// Behind the indexed decoder's launches: a unit the index pass decoded itself carried kDoneEarly past them and becomes
// kOk here; the units the indexed decoder declined (kNeedsOnePass) are listed for decode_units_kernel<false>.
__global__ __launch_bounds__(256) void decode_finish_kernel(uint32_t* status, uint64_t n_units, uint32_t* list) {
  const uint64_t u = blockIdx.x * 256ull + threadIdx.x;
  const uint32_t st = u < n_units ? status[u] : kOk;
  if (st == kDoneEarly) status[u] = kOk;
  const bool mine = st == kNeedsOnePass;
  const uint64_t m = __ballot(mine);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(list - 2, (uint32_t)__builtin_popcountll(m));
  base = __shfl(base, 0, 64);
  if (mine) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1))] = (uint32_t)u;
}

}  // namespace snappy_hip
