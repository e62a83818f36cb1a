// sparse_kernel.h -- the block decoder for units of FEW, LONG elements: element-parallel, HBM to HBM.
//
// Semantics: decodeAllTags, snappy/decoder.nim:20-155 (validity of the element stream: the index pass; copy offsets,
// decoder.nim:112: here).
//
// The indexed decoder (decode2_kernel.h) is built for text: thousands of elements of a few bytes each, resolved byte
// by byte in an LDS window, in steps of 2 KiB of stream.  A unit that is a handful of long literals with stretches
// of 64-byte copies behind them (what encodeBlock makes of repeated long strings: emitCopy cuts every match into
// copy2 elements, encoder.nim:97-120) spends its time there in the steps' fixed costs -- one wave fetching a literal of
// kilobytes, one wave extending a run, a barrier per step: ~65 us a block, 0.16 of the HBM roofline.  Such a unit has
// at most a few hundred elements, and each is a contiguous piece of the output: this kernel gives every element a
// thread.
//
//   1. The index (index_kernel.h) holds the element starts of every 16 bytes of stream and the output position of the
//      first: every thread takes a contiguous range of index entries, counts its starts (workgroup prefix sum: the
//      element's number), decodes them from the stream and leaves (output position, payload position or offset) in LDS.
//   2. Literals are copied stream -> output: short ones by their thread, long ones by the whole workgroup.
//   3. A copy can run when everything its source range overlaps has been written.  Levels: a literal has level 0, a
//      copy 1 + the highest level among the elements its source range overlaps (found by binary search in the output
//      positions); a few rounds of relaxation settle them -- a run of copies behind its literal has as many levels as
//      the string has repetitions.  Then level by level, a thread per copy, 16 bytes at a time from the output itself
//      (device-scope loads, a fence and a barrier between two levels).
//
// Anything that does not fit -- more than kSparseLevels levels (deep chains are the indexed decoder's business) --
// is handed to the whole-block instantiation of the indexed decoder (kNeedsWindow).
//
// There is no launch of its own: the INDEX PASS (index_kernel.h) counts a unit's elements, and the wave that has just
// written the index of a unit of 2 .. kSparseMax elements (a stream longer than the indexed decoder's stream ring)
// decodes it on the spot, in the LDS of its tables.  One wave per unit is what this procedure wants anyway -- a unit's
// work is a chain of a few trips to memory, and what hides them is units in flight -- and its waits run beside the other
// waves' table building, which is bound by the vector ALU.  (As a kernel behind the indexed decoder's launches it took
// 0.43 ms of a 4 GiB step.)
//
// Included by index_kernel.h, behind the index entry helpers it uses.
#pragma once

#include "common.h"

namespace snappy_hip {

constexpr uint32_t kSparseMax = 832;     // elements (its LDS arrays lie in the index pass's table: 8 448 bytes)
constexpr uint32_t kSparseThreads = 64;  // ONE wave per unit: a unit's work is a chain of a few trips to memory, and what
                                         // hides them is many units in flight per CU (7 KiB of LDS each: 22)
constexpr uint32_t kSparseLevels = 12;   // relaxation rounds = levels a unit may have
constexpr uint32_t kSparseLong = 256;    // literals from here on are copied by the whole wave
constexpr uint32_t kSparsePer = kSparseMax / kSparseThreads;  // elements per lane

__device__ __forceinline__ uint4 ld16u(const uint8_t* p) {  // 16 bytes, any alignment
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ void st16u(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }
// L <= 64 bytes, any alignment, source and destination apart: every load goes out before the first store (a piece
// that ends at the last byte overlaps the piece before it instead of a loop over the tail's bytes)
__device__ __forceinline__ void copy_apart64(uint8_t* dst, const uint8_t* src, uint32_t L) {
  if (L >= 16) {
    uint4 v[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)
      if (16 * q + 16 <= L) v[q] = ld16u(src + 16 * q);
    const uint4 vl = ld16u(src + L - 16);
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)
      if (16 * q + 16 <= L) st16u(dst + 16 * q, v[q]);
    st16u(dst + L - 16, vl);
  } else if (L >= 8) {
    uint64_t a, b;
    __builtin_memcpy(&a, src, 8);
    __builtin_memcpy(&b, src + L - 8, 8);
    __builtin_memcpy(dst, &a, 8);
    __builtin_memcpy(dst + L - 8, &b, 8);
  } else if (L >= 4) {
    uint32_t a, b;
    __builtin_memcpy(&a, src, 4);
    __builtin_memcpy(&b, src + L - 4, 4);
    __builtin_memcpy(dst, &a, 4);
    __builtin_memcpy(dst + L - 4, &b, 4);
  } else {
    const uint32_t a = L > 0 ? src[0] : 0, b = L > 1 ? src[1] : 0, c = L > 2 ? src[2] : 0;
    if (L > 0) dst[0] = (uint8_t)a;
    if (L > 1) dst[1] = (uint8_t)b;
    if (L > 2) dst[2] = (uint8_t)c;
  }
}

// The LDS it works in: kSparseLds bytes, 4-byte aligned (the caller's; nothing of the caller's survives).
constexpr uint32_t kSparseWork = 256;  // regions with element starts a unit may have
constexpr uint32_t kSparseLds = kSparseMax * 4 + kSparseMax * 2 + kSparseMax + 2 * kSparseWork * 4 + 64 * 2 + 65 * 4 +
                                (kSparseLevels + 2 + 1) * 4 + 16;
// Decodes the unit whose tag stream is in0[0 .. n) (index entries idx[], total output bytes) into gout; ONE wave.
// Returns kOk, kInvalidInput (a bad copy offset, decoder.nim:112) or kNeedsWindow (not for this procedure).
// (inlined into its one caller: the compiler then knows the arrays are LDS)
__device__ __forceinline__ uint32_t sparse_decode_unit(const uint8_t* in0, uint32_t n, const uint32_t* idx, uint8_t* gout,
                                                       uint32_t total, uint32_t* lds) {
  uint32_t* const s_src = lds;                                              // literal: stream position of its payload; copy: offset
  uint32_t* const s_work = s_src + kSparseMax;                              // work lists (see below)
  uint32_t* const s_lpre = s_work + 2 * kSparseWork;                        // running count of the long literals' 16-byte pieces
  uint32_t* const s_hist = s_lpre + 65;
  uint32_t* const s_nlong_p = s_hist + kSparseLevels + 2;
  uint16_t* const s_dst = reinterpret_cast<uint16_t*>(s_nlong_p + 1);      // first output byte of element e (the total closes the last one)
  uint16_t* const s_long = s_dst + kSparseMax;                              // long literals (element numbers)
  uint8_t* const s_lvl = reinterpret_cast<uint8_t*>(s_long + 64);           // 0 literal, 1.. a copy's level, 255 not known yet
  // work lists, so that every trip to memory is made by 64 busy lanes: first the regions of the stream in which
  // elements start (region | first element's number << 16, and the region's index entry); the same 2 KiB later hold a
  // relaxation round's results, and then the copies in the order of their levels
  constexpr uint32_t kWork = kSparseWork;
  static_assert(2 * kWork * 4 >= kSparseMax * 2, "the copies' order fits the region list's space");
  uint8_t* const s_new = reinterpret_cast<uint8_t*>(s_work);   // (a relaxation round's results, before they replace the levels)
  uint16_t* const s_order = reinterpret_cast<uint16_t*>(s_work);  // (the copies by level: the region list is done by then)
#define s_nlong (*s_nlong_p)
  const uint32_t lane = lane_id();
  {
    auto dst_at = [&](uint32_t e, uint32_t count) -> uint32_t { return e < count ? (uint32_t)s_dst[e] : total; };
    if (lane < kSparseLevels + 2) s_hist[lane] = 0;
    if (lane == 0) s_nlong = 0;
    // ---- 1a. the regions in which elements start, in stream order, each with its first element's number ----
    // (region it * 64 + lane: coalesced, sixteen loads in flight a lane -- a loop that loads, waits and adds is a chain
    // of trips to memory)
    const uint32_t n_regions = (n + kSub - 1) / kSub;
    constexpr uint32_t kBatch = 16;
    uint32_t count = 0, n_work = 0;  // (uniform) elements / regions so far
    for (uint32_t r0 = 0; r0 < n_regions; r0 += kBatch * kSparseThreads) {
      uint32_t ent[kBatch];
#pragma unroll
      for (uint32_t k = 0; k < kBatch; k++) {
        const uint32_t r = r0 + k * kSparseThreads + lane;
        ent[k] = r < n_regions ? idx[r] : 0;
      }
#pragma unroll
      for (uint32_t k = 0; k < kBatch; k++) {
        const uint32_t bm = idx_starts(ent[k]);
        const uint64_t ne = ballot(bm != 0);
        if (ne == 0) continue;
        uint32_t tot;
        const uint32_t before = wave_excl_scan((uint32_t)__builtin_popcount(bm), lane, &tot);
        const uint32_t pos = n_work + (uint32_t)__builtin_popcountll(ne & ((1ull << lane) - 1));
        if (bm && pos < kWork) {
          s_work[2 * pos] = (r0 + k * kSparseThreads + lane) | ((count + before) << 16);
          s_work[2 * pos + 1] = ent[k];
        }
        count += tot;
        n_work += (uint32_t)__builtin_popcountll(ne);
      }
    }
    if (count > kSparseMax || count == 0 || n_work > kWork) return kNeedsWindow;  // (not what this procedure is for)
    wave_fence();
    // ---- 1b. the elements: a lane per region of the list ----
    bool bad = false;
    for (uint32_t w0 = 0; w0 < n_work; w0 += kSparseThreads) {
      const uint32_t wi = w0 + lane;
      if (wi >= n_work) continue;
      const uint32_t we = s_work[2 * wi], ent = s_work[2 * wi + 1];
      const uint32_t rp = (we & 0xffffu) * kSub;
      uint32_t base = we >> 16, bm = idx_starts(ent), d = ent >> 16;
      // the region's 16 bytes and the 8 behind them hold every byte the elements that start in it are decoded
      // from: one trip for all of them (at the unit's very end: byte by byte)
      uint64_t w[3] = {0, 0, 0};
      if (rp + 24 <= n) {
        __builtin_memcpy(w, in0 + rp, 24);
      } else {
        for (uint32_t q = 0; q < 24; q++)
          if (rp + q < n) w[q >> 3] |= (uint64_t)in0[rp + q] << (8 * (q & 7));
      }
      while (bm) {
        const uint32_t bo = (uint32_t)__builtin_ctz(bm);
        bm &= bm - 1;
        const uint32_t p = rp + bo;
        // bytes bo .. bo + 7 of the 24
        const uint64_t wa_ = bo < 8 ? w[0] : w[1], wb_ = bo < 8 ? w[1] : w[2];
        const uint32_t sh = (bo & 7) * 8;
        const uint64_t el = sh ? (wa_ >> sh) | (wb_ << (64 - sh)) : wa_;
        // the element at p (decoder.nim:42-109; the index pass has checked that it lies inside the stream)
        const uint32_t tag = (uint32_t)el & 0xffu, b14 = (uint32_t)(el >> 8);
        const uint32_t ty = tag & 3, hi6 = tag >> 2;
        uint32_t L, src;
        if (ty == 0) {
          const uint32_t lenlen = hi6 >= 60 ? hi6 - 59 : 0;
          L = lenlen ? (b14 & (0xffffffffu >> (32 - 8 * lenlen))) + 1 : hi6 + 1;
          src = p + 1 + lenlen;
          s_lvl[base] = 0;
          if (L >= kSparseLong) {
            const uint32_t kk = atomicAdd(&s_nlong, 1u);
            if (kk < 64) s_long[kk] = (uint16_t)base;
          }
        } else {
          L = ty == 1 ? 4 + (hi6 & 7) : 1 + hi6;
          src = ty == 1 ? (((tag & 0xe0u) << 3) | (b14 & 0xffu)) : (ty == 2 ? (b14 & 0xffffu) : b14);
          bad = bad || src == 0 || src > d;  // decoder.nim:112
          s_lvl[base] = 255;
        }
        s_dst[base] = (uint16_t)d;
        s_src[base] = src;
        d += L;
        base++;
      }
    }
    wave_fence();
    if (ballot(bad)) return kInvalidInput;
#ifndef SPARSE_NO_LITS
    // ---- 2. literals: stream -> output ----
    const uint32_t nl_all = readfirst(s_nlong);
    const uint32_t n_long = nl_all < 64 ? nl_all : 64;
    const bool long_overflow = nl_all > 64;  // (then every literal by its own lane: slow, right)
    for (uint32_t e = lane; e < count; e += kSparseThreads) {
      if (s_lvl[e] != 0) continue;
      const uint32_t d = s_dst[e], L = dst_at(e + 1, count) - d;
      if (L < kSparseLong || long_overflow) {
        const uint8_t* src = in0 + s_src[e];
        uint32_t i = 0;
        for (; i + 64 < L; i += 64) copy_apart64(gout + d + i, src + i, 64);
        copy_apart64(gout + d + i, src + i, L - i);
      }
    }
    if (!long_overflow && n_long) {
      // The long literals as ONE list of 16-byte pieces: per literal the destination-aligned windows that lie inside
      // it, and one unaligned piece at each end (which may overlap a window: the same bytes twice).  A lane takes
      // pieces lane, lane + 64, ..., eight in flight.
      if (lane < n_long) {
        const uint32_t e = s_long[lane];
        const uint32_t d = s_dst[e], end = dst_at(e + 1, count);
        const uintptr_t g = (uintptr_t)gout;
        const uint32_t a0 = (uint32_t)(((g + d + 15) & ~(uintptr_t)15) - g), a1 = (uint32_t)(((g + end) & ~(uintptr_t)15) - g);
        s_lpre[lane + 1] = (a1 > a0 ? (a1 - a0) / 16 : 0) + 2;
      }
      if (lane == 0) s_lpre[0] = 0;
      wave_fence();
      if (lane == 0)
        for (uint32_t k = 0; k < n_long; k++) s_lpre[k + 1] += s_lpre[k];
      wave_fence();
      const uint32_t n_pieces = s_lpre[n_long];
      for (uint32_t j0 = 0; j0 < n_pieces; j0 += 8 * kSparseThreads) {
        uint32_t so[8], dof[8];
        uint4 v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
          uint32_t j = j0 + q * kSparseThreads + lane;
          j = j < n_pieces ? j : j0 + lane < n_pieces ? j0 + lane : 0;  // (clamped: a piece twice instead of a branch)
          uint32_t k = 0;
          while (k + 1 < n_long && s_lpre[k + 1] <= j) k++;
          const uint32_t e = s_long[k];
          const uint32_t d = s_dst[e], end = dst_at(e + 1, count), pj = j - s_lpre[k], np = s_lpre[k + 1] - s_lpre[k];
          const uintptr_t g = (uintptr_t)gout;
          const uint32_t a0 = (uint32_t)(((g + d + 15) & ~(uintptr_t)15) - g);
          // piece 0: the literal's first 16 bytes; the last piece: its last 16; between them the aligned windows
          dof[q] = pj == 0 ? d : (pj == np - 1 ? end - 16 : a0 + 16 * (pj - 1));
          so[q] = s_src[e] + (dof[q] - d);
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) v[q] = ld16u(in0 + so[q]);
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) st16u(gout + dof[q], v[q]);
      }
    }
#endif
    // ---- 3. the copies' levels ----
    // (a copy's source range [d - off, d - off + L) ends at its own first byte at the latest: when the copy overlaps
    // itself -- off < L -- the part of the range at and behind d is its own output, written as it goes)
    bool settled = false;
    uint32_t maxlvl = 0;
    for (uint32_t it = 0; it < kSparseLevels && !settled; it++) {
      bool left = false;
      for (uint32_t e = lane; e < count; e += kSparseThreads) {
        uint32_t res = 255;
        if (s_lvl[e] == 255) {
          const uint32_t d = s_dst[e], L = dst_at(e + 1, count) - d, off = s_src[e];
          const uint32_t lo = d - off, hi = (lo + L < d ? lo + L : d);  // [lo, hi): what I read of OTHER elements
          // the element that holds byte lo: the last one that starts at or below it
          uint32_t a = 0, b = e;  // (it is below me: lo < d)
          while (a + 1 < b) {
            const uint32_t m = (a + b) >> 1;
            if (s_dst[m] <= lo) a = m; else b = m;
          }
          uint32_t lv = 0;
          for (uint32_t j = a; j < e && s_dst[j] < hi; j++) {
            const uint32_t l = s_lvl[j];
            lv = l > lv ? l : lv;
          }
          if (lv != 255) res = lv + 1;
          else left = true;
        }
        s_new[e] = (uint8_t)res;
      }
      wave_fence();  // (every look at the levels is done: now they change)
      uint32_t mx = 0;
      for (uint32_t e = lane; e < count; e += kSparseThreads) {
        const uint32_t r = s_new[e];
        if (r != 255) {
          s_lvl[e] = (uint8_t)r;
          mx = r > mx ? r : mx;
        }
      }
      wave_fence();
      uint32_t wmx;
      (void)wave_excl_scan_max(mx, lane, &wmx);
      maxlvl = wmx > maxlvl ? wmx : maxlvl;
      settled = ballot(left) == 0;
    }
    if (!settled || maxlvl > kSparseLevels) return kNeedsWindow;  // deeper than this goes: the indexed decoder's whole-block instantiation
#ifdef SPARSE_NO_COPIES
    maxlvl = 0;
#endif
    // ---- 4. level by level: output -> output ----
    // the copies in the order of their levels (a counting sort; the order inside a level does not matter)
    for (uint32_t e = lane; e < count; e += kSparseThreads)
      if (s_lvl[e]) atomicAdd(&s_hist[s_lvl[e] + 1], 1u);
    wave_fence();
    if (lane == 0)
      for (uint32_t l = 1; l <= kSparseLevels + 1; l++) s_hist[l] += s_hist[l - 1];  // [l]: first slot of level l
    wave_fence();
    uint32_t seg[kSparseLevels + 2];
#pragma unroll
    for (uint32_t l = 0; l < kSparseLevels + 2; l++) seg[l] = readfirst(s_hist[l]);
    wave_fence();
    for (uint32_t e = lane; e < count; e += kSparseThreads)
      if (s_lvl[e]) s_order[atomicAdd(&s_hist[s_lvl[e]], 1u)] = (uint16_t)e;
    wave_fence();
    for (uint32_t lv = 1; lv <= maxlvl; lv++) {
      // (what the literals and the levels below wrote is read by other lanes of this wave: the stores are waited
      // for -- one wave, one CU, one vector cache; a device-scope fence here would write the XCD's L2 back)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      uint32_t s0 = 0, s1 = 0;
#pragma unroll
      for (uint32_t l = 1; l <= kSparseLevels; l++) {
        s0 = l == lv ? seg[l] : s0;
        s1 = l == lv ? seg[l + 1] : s1;
      }
      for (uint32_t x = s0 + lane; x < s1; x += kSparseThreads) {
        const uint32_t e = s_order[x];
        const uint32_t d = s_dst[e], L = dst_at(e + 1, count) - d, off = s_src[e];
        uint8_t* const dst = gout + d;
        const uint8_t* const src = gout + d - off;
        if (off >= L && L <= 64) {  // source and destination apart
          copy_apart64(dst, src, L);
        } else {
          // the copy overlaps itself (its output is periodic with period off, decoder.nim:130-151) -- or is longer
          // than a copy element can be (never): byte by byte, each store waited for before the next load
          for (uint32_t k = 0; k < L; k++) {
            dst[k] = src[k];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          }
        }
      }
    }
    wave_fence();
  }
  return kOk;
#undef s_nlong
}

// ---- two more kinds of unit the index pass's wave finishes itself (the indexed decoder has the same two fast paths, for
// ---- when this one is switched off; here their trips to memory run beside the other waves' table building, and a
// ---- framed stream's checksum of them can start while the indexed decoder works on the rest) ------------------------

// The unit is ONE literal (what encodeBlock makes of incompressible data, encoder.nim:249-253): its payload, `total`
// bytes at src, goes straight to gout.  One wave; destination-aligned 16-byte pieces, eight in flight a lane.
__device__ __forceinline__ void early_literal_unit(const uint8_t* src, uint8_t* gout, uint32_t total) {
  const uint32_t lane = lane_id();
  const uint32_t head = (uint32_t)((16 - ((uintptr_t)gout & 15)) & 15);
  const uint32_t hd = head < total ? head : total;
  if (lane < hd) gout[lane] = src[lane];
  const uint32_t body = (total - hd) & ~15u;
  for (uint32_t i = lane * 16; i < body; i += 8 * 64 * 16) {
    uint32_t ix[8];
    uint4 v[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) ix[q] = i + q * 1024 < body ? i + q * 1024 : i;
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) v[q] = ld16u(src + hd + ix[q]);
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) *reinterpret_cast<uint4*>(gout + hd + ix[q]) = v[q];
  }
  if (hd + body + lane < total) gout[hd + body + lane] = src[hd + body + lane];
}

// The unit is one literal followed by copies that all have ONE offset (what encodeBlock makes of a period -- zeros, a
// repeating pattern, a ramp: every match is found at the same distance and emitCopy cuts it into copy2 elements,
// encoder.nim:97-120): its output is periodic behind the literal, out[x] = out[x - offset], so it is written from one
// image of the period.  The index pass has validated the stream as a sequence of elements and its total, so "every third
// byte behind the literal is a copy2 tag with that offset" proves that those are the element starts (decoder.nim:112:
// 1 <= offset <= the literal's length).  One wave; lds: 8 224 bytes (the stream, at most 4 KiB, and the image).
// Returns false if the unit is not of this kind (nothing written).
// total == kPeriodTotalUnknown (round 5): called BEFORE the unit's chain has been walked -- the stream's shape is then all
// there is to go by, and it is enough: one literal (its length bytes obeying the 61-byte rule, decoder.nim:54-57) followed
// by nothing but whole copy2 elements of one offset <= the literal's length is a valid element sequence, its total is the
// literal's length plus the sum of the copies' lengths (a reduction over at most 1 365 tags), and that total must be
// what the caller wants (exact: the declared length, snappy.nim:107-108; else at most `limit`).  Anything else returns
// false and the walk decides.  A period unit then never walks its chain (~40 of the ~125 us it held a wave slot).
constexpr uint32_t kPeriodTotalUnknown = 0xffffffffu;
constexpr uint32_t kPeriodLdsBytes = 4096 + 4096 + 32;  // the stream, the image, what the image's last reader looks at behind it
__device__ __forceinline__ bool early_period_unit(const uint8_t* in0, uint32_t n, uint8_t* gout, uint32_t total, uint32_t* lds,
                                                  uint32_t limit = 0, bool exact = false, uint32_t* total_out = nullptr) {
  const uint32_t lane = lane_id();
  const bool total_known = total != kPeriodTotalUnknown;
  if (n > 4096 || n < 5 || ((uintptr_t)gout & 15) != 0) return false;
  uint8_t* const s_str = reinterpret_cast<uint8_t*>(lds);
  uint8_t* const s_rep = s_str + 4096;
  for (uint32_t i = lane * 16; i < n; i += 64 * 16) {  // (the whole stream: at most four pieces a lane)
    if (i + 16 <= n) {
      *reinterpret_cast<uint4*>(s_str + i) = ld16u(in0 + i);
    } else {
      for (uint32_t k = i; k < n; k++) s_str[k] = in0[k];
    }
  }
  wave_fence();
  const uint32_t tag = s_str[0], hi6 = tag >> 2;
  const uint32_t lenlen = hi6 >= 60 ? hi6 - 59 : 0;
  if ((tag & 3) != 0 || 1 + lenlen + 3 > n) return false;
  uint32_t L0 = hi6 + 1;
  if (lenlen) {
    uint32_t b = 0;
    for (uint32_t k = 0; k < lenlen; k++) b |= (uint32_t)s_str[1 + k] << (8 * k);
    L0 = b + 1;
  }
  const uint32_t h = 1 + lenlen;
  if (!total_known && lenlen && n - 1 < 61) return false;  // (decoder.nim:54-57: the walk gives the verdict)
  if (L0 == 0 || (total_known && L0 >= total) || L0 > kMaxBlockLen || h + L0 + 3 > n || (n - h - L0) % 3 != 0) return false;
  const uint32_t q0 = h + L0;  // the first copy
  const uint32_t roff = s_str[q0 + 1] | ((uint32_t)s_str[q0 + 2] << 8);
  if ((s_str[q0] & 3) != 2 || roff < 1 || roff > L0 || roff > 4096) return false;
  const uint32_t nrec = (n - q0) / 3;
  bool ok = true;
  for (uint32_t k = lane; k < nrec; k += 64) {
    const uint32_t t = q0 + 3 * k;
    ok = ok && (s_str[t] & 3) == 2 && (s_str[t + 1] | ((uint32_t)s_str[t + 2] << 8)) == roff;
  }
  if (ballot(!ok)) return false;
  if (!total_known) {
    uint32_t sum = 0;
    for (uint32_t k = lane; k < nrec; k += 64) sum += ((uint32_t)s_str[q0 + 3 * k] >> 2) + 1;
    uint32_t all;
    (void)wave_excl_scan(sum, lane, &all);
    total = L0 + all;
    if (total > kMaxBlockLen || (exact ? total != limit : total > limit)) return false;
    if (total_out) *total_out = total;
  }
  // image of the period: rep[i] = out[L0 - offset + i mod offset] for i < M + 16, M a multiple of the offset of about 4 KiB
  const uint32_t M = roff * (4096 / roff);  // (> 1024: the writer's phase advances by 1024 a trip)
  const uint32_t pb = q0 - roff;            // stream position of the period's first byte
  for (uint32_t i0 = lane * 16; i0 < M + 16; i0 += 64 * 16) {
    uint32_t r = i0 % roff;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
      s_rep[i0 + j] = s_str[pb + r];
      r = r + 1 == roff ? 0 : r + 1;
    }
  }
  wave_fence();
  const uint32_t* const rep32 = reinterpret_cast<const uint32_t*>(s_rep);
  uint32_t x = lane * 16;
  uint32_t ix = x >= L0 ? (x - L0) % M : 0;  // phase of x in the image (kept up to date from the first x >= L0 on)
  bool phased = x >= L0;
  for (; x < total; x += 1024) {
    if (x >= L0 && !phased) {
      ix = (x - L0) % M;
      phased = true;
    }
    if (x >= L0 && x + 16 <= total) {
      const uint32_t a = ix >> 2, sh8 = (ix & 3) * 8;
      const uint32_t r0 = rep32[a], r1 = rep32[a + 1], r2 = rep32[a + 2], r3 = rep32[a + 3], r4 = rep32[a + 4];
      *reinterpret_cast<uint4*>(gout + x) = make_uint4(__funnelshift_r(r0, r1, sh8), __funnelshift_r(r1, r2, sh8),
                                                       __funnelshift_r(r2, r3, sh8), __funnelshift_r(r3, r4, sh8));
    } else {
      for (uint32_t k = x; k < x + 16 && k < total; k++)
        gout[k] = k < L0 ? s_str[h + k] : s_rep[(k - L0) % M];
    }
    if (phased) {
      ix += 1024;
      ix = ix >= M ? ix - M : ix;
    }
  }
  return true;
}

}  // namespace snappy_hip
