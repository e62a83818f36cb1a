// encode_kernel.h -- bit-exact Snappy block encoder for gfx950 (wave64).
//
// Semantics: encodeBlock, snappy/encoder.nim:184-383 (+ emitLiteral :44, emitCopy :81,
// findMatchLength :130, hash :36, tableSize :27), and the chunk choice of encodeFrame,
// snappy/encoder.nim:385-426.  Output is byte-identical to the reference encoder: the greedy
// parse is a strictly sequential algorithm, so one wave EMULATES it exactly for one block:
//
//   * the uint16[16384] hash table lives in LDS (32 KiB), zeroed per block like the reference;
//   * the reference's probe sequence after a literal start is data-independent (skip starts at
//     32, step = skip>>5): 64 lanes take the next 64 probe positions of that sequence, hash
//     them, read the table, write their own position, and detect same-slot collisions inside
//     the wave by reading the slot back; a lane's candidate is the nearest earlier lane with
//     the same slot, else the table value -- exactly what the sequential loop would have seen;
//   * the lowest lane whose candidate matches 4 bytes wins (ballot + ctz); table writes of the
//     lanes after it are rolled back so the table is what the sequential loop leaves behind;
//   * the copy-loop probe at `ip` right after a copy (encoder.nim:371-380) rides along as lane 0
//     of the next round, so one round finds either "copy again" or the next literal + copy;
//   * match extension compares 256 bytes per step across the wave (ballot + ctz);
//   * elements are emitted through a 4 KiB LDS staging buffer and flushed with wide stores;
//   * the bytes around the scan position live in a 2.3 KiB LDS window (refilled with 16-byte
//     loads when the scan leaves it), so a round's only trip to HBM/L2 is the one that cannot be
//     avoided: the candidates, which may lie anywhere earlier in the block.  A candidate is
//     fetched 16 bytes wide, so matches of up to 16 bytes are measured from registers.
//
// The sixteen unrolled probes of encoder.nim:280-309 are the first sixteen steps of the same
// sequence (skip 32..47 => step 1; the guard ipLimit >= ip+16 equals the per-probe guard
// nextIp <= ipLimit for step 1), so one loop form covers both.
#pragma once

#include "common.h"

namespace snappy_hip {

constexpr uint32_t kSeqLen = 320;  // probe-sequence entries (offset passes 65536 at ~250)
constexpr uint32_t kObSize = 4096; // staging bytes
constexpr uint32_t kObCap = kObSize + 3 * 64 + 16;
constexpr uint32_t kWinSize = 2304;  // bytes of input around the scan position kept in LDS

struct EncodeParams {
  const uint8_t* in;
  uint64_t total_len;
  uint32_t block_len;
  int unit;
  uint8_t* slots;
  uint32_t slot_stride;
  uint32_t* sizes;
  uint64_t n_blocks;
  const uint32_t* crc;      // kUnitFrame: masked CRC32C per block (crc kernel ran first)
  const uint32_t* seq_off;  // probe sequence: offset of probe j from the scan start
  const uint32_t* seq_step; // ... and its step (skip >> 5)
  unsigned long long* stats;  // DEBUG: per-section cycle counters (nullptr = off)
  const uint32_t* order;      // workgroup i takes block order[i] (nullptr: block i; crc_pack_kernels.h)
};

__device__ __forceinline__ uint32_t snappy_hash(uint32_t u, uint32_t mask) {
  return ((u * 0x1e35a7bdu) >> (32 - kMaxTableBits)) & mask;  // encoder.nim:36-37
}

__global__ __launch_bounds__(64) void encode_blocks_kernel(EncodeParams prm) {
  __shared__ __attribute__((aligned(16))) uint16_t s_table[kMaxTableSize + 64];  // + one sink slot per lane
  __shared__ __attribute__((aligned(16))) uint8_t s_ob[kObCap];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kWinSize + 32];
  __shared__ uint16_t s_seq_off[kSeqLen];   // saturated at 65535 (such a probe is never valid)
  __shared__ uint16_t s_seq_step[kSeqLen];

  const uint32_t lane = lane_id();
  if (blockIdx.x >= prm.n_blocks) return;
  const uint64_t blk = prm.order ? prm.order[blockIdx.x] : blockIdx.x;

  const uint64_t in_pos = blk * (uint64_t)prm.block_len;
  const uint8_t* in = prm.in + in_pos;
  const uint32_t n = (uint32_t)(prm.total_len - in_pos < prm.block_len ? prm.total_len - in_pos
                                                                       : prm.block_len);
  uint8_t* slot = prm.slots + blk * (uint64_t)prm.slot_stride;

  // ---- unit header -------------------------------------------------------------------------
  uint32_t body_at = 0;  // where the block body starts inside the slot
  uint32_t hl = 0;       // varint bytes
  if (prm.unit != kUnitBody) {
    const uint32_t base = prm.unit == kUnitFrame ? 8 : 0;
    uint32_t v = n;
    uint8_t hb[5];
    while (v >= 0x80) {
      hb[hl++] = (uint8_t)(v | 0x80);
      v >>= 7;
    }
    hb[hl++] = (uint8_t)v;
    if (lane == 0)
      for (uint32_t i = 0; i < hl; i++) slot[base + i] = hb[i];
    body_at = base + hl;
  }
  uint8_t* gout = slot + body_at;

  uint32_t gpos = 0;   // body bytes already in HBM
  uint32_t ofill = 0;  // body bytes waiting in s_ob

  // ---- input window: s_win[i] = in[wq - shift + i], wq a multiple of 16 in "q = p + shift" units --
  const uint32_t shift = (uint32_t)((uintptr_t)in & 15);
  const uint8_t* g0 = in - shift;
  const uint32_t q_end = (shift + n + 15) & ~15u;
  uint32_t wq = 0, wend = 0;  // window covers q in [wq, wend)
  auto fill_window = [&](uint32_t p_first) {
    wq = (p_first + shift) & ~15u;
    wave_fence();
    for (uint32_t i = lane; i < kWinSize / 16; i += 64) {
      const uint32_t q = wq + 16 * i;
      if (q < q_end) *reinterpret_cast<uint4*>(s_win + 16 * i) = *reinterpret_cast<const uint4*>(g0 + q);
    }
    wend = wq + kWinSize < q_end ? wq + kWinSize : q_end;
    wave_fence();
  };
  auto in_window = [&](uint32_t p, uint32_t bytes) -> bool {
    const uint32_t q = p + shift;
    return q >= wq && (q + bytes <= wend || wend == q_end);  // (the block's end is always "inside")
  };

  auto flush = [&]() {
    wave_fence();
    for (uint32_t i = lane * 4; i < ofill; i += 256) {
      if (i + 4 <= ofill) {
        st32u(gout + gpos + i, *reinterpret_cast<const uint32_t*>(s_ob + i));
      } else {
        for (uint32_t k = i; k < ofill; k++) gout[gpos + k] = s_ob[k];
      }
    }
    gpos += ofill;
    ofill = 0;
    wave_fence();
  };
  auto reserve = [&](uint32_t bytes) {
    if (ofill + bytes > kObSize) flush();
  };

  // emitLiteral, encoder.nim:44-73: input[from ..< from+len], 1 <= len <= 65536
  auto emit_literal = [&](uint32_t from, uint32_t len) {
    if (len <= 60 && in_window(from, len)) {  // the common case: one tag byte, payload from the window
      reserve(61);
      if (lane < len) s_ob[ofill + 1 + lane] = s_win[from + shift - wq + lane];
      if (lane == 0) s_ob[ofill] = (uint8_t)((len - 1) << 2);
      ofill += 1 + len;
      return;
    }
    const uint32_t m = len - 1;
    const uint32_t w = m < 60 ? 1 : (m < 256 ? 2 : 3);
    reserve(w + (len <= 1024 ? len : 0));
    if (lane == 0) {
      if (m < 60) {
        s_ob[ofill] = (uint8_t)(m << 2);
      } else if (m < 256) {
        s_ob[ofill] = 60 << 2;
        s_ob[ofill + 1] = (uint8_t)m;
      } else {
        s_ob[ofill] = 61 << 2;
        s_ob[ofill + 1] = (uint8_t)m;
        s_ob[ofill + 2] = (uint8_t)(m >> 8);
      }
    }
    ofill += w;
    if (len <= 64 && in_window(from, len)) {  // the common case: straight from the window
      if (lane < len) s_ob[ofill + lane] = s_win[from + shift - wq + lane];
      ofill += len;
    } else if (len <= 1024) {
      for (uint32_t i = lane * 4; i < len; i += 256) {
        if (i + 4 <= len) {
          st32u(s_ob + ofill + i, ld32u(in + from + i));
        } else {
          for (uint32_t k = i; k < len; k++) s_ob[ofill + k] = in[from + k];
        }
      }
      ofill += len;
    } else {  // long literal: HBM -> HBM
      flush();
      for (uint32_t i = lane * 4; i < len; i += 256) {
        if (i + 4 <= len) {
          st32u(gout + gpos + i, ld32u(in + from + i));
        } else {
          for (uint32_t k = i; k < len; k++) gout[gpos + k] = in[from + k];
        }
      }
      gpos += len;
    }
  };

  // emitCopy, encoder.nim:81-125: 1 <= offset <= 65535, 4 <= length <= 65535
  auto emit_copy = [&](uint32_t offset, uint32_t length) {
    if (length < 68) {  // the common case, branch-free: at most a 60-byte copy2 + one more element
      reserve(8);
      const bool two = length > 64;                      // :105-112
      const uint32_t r = two ? length - 60 : length;     // 4..64
      const bool c2 = r >= 12 || offset >= 2048;         // :114-125
      const uint32_t lo = offset & 255, hi = offset >> 8;
      const uint32_t last = c2 ? ((((r - 1) << 2) | 2) | (lo << 8) | (hi << 16))
                               : (((hi << 5) | ((r - 4) << 2) | 1) | (lo << 8));
      const uint32_t first = ((59u << 2) | 2) | (lo << 8) | (hi << 16);
      const unsigned long long bytes = two ? ((unsigned long long)first | ((unsigned long long)last << 24))
                                           : (unsigned long long)last;
      const uint32_t total = (two ? 3 : 0) + (c2 ? 3 : 2);
      if (lane < total) s_ob[ofill + lane] = (uint8_t)(bytes >> (8 * lane));
      ofill += total;
      return;
    }
    uint32_t k64 = length >= 68 ? (length - 68) / 64 + 1 : 0;  // :97-103
    uint32_t rem = length - 64 * k64;                          // 4..67
    while (k64) {                                              // <= 64 elements per pass
      const uint32_t c = k64 < 64 ? k64 : 64;
      reserve(3 * c);
      if (lane < c) {
        s_ob[ofill + 3 * lane] = (63 << 2) | 2;
        s_ob[ofill + 3 * lane + 1] = (uint8_t)offset;
        s_ob[ofill + 3 * lane + 2] = (uint8_t)(offset >> 8);
      }
      ofill += 3 * c;
      k64 -= c;
    }
    reserve(8);
    if (lane == 0) {
      uint32_t o = ofill;
      if (rem > 64) {  // :105-112
        s_ob[o] = (59 << 2) | 2;
        s_ob[o + 1] = (uint8_t)offset;
        s_ob[o + 2] = (uint8_t)(offset >> 8);
        o += 3;
      }
      const uint32_t r = rem > 64 ? rem - 60 : rem;
      if (r >= 12 || offset >= 2048) {  // :114-120
        s_ob[o] = (uint8_t)(((r - 1) << 2) | 2);
        s_ob[o + 1] = (uint8_t)offset;
        s_ob[o + 2] = (uint8_t)(offset >> 8);
      } else {  // :123-125
        s_ob[o] = (uint8_t)(((offset >> 8) << 5) | ((r - 4) << 2) | 1);
        s_ob[o + 1] = (uint8_t)offset;
      }
    }
    const uint32_t r = rem > 64 ? rem - 60 : rem;
    ofill += (rem > 64 ? 3 : 0) + ((r >= 12 || offset >= 2048) ? 3 : 2);
  };

  if (n < kMinNonLiteral) {  // encoder.nim:227-229
    if (n) emit_literal(0, n);
  } else {
    // ---- per-block setup (encoder.nim:234-245) ---------------------------------------------
    uint32_t table_size = 1u << 8;
    while (table_size < kMaxTableSize && table_size < n) table_size <<= 1;
    const uint32_t mask = table_size - 1;
    for (uint32_t i = lane * 8; i < table_size; i += 64 * 8)
      *reinterpret_cast<uint4*>(&s_table[i]) = make_uint4(0, 0, 0, 0);
    for (uint32_t i = lane; i < kSeqLen; i += 64) {
      const uint32_t o = prm.seq_off[i], st = prm.seq_step[i];
      s_seq_off[i] = (uint16_t)(o < 65535 ? o : 65535);
      s_seq_step[i] = (uint16_t)(st < 65535 ? st : 65535);
    }
    // Lane roles of a round.  After a literal scan that found nothing yet (has0 = false) all 64
    // lanes are scan probes: lane L probes s0 + off[idx0 + L].  Right after a copy that ended at
    // ip (has0 = true) lane 0 is the table insert of ip - 1 (encoder.nim:371: a write that is
    // never a match candidate check), lane 1 the copy-loop probe at ip (:373-380) and lanes 2..63
    // are the first 62 probes of the scan that starts at ip + 1 -- the order of the lanes is the
    // order in which the sequential loop touches the table.
    // The first 64 sequence entries stay in registers in both arrangements (relative to s0 - 2
    // for has0 rounds, to s0 otherwise); later entries come from LDS.
    const uint32_t ra_off = prm.seq_off[lane], ra_step = prm.seq_step[lane];
    const uint32_t rb_off = lane >= 2 ? 2 + prm.seq_off[lane - 2] : lane;
    const uint32_t rb_step = lane >= 2 ? prm.seq_step[lane - 2] : 0;
    const uint32_t need0 = readlane(ra_off, 63) + 28;  // bytes after the first lane's position a fresh round may touch
    wave_fence();
    const uint32_t ip_limit = n - kInputMargin;
    fill_window(0);

    bool has0 = false;        // lanes 0,1 carry the insert of ip-1 and the copy-loop probe at ip = s0-1
    uint32_t next_emit = 0;   // start of the pending literal
    uint32_t s0 = 1;          // position of probe 0 of the current literal scan
    uint32_t idx0 = 0;        // index into the probe sequence of this round's first scan lane
    uint32_t tail_from = 0;   // where the final literal starts

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;  // DEBUG section timers
    uint32_t rounds = 0;
    auto tick = [&](int k) {
      if (prm.stats) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tacc[k] += t - tprev;
        tprev = t;
      }
    };
    if (prm.stats) tprev = __builtin_amdgcn_s_memtime();
    // The elements of a round are written out during the NEXT round's wait for its candidates
    // (the one access per round that goes to L2/HBM): emission only needs these four numbers.
    bool pend = false;
    uint32_t pend_from = 0, pend_pm = 0, pend_off = 0, pend_len = 0;
    auto drain = [&]() {
      if (pend) {
        if (pend_pm > pend_from) emit_literal(pend_from, pend_pm - pend_from);
        emit_copy(pend_off, pend_len);
        pend = false;
      }
    };
    for (;;) {
      rounds++;
      // ---- this round's position per lane, and 16 bytes of input there -------------------------
      uint32_t p;
      bool valid;
      bool pw;  // the 16 bytes at p came from the window (else only pd[0] is loaded)
      uint32_t pd[4] = {0, 0, 0, 0};
      const uint32_t first_probe = has0 ? 1 : 0;  // lanes below it only write the table
      if (idx0 == 0) {
        // fresh round: everything it touches is in the window
        const uint32_t base = has0 ? s0 - 2 : s0;
        if (!in_window(base, need0)) fill_window(base);
        p = base + (has0 ? rb_off : ra_off);
        valid = p + (has0 ? rb_step : ra_step) <= ip_limit;  // encoder.nim:318-321 (lanes 0,1: step 0, ip <= ipLimit)
        pw = valid;
        const uint32_t qa = valid ? (p + shift - wq) : 0;
        const uint32_t* w32 = reinterpret_cast<const uint32_t*>(s_win + (qa & ~3u));
        const uint32_t r0 = w32[0], r1 = w32[1], r2 = w32[2], r3 = w32[3], r4 = w32[4];
        const uint32_t sh8 = (qa & 3) * 8;
        pd[0] = __funnelshift_r(r0, r1, sh8);
        pd[1] = __funnelshift_r(r1, r2, sh8);
        pd[2] = __funnelshift_r(r2, r3, sh8);
        pd[3] = __funnelshift_r(r3, r4, sh8);
      } else {
        // a long scan, far ahead of the window: straight from memory (never with has0)
        const uint32_t si = idx0 + lane;
        p = s0;
        valid = false;
        if (si < kSeqLen) {
          p = s0 + s_seq_off[si];
          valid = p + s_seq_step[si] <= ip_limit;
        }
        pw = false;
        if (valid) pd[0] = ld32u(in + p);
      }
      const uint64_t vmask = ballot(valid);
      if (vmask == 0) {
        drain();
        tail_from = next_emit;
        break;
      }
      const uint32_t d = pd[0];
      tick(0);  // positions + input bytes
      const uint32_t h = snappy_hash(d, mask);
      // (a lane without a position works on its private sink slot instead of being branched around)
      const uint32_t tsink = kMaxTableSize + lane;
      const uint32_t ti = valid ? h : tsink;
      const uint32_t old = s_table[ti];
      wave_fence();
      s_table[ti] = (uint16_t)p;
      wave_fence();
      const uint32_t chk = s_table[ti];
      uint64_t losers = ballot(chk != (p & 0xffffu));

      // candidate as the sequential loop would see it: the position of the nearest earlier lane
      // of this round with my slot, else what the table held
      uint32_t cand = old;
      uint64_t grp = 1ull << lane;  // lanes of this round that share my slot
      const bool any_conflict = losers != 0;
      while (losers) {  // one pass per colliding slot
        const uint32_t j = ctz64(losers);
        const uint32_t hj = readlane(h, j);
        const uint64_t g = ballot(valid && h == hj);
        const bool in_g = valid && h == hj;
        const uint64_t below = g & ((1ull << lane) - 1);
        const uint32_t pred = below ? 63 - (uint32_t)__builtin_clzll(below) : lane;
        const uint32_t pp = __shfl(p, pred, 64);
        if (in_g) {
          grp = g;
          if (below) cand = pp;
        }
        losers &= ~g;
      }
      tick(1);  // table read / write / read back, conflicts

      // the candidate's 16 bytes (cand < p, so cand + 16 <= n): the 4-byte check of
      // encoder.nim:326 and, for window probes, the first 16 bytes of findMatchLength
      uint4 cv;
      __builtin_memcpy(&cv, in + (valid ? cand : 0), 16);
      drain();  // the previous round's literal + copy, while the candidates are in flight
      const uint64_t mm = ballot(valid && lane >= first_probe && cv.x == d);
      const uint32_t m_eff = mm ? ctz64(mm) : 63 - (uint32_t)__builtin_clzll(vmask);
      uint32_t eq = 4;  // equal leading bytes, 4..16 (meaningful where the 4-byte check passed)
      {
        const uint32_t x1 = pd[1] ^ cv.y, x2 = pd[2] ^ cv.z, x3 = pd[3] ^ cv.w;
        const uint32_t e3 = x3 ? 12 + ((uint32_t)__builtin_ctz(x3) >> 3) : 16;
        const uint32_t e2 = x2 ? 8 + ((uint32_t)__builtin_ctz(x2) >> 3) : e3;
        eq = x1 ? 4 + ((uint32_t)__builtin_ctz(x1) >> 3) : e2;
        // the copy-loop probe may sit at ip = n - 15: findMatchLength stops at the block's end
        eq = eq < n - p ? eq : n - p;
      }
      tick(2);  // candidate fetch + compare

      // ---- leave the table as the sequential loop would -------------------------------------
      if (mm) {
        wave_fence();
        s_table[(valid && lane > m_eff) ? h : tsink] = (uint16_t)old;  // never executed there
      }
      if (any_conflict) {
        wave_fence();
        // of several lanes <= m_eff on one slot the last one wrote last
        const uint64_t later = lane >= 63 ? 0 : (grp >> (lane + 1));
        const uint32_t span = m_eff > lane ? m_eff - lane : 0;  // lanes in (lane, m_eff]
        const uint64_t later_in = span >= 64 ? later : (later & ((1ull << span) - 1));
        s_table[(valid && lane <= m_eff && later_in == 0) ? h : tsink] = (uint16_t)p;
      }
      wave_fence();

      if (!mm) {
        if (vmask == ~0ull) {  // every probe missed: the next entries of the sequence
          idx0 += has0 ? 62 : 64;
          has0 = false;
          continue;
        }
        tail_from = next_emit;  // encoder.nim:319-321
        break;
      }
      tick(3);  // table repair

      // ---- literal + copy (encoder.nim:336-359) ---------------------------------------------
      const uint32_t pm = readlane(p, m_eff);
      const uint32_t c = readlane(cand, m_eff);
      tick(4);

      // findMatchLength, encoder.nim:130-182: exact, bounded by n
      const bool wwide = readlane(pw ? 1u : 0u, m_eff) != 0;
      uint32_t matched = wwide ? readlane(eq, m_eff) : 4;
      if (!wwide || (matched == 16 && pm + 16 < n)) {  // longer (or not measured yet): 256 bytes per step
        uint32_t a = c + matched, b = pm + matched;
        for (;;) {
          const uint32_t pb = b + lane * 4;
          uint32_t e4 = 0;
          if (pb < n) {
            const uint32_t avail = n - pb < 4 ? n - pb : 4;
            const uint32_t sh = 4 - avail;  // keep the dword load inside the block
            uint32_t x = (ld32u(in + a + lane * 4 - sh) ^ ld32u(in + pb - sh)) >> (8 * sh);
            e4 = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4;
            if (e4 > avail) e4 = avail;
          }
          const uint64_t mis = ballot(e4 < 4);
          if (mis) {
            const uint32_t f = ctz64(mis);
            matched += 4 * f + readlane(e4, f);
            break;
          }
          matched += 256;
          a += 256;
          b += 256;
        }
      }
      tick(5);  // match length
      pend = true;  // literal input[next_emit ..< pm] + copy (pm - c, matched): emitted next round
      pend_from = next_emit;
      pend_pm = pm;
      pend_off = pm - c;
      pend_len = matched;
      const uint32_t ip = pm + matched;
      if (ip > ip_limit) {  // encoder.nim:362 -- strictly greater
        tail_from = ip;
        break;
      }
      // next round: insert of ip - 1 (:371), probe at ip (:373-380), scan from ip + 1
      has0 = true;
      s0 = ip + 1;
      idx0 = 0;
      next_emit = ip;
    }
    drain();
    if (prm.stats && lane == 0) {
      for (int k = 0; k < 8; k++) atomicAdd(&prm.stats[k], tacc[k]);
      atomicAdd(&prm.stats[8], (unsigned long long)rounds);
    }
    if (tail_from < n) emit_literal(tail_from, n - tail_from);  // encoder.nim:249-253
  }
  flush();
  uint32_t body_len = gpos;

  // ---- unit trailer -------------------------------------------------------------------------
  uint32_t unit_len = body_at + body_len;
  if (prm.unit == kUnitFrame) {  // encodeFrame, encoder.nim:385-426
    const bool compressed = n >= kMinNonLiteral && body_len <= n - n / 8;  // :401, :408
    uint32_t frame_len;
    if (compressed) {
      frame_len = hl + body_len + 4;
    } else {
      frame_len = n + 4;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      for (uint32_t i = lane * 4; i < n; i += 256) {  // stored chunk: raw bytes after the CRC
        if (i + 4 <= n) {
          st32u(slot + 8 + i, ld32u(in + i));
        } else {
          for (uint32_t k = i; k < n; k++) slot[8 + k] = in[k];
        }
      }
    }
    if (lane == 0) {
      const uint32_t crc = prm.crc[blk];
      slot[0] = compressed ? 0x00 : 0x01;
      slot[1] = (uint8_t)frame_len;
      slot[2] = (uint8_t)(frame_len >> 8);
      slot[3] = (uint8_t)(frame_len >> 16);
      slot[4] = (uint8_t)crc;
      slot[5] = (uint8_t)(crc >> 8);
      slot[6] = (uint8_t)(crc >> 16);
      slot[7] = (uint8_t)(crc >> 24);
    }
    unit_len = frame_len + 4;
  }
  if (lane == 0) prm.sizes[blk] = unit_len;
}

}  // namespace snappy_hip
