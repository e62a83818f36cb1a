// encode_kernel.h -- bit-exact Snappy block encoder for gfx950 (wave64).
//
// Semantics: encodeBlock, snappy/encoder.nim:184-383 (+ emitLiteral :44, emitCopy :81,
// findMatchLength :130, hash :36, tableSize :27), and the chunk choice of encodeFrame,
// snappy/encoder.nim:385-426.  Output is byte-identical to the reference encoder: the greedy
// parse is a strictly sequential algorithm, so one wave EMULATES it exactly for one block:
//
//   * the uint16[16384] hash table lives in LDS (32 KiB), zeroed per block like the reference -- for the four blocks per CU
//     that the LDS holds; since round 5 every workgroup carries a SECOND wave whose block keeps its table in global
//     memory (32 KiB that stay in the XCD's L2; plain 16-bit loads and stores, see "the table elsewhere" below);
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
//   * elements are emitted through a 2 KiB LDS staging buffer and flushed with wide stores;
//   * the bytes around the scan position live in a 1.5 KiB LDS window (refilled with 16-byte
//     loads when the scan leaves it), so a round's only trip to HBM/L2 is the one that cannot be
//     avoided: the candidates, which may lie anywhere earlier in the block.  A candidate is
//     fetched 16 bytes wide, so matches of up to 16 bytes are measured from registers.
//
// The sixteen unrolled probes of encoder.nim:280-309 are the first sixteen steps of the same
// sequence (skip 32..47 => step 1; the guard ipLimit >= ip+16 equals the per-probe guard
// nextIp <= ipLimit for step 1), so one loop form covers both.
#pragma once

#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "encode_kernel.h relies on LDS atomics serving the lanes of one address in ascending lane order -- checked at run time, but verified (tools/probes/mskor_probe.hip, cmpst_probe.hip) on gfx950 only"
#endif
// -DENC_INJECT_ORDER_FAULT=<k> (tests/test_gpu_faults.py builds such a library; never the shipped one): every k-th
// look at an order-of-service check finds it failed, so that the recovery code behind the checks -- which the
// hardware never sends anybody into -- runs under the parity tests.
#ifdef ENC_INJECT_ORDER_FAULT
#define ENC_ORDER_FAULT() ((++enc_inject % (ENC_INJECT_ORDER_FAULT)) == 0)
#else
#define ENC_ORDER_FAULT() false
#endif

namespace snappy_hip {

constexpr uint32_t kSeqLen = 320;  // probe-sequence entries (offset passes 65536 at ~250)
// LDS per workgroup: the table (32 KiB + a sink slot per lane), per wave staging + window + 36 bytes, and the second wave's
// scratch (2 KiB): 40 104 bytes -- four workgroups a CU (LDS is handed out in pieces of 1 280 bytes: 4 x 40 960 is all of
// it), two waves each.
#ifndef ENC_OB   // (-DENC_OB / ENC_WIN / ENC_WAVES: A/B experiments, tools/mkvariant.sh)
#define ENC_OB 1024
#endif
#ifndef ENC_WIN
#define ENC_WIN 1280
#endif
#ifndef ENC_WAVES
#define ENC_WAVES 2
#endif
constexpr uint32_t kObSize = ENC_OB; // staging bytes
constexpr uint32_t kObLitMax = kObSize / 4;  // a literal up to this length goes through the staging buffer
constexpr uint32_t kObFlushAt = kObSize - kObLitMax - 3 - 3 * 64 - 128;  // drain() flushes above this fill; one of its steps adds <= kObLitMax + 3 bytes
constexpr uint32_t kObCap = kObSize + 3 * 64 + 16;
constexpr uint32_t kWinSize = ENC_WIN;  // bytes of input around the scan position kept in LDS
constexpr uint32_t kEncWaves = ENC_WAVES;  // waves per workgroup: wave 0's table is the LDS one, wave 1's lies in global memory
constexpr uint32_t kEncScratchBits = 10;   // wave 1's scratch for finding the lanes of a round that share a table slot
constexpr uint32_t kEncScratch = 1u << kEncScratchBits;  // (16-bit entries: 2 KiB)

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
  const uint32_t* order;      // the i-th block taken is order[i] (nullptr: block i; crc_pack_kernels.h)
  uint32_t* queue;            // [0] how many blocks have been taken (zeroed before the launch)
  uint16_t* gtables;          // one 32 KiB table per workgroup for its second wave (nullptr: no second waves)
  uint32_t g_per4;            // of every four workgroups of an XCD, this many run their second wave (0..4)
  uint32_t dbg;               // DEBUG builds only
};

__device__ __forceinline__ uint32_t snappy_hash(uint32_t u, uint32_t mask) {
  return ((u * 0x1e35a7bdu) >> (32 - kMaxTableBits)) & mask;  // encoder.nim:36-37
}

// One block, one wave.  GT = false: the table is s_table (LDS).  GT = true, "the table elsewhere": the table is gtab, 32 KiB
// of global memory that only this wave touches -- plain 16-bit stores and loads that go to the L2 (sc1: past the CU's
// vector cache).  A wave's vector memory operations execute in order, so "read the slot, write my position, read it back"
// is ONE trip: the first read sees the table as it was, the second who wrote last.  Lanes that share a slot find each other
// through it (a lane that lost writes once more, then each of two reads the other; three and more go through a loop over
// registers) -- no assumption about which lane of a store instruction wins.  Everything else is the same procedure.
// (the LDS pointers carry their address space: left generic, half of the accesses through them come out as flat_
// instructions, which a wave's ds_ instructions are not ordered against)
typedef __attribute__((address_space(3))) uint8_t enc_lds8;
typedef __attribute__((address_space(3))) uint16_t enc_lds16;
typedef __attribute__((address_space(3))) uint32_t enc_lds32;
typedef uint32_t enc_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) enc_v4 enc_lds128;
template <bool GT>
__device__ __forceinline__ void encode_one_block(const EncodeParams& prm, const uint64_t blk, enc_lds16* const s_table,
                                                 uint16_t* const gtab, enc_lds16* const s_gx, enc_lds8* const s_ob,
                                                 enc_lds8* const s_win, enc_lds32* const s_cold, enc_lds32* const s_walk_p) {
  // (north_star's layout -- the whole block staged in LDS next to the table, one block per CU -- was built and
  // measured in round 2: bit-exact, 3.8x slower; profiles/README.md)
  constexpr uint32_t kWin = kWinSize;
#define s_walk (*s_walk_p)

  const uint32_t lane = lane_id();
  // the table: loads by index, stores by the lanes that have something to store
  auto T_ld = [&](uint32_t idx) -> uint32_t {
    if constexpr (GT) return __hip_atomic_load(gtab + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_ushort sc1
    else return s_table[idx];
  };
  auto T_st = [&](bool pred, uint32_t idx, uint32_t val) {
    if constexpr (GT) {
      if (pred) gtab[idx] = (uint16_t)val;
    } else {
      s_table[pred ? idx : kMaxTableSize + lane] = (uint16_t)val;  // (the others: their sink slot, no branch)
    }
  };

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
    for (uint32_t i = lane; i < kWin / 16; i += 64) {
      const uint32_t q = wq + 16 * i;
      if (q < q_end) *reinterpret_cast<enc_lds128*>(s_win + 16 * i) = *reinterpret_cast<const enc_v4*>(g0 + q);
    }
    wend = wq + kWin < q_end ? wq + kWin : q_end;
    wave_fence();
  };
  auto in_window = [&](uint32_t p, uint32_t bytes) -> bool {
    const uint32_t q = p + shift;
    return q >= wq && (q + bytes <= wend || wend == q_end);  // (the block's end is always "inside")
  };

  // ---- output: elements collect in s_ob and leave with wide stores ----------------------------
  // Everything that is written goes through drain() below, which has the one flush in the kernel.
  uint32_t finished = 0;                // the parse is over; what is pending is all that is left
  // pending output, written in this order:
  uint32_t dpend = 0;                   // (1) the elements of a fresh round (position-parallel)
  uint64_t dp_ms = 0;                   //     lanes where a copy starts
  bool dp_lit = false;                  //     per lane: its byte is a literal byte
  uint32_t dp_len = 0, dp_off = 0, dp_byte = 0;  // per lane: copy length, copy offset, the byte at the lane's position
  // (2)-(4) are set on the rare paths only (a long scan, a copy of more than 64 bytes, the block's end): they live in
  // LDS, behind one flag in a register -- the parse loop is short of scalar registers (every spilled one is a
  // v_readlane / v_writelane on the round's chain of instructions)
  struct ColdU32 {
    volatile enc_lds32* p;
    __device__ __forceinline__ operator uint32_t() const { return readfirst(*p); }
    __device__ __forceinline__ ColdU32& operator=(uint32_t v) {
      if (lane_id() == 0) *p = v;
      return *this;
    }
    __device__ __forceinline__ ColdU32& operator=(const ColdU32& o) { return *this = (uint32_t)o; }  // (the value, not the slot)
  };
  if (lane < 8) s_cold[lane] = 0;
#ifndef ENC_NO_CMPST_WALK
  if (lane == 0) s_walk = 1;
#endif
  wave_fence();
  uint32_t cold_any = 0;                // one of (2)-(4) is set
  ColdU32 lit_from{&s_cold[0]}, lit_len{&s_cold[1]};    // (2) one literal ...
  ColdU32 cp_off{&s_cold[2]}, cp_len{&s_cold[3]};       // (3) ... one copy ...
  ColdU32 lit2_from{&s_cold[4]}, lit2_len{&s_cold[5]};  // (4) ... and the block's final literal (encoder.nim:249-253)
  // The elements of a fresh round.  One prefix sum places them all: a lane holding a literal byte
  // writes it (the first lane of a run also the tag, emitLiteral encoder.nim:44-73: runs are <= 63
  // bytes), a lane where a copy starts writes the copy (emitCopy :81-125; lengths <= 64 only, so
  // one element of 2 or 3 bytes).  No lane writes more than three bytes; a lane with fewer writes
  // the rest to its sink behind the buffer -- no branches.
  // (selects written as bit masks: left as ?: the compiler makes EXEC-mask branches of them, five per round, each a
  // scalar save / restore pair and a taken branch on a lone wave's path)
  auto emit_round = [&]() {
    dpend = false;
    const uint64_t LIT = ballot(dp_lit);
    const uint64_t here = LIT >> lane;  // bit 0: this lane, bit k: lane + k
    const uint32_t lit = (uint32_t)here & 1u;
    const uint32_t run_start = lit & ~(uint32_t)((LIT << 1) >> lane) & 1u;
    const uint32_t rl = ctz64(~here);   // literal bytes from here to the next copy
    const uint32_t is_copy = (uint32_t)(dp_ms >> lane) & 1u;
    const uint32_t length = dp_len, offset = dp_off;
    const uint32_t c2m = 0u - (uint32_t)((length >= 12) | (offset >= 2048));  // :114-125
    const uint32_t cv2 = (((length - 1) << 2) | 2) | (offset << 8);
    const uint32_t cv1 = (((offset >> 8) << 5) | ((length - 4) << 2) | 1) | ((offset & 255) << 8);
    const uint32_t cval = (cv2 & c2m) | (cv1 & ~c2m);
    const uint32_t t2m = 0u - (uint32_t)(rl > 60);
    const uint32_t lv2 = (60u << 2) | ((rl - 1) << 8) | (dp_byte << 16);
    const uint32_t lv1 = ((rl - 1) << 2) | (dp_byte << 8);
    const uint32_t rsm = 0u - run_start;
    const uint32_t lval = (((lv2 & t2m) | (lv1 & ~t2m)) & rsm) | (dp_byte & ~rsm);
    const uint32_t litm = 0u - lit;
    const uint32_t val = (lval & litm) | (cval & ~litm);
    const uint32_t nb_l = 1 + run_start + (run_start & t2m & 1u);  // a byte; the run's first: its tag of one or two bytes too
    const uint32_t nb_c = is_copy * (2 + (c2m & 1u));
    const uint32_t nb = (nb_l & litm) | (nb_c & ~litm);
    uint32_t total;
    const uint32_t at = ofill + wave_excl_scan(nb, lane, &total);
    const uint32_t sink = kObSize + 64 + lane;
    s_ob[nb > 0 ? at : sink] = (uint8_t)val;
    s_ob[nb > 1 ? at + 1 : sink] = (uint8_t)(val >> 8);
    s_ob[nb > 2 ? at + 2 : sink] = (uint8_t)(val >> 16);
    ofill += total;
  };
  auto drain = [&]() {
    if (dpend && ofill <= kObFlushAt && !cold_any && !finished) {  // the usual case
      emit_round();
      return;
    }
    bool want_flush = false;
    // (the rare items, all six in one LDS trip; they are used up -- or kept in these registers -- until the loop ends)
    uint32_t c_lit_from, c_lit_len, c_cp_off, c_cp_len, c_lit2_from, c_lit2_len;
    {
      wave_fence();
      const uint32_t cw = s_cold[lane & 7];
      c_lit_from = readlane(cw, 0), c_lit_len = readlane(cw, 1), c_cp_off = readlane(cw, 2), c_cp_len = readlane(cw, 3);
      c_lit2_from = readlane(cw, 4), c_lit2_len = readlane(cw, 5);
    }
    for (;;) {
      if (want_flush || ofill > kObFlushAt) {  // (a step below adds at most kObLitMax + 3 bytes)
        wave_fence();
        for (uint32_t i = lane * 4; i < ofill; i += 256) {
          if (i + 4 <= ofill) {
            st32u(gout + gpos + i, *reinterpret_cast<const enc_lds32*>(s_ob + i));
          } else {
            for (uint32_t k = i; k < ofill; k++) gout[gpos + k] = s_ob[k];
          }
        }
        gpos += ofill;
        ofill = 0;
        want_flush = false;
        wave_fence();
      }
      if (dpend) {
        emit_round();
        continue;
      }
      if (c_lit_len) {  // emitLiteral, encoder.nim:44-73: input[from ..< from+len], 1 <= len <= 65536
        const uint32_t from = c_lit_from, len = c_lit_len;
        if (len > kObLitMax && ofill) {  // a long literal goes from HBM to HBM, behind what is waiting
          want_flush = true;
          continue;
        }
        c_lit_len = 0;
        const uint32_t m = len - 1;
        const uint32_t w = m < 60 ? 1 : (m < 256 ? 2 : 3);
        const uint32_t t0 = m < 60 ? (m << 2) : (m < 256 ? (60u << 2) : (61u << 2));
        const uint32_t tag = t0 | ((m & 255) << 8) | ((m >> 8) << 16);
        if (len <= kObLitMax) {
          if (lane < w) s_ob[ofill + lane] = (uint8_t)(tag >> (8 * lane));
          ofill += w;
          if (len <= 64 && in_window(from, len)) {  // the common case: straight from the window
            if (lane < len) s_ob[ofill + lane] = s_win[from + shift - wq + lane];
          } else {
            for (uint32_t i = lane * 4; i < len; i += 256) {
              if (i + 4 <= len) {
                st32u(s_ob + ofill + i, ld32u(in + from + i));
              } else {
                for (uint32_t k = i; k < len; k++) s_ob[ofill + k] = in[from + k];
              }
            }
          }
          ofill += len;
        } else {
          if (lane < w) gout[gpos + lane] = (uint8_t)(tag >> (8 * lane));
          gpos += w;
          for (uint32_t i = lane * 4; i < len; i += 256) {
            if (i + 4 <= len) {
              st32u(gout + gpos + i, ld32u(in + from + i));
            } else {
              for (uint32_t k = i; k < len; k++) gout[gpos + k] = in[from + k];
            }
          }
          gpos += len;
        }
        continue;
      }
      if (c_cp_len) {  // emitCopy, encoder.nim:81-125: 1 <= offset <= 65535, 4 <= length <= 65535
        const uint32_t offset = c_cp_off, length = c_cp_len;
        if (length >= 68) {  // :97-103, up to 64 elements of 64 bytes per step
          const uint32_t k64 = (length - 68) / 64 + 1;
          const uint32_t c = k64 < 64 ? k64 : 64;
          if (lane < c) {
            s_ob[ofill + 3 * lane] = (63 << 2) | 2;
            s_ob[ofill + 3 * lane + 1] = (uint8_t)offset;
            s_ob[ofill + 3 * lane + 2] = (uint8_t)(offset >> 8);
          }
          ofill += 3 * c;
          c_cp_len = length - 64 * c;
          continue;
        }
        c_cp_len = 0;
        const bool two = length > 64;                      // :105-112
        const uint32_t r = two ? length - 60 : length;     // 4..64
        const bool c2 = r >= 12 || offset >= 2048;         // :114-125
        const uint32_t lo8 = offset & 255, hi8 = offset >> 8;
        const uint32_t last = c2 ? ((((r - 1) << 2) | 2) | (lo8 << 8) | (hi8 << 16))
                                 : (((hi8 << 5) | ((r - 4) << 2) | 1) | (lo8 << 8));
        const uint32_t firstw = ((59u << 2) | 2) | (lo8 << 8) | (hi8 << 16);
        const unsigned long long bytes = two ? ((unsigned long long)firstw | ((unsigned long long)last << 24))
                                             : (unsigned long long)last;
        const uint32_t total = (two ? 3 : 0) + (c2 ? 3 : 2);
        if (lane < total) s_ob[ofill + lane] = (uint8_t)(bytes >> (8 * lane));
        ofill += total;
        continue;
      }
      if (c_lit2_len) {
        c_lit_from = c_lit2_from;
        c_lit_len = c_lit2_len;
        c_lit2_len = 0;
        continue;
      }
      if (finished && ofill) {
        want_flush = true;
        continue;
      }
      break;
    }
    if (lane < 8) s_cold[lane] = 0;  // (all used up)
    wave_fence();
    cold_any = false;
  };

  // per-block setup (encoder.nim:227-245)
  uint32_t mask = 0;
  const uint32_t ip_limit = n >= kInputMargin ? n - kInputMargin : 0;
  if (n < kMinNonLiteral) {  // encoder.nim:227-229
    finished = true;
    lit2_from = 0;
    lit2_len = n;
    cold_any = true;
  } else {
    uint32_t table_size = 1u << 8;
    while (table_size < kMaxTableSize && table_size < n) table_size <<= 1;
    mask = table_size - 1;
    for (uint32_t i = lane * 8; i < table_size; i += 64 * 8) {
      if constexpr (GT) *reinterpret_cast<uint4*>(&gtab[i]) = make_uint4(0, 0, 0, 0);
      else *reinterpret_cast<enc_lds128*>(&s_table[i]) = enc_v4{0, 0, 0, 0};
    }
    wave_fence();
    fill_window(0);
  }

  // Two kinds of round.
  //
  // FRESH round (idx0 == 0): lane L <-> position base + L, 64 consecutive positions.  Right after
  // a copy that ended at ip (has0) base = ip - 1: lane 0 is the table insert of ip - 1
  // (encoder.nim:371, a write that is never a candidate check), lane 1 the copy-loop probe at ip
  // (:373-380), lanes 2.. the scan that starts at ip + 1, whose probe offsets from its start are
  // 0..31, then 32, 34, .. 62 (skip = 32, step = skip >> 5, :311-331 -- the sixteen unrolled
  // probes :280-309 are its first sixteen).  Every lane hashes ITS position and takes as candidate
  // the nearest earlier lane with the same slot, else the table value: right if every earlier
  // lane of the round was inserted.  The chain then walks the round the way the sequential loop
  // does and, at a match, carries on at the copy's end INSIDE the round (insert, copy-loop probe,
  // next scan ...).  S collects the lanes the sequential loop really touches; a probe whose
  // nearest earlier same-slot lane is not in S has seen a candidate the sequential loop would not
  // have seen, and the round stops in front of it.  At the end the table is left as S alone
  // leaves it.  A block of text takes about 1 050 such rounds instead of about 10 000 one-match
  // rounds (tools/encode_model.py is the same procedure on the CPU, checked against the oracle).
  //
  // CONTINUING round (idx0 > 0): a scan that found nothing among its first 47 probes; lanes take
  // the next 64 entries of the probe sequence (positions further and further apart), straight
  // from memory; one match ends the round.
  uint32_t has0 = 0;        // fresh round after a copy: lanes 0,1 are the insert of ip-1 and the probe at ip = s0-1
  uint32_t next_emit = 0;   // start of the pending literal
  uint32_t s0 = 1;          // position of probe 0 of the current literal scan
  uint32_t idx0 = 0;        // index into the probe sequence of a continuing round's first lane

  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;  // DEBUG section timers
  uint32_t rounds = 0, c_fast = 0, c_bail_ms = 0, c_bail_order = 0, c_cont = 0, c_fresh_nohas = 0;  // DEBUG
  uint32_t enc_inject = 0;  // (ENC_INJECT_ORDER_FAULT)
  (void)enc_inject;
  auto tick = [&](int k) {
    if (SNAPPY_STATS(prm)) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tacc[k] += t - tprev;
      tprev = t;
    }
  };
  if (SNAPPY_STATS(prm)) tprev = __builtin_amdgcn_s_memtime();
  // findMatchLength beyond what the registers hold, encoder.nim:130-182: exact, bounded by n;
  // 256 bytes per step across the wave.  a < b are the positions still to compare.
  auto extend_match = [&](uint32_t a, uint32_t b) -> uint32_t {
    uint32_t more = 0;
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
        return more + 4 * f + readlane(e4, f);
      }
      more += 256;
      a += 256;
      b += 256;
    }
  };
  constexpr uint64_t kScanPat = 0x55555555FFFFFFFFull;  // offsets of a scan's first 47 probes: 0..31, 32, 34, .. 62
  // the lanes a copy that ends at THIS lane looks at next: its copy-loop probe, then the scan's probes
  const uint64_t mine = ((kScanPat << 1) | 1ull) << lane;
  while (!finished) {
#ifndef ENC_NO_FAST_LOOP
    // ==== the common round in a loop of its own =====================================================================
    // A fresh round right after a copy, away from ipLimit (every lane has a position, every probe of the pattern
    // runs), whose chain finds a match from its first copy end: 1 057 of a text block's 1 062 rounds.  Same steps as the
    // general round below -- which takes over, from the same state, whenever this one declines (nothing found from the
    // first end, a wrong candidate in the first segment, the table's order of service not ascending) -- but with
    // nothing of the general round's state alive in it.
    while (has0 && idx0 == 0 && s0 - 2 + 96 <= ip_limit) {
      const uint32_t base = s0 - 2;
      if (!in_window(base, 96)) fill_window(base);
      rounds++;
      const uint32_t p = base + lane;
      uint32_t d, pd1, pd2, pd3;
      {
        const uint32_t qa = p + shift - wq;
        const enc_lds32* w32 = reinterpret_cast<const enc_lds32*>(s_win + (qa & ~3u));
        const uint32_t r0 = w32[0], r1 = w32[1], r2 = w32[2], r3 = w32[3], r4 = w32[4];
        const uint32_t sh8 = (qa & 3) * 8;
        d = __funnelshift_r(r0, r1, sh8);
        pd1 = __funnelshift_r(r1, r2, sh8);
        pd2 = __funnelshift_r(r2, r3, sh8);
        pd3 = __funnelshift_r(r3, r4, sh8);
      }
      tick(0);
      const uint32_t h = snappy_hash(d, mask);
      const uint32_t sh16 = (h & 1) * 16;
      const uint32_t taddr = GT ? 0u : (uint32_t)(uintptr_t)&s_table[h & ~1u];
      uint32_t old, cand, dep = 64;
      bool inround = false;
      // (GT) my slot of the scratch, and what I put there: 0x8000 | the hash bits the slot number leaves out | my lane
      const uint32_t xsh = (h & 1) * 16;
      const uint32_t xaddr = GT ? (uint32_t)(uintptr_t)&s_gx[h & (kEncScratch - 2)] : 0u;
      const uint32_t xval = 0x8000u | ((h >> kEncScratchBits) << 6) | lane;
      if constexpr (!GT) {  // the table in one trip (see the general round)
        uint32_t ret;
        wave_fence();
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ret)
                     : "v"(taddr), "v"(0xffffu << sh16), "v"((p & 0xffffu) << sh16)
                     : "memory");
        old = (ret >> sh16) & 0xffffu;
        inround = old >= base;
        cand = old;
        dep = inround ? old - base : 64;
      } else {
        // The table elsewhere.  ONE scattered load reads the slots as they are; nothing is written before the round knows
        // which lanes stay inserted (one scattered store at its end) -- a CU's vector memory pipeline takes ~140 cycles for
        // a load and ~220 for a store of 64 scattered lanes (tools/probes/gtab_probe.hip), and with eight waves a CU that
        // pipeline is what the rounds wait for.  The lanes of the round that share a slot find each other in LDS instead:
        // a scratch of kEncScratch 16-bit entries indexed by the hash's low bits, through which every lane exchanges
        // (ds_mskor_rtn_b32, lanes on one address served in ascending order -- checked, as for the LDS table) its lane
        // and the remaining hash bits for what the lane before it on that entry put there: the same bits = the nearest
        // earlier lane on my table slot; other bits = a lane of another slot in between (two per round): those few lanes
        // are looked up with ballots.  Every lane clears its entry again.
        old = T_ld(h);
        uint32_t ret;
        wave_fence();
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\tds_write_b16 %4, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ret)
                     : "v"(xaddr), "v"(0xffffu << xsh), "v"(xval << xsh), "v"(xaddr + (h & 1) * 2), "v"(0u)
                     : "memory");
        const uint32_t prev = (ret >> xsh) & 0xffffu;
        const bool has = (prev & 0x8000u) != 0;
        const bool same = has && ((prev ^ xval) & 0x7fc0u) == 0;
        if (__builtin_expect(ballot(has && (prev & 63u) >= lane) != 0 || ENC_ORDER_FAULT(), 0)) {  // not served in ascending order
          c_bail_order++;
          break;  // (nothing has been written to the table)
        }
        dep = same ? (prev & 63u) : 64u;
        uint64_t amb = ballot(has && !same);
        while (amb) {  // (a lane of another slot sits between me and a possible earlier lane of mine)
          const uint32_t j = ctz64(amb);
          const uint32_t hj = readlane(h, j);
          const uint64_t g = ballot(h == hj);
          const uint64_t below = g & ((1ull << lane) - 1);
          if (h == hj && below) dep = 63 - (uint32_t)__builtin_clzll(below);
          amb &= ~g;
        }
        inround = dep < 64;
        cand = inround ? base + dep : old;
      }
      const uint64_t conf = ballot(inround);  // lanes whose candidate is another lane of the round
      // puts the slots back to what the first-served lanes saw (the round is then done by the general form)
      auto undo_table = [&]() {
        if constexpr (!GT) {
          wave_fence();
          asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(taddr), "v"((inround ? 0u : 0xffffu) << sh16), "v"((inround ? 0u : old) << sh16)
                       : "memory");
          wave_fence();
        }  // (GT: nothing has been written)
      };
      if constexpr (!GT) {
        if (__builtin_expect(ballot(inround && old - base >= lane) != 0 || ENC_ORDER_FAULT(), 0)) {  // not served in ascending order
          undo_table();
          c_bail_order++;
          break;
        }
      }
      uint4 cv;
      // (candidates that lie inside the LDS window read from there instead -- fewer lanes in the gather -- measured 16 % slower:
      // two EXEC-masked paths and five more LDS reads a lane cost more than the lanes saved; profiles/README.md)
      __builtin_memcpy(&cv, in + cand, 16);
      tick(1);
      drain();  // the previous round's elements, while the candidates are in flight
      tick(2);
      const uint64_t m4 = ballot(cv.x == d);
      uint32_t eq;
      {
        const uint32_t x1 = pd1 ^ cv.y, x2 = pd2 ^ cv.z, x3 = pd3 ^ cv.w;
        const uint32_t e3 = x3 ? 12 + ((uint32_t)__builtin_ctz(x3) >> 3) : 16;
        const uint32_t e2 = x2 ? 8 + ((uint32_t)__builtin_ctz(x2) >> 3) : e3;
        eq = x1 ? 4 + ((uint32_t)__builtin_ctz(x1) >> 3) : e2;
      }
      tick(3);
      // ---- the chain (see the general round) ----
      uint64_t MS = 0, E = 0;
      uint32_t lens = eq, e = 1;
      {
        const uint64_t cnd = mine & m4;
        const uint32_t mv = cnd ? ctz64(cnd) : 64;
        const uint32_t lm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(mv << 2), (int)(eq | (eq == 16 ? 0x100u : 0u)));
        const uint32_t nxt = mv == 64 ? 255u : ((lm & 0x100u) ? 254u : mv + lm);
#ifndef ENC_NO_CMPST_WALK
        // No match of the round fills its 16 bytes (92 % of text's rounds): nothing on the chain needs a look at
        // memory, and ONE LDS instruction walks it -- every lane offers "if the chain stands at me, it goes on to
        // where my copy ends"; the lanes of one address are served in ascending lane order and the pointers lead
        // forward, so the word runs through the whole chain while the instruction executes, and a lane that gets
        // back its own number was on it.  (Lane 63 never offers: a copy that ends there ends the round.)
        // The order of service is not documented: out of order the word stops early, at a lane inside the round
        // (it only ever moves along the chain), which is seen at once -- the loop below then does the round.
        const bool walk_lds = (m4 & ballot(eq == 16)) == 0;
        bool walked = false;
        if (walk_lds) {
          const uint32_t wa_ = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_walk;
          uint32_t ret, fin;
          wave_fence();
          // (the word is 1 -- every chain starts at lane 1 -- when a round begins: put back right behind the look at it)
          asm volatile("ds_cmpst_rtn_b32 %0, %2, %3, %4\n\tds_read_b32 %1, %2\n\tds_write_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(ret), "=&v"(fin)
                       : "v"(wa_), "v"(lane == 63 ? 0xffffffffu : lane), "v"(nxt), "v"(1u)
                       : "memory");
          const uint64_t ON = ballot((ret == lane) & (lane != 63));  // the copy ends the chain went on from (lane 1 among them)
          const uint32_t f = readfirst(fin);
          if (__builtin_expect(f >= 63 && !ENC_ORDER_FAULT(), 1)) {
            walked = true;
            const uint32_t last = 63 - (uint32_t)__builtin_clzll(ON);
            if (f == 255) {  // nothing found from the last of them: the round ends there
              E = ON & ~(1ull << last);
              e = last;
            } else {         // a copy that ends at lane 63 or behind the round
              E = ON;
              e = f;
            }
            // the matches taken: a copy starts at me iff the match found from the nearest end at or below me is me
            const uint64_t eb = E & ((2ull << lane) - 1);
            const uint32_t ce = 63 - (uint32_t)__builtin_clzll(eb | 1ull);
            const uint32_t mvc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ce << 2), (int)mv);
            MS = ballot((eb != 0) & (mvc == lane));
            tick(4);
          }
        }
        if (!walked) {
#endif
        const uint32_t pk = nxt | (mv << 8);
        const uint32_t pk2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nxt & 63) << 2), (int)pk);
        const uint32_t P = pk | (pk2 << 16);
        tick(4);
        uint32_t x = readlane(P, e);
        uint32_t t, m;
        for (;;) {
          {
            asm volatile(
                "1:\n"
                "s_and_b32 %[t], %[x], 0xff\n"
                "s_bfe_u32 %[m], %[x], 0x80008\n"
                "s_cmp_lt_u32 %[t], 63\n"
                "s_cbranch_scc0 2f\n"
                "s_bitset1_b64 %[E], %[e]\n"
                "s_bitset1_b64 %[MS], %[m]\n"
                "s_mov_b32 %[e], %[t]\n"
                "s_bfe_u32 %[t], %[x], 0x80010\n"
                "s_lshr_b32 %[m], %[x], 24\n"
                "s_cmp_lt_u32 %[t], 63\n"
                "s_cbranch_scc0 2f\n"
                "s_bitset1_b64 %[E], %[e]\n"
                "s_bitset1_b64 %[MS], %[m]\n"
                "s_mov_b32 %[e], %[t]\n"
                "v_readlane_b32 %[x], %[P], %[e]\n"
                "s_branch 1b\n"
                "2:\n"
                : [E] "+s"(E), [MS] "+s"(MS), [e] "+s"(e), [t] "=&s"(t), [m] "=&s"(m), [x] "+s"(x)
                : [P] "v"(P)
                : "scc");
          }
          if (t == 255) break;
          E |= 1ull << e;
          MS |= 1ull << m;
          if (t == 254) {
            const uint32_t matched = 16 + extend_match(readlane(cand, m) + 16, base + m + 16);
            lens = lane == m ? matched : lens;
            e = m + matched;
          } else {
            e = t;
          }
          if (e > 62) break;
          x = readlane(P, e);
        }
#ifndef ENC_NO_CMPST_WALK
        }
#endif
        tick(5);
      }
      bool covered = false, in_s = false;
      if (MS) {
        const uint32_t endv = ((MS >> lane) & 1) ? lane + lens : 0;
        uint32_t unused;
        uint32_t ce = wave_excl_scan_max(endv, lane, &unused);
        ce = ce > 1 ? ce : 1;
        covered = lane < (ce > endv ? ce : endv);
        const uint32_t mlast = 63 - (uint32_t)__builtin_clzll(MS);
        const uint32_t o = lane - ce - 1;
        const bool on_pat = (lane == ce) | (o < 32) | ((o & 1) == 0);
        const bool last_byte = lane + 1 == ce;
        in_s = (lane <= mlast) & ((lane >= ce) ? on_pat : last_byte);
        if (conf) {
          const uint64_t S = ballot(in_s);
          const uint64_t bad = ballot(dep < 64 && !((S >> (dep & 63)) & 1)) & S & ~(E >> 1);
          if (bad) {
            const uint32_t fb = ctz64(bad);
            const uint64_t eb = E & ((2ull << fb) - 1);
            e = 63 - (uint32_t)__builtin_clzll(eb);
            MS &= (1ull << e) - 1;
            covered = covered && lane < e;
            in_s = in_s && lane + 1 < e;
          }
        }
      }
      tick(6);
      if (MS == 0) {  // nothing to keep of this round: the general round does it from the same state
        undo_table();
        c_bail_ms++;
        break;
      }
      c_fast++;
      if constexpr (!GT) {  // the table as the inserted lanes leave it (see the general round)
        const bool firstm = old < base;
        const uint32_t mk = (firstm || in_s) ? 0xffffu : 0u;
        const uint32_t dv = in_s ? (p & 0xffffu) : (firstm ? old : 0u);
        wave_fence();
        asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(taddr), "v"(mk << sh16), "v"(dv << sh16) : "memory");
        wave_fence();
      } else {
        // the table elsewhere: of the inserted lanes on one slot the last one writes its position.  The inserted lanes go
        // through the scratch once more; who is on an entry last is that lane -- unless the entry is shared with another
        // slot's lane or the order of service was not ascending: those lanes are settled with ballots.
        uint32_t fin = xval;
        if (in_s) {
          asm volatile("ds_mskor_b32 %1, %2, %3\n\tds_read_u16 %0, %4\n\tds_write_b16 %4, %5\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(fin)
                       : "v"(xaddr), "v"(0xffffu << xsh), "v"(xval << xsh), "v"(xaddr + (h & 1) * 2), "v"(0u)
                       : "memory");
        }
        bool writer = in_s && fin == xval;
        // (someone else is last on my entry: a later lane of my slot -- fine, it writes --, or not: look)
        const bool inject = ENC_ORDER_FAULT();  // (tests: every inserted lane is looked up)
        uint64_t amb = ballot(in_s && ((fin != xval && (((fin ^ xval) & 0x7fc0u) != 0 || (fin & 63u) < lane)) || inject));
        while (amb) {
          const uint32_t j = ctz64(amb);
          const uint32_t hj = readlane(h, j);
          const uint64_t g = ballot(in_s && h == hj);
          if (in_s && h == hj) writer = lane == 63 - (uint32_t)__builtin_clzll(g);
          amb &= ~g;
        }
        if (writer) gtab[h] = (uint16_t)p;
      }
      {  // hand the elements over
        const uint32_t mlast = 63 - (uint32_t)__builtin_clzll(MS);
        const uint32_t llast = readlane(lens, mlast);
        dpend = true;
        dp_ms = MS;
        dp_lit = !covered && lane >= 1u && lane < e;
        dp_len = lens;
        dp_off = p - cand;
        dp_byte = d & 0xff;
        if (llast > 64) {
          // A last copy of more than 64 bytes is several elements (emitCopy, encoder.nim:97-125: 64-byte copies while
          // 68 or more are left, one of 60 if more than 64 are left then, the rest).  The lanes behind mlast lie
          // inside this copy and have nothing to emit: lane mlast + j takes element j, as if a copy of that length
          // started there -- the round's one prefix sum places them like any other element.  (More elements than
          // lanes left: the general way.)
          const uint32_t off_l = base + mlast - readlane(cand, mlast);
          const uint32_t k64 = llast >= 68 ? (llast - 68) / 64 + 1 : 0;
          const uint32_t rem = llast - 64 * k64;                 // 4..67
          const uint32_t has60 = rem > 64 ? 1u : 0u;
          const uint32_t nel = k64 + has60 + 1;
#ifndef ENC_NO_LONG_SPLIT
          if (mlast + nel <= 64) {
            const uint32_t j = lane - mlast;                       // (lanes below mlast: huge)
            const uint32_t lj = j < k64 ? 64u : (((j == k64) & (has60 != 0)) ? 60u : rem - 60 * has60);
            const bool mine_el = j < nel;
            dp_len = mine_el ? lj : dp_len;
            dp_off = mine_el ? off_l : dp_off;
            dp_ms |= (nel >= 64 ? ~0ull : ((1ull << nel) - 1)) << mlast;
          } else
#endif
          {
            dp_ms &= ~(1ull << mlast);
            cp_off = off_l;
            cp_len = llast;
            cold_any = true;
          }
        }
      }
      tick(7);
      next_emit = base + e;
      s0 = base + e + 1;
      if (base + e > ip_limit) {  // encoder.nim:362 -- strictly greater (only after a long copy)
        finished = true;
        lit2_from = base + e;
        lit2_len = n - (base + e);
        cold_any = true;
        break;
      }
    }
    if (finished) break;
#endif
    // ---- this round's position per lane, and 16 bytes of input there ---------------------------
    const bool fresh = idx0 == 0;
    if (!fresh) c_cont++;
    if (fresh && !has0) c_fresh_nohas++;
    uint32_t p = 0, d = 0, dep = 64, base = 0;
    uint32_t pd1 = 0, pd2 = 0, pd3 = 0;
    bool valid = false;
    const uint32_t tsink = kMaxTableSize + lane;
    rounds++;
    {
      if (fresh) {
        // everything a fresh round touches is in the window
        base = has0 ? s0 - 2 : 0;  // (the block's first round: lane 0 = position 0 is never probed)
        if (!in_window(base, 96)) fill_window(base);
        p = base + lane;
        valid = p <= ip_limit && (has0 || lane > 0);
        const uint32_t qa = valid ? (p + shift - wq) : 0;
        const enc_lds32* w32 = reinterpret_cast<const enc_lds32*>(s_win + (qa & ~3u));
        const uint32_t r0 = w32[0], r1 = w32[1], r2 = w32[2], r3 = w32[3], r4 = w32[4];
        const uint32_t sh8 = (qa & 3) * 8;
        d = __funnelshift_r(r0, r1, sh8);
        pd1 = __funnelshift_r(r1, r2, sh8);
        pd2 = __funnelshift_r(r2, r3, sh8);
        pd3 = __funnelshift_r(r3, r4, sh8);
      } else {
        // a long scan, far ahead of the window: straight from memory
        const uint32_t si = idx0 + lane;
        p = s0;
        if (si < kSeqLen) {  // (the probe sequence, from global memory: these rounds are rare and far apart)
          const uint32_t so = prm.seq_off[si], st = prm.seq_step[si];
          p = s0 + (so < 65535 ? so : 65535);  // saturated (such a probe is never valid)
          valid = p + (st < 65535 ? st : 65535) <= ip_limit;  // encoder.nim:318-321
        }
        if (valid) d = ld32u(in + p);
      }
    }
    const uint64_t vmask = ballot(valid);
    tick(0);  // positions + input bytes
    if (vmask == 0) {  // (only a continuing round: the scan has reached ipLimit)
      finished = true;
      lit2_from = next_emit;
      lit2_len = n - next_emit;
      cold_any = true;
      break;
    }
    const uint32_t h = snappy_hash(d, mask);
    // (a lane without a position works on its private sink slot instead of being branched around)
    const uint32_t ti = (GT || valid) ? h : tsink;  // (GT: a lane without a position reads slot h(0) and stores nothing)
    uint4 cv;
    uint64_t grp = 1ull << lane;  // lanes of this round that share my slot
    bool any_conflict = false;
    uint32_t old = 0, cand = 0;
    // ---- the table in ONE LDS trip (fresh rounds after a copy, away from ipLimit: every lane has a position) ----
    // ds_mskor_rtn_b32 replaces my 16-bit slot inside its dword and returns the dword as it was.  Lanes on one
    // address are served one after the other; served in ASCENDING lane order, a lane gets back the position of the
    // nearest earlier lane of the round on its slot, else what the table held -- the candidate of the sequential
    // loop with every earlier lane inserted, without the write / read back / write again of the plain form.  The
    // order is not documented, so it is checked every time: a position of this round that a lane gets back must
    // be an EARLIER lane's (any other order of service shows some lane a later one's); if not, the slots go back
    // to what the first-served lanes saw and the plain form below does the round.
#ifndef ENC_NO_ATOMIC_TABLE
    bool atab = !GT && fresh && has0 && base + 96 <= ip_limit;
#else
    bool atab = false;
#endif
    const uint32_t sh16 = (h & 1) * 16;
    const uint32_t taddr = GT ? 0u : (uint32_t)(uintptr_t)&s_table[h & ~1u];
    if (!GT && atab) {
      uint32_t ret;
      wave_fence();
      asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(ret)
                   : "v"(taddr), "v"(0xffffu << sh16), "v"((p & 0xffffu) << sh16)
                   : "memory");
      old = (ret >> sh16) & 0xffffu;
      const bool inround = old >= base;  // (nothing at or behind base has been inserted before this round)
      any_conflict = ballot(inround) != 0;
      if (__builtin_expect(ballot(inround && old - base >= lane) != 0 || ENC_ORDER_FAULT(), 0)) {
        wave_fence();
        s_table[inround ? tsink : h] = (uint16_t)old;  // the first-served lane of every slot puts back what it saw
        wave_fence();
        atab = false;
      } else {
        cand = old;
        dep = inround ? old - base : 64;
      }
    }
    if (!atab) {
      old = T_ld(ti);
      cand = old;
      {
        wave_fence();
        T_st(valid, h, p);
        wave_fence();
        const uint32_t chk = T_ld(ti);
        const bool lost = valid && chk != (p & 0xffffu);  // another lane of the round has my slot, and wrote last
        uint64_t losers = ballot(lost);

        // candidate as the sequential loop would see it if every earlier lane was inserted: the
        // position of the nearest earlier lane of this round with my slot, else what the table held
        any_conflict = losers != 0;
        if (any_conflict && fresh) {
          // Nearly every shared slot is shared by two lanes: the one that lost writes once more, and
          // now each of the two reads the other's position (its lane: positions are consecutive).
          // A lane that loses again is one of three or more on a slot; those go through the loop.
          wave_fence();
          T_st(lost, h, p);
          wave_fence();
          const uint32_t chk2 = T_ld(ti);
          const bool lost2 = lost && chk2 != (p & 0xffffu);
          const uint32_t partner = ((lost ? chk : chk2) - base) & 0xffffu;
          if (valid && (lost || chk2 != (p & 0xffffu))) {
            grp |= 1ull << (partner & 63);
            if (partner < lane) {
              cand = base + partner;
              dep = partner;
            }
          }
          losers = ballot(lost2);
        }
        while (losers) {  // one pass per colliding slot
          const uint32_t j = ctz64(losers);
          const uint32_t hj = readlane(h, j);
          const uint64_t g = ballot(valid && h == hj);
          const bool in_g = valid && h == hj;
          const uint64_t below = g & ((1ull << lane) - 1);
          const uint32_t pred = below ? 63 - (uint32_t)__builtin_clzll(below) : lane;
          const uint32_t pp = fresh ? base + pred : __shfl(p, pred, 64);
          if (in_g) {
            grp = g;
            if (below) {
              cand = pp;
              dep = pred;  // the nearest earlier lane of my slot
            }
          }
          losers &= ~g;
        }
      }
    }
    // The candidate's 16 bytes (cand < p, so cand + 16 <= n): the 4-byte check of encoder.nim:326 and, in fresh
    // rounds, the first 16 bytes of findMatchLength.  This is the round's one trip to L2/HBM: it leaves as soon as
    // the table has answered, and the previous round's output is written while it is under way.
    __builtin_memcpy(&cv, in + (valid ? cand : 0), 16);
    tick(1);  // table, conflicts
    drain();  // the previous round's elements, while the candidates are in flight
    tick(2);  // drain

    const uint64_t m4 = ballot(valid && cv.x == d);
    uint32_t eq;  // equal leading bytes, 4..16 (meaningful where the 4-byte check passed)
    {
      const uint32_t x1 = pd1 ^ cv.y, x2 = pd2 ^ cv.z, x3 = pd3 ^ cv.w;
      const uint32_t e3 = x3 ? 12 + ((uint32_t)__builtin_ctz(x3) >> 3) : 16;
      const uint32_t e2 = x2 ? 8 + ((uint32_t)__builtin_ctz(x2) >> 3) : e3;
      eq = x1 ? 4 + ((uint32_t)__builtin_ctz(x1) >> 3) : e2;
      // the copy-loop probe may sit at ip = n - 15: findMatchLength stops at the block's end
      eq = eq < n - p ? eq : n - p;
    }
    // (Measuring matches that fill the 16 bytes on by their own lanes -- 64 more bytes, every such lane of the
    // round at once -- was built and measured: the second trip to the L2 costs a round more than the chain's
    // one-by-one extensions cost it: text +6 %, html +9 % time.  profiles/README.md)
    constexpr uint32_t eqcap = 16;  // what a lane has looked at: a match of that length may be longer
    tick(3);  // wait for the candidates + compare

    if (fresh) {
      uint64_t conf = any_conflict ? ballot(dep < 64) : 0;  // lanes whose candidate is another lane of the round
      uint64_t S = 0, MS = 0, COVER = 0;
      bool covered = false;    // per lane: inside a copy of this round
      bool in_s = false;       // per lane: touched by the sequential loop (its table write stays)
      uint32_t lens = eq;      // per lane: length of the copy that starts here
      uint32_t e = 1;          // lane of the current copy's end (= ip - base)
      bool ended = false;      // the block ends inside this round
      uint32_t tail_from = 0;
      // state for the next round, set where the chain stops
      bool n_has0 = true;
      uint32_t n_s0 = 0, n_idx0 = 0, n_emit = 0;
      if (has0 && base + 96 <= ip_limit) {
        // ---- the chain, common case: no lane of the round is near ipLimit ----------------------
        // (all 64 lanes are valid, every probe of the pattern runs, a copy that ends inside the
        // round ends at most at ip_limit.)  Every lane works out where the sequential loop would go
        // if a copy ENDED at it: mv = the first lane that matches among its copy-loop probe and the
        // probes of the scan behind it, nxt = where that match's copy ends.  The chain itself is
        // then two register reads per copy.
        const uint64_t cnd = mine & m4;
        const uint32_t mv = cnd ? ctz64(cnd) : 64;
        const uint32_t lm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(mv << 2), (int)(eq | (eq == eqcap ? 0x100u : 0u)));
        // 255: nothing found; 254: the match at mv may be longer than what its lane has looked at; else the lane
        // where the copy ends (< 63 inside the round, up to 63 + 79 behind it)
        const uint32_t nxt = mv == 64 ? 255u : ((lm & 0x100u) ? 254u : mv + lm);
        const uint32_t pk = nxt | (mv << 8);
        // ... and two copies per trip of the loop: next to a lane's own hop the hop of the lane it leads to
        // (one more ds_bpermute here; the loop then pays its register read across lanes and its taken branch once
        // per two copies)
        const uint32_t pk2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nxt & 63) << 2), (int)pk);
        const uint32_t P = pk | (pk2 << 16);
        tick(4);  // chain: per-lane preparation
        uint64_t E = 0;  // ends of copies from which the chain went on
        uint32_t x = readlane(P, e);
        uint32_t t, m;
        for (;;) {
          // Hops that stay inside the round (the copy found from e ends at t < 63): note e and the match, go to
          // t.  Leaves with (e, t, m) = the first hop that does not: nothing found (255), a match that may be
          // longer than its lane has looked (254), a copy that ends behind the round (63 ..).  (A taken branch costs a
          // lone wave about 25 cycles, a register read across lanes as much, a scalar instruction 5.)
          {
            asm volatile(
                "1:\n"
                "s_and_b32 %[t], %[x], 0xff\n"
                "s_bfe_u32 %[m], %[x], 0x80008\n"
                "s_cmp_lt_u32 %[t], 63\n"
                "s_cbranch_scc0 2f\n"
                "s_bitset1_b64 %[E], %[e]\n"
                "s_bitset1_b64 %[MS], %[m]\n"
                "s_mov_b32 %[e], %[t]\n"
                "s_bfe_u32 %[t], %[x], 0x80010\n"
                "s_lshr_b32 %[m], %[x], 24\n"
                "s_cmp_lt_u32 %[t], 63\n"
                "s_cbranch_scc0 2f\n"
                "s_bitset1_b64 %[E], %[e]\n"
                "s_bitset1_b64 %[MS], %[m]\n"
                "s_mov_b32 %[e], %[t]\n"
                "v_readlane_b32 %[x], %[P], %[e]\n"
                "s_branch 1b\n"
                "2:\n"
                : [E] "+s"(E), [MS] "+s"(MS), [e] "+s"(e), [t] "=&s"(t), [m] "=&s"(m), [x] "+s"(x)
                : [P] "v"(P)
                : "scc");
          }
          if (t == 255) break;  // nothing found from e: the round ends there
          E |= 1ull << e;
          MS |= 1ull << m;
          if (t == 254) {       // found, and it may go on behind what its lane has looked at
            const uint32_t seen = readlane(eq, m);
            const uint32_t matched = seen + extend_match(readlane(cand, m) + seen, base + m + seen);
            lens = lane == m ? matched : lens;
            e = m + matched;
          } else {              // found, and it ends behind the round
            e = t;
          }
          if (e > 62) break;
          x = readlane(P, e);
        }
        tick(5);  // chain: hops
        if (MS) {
          // What the sequential loop inserted, from the copies' ends: ce = the end of the last copy
          // that starts in front of a lane (1 for the copy this round started behind).  A lane
          // behind that end is the copy-loop probe (== ce) or a scan probe by the pattern; a lane
          // in front of it lies inside the copy, and only its last byte is inserted (ip - 1,
          // encoder.nim:371).  Nothing behind the last match was touched.
          const uint32_t endv = ((MS >> lane) & 1) ? lane + lens : 0;
          uint32_t unused;
          uint32_t ce = wave_excl_scan_max(endv, lane, &unused);
          ce = ce > 1 ? ce : 1;
          covered = lane < (ce > endv ? ce : endv);
          const uint32_t mlast = 63 - (uint32_t)__builtin_clzll(MS);
          const uint32_t o = lane - ce - 1;  // offset in the scan behind ce (for lanes behind it)
          // (written without short circuits: the compiler turns && / || / ?: on per-lane conditions into branches
          // over the EXEC mask, and this is on the round's critical path: -2.5 % kernel time)
          const bool on_pat = (lane == ce) | (o < 32) | ((o & 1) == 0);
          const bool last_byte = lane + 1 == ce;
          in_s = (lane <= mlast) & ((lane >= ce) ? on_pat : last_byte);
          if (conf) {
            // a probe whose nearest earlier same-slot lane was not inserted saw a wrong candidate:
            // everything from the copy end in front of the first such probe is undone
            S = ballot(in_s);
            const uint64_t bad = ballot(dep < 64 && !((S >> (dep & 63)) & 1)) & S & ~(E >> 1);
            if (bad) {
              const uint32_t fb = ctz64(bad);
              const uint64_t eb = E & ((2ull << fb) - 1);
              e = 63 - (uint32_t)__builtin_clzll(eb);  // (lane 1 is in E: eb != 0)
              MS &= (1ull << e) - 1;
              covered = covered && lane < e;
              in_s = in_s && lane + 1 < e;
            }
          }
          if (MS) {
            n_has0 = true, n_s0 = base + e + 1, n_idx0 = 0, n_emit = base + e;
            if (base + e > ip_limit) {  // encoder.nim:362 -- strictly greater (only after a long copy)
              ended = true;
              tail_from = base + e;
            }
          }
        }
      }
      if (MS == 0) {
        // ---- the chain, every case (the block's first round, rounds near ipLimit, a first
        // segment that finds nothing or runs into a wrong candidate) ---------------------------
        const uint64_t L1 = ballot(valid && p + 1 <= ip_limit);  // a step-1 probe runs here (encoder.nim:318-321)
        const uint64_t L2 = ballot(valid && p + 2 <= ip_limit);  // a step-2 probe runs here
        bool first = true;  // the round's first segment: it always makes progress
        S = 0;
        e = 1;
        for (;;) {
          const bool ops = has0 || !first;
          uint64_t opA = 0, opB = 0;
          uint32_t ls = 1;
          if (ops) {
            if (e > 62) {  // the copy ended outside the round: next round starts there
              n_has0 = true, n_s0 = base + e + 1, n_idx0 = 0, n_emit = base + e;
              break;
            }
            opA = 1ull << (e - 1);
            opB = 1ull << e;
            ls = e + 1;
          }
          const uint64_t probes = kScanPat << ls;
          const uint64_t VS = ((0xFFFFFFFFull << ls) & L1) | ((0x5555555500000000ull << ls) & L2);
          const uint64_t pmm = (opB | VS) & m4;
          uint64_t bad = 0;
          if (conf) {
            const uint64_t T = S | opA | opB | VS;
            bad = ballot(dep < 64 && !((T >> (dep & 63)) & 1)) & (opB | VS);
          }
          const uint32_t m = pmm ? ctz64(pmm) : 64;
          const uint32_t fb = bad ? ctz64(bad) : 64;
          if (fb < 64 && fb <= m) {  // a probe that saw a wrong candidate comes first
            if (first) {             // keep what lies in front of it, go on from there as a continuing scan
              const uint64_t below = (1ull << fb) - 1;
              S |= (opA | opB | VS) & below;
              n_has0 = false, n_s0 = s0, n_idx0 = (uint32_t)__builtin_popcountll(VS & below), n_emit = next_emit;
            } else {                 // undo this segment: a fresh round from the last copy's end
              n_has0 = true, n_s0 = base + e + 1, n_idx0 = 0, n_emit = base + e;
            }
            break;
          }
          if (m == 64) {  // nothing found in what is left of the round
            if (first) {
              S |= opA | opB | VS;
              if (probes & ~VS) {  // the scan reaches ipLimit: encoder.nim:319-321
                ended = true;
                tail_from = next_emit;
              } else {
                n_has0 = false, n_s0 = s0, n_idx0 = (uint32_t)__builtin_popcountll(VS), n_emit = next_emit;
              }
            } else {
              n_has0 = true, n_s0 = base + e + 1, n_idx0 = 0, n_emit = base + e;
            }
            break;
          }
          // a match at lane m: literal up to it (if any) + copy (encoder.nim:336-359)
          S |= opA | opB | (VS & ((2ull << m) - 1));
          uint32_t matched = readlane(eq, m);
          if (matched == eqcap && base + m + matched < n)
            matched += extend_match(readlane(cand, m) + matched, base + m + matched);
          MS |= 1ull << m;
          lens = lane == m ? matched : lens;
          COVER |= (matched >= 64 ? ~0ull : ((1ull << matched) - 1)) << m;
          e = m + matched;
          first = false;
          if (base + e > ip_limit) {  // encoder.nim:362 -- strictly greater
            ended = true;
            tail_from = base + e;
            break;
          }
        }
        covered = (COVER >> lane) & 1;
        in_s = (S >> lane) & 1;
      }
      tick(6);  // chain: what was inserted, what is covered (and the general chain)
      // ---- leave the table as the lanes the sequential loop touched would have left it ---------
      wave_fence();
      if (!GT && atab) {
        // (the same service in ascending lane order, checked above for these very lanes and addresses: the
        // first-served lane of a slot -- it saw what the table held -- restores that unless it stays inserted,
        // every later lane that stays inserted overwrites: the last inserted lane of a slot writes last)
        const bool firstm = old < base;
        const uint32_t mk = (firstm || in_s) ? 0xffffu : 0u;
        const uint32_t dv = in_s ? (p & 0xffffu) : (firstm ? old : 0u);  // (nothing is OR-ed in where nothing is masked out)
        asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(taddr), "v"(mk << sh16), "v"(dv << sh16) : "memory");
      } else if (!any_conflict) {  // every lane has a slot of its own: the others take their writes back
        T_st(valid && !in_s, h, old);
      } else {              // of the lanes on one slot the last one that was touched wrote last
        const uint64_t gs = grp & ballot(in_s);
        const uint32_t top = gs ? 63 - (uint32_t)__builtin_clzll(gs) : 64;
        T_st(valid && (gs == 0 || top == lane), h, gs == 0 ? old : p);
      }
      wave_fence();
      if (MS) {  // hand the elements over
        const uint32_t mlast = 63 - (uint32_t)__builtin_clzll(MS);
        const uint32_t llast = readlane(lens, mlast);
        dpend = true;
        dp_ms = MS;
        if (llast > 64) {  // a long last copy goes the general way
          dp_ms &= ~(1ull << mlast);
          cp_off = base + mlast - readlane(cand, mlast);
          cp_len = llast;
          cold_any = true;
        }
        dp_lit = !covered && lane >= (has0 ? 1u : 0u) && lane < e;
        dp_len = lens;
        dp_off = p - cand;
        dp_byte = d & 0xff;
      }
      tick(7);  // table repair, hand-over
      if (ended) {
        finished = true;
        lit2_from = tail_from;
        lit2_len = n - tail_from;  // (may be 0)
        cold_any = true;
        break;
      }
      has0 = n_has0;
      s0 = n_s0;
      idx0 = n_idx0;
      next_emit = n_emit;
      continue;
    }

    // ---- continuing round: the first match ends it ---------------------------------------------
    // A lane that shares its slot with an earlier lane of the round would need that lane's bytes,
    // which are neither fetched nor in the window: the round is cut in front of the first such lane
    // (one round in eight of random data; the rest of the scan follows in the next round).
    const uint64_t confl = any_conflict ? ballot(dep < 64) : 0;
    const uint32_t cut = confl ? ctz64(confl) : 64;           // >= 1
    const uint64_t below_cut = confl ? (1ull << cut) - 1 : ~0ull;
    const uint64_t mm = m4 & below_cut;
    const uint64_t vm = vmask & below_cut;
    const uint32_t m_eff = mm ? ctz64(mm) : 63 - (uint32_t)__builtin_clzll(vm);
    // leave the table as the sequential loop would
    if (mm || confl) {
      wave_fence();
      T_st(valid && lane > m_eff, h, old);  // never executed there
    }
    if (any_conflict) {
      wave_fence();
      // of several lanes <= m_eff on one slot the last one wrote last
      const uint64_t later = lane >= 63 ? 0 : (grp >> (lane + 1));
      const uint32_t span = m_eff > lane ? m_eff - lane : 0;  // lanes in (lane, m_eff]
      const uint64_t later_in = span >= 64 ? later : (later & ((1ull << span) - 1));
      T_st(valid && lane <= m_eff && later_in == 0, h, p);
    }
    wave_fence();
    if (!mm) {
      if (vm == below_cut) {  // every probe missed: the next entries of the sequence
        idx0 += cut;
        continue;
      }
      finished = true;  // encoder.nim:319-321
      lit2_from = next_emit;
      lit2_len = n - next_emit;
      cold_any = true;
      break;
    }
    const uint32_t pm = readlane(p, m_eff);
    const uint32_t c = readlane(cand, m_eff);
    const uint32_t matched = 4 + extend_match(c + 4, pm + 4);
    lit_from = next_emit;  // literal input[next_emit ..< pm] + copy (pm - c, matched): emitted next round
    lit_len = pm - next_emit;
    cp_off = pm - c;
    cp_len = matched;
    cold_any = true;
    const uint32_t ip = pm + matched;
    if (ip > ip_limit) {  // encoder.nim:362 -- strictly greater
      finished = true;
      lit2_from = ip;
      lit2_len = n - ip;
      cold_any = true;
      break;
    }
    // next round: insert of ip - 1 (:371), probe at ip (:373-380), scan from ip + 1
    has0 = true;
    s0 = ip + 1;
    idx0 = 0;
    next_emit = ip;
  }
  drain();  // what the last round left, the final literal, the last flush
  if (SNAPPY_STATS(prm) && lane == 0) {
    for (int k = 0; k < 8; k++) atomicAdd(&prm.stats[k], tacc[k]);
    atomicAdd(&prm.stats[8], (unsigned long long)rounds);
    atomicAdd(&prm.stats[9], (unsigned long long)c_fast);
    atomicAdd(&prm.stats[10], (unsigned long long)c_bail_ms);
    atomicAdd(&prm.stats[11], (unsigned long long)c_bail_order);
    atomicAdd(&prm.stats[12], (unsigned long long)c_cont);
    atomicAdd(&prm.stats[13], (unsigned long long)c_fresh_nohas);
  }
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
#undef s_walk
}

// Persistent workgroups of two waves: each wave takes the next block of the launch order until none is left (one counter;
// a returning atomic per block, ~1 us against the ~50-1 500 us a block takes).  Wave 0 works with the table in LDS; wave 1
// -- in g_per4 of every four workgroups of an XCD -- with its table in global memory (encode_one_block<true>): one more
// wave per such workgroup, up to eight waves a CU, whose 32 KiB tables (g_per4 MiB per XCD) compete for the XCD's 4 MiB L2
// with the input; what misses it is served by the Infinity Cache.  g_per4 is the host's choice (snappy_hip.hip: 4).
__global__ __launch_bounds__(64 * kEncWaves) void encode_blocks_kernel(EncodeParams prm) {
  __shared__ __attribute__((aligned(16))) uint16_t s_table[kMaxTableSize + 64];  // + one sink slot per lane
  __shared__ __attribute__((aligned(16))) uint8_t s_ob[kEncWaves][kObCap];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kEncWaves][kWinSize + 32];
  __shared__ uint32_t s_cold[kEncWaves][8];  // the rarely pending output items
  __shared__ uint32_t s_walk[kEncWaves];     // the lane the chain stands at (walked by one ds_cmpst)
  __shared__ __attribute__((aligned(16))) uint16_t s_gx[kEncWaves > 1 ? kEncScratch : 2];  // the second wave's scratch: all zero between its uses
  const uint32_t wave = readfirst(threadIdx.x >> 6);
  // (blocks b and b + 8 share an XCD: the second waves are spread evenly over the XCDs)
  if (wave == 1 && (prm.gtables == nullptr || ((blockIdx.x >> 3) & 3) >= (prm.g_per4 & 7))) return;
  if (SNAPPY_DBG(prm) && wave == 0 && (prm.g_per4 & 0x100)) return;  // DEBUG: the second waves alone
  uint16_t* const gtab = wave == 1 ? prm.gtables + (size_t)blockIdx.x * kMaxTableSize : nullptr;
  if (kEncWaves > 1 && wave == 1) {
    for (uint32_t i = lane_id(); i < kEncScratch / 2; i += 64) reinterpret_cast<uint32_t*>(s_gx)[i] = 0;
    wave_fence();
  }
  const unsigned long long queue_s = ((unsigned long long)readfirst((uint32_t)((uintptr_t)prm.queue >> 32)) << 32) |
                                     readfirst((uint32_t)(uintptr_t)prm.queue);  // (in scalar registers)
  for (;;) {
    // Every lane executes the same code: lane 0's returning add under a hand-set EXEC.  NOT `if (lane == 0) k = atomicAdd(..)`
    // and a broadcast: a loop is each lane's own loop to the compiler, and it turned that form into lanes 1..63 spinning
    // on the broadcast value in a loop of their own with lane 0 masked off -- which never ends.
    // (written out: the compiler's form of an add whose operand differs by lane is a loop over the 64 lanes)
    uint32_t took;
    {
      unsigned long long save;
      asm volatile(
          "s_mov_b64 %[save], exec\n\t"
          "s_mov_b64 exec, 1\n\t"
          "global_atomic_add %[r], %[zero], %[one], %[ptr] sc0\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "s_mov_b64 exec, %[save]"
          : [r] "=&v"(took), [save] "=&s"(save)
          : [zero] "v"(0u), [one] "v"(1u), [ptr] "s"(queue_s)
          : "memory");
    }
    const uint32_t k = readfirst(took);
    if (k >= prm.n_blocks) break;
    const uint64_t blk = prm.order ? prm.order[k] : k;
    if (SNAPPY_DBG(prm) && lane_id() == 0) atomicAdd(prm.queue + 1 + wave, 1u);  // DEBUG: who took how many
    if (wave == 0) {
      encode_one_block<false>(prm, blk, (enc_lds16*)s_table, nullptr, nullptr, (enc_lds8*)s_ob[0], (enc_lds8*)s_win[0], (enc_lds32*)s_cold[0],
                              (enc_lds32*)&s_walk[0]);
    } else if constexpr (kEncWaves > 1) {
      encode_one_block<true>(prm, blk, nullptr, gtab, (enc_lds16*)s_gx, (enc_lds8*)s_ob[kEncWaves - 1], (enc_lds8*)s_win[kEncWaves - 1],
                             (enc_lds32*)s_cold[kEncWaves - 1], (enc_lds32*)&s_walk[kEncWaves - 1]);
    }
  }
}

}  // namespace snappy_hip
