// framed_kernels.h -- the framing format's container on the device (snappy/framing_format.txt).
//
// Semantics: the chunk loop of uncompressFramed, snappy.nim:199-265 (header check :187-196, frame
// header parse codec.nim:166-172), for a stream that is RESIDENT IN HBM.  A framed stream has no
// chunk table: where chunk k+1 starts is written in chunk k's header, so the walk is one chain of
// dependent 16-byte loads (frame_scan_kernel, one lane; about half a microsecond per chunk).  It
// decides everything the reference decides from headers alone -- truncation, unknown chunk types,
// skippable chunks, declared lengths against the room left in the output (the (read, written)
// resume contract, snappy.nim:219-229,253-254) -- and sorts the chunks into two unit lists:
// compressed chunks for the block decoder (which checksums what it decodes), stored chunks for the
// CRC kernel and a copy.  frame_verdict_kernel then finds the first failing chunk IN STREAM ORDER
// (decode status, CRC mismatch), which is what the reference's sequential loop would have returned.
#pragma once

#include "common.h"

namespace snappy_hip {

constexpr uint32_t kStUnknownChunk = 4;  // include/snappy_hip.h: 1 + ordinal of FrameError.unknownChunk

struct FrameUnits {       // one list of chunks (structure of arrays, `cap` entries each)
  uint64_t* in_off;       // payload (behind the chunk's CRC) in the stream
  uint32_t* in_len;
  uint64_t* out_off;      // where its bytes go in the output
  uint32_t* out_cap;      // declared uncompressed length (0: checksum only, nothing delivered)
  uint32_t* crc;          // the chunk's stored masked CRC32C
  uint32_t* seq;          // ordinal of the chunk among the stream's data chunks
  uint64_t* hdr_at;       // where the chunk's header lies (the resume point)
};

struct FrameScanResult {
  uint32_t n_comp, n_stored;  // entries written to the two lists
  uint32_t overflow;          // a list was too short: run again with longer ones
  int32_t terminal;           // status that ended the walk (-1: it ran to the end of input)
  uint32_t stop_ok;           // it ended because the output is full: ok((stop_rd, stop_wr))
  int32_t tail_after;         // outcome once the LAST stored chunk's CRC has verified (-1: none;
                              //   -2: output full there; else a status) -- snappy.nim:244-254
  uint64_t stop_rd, stop_wr;
  uint64_t walk_rd;           // where the walk ended
  uint64_t deliver;           // output bytes assigned to chunks
  uint32_t need_comp, need_stored;  // list entries the walk needs (it goes on counting behind a full list)
};

struct FrameScanParams {
  const uint8_t* in;
  uint64_t n;
  uint64_t cap;          // room in the output
  int check_header;
  int check_integrity;
  FrameUnits comp, stored;
  uint32_t list_cap;
  FrameScanResult* res;
};

// LEB128, as stew/leb128 reads it (call sites snappy.nim:92, codec.nim:134); 0 = malformed
__device__ inline int dev_varint(const uint8_t* in, uint64_t n, int bits, uint64_t* val) {
  const int max_len = (bits + 6) / 7;
  uint64_t v = 0;
  for (int i = 0; i < max_len && (uint64_t)i < n; i++) {
    const uint8_t b = in[i];
    if (i == max_len - 1 && (b >> (bits - 7 * i))) return 0;
    v |= (uint64_t)(b & 0x7f) << (7 * i);
    if (!(b & 0x80)) {
      *val = v;
      return i + 1;
    }
  }
  return 0;
}

__global__ __launch_bounds__(64) void frame_scan_kernel(FrameScanParams p) {
  if (threadIdx.x != 0) return;
  const uint8_t* in = p.in;
  const uint64_t n = p.n;
  FrameScanResult r{};
  r.terminal = -1;
  r.tail_after = -1;
  uint64_t rd = 0, wr = 0;
  uint32_t seq = 0;
  bool walk = true;
  if (p.check_header) {  // snappy.nim:187-196
    const uint8_t id[10] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};
    bool same = n >= 10;
    for (int i = 0; same && i < 10; i++) same = in[i] == id[i];
    if (!same) {
      r.terminal = (int32_t)kInvalidInput;
      walk = false;
    }
    rd = 10;
  }
  while (walk && rd < n) {  // snappy.nim:199
    const uint64_t remaining = n - rd;
    if (remaining < 4) {
      r.terminal = (int32_t)kInvalidInput;
      break;
    }
    // header, CRC and the first bytes of the payload in one trip where the stream is long enough
    uint32_t w[4] = {0, 0, 0, 0};
    if (remaining >= 16) {
      uint4 v;
      __builtin_memcpy(&v, in + rd, 16);
      w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
      for (uint32_t i = 0; i < remaining; i++) w[i >> 2] |= (uint32_t)in[rd + i] << (8 * (i & 3));
    }
    const uint32_t id = w[0] & 0xff;
    const uint64_t data_len = w[0] >> 8;
    const uint64_t hdr_at = rd;
    rd += 4;
    if (remaining - 4 < data_len) {  // snappy.nim:206-207
      r.terminal = (int32_t)kInvalidInput;
      break;
    }
    if (id == 0x00) {  // snappy.nim:209-235
      if (data_len < 4) {
        r.terminal = (int32_t)kInvalidInput;
        break;
      }
      const uint64_t room = p.cap - wr;
      const uint64_t max_out = room < kMaxBlockLen ? room : kMaxBlockLen;  // snappy.nim:215
      uint8_t vb[8];
      for (int i = 0; i < 8; i++) vb[i] = (uint8_t)(w[2 + (i >> 2)] >> (8 * (i & 3)));
      uint64_t ulen = 0;
      const uint64_t avail = data_len - 4 < 8 ? data_len - 4 : 8;
      if (dev_varint(vb, avail, 32, &ulen) <= 0) {  // uncompress, snappy.nim:92
        r.terminal = (int32_t)kInvalidInput;
        break;
      }
      if (max_out < ulen) {  // bufferTooSmall inside the chunk, snappy.nim:219-227
        uint64_t u64 = 0;
        if (dev_varint(in + rd + 4, data_len - 4, 64, &u64) <= 0 || u64 > kMaxBlockLen) {
          r.terminal = (int32_t)kInvalidInput;
        } else {
          r.stop_ok = 1;
          r.stop_rd = hdr_at;
          r.stop_wr = wr;
        }
        break;
      }
      const uint32_t k = r.need_comp++;
      if (k < p.list_cap) {
        p.comp.in_off[k] = rd + 4;
        p.comp.in_len[k] = (uint32_t)(data_len - 4);
        p.comp.out_off[k] = wr;
        p.comp.out_cap[k] = (uint32_t)ulen;
        p.comp.crc[k] = w[1];
        p.comp.seq[k] = seq++;
        p.comp.hdr_at[k] = hdr_at;
      } else {
        r.overflow = 1;  // (the walk goes on, counting: the next attempt sizes the lists exactly)
      }
      wr += ulen;
    } else if (id == 0x01) {  // snappy.nim:237-257
      if (data_len < 4) {
        r.terminal = (int32_t)kInvalidInput;
        break;
      }
      const uint64_t ul = data_len - 4;
      // the reference verifies the CRC BEFORE the size checks (snappy.nim:244-254)
      if (ul > kMaxBlockLen || ul > p.cap - wr) {
        const int32_t after = ul > kMaxBlockLen ? (int32_t)kInvalidInput : -2;  // -2: output full
        if (p.check_integrity) {  // checksum it without delivering it; it ends the walk
          const uint32_t k = r.need_stored++;
          if (k < p.list_cap) {
            p.stored.in_off[k] = rd + 4;
            p.stored.in_len[k] = (uint32_t)ul;
            p.stored.out_off[k] = wr;  // (nothing is delivered: out_cap 0)
            p.stored.out_cap[k] = 0;
            p.stored.crc[k] = w[1];
            p.stored.seq[k] = seq++;
            p.stored.hdr_at[k] = hdr_at;
          } else {
            r.overflow = 1;
          }
          r.tail_after = after;
        } else if (after == -2) {
          r.stop_ok = 1;
          r.stop_rd = hdr_at;
          r.stop_wr = wr;
        } else {
          r.terminal = after;
        }
        break;
      }
      const uint32_t k = r.need_stored++;
      if (k < p.list_cap) {
        p.stored.in_off[k] = rd + 4;
        p.stored.in_len[k] = (uint32_t)ul;
        p.stored.out_off[k] = wr;
        p.stored.out_cap[k] = (uint32_t)ul;
        p.stored.crc[k] = w[1];
        p.stored.seq[k] = seq++;
        p.stored.hdr_at[k] = hdr_at;
      } else {
        r.overflow = 1;
      }
      wr += ul;
    } else if (id < 0x80) {  // snappy.nim:259-260
      r.terminal = (int32_t)kStUnknownChunk;
      break;
    }
    // 0x80..0xff skipped without validation, snappy.nim:262-263
    rd += data_len;
  }
  r.walk_rd = rd;
  r.deliver = wr;
  r.n_comp = r.need_comp < p.list_cap ? r.need_comp : p.list_cap;
  r.n_stored = r.need_stored < p.list_cap ? r.need_stored : p.list_cap;
  *p.res = r;
}

// ---------------------------------------------------------------------------------------------
// The chunk walk in parallel, for well-formed streams.  The serial walk costs about a microsecond
// per chunk (one dependent trip to HBM each): 65 ms for the 65 536 chunks of a 4 GiB stream, five
// times the decode itself.  Instead kFrameChasers waves each take a slice of the stream, FIND a chunk
// header in it by plausibility -- a data chunk's type and a sane length, three headers in a row; a
// false find has probability ~1e-13 per byte of random data -- and chase the headers from there to
// the slice's end.  frame_stitch_kernel then checks that every chaser started exactly where its
// predecessor ended (chaser 0 starts at the stream's known first chunk, so by induction every
// recorded position is a real header), frame_fill_kernel reads all chunk headers in parallel, and
// three prefix sums give every chunk its place in the output and in the two unit lists.  Anything
// irregular -- a chaser that found nothing or not where its predecessor ended, an unknown or
// malformed chunk, an output that does not fit -- sets `irregular`, and the caller runs the serial
// walk, whose verdicts are the reference's for every such case.
constexpr uint32_t kFrameChasers = 2048;
constexpr uint32_t kChaserList = 4096;   // chunk headers one chaser can record

struct FrameChase {
  uint64_t start, end;  // first header found (~0: none in the slice), where the chase stopped
  uint32_t count;       // headers recorded
  uint32_t bad;         // list overflow / a header that runs past the stream
};

__device__ __forceinline__ uint32_t ld_hdr(const uint8_t* in, uint64_t p) { return ld32u(in + p); }
// a data chunk header (snappy.nim:209, :237) as a real encoder writes it: a stored chunk of at most
// 65 536 bytes; a compressed chunk whose varint declares 1..65 536 bytes and whose body is no longer
// than maxCompressedLen of that (codec.nim:92) and no shorter than 3 bytes per 64 (the densest copy)
__device__ __forceinline__ bool plausible_data(const uint8_t* in, uint64_t n, uint64_t p, uint64_t* next) {
  if (p + 12 > n) return false;
  const uint32_t h = ld_hdr(in, p);
  const uint32_t id = h & 0xff, len = h >> 8;
  if (id > 1 || len < 5 || p + 4 + len > n) return false;
  if (id == 1) {
    if (len - 4 > kMaxBlockLen) return false;
  } else {
    uint64_t ulen = 0;
    const int vl = dev_varint(in + p + 8, len - 4 < 4 ? len - 4 : 4, 32, &ulen);
    if (vl <= 0 || ulen == 0 || ulen > kMaxBlockLen) return false;
    const uint64_t body = len - 4 - (uint32_t)vl;
    if (body > 32 + ulen + ulen / 6 || body * 64 < ulen * 3) return false;
  }
  *next = p + 4 + len;
  return true;
}
// ... or the stream's end, or a stream identifier (framing_format.txt: may repeat)
__device__ __forceinline__ bool plausible_next(const uint8_t* in, uint64_t n, uint64_t p, uint64_t* next) {
  if (p == n) {
    *next = n;
    return true;
  }
  if (plausible_data(in, n, p, next)) return true;
  if (p + 10 <= n && ld_hdr(in, p) == 0x000006ffu) {
    *next = p + 10;
    return true;
  }
  return false;
}

__global__ __launch_bounds__(64) void frame_chase_kernel(const uint8_t* in, uint64_t n, uint64_t p0, uint64_t slice,
                                                         FrameChase* chase, uint64_t* lists) {
  const uint32_t k = blockIdx.x, lane = threadIdx.x;
  const uint64_t lo = k == 0 ? p0 : (uint64_t)k * slice;
  const uint64_t hi = (uint64_t)(k + 1) * slice < n ? (uint64_t)(k + 1) * slice : n;
  FrameChase r{};
  r.start = ~0ull;
  r.end = lo;
  uint64_t start = ~0ull;
  if (k == 0) {
    start = lo < n ? lo : ~0ull;
  } else {
    // 1 KiB per step: every lane takes 16 positions and tells from their four header bytes alone (one
    // coalesced load) which could be a data chunk's header at all -- type 0 or 1, a length a chunk can have;
    // the few that pass get the full test, three headers deep (dependent trips to memory).  A step per 64
    // positions spent a millisecond on the half chunk that lies in front of a slice's first header.
    for (uint64_t a = lo; a < hi && start == ~0ull; a += 1024) {
      const uint64_t p16 = a + 16 * lane;
      uint32_t w[5] = {0, 0, 0, 0, 0};
      if (p16 + 20 <= n) {
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = ld32u(in + p16 + 4 * i);
      } else {
        for (uint32_t i = 0; i < 20 && p16 + i < n; i++) w[i >> 2] |= (uint32_t)in[p16 + i] << (8 * (i & 3));
      }
      uint32_t cand = 0;  // bit j: position p16 + j passes the first look
#pragma unroll
      for (uint32_t j = 0; j < 16; j++) {
        const uint32_t sh = (j & 3) * 8;
        const uint32_t h = sh ? __funnelshift_r(w[j >> 2], w[(j >> 2) + 1], sh) : w[j >> 2];
        const uint32_t id = h & 0xff, len = h >> 8;
        if (id <= 1 && len >= 5 && len <= kMaxCompressedBlockLen + 16 && p16 + j < hi) cand |= 1u << j;
      }
      uint64_t best = ~0ull;  // my first position that passes the full test
      while (cand && best == ~0ull) {
        const uint32_t j = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1;
        uint64_t q1 = 0, q2 = 0, q3 = 0;
        if (plausible_data(in, n, p16 + j, &q1) && plausible_next(in, n, q1, &q2) && plausible_next(in, n, q2, &q3))
          best = p16 + j;
      }
      const uint64_t m = __ballot(best != ~0ull);  // (lanes hold ascending ranges: the first lane's find is the first)
      if (m) start = __shfl(best, (int)__builtin_ctzll(m), 64);
    }
  }
  if (lane == 0) {
    uint64_t* list = lists + (uint64_t)k * kChaserList;
    if (start != ~0ull) {
      r.start = start;
      uint64_t pos = start;
      while (pos < hi) {
        if (pos + 4 > n || r.count >= kChaserList) {
          r.bad = 1;
          break;
        }
        list[r.count++] = pos;
        const uint64_t next = pos + 4 + (ld_hdr(in, pos) >> 8);
        if (next > n) {
          r.bad = 1;
          break;
        }
        pos = next;
      }
      r.end = pos;
    }
    chase[k] = r;
  }
}

struct FrameStitch {
  uint32_t irregular;  // the parallel walk does not apply: run the serial one
  uint32_t n_chunks;   // headers of the whole stream, in order
};

// One workgroup.  A chaser's find may be a false one (structured data is full of plausible-looking
// bytes), but a chain of "headers" that starts wrong falls into step with the real chain as soon as it
// lands on a real header, and stays there: where a chaser STOPPED is then real, and the next chaser's
// list must contain that position -- from there on its entries are real (chaser 0 starts at the
// stream's known first chunk; induction).  first[k] = index of that entry in chaser k's list,
// base[k] = real chunks in front of chaser k.
__global__ __launch_bounds__(1024) void frame_stitch_kernel(const FrameChase* chase, const uint64_t* lists,
                                                            uint64_t n, uint64_t p0, uint64_t slice,
                                                            uint32_t* base, uint32_t* first, FrameStitch* out) {
  __shared__ uint64_t s_expect[kFrameChasers], s_end[kFrameChasers];
  __shared__ uint32_t s_cnt[kFrameChasers], s_eff[kFrameChasers];
  __shared__ uint32_t s_irregular;
  for (uint32_t k = threadIdx.x; k < kFrameChasers; k += blockDim.x) {
    const FrameChase c = chase[k];
    s_end[k] = c.end;
    s_cnt[k] = c.bad ? 0xffffffffu : c.count;
  }
  if (threadIdx.x == 0) s_irregular = 0;
  __syncthreads();
  // Where the chain enters every slice, if all chasers ended on it: where the chaser before it stopped.  (The serial form
  // of this -- one thread, two loops over the 2 048 chasers, an LDS round trip each -- took 0.15 ms of a framed decode.  It
  // also let the chain pass OVER a slice; a data chunk is at most 76 KiB and a slice at least 1 MiB, so in a stream this
  // walk applies to that only happens to the empty slices behind the stream's end -- and to the LAST slice with bytes in it,
  // which holds n mod slice bytes: when the final chunk starts in front of it the chain ends at n without entering it, and
  // whatever its chaser took for headers inside that chunk's body is ignored.  Any other non-empty slice the chain does
  // not enter makes the stream irregular here, and the serial walk decides.)
  for (uint32_t k = threadIdx.x; k < kFrameChasers; k += blockDim.x) {
    const uint64_t lo = k == 0 ? p0 : (uint64_t)k * slice;
    const uint64_t hi = (uint64_t)(k + 1) * slice < n ? (uint64_t)(k + 1) * slice : n;
    uint64_t expect = ~0ull;
    if (lo < n || k == 0) {  // (a slice with bytes in it)
      expect = k == 0 ? p0 : s_end[k - 1];
      if (expect >= hi) {
        // (a stream of the header alone: nothing to walk; the chain ended at the stream's end in front of the last slice)
        if (!(k == 0 && p0 >= n) && !(expect == n && hi == n)) s_irregular = 1;
        expect = ~0ull;
      }
      if (hi == n && s_end[k] != n && expect != ~0ull) s_irregular = 1;  // the last chaser the chain enters must end at the stream's end
    }
    s_expect[k] = expect;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < kFrameChasers; k += blockDim.x) {  // ... and whether they did
    uint32_t eff = 0, f = 0;
    const uint64_t want = s_expect[k];
    if (want != ~0ull) {
      const uint32_t cnt = s_cnt[k];
      const uint64_t* list = lists + (uint64_t)k * kChaserList;
      uint32_t lo = 0, hi = cnt == 0xffffffffu ? 0 : cnt;  // first entry >= want (lists ascend)
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (list[mid] < want) lo = mid + 1; else hi = mid;
      }
      if (cnt == 0xffffffffu || lo >= cnt || list[lo] != want) {
        s_irregular = 1;
      } else {
        f = lo;
        eff = cnt - lo;
      }
    }
    first[k] = f;
    s_eff[k] = eff;
  }
  __syncthreads();
  // base = exclusive prefix sum of the chasers' real chunks (two per thread, a scan in place)
  static_assert(kFrameChasers == 2048, "two chasers per thread of the 1 024");
  {
    const uint32_t t = threadIdx.x;
    const uint32_t a0 = s_eff[2 * t], a1 = s_eff[2 * t + 1];
    __syncthreads();
    s_cnt[t] = a0 + a1;  // (s_cnt is free now)
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t v = t >= d ? s_cnt[t - d] : 0;
      __syncthreads();
      s_cnt[t] += v;
      __syncthreads();
    }
    const uint32_t before = s_cnt[t] - (a0 + a1);
    base[2 * t] = before;
    base[2 * t + 1] = before + a0;
    if (t == 1023) {
      FrameStitch r{};
      base[kFrameChasers] = s_cnt[t];
      r.irregular = s_irregular;
      r.n_chunks = s_cnt[t];
      *out = r;
    }
  }
}

struct FrameFillParams {
  const uint8_t* in;
  uint64_t n;
  const uint64_t* lists;
  const uint32_t* base;
  const uint32_t* first;
  const FrameStitch* stitch;
  uint64_t* pos;      // per chunk: header position
  uint32_t* ulen;     // ... declared uncompressed length (0 for chunks that carry no data)
  uint32_t* is_comp;  // ... 1 for a compressed chunk
  uint32_t* is_stored;
  uint32_t* irregular;  // set when a chunk needs the serial walk's judgement
};

__global__ __launch_bounds__(256) void frame_fill_kernel(FrameFillParams p) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const FrameStitch st = *p.stitch;
  if (st.irregular || j >= st.n_chunks) return;
  uint32_t lo = 0, hi = kFrameChasers;  // the chaser whose list holds chunk j: base[lo] <= j < base[lo + 1]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (p.base[mid] <= j) lo = mid; else hi = mid;
  }
  const uint64_t pos = p.lists[(uint64_t)lo * kChaserList + p.first[lo] + (j - p.base[lo])];
  const uint32_t h = ld_hdr(p.in, pos);
  const uint32_t id = h & 0xff;
  const uint64_t data_len = h >> 8;
  uint32_t ulen = 0, comp = 0, stored = 0;
  bool odd = false;
  if (id == 0x00) {  // snappy.nim:209-235
    uint64_t v = 0;
    odd = data_len < 4 || dev_varint(p.in + pos + 8, data_len - 4, 32, &v) <= 0 || v > kMaxBlockLen;
    ulen = (uint32_t)v;
    comp = 1;
  } else if (id == 0x01) {  // snappy.nim:237-257
    odd = data_len < 4 || data_len - 4 > kMaxBlockLen;
    ulen = (uint32_t)(data_len - 4);
    stored = 1;
  } else if (id < 0x80) {  // snappy.nim:259-260
    odd = true;
  }
  if (odd) *p.irregular = 1;
  p.pos[j] = pos;
  p.ulen[j] = odd ? 0 : ulen;
  p.is_comp[j] = comp;
  p.is_stored[j] = stored;
}

struct FrameScatterParams {
  const uint8_t* in;
  uint64_t n, cap;
  const FrameStitch* stitch;
  const uint32_t* irregular;
  const uint64_t* pos;
  const uint32_t *ulen, *is_comp, *is_stored;
  const uint64_t *out_at, *comp_at, *stored_at;  // exclusive prefix sums of the three
  FrameUnits comp, stored;
  uint32_t list_cap;
  int check_header;   // the stream must start with the stream identifier (snappy.nim:187-196)
  FrameScanResult* res;
  uint32_t* fast_ok;  // 1: the unit lists and *res are complete
};

__global__ __launch_bounds__(256) void frame_scatter_kernel(FrameScatterParams p) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const FrameStitch st = *p.stitch;
  const uint32_t nc = st.n_chunks;
  // (every thread evaluates the same verdict from the same words)
  bool ok = !st.irregular && !*p.irregular && nc > 0 && p.out_at[nc] <= p.cap &&
            p.comp_at[nc] <= p.list_cap && p.stored_at[nc] <= p.list_cap;
  if (ok && p.check_header) ok = p.n >= 10 && ld32u(p.in) == 0x000006ffu && ld32u(p.in + 4) == 0x50614e73u &&
                                 p.in[8] == 0x70 && p.in[9] == 0x59;
  if (j == 0) {
    *p.fast_ok = ok ? 1 : 0;
    if (ok) {
      FrameScanResult r{};
      r.n_comp = (uint32_t)p.comp_at[nc];
      r.n_stored = (uint32_t)p.stored_at[nc];
      r.terminal = -1;
      r.tail_after = -1;
      r.walk_rd = p.n;
      r.deliver = p.out_at[nc];
      *p.res = r;
    }
  }
  if (!ok || j >= nc) return;
  const uint64_t pos = p.pos[j];
  const uint32_t data_len = ld_hdr(p.in, pos) >> 8;
  const uint32_t seq = (uint32_t)(p.comp_at[j] + p.stored_at[j]);
  // the CRC field exists only in data chunks (frame_fill verified data_len >= 4 for them); a skippable or
  // padding chunk of fewer than 4 bytes may end the stream, and pos + 4 is then the end of the caller's buffer
  const uint32_t crc = (p.is_comp[j] || p.is_stored[j]) ? ld32u(p.in + pos + 4) : 0u;
  if (p.is_comp[j]) {
    const uint32_t k = (uint32_t)p.comp_at[j];
    p.comp.in_off[k] = pos + 8;
    p.comp.in_len[k] = data_len - 4;
    p.comp.out_off[k] = p.out_at[j];
    p.comp.out_cap[k] = p.ulen[j];
    p.comp.crc[k] = crc;
    p.comp.seq[k] = seq;
    p.comp.hdr_at[k] = pos;
  } else if (p.is_stored[j]) {
    const uint32_t k = (uint32_t)p.stored_at[j];
    p.stored.in_off[k] = pos + 8;
    p.stored.in_len[k] = data_len - 4;
    p.stored.out_off[k] = p.out_at[j];
    p.stored.out_cap[k] = p.ulen[j];
    p.stored.crc[k] = crc;
    p.stored.seq[k] = seq;
    p.stored.hdr_at[k] = pos;
  }
}

// Stored chunks: payload -> output (snappy.nim:256).  One workgroup per chunk.
__global__ __launch_bounds__(256) void copy_units_kernel(const uint8_t* in, const uint64_t* in_off,
                                                         const uint32_t* out_cap, const uint64_t* out_off,
                                                         const FrameScanResult* res, uint8_t* out) {
  const uint32_t u = blockIdx.x;
  if (u >= res->n_stored) return;
  const uint32_t n = out_cap[u];  // (0 for a chunk that is only checksummed)
  const uint8_t* src = in + in_off[u];
  uint8_t* dst = out + out_off[u];
  const uint32_t t = threadIdx.x;
  uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
  if (head > n) head = n;
  if (t < head) dst[t] = src[t];
  const uint32_t body = (n - head) & ~15u;
  for (uint32_t i = t * 16; i < body; i += 256 * 16) {
    uint4 v;
    __builtin_memcpy(&v, src + head + i, 16);
    *reinterpret_cast<uint4*>(dst + head + i) = v;
  }
  const uint32_t tail0 = head + body;
  if (tail0 + t < n) dst[tail0 + t] = src[tail0 + t];
}

struct FrameVerdict {  // what uncompressFramed returns
  uint32_t status;
  uint32_t n_units;
  uint64_t read, written;
};

struct FrameVerdictParams {
  FrameUnits comp, stored;
  const FrameScanResult* res;
  const uint32_t* comp_status;  // per compressed chunk: decode status
  const uint32_t* comp_crc;     // ... and the masked CRC32C of what was decoded
  const uint32_t* stored_crc;   // per stored chunk: the masked CRC32C of its payload
  int check_integrity;
  FrameVerdict* out;
};

// The first failing chunk in stream order wins (snappy.nim:228, :231-233, :244-246); chunks in
// front of it were delivered.  One workgroup; the lists are walked with a stride.
__global__ __launch_bounds__(1024) void frame_verdict_kernel(FrameVerdictParams p) {
  __shared__ unsigned long long s_first;  // (seq << 32) | list (0 comp, 1 stored) << 31 | index
  const FrameScanResult r = *p.res;
  if (threadIdx.x == 0) s_first = ~0ull;
  __syncthreads();
  unsigned long long mine = ~0ull;
  // (one workgroup, so what it takes is trips to HBM one behind the other: sixteen entries' loads in flight together --
  // four trips for the 59 000 compressed chunks of a 4 GiB stream instead of the 58 of a plain loop)
  constexpr uint32_t kDeep = 16;
  for (uint32_t i0 = threadIdx.x; i0 < r.n_comp; i0 += kDeep * blockDim.x) {
    uint32_t st[kDeep], got[kDeep], want[kDeep];
#pragma unroll
    for (uint32_t k = 0; k < kDeep; k++) {
      // (unconditional loads from a clamped index -- the lists have at least 16 entries --: a load under a condition is
      // waited for where the branches join, one trip after the other)
      const uint32_t i = i0 + k * blockDim.x;
      const uint32_t ic = i < r.n_comp ? i : 0;
      st[k] = p.comp_status[ic];
      got[k] = p.comp_crc[ic];
      want[k] = p.comp.crc[ic];
    }
#pragma unroll
    for (uint32_t k = 0; k < kDeep; k++) {
      const uint32_t i = i0 + k * blockDim.x;
      if (i < r.n_comp && (st[k] != kOk || (p.check_integrity && got[k] != want[k]))) {
        const unsigned long long key = ((unsigned long long)p.comp.seq[i] << 32) | i;
        mine = key < mine ? key : mine;
      }
    }
  }
  for (uint32_t i0 = threadIdx.x; i0 < r.n_stored; i0 += 8 * blockDim.x) {
    uint32_t got[8], want[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t i = i0 + k * blockDim.x;
      const uint32_t ic = i < r.n_stored ? i : 0;
      got[k] = p.stored_crc[ic];
      want[k] = p.stored.crc[ic];
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t i = i0 + k * blockDim.x;
      const bool last_tail = r.tail_after != -1 && i + 1 == r.n_stored;  // (a chunk that ends the walk)
      if (i < r.n_stored && ((p.check_integrity && got[k] != want[k]) || last_tail)) {
        const unsigned long long key = ((unsigned long long)p.stored.seq[i] << 32) | (1ull << 31) | i;
        mine = key < mine ? key : mine;
      }
    }
  }
  if (mine != ~0ull) atomicMin(&s_first, mine);
  __syncthreads();
  if (threadIdx.x != 0) return;
  FrameVerdict v{};
  v.n_units = r.n_comp + r.n_stored;
  const unsigned long long f = s_first;
  if (f != ~0ull) {
    const bool st = (f >> 31) & 1;
    const uint32_t i = (uint32_t)(f & 0x7fffffffu);
    if (!st) {
      v.status = p.comp_status[i] != kOk ? kInvalidInput : kCrcMismatch;
      v.written = p.comp.out_off[i];
    } else if (p.check_integrity && p.stored_crc[i] != p.stored.crc[i]) {
      v.status = kCrcMismatch;
      v.written = p.stored.out_off[i];
    } else {  // the checksum-only chunk at the end of the walk verified: snappy.nim:250-254
      if (r.tail_after == -2) {  // output full there: ok((read-4, written))
        v.status = kOk;
        v.read = p.stored.hdr_at[i];
        v.written = r.deliver;
      } else {
        v.status = (uint32_t)r.tail_after;
        v.written = r.deliver;
      }
    }
    *p.out = v;
    return;
  }
  if (r.terminal >= 0) {
    v.status = (uint32_t)r.terminal;
    v.written = r.deliver;
  } else if (r.stop_ok) {
    v.status = kOk;
    v.read = r.stop_rd;
    v.written = r.stop_wr;
  } else {
    v.status = kOk;
    v.read = r.walk_rd;
    v.written = r.deliver;
  }
  *p.out = v;
}

}  // namespace snappy_hip
