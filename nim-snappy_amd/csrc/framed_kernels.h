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
      if (r.n_comp >= p.list_cap) {
        r.overflow = 1;
        break;
      }
      const uint32_t k = r.n_comp++;
      p.comp.in_off[k] = rd + 4;
      p.comp.in_len[k] = (uint32_t)(data_len - 4);
      p.comp.out_off[k] = wr;
      p.comp.out_cap[k] = (uint32_t)ulen;
      p.comp.crc[k] = w[1];
      p.comp.seq[k] = seq++;
      p.comp.hdr_at[k] = hdr_at;
      wr += ulen;
    } else if (id == 0x01) {  // snappy.nim:237-257
      if (data_len < 4) {
        r.terminal = (int32_t)kInvalidInput;
        break;
      }
      const uint64_t ul = data_len - 4;
      if (r.n_stored >= p.list_cap) {
        r.overflow = 1;
        break;
      }
      // the reference verifies the CRC BEFORE the size checks (snappy.nim:244-254)
      if (ul > kMaxBlockLen || ul > p.cap - wr) {
        const int32_t after = ul > kMaxBlockLen ? (int32_t)kInvalidInput : -2;  // -2: output full
        if (p.check_integrity) {  // checksum it without delivering it; it ends the walk
          const uint32_t k = r.n_stored++;
          p.stored.in_off[k] = rd + 4;
          p.stored.in_len[k] = (uint32_t)ul;
          p.stored.out_off[k] = wr;  // (nothing is delivered: out_cap 0)
          p.stored.out_cap[k] = 0;
          p.stored.crc[k] = w[1];
          p.stored.seq[k] = seq++;
          p.stored.hdr_at[k] = hdr_at;
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
      const uint32_t k = r.n_stored++;
      p.stored.in_off[k] = rd + 4;
      p.stored.in_len[k] = (uint32_t)ul;
      p.stored.out_off[k] = wr;
      p.stored.out_cap[k] = (uint32_t)ul;
      p.stored.crc[k] = w[1];
      p.stored.seq[k] = seq++;
      p.stored.hdr_at[k] = hdr_at;
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
  *p.res = r;
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
  for (uint32_t i = threadIdx.x; i < r.n_comp; i += blockDim.x) {
    const bool fail = p.comp_status[i] != kOk || (p.check_integrity && p.comp_crc[i] != p.comp.crc[i]);
    if (fail) {
      const unsigned long long key = ((unsigned long long)p.comp.seq[i] << 32) | i;
      mine = key < mine ? key : mine;
    }
  }
  for (uint32_t i = threadIdx.x; i < r.n_stored; i += blockDim.x) {
    const bool last_tail = r.tail_after != -1 && i + 1 == r.n_stored;  // (a chunk that ends the walk)
    const bool fail = (p.check_integrity && p.stored_crc[i] != p.stored.crc[i]) || last_tail;
    if (fail) {
      const unsigned long long key = ((unsigned long long)p.stored.seq[i] << 32) | (1ull << 31) | i;
      mine = key < mine ? key : mine;
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
