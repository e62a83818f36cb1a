/*
 * snappy_oracle.c -- CPU restatement of the nim-snappy hot path.  TEST INFRASTRUCTURE ONLY:
 * see snappy_oracle.h.  Never linked into, loaded by or called from the product library.
 *
 * Written from the observable behaviour of the reference, not translated line
 * by line: writes are exact-length (the reference's 16-byte over-writes are invisible in its
 * output), match extension is a plain common-prefix scan, the CRC tables are generated.
 */
#include "snappy_oracle.h"

#include <string.h>

/* ---- format constants (codec.nim:9-34, :53) -------------------------------------------- */
#define MAX_BLOCK_LEN 65536u       /* codec.nim:14 maxBlockLen */
#define MAX_FRAME_DATA_LEN 65536u  /* codec.nim:18 maxUncompressedFrameDataLen */
#define INPUT_MARGIN 15            /* codec.nim:26 */
#define MIN_NON_LITERAL 17         /* codec.nim:53 minNonLiteralBlockSize */
#define MAX_TABLE_BITS 14          /* encoder.nim:11 */
#define MAX_TABLE_SIZE (1u << MAX_TABLE_BITS)

static const uint8_t FRAMING_HEADER[10] = {0xff, 0x06, 0x00, 0x00, 0x73,
                                           0x4e, 0x61, 0x50, 0x70, 0x59}; /* codec.nim:33 */

static inline uint32_t load32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ---- LEB128 (stew/leb128, un-vendored; call sites snappy.nim:49,92 codec.nim:134) -------
 * Published algorithm: 7 value bits per byte, little-endian groups, bit 7 = continuation.
 * Parsing stops after maxLen bytes (5 for u32, 10 for u64); the last byte may only carry the
 * bits that still fit.  Returns bytes consumed, or <= 0 on truncation / overflow.
 * Non-minimal encodings (e.g. 80 00) are accepted -- parity unpinned, see oracle/README.md. */
static int varint_decode(const uint8_t* in, size_t n, int bits, uint64_t* val) {
  const int max_len = (bits + 6) / 7;
  uint64_t v = 0;
  int shift = 0;
  for (int i = 0; i < max_len && (size_t)i < n; i++) {
    uint8_t b = in[i];
    if (i == max_len - 1) {
      /* last permitted byte: only (bits - shift) value bits fit and it must terminate */
      if (b >> (bits - shift)) return -(i + 1);
    }
    v |= (uint64_t)(b & 0x7f) << shift;
    shift += 7;
    if (!(b & 0x80)) {
      *val = v;
      return i + 1;
    }
  }
  return n == 0 ? 0 : -(int)((size_t)max_len < n ? (size_t)max_len : n);
}

static int varint_encode_u32(uint32_t v, uint8_t* out) {
  int i = 0;
  while (v >= 0x80) {
    out[i++] = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  out[i++] = (uint8_t)v;
  return i;
}

/* ---- CRC-32C (crc32c.c:204-214 single table, :676-731 slicing-by-8, :759-763 mask) -----
 * Castagnoli polynomial, reflected 0x82f63b78, init and final xor 0xffffffff.  Tables are
 * generated, not copied; the values are checked against the compiled reference
 * (oracle/_ref/libref_crc32c.so) by tests/test_oracle_crc.py. */
static uint32_t crc_tab[8][256];

__attribute__((constructor)) static void crc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0x82f63b78u : 0);
    crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++)
      crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
}

static uint32_t crc32c_update(uint32_t crc, const uint8_t* p, size_t n) {
  while (n >= 8) {
    uint32_t lo = load32(p) ^ crc, hi = load32(p + 4);
    crc = crc_tab[7][lo & 0xff] ^ crc_tab[6][(lo >> 8) & 0xff] ^ crc_tab[5][(lo >> 16) & 0xff] ^
          crc_tab[4][lo >> 24] ^ crc_tab[3][hi & 0xff] ^ crc_tab[2][(hi >> 8) & 0xff] ^
          crc_tab[1][(hi >> 16) & 0xff] ^ crc_tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = crc_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
  return crc;
}

uint32_t sor_crc32c(const uint8_t* buf, size_t n) {
  return ~crc32c_update(0xffffffffu, buf, n);
}

uint32_t sor_masked_crc32c(const uint8_t* buf, size_t n) {
  uint32_t crc = sor_crc32c(buf, n); /* crc32c.c:761 */
  return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; /* crc32c.c:762 */
}

/* ---- size helpers ------------------------------------------------------------------------ */
uint64_t sor_max_compressed_len(uint32_t n) { /* codec.nim:117-120 */
  return 32u + (uint64_t)n + (uint64_t)n / 6u;
}

uint64_t sor_max_compressed_len_framed(int64_t n) { /* codec.nim:140-164 */
  if (n <= 0) return sizeof FRAMING_HEADER;
  uint64_t frames = ((uint64_t)n + MAX_FRAME_DATA_LEN - 1) / MAX_FRAME_DATA_LEN;
  return (frames - 1) * (MAX_FRAME_DATA_LEN + 8) + sor_max_compressed_len(MAX_FRAME_DATA_LEN) + 8 +
         sizeof FRAMING_HEADER;
}

int sor_uncompressed_len(const uint8_t* in, size_t n, uint64_t* len) { /* codec.nim:129-138 */
  uint64_t v;
  if (varint_decode(in, n, 64, &v) <= 0) return SOR_INVALID_INPUT;
  *len = v;
  return SOR_OK;
}

/* ---- block encoder ------------------------------------------------------------------------ */

/* encoder.nim:44-73.  1 <= len <= 65536.  (The `fast` variant differs only in bytes past the
 * element, which later elements overwrite.) */
static size_t emit_literal(uint8_t* dst, const uint8_t* src, size_t len) {
  uint32_t n = (uint32_t)len - 1;
  size_t w;
  if (n < 60) {
    dst[0] = (uint8_t)(n << 2);
    w = 1;
  } else if (n < 256) {
    dst[0] = 60 << 2;
    dst[1] = (uint8_t)n;
    w = 2;
  } else {
    dst[0] = 61 << 2;
    dst[1] = (uint8_t)n;
    dst[2] = (uint8_t)(n >> 8);
    w = 3;
  }
  memcpy(dst + w, src, len);
  return w + len;
}

/* encoder.nim:81-125.  1 <= offset <= 65535, 4 <= length <= 65535. */
static size_t emit_copy(uint8_t* dst, uint32_t offset, uint32_t length) {
  size_t w = 0;
  while (length >= 68) { /* :97-103 length-64 copy2 */
    dst[w] = (63 << 2) | 2;
    dst[w + 1] = (uint8_t)offset;
    dst[w + 2] = (uint8_t)(offset >> 8);
    w += 3;
    length -= 64;
  }
  if (length > 64) { /* :105-112 length-60 copy2 */
    dst[w] = (59 << 2) | 2;
    dst[w + 1] = (uint8_t)offset;
    dst[w + 2] = (uint8_t)(offset >> 8);
    w += 3;
    length -= 60;
  }
  if (length >= 12 || offset >= 2048) { /* :114-120 */
    dst[w] = (uint8_t)(((length - 1) << 2) | 2);
    dst[w + 1] = (uint8_t)offset;
    dst[w + 2] = (uint8_t)(offset >> 8);
    return w + 3;
  }
  dst[w] = (uint8_t)(((offset >> 8) << 5) | ((length - 4) << 2) | 1); /* :123 copy1 */
  dst[w + 1] = (uint8_t)offset;
  return w + 2;
}

size_t sor_encode_block_ex(const uint8_t* in, size_t n, uint8_t* out, unsigned flags) {
  const long len = (long)n;
  size_t op = 0;
  if (n < MIN_NON_LITERAL) return emit_literal(out, in, n); /* encoder.nim:227-229 */

  /* encoder.nim:27-34, :234-238 */
  uint32_t table_size = 1u << 8;
  while (table_size < MAX_TABLE_SIZE && table_size < n) table_size *= 2;
  const uint32_t mask = table_size - 1;
  uint32_t shift = 32 - MAX_TABLE_BITS; /* encoder.nim:36-37: fixed >>18, then mask */
  if (flags & SOR_ENC_CPP_SHIFT) {
    uint32_t lg = 0;
    while ((1u << lg) < table_size) lg++;
    shift = 32 - lg;
  }
  uint16_t table[MAX_TABLE_SIZE];
  memset(table, 0, table_size * sizeof table[0]);
#define HASH(u) ((((uint32_t)(u) * 0x1e35a7bdu) >> shift) & mask)

  const long ip_limit = len - INPUT_MARGIN; /* encoder.nim:245 */
  long ip = 0, next_emit;
  uint32_t candidate;

  for (;;) {
    /* encoder.nim:272-273 */
    next_emit = ip;
    ip += 1;
    uint32_t skip = 32;
    int found = 0;

    /* encoder.nim:280-309: sixteen unconditional probes at ip .. ip+15 */
    if (ip_limit >= ip + 16) {
      for (int i = 0; i < 16; i++) {
        uint32_t dword = load32(in + ip + i);
        uint32_t h = HASH(dword);
        candidate = table[h];
        table[h] = (uint16_t)(ip + i);
        if (load32(in + candidate) == dword) {
          ip += i;
          found = 1;
          break;
        }
      }
      if (!found) {
        ip += 16;
        skip += 16;
      }
    }
    /* encoder.nim:311-331: skip-accelerated probing */
    while (!found) {
      uint32_t dword = load32(in + ip);
      uint32_t h = HASH(dword);
      uint32_t step = skip >> 5;
      skip += step;
      long next_ip = ip + (long)step;
      if (next_ip > ip_limit) { /* :319-321: bail out BEFORE the table write */
        ip = next_emit;
        goto remainder;
      }
      candidate = table[h];
      table[h] = (uint16_t)ip;
      if (dword == load32(in + candidate)) break;
      ip = next_ip;
    }
    /* encoder.nim:336-340: literal input[next_emit ..< ip] (empty never happens here) */
    op += emit_literal(out + op, in + next_emit, (size_t)(ip - next_emit));

    /* encoder.nim:350-380: copy, then keep copying while the next position matches too */
    for (;;) {
      long base = ip;
      long s1 = (long)candidate + 4, s2 = ip + 4;
      while (s2 < len && in[s1] == in[s2]) { /* findMatchLength, encoder.nim:130-182 */
        s1++;
        s2++;
      }
      uint32_t matched = (uint32_t)(s2 - base);
      ip += matched;
      op += emit_copy(out + op, (uint32_t)(base - (long)candidate), matched);

      if ((flags & SOR_ENC_CPP_GE_LIMIT) ? ip >= ip_limit : ip > ip_limit) goto remainder; /* :362 */

      table[HASH(load32(in + ip - 1))] = (uint16_t)(ip - 1); /* :371 */
      uint32_t dword = load32(in + ip);
      uint32_t h = HASH(dword);
      candidate = table[h]; /* :376-377 */
      table[h] = (uint16_t)ip;
      if (dword != load32(in + candidate)) break; /* :379 */
    }
  }

remainder: /* encoder.nim:249-253 */
  if (ip < len) op += emit_literal(out + op, in + ip, (size_t)(len - ip));
  return op;
#undef HASH
}

size_t sor_encode_block(const uint8_t* in, size_t n, uint8_t* out) {
  return sor_encode_block_ex(in, n, out, 0);
}

/* ---- block decoder (decoder.nim:20-155) ------------------------------------------------- */
int sor_decode_all_tags(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  *written = 0;
  if (n == 0) return SOR_OK;               /* :26-27 */
  if (cap == 0) return SOR_BUFFER_TOO_SMALL; /* :29-30 */
  size_t ip = 0, op = 0;
  while (ip < n) {
    uint8_t tag = in[ip];
    size_t length;
    uint32_t offset;
    switch (tag & 3) {
      case 0: { /* literal, :42-84 */
        ip += 1;
        length = (size_t)(tag >> 2) + 1;
        if (length >= 61) {
          /* :54-57 -- requires 61 bytes after the tag even when the literal is shorter */
          if (n - ip < 61) return SOR_INVALID_INPUT;
          size_t lenlen = length - 60;
          uint32_t m = lenlen == 4 ? 0xffffffffu : ((1u << (8 * lenlen)) - 1);
          uint32_t len32 = (load32(in + ip) & m) + 1;
          if (len32 == 0) return SOR_INVALID_INPUT; /* :67-68 */
          length = len32;
          ip += lenlen;
        }
        if (cap - op < length || n - ip < length) return SOR_INVALID_INPUT; /* :77-79 */
        memcpy(out + op, in + ip, length);
        op += length;
        ip += length;
        continue;
      }
      case 1: /* copy1, :86-94 */
        if (n - ip < 2) return SOR_INVALID_INPUT;
        length = 4 + ((tag >> 2) & 7);
        offset = ((uint32_t)(tag & 0xe0) << 3) | in[ip + 1];
        ip += 2;
        break;
      case 2: /* copy2, :95-102 */
        if (n - ip < 3) return SOR_INVALID_INPUT;
        length = 1 + (size_t)(tag >> 2);
        offset = (uint32_t)in[ip + 1] | ((uint32_t)in[ip + 2] << 8);
        ip += 3;
        break;
      default: /* copy4, :103-109 */
        if (n - ip < 5) return SOR_INVALID_INPUT;
        length = 1 + (size_t)(tag >> 2);
        offset = load32(in + ip + 1);
        ip += 5;
        break;
    }
    /* :112 -- op is compared as uint32: offset 0 wraps to 0xffffffff and always fails */
    if ((uint32_t)op <= offset - 1u) return SOR_INVALID_INPUT;
    if (cap - op < length) return SOR_INVALID_INPUT; /* :127-128 */
    const uint8_t* src = out + op - offset;
    for (size_t i = 0; i < length; i++) out[op + i] = src[i]; /* forward, overlap replicates */
    op += length;
  }
  *written = op;
  return SOR_OK;
}

/* ---- in-memory API (snappy.nim) ---------------------------------------------------------- */
int sor_compress_ex(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written,
                    unsigned flags) {
  *written = 0;
  if ((uint64_t)n > 0xffffffffu) return SOR_INVALID_INPUT;                 /* snappy.nim:41-42 */
  if ((uint64_t)cap < sor_max_compressed_len((uint32_t)n)) return SOR_BUFFER_TOO_SMALL; /* :44 */
  size_t w = (size_t)varint_encode_u32((uint32_t)n, out);                  /* :49-50 */
  size_t rd = 0;
  while (rd < n) { /* :56-62 */
    size_t bs = n - rd < MAX_BLOCK_LEN ? n - rd : MAX_BLOCK_LEN;
    w += sor_encode_block_ex(in + rd, bs, out + w, flags);
    rd += bs;
  }
  *written = w;
  return SOR_OK;
}

int sor_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  return sor_compress_ex(in, n, out, cap, written, 0);
}

int sor_uncompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  *written = 0;
  uint64_t len;
  int hdr = varint_decode(in, n, 32, &len); /* snappy.nim:92-94 */
  if (hdr <= 0) return SOR_INVALID_INPUT;
  if ((uint64_t)cap < len) return SOR_BUFFER_TOO_SMALL; /* :96-97 */
  if (len == 0) return (size_t)hdr == n ? SOR_OK : SOR_INVALID_INPUT; /* :99-102 */
  size_t w;
  int st = sor_decode_all_tags(in + hdr, n - (size_t)hdr, out, cap, &w); /* :104-105 */
  if (st != SOR_OK) return st;
  if ((uint64_t)w != len) return SOR_INVALID_INPUT; /* :107-108 */
  *written = w;
  return SOR_OK;
}

size_t sor_encode_frame(const uint8_t* in, size_t n, uint8_t* out) { /* encoder.nim:385-426 */
  uint32_t crc = sor_masked_crc32c(in, n); /* :397-398 */
  out[4] = (uint8_t)crc;
  out[5] = (uint8_t)(crc >> 8);
  out[6] = (uint8_t)(crc >> 16);
  out[7] = (uint8_t)(crc >> 24);
  if (n >= MIN_NON_LITERAL) { /* :401 */
    uint8_t hdr[5];
    size_t hl = (size_t)varint_encode_u32((uint32_t)n, hdr);
    size_t bl = sor_encode_block(in, n, out + 8 + hl);
    if (bl <= n - n / 8) { /* :408 -- block length WITHOUT the varint */
      memcpy(out + 8, hdr, hl);
      size_t fl = hl + bl + 4;
      out[0] = 0x00;
      out[1] = (uint8_t)fl;
      out[2] = (uint8_t)(fl >> 8);
      out[3] = (uint8_t)(fl >> 16);
      return fl + 4;
    }
  }
  size_t fl = n + 4; /* :419-426 */
  out[0] = 0x01;
  out[1] = (uint8_t)fl;
  out[2] = (uint8_t)(fl >> 8);
  out[3] = (uint8_t)(fl >> 16);
  memcpy(out + 8, in, n);
  return fl + 4;
}

int sor_compress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  *written = 0;
  if ((uint64_t)cap < sor_max_compressed_len_framed((int64_t)n)) return SOR_BUFFER_TOO_SMALL;
  memcpy(out, FRAMING_HEADER, sizeof FRAMING_HEADER); /* snappy.nim:142 */
  size_t w = sizeof FRAMING_HEADER, rd = 0;
  while (rd < n) { /* :146-153 */
    size_t fs = n - rd < MAX_FRAME_DATA_LEN ? n - rd : MAX_FRAME_DATA_LEN;
    w += sor_encode_frame(in + rd, fs, out + w);
    rd += fs;
  }
  *written = w;
  return SOR_OK;
}

int sor_uncompress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                          int check_header, int check_integrity, size_t* read_out,
                          size_t* written_out) {
  size_t rd = 0, wr = 0;
  *read_out = 0;
  *written_out = 0;
  if (check_header) { /* snappy.nim:187-196 */
    if (n < sizeof FRAMING_HEADER) return SOR_INVALID_INPUT;
    if (memcmp(in, FRAMING_HEADER, sizeof FRAMING_HEADER) != 0) return SOR_INVALID_INPUT;
    rd = sizeof FRAMING_HEADER;
  }
  while (rd < n) { /* :199 */
    size_t remaining = n - rd;
    if (remaining < 4) return SOR_INVALID_INPUT; /* :200-201 */
    uint32_t hdr = load32(in + rd); /* codec.nim:166-172 */
    uint8_t id = (uint8_t)(hdr & 0xff);
    size_t data_len = hdr >> 8;
    rd += 4;
    if (remaining - 4 < data_len) return SOR_INVALID_INPUT; /* :206-207 */

    if (id == 0x00) { /* compressed chunk, :209-235 */
      if (data_len < 4) return SOR_INVALID_INPUT;
      uint32_t crc = load32(in + rd);
      size_t room = cap - wr;
      size_t max_out = room < MAX_FRAME_DATA_LEN ? room : MAX_FRAME_DATA_LEN; /* :215 */
      size_t got;
      int st = sor_uncompress(in + rd + 4, data_len - 4, out + wr, max_out, &got);
      if (st == SOR_BUFFER_TOO_SMALL) { /* :219-227 */
        uint64_t ul;
        if (sor_uncompressed_len(in + rd + 4, data_len - 4, &ul) != SOR_OK ||
            ul > MAX_FRAME_DATA_LEN)
          return SOR_INVALID_INPUT;
        *read_out = rd - 4; /* resume point = this chunk's header */
        *written_out = wr;
        return SOR_OK;
      }
      if (st != SOR_OK) return SOR_INVALID_INPUT; /* :228 */
      if (check_integrity && sor_masked_crc32c(out + wr, got) != crc) return SOR_CRC_MISMATCH;
      wr += got;
    } else if (id == 0x01) { /* uncompressed chunk, :237-257 */
      if (data_len < 4) return SOR_INVALID_INPUT;
      uint32_t crc = load32(in + rd);
      /* :244-246 -- CRC is verified BEFORE the size checks */
      if (check_integrity && sor_masked_crc32c(in + rd + 4, data_len - 4) != crc)
        return SOR_CRC_MISMATCH;
      size_t ul = data_len - 4;
      if (ul > MAX_FRAME_DATA_LEN) return SOR_INVALID_INPUT; /* :250-251 */
      if (ul > cap - wr) {                                   /* :253-254 */
        *read_out = rd - 4;
        *written_out = wr;
        return SOR_OK;
      }
      memcpy(out + wr, in + rd + 4, ul);
      wr += ul;
    } else if (id < 0x80) {
      return SOR_UNKNOWN_CHUNK; /* :259-260 */
    }
    /* 0x80..0xff: skipped without validation, :262-263 */
    rd += data_len;
  }
  *read_out = rd;
  *written_out = wr;
  return SOR_OK;
}

int sor_uncompressed_len_framed(const uint8_t* in, size_t n, uint64_t* len) { /* codec.nim:178-214 */
  size_t rd = 0;
  uint64_t expected = 0;
  while (rd < n) {
    size_t remaining = n - rd;
    if (remaining < 4) return SOR_INVALID_INPUT;
    uint32_t hdr = load32(in + rd);
    uint8_t id = (uint8_t)(hdr & 0xff);
    size_t data_len = hdr >> 8;
    if (remaining < data_len + 4) return SOR_INVALID_INPUT;
    rd += 4;
    uint64_t u;
    if (id == 0x00) {
      if (data_len < 4) return SOR_INVALID_INPUT;
      if (sor_uncompressed_len(in + rd + 4, data_len - 4, &u) != SOR_OK) return SOR_INVALID_INPUT;
    } else if (id == 0x01) {
      if (data_len < 4) return SOR_INVALID_INPUT;
      u = data_len - 4;
    } else if (id < 0x80) {
      return SOR_INVALID_INPUT;
    } else {
      u = 0;
    }
    if (u > MAX_FRAME_DATA_LEN) return SOR_INVALID_INPUT; /* :208-209 */
    expected += u;
    rd += data_len;
  }
  *len = expected;
  return SOR_OK;
}

/* ---- batch helpers for the CPU baseline -------------------------------------------------- */
void sor_compress_blocks(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                         size_t slot, uint32_t* sizes) {
  size_t nb = (total_len + block_len - 1) / block_len;
  for (size_t i = 0; i < nb; i++) {
    size_t off = i * block_len;
    size_t bl = total_len - off < block_len ? total_len - off : block_len;
    size_t w;
    sor_compress(in + off, bl, out + i * slot, slot, &w);
    sizes[i] = (uint32_t)w;
  }
}

int sor_uncompress_blocks(const uint8_t* in, const uint64_t* offsets, const uint32_t* sizes,
                          size_t n_blocks, uint8_t* out, size_t block_len) {
  for (size_t i = 0; i < n_blocks; i++) {
    size_t w;
    int st = sor_uncompress(in + offsets[i], sizes[i], out + i * block_len, block_len, &w);
    if (st != SOR_OK) return st;
  }
  return SOR_OK;
}

/* ---- the same two batch helpers over n_threads host threads (bench.py's all-cores leg, SURVEY 8d
 * "B2 ... all host cores"): blocks are independent, threads take runs of 8 blocks from a shared
 * counter.  Plain pthreads; the reference itself is single-threaded (tests/benchmark.nim:93-126). */
#include <pthread.h>
#include <stdatomic.h>

typedef struct {
  const uint8_t* in;
  size_t total_len, block_len, slot, n_blocks;
  uint8_t* out;
  uint32_t* sizes;
  const uint64_t* offsets;
  atomic_size_t next;
  atomic_int status;
  int decode; /* 0 compress (raw buffers), 1 uncompress, 2 encodeFrame (framed chunks) */
} sor_mt_job;

static void* sor_mt_worker(void* arg) {
  sor_mt_job* j = (sor_mt_job*)arg;
  const size_t run = 8;
  for (;;) {
    size_t b0 = atomic_fetch_add(&j->next, run);
    if (b0 >= j->n_blocks) break;
    size_t b1 = b0 + run < j->n_blocks ? b0 + run : j->n_blocks;
    for (size_t i = b0; i < b1; i++) {
      size_t w;
      if (j->decode == 1) {
        int st = sor_uncompress(j->in + j->offsets[i], j->sizes[i], j->out + i * j->block_len,
                                j->block_len, &w);
        if (st != SOR_OK) atomic_store(&j->status, st);
      } else {
        size_t off = i * j->block_len;
        size_t bl = j->total_len - off < j->block_len ? j->total_len - off : j->block_len;
        if (j->decode == 2)
          w = sor_encode_frame(j->in + off, bl, j->out + i * j->slot);
        else
          sor_compress(j->in + off, bl, j->out + i * j->slot, j->slot, &w);
        j->sizes[i] = (uint32_t)w;
      }
    }
  }
  return NULL;
}

static int sor_mt_run(sor_mt_job* j, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 1024) n_threads = 1024;
  pthread_t th[1024];
  int started = 0;
  atomic_init(&j->next, 0);
  atomic_init(&j->status, SOR_OK);
  for (int t = 1; t < n_threads; t++) {
    if (pthread_create(&th[started], NULL, sor_mt_worker, j) != 0) break;
    started++;
  }
  sor_mt_worker(j);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  return atomic_load(&j->status);
}

void sor_compress_blocks_mt(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                            size_t slot, uint32_t* sizes, int n_threads) {
  sor_mt_job j;
  memset(&j, 0, sizeof j);
  j.in = in, j.total_len = total_len, j.block_len = block_len, j.out = out, j.slot = slot;
  j.sizes = sizes, j.n_blocks = (total_len + block_len - 1) / block_len, j.decode = 0;
  (void)sor_mt_run(&j, n_threads);
}

int sor_uncompress_blocks_mt(const uint8_t* in, const uint64_t* offsets, const uint32_t* sizes,
                             size_t n_blocks, uint8_t* out, size_t block_len, int n_threads) {
  sor_mt_job j;
  memset(&j, 0, sizeof j);
  j.in = in, j.offsets = offsets, j.sizes = (uint32_t*)sizes, j.n_blocks = n_blocks, j.out = out;
  j.block_len = block_len, j.decode = 1;
  return sor_mt_run(&j, n_threads);
}

void sor_encode_frames_mt(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                          size_t slot, uint32_t* sizes, int n_threads) {
  sor_mt_job j;
  memset(&j, 0, sizeof j);
  j.in = in, j.total_len = total_len, j.block_len = block_len, j.out = out, j.slot = slot;
  j.sizes = sizes, j.n_blocks = (total_len + block_len - 1) / block_len, j.decode = 2;
  (void)sor_mt_run(&j, n_threads);
}
