/*
 * snappy_oracle.h -- CPU restatement of the nim-snappy hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the HIP codec: a plain-C, single-threaded restatement of
 * the reference's block encoder, block decoder, framing layer and masked CRC32C.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (nim-snappy_amd/csrc) never links, loads or calls anything in this directory.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * status-im/nim-snappy checkout).  Pinning: see oracle/README.md -- golden vectors of the
 * reference's own tests (tests/test_snappy.nim, tests/test_framed.nim), the reference's
 * compiled crc32c.c (oracle/_ref) and libsnappy 1.1.8 triangulation for the encoder.
 *
 * Status codes are shared with include/snappy_hip.h:
 *   0 ok, 1 bufferTooSmall, 2 invalidInput, 3 crcMismatch, 4 unknownChunk
 * (= 1 + ordinal of CodecError / FrameError, codec.nim:55-64).
 */
#ifndef SNAPPY_ORACLE_H
#define SNAPPY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SOR_OK = 0,
  SOR_BUFFER_TOO_SMALL = 1,
  SOR_INVALID_INPUT = 2,
  SOR_CRC_MISMATCH = 3,
  SOR_UNKNOWN_CHUNK = 4
};

/* encoder flags: 0 = nim-snappy behaviour.  The two bits switch the oracle to the two
 * places where google/snappy 1.1.8 differs, so that the SAME main loop can be
 * checked byte-for-byte against libsnappy in the build container (tools/gen_golden.py). */
#define SOR_ENC_CPP_GE_LIMIT 1u  /* after a copy stop on ip >= ipLimit (C++), not ip > ipLimit */
#define SOR_ENC_CPP_SHIFT 2u     /* hash shift = 32 - log2(tableSize) (C++), not >>18 & mask  */

/* codec.nim:92-120 */
uint64_t sor_max_compressed_len(uint32_t n);
/* codec.nim:140-164 */
uint64_t sor_max_compressed_len_framed(int64_t n);
/* codec.nim:129-138 (u64 varint); returns SOR_OK / SOR_INVALID_INPUT */
int sor_uncompressed_len(const uint8_t* in, size_t n, uint64_t* len);
/* codec.nim:178-214 */
int sor_uncompressed_len_framed(const uint8_t* in, size_t n, uint64_t* len);
/* codec.nim:71-75 -> crc32c.c:759-763 */
uint32_t sor_masked_crc32c(const uint8_t* buf, size_t n);
/* plain CRC-32C (init/xorout 0xffffffff), crc32c.c:748 wrapped as in :761 */
uint32_t sor_crc32c(const uint8_t* buf, size_t n);

/* encoder.nim:184-383; 1 <= n <= 65536; out must hold sor_max_compressed_len(n) bytes.
 * Returns bytes written (block body, no length header). */
size_t sor_encode_block(const uint8_t* in, size_t n, uint8_t* out);
size_t sor_encode_block_ex(const uint8_t* in, size_t n, uint8_t* out, unsigned flags);
/* encoder.nim:385-426; returns bytes written (chunk header + crc + body). */
size_t sor_encode_frame(const uint8_t* in, size_t n, uint8_t* out);
/* decoder.nim:20-155 */
int sor_decode_all_tags(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);

/* snappy.nim:27-64 */
int sor_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);
int sor_compress_ex(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written,
                    unsigned flags);
/* snappy.nim:84-110 */
int sor_uncompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);
/* snappy.nim:130-155 */
int sor_compress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);
/* snappy.nim:169-267 */
int sor_uncompress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                          int check_header, int check_integrity, size_t* read,
                          size_t* written);

/* Batch helpers for the CPU baseline (bench.py cpu_baseline leg): n_blocks independent raw
 * Snappy buffers (own varint each), block i uncompressed at in + i*block_len (last may be
 * short via total_len), compressed into out + i*slot with its size in sizes[i]. */
void sor_compress_blocks(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                         size_t slot, uint32_t* sizes);
int sor_uncompress_blocks(const uint8_t* in, const uint64_t* offsets, const uint32_t* sizes,
                          size_t n_blocks, uint8_t* out, size_t block_len);

/* The same over n_threads host threads (bench.py's all-cores leg; blocks are independent). */
void sor_compress_blocks_mt(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                            size_t slot, uint32_t* sizes, int n_threads);
/* encodeFrame (encoder.nim:385-426) of every block: framed chunks, slot >= 8 + max_compressed_len(block_len) */
void sor_encode_frames_mt(const uint8_t* in, size_t total_len, size_t block_len, uint8_t* out,
                          size_t slot, uint32_t* sizes, int n_threads);
int sor_uncompress_blocks_mt(const uint8_t* in, const uint64_t* offsets, const uint32_t* sizes,
                             size_t n_blocks, uint8_t* out, size_t block_len, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
