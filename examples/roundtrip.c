/* roundtrip.c -- a plain C consumer of the C ABI (include/snappy_hip.h), the way a Nim `importc`
 * binding or a cgo stub would use it: compress / uncompress, the framed pair and the masked CRC on
 * host buffers.  Build:  cc examples/roundtrip.c -Iinclude -Lnim-snappy_amd -lsnappy_hip \
 *                           -Wl,-rpath,$PWD/nim-snappy_amd -o roundtrip
 * Exit code 0 = every round trip restored the input; 2 = no usable GPU (DEVICE_ERROR), which is the
 * documented behaviour of a box without one (there is no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "snappy_hip.h"

#define FAIL(what, st) do { printf("FAILED: %s (status %d)\n", what, (int)(st)); return 1; } while (0)

int main(void) {
  const size_t n = 300000;
  uint8_t* src = malloc(n);
  for (size_t i = 0; i < n; i++) src[i] = (uint8_t)("the quick brown fox jumps over the lazy dog "[i % 44] + (i / 7000));
  size_t cap = (size_t)snappy_hip_max_compressed_len((uint32_t)n);
  size_t fcap = (size_t)snappy_hip_max_compressed_len_framed((int64_t)n);
  uint8_t* comp = malloc(cap > fcap ? cap : fcap);
  uint8_t* back = malloc(n);
  size_t w = 0, w2 = 0, r = 0;

  int st = snappy_hip_compress(src, n, comp, cap, &w);
  if (st == SNAPPY_HIP_DEVICE_ERROR) {
    printf("no usable GPU: %s\n", snappy_hip_last_error());
    return 2;
  }
  if (st != SNAPPY_HIP_OK) FAIL("compress", st);
  uint64_t ulen = 0;
  if (snappy_hip_uncompressed_len(comp, w, &ulen) != SNAPPY_HIP_OK || ulen != n) FAIL("uncompressed_len", ulen);
  st = snappy_hip_uncompress(comp, w, back, n, &w2);
  if (st != SNAPPY_HIP_OK || w2 != n || memcmp(src, back, n)) FAIL("uncompress", st);
  printf("raw:    %zu -> %zu bytes, round trip ok\n", n, w);
  /* too small an output buffer: CodecError.bufferTooSmall, snappy.nim:96-97 */
  st = snappy_hip_uncompress(comp, w, back, n - 1, &w2);
  if (st != SNAPPY_HIP_BUFFER_TOO_SMALL) FAIL("uncompress into a short buffer", st);

  st = snappy_hip_compress_framed(src, n, comp, fcap, &w);
  if (st != SNAPPY_HIP_OK) FAIL("compress_framed", st);
  memset(back, 0, n);
  st = snappy_hip_uncompress_framed(comp, w, back, n, 1, 1, &r, &w2);
  if (st != SNAPPY_HIP_OK || r != w || w2 != n || memcmp(src, back, n)) FAIL("uncompress_framed", st);
  printf("framed: %zu -> %zu bytes, round trip ok\n", n, w);
  comp[w - 1] ^= 1; /* damage the last chunk: FrameError.crcMismatch or invalidInput */
  st = snappy_hip_uncompress_framed(comp, w, back, n, 1, 1, &r, &w2);
  if (st != SNAPPY_HIP_CRC_MISMATCH && st != SNAPPY_HIP_INVALID_INPUT) FAIL("damaged framed stream", st);

  int cst = -1;
  uint32_t crc = snappy_hip_masked_crc32c((const uint8_t*)"123456789", 9, &cst);
  /* CRC-32C("123456789") = 0xe3069283; masked: rotr(crc, 15) + 0xa282ead8 (codec.nim:71-75) */
  const uint32_t c = 0xe3069283u, want = ((c >> 15) | (c << 17)) + 0xa282ead8u;
  if (cst != SNAPPY_HIP_OK || crc != want) FAIL("masked crc32c", crc);
  printf("masked crc32c ok (%08x)\n", crc);
  free(src);
  free(comp);
  free(back);
  return 0;
}
