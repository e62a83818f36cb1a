/*
 * snappy_hip.h -- C ABI of the MI355X-native Snappy block / framed codec.
 *
 * Drop-in boundary for status-im/nim-snappy's in-memory API: every entry point below is what
 * a Nim `importc, cdecl` binding for that path would bind (the reference's own FFI precedent
 * is `masked_crc32c`, snappy/codec.nim:66-69, and tests/cpp_snappy.nim:6-11).  Plain pointers
 * and sizes only.  INTEGRATION.md shows the Nim shim.
 *
 * All codec work (block encode, block decode, CRC32C) runs in hand-written HIP kernels for
 * gfx950.  There is NO CPU fallback: without a usable GPU every codec call returns
 * SNAPPY_HIP_DEVICE_ERROR (and snappy_hip_last_error() says why).  Only the scalar format
 * helpers that the reference also computes on the host (size bounds, varint length readers,
 * the frame-header pre-scan) run on the CPU.
 *
 * Status codes: 0 = ok, otherwise 1 + the ordinal of the reference's error enums
 * (CodecError / FrameError, snappy/codec.nim:55-64).
 *
 * Thread safety: like the reference (all `func`, no globals) every call is re-entrant.  The
 * host-buffer calls take a context from a process-wide pool (one per concurrent call; created on
 * demand on the GPU SNAPPY_HIP_DEVICE names, default 0) and run side by side; large inputs go in
 * batches on up to three worker threads so that upload, kernels and download overlap.  An explicit
 * context (snappy_hip_ctx_create) serves one thread at a time: it owns scratch buffers that its
 * calls reuse.  Its calls may name different streams: a call on another stream than the call
 * before it first waits -- on the device, through an event -- for that call's work, so the two
 * never share the scratch (they run one after the other; for work side by side use a context
 * each).  Every entry point makes its context's device current for the call and restores the
 * caller's.
 *
 * Environment (read once, by the host-buffer calls only; the device-resident API reads one, at context creation):
 *   SNAPPY_HIP_ENC_GWAVES    "g" or "g,min_blocks" (every context): of four encoder workgroups, g (0..4, default 4) run a
 *                            second wave whose hash table lies in global memory (csrc/encode_kernel.h) -- eight blocks a
 *                            CU instead of four -- on batches of at least min_blocks blocks (default: twice what the
 *                            GPU's LDS-table waves take at once, but no more than a host-buffer call's batch,
 *                            SNAPPY_HIP_HOST_BATCH).  Output bytes never depend on it.
 *   SNAPPY_HIP_DEVICE        GPU of the pooled contexts (default 0)
 *   SNAPPY_HIP_HOST_BATCH    blocks per upload/compute/download batch (64 .. 65536, default 2048)
 *   SNAPPY_HIP_PIN_HOST      0 pageable copies (default) / 1, 2 page-lock the caller's buffers per call, per batch
 *                            (hipHostRegister on the caller's memory) / 3, 4 the contexts' page-locked staging rings
 *   SNAPPY_HIP_COPY_THREADS  host threads of the staging rings' copies (modes 3, 4)
 */
#ifndef SNAPPY_HIP_H
#define SNAPPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SNAPPY_HIP_OK = 0,
  SNAPPY_HIP_BUFFER_TOO_SMALL = 1, /* CodecError.bufferTooSmall / FrameError.bufferTooSmall */
  SNAPPY_HIP_INVALID_INPUT = 2,    /* CodecError.invalidInput   / FrameError.invalidInput   */
  SNAPPY_HIP_CRC_MISMATCH = 3,     /* FrameError.crcMismatch                                */
  SNAPPY_HIP_UNKNOWN_CHUNK = 4,    /* FrameError.unknownChunk                               */
  SNAPPY_HIP_DEVICE_ERROR = 100    /* no GPU / HIP failure: never silently falls back       */
};

/* ---- host-side scalar helpers (the reference computes these on the host too) ------------- */

/* maxCompressedLen, snappy/codec.nim:92-120: 32 + n + n/6. */
uint64_t snappy_hip_max_compressed_len(uint32_t n);
/* maxCompressedLenFramed, snappy/codec.nim:140-164. */
uint64_t snappy_hip_max_compressed_len_framed(int64_t n);
/* uncompressedLen, snappy/codec.nim:129-138 (u64 varint). 0 ok / 2 invalid. */
int snappy_hip_uncompressed_len(const uint8_t* in, size_t n, uint64_t* len);
/* uncompressedLenFramed, snappy/codec.nim:178-214. 0 ok / 2 invalid. */
int snappy_hip_uncompressed_len_framed(const uint8_t* in, size_t n, uint64_t* len);

/* ---- host-buffer codec API: same contract as snappy.nim, pointers are HOST memory --------- */

/* compress, snappy.nim:27-64.  cap must be >= snappy_hip_max_compressed_len(n). */
int snappy_hip_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);
/* uncompress, snappy.nim:84-110. */
int snappy_hip_uncompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);
/* compressFramed, snappy.nim:130-155.  cap >= snappy_hip_max_compressed_len_framed(n). */
int snappy_hip_compress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                               size_t* written);
/* uncompressFramed, snappy.nim:169-267.  Returns both counters; when the output fills, the
 * call returns ok with *read at the header of the first chunk that did not fit, so it can be
 * resumed with check_header = 0 (tests/test_framed.nim:38-59). */
int snappy_hip_uncompress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                 int check_header, int check_integrity, size_t* read,
                                 size_t* written);
/* maskedCrc, snappy/codec.nim:71-75 -> masked_crc32c, snappy/crc32c.c:759-763.
 * *status (may be NULL) receives SNAPPY_HIP_OK or SNAPPY_HIP_DEVICE_ERROR. */
uint32_t snappy_hip_masked_crc32c(const uint8_t* buf, size_t n, int* status);
/* encodeBlock, snappy/encoder.nim:184-383 (what snappy/faststreams.nim:44,51 and
 * snappy/streams.nim:38 call).  1 <= n <= 65536, cap >= max_compressed_len(n). */
int snappy_hip_encode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                            size_t* written);
/* encodeFrame, snappy/encoder.nim:385-426 (snappy/faststreams.nim:73,80). */
int snappy_hip_encode_frame(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                            size_t* written);
/* decodeAllTags, snappy/decoder.nim:20-155. */
int snappy_hip_decode_all_tags(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                               size_t* written);

/* ---- device-resident batch API (what the GPU needs; the reference has no counterpart) ------
 * Memory the kernels touch: they WRITE exactly the bytes of the ranges documented as outputs.  They
 * READ input ranges with 16-byte loads at 16-byte-ALIGNED addresses, i.e. up to 15 bytes in front of a
 * unit's first byte and behind its last byte, inside the aligned 16-byte lines that hold those two
 * bytes.  Such a line always holds a valid byte and never crosses a page, so a unit may lie flush
 * against either end of an allocation of exactly its size (tests/test_gpu_batch.py places units so);
 * the extra bytes are never interpreted.  No padding is required of the caller.
 * All `d_` pointers are DEVICE memory of the context's GPU.  `stream` is a hipStream_t (NULL =
 * the context's own stream).  Calls enqueue work and return; use snappy_hip_ctx_sync() or
 * your own stream synchronisation before reading results.  Block i of a batch is independent:
 * batches shard across GPUs by block range with no collective. */
typedef struct snappy_hip_ctx snappy_hip_ctx;

/* The host-buffer calls above run on pooled contexts (created on demand, one per concurrent call; an idle
 * context keeps its grow-only device workspace, and at most four idle ones are kept).  This frees the idle
 * ones now -- e.g. after one very large call.  Calls in flight keep theirs. */
void snappy_hip_release_pool(void);

int snappy_hip_ctx_create(snappy_hip_ctx** ctx, int device);
void snappy_hip_ctx_destroy(snappy_hip_ctx* ctx);
int snappy_hip_ctx_sync(snappy_hip_ctx* ctx, void* stream);
/* Human-readable text of the last SNAPPY_HIP_DEVICE_ERROR on this thread. */
const char* snappy_hip_last_error(void);

/* What one compressed unit is. */
enum {
  SNAPPY_HIP_UNIT_BODY = 0,  /* bare tag stream: encodeBlock / decodeAllTags                  */
  SNAPPY_HIP_UNIT_RAW = 1,   /* varint(len) + tag stream: one compress()/uncompress() buffer  */
  SNAPPY_HIP_UNIT_FRAME = 2  /* framed chunk: type, len24, masked crc, body (encodeFrame)     */
};

/* Encode ceil(total_len / block_len) slices of d_in (slice i = d_in[i*block_len ...], the
 * last may be short; block_len <= 65536), slice i into d_slots + i*slot_stride, its size
 * into d_sizes[i].  slot_stride >= max_compressed_len(block_len) + 8. */
int snappy_hip_encode_blocks_d(snappy_hip_ctx* ctx, const uint8_t* d_in, uint64_t total_len,
                               uint32_t block_len, int unit, uint8_t* d_slots,
                               uint32_t slot_stride, uint32_t* d_sizes, void* stream);
/* Second pass of the length-then-data scheme: exclusive scan of d_sizes into
 * d_offsets[0..n_blocks] (d_offsets[0] = base) and gather of the slots into one contiguous
 * stream d_out[base ...]; the end of the stream is d_offsets[n_blocks]. */
int snappy_hip_pack_d(snappy_hip_ctx* ctx, const uint8_t* d_slots, uint32_t slot_stride,
                      const uint32_t* d_sizes, uint64_t n_blocks, uint64_t base, uint8_t* d_out,
                      uint64_t* d_offsets, void* stream);
/* Decode n_units independent units: unit i = d_in[d_in_off[i] ..+ d_in_len[i]] into
 * d_out[d_out_off[i] ..+ d_out_cap[i]]; bytes written into d_out_len[i], status into
 * d_status[i] (SNAPPY_HIP_* codes).  unit = BODY or RAW.  Units that decode to at most 65536
 * bytes take the block-parallel kernel; larger RAW units take the serial whole-stream kernel.
 * d_crc (may be NULL) receives the masked CRC32C of each unit's output. */
int snappy_hip_decode_blocks_d(snappy_hip_ctx* ctx, const uint8_t* d_in, const uint64_t* d_in_off,
                               const uint32_t* d_in_len, uint64_t n_units, int unit,
                               uint8_t* d_out, const uint64_t* d_out_off,
                               const uint32_t* d_out_cap, uint32_t* d_out_len,
                               uint32_t* d_status, uint32_t* d_crc, void* stream);
/* Masked CRC32C of n_units byte ranges d_in[d_off[i] ..+ d_len[i]] into d_crc[i]. */
int snappy_hip_crc32c_d(snappy_hip_ctx* ctx, const uint8_t* d_in, const uint64_t* d_off,
                        const uint32_t* d_len, uint64_t n_units, uint32_t* d_crc, void* stream);
/* uncompress, snappy.nim:84-110, of ONE raw buffer RESIDENT IN HBM (the varint and all) into d_out.
 * A buffer of several 64 KiB blocks is split on the device and its blocks are decoded in parallel (on
 * `stream` when one is given); streams whose elements straddle block boundaries (foreign encoders) take
 * the serial kernel.  *written is host memory; returns when done. */
int snappy_hip_uncompress_d(snappy_hip_ctx* ctx, const uint8_t* d_in, uint64_t n, uint8_t* d_out,
                            uint64_t cap, uint64_t* written, void* stream);
/* compressFramed, snappy.nim:130-155, for an input RESIDENT IN HBM: stream identifier + one chunk
 * per 65536-byte slice (encodeFrame, snappy/encoder.nim:385-426) packed into d_out[0 .. *written).
 * cap >= snappy_hip_max_compressed_len_framed(n), else SNAPPY_HIP_BUFFER_TOO_SMALL.  Returns when
 * the stream is complete (*written is host memory). */
int snappy_hip_compress_framed_d(snappy_hip_ctx* ctx, const uint8_t* d_in, uint64_t n, uint8_t* d_out,
                                 uint64_t cap, uint64_t* written, void* stream);
/* uncompressFramed, snappy.nim:169-267, for a stream RESIDENT IN HBM: the chunk walk
 * (snappy.nim:199-265), the block decode, the CRC comparison (:231-233, :244-246) and the choice of
 * the first failing chunk in stream order all run on the device; the decoded bytes stay in d_out.
 * Same contract as snappy_hip_uncompress_framed, resume protocol included: status, and on ok the
 * two counters (*read, *written: host memory).  Returns when the verdict is known. */
int snappy_hip_uncompress_framed_d(snappy_hip_ctx* ctx, const uint8_t* d_in, uint64_t n, uint8_t* d_out,
                                   uint64_t cap, int check_header, int check_integrity,
                                   uint64_t* read, uint64_t* written, void* stream);
/* Block-range sharding over n contexts (one per GPU) from ONE process, with the host-side
 * concatenate: shard k = d_in[k][0 .. in_len[k]) (whole 64 KiB blocks, except the last shard's tail),
 * resident on ctxs[k]'s GPU, is encoded and packed there; the n shard totals are scanned on the host
 * (the reference's serial `written += ...`, snappy.nim:56-62, :149-153) and every shard is downloaded
 * to its offset in the ONE host buffer `out` (page-locked memory lets the downloads run side by
 * side).  framed != 0: compressFramed of the concatenated input; 0: compress (total < 2^32).
 * shard_off (may be NULL): the n + 1 scanned offsets.  The result is byte-identical to one call
 * over the concatenated input. */
int snappy_hip_compress_shards(snappy_hip_ctx* const* ctxs, int n, const uint8_t* const* d_in,
                               const uint64_t* in_len, int framed, uint8_t* out, uint64_t cap,
                               uint64_t* written, uint64_t* shard_off);
/* The same in STAGES of stage_blocks 64 KiB blocks per context, so that a GPU encodes while its earlier output
 * travels (the reference's serial loop over slices, snappy.nim:56-62, :146-153, cut across the GPUs instead of along
 * them): with S = stage_blocks, the global block range [(j n + k) S, (j n + k + 1) S) is stage j of context k, and
 * d_in[k] holds context k's stages back to back (in_len[k] must be exactly what that layout gives context k of the
 * total; the total's tail may leave the last stage short).  A context encodes + packs a stage, publishes its size,
 * downloads -- on a second stream -- every earlier stage of its own whose place in `out` is known by then (the sum of
 * all sizes in front of it: n integers per stage through host memory, the path's one exchange), and goes on with the
 * next stage.  The downloads overlap the kernels when `out` is page-locked.  stage_blocks = 0: one stage, i.e.
 * snappy_hip_compress_shards.  shard_off (may be NULL): n + 1 values, where each context's first stage lies and the
 * stream's end.  The result is byte-identical to one call over the blocks in global order. */
int snappy_hip_compress_shards_staged(snappy_hip_ctx* const* ctxs, int n, const uint8_t* const* d_in,
                                      const uint64_t* in_len, uint64_t stage_blocks, int framed, uint8_t* out,
                                      uint64_t cap, uint64_t* written, uint64_t* shard_off);
/* Batches of 512 units or more are launched in a sorted order (decode: by compressed length, longest first; encode:
 * by a sketch of the block), which costs three small launches and is worth ~10 % on mixed batches; results never
 * depend on it.  enable = 0 launches in the caller's order (default 1). */
int snappy_hip_ctx_launch_order(snappy_hip_ctx* ctx, int enable);
/* Average duration in milliseconds of the last timed kernel launches, measured with HIP
 * events on the launch stream (bench.py's roofline leg).  which: 0 block decode (the indexed
 * decode kernel, or the one-pass kernel when units carry per-unit kinds), 1 encode, 2 crc,
 * 3 pack, 4 decode index pass, 5 whole-stream decode pass, 6 framed chunk walk, 7 raw-buffer split rounds, 8 the
 * block decode's second launch (units the ring-window instantiation passes on); 9 and 10 are counts, not durations:
 * 10 the bytes (stream + output; *launches: how many) of units of few, long elements, which the index pass decodes itself,
 * since the context was created; 9 turns
 * of the indexed decoder that were given up on after a bounded wait (0 on consistent input; such a unit is decoded by
 * the one-pass kernel) since the context was created.  Timing is recorded only between
 * snappy_hip_ctx_timing(ctx, 1) and (ctx, 0). */
int snappy_hip_ctx_timing(snappy_hip_ctx* ctx, int enable);
double snappy_hip_ctx_kernel_ms(snappy_hip_ctx* ctx, int which, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
