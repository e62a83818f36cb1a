// snappy_hip.hpp -- header-only C++ mirror of nim-snappy's in-memory API (snappy.nim) over the
// C ABI of include/snappy_hip.h.  Same names, argument meaning and error behaviour as the
// reference: encode/decode return an empty vector on failure (snappy.nim:66-82, 112-128),
// compress/uncompress work on caller buffers and return a status (snappy.nim:27-64, 84-110).
#pragma once

#include <cstddef>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "../include/snappy_hip.h"

namespace snappy {

enum class CodecError { bufferTooSmall = 0, invalidInput = 1 };                        // codec.nim:55-58
enum class FrameError { bufferTooSmall = 0, invalidInput, crcMismatch, unknownChunk };  // :60-64

constexpr uint64_t maxUncompressedLen = 0xffffffffull;  // codec.nim:10

inline uint64_t maxCompressedLen(uint32_t n) { return snappy_hip_max_compressed_len(n); }
inline uint64_t maxCompressedLenFramed(int64_t n) { return snappy_hip_max_compressed_len_framed(n); }
inline uint32_t maskedCrc(const uint8_t* p, size_t n) { return snappy_hip_masked_crc32c(p, n, nullptr); }

// snappy.nim:27 -- status 0 = ok, else 1 + ord(CodecError); 100 = no device
inline int compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  return snappy_hip_compress(in, n, out, cap, written);
}
// snappy.nim:84
inline int uncompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written) {
  return snappy_hip_uncompress(in, n, out, cap, written);
}

// snappy.nim:66
inline std::vector<uint8_t> encode(const std::vector<uint8_t>& input) {
  if (input.size() > maxUncompressedLen) return {};
  std::vector<uint8_t> out(maxCompressedLen((uint32_t)input.size()));
  size_t w = 0;
  if (compress(input.data(), input.size(), out.data(), out.size(), &w) != SNAPPY_HIP_OK) return {};
  out.resize(w);
  return out;
}

// snappy.nim:112
inline std::vector<uint8_t> decode(const std::vector<uint8_t>& input,
                                   uint64_t maxSize = maxUncompressedLen) {
  uint64_t n = 0;
  if (snappy_hip_uncompressed_len(input.data(), input.size(), &n) != SNAPPY_HIP_OK) return {};
  if (n > maxSize) return {};
  std::vector<uint8_t> out(n);
  size_t w = 0;
  if (uncompress(input.data(), input.size(), out.data(), out.size(), &w) != SNAPPY_HIP_OK) return {};
  return out;
}

// snappy.nim:157
inline std::vector<uint8_t> encodeFramed(const std::vector<uint8_t>& input) {
  std::vector<uint8_t> out(maxCompressedLenFramed((int64_t)input.size()));
  size_t w = 0;
  if (snappy_hip_compress_framed(input.data(), input.size(), out.data(), out.size(), &w) !=
      SNAPPY_HIP_OK)
    return {};
  out.resize(w);
  return out;
}

// snappy.nim:169 -- (status, read, written)
struct FramedResult {
  int status;
  size_t read, written;
};
inline FramedResult uncompressFramed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                     bool checkHeader = true, bool checkIntegrity = true) {
  FramedResult r{0, 0, 0};
  r.status = snappy_hip_uncompress_framed(in, n, out, cap, checkHeader, checkIntegrity, &r.read,
                                          &r.written);
  return r;
}

// snappy.nim:269
inline std::vector<uint8_t> decodeFramed(const std::vector<uint8_t>& input,
                                         uint64_t maxSize = std::numeric_limits<int64_t>::max(),
                                         bool checkIntegrity = true) {
  uint64_t n = 0;
  if (snappy_hip_uncompressed_len_framed(input.data(), input.size(), &n) != SNAPPY_HIP_OK) return {};
  if (n > maxSize) return {};
  std::vector<uint8_t> out(n);
  FramedResult r = uncompressFramed(input.data(), input.size(), out.data(), out.size(), true,
                                    checkIntegrity);
  if (r.status != SNAPPY_HIP_OK) return {};
  return out;
}

}  // namespace snappy
