// roundtrip.cpp -- the header-only C++ mirror (nim-snappy_amd/snappy_hip.hpp) in use: the same calls
// a user of the reference makes (snappy.encode / decode / encodeFramed / decodeFramed), vectors in
// and out.  Exit code 0 = all round trips restored the input, 2 = no usable GPU.
#include <cstdio>
#include <string>
#include <vector>

#include "../nim-snappy_amd/snappy_hip.hpp"

int main() {
  std::vector<uint8_t> src;
  const std::string words[] = {"alpha ", "beta ", "gamma ", "delta ", "epsilon "};
  for (int i = 0; src.size() < 250000; i++) {
    const std::string& w = words[(i * 7 + i / 13) % 5];
    src.insert(src.end(), w.begin(), w.end());
  }
  const std::vector<uint8_t> enc = snappy::encode(src);
  if (enc.empty()) {
    std::printf("no usable GPU (or encode failed): %s\n", snappy_hip_last_error());
    return 2;
  }
  if (snappy::decode(enc) != src) return 1;
  std::printf("raw:    %zu -> %zu bytes, round trip ok\n", src.size(), enc.size());
  const std::vector<uint8_t> fenc = snappy::encodeFramed(src);
  if (fenc.empty() || snappy::decodeFramed(fenc) != src) return 1;
  std::printf("framed: %zu -> %zu bytes, round trip ok\n", src.size(), fenc.size());
  if (!snappy::decode(std::vector<uint8_t>(enc.begin(), enc.end() - 1)).empty()) return 1;  // truncated: empty
  if (snappy::maskedCrc(src.data(), 0) != 0xa282ead8u) return 1;  // masked CRC of nothing: rotr(0,15) + delta
  return 0;
}
