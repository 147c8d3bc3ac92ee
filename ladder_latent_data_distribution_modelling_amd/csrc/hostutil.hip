// Host-side helpers of the C ABI (no device code): CRC-32C for the TF checkpoint-v2 ("tensor bundle") reader/writer.
#include "ladder_hip.h"

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace {

struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);   // reflected Castagnoli polynomial
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
  }
};

const Crc32cTables& tables() {
  static const Crc32cTables tb;
  return tb;
}

}  // namespace

extern "C" uint32_t ladder_crc32c_extend(uint32_t crc, const void* data, size_t n) {
  const Crc32cTables& tb = tables();
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = crc ^ 0xFFFFFFFFu;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7u) != 0) {
    c = tb.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {                                                 // slicing-by-8, little-endian host
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = tb.t[7][w & 0xFF] ^ tb.t[6][(w >> 8) & 0xFF] ^ tb.t[5][(w >> 16) & 0xFF] ^ tb.t[4][(w >> 24) & 0xFF] ^
        tb.t[3][(w >> 32) & 0xFF] ^ tb.t[2][(w >> 40) & 0xFF] ^ tb.t[1][(w >> 48) & 0xFF] ^ tb.t[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n > 0) {
    c = tb.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    --n;
  }
  return c ^ 0xFFFFFFFFu;
}
