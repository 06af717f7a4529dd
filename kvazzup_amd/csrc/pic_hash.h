// kvazzup_amd/csrc/pic_hash.h -- decoded picture hash (H.265 D.2.19 / D.3.19, SEI payload type 132) on the host: the MD5 (RFC 1321) or the
// checksum of each colour component of a decoded picture.  Kvazaar's "hash" option (kvz_config.hash, which uvgComm sets to KVZ_HASH_NONE,
// kvazaarfilter.cpp:289) and OpenHEVC's libOpenHevcSetCheckMD5.  Not a hot path: a stream that carries the hash verifies itself in any decoder.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace kvzx {

class Md5 {
 public:
  Md5() { s_[0] = 0x67452301u; s_[1] = 0xefcdab89u; s_[2] = 0x98badcfeu; s_[3] = 0x10325476u; }
  void update(const uint8_t *p, size_t n)
  {
    total_ += n;
    if (fill_) {
      size_t k = 64 - fill_; if (k > n) k = n;
      memcpy(buf_ + fill_, p, k); fill_ += k; p += k; n -= k;
      if (fill_ < 64) return;
      block(buf_); fill_ = 0;
    }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { memcpy(buf_, p, n); fill_ = n; }
  }
  void final(uint8_t out[16])
  {
    const uint64_t bits = total_ * 8;
    static const uint8_t pad[64] = {0x80};
    update(pad, (fill_ < 56 ? 56 : 120) - fill_);
    uint8_t len[8]; for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    for (int i = 0; i < 16; i++) out[i] = (uint8_t)(s_[i >> 2] >> (8 * (i & 3)));
  }
 private:
  static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
  void block(const uint8_t *p)
  {
    // the sine table computed once (floor(2^32 * |sin(i + 1)|) needs a libm; the constants are RFC 1321's)
    static const uint32_t T[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122,
      0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6,
      0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60,
      0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039,
      0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
    uint32_t a = s_[0], b = s_[1], c = s_[2], d = s_[3];
    for (int i = 0; i < 64; i++) {
      const int round = i >> 4;
      uint32_t f; int k;
      switch (round) {
        case 0: f = d ^ (b & (c ^ d)); k = i; break;
        case 1: f = c ^ (d & (b ^ c)); k = 5 * i + 1; break;
        case 2: f = b ^ c ^ d; k = 3 * i + 5; break;
        default: f = c ^ (b | ~d); k = 7 * i; break;
      }
      const uint32_t t = d; d = c; c = b;
      b = b + rotl(a + f + T[i] + x[k & 15], R[round][i & 3]);
      a = t;
    }
    s_[0] += a; s_[1] += b; s_[2] += c; s_[3] += d;
  }
  uint32_t s_[4]; uint64_t total_ = 0; uint8_t buf_[64]; size_t fill_ = 0;
};

// hash_type as in the SEI: 0 = MD5 (16 bytes per component), 2 = checksum (4 bytes).  planes: the decoded picture, w x h luma samples.
// Returns the SEI payload (hash_type byte + the three hashes).
inline std::vector<uint8_t> picture_hash_payload(int hash_type, const uint8_t *const plane[3], const size_t pitch[3], int w, int h)
{
  std::vector<uint8_t> out;
  out.push_back((uint8_t)hash_type);
  for (int c = 0; c < 3; c++) {
    const int pw = c ? w / 2 : w, ph = c ? h / 2 : h;
    if (hash_type == 0) {
      Md5 m;
      for (int y = 0; y < ph; y++) m.update(plane[c] + (size_t)y * pitch[c], (size_t)pw);
      uint8_t d[16]; m.final(d);
      out.insert(out.end(), d, d + 16);
    } else {
      uint32_t sum = 0;                           // D.3.19: every sample xor-ed with a mask made of its coordinates
      for (int y = 0; y < ph; y++) {
        const uint8_t *row = plane[c] + (size_t)y * pitch[c];
        const uint32_t my = (uint32_t)((y & 0xff) ^ (y >> 8));
        for (int x = 0; x < pw; x++) sum += (uint32_t)row[x] ^ (my ^ (uint32_t)((x & 0xff) ^ (x >> 8)));
      }
      for (int i = 3; i >= 0; i--) out.push_back((uint8_t)(sum >> (8 * i)));
    }
  }
  return out;
}

}  // namespace kvzx
