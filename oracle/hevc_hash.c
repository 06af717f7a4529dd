/* oracle/hevc_hash.c -- see hevc_hash.h.  MD5 as RFC 1321 states it; the picture hashes as H.265 D.3.19 (8-bit samples:
 * pictureData holds one byte per sample, rows of the component one after the other). */
#include "hevc_hash.h"
#include <string.h>

static const uint32_t K[64] = {
  0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be,
  0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
  0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
  0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
  0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
  0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
static const uint8_t S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                               4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };

static void md5_block(orc_md5 *m, const uint8_t *p)
{
  uint32_t w[16], a = m->a, b = m->b, c = m->c, d = m->d;
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  for (int i = 0; i < 64; i++) {
    uint32_t f; int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    f += a + K[i] + w[g];
    a = d; d = c; c = b;
    b += (f << S[i]) | (f >> (32 - S[i]));
  }
  m->a += a; m->b += b; m->c += c; m->d += d;
}
void orc_md5_init(orc_md5 *m) { m->a = 0x67452301; m->b = 0xefcdab89; m->c = 0x98badcfe; m->d = 0x10325476; m->nbytes = 0; m->fill = 0; }
void orc_md5_update(orc_md5 *m, const uint8_t *p, size_t n)
{
  m->nbytes += n;
  while (n) {
    if (m->fill == 0 && n >= 64) { md5_block(m, p); p += 64; n -= 64; continue; }
    size_t k = 64 - (size_t)m->fill; if (k > n) k = n;
    memcpy(m->buf + m->fill, p, k); m->fill += (int)k; p += k; n -= k;
    if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
  }
}
void orc_md5_final(orc_md5 *m, uint8_t out[16])
{
  uint64_t bits = m->nbytes * 8;
  uint8_t pad[72]; size_t n = (size_t)((m->fill < 56) ? 56 - m->fill : 120 - m->fill);
  memset(pad, 0, sizeof(pad)); pad[0] = 0x80;
  for (int i = 0; i < 8; i++) pad[n + (size_t)i] = (uint8_t)(bits >> (8 * i));
  orc_md5_update(m, pad, n + 8);
  const uint32_t v[4] = { m->a, m->b, m->c, m->d };
  for (int i = 0; i < 16; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
}

void orc_picture_hash(int hash_type, const pixel *const plane[3], const int stride[3], int w, int h, uint8_t out[3][16])
{
  for (int c = 0; c < 3; c++) {
    const int pw = c ? w / 2 : w, ph = c ? h / 2 : h;
    memset(out[c], 0, 16);
    if (hash_type == 0) {
      orc_md5 m; orc_md5_init(&m);
      for (int y = 0; y < ph; y++) orc_md5_update(&m, plane[c] + (size_t)y * stride[c], (size_t)pw);
      orc_md5_final(&m, out[c]);
    } else if (hash_type == 1) {
      /* D.3.19: crc = 0xFFFF; every bit of every sample, most significant first: crcMsb = crc >> 15, crc = (crc << 1 | bit) & 0xFFFF,
       * crc ^= 0x1021 when crcMsb; then sixteen zero bits the same way */
      uint32_t crc = 0xFFFF;
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
        const int v = plane[c][(size_t)y * stride[c] + x];
        for (int bit = 7; bit >= 0; bit--) { const uint32_t msb = (crc >> 15) & 1; crc = ((crc << 1) | (uint32_t)((v >> bit) & 1)) & 0xFFFF; if (msb) crc ^= 0x1021; }
      }
      for (int bit = 0; bit < 16; bit++) { const uint32_t msb = (crc >> 15) & 1; crc = (crc << 1) & 0xFFFF; if (msb) crc ^= 0x1021; }
      out[c][0] = (uint8_t)(crc >> 8); out[c][1] = (uint8_t)crc;
    } else {
      /* D.3.19: sum += (sample & 0xFF) ^ xorMask, xorMask = (x & 0xFF) ^ (y & 0xFF) ^ (x >> 8) ^ (y >> 8); 32 bits */
      uint32_t sum = 0;
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++)
        sum += (uint32_t)((plane[c][(size_t)y * stride[c] + x] & 0xFF) ^ ((x & 0xFF) ^ (y & 0xFF) ^ (x >> 8) ^ (y >> 8)));
      out[c][0] = (uint8_t)(sum >> 24); out[c][1] = (uint8_t)(sum >> 16); out[c][2] = (uint8_t)(sum >> 8); out[c][3] = (uint8_t)sum;
    }
  }
}
