/* oracle/synth.c -- see synth.h.  Formulas: SURVEY.md section 8(d). */
#include "synth.h"
#include <stddef.h>
static inline uint32_t fmix32(uint32_t h)
{
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}
void orc_synth_frame(int kind, uint32_t seed, int w, int h, int t, uint8_t *out)
{
  uint8_t *Y = out, *U = out + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
  int cw = w / 2, ch = h / 2;
  if (kind == 1) { for (size_t i = 0; i < (size_t)w * h * 3 / 2; i++) out[i] = 128; return; }
  if (kind == 2) {
    for (size_t i = 0; i < (size_t)w * h * 3 / 2; i++) out[i] = (uint8_t)(fmix32(seed ^ ((uint32_t)t * 0x9E3779B1u) ^ ((uint32_t)i * 0x85EBCA77u)) & 255);
    return;
  }
  int S = h / 8;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = 32 + (x * 160) / w + (y * 32) / h;
      for (int k = 0; k < 8; k++) {
        int cx = (k * w / 8 + 5 * (k + 1) * t) % w, cy = (k * h / 8 + 3 * (k + 1) * t) % h;
        int dx = x - cx, dy = y - cy;
        if (dx >= 0 && dx < S && dy >= 0 && dy < S) v = 64 + 16 * k + (((dx * 7) ^ (dy * 13)) & 63);
      }
      uint32_t hsh = fmix32(seed ^ ((uint32_t)t * 0x9E3779B1u) ^ ((uint32_t)(y * w + x) * 0x85EBCA77u));
      v += (int)(hsh & 7) - 3;
      Y[(size_t)y * w + x] = (uint8_t)(v < 16 ? 16 : (v > 235 ? 235 : v));
    }
  for (int y = 0; y < ch; y++)
    for (int x = 0; x < cw; x++) {
      int u = 96 + (x * 64) / cw, v = 96 + (y * 64) / ch;
      for (int k = 0; k < 8; k++) {
        int cx = ((k * w / 8 + 5 * (k + 1) * t) % w) / 2, cy = ((k * h / 8 + 3 * (k + 1) * t) % h) / 2;
        int dx = x - cx, dy = y - cy;
        if (dx >= 0 && dx < S / 2 && dy >= 0 && dy < S / 2) u = 96 + (x * 64) / cw + 8 * k;
      }
      U[(size_t)y * cw + x] = (uint8_t)u; V[(size_t)y * cw + x] = (uint8_t)v;
    }
}
