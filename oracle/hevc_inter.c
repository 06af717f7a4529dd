/* oracle/hevc_inter.c -- see hevc_inter.h.  Test infrastructure. */
#include "hevc_inter.h"

static inline int refpix(const pixel *ref, int stride, int w, int h, int x, int y)
{
  x = orc_clip3(0, w - 1, x); y = orc_clip3(0, h - 1, y);
  return ref[y * stride + x];
}

/* shift1 = Min(4, BitDepth - 8) = 0, shift2 = 6, shift3 = Max(2, 14 - BitDepth) = 6 */
void orc_mc_luma(const pixel *ref, int stride, int pw, int ph, int x0, int y0, int w, int h,
                 int mvx, int mvy, int16_t *dst, int ds)
{
  int xf = mvx & 3, yf = mvy & 3;
  int xi = x0 + (mvx >> 2), yi = y0 + (mvy >> 2);
  const int8_t *fx = orc_luma_filter[xf], *fy = orc_luma_filter[yf];
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v;
      if (xf == 0 && yf == 0) {
        v = refpix(ref, stride, pw, ph, xi + x, yi + y) << 6;
      } else if (yf == 0) {
        v = 0;
        for (int i = 0; i < 8; i++) v += fx[i] * refpix(ref, stride, pw, ph, xi + x + i - 3, yi + y);
      } else if (xf == 0) {
        v = 0;
        for (int i = 0; i < 8; i++) v += fy[i] * refpix(ref, stride, pw, ph, xi + x, yi + y + i - 3);
      } else {
        int t[8];
        for (int j = 0; j < 8; j++) {
          t[j] = 0;
          for (int i = 0; i < 8; i++) t[j] += fx[i] * refpix(ref, stride, pw, ph, xi + x + i - 3, yi + y + j - 3);
        }
        v = 0;
        for (int j = 0; j < 8; j++) v += fy[j] * t[j];
        v >>= 6;
      }
      dst[y * ds + x] = (int16_t)v;
    }
}

void orc_mc_chroma(const pixel *ref, int stride, int pw, int ph, int x0, int y0, int w, int h,
                   int mvx, int mvy, int16_t *dst, int ds)
{
  int xf = mvx & 7, yf = mvy & 7;
  int xi = x0 + (mvx >> 3), yi = y0 + (mvy >> 3);
  const int8_t *fx = orc_chroma_filter[xf], *fy = orc_chroma_filter[yf];
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v;
      if (xf == 0 && yf == 0) {
        v = refpix(ref, stride, pw, ph, xi + x, yi + y) << 6;
      } else if (yf == 0) {
        v = 0;
        for (int i = 0; i < 4; i++) v += fx[i] * refpix(ref, stride, pw, ph, xi + x + i - 1, yi + y);
      } else if (xf == 0) {
        v = 0;
        for (int i = 0; i < 4; i++) v += fy[i] * refpix(ref, stride, pw, ph, xi + x, yi + y + i - 1);
      } else {
        int t[4];
        for (int j = 0; j < 4; j++) {
          t[j] = 0;
          for (int i = 0; i < 4; i++) t[j] += fx[i] * refpix(ref, stride, pw, ph, xi + x + i - 1, yi + y + j - 1);
        }
        v = 0;
        for (int j = 0; j < 4; j++) v += fy[j] * t[j];
        v >>= 6;
      }
      dst[y * ds + x] = (int16_t)v;
    }
}

void orc_pred_uni(const int16_t *src, int ss, pixel *dst, int ds, int w, int h)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) dst[y * ds + x] = (pixel)orc_clip_pixel((src[y * ss + x] + 32) >> 6);
}
void orc_pred_bi(const int16_t *a, const int16_t *b, int ss, pixel *dst, int ds, int w, int h)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) dst[y * ds + x] = (pixel)orc_clip_pixel((a[y * ss + x] + b[y * ss + x] + 64) >> 7);
}
