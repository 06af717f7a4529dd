/* oracle/hevc_intra.c -- H.265 8.4.4.2 restated.  Test infrastructure. */
#include "hevc_intra.h"

int orc_available(const orc_avail_ctx *a, int xc, int yc, int xn, int yn)
{
  if (xn < 0 || yn < 0 || xn >= a->pic_w || yn >= a->pic_h) return 0;
  if (orc_zaddr(xn, yn, a->ctb_log2, a->pic_w_ctbs) > orc_zaddr(xc, yc, a->ctb_log2, a->pic_w_ctbs)) return 0;
  if (a->usable4 && !a->usable4[(yn >> 2) * a->usable_stride + (xn >> 2)]) return 0;
  if (a->ctb_slice || a->ctb_tile) {
    int cn = (yn >> a->ctb_log2) * a->pic_w_ctbs + (xn >> a->ctb_log2);
    int cc = (yc >> a->ctb_log2) * a->pic_w_ctbs + (xc >> a->ctb_log2);
    if (a->ctb_slice && a->ctb_slice[cn] != a->ctb_slice[cc]) return 0;
    if (a->ctb_tile && a->ctb_tile[cn] != a->ctb_tile[cc]) return 0;
  }
  return 1;
}

void orc_intra_refs(const orc_avail_ctx *a, const pixel *plane, int stride, int cidx,
                    int x0, int y0, int n, pixel *left, pixel *top)
{
  /* order of 8.4.4.2.2: from p[-1][2n-1] up to p[-1][-1], then p[0][-1] .. p[2n-1][-1] */
  int total = 4 * n + 1;
  pixel val[129]; uint8_t av[129];
  int sh = cidx ? 1 : 0;
  int xc = x0 << sh, yc = y0 << sh;
  int any = 0;
  for (int i = 0; i < total; i++) {
    int x, y;
    if (i < 2 * n) { x = x0 - 1; y = y0 + 2 * n - 1 - i; }
    else if (i == 2 * n) { x = x0 - 1; y = y0 - 1; }
    else { x = x0 + (i - 2 * n - 1); y = y0 - 1; }
    av[i] = (uint8_t)orc_available(a, xc, yc, x << sh, y << sh);
    if (av[i]) { val[i] = plane[y * stride + x]; any = 1; }
  }
  if (!any) {
    for (int i = 0; i < total; i++) val[i] = 128;
  } else {
    if (!av[0]) {
      int j = 1;
      while (!av[j]) j++;
      val[0] = val[j];
    }
    for (int i = 1; i < total; i++) if (!av[i]) val[i] = val[i - 1];
  }
  left[0] = top[0] = val[2 * n];
  for (int i = 0; i < 2 * n; i++) { left[1 + i] = val[2 * n - 1 - i]; top[1 + i] = val[2 * n + 1 + i]; }
}

void orc_intra_predict(const pixel *left_in, const pixel *top_in, int n, int cidx, int mode,
                       int strong_enabled, pixel *pred, int ps)
{
  pixel lf[65], tp[65];
  const pixel *left = left_in, *top = top_in;
  int l2 = orc_log2((unsigned)n);
  /* 8.4.4.2.3 filtering process of neighbouring samples (luma only for 4:2:0) */
  if (cidx == 0 && mode != 1 && n != 4) {
    int d1 = orc_abs(mode - 26), d2 = orc_abs(mode - 10);
    int mind = ORC_MIN(d1, d2);
    int thr = (n == 8) ? 7 : (n == 16) ? 1 : 0;
    if (mind > thr) {
      int corner = left_in[0];
      int bi = strong_enabled && n == 32 &&
               orc_abs(corner + top_in[2 * n] - 2 * top_in[n]) < (1 << (8 - 5)) &&
               orc_abs(corner + left_in[2 * n] - 2 * left_in[n]) < (1 << (8 - 5));
      if (bi) {
        lf[0] = tp[0] = (pixel)corner;
        for (int i = 0; i < 63; i++) {
          lf[1 + i] = (pixel)(((63 - i) * corner + (i + 1) * left_in[64] + 32) >> 6);
          tp[1 + i] = (pixel)(((63 - i) * corner + (i + 1) * top_in[64] + 32) >> 6);
        }
        lf[64] = left_in[64]; tp[64] = top_in[64];
      } else {
        lf[0] = tp[0] = (pixel)((left_in[1] + 2 * corner + top_in[1] + 2) >> 2);
        for (int i = 1; i < 2 * n; i++) {
          lf[i] = (pixel)((left_in[i + 1] + 2 * left_in[i] + left_in[i - 1] + 2) >> 2);
          tp[i] = (pixel)((top_in[i + 1] + 2 * top_in[i] + top_in[i - 1] + 2) >> 2);
        }
        lf[2 * n] = left_in[2 * n]; tp[2 * n] = top_in[2 * n];
      }
      left = lf; top = tp;
    }
  }
#define L(i) left[1 + (i)]   /* p[-1][i] */
#define T(i) top[1 + (i)]    /* p[i][-1] */
  if (mode == 0) {           /* 8.4.4.2.4 planar */
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
        pred[y * ps + x] = (pixel)(((n - 1 - x) * L(y) + (x + 1) * T(n) + (n - 1 - y) * T(x) + (y + 1) * L(n) + n) >> (l2 + 1));
  } else if (mode == 1) {    /* 8.4.4.2.5 DC */
    int sum = n;
    for (int i = 0; i < n; i++) sum += L(i) + T(i);
    int dc = sum >> (l2 + 1);
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) pred[y * ps + x] = (pixel)dc;
    if (cidx == 0 && n < 32) {
      pred[0] = (pixel)((L(0) + 2 * dc + T(0) + 2) >> 2);
      for (int x = 1; x < n; x++) pred[x] = (pixel)((T(x) + 3 * dc + 2) >> 2);
      for (int y = 1; y < n; y++) pred[y * ps] = (pixel)((L(y) + 3 * dc + 2) >> 2);
    }
  } else {                   /* 8.4.4.2.6 angular */
    int angle = orc_intra_angle[mode];
    pixel refbuf[3 * 32 + 4]; pixel *ref = refbuf + 32 + 1;   /* ref[-n .. 2n] */
    int last = (n * angle) >> 5;
    if (mode >= 18) {
      for (int x = 0; x <= n; x++) ref[x] = top[x];            /* ref[x] = p[-1+x][-1] */
      if (angle < 0) {
        if (last < -1) {
          int inv = orc_inv_angle[mode];
          for (int x = last; x <= -1; x++) ref[x] = left[((x * inv + 128) >> 8)];  /* p[-1][-1+((x*inv+128)>>8)] */
        }
      } else {
        for (int x = n + 1; x <= 2 * n; x++) ref[x] = top[x];
      }
      for (int y = 0; y < n; y++) {
        int idx = ((y + 1) * angle) >> 5, fact = ((y + 1) * angle) & 31;
        for (int x = 0; x < n; x++)
          pred[y * ps + x] = fact ? (pixel)(((32 - fact) * ref[x + idx + 1] + fact * ref[x + idx + 2] + 16) >> 5)
                                  : ref[x + idx + 1];
      }
      if (mode == 26 && cidx == 0 && n < 32)
        for (int y = 0; y < n; y++) pred[y * ps] = (pixel)orc_clip_pixel(T(0) + ((L(y) - left[0]) >> 1));
    } else {
      for (int x = 0; x <= n; x++) ref[x] = left[x];           /* ref[x] = p[-1][-1+x] */
      if (angle < 0) {
        if (last < -1) {
          int inv = orc_inv_angle[mode];
          for (int x = last; x <= -1; x++) ref[x] = top[((x * inv + 128) >> 8)];   /* p[-1+((x*inv+128)>>8)][-1] */
        }
      } else {
        for (int x = n + 1; x <= 2 * n; x++) ref[x] = left[x];
      }
      for (int x = 0; x < n; x++) {
        int idx = ((x + 1) * angle) >> 5, fact = ((x + 1) * angle) & 31;
        for (int y = 0; y < n; y++)
          pred[y * ps + x] = fact ? (pixel)(((32 - fact) * ref[y + idx + 1] + fact * ref[y + idx + 2] + 16) >> 5)
                                  : ref[y + idx + 1];
      }
      if (mode == 10 && cidx == 0 && n < 32)
        for (int x = 0; x < n; x++) pred[x] = (pixel)orc_clip_pixel(L(0) + ((T(x) - left[0]) >> 1));
    }
  }
#undef L
#undef T
}
