/* oracle/hevc_inter.h -- fractional sample interpolation + default weighted prediction,
 * H.265 8.5.3.3.3 / 8.5.3.3.4.2.  8-bit 4:2:0.  Test infrastructure. */
#ifndef ORC_HEVC_INTER_H
#define ORC_HEVC_INTER_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* 14-bit intermediate prediction of a w x h luma block at (x0,y0) displaced by quarter-sample
 * mv; reference picture samples are clamped to the picture (8.5.3.3.3.1). */
void orc_mc_luma(const pixel *ref, int stride, int pic_w, int pic_h, int x0, int y0, int w, int h,
                 int mvx, int mvy, int16_t *dst, int dst_stride);
/* chroma plane block (w,h in chroma samples, x0,y0 chroma), mv in 1/8 chroma-sample units
 * (= the luma mv for 4:2:0) */
void orc_mc_chroma(const pixel *ref, int stride, int pic_w, int pic_h, int x0, int y0, int w, int h,
                   int mvx, int mvy, int16_t *dst, int dst_stride);
/* 8.5.3.3.4.2 default weighted sample prediction */
void orc_pred_uni(const int16_t *src, int ss, pixel *dst, int ds, int w, int h);
void orc_pred_bi(const int16_t *a, const int16_t *b, int ss, pixel *dst, int ds, int w, int h);
#ifdef __cplusplus
}
#endif
#endif
