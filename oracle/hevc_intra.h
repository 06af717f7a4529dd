/* oracle/hevc_intra.h -- intra sample prediction, H.265 8.4.4.2.  Test infrastructure. */
#ifndef ORC_HEVC_INTRA_H
#define ORC_HEVC_INTRA_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif

/* z-scan order address of the 4x4 block containing luma sample (x,y): CTB raster address in
 * the high bits, bit-interleaved offset inside the CTB in the low bits (H.265 6.5.2). */
static inline uint32_t orc_zaddr(int x, int y, int ctb_log2, int pic_w_ctbs)
{
  uint32_t ctb = (uint32_t)((y >> ctb_log2) * pic_w_ctbs + (x >> ctb_log2));
  uint32_t xi = (uint32_t)(x & ((1 << ctb_log2) - 1)) >> 2, yi = (uint32_t)(y & ((1 << ctb_log2) - 1)) >> 2;
  uint32_t z = 0;
  for (int b = 0; b < ctb_log2 - 2; b++) z |= ((xi >> b) & 1u) << (2 * b) | ((yi >> b) & 1u) << (2 * b + 1);
  return (ctb << (2 * (ctb_log2 - 2))) | z;
}

typedef struct {
  int pic_w, pic_h;          /* luma samples (coded size) */
  int ctb_log2, pic_w_ctbs;
  /* optional: per-4x4 (luma units) map, nonzero = usable for intra reference (slice/tile/
   * constrained-intra restrictions); NULL = only z-order and picture bounds apply */
  const uint8_t *usable4; int usable_stride;
  /* optional per-CTB maps (raster): slice address and tile id of the slice/tile containing each
   * CTB (-1 = not decoded yet).  A neighbour in a different slice or tile is unavailable. */
  const int32_t *ctb_slice; const int16_t *ctb_tile;
} orc_avail_ctx;

/* H.265 6.4.1: is luma location (xn,yn) available when decoding the block at (xc,yc)? */
int orc_available(const orc_avail_ctx *a, int xc, int yc, int xn, int yn);

/* Build the 4n+1 reference samples for a n x n block of plane component (cidx 0 luma, else
 * chroma 4:2:0) at (x0,y0) in component samples, with substitution (8.4.4.2.2).
 * left[0] = p[-1][-1], left[1+i] = p[-1][i] (i < 2n); top[0] = p[-1][-1], top[1+i] = p[i][-1]. */
void orc_intra_refs(const orc_avail_ctx *a, const pixel *plane, int stride, int cidx,
                    int x0, int y0, int n, pixel *left, pixel *top);
/* 8.4.4.2.3 filtering decision + filter (in place on copies), then 8.4.4.2.4-6 prediction. */
void orc_intra_predict(const pixel *left, const pixel *top, int n, int cidx, int mode,
                       int strong_enabled, pixel *pred, int pred_stride);
#ifdef __cplusplus
}
#endif
#endif
