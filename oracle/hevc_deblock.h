/* oracle/hevc_deblock.h -- in-loop deblocking filter, H.265 8.7.2.  8-bit 4:2:0.
 * Test infrastructure. */
#ifndef ORC_HEVC_DEBLOCK_H
#define ORC_HEVC_DEBLOCK_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
  int w, h;                    /* coded luma size */
  pixel *plane[3]; int stride[3];
  /* boundary strength per 4-sample edge segment on the 8x8 luma grid:
   * bs_v[(y>>2) * bs_stride + (x>>3)] : vertical edge at luma x (x % 8 == 0), rows y..y+3
   * bs_h[(y>>3) * bs_stride_h + (x>>2)] : horizontal edge at luma y (y % 8 == 0), cols x..x+3 */
  const uint8_t *bs_v; int bs_stride_v;
  const uint8_t *bs_h; int bs_stride_h;
  const int8_t *qp_y; int qp_stride;        /* QpY per 4x4 luma block */
  const uint8_t *no_filter; int nf_stride;  /* per 4x4: pcm+loop-filter-disabled or cu_transquant_bypass (may be NULL) */
  int beta_offset_div2, tc_offset_div2;     /* slice-level (constant over the picture in this oracle) */
  int cb_qp_offset, cr_qp_offset;           /* pps_cb_qp_offset / pps_cr_qp_offset */
} orc_deblock_ctx;

void orc_deblock_picture(const orc_deblock_ctx *d);   /* all vertical edges, then all horizontal edges */
void orc_deblock_vertical_edges(const orc_deblock_ctx *d);
void orc_deblock_horizontal_edges(const orc_deblock_ctx *d);
#ifdef __cplusplus
}
#endif
#endif
