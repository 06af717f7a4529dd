/* oracle/hevc_sao.h -- sample adaptive offset: the normative picture process (H.265 8.7.3), the CTU syntax
 * (7.3.8.3, 9.3) and the encoder-side decision "uvgx SAO decision v1".  Test infrastructure.
 * Kvazaar's `sao` option (off at the ultrafast preset uvgComm picks for camera video, on for the slower presets:
 * /root/reference/src/ui/settings/defaultsettings.cpp:287-324) and OpenHEVC's decode of such streams
 * (/root/reference/src/media/processing/openhevcfilter.cpp:145-146). */
#ifndef ORC_HEVC_SAO_H
#define ORC_HEVC_SAO_H
#include "hevc_common.h"
#include "hevc_cabac.h"
#ifdef __cplusplus
extern "C" {
#endif

/* parameters of one CTU.  type: 0 off, 1 band, 2 edge (Cr shares Cb's type and class); offset[c][k] = SaoOffsetVal[k + 1] */
typedef struct {
  uint8_t type[3], eo_class[3], band_pos[3];
  int8_t offset[3][4];
} orc_sao_params;

/* the picture the filter runs on */
typedef struct {
  int w, h;                      /* luma size */
  int ctb_log2, pic_w_ctbs;
  const pixel *src[3];           /* deblocked picture (read only) */
  pixel *dst[3];                 /* output picture */
  int stride[3];                 /* both pictures */
  const orc_sao_params *params;  /* per CTB, raster */
  /* samples of another slice / tile are used only when filtering across is enabled (NULL arrays: one slice, one tile) */
  const int32_t *ctb_slice; const int16_t *ctb_tile;
  int across_slices, across_tiles;
  /* ... or, when not NULL, per CTB the neighbouring CTBs whose samples the in-loop filters may use (orc_lf_neighbour_bit: 8 bits), which replaces the four fields
   * above: slice_loop_filter_across_slices_enabled_flag is a property of each slice, and of two slices the LATER one's flag decides (7.4.7.1) */
  const uint8_t *ctb_nb;
  const uint8_t *no_filter; int nf_stride;   /* per 4x4 luma block: pcm / transquant-bypass samples stay untouched; may be NULL */
} orc_sao_ctx;

/* bit of the neighbouring CTB at (dx, dy), each -1 .. 1 and not both 0, in a ctb_nb entry: NW N NE W E SW S SE = 0 .. 7 */
static inline int orc_lf_neighbour_bit(int dx, int dy) { const int k = (dy + 1) * 3 + (dx + 1); return k > 4 ? k - 1 : k; }
/* 8.7.3: every CTB of the picture */
void orc_sao_picture(const orc_sao_ctx *s);
/* 8.7.3.2 edgeIdx for a sample and its two neighbours (0: none, 1..4) */
static inline int orc_sao_edge_idx(int c, int a, int b)
{
  int e = 2 + ((c > a) - (c < a)) + ((c > b) - (c < b));
  return e == 2 ? 0 : (e < 2 ? e + 1 : e);
}
extern const int8_t orc_sao_eo_dx[4][2], orc_sao_eo_dy[4][2];

/* 7.3.8.3 for the CTU (rx, ry): `left` / `up` are the neighbouring CTUs' parameters when they may be merged from, else NULL */
void orc_sao_write(orc_cabac_enc *c, const orc_sao_params *p, const orc_sao_params *left, const orc_sao_params *up, int luma, int chroma);
void orc_sao_parse(orc_cabac_dec *c, orc_sao_params *p, const orc_sao_params *left, const orc_sao_params *up, int luma, int chroma);

/* "uvgx SAO decision v1" for one CTU: statistics of the deblocked picture against the source, see hevc_sao.c.
 * plane sizes: luma w x h, chroma half; (cx, cy) the CTU (64 x 64 luma samples); lambda_q4 = orc_lambda_q4[qp]. */
void orc_sao_decide_ctu(const pixel *const deb[3], const pixel *const org[3], const int stride[3], int w, int h,
                        int cx, int cy, int lambda_q4, orc_sao_params *out);
#ifdef __cplusplus
}
#endif
#endif
