/* oracle/hevc_enc.c -- see hevc_enc.h.  Test infrastructure: the CPU statement of the encoder
 * algorithm the HIP path implements (encode half of the hot path,
 * /root/reference/src/media/processing/kvazaarfilter.cpp:374-484). */
#include "hevc_enc.h"
#include "hevc_hash.h"
#include "hevc_bits.h"
#include "hevc_cabac.h"
#include "hevc_ps.h"
#include "hevc_pic.h"
#include "hevc_intra.h"
#include "hevc_inter.h"
#include "hevc_transform.h"
#include "hevc_deblock.h"
#include "hevc_mvpred.h"
#include "hevc_sao.h"

#define INTRA_P_GATE 24       /* intra-in-P: a quarter is a candidate when its inter cost exceeds this many lambda_q4 */
#define INTRA_P_BITS 16       /* ... and goes intra when the intra cost plus this many bins is below the inter cost */
#define SPLIT_BITS 8          /* rate charged for splitting a CU one level, in bins */
#define ME_PAD 64             /* padding of the reference copy used by the motion search */

const uint16_t orc_lambda_q4[52] = {
  3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 17, 19, 22, 24, 27, 30, 34, 38, 43, 48, 54,
  61, 68, 77, 86, 97, 108, 122, 137, 153, 172, 193, 217, 244, 273, 307, 344, 387, 434, 487, 547,
  614, 689, 773, 868, 974, 1093 };

struct orc_encoder {
  orc_enc_config cfg;
  uint8_t sfac[4][6][1024];            /* scaling-list default: the default lists' scaling factors per size and matrix */
  int cw, ch, b8w, b8h;
  int frame_idx, poc, intra_count;
  int qp;                              /* QP of the picture being coded (== cfg.qp without rate control) */
  int64_t rc_debt; uint32_t rc_bytes[8];  /* rate control: bits spent above target so far; sizes of the last access units */
  /* rate control v2 (rc_bands): bits per unit of level cost (Q8, smoothed), whether it has been measured yet, the level cost of the
   * last eight pictures and whether they were P pictures coded in groups */
  uint32_t rc_ratio_q8; int rc_ratio_valid; uint32_t rc_cost[8]; uint8_t rc_cost_valid[8];
  orc_vps vps; orc_sps sps; orc_pps pps;
  orc_pic pics[2]; orc_pic *cur, *ref;
  pixel *src[3];
  pixel *prev_src;                     /* cfg.me_source: the luma plane of the previous input picture (padded to the coded size), NULL until the option is set */
  int16_t *coef[3];
  pixel *predeblock[3];
  pixel *refpad; int refpad_stride;
  uint8_t *cu_log2, *cu_intra, *cu_flags, *cu_merge_idx, *cu_mvp_idx, *cu_intra_mode, *cu_cbf;
  int16_t *cu_mv, *cu_mvd;
  uint8_t *bs_v, *bs_h;
  /* intra analysis scratch: best mode and cost for each 8x8 / 16x16 / 32x32 block */
  uint8_t *im8, *im16, *im32; uint32_t *ic8, *ic16, *ic32;
  orc_bitw au;
  orc_avail_ctx av;
  int16_t *ctb_tile;                   /* tile id of every CTB (raster); NULL without tiles */
  int roi_w, roi_h; int8_t *roi;       /* delta-QP map (orc_enc_set_roi) */
  int8_t *ctu_qt, *ctu_qy, *ctu_delta; uint8_t *ctu_first;   /* per CTU: target QP, actual QpY, coded CuQpDeltaVal, z index (8x8 units) of the first CU with residual (64: none) */
  int *tile_row_bd;                    /* first CTB row of tile row i, i = 0 .. tile_rows */
  int tile_col_bd[34];                 /* first CTB column of tile column j, j = 0 .. tile_cols */
  orc_sao_params *sao; pixel *sao_in[3];   /* cfg.sao: parameters of every CTU; copy of the deblocked picture */
  int is_intra;
  int intra_p_ready;                   /* intra-in-P: the source-based intra analysis of this picture has been run (lazily, by the first candidate block) */
  uint64_t bins;
};

void orc_enc_default_config(orc_enc_config *c)
{
  memset(c, 0, sizeof(*c));
  c->qp = 32; c->intra_period = 64; c->vps_period = 1; c->search_range = 16;
  c->fps_num = 30; c->fps_den = 1; c->wpp = 1; c->deblock = 1; c->tile_rows = 1; c->me_early = 1; c->satd = 1; c->intra_chain = 1;
}

int orc_mvd_bits(int q)
{
  int a = q < 0 ? -q : q;
  if (a == 0) return 1;
  if (a == 1) return 3;
  int x = a - 2, k = 1, len = 0;
  while (x >= (1 << k)) { x -= 1 << k; k++; len++; }
  return 2 + len + 1 + k + 1;
}

static int level_for(int w, int h)
{
  long px = (long)w * h;
  if (px <= 2228224) return 123;      /* 4.1 */
  if (px <= 8912896) return 153;      /* 5.1 */
  return 183;                          /* 6.1 */
}

orc_encoder *orc_enc_open(const orc_enc_config *c)
{
  if (c->width < 16 || c->height < 16 || (c->width & 1) || (c->height & 1) || c->qp < 0 || c->qp > 51 ||
      c->search_range < 0 || c->search_range > 32 || c->tile_rows < 1 || c->tile_rows > (c->height + 63) / 64 || c->tile_cols > 31 || c->tile_cols > (c->width + 63) / 64) return NULL;
  orc_encoder *e = (orc_encoder *)calloc(1, sizeof(*e));
  orc_tables_init();
  e->cfg = *c;
  if (e->cfg.tile_cols < 1) e->cfg.tile_cols = 1;
  if (e->cfg.vaq > 0) e->cfg.qp_in_cu = 1;                 /* the deltas travel as cu_qp_delta */
  if (e->cfg.bitrate <= 0) e->cfg.rc_bands = 0;
  if ((e->cfg.slices == 1 && !e->cfg.wpp) || (e->cfg.slices == 2 && e->cfg.tile_rows * e->cfg.tile_cols < 2) || (e->cfg.slices == 1 && e->cfg.tile_cols > 1) || e->cfg.slices < 0 || e->cfg.slices > 2) e->cfg.slices = 0;
  if (e->cfg.rc_bands > 0) e->cfg.qp_in_cu = 1;
  e->qp = c->qp;
  e->cw = (c->width + 63) & ~63; e->ch = (c->height + 63) & ~63;
  if (e->cw < 128) e->cw = 128;                 /* WPP context hand-over needs two CTUs per row */
  e->b8w = e->cw / 8; e->b8h = e->ch / 8;
  size_t nb8 = (size_t)e->b8w * e->b8h, npx = (size_t)e->cw * e->ch;
  for (int i = 0; i < 2; i++) if (orc_pic_alloc(&e->pics[i], e->cw, e->ch)) return NULL;
  e->cur = &e->pics[0]; e->ref = &e->pics[1];
  for (int i = 0; i < 3; i++) {
    size_t n = i ? npx / 4 : npx;
    e->src[i] = (pixel *)malloc(n); e->coef[i] = (int16_t *)calloc(n, sizeof(int16_t)); e->predeblock[i] = (pixel *)malloc(n);
  }
  e->refpad_stride = e->cw + 2 * ME_PAD;
  e->refpad = (pixel *)malloc((size_t)e->refpad_stride * (e->ch + 2 * ME_PAD));
  e->cu_log2 = (uint8_t *)calloc(nb8, 1); e->cu_intra = (uint8_t *)calloc(nb8, 1); e->cu_flags = (uint8_t *)calloc(nb8, 1);
  e->cu_merge_idx = (uint8_t *)calloc(nb8, 1); e->cu_mvp_idx = (uint8_t *)calloc(nb8, 1); e->cu_intra_mode = (uint8_t *)calloc(nb8, 1);
  e->cu_cbf = (uint8_t *)calloc(nb8, 1); e->cu_mv = (int16_t *)calloc(nb8 * 2, sizeof(int16_t)); e->cu_mvd = (int16_t *)calloc(nb8 * 2, sizeof(int16_t));
  e->bs_v = (uint8_t *)malloc((size_t)(e->cw / 8) * (e->ch / 4)); e->bs_h = (uint8_t *)malloc((size_t)(e->cw / 4) * (e->ch / 8));
  e->im8 = (uint8_t *)malloc(nb8); e->ic8 = (uint32_t *)malloc(nb8 * 4);
  e->im16 = (uint8_t *)malloc(nb8 / 4); e->ic16 = (uint32_t *)malloc(nb8);
  e->im32 = (uint8_t *)malloc(nb8 / 16); e->ic32 = (uint32_t *)malloc(nb8 / 4);
  orc_bw_init(&e->au);

  orc_sps *s = &e->sps; memset(s, 0, sizeof(*s));
  s->general_profile_idc = 1; s->general_level_idc = level_for(e->cw, e->ch);
  s->chroma_format_idc = 1; s->width = e->cw; s->height = e->ch;
  s->conf_win_flag = (e->cw != c->width || e->ch != c->height);
  s->conf_right = (e->cw - c->width) / 2; s->conf_bottom = (e->ch - c->height) / 2;
  s->bit_depth_luma = s->bit_depth_chroma = 8; s->log2_max_poc_lsb = 8;
  s->max_dec_pic_buffering = 2; s->max_num_reorder = 0; s->max_latency_increase_plus1 = 0;
  s->log2_min_cb = 3; s->log2_diff_max_min_cb = 3; s->log2_min_tb = 2; s->log2_diff_max_min_tb = 3;
  s->max_th_depth_inter = 0; s->max_th_depth_intra = 0;
  s->amp_enabled = 0; s->sao_enabled = c->sao ? 1 : 0;
  s->num_st_rps = 1; s->st_rps[0].num_negative = 1; s->st_rps[0].delta_poc_s0[0] = -1; s->st_rps[0].used_s0[0] = 1;
  s->temporal_mvp_enabled = 0; s->strong_intra_smoothing = 1;
  s->vui_present = 1; s->vui_timing_present = 1; s->vui_num_units_in_tick = (uint32_t)c->fps_den; s->vui_time_scale = (uint32_t)c->fps_num;
  orc_sps_derive(s);
  orc_vps *v = &e->vps; memset(v, 0, sizeof(*v));
  v->timing_info_present = 1; v->num_units_in_tick = (uint32_t)c->fps_den; v->time_scale = (uint32_t)c->fps_num;
  orc_pps *p = &e->pps; memset(p, 0, sizeof(*p));
  p->num_ref_idx_l0_default = 1; p->num_ref_idx_l1_default = 1; p->init_qp = c->qp;
  p->entropy_coding_sync_enabled = c->wpp; p->loop_filter_across_slices = 1;
  p->deblocking_filter_control_present = !c->deblock; p->pps_deblocking_disabled = !c->deblock;
  p->log2_parallel_merge_level = 2; p->num_tile_columns = p->num_tile_rows = 1; p->uniform_spacing = 1;
  p->cu_qp_delta_enabled = e->cfg.qp_in_cu ? 1 : 0; p->diff_cu_qp_delta_depth = 0;
  p->dependent_slice_segments_enabled = e->cfg.slices == 1;
  { size_t nctu = (size_t)(e->cw / 64) * (e->ch / 64);
    if (c->sao) { e->sao = (orc_sao_params *)calloc(nctu, sizeof(orc_sao_params)); for (int i = 0; i < 3; i++) e->sao_in[i] = (pixel *)malloc(i ? npx / 4 : npx); }
    e->ctu_qt = (int8_t *)calloc(nctu, 1); e->ctu_qy = (int8_t *)calloc(nctu, 1); e->ctu_delta = (int8_t *)calloc(nctu, 1); e->ctu_first = (uint8_t *)calloc(nctu, 1); }

  memset(&e->av, 0, sizeof(e->av));
  e->av.pic_w = e->cw; e->av.pic_h = e->ch; e->av.ctb_log2 = 6; e->av.pic_w_ctbs = e->cw / 64;
  /* tiles: n full-width rows, uniform spacing (6.5.1: rowBd[i] = (i * PicHeightInCtbs) / n) */
  {
    int n = c->tile_rows, nc = e->cfg.tile_cols, wc = e->cw / 64, hc = e->ch / 64;
    e->tile_row_bd = (int *)calloc((size_t)n + 1, sizeof(int));
    for (int i = 0; i <= n; i++) e->tile_row_bd[i] = (i * hc) / n;
    for (int j = 0; j <= nc; j++) e->tile_col_bd[j] = (j * wc) / nc;
    if (n > 1 || nc > 1) {
      p->tiles_enabled = 1; p->num_tile_columns = nc; p->num_tile_rows = n; p->uniform_spacing = 1; p->loop_filter_across_tiles = 1;
      e->ctb_tile = (int16_t *)calloc((size_t)wc * hc, sizeof(int16_t));
      for (int i = 0; i < n; i++) for (int cy = e->tile_row_bd[i]; cy < e->tile_row_bd[i + 1]; cy++)
        for (int j = 0; j < nc; j++) for (int cx = e->tile_col_bd[j]; cx < e->tile_col_bd[j + 1]; cx++) e->ctb_tile[cy * wc + cx] = (int16_t)(i * nc + j);
      e->av.ctb_tile = e->ctb_tile;
    }
  }
  return e;
}

void orc_enc_close(orc_encoder *e)
{
  if (!e) return;
  for (int i = 0; i < 2; i++) orc_pic_free(&e->pics[i]);
  for (int i = 0; i < 3; i++) { free(e->src[i]); free(e->coef[i]); free(e->predeblock[i]); }
  free(e->refpad); free(e->prev_src); free(e->sao); for (int i = 0; i < 3; i++) free(e->sao_in[i]);
  free(e->cu_log2); free(e->cu_intra); free(e->cu_flags); free(e->cu_merge_idx); free(e->cu_mvp_idx);
  free(e->cu_intra_mode); free(e->cu_cbf); free(e->cu_mv); free(e->cu_mvd); free(e->bs_v); free(e->bs_h);
  free(e->im8); free(e->im16); free(e->im32); free(e->ic8); free(e->ic16); free(e->ic32);
  orc_bw_free(&e->au);
  free(e);
}

/* ------------------------------------------------------------------ input */
static void load_input(orc_encoder *e, const pixel *y, const pixel *u, const pixel *v)
{
  const pixel *in[3] = { y, u, v };
  for (int c = 0; c < 3; c++) {
    int w = c ? e->cfg.width / 2 : e->cfg.width, h = c ? e->cfg.height / 2 : e->cfg.height;
    int cw = c ? e->cw / 2 : e->cw, ch = c ? e->ch / 2 : e->ch;
    for (int yy = 0; yy < ch; yy++) {
      const pixel *srow = in[c] + (size_t)ORC_MIN(yy, h - 1) * w;
      pixel *drow = e->src[c] + (size_t)yy * cw;
      memcpy(drow, srow, (size_t)w);
      for (int xx = w; xx < cw; xx++) drow[xx] = srow[w - 1];
    }
  }
}

static inline int b8i(const orc_encoder *e, int x, int y) { return (y >> 3) * e->b8w + (x >> 3); }
static void set_cu(orc_encoder *e, uint8_t *arr, int x0, int y0, int n, int v)
{
  for (int y = y0; y < y0 + n; y += 8) for (int x = x0; x < x0 + n; x += 8) arr[b8i(e, x, y)] = (uint8_t)v;
}
static void fill_b4(orc_pic *p, uint8_t *arr, int x0, int y0, int n, int v)
{
  for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) arr[(y >> 2) * p->b4_w + (x >> 2)] = (uint8_t)v;
}

/* common side-info of a CU whose TU == CU */
static void mark_cu(orc_encoder *e, int x0, int y0, int log2, int pred_mode)
{
  orc_pic *p = e->cur; int n = 1 << log2;
  fill_b4(p, p->pred_mode, x0, y0, n, pred_mode);
  fill_b4(p, p->ct_depth, x0, y0, n, 6 - log2);
  for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) p->qp_y[(y >> 2) * p->b4_w + (x >> 2)] = (int8_t)e->qp;
  for (int i = 0; i < n; i += 4) {
    p->edge_v[((y0 + i) >> 2) * p->b4_w + (x0 >> 2)] |= 3;
    p->edge_h[(y0 >> 2) * p->b4_w + ((x0 + i) >> 2)] |= 3;
  }
}

/* residual -> levels (stored plane-shaped) -> reconstruction.  Returns cbf. */
static int ctu_target_qp(const orc_encoder *e, int x_luma, int y_luma) { return e->ctu_qt[(y_luma >> 6) * (e->cw / 64) + (x_luma >> 6)]; }

static int scan_idx_for(int intra, int log2, int cidx, int mode);
static int code_block(orc_encoder *e, int cidx, int x0, int y0, int n, int qp, int intra, int scan_idx)
{
  orc_pic *p = e->cur;
  int stride = p->stride[cidx];
  pixel *rec = p->plane[cidx] + y0 * stride + x0;            /* holds the prediction on entry */
  const pixel *src = e->src[cidx] + y0 * stride + x0;
  int16_t res[32 * 32], cf[32 * 32], lv[32 * 32];
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) res[y * n + x] = (int16_t)(src[y * stride + x] - rec[y * stride + x]);
  if (e->cfg.lossless) {                                       /* cu_transquant_bypass: the level at (x, y) is the residual sample at (x, y) (7.3.8.11, 8.6.2) */
    int16_t *cp = e->coef[cidx] + y0 * stride + x0;
    int any = 0;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) { cp[y * stride + x] = res[y * n + x]; any |= res[y * n + x] != 0; rec[y * stride + x] = src[y * stride + x]; }
    return any;
  }
  orc_fwd_transform(res, cf, n, 0);
  /* `scaling-list default`: the scaling factors of the block's size, colour component and prediction mode (the default lists, hevc_scaling.c) */
  const uint8_t *m = e->cfg.scaling_list ? e->sfac[orc_log2((unsigned)n) - 2][orc_scaling_matrix_id(orc_log2((unsigned)n) - 2, cidx, !intra)] : NULL;
  int nz;
  if (e->cfg.rdoq || e->cfg.signhide) {                        /* "uvgx RDOQ v1" / sign data hiding: a pass over the levels (hevc_transform.h) */
    uint16_t aux[32 * 32];
    orc_quant_aux_m(cf, lv, aux, n, qp, intra, m);
    nz = orc_adjust_levels(lv, aux, n, scan_idx, e->cfg.rdoq, e->cfg.signhide);
  } else nz = orc_quant_m(cf, lv, n, qp, intra, m);
  int16_t *cp = e->coef[cidx] + y0 * stride + x0;
  for (int y = 0; y < n; y++) memcpy(cp + y * stride, lv + y * n, sizeof(int16_t) * (size_t)n);
  if (nz) {
    orc_dequant_m(lv, cf, n, qp, m);
    orc_inv_transform(cf, res, n, 0);
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) rec[y * stride + x] = (pixel)orc_clip_pixel(rec[y * stride + x] + res[y * n + x]);
  }
  return nz != 0;
}

/* ------------------------------------------------------------------ intra pictures */
static uint32_t sad_block(const pixel *a, int as, const pixel *b, int bs, int n)
{
  uint32_t s = 0;
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) s += (uint32_t)orc_abs(a[y * as + x] - b[y * bs + x]);
  return s;
}

/* Sum of absolute transformed differences: the 8x8 blocks of the n x n difference go through the 8x8 Hadamard transform (H d H^T, H of
 * +-1 entries; which of the equivalent orderings of H's rows is used does not change the sum), a block's cost is (sum |coefficient| + 2)
 * >> 2 -- HM's and Kvazaar's normalisation, comparable in size with a SAD -- and the costs of the blocks add up. */
static uint32_t satd_block(const pixel *a, int as, const pixel *b, int bs, int n)
{
  uint32_t total = 0;
  for (int by = 0; by < n; by += 8)
    for (int bx = 0; bx < n; bx += 8) {
      int d[8][8], t[8][8];
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) d[y][x] = (int)a[(by + y) * as + bx + x] - (int)b[(by + y) * bs + bx + x];
      for (int y = 0; y < 8; y++)                      /* rows */
        for (int k = 0; k < 8; k++) {
          int acc = 0;
          for (int x = 0; x < 8; x++) acc += (__builtin_popcount((unsigned)(k & x)) & 1) ? -d[y][x] : d[y][x];
          t[y][k] = acc;
        }
      uint32_t sum = 0;
      for (int k = 0; k < 8; k++)                      /* columns */
        for (int x = 0; x < 8; x++) {
          int acc = 0;
          for (int y = 0; y < 8; y++) acc += (__builtin_popcount((unsigned)(k & y)) & 1) ? -t[y][x] : t[y][x];
          sum += (uint32_t)orc_abs(acc);
        }
      total += (sum + 2) >> 2;
    }
  return total;
}

/* Mode decision from SOURCE samples: for every aligned n x n block the mode with the least cost (SATD, or SAD with satd = 0)
 * between the source block and its prediction from source neighbours (ties: lowest mode). */
static void intra_analyse_size(orc_encoder *e, int n, uint8_t *best_mode, uint32_t *best_cost)
{
  int bw = e->cw / n, bh = e->ch / n;
  pixel left[129], top[129], pred[32 * 32];
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      int x0 = bx * n, y0 = by * n;
      orc_intra_refs(&e->av, e->src[0], e->cw, 0, x0, y0, n, left, top);
      uint32_t bc = 0xffffffffu; int bm = 0;
      /* "intra-chain" (default on): the reconstruction chain of an intra picture advances CTU by CTU along a wavefront whose lags are set by the samples a
       * CTU's blocks read from ANOTHER CTU beyond their own row / column of it -- the left CTU's below-left samples for the blocks on the CTU's left edge (they make
       * it wait for three quarters of the left CTU instead of three eighths) and the above-right CTU's samples for the CTU's above-right corner block (they make a
       * CTU row follow the row above at almost two CTUs' distance instead of one).  Those blocks choose among the modes that do not read these samples, directly or
       * through the reference filter (the masks: tests/test_intra_dependencies.py): 4 of a CTU's 16 blocks lose planar and about a third of the angular modes. */
      uint64_t excl = 0;
      if (e->cfg.intra_chain) {
        const uint64_t tr = n == 16 ? 0x7f9f80001ull : 0x7f8000001ull, bl = n == 16 ? 0x3f3fdull : 0x3fdull;      /* luma 16x16 / 8x8: modes reading p[x][-1], x >= n / p[-1][y], y >= n */
        if ((y0 & 63) == 0 && ((x0 + n) & 63) == 0 && orc_available(&e->av, x0, y0, x0 + n, y0 - 1)) excl |= tr;
        if ((x0 & 63) == 0 && orc_available(&e->av, x0, y0, x0 - 1, y0 + n)) excl |= bl;
      }
      for (int m = 0; m < 35; m++) {
        if ((excl >> m) & 1) continue;
        orc_intra_predict(left, top, n, 0, m, 1, pred, n);
        uint32_t c = e->cfg.satd ? satd_block(e->src[0] + y0 * e->cw + x0, e->cw, pred, n, n) : sad_block(e->src[0] + y0 * e->cw + x0, e->cw, pred, n, n);
        if (c < bc) { bc = c; bm = m; }
      }
      best_mode[by * bw + bx] = (uint8_t)bm; best_cost[by * bw + bx] = bc;
    }
}

static void intra_decide(orc_encoder *e)
{
  uint32_t pen = ((uint32_t)orc_lambda_q4[e->qp] * SPLIT_BITS) >> 4;
  intra_analyse_size(e, 8, e->im8, e->ic8);
  intra_analyse_size(e, 16, e->im16, e->ic16);
  /* Intra coding units are 16x16 or 8x8 -- Kvazaar's ultrafast shape (pu-depth-intra 2-3).  The reason here is
   * the reconstruction chain: a 32x32 block reads 64 samples down the left CTU and along the upper one, so a CTU could only start when
   * its neighbours are complete; with 16x16 blocks it starts when half of the left one is done (dec_kernels.hip k_dec_intra). */
  int w8 = e->cw / 8, w16 = e->cw / 16, w32 = e->cw / 32;
  for (int y32 = 0; y32 < e->ch / 32; y32++)
    for (int x32 = 0; x32 < w32; x32++) {
      uint32_t c16sum = 0; int split16[4];
      for (int k = 0; k < 4; k++) {
        int x16 = x32 * 2 + (k & 1), y16 = y32 * 2 + (k >> 1);
        uint32_t c8 = pen;
        for (int j = 0; j < 4; j++) c8 += e->ic8[(y16 * 2 + (j >> 1)) * w8 + x16 * 2 + (j & 1)];
        uint32_t c16 = e->ic16[y16 * w16 + x16];
        split16[k] = c8 < c16;
        c16sum += split16[k] ? c8 : c16;
      }
      int split32 = 1; (void)c16sum;
      int x0 = x32 * 32, y0 = y32 * 32;
      if (!split32) {
        set_cu(e, e->cu_log2, x0, y0, 32, 5); set_cu(e, e->cu_intra_mode, x0, y0, 32, e->im32[y32 * w32 + x32]);
      } else for (int k = 0; k < 4; k++) {
        int x16 = x32 * 2 + (k & 1), y16 = y32 * 2 + (k >> 1);
        if (!split16[k]) {
          set_cu(e, e->cu_log2, x16 * 16, y16 * 16, 16, 4); set_cu(e, e->cu_intra_mode, x16 * 16, y16 * 16, 16, e->im16[y16 * w16 + x16]);
        } else for (int j = 0; j < 4; j++) {
          int x8 = x16 * 2 + (j & 1), y8 = y16 * 2 + (j >> 1);
          e->cu_log2[y8 * w8 + x8] = 3; e->cu_intra_mode[y8 * w8 + x8] = e->im8[y8 * w8 + x8];
        }
      }
    }
}

static void intra_recon_cu(orc_encoder *e, int x0, int y0, int log2)
{
  orc_pic *p = e->cur;
  int n = 1 << log2, mode = e->cu_intra_mode[b8i(e, x0, y0)];
  pixel left[129], top[129];
  int qpl = ctu_target_qp(e, x0, y0), qpc = orc_chroma_qp(qpl, 0);
  mark_cu(e, x0, y0, log2, MODE_INTRA);
  fill_b4(p, p->intra_mode, x0, y0, n, mode);
  orc_intra_refs(&e->av, p->plane[0], p->stride[0], 0, x0, y0, n, left, top);
  orc_intra_predict(left, top, n, 0, mode, 1, p->plane[0] + y0 * p->stride[0] + x0, p->stride[0]);
  int cbf = code_block(e, 0, x0, y0, n, qpl, 1, scan_idx_for(1, log2, 0, mode));
  fill_b4(p, p->tu_nz, x0, y0, n, cbf);
  for (int c = 1; c <= 2; c++) {
    int cx = x0 / 2, cy = y0 / 2, cn = n / 2;
    orc_intra_refs(&e->av, p->plane[c], p->stride[c], c, cx, cy, cn, left, top);
    orc_intra_predict(left, top, cn, c, mode, 1, p->plane[c] + cy * p->stride[c] + cx, p->stride[c]);
    cbf |= code_block(e, c, cx, cy, cn, qpc, 1, scan_idx_for(1, log2 - 1, c, mode)) << c;
  }
  set_cu(e, e->cu_cbf, x0, y0, n, cbf);
  set_cu(e, e->cu_intra, x0, y0, n, 1);
  set_cu(e, e->cu_flags, x0, y0, n, 0);
}

static void intra_recon_tree(orc_encoder *e, int x0, int y0, int log2)
{
  int cl = e->cu_log2[b8i(e, x0, y0)];
  if (cl >= log2) { if (e->is_intra || e->cu_intra[b8i(e, x0, y0)]) intra_recon_cu(e, x0, y0, log2); return; }      /* (P pictures: the intra units only) */
  int h = 1 << (log2 - 1);
  intra_recon_tree(e, x0, y0, log2 - 1); intra_recon_tree(e, x0 + h, y0, log2 - 1);
  intra_recon_tree(e, x0, y0 + h, log2 - 1); intra_recon_tree(e, x0 + h, y0 + h, log2 - 1);
}

static void encode_intra_picture(orc_encoder *e)
{
  intra_decide(e);
  for (int cy = 0; cy < e->ch; cy += 64)
    for (int cx = 0; cx < e->cw; cx += 64)
      for (int k = 0; k < 4; k++) intra_recon_tree(e, cx + (k & 1) * 32, cy + (k >> 1) * 32, 5);
}

/* ------------------------------------------------------------------ inter pictures */
/* The picture the integer search looks at: the reference picture's reconstruction, or -- "uvgx search pipelining v1", option me-source -- the previous
 * INPUT picture: then the search of picture t + 1 depends on nothing picture t's reconstruction loop produces and the two run side by side (what hardware
 * encoders do).  Only the search moves: early termination, the 32x32 / 16x16 costs and the intra-in-P gate are priced on that picture; fractional
 * refinement (subme), motion compensation and everything behind them use the reconstruction as before. */
static void build_refpad(orc_encoder *e)
{
  int st = e->refpad_stride;
  const pixel *plane = e->cfg.me_source && e->prev_src ? e->prev_src : e->ref->plane[0];
  const int pstride = e->cfg.me_source && e->prev_src ? e->cw : e->ref->stride[0];
  for (int y = -ME_PAD; y < e->ch + ME_PAD; y++) {
    const pixel *srow = plane + (size_t)orc_clip3(0, e->ch - 1, y) * pstride;
    pixel *drow = e->refpad + (size_t)(y + ME_PAD) * st;
    for (int x = -ME_PAD; x < e->cw + ME_PAD; x++) drow[x + ME_PAD] = srow[orc_clip3(0, e->cw - 1, x)];
  }
}

static inline uint32_t sad16(const pixel *a, int as, const pixel *b, int bs)
{
  uint32_t s = 0;
  for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) s += (uint32_t)orc_abs(a[y * as + x] - b[y * bs + x]);
  return s;
}

/* "uvgx subme v1" (kvazaar subme 1..4): fractional-sample refinement of one CU's vector after the integer search.  Two steps
 * of eight neighbours each -- half-sample positions around the integer vector, then quarter-sample positions around the best
 * so far -- of which subme switches on the horizontal / vertical four (1, 3) and the diagonal four (2, 4).  A candidate's cost
 * is the SATD (satd_block) between the source block and the normative prediction (8.5.3.3.3 + 8.5.3.3.4.2) plus
 * (lambda * bits of the vector coded as a difference from zero) >> 4, the same rate term as the integer search; the centre
 * is re-priced the same way; ties go to the centre, then to the earlier candidate.  A fractional component needs reference
 * rows / columns up to 4 away from the displaced block: candidates whose rows would leave the tile (or, with mv-constraint
 * frame, whose rows or columns would leave the picture) are dropped. */
static int subme_allowed(const orc_encoder *e, int x0, int y0, int n, int mvx, int mvy, int ty0, int ty1)
{
  int ix = mvx >> 2, iy = mvy >> 2, mx = (mvx & 7) ? 4 : 0, my = (mvy & 7) ? 4 : 0;
  if ((ty0 > 0 && y0 + iy - my < ty0) || (ty1 < e->ch && y0 + iy + n + my > ty1)) return 0;
  {
    int tx0 = 0, tx1 = e->cw;
    for (int j = 0; j < e->cfg.tile_cols; j++) if ((x0 >> 6) >= e->tile_col_bd[j] && (x0 >> 6) < e->tile_col_bd[j + 1]) { tx0 = e->tile_col_bd[j] * 64; tx1 = e->tile_col_bd[j + 1] * 64; }
    if ((tx0 > 0 && x0 + ix - mx < tx0) || (tx1 < e->cw && x0 + ix + n + mx > tx1)) return 0;
  }
  if (e->cfg.mv_frame) {
    if (e->cfg.mv_frame == 1) { mx = (mvx & 3) ? 4 : 0; my = (mvy & 3) ? 4 : 0; }     /* plain frame constraint: integer vectors may touch the edge */
    if (x0 + ix - mx < 0 || x0 + ix + n + mx > e->cw || y0 + iy - my < 0 || y0 + iy + n + my > e->ch) return 0;
  }
  return 1;
}
static uint32_t subme_cost(orc_encoder *e, int x0, int y0, int n, int mvx, int mvy)
{
  int16_t tmp[32 * 32]; pixel pred[32 * 32];
  orc_pic *r = e->ref;
  orc_mc_luma(r->plane[0], r->stride[0], r->w, r->h, x0, y0, n, n, mvx, mvy, tmp, 32);
  orc_pred_uni(tmp, 32, pred, 32, n, n);
  uint32_t rate = ((uint32_t)orc_lambda_q4[e->qp] * (uint32_t)(orc_mvd_bits(mvx) + orc_mvd_bits(mvy))) >> 4;
  return satd_block(e->src[0] + y0 * e->cw + x0, e->cw, pred, 32, n) + rate;
}
static void subme_refine(orc_encoder *e, int x0, int y0, int n, int ty0, int ty1)
{
  static const int8_t off[8][2] = { {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1} };
  int bi = b8i(e, x0, y0), cx = e->cu_mv[bi * 2], cy = e->cu_mv[bi * 2 + 1];
  uint32_t best = subme_cost(e, x0, y0, n, cx, cy) << 4;                  /* key = cost << 4 | candidate (0 = centre) */
  for (int step = 0; step < 2; step++) {
    int scale = step ? 1 : 2, ncand = (e->cfg.subme >= (step ? 4 : 2)) ? 8 : ((e->cfg.subme >= (step ? 3 : 1)) ? 4 : 0);
    uint32_t b = best & ~15u;
    for (int k = 0; k < ncand; k++) {
      int mx = cx + off[k][0] * scale, my = cy + off[k][1] * scale;
      if (!subme_allowed(e, x0, y0, n, mx, my, ty0, ty1)) continue;
      uint32_t key = (subme_cost(e, x0, y0, n, mx, my) << 4) | (uint32_t)(k + 1);
      if (key < b) b = key;
    }
    if (b & 15u) { cx += off[(b & 15u) - 1][0] * scale; cy += off[(b & 15u) - 1][1] * scale; }
    best = b & ~15u;
  }
  for (int y = y0; y < y0 + n; y += 8) for (int x = x0; x < x0 + n; x += 8) { e->cu_mv[b8i(e, x, y) * 2] = (int16_t)cx; e->cu_mv[b8i(e, x, y) * 2 + 1] = (int16_t)cy; }
}

/* Full search for one 32x32 block: candidates in raster order (dy outer, dx inner),
 * cost = SAD + (lambda * bits(mv as mvd from zero)) >> 4, key = cost << 13 | candidate index. */
static void me_block32(orc_encoder *e, int x0, int y0)
{
  int R = e->cfg.search_range, st = e->refpad_stride;
  uint32_t lam = orc_lambda_q4[e->qp];
  uint32_t best16[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu }, best32 = 0xffffffffu;
  int idx = 0;
  if (e->cfg.me_early) {                              /* early termination: static content is not searched */
    uint32_t s0 = 0;
    for (int k = 0; k < 4; k++) {
      int bx = x0 + (k & 1) * 16, by = y0 + (k >> 1) * 16;
      s0 += sad16(e->src[0] + by * e->cw + bx, e->cw, e->refpad + (size_t)(by + ME_PAD) * st + bx + ME_PAD, st);
    }
    if (s0 <= 64u * lam) {
      set_cu(e, e->cu_log2, x0, y0, 32, 5); set_cu(e, e->cu_intra, x0, y0, 32, 0);
      for (int y = y0; y < y0 + 32; y += 8) for (int x = x0; x < x0 + 32; x += 8) { e->cu_mv[b8i(e, x, y) * 2] = 0; e->cu_mv[b8i(e, x, y) * 2 + 1] = 0; }
      return;
    }
  }
  /* Tile constraint: the 32x32 block displaced by dy (plus 4 rows each side for the chroma half-sample taps when dy is
   * odd) must stay inside its tile, except across the picture's own top and bottom edges where padding is normative. */
  int ty0 = 0, ty1 = e->ch;
  for (int i = 0; i < e->cfg.tile_rows; i++) if ((y0 >> 6) >= e->tile_row_bd[i] && (y0 >> 6) < e->tile_row_bd[i + 1]) { ty0 = e->tile_row_bd[i] * 64; ty1 = e->tile_row_bd[i + 1] * 64; }
  int tx0 = 0, tx1 = e->cw;                                      /* ... and the same in x with tile columns */
  for (int j = 0; j < e->cfg.tile_cols; j++) if ((x0 >> 6) >= e->tile_col_bd[j] && (x0 >> 6) < e->tile_col_bd[j + 1]) { tx0 = e->tile_col_bd[j] * 64; tx1 = e->tile_col_bd[j + 1] * 64; }
  for (int dy = -R; dy <= R; dy++)
    for (int dx = -R; dx <= R; dx++, idx++) {
      int m = (dy & 1) ? 4 : 0, mxt = (dx & 1) ? 4 : 0;
      if ((ty0 > 0 && y0 + dy - m < ty0) || (ty1 < e->ch && y0 + dy + 32 + m > ty1)) continue;
      if ((tx0 > 0 && x0 + dx - mxt < tx0) || (tx1 < e->cw && x0 + dx + 32 + mxt > tx1)) continue;
      if (e->cfg.mv_frame) {                                   /* mv-constraint frame: the displaced block stays inside the picture */
        int my = (e->cfg.mv_frame == 2 && (dy & 1)) ? 4 : 0, mx = (e->cfg.mv_frame == 2 && (dx & 1)) ? 4 : 0;
        if (x0 + dx - mx < 0 || x0 + dx + 32 + mx > e->cw || y0 + dy - my < 0 || y0 + dy + 32 + my > e->ch) continue;
      }
      uint32_t rate = (lam * (uint32_t)(orc_mvd_bits(dx * 4) + orc_mvd_bits(dy * 4))) >> 4;
      uint32_t s32 = 0;
      for (int k = 0; k < 4; k++) {
        int bx = x0 + (k & 1) * 16, by = y0 + (k >> 1) * 16;
        uint32_t s = sad16(e->src[0] + by * e->cw + bx, e->cw, e->refpad + (size_t)(by + dy + ME_PAD) * st + bx + dx + ME_PAD, st);
        s32 += s;
        uint32_t key = ((s + rate) << 13) | (uint32_t)idx;
        if (key < best16[k]) best16[k] = key;
      }
      uint32_t key = ((s32 + rate) << 13) | (uint32_t)idx;
      if (key < best32) best32 = key;
    }
  uint32_t pen = (lam * SPLIT_BITS) >> 4;
  uint32_t csplit = pen;
  for (int k = 0; k < 4; k++) csplit += best16[k] >> 13;
  int W = 2 * R + 1;
  if (csplit < (best32 >> 13)) {
    for (int k = 0; k < 4; k++) {
      int bx = x0 + (k & 1) * 16, by = y0 + (k >> 1) * 16, ci = (int)(best16[k] & 0x1fff);
      set_cu(e, e->cu_log2, bx, by, 16, 4);
      for (int y = by; y < by + 16; y += 8) for (int x = bx; x < bx + 16; x += 8) {
        e->cu_mv[b8i(e, x, y) * 2] = (int16_t)(((ci % W) - R) * 4); e->cu_mv[b8i(e, x, y) * 2 + 1] = (int16_t)(((ci / W) - R) * 4);
      }
    }
  } else {
    int ci = (int)(best32 & 0x1fff);
    set_cu(e, e->cu_log2, x0, y0, 32, 5);
    for (int y = y0; y < y0 + 32; y += 8) for (int x = x0; x < x0 + 32; x += 8) {
      e->cu_mv[b8i(e, x, y) * 2] = (int16_t)(((ci % W) - R) * 4); e->cu_mv[b8i(e, x, y) * 2 + 1] = (int16_t)(((ci / W) - R) * 4);
    }
  }
  set_cu(e, e->cu_intra, x0, y0, 32, 0);
  if (e->cfg.intra_in_p) {
    /* "uvgx intra-in-P v1".  Inter cost of a 16x16 quarter = what the search found for it (split) or a quarter of the 32x32 block's cost;
     * quarters above the gate are priced as intra blocks by the intra picture's analysis (source neighbours: nothing depends on other
     * blocks' decisions) and go intra when that is cheaper by INTRA_P_BITS bins (lam: the picture's lambda, also under rate control v2).  A block with an intra quarter is coded as four 16x16
     * units (the inter ones keep their vectors; an intra quarter is one 16x16 or four 8x8 intra units). */
    uint32_t ic[4]; int any = 0, cand[4];
    for (int k = 0; k < 4; k++) {
      ic[k] = csplit < (best32 >> 13) ? best16[k] >> 13 : ((best32 >> 13) + 2) >> 2;
      cand[k] = ic[k] > INTRA_P_GATE * lam; any |= cand[k];
    }
    if (any) {
      if (!e->intra_p_ready) { if (e->cfg.intra_in_p >= 2) intra_analyse_size(e, 8, e->im8, e->ic8); intra_analyse_size(e, 16, e->im16, e->ic16); e->intra_p_ready = 1; }
      int w8 = e->cw / 8, w16 = e->cw / 16, chosen = 0, sp16[4];
      for (int k = 0; k < 4; k++) {
        int x16 = x0 / 16 + (k & 1), y16 = y0 / 16 + (k >> 1);
        uint32_t c8 = pen;
        if (e->cfg.intra_in_p >= 2) for (int j = 0; j < 4; j++) c8 += e->ic8[(y16 * 2 + (j >> 1)) * w8 + x16 * 2 + (j & 1)];
        uint32_t c16 = e->ic16[y16 * w16 + x16];
        sp16[k] = e->cfg.intra_in_p >= 2 && c8 < c16;          /* level 1: one 16x16 intra unit per quarter, no 8x8 units (the fast presets: a CTU of 8x8 intra units is twice the wavefront steps in the encoder's and the decoder's chain) */
        uint32_t cintra = (sp16[k] ? c8 : c16) + ((lam * INTRA_P_BITS) >> 4);
        cand[k] = cand[k] && cintra < ic[k];
        chosen |= cand[k];
      }
      if (chosen) {
        for (int k = 0; k < 4; k++) {
          int bx = x0 + (k & 1) * 16, by = y0 + (k >> 1) * 16, x16 = bx / 16, y16 = by / 16;
          if (!cand[k]) { set_cu(e, e->cu_log2, bx, by, 16, 4); continue; }
          set_cu(e, e->cu_intra, bx, by, 16, 1);
          for (int y = by; y < by + 16; y += 8) for (int x = bx; x < bx + 16; x += 8) { e->cu_mv[b8i(e, x, y) * 2] = 0; e->cu_mv[b8i(e, x, y) * 2 + 1] = 0; }
          if (!sp16[k]) { set_cu(e, e->cu_log2, bx, by, 16, 4); set_cu(e, e->cu_intra_mode, bx, by, 16, e->im16[y16 * w16 + x16]); }
          else for (int j = 0; j < 4; j++) {
            int x8 = x16 * 2 + (j & 1), y8 = y16 * 2 + (j >> 1);
            e->cu_log2[y8 * w8 + x8] = 3; e->cu_intra_mode[y8 * w8 + x8] = e->im8[y8 * w8 + x8];
          }
        }
      }
    }
  }
  if (e->cfg.subme > 0) {                                        /* fractional-sample refinement of the (inter) CUs just decided */
    if (e->cu_log2[b8i(e, x0, y0)] == 5) subme_refine(e, x0, y0, 32, ty0, ty1);
    else for (int k = 0; k < 4; k++) if (!e->cu_intra[b8i(e, x0 + (k & 1) * 16, y0 + (k >> 1) * 16)]) subme_refine(e, x0 + (k & 1) * 16, y0 + (k >> 1) * 16, 16, ty0, ty1);
  }
  if (e->cfg.test_mv_jitter && e->cfg.tile_rows == 1) {          /* test hook: fractional vectors for the decoder tests */
    for (int y = y0; y < y0 + 32; y += 8) for (int x = x0; x < x0 + 32; x += 8) {
      int n = 1 << e->cu_log2[b8i(e, x, y)], ox = x & ~(n - 1), oy = y & ~(n - 1);
      uint32_t hsh = (uint32_t)(ox * 73856093) ^ (uint32_t)(oy * 19349663) ^ (uint32_t)(e->frame_idx * 83492791);
      hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
      if (x == ox && y == oy) {                                    /* once per CU: every 8x8 cell of the CU gets the same vector */
        int jx = (int)(hsh % 7) - 3, jy = (int)((hsh >> 8) % 7) - 3;
        for (int yy = oy; yy < oy + n; yy += 8) for (int xx = ox; xx < ox + n; xx += 8) {
          e->cu_mv[b8i(e, xx, yy) * 2] = (int16_t)(e->cu_mv[b8i(e, xx, yy) * 2] + jx);
          e->cu_mv[b8i(e, xx, yy) * 2 + 1] = (int16_t)(e->cu_mv[b8i(e, xx, yy) * 2 + 1] + jy);
        }
      }
    }
  }
}

static void inter_recon_cu(orc_encoder *e, int x0, int y0, int log2)
{
  orc_pic *p = e->cur, *r = e->ref;
  int n = 1 << log2;
  int16_t mv[2] = { e->cu_mv[b8i(e, x0, y0) * 2], e->cu_mv[b8i(e, x0, y0) * 2 + 1] };
  int16_t tmp[32 * 32];
  int qpc = orc_chroma_qp(ctu_target_qp(e, x0, y0), 0);
  mark_cu(e, x0, y0, log2, MODE_INTER);
  for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) {
    orc_mvinfo *m = &p->mvf[(y >> 2) * p->b4_w + (x >> 2)];
    m->mv[0] = mv[0]; m->mv[1] = mv[1]; m->ref_idx = 0;
  }
  orc_mc_luma(r->plane[0], r->stride[0], r->w, r->h, x0, y0, n, n, mv[0], mv[1], tmp, 32);
  orc_pred_uni(tmp, 32, p->plane[0] + y0 * p->stride[0] + x0, p->stride[0], n, n);
  int cbf = code_block(e, 0, x0, y0, n, ctu_target_qp(e, x0, y0), 0, 0);
  fill_b4(p, p->tu_nz, x0, y0, n, cbf);
  for (int c = 1; c <= 2; c++) {
    int cx = x0 / 2, cy = y0 / 2, cn = n / 2;
    orc_mc_chroma(r->plane[c], r->stride[c], r->w / 2, r->h / 2, cx, cy, cn, cn, mv[0], mv[1], tmp, 32);
    orc_pred_uni(tmp, 32, p->plane[c] + cy * p->stride[c] + cx, p->stride[c], cn, cn);
    cbf |= code_block(e, c, cx, cy, cn, qpc, 0, 0) << c;
  }
  set_cu(e, e->cu_cbf, x0, y0, n, cbf);
  set_cu(e, e->cu_intra, x0, y0, n, 0);
}

/* merge / skip / AMVP signalling decided from the FINAL motion field of the picture */
static void inter_decide_signalling(orc_encoder *e, int x0, int y0, int log2)
{
  orc_pic *p = e->cur; int n = 1 << log2, bi = b8i(e, x0, y0);
  orc_mvpred_ctx mc; memset(&mc, 0, sizeof(mc)); mc.collocated_from_l0 = 1;
  mc.pic = p; mc.av = e->av; mc.log2_par_mrg_level = 2; mc.max_num_merge_cand = 5; mc.num_ref_idx = 1;
  mc.cur_poc = e->poc; mc.ref_poc[0] = e->poc - 1;
  int16_t mvx = e->cu_mv[bi * 2], mvy = e->cu_mv[bi * 2 + 1];
  orc_mvcand cand[5];
  orc_merge_candidates(&mc, x0, y0, n, x0, y0, n, n, 0, PART_2Nx2N, cand);
  int flags = 0, midx = 0, mvp = 0; int16_t mvdx = 0, mvdy = 0;
  for (int k = 0; k < 5; k++) if (cand[k].ref_idx == 0 && cand[k].mv[0] == mvx && cand[k].mv[1] == mvy) { flags = 2; midx = k; break; }
  if (flags && e->cu_cbf[bi] == 0) flags |= 1;
  if (!flags) {
    int16_t ac[2][2];
    orc_amvp_candidates(&mc, x0, y0, n, x0, y0, n, n, 0, 0, ac);
    int b0 = orc_mvd_bits(mvx - ac[0][0]) + orc_mvd_bits(mvy - ac[0][1]);
    int b1 = orc_mvd_bits(mvx - ac[1][0]) + orc_mvd_bits(mvy - ac[1][1]);
    mvp = b1 < b0;
    mvdx = (int16_t)(mvx - ac[mvp][0]); mvdy = (int16_t)(mvy - ac[mvp][1]);
  }
  for (int y = y0; y < y0 + n; y += 8) for (int x = x0; x < x0 + n; x += 8) {
    int i = b8i(e, x, y);
    e->cu_flags[i] = (uint8_t)flags; e->cu_merge_idx[i] = (uint8_t)midx; e->cu_mvp_idx[i] = (uint8_t)mvp;
    e->cu_mvd[i * 2] = mvdx; e->cu_mvd[i * 2 + 1] = mvdy;
  }
  if (flags & 1) fill_b4(p, p->pred_mode, x0, y0, n, MODE_SKIP);
}

/* ---- "uvgx rate control v2": feedback inside the picture, on top of the picture-level controller (rate_control()).
 * Level cost of a CTU = sum over its non-zero levels (all three planes) of 3 + 2 * floor(log2 |level|): about the bins a level takes
 * (significance, greater-1 / greater-2 or remainder, sign).  One unit of it is worth rc_ratio_q8 / 256 bits, measured on the P
 * pictures coded before (rc_picture_start).  After a group of CTU rows: estimated bits so far against the rows' share of the
 * picture's target T; more than 9/8 of it -> the next group's QP one step up, less than 7/8 -> one step down; the offset stays
 * within +-3 of the picture's QP.  Nothing moves until the ratio has been measured once. */
static uint32_t rc_ctu_cost(const orc_encoder *e, int cx, int cy)
{
  uint32_t c = 0;
  for (int pl = 0; pl < 3; pl++) {
    int S = pl ? 32 : 64, pw = pl ? e->cw / 2 : e->cw;
    const int16_t *p = e->coef[pl] + (size_t)(cy * S) * pw + cx * S;
    for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) { int l = p[y * pw + x]; if (l) c += 3u + 2u * (uint32_t)orc_log2((uint32_t)orc_abs(l)); }
  }
  return c;
}
static int rc_band_decide(int off, uint32_t cost, uint32_t ratio_q8, int64_t T, int rows_done, int rows_total)
{
  if (!ratio_q8) return off;
  uint64_t est = ((uint64_t)cost * ratio_q8) >> 8, tgt = ((uint64_t)T * (uint64_t)rows_done) / (uint64_t)rows_total;
  if (est * 8 > tgt * 9) off++; else if (est * 8 < tgt * 7) off--;
  return orc_clip3(-3, 3, off);
}
/* before picture t: the size of access unit t - 3 is known (the delay of the picture-level controller); if that was a P picture coded in
 * groups, its bits per unit of level cost update the ratio (new = (3 * old + measured + 2) >> 2; the first measurement is taken as is) */
static void rc_picture_start(orc_encoder *e)
{
  const int D = e->cfg.rc_delay >= 3 && e->cfg.rc_delay <= 7 ? e->cfg.rc_delay : 3;
  if (e->cfg.rc_bands <= 0 || e->frame_idx < D) return;
  int s3 = (e->frame_idx - D) & 7;
  if (!e->rc_cost_valid[s3]) return;
  e->rc_cost_valid[s3] = 0;
  uint64_t r = ((uint64_t)8 * e->rc_bytes[s3] << 8) / (e->rc_cost[s3] ? e->rc_cost[s3] : 1);
  if (r > (1u << 20)) r = 1u << 20;
  if (r < 1) r = 1;
  e->rc_ratio_q8 = e->rc_ratio_valid ? (uint32_t)((3 * (uint64_t)e->rc_ratio_q8 + r + 2) >> 2) : (uint32_t)r;
  e->rc_ratio_valid = 1;
}

static void intra_recon_tree(orc_encoder *e, int x0, int y0, int log2);
static void encode_inter_picture(orc_encoder *e)
{
  build_refpad(e);
  e->intra_p_ready = 0;
  for (int y = 0; y < e->ch; y += 32) for (int x = 0; x < e->cw; x += 32) me_block32(e, x, y);
  /* Reconstruction, in rc_bands groups of CTU rows when rate control v2 is on: after each group the level cost so far is priced
   * against the share of the picture's target the rows done are entitled to, and the QP of the group AFTER THE NEXT follows (rc_band_decide;
   * round 4: one group of lag -- groups 0 and 1 run at the picture's QP, group b + 2 at the running step moved by what groups 0 .. b cost --, so
   * that a group never waits for the group right in front of it: four strictly serial groups were 56 us of a 1080p picture's chain, against 27) */
  {
    int hc = e->ch / 64, wc = e->cw / 64, nb = e->cfg.rc_bands > 0 ? (e->cfg.rc_bands < hc ? e->cfg.rc_bands : hc) : 1;
    int offs[16]; memset(offs, 0, sizeof(offs));
    uint32_t cost = 0;
    const int64_t T = e->cfg.bitrate > 0 ? ((int64_t)e->cfg.bitrate * e->cfg.fps_den) / (e->cfg.fps_num > 0 ? e->cfg.fps_num : 1) : 0;
    for (int b = 0; b < nb; b++) {
      int r0 = (b * hc) / nb, r1 = ((b + 1) * hc) / nb;
      for (int y = r0 * 64; y < r1 * 64; y += 32) for (int x = 0; x < e->cw; x += 32) {
        if (e->cu_log2[b8i(e, x, y)] == 5) inter_recon_cu(e, x, y, 5);
        else for (int k = 0; k < 4; k++) {
          int qx = x + (k & 1) * 16, qy = y + (k >> 1) * 16;
          if (!e->cu_intra[b8i(e, qx, qy)]) { inter_recon_cu(e, qx, qy, 4); continue; }
          /* an intra quarter: no levels yet (they come behind the whole picture's inter units; rate control v2 prices the groups without them) */
          for (int pl = 0; pl < 3; pl++) {
            int sh = pl ? 1 : 0, pw = e->cw >> sh, n = 16 >> sh;
            for (int r = 0; r < n; r++) memset(e->coef[pl] + (size_t)((qy >> sh) + r) * pw + (qx >> sh), 0, sizeof(int16_t) * n);
          }
        }
      }
      if (e->cfg.rc_bands > 0) {
        for (int cy = r0; cy < r1; cy++) for (int cx = 0; cx < wc; cx++) cost += rc_ctu_cost(e, cx, cy);
        if (b + 2 < nb) {
          const int off = offs[b + 2] = rc_band_decide(offs[b + 1], cost, e->rc_ratio_valid ? e->rc_ratio_q8 : 0, T, r1, hc);
          const int r2 = ((b + 2) * hc) / nb, r3 = ((b + 3) * hc) / nb;
          for (int cy = r2; cy < r3; cy++) for (int cx = 0; cx < wc; cx++) e->ctu_qt[cy * wc + cx] = (int8_t)orc_clip3(0, 51, e->ctu_qt[cy * wc + cx] + off);
        }
      }
    }
    if (e->cfg.rc_bands > 0) { e->rc_cost[e->frame_idx & 7] = cost; e->rc_cost_valid[e->frame_idx & 7] = 1; }
  }
  /* intra coding units of the P picture: after every inter unit (their reference samples may lie in inter units anywhere around them),
   * CTU by CTU in decoding order, z-order inside the CTU */
  if (e->intra_p_ready) {
    for (int tr = 0; tr < e->cfg.tile_rows; tr++) for (int tc = 0; tc < e->cfg.tile_cols; tc++)
      for (int cy = e->tile_row_bd[tr]; cy < e->tile_row_bd[tr + 1]; cy++)
        for (int cx = e->tile_col_bd[tc]; cx < e->tile_col_bd[tc + 1]; cx++) intra_recon_tree(e, cx * 64, cy * 64, 6);
  }
  for (int y = 0; y < e->ch; y += 32) for (int x = 0; x < e->cw; x += 32) {
    if (e->cu_log2[b8i(e, x, y)] == 5) inter_decide_signalling(e, x, y, 5);
    else for (int k = 0; k < 4; k++) if (!e->cu_intra[b8i(e, x + (k & 1) * 16, y + (k >> 1) * 16)]) inter_decide_signalling(e, x + (k & 1) * 16, y + (k >> 1) * 16, 4);
  }
}

/* ------------------------------------------------------------------ entropy coding */
static void enc_last_prefix(orc_cabac_enc *c, int base, int log2, int cidx, int prefix)
{
  int off, sh, max = (log2 << 1) - 1;
  if (cidx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); sh = (log2 + 1) >> 2; }
  else { off = 15; sh = log2 - 2; }
  for (int i = 0; i < prefix; i++) orc_cenc_bin(c, base + off + (i >> sh), 1);
  if (prefix < max) orc_cenc_bin(c, base + off + (prefix >> sh), 0);
}
static void last_bin(int v, int *prefix, int *nb, int *suffix)
{
  if (v < 4) { *prefix = v; *nb = 0; *suffix = 0; return; }
  int len = orc_log2((unsigned)v);
  *prefix = 2 * len + ((v >> (len - 1)) & 1); *nb = len - 1; *suffix = v & ((1 << (len - 1)) - 1);
}
static void enc_abs_remaining(orc_cabac_enc *c, int v, int rice)
{
  if ((v >> rice) < 4) {
    int q = v >> rice;
    for (int i = 0; i < q; i++) orc_cenc_bypass(c, 1);
    orc_cenc_bypass(c, 0);
    orc_cenc_bypass_bits(c, (uint32_t)(v & ((1 << rice) - 1)), rice);
  } else {
    int x = v - (4 << rice), k = rice + 1;
    for (int i = 0; i < 4; i++) orc_cenc_bypass(c, 1);
    while (x >= (1 << k)) { orc_cenc_bypass(c, 1); x -= 1 << k; k++; }
    orc_cenc_bypass(c, 0);
    orc_cenc_bypass_bits(c, (uint32_t)x, k);
  }
}

static const uint8_t ctx_idx_map_4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

/* 7.3.8.11 residual_coding for a block with at least one non-zero level; lv is plane-shaped */
static void enc_residual(orc_cabac_enc *c, const int16_t *lv, int stride, int log2, int cidx, int scan_idx, int sign_hiding)
{
  int sb_log2 = log2 - 2, nsb = 1 << sb_log2;
  const uint8_t *sbx = orc_scan_x[scan_idx][sb_log2], *sby = orc_scan_y[scan_idx][sb_log2];
  const uint8_t *px = orc_scan_x[scan_idx][2], *py = orc_scan_y[scan_idx][2];
  uint8_t csbf[8][8]; memset(csbf, 0, sizeof(csbf));
  int last_sb = -1, last_pos = -1;
  for (int i = (1 << (2 * sb_log2)) - 1; i >= 0 && last_sb < 0; i--)
    for (int k = 15; k >= 0; k--)
      if (lv[((sby[i] << 2) + py[k]) * stride + (sbx[i] << 2) + px[k]]) { last_sb = i; last_pos = k; break; }
  for (int i = 0; i <= last_sb; i++) {
    int any = 0;
    for (int k = 0; k < 16 && !any; k++) any = lv[((sby[i] << 2) + py[k]) * stride + (sbx[i] << 2) + px[k]] != 0;
    csbf[sby[i]][sbx[i]] = (uint8_t)any;
  }
  int lx = (sbx[last_sb] << 2) + px[last_pos], ly = (sby[last_sb] << 2) + py[last_pos];
  if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
  int pxv, nbx, sfx, pyv, nby, sfy;
  last_bin(lx, &pxv, &nbx, &sfx); last_bin(ly, &pyv, &nby, &sfy);
  enc_last_prefix(c, CTX_LAST_X, log2, cidx, pxv);
  enc_last_prefix(c, CTX_LAST_Y, log2, cidx, pyv);
  if (pxv > 3) orc_cenc_bypass_bits(c, (uint32_t)sfx, nbx);
  if (pyv > 3) orc_cenc_bypass_bits(c, (uint32_t)sfy, nby);
  int c1 = 1;
  for (int i = last_sb; i >= 0; i--) {
    int xs = sbx[i], ys = sby[i], infer_dc = 0;
    int right = (xs < nsb - 1) ? csbf[ys][xs + 1] : 0, below = (ys < nsb - 1) ? csbf[ys + 1][xs] : 0;
    if (i < last_sb && i > 0) {
      orc_cenc_bin(c, CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0), csbf[ys][xs]);
      infer_dc = 1;
    } else csbf[ys][xs] = 1;      /* inferred for the last and the DC sub-block (affects neighbour contexts) */
    if (!csbf[ys][xs]) continue;
    int16_t v[16]; int nsig = 0;
    for (int k = 0; k < 16; k++) { v[k] = lv[((ys << 2) + py[k]) * stride + (xs << 2) + px[k]]; nsig += v[k] != 0; }
    int start = (i == last_sb) ? last_pos - 1 : 15;
    int prev_csbf = right | (below << 1);
    for (int k = start; k >= 0; k--) {
      int xp = px[k], yp = py[k], xc = (xs << 2) + xp, yc = (ys << 2) + yp;
      if (k > 0 || !infer_dc) {
        int sc;
        if (log2 == 2) sc = ctx_idx_map_4x4[(yc << 2) + xc];
        else if (xc + yc == 0) sc = 0;
        else {
          if (prev_csbf == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
          else if (prev_csbf == 1) sc = (yp == 0) ? 2 : (yp == 1) ? 1 : 0;
          else if (prev_csbf == 2) sc = (xp == 0) ? 2 : (xp == 1) ? 1 : 0;
          else sc = 2;
          if (cidx == 0) { if (i > 0) sc += 3; sc += (log2 == 3) ? ((scan_idx == 0) ? 9 : 15) : 21; }
          else sc += (log2 == 3) ? 9 : 12;
        }
        orc_cenc_bin(c, CTX_SIG + (cidx ? 27 : 0) + sc, v[k] != 0);
        if (v[k]) infer_dc = 0;
      }
    }
    if (!nsig) continue;
    int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
    if (c1 == 0) ctx_set++;
    c1 = 1;
    int ng1 = 0, last_g1_pos = -1;
    for (int k = 15; k >= 0; k--) if (v[k]) {
      if (ng1 < 8) {
        int g1 = orc_abs(v[k]) > 1;
        orc_cenc_bin(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1, g1);
        ng1++;
        if (g1) { c1 = 0; if (last_g1_pos == -1) last_g1_pos = k; }
        else if (c1 > 0 && c1 < 3) c1++;
      }
    }
    if (last_g1_pos != -1) orc_cenc_bin(c, CTX_GT2 + (cidx ? 4 : 0) + ctx_set, orc_abs(v[last_g1_pos]) > 2);
    /* sign_data_hiding (7.3.8.11): the sign of the group's first coefficient in scan order is not sent when it lies more than three
     * positions below the last one (in the block's last group: below the last significant position) */
    int first_sig = 16, last_sig = -1;
    for (int k = 0; k < 16; k++) if (v[k]) { if (first_sig == 16) first_sig = k; last_sig = k; }
    const int hidden = sign_hiding && last_sig - first_sig > 3;
    for (int k = 15; k >= 0; k--) if (v[k] && !(hidden && k == first_sig)) orc_cenc_bypass(c, v[k] < 0);
    int num_sig = 0, rice = 0;
    for (int k = 15; k >= 0; k--) if (v[k]) {
      int a = orc_abs(v[k]);
      int base = (num_sig < 8) ? ((k == last_g1_pos) ? 3 : 2) : 1;
      if (a >= base) {
        enc_abs_remaining(c, a - base, rice);
        if (a > 3 * (1 << rice)) rice = ORC_MIN(rice + 1, 4);
      }
      num_sig++;
    }
  }
}

static int scan_idx_for(int intra, int log2, int cidx, int mode)
{
  if (!intra) return 0;
  if (log2 == 2 || (log2 == 3 && cidx == 0)) {
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
  }
  return 0;
}

static void enc_mvd(orc_cabac_enc *c, int dx, int dy)
{
  int ax = orc_abs(dx), ay = orc_abs(dy);
  orc_cenc_bin(c, CTX_MVD_GT0, ax > 0); orc_cenc_bin(c, CTX_MVD_GT0, ay > 0);
  if (ax > 0) orc_cenc_bin(c, CTX_MVD_GT1, ax > 1);
  if (ay > 0) orc_cenc_bin(c, CTX_MVD_GT1, ay > 1);
  for (int k = 0; k < 2; k++) {
    int a = k ? ay : ax, s = (k ? dy : dx) < 0;
    if (a == 0) continue;
    if (a > 1) {   /* abs_mvd_minus2, EG1 */
      int x = a - 2, kk = 1;
      while (x >= (1 << kk)) { orc_cenc_bypass(c, 1); x -= 1 << kk; kk++; }
      orc_cenc_bypass(c, 0);
      orc_cenc_bypass_bits(c, (uint32_t)x, kk);
    }
    orc_cenc_bypass(c, s);
  }
}

static void enc_cu(orc_encoder *e, orc_cabac_enc *c, int x0, int y0, int log2)
{
  orc_pic *p = e->cur;
  int bi = b8i(e, x0, y0), n = 1 << log2;
  int intra = e->cu_intra[bi], flags = e->cu_flags[bi], cbf = e->cu_cbf[bi];
  if (e->pps.transquant_bypass_enabled) orc_cenc_bin(c, CTX_TQ_BYPASS, e->cfg.lossless ? 1 : 0);      /* cu_transquant_bypass_flag: first in the coding unit (7.3.8.5) */
  if (!e->is_intra) {
    int l = orc_available(&e->av, x0, y0, x0 - 1, y0) && p->pred_mode[(y0 >> 2) * p->b4_w + ((x0 - 1) >> 2)] == MODE_SKIP;
    int a = orc_available(&e->av, x0, y0, x0, y0 - 1) && p->pred_mode[((y0 - 1) >> 2) * p->b4_w + (x0 >> 2)] == MODE_SKIP;
    orc_cenc_bin(c, CTX_SKIP + l + a, flags & 1);
    if (flags & 1) {
      int idx = e->cu_merge_idx[bi];
      orc_cenc_bin(c, CTX_MERGE_IDX, idx > 0);
      if (idx > 0) for (int i = 1; i < 4; i++) { orc_cenc_bypass(c, idx > i); if (idx <= i) break; }
      return;
    }
    orc_cenc_bin(c, CTX_PRED_MODE, intra);
  }
  if (!intra || log2 == 3) orc_cenc_bin(c, CTX_PART_MODE, 1);      /* PART_2Nx2N */
  int mode = 0;
  if (intra) {
    mode = e->cu_intra_mode[bi];
    int ca = 1, cb = 1;
    if (orc_available(&e->av, x0, y0, x0 - 1, y0) && p->pred_mode[(y0 >> 2) * p->b4_w + ((x0 - 1) >> 2)] == MODE_INTRA)
      ca = p->intra_mode[(y0 >> 2) * p->b4_w + ((x0 - 1) >> 2)];
    if (orc_available(&e->av, x0, y0, x0, y0 - 1) && p->pred_mode[((y0 - 1) >> 2) * p->b4_w + (x0 >> 2)] == MODE_INTRA && (y0 - 1) >= ((y0 >> 6) << 6))
      cb = p->intra_mode[((y0 - 1) >> 2) * p->b4_w + (x0 >> 2)];
    int cand[3];
    if (ca == cb) {
      if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
      else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
    } else {
      cand[0] = ca; cand[1] = cb;
      if (ca != 0 && cb != 0) cand[2] = 0; else if (ca != 1 && cb != 1) cand[2] = 1; else cand[2] = 26;
    }
    int mpm = -1;
    for (int k = 0; k < 3; k++) if (cand[k] == mode) { mpm = k; break; }
    orc_cenc_bin(c, CTX_PREV_INTRA, mpm >= 0);
    if (mpm >= 0) { orc_cenc_bypass(c, mpm > 0); if (mpm > 0) orc_cenc_bypass(c, mpm > 1); }
    else {
      if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
      if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
      if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
      int rem = mode;
      for (int k = 2; k >= 0; k--) if (rem > cand[k]) rem--;
      orc_cenc_bypass_bits(c, (uint32_t)rem, 5);
    }
    orc_cenc_bin(c, CTX_CHROMA_MODE, 0);                             /* intra_chroma_pred_mode = 4 (DM) */
  } else {
    orc_cenc_bin(c, CTX_MERGE_FLAG, (flags >> 1) & 1);
    if (flags & 2) {
      int idx = e->cu_merge_idx[bi];
      orc_cenc_bin(c, CTX_MERGE_IDX, idx > 0);
      if (idx > 0) for (int i = 1; i < 4; i++) { orc_cenc_bypass(c, idx > i); if (idx <= i) break; }
    } else {
      enc_mvd(c, e->cu_mvd[bi * 2], e->cu_mvd[bi * 2 + 1]);
      orc_cenc_bin(c, CTX_MVP_FLAG, e->cu_mvp_idx[bi]);
      orc_cenc_bin(c, CTX_RQT_ROOT_CBF, cbf != 0);
    }
    if (!cbf) return;
  }
  /* transform_tree at depth 0, no split: cbf_cb, cbf_cr, [cbf_luma], residuals */
  orc_cenc_bin(c, CTX_CBF_CHROMA + 0, (cbf >> 1) & 1);
  orc_cenc_bin(c, CTX_CBF_CHROMA + 0, (cbf >> 2) & 1);
  if (intra || (cbf & 6)) orc_cenc_bin(c, CTX_CBF_LUMA + 1, cbf & 1);
  if (e->cfg.qp_in_cu && cbf) {                      /* transform_unit(): cu_qp_delta once per quantisation group (= CTU), in its first TU with a coded block */
    int ctu = (y0 >> 6) * (e->cw / 64) + (x0 >> 6);
    int z = 0; for (int b = 0; b < 3; b++) z |= ((((x0 & 63) >> 3) >> b) & 1) << (2 * b) | ((((y0 & 63) >> 3) >> b) & 1) << (2 * b + 1);
    if (e->ctu_first[ctu] == z) {
      int d = e->ctu_delta[ctu], a = orc_abs(d), v = 0;
      while (v < 5 && v < a) { orc_cenc_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0), 1); v++; }      /* prefix: truncated unary, cMax 5 */
      if (a < 5) orc_cenc_bin(c, CTX_CU_QP_DELTA + (a ? 1 : 0), 0);
      else { int x = a - 5, k = 0; while (x >= (1 << k)) { orc_cenc_bypass(c, 1); x -= 1 << k; k++; } orc_cenc_bypass(c, 0); orc_cenc_bypass_bits(c, (uint32_t)x, k); }   /* suffix EG0 */
      if (a) orc_cenc_bypass(c, d < 0);
    }
  }
  if (cbf & 1) enc_residual(c, e->coef[0] + y0 * e->cw + x0, e->cw, log2, 0, scan_idx_for(intra, log2, 0, mode), e->cfg.signhide);
  for (int ci = 1; ci <= 2; ci++)
    if ((cbf >> ci) & 1)
      enc_residual(c, e->coef[ci] + (y0 / 2) * (e->cw / 2) + x0 / 2, e->cw / 2, log2 - 1, ci, scan_idx_for(intra, log2 - 1, ci, mode), e->cfg.signhide);
  (void)n;
}

static void enc_quadtree(orc_encoder *e, orc_cabac_enc *c, int x0, int y0, int log2, int depth)
{
  orc_pic *p = e->cur;
  int cl = e->cu_log2[b8i(e, x0, y0)];
  int split = cl < log2;
  if (log2 > 3) {
    int l = orc_available(&e->av, x0, y0, x0 - 1, y0) && p->ct_depth[(y0 >> 2) * p->b4_w + ((x0 - 1) >> 2)] > depth;
    int a = orc_available(&e->av, x0, y0, x0, y0 - 1) && p->ct_depth[((y0 - 1) >> 2) * p->b4_w + (x0 >> 2)] > depth;
    orc_cenc_bin(c, CTX_SPLIT_CU + l + a, split);
  }
  if (split) {
    int h = 1 << (log2 - 1);
    enc_quadtree(e, c, x0, y0, log2 - 1, depth + 1); enc_quadtree(e, c, x0 + h, y0, log2 - 1, depth + 1);
    enc_quadtree(e, c, x0, y0 + h, log2 - 1, depth + 1); enc_quadtree(e, c, x0 + h, y0 + h, log2 - 1, depth + 1);
  } else enc_cu(e, c, x0, y0, log2);
}

static void write_picture(orc_encoder *e, int write_ps)
{
  orc_bitw ps, hdr, *rows;
  int wc = e->cw / 64, hc = e->ch / 64;
  int nal = e->is_intra ? NAL_IDR_W_RADL : NAL_TRAIL_R;
  e->au.len = 0; e->au.nbits = 0; e->au.cur = 0;
  if (write_ps) {
    orc_bw_init(&ps); orc_write_vps(&ps, &e->vps, &e->sps); orc_write_nal(&e->au, NAL_VPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
    orc_bw_init(&ps); orc_write_sps(&ps, &e->sps); orc_write_nal(&e->au, NAL_SPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
    orc_bw_init(&ps); orc_write_pps(&ps, &e->pps); orc_write_nal(&e->au, NAL_PPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
  }
  /* slice data: tile after tile (6.5.1), the CTBs of a tile in raster order; one substream per tile, with WPP one per CTB row of a tile */
  const int ncols = e->cfg.tile_cols, ntr = e->cfg.tile_rows;
  int nsub = (e->cfg.wpp ? hc : ntr) * ncols;
  rows = (orc_bitw *)calloc((size_t)nsub, sizeof(orc_bitw));
  int *sub_tile_first = (int *)calloc((size_t)nsub + 1, sizeof(int)), *sub_addr = (int *)calloc((size_t)nsub + 1, sizeof(int));   /* substream starts a tile; its first CTB */
  orc_cabac_enc c; memset(&c, 0, sizeof(c));
  orc_ctx saved[CTX_COUNT];
  int init_type = e->is_intra ? 0 : 1, sub = -1;
  for (int tr = 0; tr < ntr; tr++) for (int tc = 0; tc < ncols; tc++) {
    const int cx0 = e->tile_col_bd[tc], cx1 = e->tile_col_bd[tc + 1];
    for (int cy = e->tile_row_bd[tr]; cy < e->tile_row_bd[tr + 1]; cy++) {
      const int tile_start = cy == e->tile_row_bd[tr], tile_end = cy + 1 == e->tile_row_bd[tr + 1];
      if (tile_start || e->cfg.wpp) {
        sub++;
        sub_tile_first[sub] = tile_start; sub_addr[sub] = cy * wc + cx0;
        orc_bw_init(&rows[sub]);
        orc_cenc_start(&c, &rows[sub]);
        if (tile_start || cx1 - cx0 < 2) orc_cabac_init_contexts(c.ctx, init_type, e->qp);   /* 9.3.1: first CTB of a tile (or no second CTB above to take over from) */
        else memcpy(c.ctx, saved, sizeof(saved));       /* WPP: state after the 2nd CTU of the row above inside the tile */
      }
      for (int cx = cx0; cx < cx1; cx++) {
        if (e->cfg.sao) orc_sao_write(&c, &e->sao[cy * wc + cx], cx > cx0 ? &e->sao[cy * wc + cx - 1] : NULL,
                                      (cy > 0 && !tile_start) ? &e->sao[(cy - 1) * wc + cx] : NULL, 1, 1);
        enc_quadtree(e, &c, cx * 64, cy * 64, 6, 0);
        if (e->cfg.wpp && cx == cx0 + 1) memcpy(saved, c.ctx, sizeof(saved));
        int last = (tr == ntr - 1 && tc == ncols - 1 && tile_end && cx == cx1 - 1);
        int sub_end = cx == cx1 - 1 && (e->cfg.wpp || tile_end);
        int seg_end = last || (cx == cx1 - 1 && (e->cfg.slices == 1 || (e->cfg.slices == 2 && tile_end)));
        orc_cenc_terminate(&c, seg_end);                /* end_of_slice_segment_flag */
        if (!seg_end && sub_end) orc_cenc_terminate(&c, 1);   /* end_of_subset_one_bit */
        if (seg_end || sub_end) orc_bw_align_zero(c.bw);
      }
    }
  }
  e->bins = c.bins;
  orc_slice_hdr sh; memset(&sh, 0, sizeof(sh));
  sh.slice_type = e->is_intra ? SLICE_I : SLICE_P; sh.pic_output_flag = 1;
  sh.slice_qp_delta = e->qp - e->cfg.qp;            /* PPS init_qp is the configured QP */
  sh.poc_lsb = e->poc & 255; sh.short_term_ref_pic_set_sps_flag = 1;
  sh.num_ref_idx_l0 = 1; sh.num_ref_idx_l1 = 1; sh.max_num_merge_cand = 5; sh.collocated_from_l0 = 1;
  sh.slice_deblocking_disabled = !e->cfg.deblock;
  sh.loop_filter_across_slices = 1;
  sh.sao_luma = sh.sao_chroma = e->cfg.sao ? 1 : 0;
  /* slice segments, one NAL unit each: the whole picture; or a dependent segment per CTU row (its header: address and entry points
   * only); or an independent slice per tile (the same header with its own address) */
  for (int s0 = 0; s0 < nsub; ) {
    int n = nsub - s0, addr = sub_addr[s0];
    if (e->cfg.slices == 1) n = 1;                                                        /* (wpp, one tile column: substream = CTU row) */
    if (e->cfg.slices == 2) { n = 1; while (s0 + n < nsub && !sub_tile_first[s0 + n]) n++; }   /* the tile's substreams */
    uint32_t *ep = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
    sh.first_slice_segment_in_pic = s0 == 0; sh.dependent_slice_segment = (s0 > 0 && e->cfg.slices == 1); sh.slice_segment_address = addr;
    sh.num_entry_points = n - 1; sh.entry_point_offset = ep;
    for (int i = 0; i < n - 1; i++) ep[i] = (uint32_t)orc_escaped_size(rows[s0 + i].buf, rows[s0 + i].len);
    orc_bw_init(&hdr);
    orc_write_slice_header(&hdr, &sh, &e->sps, &e->pps, nal);
    for (int i = 0; i < n; i++) { orc_bw_bytes(&hdr, rows[s0 + i].buf, rows[s0 + i].len); orc_bw_free(&rows[s0 + i]); }
    orc_write_nal(&e->au, nal, 0, hdr.buf, hdr.len, 1);
    orc_bw_free(&hdr); free(ep);
    s0 += n;
  }
  free(rows); free(sub_tile_first); free(sub_addr);
}

/* ------------------------------------------------------------------ top level */
/* "uvgx rate control v1": picture-level, deterministic, with a fixed feedback delay of three pictures so that an
 * encoder that pipelines pictures (owf <= 2) decides exactly like one that does not.  T = bits per picture at the
 * target rate.  Before picture t (t >= 3) the size of access unit t - 3 is booked: debt += bits(t-3) - T.  The QP
 * moves one step (two when |debt| > 16 T) towards paying the debt back, but only while the booked picture was still
 * on the wrong side of T -- which damps the oscillation the delay would otherwise cause.  QP stays in [10, 51]. */
static void rate_control(orc_encoder *e)
{
  const int D = e->cfg.rc_delay >= 3 && e->cfg.rc_delay <= 7 ? e->cfg.rc_delay : 3;      /* (the delay: three pictures unless "rc-delay" says 4 .. 7 -- an encoder with more pictures in flight) */
  if (e->cfg.bitrate <= 0 || e->frame_idx < D) return;
  const int64_t T = ((int64_t)e->cfg.bitrate * e->cfg.fps_den) / (e->cfg.fps_num > 0 ? e->cfg.fps_num : 1);
  const int64_t trend = (int64_t)8 * e->rc_bytes[(e->frame_idx - D) & 7] - T;
  e->rc_debt += trend;
  int step = 0;
  if (e->rc_debt > 4 * T && trend > 0) step = e->rc_debt > 16 * T ? 2 : 1;
  if (e->rc_debt < -4 * T && trend < 0) step = e->rc_debt < -16 * T ? -2 : -1;
  e->qp = orc_clip3(10, 51, e->qp + step);
}

void orc_enc_set_roi(orc_encoder *e, int w, int h, const int8_t *map)
{
  free(e->roi); e->roi = NULL; e->roi_w = e->roi_h = 0;
  if (w > 0 && h > 0 && map) { e->roi = (int8_t *)malloc((size_t)w * h); memcpy(e->roi, map, (size_t)w * h); e->roi_w = w; e->roi_h = h; }
}

/* "uvgx VAQ v1" (kvazaar "vaq", uvgComm video/VAQ 1..20, kvazaarfilter.cpp:280-284).  Per CTU, over the 64x64 luma samples of the
 * padded source: S1 = sum x, S2 = sum x^2, variance v = (4096 S2 - S1^2) >> 24.  Activity e = 16 floor(log2(v + 1)) + the next four
 * bits of v + 1 below its leading one (log2 in 1/16 steps, piecewise linear).  With E = round(mean of e over the picture's CTUs):
 * delta = trunc(vaq * (e - E) / 48) towards zero, later clamped together with the ROI delta to [-12, 12]. */
static int vaq_activity(const pixel *p, int stride)
{
  uint32_t s1 = 0, s2 = 0;
  for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) { uint32_t v = p[y * stride + x]; s1 += v; s2 += v * v; }
  uint64_t var = ((uint64_t)s2 * 4096 - (uint64_t)s1 * s1) >> 24;
  uint32_t v1 = (uint32_t)var + 1;
  int l = orc_log2(v1);
  return 16 * l + (int)(((v1 << 4) >> l) & 15);
}
static void vaq_deltas(orc_encoder *e, int *delta)
{
  int wc = e->cw / 64, hc = e->ch / 64, n = wc * hc;
  int64_t sum = 0;
  for (int cy = 0; cy < hc; cy++) for (int cx = 0; cx < wc; cx++) { delta[cy * wc + cx] = vaq_activity(e->src[0] + (size_t)cy * 64 * e->cw + cx * 64, e->cw); sum += delta[cy * wc + cx]; }
  int mean = (int)((sum + n / 2) / n);
  for (int i = 0; i < n; i++) { int d = e->cfg.vaq * (delta[i] - mean); delta[i] = d >= 0 ? d / 48 : -((-d) / 48); }
}

/* target QP of every CTU for this picture (after the source picture has been loaded) */
static void roi_targets(orc_encoder *e)
{
  int wc = e->cw / 64, hc = e->ch / 64;
  int *vd = NULL;
  if (e->cfg.vaq > 0) { vd = (int *)calloc((size_t)wc * hc, sizeof(int)); vaq_deltas(e, vd); }
  for (int cy = 0; cy < hc; cy++) for (int cx = 0; cx < wc; cx++) {
    int d = 0;
    if (e->cfg.qp_in_cu && e->roi) d = e->roi[(cy * e->roi_h / hc) * e->roi_w + (cx * e->roi_w / wc)];
    d = orc_clip3(-12, 12, d);
    if (vd) d = orc_clip3(-12, 12, d + vd[cy * wc + cx]);
    e->ctu_qt[cy * wc + cx] = (int8_t)orc_clip3(0, 51, e->qp + d);
  }
  free(vd);
}

/* After reconstruction: which CTUs code a delta, their actual QpY (8.6.1 with quantisation group = CTU: the prediction is the
 * QpY of the previous CTU in decoding order, or the slice QP at the start of a slice, a tile or -- with WPP -- a CTU row). */
static void roi_resolve(orc_encoder *e)
{
  orc_pic *p = e->cur;
  int wc = e->cw / 64, prev = e->qp;
  for (int tr = 0; tr < e->cfg.tile_rows; tr++) for (int tc = 0; tc < e->cfg.tile_cols; tc++)
  for (int cy = e->tile_row_bd[tr]; cy < e->tile_row_bd[tr + 1]; cy++) {
    int tile_start = cy == e->tile_row_bd[tr];
    if (e->cfg.wpp || tile_start) prev = e->qp;
    for (int cx = e->tile_col_bd[tc]; cx < e->tile_col_bd[tc + 1]; cx++) {
      int ctu = cy * wc + cx, first = 64;
      for (int z = 63; z >= 0; z--) {
        int xi = 0, yi = 0; for (int b = 0; b < 3; b++) { xi |= ((z >> (2 * b)) & 1) << b; yi |= ((z >> (2 * b + 1)) & 1) << b; }
        int x = cx * 64 + xi * 8, y = cy * 64 + yi * 8, bi = b8i(e, x, y), n = 1 << e->cu_log2[bi];
        if ((x & (n - 1)) || (y & (n - 1))) continue;          /* not a CU origin */
        if (e->cu_cbf[bi]) first = z;
      }
      int qy = (first < 64 && e->cfg.qp_in_cu) ? e->ctu_qt[ctu] : prev;
      e->ctu_first[ctu] = (uint8_t)first; e->ctu_qy[ctu] = (int8_t)qy; e->ctu_delta[ctu] = (int8_t)(qy - prev);
      prev = qy;
      /* CUs decoded before the delta arrives keep the predicted QP (CuQpDeltaVal is still 0 for them, 8.6.1) */
      for (int z = 0; z < 64; z++) {
        int xi = 0, yi = 0; for (int b = 0; b < 3; b++) { xi |= ((z >> (2 * b)) & 1) << b; yi |= ((z >> (2 * b + 1)) & 1) << b; }
        int q = (z >= first) ? qy : (qy - e->ctu_delta[ctu]);
        for (int y = cy * 64 + yi * 8; y < cy * 64 + yi * 8 + 8; y += 4) for (int x = cx * 64 + xi * 8; x < cx * 64 + xi * 8 + 8; x += 4)
          p->qp_y[(y >> 2) * p->b4_w + (x >> 2)] = (int8_t)q;
      }
    }
  }
}

size_t orc_enc_encode(orc_encoder *e, const pixel *y, const pixel *u, const pixel *v, const uint8_t **au)
{
  int period = e->cfg.intra_period;
  e->is_intra = (e->frame_idx == 0) || (period > 0 && (e->frame_idx % period) == 0);
  if (e->is_intra) e->poc = 0; else e->poc++;
  rate_control(e);
  rc_picture_start(e);
  load_input(e, y, u, v);
  roi_targets(e);
  orc_pic_reset_side(e->cur);
  for (int i = 0; i < 16; i++) e->cur->ref_poc_list[i] = e->poc - 1;       /* one reference picture */
  for (int c = 0; c < 3; c++) memset(e->coef[c], 0, sizeof(int16_t) * (size_t)(c ? e->cw * e->ch / 4 : e->cw * e->ch));
  if (e->is_intra) encode_intra_picture(e); else encode_inter_picture(e);
  roi_resolve(e);
  for (int c = 0; c < 3; c++) memcpy(e->predeblock[c], e->cur->plane[c], (size_t)(c ? e->cw * e->ch / 4 : e->cw * e->ch));
  orc_compute_bs(e->cur, e->bs_v, e->bs_h);
  if (e->cfg.deblock) {
    orc_deblock_ctx db; memset(&db, 0, sizeof(db));
    db.w = e->cw; db.h = e->ch;
    for (int i = 0; i < 3; i++) { db.plane[i] = e->cur->plane[i]; db.stride[i] = e->cur->stride[i]; }
    db.bs_v = e->bs_v; db.bs_stride_v = e->cw / 8; db.bs_h = e->bs_h; db.bs_stride_h = e->cw / 4;
    db.qp_y = e->cur->qp_y; db.qp_stride = e->cur->b4_w;
    orc_deblock_picture(&db);
  }
  if (e->cfg.sao) {                                   /* 8.7.3 on the deblocked picture, parameters from the decision below */
    int wc = e->cw / 64, hc = e->ch / 64;
    const pixel *deb[3], *org[3];
    for (int i = 0; i < 3; i++) { memcpy(e->sao_in[i], e->cur->plane[i], (size_t)e->cur->stride[i] * (i ? e->ch / 2 : e->ch)); deb[i] = e->sao_in[i]; org[i] = e->src[i]; }
    if (e->cur->stride[0] != e->cw || e->cur->stride[1] != e->cw / 2) abort();
    for (int cy = 0; cy < hc; cy++) for (int cx = 0; cx < wc; cx++)
      orc_sao_decide_ctu(deb, org, e->cur->stride, e->cw, e->ch, cx, cy, orc_lambda_q4[e->qp], &e->sao[cy * wc + cx]);
    orc_sao_ctx sc; memset(&sc, 0, sizeof(sc));
    sc.w = e->cw; sc.h = e->ch; sc.ctb_log2 = 6; sc.pic_w_ctbs = wc; sc.params = e->sao; sc.across_slices = sc.across_tiles = 1;
    for (int i = 0; i < 3; i++) { sc.src[i] = e->sao_in[i]; sc.dst[i] = e->cur->plane[i]; sc.stride[i] = e->cur->stride[i]; }
    orc_sao_picture(&sc);
  }
  int write_ps = 0;
  if (e->is_intra) {
    write_ps = (e->intra_count == 0) || (e->cfg.vps_period > 0 && (e->intra_count % e->cfg.vps_period) == 0);
    e->intra_count++;
  }
  write_picture(e, write_ps);
  if (e->cfg.hash) {
    /* decoded picture hash SEI (D.2.19): suffix SEI NAL unit (type 40) behind the picture's last slice segment; payload type 132, then
     * hash_type (0 MD5, 2 checksum) and one hash per colour component over the whole decoded picture (coded size, after the loop filters) */
    const int type = e->cfg.hash == 2 ? 0 : 2, nb = orc_hash_bytes(type);
    uint8_t hv[3][16];
    const pixel *pl[3] = { e->cur->plane[0], e->cur->plane[1], e->cur->plane[2] };
    orc_picture_hash(type, pl, e->cur->stride, e->cw, e->ch, hv);
    orc_bitw sei; orc_bw_init(&sei);
    orc_bw_put(&sei, 132, 8); orc_bw_put(&sei, (uint32_t)(1 + 3 * nb), 8); orc_bw_put(&sei, (uint32_t)type, 8);
    for (int c = 0; c < 3; c++) for (int i = 0; i < nb; i++) orc_bw_put(&sei, hv[c][i], 8);
    orc_bw_trailing(&sei);
    orc_write_nal(&e->au, 40, 0, sei.buf, sei.len, 1);
    orc_bw_free(&sei);
  }
  e->rc_bytes[e->frame_idx & 7] = (uint32_t)e->au.len;
  e->frame_idx++;
  orc_pic *t = e->cur; e->cur = e->ref; e->ref = t;     /* e->ref now holds the picture just coded */
  if (e->cfg.me_source) {                               /* the next picture's search looks at this input picture */
    if (!e->prev_src) e->prev_src = (pixel *)malloc((size_t)e->cw * e->ch);
    memcpy(e->prev_src, e->src[0], (size_t)e->cw * e->ch);
  }
  *au = e->au.buf;
  return e->au.len;
}

int orc_enc_set_option(orc_encoder *e, const char *name, int value)
{
  if (!strcmp(name, "hash")) { e->cfg.hash = value; return 1; }
  if (!strcmp(name, "intra-in-p")) { e->cfg.intra_in_p = value < 0 ? 0 : (value > 2 ? 2 : value); return 1; }      /* 0 off, 1: 16x16 intra units in P pictures, 2: 16x16 and 8x8 */
  if (!strcmp(name, "rc-delay")) { if (value < 3 || value > 7) return 0; e->cfg.rc_delay = value; return 1; }
  if (!strcmp(name, "scaling-list")) {                  /* 1: `scaling-list default` -- takes effect with the next parameter sets (set it before the first picture) */
    e->cfg.scaling_list = value != 0; e->sps.scaling_list_enabled = e->cfg.scaling_list; e->sps.scaling_list_data_present = 0;
    orc_scaling_default(&e->sps.scaling);
    for (int sz = 0; sz < 4; sz++) for (int mi = 0; mi < (sz == 3 ? 2 : 6); mi++) orc_scaling_factor(&e->sps.scaling, sz, mi, e->sfac[sz][mi]);
    return 1;
  }
  if (!strcmp(name, "intra-chain")) { e->cfg.intra_chain = value != 0; return 1; }
  if (!strcmp(name, "lossless")) {                     /* (set it before the first picture: the PPS says transquant_bypass_enabled_flag) */
    if (!value) return e->cfg.lossless == 0;
    e->cfg.lossless = 1; e->pps.transquant_bypass_enabled = 1;
    e->cfg.deblock = 0; e->pps.deblocking_filter_control_present = 1; e->pps.pps_deblocking_disabled = 1;
    e->cfg.sao = 0; e->sps.sao_enabled = 0;
    e->cfg.rdoq = 0; e->cfg.signhide = 0; e->pps.sign_data_hiding = 0;
    e->cfg.bitrate = 0; e->cfg.rc_bands = 0;
    return 1;
  }
  if (!strcmp(name, "me-source")) { e->cfg.me_source = value != 0; return 1; }      /* the integer search on the previous input picture (build_refpad) */
  if (!strcmp(name, "rdoq")) { e->cfg.rdoq = value != 0; return 1; }
  if (!strcmp(name, "signhide")) { e->cfg.signhide = value != 0; e->pps.sign_data_hiding = e->cfg.signhide; return 1; }
  return 0;
}

void orc_enc_get_debug(orc_encoder *e, orc_enc_debug *d)
{
  memset(d, 0, sizeof(*d));
  d->coded_w = e->cw; d->coded_h = e->ch; d->is_intra = e->is_intra; d->poc = e->poc;
  d->cu_log2 = e->cu_log2; d->cu_intra = e->cu_intra; d->cu_flags = e->cu_flags; d->cu_merge_idx = e->cu_merge_idx;
  d->cu_mvp_idx = e->cu_mvp_idx; d->cu_intra_mode = e->cu_intra_mode; d->cu_cbf = e->cu_cbf; d->cu_mv = e->cu_mv;
  for (int i = 0; i < 3; i++) { d->coef[i] = e->coef[i]; d->predeblock[i] = e->predeblock[i]; d->recon[i] = e->ref->plane[i]; }
  d->bs_v = e->bs_v; d->bs_h = e->bs_h; d->bins = e->bins;
}

void orc_enc_get_recon(orc_encoder *e, pixel *y, pixel *u, pixel *v)
{
  pixel *out[3] = { y, u, v };
  for (int c = 0; c < 3; c++) {
    int w = c ? e->cfg.width / 2 : e->cfg.width, h = c ? e->cfg.height / 2 : e->cfg.height;
    for (int yy = 0; yy < h; yy++) memcpy(out[c] + (size_t)yy * w, e->ref->plane[c] + (size_t)yy * e->ref->stride[c], (size_t)w);
  }
}
