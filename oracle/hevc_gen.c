/* oracle/hevc_gen.c -- see hevc_gen.h.  Test infrastructure: a conformance-style stream SYNTHESISER.
 *
 * The product's decoder sits behind libOpenHevcDecode, which uvgComm feeds with whatever the remote
 * peer's encoder produced (/root/reference/src/media/processing/openhevcfilter.cpp:134-172) -- normally
 * Kvazaar with gop=lp-g4d3t1 (kvazaarfilter.cpp:233): several reference pictures, TMVP, intra CUs in P
 * pictures, all CU sizes, picture sizes that are multiples of 8 only.  This project's own encoder emits
 * a much narrower tool set, so it cannot exercise those paths.  This file writes syntactically valid
 * Main-profile streams whose every decision is drawn from a seeded generator: random coding quadtrees
 * (with the implicit splits of partial CTUs), prediction modes, partitionings, merge / AMVP data,
 * reference indices, transform trees, residuals (sign hiding, transform skip, escape codes), cu_qp_delta,
 * SAO parameters, deblocking overrides, tiles / WPP substreams.  No source picture is involved: what the
 * pictures look like is whatever the syntax decodes to, and the general CPU decoder (hevc_dec.c) defines
 * the expected result.  The generator only tracks what the SYNTAX depends on (split / skip contexts,
 * intra modes for the MPM list and the scan order, cbf inheritance, IsCuQpDeltaCoded). */
#include "hevc_gen.h"
#include "hevc_bits.h"
#include "hevc_cabac.h"
#include "hevc_ps.h"
#include "hevc_pic.h"
#include "hevc_intra.h"
#include "hevc_sao.h"

struct orc_gen {
  orc_gen_config cfg;
  uint64_t rng;
  orc_vps vps; orc_sps sps; orc_pps pps; orc_slice_hdr sh;
  orc_pic side;                       /* per-4x4 syntax state of the picture being written (planes unused) */
  orc_avail_ctx av; int16_t *ctb_tile; int32_t *ctb_slice;
  int row_bd[34], nrows_t, col_bd[34], ncols_t;
  orc_sao_params *sao;
  orc_cabac_enc c;
  orc_bitw au;
  int frame_idx, poc, since_idr;
  int hist_tid[8], layers, cur_tid;   /* temporal_layers: TemporalId of the pictures in hist_poc; number of sub-layers (0: off); the current picture's */
  int hist_cls[8];                    /* open_gop: 1 = decoded before the last CRA picture, or a RASL picture -- nothing a RADL picture may use */
  int cra_poc, cra_split, cur_nal;    /* open_gop: POC of the last CRA picture (-1: none since the IDR), the POC below which its leading pictures are RASL; the picture's NAL unit type */
  int hist_poc[8], hist_n;            /* POCs of the pictures decoded since the IDR, newest first (the reference picture set is the first num_refs of them) */
  int gop_order[8];                   /* cfg.gop > 1: POC offsets inside a group in decoding order */
  int slice_is_b;
  int lt_alive, lt_marked;            /* long_term: the sequence's first picture (POC 0) is still in the reference picture set / has become a long-term reference picture */
  int slice_is_intra;
  /* coding-unit state */
  int cu_qp_delta_coded, log2_qg;
  int cu_pred_mode, part_mode, intra_split, max_trafo_depth, cu_bypass;
  int intra_modes[4], chroma_mode;
};

/* ------------------------------------------------------------------ random numbers */
static uint32_t rnd(orc_gen *g)
{
  g->rng ^= g->rng << 13; g->rng ^= g->rng >> 7; g->rng ^= g->rng << 17;
  return (uint32_t)(g->rng >> 16);
}
static int rrange(orc_gen *g, int lo, int hi) { return lo + (int)(rnd(g) % (uint32_t)(hi - lo + 1)); }   /* inclusive */
static int rpct(orc_gen *g, int pct) { return (int)(rnd(g) % 100u) < pct; }
/* configuration value: >= 0 as given, -1 = drawn once per stream from [lo, hi] */
static int pick(orc_gen *g, int v, int lo, int hi) { return v >= 0 ? v : rrange(g, lo, hi); }

static inline int b4(const orc_pic *p, int x, int y) { return (y >> 2) * p->b4_w + (x >> 2); }
static void fill4(orc_pic *p, uint8_t *arr, int x0, int y0, int w, int h, int v)
{
  for (int y = y0; y < y0 + h && y < p->h; y += 4)
    for (int x = x0; x < x0 + w && x < p->w; x += 4) arr[b4(p, x, y)] = (uint8_t)v;
}

void orc_gen_default_config(orc_gen_config *c)
{
  memset(c, 0xff, sizeof(*c));             /* every switch -1: drawn from the seed */
  c->width = 416; c->height = 240; c->seed = 1; c->intra_period = 8; c->qp = 30;
  c->density = 30;
}

/* scaling_list_data() with every way of coding a list: the default lists (pred_mode 0, delta 0), a copy of an earlier matrix of the size (delta > 0),
 * explicit entries (a random walk through 1..255 with the occasional jump to an extreme) */
static void gen_scaling(orc_gen *g, orc_scaling_lists *sl, uint8_t pred_mode[4][6], uint8_t pred_delta[4][6])
{
  orc_scaling_default(sl);
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < (s == 3 ? 2 : 6); m++) {
      const int r = rrange(g, 0, 99);
      pred_mode[s][m] = 0; pred_delta[s][m] = 0;
      if (r < 25) continue;                                             /* the default list */
      if (r < 45 && m > 0) { pred_delta[s][m] = (uint8_t)rrange(g, 1, m); continue; }   /* a copy (the parser takes the referenced list as it stands) */
      pred_mode[s][m] = 1;
      int v = rrange(g, 8, 40);
      if (s >= 2) sl->dc[s - 2][m] = (uint8_t)rrange(g, 1, 255);
      for (int i = 0; i < (s == 0 ? 16 : 64); i++) {
        if (rpct(g, 3)) v = rpct(g, 50) ? 1 : 255;
        else v = orc_clip3(1, 255, v + rrange(g, -6, 10));
        sl->list[s][m][i] = (uint8_t)v;
      }
    }
}

static void gen_sps_rps(orc_gen *g, orc_sps *s);
orc_gen *orc_gen_open(const orc_gen_config *cfg)
{
  if (cfg->width < 16 || cfg->height < 16 || (cfg->width & 7) || (cfg->height & 7)) return NULL;
  orc_gen *g = (orc_gen *)calloc(1, sizeof(*g));
  orc_tables_init();
  g->cfg = *cfg;
  g->rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)(uint32_t)cfg->seed * 0xD1342543DE82EF95ull);
  for (int i = 0; i < 8; i++) rnd(g);
  orc_gen_config *c = &g->cfg;
  c->num_refs = pick(g, c->num_refs, 1, 4);
  c->tmvp = pick(g, c->tmvp, 0, 1);
  c->amp = pick(g, c->amp, 0, 1);
  c->sao = pick(g, c->sao, 0, 1);
  c->strong_intra = pick(g, c->strong_intra, 0, 1);
  c->sign_hiding = pick(g, c->sign_hiding, 0, 1);
  c->transform_skip = pick(g, c->transform_skip, 0, 1);
  c->cabac_init = pick(g, c->cabac_init, 0, 1);
  c->wpp = pick(g, c->wpp, 0, 1);
  c->tile_rows = pick(g, c->tile_rows, 1, 3);
  c->th_depth_inter = pick(g, c->th_depth_inter, 0, 2);
  c->th_depth_intra = pick(g, c->th_depth_intra, 0, 2);
  c->qp_delta = pick(g, c->qp_delta, 0, 4);
  c->chroma_qp_offsets = pick(g, c->chroma_qp_offsets, 0, 1);
  c->deblock_mode = pick(g, c->deblock_mode, 0, 3);                 /* 0 default, 1 disabled in the PPS, 2 PPS offsets, 3 slice override */
  c->par_mrg_level = pick(g, c->par_mrg_level, 2, 4);
  c->intra_in_p = pick(g, c->intra_in_p, 0, 40);
  c->all_part_modes = pick(g, c->all_part_modes, 0, 1);
  c->chroma_modes = pick(g, c->chroma_modes, 0, 1);
  c->nxn_intra = pick(g, c->nxn_intra, 0, 1);
  c->max_cu_log2 = pick(g, c->max_cu_log2, 4, 6);
  c->min_cu_log2 = pick(g, c->min_cu_log2, 3, 4);
  if (c->min_cu_log2 > c->max_cu_log2) c->min_cu_log2 = c->max_cu_log2;
  c->uniform_tiles = pick(g, c->uniform_tiles, 0, 1);
  c->big_mvd = pick(g, c->big_mvd, 0, 1);
  if (c->tq_bypass < 0) c->tq_bypass = 0;
  if (c->scaling_lists < 0 || c->scaling_lists > 4) c->scaling_lists = 0;
  if (c->b_slices < 0) c->b_slices = 0;
  if (c->weighted < 0) c->weighted = 0;
  if (c->list_mod < 0) c->list_mod = 0;
  if (c->gop != 2 && c->gop != 4 && c->gop != 8) c->gop = 0;
  if (c->ctb_log2 != 4 && c->ctb_log2 != 5) c->ctb_log2 = 6;           /* (never drawn: -1 is 64, the streams of the earlier rounds stay what they were) */
  if (c->max_cu_log2 > c->ctb_log2) c->max_cu_log2 = c->ctb_log2;
  if (c->min_cu_log2 > c->max_cu_log2) c->min_cu_log2 = c->max_cu_log2;
  if (c->min_cb_log2 != 4 && c->min_cb_log2 != 5) c->min_cb_log2 = 3;   /* (never drawn) */
  if (c->min_cb_log2 > c->ctb_log2 || (cfg->width & ((1 << c->min_cb_log2) - 1)) || (cfg->height & ((1 << c->min_cb_log2) - 1))) c->min_cb_log2 = 3;
  if (c->min_cu_log2 < c->min_cb_log2) c->min_cu_log2 = c->min_cb_log2;
  if (c->max_cu_log2 < c->min_cu_log2) c->max_cu_log2 = c->min_cu_log2;
  if (c->qp_delta - 1 > c->ctb_log2 - c->min_cb_log2) c->qp_delta = c->ctb_log2 - c->min_cb_log2 + 1;      /* (diff_cu_qp_delta_depth <= log2_diff_max_min_luma_coding_block_size) */
  const int ctbs = 1 << c->ctb_log2, wc = (cfg->width + ctbs - 1) / ctbs, hc = (cfg->height + ctbs - 1) / ctbs;
  if (c->tile_rows > hc) c->tile_rows = hc;
  if (c->tile_rows < 1) c->tile_rows = 1;

  orc_sps *s = &g->sps; memset(s, 0, sizeof(*s));
  s->general_profile_idc = 1; s->general_level_idc = 153;
  s->chroma_format_idc = 1; s->width = cfg->width; s->height = cfg->height;
  s->bit_depth_luma = s->bit_depth_chroma = 8; s->log2_max_poc_lsb = pick(g, -1, 4, 8);
  s->max_dec_pic_buffering = c->num_refs + 1; s->max_num_reorder = 0;
  if (c->gop) {
    /* decoding order of a group: the last picture, then the middle of every interval, depth first */
    int n = 0, lo[8], hi[8], sp = 0;
    g->gop_order[n++] = c->gop; lo[sp] = 0; hi[sp] = c->gop; sp++;
    while (sp) { sp--; const int a = lo[sp], b = hi[sp], m = (a + b) / 2; if (m == a) continue; g->gop_order[n++] = m; lo[sp] = m; hi[sp] = b; sp++; lo[sp] = a; hi[sp] = m; sp++; }
    int lg = 0; while ((1 << lg) < c->gop) lg++;
    s->max_num_reorder = lg; s->max_dec_pic_buffering = c->num_refs + lg + 2 + (c->open_gop == 1);      /* (roomy: output is driven by the reorder count alone) */
    if (s->log2_max_poc_lsb < 6) s->log2_max_poc_lsb = 6;
  }
  s->log2_min_cb = c->min_cb_log2; s->log2_diff_max_min_cb = c->ctb_log2 - c->min_cb_log2; s->log2_min_tb = 2; s->log2_diff_max_min_tb = c->ctb_log2 < 5 ? c->ctb_log2 - 2 : 3;      /* (MaxTbLog2SizeY <= Min(CtbLog2SizeY, 5)) */
  s->max_th_depth_inter = c->th_depth_inter; s->max_th_depth_intra = c->th_depth_intra;
  s->amp_enabled = c->amp; s->sao_enabled = c->sao;
  if (c->pcm > 0) {
    s->pcm_enabled = 1; s->pcm_bit_depth_luma = rrange(g, 1, 8); s->pcm_bit_depth_chroma = rrange(g, 1, 8);
    s->log2_min_pcm_cb = rrange(g, 3, ORC_MIN(5, c->ctb_log2)); s->log2_diff_max_min_pcm_cb = rrange(g, 0, ORC_MIN(5, c->ctb_log2) - s->log2_min_pcm_cb);
    if (s->log2_min_pcm_cb < c->min_cb_log2) { s->log2_min_pcm_cb = c->min_cb_log2; s->log2_diff_max_min_pcm_cb = rrange(g, 0, ORC_MIN(5, c->ctb_log2) - s->log2_min_pcm_cb); }      /* (7.4.3.2.1: Log2MinIpcmCbSizeY in MinCbLog2SizeY .. Min(CtbLog2SizeY, 5)) */
    s->pcm_loop_filter_disabled = rpct(g, 50);
  }
  s->scaling_list_enabled = c->scaling_lists > 0; s->scaling_list_data_present = c->scaling_lists == 2 || c->scaling_lists == 4;
  if (s->scaling_list_data_present) gen_scaling(g, &s->scaling, s->sl_pred_mode, s->sl_pred_delta);
  s->num_st_rps = 0;
  if (c->rps_forms != 1) c->rps_forms = 0;
  if (c->hdr_extras != 1) c->hdr_extras = 0;
  if (c->hdr_extras) s->ext_data = rrange(g, 0, 3);
  if (c->rps_forms) gen_sps_rps(g, s);
  if (c->long_term > 0) {                                   /* candidates in the SPS: POC LSB 0 (the sequence's first picture), used by the current picture or only kept */
    s->long_term_ref_pics_present = 1; s->num_lt_sps = rrange(g, 0, 2);
    for (int i = 0; i < s->num_lt_sps; i++) { s->lt_poc_lsb_sps[i] = 0; s->lt_used_sps[i] = (uint8_t)(i == 0); }
  }
  s->temporal_mvp_enabled = c->tmvp; s->strong_intra_smoothing = c->strong_intra;
  s->vui_present = 1; s->vui_timing_present = 1; s->vui_num_units_in_tick = 1; s->vui_time_scale = 30;
  orc_sps_derive(s);
  orc_vps *v = &g->vps; memset(v, 0, sizeof(*v));
  v->timing_info_present = 1; v->num_units_in_tick = 1; v->time_scale = 30;
  if (c->vui_extras != 1) c->vui_extras = 0;
  if (c->vui_extras) {
    s->vui_extras = (int)(rnd(g) & 0x3ffu);
    switch (rrange(g, 0, 3)) {          /* where the picture rate is said */
      case 0: v->timing_info_present = 0; s->vui_time_scale = 25; break;                              /* the VUI alone */
      case 1: s->vui_timing_present = 0; v->time_scale = 24; break;                                    /* the VPS alone (a VUI without timing) */
      case 2: v->time_scale = 50; s->vui_time_scale = 60000; s->vui_num_units_in_tick = 1001; break;   /* both, and they differ: the VUI's counts */
      default: s->vui_present = 0; v->timing_info_present = 0; break;                                  /* nowhere */
    }
  }
  orc_pps *p = &g->pps; memset(p, 0, sizeof(*p));
  if (c->hdr_extras == 1) { p->num_extra_slice_header_bits = rrange(g, 0, 2); p->slice_header_extension_present = rrange(g, 0, 5); p->ext_data = rrange(g, 0, 3); }
  p->sign_data_hiding = c->sign_hiding; p->cabac_init_present = c->cabac_init; if (c->cip != 1) c->cip = 0; p->constrained_intra_pred = c->cip;
  p->num_ref_idx_l0_default = rrange(g, 1, c->num_refs); p->num_ref_idx_l1_default = 1; p->init_qp = cfg->qp;
  p->transform_skip_enabled = c->transform_skip;
  p->transquant_bypass_enabled = c->tq_bypass > 0;
  p->weighted_pred = p->weighted_bipred = c->weighted > 0;
  p->lists_modification_present = c->list_mod > 0;
  p->scaling_list_data_present = c->scaling_lists >= 3;
  if (p->scaling_list_data_present) gen_scaling(g, &p->scaling, p->sl_pred_mode, p->sl_pred_delta);
  p->cu_qp_delta_enabled = c->qp_delta > 0; p->diff_cu_qp_delta_depth = c->qp_delta > 0 ? c->qp_delta - 1 : 0;
  if (c->chroma_qp_offsets) { p->cb_qp_offset = rrange(g, -4, 4); p->cr_qp_offset = rrange(g, -4, 4); p->slice_chroma_qp_offsets_present = rpct(g, 50); }
  if (c->lf_across < 0 || c->lf_across > 2) c->lf_across = 0;
  if (c->pcm < 0) c->pcm = 0;
  if (c->long_term < 0 || c->gop || c->b_slices > 0) c->long_term = 0;
  if (c->open_gop != 1 || !c->gop) c->open_gop = 0;
  if (c->temporal_layers != 1 || !c->gop) c->temporal_layers = 0;
  if (c->temporal_layers) {
    int lg = 0; while ((1 << lg) < c->gop) lg++;
    g->layers = lg + 1; s->max_sub_layers = lg + 1;
    s->sl_ordering_absent = rpct(g, 50); s->sl_present = (int)(rnd(g) & 0x3fffu);
  }
  if (c->hidden_pics < 0) c->hidden_pics = 0;
  p->output_flag_present = c->hidden_pics > 0;
  p->entropy_coding_sync_enabled = c->wpp; p->loop_filter_across_slices = c->lf_across == 0 ? 1 : (c->lf_across == 2 ? 0 : rpct(g, 70));
  if (c->slices < 0 || c->slices > 3) c->slices = 0;
  if (c->tile_cols < 1) c->tile_cols = 1;
  if (c->tile_cols > wc) c->tile_cols = wc;
  if (c->slices == 3) c->tile_cols = c->tile_rows = 1;                  /* (free slices: one tile) */
  if (c->tile_cols > 1 && c->slices == 1) c->slices = 0;
  p->dependent_slice_segments_enabled = c->slices == 1 || (c->slices == 3 && rpct(g, 60));
  p->num_tile_columns = 1; p->num_tile_rows = 1; p->uniform_spacing = 1;
  p->deblocking_filter_control_present = c->deblock_mode != 0;
  p->pps_deblocking_disabled = c->deblock_mode == 1;
  if (c->deblock_mode >= 2) { p->pps_beta_offset_div2 = rrange(g, -3, 3); p->pps_tc_offset_div2 = rrange(g, -3, 3); }
  p->deblocking_filter_override_enabled = c->deblock_mode == 3;
  p->log2_parallel_merge_level = c->par_mrg_level;
  g->nrows_t = c->tile_rows; g->ncols_t = c->tile_cols;
  for (int i = 0; i <= c->tile_rows; i++) g->row_bd[i] = (i * hc) / c->tile_rows;
  for (int i = 0; i <= c->tile_cols; i++) g->col_bd[i] = (i * wc) / c->tile_cols;
  if (c->tile_rows > 1 || c->tile_cols > 1) {
    p->tiles_enabled = 1; p->num_tile_rows = c->tile_rows; p->num_tile_columns = c->tile_cols; p->loop_filter_across_tiles = c->lf_across == 0 ? 1 : (c->lf_across == 2 ? 0 : rpct(g, 50)); p->uniform_spacing = c->uniform_tiles;
    if (!c->uniform_tiles) for (int i = 0; i < c->tile_cols - 1; i++) p->column_width[i] = g->col_bd[i + 1] - g->col_bd[i];   /* (explicit widths, the uniform values) */
    if (!c->uniform_tiles) {                                /* explicit row heights: a random monotone partition */
      int left = hc;
      for (int i = 0; i < c->tile_rows - 1; i++) {
        int maxh = left - (c->tile_rows - 1 - i), hgt = rrange(g, 1, maxh);
        p->row_height[i] = hgt; g->row_bd[i + 1] = g->row_bd[i] + hgt; left -= hgt;
      }
      g->row_bd[c->tile_rows] = hc;
    }
    g->ctb_tile = (int16_t *)calloc((size_t)wc * hc, sizeof(int16_t));
    for (int i = 0; i < c->tile_rows; i++) for (int cy = g->row_bd[i]; cy < g->row_bd[i + 1]; cy++)
      for (int j = 0; j < c->tile_cols; j++) for (int cx = g->col_bd[j]; cx < g->col_bd[j + 1]; cx++) g->ctb_tile[cy * wc + cx] = (int16_t)(i * c->tile_cols + j);
  }
  if (orc_pic_alloc(&g->side, cfg->width, cfg->height)) { free(g); return NULL; }
  memset(&g->av, 0, sizeof(g->av));
  g->av.pic_w = cfg->width; g->av.pic_h = cfg->height; g->av.ctb_log2 = c->ctb_log2; g->av.pic_w_ctbs = wc; g->av.ctb_tile = g->ctb_tile;
  if (c->slices == 3) { g->ctb_slice = (int32_t *)calloc((size_t)wc * hc, sizeof(int32_t)); g->av.ctb_slice = g->ctb_slice; }
  g->sao = (orc_sao_params *)calloc((size_t)wc * hc, sizeof(orc_sao_params));
  g->log2_qg = c->ctb_log2 - p->diff_cu_qp_delta_depth;
  orc_bw_init(&g->au);
  return g;
}

void orc_gen_close(orc_gen *g)
{
  if (!g) return;
  orc_pic_free(&g->side); free(g->ctb_tile); free(g->ctb_slice); free(g->sao); orc_bw_free(&g->au); free(g);
}
void orc_gen_get_config(const orc_gen *g, orc_gen_config *out) { *out = g->cfg; }

/* ------------------------------------------------------------------ residual_coding, 7.3.8.11 */
static void put_last_prefix(orc_cabac_enc *c, int base, int log2, int cidx, int prefix)
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
static void put_abs_remaining(orc_cabac_enc *c, int v, int rice)
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

static int draw_abs_level(orc_gen *g)
{
  int r = (int)(rnd(g) % 100u);
  if (r < 55) return 1;
  if (r < 75) return 2;
  if (r < 87) return 3;
  if (r < 97) return rrange(g, 4, 24);
  if (r < 99) return rrange(g, 25, 400);
  return rrange(g, 401, 20000);                       /* long escape codes */
}

static void gen_residual(orc_gen *g, int log2, int cidx, int scan_idx)
{
  orc_cabac_enc *c = &g->c;
  const int n = 1 << log2, sb_log2 = log2 - 2, nsb = 1 << sb_log2;
  const uint8_t *sbx = orc_scan_x[scan_idx][sb_log2], *sby = orc_scan_y[scan_idx][sb_log2];
  const uint8_t *px = orc_scan_x[scan_idx][2], *py = orc_scan_y[scan_idx][2];
  uint8_t csbf[8][8]; memset(csbf, 0, sizeof(csbf));
  if (g->pps.transform_skip_enabled && !g->cu_bypass && log2 <= 2) orc_cenc_bin(c, CTX_TS_FLAG + (cidx ? 1 : 0), rpct(g, 30));
  /* the last significant coefficient: mostly in the low-frequency corner */
  int last_sb, last_pos;
  if (rpct(g, 60) || nsb == 1) { last_sb = 0; last_pos = rrange(g, 0, 15); }
  else if (rpct(g, 70)) { last_sb = rrange(g, 0, ORC_MIN(3, nsb * nsb - 1)); last_pos = rrange(g, 0, 15); }
  else { last_sb = rrange(g, 0, nsb * nsb - 1); last_pos = rrange(g, 0, 15); }
  int lx = (sbx[last_sb] << 2) + px[last_pos], ly = (sby[last_sb] << 2) + py[last_pos];
  if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
  int pxv, nbx, sfx, pyv, nby, sfy;
  last_bin(lx, &pxv, &nbx, &sfx); last_bin(ly, &pyv, &nby, &sfy);
  put_last_prefix(c, CTX_LAST_X, log2, cidx, pxv);
  put_last_prefix(c, CTX_LAST_Y, log2, cidx, pyv);
  if (pxv > 3) orc_cenc_bypass_bits(c, (uint32_t)sfx, nbx);
  if (pyv > 3) orc_cenc_bypass_bits(c, (uint32_t)sfy, nby);
  const int dens = g->cfg.density;
  int c1 = 1;
  (void)n;
  for (int i = last_sb; i >= 0; i--) {
    int xs = sbx[i], ys = sby[i], infer_dc = 0;
    int right = (xs < nsb - 1) ? csbf[ys][xs + 1] : 0, below = (ys < nsb - 1) ? csbf[ys + 1][xs] : 0;
    if (i < last_sb && i > 0) {
      csbf[ys][xs] = (uint8_t)rpct(g, 40);
      orc_cenc_bin(c, CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0), csbf[ys][xs]);
      infer_dc = 1;
    } else csbf[ys][xs] = 1;
    if (!csbf[ys][xs]) continue;
    uint8_t sig[16]; memset(sig, 0, sizeof(sig));
    int start = (i == last_sb) ? last_pos - 1 : 15;
    if (i == last_sb) sig[last_pos] = 1;
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
        sig[k] = (uint8_t)rpct(g, dens);
        orc_cenc_bin(c, CTX_SIG + (cidx ? 27 : 0) + sc, sig[k]);
        if (sig[k]) infer_dc = 0;
      } else sig[0] = 1;                                /* inferred: a coded sub-block whose other flags are all zero */
    }
    int nsig = 0; for (int k = 0; k < 16; k++) nsig += sig[k];
    if (!nsig) continue;
    int absv[16]; for (int k = 0; k < 16; k++) absv[k] = sig[k] ? draw_abs_level(g) : 0;
    int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
    if (c1 == 0) ctx_set++;
    c1 = 1;
    int ng1 = 0, last_g1_pos = -1, first_sig = 16, last_sig = -1;
    for (int k = 15; k >= 0; k--) if (sig[k]) {
      if (ng1 < 8) {
        int g1 = absv[k] > 1;
        orc_cenc_bin(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1, g1);
        ng1++;
        if (g1) { c1 = 0; if (last_g1_pos == -1) last_g1_pos = k; }
        else if (c1 > 0 && c1 < 3) c1++;
      }
      if (last_sig == -1) last_sig = k;
      first_sig = k;
    }
    if (last_g1_pos != -1) orc_cenc_bin(c, CTX_GT2 + (cidx ? 4 : 0) + ctx_set, absv[last_g1_pos] > 2);
    int sign_hidden = g->pps.sign_data_hiding && !g->cu_bypass && (last_sig - first_sig > 3);
    for (int k = 15; k >= 0; k--) if (sig[k] && (!sign_hidden || k != first_sig)) orc_cenc_bypass(c, (int)(rnd(g) & 1u));
    int num_sig = 0, rice = 0;
    for (int k = 15; k >= 0; k--) if (sig[k]) {
      int base = (num_sig < 8) ? ((k == last_g1_pos) ? 3 : 2) : 1;
      if (absv[k] >= base) {
        put_abs_remaining(c, absv[k] - base, rice);
        if (absv[k] > 3 * (1 << rice)) rice = ORC_MIN(rice + 1, 4);
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

/* ------------------------------------------------------------------ transform tree, 7.3.8.8-7.3.8.10 */
static void gen_transform_unit(orc_gen *g, int x0, int y0, int log2, int blk, int cbf_luma, int cbf_cb, int cbf_cr, int cbf_cb_parent, int cbf_cr_parent)
{
  orc_cabac_enc *c = &g->c;
  const int intra = g->cu_pred_mode == MODE_INTRA;
  const int chroma_here = log2 > 2, chroma_parent = (log2 == 2 && blk == 3);
  const int ccb = chroma_here ? cbf_cb : (chroma_parent ? cbf_cb_parent : 0);
  const int ccr = chroma_here ? cbf_cr : (chroma_parent ? cbf_cr_parent : 0);
  const int cbf_chroma_any = (log2 > 2) ? (cbf_cb || cbf_cr) : (cbf_cb_parent || cbf_cr_parent);
  if ((cbf_luma || cbf_chroma_any) && g->pps.cu_qp_delta_enabled && !g->cu_qp_delta_coded) {
    int d = rpct(g, 50) ? 0 : (rpct(g, 85) ? rrange(g, -3, 3) : rrange(g, -26, 25));
    int a = orc_abs(d), v = 0;
    while (v < 5 && v < a) { orc_cenc_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0), 1); v++; }
    if (a < 5) orc_cenc_bin(c, CTX_CU_QP_DELTA + (a ? 1 : 0), 0);
    else { int x = a - 5, k = 0; while (x >= (1 << k)) { orc_cenc_bypass(c, 1); x -= 1 << k; k++; } orc_cenc_bypass(c, 0); orc_cenc_bypass_bits(c, (uint32_t)x, k); }
    if (a) orc_cenc_bypass(c, d < 0);
    g->cu_qp_delta_coded = 1;
  }
  const int lmode = intra ? g->side.intra_mode[b4(&g->side, x0, y0)] : 0;
  if (cbf_luma) gen_residual(g, log2, 0, scan_idx_for(intra, log2, 0, lmode));
  if (chroma_here || chroma_parent) {
    const int clog2 = chroma_here ? log2 - 1 : 2;
    if (ccb) gen_residual(g, clog2, 1, scan_idx_for(intra, clog2, 1, g->chroma_mode));
    if (ccr) gen_residual(g, clog2, 2, scan_idx_for(intra, clog2, 2, g->chroma_mode));
  }
}

static void gen_transform_tree(orc_gen *g, int x0, int y0, int log2, int depth, int blk, int cbf_cb_parent, int cbf_cr_parent)
{
  orc_cabac_enc *c = &g->c;
  const orc_sps *s = &g->sps;
  int split;
  if (log2 <= s->log2_max_tb && log2 > s->log2_min_tb && depth < g->max_trafo_depth && !(g->intra_split && depth == 0)) {
    split = rpct(g, 35);
    orc_cenc_bin(c, CTX_SPLIT_TRANSFORM + 5 - log2, split);
  } else {
    int inter_split = (s->max_th_depth_inter == 0 && g->cu_pred_mode == MODE_INTER && g->part_mode != PART_2Nx2N && depth == 0);
    split = (log2 > s->log2_max_tb || (g->intra_split && depth == 0) || inter_split) ? 1 : 0;
  }
  int cbf_cb = 0, cbf_cr = 0;
  if (log2 > 2) {
    if (depth == 0 || cbf_cb_parent) { cbf_cb = rpct(g, 35); orc_cenc_bin(c, CTX_CBF_CHROMA + depth, cbf_cb); }
    if (depth == 0 || cbf_cr_parent) { cbf_cr = rpct(g, 35); orc_cenc_bin(c, CTX_CBF_CHROMA + depth, cbf_cr); }
  } else { cbf_cb = cbf_cb_parent; cbf_cr = cbf_cr_parent; }
  if (split) {
    int h = 1 << (log2 - 1);
    gen_transform_tree(g, x0, y0, log2 - 1, depth + 1, 0, cbf_cb, cbf_cr);
    gen_transform_tree(g, x0 + h, y0, log2 - 1, depth + 1, 1, cbf_cb, cbf_cr);
    gen_transform_tree(g, x0, y0 + h, log2 - 1, depth + 1, 2, cbf_cb, cbf_cr);
    gen_transform_tree(g, x0 + h, y0 + h, log2 - 1, depth + 1, 3, cbf_cb, cbf_cr);
  } else {
    int cbf_luma = 1;
    if (g->cu_pred_mode == MODE_INTRA || depth != 0 || cbf_cb || cbf_cr) {
      cbf_luma = rpct(g, 55);
      orc_cenc_bin(c, CTX_CBF_LUMA + (depth == 0 ? 1 : 0), cbf_luma);
    }
    gen_transform_unit(g, x0, y0, log2, blk, cbf_luma, log2 > 2 ? cbf_cb : 0, log2 > 2 ? cbf_cr : 0, cbf_cb_parent, cbf_cr_parent);
  }
}

/* ------------------------------------------------------------------ prediction unit, 7.3.8.6 / 7.3.8.9 */
static void put_mvd_comp_rest(orc_cabac_enc *c, int a)       /* abs_mvd_minus2 (EG1) for |mvd| > 1 */
{
  int x = a - 2, kk = 1;
  while (x >= (1 << kk)) { orc_cenc_bypass(c, 1); x -= 1 << kk; kk++; }
  orc_cenc_bypass(c, 0);
  orc_cenc_bypass_bits(c, (uint32_t)x, kk);
}
static int draw_mvd(orc_gen *g)
{
  int r = (int)(rnd(g) % 100u), v;
  if (r < 30) v = 0;
  else if (r < 60) v = rrange(g, 1, 6);
  else if (r < 90) v = rrange(g, 1, 40);
  else if (r < 99 || !g->cfg.big_mvd) v = rrange(g, 1, 300);
  else v = rrange(g, 300, 9000);
  return (rnd(g) & 1u) ? -v : v;
}
static void put_merge_idx(orc_gen *g)
{
  orc_cabac_enc *c = &g->c;
  const int mx = g->sh.max_num_merge_cand;
  if (mx <= 1) return;
  int idx = rpct(g, 50) ? 0 : rrange(g, 0, mx - 1);
  orc_cenc_bin(c, CTX_MERGE_IDX, idx > 0);
  if (idx > 0) for (int i = 1; i < mx - 1; i++) { orc_cenc_bypass(c, idx > i); if (idx <= i) break; }
}
static void gen_ref_idx(orc_gen *g, int num_active)
{
  orc_cabac_enc *c = &g->c;
  if (num_active <= 1) return;
  const int mx = num_active - 1, ref = rrange(g, 0, mx);
  for (int i = 0; i < mx; i++) {                           /* truncated Rice, cMax = mx: first two bins context coded */
    const int b = ref > i;
    if (i < 2) orc_cenc_bin(c, CTX_REF_IDX + i, b); else orc_cenc_bypass(c, b);
    if (!b) break;
  }
}
static void gen_mvd(orc_gen *g)
{
  orc_cabac_enc *c = &g->c;
  const int dx = draw_mvd(g), dy = draw_mvd(g), ax = orc_abs(dx), ay = orc_abs(dy);
  orc_cenc_bin(c, CTX_MVD_GT0, ax > 0); orc_cenc_bin(c, CTX_MVD_GT0, ay > 0);
  if (ax > 0) orc_cenc_bin(c, CTX_MVD_GT1, ax > 1);
  if (ay > 0) orc_cenc_bin(c, CTX_MVD_GT1, ay > 1);
  if (ax > 0) { if (ax > 1) put_mvd_comp_rest(c, ax); orc_cenc_bypass(c, dx < 0); }
  if (ay > 0) { if (ay > 1) put_mvd_comp_rest(c, ay); orc_cenc_bypass(c, dy < 0); }
}
/* prediction_unit() of a w x h prediction block in a coding block of quadtree depth ct_depth (7.3.8.6) */
static void gen_prediction_unit(orc_gen *g, int skip, int *merge_out, int w, int h, int ct_depth)
{
  orc_cabac_enc *c = &g->c;
  int merge = 1;
  if (!skip) { merge = rpct(g, 45); orc_cenc_bin(c, CTX_MERGE_FLAG, merge); }
  if (merge_out) *merge_out = merge;
  if (merge) { put_merge_idx(g); return; }
  int idc = 0;                                             /* inter_pred_idc: 0 PRED_L0, 1 PRED_L1, 2 PRED_BI (never for 8x4 / 4x8) */
  if (g->slice_is_b) {
    idc = (w + h != 12 && rpct(g, 40)) ? 2 : (int)(rnd(g) & 1u);
    if (w + h != 12) orc_cenc_bin(c, CTX_INTER_PRED_IDC + ct_depth, idc == 2);
    if (idc != 2) orc_cenc_bin(c, CTX_INTER_PRED_IDC + 4, idc);
  }
  if (idc != 1) { gen_ref_idx(g, g->sh.num_ref_idx_l0); gen_mvd(g); orc_cenc_bin(c, CTX_MVP_FLAG, (int)(rnd(g) & 1u)); }
  if (idc != 0) {
    gen_ref_idx(g, g->sh.num_ref_idx_l1);
    if (!(g->sh.mvd_l1_zero && idc == 2)) gen_mvd(g);
    orc_cenc_bin(c, CTX_MVP_FLAG, (int)(rnd(g) & 1u));
  }
}

/* ------------------------------------------------------------------ coding unit, 7.3.8.5 */
static void gen_coding_unit(orc_gen *g, int x0, int y0, int log2cb, int ct_depth)
{
  orc_cabac_enc *c = &g->c;
  orc_pic *pic = &g->side;
  const orc_sps *s = &g->sps;
  const int n = 1 << log2cb;
  int skip = 0;
  g->cu_bypass = 0;
  if (g->pps.transquant_bypass_enabled) { g->cu_bypass = rpct(g, g->cfg.tq_bypass); orc_cenc_bin(c, CTX_TQ_BYPASS, g->cu_bypass); }      /* 7.3.8.5: first in the coding unit */
  if (!g->slice_is_intra) {
    int l = orc_available(&g->av, x0, y0, x0 - 1, y0) && pic->pred_mode[b4(pic, x0 - 1, y0)] == MODE_SKIP;
    int a = orc_available(&g->av, x0, y0, x0, y0 - 1) && pic->pred_mode[b4(pic, x0, y0 - 1)] == MODE_SKIP;
    skip = rpct(g, 25);
    orc_cenc_bin(c, CTX_SKIP + l + a, skip);
  }
  g->part_mode = PART_2Nx2N; g->intra_split = 0;
  int rqt_root_cbf = 1, merge_2nx2n = 0;
  fill4(pic, pic->ct_depth, x0, y0, n, n, ct_depth);
  if (skip) {
    g->cu_pred_mode = MODE_INTER;
    fill4(pic, pic->pred_mode, x0, y0, n, n, MODE_SKIP);
    gen_prediction_unit(g, 1, NULL, n, n, ct_depth);
    rqt_root_cbf = 0;
  } else {
    g->cu_pred_mode = MODE_INTRA;
    if (!g->slice_is_intra) {
      g->cu_pred_mode = rpct(g, g->cfg.intra_in_p) ? MODE_INTRA : MODE_INTER;
      orc_cenc_bin(c, CTX_PRED_MODE, g->cu_pred_mode == MODE_INTRA);
    }
    if (g->cu_pred_mode != MODE_INTRA || log2cb == s->log2_min_cb) {
      /* part_mode, binarisation 9.3.3.7 */
      if (g->cu_pred_mode == MODE_INTRA) {
        g->part_mode = (g->cfg.nxn_intra && rpct(g, 50)) ? PART_NxN : PART_2Nx2N;
        orc_cenc_bin(c, CTX_PART_MODE, g->part_mode == PART_2Nx2N);
      } else {
        int pm = PART_2Nx2N;
        if (g->cfg.all_part_modes && rpct(g, 50)) {
          if (log2cb == s->log2_min_cb) pm = (log2cb > 3 && rpct(g, 34)) ? PART_NxN : (rpct(g, 50) ? PART_2NxN : PART_Nx2N);      /* (NxN: a minimum coding block above 8 samples) */
          else if (!s->amp_enabled) pm = rpct(g, 50) ? PART_2NxN : PART_Nx2N;
          else { static const int m[6] = { PART_2NxN, PART_Nx2N, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N }; pm = m[rrange(g, 0, 5)]; }
        }
        g->part_mode = pm;
        if (pm == PART_2Nx2N) orc_cenc_bin(c, CTX_PART_MODE, 1);
        else {
          orc_cenc_bin(c, CTX_PART_MODE, 0);
          const int horiz = (pm == PART_2NxN || pm == PART_2NxnU || pm == PART_2NxnD);
          if (log2cb == s->log2_min_cb) {
            orc_cenc_bin(c, CTX_PART_MODE + 1, horiz);             /* log2 == 3: '01' 2NxN, '00' Nx2N; above: '01' 2NxN, '001' Nx2N, '000' NxN */
            if (!horiz && log2cb > 3) orc_cenc_bin(c, CTX_PART_MODE + 2, pm == PART_Nx2N);
          } else if (!s->amp_enabled) orc_cenc_bin(c, CTX_PART_MODE + 1, horiz);
          else {
            orc_cenc_bin(c, CTX_PART_MODE + 1, horiz);
            const int sym = (pm == PART_2NxN || pm == PART_Nx2N);
            orc_cenc_bin(c, CTX_PART_MODE + 3, sym);
            if (!sym) orc_cenc_bypass(c, pm == PART_2NxnD || pm == PART_nRx2N);
          }
        }
      }
    }
    fill4(pic, pic->pred_mode, x0, y0, n, n, g->cu_pred_mode);
    if (g->cu_pred_mode == MODE_INTRA && g->part_mode == PART_2Nx2N && s->pcm_enabled && log2cb >= s->log2_min_pcm_cb && log2cb <= s->log2_min_pcm_cb + s->log2_diff_max_min_pcm_cb) {
      /* pcm_flag (a terminating bin); a PCM unit: the arithmetic codeword ends (its last bit is 1), zero bits to the byte boundary, the samples at their bit depths,
       * and the coder starts again with the contexts as they are (9.3.2.5).  For the syntax around it the unit is an intra unit in DC mode (8.4.2) */
      const int pcm = rpct(g, g->cfg.pcm);
      orc_cenc_terminate(c, pcm);
      if (pcm) {
        orc_bw_align_zero(c->bw);
        for (int i = 0; i < n * n; i++) orc_bw_put(c->bw, rnd(g) & ((1u << s->pcm_bit_depth_luma) - 1), s->pcm_bit_depth_luma);
        for (int i = 0; i < n * n / 2; i++) orc_bw_put(c->bw, rnd(g) & ((1u << s->pcm_bit_depth_chroma) - 1), s->pcm_bit_depth_chroma);
        orc_cenc_start(c, c->bw);
        fill4(pic, pic->intra_mode, x0, y0, n, n, 1);
        return;
      }
    }
    if (g->cu_pred_mode == MODE_INTRA) {
      g->intra_split = (g->part_mode == PART_NxN);
      const int parts = g->intra_split ? 2 : 1, pb = n / parts;
      int prev[4], np = parts * parts;
      for (int k = 0; k < np; k++) { prev[k] = rpct(g, 50); orc_cenc_bin(c, CTX_PREV_INTRA, prev[k]); }
      int k = 0;
      for (int j = 0; j < parts; j++)
        for (int i = 0; i < parts; i++, k++) {
          const int xp = x0 + i * pb, yp = y0 + j * pb;
          int ca = 1, cb = 1;                                   /* 8.4.2 candidate modes */
          if (orc_available(&g->av, xp, yp, xp - 1, yp) && pic->pred_mode[b4(pic, xp - 1, yp)] == MODE_INTRA) ca = pic->intra_mode[b4(pic, xp - 1, yp)];
          if (orc_available(&g->av, xp, yp, xp, yp - 1) && pic->pred_mode[b4(pic, xp, yp - 1)] == MODE_INTRA &&
              (yp - 1) >= ((yp >> s->ctb_log2) << s->ctb_log2)) cb = pic->intra_mode[b4(pic, xp, yp - 1)];
          int cand[3];
          if (ca == cb) {
            if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
            else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
          } else {
            cand[0] = ca; cand[1] = cb;
            if (ca != 0 && cb != 0) cand[2] = 0; else if (ca != 1 && cb != 1) cand[2] = 1; else cand[2] = 26;
          }
          int mode;
          if (prev[k]) {
            const int idx = rrange(g, 0, 2);
            orc_cenc_bypass(c, idx > 0); if (idx > 0) orc_cenc_bypass(c, idx > 1);
            mode = cand[idx];
          } else {
            const int rem = rrange(g, 0, 31);
            orc_cenc_bypass_bits(c, (uint32_t)rem, 5);
            mode = rem;
            if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
            if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
            if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
            for (int q = 0; q < 3; q++) if (mode >= cand[q]) mode++;
          }
          g->intra_modes[k] = mode;
          fill4(pic, pic->intra_mode, xp, yp, pb, pb, mode);
        }
      int icpm = 4;
      if (g->cfg.chroma_modes && rpct(g, 60)) icpm = rrange(g, 0, 3);
      orc_cenc_bin(c, CTX_CHROMA_MODE, icpm != 4);
      if (icpm != 4) orc_cenc_bypass_bits(c, (uint32_t)icpm, 2);
      static const int cm[4] = { 0, 26, 10, 1 };
      if (icpm == 4) g->chroma_mode = g->intra_modes[0];
      else { g->chroma_mode = cm[icpm]; if (g->chroma_mode == g->intra_modes[0]) g->chroma_mode = 34; }
    } else {
      int mf = 0;
      const int npu = (g->part_mode == PART_2Nx2N) ? 1 : (g->part_mode == PART_NxN ? 4 : 2);
      for (int i = 0; i < npu; i++) {
        int pw = n, ph = n;                                /* size of prediction block i */
        switch (g->part_mode) {
          case PART_2NxN: ph = n / 2; break;
          case PART_Nx2N: pw = n / 2; break;
          case PART_NxN: pw = ph = n / 2; break;
          case PART_2NxnU: ph = i == 0 ? n / 4 : n - n / 4; break;
          case PART_2NxnD: ph = i == 0 ? n - n / 4 : n / 4; break;
          case PART_nLx2N: pw = i == 0 ? n / 4 : n - n / 4; break;
          case PART_nRx2N: pw = i == 0 ? n - n / 4 : n / 4; break;
          default: break;
        }
        gen_prediction_unit(g, 0, i == 0 ? &merge_2nx2n : &mf, pw, ph, ct_depth);
      }
      if (!(g->part_mode == PART_2Nx2N && merge_2nx2n)) { rqt_root_cbf = rpct(g, 65); orc_cenc_bin(c, CTX_RQT_ROOT_CBF, rqt_root_cbf); }
    }
  }
  if (rqt_root_cbf) {
    g->max_trafo_depth = (g->cu_pred_mode == MODE_INTRA) ? s->max_th_depth_intra + g->intra_split : s->max_th_depth_inter;
    gen_transform_tree(g, x0, y0, log2cb, 0, 0, 0, 0);
  }
}

static void gen_coding_quadtree(orc_gen *g, int x0, int y0, int log2cb, int depth)
{
  orc_cabac_enc *c = &g->c;
  orc_pic *pic = &g->side;
  const orc_sps *s = &g->sps;
  const int n = 1 << log2cb;
  int split;
  if (x0 + n <= s->width && y0 + n <= s->height && log2cb > s->log2_min_cb) {
    int l = orc_available(&g->av, x0, y0, x0 - 1, y0) && pic->ct_depth[b4(pic, x0 - 1, y0)] > depth;
    int a = orc_available(&g->av, x0, y0, x0, y0 - 1) && pic->ct_depth[b4(pic, x0, y0 - 1)] > depth;
    if (log2cb > g->cfg.max_cu_log2) split = 1;
    else if (log2cb <= g->cfg.min_cu_log2) split = 0;
    else split = rpct(g, log2cb == 6 ? 70 : (log2cb == 5 ? 50 : 40));
    orc_cenc_bin(c, CTX_SPLIT_CU + l + a, split);
  } else split = (log2cb > s->log2_min_cb);
  if (g->pps.cu_qp_delta_enabled && log2cb >= g->log2_qg) g->cu_qp_delta_coded = 0;
  if (split) {
    const int h = n >> 1;
    gen_coding_quadtree(g, x0, y0, log2cb - 1, depth + 1);
    if (x0 + h < s->width) gen_coding_quadtree(g, x0 + h, y0, log2cb - 1, depth + 1);
    if (y0 + h < s->height) gen_coding_quadtree(g, x0, y0 + h, log2cb - 1, depth + 1);
    if (x0 + h < s->width && y0 + h < s->height) gen_coding_quadtree(g, x0 + h, y0 + h, log2cb - 1, depth + 1);
  } else gen_coding_unit(g, x0, y0, log2cb, depth);
}

/* ------------------------------------------------------------------ SAO parameters of one CTU */
static void draw_sao(orc_gen *g, orc_sao_params *p, const orc_sao_params *left, const orc_sao_params *up, int luma, int chroma)
{
  memset(p, 0, sizeof(*p));
  if (left && rpct(g, 25)) { *p = *left; return; }
  if (up && rpct(g, 25)) { *p = *up; return; }
  for (int ci = 0; ci < 3; ci++) {
    if (!(ci ? chroma : luma)) continue;
    if (ci == 2) { p->type[2] = p->type[1]; p->eo_class[2] = p->eo_class[1]; }
    else { p->type[ci] = (uint8_t)(rpct(g, 30) ? 0 : rrange(g, 1, 2)); p->eo_class[ci] = (uint8_t)rrange(g, 0, 3); }
    if (!p->type[ci]) { p->eo_class[ci] = 0; continue; }
    for (int k = 0; k < 4; k++) {
      int a = rpct(g, 40) ? 0 : rrange(g, 1, 7);
      if (p->type[ci] == 1) p->offset[ci][k] = (int8_t)((rnd(g) & 1u) ? -a : a);
      else p->offset[ci][k] = (int8_t)(k < 2 ? a : -a);
    }
    if (p->type[ci] == 1) { p->band_pos[ci] = (uint8_t)rrange(g, 0, 31); p->eo_class[ci] = 0; }
  }
  /* identical to a merge candidate by accident: the writer would code a merge -- fine, same parameters */
}

/* cfg.slices == 3: slice segments that begin at ANY coding tree block (one tile): what encoders that cut slices by bytes or by block counts send (an MTU per
 * slice), with independent slices of whole rows as a special case.  Every cut starts a new arithmetic codeword; an independent slice initialises the contexts,
 * has its own slice_qp_delta and is a new slice for every availability rule (6.4.1: prediction, context selection, merge candidates, SAO merging stop at its
 * border); a dependent segment goes on with the contexts the segment before it ended with -- unless it starts a CTB row under WPP, where 9.3.1 looks at the
 * block above-right first.  With WPP a row start inside a segment is an entry point as ever. */
static void write_free_slices(orc_gen *g, int nal)
{
  const orc_sps *s = &g->sps; const orc_pps *p = &g->pps; orc_slice_hdr *sh = &g->sh;
  const int wc = s->pic_w_ctbs, hc = s->pic_h_ctbs, total = wc * hc, wpp = p->entropy_coding_sync_enabled, ctb = s->ctb_log2;
  const int init_type = g->slice_is_intra ? 0 : (g->slice_is_b ? (sh->cabac_init_flag ? 1 : 2) : (sh->cabac_init_flag ? 2 : 1));
  const int qpd0 = sh->slice_qp_delta;
  orc_bitw *subs = (orc_bitw *)calloc((size_t)total + hc + 1, sizeof(orc_bitw));
  uint8_t *cut = (uint8_t *)calloc((size_t)total + 1, 1);         /* 0 no segment starts at this block, 1 an independent slice, 2 a dependent segment */
  int *seg_first = (int *)calloc((size_t)total + 2, sizeof(int)), *seg_addr = (int *)calloc((size_t)total + 2, sizeof(int)), *seg_qpd = (int *)calloc((size_t)total + 2, sizeof(int));
  int nseg = 0, nsub = 0;
  const int target = rrange(g, 1, 7);                             /* cuts per picture, about; three in ten wait for the next row start */
  cut[0] = 1;
  for (int a = 1, snap = 0; a < total; a++) {
    int c = 0;
    if ((int)(rnd(g) % (uint32_t)total) < target) { if (rpct(g, 30) && a % wc) snap = 1; else c = 1; }
    if (a % wc == 0 && (snap || rpct(g, 6))) { c = 1; snap = 0; }
    if (c) cut[a] = (uint8_t)((p->dependent_slice_segments_enabled && rpct(g, 50)) ? 2 : 1);
  }
  orc_ctx saved[CTX_COUNT];
  int slice_addr = 0, slice_qp = sh->slice_qp, slice_lf = sh->loop_filter_across_slices;
  int *seg_lf = (int *)calloc((size_t)total + 2, sizeof(int));
  orc_pic_reset_side(&g->side);
  memset(&g->c, 0, sizeof(g->c));
  for (int a = 0; a < total; a++) {
    const int cx = a % wc, cy = a / wc;
    if (cut[a]) {
      if (cut[a] == 1) {
        slice_addr = a;
        if (a > 0) { slice_qp = sh->slice_qp + rrange(g, -2, 2); if (slice_qp < 0) slice_qp = 0; if (slice_qp > 51) slice_qp = 51; }
      }
      if (cut[a] == 1 && a > 0 && g->cfg.lf_across == 1 && p->loop_filter_across_slices) slice_lf = rpct(g, 50);
      seg_first[nseg] = nsub; seg_addr[nseg] = a; seg_qpd[nseg] = slice_qp - p->init_qp; seg_lf[nseg] = slice_lf; nseg++;
    }
    g->ctb_slice[a] = slice_addr;
    if (cut[a] || (wpp && cx == 0)) {
      orc_bw_init(&subs[nsub]);
      orc_cenc_start(&g->c, &subs[nsub]);
      nsub++;
      if (cut[a] == 1) orc_cabac_init_contexts(g->c.ctx, init_type, slice_qp);
      else if (wpp && cx == 0) {
        if (orc_available(&g->av, cx << ctb, cy << ctb, (cx + 1) << ctb, (cy - 1) << ctb)) memcpy(g->c.ctx, saved, sizeof(saved));
        else if (!cut[a] || wc < 2) orc_cabac_init_contexts(g->c.ctx, init_type, slice_qp);
        /* (a dependent segment whose above-right block is not available: the contexts the previous segment ended with -- still in g->c.ctx; pictures one block
         * wide initialise, as HM does) */
      }
    }
    if (sh->sao_luma || sh->sao_chroma) {
      orc_sao_params *sp = &g->sao[a];
      const orc_sao_params *left = (cx > 0 && g->ctb_slice[a - 1] == slice_addr) ? sp - 1 : NULL, *up = (cy > 0 && g->ctb_slice[a - wc] == slice_addr) ? sp - wc : NULL;
      draw_sao(g, sp, left, up, sh->sao_luma, sh->sao_chroma);
      orc_sao_write(&g->c, sp, left, up, sh->sao_luma, sh->sao_chroma);
    }
    gen_coding_quadtree(g, cx << ctb, cy << ctb, ctb, 0);
    if (wpp && cx == 1) memcpy(saved, g->c.ctx, sizeof(saved));
    const int seg_end = a + 1 == total || cut[a + 1], sub_end = wpp && cx == wc - 1;
    orc_cenc_terminate(&g->c, seg_end);                           /* end_of_slice_segment_flag */
    if (!seg_end && sub_end) orc_cenc_terminate(&g->c, 1);        /* end_of_subset_one_bit */
    if (seg_end || sub_end) orc_bw_align_zero(g->c.bw);
  }
  seg_first[nseg] = nsub;
  orc_bitw hdr;
  for (int k = 0; k < nseg; k++) {
    const int s0 = seg_first[k], n = seg_first[k + 1] - s0;
    uint32_t *ep = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
    sh->first_slice_segment_in_pic = k == 0; sh->slice_segment_address = seg_addr[k]; sh->dependent_slice_segment = cut[seg_addr[k]] == 2;
    sh->slice_qp_delta = seg_qpd[k]; sh->slice_qp = p->init_qp + seg_qpd[k]; sh->loop_filter_across_slices = seg_lf[k];
    sh->num_entry_points = n - 1; sh->entry_point_offset = ep;
    for (int i = 0; i < n - 1; i++) ep[i] = (uint32_t)orc_escaped_size(subs[s0 + i].buf, subs[s0 + i].len);
    orc_bw_init(&hdr);
    orc_write_slice_header(&hdr, sh, s, p, nal);
    for (int i = 0; i < n; i++) { orc_bw_bytes(&hdr, subs[s0 + i].buf, subs[s0 + i].len); orc_bw_free(&subs[s0 + i]); }
    orc_write_nal(&g->au, nal, g->cur_tid, hdr.buf, hdr.len, 1);
    orc_bw_free(&hdr); free(ep);
  }
  sh->slice_qp_delta = qpd0; sh->slice_qp = p->init_qp + qpd0;
  free(subs); free(cut); free(seg_first); free(seg_addr); free(seg_qpd); free(seg_lf);
  sh->entry_point_offset = NULL;
}

/* ------------------------------------------------------------------ reference picture sets by prediction (7.3.7, 7.4.8)
 * `t` holds a set in its explicit form (nearest first on either side); can it be derived from `ref` moved by d?  Every entry of t must be an entry of ref plus d, or d
 * itself; the flags say which of those are taken and which of the taken ones the picture uses.  On success the coded form is filled in. */
static int rps_try_inter(orc_st_rps *t, const orc_st_rps *ref, int d)
{
  const int nd = ref->num_negative + ref->num_positive;
  int hit = 0;
  if (d == 0 || d < -32768 || d > 32768) return 0;
  for (int j = 0; j <= nd; j++) {
    const int e = (j < ref->num_negative ? ref->delta_poc_s0[j] : j < nd ? ref->delta_poc_s1[j - ref->num_negative] : 0) + d;
    t->used_flag[j] = 0; t->use_delta[j] = 0;
    for (int i = 0; i < t->num_negative; i++) if (t->delta_poc_s0[i] == e) { t->used_flag[j] = (uint8_t)t->used_s0[i]; t->use_delta[j] = 1; hit++; }
    for (int i = 0; i < t->num_positive; i++) if (t->delta_poc_s1[i] == e) { t->used_flag[j] = (uint8_t)t->used_s1[i]; t->use_delta[j] = 1; hit++; }
  }
  if (hit != t->num_negative + t->num_positive) return 0;
  t->inter = 1; t->delta_rps = d; t->nflags = nd + 1;
  return 1;
}
static int rps_same(const orc_st_rps *a, const orc_st_rps *b)
{
  if (a->num_negative != b->num_negative || a->num_positive != b->num_positive) return 0;
  for (int i = 0; i < a->num_negative; i++) if (a->delta_poc_s0[i] != b->delta_poc_s0[i] || a->used_s0[i] != b->used_s0[i]) return 0;
  for (int i = 0; i < a->num_positive; i++) if (a->delta_poc_s1[i] != b->delta_poc_s1[i] || a->used_s1[i] != b->used_s1[i]) return 0;
  return 1;
}
/* the candidate sets of the SPS: the first one explicit, the others explicit or predicted from the one before (in the SPS that is the only choice) */
static void gen_sps_rps(orc_gen *g, orc_sps *s)
{
  const int n = rrange(g, 2, 6);
  memset(s->st_rps, 0, sizeof(s->st_rps[0]) * (size_t)n);
  s->st_rps[0].num_negative = 1; s->st_rps[0].delta_poc_s0[0] = -1; s->st_rps[0].used_s0[0] = 1;
  for (int k = 1; k < n; k++) {
    orc_st_rps *t = &s->st_rps[k]; const orc_st_rps *ref = &s->st_rps[k - 1];
    if (rpct(g, 65)) {
      /* predicted: the entries of the set before moved by d, and d itself, some of them dropped */
      static const int ds[6] = { -1, -1, -2, 1, 2, -4 };
      const int d = ds[rrange(g, 0, 5)], nd = ref->num_negative + ref->num_positive;
      int val[17], m = 0;
      for (int j = 0; j <= nd; j++) {
        const int e = (j < ref->num_negative ? ref->delta_poc_s0[j] : j < nd ? ref->delta_poc_s1[j - ref->num_negative] : 0) + d;
        if (e != 0 && (rpct(g, 80) || (j == nd && m == 0))) val[m++] = e;
      }
      for (int i = 1; i < m; i++) for (int j = i; j > 0 && val[j] > val[j - 1]; j--) { const int x = val[j]; val[j] = val[j - 1]; val[j - 1] = x; }      /* descending */
      for (int i = 0; i < m; i++) if (val[i] < 0) { t->delta_poc_s0[t->num_negative] = val[i]; t->used_s0[t->num_negative++] = rpct(g, 85); }
      for (int i = m - 1; i >= 0; i--) if (val[i] > 0) { t->delta_poc_s1[t->num_positive] = val[i]; t->used_s1[t->num_positive++] = rpct(g, 85); }
      if (t->num_negative + t->num_positive > 0 && t->num_negative + t->num_positive <= 6 && rps_try_inter(t, ref, d)) continue;
      memset(t, 0, sizeof(*t));
    }
    t->num_negative = rrange(g, 1, 3); t->num_positive = rrange(g, 0, 2);
    for (int i = 0, at = 0; i < t->num_negative; i++) { at -= rrange(g, 1, 3); t->delta_poc_s0[i] = at; t->used_s0[i] = rpct(g, 85); }
    for (int i = 0, at = 0; i < t->num_positive; i++) { at += rrange(g, 1, 3); t->delta_poc_s1[i] = at; t->used_s1[i] = rpct(g, 85); }
  }
  s->num_st_rps = n;
}
/* the slice's set is in sh->st_rps, explicit: name a candidate of the SPS that equals it, or predict it from one, where that can be done */
static void choose_rps_form(orc_gen *g, orc_slice_hdr *sh, const orc_sps *s)
{
  orc_st_rps *t = &sh->st_rps;
  if (!g->cfg.rps_forms || s->num_st_rps == 0) return;
  for (int k = 0; k < s->num_st_rps; k++) if (rps_same(t, &s->st_rps[k]) && rpct(g, 80)) { sh->short_term_ref_pic_set_sps_flag = 1; sh->short_term_rps_idx = k; return; }
  if (t->num_negative + t->num_positive == 0 || !rpct(g, 80)) return;
  const int k0 = rrange(g, 0, s->num_st_rps - 1);
  for (int kk = 0; kk < s->num_st_rps; kk++) {
    const int k = (k0 + kk) % s->num_st_rps; const orc_st_rps *ref = &s->st_rps[k];
    const int nd = ref->num_negative + ref->num_positive;
    for (int i = 0; i < t->num_negative + t->num_positive; i++) {
      const int tv = i < t->num_negative ? t->delta_poc_s0[i] : t->delta_poc_s1[i - t->num_negative];
      for (int j = 0; j <= nd; j++) {
        const int rv = j < ref->num_negative ? ref->delta_poc_s0[j] : j < nd ? ref->delta_poc_s1[j - ref->num_negative] : 0;
        if (rps_try_inter(t, ref, tv - rv)) { t->delta_idx_minus1 = s->num_st_rps - 1 - k; return; }
      }
    }
  }
}

/* an SEI message no decoder needs (user_data_unregistered, payload type 5: sixteen bytes of identifier and a few more), and a filler data NAL unit */
static void gen_unknown_sei(orc_gen *g, int nal_type)
{
  uint8_t b[40]; const int n = 16 + rrange(g, 0, 12); int k = 0;
  b[k++] = 5; b[k++] = (uint8_t)n;
  for (int i = 0; i < n; i++) b[k++] = (uint8_t)(rnd(g) & 0xff);
  b[k++] = 0x80;
  orc_write_nal(&g->au, nal_type, g->cur_tid, b, (size_t)k, 1);
}
static void gen_au_tail(orc_gen *g)
{
  if (!g->cfg.hdr_extras) return;
  if (rpct(g, 30)) { uint8_t b[12]; const int n = rrange(g, 1, 10); memset(b, 0xff, (size_t)n); b[n] = 0x80; orc_write_nal(&g->au, NAL_FD, g->cur_tid, b, (size_t)n + 1, 1); }
  if (rpct(g, 40)) gen_unknown_sei(g, NAL_SEI_SUFFIX);
}

/* ------------------------------------------------------------------ one picture */
static void write_picture(orc_gen *g, int idr, int write_ps)
{
  orc_bitw ps, hdr, *subs;
  const orc_sps *s = &g->sps; orc_pps *p = &g->pps; orc_slice_hdr *sh = &g->sh;
  const int wc = s->pic_w_ctbs, hc = s->pic_h_ctbs;
  const int nal = g->cur_nal, cra = nal == NAL_CRA, radl = nal == NAL_RADL_R || nal == NAL_RADL_N;
  g->au.len = 0; g->au.nbits = 0; g->au.cur = 0;
  if (g->cfg.hdr_extras && rpct(g, 50)) { const uint8_t aud[1] = { (uint8_t)((idr ? 0 : 2) << 5 | 0x10) }; orc_write_nal(&g->au, NAL_AUD, g->cur_tid, aud, 1, 1); }      /* pic_type, rbsp trailing bits */
  if (write_ps) {
    orc_bw_init(&ps); orc_write_vps(&ps, &g->vps, &g->sps); orc_write_nal(&g->au, NAL_VPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
    orc_bw_init(&ps); orc_write_sps(&ps, &g->sps); orc_write_nal(&g->au, NAL_SPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
    orc_bw_init(&ps); orc_write_pps(&ps, &g->pps); orc_write_nal(&g->au, NAL_PPS, 0, ps.buf, ps.len, 1); orc_bw_free(&ps);
  }
  if (g->cfg.hdr_extras && rpct(g, 50)) gen_unknown_sei(g, NAL_SEI_PREFIX);
  /* ---- slice header */
  memset(sh, 0, sizeof(*sh));
  sh->first_slice_segment_in_pic = 1; sh->pic_output_flag = p->output_flag_present ? !rpct(g, g->cfg.hidden_pics) : 1;
  g->slice_is_intra = idr || cra || g->since_idr == 0 || rpct(g, 8);
  g->slice_is_b = !g->slice_is_intra && g->cfg.b_slices > 0 && rpct(g, g->cfg.b_slices);
  sh->slice_type = g->slice_is_intra ? SLICE_I : (g->slice_is_b ? SLICE_B : SLICE_P);
  sh->poc_lsb = g->poc & ((1 << s->log2_max_poc_lsb) - 1);
  sh->num_ref_idx_l1 = p->num_ref_idx_l1_default;
  if (!idr && !g->cfg.gop && !g->cfg.b_slices) {
    /* reference picture set in the slice header, the way Kvazaar writes it: the previous pictures back to the IDR, at most num_refs */
    int nneg = ORC_MIN(g->cfg.num_refs, g->since_idr);
    int lt_used = 0;
    if (s->long_term_ref_pics_present && g->since_idr >= 1) {
      /* the sequence's first picture (POC 0; in this branch POC = pictures since the IDR) as a long-term reference picture: marked at once now and then, else
       * when it is about to leave the short-term window; from then on it is no short-term picture any more, and it stays until a slice stops naming it */
      if (g->lt_alive && !g->lt_marked && (g->since_idr > g->cfg.num_refs || (g->since_idr >= 2 && rpct(g, 30)))) g->lt_marked = 1;
      if (g->lt_alive && g->lt_marked && rpct(g, 4)) g->lt_alive = 0;
      if (g->lt_marked) nneg = ORC_MIN(nneg, g->since_idr - 1);
      if (!g->lt_marked && g->since_idr > g->cfg.num_refs) g->lt_alive = 0;
      if (g->lt_alive && g->lt_marked) {
        const int want_used = rpct(g, 70), max_lsb = 1 << s->log2_max_poc_lsb;
        int via_sps = -1;
        for (int i = 0; i < s->num_lt_sps; i++) if (s->lt_used_sps[i] == want_used && rpct(g, 60)) via_sps = i;
        sh->num_long_term_sps = via_sps >= 0; sh->num_long_term_pics = via_sps < 0; sh->num_lt = 1;
        sh->lt_idx_sps[0] = via_sps >= 0 ? via_sps : 0; sh->lt_poc_lsb[0] = 0; sh->lt_used[0] = (uint8_t)want_used;
        sh->lt_msb_present[0] = (uint8_t)(g->poc >= max_lsb || rpct(g, 40));      /* (needed once another picture may share the LSBs) */
        sh->lt_msb_cycle_delta[0] = sh->lt_msb_cycle[0] = sh->lt_msb_present[0] ? g->poc >> s->log2_max_poc_lsb : 0;
        lt_used = want_used;
      }
    }
    sh->short_term_ref_pic_set_sps_flag = 0;
    sh->st_rps.num_negative = nneg;
    int used = 0;
    for (int i = 0; i < nneg; i++) { sh->st_rps.delta_poc_s0[i] = -(i + 1); sh->st_rps.used_s0[i] = (i == 0) ? 1 : rpct(g, 85); used += sh->st_rps.used_s0[i]; }
    used += lt_used;
    if (!used) { sh->lt_used[0] = 1; lt_used = used = 1; if (sh->num_long_term_sps) { sh->num_long_term_sps = 0; sh->num_long_term_pics = 1; } }      /* (nothing short-term left -- the picture behind the IDR after an early marking: the long-term one is used) */
    choose_rps_form(g, sh, s);
    sh->num_ref_idx_l0 = g->slice_is_intra ? p->num_ref_idx_l0_default : rrange(g, 1, ORC_MIN(4, used + 1));   /* (more entries than pictures: the list wraps) */
    sh->slice_temporal_mvp_enabled = s->temporal_mvp_enabled ? rpct(g, 80) : 0;
    sh->collocated_from_l0 = 1;
    sh->collocated_ref_idx = (sh->slice_temporal_mvp_enabled && !g->slice_is_intra) ? rrange(g, 0, sh->num_ref_idx_l0 - 1) : 0;
  } else if (!idr) {
    /* the general form (B slices, reordered groups): the set is the last num_refs pictures in DECODING order, which lie on both sides of the current
     * one once pictures are reordered -- the ones before it in output order nearest first (S0), then the ones after it (S1) */
    const int nh = ORC_MIN(g->cfg.num_refs, g->hist_n);
    int neg[8], pos[8], nn = 0, np = 0, nbad[8], pbad[8];
    int has_cra = 0;
    /* "bad": not to be predicted from -- for a RADL picture what came before its CRA picture and RASL pictures; with sub-layers what has a higher TemporalId */
    for (int i = 0; i < nh; i++) {
      const int bad = (radl && g->hist_cls[i]) || (g->layers && g->hist_tid[i] > g->cur_tid);
      if (g->hist_poc[i] < g->poc) { nbad[nn] = bad; neg[nn++] = g->hist_poc[i]; } else { pbad[np] = bad; pos[np++] = g->hist_poc[i]; }
      has_cra |= g->hist_poc[i] == g->cra_poc;
    }
    if (g->cra_poc >= 0 && g->poc < g->cra_poc && !has_cra) { pbad[np] = 0; pos[np++] = g->cra_poc; }      /* a leading picture keeps its CRA picture: all that the trailing pictures may start from */
    for (int i = 1; i < nn; i++) for (int j = i; j > 0 && neg[j] > neg[j - 1]; j--) { int t = neg[j]; neg[j] = neg[j - 1]; neg[j - 1] = t; t = nbad[j]; nbad[j] = nbad[j - 1]; nbad[j - 1] = t; }
    for (int i = 1; i < np; i++) for (int j = i; j > 0 && pos[j] < pos[j - 1]; j--) { int t = pos[j]; pos[j] = pos[j - 1]; pos[j - 1] = t; t = pbad[j]; pbad[j] = pbad[j - 1]; pbad[j - 1] = t; }
    sh->short_term_ref_pic_set_sps_flag = 0;
    sh->st_rps.num_negative = nn; sh->st_rps.num_positive = np;
    int used = 0;
    for (int i = 0; i < nn; i++) { sh->st_rps.delta_poc_s0[i] = neg[i] - g->poc; sh->st_rps.used_s0[i] = rpct(g, 85); used += sh->st_rps.used_s0[i]; }
    for (int i = 0; i < np; i++) { sh->st_rps.delta_poc_s1[i] = pos[i] - g->poc; sh->st_rps.used_s1[i] = rpct(g, 85); used += sh->st_rps.used_s1[i]; }
    if (radl || g->layers) {
      /* a RADL picture predicts from its CRA picture and other RADL pictures only (the rest of the set stays, unused: RASL pictures that follow may need it);
       * no picture predicts from a higher sub-layer */
      int ok = -1; used = 0;
      for (int i = 0; i < nn; i++) { if (nbad[i]) sh->st_rps.used_s0[i] = 0; else if (ok < 0) ok = i; used += sh->st_rps.used_s0[i]; }
      for (int i = 0; i < np; i++) { if (pbad[i]) sh->st_rps.used_s1[i] = 0; else if (ok < 0) ok = 8 + i; used += sh->st_rps.used_s1[i]; }
      if (!used && ok >= 0) { if (ok < 8) sh->st_rps.used_s0[ok] = 1; else sh->st_rps.used_s1[ok - 8] = 1; used = 1; }
      if (!used) { g->slice_is_intra = 1; g->slice_is_b = 0; sh->slice_type = SLICE_I; used = 1; }      /* (the CRA picture is out of reach: an intra picture) */
    }
    else if (!used) { if (nn) sh->st_rps.used_s0[0] = 1; else if (np) sh->st_rps.used_s1[0] = 1; else { g->slice_is_intra = 1; g->slice_is_b = 0; sh->slice_type = SLICE_I; } used = 1; }
    choose_rps_form(g, sh, s);
    sh->num_ref_idx_l0 = g->slice_is_intra ? p->num_ref_idx_l0_default : rrange(g, 1, ORC_MIN(4, used + 1));
    if (g->slice_is_b) { sh->num_ref_idx_l1 = rrange(g, 1, ORC_MIN(4, used + 1)); sh->mvd_l1_zero = rpct(g, 30); }
    sh->slice_temporal_mvp_enabled = s->temporal_mvp_enabled ? rpct(g, 80) : 0;
    sh->collocated_from_l0 = g->slice_is_b ? (int)(rnd(g) & 1u) : 1;
    sh->collocated_ref_idx = (sh->slice_temporal_mvp_enabled && !g->slice_is_intra) ? rrange(g, 0, (sh->collocated_from_l0 ? sh->num_ref_idx_l0 : sh->num_ref_idx_l1) - 1) : 0;
  } else sh->num_ref_idx_l0 = p->num_ref_idx_l0_default;
  sh->cabac_init_flag = p->cabac_init_present ? rpct(g, 50) : 0;
  sh->rpl_mod_flag[0] = sh->rpl_mod_flag[1] = 0;
  if (!g->slice_is_intra && p->lists_modification_present) {
    int total = 0;
    for (int i = 0; i < sh->st_rps.num_negative; i++) total += sh->st_rps.used_s0[i];
    for (int i = 0; i < sh->st_rps.num_positive; i++) total += sh->st_rps.used_s1[i];
    for (int i = 0; i < sh->num_lt; i++) total += sh->lt_used[i];
    if (total > 1) for (int l = 0; l < (g->slice_is_b ? 2 : 1); l++) if (rpct(g, g->cfg.list_mod)) {
      sh->rpl_mod_flag[l] = 1;
      for (int i = 0; i < (l ? sh->num_ref_idx_l1 : sh->num_ref_idx_l0); i++) sh->list_entry[l][i] = (uint8_t)rrange(g, 0, total - 1);
    }
  }
  sh->weighted = 0;
  if (!g->slice_is_intra && g->cfg.weighted > 0) {
    /* pred_weight_table(): denominators 0 .. 7, weights and offsets over their whole ranges now and then, mostly near the defaults (a fade) */
    sh->weighted = 1;
    memset(sh->luma_weight_flag, 0, sizeof(sh->luma_weight_flag)); memset(sh->chroma_weight_flag, 0, sizeof(sh->chroma_weight_flag));
    sh->luma_log2_weight_denom = rrange(g, 0, 7);
    sh->delta_chroma_log2_weight_denom = rrange(g, 0, 7) - sh->luma_log2_weight_denom;
    for (int l = 0; l < (g->slice_is_b ? 2 : 1); l++) for (int i = 0; i < (l ? sh->num_ref_idx_l1 : sh->num_ref_idx_l0); i++) {
      const int wide = rpct(g, 15);
      if (rpct(g, g->cfg.weighted)) {
        sh->luma_weight_flag[l][i] = 1;
        sh->delta_luma_weight[l][i] = (int16_t)(wide ? rrange(g, -128, 127) : rrange(g, -6, 6));
        sh->luma_offset[l][i] = (int16_t)(wide ? rrange(g, -128, 127) : rrange(g, -10, 10));
      }
      if (rpct(g, g->cfg.weighted)) {
        sh->chroma_weight_flag[l][i] = 1;
        for (int j = 0; j < 2; j++) {
          sh->delta_chroma_weight[l][i][j] = (int16_t)(wide ? rrange(g, -128, 127) : rrange(g, -6, 6));
          sh->delta_chroma_offset[l][i][j] = (int16_t)(wide ? rrange(g, -512, 511) : rrange(g, -20, 20));
        }
      }
    }
    orc_derive_pred_weights(sh);
  }
  sh->max_num_merge_cand = rrange(g, 1, 5);
  sh->slice_qp_delta = rrange(g, -4, 4);
  sh->slice_qp = p->init_qp + sh->slice_qp_delta;
  if (p->slice_chroma_qp_offsets_present) { sh->slice_cb_qp_offset = rrange(g, -2, 2); sh->slice_cr_qp_offset = rrange(g, -2, 2); }
  sh->slice_deblocking_disabled = p->pps_deblocking_disabled; sh->beta_offset_div2 = p->pps_beta_offset_div2; sh->tc_offset_div2 = p->pps_tc_offset_div2;
  if (p->deblocking_filter_override_enabled && rpct(g, 60)) {
    sh->deblocking_filter_override = 1;
    sh->slice_deblocking_disabled = rpct(g, 25);
    if (!sh->slice_deblocking_disabled) { sh->beta_offset_div2 = rrange(g, -6, 6); sh->tc_offset_div2 = rrange(g, -6, 6); }
  }
  sh->loop_filter_across_slices = p->loop_filter_across_slices ? (g->cfg.lf_across ? rpct(g, 50) : 1) : 0;      /* (absent from the header: the PPS's value, 7.4.7.1) */
  if (s->sao_enabled) { sh->sao_luma = rpct(g, 80); sh->sao_chroma = rpct(g, 80); }
  /* ---- slice data: one substream per CTU row with WPP (or with a slice segment per row), else one per tile */
  const int wpp = p->entropy_coding_sync_enabled, slices = g->cfg.slices;
  if (slices == 3) { write_free_slices(g, nal); gen_au_tail(g); return; }
  const int row_subs = wpp || slices == 1, cols = g->ncols_t;
  const int nsub = (row_subs ? hc : g->nrows_t) * cols;
  subs = (orc_bitw *)calloc((size_t)nsub, sizeof(orc_bitw));
  int *seg_first = (int *)calloc((size_t)nsub + 1, sizeof(int)), *seg_addr = (int *)calloc((size_t)nsub + 1, sizeof(int)), nseg = 0;   /* slice segments: first substream, CTB address */
  orc_ctx saved[CTX_COUNT];
  const int init_type = g->slice_is_intra ? 0 : (g->slice_is_b ? (sh->cabac_init_flag ? 1 : 2) : (sh->cabac_init_flag ? 2 : 1));      /* 9.3.2.2: cabac_init_flag swaps the P and B tables */
  int sub = -1;
  orc_pic_reset_side(&g->side);
  memset(&g->c, 0, sizeof(g->c));
  /* 6.5.1: tile after tile (raster order of tiles), the CTBs of a tile in raster order */
  for (int tr = 0; tr < g->nrows_t; tr++) for (int tc = 0; tc < cols; tc++) {
    const int x0 = g->col_bd[tc], x1 = g->col_bd[tc + 1], tw = x1 - x0;
    for (int cy = g->row_bd[tr]; cy < g->row_bd[tr + 1]; cy++) {
      const int tile_start = cy == g->row_bd[tr], tile_end = cy + 1 == g->row_bd[tr + 1];
      if (tile_start || row_subs) {
        sub++;
        if (sub == 0 || slices == 1 || (slices == 2 && tile_start)) { seg_first[nseg] = sub; seg_addr[nseg] = cy * wc + x0; nseg++; }
        orc_bw_init(&subs[sub]);
        orc_cenc_start(&g->c, &subs[sub]);
        /* 9.3.1: first CTB of a tile initialises; a WPP row synchronises with the state after the 2nd CTB of the row above inside the
         * tile when that CTB exists (tiles one CTB wide: it does not, the row initialises afresh); a dependent slice segment that
         * starts anywhere else goes on with the contexts the previous segment ended with (they are still in g->c.ctx) */
        if (tile_start || (wpp && tw < 2)) orc_cabac_init_contexts(g->c.ctx, init_type, sh->slice_qp);
        else if (wpp) memcpy(g->c.ctx, saved, sizeof(saved));
      }
      for (int cx = x0; cx < x1; cx++) {
        if (sh->sao_luma || sh->sao_chroma) {
          orc_sao_params *sp = &g->sao[cy * wc + cx];
          const orc_sao_params *left = cx > x0 ? sp - 1 : NULL, *up = (cy > 0 && !tile_start) ? sp - wc : NULL;
          draw_sao(g, sp, left, up, sh->sao_luma, sh->sao_chroma);
          orc_sao_write(&g->c, sp, left, up, sh->sao_luma, sh->sao_chroma);
        }
        gen_coding_quadtree(g, cx << s->ctb_log2, cy << s->ctb_log2, s->ctb_log2, 0);
        if (wpp && cx == x0 + 1) memcpy(saved, g->c.ctx, sizeof(saved));
        const int last = (tr == g->nrows_t - 1 && tc == cols - 1 && tile_end && cx == x1 - 1);
        const int sub_end = cx == x1 - 1 && (row_subs || tile_end);
        const int seg_end = last || (cx == x1 - 1 && (slices == 1 || (slices == 2 && tile_end)));
        orc_cenc_terminate(&g->c, seg_end);               /* end_of_slice_segment_flag */
        if (!seg_end && sub_end) orc_cenc_terminate(&g->c, 1);   /* end_of_subset_one_bit */
        if (seg_end || sub_end) orc_bw_align_zero(g->c.bw);
      }
    }
  }
  seg_first[nseg] = nsub;
  /* ---- one NAL unit per slice segment: the first carries the full header; a dependent segment only its address and entry points,
   * an independent slice (slices = 2) the same header again with its own address */
  for (int k = 0; k < nseg; k++) {
    const int s0 = seg_first[k], n = seg_first[k + 1] - s0;
    uint32_t *ep = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
    sh->first_slice_segment_in_pic = k == 0; sh->slice_segment_address = seg_addr[k]; sh->dependent_slice_segment = (k > 0 && slices == 1);
    if (k > 0 && slices == 2 && g->cfg.lf_across == 1 && p->loop_filter_across_slices) sh->loop_filter_across_slices = rpct(g, 50);      /* (a slice per tile: each its own flag) */
    sh->num_entry_points = n - 1; sh->entry_point_offset = ep;
    for (int i = 0; i < n - 1; i++) ep[i] = (uint32_t)orc_escaped_size(subs[s0 + i].buf, subs[s0 + i].len);
    orc_bw_init(&hdr);
    orc_write_slice_header(&hdr, sh, s, p, nal);
    for (int i = 0; i < n; i++) { orc_bw_bytes(&hdr, subs[s0 + i].buf, subs[s0 + i].len); orc_bw_free(&subs[s0 + i]); }
    orc_write_nal(&g->au, nal, g->cur_tid, hdr.buf, hdr.len, 1);
    orc_bw_free(&hdr); free(ep);
  }
  free(subs); free(seg_first); free(seg_addr);
  sh->entry_point_offset = NULL;
  gen_au_tail(g);
}

size_t orc_gen_picture(orc_gen *g, const uint8_t **au)
{
  const int period = g->cfg.intra_period;
  const int idr = (g->frame_idx == 0) || (period > 0 && (g->frame_idx % period) == 0);
  int cra = 0;
  if (idr) { g->poc = 0; g->since_idr = 0; g->hist_n = 0; g->lt_alive = 1; g->lt_marked = 0; g->cra_poc = -1; }
  else {
    g->since_idr++;
    if (g->cfg.gop) { const int k = g->since_idr - 1; g->poc = (k / g->cfg.gop) * g->cfg.gop + g->gop_order[k % g->cfg.gop]; cra = g->cfg.open_gop && k % g->cfg.gop == 0 && rpct(g, 45); }
    else g->poc++;
  }
  /* open_gop: the picture a group starts with (its last one in output order) is a CRA picture now and then -- the group's other pictures are its LEADING pictures: RASL
   * below a drawn POC (they may predict from anything), RADL from there on (from the CRA picture and from each other only; 7.4.2.2: RASL before RADL in output order) */
  g->cur_nal = idr ? NAL_IDR_W_RADL : NAL_TRAIL_R;
  if (!idr && g->cra_poc >= 0 && g->poc > g->cra_poc) {
    /* 8.3.2: a trailing picture of a CRA picture, and the next CRA picture too, have nothing in their sets that precedes that CRA picture in decoding or
     * output order -- those pictures leave for good */
    int n = 0;
    for (int i = 0; i < g->hist_n; i++) if (g->hist_poc[i] >= g->cra_poc) { g->hist_poc[n] = g->hist_poc[i]; g->hist_cls[n] = g->hist_cls[i]; g->hist_tid[n] = g->hist_tid[i]; n++; }
    g->hist_n = n;
  }
  if (cra) {
    g->cur_nal = NAL_CRA; g->cra_poc = g->poc; g->cra_split = g->poc - g->cfg.gop + 1 + rrange(g, 0, g->cfg.gop - 1);
    for (int i = 0; i < g->hist_n; i++) g->hist_cls[i] = 1;
  } else if (!idr && g->cra_poc >= 0 && g->poc < g->cra_poc) g->cur_nal = g->poc < g->cra_split ? NAL_RASL_R : NAL_RADL_R;
  /* temporal_layers: a picture's place in its group is its sub-layer (the group's last picture 0, the middle one 1, ...); the top layer's pictures are predicted from
   * by nobody -- sub-layer non-reference pictures (the _N types), which never enter the reference picture sets */
  g->cur_tid = 0;
  int leaf = 0;
  if (g->layers && !idr) {
    const int o = (g->since_idr - 1) % g->cfg.gop, off = g->gop_order[o] % g->cfg.gop;
    if (off) { int z = 0; while (!((off >> z) & 1)) z++; g->cur_tid = g->layers - 1 - z; }
    leaf = g->cur_tid == g->layers - 1;
    if (leaf) g->cur_nal -= 1;                                   /* TRAIL_R -> TRAIL_N, RASL_R -> RASL_N, RADL_R -> RADL_N */
  }
  write_picture(g, idr, idr || cra);
  if (!leaf) {
    for (int i = ORC_MIN(g->hist_n, 7); i > 0; i--) { g->hist_poc[i] = g->hist_poc[i - 1]; g->hist_cls[i] = g->hist_cls[i - 1]; g->hist_tid[i] = g->hist_tid[i - 1]; }
    g->hist_poc[0] = g->poc; g->hist_cls[0] = g->cur_nal == NAL_RASL_R; g->hist_tid[0] = g->cur_tid; if (g->hist_n < 8) g->hist_n++;
  }
  g->frame_idx++;
  *au = g->au.buf;
  return g->au.len;
}
