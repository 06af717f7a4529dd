/* oracle/hevc_dec.c -- see hevc_dec.h.  Test infrastructure: a from-the-standard HEVC decoder
 * used as the CPU checker for the decode half of the hot path
 * (/root/reference/src/media/processing/openhevcfilter.cpp:103-239). */
#include "hevc_dec.h"
#include "hevc_hash.h"
#include "hevc_bits.h"
#include "hevc_cabac.h"
#include "hevc_ps.h"
#include "hevc_pic.h"
#include "hevc_intra.h"
#include "hevc_inter.h"
#include "hevc_transform.h"
#include "hevc_deblock.h"
#include "hevc_sao.h"
#include "hevc_mvpred.h"
#include <stdio.h>

#define MAX_DPB 17
#define ERR_UNSUPPORTED (-2)
#define ERR_INVALID (-1)

struct orc_decoder {
  orc_vps vps[16]; orc_sps sps[16]; orc_pps pps[64];
  orc_pic dpb[MAX_DPB];
  orc_pic *cur;
  const orc_sps *s; const orc_pps *p;
  orc_slice_hdr sh;
  int nal_type;
  int prev_tid0_poc;
  int seen_irap;
  int tid;                                /* TemporalId of the NAL unit at hand */
  uint8_t pre_ref[MAX_DPB];               /* the reference pictures of the DPB as the current picture found it, before its reference picture set was applied (missing_ref) */
  int concealed;                          /* missing reference pictures replaced so far (missing_ref) */
  int after_eos;                          /* an end of sequence NAL unit came: the next picture starts a coded video sequence (a CRA picture then has NoRaslOutputFlag = 1) */
  int skip_rasl;                          /* NoRaslOutputFlag of the last IRAP picture: its RASL pictures are not decoded (8.1.3) */
  orc_pic *ref_list0[16]; int ref_poc[16]; int num_ref;
  orc_pic *ref_list1[16]; int ref_poc1[16]; int num_ref1;     /* RefPicList1 (B slices) */
  int no_backward_pred;                   /* NoBackwardPredFlag, 8.5.3.2.9 */
  int cvs;                                /* coded video sequences started so far (output order: sequence after sequence, POC inside one) */
  int ctbs_decoded; int pic_active;
  uint8_t *bs_v, *bs_h; size_t bs_cap;
  pixel *predeblock[3]; size_t predeblock_cap;
  orc_pic *out_queue[MAX_DPB + 1]; int out_n;
  orc_pic *last_output;
  orc_pic *last_finished;                 /* the picture finish_picture() completed last: what a decoded picture hash SEI behind it refers to */
  int hash_checked, hash_mismatch;        /* decoded picture hash SEI messages seen / that did not match the decoded picture */
  uint8_t *rbsp; size_t rbsp_cap;
  int64_t cur_pts;
  /* tile / slice maps for the current picture */
  orc_sao_params *sao; int sao_used; pixel *sao_in[3];
  int32_t *ctb_slice; int16_t *ctb_tile; int *ts_to_rs, *rs_to_ts; int *tile_first_x; size_t ctb_cap;
  uint8_t *intra4; size_t intra4_cap;     /* constrained intra prediction: per 4x4 luma block of the current picture, 1 = intra */
  uint8_t *ctb_lfx, *ctb_nb;              /* per CTB: slice_loop_filter_across_slices_enabled_flag of its slice; the neighbouring CTBs the in-loop filters may use (finish_picture) */
  int lf_restricted;                      /* some slice of the picture has that flag 0 */
  int slice_addr_rs;                      /* SliceAddrRs: address of the slice (= its first, independent segment) being decoded */
  orc_ctx ds_ctx[CTX_COUNT];              /* TableStateIdxDs: contexts at the end of the previous slice segment (9.3.2.2) */
  int col_bd[34], row_bd[34];
  /* slice decoding state */
  orc_cabac_dec cabac; orc_ctx wpp_ctx[CTX_COUNT];
  orc_avail_ctx av;
  int cu_transquant_bypass;
  int scaling_on; uint8_t sfac[4][6][1024];     /* scaling factors of the active lists (PPS, else SPS, else default), per size and matrix; scaling_on = scaling_list_enabled_flag */
  int is_cu_qp_delta_coded, cu_qp_delta_val;
  int qp_y, last_qp_y, qg_x, qg_y, qp_y_pred;
  int intra_chroma_pred_mode;
  int max_trafo_depth, intra_split;
  int cu_pred_mode, part_mode;
  int err;
  int last_slice_type;
};

/* ------------------------------------------------------------------ helpers */
static inline int b4(const orc_pic *p, int x, int y) { return (y >> 2) * p->b4_w + (x >> 2); }
static void fill_b4_u8(orc_pic *p, uint8_t *arr, int x0, int y0, int w, int h, int v)
{
  for (int y = y0; y < y0 + h && y < p->h; y += 4)
    for (int x = x0; x < x0 + w && x < p->w; x += 4) arr[b4(p, x, y)] = (uint8_t)v;
}

orc_decoder *orc_dec_open(void)
{
  orc_decoder *d = (orc_decoder *)calloc(1, sizeof(*d));
  orc_tables_init();
  return d;
}
void orc_dec_close(orc_decoder *d)
{
  if (!d) return;
  for (int i = 0; i < MAX_DPB; i++) if (d->dpb[i].plane[0]) orc_pic_free(&d->dpb[i]);
  free(d->bs_v); free(d->bs_h); free(d->rbsp);
  for (int i = 0; i < 3; i++) free(d->predeblock[i]);
  free(d->sao); for (int i = 0; i < 3; i++) free(d->sao_in[i]);
  free(d->ctb_slice); free(d->ctb_tile); free(d->ts_to_rs); free(d->rs_to_ts); free(d->tile_first_x); free(d->ctb_lfx); free(d->ctb_nb); free(d->intra4);
  free(d->sh.entry_point_offset);
  free(d);
}
const pixel *orc_dec_predeblock_plane(orc_decoder *d, int c) { return d->predeblock[c]; }

/* ------------------------------------------------------------------ residual_coding, 7.3.8.11 */
static int decode_last_prefix(orc_decoder *d, int base, int log2, int cidx)
{
  int off, sh, max = (log2 << 1) - 1, v = 0;
  if (cidx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); sh = (log2 + 1) >> 2; }
  else { off = 15; sh = log2 - 2; }
  while (v < max && orc_cdec_bin(&d->cabac, base + off + (v >> sh))) v++;
  return v;
}

static int decode_abs_remaining(orc_decoder *d, int rice)
{
  int prefix = 0;
  while (prefix < 32 && orc_cdec_bypass(&d->cabac)) prefix++;
  if (prefix >= 32) { d->err = ERR_INVALID; return 0; }
  if (prefix <= 3) return (prefix << rice) + (int)orc_cdec_bypass_bits(&d->cabac, rice);
  return (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)orc_cdec_bypass_bits(&d->cabac, prefix - 3 + rice);
}

static const uint8_t ctx_idx_map_4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

/* Parses one transform block into coeff[n*n] (row-major, [y][x]); returns transform_skip_flag. */
static int residual_coding(orc_decoder *d, int log2, int cidx, int scan_idx, int16_t *coeff)
{
  orc_cabac_dec *c = &d->cabac;
  int n = 1 << log2, ts = 0;
  uint8_t csbf[8][8];
  memset(coeff, 0, sizeof(int16_t) * (size_t)n * n);
  memset(csbf, 0, sizeof(csbf));
  if (d->p->transform_skip_enabled && !d->cu_transquant_bypass && log2 <= 2)
    ts = orc_cdec_bin(c, CTX_TS_FLAG + (cidx ? 1 : 0));
  int lx = decode_last_prefix(d, CTX_LAST_X, log2, cidx);
  int ly = decode_last_prefix(d, CTX_LAST_Y, log2, cidx);
  if (lx > 3) { int nb = (lx >> 1) - 1; lx = (1 << nb) * (2 + (lx & 1)) + (int)orc_cdec_bypass_bits(c, nb); }
  if (ly > 3) { int nb = (ly >> 1) - 1; ly = (1 << nb) * (2 + (ly & 1)) + (int)orc_cdec_bypass_bits(c, nb); }
  if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
  int sb_log2 = log2 - 2, nsb = 1 << sb_log2;
  const uint8_t *sbx = orc_scan_x[scan_idx][sb_log2], *sby = orc_scan_y[scan_idx][sb_log2];
  const uint8_t *px = orc_scan_x[scan_idx][2], *py = orc_scan_y[scan_idx][2];
  int last_sb = (1 << (2 * sb_log2)) - 1, last_pos = 16;
  for (;;) {
    if (last_pos == 0) { last_pos = 16; last_sb--; if (last_sb < 0) { d->err = ERR_INVALID; return ts; } }
    last_pos--;
    int xc = (sbx[last_sb] << 2) + px[last_pos], yc = (sby[last_sb] << 2) + py[last_pos];
    if (xc == lx && yc == ly) break;
  }
  int c1 = 1;   /* greater1Ctx carried across sub-blocks, 9.3.4.2.6 */
  for (int i = last_sb; i >= 0; i--) {
    int xs = sbx[i], ys = sby[i];
    int infer_dc = 0;
    int right = (xs < nsb - 1) ? csbf[ys][xs + 1] : 0, below = (ys < nsb - 1) ? csbf[ys + 1][xs] : 0;
    if (i < last_sb && i > 0) {
      csbf[ys][xs] = (uint8_t)orc_cdec_bin(c, CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0));
      infer_dc = 1;
    } else csbf[ys][xs] = 1;
    uint8_t sig[16]; memset(sig, 0, sizeof(sig));
    int start = (i == last_sb) ? last_pos - 1 : 15;
    if (i == last_sb) sig[last_pos] = 1;
    int prev_csbf = right | (below << 1);
    for (int k = start; k >= 0; k--) {
      int xp = px[k], yp = py[k], xc = (xs << 2) + xp, yc = (ys << 2) + yp;
      if (csbf[ys][xs] && (k > 0 || !infer_dc)) {
        int sc;
        if (log2 == 2) sc = ctx_idx_map_4x4[(yc << 2) + xc];
        else if (xc + yc == 0) sc = 0;
        else {
          if (prev_csbf == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
          else if (prev_csbf == 1) sc = (yp == 0) ? 2 : (yp == 1) ? 1 : 0;
          else if (prev_csbf == 2) sc = (xp == 0) ? 2 : (xp == 1) ? 1 : 0;
          else sc = 2;
          if (cidx == 0) {
            if (i > 0) sc += 3;
            sc += (log2 == 3) ? ((scan_idx == 0) ? 9 : 15) : 21;
          } else {
            sc += (log2 == 3) ? 9 : 12;
          }
        }
        sig[k] = (uint8_t)orc_cdec_bin(c, CTX_SIG + (cidx ? 27 : 0) + sc);
        if (sig[k]) infer_dc = 0;
      } else if (k == 0 && csbf[ys][xs] && infer_dc) {
        sig[0] = 1;
      }
    }
    int nsig = 0; for (int k = 0; k < 16; k++) nsig += sig[k];
    if (!nsig) continue;
    /* greater1 / greater2 flags */
    int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
    if (c1 == 0) ctx_set++;
    c1 = 1;
    uint8_t g1[16], g2[16]; memset(g1, 0, sizeof(g1)); memset(g2, 0, sizeof(g2));
    int first_sig = 16, last_sig = -1, ng1 = 0, last_g1_pos = -1;
    for (int k = 15; k >= 0; k--) if (sig[k]) {
      if (ng1 < 8) {
        g1[k] = (uint8_t)orc_cdec_bin(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1);
        ng1++;
        if (g1[k]) { c1 = 0; if (last_g1_pos == -1) last_g1_pos = k; }
        else if (c1 > 0 && c1 < 3) c1++;
      }
      if (last_sig == -1) last_sig = k;
      first_sig = k;
    }
    int sign_hidden = d->p->sign_data_hiding && !d->cu_transquant_bypass && (last_sig - first_sig > 3);
    if (last_g1_pos != -1) g2[last_g1_pos] = (uint8_t)orc_cdec_bin(c, CTX_GT2 + (cidx ? 4 : 0) + ctx_set);
    uint8_t sign[16]; memset(sign, 0, sizeof(sign));
    for (int k = 15; k >= 0; k--) if (sig[k] && (!sign_hidden || k != first_sig)) sign[k] = (uint8_t)orc_cdec_bypass(c);
    int num_sig = 0, sum_abs = 0, rice = 0;
    for (int k = 15; k >= 0; k--) if (sig[k]) {
      int base = 1 + g1[k] + g2[k];
      int absv = base;
      if (base == ((num_sig < 8) ? ((k == last_g1_pos) ? 3 : 2) : 1)) {
        int rem = decode_abs_remaining(d, rice);
        absv = base + rem;
        if (absv > 3 * (1 << rice)) rice = ORC_MIN(rice + 1, 4);
      }
      int v = sign[k] ? -absv : absv;
      if (sign_hidden) { sum_abs += absv; if (k == first_sig && (sum_abs & 1)) v = -v; }
      int xc = (xs << 2) + px[k], yc = (ys << 2) + py[k];
      coeff[yc * n + xc] = (int16_t)orc_clip3(-32768, 32767, v);
      num_sig++;
    }
  }
  return ts;
}

/* ------------------------------------------------------------------ reconstruction of one TB */
static void recon_tb(orc_decoder *d, int cidx, int x0, int y0, int log2, const int16_t *level, int ts, int qp, int dst_mode)
{
  /* x0,y0 in component samples */
  orc_pic *pic = d->cur;
  int n = 1 << log2;
  int16_t coeff[32 * 32], res[32 * 32];
  if (d->cu_transquant_bypass) {
    memcpy(res, level, sizeof(int16_t) * (size_t)n * n);
  } else {
    orc_dequant_m(level, coeff, n, qp, d->scaling_on ? d->sfac[log2 - 2][orc_scaling_matrix_id(log2 - 2, cidx, d->cu_pred_mode != MODE_INTRA)] : NULL);
    if (ts) { for (int i = 0; i < n * n; i++) res[i] = (int16_t)((((int)coeff[i] << 7) + (1 << 11)) >> 12); }
    else orc_inv_transform(coeff, res, n, dst_mode);
  }
  pixel *dst = pic->plane[cidx] + y0 * pic->stride[cidx] + x0;
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) dst[y * pic->stride[cidx] + x] = (pixel)orc_clip_pixel(dst[y * pic->stride[cidx] + x] + res[y * n + x]);
}

static void intra_pred_tb(orc_decoder *d, int cidx, int x0, int y0, int log2, int mode)
{
  orc_pic *pic = d->cur;
  int n = 1 << log2;
  pixel left[65 + 64], top[65 + 64];
  if (d->p->constrained_intra_pred) {
    /* constrained_intra_pred_flag (8.4.4.2.2): a neighbouring sample of a block that is not intra-coded is marked "not available" -- for the reference samples only;
     * the substitution process is the usual one.  intra4: per 4x4 luma block, 1 = intra (coding_unit keeps it) */
    orc_avail_ctx av = d->av; av.usable4 = d->intra4; av.usable_stride = pic->b4_w;
    orc_intra_refs(&av, pic->plane[cidx], pic->stride[cidx], cidx, x0, y0, n, left, top);
  } else
  orc_intra_refs(&d->av, pic->plane[cidx], pic->stride[cidx], cidx, x0, y0, n, left, top);
  orc_intra_predict(left, top, n, cidx, mode, d->s->strong_intra_smoothing,
                    pic->plane[cidx] + y0 * pic->stride[cidx] + x0, pic->stride[cidx]);
}

/* ------------------------------------------------------------------ QP derivation 8.6.1 */
static void derive_qp_pred(orc_decoder *d, int xqg, int yqg, int first_qg_in_ctb_row_or_slice)
{
  orc_pic *pic = d->cur;
  int prev = first_qg_in_ctb_row_or_slice ? d->sh.slice_qp : d->last_qp_y;
  int ctb = d->s->ctb_log2;
  int qa = prev, qb = prev;
  if (orc_available(&d->av, xqg, yqg, xqg - 1, yqg) && ((xqg - 1) >> ctb) == (xqg >> ctb))
    qa = pic->qp_y[b4(pic, xqg - 1, yqg)];
  if (orc_available(&d->av, xqg, yqg, xqg, yqg - 1) && ((yqg - 1) >> ctb) == (yqg >> ctb))
    qb = pic->qp_y[b4(pic, xqg, yqg - 1)];
  d->qp_y_pred = (qa + qb + 1) >> 1;
}

/* ------------------------------------------------------------------ transform tree */
typedef struct { int x0, y0, log2cb; int intra_modes[4]; int chroma_mode; } cu_info;

static int scan_idx_for(int pred_intra, int log2, int cidx, int mode)
{
  /* 7.4.9.11: mode dependent scan for intra 4x4 (luma+chroma) and luma 8x8 */
  if (!pred_intra) return 0;
  if (log2 == 2 || (log2 == 3 && cidx == 0)) {
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
  }
  return 0;
}

static void transform_unit(orc_decoder *d, const cu_info *cu, int x0, int y0, int xbase, int ybase, int log2, int depth, int blk,
                           int cbf_luma, int cbf_cb, int cbf_cr, int cbf_cb_parent, int cbf_cr_parent)
{
  orc_pic *pic = d->cur;
  orc_cabac_dec *c = &d->cabac;
  int intra = d->cu_pred_mode == MODE_INTRA;
  int n = 1 << log2;
  int chroma_here = log2 > 2, chroma_parent = (log2 == 2 && blk == 3);
  int ccb = chroma_here ? cbf_cb : (chroma_parent ? cbf_cb_parent : 0);
  int ccr = chroma_here ? cbf_cr : (chroma_parent ? cbf_cr_parent : 0);
  int cbf_chroma_any = (log2 > 2) ? (cbf_cb || cbf_cr) : (cbf_cb_parent || cbf_cr_parent);
  int16_t lev[32 * 32];
  (void)depth;
  if (cbf_luma || cbf_chroma_any) {
    if (d->p->cu_qp_delta_enabled && !d->is_cu_qp_delta_coded) {
      int v = 0;
      while (v < 5 && orc_cdec_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0))) v++;
      if (v == 5) { int k = 0; while (k < 16 && orc_cdec_bypass(c)) { v += 1 << k; k++; } v += (int)orc_cdec_bypass_bits(c, k); }
      if (v && orc_cdec_bypass(c)) v = -v;
      d->is_cu_qp_delta_coded = 1; d->cu_qp_delta_val = v;
      d->qp_y = ((d->qp_y_pred + v + 52) % 52);
    }
  }
  int qp_y = d->qp_y;
  int qp_cb = orc_chroma_qp(qp_y, d->p->cb_qp_offset + d->sh.slice_cb_qp_offset);
  int qp_cr = orc_chroma_qp(qp_y, d->p->cr_qp_offset + d->sh.slice_cr_qp_offset);
  /* luma */
  int lmode = 0;
  if (intra) {
    lmode = pic->intra_mode[b4(pic, x0, y0)];
    intra_pred_tb(d, 0, x0, y0, log2, lmode);
  }
  if (cbf_luma) {
    int ts = residual_coding(d, log2, 0, scan_idx_for(intra, log2, 0, lmode), lev);
    recon_tb(d, 0, x0, y0, log2, lev, ts, qp_y, intra && log2 == 2);
    fill_b4_u8(pic, pic->tu_nz, x0, y0, n, n, 1);
  }
  /* chroma */
  if (chroma_here || chroma_parent) {
    int cx = (chroma_here ? x0 : xbase) >> 1, cy = (chroma_here ? y0 : ybase) >> 1;
    int clog2 = chroma_here ? log2 - 1 : 2;
    int cmode = cu->chroma_mode;
    for (int ci = 1; ci <= 2; ci++) {
      if (intra) intra_pred_tb(d, ci, cx, cy, clog2, cmode);
      if (ci == 1 ? ccb : ccr) {
        int ts = residual_coding(d, clog2, ci, scan_idx_for(intra, clog2, ci, cmode), lev);
        recon_tb(d, ci, cx, cy, clog2, lev, ts, ci == 1 ? qp_cb : qp_cr, 0);
      }
    }
  }
}

static void transform_tree(orc_decoder *d, const cu_info *cu, int x0, int y0, int xbase, int ybase, int log2, int depth, int blk,
                           int cbf_cb_parent, int cbf_cr_parent)
{
  orc_pic *pic = d->cur;
  orc_cabac_dec *c = &d->cabac;
  const orc_sps *s = d->s;
  int split;
  if (d->err) return;
  if (log2 <= s->log2_max_tb && log2 > s->log2_min_tb && depth < d->max_trafo_depth && !(d->intra_split && depth == 0))
    split = orc_cdec_bin(c, CTX_SPLIT_TRANSFORM + 5 - log2);
  else {
    int inter_split = (s->max_th_depth_inter == 0 && d->cu_pred_mode == MODE_INTER && d->part_mode != PART_2Nx2N && depth == 0);
    split = (log2 > s->log2_max_tb || (d->intra_split && depth == 0) || inter_split) ? 1 : 0;
  }
  int cbf_cb = 0, cbf_cr = 0;
  if (log2 > 2) {
    if (depth == 0 || cbf_cb_parent) cbf_cb = orc_cdec_bin(c, CTX_CBF_CHROMA + depth);
    if (depth == 0 || cbf_cr_parent) cbf_cr = orc_cdec_bin(c, CTX_CBF_CHROMA + depth);
  } else { cbf_cb = cbf_cb_parent; cbf_cr = cbf_cr_parent; }   /* 7.4.9.8: inferred from parent for 4x4 luma */
  if (split) {
    int h = 1 << (log2 - 1);
    transform_tree(d, cu, x0, y0, x0, y0, log2 - 1, depth + 1, 0, cbf_cb, cbf_cr);
    transform_tree(d, cu, x0 + h, y0, x0, y0, log2 - 1, depth + 1, 1, cbf_cb, cbf_cr);
    transform_tree(d, cu, x0, y0 + h, x0, y0, log2 - 1, depth + 1, 2, cbf_cb, cbf_cr);
    transform_tree(d, cu, x0 + h, y0 + h, x0, y0, log2 - 1, depth + 1, 3, cbf_cb, cbf_cr);
  } else {
    int cbf_luma = 1;
    if (d->cu_pred_mode == MODE_INTRA || depth != 0 || cbf_cb || cbf_cr)
      cbf_luma = orc_cdec_bin(c, CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
    /* transform block edges for deblocking */
    int n = 1 << log2;
    for (int i = 0; i < n; i += 4) {
      if (y0 + i < pic->h) pic->edge_v[b4(pic, x0, y0 + i)] |= 1;
      if (x0 + i < pic->w) pic->edge_h[b4(pic, x0 + i, y0)] |= 1;
    }
    transform_unit(d, cu, x0, y0, xbase, ybase, log2, depth, blk, cbf_luma,
                   log2 > 2 ? cbf_cb : 0, log2 > 2 ? cbf_cr : 0, cbf_cb_parent, cbf_cr_parent);
  }
}

/* ------------------------------------------------------------------ prediction units */
static int decode_mvd_comp_abs(orc_decoder *d, int gt0, int gt1)
{
  if (!gt0) return 0;
  if (!gt1) return 1;
  /* abs_mvd_minus2: EG1 */
  int k = 1, v = 0;
  while (k < 32 && orc_cdec_bypass(&d->cabac)) { v += 1 << k; k++; }
  if (k >= 32) { d->err = ERR_INVALID; return 0; }
  v += (int)orc_cdec_bypass_bits(&d->cabac, k);
  return v + 2;
}

/* 8.5.3.3: the prediction block from one list (default weighted prediction of a single 14-bit array) or from both (their rounded average) */
static void mc_pu(orc_decoder *d, int xp, int yp, int w, int h, const orc_mvinfo *m)
{
  orc_pic *pic = d->cur;
  const orc_pic *r0 = m->ref_idx >= 0 ? d->ref_list0[m->ref_idx] : NULL, *r1 = m->ref_idx1 >= 0 ? d->ref_list1[m->ref_idx1] : NULL;
  static int16_t t0[64 * 64], t1[64 * 64];
  for (int ci = 0; ci < 3; ci++) {
    const int sh = ci ? 1 : 0, X = xp >> sh, Y = yp >> sh, W = w >> sh, H = h >> sh;
    if (r0) { if (ci) orc_mc_chroma(r0->plane[ci], r0->stride[ci], r0->w / 2, r0->h / 2, X, Y, W, H, m->mv[0], m->mv[1], t0, 64);
              else orc_mc_luma(r0->plane[0], r0->stride[0], r0->w, r0->h, X, Y, W, H, m->mv[0], m->mv[1], t0, 64); }
    if (r1) { if (ci) orc_mc_chroma(r1->plane[ci], r1->stride[ci], r1->w / 2, r1->h / 2, X, Y, W, H, m->mv1[0], m->mv1[1], t1, 64);
              else orc_mc_luma(r1->plane[0], r1->stride[0], r1->w, r1->h, X, Y, W, H, m->mv1[0], m->mv1[1], t1, 64); }
    pixel *dst = pic->plane[ci] + Y * pic->stride[ci] + X;
    if (d->sh.weighted) {
      /* explicit weighted sample prediction (8.5.3.3.4.3) on the 14-bit arrays; shift1 = 14 - bitDepth = 6 */
      const int c = ci, log2wd = d->sh.wp_log2wd[ci ? 1 : 0] + 6;
      const int w0 = r0 ? d->sh.wp_w[0][m->ref_idx][c] : 0, o0 = r0 ? d->sh.wp_o[0][m->ref_idx][c] : 0;
      const int w1 = r1 ? d->sh.wp_w[1][m->ref_idx1][c] : 0, o1 = r1 ? d->sh.wp_o[1][m->ref_idx1][c] : 0;
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int v;
        if (r0 && r1) v = (t0[y * 64 + x] * w0 + t1[y * 64 + x] * w1 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
        else if (r0) v = ((t0[y * 64 + x] * w0 + (1 << (log2wd - 1))) >> log2wd) + o0;
        else v = ((t1[y * 64 + x] * w1 + (1 << (log2wd - 1))) >> log2wd) + o1;
        dst[y * pic->stride[ci] + x] = (pixel)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    } else
    if (r0 && r1) orc_pred_bi(t0, t1, 64, dst, pic->stride[ci], W, H);
    else orc_pred_uni(r0 ? t0 : t1, 64, dst, pic->stride[ci], W, H);
  }
}

/* mvd_coding() (7.3.8.9) -> one vector difference */
static void decode_mvd(orc_decoder *d, int mvd[2])
{
  orc_cabac_dec *c = &d->cabac;
  int gt0x = orc_cdec_bin(c, CTX_MVD_GT0), gt0y = orc_cdec_bin(c, CTX_MVD_GT0);
  int gt1x = gt0x ? orc_cdec_bin(c, CTX_MVD_GT1) : 0, gt1y = gt0y ? orc_cdec_bin(c, CTX_MVD_GT1) : 0;
  mvd[0] = decode_mvd_comp_abs(d, gt0x, gt1x); if (gt0x && orc_cdec_bypass(c)) mvd[0] = -mvd[0];
  mvd[1] = decode_mvd_comp_abs(d, gt0y, gt1y); if (gt0y && orc_cdec_bypass(c)) mvd[1] = -mvd[1];
}
static int decode_ref_idx(orc_decoder *d, int num_active)
{
  orc_cabac_dec *c = &d->cabac;
  int ref_idx = 0;
  const int mx = num_active - 1;
  while (ref_idx < mx && ref_idx < 2 && orc_cdec_bin(c, CTX_REF_IDX + ref_idx)) ref_idx++;
  if (ref_idx == 2) while (ref_idx < mx && orc_cdec_bypass(c)) ref_idx++;
  return ref_idx;
}

static void prediction_unit(orc_decoder *d, int xcb, int ycb, int ncbs, int xp, int yp, int w, int h, int part_idx, int skip, int *merge_flag_out)
{
  orc_cabac_dec *c = &d->cabac;
  orc_pic *pic = d->cur;
  const int is_b = d->sh.slice_type == SLICE_B;
  orc_mvpred_ctx mc; memset(&mc, 0, sizeof(mc));
  mc.pic = pic; mc.av = d->av; mc.log2_par_mrg_level = d->p->log2_parallel_merge_level;
  mc.max_num_merge_cand = d->sh.max_num_merge_cand; mc.num_ref_idx = d->sh.num_ref_idx_l0;
  mc.cur_poc = pic->poc; memcpy(mc.ref_poc, d->ref_poc, sizeof(mc.ref_poc)); memcpy(mc.ref_lt, pic->ref_lt_list, 16); memcpy(mc.ref_lt1, pic->ref_lt_list1, 16);
  mc.is_b = is_b; mc.num_ref_idx1 = d->sh.num_ref_idx_l1; memcpy(mc.ref_poc1, d->ref_poc1, sizeof(mc.ref_poc1));
  mc.collocated_from_l0 = d->sh.collocated_from_l0; mc.no_backward_pred = d->no_backward_pred;
  mc.col = NULL;
  if (d->sh.slice_temporal_mvp_enabled) {                  /* 8.5.3.2.8: the collocated picture out of list 1 when a B slice says so */
    if (is_b && !d->sh.collocated_from_l0) { if (d->sh.collocated_ref_idx < d->num_ref1) mc.col = d->ref_list1[d->sh.collocated_ref_idx]; }
    else if (d->sh.collocated_ref_idx < d->num_ref) mc.col = d->ref_list0[d->sh.collocated_ref_idx];
  }
  int merge = skip ? 1 : orc_cdec_bin(c, CTX_MERGE_FLAG);
  if (merge_flag_out) *merge_flag_out = merge;
  orc_mvinfo m; memset(&m, 0, sizeof(m)); m.ref_idx = m.ref_idx1 = -1;
  if (merge) {
    int idx = 0;
    if (d->sh.max_num_merge_cand > 1) {
      if (orc_cdec_bin(c, CTX_MERGE_IDX)) { idx = 1; while (idx < d->sh.max_num_merge_cand - 1 && orc_cdec_bypass(c)) idx++; }
    }
    orc_mvcand cand[5];
    orc_merge_candidates(&mc, xcb, ycb, ncbs, xp, yp, w, h, part_idx, d->part_mode, cand);
    m = cand[idx];
    if (m.ref_idx >= 0 && m.ref_idx1 >= 0 && w + h == 12) { m.ref_idx1 = -1; m.mv1[0] = m.mv1[1] = 0; }      /* 8.5.3.2.2 step 9: 8x4 / 4x8 blocks are never bi-predicted */
  } else {
    /* inter_pred_idc (9.3.4.2): "both lists" is asked first (context = coding quadtree depth) unless the block is 8x4 / 4x8, then which list */
    int idc = 0;                                           /* 0 PRED_L0, 1 PRED_L1, 2 PRED_BI */
    if (is_b) {
      if (w + h != 12 && orc_cdec_bin(c, CTX_INTER_PRED_IDC + pic->ct_depth[b4(pic, xcb, ycb)])) idc = 2;
      else idc = orc_cdec_bin(c, CTX_INTER_PRED_IDC + 4);
    }
    int mvd[2], mvp; int16_t cand[2][2];
    if (idc != 1) {
      m.ref_idx = (int8_t)(d->sh.num_ref_idx_l0 > 1 ? decode_ref_idx(d, d->sh.num_ref_idx_l0) : 0);
      decode_mvd(d, mvd);
      mvp = orc_cdec_bin(c, CTX_MVP_FLAG);
      if (m.ref_idx >= d->num_ref) { d->err = ERR_INVALID; return; }
      orc_amvp_candidates_lx(&mc, xcb, ycb, ncbs, xp, yp, w, h, part_idx, 0, m.ref_idx, cand);
      /* 8.5.3.2.6: uLX = (mvp + mvd + 2^16) % 2^16, wrapped to int16 */
      m.mv[0] = (int16_t)(uint16_t)(cand[mvp][0] + mvd[0]); m.mv[1] = (int16_t)(uint16_t)(cand[mvp][1] + mvd[1]);
    }
    if (idc != 0) {
      m.ref_idx1 = (int8_t)(d->sh.num_ref_idx_l1 > 1 ? decode_ref_idx(d, d->sh.num_ref_idx_l1) : 0);
      if (d->sh.mvd_l1_zero && idc == 2) mvd[0] = mvd[1] = 0; else decode_mvd(d, mvd);
      mvp = orc_cdec_bin(c, CTX_MVP_FLAG);
      if (m.ref_idx1 >= d->num_ref1) { d->err = ERR_INVALID; return; }
      orc_amvp_candidates_lx(&mc, xcb, ycb, ncbs, xp, yp, w, h, part_idx, 1, m.ref_idx1, cand);
      m.mv1[0] = (int16_t)(uint16_t)(cand[mvp][0] + mvd[0]); m.mv1[1] = (int16_t)(uint16_t)(cand[mvp][1] + mvd[1]);
    }
  }
  if (m.ref_idx < 0 && m.ref_idx1 < 0) { d->err = ERR_INVALID; return; }
  if ((m.ref_idx >= 0 && (m.ref_idx >= d->num_ref || !d->ref_list0[m.ref_idx])) || (m.ref_idx1 >= 0 && (m.ref_idx1 >= d->num_ref1 || !d->ref_list1[m.ref_idx1]))) { d->err = ERR_INVALID; return; }
  for (int y = yp; y < yp + h; y += 4)
    for (int x = xp; x < xp + w; x += 4) pic->mvf[b4(pic, x, y)] = m;
  /* prediction block edges for deblocking */
  for (int i = 0; i < h; i += 4) pic->edge_v[b4(pic, xp, yp + i)] |= 2;
  for (int i = 0; i < w; i += 4) pic->edge_h[b4(pic, xp + i, yp)] |= 2;
  mc_pu(d, xp, yp, w, h, &m);
}

/* ------------------------------------------------------------------ coding unit */
static void coding_unit(orc_decoder *d, int x0, int y0, int log2cb, int ct_depth)
{
  orc_cabac_dec *c = &d->cabac;
  orc_pic *pic = d->cur;
  const orc_sps *s = d->s;
  int n = 1 << log2cb;
  cu_info cu; memset(&cu, 0, sizeof(cu));
  cu.x0 = x0; cu.y0 = y0; cu.log2cb = log2cb;
  d->cu_transquant_bypass = 0;
  if (d->p->transquant_bypass_enabled) d->cu_transquant_bypass = orc_cdec_bin(c, CTX_TQ_BYPASS);
  int skip = 0;
  if (d->sh.slice_type != SLICE_I) {
    int l = orc_available(&d->av, x0, y0, x0 - 1, y0) && pic->pred_mode[b4(pic, x0 - 1, y0)] == MODE_SKIP;
    int a = orc_available(&d->av, x0, y0, x0, y0 - 1) && pic->pred_mode[b4(pic, x0, y0 - 1)] == MODE_SKIP;
    skip = orc_cdec_bin(c, CTX_SKIP + l + a);
  }
  d->part_mode = PART_2Nx2N; d->intra_split = 0;
  int rqt_root_cbf = 1, merge_2nx2n = 0;
  fill_b4_u8(pic, pic->ct_depth, x0, y0, n, n, ct_depth);
  fill_b4_u8(pic, pic->no_filter, x0, y0, n, n, d->cu_transquant_bypass);
  if (skip) {
    d->cu_pred_mode = MODE_INTER;
    fill_b4_u8(pic, pic->pred_mode, x0, y0, n, n, MODE_SKIP);
    prediction_unit(d, x0, y0, n, x0, y0, n, n, 0, 1, NULL);
    rqt_root_cbf = 0;
  } else {
    d->cu_pred_mode = MODE_INTRA;
    if (d->sh.slice_type != SLICE_I) d->cu_pred_mode = orc_cdec_bin(c, CTX_PRED_MODE) ? MODE_INTRA : MODE_INTER;
    if (d->cu_pred_mode != MODE_INTRA || log2cb == s->log2_min_cb) {
      /* part_mode binarisation, 9.3.3.7 */
      if (d->cu_pred_mode == MODE_INTRA) {
        d->part_mode = orc_cdec_bin(c, CTX_PART_MODE) ? PART_2Nx2N : PART_NxN;
      } else if (orc_cdec_bin(c, CTX_PART_MODE)) {
        d->part_mode = PART_2Nx2N;
      } else if (log2cb == s->log2_min_cb) {
        if (orc_cdec_bin(c, CTX_PART_MODE + 1)) d->part_mode = PART_2NxN;
        else if (log2cb == 3) d->part_mode = PART_Nx2N;
        else d->part_mode = orc_cdec_bin(c, CTX_PART_MODE + 2) ? PART_Nx2N : PART_NxN;
      } else if (!s->amp_enabled) {
        d->part_mode = orc_cdec_bin(c, CTX_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
      } else {
        int horiz = orc_cdec_bin(c, CTX_PART_MODE + 1);
        if (orc_cdec_bin(c, CTX_PART_MODE + 3)) d->part_mode = horiz ? PART_2NxN : PART_Nx2N;
        else {
          int b = orc_cdec_bypass(c);
          d->part_mode = horiz ? (b ? PART_2NxnD : PART_2NxnU) : (b ? PART_nRx2N : PART_nLx2N);
        }
      }
    }
    fill_b4_u8(pic, pic->pred_mode, x0, y0, n, n, d->cu_pred_mode);
    if (d->cu_pred_mode == MODE_INTRA) fill_b4_u8(pic, d->intra4, x0, y0, n, n, 1);
    if (d->cu_pred_mode == MODE_INTRA && d->part_mode == PART_2Nx2N && s->pcm_enabled && log2cb >= s->log2_min_pcm_cb && log2cb <= s->log2_min_pcm_cb + s->log2_diff_max_min_pcm_cb &&
        orc_cdec_terminate(c)) {
      /* pcm_flag = 1 (7.3.8.5, 7.3.8.7): the arithmetic codeword has ended; pcm_alignment_zero_bits, then the samples -- 8.4.4.1? no prediction, no residual:
       * recSamples = pcm_sample << (BitDepth - PcmBitDepth) -- and the arithmetic decoder starts again behind them with the contexts as they are (9.3.2.5).
       * IntraPredModeY of the unit is DC for its neighbours (8.4.2); QpY is the predicted one (no cu_qp_delta); pcm_loop_filter_disabled_flag keeps the loop filters off
       * its samples like cu_transquant_bypass_flag does */
      const uint8_t *base = c->br.buf; const size_t blen = c->br.len; size_t pos = orc_cdec_bytes_consumed(c);
      const size_t need = ((size_t)n * n * s->pcm_bit_depth_luma + (size_t)n * n / 2 * s->pcm_bit_depth_chroma) / 8;
      if (pos + need > blen) { d->err = ERR_INVALID; return; }
      orc_bitr pr; orc_br_init(&pr, base + pos, need);
      for (int ci = 0; ci < 3; ci++) {
        const int sh2 = ci ? 1 : 0, m = n >> sh2, depth = ci ? s->pcm_bit_depth_chroma : s->pcm_bit_depth_luma;
        for (int y = 0; y < m; y++) for (int x = 0; x < m; x++) {
          const int v = (int)orc_br_get(&pr, depth) << (8 - depth);
          if (((y0 >> sh2) + y) < (pic->h >> sh2) && ((x0 >> sh2) + x) < (pic->w >> sh2)) pic->plane[ci][((y0 >> sh2) + y) * pic->stride[ci] + (x0 >> sh2) + x] = (pixel)v;
        }
      }
      orc_cdec_start(c, base + pos + need, blen - pos - need);
      fill_b4_u8(pic, pic->intra_mode, x0, y0, n, n, 1);
      if (s->pcm_loop_filter_disabled) fill_b4_u8(pic, pic->no_filter, x0, y0, n, n, 1);
      for (int i = 0; i < n; i += 4) {
        if (y0 + i < pic->h) pic->edge_v[b4(pic, x0, y0 + i)] |= 3;
        if (x0 + i < pic->w) pic->edge_h[b4(pic, x0 + i, y0)] |= 3;
      }
      d->qp_y = (d->qp_y_pred + d->cu_qp_delta_val + 52) % 52;
      for (int y = y0; y < y0 + n && y < pic->h; y += 4)
        for (int x = x0; x < x0 + n && x < pic->w; x += 4) { pic->qp_y[b4(pic, x, y)] = (int8_t)d->qp_y; pic->tu_nz[b4(pic, x, y)] = 0; }
      d->last_qp_y = d->qp_y;
      return;
    }
    if (d->cu_pred_mode == MODE_INTRA) {
      d->intra_split = (d->part_mode == PART_NxN);
      int parts = d->intra_split ? 2 : 1, pb = n / parts;
      int prev[4], k = 0;
      for (int j = 0; j < parts; j++) for (int i = 0; i < parts; i++) prev[k++] = orc_cdec_bin(c, CTX_PREV_INTRA);
      k = 0;
      for (int j = 0; j < parts; j++)
        for (int i = 0; i < parts; i++, k++) {
          int xp = x0 + i * pb, yp = y0 + j * pb;
          /* 8.4.2 candidate modes */
          int ca = 1, cb = 1;
          if (orc_available(&d->av, xp, yp, xp - 1, yp) && pic->pred_mode[b4(pic, xp - 1, yp)] == MODE_INTRA) ca = pic->intra_mode[b4(pic, xp - 1, yp)];
          if (orc_available(&d->av, xp, yp, xp, yp - 1) && pic->pred_mode[b4(pic, xp, yp - 1)] == MODE_INTRA &&
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
            int idx = 0;
            if (orc_cdec_bypass(c)) { idx = 1; if (orc_cdec_bypass(c)) idx = 2; }
            mode = cand[idx];
          } else {
            mode = (int)orc_cdec_bypass_bits(c, 5);
            if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
            if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
            if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
            for (int q = 0; q < 3; q++) if (mode >= cand[q]) mode++;
          }
          cu.intra_modes[k] = mode;
          fill_b4_u8(pic, pic->intra_mode, xp, yp, pb, pb, mode);
        }
      int icpm = 4;
      if (orc_cdec_bin(c, CTX_CHROMA_MODE)) icpm = (int)orc_cdec_bypass_bits(c, 2);
      static const int cm[4] = { 0, 26, 10, 1 };
      if (icpm == 4) cu.chroma_mode = cu.intra_modes[0];
      else { cu.chroma_mode = cm[icpm]; if (cu.chroma_mode == cu.intra_modes[0]) cu.chroma_mode = 34; }
    } else {
      int h = n / 2, q = n / 4, mf = 0;
      switch (d->part_mode) {
      case PART_2Nx2N: prediction_unit(d, x0, y0, n, x0, y0, n, n, 0, 0, &merge_2nx2n); break;
      case PART_2NxN:  prediction_unit(d, x0, y0, n, x0, y0, n, h, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0, y0 + h, n, h, 1, 0, &mf); break;
      case PART_Nx2N:  prediction_unit(d, x0, y0, n, x0, y0, h, n, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0 + h, y0, h, n, 1, 0, &mf); break;
      case PART_2NxnU: prediction_unit(d, x0, y0, n, x0, y0, n, q, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0, y0 + q, n, n - q, 1, 0, &mf); break;
      case PART_2NxnD: prediction_unit(d, x0, y0, n, x0, y0, n, n - q, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0, y0 + n - q, n, q, 1, 0, &mf); break;
      case PART_nLx2N: prediction_unit(d, x0, y0, n, x0, y0, q, n, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0 + q, y0, n - q, n, 1, 0, &mf); break;
      case PART_nRx2N: prediction_unit(d, x0, y0, n, x0, y0, n - q, n, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0 + n - q, y0, q, n, 1, 0, &mf); break;
      default: /* NxN */
        prediction_unit(d, x0, y0, n, x0, y0, h, h, 0, 0, &mf); prediction_unit(d, x0, y0, n, x0 + h, y0, h, h, 1, 0, &mf);
        prediction_unit(d, x0, y0, n, x0, y0 + h, h, h, 2, 0, &mf); prediction_unit(d, x0, y0, n, x0 + h, y0 + h, h, h, 3, 0, &mf); break;
      }
      if (!(d->part_mode == PART_2Nx2N && merge_2nx2n)) rqt_root_cbf = orc_cdec_bin(c, CTX_RQT_ROOT_CBF);
    }
  }
  if (d->err) return;
  /* coding block edges are both transform and prediction edges */
  for (int i = 0; i < n; i += 4) {
    if (y0 + i < pic->h) pic->edge_v[b4(pic, x0, y0 + i)] |= 3;
    if (x0 + i < pic->w) pic->edge_h[b4(pic, x0 + i, y0)] |= 3;
  }
  d->qp_y = d->qp_y_pred + d->cu_qp_delta_val;   /* CuQpDeltaVal of the current quantisation group so far */
  d->qp_y = (d->qp_y + 52) % 52;
  if (rqt_root_cbf) {
    d->max_trafo_depth = (d->cu_pred_mode == MODE_INTRA) ? s->max_th_depth_intra + d->intra_split : s->max_th_depth_inter;
    transform_tree(d, &cu, x0, y0, x0, y0, log2cb, 0, 0, 0, 0);
  }
  for (int y = y0; y < y0 + n && y < pic->h; y += 4)
    for (int x = x0; x < x0 + n && x < pic->w; x += 4) pic->qp_y[b4(pic, x, y)] = (int8_t)d->qp_y;
  d->last_qp_y = d->qp_y;
}

static void coding_quadtree(orc_decoder *d, int x0, int y0, int log2cb, int depth)
{
  orc_cabac_dec *c = &d->cabac;
  orc_pic *pic = d->cur;
  const orc_sps *s = d->s;
  int n = 1 << log2cb, split;
  if (d->err) return;
  if (x0 + n <= s->width && y0 + n <= s->height && log2cb > s->log2_min_cb) {
    int l = orc_available(&d->av, x0, y0, x0 - 1, y0) && pic->ct_depth[b4(pic, x0 - 1, y0)] > depth;
    int a = orc_available(&d->av, x0, y0, x0, y0 - 1) && pic->ct_depth[b4(pic, x0, y0 - 1)] > depth;
    split = orc_cdec_bin(c, CTX_SPLIT_CU + l + a);
  } else split = (log2cb > s->log2_min_cb);
  int log2_qg = s->ctb_log2 - d->p->diff_cu_qp_delta_depth;
  if (d->p->cu_qp_delta_enabled && log2cb >= log2_qg) {
    d->is_cu_qp_delta_coded = 0; d->cu_qp_delta_val = 0;
    d->qg_x = x0; d->qg_y = y0;
    derive_qp_pred(d, x0, y0, 0);
  }
  if (split) {
    int h = n >> 1;
    coding_quadtree(d, x0, y0, log2cb - 1, depth + 1);
    if (x0 + h < s->width) coding_quadtree(d, x0 + h, y0, log2cb - 1, depth + 1);
    if (y0 + h < s->height) coding_quadtree(d, x0, y0 + h, log2cb - 1, depth + 1);
    if (x0 + h < s->width && y0 + h < s->height) coding_quadtree(d, x0 + h, y0 + h, log2cb - 1, depth + 1);
  } else {
    coding_unit(d, x0, y0, log2cb, depth);
  }
}

/* ------------------------------------------------------------------ picture management */
static orc_pic *find_poc(orc_decoder *d, int poc)
{
  for (int i = 0; i < MAX_DPB; i++) if (d->dpb[i].in_use && d->dpb[i].is_ref && d->dpb[i].poc == poc && &d->dpb[i] != d->cur) return &d->dpb[i];
  return NULL;
}

static orc_pic *alloc_pic(orc_decoder *d, int w, int h)
{
  for (int i = 0; i < MAX_DPB; i++) {
    orc_pic *p = &d->dpb[i];
    if (p->in_use && (p->is_ref || p->needed_for_output || p == d->last_output || d->pre_ref[i])) continue;      /* (pre_ref: still a source for stand-ins of the picture being started) */
    if (p->plane[0] && (p->w != w || p->h != h)) orc_pic_free(p);
    if (!p->plane[0]) { if (orc_pic_alloc(p, w, h)) return NULL; }
    else orc_pic_reset_side(p);
    p->in_use = 1;
    return p;
  }
  return NULL;
}

static void setup_tiles(orc_decoder *d)
{
  const orc_sps *s = d->s; const orc_pps *p = d->p;
  int wc = s->pic_w_ctbs, hc = s->pic_h_ctbs, nc = p->num_tile_columns, nr = p->num_tile_rows;
  size_t n = (size_t)wc * hc;
  if (n > d->ctb_cap) {
    d->sao = (orc_sao_params *)realloc(d->sao, n * sizeof(orc_sao_params));
    d->ctb_slice = (int32_t *)realloc(d->ctb_slice, n * sizeof(int32_t));
    d->ctb_tile = (int16_t *)realloc(d->ctb_tile, n * sizeof(int16_t));
    d->ctb_lfx = (uint8_t *)realloc(d->ctb_lfx, n); d->ctb_nb = (uint8_t *)realloc(d->ctb_nb, n);
    d->ts_to_rs = (int *)realloc(d->ts_to_rs, n * sizeof(int));
    d->rs_to_ts = (int *)realloc(d->rs_to_ts, n * sizeof(int));
    d->tile_first_x = (int *)realloc(d->tile_first_x, n * sizeof(int));
    d->ctb_cap = n;
  }
  /* 6.5.1 */
  d->col_bd[0] = 0; d->row_bd[0] = 0;
  for (int i = 0; i < nc; i++) {
    int wcol = p->uniform_spacing ? ((i + 1) * wc) / nc - (i * wc) / nc : (i < nc - 1 ? p->column_width[i] : wc - d->col_bd[i]);
    d->col_bd[i + 1] = d->col_bd[i] + wcol;
  }
  for (int j = 0; j < nr; j++) {
    int hr = p->uniform_spacing ? ((j + 1) * hc) / nr - (j * hc) / nr : (j < nr - 1 ? p->row_height[j] : hc - d->row_bd[j]);
    d->row_bd[j + 1] = d->row_bd[j] + hr;
  }
  int ts = 0;
  for (int j = 0; j < nr; j++)
    for (int i = 0; i < nc; i++)
      for (int y = d->row_bd[j]; y < d->row_bd[j + 1]; y++)
        for (int x = d->col_bd[i]; x < d->col_bd[i + 1]; x++) {
          int rs = y * wc + x;
          d->ts_to_rs[ts] = rs; d->rs_to_ts[rs] = ts; d->ctb_tile[rs] = (int16_t)(j * nc + i);
          d->tile_first_x[rs] = d->col_bd[i];
          ts++;
        }
  for (size_t i = 0; i < n; i++) d->ctb_slice[i] = -1;
  d->lf_restricted = 0;
  d->sao_used = 0;
}

/* C.5.2.4 "bumping": of the pictures waiting for output the one with the smallest POC goes out */
static int waiting_for_output(const orc_decoder *d)
{
  int n = 0;
  for (int i = 0; i < MAX_DPB; i++) n += d->dpb[i].in_use && d->dpb[i].needed_for_output && !d->dpb[i].out_queued;
  return n;
}
static void bump(orc_decoder *d)
{
  orc_pic *best = NULL;
  for (int i = 0; i < MAX_DPB; i++) {
    orc_pic *q = &d->dpb[i];
    if (q->in_use && q->needed_for_output && !q->out_queued && (!best || q->poc < best->poc)) best = q;
  }
  if (best && d->out_n < MAX_DPB + 1) { best->out_queued = 1; d->out_queue[d->out_n++] = best; }
}
void orc_dec_flush(orc_decoder *d) { if (d->pic_active) return; while (waiting_for_output(d)) bump(d); }

static int start_picture(orc_decoder *d)
{
  const orc_sps *s = d->s; orc_slice_hdr *sh = &d->sh;
  for (int i = 0; i < MAX_DPB; i++) { d->dpb[i].stand_in_fresh = 0; d->pre_ref[i] = (uint8_t)(d->dpb[i].in_use && d->dpb[i].is_ref); }
  int irap = d->nal_type >= NAL_BLA_W_LP && d->nal_type <= NAL_RSV_IRAP_VCL23;
  int idr = d->nal_type == NAL_IDR_W_RADL || d->nal_type == NAL_IDR_N_LP;
  const int bla = d->nal_type >= NAL_BLA_W_LP && d->nal_type <= NAL_BLA_N_LP;
  if (!d->seen_irap && !irap) return 0;           /* cannot start decoding before a random access point */
  /* 8.1.3: NoRaslOutputFlag -- an IDR or BLA picture, or a CRA picture that is the first of the stream or follows an end of sequence NAL unit.  The RASL
   * pictures associated with such a picture predict from pictures that are not there: they are not decoded and not output */
  const int no_rasl_out = idr || bla || (irap && (!d->seen_irap || d->after_eos));
  if (irap) d->skip_rasl = no_rasl_out;
  if ((d->nal_type == NAL_RASL_N || d->nal_type == NAL_RASL_R) && d->skip_rasl) return 0;
  /* 8.3.1 picture order count */
  int poc;
  int max_lsb = 1 << s->log2_max_poc_lsb;
  if (idr) poc = 0;
  else {
    int prev_lsb = d->prev_tid0_poc & (max_lsb - 1), prev_msb = d->prev_tid0_poc - prev_lsb, msb;
    if (irap && no_rasl_out) msb = 0;
    else if (sh->poc_lsb < prev_lsb && prev_lsb - sh->poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (sh->poc_lsb > prev_lsb && sh->poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    else msb = prev_msb;
    poc = msb + sh->poc_lsb;
  }
  if (no_rasl_out) {                              /* 8.3.2: every reference picture in the DPB is marked unused */
    for (int i = 0; i < MAX_DPB; i++) d->dpb[i].is_ref = 0;
  }
  /* C.5.2.2: an IRAP picture that starts a coded video sequence empties the DPB first -- in output order, or (no_output_of_prior_pics_flag of an IDR or BLA
   * picture) without output of what still waits.  (A CRA picture gets here behind an end of sequence NAL unit only, which has output everything: see orc_dec_decode_nal.) */
  if (no_rasl_out) {
    if (d->seen_irap && !d->after_eos && (idr || bla) && sh->no_output_of_prior_pics) {
      for (int i = 0; i < MAX_DPB; i++) { orc_pic *q = &d->dpb[i]; if (q->in_use && !q->out_queued) q->needed_for_output = 0; }
    }
    while (waiting_for_output(d)) bump(d);
    d->cvs++;
  }
  d->seen_irap = 1; d->after_eos = 0;
  /* prevTid0Pic (8.3.1): TemporalId 0 and no RASL, RADL or sub-layer non-reference picture */
  if (d->tid == 0 && (d->nal_type >= NAL_BLA_W_LP || ((d->nal_type & 1) && d->nal_type < NAL_RADL_N))) d->prev_tid0_poc = poc;
  d->cur = alloc_pic(d, s->width, s->height);
  if (!d->cur) return ERR_INVALID;
  d->cur->poc = poc; d->cur->pts = d->cur_pts;
  d->cur->is_ref = 1; d->cur->is_lt = 0; d->cur->needed_for_output = 0; d->cur->out_queued = 0;
  /* 8.3.2 reference picture set: everything not in the RPS is marked unused */
  if (!idr) {
    for (int i = 0; i < MAX_DPB; i++) {
      orc_pic *q = &d->dpb[i];
      if (!q->in_use || !q->is_ref || q == d->cur) continue;
      int keep = 0;
      /* 8.3.2: the long-term entries first, among ALL reference pictures -- by the POC's LSBs, or by the whole POC when delta_poc_msb_present_flag says how many
       * LSB cycles back; what they name is a long-term reference picture from now on.  Then the short-term entries among the rest. */
      for (int k = 0; k < sh->num_lt; k++) {
        const int full = poc - sh->lt_msb_cycle[k] * max_lsb - (poc & (max_lsb - 1)) + sh->lt_poc_lsb[k];
        if (sh->lt_msb_present[k] ? q->poc == full : (q->poc & (max_lsb - 1)) == sh->lt_poc_lsb[k]) { keep = 1; q->is_lt = 1; }
      }
      if (!keep && !q->is_lt) {
        for (int k = 0; k < sh->st_rps.num_negative; k++) if (q->poc == poc + sh->st_rps.delta_poc_s0[k]) keep = 1;
        for (int k = 0; k < sh->st_rps.num_positive; k++) if (q->poc == poc + sh->st_rps.delta_poc_s1[k]) keep = 1;
      }
      if (!keep) q->is_ref = 0;
    }
    /* C.5.2.2: room for the current picture -- more pictures waiting than may be reordered, or no free picture buffer */
    for (;;) {
      int full = 0;
      for (int i = 0; i < MAX_DPB; i++) { const orc_pic *q = &d->dpb[i]; full += q != d->cur && q->in_use && (q->is_ref || (q->needed_for_output && !q->out_queued)); }
      const int w = waiting_for_output(d);
      if (w > 0 && (w > s->max_num_reorder || full >= s->max_dec_pic_buffering)) bump(d); else break;
    }
  }
  d->cur->needed_for_output = sh->pic_output_flag;
  size_t need_v = (size_t)(s->width / 8) * (s->height / 4), need_h = (size_t)(s->width / 4) * (s->height / 8);
  if (need_v + need_h > d->bs_cap) {
    d->bs_v = (uint8_t *)realloc(d->bs_v, need_v); d->bs_h = (uint8_t *)realloc(d->bs_h, need_h); d->bs_cap = need_v + need_h;
  }
  size_t pc = (size_t)s->width * s->height;
  if (pc > d->predeblock_cap) {
    d->predeblock[0] = (pixel *)realloc(d->predeblock[0], pc);
    d->predeblock[1] = (pixel *)realloc(d->predeblock[1], pc / 4);
    d->predeblock[2] = (pixel *)realloc(d->predeblock[2], pc / 4);
    d->predeblock_cap = pc;
  }
  { const size_t n4 = (size_t)d->cur->b4_w * d->cur->b4_h; if (n4 > d->intra4_cap) { d->intra4 = (uint8_t *)realloc(d->intra4, n4); d->intra4_cap = n4; } memset(d->intra4, 0, n4); }
  setup_tiles(d);
  d->ctbs_decoded = 0;
  d->pic_active = 1;
  return 1;
}

/* 8.3.4: RefPicList0 = the used pictures before the current one (nearest first), then those after it, repeated until the list is full; RefPicList1
 * the other way round.  No list modification, no long-term pictures. */
/* A picture the reference picture set says the current picture predicts from is not there -- its access unit never arrived (the streams come over RTP).  What a
 * decoder does then is not the standard's business; this project's rule ("concealment v2"): a stand-in with the missing picture order count joins the DPB, marked as
 * the real picture would be, without motion (a block of it gives no temporal candidate), never output, and decoding goes on.  Its samples are a COPY of the reference
 * picture that is nearest in output order among those the DPB held when the current picture arrived -- before its reference picture set was applied: with one
 * reference picture per picture, as Kvazaar codes by default, the set names the lost picture and nothing else -- of two equally near the earlier one: the repeated
 * frame a viewer hardly notices; mid-grey when there is none (libavcodec's generate_missing_ref always takes grey).  The product's decoder does the same
 * (csrc/decoder.hip conceal_ref). */
static orc_pic *missing_ref(orc_decoder *d, int poc, int is_lt)
{
  const orc_pic *src = NULL;
  for (int i = 0; i < MAX_DPB; i++) {
    const orc_pic *q = &d->dpb[i];
    if (!d->pre_ref[i] || q == d->cur || q->stand_in_fresh) continue;
    const int dq = q->poc > poc ? q->poc - poc : poc - q->poc, ds = src ? (src->poc > poc ? src->poc - poc : poc - src->poc) : 0;
    if (!src || dq < ds || (dq == ds && q->poc < src->poc)) src = q;
  }
  if (getenv("ORC_CONCEAL_GREY")) src = NULL;      /* (measurement aid: what the grey rule would show, HISTORY.md) */
  orc_pic *p = alloc_pic(d, d->s->width, d->s->height);
  if (!p) return NULL;
  orc_pic_reset_side(p);
  for (int c = 0; c < 3; c++) {
    const size_t n = (size_t)p->stride[c] * (size_t)(c ? p->h / 2 : p->h);
    if (src) memcpy(p->plane[c], src->plane[c], n); else memset(p->plane[c], 128, n);
  }
  p->poc = poc; p->pts = 0; p->is_ref = 1; p->is_lt = is_lt; p->needed_for_output = 0; p->out_queued = 0; p->slice_type = SLICE_I; p->stand_in_fresh = 1;
  d->concealed++;
  return p;
}

static int build_ref_list(orc_decoder *d)
{
  orc_slice_hdr *sh = &d->sh;
  d->num_ref = d->num_ref1 = 0; d->no_backward_pred = 1;
  for (int i = 0; i < 16; i++) d->cur->ref_poc_list[i] = d->cur->ref_poc_list1[i] = d->cur->poc;      /* unused entries: never compared */
  if (sh->slice_type == SLICE_I) return 0;
  orc_pic *before[16], *after[16], *lt[16]; int nb = 0, na = 0, nl = 0;
  int poc = d->cur->poc;
  const int max_lsb = 1 << d->s->log2_max_poc_lsb;
  for (int k = 0; k < sh->st_rps.num_negative && nb < 16; k++) if (sh->st_rps.used_s0[k]) {
    before[nb] = find_poc(d, poc + sh->st_rps.delta_poc_s0[k]);
    if (!before[nb]) before[nb] = missing_ref(d, poc + sh->st_rps.delta_poc_s0[k], 0); else if (before[nb]->is_lt) before[nb] = NULL;
    nb++;
  }
  for (int k = 0; k < sh->st_rps.num_positive && na < 16; k++) if (sh->st_rps.used_s1[k]) {
    after[na] = find_poc(d, poc + sh->st_rps.delta_poc_s1[k]);
    if (!after[na]) after[na] = missing_ref(d, poc + sh->st_rps.delta_poc_s1[k], 0); else if (after[na]->is_lt) after[na] = NULL;
    na++;
  }
  for (int k = 0; k < sh->num_lt && nl < 16; k++) if (sh->lt_used[k]) {      /* RefPicSetLtCurr (8.3.2) */
    const int full = poc - sh->lt_msb_cycle[k] * max_lsb - (poc & (max_lsb - 1)) + sh->lt_poc_lsb[k];
    lt[nl] = NULL;
    for (int i = 0; i < MAX_DPB; i++) {
      orc_pic *q = &d->dpb[i];
      if (q->in_use && q->is_ref && q->is_lt && q != d->cur && (sh->lt_msb_present[k] ? q->poc == full : (q->poc & (max_lsb - 1)) == sh->lt_poc_lsb[k])) lt[nl] = q;
    }
    if (!lt[nl]) {      /* (known by its LSBs alone: the nearest picture order count before the current one that has them) */
      int at = full; if (!sh->lt_msb_present[k]) { at = poc - (poc & (max_lsb - 1)) + sh->lt_poc_lsb[k]; if (at >= poc) at -= max_lsb; }
      lt[nl] = missing_ref(d, at, 1);
    }
    nl++;
  }
  const int nc = nb + na + nl;
  if (nc == 0 || nc > 16) return ERR_INVALID;
  memset(d->cur->ref_lt_list, 0, sizeof(d->cur->ref_lt_list)); memset(d->cur->ref_lt_list1, 0, sizeof(d->cur->ref_lt_list1));
  for (int i = 0; i < sh->num_ref_idx_l0; i++) {
    const int k = sh->rpl_mod_flag[0] ? sh->list_entry[0][i] : i % nc;      /* 8.3.4: an entry of the temporary list (before, after, long-term, before, ...) */
    d->ref_list0[i] = k < nb ? before[k] : (k < nb + na ? after[k - nb] : lt[k - nb - na]);
    if (!d->ref_list0[i]) return ERR_INVALID;      /* missing reference picture */
    d->cur->ref_lt_list[i] = (uint8_t)(k >= nb + na);
    d->ref_poc[i] = d->cur->ref_poc_list[i] = d->ref_list0[i]->poc;
    if (d->ref_poc[i] > poc) d->no_backward_pred = 0;
  }
  d->num_ref = sh->num_ref_idx_l0;
  if (sh->slice_type == SLICE_B) {
    for (int i = 0; i < sh->num_ref_idx_l1; i++) {
      const int k = sh->rpl_mod_flag[1] ? sh->list_entry[1][i] : i % nc;
      d->ref_list1[i] = k < na ? after[k] : (k < na + nb ? before[k - na] : lt[k - na - nb]);
      if (!d->ref_list1[i]) return ERR_INVALID;
      d->cur->ref_lt_list1[i] = (uint8_t)(k >= na + nb);
      d->ref_poc1[i] = d->cur->ref_poc_list1[i] = d->ref_list1[i]->poc;
      if (d->ref_poc1[i] > poc) d->no_backward_pred = 0;
    }
    d->num_ref1 = sh->num_ref_idx_l1;
  }
  return 0;
}

static void finish_picture(orc_decoder *d)
{
  orc_pic *pic = d->cur;
  d->last_finished = pic;
  const orc_sps *s = d->s;
  for (int ci = 0; ci < 3; ci++) {
    size_t n = (size_t)(ci ? s->width / 2 : s->width) * (ci ? s->height / 2 : s->height);
    memcpy(d->predeblock[ci], pic->plane[ci], n);
  }
  /* in-loop filtering across slice and tile boundaries (7.4.3.3.1 loop_filter_across_tiles_enabled_flag, 7.4.7.1 slice_loop_filter_across_slices_enabled_flag: the
   * left and upper boundaries of the slice that has it, i.e. of two slices the LATER one's flag): per CTB, which neighbouring CTBs' samples may be used.  Deblocking
   * (8.7.2.3 filterEdgeFlag) then does not see the edges on a closed boundary at all; SAO (8.7.3.2) leaves a sample alone whose neighbour lies across one. */
  const uint8_t *nb_map = NULL;
  if (d->lf_restricted || (d->p->tiles_enabled && !d->p->loop_filter_across_tiles)) {
    const int wc = s->pic_w_ctbs, hc = s->pic_h_ctbs, cu = (1 << s->ctb_log2) >> 2;
    for (int cy = 0; cy < hc; cy++) for (int cx = 0; cx < wc; cx++) {
      const int c = cy * wc + cx; uint8_t m = 0xff;
      for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        const int nx = cx + dx, ny = cy + dy;
        if ((!dx && !dy) || nx < 0 || ny < 0 || nx >= wc || ny >= hc) continue;
        const int n = ny * wc + nx, later = d->rs_to_ts[n] > d->rs_to_ts[c] ? n : c;
        if ((d->ctb_tile[c] != d->ctb_tile[n] && !d->p->loop_filter_across_tiles) || (d->ctb_slice[c] != d->ctb_slice[n] && !d->ctb_lfx[later])) m &= (uint8_t)~(1u << orc_lf_neighbour_bit(dx, dy));
      }
      d->ctb_nb[c] = m;
      for (int k = 0; k < cu; k++) {
        const int ux = cx * cu, uy = cy * cu;
        if (!((m >> orc_lf_neighbour_bit(-1, 0)) & 1) && uy + k < pic->b4_h && ux < pic->b4_w) pic->edge_v[(uy + k) * pic->b4_w + ux] = 0;
        if (!((m >> orc_lf_neighbour_bit(0, -1)) & 1) && ux + k < pic->b4_w && uy < pic->b4_h) pic->edge_h[uy * pic->b4_w + ux + k] = 0;
      }
    }
    nb_map = d->ctb_nb;
  }
  if (!d->sh.slice_deblocking_disabled) {
    orc_deblock_ctx db; memset(&db, 0, sizeof(db));
    orc_compute_bs(pic, d->bs_v, d->bs_h);
    db.w = pic->w; db.h = pic->h;
    for (int i = 0; i < 3; i++) { db.plane[i] = pic->plane[i]; db.stride[i] = pic->stride[i]; }
    db.bs_v = d->bs_v; db.bs_stride_v = pic->w / 8; db.bs_h = d->bs_h; db.bs_stride_h = pic->w / 4;
    db.qp_y = pic->qp_y; db.qp_stride = pic->b4_w;
    db.no_filter = pic->no_filter; db.nf_stride = pic->b4_w;
    db.beta_offset_div2 = d->sh.beta_offset_div2; db.tc_offset_div2 = d->sh.tc_offset_div2;
    db.cb_qp_offset = d->p->cb_qp_offset; db.cr_qp_offset = d->p->cr_qp_offset;
    orc_deblock_picture(&db);
  }
  if (s->sao_enabled && d->sao_used) {                /* 8.7.3 on the deblocked picture */
    orc_sao_ctx sc; memset(&sc, 0, sizeof(sc));
    sc.w = pic->w; sc.h = pic->h; sc.ctb_log2 = s->ctb_log2; sc.pic_w_ctbs = s->pic_w_ctbs; sc.params = d->sao;
    sc.ctb_slice = d->ctb_slice; sc.ctb_tile = d->ctb_tile;
    sc.across_slices = 1; sc.across_tiles = 1; sc.ctb_nb = nb_map;
    sc.no_filter = pic->no_filter; sc.nf_stride = pic->b4_w;
    for (int i = 0; i < 3; i++) {
      size_t n = (size_t)pic->stride[i] * (i ? pic->h / 2 : pic->h);
      d->sao_in[i] = (pixel *)realloc(d->sao_in[i], n); memcpy(d->sao_in[i], pic->plane[i], n);
      sc.src[i] = d->sao_in[i]; sc.dst[i] = pic->plane[i]; sc.stride[i] = pic->stride[i];
    }
    orc_sao_picture(&sc);
  }
  d->pic_active = 0;
  while (waiting_for_output(d) > s->max_num_reorder) bump(d);   /* C.5.2.3: a low-delay stream (no reordering) hands every picture on at once */
}

/* ------------------------------------------------------------------ slice data 7.3.8.1 */
static int decode_slice_data(orc_decoder *d, const uint8_t *data, size_t len)
{
  const orc_sps *s = d->s; const orc_pps *p = d->p; orc_slice_hdr *sh = &d->sh;
  orc_pic *pic = d->cur;
  int wc = s->pic_w_ctbs, total = wc * s->pic_h_ctbs;
  int init_type = sh->slice_type == SLICE_I ? 0 : (sh->slice_type == SLICE_P ? (sh->cabac_init_flag ? 2 : 1) : (sh->cabac_init_flag ? 1 : 2));
  size_t pos = 0;
  int ts = d->rs_to_ts[sh->slice_segment_address];
  if (!sh->dependent_slice_segment) d->slice_addr_rs = sh->slice_segment_address;
  int slice_addr = d->slice_addr_rs;
  int first = 1;
  memset(&d->av, 0, sizeof(d->av));
  d->av.pic_w = s->width; d->av.pic_h = s->height; d->av.ctb_log2 = s->ctb_log2; d->av.pic_w_ctbs = wc;
  d->av.ctb_slice = d->ctb_slice; d->av.ctb_tile = d->ctb_tile;
  orc_cdec_start(&d->cabac, data, len);
  for (;;) {
    if (ts >= total) return ERR_INVALID;
    int rs = d->ts_to_rs[ts], cx = rs % wc, cy = rs / wc;
    int first_in_tile = (ts == 0) || d->ctb_tile[rs] != d->ctb_tile[d->ts_to_rs[ts - 1]];
    int row_start = (cx == d->tile_first_x[rs]);
    d->ctb_slice[rs] = slice_addr;
    d->ctb_lfx[rs] = (uint8_t)sh->loop_filter_across_slices; if (!sh->loop_filter_across_slices) d->lf_restricted = 1;
    int new_qg_row = 0;
    if ((first && !sh->dependent_slice_segment) || first_in_tile) {
      orc_cabac_init_contexts(d->cabac.ctx, init_type, sh->slice_qp);
      new_qg_row = 1;
    } else if (first && sh->dependent_slice_segment && !(p->entropy_coding_sync_enabled && row_start)) {
      memcpy(d->cabac.ctx, d->ds_ctx, sizeof(d->ds_ctx));        /* 9.3.1: a dependent slice segment goes on where the previous segment stopped */
    } else if (p->entropy_coding_sync_enabled && row_start) {
      int xt = ((cx + 1) << s->ctb_log2), yt = ((cy - 1) << s->ctb_log2);
      if (orc_available(&d->av, cx << s->ctb_log2, cy << s->ctb_log2, xt, yt)) memcpy(d->cabac.ctx, d->wpp_ctx, sizeof(d->wpp_ctx));
      else if (first && sh->dependent_slice_segment && wc >= 2) memcpy(d->cabac.ctx, d->ds_ctx, sizeof(d->ds_ctx));    /* (9.3.1, the nesting of the 2nd edition on; HM: TDecSlice loads the segment-end state first, the row-start rule overrides it only when T is available) */
      else orc_cabac_init_contexts(d->cabac.ctx, init_type, sh->slice_qp);
      new_qg_row = 1;
    }
    if (new_qg_row) d->last_qp_y = sh->slice_qp;      /* qPY_PREV at first QG of slice / tile / CTB row (WPP) */
    first = 0;
    if (!p->cu_qp_delta_enabled) { d->qp_y_pred = sh->slice_qp; d->cu_qp_delta_val = 0; }
    if (sh->sao_luma || sh->sao_chroma) {              /* 7.3.8.2: sao() before the coding quadtree */
      const orc_sao_params *left = NULL, *up = NULL;
      if (cx > 0 && d->ctb_slice[rs - 1] == slice_addr && d->ctb_tile[rs - 1] == d->ctb_tile[rs]) left = &d->sao[rs - 1];
      if (cy > 0 && d->ctb_slice[rs - wc] == slice_addr && d->ctb_tile[rs - wc] == d->ctb_tile[rs]) up = &d->sao[rs - wc];
      orc_sao_parse(&d->cabac, &d->sao[rs], left, up, sh->sao_luma, sh->sao_chroma);
      d->sao_used = 1;
    } else memset(&d->sao[rs], 0, sizeof(d->sao[rs]));
    coding_quadtree(d, cx << s->ctb_log2, cy << s->ctb_log2, s->ctb_log2, 0);
    if (d->err) return d->err;
    if (d->cabac.br.error) return ERR_INVALID;
    d->ctbs_decoded++;
    if (p->entropy_coding_sync_enabled && cx == d->tile_first_x[rs] + 1 - 0 && (cx - d->tile_first_x[rs]) == 1)
      memcpy(d->wpp_ctx, d->cabac.ctx, sizeof(d->wpp_ctx));
    int end_of_slice = orc_cdec_terminate(&d->cabac);
    ts++;
    if (end_of_slice) { memcpy(d->ds_ctx, d->cabac.ctx, sizeof(d->ds_ctx)); break; }
    if (ts >= total) return ERR_INVALID;
    int nrs = d->ts_to_rs[ts];
    int tile_change = d->ctb_tile[nrs] != d->ctb_tile[rs];
    if ((p->tiles_enabled && tile_change) ||
        (p->entropy_coding_sync_enabled && (tile_change || (nrs % wc) == d->tile_first_x[nrs]))) {
      if (!orc_cdec_terminate(&d->cabac)) return ERR_INVALID;     /* end_of_subset_one_bit */
      pos = (size_t)(d->cabac.br.buf - data) + orc_cdec_bytes_consumed(&d->cabac);      /* (a PCM unit has moved the decoder's window on: its start is no longer the substream's) */
      if (pos >= len) return ERR_INVALID;
      orc_ctx keep[CTX_COUNT]; memcpy(keep, d->cabac.ctx, sizeof(keep));
      orc_cdec_start(&d->cabac, data + pos, len - pos);
      memcpy(d->cabac.ctx, keep, sizeof(keep));
    }
  }
  (void)pic;
  return 0;
}

/* ------------------------------------------------------------------ NAL entry */
int orc_dec_decode_nal(orc_decoder *d, const uint8_t *data, size_t len, int64_t pts)
{
  /* strip Annex-B start code if present */
  size_t i = 0;
  while (i + 2 < len && data[i] == 0) i++;
  if (i >= 2 && i < len && data[i] == 1) { data += i + 1; len -= i + 1; }
  if (len < 2) return ERR_INVALID;
  if (data[0] & 0x80) return ERR_INVALID;
  int nal_type = (data[0] >> 1) & 0x3f;
  int layer = ((data[0] & 1) << 5) | (data[1] >> 3);
  if (layer != 0) return 0;
  d->tid = (data[1] & 7) - 1;
  if (len > d->rbsp_cap) { d->rbsp = (uint8_t *)realloc(d->rbsp, len); d->rbsp_cap = len; }
  size_t rlen = orc_unescape(data + 2, len - 2, d->rbsp, NULL, 0, NULL);
  orc_bitr br; orc_br_init(&br, d->rbsp, rlen);
  d->cur_pts = pts;
  if (nal_type == NAL_VPS) { orc_vps v; int r = orc_parse_vps(&br, &v); if (r) return r; d->vps[v.vps_id] = v; return 0; }
  if (nal_type == NAL_SPS) { orc_sps s; int r = orc_parse_sps(&br, &s); if (r) return r; d->sps[s.sps_id] = s; return 0; }
  if (nal_type == NAL_PPS) { orc_pps p; int r = orc_parse_pps(&br, &p); if (r) return r; d->pps[p.pps_id] = p; return 0; }
  if (nal_type == 40 && d->last_finished) {           /* suffix SEI: a decoded picture hash (D.2.19) of the picture just finished is checked */
    while (br.pos + 16 <= rlen * 8 && !br.error) {
      int type = 0, size = 0, b;
      do { b = (int)orc_br_get(&br, 8); type += b; } while (b == 255 && !br.error);
      do { b = (int)orc_br_get(&br, 8); size += b; } while (b == 255 && !br.error);
      if (br.error || br.pos + (size_t)size * 8 > rlen * 8) break;
      if (type == 132 && size >= 1) {
        const int ht = (int)orc_br_get(&br, 8), nb = ht <= 2 ? orc_hash_bytes(ht) : 0;
        if (nb && size == 1 + 3 * nb) {
          uint8_t want[3][16], got[3][16]; memset(want, 0, sizeof(want));
          for (int c = 0; c < 3; c++) for (int i = 0; i < nb; i++) want[c][i] = (uint8_t)orc_br_get(&br, 8);
          const orc_pic *pic = d->last_finished;
          const pixel *pl[3] = { pic->plane[0], pic->plane[1], pic->plane[2] };
          orc_picture_hash(ht, pl, pic->stride, pic->w, pic->h, got);
          d->hash_checked++;
          if (memcmp(want, got, sizeof(want))) d->hash_mismatch++;
        } else br.pos += (size_t)(size - 1) * 8;
      } else br.pos += (size_t)size * 8;
    }
    return 0;
  }
  if (nal_type == NAL_EOS || nal_type == NAL_EOB) {
    /* the coded video sequence ends: what still waits goes out now (a decoder at the end of a call would not keep pictures back for a sequence that may never
     * come), and the picture that follows starts a sequence whatever its type */
    if (d->pic_active) finish_picture(d);
    while (waiting_for_output(d)) bump(d);
    d->after_eos = 1;
    return d->out_n > 0 ? 1 : 0;
  }
  if (nal_type > 31) return 0;                        /* AUD, other SEI, ... ignored */
  if ((nal_type > NAL_TRAIL_R + 8 && nal_type < NAL_BLA_W_LP) || nal_type > NAL_CRA) return 0;   /* reserved */
  free(d->sh.entry_point_offset); d->sh.entry_point_offset = NULL;
  d->nal_type = nal_type;
  int r = orc_parse_slice_header(&br, &d->sh, nal_type, d->sps, d->pps);
  if (r) return r;
  d->p = &d->pps[d->sh.pps_id]; d->s = &d->sps[d->p->sps_id];
  if (d->sh.first_slice_segment_in_pic) {
    if (d->pic_active) finish_picture(d);             /* previous picture was incomplete */
    r = start_picture(d);
    if (r <= 0) return r;
  } else if (!d->pic_active) return 0;
  d->scaling_on = d->s->scaling_list_enabled;
  if (d->scaling_on) {                                 /* 7.4.5: the PPS's lists when it carries any, else the SPS's (the default ones without data) */
    const orc_scaling_lists *sl = d->p->scaling_list_data_present ? &d->p->scaling : &d->s->scaling;
    for (int sz = 0; sz < 4; sz++) for (int m = 0; m < (sz == 3 ? 2 : 6); m++) orc_scaling_factor(sl, sz, m, d->sfac[sz][m]);
  }
  r = build_ref_list(d);
  if (r) { d->pic_active = 0; d->cur->is_ref = 0; d->cur->needed_for_output = 0; return r; }
  d->err = 0;
  d->last_slice_type = d->sh.slice_type; d->cur->slice_type = d->sh.slice_type;
  size_t hdr_bytes = br.pos >> 3;
  r = decode_slice_data(d, d->rbsp + hdr_bytes, rlen - hdr_bytes);
  if (r) { d->pic_active = 0; d->cur->is_ref = 0; d->cur->needed_for_output = 0; return r; }
  if (d->ctbs_decoded >= d->s->pic_w_ctbs * d->s->pic_h_ctbs) { finish_picture(d); return d->out_n > 0 ? 1 : 0; }
  return 0;
}

/* test aid: the side information of the picture finished last, per 4x4 luma block (raster, b4_w per row) -- returns the number of blocks */
int orc_dec_debug_side(orc_decoder *d, int16_t *mv, int8_t *ref, uint8_t *pm, uint8_t *im, int8_t *qp)
{
  const orc_pic *p = d->last_finished;
  if (!p) return 0;
  const int n = p->b4_w * p->b4_h;
  for (int i = 0; i < n; i++) {
    mv[2 * i] = p->mvf[i].mv[0]; mv[2 * i + 1] = p->mvf[i].mv[1]; ref[i] = p->mvf[i].ref_idx;
    pm[i] = p->pred_mode[i]; im[i] = p->intra_mode[i]; qp[i] = p->qp_y[i];
  }
  return n;
}
int orc_dec_concealed(const orc_decoder *d) { return d->concealed; }
void orc_dec_hash_stats(orc_decoder *d, int *checked, int *mismatch) { if (checked) *checked = d->hash_checked; if (mismatch) *mismatch = d->hash_mismatch; }

int orc_dec_get_frame(orc_decoder *d, orc_dec_frame *out)
{
  if (d->out_n == 0) return 0;
  orc_pic *pic = d->out_queue[0];
  for (int i = 1; i < d->out_n; i++) d->out_queue[i - 1] = d->out_queue[i];
  d->out_n--;
  pic->needed_for_output = 0; pic->out_queued = 0;
  d->last_output = pic;
  const orc_sps *s = d->s;
  int cl = s->conf_win_flag ? s->conf_left * 2 : 0, cr = s->conf_win_flag ? s->conf_right * 2 : 0;
  int ct = s->conf_win_flag ? s->conf_top * 2 : 0, cb = s->conf_win_flag ? s->conf_bottom * 2 : 0;
  out->coded_width = pic->w; out->coded_height = pic->h;
  out->width = pic->w - cl - cr; out->height = pic->h - ct - cb;
  out->plane[0] = pic->plane[0] + ct * pic->stride[0] + cl;
  out->plane[1] = pic->plane[1] + (ct / 2) * pic->stride[1] + cl / 2;
  out->plane[2] = pic->plane[2] + (ct / 2) * pic->stride[2] + cl / 2;
  for (int i = 0; i < 3; i++) out->stride[i] = pic->stride[i];
  out->poc = pic->poc; out->pts = pic->pts;
  out->slice_type = pic->slice_type;
  const orc_vps *v = &d->vps[s->vps_id];
  out->fps_num = out->fps_den = 0;
  if (s->vui_timing_present) { out->fps_num = s->vui_time_scale; out->fps_den = s->vui_num_units_in_tick; }
  else if (v->valid && v->timing_info_present) { out->fps_num = v->time_scale; out->fps_den = v->num_units_in_tick; }
  return 1;
}
