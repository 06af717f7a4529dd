/* oracle/hevc_mvpred.c -- see hevc_mvpred.h.  Test infrastructure. */
#include "hevc_mvpred.h"

/* H.265 6.4.2 availability of a neighbouring prediction block */
static int pb_available(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                        int npbw, int npbh, int part_idx, int xn, int yn)
{
  int same_cb = (xcb <= xn && ycb <= yn && xcb + ncbs > xn && ycb + ncbs > yn);
  int avail;
  if (!same_cb) avail = orc_available(&c->av, xpb, ypb, xn, yn);
  else if ((npbw << 1) == ncbs && (npbh << 1) == ncbs && part_idx == 1 && (ycb + npbh <= yn) && (xcb + npbw > xn)) avail = 0;
  else avail = 1;
  if (avail && c->pic->pred_mode[(yn >> 2) * c->pic->b4_w + (xn >> 2)] == MODE_INTRA) avail = 0;
  return avail;
}
static const orc_mvinfo *mvat(const orc_mvpred_ctx *c, int x, int y) { return &c->pic->mvf[(y >> 2) * c->pic->b4_w + (x >> 2)]; }
static int same_motion(const orc_mvinfo *a, const orc_mvinfo *b)
{
  return a->ref_idx == b->ref_idx && a->mv[0] == b->mv[0] && a->mv[1] == b->mv[1];
}

/* 8.5.3.2.8 / 8.5.3.2.9: temporal luma motion vector prediction for the prediction block (xpb, ypb, w x h) and target reference
 * index ref_idx.  Collocated pictures of P streams carry list-0 motion only; no long-term pictures.  Returns availableFlagLXCol. */
static int temporal_mv(const orc_mvpred_ctx *c, int xpb, int ypb, int npbw, int npbh, int ref_idx, int16_t mv[2])
{
  const orc_pic *col = c->col;
  if (!col) return 0;
  const int ctb = c->av.ctb_log2;
  int cand[2][2] = { { xpb + npbw, ypb + npbh }, { xpb + (npbw >> 1), ypb + (npbh >> 1) } };
  for (int k = 0; k < 2; k++) {
    int x = cand[k][0], y = cand[k][1];
    if (k == 0 && !((ypb >> ctb) == (y >> ctb) && y < c->av.pic_h && x < c->av.pic_w)) continue;   /* bottom right: same CTB row, inside the picture */
    x = (x >> 4) << 4; y = (y >> 4) << 4;                        /* motion is stored at 16x16 granularity */
    const int i = (y >> 2) * col->b4_w + (x >> 2);
    if (col->pred_mode[i] == MODE_INTRA || col->pred_mode[i] == 255 || col->mvf[i].ref_idx < 0) continue;
    const int col_poc_diff = col->poc - col->ref_poc_list[col->mvf[i].ref_idx & 15];
    const int cur_poc_diff = c->cur_poc - c->ref_poc[ref_idx];
    mv[0] = col->mvf[i].mv[0]; mv[1] = col->mvf[i].mv[1];
    if (col_poc_diff != cur_poc_diff && col_poc_diff != 0) {
      const int td = orc_clip3(-128, 127, col_poc_diff), tb = orc_clip3(-128, 127, cur_poc_diff);
      const int tx = (16384 + (orc_abs(td) >> 1)) / td;
      const int dsf = orc_clip3(-4096, 4095, (tb * tx + 32) >> 6);
      for (int q = 0; q < 2; q++) {
        const int prod = dsf * mv[q], sg = prod < 0 ? -1 : 1;
        mv[q] = (int16_t)orc_clip3(-32768, 32767, sg * ((orc_abs(prod) + 127) >> 8));
      }
    }
    return 1;
  }
  return 0;
}

void orc_merge_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                          int npbw, int npbh, int part_idx, int part_mode, orc_mvcand *cand)
{
  int lvl = c->log2_par_mrg_level, n = 0;
  if (lvl > 2 && ncbs == 8) { xpb = xcb; ypb = ycb; npbw = npbh = ncbs; part_idx = 0; part_mode = PART_2Nx2N; }
#define PAR(xn, yn) (((xpb >> lvl) == ((xn) >> lvl)) && ((ypb >> lvl) == ((yn) >> lvl)))
  int xa1 = xpb - 1, ya1 = ypb + npbh - 1;
  int xb1 = xpb + npbw - 1, yb1 = ypb - 1;
  int xb0 = xpb + npbw, yb0 = ypb - 1;
  int xa0 = xpb - 1, ya0 = ypb + npbh;
  int xb2 = xpb - 1, yb2 = ypb - 1;
  /* nbX: availableX of 8.5.3.2.3 (neighbour usable); flX: availableFlagX (enters the list) */
  int part1 = (part_idx == 1);
  int nbA1 = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa1, ya1) && !PAR(xa1, ya1) &&
             !(part1 && (part_mode == PART_Nx2N || part_mode == PART_nLx2N || part_mode == PART_nRx2N));
  int nbB1 = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb1, yb1) && !PAR(xb1, yb1) &&
             !(part1 && (part_mode == PART_2NxN || part_mode == PART_2NxnU || part_mode == PART_2NxnD));
  int nbB0 = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb0, yb0) && !PAR(xb0, yb0);
  int nbA0 = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa0, ya0) && !PAR(xa0, ya0);
  int nbB2 = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb2, yb2) && !PAR(xb2, yb2);
  const orc_mvinfo *A1 = nbA1 ? mvat(c, xa1, ya1) : NULL, *B1 = nbB1 ? mvat(c, xb1, yb1) : NULL;
  const orc_mvinfo *B0 = nbB0 ? mvat(c, xb0, yb0) : NULL, *A0 = nbA0 ? mvat(c, xa0, ya0) : NULL;
  const orc_mvinfo *B2 = nbB2 ? mvat(c, xb2, yb2) : NULL;
  int avA1 = nbA1;
  int avB1 = nbB1 && !(nbA1 && same_motion(A1, B1));
  int avB0 = nbB0 && !(nbB1 && same_motion(B1, B0));
  int avA0 = nbA0 && !(nbA1 && same_motion(A1, A0));
  int avB2 = nbB2 && !(nbA1 && same_motion(A1, B2)) && !(nbB1 && same_motion(B1, B2)) &&
             (avA0 + avA1 + avB0 + avB1 != 4);
#undef PAR
  int maxc = c->max_num_merge_cand;
#define ADD(m) do { if (n < maxc) { cand[n].mv[0] = (m)->mv[0]; cand[n].mv[1] = (m)->mv[1]; cand[n].ref_idx = (m)->ref_idx; n++; } } while (0)
  if (avA1) ADD(A1);
  if (avB1) ADD(B1);
  if (avB0) ADD(B0);
  if (avA0) ADD(A0);
  if (avB2) ADD(B2);
#undef ADD
  { int16_t tmv[2];                               /* temporal candidate: refIdxL0Col = 0 (8.5.3.2.2 step 3-4) */
    if (n < maxc && temporal_mv(c, xpb, ypb, npbw, npbh, 0, tmv)) { cand[n].mv[0] = tmv[0]; cand[n].mv[1] = tmv[1]; cand[n].ref_idx = 0; n++; } }
  /* 8.5.3.2.5 zero motion vector merging candidates (P slices) */
  int zero_idx = 0;
  while (n < maxc) {
    cand[n].mv[0] = cand[n].mv[1] = 0;
    cand[n].ref_idx = (int8_t)((zero_idx < c->num_ref_idx) ? zero_idx : 0);
    n++; zero_idx++;
  }
}

static void scale_mv(const orc_mvpred_ctx *c, int16_t mv[2], int ref_a, int ref_target)
{
  int td = orc_clip3(-128, 127, c->cur_poc - c->ref_poc[ref_a]);
  int tb = orc_clip3(-128, 127, c->cur_poc - c->ref_poc[ref_target]);
  if (td == 0) return;
  int tx = (16384 + (orc_abs(td) >> 1)) / td;
  int dsf = orc_clip3(-4096, 4095, (tb * tx + 32) >> 6);
  for (int k = 0; k < 2; k++) {
    int prod = dsf * mv[k];
    int s = prod < 0 ? -1 : 1;
    mv[k] = (int16_t)orc_clip3(-32768, 32767, s * ((orc_abs(prod) + 127) >> 8));
  }
}

void orc_amvp_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                         int npbw, int npbh, int part_idx, int ref_idx, int16_t cand[2][2])
{
  int xa[2] = { xpb - 1, xpb - 1 }, ya[2] = { ypb + npbh, ypb + npbh - 1 };                 /* A0, A1 */
  int xb[3] = { xpb + npbw, xpb + npbw - 1, xpb - 1 }, yb[3] = { ypb - 1, ypb - 1, ypb - 1 }; /* B0, B1, B2 */
  int avA[2], avB[3];
  for (int k = 0; k < 2; k++) avA[k] = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa[k], ya[k]);
  for (int k = 0; k < 3; k++) avB[k] = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb[k], yb[k]);
  int is_scaled = avA[0] || avA[1];
  int flagA = 0, flagB = 0; int16_t mvA[2] = {0, 0}, mvB[2] = {0, 0};
  int target_poc = c->ref_poc[ref_idx];
  /* A: same reference picture first */
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) {
    const orc_mvinfo *m = mvat(c, xa[k], ya[k]);
    if (m->ref_idx >= 0 && c->ref_poc[m->ref_idx] == target_poc) { flagA = 1; mvA[0] = m->mv[0]; mvA[1] = m->mv[1]; }
  }
  /* A: then any reference picture, scaled */
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) {
    const orc_mvinfo *m = mvat(c, xa[k], ya[k]);
    if (m->ref_idx >= 0) { flagA = 1; mvA[0] = m->mv[0]; mvA[1] = m->mv[1]; scale_mv(c, mvA, m->ref_idx, ref_idx); }
  }
  /* B: same reference picture */
  for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) {
    const orc_mvinfo *m = mvat(c, xb[k], yb[k]);
    if (m->ref_idx >= 0 && c->ref_poc[m->ref_idx] == target_poc) { flagB = 1; mvB[0] = m->mv[0]; mvB[1] = m->mv[1]; }
  }
  if (!is_scaled && flagB) { flagA = 1; mvA[0] = mvB[0]; mvA[1] = mvB[1]; }
  if (!is_scaled) {
    flagB = 0;
    for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) {
      const orc_mvinfo *m = mvat(c, xb[k], yb[k]);
      if (m->ref_idx >= 0) {
        flagB = 1; mvB[0] = m->mv[0]; mvB[1] = m->mv[1];
        if (c->ref_poc[m->ref_idx] != target_poc) scale_mv(c, mvB, m->ref_idx, ref_idx);
      }
    }
  }
  int n = 0;
  if (flagA) { cand[n][0] = mvA[0]; cand[n][1] = mvA[1]; n++; }
  if (flagB && !(flagA && mvA[0] == mvB[0] && mvA[1] == mvB[1])) { cand[n][0] = mvB[0]; cand[n][1] = mvB[1]; n++; }
  if (n < 2 && !(flagA && flagB && n == 2)) {        /* temporal candidate unless A and B are both there and differ (8.5.3.2.6) */
    int16_t tmv[2];
    if (temporal_mv(c, xpb, ypb, npbw, npbh, ref_idx, tmv)) { cand[n][0] = tmv[0]; cand[n][1] = tmv[1]; n++; }
  }
  while (n < 2) { cand[n][0] = cand[n][1] = 0; n++; }
}
