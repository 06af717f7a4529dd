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
static int mi_ref(const orc_mvinfo *m, int l) { return l ? m->ref_idx1 : m->ref_idx; }
static const int16_t *mi_mv(const orc_mvinfo *m, int l) { return l ? m->mv1 : m->mv; }
static int list_poc(const orc_mvpred_ctx *c, int l, int idx) { return l ? c->ref_poc1[idx & 15] : c->ref_poc[idx & 15]; }
static int list_lt(const orc_mvpred_ctx *c, int l, int idx) { return l ? c->ref_lt1[idx & 15] : c->ref_lt[idx & 15]; }
/* "the same motion vectors and the same reference indices" (8.5.3.2.3): per list, the vector of a list that is not used does not count */
static int same_motion(const orc_mvinfo *a, const orc_mvinfo *b)
{
  if (a->ref_idx != b->ref_idx || a->ref_idx1 != b->ref_idx1) return 0;
  if (a->ref_idx >= 0 && (a->mv[0] != b->mv[0] || a->mv[1] != b->mv[1])) return 0;
  if (a->ref_idx1 >= 0 && (a->mv1[0] != b->mv1[0] || a->mv1[1] != b->mv1[1])) return 0;
  return 1;
}
static void scale_by(int16_t mv[2], int td_, int tb_)
{
  const int td = orc_clip3(-128, 127, td_), tb = orc_clip3(-128, 127, tb_);
  if (td == 0) return;
  const int tx = (16384 + (orc_abs(td) >> 1)) / td;
  const int dsf = orc_clip3(-4096, 4095, (tb * tx + 32) >> 6);
  for (int q = 0; q < 2; q++) {
    const int prod = dsf * mv[q], sg = prod < 0 ? -1 : 1;
    mv[q] = (int16_t)orc_clip3(-32768, 32767, sg * ((orc_abs(prod) + 127) >> 8));
  }
}

/* 8.5.3.2.8 / 8.5.3.2.9: temporal luma motion vector prediction for the prediction block (xpb, ypb, w x h), list X and target reference
 * index ref_idx.  No long-term pictures.  Returns availableFlagLXCol. */
static int temporal_mv(const orc_mvpred_ctx *c, int xpb, int ypb, int npbw, int npbh, int X, int ref_idx, int16_t mv[2])
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
    const orc_mvinfo *m = &col->mvf[i];
    if (col->pred_mode[i] == MODE_INTRA || col->pred_mode[i] == 255 || (m->ref_idx < 0 && m->ref_idx1 < 0)) continue;
    /* which of the collocated block's vectors: the only one; of two, the one of the list being derived when no reference picture of the current
     * slice follows it in output order, else the one of list collocated_from_l0_flag (i.e. the vector that crosses the current picture) */
    int l;
    if (m->ref_idx < 0) l = 1; else if (m->ref_idx1 < 0) l = 0; else l = c->no_backward_pred ? X : c->collocated_from_l0;
    const int col_ref_poc = l ? col->ref_poc_list1[m->ref_idx1 & 15] : col->ref_poc_list[m->ref_idx & 15];
    const int col_lt = l ? col->ref_lt_list1[m->ref_idx1 & 15] : col->ref_lt_list[m->ref_idx & 15], cur_lt = list_lt(c, X, ref_idx);
    if (col_lt != cur_lt) continue;                              /* 8.5.3.2.9: one of the two reference pictures long-term, the other not: no candidate from this block */
    const int col_poc_diff = col->poc - col_ref_poc, cur_poc_diff = c->cur_poc - list_poc(c, X, ref_idx);
    mv[0] = mi_mv(m, l)[0]; mv[1] = mi_mv(m, l)[1];
    if (!cur_lt && col_poc_diff != cur_poc_diff && col_poc_diff != 0) scale_by(mv, col_poc_diff, cur_poc_diff);      /* (long-term: taken as it is) */
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
#define ADD(m) do { if (n < maxc) { cand[n] = *(m); if (cand[n].ref_idx < 0) cand[n].mv[0] = cand[n].mv[1] = 0; if (cand[n].ref_idx1 < 0) cand[n].mv1[0] = cand[n].mv1[1] = 0; n++; } } while (0)
  if (avA1) ADD(A1);
  if (avB1) ADD(B1);
  if (avB0) ADD(B0);
  if (avA0) ADD(A0);
  if (avB2) ADD(B2);
#undef ADD
  if (n < maxc) {                                   /* temporal candidate: refIdxLXCol = 0 for both lists (8.5.3.2.2 steps 2-4) */
    int16_t t0[2] = {0, 0}, t1[2] = {0, 0};
    const int f0 = temporal_mv(c, xpb, ypb, npbw, npbh, 0, 0, t0), f1 = c->is_b ? temporal_mv(c, xpb, ypb, npbw, npbh, 1, 0, t1) : 0;
    if (f0 || f1) {
      cand[n].mv[0] = t0[0]; cand[n].mv[1] = t0[1]; cand[n].ref_idx = f0 ? 0 : -1;
      cand[n].mv1[0] = t1[0]; cand[n].mv1[1] = t1[1]; cand[n].ref_idx1 = f1 ? 0 : -1;
      n++;
    }
  }
  /* 8.5.3.2.4 combined bi-predictive candidates (B slices): the list-0 motion of one original candidate with the list-1 motion of another,
   * unless the two are the same prediction */
  if (c->is_b && n > 1 && n < maxc) {
    static const uint8_t l0c[12] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3 }, l1c[12] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2 };
    const int norig = n;
    for (int comb = 0; comb < norig * (norig - 1) && n < maxc; comb++) {
      const orc_mvcand *p0 = &cand[l0c[comb]], *p1 = &cand[l1c[comb]];
      if (p0->ref_idx < 0 || p1->ref_idx1 < 0) continue;
      if (c->ref_poc[p0->ref_idx & 15] == c->ref_poc1[p1->ref_idx1 & 15] && p0->mv[0] == p1->mv1[0] && p0->mv[1] == p1->mv1[1]) continue;
      cand[n].mv[0] = p0->mv[0]; cand[n].mv[1] = p0->mv[1]; cand[n].ref_idx = p0->ref_idx;
      cand[n].mv1[0] = p1->mv1[0]; cand[n].mv1[1] = p1->mv1[1]; cand[n].ref_idx1 = p1->ref_idx1;
      n++;
    }
  }
  /* 8.5.3.2.5 zero motion vector merging candidates */
  const int nrefs = c->is_b ? ORC_MIN(c->num_ref_idx, c->num_ref_idx1) : c->num_ref_idx;
  int zero_idx = 0;
  while (n < maxc) {
    const int r = (zero_idx < nrefs) ? zero_idx : 0;
    cand[n].mv[0] = cand[n].mv[1] = cand[n].mv1[0] = cand[n].mv1[1] = 0;
    cand[n].ref_idx = (int8_t)r; cand[n].ref_idx1 = (int8_t)(c->is_b ? r : -1);
    n++; zero_idx++;
  }
}

void orc_amvp_candidates_lx(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                            int npbw, int npbh, int part_idx, int X, int ref_idx, int16_t cand[2][2])
{
  int xa[2] = { xpb - 1, xpb - 1 }, ya[2] = { ypb + npbh, ypb + npbh - 1 };                 /* A0, A1 */
  int xb[3] = { xpb + npbw, xpb + npbw - 1, xpb - 1 }, yb[3] = { ypb - 1, ypb - 1, ypb - 1 }; /* B0, B1, B2 */
  int avA[2], avB[3];
  for (int k = 0; k < 2; k++) avA[k] = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa[k], ya[k]);
  for (int k = 0; k < 3; k++) avB[k] = pb_available(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb[k], yb[k]);
  int is_scaled = avA[0] || avA[1];
  int flagA = 0, flagB = 0; int16_t mvA[2] = {0, 0}, mvB[2] = {0, 0};
  const int target_poc = list_poc(c, X, ref_idx), Y = !X, target_lt = list_lt(c, X, ref_idx);
  /* (8.5.3.2.7 step 7: a neighbour's vector into ANOTHER picture counts when that picture and the target are both long-term -- then as it is -- or both short-term -- scaled) */
#define LT_MATCH(m, L) (mi_ref(m, L) >= 0 && list_lt(c, L, mi_ref(m, L)) == target_lt)
  /* a neighbour's vector that points into the target picture: list X first, then list Y (8.5.3.2.7 steps 3 / 5 of the A and B derivations) */
#define SAME_PIC(m, L) (mi_ref(m, L) >= 0 && list_poc(c, L, mi_ref(m, L)) == target_poc)
#define TAKE(dst, m, L) do { (dst)[0] = mi_mv(m, L)[0]; (dst)[1] = mi_mv(m, L)[1]; } while (0)
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) {
    const orc_mvinfo *m = mvat(c, xa[k], ya[k]);
    if (SAME_PIC(m, X)) { flagA = 1; TAKE(mvA, m, X); } else if (SAME_PIC(m, Y)) { flagA = 1; TAKE(mvA, m, Y); }
  }
  /* A: then any reference picture, scaled by the ratio of the POC distances */
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) {
    const orc_mvinfo *m = mvat(c, xa[k], ya[k]);
    const int L = LT_MATCH(m, X) ? X : (LT_MATCH(m, Y) ? Y : -1);
    if (L >= 0) { flagA = 1; TAKE(mvA, m, L); if (!target_lt) scale_by(mvA, c->cur_poc - list_poc(c, L, mi_ref(m, L)), c->cur_poc - target_poc); }
  }
  /* B: same reference picture */
  for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) {
    const orc_mvinfo *m = mvat(c, xb[k], yb[k]);
    if (SAME_PIC(m, X)) { flagB = 1; TAKE(mvB, m, X); } else if (SAME_PIC(m, Y)) { flagB = 1; TAKE(mvB, m, Y); }
  }
  if (!is_scaled && flagB) { flagA = 1; mvA[0] = mvB[0]; mvA[1] = mvB[1]; }
  if (!is_scaled) {
    flagB = 0;
    for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) {
      const orc_mvinfo *m = mvat(c, xb[k], yb[k]);
      const int L = LT_MATCH(m, X) ? X : (LT_MATCH(m, Y) ? Y : -1);
      if (L >= 0) {
        flagB = 1; TAKE(mvB, m, L);
        const int poc = list_poc(c, L, mi_ref(m, L));
        if (!target_lt && poc != target_poc) scale_by(mvB, c->cur_poc - poc, c->cur_poc - target_poc);
      }
    }
  }
#undef SAME_PIC
#undef TAKE
#undef LT_MATCH
  int n = 0;
  if (flagA) { cand[n][0] = mvA[0]; cand[n][1] = mvA[1]; n++; }
  if (flagB && !(flagA && mvA[0] == mvB[0] && mvA[1] == mvB[1])) { cand[n][0] = mvB[0]; cand[n][1] = mvB[1]; n++; }
  if (n < 2) {                                       /* temporal candidate unless A and B are both there and differ (8.5.3.2.6) */
    int16_t tmv[2];
    if (temporal_mv(c, xpb, ypb, npbw, npbh, X, ref_idx, tmv)) { cand[n][0] = tmv[0]; cand[n][1] = tmv[1]; n++; }
  }
  while (n < 2) { cand[n][0] = cand[n][1] = 0; n++; }
}
void orc_amvp_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                         int npbw, int npbh, int part_idx, int ref_idx, int16_t cand[2][2])
{
  orc_amvp_candidates_lx(c, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, 0, ref_idx, cand);
}
