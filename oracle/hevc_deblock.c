/* oracle/hevc_deblock.c -- H.265 8.7.2 restated.  Test infrastructure. */
#include "hevc_deblock.h"

/* 8.7.2.5.3 decisions + 8.7.2.5.7 luma sample filtering for one 4-line segment.
 * p points at q0 sample of line 0; `xs` steps across the edge, `ls` steps along it. */
static void luma_segment(pixel *q0p, int xs, int ls, int bs, int qp, int beta_off, int tc_off,
                         int nf_p, int nf_q)
{
  int qb = orc_clip3(0, 51, qp + (beta_off << 1));
  int beta = orc_beta_table[qb];
  int qt = orc_clip3(0, 53, qp + 2 * (bs - 1) + (tc_off << 1));
  int tc = orc_tc_table[qt];
#define P(i,l) q0p[-((i) + 1) * xs + (l) * ls]
#define Q(i,l) q0p[(i) * xs + (l) * ls]
  int dp0 = orc_abs(P(2,0) - 2 * P(1,0) + P(0,0)), dp3 = orc_abs(P(2,3) - 2 * P(1,3) + P(0,3));
  int dq0 = orc_abs(Q(2,0) - 2 * Q(1,0) + Q(0,0)), dq3 = orc_abs(Q(2,3) - 2 * Q(1,3) + Q(0,3));
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
  int d = dpq0 + dpq3;
  if (d >= beta) return;
  int ds0 = (2 * dpq0 < (beta >> 2)) && (orc_abs(P(3,0) - P(0,0)) + orc_abs(Q(0,0) - Q(3,0)) < (beta >> 3)) &&
            (orc_abs(P(0,0) - Q(0,0)) < ((5 * tc + 1) >> 1));
  int ds3 = (2 * dpq3 < (beta >> 2)) && (orc_abs(P(3,3) - P(0,3)) + orc_abs(Q(0,3) - Q(3,3)) < (beta >> 3)) &&
            (orc_abs(P(0,3) - Q(0,3)) < ((5 * tc + 1) >> 1));
  int strong = ds0 && ds3;
  int dep = dp < ((beta + (beta >> 1)) >> 3);
  int deq = dq < ((beta + (beta >> 1)) >> 3);
  for (int l = 0; l < 4; l++) {
    int p0 = P(0,l), p1 = P(1,l), p2 = P(2,l), p3 = P(3,l);
    int q0 = Q(0,l), q1 = Q(1,l), q2 = Q(2,l), q3 = Q(3,l);
    if (strong) {
      if (!nf_p) {
        P(0,l) = (pixel)orc_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        P(1,l) = (pixel)orc_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
        P(2,l) = (pixel)orc_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (!nf_q) {
        Q(0,l) = (pixel)orc_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        Q(1,l) = (pixel)orc_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
        Q(2,l) = (pixel)orc_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (orc_abs(delta) < tc * 10) {
        delta = orc_clip3(-tc, tc, delta);
        if (!nf_p) P(0,l) = (pixel)orc_clip_pixel(p0 + delta);
        if (!nf_q) Q(0,l) = (pixel)orc_clip_pixel(q0 - delta);
        if (dep && !nf_p) {
          int dlt = orc_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1);
          P(1,l) = (pixel)orc_clip_pixel(p1 + dlt);
        }
        if (deq && !nf_q) {
          int dlt = orc_clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1);
          Q(1,l) = (pixel)orc_clip_pixel(q1 + dlt);
        }
      }
    }
  }
#undef P
#undef Q
}

/* 8.7.2.5.5 + 8.7.2.5.8 chroma: only bS == 2; 4 chroma lines <-> 8 luma lines?  One call
 * handles the chroma lines covered by one 4-luma-row bS segment (= 2 chroma lines). */
static void chroma_segment(pixel *q0p, int xs, int ls, int nlines, int qp_avg, int c_off, int tc_off,
                           int nf_p, int nf_q)
{
  int qpc = orc_chroma_qp_table[orc_clip3(0, 57, qp_avg + c_off)];
  int qt = orc_clip3(0, 53, qpc + 2 + (tc_off << 1));
  int tc = orc_tc_table[qt];
  for (int l = 0; l < nlines; l++) {
    pixel *q = q0p + l * ls;
    int p0 = q[-xs], p1 = q[-2 * xs], q0 = q[0], q1 = q[xs];
    int delta = orc_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!nf_p) q[-xs] = (pixel)orc_clip_pixel(p0 + delta);
    if (!nf_q) q[0] = (pixel)orc_clip_pixel(q0 - delta);
  }
}

static inline int nf_at(const orc_deblock_ctx *d, int x, int y)
{
  return d->no_filter ? d->no_filter[(y >> 2) * d->nf_stride + (x >> 2)] : 0;
}
static inline int qp_at(const orc_deblock_ctx *d, int x, int y)
{
  return d->qp_y[(y >> 2) * d->qp_stride + (x >> 2)];
}

void orc_deblock_vertical_edges(const orc_deblock_ctx *d)
{
  for (int y = 0; y < d->h; y += 4)
    for (int x = 8; x < d->w; x += 8) {
      int bs = d->bs_v[(y >> 2) * d->bs_stride_v + (x >> 3)];
      if (!bs) continue;
      int qp = (qp_at(d, x, y) + qp_at(d, x - 1, y) + 1) >> 1;
      int nfp = nf_at(d, x - 1, y), nfq = nf_at(d, x, y);
      luma_segment(d->plane[0] + y * d->stride[0] + x, 1, d->stride[0], bs, qp,
                   d->beta_offset_div2, d->tc_offset_div2, nfp, nfq);
      if (bs == 2 && (x & 15) == 0) {       /* chroma edge spacing: 8 chroma samples */
        int cx = x >> 1, cy = y >> 1;
        chroma_segment(d->plane[1] + cy * d->stride[1] + cx, 1, d->stride[1], 2, qp, d->cb_qp_offset, d->tc_offset_div2, nfp, nfq);
        chroma_segment(d->plane[2] + cy * d->stride[2] + cx, 1, d->stride[2], 2, qp, d->cr_qp_offset, d->tc_offset_div2, nfp, nfq);
      }
    }
}

void orc_deblock_horizontal_edges(const orc_deblock_ctx *d)
{
  for (int y = 8; y < d->h; y += 8)
    for (int x = 0; x < d->w; x += 4) {
      int bs = d->bs_h[(y >> 3) * d->bs_stride_h + (x >> 2)];
      if (!bs) continue;
      int qp = (qp_at(d, x, y) + qp_at(d, x, y - 1) + 1) >> 1;
      int nfp = nf_at(d, x, y - 1), nfq = nf_at(d, x, y);
      luma_segment(d->plane[0] + y * d->stride[0] + x, d->stride[0], 1, bs, qp,
                   d->beta_offset_div2, d->tc_offset_div2, nfp, nfq);
      if (bs == 2 && (y & 15) == 0) {
        int cx = x >> 1, cy = y >> 1;
        chroma_segment(d->plane[1] + cy * d->stride[1] + cx, d->stride[1], 1, 2, qp, d->cb_qp_offset, d->tc_offset_div2, nfp, nfq);
        chroma_segment(d->plane[2] + cy * d->stride[2] + cx, d->stride[2], 1, 2, qp, d->cr_qp_offset, d->tc_offset_div2, nfp, nfq);
      }
    }
}

void orc_deblock_picture(const orc_deblock_ctx *d)
{
  orc_deblock_vertical_edges(d);
  orc_deblock_horizontal_edges(d);
}
