/* oracle/hevc_pic.c -- see hevc_pic.h.  Test infrastructure. */
#include "hevc_pic.h"

int orc_pic_alloc(orc_pic *p, int w, int h)
{
  memset(p, 0, sizeof(*p));
  p->w = w; p->h = h;
  p->stride[0] = w; p->stride[1] = p->stride[2] = w / 2;
  p->plane[0] = (pixel *)calloc((size_t)w * h, 1);
  p->plane[1] = (pixel *)calloc((size_t)(w / 2) * (h / 2), 1);
  p->plane[2] = (pixel *)calloc((size_t)(w / 2) * (h / 2), 1);
  p->b4_w = w / 4; p->b4_h = h / 4;
  size_t n = (size_t)p->b4_w * p->b4_h;
  p->pred_mode = (uint8_t *)malloc(n); p->ct_depth = (uint8_t *)malloc(n); p->intra_mode = (uint8_t *)malloc(n);
  p->qp_y = (int8_t *)malloc(n); p->tu_nz = (uint8_t *)malloc(n); p->edge_v = (uint8_t *)malloc(n);
  p->edge_h = (uint8_t *)malloc(n); p->no_filter = (uint8_t *)malloc(n);
  p->mvf = (orc_mvinfo *)malloc(n * sizeof(orc_mvinfo));
  if (!p->plane[0] || !p->plane[1] || !p->plane[2] || !p->mvf) return -1;
  orc_pic_reset_side(p);
  return 0;
}
void orc_pic_free(orc_pic *p)
{
  for (int i = 0; i < 3; i++) free(p->plane[i]);
  free(p->pred_mode); free(p->ct_depth); free(p->intra_mode); free(p->qp_y); free(p->tu_nz);
  free(p->edge_v); free(p->edge_h); free(p->no_filter); free(p->mvf);
  memset(p, 0, sizeof(*p));
}
void orc_pic_reset_side(orc_pic *p)
{
  size_t n = (size_t)p->b4_w * p->b4_h;
  memset(p->pred_mode, 255, n); memset(p->ct_depth, 0, n); memset(p->intra_mode, 1, n);
  memset(p->qp_y, 0, n); memset(p->tu_nz, 0, n); memset(p->edge_v, 0, n); memset(p->edge_h, 0, n);
  memset(p->no_filter, 0, n);
  for (size_t i = 0; i < n; i++) { p->mvf[i].mv[0] = p->mvf[i].mv[1] = 0; p->mvf[i].ref_idx = -1; p->mvf[i].ref_idx1 = -1; p->mvf[i].mv1[0] = p->mvf[i].mv1[1] = 0; }
}

static int bs_pair(const orc_pic *p, int ip, int iq, int edge_bits)
{
  if (p->pred_mode[ip] == MODE_INTRA || p->pred_mode[iq] == MODE_INTRA) return 2;
  if ((edge_bits & 1) && (p->tu_nz[ip] || p->tu_nz[iq])) return 1;
  const orc_mvinfo *a = &p->mvf[ip], *b = &p->mvf[iq];
  /* 8.7.2.4: the motion of each side as a set of (reference PICTURE, vector) pairs -- which list an index belongs to does not matter, and two indices
   * may name one picture */
  int na = 0, nb = 0, pa[2], pb[2]; const int16_t *va[2], *vb[2];
  if (a->ref_idx >= 0) { pa[na] = p->ref_poc_list[a->ref_idx & 15]; va[na++] = a->mv; }
  if (a->ref_idx1 >= 0) { pa[na] = p->ref_poc_list1[a->ref_idx1 & 15]; va[na++] = a->mv1; }
  if (b->ref_idx >= 0) { pb[nb] = p->ref_poc_list[b->ref_idx & 15]; vb[nb++] = b->mv; }
  if (b->ref_idx1 >= 0) { pb[nb] = p->ref_poc_list1[b->ref_idx1 & 15]; vb[nb++] = b->mv1; }
#define FAR(u, v) (orc_abs((u)[0] - (v)[0]) >= 4 || orc_abs((u)[1] - (v)[1]) >= 4)
  if (na != nb) return 1;                                                  /* different number of motion vectors */
  if (na == 1) return (pa[0] != pb[0] || FAR(va[0], vb[0])) ? 1 : 0;
  if (!((pa[0] == pb[0] && pa[1] == pb[1]) || (pa[0] == pb[1] && pa[1] == pb[0]))) return 1;      /* different reference pictures */
  if (pa[0] != pa[1]) {                                                    /* two pictures: the vectors that point into the same picture are compared */
    if (pa[0] == pb[0]) return (FAR(va[0], vb[0]) || FAR(va[1], vb[1])) ? 1 : 0;
    return (FAR(va[0], vb[1]) || FAR(va[1], vb[0])) ? 1 : 0;
  }
  /* both vectors of both sides point into one picture: either pairing may match */
  return ((FAR(va[0], vb[0]) || FAR(va[1], vb[1])) && (FAR(va[0], vb[1]) || FAR(va[1], vb[0]))) ? 1 : 0;
#undef FAR
}

void orc_compute_bs(const orc_pic *p, uint8_t *bs_v, uint8_t *bs_h)
{
  int w8 = p->w / 8, w4 = p->b4_w, h4 = p->b4_h, h8 = p->h / 8;
  memset(bs_v, 0, (size_t)w8 * h4);
  memset(bs_h, 0, (size_t)w4 * h8);
  for (int y4 = 0; y4 < h4; y4++)
    for (int x8 = 1; x8 < w8; x8++) {
      int iq = y4 * w4 + x8 * 2, ip = iq - 1;
      if (p->edge_v[iq]) bs_v[y4 * w8 + x8] = (uint8_t)bs_pair(p, ip, iq, p->edge_v[iq]);
    }
  for (int y8 = 1; y8 < h8; y8++)
    for (int x4 = 0; x4 < w4; x4++) {
      int iq = (y8 * 2) * w4 + x4, ip = iq - w4;
      if (p->edge_h[iq]) bs_h[y8 * w4 + x4] = (uint8_t)bs_pair(p, ip, iq, p->edge_h[iq]);
    }
}
