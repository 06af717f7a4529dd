/* oracle/hevc_sao.c -- see hevc_sao.h.  Test infrastructure. */
#include "hevc_sao.h"

/* Table 8-13: hPos / vPos of the two neighbours for SaoEoClass 0..3 */
const int8_t orc_sao_eo_dx[4][2] = { { -1, 1 }, { 0, 0 }, { -1, 1 }, { 1, -1 } };
const int8_t orc_sao_eo_dy[4][2] = { { 0, 0 }, { -1, 1 }, { -1, 1 }, { -1, 1 } };

/* ------------------------------------------------------------------ 8.7.3 picture process */
static int neighbour_usable(const orc_sao_ctx *s, int ctb, int nctb)
{
  if (ctb == nctb) return 1;
  if (s->ctb_nb) {
    const int w = s->pic_w_ctbs, dx = nctb % w - ctb % w, dy = nctb / w - ctb / w;
    return (s->ctb_nb[ctb] >> orc_lf_neighbour_bit(dx, dy)) & 1;
  }
  if (s->ctb_slice && s->ctb_slice[ctb] != s->ctb_slice[nctb] && !s->across_slices) return 0;
  if (s->ctb_tile && s->ctb_tile[ctb] != s->ctb_tile[nctb] && !s->across_tiles) return 0;
  return 1;
}

static void sao_ctb(const orc_sao_ctx *s, int rx, int ry, int c)
{
  const int sh = c ? 1 : 0, pw = s->w >> sh, ph = s->h >> sh, n = (1 << s->ctb_log2) >> sh;
  const int ctb = ry * s->pic_w_ctbs + rx;
  const orc_sao_params *p = &s->params[ctb];
  const int x0 = rx * n, y0 = ry * n, st = s->stride[c];
  int band_table[32]; memset(band_table, 0, sizeof(band_table));
  if (p->type[c] == 1) for (int k = 0; k < 4; k++) band_table[(k + p->band_pos[c]) & 31] = k + 1;
  for (int y = y0; y < y0 + n && y < ph; y++) for (int x = x0; x < x0 + n && x < pw; x++) {
    const int v = s->src[c][y * st + x];
    int out = v;
    const int skip = s->no_filter && s->no_filter[((y << sh) >> 2) * s->nf_stride + ((x << sh) >> 2)];
    if (!skip && p->type[c] == 1) {
      const int b = band_table[v >> 3];                         /* bandShift = bitDepth - 5 */
      if (b) out = orc_clip_pixel(v + p->offset[c][b - 1]);
    } else if (!skip && p->type[c] == 2) {
      const int e = p->eo_class[c];
      int nb[2], ok = 1;
      for (int k = 0; k < 2; k++) {
        const int xn = x + orc_sao_eo_dx[e][k], yn = y + orc_sao_eo_dy[e][k];
        if (xn < 0 || yn < 0 || xn >= pw || yn >= ph) { ok = 0; break; }
        if (!neighbour_usable(s, ctb, (yn / n) * s->pic_w_ctbs + xn / n)) { ok = 0; break; }
        nb[k] = s->src[c][yn * st + xn];
      }
      if (ok) {
        const int idx = orc_sao_edge_idx(v, nb[0], nb[1]);
        if (idx) out = orc_clip_pixel(v + p->offset[c][idx - 1]);
      }
    }
    s->dst[c][y * st + x] = (pixel)out;
  }
}

void orc_sao_picture(const orc_sao_ctx *s)
{
  const int n = 1 << s->ctb_log2, hc = (s->h + n - 1) / n;
  for (int ry = 0; ry < hc; ry++) for (int rx = 0; rx < s->pic_w_ctbs; rx++) for (int c = 0; c < 3; c++) sao_ctb(s, rx, ry, c);
}

/* ------------------------------------------------------------------ 7.3.8.3 syntax */
static int same_params(const orc_sao_params *a, const orc_sao_params *b) { return memcmp(a, b, sizeof(*a)) == 0; }

void orc_sao_write(orc_cabac_enc *c, const orc_sao_params *p, const orc_sao_params *left, const orc_sao_params *up, int luma, int chroma)
{
  if (left) { const int m = same_params(p, left); orc_cenc_bin(c, CTX_SAO_MERGE, m); if (m) return; }
  if (up) { const int m = same_params(p, up); orc_cenc_bin(c, CTX_SAO_MERGE, m); if (m) return; }
  for (int ci = 0; ci < 3; ci++) {
    if (!(ci ? chroma : luma)) continue;
    if (ci < 2) {                                              /* sao_type_idx_luma / _chroma: TR cMax 2, first bin coded */
      orc_cenc_bin(c, CTX_SAO_TYPE, p->type[ci] != 0);
      if (p->type[ci]) orc_cenc_bypass(c, p->type[ci] == 2);
    }
    if (!p->type[ci]) continue;
    for (int i = 0; i < 4; i++) {                               /* sao_offset_abs: TR cMax 7, bypass */
      const int a = orc_abs(p->offset[ci][i]);
      for (int k = 0; k < a; k++) orc_cenc_bypass(c, 1);
      if (a < 7) orc_cenc_bypass(c, 0);
    }
    if (p->type[ci] == 1) {
      for (int i = 0; i < 4; i++) if (p->offset[ci][i]) orc_cenc_bypass(c, p->offset[ci][i] < 0);
      orc_cenc_bypass_bits(c, p->band_pos[ci], 5);
    } else if (ci < 2) orc_cenc_bypass_bits(c, p->eo_class[ci], 2);
  }
}

void orc_sao_parse(orc_cabac_dec *c, orc_sao_params *p, const orc_sao_params *left, const orc_sao_params *up, int luma, int chroma)
{
  memset(p, 0, sizeof(*p));
  if (left && orc_cdec_bin(c, CTX_SAO_MERGE)) { *p = *left; return; }
  if (up && orc_cdec_bin(c, CTX_SAO_MERGE)) { *p = *up; return; }
  for (int ci = 0; ci < 3; ci++) {
    if (!(ci ? chroma : luma)) continue;
    if (ci < 2) {
      int t = 0;
      if (orc_cdec_bin(c, CTX_SAO_TYPE)) t = orc_cdec_bypass(c) ? 2 : 1;
      p->type[ci] = (uint8_t)t;
    } else { p->type[2] = p->type[1]; p->eo_class[2] = p->eo_class[1]; }
    if (!p->type[ci]) continue;
    int a[4];
    for (int i = 0; i < 4; i++) { a[i] = 0; while (a[i] < 7 && orc_cdec_bypass(c)) a[i]++; }
    if (p->type[ci] == 1) {
      for (int i = 0; i < 4; i++) if (a[i] && orc_cdec_bypass(c)) a[i] = -a[i];
      p->band_pos[ci] = (uint8_t)orc_cdec_bypass_bits(c, 5);
    } else {
      if (ci < 2) p->eo_class[ci] = (uint8_t)orc_cdec_bypass_bits(c, 2);
      a[2] = -a[2]; a[3] = -a[3];                               /* edge offsets: categories 1, 2 positive, 3, 4 negative */
    }
    for (int i = 0; i < 4; i++) p->offset[ci][i] = (int8_t)a[i];
  }
}

/* ------------------------------------------------------------------ encoder decision
 * "uvgx SAO decision v1".  Per CTU and colour component, over the samples of the CTB, with d = source - deblocked:
 *   edge class e, category k = edgeIdx 1..4 (samples with a neighbour outside the picture have none): count N, sum S of d;
 *   band b = deblocked >> 3: count, sum.
 * offset o = round(S / N) (half away from zero; 0 when N = 0), clipped to [0, 7] for categories 1, 2, [-7, 0] for 3, 4,
 * [-7, 7] for bands.  Estimated change of the squared error: N o^2 - 2 o S.  Band start: the s in 0..28 whose four bands
 * s..s+3 give the smallest sum (smallest s on ties).  bins(o) = |o| + 1 (7 when |o| = 7).
 * cost = 256 * error change + lambda_q4^2 * bins, with bins = type (1 for off, else 2) + offsets (+ signs of non-zero band
 * offsets + 5 for a band position; + 2 for an edge class).  Luma takes the cheapest of [off, EO 0..3, BO] (first on ties).
 * Cb and Cr share type and class: the cheapest sum of both, type / class bins counted once, offsets and band positions
 * each their own.  Merge flags are pure syntax: left if the left CTU has identical parameters, else up likewise. */
static int rdiv(int64_t a, int64_t b)
{
  if (b == 0) return 0;
  return a >= 0 ? (int)((2 * a + b) / (2 * b)) : -(int)((-2 * a + b) / (2 * b));
}
static int off_bins(int o) { const int a = orc_abs(o); return a < 7 ? a + 1 : 7; }

typedef struct { int64_t dist; int bins; int8_t off[4]; uint8_t band; } sao_cand;

static void component_candidates(const pixel *deb, const pixel *org, int st, int pw, int ph, int x0, int y0, int n, sao_cand cand[5])
{
  int64_t en[4][5], es[4][5], bn[32], bs[32];
  memset(en, 0, sizeof(en)); memset(es, 0, sizeof(es)); memset(bn, 0, sizeof(bn)); memset(bs, 0, sizeof(bs));
  for (int y = y0; y < y0 + n && y < ph; y++) for (int x = x0; x < x0 + n && x < pw; x++) {
    const int v = deb[y * st + x], d = org[y * st + x] - v;
    bn[v >> 3]++; bs[v >> 3] += d;
    for (int e = 0; e < 4; e++) {
      const int xa = x + orc_sao_eo_dx[e][0], ya = y + orc_sao_eo_dy[e][0], xb = x + orc_sao_eo_dx[e][1], yb = y + orc_sao_eo_dy[e][1];
      if (xa < 0 || ya < 0 || xa >= pw || ya >= ph || xb < 0 || yb < 0 || xb >= pw || yb >= ph) continue;
      const int k = orc_sao_edge_idx(v, deb[ya * st + xa], deb[yb * st + xb]);
      en[e][k]++; es[e][k] += d;
    }
  }
  for (int e = 0; e < 4; e++) {
    sao_cand *c = &cand[e]; c->dist = 0; c->bins = 0; c->band = 0;
    for (int k = 1; k <= 4; k++) {
      int o = rdiv(es[e][k], en[e][k]);
      o = k <= 2 ? orc_clip3(0, 7, o) : orc_clip3(-7, 0, o);
      c->off[k - 1] = (int8_t)o; c->dist += en[e][k] * o * o - 2 * o * es[e][k]; c->bins += off_bins(o);
    }
  }
  int bo[32]; int64_t bg[32];
  for (int b = 0; b < 32; b++) { bo[b] = orc_clip3(-7, 7, rdiv(bs[b], bn[b])); bg[b] = bn[b] * bo[b] * bo[b] - 2 * bo[b] * bs[b]; }
  int best = 0; int64_t bestg = 0;
  for (int s = 0; s <= 28; s++) { const int64_t g = bg[s] + bg[s + 1] + bg[s + 2] + bg[s + 3]; if (s == 0 || g < bestg) { best = s; bestg = g; } }
  sao_cand *c = &cand[4]; c->dist = bestg; c->bins = 5; c->band = (uint8_t)best;
  for (int k = 0; k < 4; k++) { c->off[k] = (int8_t)bo[best + k]; c->bins += off_bins(bo[best + k]) + (bo[best + k] != 0); }
}

void orc_sao_decide_ctu(const pixel *const deb[3], const pixel *const org[3], const int stride[3], int w, int h,
                        int cx, int cy, int lambda_q4, orc_sao_params *out)
{
  const int64_t l2 = (int64_t)lambda_q4 * lambda_q4;
  sao_cand cand[3][5];
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, n = 64 >> sh;
    component_candidates(deb[c], org[c], stride[c], w >> sh, h >> sh, cx * n, cy * n, n, cand[c]);
  }
  memset(out, 0, sizeof(*out));
  /* candidate index: 0 = off, 1..4 = edge class 0..3, 5 = band */
  int64_t best = l2 * 1; int pick = 0;
  for (int t = 1; t <= 5; t++) {
    const sao_cand *c = &cand[0][t - 1];
    const int64_t cost = 256 * c->dist + l2 * (2 + c->bins + (t <= 4 ? 2 : 0));
    if (cost < best) { best = cost; pick = t; }
  }
  int pickc = 0; best = l2 * 1;
  for (int t = 1; t <= 5; t++) {
    const sao_cand *a = &cand[1][t - 1], *b = &cand[2][t - 1];
    const int64_t cost = 256 * (a->dist + b->dist) + l2 * (2 + a->bins + b->bins + (t <= 4 ? 2 : 0));
    if (cost < best) { best = cost; pickc = t; }
  }
  for (int c = 0; c < 3; c++) {
    const int t = c ? pickc : pick;
    if (!t) continue;
    const sao_cand *k = &cand[c][t - 1];
    out->type[c] = t == 5 ? 1 : 2;
    if (t == 5) out->band_pos[c] = k->band; else out->eo_class[c] = (uint8_t)(t - 1);
    for (int i = 0; i < 4; i++) out->offset[c][i] = k->off[i];
  }
}
