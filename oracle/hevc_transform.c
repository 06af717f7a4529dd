/* oracle/hevc_transform.c -- see hevc_transform.h.  Test infrastructure. */
#include "hevc_transform.h"

static inline int tcoef(int n, int dst_mode, int k, int i)
{
  if (dst_mode) return orc_dst_mat[k][i];
  return orc_dct_mat[k * (32 / n)][i];
}

/* Forward: stage 1 transforms rows (horizontal), shift log2N + BitDepth - 9;
 * stage 2 transforms columns (vertical), shift log2N + 6.   */
void orc_fwd_transform(const int16_t *resid, int16_t *coeff, int n, int dst_mode)
{
  int32_t tmp[32 * 32];
  int l2 = orc_log2((unsigned)n);
  int s1 = l2 + 8 - 9, s2 = l2 + 6;
  int r1 = s1 > 0 ? (1 << (s1 - 1)) : 0, r2 = 1 << (s2 - 1);
  orc_tables_init();
  for (int y = 0; y < n; y++)
    for (int k = 0; k < n; k++) {
      int32_t acc = 0;
      for (int x = 0; x < n; x++) acc += tcoef(n, dst_mode, k, x) * resid[y * n + x];
      tmp[y * n + k] = (acc + r1) >> s1;
    }
  for (int x = 0; x < n; x++)
    for (int k = 0; k < n; k++) {
      int32_t acc = 0;
      for (int y = 0; y < n; y++) acc += tcoef(n, dst_mode, k, y) * tmp[y * n + x];
      coeff[k * n + x] = (int16_t)orc_clip3(-32768, 32767, (acc + r2) >> s2);
    }
}

/* H.265 8.6.4.2: columns first (shift 7, clip to 16 bit), then rows (shift 20 - BitDepth). */
void orc_inv_transform(const int16_t *coeff, int16_t *resid, int n, int dst_mode)
{
  int32_t g[32 * 32];
  orc_tables_init();
  for (int x = 0; x < n; x++)
    for (int y = 0; y < n; y++) {
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += tcoef(n, dst_mode, k, y) * coeff[k * n + x];
      g[y * n + x] = orc_clip3(-32768, 32767, (acc + 64) >> 7);
    }
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += tcoef(n, dst_mode, k, x) * g[y * n + k];
      resid[y * n + x] = (int16_t)orc_clip3(-32768, 32767, (acc + (1 << 11)) >> 12);
    }
}

/* Encoder scalar quantiser with dead zone (HM/Kvazaar convention, flat scaling):
 * level = sign * ((|c| * f[qp%6] + offset) >> (14 + qp/6 + ts)), ts = 15 - BitDepth - log2N,
 * offset = (171 intra | 85 inter) << (shift - 9). */
/* With scaling lists (`scaling-list default`; m: the n x n scaling factors of hevc_scaling.h, NULL = flat 16): the forward scale of a position is
 * (f[qp%6] << 4) / m -- how Kvazaar (and HM) build their quantisation matrices from the lists --, the dequantiser is the normative one. */
int orc_quant_m(const int16_t *coeff, int16_t *level, int n, int qp, int intra, const uint8_t *m)
{
  int l2 = orc_log2((unsigned)n);
  int shift = 14 + qp / 6 + (15 - 8 - l2);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int nz = 0;
  for (int i = 0; i < n * n; i++) {
    int f = (orc_quant_scale[qp % 6] << 4) / (m ? m[i] : 16);
    int c = coeff[i], a = c < 0 ? -c : c;
    int64_t q = ((int64_t)a * f + off) >> shift;
    if (q > 32767) q = 32767;
    level[i] = (int16_t)(c < 0 ? -q : q);
    nz += (q != 0);
  }
  return nz;
}
int orc_quant_aux_m(const int16_t *coeff, int16_t *level, uint16_t *aux, int n, int qp, int intra, const uint8_t *m)
{
  int l2 = orc_log2((unsigned)n);
  int shift = 14 + qp / 6 + (15 - 8 - l2);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int nz = 0;
  for (int i = 0; i < n * n; i++) {
    int f = (orc_quant_scale[qp % 6] << 4) / (m ? m[i] : 16);
    int c = coeff[i], a = c < 0 ? -c : c;
    int64_t q = ((int64_t)a * f + off) >> shift;
    if (q > 32767) q = 32767;
    int64_t du = (((int64_t)a * f) >> (shift - 8)) - (q << 8);
    if (du < -256) du = -256;
    if (du > 511) du = 511;
    level[i] = (int16_t)(c < 0 ? -q : q);
    aux[i] = (uint16_t)((du + 256) | (c < 0 ? 0x8000 : 0));
    nz += (q != 0);
  }
  return nz;
}

int orc_quant(const int16_t *coeff, int16_t *level, int n, int qp, int intra)
{
  int l2 = orc_log2((unsigned)n);
  int shift = 14 + qp / 6 + (15 - 8 - l2);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int f = orc_quant_scale[qp % 6], nz = 0;
  for (int i = 0; i < n * n; i++) {
    int c = coeff[i], a = c < 0 ? -c : c;
    int64_t q = ((int64_t)a * f + off) >> shift;
    if (q > 32767) q = 32767;
    level[i] = (int16_t)(c < 0 ? -q : q);
    nz += (q != 0);
  }
  return nz;
}

int orc_quant_aux(const int16_t *coeff, int16_t *level, uint16_t *aux, int n, int qp, int intra)
{
  int l2 = orc_log2((unsigned)n);
  int shift = 14 + qp / 6 + (15 - 8 - l2);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int f = orc_quant_scale[qp % 6], nz = 0;
  for (int i = 0; i < n * n; i++) {
    int c = coeff[i], a = c < 0 ? -c : c;
    int64_t q = ((int64_t)a * f + off) >> shift;
    if (q > 32767) q = 32767;
    int64_t du = (((int64_t)a * f) >> (shift - 8)) - (q << 8);
    if (du < -256) du = -256;
    if (du > 511) du = 511;
    level[i] = (int16_t)(c < 0 ? -q : q);
    aux[i] = (uint16_t)((du + 256) | (c < 0 ? 0x8000 : 0));
    nz += (q != 0);
  }
  return nz;
}

int orc_adjust_levels(int16_t *level, const uint16_t *aux, int n, int scan_idx, int rdoq, int signhide)
{
  const int nsb = n >> 2;
  const uint8_t *px = orc_scan_x[scan_idx][2], *py = orc_scan_y[scan_idx][2];
  int nz_total = 0;
  for (int ys = 0; ys < nsb; ys++) for (int xs = 0; xs < nsb; xs++) {
    int16_t *lv[16]; int du[16], neg[16], nz = 0, ones = 1;
    for (int k = 0; k < 16; k++) {
      const int i = ((ys << 2) + py[k]) * n + (xs << 2) + px[k];
      lv[k] = &level[i]; du[k] = (int)(aux[i] & 0x3ff) - 256; neg[k] = aux[i] >> 15;
      if (*lv[k]) { nz++; if (*lv[k] != 1 && *lv[k] != -1) ones = 0; }
    }
    if (rdoq && (xs | ys) && nz >= 1 && nz <= 2 && ones) {
      int benefit = 0;
      for (int k = 0; k < 16; k++) if (*lv[k]) benefit += 2 * (du[k] + 256) - 256;
      if (benefit < 92 * nz + 92) { for (int k = 0; k < 16; k++) *lv[k] = 0; nz = 0; }
    }
    if (signhide && nz >= 2) {
      int first = -1, last = -1, sum = 0;
      for (int k = 0; k < 16; k++) if (*lv[k]) { if (first < 0) first = k; last = k; sum += orc_abs(*lv[k]); }
      if (last - first >= 4 && (sum & 1) != (*lv[first] < 0)) {
        /* what moving one level by one costs: the change of the squared error in 1/256 step^2 (the level is off by d = du / 256 steps:
         * up 1 - 2 d, down 1 + 2 d) plus lambda times the bins gained or lost, lambda / step^2 = 0.09: one bin for a step between
         * non-zero magnitudes (23), three for a +-1 that goes (69), three and a half for one that appears (80) */
        int best = -1, best_cost = 1 << 30, best_change = 0;
        for (int k = 15; k >= 0; k--) {
          const int a = orc_abs(*lv[k]);
          int cost[2], ok[2];                               /* [0] up, [1] down */
          if (a) {
            ok[0] = a < 32767; cost[0] = 256 - 2 * du[k] + 23;
            ok[1] = !(a == 1 && (k == first || k == last)); cost[1] = 256 + 2 * du[k] - (a == 1 ? 69 : 23);
          } else { ok[0] = k > first; cost[0] = 256 - 2 * du[k] + 80; ok[1] = 0; cost[1] = 0; }
          for (int o = 0; o < 2; o++) if (ok[o] && cost[o] < best_cost) { best_cost = cost[o]; best = k; best_change = o ? -1 : 1; }
        }
        if (best >= 0) {
          const int a = orc_abs(*lv[best]);
          if (a == 0) { *lv[best] = (int16_t)(neg[best] ? -1 : 1); nz++; }
          else { const int na = a + best_change; *lv[best] = (int16_t)(*lv[best] < 0 ? -na : na); if (na == 0) nz--; }
        }
      }
    }
    nz_total += nz;
  }
  return nz_total;
}

/* H.265 8.6.4.2: d = Clip3(-32768, 32767, (level * m[x][y] * levelScale[qP % 6] << (qP / 6)) + (1 << (bdShift - 1))) >> bdShift),
 * bdShift = BitDepth + log2N - 5; m = the scaling factors of the block (hevc_scaling.h, n x n raster) or NULL: 16 everywhere
 * (scaling_list_enabled_flag == 0) */
void orc_dequant_m(const int16_t *level, int16_t *coeff, int n, int qp, const uint8_t *m)
{
  int l2 = orc_log2((unsigned)n);
  int bd = 8 + l2 - 5;
  int scale = orc_level_scale[qp % 6] << (qp / 6);
  for (int i = 0; i < n * n; i++) {
    int64_t v = ((int64_t)level[i] * (m ? m[i] : 16) * scale + ((int64_t)1 << (bd - 1))) >> bd;
    coeff[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
}
void orc_dequant(const int16_t *level, int16_t *coeff, int n, int qp) { orc_dequant_m(level, coeff, n, qp, NULL); }

int orc_chroma_qp(int qp_y, int offset)
{
  int qpi = orc_clip3(0, 57, qp_y + offset);
  return orc_chroma_qp_table[qpi];
}
