/* oracle/hevc_scaling.c -- see hevc_scaling.h.  Restated from ITU-T H.265 (04/2013) 7.3.4, 7.4.5; the tables are typed from the
 * standard's Tables 7-5 and 7-6 (the 8x8 ones are symmetric matrices listed in diagonal scan order). */
#include <string.h>
#include "hevc_scaling.h"

/* Table 7-6, ScalingList[1..3][matrixId][i]: matrixId 0..2 (intra), 3..5 (inter) */
static const uint8_t k_default_intra[64] = {
  16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 16, 17, 16, 17, 18, 17, 18, 18, 17, 18, 21, 19, 20, 21, 20, 19, 21, 24, 22, 22, 24,
  24, 22, 22, 24, 25, 25, 27, 30, 27, 25, 25, 29, 31, 35, 35, 31, 29, 36, 41, 44, 41, 36, 47, 54, 54, 47, 65, 70, 65, 88, 88, 115 };
static const uint8_t k_default_inter[64] = {
  16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 20, 20, 20, 20, 20, 20, 20, 24, 24, 24, 24,
  24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 28, 28, 28, 28, 28, 28, 33, 33, 33, 33, 33, 41, 41, 41, 41, 54, 54, 54, 71, 71, 91 };

static int n_matrices(int size_id) { return size_id == 3 ? 2 : 6; }
static int n_coef(int size_id) { return size_id == 0 ? 16 : 64; }

static void default_list(orc_scaling_lists *sl, int size_id, int matrix_id)
{
  if (size_id == 0) memset(sl->list[0][matrix_id], 16, 16);                    /* Table 7-5 */
  else memcpy(sl->list[size_id][matrix_id], (size_id == 3 ? matrix_id >= 1 : matrix_id >= 3) ? k_default_inter : k_default_intra, 64);
  if (size_id >= 2) sl->dc[size_id - 2][matrix_id] = 16;
}

void orc_scaling_default(orc_scaling_lists *sl)
{
  memset(sl, 16, sizeof(*sl));
  for (int s = 0; s < 4; s++) for (int m = 0; m < n_matrices(s); m++) default_list(sl, s, m);
}

int orc_scaling_parse(orc_bitr *r, orc_scaling_lists *sl)
{
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < n_matrices(s); m++) {
      if (!orc_br_get(r, 1)) {                                                  /* scaling_list_pred_mode_flag = 0 */
        const int delta = (int)orc_br_ue(r);                                    /* scaling_list_pred_matrix_id_delta */
        if (delta > m) return -1;
        if (delta == 0) default_list(sl, s, m);
        else {
          memcpy(sl->list[s][m], sl->list[s][m - delta], 64);
          if (s >= 2) sl->dc[s - 2][m] = sl->dc[s - 2][m - delta];
        }
      } else {
        int next = 8;
        if (s >= 2) {
          const int dc = orc_br_se(r);                                          /* scaling_list_dc_coef_minus8 */
          if (dc < -7 || dc > 247) return -1;
          next = dc + 8; sl->dc[s - 2][m] = (uint8_t)next;
        }
        for (int i = 0; i < n_coef(s); i++) {
          const int d = orc_br_se(r);                                           /* scaling_list_delta_coef */
          if (d < -128 || d > 127) return -1;
          next = (next + d + 256) % 256;
          if (next == 0) return -1;                                             /* (ScalingList entries shall be greater than 0) */
          sl->list[s][m][i] = (uint8_t)next;
        }
      }
      if (r->error) return -1;
    }
  return 0;
}

void orc_scaling_write(orc_bitw *w, const orc_scaling_lists *sl, const uint8_t pred_mode[4][6], const uint8_t pred_delta[4][6])
{
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < n_matrices(s); m++) {
      orc_bw_put(w, pred_mode[s][m], 1);
      if (!pred_mode[s][m]) { orc_bw_ue(w, pred_delta[s][m]); continue; }
      int next = 8;
      if (s >= 2) { orc_bw_se(w, (int)sl->dc[s - 2][m] - 8); next = sl->dc[s - 2][m]; }
      for (int i = 0; i < n_coef(s); i++) {
        int d = (int)sl->list[s][m][i] - next;
        if (d > 127) d -= 256;
        if (d < -128) d += 256;
        orc_bw_se(w, d);
        next = sl->list[s][m][i];
      }
    }
}

void orc_scaling_factor(const orc_scaling_lists *sl, int size_id, int matrix_id, uint8_t *m)
{
  const int n = 4 << size_id;
  if (size_id == 0) {
    for (int i = 0; i < 16; i++) m[orc_scan_y[0][2][i] * 4 + orc_scan_x[0][2][i]] = sl->list[0][matrix_id][i];
    return;
  }
  const int rep = n >> 3;                                                       /* 1, 2, 4: every 8x8 entry covers rep x rep positions (7.4.5) */
  for (int i = 0; i < 64; i++) {
    const int x0 = orc_scan_x[0][3][i] * rep, y0 = orc_scan_y[0][3][i] * rep;
    for (int j = 0; j < rep; j++) for (int k = 0; k < rep; k++) m[(y0 + j) * n + x0 + k] = sl->list[size_id][matrix_id][i];
  }
  if (size_id >= 2) m[0] = sl->dc[size_id - 2][matrix_id];
}
