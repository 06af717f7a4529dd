/* oracle/hevc_scaling.h -- scaling lists (H.265 7.3.4 scaling_list_data, 7.4.5 semantics, Tables 7-5 / 7-6) and the scaling factors
 * m[x][y] of the scaling process for transform coefficients (8.6.4.2).  Test infrastructure (CPU checker).
 * uvgComm side: the settings dialog's "scaling list" checkbox turns into Kvazaar's `scaling-list default`
 * (/root/reference/src/media/processing/kvazaarfilter.cpp:235-242): scaling_list_enabled_flag = 1 with the default lists. */
#ifndef ORC_HEVC_SCALING_H
#define ORC_HEVC_SCALING_H
#include "hevc_bits.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ScalingList[sizeId][matrixId][i] in up-right diagonal scan order: 16 entries for sizeId 0, 64 for sizeId 1..3; the DC entries of sizeId 2, 3.
 * matrixId = 3 * (inter ? 1 : 0) + cIdx for sizeId 0..2; sizeId 3 (32x32, luma only in 4:2:0): 0 intra, 1 inter. */
typedef struct {
  uint8_t list[4][6][64];
  uint8_t dc[2][6];
} orc_scaling_lists;

void orc_scaling_default(orc_scaling_lists *sl);                                /* Tables 7-5 / 7-6, DC 16 */
int  orc_scaling_parse(orc_bitr *r, orc_scaling_lists *sl);                     /* scaling_list_data(); < 0: invalid */
/* the writer's choices per list (generator): pred_mode 0 with delta (0 = default lists), or 1 with explicit entries */
void orc_scaling_write(orc_bitw *w, const orc_scaling_lists *sl, const uint8_t pred_mode[4][6], const uint8_t pred_delta[4][6]);
/* ScalingFactor of one transform block size as an n x n raster (m[y * n + x]), n = 4 << sizeId */
void orc_scaling_factor(const orc_scaling_lists *sl, int size_id, int matrix_id, uint8_t *m);
static inline int orc_scaling_matrix_id(int size_id, int cidx, int inter) { return size_id == 3 ? (inter ? 1 : 0) : 3 * (inter ? 1 : 0) + cidx; }

#ifdef __cplusplus
}
#endif
#endif
