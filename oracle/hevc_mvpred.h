/* oracle/hevc_mvpred.h -- merge candidates (H.265 8.5.3.2.2-8.5.3.2.5) and luma motion vector
 * prediction (8.5.3.2.6-8.5.3.2.7) for list-0 uni-prediction, with the temporal candidate of 8.5.3.2.8-8.5.3.2.9.
 * Test infrastructure. */
#ifndef ORC_HEVC_MVPRED_H
#define ORC_HEVC_MVPRED_H
#include "hevc_pic.h"
#include "hevc_intra.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
  const orc_pic *pic;            /* side info of the picture being coded */
  orc_avail_ctx av;              /* picture geometry for z-scan availability */
  int log2_par_mrg_level;
  int max_num_merge_cand;
  int num_ref_idx;               /* num_ref_idx_l0_active */
  int cur_poc; int ref_poc[16];  /* POC of RefPicList0 entries */
  const orc_pic *col;            /* collocated picture (slice_temporal_mvp_enabled_flag, collocated_ref_idx), NULL = no temporal candidates */
} orc_mvpred_ctx;

typedef struct { int16_t mv[2]; int8_t ref_idx; } orc_mvcand;

/* Fills cand[0..max_num_merge_cand-1]. */
void orc_merge_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                          int npbw, int npbh, int part_idx, int part_mode, orc_mvcand *cand);
/* Fills the two AMVP candidates for reference index ref_idx. */
void orc_amvp_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                         int npbw, int npbh, int part_idx, int ref_idx, int16_t cand[2][2]);
#ifdef __cplusplus
}
#endif
#endif
