/* oracle/hevc_mvpred.h -- merge candidates (H.265 8.5.3.2.2-8.5.3.2.5, with the combined bi-predictive candidates of B slices) and luma
 * motion vector prediction (8.5.3.2.6-8.5.3.2.7) for either list, with the temporal candidate of 8.5.3.2.8-8.5.3.2.9.
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
  const orc_pic *col;            /* collocated picture (slice_temporal_mvp_enabled_flag, collocated_from_l0_flag, collocated_ref_idx), NULL = no temporal candidates */
  /* B slices (a context that is zeroed and filled for a P slice leaves these at 0) */
  int is_b, num_ref_idx1; int ref_poc1[16];   /* slice_type == B, num_ref_idx_l1_active, POC of RefPicList1 entries */
  int collocated_from_l0;        /* collocated_from_l0_flag (1 in P slices) */
  int no_backward_pred;          /* NoBackwardPredFlag (8.5.3.2.9): no entry of either list follows the current picture in output order */
  uint8_t ref_lt[16], ref_lt1[16];   /* the entry is a long-term reference picture (all 0: none -- the encoder's context) */
} orc_mvpred_ctx;

typedef orc_mvinfo orc_mvcand;   /* mv / ref_idx: list 0, mv1 / ref_idx1: list 1; an index of -1 = the list is not used */

/* Fills cand[0..max_num_merge_cand-1]. */
void orc_merge_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                          int npbw, int npbh, int part_idx, int part_mode, orc_mvcand *cand);
/* Fills the two AMVP candidates of list X (0 / 1) for reference index ref_idx; orc_amvp_candidates: list 0. */
void orc_amvp_candidates_lx(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                            int npbw, int npbh, int part_idx, int X, int ref_idx, int16_t cand[2][2]);
void orc_amvp_candidates(const orc_mvpred_ctx *c, int xcb, int ycb, int ncbs, int xpb, int ypb,
                         int npbw, int npbh, int part_idx, int ref_idx, int16_t cand[2][2]);
#ifdef __cplusplus
}
#endif
#endif
