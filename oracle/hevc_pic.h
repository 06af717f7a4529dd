/* oracle/hevc_pic.h -- picture + per-4x4 side information shared by the oracle's encoder and
 * decoder (reconstruction, merge/AMVP derivation, boundary strength).  Test infrastructure. */
#ifndef ORC_HEVC_PIC_H
#define ORC_HEVC_PIC_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int16_t mv[2];        /* L0 motion vector, quarter luma samples */
  int8_t  ref_idx;      /* L0 reference index, -1 = list 0 not used (predFlagL0 == 0) */
  int8_t  ref_idx1;     /* L1 reference index, -1 = list 1 not used (always in I and P slices); both -1 = not inter predicted */
  int16_t mv1[2];       /* L1 motion vector */
} orc_mvinfo;

typedef struct {
  int w, h;                          /* coded luma size */
  pixel *plane[3]; int stride[3];
  int poc;
  int is_ref, needed_for_output, in_use;
  int stand_in_fresh;                     /* made by missing_ref for the picture being started: no source for another stand-in of the same picture */
  int slice_type;                    /* decoder: of the picture's (last) slice */
  int out_queued;                    /* decoder: the picture has left the DPB's output process (C.5.2.4) and waits for the caller to fetch it */
  int64_t pts;
  /* per 4x4 luma block side info, stride b4_stride */
  int b4_w, b4_h;
  uint8_t *pred_mode;                /* MODE_INTRA / MODE_INTER / MODE_SKIP; 255 = not yet decoded */
  uint8_t *ct_depth;                 /* coding quadtree depth */
  uint8_t *intra_mode;               /* IntraPredModeY */
  int8_t  *qp_y;
  uint8_t *tu_nz;                    /* luma TB of this 4x4 has non-zero coefficients */
  uint8_t *edge_v, *edge_h;          /* bit0: transform edge at left/top of this 4x4, bit1: prediction edge */
  uint8_t *no_filter;                /* cu_transquant_bypass / pcm with loop filter disabled */
  orc_mvinfo *mvf;
  int ref_poc_list[16];              /* POC of RefPicList0[i] of the (single) slice of this picture: boundary strength compares reference PICTURES
                                      * (8.7.2.4), temporal motion vector prediction scales by POC distances (8.5.3.2.9) */
  int ref_poc_list1[16];             /* ... of RefPicList1[i] (B slices) */
  int is_lt;                         /* decoder: marked "used for long-term reference" (8.3.2) */
  uint8_t ref_lt_list[16], ref_lt_list1[16];   /* RefPicListX[i] was a long-term reference picture when this picture was the current one (LongTermRefPic of 8.5.3.2.9) */
} orc_pic;

int  orc_pic_alloc(orc_pic *p, int w, int h);
void orc_pic_free(orc_pic *p);
void orc_pic_reset_side(orc_pic *p);

/* H.265 8.7.2.4: boundary strengths from the side info. bs_v: [(h/4) x (w/8)], bs_h: [(h/8) x (w/4)].
 * Picture edges get 0.  (Single slice, single tile, or filtering across them enabled.) */
void orc_compute_bs(const orc_pic *p, uint8_t *bs_v, uint8_t *bs_h);
#ifdef __cplusplus
}
#endif
#endif
