/* oracle/hevc_transform.h -- forward/inverse integer transforms and (de)quantisation.
 * Inverse + dequant are normative (H.265 8.6.2-8.6.4); forward + quant are the encoder's
 * choice and follow the HM/Kvazaar convention quoted in SURVEY.md Appendix B.
 * Test infrastructure. */
#ifndef ORC_HEVC_TRANSFORM_H
#define ORC_HEVC_TRANSFORM_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* All blocks are row-major n x n, stride n.  dst_mode: use 4x4 DST-VII (intra luma 4x4). */
void orc_fwd_transform(const int16_t *resid, int16_t *coeff, int n, int dst_mode);
void orc_inv_transform(const int16_t *coeff, int16_t *resid, int n, int dst_mode);
/* returns number of non-zero levels */
int  orc_quant(const int16_t *coeff, int16_t *level, int n, int qp, int intra);
void orc_dequant(const int16_t *level, int16_t *coeff, int n, int qp);
void orc_dequant_m(const int16_t *level, int16_t *coeff, int n, int qp, const uint8_t *m);   /* m: n x n scaling factors (hevc_scaling.h) or NULL = flat 16 */
/* The quantiser with what the level-adjustment pass needs beside the levels: aux[i] = 256 + du in bits 0..9, where du = (|c| * scale >> (shift - 8))
 * - (|level| << 8) is the part of the coefficient the level does not account for, in 1/256 quantiser steps (-0.34 .. 0.84 steps with the
 * dead-zone rounding), and bit 15 = the coefficient is negative. */
int  orc_quant_aux(const int16_t *coeff, int16_t *level, uint16_t *aux, int n, int qp, int intra);
/* ... with the scaling factors m of the block (hevc_scaling.h; NULL = flat): forward scale (f << 4) / m per position */
int  orc_quant_m(const int16_t *coeff, int16_t *level, int n, int qp, int intra, const uint8_t *m);
int  orc_quant_aux_m(const int16_t *coeff, int16_t *level, uint16_t *aux, int n, int qp, int intra, const uint8_t *m);
/* "uvgx RDOQ v1" and sign data hiding: one pass over the levels of a transform block, one 4x4 coefficient group at a time (scan_idx 0 diagonal,
 * 1 horizontal, 2 vertical: the scan the block will be coded with).  Per group, in this order:
 *   rdoq: a group other than the block's DC group whose non-zero levels are one or two +-1 is dropped when that costs less than coding it:
 *         sum over them of (2 u - 256) < 92 n + 92, u = the coefficient in 1/256 steps (what zeroing adds to the squared error against
 *         lambda times about four bins per coefficient and four for the group; lambda / step^2 = 0.57 * 2^(-8/3) whatever the QP);
 *   signhide: where the first and the last non-zero level of the group are at least four scan positions apart the decoder derives the sign
 *         of the first one from the parity of the sum of the magnitudes (9.3.4.3 / 7.3.8.11); when the parity says the wrong sign ONE level of
 *         the group moves by one step up or down, whichever move costs least over all sixteen positions (positions 15 .. 0, up before down,
 *         the first minimum is kept): cost = change of the squared error in 1/256 step^2 -- 256 - 2 du up, 256 + 2 du down -- plus lambda times the
 *         bins the move adds or saves (+23 between non-zero magnitudes up, -23 down, -69 for a +-1 that disappears, +80 for a zero that becomes
 *         +-1 with the coefficient's sign).  Not allowed: a +-1 at the first or last position going to zero, a zero at or below the first
 *         position (the positions that define the hidden sign must stay).
 * Returns the number of non-zero levels left. */
int  orc_adjust_levels(int16_t *level, const uint16_t *aux, int n, int scan_idx, int rdoq, int signhide);
int  orc_chroma_qp(int qp_y, int offset);   /* H.265 8.6.1 (ChromaArrayType 1) */
#ifdef __cplusplus
}
#endif
#endif
