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
int  orc_chroma_qp(int qp_y, int offset);   /* H.265 8.6.1 (ChromaArrayType 1) */
#ifdef __cplusplus
}
#endif
#endif
