/* oracle/hevc_cabac.h -- CABAC engines, spec-literal (H.265 9.3.2 init, 9.3.4.3 arithmetic
 * decoding, 9.3.4.x (informative) arithmetic encoding).  Test infrastructure. */
#ifndef ORC_HEVC_CABAC_H
#define ORC_HEVC_CABAC_H
#include "hevc_common.h"
#include "hevc_bits.h"
#ifdef __cplusplus
extern "C" {
#endif

/* context variable layout (own enumeration; sizes follow H.265 Tables 9-5..9-37, version 1) */
enum {
  CTX_SAO_MERGE = 0,          /* 1 */
  CTX_SAO_TYPE = 1,           /* 1 */
  CTX_SPLIT_CU = 2,           /* 3 */
  CTX_TQ_BYPASS = 5,          /* 1 */
  CTX_SKIP = 6,               /* 3 */
  CTX_PRED_MODE = 9,          /* 1 */
  CTX_PART_MODE = 10,         /* 4 */
  CTX_PREV_INTRA = 14,        /* 1 */
  CTX_CHROMA_MODE = 15,       /* 1 */
  CTX_RQT_ROOT_CBF = 16,      /* 1 */
  CTX_MERGE_FLAG = 17,        /* 1 */
  CTX_MERGE_IDX = 18,         /* 1 */
  CTX_INTER_PRED_IDC = 19,    /* 5 */
  CTX_REF_IDX = 24,           /* 2 */
  CTX_MVP_FLAG = 26,          /* 1 */
  CTX_SPLIT_TRANSFORM = 27,   /* 3 */
  CTX_CBF_LUMA = 30,          /* 2 */
  CTX_CBF_CHROMA = 32,        /* 4 */
  CTX_MVD_GT0 = 36,           /* 1 */
  CTX_MVD_GT1 = 37,           /* 1 */
  CTX_CU_QP_DELTA = 38,       /* 2 */
  CTX_TS_FLAG = 40,           /* 2 */
  CTX_LAST_X = 42,            /* 18 */
  CTX_LAST_Y = 60,            /* 18 */
  CTX_CSBF = 78,              /* 4 */
  CTX_SIG = 82,               /* 42 */
  CTX_GT1 = 124,              /* 24 */
  CTX_GT2 = 148,              /* 6 */
  CTX_COUNT = 154
};

typedef struct { uint8_t state; uint8_t mps; } orc_ctx;

/* H.265 9.3.2.2: initialise all context variables for a slice. init_type 0 (I), 1, 2. */
void orc_cabac_init_contexts(orc_ctx *ctx, int init_type, int slice_qp);
extern const uint8_t orc_cabac_init_values[3][CTX_COUNT];

typedef struct {
  orc_bitw *bw;
  uint32_t low, range;
  int first_bit, outstanding;
  orc_ctx ctx[CTX_COUNT];
  uint64_t bins;   /* statistics */
} orc_cabac_enc;

void orc_cenc_start(orc_cabac_enc *c, orc_bitw *bw);     /* 9.3.4.x InitEncoder (contexts untouched) */
void orc_cenc_bin(orc_cabac_enc *c, int ctx_idx, int bin);
void orc_cenc_bypass(orc_cabac_enc *c, int bin);
void orc_cenc_bypass_bits(orc_cabac_enc *c, uint32_t val, int n);
void orc_cenc_terminate(orc_cabac_enc *c, int bin);       /* bin==1 flushes; stop bit written */

typedef struct {
  orc_bitr br;
  uint32_t range, offset;
  orc_ctx ctx[CTX_COUNT];
} orc_cabac_dec;

void orc_cdec_start(orc_cabac_dec *c, const uint8_t *buf, size_t len);   /* 9.3.2.5 */
int  orc_cdec_bin(orc_cabac_dec *c, int ctx_idx);
int  orc_cdec_bypass(orc_cabac_dec *c);
uint32_t orc_cdec_bypass_bits(orc_cabac_dec *c, int n);
int  orc_cdec_terminate(orc_cabac_dec *c);
/* byte position right after the terminating bin==1 (stop bit consumed, alignment skipped) */
size_t orc_cdec_bytes_consumed(const orc_cabac_dec *c);

#ifdef __cplusplus
}
#endif
#endif
