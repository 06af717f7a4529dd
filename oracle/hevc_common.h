/*
 * oracle/ -- TEST INFRASTRUCTURE ONLY.
 *
 * From-the-standard (ITU-T H.265) CPU restatement of the HEVC encode/decode
 * hot path that uvgComm reaches through KvazaarFilter -> kvz_api
 * (/root/reference/src/media/processing/kvazaarfilter.cpp:435-448) and
 * OpenHEVCFilter -> libOpenHevc* (/root/reference/src/media/processing/
 * openhevcfilter.cpp:145-146,195-199).
 *
 * PARITY UNPINNED vs. the reference's own binaries: Kvazaar 2.3.1
 * (dependencies/kvazaar.cmake:10-14) and OpenHEVC (dependencies/openhevc.cmake:
 * 10-14) are FetchContent dependencies whose sources are not in /root/reference,
 * and the reference's tests hold no vectors for this path.
 * What pins this oracle instead: the normative constants of H.265 (transform
 * matrices, filters, CABAC tables), closed-loop decode(encode(x)) == recon, and
 * a numpy second restatement of the arithmetic stages (tests/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in this directory.  The product (kvazzup_amd/) never links it.
 */
#ifndef ORC_HEVC_COMMON_H
#define ORC_HEVC_COMMON_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t pixel;

#define ORC_MIN(a,b) ((a)<(b)?(a):(b))
#define ORC_MAX(a,b) ((a)>(b)?(a):(b))
static inline int orc_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int orc_clip_pixel(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int orc_abs(int v) { return v < 0 ? -v : v; }
static inline int orc_log2(unsigned v) { int n = 0; while (v > 1) { v >>= 1; n++; } return n; }

enum { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2 };
enum { MODE_INTER = 0, MODE_INTRA = 1, MODE_SKIP = 2 };
enum { PART_2Nx2N = 0, PART_2NxN = 1, PART_Nx2N = 2, PART_NxN = 3,
       PART_2NxnU = 4, PART_2NxnD = 5, PART_nLx2N = 6, PART_nRx2N = 7 };
enum { NAL_TRAIL_N = 0, NAL_TRAIL_R = 1, NAL_RADL_N = 6, NAL_RADL_R = 7, NAL_RASL_N = 8, NAL_RASL_R = 9, NAL_BLA_W_LP = 16, NAL_BLA_W_RADL = 17, NAL_BLA_N_LP = 18, NAL_IDR_W_RADL = 19, NAL_IDR_N_LP = 20,
       NAL_CRA = 21, NAL_RSV_IRAP_VCL23 = 23, NAL_VPS = 32, NAL_SPS = 33, NAL_PPS = 34,
       NAL_AUD = 35, NAL_EOS = 36, NAL_EOB = 37, NAL_FD = 38, NAL_SEI_PREFIX = 39, NAL_SEI_SUFFIX = 40 };

/* ---- tables (hevc_tables.c) ---- */
extern int8_t orc_dct_mat_rw[32][32];     /* H.265 8.6.4.2 transMatrix (32x32), filled by orc_tables_init(); N-point uses rows k*32/N, cols < N */
#define orc_dct_mat orc_dct_mat_rw
extern const int8_t  orc_dst_mat[4][4];       /* H.265 8.6.4.2 DST-VII 4x4 */
extern const int16_t orc_quant_scale[6];      /* encoder forward quant multipliers (HM convention) */
extern const uint8_t orc_level_scale[6];      /* H.265 8.6.3 levelScale[] */
extern const int8_t  orc_intra_angle[35];     /* H.265 Table 8-4 intraPredAngle (index = mode) */
extern const int16_t orc_inv_angle[35];       /* H.265 Table 8-5 invAngle (modes 11..25) */
extern const int8_t  orc_luma_filter[4][8];   /* H.265 Table 8-11 */
extern const int8_t  orc_chroma_filter[8][4]; /* H.265 Table 8-12 */
extern const uint8_t orc_beta_table[52];      /* H.265 Table 8-12/8-23 beta' */
extern const uint8_t orc_tc_table[54];        /* tc' */
extern const uint8_t orc_chroma_qp_table[58]; /* H.265 Table 8-10 QpC as function of qPi (ChromaArrayType==1) */
extern const uint8_t orc_range_tab_lps[64][4];/* H.265 Table 9-46 */
extern const uint8_t orc_trans_idx_lps[64];   /* H.265 Table 9-47 */
extern const uint8_t orc_trans_idx_mps[64];
/* scan orders: [scanIdx 0 diag,1 horiz,2 vert][log2 size 1..3 -> idx 0..2 (2x2,4x4,8x8)] -> packed (y<<4|x)? see .c */
extern uint8_t orc_scan_x[3][4][64];          /* [scanIdx][log2BlockSize(0..3)][pos] */
extern uint8_t orc_scan_y[3][4][64];
void orc_tables_init(void);

#ifdef __cplusplus
}
#endif
#endif
