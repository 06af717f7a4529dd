/* oracle/hevc_enc.h -- the encoder the HIP path must match bit for bit ("uvgx encoder algorithm
 * v1").  It is the CPU checker for what uvgComm's KvazaarFilter gets from
 * kvz_api->encoder_encode (/root/reference/src/media/processing/kvazaarfilter.cpp:435-448):
 * one access unit of Annex-B NAL units per input picture plus the reconstructed picture.
 *
 * Kvazaar's own search heuristics are not available (source absent), so the
 * decisions below are this project's, chosen to be data-parallel; everything normative
 * (prediction, transforms, deblocking, CABAC, syntax) follows H.265.  PARITY UNPINNED vs Kvazaar.
 *
 * Tool set (mirrors Kvazaar preset=ultrafast as far as SURVEY.md Appendix A records it):
 *   CTU 64, CUs 32/16 (inter, 2Nx2N) and 32/16/8 (intra, 2Nx2N), TU = CU (chroma half),
 *   integer-sample full-search motion estimation over +-range, 1 reference (previous picture),
 *   merge/skip with 5 candidates, AMVP, no TMVP, plain dead-zone quantiser, deblocking on,
 *   SAO off, sign hiding off, transform skip off, WPP on, one slice per picture,
 *   IDR every `period` pictures with VPS/SPS/PPS, constant QP or picture-level rate control (bitrate > 0).
 * Test infrastructure. */
#ifndef ORC_HEVC_ENC_H
#define ORC_HEVC_ENC_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_encoder orc_encoder;

typedef struct {
  int width, height;          /* input luma size (multiple of 2) */
  int qp;                     /* 0..51 */
  int intra_period;           /* 0: only the first picture is intra; 1: all intra; n: every n-th */
  int vps_period;             /* 0: parameter sets only with the first picture; n: with every n-th intra */
  int search_range;           /* integer samples, <= 32 */
  int fps_num, fps_den;
  int wpp;                    /* entropy_coding_sync_enabled_flag */
  int deblock;                /* 1 = enabled */
  int tile_rows;              /* 1 = no tiles; n > 1: n full-width tile rows, uniform spacing, loop filter across tiles on,
                               * motion vectors constrained to the tile (see me_block32) */
  int tile_cols;              /* 1 = none; n > 1: n tile columns, uniform spacing (kvazaar tiles=CxR): with tile_rows a C x R grid, coded in tile-scan order;
                               * motion vectors are constrained to the tile in x as in y */
  int qp_in_cu;               /* 1: cu_qp_delta_enabled_flag, quantisation group = CTU: a delta-QP map set with orc_enc_set_roi()
                               * gives every CTU its own QP (kvz_picture.roi, kvazaarfilter.cpp:423-431) */
  int bitrate;                /* bits per second; 0 = constant QP.  > 0: "uvgx rate control v1" (see hevc_enc.c) */
  int satd;                   /* 1 (default): the intra mode search compares 8x8 Hadamard sums (SATD), as Kvazaar's rough search does; 0: SAD */
  int me_early;               /* kvazaar me-early-termination (on by default, as in Kvazaar): a 32x32 block whose SAD against the co-located
                               * block of the reference is at most 64 * lambda_q4 (about what quantisation noise alone leaves at this QP) is
                               * coded unsplit with the zero vector, without a search */
  int vaq;                    /* 0 off, 1..20: variance adaptive quantisation "uvgx VAQ v1" (see vaq_deltas() in hevc_enc.c): CTUs with less
                               * texture than the picture's average get a lower QP, busier ones a higher one; implies qp_in_cu */
  int mv_frame;               /* kvazaar mv-constraint frame / frametile (1) / frametilemargin (2): a candidate is dropped when the
                               * displaced 32x32 block would leave the coded picture (2: plus 4 samples on an axis whose displacement is
                               * odd -- the chroma half-sample taps) */
  int test_mv_jitter;         /* TEST HOOK (decoder tests): after the integer search every block's vector gets a pseudo-random
                               * offset of -3..3 quarter samples per component, so that the stream exercises fractional-sample
                               * interpolation (8.5.3.3.3) -- the encoder algorithm proper never produces fractional vectors */
  int sao;                    /* 1: sample adaptive offset on, parameters by "uvgx SAO decision v1" (hevc_sao.c) */
  int slices;                 /* kvazaar slices (uvgComm video/Slices, kvazaarfilter.cpp:205-215): 0 one slice per picture; 1 = "wpp": a DEPENDENT slice segment
                               * per CTU row (needs wpp); 2 = "tiles": an independent slice per tile (needs tile_rows > 1).  One NAL unit per segment;
                               * what is coded below the slice level does not change */
  int rc_bands;               /* with bitrate > 0: "uvgx rate control v2" -- the CTU rows of a P picture are reconstructed in this many groups, one after
                               * the other, and after each group the QP of the next one moves by at most one step (within +-3 of the picture's QP)
                               * according to what the levels coded so far will cost against the picture's target (rc_band_decide() in hevc_enc.c);
                               * 0 = picture-level control only (v1).  Implies qp_in_cu: the steps travel as cu_qp_delta */
  int subme;                  /* kvazaar subme 0..4: fractional-sample refinement of every searched CU's vector, "uvgx subme v1" (subme_refine()
                               * in hevc_enc.c): 0 off (Kvazaar's ultrafast), 1 half-sample positions left/right/above/below, 2 + the four
                               * half-sample diagonals, 3 + quarter-sample left/right/above/below of the best so far, 4 + its quarter-sample
                               * diagonals; candidates are compared by SATD (8x8 Hadamard) + lambda * vector bits */
  int intra_chain;            /* "intra-chain" (default 1): mode restriction for the blocks whose below-left / above-right samples lie in another CTU (intra_analyse_size) */
  int scaling_list;           /* kvazaar scaling-list default: scaling_list_enabled_flag with the default lists; quantiser scale (f << 4) / m per position (orc_quant_m) */
  int rdoq;                   /* kvazaar rdoq: "uvgx RDOQ v1" -- sparse high-frequency coefficient groups are dropped when that is cheaper (orc_adjust_levels, hevc_transform.h) */
  int signhide;               /* kvazaar signhide: sign_data_hiding_enabled_flag; the quantiser makes the parity of every eligible coefficient group say the hidden sign */
  int rc_delay;               /* rate control: the size of access unit t - rc_delay is what is booked before picture t (0 = 3: the form of rounds 1-2; 3 .. 7: an encoder with rc_delay - 1 pictures in flight decides like a synchronous one with the same delay) */
  int intra_in_p;             /* 1: "uvgx intra-in-P v1" -- intra coding units in P pictures (hevc_enc.c intra_p_decide): a 16x16 quarter of a searched 32x32 block whose inter cost
                               * is above 24 lambda is priced as an intra block (the intra picture's source-based analysis) and coded intra when that is cheaper; ignored with rc_bands */
  int hash;                   /* kvazaar hash: 0 none, 1 checksum, 2 md5 -- a decoded picture hash SEI (D.2.19, suffix SEI NAL unit) after every picture's slices */
  int me_source;              /* "uvgx search pipelining v1" (option me-source): the integer motion search looks at the previous INPUT picture instead of the
                               * reference picture's reconstruction (build_refpad() in hevc_enc.c); prediction always uses the reconstruction */
  int lossless;               /* kvazaar lossless (uvgComm's check box, kvazaarfilter.cpp:244): every coding unit with cu_transquant_bypass_flag -- the residual IS the
                               * level array, the reconstruction is the source picture; the decisions (modes, vectors, splits) are the lossy encoder's; no deblocking, no
                               * SAO (they would leave these samples alone anyway, 8.7.2.5.7 / 8.7.3), no RDOQ, no sign hiding, no rate control */
} orc_enc_config;

typedef struct {
  /* geometry of the coded (padded) picture */
  int coded_w, coded_h;
  int is_intra, poc;
  /* per 8x8 block arrays, stride coded_w/8 */
  const uint8_t *cu_log2, *cu_intra, *cu_flags /* bit0 skip, bit1 merge */, *cu_merge_idx, *cu_mvp_idx,
                *cu_intra_mode, *cu_cbf /* bit0 Y, bit1 Cb, bit2 Cr */;
  const int16_t *cu_mv;       /* [b8][2], quarter samples */
  const int16_t *coef[3];     /* quantised levels, plane shaped */
  const pixel *predeblock[3]; /* reconstruction before deblocking */
  const pixel *recon[3];      /* final reconstruction (coded size) */
  const uint8_t *bs_v, *bs_h;
  uint64_t bins;              /* CABAC bins in this picture */
} orc_enc_debug;

void orc_enc_default_config(orc_enc_config *c);
/* options added after the packed open calls ran out of bits: by name, before the first picture.  "hash" 0/1/2, "rdoq" 0/1, "signhide" 0/1, "intra-in-p" 0/1.  Returns 1 when known. */
int orc_enc_set_option(orc_encoder *e, const char *name, int value);
orc_encoder *orc_enc_open(const orc_enc_config *c);
void orc_enc_close(orc_encoder *e);
/* Encodes one picture given as three packed planes (stride = width, width/2).  The returned
 * buffer is owned by the encoder and valid until the next call.  Returns AU size in bytes. */
/* delta-QP map for the following pictures: w x h cells spread uniformly over the picture, one int8 per cell (clamped to
 * [-12, 12]); w == 0 removes it.  CTU (cx, cy) uses cell (cx * w / ctus_x, cy * h / ctus_y).  Needs cfg.qp_in_cu. */
void orc_enc_set_roi(orc_encoder *e, int w, int h, const int8_t *map);
size_t orc_enc_encode(orc_encoder *e, const pixel *y, const pixel *u, const pixel *v, const uint8_t **au);
void orc_enc_get_debug(orc_encoder *e, orc_enc_debug *dbg);
/* copy cropped reconstruction (width x height I420, packed) */
void orc_enc_get_recon(orc_encoder *e, pixel *y, pixel *u, pixel *v);

extern const uint16_t orc_lambda_q4[52];   /* 16 * sqrt(0.57 * 2^((qp-12)/3)), rounded */
int orc_mvd_bits(int q);                   /* bins to code one mvd component (quarter samples) */
#ifdef __cplusplus
}
#endif
#endif
