/* oracle/hevc_gen.h -- conformance-style HEVC stream synthesiser (see hevc_gen.c).  Test infrastructure:
 * it produces the streams the product's decoder must accept from a foreign encoder (a Kvazaar peer,
 * /root/reference/src/media/processing/openhevcfilter.cpp:134-172) but this project's own encoder never writes. */
#ifndef ORC_HEVC_GEN_H
#define ORC_HEVC_GEN_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_gen orc_gen;

/* Every switch: -1 = drawn from the seed, otherwise as given.  All int: tests pass the struct as a flat array. */
typedef struct {
  int width, height;          /* coded size: multiples of 8 (the minimum coding block), not necessarily of 64 */
  int seed;
  int intra_period;           /* IDR every n-th picture; other pictures are P slices (8 % of them I slices in TRAIL_R) */
  int qp;                     /* PPS init_qp; slices add -4..4, CUs their cu_qp_delta */
  int density;                /* probability (%) of a significant coefficient */
  int num_refs;               /* 1..4 reference pictures kept (RPS in the slice header, Kvazaar style); active entries drawn per slice */
  int tmvp;                   /* sps_temporal_mvp_enabled_flag (slices switch it on 80 % of the time, random collocated_ref_idx) */
  int amp;                    /* amp_enabled_flag */
  int sao;                    /* sample_adaptive_offset_enabled_flag */
  int strong_intra;           /* strong_intra_smoothing_enabled_flag */
  int sign_hiding;            /* sign_data_hiding_enabled_flag */
  int transform_skip;         /* transform_skip_enabled_flag */
  int cabac_init;             /* cabac_init_present_flag (cabac_init_flag per slice) */
  int wpp;                    /* entropy_coding_sync_enabled_flag */
  int tile_rows;              /* full-width tile rows, 1 = none */
  int uniform_tiles;          /* uniform_spacing_flag (else explicit row heights) */
  int th_depth_inter, th_depth_intra;   /* max_transform_hierarchy_depth_* 0..2 */
  int qp_delta;               /* 0: cu_qp_delta off; 1..4: on with diff_cu_qp_delta_depth = value - 1 */
  int chroma_qp_offsets;      /* pps_cb/cr_qp_offset (+ slice offsets half of the time) */
  int deblock_mode;           /* 0 default, 1 disabled in the PPS, 2 PPS beta/tc offsets, 3 slice-level override */
  int par_mrg_level;          /* Log2ParMrgLevel 2..4 */
  int intra_in_p;             /* probability (%) of an intra CU in a P slice */
  int all_part_modes;         /* inter partitionings other than 2Nx2N (2NxN, Nx2N, and the AMP ones with amp) */
  int chroma_modes;           /* intra_chroma_pred_mode 0..3 besides 4 (derived) */
  int nxn_intra;              /* PART_NxN intra at 8x8 */
  int max_cu_log2, min_cu_log2;   /* coding block sizes used (the syntax allows 3..6 regardless) */
  int big_mvd;                /* 1 % of the vector differences are huge (reference blocks far outside the picture) */
  int slices;                 /* slice segments per picture, the two ways Kvazaar cuts them (uvgComm video/Slices, kvazaarfilter.cpp:205-215): 0 (also -1) one
                               * slice; 1 = a DEPENDENT slice segment per CTU row (kvazaar slices=wpp; here with or without WPP); 2 = an independent
                               * slice per tile (kvazaar slices=tiles; one slice when there are no tiles); 3 = FREE slices (one tile): segments that begin at
                               * any coding tree block, independent slices (own slice_qp_delta) and -- six streams in ten -- dependent segments mixed, with or
                               * without WPP: what an encoder that cuts slices by bytes or block counts sends */
  int tile_cols;              /* tile columns (uniform spacing), 1 (also -1) = none; with columns the slice forms are 0 and 2 */
  int tq_bypass;              /* probability (%) of cu_transquant_bypass_flag = 1; > 0 sets transquant_bypass_enabled_flag (-1 = 0: existing seeds keep their streams) */
  int scaling_lists;          /* 0 (also -1) scaling_list_enabled_flag = 0; 1 enabled with the default lists; 2 lists in the SPS; 3 default in the SPS, lists in the PPS; 4 both */
  int b_slices;               /* probability (%) that an inter picture is a B slice (both lists, bi-prediction, mvd_l1_zero_flag, collocated_from_l0_flag drawn per
                               * slice); 0 (also -1): P slices only -- what a Kvazaar peer sends with bipred=1 (kvazaarfilter.cpp:351-371) */
  int gop;                    /* 0 (also -1, 1): decoding order = output order; 2, 4 or 8: pictures come in groups of this size, the last one first and the ones in
                               * between in the order of a binary hierarchy (8 4 2 1 3 6 5 7), with references on both sides: output REORDERING
                               * (sps_max_num_reorder_pics = log2 of the size), what gop=8 makes Kvazaar write */
  int weighted;               /* probability (%) that a reference index of an inter slice gets explicit luma / chroma weights; > 0 sets weighted_pred_flag and
                               * weighted_bipred_flag (pred_weight_table() in every P / B slice header): what x265 writes by default (weightp) -- 0 (also -1): off */
  int list_mod;               /* probability (%) that a reference list of an inter slice is modified (ref_pic_lists_modification(): the entries of the initial list
                               * in any order, repeats allowed); > 0 sets lists_modification_present_flag -- 0 (also -1): off */
  int ctb_log2;               /* CtbLog2SizeY: 6 (also -1 / 0: the streams of rounds 1-5), 5 or 4 -- what encoders other than Kvazaar choose (hardware encoders: 32 or 16);
                               * max_cu_log2 is capped to it */
  int cip;                    /* 1: constrained_intra_pred_flag (samples of blocks that are not intra-coded are no reference samples) -- 0 (also -1): off */
  int long_term;              /* > 0: long-term reference pictures (P streams without reordering: the sequence's first picture becomes a long-term reference picture -- at once in a
                               * third of the cases, else when it leaves the short-term window -- and stays one until dropped; named through the SPS's candidates or explicitly,
                               * with and without delta_poc_msb_present_flag, used by the current picture or only kept) -- 0 (also -1): off */
  int pcm;                    /* probability (%) that an intra 2Nx2N coding unit of 8 .. 32 samples is a PCM unit (pcm_flag, raw samples at drawn bit depths, the arithmetic coder
                               * restarted behind them); > 0 sets pcm_enabled_flag, pcm_loop_filter_disabled_flag drawn -- 0 (also -1): off */
  int lf_across;              /* in-loop filtering across slice and tile boundaries: 0 (also -1: every stream of the earlier rounds) everywhere on; 1 = drawn -- half of the
                               * streams with tiles switch it off across tiles (loop_filter_across_tiles_enabled_flag = 0: what Kvazaar writes), and where the PPS allows
                               * it every slice draws its slice_loop_filter_across_slices_enabled_flag; 2 = everything off */
  int min_cb_log2;            /* MinCbLog2SizeY: 3 (also -1 / 0: every stream of the earlier rounds), 4 or 5 -- no coding unit below 16 / 32 samples; taken back to 3 when the
                               * picture size is no multiple of it or it exceeds the CTB.  At the minimum size above 8 an inter unit may be cut into four (PART_NxN) */
  int open_gop;               /* 1 (with gop > 1): groups that start with a CRA picture instead of following an IDR picture's lead -- their other pictures are RASL / RADL leading
                               * pictures, the parameter sets are repeated: a decoder may start there (RASL pictures dropped), a splicer may call the picture BLA -- 0 (also -1): off */
  int hidden_pics;            /* probability (%) of pic_output_flag = 0 (a picture that is decoded and referenced but never handed out); > 0 sets output_flag_present_flag -- 0 (also -1): off */
  int temporal_layers;        /* 1 (with gop > 1): the hierarchy's levels are temporal sub-layers -- TemporalId in the NAL unit headers, sps / vps_max_sub_layers_minus1 > 0 with drawn
                               * sub_layer_profile / level_present flags and ordering info for every sub-layer or the highest only, the top layer's pictures sub-layer non-reference
                               * pictures (TRAIL_N, RASL_N, RADL_N), no picture predicts from a higher sub-layer: what Kvazaar's gop=8 sends -- 0 (also -1): off */
  int vui_extras;             /* 1: the VUI's optional parts drawn (extended aspect ratio, overscan, video signal type with colour description, chroma sample location, default
                               * display window, POC-proportional timing, bitstream restriction -- what x265 writes by default), and the timing information in the VUI, in the
                               * VPS alone, in both (different rates: the VUI's counts) or nowhere -- 0 (also -1): timing in the VUI only */
  int rps_forms;              /* 1: every way a short-term reference picture set can be written (7.3.7) -- candidate sets in the SPS, explicit or predicted from the set before;
                               * a slice names a candidate (short_term_ref_pic_set_sps_flag) when its set equals one, else codes its set by inter RPS prediction from a
                               * candidate where that is possible (delta_idx_minus1, delta_rps, used_by_curr_pic_flag / use_delta_flag), else explicitly: what HM's
                               * configurations do -- 0 (also -1): always explicit in the slice header, as Kvazaar writes */
  int hdr_extras;             /* 1: what a decoder has to step over -- slice_reserved_flag bits (num_extra_slice_header_bits), slice segment header extension bytes, SPS / PPS
                               * extension data, access unit delimiters, prefix and suffix SEI messages a decoder does not know, filler data NAL units -- 0 (also -1): none */
} orc_gen_config;

void orc_gen_default_config(orc_gen_config *c);    /* everything random, 416x240 */
orc_gen *orc_gen_open(const orc_gen_config *c);
void orc_gen_get_config(const orc_gen *g, orc_gen_config *out);   /* with the drawn values filled in */
/* next access unit (parameter sets with every IDR); the buffer is owned by the generator and valid until the next call */
size_t orc_gen_picture(orc_gen *g, const uint8_t **au);
void orc_gen_close(orc_gen *g);
#ifdef __cplusplus
}
#endif
#endif
