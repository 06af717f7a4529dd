/* oracle/hevc_ps.h -- parameter sets and slice segment header: structs, writer, parser.
 * H.265 7.3.2.1 (VPS), 7.3.2.2 (SPS), 7.3.2.3 (PPS), 7.3.3 (profile_tier_level),
 * 7.3.6.1 (slice_segment_header), 7.3.7 (st_ref_pic_set), E.2.1 (VUI timing subset).
 * Test infrastructure. */
#ifndef ORC_HEVC_PS_H
#define ORC_HEVC_PS_H
#include "hevc_common.h"
#include "hevc_bits.h"
#include "hevc_scaling.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int num_negative, num_positive;
  int delta_poc_s0[16], used_s0[16];    /* negative pics: delta (negative numbers) */
  int delta_poc_s1[16], used_s1[16];
  /* writer only: the set coded by inter RPS prediction (7.3.7) -- from the set delta_idx_minus1 + 1 before it, moved by delta_rps, one used_by_curr_pic_flag /
   * use_delta_flag pair per entry of that set and one more for delta_rps itself.  The fields above hold what it derives to. */
  int inter, delta_idx_minus1, delta_rps, nflags;
  uint8_t used_flag[33], use_delta[33];
} orc_st_rps;

typedef struct {
  int valid;
  int vps_id, max_sub_layers, temporal_id_nesting;
  int timing_info_present; uint32_t num_units_in_tick, time_scale;
} orc_vps;

typedef struct {
  int valid;
  int sps_id, vps_id, max_sub_layers;
  int sl_ordering_absent;                  /* writer: sps/vps_sub_layer_ordering_info_present_flag = 0 (only the highest sub-layer's values are sent) */
  int sl_present;                          /* writer: bit 2i / 2i+1 = sub_layer_profile_present_flag[i] / sub_layer_level_present_flag[i] */
  int general_profile_idc, general_level_idc;
  int chroma_format_idc;
  int width, height;                       /* pic_width/height_in_luma_samples */
  int conf_win_flag, conf_left, conf_right, conf_top, conf_bottom;   /* in chroma units (x2 luma for 4:2:0) */
  int bit_depth_luma, bit_depth_chroma;
  int log2_max_poc_lsb;
  int max_dec_pic_buffering, max_num_reorder, max_latency_increase_plus1;
  int log2_min_cb, log2_diff_max_min_cb, log2_min_tb, log2_diff_max_min_tb;
  int max_th_depth_inter, max_th_depth_intra;
  int scaling_list_enabled, amp_enabled, sao_enabled, pcm_enabled;
  int scaling_list_data_present;                       /* sps_scaling_list_data_present_flag: else the default lists (Tables 7-5 / 7-6) */
  orc_scaling_lists scaling;                           /* the SPS's lists (the default ones without data) */
  uint8_t sl_pred_mode[4][6], sl_pred_delta[4][6];     /* writer only: how every list is coded (hevc_scaling.h orc_scaling_write) */
  int pcm_bit_depth_luma, pcm_bit_depth_chroma, log2_min_pcm_cb, log2_diff_max_min_pcm_cb, pcm_loop_filter_disabled;
  int num_st_rps; orc_st_rps st_rps[65];
  int long_term_ref_pics_present;
  int num_lt_sps; int lt_poc_lsb_sps[32]; uint8_t lt_used_sps[32];      /* num_long_term_ref_pics_sps candidates: lt_ref_pic_poc_lsb_sps, used_by_curr_pic_lt_sps_flag */
  int temporal_mvp_enabled, strong_intra_smoothing;
  int vui_present, vui_timing_present; uint32_t vui_num_units_in_tick, vui_time_scale;
  int ext_data;                            /* writer: as orc_pps.ext_data, for the SPS */
  int vui_extras;                          /* writer: bits 0..5 = aspect ratio (extended SAR), overscan, video signal type with colour description, chroma sample location,
                                            * default display window, bitstream restriction; bit 6 = vui_poc_proportional_to_timing_flag; bits 7..9 = the three single flags */
  /* derived */
  int ctb_log2, ctb_size, pic_w_ctbs, pic_h_ctbs, log2_max_tb;
} orc_sps;

typedef struct {
  int valid;
  int pps_id, sps_id;
  int dependent_slice_segments_enabled, output_flag_present, num_extra_slice_header_bits;
  int ext_data;                            /* writer: > 0 = pps_extension_present_flag with pps_extension_4bits set and this many bytes of extension data (to be ignored, 7.4.3.3.1) */
  int sign_data_hiding, cabac_init_present;
  int num_ref_idx_l0_default, num_ref_idx_l1_default;
  int init_qp;                              /* 26 + init_qp_minus26 */
  int constrained_intra_pred, transform_skip_enabled;
  int cu_qp_delta_enabled, diff_cu_qp_delta_depth;
  int cb_qp_offset, cr_qp_offset, slice_chroma_qp_offsets_present;
  int weighted_pred, weighted_bipred, transquant_bypass_enabled;
  int tiles_enabled, entropy_coding_sync_enabled;
  int num_tile_columns, num_tile_rows, uniform_spacing;
  int column_width[32], row_height[32];
  int loop_filter_across_tiles;
  int loop_filter_across_slices;
  int deblocking_filter_control_present, deblocking_filter_override_enabled;
  int pps_deblocking_disabled, pps_beta_offset_div2, pps_tc_offset_div2;
  int scaling_list_data_present, lists_modification_present;
  orc_scaling_lists scaling;                           /* pps_scaling_list_data: overrides the SPS's lists */
  uint8_t sl_pred_mode[4][6], sl_pred_delta[4][6];     /* writer only */
  int log2_parallel_merge_level;
  int slice_header_extension_present;
} orc_pps;

typedef struct {
  int first_slice_segment_in_pic, no_output_of_prior_pics;
  int pps_id, dependent_slice_segment, slice_segment_address;
  int slice_type, pic_output_flag;
  int poc_lsb;
  int short_term_ref_pic_set_sps_flag, short_term_rps_idx;
  orc_st_rps st_rps;                       /* active RPS (copied from SPS or parsed) */
  /* long-term reference pictures (7.3.6.1; SPS long_term_ref_pics_present_flag): num_lt entries = num_long_term_sps candidates of the SPS (lt_idx_sps) followed by
   * num_long_term_pics explicit ones (poc_lsb_lt, used_by_curr_pic_lt_flag), each with delta_poc_msb_present_flag [+ delta_poc_msb_cycle_lt].  The parser resolves
   * them: lt_poc_lsb / lt_used (from the SPS for the first kind), lt_msb_cycle = DeltaPocMsbCycleLt (7-52: the cycles accumulate inside each of the two groups) */
  int num_long_term_sps, num_long_term_pics, num_lt;
  int lt_idx_sps[16], lt_poc_lsb[16], lt_msb_cycle[16]; uint8_t lt_used[16], lt_msb_present[16];
  int lt_msb_cycle_delta[16];                /* writer: delta_poc_msb_cycle_lt as coded */
  int slice_temporal_mvp_enabled;
  int sao_luma, sao_chroma;
  int num_ref_idx_l0, num_ref_idx_l1;
  int mvd_l1_zero, cabac_init_flag, collocated_from_l0, collocated_ref_idx;
  int max_num_merge_cand;
  int slice_qp_delta, slice_cb_qp_offset, slice_cr_qp_offset;
  int deblocking_filter_override, slice_deblocking_disabled, beta_offset_div2, tc_offset_div2;
  int loop_filter_across_slices;
  int num_entry_points; uint32_t *entry_point_offset;   /* offset_minus1 + 1, malloc'ed by parser */
  int slice_qp;
  /* ref_pic_lists_modification() (7.3.6.2; PPS lists_modification_present_flag, more than one picture in the reference picture set): entries of the
   * temporary list (8.3.4) in the order the slice wants them */
  int rpl_mod_flag[2]; uint8_t list_entry[2][16];
  /* pred_weight_table() (7.3.6.3), present when the PPS says weighted_pred_flag (P slices) / weighted_bipred_flag (B slices): the syntax elements ... */
  int weighted;                                          /* the table is present: explicit weighted sample prediction for every block of the slice (8.5.3.3.4.3) */
  int luma_log2_weight_denom, delta_chroma_log2_weight_denom;
  uint8_t luma_weight_flag[2][16], chroma_weight_flag[2][16];
  int16_t delta_luma_weight[2][16], luma_offset[2][16], delta_chroma_weight[2][16][2], delta_chroma_offset[2][16][2];
  /* ... and what 7.4.7.3 derives from them: weights and offsets per list, reference index and component (0 luma, 1 Cb, 2 Cr) */
  int wp_log2wd[2];                                      /* luma / chroma denominators (without the 14 - bitDepth of the formula) */
  int16_t wp_w[2][16][3], wp_o[2][16][3];
} orc_slice_hdr;
void orc_derive_pred_weights(orc_slice_hdr *h);        /* wp_* from the syntax elements */

void orc_write_vps(orc_bitw *w, const orc_vps *v, const orc_sps *s);
void orc_write_sps(orc_bitw *w, const orc_sps *s);
void orc_write_pps(orc_bitw *w, const orc_pps *p);
/* writes header incl. entry points and byte_alignment() */
void orc_write_slice_header(orc_bitw *w, const orc_slice_hdr *h, const orc_sps *s, const orc_pps *p, int nal_type);

/* parsers return 0 on success, <0 on unsupported/invalid */
int orc_parse_vps(orc_bitr *r, orc_vps *v);
int orc_parse_sps(orc_bitr *r, orc_sps *s);
int orc_parse_pps(orc_bitr *r, orc_pps *p);
/* sps_tab/pps_tab: arrays indexed by id (16 / 64 entries) */
int orc_parse_slice_header(orc_bitr *r, orc_slice_hdr *h, int nal_type, const orc_sps *sps_tab, const orc_pps *pps_tab);
void orc_sps_derive(orc_sps *s);
#ifdef __cplusplus
}
#endif
#endif
