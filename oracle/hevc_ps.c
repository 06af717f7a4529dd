/* oracle/hevc_ps.c -- see hevc_ps.h.  Test infrastructure. */
#include "hevc_ps.h"

/* ------------------------------------------------------------------ writer */
static void write_ptl_general(orc_bitw *w, int profile_idc)
{
  orc_bw_put(w, 0, 2);                 /* profile_space */
  orc_bw_put(w, 0, 1);                 /* tier_flag */
  orc_bw_put(w, (uint32_t)profile_idc, 5);
  for (int j = 0; j < 32; j++) orc_bw_put(w, (j == profile_idc || (profile_idc == 1 && j == 2)) ? 1 : 0, 1);
  orc_bw_put(w, 1, 1);                 /* progressive_source_flag */
  orc_bw_put(w, 0, 1);                 /* interlaced_source_flag */
  orc_bw_put(w, 0, 1);                 /* non_packed_constraint_flag */
  orc_bw_put(w, 1, 1);                 /* frame_only_constraint_flag */
  orc_bw_put(w, 0, 32); orc_bw_put(w, 0, 11);   /* 43 reserved zero bits */
  orc_bw_put(w, 0, 1);                 /* inbld_flag / reserved */
}
static void write_ptl(orc_bitw *w, int profile_idc, int level_idc, int max_sub_layers_minus1, int sl_present)
{
  /* 7.3.3 profile_tier_level(1, maxNumSubLayersMinus1) */
  write_ptl_general(w, profile_idc);
  orc_bw_put(w, (uint32_t)level_idc, 8);
  for (int i = 0; i < max_sub_layers_minus1; i++) orc_bw_put(w, (uint32_t)(sl_present >> (2 * i)) & 3u, 2);      /* sub_layer_profile_present_flag, sub_layer_level_present_flag */
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) orc_bw_put(w, 0, 2);
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    if ((sl_present >> (2 * i)) & 2) write_ptl_general(w, profile_idc);
    if ((sl_present >> (2 * i)) & 1) orc_bw_put(w, (uint32_t)level_idc, 8);
  }
}
/* sub_layer_ordering_info: every sub-layer's values (non-decreasing, the highest one's are the stream's) or the highest one's only */
static void write_ordering_info(orc_bitw *w, const orc_sps *s)
{
  const int n = s->max_sub_layers > 1 ? s->max_sub_layers : 1;
  orc_bw_put(w, s->sl_ordering_absent ? 0 : 1, 1);
  for (int i = s->sl_ordering_absent ? n - 1 : 0; i < n; i++) {
    const int less = n - 1 - i;
    orc_bw_ue(w, (uint32_t)(s->max_dec_pic_buffering - less > 1 ? s->max_dec_pic_buffering - less : 1) - 1);
    orc_bw_ue(w, (uint32_t)(s->max_num_reorder - less > 0 ? s->max_num_reorder - less : 0));
    orc_bw_ue(w, (uint32_t)s->max_latency_increase_plus1);
  }
}

void orc_write_vps(orc_bitw *w, const orc_vps *v, const orc_sps *s)
{
  orc_bw_put(w, (uint32_t)v->vps_id, 4);
  orc_bw_put(w, 3, 2);                 /* vps_base_layer_internal_flag, vps_base_layer_available_flag */
  orc_bw_put(w, 0, 6);                 /* vps_max_layers_minus1 */
  const int msl = s->max_sub_layers > 1 ? s->max_sub_layers - 1 : 0;
  orc_bw_put(w, (uint32_t)msl, 3);     /* vps_max_sub_layers_minus1 */
  orc_bw_put(w, msl ? 0 : 1, 1);       /* vps_temporal_id_nesting_flag */
  orc_bw_put(w, 0xffff, 16);
  write_ptl(w, s->general_profile_idc, s->general_level_idc, msl, s->sl_present);
  write_ordering_info(w, s);           /* vps_sub_layer_ordering_info_present_flag ... */
  orc_bw_put(w, 0, 6);                 /* vps_max_layer_id */
  orc_bw_ue(w, 0);                     /* vps_num_layer_sets_minus1 */
  orc_bw_put(w, (uint32_t)v->timing_info_present, 1);
  if (v->timing_info_present) {
    orc_bw_put(w, v->num_units_in_tick, 32);
    orc_bw_put(w, v->time_scale, 32);
    orc_bw_put(w, 0, 1);               /* vps_poc_proportional_to_timing_flag */
    orc_bw_ue(w, 0);                   /* vps_num_hrd_parameters */
  }
  orc_bw_put(w, 0, 1);                 /* vps_extension_flag */
  orc_bw_trailing(w);
}

static void write_st_rps(orc_bitw *w, const orc_st_rps *r, int idx, int num_in_sps)
{
  if (idx != 0) orc_bw_put(w, r->inter ? 1 : 0, 1);   /* inter_ref_pic_set_prediction_flag */
  if (idx != 0 && r->inter) {
    if (idx == num_in_sps) orc_bw_ue(w, (uint32_t)r->delta_idx_minus1);
    orc_bw_put(w, r->delta_rps < 0, 1);
    orc_bw_ue(w, (uint32_t)(r->delta_rps < 0 ? -r->delta_rps : r->delta_rps) - 1);
    for (int j = 0; j < r->nflags; j++) { orc_bw_put(w, r->used_flag[j], 1); if (!r->used_flag[j]) orc_bw_put(w, r->use_delta[j], 1); }
    return;
  }
  orc_bw_ue(w, (uint32_t)r->num_negative);
  orc_bw_ue(w, (uint32_t)r->num_positive);
  int prev = 0;
  for (int i = 0; i < r->num_negative; i++) {
    orc_bw_ue(w, (uint32_t)(prev - r->delta_poc_s0[i] - 1));
    orc_bw_put(w, (uint32_t)r->used_s0[i], 1);
    prev = r->delta_poc_s0[i];
  }
  prev = 0;
  for (int i = 0; i < r->num_positive; i++) {
    orc_bw_ue(w, (uint32_t)(r->delta_poc_s1[i] - prev - 1));
    orc_bw_put(w, (uint32_t)r->used_s1[i], 1);
    prev = r->delta_poc_s1[i];
  }
}

void orc_write_sps(orc_bitw *w, const orc_sps *s)
{
  orc_bw_put(w, (uint32_t)s->vps_id, 4);
  const int msl = s->max_sub_layers > 1 ? s->max_sub_layers - 1 : 0;
  orc_bw_put(w, (uint32_t)msl, 3);     /* sps_max_sub_layers_minus1 */
  orc_bw_put(w, msl ? 0 : 1, 1);       /* sps_temporal_id_nesting_flag */
  write_ptl(w, s->general_profile_idc, s->general_level_idc, msl, s->sl_present);
  orc_bw_ue(w, (uint32_t)s->sps_id);
  orc_bw_ue(w, (uint32_t)s->chroma_format_idc);
  orc_bw_ue(w, (uint32_t)s->width);
  orc_bw_ue(w, (uint32_t)s->height);
  orc_bw_put(w, (uint32_t)s->conf_win_flag, 1);
  if (s->conf_win_flag) {
    orc_bw_ue(w, (uint32_t)s->conf_left); orc_bw_ue(w, (uint32_t)s->conf_right);
    orc_bw_ue(w, (uint32_t)s->conf_top);  orc_bw_ue(w, (uint32_t)s->conf_bottom);
  }
  orc_bw_ue(w, (uint32_t)s->bit_depth_luma - 8);
  orc_bw_ue(w, (uint32_t)s->bit_depth_chroma - 8);
  orc_bw_ue(w, (uint32_t)s->log2_max_poc_lsb - 4);
  write_ordering_info(w, s);           /* sps_sub_layer_ordering_info_present_flag ... */
  orc_bw_ue(w, (uint32_t)s->log2_min_cb - 3);
  orc_bw_ue(w, (uint32_t)s->log2_diff_max_min_cb);
  orc_bw_ue(w, (uint32_t)s->log2_min_tb - 2);
  orc_bw_ue(w, (uint32_t)s->log2_diff_max_min_tb);
  orc_bw_ue(w, (uint32_t)s->max_th_depth_inter);
  orc_bw_ue(w, (uint32_t)s->max_th_depth_intra);
  orc_bw_put(w, (uint32_t)s->scaling_list_enabled, 1);
  if (s->scaling_list_enabled) {
    orc_bw_put(w, (uint32_t)s->scaling_list_data_present, 1);
    if (s->scaling_list_data_present) orc_scaling_write(w, &s->scaling, s->sl_pred_mode, s->sl_pred_delta);
  }
  orc_bw_put(w, (uint32_t)s->amp_enabled, 1);
  orc_bw_put(w, (uint32_t)s->sao_enabled, 1);
  orc_bw_put(w, (uint32_t)s->pcm_enabled, 1);
  if (s->pcm_enabled) {
    orc_bw_put(w, (uint32_t)s->pcm_bit_depth_luma - 1, 4); orc_bw_put(w, (uint32_t)s->pcm_bit_depth_chroma - 1, 4);
    orc_bw_ue(w, (uint32_t)s->log2_min_pcm_cb - 3); orc_bw_ue(w, (uint32_t)s->log2_diff_max_min_pcm_cb);
    orc_bw_put(w, (uint32_t)s->pcm_loop_filter_disabled, 1);
  }
  orc_bw_ue(w, (uint32_t)s->num_st_rps);
  for (int i = 0; i < s->num_st_rps; i++) write_st_rps(w, &s->st_rps[i], i, s->num_st_rps);
  orc_bw_put(w, (uint32_t)s->long_term_ref_pics_present, 1);
  if (s->long_term_ref_pics_present) {
    orc_bw_ue(w, (uint32_t)s->num_lt_sps);
    for (int i = 0; i < s->num_lt_sps; i++) { orc_bw_put(w, (uint32_t)s->lt_poc_lsb_sps[i], s->log2_max_poc_lsb); orc_bw_put(w, s->lt_used_sps[i], 1); }
  }
  orc_bw_put(w, (uint32_t)s->temporal_mvp_enabled, 1);
  orc_bw_put(w, (uint32_t)s->strong_intra_smoothing, 1);
  orc_bw_put(w, (uint32_t)s->vui_present, 1);
  if (s->vui_present) {
    /* E.2.1 vui_parameters(): the timing info is what a decoder here uses; the optional parts (vui_extras) are there to be skipped correctly */
    const int x = s->vui_extras;
    orc_bw_put(w, x & 1, 1);           /* aspect_ratio_info_present_flag */
    if (x & 1) { orc_bw_put(w, 255, 8); orc_bw_put(w, 40, 16); orc_bw_put(w, 33, 16); }      /* EXTENDED_SAR */
    orc_bw_put(w, (x >> 1) & 1, 1);    /* overscan_info_present_flag */
    if (x & 2) orc_bw_put(w, 1, 1);
    orc_bw_put(w, (x >> 2) & 1, 1);    /* video_signal_type_present_flag */
    if (x & 4) { orc_bw_put(w, 5, 3); orc_bw_put(w, 1, 1); orc_bw_put(w, 1, 1); orc_bw_put(w, 1, 8); orc_bw_put(w, 13, 8); orc_bw_put(w, 1, 8); }
    orc_bw_put(w, (x >> 3) & 1, 1);    /* chroma_loc_info_present_flag */
    if (x & 8) { orc_bw_ue(w, 2); orc_bw_ue(w, 3); }
    orc_bw_put(w, (x >> 7) & 1, 1);    /* neutral_chroma_indication_flag */
    orc_bw_put(w, 0, 1);               /* field_seq_flag */
    orc_bw_put(w, (x >> 9) & 1, 1);    /* frame_field_info_present_flag */
    orc_bw_put(w, (x >> 4) & 1, 1);    /* default_display_window_flag */
    if (x & 16) { orc_bw_ue(w, 1); orc_bw_ue(w, 2); orc_bw_ue(w, 0); orc_bw_ue(w, 3); }
    orc_bw_put(w, (uint32_t)s->vui_timing_present, 1);
    if (s->vui_timing_present) {
      orc_bw_put(w, s->vui_num_units_in_tick, 32);
      orc_bw_put(w, s->vui_time_scale, 32);
      orc_bw_put(w, (x >> 6) & 1, 1);  /* vui_poc_proportional_to_timing_flag */
      if (x & 64) orc_bw_ue(w, 0);     /* vui_num_ticks_poc_diff_one_minus1 */
      orc_bw_put(w, 0, 1);             /* vui_hrd_parameters_present_flag */
    }
    orc_bw_put(w, (x >> 5) & 1, 1);    /* bitstream_restriction_flag */
    if (x & 32) { orc_bw_put(w, 0, 1); orc_bw_put(w, 1, 1); orc_bw_put(w, 0, 1); orc_bw_ue(w, 0); orc_bw_ue(w, 2); orc_bw_ue(w, 1); orc_bw_ue(w, 15); orc_bw_ue(w, 15); }
  }
  orc_bw_put(w, s->ext_data > 0, 1);   /* sps_extension_present_flag */
  if (s->ext_data > 0) {               /* range, multilayer, 3D extension flags 0; sps_extension_4bits set: sps_extension_data_flag to the end -- "decoders shall ignore" */
    orc_bw_put(w, 0, 3); orc_bw_put(w, 1, 1); orc_bw_put(w, 5, 4);
    for (int i = 0; i < s->ext_data; i++) orc_bw_put(w, (uint32_t)(0xA5 + 37 * i) & 0xff, 8);
  }
  orc_bw_trailing(w);
}

void orc_write_pps(orc_bitw *w, const orc_pps *p)
{
  orc_bw_ue(w, (uint32_t)p->pps_id);
  orc_bw_ue(w, (uint32_t)p->sps_id);
  orc_bw_put(w, (uint32_t)p->dependent_slice_segments_enabled, 1);
  orc_bw_put(w, (uint32_t)p->output_flag_present, 1);
  orc_bw_put(w, (uint32_t)p->num_extra_slice_header_bits, 3);
  orc_bw_put(w, (uint32_t)p->sign_data_hiding, 1);
  orc_bw_put(w, (uint32_t)p->cabac_init_present, 1);
  orc_bw_ue(w, (uint32_t)p->num_ref_idx_l0_default - 1);
  orc_bw_ue(w, (uint32_t)p->num_ref_idx_l1_default - 1);
  orc_bw_se(w, p->init_qp - 26);
  orc_bw_put(w, (uint32_t)p->constrained_intra_pred, 1);
  orc_bw_put(w, (uint32_t)p->transform_skip_enabled, 1);
  orc_bw_put(w, (uint32_t)p->cu_qp_delta_enabled, 1);
  if (p->cu_qp_delta_enabled) orc_bw_ue(w, (uint32_t)p->diff_cu_qp_delta_depth);
  orc_bw_se(w, p->cb_qp_offset);
  orc_bw_se(w, p->cr_qp_offset);
  orc_bw_put(w, (uint32_t)p->slice_chroma_qp_offsets_present, 1);
  orc_bw_put(w, (uint32_t)p->weighted_pred, 1);
  orc_bw_put(w, (uint32_t)p->weighted_bipred, 1);
  orc_bw_put(w, (uint32_t)p->transquant_bypass_enabled, 1);
  orc_bw_put(w, (uint32_t)p->tiles_enabled, 1);
  orc_bw_put(w, (uint32_t)p->entropy_coding_sync_enabled, 1);
  if (p->tiles_enabled) {
    orc_bw_ue(w, (uint32_t)p->num_tile_columns - 1);
    orc_bw_ue(w, (uint32_t)p->num_tile_rows - 1);
    orc_bw_put(w, (uint32_t)p->uniform_spacing, 1);
    if (!p->uniform_spacing) {
      for (int i = 0; i < p->num_tile_columns - 1; i++) orc_bw_ue(w, (uint32_t)p->column_width[i] - 1);
      for (int i = 0; i < p->num_tile_rows - 1; i++) orc_bw_ue(w, (uint32_t)p->row_height[i] - 1);
    }
    orc_bw_put(w, (uint32_t)p->loop_filter_across_tiles, 1);
  }
  orc_bw_put(w, (uint32_t)p->loop_filter_across_slices, 1);
  orc_bw_put(w, (uint32_t)p->deblocking_filter_control_present, 1);
  if (p->deblocking_filter_control_present) {
    orc_bw_put(w, (uint32_t)p->deblocking_filter_override_enabled, 1);
    orc_bw_put(w, (uint32_t)p->pps_deblocking_disabled, 1);
    if (!p->pps_deblocking_disabled) { orc_bw_se(w, p->pps_beta_offset_div2); orc_bw_se(w, p->pps_tc_offset_div2); }
  }
  orc_bw_put(w, (uint32_t)p->scaling_list_data_present, 1);
  if (p->scaling_list_data_present) orc_scaling_write(w, &p->scaling, p->sl_pred_mode, p->sl_pred_delta);
  orc_bw_put(w, (uint32_t)p->lists_modification_present, 1);
  orc_bw_ue(w, (uint32_t)p->log2_parallel_merge_level - 2);
  orc_bw_put(w, p->slice_header_extension_present != 0, 1);
  orc_bw_put(w, p->ext_data > 0, 1);   /* pps_extension_present_flag */
  if (p->ext_data > 0) {
    orc_bw_put(w, 0, 3); orc_bw_put(w, 1, 1); orc_bw_put(w, 9, 4);
    for (int i = 0; i < p->ext_data; i++) orc_bw_put(w, (uint32_t)(0x3C + 91 * i) & 0xff, 8);
  }
  orc_bw_trailing(w);
}

static int ceil_log2(unsigned v) { int n = 0; while ((1u << n) < v) n++; return n; }

void orc_write_slice_header(orc_bitw *w, const orc_slice_hdr *h, const orc_sps *s, const orc_pps *p, int nal_type)
{
  orc_bw_put(w, (uint32_t)h->first_slice_segment_in_pic, 1);
  if (nal_type >= NAL_BLA_W_LP && nal_type <= NAL_RSV_IRAP_VCL23) orc_bw_put(w, (uint32_t)h->no_output_of_prior_pics, 1);
  orc_bw_ue(w, (uint32_t)h->pps_id);
  if (!h->first_slice_segment_in_pic) {
    if (p->dependent_slice_segments_enabled) orc_bw_put(w, (uint32_t)h->dependent_slice_segment, 1);
    orc_bw_put(w, (uint32_t)h->slice_segment_address, ceil_log2((unsigned)(s->pic_w_ctbs * s->pic_h_ctbs)));
  }
  if (!h->dependent_slice_segment) {
    for (int i = 0; i < p->num_extra_slice_header_bits; i++) orc_bw_put(w, (uint32_t)(h->slice_qp_delta + i) & 1, 1);      /* slice_reserved_flag[]: any value */
    orc_bw_ue(w, (uint32_t)h->slice_type);
    if (p->output_flag_present) orc_bw_put(w, (uint32_t)h->pic_output_flag, 1);
    if (nal_type != NAL_IDR_W_RADL && nal_type != NAL_IDR_N_LP) {
      orc_bw_put(w, (uint32_t)h->poc_lsb, s->log2_max_poc_lsb);
      orc_bw_put(w, (uint32_t)h->short_term_ref_pic_set_sps_flag, 1);
      if (!h->short_term_ref_pic_set_sps_flag) write_st_rps(w, &h->st_rps, s->num_st_rps, s->num_st_rps);
      else if (s->num_st_rps > 1) orc_bw_put(w, (uint32_t)h->short_term_rps_idx, ceil_log2((unsigned)s->num_st_rps));
      if (s->long_term_ref_pics_present) {
        if (s->num_lt_sps > 0) orc_bw_ue(w, (uint32_t)h->num_long_term_sps);
        orc_bw_ue(w, (uint32_t)h->num_long_term_pics);
        for (int i = 0; i < h->num_long_term_sps + h->num_long_term_pics; i++) {
          if (i < h->num_long_term_sps) { if (s->num_lt_sps > 1) orc_bw_put(w, (uint32_t)h->lt_idx_sps[i], ceil_log2((unsigned)s->num_lt_sps)); }
          else { orc_bw_put(w, (uint32_t)h->lt_poc_lsb[i], s->log2_max_poc_lsb); orc_bw_put(w, h->lt_used[i], 1); }
          orc_bw_put(w, h->lt_msb_present[i], 1);
          if (h->lt_msb_present[i]) orc_bw_ue(w, (uint32_t)h->lt_msb_cycle_delta[i]);
        }
      }
      if (s->temporal_mvp_enabled) orc_bw_put(w, (uint32_t)h->slice_temporal_mvp_enabled, 1);
    }
    if (s->sao_enabled) { orc_bw_put(w, (uint32_t)h->sao_luma, 1); orc_bw_put(w, (uint32_t)h->sao_chroma, 1); }
    if (h->slice_type != SLICE_I) {
      int override = (h->num_ref_idx_l0 != p->num_ref_idx_l0_default) ||
                     (h->slice_type == SLICE_B && h->num_ref_idx_l1 != p->num_ref_idx_l1_default);
      orc_bw_put(w, (uint32_t)override, 1);
      if (override) {
        orc_bw_ue(w, (uint32_t)h->num_ref_idx_l0 - 1);
        if (h->slice_type == SLICE_B) orc_bw_ue(w, (uint32_t)h->num_ref_idx_l1 - 1);
      }
      {
        int num_pic_total = 0;
        for (int i = 0; i < h->st_rps.num_negative; i++) num_pic_total += h->st_rps.used_s0[i];
        for (int i = 0; i < h->st_rps.num_positive; i++) num_pic_total += h->st_rps.used_s1[i];
        for (int i = 0; i < h->num_long_term_sps + h->num_long_term_pics; i++) num_pic_total += h->lt_used[i];
        if (p->lists_modification_present && num_pic_total > 1) {
          const int bits = ceil_log2((unsigned)num_pic_total);
          for (int l = 0; l < (h->slice_type == SLICE_B ? 2 : 1); l++) {
            orc_bw_put(w, (uint32_t)h->rpl_mod_flag[l], 1);
            if (h->rpl_mod_flag[l]) for (int i = 0; i < (l ? h->num_ref_idx_l1 : h->num_ref_idx_l0); i++) orc_bw_put(w, h->list_entry[l][i], bits);
          }
        }
      }
      if (h->slice_type == SLICE_B) orc_bw_put(w, (uint32_t)h->mvd_l1_zero, 1);
      if (p->cabac_init_present) orc_bw_put(w, (uint32_t)h->cabac_init_flag, 1);
      if (h->slice_temporal_mvp_enabled) {
        if (h->slice_type == SLICE_B) orc_bw_put(w, (uint32_t)h->collocated_from_l0, 1);
        if ((h->collocated_from_l0 && h->num_ref_idx_l0 > 1) || (!h->collocated_from_l0 && h->num_ref_idx_l1 > 1))
          orc_bw_ue(w, (uint32_t)h->collocated_ref_idx);
      }
      if ((p->weighted_pred && h->slice_type == SLICE_P) || (p->weighted_bipred && h->slice_type == SLICE_B)) {
        /* pred_weight_table(): the flags of a list first, then the values of the entries whose flags are set (single layer: every entry's POC differs from the picture's) */
        orc_bw_ue(w, (uint32_t)h->luma_log2_weight_denom);
        orc_bw_se(w, h->delta_chroma_log2_weight_denom);
        for (int l = 0; l < (h->slice_type == SLICE_B ? 2 : 1); l++) {
          const int n = l ? h->num_ref_idx_l1 : h->num_ref_idx_l0;
          for (int i = 0; i < n; i++) orc_bw_put(w, h->luma_weight_flag[l][i], 1);
          for (int i = 0; i < n; i++) orc_bw_put(w, h->chroma_weight_flag[l][i], 1);
          for (int i = 0; i < n; i++) {
            if (h->luma_weight_flag[l][i]) { orc_bw_se(w, h->delta_luma_weight[l][i]); orc_bw_se(w, h->luma_offset[l][i]); }
            if (h->chroma_weight_flag[l][i]) for (int j = 0; j < 2; j++) { orc_bw_se(w, h->delta_chroma_weight[l][i][j]); orc_bw_se(w, h->delta_chroma_offset[l][i][j]); }
          }
        }
      }
      orc_bw_ue(w, (uint32_t)(5 - h->max_num_merge_cand));
    }
    orc_bw_se(w, h->slice_qp_delta);
    if (p->slice_chroma_qp_offsets_present) { orc_bw_se(w, h->slice_cb_qp_offset); orc_bw_se(w, h->slice_cr_qp_offset); }
    if (p->deblocking_filter_override_enabled) orc_bw_put(w, (uint32_t)h->deblocking_filter_override, 1);
    if (h->deblocking_filter_override) {
      orc_bw_put(w, (uint32_t)h->slice_deblocking_disabled, 1);
      if (!h->slice_deblocking_disabled) { orc_bw_se(w, h->beta_offset_div2); orc_bw_se(w, h->tc_offset_div2); }
    }
    if (p->loop_filter_across_slices && (h->sao_luma || h->sao_chroma || !h->slice_deblocking_disabled))
      orc_bw_put(w, (uint32_t)h->loop_filter_across_slices, 1);
  }
  if (p->tiles_enabled || p->entropy_coding_sync_enabled) {
    orc_bw_ue(w, (uint32_t)h->num_entry_points);
    if (h->num_entry_points > 0) {
      uint32_t mx = 0;
      for (int i = 0; i < h->num_entry_points; i++) if (h->entry_point_offset[i] - 1 > mx) mx = h->entry_point_offset[i] - 1;
      int len = 1; while (len < 32 && (mx >> len)) len++;
      orc_bw_ue(w, (uint32_t)len - 1);
      for (int i = 0; i < h->num_entry_points; i++) orc_bw_put(w, h->entry_point_offset[i] - 1, len);
    }
  }
  if (p->slice_header_extension_present) {     /* slice_segment_header_extension_length and that many bytes of any value */
    const int n = p->slice_header_extension_present - 1;
    orc_bw_ue(w, (uint32_t)n);
    for (int i = 0; i < n; i++) orc_bw_put(w, (uint32_t)(h->slice_segment_address * 7 + 0x11 * i + 1) & 0xff, 8);
  }
  orc_bw_trailing(w);   /* byte_alignment() */
}

/* ------------------------------------------------------------------ parser */
static int parse_ptl(orc_bitr *r, int max_sub_layers_minus1, int *profile_idc, int *level_idc)
{
  orc_br_get(r, 2); orc_br_get(r, 1);
  *profile_idc = (int)orc_br_get(r, 5);
  orc_br_get(r, 32);
  orc_br_get(r, 4);
  orc_br_get(r, 32); orc_br_get(r, 11); orc_br_get(r, 1);
  *level_idc = (int)orc_br_get(r, 8);
  int pp[8], lp[8];
  for (int i = 0; i < max_sub_layers_minus1; i++) { pp[i] = (int)orc_br_get(r, 1); lp[i] = (int)orc_br_get(r, 1); }
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) orc_br_get(r, 2);
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    if (pp[i]) { orc_br_get(r, 32); orc_br_get(r, 32); orc_br_get(r, 24); }
    if (lp[i]) orc_br_get(r, 8);
  }
  return r->error ? -1 : 0;
}

int orc_parse_vps(orc_bitr *r, orc_vps *v)
{
  memset(v, 0, sizeof(*v));
  v->vps_id = (int)orc_br_get(r, 4);
  orc_br_get(r, 2); orc_br_get(r, 6);
  v->max_sub_layers = (int)orc_br_get(r, 3) + 1;
  v->temporal_id_nesting = (int)orc_br_get(r, 1);
  orc_br_get(r, 16);
  int prof, lvl;
  if (parse_ptl(r, v->max_sub_layers - 1, &prof, &lvl)) return -1;
  int oi = (int)orc_br_get(r, 1);
  for (int i = oi ? 0 : v->max_sub_layers - 1; i < v->max_sub_layers; i++) { orc_br_ue(r); orc_br_ue(r); orc_br_ue(r); }
  int max_layer_id = (int)orc_br_get(r, 6);
  int num_layer_sets = (int)orc_br_ue(r) + 1;
  for (int i = 1; i < num_layer_sets; i++) for (int j = 0; j <= max_layer_id; j++) orc_br_get(r, 1);
  v->timing_info_present = (int)orc_br_get(r, 1);
  if (v->timing_info_present) {
    v->num_units_in_tick = orc_br_get(r, 32);
    v->time_scale = orc_br_get(r, 32);
    /* remaining VPS fields (poc proportional, hrd) are not needed */
  }
  if (r->error) return -1;
  v->valid = 1;
  return 0;
}

static int parse_st_rps(orc_bitr *r, orc_st_rps *out, int idx, int num_in_sps, const orc_st_rps *all)
{
  int inter = 0;
  memset(out, 0, sizeof(*out));
  if (idx != 0) inter = (int)orc_br_get(r, 1);
  if (inter) {
    int delta_idx = 1;
    if (idx == num_in_sps) delta_idx = (int)orc_br_ue(r) + 1;
    if (delta_idx > idx) return -1;
    const orc_st_rps *ref = &all[idx - delta_idx];
    int sign = (int)orc_br_get(r, 1);
    int absd = (int)orc_br_ue(r) + 1;
    int drps = (1 - 2 * sign) * absd;
    int nd = ref->num_negative + ref->num_positive;
    int used[33], use_delta[33];
    for (int j = 0; j <= nd; j++) {
      used[j] = (int)orc_br_get(r, 1);
      use_delta[j] = 1;
      if (!used[j]) use_delta[j] = (int)orc_br_get(r, 1);
    }
    int i = 0;
    for (int j = ref->num_positive - 1; j >= 0; j--) {
      int d = ref->delta_poc_s1[j] + drps;
      if (d < 0 && use_delta[ref->num_negative + j]) { out->delta_poc_s0[i] = d; out->used_s0[i++] = used[ref->num_negative + j]; }
    }
    if (drps < 0 && use_delta[nd]) { out->delta_poc_s0[i] = drps; out->used_s0[i++] = used[nd]; }
    for (int j = 0; j < ref->num_negative; j++) {
      int d = ref->delta_poc_s0[j] + drps;
      if (d < 0 && use_delta[j]) { out->delta_poc_s0[i] = d; out->used_s0[i++] = used[j]; }
    }
    out->num_negative = i;
    i = 0;
    for (int j = ref->num_negative - 1; j >= 0; j--) {
      int d = ref->delta_poc_s0[j] + drps;
      if (d > 0 && use_delta[j]) { out->delta_poc_s1[i] = d; out->used_s1[i++] = used[j]; }
    }
    if (drps > 0 && use_delta[nd]) { out->delta_poc_s1[i] = drps; out->used_s1[i++] = used[nd]; }
    for (int j = 0; j < ref->num_positive; j++) {
      int d = ref->delta_poc_s1[j] + drps;
      if (d > 0 && use_delta[ref->num_negative + j]) { out->delta_poc_s1[i] = d; out->used_s1[i++] = used[ref->num_negative + j]; }
    }
    out->num_positive = i;
  } else {
    out->num_negative = (int)orc_br_ue(r);
    out->num_positive = (int)orc_br_ue(r);
    if (out->num_negative > 16 || out->num_positive > 16) return -1;
    int prev = 0;
    for (int i = 0; i < out->num_negative; i++) {
      prev -= (int)orc_br_ue(r) + 1;
      out->delta_poc_s0[i] = prev; out->used_s0[i] = (int)orc_br_get(r, 1);
    }
    prev = 0;
    for (int i = 0; i < out->num_positive; i++) {
      prev += (int)orc_br_ue(r) + 1;
      out->delta_poc_s1[i] = prev; out->used_s1[i] = (int)orc_br_get(r, 1);
    }
  }
  return r->error ? -1 : 0;
}

void orc_sps_derive(orc_sps *s)
{
  s->ctb_log2 = s->log2_min_cb + s->log2_diff_max_min_cb;
  s->ctb_size = 1 << s->ctb_log2;
  s->pic_w_ctbs = (s->width + s->ctb_size - 1) >> s->ctb_log2;
  s->pic_h_ctbs = (s->height + s->ctb_size - 1) >> s->ctb_log2;
  s->log2_max_tb = s->log2_min_tb + s->log2_diff_max_min_tb;
}

int orc_parse_sps(orc_bitr *r, orc_sps *s)
{
  memset(s, 0, sizeof(*s));
  s->vps_id = (int)orc_br_get(r, 4);
  s->max_sub_layers = (int)orc_br_get(r, 3) + 1;
  orc_br_get(r, 1);
  if (parse_ptl(r, s->max_sub_layers - 1, &s->general_profile_idc, &s->general_level_idc)) return -1;
  s->sps_id = (int)orc_br_ue(r);
  if (s->sps_id > 15) return -1;
  s->chroma_format_idc = (int)orc_br_ue(r);
  if (s->chroma_format_idc != 1) return -2;                 /* only 4:2:0 */
  s->width = (int)orc_br_ue(r);
  s->height = (int)orc_br_ue(r);
  s->conf_win_flag = (int)orc_br_get(r, 1);
  if (s->conf_win_flag) {
    s->conf_left = (int)orc_br_ue(r); s->conf_right = (int)orc_br_ue(r);
    s->conf_top = (int)orc_br_ue(r);  s->conf_bottom = (int)orc_br_ue(r);
  }
  s->bit_depth_luma = (int)orc_br_ue(r) + 8;
  s->bit_depth_chroma = (int)orc_br_ue(r) + 8;
  if (s->bit_depth_luma != 8 || s->bit_depth_chroma != 8) return -2;
  s->log2_max_poc_lsb = (int)orc_br_ue(r) + 4;
  int oi = (int)orc_br_get(r, 1);
  for (int i = oi ? 0 : s->max_sub_layers - 1; i < s->max_sub_layers; i++) {
    s->max_dec_pic_buffering = (int)orc_br_ue(r) + 1;
    s->max_num_reorder = (int)orc_br_ue(r);
    s->max_latency_increase_plus1 = (int)orc_br_ue(r);
  }
  s->log2_min_cb = (int)orc_br_ue(r) + 3;
  s->log2_diff_max_min_cb = (int)orc_br_ue(r);
  s->log2_min_tb = (int)orc_br_ue(r) + 2;
  s->log2_diff_max_min_tb = (int)orc_br_ue(r);
  s->max_th_depth_inter = (int)orc_br_ue(r);
  s->max_th_depth_intra = (int)orc_br_ue(r);
  s->scaling_list_enabled = (int)orc_br_get(r, 1);
  if (s->scaling_list_enabled) {
    orc_scaling_default(&s->scaling);
    s->scaling_list_data_present = (int)orc_br_get(r, 1);
    if (s->scaling_list_data_present && orc_scaling_parse(r, &s->scaling) < 0) return -1;
  }
  s->amp_enabled = (int)orc_br_get(r, 1);
  s->sao_enabled = (int)orc_br_get(r, 1);
  s->pcm_enabled = (int)orc_br_get(r, 1);
  if (s->pcm_enabled) {
    s->pcm_bit_depth_luma = (int)orc_br_get(r, 4) + 1;
    s->pcm_bit_depth_chroma = (int)orc_br_get(r, 4) + 1;
    s->log2_min_pcm_cb = (int)orc_br_ue(r) + 3;
    s->log2_diff_max_min_pcm_cb = (int)orc_br_ue(r);
    s->pcm_loop_filter_disabled = (int)orc_br_get(r, 1);
  }
  s->num_st_rps = (int)orc_br_ue(r);
  if (s->num_st_rps > 64) return -1;
  for (int i = 0; i < s->num_st_rps; i++)
    if (parse_st_rps(r, &s->st_rps[i], i, s->num_st_rps, s->st_rps)) return -1;
  s->long_term_ref_pics_present = (int)orc_br_get(r, 1);
  s->num_lt_sps = 0;
  if (s->long_term_ref_pics_present) {
    s->num_lt_sps = (int)orc_br_ue(r);
    if (s->num_lt_sps > 32) return -1;
    for (int i = 0; i < s->num_lt_sps; i++) { s->lt_poc_lsb_sps[i] = (int)orc_br_get(r, s->log2_max_poc_lsb); s->lt_used_sps[i] = (uint8_t)orc_br_get(r, 1); }
  }
  s->temporal_mvp_enabled = (int)orc_br_get(r, 1);
  s->strong_intra_smoothing = (int)orc_br_get(r, 1);
  s->vui_present = (int)orc_br_get(r, 1);
  if (s->vui_present) {
    if (orc_br_get(r, 1)) { if (orc_br_get(r, 8) == 255) { orc_br_get(r, 16); orc_br_get(r, 16); } }
    if (orc_br_get(r, 1)) orc_br_get(r, 1);
    if (orc_br_get(r, 1)) { orc_br_get(r, 4); if (orc_br_get(r, 1)) orc_br_get(r, 24); }
    if (orc_br_get(r, 1)) { orc_br_ue(r); orc_br_ue(r); }
    orc_br_get(r, 3);
    if (orc_br_get(r, 1)) { orc_br_ue(r); orc_br_ue(r); orc_br_ue(r); orc_br_ue(r); }
    s->vui_timing_present = (int)orc_br_get(r, 1);
    if (s->vui_timing_present) {
      s->vui_num_units_in_tick = orc_br_get(r, 32);
      s->vui_time_scale = orc_br_get(r, 32);
    }
    /* the rest of the VUI and any SPS extension carry nothing the decoding process needs */
  }
  if (r->error) return -1;
  if (s->log2_min_cb < 3 || s->log2_min_cb + s->log2_diff_max_min_cb > 6 || s->log2_min_cb + s->log2_diff_max_min_cb < 4) return -1;
  if (s->log2_min_tb < 2 || s->log2_min_tb + s->log2_diff_max_min_tb > 5) return -1;
  if (s->width <= 0 || s->height <= 0 || (s->width & ((1 << s->log2_min_cb) - 1)) || (s->height & ((1 << s->log2_min_cb) - 1))) return -1;
  orc_sps_derive(s);
  s->valid = 1;
  return 0;
}

int orc_parse_pps(orc_bitr *r, orc_pps *p)
{
  memset(p, 0, sizeof(*p));
  p->pps_id = (int)orc_br_ue(r);
  p->sps_id = (int)orc_br_ue(r);
  if (p->pps_id > 63 || p->sps_id > 15) return -1;
  p->dependent_slice_segments_enabled = (int)orc_br_get(r, 1);
  p->output_flag_present = (int)orc_br_get(r, 1);
  p->num_extra_slice_header_bits = (int)orc_br_get(r, 3);
  p->sign_data_hiding = (int)orc_br_get(r, 1);
  p->cabac_init_present = (int)orc_br_get(r, 1);
  p->num_ref_idx_l0_default = (int)orc_br_ue(r) + 1;
  p->num_ref_idx_l1_default = (int)orc_br_ue(r) + 1;
  p->init_qp = 26 + orc_br_se(r);
  p->constrained_intra_pred = (int)orc_br_get(r, 1);
  p->transform_skip_enabled = (int)orc_br_get(r, 1);
  p->cu_qp_delta_enabled = (int)orc_br_get(r, 1);
  if (p->cu_qp_delta_enabled) p->diff_cu_qp_delta_depth = (int)orc_br_ue(r);
  p->cb_qp_offset = orc_br_se(r);
  p->cr_qp_offset = orc_br_se(r);
  p->slice_chroma_qp_offsets_present = (int)orc_br_get(r, 1);
  p->weighted_pred = (int)orc_br_get(r, 1);
  p->weighted_bipred = (int)orc_br_get(r, 1);
  p->transquant_bypass_enabled = (int)orc_br_get(r, 1);
  p->tiles_enabled = (int)orc_br_get(r, 1);
  p->entropy_coding_sync_enabled = (int)orc_br_get(r, 1);
  p->num_tile_columns = p->num_tile_rows = 1; p->uniform_spacing = 1;
  p->loop_filter_across_tiles = 1;
  if (p->tiles_enabled) {
    p->num_tile_columns = (int)orc_br_ue(r) + 1;
    p->num_tile_rows = (int)orc_br_ue(r) + 1;
    if (p->num_tile_columns > 20 || p->num_tile_rows > 22) return -1;
    p->uniform_spacing = (int)orc_br_get(r, 1);
    if (!p->uniform_spacing) {
      for (int i = 0; i < p->num_tile_columns - 1; i++) p->column_width[i] = (int)orc_br_ue(r) + 1;
      for (int i = 0; i < p->num_tile_rows - 1; i++) p->row_height[i] = (int)orc_br_ue(r) + 1;
    }
    p->loop_filter_across_tiles = (int)orc_br_get(r, 1);
  }
  p->loop_filter_across_slices = (int)orc_br_get(r, 1);
  p->deblocking_filter_control_present = (int)orc_br_get(r, 1);
  if (p->deblocking_filter_control_present) {
    p->deblocking_filter_override_enabled = (int)orc_br_get(r, 1);
    p->pps_deblocking_disabled = (int)orc_br_get(r, 1);
    if (!p->pps_deblocking_disabled) { p->pps_beta_offset_div2 = orc_br_se(r); p->pps_tc_offset_div2 = orc_br_se(r); }
  }
  p->scaling_list_data_present = (int)orc_br_get(r, 1);
  if (p->scaling_list_data_present) { orc_scaling_default(&p->scaling); if (orc_scaling_parse(r, &p->scaling) < 0) return -1; }
  p->lists_modification_present = (int)orc_br_get(r, 1);
  p->log2_parallel_merge_level = (int)orc_br_ue(r) + 2;
  p->slice_header_extension_present = (int)orc_br_get(r, 1);
  if (r->error) return -1;
  p->valid = 1;
  return 0;
}

/* *h holds the previous slice segment's header on entry (entry_point_offset already released by the caller): a dependent slice
 * segment (7.3.6.1) takes over everything but its address and its entry points from it */
/* 7.4.7.3: LumaWeightLX, luma_offset_lX, ChromaWeightLX, ChromaOffsetLX (8-bit video: no scaling of the offsets) */
void orc_derive_pred_weights(orc_slice_hdr *h)
{
  const int ld = h->luma_log2_weight_denom, cd = ld + h->delta_chroma_log2_weight_denom;
  h->wp_log2wd[0] = ld; h->wp_log2wd[1] = cd;
  for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) {
    h->wp_w[l][i][0] = (int16_t)((1 << ld) + (h->luma_weight_flag[l][i] ? h->delta_luma_weight[l][i] : 0));
    h->wp_o[l][i][0] = (int16_t)(h->luma_weight_flag[l][i] ? h->luma_offset[l][i] : 0);
    for (int j = 0; j < 2; j++) {
      const int w = (1 << cd) + (h->chroma_weight_flag[l][i] ? h->delta_chroma_weight[l][i][j] : 0);
      int o = 0;
      if (h->chroma_weight_flag[l][i]) { o = 128 + h->delta_chroma_offset[l][i][j] - ((128 * w) >> cd); o = o < -128 ? -128 : (o > 127 ? 127 : o); }
      h->wp_w[l][i][1 + j] = (int16_t)w; h->wp_o[l][i][1 + j] = (int16_t)o;
    }
  }
}

int orc_parse_slice_header(orc_bitr *r, orc_slice_hdr *h, int nal_type, const orc_sps *sps_tab, const orc_pps *pps_tab)
{
  const orc_slice_hdr prev = *h;
  memset(h, 0, sizeof(*h));
  h->first_slice_segment_in_pic = (int)orc_br_get(r, 1);
  if (nal_type >= NAL_BLA_W_LP && nal_type <= NAL_RSV_IRAP_VCL23) h->no_output_of_prior_pics = (int)orc_br_get(r, 1);
  h->pps_id = (int)orc_br_ue(r);
  if (h->pps_id > 63 || !pps_tab[h->pps_id].valid) return -1;
  const orc_pps *p = &pps_tab[h->pps_id];
  if (!sps_tab[p->sps_id].valid) return -1;
  const orc_sps *s = &sps_tab[p->sps_id];
  if (!h->first_slice_segment_in_pic) {
    if (p->dependent_slice_segments_enabled) h->dependent_slice_segment = (int)orc_br_get(r, 1);
    h->slice_segment_address = (int)orc_br_get(r, ceil_log2((unsigned)(s->pic_w_ctbs * s->pic_h_ctbs)));
  }
  if (h->dependent_slice_segment) {
    const int addr = h->slice_segment_address, pps_id = h->pps_id;
    if (prev.pps_id != pps_id) return -1;
    *h = prev;
    h->first_slice_segment_in_pic = 0; h->dependent_slice_segment = 1; h->slice_segment_address = addr;
    h->num_entry_points = 0; h->entry_point_offset = NULL;
    goto entry_points;
  }
  for (int i = 0; i < p->num_extra_slice_header_bits; i++) orc_br_get(r, 1);
  h->slice_type = (int)orc_br_ue(r);
  if (h->slice_type > 2) return -1;
  h->pic_output_flag = 1;
  if (p->output_flag_present) h->pic_output_flag = (int)orc_br_get(r, 1);
  if (nal_type != NAL_IDR_W_RADL && nal_type != NAL_IDR_N_LP) {
    h->poc_lsb = (int)orc_br_get(r, s->log2_max_poc_lsb);
    h->short_term_ref_pic_set_sps_flag = (int)orc_br_get(r, 1);
    if (!h->short_term_ref_pic_set_sps_flag) {
      if (parse_st_rps(r, &h->st_rps, s->num_st_rps, s->num_st_rps, s->st_rps)) return -1;
    } else {
      if (s->num_st_rps > 1) h->short_term_rps_idx = (int)orc_br_get(r, ceil_log2((unsigned)s->num_st_rps));
      if (h->short_term_rps_idx >= s->num_st_rps) return -1;
      h->st_rps = s->st_rps[h->short_term_rps_idx];
    }
    h->num_long_term_sps = h->num_long_term_pics = h->num_lt = 0;
    if (s->long_term_ref_pics_present) {
      if (s->num_lt_sps > 0) h->num_long_term_sps = (int)orc_br_ue(r);
      h->num_long_term_pics = (int)orc_br_ue(r);
      if (h->num_long_term_sps < 0 || h->num_long_term_sps > s->num_lt_sps || h->num_long_term_pics < 0 || h->num_long_term_sps + h->num_long_term_pics > 16) return -1;
      h->num_lt = h->num_long_term_sps + h->num_long_term_pics;
      for (int i = 0; i < h->num_lt; i++) {
        if (i < h->num_long_term_sps) {
          h->lt_idx_sps[i] = s->num_lt_sps > 1 ? (int)orc_br_get(r, ceil_log2((unsigned)s->num_lt_sps)) : 0;
          if (h->lt_idx_sps[i] >= s->num_lt_sps) return -1;
          h->lt_poc_lsb[i] = s->lt_poc_lsb_sps[h->lt_idx_sps[i]]; h->lt_used[i] = s->lt_used_sps[h->lt_idx_sps[i]];
        } else { h->lt_poc_lsb[i] = (int)orc_br_get(r, s->log2_max_poc_lsb); h->lt_used[i] = (uint8_t)orc_br_get(r, 1); }
        h->lt_msb_present[i] = (uint8_t)orc_br_get(r, 1);
        h->lt_msb_cycle_delta[i] = h->lt_msb_present[i] ? (int)orc_br_ue(r) : 0;
        h->lt_msb_cycle[i] = h->lt_msb_cycle_delta[i] + ((i == 0 || i == h->num_long_term_sps) ? 0 : h->lt_msb_cycle[i - 1]);      /* (7-52) */
      }
    }
    if (s->temporal_mvp_enabled) h->slice_temporal_mvp_enabled = (int)orc_br_get(r, 1);
  }
  if (s->sao_enabled) { h->sao_luma = (int)orc_br_get(r, 1); h->sao_chroma = (int)orc_br_get(r, 1); }
  h->collocated_from_l0 = 1;
  if (h->slice_type != SLICE_I) {
    h->num_ref_idx_l0 = p->num_ref_idx_l0_default; h->num_ref_idx_l1 = p->num_ref_idx_l1_default;
    if (orc_br_get(r, 1)) {
      h->num_ref_idx_l0 = (int)orc_br_ue(r) + 1;
      if (h->slice_type == SLICE_B) h->num_ref_idx_l1 = (int)orc_br_ue(r) + 1;
    }
    if (h->num_ref_idx_l0 > 16 || h->num_ref_idx_l1 > 16) return -1;
    int num_pic_total = 0;
    for (int i = 0; i < h->st_rps.num_negative; i++) num_pic_total += h->st_rps.used_s0[i];
    for (int i = 0; i < h->st_rps.num_positive; i++) num_pic_total += h->st_rps.used_s1[i];
    for (int i = 0; i < h->num_lt; i++) num_pic_total += h->lt_used[i];
    h->rpl_mod_flag[0] = h->rpl_mod_flag[1] = 0;
    if (p->lists_modification_present && num_pic_total > 1) {              /* ref_pic_lists_modification() */
      const int bits = ceil_log2((unsigned)num_pic_total);
      for (int l = 0; l < (h->slice_type == SLICE_B ? 2 : 1); l++) {
        h->rpl_mod_flag[l] = (int)orc_br_get(r, 1);
        if (h->rpl_mod_flag[l]) for (int i = 0; i < (l ? h->num_ref_idx_l1 : h->num_ref_idx_l0); i++) {
          h->list_entry[l][i] = (uint8_t)orc_br_get(r, bits);
          if (h->list_entry[l][i] >= num_pic_total) return -1;
        }
      }
    }
    if (h->slice_type == SLICE_B) h->mvd_l1_zero = (int)orc_br_get(r, 1);
    if (p->cabac_init_present) h->cabac_init_flag = (int)orc_br_get(r, 1);
    if (h->slice_temporal_mvp_enabled) {
      if (h->slice_type == SLICE_B) h->collocated_from_l0 = (int)orc_br_get(r, 1);
      if ((h->collocated_from_l0 && h->num_ref_idx_l0 > 1) || (!h->collocated_from_l0 && h->num_ref_idx_l1 > 1))
        h->collocated_ref_idx = (int)orc_br_ue(r);
    }
    h->weighted = 0;
    if ((p->weighted_pred && h->slice_type == SLICE_P) || (p->weighted_bipred && h->slice_type == SLICE_B)) {
      h->weighted = 1;
      memset(h->luma_weight_flag, 0, sizeof(h->luma_weight_flag)); memset(h->chroma_weight_flag, 0, sizeof(h->chroma_weight_flag));
      h->luma_log2_weight_denom = (int)orc_br_ue(r);
      h->delta_chroma_log2_weight_denom = orc_br_se(r);
      if (h->luma_log2_weight_denom > 7 || h->luma_log2_weight_denom + h->delta_chroma_log2_weight_denom < 0 || h->luma_log2_weight_denom + h->delta_chroma_log2_weight_denom > 7) return -1;
      for (int l = 0; l < (h->slice_type == SLICE_B ? 2 : 1); l++) {
        const int n = l ? h->num_ref_idx_l1 : h->num_ref_idx_l0;
        for (int i = 0; i < n; i++) h->luma_weight_flag[l][i] = (uint8_t)orc_br_get(r, 1);
        for (int i = 0; i < n; i++) h->chroma_weight_flag[l][i] = (uint8_t)orc_br_get(r, 1);
        for (int i = 0; i < n; i++) {
          if (h->luma_weight_flag[l][i]) {
            const int dw = orc_br_se(r), lo = orc_br_se(r);
            if (dw < -128 || dw > 127 || lo < -128 || lo > 127) return -1;
            h->delta_luma_weight[l][i] = (int16_t)dw; h->luma_offset[l][i] = (int16_t)lo;
          }
          if (h->chroma_weight_flag[l][i]) for (int j = 0; j < 2; j++) {
            const int dw = orc_br_se(r), dof = orc_br_se(r);
            if (dw < -128 || dw > 127 || dof < -512 || dof > 511) return -1;
            h->delta_chroma_weight[l][i][j] = (int16_t)dw; h->delta_chroma_offset[l][i][j] = (int16_t)dof;
          }
        }
      }
      orc_derive_pred_weights(h);
    }
    h->max_num_merge_cand = 5 - (int)orc_br_ue(r);
    if (h->max_num_merge_cand < 1 || h->max_num_merge_cand > 5) return -1;
  }
  h->slice_qp_delta = orc_br_se(r);
  if (p->slice_chroma_qp_offsets_present) { h->slice_cb_qp_offset = orc_br_se(r); h->slice_cr_qp_offset = orc_br_se(r); }
  if (p->deblocking_filter_override_enabled) h->deblocking_filter_override = (int)orc_br_get(r, 1);
  h->slice_deblocking_disabled = p->pps_deblocking_disabled;
  h->beta_offset_div2 = p->pps_beta_offset_div2; h->tc_offset_div2 = p->pps_tc_offset_div2;
  if (h->deblocking_filter_override) {
    h->slice_deblocking_disabled = (int)orc_br_get(r, 1);
    if (!h->slice_deblocking_disabled) { h->beta_offset_div2 = orc_br_se(r); h->tc_offset_div2 = orc_br_se(r); }
  }
  h->loop_filter_across_slices = p->loop_filter_across_slices;
  if (p->loop_filter_across_slices && (h->sao_luma || h->sao_chroma || !h->slice_deblocking_disabled))
    h->loop_filter_across_slices = (int)orc_br_get(r, 1);
entry_points:
  if (p->tiles_enabled || p->entropy_coding_sync_enabled) {
    h->num_entry_points = (int)orc_br_ue(r);
    if (h->num_entry_points > 440 * 135) return -1;
    if (h->num_entry_points > 0) {
      int len = (int)orc_br_ue(r) + 1;
      if (len > 32) return -1;
      h->entry_point_offset = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)h->num_entry_points);
      for (int i = 0; i < h->num_entry_points; i++) h->entry_point_offset[i] = orc_br_get(r, len) + 1;
    }
  }
  if (p->slice_header_extension_present) {
    int n = (int)orc_br_ue(r);
    for (int i = 0; i < n; i++) orc_br_get(r, 8);
  }
  /* byte_alignment() */
  if (!orc_br_get(r, 1)) return -1;
  while (r->pos & 7) orc_br_get(r, 1);
  h->slice_qp = p->init_qp + h->slice_qp_delta;
  if (r->error) { free(h->entry_point_offset); h->entry_point_offset = NULL; return -1; }
  return 0;
}
