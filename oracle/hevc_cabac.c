/* oracle/hevc_cabac.c -- see hevc_cabac.h.  Test infrastructure. */
#include "hevc_cabac.h"

#define CNU 154
/* H.265 Tables 9-5 .. 9-37 initValue, laid out in the CTX_* order; row = initType
 * (0: I slices, 1: P (cabac_init_flag 0), 2: B (cabac_init_flag 0)). */
const uint8_t orc_cabac_init_values[3][CTX_COUNT] = {
 { /* initType 0 */
  153,                         /* sao_merge */
  200,                         /* sao_type_idx */
  139, 141, 157,               /* split_cu_flag */
  154,                         /* cu_transquant_bypass_flag */
  CNU, CNU, CNU,               /* cu_skip_flag */
  CNU,                         /* pred_mode_flag */
  184, CNU, CNU, CNU,          /* part_mode */
  184,                         /* prev_intra_luma_pred_flag */
  63,                          /* intra_chroma_pred_mode */
  CNU,                         /* rqt_root_cbf */
  CNU,                         /* merge_flag */
  CNU,                         /* merge_idx */
  CNU, CNU, CNU, CNU, CNU,     /* inter_pred_idc */
  CNU, CNU,                    /* ref_idx */
  CNU,                         /* mvp_flag */
  153, 138, 138,               /* split_transform_flag */
  111, 141,                    /* cbf_luma */
  94, 138, 182, 154,           /* cbf_cb / cbf_cr */
  CNU, CNU,                    /* abs_mvd_greater0, greater1 */
  154, 154,                    /* cu_qp_delta_abs */
  139, 139,                    /* transform_skip_flag luma, chroma */
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, /* last_x */
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, /* last_y */
  91, 171, 134, 141,           /* coded_sub_block_flag */
  111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107,
  125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
  140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,   /* sig_coeff_flag */
  140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152,
  140, 179, 166, 182, 140, 227, 122, 197,                                        /* greater1 */
  138, 153, 136, 167, 152, 152 },                                                /* greater2 */
 { /* initType 1 */
  153,
  185,
  107, 139, 126,
  154,
  197, 185, 201,
  149,
  154, 139, 154, 154,
  154,
  152,
  79,
  110,
  122,
  95, 79, 63, 31, 31,
  153, 153,
  168,
  124, 138, 94,
  153, 111,
  149, 107, 167, 154,
  140, 198,
  154, 154,
  139, 139,
  125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
  125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
  121, 140, 61, 154,
  155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166,
  183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
  170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
  154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137,
  169, 194, 166, 167, 154, 167, 137, 182,
  107, 167, 91, 122, 107, 167 },
 { /* initType 2 */
  153,
  160,
  107, 139, 126,
  154,
  197, 185, 201,
  134,
  154, 139, 154, 154,
  183,
  152,
  79,
  154,
  137,
  95, 79, 63, 31, 31,
  153, 153,
  168,
  224, 167, 122,
  153, 111,
  149, 92, 167, 154,
  169, 198,
  154, 154,
  139, 139,
  125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
  125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
  121, 140, 61, 154,
  170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166,
  183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
  170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140,
  154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122,
  169, 208, 166, 167, 154, 152, 167, 182,
  107, 167, 91, 107, 107, 167 }
};

/* H.265 9.3.2.2 */
void orc_cabac_init_contexts(orc_ctx *ctx, int init_type, int slice_qp)
{
  int qp = orc_clip3(0, 51, slice_qp);
  for (int i = 0; i < CTX_COUNT; i++) {
    int v = orc_cabac_init_values[init_type][i];
    int slope = (v >> 4) * 5 - 45;
    int offs = ((v & 15) << 3) - 16;
    int pre = orc_clip3(1, 126, ((slope * qp) >> 4) + offs);
    ctx[i].mps = (pre <= 63) ? 0 : 1;
    ctx[i].state = (uint8_t)(ctx[i].mps ? (pre - 64) : (63 - pre));
  }
}

/* ---------------- encoder (H.265 9.3.4.x informative flowcharts; 10-bit low) -------------- */
void orc_cenc_start(orc_cabac_enc *c, orc_bitw *bw)
{
  c->bw = bw; c->low = 0; c->range = 510; c->first_bit = 1; c->outstanding = 0;
}
static void put_bit(orc_cabac_enc *c, int b)
{
  if (c->first_bit) c->first_bit = 0; else orc_bw_put(c->bw, (uint32_t)b, 1);
  while (c->outstanding > 0) { orc_bw_put(c->bw, (uint32_t)(1 - b), 1); c->outstanding--; }
}
static void renorm_e(orc_cabac_enc *c)
{
  while (c->range < 256) {
    if (c->low < 256) put_bit(c, 0);
    else if (c->low >= 512) { c->low -= 512; put_bit(c, 1); }
    else { c->low -= 256; c->outstanding++; }
    c->range <<= 1; c->low <<= 1;
  }
}
void orc_cenc_bin(orc_cabac_enc *c, int ci, int bin)
{
  orc_ctx *x = &c->ctx[ci];
  uint32_t lps = orc_range_tab_lps[x->state][(c->range >> 6) & 3];
  c->bins++;
  c->range -= lps;
  if (bin != x->mps) {
    c->low += c->range; c->range = lps;
    if (x->state == 0) x->mps = (uint8_t)(1 - x->mps);
    x->state = orc_trans_idx_lps[x->state];
  } else {
    x->state = orc_trans_idx_mps[x->state];
  }
  renorm_e(c);
}
void orc_cenc_bypass(orc_cabac_enc *c, int bin)
{
  c->bins++;
  c->low <<= 1;
  if (bin) c->low += c->range;
  if (c->low >= 1024) { put_bit(c, 1); c->low -= 1024; }
  else if (c->low < 512) put_bit(c, 0);
  else { c->low -= 512; c->outstanding++; }
}
void orc_cenc_bypass_bits(orc_cabac_enc *c, uint32_t val, int n)
{
  for (int i = n - 1; i >= 0; i--) orc_cenc_bypass(c, (int)((val >> i) & 1));
}
void orc_cenc_terminate(orc_cabac_enc *c, int bin)
{
  c->bins++;
  c->range -= 2;
  if (bin) {
    c->low += c->range;
    /* EncodeFlush */
    c->range = 2;
    renorm_e(c);
    put_bit(c, (int)((c->low >> 9) & 1));
    orc_bw_put(c->bw, ((c->low >> 7) & 3) | 1, 2);   /* last bit = rbsp_stop_one_bit / alignment_bit_equal_to_one */
  } else {
    renorm_e(c);
  }
}

/* ---------------- decoder (H.265 9.3.4.3) -------------- */
void orc_cdec_start(orc_cabac_dec *c, const uint8_t *buf, size_t len)
{
  orc_br_init(&c->br, buf, len);
  c->range = 510;
  c->offset = orc_br_get(&c->br, 9);
}
int orc_cdec_bin(orc_cabac_dec *c, int ci)
{
  orc_ctx *x = &c->ctx[ci];
  uint32_t lps = orc_range_tab_lps[x->state][(c->range >> 6) & 3];
  int bin;
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = 1 - x->mps;
    c->offset -= c->range; c->range = lps;
    if (x->state == 0) x->mps = (uint8_t)(1 - x->mps);
    x->state = orc_trans_idx_lps[x->state];
  } else {
    bin = x->mps;
    x->state = orc_trans_idx_mps[x->state];
  }
  while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | (uint32_t)orc_br_bit(&c->br); }
  return bin;
}
int orc_cdec_bypass(orc_cabac_dec *c)
{
  c->offset = (c->offset << 1) | (uint32_t)orc_br_bit(&c->br);
  if (c->offset >= c->range) { c->offset -= c->range; return 1; }
  return 0;
}
uint32_t orc_cdec_bypass_bits(orc_cabac_dec *c, int n)
{
  uint32_t v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | (uint32_t)orc_cdec_bypass(c);
  return v;
}
int orc_cdec_terminate(orc_cabac_dec *c)
{
  c->range -= 2;
  if (c->offset >= c->range) return 1;   /* no renormalisation; last bit read was the stop bit */
  while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | (uint32_t)orc_br_bit(&c->br); }
  return 0;
}
size_t orc_cdec_bytes_consumed(const orc_cabac_dec *c) { return (c->br.pos + 7) >> 3; }
