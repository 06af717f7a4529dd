/* oracle/oracle_api.c -- flat entry points for ctypes (tests/, bench.py cpu_baseline).
 * Test infrastructure. */
#include "hevc_enc.h"
#include "hevc_dec.h"
#include "hevc_transform.h"
#include "hevc_intra.h"
#include "hevc_inter.h"
#include "hevc_deblock.h"
#include "hevc_cabac.h"
#include "synth.h"
#include "hevc_gen.h"

int orc_api_version(void) { return 1; }

/* Closed loop on a synthetic clip: encode `frames` pictures, decode the stream NAL by NAL and
 * compare every decoded picture with the encoder's reconstruction.  Returns the number of
 * mismatching pictures (0 = pass), or a negative decoder error. total_bytes receives the
 * stream size. */
static size_t next_start(const uint8_t *p, size_t n, size_t from)
{
  for (size_t i = from; i + 3 < n; i++) if (p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 0 && p[i + 3] == 1) return i;
  return n;
}
int orc_api_closed_loop(int w, int h, int frames, int qp, int period, int range, int kind, uint32_t seed, int wpp,
                        uint64_t *total_bytes, uint64_t *total_bins)
{
  orc_enc_config cfg; orc_enc_default_config(&cfg);
  cfg.width = w; cfg.height = h; cfg.qp = qp; cfg.intra_period = period; cfg.search_range = range; cfg.wpp = wpp;
  orc_encoder *e = orc_enc_open(&cfg);
  if (!e) return -100;
  orc_decoder *d = orc_dec_open();
  uint8_t *in = (uint8_t *)malloc((size_t)w * h * 3 / 2), *rec = (uint8_t *)malloc((size_t)w * h * 3 / 2);
  int bad = 0; uint64_t bytes = 0, bins = 0;
  for (int t = 0; t < frames && bad >= 0; t++) {
    orc_synth_frame(kind, seed, w, h, t, in);
    const uint8_t *au; size_t n = orc_enc_encode(e, in, in + (size_t)w * h, in + (size_t)w * h * 5 / 4, &au);
    bytes += n;
    orc_enc_debug dbg; orc_enc_get_debug(e, &dbg); bins += dbg.bins;
    orc_enc_get_recon(e, rec, rec + (size_t)w * h, rec + (size_t)w * h * 5 / 4);
    size_t pos = 0; int got = 0;
    while (pos < n) {
      size_t nx = next_start(au, n, pos + 4);
      int r = orc_dec_decode_nal(d, au + pos, nx - pos, t);
      if (r < 0) { bad = r; break; }
      if (r > 0) got = 1;
      pos = nx;
    }
    if (bad < 0) break;
    orc_dec_frame f;
    if (!got || !orc_dec_get_frame(d, &f) || f.width != w || f.height != h) { bad++; continue; }
    int diff = 0;
    for (int c = 0; c < 3 && !diff; c++) {
      int pw = c ? w / 2 : w, ph = c ? h / 2 : h;
      const uint8_t *r0 = c == 0 ? rec : (c == 1 ? rec + (size_t)w * h : rec + (size_t)w * h * 5 / 4);
      for (int y = 0; y < ph && !diff; y++) if (memcmp(f.plane[c] + (size_t)y * f.stride[c], r0 + (size_t)y * pw, (size_t)pw)) diff = 1;
    }
    bad += diff;
  }
  if (total_bytes) *total_bytes = bytes;
  if (total_bins) *total_bins = bins;
  free(in); free(rec); orc_enc_close(e); orc_dec_close(d);
  return bad;
}

/* ---- thin wrappers for ctypes ---- */
orc_encoder *orc_api_enc_open(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock)
{
  orc_enc_config c; orc_enc_default_config(&c);
  c.width = w; c.height = h; c.qp = qp; c.intra_period = period; c.vps_period = vps_period; c.search_range = range;
  c.fps_num = fps_num; c.fps_den = fps_den; c.wpp = wpp; c.deblock = deblock;
  return orc_enc_open(&c);
}
orc_encoder *orc_api_enc_open_ex(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock, int bitrate, int tile_rows);
orc_encoder *orc_api_enc_open_ex2(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock, int bitrate, int tile_rows, int tile_cols);
/* same with picture-level rate control (bits per second) */
orc_encoder *orc_api_enc_open_rc(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock, int bitrate)
{
  return orc_api_enc_open_ex(w, h, qp, period, vps_period, range, fps_num, fps_den, wpp, deblock, bitrate, 1);
}
/* ... tile rows; qp_in_cu and sao packed into bits 16, 17 of tile_rows keep the ctypes signature short */
orc_encoder *orc_api_enc_open_ex2(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock, int bitrate, int tile_rows, int tile_cols)
{
  orc_enc_config c; orc_enc_default_config(&c); c.tile_cols = tile_cols; c.tile_rows = tile_rows & 0xff; c.rc_bands = (tile_rows >> 8) & 0xf; c.slices = (tile_rows >> 12) & 3; c.qp_in_cu = (tile_rows >> 16) & 1; c.sao = (tile_rows >> 17) & 1; c.test_mv_jitter = (tile_rows >> 18) & 1; c.mv_frame = (tile_rows >> 19) & 3; c.vaq = (tile_rows >> 21) & 31; if ((tile_rows >> 26) & 1) c.me_early = 0; if ((tile_rows >> 27) & 1) c.satd = 0; c.subme = (tile_rows >> 28) & 7;
  c.width = w; c.height = h; c.qp = qp; c.intra_period = period; c.vps_period = vps_period; c.search_range = range;
  c.fps_num = fps_num; c.fps_den = fps_den; c.wpp = wpp; c.deblock = deblock; c.bitrate = bitrate;
  return orc_enc_open(&c);
}
orc_encoder *orc_api_enc_open_ex(int w, int h, int qp, int period, int vps_period, int range, int fps_num, int fps_den, int wpp, int deblock, int bitrate, int tile_rows)
{
  return orc_api_enc_open_ex2(w, h, qp, period, vps_period, range, fps_num, fps_den, wpp, deblock, bitrate, tile_rows, 1);
}
void orc_api_enc_set_roi(orc_encoder *e, int w, int h, const int8_t *map) { orc_enc_set_roi(e, w, h, map); }
/* returns AU size; copies it to out when it fits */
long orc_api_enc_encode(orc_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, uint8_t *out, long cap)
{
  const uint8_t *au; size_t n = orc_enc_encode(e, y, u, v, &au);
  if ((long)n <= cap) memcpy(out, au, n);
  return (long)n;
}
/* split an Annex-B access unit into NAL units (4-byte start codes): writes offsets, returns count */
int orc_api_split_nals(const uint8_t *au, long n, long *offsets, int max)
{
  int cnt = 0;
  for (long i = 0; i + 3 < n; i++)
    if (au[i] == 0 && au[i + 1] == 0 && au[i + 2] == 0 && au[i + 3] == 1) { if (cnt < max) offsets[cnt] = i; cnt++; i += 3; }
  return cnt;
}

/* ---- stage-level entry points for the unit tests ---- */
void orc_api_tables(int which, void *dst)
{
  orc_tables_init();
  switch (which) {
    case 0: memcpy(dst, orc_dct_mat, 32 * 32); break;
    case 1: memcpy(dst, orc_range_tab_lps, 256); break;
    case 2: memcpy(dst, orc_trans_idx_lps, 64); break;
    case 3: memcpy(dst, orc_cabac_init_values, 3 * CTX_COUNT); break;
    case 4: memcpy(dst, orc_lambda_q4, 52 * 2); break;
    case 5: memcpy(dst, orc_dst_mat, 16); break;
    case 6: memcpy(dst, orc_beta_table, 52); break;
    case 7: memcpy(dst, orc_tc_table, 54); break;
    case 8: memcpy(dst, orc_chroma_qp_table, 58); break;
    case 9: memcpy(dst, orc_intra_angle, 35); break;
    case 10: memcpy(dst, orc_inv_angle, 35 * 2); break;
    case 11: memcpy(dst, orc_luma_filter, 32); break;
    case 12: memcpy(dst, orc_chroma_filter, 32); break;
    case 13: memcpy(dst, orc_trans_idx_mps, 64); break;
  }
}
void orc_api_intra_predict(const uint8_t *left, const uint8_t *top, int n, int cidx, int mode, int strong, uint8_t *pred)
{
  orc_intra_predict(left, top, n, cidx, mode, strong, pred, n);
}
/* deblock a coded-size picture in place given explicit bS maps and a constant QP */
void orc_api_deblock(int w, int h, uint8_t *y, uint8_t *u, uint8_t *v, const uint8_t *bs_v, const uint8_t *bs_h, int qp)
{
  orc_deblock_ctx d; memset(&d, 0, sizeof(d));
  int8_t *qpm = (int8_t *)malloc((size_t)(w / 4) * (h / 4));
  memset(qpm, qp, (size_t)(w / 4) * (h / 4));
  d.w = w; d.h = h; d.plane[0] = y; d.plane[1] = u; d.plane[2] = v; d.stride[0] = w; d.stride[1] = d.stride[2] = w / 2;
  d.bs_v = bs_v; d.bs_stride_v = w / 8; d.bs_h = bs_h; d.bs_stride_h = w / 4; d.qp_y = qpm; d.qp_stride = w / 4;
  orc_deblock_picture(&d);
  free(qpm);
}
/* CABAC context initialisation (9.3.2.2): out[i] = state << 1 | mps */
void orc_api_cabac_init(int init_type, int qp, uint8_t *out)
{
  orc_ctx c[CTX_COUNT];
  orc_cabac_init_contexts(c, init_type, qp);
  for (int i = 0; i < CTX_COUNT; i++) out[i] = (uint8_t)((c[i].state << 1) | c[i].mps);
}
/* encode a bin string with the spec-literal arithmetic coder and decode it again: kinds 0 ctx (ci), 1 bypass, 2 terminate */
long orc_api_cabac_roundtrip(const uint8_t *kinds, const uint8_t *ci, const uint8_t *bins, long n, int qp, uint8_t *out, long cap, uint8_t *decoded)
{
  orc_bitw bw; orc_bw_init(&bw);
  orc_cabac_enc e; memset(&e, 0, sizeof(e));
  orc_cenc_start(&e, &bw);
  orc_cabac_init_contexts(e.ctx, 0, qp);
  for (long i = 0; i < n; i++) {
    if (kinds[i] == 0) orc_cenc_bin(&e, ci[i], bins[i]);
    else if (kinds[i] == 1) orc_cenc_bypass(&e, bins[i]);
    else orc_cenc_terminate(&e, 0);
  }
  orc_cenc_terminate(&e, 1);
  orc_bw_align_zero(&bw);
  long len = (long)bw.len;
  if (len <= cap) memcpy(out, bw.buf, bw.len);
  orc_cabac_dec d; memset(&d, 0, sizeof(d));
  orc_cdec_start(&d, bw.buf, bw.len);
  orc_cabac_init_contexts(d.ctx, 0, qp);
  for (long i = 0; i < n; i++) {
    if (kinds[i] == 0) decoded[i] = (uint8_t)orc_cdec_bin(&d, ci[i]);
    else if (kinds[i] == 1) decoded[i] = (uint8_t)orc_cdec_bypass(&d);
    else decoded[i] = (uint8_t)orc_cdec_terminate(&d);
  }
  int term = orc_cdec_terminate(&d);
  long consumed = (long)orc_cdec_bytes_consumed(&d);
  orc_bw_free(&bw);
  return (term == 1 && consumed == len) ? len : -len;
}

/* ---- stream synthesiser (hevc_gen.c): cfg = the orc_gen_config fields as a flat int array ---- */
orc_gen *orc_api_gen_open(const int *cfg, int n)
{
  orc_gen_config c; orc_gen_default_config(&c);
  int *f = (int *)&c;
  for (int i = 0; i < n && i < (int)(sizeof(c) / sizeof(int)); i++) f[i] = cfg[i];
  return orc_gen_open(&c);
}
int orc_api_gen_config(const orc_gen *g, int *out, int n)
{
  orc_gen_config c; orc_gen_get_config(g, &c);
  const int *f = (const int *)&c, m = (int)(sizeof(c) / sizeof(int));
  for (int i = 0; i < n && i < m; i++) out[i] = f[i];
  return m;
}
long orc_api_gen_picture(orc_gen *g, uint8_t *out, long cap)
{
  const uint8_t *au; size_t n = orc_gen_picture(g, &au);
  if ((long)n <= cap) memcpy(out, au, n);
  return (long)n;
}
