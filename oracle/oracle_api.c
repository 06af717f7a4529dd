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
