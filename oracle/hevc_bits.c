/* oracle/hevc_bits.c -- see hevc_bits.h.  Test infrastructure. */
#include "hevc_bits.h"

static void bw_reserve(orc_bitw *w, size_t extra)
{
  if (w->len + extra + 8 > w->cap) {
    size_t nc = w->cap ? w->cap * 2 : 4096;
    while (nc < w->len + extra + 8) nc *= 2;
    w->buf = (uint8_t *)realloc(w->buf, nc);
    w->cap = nc;
  }
}
void orc_bw_init(orc_bitw *w) { memset(w, 0, sizeof(*w)); }
void orc_bw_free(orc_bitw *w) { free(w->buf); memset(w, 0, sizeof(*w)); }

void orc_bw_put(orc_bitw *w, uint32_t val, int n)
{
  bw_reserve(w, 8);
  for (int i = n - 1; i >= 0; i--) {
    w->cur = (w->cur << 1) | ((val >> i) & 1u);
    if (++w->nbits == 8) { w->buf[w->len++] = (uint8_t)w->cur; w->cur = 0; w->nbits = 0; }
  }
}
/* H.265 9.2: ue(v) = prefix zeros + 1 + info bits */
void orc_bw_ue(orc_bitw *w, uint32_t v)
{
  uint32_t x = v + 1; int len = 0;
  while ((x >> len) > 1) len++;
  orc_bw_put(w, 0, len);
  orc_bw_put(w, x, len + 1);
}
void orc_bw_se(orc_bitw *w, int32_t v) { orc_bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
void orc_bw_trailing(orc_bitw *w) { orc_bw_put(w, 1, 1); orc_bw_align_zero(w); }
void orc_bw_align_zero(orc_bitw *w) { while (w->nbits) orc_bw_put(w, 0, 1); }
void orc_bw_bytes(orc_bitw *w, const uint8_t *p, size_t n)
{
  bw_reserve(w, n);
  memcpy(w->buf + w->len, p, n); w->len += n;
}

size_t orc_escaped_size(const uint8_t *p, size_t n)
{
  size_t out = 0; int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && p[i] <= 3) { out++; zeros = 0; }
    out++;
    zeros = (p[i] == 0) ? zeros + 1 : 0;
  }
  return out;
}

void orc_write_nal(orc_bitw *out, int nal_type, int temporal_id, const uint8_t *rbsp, size_t n, int long_start_code)
{
  bw_reserve(out, n + n / 2 + 16);
  if (long_start_code) out->buf[out->len++] = 0;
  out->buf[out->len++] = 0; out->buf[out->len++] = 0; out->buf[out->len++] = 1;
  /* nal_unit_header(): forbidden_zero_bit, nal_unit_type(6), nuh_layer_id(6), nuh_temporal_id_plus1(3) */
  out->buf[out->len++] = (uint8_t)(nal_type << 1);
  out->buf[out->len++] = (uint8_t)(temporal_id + 1);
  int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    bw_reserve(out, 4);
    if (zeros >= 2 && rbsp[i] <= 3) { out->buf[out->len++] = 3; zeros = 0; }
    out->buf[out->len++] = rbsp[i];
    zeros = (rbsp[i] == 0) ? zeros + 1 : 0;
  }
}

void orc_br_init(orc_bitr *r, const uint8_t *buf, size_t len) { r->buf = buf; r->len = len; r->pos = 0; r->error = 0; }
uint32_t orc_br_get(orc_bitr *r, int n)
{
  uint32_t v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | (uint32_t)orc_br_bit(r);
  return v;
}
uint32_t orc_br_ue(orc_bitr *r)
{
  int zeros = 0;
  while (!orc_br_bit(r)) { if (++zeros > 32 || r->error) { r->error = 1; return 0; } }
  if (zeros == 0) return 0;
  if (zeros == 32) return 0xffffffffu;
  return ((1u << zeros) - 1) + orc_br_get(r, zeros);
}
int32_t orc_br_se(orc_bitr *r)
{
  uint32_t k = orc_br_ue(r);
  return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
}

size_t orc_unescape(const uint8_t *in, size_t n, uint8_t *out, size_t *epb_pos, int max_epb, int *n_epb)
{
  size_t o = 0; int zeros = 0, ne = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && in[i] == 3) {
      if (epb_pos && ne < max_epb) epb_pos[ne] = i;
      ne++; zeros = 0; continue;
    }
    out[o++] = in[i];
    zeros = (in[i] == 0) ? zeros + 1 : 0;
  }
  if (n_epb) *n_epb = ne;
  return o;
}
