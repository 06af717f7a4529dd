/* oracle/hevc_bits.h -- bit writer/reader, Exp-Golomb (H.265 9.2), NAL byte-stream
 * framing with emulation prevention (H.265 7.3.1.1, Annex B).  Test infrastructure. */
#ifndef ORC_HEVC_BITS_H
#define ORC_HEVC_BITS_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint8_t *buf; size_t cap; size_t len;  /* complete bytes written */
  uint32_t cur; int nbits;               /* pending bits (MSB-first), nbits < 8 */
} orc_bitw;

void orc_bw_init(orc_bitw *w);
void orc_bw_free(orc_bitw *w);
void orc_bw_put(orc_bitw *w, uint32_t val, int n);   /* n <= 32 */
void orc_bw_ue(orc_bitw *w, uint32_t v);
void orc_bw_se(orc_bitw *w, int32_t v);
void orc_bw_trailing(orc_bitw *w);                    /* rbsp_trailing_bits / byte_alignment: 1 then zeros */
void orc_bw_align_zero(orc_bitw *w);
void orc_bw_bytes(orc_bitw *w, const uint8_t *p, size_t n); /* must be byte aligned */
static inline int orc_bw_aligned(const orc_bitw *w) { return w->nbits == 0; }

/* Append one NAL unit to `out` (byte aligned): start code (3 or 4 bytes), 2-byte NAL header,
 * payload with emulation prevention. */
void orc_write_nal(orc_bitw *out, int nal_type, int temporal_id, const uint8_t *rbsp, size_t n, int long_start_code);
/* escaped size of a byte run assuming it starts after a non-zero byte */
size_t orc_escaped_size(const uint8_t *p, size_t n);

typedef struct {
  const uint8_t *buf; size_t len; size_t pos;  /* pos in bits */
  int error;
} orc_bitr;

void     orc_br_init(orc_bitr *r, const uint8_t *buf, size_t len);
uint32_t orc_br_get(orc_bitr *r, int n);      /* n <= 32 */
uint32_t orc_br_ue(orc_bitr *r);
int32_t  orc_br_se(orc_bitr *r);
static inline int orc_br_bit(orc_bitr *r) {
  if (r->pos >= r->len * 8) { r->error = 1; r->pos++; return 0; }
  int b = (r->buf[r->pos >> 3] >> (7 - (r->pos & 7))) & 1; r->pos++; return b;
}
/* Remove emulation prevention bytes: returns rbsp length; epb_pos (optional) receives the
 * positions (in the ESCAPED payload) of removed bytes, up to max_epb. */
size_t orc_unescape(const uint8_t *in, size_t n, uint8_t *out, size_t *epb_pos, int max_epb, int *n_epb);

#ifdef __cplusplus
}
#endif
#endif
