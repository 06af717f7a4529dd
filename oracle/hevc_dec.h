/* oracle/hevc_dec.h -- HEVC decoder restated from ITU-T H.265 (decoding process clauses 8.x,
 * parsing 9.3, syntax 7.3).  This is the CPU checker for what uvgComm's OpenHEVCFilter gets
 * from libOpenHevcDecode / libOpenHevcGetOutput
 * (/root/reference/src/media/processing/openhevcfilter.cpp:145-146,195-229).
 * Supported: Main profile 8-bit 4:2:0, I, P and B slices (both lists, bi-prediction, output reordering), all CB/TB sizes and
 * partitionings, temporal MVP, transform skip, sign data hiding, cu_qp_delta, transquant bypass, scaling lists, WPP/tiles
 * entry points, slices and dependent slice segments, deblocking, SAO.  Unsupported (returns < 0): weighted prediction,
 * PCM, long-term refs, reference list modification.  Test infrastructure. */
#ifndef ORC_HEVC_DEC_H
#define ORC_HEVC_DEC_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_decoder orc_decoder;
typedef struct {
  const pixel *plane[3]; int stride[3];
  int width, height;            /* cropped (conformance window) luma size */
  int coded_width, coded_height;
  int poc; int64_t pts;
  uint32_t fps_num, fps_den;    /* from VPS/VUI timing (time_scale / num_units_in_tick), 0 if absent */
  int slice_type;
} orc_dec_frame;

orc_decoder *orc_dec_open(void);
void orc_dec_close(orc_decoder *d);
/* One NAL unit per call, with or without its Annex-B start code.
 * Returns <0 on error/unsupported, 0 when no picture became available, 1 when one did. */
int orc_dec_decode_nal(orc_decoder *d, const uint8_t *data, size_t len, int64_t pts);
int orc_dec_get_frame(orc_decoder *d, orc_dec_frame *out);   /* 1 = filled, 0 = none; pictures come in OUTPUT order (C.5.2: by POC inside a coded video
                                                               * sequence, held back while up to sps_max_num_reorder_pics later ones may precede them), so
                                                               * one NAL unit may release several: call until 0 */
void orc_dec_flush(orc_decoder *d);                            /* end of stream: the pictures still held back become available */
/* decoded picture hash SEI messages (D.2.19: MD5, CRC or checksum, in a suffix SEI NAL unit behind the picture) met so far, and how many
 * of them did NOT match the picture as decoded here: a stream that carries them verifies itself */
void orc_dec_hash_stats(orc_decoder *d, int *checked, int *mismatch);
int orc_dec_concealed(const orc_decoder *d);      /* reference pictures that were missing (lost access units) and replaced by grey ones so far */
int orc_dec_debug_side(orc_decoder *d, int16_t *mv, int8_t *ref, uint8_t *pm, uint8_t *im, int8_t *qp);
/* debug: copy of the last picture before deblocking (same geometry as coded picture) */
const pixel *orc_dec_predeblock_plane(orc_decoder *d, int c);
#ifdef __cplusplus
}
#endif
#endif
