/* oracle/hevc_hash.h -- decoded picture hash (H.265 D.2.19 / D.3.19: SEI payload 132): MD5 (RFC 1321), CRC and checksum of the three
 * colour components of a decoded picture, 8-bit samples.  Test infrastructure. */
#ifndef ORC_HEVC_HASH_H
#define ORC_HEVC_HASH_H
#include "hevc_common.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { uint32_t a, b, c, d; uint64_t nbytes; uint8_t buf[64]; int fill; } orc_md5;
void orc_md5_init(orc_md5 *m);
void orc_md5_update(orc_md5 *m, const uint8_t *p, size_t n);
void orc_md5_final(orc_md5 *m, uint8_t out[16]);
/* hash_type 0 = MD5 (16 bytes per component), 1 = CRC (2 bytes), 2 = checksum (4 bytes): out[c] gets the bytes as they stand in the SEI */
void orc_picture_hash(int hash_type, const pixel *const plane[3], const int stride[3], int w, int h, uint8_t out[3][16]);
static inline int orc_hash_bytes(int hash_type) { return hash_type == 0 ? 16 : (hash_type == 1 ? 2 : 4); }
#ifdef __cplusplus
}
#endif
#endif
