/* oracle/synth.h -- "uvgx-synth-v1" integer-only synthetic clips.
 * Test infrastructure (bench.py has the numpy twin). */
#ifndef ORC_SYNTH_H
#define ORC_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* kind: 0 moving objects, 1 flat (128), 2 noise.  Writes packed I420 (w*h*3/2 bytes). */
void orc_synth_frame(int kind, uint32_t seed, int w, int h, int t, uint8_t *out);
#ifdef __cplusplus
}
#endif
#endif
