// kvazzup_amd/csrc/scaling_tables.h -- scaling lists on the host (H.265 7.3.4 / 7.4.5, Tables 7-5 / 7-6): the default lists and the per-position scaling
// factors m[x][y] of 8.6.4.2 in the layout the kernels index (dec_frame.h KVZ_SCALING_BYTES / scaling_offset).  Used by the decoder (a peer's stream with
// scaling_list_enabled_flag) and by the encoder (`scaling-list default`: uvgComm's checkbox, /root/reference/src/media/processing/kvazaarfilter.cpp:235-242).
#pragma once
#include <stdint.h>
#include <string.h>
#include "dec_frame.h"

namespace kvzx {

// Table 7-6: the default 8x8 lists (symmetric matrices, here in raster order); Table 7-5: 16 everywhere
static const uint8_t kScalingIntra8[64] = {16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29, 16, 16, 18, 21, 24, 27, 31, 36,
                                    17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115};
static const uint8_t kScalingInter8[64] = {16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28, 16, 17, 18, 20, 24, 25, 28, 33,
                                    17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91};
// the lists as coded: 8x8 rasters (4x4 for size 0) + the DC entries of the 16x16 / 32x32 lists
struct ScalingLists { uint8_t m[4][6][64]; uint8_t dc[2][6]; };
inline void scaling_default_one(ScalingLists &sl, int s, int m)
{
  if (s == 0) memset(sl.m[0][m], 16, 16);
  else memcpy(sl.m[s][m], (s == 3 ? m >= 1 : m >= 3) ? kScalingInter8 : kScalingIntra8, 64);
  if (s >= 2) sl.dc[s - 2][m] = 16;
}
inline ScalingLists scaling_defaults() { ScalingLists sl; memset(&sl, 16, sizeof(sl)); for (int s = 0; s < 4; s++) for (int m = 0; m < (s == 3 ? 2 : 6); m++) scaling_default_one(sl, s, m); return sl; }
// 7.4.5: the factors of every block size as rasters (scaling_offset): 16x16 / 32x32 repeat the 8x8 entries 2 x 2 / 4 x 4 times, their DC entries apart
inline void scaling_factors(const ScalingLists &sl, uint8_t *p /* KVZ_SCALING_BYTES */)
{
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < (s == 3 ? 2 : 6); m++) {
      const int n = 4 << s, rep = s == 0 ? 1 : n >> 3, src_n = s == 0 ? 4 : 8;
      uint8_t *q = p + scaling_offset(s + 2, s == 3 ? 0 : m % 3, s == 3 ? m : m / 3);
      for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) q[y * n + x] = sl.m[s][m][(y / rep) * src_n + x / rep];
      if (s >= 2) q[0] = sl.dc[s - 2][m];
    }
}

}  // namespace kvzx
