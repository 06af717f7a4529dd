// kvazzup_amd/csrc/subpel_kernels.hip -- k_subpel: fractional-sample refinement of the motion vectors k_me found
// (kvazaar "subme" 1..4; what Kvazaar's presets above ultrafast add to the inter search).
//
// Statement of record: subme_refine() in oracle/hevc_enc.c ("uvgx subme v1").  Per searched coding unit: half-sample
// neighbours of the integer vector, then quarter-sample neighbours of the best so far; a candidate costs the SATD (8x8
// Hadamard sums) of source minus normative prediction plus lambda * vector bits.
//
// One workgroup per 32x32 block, one wave per 16x16 quadrant -- of the block's one 32x32 unit or of its own 16x16 unit.
// Everything a quadrant's seventeen candidates can touch is a 24 x 24 window of the reference (the eight taps reach -3 .. +4
// around positions at most one sample from the integer vector), staged in LDS once.  A candidate is priced in three steps:
// eight-tap horizontal pass (23 x 16 intermediate values), eight-tap vertical pass into the lane layout of the matrix cores
// (lane (g, c) owns samples 4g .. 4g + 3 of row c), and the tile's four 8x8 Hadamard transforms as ONE pair of
// v_mfma_f32_16x16x16_f16 products with H16 = H8 (+) H8 (exact: |difference| <= 255, |H d| <= 2040 < 2^11), the form
// k_intra_analyse uses.  The waves meet twice, after each step's candidates, to add up the quadrants of a 32x32 unit.
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

namespace {

struct SubpelLds {
  alignas(16) uint8_t win[4][24 * 24];       // per quadrant: reference rows iy - 4 .. iy + 19, columns ix - 4 .. ix + 19
  alignas(16) int16_t tmp[4][23 * 16];       // horizontally filtered rows of the candidate being priced
  uint32_t satd[4][9];                       // per quadrant: SATD of the centre (0) and of the step's candidates (1 .. 8)
};

// SATD of a 16x16 tile: sum over its four 8x8 blocks of (sum |H d H^T| + 2) >> 2; d[r] = difference at (x = 4g + r, y = c)
__device__ __forceinline__ uint32_t tile_satd(const int (&d)[4], int lane)
{
  const int g = lane >> 4, c = lane & 15;
  kv_f16x4 h;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = 4 * g + r;
    h[r] = ((c ^ j) & 8) ? (_Float16)0.f : ((__builtin_popcount((unsigned)(c & j & 7)) & 1) ? (_Float16)-1.f : (_Float16)1.f);
  }
  int y[4];
  mfma16_data_a(d, h, y);
  kv_f16x4 yb;
#pragma unroll
  for (int r = 0; r < 4; r++) yb[r] = (_Float16)(short)y[r];
  const kv_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const kv_f32x4 zf = __builtin_amdgcn_mfma_f32_16x16x16f16(h, yb, zero, 0, 0, 0);
  uint32_t a = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) a += (uint32_t)iabs((int)zf[r]);
  a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, false);
  a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, false);
  a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, false);
  // output (u = 4g + r, v = c) lies in 8x8 block (u >= 8) * 2 + (v >= 8): lanes 0-7 + 16-23, 8-15 + 24-31, 32-39 + 48-55, 40-47 + 56-63
  const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 0) + (uint32_t)__builtin_amdgcn_readlane((int)a, 16);
  const uint32_t q1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 8) + (uint32_t)__builtin_amdgcn_readlane((int)a, 24);
  const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 32) + (uint32_t)__builtin_amdgcn_readlane((int)a, 48);
  const uint32_t q3 = (uint32_t)__builtin_amdgcn_readlane((int)a, 40) + (uint32_t)__builtin_amdgcn_readlane((int)a, 56);
  return ((q0 + 2) >> 2) + ((q1 + 2) >> 2) + ((q2 + 2) >> 2) + ((q3 + 2) >> 2);
}

// SATD between the quadrant's source samples (s4: this lane's four) and its prediction with vector (mvx, mvy); (ix, iy) = integer
// vector the window is centred on.  8.5.3.3.3.1 as one separable form: the horizontal pass with the {0,0,0,64,..} filter at fraction
// 0 leaves 64 * sample, the vertical pass shifts by 6 -- which reproduces every case of the standard's table exactly.
__device__ __forceinline__ uint32_t price(const uint8_t *win, int16_t *tmp, uint32_t s4, int mvx, int mvy, int ix, int iy, int lane)
{
  const int xf = mvx & 3, yf = mvy & 3, ox = (mvx >> 2) - ix + 1, oy = (mvy >> 2) - iy + 1;
  int fx[8], fy[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { fx[k] = kLumaFilter[xf][k]; fy[k] = kLumaFilter[yf][k]; }
  for (int i = lane; i < 23 * 16; i += 64) {
    const int row = i >> 4, col = i & 15;
    const uint8_t *p = win + (oy + row) * 24 + ox + col;
    int h = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) h += fx[k] * (int)p[k];
    tmp[i] = (int16_t)h;
  }
  wave_sync();
  const int g = lane >> 4, c = lane & 15;
  int d[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int16_t *t = tmp + c * 16 + 4 * g + r;
    int v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v += fy[j] * (int)t[j * 16];
    d[r] = (int)((s4 >> (8 * r)) & 255u) - clip8(((v >> 6) + 32) >> 6);
  }
  wave_sync();                                              // (tmp is rewritten by the next candidate)
  return tile_satd(d, lane);
}

// subme_allowed() of oracle/hevc_enc.c
__device__ __forceinline__ bool allowed(const EncFrame &f, int x0, int y0, int n, int mvx, int mvy, int ty0, int ty1)
{
  const int ix = mvx >> 2, iy = mvy >> 2;
  int mx = (mvx & 7) ? 4 : 0, my = (mvy & 7) ? 4 : 0;
  if ((ty0 > 0 && y0 + iy - my < ty0) || (ty1 < f.ch && y0 + iy + n + my > ty1)) return false;
  if (f.mv_frame) {
    if (f.mv_frame == 1) { mx = (mvx & 3) ? 4 : 0; my = (mvy & 3) ? 4 : 0; }
    if (x0 + ix - mx < 0 || x0 + ix + n + mx > f.cw || y0 + iy - my < 0 || y0 + iy + n + my > f.ch) return false;
  }
  return true;
}

}  // namespace

__global__ __launch_bounds__(256) void k_subpel(EncFrame f)
{
  __shared__ SubpelLds s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int bx_, by_; xcd_block_2d(bx_, by_);
  const int x0 = bx_ * 32, y0 = by_ * 32 + f.row0 * 64;
  const int bi0 = b8idx(f, x0, y0);
  if (!f.cu_mvp_idx[bi0]) return;                           // k_me's mark: the block was not searched (me-early-termination)
  const bool split = f.cu_log2[bi0] == 4;
  const int X = x0 + (w & 1) * 16, Y = y0 + (w >> 1) * 16;  // this wave's quadrant
  const int bq = b8idx(f, X, Y);
  const int mvx0 = f.cu_mv[bq * 2], mvy0 = f.cu_mv[bq * 2 + 1], ix = mvx0 >> 2, iy = mvy0 >> 2;    // integer vector (multiples of 4)
  const int ux = split ? X : x0, uy = split ? Y : y0, un = split ? 16 : 32;                          // the coding unit the quadrant belongs to
  int ty0 = 0, ty1 = f.ch;
  if (f.tile_rows > 1) {
    const int hc = f.ch >> 6, tr = tile_row_of(hc, f.tile_rows, y0 >> 6);
    ty0 = tile_row_first(hc, f.tile_rows, tr) * 64; ty1 = tile_row_first(hc, f.tile_rows, tr + 1) * 64;
  }
  uint8_t *win = s.win[w];
  for (int i = lane; i < 24 * 24; i += 64) {
    const int wy = i / 24, wx = i - wy * 24;
    win[i] = f.ref[0][(size_t)clip3(0, f.ch - 1, Y + iy - 4 + wy) * f.cw + clip3(0, f.cw - 1, X + ix - 4 + wx)];
  }
  const uint32_t s4 = *(const uint32_t *)&f.src[0][(size_t)(Y + (lane & 15)) * f.cw + X + 4 * (lane >> 4)];
  wave_sync();
  const uint32_t lam = (uint32_t)f.lambda_q4;
  const int offx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, offy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
  int cx = mvx0, cy = mvy0;
  uint32_t best = 0;
#pragma unroll 1
  for (int step = 0; step < 2; step++) {
    const int scale = step ? 1 : 2;
    const int ncand = f.subme >= (step ? 4 : 2) ? 8 : (f.subme >= (step ? 3 : 1) ? 4 : 0);
    if (step == 0) { const uint32_t v = price(win, s.tmp[w], s4, cx, cy, ix, iy, lane); if (lane == 0) s.satd[w][0] = v; }
#pragma unroll 1
    for (int k = 0; k < ncand; k++) {
      const uint32_t v = price(win, s.tmp[w], s4, cx + offx[k] * scale, cy + offy[k] * scale, ix, iy, lane);
      if (lane == 0) s.satd[w][k + 1] = v;
    }
    __syncthreads();
    // the decision, by every wave for its own unit (the four waves of a 32x32 unit compute the same thing)
    auto unit_satd = [&](int k) { return split ? s.satd[w][k] : s.satd[0][k] + s.satd[1][k] + s.satd[2][k] + s.satd[3][k]; };
    if (step == 0) best = (unit_satd(0) + ((lam * (uint32_t)(mvd_bits(cx) + mvd_bits(cy))) >> 4)) << 4;
    uint32_t b = best & ~15u;
    for (int k = 0; k < ncand; k++) {
      const int mx = cx + offx[k] * scale, my = cy + offy[k] * scale;
      if (!allowed(f, ux, uy, un, mx, my, ty0, ty1)) continue;
      const uint32_t key = ((unit_satd(k + 1) + ((lam * (uint32_t)(mvd_bits(mx) + mvd_bits(my))) >> 4)) << 4) | (uint32_t)(k + 1);
      if (key < b) b = key;
    }
    if (b & 15u) { cx += offx[(b & 15u) - 1] * scale; cy += offy[(b & 15u) - 1] * scale; }
    best = b & ~15u;
    __syncthreads();                                        // (s.satd is rewritten by the next step)
  }
  if (lane < 4) {
    const int i = b8idx(f, X + (lane & 1) * 8, Y + (lane >> 1) * 8);
    f.cu_mv[i * 2] = (int16_t)cx; f.cu_mv[i * 2 + 1] = (int16_t)cy;
  }
}

void launch_subpel(const EncFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_subpel, dim3(f.cw / 32, band_rows(f) * 2), dim3(256), 0, st, f); }

}  // namespace kvzx
