// kvazzup_amd/csrc/subpel_kernels.hip -- k_subpel: fractional-sample refinement of the motion vectors k_me found
// (kvazaar "subme" 1..4; what Kvazaar's presets above ultrafast add to the inter search).
//
// Statement of record: subme_refine() in oracle/hevc_enc.c ("uvgx subme v1").  Per searched coding unit: half-sample
// neighbours of the integer vector, then quarter-sample neighbours of the best so far; a candidate costs the SATD (8x8
// Hadamard sums) of source minus normative prediction plus lambda * vector bits.
//
// One workgroup per 32x32 block, two waves per 16x16 quadrant -- of the block's one 32x32 unit or of its own 16x16 unit.
// Everything a quadrant's seventeen candidates can touch is a 24 x 24 window of the reference (the eight taps reach -3 .. +4
// around positions at most one sample from the integer vector), staged in LDS once.  The eight candidates of a step and their
// centre differ in x by three values only, so a step runs the eight-tap HORIZONTAL pass three times (one 24 x 16 plane of
// intermediate values per x: four samples per v_dot4_i32_i8 on bytes biased by -128, the filters' taps add up to 64) and the
// VERTICAL pass once per candidate, straight into the lane layout of the matrix cores (lane (g, c) owns samples 4g .. 4g + 3
// of row c); the tile's four 8x8 Hadamard transforms are ONE pair of v_mfma_f32_16x16x16_f16 products with H16 = H8 (+) H8
// (exact: |difference| <= 255, |H d| <= 2040 < 2^11), the form k_intra_analyse uses.  The waves meet after each step's
// planes and after its candidates, where the quadrants of a 32x32 unit are added up.
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

namespace {

struct SubpelLds {
  alignas(16) uint8_t win[4][24 * 24 + 16];  // per quadrant: reference rows iy - 4 .. iy + 19, columns ix - 4 .. ix + 19 (+ slack for whole-dword reads)
  alignas(16) int16_t hp[4][3][24 * 16];     // per quadrant: the step's three horizontally filtered planes (x = centre - scale, centre, centre + scale), all 24 window rows
  uint32_t satd[4][9];                       // per quadrant: SATD of the centre (0) and of the step's candidates (1 .. 8)
};

// SATD of a 16x16 tile: sum over its four 8x8 blocks of (sum |H d H^T| + 2) >> 2; d[r] = difference at (x = 4g + r, y = c)
// this lane's entries of H16 = H8 (+) H8: H16[c][4g + r], zero across the two 8x8 blocks, else (-1)^popcount(c & j & 7)
__device__ __forceinline__ kv_f16x4 hadamard_operand(int lane)
{
  const int g = lane >> 4, c = lane & 15;
  kv_f16x4 h;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = 4 * g + r;
    h[r] = ((c ^ j) & 8) ? (_Float16)0.f : ((__builtin_popcount((unsigned)(c & j & 7)) & 1) ? (_Float16)-1.f : (_Float16)1.f);
  }
  return h;
}
__device__ __forceinline__ uint32_t tile_satd(const int (&d)[4], kv_f16x4 h)
{
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

// One horizontally filtered plane: hp[r * 16 + c] = sum_k f[k] * win[r][ox + c + k] for the 24 window rows, x component mvx of the
// candidates it serves; (ix) = integer vector the window is centred on.  With the {0,0,0,64,..} filter at fraction 0 this is
// 64 * sample, and the vertical pass's shift by 6 then reproduces every case of 8.5.3.3.3.1 exactly.
// the four luma filters, taps as signed bytes: {taps 0..3, taps 4..7} of fraction q at [2q], [2q + 1]
struct Filters { uint32_t w[8]; };
__device__ __forceinline__ Filters load_filters()
{
  Filters F;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { lo |= (uint32_t)(uint8_t)kLumaFilter[q][k] << (8 * k); hi |= (uint32_t)(uint8_t)kLumaFilter[q][4 + k] << (8 * k); }
    F.w[2 * q] = lo; F.w[2 * q + 1] = hi;
  }
  return F;
}
__device__ __forceinline__ uint32_t pick(const Filters &F, int q, int half)       // (selects, not an indexed load: the table stays in registers)
{
  const uint32_t a = half ? F.w[1] : F.w[0], b = half ? F.w[3] : F.w[2], c = half ? F.w[5] : F.w[4], d = half ? F.w[7] : F.w[6];
  return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}
__device__ __forceinline__ void hplane(const uint8_t *win, int16_t *hp, int mvx, int ix, int lane, const Filters &F)
{
  const int xf = mvx & 3, ox = (mvx >> 2) - ix + 1;
  const uint32_t flo = pick(F, xf, 0), fhi = pick(F, xf, 1);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int grp = it * 64 + lane;                          // (row, group of four columns): 24 x 4 = 96
    if (grp < 96) {
      const int row = grp >> 2, c0 = (grp & 3) * 4, a = row * 24 + ox + c0;      // first byte of the eleven the four outputs read
      const uint32_t *q = (const uint32_t *)(win + (a & ~3));
      const uint32_t sh = (uint32_t)(a & 3);
      const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = (a & 3) ? q[3] : 0u;  // (row 23's last group ends with the window)
      const uint32_t s0 = __builtin_amdgcn_alignbyte(w1, w0, sh) ^ 0x80808080u, s1 = __builtin_amdgcn_alignbyte(w2, w1, sh) ^ 0x80808080u,
                     s2 = __builtin_amdgcn_alignbyte(w3, w2, sh) ^ 0x80808080u;
      int h[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t lo = r ? __builtin_amdgcn_alignbyte(s1, s0, (uint32_t)r) : s0, hi = r ? __builtin_amdgcn_alignbyte(s2, s1, (uint32_t)r) : s1;
        h[r] = __builtin_amdgcn_sdot4((int)flo, (int)lo, __builtin_amdgcn_sdot4((int)fhi, (int)hi, 8192, false), false);
      }
      *(uint2 *)&hp[row * 16 + c0] = make_uint2(pack_i16(h[0], h[1]), pack_i16(h[2], h[3]));
    }
  }
}
// SATD between the quadrant's source samples (s4: this lane's four) and its prediction from plane hp with vertical component mvy
__device__ __forceinline__ uint32_t price(const int16_t *hp, uint32_t s4, int mvy, int iy, int lane, const Filters &F, kv_f16x4 hm)
{
  const int yf = mvy & 3, oy = (mvy >> 2) - iy + 1, g = lane >> 4, c = lane & 15;
  const uint32_t flo = pick(F, yf, 0), fhi = pick(F, yf, 1);
  int v[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint2 t = *(const uint2 *)&hp[(oy + c + j) * 16 + 4 * g];
    const int f = (int)(int8_t)(((j < 4 ? flo : fhi) >> (8 * (j & 3))) & 0xffu);
    v[0] += f * (int)(int16_t)(t.x & 0xffffu); v[1] += f * ((int)t.x >> 16); v[2] += f * (int)(int16_t)(t.y & 0xffffu); v[3] += f * ((int)t.y >> 16);
  }
  int d[4];
#pragma unroll
  for (int r = 0; r < 4; r++) d[r] = (int)((s4 >> (8 * r)) & 255u) - clip8(((v[r] >> 6) + 32) >> 6);
  return tile_satd(d, hm);
}

// mvd_bits() of hevc_core.h in closed form: 1, 3, then 2 * floor(log2 |q|) + 3 (prefix "11", first-order Exp-Golomb remainder, sign)
__device__ __forceinline__ int mvd_bits_fast(int q) { const int a = iabs(q); return a == 0 ? 1 : (a == 1 ? 3 : 2 * (31 - __builtin_clz((unsigned)a)) + 3); }

// subme_allowed() of oracle/hevc_enc.c
__device__ __forceinline__ bool allowed(const EncFrame &f, int x0, int y0, int n, int mvx, int mvy, int ty0, int ty1, int tx0, int tx1)
{
  const int ix = mvx >> 2, iy = mvy >> 2;
  int mx = (mvx & 7) ? 4 : 0, my = (mvy & 7) ? 4 : 0;
  if ((ty0 > 0 && y0 + iy - my < ty0) || (ty1 < f.ch && y0 + iy + n + my > ty1)) return false;
  if ((tx0 > 0 && x0 + ix - mx < tx0) || (tx1 < f.cw && x0 + ix + n + mx > tx1)) return false;
  if (f.mv_frame) {
    if (f.mv_frame == 1) { mx = (mvx & 3) ? 4 : 0; my = (mvy & 3) ? 4 : 0; }
    if (x0 + ix - mx < 0 || x0 + ix + n + mx > f.cw || y0 + iy - my < 0 || y0 + iy + n + my > f.ch) return false;
  }
  return true;
}

}  // namespace

// 512 threads: two waves per quadrant.  Wave (quadrant, 0) prices the centre and the step's horizontal / vertical four, wave
// (quadrant, 1) its diagonal four; the three horizontally filtered planes of the quadrant are shared (part 0 builds the first
// two, part 1 the third).  Within a wave the four candidates are priced in one unrolled block, so that their dependent chains
// (LDS reads, multiply-adds, conversions, the two matrix products) interleave: a wave alone on its SIMD has nothing else to
// cover latencies with.
__global__ __launch_bounds__(512) void k_subpel(EncFrame f)
{
  __shared__ SubpelLds s;
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, part = tid >> 8;
  int bx_, by_; xcd_block_2d(bx_, by_);
  const int x0 = bx_ * 32, y0 = by_ * 32 + f.row0 * 64;
  const int bi0 = b8idx(f, x0, y0);
  // me-source: the search ran ahead on the input stream, so this launch is the first of the picture's chain and carries its head as k_me does otherwise
  // (nothing here reads what it writes -- rate control state, the CTUs' target QPs: k_inter_recon is the first)
  if (f.pb_on && blockIdx.x == 0 && blockIdx.y == 0) picture_begin_body(f.pb_rc, f.pb_bits3, f.pb_slot3, f.pb_have3, f.pb_qt, f.pb_roi, f.pb_nctu, f.qp, 0, tid, 512);
  if (!f.cu_mvp_idx[bi0]) return;                           // k_me's mark: the block was not searched (me-early-termination)
  // KVAZZUP_AMD_INTRA_TRACE (tools/subpel_timeline.py): eight 100 MHz stamps per block
  unsigned long long *tr = f.trace ? f.trace + (size_t)((y0 >> 5) * (f.cw >> 5) + (x0 >> 5)) * 8 : nullptr;
#define SP_STAMP(k) do { if (tr && tid == 0) tr[k] = wall_clock64(); } while (0)
  SP_STAMP(0);
  const bool split = f.cu_log2[bi0] != 5;                   // (four 16x16 units; with intra-in-P some of them may be intra units, whose waves only keep the barriers company)
  const int X = x0 + (w & 1) * 16, Y = y0 + (w >> 1) * 16;  // this wave's quadrant
  const int bq = b8idx(f, X, Y);
  const int mvx0 = f.cu_mv[bq * 2], mvy0 = f.cu_mv[bq * 2 + 1], ix = mvx0 >> 2, iy = mvy0 >> 2;    // integer vector (multiples of 4)
  const int ux = split ? X : x0, uy = split ? Y : y0, un = split ? 16 : 32;                          // the coding unit the quadrant belongs to
  int ty0 = 0, ty1 = f.ch;
  if (f.tile_rows > 1) {
    const int hc = f.ch >> 6, tr = tile_row_of(hc, f.tile_rows, y0 >> 6);
    ty0 = tile_row_first(hc, f.tile_rows, tr) * 64; ty1 = tile_row_first(hc, f.tile_rows, tr + 1) * 64;
  }
  int tx0 = 0, tx1 = f.cw;
  if (f.tile_cols > 1) {
    const int wc = f.cw >> 6, tc = tile_col_of(wc, f.tile_cols, x0 >> 6);
    tx0 = tile_col_first(wc, f.tile_cols, tc) * 64; tx1 = tile_col_first(wc, f.tile_cols, tc + 1) * 64;
  }
  uint8_t *win = s.win[w];
  {
    // the 24 x 24 window (both waves of the quadrant load half of it): six dwords per row when it lies inside the picture (any
    // alignment), else sample by sample with the coordinates clamped (8.5.3.3.3.1)
    const int wx0 = X + ix - 4, wy0 = Y + iy - 4, l2 = part * 64 + lane;
    if (wx0 >= 0 && wx0 + 24 <= f.cw && wy0 >= 0 && wy0 + 24 <= f.ch) {
      for (int i = l2; i < 144; i += 128) {
        const int wy = i / 6, k = i - wy * 6;
        const uint8_t *p = f.ref[0] + (size_t)(wy0 + wy) * f.cw + wx0 + 4 * k;
        const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
        *(uint32_t *)&win[wy * 24 + 4 * k] = sh ? __builtin_amdgcn_alignbyte(q[1], q[0], sh) : q[0];
      }
    } else {
      for (int i = l2; i < 24 * 24; i += 128) {
        const int wy = i / 24, wx = i - wy * 24;
        win[i] = f.ref[0][(size_t)clip3(0, f.ch - 1, wy0 + wy) * f.cw + clip3(0, f.cw - 1, wx0 + wx)];
      }
    }
  }
  const uint32_t s4 = *(const uint32_t *)&f.src[0][(size_t)(Y + (lane & 15)) * f.cw + X + 4 * (lane >> 4)];
  const uint32_t lam = (uint32_t)f.lambda_q4;
  const Filters F = load_filters();
  const kv_f16x4 hm = hadamard_operand(lane);
  const int offx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, offy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
  int cx = mvx0, cy = mvy0;
  uint32_t best = 0;
  __syncthreads();
  SP_STAMP(1);
#pragma unroll 1
  for (int step = 0; step < 2; step++) {
    const int scale = step ? 1 : 2;
    const int ncand = f.subme >= (step ? 4 : 2) ? 8 : (f.subme >= (step ? 3 : 1) ? 4 : 0);
    if (ncand) {
      if (part == 0) { hplane(win, s.hp[w][0], cx - scale, ix, lane, F); hplane(win, s.hp[w][1], cx, ix, lane, F); }
      else hplane(win, s.hp[w][2], cx + scale, ix, lane, F);
    }
    __syncthreads();
    SP_STAMP(2 + step * 3);
    if (ncand) {
      if (part == 0) {
        uint32_t v[5];
        v[0] = step == 0 ? price(s.hp[w][1], s4, cy, iy, lane, F, hm) : 0u;
        v[1] = price(s.hp[w][0], s4, cy, iy, lane, F, hm);                       // (-1, 0)
        v[2] = price(s.hp[w][2], s4, cy, iy, lane, F, hm);                       // (+1, 0)
        v[3] = price(s.hp[w][1], s4, cy - scale, iy, lane, F, hm);               // (0, -1)
        v[4] = price(s.hp[w][1], s4, cy + scale, iy, lane, F, hm);               // (0, +1)
        if (lane < 5 && (lane || step == 0)) s.satd[w][lane] = lane == 0 ? v[0] : (lane == 1 ? v[1] : (lane == 2 ? v[2] : (lane == 3 ? v[3] : v[4])));
      } else if (ncand == 8) {
        uint32_t v[4];
        v[0] = price(s.hp[w][0], s4, cy - scale, iy, lane, F, hm);               // (-1, -1)
        v[1] = price(s.hp[w][2], s4, cy - scale, iy, lane, F, hm);               // (+1, -1)
        v[2] = price(s.hp[w][0], s4, cy + scale, iy, lane, F, hm);               // (-1, +1)
        v[3] = price(s.hp[w][2], s4, cy + scale, iy, lane, F, hm);               // (+1, +1)
        if (lane < 4) s.satd[w][5 + lane] = lane == 0 ? v[0] : (lane == 1 ? v[1] : (lane == 2 ? v[2] : v[3]));
      }
    }
    __syncthreads();
    SP_STAMP(3 + step * 3);
    // the decision, by every wave for its own unit (the waves of a unit compute the same thing): lane k prices candidate k (lane 0 the
    // centre), the least key wins
    auto unit_satd = [&](int k) { return split ? s.satd[w][k] : s.satd[0][k] + s.satd[1][k] + s.satd[2][k] + s.satd[3][k]; };
    uint32_t key = 0xffffffffu;
    if (lane <= ncand && ncand) {
      const int k = lane - 1, mx = k < 0 ? cx : cx + offx[k & 7] * scale, my = k < 0 ? cy : cy + offy[k & 7] * scale;
      if (k < 0) key = step == 0 ? (unit_satd(0) + ((lam * (uint32_t)(mvd_bits_fast(cx) + mvd_bits_fast(cy))) >> 4)) << 4 : best;
      else if (allowed(f, ux, uy, un, mx, my, ty0, ty1, tx0, tx1)) key = ((unit_satd(lane) + ((lam * (uint32_t)(mvd_bits_fast(mx) + mvd_bits_fast(my))) >> 4)) << 4) | (uint32_t)lane;
    }
    // minimum over lanes 0 .. 8 (two rows of the DPP network: lanes 0-7 by quad / half-row steps, lane 8 read directly)
    uint32_t m = key;
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)m, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)m, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)m, 0x141, 0xf, 0xf, false));    // row_half_mirror: lanes 0-7 hold their minimum
    const uint32_t b = ncand ? min((uint32_t)__builtin_amdgcn_readlane((int)m, 0), (uint32_t)__builtin_amdgcn_readlane((int)key, 8)) : best;
    if (b & 15u) { cx += offx[(b & 15u) - 1] * scale; cy += offy[(b & 15u) - 1] * scale; }
    best = b & ~15u;
    __syncthreads();                                        // (s.satd and the planes are rewritten by the next step)
    SP_STAMP(4 + step * 3);
  }
  if (part == 0 && lane < 4 && !(split && f.cu_intra[bq])) {
    const int i = b8idx(f, X + (lane & 1) * 8, Y + (lane >> 1) * 8);
    f.cu_mv[i * 2] = (int16_t)cx; f.cu_mv[i * 2 + 1] = (int16_t)cy;
  }
}

void launch_subpel(const EncFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_subpel, dim3(f.cw / 32, band_rows(f) * 2), dim3(512), 0, st, f); }

}  // namespace kvzx
