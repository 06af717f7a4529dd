// kvazzup_amd/csrc/dec_kernels.hip -- CDNA4 (gfx950) kernels of the HEVC decoder hot path: what OpenHEVC does inside
// libOpenHevcDecode (/root/reference/src/media/processing/openhevcfilter.cpp:145-146) after entropy decoding, for ANY
// Main-profile I / P picture (dec_frame.h).  The arithmetic is checked against oracle/hevc_dec.c (tests/).
//
//   k_dec_inter    one workgroup per 32x32 luma region: motion compensation of its inter blocks at 4x4 granularity (integer
//                  vectors: aligned dwords + v_alignbyte; fractional: separable 8-/4-tap through LDS windows per 8x8 cell),
//                  then every inter transform block of the region -- level words scattered straight into LDS, dequantised,
//                  inverse DCT by size class (4x4 .. 16x16: int16 v_dot2; 32x32: v_mfma_i32_16x16x64_i8), reconstruction
//   k_dec_intra    one wave per (CTU, colour plane): the CTU's intra transform blocks in decoding order (prediction from the
//                  neighbours, residual, reconstruction); CTUs are coupled by progress counters with 8x8 granularity, so a CTU
//                  starts as soon as the part of its left / upper neighbours it reads is final, not when they are complete
//   k_dec_deblock  one workgroup per 64x64 tile shifted by (-4, -4): boundary strengths from the 4x4 records, vertical then
//                  horizontal edges in LDS
//   k_dec_sao      one workgroup per CTU
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "dec_frame.h"
#include "dec_kernels.h"
#include "kernel_common.h"

namespace kvzx {

// A kernel body runs on a frame descriptor that is either the kernel's argument (DecFrame) or lies in device memory and is read through the
// constant address space (CDecFrame: the batched kernels, see the end of this file); Wg = the workgroup's index inside its frame's share of
// the grid and the number of workgroups of that share which have work.
#define KVZ_CONST_AS __attribute__((address_space(4)))
typedef const KVZ_CONST_AS DecFrame CDecFrame;
struct Wg { int id, n; };

// level -> coefficient of one transform block (8.6.4.2): flat scaling, the scaling factor of the position (scaling lists), or -- coding units with
// cu_transquant_bypass_flag -- the level itself, which then IS the residual sample (8.6.2)
template <class F> __device__ __forceinline__ int dec_dequant(const F &f, const DecTu &d, int pos, int level)
{
  if (d.flags & TU_BYPASS) return level;
  if (f.scaling) return dequant_coef_m(level, d.qp, d.log2, f.scaling[scaling_offset(d.log2, d.plane, (d.flags & TU_INTRA) ? 0 : 1) + pos]);
  return dequant_coef(level, d.qp, d.log2);
}

__device__ __forceinline__ int zorder3(int x, int y)      // 3 + 3 bits
{
  int z = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) z |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1);
  return z;
}
__device__ __forceinline__ void unzorder3(int z, int &x, int &y)
{
  x = 0; y = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) { x |= ((z >> (2 * b)) & 1) << b; y |= ((z >> (2 * b + 1)) & 1) << b; }
}

// ---- motion compensation primitives (8.5.3.3.3): `pitch` = allocated row length, (w, h) = picture size for the padding clamp
__device__ __forceinline__ int refpx(const uint8_t *p, int pitch, int w, int h, int x, int y)
{
  return p[(size_t)clip3(0, h - 1, y) * pitch + clip3(0, w - 1, x)];
}
__device__ __forceinline__ int mc_luma_px(const uint8_t *p, int pitch, int w, int h, int x, int y, int mvx, int mvy)
{
  const int xf = mvx & 3, yf = mvy & 3, xi = x + (mvx >> 2), yi = y + (mvy >> 2);
  int v;
  if (!xf && !yf) return refpx(p, pitch, w, h, xi, yi);
  if (!yf) { v = 0; for (int i = 0; i < 8; i++) v += kLumaFilter[xf][i] * refpx(p, pitch, w, h, xi + i - 3, yi); }
  else if (!xf) { v = 0; for (int i = 0; i < 8; i++) v += kLumaFilter[yf][i] * refpx(p, pitch, w, h, xi, yi + i - 3); }
  else {
    v = 0;
    for (int j = 0; j < 8; j++) {
      int t = 0;
      for (int i = 0; i < 8; i++) t += kLumaFilter[xf][i] * refpx(p, pitch, w, h, xi + i - 3, yi + j - 3);
      v += kLumaFilter[yf][j] * t;
    }
    v >>= 6;
  }
  return clip8((v + 32) >> 6);
}
// the same sample at the 14-bit precision of 8.5.3.3.3, before the weighted sample prediction: what bi-prediction averages (8.5.3.3.4.2)
__device__ __forceinline__ int mc_luma_14(const uint8_t *p, int pitch, int w, int h, int x, int y, int mvx, int mvy)
{
  const int xf = mvx & 3, yf = mvy & 3, xi = x + (mvx >> 2), yi = y + (mvy >> 2);
  int v = 0;
  if (!xf && !yf) return refpx(p, pitch, w, h, xi, yi) << 6;
  if (!yf) { for (int i = 0; i < 8; i++) v += kLumaFilter[xf][i] * refpx(p, pitch, w, h, xi + i - 3, yi); return v; }
  if (!xf) { for (int i = 0; i < 8; i++) v += kLumaFilter[yf][i] * refpx(p, pitch, w, h, xi, yi + i - 3); return v; }
  for (int j = 0; j < 8; j++) {
    int t = 0;
    for (int i = 0; i < 8; i++) t += kLumaFilter[xf][i] * refpx(p, pitch, w, h, xi + i - 3, yi + j - 3);
    v += kLumaFilter[yf][j] * t;
  }
  return v >> 6;
}
// ... and a chroma sample (vector in 1/8 samples; the separable form with the {0, 64, 0, 0} filter at fraction 0 is exact for every case)
__device__ __forceinline__ int mc_chroma_14(const uint8_t *p, int pitch, int w, int h, int x, int y, int mvx, int mvy)
{
  const int xf = mvx & 7, yf = mvy & 7, xi = x + (mvx >> 3) - 1, yi = y + (mvy >> 3) - 1;
  int v = 0;
  for (int j = 0; j < 4; j++) {
    int t = 0;
    for (int i = 0; i < 4; i++) t += kChromaFilter[xf][i] * refpx(p, pitch, w, h, xi + i, yi + j);
    v += kChromaFilter[yf][j] * t;
  }
  return v >> 6;
}
// four luma samples (x .. x + 3, y) with one INTEGER vector, packed little-endian
__device__ __forceinline__ uint32_t mc_luma4_int(const uint8_t *p, int pitch, int w, int h, int x, int y, int mvx, int mvy)
{
  const int xi = x + (mvx >> 2), yi = clip3(0, h - 1, y + (mvy >> 2));
  const uint8_t *row = p + (size_t)yi * pitch;
  if (xi >= 0 && xi + 3 < w) {
    const uint32_t *q = (const uint32_t *)(row + (xi & ~3));                 // planes are 4-byte aligned, pitch a multiple of 64, w of 8
    const uint32_t lo = q[0], hi = (xi & 3) ? q[1] : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(xi & 3));
  }
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) v |= (uint32_t)row[clip3(0, w - 1, xi + i)] << (8 * i);
  return v;
}

// =============================================================================================
// Inter prediction + residual of one 32x32 luma region
// =============================================================================================
#define DEC_MAX_TU 96                 // transform blocks of one 32x32 region: 64 luma 4x4 + 32 chroma 4x4
struct DecInterLds {
  // coefficients of every coded inter block of the region, each block contiguous and TRANSPOSED ([column][row]) at 16 x the
  // z-order index of its first 4x4 unit: luma units 0..63 at [0, 1024), Cb 0..15 at [1024, 1280), Cr at [1280, 1536)
  alignas(16) int16_t C[1536], B[1024];
  alignas(16) int16_t M[2][KV_MATRIX_ENTRIES];
  alignas(16) int8_t M8[32 * 32]; int rowsum[32];        // transposed 32-point matrix as int8 + its row sums (MFMA operands)
  alignas(16) uint8_t px[1024 + 512];                    // prediction, then reconstruction: luma 32x32 raster; Cb, Cr 16x16 each
  B4Rec recs[64];
  B4L1 recx[64];                                         // the second vectors of the region's bi-predicted blocks (B4_BI; pictures with f.b4x)
  uint8_t cflag[16];                                     // per 8x8 cell: bit0 inter, bit1 one motion for the whole cell, bit2 fractional luma vector
  alignas(16) uint8_t lwin[16][15 * 16];                 // fractional luma: per cell the 15 x 15 reference window ...
  int ltmp[16][15 * 8];                                  // ... and its horizontally filtered rows (32-bit: see enc_kernels.hip InterLds)
  alignas(4) uint8_t cwin[2][16][7 * 8];                 // chroma: per plane and cell the 7 x 7 window of its 4 x 4 samples
  DecTu td[DEC_MAX_TU]; uint32_t tstart[DEC_MAX_TU + 1];
  uint32_t mask[2][4][2];                                // [luma, chroma][log2 - 2][word]: coded blocks of the class, bit = block index
};

// inverse transform of the blocks of one size class whose bit is set in (mlo, mhi): block t at C0 + t * N * N
template <int L2, int OPL, class PX>
__device__ __forceinline__ void dec_itx_class(DecInterLds &s, int16_t *C0, uint32_t mlo, uint32_t mhi, PX px_index, int tid)
{
  constexpr int N = 1 << L2, G = XF<L2, OPL>::G, LPT = XF<L2, OPL>::LANES;
  const int tu = tid / LPT, l = tid % LPT, rp = l / G, g = l % G;
  const bool has = ((tu < 32 ? mlo : mhi) >> (tu & 31)) & 1;
  int16_t *A = C0 + tu * N * N, *B = s.B + tu * N * N;
  const int16_t *Mt = s.M[1] + matrix_offset(L2);
  if (has) xf_stage<L2, OPL>(A, B, Mt, 7, rp, g);
  __syncthreads();
  if (has) {
    int acc[2][OPL];
    xf_sums<L2, OPL>(B, Mt, rp, g, acc);
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int o = 0; o < OPL; o++) {
        uint8_t *q = &s.px[px_index(tu, 2 * rp + e, g * OPL + o)];
        *q = (uint8_t)clip8(*q + ((acc[e][o] + 2048) >> 12));
      }
  }
  __syncthreads();
}

template <class F> __device__ __forceinline__ void dec_inter_body(const F &f, const Wg wg)
{
  __shared__ DecInterLds s;
  const int tid = threadIdx.x;
  if (wg.id >= wg.n) return;
  const int gx = f.wc * 2, lin0 = xcd_contiguous(wg.id, wg.n);
  int by = lin0 / gx; const int bx = lin0 - by * gx;
  by += f.row0 * 2;                                          // (band of CTU rows: the grid covers f.nrows of them)
  const int x0 = bx * 32, y0 = by * 32;
  if (x0 >= f.w || y0 >= f.h) return;
  const int b4w = f.pw >> 2, cpitch = f.pw >> 1, wC = f.w >> 1, hC = f.h >> 1;
  const TuRange reg = f.region[by * (2 * f.wc) + bx];
  const int ntu = (int)(reg.count < DEC_MAX_TU ? reg.count : DEC_MAX_TU);
  if (tid < 64) {
    const int X = x0 + (tid & 7) * 4, Y = y0 + (tid >> 3) * 4;
    B4Rec r; r.mvx = 0; r.mvy = 0; r.ref_idx = -1; r.flags = 0; r.qp_y = 0; r.slot = 0;
    if (X < f.w && Y < f.h) r = f.b4[(size_t)(Y >> 2) * b4w + (X >> 2)];
    s.recs[tid] = r;
    if (f.b4x && (r.flags & (B4_BI | B4_WT))) s.recx[tid] = f.b4x[(size_t)(Y >> 2) * b4w + (X >> 2)];
  }
  if (tid >= 64 && tid < 80) ((uint32_t *)s.mask)[tid - 64] = 0;
  __syncthreads();
  // ---- the region of a still background: one motion for all of it, the zero vector, no transform block -- a copy of the
  // reference region (most regions of a video call's inter pictures; everything below is for the others)
  {
    const B4Rec r0 = s.recs[0];
    bool same = true;
    if (tid < 64) { const B4Rec r = s.recs[tid]; same = r.ref_idx == r0.ref_idx && r.slot == r0.slot && r.mvx == 0 && r.mvy == 0 && !(r.flags & (B4_BI | B4_WT)); }
    if (__syncthreads_and(same) && r0.ref_idx >= 0 && reg.count == 0) {
      const int slot = r0.slot & 15;
      {
        const size_t o = (size_t)(y0 + (tid >> 3)) * f.pw + x0 + (tid & 7) * 4;
        *(uint32_t *)&f.rec[0][o] = *(const uint32_t *)&f.ref[slot][0][o];
      }
      if (tid < 128) {
        const int pl = 1 + (tid >> 6), y = (tid >> 2) & 15, x = (tid & 3) * 4;
        const size_t o = (size_t)((y0 >> 1) + y) * cpitch + (x0 >> 1) + x;
        *(uint32_t *)&f.rec[pl][o] = *(const uint32_t *)&f.ref[slot][pl][o];
      }
      return;
    }
  }
  bool my_coded = false;
  if (tid < 16) {
    const B4Rec *r = &s.recs[(tid >> 2) * 16 + (tid & 3) * 2];
    const B4Rec a = r[0], b = r[1], c = r[8], d = r[9];
    const bool inter = a.ref_idx >= 0;
    const bool uni = a.mvx == b.mvx && a.mvx == c.mvx && a.mvx == d.mvx && a.mvy == b.mvy && a.mvy == c.mvy && a.mvy == d.mvy &&
                     a.slot == b.slot && a.slot == c.slot && a.slot == d.slot;
    const bool frac = ((a.mvx | a.mvy | b.mvx | b.mvy | c.mvx | c.mvy | d.mvx | d.mvy) & 3) != 0;
    const bool bi = ((a.flags | b.flags | c.flags | d.flags) & (B4_BI | B4_WT)) != 0;      // a cell with a bi-predicted or explicitly weighted block takes the general path below (bit 3), none of the window forms
    s.cflag[tid] = (uint8_t)(bi ? ((inter ? 1 : 0) | 8) : ((inter ? 1 : 0) | (uni ? 2 : 0) | (frac ? 4 : 0)));
  }
  if (tid >= 64 && tid - 64 < ntu) {
    const int t = tid - 64;
    DecTu d = f.tus[f.tu_index ? f.tu_index[reg.first + t] : reg.first + t];
    if (d.flags & TU_INTRA) d.count = 0;                       // the intra kernel's business
    s.td[t] = d;
    if (d.count) {
      my_coded = true;
      if (!(d.flags & (TU_TSKIP | TU_BYPASS))) {
        const int cls = d.log2 - 2;
        int blk;
        if (d.plane == 0) blk = zorder3((d.x - x0) >> 2, (d.y - y0) >> 2) >> (2 * cls);
        else blk = (((d.plane - 1) << 4) | zorder3((d.x - (x0 >> 1)) >> 2, (d.y - (y0 >> 1)) >> 2)) >> (2 * cls);
        atomicOr(&s.mask[d.plane ? 1 : 0][cls][blk >> 5], 1u << (blk & 31));
      }
    }
  }
  const bool any_inter = __syncthreads_or(tid < 16 && (s.cflag[tid] & 1));
  if (!any_inter) return;
  const bool coded = __syncthreads_or(my_coded);
  const bool any_lwin = __syncthreads_or(tid < 16 && (s.cflag[tid] & 7) == 7);
  // ---- fractional luma vectors, one motion per 8x8 cell: window -> LDS, horizontal pass -> LDS (8.5.3.3.3.1)
  if (any_lwin) {
    for (int i = tid; i < 16 * 225; i += 256) {
      const int k = i / 225, r = i - k * 225;
      if ((s.cflag[k] & 7) != 7) continue;
      const int wy = r / 15, wx = r - wy * 15;
      const B4Rec m = s.recs[(k >> 2) * 16 + (k & 3) * 2];
      const int gx = x0 + (k & 3) * 8 + (m.mvx >> 2) - 3 + wx, gy = y0 + (k >> 2) * 8 + (m.mvy >> 2) - 3 + wy;
      s.lwin[k][wy * 16 + wx] = (uint8_t)refpx(f.ref[m.slot & 15][0], f.pw, f.w, f.h, gx, gy);
    }
    __syncthreads();
    for (int i = tid; i < 16 * 120; i += 256) {
      const int k = i / 120, r = i - k * 120;
      if ((s.cflag[k] & 7) != 7) continue;
      const int wy = r >> 3, c = r & 7, xf = s.recs[(k >> 2) * 16 + (k & 3) * 2].mvx & 3;
      const uint8_t *wp = &s.lwin[k][wy * 16 + c];
      int v = wp[3];
      if (xf) { v = 0; for (int t = 0; t < 8; t++) v += kLumaFilter[xf][t] * wp[t]; }
      s.ltmp[k][wy * 8 + c] = v;
    }
    __syncthreads();
  }
  // ---- luma prediction: four samples of a row per thread, each 4x4 unit with its own motion
  {
    const int y = tid >> 3, x = (tid & 7) * 4, cell = (y >> 3) * 4 + (x >> 3), cf = s.cflag[cell];
    const B4Rec m = s.recs[(y >> 2) * 8 + (x >> 2)];
    uint32_t p4 = 0;
    if (cf & 8) {
      // B pictures: every 4x4 unit by itself, each sample at 14 bits from one list or the rounded mean of both (8.5.3.3.4.2)
      if (m.ref_idx >= 0) {
        const uint8_t *r0 = f.ref[m.slot & 15][0];
        const bool bi = (m.flags & B4_BI) != 0;
        const B4L1 m1 = s.recx[(y >> 2) * 8 + (x >> 2)];
        const uint8_t *r1 = f.ref[m1.slot & 15][0];
        // explicit weights (8.5.3.3.4.3; f.wt): w0, o0 and w1, o1 from the table entries the record names; log2WD = denominator + 14 - 8
        const bool wt = f.wt && (m.flags & B4_WT);
        const int lw = f.wt_log2[0] + 6, w0 = wt ? f.wt[m1.pad[0] & 31].w[0] : 0, o0 = wt ? f.wt[m1.pad[0] & 31].o[0] : 0, w1 = wt ? f.wt[m1.pad[1] & 31].w[0] : 0, o1 = wt ? f.wt[m1.pad[1] & 31].o[0] : 0;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
          const int a = mc_luma_14(r0, f.pw, f.w, f.h, x0 + x + i, y0 + y, m.mvx, m.mvy);
          const int b = bi ? mc_luma_14(r1, f.pw, f.w, f.h, x0 + x + i, y0 + y, m1.mvx, m1.mvy) : 0;
          int v;
          if (wt) v = bi ? (a * w0 + b * w1 + ((o0 + o1 + 1) << lw)) >> (lw + 1) : ((a * w0 + (1 << (lw - 1))) >> lw) + o0;
          else v = bi ? (a + b + 64) >> 7 : (a + 32) >> 6;
          p4 |= (uint32_t)clip8(v) << (8 * i);
        }
      }
    } else if (cf & 1) {
      const uint8_t *rp = f.ref[m.slot & 15][0];
      if (!((m.mvx | m.mvy) & 3)) p4 = mc_luma4_int(rp, f.pw, f.w, f.h, x0 + x, y0 + y, m.mvx, m.mvy);
      else if ((cf & 6) == 6) {
        const int xf = m.mvx & 3, yf = m.mvy & 3;
        const int *tp = &s.ltmp[cell][(y & 7) * 8 + (x & 7)];
#pragma unroll 1
        for (int i = 0; i < 4; i++) {            // (kept rolled: see enc_kernels.hip k_inter_recon)
          int v;
          if (yf) {
            int a = 0;
            for (int j = 0; j < 8; j++) a += (int)kLumaFilter[yf][j] * tp[j * 8 + i];
            v = xf ? (a >> 6) : a;
          } else v = xf ? tp[3 * 8 + i] : tp[3 * 8 + i] * 64;
          p4 |= (uint32_t)clip8((v + 32) >> 6) << (8 * i);
        }
      } else {
#pragma unroll 1
        for (int i = 0; i < 4; i++) p4 |= (uint32_t)mc_luma_px(rp, f.pw, f.w, f.h, x0 + x + i, y0 + y, m.mvx, m.mvy) << (8 * i);
      }
    }
    *(uint32_t *)&s.px[y * 32 + x] = p4;
  }
  // ---- chroma: 7 x 7 windows of the cells with one motion -> LDS; vector in 1/8 samples
  for (int i = tid; i < 2 * 16 * 49; i += 256) {
    const int pl = i / 784, r = i - pl * 784, k = r / 49, q = r - k * 49;
    if ((s.cflag[k] & 3) != 3) continue;
    const int wy = q / 7, wx = q - wy * 7;
    const B4Rec m = s.recs[(k >> 2) * 16 + (k & 3) * 2];
    const int xi = (x0 >> 1) + (k & 3) * 4 + (m.mvx >> 3) + wx - 1, yi = (y0 >> 1) + (k >> 2) * 4 + (m.mvy >> 3) + wy - 1;
    s.cwin[pl][k][wy * 8 + wx] = (uint8_t)refpx(f.ref[m.slot & 15][1 + pl], cpitch, wC, hC, xi, yi);
  }
  __syncthreads();
  {
    // two samples per thread; the separable 4-tap form with the {0, 64, 0, 0} filter at fraction 0 covers every case of
    // 8.5.3.3.3.2 exactly: (64 * t) >> 6 == t
    const int pl = tid >> 7, y = (tid >> 3) & 15, x = (tid & 7) * 2, cell = (y >> 2) * 4 + (x >> 2), cf = s.cflag[cell];
    const B4Rec m = s.recs[(y >> 1) * 8 + (x >> 1)];
    int p0 = 0, p1 = 0;
    if (cf & 8) {
      if (m.ref_idx >= 0) {
        const bool bi = (m.flags & B4_BI) != 0;
        const B4L1 m1 = s.recx[(y >> 1) * 8 + (x >> 1)];
        const uint8_t *r0 = f.ref[m.slot & 15][1 + pl], *r1 = f.ref[m1.slot & 15][1 + pl];
        int v[2];
        const bool wt = f.wt && (m.flags & B4_WT);
        const int lw = f.wt_log2[1] + 6, w0 = wt ? f.wt[m1.pad[0] & 31].w[1 + pl] : 0, o0 = wt ? f.wt[m1.pad[0] & 31].o[1 + pl] : 0, w1 = wt ? f.wt[m1.pad[1] & 31].w[1 + pl] : 0, o1 = wt ? f.wt[m1.pad[1] & 31].o[1 + pl] : 0;
#pragma unroll 1
        for (int i = 0; i < 2; i++) {
          const int a = mc_chroma_14(r0, cpitch, wC, hC, (x0 >> 1) + x + i, (y0 >> 1) + y, m.mvx, m.mvy);
          const int b = bi ? mc_chroma_14(r1, cpitch, wC, hC, (x0 >> 1) + x + i, (y0 >> 1) + y, m1.mvx, m1.mvy) : 0;
          if (wt) v[i] = clip8(bi ? (a * w0 + b * w1 + ((o0 + o1 + 1) << lw)) >> (lw + 1) : ((a * w0 + (1 << (lw - 1))) >> lw) + o0);
          else v[i] = clip8(bi ? (a + b + 64) >> 7 : (a + 32) >> 6);
        }
        p0 = v[0]; p1 = v[1];
      }
    } else if (cf & 1) {
      const int xf = m.mvx & 7, yf = m.mvy & 7;
      int v0 = 0, v1 = 0;
      const uint8_t *rp = f.ref[m.slot & 15][1 + pl];
      const int gx = (x0 >> 1) + x + (m.mvx >> 3) - 1, gy = (y0 >> 1) + y + (m.mvy >> 3) - 1;
      const uint8_t *w = &s.cwin[pl][cell][(y & 3) * 8 + (x & 3)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int c[5];
        if (cf & 2) {
#pragma unroll
          for (int i = 0; i < 5; i++) c[i] = w[j * 8 + i];
        } else {
#pragma unroll
          for (int i = 0; i < 5; i++) c[i] = refpx(rp, cpitch, wC, hC, gx + i, gy + j);
        }
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { t0 += kChromaFilter[xf][i] * c[i]; t1 += kChromaFilter[xf][i] * c[i + 1]; }
        v0 += kChromaFilter[yf][j] * t0; v1 += kChromaFilter[yf][j] * t1;
      }
      p0 = clip8(((v0 >> 6) + 32) >> 6); p1 = clip8(((v1 >> 6) + 32) >> 6);
    }
    *(uint16_t *)&s.px[1024 + pl * 256 + y * 16 + x] = (uint16_t)(p0 | (p1 << 8));
  }
  // ---- residual of the coded inter blocks
  if (coded) {
    load_matrices(s.M, 0, KV_MATRIX_ENTRIES, tid, 256);
    if (s.mask[0][3][0]) {
      if (tid < 64) ((uint4 *)s.M8)[tid] = ((const uint4 *)g_xf.M8[1])[tid];
      else if (tid < 72) ((uint4 *)s.rowsum)[tid - 64] = ((const uint4 *)g_xf.rowsum[1])[tid - 64];
    }
    for (int i = tid; i < 768; i += 256) ((uint32_t *)s.C)[i] = 0;
    if (tid == 0) { uint32_t a = 0; for (int t = 0; t < ntu; t++) { s.tstart[t] = a; a += s.td[t].count; } s.tstart[ntu] = a; }
    __syncthreads();
    const uint32_t nwords = s.tstart[ntu];
    for (uint32_t i = tid; i < nwords; i += 256) {
      int lo = 0, hi = ntu;                               // last block with tstart <= i (empty blocks share a start: take the last)
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s.tstart[mid] <= i) lo = mid; else hi = mid; }
      const DecTu d = s.td[lo];
      const uint32_t wd = f.lev[d.offset + (i - s.tstart[lo])];
      const int n = 1 << d.log2, pos = (int)(wd >> 16) & (n * n - 1), row = pos >> d.log2, col = pos & (n - 1);
      const int coef = dec_dequant(f, d, pos, (int16_t)(wd & 0xffffu));
      const int lx = d.plane ? d.x - (x0 >> 1) : d.x - x0, ly = d.plane ? d.y - (y0 >> 1) : d.y - y0;
      if (d.flags & (TU_TSKIP | TU_BYPASS)) {             // 8.6.4.2 with transform_skip_flag: residual = coefficient << 7, then the common shift; transquant bypass: the level
        uint8_t *q = d.plane ? &s.px[1024 + (d.plane - 1) * 256 + (ly + row) * 16 + lx + col] : &s.px[(ly + row) * 32 + lx + col];
        *q = (uint8_t)clip8(*q + ((d.flags & TU_BYPASS) ? coef : (((coef << 7) + 2048) >> 12)));
      } else {
        const int base = d.plane ? 1024 + (d.plane - 1) * 256 + zorder3(lx >> 2, ly >> 2) * 16 : zorder3(lx >> 2, ly >> 2) * 16;
        s.C[base + col * n + row] = (int16_t)coef;
      }
    }
    __syncthreads();
    auto lpx = [](int cls) { return [cls](int t, int y, int x) { int ux, uy; unzorder3(t << (2 * cls), ux, uy); return (uy * 4 + y) * 32 + ux * 4 + x; }; };
    auto cpx = [](int cls) { return [cls](int t, int y, int x) {
      const int per = 16 >> (2 * cls), pl = t / per; int ux, uy; unzorder3((t - pl * per) << (2 * cls), ux, uy);
      return 1024 + pl * 256 + (uy * 4 + y) * 16 + ux * 4 + x; }; };
    if (s.mask[0][3][0]) {                                // the 32x32 block: the four stages' inverse half on the matrix cores
      const int wave = tid >> 6, lane = tid & 63;
      const int j = (wave & 1) * 16 + (lane & 15), i0 = (wave >> 1) * 16 + (lane >> 4) * 4;
      mfma_stage(s.C, s.B, s.M8, s.rowsum, 7, wave, lane);
      __syncthreads();
      int acc[4];
      mfma_tile_sums(s.B, s.M8, s.rowsum, wave, lane, acc);
#pragma unroll
      for (int r = 0; r < 4; r++) { uint8_t *q = &s.px[(i0 + r) * 32 + j]; *q = (uint8_t)clip8(*q + ((acc[r] + 2048) >> 12)); }
      __syncthreads();
    }
    if (s.mask[0][2][0]) dec_itx_class<4, 2>(s, s.C, s.mask[0][2][0], 0u, lpx(2), tid);
    if (s.mask[0][1][0]) dec_itx_class<3, 2>(s, s.C, s.mask[0][1][0], 0u, lpx(1), tid);
    if (s.mask[0][0][0] | s.mask[0][0][1]) dec_itx_class<2, 2>(s, s.C, s.mask[0][0][0], s.mask[0][0][1], lpx(0), tid);
    if (s.mask[1][2][0]) dec_itx_class<4, 1>(s, s.C + 1024, s.mask[1][2][0], 0u, cpx(2), tid);
    if (s.mask[1][1][0]) dec_itx_class<3, 1>(s, s.C + 1024, s.mask[1][1][0], 0u, cpx(1), tid);
    if (s.mask[1][0][0]) dec_itx_class<2, 1>(s, s.C + 1024, s.mask[1][0][0], 0u, cpx(0), tid);
  } else __syncthreads();
  // ---- region -> picture (intra blocks of the region hold nothing meaningful yet: the intra kernel writes them afterwards)
  *(uint32_t *)&f.rec[0][(size_t)(y0 + (tid >> 3)) * f.pw + x0 + (tid & 7) * 4] = *(const uint32_t *)&s.px[(tid >> 3) * 32 + (tid & 7) * 4];
  if (tid < 128) {
    const int pl = tid >> 6, y = (tid >> 2) & 15, x = (tid & 3) * 4;
    *(uint32_t *)&f.rec[1 + pl][(size_t)((y0 >> 1) + y) * cpitch + (x0 >> 1) + x] = *(const uint32_t *)&s.px[1024 + pl * 256 + y * 16 + x];
  }
}

// =============================================================================================
// Intra blocks: one workgroup of T threads per (CTU, colour plane)
// =============================================================================================
#define DI_P 144                       // pitch of the CTU picture in LDS
struct DecIntraLds {
  // The CTU with its borders as one padded picture (the layout of enc_kernels.hip IntraCtuLds): row 0 = the sample row above the
  // CTU (corner at column 15, then above and above-right), column 15 = the sample column to its left, sample (x, y) of the CTU at
  // pic[(y + 1) * DI_P + 16 + x].  The borders are copied from the picture piecewise, as the neighbouring CTUs publish them.
  alignas(16) uint8_t pic[65 * DI_P];
  alignas(16) XfLaneF16 xf[4][64];     // matrix operands of the transform stages, per (transform, lane): blocks up to 16x16 (kernel_common.h)
  alignas(16) IntraBlk blk[256];       // what the chain needs to know about each block of list[], worked out ahead of it
  alignas(16) DecTu list[256];         // this plane's intra blocks of the CTU, decoding order (luma: at most 256 4x4 blocks)
  // 32x32 blocks only (one wave running the workgroup-shaped code of the other sizes' predecessor):
  alignas(16) int16_t A[1024], B[1024];
  alignas(16) int16_t M[2][KV_MATRIX_ENTRIES];
  alignas(16) uint8_t R[2][144];       // reference samples in the scan order of 8.4.4.2.2, as built / filtered
};

// 6.4.1 for one slice: inside the picture, same tile, not later in z-scan order (luma locations)
template <class F> __device__ __forceinline__ bool dec_avail(const F &f, int xc, int yc, int xn, int yn)
{
  if (xn < 0 || yn < 0 || xn >= f.w || yn >= f.h) return false;
  const int l = f.ctb_log2;
  if (f.tiles && f.ctu_tile[(yn >> l) * f.cwc + (xn >> l)] != f.ctu_tile[(yc >> l) * f.cwc + (xc >> l)]) return false;
  return zaddr_ctb(xn, yn, f.cwc, l) <= zaddr_ctb(xc, yc, f.cwc, l);
}

// one intra transform block of plane c: N x N samples at (rx, ry) of the CTU (component samples); its borders are in s.pic
// the threads of one block's team: a workgroup of T threads, or -- T = 64 -- ONE wave of a larger workgroup (whose LDS instructions execute in order)
template <int T> __device__ __forceinline__ void tsync() { if (T == 64) wave_sync(); else __syncthreads(); }
template <int L2, int T, bool CIP, bool GEN, class F>
__device__ __forceinline__ void dec_intra_block(const F &f, DecIntraLds &s, const DecTu &d, int c, int cx, int cy, int rx, int ry, int lane, const uint32_t (&wreg)[4])
{
  constexpr int N = XW<L2, T>::N, OPL = XW<L2, T>::OPL, G = XW<L2, T>::G;
  const int sh = c ? 1 : 0, S = (1 << f.ctb_log2) >> sh, nl = N << sh, wC = f.w >> sh, hC = f.h >> sh;
  const int mode = d.mode, cidx = c ? 1 : 0;
  const bool filt = (!GEN || mode < 35) && intra_filter_needed(N, cidx, mode);      // (mode 35: a PCM unit, no prediction)
  const int Xc = cx * S + rx, Yc = cy * S + ry, X = Xc << sh, Y = Yc << sh;
  // (the block's first 4 T level words arrive in wreg: loaded by the caller one block ahead)
  const bool has = d.count != 0, bypass = (d.flags & TU_BYPASS) != 0, tskip = (d.flags & TU_TSKIP) != 0 || bypass;
  // ---- reference samples (8.4.4.2.2) and their filtered version (8.4.4.2.3).  The available samples are contiguous in scan
  // order (one slice, tiles are full-width rows), so the substitution process is a clamp of the scan index into [lo, hi].
  {
    const bool aL = dec_avail(f, X, Y, X - 1, Y), aT = dec_avail(f, X, Y, X, Y - 1), aTL = aL && aT && dec_avail(f, X, Y, X - 1, Y - 1);
    const int nBL = (aL && dec_avail(f, X, Y, X - 1, Y + nl)) ? imin(N, hC - (Yc + N)) : 0;
    const int nTR = ((aT || f.tiles) && dec_avail(f, X, Y, X + nl, Y - 1)) ? imin(N, wC - (Xc + N)) : 0;      // (a slice that begins with the block above-right: available without the one above)
    // ... two runs of available samples, [lo, 2N - 1] (lo = 2N: none) and [lob, hi]: the slice begins with the block above (the corner belongs to another) or with the
    // one above-right; a sample of the gap takes the last one of the run before it, or the first one there is (8.4.4.2.2)
    const bool hole = GEN && f.tiles && ((!aT && nTR > 0) || (aL && aT && !aTL));
    const int lob = aT ? 2 * N + 1 : 3 * N + 1;
    const int lo = aL ? N - nBL : (aTL || hole ? 2 * N : 2 * N + 1), hi = aT || hole ? 3 * N + nTR : (aTL ? 2 * N : (aL ? 2 * N - 1 : -1));
    // constrained_intra_pred_flag: available in the usual sense AND in an intra-coded block, per unit of four samples (a 4x4 record); unit u of the scan order = samples
    // 4u .. 4u + 3 of the left column (u < N / 2), the corner (u = N / 2), samples 2N + 1 + 4 (u - N / 2 - 1) .. of the row above.  8.4.4.2.2: a sample of a unit
    // that does not count takes the last sample of the nearest counting unit before it, the ones in front of the first counting unit its first sample.
    unsigned long long cipm = ~0ull;
    if (CIP) {
      constexpr int NU = N / 2;                            // units of the left column (2N samples); the row above has as many
      const int u = lane, i0 = u < NU ? 4 * u : (u == NU ? 2 * N : 2 * N + 1 + 4 * (u - NU - 1));
      bool av = false;
      if (u <= 2 * NU) {
        const bool sp = hole ? ((i0 >= lo && i0 < 2 * N) || (i0 >= lob && i0 <= hi)) : (i0 >= lo && i0 <= hi);
        if (sp) {
          const int px = i0 <= 2 * N ? Xc - 1 : Xc + i0 - 2 * N - 1, py = i0 < 2 * N ? Yc + 2 * N - 1 - i0 : Yc - 1;
          av = f.b4[(size_t)((py << sh) >> 2) * (f.pw >> 2) + ((px << sh) >> 2)].ref_idx < 0;
        }
      }
      cipm = __ballot(av);
    }
    auto fetch = [&](int i) -> int {
      int j = imin(imax(i, lo), hi);
      if (hole && j < lob) j = lo < 2 * N ? imin(j, 2 * N - 1) : lob;
      if (CIP) {
        constexpr int NU = N / 2;
        const int ii = imin(imax(i, 0), 4 * N), u = ii < 2 * N ? ii >> 2 : (ii == 2 * N ? NU : NU + 1 + ((ii - 2 * N - 1) >> 2));
        auto first_of = [&](int q) { return q < NU ? 4 * q : (q == NU ? 2 * N : 2 * N + 1 + 4 * (q - NU - 1)); };
        if ((cipm >> u) & 1ull) j = ii;
        else {
          const unsigned long long below = u ? cipm & (~0ull >> (64 - u)) : 0ull;
          if (below) { const int q = 63 - __builtin_clzll(below); j = q == NU ? 2 * N : first_of(q) + 3; }
          else j = first_of(__builtin_ctzll(cipm | (1ull << 63)));      // (cipm == 0: no reference sample at all -- the callers' 128)
        }
      }
      const int col = j < 2 * N ? rx - 1 : rx + j - 2 * N - 1, rowp = j < 2 * N ? ry + 2 * N - j : ry;    // rowp = y + 1
      return s.pic[rowp * DI_P + 16 + col];
    };
    int c0 = 0, e0 = 0, e1 = 0; bool strong = false;
    if (filt && N == 32 && (CIP ? cipm != 0 : hi >= 0) && f.strong_intra) {
      c0 = fetch(2 * N); e0 = fetch(0); e1 = fetch(4 * N);
      strong = iabs(c0 + e1 - 2 * fetch(3 * N)) < 8 && iabs(c0 + e0 - 2 * fetch(N)) < 8;
    }
    const bool any_ref = CIP ? cipm != 0 : (hi >= lo && hi >= 0);
    for (int i = lane; i <= 4 * N; i += T) {
      int v = 128, fv = 128;
      if (any_ref) {
        v = fetch(i); fv = v;
        if (filt && i != 0 && i != 4 * N) {
          if (strong) { if (i != 2 * N) { int k = i < 2 * N ? 2 * N - i : i - 2 * N; fv = ((64 - k) * c0 + k * (i < 2 * N ? e0 : e1) + 32) >> 6; } }
          else fv = (fetch(i - 1) + 2 * v + fetch(i + 1) + 2) >> 2;
        }
      }
      s.R[0][3 + i] = (uint8_t)v;
      if (filt) s.R[1][3 + i] = (uint8_t)fv;
    }
  }
  // ---- levels -> dequantised coefficients, transposed ([column][row])
  if (has) {
    for (int i = lane; i < N * N / 2; i += T) ((uint32_t *)s.A)[i] = 0;
    tsync<T>();
    auto put = [&](uint32_t wd) {
      const int pos = (int)(wd >> 16) & (N * N - 1), row = pos >> L2, col = pos & (N - 1);
      s.A[col * N + row] = (int16_t)dec_dequant(f, d, pos, (int16_t)(wd & 0xffffu));
    };
#pragma unroll
    for (int k = 0; k < 4; k++) if (lane + k * T < (int)d.count) put(wreg[k]);
    for (int i = lane + 4 * T; i < (int)d.count; i += T) put(f.lev[d.offset + i]);
  }
  tsync<T>();
  const uint8_t *R = s.R[filt ? 1 : 0] + 3;
  int dcv = 0;
  if (mode == 1) {
    const uint32_t *r4 = (const uint32_t *)(s.R[0] + 3 + N + 1);
    uint32_t acc = N + s.R[0][3 + N] - s.R[0][3 + 2 * N];
#pragma unroll
    for (int k = 0; k < N / 2; k++) acc = __builtin_amdgcn_sad_u8(r4[k], 0u, acc);
    dcv = (int)(acc >> (L2 + 1));
  }
  const bool active = lane < XW<L2, T>::LANES;
  const int rp = lane / G, g = lane % G;
  const bool edge = cidx == 0 && N < 32;
  int pred[2][OPL];
  if (active) {
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int o = 0; o < OPL; o++) pred[e][o] = pred_planar<L2>(R, g * OPL + o, 2 * rp + e);
    } else if (mode == 1) {
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int o = 0; o < OPL; o++) pred[e][o] = pred_dc<L2>(R, edge, dcv, g * OPL + o, 2 * rp + e);
    } else if (GEN && mode == 35) {
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int o = 0; o < OPL; o++) pred[e][o] = 0;
    } else {
      const int angle = kIntraAngle[mode], inv = kInvAngle[mode];
      const bool vert = mode >= 18, e2 = edge && (mode == 26 || mode == 10);
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int o = 0; o < OPL; o++) pred[e][o] = pred_angular<L2>(R, vert, e2, angle, inv, g * OPL + o, 2 * rp + e);
    }
  }
  if (has) {
    if (tskip) {
      if (active) {
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int o = 0; o < OPL; o++) { const int a = s.A[(g * OPL + o) * N + 2 * rp + e]; pred[e][o] = clip8(pred[e][o] + (bypass ? a : (((a << 7) + 2048) >> 12))); }
      }
    } else {
      const int16_t *Mt = s.M[1] + ((L2 == 2 && (d.flags & TU_DST)) ? KV_DST_OFFSET : matrix_offset(L2));
      if (active) xf_stage<L2, OPL>(s.A, s.B, Mt, 7, rp, g);
      tsync<T>();
      if (active) {
        int acc[2][OPL];
        xf_sums<L2, OPL>(s.B, Mt, rp, g, acc);
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int o = 0; o < OPL; o++) pred[e][o] = clip8(pred[e][o] + ((acc[e][o] + 2048) >> 12));
      }
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int o = 0; o < OPL; o++) s.pic[(ry + 2 * rp + e + 1) * DI_P + 16 + rx + g * OPL + o] = (uint8_t)pred[e][o];
  }
  tsync<T>();
  // (the block stays in the CTU picture in LDS: the caller hands its last column / row to the neighbouring CTUs as tagged words, and the CTU goes to the
  // picture in full lines at the end)
}

// The residual of an intra transform block does not depend on its prediction: all of a picture's are computed at once, one wave per
// transform block (levels -> dequantised coefficients -> the two inverse stages on the matrix cores, or the transform-skip shift),
// into plane-shaped int16 arrays.  The dependency chain of k_dec_intra then only predicts, adds and stores.  (32x32 blocks stay
// with the chain's workgroup-shaped code.)  Lane (g, c) owns the four samples x = 4g .. 4g + 3 of row c, as in the chain.
template <class F> __device__ __forceinline__ void dec_intra_resid_body(const F &f, const Wg wg)
{
  __shared__ IntraWaveScratch wsv[4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, idx = wg.id * 4 + wv;
  // (the progress counters and the ticket counter k_dec_intra starts from: this kernel runs in front of it on the same stream, a memset of their own was a launch)
  if (wg.id == 0) for (int i = threadIdx.x; i < 3 * f.cwc * f.chc + 1; i += 256) f.progress[i] = 0;
  if (idx >= f.ntu) return;
  const DecTu d = f.tus[idx];
  if (!(d.flags & TU_INTRA) || !d.count || d.log2 > 4) return;
  IntraWaveScratch &ws = wsv[wv];
  const int L2 = d.log2, N = 1 << L2, g = lane >> 4, c = lane & 15;
  const uint32_t dqc = dequant_pack(d.qp, L2);
  *(uint2 *)&ws.tr[lane * 4] = make_uint2(0u, 0u);
  wave_sync();
  for (int i = lane; i < (int)d.count; i += 64) {
    const uint32_t wd = f.lev[d.offset + i];
    const int pos = (int)(wd >> 16) & (N * N - 1);
    const int lv = (int16_t)(wd & 0xffffu);
    ws.tr[(pos & (N - 1)) * 16 + (pos >> L2)] = (int16_t)((f.scaling || (d.flags & TU_BYPASS)) ? dec_dequant(f, d, pos, lv) : dequant_coef_p(lv, dqc));
  }
  wave_sync();
  int res[4];
  if (d.flags & TU_BYPASS) {
#pragma unroll
    for (int r = 0; r < 4; r++) res[r] = ws.tr[(4 * g + r) * 16 + c];                                  // transquant bypass: the residual is the level
  } else if (d.flags & TU_TSKIP) {
#pragma unroll
    for (int r = 0; r < 4; r++) res[r] = (((int)ws.tr[(4 * g + r) * 16 + c] << 7) + 2048) >> 12;      // residual (x = 4g + r, y = c) = level at row c, column 4g + r
  } else {
    const uint2 t = *(const uint2 *)&ws.tr[c * 16 + 4 * g];
    const int dq[4] = {(int)(int16_t)(t.x & 0xffffu), (int)(int16_t)(t.x >> 16), (int)(int16_t)(t.y & 0xffffu), (int)(int16_t)(t.y >> 16)};
    const int xf = (L2 == 2 && (d.flags & TU_DST)) ? XF16_DST4 : (L2 - 1) & 3;
    wave_sync();
    wave_inverse16(ws, kv_h4(g_xf16.t[xf][lane].tb), dq, g, c, res);
  }
  if (c < N && 4 * g < N) {
    const int pitch = d.plane ? f.pw >> 1 : f.pw;
    *(uint2 *)&f.resid[d.plane][(size_t)(d.y + c) * pitch + d.x + 4 * g] =
        make_uint2(((uint32_t)res[0] & 0xffffu) | ((uint32_t)res[1] << 16), ((uint32_t)res[2] & 0xffffu) | ((uint32_t)res[3] << 16));
  }
}

// one intra transform block of at most 16x16 samples on one wave (kernel_common.h "One intra block per WAVE"); `rres`: the lane's four
// residual samples (k_dec_intra_resid), loaded by the caller one block ahead
template <int L2, bool GEN>
__device__ __forceinline__ void dec_intra_block_wave(DecIntraLds &s, IntraWaveScratch &ws, const IntraBlk &d, bool luma,
                                                     int lane, uint2 rres, uint32_t *ecol, unsigned long long *erow, uint32_t gen, const CipCtx *cip)
{
  constexpr int N = 1 << L2;
  const int g = lane >> 4, c = lane & 15, rx = d.rx, ry = d.ry;
  const bool active = c < N && 4 * g < N;
  int pred[4];
  wave_intra_predict<L2, GEN>(s.pic, DI_P, ws, d, luma, lane, g, c, pred, cip);
  if (d.flags & IB_LEVELS) {
    const int res[4] = {(int)(int16_t)(rres.x & 0xffffu), (int)(int16_t)(rres.x >> 16), (int)(int16_t)(rres.y & 0xffffu), (int)(int16_t)(rres.y >> 16)};
#pragma unroll
    for (int r = 0; r < 4; r++) pred[r] = clip8(pred[r] + res[r]);
  }
  if (active) {
    const uint32_t o = (uint32_t)pred[0] | ((uint32_t)pred[1] << 8) | ((uint32_t)pred[2] << 16) | ((uint32_t)pred[3] << 24);
    *(uint32_t *)&s.pic[(ry + c + 1) * DI_P + 16 + rx + 4 * g] = o;
    // what a neighbouring CTU's wave will read -- the CTU's last row (IB_EDGE) and last column (IB_EDGE_R) -- leaves at once, as self-validating words
    // nobody waits for (kernel_common.h IntraNeighbours; round 4, as in the encoder's chain); the CTU itself goes to the picture in full lines at the end
    if ((d.flags & IB_EDGE) && c == N - 1) st_wt_u64(erow + ((rx + 4 * g) >> 2), (unsigned long long)o | ((unsigned long long)gen << 32));
    if ((d.flags & IB_EDGE_R) && g == (N >> 2) - 1) st_wt_u32(ecol + ry + c, (o >> 24) | (gen << 8));
  }
  wave_sync();
}

// One workgroup of KVZ_DEC_INTRA_WAVES waves per (CTU, colour plane) -- round 4: the encoder's chain (enc_kernels.hip k_intra_recon, kernel_common.h
// IntraChain) driven by the transform-block list.  The plane's intra blocks form ITEMS in decoding order: a block of 8x8 luma samples or more is one, the
// four 4x4 luma blocks of an 8x8 unit are one (they read each other).  A wave takes the next item, fetches the neighbouring CTUs' samples it reads (tagged
// words), waits until the 8x8 units of its own CTU it reads are final -- by the blocks' MODES where that is cheap to say: items whose quadrants do not
// depend on each other run side by side --, predicts, adds the residual (k_dec_intra_resid) and marks its units.  (Rounds 2-3: one wave per (CTU, plane),
// every block behind the one before it.)
#ifndef KVZ_DEC_INTRA_WAVES
#define KVZ_DEC_INTRA_WAVES 4
#endif
template <bool CIP, bool GEN, class F> __device__ __forceinline__ void dec_intra_body(const F &f, const Wg wg)
{
  constexpr int W = KVZ_DEC_INTRA_WAVES, T = 64 * W;
  __shared__ DecIntraLds s;
  __shared__ IntraWaveScratch wss[W];
  __shared__ IntraChain ch;
  __shared__ uint16_t item_first[257];                     // list index of every item's first block (+ the end)
  __shared__ uint2 item_dep[256], item_cover[256];         // per item: the units of this CTU it waits for, the units it finishes
  __shared__ uint32_t nitems_s, ticket_s, im_s[2], lock32;
  // (enc_kernels.hip k_intra_recon: the launch has as many workgroups as the wavefront keeps busy, each takes the next (CTU, plane) in
  // anti-diagonal order from a ticket counter -- f.progress[3 * CTUs] -- so that the chain does not park a workgroup per (CTU, plane) on the chip)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wg.id >= wg.n) return;
  const uint32_t nticket = 3u * (uint32_t)f.cwc * (uint32_t)(f.nrows > 0 ? f.nrows : f.chc);      // (bands -- nrows > 0 -- exist for 64x64 CTBs only)
  for (int i = tid; i < 4 * 64; i += T) ((uint4 *)s.xf)[i] = ((const uint4 *)g_xf16.t)[i];      // the transforms' matrix operands: once per workgroup, not per (CTU, plane)
  if (tid == 0) lock32 = 0;
  for (bool once = true;; once = false) {
  uint32_t ticket = (uint32_t)wg.id;
  if (f.intra_direct) { if (!once) break; }
  else {
    __syncthreads();                                        // (everybody is done with the last (CTU, plane): LDS and the ticket word are free)
    if (tid == 0) ticket_s = atomicAdd(f.progress + (size_t)3 * f.cwc * f.chc, 1u);
    __syncthreads();
    ticket = ticket_s;
  }
  if (ticket >= nticket) break;
  const int ctu = (int)f.intra_order[ticket / 3u], c = (int)(ticket % 3u), cx = ctu % f.cwc, cy = ctu / f.cwc;
  const int sh = c ? 1 : 0, CTB = 1 << f.ctb_log2, S = CTB >> sh, cpitch = f.pw >> sh, wC = f.w >> sh, hC = f.h >> sh;      // (the work unit is the stream's coding tree block: 64, 32 or 16 luma samples a side)
  const TuRange ct = f.ctu[ctu];
  const int count = (int)(ct.count & 0xffffffu);
  if (!((ct.count >> (24 + c)) & 1)) continue;          // no intra block of this plane in the CTU: nothing to wait for, nothing written
  uint8_t *plane = f.rec[c];
  // the CTU as the inter kernel left it (its inter blocks are final, the intra ones get written below) -> LDS
  {
    const uint8_t *src = plane + (size_t)(cy * S) * cpitch + cx * S;
    if (S >= 16) for (int i = tid; i < S * S / 16; i += T) { const int y = i / (S / 16), xq = i % (S / 16); *(uint4 *)&s.pic[(y + 1) * DI_P + 16 + xq * 16] = *(const uint4 *)&src[(size_t)y * cpitch + xq * 16]; }
    else if (tid < S) *(uint2 *)&s.pic[(tid + 1) * DI_P + 16] = *(const uint2 *)&src[(size_t)tid * cpitch];      // (the 8x8 chroma block of a 16x16 CTB)
  }
  // this plane's intra blocks, compacted in order (wave 0: one pass over the CTU's list, 64 descriptors at a time), and the items they form
  if (wave == 0) {
    int nlist = 0;
    for (int base = 0; base < count; base += 64) {
      DecTu d; d.plane = 255; d.flags = 0;
      if (base + lane < count) d = f.tus[ct.first + base + lane];
      const bool keep = d.plane == c && (d.flags & TU_INTRA);
      const uint64_t m = __ballot(keep);
      const int at = nlist + __popcll(m & ((1ull << lane) - 1ull));
      if (keep && at < 256) s.list[at] = d;
      nlist += __popcll(m);
    }
    nlist = nlist > 256 ? 256 : nlist;
    wave_sync();
    int nitems = 0;
    for (int base = 0; base < nlist; base += 64) {
      const int k = base + lane;
      bool starts = false;
      if (k < nlist) {
        const DecTu t = s.list[k];
        starts = true;
        if (c == 0 && t.log2 == 2 && k > 0) { const DecTu p = s.list[k - 1]; starts = !(p.log2 == 2 && (p.x >> 3) == (t.x >> 3) && (p.y >> 3) == (t.y >> 3)); }      // a 4x4 luma block inside the unit of the block before it
      }
      const uint64_t m = __ballot(starts);
      if (starts) item_first[nitems + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
      nitems += __popcll(m);
    }
    if (lane == 0) { item_first[nitems] = (uint16_t)nlist; nitems_s = (uint32_t)nitems; im_s[0] = 0; im_s[1] = 0; }
  }
  __syncthreads();
  const int nitems = wave_uniform_int((int)nitems_s), nlist = (int)item_first[nitems];      // (nitems bounds the claim loop: uniform by construction, kernel_common.h chain_claim)
  // ---- what the chain needs to know about each block, one thread per block: position, available reference samples (8.4.4.2.2:
  // contiguous in scan order for one slice with full-width tiles), the mode's constants
  bool any32 = false;
  for (int k = tid; k < nlist; k += T) {
    const DecTu t = s.list[k];
    const int N = 1 << t.log2, nl = N << sh, Xc = t.x, Yc = t.y, X = Xc << sh, Y = Yc << sh, rx = Xc - cx * S, ry = Yc - cy * S;
    const bool aL = dec_avail(f, X, Y, X - 1, Y), aT = dec_avail(f, X, Y, X, Y - 1), aTL = aL && aT && dec_avail(f, X, Y, X - 1, Y - 1);
    const int nBL = (aL && dec_avail(f, X, Y, X - 1, Y + nl)) ? imin(N, hC - (Yc + N)) : 0;
    const int nTR = ((aT || f.tiles) && dec_avail(f, X, Y, X + nl, Y - 1)) ? imin(N, wC - (Xc + N)) : 0;      // (a slice that begins with the block above-right: available without the one above)
    const bool hole = GEN && f.tiles && ((!aT && nTR > 0) || (aL && aT && !aTL));      // (two runs of available samples: kernel_common.h IB_HOLE)
    const int lo = aL ? N - nBL : (aTL || hole ? 2 * N : 2 * N + 1), hi = aT || hole ? 3 * N + nTR : (aTL ? 2 * N : (aL ? 2 * N - 1 : 0));
    const int zu = zunit8((rx << sh) >> 3, (ry << sh) >> 3);
    IntraBlk d;
    d.rx = (uint8_t)rx; d.ry = (uint8_t)ry; d.lo = (uint8_t)lo; d.hi = (uint8_t)hi; d.mode = t.mode; d.l2 = t.log2;
    d.flags = (uint8_t)((((!GEN || t.mode < 35) && intra_filter_needed(N, c ? 1 : 0, t.mode)) ? IB_FILT : 0) | ((rx == 0 || ry == 0) ? IB_BORDER : 0) |
                        (t.count ? IB_LEVELS : 0) | ((t.flags & TU_TSKIP) ? IB_TSKIP : 0) | (hole ? IB_HOLE : 0) |
                        ((ry + N >= S) ? IB_EDGE : 0) | ((rx + N >= S) ? IB_EDGE_R : 0));
    d.xf = (uint8_t)(hole ? (aT ? 2 * N + 1 : 3 * N + 1) : 0);      // (the decoder's chain adds residuals computed elsewhere: the field is the second run's start of an IB_HOLE block)
    d.angle = (int16_t)((!GEN || t.mode < 35) ? kIntraAngle[t.mode] : 0); d.inv = (int16_t)((!GEN || t.mode < 35) ? kInvAngle[t.mode] : 0);      // (mode 35: a PCM unit, no prediction)
    d.zu = (uint16_t)zu; d.next = 0;
    s.blk[k] = d;
    any32 |= t.log2 == 5;
  }
  // ... and about each item: the 8x8 luma units it covers and the units of this CTU its blocks read (kernel_common.h chain_dependencies: by the block's
  // mode for a block of its own; everything around for the four 4x4 blocks of a unit -- a superset is as good, it only has to precede in z-order)
  for (int i = tid; i < nitems; i += T) {
    const DecTu t = s.list[item_first[i]];
    const bool quad = item_first[i + 1] - item_first[i] > 1;
    const int rx = t.x - cx * S, ry = t.y - cy * S, ux = (rx << sh) >> 3, uy = (ry << sh) >> 3, su = quad ? 1 : imax(1, ((1 << t.log2) << sh) >> 3);
    const int ci = c ? 1 : 0;
    // (constrained intra prediction: a sample that does not count takes its value from wherever the nearest counting one is -- any block may read any of its borders)
    const bool bl = quad || t.log2 == 2 || CIP || ((intra_uses_below_left(t.log2, ci) >> t.mode) & 1) != 0, tr = quad || t.log2 == 2 || CIP || ((intra_uses_above_right(t.log2, ci) >> t.mode) & 1) != 0;
    item_dep[i] = chain_dependencies(ux, uy, su, bl, tr);
    const uint2 cv = chain_cover(zunit8(ux, uy), su);
    item_cover[i] = cv;
    if (cv.x) atomicOr(&im_s[0], cv.x);
    if (cv.y) atomicOr(&im_s[1], cv.y);
  }
  if (__syncthreads_or(any32)) load_matrices(s.M, 0, KV_MATRIX_ENTRIES, tid, T);
  IntraNeighbours bd;
  uint32_t *const ecol = f.edge_col[c] + (size_t)ctu * S;                              // this CTU's right column / bottom row for its neighbours
  unsigned long long *const erow = f.edge_row[c] + (size_t)ctu * (S >> 2);
  {
    const int tile = f.ctu_tile[ctu];
    bd.nb_left = cx > 0 && f.ctu_tile[ctu - 1] == tile; bd.nb_up = cy > 0 && f.ctu_tile[ctu - f.cwc] == tile;
    bd.nb_ur = cy > 0 && cx + 1 < f.cwc && f.ctu_tile[ctu - f.cwc + 1] == tile; bd.nb_ul = cy > 0 && cx > 0 && f.ctu_tile[ctu - f.cwc - 1] == tile;
    bd.ecol_left = ecol - S; bd.gen = f.chain_gen;
    bd.erow_up = erow - (size_t)f.cwc * (S >> 2); bd.erow_ur = bd.erow_up + (S >> 2); bd.erow_ul = bd.erow_up - (S >> 2);
    // which of the neighbours' edge units are intra units (a P picture's inter blocks are final before this kernel starts: nothing to wait
    // for there) -- lanes 0-7: the left CTU's right column, 8-15 / 16-23: the bottom rows of the upper / upper-right CTU, 24: the corner (every wave for itself)
    const int g = lane >> 3, u = lane & 7;
    int X = -1, Y = -1;
    const bool uin = u * 8 < CTB;                            // (a CTB of 32 / 16 samples has four / two edge units, not eight)
    if (g == 0 && bd.nb_left && uin) { X = cx * CTB - 8; Y = cy * CTB + u * 8; }
    else if (g == 1 && bd.nb_up && uin) { X = cx * CTB + u * 8; Y = cy * CTB - 8; }
    else if (g == 2 && bd.nb_ur && uin) { X = (cx + 1) * CTB + u * 8; Y = cy * CTB - 8; }
    else if (lane == 24 && bd.nb_ul) { X = cx * CTB - 8; Y = cy * CTB - 8; }
    const bool in = X >= 0 && X < f.w && Y < f.h && f.b4[(size_t)(Y >> 2) * (f.pw >> 2) + (X >> 2)].ref_idx < 0;
    const uint64_t m = __ballot(in);
    bd.il = (uint32_t)m & 0xffu; bd.iu = (uint32_t)(m >> 8) & 0xffu; bd.iur = (uint32_t)(m >> 16) & 0xffu; bd.iul = (uint32_t)(m >> 24) & 1u;
  }
  chain_init(ch, ~im_s[0], ~im_s[1]);                      // (units no intra block of this plane covers -- inter units, units outside the picture -- are final from the start)
  __syncthreads();                                         // (also publishes blk[] and the items)
  const int16_t *rplane = f.resid[c] + (size_t)(cy * S) * cpitch + cx * S;
  IntraWaveScratch &ws = wss[wave];
  for (int rounds = 0;; rounds++) {
    const int it = chain_claim(ch, lane);
    if (it >= nitems) break;
    // (bounded like every loop of the chain: a CTU has at most 256 items.  Round 4 needed the counter for another reason: with `nitems` in a vector register
    // the loop written as `for (;;)` was compiled into one the wave never left -- kernel_common.h chain_claim has the story; the bound is scalar now and
    // either spelling is a plain scalar loop, checked in the listing)
    if (rounds > 300) { if (lane == 0) atomicOr(f.err, 4u); break; }
    const int k0 = (int)item_first[it], k1 = (int)item_first[it + 1];
    const uint2 dp = item_dep[it], cv = item_cover[it];
    for (int k = k0; k < k1; k++) {
      const IntraBlk d = wave_uniform(&s.blk[k]);          // (wave-uniform: what is derived from it runs on the scalar unit)
      // what the block needs from memory is asked for first: its residual samples (blocks up to 16x16: k_dec_intra_resid) or the first level words of a
      // 32x32 block, then the neighbouring CTUs' samples -- all of it arrives while the wave waits for the blocks in front of it
      uint32_t wreg[4] = {0, 0, 0, 0};
      uint2 rres = make_uint2(0u, 0u);
      if (d.flags & IB_LEVELS) {
        if (d.l2 > 4) {
          const uint32_t off = s.list[k].offset; const int cnt = s.list[k].count;
#pragma unroll
          for (int q = 0; q < 4; q++) if (lane + q * 64 < cnt) wreg[q] = f.lev[off + lane + q * 64];
        } else {
          const int n = 1 << d.l2, g = lane >> 4, r = lane & 15;
          if (r < n && 4 * g < n) rres = *(const uint2 *)&rplane[(size_t)(d.ry + r) * cpitch + d.rx + 4 * g];
        }
      }
      if (d.flags & IB_BORDER) {
        // (a neighbouring CTU is waited for only as far as the block's MODE reads it: hevc_core.h intra_uses_* -- any stream; this project's encoder keeps the blocks
        // on a CTU's left edge and its above-right corner block to the modes that make these waits short, "intra-chain")
        const int n = 1 << d.l2, ci = c ? 1 : 0;
        const int nl2 = (((intra_uses_below_left(d.l2, ci) >> d.mode) & 1) || CIP) ? 2 * n : n, nt2 = (((intra_uses_above_right(d.l2, ci) >> d.mode) & 1) || (GEN && (d.flags & IB_HOLE)) || CIP) ? 2 * n : n;      // (IB_HOLE: the samples above may be copies of the first one above-right)
        borders_need_wave(ch, bd, s.pic, DI_P, plane, cpitch, cx, cy, S, sh, wC - cx * S, d.rx, d.ry, n, f.err, lane, nl2, nt2);
      }
      if (k == k0) chain_wait_done(ch, make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)dp.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)dp.y)), f.err, lane);
      CipCtx cipc; const CipCtx *cip = nullptr;
      if (CIP) { cipc.ridx = (const int8_t *)f.b4 + 4; cipc.b4w = f.pw >> 2; cipc.xl = cx * S + d.rx; cipc.yl = cy * S + d.ry; cipc.sh = sh; cip = &cipc; }      // (constrained intra prediction: B4Rec::ref_idx, byte 4 of a record)
      switch (d.l2) {
        case 2: dec_intra_block_wave<2, GEN>(s, ws, d, c == 0, lane, rres, ecol, erow, f.chain_gen, cip); break;
        case 3: dec_intra_block_wave<3, GEN>(s, ws, d, c == 0, lane, rres, ecol, erow, f.chain_gen, cip); break;
        case 4: dec_intra_block_wave<4, GEN>(s, ws, d, c == 0, lane, rres, ecol, erow, f.chain_gen, cip); break;
        default: {
          // a 32x32 block: the wave runs the workgroup-shaped code of the other sizes' predecessor by itself; its scratch arrays (s.A, s.B, s.R) exist
          // once per workgroup, so one 32x32 block at a time
          if (lane == 0) { while (atomicCAS(&lock32, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2); }
          wave_sync();
          DecTu t;
          const uint32_t *q = (const uint32_t *)&s.list[k];
          uint32_t u[4];
#pragma unroll
          for (int i = 0; i < 4; i++) u[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q[i]);
          memcpy(&t, u, sizeof(t));
          dec_intra_block<5, 64, CIP, GEN>(f, s, t, c, cx, cy, d.rx, d.ry, lane, wreg);
          // (on the CTU's right edge / bottom: its last column / row from the CTU picture in LDS, tagged like the small blocks')
          if ((d.flags & IB_EDGE_R) && lane < 32) st_wt_u32(ecol + d.ry + lane, (uint32_t)s.pic[(d.ry + lane + 1) * DI_P + 16 + d.rx + 31] | (f.chain_gen << 8));
          if ((d.flags & IB_EDGE) && lane < 8) st_wt_u64(erow + ((d.rx + 4 * lane) >> 2), (unsigned long long)*(const uint32_t *)&s.pic[(d.ry + 32) * DI_P + 16 + d.rx + 4 * lane] | ((unsigned long long)f.chain_gen << 32));
          wave_sync();
          if (lane == 0) __hip_atomic_store(&lock32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    chain_mark_done(ch, make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)cv.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)cv.y)), lane);
  }
  // the CTU's samples -> the picture, in whole lines (the chain stored only what neighbouring workgroups read)
  __syncthreads();
  uint8_t *gdst = plane + (size_t)(cy * S) * cpitch + cx * S;
  if (S >= 16) for (int i = tid; i < S * S / 16; i += T) { const int y = i / (S / 16), xq = i % (S / 16); *(uint4 *)&gdst[(size_t)y * cpitch + xq * 16] = *(const uint4 *)&s.pic[(y + 1) * DI_P + 16 + xq * 16]; }
  else if (tid < S) *(uint2 *)&gdst[(size_t)tid * cpitch] = *(const uint2 *)&s.pic[(tid + 1) * DI_P + 16];
  }
}

// =============================================================================================
// Deblocking (8.7.2): both passes in one launch, one workgroup per 64x64 tile shifted by (-4, -4) against the CTU grid
// (enc_kernels.hip k_deblock_tile has the geometry argument); boundary strengths from the 4x4 records
// =============================================================================================
// px / qx: where the second vectors of p / q are when they are bi-predicted (B4_BI; B pictures)
__device__ __forceinline__ int dec_bs(const B4Rec &p, const B4Rec &q, bool tu_edge, const B4L1 *px = nullptr, const B4L1 *qx = nullptr)
{
  if (p.ref_idx < 0 || q.ref_idx < 0) return 2;
  if (tu_edge && ((p.flags | q.flags) & B4_NZ)) return 1;
  if ((p.flags ^ q.flags) & B4_BI) return 1;              // a different number of motion vectors
  if (!(p.flags & B4_BI)) {
    if (p.slot != q.slot) return 1;                        // different reference pictures (8.7.2.4 looks at pictures: which list or index names them does not matter)
    if (iabs(p.mvx - q.mvx) >= 4 || iabs(p.mvy - q.mvy) >= 4) return 1;
    return 0;
  }
  const B4L1 p1 = *px, q1 = *qx;
  auto far = [](int ax, int ay, int bx, int by) { return iabs(ax - bx) >= 4 || iabs(ay - by) >= 4; };
  const bool straight = p.slot == q.slot && p1.slot == q1.slot, crossed = p.slot == q1.slot && p1.slot == q.slot;
  if (!straight && !crossed) return 1;                     // different reference pictures
  if (p.slot != p1.slot)                                   // two pictures: the vectors into the same picture are compared
    return straight ? (far(p.mvx, p.mvy, q.mvx, q.mvy) || far(p1.mvx, p1.mvy, q1.mvx, q1.mvy)) : (far(p.mvx, p.mvy, q1.mvx, q1.mvy) || far(p1.mvx, p1.mvy, q.mvx, q.mvy));
  // all four vectors point into one picture: either pairing may be the close one
  return (far(p.mvx, p.mvy, q.mvx, q.mvy) || far(p1.mvx, p1.mvy, q1.mvx, q1.mvy)) && (far(p.mvx, p.mvy, q1.mvx, q1.mvy) || far(p1.mvx, p1.mvy, q.mvx, q.mvy));
}

template <class F> __device__ __forceinline__ void dec_deblock_body(const F &f, const Wg wg)
{
  constexpr int P = 80, PC = 48;                           // LDS pitches (enc_kernels.hip k_deblock_tile: same tile geometry and I/O)
  __shared__ __attribute__((aligned(16))) uint8_t ty_[68 * P];
  __shared__ __attribute__((aligned(16))) uint8_t tc_[2][34 * PC];
  __shared__ B4Rec recs[18 * 18];                          // the tile's 4x4 records and one ring: units -1 .. 16 in both directions
  const int tid = threadIdx.x, wc = f.wc, hc = f.hc, b4w = f.pw >> 2;
  if (wg.id >= wg.n) return;
  const int lin = xcd_contiguous(wg.id, wg.n), tx = lin % wc, tyi = lin / wc + f.row0;
  const int X0 = tx * 64 - 4, Y0 = tyi * 64 - 4, CX0 = X0 >> 1, CY0 = Y0 >> 1, cw2 = f.pw >> 1;
  const int TW = tx == wc - 1 ? 68 : 64, TH = tyi == hc - 1 ? 68 : 64;
  for (int i = tid; i < 18 * 18; i += 256) {
    const int ux = tx * 16 - 1 + i % 18, uy = tyi * 16 - 1 + i / 18;
    B4Rec r; r.mvx = 0; r.mvy = 0; r.ref_idx = -1; r.flags = 0; r.qp_y = 0; r.slot = 0;
    if (ux >= 0 && uy >= 0 && ux * 4 < f.w && uy * 4 < f.h) r = f.b4[(size_t)uy * b4w + ux];
    recs[i] = r;
  }
  __syncthreads();
  auto unit = [&](int x, int y) { return ((y >> 2) - (tyi * 16 - 1)) * 18 + ((x >> 2) - (tx * 16 - 1)); };
  // ---- boundary strengths of this thread's vertical and horizontal edge segment, from the records alone.  A tile none of whose
  // segments is filtered (still background) leaves without touching a sample.
  int bsv = 0, bsh = 0, qpv = 0, qph = 0;
  if (tid < 8 * (TH / 4)) {
    const int x = tx * 64 + (tid & 7) * 8, y = Y0 + (tid >> 3) * 4;
    if (x > 0 && x < f.w && y >= 0 && y < f.h) {
      const int uq = unit(x, y);
      const B4Rec q = recs[uq], p = recs[uq - 1];
      if (q.flags & B4_EDGE_V) { const B4L1 *qx = f.b4x ? f.b4x + (size_t)(y >> 2) * b4w + (x >> 2) : nullptr; bsv = dec_bs(p, q, (q.flags & B4_TU_V) != 0, qx - 1, qx); qpv = (p.qp_y + q.qp_y + 1) >> 1; }
    }
  }
  if (tid < 8 * (TW / 4)) {
    const int y = tyi * 64 + (tid & 7) * 8, x = X0 + (tid >> 3) * 4;
    if (y > 0 && y < f.h && x >= 0 && x < f.w) {
      const int uq = unit(x, y);
      const B4Rec q = recs[uq], p = recs[uq - 18];
      if (q.flags & B4_EDGE_H) { const B4L1 *qx = f.b4x ? f.b4x + (size_t)(y >> 2) * b4w + (x >> 2) : nullptr; bsh = dec_bs(p, q, (q.flags & B4_TU_H) != 0, qx - b4w, qx); qph = (p.qp_y + q.qp_y + 1) >> 1; }
    }
  }
  if (!__syncthreads_or(bsv | bsh)) return;
  for (int i = tid; i < TH * 5; i += 256) {
    const int y = i / 5, k = i - y * 5, x = k * 16, gx = X0 + x, gy = Y0 + y;
    if (gy < 0 || x >= TW) continue;
    const uint8_t *g = &f.rec[0][(size_t)gy * f.pw + gx];
    if (k < 4) { if (gx >= 0) *(kv_u32x4 *)&ty_[y * P + x] = *(const kv_u32x4 *)g; else { kv_u32x4 v; v.x = 0; v.y = *(const uint32_t *)(g + 4); v.z = *(const uint32_t *)(g + 8); v.w = *(const uint32_t *)(g + 12); *(kv_u32x4 *)&ty_[y * P + x] = v; } }
    else *(uint32_t *)&ty_[y * P + x] = *(const uint32_t *)g;
  }
  for (int i = tid; i < 2 * (TH / 2) * 9; i += 256) {
    const int pl = i / ((TH / 2) * 9), r = i - pl * ((TH / 2) * 9), y = r / 9, k = r - y * 9, gx = CX0 - 2 + 4 * k, gy = CY0 + y;
    if (gy >= 0 && gx >= 0) *(uint32_t *)&tc_[pl][y * PC + 4 * k] = *(const uint32_t *)&f.rec[1 + pl][(size_t)gy * cw2 + gx];
  }
  __syncthreads();
  // ---- vertical edges: 8 edges x 16 (17) four-row segments
  if (bsv) {
    const int x = tx * 64 + (tid & 7) * 8, y = Y0 + (tid >> 3) * 4;
    deblock_luma_segment(&ty_[(y - Y0) * P + (x - X0)], 1, P, bsv, qpv, f.beta_offset, f.tc_offset);
    if (bsv == 2 && (x & 15) == 0) {
      const int o = ((y >> 1) - CY0) * PC + ((x >> 1) - CX0) + 2;
      deblock_chroma_segment(&tc_[0][o], 1, PC, 2, qpv, f.cb_qp_offset, f.tc_offset);
      deblock_chroma_segment(&tc_[1][o], 1, PC, 2, qpv, f.cr_qp_offset, f.tc_offset);
    }
  }
  __syncthreads();
  // coding units with cu_transquant_bypass_flag are not modified by the filter (8.7.2.5.7: nDp / nDq = 0).  The picture in memory still holds the unfiltered
  // samples (this kernel writes at its end): their units are fetched again after each pass.
  auto restore_bypass = [&]() {
    for (int i = tid; i < 17 * 17; i += 256) {
      const int uy = i / 17, ux = i - uy * 17;                    // unit inside the tile: luma (4 ux, 4 uy) from (X0, Y0)
      if (4 * ux >= TW || 4 * uy >= TH) continue;
      const int gx = X0 + 4 * ux, gy = Y0 + 4 * uy;
      if (gx < 0 || gy < 0 || gx >= f.w || gy >= f.h) continue;
      if (!(recs[uy * 18 + ux].flags & B4_BYPASS)) continue;      // (recs: units -1 .. 16 of the CTU = units 0 .. 17 of the tile shifted by one unit)
      for (int r = 0; r < 4; r++) *(uint32_t *)&ty_[(4 * uy + r) * P + 4 * ux] = *(const uint32_t *)&f.rec[0][(size_t)(gy + r) * f.pw + gx];
      for (int pl = 0; pl < 2; pl++)
        for (int r = 0; r < 2; r++) *(uint16_t *)&tc_[pl][(2 * uy + r) * PC + 2 * ux + 2] = *(const uint16_t *)&f.rec[1 + pl][(size_t)((gy >> 1) + r) * cw2 + (gx >> 1)];
    }
    __syncthreads();
  };
  if (f.tq_bypass) restore_bypass();
  // ---- horizontal edges on the vertically filtered samples
  if (bsh) {
    const int y = tyi * 64 + (tid & 7) * 8, x = X0 + (tid >> 3) * 4;
    deblock_luma_segment(&ty_[(y - Y0) * P + (x - X0)], P, 1, bsh, qph, f.beta_offset, f.tc_offset);
    if (bsh == 2 && (y & 15) == 0) {
      const int o = ((y >> 1) - CY0) * PC + ((x >> 1) - CX0) + 2;
      deblock_chroma_segment(&tc_[0][o], PC, 1, 2, qph, f.cb_qp_offset, f.tc_offset);
      deblock_chroma_segment(&tc_[1][o], PC, 1, 2, qph, f.cr_qp_offset, f.tc_offset);
    }
  }
  __syncthreads();
  if (f.tq_bypass) restore_bypass();
  for (int i = tid; i < TH * 5; i += 256) {
    const int y = i / 5, k = i - y * 5, x = k * 16, gx = X0 + x, gy = Y0 + y;
    if (gy < 0 || x >= TW) continue;
    uint8_t *g = &f.rec[0][(size_t)gy * f.pw + gx];
    if (k < 4) {
      const kv_u32x4 v = *(const kv_u32x4 *)&ty_[y * P + x];
      if (gx >= 0) *(kv_u32x4 *)g = v; else { *(uint32_t *)(g + 4) = v.y; *(uint32_t *)(g + 8) = v.z; *(uint32_t *)(g + 12) = v.w; }
    } else *(uint32_t *)g = *(const uint32_t *)&ty_[y * P + x];
  }
  for (int i = tid; i < 2 * (TH / 2) * 9; i += 256) {
    const int pl = i / ((TH / 2) * 9), r = i - pl * ((TH / 2) * 9), y = r / 9, k = r - y * 9, gx = CX0 - 2 + 4 * k, gy = CY0 + y;
    if (gy < 0 || gx < 0) continue;
    uint8_t *g = &f.rec[1 + pl][(size_t)gy * cw2 + gx];
    const uint32_t v = *(const uint32_t *)&tc_[pl][y * PC + 4 * k];
    if (k == 0) *(uint16_t *)(g + 2) = (uint16_t)(v >> 16);
    else if (k == 8 && TW == 64) *(uint16_t *)g = (uint16_t)v;
    else *(uint32_t *)g = v;
  }
}

// =============================================================================================
// Sample adaptive offset (8.7.3) with the parsed parameters: one workgroup per CTU, the deblocked CTU with a one-sample ring in LDS
// =============================================================================================
struct DecSaoLds {
  alignas(4) uint8_t win[66 * 72];
  alignas(4) uint8_t winc[2][34 * 40];
  SaoParams p[16];                     // the parameters of the tile's coding tree blocks: one (64x64 CTBs), four (32x32) or sixteen (16x16)
};
__device__ __forceinline__ int dsao_edge_idx(int c, int a, int b)
{
  const int e = 2 + ((c > a) - (c < a)) + ((c > b) - (c < b));
  return e == 2 ? 0 : (e < 2 ? e + 1 : e);
}

template <class F> __device__ __forceinline__ void dec_sao_body(const F &f, const Wg wg)
{
  __shared__ DecSaoLds s;
  if (wg.id >= wg.n) return;
  const int tid = threadIdx.x, wc = f.wc, ctu = xcd_contiguous(wg.id, wg.n), cx = ctu % wc, cy = ctu / wc;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, l2n = 6 - sh, n = 1 << l2n, pitch_g = f.pw >> sh, pwid = f.w >> sh, phei = f.h >> sh, X0 = cx * n, Y0 = cy * n;
    uint8_t *w = c ? s.winc[c - 1] : s.win; const int pitch = c ? 40 : 72;
    const uint8_t *src = f.rec[c];
#pragma unroll
    for (int i = tid; i < n * n / 4; i += 256) {
      const int y = i >> (l2n - 2), x = (i & ((n >> 2) - 1)) * 4;
      *(uint32_t *)&w[(y + 1) * pitch + 4 + x] = *(const uint32_t *)&src[(size_t)(Y0 + y) * pitch_g + X0 + x];   // (inside the padded allocation)
    }
    for (int q = tid; q < 4 * n + 4; q += 256) {
      int x, y;
      if (q < n + 2) { x = q - 1; y = -1; } else if (q < 2 * n + 4) { x = q - (n + 2) - 1; y = n; }
      else if (q < 3 * n + 4) { x = -1; y = q - (2 * n + 4); } else { x = n; y = q - (3 * n + 4); }
      w[(y + 1) * pitch + 4 + x] = src[(size_t)clip3(0, phei - 1, Y0 + y) * pitch_g + clip3(0, pwid - 1, X0 + x)];
    }
  }
  // the 64x64 tile's coding tree blocks (raster inside the tile): their parameters; blocks outside the picture never get a sample filtered
  const int cl = f.ctb_log2, per = 64 >> cl;
  if (tid < per * per) {
    const int ccx = cx * per + tid % per, ccy = cy * per + tid / per;
    if (ccx < f.cwc && ccy < f.chc) s.p[tid] = f.sao[ccy * f.cwc + ccx];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, l2n = 6 - sh, n = 1 << l2n, pitch_g = f.pw >> sh, pwid = f.w >> sh, phei = f.h >> sh, X0 = cx * n, Y0 = cy * n;
    const uint8_t *w = c ? s.winc[c - 1] : s.win; const int pitch = c ? 40 : 72;
    int type = s.p[0].type[c], e = s.p[0].eo_class[c], bp = s.p[0].band_pos[c];
    int off[4];
    for (int k = 0; k < 4; k++) off[k] = s.p[0].offset[c][k];
#pragma unroll
    for (int q = tid; q < n * n / 4; q += 256) {
      const int y = q >> (l2n - 2), x4 = (q & ((n >> 2) - 1)) * 4;
      if (Y0 + y >= phei || X0 + x4 >= pwid) continue;                          // partial CTU: outside the picture
      if (cl != 6) {                                                            // (CTBs smaller than the tile: the four samples' own block's parameters)
        const SaoParams &P = s.p[(((y << sh) >> cl) * per) + ((x4 << sh) >> cl)];
        type = P.type[c]; e = P.eo_class[c]; bp = P.band_pos[c];
        for (int k = 0; k < 4; k++) off[k] = P.offset[c][k];
      }
      const uint32_t *row = (const uint32_t *)&w[(y + 1) * pitch + x4];
      uint32_t out = row[1];
      if (type == 1) {
        uint32_t o = 0;
        for (int i = 0; i < 4; i++) {
          const int v = (out >> (8 * i)) & 255, k = ((v >> 3) - bp) & 31;
          o |= (uint32_t)(k < 4 ? clip8(v + off[k]) : v) << (8 * i);
        }
        out = o;
      } else if (type == 2) {
        auto span = [&](const uint32_t *r) -> uint64_t { return ((uint64_t)r[0] >> 24) | ((uint64_t)r[1] << 8) | ((uint64_t)(r[2] & 255u) << 40); };
        const int dq = pitch >> 2;
        const uint64_t mid = span(row), up = span(row - dq), dn = span(row + dq);
        const bool okv = Y0 + y - 1 >= 0 && Y0 + y + 1 < phei;
        // closed slice / tile boundaries (DecFrame::ctu_nb): the four samples lie in one coding tree block; a neighbour across a closed boundary of it leaves the
        // sample as it is (8.7.3.2: edgeIdx 0)
        uint32_t nbm = 0xffu; int px = 0, py = 0, S = 0;
        if (f.ctu_nb) {
          nbm = f.ctu_nb[((((Y0 + y) << sh) >> cl)) * f.cwc + (((X0 + x4) << sh) >> cl)];
          S = (1 << cl) >> sh; px = (X0 + x4) & (S - 1); py = (Y0 + y) & (S - 1);
        }
        auto usable = [&](int i, int dx, int dy) -> bool {
          const int cdx = (dx < 0 && px + i == 0) ? -1 : ((dx > 0 && px + i == S - 1) ? 1 : 0), cdy = (dy < 0 && py == 0) ? -1 : ((dy > 0 && py == S - 1) ? 1 : 0);
          if (!cdx && !cdy) return true;
          const int k = (cdy + 1) * 3 + cdx + 1;
          return ((nbm >> (k > 4 ? k - 1 : k)) & 1u) != 0;
        };
        uint32_t o = 0;
        for (int i = 0; i < 4; i++) {
          const int v = (int)((mid >> (8 * (i + 1))) & 255);
          int a, b; bool ok;
          const bool okh = X0 + x4 + i - 1 >= 0 && X0 + x4 + i + 1 < pwid;
          if (e == 0) { a = (int)((mid >> (8 * i)) & 255); b = (int)((mid >> (8 * (i + 2))) & 255); ok = okh; }
          else if (e == 1) { a = (int)((up >> (8 * (i + 1))) & 255); b = (int)((dn >> (8 * (i + 1))) & 255); ok = okv; }
          else if (e == 2) { a = (int)((up >> (8 * i)) & 255); b = (int)((dn >> (8 * (i + 2))) & 255); ok = okh && okv; }
          else { a = (int)((up >> (8 * (i + 2))) & 255); b = (int)((dn >> (8 * i)) & 255); ok = okh && okv; }
          if (nbm != 0xffu) {
            const int dxa = (e == 1) ? 0 : (e == 3 ? 1 : -1), dya = (e == 0) ? 0 : -1;      // Table 8-12: the first neighbour; the second one is opposite
            ok = ok && usable(i, dxa, dya) && usable(i, -dxa, -dya);
          }
          const int k = ok ? dsao_edge_idx(v, a, b) : 0;
          o |= (uint32_t)(k ? clip8(v + off[k - 1]) : v) << (8 * i);
        }
        out = o;
      }
      if (f.tq_bypass) {                                                        // 8.7.3: samples of coding units with cu_transquant_bypass_flag are not modified
        const int lx = (X0 + x4) << sh, ly = (Y0 + y) << sh, b4w = f.pw >> 2;
        const bool k0 = (f.b4[(size_t)(ly >> 2) * b4w + (lx >> 2)].flags & B4_BYPASS) != 0;
        const bool k1 = c ? (f.b4[(size_t)(ly >> 2) * b4w + (lx >> 2) + 1].flags & B4_BYPASS) != 0 : k0;      // (four chroma samples span two luma units)
        if (k0) out = (out & 0xffff0000u) | (row[1] & 0x0000ffffu);
        if (k1) out = (out & 0x0000ffffu) | (row[1] & 0xffff0000u);
      }
      *(uint32_t *)&f.out[c][(size_t)(Y0 + y) * pitch_g + X0 + x4] = out;
    }
  }
}

// =============================================================================================
// launch wrappers
// =============================================================================================
// the single-picture kernels: the frame is the kernel argument
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_dec_inter(DecFrame f) { dec_inter_body(f, Wg{(int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y)}); }
__global__ __launch_bounds__(256) void k_dec_intra_resid(DecFrame f) { dec_intra_resid_body(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }
// three forms of the chain: the plain one (one slice, one tile, no PCM: reference samples are one run, 209 registers), the general one (DecFrame::general: two runs,
// PCM units), and the one for constrained intra prediction (DecFrame::cip: any pattern)
__global__ __launch_bounds__(64 * KVZ_DEC_INTRA_WAVES) void k_dec_intra(DecFrame f) { dec_intra_body<false, false>(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }
__global__ __launch_bounds__(64 * KVZ_DEC_INTRA_WAVES) void k_dec_intra_gen(DecFrame f) { dec_intra_body<false, true>(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }
// ... with constrained_intra_pred_flag (DecFrame::cip): reference samples of blocks that are not intra-coded do not count -- a form of its own: the general substitution
// costs the chain 37 registers
__global__ __launch_bounds__(64 * KVZ_DEC_INTRA_WAVES) void k_dec_intra_cip(DecFrame f) { dec_intra_body<true, true>(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }
__global__ __launch_bounds__(256) void k_dec_deblock(DecFrame f) { dec_deblock_body(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }
__global__ __launch_bounds__(256) void k_dec_sao(DecFrame f) { dec_sao_body(f, Wg{(int)blockIdx.x, (int)gridDim.x}); }

// the batched kernels (dec_frame.h DecBatch): a workgroup finds its frame by its index (wave-uniform: scalar compares), then runs the same body
// on the frame descriptor where it lies in device memory -- read through the constant address space, i.e. with scalar loads on demand,
// exactly as a kernel argument is
__device__ __forceinline__ int batch_frame(const DecBatch &b, Wg &wg)
{
  int i = 0;
#pragma unroll
  for (int k = 1; k < KVZ_DEC_BATCH_MAX; k++) i += blockIdx.x >= b.first[k] ? 1 : 0;
  wg.id = (int)(blockIdx.x - b.first[i]); wg.n = (int)b.count[i];
  return i;
}
#define KVZ_BATCH_FRAME(b) Wg wg; const CDecFrame &f = *(const CDecFrame *)(b).f[batch_frame(b, wg)]
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_dec_inter_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_inter_body(f, wg); }
__global__ __launch_bounds__(256) void k_dec_intra_resid_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_intra_resid_body(f, wg); }
__global__ __launch_bounds__(64 * KVZ_DEC_INTRA_WAVES) void k_dec_intra_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_intra_body<false, false>(f, wg); }
__global__ __launch_bounds__(64 * KVZ_DEC_INTRA_WAVES) void k_dec_intra_gen_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_intra_body<false, true>(f, wg); }
__global__ __launch_bounds__(256) void k_dec_deblock_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_deblock_body(f, wg); }
__global__ __launch_bounds__(256) void k_dec_sao_n(DecBatch b) { KVZ_BATCH_FRAME(b); dec_sao_body(f, wg); }

// =============================================================================================
// launch wrappers
// =============================================================================================
static inline int dec_rows(const DecFrame &f) { return f.nrows > 0 ? f.nrows : f.hc; }
static inline int dec_intra_wgs(const DecFrame &f)
{
  // as many one-wave workgroups as three anti-diagonals of the CTU wavefront hold, in three planes (k_dec_intra: tickets; f.intra_order lists the band's CTUs)
  static const int diags = getenv("KVAZZUP_AMD_INTRA_DIAGS") ? atoi(getenv("KVAZZUP_AMD_INTRA_DIAGS")) : 3;      // (measurement aid; 0: a workgroup per (CTU, plane))
  const int nr = f.nrows > 0 ? f.nrows : f.chc, diag = nr < (f.cwc + 1) / 2 ? nr : (f.cwc + 1) / 2, all = f.cwc * nr * 3, want = (diags > 0 && !f.intra_direct) ? 3 * diags * diag + 32 : all;
  return want < all ? want : all;
}
void launch_dec_inter(const DecFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_dec_inter, dim3(f.wc * 2, dec_rows(f) * 2), dim3(256), 0, st, f); }
void launch_dec_intra_resid(const DecFrame &f, hipStream_t st) { if (f.ntu > 0) hipLaunchKernelGGL(k_dec_intra_resid, dim3((f.ntu + 3) / 4), dim3(256), 0, st, f); }
void launch_dec_intra(const DecFrame &f, hipStream_t st)
{
  if (f.cip) hipLaunchKernelGGL(k_dec_intra_cip, dim3(dec_intra_wgs(f)), dim3(64 * KVZ_DEC_INTRA_WAVES), 0, st, f);
  else if (f.general) hipLaunchKernelGGL(k_dec_intra_gen, dim3(dec_intra_wgs(f)), dim3(64 * KVZ_DEC_INTRA_WAVES), 0, st, f);
  else hipLaunchKernelGGL(k_dec_intra, dim3(dec_intra_wgs(f)), dim3(64 * KVZ_DEC_INTRA_WAVES), 0, st, f);
}
void launch_dec_deblock(const DecFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_dec_deblock, dim3(f.wc * dec_rows(f)), dim3(256), 0, st, f); }
void launch_dec_sao(const DecFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_dec_sao, dim3(f.wc * f.hc), dim3(256), 0, st, f); }

// share-out of a batched launch: wgs(frame) workgroups for every frame with a device descriptor, each share rounded up to a multiple of 8
template <class W> static uint32_t batch_layout(DecBatch &b, const DecFrame *const *h, const DecFrame *const *d, int n, W &&wgs)
{
  uint32_t at = 0;
  for (int i = 0; i < KVZ_DEC_BATCH_MAX; i++) {
    b.first[i] = at; b.f[i] = nullptr; b.count[i] = 0;
    if (i < n && d[i]) { b.f[i] = d[i]; b.count[i] = (uint32_t)wgs(*h[i]); at += (b.count[i] + 7u) & ~7u; }
  }
  b.first[KVZ_DEC_BATCH_MAX] = at;
  return at;
}
void launch_dec_inter_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st)
{
  DecBatch b; const uint32_t total = batch_layout(b, h, d, n, [](const DecFrame &f) { return f.wc * 2 * dec_rows(f) * 2; });
  if (total) hipLaunchKernelGGL(k_dec_inter_n, dim3(total), dim3(256), 0, st, b);
}
void launch_dec_intra_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st)
{
  DecBatch b; uint32_t total = batch_layout(b, h, d, n, [](const DecFrame &f) { return (f.ntu + 3) / 4; });
  if (total) hipLaunchKernelGGL(k_dec_intra_resid_n, dim3(total), dim3(256), 0, st, b);
  total = batch_layout(b, h, d, n, [](const DecFrame &f) { return dec_intra_wgs(f); });
  bool general = false;
  for (int i = 0; i < n; i++) general |= h[i]->general != 0;
  if (total && general) hipLaunchKernelGGL(k_dec_intra_gen_n, dim3(total), dim3(64 * KVZ_DEC_INTRA_WAVES), 0, st, b);
  else if (total) hipLaunchKernelGGL(k_dec_intra_n, dim3(total), dim3(64 * KVZ_DEC_INTRA_WAVES), 0, st, b);
}
void launch_dec_deblock_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st)
{
  DecBatch b; const uint32_t total = batch_layout(b, h, d, n, [](const DecFrame &f) { return f.wc * dec_rows(f); });
  if (total) hipLaunchKernelGGL(k_dec_deblock_n, dim3(total), dim3(256), 0, st, b);
}
void launch_dec_sao_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st)
{
  DecBatch b; const uint32_t total = batch_layout(b, h, d, n, [](const DecFrame &f) { return f.wc * f.hc; });
  if (total) hipLaunchKernelGGL(k_dec_sao_n, dim3(total), dim3(256), 0, st, b);
}

}  // namespace kvzx
