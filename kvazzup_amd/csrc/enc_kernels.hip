// kvazzup_amd/csrc/enc_kernels.hip -- CDNA4 (gfx950) kernels of the HEVC encoder hot path.
//
// What they replace: the work Kvazaar does inside kvz_api->encoder_encode
// (/root/reference/src/media/processing/kvazaarfilter.cpp:435-438): motion search, prediction,
// transform/quantisation, reconstruction, deblocking and CABAC.  The arithmetic is checked
// against oracle/ (tests/); the decisions follow "uvgx encoder algorithm v1" (oracle/hevc_enc.h).
//
// Launch geometry (coded size is a multiple of 64; DESIGN.md section 5 has the table with timings):
//   k_pad_input     packed I420 -> padded planes, sixteen samples per thread
//   k_vaq_stats/apply  per-CTU luma variance -> delta QP (vaq only)
//   k_me            one workgroup per 32x32 luma block (XCD-aware order): early termination on the co-located block, else the
//                   search window staged in LDS, v_qsad_pk_u16_u8 on quads x pairs of candidates, wave min of (cost << 13 | index)
//   k_inter_recon   one workgroup per 32x32 block: MC, residual, DCT (32x32: MFMA i8), quant, dequant, IDCT, reconstruction
//                   (two forms: with fractional vectors -- subme > 0, separable LDS passes -- and without)
//   k_inter_signal  one thread per 16x16 block: merge / skip / AMVP signalling
//   k_intra_analyse one workgroup per 32x32 region: 35-mode search on source samples, cost = 8x8 Hadamard sums on the matrix cores
//                   (intra-satd, default) or SAD; work item = (16x16 tile, mode, kind) on one wave
//   k_intra_recon   four waves per (CTU, colour plane), one intra block per WAVE and no workgroup barrier in the chain (transforms on
//                   v_mfma_f32_16x16x16_f16, references over DPP / ds_bpermute); CTUs coupled by progress counters with 8x8 granularity
//   k_qp_first/chain  per-CTU QP bookkeeping (cu_qp_delta)
//   k_deblock_tile  one workgroup per 64x64 tile shifted by (-4, -4): vertical then horizontal edges in LDS
//                   (k_deblock_v / k_deblock_h: one thread per 4-sample edge segment, band mode of the tile-row split)
//   k_sao           one workgroup per CTU: statistics + decision (encoder) and the filter
//   k_tokenize      one wave per 16x16 unit (and colour component): binarisation + context selection -> bins as 16-bit tokens;
//   k_tok_compact   restores coding order per CTU, dense copy to host-mapped memory for the host arithmetic coder
// (the decoder's kernels live in dec_kernels.hip, the fractional-sample search in subpel_kernels.hip, rate control in rc_kernels.hip)
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

#define SPLIT_BITS 8
#define INTRA_P_GATE 24       // intra-in-P: a quarter is a candidate when its inter cost exceeds this many lambda_q4 (oracle/hevc_enc.c)
#define INTRA_P_BITS 16       // ... and goes intra when the intra cost plus this many bins is below the inter cost

// =============================================================================================
// Motion estimation
// =============================================================================================
#define ME_MAXR 32
#define ME_WPITCH 100      // bytes per window row in LDS: 32 + 2*32 = 96, + 4 so the 9th dword read stays inside

// Full search over (2R+1)^2 integer displacements for one 32x32 block, with the four 16x16 quarters
// kept apart.  The search window sits in LDS.  A work item is a QUAD of horizontally adjacent
// candidates times a PAIR of vertically adjacent ones: v_qsad_pk_u16_u8 produces the four SADs of a
// quad against four current samples in one instruction (16-bit accumulators: a 16x16 quarter sums
// to at most 65280), and each window row read from LDS serves both candidates of the pair.
// (Measured on MI355X: v_qsad_pk_u16_u8 issues at ~24 cycles per wave, v_sad_u8 at ~4.7; the quad form
// still wins because it needs no v_alignbyte to line the window up with the candidate.)
__global__ __launch_bounds__(256) void k_me(EncFrame f)
{
  __shared__ __attribute__((aligned(16))) uint8_t win[(32 + 2 * ME_MAXR + 1) * ME_WPITCH + 16];
  __shared__ __attribute__((aligned(16))) uint32_t cur[32 * 8];
  __shared__ uint32_t red[5];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  int bx_, by_; xcd_block_2d(bx_, by_);
  const int x0 = bx_ * 32, y0 = by_ * 32 + f.row0 * 64;
  const int R = f.range, W = 2 * R + 1, WW = 32 + 2 * R;
  const uint8_t *ref = f.me_ref, *src = f.src[0];
  const uint32_t lam = (uint32_t)f.lambda_q4;
  if (f.pb_on && blockIdx.x == 0 && blockIdx.y == 0) picture_begin_body(f.pb_rc, f.pb_bits3, f.pb_slot3, f.pb_have3, f.pb_qt, f.pb_roi, f.pb_nctu, f.qp, 0, tid, nthreads);      // (nothing in this launch reads what it writes: k_inter_recon is the first)
  // the block itself, and (me-early-termination) its SAD against the co-located block of the reference: a block that differs
  // from it by no more than quantisation noise is coded unsplit with the zero vector, without a search
  if (tid == 0) red[0] = 0;
  __syncthreads();
  {
    uint32_t s0 = 0;
    for (int i = tid; i < 256; i += nthreads) {
      const size_t g = (size_t)(y0 + (i >> 3)) * f.cw + x0 + (i & 7) * 4;
      const uint32_t c = *reinterpret_cast<const uint32_t *>(src + g);
      cur[i] = c;
      if (f.me_early) s0 = __builtin_amdgcn_sad_u8(c, *reinterpret_cast<const uint32_t *>(ref + g), s0);
    }
    if (f.me_early) { s0 = wave_sum_u32(s0); if ((tid & 63) == 0) atomicAdd(&red[0], s0); }
  }
  __syncthreads();
  if (f.me_early && red[0] <= 64u * lam) {
    if (tid < 16) {
      const int i = b8idx(f, x0 + (tid & 3) * 8, y0 + (tid >> 2) * 8);
      f.cu_log2[i] = 5; f.cu_intra[i] = 0; f.cu_mv[i * 2] = 0; f.cu_mv[i * 2 + 1] = 0;
      f.cu_mvp_idx[i] = 0;                               // mark for k_subpel: not searched (k_inter_signal writes the real value later)
    }
    if (f.intra_p && tid < 4) f.me_cost16[((y0 >> 4) + (tid >> 1)) * (f.cw >> 4) + (x0 >> 4) + (tid & 1)] = 0;      // never an intra candidate
    return;
  }
  __syncthreads();                                                     // (red[] is about to be re-initialised)
  for (int i = tid; i < (WW + 1) * (ME_WPITCH / 4); i += nthreads) {   // four window samples per thread; columns >= WW and row WW are padding
    const int wy = i / (ME_WPITCH / 4), wx = (i - wy * (ME_WPITCH / 4)) * 4;
    const int gy = clip3(0, f.ch - 1, y0 - R + wy), gx = x0 - R + wx;
    const uint8_t *row = ref + (size_t)gy * f.cw;
    uint32_t v;
    if (gx >= 0 && gx + 7 < f.cw) {
      const uint32_t *q = (const uint32_t *)(row + (gx & ~3));
      v = __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(gx & 3));
    } else {
      v = 0;
      for (int k = 0; k < 4; k++) v |= (uint32_t)row[clip3(0, f.cw - 1, gx + k)] << (8 * k);
    }
    *(uint32_t *)&win[wy * ME_WPITCH + wx] = v;
  }
  if (tid < 5) red[tid] = 0xffffffffu;
  __syncthreads();
  uint32_t best[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  // tile constraint (statement: me_block32() in oracle/hevc_enc.c): the displaced 32x32 block, plus 4 rows each side for
  // the chroma half-sample taps when the displacement is odd, stays inside its tile -- except across the picture's own edges
  int ty0 = 0, ty1 = f.ch, tx0 = 0, tx1 = f.cw;
  if (f.tile_rows > 1) {
    const int hc = f.ch >> 6, tr = tile_row_of(hc, f.tile_rows, y0 >> 6);
    ty0 = tile_row_first(hc, f.tile_rows, tr) * 64; ty1 = tile_row_first(hc, f.tile_rows, tr + 1) * 64;
  }
  if (f.tile_cols > 1) {                                       // ... and in x with tile columns
    const int wc = f.cw >> 6, tc = tile_col_of(wc, f.tile_cols, x0 >> 6);
    tx0 = tile_col_first(wc, f.tile_cols, tc) * 64; tx1 = tile_col_first(wc, f.tile_cols, tc + 1) * 64;
  }
  const int NQ = (W + 3) >> 2, NG = (W + 1) >> 1;
  for (int item = tid; item < NQ * NG; item += nthreads) {
    const int g = item / NQ, q = item - g * NQ, dy0 = 2 * g;
    uint64_t acc[2][4];                                          // [candidate of the pair][quarter]: four u16 sums, one per candidate of the quad
    const uint8_t *wbase = win + dy0 * ME_WPITCH + 4 * q;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      uint64_t al = 0, ar = 0, bl = 0, br = 0;                   // left / right quarter of this half, candidates A and B
#pragma unroll 1
      for (int rr = 0; rr < 17; rr++) {
        // window row dy0 + wr is row wr of candidate A (dy0) and row wr - 1 of candidate B (dy0 + 1); each half covers
        // the rows of both candidates that fall into its quarters: rows half*16 .. +15
        const int wr = half * 16 + rr;
        const uint32_t *wp = (const uint32_t *)(wbase + wr * ME_WPITCH);
        uint32_t d[9];
#pragma unroll
        for (int j = 0; j < 9; j++) d[j] = wp[j];
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = ((uint64_t)d[j + 1] << 32) | d[j];
        if (rr < 16) {
          const uint32_t *c = &cur[wr * 8];
#pragma unroll
          for (int j = 0; j < 4; j++) { al = __builtin_amdgcn_qsad_pk_u16_u8(p[j], c[j], al); ar = __builtin_amdgcn_qsad_pk_u16_u8(p[j + 4], c[j + 4], ar); }
        }
        if (rr > 0) {
          const uint32_t *c = &cur[(wr - 1) * 8];
#pragma unroll
          for (int j = 0; j < 4; j++) { bl = __builtin_amdgcn_qsad_pk_u16_u8(p[j], c[j], bl); br = __builtin_amdgcn_qsad_pk_u16_u8(p[j + 4], c[j + 4], br); }
        }
      }
      acc[0][half * 2] = al; acc[0][half * 2 + 1] = ar; acc[1][half * 2] = bl; acc[1][half * 2 + 1] = br;
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int dyi = dy0 + e;
      if (dyi >= W) continue;
      { const int dy = dyi - R, m = (dy & 1) ? 4 : 0; if ((ty0 > 0 && y0 + dy - m < ty0) || (ty1 < f.ch && y0 + dy + 32 + m > ty1)) continue; }
      if (f.mv_frame) { const int dy = dyi - R, my = (f.mv_frame == 2 && (dy & 1)) ? 4 : 0; if (y0 + dy - my < 0 || y0 + dy + 32 + my > f.ch) continue; }
      const int ry = mvd_bits((dyi - R) * 4);
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++) {
        const int dxi = 4 * q + k4;
        if (dxi >= W) continue;
        { const int dx = dxi - R, m = (dx & 1) ? 4 : 0; if ((tx0 > 0 && x0 + dx - m < tx0) || (tx1 < f.cw && x0 + dx + 32 + m > tx1)) continue; }
        if (f.mv_frame) { const int dx = dxi - R, mx = (f.mv_frame == 2 && (dx & 1)) ? 4 : 0; if (x0 + dx - mx < 0 || x0 + dx + 32 + mx > f.cw) continue; }
        const uint32_t cand = (uint32_t)(dyi * W + dxi);
        const uint32_t rate = (lam * (uint32_t)(mvd_bits((dxi - R) * 4) + ry)) >> 4;
        uint32_t sq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sq[k] = (uint32_t)(acc[e][k] >> (16 * k4)) & 0xffffu;
#pragma unroll
        for (int k = 0; k < 4; k++) best[k] = min(best[k], ((sq[k] + rate) << 13) | cand);
        best[4] = min(best[4], ((sq[0] + sq[1] + sq[2] + sq[3] + rate) << 13) | cand);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; k++) {
    uint32_t v = best[k];
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    if ((tid & 63) == 0) atomicMin(&red[k], v);
  }
  __syncthreads();
  if (tid < 16) {
    uint32_t pen = (lam * SPLIT_BITS) >> 4;
    uint32_t csplit = pen + (red[0] >> 13) + (red[1] >> 13) + (red[2] >> 13) + (red[3] >> 13);
    bool split = csplit < (red[4] >> 13);
    int bx = tid & 3, by = tid >> 2;                     // 8x8 block inside the 32x32 block
    int k = (by >> 1) * 2 + (bx >> 1);
    uint32_t ci = (split ? red[k] : red[4]) & 0x1fff;
    int i = b8idx(f, x0 + bx * 8, y0 + by * 8);
    f.cu_log2[i] = split ? 4 : 5;
    f.cu_intra[i] = 0;
    f.cu_mvp_idx[i] = 1;                                 // mark for k_subpel: searched
    f.cu_mv[i * 2] = (int16_t)(((int)(ci % W) - R) * 4);
    f.cu_mv[i * 2 + 1] = (int16_t)(((int)(ci / W) - R) * 4);
    // intra-in-P: the inter cost of the block's 16x16 quarters -- what the search found for a quarter, or a quarter of the 32x32 block's cost
    if (f.intra_p && tid < 4) f.me_cost16[((y0 >> 4) + (tid >> 1)) * (f.cw >> 4) + (x0 >> 4) + (tid & 1)] = split ? red[tid] >> 13 : ((red[4] >> 13) + 2) >> 2;
    if (f.intra_p && tid == 0) {
      // ... and the block joins the list k_intra_analyse<P> works through when a quarter is above the gate (f.me_cand: count, then block indices)
      bool any = false;
      for (int k = 0; k < 4; k++) any |= (split ? red[k] >> 13 : ((red[4] >> 13) + 2) >> 2) > (uint32_t)INTRA_P_GATE * lam;
      if (any) f.me_cand[1 + atomicAdd(&f.me_cand[0], 1u)] = (uint32_t)((y0 >> 5) * (f.cw >> 5) + (x0 >> 5));
    }
  }
}

// =============================================================================================
// Inter reconstruction of one 32x32 block (one 32x32 CU or four 16x16 CUs), one workgroup each
// =============================================================================================
__device__ __forceinline__ int ref_at(const uint8_t *p, int w, int h, int x, int y)
{
  return p[clip3(0, h - 1, y) * w + clip3(0, w - 1, x)];
}
// luma sample with the 8-tap filters of 8.5.3.3.3.1; mv in quarter samples (general path: fractional vectors)
__device__ __forceinline__ int mc_luma_sample(const uint8_t *p, int w, int h, int x, int y, int mvx, int mvy)
{
  int xf = mvx & 3, yf = mvy & 3, xi = x + (mvx >> 2), yi = y + (mvy >> 2), v;
  if (!xf && !yf) return ref_at(p, w, h, xi, yi);
  if (!yf) { v = 0; for (int i = 0; i < 8; i++) v += kLumaFilter[xf][i] * ref_at(p, w, h, xi + i - 3, yi); }
  else if (!xf) { v = 0; for (int i = 0; i < 8; i++) v += kLumaFilter[yf][i] * ref_at(p, w, h, xi, yi + i - 3); }
  else {
    v = 0;
    for (int j = 0; j < 8; j++) {
      int t = 0;
      for (int i = 0; i < 8; i++) t += kLumaFilter[xf][i] * ref_at(p, w, h, xi + i - 3, yi + j - 3);
      v += kLumaFilter[yf][j] * t;
    }
    v >>= 6;
  }
  return clip8((v + 32) >> 6);
}
// four luma samples (x .. x + 3, y) predicted with one vector, packed little-endian
__device__ __forceinline__ uint32_t mc_luma4(const uint8_t *p, int w, int h, int x, int y, int mvx, int mvy)
{
  if (((mvx | mvy) & 3) == 0) {
    const int xi = x + (mvx >> 2), yi = clip3(0, h - 1, y + (mvy >> 2));
    const uint8_t *row = p + (size_t)yi * w;
    if (xi >= 0 && xi + 3 < w) {
      const uint32_t *q = (const uint32_t *)(row + (xi & ~3));                 // planes are 4-byte aligned, w is a multiple of 64
      const uint32_t lo = q[0], hi = (xi & 3) ? q[1] : 0u;
      return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(xi & 3));
    }
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) v |= (uint32_t)row[clip3(0, w - 1, xi + i)] << (8 * i);
    return v;
  }
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) v |= (uint32_t)mc_luma_sample(p, w, h, x + i, y, mvx, mvy) << (8 * i);
  return v;
}

struct InterLds {
  alignas(16) int16_t A[1024], B[1024];
  alignas(16) int16_t M[2][KV_MATRIX_ENTRIES];
  alignas(16) uint8_t px[1024];            // prediction, then reconstruction: luma 32x32 raster; chroma 2 planes x 16x16
  alignas(16) uint8_t win[2][4][11 * 12];  // chroma reference windows: plane x 8x8 sub-block, 11 x 11 samples each
  uint32_t nz[2];
  int mv[4][2];
  int intra_q[4];                          // intra-in-P: the 16x16 quarter is an intra unit
  int rc_qp, rc_last; uint32_t rc_part[4]; // rate control v2 inside the launch (k_inter_recon<.., RC>)
  alignas(16) int8_t M8[2][32 * 32];       // the 32-point matrix and its transpose as int8: MFMA B operands
  int rowsum[2][32];                       // sum over m of M8[.][j][m]
};
// fractional-sample luma interpolation (subme > 0), per 16x16 quadrant: the 23 x 23 reference window and the horizontally filtered
// rows (8.5.3.3.3.1).  Only the FRAC form of k_inter_recon has it: without these 8 KB eight workgroups fit a compute unit.
struct InterFracLds {
  alignas(16) uint8_t lwin[4][23 * 24];
  alignas(16) int ltmp[4][23 * 16];        // (32-bit on purpose: with int16 entries hipcc 7.2 mis-extends the upper halves of the packed loads)
};

// Level adjustment ("uvgx RDOQ v1" / sign data hiding, hevc_core.h adjust_group) of a block whose levels lie row-major in lev[n * n] and
// whose quantiser remainders lie in aux[n * n]: thread `t` of the block's threads takes the coefficient groups t, t + nthreads, ...
__device__ __forceinline__ void adjust_block_lds(int16_t *lev, const int16_t *aux, int n, int pitch, int scan_idx, int rdoq, int signhide, int t, int nthreads)
{
  const int nsb = n >> 2, ngroups = nsb * nsb;
  for (int gi = t; gi < ngroups; gi += nthreads) {
    const int xs = gi % nsb, ys = gi / nsb;
    int16_t lv[16]; uint16_t ax[16]; int at[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int xp = scan_idx == 0 ? kDiag4x[k] : (scan_idx == 1 ? (k & 3) : (k >> 2)), yp = scan_idx == 0 ? kDiag4y[k] : (scan_idx == 1 ? (k >> 2) : (k & 3));
      at[k] = ((ys << 2) + yp) * pitch + (xs << 2) + xp;
      lv[k] = lev[at[k]]; ax[k] = (uint16_t)aux[((ys << 2) + yp) * n + (xs << 2) + xp];
    }
    adjust_group(lv, ax, gi == 0, rdoq, signhide);
#pragma unroll
    for (int k = 0; k < 16; k++) lev[at[k]] = lv[k];
  }
}

// The four transform stages of `NTU` blocks of n = 1 << L2 held TU-major in s.A (encoder: residual, row-major;
// decoder: dequantised levels, transposed), NTU * XF<L2, OPL>::LANES == 256.  Encoder: levels go to `coef`
// for blocks that have any, *nz gets one bit per block.  Ends with the residual added into s.px.
//   px_index(tu, y, x) -> index of sample (x, y) of block tu in s.px;  coef_at(tu, y, x) -> its level in the plane
//   RC (rate control v2 inside the launch): qp_late() is called between the forward transform and the quantiser -- it waits for the QP if it has to --
//   and the cost of the levels (3 + 2 floor(log2 |level|) each) is added to `cost`
struct NoLateQp { __device__ int operator()() const { return 0; } };
__device__ __forceinline__ uint32_t rc_level_cost(int lv) { return lv ? 3u + 2u * (uint32_t)(31 - __builtin_clz((unsigned)iabs(lv))) : 0u; }
//   mtab: the scaling factors of this block size and plane for inter blocks (`scaling-list default`), NULL: flat
template <bool DEC, int L2, int OPL, bool RC = false, class PX, class CI, class QL = NoLateQp>
__device__ __forceinline__ void inter_transform(InterLds &s, int qp, uint32_t *nz, PX px_index, CI coef_at, int tid, int adj = 0, QL qp_late = QL(), uint32_t *cost = nullptr, const uint8_t *mtab0 = nullptr, int tu_per_plane = 1 << 30)
{
  constexpr int N = 1 << L2, G = XF<L2, OPL>::G, LPT = XF<L2, OPL>::LANES;
  const int tu = tid / LPT, l = tid % LPT, rp = l / G, g = l % G;
  const uint8_t *mtab = mtab0 ? mtab0 + (tu / tu_per_plane) * N * N : nullptr;      // (the chroma call holds Cb and Cr blocks: their matrices follow one another)
  int16_t *A = s.A + tu * N * N, *B = s.B + tu * N * N;
  if (!DEC) {
    const int16_t *Mf = s.M[0] + matrix_offset(L2);
    xf_stage<L2, OPL>(A, B, Mf, L2 - 1, rp, g);
    __syncthreads();
    int acc[2][OPL], lv[2][OPL];
    xf_sums<L2, OPL>(B, Mf, rp, g, acc);
    if constexpr (RC) qp = qp_late();
    const int shift = L2 + 6, rnd = 1 << (shift - 1);
    bool any = false;
    if (adj) {
      // rdoq / signhide (bit 0 / bit 1 of adj): levels and quantiser remainders go to B and A in the block's own layout, a pass over the
      // 4x4 coefficient groups adjusts the levels (one thread per group), then every thread takes its levels back
      __syncthreads();                                           // (all threads are done reading B: xf_sums)
#pragma unroll
      for (int o = 0; o < OPL; o++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int c = clip3(-32768, 32767, (acc[e][o] + rnd) >> shift);
          uint16_t ax;
          B[(g * OPL + o) * N + 2 * rp + e] = (int16_t)quant_level_aux(c, qp, L2, 0, &ax, mtab ? mtab[(g * OPL + o) * N + 2 * rp + e] : 16);
          A[(g * OPL + o) * N + 2 * rp + e] = (int16_t)ax;
        }
      __syncthreads();
      adjust_block_lds(B, A, N, N, 0, adj & 1, adj & 2, l, LPT);
      __syncthreads();
#pragma unroll
      for (int o = 0; o < OPL; o++)
#pragma unroll
        for (int e = 0; e < 2; e++) lv[e][o] = B[(g * OPL + o) * N + 2 * rp + e];
      __syncthreads();                                           // (A still holds remainders other threads' groups have read: all done now)
    }
#pragma unroll
    for (int o = 0; o < OPL; o++)
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int m = mtab ? mtab[(g * OPL + o) * N + 2 * rp + e] : 16;
        if (!adj) { int c = clip3(-32768, 32767, (acc[e][o] + rnd) >> shift); lv[e][o] = quant_level(c, qp, L2, 0, m); }
        any |= lv[e][o] != 0;
        if constexpr (RC) *cost += rc_level_cost(lv[e][o]);
        A[(2 * rp + e) * N + g * OPL + o] = (int16_t)dequant_coef(lv[e][o], qp, L2, m);     // transposed: [column][row]
      }
    if (any) atomicOr(nz, 1u << tu);
    __syncthreads();
    if ((*nz >> tu) & 1) {
#pragma unroll
      for (int o = 0; o < OPL; o++) *(uint32_t *)coef_at(tu, g * OPL + o, 2 * rp) = pack_i16(lv[0][o], lv[1][o]);   // row g*OPL+o, columns 2rp, 2rp+1
    }
  }
  const bool has = (*nz >> tu) & 1;
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

// The 32x32 luma block of a 32x32 CU: same contract as inter_transform<DEC, 5, .>
template <bool DEC, bool RC = false, class QL = NoLateQp>
__device__ __forceinline__ void inter_transform_32(InterLds &s, int qp, uint32_t *nz, int16_t *coef, int cw, int tid, int adj = 0, QL qp_late = QL(), uint32_t *cost = nullptr, const uint8_t *mtab = nullptr)
{
  const int wave = tid >> 6, lane = tid & 63;
  const int j = (wave & 1) * 16 + (lane & 15), i0 = (wave >> 1) * 16 + (lane >> 4) * 4;   // result column, first of four result rows
  int lv[4];
  if (!DEC) {
    mfma_stage(s.A, s.B, s.M8[0], s.rowsum[0], 4, wave, lane);                             // forward rows (shift log2 n - 1)
    __syncthreads();
    {
      int acc[4];
      mfma_tile_sums(s.B, s.M8[0], s.rowsum[0], wave, lane, acc);                          // forward columns: coefficient (row j, columns i0 ..)
      if constexpr (RC) qp = qp_late();
      bool any = false;
      if (adj) {                                                                           // rdoq / signhide: see inter_transform
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int c = clip3(-32768, 32767, (acc[r] + 1024) >> 11);
          uint16_t ax;
          s.B[j * 32 + i0 + r] = (int16_t)quant_level_aux(c, qp, 5, 0, &ax, mtab ? mtab[j * 32 + i0 + r] : 16);
          s.A[j * 32 + i0 + r] = (int16_t)ax;
        }
        __syncthreads();
        adjust_block_lds(s.B, s.A, 32, 32, 0, adj & 1, adj & 2, tid, 256);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) lv[r] = s.B[j * 32 + i0 + r];
        __syncthreads();
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = mtab ? mtab[j * 32 + i0 + r] : 16;
        if (!adj) { const int c = clip3(-32768, 32767, (acc[r] + 1024) >> 11); lv[r] = quant_level(c, qp, 5, 0, m); }
        any |= lv[r] != 0;
        if constexpr (RC) *cost += rc_level_cost(lv[r]);
        s.A[(i0 + r) * 32 + j] = (int16_t)dequant_coef(lv[r], qp, 5, m);                   // transposed: [column][row]
      }
      if (any) atomicOr(nz, 1u);
    }
    __syncthreads();
    if (*nz & 1) *(uint2 *)&coef[(size_t)j * cw + i0] = make_uint2(pack_i16(lv[0], lv[1]), pack_i16(lv[2], lv[3]));
  }
  const bool has = *nz & 1;
  if (has) mfma_stage(s.A, s.B, s.M8[1], s.rowsum[1], 7, wave, lane);                      // inverse columns
  __syncthreads();
  if (has) {
    int acc[4];
    mfma_tile_sums(s.B, s.M8[1], s.rowsum[1], wave, lane, acc);                            // inverse rows: residual (rows i0 .., column j)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint8_t *q = &s.px[(i0 + r) * 32 + j];
      *q = (uint8_t)clip8(*q + ((acc[r] + 2048) >> 12));
    }
  }
  __syncthreads();
}

// Rate control v2, the tail of a workgroup of k_inter_recon<.., RC>: its level cost joins its group's (one atomic: arrivals << 40 | cost), and the workgroup
// that completes the group decides for the next one (statement: rc_band_decide() in oracle/hevc_enc.c) -- prices the rows done against their share of the
// picture's target, moves the running QP step and publishes it -- the step of the group AFTER THE NEXT (round 4: one group of lag, so that no group waits
// for the one right in front of it) -- in RcState::decided: bits 0..3 the number of groups closed, bits 4 + 3 (g - 1) .. the step of group g, biased by 3.  The per-CTU QP array is left alone while workgroups may still read it: the workgroup that completes the LAST group applies
// every group's step to it (for k_qp_first / k_qp_chain, deblocking and k_intra_recon) and files the picture's cost for k_rc_begin of a later picture.
__device__ __forceinline__ int rc_word_step(uint32_t word, int grp) { return grp > 0 ? (int)((word >> (4 + 3 * (grp - 1))) & 7u) - 3 : 0; }
__device__ __forceinline__ void rc_group_done(const EncFrame &f, InterLds &s, int grp, uint32_t cost, int tid)
{
  RcState *rc = f.rc;
  const int rows = f.ch >> 6, wc = f.cw >> 6, nb = f.rc_nb;
  const int r0 = (grp * rows) / nb, r1 = ((grp + 1) * rows) / nb;
  cost = wave_sum_u32(cost);
  if ((tid & 63) == 0) s.rc_part[tid >> 6] = cost;
  __syncthreads();
  if (tid == 0) {
    const unsigned long long mine = (unsigned long long)s.rc_part[0] + s.rc_part[1] + s.rc_part[2] + s.rc_part[3];
    const unsigned long long old = atomicAdd(&rc->acc[grp * KVZ_RC_ACC_STRIDE], (1ull << 40) | mine);
    const bool last = (old >> 40) == (unsigned long long)((r1 - r0) * 2 * (f.cw >> 5)) - 1ull;
    uint32_t word = 0;
    if (last) {
      // The groups are processed in order: cost_sofar and the word are what the group before this one left -- whose workgroups this group did NOT wait
      // for (a group waits for the one before the last, see k_inter_recon), so the one thread that closes this group does (rarely long: that group
      // started earlier).
      if (grp > 0) { uint32_t spins = 0; while ((ld_l2_u32(&rc->decided) & 15u) < (uint32_t)grp) { __builtin_amdgcn_s_sleep(8); if (++spins > (1u << 22)) { atomicOr(f.err, 1u); break; } } }
      const uint32_t total = ld_l2_u32(&rc->cost_sofar) + (uint32_t)((old & ((1ull << 40) - 1)) + mine);
      word = ld_l2_u32(&rc->decided);
      if (grp + 1 < nb) {
        // the step of group grp + 2: the running step (that of group grp + 1, decided a group ago; groups 0 and 1: none) moved by what groups 0 .. grp cost
        if (grp == 0) word |= 3u << 4;                        // (group 1's step, biased: 0)
        if (grp + 2 < nb) {
          int off = rc_word_step(word, grp + 1);
          if (ld_l2_u32(&rc->ratio_valid)) {
            const unsigned long long est = ((unsigned long long)total * ld_l2_u32(&rc->ratio_q8)) >> 8, tgt = ((unsigned long long)f.rc_target * (unsigned long long)r1) / (unsigned long long)rows;
            if (est * 8 > tgt * 9) off++; else if (est * 8 < tgt * 7) off--;
            off = clip3(-3, 3, off);
          }
          word |= (uint32_t)(off + 3) << (4 + 3 * (grp + 1));
        }
        st_wt_u32(&rc->cost_sofar, total);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        word = (word & ~15u) | (uint32_t)(grp + 1);
        st_wt_u32(&rc->decided, word);
      } else { rc->cost[f.rc_slot] = total; rc->cost_valid[f.rc_slot] = 1; }
    }
    s.rc_last = (last && grp + 1 == nb) ? 1 : 0; s.rc_qp = (int)word;
  }
  __syncthreads();
  if (!s.rc_last) return;
  // every other workgroup of the launch is done: the steps go into the QP array
  const uint32_t word = (uint32_t)s.rc_qp;
  int8_t *qt = const_cast<int8_t *>(f.ctu_qt);
  for (int g = 1; g < nb; g++) {
    const int off = rc_word_step(word, g);
    if (off) for (int i = ((g * rows) / nb) * wc + tid; i < (((g + 1) * rows) / nb) * wc; i += 256) qt[i] = (int8_t)clip3(0, 51, (int)qt[i] + off);
  }
}

// DEC = false: encoder (residual from the source picture, levels written out).
// DEC = true: decoder (levels and cbf given, prediction + residual only).
// ADJ: rdoq / signhide -- a kernel of its own: the plain form fits eight workgroups per compute unit in 64 registers, and with the level adjustment compiled
// in it no longer did (159 registers spilled, 14 MB of scratch writes per 1080p launch, 23 -> 28 us -- found in the PMC pass, not in a test).
// RC (encoder, rate control v2): the CTU rows in f.rc_nb groups, the QP of a group decided by the last workgroup of the group before it.  Workgroups are
// dispatched in the order of their linear index (rows top to bottom here: no XCD permutation), so every workgroup of a group is resident or done when one of
// the next group starts to wait -- the wait cannot starve what it waits for (the intra chains' argument).
// LL: `lossless` (cu_transquant_bypass): the residual itself goes to the level planes, sample by sample, and the reconstruction is the source
template <bool DEC, bool FRAC, bool ADJ = false, bool RC = false, bool LL = false>
// waves per SIMD the fractional / rate-control forms are held to: 7 (72 registers, one spilled in the rate-control form) -- their 22 KB of LDS allow seven workgroups
// per compute unit, 78 registers only six; default mode 5 450-5 530 -> 5 580-5 660 frames/s, the launch 51.3 -> 49.3 us (profiles/r06_recon_waves_ab.txt).  The forms
// with the level-adjustment pass (ADJ: rdoq / signhide) keep 5: they spill at anything tighter.
#ifndef KVZ_RECON_WAVES
#define KVZ_RECON_WAVES 7
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((FRAC || ADJ || RC) ? (ADJ ? 5 : KVZ_RECON_WAVES) : 8))) void k_inter_recon(EncFrame f)
{
  __shared__ InterLds s;
  InterFracLds *fr = nullptr;
  if constexpr (FRAC) { __shared__ InterFracLds fr_s; fr = &fr_s; }
  const int tid = threadIdx.x;
  int bx_, by_;
  if (RC) { bx_ = (int)blockIdx.x; by_ = (int)blockIdx.y; } else xcd_block_2d(bx_, by_);
  const int x0 = bx_ * 32, y0 = by_ * 32 + f.row0 * 64;
  const int bi0 = b8idx(f, x0, y0);
  const bool split = f.cu_log2[bi0] != 5;                              // (four 16x16 units: with intra-in-P a quarter may also be an intra unit, 16x16 or four 8x8)
  const int cw2 = f.cw >> 1, ch2 = f.ch >> 1;
  int grp = 0;                                                         // RC: the group of CTU rows this block belongs to: rows [g * rows / nb, (g + 1) * rows / nb)
  if (RC) { const int rows = f.ch >> 6; while (grp + 1 < f.rc_nb && (y0 >> 6) >= ((grp + 1) * rows) / f.rc_nb) grp++; }
  int qp = ctu_quant_qp(f, x0, y0), qpc = kChromaQp[qp];             // the 32x32 block lies inside one CTU (RC, groups behind the first: before the group's step)
  uint32_t rc_cost = 0;
  // RC: called by the luma transform in front of its quantiser -- the CTU's QP once the group before this one has decided (rc_group_done): the word
  // that counts the decided groups carries their QP steps as well, so the wait costs one trip to memory
  auto qp_late = [&]() -> int {
    if (RC && grp > 1) {                                             // (groups 0 and 1 run at the picture's QP; group g's step was decided when group g - 2 closed)
      if (tid == 0) {
        uint32_t spins = 0, v;
        while (((v = ld_l2_u32(&f.rc->decided)) & 15u) + 1u < (uint32_t)grp) {
          __builtin_amdgcn_s_sleep(24);                                  // (~0.6 us: hundreds of workgroups poll this word, and the polls travel to memory)
          if (++spins > (1u << 22)) { atomicOr(f.err, 1u); break; }      // bounded spin: never hang the GPU
        }
        s.rc_qp = clip3(0, 51, qp + rc_word_step(v, grp));
      }
      __syncthreads();
      qp = s.rc_qp; qpc = kChromaQp[qp];
    }
    return qp;
  };
  const int adj = (DEC || !ADJ) ? 0 : ((f.rdoq ? 1 : 0) | (f.signhide ? 2 : 0));    // level adjustment behind the quantiser (hevc_core.h adjust_group)
  // Decoder: most blocks of an inter picture carry no residual at all -- they skip the matrices, the transform stages and all
  // but two barriers (prediction straight to the picture).  `coded` is uniform over the workgroup.
  const bool coded = DEC ? __syncthreads_or(tid < 16 ? (int)(f.cu_cbf[b8idx(f, x0 + (tid & 3) * 8, y0 + (tid >> 2) * 8)] & 7) : 0) != 0 : true;
  // matrices of the two block sizes in use (luma n, chroma n / 2): adjacent in the table
  if (!coded) { }
  else if (split) load_matrices(s.M, 16, 64 + 256, tid, 256);
  else {
    load_matrices(s.M, 80, 256, tid, 256);                       // chroma 16-point matrices (dot2 path)
    if (tid >= 64 && tid < 192) ((uint4 *)s.M8)[tid - 64] = ((const uint4 *)g_xf.M8)[tid - 64];
    else if (tid >= 192 && tid < 208) ((uint4 *)s.rowsum)[tid - 192] = ((const uint4 *)g_xf.rowsum)[tid - 192];
  }
  if (tid < 2) s.nz[tid] = 0;
  if (tid < 4) {
    int bi = b8idx(f, x0 + (tid & 1) * 16, y0 + (tid >> 1) * 16);
    s.mv[tid][0] = f.cu_mv[bi * 2]; s.mv[tid][1] = f.cu_mv[bi * 2 + 1];
    // intra-in-P: a quarter that is an intra unit gets no residual here (levels, cbf and reconstruction are k_intra_recon's, which runs behind this kernel)
    s.intra_q[tid] = (!DEC && f.intra_p && split) ? f.cu_intra[bi] : 0;
  }
  __syncthreads();
  // ---- luma with fractional vectors (the encoder's with subme > 0): window -> LDS, horizontal pass -> LDS
  const bool anyfrac = FRAC && (((s.mv[0][0] | s.mv[0][1] | s.mv[1][0] | s.mv[1][1] | s.mv[2][0] | s.mv[2][1] | s.mv[3][0] | s.mv[3][1]) & 3) != 0);
  if (anyfrac) {
    for (int i = tid; i < 4 * 23 * 23; i += 256) {
      const int k = i / 529, r = i - k * 529, wy = r / 23, wx = r - wy * 23;
      const int gx = x0 + (k & 1) * 16 + (s.mv[k][0] >> 2) - 3 + wx, gy = y0 + (k >> 1) * 16 + (s.mv[k][1] >> 2) - 3 + wy;
      fr->lwin[k][wy * 24 + wx] = f.ref[0][(size_t)clip3(0, f.ch - 1, gy) * f.cw + clip3(0, f.cw - 1, gx)];
    }
    __syncthreads();
    for (int i = tid; i < 4 * 23 * 16; i += 256) {
      const int k = i / 368, r = i - k * 368, wy = r >> 4, c = r & 15, xf = s.mv[k][0] & 3;
      const uint8_t *wp = &fr->lwin[k][wy * 24 + c];
      int v = wp[3];
      if (xf) { v = 0; for (int t = 0; t < 8; t++) v += kLumaFilter[xf][t] * wp[t]; }
      fr->ltmp[k][wy * 16 + c] = v;
    }
    __syncthreads();
  }
  // ---- luma: prediction (four samples per thread), residual / dequantised levels TU-major into s.A
  {
    const int y = tid >> 3, x = (tid & 7) * 4, k = (y >> 4) * 2 + (x >> 4);
    const int l2 = split ? 4 : 5, n = 1 << l2, tu = split ? k : 0;
    uint32_t p4 = 0;
    if (anyfrac) {                             // (block-uniform) separable 8-tap interpolation through LDS
      const int mvx = s.mv[k][0], mvy = s.mv[k][1], xf = mvx & 3, yf = mvy & 3;
      const int *tp = &fr->ltmp[k][(y & 15) * 16 + (x & 15)];
#pragma unroll 1
      for (int i = 0; i < 4; i++) {            // (kept rolled: the unrolled form came out wrong for the third and fourth sample with hipcc 7.2)
        int v;
        if (yf) {
          int a = 0;
          for (int j = 0; j < 8; j++) a += (int)kLumaFilter[yf][j] * tp[j * 16 + i];
          v = xf ? (a >> 6) : a;
        } else v = xf ? tp[3 * 16 + i] : tp[3 * 16 + i] * 64;
        p4 |= (uint32_t)clip8((v + 32) >> 6) << (8 * i);
      }
    } else p4 = mc_luma4(f.ref[0], f.cw, f.ch, x0 + x, y0 + y, s.mv[k][0], s.mv[k][1]);
    const size_t g = (size_t)(y0 + y) * f.cw + x0 + x;
    if (!coded) *(uint32_t *)&f.rec[0][g] = p4;
    *(uint32_t *)&s.px[y * 32 + x] = p4;
    int16_t *A = s.A + (tu << (2 * l2));
    const int ty = y & (n - 1), tx = x & (n - 1);
    if (DEC) {
      const bool has = f.cu_cbf[b8idx(f, x0 + x, y0 + y)] & 1;
      if (has) {
        atomicOr(&s.nz[0], 1u << tu);
        const uint2 l4 = *(const uint2 *)&f.coef[0][g];
        const int lv[4] = {(int16_t)(l4.x & 0xffff), (int16_t)(l4.x >> 16), (int16_t)(l4.y & 0xffff), (int16_t)(l4.y >> 16)};
        for (int i = 0; i < 4; i++) A[(tx + i) * n + ty] = (int16_t)dequant_coef(lv[i], qp, l2);
      }
    } else {
      const uint32_t s4 = *(const uint32_t *)&f.src[0][g];
      int r[4];
      for (int i = 0; i < 4; i++) r[i] = s.intra_q[k] ? 0 : (int)((s4 >> (8 * i)) & 255) - (int)((p4 >> (8 * i)) & 255);
      if constexpr (LL) {
        *(uint2 *)&f.coef[0][g] = make_uint2(pack_i16(r[0], r[1]), pack_i16(r[2], r[3]));
        if (r[0] | r[1] | r[2] | r[3]) atomicOr(&s.nz[0], 1u << tu);
        if (!s.intra_q[k]) *(uint32_t *)&s.px[y * 32 + x] = s4;
      } else *(uint2 *)&A[ty * n + tx] = make_uint2(pack_i16(r[0], r[1]), pack_i16(r[2], r[3]));
    }
  }
  // ---- chroma: reference windows of the eight 8x8 sub-blocks (plane x quadrant) -> LDS; mv in 1/8 samples
  auto chroma_windows = [&]() {
    for (int i = tid; i < 8 * 121; i += 256) {
      const int w8 = i / 121, r = i - w8 * 121, wy = r / 11, wx = r - wy * 11, pl = w8 >> 2, sub = w8 & 3, k = split ? sub : 0;
      const int xi = (x0 >> 1) + (sub & 1) * 8 + (s.mv[k][0] >> 3) + wx - 1, yi = (y0 >> 1) + (sub >> 1) * 8 + (s.mv[k][1] >> 3) + wy - 1;
      s.win[pl][sub][wy * 12 + wx] = f.ref[1 + pl][(size_t)clip3(0, ch2 - 1, yi) * cw2 + clip3(0, cw2 - 1, xi)];
    }
  };
  // RC: everything chroma reads from memory is fetched in front of the luma transform, where a workgroup may wait for its QP -- behind the wait it is on the
  // path from one group's decision to the next
  uint32_t src_c = 0;
  if (RC) {
    chroma_windows();
    src_c = *(const uint16_t *)&f.src[1 + (tid >> 7)][(size_t)((y0 >> 1) + ((tid >> 3) & 15)) * cw2 + (x0 >> 1) + (tid & 7) * 2];
  }
  if (coded) {
    __syncthreads();
    auto px16 = [](int tu, int y, int x) { return ((tu >> 1) * 16 + y) * 32 + (tu & 1) * 16 + x; };
    const int cw = f.cw; int16_t *base = f.coef[0] + (size_t)y0 * cw + x0;
    auto ci16 = [=](int tu, int y, int x) { return base + (size_t)((tu >> 1) * 16 + y) * cw + (tu & 1) * 16 + x; };
    if constexpr (LL) { }
    else if (split) inter_transform<DEC, 4, 2, RC>(s, qp, &s.nz[0], px16, ci16, tid, adj, qp_late, &rc_cost, f.scaling ? f.scaling + scaling_offset(4, 0, 1) : nullptr, 4);
    else inter_transform_32<DEC, RC>(s, qp, &s.nz[0], base, cw, tid, adj, qp_late, &rc_cost, f.scaling ? f.scaling + scaling_offset(5, 0, 1) : nullptr);
    *(uint32_t *)&f.rec[0][(size_t)(y0 + (tid >> 3)) * f.cw + x0 + (tid & 7) * 4] = *(const uint32_t *)&s.px[(tid >> 3) * 32 + (tid & 7) * 4];
  }
  if (!RC) chroma_windows();
  __syncthreads();                           // (also: every thread has copied its luma samples out of s.px)
  {
    // two samples per thread; the separable 4-tap form with the {0, 64, 0, 0} filter at fraction 0 covers every case of
    // 8.5.3.3.3.2 exactly: (64 * t) >> 6 == t
    const int pl = tid >> 7, y = (tid >> 3) & 15, x = (tid & 7) * 2, sub = (y >> 3) * 2 + (x >> 3), k = split ? sub : 0;
    const int xf = s.mv[k][0] & 7, yf = s.mv[k][1] & 7;
    const uint8_t *w = &s.win[pl][sub][(y & 7) * 12 + (x & 7)];
    int v0 = 0, v1 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int c[5];
#pragma unroll
      for (int i = 0; i < 5; i++) c[i] = w[j * 12 + i];
      int t0 = 0, t1 = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) { t0 += kChromaFilter[xf][i] * c[i]; t1 += kChromaFilter[xf][i] * c[i + 1]; }
      v0 += kChromaFilter[yf][j] * t0; v1 += kChromaFilter[yf][j] * t1;
    }
    const int p0 = clip8(((v0 >> 6) + 32) >> 6), p1 = clip8(((v1 >> 6) + 32) >> 6);
    *(uint16_t *)&s.px[pl * 256 + y * 16 + x] = (uint16_t)(p0 | (p1 << 8));
    const int cl2 = split ? 3 : 4, cn = 1 << cl2, tu = pl * (split ? 4 : 1) + (split ? sub : 0);
    const size_t g = (size_t)((y0 >> 1) + y) * cw2 + (x0 >> 1) + x;
    if (!coded) *(uint16_t *)&f.rec[1 + pl][g] = (uint16_t)(p0 | (p1 << 8));
    int16_t *A = s.A + (tu << (2 * cl2));
    const int ty = y & (cn - 1), tx = x & (cn - 1);
    if (DEC) {
      const bool has = (f.cu_cbf[b8idx(f, x0 + 2 * x, y0 + 2 * y)] >> (1 + pl)) & 1;
      if (has) {
        atomicOr(&s.nz[1], 1u << tu);
        const uint32_t l2v = *(const uint32_t *)&f.coef[1 + pl][g];
        A[tx * cn + ty] = (int16_t)dequant_coef((int16_t)(l2v & 0xffff), qpc, cl2);
        A[(tx + 1) * cn + ty] = (int16_t)dequant_coef((int16_t)(l2v >> 16), qpc, cl2);
      }
    } else {
      const uint32_t s2 = RC ? src_c : *(const uint16_t *)&f.src[1 + pl][g];
      const uint32_t r2 = s.intra_q[sub] ? 0u : pack_i16((int)(s2 & 255) - p0, (int)(s2 >> 8) - p1);
      if constexpr (LL) {
        *(uint32_t *)&f.coef[1 + pl][g] = r2;
        if (r2) atomicOr(&s.nz[1], 1u << tu);
        if (!s.intra_q[sub]) *(uint16_t *)&s.px[pl * 256 + y * 16 + x] = (uint16_t)s2;
      } else *(uint32_t *)&A[ty * cn + tx] = r2;
    }
  }
  if (!coded) return;                          // (decoder only; the encoder's cbf bookkeeping below never applies)
  __syncthreads();
  {
    auto px1 = [](int tu, int y, int x) { return tu * 256 + y * 16 + x; };
    auto px4 = [](int tu, int y, int x) { return (tu >> 2) * 256 + (((tu >> 1) & 1) * 8 + y) * 16 + (tu & 1) * 8 + x; };
    const size_t base = (size_t)(y0 >> 1) * cw2 + (x0 >> 1);
    int16_t *cb = f.coef[1] + base, *cr = f.coef[2] + base;
    auto ci1 = [=](int tu, int y, int x) { return (tu ? cr : cb) + (size_t)y * cw2 + x; };
    auto ci4 = [=](int tu, int y, int x) { return ((tu >> 2) ? cr : cb) + (size_t)(((tu >> 1) & 1) * 8 + y) * cw2 + (tu & 1) * 8 + x; };
    auto qpc_known = [&]() -> int { return qpc; };
    if constexpr (LL) { }
    else if (split) inter_transform<DEC, 3, 1, RC>(s, qpc, &s.nz[1], px4, ci4, tid, adj, qpc_known, &rc_cost, f.scaling ? f.scaling + scaling_offset(3, 1, 1) : nullptr, 4);
    else inter_transform<DEC, 4, 1, RC>(s, qpc, &s.nz[1], px1, ci1, tid, adj, qpc_known, &rc_cost, f.scaling ? f.scaling + scaling_offset(4, 1, 1) : nullptr, 1);
  }
  if (tid < 128) {
    const int pl = tid >> 6, y = (tid >> 2) & 15, x = (tid & 3) * 4;
    *(uint32_t *)&f.rec[1 + pl][(size_t)((y0 >> 1) + y) * cw2 + (x0 >> 1) + x] = *(const uint32_t *)&s.px[pl * 256 + y * 16 + x];
  }
  if (!DEC && tid < 16) {
    int bx = tid & 3, by = tid >> 2, k = split ? ((by >> 1) * 2 + (bx >> 1)) : 0;
    int cbf = (int)((s.nz[0] >> k) & 1);
    if (split) cbf |= (int)((s.nz[1] >> k) & 1) << 1 | (int)((s.nz[1] >> (4 + k)) & 1) << 2;
    else cbf |= (int)(s.nz[1] & 1) << 1 | (int)((s.nz[1] >> 1) & 1) << 2;
    f.cu_cbf[b8idx(f, x0 + bx * 8, y0 + by * 8)] = (uint8_t)cbf;
  }
  if constexpr (RC) rc_group_done(f, s, grp, rc_cost, tid);
}

// One thread per 8x8 unit: it derives the signalling of the CU it lies in and writes its OWN five entries -- every unit of a CU (4 of a 16x16, 16 of a
// 32x32) derives the same values from the same few bytes (cache hits), so nobody loops over a CU's units storing single bytes and a wave's stores are
// consecutive.  In workgroups of 1024: the kernel is ~1 600 instructions of straight-line code that every wave runs once, and what it costs is FETCHING that
// code -- a launch that only reads the size byte and writes the entries takes 6.0 us (the floor), the derivation with the same threads in workgroups of 64
// (510 of them, a wave or two on every compute unit, each fetching the code for itself) 31 us, in workgroups of 256 13.8 - 21 us, 512 10.9, 1024 10.0
// (profiles/r05_inter_signal_variants.txt; one box, isolated).  Round 4's form (a thread per 16x16 block, 128 workgroups of 64): 22 us; staging the records
// in LDS, one workgroup per CTU: 28 - 32 us (twice the waves, a barrier) -- it was never the round trips to memory.
__global__ __launch_bounds__(1024) void k_inter_signal(EncFrame f)
{
  const int w8 = f.cw >> 3, h8 = band_rows(f) * 8;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < w8 * h8;
  const int x = valid ? (i % w8) * 8 : 0, y = valid ? (i / w8) * 8 + f.row0 * 64 : 0, g = b8idx(f, x, y);
  const int cl = f.cu_log2[g];
  const bool intra = f.cu_intra[g] != 0;                // an intra unit in a P picture (intra-in-P)
  int flags = 0;
  if (valid && !intra) {
    const int n = 1 << cl;
    const CuSignal r = decide_signalling_values(f, x & ~(n - 1), y & ~(n - 1), cl);
    flags = r.flags;
    f.cu_flags[g] = (uint8_t)r.flags; f.cu_merge_idx[g] = (uint8_t)r.midx; f.cu_mvp_idx[g] = (uint8_t)r.mvp;
    *reinterpret_cast<uint32_t *>(&f.cu_mvd[g * 2]) = ((uint32_t)r.mvdx & 0xffffu) | ((uint32_t)r.mvdy << 16);
  }
  // The tokenizer's work list (EncFrame::tok_list): of a P picture's 4 x units (unit, role) pairs nine in ten have nothing to say -- units inside a 32x32 coding
  // unit that starts elsewhere, colour components without residual -- and a wave each to find that out WAS k_tokenize's launch (32 640 waves at 1080p, 130 000
  // at 2160p).  The thread at a 16x16 unit's origin knows all of it: the unit's coding unit(s), this very thread's skip decision, the cbf bits (final: the
  // reconstruction is behind us).  Roles: 0 luma, 1 Cb, 2 Cr, 3 headers (and, for the CTU's last unit, its terminating bins).  One atomic per wave.
  if (f.tok_list) {
    uint32_t roles = 0;
    if (valid && !((x | y) & 15)) {
      const bool owner = !(cl == 5 && ((x | y) & 31));
      if (owner || ((x & 63) == 48 && (y & 63) == 48)) roles |= 8u;
      if (owner) {
        uint32_t cbf = 0;
        if (cl == 3) { cbf = f.cu_cbf[g] | f.cu_cbf[g + 1] | f.cu_cbf[g + w8] | f.cu_cbf[g + w8 + 1]; }      // four 8x8 coding units (intra units: never skipped)
        else if (!(flags & CU_SKIP)) cbf = f.cu_cbf[g];
        roles |= cbf & 7u;
      }
    }
    const int lane = threadIdx.x & 63, nr = __builtin_popcount(roles);
    int pre = nr;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o); if (lane >= o) pre += t; }
    const int total = __shfl(pre, 63);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(&f.tok_list[0], (uint32_t)total);
    base = (uint32_t)__shfl((int)base, 63);
    uint32_t o = base + (uint32_t)(pre - nr);
    const uint32_t unit = (uint32_t)((y >> 4) * (f.cw >> 4) + (x >> 4));
    for (uint32_t m = roles; m; m &= m - 1) f.tok_list[1 + o++] = (unit << 2) | (uint32_t)__builtin_ctz(m);
  }
}

// =============================================================================================
// Intra reconstruction.  Intra prediction of a block needs the reconstructed samples of its left /
// above neighbours, so inside one colour plane the blocks of a picture form a dependency chain:
// z-order inside a CTU, a two-CTU lag between CTU rows.  What is independent is the three colour
// planes (DM chroma uses the luma *mode*, not luma samples).  So: one WAVE per (CTU row, plane),
// launched as a 64-thread workgroup; nothing inside the chain ever waits for another wave, the
// steps of a block talk through LDS under wave-local barriers, and rows hand over through agent
// -scope progress counters (one per row and plane).  The CTU being coded, its borders, its source
// samples and its levels stay in LDS; global memory is touched once per CTU on the way in and out.
//
// Transforms: one primitive, P(X, T)[j][i] = sum_m X[i][m] * T[j][m] (rows of X times rows of T,
// result stored transposed), run four times -- forward rows, forward columns, inverse columns,
// inverse rows -- on int16 data with v_dot2_i32_i16.  A lane owns a pair of rows of X and OPL
// outputs of each, so the matrix rows it reads are shared by both rows and every LDS write is a
// packed pair.  Quantisation and dequantisation are the epilogue of the second forward stage, the
// reconstruction is the epilogue of the last inverse stage.
// =============================================================================================
#ifdef KVZ_PROF     // tools/intra_prof.py: cycle attribution inside the chain (never defined in the product build)
__shared__ long long g_prof[16];
#define PROF(k) do { if (threadIdx.x == 0) { long long t_ = clock64(); g_prof[k] += t_ - g_prof[15]; g_prof[15] = t_; } } while (0)
#else
#define PROF(k) do { } while (0)
#endif
struct IntraCtuLds {
  // The CTU being reconstructed with its borders, as one padded picture: row 0 = the sample row above the CTU
  // (above-left corner, above, above-right), column 15 = the sample column to its left, sample (x, y) of the
  // CTU at pic[(y + 1) * P + 16 + x], P = 16 + 2S.
  alignas(16) uint8_t pic[65 * 144];
  alignas(16) uint8_t src[64 * 64];          // source samples of the CTU
  alignas(16) int16_t lev[64 * 64];          // levels produced
  alignas(16) XfLaneF16 xf[4][64];           // matrix operands of the transform stages, per (transform, lane)
};

// =============================================================================================
// Intra analysis (IDR pictures): for every 8x8 and 16x16 block of a 32x32 region the SAD of all 35
// prediction modes against the SOURCE picture, predictions built from source neighbours (so nothing depends
// on reconstruction and the whole picture is searched in parallel); then the bottom-up split decision.  Intra
// coding units are 16x16 or 8x8 (Kvazaar's ultrafast shape; oracle/hevc_enc.c intra_decide() has the reason).
// The reference arrays of all 20 blocks are built once into LDS (scan order, as in the reconstruction
// kernel); a work item is (block, mode), one wave each: the mode is wave-uniform, 64 samples per step, the
// SAD is summed with DPP row operations.
// =============================================================================================

struct AnalyseLds {
  alignas(16) uint8_t src[32 * 32];
  // per block b (0..15: 8x8 raster, 16..19: 16x16 raster, 20: 32x32): unfiltered / filtered references, scan order, at
  // R[f][roff(b)]; sizes 33 / 65 / 129 entries
  alignas(16) uint8_t R[2][16 * 36 + 4 * 68 + 132];
  int dc[21];
  uint32_t cost[21][35];
  uint32_t bestc[21]; int bestm[21];
  int last;                                // intra-in-P: this workgroup is the last of its region's to arrive
  alignas(16) uint8_t srcT[32 * 32];       // src transposed (the horizontal modes' tiles: analyse_tile_satd)
  alignas(16) uint8_t ext[16][4 * 32];     // per wave: the reference samples of the item's block(s) as the array its mode walks along (analyse_tile_satd)
};
__device__ __forceinline__ int an_tile_of(int b) { return b < 16 ? (b >> 3) * 2 + ((b >> 1) & 1) : b - 16; }      // the 16x16 quarter block b (0..19) lies in
__device__ __forceinline__ int an_roff(int b) { return b < 16 ? b * 36 : (b < 20 ? 16 * 36 + (b - 16) * 68 : 16 * 36 + 4 * 68); }

template <int L2>
__device__ __forceinline__ uint32_t analyse_item(const AnalyseLds &s, int b, int bx, int by, int mode, int lane)
{
  constexpr int N = 1 << L2;
  const bool filt = intra_filter_needed(N, 0, mode);
  const uint8_t *R = s.R[filt ? 1 : 0] + an_roff(b);
  const int angle = kIntraAngle[mode], inv = kInvAngle[mode];
  const bool edge = N < 32, vert = mode >= 18, e2 = edge && (mode == 26 || mode == 10);
  const int dcv = s.dc[b];
  uint32_t sad = 0;
#pragma unroll
  for (int it = 0; it < N * N / 64; it++) {
    const int pxi = it * 64 + lane, y = pxi >> L2, x = pxi & (N - 1);
    int p;
    if (mode == 0) p = pred_planar<L2>(R, x, y);
    else if (mode == 1) p = pred_dc<L2>(R, edge, dcv, x, y);
    else p = pred_angular<L2>(R, vert, e2, angle, inv, x, y);
    sad += (uint32_t)iabs((int)s.src[(by + y) * 32 + bx + x] - p);
  }
  return wave_sum_u32(sad);
}

// SATD form of the search cost (f.satd): one item = (16x16 tile of the region, mode, kind) on one wave.  kind 0 predicts the tile as
// one 16x16 block, kind 1 as its four 8x8 blocks; either way the tile's difference to the source goes through the 8x8 Hadamard
// transform of each of its quadrants as ONE pair of matrix-core products: with H16 = H8 (+) H8 (block diagonal, entries +-1)
//   Y = D H16^T    (A = the differences, lane (g, c) = (lane >> 4, lane & 15) owns x = 4g .. 4g + 3 of row y = c)
//   Z = H16 Y      (B = Y, which the first product leaves in exactly the operand layout the second one reads)
// on v_mfma_f32_16x16x16_f16 -- exact: |D| <= 255, |Y| <= 2040 < 2^11, |Z| <= 16320.  q[k] = sum |Z| over quadrant k (raster).
__device__ __forceinline__ void analyse_tile_satd(AnalyseLds &s, int wave, int tile, int kind, int mode, int lane, uint32_t (&q)[4])
{
  const int g = lane >> 4, c = lane & 15, tx = (tile & 1) * 16, ty = (tile >> 1) * 16;
  int d[4];
  if (mode < 2) {
    // planar and DC (2 of 35 items): sample by sample
    const uint32_t s4 = *(const uint32_t *)&s.src[(ty + c) * 32 + tx + 4 * g];
    if (kind == 0) {
      const int b = 16 + tile;
      const uint8_t *R = s.R[intra_filter_needed(16, 0, mode) ? 1 : 0] + an_roff(b);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int x = 4 * g + r, y = c;
        const int p = mode == 0 ? pred_planar<4>(R, x, y) : pred_dc<4>(R, true, s.dc[b], x, y);
        d[r] = (int)((s4 >> (8 * r)) & 255u) - p;
      }
    } else {
      const int b = ((ty >> 3) + (c >> 3)) * 4 + (tx >> 3) + (g >> 1);
      const uint8_t *R = s.R[intra_filter_needed(8, 0, mode) ? 1 : 0] + an_roff(b);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int x = (4 * g + r) & 7, y = c & 7;
        const int p = mode == 0 ? pred_planar<3>(R, x, y) : pred_dc<3>(R, true, s.dc[b], x, y);
        d[r] = (int)((s4 >> (8 * r)) & 255u) - p;
      }
    }
  } else {
    // The 33 angular modes, all in the VERTICAL form: a horizontal mode predicts the transposed block from the left column the way a vertical mode predicts
    // the block from the row above (8.4.4.2.6 is symmetric in x and y), and the Hadamard cost of a transposed difference block is the cost of the block
    // (H D' H = (H D H)' for the symmetric H) -- so the tile is taken from the transposed copy of the source, and the quadrant sums q[1], q[2] change places
    // at the end.  Per item the wave first lays the block's reference samples out as the ONE array the mode walks along, ext[N + k] = ref[k] for
    // k = -N .. 2N + 1 (negative k: the other side's samples projected with the inverse angle), N bytes in front of the corner; then lane (g, c) -- row c,
    // samples 4g .. 4g + 3 -- needs ONE offset and ONE fraction (both depend on the row only), five consecutive bytes of ext (one aligned 8-byte read, a
    // byte shift) and four blends: no branch, no sign test per sample, no multiplication by the inverse angle outside the lay-out.
    const int ai = kAngInv.v[mode], angle = (int)(int16_t)(ai & 0xffff), inv = ai >> 16;
    const bool vert = mode >= 18, e2 = mode == 26 || mode == 10;
    const int sgn = vert ? 1 : -1;
    uint8_t *ext = s.ext[wave];
    const uint32_t s4 = vert ? *(const uint32_t *)&s.src[(ty + c) * 32 + tx + 4 * g] : *(const uint32_t *)&s.srcT[(tx + c) * 32 + ty + 4 * g];
    wave_sync();                                              // (the last item's reads of ext are done)
    int row, x0, N;
    const uint8_t *Rl, *el;                                   // the lane's block: its reference array, its ext
    if (kind == 0) {
      const uint8_t *R = s.R[intra_filter_needed(16, 0, mode) ? 1 : 0] + an_roff(16 + tile);
      const int k = lane - 16;
      const int idx = k >= 0 ? imin(k, 32) : imax(-((k * inv + 128) >> 8), -32);
      ext[lane] = R[32 + sgn * idx];
      row = c; x0 = 4 * g; N = 16; Rl = R; el = ext;
    } else {
      const uint8_t *R0 = s.R[intra_filter_needed(8, 0, mode) ? 1 : 0];
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int e = lane + 64 * hh, slot = e >> 5, k = (e & 31) - 8;       // slot: the quadrant (sy', sx') of the tile as the lanes see it
        const int sx = vert ? (slot & 1) : (slot >> 1), sy = vert ? (slot >> 1) : (slot & 1);
        const uint8_t *R = R0 + an_roff(((ty >> 3) + sy) * 4 + (tx >> 3) + sx);
        const int idx = k >= 0 ? imin(k, 16) : imax(-((k * inv + 128) >> 8), -16);
        ext[e] = R[16 + sgn * idx];
      }
      const int slot = (c >> 3) * 2 + (g >> 1);
      const int sx = vert ? (slot & 1) : (slot >> 1), sy = vert ? (slot >> 1) : (slot & 1);
      row = c & 7; x0 = (4 * g) & 7; N = 8; Rl = R0 + an_roff(((ty >> 3) + sy) * 4 + (tx >> 3) + sx); el = ext + slot * 32;
    }
    wave_sync();
    const int t = (row + 1) * angle, fr = t & 31, j0 = N + x0 + (t >> 5) + 1;
    const uint32_t *w = (const uint32_t *)(el + (j0 & ~3));
    const uint32_t lo = w[0], hi = w[1];
    const uint32_t sh = (uint32_t)(j0 & 3);
    const uint32_t b03 = __builtin_amdgcn_alignbyte(hi, lo, sh), b4 = (hi >> (8 * sh)) & 255u;
    int B[5] = {(int)(b03 & 255u), (int)((b03 >> 8) & 255u), (int)((b03 >> 16) & 255u), (int)(b03 >> 24), (int)b4};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int p = (32 * B[r] + fr * (B[r + 1] - B[r]) + 16) >> 5;
      d[r] = (int)((s4 >> (8 * r)) & 255u) - p;
    }
    if (e2 && x0 == 0) d[0] = (int)(s4 & 255u) - clip8(Rl[2 * N + sgn] + ((Rl[2 * N - sgn * (1 + row)] - Rl[2 * N]) >> 1));      // (8.4.4.2.6: the edge of the pure vertical / horizontal mode)
  }
  // H16[c][4g + r]: zero across the two 8x8 blocks, else the sign of the natural-ordered Hadamard matrix, (-1)^popcount(i & j)
  kv_f16x4 h;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = 4 * g + r;
    h[r] = ((c ^ j) & 8) ? (_Float16)0.f : ((__builtin_popcount((unsigned)(c & j & 7)) & 1) ? (_Float16)-1.f : (_Float16)1.f);
  }
  // Both products and the sum of magnitudes stay in floating point -- every value is an integer below 2^17, exact in f32, and Y (|y| <= 2040) is exact in
  // f16 --: the first product's result goes back in as packed halves (v_cvt_pkrtz: two per instruction, nothing to round), the magnitudes are source
  // modifiers of the adds, and ONE conversion per lane is left, after the lane sums (was 4 + 4 + 4 conversions and 8 integer abs steps per item: 154 -> 142 us).
  // (Tried and dropped: the 18 modes that never reach behind the corner reading their five reference bytes straight from R[] instead of laying out ext[] --
  // 142 -> 154 us: two code paths in the item loop cost more than the lay-out they save.)
  kv_f16x4 da;
#pragma unroll
  for (int r = 0; r < 4; r++) da[r] = (_Float16)(short)d[r];
  const kv_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const kv_f32x4 yf = __builtin_amdgcn_mfma_f32_16x16x16f16(da, h, zero, 0, 0, 0);
  typedef __fp16 kv_h2_ __attribute__((ext_vector_type(2)));
  const kv_h2_ y01 = __builtin_amdgcn_cvt_pkrtz(yf[0], yf[1]), y23 = __builtin_amdgcn_cvt_pkrtz(yf[2], yf[3]);
  struct { kv_h2_ lo, hi; } ypk = {y01, y23};
  const kv_f16x4 yb = __builtin_bit_cast(kv_f16x4, ypk);
  const kv_f32x4 zf = __builtin_amdgcn_mfma_f32_16x16x16f16(h, yb, zero, 0, 0, 0);
  float af = (__builtin_fabsf(zf[0]) + __builtin_fabsf(zf[1])) + (__builtin_fabsf(zf[2]) + __builtin_fabsf(zf[3]));
  // sums over the 8-lane groups (three DPP steps), then the two groups of each quadrant: output (u = 4g + r, v = c) lies in quadrant
  // (u >= 8) * 2 + (v >= 8)
  af += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, af), 0xB1, 0xf, 0xf, false));
  af += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, af), 0x4E, 0xf, 0xf, false));
  af += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, af), 0x141, 0xf, 0xf, false));
  const uint32_t a = (uint32_t)af;
  q[0] = (uint32_t)__builtin_amdgcn_readlane((int)a, 0) + (uint32_t)__builtin_amdgcn_readlane((int)a, 16);
  q[1] = (uint32_t)__builtin_amdgcn_readlane((int)a, 8) + (uint32_t)__builtin_amdgcn_readlane((int)a, 24);
  q[2] = (uint32_t)__builtin_amdgcn_readlane((int)a, 32) + (uint32_t)__builtin_amdgcn_readlane((int)a, 48);
  q[3] = (uint32_t)__builtin_amdgcn_readlane((int)a, 40) + (uint32_t)__builtin_amdgcn_readlane((int)a, 56);
  if (mode >= 2 && mode < 18) { const uint32_t t_ = q[1]; q[1] = q[2]; q[2] = t_; }      // (a horizontal mode's tile was transposed)
}

// PP = false: intra pictures.  PP = true ("uvgx intra-in-P v1"): launched behind k_me in a P picture; a region none of whose quarters'
// inter cost is above the gate leaves at once (nearly all of them), the others are analysed like an intra picture's and the quarters
// that come out cheaper as intra blocks are turned into intra units.
template <bool PP>
__global__ __launch_bounds__(PP ? 1024 : 256) void k_intra_analyse(EncFrame f)
{
  // (PP: few regions get past the gate, so what counts is how long ONE of them takes, not how many fit on the chip: sixteen waves share its items)
  constexpr int T = PP ? 1024 : 256, NW = T / 64;
  __shared__ AnalyseLds s;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: item, mode, kind and tile of the search loops are wave-uniform, and the compiler cannot know)
  int bx_ = 0, by_ = 0;
  if (!PP) xcd_block_2d(bx_, by_);
  // PP: a fixed number of workgroups works through the list of blocks k_me found above the gate (f.me_cand); the first things they do is
  // zero what k_intra_recon<.., true> starts from: the progress counters of every CTU and the ticket counter behind them
  const uint32_t ncand = PP ? f.me_cand[0] : 1u;
  if (PP) for (uint32_t i = blockIdx.x * T + tid; i < 3u * (uint32_t)(f.cw >> 6) * (uint32_t)(f.ch >> 6) + 1u; i += gridDim.x * T) f.sync[i] = 0;      // (the word behind the ticket counter says whether the picture has an intra unit at all: set below, zeroed by k_deblock_tile)
  // PP: a work item is one QUARTER of a listed block (a quarter's 70 (size, mode) prices over sixteen waves are a third of the latency of the block's 280:
  // 43 -> 16 us for the launch); the quarters' workgroups leave their best prices in f.ip_scratch, and the last of a block's to arrive takes the decision
  for (uint32_t item = PP ? blockIdx.x : 0u; item < (PP ? 4u * ncand : 1u); item += PP ? gridDim.x : 1u) {
  const int mytile = PP ? (int)(item & 3u) : -1;
  const uint32_t ci = PP ? item >> 2 : 0u;
  auto mine = [&](int b) { return !PP || an_tile_of(b) == mytile; };
  if (PP) { __syncthreads(); const uint32_t r = f.me_cand[1 + ci]; bx_ = (int)(r % (uint32_t)(f.cw >> 5)); by_ = (int)(r / (uint32_t)(f.cw >> 5)); }      // (the barrier: LDS of the last item is free)
  const int X0 = bx_ * 32, Y0 = by_ * 32 + f.row0 * 64;
  uint32_t icost[4] = {0, 0, 0, 0}; bool cand[4] = {false, false, false, false};
  if (PP) {
    for (int k = 0; k < 4; k++) {
      icost[k] = f.me_cost16[((Y0 >> 4) + (k >> 1)) * (f.cw >> 4) + (X0 >> 4) + (k & 1)];
      cand[k] = icost[k] > (uint32_t)INTRA_P_GATE * (uint32_t)f.lambda_q4;
    }
    if (!cand[mytile]) continue;                                // (only the quarters above the gate are priced)
  }
  const uint8_t *src = f.src[0];
  if (tid < 256) *(uint32_t *)&s.src[tid * 4] = *(const uint32_t *)&src[(size_t)(Y0 + (tid >> 3)) * f.cw + X0 + (tid & 7) * 4];
  // ---- references of the 21 blocks from the source picture (8.4.4.2.2 substitution as an index clamp: the available
  // groups are contiguous in scan order; availability by picture bounds and z-scan order)
  for (int e = tid; e < 16 * 33 + 4 * 65; e += T) {
    int b, i, l2;
    if (e < 16 * 33) { b = e / 33; i = e - b * 33; l2 = 3; } else { int r = e - 16 * 33; b = 16 + r / 65; i = r % 65; l2 = 4; }
    if (!mine(b)) continue;
    const int n = 1 << l2, bi = b < 16 ? b : b - 16, nb = 32 >> l2;
    const int x0 = X0 + (bi % nb) * n, y0 = Y0 + (bi / nb) * n;
    const bool aL = avail64(f.cw, f.chp, x0, y0, x0 - 1, y0), aT = avail64(f.cw, f.chp, x0, y0, x0, y0 - 1);
    const bool aBL = aL && avail64(f.cw, f.chp, x0, y0, x0 - 1, y0 + n), aTR = aT && avail64(f.cw, f.chp, x0, y0, x0 + n, y0 - 1);
    const int lo = aBL ? 0 : (aL ? n : 2 * n + 1), hi = aTR ? 4 * n : (aT ? 3 * n : (aL ? 2 * n - 1 : -1));
    int v = 128;
    if (hi >= 0) {
      const int j = imin(imax(i, lo), hi);
      const int x = j < 2 * n ? x0 - 1 : x0 + j - 2 * n - 1, y = j < 2 * n ? y0 + 2 * n - 1 - j : y0 - 1;
      v = src[(size_t)y * f.cw + x];
    }
    s.R[0][an_roff(b) + i] = (uint8_t)v;
  }
  __syncthreads();
  // ---- filtered references (8.4.4.2.3) and DC values; the source block transposed (analyse_tile_satd)
  if (f.satd && tid < 256) {
    const uint32_t v = *(const uint32_t *)&s.src[tid * 4];
    const int y = tid >> 3, x = (tid & 7) * 4;
#pragma unroll
    for (int r = 0; r < 4; r++) s.srcT[(x + r) * 32 + y] = (uint8_t)(v >> (8 * r));
  }
  for (int e = tid; e < 16 * 33 + 4 * 65; e += T) {
    int b, i, l2;
    if (e < 16 * 33) { b = e / 33; i = e - b * 33; l2 = 3; } else { int r = e - 16 * 33; b = 16 + r / 65; i = r % 65; l2 = 4; }
    if (!mine(b)) continue;
    const int n = 1 << l2;
    const uint8_t *R = s.R[0] + an_roff(b);
    int fv = R[i];
    if (i != 0 && i != 4 * n) {
      const bool strong = n == 32 && iabs(R[2 * n] + R[4 * n] - 2 * R[3 * n]) < 8 && iabs(R[2 * n] + R[0] - 2 * R[n]) < 8;
      if (strong) { if (i != 2 * n) { int k = i < 2 * n ? 2 * n - i : i - 2 * n; fv = ((64 - k) * R[2 * n] + k * (i < 2 * n ? R[0] : R[4 * n]) + 32) >> 6; } }
      else fv = (R[i - 1] + 2 * R[i] + R[i + 1] + 2) >> 2;
    }
    s.R[1][an_roff(b) + i] = (uint8_t)fv;
  }
  if (tid < 20 && mine(tid)) {
    const int l2 = tid < 16 ? 3 : 4, n = 1 << l2;
    const uint8_t *R = s.R[0] + an_roff(tid);
    int a = n;
    for (int i = n; i <= 3 * n; i++) a += R[i];
    s.dc[tid] = (a - R[2 * n]) >> (l2 + 1);
  }
  __syncthreads();
  // ---- cost of every (block, mode): 8x8 Hadamard sums, (sum + 2) >> 2 per 8x8 block (oracle/hevc_enc.c satd_block()), or SADs
  if (f.satd) {
    const bool no8 = PP && f.intra_p == 1;                          // intra-in-p = 1: 16x16 intra units only -- the 8x8 blocks are not priced
    for (int j = wave; j < (PP ? (no8 ? 35 : 35 * 2) : 4 * 35 * 2); j += NW) {
      const int item = PP ? (no8 ? (j << 3) | (mytile << 1) : ((j >> 1) << 3) | (mytile << 1) | (j & 1)) : j;
      const int kind = item & 1, tile = (item >> 1) & 3, mode = item >> 3;
      uint32_t q[4];
      analyse_tile_satd(s, wave, tile, kind, mode, lane, q);
      if (lane == 0) {
        if (kind == 0) s.cost[16 + tile][mode] = ((q[0] + 2) >> 2) + ((q[1] + 2) >> 2) + ((q[2] + 2) >> 2) + ((q[3] + 2) >> 2);
        else {
          const int b0 = (tile >> 1) * 8 + (tile & 1) * 2;           // first of the tile's four 8x8 blocks (raster of 4 x 4)
          s.cost[b0][mode] = (q[0] + 2) >> 2; s.cost[b0 + 1][mode] = (q[1] + 2) >> 2; s.cost[b0 + 4][mode] = (q[2] + 2) >> 2; s.cost[b0 + 5][mode] = (q[3] + 2) >> 2;
        }
      }
    }
  } else
  for (int j = wave; j < (PP ? (f.intra_p == 1 ? 35 : 5 * 35) : 20 * 35); j += NW) {
    int b = j / 35; const int mode = j - b * 35;
    if (PP && f.intra_p == 1) b = 16 + mytile;
    else if (PP) b = b < 4 ? (mytile >> 1) * 8 + (mytile & 1) * 2 + (b >> 1) * 4 + (b & 1) : 16 + mytile;      // the quarter's four 8x8 blocks and its 16x16 block
    uint32_t c;
    if (b < 16) c = analyse_item<3>(s, b, (b & 3) * 8, (b >> 2) * 8, mode, lane);
    else c = analyse_item<4>(s, b, ((b - 16) & 1) * 16, ((b - 16) >> 1) * 16, mode, lane);
    if (lane == 0) s.cost[b][mode] = c;
  }
  __syncthreads();
  if (tid < 20 && mine(tid)) {
    uint32_t bc = 0xffffffffu; int bm = 0;
    uint64_t excl = 0;                                       // "intra-chain": modes this block may not take (statement: intra_analyse_size(), oracle/hevc_enc.c)
    if (f.intra_chain) {
      const int l2b = tid < 16 ? 3 : 4, nb_ = 1 << l2b, bib = tid < 16 ? tid : tid - 16, nbb = 32 >> l2b;
      const int xb = X0 + (bib % nbb) * nb_, yb = Y0 + (bib / nbb) * nb_;
      if ((yb & 63) == 0 && ((xb + nb_) & 63) == 0 && avail64(f.cw, f.chp, xb, yb, xb + nb_, yb - 1)) excl |= intra_uses_above_right(l2b, 0);      // the CTU's above-right corner block
      if ((xb & 63) == 0 && avail64(f.cw, f.chp, xb, yb, xb - 1, yb + nb_)) excl |= intra_uses_below_left(l2b, 0);                                  // a block on the CTU's left edge
    }
    for (int m = 0; m < 35; m++) if (!((excl >> m) & 1) && s.cost[tid][m] < bc) { bc = s.cost[tid][m]; bm = m; }
    s.bestc[tid] = bc; s.bestm[tid] = bm;
    if (PP && cand[0] + cand[1] + cand[2] + cand[3] > 1) st_wt_u64(&f.ip_scratch[(size_t)ci * 20 + tid], (uint64_t)bc | ((uint64_t)bm << 32));      // for the block's other quarters
    const int l2 = tid < 16 ? 3 : 4, n = 1 << l2, bi = tid < 16 ? tid : tid - 16, nb = 32 >> l2;
    const int x0 = X0 + (bi % nb) * n, y0 = Y0 + (bi / nb) * n;
    const int bw = f.cw >> l2, ib = (y0 >> l2) * bw + (x0 >> l2);
    if (!PP) {
      if (l2 == 3) { f.im8[ib] = (uint8_t)bm; f.ic8[ib] = bc; }
      else { f.im16[ib] = (uint8_t)bm; f.ic16[ib] = bc; }
    }
  }
  if (PP && cand[0] + cand[1] + cand[2] + cand[3] > 1) {
    // the block's other quarters above the gate are other workgroups': the last one to arrive has everybody's prices (write-through stores, drained
    // before the count goes up; read past the caches) and decides -- nobody waits
    if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      const bool last = atomicAdd(&f.ip_arrive[ci], 1u) == (uint32_t)(cand[0] + cand[1] + cand[2] + cand[3]) - 1u;
      if (last) f.ip_arrive[ci] = 0;                          // (for the next picture)
      s.last = last ? 1 : 0;
    }
    __syncthreads();
    if (!s.last) continue;
    if (tid < 20 && !mine(tid) && cand[an_tile_of(tid)]) { const uint64_t v = ld_l2_u64(&f.ip_scratch[(size_t)ci * 20 + tid]); s.bestc[tid] = (uint32_t)v; s.bestm[tid] = (int)(v >> 32); }
  }
  __syncthreads();
  // ---- bottom-up split decision for this 32x32 block: thread per 8x8 cell
  if (tid < 16) {
    const uint32_t pen = ((uint32_t)f.lambda_q4 * SPLIT_BITS) >> 4;
    bool split16[4], chosen = false;
    for (int k = 0; k < 4; k++) {
      uint32_t c8 = pen;
      for (int j = 0; j < 4; j++) c8 += s.bestc[((k >> 1) * 2 + (j >> 1)) * 4 + (k & 1) * 2 + (j & 1)];
      split16[k] = !(PP && f.intra_p == 1) && c8 < s.bestc[16 + k];
      if (PP) {                                                  // intra-in-P: the quarter goes intra when that is cheaper than what the search found
        const uint32_t cintra = (split16[k] ? c8 : s.bestc[16 + k]) + (((uint32_t)f.lambda_q4 * INTRA_P_BITS) >> 4);
        cand[k] = cand[k] && cintra < icost[k];
        chosen |= cand[k];
      }
    }
    const int bx = tid & 3, by = tid >> 2, k = (by >> 1) * 2 + (bx >> 1);
    const int i = ((Y0 >> 3) + by) * (f.cw >> 3) + (X0 >> 3) + bx;
    if (PP && chosen && tid == 0) f.sync[(size_t)3 * (f.cw >> 6) * (f.ch >> 6) + 1] = 1u;
    if (PP && !chosen) { }                                      // the block stays as the search left it
    else if (PP && !cand[k]) f.cu_log2[i] = 4;                  // an inter quarter beside an intra one: a 16x16 unit with the vector it has
    else {
      if (PP) { f.cu_mv[i * 2] = 0; f.cu_mv[i * 2 + 1] = 0; }
      int l2, mode;
      if (!split16[k]) { l2 = 4; mode = s.bestm[16 + k]; }
      else { l2 = 3; mode = s.bestm[tid]; }
      f.cu_log2[i] = (uint8_t)l2; f.cu_intra_mode[i] = (uint8_t)mode; f.cu_intra[i] = 1; f.cu_flags[i] = 0;
    }
  }
  }
}

// One plane of one CU on one wave (kernel_common.h "One intra block per WAVE"): block of n = 1 << L2 component samples.
// Returns whether the block has non-zero levels.
template <int L2, bool ADJ, bool LL = false>
__device__ __forceinline__ bool intra_block_wave(IntraCtuLds &s, IntraWaveScratch &ws, const IntraBlk &d, int cidx, int S, const QuantConst &q,
                                                 int lane, int adj, uint32_t *ecol, unsigned long long *erow, uint32_t gen, const uint8_t *mtab = nullptr)
{
  constexpr int N = 1 << L2;
  const int P = 16 + 2 * S, g = lane >> 4, c = lane & 15, rx = d.rx, ry = d.ry;
  const bool active = c < N && 4 * g < N;
  int pred[4], res[4] = {0, 0, 0, 0};
  wave_intra_predict<L2>(s.pic, P, ws, d, cidx == 0, lane, g, c, pred);
  if (active) {
    const uint32_t s4 = *(const uint32_t *)&s.src[(ry + c) * S + rx + 4 * g];
#pragma unroll
    for (int r = 0; r < 4; r++) res[r] = (int)((s4 >> (8 * r)) & 255u) - pred[r];
  }
  PROF(5);
  bool cbf;
  if constexpr (LL) {
    // cu_transquant_bypass (EncFrame.lossless): the levels ARE the residual, sample by sample, and the reconstruction is the source
    bool nzl = false;
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; r++) { s.lev[(ry + c) * S + rx + 4 * g + r] = (int16_t)res[r]; nzl |= res[r] != 0; pred[r] += res[r]; }
    }
    cbf = __ballot(nzl) != 0;
  } else {
  // ---- forward rows, forward columns, quantiser; levels -> s.lev, dequantised coefficients stay in registers
  const XfLaneF16 &xl = s.xf[d.xf][lane];
  const kv_f16x4 ta = kv_h4(xl.ta), tb = kv_h4(xl.tb);
  int y[4], co[4], dq[4] = {0, 0, 0, 0};
  mfma16_data_a(res, ta, y);
#pragma unroll
  for (int r = 0; r < 4; r++) y[r] = (y[r] + (1 << (L2 - 2))) >> (L2 - 1);
  mfma16_data_b(ta, y, co);
  PROF(6);
  bool nz = false;
  if (ADJ) {
    // rdoq / signhide (bit 0 / bit 1): levels to s.lev, quantiser remainders to the wave's scratch, one lane per 4x4 coefficient group adjusts
    // the levels (hevc_core.h adjust_group), then every lane takes its levels back -- all inside the wave, no workgroup barrier
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cf = clip3(-32768, 32767, (co[r] + (1 << (L2 + 5))) >> (L2 + 6));
        const uint32_t a = (uint32_t)iabs(cf), prod = a * (uint32_t)(mtab ? (q.qscale << 4) / mtab[(4 * g + r) * N + c] : q.qscale);
        const int lvm = imin((int)((prod + (uint32_t)q.qoff) >> q.qshift), 32767);
        const int du = clip3(-256, 511, (int)(prod >> (q.qshift - 8)) - (lvm << 8));
        s.lev[(ry + 4 * g + r) * S + rx + c] = (int16_t)(cf < 0 ? -lvm : lvm);
        ws.tr[(4 * g + r) * N + c] = (int16_t)((du + 256) | (cf < 0 ? 0x8000 : 0));
      }
    }
    wave_sync();
    adjust_block_lds(&s.lev[ry * S + rx], ws.tr, N, S, intra_scan_idx(1, L2, cidx, d.mode), adj & 1, adj & 2, lane, 64);
    wave_sync();
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; r++) { const int lv = s.lev[(ry + 4 * g + r) * S + rx + c]; nz |= lv != 0; dq[r] = mtab ? dequant_coef_qm(lv, q, mtab[(4 * g + r) * N + c]) : dequant_coef_q(lv, q); }
    }
    wave_sync();                                              // (ws.tr is free again for the inverse transform)
  } else if (active) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int cf = clip3(-32768, 32767, (co[r] + (1 << (L2 + 5))) >> (L2 + 6));
      const int lv = mtab ? quant_level_qm(cf, q, mtab[(4 * g + r) * N + c], nullptr) : quant_level_q(cf, q);
      nz |= lv != 0;
      dq[r] = mtab ? dequant_coef_qm(lv, q, mtab[(4 * g + r) * N + c]) : dequant_coef_q(lv, q);
      s.lev[(ry + 4 * g + r) * S + rx + c] = (int16_t)lv;
    }
  }
  cbf = __ballot(nz) != 0;
  PROF(7);
  if (cbf) {
    wave_inverse16(ws, tb, dq, g, c, res);
#pragma unroll
    for (int r = 0; r < 4; r++) pred[r] = clip8(pred[r] + res[r]);
  }
  }
  PROF(8);
  if (active) {
    const uint32_t o = (uint32_t)pred[0] | ((uint32_t)pred[1] << 8) | ((uint32_t)pred[2] << 16) | ((uint32_t)pred[3] << 24);
    *(uint32_t *)&s.pic[(ry + c + 1) * P + 16 + rx + 4 * g] = o;
    // what a neighbouring CTU's workgroup will read -- the CTU's last row (IB_EDGE) and last column (IB_EDGE_R) -- leaves at once, as self-validating
    // words nobody waits for (kernel_common.h IntraNeighbours); the CTU itself goes to the picture in full lines at the end
    if ((d.flags & IB_EDGE) && c == N - 1) st_wt_u64(erow + ((rx + 4 * g) >> 2), (unsigned long long)o | ((unsigned long long)gen << 32));      // the block's last row, four samples and the generation per word
    if ((d.flags & IB_EDGE_R) && g == (N >> 2) - 1) st_wt_u32(ecol + ry + c, (o >> 24) | (gen << 8));      // the block's last column, one sample and the generation per word
  }
  wave_sync();
  PROF(9);
  return cbf;
}

// One workgroup of KVZ_INTRA_WAVES waves per (CTU, colour plane).  The CTU's coding units form a list in z-order; a wave takes the
// next one, waits until the 8x8 units its reference samples lie in are final (kernel_common.h IntraChain: independent quadrants
// run side by side), reconstructs it and marks its units.  CTUs are coupled by the samples themselves: a CTU's right column and bottom
// row leave as tagged words (f.edge_col / f.edge_row, kernel_common.h IntraNeighbours) the moment the block that holds them is done, and a
// block of the neighbouring CTU polls exactly the words its MODE reads -- a CTU starts when the two or three blocks of its left and upper
// neighbours its first block reads are done, not when a counted prefix of them is.  (Rounds 2-3 published progress counters per CTU; the
// decoder's k_dec_intra still does, driven by the transform-block list.)
#ifndef KVZ_INTRA_WAVES
#define KVZ_INTRA_WAVES 8          // (measured at 1080p: 2 waves 1.58 ms, 4: 1.41, 8: 1.36 -- a wave that has just finished a block spends a microsecond on its bookkeeping before it can take the next)
#endif
// ADJ: rdoq / signhide -- a kernel of its own, so that the plain chain (every step of it is on the critical path) stays as it was.
// PP: the intra units of a P picture (intra-in-P), launched behind k_inter_recon: a CTU without intra units (nearly all) leaves at
// once; in the others the inter units count as finished from the start, the CTU picture in LDS starts as the inter
// reconstruction left it, and only the intra units' levels and cbf bits are written.
// SCAL: `scaling-list default` -- a form of its own for the same reason as ADJ (the per-position factors cost the plain chain 30 registers when they are a run-time branch)
// LL: `lossless` -- see intra_block_wave
template <bool ADJ, bool PP, bool SCAL = false, bool LL = false>
__global__ __launch_bounds__(64 * KVZ_INTRA_WAVES) void k_intra_recon(EncFrame f)
{
  constexpr int W = KVZ_INTRA_WAVES, T = 64 * W;
  __shared__ IntraCtuLds s;
  __shared__ IntraWaveScratch wss[W];
  __shared__ IntraChain ch;
  __shared__ IntraBlk blk[64];                              // the CTU's coding units in z-order
  __shared__ uint2 dep[64], cover[64];                      // per coding unit: units it waits for, units it finishes
  __shared__ uint32_t nblk_s, ticket_s;
  __shared__ uint8_t cu_cbf_s[64];
  // Workgroups are dispatched in blockIdx order and a picture has more of them than fit on the chip at once, so they are numbered
  // the way the wavefront advances (f.intra_order: by cx + 2 cy -- every CTU a block depends on comes earlier) instead of in raster
  // order, where the right ends of the upper rows would hold the slots the lower left needs.
  // ... and a workgroup does not belong to one (CTU, plane): the launch has only as many workgroups as the wavefront keeps busy (a few
  // anti-diagonals of CTUs), each taking the next (CTU, plane) in that order from a ticket counter when it is done with the last.
  // Whatever a workgroup waits for has a lower ticket, so it is held by a workgroup that runs or is finished.  The point is what the
  // OTHER kernels on the GPU get: a workgroup per (CTU, plane) parks 1500 waiting workgroups with 30 KB of LDS each on the chip for the
  // 1.6 ms the chain takes, and the decoder's (or, beside k_dec_intra, the encoder's) P pictures stand still until it is over.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wc = f.cw >> 6;
  const uint32_t nticket = 3u * (uint32_t)wc * (uint32_t)band_rows(f);
  uint32_t *const ticket_ctr = f.sync + (size_t)3 * wc * (f.ch >> 6);
  if (PP && !ticket_ctr[1]) return;                         // no intra unit in this P picture (k_intra_analyse<true> says): nothing to do
  for (int i = tid; i < 4 * 64; i += T) ((uint4 *)s.xf)[i] = ((const uint4 *)g_xf16.t)[i];      // the transforms' matrix operands: once per workgroup, not per (CTU, plane)
  for (bool once = true;; once = false) {
  // (PP: a workgroup per (CTU, plane), in dispatch order -- nearly all of them find nothing to do, and a ticket would only add a round trip)
  if (PP) { if (!once) break; } else {
  __syncthreads();                                          // (everybody is done with the last (CTU, plane): LDS and the ticket word are free)
  if (tid == 0) ticket_s = atomicAdd(ticket_ctr, 1u);
  __syncthreads();
  }
  const uint32_t ticket = PP ? blockIdx.x : ticket_s;
  if (ticket >= nticket) break;
  const int c = (int)(ticket % 3u), ctu = (int)f.intra_order[ticket / 3u];
  const int row = ctu / wc, cx = ctu % wc;
#ifdef KVZ_PROF
  if (threadIdx.x == 0) { for (int k = 0; k < 15; k++) g_prof[k] = 0; g_prof[15] = clock64(); }
#endif
  const int adj = (f.rdoq ? 1 : 0) | (f.signhide ? 2 : 0);      // level adjustment behind the quantiser (hevc_core.h adjust_group)
  const int sh = c ? 1 : 0, S = 64 >> sh, pw = f.cw >> sh, P = 16 + 2 * S;
  uint64_t im = ~0ull;                                      // the CTU's 8x8 units (z-order) that belong to intra coding units
  if (PP) {
    int zx, zy; ctu_z_to_xy(lane, zx, zy);
    im = __ballot(f.cu_intra[b8idx(f, cx * 64 + zx * 8, row * 64 + zy * 8)] != 0);      // (every wave for itself: no barrier in front of the exit)
    if (!im) continue;                                      // (its samples are k_inter_recon's and final: the neighbours read them from the picture)
  }
  const int qpl = ctu_quant_qp(f, cx * 64, row * 64), qp = c ? kChromaQp[qpl] : qpl;
  const int hc = f.ch >> 6;
  IntraNeighbours nb;
  nb.nb_up = row > 0 && !tile_row_starts_at(hc, f.tile_rows, row); nb.nb_left = cx > 0 && !tile_col_starts_at(wc, f.tile_cols, cx);
  nb.nb_ur = nb.nb_up && cx + 1 < wc && !tile_col_starts_at(wc, f.tile_cols, cx + 1); nb.nb_ul = nb.nb_up && nb.nb_left;
  uint32_t *const ecol = f.edge_col[c] + (size_t)ctu * S;      // this CTU's right column for its right neighbour (IB_EDGE_R)
  nb.ecol_left = ecol - S; nb.gen = f.chain_gen;
  unsigned long long *const erow = f.edge_row[c] + (size_t)ctu * (S >> 2);     // this CTU's bottom row for the CTUs below (IB_EDGE)
  nb.erow_up = erow - (size_t)wc * (S >> 2); nb.erow_ur = nb.erow_up + (S >> 2); nb.erow_ul = nb.erow_up - (S >> 2);
  if (PP) {
    // which of the neighbours' edge units are intra units (kernel_common.h IntraBorders: the inter units around are final, nothing to wait for
    // there) -- lanes 0-7: the left CTU's right column, 8-15 / 16-23: the bottom rows of the upper / upper-right CTU, 24: the corner
    const int g = lane >> 3, u = lane & 7;
    int X = -1, Y = -1;
    if (g == 0 && nb.nb_left) { X = cx * 64 - 8; Y = row * 64 + u * 8; }
    else if (g == 1 && nb.nb_up) { X = cx * 64 + u * 8; Y = row * 64 - 8; }
    else if (g == 2 && nb.nb_ur) { X = (cx + 1) * 64 + u * 8; Y = row * 64 - 8; }
    else if (lane == 24 && nb.nb_ul) { X = cx * 64 - 8; Y = row * 64 - 8; }
    const uint64_t m = __ballot(X >= 0 && f.cu_intra[b8idx(f, X < 0 ? 0 : X, Y < 0 ? 0 : Y)] != 0);
    nb.il = (uint32_t)m & 0xffu; nb.iu = (uint32_t)(m >> 8) & 0xffu; nb.iur = (uint32_t)(m >> 16) & 0xffu; nb.iul = (uint32_t)(m >> 24) & 1u;
  }
  chain_init(ch, (uint32_t)~im, (uint32_t)(~im >> 32));
  // ---- everything the chain needs to know about the CTU's coding units, one lane per 8x8 unit (z-order), compacted into a list
  if (wave == 0) {
    int zx, zy; ctu_z_to_xy(lane, zx, zy);
    const int X = cx * 64 + zx * 8, Y = row * 64 + zy * 8, bi = b8idx(f, X, Y);
    const int l2 = f.cu_log2[bi], mode = f.cu_intra_mode[bi], n = 1 << (l2 - sh), nl = 1 << l2, su = nl >> 3;
    cu_cbf_s[lane] = 0;
    const bool start = (lane & (su * su - 1)) == 0 && ((im >> lane) & 1);
    const uint64_t starts = __ballot(start);
    const int k = __popcll(starts & ((1ull << lane) - 1ull));
    if (start) {
      const bool aL = avail64(f.cw, f.chp, X, Y, X - 1, Y), aT = avail64(f.cw, f.chp, X, Y, X, Y - 1);
      const bool aBL = aL && avail64(f.cw, f.chp, X, Y, X - 1, Y + nl), aTR = aT && avail64(f.cw, f.chp, X, Y, X + nl, Y - 1);
      IntraBlk d;
      d.rx = (uint8_t)((zx * 8) >> sh); d.ry = (uint8_t)((zy * 8) >> sh);
      // availability is decided per group of n samples (below-left, left, corner, above, above-right): each group lies in one block
      // of this block's size, which either precedes this block in z-order or does not; the available groups are contiguous
      d.lo = (uint8_t)(aBL ? 0 : (aL ? n : 2 * n + 1)); d.hi = (uint8_t)(aTR ? 4 * n : (aT ? 3 * n : (aL ? 2 * n - 1 : 0)));      // (nothing available: lo = 2n + 1 > hi = 0)
      d.mode = (uint8_t)mode; d.l2 = (uint8_t)(l2 - sh);
      d.flags = (uint8_t)((intra_filter_needed(n, c ? 1 : 0, mode) ? IB_FILT : 0) | ((zx == 0 || zy == 0) ? IB_BORDER : 0) | ((zy + su == 8) ? IB_EDGE : 0) | ((zx + su == 8) ? IB_EDGE_R : 0));
      d.xf = (uint8_t)((l2 - sh == 2 && c == 0) ? XF16_DST4 : l2 - sh - 1);
      d.angle = (int16_t)kIntraAngle[mode]; d.inv = (int16_t)kInvAngle[mode];
      d.zu = (uint16_t)lane; d.next = (uint16_t)(lane + su * su);
      blk[k] = d;
      dep[k] = chain_dependencies(zx, zy, su, ((intra_uses_below_left(l2 - sh, c ? 1 : 0) >> mode) & 1) != 0, ((intra_uses_above_right(l2 - sh, c ? 1 : 0) >> mode) & 1) != 0);
      cover[k] = chain_cover(lane, su);
    }
    if (lane == 0) nblk_s = (uint32_t)__popcll(starts);
  }
  const uint8_t *gsrc = f.src[c] + (size_t)(row * S) * pw + cx * S;
  uint8_t *plane = f.rec[c];
  uint8_t *grec = plane + (size_t)(row * S) * pw + cx * S;
  int16_t *gcoef = f.coef[c] + (size_t)(row * S) * pw + cx * S;
  for (int k = tid; k < S * S / 16; k += T) {
    int y = k / (S / 16), xq = k % (S / 16);
    *(uint4 *)&s.src[y * S + xq * 16] = *(const uint4 *)&gsrc[(size_t)y * pw + xq * 16];
    if (PP) *(uint4 *)&s.pic[(y + 1) * P + 16 + xq * 16] = *(const uint4 *)&grec[(size_t)y * pw + xq * 16];      // the CTU as k_inter_recon left it
  }
  if (f.trace && tid == 0) { unsigned long long *t = f.trace + ((size_t)ctu * 3 + c) * 8; t[0] = wall_clock64(); t[3] = 0; t[4] = 0; }
  __syncthreads();
  const int nblk = wave_uniform_int((int)nblk_s);         // (bounds the claim loop: uniform by construction, kernel_common.h chain_claim)
  const QuantConst q8 = quant_const(qp, 3 - sh, 1), q16 = quant_const(qp, 4 - sh, 1);      // (coding units are 8x8 or 16x16)
  IntraWaveScratch &ws = wss[wave];
  bool first = true;
  for (;;) {
    const int k = chain_claim(ch, lane);
    if (k >= nblk) break;
    PROF(1);                                                // claim
    const IntraBlk d = wave_uniform(&blk[k]);              // (wave-uniform: what is derived from it runs on the scalar unit)
    const uint2 dp = dep[k], cv = cover[k];
    // The neighbouring CTUs' samples this block reads come FIRST: they do not depend on this CTU's own blocks, so the memory round trip that fetches them
    // (and the poll, when the neighbour is not there yet) runs while the blocks in front of this one are still being coded, instead of behind them on the
    // chain's critical path -- three of the four blocks of a CTU's top row paid it there.  (A form that looks once and polls only when the block's turn has
    // come -- tried because a two-process run beside the test suite crawled, which turned out to have another cause -- is 6 % slower: 0.71 against 0.67 ms.)
    if (d.flags & IB_BORDER) {
      // the neighbouring CTUs are waited for only as far as the block's MODE reads them (hevc_core.h intra_uses_*): with "intra-chain" the left-edge blocks never
      // read the left CTU's below-left samples and the above-right CTU is not read at all
      const int n = 1 << d.l2;
      const int nl2 = ((intra_uses_below_left(d.l2, c) >> d.mode) & 1) ? 2 * n : n, nt2 = ((intra_uses_above_right(d.l2, c) >> d.mode) & 1) ? 2 * n : n;
      borders_need_wave(ch, nb, s.pic, P, plane, pw, cx, row, S, sh, 2 * S, d.rx, d.ry, n, f.err, lane, nl2, nt2);
    }
    PROF(2);                                                // neighbouring CTUs: waits and copies
    chain_wait_done(ch, make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)dp.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)dp.y)), f.err, lane);
    PROF(3);                                                // waiting for the units of this CTU the block reads
    if (f.trace && first && k == 0 && lane == 0) f.trace[((size_t)ctu * 3 + c) * 8 + 1] = wall_clock64();
    first = false;
    bool cbf;
    switch (d.l2) {
      case 2: cbf = intra_block_wave<2, ADJ, LL>(s, ws, d, c, S, q8, lane, adj, ecol, erow, f.chain_gen, SCAL ? f.scaling + scaling_offset(2, c, 0) : nullptr); break;
      case 3: cbf = intra_block_wave<3, ADJ, LL>(s, ws, d, c, S, c ? q16 : q8, lane, adj, ecol, erow, f.chain_gen, SCAL ? f.scaling + scaling_offset(3, c, 0) : nullptr); break;
      default: cbf = intra_block_wave<4, ADJ, LL>(s, ws, d, c, S, q16, lane, adj, ecol, erow, f.chain_gen, SCAL ? f.scaling + scaling_offset(4, c, 0) : nullptr); break;
    }
    const uint2 cvu = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)cv.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)cv.y));
    chain_mark_done(ch, cvu, lane);
#ifndef KVZ_PROF
    if (f.trace && c == 0 && lane == 0 && k < 16) f.trace[(size_t)wc * (f.ch >> 6) * 56 + (size_t)ctu * 16 + k] = wall_clock64() | ((unsigned long long)d.l2 << 60);      // (tools/intra_timeline.py: when the luma blocks of the CTU were done)
#endif
    PROF(10);                                               // mark
#ifdef KVZ_PROF
    if (threadIdx.x == 0) g_prof[14] += 1;                   // blocks wave 0 did
#endif
    if (cbf && (int)d.zu + lane < (int)d.next) cu_cbf_s[d.zu + lane] = 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the CTU's samples -> the picture, in whole lines (the chain itself stored only what neighbouring workgroups read; PP: the inter units' samples are what they were)
  for (int k = tid; k < S * S / 16; k += T) { const int y = k / (S / 16), xq = k % (S / 16); *(uint4 *)&grec[(size_t)y * pw + xq * 16] = *(const uint4 *)&s.pic[(y + 1) * P + 16 + xq * 16]; }
  for (int k = tid; k < S * S / 8; k += T) {
    int y = k / (S / 8), xq = k % (S / 8);
    if (PP && !((im >> kv_zunit8(((xq * 8) << sh) >> 3, (y << sh) >> 3)) & 1)) continue;       // (an inter unit's levels are k_inter_recon's)
    *(uint4 *)&gcoef[(size_t)y * pw + xq * 8] = *(const uint4 *)&s.lev[y * S + xq * 8];
  }
  if (tid < 64 && cu_cbf_s[tid]) {
    int zx, zy; ctu_z_to_xy(tid, zx, zy);
    const int bi = b8idx(f, cx * 64 + zx * 8, row * 64 + zy * 8);
    atomicOr((uint32_t *)(f.cu_cbf + (bi & ~3)), (1u << c) << (8 * (bi & 3)));   // the three planes own one bit each of the byte
  }
  if (f.trace && tid == 0) { unsigned long long *t = f.trace + ((size_t)ctu * 3 + c) * 8; t[2] = wall_clock64(); t[7] = (unsigned long long)nblk; }
#ifdef KVZ_PROF
  if (f.trace && tid == 0 && c == 0) for (int k = 0; k < 16; k++) f.trace[(size_t)wc * (f.ch >> 6) * 24 + (size_t)ctu * 16 + k] = (unsigned long long)g_prof[k];
#endif
  }
}

// =============================================================================================
// Per-CTU QP (delta-QP map): after reconstruction, which CU of every CTU is the first with residual, and from that the
// QpY every CU ends up with (H.265 8.6.1, quantisation group = CTU: the prediction is the QpY of the previous CTU in
// decoding order, or the slice QP at the start of a tile and -- with WPP -- of a CTU row).  Statement: roi_resolve() in
// oracle/hevc_enc.c.
// =============================================================================================
__global__ __launch_bounds__(64) void k_qp_first(EncFrame f)                // one wave per CTU of the band
{
  const int wc = f.cw >> 6, ctu = blockIdx.x + f.row0 * wc, cx = ctu % wc, cy = ctu / wc, lane = threadIdx.x;
  int xi, yi; ctu_z_to_xy(lane, xi, yi);
  const int x = cx * 64 + xi * 8, y = cy * 64 + yi * 8, bi = b8idx(f, x, y), n = 1 << f.cu_log2[bi];
  const bool coded_origin = !((x | y) & (n - 1)) && f.cu_cbf[bi] != 0;
  const uint64_t m = __ballot(coded_origin);
  if (lane == 0) f.ctu_first[ctu] = (uint8_t)(m ? __builtin_ctzll(m) : 64);
}
__global__ __launch_bounds__(256) void k_qp_chain(EncFrame f)                // one thread per CTU of the band
{
  const int wc = f.cw >> 6, hc = f.ch >> 6, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= wc * band_rows(f)) return;
  const int ctu = t + f.row0 * wc, cx = ctu % wc, cy = ctu / wc;
  // QpY of the nearest CTU at or before this one in its chain that codes a delta; the chain runs in decoding order (tile scan) and
  // starts at the CTU row of the tile (WPP) or at the tile
  const int tc = tile_col_of(wc, f.tile_cols, cx), cx0 = tile_col_first(wc, f.tile_cols, tc), cx1 = tile_col_first(wc, f.tile_cols, tc + 1);
  const int first_cy = f.wpp ? cy : tile_row_first(hc, f.tile_rows, tile_row_of(hc, f.tile_rows, cy));
  int qy = f.qp, prev = f.qp; bool have = false, have_prev = false;
  for (int y = cy, x = cx; y >= first_cy && !have_prev; y--, x = cx1 - 1)
    for (; x >= cx0; x--) {
      const int c = y * wc + x;
      if (f.ctu_first[c] < 64) {
        if (c == ctu) { qy = f.ctu_qt[c]; have = true; }
        else { prev = f.ctu_qt[c]; have_prev = true; break; }
      }
    }
  (void)have_prev;
  if (!have) qy = prev;
  f.ctu_qy[ctu] = (int8_t)qy;
  f.ctu_delta[ctu] = (int8_t)(have ? qy - prev : 0);
}

// The two kernels above as ONE launch where the chain of a CTU never leaves its CTU row -- WPP (uvgComm's default, kvazaarfilter.cpp:194): the QpY prediction
// restarts with the slice QP at every CTU row of a tile.  A workgroup per CTU row: its waves find the rows' first coded units (one CTU per wave at a time), then
// a thread per CTU walks back along the row inside its tile.  Same results as k_qp_first + k_qp_chain (the launch is one of ten on a P picture's chain in
// uvgComm's default mode, each worth ~6 us whatever it does).
__global__ __launch_bounds__(256) void k_qp_rows(EncFrame f)
{
  __shared__ uint8_t first_s[256];                           // (a CTU row: at most 16384 / 64 CTUs)
  const int wc = f.cw >> 6, cy = blockIdx.x + f.row0, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int xi, yi; ctu_z_to_xy(lane, xi, yi);
  for (int cx = wave; cx < wc; cx += 4) {
    const int x = cx * 64 + xi * 8, y = cy * 64 + yi * 8, bi = b8idx(f, x, y), n = 1 << f.cu_log2[bi];
    const bool coded_origin = !((x | y) & (n - 1)) && f.cu_cbf[bi] != 0;
    const uint64_t m = __ballot(coded_origin);
    if (lane == 0) { const uint8_t v = (uint8_t)(m ? __builtin_ctzll(m) : 64); first_s[cx] = v; f.ctu_first[cy * wc + cx] = v; }
  }
  __syncthreads();
  for (int cx = threadIdx.x; cx < wc; cx += 256) {
    const int ctu = cy * wc + cx;
    const int tc = tile_col_of(wc, f.tile_cols, cx), cx0 = tile_col_first(wc, f.tile_cols, tc);
    int qy = f.qp, prev = f.qp; bool have = false;
    for (int x = cx; x >= cx0; x--) {
      if (first_s[x] < 64) {
        if (x == cx) { qy = f.ctu_qt[ctu]; have = true; }
        else { prev = f.ctu_qt[cy * wc + x]; break; }
      }
    }
    if (!have) qy = prev;
    f.ctu_qy[ctu] = (int8_t)qy;
    f.ctu_delta[ctu] = (int8_t)(have ? qy - prev : 0);
  }
}

// =============================================================================================
// Deblocking: all vertical edges of the picture, then (second launch) all horizontal edges
// =============================================================================================
// Both deblocking passes in one launch (whole pictures; the band mode of the tile-row split keeps the two kernels below: it
// exchanges halo rows between them).  A workgroup owns the 64x64 tile shifted by (-4, -4) against the CTU grid: every sample
// a vertical edge x = 64 tx + 8 e or a horizontal edge y = 64 ty + 8 e of this CTU can modify (three each side) or read (four)
// lies inside the tile, and the tiles partition the picture -- so the tile goes to LDS once, is filtered vertically, then
// horizontally on the vertically filtered samples (8.7.2: the whole picture's vertical edges come first), and goes back.
__global__ __launch_bounds__(256) void k_deblock_tile(EncFrame f)
{
  constexpr int P = 80, PC = 48;                           // LDS pitches: 68 (36) used, multiples of 16 so that the 16-byte pieces stay aligned
  __shared__ __attribute__((aligned(16))) uint8_t ty_[68 * P];
  __shared__ __attribute__((aligned(16))) uint8_t tc_[2][34 * PC];
  // the CU records of the tile's 8x8 cells and one ring around them (cells -1 .. 8 in both directions): everything the boundary
  // strength needs, fetched in one go beside the samples instead of per edge segment
  __shared__ uint8_t r_log2[100], r_intra[100], r_cbf[100]; __shared__ uint32_t r_mv[100];
  const int tid = threadIdx.x, wc = f.cw >> 6, hc = f.ch >> 6;
  if (f.me_cand && blockIdx.x == 0 && tid == 0) { f.me_cand[0] = 0; f.sync[(size_t)3 * wc * hc + 1] = 0; }      // intra-in-P: the next picture's k_me starts its list of candidate blocks from nothing, its "has intra units" word from zero
  const int lin = xcd_contiguous((int)blockIdx.x, (int)gridDim.x), tx = lin % wc, tyi = lin / wc;
  const int X0 = tx * 64 - 4, Y0 = tyi * 64 - 4, CX0 = X0 >> 1, CY0 = Y0 >> 1, cw2 = f.cw >> 1;
  const int TW = tx == wc - 1 ? 68 : 64, TH = tyi == hc - 1 ? 68 : 64;
  if (tid < 100) {
    const int bx = tx * 8 - 1 + tid % 10, by = tyi * 8 - 1 + tid / 10;
    uint32_t l2 = 6, in = 0, cb = 0, mv = 0;
    if (bx >= 0 && by >= 0 && bx < f.b8w && by < f.b8h) {
      const int i = by * f.b8w + bx;
      l2 = f.cu_log2[i]; in = f.cu_intra[i]; cb = f.cu_cbf[i]; mv = *(const uint32_t *)&f.cu_mv[i * 2];
    }
    r_log2[tid] = (uint8_t)l2; r_intra[tid] = (uint8_t)in; r_cbf[tid] = (uint8_t)cb; r_mv[tid] = mv;
  }
  __syncthreads();
  auto cell = [&](int x, int y) { return ((y >> 3) - (tyi * 8 - 1)) * 10 + ((x >> 3) - (tx * 8 - 1)); };
  auto bs_of = [&](int cp, int cq) -> int {
    if (r_intra[cp] | r_intra[cq]) return 2;
    if ((r_cbf[cp] | r_cbf[cq]) & 1) return 1;
    const uint32_t a = r_mv[cp], b = r_mv[cq];
    if (iabs((int)(int16_t)(a & 0xffff) - (int)(int16_t)(b & 0xffff)) >= 4 || iabs((int)(int16_t)(a >> 16) - (int)(int16_t)(b >> 16)) >= 4) return 1;
    return 0;
  };
  // ---- boundary strengths of this thread's vertical and horizontal edge segment, from the records alone.  A tile none of whose
  // segments is filtered (still background: large units, equal vectors, no coefficients) leaves without touching a sample.
  int bsv = 0, bsh = 0, qpv = 0, qph = 0;
  if (tid < 8 * (TH / 4)) {
    const int x = tx * 64 + (tid & 7) * 8, y = Y0 + (tid >> 3) * 4;
    if (x > 0 && y >= 0) {
      const int cq = cell(x, y), cp = cq - 1;
      if ((x & ((1 << r_log2[cq]) - 1)) == 0) {
        bsv = bs_of(cp, cq);
        if (bsv) qpv = f.ctu_qy ? (cu_qpy(f, x - 1, y) + cu_qpy(f, x, y) + 1) >> 1 : f.qp;      // QpP and QpQ averaged (8.7.2.5.3)
      }
    }
  }
  if (tid < 8 * (TW / 4)) {
    const int y = tyi * 64 + (tid & 7) * 8, x = X0 + (tid >> 3) * 4;
    if (y > 0 && x >= 0) {
      const int cq = cell(x, y), cp = cq - 10;
      if ((y & ((1 << r_log2[cq]) - 1)) == 0) {
        bsh = bs_of(cp, cq);
        if (bsh) qph = f.ctu_qy ? (cu_qpy(f, x, y - 1) + cu_qpy(f, x, y) + 1) >> 1 : f.qp;
      }
    }
  }
  if (!__syncthreads_or(bsv | bsh)) return;
  // ---- tile in: 16-byte pieces (X0 is 4-byte aligned: dwordx4 with dword alignment), the last piece of a 68-wide row 4 bytes
  for (int i = tid; i < TH * 5; i += 256) {
    const int y = i / 5, k = i - y * 5, x = k * 16, gx = X0 + x, gy = Y0 + y;
    if (gy < 0 || x >= TW) continue;
    const uint8_t *g = &f.rec[0][(size_t)gy * f.cw + gx];
    if (k < 4) { if (gx >= 0) *(kv_u32x4 *)&ty_[y * P + x] = *(const kv_u32x4 *)g; else { kv_u32x4 v; v.x = 0; v.y = *(const uint32_t *)(g + 4); v.z = *(const uint32_t *)(g + 8); v.w = *(const uint32_t *)(g + 12); *(kv_u32x4 *)&ty_[y * P + x] = v; } }
    else *(uint32_t *)&ty_[y * P + x] = *(const uint32_t *)g;
  }
  // chroma: aligned dwords from two samples left of the tile (CX0 - 2 is a multiple of 4): sample x of the tile sits at LDS column x + 2
  for (int i = tid; i < 2 * (TH / 2) * 9; i += 256) {
    const int pl = i / ((TH / 2) * 9), r = i - pl * ((TH / 2) * 9), y = r / 9, k = r - y * 9, gx = CX0 - 2 + 4 * k, gy = CY0 + y;
    if (gy >= 0 && gx >= 0) *(uint32_t *)&tc_[pl][y * PC + 4 * k] = *(const uint32_t *)&f.rec[1 + pl][(size_t)gy * cw2 + gx];
  }
  __syncthreads();
  // ---- vertical edges: 8 edges x 16 (17) four-row segments
  if (bsv) {
    const int x = tx * 64 + (tid & 7) * 8, y = Y0 + (tid >> 3) * 4;
    deblock_luma_segment(&ty_[(y - Y0) * P + (x - X0)], 1, P, bsv, qpv);
    if (bsv == 2 && (x & 15) == 0) {
      const int o = ((y >> 1) - CY0) * PC + ((x >> 1) - CX0) + 2;
      deblock_chroma_segment(&tc_[0][o], 1, PC, 2, qpv);
      deblock_chroma_segment(&tc_[1][o], 1, PC, 2, qpv);
    }
  }
  __syncthreads();
  // ---- horizontal edges: 8 edges x 16 (17) four-column segments, on the vertically filtered samples
  if (bsh) {
    const int y = tyi * 64 + (tid & 7) * 8, x = X0 + (tid >> 3) * 4;
    deblock_luma_segment(&ty_[(y - Y0) * P + (x - X0)], P, 1, bsh, qph);
    if (bsh == 2 && (y & 15) == 0) {
      const int o = ((y >> 1) - CY0) * PC + ((x >> 1) - CX0) + 2;
      deblock_chroma_segment(&tc_[0][o], PC, 1, 2, qph);
      deblock_chroma_segment(&tc_[1][o], PC, 1, 2, qph);
    }
  }
  __syncthreads();
  // ---- tile out
  for (int i = tid; i < TH * 5; i += 256) {
    const int y = i / 5, k = i - y * 5, x = k * 16, gx = X0 + x, gy = Y0 + y;
    if (gy < 0 || x >= TW) continue;
    uint8_t *g = &f.rec[0][(size_t)gy * f.cw + gx];
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
    if (k == 0) *(uint16_t *)(g + 2) = (uint16_t)(v >> 16);                 // the first two bytes belong to the tile on the left
    else if (k == 8 && TW == 64) *(uint16_t *)g = (uint16_t)v;              // ... the last two to the tile on the right
    else *(uint32_t *)g = v;
  }
}

__global__ __launch_bounds__(256) void k_deblock_v(EncFrame f)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int ne = (f.cw >> 3) - 1, ns = band_rows(f) * 16;    // edges per row of segments, segment rows of the band
  if (t >= ne * ns) return;
  int x = ((t % ne) + 1) * 8, y = (t / ne) * 4 + f.row0 * 64;
  if (!is_cu_edge_v(f, x, y)) return;
  int bs = edge_bs(f, x - 1, y, x, y);
  if (!bs) return;
  const int qp = (cu_qpy(f, x - 1, y) + cu_qpy(f, x, y) + 1) >> 1;            // QpP and QpQ averaged (8.7.2.5.3)
  deblock_luma_segment(f.rec[0] + y * f.cw + x, 1, f.cw, bs, qp);
  if (bs == 2 && (x & 15) == 0) {
    int cw2 = f.cw >> 1;
    deblock_chroma_segment(f.rec[1] + (y >> 1) * cw2 + (x >> 1), 1, cw2, 2, qp);
    deblock_chroma_segment(f.rec[2] + (y >> 1) * cw2 + (x >> 1), 1, cw2, 2, qp);
  }
}
// part: 0 = every horizontal edge of the band, 1 = the edges inside it (they need nothing from the neighbouring bands), 2 = its two
// boundary edges (after the halo rows have arrived).  Horizontal edges lie 8 rows apart and change at most 3 rows on either side:
// the order in which they are filtered does not matter.
__global__ __launch_bounds__(256) void k_deblock_h(EncFrame f, int part)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  // horizontal edges of the band: every 8 rows from its first row (when that is not the picture's) to its end (likewise);
  // the two boundary edges are filtered by both neighbours, each keeping its own side (halo rows hold the other's samples)
  const int ylo = imax(8, f.row0 * 64), yhi = imin(f.ch - 8, (f.row0 + band_rows(f)) * 64);
  int ns = f.cw >> 2, ne = (yhi - ylo) / 8 + 1;
  if (t >= ne * ns) return;
  int x = (t % ns) * 4, y = ylo + (t / ns) * 8;
  if (part) {
    const bool boundary = y == f.row0 * 64 || y == (f.row0 + band_rows(f)) * 64;
    if (boundary != (part == 2)) return;
  }
  if (!is_cu_edge_h(f, x, y)) return;
  int bs = edge_bs(f, x, y - 1, x, y);
  if (!bs) return;
  const int qp = (cu_qpy(f, x, y - 1) + cu_qpy(f, x, y) + 1) >> 1;
  deblock_luma_segment(f.rec[0] + y * f.cw + x, f.cw, 1, bs, qp);
  if (bs == 2 && (y & 15) == 0) {
    int cw2 = f.cw >> 1;
    deblock_chroma_segment(f.rec[1] + (y >> 1) * cw2 + (x >> 1), cw2, 1, 2, qp);
    deblock_chroma_segment(f.rec[2] + (y >> 1) * cw2 + (x >> 1), cw2, 1, 2, qp);
  }
}

// =============================================================================================
// Entropy coding, parallel half: one wave per 16x16 block turns its coding unit(s) into the list
// of CABAC bins in coding order (16-bit tokens, hevc_core.h TokOut).  The wave stages the per-CU
// records of the block and its left / above neighbours in LDS, lane 0 emits the (short) CU headers,
// and every transform block is tokenised by all lanes at once: one lane per 4x4 sub-block, whose
// only cross-sub-block dependency (the greater1 context set) is resolved from a ballot of
// "has a level > 1".  The serial arithmetic coder runs on host threads (entropy_host.h).
// =============================================================================================
struct TileView {
  const CuRec *tile;                 // [3][3] records: b8 (bx0 + tx, by0 + ty): the unit and its left / above neighbours
  int bx0, by0;
  __device__ CuRec at(int x, int y) const { return tile[((y >> 3) - by0) * 3 + ((x >> 3) - bx0)]; }
};

// The tokenizer's tables as constants of the code object (round 5; until then every wave computed them -- a dozen dependent loads from the constant tables,
// 2 us of a 17 us wave): CoreTabs as the host code has it, and the sig_coeff_flag context patterns by SCAN POSITION -- sigk[scan][pattern][k] =
// sigpat[pattern][pos4[scan][k]], pattern 4 = the 4x4 block's map -- so that a sub-block's sixteen patterns are one 16-byte LDS read.
struct alignas(16) TokTabs { uint8_t sigk[3][5][16]; };
constexpr CoreTabs make_core_tabs() { CoreTabs t{}; for (int i = 0; i < 64; i++) core_tabs_fill_entry(t, i); return t; }
constexpr TokTabs make_tok_tabs()
{
  const CoreTabs c = make_core_tabs();
  TokTabs t{};
  for (int sc = 0; sc < 3; sc++) for (int k = 0; k < 16; k++) {
    for (int pc = 0; pc < 4; pc++) t.sigk[sc][pc][k] = c.sigpat[pc][c.pos4[sc][k]];
    t.sigk[sc][4][k] = c.ctxmap4x4[c.pos4[sc][k]];
  }
  return t;
}
static __device__ const CoreTabs g_core_tabs = make_core_tabs();
static __device__ const TokTabs g_tok_tabs = make_tok_tabs();
static_assert(sizeof(CoreTabs) % 4 == 0 && sizeof(TokTabs) % 16 == 0, "copied by dwords / 16-byte words");

// A lane's 4x4 sub-block in registers: its sixteen levels in scan order as int16 pairs and the significance mask (bit k = scan position k).  Everything the
// tokenizer does with a sub-block -- two passes over its coefficients -- reads these eight registers with compile-time indices (the loops below are unrolled
// over the sixteen positions) instead of a dependent LDS read per coefficient and pass (round 4: 10 of a wave's 17.7 us, profiles/r04_tok_phases.txt).
struct SbRegs { uint32_t w[8]; uint32_t m; };
__device__ __forceinline__ int sb_level(const SbRegs &r, int k) { return (int)(int16_t)(uint16_t)(r.w[k >> 1] >> ((k & 1) * 16)); }      // k: a compile-time constant where it matters

template <bool KEEP_LDS>
__device__ __forceinline__ void digest_build_wave(const CoreTabs *t, TuDigest &d, const int16_t *lv, int stride, int log2, int scan_idx, int lane, SbRegs &sb)
{
  const int sbl = log2 - 2, nsb2 = 1 << (2 * sbl);
  d.csbf[lane] = 0;
  wave_sync();
  bool nz = false;
  sb.m = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) sb.w[k] = 0;
  if (lane < nsb2) {
    int xs, ys; scan_pos(t, scan_idx, sbl, lane, xs, ys);
    const int16_t *p = lv + (ys << 2) * stride + (xs << 2);
    uint2 r0 = *reinterpret_cast<const uint2 *>(p), r1 = *reinterpret_cast<const uint2 *>(p + stride);
    uint2 r2 = *reinterpret_cast<const uint2 *>(p + 2 * stride), r3 = *reinterpret_cast<const uint2 *>(p + 3 * stride);
    // the lane's 32 bytes of the digest's scratch: the sub-block in raster order, read back through the scan table in scan order
    uint2 *dst = reinterpret_cast<uint2 *>(&d.scan[lane * 16]);
    dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
    const uint8_t *pos = t->pos4[scan_idx];
    int v[16];
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = d.scan[lane * 16 + pos[k]]; if (v[k]) m |= 1u << k; }
#pragma unroll
    for (int k = 0; k < 8; k++) sb.w[k] = ((uint32_t)v[2 * k] & 0xffffu) | ((uint32_t)v[2 * k + 1] << 16);
    if (KEEP_LDS) {                                        // (all reads of the raster form are done: the scan-order form takes its place for enc_subblock())
      uint32_t *d32 = reinterpret_cast<uint32_t *>(&d.scan[lane * 16]);
#pragma unroll
      for (int k = 0; k < 8; k++) d32[k] = sb.w[k];
    }
    sb.m = m;
    d.mask[lane] = (uint16_t)m;
    nz = m != 0;
    if (nz) d.csbf[ys * 8 + xs] = 1;
  }
  uint64_t sb64 = __ballot(nz);
  if (lane == 0) d.sbmask = sb64;
  wave_sync();
}

// subblock_g1_any() from registers: one of the sub-block's first eight coefficients in coding order has |level| > 1
__device__ __forceinline__ bool sb_g1_any(const SbRegs &r)
{
  bool any = false; int seen = 0;
#pragma unroll
  for (int k = 15; k >= 0; k--) if ((r.m >> k) & 1) { const int v = sb_level(r, k); if (seen < 8 && (v > 1 || v < -1)) any = true; seen++; }
  return any;
}

// enc_subblock() (hevc_core.h: the statement both forms follow, and the one the host tests drive) for a sub-block held in registers.  sk: the sixteen context
// patterns of the sub-block's scan positions (TokTabs::sigk row, four dwords).
template <class S>
__device__ __forceinline__ void enc_subblock_regs(S &c, const TuDigest &d, const SbRegs &r, const uint32_t (&sk)[4], int i, int last_sb, int last_pos, bool prev_g1, int log2, int cidx, int scan_idx, int sign_hiding)
{
  const int sbl = log2 - 2, nsb = 1 << sbl;
  int xs, ys; scan_pos(c.tabs, scan_idx, sbl, i, xs, ys);
  const int right = (xs < nsb - 1) ? d.csbf[ys * 8 + xs + 1] : 0;
  const int below = (ys < nsb - 1) ? d.csbf[(ys + 1) * 8 + xs] : 0;
  const uint32_t m = r.m;
  int coded = m != 0, infer_dc = 0;
  if (i < last_sb && i > 0) {
    cabac_bin(c, CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0), coded);
    infer_dc = 1;
  } else coded = 1;                        // inferred 1 for the last and the DC sub-block
  if (!coded) return;
  const int sigbase = CTX_SIG + (cidx ? 27 : 0);
  const int off = log2 == 2 ? 0 : (cidx == 0 ? ((i > 0 ? 3 : 0) + ((log2 == 3) ? ((scan_idx == 0) ? 9 : 15) : 21)) : ((log2 == 3) ? 9 : 12));
  const int start = (i == last_sb) ? last_pos - 1 : 15;
  if constexpr (sink_counts_only<S>::value) {
    if (start >= 0) c.n += start + 1 - ((infer_dc && (m >> 1) == 0) ? 1 : 0);
  } else {
#pragma unroll
    for (int k = 15; k >= 1; k--) if (k <= start) cabac_bin(c, sigbase + (int)((sk[k >> 2] >> ((k & 3) * 8)) & 0xffu) + off, (int)((m >> k) & 1));
    if (start >= 0 && !(infer_dc && (m >> 1) == 0))                 // (position 0: not sent when every other flag of a coded sub-block is zero; the block's DC coefficient has its own context)
      cabac_bin(c, sigbase + ((i == 0 && log2 != 2) ? 0 : (int)(sk[0] & 0xffu) + off), (int)(m & 1));
  }
  if (!m) return;
  int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
  if (prev_g1) ctx_set++;
  int c1 = 1, nsig = 0, g1idx = -1, g2 = 0;
  uint32_t signs = 0;
#pragma unroll
  for (int k = 15; k >= 0; k--) if ((m >> k) & 1) {
    const int v = sb_level(r, k), a = v < 0 ? -v : v;
    signs = (signs << 1) | (v < 0 ? 1u : 0u);
    if (nsig < 8) {
      const int g1 = a > 1;
      cabac_bin(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1, g1);
      if (g1) { c1 = 0; if (g1idx < 0) { g1idx = nsig; g2 = a > 2; } }
      else if (c1 > 0 && c1 < 3) c1++;
    }
    nsig++;
  }
  if (g1idx >= 0) cabac_bin(c, CTX_GT2 + (cidx ? 4 : 0) + ctx_set, g2);
  if (sign_hiding && (31 - __builtin_clz(m)) - __builtin_ctz(m) > 3) cabac_bypass_bits(c, signs >> 1, nsig - 1);
  else cabac_bypass_bits(c, signs, nsig);
  int rice = 0, j = 0;
#pragma unroll
  for (int k = 15; k >= 0; k--) if ((m >> k) & 1) {
    const int v = sb_level(r, k), a = v < 0 ? -v : v;
    const int base = (j < 8) ? ((j == g1idx) ? 3 : 2) : 1;
    if (a >= base) {
      enc_abs_remaining(c, a - base, rice);
      if (a > 3 * (1 << rice)) rice = imin(rice + 1, 4);
    }
    j++;
  }
}

#define TOK_HDR_CAP 192        // split flags + CU header + last-position bins waiting for the piece they open
#define TOK_ARENA 512          // tokens of one piece staged in LDS (larger pieces are written straight to the slot)
#define TOK_PIECES 17          // pieces one unit can produce: 4 CUs x (header-only | one per coded component) + the CTU's terminating bins

// One wave per 16x16 luma block ("unit") and ROLE -- luma, Cb, Cr, and (round 5) the CU HEADERS on a wave of their own: the header bins are a serial
// walk on one lane (3.3 us), which the luma wave -- the longest one -- no longer carries in front of its transform block -- or, ALLC, one
// wave per unit that takes the roles in turn.  Every wave works on its own, with its own LDS state and no workgroup barrier
// (NW waves per workgroup: NW = 4, the units of a 32x32 quadrant, was measured and is no faster than NW = 1 -- the kernel is bound by
// the number of waves to start and by its longest waves, not by workgroup dispatch; more waves per SIMD is: six -- 79 registers, the
// most that needs no scratch memory; eight spill, 5 MB of extra traffic per 1080p picture -- 4K 73 -> 67 us, profiles/r02_tokenizer_timeline.txt).
// A unit owns the CU that starts at its origin (32x32 or
// 16x16) or the four 8x8 CUs inside it; units covered by a 32x32 CU that starts elsewhere emit
// nothing.  The tokens leave the unit in PIECES, four per CU in coding order: its header (split flags, CU header, cu_qp_delta; the CTU's SAO syntax in
// front of the first), then one per coded transform block (luma -- led by its last-position bins --, Cb, Cr); piece 16: the CTU's terminating bins.  For each piece the
// lanes first COUNT the tokens of their 4x4 sub-blocks (the emitters run with a zero-capacity
// sink), the piece reserves its place in the CTU's slot with one atomicAdd, and a second run of the
// emitters writes the tokens at their final offsets -- through a small LDS arena when the piece
// fits, which keeps the kernel at ~10 KB of LDS per wave.  k_tok_compact restores coding order
// from the (offset, length) table [ctu][unit][piece].
// waves per SIMD the register budget is cut for: six (80 registers) spilled eight of them once a sub-block's levels stayed in registers (round 5); five (102) holds all
#ifndef KVZ_TOK_WAVES
#define KVZ_TOK_WAVES 5
#endif
struct alignas(16) TokWave {
  TuDigest dg;
  TokTabs tk;
  CoreTabs tabs;
  CuRec tile[9];
  uint16_t hdr[TOK_HDR_CAP];
  uint16_t arena[TOK_ARENA];
  int hdr_n;
  uint32_t piece_off;
  uint32_t seg[TOK_PIECES][2];
};
// HDRW: the headers on a wave of their own (grid z = 4); else the luma wave carries them (z = 3).  REGS: a sub-block's levels in registers through both passes
// (enc_subblock_regs: sixteen exec-masked positions per pass) or in LDS (hevc_core.h enc_subblock: a dependent LDS read per significant level).  At 1080p the
// kernel is as long as its longest wave and both help (31 -> 27.7 -> 24.2 us); at 2160p it is bound by the number of waves and of instructions, and both cost
// (55 -> 60 -> 63 us): pictures of more than 16 384 units keep round 4's form.
// one (unit, role) of k_tokenize on one wave.  have_tabs: the wave's tables are in LDS already (the list form: a wave's second item and later)
template <bool ALLC, bool HDRW, bool REGS>
__device__ __forceinline__ void tok_unit(const EncFrame &f, TokWave &W, const int ux, const int uy, const int comp, const int lane, const bool have_tabs)
{
  CuRec *tile = W.tile; TuDigest &dg = W.dg; CoreTabs &tabs = W.tabs; uint16_t *hdr = W.hdr, *arena = W.arena;
  int &hdr_n = W.hdr_n; uint32_t &piece_off = W.piece_off; uint32_t (&seg)[TOK_PIECES][2] = W.seg;
  // comp: 0 luma, 1 Cb, 2 Cr, 3 the headers and the CTU's terminating bins (ALLC: 0 stands for all of them)
  const int wc = f.cw >> 6, hc = f.ch >> 6;
  const int hdr_comp = HDRW ? 3 : 0;
  const bool hdr_role = ALLC || comp == hdr_comp;
  const int cx = ux >> 2, cy = uy >> 2, ctu = cy * wc + cx;
  const int X0 = ux * 16, Y0 = uy * 16;
  int z4 = 0;                                          // z-order index of the unit inside its CTU
  for (int b = 0; b < 2; b++) z4 |= (((ux & 3) >> b) & 1) << (2 * b) | (((uy & 3) >> b) & 1) << (2 * b + 1);
  // Most waves of an inter picture have nothing to say (units inside a 32x32 CU that starts elsewhere, chroma of CUs without
  // chroma residual): they find that out from a few bytes and leave with their table entries zeroed.
  unsigned long long *tr = (f.trace && hdr_role && lane == 0 && (z4 == 0 || z4 == 15)) ? f.trace + (size_t)wc * hc * 32 + (size_t)ctu * 8 : nullptr;   // (tools/tok_timeline.py)
  if (tr) tr[z4 == 0 ? 5 : 6] = wall_clock64();
  // wave census (tools/tok_timeline.py): per class {left at once, header only, with residual} x {P, I}: waves and 10 ns ticks
  unsigned long long *census = f.trace ? f.trace + (size_t)wc * hc * 40 + (size_t)ctu * 16 + (f.is_intra ? 6 : 0) : nullptr;
  const unsigned long long t_begin = census ? wall_clock64() : 0;
  int wave_class = 1;
  // phases of a luma wave that codes ONE coding unit with residual (tools/tok_phases.py; a build with EXTRA=-DKVZ_TOK_PHASES -- nine time stamps
  // held in registers cost the kernel 88 scalar spills): ticks per phase summed per CTU, slot 15 counts the waves
#ifdef KVZ_TOK_PHASES
  unsigned long long *ph = (f.trace && comp == 0 && !f.is_intra) ? f.trace + (size_t)wc * hc * 56 + (size_t)ctu * 16 : nullptr;
  unsigned long long ph_t[9]; int ph_n = 0;
#define TOK_PH() do { if (ph && ph_n < 9) ph_t[ph_n++] = wall_clock64(); } while (0)
#define TOK_PH_SYNC() wave_sync()
#else
#define TOK_PH() do { } while (0)
#define TOK_PH_SYNC() do { } while (0)
#endif
  TOK_PH();
  // table entries this wave writes: piece 4k + 0 the header of CU k (and piece 16, the terminators): the header role; 4k + 1 + c: component c
  auto mine = [&](int piece) { return ALLC || ((piece == 16 || (piece & 3) == 0) ? hdr_comp : (piece & 3) - 1) == comp; };
  if (!(hdr_role && z4 == 15)) {
    const int g0 = (uy * 2) * f.b8w + ux * 2, l0 = f.cu_log2[g0];
    bool work = !(l0 == 5 && ((X0 | Y0) & 31));
    if (!ALLC && work && comp != hdr_comp) {              // a component's wave (that does not also carry the headers): one of the unit's CUs has residual of it
      bool c = false;
      if (lane < (l0 == 3 ? 4 : 1)) { const int g = g0 + (lane >> 1) * f.b8w + (lane & 1); c = !(f.cu_flags[g] & CU_SKIP) && ((f.cu_cbf[g] >> comp) & 1); }
      work = __ballot(c) != 0;
    }
    if (!work) {
      if (lane < TOK_PIECES && mine(lane)) { uint32_t *e = f.tok_seg + ((size_t)(ctu * 16 + z4) * TOK_PIECES + lane) * 2; e[0] = 0; e[1] = 0; }
      if (census && lane == 0) { atomicAdd(&census[0], 1ull); atomicAdd(&census[1], wall_clock64() - t_begin); }
      return;
    }
  }
  TOK_PH();                                                        // 1: the unit has something to say
  if (!have_tabs) {
    for (int i = lane; i < (int)(sizeof(CoreTabs) / 4); i += 64) reinterpret_cast<uint32_t *>(&tabs)[i] = reinterpret_cast<const uint32_t *>(&g_core_tabs)[i];
    if (lane < (int)(sizeof(TokTabs) / 16)) reinterpret_cast<uint4 *>(&W.tk)[lane] = reinterpret_cast<const uint4 *>(&g_tok_tabs)[lane];
  }
  if (lane < TOK_PIECES) { seg[lane][0] = 0; seg[lane][1] = 0; }
  if (lane == 0) hdr_n = 0;
  const int bx0 = ux * 2 - 1, by0 = uy * 2 - 1;
  if (lane < 9) {
    int bx = bx0 + lane % 3, by = by0 + lane / 3;
    CuRec r; r.log2 = 0; r.intra = 0; r.flags = 0; r.merge_idx = 0; r.mvp_idx = 0; r.intra_mode = 0; r.cbf = 0; r.pad = 0; r.mvdx = 0; r.mvdy = 0;
    if (bx >= 0 && by >= 0) {
      int g = by * f.b8w + bx;
      r.log2 = f.cu_log2[g]; r.intra = f.cu_intra[g]; r.flags = f.cu_flags[g]; r.merge_idx = f.cu_merge_idx[g];
      r.mvp_idx = f.cu_mvp_idx[g]; r.intra_mode = f.cu_intra_mode[g]; r.cbf = f.cu_cbf[g];
      r.mvdx = f.cu_mvd[g * 2]; r.mvdy = f.cu_mvd[g * 2 + 1];
    }
    tile[lane] = r;
  }
  wave_sync();
  TOK_PH();                                                        // 2: tables and the 3 x 3 records in LDS
  uint16_t *slot = f.tok_buf + (size_t)ctu * f.tok_cap;
  int np = 0;                                          // piece being produced (wave-uniform): CU k, component c -> 4k + c; 16 = terminators
  // reserve `total` tokens of the CTU's slot for piece np; all lanes get the offset (or ~0u when the slot is full)
  auto reserve = [&](int total) -> uint32_t {
    if (lane == 0) {
      uint32_t o = atomicAdd(&f.tok_cursor[ctu], (uint32_t)total);
      if (o + (uint32_t)total > (uint32_t)f.tok_cap) { atomicOr(f.err, 16u); o = ~0u; }
      else { seg[np][0] = o; seg[np][1] = (uint32_t)total; }
      piece_off = o;
    }
    wave_sync();
    return piece_off;
  };
  TileView v; v.tile = tile; v.bx0 = bx0; v.by0 = by0;
  const int cl0 = v.at(X0, Y0).log2;
  const bool owner = !(cl0 == 5 && ((X0 | Y0) & 31));
  const int ncu = owner ? (cl0 == 3 ? 4 : 1) : 0;
  for (int k = 0; k < ncu; k++) {
    const int x0 = X0 + ((cl0 == 3) ? (k & 1) * 8 : 0), y0 = Y0 + ((cl0 == 3) ? (k >> 1) * 8 : 0);
    const CuRec cu = v.at(x0, y0);
    int z = 0;                                         // z-order index (8x8 units) of the CU origin inside the CTU
    for (int b = 0; b < 3; b++) z |= ((((x0 & 63) >> 3) >> b) & 1) << (2 * b) | ((((y0 & 63) >> 3) >> b) & 1) << (2 * b + 1);
    if (lane == 0 && hdr_role) {
      TokOut t; t.tabs = &tabs; t.p = hdr; t.n = 0; t.cap = TOK_HDR_CAP;
      if (f.sao && z4 == 0 && k == 0) {                 // coding_tree_unit() starts with sao()
        const bool hl = cx > 0 && !tile_col_starts_at(wc, f.tile_cols, cx), hu = cy > 0 && !tile_row_starts_at(hc, f.tile_rows, cy);
        enc_sao(t, f.sao[ctu], hl ? &f.sao[ctu - 1] : nullptr, hu ? &f.sao[ctu - wc] : nullptr);     // (read where they lie: a local copy indexed at run time would live in scratch memory)
      }
      enc_split_flags(v, t, f.cw, f.chp, x0, y0, z, cu.log2);
      enc_cu_header(v, t, f.cw, f.chp, f.is_intra != 0, x0, y0, cu, f.lossless != 0);
      // the quantisation group's (= CTU's) delta QP goes with its first CU that has residual, right after the cbf flags
      if (f.ctu_qy && !(cu.flags & CU_SKIP) && cu.cbf && z == f.ctu_first[ctu]) enc_cu_qp_delta(t, f.ctu_delta[ctu]);
      hdr_n = t.n;
    }
    TOK_PH_SYNC(); TOK_PH();                                       // 3: lane 0 has the header bins
    if (hdr_role) {                                                 // the CU's header: piece 4k + 0
      np = 4 * k;
      wave_sync();
      const int hn = hdr_n > TOK_HDR_CAP ? TOK_HDR_CAP : hdr_n;
      if (hdr_n > TOK_HDR_CAP && lane == 0) atomicOr(f.err, 8u);
      const uint32_t o = reserve(hn);
      if (o != ~0u) for (int i = lane; i < hn; i += 64) slot[o + i] = hdr[i];
      wave_sync();
      if (lane == 0) hdr_n = 0;
      wave_sync();
    }
    const int cbf = (cu.flags & CU_SKIP) ? 0 : cu.cbf;          // wave-uniform
    for (int ci = (ALLC || comp == 3) ? 0 : comp; ci < (ALLC ? 3 : (comp == 3 ? 0 : comp + 1)); ci++) {   // the CU's transform blocks: luma, Cb, Cr (ALLC) or this wave's component (a wave that only carries headers: none)
      np = 4 * k + 1 + ci;
      if ((cbf >> ci) & 1) {
        wave_class = 2;
        const int l2 = ci ? cu.log2 - 1 : cu.log2, pw = ci ? (f.cw >> 1) : f.cw;
        const int px = ci ? (x0 >> 1) : x0, py = ci ? (y0 >> 1) : y0;
        const int scan = intra_scan_idx(cu.intra, l2, ci, cu.intra_mode);
        SbRegs sb;
        digest_build_wave<!REGS>(&tabs, dg, f.coef[ci] + py * pw + px, pw, l2, scan, lane, sb);   // ends with a barrier
        TOK_PH();                                                  // 4: digest
        const uint64_t sbm = dg.sbmask;
        const int last_sb = 63 - __builtin_clzll(sbm);
        const int last_pos = 31 - __builtin_clz((uint32_t)dg.mask[last_sb]);
        if (lane == 0) {
          TokOut t; t.tabs = &tabs; t.p = hdr; t.n = hdr_n; t.cap = TOK_HDR_CAP;
          int a, b; enc_last_pos(t, dg, l2, ci, scan, a, b);
          hdr_n = t.n;
        }
        // greater1 context-set carry: sub-block i inherits from the next non-empty sub-block above it
        const bool nzsb = (sbm >> lane) & 1;
        const bool g1 = nzsb && sb_g1_any(sb);
        const uint64_t g1m = __ballot(g1);
        bool prev_g1 = false;
        if (lane <= last_sb) {
          uint64_t above = (lane < 63) ? (sbm >> (lane + 1)) : 0;      // non-empty sub-blocks coded before this one
          if (above) { int j = lane + 1 + __builtin_ctzll(above); prev_g1 = (g1m >> j) & 1; }
        }
        // the sub-block's sixteen sig_coeff_flag context patterns: by the neighbouring sub-blocks' coded flags, or the 4x4 block's own map
        uint32_t sk[4] = {0, 0, 0, 0};
        if (lane <= last_sb) {
          const int sbl = l2 - 2, nsb = 1 << sbl;
          int xs, ys; scan_pos(&tabs, scan, sbl, lane, xs, ys);
          const int right = (xs < nsb - 1) ? dg.csbf[ys * 8 + xs + 1] : 0, below = (ys < nsb - 1) ? dg.csbf[(ys + 1) * 8 + xs] : 0;
          const uint4 q = *reinterpret_cast<const uint4 *>(W.tk.sigk[scan][l2 == 2 ? 4 : (right | (below << 1))]);
          sk[0] = q.x; sk[1] = q.y; sk[2] = q.z; sk[3] = q.w;
        }
        // pass 1: count
        int n_l = 0;
        if (lane <= last_sb) {
          TokCount t; t.tabs = &tabs; t.n = 0;
          if (REGS) enc_subblock_regs(t, dg, sb, sk, lane, last_sb, last_pos, prev_g1, l2, ci, scan, f.signhide);
          else enc_subblock(t, dg, lane, last_sb, last_pos, prev_g1, l2, ci, scan, f.signhide);
          n_l = t.n;
        }
        // offsets in coding order: sub-block last_sb first, then downwards
        int suffix = n_l;
        for (int o = 1; o < 64; o <<= 1) { int other = __shfl_down(suffix, o); if (lane + o < 64) suffix += other; }
        const int body = __shfl(suffix, 0);
        const int off = suffix - n_l;                                 // tokens of all sub-blocks with a higher index
        wave_sync();                                              // hdr_n of lane 0 visible
        const int hn = hdr_n > TOK_HDR_CAP ? TOK_HDR_CAP : hdr_n, total = hn + body;
        if (hdr_n > TOK_HDR_CAP && lane == 0) atomicOr(f.err, 8u);
        TOK_PH();                                                  // 5: last position, greater1 carry, count pass, offsets
        const uint32_t o = reserve(total);
        TOK_PH();                                                  // 6: place reserved
        if (o != ~0u) {
          const bool staged = total <= TOK_ARENA;
          uint16_t *dst = staged ? arena : slot + o;
          for (int i = lane; i < hn; i += 64) dst[i] = hdr[i];
          if (lane <= last_sb && n_l) {                               // pass 2: the same emitters, now writing at the final offsets
            TokOut t; t.tabs = &tabs; t.p = dst + hn + off; t.n = 0; t.cap = n_l;
            if (REGS) enc_subblock_regs(t, dg, sb, sk, lane, last_sb, last_pos, prev_g1, l2, ci, scan, f.signhide);
            else enc_subblock(t, dg, lane, last_sb, last_pos, prev_g1, l2, ci, scan, f.signhide);
          }
          if (staged) {
            wave_sync();
            for (int i = lane; i < total; i += 64) slot[o + i] = arena[i];
          }
        }
        wave_sync();
        TOK_PH();                                                  // 7: tokens written
        if (lane == 0) hdr_n = 0;
      }
      wave_sync();
    }
  }
  if (z4 == 15 && hdr_role) {                                       // the last unit of the CTU closes it
    const bool last = (cy == hc - 1 && cx == wc - 1);
    const bool row_end = tile_col_ends_at(wc, f.tile_cols, cx);                                   // last CTU of its row inside the tile
    const bool sub_end = row_end && (f.wpp || tile_row_ends_at(hc, f.tile_rows, cy));
    const bool seg_end = last || (row_end && (f.slices == 1 || (f.slices == 2 && tile_row_ends_at(hc, f.tile_rows, cy))));   // slice segments per CTU row / per tile
    const int n = 1 + ((sub_end && !seg_end) ? 1 : 0);
    wave_sync();
    np = 16;
    const uint32_t o = reserve(n);
    if (o != ~0u && lane == 0) {
      slot[o] = (uint16_t)(0xC000u | (seg_end ? 1u : 0u));          // end_of_slice_segment_flag
      if (n == 2) slot[o + 1] = 0xC001u;                            // end_of_subset_one_bit
    }
  }
  wave_sync();
  if (lane < TOK_PIECES && mine(lane)) {
    uint32_t *e = f.tok_seg + ((size_t)(ctu * 16 + z4) * TOK_PIECES + lane) * 2;
    e[0] = seg[lane][0]; e[1] = seg[lane][1];
  }
  TOK_PH();                                                        // 8: table entries out
#ifdef KVZ_TOK_PHASES
  if (ph && lane == 0 && ph_n == 9 && ncu == 1 && wave_class == 2) { for (int k = 0; k < 8; k++) atomicAdd(&ph[k], ph_t[k + 1] - ph_t[k]); atomicAdd(&ph[15], 1ull); }
#endif
#undef TOK_PH
#undef TOK_PH_SYNC
  if (tr && z4 == 15) tr[7] = wall_clock64();
  if (census && lane == 0) { atomicAdd(&census[wave_class * 2], 1ull); atomicAdd(&census[wave_class * 2 + 1], wall_clock64() - t_begin); }
}
// LIST (P pictures, whole pictures; round 6): the launch is a fixed number of waves that work off k_inter_signal's list of (unit, role) pairs with something to
// say (EncFrame::tok_list) instead of a wave per pair -- the pairs that are not listed keep the zero table entries k_tok_compact leaves behind.  The kernel was
// the time to start 32 640 waves (130 000 at 2160p) of which nine in ten left at once, plus its longest wave; now it is its longest wave.
template <bool ALLC, int NW, bool HDRW = true, bool REGS = true, bool LIST = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(REGS ? (LIST ? 3 : KVZ_TOK_WAVES) : 6))) void k_tokenize(EncFrame f)      // (the list form: few waves, each with work -- the registers it wants: at five waves per SIMD the item loop spilled 23)
{
  __shared__ TokWave tw[NW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  TokWave &W = tw[wv];
  if constexpr (LIST) {
    const uint32_t count = f.tok_list[0], uw = (uint32_t)(f.cw >> 4);
    bool have = false;
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
      const uint32_t v = f.tok_list[1 + it], unit = v >> 2;
      tok_unit<ALLC, HDRW, REGS>(f, W, (int)(unit % uw), (int)(unit / uw), (int)(v & 3u), lane, have);
      have = true;
      wave_sync();
    }
  } else {
    const int ux = NW == 4 ? blockIdx.x * 2 + (wv & 1) : blockIdx.x, uy = NW == 4 ? (blockIdx.y + f.row0 * 2) * 2 + (wv >> 1) : blockIdx.y + f.row0 * 4, comp = ALLC ? 0 : (int)blockIdx.z;
    tok_unit<ALLC, HDRW, REGS>(f, W, ux, uy, comp, lane, false);
  }
}

// one workgroup per CTU: the pieces of its 16 units in z-order, piece after piece -> the dense token array (CTUs in coding order)
// (tok_count_out[ctu] < 0 tells the host that the CTU did not fit or its table was inconsistent)
__global__ __launch_bounds__(256) void k_tok_compact(EncFrame f)
{
  constexpr int NSEG = 16 * TOK_PIECES;
  __shared__ uint32_t soff[NSEG], start[NSEG + 1], utot[17];
  __shared__ uint32_t wsum[4];
  const int first = f.row0 * (f.cw >> 6), ctu = blockIdx.x + first, tid = threadIdx.x;
  unsigned long long *tr = f.trace ? f.trace + (size_t)(f.cw >> 6) * (f.ch >> 6) * 32 + (size_t)ctu * 8 : nullptr;     // (tools/tok_timeline.py)
#define TC_STAMP(k) do { if (tr && tid == 0) tr[k] = wall_clock64(); } while (0)
  TC_STAMP(0);
  const uint32_t n = f.tok_cursor[ctu];
  // The CTU's place in the dense array: the tokens of all CTUs before it (every workgroup adds up the counts itself -- a few KB out of
  // L2 -- rather than queueing at one atomic counter: 2040 same-address atomics cost 25 ns each, which WAS this kernel's run time).
  // The next picture's cursors (the other of two arrays) go back to zero, and the device error word over to the host.
  uint32_t part = 0;
  for (int i = first + tid; i < ctu; i += 256) part += f.tok_cursor[i];
  part = wave_sum_u32(part);
  if ((tid & 63) == 0) wsum[tid >> 6] = part;
  if (tid == 0) { f.tok_cursor_next[ctu] = 0; if (blockIdx.x == 0) { *f.err_out = *f.err; if (f.ent_cursors) { f.ent_cursors[0] = 0; f.ent_cursors[1] = 0; } } }
  __syncthreads();
  const uint32_t base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  TC_STAMP(1);
  // the CTU's table goes to LDS and back to zero in global memory: the next picture's tokenizer may be the list form, whose waves only write the entries of
  // the (unit, role) pairs that have something to say -- every other entry must read "no tokens"
  uint32_t *tab = f.tok_seg + (size_t)ctu * NSEG * 2;
  for (int i = tid; i < NSEG; i += 256) { const uint2 e = *reinterpret_cast<const uint2 *>(&tab[i * 2]); soff[i] = e.x; start[i] = e.y; *reinterpret_cast<uint2 *>(&tab[i * 2]) = make_uint2(0u, 0u); }        // start[] holds lengths for now
  if (f.tok_list && blockIdx.x == 0 && tid == 0) f.tok_list[0] = 0;      // (the list has been worked off: k_tokenize ran before this launch on the same stream)
  if (base + n > f.tok_dense_cap) { if (tid == 0) f.tok_count_out[ctu] = -1; return; }
  __syncthreads();
  TC_STAMP(2);
  if (tid < 16) { uint32_t a = 0; for (int p = 0; p < TOK_PIECES; p++) { uint32_t l = start[tid * TOK_PIECES + p]; start[tid * TOK_PIECES + p] = a; a += l; } utot[tid] = a; }
  __syncthreads();
  if (tid == 0) { uint32_t a = 0; for (int u = 0; u < 16; u++) { uint32_t t = utot[u]; utot[u] = a; a += t; } utot[16] = a; }
  __syncthreads();
  for (int i = tid; i < NSEG; i += 256) start[i] += utot[i / TOK_PIECES];
  if (tid == 0) start[NSEG] = utot[16];
  __syncthreads();
  const uint32_t total = start[NSEG];
  TC_STAMP(3);
  if (total != n) { if (tid == 0) f.tok_count_out[ctu] = -1; return; }
  const uint16_t *slot = f.tok_buf + (size_t)ctu * f.tok_cap;
  uint16_t *dst = f.tok_dense + base;
  int lo = 0;                                           // last segment with start <= i (empty segments share a start: take the last)
  for (uint32_t i = tid; i < n; i += 256) {
    if (start[lo + 1] <= i) {                           // past the segment of the previous round (long pieces: usually not)
      int hi = NSEG;
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (start[mid] <= i) lo = mid; else hi = mid; }
    }
    dst[i] = slot[soff[lo] + (i - start[lo])];
  }
  if (tid == 0) { f.tok_off_out[ctu] = base; f.tok_count_out[ctu] = (int32_t)n; }
  TC_STAMP(4);
#undef TC_STAMP
}

// =============================================================================================
// Input staging: packed I420 picture (w x h) -> coded planes padded to (cw x ch) by edge
// replication (what the oracle's load_input does)
// =============================================================================================
__global__ __launch_bounds__(256) void k_pad_input(const uint8_t *in, int w, int h, uint8_t *dy, uint8_t *du, uint8_t *dv, int cw, int ch)
{
  // A workgroup moves 1024 bytes x 16 rows: thread (x, q) the 16-byte piece x of rows 4q .. 4q + 3 -- four loads in flight, then four
  // stores.  blockIdx.y counts groups of 16 rows through luma, Cb, Cr (packed I420 input: Y, U, V planes back to back; the planes'
  // coded heights are multiples of 32, so a group never straddles two planes).
  int y0 = blockIdx.y * 16 + threadIdx.y * 4, plane = 0;
  if (y0 >= ch) { y0 -= ch; plane = 1; if (y0 >= ch / 2) { y0 -= ch / 2; plane = 2; } }
  const int pw = plane ? w / 2 : w, ph = plane ? h / 2 : h, pcw = plane ? cw / 2 : cw;
  const int x = (blockIdx.x * 64 + threadIdx.x) * 16;
  if (x >= pcw) return;
  const uint8_t *src = in + (plane == 0 ? 0 : (plane == 1 ? (size_t)w * h : (size_t)w * h + (size_t)(w / 2) * (h / 2)));
  uint8_t *dst = plane == 0 ? dy : (plane == 1 ? du : dv);
  kv_u32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint8_t *row = src + (size_t)imin(y0 + k, ph - 1) * pw;
    if (x + 16 <= pw && (((uintptr_t)(row + x)) & 3) == 0) v[k] = *reinterpret_cast<const kv_u32x4 *>(row + x);      // the common case: one 16-byte load
    else {                                                     // the picture's right edge (replicated) or an odd alignment: byte by byte
      uint32_t q[4] = {0, 0, 0, 0};
      for (int i = 0; i < 16; i++) q[i >> 2] |= (uint32_t)row[imin(x + i, pw - 1)] << (8 * (i & 3));
      v[k].x = q[0]; v[k].y = q[1]; v[k].z = q[2]; v[k].w = q[3];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) *reinterpret_cast<kv_u32x4 *>(dst + (size_t)(y0 + k) * pcw + x) = v[k];          // coded widths are multiples of 64: 16-byte aligned
}

// =============================================================================================
// Sample adaptive offset (8.7.3).  One workgroup per CTU: the deblocked CTU with a one-sample border goes to LDS for all
// three components.  Encoder: statistics against the source picture and "uvgx SAO decision v1" (statement of record:
// oracle/hevc_sao.c), parameters out; decoder: parameters in.  Both: the filtered CTU goes to the output picture.
// =============================================================================================
#define KVZ_SAO_THREADS 1024   // (a CTU's workgroup is alone on its compute unit in uvgComm's default mode: four waves per SIMD hide what one wave waited for -- tools/sao_timeline.py)
struct SaoLds {
  // the deblocked CTU with a one-sample border, one padded picture per component: sample (x, y), x and y in -1 .. n, at
  // [(y + 1) * pitch + 4 + x]; pitch 72 / 40 keeps the CTB's own samples dword aligned
  alignas(4) uint8_t win[66 * 72];
  alignas(4) uint8_t winc[2][34 * 40];
  int en[3][4][5], es[3][4][5], bn[3][32][16], bs[3][32][16];      // (band statistics in 16 copies, by lane: neighbouring columns mostly fall into the same band, and LDS atomics on one address go one after the other)
  int eo_off[3][4][4], eo_dist[3][4][4], bo_off[3][32], bo_gain[3][32];
  int cand_dist[3][5], cand_bins[3][5], cand_band[3];
  SaoParams p;
};
__device__ __forceinline__ int sao_edge_idx(int c, int a, int b)
{
  const int e = 2 + ((c > a) - (c < a)) + ((c > b) - (c < b));
  return e == 2 ? 0 : (e < 2 ? e + 1 : e);
}
__device__ __forceinline__ int sao_rdiv(int a, int b) { return b == 0 ? 0 : (a >= 0 ? (2 * a + b) / (2 * b) : -((-2 * a + b) / (2 * b))); }
__device__ __forceinline__ int sao_off_bins(int o) { const int a = o < 0 ? -o : o; return a < 7 ? a + 1 : 7; }

// Statistics of one piece of a column of a CTB (ROWS rows from row ys of column x) with the 3 x 3 neighbourhood in registers.  Edge classes: four packed
// accumulators (see inside), reduced over the wave with DPP at the end -- one LDS atomic per wave and statistic.  Bands: runs of equal band index along the
// column are summed in registers and flushed when the band changes.  o8: the source samples of the piece (fetched by the caller long before).
template <int ROWS>
__device__ __forceinline__ void sao_stats_column(SaoLds &s, int c, const uint8_t *w, int pitch, int x, int ys, bool okh, int Y0, int ph, const int *o8, int tid)
{
  // the sign of a difference as med3(d, -1, 1); the edge index e = 2 + sign(c - a) + sign(c - b) is 0, 1 / 3, 4 for the categories 1, 2 / 3, 4 (2: none)
  auto sgn = [](int d) { return d > 1 ? 1 : (d < -1 ? -1 : d); };
  const uint8_t *col = w + 4 + x - 1;                                   // column x - 1 of window row 0 (= sample row -1)
  int r0[3], r1[3], r2[3];
  for (int k = 0; k < 3; k++) { r0[k] = col[ys * pitch + k]; r1[k] = col[(ys + 1) * pitch + k]; }
  // per class one 64-bit accumulator with four 16-bit fields, one per category: sum of (d + 256) + 2048 per sample -- the count rides above the sum
  // (ROWS <= 4: sum <= 4 * 511 < 2048)
  uint64_t acc[4] = {0, 0, 0, 0};
  int run_b = -1, run_n = 0, run_s = 0;
  int s_up = sgn(r1[1] - r0[1]);                                        // sign(c - above) of the row to come
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int y = ys + j;
    for (int k = 0; k < 3; k++) r2[k] = col[(y + 2) * pitch + k];
    const int v = r1[1], d = o8[j] - v, b = v >> 3;
    if (b != run_b) {
      if (run_n) { atomicAdd(&s.bn[c][run_b][tid & 15], run_n); atomicAdd(&s.bs[c][run_b][tid & 15], run_s); }
      run_b = b; run_n = 0; run_s = 0;
    }
    run_n++; run_s += d;
    const bool okv = Y0 + y - 1 >= 0 && Y0 + y + 1 < ph;
    const int s_dn = sgn(v - r2[1]);
    const int ea = okh ? 2 + sgn(v - r1[0]) + sgn(v - r1[2]) : 2, eb = okv ? 2 + s_up + s_dn : 2;
    const int ec = (okh && okv) ? 2 + sgn(v - r0[0]) + sgn(v - r2[2]) : 2, ed = (okh && okv) ? 2 + sgn(v - r0[2]) + sgn(v - r2[0]) : 2;
    const int ee[4] = {ea, eb, ec, ed};
    const uint32_t val = (uint32_t)(d + 256 + 2048);
#pragma unroll
    for (int e = 0; e < 4; e++) acc[e] += (uint64_t)(ee[e] == 2 ? 0u : val) << (16 * (ee[e] - (ee[e] > 2)));
    s_up = -s_dn;                                                       // the next row's sign(c - above)
    for (int k = 0; k < 3; k++) { r0[k] = r1[k]; r1[k] = r2[k]; }
  }
  if (run_n) { atomicAdd(&s.bn[c][run_b][tid & 15], run_n); atomicAdd(&s.bs[c][run_b][tid & 15], run_s); }
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int k = 1; k <= 4; k++) {
      // one reduction for both: the wave's count (<= 64 * ROWS) above bit 20, its biased sum (<= 64 * ROWS * 511) below
      const uint32_t fld = (uint32_t)(acc[e] >> (16 * (k - 1))) & 0xffffu, cnt = fld >> 11, sum = fld & 2047u;
      const uint32_t r = wave_sum_u32((cnt << 20) | sum), N = r >> 20, S = r & 0xfffffu;
      if ((tid & 63) == 0) { atomicAdd(&s.en[c][e][k], (int)N); atomicAdd(&s.es[c][e][k], (int)S - 256 * (int)N); }
    }
}

template <bool DEC>
__global__ __launch_bounds__(KVZ_SAO_THREADS) void k_sao(EncFrame f)
{
  __shared__ SaoLds s;
  const int tid = threadIdx.x, wc = f.cw >> 6, ctu = xcd_contiguous((int)blockIdx.x, (int)gridDim.x), cx = ctu % wc, cy = ctu / wc;
  unsigned long long *tr = (!DEC && f.trace && tid == 0) ? f.trace + (size_t)wc * (f.ch >> 6) * 56 + (size_t)ctu * 16 : nullptr;      // (tools/sao_timeline.py)
#define SAO_STAMP(k) do { if (tr) tr[k] = wall_clock64(); } while (0)
  SAO_STAMP(0);
  constexpr int T = KVZ_SAO_THREADS;
  // the source samples of this thread's pieces of columns (sao_stats_column): in flight while the window is fetched
  const int lx = tid & 63, lys = (tid >> 6) * 4;                                         // luma: column lx, rows lys .. lys + 3
  const int cc = 1 + (tid >> 9), cxx = tid & 31, cys = ((tid & 511) >> 5) * 2;           // chroma component cc: column cxx, rows cys, cys + 1
  int o8l[4] = {0, 0, 0, 0}, o8c[2] = {0, 0};
  if (!DEC) {
    const uint8_t *ol = f.src[0] + (size_t)(cy * 64 + lys) * f.cw + cx * 64 + lx, *oc = f.src[cc] + (size_t)(cy * 32 + cys) * (f.cw >> 1) + cx * 32 + cxx;
#pragma unroll
    for (int j = 0; j < 4; j++) o8l[j] = ol[(size_t)j * f.cw];
#pragma unroll
    for (int j = 0; j < 2; j++) o8c[j] = oc[(size_t)j * (f.cw >> 1)];
  }
  // ---- window: the CTB's samples as dwords, the border ring byte by byte (clamped at the picture edges; those samples are
  // never used: a neighbour outside the picture switches the edge offset off)
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, l2n = 6 - sh, n = 1 << l2n, pw = f.cw >> sh, ph = f.ch >> sh, X0 = cx * n, Y0 = cy * n;
    uint8_t *w = c ? s.winc[c - 1] : s.win; const int pitch = c ? 40 : 72;
    const uint8_t *src = f.rec[c];
#pragma unroll
    for (int i = tid; i < n * n / 4; i += T) {
      const int y = i >> (l2n - 2), x = (i & ((n >> 2) - 1)) * 4;
      *(uint32_t *)&w[(y + 1) * pitch + 4 + x] = *(const uint32_t *)&src[(size_t)(Y0 + y) * pw + X0 + x];
    }
    for (int q = tid; q < 4 * n + 4; q += T) {
      int x, y;
      if (q < n + 2) { x = q - 1; y = -1; } else if (q < 2 * n + 4) { x = q - (n + 2) - 1; y = n; }
      else if (q < 3 * n + 4) { x = -1; y = q - (2 * n + 4); } else { x = n; y = q - (3 * n + 4); }
      w[(y + 1) * pitch + 4 + x] = src[(size_t)clip3(0, ph - 1, Y0 + y) * pw + clip3(0, pw - 1, X0 + x)];
    }
  }
  if (DEC) { if (tid == 0) s.p = f.sao[ctu]; }
  else for (int i = tid; i < (int)((sizeof(s.en) + sizeof(s.es) + sizeof(s.bn) + sizeof(s.bs)) / sizeof(int)); i += T) (&s.en[0][0][0])[i] = 0;
  __syncthreads();
  SAO_STAMP(1);
  if (!DEC) {
    // Statistics: sao_stats_column -- the luma CTB's columns in pieces of four rows over all 1024 threads, then both chroma CTBs' in pieces of two
    sao_stats_column<4>(s, 0, s.win, 72, lx, lys, cx * 64 + lx - 1 >= 0 && cx * 64 + lx + 1 < f.cw, cy * 64, f.ch, o8l, tid);
    sao_stats_column<2>(s, cc, s.winc[cc - 1], 40, cxx, cys, cx * 32 + cxx - 1 >= 0 && cx * 32 + cxx + 1 < (f.cw >> 1), cy * 32, f.ch >> 1, o8c, tid);
    __syncthreads();
    SAO_STAMP(2);
    if (tid < 48) {                                       // edge offsets: component x class x category
      const int c = tid >> 4, e = (tid >> 2) & 3, k = (tid & 3) + 1, N = s.en[c][e][k], S = s.es[c][e][k];
      int o = sao_rdiv(S, N);
      o = k <= 2 ? clip3(0, 7, o) : clip3(-7, 0, o);
      s.eo_off[c][e][k - 1] = o; s.eo_dist[c][e][k - 1] = N * o * o - 2 * o * S;
    } else if (tid >= 64 && tid < 160) {                  // band offsets: component x band
      const int c = (tid - 64) >> 5, b = (tid - 64) & 31;
      int N = 0, S = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) { N += s.bn[c][b][k]; S += s.bs[c][b][k]; }
      const int o = clip3(-7, 7, sao_rdiv(S, N));
      s.bo_off[c][b] = o; s.bo_gain[c][b] = N * o * o - 2 * o * S;
    }
    __syncthreads();
    if (tid < 15) {                                       // candidates: component x [EO 0..3, BO]
      const int c = tid / 5, j = tid - c * 5;
      int dist = 0, bins = 0;
      if (j < 4) { for (int k = 0; k < 4; k++) { dist += s.eo_dist[c][j][k]; bins += sao_off_bins(s.eo_off[c][j][k]); } }
      else {
        int best = 0;
        for (int st = 0; st <= 28; st++) {
          const int g = s.bo_gain[c][st] + s.bo_gain[c][st + 1] + s.bo_gain[c][st + 2] + s.bo_gain[c][st + 3];
          if (st == 0 || g < dist) { dist = g; best = st; }
        }
        bins = 5;
        for (int k = 0; k < 4; k++) { const int o = s.bo_off[c][best + k]; bins += sao_off_bins(o) + (o != 0); }
        s.cand_band[c] = best;
      }
      s.cand_dist[c][j] = dist; s.cand_bins[c][j] = bins;
    }
    __syncthreads();
    SAO_STAMP(3);
    if (tid == 0) {
      const long long l2 = (long long)f.lambda_q4 * f.lambda_q4;
      int pick[2] = {0, 0};
      long long best = l2;
      for (int t = 1; t <= 5; t++) {
        const long long cost = 256LL * s.cand_dist[0][t - 1] + l2 * (2 + s.cand_bins[0][t - 1] + (t <= 4 ? 2 : 0));
        if (cost < best) { best = cost; pick[0] = t; }
      }
      best = l2;
      for (int t = 1; t <= 5; t++) {
        const long long cost = 256LL * ((long long)s.cand_dist[1][t - 1] + s.cand_dist[2][t - 1]) + l2 * (2 + s.cand_bins[1][t - 1] + s.cand_bins[2][t - 1] + (t <= 4 ? 2 : 0));
        if (cost < best) { best = cost; pick[1] = t; }
      }
      SaoParams p;
      for (int c = 0; c < 3; c++) {
        const int t = pick[c ? 1 : 0];
        p.type[c] = t == 0 ? 0 : (t == 5 ? 1 : 2);
        p.eo_class[c] = (t >= 1 && t <= 4) ? (uint8_t)(t - 1) : 0;
        p.band_pos[c] = t == 5 ? (uint8_t)s.cand_band[c] : 0;
        for (int k = 0; k < 4; k++) p.offset[c][k] = (int8_t)(t == 0 ? 0 : (t == 5 ? s.bo_off[c][s.cand_band[c] + k] : s.eo_off[c][t - 1][k]));
      }
      s.p = p; f.sao[ctu] = p;
    }
  }
  __syncthreads();
  SAO_STAMP(4);
  // ---- the filter: four samples of a row per thread, the rows above and below as 6-byte spans (samples x - 1 .. x + 4)
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, l2n = 6 - sh, n = 1 << l2n, pw = f.cw >> sh, ph = f.ch >> sh, X0 = cx * n, Y0 = cy * n;
    const uint8_t *w = c ? s.winc[c - 1] : s.win; const int pitch = c ? 40 : 72;
    const int type = s.p.type[c], e = s.p.eo_class[c], bp = s.p.band_pos[c];
    int off[4];
    for (int k = 0; k < 4; k++) off[k] = s.p.offset[c][k];
#pragma unroll
    for (int q = tid; q < n * n / 4; q += T) {
      const int y = q >> (l2n - 2), x4 = (q & ((n >> 2) - 1)) * 4;
      const uint32_t *row = (const uint32_t *)&w[(y + 1) * pitch + x4];        // dword holding samples x4 - 4 .. x4 - 1
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
        const bool okv = Y0 + y - 1 >= 0 && Y0 + y + 1 < ph;
        uint32_t o = 0;
        for (int i = 0; i < 4; i++) {
          const int v = (int)((mid >> (8 * (i + 1))) & 255);
          int a, b; bool ok;
          const bool okh = X0 + x4 + i - 1 >= 0 && X0 + x4 + i + 1 < pw;
          if (e == 0) { a = (int)((mid >> (8 * i)) & 255); b = (int)((mid >> (8 * (i + 2))) & 255); ok = okh; }
          else if (e == 1) { a = (int)((up >> (8 * (i + 1))) & 255); b = (int)((dn >> (8 * (i + 1))) & 255); ok = okv; }
          else if (e == 2) { a = (int)((up >> (8 * i)) & 255); b = (int)((dn >> (8 * (i + 2))) & 255); ok = okh && okv; }
          else { a = (int)((up >> (8 * (i + 2))) & 255); b = (int)((dn >> (8 * i)) & 255); ok = okh && okv; }
          const int k = ok ? sao_edge_idx(v, a, b) : 0;
          o |= (uint32_t)(k ? clip8(v + off[k - 1]) : v) << (8 * i);
        }
        out = o;
      }
      *(uint32_t *)&f.sao_out[c][(size_t)(Y0 + y) * pw + X0 + x4] = out;
    }
  }
  SAO_STAMP(5);
#undef SAO_STAMP
}

// =============================================================================================
// Variance adaptive quantisation, "uvgx VAQ v1" (statement of record: vaq_deltas() in oracle/hevc_enc.c).
// k_vaq_stats: one workgroup per CTU -- sum and sum of squares of its 64x64 source luma samples, activity e = log2 of the
// variance in 1/16 steps, e to act[ctu] and into the picture's sum.  k_vaq_apply: the picture mean, the CTU's delta, and the
// target QP = clip(qp + clip(roi delta + vaq delta)) written over the ROI delta the host put into ctu_qt.
// =============================================================================================
__global__ __launch_bounds__(256) void k_vaq_stats(EncFrame f, int *act, int *sum)
{
  __shared__ uint32_t p1[4], p2[4];
  const int tid = threadIdx.x, wc = f.cw >> 6, cx = blockIdx.x % wc, cy = blockIdx.x / wc;
  const uint8_t *src = f.src[0] + (size_t)(cy * 64) * f.cw + cx * 64;
  uint32_t s1 = 0, s2 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {                                     // 1024 dwords of the CTU, four per thread
    const int i = tid + 256 * k, y = i >> 4, x = (i & 15) * 4;
    const uint32_t v = *(const uint32_t *)&src[(size_t)y * f.cw + x];
#pragma unroll
    for (int b = 0; b < 4; b++) { const uint32_t px = (v >> (8 * b)) & 255u; s1 += px; s2 += px * px; }
  }
  s1 = wave_sum_u32(s1); s2 = wave_sum_u32(s2);
  if ((tid & 63) == 0) { p1[tid >> 6] = s1; p2[tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    const uint32_t a = p1[0] + p1[1] + p1[2] + p1[3], b = p2[0] + p2[1] + p2[2] + p2[3];
    const uint64_t var = ((uint64_t)b * 4096 - (uint64_t)a * a) >> 24;
    const uint32_t v1 = (uint32_t)var + 1;
    const int l = 31 - __builtin_clz(v1), e = 16 * l + (int)(((v1 << 4) >> l) & 15);
    act[blockIdx.x] = e;
    atomicAdd(sum, e);
  }
}
__global__ __launch_bounds__(256) void k_vaq_apply(EncFrame f, int vaq, const int *act, const int *sum, int nctu)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nctu) return;
  const int mean = (int)(((long long)*sum + nctu / 2) / nctu);
  const int d = vaq * (act[i] - mean), delta = d >= 0 ? d / 48 : -((-d) / 48);
  int8_t *qt = const_cast<int8_t *>(f.ctu_qt);
  qt[i] = (int8_t)clip3(0, 51, f.qp + clip3(-12, 12, (int)qt[i] + delta));      // qt[i] holds the (clamped) ROI delta on entry
}

// =============================================================================================
// launch wrappers
// =============================================================================================
void launch_pad_input(const uint8_t *in, int w, int h, uint8_t *dy, uint8_t *du, uint8_t *dv, int cw, int ch, hipStream_t st)
{
  dim3 g((cw / 16 + 63) / 64, ch * 2 / 16);
  hipLaunchKernelGGL(k_pad_input, g, dim3(64, 4), 0, st, in, w, h, dy, du, dv, cw, ch);
}
void launch_me(const EncFrame &f, hipStream_t st)
{
  const int W = 2 * f.range + 1, items = ((W + 3) / 4) * ((W + 1) / 2);          // quads x pairs; R = 16: 153 items -> 192 threads
  const int threads = items >= 256 ? 256 : ((items + 63) / 64) * 64;
  hipLaunchKernelGGL(k_me, dim3(f.cw / 32, band_rows(f) * 2), dim3(threads), 0, st, f);
}
void launch_inter_recon(const EncFrame &f, hipStream_t st)
{
  const dim3 grid(f.cw / 32, band_rows(f) * 2);
  if (f.rc) {                                  // rate control v2: the groups of CTU rows inside the one launch
    const int k = ((f.rdoq || f.signhide) ? 2 : 0) | (f.subme > 0 ? 1 : 0);
    if (k == 3) hipLaunchKernelGGL((k_inter_recon<false, true, true, true>), grid, dim3(256), 0, st, f);
    else if (k == 2) hipLaunchKernelGGL((k_inter_recon<false, false, true, true>), grid, dim3(256), 0, st, f);
    else if (k == 1) hipLaunchKernelGGL((k_inter_recon<false, true, false, true>), grid, dim3(256), 0, st, f);
    else hipLaunchKernelGGL((k_inter_recon<false, false, false, true>), grid, dim3(256), 0, st, f);
  } else if (f.rdoq || f.signhide) {
    if (f.subme > 0) hipLaunchKernelGGL((k_inter_recon<false, true, true>), grid, dim3(256), 0, st, f);
    else hipLaunchKernelGGL((k_inter_recon<false, false, true>), grid, dim3(256), 0, st, f);
  } else if (f.lossless) {
    if (f.subme > 0) hipLaunchKernelGGL((k_inter_recon<false, true, false, false, true>), grid, dim3(256), 0, st, f);
    else hipLaunchKernelGGL((k_inter_recon<false, false, false, false, true>), grid, dim3(256), 0, st, f);
  } else if (f.subme > 0) hipLaunchKernelGGL((k_inter_recon<false, true>), grid, dim3(256), 0, st, f);
  else hipLaunchKernelGGL((k_inter_recon<false, false>), grid, dim3(256), 0, st, f);     // integer vectors only
}
void launch_inter_signal(const EncFrame &f, hipStream_t st)
{
  const int n = (f.cw / 8) * (band_rows(f) * 8);
  hipLaunchKernelGGL(k_inter_signal, dim3((n + 1023) / 1024), dim3(1024), 0, st, f);
}
void launch_intra_analyse(const EncFrame &f, hipStream_t st)
{
  // An intra picture's mode search is the one throughput-bound kernel of the pipeline: 2 040 independent workgroups at 1080p, all of them resident at once if
  // nothing stops them -- eight waves on every SIMD of the chip, i.e. EVERY wave slot, for the length of the launch, and whatever else is queued beside it
  // (the P pictures in front of an IDR, the other pictures' chains of an all-intra stream) cannot start a workgroup until one of the search's retires.  Dynamic
  // LDS it does not use caps it at kAnalysePerCu workgroups per compute unit (160 KB of LDS each), which leaves the other slots to everybody else.
  // Measured (profiles/r05_analyse_cap.txt; 4K, pipelined, worst launch of 504): k_me 948 -> 135 us, k_tok_compact 881 -> 59, k_inter_signal 540 -> 280 with a cap of 4;
  // the search itself 156 -> 173 us at 1080p (197 at 3); the encoder alone on an all-intra stream 2 450 -> 2 750 frames/s.  KVAZZUP_AMD_ANALYSE_PER_CU=0: no cap.
  static const int per_cu = getenv("KVAZZUP_AMD_ANALYSE_PER_CU") ? atoi(getenv("KVAZZUP_AMD_ANALYSE_PER_CU")) : 4;
  const size_t pad = per_cu > 1 && per_cu < 8 && !f.analyse_alone ? (size_t)(160 * 1024 / (per_cu + 1) + 1024 - 9264) & ~(size_t)255 : 0;
  if (f.is_intra) hipLaunchKernelGGL(k_intra_analyse<false>, dim3(f.cw / 32, band_rows(f) * 2), dim3(256), pad, st, f);
  else hipLaunchKernelGGL(k_intra_analyse<true>, dim3(512), dim3(1024), 0, st, f);       // intra-in-P, behind k_me: 512 workgroups (two per compute unit) share the candidate list, a quarter of a listed block at a time
}
void launch_intra_recon(const EncFrame &f, hipStream_t st)
{
  // as many workgroups as three anti-diagonals of the CTU wavefront hold, in three planes (k_intra_recon: tickets)
  static const int diags_env = getenv("KVAZZUP_AMD_INTRA_DIAGS") ? atoi(getenv("KVAZZUP_AMD_INTRA_DIAGS")) : -1;      // (measurement aid; 0: a workgroup per (CTU, plane))
  const int diags = diags_env >= 0 ? diags_env : (f.chain_diags ? f.chain_diags : 3);
  const int wc = f.cw / 64, nr = band_rows(f), diag = nr < (wc + 1) / 2 ? nr : (wc + 1) / 2, want = diags > 0 ? 3 * diags * diag + 32 : 3 * wc * nr;
  const dim3 grid(want < 3 * wc * nr ? want : 3 * wc * nr), block(64 * KVZ_INTRA_WAVES);
  const bool adj = f.rdoq || f.signhide;
  if (!f.is_intra) {                                                                                            // intra-in-P, behind k_inter_recon
    // (a workgroup per (CTU, plane) here: nearly all of them find no intra unit, and a workgroup that checks nine tickets in turn pays nine memory round trips)
    const dim3 all(3 * wc * nr);
    if (f.lossless) { hipLaunchKernelGGL((k_intra_recon<false, true, false, true>), all, block, 0, st, f); return; }
    if (f.scaling) { if (adj) hipLaunchKernelGGL((k_intra_recon<true, true, true>), all, block, 0, st, f); else hipLaunchKernelGGL((k_intra_recon<false, true, true>), all, block, 0, st, f); }
    else if (adj) hipLaunchKernelGGL((k_intra_recon<true, true>), all, block, 0, st, f); else hipLaunchKernelGGL((k_intra_recon<false, true>), all, block, 0, st, f);
    return;
  }
  if (f.lossless) hipLaunchKernelGGL((k_intra_recon<false, false, false, true>), grid, block, 0, st, f);
  else if (f.scaling) { if (adj) hipLaunchKernelGGL((k_intra_recon<true, false, true>), grid, block, 0, st, f); else hipLaunchKernelGGL((k_intra_recon<false, false, true>), grid, block, 0, st, f); }
  else if (adj) hipLaunchKernelGGL((k_intra_recon<true, false>), grid, block, 0, st, f);
  else hipLaunchKernelGGL((k_intra_recon<false, false>), grid, block, 0, st, f);
}
void launch_qp_resolve(const EncFrame &f, hipStream_t st)
{
  if (!f.ctu_qy) return;
  const int n = (f.cw / 64) * band_rows(f);
  if (f.wpp && f.cw / 64 <= 256) { hipLaunchKernelGGL(k_qp_rows, dim3(band_rows(f)), dim3(256), 0, st, f); return; }
  hipLaunchKernelGGL(k_qp_first, dim3(n), dim3(64), 0, st, f);
  hipLaunchKernelGGL(k_qp_chain, dim3((n + 255) / 256), dim3(256), 0, st, f);
}
void launch_deblock_v(const EncFrame &f, hipStream_t st)
{
  int nv = ((f.cw >> 3) - 1) * (band_rows(f) * 16);
  hipLaunchKernelGGL(k_deblock_v, dim3((nv + 255) / 256), dim3(256), 0, st, f);
}
void launch_deblock_h(const EncFrame &f, hipStream_t st, int part)
{
  const int ylo = imax(8, f.row0 * 64), yhi = imin(f.ch - 8, (f.row0 + band_rows(f)) * 64);
  int nh = (f.cw >> 2) * ((yhi - ylo) / 8 + 1);
  hipLaunchKernelGGL(k_deblock_h, dim3((nh + 255) / 256), dim3(256), 0, st, f, part);
}
void launch_deblock(const EncFrame &f, hipStream_t st)
{
  if (f.nrows > 0) { launch_deblock_v(f, st); launch_deblock_h(f, st, 0); return; }         // band of a tile-row split: two passes
  hipLaunchKernelGGL(k_deblock_tile, dim3((f.cw / 64) * (f.ch / 64)), dim3(256), 0, st, f);
}
void launch_vaq(const EncFrame &f, int vaq, int *act, int *sum, hipStream_t st)
{
  const int nctu = (f.cw / 64) * (f.ch / 64);
  hipMemsetAsync(sum, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_vaq_stats, dim3(nctu), dim3(256), 0, st, f, act, sum);
  hipLaunchKernelGGL(k_vaq_apply, dim3((nctu + 255) / 256), dim3(256), 0, st, f, vaq, (const int *)act, (const int *)sum, nctu);
}
void launch_sao(const EncFrame &f, hipStream_t st) { hipLaunchKernelGGL(k_sao<false>, dim3((f.cw / 64) * (f.ch / 64)), dim3(KVZ_SAO_THREADS), 0, st, f); }
void launch_tokenize(const EncFrame &f, hipStream_t st)
{
  // One wave per unit and colour component keeps the longest wave short: the kernel lasts as long as its slowest wave.  (One wave per unit that takes
  // the three components in turn was the faster form at 2160p while a wave that finds nothing to do cost what it did in round 1 -- 76 vs 137 us; with
  // today's early exit it is the slower one there as well: 66 vs 55 us per launch at 2160p, 51 vs 31 us at 1080p, a luma wave + a chroma wave per unit 60 /
  // 35 us -- and is kept for pictures beyond that, where nothing has been measured.)
  const int units = (f.cw / 16) * band_rows(f) * 4;
  // Measured (profiles/r06_tok_list_ab.txt, isolated): 2160p 49.1 against 56.9 us -- there the grid form is bound by the 130 000 waves it starts --, 1080p 27.4
  // against 24.1 us: there both forms last as long as their longest wave, and the list form's is the longer one (three waves per SIMD's worth of registers).
  // So: the list from 16 384 units on.  KVAZZUP_AMD_TOK_LIST=1 forces it everywhere (tests), =0 switches it off (A/B).
  static const int list_mode = [] { const char *e = getenv("KVAZZUP_AMD_TOK_LIST"); return e ? atoi(e) : -1; }();
  if (!f.is_intra && f.tok_list && f.nrows == 0 && units < 65536 && (list_mode > 0 || (list_mode < 0 && units > 16384))) {
    // the list form: as many waves as a picture of this size usually has pairs to say something about (half the units; a busier picture's waves take several)
    hipLaunchKernelGGL((k_tokenize<false, 1, true, true, true>), dim3(units / 2 < 256 ? 256 : units / 2), dim3(64), 0, st, f);
    return;
  }
  if (units >= 65536) hipLaunchKernelGGL((k_tokenize<true, 1>), dim3(f.cw / 16, band_rows(f) * 4), dim3(64), 0, st, f);
  else if (units > 16384) hipLaunchKernelGGL((k_tokenize<false, 1, false, false>), dim3(f.cw / 16, band_rows(f) * 4, 3), dim3(64), 0, st, f);     // (z: luma + headers, Cb, Cr)
  else hipLaunchKernelGGL((k_tokenize<false, 1, true>), dim3(f.cw / 16, band_rows(f) * 4, 4), dim3(64), 0, st, f);     // (z: luma, Cb, Cr, headers; tok_cursor is zero: the previous picture's k_tok_compact left it so)
}
void launch_tok_compact(const EncFrame &f, hipStream_t st)
{
  hipLaunchKernelGGL(k_tok_compact, dim3((f.cw / 64) * band_rows(f)), dim3(256), 0, st, f);
}

}  // namespace kvzx
