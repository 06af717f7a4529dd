// kvazzup_amd/csrc/kernel_common.h -- device-side building blocks shared by the encoder kernels (enc_kernels.hip) and the
// decoder kernels (dec_kernels.hip): XCD-aware workgroup order, wave reductions, inter-workgroup progress counters, the
// int16 dot2 / int8 MFMA transform stages and the intra sample predictors.  gfx950 only; included by .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include "hevc_core.h"

namespace kvzx {

struct alignas(4) kv_u32x4 { uint32_t x, y, z, w; };          // 16 bytes with dword alignment: one global_load/store_dwordx4

// Workgroups are observed to land on the eight XCDs round-robin (workgroup b -> XCD b % 8), each XCD with its own L2.  A raster
// of small blocks dealt out that way makes every XCD fetch every 128-byte line its neighbours also fetch (a 32-sample block
// row is a quarter of a line).  This permutation of the linear workgroup id gives each XCD one contiguous run of the raster
// instead.  Speed only: any placement computes the same result.
__device__ __forceinline__ int xcd_contiguous(int lin, int total)
{
  const int q = total >> 3, r = total & 7, xcd = lin & 7;
  return xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
}
// the same for a two-dimensional grid of blocks: (bx, by) of this workgroup after the permutation
__device__ __forceinline__ void xcd_block_2d(int &bx, int &by)
{
  const int gx = (int)gridDim.x, lin = xcd_contiguous((int)(blockIdx.y * gridDim.x + blockIdx.x), gx * (int)gridDim.y);
  by = lin / gx; bx = lin - by * gx;
}

// sum over the 64 lanes of a wave (every lane gets it); all lanes must be active
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);   // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);   // row_mirror: every lane holds its row-of-16 sum
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}



// ---------------------------------------------------------------------------------------------
// inter-workgroup progress counters (cdna_hip_programming.md section 6, Guideline 16):
// producer: stores -> __syncthreads -> lane 0: release fence + s_waitcnt + relaxed agent store
// consumer: lane 0 polls relaxed, then ONE acquire fence, then __syncthreads
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish_progress(uint32_t *ctr, uint32_t value)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void wait_progress(const uint32_t *ctr, uint32_t at_least, uint32_t *err)
{
  if (threadIdx.x == 0) {
    uint32_t spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < at_least) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 26)) { atomicOr(err, 1u); break; }       // bounded spin: never hang the GPU
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}


// =============================================================================================
// Transforms.  One primitive, P(X, T)[j][i] = sum_m X[i][m] * T[j][m] (rows of X times rows of T, the
// result stored transposed), run four times per block -- forward rows, forward columns, inverse
// columns, inverse rows -- on int16 data with v_dot2_i32_i16 (every intermediate of the 8-bit HEVC
// transforms fits 16 bits).  A lane owns a PAIR of rows of X and OPL outputs of each, so the matrix
// rows it reads serve both rows and every LDS store is a packed pair.  Quantisation + dequantisation
// are the epilogue of the second forward stage, reconstruction the epilogue of the last inverse one.
// Matrices live in LDS as int16: M[0] = M_n[j][m] (n-point DCT = rows of kDct32 subsampled),
// M[1] = its transpose, for n = 4, 8, 16, 32 at matrix_offset(log2 n).
// =============================================================================================
typedef short kv_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_i16(uint32_t a, uint32_t b, int c)
{
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(kv_short2, a), __builtin_bit_cast(kv_short2, b), c, false);
}
__device__ __forceinline__ uint32_t pack_i16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int matrix_offset(int l2) { return l2 == 2 ? 0 : (l2 == 3 ? 16 : (l2 == 4 ? 80 : 336)); }
#define KV_MATRIX_ENTRIES 1376      // 4-, 8-, 16-, 32-point DCT at matrix_offset(log2 n), then the 4-point DST at KV_DST_OFFSET
#define KV_DST_OFFSET 1360
// Both matrix sets, the 32-point matrix as int8 (MFMA operand) with its row sums: built at compile time, so a kernel
// fetches what it needs with one 16-byte load per thread instead of recomputing entries from kDct32.
struct alignas(16) XfTables {
  int16_t M[2][KV_MATRIX_ENTRIES];
  int8_t M8[2][32 * 32];
  int rowsum[2][32];
};
constexpr XfTables make_xf_tables()
{
  XfTables t{};
  for (int i = 0; i < 16; i++) { t.M[0][KV_DST_OFFSET + i] = kDst4[i >> 2][i & 3]; t.M[1][KV_DST_OFFSET + i] = kDst4[i & 3][i >> 2]; }
  for (int i = 0; i < KV_DST_OFFSET; i++) {
    const int l2 = i < 16 ? 2 : (i < 80 ? 3 : (i < 336 ? 4 : 5)), first = l2 == 2 ? 0 : (l2 == 3 ? 16 : (l2 == 4 ? 80 : 336));
    const int k = i - first, j = k >> l2, m = k & ((1 << l2) - 1);
    t.M[0][i] = kDct32[j << (5 - l2)][m];
    t.M[1][i] = kDct32[m << (5 - l2)][j];
  }
  for (int j = 0; j < 32; j++) {
    int a0 = 0, a1 = 0;
    for (int m = 0; m < 32; m++) {
      t.M8[0][j * 32 + m] = kDct32[j][m]; t.M8[1][j * 32 + m] = kDct32[m][j];
      a0 += kDct32[j][m]; a1 += kDct32[m][j];
    }
    t.rowsum[0][j] = a0; t.rowsum[1][j] = a1;
  }
  return t;
}
static __device__ const XfTables g_xf = make_xf_tables();

// fills entries [first, first + count) of both matrix sets (all sizes: first = 0, count = KV_MATRIX_ENTRIES); both
// multiples of 8
__device__ __forceinline__ void load_matrices(int16_t (*M)[KV_MATRIX_ENTRIES], int first, int count, int tid, int nthreads)
{
  const int per = count >> 3;
  for (int i = tid; i < 2 * per; i += nthreads) {
    const int t = i >= per, k = first + ((i - t * per) << 3);
    *(uint4 *)&M[t][k] = *(const uint4 *)&g_xf.M[t][k];
  }
}

// lanes that share one n x n block: (n / 2) row pairs x (n / OPL) output groups
template <int L2, int OPL> struct XF { static constexpr int N = 1 << L2, G = N / OPL, LANES = (N / 2) * G; };


// The head of a picture's chain (rc_kernels.hip k_picture_begin; a P picture without VAQ: the first workgroup of k_me -- a launch of its own cost the chain
// its 6 us and the gap in front of it): rate control v2's counters back to zero and its ratio updated from the access unit sized meanwhile, every CTU's target QP.
// Called by all `nthreads` threads of ONE workgroup.
__device__ __forceinline__ void picture_begin_body(RcState *rc, uint32_t bits3, int slot3, int have3, int8_t *ctu_qt, const int8_t *roi, int nctu, int qp, int vaq, int tid, int nthreads)
{
  if (ctu_qt) for (int i = tid; i < nctu; i += nthreads) { const int d = roi ? roi[i] : 0; ctu_qt[i] = (int8_t)(vaq ? d : clip3(0, 51, qp + d)); }
  if (!rc || tid) return;
  rc->cost_sofar = 0; rc->decided = 0;
  for (int g = 0; g < 8; g++) rc->acc[g * KVZ_RC_ACC_STRIDE] = 0;
  if (!have3 || !rc->cost_valid[slot3]) return;
  rc->cost_valid[slot3] = 0;
  const uint32_t c = rc->cost[slot3];
  unsigned long long r = ((unsigned long long)bits3 << 8) / (c ? c : 1u);
  if (r > (1u << 20)) r = 1u << 20;
  if (r < 1) r = 1;
  rc->ratio_q8 = rc->ratio_valid ? (uint32_t)((3ull * rc->ratio_q8 + r + 2) >> 2) : (uint32_t)r;
  rc->ratio_valid = 1;
}

// ---------------------------------------------------------------------------------------------
// Fence-free hand-off between workgroups inside a launch (cdna_hip_programming.md section 6, Guideline 16, recipe R1): the
// producer stores its payload WRITE-THROUGH (agent-scope relaxed atomic stores = `sc1`), every storing wave drains its stores,
// one lane stores the flag; the consumer polls the flag relaxed and reads the payload with agent-scope relaxed loads (`sc1`:
// served past the CU's L1).  No agent-scope release / acquire fence (1.7 .. 7 us each on gfx950) anywhere.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_wt_u32(void *p, uint32_t v) { __hip_atomic_store((uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt_u64(void *p, uint64_t v) { __hip_atomic_store((uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt_u8(void *p, uint32_t v) { __hip_atomic_store((uint8_t *)p, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_l2_u32(const void *p) { return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_l2_u64(const void *p) { return __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_l2_u8(const void *p) { return __hip_atomic_load((const uint8_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every thread of the workgroup calls it after its write-through stores
__device__ __forceinline__ void publish_wt(uint32_t *ctr, uint32_t value)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until *ctr >= need; `seen` = the last value observed (workgroup-uniform), so that satisfied waits cost nothing.
// bcast: one LDS word.  The payload is then read with ld_l2_*.
__device__ __forceinline__ uint32_t wait_wt(const uint32_t *ctr, uint32_t need, uint32_t seen, uint32_t *bcast, uint32_t *err)
{
  if (seen >= need) return seen;
  if (threadIdx.x == 0) {
    uint32_t spins = 0, v;
    while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
      if (++spins < 16) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(16);      // (every poll is a trip to memory: back off once it is clear that the wait is long)
      if (spins > (1u << 22)) { atomicOr(err, 1u); v = 64; break; }         // bounded spin: never hang the GPU
    }
    *bcast = v;
  }
  __syncthreads();
  const uint32_t v = *bcast;
  __syncthreads();                                       // (the word is reused by the next wait)
  return v;
}
// rows of an n x n block of bytes from LDS (pitch lp) to the picture (pitch gp), write-through; n = 4, 8, 16, 32; x0 a multiple of n
__device__ __forceinline__ void store_block_wt(uint8_t *g, int gp, const uint8_t *l, int lp, int n, int tid, int nthreads)
{
  if (n == 4) { if (tid < 4) st_wt_u32(g + (size_t)tid * gp, *(const uint32_t *)(l + tid * lp)); return; }
  const int per = n >> 3;                                // 8-byte pieces per row
  for (int i = tid; i < n * per; i += nthreads) { const int y = i / per, x = (i - y * per) * 8; st_wt_u64(g + (size_t)y * gp + x, *(const uint64_t *)(l + y * lp + x)); }
}

// Intra wavefront bookkeeping shared by the encoder's k_intra_recon and the decoder's k_dec_intra: progress of a (CTU, plane)
// workgroup is counted in 8x8 luma units of the CTU in z-order -- every block that starts in a unit below the counter is final
// in the picture -- and published at the values neighbours wait for.
__device__ __forceinline__ int kv_zunit8(int xi, int yi)
{
  int z = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) z |= ((xi >> b) & 1) << (2 * b) | ((yi >> b) & 1) << (2 * b + 1);
  return z;
}
__device__ __forceinline__ int kv_intra_milestone(int z) { return z >= 64 ? 7 : (z >= 60 ? 6 : (z >= 56 ? 5 : (z >= 48 ? 4 : (z >= 44 ? 3 : (z >= 32 ? 2 : (z >= 24 ? 1 : 0)))))); }
// 8x8 units of the bottom row (xi, 7) / right column (7, yi) of a CTU that are final at progress p, counted from 0
__device__ __forceinline__ int kv_units_bottom(uint32_t p) { return p >= 64 ? 8 : (p >= 60 ? 6 : (p >= 48 ? 4 : (p >= 44 ? 2 : 0))); }
__device__ __forceinline__ int kv_units_right(uint32_t p) { return p >= 64 ? 8 : (p >= 62 ? 7 : (p >= 56 ? 6 : (p >= 54 ? 5 : (p >= 32 ? 4 : (p >= 30 ? 3 : (p >= 24 ? 2 : (p >= 22 ? 1 : 0))))))); }

// The borders of a CTU picture in LDS (row 0 / column 15 of the padded layout, pitch lp) are filled piecewise from the neighbouring
// CTUs' samples in the picture, as far as those CTUs' progress allows.
// Pictures that mix intra and inter blocks (P pictures: the inter blocks are final before the intra kernel starts): a block waits for a
// neighbouring CTU only as far as the edge units it reads ARE intra units -- masks over the eight 8x8 units of the neighbour's right
// column (il) / bottom row (iu, iur) and its corner unit (iul); all ones in an intra picture.  kv_edge_need: the progress value at which
// the intra units among edge units 0 .. last are final, 0 when there is none (nothing to wait for).
__device__ __forceinline__ uint32_t kv_edge_need(uint32_t mask, int last, bool column)
{
  mask &= (2u << last) - 1u;
  if (!mask) return 0u;
  const int top = 31 - __builtin_clz(mask);
  return (uint32_t)(column ? kv_zunit8(7, top) : kv_zunit8(top, 7)) + 1u;
}
struct IntraBorders {
  const uint32_t *pl, *pu, *pur, *pul;       // progress counters of the left / upper / upper-right / upper-left CTU (same plane)
  bool nb_left, nb_up, nb_ur, nb_ul;         // which of them exist (inside the picture, same tile)
  uint32_t il = 0xffu, iu = 0xffu, iur = 0xffu, iul = 1u;
  const uint8_t *ecol_left = nullptr;        // the left CTU's entry of the edge-column array (IB_EDGE_R), or null
  uint32_t seen_l, seen_u, seen_ur, seen_ul;
  int top_loaded, left_loaded; bool corner_loaded;
};
// one round trip for all four counters (threads 0..3), then the borders everything seen so far allows
__device__ __forceinline__ void borders_begin(IntraBorders &b, uint32_t *bc4)
{
  if (threadIdx.x < 4) {
    const uint32_t *p = threadIdx.x == 0 ? b.pl : (threadIdx.x == 1 ? b.pu : (threadIdx.x == 2 ? b.pur : b.pul));
    const bool on = threadIdx.x == 0 ? b.nb_left : (threadIdx.x == 1 ? b.nb_up : (threadIdx.x == 2 ? b.nb_ur : b.nb_ul));
    bc4[threadIdx.x] = on ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  __syncthreads();
  b.seen_l = bc4[0]; b.seen_u = bc4[1]; b.seen_ur = bc4[2]; b.seen_ul = bc4[3];
  __syncthreads();
  b.top_loaded = 0; b.left_loaded = 0; b.corner_loaded = false;
}
// make sure the borders the block at (rx, ry), size n, reads are in LDS.  S: CTU size in this plane's samples, sh: 1 for chroma,
// lim_w: samples of the row above that exist (picture width, upper-right CTU); plane / gp: the picture plane and its pitch;
// (cx, cy): CTU coordinates in units of S.  Ends with a barrier when anything was loaded.
__device__ __forceinline__ void borders_need(IntraBorders &b, uint8_t *pic, int lp, const uint8_t *plane, int gp, int cx, int cy, int S, int sh,
                                             int lim_w, int rx, int ry, int n, uint32_t *bcast, uint32_t *err, int tid, int nthreads, int nl2 = -1, int nt2 = -1)
{
  // nl2 / nt2: rows of the left border / samples of the row above the block's mode reads from (rx, ry) on: 2 n, or n when the mode does not use the
  // below-left / above-right samples (hevc_core.h intra_uses_below_left / intra_uses_above_right)
  if (nl2 < 0) nl2 = 2 * n;
  if (nt2 < 0) nt2 = 2 * n;
  bool loaded = false;
  if (rx == 0 && b.nb_left) {
    const int need = imin(S, ry + nl2);
    if (need > b.left_loaded) {
      b.seen_l = wait_wt(b.pl, kv_edge_need(b.il, ((need - 1) << sh) >> 3, true), b.seen_l, bcast, err);
      const int upto = imax(need, imin(S, kv_units_right(b.seen_l) * (8 >> sh)));
      for (int i = b.left_loaded + tid; i < upto; i += nthreads)
        pic[(i + 1) * lp + 15] = (uint8_t)((b.ecol_left && ((b.il >> ((i << sh) >> 3)) & 1u)) ? ld_l2_u8(b.ecol_left + i) : ld_l2_u8(plane + (size_t)(cy * S + i) * gp + cx * S - 1));
      b.left_loaded = upto; loaded = true;
    }
  }
  if (ry == 0 && b.nb_up) {
    const int lim = imin(lim_w, b.nb_ur ? 2 * S : S), need = imin(lim, rx + nt2);
    if (need > b.top_loaded) {
      int upto = need;
      if (b.top_loaded < S) {
        b.seen_u = wait_wt(b.pu, kv_edge_need(b.iu, ((imin(S, need) - 1) << sh) >> 3, false), b.seen_u, bcast, err);
        upto = imax(upto, imin(imin(S, lim), kv_units_bottom(b.seen_u) * (8 >> sh)));
      }
      if (need > S) {
        b.seen_ur = wait_wt(b.pur, kv_edge_need(b.iur, ((need - S - 1) << sh) >> 3, false), b.seen_ur, bcast, err);
        upto = imax(upto, imin(lim, S + kv_units_bottom(b.seen_ur) * (8 >> sh)));
      }
      // (both ends are multiples of 4: block sizes, CTU sizes and picture widths are)
      for (int i = b.top_loaded + 4 * tid; i < upto; i += 4 * nthreads) *(uint32_t *)&pic[16 + i] = ld_l2_u32(plane + (size_t)(cy * S - 1) * gp + cx * S + i);
      b.top_loaded = upto; loaded = true;
    }
  }
  if (rx == 0 && ry == 0 && b.nb_ul && !b.corner_loaded) {
    b.seen_ul = wait_wt(b.pul, b.iul ? 64u : 0u, b.seen_ul, bcast, err);
    if (tid == 0) pic[15] = (uint8_t)ld_l2_u8(plane + (size_t)(cy * S - 1) * gp + cx * S - 1);
    b.corner_loaded = true; loaded = true;
  }
  if (loaded) __syncthreads();
}

// LDS traffic inside one wave needs no barrier instruction (a wave's LDS instructions execute in order); this keeps the compiler
// from moving accesses across the point where lanes exchange data
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Several waves per (CTU, plane): blocks whose neighbours are final run side by side.  Inside a CTU a block depends only on the
// blocks its reference samples lie in; in z-order that leaves independent chains -- the quadrant right of a finished quadrant
// and the one below it, at every level -- so a CTU of sixty-four 8x8 blocks has a critical path of 36 blocks, not 64.  Waves take
// blocks from the list in order (so whatever a wave waits for has been taken by another wave), wait on a mask of finished 8x8
// units in LDS.  What the NEIGHBOURING CTUs read -- the CTU's right column and bottom row -- leaves as self-validating words
// (IntraNeighbours), so there is no progress counter, no acknowledgement of stores and nothing to publish (rounds 2 and 3 had all three).
// ---------------------------------------------------------------------------------------------
// bounds of the chain's spins (they only ever end a launch that would otherwise hang; -DKVZ_POLL_LIMIT_LOG2=12 -DKVZ_WAIT_LIMIT_LOG2=12 makes a broken chain fail fast)
#ifndef KVZ_POLL_LIMIT_LOG2
#define KVZ_POLL_LIMIT_LOG2 21
#endif
#ifndef KVZ_WAIT_LIMIT_LOG2
#define KVZ_WAIT_LIMIT_LOG2 24
#endif
struct IntraChain {
  uint32_t done[2];            // bit z: the 8x8 luma unit z (z-order) is final in the CTU picture in LDS
  uint32_t claim;              // next list entry to be handed to a wave
  int left_loaded, top_loaded, corner_loaded;       // how much of the borders has been copied into the CTU picture
};
struct IntraNeighbours {
  bool nb_left, nb_up, nb_ur, nb_ul;         // which of them exist (inside the picture, same tile)
  uint32_t il = 0xffu, iu = 0xffu, iur = 0xffu, iul = 1u;      // (IntraBorders: the neighbours' edge units that are intra units)
  // The left CTU's entry of the edge-column array (IB_EDGE_R), or null: its right column is read from the picture.  Round 4: the entries are SELF-VALIDATING
  // words, sample | launch generation << 8 -- the right neighbour polls the words themselves until they carry this launch's generation.  The producer
  // neither waits for these stores nor publishes anything for them, and the consumer's poll IS its load: the hop of the CTU wavefront loses the
  // producer's store drain and one memory round trip of the consumer.
  const uint32_t *ecol_left = nullptr;
  uint32_t gen = 0;
  // ... and the same for the upper neighbours' BOTTOM ROWS: per CTU S / 4 words {four samples, generation} -- the upper, upper-right and upper-left CTU's
  // entries (the corner is the last sample of the upper-left one), or null: read from the picture
  const unsigned long long *erow_up = nullptr, *erow_ur = nullptr, *erow_ul = nullptr;
};
__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The next entry of the workgroup's claim counter, for the whole wave.  RULE for every loop a wave runs as a whole (claim loops, polls): the values that steer it
// -- the claimed index AND the bound it is compared with -- must be uniform BY CONSTRUCTION (readfirstlane / kernel arguments), never "uniform because every
// lane loaded the same LDS word".  The compiler takes a compare against a vector register for a divergent exit and rebuilds the loop with execution masks; with
// the `if (lane == 0) atomic` of the next round folded into that rebuilt loop it produced, for k_dec_intra's loop written as `for (;;)`, an inner loop that
// lane 0 LEAVES to do the atomic while lanes 1..63 go round again with the initial value 0 in their copy of the index -- readfirstlane then reads lane 1's 0,
// the wave codes item 0 for ever (profiles/r05_claim_loop_isa.txt has both listings).  With a scalar bound the exit is s_cmp + s_cbranch_scc and the loop stays
// a plain scalar loop in either spelling.
__device__ __forceinline__ int chain_claim(IntraChain &ch, int lane)
{
  __builtin_amdgcn_wave_barrier();                        // (convergent, emits nothing: the block that tests `lane == 0` cannot be copied into the loop's latch)
  int k = 0;
  if (lane == 0) k = (int)atomicAdd(&ch.claim, 1u);
  return __builtin_amdgcn_readfirstlane(k);
}
__device__ __forceinline__ int wave_uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

// thread 0 of the workgroup; followed by a barrier in the caller
__device__ __forceinline__ void chain_init(IntraChain &ch, uint32_t done0, uint32_t done1)
{
  if (threadIdx.x == 0) { ch.done[0] = done0; ch.done[1] = done1; ch.claim = 0; ch.left_loaded = 0; ch.top_loaded = 0; ch.corner_loaded = 0; }
}
// wave-level borders_need(): the borders the block at (rx, ry), size n, reads are in LDS when it returns.  Two waves may copy
// overlapping pieces (the same bytes); the *_loaded marks only ever grow.
// try_only (a measurement aid): ONE look -- when something is not there yet nothing is copied and false comes back
__device__ __forceinline__ bool borders_need_wave(IntraChain &ch, const IntraNeighbours &b, uint8_t *pic, int lp, const uint8_t *plane, int gp, int cx, int cy, int S, int sh,
                                                  int lim_w, int rx, int ry, int n, uint32_t *err, int lane, int nl2, int nt2, bool try_only = false)
{
  // (nl2 / nt2: as in borders_need)
  // Round 4: nothing is waited for through progress counters any more.  What a neighbouring CTU's INTRA units contribute arrives as self-validating words
  // (IntraNeighbours: the left CTU's right column one sample per word, the upper CTUs' bottom rows four samples per word, each with the launch's generation)
  // which are simply loaded again until they carry this launch's tag; what its inter units contribute (P pictures) is final in the picture.  All loads of
  // the left column, the top row and the corner are in flight together, and the poll is the load.
  int haveL = 0, uptoL = 0, haveT = 0, uptoT = 0;
  bool doC = false;
  (void)n;
  if (rx == 0 && b.nb_left) {
    const int need = imin(S, ry + nl2);
    haveL = __builtin_amdgcn_readfirstlane(lds_load(&ch.left_loaded));
    if (need > haveL) uptoL = need;
  }
  if (ry == 0 && (b.nb_up || b.nb_ur)) {                 // (decoder, several slices inside a tile: the block above-right may be the only one there is -- the row piece above is then copied with it, unused)
    const int lim = imin(lim_w, b.nb_ur ? 2 * S : S), need = imin(lim, rx + nt2);
    haveT = __builtin_amdgcn_readfirstlane(lds_load(&ch.top_loaded));
    if (need > haveT) uptoT = need;
  }
  if (rx == 0 && ry == 0 && b.nb_ul && !__builtin_amdgcn_readfirstlane(lds_load(&ch.corner_loaded))) doC = true;
  // (both ends of the row piece are multiples of 4: block sizes, CTU sizes and picture widths are)
  const int iL = haveL + lane, iT = haveT + 4 * lane;
  const bool rowL = iL < uptoL, tagL = rowL && b.ecol_left && ((b.il >> ((iL << sh) >> 3)) & 1u);
  const bool colT = iT < uptoT;
  bool tagT = false; const unsigned long long *wT = nullptr;
  if (colT) {
    if (iT < S) { tagT = b.erow_up && ((b.iu >> ((iT << sh) >> 3)) & 1u); wT = b.erow_up + (iT >> 2); }
    else { tagT = b.erow_ur && ((b.iur >> (((iT - S) << sh) >> 3)) & 1u); wT = b.erow_ur + ((iT - S) >> 2); }
  }
  const bool cornC = doC && lane == 0, tagC = cornC && b.erow_ul && b.iul;
  uint32_t vL = 0, vT = 0, vC = 0;
  unsigned long long qT = 0, qC = 0;
  if (rowL) vL = tagL ? ld_l2_u32(b.ecol_left + iL) : ld_l2_u8(plane + (size_t)(cy * S + iL) * gp + cx * S - 1);
  if (colT) { if (tagT) qT = ld_l2_u64(wT); else vT = ld_l2_u32(plane + (size_t)(cy * S - 1) * gp + cx * S + iT); }
  if (cornC) { if (tagC) qC = ld_l2_u64(b.erow_ul + (S >> 2) - 1); else vC = ld_l2_u8(plane + (size_t)(cy * S - 1) * gp + cx * S - 1); }
  {
    const unsigned long long gen = b.gen;
    uint32_t spins = 0;
    for (;;) {
      const bool pL = tagL && (vL >> 8) != b.gen, pT = tagT && (qT >> 32) != gen, pC = tagC && (qC >> 32) != gen;
      if (__ballot(pL || pT || pC) == 0) break;
      if (try_only) return false;
      if (++spins < 16) __builtin_amdgcn_s_sleep(1); else if (spins < 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
      if (spins > (1u << KVZ_POLL_LIMIT_LOG2)) { if (lane == 0) atomicOr(err, 1u); break; }           // bounded spin: never hang the GPU
      if (pL) vL = ld_l2_u32(b.ecol_left + iL);
      if (pT) qT = ld_l2_u64(wT);
      if (pC) qC = ld_l2_u64(b.erow_ul + (S >> 2) - 1);
    }
    if (tagL) vL &= 255u;
    if (tagT) vT = (uint32_t)qT;
    if (tagC) vC = (uint32_t)(qC >> 24) & 255u;              // (the upper-left CTU's bottom-right sample: byte 3 of its last word)
  }
  if (rowL) pic[(iL + 1) * lp + 15] = (uint8_t)vL;
  if (colT) *(uint32_t *)&pic[16 + iT] = vT;
  if (cornC) pic[15] = (uint8_t)vC;
  wave_sync();
  if (lane == 0) {
    if (uptoL > haveL) atomicMax(&ch.left_loaded, uptoL);
    if (uptoT > haveT) atomicMax(&ch.top_loaded, uptoT);
    if (doC) atomicMax(&ch.corner_loaded, 1);
  }
  wave_sync();
  return true;
}
// 8x8 luma units (bits in z-order) the block with its first unit at (ux, uy), su x su units, reads reference samples from inside its
// own CTU: the units left, below-left, above, above-right and above-left of it that precede it in z-order (inside a CTU:
// precedes = available)
// bl / tr: whether the block's MODE reads its below-left / above-right samples (hevc_core.h intra_uses_*): a unit nothing is read from is not waited for --
// e.g. the third 16x16 block of a CTU's top row then only waits for the second, not for the two blocks of the row below that precede it in z-order
__device__ __forceinline__ uint2 chain_dependencies(int ux, int uy, int su, bool bl = true, bool tr = true)
{
  const int z0 = kv_zunit8(ux, uy);
  uint32_t m0 = 0, m1 = 0;
  auto add = [&](int x, int y) {
    if (x < 0 || y < 0 || x > 7 || y > 7) return;
    const int z = kv_zunit8(x, y);
    if (z >= z0) return;
    if (z < 32) m0 |= 1u << z; else m1 |= 1u << (z - 32);
  };
  for (int i = -1; i < (bl ? 2 * su : su); i++) add(ux - 1, uy + i);
  for (int i = -1; i < (tr ? 2 * su : su); i++) add(ux + i, uy - 1);
  return make_uint2(m0, m1);
}
__device__ __forceinline__ uint2 chain_cover(int z0, int su)     // the su x su units starting at z0: a contiguous run in z-order
{
  const int n = su * su;
  const unsigned long long m = (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << z0;
  return make_uint2((uint32_t)m, (uint32_t)(m >> 32));
}
__device__ __forceinline__ void chain_wait_done(const IntraChain &ch, uint2 dep, uint32_t *err, int lane)
{
  uint32_t spins = 0;
  for (;;) {
    const uint32_t d0 = lds_load(&ch.done[0]), d1 = lds_load(&ch.done[1]);
    if ((d0 & dep.x) == dep.x && (d1 & dep.y) == dep.y) break;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << KVZ_WAIT_LIMIT_LOG2)) { if (lane == 0) atomicOr(err, 2u); break; }
  }
}
__device__ __forceinline__ void chain_mark_done(IntraChain &ch, uint2 cover, int lane)
{
  wave_sync();                                              // (the block's samples are in the CTU picture before the mark)
  if (lane == 0) { if (cover.x) atomicOr(&ch.done[0], cover.x); if (cover.y) atomicOr(&ch.done[1], cover.y); }
}
// thread layout of one block inside a workgroup of T threads (64: one wave, 256: four): OPL outputs per thread so that
// (n / 2) row pairs x (n / OPL) output groups fit
template <int L2, int T> struct XW {
  static constexpr int N = 1 << L2, OPL = (N * N / 2 + T - 1) / T < 1 ? 1 : (N * N / 2 + T - 1) / T, G = N / OPL, LANES = (N / 2) * G;
};

// raw sums of P for the lane's two rows (2rp, 2rp + 1) and its OPL outputs g * OPL + o
template <int L2, int OPL>
__device__ __forceinline__ void xf_sums(const int16_t *in, const int16_t *T, int rp, int g, int (&acc)[2][OPL])
{
  constexpr int N = 1 << L2, H = N / 2;
  const uint32_t *r0 = (const uint32_t *)(in + 2 * rp * N);
  uint32_t a0[H], a1[H];
#pragma unroll
  for (int m = 0; m < H; m++) { a0[m] = r0[m]; a1[m] = r0[H + m]; }
#pragma unroll
  for (int o = 0; o < OPL; o++) {
    const uint32_t *t = (const uint32_t *)(T + (g * OPL + o) * N);
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int m = 0; m < H; m++) { uint32_t tv = t[m]; s0 = dot2_i16(a0[m], tv, s0); s1 = dot2_i16(a1[m], tv, s1); }
    acc[0][o] = s0; acc[1][o] = s1;
  }
}
// plain stage: out[j][i] = clip16((sum + rnd) >> shift), stored as packed row pairs
template <int L2, int OPL>
__device__ __forceinline__ void xf_stage(const int16_t *in, int16_t *out, const int16_t *T, int shift, int rp, int g)
{
  constexpr int N = 1 << L2;
  const int rnd = 1 << (shift - 1);
  int acc[2][OPL];
  xf_sums<L2, OPL>(in, T, rp, g, acc);
#pragma unroll
  for (int o = 0; o < OPL; o++) {
    int v0 = clip3(-32768, 32767, (acc[0][o] + rnd) >> shift), v1 = clip3(-32768, 32767, (acc[1][o] + rnd) >> shift);
    ((uint32_t *)out)[((g * OPL + o) * N) / 2 + rp] = pack_i16(v0, v1);
  }
}


// ---- the 32x32 transform on the matrix cores --------------------------------------------------------------------
// One stage P(X, T)[j][i] = sum_m X[i][m] * T[j][m] of a 32x32 block is a 32x32x32 integer matrix product.  T fits
// int8 (|coefficient| <= 90); X is int16, so it is split into bytes: x = 256 * hi + (lo - 128) + 128 with hi = x >> 8
// and lo - 128 both in [-128, 127], giving
//   sum_m x T = 256 * (hi . T) + ((lo - 128) . T) + 128 * rowsum(T)        -- two MFMAs, exact in int32.
// Each of the four waves owns one 16x16 tile of the result (rows 16 * (w >> 1) .., columns 16 * (w & 1) ..) and uses
// v_mfma_i32_16x16x64_i8 with the upper half of K zero: lane l < 32 holds row (l & 15), k = (l >> 4) * 16 .. + 15 of
// its A tile (16 bytes) and the same k range of row (l & 15) of T for B; result: column l & 15, rows (l >> 4) * 4 + r.
typedef int kv_i32x4 __attribute__((ext_vector_type(4)));

// sums of the lane's four results: tile rows (lane >> 4) * 4 + r, tile column lane & 15
__device__ __forceinline__ void mfma_tile_sums(const int16_t *X, const int8_t *T8, const int *rowsum, int wave, int lane, int (&acc)[4])
{
  const int ti = (wave >> 1) * 16, tj = (wave & 1) * 16, kb = (lane >> 4) * 16;
  kv_i32x4 ahi = {0, 0, 0, 0}, alo = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (lane < 32) {
    const int16_t *xr = X + (ti + (lane & 15)) * 32 + kb;
    const uint4 x0 = *(const uint4 *)xr, x1 = *(const uint4 *)(xr + 8);
    const uint32_t d[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      ahi[q] = (int)__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07050301u);                  // high bytes of four int16
      alo[q] = (int)(__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x06040200u) ^ 0x80808080u);  // low bytes - 128
    }
    b = *(const kv_i32x4 *)(T8 + (tj + (lane & 15)) * 32 + kb);
  }
  kv_i32x4 chi = {0, 0, 0, 0}, clo = {0, 0, 0, 0};
  chi = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, b, chi, 0, 0, 0);
  clo = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, b, clo, 0, 0, 0);
  const int bias = 128 * rowsum[tj + (lane & 15)];
#pragma unroll
  for (int r = 0; r < 4; r++) acc[r] = 256 * chi[r] + clo[r] + bias;
}

// plain stage: out[j][i] = clip16((sum + rnd) >> shift); the lane's column j is a row of `out`, its four rows i one store
__device__ __forceinline__ void mfma_stage(const int16_t *X, int16_t *out, const int8_t *T8, const int *rowsum, int shift, int wave, int lane)
{
  int acc[4];
  mfma_tile_sums(X, T8, rowsum, wave, lane, acc);
  const int rnd = 1 << (shift - 1), j = (wave & 1) * 16 + (lane & 15), i0 = (wave >> 1) * 16 + (lane >> 4) * 4;
  int v[4];
#pragma unroll
  for (int e = 0; e < 4; e++) v[e] = clip3(-32768, 32767, (acc[e] + rnd) >> shift);
  *(uint2 *)&out[j * 32 + i0] = make_uint2(pack_i16(v[0], v[1]), pack_i16(v[2], v[3]));
}


// Intra sample prediction (8.4.4.2.4-6) from the reference array R (scan order, see IntraWaveLds); the same
// arithmetic as intra_pred_sample() of hevc_core.h, indexed for R.  s = +1 / -1 walks the main side.
template <int L2>
__device__ __forceinline__ int pred_planar(const uint8_t *R, int x, int y)
{
  constexpr int N = 1 << L2;
  return ((N - 1 - x) * R[2 * N - 1 - y] + (x + 1) * R[3 * N + 1] + (N - 1 - y) * R[2 * N + 1 + x] + (y + 1) * R[N - 1] + N) >> (L2 + 1);
}
template <int L2>
__device__ __forceinline__ int pred_dc(const uint8_t *R, bool edge, int dc, int x, int y)
{
  constexpr int N = 1 << L2;
  if (edge) {
    if (x == 0 && y == 0) return (R[2 * N - 1] + 2 * dc + R[2 * N + 1] + 2) >> 2;
    if (y == 0) return (R[2 * N + 1 + x] + 3 * dc + 2) >> 2;
    if (x == 0) return (R[2 * N - 1 - y] + 3 * dc + 2) >> 2;
  }
  return dc;
}
template <int L2>
__device__ __forceinline__ int pred_angular(const uint8_t *R, bool vert, bool edge, int angle, int inv, int x, int y)
{
  constexpr int N = 1 << L2;
  const int a = vert ? x : y, b = vert ? y : x, sgn = vert ? 1 : -1;        // a runs along the main reference side
  if (edge && a == 0) return clip8(R[2 * N + sgn] + ((R[2 * N - sgn * (1 + b)] - R[2 * N]) >> 1));
  const int t = (b + 1) * angle, k0 = a + (t >> 5) + 1, k1 = k0 + 1, fact = t & 31;
  const int i0 = k0 >= 0 ? k0 : -((k0 * inv + 128) >> 8), i1 = k1 >= 0 ? k1 : -((k1 * inv + 128) >> 8);
  return ((32 - fact) * R[2 * N + sgn * i0] + fact * R[2 * N + sgn * i1] + 16) >> 5;
}


// =============================================================================================
// One intra block per WAVE, transforms on the matrix cores.
//
// A block of n x n samples (n = 4, 8, 16) is handled by one wave with no workgroup barrier anywhere; lane (g, c) = (lane >> 4,
// lane & 15) owns the four samples x = 4g .. 4g + 3 of row y = c ("row owner"; lanes outside the block idle).  The four
// transform stages are 16x16x16 matrix products on v_mfma_f32_16x16x16_f16, exact because every operand is an integer of at
// most 11 bits (matrix entries |t| <= 90, residuals |r| <= 255, 16-bit intermediates split into a signed high byte and an
// unsigned low byte -> two products) and every sum stays below 2^24.  MFMA operand layouts (A[m][k]: lane holds m = c,
// k = 4g + r; B[k][n]: k = 4g + r, n = c; D[m][n]: m = 4g + r, n = c) chain the stages through registers:
//   forward rows      Y = X Tt      A = X (row owner)        B[k][n] = T[n][k]  (ta)      -> Y[y = 4g + r][j = c]
//   forward columns   C = T Y       A[m][k] = T[m][k] (ta)   B = Y                        -> C[u = 4g + r][j = c]
//   inverse columns   W = Tt C'     A[m][k] = T[k][m] (tb)   B = C'                       -> W[y = 4g + r][j = c]
//   inverse rows      X't = Tt Wt   A[m][k] = T[k][m] (tb)   B = Wt                       -> X'[y = c][x = 4g + r]
// and only W is transposed through LDS (256 int16 of per-wave scratch).  Smaller transforms are the 16-point product with the
// matrix zero outside n x n.  ta / tb per lane: XfLaneF16, one table entry per (transform, lane).
// =============================================================================================
typedef _Float16 kv_f16x4 __attribute__((ext_vector_type(4)));
typedef float kv_f32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) XfLaneF16 { uint16_t ta[4], tb[4]; };        // f16 bit patterns: T[c][4g + r], T[4g + r][c]
enum { XF16_DST4 = 0, XF16_DCT4 = 1, XF16_DCT8 = 2, XF16_DCT16 = 3 };
struct XfF16Tables { XfLaneF16 t[4][64]; };
constexpr uint16_t f16_bits_of_int(int v)                       // |v| < 2048
{
  if (v == 0) return 0;
  const uint16_t sign = v < 0 ? 0x8000 : 0;
  const int a = v < 0 ? -v : v;
  int e = 0;
  while ((a >> (e + 1)) != 0) e++;
  return (uint16_t)(sign | ((e + 15) << 10) | ((a << (10 - e)) & 0x3ff));
}
constexpr XfF16Tables make_xf_f16_tables()
{
  XfF16Tables t{};
  for (int type = 0; type < 4; type++) {
    const int l2 = type == XF16_DCT16 ? 4 : (type == XF16_DCT8 ? 3 : 2), n = 1 << l2;
    for (int lane = 0; lane < 64; lane++) {
      const int g = lane >> 4, c = lane & 15;
      for (int r = 0; r < 4; r++) {
        const int k = 4 * g + r;
        int a = 0, b = 0;
        if (c < n && k < n) {
          a = type == XF16_DST4 ? kDst4[c][k] : kDct32[c << (5 - l2)][k];
          b = type == XF16_DST4 ? kDst4[k][c] : kDct32[k << (5 - l2)][c];
        }
        t.t[type][lane].ta[r] = f16_bits_of_int(a); t.t[type][lane].tb[r] = f16_bits_of_int(b);
      }
    }
  }
  return t;
}
static __device__ const XfF16Tables g_xf16 = make_xf_f16_tables();

struct alignas(16) IntraWaveScratch {
  // what the prediction reads.  Planar / DC: the reference samples in the scan order of 8.4.4.2.2 (index 0 = bottom of the
  // below-left group, 2n = corner, 4n = end of above-right; as built or filtered, 8.4.4.2.3), sample i at byte 3 + i, which makes
  // the "above" run (2n + 1 ...) dword aligned: left[k] = R[2n - k], top[k] = R[2n + k].  Angular: the array ref[] of 8.4.4.2.6
  // with its projected part, ref[k] at byte n + k, k = -n .. 2n.
  uint8_t R[80];
  int16_t tr[256];                    // W between the inverse stages; decoder: also where the level words are scattered to
};

__device__ __forceinline__ kv_f16x4 kv_h4(const uint16_t (&b)[4])
{
  const uint2 u = *(const uint2 *)b;
  return __builtin_bit_cast(kv_f16x4, u);
}
// D = A B with the DATA as B operand: 16-bit signed integers d[r] = B[4g + r][c]; A = matrix constants
__device__ __forceinline__ void mfma16_data_b(kv_f16x4 a, const int (&d)[4], int (&out)[4])
{
  kv_f16x4 bh, bl;
#pragma unroll
  for (int r = 0; r < 4; r++) { bh[r] = (_Float16)(short)(d[r] >> 8); bl[r] = (_Float16)(unsigned short)(d[r] & 255); }
  const kv_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const kv_f32x4 sh = __builtin_amdgcn_mfma_f32_16x16x16f16(a, bh, z, 0, 0, 0);
  const kv_f32x4 sl = __builtin_amdgcn_mfma_f32_16x16x16f16(a, bl, z, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) out[r] = ((int)sh[r] << 8) + (int)sl[r];
}
// D = A B with the DATA as A operand: integers of at most 11 bits d[r] = A[c][4g + r]
__device__ __forceinline__ void mfma16_data_a(const int (&d)[4], kv_f16x4 b, int (&out)[4])
{
  kv_f16x4 a;
#pragma unroll
  for (int r = 0; r < 4; r++) a[r] = (_Float16)(short)d[r];
  const kv_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const kv_f32x4 s = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, z, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) out[r] = (int)s[r];
}

// What the serial chain of a (CTU, plane) needs to know about one intra block, worked out for all blocks of the CTU in parallel
// before the chain starts (positions, availability of the neighbours, the mode's constants): the chain itself only reads it.
struct alignas(16) IntraBlk {
  uint8_t rx, ry;            // position in the CTU, samples of the block's plane
  uint8_t lo, hi;            // the available reference samples are [lo, hi] in scan order (contiguous; hi < lo: none)
  uint8_t mode, l2;          // prediction mode; log2 of the size in samples of the block's plane (2 .. 5)
  uint8_t flags, xf;         // IB_*; transform table (XF16_*)
  int16_t angle, inv;        // intraPredAngle, invAngle (8.4.4.2.6)
  uint16_t zu, next;         // first 8x8 luma unit of the CTU the block lies in (z-order); encoder: z of the next block
};
enum { IB_FILT = 1,          // the filtered reference samples are used (8.4.4.2.3)
       IB_BORDER = 2,        // the block touches the CTU's left or upper border (the neighbouring CTUs' samples may have to be waited for)
       IB_PUBLISH = 4,       // progress `zu` is worth publishing before this block (a neighbour may be waiting for it)
       IB_LEVELS = 8, IB_TSKIP = 16,        // decoder: the block has levels; transform_skip_flag
       IB_EDGE_R = 64,       // the block holds samples of the CTU's right column: they also go, one byte per row, to the CTU's entry of the edge-column array -- the right neighbour reads its left border there in ONE transaction instead of one per picture line
       IB_HOLE = 128,        // decoder, pictures of several slices inside a tile: the available reference samples are TWO runs, [lo, 2N - 1] (the left part; lo = 2N: none) and
                             // [xf, hi] -- xf = 2N + 1: the slice begins with the coding tree block above, the corner sample belongs to another; xf = 3N + 1: it begins
                             // with the block above-right (IntraBlk::xf holds the second run's start there: the decoder's chain has no use for a transform table)
       IB_EDGE = 32 };       // the block ends on the CTU's bottom row: that row is stored write-through (the CTUs below read it, and the corner) -- the only ones another workgroup ever reads: they are stored write-through as soon as the block is done, everything else goes to the picture with the CTU's final copy
__device__ __forceinline__ IntraBlk wave_uniform(const IntraBlk *p)
{
  const uint4 u = *(const uint4 *)p;
  uint32_t w[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)u.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)u.y),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)u.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)u.w)};
  IntraBlk d;
  memcpy(&d, w, sizeof(d));
  return d;
}

// Prediction of one block on one wave: pred[r] = sample (x = 4g + r, y = c) for the lanes inside the block.  `pic` is the CTU
// picture with its borders (pitch P, sample (x, y) at pic[(y + 1) * P + 16 + x]).  Lane i first holds reference sample i of the
// scan order (the substitution process of 8.4.4.2.2 is a clamp of the scan index into [lo, hi]; the [1 2 1] filter takes its
// neighbours over DPP); what the mode reads is then laid out in ws.R -- the scan-order array for planar and DC, ref[] with the
// projected side samples for the angular modes (lanes fetch their entry from the lane that holds it: ds_bpermute).
// constrained intra prediction (decoder): what wave_intra_predict needs to ask whether a neighbouring sample belongs to an intra-coded block -- the 4x4 records'
// reference indices (ridx[8 * unit] < 0: intra; a record is 8 bytes), the block's place in its plane
struct CipCtx { const int8_t *ridx; int b4w; int xl, yl, sh; };
template <int L2, bool HOLES = false>
__device__ __forceinline__ void wave_intra_predict(const uint8_t *pic, int P, IntraWaveScratch &ws, const IntraBlk &d, bool luma, int lane, int g, int c, int (&pred)[4], const CipCtx *cip = nullptr)
{
  constexpr int N = 1 << L2;
  const int rx = d.rx, ry = d.ry, lo = d.lo, hi = d.hi, mode = d.mode;
  const bool hole = HOLES && (d.flags & IB_HOLE) != 0;      // (two available runs with a gap between them: a sample of the gap takes the last one of the run before it, 8.4.4.2.2)
  if (HOLES && mode == 35) { pred[0] = pred[1] = pred[2] = pred[3] = 0; return; }      // (decoder: a PCM unit -- its "residual" is the samples themselves)
  int v = 128, last = 128;
  if (HOLES && cip) {
    // constrained_intra_pred_flag: a sample is a reference sample when it is available in the usual sense AND its block is intra-coded -- any pattern along the two
    // borders.  8.4.4.2.2 as it stands: a sample that is not available takes the nearest available one BEFORE it in scan order, the ones in front of the first
    // available one take that one.  Lane i holds sample i; the ballot is the availability; the source lane comes from its bits.
    auto avail = [&](int i) -> bool {
      const bool sp = i <= 4 * N && (hole ? ((i >= lo && i < 2 * N) || (i >= (int)d.xf && i <= hi)) : (i >= lo && i <= hi));
      if (!sp) return false;
      const int px = i <= 2 * N ? cip->xl - 1 : cip->xl + i - 2 * N - 1, py = i < 2 * N ? cip->yl + 2 * N - 1 - i : cip->yl - 1;
      return cip->ridx[8 * ((size_t)((py << cip->sh) >> 2) * cip->b4w + ((px << cip->sh) >> 2))] < 0;
    };
    auto pos = [&](int j) -> int { const bool left = j < 2 * N; return (left ? ry + 2 * N - j : ry) * P + 16 + (left ? rx - 1 : rx + j - 2 * N - 1); };
    const bool av = avail(lane);
    const unsigned long long m = __ballot(av);
    const bool av64 = N == 16 && avail(64);
    const int own = av ? pic[pos(lane)] : 0, own64 = av64 ? pic[pos(64)] : 0;
    if (m || av64) {
      const unsigned long long below = lane ? m & (~0ull >> (64 - lane)) : 0ull;
      const int src = av ? lane : (below ? 63 - __builtin_clzll(below) : (m ? __builtin_ctzll(m) : 64));
      v = __builtin_amdgcn_ds_bpermute(4 * (src & 63), own);
      if (src == 64) v = own64;
      if (N == 16) last = av64 ? own64 : __builtin_amdgcn_readlane(own, 63 - __builtin_clzll(m | 1ull));      // (not available: the nearest one before it -- m != 0 here or av64 held)
    }
  } else
  if (hi >= lo) {
    auto at = [&](int i) -> int {
      int j = imin(imax(i, lo), hi);
      if (hole && j < d.xf) j = lo < 2 * N ? imin(j, 2 * N - 1) : d.xf;
      const bool left = j < 2 * N;
      return (left ? ry + 2 * N - j : ry) * P + 16 + (left ? rx - 1 : rx + j - 2 * N - 1);
    };
    v = pic[at(imin(lane, 4 * N))];
    if (N == 16) last = pic[at(4 * N)];
  }
  if (d.flags & IB_FILT) {
    const int prev = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);      // wave_shr:1 -- lane i <- lane i - 1
    int next = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);            // wave_shl:1 -- lane i <- lane i + 1
    if (N == 16 && lane == 63) next = last;
    if (lane != 0 && lane < 4 * N) v = (prev + 2 * v + next + 2) >> 2;
  }
  if (mode >= 2) {
    // ---- angular (8.4.4.2.6).  sgn walks the main side in scan order: +1 above (modes 18 ..), -1 left
    const int angle = d.angle, inv = d.inv;
    const bool vert = mode >= 18;
    const int sgn = vert ? 1 : -1;
    {
      const int k = lane - N, i = 2 * N + sgn * (k >= 0 ? k : -((k * inv + 128) >> 8));
      int e = __builtin_amdgcn_ds_bpermute(4 * i, v);
      if (N == 16 && i == 64) e = last;
      if (lane <= 3 * N) ws.R[lane] = (uint8_t)e;
    }
    int side = 0;
    const bool edge = luma && angle == 0;                   // (n < 32) modes 10 and 26: the first column / row follows the side samples
    if (edge) side = __builtin_amdgcn_ds_bpermute(4 * (2 * N - sgn * (1 + (vert ? c : 4 * g))), v);      // (unfiltered: these modes never filter)
    wave_sync();
    if (vert) {
      const int t = (c + 1) * angle, f = t & 31;
      const uint8_t *e = ws.R + N + 4 * g + (t >> 5) + 1;
      int s[5];
#pragma unroll
      for (int r = 0; r < 5; r++) s[r] = e[r];
#pragma unroll
      for (int r = 0; r < 4; r++) pred[r] = ((32 - f) * s[r] + f * s[r + 1] + 16) >> 5;
      if (edge && g == 0) pred[0] = clip8(ws.R[N + 1] + ((side - ws.R[N]) >> 1));
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int t = (4 * g + r + 1) * angle, f = t & 31;
        const uint8_t *e = ws.R + N + c + (t >> 5) + 1;
        pred[r] = ((32 - f) * e[0] + f * e[1] + 16) >> 5;
      }
      if (edge) {
        // row 0 of mode 10 follows the above samples: side of lane (g, 0) = top[1 + 4g]; the other three come the same way
        int sd[4] = {side, 0, 0, 0};
#pragma unroll
        for (int r = 1; r < 4; r++) sd[r] = __builtin_amdgcn_ds_bpermute(4 * (2 * N + 1 + 4 * g + r), v);
        if (c == 0) {
          const int corner = ws.R[N], first = ws.R[N + 1];
#pragma unroll
          for (int r = 0; r < 4; r++) pred[r] = clip8(first + ((sd[r] - corner) >> 1));
        }
      }
    }
  } else {
    if (N == 16 || lane <= 4 * N) ws.R[3 + lane] = (uint8_t)v;
    if (N == 16 && lane == 0) ws.R[3 + 64] = (uint8_t)last;
    uint32_t dcv = 0;
    if (mode == 1) {
      const uint32_t part = (lane >= N && lane <= 3 * N && lane != 2 * N) ? (uint32_t)v : 0u;    // left[1 .. n] and top[1 .. n]
      dcv = (wave_sum_u32(part) + N) >> (L2 + 1);
    }
    wave_sync();
    const uint8_t *R = ws.R + 3;
    const uint32_t top4 = *(const uint32_t *)(R + 2 * N + 1 + 4 * g);      // top[1 + 4g ..]
    const int lft = R[2 * N - 1 - c];                                        // left[1 + y]
    if (mode == 0) {
      const int tr = R[3 * N + 1], bl = R[N - 1];
      const int base = (c + 1) * bl + N;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int x = 4 * g + r;
        pred[r] = ((N - 1 - x) * lft + (x + 1) * tr + (N - 1 - c) * (int)((top4 >> (8 * r)) & 255u) + base) >> (L2 + 1);
      }
    } else {
      const int dc = (int)dcv;
#pragma unroll
      for (int r = 0; r < 4; r++) pred[r] = dc;
      if (luma) {                                           // (n < 32) boundary smoothing
        if (c == 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) pred[r] = ((int)((top4 >> (8 * r)) & 255u) + 3 * dc + 2) >> 2;
        }
        if (g == 0) pred[0] = c == 0 ? (lft + 2 * dc + (int)(top4 & 255u) + 2) >> 2 : (lft + 3 * dc + 2) >> 2;
      }
    }
  }
}

// per-block constants of the flat quantiser / dequantiser (hevc_core.h quant_level / dequant_coef, the same arithmetic in 32 bits
// where that is exact: |coefficient| <= 32768 times a scale <= 26214 plus the offset stays below 2^31)
struct QuantConst { int qscale, qshift, qoff, dscale, dsa, drnd, dsb; };
__device__ __forceinline__ QuantConst quant_const(int qp, int log2n, int intra)
{
  QuantConst q;
  q.qshift = 14 + qp / 6 + (15 - 8 - log2n);
  q.qscale = kQuantScale[qp % 6];
  q.qoff = (intra ? 171 : 85) << (q.qshift - 9);
  // (level * 16 * (scale << qp / 6) + (1 << (bd - 1))) >> bd with bd = log2n + 3: either the left shift absorbs bd (no rounding
  // happens) or the whole product stays below 2^31
  const int s1 = 4 + qp / 6, bd = 8 + log2n - 5;
  q.dscale = kLevelScale[qp % 6];
  if (s1 >= bd) { q.dsa = s1 - bd; q.drnd = 0; q.dsb = 0; } else { q.dsa = s1; q.drnd = 1 << (bd - 1); q.dsb = bd; }
  return q;
}
__device__ __forceinline__ int quant_level_q(int coef, const QuantConst &q)
{
  const uint32_t a = (uint32_t)iabs(coef);
  const int lv = imin((int)((a * (uint32_t)q.qscale + (uint32_t)q.qoff) >> q.qshift), 32767);
  return coef < 0 ? -lv : lv;
}
__device__ __forceinline__ int dequant_coef_q(int level, const QuantConst &q)      // level: 16 bits
{
  // |level * scale| < 2^22, shifted left by at most 7: 32 bits hold it
  return clip3(-32768, 32767, (((level * q.dscale) << q.dsa) + q.drnd) >> q.dsb);
}

// The same two with the scaling factor m of the coefficient's position (`scaling-list default`: m = 16 .. 115).  Forward scale (qscale << 4) / m <= qscale,
// so the product stays inside 32 bits as above; the dequantiser is 8.6.4.2 as written: (level * m * levelScale << qp / 6) >> bdShift with
// qp / 6 - bdShift = qshift - 24 (quant_const: qshift = 14 + qp / 6 + 7 - log2 n, bdShift = log2 n + 3).
__device__ __forceinline__ int quant_level_qm(int coef, const QuantConst &q, int m, int *du)
{
  const uint32_t a = (uint32_t)iabs(coef), prod = a * (uint32_t)((q.qscale << 4) / m);
  const int lv = imin((int)((prod + (uint32_t)q.qoff) >> q.qshift), 32767);
  if (du) *du = clip3(-256, 511, (int)(prod >> (q.qshift - 8)) - (lv << 8));
  return coef < 0 ? -lv : lv;
}
__device__ __forceinline__ int dequant_coef_qm(int level, const QuantConst &q, int m)
{
  const int k = q.qshift - 24;
  const long long v = (long long)level * m * q.dscale;
  const long long r = k >= 0 ? v << k : (v + (1ll << (-k - 1))) >> -k;
  return (int)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

// the dequantiser's constants of one block in one word (decoder: per transform block, worked out ahead of the chain)
__device__ __forceinline__ uint32_t dequant_pack(int qp, int log2n)
{
  const QuantConst q = quant_const(qp, log2n, 0);
  return (uint32_t)q.dscale | ((uint32_t)q.dsa << 8) | ((uint32_t)q.dsb << 16);
}
__device__ __forceinline__ int dequant_coef_p(int level, uint32_t pk)
{
  const int dsb = (int)(pk >> 16) & 255;
  return clip3(-32768, 32767, (((level * (int)(pk & 255u)) << ((pk >> 8) & 255u)) + (dsb ? 1 << (dsb - 1) : 0)) >> dsb);
}

// inverse transform of the dequantised coefficients dq[r] = C'[u = 4g + r][j = c] -> residual of the lane's samples (row owner)
__device__ __forceinline__ void wave_inverse16(IntraWaveScratch &ws, kv_f16x4 tb, const int (&dq)[4], int g, int c, int (&res)[4])
{
  int w[4];
  mfma16_data_b(tb, dq, w);
#pragma unroll
  for (int r = 0; r < 4; r++) ws.tr[(4 * g + r) * 16 + c] = (int16_t)clip3(-32768, 32767, (w[r] + 64) >> 7);
  wave_sync();
  const uint2 t = *(const uint2 *)&ws.tr[c * 16 + 4 * g];
  const int wt[4] = {(int)(int16_t)(t.x & 0xffffu), (int)(int16_t)(t.x >> 16), (int)(int16_t)(t.y & 0xffffu), (int)(int16_t)(t.y >> 16)};
  int x[4];
  mfma16_data_b(tb, wt, x);
#pragma unroll
  for (int r = 0; r < 4; r++) res[r] = (x[r] + 2048) >> 12;
  wave_sync();                                             // (ws.tr is free again)
}

}  // namespace kvzx
