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


// ---------------------------------------------------------------------------------------------
// Fence-free hand-off between workgroups inside a launch (cdna_hip_programming.md section 6, Guideline 16, recipe R1): the
// producer stores its payload WRITE-THROUGH (agent-scope relaxed atomic stores = `sc1`), every storing wave drains its stores,
// one lane stores the flag; the consumer polls the flag relaxed and reads the payload with agent-scope relaxed loads (`sc1`:
// served past the CU's L1).  No agent-scope release / acquire fence (1.7 .. 7 us each on gfx950) anywhere.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_wt_u32(void *p, uint32_t v) { __hip_atomic_store((uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt_u64(void *p, uint64_t v) { __hip_atomic_store((uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_l2_u32(const void *p) { return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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
struct IntraBorders {
  const uint32_t *pl, *pu, *pur, *pul;       // progress counters of the left / upper / upper-right / upper-left CTU (same plane)
  bool nb_left, nb_up, nb_ur, nb_ul;         // which of them exist (inside the picture, same tile)
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
                                             int lim_w, int rx, int ry, int n, uint32_t *bcast, uint32_t *err, int tid, int nthreads)
{
  bool loaded = false;
  if (rx == 0 && b.nb_left) {
    const int need = imin(S, ry + 2 * n);
    if (need > b.left_loaded) {
      b.seen_l = wait_wt(b.pl, (uint32_t)kv_zunit8(7, ((need - 1) << sh) >> 3) + 1, b.seen_l, bcast, err);
      const int upto = imax(need, imin(S, kv_units_right(b.seen_l) * (8 >> sh)));
      for (int i = b.left_loaded + tid; i < upto; i += nthreads) pic[(i + 1) * lp + 15] = (uint8_t)ld_l2_u8(plane + (size_t)(cy * S + i) * gp + cx * S - 1);
      b.left_loaded = upto; loaded = true;
    }
  }
  if (ry == 0 && b.nb_up) {
    const int lim = imin(lim_w, b.nb_ur ? 2 * S : S), need = imin(lim, rx + 2 * n);
    if (need > b.top_loaded) {
      int upto = need;
      if (b.top_loaded < S) {
        b.seen_u = wait_wt(b.pu, (uint32_t)kv_zunit8(((imin(S, need) - 1) << sh) >> 3, 7) + 1, b.seen_u, bcast, err);
        upto = imax(upto, imin(imin(S, lim), kv_units_bottom(b.seen_u) * (8 >> sh)));
      }
      if (need > S) {
        b.seen_ur = wait_wt(b.pur, (uint32_t)kv_zunit8(((need - S - 1) << sh) >> 3, 7) + 1, b.seen_ur, bcast, err);
        upto = imax(upto, imin(lim, S + kv_units_bottom(b.seen_ur) * (8 >> sh)));
      }
      // (both ends are multiples of 4: block sizes, CTU sizes and picture widths are)
      for (int i = b.top_loaded + 4 * tid; i < upto; i += 4 * nthreads) *(uint32_t *)&pic[16 + i] = ld_l2_u32(plane + (size_t)(cy * S - 1) * gp + cx * S + i);
      b.top_loaded = upto; loaded = true;
    }
  }
  if (rx == 0 && ry == 0 && b.nb_ul && !b.corner_loaded) {
    b.seen_ul = wait_wt(b.pul, 64u, b.seen_ul, bcast, err);
    if (tid == 0) pic[15] = (uint8_t)ld_l2_u8(plane + (size_t)(cy * S - 1) * gp + cx * S - 1);
    b.corner_loaded = true; loaded = true;
  }
  if (loaded) __syncthreads();
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


}  // namespace kvzx
