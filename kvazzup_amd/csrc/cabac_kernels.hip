// kvazzup_amd/csrc/cabac_kernels.hip -- the serial half of entropy coding on the GPU: k_cabac_rows.
//
// k_tokenize / k_tok_compact leave, per CTU, the bins of the picture in coding order as 16-bit tokens (hevc_core.h TokOut).
// What remains of CABAC (H.265 9.3.4.3) is one state machine per substream -- a CTU row with WPP, else a tile -- in which every
// bin depends on the one before.  One WAVE runs one substream, as a scalar program: a wave alone on its SIMD issues one
// instruction every four to five cycles whatever the instruction is, so the cost of a bin is its instruction count, and the
// design below is about keeping that count (and every memory round trip) off the chain:
//   * the coder's registers (low, range, bits_left, the pending byte run) are wave-uniform values, i.e. SGPRs;
//   * the 154 context variables live in ONE vector register -- four to a lane, lane = context >> 2 -- read with v_readlane and
//     written with a compare + select on the lane index; rangeTabLps (a row of four bytes per state) and the LPS transition table are one register each,
//     indexed by v_readlane with the state: no LDS or memory access per bin;
//   * tokens are fetched 64 at a time (one coalesced 128-byte load, issued one batch ahead) and handed to the scalar loop by
//     v_readlane; output bytes are gathered four to a dword, 64 dwords to a register, and leave as one 256-byte
//     store to a staging range in HBM (reserved for the worst case, two bytes per token); the finished substream, now of known
//     length, is copied densely into the host-mapped output buffer;
//   * WPP: row r starts from the contexts row r - 1 had after its second CTU (9.3.2.2): 39 lanes store the context register
//     write-through, the flag follows the drained stores (cdna_hip_programming.md section 6, recipe R1), the row below polls it.
// Measured on MI355X: see DESIGN.md section 5.  The same arithmetic, on host threads, is EntropyHost (entropy_host.h); the
// byte-exact statement of record is oracle/hevc_cabac.c.
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

namespace {

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// value v into lane `lane` of `old` (v_writelane_b32 takes one scalar operand on gfx9, so it would need M0 for the lane: a compare and a select cost the same two issues)
__device__ __forceinline__ uint32_t wl(uint32_t v, uint32_t lane, uint32_t old) { return threadIdx.x == lane ? v : old; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The arithmetic coder of one substream.  Every member except `outv` is wave-uniform.
struct RowCoder {
  uint32_t low, range, buffered_byte, acc, nacc, nd, written, pos, nbins;
  int bits_left, num_buffered;
  uint32_t outv;              // output staging: lane k holds dword k of the 256-byte block being filled
  uint32_t *outw; int lane;

  __device__ __forceinline__ void put_byte(uint32_t b)
  {
    acc |= (b & 0xffu) << (8 * nacc);
    pos++;
    if (++nacc == 4) {
      outv = wl(acc, nd, outv); acc = 0; nacc = 0;
      if (++nd == 64) { outw[written + lane] = outv; written += 64; nd = 0; }
    }
  }
  __device__ __forceinline__ void write_out()
  {
    const uint32_t lead = low >> (24 - bits_left);
    bits_left += 8;
    low &= 0xffffffffu >> bits_left;
    if (lead == 0xff) { num_buffered++; return; }
    if (num_buffered > 0) {
      const uint32_t carry = lead >> 8;
      put_byte(buffered_byte + carry);
      buffered_byte = lead & 0xff;
      const uint32_t fill = (0xff + carry) & 0xff;
      while (num_buffered > 1) { put_byte(fill); num_buffered--; }
    } else { num_buffered = 1; buffered_byte = lead; }
  }
  __device__ __forceinline__ void bypass_bits(uint32_t val, int n)       // n in [1, 10]
  {
    nbins += (uint32_t)n;
    if (n > 8) {
      n -= 8;
      const uint32_t pat = val >> n;
      low = (low << 8) + range * pat;
      val -= pat << n;
      bits_left -= 8;
      if (bits_left < 12) write_out();
    }
    low = (low << n) + range * val;
    bits_left -= n;
    if (bits_left < 12) write_out();
  }
  __device__ __forceinline__ void terminate(uint32_t bin)
  {
    nbins++;
    range -= 2;
    if (bin) { low += range; low <<= 7; range = 2 << 7; bits_left -= 7; }
    else if (range >= 256) return;
    else { low <<= 1; range <<= 1; bits_left--; }
    if (bits_left < 12) write_out();
  }
  // after the substream's last terminating bin (= 1): the rest of `low`, the stop bit, alignment zeros (hevc_core.h cabac_finish)
  __device__ __forceinline__ void finish()
  {
    if (low >> (32 - bits_left)) {
      put_byte(buffered_byte + 1);
      while (num_buffered > 1) { put_byte(0x00); num_buffered--; }
      low -= 1u << (32 - bits_left);
    } else {
      if (num_buffered > 0) put_byte(buffered_byte);
      while (num_buffered > 1) { put_byte(0xff); num_buffered--; }
    }
    int nbits = 24 - bits_left;
    uint32_t v = ((low >> 8) << 1) | 1u; nbits += 1;
    const int pad = (8 - (nbits & 7)) & 7;
    v <<= pad; nbits += pad;
    for (int sh = nbits - 8; sh >= 0; sh -= 8) put_byte((v >> sh) & 0xff);
    // what is left in the staging register
    if (nacc) { outv = wl(acc, nd, outv); nd++; }
    if ((uint32_t)lane < nd) outw[written + lane] = outv;
  }
};

}  // namespace

__global__ __launch_bounds__(64) void k_cabac_rows(CabacRowsArgs a)
{
  const int lane = threadIdx.x;
  const int r = a.first_sub + (int)blockIdx.x;          // substream: CTU row r with WPP, else tile row r
  const int first_cy = a.wpp ? r : tile_row_first(a.hc, a.tile_rows, r);
  const int ncy = a.wpp ? 1 : tile_row_first(a.hc, a.tile_rows, r + 1) - first_cy;
  const int nctu = ncy * a.wc, ctu0 = first_cy * a.wc;
  // ---- the substream's tokens -> a range of the output buffer (a token never makes more than two bytes)
  uint32_t ntok = 0; bool bad = false;
  for (int i = lane; i < nctu; i += 64) { const int c = a.count[ctu0 + i]; if (c < 0) bad = true; else ntok += (uint32_t)c; }
  ntok = wave_sum_u32(ntok);
  bad = __ballot(bad) != 0;
  const uint32_t need = (ntok * 2 + 64 + 255) & ~255u;
  uint32_t o = 0;
  if (lane == 0) o = atomicAdd(a.cursors, need);
  o = uni(o);
  const bool fits = !bad && ntok < (1u << 30) && o + need <= a.stage_cap;
  // ---- tables and contexts into registers
  const uint32_t lpsv = (uint32_t)kRangeLps[lane][0] | ((uint32_t)kRangeLps[lane][1] << 8) | ((uint32_t)kRangeLps[lane][2] << 16) | ((uint32_t)kRangeLps[lane][3] << 24);
  const uint32_t nlpsv = kNextLps[lane];
  uint32_t ctxv = 0;
  bool failed = false;
  const bool fresh = !a.wpp || tile_row_starts_at(a.hc, a.tile_rows, r);
  if (fresh) {
    const int qp = clip3(0, 51, a.qp);
    for (int k = 0; k < 4; k++) {
      const int i = lane * 4 + k;
      if (i < CTX_COUNT) {
        const int v = kCabacInit[a.init_type][i];
        const int slope = (v >> 4) * 5 - 45, offs = ((v & 15) << 3) - 16;
        const int pre = clip3(1, 126, ((slope * qp) >> 4) + offs);
        const int mps = pre <= 63 ? 0 : 1;
        ctxv |= (uint32_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps) << (8 * k);
      }
    }
  } else {
    // (the row above was dispatched before this one: it is running or done)
    uint32_t gave_up = 0;
    if (lane == 0) {
      uint32_t spins = 0;
      while (__hip_atomic_load(a.ctx_ready + (r - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.gen) {
        if (++spins < 64) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
        if (spins > (1u << 24)) { atomicOr(a.err, 64u); gave_up = 1; break; }         // bounded: never hang the GPU
      }
    }
    failed = uni(gave_up) != 0;
    if (lane < 40) ctxv = ld_l2_u32(a.ctx_save + (size_t)(r - 1) * 40 + lane);
  }
  if (!fits) {                                          // the picture is lost (the host sees the length); the row below must still start
    if (a.wpp && lane == 0) __hip_atomic_store(a.ctx_ready + r, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) { a.sub_off[blockIdx.x] = 0; a.sub_len[blockIdx.x] = ~0u; a.sub_bins[blockIdx.x] = 0; }
    return;
  }
  RowCoder c;
  c.low = 0; c.range = 510; c.bits_left = 23; c.num_buffered = 0; c.buffered_byte = 0xff;
  c.acc = 0; c.nacc = 0; c.nd = 0; c.written = 0; c.pos = 0; c.nbins = 0; c.outv = 0;
  c.outw = (uint32_t *)(a.stage + o); c.lane = lane;

  // ---- the token stream, CTU after CTU, 64 tokens per batch, the next batch in flight while this one is coded
  int cx = 0, i0 = 0;
  int n = a.count[ctu0]; uint32_t base = a.off[ctu0];
  uint32_t tokv = (i0 + lane < n) ? a.tok[base + i0 + lane] : 0;
  while (cx < nctu) {
    int ncx = cx, ni0 = i0 + 64, nn = n; uint32_t nbase = base;
    if (ni0 >= n) {
      ncx = cx + 1; ni0 = 0;
      while (ncx < nctu) { nn = a.count[ctu0 + ncx]; nbase = a.off[ctu0 + ncx]; if (nn > 0) break; ncx++; }     // (every CTU has its terminating bin: never empty in practice)
    }
    uint32_t tokn = 0;
    if (ncx < nctu && ni0 + lane < nn) tokn = a.tok[nbase + ni0 + lane];
    const int m = n - i0 < 64 ? n - i0 : 64;
    for (int j = 0; j < m; j++) {
      const uint32_t t = rl(tokv, (uint32_t)j);
      if (!(t & 0x8000u)) {
        const uint32_t ci = t >> 1, cl = ci >> 2, sh = (ci & 3) * 8;
        const uint32_t w = rl(ctxv, cl), s = (w >> sh) & 0xffu, st = s >> 1;
        const uint32_t lps = (rl(lpsv, st) >> (((c.range >> 6) & 3) * 8)) & 0xffu;
        uint32_t ns;
        c.nbins++;
        c.range -= lps;
        if ((t ^ s) & 1u) {                               // least probable symbol
          const int nb = __builtin_clz(lps) - 23;
          c.low = (c.low + c.range) << nb; c.range = lps << nb; c.bits_left -= nb;
          ns = (rl(nlpsv, st) << 1) | ((s & 1u) ^ (st == 0 ? 1u : 0u));
        } else {
          ns = s + (st < 62 ? 2u : 0u);
          if (c.range < 256) { c.low <<= 1; c.range <<= 1; c.bits_left--; }
        }
        ctxv = wl((w & ~(0xffu << sh)) | (ns << sh), cl, ctxv);
        if (c.bits_left < 12) c.write_out();
      } else if (!(t & 0x4000u)) c.bypass_bits(t & 0x3ffu, (int)((t >> 10) & 15) + 1);
      else c.terminate(t & 1u);
    }
    if (i0 + 64 >= n) {                                   // the CTU is complete
      if (a.wpp && cx == 1) {
        if (lane < 40) st_wt_u32(a.ctx_save + (size_t)r * 40 + lane, ctxv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.ctx_ready + r, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      n = nn; base = nbase;
    }
    cx = ncx; i0 = ni0; tokv = tokn;
  }
  c.finish();
  // ---- the substream's bytes, now of known length, move from the staging range to the host-visible buffer (dense, in completion order)
  const uint32_t len4 = (c.pos + 15u) & ~15u;
  uint32_t o2 = 0;
  if (lane == 0) o2 = atomicAdd(a.cursors + 1, len4);
  o2 = uni(o2);
  if (o2 + len4 > a.out_cap) { if (lane == 0) { a.sub_off[blockIdx.x] = 0; a.sub_len[blockIdx.x] = ~0u; a.sub_bins[blockIdx.x] = 0; } return; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the wave's own staging stores (its CU's L1 is write-through and coherent for its own waves)
  const kv_u32x4 *sp = (const kv_u32x4 *)(a.stage + o);
  kv_u32x4 *dp = (kv_u32x4 *)(a.out + o2);
  for (uint32_t i = lane; i < len4 / 16; i += 64) dp[i] = sp[i];
  if (lane == 0) { a.sub_off[blockIdx.x] = o2; a.sub_len[blockIdx.x] = failed ? ~0u : c.pos; a.sub_bins[blockIdx.x] = c.nbins; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Measurement aid (KVAZZUP_AMD_PARSE_PROBE=1 with gpu-entropy=1; profiles/r05_gpu_parse_ab.txt): the arithmetic DECODER of 9.3.4.3 as a wave-scalar
// program, the mirror image of k_cabac_rows -- the lower bound of what parsing slice data on the GPU would cost per bin.  One wave per substream decodes
// the bytes k_cabac_rows just wrote, with the context of every bin taken from the substream's own token list (a real parser has to derive it from the
// syntax: more instructions per bin, never fewer) and checks every decoded bin against the token's value; mismatches are counted (none may occur).
// Registers as in k_cabac_rows: contexts four to a lane, rangeTabLps and both state transitions one lane per state; the substream's bytes arrive 256 at a
// time (one dword per lane) and are taken out with v_readlane.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cabac_decode_probe(CabacRowsArgs a, uint32_t *mismatch)
{
  const int lane = threadIdx.x;
  const int r = a.first_sub + (int)blockIdx.x;
  const int first_cy = a.wpp ? r : tile_row_first(a.hc, a.tile_rows, r);
  const int ncy = a.wpp ? 1 : tile_row_first(a.hc, a.tile_rows, r + 1) - first_cy;
  const int nctu = ncy * a.wc, ctu0 = first_cy * a.wc;
  const uint32_t len = a.sub_len[blockIdx.x];
  if (len == ~0u || len == 0) return;
  const uint32_t *bytes = (const uint32_t *)(a.out + a.sub_off[blockIdx.x]);      // (substreams start on 16-byte boundaries of the output buffer)
  const uint32_t lpsv = (uint32_t)kRangeLps[lane][0] | ((uint32_t)kRangeLps[lane][1] << 8) | ((uint32_t)kRangeLps[lane][2] << 16) | ((uint32_t)kRangeLps[lane][3] << 24);
  const uint32_t nlpsv = kNextLps[lane];
  uint32_t ctxv = 0;
  const bool fresh = !a.wpp || tile_row_starts_at(a.hc, a.tile_rows, r);
  if (fresh) {
    const int qp = clip3(0, 51, a.qp);
    for (int k = 0; k < 4; k++) {
      const int i = lane * 4 + k;
      if (i < CTX_COUNT) {
        const int v = kCabacInit[a.init_type][i];
        const int slope = (v >> 4) * 5 - 45, offs = ((v & 15) << 3) - 16;
        const int pre = clip3(1, 126, ((slope * qp) >> 4) + offs);
        const int mps = pre <= 63 ? 0 : 1;
        ctxv |= (uint32_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps) << (8 * k);
      }
    }
  } else if (lane < 40) ctxv = ld_l2_u32(a.ctx_save + (size_t)(r - 1) * 40 + lane);      // (k_cabac_rows of this picture has finished: every row's hand-over copy is there)
  // ---- the bitstream: dword `w` of the substream, big-endian bits; wv holds dwords [wbase, wbase + 64)
  uint32_t wbase = 0, wv = (wbase + lane) * 4 < len + 3 ? bytes[wbase + lane] : 0;
  uint32_t nextw = 0;                                    // next dword to take
  auto take_word = [&]() -> uint32_t {
    if (nextw - wbase >= 64) { wbase += 64; wv = (wbase + lane) * 4 < len + 3 ? bytes[wbase + lane] : 0; }
    const uint32_t w = rl(wv, nextw - wbase); nextw++;
    return __builtin_bswap32(w);
  };
  // value = ivlOffset scaled by 2^bits, followed by `bits` not yet consumed stream bits (the host decoder's form, decoder.hip CabacRegs)
  unsigned long long value = take_word(); int bits = 32 - 9; uint32_t range = 510;
  { value = (value << 32) | take_word(); bits += 32; }
  uint32_t bad = 0, nbins = 0;
  int cx = 0, i0 = 0;
  int n = a.count[ctu0]; uint32_t base = a.off[ctu0];
  uint32_t tokv = (i0 + lane < n) ? a.tok[base + i0 + lane] : 0;
  while (cx < nctu) {
    int ncx = cx, ni0 = i0 + 64, nn = n; uint32_t nbase = base;
    if (ni0 >= n) {
      ncx = cx + 1; ni0 = 0;
      while (ncx < nctu) { nn = a.count[ctu0 + ncx]; nbase = a.off[ctu0 + ncx]; if (nn > 0) break; ncx++; }
    }
    uint32_t tokn = 0;
    if (ncx < nctu && ni0 + lane < nn) tokn = a.tok[nbase + ni0 + lane];
    const int m = n - i0 < 64 ? n - i0 : 64;
    for (int j = 0; j < m; j++) {
      const uint32_t t = rl(tokv, (uint32_t)j);
      if (!(t & 0x8000u)) {
        const uint32_t ci = t >> 1, cl = ci >> 2, sh = (ci & 3) * 8;
        const uint32_t w = rl(ctxv, cl), s = (w >> sh) & 0xffu, st = s >> 1;
        const uint32_t lps = (rl(lpsv, st) >> (((range >> 6) & 3) * 8)) & 0xffu;
        nbins++;
        range -= lps;
        const unsigned long long scaled = (unsigned long long)range << bits;
        uint32_t ns, bin;
        if (value >= scaled) {                              // least probable symbol
          value -= scaled; bin = (s & 1u) ^ 1u;
          const int nb = __builtin_clz(lps) - 23;
          range = lps << nb; bits -= nb;
          ns = (rl(nlpsv, st) << 1) | ((s & 1u) ^ (st == 0 ? 1u : 0u));
        } else {
          bin = s & 1u;
          ns = s + (st < 62 ? 2u : 0u);
          if (range < 256) { range <<= 1; bits--; }
        }
        ctxv = wl((w & ~(0xffu << sh)) | (ns << sh), cl, ctxv);
        bad += bin != (t & 1u);
      } else if (!(t & 0x4000u)) {
        const int nb = (int)((t >> 10) & 15) + 1;
        nbins += (uint32_t)nb;
        uint32_t v = 0;
        for (int k = 0; k < nb; k++) { bits--; const unsigned long long scaled = (unsigned long long)range << bits; v <<= 1; if (value >= scaled) { value -= scaled; v |= 1u; } }
        bad += v != (t & ((1u << nb) - 1u));
      } else {
        nbins++;
        range -= 2;
        uint32_t bin = 0;
        if (value >= ((unsigned long long)range << bits)) bin = 1;
        else if (range < 256) { range <<= 1; bits--; }
        bad += bin != (t & 1u);
      }
      if (bits < 16) { value = (value << 32) | take_word(); bits += 32; }
    }
    if (i0 + 64 >= n) { n = nn; base = nbase; }
    cx = ncx; i0 = ni0; tokv = tokn;
  }
  if (lane == 0) { atomicAdd(mismatch, bad); atomicAdd(mismatch + 1, nbins); }
}

void launch_cabac_decode_probe(const CabacRowsArgs &a, int nsub, uint32_t *mismatch, hipStream_t st)
{
  hipLaunchKernelGGL(k_cabac_decode_probe, dim3(nsub), dim3(64), 0, st, a, mismatch);
}

void launch_cabac_rows(const CabacRowsArgs &a, int nsub, hipStream_t st)
{
  hipLaunchKernelGGL(k_cabac_rows, dim3(nsub), dim3(64), 0, st, a);
}

}  // namespace kvzx
