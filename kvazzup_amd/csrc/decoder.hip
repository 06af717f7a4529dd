// kvazzup_amd/csrc/decoder.hip -- see decoder.h.  Host half of the decoder: NAL units, parameter sets, slice headers,
// reference picture management and the CABAC slice-data parser (H.265 7.3, 8.3, 9.3); the sample work is dec_kernels.hip.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "stream_pool.h"
#include "pic_hash.h"
#include "decoder.h"
#include "scaling_tables.h"

namespace kvzx {

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "kvazzup_amd: %s failed: %s\n", #expr, hipGetErrorString(e_)); return false; } } while (0)
enum { DEC_ERR_INVALID = -1, DEC_ERR_UNSUPPORTED = -2, DEC_ERR_GPU = -3, DEC_ERR_HASH = -4, DEC_SEG_ENDS_EARLY = -100 /* internal: PicJob::ambiguous_end */ };      // (-4: a decoded picture hash SEI did not match, libOpenHevcSetCheckMD5)
enum { PM_INTER = 0, PM_INTRA = 1, PM_SKIP = 2, PM_NONE = 255 };
enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };

namespace {

// ------------------------------------------------------------------------------------------ bits
struct BitReader {
  const uint8_t *p; size_t n, pos = 0; bool err = false;
  BitReader(const uint8_t *b, size_t len) : p(b), n(len) {}
  uint32_t bit() { if (pos >= n * 8) { err = true; pos++; return 0; } uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
  uint32_t get(int k) { uint32_t v = 0; for (int i = 0; i < k; i++) v = (v << 1) | bit(); return v; }
  // (at most 30 leading zeros: the value stays below 2^31, so that every `(int)r.ue()` and `r.ue() + 1` below is a non-negative int and the callers' upper-bound
  // checks are range checks -- a 32-bit code word used to come out as a NEGATIVE index that passed `id > 63`; found by tools/fuzz_parser.py under ASan)
  uint32_t ue() { int z = 0; while (!bit()) { if (++z > 30 || err) { err = true; return 0; } } return z ? ((1u << z) - 1) + get(z) : 0; }
  int32_t se() { uint32_t k = ue(); return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1); }
};

bool skip_ptl(BitReader &r, int max_sub_layers_minus1)
{
  r.get(8); r.get(32); r.get(4); r.get(32); r.get(11); r.get(1); r.get(8);
  int pp[8], lp[8];
  for (int i = 0; i < max_sub_layers_minus1; i++) { pp[i] = r.get(1); lp[i] = r.get(1); }
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) r.get(2);
  for (int i = 0; i < max_sub_layers_minus1; i++) { if (pp[i]) { r.get(32); r.get(32); r.get(24); } if (lp[i]) r.get(8); }
  return !r.err;
}

// st_ref_pic_set(idx) (7.3.7, 7.4.8): explicit or predicted from an earlier set
bool parse_st_rps(BitReader &r, int idx, int num_in_sps, const StRps *all, StRps &out)
{
  out = StRps();
  int inter = 0;
  if (idx != 0) inter = r.get(1);
  if (inter) {
    int delta_idx = 1;
    if (idx == num_in_sps) delta_idx = (int)r.ue() + 1;
    if (delta_idx > idx) return false;
    const StRps &ref = all[idx - delta_idx];
    const int sign = r.get(1), absd = (int)r.ue() + 1, drps = (1 - 2 * sign) * absd, nd = ref.n_neg + ref.n_pos;
    int used[17], use_delta[17];
    for (int j = 0; j <= nd; j++) { used[j] = r.get(1); use_delta[j] = 1; if (!used[j]) use_delta[j] = r.get(1); }
    int s0[16], u0[16], s1[16], u1[16], n0 = 0, n1 = 0;
    const int *rs0 = ref.dpoc, *rs1 = ref.dpoc + ref.n_neg; const uint8_t *unused = nullptr; (void)unused;
    for (int j = ref.n_pos - 1; j >= 0; j--) { const int d = rs1[j] + drps; if (d < 0 && use_delta[ref.n_neg + j] && n0 < 16) { s0[n0] = d; u0[n0++] = used[ref.n_neg + j]; } }
    if (drps < 0 && use_delta[nd] && n0 < 16) { s0[n0] = drps; u0[n0++] = used[nd]; }
    for (int j = 0; j < ref.n_neg; j++) { const int d = rs0[j] + drps; if (d < 0 && use_delta[j] && n0 < 16) { s0[n0] = d; u0[n0++] = used[j]; } }
    for (int j = ref.n_neg - 1; j >= 0; j--) { const int d = rs0[j] + drps; if (d > 0 && use_delta[j] && n1 < 16) { s1[n1] = d; u1[n1++] = used[j]; } }
    if (drps > 0 && use_delta[nd] && n1 < 16) { s1[n1] = drps; u1[n1++] = used[nd]; }
    for (int j = 0; j < ref.n_pos; j++) { const int d = rs1[j] + drps; if (d > 0 && use_delta[ref.n_neg + j] && n1 < 16) { s1[n1] = d; u1[n1++] = used[ref.n_neg + j]; } }
    if (n0 + n1 > 16) return false;
    out.n_neg = n0; out.n_pos = n1;
    for (int j = 0; j < n0; j++) { out.dpoc[j] = s0[j]; out.used[j] = (uint8_t)u0[j]; }
    for (int j = 0; j < n1; j++) { out.dpoc[n0 + j] = s1[j]; out.used[n0 + j] = (uint8_t)u1[j]; }
  } else {
    const int nneg = (int)r.ue(), npos = (int)r.ue();
    if (nneg > 16 || npos > 16 || nneg + npos > 16 || r.err) return false;
    out.n_neg = nneg; out.n_pos = npos;
    int prev = 0;
    for (int j = 0; j < nneg; j++) { prev -= (int)r.ue() + 1; out.dpoc[j] = prev; out.used[j] = (uint8_t)r.get(1); }
    prev = 0;
    for (int j = 0; j < npos; j++) { prev += (int)r.ue() + 1; out.dpoc[nneg + j] = prev; out.used[nneg + j] = (uint8_t)r.get(1); }
  }
  return !r.err;
}

// ------------------------------------------------------------------------------------------ CABAC decoding (H.265 9.3.4.3)
// Arithmetic decoder with the offset kept scaled in a 64-bit register: value = offset << bits | next
// `bits` stream bits, so a renormalisation by n is just bits -= n and the stream is touched 32 bits at
// a time.  Context variable = pStateIdx << 1 | valMps with precomputed transitions.
// next[variable][LPS decoded], lps[variable][(range >> 6) & 3]; lpsn[variable][q] = the LPS range already renormalised (bits 0..8) | its shift << 16: an LPS
// range's renormalisation depends on the table entry alone, so it is looked up with it instead of counted (a leading-zero count, a subtraction and a shift
// less on the range's dependency chain); an MPS range (>= 128) is shifted by one at most
struct StateTabs { uint8_t next_mps[128], next_lps[128]; uint8_t next[128][2]; uint8_t lps[128][4]; uint32_t lpsn[128][4]; };
const StateTabs &state_tabs()               // (function-local statics: initialised once, thread-safe -- parse workers race to the first call)
{
  static const StateTabs t = [] {
    StateTabs t;
    for (int s = 0; s < 128; s++) {
      int st = s >> 1, mps = s & 1;
      t.next_mps[s] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
      t.next_lps[s] = (uint8_t)((kNextLps[st] << 1) | (st == 0 ? mps ^ 1 : mps));
      t.next[s][0] = t.next_mps[s]; t.next[s][1] = t.next_lps[s];
      for (int q = 0; q < 4; q++) { t.lps[s][q] = kRangeLps[st][q]; const int n = __builtin_clz((uint32_t)kRangeLps[st][q]) - 23; t.lpsn[s][q] = ((uint32_t)kRangeLps[st][q] << n) | ((uint32_t)n << 16) | ((uint32_t)kRangeLps[st][q] << 24); }
    }
    return t;
  }();
  return t;
}
// The decoder's registers apart from the context variables: a function that decodes many bins in a row (parse_residual) works on a LOCAL copy -- a local
// whose address never leaves the function lives in registers, while the members of an object reached through a reference are re-loaded and written back
// around every context store (measured on the parser alone, tools/measure/parse_rate.py: 11.3 -> see HISTORY.md ns per bin at 1080p / QP 32).
struct CabacRegs {
  const uint8_t *p = nullptr, *end = nullptr;                   // next unread byte, end of the substream; reads past `end` deliver zeros (a malformed NAL cannot walk off the buffer)
  uint64_t value = 0; int bits = 0;
  uint32_t range = 510; uint32_t past = 0;                      // 32-bit words fetched beyond the end
  const StateTabs *st = nullptr;
  uint16_t *ctx = nullptr;                                      // 16-bit entries: a byte store may alias anything; a uint16_t store cannot alias the fields above
  // (every member function is forced inline: one call with `this` would pin a local copy to the stack)
  static __attribute__((noinline)) uint32_t word_tail(const uint8_t *p, const uint8_t *end) { uint32_t w = 0; for (int i = 0; i < 4; i++) w = (w << 8) | (p + i < end ? p[i] : 0u); return w; }
  __attribute__((always_inline)) inline uint32_t word()
  {
    uint32_t w;
    if (__builtin_expect(p + 4 <= end, 1)) { memcpy(&w, p, 4); w = __builtin_bswap32(w); }
    else { w = word_tail(p, end); past++; }
    p += 4;
    return w;
  }
  __attribute__((always_inline)) inline void refill() { if (__builtin_expect(bits < 16, 0)) { value = (value << 32) | word(); bits += 32; } }
  bool overrun() const { return past > 3; }
  __attribute__((always_inline)) inline int bin(int ci)
  {
    // (both outcomes are computed and selected: the bin values of sig / greater1 flags are close to coin flips for a branch predictor)
    const uint32_t s = ctx[ci];
    const uint32_t e = st->lpsn[s][(range >> 6) & 3];     // LPS range: as it is (bits 24..31), renormalised (bits 0..8), its shift (bits 16..19)
    const uint32_t rmps = range - (e >> 24);
    const uint64_t scaled = (uint64_t)rmps << bits;
    const bool isl = value >= scaled;
    value -= isl ? scaled : 0;
    const uint32_t nm = (rmps >> 8) ^ 1u;                 // an MPS range is in [128, 510]: one shift when below 256
    range = isl ? (e & 0x1ffu) : (rmps << nm);
    bits -= (int)(isl ? ((e >> 16) & 15u) : nm);
    ctx[ci] = st->next[s][isl];
    refill();
    return (int)((s & 1u) ^ (uint32_t)isl);
  }
  __attribute__((always_inline)) inline int bypass()
  {
    bits--;
    const uint64_t scaled = (uint64_t)range << bits;
    int b = 0;
    if (value >= scaled) { value -= scaled; b = 1; }
    refill();
    return b;
  }
  // n bypass bins at once: they are the n-bit quotient of value by range << (bits - n) (binary long division, one step per
  // bin); refill() keeps bits >= 16, so up to 16 bins go in one division
  __attribute__((always_inline)) inline uint32_t bypass_bits(int n)
  {
    uint32_t v = 0;
    while (n > 0) {
      const int m = n > 16 ? 16 : n;
      if (m <= 2) { for (int i = 0; i < m; i++) v = (v << 1) | (uint32_t)bypass(); }
      else {
        bits -= m;
        const uint64_t scaled = (uint64_t)range << bits;
        const uint64_t q = value / scaled;
        value -= q * scaled;
        v = (v << m) | (uint32_t)(q & 0xffffu);
        refill();
      }
      n -= m;
    }
    return v;
  }
  __attribute__((always_inline)) inline int terminate()
  {
    range -= 2;
    if (value >= ((uint64_t)range << bits)) return 1;
    if (range < 256) { range <<= 1; bits--; }
    refill();
    return 0;
  }
};
struct CabacDec : CabacRegs {
  const uint8_t *buf = nullptr;                                 // the substream
  uint16_t ctx_store[CTX_COUNT];
  CabacDec() { ctx = ctx_store; }
  CabacDec(const CabacDec &) = delete;
  void load_ctx(const uint8_t *src) { for (int i = 0; i < CTX_COUNT; i++) ctx_store[i] = src[i]; }
  void save_ctx(uint8_t *dst) const { for (int i = 0; i < CTX_COUNT; i++) dst[i] = (uint8_t)ctx_store[i]; }
  void start(const uint8_t *b, size_t l)
  {
    buf = b; p = b; end = b + l; past = 0; st = &state_tabs(); range = 510; ctx = ctx_store;
    value = word(); bits = 32 - 9;
    refill();
  }
  // bytes from the start of the substream up to and including the byte holding the last consumed bit (after a terminating bin == 1:
  // 9.3.2.5 reads rbsp_trailing / alignment, i.e. the arithmetic codeword ends at the byte boundary after the 7 bits it consumed last)
  size_t bytes_consumed() const { const size_t consumed_bits = (size_t)(p - buf) * 8 - (size_t)bits; return (consumed_bits + 7) >> 3; }
};

std::atomic<long> g_yields{0};
struct Tick { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
const CoreTabs *host_tabs()
{
  static const CoreTabs t = [] { CoreTabs t; for (int i = 0; i < 64; i++) core_tabs_fill_entry(t, i); return t; }();
  return &t;
}

// scan position -> (x, y) for the three scans and block sizes 1..8 (H.265 6.5.3-6.5.5)
struct ScanTabs { uint8_t x[3][4][64], y[3][4][64], inv[3][4][64], sigk[3][5][16]; };   // sigk[scan][prev_csbf, 4 = 4x4 block][scan position k] = context pattern; inv[scan][log2 of the grid][y << log2 | x] = scan position
const ScanTabs &scan_tabs()
{
  static const ScanTabs t = [] {
    ScanTabs t;
    const CoreTabs *ct = host_tabs();
    for (int sc = 0; sc < 3; sc++) for (int l2 = 0; l2 < 4; l2++) for (int i = 0; i < (1 << (2 * l2)); i++) {
      int x, y; scan_pos(ct, sc, l2, i, x, y); t.x[sc][l2][i] = (uint8_t)x; t.y[sc][l2][i] = (uint8_t)y; t.inv[sc][l2][(y << l2) | x] = (uint8_t)i;
    }
    for (int sc = 0; sc < 3; sc++) for (int k = 0; k < 16; k++) {
      for (int pc = 0; pc < 4; pc++) t.sigk[sc][pc][k] = ct->sigpat[pc][ct->pos4[sc][k]];
      t.sigk[sc][4][k] = ct->ctxmap4x4[ct->pos4[sc][k]];
    }
    return t;
  }();
  return t;
}

// sao() of one CTU (7.3.8.3); `left` / `up`: the neighbours that may be merged from
void parse_sao(CabacDec &c, SaoParams &p, const SaoParams *left, const SaoParams *up, bool luma, bool chroma)
{
  memset(&p, 0, sizeof(p));
  if (left && c.bin(CTX_SAO_MERGE)) { p = *left; return; }
  if (up && c.bin(CTX_SAO_MERGE)) { p = *up; return; }
  for (int ci = 0; ci < 3; ci++) {
    if (!(ci ? chroma : luma)) continue;
    if (ci < 2) p.type[ci] = (uint8_t)(c.bin(CTX_SAO_TYPE) ? (c.bypass() ? 2 : 1) : 0);
    else { p.type[2] = p.type[1]; p.eo_class[2] = p.eo_class[1]; }
    if (!p.type[ci]) continue;
    int a[4];
    for (int i = 0; i < 4; i++) { a[i] = 0; while (a[i] < 7 && c.bypass()) a[i]++; }
    if (p.type[ci] == 1) {
      for (int i = 0; i < 4; i++) if (a[i] && c.bypass()) a[i] = -a[i];
      p.band_pos[ci] = (uint8_t)c.bypass_bits(5);
    } else {
      if (ci < 2) p.eo_class[ci] = (uint8_t)c.bypass_bits(2);
      a[2] = -a[2]; a[3] = -a[3];                                  // edge offsets: categories 1, 2 positive, 3, 4 negative
    }
    for (int i = 0; i < 4; i++) p.offset[ci][i] = (int8_t)a[i];
  }
}

// residual_coding() (7.3.8.11): appends (raster position << 16 | level) words; *tskip receives transform_skip_flag
// The two loops that decode most of a picture's bins, as functions of their own: inside parse_residual the compiler has no registers left for the decoder's
// (x86-64: sixteen for a function with thirty live values) and keeps them on the stack -- a store and a load on every bin's dependency chain; here they are
// the only live state.
__attribute__((noinline)) uint32_t sig_flag_run(CabacRegs &cr, const uint8_t *pk, int base, int k0)       // sig_coeff_flag of scan positions k0 .. 1
{
  CabacRegs c = cr;
  uint32_t sig = 0;
  for (int k = k0; k >= 1; k--) sig |= (uint32_t)c.bin(base + pk[k]) << k;
  cr.p = c.p; cr.value = c.value; cr.bits = c.bits; cr.range = c.range; cr.past = c.past;
  return sig;
}
// coeff_abs_level_greater1_flag of the first (up to eight) coefficients of a sub-block: bit j of the result = flag of coefficient j; *c1_io: greater1Ctx
__attribute__((noinline)) uint32_t greater1_run(CabacRegs &cr, int ctx_base, int n, int *c1_io)
{
  CabacRegs c = cr;
  uint32_t g = 0; int c1 = *c1_io;
  for (int j = 0; j < n; j++) {
    const int g1 = c.bin(ctx_base + c1);
    g |= (uint32_t)g1 << j;
    if (g1) c1 = 0; else if (c1 > 0 && c1 < 3) c1++;
  }
  *c1_io = c1;
  cr.p = c.p; cr.value = c.value; cr.bits = c.bits; cr.range = c.range; cr.past = c.past;
  return g;
}

__attribute__((always_inline)) inline bool parse_residual_regs(CabacRegs &c, int log2, int cidx, int scan_idx, bool sign_hiding, bool ts_enabled, int *tskip, std::vector<uint32_t> &out)
{
  const ScanTabs &S = scan_tabs();
  const int n = 1 << log2, sbl = log2 - 2, nsb = 1 << sbl;
  const uint8_t *SX = S.x[scan_idx][sbl], *SY = S.y[scan_idx][sbl], *PX = S.x[scan_idx][2], *PY = S.y[scan_idx][2];
  uint8_t csbf[8][8]; memset(csbf, 0, sizeof(csbf));
  *tskip = (ts_enabled && log2 == 2) ? c.bin(CTX_TS_FLAG + (cidx ? 1 : 0)) : 0;
  int pre[2];
  for (int d = 0; d < 2; d++) {
    int off, sh, mx = (log2 << 1) - 1, v = 0;
    if (cidx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); sh = (log2 + 1) >> 2; } else { off = 15; sh = log2 - 2; }
    while (v < mx && c.bin((d ? CTX_LAST_Y : CTX_LAST_X) + off + (v >> sh))) v++;
    pre[d] = v;
  }
  int lx = pre[0], ly = pre[1];
  if (lx > 3) { int nb = (lx >> 1) - 1; lx = (1 << nb) * (2 + (lx & 1)) + (int)c.bypass_bits(nb); }
  if (ly > 3) { int nb = (ly >> 1) - 1; ly = (1 << nb) * (2 + (ly & 1)) + (int)c.bypass_bits(nb); }
  if (scan_idx == 2) { int tt = lx; lx = ly; ly = tt; }
  if (lx >= n || ly >= n) return false;
  // the last significant coefficient as (sub-block, position inside it) in scan order: inverse scan tables
  const int last_sb = S.inv[scan_idx][sbl][((ly >> 2) << sbl) | (lx >> 2)], last_pos = S.inv[scan_idx][2][((ly & 3) << 2) | (lx & 3)];
  int c1 = 1;
  for (int i = last_sb; i >= 0; i--) {
    const int xs = SX[i], ys = SY[i];
    int right = (xs < nsb - 1) ? csbf[ys][xs + 1] : 0, below = (ys < nsb - 1) ? csbf[ys + 1][xs] : 0, infer_dc = 0;
    if (i < last_sb && i > 0) { csbf[ys][xs] = (uint8_t)c.bin(CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0)); infer_dc = 1; }
    else csbf[ys][xs] = 1;
    if (!csbf[ys][xs]) continue;
    uint32_t sig = 0;
    if (i == last_sb) sig |= 1u << last_pos;
    const int prev_csbf = right | (below << 1);
    // sig_coeff_flag contexts (9.3.4.2.5) from tables: pattern by the neighbouring sub-blocks' flags and the position inside
    // the sub-block, plus an offset that is constant over the sub-block
    const uint8_t *pk = S.sigk[scan_idx][log2 == 2 ? 4 : prev_csbf];
    const int sig_base = CTX_SIG + (cidx ? 27 : 0);
    const int sig_off = log2 == 2 ? 0 : (cidx == 0 ? ((i > 0 ? 3 : 0) + ((log2 == 3) ? ((scan_idx == 0) ? 9 : 15) : 21)) : ((log2 == 3) ? 9 : 12));
    {
      const int k0 = (i == last_sb) ? last_pos - 1 : 15;
      const int base = sig_base + sig_off;
      if (k0 >= 1) sig |= sig_flag_run(c, pk, base, k0);      // (position 0 apart: no per-flag conditions in that loop)
      if (k0 >= 0) {
        if (infer_dc && !(sig >> 1)) sig |= 1u;            // every other flag of a coded sub-block zero: inferred
        else sig |= (uint32_t)c.bin((i == 0 && log2 != 2) ? sig_base : base + pk[0]);      // (the DC coefficient of the block has its own context)
      }
    }
    if (!sig) continue;
    int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
    if (c1 == 0) ctx_set++;
    c1 = 1;
    int pos[16], lev[16], nsig = 0, g1idx = -1;
    for (uint32_t m = sig; m;) { const int k = 31 - __builtin_clz(m); pos[nsig++] = k; m &= ~(1u << k); }     // highest scan position first
    for (int j = 0; j < nsig; j++) lev[j] = 1;
    {
      const uint32_t g = greater1_run(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4, nsig < 8 ? nsig : 8, &c1);
      if (g) { g1idx = __builtin_ctz(g); for (uint32_t m = g; m; m &= m - 1) lev[__builtin_ctz(m)] = 2; }
    }
    if (g1idx >= 0 && c.bin(CTX_GT2 + (cidx ? 4 : 0) + ctx_set)) lev[g1idx] = 3;
    // sign_data_hiding (7.3.8.11, 9.3.4.3): the sign of the sub-block's first coefficient in scan order is not sent when its
    // first and last significant positions are more than three apart; it follows from the parity of the sum of the levels
    const bool hidden = sign_hiding && (pos[0] - pos[nsig - 1] > 3);
    const int nsigns = hidden ? nsig - 1 : nsig;
    uint32_t signs = c.bypass_bits(nsigns) << (nsig - nsigns);
    int rice = 0, sum = 0;
    for (int j = 0; j < nsig; j++) {
      int base = (j < 8) ? ((j == g1idx) ? 3 : 2) : 1;
      if (lev[j] == base) {
        int prefix = 0;
        while (prefix < 32 && c.bypass()) prefix++;
        if (prefix - 3 + rice > 16) return false;              // (9.3.3.11: an 8-bit stream's escape suffix has at most 16 bits; anything longer is not a level, and the shifts below must not see it)
        int rem = prefix <= 3 ? (prefix << rice) + (int)c.bypass_bits(rice)
                              : (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)c.bypass_bits(prefix - 3 + rice);
        lev[j] = base + rem;
        if (lev[j] > 3 * (1 << rice)) rice = imin(rice + 1, 4);
      }
      sum += lev[j];
    }
    if (hidden && (sum & 1)) signs |= 1u;
    const size_t o0 = out.size();
    out.resize(o0 + (size_t)nsig);                       // (one capacity check per sub-block instead of one per level)
    uint32_t *dst = out.data() + o0;
    for (int j = 0; j < nsig; j++) {
      const int v = ((signs >> (nsig - 1 - j)) & 1) ? -lev[j] : lev[j];
      const int xp = PX[pos[j]], yp = PY[pos[j]];
      dst[j] = (uint32_t)((((ys << 2) + yp) * n + (xs << 2) + xp) << 16) | ((uint32_t)clip3(-32768, 32767, v) & 0xffffu);
    }
    if (c.overrun()) return false;
  }
  return !c.overrun();
}

bool parse_residual(CabacDec &cd, int log2, int cidx, int scan_idx, bool sign_hiding, bool ts_enabled, int *tskip, std::vector<uint32_t> &out)
{
  CabacRegs c = cd;                                        // the decoder's registers in locals for the whole block (CabacRegs)
  const bool ok = parse_residual_regs(c, log2, cidx, scan_idx, sign_hiding, ts_enabled, tskip, out);
  static_cast<CabacRegs &>(cd) = c;
  return ok;
}

// ------------------------------------------------------------------------------------------ slice data (7.3.8) of one substream
struct MvCand { int mvx, mvy, ref_idx; };

struct SliceParser {
  Decoder::PicJob &job; Decoder::SubOut &out;
  const DecSps &sps; const DecPps &pps; const Decoder::SliceHdr &sh;
  CabacDec c;
  const int w, h, b4w, b8w, ctbl, mincb, wc, hc;      // ctbl: CtbLog2SizeY; mincb: MinCbLog2SizeY (3, 4 or 5); wc, hc: the picture in coding tree blocks
  B4Rec *b4; uint8_t *pm, *ctd, *im;     // pm, ctd: per 8x8 (the minimum coding block); im: per 4x4 (NxN parts)
  int ref_y0 = -(1 << 30), ref_y1 = 1 << 30;                              // band mode: the luma rows of a reference picture this decoder holds (the picture's outer edges open)
  int tile_y0 = 0, tile_y1 = 1 << 30, tile_x0 = 0, tile_x1 = 1 << 30;    // luma rows / columns of the tile being parsed: nothing outside is available (other tiles may be parsed concurrently)
  const uint8_t *slice_of = nullptr; int cur_slice = 0;                  // pictures of several slices inside a tile (PicJob::ctb_slice): the slice of every coding tree block, the one being parsed
  int slice_qp = 26;                                                      // SliceQpY of the slice being parsed
  int err = 0;
  // quantisation (8.6.1)
  int qp_y = 0, qp_y_pred = 0, last_qp_y = 0, cu_qp_delta_val = 0, log2_qg = 6; bool qp_delta_coded = false;
  // coding unit being parsed
  int cu_pred_mode = 0, part_mode = 0, max_trafo_depth = 0, intra_modes[4] = {0, 0, 0, 0}, chroma_mode = 0; bool intra_split = false;
  uint32_t ctu_intra_mask = 0;

  SliceParser(Decoder::PicJob &j, Decoder::SubOut &o, int pw)
      : job(j), out(o), sps(*j.sps), pps(j.pps), sh(j.sh), w(j.sps->width), h(j.sps->height), b4w(pw / 4), b8w(pw / 8), ctbl(j.sps->ctb_log2), mincb(j.sps->min_cb_log2), wc((j.sps->width + (1 << j.sps->ctb_log2) - 1) >> j.sps->ctb_log2),
        hc((j.sps->height + (1 << j.sps->ctb_log2) - 1) >> j.sps->ctb_log2), b4(j.b4), pm(j.pred_mode.data()), ctd(j.ct_depth.data()), im(j.intra_mode.data())
  { log2_qg = ctbl - pps.qp_delta_depth; }

  inline int bi(int x, int y) const { return (y >> 2) * b4w + (x >> 2); }
  inline int b8(int x, int y) const { return (y >> 3) * b8w + (x >> 3); }
  // 6.4.1: inside the picture, inside the tile, in the same slice (slice_of: pictures with several slices inside a tile), already decoded.  "Already
  // decoded" is read off the prediction-mode array, which starts every picture as PM_NONE: in decoding order a block is marked
  // when its coding unit starts, and the WPP row hand-over (two CTUs behind the row above, parse_substream) guarantees that every
  // neighbour that precedes the current block in z-scan order has been parsed while none that follows it has been.
  inline bool avail(int, int, int xn, int yn) const
  {
    return xn >= tile_x0 && yn >= tile_y0 && xn < w && yn < h && yn < tile_y1 && xn < tile_x1 && pm[b8(xn, yn)] != PM_NONE && (!slice_of || slice_of[(yn >> ctbl) * wc + (xn >> ctbl)] == cur_slice);
  }
  // coding-unit wide values of the per-8x8 arrays
  void fill_cu8(uint8_t *arr, int x0, int y0, int n, int v)
  {
    const int cols = imin(n, w - x0) >> 3;
    for (int y = y0; y < y0 + n && y < h; y += 8) memset(arr + b8(x0, y), v, (size_t)cols);
  }
  void fill_u8(uint8_t *arr, int x0, int y0, int bw, int bh, int v)       // per-4x4 array
  {
    const int cols = imin(bw, w - x0) >> 2;
    for (int y = y0; y < y0 + bh && y < h; y += 4) memset(arr + bi(x0, y), v, (size_t)cols);
  }
  // one record for every 4x4 unit of a rectangle
  // cu_edges: the rectangle is a whole coding block -- its top row and left column get the coding block's edge flags in the same pass (a
  // read-modify-write of the column afterwards waits for every one of the stores just issued: it was the hottest line of the parser)
  bool cu_edges_done = false;
  int cu_bypass = 0;                                       // cu_transquant_bypass_flag of the coding unit being parsed
  void fill_recs(int x0, int y0, int bw, int bh, const B4Rec &r, bool cu_edges = false)
  {
    uint64_t v; memcpy(&v, &r, 8);
    const int cols = imin(bw, w - x0) >> 2;
    if (!cu_edges) { for (int y = y0; y < y0 + bh && y < h; y += 4) { uint64_t *p = (uint64_t *)&b4[bi(x0, y)]; for (int i = 0; i < cols; i++) p[i] = v; } return; }
    B4Rec e = r;
    e.flags |= B4_EDGE_V | B4_TU_V; uint64_t vl; memcpy(&vl, &e, 8);                    // first column
    e.flags = r.flags | B4_EDGE_H | B4_TU_H; uint64_t vt; memcpy(&vt, &e, 8);           // first row
    e.flags |= B4_EDGE_V | B4_TU_V; uint64_t vc; memcpy(&vc, &e, 8);                    // the corner
    for (int y = y0; y < y0 + bh && y < h; y += 4) {
      uint64_t *p = (uint64_t *)&b4[bi(x0, y)];
      const bool top = y == y0;
      if (cols > 0) p[0] = top ? vc : vl;
      for (int i = 1; i < cols; i++) p[i] = top ? vt : v;
    }
    cu_edges_done = true;
  }
  void emit_tu(const DecTu &td)
  {
    const int X = td.plane ? td.x * 2 : td.x, Y = td.plane ? td.y * 2 : td.y;
    TuRange &r = job.region[(Y >> 5) * (b4w >> 3) + (X >> 5)];      // (32x32 regions of the padded picture)
    if (!r.count) r.first = (uint32_t)out.tus.size();
    r.count++;
    out.tus.push_back(td);
  }

  // ---------------------------------------------------------------- motion vector prediction (8.5.3.2)
  bool pb_avail(int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int xn, int yn) const
  {
    const bool same_cb = xcb <= xn && ycb <= yn && xcb + ncbs > xn && ycb + ncbs > yn;
    bool a;
    if (!same_cb) a = avail(xpb, ypb, xn, yn);
    else a = !((npbw << 1) == ncbs && (npbh << 1) == ncbs && part_idx == 1 && (ycb + npbh <= yn) && (xcb + npbw > xn));
    if (a && pm[b8(xn, yn)] == PM_INTRA) a = false;
    return a;
  }
  static bool same_motion(const B4Rec &a, const B4Rec &b) { return a.ref_idx == b.ref_idx && a.mvx == b.mvx && a.mvy == b.mvy; }

  // temporal candidate of list X (8.5.3.2.8, 8.5.3.2.9).  The collocated block may be bi-predicted (a B picture): of its two vectors the one of list X
  // counts when no reference picture of this slice follows it in output order, else the one of list collocated_from_l0_flag.
  inline int list_poc(int L, int idx) const { return L ? job.ref_poc1[idx & 15] : job.ref_poc[idx & 15]; }
  inline bool list_lt(int L, int idx) const { return (L ? job.ref_lt1[idx & 15] : job.ref_lt[idx & 15]) != 0; }      // the entry is a long-term reference picture
  static void scale_by(int &mvx, int &mvy, int td_, int tb_)
  {
    const int td = clip3(-128, 127, td_), tb = clip3(-128, 127, tb_);
    if (td == 0) return;
    const int tx = (16384 + (iabs(td) >> 1)) / td, dsf = clip3(-4096, 4095, (tb * tx + 32) >> 6);
    const int px = dsf * mvx, py = dsf * mvy;
    mvx = clip3(-32768, 32767, (px < 0 ? -1 : 1) * ((iabs(px) + 127) >> 8));
    mvy = clip3(-32768, 32767, (py < 0 ? -1 : 1) * ((iabs(py) + 127) >> 8));
  }
  bool temporal_mv(int xpb, int ypb, int npbw, int npbh, int X, int ref_idx, int &mvx, int &mvy)
  {
    ColMotion *col = job.col.get();
    if (!col) return false;
    const int cy = ypb >> ctbl;
    if (cy < col->hc) {                                  // the collocated picture may still be in the hands of its own parser (frame threads)
      std::atomic<uint8_t> &d = col->row_done[(size_t)cy];
      int spins = 0;
      while (!d.load(std::memory_order_acquire)) { if (++spins < 2000) __builtin_ia32_pause(); else std::this_thread::yield(); }
    }
    const int cand[2][2] = {{xpb + npbw, ypb + npbh}, {xpb + (npbw >> 1), ypb + (npbh >> 1)}};
    for (int k = 0; k < 2; k++) {
      int x = cand[k][0], y = cand[k][1];
      if (k == 0 && !((ypb >> ctbl) == (y >> ctbl) && y < h && x < w)) continue;      // bottom right: same CTB row, inside the picture
      x >>= 4; y >>= 4;
      if (x >= col->w16 || y >= col->h16) continue;
      const ColMotion::Mv &m = col->mv[(size_t)y * col->w16 + x];
      if (!m.used) continue;
      const int L = m.used == 2 ? 1 : (m.used == 1 ? 0 : (job.no_backward ? X : sh.collocated_from_l0));
      const bool cur_lt = list_lt(X, ref_idx);
      if ((((m.lt >> L) & 1) != 0) != cur_lt) continue;      // 8.5.3.2.9: one of the two reference pictures long-term, the other not: no candidate from this block
      const int col_diff = col->poc - m.ref_poc[L], cur_diff = sh.poc - list_poc(X, ref_idx);
      mvx = m.mv[L][0]; mvy = m.mv[L][1];
      if (!cur_lt && col_diff != cur_diff && col_diff != 0) scale_by(mvx, mvy, col_diff, cur_diff);      // (long-term: taken as it is)
      return true;
    }
    return false;
  }

  // `want`: the index the bitstream chose -- only cand[want] is read afterwards (merge_idx 0 with the left neighbour there, the common case of a
  // skipped CU, needs one look-up instead of five and the pruning)
  void merge_candidates(int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int pmode, MvCand *cand, int want)
  {
    const int lvl = pps.par_mrg_level;
    int n = 0;
    if (lvl > 2 && ncbs == 8) { xpb = xcb; ypb = ycb; npbw = npbh = ncbs; part_idx = 0; pmode = PART_2Nx2N; }
    auto par = [&](int xn, int yn) { return ((xpb >> lvl) == (xn >> lvl)) && ((ypb >> lvl) == (yn >> lvl)); };
    const int xa1 = xpb - 1, ya1 = ypb + npbh - 1, xb1 = xpb + npbw - 1, yb1 = ypb - 1, xb0 = xpb + npbw, yb0 = ypb - 1;
    const int xa0 = xpb - 1, ya0 = ypb + npbh, xb2 = xpb - 1, yb2 = ypb - 1;
    const bool part1 = part_idx == 1;
    const bool nbA1 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa1, ya1) && !par(xa1, ya1) &&
                      !(part1 && (pmode == PART_Nx2N || pmode == PART_nLx2N || pmode == PART_nRx2N));
    if (want == 0 && nbA1) { const B4Rec &m = b4[bi(xa1, ya1)]; cand[0].mvx = m.mvx; cand[0].mvy = m.mvy; cand[0].ref_idx = m.ref_idx; return; }
    const bool nbB1 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb1, yb1) && !par(xb1, yb1) &&
                      !(part1 && (pmode == PART_2NxN || pmode == PART_2NxnU || pmode == PART_2NxnD));
    const bool nbB0 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb0, yb0) && !par(xb0, yb0);
    const bool nbA0 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa0, ya0) && !par(xa0, ya0);
    const bool nbB2 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb2, yb2) && !par(xb2, yb2);
    const B4Rec zero = B4Rec();
    const B4Rec &A1 = nbA1 ? b4[bi(xa1, ya1)] : zero, &B1 = nbB1 ? b4[bi(xb1, yb1)] : zero, &B0 = nbB0 ? b4[bi(xb0, yb0)] : zero;
    const B4Rec &A0 = nbA0 ? b4[bi(xa0, ya0)] : zero, &B2 = nbB2 ? b4[bi(xb2, yb2)] : zero;
    const bool avA1 = nbA1, avB1 = nbB1 && !(nbA1 && same_motion(A1, B1)), avB0 = nbB0 && !(nbB1 && same_motion(B1, B0));
    const bool avA0 = nbA0 && !(nbA1 && same_motion(A1, A0));
    const bool avB2 = nbB2 && !(nbA1 && same_motion(A1, B2)) && !(nbB1 && same_motion(B1, B2)) && ((int)avA0 + avA1 + avB0 + avB1 != 4);
    const int maxc = sh.max_merge;
    auto add = [&](const B4Rec &m) { if (n < maxc) { cand[n].mvx = m.mvx; cand[n].mvy = m.mvy; cand[n].ref_idx = m.ref_idx; n++; } };
    if (avA1) add(A1);
    if (avB1) add(B1);
    if (avB0) add(B0);
    if (avA0) add(A0);
    if (avB2) add(B2);
    if (n < maxc) { int tx, ty; if (temporal_mv(xpb, ypb, npbw, npbh, 0, 0, tx, ty)) { cand[n].mvx = tx; cand[n].mvy = ty; cand[n].ref_idx = 0; n++; } }
    for (int zi = 0; n < maxc; n++, zi++) { cand[n].mvx = cand[n].mvy = 0; cand[n].ref_idx = zi < sh.num_ref_idx ? zi : 0; }     // 8.5.3.2.5, P slices
  }

  void scale_mv(int &mvx, int &mvy, int ref_a, int ref_target) const
  {
    const int td = clip3(-128, 127, sh.poc - job.ref_poc[ref_a]), tb = clip3(-128, 127, sh.poc - job.ref_poc[ref_target]);
    if (td == 0) return;
    const int tx = (16384 + (iabs(td) >> 1)) / td, dsf = clip3(-4096, 4095, (tb * tx + 32) >> 6);
    const int px = dsf * mvx, py = dsf * mvy;
    mvx = clip3(-32768, 32767, (px < 0 ? -1 : 1) * ((iabs(px) + 127) >> 8));
    mvy = clip3(-32768, 32767, (py < 0 ? -1 : 1) * ((iabs(py) + 127) >> 8));
  }

  void amvp_candidates(int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int ref_idx, int cand[2][2])
  {
    const int xa[2] = {xpb - 1, xpb - 1}, ya[2] = {ypb + npbh, ypb + npbh - 1};                       // A0, A1
    const int xb[3] = {xpb + npbw, xpb + npbw - 1, xpb - 1}, yb[3] = {ypb - 1, ypb - 1, ypb - 1};     // B0, B1, B2
    bool avA[2], avB[3];
    for (int k = 0; k < 2; k++) avA[k] = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa[k], ya[k]);
    for (int k = 0; k < 3; k++) avB[k] = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb[k], yb[k]);
    const bool is_scaled = avA[0] || avA[1];
    bool flagA = false, flagB = false; int ax = 0, ay = 0, bx = 0, by = 0;
    const int target = job.ref_poc[ref_idx];
    const bool tlt = job.ref_lt[ref_idx & 15] != 0;          // (8.5.3.2.7 step 7: a vector into ANOTHER picture counts when that picture and the target are both long-term -- as it is -- or both short-term -- scaled)
    for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) { const B4Rec &m = b4[bi(xa[k], ya[k])]; if (job.ref_poc[m.ref_idx & 15] == target) { flagA = true; ax = m.mvx; ay = m.mvy; } }
    for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) { const B4Rec &m = b4[bi(xa[k], ya[k])]; if ((job.ref_lt[m.ref_idx & 15] != 0) != tlt) continue; flagA = true; ax = m.mvx; ay = m.mvy; if (!tlt) scale_mv(ax, ay, m.ref_idx & 15, ref_idx); }
    for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) { const B4Rec &m = b4[bi(xb[k], yb[k])]; if (job.ref_poc[m.ref_idx & 15] == target) { flagB = true; bx = m.mvx; by = m.mvy; } }
    if (!is_scaled && flagB) { flagA = true; ax = bx; ay = by; }
    if (!is_scaled) {
      flagB = false;
      for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) {
        const B4Rec &m = b4[bi(xb[k], yb[k])];
        if ((job.ref_lt[m.ref_idx & 15] != 0) != tlt) continue;
        flagB = true; bx = m.mvx; by = m.mvy;
        if (!tlt && job.ref_poc[m.ref_idx & 15] != target) scale_mv(bx, by, m.ref_idx & 15, ref_idx);
      }
    }
    int n = 0;
    if (flagA) { cand[n][0] = ax; cand[n][1] = ay; n++; }
    if (flagB && !(flagA && ax == bx && ay == by)) { cand[n][0] = bx; cand[n][1] = by; n++; }
    if (n < 2) { int tx, ty; if (temporal_mv(xpb, ypb, npbw, npbh, 0, ref_idx, tx, ty)) { cand[n][0] = tx; cand[n][1] = ty; n++; } }
    for (; n < 2; n++) cand[n][0] = cand[n][1] = 0;
  }

  // ---------------------------------------------------------------- B slices: the same derivations over two-list motion (job.mvf)
  typedef Decoder::PicJob::MvF MvF;
  MvF *mvf = nullptr;                                      // [ph / 4][pw / 4], B slices only
  static bool same_motion_b(const MvF &a, const MvF &b)
  {
    if (a.ref[0] != b.ref[0] || a.ref[1] != b.ref[1]) return false;
    if (a.ref[0] >= 0 && (a.mv[0][0] != b.mv[0][0] || a.mv[0][1] != b.mv[0][1])) return false;
    if (a.ref[1] >= 0 && (a.mv[1][0] != b.mv[1][0] || a.mv[1][1] != b.mv[1][1])) return false;
    return true;
  }
  // 8.5.3.2.2 - 8.5.3.2.5 for a B slice: spatial candidates, the temporal one for both lists, combined bi-predictive candidates, two-list zero candidates
  void merge_candidates_b(int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int pmode, MvF *cand)
  {
    const int lvl = pps.par_mrg_level;
    int n = 0;
    if (lvl > 2 && ncbs == 8) { xpb = xcb; ypb = ycb; npbw = npbh = ncbs; part_idx = 0; pmode = PART_2Nx2N; }
    auto par = [&](int xn, int yn) { return ((xpb >> lvl) == (xn >> lvl)) && ((ypb >> lvl) == (yn >> lvl)); };
    const int xa1 = xpb - 1, ya1 = ypb + npbh - 1, xb1 = xpb + npbw - 1, yb1 = ypb - 1, xb0 = xpb + npbw, yb0 = ypb - 1;
    const int xa0 = xpb - 1, ya0 = ypb + npbh, xb2 = xpb - 1, yb2 = ypb - 1;
    const bool part1 = part_idx == 1;
    const bool nbA1 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa1, ya1) && !par(xa1, ya1) &&
                      !(part1 && (pmode == PART_Nx2N || pmode == PART_nLx2N || pmode == PART_nRx2N));
    const bool nbB1 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb1, yb1) && !par(xb1, yb1) &&
                      !(part1 && (pmode == PART_2NxN || pmode == PART_2NxnU || pmode == PART_2NxnD));
    const bool nbB0 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb0, yb0) && !par(xb0, yb0);
    const bool nbA0 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa0, ya0) && !par(xa0, ya0);
    const bool nbB2 = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb2, yb2) && !par(xb2, yb2);
    MvF zero; memset(&zero, 0, sizeof(zero)); zero.ref[0] = zero.ref[1] = -1;
    const MvF &A1 = nbA1 ? mvf[bi(xa1, ya1)] : zero, &B1 = nbB1 ? mvf[bi(xb1, yb1)] : zero, &B0 = nbB0 ? mvf[bi(xb0, yb0)] : zero;
    const MvF &A0 = nbA0 ? mvf[bi(xa0, ya0)] : zero, &B2 = nbB2 ? mvf[bi(xb2, yb2)] : zero;
    const bool avA1 = nbA1, avB1 = nbB1 && !(nbA1 && same_motion_b(A1, B1)), avB0 = nbB0 && !(nbB1 && same_motion_b(B1, B0));
    const bool avA0 = nbA0 && !(nbA1 && same_motion_b(A1, A0));
    const bool avB2 = nbB2 && !(nbA1 && same_motion_b(A1, B2)) && !(nbB1 && same_motion_b(B1, B2)) && ((int)avA0 + avA1 + avB0 + avB1 != 4);
    const int maxc = sh.max_merge;
    auto add = [&](const MvF &m) { if (n < maxc) cand[n++] = m; };
    if (avA1) add(A1);
    if (avB1) add(B1);
    if (avB0) add(B0);
    if (avA0) add(A0);
    if (avB2) add(B2);
    if (n < maxc) {
      int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
      const bool f0 = temporal_mv(xpb, ypb, npbw, npbh, 0, 0, x0, y0), f1 = temporal_mv(xpb, ypb, npbw, npbh, 1, 0, x1, y1);
      if (f0 || f1) {
        MvF &t = cand[n++];
        t.mv[0][0] = (int16_t)x0; t.mv[0][1] = (int16_t)y0; t.ref[0] = f0 ? 0 : -1;
        t.mv[1][0] = (int16_t)x1; t.mv[1][1] = (int16_t)y1; t.ref[1] = f1 ? 0 : -1;
      }
    }
    if (n > 1 && n < maxc) {                               // 8.5.3.2.4: list-0 motion of one candidate with list-1 motion of another (Table 8-7), unless they are one prediction
      static const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
      const int norig = n;
      for (int comb = 0; comb < norig * (norig - 1) && n < maxc; comb++) {
        const MvF &p0 = cand[l0c[comb]], &p1 = cand[l1c[comb]];
        if (p0.ref[0] < 0 || p1.ref[1] < 0) continue;
        if (list_poc(0, p0.ref[0]) == list_poc(1, p1.ref[1]) && p0.mv[0][0] == p1.mv[1][0] && p0.mv[0][1] == p1.mv[1][1]) continue;
        MvF &t = cand[n++];
        t.mv[0][0] = p0.mv[0][0]; t.mv[0][1] = p0.mv[0][1]; t.ref[0] = p0.ref[0];
        t.mv[1][0] = p1.mv[1][0]; t.mv[1][1] = p1.mv[1][1]; t.ref[1] = p1.ref[1];
      }
    }
    const int nrefs = imin(sh.num_ref_idx, sh.num_ref_idx1);
    for (int zi = 0; n < maxc; n++, zi++) { MvF &t = cand[n]; memset(&t, 0, sizeof(t)); t.ref[0] = t.ref[1] = (int8_t)(zi < nrefs ? zi : 0); }
  }
  // 8.5.3.2.6 / 8.5.3.2.7 for list X of a B slice: a neighbour's vector into the target picture from either of its lists (its list X first), else any of
  // its vectors scaled by the ratio of the POC distances
  void amvp_candidates_b(int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int X, int ref_idx, int cand[2][2])
  {
    const int xa[2] = {xpb - 1, xpb - 1}, ya[2] = {ypb + npbh, ypb + npbh - 1};                       // A0, A1
    const int xb[3] = {xpb + npbw, xpb + npbw - 1, xpb - 1}, yb[3] = {ypb - 1, ypb - 1, ypb - 1};     // B0, B1, B2
    bool avA[2], avB[3];
    for (int k = 0; k < 2; k++) avA[k] = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa[k], ya[k]);
    for (int k = 0; k < 3; k++) avB[k] = pb_avail(xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb[k], yb[k]);
    const bool is_scaled = avA[0] || avA[1];
    bool flagA = false, flagB = false; int ax = 0, ay = 0, bx = 0, by = 0;
    const int target = list_poc(X, ref_idx), Y = X ^ 1;
    auto same_pic = [&](const MvF &m, int &vx, int &vy) {
      for (int L : {X, Y}) if (m.ref[L] >= 0 && list_poc(L, m.ref[L]) == target) { vx = m.mv[L][0]; vy = m.mv[L][1]; return true; }
      return false;
    };
    const bool tlt = list_lt(X, ref_idx);
    auto any_pic = [&](const MvF &m, int &vx, int &vy) {
      for (int L : {X, Y}) if (m.ref[L] >= 0 && list_lt(L, m.ref[L]) == tlt) {
        vx = m.mv[L][0]; vy = m.mv[L][1];
        const int poc = list_poc(L, m.ref[L]);
        if (!tlt && poc != target) scale_by(vx, vy, sh.poc - poc, sh.poc - target);
        return true;
      }
      return false;
    };
    for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) flagA = same_pic(mvf[bi(xa[k], ya[k])], ax, ay);
    for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) flagA = any_pic(mvf[bi(xa[k], ya[k])], ax, ay);
    for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) flagB = same_pic(mvf[bi(xb[k], yb[k])], bx, by);
    if (!is_scaled && flagB) { flagA = true; ax = bx; ay = by; }
    if (!is_scaled) {
      flagB = false;
      for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) flagB = any_pic(mvf[bi(xb[k], yb[k])], bx, by);
    }
    int n = 0;
    if (flagA) { cand[n][0] = ax; cand[n][1] = ay; n++; }
    if (flagB && !(flagA && ax == bx && ay == by)) { cand[n][0] = bx; cand[n][1] = by; n++; }
    if (n < 2) { int tx, ty; if (temporal_mv(xpb, ypb, npbw, npbh, X, ref_idx, tx, ty)) { cand[n][0] = tx; cand[n][1] = ty; n++; } }
    for (; n < 2; n++) cand[n][0] = cand[n][1] = 0;
  }
  int parse_ref_idx(int num_active)
  {
    int ref_idx = 0;
    const int mx = num_active - 1;
    while (ref_idx < mx && ref_idx < 2 && c.bin(CTX_REF_IDX + ref_idx)) ref_idx++;
    if (ref_idx == 2) while (ref_idx < mx && c.bypass()) ref_idx++;
    return ref_idx;
  }
  void parse_mvd(int &dx, int &dy)
  {
    const int g0x = c.bin(CTX_MVD_GT0), g0y = c.bin(CTX_MVD_GT0);
    const int g1x = g0x ? c.bin(CTX_MVD_GT1) : 0, g1y = g0y ? c.bin(CTX_MVD_GT1) : 0;
    dx = mvd_abs(g0x, g1x); if (g0x && c.bypass()) dx = -dx;
    dy = mvd_abs(g0y, g1y); if (g0y && c.bypass()) dy = -dy;
  }
  // prediction_unit() of a B slice (7.3.8.6): inter_pred_idc, a reference index / vector difference / predictor flag per used list
  void prediction_unit_b(int xcb, int ycb, int ncbs, int xp, int yp, int bw, int bh, int part_idx, bool skip, int *merge_out)
  {
    const int merge = skip ? 1 : c.bin(CTX_MERGE_FLAG);
    if (merge_out) *merge_out = merge;
    MvF m; memset(&m, 0, sizeof(m)); m.ref[0] = m.ref[1] = -1;
    if (merge) {
      int idx = 0;
      if (sh.max_merge > 1 && c.bin(CTX_MERGE_IDX)) { idx = 1; while (idx < sh.max_merge - 1 && c.bypass()) idx++; }
      MvF cand[5];
      merge_candidates_b(xcb, ycb, ncbs, xp, yp, bw, bh, part_idx, part_mode, cand);
      m = cand[idx];
      if (m.ref[0] >= 0 && m.ref[1] >= 0 && bw + bh == 12) { m.ref[1] = -1; m.mv[1][0] = m.mv[1][1] = 0; }      // 8x4 / 4x8 blocks are never bi-predicted
    } else {
      int idc;                                             // 0 PRED_L0, 1 PRED_L1, 2 PRED_BI (9.3.4.2: "both" is asked first, with the coding quadtree depth as context, unless the block is 8x4 / 4x8)
      if (bw + bh != 12 && c.bin(CTX_INTER_PRED_IDC + ctd[b8(xcb, ycb)])) idc = 2;
      else idc = c.bin(CTX_INTER_PRED_IDC + 4);
      for (int X = 0; X < 2; X++) {
        if (idc == 1 - X) continue;
        const int na = X ? sh.num_ref_idx1 : sh.num_ref_idx;
        const int ref_idx = na > 1 ? parse_ref_idx(na) : 0;
        int dx = 0, dy = 0;
        if (!(X == 1 && sh.mvd_l1_zero && idc == 2)) parse_mvd(dx, dy);
        const int mvp = c.bin(CTX_MVP_FLAG);
        if (ref_idx >= (X ? job.nref1 : job.nref)) { err = DEC_ERR_INVALID; return; }
        int cand[2][2];
        amvp_candidates_b(xcb, ycb, ncbs, xp, yp, bw, bh, part_idx, X, ref_idx, cand);
        m.mv[X][0] = (int16_t)(uint16_t)(cand[mvp][0] + dx); m.mv[X][1] = (int16_t)(uint16_t)(cand[mvp][1] + dy);      // 8.5.3.2.6: modulo 2^16
        m.ref[X] = (int8_t)ref_idx;
      }
    }
    if ((m.ref[0] < 0 && m.ref[1] < 0) || m.ref[0] >= job.nref || m.ref[1] >= job.nref1) { err = DEC_ERR_INVALID; return; }
    if (ref_y1 != (1 << 30) || ref_y0 != -(1 << 30)) { err = DEC_ERR_UNSUPPORTED; return; }      // (band mode is the split encoder's streams: P pictures)
    const int P = m.ref[0] >= 0 ? 0 : 1;                   // the list whose motion rides in the B4Rec
    const bool bi = m.ref[0] >= 0 && m.ref[1] >= 0;
    // explicit weights only where an entry the block uses differs from the defaults: ((p 2^d + 2^(d + 5)) >> (d + 6)) = (p + 32) >> 6 and the mean likewise
    const bool wtd = sh.weighted && (((m.ref[0] >= 0) && ((sh.wt_explicit >> m.ref[0]) & 1)) || ((m.ref[1] >= 0) && ((sh.wt_explicit >> (16 + m.ref[1])) & 1)));
    B4Rec r; r.mvx = m.mv[P][0]; r.mvy = m.mv[P][1]; r.ref_idx = m.ref[P]; r.flags = (uint8_t)((cu_bypass ? B4_BYPASS : 0) | (bi ? B4_BI : 0) | (wtd ? B4_WT : 0)); r.qp_y = (int8_t)qp_y;
    r.slot = P ? job.ref_slot1[m.ref[1]] : job.ref_slot[m.ref[0]];
    fill_recs(xp, yp, bw, bh, r, bw == ncbs && bh == ncbs);
    B4L1 x; x.mvx = m.mv[1][0]; x.mvy = m.mv[1][1]; x.slot = bi ? job.ref_slot1[m.ref[1]] : 0; x.pad[0] = (uint8_t)(P * 16 + m.ref[P]); x.pad[1] = (uint8_t)(bi ? 16 + m.ref[1] : 0); x.pad[2] = 0;
    const int cols = imin(bw, w - xp) >> 2;
    const bool ext = bi || wtd;                            // the block has an entry in b4x[]
    for (int y = yp; y < yp + bh && y < h; y += 4) { const int i0 = bi_(xp, y); for (int i = 0; i < cols; i++) { mvf[i0 + i] = m; if (ext) job.b4x[(size_t)(i0 + i)] = x; } }
    if (ext) if (!job.any_bi.load(std::memory_order_relaxed)) job.any_bi.store(1, std::memory_order_relaxed);
    if (bw != ncbs || bh != ncbs) {                          // prediction block edges inside the coding block (deblocking)
      for (int i = 0; i < bh && yp + i < h; i += 4) b4[bi_(xp, yp + i)].flags |= B4_EDGE_V;
      for (int i = 0; i < bw && xp + i < w; i += 4) b4[bi_(xp + i, yp)].flags |= B4_EDGE_H;
    }
  }
  inline int bi_(int x, int y) const { return (y >> 2) * b4w + (x >> 2); }      // (bi() under a name that does not collide with the local `bi`)

  int mvd_abs(int gt0, int gt1)
  {
    if (!gt0) return 0;
    if (!gt1) return 1;
    int k = 1, v = 0;
    while (k < 17 && c.bypass()) { v += 1 << k; k++; }           // (EG1 prefix: a vector difference fits 16 bits, 7.4.9.9 -- at most 15 ones follow the first order bit)
    if (k >= 17) { err = DEC_ERR_INVALID; return 0; }
    return v + (int)c.bypass_bits(k) + 2;
  }

  void prediction_unit(int xcb, int ycb, int ncbs, int xp, int yp, int bw, int bh, int part_idx, bool skip, int *merge_out)
  {
    if (sh.is_b) { prediction_unit_b(xcb, ycb, ncbs, xp, yp, bw, bh, part_idx, skip, merge_out); return; }
    const int merge = skip ? 1 : c.bin(CTX_MERGE_FLAG);
    if (merge_out) *merge_out = merge;
    int mvx, mvy, ref_idx = 0;
    if (merge) {
      int idx = 0;
      if (sh.max_merge > 1 && c.bin(CTX_MERGE_IDX)) { idx = 1; while (idx < sh.max_merge - 1 && c.bypass()) idx++; }
      MvCand cand[5];
      merge_candidates(xcb, ycb, ncbs, xp, yp, bw, bh, part_idx, part_mode, cand, idx);
      mvx = cand[idx].mvx; mvy = cand[idx].mvy; ref_idx = cand[idx].ref_idx;
    } else {
      if (sh.num_ref_idx > 1) {
        const int mx = sh.num_ref_idx - 1;
        while (ref_idx < mx && ref_idx < 2 && c.bin(CTX_REF_IDX + ref_idx)) ref_idx++;
        if (ref_idx == 2) while (ref_idx < mx && c.bypass()) ref_idx++;
      }
      const int g0x = c.bin(CTX_MVD_GT0), g0y = c.bin(CTX_MVD_GT0);
      const int g1x = g0x ? c.bin(CTX_MVD_GT1) : 0, g1y = g0y ? c.bin(CTX_MVD_GT1) : 0;
      int dx = mvd_abs(g0x, g1x); if (g0x && c.bypass()) dx = -dx;
      int dy = mvd_abs(g0y, g1y); if (g0y && c.bypass()) dy = -dy;
      const int mvp = c.bin(CTX_MVP_FLAG);
      int cand[2][2];
      amvp_candidates(xcb, ycb, ncbs, xp, yp, bw, bh, part_idx, ref_idx, cand);
      mvx = (int16_t)(uint16_t)(cand[mvp][0] + dx); mvy = (int16_t)(uint16_t)(cand[mvp][1] + dy);      // 8.5.3.2.6: modulo 2^16
    }
    if (ref_idx < 0 || ref_idx >= job.nref) { err = DEC_ERR_INVALID; ref_idx = 0; }
    if (ref_y1 != (1 << 30) || ref_y0 != -(1 << 30)) {        // band mode: the vector must stay inside this decoder's rows (luma 8-tap, chroma 4-tap windows)
      const int fy = mvy & 3, fc = mvy & 7;
      const int top = imin(yp + (mvy >> 2) - (fy ? 3 : 0), 2 * ((yp >> 1) + (mvy >> 3) - (fc ? 1 : 0)));
      const int bot = imax(yp + bh + (mvy >> 2) + (fy ? 4 : 0), 2 * ((yp >> 1) + (bh >> 1) + (mvy >> 3) + (fc ? 2 : 0)));
      if (top < ref_y0 || bot > ref_y1) err = DEC_ERR_UNSUPPORTED;
    }
    B4Rec r; r.mvx = (int16_t)mvx; r.mvy = (int16_t)mvy; r.ref_idx = (int8_t)ref_idx; r.flags = (uint8_t)((cu_bypass ? B4_BYPASS : 0) | ((sh.weighted && ((sh.wt_explicit >> ref_idx) & 1)) ? B4_WT : 0)); r.qp_y = (int8_t)qp_y; r.slot = job.ref_slot[ref_idx];
    fill_recs(xp, yp, bw, bh, r, bw == ncbs && bh == ncbs);
    if (r.flags & B4_WT) {                                       // explicit weights: the block's table entry rides where B pictures keep their second vectors
      B4L1 x; x.mvx = 0; x.mvy = 0; x.slot = 0; x.pad[0] = (uint8_t)ref_idx; x.pad[1] = x.pad[2] = 0;
      const int cols = imin(bw, w - xp) >> 2;
      for (int y = yp; y < yp + bh && y < h; y += 4) { const int i0 = bi(xp, y); for (int i = 0; i < cols; i++) job.b4x[(size_t)(i0 + i)] = x; }
      if (!job.any_bi.load(std::memory_order_relaxed)) job.any_bi.store(1, std::memory_order_relaxed);
    }
    if (bw != ncbs || bh != ncbs) {                          // prediction block edges inside the coding block (deblocking); the block's own are set by coding_unit
      for (int i = 0; i < bh && yp + i < h; i += 4) b4[bi(xp, yp + i)].flags |= B4_EDGE_V;
      for (int i = 0; i < bw && xp + i < w; i += 4) b4[bi(xp + i, yp)].flags |= B4_EDGE_H;
    }
  }

  // ---------------------------------------------------------------- transform tree (7.3.8.8 - 7.3.8.10)
  void transform_unit(int x0, int y0, int xbase, int ybase, int log2, int blk, int cbf_luma, int cbf_cb, int cbf_cr, int cbf_cb_parent, int cbf_cr_parent)
  {
    const bool intra = cu_pred_mode == PM_INTRA;
    const bool chroma_here = log2 > 2, chroma_parent = log2 == 2 && blk == 3;
    const int ccb = chroma_here ? cbf_cb : (chroma_parent ? cbf_cb_parent : 0), ccr = chroma_here ? cbf_cr : (chroma_parent ? cbf_cr_parent : 0);
    const bool cbf_chroma_any = log2 > 2 ? (cbf_cb || cbf_cr) : (cbf_cb_parent || cbf_cr_parent);
    if ((cbf_luma || cbf_chroma_any) && pps.cu_qp_delta && !qp_delta_coded) {
      int v = 0;
      while (v < 5 && c.bin(CTX_CU_QP_DELTA + (v ? 1 : 0))) v++;
      if (v == 5) { int k = 0; while (k < 16 && c.bypass()) { v += 1 << k; k++; } if (k >= 16) { err = DEC_ERR_INVALID; return; } v += (int)c.bypass_bits(k); }
      if (v && c.bypass()) v = -v;
      if (v < -26 || v > 25) { err = DEC_ERR_INVALID; return; }
      qp_delta_coded = true; cu_qp_delta_val = v;
      qp_y = (qp_y_pred + v + 52) % 52;
    }
    const int n = 1 << log2;
    DecTu td; td.pad = 0;
    const int lmode = intra ? im[bi(x0, y0)] : 0;
    if (intra || cbf_luma) {
      td.x = (uint16_t)x0; td.y = (uint16_t)y0; td.plane = 0; td.log2 = (uint8_t)log2; td.mode = (uint8_t)lmode; td.qp = (int8_t)qp_y;
      td.flags = (uint8_t)((intra ? TU_INTRA : 0) | ((intra && log2 == 2) ? TU_DST : 0) | (cu_bypass ? TU_BYPASS : 0));
      td.offset = (uint32_t)out.levels.size(); td.count = 0;
      if (cbf_luma) {
        int ts;
        if (!parse_residual(c, log2, 0, intra_scan_idx(intra, log2, 0, lmode), pps.sign_hiding != 0 && !cu_bypass, pps.tskip != 0 && !cu_bypass, &ts, out.levels)) { err = DEC_ERR_INVALID; return; }
        td.count = (uint16_t)(out.levels.size() - td.offset);
        if (ts) td.flags |= TU_TSKIP;
        for (int y = y0; y < y0 + n && y < h; y += 4) for (int x = x0; x < x0 + n && x < w; x += 4) b4[bi(x, y)].flags |= B4_NZ;
      }
      emit_tu(td);
      if (intra) ctu_intra_mask |= 1u;
    }
    if (chroma_here || chroma_parent) {
      const int cx = (chroma_here ? x0 : xbase) >> 1, cy = (chroma_here ? y0 : ybase) >> 1, clog2 = chroma_here ? log2 - 1 : 2;
      for (int ci = 1; ci <= 2; ci++) {
        const int cbf = ci == 1 ? ccb : ccr;
        if (!intra && !cbf) continue;
        td.x = (uint16_t)cx; td.y = (uint16_t)cy; td.plane = (uint8_t)ci; td.log2 = (uint8_t)clog2; td.mode = (uint8_t)chroma_mode;
        td.qp = (int8_t)kChromaQp[clip3(0, 57, qp_y + (ci == 1 ? sh.cb_qp_offset : sh.cr_qp_offset))];
        td.flags = (uint8_t)((intra ? TU_INTRA : 0) | (cu_bypass ? TU_BYPASS : 0));
        td.offset = (uint32_t)out.levels.size(); td.count = 0;
        if (cbf) {
          int ts;
          if (!parse_residual(c, clog2, ci, intra_scan_idx(intra, clog2, ci, chroma_mode), pps.sign_hiding != 0 && !cu_bypass, pps.tskip != 0 && !cu_bypass, &ts, out.levels)) { err = DEC_ERR_INVALID; return; }
          td.count = (uint16_t)(out.levels.size() - td.offset);
          if (ts) td.flags |= TU_TSKIP;
        }
        emit_tu(td);
        if (intra) ctu_intra_mask |= 1u << ci;
      }
    }
  }

  void transform_tree(int x0, int y0, int xbase, int ybase, int log2, int depth, int blk, int cbf_cb_parent, int cbf_cr_parent)
  {
    if (err) return;
    int split;
    if (log2 <= 5 && log2 > 2 && depth < max_trafo_depth && !(intra_split && depth == 0)) split = c.bin(CTX_SPLIT_TRANSFORM + 5 - log2);
    else {
      const bool inter_split = sps.th_depth_inter == 0 && cu_pred_mode == PM_INTER && part_mode != PART_2Nx2N && depth == 0;
      split = (log2 > imin(5, ctbl) || (intra_split && depth == 0) || inter_split) ? 1 : 0;      // (MaxTbLog2SizeY = min(5, CtbLog2SizeY): checked against the SPS)
    }
    int cbf_cb = 0, cbf_cr = 0;
    if (log2 > 2) {
      if (depth == 0 || cbf_cb_parent) cbf_cb = c.bin(CTX_CBF_CHROMA + depth);
      if (depth == 0 || cbf_cr_parent) cbf_cr = c.bin(CTX_CBF_CHROMA + depth);
    } else { cbf_cb = cbf_cb_parent; cbf_cr = cbf_cr_parent; }
    if (split) {
      if (log2 <= 2) { err = DEC_ERR_INVALID; return; }
      const int hh = 1 << (log2 - 1);
      transform_tree(x0, y0, x0, y0, log2 - 1, depth + 1, 0, cbf_cb, cbf_cr);
      transform_tree(x0 + hh, y0, x0, y0, log2 - 1, depth + 1, 1, cbf_cb, cbf_cr);
      transform_tree(x0, y0 + hh, x0, y0, log2 - 1, depth + 1, 2, cbf_cb, cbf_cr);
      transform_tree(x0 + hh, y0 + hh, x0, y0, log2 - 1, depth + 1, 3, cbf_cb, cbf_cr);
    } else {
      int cbf_luma = 1;
      if (cu_pred_mode == PM_INTRA || depth != 0 || cbf_cb || cbf_cr) cbf_luma = c.bin(CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
      if (depth > 0) {                                     // transform block edges inside the coding block (deblocking)
        const int n = 1 << log2;
        for (int i = 0; i < n; i += 4) {
          if (y0 + i < h) b4[bi(x0, y0 + i)].flags |= B4_EDGE_V | B4_TU_V;
          if (x0 + i < w) b4[bi(x0 + i, y0)].flags |= B4_EDGE_H | B4_TU_H;
        }
      }
      transform_unit(x0, y0, xbase, ybase, log2, blk, cbf_luma, log2 > 2 ? cbf_cb : 0, log2 > 2 ? cbf_cr : 0, cbf_cb_parent, cbf_cr_parent);
    }
  }

  // ---------------------------------------------------------------- coding unit (7.3.8.5)
  void coding_unit(int x0, int y0, int log2cb, int depth)
  {
    const int n = 1 << log2cb;
    int skip = 0;
    cu_bypass = pps.tq_bypass ? c.bin(CTX_TQ_BYPASS) : 0;      // cu_transquant_bypass_flag (7.3.8.5: first in the coding unit)
    if (!sh.is_intra) {
      const int l = avail(x0, y0, x0 - 1, y0) && pm[b8(x0 - 1, y0)] == PM_SKIP, a = avail(x0, y0, x0, y0 - 1) && pm[b8(x0, y0 - 1)] == PM_SKIP;
      skip = c.bin(CTX_SKIP + l + a);
    }
    part_mode = PART_2Nx2N; intra_split = false; cu_edges_done = false;
    int rqt_root_cbf = 1, merge_2nx2n = 0;
    fill_cu8(ctd, x0, y0, n, depth);
    qp_y = (qp_y_pred + cu_qp_delta_val + 52) % 52;          // CuQpDeltaVal of the quantisation group so far
    if (skip) {
      cu_pred_mode = PM_INTER;
      fill_cu8(pm, x0, y0, n, PM_SKIP);
      prediction_unit(x0, y0, n, x0, y0, n, n, 0, true, nullptr);
      rqt_root_cbf = 0;
      if (!job.any_inter) job.any_inter = true;
    } else {
      cu_pred_mode = PM_INTRA;
      if (!sh.is_intra) cu_pred_mode = c.bin(CTX_PRED_MODE) ? PM_INTRA : PM_INTER;
      if (cu_pred_mode != PM_INTRA || log2cb == mincb) {
        if (cu_pred_mode == PM_INTRA) part_mode = c.bin(CTX_PART_MODE) ? PART_2Nx2N : PART_NxN;
        else if (c.bin(CTX_PART_MODE)) part_mode = PART_2Nx2N;
        else if (log2cb == mincb) {                          // 9.3.3.7 at the minimum size: 01 2NxN, 00 Nx2N at 8x8 (no NxN there); above it 01, 001, 000 = NxN
          if (c.bin(CTX_PART_MODE + 1)) part_mode = PART_2NxN;
          else if (log2cb == 3) part_mode = PART_Nx2N;
          else part_mode = c.bin(CTX_PART_MODE + 2) ? PART_Nx2N : PART_NxN;
        }
        else if (!sps.amp) part_mode = c.bin(CTX_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
        else {
          const int horiz = c.bin(CTX_PART_MODE + 1);
          if (c.bin(CTX_PART_MODE + 3)) part_mode = horiz ? PART_2NxN : PART_Nx2N;
          else { const int b = c.bypass(); part_mode = horiz ? (b ? PART_2NxnD : PART_2NxnU) : (b ? PART_nRx2N : PART_nLx2N); }
        }
      }
      fill_cu8(pm, x0, y0, n, cu_pred_mode);
      if (cu_pred_mode == PM_INTRA && part_mode == PART_2Nx2N && sps.pcm_depth[0] && log2cb >= sps.pcm_min_log2 && log2cb <= sps.pcm_max_log2 && c.terminate()) {
        // pcm_flag = 1 (7.3.8.5, 7.3.8.7): the arithmetic codeword has ended; zero bits to the byte boundary, the samples at their bit depths, and the arithmetic
        // decoder starts again behind them with the contexts as they are (9.3.2.5).  For the kernels the unit is an intra unit of one transform block per plane
        // whose "levels" ARE the samples (shifted up to 8 bits) -- the residual path of cu_transquant_bypass_flag -- over a prediction of zero (mode 35: none);
        // for its neighbours its mode is DC (8.4.2); pcm_loop_filter_disabled_flag keeps the loop filters off it the way the bypass flag does.
        const uint8_t *q = c.buf + c.bytes_consumed();
        const size_t need = ((size_t)n * n * sps.pcm_depth[0] + (size_t)n * n / 2 * sps.pcm_depth[1]) / 8;
        if (q > c.end || need > (size_t)(c.end - q)) { err = DEC_ERR_INVALID; return; }
        BitReader pr(q, need);
        for (int ci = 0; ci < 3; ci++) {
          const int shp = ci ? 1 : 0, m = n >> shp, depth = sps.pcm_depth[ci ? 1 : 0];
          DecTu td; td.pad = 0; td.x = (uint16_t)(x0 >> shp); td.y = (uint16_t)(y0 >> shp); td.plane = (uint8_t)ci; td.log2 = (uint8_t)(log2cb - shp); td.mode = 35; td.qp = 0;
          td.flags = (uint8_t)(TU_INTRA | TU_BYPASS); td.offset = (uint32_t)out.levels.size();
          for (int y = 0; y < m; y++)
            for (int x = 0; x < m; x++) { const uint32_t v = pr.get(depth) << (8 - depth); if (v) out.levels.push_back((uint32_t)((y * m + x) << 16) | v); }
          td.count = (uint16_t)(out.levels.size() - td.offset);
          emit_tu(td);
          ctu_intra_mask |= 1u << ci;
        }
        c.start(q + need, (size_t)(c.end - (q + need)));
        fill_u8(im, x0, y0, n, n, 1);
        B4Rec r; r.mvx = 0; r.mvy = 0; r.ref_idx = -1; r.flags = (uint8_t)((cu_bypass || sps.pcm_no_filter) ? B4_BYPASS : 0); r.qp_y = (int8_t)qp_y; r.slot = 0;
        fill_recs(x0, y0, n, n, r, true);
        if (!job.any_intra) job.any_intra = true;
        rqt_root_cbf = 0;
      } else
      if (cu_pred_mode == PM_INTRA) {
        intra_split = part_mode == PART_NxN;
        const int parts = intra_split ? 2 : 1, pb = n / parts;
        int prev[4], k = 0;
        for (int j = 0; j < parts * parts; j++) prev[j] = c.bin(CTX_PREV_INTRA);
        for (int j = 0; j < parts; j++)
          for (int i = 0; i < parts; i++, k++) {
            const int xp = x0 + i * pb, yp = y0 + j * pb;
            int ca = 1, cb = 1;                                   // 8.4.2 candidate modes
            if (avail(xp, yp, xp - 1, yp) && pm[b8(xp - 1, yp)] == PM_INTRA) ca = im[bi(xp - 1, yp)];
            if (avail(xp, yp, xp, yp - 1) && pm[b8(xp, yp - 1)] == PM_INTRA && (yp - 1) >= ((yp >> ctbl) << ctbl)) cb = im[bi(xp, yp - 1)];
            int cand[3];
            if (ca == cb) {
              if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
              else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
            } else {
              cand[0] = ca; cand[1] = cb;
              cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26);
            }
            int mode;
            if (prev[k]) { int idx = 0; if (c.bypass()) { idx = 1; if (c.bypass()) idx = 2; } mode = cand[idx]; }
            else {
              mode = (int)c.bypass_bits(5);
              int t;
              if (cand[0] > cand[1]) { t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
              if (cand[0] > cand[2]) { t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
              if (cand[1] > cand[2]) { t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
              for (int q = 0; q < 3; q++) if (mode >= cand[q]) mode++;
            }
            intra_modes[k] = mode;
            fill_u8(im, xp, yp, pb, pb, mode);
          }
        int icpm = 4;
        if (c.bin(CTX_CHROMA_MODE)) icpm = (int)c.bypass_bits(2);
        static const int cm[4] = {0, 26, 10, 1};
        if (icpm == 4) chroma_mode = intra_modes[0];
        else { chroma_mode = cm[icpm]; if (chroma_mode == intra_modes[0]) chroma_mode = 34; }
        B4Rec r; r.mvx = 0; r.mvy = 0; r.ref_idx = -1; r.flags = (uint8_t)(cu_bypass ? B4_BYPASS : 0); r.qp_y = (int8_t)qp_y; r.slot = 0;
        fill_recs(x0, y0, n, n, r, true);
        if (!job.any_intra) job.any_intra = true;
      } else {
        const int hh = n / 2, q = n / 4; int mf = 0;
        switch (part_mode) {
          case PART_2Nx2N: prediction_unit(x0, y0, n, x0, y0, n, n, 0, false, &merge_2nx2n); break;
          case PART_2NxN: prediction_unit(x0, y0, n, x0, y0, n, hh, 0, false, &mf); prediction_unit(x0, y0, n, x0, y0 + hh, n, hh, 1, false, &mf); break;
          case PART_Nx2N: prediction_unit(x0, y0, n, x0, y0, hh, n, 0, false, &mf); prediction_unit(x0, y0, n, x0 + hh, y0, hh, n, 1, false, &mf); break;
          case PART_2NxnU: prediction_unit(x0, y0, n, x0, y0, n, q, 0, false, &mf); prediction_unit(x0, y0, n, x0, y0 + q, n, n - q, 1, false, &mf); break;
          case PART_2NxnD: prediction_unit(x0, y0, n, x0, y0, n, n - q, 0, false, &mf); prediction_unit(x0, y0, n, x0, y0 + n - q, n, q, 1, false, &mf); break;
          case PART_nLx2N: prediction_unit(x0, y0, n, x0, y0, q, n, 0, false, &mf); prediction_unit(x0, y0, n, x0 + q, y0, n - q, n, 1, false, &mf); break;
          case PART_NxN:                                       // (a minimum coding block above 8 samples: four square prediction blocks in z-order)
            for (int k = 0; k < 4; k++) prediction_unit(x0, y0, n, x0 + (k & 1) * hh, y0 + (k >> 1) * hh, hh, hh, k, false, &mf);
            break;
          default: prediction_unit(x0, y0, n, x0, y0, n - q, n, 0, false, &mf); prediction_unit(x0, y0, n, x0 + n - q, y0, q, n, 1, false, &mf); break;     // nRx2N
        }
        if (!(part_mode == PART_2Nx2N && merge_2nx2n)) rqt_root_cbf = c.bin(CTX_RQT_ROOT_CBF);
        if (!job.any_inter) job.any_inter = true;
      }
    }
    if (err) return;
    if (!cu_edges_done) {                                  // coding block edges are transform and prediction edges (a block of one prediction block: written with its records)
      B4Rec *const r0 = &b4[bi(x0, y0)];
      const int rows = (imin(n, h - y0) + 3) >> 2, cols = (imin(n, w - x0) + 3) >> 2;
      for (int i = 0; i < rows; i++) r0[(size_t)i * b4w].flags |= B4_EDGE_V | B4_TU_V;
      for (int i = 0; i < cols; i++) r0[i].flags |= B4_EDGE_H | B4_TU_H;
    }
    const int qp_before = qp_y;
    if (rqt_root_cbf) {
      max_trafo_depth = cu_pred_mode == PM_INTRA ? sps.th_depth_intra + (intra_split ? 1 : 0) : sps.th_depth_inter;
      transform_tree(x0, y0, x0, y0, log2cb, 0, 0, 0, 0);
    }
    if (qp_y != qp_before)                                   // a cu_qp_delta arrived inside this unit: its QpY is the new one (8.6.1)
      for (int y = y0; y < y0 + n && y < h; y += 4) for (int x = x0; x < x0 + n && x < w; x += 4) b4[bi(x, y)].qp_y = (int8_t)qp_y;
    last_qp_y = qp_y;
  }

  void coding_quadtree(int x0, int y0, int log2cb, int depth)
  {
    if (err) return;
    const int n = 1 << log2cb;
    int split;
    if (x0 + n <= w && y0 + n <= h && log2cb > mincb) {
      const int l = avail(x0, y0, x0 - 1, y0) && ctd[b8(x0 - 1, y0)] > depth, a = avail(x0, y0, x0, y0 - 1) && ctd[b8(x0, y0 - 1)] > depth;
      split = c.bin(CTX_SPLIT_CU + l + a);
    } else split = log2cb > mincb;
    if (pps.cu_qp_delta && log2cb >= log2_qg) {            // a quantisation group starts here (7.3.8.4, 8.6.1)
      qp_delta_coded = false; cu_qp_delta_val = 0;
      int qa = last_qp_y, qb = last_qp_y;
      if (avail(x0, y0, x0 - 1, y0) && ((x0 - 1) >> ctbl) == (x0 >> ctbl)) qa = b4[bi(x0 - 1, y0)].qp_y;
      if (avail(x0, y0, x0, y0 - 1) && ((y0 - 1) >> ctbl) == (y0 >> ctbl)) qb = b4[bi(x0, y0 - 1)].qp_y;
      qp_y_pred = (qa + qb + 1) >> 1;
    }
    if (split) {
      const int hh = n >> 1;
      coding_quadtree(x0, y0, log2cb - 1, depth + 1);
      if (x0 + hh < w) coding_quadtree(x0 + hh, y0, log2cb - 1, depth + 1);
      if (y0 + hh < h) coding_quadtree(x0, y0 + hh, log2cb - 1, depth + 1);
      if (x0 + hh < w && y0 + hh < h) coding_quadtree(x0 + hh, y0 + hh, log2cb - 1, depth + 1);
    } else coding_unit(x0, y0, log2cb, depth);
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------ lifecycle
// Pictures still being parsed by frame workers are waited for and dropped (close / resolution change).
void Decoder::drop_pending()
{
  if (gpu_job_ || !gpu_q_.empty()) { sync_main(); if (stream_dl_) hipStreamSynchronize(stream_dl_); }
  if (gpu_job_) { gpu_job_->ev_used = 0; gpu_job_->dl_buf = -1; gpu_job_ = nullptr; }
  for (PicJob *j : gpu_q_) { j->ev_used = 0; j->dl_buf = -1; }
  gpu_q_.clear();
  for (; job_tail_ != job_head_; job_tail_++) {
    PicJob &job = jobs_[(size_t)(job_tail_ % jobs_.size())];
    while (job.state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
    job.state.store(0, std::memory_order_relaxed);
  }
}

void Decoder::sync_main()
{
  if (batch_used_) { DecBatcher::get(device_).drain(this); batch_used_ = false; }
  hipStreamSynchronize(stream_);
  if (stream_alt_) hipStreamSynchronize(stream_alt_);
}

Decoder::~Decoder()
{
  if (getenv("KVAZZUP_AMD_TRACE")) fprintf(stderr, "kvazzup_amd decoder thread ms: nal %.1f  wait_parse %.1f  stage %.1f  gpu_api %.1f  gpu_sync %.1f  longest parse %.2f  (pictures %ld)\n", t_nal_, t_wait_, t_stage_, t_api_, t_sync_, t_parse_max_, job_tail_);
  if (parse_only_) { for (auto &j : jobs_) free(j.h_in); return; }
  drop_pending();
  workers_.reset();
  if (stream_) sync_main();
  if (batch_attached_) { DecBatcher::get(device_).detach(); batch_attached_ = false; }
  if (h_err_ && *h_err_) fprintf(stderr, "kvazzup_amd: decoder device error flags 0x%x (last picture)\n", *h_err_);
  for (auto &j : jobs_) { for (auto &e : j.ev) { hipEventDestroy(e.a); hipEventDestroy(e.b); } if (j.done) hipEventDestroy(j.done); if (j.dl_done) hipEventDestroy(j.dl_done); }
  free_buffers();
  free_retired(true);
  for (auto &o : ready_q_) owned_free(o.dev);
  for (auto &w : reorder_q_) owned_free(w.pic.dev);
  owned_free(cur_owned_.dev);
  for (auto &b : owned_pool_) owned_free(b.second);
  owned_pool_.clear();
  if (owned_ev_) hipEventDestroy(owned_ev_);
  if (stream_dl_ != stream_up_) stream_release(stream_dl_, device_, 'L', 'l');
  stream_release(stream_up_, device_, 'U', prio_up_);
  for (auto &e : up_done_) if (e) hipEventDestroy(e);
  if (h_err_) hipHostFree(h_err_);
  if (stream_alt_) stream_release(stream_alt_, device_, 'E', alt_prio_);
  stream_release(stream_, device_, 'D', prio_);
}

bool Decoder::start(std::string *error)
{
  if (parse_only_) { frame_threads_ = 1; started_ = true; return true; }      // (set_parse_only: the host half alone, no device)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_) {
    if (error) *error = "no usable HIP device (this library has no CPU fallback)";
    return false;
  }
  HIP_TRY(hipSetDevice(device_));
  spin_wait_ = getenv("KVAZZUP_AMD_SPIN") != nullptr;
  {
    const char *prio = getenv("KVAZZUP_AMD_PRIO");
    prio_ = (prio && strlen(prio) >= 4) ? prio[3] : 'n';
    HIP_TRY(stream_acquire(&stream_, device_, 'D', prio_));        // (stream_pool.h: a re-created decoder gets its predecessor's streams)
  }
  {
    const char *prio = getenv("KVAZZUP_AMD_PRIO");          // (letters 5 and 6: the download and the upload stream)
    prio_dl_ = (prio && strlen(prio) >= 5) ? prio[4] : 'n'; prio_up_ = (prio && strlen(prio) >= 6) ? prio[5] : 'n';
  }
  // ONE transfer stream: the input blocks go up and the finished pictures come down on it, all through the copy engine and none of them ever
  // waiting inside the queue (a download is only queued once its picture's kernels are known to be done).  HIP spreads the streams of a priority
  // level over four hardware queues; with encoder and decoder in one process every further stream shares a queue with one that matters.
  HIP_TRY(stream_acquire(&stream_up_, device_, 'U', prio_up_));
  stream_dl_ = stream_up_;
  if (const char *e = getenv("KVAZZUP_AMD_DL")) if (!strcmp(e, "own")) HIP_TRY(stream_acquire(&stream_dl_, device_, 'L', 'l'));   // experiment: downloads on a stream of their own at the lowest priority level (its own pool of hardware queues)
  for (auto &e : up_done_) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // the kernels' error word (a wavefront that gave up waiting ORs a flag in) lives in host memory the device writes straight into, like the
  // encoder's: looked at when a picture completes, no copy (a copy queued on the download stream waited behind the pictures' own downloads)
  HIP_TRY(hipHostMalloc(&h_err_, sizeof(uint32_t), hipHostMallocMapped));
  *h_err_ = 0;
  { void *dp = nullptr; HIP_TRY(hipHostGetDevicePointer(&dp, h_err_, 0)); err_ = (uint32_t *)dp; }
  DecBatcher::get(device_).attach(stream_); batch_attached_ = true;
  started_ = true;
  return true;
}

void Decoder::free_buffers()
{
  for (auto &j : jobs_) { if (j.h_in) hipHostFree(j.h_in); j.h_in = nullptr; j.h_in_cap = 0; j.col.reset(); j.own.reset(); }
  if (stream_dl_) hipStreamSynchronize(stream_dl_);
  for (auto &p : h_out_) { if (p) retired_out_.emplace_back(nal_calls_, p); p = nullptr; }      // (freed kOutHold calls later: free_retired)
  for (auto &p : d_in_) { hipFree(p); p = nullptr; }
  for (auto &c : d_in_cap_) c = 0;
  hipFree(progress_); hipFree(edge_col_); edge_col_ = nullptr; hipFree(edge_row_); edge_row_ = nullptr; hipFree(intra_order_); intra_order_ = nullptr;
  if (stream_alt_) {
    hipStreamSynchronize(stream_alt_);
    hipFree(progress_alt_); progress_alt_ = nullptr; hipFree(edge_col_alt_); edge_col_alt_ = nullptr; hipFree(edge_row_alt_); edge_row_alt_ = nullptr;
    for (int c = 0; c < 3; c++) { hipFree(resid_alt_[c]); resid_alt_[c] = nullptr; hipFree(work_alt_[c]); work_alt_[c] = nullptr; }
    stream_release(stream_alt_, device_, 'E', alt_prio_); stream_alt_ = nullptr;
  }
  alt_failed_ = false;                                       // (another picture size: the second chain's arrays may fit now)
  for (auto &p : dpb_) { hipFree(p.plane[0]); p = DpbPic(); }      // (a buffer's three planes are one allocation)
  for (int c = 0; c < 3; c++) { hipFree(work_[c]); work_[c] = nullptr; hipFree(resid_[c]); resid_[c] = nullptr; }
  progress_ = nullptr;
  w_ = h_ = pw_ = ph_ = 0;
}

// host views into a job's input block
void Decoder::bind_job(PicJob &job)
{
  uint8_t *p = job.h_in;
  job.b4 = (B4Rec *)p; job.region = (TuRange *)(p + off_region()); job.ctu = (TuRange *)(p + off_ctu());
  job.ctu_tile = p + off_tile(); job.sao = (SaoParams *)(p + off_sao());
}

// pinned input block of a job: at least `bytes`; the fixed part written so far is kept.  Called from the
// thread that owns the job (decoder thread at allocation, the job's parse worker later).
bool Decoder::grow_job_input(PicJob &job, size_t bytes)
{
  if (bytes <= job.h_in_cap) return true;
  const size_t cap = bytes + bytes / 2;
  uint8_t *p = nullptr;
  if (parse_only_) {
    p = (uint8_t *)aligned_alloc(64, (cap + 63) & ~(size_t)63);
    if (!p) return false;
    if (job.h_in) { memcpy(p, job.h_in, fixed_bytes() < job.h_in_cap ? fixed_bytes() : job.h_in_cap); free(job.h_in); }
    job.h_in = p; job.h_in_cap = cap;
    bind_job(job);
    return true;
  }
  if (hipSetDevice(device_) != hipSuccess) return false;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return false;
  if (job.h_in) {
    memcpy(p, job.h_in, fixed_bytes() < job.h_in_cap ? fixed_bytes() : job.h_in_cap);
    if (job.early_rows.load(std::memory_order_acquire) > 0) hipStreamSynchronize(stream_up_);      // (rows of records on their way up read the old block)
    hipHostFree(job.h_in);
  }
  job.h_in = p; job.h_in_cap = cap;
  bind_job(job);
  return true;
}

bool Decoder::ensure_buffers(int w, int h, int ctb_log2)
{
  if (w == w_ && h == h_ && ctb_log2 == ctbl_) return true;
  // Resolution change (a new SPS took effect at this IRAP picture): what the ring still holds is completed now and queued -- the
  // following calls hand it out one picture at a time, as a software decoder's bumping process would -- before the buffers go
  while (!parse_only_ && w_ && (!gpu_q_.empty() || job_tail_ != job_head_)) {      // (a band decoder's picture between its reconstruction and band_finish -- gpu_job_ -- is not finish_oldest's to complete: drop_pending below lets it go)
    const int rc = finish_oldest();
    if (rc < 0 || (rc > 0 && pic_ready_ && !stash_current_output())) { drop_pending(); break; }
  }
  while (!parse_only_ && w_ && pop_reordered(true)) {}                     // a new sequence follows: what waited for later pictures of the old one leaves in POC order, one picture per call
  if (parse_only_) {
    for (auto &j : jobs_) { free(j.h_in); j.h_in = nullptr; j.h_in_cap = 0; }
    if (jobs_.empty()) jobs_ = std::vector<PicJob>(3);
    gpu_depth_ = 1;
    w_ = w; h_ = h; pw_ = (w + 63) & ~63; ph_ = (h + 63) & ~63; ctbl_ = ctb_log2;
    const size_t nb4 = (size_t)pw_ * ph_ / 16;
    for (auto &j : jobs_) {
      if (!grow_job_input(j, fixed_bytes() + (1 << 16))) return false;
      memset(j.h_in, 0, fixed_bytes());
      j.pred_mode.assign(nb4 / 4, PM_NONE); j.ct_depth.assign(nb4 / 4, 0); j.intra_mode.assign(nb4, 1);
    }
    for (auto &d : dpb_) { d = DpbPic(); d.plane[0] = (uint8_t *)(uintptr_t)64; }      // (never dereferenced: slots are only book-keeping here)
    seen_irap_ = false;
    return true;
  }
  drop_pending();
  sync_main();
  free_buffers();
  if (jobs_.empty()) {
    const char *e = getenv("KVAZZUP_AMD_DEC_GPU_DEPTH");
    gpu_depth_ = e ? atoi(e) : (frame_threads_ >= 4 ? (frame_threads_ >= 24 ? 8 : (frame_threads_ >= 12 ? 4 : 3)) : 1);      // (an intra picture's chain is ~1.5 ms at 1080p: ten picture intervals)
    if (gpu_depth_ > kMaxGpuDepth) gpu_depth_ = kMaxGpuDepth;
    if (gpu_depth_ > frame_threads_ - 1) gpu_depth_ = frame_threads_ - 1;
    if (gpu_depth_ < 1 || band_nrows_ > 0) gpu_depth_ = 1;
    jobs_ = std::vector<PicJob>((size_t)frame_threads_ + 2);   // the pictures being parsed and queued on the GPU (frame_threads_ of them), the one being handed out, the one being filled
  }
  w_ = w; h_ = h; pw_ = (w + 63) & ~63; ph_ = (h + 63) & ~63; ctbl_ = ctb_log2;
  const size_t npx = (size_t)pw_ * ph_, nb4 = npx / 16;
  for (auto &j : jobs_) {
    if (!grow_job_input(j, fixed_bytes() + (1 << 16))) return false;
    memset(j.h_in, 0, fixed_bytes());
    j.pred_mode.assign(nb4 / 4, PM_NONE); j.ct_depth.assign(nb4 / 4, 0); j.intra_mode.assign(nb4, 1);
  }
  h_out_cap_ = npx * 3 / 2;                              // (allocated by the first picture that is downloaded)
  for (int i = 0; i <= gpu_depth_; i++) { d_in_cap_[i] = fixed_bytes() + (1 << 20); HIP_TRY(hipMalloc(&d_in_[i], d_in_cap_[i])); }
  HIP_TRY(hipMalloc(&progress_, sizeof(uint32_t) * (3 * nctb() + 1)));
  { const size_t ec = nctb() * (size_t)(2 << ctbl_), er = nctb() * (size_t)(2 << ctbl_) / 4;          // per CTB of S luma samples a side: S + 2 x S / 2 column words, a quarter as many row words; tagged words: generation 0 = never written
    HIP_TRY(hipMalloc(&edge_col_, ec * sizeof(uint32_t))); HIP_TRY(hipMemset(edge_col_, 0, ec * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&edge_row_, er * 8)); HIP_TRY(hipMemset(edge_row_, 0, er * 8)); chain_gen_ = 0; }      // the CTUs' right columns (k_dec_intra)      // (+ k_dec_intra's ticket counter)
  {
    // dispatch order of k_dec_intra's workgroups: CTUs by anti-diagonal cx + 2 cy (every CTU a block depends on comes earlier)
    const int wc = (w_ + (1 << ctbl_) - 1) >> ctbl_, hc = (h_ + (1 << ctbl_) - 1) >> ctbl_;      // (the picture's coding tree blocks -- with CTBs smaller than 64 fewer than the padded size holds)
    std::vector<uint32_t> order;
    const int r0 = band_nrows_ > 0 ? band_row0_ : 0, nr = band_nrows_ > 0 ? band_nrows_ : hc;      // (band mode: this decoder's CTU rows)
    if (r0 < 0 || r0 + nr > hc) return false;
    for (int d = 0; d < wc + 2 * nr; d++) for (int cy = 0; cy < nr; cy++) { const int cx = d - 2 * cy; if (cx >= 0 && cx < wc) order.push_back((uint32_t)((r0 + cy) * wc + cx)); }
    HIP_TRY(hipMalloc(&intra_order_, sizeof(uint32_t) * order.size()));
    HIP_TRY(hipMemcpy(intra_order_, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice));
  }
  for (int c = 0; c < 3; c++) HIP_TRY(hipMalloc(&work_[c], c ? npx / 4 : npx));
  for (int c = 0; c < 3; c++) HIP_TRY(hipMalloc(&resid_[c], sizeof(int16_t) * (c ? npx / 4 : npx)));
  // a picture buffer: Y | Cb | Cr back to back in one allocation (the whole picture goes to the host in ONE copy)
  for (int s = 0; s < 6; s++) { HIP_TRY(hipMalloc(&dpb_[s].plane[0], npx * 3 / 2)); HIP_TRY(hipMemset(dpb_[s].plane[0], 128, npx * 3 / 2)); dpb_[s].plane[1] = dpb_[s].plane[0] + npx; dpb_[s].plane[2] = dpb_[s].plane[1] + npx / 4; }
  seen_irap_ = false;
  HIP_TRY(hipDeviceSynchronize());                       // (the clears above ran on the null stream: done before anything is queued on the decoder's non-blocking streams)
  return true;
}

// a picture buffer for the picture about to be decoded: not a reference any more and, if it was output, output long enough ago
// Concealment v2 (decoder.h): the buffer that stands in for a reference picture that never arrived.  Marked like the real one would be; its samples are set when
// the picture that asked for it is launched (the GPU may still be reading the buffer's former picture for older pictures that are in flight).
int Decoder::conceal_ref(int poc, bool is_lt)
{
  const int s = alloc_slot();
  if (s < 0) return -1;
  DpbPic &d = dpb_[s];
  d.poc = poc; d.is_ref = true; d.used = true; d.is_lt = is_lt; d.motion.reset();      // (decode_idx stays: the buffer was old enough to be taken and is no picture anybody waits for -- free again the moment no set names it)
  pending_conceal_.emplace_back(s, -2);                           // (-2: the source is chosen once the reference picture set has been applied, decode_slice)
  if (concealed_++ < 3) fprintf(stderr, "kvazzup_amd: decoder: reference picture with POC %d never arrived -- the nearest reference picture (or a grey one) stands in\n", poc);
  return s;
}

int Decoder::alloc_slot()
{
  for (int s = 0; s < KVZ_DEC_MAX_REFS; s++) {
    DpbPic &p = dpb_[s];
    if (p.is_ref || job_head_ - p.decode_idx <= output_hold_ + (gpu_depth_ - 1)) continue;      // (pictures queued on the GPU behind the one handed out may already write their buffers)
    if (!p.plane[0] && parse_only_) p.plane[0] = (uint8_t *)(uintptr_t)64;
    if (!p.plane[0]) {
      const size_t npx = (size_t)pw_ * ph_;
      if (hipSetDevice(device_) != hipSuccess) return -1;
      if (hipMalloc(&p.plane[0], npx * 3 / 2) != hipSuccess) { p.plane[0] = nullptr; return -1; }
      hipMemset(p.plane[0], 128, npx * 3 / 2); p.plane[1] = p.plane[0] + npx; p.plane[2] = p.plane[1] + npx / 4;
      hipDeviceSynchronize();                              // (a clear on the null stream is ordered with nothing on the decoder's non-blocking streams; a buffer is added at most ten times)
    }
    return s;
  }
  return -1;
}

template <class F> void Decoder::timed(int id, F &&launch, hipStream_t st)
{
  if (!prof_now_) { launch(); return; }
  if (!st) st = stream_;
  PicJob &j = *timed_job_;
  if (j.ev_used == j.ev.size()) { PicJob::EvPair p; hipEventCreate(&p.a); hipEventCreate(&p.b); p.id = id; j.ev.push_back(p); }
  PicJob::EvPair &p = j.ev[j.ev_used++]; p.id = id;
  hipEventRecord(p.a, st); launch(); hipEventRecord(p.b, st);
}
// the second chain's stream and arrays (decoder.h stream_alt_), on first use
bool Decoder::ensure_alt()
{
  if (stream_alt_) return true;
  if (alt_failed_) return false;                             // (it did not fit once: the intra-only pictures keep to the one chain instead of trying again for every picture)
  if (hipSetDevice(device_) != hipSuccess) return false;
  const size_t nctu = nctb(), ecw = nctu * (size_t)(2 << ctbl_), npx = (size_t)pw_ * ph_;
  // the second chain's priority level = its pool of hardware queues: the LOWEST level, where nothing else of this library lives (measured, all-intra 1080p with
  // the encoder's second chain at the main stream's level: second decoder chain at the default level 1 563 frames/s -- the level's four queues are taken by
  // tokenizer, input, decoder and transfers --, at the high level 1 681, at the low one 1 724; one chain each side: 1 279)
  { const char *e = getenv("KVAZZUP_AMD_DEC_ALT_PRIO"); alt_prio_ = e ? e[0] : 'l'; }
  // Everything is built in locals and handed to the members in one piece: stream_alt_ != NULL is what launch_gpu takes for "the second chain exists", so a
  // half-built state (a failed allocation at 4K) must never be visible -- on any failure what was allocated is freed and the stream released.
  hipStream_t st = nullptr;
  uint32_t *prog = nullptr, *ecol = nullptr; unsigned long long *erow = nullptr; int16_t *res[3] = {nullptr, nullptr, nullptr}; uint8_t *wrk[3] = {nullptr, nullptr, nullptr};
  auto build = [&]() -> bool {
    HIP_TRY(stream_acquire(&st, device_, 'E', alt_prio_));
    // (cleared ON the stream that is about to use them: the decoder's streams are non-blocking, a clear on the null stream is ordered with nothing -- the first
    // picture of the second chain had its hand-off words zeroed under its hands, every wait in it gave up: error flags 3, found by the serial suite)
    HIP_TRY(hipMalloc(&prog, sizeof(uint32_t) * (3 * nctu + 1))); HIP_TRY(hipMemsetAsync(prog, 0, sizeof(uint32_t) * (3 * nctu + 1), st));
    HIP_TRY(hipMalloc(&ecol, ecw * sizeof(uint32_t))); HIP_TRY(hipMemsetAsync(ecol, 0, ecw * sizeof(uint32_t), st));
    HIP_TRY(hipMalloc(&erow, ecw / 4 * 8)); HIP_TRY(hipMemsetAsync(erow, 0, ecw / 4 * 8, st));
    for (int c = 0; c < 3; c++) { HIP_TRY(hipMalloc(&res[c], sizeof(int16_t) * (c ? npx / 4 : npx))); HIP_TRY(hipMalloc(&wrk[c], c ? npx / 4 : npx)); }
    return true;
  };
  if (!build()) {
    if (st) hipStreamSynchronize(st);                        // (a clear may be queued on it)
    hipFree(prog); hipFree(ecol); hipFree(erow);
    for (int c = 0; c < 3; c++) { hipFree(res[c]); hipFree(wrk[c]); }
    if (st) stream_release(st, device_, 'E', alt_prio_);
    (void)hipGetLastError();
    alt_failed_ = true;
    return false;
  }
  progress_alt_ = prog; edge_col_alt_ = ecol; edge_row_alt_ = erow;
  for (int c = 0; c < 3; c++) { resid_alt_[c] = res[c]; work_alt_[c] = wrk[c]; }
  stream_alt_ = st;
  return true;
}
void Decoder::get_kernel_times(double *ms, uint64_t *launches, bool reset)
{
  for (int i = 0; i < DK_COUNT; i++) { if (ms) ms[i] = k_ms_[i]; if (launches) launches[i] = k_n_[i]; }
  if (reset) for (int i = 0; i < DK_COUNT; i++) { k_ms_[i] = 0; k_n_[i] = 0; }
}

// ------------------------------------------------------------------------------------------ NAL units
// One picture out per call at most (the libOpenHevcDecode contract).  Pictures that were completed ahead of their turn -- the ring's
// contents at a resolution change -- are handed out first, one per call, in order; a picture this call itself completes meanwhile joins
// the end of that queue.
void Decoder::free_retired(bool all)
{
  while (!retired_out_.empty() && (all || nal_calls_ - retired_out_.front().first > kOutHold)) { hipHostFree(retired_out_.front().second); retired_out_.pop_front(); }
  while (!retired_owned_.empty() && (all || nal_calls_ - retired_owned_.front().first > kOutHold)) { if (all) owned_free(retired_owned_.front().second.dev); else owned_release(retired_owned_.front().second.dev); retired_owned_.pop_front(); }
}

int Decoder::decode_nal(const uint8_t *data, size_t len, int64_t pts)
{
  nal_calls_++;
  if (!retired_out_.empty() || !retired_owned_.empty()) free_retired(false);
  const int rc = decode_nal_inner(data, len, pts);
  // ---- output order (C.5.2).  A picture of a stream that may reorder (its SPS says how many pictures may overtake one) is copied out of the ring as
  // it completes and waits in reorder_q_; the smallest POC of the oldest coded video sequence goes out once more pictures wait than may overtake it,
  // or a later sequence has begun, or the stream has ended (EOS / EOB with nothing left in the pipeline).  One picture per call, as ever.
  if (rc >= 0 && ((pic_ready_ && out_.num_reorder > 0) || !reorder_q_.empty())) {
    if (pic_ready_ && out_.num_reorder == 0) {
      // a sequence that does not reorder behind one that did: everything still waiting precedes this picture -- all of it moves to ready_q_ at once
      // (in order), the picture behind it; every later call, parameter sets included, hands one out, so the lag the old sequence needed drains
      // instead of staying until the end of the stream
      while (pop_reordered(true)) {}
      if (!queue_current_output()) return last_error_ = DEC_ERR_GPU;
    } else if (pic_ready_) {
      const int cvs = out_.cvs; reorder_ = out_.num_reorder;
      if (!queue_current_output()) return last_error_ = DEC_ERR_GPU;
      reorder_q_.push_back(Waiting{std::move(ready_q_.back()), cvs}); ready_q_.pop_back();
    }
    const bool eos = len >= 2 && [&] { size_t i = 0; while (i + 2 < len && data[i] == 0) i++; const uint8_t *q = (i >= 2 && i < len && data[i] == 1) ? data + i + 1 : data; const int t = (q[0] >> 1) & 0x3f; return t == 36 || t == 37; }();
    if (ready_q_.empty() && !pop_reordered(eos && pending() == 0)) return 0;
  } else {
  if (ready_q_.empty()) return rc;
  if (rc < 0) return rc;
  if (pic_ready_ && !queue_current_output()) return last_error_ = DEC_ERR_GPU;
  }
  if (cur_owned_.dev || !cur_owned_.host.empty()) retired_owned_.emplace_back(nal_calls_, std::move(cur_owned_));      // (the caller may still be copying it out)
  cur_owned_ = OwnedPic();
  cur_owned_ = std::move(ready_q_.front());
  ready_q_.pop_front();
  out_ = cur_owned_.pic;
  size_t off = 0;
  for (int c = 0; c < 3; c++) {
    const int h = c ? out_.height / 2 : out_.height;
    if (!cur_owned_.host.empty()) { out_.host[c] = cur_owned_.host.data() + off; off += (size_t)out_.host_pitch[c] * h; }
  }
  pic_ready_ = true;
  return 1;
}

// C.5.2.4: of the waiting pictures of the oldest coded video sequence the one with the smallest POC becomes the next of ready_q_ -- when more of
// them wait than may overtake a picture, when a later sequence has begun, or when the stream is over
bool Decoder::pop_reordered(bool flush)
{
  if (reorder_q_.empty()) return false;
  int first_cvs = reorder_q_.front().cvs, n = 0; size_t best = 0;
  for (const Waiting &w : reorder_q_) if (w.cvs < first_cvs) first_cvs = w.cvs;
  bool later = false;
  for (size_t i = 0; i < reorder_q_.size(); i++) {
    const Waiting &w = reorder_q_[i];
    if (w.cvs != first_cvs) { later = true; continue; }
    if (n++ == 0 || w.pic.pic.poc < reorder_q_[best].pic.pic.poc) best = i;
  }
  if (!(flush || later || n > reorder_)) return false;
  ready_q_.push_back(std::move(reorder_q_[best].pic));
  reorder_q_.erase(reorder_q_.begin() + (long)best);
  return true;
}

// the picture complete_gpu() just made the output, copied into storage of its own at the end of the queue.  The device copy comes from a small pool
// of buffers (a hipMalloc / hipFree per picture of a reordering stream synchronised the whole device -- every other encoder and decoder of the
// process with it) and is made by the download stream, which this thread waits for by event.
uint8_t *Decoder::owned_alloc(size_t bytes)
{
  for (size_t i = 0; i < owned_pool_.size(); i++) if (owned_pool_[i].first == bytes) { uint8_t *p = owned_pool_[i].second; owned_pool_.erase(owned_pool_.begin() + (long)i); return p; }
  for (auto &b : owned_pool_) owned_free(b.second);      // (another size: the stream changed its resolution -- every pooled buffer of the old size goes at once: at 4K two dozen of them are 288 MB)
  owned_pool_.clear();
  uint8_t *p = nullptr;
  if (hipSetDevice(device_) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  owned_bytes_[p] = bytes;
  return p;
}
void Decoder::owned_release(uint8_t *p)
{
  if (!p) return;
  auto it = owned_bytes_.find(p);
  if (it != owned_bytes_.end() && owned_pool_.size() < kOwnedPoolMax && (owned_pool_.empty() || owned_pool_.front().first == it->second)) { owned_pool_.emplace_back(it->second, p); return; }      // (the pool holds ONE size: owned_alloc empties it when another one is asked for)
  owned_free(p);
}
// every hipFree of an owned buffer comes through here: the address leaves the size map with it (a stale entry would pool a later allocation at the same
// address under the wrong size)
void Decoder::owned_free(uint8_t *p)
{
  if (!p) return;
  owned_bytes_.erase(p);
  hipFree(p);
}
bool Decoder::queue_current_output()
{
  OwnedPic o;
  o.pic = out_;
  if (download_) {
    for (int c = 0; c < 3; c++) {                                 // plane after plane, pitch x rows each (decode_nal rebuilds the pointers the same way)
      const uint8_t *p = out_.host[c];
      o.host.insert(o.host.end(), p, p + (size_t)out_.host_pitch[c] * (c ? out_.height / 2 : out_.height));
    }
  }
  {
    // the device view: a dense copy (pitch = width), so that device-resident consumers keep working across the re-allocation
    const size_t ny = (size_t)out_.width * out_.height;
    o.dev = owned_alloc(ny * 3 / 2);
    if (!o.dev) return false;
    size_t off = 0;
    hipStream_t st = stream_dl_ ? stream_dl_ : stream_;
    for (int c = 0; c < 3; c++) {
      const int w = c ? out_.width / 2 : out_.width, h = c ? out_.height / 2 : out_.height;
      if (hipMemcpy2DAsync(o.dev + off, (size_t)w, out_.dev[c], (size_t)out_.dev_pitch[c], (size_t)w, (size_t)h, hipMemcpyDeviceToDevice, st) != hipSuccess) { owned_release(o.dev); return false; }
      o.pic.dev[c] = o.dev + off; o.pic.dev_pitch[c] = w;
      off += (size_t)w * h;
    }
    // (the picture's kernels are complete -- complete_gpu saw its event --, so the copies depend on nothing in another stream)
    if (!owned_ev_ && hipEventCreateWithFlags(&owned_ev_, hipEventDisableTiming) != hipSuccess) { owned_release(o.dev); return false; }
    if (hipEventRecord(owned_ev_, st) != hipSuccess || hipEventSynchronize(owned_ev_) != hipSuccess) { owned_release(o.dev); return false; }
  }
  ready_q_.push_back(std::move(o));
  pic_ready_ = false;
  return true;
}

// queue_current_output() for a picture completed AHEAD of its turn (the ring emptied at a resolution change, a picture closed by the next one's first
// NAL unit): a stream that reorders keeps its output order -- the picture joins reorder_q_ with its sequence and POC like any other.
bool Decoder::stash_current_output()
{
  const int cvs = out_.cvs, nr = out_.num_reorder;
  if (!queue_current_output()) return false;
  if (nr > 0 || !reorder_q_.empty()) { if (nr > 0) reorder_ = nr; reorder_q_.push_back(Waiting{std::move(ready_q_.back()), cvs}); ready_q_.pop_back(); }
  return true;
}

// ------------------------------------------------------------------------------------------ scaling lists (7.3.4, 7.4.5)
namespace {
// scaling_list_data(): every list either the default one, a copy of an earlier list of its size, or 16 / 64 entries in diagonal scan order
bool parse_scaling_list_data(BitReader &r, ScalingLists &sl)
{
  const ScanTabs &st = scan_tabs();
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < (s == 3 ? 2 : 6); m++) {
      if (!r.get(1)) {
        const uint32_t delta = r.ue();
        if (delta > (uint32_t)m) return false;
        if (delta == 0) scaling_default_one(sl, s, m);
        else { memcpy(sl.m[s][m], sl.m[s][m - (int)delta], 64); if (s >= 2) sl.dc[s - 2][m] = sl.dc[s - 2][m - (int)delta]; }
      } else {
        int next = 8;
        if (s >= 2) { const int dc = r.se(); if (dc < -7 || dc > 247) return false; next = dc + 8; sl.dc[s - 2][m] = (uint8_t)next; }
        const int l2 = s == 0 ? 2 : 3, n = 1 << l2;
        for (int i = 0; i < n * n; i++) {
          const int d = r.se();
          if (d < -128 || d > 127) return false;
          next = (next + d + 256) & 255;
          if (!next) return false;
          sl.m[s][m][st.y[0][l2][i] * n + st.x[0][l2][i]] = (uint8_t)next;      // diagonal scan position i -> (x, y)
        }
      }
      if (r.err) return false;
    }
  return true;
}
std::shared_ptr<const std::vector<uint8_t>> build_scaling(const ScalingLists &sl)
{
  auto out = std::make_shared<std::vector<uint8_t>>((size_t)KVZ_SCALING_BYTES);
  scaling_factors(sl, out->data());
  return out;
}
}  // namespace

int Decoder::decode_nal_inner(const uint8_t *data, size_t len, int64_t pts)
{
  Tick tk_nal;
  struct Acc { double &d; Tick &t; double &w, &s, &a, &y; double w0, s0, a0, y0; ~Acc() { d += t.ms() - ((w - w0) + (s - s0) + (a - a0) + (y - y0)); } } acc_{t_nal_, tk_nal, t_wait_, t_stage_, t_api_, t_sync_, t_wait_, t_stage_, t_api_, t_sync_};
  pic_ready_ = false;
  if (!started_) return last_error_ = DEC_ERR_GPU;
  size_t i = 0;
  while (i + 2 < len && data[i] == 0) i++;
  if (i >= 2 && i < len && data[i] == 1) { data += i + 1; len -= i + 1; }
  if (len < 2 || (data[0] & 0x80)) return last_error_ = DEC_ERR_INVALID;      // EOS / EOB are header-only
  const int nal_type = (data[0] >> 1) & 0x3f, layer = ((data[0] & 1) << 5) | (data[1] >> 3);
  if (layer != 0) return 0;
  cur_tid_ = (data[1] & 7) - 1;
  if (nal_type < 32 && cur_tid_ > max_tid_) return 0;            // (set_max_temporal_id: a sub-layer the caller does not want)
  if (rbsp_.size() < len + 32) rbsp_.resize(len + 32);
  epb_.clear();
  // emulation prevention bytes out (7.4.2; hevc_headers.h append_nal is the inverse): the bytes between zero bytes in one piece
  size_t n = 0; int zeros = 0;
  for (size_t k = 2; k < len;) {
    if (zeros == 0) {
      const uint8_t *z = (const uint8_t *)memchr(data + k, 0, len - k);
      const size_t m = z ? (size_t)(z - (data + k)) : len - k;
      memcpy(rbsp_.data() + n, data + k, m); n += m; k += m;
      if (k >= len) break;
    }
    if (zeros >= 2 && data[k] == 3) { zeros = 0; epb_.push_back(n); k++; continue; }
    rbsp_[n++] = data[k]; zeros = data[k] == 0 ? zeros + 1 : 0; k++;
  }
  memset(rbsp_.data() + n, 0, 32);                          // (readers may look a few bytes past the end)
  BitReader r(rbsp_.data(), n);
  if (asm_active_ && (asm_guessed_one_row_ || asm_free_) && nal_type >= 32 && nal_type <= 40 && nal_type != 38) { const int rc = close_open_picture(); if (rc < 0) return rc; }   // (what can only open the next access unit, or end the sequence)
  if (nal_type == 32) {                                          // VPS: only the timing information is used
    r.get(4); r.get(2); r.get(6); int msl = r.get(3); r.get(1); r.get(16);
    if (!skip_ptl(r, msl)) return last_error_ = DEC_ERR_INVALID;
    int oi = r.get(1);
    for (int k = oi ? 0 : msl; k <= msl; k++) { r.ue(); r.ue(); r.ue(); }
    int max_layer_id = r.get(6); int nls = r.ue() + 1;
    if (nls > 1024) return last_error_ = DEC_ERR_INVALID;
    for (int a = 1; a < nls; a++) for (int b = 0; b <= max_layer_id; b++) r.get(1);
    if (r.get(1)) { vps_fps_den_ = r.get(32); vps_fps_num_ = r.get(32); }
    return r.err ? (last_error_ = DEC_ERR_INVALID) : 0;
  }
  if (nal_type == 33) {                                          // SPS (7.3.2.2)
    DecSps s;
    r.get(4); int msl = r.get(3); r.get(1);
    if (!skip_ptl(r, msl)) return last_error_ = DEC_ERR_INVALID;
    int id = r.ue(); if (id > 15) return last_error_ = DEC_ERR_INVALID;
    if (r.ue() != 1) return last_error_ = DEC_ERR_UNSUPPORTED;   // 4:2:0 only
    s.width = r.ue(); s.height = r.ue();
    if (r.get(1)) {
      const uint32_t cl = r.ue(), cr = r.ue(), ct = r.ue(), cb = r.ue();
      if (cl > 8192 || cr > 8192 || ct > 8192 || cb > 8192) return last_error_ = DEC_ERR_INVALID;
      s.crop_l = 2 * (int)cl; s.crop_r = 2 * (int)cr; s.crop_t = 2 * (int)ct; s.crop_b = 2 * (int)cb;
    }
    if (r.ue() != 0 || r.ue() != 0) return last_error_ = DEC_ERR_UNSUPPORTED;   // 8 bit only
    s.log2_max_poc_lsb = r.ue() + 4;
    if (s.log2_max_poc_lsb > 16) return last_error_ = DEC_ERR_INVALID;
    int oi = r.get(1);
    for (int k = oi ? 0 : msl; k <= msl; k++) { r.ue(); s.num_reorder = r.ue(); r.ue(); }      // (max_dec_pic_buffering, max_num_reorder_pics, max_latency_increase: the highest sub-layer's stay)
    if (s.num_reorder < 0 || s.num_reorder > 15) return last_error_ = DEC_ERR_INVALID;
    int log2_min_cb = r.ue() + 3, diff_cb = r.ue(), log2_min_tb = r.ue() + 2, diff_tb = r.ue();
    s.th_depth_inter = r.ue(); s.th_depth_intra = r.ue();
    if (r.get(1)) {                                             // scaling_list_enabled_flag: the default lists, or sps_scaling_list_data
      ScalingLists sl = scaling_defaults();
      if (r.get(1) && !parse_scaling_list_data(r, sl)) return last_error_ = DEC_ERR_INVALID;
      s.scaling = build_scaling(sl);
    }
    s.amp = r.get(1); s.sao = r.get(1);
    if (r.get(1)) {                                               // pcm_enabled_flag
      s.pcm_depth[0] = (int)r.get(4) + 1; s.pcm_depth[1] = (int)r.get(4) + 1;
      s.pcm_min_log2 = (int)r.ue() + 3; s.pcm_max_log2 = s.pcm_min_log2 + (int)r.ue(); s.pcm_no_filter = r.get(1);
      if (r.err || s.pcm_depth[0] > 8 || s.pcm_depth[1] > 8 || s.pcm_min_log2 < log2_min_cb || s.pcm_max_log2 > imin(5, log2_min_cb + diff_cb)) return last_error_ = DEC_ERR_INVALID;
    }
    if (r.err) return last_error_ = DEC_ERR_INVALID;
    // coding geometry: CTB 64 (what Kvazaar always writes), 32 or 16 (round 6: other encoders); coding blocks from 8 (Kvazaar), 16 or 32 up; transform blocks 4 .. min(32, CTB)
    if (log2_min_cb < 3 || log2_min_cb > 5 || diff_cb < 0 || diff_cb > 3) return last_error_ = DEC_ERR_INVALID;
    s.ctb_log2 = log2_min_cb + diff_cb; s.min_cb_log2 = log2_min_cb;
    if (s.ctb_log2 < 4 || s.ctb_log2 > 6 || log2_min_tb != 2 || diff_tb != imin(3, s.ctb_log2 - 2) || s.th_depth_inter > 4 || s.th_depth_intra > 4)
      return last_error_ = DEC_ERR_UNSUPPORTED;
    s.num_st_rps = r.ue();
    if (s.num_st_rps > 64) return last_error_ = DEC_ERR_INVALID;
    for (int k = 0; k < s.num_st_rps; k++) if (!parse_st_rps(r, k, s.num_st_rps, s.st_rps, s.st_rps[k])) return last_error_ = DEC_ERR_INVALID;
    if (r.get(1)) {                                               // long_term_ref_pics_present_flag: candidates by POC LSBs
      s.num_lt_sps = (int)r.ue();
      if (s.num_lt_sps > 32) return last_error_ = DEC_ERR_INVALID;
      for (int k = 0; k < s.num_lt_sps; k++) { s.lt_lsb_sps[k] = (uint16_t)r.get(s.log2_max_poc_lsb); s.lt_used_sps[k] = (uint8_t)r.get(1); }
    }
    s.tmvp = r.get(1);
    s.strong_intra = r.get(1);
    if (r.get(1)) {                                               // VUI: timing only
      if (r.get(1)) { if (r.get(8) == 255) { r.get(16); r.get(16); } }
      if (r.get(1)) r.get(1);
      if (r.get(1)) { r.get(4); if (r.get(1)) r.get(24); }
      if (r.get(1)) { r.ue(); r.ue(); }
      r.get(3);
      if (r.get(1)) { r.ue(); r.ue(); r.ue(); r.ue(); }
      if (r.get(1)) { s.fps_den = r.get(32); s.fps_num = r.get(32); }
    }
    if (r.err) return last_error_ = DEC_ERR_INVALID;
    // sizes: multiples of the minimum coding block; the upper bound is the encoder's (and keeps every index inside 32 bits)
    if ((s.width & ((1 << s.min_cb_log2) - 1)) || (s.height & ((1 << s.min_cb_log2) - 1))) return last_error_ = DEC_ERR_INVALID;      // (7.4.3.2.1: multiples of MinCbSizeY)
    if ((s.width & 7) || (s.height & 7) || s.width < 16 || s.height < 16 || s.width > 16384 || s.height > 16384) return last_error_ = DEC_ERR_UNSUPPORTED;
    if (s.crop_l + s.crop_r >= s.width || s.crop_t + s.crop_b >= s.height) return last_error_ = DEC_ERR_INVALID;
    s.valid = true; sps_[id] = std::make_shared<const DecSps>(s);      // (a new object: pictures still being parsed keep the one they were coded with)
    return 0;
  }
  if (nal_type == 34) {                                          // PPS (7.3.2.3)
    DecPps p;
    int id = r.ue(); p.sps_id = r.ue();
    if (id > 63 || p.sps_id > 15) return last_error_ = DEC_ERR_INVALID;
    int dep = r.get(1); p.output_flag_present = r.get(1); p.extra_header_bits = r.get(3); p.sign_hiding = r.get(1);
    p.cabac_init_present = r.get(1);
    p.num_ref_idx_default = (int)r.ue() + 1; p.num_ref_idx1_default = (int)r.ue() + 1;
    p.init_qp = 26 + r.se();
    if (p.init_qp < 0 || p.init_qp > 51) return last_error_ = DEC_ERR_INVALID;
    int cip = r.get(1); p.tskip = r.get(1); p.cu_qp_delta = r.get(1);
    if (p.cu_qp_delta) { p.qp_delta_depth = r.ue(); if (p.qp_delta_depth > 3) return last_error_ = DEC_ERR_INVALID; }
    p.cb_qp_offset = r.se(); p.cr_qp_offset = r.se(); p.slice_chroma_offsets = r.get(1);
    int wp = r.get(1), wbp = r.get(1), tqb = r.get(1), tiles = r.get(1);
    p.wpp = r.get(1);
    if (r.err || p.num_ref_idx_default > 15 || p.num_ref_idx1_default > 15 || p.cb_qp_offset < -12 || p.cb_qp_offset > 12 || p.cr_qp_offset < -12 || p.cr_qp_offset > 12) return last_error_ = DEC_ERR_INVALID;
    p.dependent_slices = dep;
    p.cip = cip;                                                 // constrained_intra_pred_flag: the kernels' business (reference samples of blocks that are not intra-coded do not count)
    p.weighted_pred = wp; p.weighted_bipred = wbp;
    p.tq_bypass = tqb;
    if (tiles) {                                                 // supported: the level limits of 20 columns x 22 rows (A.4.2); loop filter across tiles on
      const int cols = r.ue() + 1, rows = r.ue() + 1; p.uniform_tiles = r.get(1);
      if (cols > 20 || rows > 22) return last_error_ = DEC_ERR_UNSUPPORTED;
      if (!p.uniform_tiles) {
        for (int k = 0; k < cols - 1; k++) { p.col_width[k] = (int)r.ue() + 1; if (p.col_width[k] > 1024) return last_error_ = DEC_ERR_INVALID; }
        for (int k = 0; k < rows - 1; k++) { p.row_height[k] = (int)r.ue() + 1; if (p.row_height[k] > 1024) return last_error_ = DEC_ERR_INVALID; }
      }
      p.across_tiles = r.get(1);                                 // loop_filter_across_tiles_enabled_flag (Kvazaar writes 0: its tiles are filtered one by one)
      p.tile_rows = rows; p.tile_cols = cols;
    }
    p.loop_filter_across_slices = r.get(1);
    p.deblock_control = r.get(1);
    if (p.deblock_control) {
      p.deblock_override = r.get(1);
      p.deblock_disabled = r.get(1);
      if (!p.deblock_disabled) { p.beta_offset_div2 = r.se(); p.tc_offset_div2 = r.se(); }
    }
    if (r.get(1)) {                                              // pps_scaling_list_data: instead of the SPS's lists
      ScalingLists sl = scaling_defaults();
      if (!parse_scaling_list_data(r, sl)) return last_error_ = DEC_ERR_INVALID;
      p.scaling = build_scaling(sl);
    }
    p.lists_mod = r.get(1);                                       // lists_modification_present_flag
    p.par_mrg_level = (int)r.ue() + 2;
    p.header_extension = r.get(1);
    if (r.err || p.par_mrg_level > 6 || p.beta_offset_div2 < -6 || p.beta_offset_div2 > 6 || p.tc_offset_div2 < -6 || p.tc_offset_div2 > 6) return last_error_ = DEC_ERR_INVALID;
    p.valid = true; pps_[id] = p;
    return 0;
  }
  if (nal_type == 36 || nal_type == 37) { after_eos_ = true; vwait_.clear(); int rc = finish_oldest(); if (rc < 0) last_error_ = rc; return rc; }   // EOS / EOB: drain one delayed picture; whatever picture follows starts a sequence
  if (nal_type == 40 && check_hash_) return hash_sei(rbsp_.data(), n);       // suffix SEI: decoded picture hash (libOpenHevcSetCheckMD5)
  if (nal_type > 31) return 0;                                    // AUD / other SEI / ...
  if (!(nal_type <= 9 || (nal_type >= 16 && nal_type <= 21))) return last_error_ = DEC_ERR_UNSUPPORTED;      // (reserved types)
  int rc = decode_slice(rbsp_.data(), n, nal_type, pts);
  if (rc < 0) last_error_ = rc;
  return rc;
}

// Decoded picture hash SEI (D.2.19, payload type 132) with libOpenHevcSetCheckMD5(h, 1): the message follows its picture's last slice
// segment.  Frame-threaded decoder: the picture is still in the ring -- the expected hashes travel with its job and are compared when the
// picture completes.  Synchronous decoder: the picture has just been output and still sits untouched in its buffer -- compared at once.
int Decoder::hash_sei(const uint8_t *rbsp, size_t len)
{
  BitReader r(rbsp, len);
  while (r.pos + 16 <= len * 8 && !r.err) {
    int type = 0, size = 0, b;
    do { b = (int)r.get(8); type += b; } while (b == 255 && !r.err);
    do { b = (int)r.get(8); size += b; } while (b == 255 && !r.err);
    if (r.err || r.pos + (size_t)size * 8 > len * 8) break;
    if (type != 132 || size < 1) { for (int k = 0; k < size; k++) r.get(8); continue; }
    std::vector<uint8_t> want((size_t)size);
    for (int k = 0; k < size; k++) want[(size_t)k] = (uint8_t)r.get(8);
    const int ht = want[0];
    if ((ht != 0 && ht != 2) || size != 1 + 3 * (ht == 0 ? 16 : 4)) continue;      // (CRC: not checked)
    if (job_head_ == 0) continue;
    PicJob &job = jobs_[(size_t)((job_head_ - 1) % (long)jobs_.size())];
    if (parse_only_) continue;
    if (frame_threads_ > 1 || band_nrows_ > 0) { job.expect_hash = std::move(want); continue; }
    const int rc = verify_hash(job, want);
    if (rc < 0) return last_error_ = rc;
  }
  return 0;
}

// the picture of `job` (complete on the device) against the hashes of its SEI message; the samples come down once more for it
int Decoder::verify_hash(const PicJob &job, const std::vector<uint8_t> &want)
{
  const size_t npx = (size_t)pw_ * ph_;
  std::vector<uint8_t> pic(npx * 3 / 2);
  if (hipSetDevice(device_) != hipSuccess || hipMemcpy(pic.data(), dpb_[job.slot].plane[0], pic.size(), hipMemcpyDeviceToHost) != hipSuccess) return DEC_ERR_GPU;
  const uint8_t *pl[3] = {pic.data(), pic.data() + npx, pic.data() + npx + npx / 4};
  const size_t pitch[3] = {(size_t)pw_, (size_t)pw_ / 2, (size_t)pw_ / 2};
  const std::vector<uint8_t> got = picture_hash_payload(want[0], pl, pitch, w_, h_);
  hash_checked_++;
  if (got != want) {
    hash_mismatch_++;
    fprintf(stderr, "kvazzup_amd: decoded picture hash mismatch (POC %d)\n", job.sh.poc);
    return DEC_ERR_HASH;
  }
  return 0;
}

// A picture whose last slice segment has not arrived when something that can only belong to the NEXT access unit turns up (a first slice
// segment, a parameter set, an access unit delimiter, a prefix SEI, the end of the sequence).  Either its first segment's extent was guessed
// wrong -- a one-tile picture without WPP whose PPS allows dependent slice segments is taken to begin with a one-row segment
// (append_segment), but the flag does not forbid whole pictures in one segment: then the segment IS the picture, it ends where its
// end_of_slice_segment_flag says and is submitted now, ahead of the NAL unit at hand -- or segments were lost and the picture is dropped,
// which must not pass unnoticed: the error code is left for kvzx_decoder_last_error and said once on stderr.
// A picture of free slices (append_segment) when its access unit has ended: every segment ends where the next begins, the last one with the picture.  What
// the parser needs per coding tree block -- which slice, whether a segment begins or ends there, where its bytes are -- is laid out here; the substreams
// stay what they are for any one-tile picture (a CTB row each with WPP, else the picture), a segment that begins inside one restarts the arithmetic decoder there.
int Decoder::close_free_picture(PicJob &job)
{
  const int wc = (w_ + (1 << ctbl_) - 1) >> ctbl_, hc = (h_ + (1 << ctbl_) - 1) >> ctbl_, total = wc * hc, n = (int)asm_segs_.size();
  const bool wpp = job.pps.wpp != 0;
  if (n < 1 || asm_segs_[0].address != 0 || asm_segs_[0].dependent) return DEC_ERR_INVALID;
  if (n == 1 && frame_threads_ == 1) free_stream_ = false;      // (one segment: the stream may be back to whole pictures -- the synchronous decoder can afford to find out, submit_job take_back)
  if (n == 1) {
    // the whole picture in one segment: the layout of Kvazaar's forms -- nothing per coding tree block, the parser's availability tests as they were
    if ((int)asm_segs_[0].subs.size() != (wpp ? hc : 1)) return DEC_ERR_INVALID;
    job.ctb_cut.clear(); job.ctb_slice.clear(); job.ctb_data.clear(); job.slice_qps.clear();
    job.sub_start = asm_segs_[0].subs;
    job.seg_end_row.assign((size_t)hc, 0); job.seg_end_row[(size_t)hc - 1] = 1; job.row_restart.assign((size_t)hc, SIZE_MAX);
    job.seg_end_sub.assign(job.geom.size(), 0);
    return 0;
  }
  job.ctb_cut.assign((size_t)total, 0); job.ctb_slice.assign((size_t)total, 0); job.ctb_data.assign((size_t)total, SIZE_MAX); job.slice_qps.clear();
  job.sub_start.assign(wpp ? (size_t)hc : 1, SIZE_MAX);
  job.seg_end_row.assign((size_t)hc, 0); job.row_restart.assign((size_t)hc, SIZE_MAX);
  int slice = -1;
  for (int k = 0; k < n; k++) {
    const FreeSeg &sg = asm_segs_[(size_t)k];
    const int a = sg.address, e = k + 1 < n ? asm_segs_[(size_t)k + 1].address : total;
    if (a < 0 || e <= a || e > total || sg.subs.empty()) return DEC_ERR_INVALID;
    if (!sg.dependent) { if (++slice > 255) return DEC_ERR_UNSUPPORTED; job.slice_qps.push_back((int8_t)sg.slice_qp); }      // (the kernels tell slices apart by a byte per block)
    job.ctb_cut[(size_t)a] |= sg.dependent ? 2 : 1; job.ctb_cut[(size_t)e - 1] |= 4; job.ctb_data[(size_t)a] = sg.subs[0];
    memset(job.ctb_slice.data() + a, slice, (size_t)(e - a));
    const int rows = (e - 1) / wc - a / wc + 1;
    if (wpp) {
      if ((int)sg.subs.size() != rows) return DEC_ERR_INVALID;      // (an entry point per CTB row the segment goes on into)
      for (int q = 0; q < rows; q++) if (q > 0 || a % wc == 0) job.sub_start[(size_t)(a / wc + q)] = sg.subs[(size_t)q];
    } else {
      if (sg.subs.size() != 1) return DEC_ERR_INVALID;
      if (k == 0) job.sub_start[0] = sg.subs[0];
    }
  }
  for (size_t q = 0; q < job.sub_start.size(); q++) if (job.sub_start[q] == SIZE_MAX || (q > 0 && job.sub_start[q] <= job.sub_start[q - 1])) return DEC_ERR_INVALID;
  if (wpp) job.ds_saved.assign((size_t)hc * CTX_COUNT, 0);
  for (int cy = 0; cy < hc; cy++) memcpy(job.ctu_tile + (size_t)cy * wc, job.ctb_slice.data() + (size_t)cy * wc, (size_t)wc);      // (one tile: the byte the kernels compare between adjacent blocks is the slice's)
  job.sh.slice_qp = job.slice_qps[0];
  job.seg_end_sub.assign(job.geom.size(), 0);
  return 0;
}

int Decoder::close_open_picture()
{
  if (!asm_active_) return 0;
  asm_active_ = false;
  PicJob &old = jobs_[(size_t)(job_head_ % jobs_.size())];
  const int old_hc = (h_ + (1 << ctbl_) - 1) >> ctbl_;
  if (asm_free_) {
    asm_free_ = false;
    int rc = close_free_picture(old);
    old.ambiguous_end = false;
    if (rc >= 0) rc = submit_job(old, asm_nal_type_, asm_irap_);
    if (rc < 0) { last_error_ = rc; return 0; }
    if (rc > 0 && pic_ready_ && !stash_current_output()) return last_error_ = DEC_ERR_GPU;
    return 0;
  }
  if (asm_guessed_one_row_ && asm_rows_ == 1 && old_hc > 1) {
    asm_guessed_one_row_ = false;
    old.seg_end_row[0] = 0; old.seg_end_row[(size_t)(old_hc - 1)] = 1;
    const int rc = submit_job(old, asm_nal_type_, asm_irap_);
    if (rc < 0) { last_error_ = rc; return 0; }
    if (rc > 0 && pic_ready_ && !stash_current_output()) return last_error_ = DEC_ERR_GPU;     // (handed out first, by the next call: the NAL unit at hand may produce a picture of its own)
    return 0;
  }
  if (old.pps.tile_rows == 1 && old.pps.tile_cols == 1 && band_nrows_ == 0 && asm_segs_.size() > 1) {
    // One tile, and the rows counted so far do not make the picture: they were counted on guesses -- without WPP no header says how far a segment reaches, a
    // dependent segment at a row's start was taken for that one row (the form Kvazaar's slices=wpp has WITH WPP).  The stream cuts its pictures as it likes after
    // all (it had looked like whole pictures again: close_free_picture's single-segment rule): the segments are all here, each ends where the next begins, the last
    // one with the picture.  If one was lost instead, the parser finds a segment ending early and the picture fails there.
    asm_guessed_one_row_ = false;
    int rc = close_free_picture(old);
    old.ambiguous_end = false;
    if (rc >= 0) { free_stream_ = true; rc = submit_job(old, asm_nal_type_, asm_irap_); }
    if (rc < 0) { last_error_ = rc; return 0; }
    if (rc > 0 && pic_ready_ && !stash_current_output()) return last_error_ = DEC_ERR_GPU;
    return 0;
  }
  last_error_ = DEC_ERR_INVALID;
  static bool said = false;
  if (!said) { said = true; fprintf(stderr, "kvazzup_amd: decoder dropped a picture whose slice segments did not complete before the next picture began\n"); }
  return 0;
}

int Decoder::decode_slice(const uint8_t *rbsp, size_t len, int nal_type, int64_t pts)
{
  BitReader r(rbsp, len);
  const bool idr = nal_type == 19 || nal_type == 20, irap = nal_type >= 16 && nal_type <= 23;
  // A picture may come in several slice segments, one NAL unit each -- the two ways a Kvazaar peer cuts them (uvgComm video/Slices,
  // kvazaarfilter.cpp:205-215): a DEPENDENT slice segment per CTU row ("slices=wpp"), an independent slice per tile ("slices=tiles").
  // Supported: segments that arrive in order and consist of whole CTU rows (WPP) or whole tiles; independent slices repeat the first
  // one's header (the picture keeps one set of slice parameters).  The job is filled segment by segment and submitted with the last.
  const bool first_seg = r.get(1) != 0;
  const bool prior_flag = irap && r.get(1) != 0;                 // no_output_of_prior_pics_flag
  if ((nal_type == 8 || nal_type == 9) && skip_rasl_) {
    // a RASL picture of an IRAP picture that starts a coded video sequence (8.1.3: decoding began there, or a splicer called it BLA, or an end of sequence
    // NAL unit precedes it): it predicts from pictures of the sequence before, which are not there -- not decoded, not output
    if (first_seg && asm_active_) return close_open_picture();
    return 0;
  }
  const int pps_id = r.ue();
  if (pps_id < 0 || pps_id > 63 || !pps_[pps_id].valid || !sps_[pps_[pps_id].sps_id] || !sps_[pps_[pps_id].sps_id]->valid) return DEC_ERR_INVALID;
  const DecPps &p = pps_[pps_id]; const std::shared_ptr<const DecSps> sps_ref = sps_[p.sps_id]; const DecSps &s = *sps_ref;
  bool dependent = false; int seg_address = 0;
  if (!first_seg) {
    const int ctbs = 1 << s.ctb_log2, nctb = ((s.width + ctbs - 1) >> s.ctb_log2) * ((s.height + ctbs - 1) >> s.ctb_log2);
    int bits = 0; while ((1 << bits) < nctb) bits++;
    if (p.dependent_slices) dependent = r.get(1) != 0;
    seg_address = r.get(bits);
    if (!asm_active_ && frame_threads_ > 1 && !parse_only_ && job_head_ > job_tail_ && pps_id == asm_pps_id_ && nal_type == asm_nal_type_) {
      // frame threads: the picture this segment may belong to is with a worker -- submitted because its segments covered it row by row (PicJob::ambiguous_end).
      // The worker's verdict is waited for (a parse that ends early is a short one): "its last segment ends before the picture does" takes the picture back,
      // open again, and this segment joins it.
      PicJob &last = jobs_[(size_t)((job_head_ - 1) % jobs_.size())];
      if (last.ambiguous_end) {
        for (int st; (st = last.state.load(std::memory_order_acquire)) == 1;) futex_wait(last.state, st);
        if (last.state.load(std::memory_order_acquire) == 2 && last.rc == DEC_SEG_ENDS_EARLY) take_back_job(last);
      }
    }
    if (!asm_active_ && p.tile_cols == 1 && p.tile_rows == 1 && (!dependent || seg_address % ((s.width + ctbs - 1) >> s.ctb_log2) != 0)) {
      // no picture is open, and this is no segment of Kvazaar's forms (a dependent segment per CTU row): the stream cuts its pictures into slices as it likes -- its
      // first picture went off as one segment (where a picture without WPP ends is not in its first segment's header).  From here on a picture of this stream is put
      // together from its segments when its access unit ends (append_segment, close_free_picture); this one is lost.
      free_stream_ = true;
    }
    if (!asm_active_ || pps_id != asm_pps_id_ || nal_type != asm_nal_type_) return DEC_ERR_INVALID;      // a segment without its picture's first one (lost), or of another picture
  } else if (asm_active_) {
    // the previous picture never got its last segment: close_open_picture() submits or drops it; this NAL unit -- a new picture -- is decoded normally
    const int rc = close_open_picture();
    if (rc < 0) return rc;
  }
  PicJob *const open_job = first_seg ? nullptr : &jobs_[(size_t)(job_head_ % jobs_.size())];
  if (dependent && open_job->pps.tile_cols > 1) { asm_active_ = false; return DEC_ERR_UNSUPPORTED; }     // (with tile columns: whole pictures or slices of whole tiles)
  if (dependent) {
    // 7.3.6.1: everything but the address and the entry points is taken over from the slice's first segment
    asm_cur_dependent_ = true;
    const int wc = (s.width + (1 << s.ctb_log2) - 1) >> s.ctb_log2, hc = (s.height + (1 << s.ctb_log2) - 1) >> s.ctb_log2;
    return append_segment(*open_job, r.pos, rbsp, len, p, open_job->pps, wc, hc, seg_address, pts);
  }
  for (int k = 0; k < p.extra_header_bits; k++) r.get(1);
  const int slice_type = r.ue();
  if (slice_type < 0 || slice_type > 2) return DEC_ERR_INVALID;
  SliceHdr sh;
  sh.is_intra = slice_type == 2; sh.is_b = slice_type == 0;
  if (p.output_flag_present) sh.no_output = r.get(1) == 0;
  // 8.1.3 NoRaslOutputFlag: an IDR or BLA picture, or a CRA picture that is the first one decoded or follows an end of sequence NAL unit, starts a coded video
  // sequence -- POC MSBs from zero, no reference picture survives, its RASL pictures are dropped.  (first_seg: an open picture has been closed above, seen_irap_ is current.)
  if (first_seg) {
    cur_no_rasl_ = irap && (idr || nal_type <= 18 || !seen_irap_ || after_eos_); if (irap) skip_rasl_ = cur_no_rasl_;
    // C.5.2.2: an IDR or BLA picture that is not the first one empties the buffer WITHOUT output when its flag says so (a CRA picture gets here behind an end of
    // sequence NAL unit only, which has put out everything already)
    cur_discard_ = cur_no_rasl_ && prior_flag && nal_type != 21 && seen_irap_ && !after_eos_;
    after_eos_ = false;
  }
  const bool no_rasl_out = cur_no_rasl_;
  StRps rps;
  int nlt = 0, lt_lsb[16] = {}, lt_cycle[16] = {}; bool lt_used[16] = {}, lt_msb[16] = {};
  if (!idr) {
    const int lsb = r.get(s.log2_max_poc_lsb), max_lsb = 1 << s.log2_max_poc_lsb;
    const int prev_lsb = prev_poc_ & (max_lsb - 1), prev_msb = prev_poc_ - prev_lsb;
    int msb = prev_msb;
    if (lsb < prev_lsb && prev_lsb - lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (lsb > prev_lsb && lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    if (no_rasl_out) msb = 0;
    sh.poc = msb + lsb;
    if (r.get(1)) {
      int idx = 0, bits = 0; while ((1 << bits) < s.num_st_rps) bits++;
      if (s.num_st_rps == 0) return DEC_ERR_INVALID;
      if (bits) idx = r.get(bits);
      if (idx >= s.num_st_rps) return DEC_ERR_INVALID;
      rps = s.st_rps[idx];
    } else if (!parse_st_rps(r, s.num_st_rps, s.num_st_rps, s.st_rps, rps)) return DEC_ERR_INVALID;
    if (s.num_lt_sps >= 0) {
      // long-term reference pictures (7.3.6.1): candidates of the SPS by index, then explicit ones; DeltaPocMsbCycleLt accumulates inside each group (7-52)
      const int n_sps = s.num_lt_sps > 0 ? (int)r.ue() : 0, n_pics = (int)r.ue();
      if (r.err || n_sps < 0 || n_sps > s.num_lt_sps || n_pics < 0 || n_sps + n_pics > 16) return DEC_ERR_INVALID;
      nlt = n_sps + n_pics;
      int bits = 0; while ((1 << bits) < s.num_lt_sps) bits++;
      for (int k = 0; k < nlt; k++) {
        if (k < n_sps) { const int idx = bits ? (int)r.get(bits) : 0; if (idx >= s.num_lt_sps) return DEC_ERR_INVALID; lt_lsb[k] = s.lt_lsb_sps[idx]; lt_used[k] = s.lt_used_sps[idx] != 0; }
        else { lt_lsb[k] = (int)r.get(s.log2_max_poc_lsb); lt_used[k] = r.get(1) != 0; }
        lt_msb[k] = r.get(1) != 0;
        const int delta = lt_msb[k] ? (int)r.ue() : 0;
        if (delta < 0 || delta > (1 << 20) || lt_cycle[k ? k - 1 : 0] > (1 << 24)) return DEC_ERR_INVALID;
        lt_cycle[k] = delta + ((k == 0 || k == n_sps) ? 0 : lt_cycle[k - 1]);
      }
      if (r.err) return DEC_ERR_INVALID;
    }
    if (s.tmvp) sh.tmvp = r.get(1);
  }
  if (s.sao) { sh.sao_luma = r.get(1); sh.sao_chroma = r.get(1); }
  sh.num_ref_idx = p.num_ref_idx_default; sh.num_ref_idx1 = sh.is_b ? p.num_ref_idx1_default : 0;
  if (!sh.is_intra) {
    if (r.get(1)) { sh.num_ref_idx = (int)r.ue() + 1; if (sh.is_b) sh.num_ref_idx1 = (int)r.ue() + 1; }
    if (sh.num_ref_idx < 1 || sh.num_ref_idx > 15 || (sh.is_b && (sh.num_ref_idx1 < 1 || sh.num_ref_idx1 > 15))) return DEC_ERR_INVALID;
    if (p.lists_mod) {                                            // ref_pic_lists_modification(): NumPicTotalCurr = the set's used pictures, short-term and long-term
      int total = 0;
      for (int k = 0; k < rps.n_neg + rps.n_pos; k++) total += rps.used[k] ? 1 : 0;
      for (int k = 0; k < nlt; k++) total += lt_used[k] ? 1 : 0;
      if (total > 1) {
        int bits = 0; while ((1 << bits) < total) bits++;
        for (int l = 0; l < (sh.is_b ? 2 : 1); l++) {
          sh.list_mod[l] = (uint8_t)r.get(1);
          if (sh.list_mod[l]) for (int i = 0; i < (l ? sh.num_ref_idx1 : sh.num_ref_idx); i++) { const int e = r.get(bits); if (e >= total) return DEC_ERR_INVALID; sh.list_entry[l][i] = (uint8_t)e; }
        }
      }
    }
    if (sh.is_b) sh.mvd_l1_zero = r.get(1);
    if (p.cabac_init_present) sh.cabac_init_flag = r.get(1);
    if (sh.tmvp) {
      if (sh.is_b) sh.collocated_from_l0 = r.get(1);
      const int n = sh.collocated_from_l0 ? sh.num_ref_idx : sh.num_ref_idx1;
      if (n > 1) { sh.collocated_ref_idx = r.ue(); if (sh.collocated_ref_idx < 0 || sh.collocated_ref_idx >= n) return DEC_ERR_INVALID; }
    }
    if (sh.is_b ? p.weighted_bipred : p.weighted_pred) {
      // pred_weight_table() (7.3.6.3; one layer: every entry's picture differs from the current one, so every flag is there) and 7.4.7.3
      sh.weighted = true;
      const int ld = (int)r.ue(), cd = ld + r.se();
      if (ld < 0 || ld > 7 || cd < 0 || cd > 7) return DEC_ERR_INVALID;
      sh.wt_log2[0] = (uint8_t)ld; sh.wt_log2[1] = (uint8_t)cd;
      for (int k = 0; k < 32; k++) { sh.wt[k].w[0] = (int16_t)(1 << ld); sh.wt[k].w[1] = sh.wt[k].w[2] = (int16_t)(1 << cd); sh.wt[k].o[0] = sh.wt[k].o[1] = sh.wt[k].o[2] = 0; }
      for (int l = 0; l < (sh.is_b ? 2 : 1); l++) {
        const int n = l ? sh.num_ref_idx1 : sh.num_ref_idx;
        uint32_t lf = 0, cf = 0;
        for (int i = 0; i < n; i++) lf |= (uint32_t)r.get(1) << i;
        for (int i = 0; i < n; i++) cf |= (uint32_t)r.get(1) << i;
        for (int i = 0; i < n; i++) {
          DecWt &e = sh.wt[l * 16 + i];
          if ((lf >> i) & 1) {
            const int dw = r.se(), lo = r.se();
            if (dw < -128 || dw > 127 || lo < -128 || lo > 127) return DEC_ERR_INVALID;
            e.w[0] = (int16_t)((1 << ld) + dw); e.o[0] = (int16_t)lo;
          }
          if ((cf >> i) & 1) for (int j = 0; j < 2; j++) {
            const int dw = r.se(), dof = r.se();
            if (dw < -128 || dw > 127 || dof < -512 || dof > 511) return DEC_ERR_INVALID;
            const int w = (1 << cd) + dw;
            e.w[1 + j] = (int16_t)w; e.o[1 + j] = (int16_t)clip3(-128, 127, 128 + dof - ((128 * w) >> cd));
          }
        }
      }
      if (r.err) return DEC_ERR_INVALID;
      for (int k = 0; k < 32; k++) { const DecWt &e = sh.wt[k]; if (e.w[0] != (1 << ld) || e.w[1] != (1 << cd) || e.w[2] != (1 << cd) || e.o[0] || e.o[1] || e.o[2]) sh.wt_explicit |= 1u << k; }
    }
    sh.max_merge = 5 - (int)r.ue();
    if (sh.max_merge < 1 || sh.max_merge > 5) return DEC_ERR_INVALID;
  }
  sh.slice_qp = p.init_qp + r.se();
  if (sh.slice_qp < 0 || sh.slice_qp > 51) return DEC_ERR_INVALID;
  sh.cb_qp_offset = p.cb_qp_offset; sh.cr_qp_offset = p.cr_qp_offset;
  if (p.slice_chroma_offsets) { sh.cb_qp_offset += r.se(); sh.cr_qp_offset += r.se(); }
  if (sh.cb_qp_offset < -12 || sh.cb_qp_offset > 12 || sh.cr_qp_offset < -12 || sh.cr_qp_offset > 12) return DEC_ERR_INVALID;
  sh.deblock_disabled = p.deblock_disabled; sh.beta_offset_div2 = p.beta_offset_div2; sh.tc_offset_div2 = p.tc_offset_div2;
  if (p.deblock_override && r.get(1)) {
    sh.deblock_disabled = r.get(1);
    if (!sh.deblock_disabled) { sh.beta_offset_div2 = r.se(); sh.tc_offset_div2 = r.se(); }
    if (sh.beta_offset_div2 < -6 || sh.beta_offset_div2 > 6 || sh.tc_offset_div2 < -6 || sh.tc_offset_div2 > 6) return DEC_ERR_INVALID;
  }
  bool across_slices = p.loop_filter_across_slices != 0;
  if (p.loop_filter_across_slices && (!sh.deblock_disabled || sh.sao_luma || sh.sao_chroma)) across_slices = r.get(1) != 0;
  const int wc = (s.width + (1 << s.ctb_log2) - 1) >> s.ctb_log2, hc = (s.height + (1 << s.ctb_log2) - 1) >> s.ctb_log2;      // the picture in coding tree blocks
  if (p.tile_rows > hc) return DEC_ERR_INVALID;
  DecPps pp = p;                                                 // tile row boundaries (6.5.1) for this picture size
  pp.row_bd[0] = 0;
  for (int k = 0; k < p.tile_rows; k++) {
    const int hgt = p.uniform_tiles ? ((k + 1) * hc) / p.tile_rows - (k * hc) / p.tile_rows : (k < p.tile_rows - 1 ? p.row_height[k] : hc - pp.row_bd[k]);
    if (hgt < 1) return DEC_ERR_INVALID;
    pp.row_bd[k + 1] = pp.row_bd[k] + hgt;
  }
  if (pp.row_bd[p.tile_rows] != hc) return DEC_ERR_INVALID;
  if (p.tile_cols > wc) return DEC_ERR_INVALID;
  pp.col_bd[0] = 0;
  for (int k = 0; k < p.tile_cols; k++) {
    const int wid = p.uniform_tiles ? ((k + 1) * wc) / p.tile_cols - (k * wc) / p.tile_cols : (k < p.tile_cols - 1 ? p.col_width[k] : wc - pp.col_bd[k]);
    if (wid < 1) return DEC_ERR_INVALID;
    pp.col_bd[k + 1] = pp.col_bd[k] + wid;
  }
  if (pp.col_bd[p.tile_cols] != wc) return DEC_ERR_INVALID;
  if (!first_seg) {
    // an independent slice of a picture under way: the same slice parameters as the first (what this decoder keeps per picture)
    const SliceHdr &a = open_job->sh;
    if (sh.is_intra != a.is_intra || sh.is_b != a.is_b || sh.num_ref_idx1 != a.num_ref_idx1 || sh.mvd_l1_zero != a.mvd_l1_zero || sh.collocated_from_l0 != a.collocated_from_l0 || sh.poc != a.poc || sh.tmvp != a.tmvp || sh.collocated_ref_idx != a.collocated_ref_idx || sh.sao_luma != a.sao_luma ||
        sh.sao_chroma != a.sao_chroma || sh.num_ref_idx != a.num_ref_idx || sh.cabac_init_flag != a.cabac_init_flag || sh.max_merge != a.max_merge ||
        (sh.slice_qp != a.slice_qp && (pp.tile_cols > 1 || pp.tile_rows > 1)) || sh.cb_qp_offset != a.cb_qp_offset || sh.cr_qp_offset != a.cr_qp_offset || sh.deblock_disabled != a.deblock_disabled ||
        sh.beta_offset_div2 != a.beta_offset_div2 || sh.tc_offset_div2 != a.tc_offset_div2 ||
        memcmp(sh.list_mod, a.list_mod, 2) || memcmp(sh.list_entry, a.list_entry, sizeof(sh.list_entry)) ||
        sh.wt_explicit != a.wt_explicit || sh.weighted != a.weighted || (sh.weighted && (memcmp(sh.wt, a.wt, sizeof(sh.wt)) || sh.wt_log2[0] != a.wt_log2[0] || sh.wt_log2[1] != a.wt_log2[1]))) return DEC_ERR_UNSUPPORTED;
    asm_cur_dependent_ = false; asm_cur_qp_ = sh.slice_qp;      // (inside one tile a slice may have its own SliceQpY: free slices, close_free_picture)
    asm_lf_.push_back(LfSlice{seg_address, across_slices});
    return append_segment(*open_job, r.pos, rbsp, len, p, pp, wc, hc, seg_address, pts);
  }
  if (!sh.is_intra && !seen_irap_) return DEC_ERR_INVALID;       // nothing to predict from before the first random access point
  if (band_nrows_ > 0 && s.ctb_log2 != 6) return DEC_ERR_UNSUPPORTED;      // (the tile-row split hands over bands of 64-sample rows)
  if (p.cu_qp_delta && p.qp_delta_depth > s.ctb_log2 - s.min_cb_log2) return DEC_ERR_INVALID;      // (7.4.3.3.1: a quantisation group is no smaller than the minimum coding block)
  if (!ensure_buffers(s.width, s.height, s.ctb_log2)) return DEC_ERR_GPU;
  // ---- reference picture set (8.3.2) and RefPicList0 (8.3.4): pictures not in the set stop being references
  if (no_rasl_out) for (auto &d : dpb_) d.is_ref = false;         // (8.3.2; what a CRA or BLA picture's set names is for its RASL pictures)
  int nref = 0, ref_poc[16]; uint8_t ref_slot[16], ref_lt[16] = {};
  int nref1 = 0, ref_poc1[16]; uint8_t ref_slot1[16], ref_lt1[16] = {};
  bool no_backward = true;
  if (!idr) {
    int cand_slot[32], nc = 0, nbefore = 0;             // the used pictures: those before the current one in output order (nearest first), then those after it, then the long-term ones
    bool keep[KVZ_DEC_MAX_REFS] = {false};
    // the long-term entries first, among all reference pictures (8.3.2): by the POC's LSBs, or by the whole POC when delta_poc_msb_present_flag says how many LSB
    // cycles back -- what they name is a long-term reference picture from now on; the short-term entries name pictures among the rest
    int lt_slot[16], nl = 0;
    const int max_lsb = 1 << s.log2_max_poc_lsb;
    for (int k = 0; k < nlt; k++) {
      const long long full = (long long)sh.poc - (long long)lt_cycle[k] * max_lsb - (sh.poc & (max_lsb - 1)) + lt_lsb[k];      // (64 bits: a hostile cycle count must not wrap into a POC that exists)
      int found = -1;
      for (int q = 0; q < KVZ_DEC_MAX_REFS; q++) if (dpb_[q].is_ref && dpb_[q].used && (lt_msb[k] ? dpb_[q].poc == full : (dpb_[q].poc & (max_lsb - 1)) == lt_lsb[k])) found = q;
      if (found < 0 && lt_used[k] && !sh.is_intra) {               // lost on the way: a grey picture stands in (known by its LSBs alone: the nearest POC before the current one that has them)
        long long at = full; if (!lt_msb[k]) { at = (long long)sh.poc - (sh.poc & (max_lsb - 1)) + lt_lsb[k]; if (at >= sh.poc) at -= max_lsb; }
        found = conceal_ref((int)at, true);
        if (found < 0) return DEC_ERR_INVALID;
      }
      if (found >= 0) { keep[found] = true; dpb_[found].is_lt = true; }
      if (lt_used[k]) { if (found < 0 && !sh.is_intra) return DEC_ERR_INVALID; if (found >= 0) lt_slot[nl++] = found; }
    }
    for (int k = 0; k < rps.n_neg + rps.n_pos; k++) {
      const int poc = sh.poc + rps.dpoc[k];
      int found = -1;
      for (int q = 0; q < KVZ_DEC_MAX_REFS; q++) if (dpb_[q].is_ref && dpb_[q].used && !dpb_[q].is_lt && dpb_[q].poc == poc) found = q;
      if (found < 0 && rps.used[k] && !sh.is_intra) {
        bool lt_has_it = false;                                  // (a long-term picture of that POC is no short-term entry's picture: that stays an error)
        for (int q = 0; q < KVZ_DEC_MAX_REFS; q++) lt_has_it |= dpb_[q].is_ref && dpb_[q].used && dpb_[q].is_lt && dpb_[q].poc == poc;
        if (!lt_has_it) found = conceal_ref(poc, false);
      }
      if (found >= 0) keep[found] = true;
      if (rps.used[k]) { if (found < 0 && !sh.is_intra) return DEC_ERR_INVALID; if (found >= 0 && nc < 16) { cand_slot[nc++] = found; if (k < rps.n_neg) nbefore = nc; } }      // a missing reference picture (lost access unit)
    }
    // stand-ins made above (conceal_ref): each a copy of the reference picture nearest in output order among those the DPB holds NOW, before this picture's set is
    // applied (with one reference picture per picture the set names the lost picture and nothing else), of two equally near the earlier one; no stand-in of this
    // same picture is a source.  (The source may be dropped by the set and become this picture's own buffer: the copy runs before the picture's kernels.)
    std::vector<int> fresh;
    for (const auto &c : pending_conceal_) if (c.second == -2) fresh.push_back(c.first);
    for (auto &c : pending_conceal_) if (c.second == -2) {
      int best = -1;
      for (int q = 0; q < KVZ_DEC_MAX_REFS; q++) {
        if (!dpb_[q].is_ref || !dpb_[q].used || std::find(fresh.begin(), fresh.end(), q) != fresh.end()) continue;
        const long long dq = llabs((long long)dpb_[q].poc - dpb_[c.first].poc), db = best < 0 ? 0 : llabs((long long)dpb_[best].poc - dpb_[c.first].poc);
        if (best < 0 || dq < db || (dq == db && dpb_[q].poc < dpb_[best].poc)) best = q;
      }
      c.second = best;                                            // (-1: none -- grey)
    }
    for (int q = 0; q < KVZ_DEC_MAX_REFS; q++) if (!keep[q]) dpb_[q].is_ref = false;
    if (!sh.is_intra) {
      const int nst = nc;                                 // (the short-term part of the temporary lists)
      for (int k = 0; k < nl && nc < 32; k++) cand_slot[nc++] = lt_slot[k];
      if (nc == 0 || nc > 16) return DEC_ERR_INVALID;
      // 8.3.4: RefPicList0 = before, after, long-term, repeated; RefPicList1 = after, before, long-term, repeated
      nref = sh.num_ref_idx;
      // (a modified list names entries of the temporary list; nc = NumPicTotalCurr here: a used picture that is missing ended the call above)
      for (int k = 0; k < nref; k++) {
        const int q = sh.list_mod[0] ? imin(sh.list_entry[0][k], nc - 1) : k % nc;
        ref_slot[k] = (uint8_t)cand_slot[q]; ref_lt[k] = (uint8_t)(q >= nst); ref_poc[k] = dpb_[ref_slot[k]].poc; if (ref_poc[k] > sh.poc) no_backward = false;
      }
      nref1 = sh.is_b ? sh.num_ref_idx1 : 0;
      const int nafter = nst - nbefore;
      for (int k = 0; k < nref1; k++) {
        const int q = sh.list_mod[1] ? imin(sh.list_entry[1][k], nc - 1) : k % nc;
        ref_slot1[k] = (uint8_t)cand_slot[q >= nst ? q : (q < nafter ? nbefore + q : q - nafter)]; ref_lt1[k] = (uint8_t)(q >= nst); ref_poc1[k] = dpb_[ref_slot1[k]].poc;
        if (ref_poc1[k] > sh.poc) no_backward = false;
      }
    }
  }
  const int slot = alloc_slot();
  if (slot < 0) return DEC_ERR_GPU;
  // ---- hand the picture to a parse job.  With frame threads (libOpenHevcInit thread_type FRAME / FRAMESLICE)
  // up to `frame_threads_` pictures are parsed concurrently on worker threads and the output is delayed accordingly,
  // like OpenHEVC's frame threading; temporal motion prediction makes a picture's parser follow the collocated
  // picture's parser row by row (ColMotion::row_done).
  PicJob &job = jobs_[(size_t)(job_head_ % jobs_.size())];
  job.rbsp.clear(); job.data_off = 0; job.data_len = 0; job.sub_start.clear(); job.expect_hash.clear();
  job.seg_end_row.assign((size_t)hc, 0); job.row_restart.assign((size_t)hc, SIZE_MAX);
  job.across_slices = across_slices;
  job.sh = sh; job.sps = sps_ref; job.pps = pp; job.pts = pts;
  job.crop[0] = s.crop_l; job.crop[1] = s.crop_r; job.crop[2] = s.crop_t; job.crop[3] = s.crop_b;
  if (no_crop_) job.crop[0] = job.crop[1] = job.crop[2] = job.crop[3] = 0;
  job.fps_num = s.fps_num ? s.fps_num : vps_fps_num_; job.fps_den = s.fps_num ? s.fps_den : vps_fps_den_;
  job.slot = slot; job.nref = nref;
  for (int k = 0; k < 16; k++) { job.ref_poc[k] = k < nref ? ref_poc[k] : sh.poc; job.ref_slot[k] = k < nref ? ref_slot[k] : 0; job.ref_lt[k] = k < nref ? ref_lt[k] : 0; job.ref_lt1[k] = k < nref1 ? ref_lt1[k] : 0; }
  if (no_rasl_out) cvs_++;
  job.starts_cvs = no_rasl_out; job.discard_prior = cur_discard_;
  job.conceal = std::move(pending_conceal_); pending_conceal_.clear();                     // a new coded video sequence: its pictures follow ALL of the last one's in output order
  job.cvs = cvs_;
  job.nref1 = nref1; job.no_backward = no_backward;
  for (int k = 0; k < 16; k++) { job.ref_poc1[k] = k < nref1 ? ref_poc1[k] : sh.poc; job.ref_slot1[k] = k < nref1 ? ref_slot1[k] : 0; }
  job.col.reset();
  if (sh.tmvp && !sh.is_intra) job.col = dpb_[(sh.is_b && !sh.collocated_from_l0) ? ref_slot1[sh.collocated_ref_idx] : ref_slot[sh.collocated_ref_idx]].motion;      // 8.5.3.2.8
  job.any_bi.store(0, std::memory_order_relaxed);
  if (sh.is_b) {                                                 // two-list motion for the parser's derivations, second vectors for the kernels
    const size_t nb4 = (size_t)(pw_ / 4) * (ph_ / 4);
    PicJob::MvF none; memset(&none, 0, sizeof(none)); none.ref[0] = none.ref[1] = -1;
    job.mvf.assign(nb4, none);
    if (job.b4x.size() != nb4) job.b4x.assign(nb4, B4L1());
  } else if (sh.weighted) {                                      // a P slice with pred_weight_table(): the blocks' weight table entries
    const size_t nb4 = (size_t)(pw_ / 4) * (ph_ / 4);
    if (job.b4x.size() != nb4) job.b4x.assign(nb4, B4L1());
  }
  job.own.reset();
  if (s.tmvp) {                                                  // (only streams with temporal prediction ever read it)
    job.own = std::make_shared<ColMotion>();
    job.own->w16 = (s.width + 15) / 16; job.own->h16 = (s.height + 15) / 16; job.own->hc = hc; job.own->poc = sh.poc;
    job.own->mv.resize((size_t)job.own->w16 * job.own->h16);
    job.own->row_done.reset(new std::atomic<uint8_t>[(size_t)hc]);
    for (int k = 0; k < hc; k++) job.own->row_done[(size_t)k].store(0, std::memory_order_relaxed);
  }
  for (int cy = 0, t = 0; cy < hc; cy++) {
    while (cy >= pp.row_bd[t + 1]) t++;
    // the kernels only ever ask whether two ADJACENT CTUs (side by side, above, diagonal) lie in the same tile: (row mod 16, column mod 16)
    // tells adjacent tiles apart in one byte for any grid (a row-major index would need 20 x 22 = 440 values)
    for (int tc = 0; tc < pp.tile_cols; tc++) memset(job.ctu_tile + (size_t)cy * wc + pp.col_bd[tc], ((t & 15) << 4) | (tc & 15), (size_t)(pp.col_bd[tc + 1] - pp.col_bd[tc]));
  }
  // the substreams in decoding order (6.5.1 tile scan): tile after tile; with WPP every CTB row of a tile is one
  job.geom.clear();
  for (int tr = 0; tr < pp.tile_rows; tr++)
    for (int tc = 0; tc < pp.tile_cols; tc++) {
      PicJob::SubGeom g; g.tile_cy0 = pp.row_bd[tr]; g.tile_cy1 = pp.row_bd[tr + 1]; g.cx0 = pp.col_bd[tc]; g.cx1 = pp.col_bd[tc + 1]; g.tc = tc;
      if (pp.wpp) for (int cy = g.tile_cy0; cy < g.tile_cy1; cy++) { g.cy0 = cy; g.cy1 = cy + 1; job.geom.push_back(g); }
      else { g.cy0 = g.tile_cy0; g.cy1 = g.tile_cy1; job.geom.push_back(g); }
    }
  job.seg_end_sub.assign(job.geom.size(), 0);
  if (job.own) { job.own->cols = pp.tile_cols; job.own->row_cols.reset(new std::atomic<uint8_t>[(size_t)hc]); for (int k = 0; k < hc; k++) job.own->row_cols[(size_t)k].store(0, std::memory_order_relaxed); }
  job.rc = 0; job.any_intra = job.any_inter = false;
  asm_active_ = true; asm_guessed_one_row_ = false; asm_rows_ = 0; asm_subs_ = 0; asm_pps_id_ = pps_id; asm_nal_type_ = nal_type; asm_irap_ = irap;
  asm_segs_.clear(); asm_free_ = false; asm_cur_dependent_ = false; asm_cur_qp_ = sh.slice_qp; job.ambiguous_end = false;
  asm_lf_.clear(); asm_lf_.push_back(LfSlice{0, across_slices}); job.lf_restricted = false; job.lf_slices.clear();
  job.ctb_cut.clear(); job.ctb_slice.clear(); job.ctb_data.clear(); job.slice_qps.clear();
  return append_segment(job, r.pos, rbsp, len, p, pp, wc, hc, 0, pts);
}

// The slice data of one segment joins the picture's job: entry points (the segment's substreams: whole CTU rows with WPP, else whole
// tiles -- or, for a dependent segment without either, one run of CTU rows inside the current substream), then the bytes.  The last
// segment submits the job.  `r` stands behind the part of the slice segment header that precedes the entry points.
int Decoder::append_segment(PicJob &job, size_t bitpos, const uint8_t *rbsp, size_t len, const DecPps &p, const DecPps &pp, int wc, int hc, int address, int64_t pts)
{
  (void)pts;
  BitReader r(rbsp, len); r.pos = bitpos;
  auto fail = [&](int rc) { asm_active_ = false; return rc; };
  if (pp.tile_cols > 1) return append_segment_tiles(job, r.pos, rbsp, len, p, pp, wc, hc, address);
  // One tile: a segment that Kvazaar's forms do not have -- one that begins inside a CTB row, an independent slice behind the picture's first -- makes the picture
  // (and the stream: free_stream_) one of FREE slices: its segments are only collected here; where each ends is where the next begins, and the picture is put
  // together when its access unit ends (close_free_picture).
  const bool one_tile = pp.tile_rows == 1;
  // (frame threads: a picture whose segments turn out not to reach its end is with a worker by then; it is taken back when the segment that shows as much arrives
  // -- decode_slice -- unless the ring has handed it on already.  A picture without WPP, whose first segment says nothing at all about how far it reaches, does not
  // take that chance: it always waits for the end of its access unit there -- the next NAL unit, one more picture of delay on top of the ring's; a picture that
  // then has ONE segment is parsed as ever, close_free_picture)
  const bool wait_always = one_tile && frame_threads_ > 1 && !p.wpp && band_nrows_ == 0 && !parse_only_;
  if (one_tile && !asm_free_ && (wait_always || free_stream_ || (!asm_segs_.empty() && !asm_cur_dependent_) || address % wc != 0 || (!p.wpp && address != asm_rows_ * wc))) {      // (without WPP no header says how many rows a segment has: one that begins elsewhere than guessed is no loss)
    if (band_nrows_ > 0) return fail(DEC_ERR_UNSUPPORTED);
    asm_free_ = true;
    if (!wait_always) free_stream_ = true;
  }
  if (!asm_free_ && address != asm_rows_ * wc) return fail(address % wc ? DEC_ERR_UNSUPPORTED : DEC_ERR_INVALID);     // whole CTU rows, in order
  if (asm_free_ && (address >= wc * hc || (asm_segs_.empty() ? address != 0 : address <= asm_segs_.back().address))) return fail(DEC_ERR_INVALID);
  std::vector<uint32_t> entry;
  if (p.wpp || p.tile_rows > 1) {
    const int nep = r.ue();
    if (nep < 0 || nep > 1024) return fail(DEC_ERR_INVALID);
    if (nep > 0) { int bits = r.ue() + 1; if (bits > 32) return fail(DEC_ERR_INVALID); for (int k = 0; k < nep; k++) entry.push_back(r.get(bits) + 1); }
  }
  if (p.header_extension) { const int n = r.ue(); if (n > 256) return fail(DEC_ERR_INVALID); for (int k = 0; k < n; k++) r.get(8); }
  if (!r.get(1)) return fail(DEC_ERR_INVALID);                   // byte_alignment()
  while (r.pos & 7) r.get(1);
  if (r.err) return fail(DEC_ERR_INVALID);
  // CTU rows the segment covers
  const int nss = (int)entry.size() + 1, row0 = asm_rows_;
  int rows = 0;
  bool mid_substream = false;                                    // no WPP, and the segment does not start a tile: it continues the tile's substream
  if (asm_free_) { if (nss > hc || (!p.wpp && nss != 1)) return fail(DEC_ERR_INVALID); }
  else if (p.wpp) rows = nss;
  else {
    int t = 0; while (t < pp.tile_rows && pp.row_bd[t] != row0) t++;
    if (t < pp.tile_rows) { if (t + nss > pp.tile_rows) return fail(DEC_ERR_INVALID); rows = pp.row_bd[t + nss] - row0; }
    else {
      // inside a tile: only a dependent segment can start here; how many rows it holds is not in its header -- one (the form the
      // synthesiser writes); a longer one fails at its first end_of_slice_segment_flag
      if (nss != 1) return fail(DEC_ERR_UNSUPPORTED);
      rows = 1; mid_substream = true;
    }
    if (!mid_substream && nss == 1 && pp.tile_rows == 1 && row0 == 0) {
      // one tile, no WPP: the first segment's length is unknown as well: the whole picture unless dependent segments follow
      rows = p.dependent_slices ? 1 : hc;
      asm_guessed_one_row_ = p.dependent_slices != 0;
    } else if (!mid_substream && nss == 1 && p.dependent_slices) rows = 1;      // a tile begun by one row; the rest follows as dependent segments
  }
  if (!asm_free_ && (rows < 1 || row0 + rows > hc)) return fail(DEC_ERR_INVALID);
  // Substream starts inside the unescaped slice data.  entry_point offsets count bytes of the NAL
  // unit payload INCLUDING emulation prevention bytes (7.4.7.1); epb_[] holds, for every removed
  // byte, how many unescaped payload bytes preceded it.
  const size_t hdr = r.pos >> 3, base = job.rbsp.size();
  if (hdr > len) return fail(DEC_ERR_INVALID);
  std::vector<size_t> starts(1, base);                           // where the segment's substreams begin in job.rbsp
  {
    size_t esc = hdr;                                            // escaped offset of the slice data in the payload
    for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] < hdr) esc++;
    for (uint32_t e : entry) {
      esc += e;
      size_t removed = 0;
      for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] + k < esc) removed++;     // epb k sits at escaped offset epb_[k] + k
      starts.push_back(base + esc - removed - hdr);
    }
  }
  job.rbsp.insert(job.rbsp.end(), rbsp + hdr, rbsp + len);
  job.data_off = 0; job.data_len = job.rbsp.size();
  if (one_tile) {
    if (asm_segs_.size() >= 1024) return fail(DEC_ERR_UNSUPPORTED);
    asm_segs_.push_back(FreeSeg{address, asm_cur_dependent_, asm_cur_qp_, starts});
  }
  if (asm_free_) return 0;                                       // (the access unit's end closes the picture: decode_nal_inner, close_open_picture)
  if (mid_substream) job.row_restart[(size_t)row0] = base;
  else job.sub_start.insert(job.sub_start.end(), starts.begin(), starts.end());
  asm_rows_ = row0 + rows;
  job.seg_end_row[(size_t)(asm_rows_ - 1)] = 1;
  if (asm_rows_ < hc) return 0;                                  // more segments to come: no output for this NAL unit
  asm_active_ = false;
  job.ambiguous_end = one_tile && band_nrows_ == 0;
  return submit_job(job, asm_nal_type_, asm_irap_);
}

// Tile columns: a slice segment is the whole picture or one or more whole tiles, in tile-scan order (independent slices; a Kvazaar
// peer's slices=tiles).  Progress is counted in substreams.
int Decoder::append_segment_tiles(PicJob &job, size_t bitpos, const uint8_t *rbsp, size_t len, const DecPps &p, const DecPps &pp, int wc, int hc, int address)
{
  (void)hc; (void)pp;
  BitReader r(rbsp, len); r.pos = bitpos;
  auto fail = [&](int rc) { asm_active_ = false; return rc; };
  const int nsub = (int)job.geom.size();
  if (asm_subs_ >= nsub) return fail(DEC_ERR_INVALID);
  const PicJob::SubGeom &g0 = job.geom[(size_t)asm_subs_];
  if (g0.cy0 != g0.tile_cy0 || address != g0.cy0 * wc + g0.cx0) return fail(DEC_ERR_UNSUPPORTED);        // segments start where a tile starts
  std::vector<uint32_t> entry;
  {
    const int nep = r.ue();
    if (nep < 0 || nep > 1024) return fail(DEC_ERR_INVALID);
    if (nep > 0) { int bits = r.ue() + 1; if (bits > 32) return fail(DEC_ERR_INVALID); for (int k = 0; k < nep; k++) entry.push_back(r.get(bits) + 1); }
  }
  if (p.header_extension) { const int n = r.ue(); if (n > 256) return fail(DEC_ERR_INVALID); for (int k = 0; k < n; k++) r.get(8); }
  if (!r.get(1)) return fail(DEC_ERR_INVALID);                   // byte_alignment()
  while (r.pos & 7) r.get(1);
  if (r.err) return fail(DEC_ERR_INVALID);
  const int nss = (int)entry.size() + 1, last = asm_subs_ + nss - 1;
  if (last >= nsub) return fail(DEC_ERR_INVALID);
  if (job.geom[(size_t)last].cy1 != job.geom[(size_t)last].tile_cy1) return fail(DEC_ERR_UNSUPPORTED);   // ... and end where one ends
  const size_t hdr = r.pos >> 3, base = job.rbsp.size();
  if (hdr > len) return fail(DEC_ERR_INVALID);
  job.sub_start.push_back(base);
  {
    size_t esc = hdr;
    for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] < hdr) esc++;
    for (uint32_t e : entry) {
      esc += e;
      size_t removed = 0;
      for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] + k < esc) removed++;
      job.sub_start.push_back(base + esc - removed - hdr);
    }
  }
  job.rbsp.insert(job.rbsp.end(), rbsp + hdr, rbsp + len);
  job.data_off = 0; job.data_len = job.rbsp.size();
  asm_subs_ += nss;
  job.seg_end_sub[(size_t)last] = 1;
  if (asm_subs_ < nsub) return 0;
  asm_active_ = false;
  return submit_job(job, asm_nal_type_, asm_irap_);
}

void Decoder::take_back_job(PicJob &job)
{
  DpbPic &d = dpb_[job.slot];
  d.poc = job.undo.poc; d.is_ref = job.undo.is_ref; d.used = job.undo.used; d.decode_idx = job.undo.decode_idx; d.motion = job.undo.motion; prev_poc_ = job.undo.prev_poc; seen_irap_ = job.undo.seen_irap;
  vwait_ = std::move(job.undo.vwait);
  job.undo.motion.reset();
  job_head_--; job.state.store(0, std::memory_order_relaxed); job.rc = 0; job.early_dst = nullptr;
  asm_active_ = true; asm_free_ = free_stream_ = true;
}

// closed boundaries inside the picture?  (7.4.3.3.1 loop_filter_across_tiles_enabled_flag = 0 with more than one tile; 7.4.7.1 a slice with
// slice_loop_filter_across_slices_enabled_flag = 0 in a picture of several slices.)  The picture's slices go with the job; parse_job lays the map out.
void Decoder::note_lf_restrictions(PicJob &job)
{
  bool closed = (job.pps.tile_rows > 1 || job.pps.tile_cols > 1) && !job.pps.across_tiles;
  if (asm_lf_.size() > 1) for (const LfSlice &s : asm_lf_) closed |= !s.across;
  job.lf_restricted = closed && band_nrows_ == 0;
  job.lf_slices.clear();
  if (job.lf_restricted) for (const LfSlice &s : asm_lf_) job.lf_slices.emplace_back(s.address, (uint8_t)s.across);
}

int Decoder::submit_job(PicJob &job, int nal_type, bool irap)
{
  note_lf_restrictions(job);
  if (job.lf_slices.empty() && band_nrows_ > 0 && (job.pps.tile_rows > 1 || job.pps.tile_cols > 1) && !job.pps.across_tiles) return DEC_ERR_UNSUPPORTED;      // (the tile-row split exchanges halos FOR the filters)
  const SliceHdr &sh = job.sh;
  DpbPic &d = dpb_[job.slot];
  // (PicJob::ambiguous_end: when the picture's last segment ends before the picture does, more segments are to come -- everything is put back as it was before
  // the call and the picture is open again, as one of free slices; append_segment collects the rest, the end of the access unit closes it.  The synchronous decoder
  // knows at once; with frame threads the worker finds out and decode_slice looks when the next segment arrives.)
  job.undo.poc = d.poc; job.undo.prev_poc = prev_poc_; job.undo.is_ref = d.is_ref; job.undo.used = d.used; job.undo.seen_irap = seen_irap_; job.undo.decode_idx = d.decode_idx; job.undo.motion = d.motion;
  auto take_back = [&] { take_back_job(job); return 0; };
  d.poc = sh.poc; d.is_ref = true; d.used = true; d.is_lt = false; d.decode_idx = job_head_; d.motion = job.own;
  if (cur_tid_ == 0 && (nal_type > 9 || ((nal_type & 1) && nal_type < 6))) prev_poc_ = sh.poc;   // prevTid0Pic (8.3.1): TemporalId 0, not RASL / RADL / sub-layer non-reference
  if (irap) seen_irap_ = true;
  {
    // output bookkeeping in decoding order (C.5.2.2, C.5.2.3; Decoder::vwait_)
    job.undo.vwait = vwait_; job.serial = ++pic_serial_;
    const int nr = job.sps->num_reorder;
    auto bump = [&] { size_t b = 0; for (size_t i = 1; i < vwait_.size(); i++) if (vwait_[i].second < vwait_[b].second) b = i; vwait_.erase(vwait_.begin() + (long)b); };
    if (job.starts_cvs) {
      if (job.discard_prior && !vwait_.empty()) {
        for (const auto &v : vwait_) discarded_.push_back(v.first);
        for (size_t i = 0; i < reorder_q_.size();) {             // (the ones that have completed wait here; the others find their number in discarded_ when they complete)
          if (std::find(discarded_.begin(), discarded_.end(), reorder_q_[i].pic.pic.serial) != discarded_.end()) { owned_release(reorder_q_[i].pic.dev); reorder_q_.erase(reorder_q_.begin() + (long)i); }
          else i++;
        }
        if (discarded_.size() > 64) discarded_.erase(discarded_.begin(), discarded_.end() - 64);
      }
      vwait_.clear();
    } else while ((int)vwait_.size() > nr) bump();
    if (!sh.no_output) { vwait_.emplace_back(job.serial, sh.poc); while ((int)vwait_.size() > nr) bump(); }
  }
  job_head_++;
  if (parse_only_) {
    auto t0 = std::chrono::steady_clock::now();
    job.rc = parse_job(job, parse_threads_ > 1);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (job.rc == DEC_SEG_ENDS_EARLY) return take_back();
    job_tail_ = job_head_;
    if (job.rc < 0) return job.rc;
    probe_book(job, ms);
    return 0;
  }
  if (frame_threads_ == 1) {
    auto t0 = std::chrono::steady_clock::now();
    job.early_dst = nullptr; job.early_rows.store(0, std::memory_order_relaxed);
    {
      // (decoder.h PicJob::early_dst; the buffer is the one launch_gpu will pick -- nothing is launched between here and there)
      static const bool early_off = [] { const char *e = getenv("KVAZZUP_AMD_DEC_EARLY_UP"); return e && atoi(e) == 0; }();
      const int ib = (int)(launched_ % (gpu_depth_ + 1));
      if (!early_off && gpu_depth_ == 1 && band_nrows_ == 0 && job.pps.tile_cols == 1 && !job.lf_restricted && !(batch_attached_ && DecBatcher::get(device_).active()) && d_in_[ib] && d_in_cap_[ib] >= fixed_bytes()) job.early_dst = d_in_[ib];      // (lf_restricted: parse_job's last step still changes records)
    }
    job.rc = parse_job(job, true);
    if (job.rc == DEC_SEG_ENDS_EARLY) { if (job.early_dst) hipStreamSynchronize(stream_up_); return take_back(); }
    if (profiling_) { k_ms_[DK_HOST_PARSE] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); k_n_[DK_HOST_PARSE]++; }
    job.state.store(2, std::memory_order_release);
  } else {
    job.state.store(1, std::memory_order_release);
    if (!workers_) workers_.reset(new FrameWorkers(frame_threads_));
    PicJob *jp = &job;
    workers_->submit([this, jp] {
      auto t0 = std::chrono::steady_clock::now();
      // a large picture (an intra picture: several milliseconds on one core, which the pictures behind it in the ring cannot
      // hide) has its substreams parsed side by side on the row pool; small ones stay on this worker
      bool rows = false;
      // (64 KB: a 4K intra picture at QP 32 is ~125 KB and ~10 ms on one core -- longer than the eleven pictures behind it in a ring
      // of twelve take -- a 1080p one ~37 KB and stays on its worker: at 1080p every core is busy anyway)
      static const size_t row_parse_bytes = [] { const char *e = getenv("KVAZZUP_AMD_ROWPARSE_KB"); return (size_t)(e ? atoi(e) : 64) << 10; }();
      if (jp->data_len > row_parse_bytes && pool_mutex_.try_lock()) rows = true;
      jp->rc = parse_job(*jp, rows);
      if (rows) pool_mutex_.unlock();
      jp->parse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      jp->state.store(2, std::memory_order_release);
      futex_wake_all(jp->state);
    });
  }
  if (job_head_ - job_tail_ < frame_threads_ - (gpu_depth_ - 1)) return 0;          // pipeline still filling: no output for this NAL
  return finish_oldest();
}

// set_parse_only: what launch_gpu would upload for this picture, folded into the running digest
void Decoder::probe_book(PicJob &job, double ms)
{
  const size_t tu_off = fixed_bytes(), lev_off = (tu_off + job.ntu * sizeof(DecTu) + 15) & ~(size_t)15;
  if (const char *dump = getenv("KVAZZUP_AMD_PROBE_DUMP")) {      // (debugging aid: every picture's 4x4 records and intra modes, rows of pw / 4, appended to the file)
    if (FILE *fp = fopen(dump, "ab")) {
      const int32_t hdr[4] = {pw_ / 4, ph_ / 4, w_ / 4, h_ / 4};
      fwrite(hdr, sizeof(hdr), 1, fp); fwrite(job.b4, sizeof(B4Rec), (size_t)(pw_ / 4) * (ph_ / 4), fp); fwrite(job.intra_mode.data(), 1, (size_t)(pw_ / 4) * (ph_ / 4), fp);
      fclose(fp);
    }
  }
  uint64_t d = probe_.digest;
  auto fold = [&](const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) { d ^= p[i]; d *= 0x100000001b3ull; } };
  fold(job.h_in, off_scaling());                                              // records, region / CTU tables, tile ids, SAO parameters
  fold(job.h_in + tu_off, job.ntu * sizeof(DecTu));
  fold(job.h_in + lev_off, job.nlev * sizeof(uint32_t));
  probe_.digest = d; probe_.pictures++; probe_.tus += job.ntu; probe_.levels += job.nlev; probe_.parse_ms += ms;
}

// Waits for the oldest submitted picture to be parsed, reconstructs it on the GPU and makes it the output.
// Output stage.  Synchronous mode (frame_threads_ == 1): the picture just parsed is reconstructed and output.
// Frame-threaded mode: first the picture launched by the previous call is completed and becomes the output,
// then the oldest parsed picture is launched -- its kernels run while this thread goes on parsing headers.
int Decoder::finish_oldest()
{
  if (parse_only_) return 0;
  int rc_launch = 0;
  bool launched = false;
  if (job_head_ != job_tail_) {
    PicJob &job = jobs_[(size_t)(job_tail_ % jobs_.size())];
    job_tail_++;
    { Tick tk; for (int st; (st = job.state.load(std::memory_order_acquire)) != 2;) futex_wait(job.state, st); t_wait_ += tk.ms(); }
    job.state.store(0, std::memory_order_relaxed);
    if (frame_threads_ > 1 && profiling_) { k_ms_[DK_HOST_PARSE] += job.parse_ms; k_n_[DK_HOST_PARSE]++; }
    if (job.parse_ms > t_parse_max_) t_parse_max_ = job.parse_ms;
    // the next picture's kernels are queued BEFORE an earlier picture is waited for: the GPU goes from one to the other without
    // this thread's launch latency in between
    tl("dlaunch0", job.pts);
    rc_launch = job.rc < 0 ? (job.rc == DEC_SEG_ENDS_EARLY ? DEC_ERR_UNSUPPORTED : job.rc) : launch_gpu(job);      // (a picture whose last segment ended early and whose rest never came)
    // a picture that could not be parsed (damaged on the way) is never reconstructed: its buffer is no reference picture -- the pictures that name it find it
    // missing and get a stand-in (conceal_ref) instead of whatever the buffer held.  (Frame threads: pictures submitted before this was known keep the buffer.)
    if (job.rc < 0 && dpb_[job.slot].is_ref && dpb_[job.slot].poc == job.sh.poc && dpb_[job.slot].decode_idx == job_tail_ - 1) dpb_[job.slot].is_ref = false;
    tl("dlaunch1", job.pts);
    launched = rc_launch >= 0;
  }
  int produced = 0;
  { const int rc = start_ready_downloads(); if (rc < 0) return rc; }
  tl("ddl", 0);
  // frame-threaded mode: gpu_depth_ pictures stay queued on the GPU; a call that launches nothing (end of sequence / drain) takes one out
  if (!gpu_q_.empty() && (!launched || (int)gpu_q_.size() > gpu_depth_)) {
    PicJob *j = gpu_q_.front(); gpu_q_.pop_front();
    const int rc = complete_gpu(*j);
    tl("dcomplete", j->pts);
    if (rc < 0) return rc;
    produced = rc;
  }
  if (rc_launch < 0) return rc_launch;
  if (frame_threads_ == 1 && !gpu_q_.empty()) { PicJob *j = gpu_q_.front(); gpu_q_.pop_front(); const int rc = complete_gpu(*j); if (rc < 0) return rc; produced |= rc; }
  return produced;
}

// what libOpenHevcGetOutput / kvzx_decoder_output_device say about the picture of `job`; buf: its host buffer (download mode) or -1.
// The host buffer holds the picture buffer as it is -- coded size, Y | Cb | Cr back to back, pitch = coded width -- so that it comes down in one
// copy; the cropped picture is addressed through the plane pointers and pitches, as with any decoder's frame (openhevcfilter.cpp:209,224-227
// reads chroma row i/2 at pvU + i * (nUPitch / 2): the pitches are even).
void Decoder::describe_output(const PicJob &job, DecodedPicture &o, int buf) const
{
  o = DecodedPicture();
  o.coded_w = w_; o.coded_h = h_;
  o.width = w_ - job.crop[0] - job.crop[1]; o.height = h_ - job.crop[2] - job.crop[3];
  o.poc = job.sh.poc; o.pts = job.pts; o.is_intra = job.sh.is_intra; o.cvs = job.cvs; o.num_reorder = job.sps->num_reorder; o.serial = job.serial;
  o.fps_num = job.fps_num; o.fps_den = job.fps_den;
  for (int c = 0; c < 3; c++) {
    const int pw = c ? pw_ / 2 : pw_, ox = c ? job.crop[0] / 2 : job.crop[0], oy = c ? job.crop[2] / 2 : job.crop[2];
    const size_t off = (size_t)oy * pw + ox;
    o.dev[c] = dpb_[job.slot].plane[c] + off; o.dev_pitch[c] = pw;
    if (buf >= 0) { o.host[c] = h_out_[buf] + (dpb_[job.slot].plane[c] - dpb_[job.slot].plane[0]) + off; o.host_pitch[c] = pw; }
  }
}

// The picture buffer of `job` -> host buffer job.dl_buf: one copy-engine transfer on the download stream.  Only called when the picture's
// kernels are KNOWN to have finished (the caller has seen job.done): the copy command then carries no dependency, the copy engine takes
// it at once and nothing waits inside a hardware queue.  (Copies that waited on an event in the stream were executed by the runtime as
// blit kernels, four per picture with a barrier each; a kernel that stores across PCIe -- tried too -- slows every kernel running beside it
// by a factor of two to ten, tools/measure/pcie_copy_vs_kernels.hip.  The copy engine disturbs nothing.)
int Decoder::queue_download(PicJob &job)
{
  job.dl_buf = (int)(job.launch_idx % kOutRing);
  if (!h_out_[job.dl_buf] && hipHostMalloc(&h_out_[job.dl_buf], h_out_cap_, hipHostMallocDefault) != hipSuccess) { h_out_[job.dl_buf] = nullptr; return DEC_ERR_GPU; }
  if (!job.dl_done && hipEventCreateWithFlags(&job.dl_done, hipEventDisableTiming) != hipSuccess) return DEC_ERR_GPU;
  // rows above the crop window are not needed, rows below the picture's last row neither: the copy covers the planes from the first to the last row used
  const size_t npx = (size_t)pw_ * ph_;
  const size_t used = npx + npx / 4 + (size_t)(pw_ / 2) * ((h_ + 1) / 2);             // up to the end of the last Cr row
  if (hipMemcpyAsync(h_out_[job.dl_buf], dpb_[job.slot].plane[0], used, hipMemcpyDeviceToHost, stream_dl_) != hipSuccess) return DEC_ERR_GPU;
  dpb_[job.slot].last_dl = job.dl_done;
  return hipEventRecord(job.dl_done, stream_dl_) == hipSuccess ? 0 : DEC_ERR_GPU;
}

// frame-threaded download mode: start the copy of every queued picture whose kernels have finished (oldest first).  (Queueing the copy at launch instead, behind
// the picture's event on the download stream, was measured in round 6 and HALVES the host-boundary rate -- 3 200 against 6 290 frames/s at 1080p, 1 060 against
// 1 995 at 4K: hipMemcpyAsync to the host behind an unresolved hipStreamWaitEvent holds the calling thread, 135 ms of launches instead of 22 per 384 pictures;
// profiles/r06_dl_at_launch_ab.txt.)
int Decoder::start_ready_downloads()
{
  if (!download_) return 0;
  for (PicJob *j : gpu_q_) {
    if (j->dl_buf >= 0) continue;
    if (!j->launched.load(std::memory_order_acquire)) break;      // (still in the submission layer's queue: its event has not been recorded)
    const hipError_t r = hipEventQuery(j->done);
    if (r == hipErrorNotReady) break;
    if (r != hipSuccess) return DEC_ERR_GPU;
    const int rc = queue_download(*j);
    if (rc < 0) return rc;
  }
  return 0;
}

int Decoder::complete_gpu(PicJob &job)
{
  {
    Tick tk;
    while (!job.launched.load(std::memory_order_acquire)) futex_wait(job.launched, 0);      // (batch.h: the submitter thread records job.done)
    auto wait = [&](hipEvent_t ev) {
      if (frame_threads_ > 1 && !spin_wait_)    // the output lags anyway: nap between queries instead of polling (see nap_until)
        return nap_until([&] { hipError_t r = hipEventQuery(ev); return r == hipSuccess ? 1 : (r == hipErrorNotReady ? 0 : -1); });
      return hipEventSynchronize(ev) == hipSuccess;
    };
    if (download_ && job.dl_buf < 0) {          // its copy has not been started yet (synchronous decoder; the GPU queue was short): kernels first
      if (!wait(job.done)) return DEC_ERR_GPU;
      const int rc = queue_download(job); if (rc < 0) return rc;
    }
    if (!wait(download_ ? job.dl_done : job.done)) return DEC_ERR_GPU;
    t_sync_ += tk.ms();
  }
  if (*h_err_) { fprintf(stderr, "kvazzup_amd: decoder device error flags 0x%x\n", *h_err_); return DEC_ERR_GPU; }
  if (!job.expect_hash.empty()) {                  // libOpenHevcSetCheckMD5: the picture's hash SEI arrived while it was in the ring
    const std::vector<uint8_t> want = std::move(job.expect_hash); job.expect_hash.clear();
    const int rc = verify_hash(job, want);
    if (rc < 0) last_error_ = rc;                   // (the picture is still handed out: the application decides, as with OpenHEVC)
  }
  if (job.ev_used) {
    for (size_t i = 0; i < job.ev_used; i++) { float ms = 0; hipEventElapsedTime(&ms, job.ev[i].a, job.ev[i].b); k_ms_[job.ev[i].id] += ms; k_n_[job.ev[i].id]++; }
    job.ev_used = 0;
  }
  if (job.sh.no_output) { job.dl_buf = -1; return 0; }      // pic_output_flag = 0 (7.4.7.1): reconstructed -- later pictures predict from it -- and never handed out
  if (!discarded_.empty() && std::find(discarded_.begin(), discarded_.end(), job.serial) != discarded_.end()) { job.dl_buf = -1; return 0; }      // (an IDR / BLA picture behind it said so: C.5.2.2)
  describe_output(job, out_, download_ ? job.dl_buf : -1);
  out_slot_ = job.slot;
  job.dl_buf = -1;
  pic_ready_ = true;
  return 1;
}

// ------------------------------------------------------------------------------------------ slice data (7.3.8)
// One task per substream -- a CTU row with WPP, else a tile -- run by a pool of host threads.  With WPP row r follows row r-1
// at a distance of two CTUs: it starts from the context states saved after the second CTU of the row above and needs that row's
// records up to the above-right CTU.
int Decoder::parse_substream(PicJob &job, int sub, const uint8_t *data, size_t len, SubOut &out)
{
  SliceParser sp(job, out, pw_);
  tl("row0", sub);
  struct RowEnd { int s; ~RowEnd() { tl("row1", s); } } row_end_{sub};
  if (job.sh.is_b) sp.mvf = job.mvf.data();
  const int wc = sp.wc;
  const DecPps &pps = job.pps; const SliceHdr &sh = job.sh;
  const bool wpp = pps.wpp != 0;
  const PicJob::SubGeom g = job.geom[(size_t)sub];
  const int first_cy = g.cy0, ncy = g.cy1 - g.cy0, cx0 = g.cx0, cx1 = g.cx1, tw = cx1 - cx0, cols = pps.tile_cols;
  auto tile_starts_at = [&](int cy) { return cy == g.tile_cy0; };
  auto tile_ends_at = [&](int cy) { return cy + 1 == g.tile_cy1; };
  int seen_above = 0;                                  // last observed progress of the row above inside the tile (monotonic)
  auto wait_above = [&](int cy, int need) {            // CTUs of the tile's row cy-1 that must be complete
    if (!wpp || tile_starts_at(cy)) return true;       // nothing above inside the tile
    if (need > tw) need = tw;
    if (seen_above < need) {
      std::atomic<int> &p = job.row_progress[(size_t)(cy - 1) * cols + g.tc].v;
      int spins = 0;
      while ((seen_above = p.load(std::memory_order_acquire)) < need) {
        if (++spins < 2000) __builtin_ia32_pause(); else { g_yields.fetch_add(1, std::memory_order_relaxed); std::this_thread::yield(); }
      }
    }
    return seen_above < (1 << 29);                     // >= 1 << 29: that row failed
  };
  const int init_type = sh.is_intra ? 0 : (sh.is_b ? (sh.cabac_init_flag ? 1 : 2) : (sh.cabac_init_flag ? 2 : 1));      // 9.3.2.2: cabac_init_flag swaps the P and the B tables
  CabacDec &c = sp.c;
  c.start(data, len);
  // free slices (PicJob::ctb_cut; one tile): the slice of every coding tree block, its SliceQpY
  const bool free_slices = !job.ctb_cut.empty();
  sp.slice_qp = sh.slice_qp;
  if (free_slices) { sp.slice_of = job.ctb_slice.data(); sp.cur_slice = job.ctb_slice[(size_t)first_cy * wc + cx0]; sp.slice_qp = job.slice_qps[(size_t)sp.cur_slice]; }
  auto init_contexts = [&] { uint8_t init[CTX_COUNT]; cabac_init_contexts(init, init_type, sp.slice_qp); c.load_ctx(init); };
  // 9.3.1: the first CTB of a tile initialises the contexts; a WPP row takes them over from the row above after its second
  // CTB when that CTB exists (pictures one CTB wide: it does not, and the row initialises afresh) -- and is AVAILABLE: with free slices it may
  // belong to another slice; then a dependent segment that begins with this row goes on where the segment before it stopped (the end of the row
  // above), anything else initialises
  if (!wpp || tile_starts_at(first_cy) || tw < 2) init_contexts();
  else if (free_slices && job.ctb_slice[(size_t)(first_cy - 1) * wc + cx0 + 1] != sp.cur_slice) {
    if (job.ctb_cut[(size_t)first_cy * wc + cx0] & 2) {
      if (!wait_above(first_cy, tw)) return DEC_ERR_INVALID;
      c.load_ctx(&job.ds_saved[(size_t)(first_cy - 1) * CTX_COUNT]);
    } else init_contexts();
  } else {
    if (!wait_above(first_cy, 2)) return DEC_ERR_INVALID;
    c.load_ctx(&job.wpp_saved[((size_t)(first_cy - 1) * cols + g.tc) * CTX_COUNT]);
  }
  sp.last_qp_y = sp.slice_qp;                          // qPY_PREV at the start of a slice, a tile, a CTB row with WPP (8.6.1)
  sp.tile_y0 = g.tile_cy0 << ctbl_; sp.tile_y1 = g.tile_cy1 << ctbl_; sp.tile_x0 = cx0 << ctbl_; sp.tile_x1 = cx1 << ctbl_;
  if (band_nrows_ > 0) {
    if (band_row0_ > 0) sp.ref_y0 = band_row0_ * 64 - 4;
    if (band_row0_ + band_nrows_ < (h_ + 63) / 64) sp.ref_y1 = (band_row0_ + band_nrows_) * 64;
  }
  ColMotion *own = job.own.get();
  for (int cy = first_cy; cy < first_cy + ncy; cy++) {
    if (cy > first_cy && job.row_restart[(size_t)cy] != SIZE_MAX) {
      // a dependent slice segment begins inside this substream: new arithmetic codeword, the contexts go on (9.3.1)
      const uint8_t *q = job.rbsp.data() + job.data_off + job.row_restart[(size_t)cy];
      if (q < data || q >= data + len) return DEC_ERR_INVALID;
      c.start(q, (size_t)(data + len - q));
    }
    for (int cx = cx0; cx < cx1; cx++) {
      if (!wait_above(cy, cx - cx0 + 2)) return DEC_ERR_INVALID;
      const int ctu = cy * wc + cx;
      const uint8_t cut = free_slices ? job.ctb_cut[(size_t)ctu] : 0;
      if (cut & 3) {
        // a slice segment begins with this block: its own arithmetic codeword (the substream's first block: started above); an independent slice is a new
        // slice for every availability rule, starts from the initial context states and from its own SliceQpY; a dependent one goes on with the states at hand
        if (cx != cx0 || cy != first_cy) {
          const uint8_t *q = job.rbsp.data() + job.data_off + job.ctb_data[(size_t)ctu];
          if (q < data || q >= data + len) return DEC_ERR_INVALID;
          c.start(q, (size_t)(data + len - q));
        }
        if (cut & 1) {
          sp.cur_slice = job.ctb_slice[(size_t)ctu]; sp.slice_qp = job.slice_qps[(size_t)sp.cur_slice];
          init_contexts();
          sp.last_qp_y = sp.slice_qp;
        }
      }
      const uint32_t tu0 = (uint32_t)out.tus.size();
      sp.ctu_intra_mask = 0;
      if (!pps.cu_qp_delta) { sp.qp_y_pred = sp.slice_qp; sp.cu_qp_delta_val = 0; }
      if (sh.sao_luma || sh.sao_chroma) {                  // sao() (7.3.8.3) opens the CTU
        SaoParams *s = &job.sao[ctu];
        const SaoParams *left = cx > cx0 ? s - 1 : nullptr, *up = (cy > 0 && !tile_starts_at(cy)) ? s - wc : nullptr;
        if (free_slices) { if (left && job.ctb_slice[(size_t)ctu - 1] != sp.cur_slice) left = nullptr; if (up && job.ctb_slice[(size_t)ctu - wc] != sp.cur_slice) up = nullptr; }      // (merging stays inside the slice)
        parse_sao(c, *s, left, up, sh.sao_luma != 0, sh.sao_chroma != 0);
      }
      // (measurement aid, tools/measure/wpp_critical_path.py: KVAZZUP_AMD_CTU_DUMP=<file> -- picture, row, column, nanoseconds of every coding tree unit's parse)
      static const char *const ctu_dump = getenv("KVAZZUP_AMD_CTU_DUMP");
      std::chrono::steady_clock::time_point ctu_t0;
      if (__builtin_expect(ctu_dump != nullptr, 0)) ctu_t0 = std::chrono::steady_clock::now();
      sp.coding_quadtree(cx << ctbl_, cy << ctbl_, ctbl_, 0);
      if (__builtin_expect(ctu_dump != nullptr, 0)) {
        const long ns = (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ctu_t0).count();
        static FILE *const fp = fopen(ctu_dump, "w"); static std::mutex m;
        if (fp) { std::lock_guard<std::mutex> l(m); fprintf(fp, "%d %d %d %ld\n", job.sh.poc, cy, cx, ns); fflush(fp); }
      }
      if (sp.err) return sp.err;
      if (c.overrun()) return DEC_ERR_INVALID;
      job.ctu[ctu].first = tu0;
      job.ctu[ctu].count = ((uint32_t)out.tus.size() - tu0) | (sp.ctu_intra_mask << 24);
      if (out.tus.size() - tu0 >= (1u << 24)) return DEC_ERR_INVALID;
      if (wpp && cx == cx0 + 1) c.save_ctx(&job.wpp_saved[((size_t)cy * cols + g.tc) * CTX_COUNT]);
      if ((cut & 4) && wpp && cx == cx1 - 1) c.save_ctx(&job.ds_saved[(size_t)cy * CTX_COUNT]);      // (a segment ends with the row: what a dependent segment that begins the next row may have to go on with)
      if (wpp) job.row_progress[(size_t)cy * cols + g.tc].v.store(cx - cx0 + 1, std::memory_order_release);
      // end_of_slice_segment_flag: 1 exactly where the picture's slice segments end (decode_slice noted the rows; one segment: the
      // last CTU of the picture); inside a segment a substream ends with end_of_subset_one_bit
      const bool seg_last = free_slices ? (cut & 4) != 0 : cx == cx1 - 1 && (cols > 1 ? (cy == g.cy1 - 1 && job.seg_end_sub[(size_t)sub]) : job.seg_end_row[(size_t)cy] != 0);
      const int end = c.terminate();
      if (end != (seg_last ? 1 : 0)) return seg_last ? DEC_ERR_INVALID : (job.ambiguous_end ? DEC_SEG_ENDS_EARLY : DEC_ERR_UNSUPPORTED);      // (a segment that ends elsewhere: not whole CTU rows / tiles)
      if (!seg_last && cx == cx1 - 1 && (wpp || tile_ends_at(cy)) && !c.terminate()) return DEC_ERR_INVALID;   // end_of_subset_one_bit
    }
    if (job.early_dst) {
      // the row's 4x4 records are final (a coding unit writes inside its own CTU only): up they go, from whichever thread parsed the row
      const size_t rowb = (size_t)(4 << (ctbl_ - 4)) * (pw_ / 4) * sizeof(B4Rec), off = (size_t)cy * rowb;      // (a CTB row: CTB / 4 rows of records)
      if (hipSetDevice(device_) == hipSuccess && hipMemcpyAsync(job.early_dst + off, job.h_in + off, rowb, hipMemcpyHostToDevice, stream_up_) == hipSuccess)
        job.early_rows.fetch_add(1, std::memory_order_acq_rel);
    }
    // this CTB row's motion (the tile's columns of it) as later pictures see it (one entry per 16x16 block)
    if (!own) continue;
    const int per16 = 1 << (ctbl_ - 4);                     // 16x16 blocks a CTB is wide
    for (int y16 = cy * per16; y16 < (cy + 1) * per16 && y16 < own->h16; y16++)
      for (int x16 = cx0 * per16; x16 < cx1 * per16 && x16 < own->w16; x16++) {
        const size_t i4 = (size_t)(y16 * 4) * (pw_ / 4) + x16 * 4;
        const B4Rec &m = job.b4[i4];
        ColMotion::Mv &o = own->mv[(size_t)y16 * own->w16 + x16];
        memset(&o, 0, sizeof(o));
        if (m.ref_idx < 0) continue;                           // intra
        if (sh.is_b) {
          const PicJob::MvF &f = job.mvf[i4];
          for (int L = 0; L < 2; L++) if (f.ref[L] >= 0) { o.used |= (uint8_t)(1 << L); o.mv[L][0] = f.mv[L][0]; o.mv[L][1] = f.mv[L][1]; o.ref_poc[L] = L ? job.ref_poc1[f.ref[L] & 15] : job.ref_poc[f.ref[L] & 15]; if ((L ? job.ref_lt1 : job.ref_lt)[f.ref[L] & 15]) o.lt |= (uint8_t)(1 << L); }
        } else { o.used = 1; o.mv[0][0] = m.mvx; o.mv[0][1] = m.mvy; o.ref_poc[0] = job.ref_poc[m.ref_idx & 15]; o.lt = job.ref_lt[m.ref_idx & 15] ? 1 : 0; }
      }
    if (own->row_cols[(size_t)cy].fetch_add(1, std::memory_order_acq_rel) + 1 >= own->cols) own->row_done[(size_t)cy].store(1, std::memory_order_release);
  }
  return 0;
}

int Decoder::parse_job(PicJob &job, bool row_parallel)
{
  const uint8_t *data = job.rbsp.data() + job.data_off; const size_t len = job.data_len;
  const int wc = (w_ + (1 << ctbl_) - 1) >> ctbl_, hc = (h_ + (1 << ctbl_) - 1) >> ctbl_, nsub = (int)job.geom.size(), cols = job.pps.tile_cols;
  auto release_all = [&] { if (job.own) for (int r = 0; r < hc; r++) job.own->row_done[(size_t)r].store(1, std::memory_order_release); };   // never leave a later picture's parser waiting
  if ((int)job.sub_start.size() != nsub) { release_all(); return DEC_ERR_INVALID; }
  for (int r = 0; r < nsub; r++) if (job.sub_start[(size_t)r] >= len) { release_all(); return DEC_ERR_INVALID; }
  job.subs.resize((size_t)nsub);
  for (auto &r : job.subs) { r.levels.clear(); r.tus.clear(); r.rc = 0; }
  job.wpp_saved.resize((size_t)hc * cols * CTX_COUNT);
  if (!job.row_progress || job.row_progress_n < hc * cols) { job.row_progress.reset(new Progress[(size_t)hc * cols]); job.row_progress_n = hc * cols; }
  for (int r = 0; r < hc * cols; r++) job.row_progress[(size_t)r].v.store(0, std::memory_order_relaxed);
  memset(job.region, 0, (size_t)(pw_ / 32) * (ph_ / 32) * sizeof(TuRange));
  memset(job.ctu, 0, nctb() * sizeof(TuRange));
  memset(job.pred_mode.data(), PM_NONE, job.pred_mode.size());
  auto one = [&](int r) {
    if (band_nrows_ > 0 && (job.geom[(size_t)r].cy0 < band_row0_ || job.geom[(size_t)r].cy1 > band_row0_ + band_nrows_)) { job.subs[(size_t)r].rc = 0; return; }   // another decoder's rows
    size_t start = job.sub_start[(size_t)r], end = (r + 1 < nsub) ? job.sub_start[(size_t)r + 1] : len;
    int rc = end > start ? parse_substream(job, r, data + start, end - start, job.subs[(size_t)r]) : DEC_ERR_INVALID;
    job.subs[(size_t)r].rc = rc;
    if (rc < 0 && job.pps.wpp) job.row_progress[(size_t)job.geom[(size_t)r].cy0 * cols + job.geom[(size_t)r].tc].v.store(1 << 30, std::memory_order_release);   // release any waiter
    if (rc < 0) release_all();
  };
  if (row_parallel && nsub > 1) {
    if (!pool_) {
      const char *e = getenv("KVAZZUP_AMD_PARSE_THREADS");
      if (e) parse_threads_ = atoi(e) < 1 ? 1 : atoi(e);
      else if (frame_threads_ > 1 && parse_threads_ > 8) parse_threads_ = 8;      // beside the frame workers: measured best at 4K (2231 against 2085 frames/s with 16)
      pool_.reset(new OrderedPool(parse_threads_));
    }
    pool_->run(nsub, one);
  } else {
    for (int r = 0; r < nsub; r++) one(r);              // frame-parallel mode: substreams in sequence on this worker
  }
  // the substreams' transform blocks and level words follow the fixed part of the job's input block; table entries and word
  // offsets become picture-wide
  size_t ntu = 0, nlev = 0;
  for (auto &r : job.subs) { if (r.rc < 0) return r.rc; ntu += r.tus.size(); nlev += r.levels.size(); }
  const size_t tu_off = fixed_bytes(), lev_off = (tu_off + ntu * sizeof(DecTu) + 15) & ~(size_t)15;
  // (a picture with bi-predicted blocks: their second vectors ride behind the level words)
  const bool bi = (job.sh.is_b || job.sh.weighted) && job.any_bi.load(std::memory_order_relaxed) != 0;
  const size_t x_off = (lev_off + nlev * sizeof(uint32_t) + 15) & ~(size_t)15, x_bytes = bi ? job.b4x.size() * sizeof(B4L1) : 0;
  // CTBs smaller than 64: a 32x32 region's transform blocks are no run of the list any more (CTB 16: four CTBs of two CTB rows, i.e. of two substreams) -- the
  // regions get a list of INDICES into it, behind everything else in the block (DecFrame::tu_index)
  const size_t i_off = (x_off + x_bytes + 15) & ~(size_t)15, i_bytes = ctbl_ < 6 ? ntu * sizeof(uint32_t) : 0;
  if (!grow_job_input(job, i_off + i_bytes)) return DEC_ERR_GPU;
  if (bi) memcpy(job.h_in + x_off, job.b4x.data(), x_bytes);
  DecTu *tus = (DecTu *)(job.h_in + tu_off); uint32_t *lev = (uint32_t *)(job.h_in + lev_off);
  size_t t = 0, l = 0;
  for (int r = 0; r < nsub; r++) {
    SubOut &so = job.subs[(size_t)r];
    const int cy0 = job.geom[(size_t)r].cy0, cy1 = job.geom[(size_t)r].cy1, cx0 = job.geom[(size_t)r].cx0, cx1 = job.geom[(size_t)r].cx1;
    if (t) {
      for (int cy = cy0; cy < cy1; cy++) {
        for (int cx = cx0; cx < cx1; cx++) if (job.ctu[cy * wc + cx].count & 0xffffffu) job.ctu[cy * wc + cx].first += (uint32_t)t;
        if (ctbl_ == 6) for (int ry = 2 * cy; ry < 2 * cy + 2; ry++) for (int rx = 2 * cx0; rx < 2 * cx1; rx++) { TuRange &g = job.region[ry * 2 * wc + rx]; if (g.count) g.first += (uint32_t)t; }
      }
    }
    for (DecTu td : so.tus) { td.offset += (uint32_t)l; tus[t++] = td; }
    if (!so.levels.empty()) memcpy(lev + l, so.levels.data(), so.levels.size() * sizeof(uint32_t));
    l += so.levels.size();
  }
  if (ctbl_ < 6) {
    const int rw = pw_ >> 5, nreg = rw * (ph_ >> 5);
    auto region_of = [&](const DecTu &d) { const int X = d.plane ? d.x * 2 : d.x, Y = d.plane ? d.y * 2 : d.y; return (Y >> 5) * rw + (X >> 5); };
    for (int g = 0; g < nreg; g++) { job.region[g].first = 0; job.region[g].count = 0; }
    for (size_t k = 0; k < ntu; k++) job.region[region_of(tus[k])].count++;
    uint32_t at = 0;
    for (int g = 0; g < nreg; g++) { job.region[g].first = at; at += job.region[g].count; job.region[g].count = 0; }
    uint32_t *idx = (uint32_t *)(job.h_in + i_off);
    for (size_t k = 0; k < ntu; k++) { TuRange &g = job.region[region_of(tus[k])]; idx[g.first + g.count++] = (uint32_t)k; }      // (list order = decoding order inside a region)
  }
  job.ntu = ntu; job.nlev = nlev;
  if (job.lf_restricted) {
    // ---- closed slice / tile boundaries (PicJob::lf_restricted).  Every coding tree block's slice: the slices are runs of the DECODING order (tile after tile), so the
    // walk goes through the substreams' geometry; then per block which of its eight neighbours the in-loop filters may use -- not across a tile boundary when the
    // PPS says so, not across a slice boundary when the LATER of the two slices says so (its left and upper boundaries are the closed ones, 7.4.7.1).  SAO reads the
    // map; deblocking needs none: an edge on a closed boundary is no edge (8.7.2.3 filterEdgeFlag = 0), its marks come off the records here.
    const int n = wc * hc;
    std::vector<int> order((size_t)n, 0), slice((size_t)n, 0);
    {
      int ts = 0, cur = -1; size_t next = 0;
      for (const PicJob::SubGeom &g : job.geom)
        for (int cy = g.cy0; cy < g.cy1; cy++)
          for (int cx = g.cx0; cx < g.cx1; cx++) {
            const int a = cy * wc + cx;
            while (next < job.lf_slices.size() && job.lf_slices[next].first == a) { cur = (int)next; next++; }      // (a slice begins with this block)
            order[(size_t)a] = ts++; slice[(size_t)a] = cur < 0 ? 0 : cur;
          }
      if (next != job.lf_slices.size()) return DEC_ERR_INVALID;                      // (a slice that begins where no substream's walk comes by)
    }
    uint8_t *nb = job.h_in + off_nb();
    const int per = 1 << (ctbl_ - 2), b4w = pw_ / 4;
    for (int cy = 0; cy < hc; cy++)
      for (int cx = 0; cx < wc; cx++) {
        const int c = cy * wc + cx; uint8_t m = 0xff;
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            const int nx = cx + dx, ny = cy + dy;
            if ((!dx && !dy) || nx < 0 || ny < 0 || nx >= wc || ny >= hc) continue;
            const int q = ny * wc + nx, later = order[(size_t)q] > order[(size_t)c] ? q : c;
            const bool closed = (job.ctu_tile[c] != job.ctu_tile[q] && !job.pps.across_tiles && job.ctb_cut.empty()) ||      // (free slices: the byte holds the slice, there is one tile)
                                (slice[(size_t)c] != slice[(size_t)q] && !job.lf_slices[(size_t)slice[(size_t)later]].second);
            const int k = (dy + 1) * 3 + (dx + 1);
            if (closed) m &= (uint8_t)~(1u << (k > 4 ? k - 1 : k));
          }
        nb[c] = m;
        if (!(m & (1u << 3))) for (int k = 0; k < per && (cy * per + k) * 4 < h_; k++) job.b4[(size_t)(cy * per + k) * b4w + cx * per].flags &= (uint8_t)~(B4_EDGE_V | B4_TU_V);      // W
        if (!(m & (1u << 1))) for (int k = 0; k < per && (cx * per + k) * 4 < w_; k++) job.b4[(size_t)(cy * per) * b4w + cx * per + k].flags &= (uint8_t)~(B4_EDGE_H | B4_TU_H);      // N
      }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ GPU reconstruction
int Decoder::launch_gpu(PicJob &job)
{
  if (hipSetDevice(device_) != hipSuccess) return DEC_ERR_GPU;
  if (!job.conceal.empty()) {
    // buffers that stand in for lost reference pictures (conceal_ref): grey BEFORE this picture's kernels and AFTER everything older has left the GPU -- older pictures in
    // flight may still read what the buffers held.  A loss is rare: the device is simply drained (pictures handed to the submission layer are launched first).
    for (PicJob *j : gpu_q_) while (!j->launched.load(std::memory_order_acquire)) futex_wait(j->launched, 0);
    if (hipDeviceSynchronize() != hipSuccess) return DEC_ERR_GPU;
    const size_t npx = (size_t)pw_ * ph_;
    // (unconditionally: headers run ahead of launches -- by now a LATER picture may have been given the buffer, which is fine, it launches after this one and
    // overwrites it; a list inherited from a header that failed names buffers that are still stand-ins or have become this picture's own)
    for (const auto &c : job.conceal) {
      uint8_t *dst = dpb_[c.first].plane[0], *src = c.second >= 0 ? dpb_[c.second].plane[0] : nullptr;
      if (!dst) continue;
      if ((src && src != dst ? hipMemcpy(dst, src, npx * 3 / 2, hipMemcpyDeviceToDevice) : hipMemset(dst, 128, npx * 3 / 2)) != hipSuccess) return DEC_ERR_GPU;
    }
    if (hipDeviceSynchronize() != hipSuccess) return DEC_ERR_GPU;
    job.conceal.clear();
  }
  const size_t ntu = job.ntu, nlev = job.nlev;
  const size_t tu_off = fixed_bytes(), lev_off = (tu_off + ntu * sizeof(DecTu) + 15) & ~(size_t)15;
  const bool bi = (job.sh.is_b || job.sh.weighted) && job.any_bi.load(std::memory_order_relaxed) != 0;
  const size_t x_off = (lev_off + nlev * sizeof(uint32_t) + 15) & ~(size_t)15;
  const size_t i_off = (x_off + (bi ? job.b4x.size() * sizeof(B4L1) : 0) + 15) & ~(size_t)15;      // (parse_job: the regions' index list of a stream with CTBs smaller than 64)
  const size_t bytes = ctbl_ < 6 ? i_off + ntu * sizeof(uint32_t) : (bi ? x_off + job.b4x.size() * sizeof(B4L1) : lev_off + nlev * sizeof(uint32_t));
  prof_now_ = profiling_ && (launched_ % prof_every_) == 0;
  timed_job_ = &job; job.ev_used = 0;
  if (!job.done && hipEventCreateWithFlags(&job.done, hipEventDisableTiming) != hipSuccess) return DEC_ERR_GPU;
  Tick tk_api;
  // The input block goes up on its own stream into one of gpu_depth_ + 1 device buffers: the gpu_depth_ pictures launched before this
  // one may still be running (finish_oldest launches before it completes the oldest of them) and read the other buffers; the one before
  // those has been completed, so this buffer is free.
  const int ib = (int)(launched_ % (gpu_depth_ + 1));
  uint8_t *&d_in_ = this->d_in_[ib];
  if (bytes > d_in_cap_[ib]) {
    if (job.early_dst) { hipStreamSynchronize(stream_up_); job.early_dst = nullptr; }      // (the rows that went up early went into the buffer being replaced)
    hipFree(d_in_);
    d_in_cap_[ib] = bytes + bytes / 2;
    if (hipMalloc(&d_in_, d_in_cap_[ib]) != hipSuccess) { d_in_ = nullptr; d_in_cap_[ib] = 0; return DEC_ERR_GPU; }
  }
  DecFrame f; memset(&f, 0, sizeof(f));
  f.w = w_; f.h = h_; f.pw = pw_; f.ph = ph_; f.wc = (w_ + 63) / 64; f.hc = (h_ + 63) / 64;      // (wc, hc: the picture in 64x64 tiles -- what the region, deblocking and SAO grids are made of)
  f.ctb_log2 = ctbl_; f.cwc = (w_ + (1 << ctbl_) - 1) >> ctbl_; f.chc = (h_ + (1 << ctbl_) - 1) >> ctbl_;      // ... and in coding tree blocks: the intra chain's work units, the tile ids, the SAO parameters
  f.tu_index = ctbl_ < 6 ? (const uint32_t *)(d_in_ + i_off) : nullptr;
  f.b4 = (const B4Rec *)d_in_; f.b4x = bi ? (const B4L1 *)(d_in_ + x_off) : nullptr; f.region = (const TuRange *)(d_in_ + off_region()); f.ctu = (const TuRange *)(d_in_ + off_ctu());
  f.ctu_tile = d_in_ + off_tile(); f.tus = (const DecTu *)(d_in_ + tu_off); f.lev = (const uint32_t *)(d_in_ + lev_off); f.ntu = (int)ntu;
  // which chain: a picture without inter blocks, every other one of them, with the frame-threaded decoder on its own (decoder.h stream_alt_)
  DecBatcher &batcher = DecBatcher::get(device_);
  const bool batched = band_nrows_ == 0 && batch_attached_ && batcher.active() && !(job.pps.cip && job.any_inter && job.any_intra);      // (constrained intra prediction: the chain's own form, k_dec_intra_cip -- launched by this decoder itself)
  static const bool alt_off = getenv("KVAZZUP_AMD_DEC_ONE_CHAIN") != nullptr;
  bool alt = !alt_off && !batched && band_nrows_ == 0 && frame_threads_ > 1 && gpu_depth_ > 1 && job.any_intra && !job.any_inter && ((intra_seq_++) & 1);
  if (alt && !ensure_alt()) alt = false;
  const hipStream_t st = alt ? stream_alt_ : stream_;
  for (int c = 0; c < 3; c++) f.resid[c] = alt ? resid_alt_[c] : resid_[c];
  const bool sao = job.sh.sao_luma || job.sh.sao_chroma;      // the picture is then built in work_ and filtered into its buffer
  for (int c = 0; c < 3; c++) { f.rec[c] = sao ? (alt ? work_alt_[c] : work_[c]) : dpb_[job.slot].plane[c]; f.out[c] = dpb_[job.slot].plane[c]; }
  for (int k = 0; k < KVZ_DEC_MAX_REFS; k++) for (int c = 0; c < 3; c++) f.ref[k][c] = dpb_[k].plane[c];
  f.sao = sao ? (const SaoParams *)(d_in_ + off_sao()) : nullptr;
  f.ctu_nb = job.lf_restricted ? d_in_ + off_nb() : nullptr;
  f.progress = alt ? progress_alt_ : progress_; f.intra_order = intra_order_; f.err = err_;
  { const size_t nctu = (size_t)f.cwc * f.chc, S = (size_t)1 << ctbl_; uint32_t *ec = alt ? edge_col_alt_ : edge_col_; unsigned long long *er = alt ? edge_row_alt_ : edge_row_;
    f.edge_col[0] = ec; f.edge_col[1] = ec + nctu * S; f.edge_col[2] = ec + nctu * (S + S / 2);      // per CTB: S words (luma), S / 2 (Cb), S / 2 (Cr)
    f.edge_row[0] = er; f.edge_row[1] = er + nctu * (S / 4); f.edge_row[2] = er + nctu * (S / 4 + S / 8); }
  if (job.any_intra) {                                     // a generation of its own for every launch of the chain: 1 .. 2^24 - 1; at the wrap both arrays go back to "never written"
    if (++chain_gen_ >= (1u << 24)) {
      const size_t nctu = nctb() * (size_t)(2 << ctbl_) / 128;      // (in units of the 128 words a 64x64 CTB has)
      sync_main();                                         // (everything this decoder has submitted has run: nothing reads the words while they are cleared)
      // (on the consuming stream: the decoder's streams are non-blocking, nothing would order the next chain behind a clear on the null stream)
      if (hipMemsetAsync(edge_col_, 0, nctu * 128 * sizeof(uint32_t), stream_) != hipSuccess || hipMemsetAsync(edge_row_, 0, nctu * 32 * 8, stream_) != hipSuccess) return DEC_ERR_GPU;
      if (stream_alt_ && (hipMemsetAsync(edge_col_alt_, 0, nctu * 128 * sizeof(uint32_t), stream_alt_) != hipSuccess || hipMemsetAsync(edge_row_alt_, 0, nctu * 32 * 8, stream_alt_) != hipSuccess)) return DEC_ERR_GPU;
      chain_gen_ = 1;
    }
    f.chain_gen = chain_gen_;
  }
  f.cb_qp_offset = (int8_t)job.pps.cb_qp_offset; f.cr_qp_offset = (int8_t)job.pps.cr_qp_offset;
  f.beta_offset = (int8_t)(2 * job.sh.beta_offset_div2); f.tc_offset = (int8_t)(2 * job.sh.tc_offset_div2);
  f.intra_direct = job.any_inter ? 1 : 0;                 // (a picture with inter blocks: few (CTU, plane) pairs hold intra blocks)
  f.strong_intra = (uint8_t)job.sps->strong_intra; f.tiles = job.pps.tile_rows > 1 || job.pps.tile_cols > 1 || job.slice_qps.size() > 1;
  f.general = (uint8_t)(job.pps.tile_rows > 1 || job.pps.tile_cols > 1 || job.slice_qps.size() > 1 || job.sps->pcm_depth[0] != 0);
  f.cip = (uint8_t)(job.pps.cip && job.any_inter);      // (a picture without inter blocks: every neighbour is intra)
  f.tq_bypass = (uint8_t)(job.pps.tq_bypass || (job.sps->pcm_depth[0] && job.sps->pcm_no_filter));      // (units the loop filters keep out of: B4_BYPASS records)
  // scaling lists: the picture's factors (the PPS's lists when it carries any, else the SPS's) ride in the input block
  const std::vector<uint8_t> *sc = job.pps.scaling ? job.pps.scaling.get() : job.sps->scaling.get();
  if (!job.sps->scaling) sc = nullptr;                     // (scaling_list_enabled_flag = 0: a PPS's lists are not used)
  f.scaling = nullptr;
  if (sc) { memcpy(job.h_in + off_scaling(), sc->data(), KVZ_SCALING_BYTES); f.scaling = d_in_ + off_scaling(); }
  f.wt = nullptr;
  if (job.sh.weighted && bi) { memcpy(job.h_in + off_wt(), job.sh.wt, sizeof(job.sh.wt)); f.wt = (const DecWt *)(d_in_ + off_wt()); f.wt_log2[0] = job.sh.wt_log2[0]; f.wt_log2[1] = job.sh.wt_log2[1]; }
  if (band_nrows_ > 0) {
    // a band starts and ends on tile boundaries of full-width tiles; no SAO, no temporal prediction (what the split encoder writes)
    bool ok = !sao && !job.sps->tmvp && job.pps.tile_cols == 1 && frame_threads_ == 1, top = false, bottom = false;
    for (int t = 0; t <= job.pps.tile_rows; t++) { top |= job.pps.row_bd[t] == band_row0_; bottom |= job.pps.row_bd[t] == band_row0_ + band_nrows_; }
    if (!ok || !top || !bottom) return DEC_ERR_UNSUPPORTED;
    f.row0 = band_row0_; f.nrows = band_nrows_;
  }
  // Several decoders open on this device: the picture goes to the device's submission layer (batch.h), which launches the waiting pictures of
  // all of them together; its descriptor travels inside the input block.  One decoder: it launches for itself, frame by value.
  if (!batched && batch_used_) { batcher.drain(this); batch_used_ = false; }
  if (batched && stream_alt_) hipStreamSynchronize(stream_alt_);      // (another decoder has opened: from here on the submission layer launches on the shared stream; the second chain's last pictures first)       // (the other decoder has just closed: what this one still has queued there comes first)
  if (batched) memcpy(job.h_in + off_frame(), &f, sizeof(f));
  // (PicJob::early_dst: the records of every CTU row are on the device already -- queued on this stream by the row parsers -- when all rows made it)
  const size_t up_from = (!batched && job.early_dst && job.early_dst == d_in_ && job.early_rows.load(std::memory_order_acquire) == f.chc) ? off_region() : 0;
  job.early_dst = nullptr;
  if (hipMemcpyAsync(d_in_ + up_from, job.h_in + up_from, bytes - up_from, hipMemcpyHostToDevice, stream_up_) != hipSuccess) return DEC_ERR_GPU;
  if (hipEventRecord(up_done_[ib], stream_up_) != hipSuccess) return DEC_ERR_GPU;
  if (batched) {
    DecBatchItem it;
    it.f = f; it.d_f = (const DecFrame *)(d_in_ + off_frame());
    it.inter = job.any_inter; it.intra = job.any_intra; it.deblock = !job.sh.deblock_disabled; it.sao = sao;
    it.wait0 = up_done_[ib]; it.wait1 = dpb_[job.slot].last_dl; dpb_[job.slot].last_dl = nullptr;
    it.done = job.done; it.launched = &job.launched; it.owner = this; it.profile = prof_now_;
    job.launched.store(0, std::memory_order_relaxed);
    job.dl_buf = -1; job.launch_idx = launched_;
    batcher.submit(it);
    batch_used_ = true;
    t_api_ += tk_api.ms();
    launched_++;
    gpu_q_.push_back(&job);
    return 0;
  }
  if (hipStreamWaitEvent(st, up_done_[ib], 0) != hipSuccess) return DEC_ERR_GPU;
  // the buffer the picture is built in: the copy to the host of the picture last reconstructed there (queued, maybe not yet run) comes first
  if (dpb_[job.slot].last_dl) { if (hipStreamWaitEvent(st, dpb_[job.slot].last_dl, 0) != hipSuccess) return DEC_ERR_GPU; dpb_[job.slot].last_dl = nullptr; }
  // two chains: whatever this picture writes (its buffer) or reads (its reference pictures) was last touched by a picture on the OTHER stream -> after that one
  if (stream_alt_) {
    auto after = [&](DpbPic &d) { return !(d.last_use && d.last_use_alt != alt) || hipStreamWaitEvent(st, d.last_use, 0) == hipSuccess; };
    if (!after(dpb_[job.slot])) return DEC_ERR_GPU;
    for (int k = 0; k < job.nref; k++) if (!after(dpb_[job.ref_slot[k]])) return DEC_ERR_GPU;
    for (int k = 0; k < job.nref1; k++) if (!after(dpb_[job.ref_slot1[k]])) return DEC_ERR_GPU;
  }
  if (job.any_inter) timed(DK_INTER, [&] { launch_dec_inter(f, st); }, st);
  if (job.any_intra) {
    timed(job.any_inter ? DK_INTRA_P : DK_INTRA, [&] { launch_dec_intra_resid(f, st); launch_dec_intra(f, st); }, st);
  }
  if (band_nrows_ > 0) { band_f_ = f; band_din_ = d_in_; }       // deblocking follows the halo exchange (band_deblock)
  else if (!job.sh.deblock_disabled) timed(DK_DEBLOCK, [&] { launch_dec_deblock(f, st); }, st);
  if (sao) timed(DK_SAO, [&] { launch_dec_sao(f, st); }, st);
  if (hipEventRecord(job.done, st) != hipSuccess) return DEC_ERR_GPU;
  // (the picture's event now stands for the last use of its buffer and of every buffer it read; a job's event is recorded again only when its ring entry is
  // reused -- by a later picture, whose kernels are queued behind this one's on one of the two streams or wait for it through these same marks)
  dpb_[job.slot].last_use = job.done; dpb_[job.slot].last_use_alt = alt;
  for (int k = 0; k < job.nref; k++) { dpb_[job.ref_slot[k]].last_use = job.done; dpb_[job.ref_slot[k]].last_use_alt = alt; }
  for (int k = 0; k < job.nref1; k++) { dpb_[job.ref_slot1[k]].last_use = job.done; dpb_[job.ref_slot1[k]].last_use_alt = alt; }
  job.dl_buf = -1; job.launch_idx = launched_;
  t_api_ += tk_api.ms();
  launched_++;
  if (band_nrows_ > 0) gpu_job_ = &job; else gpu_q_.push_back(&job);
  return 0;
}

// ---- band mode (tile-row split): halo blocks are [luma 4 rows | Cb 2 rows | Cr 2 rows | the 4x4 records of one unit row], pw_ * 8 bytes
bool Decoder::band_export(int stage, uint8_t *d_buf)
{
  if (band_nrows_ <= 0 || !gpu_job_ || !d_buf) return false;
  PicJob &job = *gpu_job_;
  const int y = stage == 0 ? (band_row0_ + band_nrows_) * 64 - 4 : band_row0_ * 64 - 4;     // first of the four luma rows
  if (y < 0 || y + 4 > ph_) return false;
  size_t o = 0;
  if (hipMemcpyAsync(d_buf + o, dpb_[job.slot].plane[0] + (size_t)y * pw_, (size_t)pw_ * 4, hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false;
  o += (size_t)pw_ * 4;
  for (int c = 1; c < 3; c++) { if (hipMemcpyAsync(d_buf + o, dpb_[job.slot].plane[c] + (size_t)(y / 2) * (pw_ / 2), (size_t)pw_, hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false; o += (size_t)pw_; }
  if (stage == 0 && hipMemcpyAsync(d_buf + o, band_din_ + (size_t)(y / 4) * (pw_ / 4) * sizeof(B4Rec), (size_t)(pw_ / 4) * sizeof(B4Rec), hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false;
  return hipStreamSynchronize(stream_) == hipSuccess;
}
bool Decoder::band_import(int stage, const uint8_t *d_buf)
{
  if (band_nrows_ <= 0 || !gpu_job_ || !d_buf) return false;
  PicJob &job = *gpu_job_;
  const int y = stage == 0 ? band_row0_ * 64 - 4 : (band_row0_ + band_nrows_) * 64 - 4;
  if (y < 0 || y + 4 > ph_) return false;
  size_t o = 0;
  if (hipMemcpyAsync(dpb_[job.slot].plane[0] + (size_t)y * pw_, d_buf + o, (size_t)pw_ * 4, hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false;
  o += (size_t)pw_ * 4;
  for (int c = 1; c < 3; c++) { if (hipMemcpyAsync(dpb_[job.slot].plane[c] + (size_t)(y / 2) * (pw_ / 2), d_buf + o, (size_t)pw_, hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false; o += (size_t)pw_; }
  if (stage == 0 && hipMemcpyAsync(band_din_ + (size_t)(y / 4) * (pw_ / 4) * sizeof(B4Rec), d_buf + o, (size_t)(pw_ / 4) * sizeof(B4Rec), hipMemcpyDeviceToDevice, stream_) != hipSuccess) return false;
  return hipStreamSynchronize(stream_) == hipSuccess;
}
bool Decoder::band_deblock()
{
  if (band_nrows_ <= 0 || !gpu_job_) return false;
  if (!gpu_job_->sh.deblock_disabled) launch_dec_deblock(band_f_, stream_);
  return hipEventRecord(gpu_job_->done, stream_) == hipSuccess;
}
int Decoder::band_finish()
{
  if (band_nrows_ <= 0 || !gpu_job_) return 0;
  PicJob *j = gpu_job_; gpu_job_ = nullptr;
  if (hipEventRecord(j->done, stream_) != hipSuccess) return DEC_ERR_GPU;
  return complete_gpu(*j);
}

bool Decoder::get_picture(DecodedPicture *out)
{
  if (!pic_ready_) return false;
  *out = out_;
  return true;
}

bool Decoder::debug_copy(const char *what, void *dst, size_t bytes)
{
  if (!w_) return false;
  const size_t npx = (size_t)pw_ * ph_;
  std::string w(what);
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    if (w == std::string("rec") + char('0' + c)) { if (bytes > n) return false; return hipMemcpy(dst, dpb_[out_slot_].plane[c], bytes, hipMemcpyDeviceToHost) == hipSuccess; }
  }
  return false;
}

}  // namespace kvzx
