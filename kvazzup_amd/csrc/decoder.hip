// kvazzup_amd/csrc/decoder.hip -- see decoder.h
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "decoder.h"

namespace kvzx {

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "kvazzup_amd: %s failed: %s\n", #expr, hipGetErrorString(e_)); return false; } } while (0)
enum { DEC_ERR_INVALID = -1, DEC_ERR_UNSUPPORTED = -2, DEC_ERR_GPU = -3 };

namespace {

// ------------------------------------------------------------------------------------------ bits
struct BitReader {
  const uint8_t *p; size_t n, pos = 0; bool err = false;
  BitReader(const uint8_t *b, size_t len) : p(b), n(len) {}
  uint32_t bit() { if (pos >= n * 8) { err = true; pos++; return 0; } uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
  uint32_t get(int k) { uint32_t v = 0; for (int i = 0; i < k; i++) v = (v << 1) | bit(); return v; }
  uint32_t ue() { int z = 0; while (!bit()) { if (++z > 32 || err) { err = true; return 0; } } return z ? ((1u << z) - 1) + get(z) : 0; }
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

// ------------------------------------------------------------------------------------------ CABAC decoding (H.265 9.3.4.3)
// Arithmetic decoder with the offset kept scaled in a 64-bit register: value = offset << bits | next
// `bits` stream bits, so a renormalisation by n is just bits -= n and the stream is touched 32 bits at
// a time.  Context variable = pStateIdx << 1 | valMps with precomputed transitions.
struct StateTabs { uint8_t next_mps[128], next_lps[128]; };
const StateTabs &state_tabs()               // (function-local statics: initialised once, thread-safe -- parse workers race to the first call)
{
  static const StateTabs t = [] {
    StateTabs t;
    for (int s = 0; s < 128; s++) {
      int st = s >> 1, mps = s & 1;
      t.next_mps[s] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
      t.next_lps[s] = (uint8_t)((kNextLps[st] << 1) | (st == 0 ? mps ^ 1 : mps));
    }
    return t;
  }();
  return t;
}
struct CabacDec {
  const uint8_t *buf = nullptr, *p = nullptr; size_t len = 0;   // buf is padded with >= 16 readable bytes
  uint64_t value = 0; int bits = 0;
  uint32_t range = 510;
  const StateTabs *st = nullptr;
  uint8_t ctx[CTX_COUNT];
  inline void refill()
  {
    if (bits < 16) {
      uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
      value = (value << 32) | w; p += 4; bits += 32;
    }
  }
  bool overrun() const { return (size_t)(p - buf) > len + 12; }
  void start(const uint8_t *b, size_t l)
  {
    buf = b; p = b; len = l; st = &state_tabs(); range = 510;
    value = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    p += 4; bits = 32 - 9;
    refill();
  }
  inline int bin(int ci)
  {
    // (both outcomes are computed and selected: the bin values of sig / greater1 flags are close to coin flips for a branch predictor)
    const uint32_t s = ctx[ci];
    const uint32_t lps = kRangeLps[s >> 1][(range >> 6) & 3];
    const uint32_t rmps = range - lps;
    const uint64_t scaled = (uint64_t)rmps << bits;
    const bool isl = value >= scaled;
    value -= isl ? scaled : 0;
    const uint32_t r = isl ? lps : rmps;
    ctx[ci] = isl ? st->next_lps[s] : st->next_mps[s];
    const int n = __builtin_clz(r) - 23;                  // renormalisation: r in [1, 510] -> [256, 510]
    range = r << n; bits -= n;
    refill();
    return (int)((s & 1u) ^ (uint32_t)isl);
  }
  inline int bypass()
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
  inline uint32_t bypass_bits(int n)
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
        v = (v << m) | (uint32_t)q;
        refill();
      }
      n -= m;
    }
    return v;
  }
  int terminate()
  {
    range -= 2;
    if (value >= ((uint64_t)range << bits)) return 1;
    if (range < 256) { range <<= 1; bits--; }
    refill();
    return 0;
  }
  // bytes from the start of the substream up to and including the byte holding the last consumed bit
  size_t bytes_consumed() const { size_t consumed_bits = (size_t)(p - buf) * 8 - (size_t)bits; return (consumed_bits + 7) >> 3; }
};

std::atomic<long> g_yields{0};
std::atomic<uint64_t> g_tc[6];
#define TSC() __builtin_ia32_rdtsc()
struct Tick { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
const CoreTabs *host_tabs()
{
  static const CoreTabs t = [] { CoreTabs t; for (int i = 0; i < 64; i++) core_tabs_fill_entry(t, i); return t; }();
  return &t;
}

// scan position -> (x, y) for the three scans and block sizes 1..8 (H.265 6.5.3-6.5.5)
struct ScanTabs { uint8_t x[3][4][64], y[3][4][64], inv[3][4][64], sigk[3][5][16]; };   // sigk[scan][prev_csbf, 4 = 4x4 block][scan position k] = context pattern     // inv[scan][log2 of the grid][y << log2 | x] = scan position
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

// residual_coding() (7.3.8.11) without transform skip / sign hiding; writes n*n levels row-major
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

bool parse_residual(CabacDec &c, int log2, int cidx, int scan_idx, std::vector<uint32_t> &out)
{
  const CoreTabs *t = host_tabs();
  const ScanTabs &S = scan_tabs();
  const int n = 1 << log2, sbl = log2 - 2, nsb = 1 << sbl;
  const uint8_t *SX = S.x[scan_idx][sbl], *SY = S.y[scan_idx][sbl], *PX = S.x[scan_idx][2], *PY = S.y[scan_idx][2];
  uint8_t csbf[8][8]; memset(csbf, 0, sizeof(csbf));
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
    for (int k = (i == last_sb) ? last_pos - 1 : 15; k >= 0; k--) {
      if (k > 0 || !infer_dc) {
        const int sc = (k == 0 && i == 0 && log2 != 2) ? 0 : pk[k] + sig_off;      // (the DC coefficient of the block has its own context)
        const int b = c.bin(sig_base + sc);
        sig |= (uint32_t)b << k; infer_dc &= b ^ 1;
      } else sig |= 1u;            // k == 0 with every other flag of a coded sub-block zero: inferred
    }
    if (!sig) continue;
    int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
    if (c1 == 0) ctx_set++;
    c1 = 1;
    int pos[16], lev[16], nsig = 0, g1idx = -1;
    for (uint32_t m = sig; m;) { const int k = 31 - __builtin_clz(m); pos[nsig++] = k; m &= ~(1u << k); }     // highest scan position first
    for (int j = 0; j < nsig; j++) lev[j] = 1;
    for (int j = 0; j < nsig && j < 8; j++) {
      int g1 = c.bin(CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1);
      if (g1) { lev[j] = 2; c1 = 0; if (g1idx < 0) g1idx = j; }
      else if (c1 > 0 && c1 < 3) c1++;
    }
    if (g1idx >= 0 && c.bin(CTX_GT2 + (cidx ? 4 : 0) + ctx_set)) lev[g1idx] = 3;
    uint32_t signs = c.bypass_bits(nsig);
    int rice = 0;
    for (int j = 0; j < nsig; j++) {
      int base = (j < 8) ? ((j == g1idx) ? 3 : 2) : 1;
      if (lev[j] == base) {
        int prefix = 0;
        while (prefix < 32 && c.bypass()) prefix++;
        if (prefix >= 32) return false;
        int rem = prefix <= 3 ? (prefix << rice) + (int)c.bypass_bits(rice)
                              : (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)c.bypass_bits(prefix - 3 + rice);
        lev[j] = base + rem;
        if (lev[j] > 3 * (1 << rice)) rice = imin(rice + 1, 4);
      }
      int v = ((signs >> (nsig - 1 - j)) & 1) ? -lev[j] : lev[j];
      const int xp = PX[pos[j]], yp = PY[pos[j]];
      out.push_back((uint32_t)((((ys << 2) + yp) * n + (xs << 2) + xp) << 16) | ((uint32_t)clip3(-32768, 32767, v) & 0xffffu));
    }
  }
  return !c.overrun();
}

}  // namespace

// ------------------------------------------------------------------------------------------ lifecycle
// Pictures still being parsed by frame workers are waited for and dropped (close / resolution change).
void Decoder::drop_pending()
{
  if (gpu_job_) { hipStreamSynchronize(stream_); gpu_job_ = nullptr; ev_used_ = 0; }
  for (; job_tail_ != job_head_; job_tail_++) {
    PicJob &job = jobs_[(size_t)(job_tail_ % (frame_threads_ + 1))];
    while (job.state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
    job.state.store(0, std::memory_order_relaxed);
  }
}

Decoder::~Decoder()
{
  if (getenv("KVAZZUP_AMD_TRACE")) fprintf(stderr, "kvazzup_amd parse Mcycles: split %.1f  skip/pred %.1f  intra/merge/amvp %.1f  residual %.1f  records %.1f  ctu-end %.1f\n", g_tc[0] * 1e-6, g_tc[1] * 1e-6, g_tc[2] * 1e-6, g_tc[3] * 1e-6, g_tc[4] * 1e-6, g_tc[5] * 1e-6);
  if (getenv("KVAZZUP_AMD_TRACE")) fprintf(stderr, "kvazzup_amd decoder thread ms: nal %.1f  wait_parse %.1f  stage %.1f  gpu_api %.1f  gpu_sync %.1f  longest parse %.2f  (pictures %ld)\n", t_nal_, t_wait_, t_stage_, t_api_, t_sync_, t_parse_max_, job_tail_);
  drop_pending();
  workers_.reset();
  if (stream_) hipStreamSynchronize(stream_);
  for (auto &e : ev_pool_) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  free_buffers();
  if (h_err_) hipHostFree(h_err_);
  if (err_) hipFree(err_);
  if (stream_) hipStreamDestroy(stream_);
}

bool Decoder::start(std::string *error)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_) {
    if (error) *error = "no usable HIP device (this library has no CPU fallback)";
    return false;
  }
  HIP_TRY(hipSetDevice(device_));
  spin_wait_ = getenv("KVAZZUP_AMD_SPIN") != nullptr;
  {
    const char *prio = getenv("KVAZZUP_AMD_PRIO"); int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char lv = (prio && strlen(prio) >= 4) ? prio[3] : 'n';
    if (lv == 'h') HIP_TRY(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
    else if (lv == 'l') HIP_TRY(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, lo));
    else HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  HIP_TRY(hipMalloc(&err_, sizeof(uint32_t)));
  HIP_TRY(hipMemset(err_, 0, sizeof(uint32_t)));
  HIP_TRY(hipHostMalloc(&h_err_, sizeof(uint32_t), hipHostMallocDefault));
  started_ = true;
  return true;
}

void Decoder::free_buffers()
{
  for (auto &j : jobs_) { if (j.h_in) hipHostFree(j.h_in); j.h_in = nullptr; j.h_in_cap = 0; }
  if (h_out_) hipHostFree(h_out_);
  hipFree(d_in_); hipFree(d_mvd_); hipFree(sync_);
  for (int c = 0; c < 3; c++) { for (int b = 0; b < 3; b++) { hipFree(rec_[b][c]); rec_[b][c] = nullptr; } hipFree(work_[c]); work_[c] = nullptr; hipFree(coef_[c]); coef_[c] = nullptr; }
  h_out_ = nullptr; d_in_ = nullptr; d_in_cap_ = 0; d_mvd_ = nullptr; sync_ = nullptr;
  cw_ = ch_ = 0;
}

// EncFrame view of the CU records and motion vectors at the start of an input block (host or device)
void Decoder::bind_views(EncFrame &f, uint8_t *base)
{
  const size_t nb8 = (size_t)cw_ * ch_ / 64;
  const int wpp = f.wpp, is_intra = f.is_intra, qp = f.qp, tile_rows = f.tile_rows > 0 ? f.tile_rows : 1;
  EncFrame keep = f;
  memset(&f, 0, sizeof(f));
  f.cw = cw_; f.ch = ch_; f.b8w = cw_ / 8; f.b8h = ch_ / 8; f.wpp = wpp; f.is_intra = is_intra; f.qp = qp;
  f.tile_rows = tile_rows; f.chp = pack_height(ch_, tile_rows);
  f.cu_log2 = base; f.cu_intra = base + nb8; f.cu_flags = base + 2 * nb8; f.cu_merge_idx = base + 3 * nb8;
  f.cu_mvp_idx = base + 4 * nb8; f.cu_intra_mode = base + 5 * nb8; f.cu_cbf = base + 6 * nb8; f.cu_mv = (int16_t *)(base + 7 * nb8);
  f.cu_mvd = keep.cu_mvd; f.sync = keep.sync; f.err = keep.err;
  for (int c = 0; c < 3; c++) { f.coef[c] = keep.coef[c]; f.rec[c] = keep.rec[c]; f.ref[c] = keep.ref[c]; }
  // per-CTU QP arrays follow the motion vectors; the frame's pointers are switched on per picture (PPS cu_qp_delta_enabled_flag)
}

// pointers of the per-CTU QP arrays inside an input block
static inline void bind_qp_arrays(EncFrame &f, uint8_t *base, int cw, int ch, bool on)
{
  const size_t nb8 = (size_t)cw * ch / 64, nctu = (size_t)(cw / 64) * (ch / 64);
  int8_t *q = (int8_t *)(base + 11 * nb8);
  f.ctu_qy = on ? q : nullptr; f.ctu_qt = on ? q : nullptr; f.ctu_delta = on ? q + nctu : nullptr; f.ctu_first = on ? (uint8_t *)(q + 2 * nctu) : nullptr;
}

// pinned input block of a job: at least `bytes`; the CU / motion part written so far is kept.  Called from the
// thread that owns the job (decoder thread at allocation, the job's parse worker later).
bool Decoder::grow_job_input(PicJob &job, size_t bytes)
{
  if (bytes <= job.h_in_cap) return true;
  if (hipSetDevice(device_) != hipSuccess) return false;
  const size_t cap = bytes + bytes / 2;
  uint8_t *p = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return false;
  if (job.h_in) { memcpy(p, job.h_in, fixed_bytes() < job.h_in_cap ? fixed_bytes() : job.h_in_cap); hipHostFree(job.h_in); }
  job.h_in = p; job.h_in_cap = cap;
  bind_views(job.hf, p);
  return true;
}

bool Decoder::ensure_buffers(int cw, int ch)
{
  if (cw == cw_ && ch == ch_) return true;
  drop_pending();                                          // (resolution change: pictures not yet output are dropped)
  hipStreamSynchronize(stream_);
  free_buffers();
  const size_t npx = (size_t)cw * ch, nb8 = npx / 64;
  if (jobs_.empty()) jobs_ = std::vector<PicJob>((size_t)frame_threads_ + 1);   // parse ring + the picture in flight on the GPU
  cw_ = cw; ch_ = ch;
  for (auto &j : jobs_) {
    if (!grow_job_input(j, fixed_bytes() + (1 << 16))) return false;
    memset(j.h_in, 0, fixed_bytes());
  }
  HIP_TRY(hipHostMalloc(&h_out_, npx * 3 / 2, hipHostMallocDefault));
  h_out_cap_ = npx * 3 / 2;
  d_in_cap_ = fixed_bytes() + (1 << 20);
  HIP_TRY(hipMalloc(&d_in_, d_in_cap_));
  HIP_TRY(hipMalloc(&d_mvd_, nb8 * 2 * sizeof(int16_t)));
  HIP_TRY(hipMalloc(&sync_, sizeof(uint32_t) * 3 * (size_t)(ch / 64)));
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    for (int b = 0; b < 3; b++) { HIP_TRY(hipMalloc(&rec_[b][c], n)); HIP_TRY(hipMemset(rec_[b][c], 128, n)); }
    HIP_TRY(hipMalloc(&work_[c], n));
    HIP_TRY(hipMalloc(&coef_[c], n * sizeof(int16_t)));
  }
  for (auto &j : jobs_) bind_views(j.hf, j.h_in);
  bind_views(f_, d_in_);
  f_.cu_mvd = d_mvd_;
  for (int c = 0; c < 3; c++) f_.coef[c] = coef_[c];
  f_.sync = sync_; f_.err = err_;
  have_ref_ = false;
  return true;
}

template <class F> void Decoder::timed(int id, F &&launch)
{
  if (!prof_now_) { launch(); return; }
  if (ev_used_ == ev_pool_.size()) { EvPair p; hipEventCreate(&p.a); hipEventCreate(&p.b); p.id = id; ev_pool_.push_back(p); }
  EvPair &p = ev_pool_[ev_used_++]; p.id = id;
  hipEventRecord(p.a, stream_); launch(); hipEventRecord(p.b, stream_);
}
void Decoder::get_kernel_times(double *ms, uint64_t *launches, bool reset)
{
  for (int i = 0; i < DK_COUNT; i++) { if (ms) ms[i] = k_ms_[i]; if (launches) launches[i] = k_n_[i]; }
  if (reset) for (int i = 0; i < DK_COUNT; i++) { k_ms_[i] = 0; k_n_[i] = 0; }
}

// ------------------------------------------------------------------------------------------ NAL units
int Decoder::decode_nal(const uint8_t *data, size_t len, int64_t pts)
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
  rbsp_.assign(len + 32, 0);
  epb_.clear();
  size_t n = 0; int zeros = 0;
  for (size_t k = 2; k < len; k++) {
    if (zeros >= 2 && data[k] == 3) { zeros = 0; epb_.push_back(n); continue; }
    rbsp_[n++] = data[k]; zeros = data[k] == 0 ? zeros + 1 : 0;
  }
  BitReader r(rbsp_.data(), n);
  if (nal_type == 32) {                                          // VPS: only the timing information is used
    r.get(4); r.get(2); r.get(6); int msl = r.get(3); r.get(1); r.get(16);
    if (!skip_ptl(r, msl)) return last_error_ = DEC_ERR_INVALID;
    int oi = r.get(1);
    for (int k = oi ? 0 : msl; k <= msl; k++) { r.ue(); r.ue(); r.ue(); }
    int max_layer_id = r.get(6); int nls = r.ue() + 1;
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
    if (r.get(1)) { s.crop_l = 2 * r.ue(); s.crop_r = 2 * r.ue(); s.crop_t = 2 * r.ue(); s.crop_b = 2 * r.ue(); }
    if (r.ue() != 0 || r.ue() != 0) return last_error_ = DEC_ERR_UNSUPPORTED;   // 8 bit only
    s.log2_max_poc_lsb = r.ue() + 4;
    int oi = r.get(1);
    for (int k = oi ? 0 : msl; k <= msl; k++) { r.ue(); r.ue(); r.ue(); }
    int log2_min_cb = r.ue() + 3, diff_cb = r.ue(), log2_min_tb = r.ue() + 2, diff_tb = r.ue(), dinter = r.ue(), dintra = r.ue();
    int scaling = r.get(1); int amp = r.get(1), sao = r.get(1), pcm = r.get(1);
    s.sao = sao;
    if (scaling || amp || pcm || log2_min_cb != 3 || diff_cb != 3 || log2_min_tb != 2 || diff_tb != 3 || dinter != 0 || dintra != 0)
      return last_error_ = DEC_ERR_UNSUPPORTED;
    s.num_st_rps = r.ue();
    if (s.num_st_rps > 64) return last_error_ = DEC_ERR_INVALID;
    for (int k = 0; k < s.num_st_rps; k++) {
      if (k != 0 && r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;            // inter RPS prediction
      int nneg = r.ue(), npos = r.ue();
      if (nneg != 1 || npos != 0) return last_error_ = DEC_ERR_UNSUPPORTED;        // exactly one (previous) reference
      s.rps_neg[k] = -(int)(r.ue() + 1); s.rps_used[k] = r.get(1);
    }
    if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;      // long-term references
    if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;      // temporal MVP
    s.strong_intra = r.get(1);
    if (!s.strong_intra) return last_error_ = DEC_ERR_UNSUPPORTED;   // kernels assume strong_intra_smoothing_enabled_flag = 1
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
    if ((s.width & 63) || (s.height & 63) || s.width < 128 || s.height < 64) return last_error_ = DEC_ERR_UNSUPPORTED;
    s.valid = true; sps_[id] = s;
    return 0;
  }
  if (nal_type == 34) {                                          // PPS (7.3.2.3)
    DecPps p;
    int id = r.ue(), sid = r.ue();
    if (id > 63 || sid != 0) return last_error_ = (id > 63 ? DEC_ERR_INVALID : DEC_ERR_UNSUPPORTED);
    int dep = r.get(1), outflag = r.get(1), extra = r.get(3), signhide = r.get(1);
    p.cabac_init_present = r.get(1);
    int l0 = r.ue(), l1 = r.ue(); (void)l1;
    p.init_qp = 26 + r.se();
    int cip = r.get(1), tskip = r.get(1), cuqpd = r.get(1);
    if (cuqpd) { if (r.ue() != 0) return last_error_ = DEC_ERR_UNSUPPORTED; p.qp_in_cu = 1; }   // quantisation group = CTU only
    int cbo = r.se(), cro = r.se(), sco = r.get(1), wp = r.get(1), wbp = r.get(1), tqb = r.get(1), tiles = r.get(1);
    p.wpp = r.get(1);
    if (dep || outflag || extra || signhide || p.cabac_init_present || l0 != 0 || cip || tskip || cbo || cro || sco || wp || wbp || tqb)
      return last_error_ = DEC_ERR_UNSUPPORTED;
    if (tiles) {                                                 // supported: one column, uniform spacing, loop filter across tiles on
      const int cols = r.ue() + 1, rows = r.ue() + 1, uniform = r.get(1);
      if (cols != 1 || !uniform || rows > 1024) return last_error_ = DEC_ERR_UNSUPPORTED;
      if (!r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;   // loop_filter_across_tiles_enabled_flag
      p.tile_rows = rows;
    }
    p.loop_filter_across_slices = r.get(1);
    p.deblock_control = r.get(1);
    if (p.deblock_control) {
      if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;    // deblocking_filter_override_enabled_flag
      p.deblock_disabled = r.get(1);
      if (!p.deblock_disabled && (r.se() != 0 || r.se() != 0)) return last_error_ = DEC_ERR_UNSUPPORTED;
    }
    if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;      // scaling list data
    if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;      // lists_modification_present_flag
    if (r.ue() != 0) return last_error_ = DEC_ERR_UNSUPPORTED;   // log2_parallel_merge_level_minus2
    if (r.get(1)) return last_error_ = DEC_ERR_UNSUPPORTED;      // slice header extension
    if (r.err) return last_error_ = DEC_ERR_INVALID;
    p.valid = true; pps_[id] = p;
    return 0;
  }
  if (nal_type == 36 || nal_type == 37) { int rc = finish_oldest(); if (rc < 0) last_error_ = rc; return rc; }   // EOS / EOB: drain one delayed picture
  if (nal_type > 31) return 0;                                    // AUD / SEI / ...
  if (!(nal_type == 0 || nal_type == 1 || nal_type == 19 || nal_type == 20)) return last_error_ = DEC_ERR_UNSUPPORTED;
  int rc = decode_slice(rbsp_.data(), n, nal_type, pts);
  if (rc < 0) last_error_ = rc;
  return rc;
}

int Decoder::decode_slice(const uint8_t *rbsp, size_t len, int nal_type, int64_t pts)
{
  BitReader r(rbsp, len);
  const bool idr = nal_type == 19 || nal_type == 20;
  if (!r.get(1)) return DEC_ERR_UNSUPPORTED;                     // one slice per picture
  if (idr) r.get(1);
  int pps_id = r.ue();
  if (pps_id > 63 || !pps_[pps_id].valid || !sps_[0].valid) return DEC_ERR_INVALID;
  const DecPps &p = pps_[pps_id]; const DecSps &s = sps_[0];
  int slice_type = r.ue();
  if (slice_type != 1 && slice_type != 2) return DEC_ERR_UNSUPPORTED;
  const bool is_intra = slice_type == 2;
  if (!is_intra && idr) return DEC_ERR_INVALID;
  int poc = 0;
  if (!idr) {
    int lsb = r.get(s.log2_max_poc_lsb), max_lsb = 1 << s.log2_max_poc_lsb;
    int prev_lsb = prev_poc_ & (max_lsb - 1), prev_msb = prev_poc_ - prev_lsb, msb;
    if (lsb < prev_lsb && prev_lsb - lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (lsb > prev_lsb && lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    else msb = prev_msb;
    poc = msb + lsb;
    int neg = 0, used = 0;
    if (r.get(1)) {
      int idx = 0, bits = 0; while ((1 << bits) < s.num_st_rps) bits++;
      if (s.num_st_rps == 0) return DEC_ERR_INVALID;
      if (bits) idx = r.get(bits);
      if (idx >= s.num_st_rps) return DEC_ERR_INVALID;
      neg = s.rps_neg[idx]; used = s.rps_used[idx];
    } else {
      if (s.num_st_rps != 0 && r.get(1)) return DEC_ERR_UNSUPPORTED;
      if (r.ue() != 1 || r.ue() != 0) return DEC_ERR_UNSUPPORTED;
      neg = -(int)(r.ue() + 1); used = r.get(1);
    }
    if (!is_intra && (neg != -1 || !used || !have_ref_ || poc - 1 != prev_poc_)) return DEC_ERR_UNSUPPORTED;   // reference = previous picture
  }
  int sao_luma = 0, sao_chroma = 0;
  if (s.sao) { sao_luma = r.get(1); sao_chroma = r.get(1); }
  int max_merge = 5;
  if (!is_intra) {
    if (r.get(1)) { if (r.ue() != 0) return DEC_ERR_UNSUPPORTED; }        // num_ref_idx_active override: still one reference
    max_merge = 5 - (int)r.ue();
    if (max_merge < 1 || max_merge > 5) return DEC_ERR_INVALID;
  }
  const int slice_qp = p.init_qp + r.se();
  if (slice_qp < 0 || slice_qp > 51) return DEC_ERR_INVALID;
  const bool deblock = !p.deblock_disabled;
  if (p.loop_filter_across_slices && (deblock || sao_luma || sao_chroma)) r.get(1);
  std::vector<uint32_t> entry;
  if (p.wpp || p.tile_rows > 1) {
    int nep = r.ue();
    if (nep < 0 || nep > 1024) return DEC_ERR_INVALID;
    if (nep > 0) { int bits = r.ue() + 1; if (bits > 32) return DEC_ERR_INVALID; for (int k = 0; k < nep; k++) entry.push_back(r.get(bits) + 1); }
    if (nep != (p.wpp ? s.height / 64 : p.tile_rows) - 1) return DEC_ERR_UNSUPPORTED;    // one substream per CTU row (WPP) or per tile
  }
  if (!r.get(1)) return DEC_ERR_INVALID;                         // byte_alignment()
  while (r.pos & 7) r.get(1);
  if (r.err) return DEC_ERR_INVALID;
  // Substream starts inside the unescaped slice data.  entry_point offsets count bytes of the NAL
  // unit payload INCLUDING emulation prevention bytes (7.4.7.1); epb_[] holds, for every removed
  // byte, how many unescaped payload bytes preceded it.
  {
    const size_t hdr = r.pos >> 3;
    sub_start_.assign(1, 0);
    size_t esc = hdr;                                            // escaped offset of the slice data in the payload
    for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] < hdr) esc++;
    for (uint32_t e : entry) {
      esc += e;
      size_t removed = 0;
      for (size_t k = 0; k < epb_.size(); k++) if (epb_[k] + k < esc) removed++;     // epb k sits at escaped offset epb_[k] + k
      sub_start_.push_back(esc - removed - hdr);
    }
  }
  if (!ensure_buffers(s.width, s.height)) return DEC_ERR_GPU;
  active_sps_ = &s;
  // ---- hand the picture to a parse job.  With frame threads (libOpenHevcInit thread_type FRAME / FRAMESLICE)
  // up to `frame_threads_` pictures are parsed concurrently on worker threads -- CABAC parsing of a picture
  // needs nothing from other pictures -- and the output is delayed accordingly, like OpenHEVC's frame threading.
  PicJob &job = jobs_[(size_t)(job_head_ % (frame_threads_ + 1))];
  job.rbsp.assign(rbsp, rbsp + len + 32);                        // keeps the zero padding the CABAC reader relies on
  job.data_off = r.pos >> 3; job.data_len = len - (r.pos >> 3);
  job.sub_start = sub_start_;
  job.slice_qp = slice_qp; job.is_intra = is_intra; job.max_merge = max_merge; job.deblock = deblock; job.poc = poc; job.pts = pts;
  job.crop[0] = s.crop_l; job.crop[1] = s.crop_r; job.crop[2] = s.crop_t; job.crop[3] = s.crop_b;
  job.fps_num = s.fps_num ? s.fps_num : vps_fps_num_; job.fps_den = s.fps_num ? s.fps_den : vps_fps_den_;
  job.hf.is_intra = is_intra; job.hf.wpp = p.wpp; job.hf.qp = slice_qp;
  if (p.tile_rows > s.height / 64) return DEC_ERR_INVALID;
  job.tile_rows = p.tile_rows; job.hf.tile_rows = p.tile_rows; job.hf.chp = pack_height(ch_, p.tile_rows);
  job.qp_in_cu = p.qp_in_cu;
  job.sao_luma = sao_luma; job.sao_chroma = sao_chroma;
  job.hf.sao = (sao_luma || sao_chroma) ? (SaoParams *)(job.h_in + sao_offset()) : nullptr;
  bind_qp_arrays(job.hf, job.h_in, cw_, ch_, true);               // the parser always fills them (one QP everywhere without cu_qp_delta)
  job.rc = 0;
  prev_poc_ = poc; have_ref_ = true;                             // header checks of the next picture run before this one is reconstructed
  job_head_++;
  if (frame_threads_ == 1) {
    auto t0 = std::chrono::steady_clock::now();
    job.rc = parse_job(job, true);
    if (profiling_) { k_ms_[DK_HOST_PARSE] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); k_n_[DK_HOST_PARSE]++; }
    job.state.store(2, std::memory_order_release);
  } else {
    job.state.store(1, std::memory_order_release);
    if (!workers_) workers_.reset(new FrameWorkers(frame_threads_));
    PicJob *jp = &job;
    workers_->submit([this, jp] {
      auto t0 = std::chrono::steady_clock::now();
      jp->rc = parse_job(*jp, false);
      jp->parse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      jp->state.store(2, std::memory_order_release);
    });
  }
  if (job_head_ - job_tail_ < frame_threads_) return 0;          // pipeline still filling: no output for this NAL
  return finish_oldest();
}

// Waits for the oldest submitted picture to be parsed, reconstructs it on the GPU and makes it the output.
// Output stage.  Synchronous mode (frame_threads_ == 1): the picture just parsed is reconstructed and output.
// Frame-threaded mode: first the picture launched by the previous call is completed and becomes the output,
// then the oldest parsed picture is launched -- its kernels run while this thread goes on parsing headers.
int Decoder::finish_oldest()
{
  int produced = 0;
  if (gpu_job_) { int rc = complete_gpu(); if (rc < 0) return rc; produced = 1; }
  if (job_head_ != job_tail_) {
    PicJob &job = jobs_[(size_t)(job_tail_ % (frame_threads_ + 1))];
    job_tail_++;
    { Tick tk; while (job.state.load(std::memory_order_acquire) != 2) std::this_thread::yield(); t_wait_ += tk.ms(); }
    job.state.store(0, std::memory_order_relaxed);
    if (frame_threads_ > 1 && profiling_) { k_ms_[DK_HOST_PARSE] += job.parse_ms; k_n_[DK_HOST_PARSE]++; }
    if (job.parse_ms > t_parse_max_) t_parse_max_ = job.parse_ms;
    if (job.rc < 0) return job.rc;
    int rc = launch_gpu(job);
    if (rc < 0) return rc;
    if (frame_threads_ == 1) { rc = complete_gpu(); if (rc < 0) return rc; produced = 1; }
  }
  return produced;
}

int Decoder::complete_gpu()
{
  PicJob &job = *gpu_job_;
  gpu_job_ = nullptr;
  {
    Tick tk;
    if (frame_threads_ > 1 && !spin_wait_) {   // the output lags anyway: nap between queries instead of polling (see nap_until)
      if (!nap_until([&] { hipError_t r = hipStreamQuery(stream_); return r == hipSuccess ? 1 : (r == hipErrorNotReady ? 0 : -1); })) return DEC_ERR_GPU;
    } else if (hipStreamSynchronize(stream_) != hipSuccess) return DEC_ERR_GPU;
    t_sync_ += tk.ms();
  }
  if (*h_err_) { fprintf(stderr, "kvazzup_amd: decoder device error flags 0x%x\n", *h_err_); return DEC_ERR_GPU; }
  if (ev_used_) {
    for (size_t i = 0; i < ev_used_; i++) { float ms = 0; hipEventElapsedTime(&ms, ev_pool_[i].a, ev_pool_[i].b); k_ms_[ev_pool_[i].id] += ms; k_n_[ev_pool_[i].id]++; }
    ev_used_ = 0;
  }
  poc_ = job.poc;
  out_ = DecodedPicture();
  out_.coded_w = cw_; out_.coded_h = ch_;
  out_.width = cw_ - job.crop[0] - job.crop[1]; out_.height = ch_ - job.crop[2] - job.crop[3];
  out_.poc = job.poc; out_.pts = job.pts; out_.is_intra = job.is_intra;
  out_.fps_num = job.fps_num; out_.fps_den = job.fps_den;
  out_idx_ = job.rec_idx;
  for (int c = 0; c < 3; c++) {
    int pw = c ? cw_ / 2 : cw_, ox = c ? job.crop[0] / 2 : job.crop[0], oy = c ? job.crop[2] / 2 : job.crop[2];
    out_.dev[c] = rec_[out_idx_][c] + (size_t)oy * pw + ox; out_.dev_pitch[c] = pw;
  }
  if (download_) {
    // Host pitches are kept even and aligned like a software decoder's line sizes: the reference
    // addresses chroma row i/2 as pvU + i * (nUPitch / 2) (openhevcfilter.cpp:209,224-227).
    size_t off = 0;
    const int ypitch = (out_.width + 63) & ~63;
    for (int c = 0; c < 3; c++) {
      int w = c ? out_.width / 2 : out_.width, h = c ? out_.height / 2 : out_.height, pitch = c ? ypitch / 2 : ypitch;
      if (hipMemcpy2DAsync(h_out_ + off, (size_t)pitch, out_.dev[c], (size_t)out_.dev_pitch[c], (size_t)w, (size_t)h, hipMemcpyDeviceToHost, stream_) != hipSuccess) return DEC_ERR_GPU;
      out_.host[c] = h_out_ + off; out_.host_pitch[c] = pitch;
      off += (size_t)pitch * h;
    }
    if (hipStreamSynchronize(stream_) != hipSuccess) return DEC_ERR_GPU;
  }
  pic_ready_ = true;
  return 1;
}

// ------------------------------------------------------------------------------------------ slice data (7.3.8)
// One task per WPP substream (CTU row), run by a pool of host threads.  Row r follows row r-1 at
// a distance of two CTUs: it starts from the context states saved after the second CTU of the row
// above and needs that row's CU records up to the above-right CTU.
int Decoder::parse_row(PicJob &job, int row, const uint8_t *data, size_t len, RowState &rs)
{
  const int wc = cw_ / 64, hc = ch_ / 64;
  const int slice_qp = job.slice_qp, max_merge = job.max_merge; const bool is_intra = job.is_intra;
  EncFrame &f = job.hf;
  FrameView v; v.f = &f;
  CabacDec c;
  const bool wpp = f.wpp != 0;
  const int T = job.tile_rows, chp = f.chp;
  const int first_cy = wpp ? row : tile_row_first(hc, T, row), ncy = wpp ? 1 : tile_row_first(hc, T, row + 1) - first_cy;
  int seen_above = 0;                                  // last observed progress of the row above (monotonic)
  auto wait_above = [&](int cy, int need) {            // CTUs of row cy-1 that must be complete
    if (!wpp || tile_row_starts_at(hc, T, cy)) return true;       // nothing above inside the tile
    if (need > wc) need = wc;
    if (seen_above < need) {
      std::atomic<int> &p = job.row_progress[(size_t)(cy - 1)].v;
      int spins = 0;
      while ((seen_above = p.load(std::memory_order_acquire)) < need) {
        if (++spins < 2000) __builtin_ia32_pause(); else { g_yields.fetch_add(1, std::memory_order_relaxed); std::this_thread::yield(); }
      }
    }
    return seen_above < (1 << 29);                     // >= 1 << 29: that row failed
  };
  int prev_qy = slice_qp;                              // qPY_PREV: slice QP at the start of a substream (tile, or CTU row with WPP)
  c.start(data, len);
  if (!wpp || tile_row_starts_at(hc, T, row)) cabac_init_contexts(c.ctx, is_intra ? 0 : 1, slice_qp);   // first CTU of a tile (9.3.1)
  else {
    if (!wait_above(row, 2)) return DEC_ERR_INVALID;
    memcpy(c.ctx, &job.wpp_saved[(size_t)(row - 1) * CTX_COUNT], CTX_COUNT);
  }
  uint64_t tc[6] = {0, 0, 0, 0, 0, 0}, t0 = TSC(), t1;
#define LAP(k) do { t1 = TSC(); tc[k] += t1 - t0; t0 = t1; } while (0)
  for (int cy = first_cy; cy < first_cy + ncy; cy++) {
    for (int cx = 0; cx < wc; cx++) {
      if (!wait_above(cy, cx + 2)) return DEC_ERR_INVALID;
      int ctu_qy = prev_qy, ctu_first = 64;                // quantisation group = CTU (8.6.1)
      if (f.sao) {                                         // sao() (7.3.8.3) opens the CTU
        SaoParams *sp = &f.sao[cy * wc + cx];
        const SaoParams *left = cx > 0 ? sp - 1 : nullptr, *up = (cy > 0 && !tile_row_starts_at(hc, T, cy)) ? sp - wc : nullptr;
        parse_sao(c, *sp, left, up, job.sao_luma != 0, job.sao_chroma != 0);
      }
      // coding_quadtree, iteratively in z-order over the 8x8 grid of the CTU
      for (int z = 0; z < 64;) {
        int xi, yi; ctu_z_to_xy(z, xi, yi);
        const int x0 = cx * 64 + xi * 8, y0 = cy * 64 + yi * 8;
        int log2 = 6;
        for (; log2 > 3; log2--) {                 // split_cu_flag at every level whose block starts here
          if (z & ((1 << (2 * (log2 - 3))) - 1)) continue;
          int depth = 6 - log2;
          int l = avail64(cw_, chp, x0, y0, x0 - 1, y0) && (6 - f.cu_log2[b8idx(f, x0 - 1, y0)]) > depth;
          int a = avail64(cw_, chp, x0, y0, x0, y0 - 1) && (6 - f.cu_log2[b8idx(f, x0, y0 - 1)]) > depth;
          if (!c.bin(CTX_SPLIT_CU + l + a)) break;
        }
        LAP(0);
        if (log2 == 6) return DEC_ERR_UNSUPPORTED;  // 64x64 coding units
        const int n = 1 << log2;
        int skip = 0, intra = is_intra ? 1 : 0, flags = 0, mode = 0, cbf = 0, mvx = 0, mvy = 0;
        if (!is_intra) {
          int l = avail64(cw_, chp, x0, y0, x0 - 1, y0) && (f.cu_flags[b8idx(f, x0 - 1, y0)] & CU_SKIP);
          int a = avail64(cw_, chp, x0, y0, x0, y0 - 1) && (f.cu_flags[b8idx(f, x0, y0 - 1)] & CU_SKIP);
          skip = c.bin(CTX_SKIP + l + a);
          if (!skip) intra = c.bin(CTX_PRED_MODE);
        }
        if (!is_intra && intra) return DEC_ERR_UNSUPPORTED;       // intra CUs in P pictures
        if (!intra && log2 == 3) return DEC_ERR_UNSUPPORTED;      // 8x8 inter CUs
        if (!skip && (!intra || log2 == 3) && !c.bin(CTX_PART_MODE)) return DEC_ERR_UNSUPPORTED;   // only PART_2Nx2N
        LAP(1);
        bool root_cbf = true;
        if (intra) {
          int prev = c.bin(CTX_PREV_INTRA);
          int cand[3]; intra_mpm(v, cw_, chp, x0, y0, cand);
          if (prev) { int idx = 0; if (c.bypass()) { idx = 1; if (c.bypass()) idx = 2; } mode = cand[idx]; }
          else {
            mode = (int)c.bypass_bits(5);
            int t;
            if (cand[0] > cand[1]) { t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
            if (cand[0] > cand[2]) { t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
            if (cand[1] > cand[2]) { t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
            for (int q = 0; q < 3; q++) if (mode >= cand[q]) mode++;
          }
          if (c.bin(CTX_CHROMA_MODE)) return DEC_ERR_UNSUPPORTED; // chroma mode other than "derived from luma"
        } else {
          int merge = skip ? 1 : c.bin(CTX_MERGE_FLAG);
          if (merge) {
            int idx = 0;
            if (max_merge > 1 && c.bin(CTX_MERGE_IDX)) { idx = 1; while (idx < max_merge - 1 && c.bypass()) idx++; }
            int cmx[5], cmy[5]; merge_cand_list(f, x0, y0, n, cmx, cmy);
            mvx = cmx[idx]; mvy = cmy[idx];
            flags = CU_MERGE | (skip ? CU_SKIP : 0);
            root_cbf = !skip;
          } else {
            int g0x = c.bin(CTX_MVD_GT0), g0y = c.bin(CTX_MVD_GT0);
            int g1x = g0x ? c.bin(CTX_MVD_GT1) : 0, g1y = g0y ? c.bin(CTX_MVD_GT1) : 0;
            int d[2];
            for (int k = 0; k < 2; k++) {
              int g0 = k ? g0y : g0x, g1 = k ? g1y : g1x, a = 0;
              if (g0) {
                a = 1;
                if (g1) { int kk = 1, vv = 0; while (kk < 32 && c.bypass()) { vv += 1 << kk; kk++; } if (kk >= 32) return DEC_ERR_INVALID; vv += (int)c.bypass_bits(kk); a = vv + 2; }
                if (c.bypass()) a = -a;
              }
              d[k] = a;
            }
            int mvp = c.bin(CTX_MVP_FLAG);
            int px[2], py[2]; amvp_cand_list(f, x0, y0, n, px, py);
            mvx = (int16_t)(uint16_t)(px[mvp] + d[0]); mvy = (int16_t)(uint16_t)(py[mvp] + d[1]);
            root_cbf = c.bin(CTX_RQT_ROOT_CBF) != 0;
          }
        }
        LAP(2);
        if (intra || root_cbf) {
          int cb = c.bin(CTX_CBF_CHROMA), cr = c.bin(CTX_CBF_CHROMA);
          int luma = (intra || cb || cr) ? c.bin(CTX_CBF_LUMA + 1) : 1;
          cbf = luma | (cb << 1) | (cr << 2);
          if (cbf && job.qp_in_cu && ctu_first == 64) {             // cu_qp_delta_abs / sign: once per CTU, in its first TU with a coded block
            int v = 0;
            while (v < 5 && c.bin(CTX_CU_QP_DELTA + (v ? 1 : 0))) v++;
            if (v == 5) { int k = 0; while (k < 16 && c.bypass()) { v += 1 << k; k++; } if (k >= 16) return DEC_ERR_INVALID; v += (int)c.bypass_bits(k); }
            if (v && c.bypass()) v = -v;
            if (v < -26 || v > 25) return DEC_ERR_INVALID;
            ctu_qy = (prev_qy + v + 52) % 52;
            ctu_first = z;
          }
          for (int ci = 0; ci < 3; ci++) {
            if (!((cbf >> ci) & 1)) continue;
            const int l2 = ci ? log2 - 1 : log2;
            TuDesc td; td.x = (uint16_t)(ci ? x0 >> 1 : x0); td.y = (uint16_t)(ci ? y0 >> 1 : y0); td.plane = (uint8_t)ci; td.log2 = (uint8_t)l2;
            td.offset = (uint32_t)rs.levels.size();
            if (!parse_residual(c, l2, ci, intra_scan_idx(intra, l2, ci, mode), rs.levels)) return DEC_ERR_INVALID;
            td.count = (uint16_t)(rs.levels.size() - td.offset);
            rs.tus.push_back(td);
          }
        }
        LAP(3);
        for (int yy = y0; yy < y0 + n; yy += 8)
          for (int xx = x0; xx < x0 + n; xx += 8) {
            int i = b8idx(f, xx, yy);
            f.cu_log2[i] = (uint8_t)log2; f.cu_intra[i] = (uint8_t)intra; f.cu_flags[i] = (uint8_t)flags;
            f.cu_intra_mode[i] = (uint8_t)mode; f.cu_cbf[i] = (uint8_t)cbf;
            f.cu_mv[i * 2] = (int16_t)mvx; f.cu_mv[i * 2 + 1] = (int16_t)mvy;
          }
        LAP(4);
        if (c.overrun()) return DEC_ERR_INVALID;
        z += 1 << (2 * (log2 - 3));
      }
      { const int ctu = cy * wc + cx; f.ctu_qy[ctu] = (int8_t)ctu_qy; f.ctu_delta[ctu] = (int8_t)(ctu_qy - prev_qy); f.ctu_first[ctu] = (uint8_t)ctu_first; prev_qy = ctu_qy; }
      if (wpp && cx == 1) memcpy(&job.wpp_saved[(size_t)cy * CTX_COUNT], c.ctx, CTX_COUNT);
      if (wpp) job.row_progress[(size_t)cy].v.store(cx + 1, std::memory_order_release);
      const bool last = (cy == hc - 1 && cx == wc - 1);
      int end = c.terminate();
      if (end != (last ? 1 : 0)) return DEC_ERR_UNSUPPORTED;      // slice must cover the whole picture
      if (!last && cx == wc - 1 && (wpp || tile_row_ends_at(hc, T, cy)) && !c.terminate()) return DEC_ERR_INVALID;   // end_of_subset_one_bit
      LAP(5);
    }
  }
  for (int k = 0; k < 6; k++) g_tc[k] += tc[k];
  return 0;
}

int Decoder::parse_job(PicJob &job, bool row_parallel)
{
  const uint8_t *data = job.rbsp.data() + job.data_off; const size_t len = job.data_len;
  const int hc = ch_ / 64, nsub = job.hf.wpp ? hc : job.tile_rows;                 // one substream per CTU row (WPP) or per tile
  if ((int)job.sub_start.size() != nsub) return DEC_ERR_INVALID;
  for (int r = 0; r < nsub; r++) if (job.sub_start[(size_t)r] >= len) return DEC_ERR_INVALID;
  job.rows.resize((size_t)nsub);
  for (auto &r : job.rows) { r.levels.clear(); r.tus.clear(); r.rc = 0; }
  job.wpp_saved.resize((size_t)hc * CTX_COUNT);
  if (!job.row_progress || job.row_progress_n < hc) { job.row_progress.reset(new Progress[(size_t)hc]); job.row_progress_n = hc; }
  for (int r = 0; r < hc; r++) job.row_progress[(size_t)r].v.store(0, std::memory_order_relaxed);
  auto one = [&](int r) {
    size_t start = job.sub_start[(size_t)r], end = (r + 1 < nsub) ? job.sub_start[(size_t)r + 1] : len;
    int rc = parse_row(job, r, data + start, end - start, job.rows[(size_t)r]);
    job.rows[(size_t)r].rc = rc;
    if (rc < 0 && job.hf.wpp) job.row_progress[(size_t)r].v.store(1 << 30, std::memory_order_release);   // release any waiter
  };
  if (row_parallel && nsub > 1) {
    if (!pool_) { const char *e = getenv("KVAZZUP_AMD_PARSE_THREADS"); if (e) parse_threads_ = atoi(e) < 1 ? 1 : atoi(e); pool_.reset(new OrderedPool(parse_threads_)); }
    pool_->run(nsub, one);
  } else {
    for (int r = 0; r < nsub; r++) one(r);              // frame-parallel mode: rows in sequence on this worker
  }
  // the rows' transform blocks and level words follow the CU records in the job's input block
  size_t ntu = 0, nlev = 0;
  for (auto &r : job.rows) { if (r.rc < 0) return r.rc; ntu += r.tus.size(); nlev += r.levels.size(); }
  const size_t tu_off = (fixed_bytes() + 15) & ~(size_t)15, lev_off = (tu_off + ntu * sizeof(TuDesc) + 15) & ~(size_t)15;
  if (!grow_job_input(job, lev_off + nlev * sizeof(uint32_t))) return DEC_ERR_GPU;
  TuDesc *tus = (TuDesc *)(job.h_in + tu_off); uint32_t *lev = (uint32_t *)(job.h_in + lev_off);
  size_t t = 0, l = 0;
  for (auto &r : job.rows) {
    for (TuDesc td : r.tus) { td.offset += (uint32_t)l; tus[t++] = td; }
    if (!r.levels.empty()) memcpy(lev + l, r.levels.data(), r.levels.size() * sizeof(uint32_t));
    l += r.levels.size();
  }
  job.ntu = ntu; job.nlev = nlev;
  return 0;
}

// ------------------------------------------------------------------------------------------ GPU reconstruction
int Decoder::launch_gpu(PicJob &job)
{
  const bool is_intra = job.is_intra, deblock = job.deblock; const int slice_qp = job.slice_qp;
  if (hipSetDevice(device_) != hipSuccess) return DEC_ERR_GPU;
  const size_t ntu = job.ntu, nlev = job.nlev;
  const size_t tu_off = (fixed_bytes() + 15) & ~(size_t)15, lev_off = (tu_off + ntu * sizeof(TuDesc) + 15) & ~(size_t)15;
  const size_t bytes = lev_off + nlev * sizeof(uint32_t);
  prof_now_ = profiling_ && (launched_ % prof_every_) == 0;
  Tick tk_api;
  if (bytes > d_in_cap_) {                               // (the stream is idle here: the previous picture has been completed)
    hipFree(d_in_);
    d_in_cap_ = bytes + bytes / 2;
    if (hipMalloc(&d_in_, d_in_cap_) != hipSuccess) { d_in_ = nullptr; d_in_cap_ = 0; return DEC_ERR_GPU; }
    bind_views(f_, d_in_);
  }
  if (hipMemcpyAsync(d_in_, job.h_in, bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) return DEC_ERR_GPU;
  const TuDesc *d_tus = (const TuDesc *)(d_in_ + tu_off); const uint32_t *d_lev = (const uint32_t *)(d_in_ + lev_off);
  f_.qp = slice_qp; f_.qpc = kChromaQp[slice_qp]; f_.is_intra = is_intra;
  f_.tile_rows = job.tile_rows; f_.chp = pack_height(ch_, job.tile_rows);
  bind_qp_arrays(f_, d_in_, cw_, ch_, job.qp_in_cu != 0);
  const int cur = (int)(launched_ % 3), ref = (int)((launched_ + 2) % 3);     // three buffers: the picture output by the previous call stays intact
  const bool sao = job.sao_luma || job.sao_chroma;                            // the picture is then built in work_ and filtered into the ring
  for (int c = 0; c < 3; c++) { f_.rec[c] = sao ? work_[c] : rec_[cur][c]; f_.sao_out[c] = rec_[cur][c]; f_.ref[c] = rec_[ref][c]; }
  f_.sao = sao ? (SaoParams *)(d_in_ + sao_offset()) : nullptr;
  const EncFrame f = f_;
  timed(DK_SCATTER, [&] { launch_scatter_levels(f, d_tus, (int)ntu, d_lev, stream_); });
  if (is_intra) {
    if (hipMemsetAsync(sync_, 0, sizeof(uint32_t) * 3 * (size_t)(ch_ / 64), stream_) != hipSuccess) return DEC_ERR_GPU;
    timed(DK_INTRA_RECON, [&] { launch_dec_intra_recon(f, stream_); });
  } else {
    timed(DK_INTER_RECON, [&] { launch_dec_inter_recon(f, stream_); });
  }
  if (deblock) timed(DK_DEBLOCK, [&] { launch_deblock(f, stream_); });
  if (sao) timed(DK_SAO, [&] { launch_dec_sao(f, stream_); });
  if (hipMemcpyAsync(h_err_, err_, sizeof(uint32_t), hipMemcpyDeviceToHost, stream_) != hipSuccess) return DEC_ERR_GPU;
  t_api_ += tk_api.ms();
  job.rec_idx = cur;
  launched_++;
  gpu_job_ = &job;
  return 0;
}

bool Decoder::get_picture(DecodedPicture *out)
{
  if (!pic_ready_) return false;
  *out = out_;
  return true;
}

bool Decoder::debug_copy(const char *what, void *dst, size_t bytes)
{
  if (!cw_) return false;
  const size_t npx = (size_t)cw_ * ch_;
  std::string w(what);
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    if (w == std::string("rec") + char('0' + c)) { if (bytes > n) return false; return hipMemcpy(dst, rec_[out_idx_][c], bytes, hipMemcpyDeviceToHost) == hipSuccess; }
  }
  return false;
}

}  // namespace kvzx
