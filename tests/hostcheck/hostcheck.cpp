// tests/hostcheck/hostcheck.cpp -- TEST-ONLY host build of the serial building blocks of the HIP
// path (kvazzup_amd/csrc/hevc_core.h, hevc_headers.h) so that `pytest -m "not gpu"` can check the
// product's CABAC writer, CU syntax, merge/AMVP signalling, intra prediction and deblocking
// against oracle/ without a GPU.  Nothing in the product links this file; the product has no CPU
// path (kvzx::Encoder::create fails without a HIP device).
#include <cstring>
#include <vector>
#include "../../kvazzup_amd/csrc/hevc_core.h"
#include "../../kvazzup_amd/csrc/hevc_headers.h"
#include "../../kvazzup_amd/csrc/entropy_host.h"
#include <chrono>

using namespace kvzx;

extern "C" {

// copy of a normative table for cross-checks: which = 0 dct32 (1024 int8), 1 rangeLps (256), 2 nextLps (64),
// 3 cabac init values (3*CTX_COUNT), 4 lambda (52 u16)
int hc_table(int which, void *dst)
{
  switch (which) {
    case 0: memcpy(dst, kDct32, sizeof(kDct32)); return (int)sizeof(kDct32);
    case 1: memcpy(dst, kRangeLps, sizeof(kRangeLps)); return (int)sizeof(kRangeLps);
    case 2: memcpy(dst, kNextLps, sizeof(kNextLps)); return (int)sizeof(kNextLps);
    case 3: memcpy(dst, kCabacInit, sizeof(kCabacInit)); return (int)sizeof(kCabacInit);
    case 4: memcpy(dst, kLambdaQ4, sizeof(kLambdaQ4)); return (int)sizeof(kLambdaQ4);
  }
  return 0;
}

struct HcFrame {
  int cw, ch, width, height, qp, is_intra, poc, wpp, deblock, fps_num, fps_den, write_ps;
  uint8_t *cu_log2, *cu_intra, *cu_flags, *cu_merge_idx, *cu_mvp_idx, *cu_intra_mode, *cu_cbf;
  int16_t *cu_mv, *cu_mvd;
  int16_t *coef[3];
};

static void fill(EncFrame &f, const HcFrame &h)
{
  memset(&f, 0, sizeof(f));
  f.cw = h.cw; f.ch = h.ch; f.tile_rows = 1; f.chp = pack_height(h.ch, 1); f.b8w = h.cw / 8; f.b8h = h.ch / 8; f.qp = h.qp; f.qpc = kChromaQp[h.qp];
  f.lambda_q4 = kLambdaQ4[h.qp]; f.is_intra = h.is_intra; f.poc = h.poc; f.wpp = h.wpp;
  f.cu_log2 = h.cu_log2; f.cu_intra = h.cu_intra; f.cu_flags = h.cu_flags; f.cu_merge_idx = h.cu_merge_idx;
  f.cu_mvp_idx = h.cu_mvp_idx; f.cu_intra_mode = h.cu_intra_mode; f.cu_cbf = h.cu_cbf; f.cu_mv = h.cu_mv; f.cu_mvd = h.cu_mvd;
  for (int c = 0; c < 3; c++) f.coef[c] = h.coef[c];
}

// Runs decide_signalling() for every inter CU: fills cu_flags / cu_merge_idx / cu_mvp_idx / cu_mvd.
void hc_inter_signal(HcFrame *h)
{
  EncFrame f; fill(f, *h);
  for (int y = 0; y < f.ch; y += 16)
    for (int x = 0; x < f.cw; x += 16) {
      int cl = f.cu_log2[b8idx(f, x, y)];
      if (cl == 5 && ((x | y) & 31)) continue;
      decide_signalling(f, x, y, cl);
    }
}

// Entropy-codes the picture described by the arrays (what k_entropy does, rows in sequence) and
// assembles the access unit.  Returns its size (or -needed if cap is too small).
int hc_encode_au(HcFrame *h, uint8_t *out, int cap, unsigned long long *bins)
{
  EncFrame f; fill(f, *h);
  const int wc = f.cw / 64, hc = f.ch / 64, nsub = f.wpp ? hc : 1;
  const int row_cap = f.cw * 64 * 3;
  std::vector<uint8_t> rows((size_t)row_cap * hc);
  std::vector<int32_t> lens(hc, 0);
  std::vector<std::vector<uint8_t>> rowv;
  uint8_t ctx[CTX_COUNT], saved[CTX_COUNT];
  static CoreTabs tabs; for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs, i);
  CabacEnc c; c.nbins = 0;
  const int init_type = f.is_intra ? 0 : 1;
  if (f.wpp) {
    for (int row = 0; row < hc; row++) {
      cabac_start(c, rows.data() + (size_t)row * row_cap, row_cap, ctx, &tabs);
      if (row == 0) cabac_init_contexts(ctx, init_type, f.qp); else memcpy(ctx, saved, sizeof(saved));
      for (int cx = 0; cx < wc; cx++) {
        enc_ctu(f, c, cx * 64, row * 64);
        if (cx == 1) memcpy(saved, ctx, sizeof(saved));
        bool last = (row == hc - 1 && cx == wc - 1);
        cabac_terminate(c, last);
        if (!last && cx == wc - 1) cabac_terminate(c, 1);
      }
      cabac_finish(c);
      lens[row] = c.pos;
    }
  } else {
    cabac_start(c, rows.data(), row_cap * hc, ctx, &tabs);
    cabac_init_contexts(ctx, init_type, f.qp);
    for (int cy = 0; cy < hc; cy++)
      for (int cx = 0; cx < wc; cx++) { enc_ctu(f, c, cx * 64, cy * 64); cabac_terminate(c, cy == hc - 1 && cx == wc - 1); }
    cabac_finish(c);
    lens[0] = c.pos;
  }
  if (bins) *bins = c.nbins;
  StreamParams sp; sp.cw = f.cw; sp.ch = f.ch; sp.width = h->width; sp.height = h->height; sp.qp = f.qp; sp.wpp = f.wpp;
  sp.deblock = h->deblock; sp.fps_num = h->fps_num; sp.fps_den = h->fps_den;
  std::vector<uint8_t> au;
  for (int r = 0; r < nsub; r++) rowv.emplace_back(rows.begin() + (size_t)r * row_cap, rows.begin() + (size_t)r * row_cap + lens[r]);
  assemble_access_unit(au, sp, f.is_intra != 0, f.poc, h->write_ps != 0, rowv, nsub);
  if ((int)au.size() > cap) return -(int)au.size();
  memcpy(out, au.data(), au.size());
  return (int)au.size();
}

// Same picture through the two-stage path of the product: every CTU is first turned into bin
// tokens the way k_tokenize does it (CU headers, then per transform block the last position and
// the sub-blocks, each sub-block tokenised independently with the greater1 carry taken from the
// next non-empty sub-block above it), then the tokens are replayed into the arithmetic coder the
// way entropy_host.h does it.  Must give the same access unit as hc_encode_au().
int hc_encode_au_tokens(HcFrame *h, uint8_t *out, int cap, unsigned long long *ntokens)
{
  EncFrame f; fill(f, *h);
  const int wc = f.cw / 64, hc = f.ch / 64, nsub = f.wpp ? hc : 1;
  static CoreTabs tabs; for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs, i);
  std::vector<std::vector<uint16_t>> ctu_tok((size_t)wc * hc);
  FrameView v; v.f = &f;
  unsigned long long total = 0;
  for (int cy = 0; cy < hc; cy++)
    for (int cx = 0; cx < wc; cx++) {
      std::vector<uint16_t> buf(65536);
      TokOut t; t.tabs = &tabs; t.p = buf.data(); t.n = 0; t.cap = (int)buf.size();
      for (int z = 0; z < 64;) {
        int xi, yi; ctu_z_to_xy(z, xi, yi);
        int x0 = cx * 64 + xi * 8, y0 = cy * 64 + yi * 8;
        CuRec cu = v.at(x0, y0);
        enc_split_flags(v, t, f.cw, f.chp, x0, y0, z, cu.log2);
        int cbf = enc_cu_header(v, t, f.cw, f.chp, f.is_intra != 0, x0, y0, cu);
        for (int ci = 0; ci < 3; ci++) {
          if (!((cbf >> ci) & 1)) continue;
          int l2 = ci ? cu.log2 - 1 : cu.log2, pw = ci ? f.cw / 2 : f.cw, px = ci ? x0 / 2 : x0, py = ci ? y0 / 2 : y0;
          int scan = intra_scan_idx(cu.intra, l2, ci, cu.intra_mode);
          TuDigest d; digest_build_serial(&tabs, d, f.coef[ci] + py * pw + px, pw, l2, scan);
          int last_sb, last_pos; enc_last_pos(t, d, l2, ci, scan, last_sb, last_pos);
          // independent per sub-block tokenisation, concatenated from last_sb downwards
          for (int i = last_sb; i >= 0; i--) {
            bool prev_g1 = false;
            uint64_t above = (i < 63) ? (d.sbmask >> (i + 1)) : 0;
            if (above) { int j = i + 1 + __builtin_ctzll(above); prev_g1 = subblock_g1_any(&tabs, d, j, scan); }
            uint16_t lt[160]; TokOut s2; s2.tabs = &tabs; s2.p = lt; s2.n = 0; s2.cap = 160;
            enc_subblock(s2, d, i, last_sb, last_pos, prev_g1, l2, ci, scan);
            if (s2.n > 128) return -1000000;                 // TOK_LANE_CAP of the kernel
            for (int k = 0; k < s2.n; k++) tok_push(t, lt[k]);
          }
        }
        z += 1 << (2 * (cu.log2 - 3));
      }
      bool last = (cy == hc - 1 && cx == wc - 1);
      cabac_terminate(t, last);
      if (f.wpp && !last && cx == wc - 1) cabac_terminate(t, 1);
      if (t.n > t.cap) return -2000000;
      buf.resize((size_t)t.n); total += (unsigned long long)t.n;
      ctu_tok[(size_t)cy * wc + cx] = buf;
    }
  if (ntokens) *ntokens = total;
  std::vector<std::vector<uint8_t>> rows((size_t)nsub);
  uint8_t ctx[CTX_COUNT], saved[CTX_COUNT];
  for (int r = 0; r < nsub; r++) {
    rows[(size_t)r].resize((size_t)f.cw * 64 * 3 * (f.wpp ? 1 : hc));
    CabacEnc c; c.nbins = 0;
    cabac_start(c, rows[(size_t)r].data(), (int)rows[(size_t)r].size(), ctx, &tabs);
    if (r == 0) cabac_init_contexts(ctx, f.is_intra ? 0 : 1, f.qp); else memcpy(ctx, saved, sizeof(saved));
    for (int cy = f.wpp ? r : 0; cy < (f.wpp ? r + 1 : hc); cy++)
      for (int cx = 0; cx < wc; cx++) {
        const std::vector<uint16_t> &tk = ctu_tok[(size_t)cy * wc + cx];
        cabac_play_tokens(c, tk.data(), (int)tk.size());
        if (f.wpp && cx == 1) memcpy(saved, ctx, sizeof(saved));
      }
    cabac_finish(c);
    rows[(size_t)r].resize((size_t)c.pos);
  }
  StreamParams sp; sp.cw = f.cw; sp.ch = f.ch; sp.width = h->width; sp.height = h->height; sp.qp = f.qp; sp.wpp = f.wpp;
  sp.deblock = h->deblock; sp.fps_num = h->fps_num; sp.fps_den = h->fps_den;
  std::vector<uint8_t> au;
  assemble_access_unit(au, sp, f.is_intra != 0, f.poc, h->write_ps != 0, rows, nsub);
  if ((int)au.size() > cap) return -(int)au.size();
  memcpy(out, au.data(), au.size());
  return (int)au.size();
}

// Intra prediction of one n x n block from explicit reference arrays (left/top as in hevc_core.h)
// hevc_core.h intra_uses_above_right / intra_uses_below_left: the masks the intra chains and the encoder's "intra-chain" mode restriction use
uint64_t hc_intra_uses(int log2n, int cidx, int below_left) { return below_left ? intra_uses_below_left(log2n, cidx) : intra_uses_above_right(log2n, cidx); }

void hc_intra_predict(const uint8_t *left, const uint8_t *top, int n, int cidx, int mode, uint8_t *pred)
{
  int l2 = ilog2((unsigned)n);
  uint8_t lf[65], tf[65];
  bool filt = intra_filter_needed(n, cidx, mode);
  if (filt) {
    bool strong = intra_strong_filter(left, top, n);
    for (int i = 0; i <= 2 * n; i++) { lf[i] = (uint8_t)intra_filtered_ref(left, top, n, i, strong); tf[i] = (uint8_t)intra_filtered_ref(top, left, n, i, strong); }
  }
  int s = n; for (int i = 0; i < n; i++) s += left[1 + i] + top[1 + i];
  int dc = s >> (l2 + 1);
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
    pred[y * n + x] = (uint8_t)intra_pred_sample(filt ? lf : left, filt ? tf : top, n, l2, cidx, mode, dc, x, y);
}

// Deblocking of a whole picture with the product's segment filters and on-the-fly boundary
// strengths (what k_deblock_v / k_deblock_h do), in place on rec planes (pitch cw, cw/2).
void hc_deblock(HcFrame *h, uint8_t *y, uint8_t *u, uint8_t *v)
{
  EncFrame f; fill(f, *h);
  f.rec[0] = y; f.rec[1] = u; f.rec[2] = v;
  int cw2 = f.cw >> 1;
  for (int yy = 0; yy < f.ch; yy += 4)
    for (int x = 8; x < f.cw; x += 8) {
      if (!is_cu_edge_v(f, x, yy)) continue;
      int bs = edge_bs(f, x - 1, yy, x, yy);
      if (!bs) continue;
      deblock_luma_segment(f.rec[0] + yy * f.cw + x, 1, f.cw, bs, f.qp);
      if (bs == 2 && (x & 15) == 0) {
        deblock_chroma_segment(f.rec[1] + (yy >> 1) * cw2 + (x >> 1), 1, cw2, 2, f.qp);
        deblock_chroma_segment(f.rec[2] + (yy >> 1) * cw2 + (x >> 1), 1, cw2, 2, f.qp);
      }
    }
  for (int yy = 8; yy < f.ch; yy += 8)
    for (int x = 0; x < f.cw; x += 4) {
      if (!is_cu_edge_h(f, x, yy)) continue;
      int bs = edge_bs(f, x, yy - 1, x, yy);
      if (!bs) continue;
      deblock_luma_segment(f.rec[0] + yy * f.cw + x, f.cw, 1, bs, f.qp);
      if (bs == 2 && (yy & 15) == 0) {
        deblock_chroma_segment(f.rec[1] + (yy >> 1) * cw2 + (x >> 1), cw2, 1, 2, f.qp);
        deblock_chroma_segment(f.rec[2] + (yy >> 1) * cw2 + (x >> 1), cw2, 1, 2, f.qp);
      }
    }
}

int hc_quant(int coef, int qp, int log2n, int intra) { return quant_level(coef, qp, log2n, intra); }
int hc_dequant(int level, int qp, int log2n) { return dequant_coef(level, qp, log2n); }
int hc_mvd_bits(int q) { return mvd_bits(q); }

// Microbenchmark of the host arithmetic coder's bin loop (entropy_host.h cabac_play_tokens_host) on the tokens of one picture:
// ns per token, all CTUs through one coder (contexts carried along; the byte output is discarded).  tools/arith_bench.py.
double hc_bench_play_tokens(const uint16_t *tok, long n, int reps, int variant)
{
  static CoreTabs tabs; for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs, i);
  static HostCabacTabs ht;
  std::vector<uint8_t> out((size_t)n * 2 + 64);
  uint8_t ctx[CTX_COUNT];
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    CabacEnc c; c.nbins = 0;
    cabac_start(c, out.data(), (int)out.size(), ctx, &tabs);
    cabac_init_contexts(ctx, 1, 32);
    auto t0 = std::chrono::steady_clock::now();
    if (variant == 0) cabac_play_tokens(c, tok, (int)n); else cabac_play_tokens_host(c, ht, tok, (int)n);
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    if (ns < best) best = ns;
  }
  return best / (double)n;
}
// the picture's tokens, CTU after CTU (what k_tok_compact delivers); returns the count (<= cap)
long hc_picture_tokens(HcFrame *h, uint16_t *dst, long cap);

}  // extern "C"

extern "C" long hc_picture_tokens(HcFrame *h, uint16_t *dst, long cap)
{
  EncFrame f; fill(f, *h);
  const int wc = f.cw / 64, hc = f.ch / 64;
  static CoreTabs tabs; for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs, i);
  std::vector<std::vector<uint16_t>> ctu_tok((size_t)wc * hc);
  FrameView v; v.f = &f;
  unsigned long long total = 0;
  for (int cy = 0; cy < hc; cy++)
    for (int cx = 0; cx < wc; cx++) {
      std::vector<uint16_t> buf(65536);
      TokOut t; t.tabs = &tabs; t.p = buf.data(); t.n = 0; t.cap = (int)buf.size();
      for (int z = 0; z < 64;) {
        int xi, yi; ctu_z_to_xy(z, xi, yi);
        int x0 = cx * 64 + xi * 8, y0 = cy * 64 + yi * 8;
        CuRec cu = v.at(x0, y0);
        enc_split_flags(v, t, f.cw, f.chp, x0, y0, z, cu.log2);
        int cbf = enc_cu_header(v, t, f.cw, f.chp, f.is_intra != 0, x0, y0, cu);
        for (int ci = 0; ci < 3; ci++) {
          if (!((cbf >> ci) & 1)) continue;
          int l2 = ci ? cu.log2 - 1 : cu.log2, pw = ci ? f.cw / 2 : f.cw, px = ci ? x0 / 2 : x0, py = ci ? y0 / 2 : y0;
          int scan = intra_scan_idx(cu.intra, l2, ci, cu.intra_mode);
          TuDigest d; digest_build_serial(&tabs, d, f.coef[ci] + py * pw + px, pw, l2, scan);
          int last_sb, last_pos; enc_last_pos(t, d, l2, ci, scan, last_sb, last_pos);
          // independent per sub-block tokenisation, concatenated from last_sb downwards
          for (int i = last_sb; i >= 0; i--) {
            bool prev_g1 = false;
            uint64_t above = (i < 63) ? (d.sbmask >> (i + 1)) : 0;
            if (above) { int j = i + 1 + __builtin_ctzll(above); prev_g1 = subblock_g1_any(&tabs, d, j, scan); }
            uint16_t lt[160]; TokOut s2; s2.tabs = &tabs; s2.p = lt; s2.n = 0; s2.cap = 160;
            enc_subblock(s2, d, i, last_sb, last_pos, prev_g1, l2, ci, scan);
            if (s2.n > 128) return -1000000;                 // TOK_LANE_CAP of the kernel
            for (int k = 0; k < s2.n; k++) tok_push(t, lt[k]);
          }
        }
        z += 1 << (2 * (cu.log2 - 3));
      }
      bool last = (cy == hc - 1 && cx == wc - 1);
      cabac_terminate(t, last);
      if (f.wpp && !last && cx == wc - 1) cabac_terminate(t, 1);
      if (t.n > t.cap) return -2000000;
      buf.resize((size_t)t.n); total += (unsigned long long)t.n;
      ctu_tok[(size_t)cy * wc + cx] = buf;
    }
  long n = 0;
  for (auto &v2 : ctu_tok) for (uint16_t t : v2) { if (n < cap) dst[n] = t; n++; }
  (void)total;
  return n;
}

