// kvazzup_amd/csrc/entropy_host.h -- the serial half of entropy coding, on host threads.
//
// Binarisation and context selection -- everything in CABAC that can be done in parallel -- runs
// on the GPU (k_tokenize, enc_kernels.hip) and produces, per CTU, the list of bins in coding
// order as 16-bit tokens.  What is left is the arithmetic coder proper: one state machine per
// WPP substream (CTU row), each bin depending on the previous one.  A GPU lane does that at
// ~80 ns per bin; a host core at ~8 ns.  So the rows of a picture are coded here by a small pool
// of host threads (the reference's encoder does its CABAC on host threads too: Kvazaar's WPP
// worker pool behind kvz_api->encoder_encode, kvazaarfilter.cpp:176-194 "threads"/"wpp").
// Row r starts from the context states row r-1 had after its second CTU (H.265 9.3.2.2).
#pragma once
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include "hevc_core.h"
#include "host_pool.h"

namespace kvzx {

// cabac_play_tokens() of hevc_core.h for the host: the same arithmetic with the coder's registers in locals (the context array is
// written through a byte pointer, which the compiler must assume to alias the coder's own fields: kept in the struct they are
// reloaded after every bin), the context update as one table look-up and the renormalisation shift from the leading-zero count.
struct HostCabacTabs {
  uint8_t next_mps[128], next_lps[128];     // context variable = pStateIdx << 1 | valMps
  uint8_t next[128][2];                     // the same, indexed by [variable][least probable symbol coded]
  uint8_t lps[128][4];                      // rangeTabLps by context variable and (range >> 6) & 3
  HostCabacTabs()
  {
    for (int s = 0; s < 128; s++) {
      const int st = s >> 1, mps = s & 1;
      next_mps[s] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
      next_lps[s] = (uint8_t)((kNextLps[st] << 1) | (st == 0 ? mps ^ 1 : mps));
      next[s][0] = next_mps[s]; next[s][1] = next_lps[s];
      for (int q = 0; q < 4; q++) lps[s][q] = kRangeLps[st][q];
    }
  }
};
inline void cabac_play_tokens_host(CabacEnc &c, const HostCabacTabs &T, const uint16_t *tok, int n)
{
  uint32_t low = c.low, range = c.range; int bits_left = c.bits_left;
  uint8_t *const ctx = c.ctx;
  uint32_t nbins = 0;
  auto spill = [&] { c.low = low; c.range = range; c.bits_left = bits_left; };
  auto fill = [&] { low = c.low; range = c.range; bits_left = c.bits_left; };
  for (int i = 0; i < n; i++) {
    const uint32_t t = tok[i];
    if (__builtin_expect(!(t & 0x8000u), 1)) {
      // both outcomes computed and selected (the bin values of sig / greater1 flags are close to coin flips for a branch predictor):
      // either way the new range is renormalised by its leading zeros
      const uint32_t ci = t >> 1, s = ctx[ci];
      const uint32_t lps = T.lps[s][(range >> 6) & 3];
      const uint32_t rmps = range - lps;
      const uint32_t isl = (t ^ s) & 1u;
      const uint32_t r = isl ? lps : rmps;
      const int nb = __builtin_clz(r) - 23;                               // r in [6, 510] -> [256, 510] (an MPS range is at least 128: one bit at most)
      nbins++;
      low = (low + (isl ? rmps : 0u)) << nb; range = r << nb; bits_left -= nb;
      ctx[ci] = T.next[s][isl];
      if (__builtin_expect(bits_left < 12, 0)) { spill(); cabac_write_out(c); fill(); }
    } else {
      spill();
      if (!(t & 0x4000u)) cabac_bypass_bits(c, t & 0x3ffu, (int)((t >> 10) & 15) + 1);
      else cabac_terminate(c, (int)(t & 1));
      fill();
    }
  }
  spill();
  c.nbins += nbins;
}

class EntropyHost {
 public:
  explicit EntropyHost(int max_threads) : pool_(max_threads) { for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs_, i); }
  // Codes one picture.  tokens: dense token array; CTU i has count[i] tokens starting at offset[i].
  // rows_out[r] receives the bytes of substream r (one per CTU row with WPP, else a single one).
  void code_picture(const uint16_t *tokens, const int32_t *count, const uint32_t *offset, int wc, int hc, bool wpp, int tile_rows, int init_type, int qp,
                    std::vector<std::vector<uint8_t>> &rows_out, uint64_t *bins, int tile_cols = 1)
  {
    tokens_ = tokens; count_ = count; offset_ = offset; wc_ = wc; hc_ = hc; wpp_ = wpp; tiles_ = tile_rows < 1 ? 1 : tile_rows; init_type_ = init_type; qp_ = qp;
    cols_ = tile_cols < 1 ? 1 : tile_cols;
    build_geometry();
    const int nsub = (int)geom_.size();
    rows_out.resize((size_t)nsub);
    rows_ = &rows_out;
    saved_.resize((size_t)hc * cols_ * CTX_COUNT);
    bins_.store(0);
    run_rows(nsub, 0);
    if (bins) *bins = bins_.load();
  }
  // The substreams of CTU rows [row0, row0 + nrows) only (whole tiles): rows_out[k] = substream of CTU row row0 + k with WPP,
  // else of the k-th tile of the band.  Arrays are indexed by picture CTU as in code_picture().
  void code_band(const uint16_t *tokens, const int32_t *count, const uint32_t *offset, int wc, int hc, bool wpp, int tile_rows, int init_type, int qp,
                 int row0, int nrows, std::vector<std::vector<uint8_t>> &rows_out, uint64_t *bins)
  {
    tokens_ = tokens; count_ = count; offset_ = offset; wc_ = wc; hc_ = hc; wpp_ = wpp; tiles_ = tile_rows < 1 ? 1 : tile_rows; init_type_ = init_type; qp_ = qp;
    cols_ = 1;                                       // (bands are whole tile rows of full-width tiles)
    build_geometry();
    const int first = wpp ? row0 : tile_row_of(hc, tiles_, row0);
    const int nsub = wpp ? nrows : tile_row_of(hc, tiles_, row0 + nrows - 1) - first + 1;
    all_rows_.resize((size_t)(wpp ? hc : tiles_));
    rows_ = &all_rows_;
    saved_.resize((size_t)hc * CTX_COUNT);
    bins_.store(0);
    run_rows(nsub, first);
    rows_out.resize((size_t)nsub);
    for (int k = 0; k < nsub; k++) rows_out[(size_t)k].swap(all_rows_[(size_t)(first + k)]);
    if (bins) *bins = bins_.load();
  }

 private:
  // A picture with very few tokens (a still scene: all skip) is coded by the calling thread, row after row: handing
  // few-microsecond rows to a pool costs more in wake-ups than the rows take.  Everything else goes to the pool, where row r
  // follows row r - 1 at a distance of two CTUs (a 1080p inter picture of the benchmark clip has some 100 000 tokens, 0.8 ms
  // of coding on one core).
  // WPP: the contexts every CTU row starts from (those of the row above after its second CTU) for ALL rows ahead of the coding, on the
  // calling thread: a context variable follows the bin values alone, not the arithmetic coder's registers, so the hand-over chain is a
  // replay of two CTUs' tokens per row through the state table (~10 us per 1080p picture).  The rows are then coded side by side
  // without ever waiting for each other (before: every row's thread spun until the row above had coded two CTUs -- a quarter of
  // the coder threads' CPU time went into that wait at 6000 frames/s).
  void prepass_contexts(int nsub, int first)
  {
    uint8_t ctx[CTX_COUNT];
    for (int k = 0; k < nsub; k++) {
      const Sub g = geom_[(size_t)(first + k)];
      if (g.cx1 - g.cx0 < 2) continue;                                   // (a one-CTU-wide tile: every row starts from the initial values)
      if (g.cy0 == g.tile_cy0) cabac_init_contexts(ctx, init_type_, qp_);
      else memcpy(ctx, &saved_[((size_t)(g.cy0 - 1) * cols_ + g.tc) * CTX_COUNT], CTX_COUNT);
      for (int cx = g.cx0; cx < g.cx0 + 2; cx++) {
        const size_t ctu = (size_t)g.cy0 * wc_ + cx;
        const uint16_t *tok = tokens_ + offset_[ctu];
        for (int i = 0, n = count_[ctu]; i < n; i++) {
          const uint32_t t = tok[i];
          if (t & 0x8000u) continue;                                     // bypass / terminating bins: no context
          const uint32_t ci = t >> 1, s = ctx[ci];
          ctx[ci] = ((t ^ s) & 1u) ? htabs_.next_lps[s] : htabs_.next_mps[s];
        }
      }
      memcpy(&saved_[((size_t)g.cy0 * cols_ + g.tc) * CTX_COUNT], ctx, CTX_COUNT);
    }
  }
  void run_rows(int nsub, int first)
  {
    if (wpp_) prepass_contexts(nsub, first);
    size_t ntok = 0;
    for (int i = 0; i < wc_ * hc_; i++) ntok += (size_t)count_[i];
    if (ntok < 16000) { for (int k = 0; k < nsub; k++) code_row(first + k); }
    else pool_.run(nsub, [this, first](int k) { code_row(first + k); });
  }
  // the substreams in decoding order (6.5.1 tile scan: tile after tile; with WPP every CTU row of a tile is one)
  struct Sub { int cy0, cy1, cx0, cx1, tile_cy0, tc; };
  void build_geometry()
  {
    geom_.clear();
    for (int tr = 0; tr < tiles_; tr++)
      for (int tc = 0; tc < cols_; tc++) {
        Sub g; g.tile_cy0 = tile_row_first(hc_, tiles_, tr); g.cx0 = tile_col_first(wc_, cols_, tc); g.cx1 = tile_col_first(wc_, cols_, tc + 1); g.tc = tc;
        const int tile_cy1 = tile_row_first(hc_, tiles_, tr + 1);
        if (wpp_) for (int cy = g.tile_cy0; cy < tile_cy1; cy++) { g.cy0 = cy; g.cy1 = cy + 1; geom_.push_back(g); }
        else { g.cy0 = g.tile_cy0; g.cy1 = tile_cy1; geom_.push_back(g); }
      }
  }
  // substream r of the geometry list: a CTU row of a tile with WPP, else a tile
  void code_row(int r)
  {
    uint8_t ctx[CTX_COUNT];
    std::vector<uint8_t> &out = (*rows_)[(size_t)r];
    const Sub g = geom_[(size_t)r];
    size_t ntok = 0;
    for (int cy = g.cy0; cy < g.cy1; cy++) for (int cx = g.cx0; cx < g.cx1; cx++) ntok += (size_t)count_[(size_t)cy * wc_ + cx];
    out.resize(ntok * 2 + 64);                    // a token never produces more than two bytes
    CabacEnc c; c.nbins = 0;
    cabac_start(c, out.data(), (int)out.size(), ctx, &tabs_);
    // contexts: initialised at the first CTU of a tile (9.3.1), else (WPP) inherited from the row above inside the tile after its second
    // CTU -- when the tile is at least two CTUs wide
    const bool fresh = !wpp_ || g.cy0 == g.tile_cy0 || g.cx1 - g.cx0 < 2;
    if (fresh) cabac_init_contexts(ctx, init_type_, qp_);
    else {
      const size_t above = (size_t)(g.cy0 - 1) * cols_ + g.tc;                 // (prepass_contexts)
      memcpy(ctx, &saved_[above * CTX_COUNT], CTX_COUNT);
    }
    for (int cy = g.cy0; cy < g.cy1; cy++)
      for (int cx = g.cx0; cx < g.cx1; cx++) {
        const size_t ctu = (size_t)cy * wc_ + cx;
        cabac_play_tokens_host(c, htabs_, tokens_ + offset_[ctu], count_[ctu]);
      }
    cabac_finish(c);
    out.resize((size_t)c.pos);
    bins_.fetch_add(c.nbins);
  }

  OrderedPool pool_;
  CoreTabs tabs_;
  HostCabacTabs htabs_;
  const uint16_t *tokens_ = nullptr; const int32_t *count_ = nullptr; const uint32_t *offset_ = nullptr;
  int wc_ = 0, hc_ = 0, tiles_ = 1, cols_ = 1, init_type_ = 0, qp_ = 0; bool wpp_ = true;
  std::vector<Sub> geom_;
  std::vector<uint8_t> saved_;
  std::vector<std::vector<uint8_t>> *rows_ = nullptr;
  std::vector<std::vector<uint8_t>> all_rows_;
  std::atomic<uint64_t> bins_{0};
};

}  // namespace kvzx
