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

class EntropyHost {
 public:
  explicit EntropyHost(int max_threads) : pool_(max_threads) { for (int i = 0; i < 64; i++) core_tabs_fill_entry(tabs_, i); }
  // Codes one picture.  tokens: dense token array; CTU i has count[i] tokens starting at offset[i].
  // rows_out[r] receives the bytes of substream r (one per CTU row with WPP, else a single one).
  void code_picture(const uint16_t *tokens, const int32_t *count, const uint32_t *offset, int wc, int hc, bool wpp, int tile_rows, int init_type, int qp,
                    std::vector<std::vector<uint8_t>> &rows_out, uint64_t *bins)
  {
    tokens_ = tokens; count_ = count; offset_ = offset; wc_ = wc; hc_ = hc; wpp_ = wpp; tiles_ = tile_rows < 1 ? 1 : tile_rows; init_type_ = init_type; qp_ = qp;
    const int nsub = wpp ? hc : tiles_;
    rows_out.resize((size_t)nsub);
    rows_ = &rows_out;
    saved_.resize((size_t)hc * CTX_COUNT);
    ready_.reset(new std::atomic<int>[(size_t)hc]);
    for (int r = 0; r < hc; r++) ready_[(size_t)r].store(0, std::memory_order_relaxed);
    bins_.store(0);
    pool_.run(nsub, [this](int r) { code_row(r); });
    if (bins) *bins = bins_.load();
  }
  // The substreams of CTU rows [row0, row0 + nrows) only (whole tiles): rows_out[k] = substream of CTU row row0 + k with WPP,
  // else of the k-th tile of the band.  Arrays are indexed by picture CTU as in code_picture().
  void code_band(const uint16_t *tokens, const int32_t *count, const uint32_t *offset, int wc, int hc, bool wpp, int tile_rows, int init_type, int qp,
                 int row0, int nrows, std::vector<std::vector<uint8_t>> &rows_out, uint64_t *bins)
  {
    tokens_ = tokens; count_ = count; offset_ = offset; wc_ = wc; hc_ = hc; wpp_ = wpp; tiles_ = tile_rows < 1 ? 1 : tile_rows; init_type_ = init_type; qp_ = qp;
    const int first = wpp ? row0 : tile_row_of(hc, tiles_, row0);
    const int nsub = wpp ? nrows : tile_row_of(hc, tiles_, row0 + nrows - 1) - first + 1;
    all_rows_.resize((size_t)(wpp ? hc : tiles_));
    rows_ = &all_rows_;
    saved_.resize((size_t)hc * CTX_COUNT);
    ready_.reset(new std::atomic<int>[(size_t)hc]);
    for (int r = 0; r < hc; r++) ready_[(size_t)r].store(0, std::memory_order_relaxed);
    bins_.store(0);
    pool_.run(nsub, [this, first](int k) { code_row(first + k); });
    rows_out.resize((size_t)nsub);
    for (int k = 0; k < nsub; k++) rows_out[(size_t)k].swap(all_rows_[(size_t)(first + k)]);
    if (bins) *bins = bins_.load();
  }

 private:
  // substream r: CTU row r with WPP, else tile row r (all of its CTU rows)
  void code_row(int r)
  {
    uint8_t ctx[CTX_COUNT];
    std::vector<uint8_t> &out = (*rows_)[(size_t)r];
    const int first_cy = wpp_ ? r : tile_row_first(hc_, tiles_, r), ncy = wpp_ ? 1 : tile_row_first(hc_, tiles_, r + 1) - first_cy;
    size_t ntok = 0;
    for (int cy = first_cy; cy < first_cy + ncy; cy++) for (int cx = 0; cx < wc_; cx++) ntok += (size_t)count_[(size_t)cy * wc_ + cx];
    out.resize(ntok * 2 + 64);                    // a token never produces more than two bytes
    CabacEnc c; c.nbins = 0;
    cabac_start(c, out.data(), (int)out.size(), ctx, &tabs_);
    // contexts: initialised at the first CTU of a tile (9.3.1), else (WPP) inherited from the row above after its second CTU
    const bool fresh = !wpp_ || tile_row_starts_at(hc_, tiles_, r);
    if (fresh) cabac_init_contexts(ctx, init_type_, qp_);
    else {
      while (!ready_[(size_t)(r - 1)].load(std::memory_order_acquire)) std::this_thread::yield();
      memcpy(ctx, &saved_[(size_t)(r - 1) * CTX_COUNT], CTX_COUNT);
    }
    for (int cy = first_cy; cy < first_cy + ncy; cy++)
      for (int cx = 0; cx < wc_; cx++) {
        const size_t ctu = (size_t)cy * wc_ + cx;
        cabac_play_tokens(c, tokens_ + offset_[ctu], count_[ctu]);
        if (wpp_ && cx == 1) {
          memcpy(&saved_[(size_t)r * CTX_COUNT], ctx, CTX_COUNT);
          ready_[(size_t)r].store(1, std::memory_order_release);
        }
      }
    cabac_finish(c);
    out.resize((size_t)c.pos);
    bins_.fetch_add(c.nbins);
  }

  OrderedPool pool_;
  CoreTabs tabs_;
  const uint16_t *tokens_ = nullptr; const int32_t *count_ = nullptr; const uint32_t *offset_ = nullptr;
  int wc_ = 0, hc_ = 0, tiles_ = 1, init_type_ = 0, qp_ = 0; bool wpp_ = true;
  std::vector<uint8_t> saved_;
  std::unique_ptr<std::atomic<int>[]> ready_;
  std::vector<std::vector<uint8_t>> *rows_ = nullptr;
  std::vector<std::vector<uint8_t>> all_rows_;
  std::atomic<uint64_t> bins_{0};
};

}  // namespace kvzx
