// kvazzup_amd/csrc/encoder.hip -- see encoder.h
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "encoder.h"
#include "enc_kernels.h"
#include "stream_pool.h"
#include "scaling_tables.h"
#include "pic_hash.h"

namespace kvzx {

// Events that only order this library's streams among themselves (never waited for or queried by the host): no system-scope fence when they complete --
// the default makes every record write back and invalidate the caches for the host's benefit, a bubble of its own in a chain of 10-50 us kernels.
// KVAZZUP_AMD_EVENT_FENCE=1 restores the default (measurement aid).
static const unsigned kDeviceEvent = hipEventDisableTiming | (getenv("KVAZZUP_AMD_EVENT_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (error) *error = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "kvazzup_amd: %s failed: %s\n", #expr, hipGetErrorString(e_)); return false; } } while (0)

Encoder *Encoder::create(const EncoderConfig &cfg, std::string *error)
{
  Encoder *e = new Encoder();
  if (!e->init(cfg, error)) { delete e; return nullptr; }
  return e;
}

bool Encoder::init(const EncoderConfig &cfg_in, std::string *error)
{
  EncoderConfig cfg = cfg_in;
  if (cfg.vaq > 0) cfg.qp_in_cu = 1;                       // the deltas travel as cu_qp_delta
  if (cfg.lossless) { cfg.deblock = 0; cfg.sao = 0; cfg.rdoq = 0; cfg.signhide = 0; cfg.bitrate = 0; cfg.rc_bands = 0; }      // (oracle/hevc_enc.c "lossless")
  if (cfg.bitrate <= 0 || cfg.band_rows > 0) cfg.rc_bands = 0;
  if (cfg.rc_bands > 8) cfg.rc_bands = 8;                  // (RcState::acc)
  if ((cfg.slices == 1 && !cfg.wpp) || (cfg.slices == 2 && cfg.tile_rows * cfg.tile_cols < 2) || (cfg.slices == 1 && cfg.tile_cols > 1) || cfg.slices < 0 || cfg.slices > 2) cfg.slices = 0;
  if (cfg.band_rows > 0) cfg.intra_in_p = 0;               // (intra-in-P runs behind the whole picture's inter reconstruction: not in band mode, where a picture is coded in parts by several instances)
  if (cfg.rc_bands > 0) cfg.qp_in_cu = 1;                  // ... and so do the steps of rate control v2
  if (cfg.band_rows > 0) cfg.me_source = 0;                // (a band's search window reaches into the neighbouring bands' rows: the halo carries reconstruction rows, not source rows)

  const char *prio = getenv("KVAZZUP_AMD_PRIO"); if (!prio || strlen(prio) < 4) prio = "hnnn";   // main, tokenizer, input, decoder: the chain the next picture waits for is the urgent one (+6 % at 1080p; any explicit priority also gives the stream a hardware queue of its own)

  if (cfg.width < 16 || cfg.height < 16 || (cfg.width & 1) || (cfg.height & 1) || cfg.width > 16384 || cfg.height > 16384) {
    if (error) *error = "unsupported picture size"; return false;
  }
  if (cfg.qp < 0 || cfg.qp > 51 || cfg.me_range < 1 || cfg.me_range > 32) { if (error) *error = "qp or me-range out of range"; return false; }
  if (cfg.tile_cols < 1) cfg.tile_cols = 1;
  if (cfg.tile_rows < 1 || cfg.tile_rows > (cfg.height + 63) / 64 || cfg.tile_rows > 22 || cfg.tile_cols > 20 || cfg.tile_cols > (cfg.width + 63) / 64 || (cfg.band_rows > 0 && cfg.tile_cols > 1)) {
    if (error) *error = "tile grid out of range (at most 20 x 22 tiles, none smaller than a CTU; band mode: full-width tile rows)"; return false;
  }
  if (cfg.tile_cols > 1) cfg.entropy_gpu = 0;                // (k_cabac_rows runs full-width substreams)
  if (cfg.band_rows > 0) {
    const int hc = (cfg.height + 63) / 64, T = cfg.tile_rows;
    const bool ok = cfg.band_row0 >= 0 && cfg.band_row0 + cfg.band_rows <= hc && tile_row_starts_at(hc, T, cfg.band_row0) &&
                    tile_row_ends_at(hc, T, cfg.band_row0 + cfg.band_rows - 1) && !cfg.sao && cfg.vaq == 0;
    if (!ok) { if (error) *error = "a band must consist of whole tile rows (and SAO and VAQ are not available in band mode)"; return false; }
  }
  if (const char *e = getenv("KVAZZUP_AMD_ENTROPY")) cfg.entropy_gpu = strcmp(e, "gpu") == 0;    // A/B knob: host | gpu
  if (cfg.band_rows > 0) cfg.entropy_gpu = 0;               // (band mode hands its substreams to the caller from the host pool)
  if (cfg.band_rows > 0) cfg.hash = 0;                      // (a band encoder holds a part of the picture only)
  cfg_ = cfg;
  qp_cur_ = cfg.qp;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg.device) {
    if (error) *error = "no usable HIP device (this library has no CPU fallback)"; return false;
  }
  HIP_OK(hipSetDevice(cfg.device));
  cw_ = (cfg.width + 63) & ~63; ch_ = (cfg.height + 63) & ~63;
  if (cw_ < 128) cw_ = 128;
  rows_ = ch_ / 64;
  // pictures in flight behind the one being submitted (output lag).  Rate control books picture t - 3 before picture t: lag <= 2.  Beyond
  // 3 the gain is buffering: while an intra picture's chain holds the main stream (1.6 ms at 1080p, ten picture intervals) the coder threads
  // work off the pictures queued before it (1080p, host-bound: owf 3 -> 4 measured +4 %; more changes nothing).  The GPU arithmetic coder
  // is a longer stage than the host pool (a substream is one serial chain) and profits from up to 8.
  // With a bitrate the controller books picture t - rc_delay_ before picture t, rc_delay_ = pictures in flight + 1 (3 .. 7: the size ring holds eight):
  // a pipelined encoder then decides exactly like a synchronous one with the same delay (oracle/hevc_enc.c rate_control, "rc-delay").
  depth_ = cfg.owf >= 3 ? (cfg.bitrate == 0 ? (cfg.owf > kMaxDepth ? kMaxDepth : cfg.owf) : (cfg.owf > 6 ? 6 : cfg.owf)) : (cfg.owf >= 2 ? 2 : (cfg.owf == 1 ? 1 : 0));
  rc_delay_ = cfg.bitrate > 0 && depth_ + 1 > 3 && cfg.band_rows == 0 ? depth_ + 1 : 3;
  nrec_ = depth_ + 2 < 3 ? 3 : depth_ + 2;              // (+ 1: an intra picture is written ahead of its turn, beside the P pictures in front of it, which still read theirs)
  prio_[0] = prio[0]; prio_[1] = prio[1]; prio_[2] = prio[2];
  HIP_OK(stream_acquire(&stream_, cfg.device, 'M', prio_[0]));
  const size_t npx = (size_t)cw_ * ch_, nb8 = npx / 64, in_bytes = (size_t)cfg.width * cfg.height * 3 / 2;
  (void)in_bytes;                                          // (the host-picture ring is allocated by the first encode_host)
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    for (int k = 0; k < kSets; k++) HIP_OK(hipMalloc(&src_[k][c], n));
    for (int b = 0; b < nrec_; b++) { HIP_OK(hipMalloc(&rec_[b][c], n)); HIP_OK(hipMemset(rec_[b][c], 0, n)); }
    for (int k = 0; k < kSets; k++) { HIP_OK(hipMalloc(&coef_[k][c], n * sizeof(int16_t))); HIP_OK(hipMemset(coef_[k][c], 0, n * sizeof(int16_t))); }
  }
  for (int k = 0; k < kSets; k++) {
    HIP_OK(hipMalloc(&cu_bytes_[k], nb8 * 7)); HIP_OK(hipMemset(cu_bytes_[k], 0, nb8 * 7));
    HIP_OK(hipMalloc(&cu_mv_[k], nb8 * 2 * sizeof(int16_t))); HIP_OK(hipMemset(cu_mv_[k], 0, nb8 * 2 * sizeof(int16_t)));
    HIP_OK(hipMalloc(&cu_mvd_[k], nb8 * 2 * sizeof(int16_t))); HIP_OK(hipMemset(cu_mvd_[k], 0, nb8 * 2 * sizeof(int16_t)));
    HIP_OK(hipEventCreateWithFlags(&ev_tok_done_[k], kDeviceEvent));
  }
  if (cfg.qp_in_cu) {
    const size_t nctu = (size_t)(cw_ / 64) * rows_;
    for (int k = 0; k < kSets; k++) {
      HIP_OK(hipMalloc(&ctu_qt_[k], nctu)); HIP_OK(hipMalloc(&ctu_qy_[k], nctu)); HIP_OK(hipMalloc(&ctu_delta_[k], nctu)); HIP_OK(hipMalloc(&ctu_first_[k], nctu));
      HIP_OK(hipHostMalloc(&h_ctu_qt_[k], nctu, hipHostMallocDefault));
      HIP_OK(hipMalloc(&ctu_roi_[k], nctu));
    }
    if (cfg.vaq > 0) { HIP_OK(hipMalloc(&vaq_act_, nctu * sizeof(int))); HIP_OK(hipMalloc(&vaq_sum_, sizeof(int))); }
  }
  if (cfg.sao) {
    for (int c = 0; c < 3; c++) HIP_OK(hipMalloc(&work_[c], c ? npx / 4 : npx));
    for (int k = 0; k < kSets; k++) HIP_OK(hipMalloc(&sao_[k], sizeof(SaoParams) * (size_t)(cw_ / 64) * rows_));
  }
  if (cfg.rc_bands > 0) { HIP_OK(hipMalloc(&rc_state_, sizeof(RcState))); HIP_OK(hipMemset(rc_state_, 0, sizeof(RcState))); }
  HIP_OK(stream_acquire(&stream_tok_, cfg.device, 'T', prio_[1]));
  HIP_OK(stream_acquire(&stream_in_, cfg.device, 'I', prio_[2]));
  for (int k = 0; k < kSets; k++) HIP_OK(hipEventCreateWithFlags(&ev_src_free_[k], kDeviceEvent));
  HIP_OK(hipEventCreateWithFlags(&ev_signalled_, kDeviceEvent));
  // the intra pictures' own stream: where the chain shares nothing with the P pictures' kernels (no SAO work picture, no intra units in P pictures
  // -- they use the same progress counters --, no per-CTU QP upload, no row groups) and pictures are queued ahead at all
  // (round 4: SAO, intra units in P pictures, per-CTU QPs and the row groups of rate control v2 no longer keep it on the main stream -- the side chain has its own
  // progress counters, edge columns and SAO work picture; VAQ still does: its activity scratch is shared)
  // (not when every picture is an intra picture: then the chains ARE the main stream's work, and the next picture's analysis belongs beside them on the
  // input stream, not behind them -- all-intra 1080p 940 -> 1 300 frames/s)
  idr_side_ = depth_ >= 2 && cfg.vaq == 0 && cfg.band_rows == 0 && cfg.intra_period != 1 && !getenv("KVAZZUP_AMD_IDR_INLINE");
  // Every picture an intra picture (BASELINE configs[0]): no picture depends on another, and one chain keeps a few dozen compute units busy for 0.7 ms --
  // the pictures ALTERNATE between the main stream with its arrays and a second stream with the side chain's (round 5: all-intra 1080p was bound by
  // one chain after the other, 1 / (0.67 + 0.03 ms)).  The second stream is one of its own here: the input stream carries every picture's analysis.
  all_intra_alt_ = depth_ >= 2 && cfg.vaq == 0 && cfg.band_rows == 0 && cfg.intra_period == 1 && !getenv("KVAZZUP_AMD_IDR_INLINE");
  if (idr_side_ || all_intra_alt_) {
    const size_t nsync = ((size_t)rows_ * (cw_ / 64) * 3 + 2 + 3) & ~(size_t)3, nctu_ = (size_t)(cw_ / 64) * rows_;
    HIP_OK(hipMalloc(&sync_idr_, sizeof(uint32_t) * nsync)); HIP_OK(hipMemset(sync_idr_, 0, sizeof(uint32_t) * nsync));
    HIP_OK(hipMalloc(&edge_col_idr_, nctu_ * 128 * sizeof(uint32_t))); HIP_OK(hipMemset(edge_col_idr_, 0, nctu_ * 128 * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&edge_row_idr_, nctu_ * 32 * 8)); HIP_OK(hipMemset(edge_row_idr_, 0, nctu_ * 32 * 8));
    if (cfg.sao) { const size_t npx_ = (size_t)cw_ * ch_; for (int c = 0; c < 3; c++) HIP_OK(hipMalloc(&work_idr_[c], c ? npx_ / 4 : npx_)); }
    // ... which is the INPUT stream: the pictures behind an intra picture need it anyway, so their input stages lose nothing by queueing behind
    // its chain, and a further stream would share a hardware queue with one that matters (HIP spreads a priority level's streams over four;
    // measured with a stream of its own: no gain at the default level, half the rate at any other -- KVAZZUP_AMD_IDR_PRIO)
    const char *lv = getenv("KVAZZUP_AMD_IDR_PRIO");
    if (lv || all_intra_alt_) HIP_OK(stream_acquire(&stream_idr_, cfg.device, 'X', lv ? lv[0] : prio_[0])); else stream_idr_ = stream_in_;
    HIP_OK(hipEventCreateWithFlags(&ev_idr_done_, kDeviceEvent));
  }
  // intra scratch: ic8 (nb8 u32) | ic16 (nb8/4 u32) | ic32 (nb8/16 u32) | im8 | im16 | im32
  size_t isz = nb8 * 4 + nb8 + nb8 / 4 + nb8 + nb8 / 4 + nb8 / 16 + 64;
  HIP_OK(hipMalloc(&intra_scratch_, isz));
  const int nctu = (cw_ / 64) * rows_;
  tok_cap_ = 49152;                                // tokens per CTU slot (worst case of a 64x64 CTU is ~43k)
  HIP_OK(hipMalloc(&tok_buf_, (size_t)nctu * tok_cap_ * sizeof(uint16_t)));
  HIP_OK(hipMalloc(&tok_count_, sizeof(uint32_t) * nctu * 2));        // two sets of cursors take turns (k_tok_compact)
  HIP_OK(hipMalloc(&tok_seg_, sizeof(uint32_t) * nctu * 16 * 17 * 2));       // [ctu][unit][piece] {offset, length}
  HIP_OK(hipMemset(tok_seg_, 0, sizeof(uint32_t) * nctu * 16 * 17 * 2));     // (all "no tokens": the list form of k_tokenize writes only the entries that have some; k_tok_compact zeroes behind itself)
  if (cfg.band_rows == 0) { HIP_OK(hipMalloc(&tok_list_, sizeof(uint32_t) * (1 + (size_t)nctu * 16 * 4))); HIP_OK(hipMemset(tok_list_, 0, sizeof(uint32_t) * (1 + (size_t)nctu * 16 * 4))); }      // (unit, role) pairs with something to say, P pictures (hevc_core.h EncFrame::tok_list)
  HIP_OK(hipMemset(tok_count_, 0, sizeof(uint32_t) * nctu * 2)); tok_nctu_ = nctu;
  tok_dense_cap_ = (size_t)nctu * tok_cap_;
  if (tok_dense_cap_ > ((size_t)1 << 27)) tok_dense_cap_ = (size_t)1 << 27;
  spin_wait_ = getenv("KVAZZUP_AMD_SPIN") != nullptr;
  nslots_ = depth_ + 1;
  stage_cap_ = tok_dense_cap_ * 2 + 256 * (size_t)rows_; if (stage_cap_ > 0xfff00000u) stage_cap_ = 0xfff00000u;
  out_cap_ = (size_t)cw_ * ch_ * 3 + 4096; if (out_cap_ > stage_cap_) out_cap_ = stage_cap_;       // twice the raw picture: a larger access unit is not a picture this encoder makes
  for (int i = 0; i < nslots_; i++) {
    Slot &sl = slot_[i];
    void *dp = nullptr;
    if (cfg.entropy_gpu) {
      HIP_OK(hipMalloc(&sl.g_tok, tok_dense_cap_ * sizeof(uint16_t)));
      HIP_OK(hipMalloc(&sl.g_count, sizeof(int32_t) * nctu)); HIP_OK(hipMalloc(&sl.g_off, sizeof(uint32_t) * nctu));
      sl.d_tok_dense = sl.g_tok; sl.d_tok_count = sl.g_count; sl.d_tok_off = sl.g_off;
      HIP_OK(hipMalloc(&sl.g_stage, stage_cap_));
      HIP_OK(hipMalloc(&sl.g_cursors, 2 * sizeof(uint32_t))); HIP_OK(hipMemset(sl.g_cursors, 0, 2 * sizeof(uint32_t)));
      HIP_OK(hipMalloc(&sl.g_ctx_save, sizeof(uint32_t) * 40 * rows_));
      HIP_OK(hipMalloc(&sl.g_ctx_ready, sizeof(uint32_t) * rows_)); HIP_OK(hipMemset(sl.g_ctx_ready, 0, sizeof(uint32_t) * rows_));
      HIP_OK(hipHostMalloc(&sl.h_out, out_cap_, hipHostMallocMapped));
      HIP_OK(hipHostGetDevicePointer(&dp, sl.h_out, 0)); sl.d_out = (uint8_t *)dp;
      HIP_OK(hipHostMalloc(&sl.h_sub, sizeof(uint32_t) * 3 * rows_, hipHostMallocMapped));
      HIP_OK(hipHostGetDevicePointer(&dp, sl.h_sub, 0)); sl.d_sub = (uint32_t *)dp;
      HIP_OK(stream_acquire(&sl.ent_stream, cfg.device, 'E', 'l'));           // (streams of one priority level share that level's hardware queues: the long coder kernels get the lowest level to themselves)
      HIP_OK(hipEventCreateWithFlags(&sl.tok_ev, hipEventDisableTiming));
    } else {
      HIP_OK(hipHostMalloc(&sl.h_tok_dense, tok_dense_cap_ * sizeof(uint16_t), hipHostMallocMapped));
      HIP_OK(hipHostMalloc(&sl.h_tok_count, sizeof(int32_t) * nctu, hipHostMallocMapped));
      HIP_OK(hipHostGetDevicePointer(&dp, sl.h_tok_dense, 0)); sl.d_tok_dense = (uint16_t *)dp;
      HIP_OK(hipHostGetDevicePointer(&dp, sl.h_tok_count, 0)); sl.d_tok_count = (int32_t *)dp;
      HIP_OK(hipHostMalloc(&sl.h_tok_off, sizeof(uint32_t) * nctu, hipHostMallocMapped));
      HIP_OK(hipHostGetDevicePointer(&dp, sl.h_tok_off, 0)); sl.d_tok_off = (uint32_t *)dp;
    }
    HIP_OK(hipHostMalloc(&sl.h_err, sizeof(uint32_t), hipHostMallocMapped)); *sl.h_err = 0;
    HIP_OK(hipHostGetDevicePointer(&dp, sl.h_err, 0)); sl.d_err = (uint32_t *)dp;
    HIP_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&sl.rec_done, hipEventDisableTiming));
  }
  HIP_OK(hipEventCreateWithFlags(&in_done_, kDeviceEvent));
  HIP_OK(hipMalloc(&sync_, sizeof(uint32_t) * ((rows_ * (cw_ / 64) * 3 + 2 + 3) & ~3))); HIP_OK(hipMemset(sync_, 0, sizeof(uint32_t) * ((rows_ * (cw_ / 64) * 3 + 2 + 3) & ~3)));      // (a multiple of 16 bytes: k_picture_begin zeroes it in 16-byte pieces)       // one progress counter per CTU and colour plane, and the ticket counter of k_intra_recon's workgroups
  if (cfg.intra_in_p) { const size_t n16 = (size_t)(cw_ / 16) * (ch_ / 16); HIP_OK(hipMalloc(&me_cost16_, sizeof(uint32_t) * (n16 + 2 + n16 / 4 + n16 / 4 + (n16 / 4) * 40))); HIP_OK(hipMemset(me_cost16_, 0, sizeof(uint32_t) * (n16 + 2 + n16 / 4 + n16 / 4 + (n16 / 4) * 40))); }      // k_me's inter cost per 16x16 block (intra-in-P)
  me_ahead_ = cfg.me_source != 0;
  if (me_ahead_ && cfg.intra_in_p) {                       // (encoder.h me_block_: one block of these and one array of progress counters per working set)
    const size_t n16 = (size_t)(cw_ / 16) * (ch_ / 16), nme = sizeof(uint32_t) * (n16 + 2 + n16 / 4 + n16 / 4 + (n16 / 4) * 40), nsy = sizeof(uint32_t) * ((rows_ * (cw_ / 64) * 3 + 2 + 3) & ~3);
    for (int k = 0; k < kSets; k++) { HIP_OK(hipMalloc(&me_block_[k], nme)); HIP_OK(hipMemset(me_block_[k], 0, nme)); HIP_OK(hipMalloc(&sync_set_[k], nsy)); HIP_OK(hipMemset(sync_set_[k], 0, nsy)); }
  }
  {
    // dispatch order of the intra reconstruction's workgroups: the CTUs of the rows this instance codes, by anti-diagonal cx + 2 cy
    const int wc = cw_ / 64, r0 = cfg.band_rows > 0 ? cfg.band_row0 : 0, nr = cfg.band_rows > 0 ? cfg.band_rows : rows_;
    std::vector<uint32_t> order;
    const int slope = getenv("KVAZZUP_AMD_INTRA_SLOPE") ? atoi(getenv("KVAZZUP_AMD_INTRA_SLOPE")) : 2;
    for (int d = 0; d < wc + slope * nr; d++) for (int cy = 0; cy < nr; cy++) { const int cx = d - slope * cy; if (cx >= 0 && cx < wc) order.push_back((uint32_t)((r0 + cy) * wc + cx)); }
    HIP_OK(hipMalloc(&intra_order_, sizeof(uint32_t) * order.size()));
    HIP_OK(hipMemcpy(intra_order_, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice));
  }
  { const size_t nctu = (size_t)(cw_ / 64) * rows_; HIP_OK(hipMalloc(&edge_col_, nctu * 128 * sizeof(uint32_t))); HIP_OK(hipMemset(edge_col_, 0, nctu * 128 * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&edge_row_, nctu * 32 * 8)); HIP_OK(hipMemset(edge_row_, 0, nctu * 32 * 8)); }      // (tagged words: generation 0 = never written)      // the CTUs' right columns (k_intra_recon)
  HIP_OK(hipMalloc(&err_, sizeof(uint32_t))); HIP_OK(hipMemset(err_, 0, sizeof(uint32_t)));
  if (getenv("KVAZZUP_AMD_INTRA_TRACE")) { HIP_OK(hipMalloc(&trace_, sizeof(unsigned long long) * (rows_ * (cw_ / 64) * 72))); HIP_OK(hipMemset(trace_, 0, sizeof(unsigned long long) * (rows_ * (cw_ / 64) * 72))); }
  int eth = cfg.entropy_threads;
  if (const char *e = getenv("KVAZZUP_AMD_ENTROPY_THREADS")) eth = atoi(e) < 1 ? 1 : atoi(e);     // tuning knob (containers with a small CPU quota)
  if (cfg.entropy_gpu) { entropy_ = nullptr; entropy2_ = nullptr; }
  else if (cfg.owf >= 2 && eth >= 4) {                        // two pictures side by side, half the threads each
    entropy_ = new EntropyHost((eth + 1) / 2 < rows_ ? (eth + 1) / 2 : rows_);
    entropy2_ = new EntropyHost(eth / 2 < rows_ ? eth / 2 : rows_);
  } else entropy_ = new EntropyHost(eth < rows_ ? eth : rows_);

  memset(&f_, 0, sizeof(f_));
  f_.cw = cw_; f_.ch = ch_; f_.b8w = cw_ / 8; f_.b8h = ch_ / 8;
  f_.tile_rows = cfg.tile_rows; f_.tile_cols = cfg.tile_cols; f_.chp = pack_height(ch_, cfg.tile_rows, cfg.tile_cols);
  f_.row0 = cfg.band_rows > 0 ? cfg.band_row0 : 0; f_.nrows = cfg.band_rows > 0 ? cfg.band_rows : 0;
  f_.qp = cfg.qp; f_.qpc = kChromaQp[cfg.qp]; f_.lambda_q4 = kLambdaQ4[cfg.qp]; f_.range = cfg.me_range;
  if (cfg.scaling_list) {                                 // `scaling-list default`: the default lists' factors, once
    uint8_t tab[KVZ_SCALING_BYTES];
    scaling_factors(scaling_defaults(), tab);
    HIP_OK(hipMalloc(&d_scaling_, KVZ_SCALING_BYTES)); HIP_OK(hipMemcpy(d_scaling_, tab, KVZ_SCALING_BYTES, hipMemcpyHostToDevice));
  }
  f_.scaling = d_scaling_; f_.intra_chain = cfg.intra_chain;
  f_.lossless = cfg.lossless; f_.rdoq = cfg.rdoq; f_.signhide = cfg.signhide; f_.intra_p = cfg.intra_in_p; f_.me_cost16 = me_cost16_; f_.me_cand = me_cost16_ ? me_cost16_ + (size_t)(cw_ / 16) * (ch_ / 16) : nullptr;
  if (me_cost16_) { const size_t n16 = (size_t)(cw_ / 16) * (ch_ / 16); f_.ip_arrive = f_.me_cand + 1 + n16 / 4; f_.ip_scratch = (uint64_t *)(me_cost16_ + ((n16 + 1 + n16 / 4 + n16 / 4 + 1) & ~(size_t)1)); }      // (8-byte aligned)
  f_.wpp = cfg.wpp; f_.mv_frame = cfg.mv_frame; f_.me_early = cfg.me_early; f_.satd = cfg.satd; f_.subme = cfg.subme; f_.slices = cfg.slices;
  bind_set(0);
  uint8_t *p = intra_scratch_;
  f_.ic8 = (uint32_t *)p; p += nb8 * 4; f_.ic16 = (uint32_t *)p; p += nb8; f_.ic32 = (uint32_t *)p; p += nb8 / 4;
  f_.im8 = p; p += nb8; f_.im16 = p; p += nb8 / 4; f_.im32 = p;
  f_.tok_buf = tok_buf_; f_.tok_cap = tok_cap_; f_.tok_cursor = (uint32_t *)tok_count_; f_.tok_cursor_next = (uint32_t *)tok_count_ + tok_nctu_; f_.tok_seg = tok_seg_; f_.tok_list = tok_list_;
  f_.tok_dense_cap = (uint32_t)tok_dense_cap_;
  { const size_t nctu = (size_t)(cw_ / 64) * rows_; f_.edge_col[0] = edge_col_; f_.edge_col[1] = edge_col_ + nctu * 64; f_.edge_col[2] = edge_col_ + nctu * 96;
    f_.edge_row[0] = edge_row_; f_.edge_row[1] = edge_row_ + nctu * 16; f_.edge_row[2] = edge_row_ + nctu * 24; }
  f_.sync = sync_; f_.err = err_; f_.trace = trace_; f_.intra_order = intra_order_;

  sp_.cw = cw_; sp_.ch = ch_; sp_.width = cfg.width; sp_.height = cfg.height; sp_.qp = cfg.qp; sp_.wpp = cfg.wpp; sp_.tile_rows = cfg.tile_rows; sp_.tile_cols = cfg.tile_cols; sp_.qp_in_cu = cfg.qp_in_cu; sp_.sao = cfg.sao; sp_.slices = cfg.slices; sp_.signhide = cfg.signhide; sp_.scaling_list = cfg.scaling_list; sp_.tq_bypass = cfg.lossless;
  sp_.deblock = cfg.deblock; sp_.fps_num = cfg.fps_num; sp_.fps_den = cfg.fps_den;
  HIP_OK(hipStreamSynchronize(stream_));
  HIP_OK(hipDeviceSynchronize());
  tok_deferred_ = depth_ >= 2 && cfg.sao && !cfg.entropy_gpu && cfg.band_rows == 0 && cfg.owf <= kSets - 1 && !getenv("KVAZZUP_AMD_TOK_INLINE");
  HIP_OK(hipDeviceSynchronize());                        // (every clear above ran on the null stream: done before anything is queued on the encoder's non-blocking streams)
  if (tok_deferred_) tok_thread_ = std::thread([this] { name_this_thread("kvzx-enc-tok"); tok_launcher(); });
  if (depth_ >= 2) { bg_[0] = std::thread([this] { name_this_thread("kvzx-enc-bg0"); background(0); }); if (entropy2_) bg_[1] = std::thread([this] { name_this_thread("kvzx-enc-bg1"); background(1); }); }
  if (depth_ >= 2 && cfg.band_rows == 0 && !getenv("KVAZZUP_AMD_SYNC_SUBMIT")) sub_thread_ = std::thread([this] { name_this_thread("kvzx-enc-sub"); submitter(); });
  return true;
}

namespace { struct Tick { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } }; }

void Encoder::bind_set(int k)
{
  const size_t nb8 = (size_t)cw_ * ch_ / 64;
  uint8_t *cu = cu_bytes_[k];
  for (int c = 0; c < 3; c++) { f_.coef[c] = coef_[k][c]; f_.src[c] = src_[k][c]; }
  f_.cu_log2 = cu; f_.cu_intra = cu + nb8; f_.cu_flags = cu + 2 * nb8; f_.cu_merge_idx = cu + 3 * nb8;
  f_.cu_mvp_idx = cu + 4 * nb8; f_.cu_intra_mode = cu + 5 * nb8; f_.cu_cbf = cu + 6 * nb8;
  f_.cu_mv = cu_mv_[k]; f_.cu_mvd = cu_mvd_[k];
  f_.sao = sao_[k];                                    // NULL without SAO
  f_.ctu_qt = ctu_qt_[k]; f_.ctu_qy = ctu_qy_[k]; f_.ctu_delta = ctu_delta_[k]; f_.ctu_first = ctu_first_[k];     // all NULL without qp_in_cu
}

Encoder::~Encoder()
{
  if (probe_words_) {
    uint32_t w[2] = {0, 0};
    hipDeviceSynchronize(); hipMemcpy(w, probe_words_, sizeof(w), hipMemcpyDeviceToHost);
    fprintf(stderr, "kvazzup_amd parse probe: %u bins decoded on the GPU, %u differ from the token list\n", w[1], w[0]);
    hipFree(probe_words_);
  }
  if (getenv("KVAZZUP_AMD_TRACE")) fprintf(stderr, "kvazzup_amd encoder thread ms: submit %.1f  wait_gpu %.1f  arith %.1f  assemble %.1f  wait_input %.1f  (pictures %ld)\n", t_submit_, t_wait_, t_arith_, t_asm_, t_in_, collected_);
  { std::lock_guard<std::mutex> l(sm_); squit_ = true; }
  scv_.notify_all();
  if (sub_thread_.joinable()) sub_thread_.join();
  { std::lock_guard<std::mutex> l(tm_); tquit_ = true; }
  tcv_.notify_all();
  if (tok_thread_.joinable()) tok_thread_.join();           // (what it still had queued has gone to the workers)
  { std::lock_guard<std::mutex> l(bm_); bquit_ = true; }
  bcv_.notify_all();
  for (auto &t : bg_) if (t.joinable()) t.join();
  if (stream_) hipStreamSynchronize(stream_);
  if (stream_idr_ && stream_idr_ != stream_in_) hipStreamSynchronize(stream_idr_);
  if (stream_tok_) hipStreamSynchronize(stream_tok_);
  if (stream_in_) hipStreamSynchronize(stream_in_);
  for (Slot &sl : slot_) {
    for (auto &e : sl.ev) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    if (sl.sink_done) { hipEventSynchronize(sl.sink_done); hipEventDestroy(sl.sink_done); }
    if (sl.h_tok_dense) hipHostFree(sl.h_tok_dense);
    if (sl.h_tok_count) hipHostFree(sl.h_tok_count);
    if (sl.h_err) hipHostFree(sl.h_err);
    if (sl.h_tok_off) hipHostFree(sl.h_tok_off);
    stream_release(sl.ent_stream, cfg_.device, 'E', 'l');
    if (sl.tok_ev) hipEventDestroy(sl.tok_ev);
    hipFree(sl.g_tok); hipFree(sl.g_count); hipFree(sl.g_off); hipFree(sl.g_stage); hipFree(sl.g_cursors); hipFree(sl.g_ctx_save); hipFree(sl.g_ctx_ready);
    if (sl.h_out) hipHostFree(sl.h_out);
    if (sl.h_sub) hipHostFree(sl.h_sub);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.rec_done) hipEventDestroy(sl.rec_done);
  }
  if (in_done_) hipEventDestroy(in_done_);
  if (stream_h2d_ && stream_h2d_ != stream_in_) stream_release(stream_h2d_, cfg_.device, 'H', 'l');
  for (int k = 0; k < kInRing; k++) { hipFree(d_in_[k]); if (h_in_[k]) hipHostFree(h_in_[k]); if (ev_h2d_[k]) hipEventDestroy(ev_h2d_[k]); if (ev_pad_[k]) hipEventDestroy(ev_pad_[k]); }
  stream_release(stream_rec_, cfg_.device, 'R', 'n');
  for (int c = 0; c < 3; c++) { for (int k = 0; k < kSets; k++) { hipFree(src_[k][c]); hipFree(coef_[k][c]); } for (int b = 0; b < kMaxDepth + 4; b++) hipFree(rec_[b][c]); }
  hipFree(vaq_act_); hipFree(vaq_sum_); hipFree(rc_state_);
  for (int k = 0; k < kSets; k++) { hipFree(ctu_qt_[k]); hipFree(ctu_qy_[k]); hipFree(ctu_delta_[k]); hipFree(ctu_first_[k]); if (h_ctu_qt_[k]) hipHostFree(h_ctu_qt_[k]); hipFree(ctu_roi_[k]); }
  for (int k = 0; k < kSets; k++) { hipFree(cu_bytes_[k]); hipFree(cu_mv_[k]); hipFree(cu_mvd_[k]); if (ev_tok_done_[k]) hipEventDestroy(ev_tok_done_[k]); }
  for (int c = 0; c < 3; c++) { hipFree(work_[c]); hipFree(work_idr_[c]); }
  hipFree(sync_idr_); hipFree(edge_col_idr_); hipFree(edge_row_); hipFree(edge_row_idr_);
  for (int k = 0; k < kSets; k++) hipFree(sao_[k]);
  if (ev_signalled_) hipEventDestroy(ev_signalled_);
  for (int k = 0; k < kSets; k++) if (ev_src_free_[k]) hipEventDestroy(ev_src_free_[k]);
  stream_release(stream_tok_, cfg_.device, 'T', prio_[1]);
  stream_release(stream_in_, cfg_.device, 'I', prio_[2]);
  if (stream_idr_ && stream_idr_ != stream_in_) { const char *lv = getenv("KVAZZUP_AMD_IDR_PRIO"); stream_release(stream_idr_, cfg_.device, 'X', lv ? lv[0] : (all_intra_alt_ ? prio_[0] : 'n')); }
  if (ev_idr_done_) hipEventDestroy(ev_idr_done_);
  hipFree(intra_scratch_); hipFree(d_scaling_);
  delete entropy_; delete entropy2_;
  for (int k = 0; k < kSets; k++) { hipFree(me_block_[k]); hipFree(sync_set_[k]); }
  hipFree(trace_); hipFree(intra_order_); hipFree(tok_buf_); hipFree(tok_count_); hipFree(tok_seg_); hipFree(tok_list_); hipFree(sync_); hipFree(me_cost16_); hipFree(edge_col_); hipFree(err_);
  stream_release(stream_, cfg_.device, 'M', prio_[0]);
}

void Encoder::timed(KernelId id, hipStream_t st, const std::function<void()> &launch) { timed_slot(*cur_slot_, prof_now_, id, st, launch); }
void Encoder::timed_slot(Slot &sl, bool prof, KernelId id, hipStream_t st, const std::function<void()> &launch)
{
  if (!prof) { launch(); return; }
  if (sl.ev_used == sl.ev.size()) {
    EvPair p; hipEventCreate(&p.a); hipEventCreate(&p.b); p.id = id; sl.ev.push_back(p);
  }
  EvPair &p = sl.ev[sl.ev_used++]; p.id = id;
  hipEventRecord(p.a, st);
  launch();
  hipEventRecord(p.b, st);
}

void Encoder::get_kernel_times(double *ms, uint64_t *launches, bool reset)
{
  std::lock_guard<std::mutex> l(stat_m_);
  for (int i = 0; i < K_COUNT; i++) { if (ms) ms[i] = k_ms_[i]; if (launches) launches[i] = k_n_[i]; }
  if (reset) for (int i = 0; i < K_COUNT; i++) { k_ms_[i] = 0; k_n_[i] = 0; }
}

// encode = submit the picture's kernels, then finish ("collect") the oldest picture in flight.  With
// owf == 0 that is the picture just submitted; with owf >= 1 it is the previous one, whose arithmetic
// coding on the host then runs while the GPU works on the new picture.
// Host picture in: the copy engine moves it into device buffer k = t mod kInRing on the upload stream while earlier pictures' kernels
// run; the input stage of picture t (stream_in_) waits for that copy.  The ring is longer than the pictures that can be in flight, so the
// buffer's previous reader (the input stage of picture t - kInRing) has long finished and the copy command carries no dependency: the copy
// engine takes it at once.  (A copy waiting inside the engine's queue holds up every copy behind it, the decoder's included; the input
// kernel reading the host picture itself across PCIe -- tried -- slows the kernels running beside it 2-10x, tools/measure/pcie_copy_vs_kernels.hip.)
// Nothing here waits on the calling thread except a staging buffer coming free (callers without page-locked planes).
bool Encoder::upload_and_submit(const uint8_t *y, const uint8_t *u, const uint8_t *v, bool pinned)
{
  const size_t ny = (size_t)cfg_.width * cfg_.height, bytes = ny * 3 / 2;
  const int k = (int)(in_count_++ % kInRing);
  // The copy rides on the input stream itself, ahead of the picture's input kernel.  HIP spreads the streams of one priority level over four
  // hardware queues, and at the default level those are taken (tokenizer, input, decoder, decoder transfers): a fifth stream there shares a
  // queue with one of them and is serialised behind that stream's event waits (measured: the input stream behind the tokenizer's, 3700 instead
  // of 5200 frames/s at 1080p); a stream at the lowest level (KVAZZUP_AMD_H2D=own) has a queue to itself and measured no better for the
  // encoder alone, worse with the decoder beside it.
  static const bool h2d_on_in = [] { const char *e = getenv("KVAZZUP_AMD_H2D"); return !e || strcmp(e, "own"); }();
  if (!stream_h2d_) { if (h2d_on_in) stream_h2d_ = stream_in_; else HIP_CHECK(stream_acquire(&stream_h2d_, cfg_.device, 'H', 'l')); }
  if (!d_in_[k]) {
    HIP_CHECK(hipMalloc(&d_in_[k], bytes));
    HIP_CHECK(hipEventCreateWithFlags(&ev_h2d_[k], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ev_pad_[k], hipEventDisableTiming));
  }
  const uint8_t *src = y;
  if (!pinned) {
    if (!h_in_[k]) HIP_CHECK(hipHostMalloc(&h_in_[k], bytes, hipHostMallocDefault));
    if (h2d_pending_[k]) { Tick tk; HIP_CHECK(hipEventSynchronize(ev_h2d_[k])); h2d_pending_[k] = false; t_in_ += tk.ms(); }   // the staging buffer's last upload
    memcpy(h_in_[k], y, ny); memcpy(h_in_[k] + ny, u, ny / 4); memcpy(h_in_[k] + ny + ny / 4, v, ny / 4);
    src = h_in_[k];
  }
  if (pad_pending_[k]) {                             // d_in_[k]'s last reader: done long ago, normally
    if (hipEventQuery(ev_pad_[k]) != hipSuccess) HIP_CHECK(hipStreamWaitEvent(stream_h2d_, ev_pad_[k], 0));
    pad_pending_[k] = false;
  }
  HIP_CHECK(hipMemcpyAsync(d_in_[k], src, bytes, hipMemcpyHostToDevice, stream_h2d_));
  HIP_CHECK(hipEventRecord(ev_h2d_[k], stream_h2d_)); h2d_pending_[k] = true;
  if (stream_h2d_ != stream_in_) HIP_CHECK(hipStreamWaitEvent(stream_in_, ev_h2d_[k], 0));
  { Tick tk; if (!submit(d_in_[k], k)) return false; t_submit_ += tk.ms(); }
  in_pending_ = false;                               // (nobody but this ring reads d_in_[k])
  return true;
}

bool Encoder::encode_host(const uint8_t *y, const uint8_t *u, const uint8_t *v, EncodedPicture *out, bool pinned)
{
  const size_t ny = (size_t)cfg_.width * cfg_.height;
  HIP_CHECK(hipSetDevice(cfg_.device));
  out->valid = false; out->au.clear();
  pinned = pinned && u == y + ny && v == u + ny / 4;
  if (pinned && sub_thread_.joinable()) return enqueue(y, true, out);
  drain_submitter();
  accepted_++;
  roi_sub_ = roi_; roi_sub_w_ = roi_w_; roi_sub_h_ = roi_h_;
  for (int c = 0; c < 3; c++) { sink_sub_[c] = sink_[c]; sink_[c] = nullptr; }
  tl("up0", submitted_);
  if (!upload_and_submit(y, u, v, pinned)) { accepted_--; return false; }
  tl("sub1", submitted_ - 1);
  return pending() > depth_ ? collect(out) : true;
}

bool Encoder::encode_device(const uint8_t *d_i420, EncodedPicture *out)
{
  HIP_CHECK(hipSetDevice(cfg_.device));
  out->valid = false; out->au.clear();
  if (cfg_.input_hold && sub_thread_.joinable()) return enqueue(d_i420, false, out);
  drain_submitter();
  accepted_++;
  roi_sub_ = roi_; roi_sub_w_ = roi_w_; roi_sub_h_ = roi_h_;
  { Tick tk; if (!submit(d_i420, -1)) { accepted_--; return false; } t_submit_ += tk.ms(); }
  bool ok = true;
  if (pending() > depth_) ok = collect(out);
  // the caller may reuse its input buffer when this returns (the pad kernel is first in the picture's queue,
  // and by now it has had the whole host coding stage of the previous picture to run)
  if (in_pending_) { Tick tk; HIP_CHECK(hipEventSynchronize(in_done_)); in_pending_ = false; t_in_ += tk.ms(); }
  return ok;
}

// owf >= 2: hand the picture to the submitter thread, then finish the oldest picture in flight if the pipeline is full
bool Encoder::enqueue(const uint8_t *src, bool host, EncodedPicture *out)
{
  tl("enq", accepted_);
  SubmitJob j; j.src = src; j.host = host; j.roi = roi_; j.roi_w = roi_w_; j.roi_h = roi_h_; j.slot = (int)(accepted_ % nslots_);
  for (int c = 0; c < 3; c++) { j.sink[c] = sink_[c]; sink_[c] = nullptr; }
  { std::lock_guard<std::mutex> l(bm_); slot_[j.slot].ready = false; slot_[j.slot].ok = true; }
  accepted_++;
  { std::lock_guard<std::mutex> l(sm_); sq_.push_back(std::move(j)); }
  scv_.notify_one();
  return pending() > depth_ ? collect(out) : true;
}

void Encoder::submitter()
{
  hipSetDevice(cfg_.device);
  for (;;) {
    SubmitJob j;
    {
      std::unique_lock<std::mutex> l(sm_);
      sbusy_ = false; scv_.notify_all();
      scv_.wait(l, [&] { return squit_ || !sq_.empty(); });
      if (squit_ || sq_.empty()) return;               // (closing: what is still queued reads caller pictures that may be gone -- it is dropped, nobody collects it)
      j = std::move(sq_.front()); sq_.pop_front(); sbusy_ = true;
    }
    const long before = submitted_;
    roi_sub_.swap(j.roi); roi_sub_w_ = j.roi_w; roi_sub_h_ = j.roi_h;
    for (int c = 0; c < 3; c++) sink_sub_[c] = j.sink[c];
    tl("sub0", submitted_);
    const size_t ny = (size_t)cfg_.width * cfg_.height;
    bool ok;
    if (j.host) ok = upload_and_submit(j.src, j.src + ny, j.src + ny + ny / 4, true);
    else { Tick tk; ok = submit(j.src, -1); t_submit_ += tk.ms(); in_pending_ = false; }
    tl("sub1", submitted_ - 1);
    if (!ok) {                                       // nothing was queued for this picture: its slot reports the failure
      if (submitted_ == before) submitted_++;        // (the slots are numbered by pictures accepted: a picture that failed still took its turn)
      { std::lock_guard<std::mutex> l(bm_); slot_[j.slot].ok = false; slot_[j.slot].ready = true; }
      bcv_.notify_all();
    }
  }
}

// a synchronous call after asynchronous ones: everything handed over so far has been queued on the GPU
void Encoder::drain_submitter()
{
  if (!sub_thread_.joinable()) return;
  std::unique_lock<std::mutex> l(sm_);
  scv_.wait(l, [&] { return sq_.empty() && !sbusy_; });
}

bool Encoder::flush(EncodedPicture *out)
{
  out->valid = false; out->au.clear();
  if (!pending()) return true;
  HIP_CHECK(hipSetDevice(cfg_.device));
  return collect(out);
}

bool Encoder::poll(EncodedPicture *out)
{
  out->valid = false; out->au.clear();
  if (!pending()) return true;
  Slot &sl = slot_[collected_ % nslots_];
  if (depth_ >= 2) { std::lock_guard<std::mutex> l(bm_); if (!sl.ready) return true; }
  else {
    HIP_CHECK(hipSetDevice(cfg_.device));
    if (sub_thread_.joinable()) drain_submitter();
    if (hipEventQuery(sl.done) != hipSuccess || hipEventQuery(sl.rec_done) != hipSuccess) return true;
  }
  return collect(out);
}

void Encoder::set_roi(int w, int h, const int8_t *map)
{
  roi_.clear(); roi_w_ = roi_h_ = 0;
  if (w > 0 && h > 0 && map) { roi_.assign(map, map + (size_t)w * h); roi_w_ = w; roi_h_ = h; }
}

// target QP of every CTU of the picture being submitted (set set_): host map -> pinned -> device, on stream_
// Per-CTU quantiser targets (cu_qp_delta): target = clip(picture QP + ROI delta), with VAQ the device adds its own delta.  The ROI deltas of the picture's map
// (kvz_picture.roi, spread over the CTU grid here) go up on `st` -- the INPUT stream, ahead of the event the main stream waits for anyway -- and only when
// the picture brings a map; the targets themselves are written by the kernel at the head of the picture's chain (k_picture_begin).  Nothing is copied on
// the main stream: a 510-byte copy there cost every picture of uvgComm's default mode a ~28 us bubble.
bool Encoder::stage_roi(hipStream_t st)
{
  roi_dev_ = nullptr;
  if (!cfg_.qp_in_cu || roi_sub_.empty()) return true;
  const int wc = cw_ / 64, hc = rows_;
  int8_t *h = h_ctu_qt_[set_];
  for (int cy = 0; cy < hc; cy++) for (int cx = 0; cx < wc; cx++)
    h[cy * wc + cx] = (int8_t)clip3(-12, 12, (int)roi_sub_[(size_t)(cy * roi_sub_h_ / hc) * roi_sub_w_ + (cx * roi_sub_w_ / wc)]);
  HIP_CHECK(hipMemcpyAsync(ctu_roi_[set_], h, (size_t)wc * hc, hipMemcpyHostToDevice, st));
  roi_dev_ = ctu_roi_[set_];
  return true;
}
// generation of the picture's intra chain launch (the tag of the edge-column words, kernel_common.h IntraNeighbours): 1 .. 2^24 - 1; when the counter
// wraps, both arrays go back to "never written" on the streams that use them
uint32_t Encoder::next_chain_gen()
{
  if (++chain_gen_ >= (1u << 24)) {
    const size_t bytes = (size_t)(cw_ / 64) * rows_ * 128 * sizeof(uint32_t);
    hipMemsetAsync(edge_col_, 0, bytes, stream_); hipMemsetAsync(edge_row_, 0, bytes / 2, stream_);
    if (edge_col_idr_) { hipMemsetAsync(edge_col_idr_, 0, bytes, stream_idr_); hipMemsetAsync(edge_row_idr_, 0, bytes / 2, stream_idr_); }
    chain_gen_ = 1;
  }
  return chain_gen_;
}

bool Encoder::picture_begin(hipStream_t qt_stream, EncFrame *fold, bool zero)
{
  const bool have = frame_idx_ >= rc_delay_;
  const uint32_t bits3 = have ? 8u * rc_bytes_[(frame_idx_ - rc_delay_) & 7] : 0u;
  const int slot3 = (frame_idx_ - rc_delay_) & 7, n = (cw_ / 64) * rows_;
  int8_t *qt = cfg_.qp_in_cu ? ctu_qt_[set_] : nullptr;
  if (fold && qt_stream == stream_ && cfg_.vaq == 0) {
    fold->pb_on = (rc_state_ || qt) ? 1 : 0; fold->pb_rc = rc_state_; fold->pb_qt = qt; fold->pb_roi = roi_dev_; fold->pb_bits3 = bits3; fold->pb_nctu = n;
    fold->pb_slot3 = (int8_t)slot3; fold->pb_have3 = have ? 1 : 0;
    return true;
  }
  // (an intra picture, `zero`: the launch also takes the chain's two arrays back to zero -- the ticket counter's array with the "has intra units" word of the
  // P pictures behind it, which no P picture has pending while an intra picture starts on these arrays, and the CU cbf bits)
  void *za = zero ? (void *)f_.sync : nullptr, *zb = zero ? (void *)f_.cu_cbf : nullptr;
  const size_t na = zero ? sizeof(uint32_t) * (((size_t)rows_ * (cw_ / 64) * 3 + 2 + 3) & ~(size_t)3) : 0, nb = zero ? (size_t)f_.b8w * f_.b8h : 0;
  if (qt_stream == stream_) launch_picture_begin(rc_state_, bits3, slot3, have ? 1 : 0, qt, roi_dev_, n, qp_cur_, cfg_.vaq > 0 ? 1 : 0, stream_, za, na, zb, nb);
  else {
    // an intra picture on its side stream: the rate control state is updated in PICTURE ORDER on the main stream (between the P pictures' row groups, which
    // run there), the picture's own per-CTU targets on its own stream
    launch_picture_begin(rc_state_, bits3, slot3, have ? 1 : 0, nullptr, nullptr, 0, qp_cur_, 0, stream_);
    launch_picture_begin(nullptr, 0, 0, 0, qt, roi_dev_, n, qp_cur_, 0, qt_stream, za, na, zb, nb);
  }
  if (cfg_.qp_in_cu && cfg_.vaq > 0) launch_vaq(f_, cfg_.vaq, vaq_act_, vaq_sum_, stream_);          // (f_.qp, f_.src and f_.ctu_qt of this picture are set; the source is padded: stream_ waits for in_done_)
  return true;
}

// picture-level rate control: the statement of record is rate_control() in oracle/hevc_enc.c
void Encoder::rate_control()
{
  if (cfg_.bitrate <= 0 || frame_idx_ < rc_delay_) return;
  const int64_t T = ((int64_t)cfg_.bitrate * cfg_.fps_den) / (cfg_.fps_num > 0 ? cfg_.fps_num : 1);
  const int64_t trend = (int64_t)8 * rc_bytes_[(frame_idx_ - rc_delay_) & 7] - T;
  rc_debt_ += trend;
  int step = 0;
  if (rc_debt_ > 4 * T && trend > 0) step = rc_debt_ > 16 * T ? 2 : 1;
  if (rc_debt_ < -4 * T && trend < 0) step = rc_debt_ < -16 * T ? -2 : -1;
  qp_cur_ = clip3(10, 51, qp_cur_ + step);
}

bool Encoder::submit(const uint8_t *d_i420, int in_ring)
{
  const int w = cfg_.width, h = cfg_.height;
  Slot &sl = slot_[submitted_ % nslots_];
  cur_slot_ = &sl;
  prof_now_ = profiling_ && (frame_idx_ % prof_every_) == 0;
  // Three HIP streams per picture t.
  //   stream_in_:  input padding into source set t & 1 -- runs while stream_ still works on picture t - 1.
  //   stream_:     motion search / intra decisions, reconstruction, deblocking: the chain picture t + 1 depends on.
  //   stream_tok_: merge/AMVP signalling, tokenizer, compaction, which only feed the host; they read set t & 1 of the
  //                level / CU arrays while stream_ already fills the other set for t + 1.
  set_ = (int)(submitted_ % kSets);
  bind_set(set_);
  const int period = cfg_.intra_period;
  const bool intra = (frame_idx_ == 0) || (period > 0 && (frame_idx_ % period) == 0);
  if (intra) poc_ = 0; else poc_++;
  rate_control();
  f_.qp = qp_cur_; f_.qpc = kChromaQp[qp_cur_]; f_.lambda_q4 = kLambdaQ4[qp_cur_];
  f_.is_intra = intra; f_.poc = poc_;
  for (int c = 0; c < 3; c++) { f_.rec[c] = cfg_.sao ? work_[c] : rec_[cur_idx_][c]; f_.sao_out[c] = rec_[cur_idx_][c]; f_.ref[c] = rec_[ref_idx_][c]; }
  // me-source: the search looks at the previous input picture (still in its working set: the set is not padded into again before kSets - 1 more pictures have
  // gone through the input stream, behind this picture's k_me), and what the search and the intra pricing behind it write is the set's own
  const bool ahead = me_ahead_ && !intra;
  f_.me_ref = ahead ? src_[prev_set_][0] : f_.ref[0];
  auto bind_me_block = [&](uint32_t *base, uint32_t *sy) {
    const size_t n16 = (size_t)(cw_ / 16) * (ch_ / 16);
    f_.me_cost16 = base; f_.me_cand = base ? base + n16 : nullptr; f_.sync = sy;
    if (base) { f_.ip_arrive = f_.me_cand + 1 + n16 / 4; f_.ip_scratch = (uint64_t *)(base + ((n16 + 1 + n16 / 4 + n16 / 4 + 1) & ~(size_t)1)); }
  };
  if (ahead && me_block_[set_]) bind_me_block(me_block_[set_], sync_set_[set_]);
  f_.tok_dense = sl.d_tok_dense; f_.tok_count_out = sl.d_tok_count; f_.tok_off_out = sl.d_tok_off; f_.err_out = sl.d_err; f_.ent_cursors = sl.g_cursors;
  f_.tok_cursor = (uint32_t *)tok_count_ + (size_t)(frame_idx_ & 1) * tok_nctu_; f_.tok_cursor_next = (uint32_t *)tok_count_ + (size_t)((frame_idx_ + 1) & 1) * tok_nctu_;
  // the stream this picture's chain runs on: an intra picture's own (encoder.h stream_idr_), else the main stream -- behind the last intra picture's chain
  const bool side = intra && (idr_side_ || (all_intra_alt_ && (frame_idx_ & 1)));
  const hipStream_t ms = side ? stream_idr_ : stream_;
  f_.chain_gen = next_chain_gen();
  f_.analyse_alone = depth_ < 2 ? 1 : 0;                     // (a synchronous encoder: the cap that keeps the search from holding every wave slot protects nobody -- 185 -> 142 us of an intra picture's encoding delay at 1080p)
  f_.chain_diags = all_intra_alt_ ? 2 : 0;                  // (two pictures' chains side by side: two anti-diagonals of workgroups each -- all-intra 2 030 -> 2 110 frames/s; a lone chain is 2.5 % slower with two, 670 -> 687 us: profiles/r05_intra_diags.txt)
  if (side) {                                               // beside the P pictures still on the main stream: nothing of theirs is touched
    const size_t nctu = (size_t)(cw_ / 64) * rows_;
    f_.sync = sync_idr_; f_.me_cand = nullptr;              // (me_cand NULL: the picture's deblocking kernel leaves the P pictures' candidate list and "has intra units" word alone)
    f_.edge_col[0] = edge_col_idr_; f_.edge_col[1] = edge_col_idr_ + nctu * 64; f_.edge_col[2] = edge_col_idr_ + nctu * 96;
    f_.edge_row[0] = edge_row_idr_; f_.edge_row[1] = edge_row_idr_ + nctu * 16; f_.edge_row[2] = edge_row_idr_ + nctu * 24;
    if (cfg_.sao) for (int c = 0; c < 3; c++) f_.rec[c] = work_idr_[c];
  }
  const EncFrame f = f_;
  if (ahead && me_block_[set_]) bind_me_block(me_cost16_, sync_);      // (f_ goes back to the shared arrays: intra pictures, band mode)
  if (side) {                                               // (f_ goes back to the shared arrays for the pictures that follow)
    const size_t nctu = (size_t)(cw_ / 64) * rows_;
    f_.sync = sync_; f_.me_cand = me_cost16_ ? me_cost16_ + (size_t)(cw_ / 16) * (ch_ / 16) : nullptr;
    f_.edge_col[0] = edge_col_; f_.edge_col[1] = edge_col_ + nctu * 64; f_.edge_col[2] = edge_col_ + nctu * 96;
    f_.edge_row[0] = edge_row_; f_.edge_row[1] = edge_row_ + nctu * 16; f_.edge_row[2] = edge_row_ + nctu * 24;
  }
  if (all_intra_alt_) idr_pending_ = false;                 // (alternating intra pictures: the main stream's next picture has nothing to do with the side stream's last)
  if (!side && idr_pending_) { HIP_CHECK(hipStreamWaitEvent(stream_, ev_idr_done_, 0)); idr_pending_ = false; }
  if (src_busy_[set_]) { HIP_CHECK(hipStreamWaitEvent(stream_in_, ev_src_free_[set_], 0)); src_busy_[set_] = false; }   // the last picture that used this set (t - kSets) has been reconstructed
  timed(K_PAD, stream_in_, [&] { launch_pad_input(d_i420, w, h, src_[set_][0], src_[set_][1], src_[set_][2], cw_, ch_, stream_in_); });
  if (in_ring >= 0) { HIP_CHECK(hipEventRecord(ev_pad_[in_ring], stream_in_)); pad_pending_[in_ring] = true; }
  // The tokenizer of the set's previous picture must be done with the set's CU arrays before this picture writes them: the INPUT stream waits for it (the
  // event is long past when it gets there), so that the main stream's one wait for in_done_ says both -- a wait of its own in front of every picture's chain
  // cost the chain a barrier packet, a few microseconds with nothing running.
  if (tok_pending_[set_]) { HIP_CHECK(hipStreamWaitEvent(stream_in_, ev_tok_done_[set_], 0)); tok_pending_[set_] = false; }
  if (intra) {
    // The intra decisions need the source picture only: they run on the input stream, beside what is left of picture t - 1 on the main stream.
    timed(K_INTRA_ANALYSE, stream_in_, [&] { launch_intra_analyse(f, stream_in_); });
  }
  if (ahead) {
    // "uvgx search pipelining v1": the search needs the two input pictures only, the pricing of its expensive quarters as intra blocks the search and the source --
    // both run HERE, on the input stream, beside what the main stream still has of the pictures in front (k_me 31-34 us and k_intra_analyse<P> 19-24 us at
    // 1080p leave the chain the next picture waits for; the head of the chain, when there is one, is a launch of its own on the main stream)
    timed(K_ME, stream_in_, [&] { launch_me(f, stream_in_); });
    if (cfg_.intra_in_p) timed(K_INTRA_ANALYSE_P, stream_in_, [&] { launch_intra_analyse(f, stream_in_); });
  }
  if (!stage_roi(stream_in_)) return false;
  HIP_CHECK(hipEventRecord(in_done_, stream_in_)); in_pending_ = true;
  HIP_CHECK(hipStreamWaitEvent(ms, in_done_, 0));      // (measured by leaving it out: 8 of the ~28 us between a picture's last kernel and the next one's first; the rest is the record behind k_sao that two other streams wait for)
  EncFrame fm = f;                                               // (the picture's first kernel may carry the head of the chain)
  if (intra) { const uint32_t *keep = f_.sync; f_.sync = f.sync; const bool ok = picture_begin(ms, nullptr, true); f_.sync = const_cast<uint32_t *>(keep); if (!ok) return false; }      // (f.sync: the side stream's array when the picture runs there; f_ is back on the shared one)
  else if (!picture_begin(ms, ahead && cfg_.subme == 0 ? nullptr : &fm)) return false;      // (the search ahead on the input stream: the chain's head rides in k_subpel, or is a launch of its own without one)
  if (intra) {
    timed(K_INTRA_RECON, ms, [&] { launch_intra_recon(f, ms); });
  } else {
    if (!ahead) timed(K_ME, stream_, [&] { launch_me(fm, stream_); });
    // intra-in-P: quarters whose inter cost is high are priced as intra blocks and may become intra units (the launch leaves at once where none is)
    if (cfg_.intra_in_p && !ahead) timed(K_INTRA_ANALYSE_P, stream_, [&] { launch_intra_analyse(f, stream_); });
    if (cfg_.subme > 0) timed(K_SUBPEL, stream_, [&] { launch_subpel(ahead ? fm : f, stream_); });
    if (rc_state_) {
      // rate control v2: the CTU rows in groups inside the one launch, the next group's QP decided on the device from the levels of the groups before
      EncFrame fb = f;
      fb.rc = rc_state_; fb.rc_nb = cfg_.rc_bands < rows_ ? cfg_.rc_bands : rows_; fb.rc_slot = frame_idx_ & 7;
      fb.rc_target = ((long long)cfg_.bitrate * cfg_.fps_den) / (cfg_.fps_num > 0 ? cfg_.fps_num : 1);
      timed(K_INTER_RECON, stream_, [&] { launch_inter_recon(fb, stream_); });
    } else
    timed(K_INTER_RECON, stream_, [&] { launch_inter_recon(f, stream_); });
    // ... and are reconstructed behind every inter unit (their reference samples may lie in inter units anywhere around them)
    if (cfg_.intra_in_p) timed(K_INTRA_RECON_P, stream_, [&] { launch_intra_recon(f, stream_); });
    if (cfg_.intra_in_p && !cfg_.deblock) { HIP_CHECK(hipMemsetAsync(f.me_cand, 0, sizeof(uint32_t), stream_)); HIP_CHECK(hipMemsetAsync(f.sync + rows_ * (cw_ / 64) * 3 + 1, 0, sizeof(uint32_t), stream_)); }      // (k_deblock_tile does it otherwise)
  }
  launch_qp_resolve(f, ms);                                      // per-CTU QP: which CU carries the delta, QpY for deblocking
  // levels, cbf and motion of the picture are final: the tokenizer's stream may start.  With SAO it waits for the filter anyway (the CTUs' SAO parameters
  // are coded) -- then no event is recorded in the middle of the chain (an event between two kernels of a stream costs the chain ~7 us)
  if (!cfg_.sao) HIP_CHECK(hipEventRecord(ev_signalled_, ms));
  if (cfg_.deblock) timed(K_DEBLOCK, ms, [&] { launch_deblock(f, ms); });
  if (cfg_.sao) timed(K_SAO, ms, [&] { launch_sao(f, ms); });      // (ONE event behind the chain's last kernel says "SAO done" and "the set is free": every record in the chain is a packet the next picture waits behind)
  // Last reader of this set on the main stream: k_sao reads the source picture for its statistics, deblocking the CU records.
  // Input padding and intra analysis of the next picture with this set (input stream) overwrite both and wait for this event.
  HIP_CHECK(hipEventRecord(ev_src_free_[set_], ms)); src_busy_[set_] = true;
  if (side) { HIP_CHECK(hipEventRecord(ev_idr_done_, ms)); idr_pending_ = true; }
  sl.set = set_;
  if (tok_deferred_) { sl.f_tok = f; sl.prof = prof_now_; }          // (encoder.h tok_deferred_: the launcher thread makes these launches when the chain is done)
  else if (!launch_tokenizer(sl, f, intra, true, prof_now_)) return false;
  tok_pending_[set_] = true;
  // the slot is complete when both streams are: the tokens (stream_tok_) and the reconstruction (stream_)
  if (tok_deferred_) { }
  else if (cfg_.entropy_gpu) {
    // arithmetic coding on the slot's own stream, behind the compaction: the coders of several pictures run side by side
    HIP_CHECK(hipEventRecord(sl.tok_ev, stream_tok_));
    HIP_CHECK(hipStreamWaitEvent(sl.ent_stream, sl.tok_ev, 0));
    const int nsub = cfg_.wpp ? rows_ : cfg_.tile_rows;
    CabacRowsArgs a;
    a.tok = sl.g_tok; a.count = sl.g_count; a.off = sl.g_off; a.stage = sl.g_stage; a.stage_cap = (uint32_t)stage_cap_; a.out = sl.d_out; a.out_cap = (uint32_t)out_cap_;
    a.cursors = sl.g_cursors; a.sub_off = sl.d_sub; a.sub_len = sl.d_sub + rows_; a.sub_bins = sl.d_sub + 2 * rows_;
    a.ctx_save = sl.g_ctx_save; a.ctx_ready = sl.g_ctx_ready; a.gen = ++sl.gen; a.err = err_;
    a.wc = cw_ / 64; a.hc = rows_; a.wpp = cfg_.wpp; a.tile_rows = cfg_.tile_rows; a.init_type = intra ? 0 : 1; a.qp = qp_cur_; a.first_sub = 0;
    timed(K_CABAC_ROWS, sl.ent_stream, [&] { launch_cabac_rows(a, nsub, sl.ent_stream); });
    // measurement aid (KVAZZUP_AMD_PARSE_PROBE=1): the decoder's mirror of the coder over the substreams just written -- what arithmetic DECODING costs a wave
    // per bin on this GPU, the lower bound of a slice-data parser there (cabac_kernels.hip k_cabac_decode_probe; the count of wrong bins must stay 0)
    static const bool parse_probe = getenv("KVAZZUP_AMD_PARSE_PROBE") != nullptr;
    if (parse_probe) {
      if (!probe_words_) { HIP_CHECK(hipMalloc(&probe_words_, 2 * sizeof(uint32_t))); HIP_CHECK(hipMemset(probe_words_, 0, 2 * sizeof(uint32_t))); }
      launch_cabac_decode_probe(a, nsub, probe_words_, sl.ent_stream);
    }
    HIP_CHECK(hipEventRecord(sl.done, sl.ent_stream));
  } else
  HIP_CHECK(hipEventRecord(sl.done, stream_tok_));
  // reconstruction final: with SAO the tokenizer's stream waited for the filter -- the chain's last kernel --, so sl.done already says it (and a record the
  // host can inspect ends with a system-scope fence: one fewer at the end of every picture's chain)
  if (!cfg_.sao) HIP_CHECK(hipEventRecord(sl.rec_done, ms));
  sl.has_sink = false;
  if (sink_sub_[0] && in_ring >= 0) {                      // (host pictures only: set_recon_sink)
    // the reconstruction is final behind the chain's last kernel: rec_done, with SAO the event behind the filter
    if (!stream_rec_) HIP_CHECK(stream_acquire(&stream_rec_, cfg_.device, 'R', 'n'));
    if (!sl.sink_done) HIP_CHECK(hipEventCreateWithFlags(&sl.sink_done, hipEventDisableTiming));
    HIP_CHECK(hipStreamWaitEvent(stream_rec_, cfg_.sao ? ev_src_free_[set_] : sl.rec_done, 0));
    for (int c = 0; c < 3; c++) {
      const int pw = c ? w / 2 : w, ph = c ? h / 2 : h, cp = c ? cw_ / 2 : cw_;
      if (pw == cp) HIP_CHECK(hipMemcpyAsync(sink_sub_[c], rec_[cur_idx_][c], (size_t)pw * ph, hipMemcpyDeviceToHost, stream_rec_));
      else HIP_CHECK(hipMemcpy2DAsync(sink_sub_[c], (size_t)pw, rec_[cur_idx_][c], (size_t)cp, (size_t)pw, (size_t)ph, hipMemcpyDeviceToHost, stream_rec_));
    }
    HIP_CHECK(hipEventRecord(sl.sink_done, stream_rec_));
    sl.has_sink = true;
  }
  sink_sub_[0] = sink_sub_[1] = sink_sub_[2] = nullptr;
  sl.pic_idx = submitted_; sl.poc = poc_; sl.intra = intra; sl.rec_idx = cur_idx_; sl.set = set_; sl.qp = qp_cur_; sl.write_ps = false;
  if (intra) {
    sl.write_ps = (intra_count_ == 0) || (cfg_.vps_period > 0 && (intra_count_ % cfg_.vps_period) == 0);
    intra_count_++;
  }
  frame_idx_++; prev_set_ = set_;
  ref_idx_ = cur_idx_; cur_idx_ = (cur_idx_ + 1) % nrec_;      // rec_[ref_idx_] holds the picture just submitted
  submitted_++;
  if (tok_deferred_) {
    { std::lock_guard<std::mutex> l(bm_); sl.ready = false; }
    { std::lock_guard<std::mutex> l(tm_); tq_.push_back((int)((submitted_ - 1) % nslots_)); }
    tcv_.notify_all();
  } else if (depth_ >= 2) {
    { std::lock_guard<std::mutex> l(bm_); sl.ready = false; bq_.push_back((int)((submitted_ - 1) % nslots_)); }
    bcv_.notify_all();
  }
  return true;
}

// the tokenizer's three launches of one picture (stream_tok_) and the events behind them
bool Encoder::launch_tokenizer(Slot &sl, const EncFrame &f, bool intra, bool wait_on_stream, bool prof)
{
  if (wait_on_stream) HIP_CHECK(hipStreamWaitEvent(stream_tok_, cfg_.sao ? ev_src_free_[sl.set] : ev_signalled_, 0));
  if (!intra) timed_slot(sl, prof, K_INTER_SIGNAL, stream_tok_, [&] { launch_inter_signal(f, stream_tok_); });
  timed_slot(sl, prof, K_TOKENIZE, stream_tok_, [&] { launch_tokenize(f, stream_tok_); });
  timed_slot(sl, prof, K_TOK_COMPACT, stream_tok_, [&] { launch_tok_compact(f, stream_tok_); });
  HIP_CHECK(hipEventRecord(ev_tok_done_[sl.set], stream_tok_));
  if (tok_deferred_) HIP_CHECK(hipEventRecord(sl.done, stream_tok_));
  return true;
}

// encoder.h tok_deferred_: pictures in submission order -- wait (on the host, in naps) for the picture's chain, launch its tokenizer, hand the slot to the workers
void Encoder::tok_launcher()
{
  hipSetDevice(cfg_.device);
  for (;;) {
    int idx;
    { std::unique_lock<std::mutex> l(tm_); tcv_.wait(l, [&] { return tquit_ || !tq_.empty(); }); if (tq_.empty()) return; idx = tq_.front(); tq_.pop_front(); }
    Slot &sl = slot_[idx];
    hipEvent_t e = ev_src_free_[sl.set];                   // recorded behind the chain's last kernel (k_sao); not recorded again before this picture has been collected (owf < kSets)
    const bool chain_ok = nap_until([&] { hipError_t r = hipEventQuery(e); return r == hipSuccess ? 1 : (r == hipErrorNotReady ? 0 : -1); });
    tl("tok0", sl.pic_idx);
    // (a failed query -- device fault, the chain's last kernel failed -- means the CU and SAO arrays are not final: the picture fails like one whose
    // tokenizer could not be launched, finish_slot never codes what the slot held)
    sl.tok_failed = !chain_ok || !launch_tokenizer(sl, sl.f_tok, sl.intra, false, sl.prof);
    if (sl.tok_failed) fprintf(stderr, chain_ok ? "kvazzup_amd: the tokenizer of picture %ld could not be launched\n" : "kvazzup_amd: the kernels of picture %ld failed; it is not entropy coded\n", sl.pic_idx);
    { std::lock_guard<std::mutex> l(bm_); bq_.push_back(idx); }
    bcv_.notify_all();
  }
}

bool Encoder::collect(EncodedPicture *out)
{
  Slot &sl = slot_[collected_ % nslots_];
  tl("col0", collected_);
  struct OnExit { long p; ~OnExit() { tl("col1", p); } } on_exit_{collected_};
  collected_++;
  bool ok;
  if (depth_ >= 2) {
    Tick tk;
    std::unique_lock<std::mutex> l(bm_);
    bcv_.wait(l, [&] { return sl.ready; });
    t_wait_ += tk.ms();
    ok = sl.ok;
    std::swap(*out, sl.result);
  } else ok = finish_slot(sl, out);
  out_idx_ = sl.rec_idx; out_set_ = sl.set;
  rc_bytes_[(collected_ - 1) & 7] = (uint32_t)out->au.size();   // (collected_ - 1 = index of this picture)
  return ok;
}

// owf >= 2: pictures are finished here, in submission order, while the calling thread keeps launching kernels
void Encoder::background(int worker)
{
  hipSetDevice(cfg_.device);
  for (;;) {
    int idx;
    { std::unique_lock<std::mutex> l(bm_); bcv_.wait(l, [&] { return bquit_ || !bq_.empty(); }); if (bq_.empty()) return; idx = bq_.front(); bq_.pop_front(); }
    Slot &sl = slot_[idx];
    tl("bg0", sl.pic_idx);
    const bool ok = finish_slot(sl, &sl.result, worker);
    tl("bg1", sl.pic_idx);
    { std::lock_guard<std::mutex> l(bm_); sl.ok = ok; sl.ready = true; }
    bcv_.notify_all();
  }
}

bool Encoder::finish_slot(Slot &sl, EncodedPicture *out, int worker)
{
  out->valid = false; out->au.clear(); out->recon_delivered = false;
  // (whatever way this function is left: the copy into the caller's reconstruction picture is not in flight any more -- the caller frees it on failure)
  struct SinkGuard { Slot &s; ~SinkGuard() { if (s.has_sink) { hipEventSynchronize(s.sink_done); s.has_sink = false; } } } sink_guard_{sl};
  if (sl.tok_failed) { sl.tok_failed = false; return false; }
  {
    Tick tk;
    if (depth_ >= 2 && !spin_wait_) {          // background worker: naps between queries (see nap_until)
      auto q = [&](hipEvent_t e) { return nap_until([&] { hipError_t r = hipEventQuery(e); return r == hipSuccess ? 1 : (r == hipErrorNotReady ? 0 : -1); }); };
      if (!q(sl.done) || !q(sl.rec_done)) return false;
    } else { HIP_CHECK(hipEventSynchronize(sl.done)); HIP_CHECK(hipEventSynchronize(sl.rec_done)); }
    if (depth_ < 2) t_wait_ += tk.ms();
  }
  if (*sl.h_err) { fprintf(stderr, "kvazzup_amd: device error flags 0x%x (8/16/32: token buffer overflow)\n", *sl.h_err); return false; }
  if (sl.ev_used) {
    std::lock_guard<std::mutex> l(stat_m_);
    for (size_t i = 0; i < sl.ev_used; i++) {
      float ms = 0; hipEventElapsedTime(&ms, sl.ev[i].a, sl.ev[i].b);
      k_ms_[sl.ev[i].id] += ms; k_n_[sl.ev[i].id]++;
    }
    sl.ev_used = 0;
  }
  tl("gpudone", sl.pic_idx);
  // ---- serial half of entropy coding: host threads turn the bins into the WPP substreams
  const int nsub = (cfg_.wpp ? rows_ : cfg_.tile_rows) * cfg_.tile_cols;
  uint64_t bins = 0;
  Tick tk_ar;
  std::vector<std::vector<uint8_t>> &rows_out = worker ? rows_out2_ : rows_out_;
  if (cfg_.entropy_gpu) {
    // the substreams were coded on the GPU (k_cabac_rows) and lie in the slot's host-mapped buffer
    rows_out.resize((size_t)nsub);
    const uint32_t *off = sl.h_sub, *len = sl.h_sub + rows_, *nb = sl.h_sub + 2 * rows_;
    for (int k = 0; k < nsub; k++) {
      if (len[k] == ~0u || (size_t)off[k] + len[k] > out_cap_) { fprintf(stderr, "kvazzup_amd: substream %d was not coded (token or output buffer overflow)\n", k); return false; }
      rows_out[(size_t)k].assign(sl.h_out + off[k], sl.h_out + off[k] + len[k]);
      bins += nb[k];
    }
  } else {
    for (int i = 0, n = (cw_ / 64) * rows_; i < n; i++) if (sl.h_tok_count[i] < 0) { fprintf(stderr, "kvazzup_amd: token array overflow (CTU %d)\n", i); return false; }
    EntropyHost *coder = worker ? entropy2_ : entropy_;
    coder->code_picture(sl.h_tok_dense, sl.h_tok_count, sl.h_tok_off, cw_ / 64, rows_, cfg_.wpp != 0, cfg_.tile_rows, sl.intra ? 0 : 1, sl.qp, rows_out, &bins, cfg_.tile_cols);
  }
  const double ar = tk_ar.ms();
  tl("arith", sl.pic_idx);
  if (profiling_ && !cfg_.entropy_gpu) { std::lock_guard<std::mutex> l(stat_m_); k_ms_[K_HOST_ARITH] += ar; k_n_[K_HOST_ARITH]++; }
  { std::lock_guard<std::mutex> l(stat_m_); t_arith_ += ar; }
  // ---- access unit assembly (host): parameter sets with IDR pictures, then the slice NAL
  out->valid = true; out->poc = sl.poc; out->qp = sl.qp; out->is_intra = sl.intra; out->bins = bins;
  bool assembled;
  { Tick tk; assembled = assemble_access_unit(out->au, sp_, sl.intra, sl.poc, sl.write_ps, rows_out, nsub, sl.qp - cfg_.qp); const double a = tk.ms(); std::lock_guard<std::mutex> l(stat_m_); t_asm_ += a; }
  if (!assembled) { fprintf(stderr, "kvazzup_amd: %d substreams do not fit the tile grid\n", nsub); out->valid = false; return false; }
  out->recon_delivered = false;
  if (sl.has_sink) {                                       // the reconstruction's copy into the caller's picture: queued at submission, long done by now
    tl("rec0", sl.pic_idx);
    HIP_CHECK(hipEventSynchronize(sl.sink_done));
    tl("rec1", sl.pic_idx);
    sl.has_sink = false; out->recon_delivered = true;
  }
  if (cfg_.hash) {
    // decoded picture hash SEI: the picture's reconstruction (coded size, after the loop filters) comes down once more for it -- a
    // verification aid, not part of the hot path (uvgComm sets hash = none).  The ring entry is not written again before this picture is output.
    const size_t npx = (size_t)cw_ * ch_;
    std::vector<uint8_t> pic(npx * 3 / 2);
    const uint8_t *pl[3] = {pic.data(), pic.data() + npx, pic.data() + npx + npx / 4};
    const size_t pitch[3] = {(size_t)cw_, (size_t)cw_ / 2, (size_t)cw_ / 2};
    for (int c = 0; c < 3; c++) HIP_CHECK(hipMemcpy(const_cast<uint8_t *>(pl[c]), rec_[sl.rec_idx][c], c ? npx / 4 : npx, hipMemcpyDeviceToHost));
    const std::vector<uint8_t> payload = picture_hash_payload(cfg_.hash == 2 ? 0 : 2, pl, pitch, cw_, ch_);
    BitWriter sei;
    sei.put(132, 8); sei.put((uint32_t)payload.size(), 8);
    sei.bytes(payload.data(), payload.size());
    sei.trailing();
    append_nal(out->au, 40, sei.data().data(), sei.data().size());      // SUFFIX_SEI_NUT
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Band mode: this encoder codes CTU rows [row0, row0 + nrows) of every picture; its neighbours (other processes /
// GPUs) code the rest.  Everything up to deblocking is confined to the band by the tile rules; deblocking across the
// band's boundaries needs the neighbour's boundary rows, which arrive through the halo blocks.
// ---------------------------------------------------------------------------------------------------------------
bool Encoder::band_picture_setup()
{
  HIP_CHECK(hipSetDevice(cfg_.device));
  set_ = 0; bind_set(0);
  cur_slot_ = &slot_[0];
  prof_now_ = false;
  const int period = cfg_.intra_period;
  band_intra_ = (frame_idx_ == 0) || (period > 0 && (frame_idx_ % period) == 0);
  if (band_intra_) poc_ = 0; else poc_++;
  // rate control: every band's encoder runs the same controller on the same access-unit sizes (band_report_au), so all of them
  // arrive at the same QP without talking to each other
  if (cfg_.bitrate > 0 && frame_idx_ >= 3 && !(rc_known_ & (1u << ((frame_idx_ - 3) & 7)))) {
    fprintf(stderr, "kvazzup_amd: band mode with rate control: the size of access unit %d has not been reported (kvzx_encoder_band_report_au)\n", frame_idx_ - 3);
    return false;
  }
  rate_control();
  if (frame_idx_ >= 3) rc_known_ &= ~(1u << ((frame_idx_ - 3) & 7));
  f_.qp = qp_cur_; f_.qpc = kChromaQp[qp_cur_]; f_.lambda_q4 = kLambdaQ4[qp_cur_];
  f_.is_intra = band_intra_; f_.poc = poc_;
  for (int c = 0; c < 3; c++) { f_.rec[c] = rec_[cur_idx_][c]; f_.ref[c] = rec_[ref_idx_][c]; }
  f_.me_ref = f_.ref[0];
  Slot &sl = slot_[0];
  f_.tok_dense = sl.d_tok_dense; f_.tok_count_out = sl.d_tok_count; f_.tok_off_out = sl.d_tok_off; f_.err_out = sl.d_err;
  f_.tok_cursor = (uint32_t *)tok_count_ + (size_t)(frame_idx_ & 1) * tok_nctu_; f_.tok_cursor_next = (uint32_t *)tok_count_ + (size_t)((frame_idx_ + 1) & 1) * tok_nctu_;
  return true;
}

// band mode: the size of a finished access unit (picture index = count of pictures before it), for the rate controller
void Encoder::band_report_au(long picture, uint32_t bytes) { rc_bytes_[picture & 7] = bytes; rc_known_ |= 1u << (picture & 7); }

bool Encoder::band_phase1(const uint8_t *d_i420)
{
  if (cfg_.band_rows <= 0 || !d_i420) return false;
  if (!band_picture_setup()) return false;
  roi_sub_ = roi_; roi_sub_w_ = roi_w_; roi_sub_h_ = roi_h_;
  if (!stage_roi(stream_) || !picture_begin(stream_)) return false;
  f_.chain_gen = next_chain_gen();
  const EncFrame f = f_;
  launch_pad_input(d_i420, cfg_.width, cfg_.height, src_[0][0], src_[0][1], src_[0][2], cw_, ch_, stream_);
  if (band_intra_) {
    launch_intra_analyse(f, stream_);
    HIP_CHECK(hipMemsetAsync(sync_, 0, sizeof(uint32_t) * (rows_ * (cw_ / 64) * 3 + 1), stream_));
    HIP_CHECK(hipMemsetAsync(f_.cu_cbf + (size_t)f.row0 * 8 * f.b8w, 0, (size_t)f.b8w * 8 * band_rows(f), stream_));
    launch_intra_recon(f, stream_);
  } else {
    launch_me(f, stream_);
    if (cfg_.subme > 0) launch_subpel(f, stream_);
    launch_inter_recon(f, stream_);
    launch_inter_signal(f, stream_);
  }
  launch_qp_resolve(f, stream_);
  if (cfg_.deblock) launch_deblock_v(f, stream_);
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

// one halo block: [luma 4 rows | Cb 2 rows | Cr 2 rows | cu_log2 | cu_intra | cu_cbf (one 8x8 row each) | cu_mv (one 8x8 row)]
size_t Encoder::halo_bytes() const { return (size_t)cw_ * 4 + (size_t)(cw_ / 2) * 2 * 2 + (size_t)(cw_ / 8) * 3 + (size_t)(cw_ / 8) * 4; }

bool Encoder::band_export_halo(uint8_t *d_up, uint8_t *d_down)
{
  const int Y0 = f_.row0 * 64, Y1 = (f_.row0 + band_rows(f_)) * 64, b8w = f_.b8w, cw2 = cw_ / 2;
  for (int side = 0; side < 2; side++) {
    uint8_t *dst = side ? d_down : d_up;
    if (!dst) continue;
    const int y = side ? Y1 - 4 : Y0, b8y = side ? Y1 / 8 - 1 : Y0 / 8;        // first luma row of the 4-row strip; CU row
    size_t o = 0;
    HIP_CHECK(hipMemcpyAsync(dst + o, f_.rec[0] + (size_t)y * cw_, (size_t)cw_ * 4, hipMemcpyDeviceToDevice, stream_)); o += (size_t)cw_ * 4;
    for (int c = 1; c < 3; c++) { HIP_CHECK(hipMemcpyAsync(dst + o, f_.rec[c] + (size_t)(y / 2) * cw2, (size_t)cw2 * 2, hipMemcpyDeviceToDevice, stream_)); o += (size_t)cw2 * 2; }
    const uint8_t *arr[3] = {f_.cu_log2, f_.cu_intra, f_.cu_cbf};
    for (int k = 0; k < 3; k++) { HIP_CHECK(hipMemcpyAsync(dst + o, arr[k] + (size_t)b8y * b8w, (size_t)b8w, hipMemcpyDeviceToDevice, stream_)); o += (size_t)b8w; }
    HIP_CHECK(hipMemcpyAsync(dst + o, f_.cu_mv + (size_t)b8y * b8w * 2, (size_t)b8w * 4, hipMemcpyDeviceToDevice, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

bool Encoder::band_import_halo(const uint8_t *d_from_up, const uint8_t *d_from_down)
{
  const int Y0 = f_.row0 * 64, Y1 = (f_.row0 + band_rows(f_)) * 64, b8w = f_.b8w, cw2 = cw_ / 2;
  for (int side = 0; side < 2; side++) {
    const uint8_t *src = side ? d_from_down : d_from_up;
    if (!src) continue;
    // from above: the neighbour's LAST rows land just above this band; from below: its FIRST rows just below
    const int y = side ? Y1 : Y0 - 4, b8y = side ? Y1 / 8 : Y0 / 8 - 1;
    if (y < 0 || y + 4 > ch_) return false;
    size_t o = 0;
    HIP_CHECK(hipMemcpyAsync(f_.rec[0] + (size_t)y * cw_, src + o, (size_t)cw_ * 4, hipMemcpyDeviceToDevice, stream_)); o += (size_t)cw_ * 4;
    for (int c = 1; c < 3; c++) { HIP_CHECK(hipMemcpyAsync(f_.rec[c] + (size_t)(y / 2) * cw2, src + o, (size_t)cw2 * 2, hipMemcpyDeviceToDevice, stream_)); o += (size_t)cw2 * 2; }
    uint8_t *arr[3] = {f_.cu_log2, f_.cu_intra, f_.cu_cbf};
    for (int k = 0; k < 3; k++) { HIP_CHECK(hipMemcpyAsync(arr[k] + (size_t)b8y * b8w, src + o, (size_t)b8w, hipMemcpyDeviceToDevice, stream_)); o += (size_t)b8w; }
    HIP_CHECK(hipMemcpyAsync(f_.cu_mv + (size_t)b8y * b8w * 2, src + o, (size_t)b8w * 4, hipMemcpyDeviceToDevice, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

// Second phase in two halves, so that the halo exchange can run beside the first: (a) needs nothing from the neighbouring bands --
// the band's inner horizontal edges, the tokenizer, the host arithmetic coder (the tokens do not depend on deblocked samples) --
// and (b), after the halo rows have been imported, filters the two boundary edges and closes the picture.
bool Encoder::band_phase2a()
{
  if (cfg_.band_rows <= 0) return false;
  HIP_CHECK(hipSetDevice(cfg_.device));
  const EncFrame f = f_;
  Slot &sl = slot_[0];
  if (cfg_.deblock) launch_deblock_h(f, stream_, 1);
  launch_tokenize(f, stream_); launch_tok_compact(f, stream_);
  HIP_CHECK(hipStreamSynchronize(stream_));
  if (*sl.h_err) { fprintf(stderr, "kvazzup_amd: device error flags 0x%x\n", *sl.h_err); return false; }
  const int wc = cw_ / 64;
  for (int i = f.row0 * wc; i < (f.row0 + band_rows(f)) * wc; i++) if (sl.h_tok_count[i] < 0) { fprintf(stderr, "kvazzup_amd: token array overflow (CTU %d)\n", i); return false; }
  band_bins_ = 0;
  entropy_->code_band(sl.h_tok_dense, sl.h_tok_count, sl.h_tok_off, wc, rows_, cfg_.wpp != 0, cfg_.tile_rows, band_intra_ ? 0 : 1, qp_cur_,
                      f.row0, band_rows(f), band_subs_, &band_bins_);
  band_coded_ = true;
  return true;
}

bool Encoder::band_phase2b(std::vector<std::vector<uint8_t>> *substreams, EncodedPicture *info)
{
  if (cfg_.band_rows <= 0 || !substreams || !band_coded_) return false;
  HIP_CHECK(hipSetDevice(cfg_.device));
  if (cfg_.deblock) { launch_deblock_h(f_, stream_, 2); HIP_CHECK(hipStreamSynchronize(stream_)); }
  substreams->swap(band_subs_);
  band_coded_ = false;
  if (info) { info->valid = true; info->poc = poc_; info->qp = qp_cur_; info->is_intra = band_intra_; info->bins = band_bins_; info->au.clear(); }
  frame_idx_++;
  if (band_intra_) intra_count_++;
  ref_idx_ = cur_idx_; cur_idx_ = (cur_idx_ + 1) % nrec_; out_idx_ = ref_idx_;
  return true;
}

bool Encoder::band_phase2(std::vector<std::vector<uint8_t>> *substreams, EncodedPicture *info) { return band_phase2a() && band_phase2b(substreams, info); }

// The picture last output: its reconstruction is complete (collect waited for the slot's events) and its ring entry is not written again
// before the next picture is submitted, so the copy goes on a stream of its own -- waiting on the main stream would wait for every
// picture queued behind this one (owf).
bool Encoder::download_recon(uint8_t *y, uint8_t *u, uint8_t *v)
{
  uint8_t *dst[3] = {y, u, v};
  HIP_CHECK(hipSetDevice(cfg_.device));
  if (!stream_rec_) HIP_CHECK(stream_acquire(&stream_rec_, cfg_.device, 'R', 'n'));
  if (pending() == 0) HIP_CHECK(hipStreamSynchronize(stream_));      // (band mode, debugging: nothing vouches for the picture but the main stream)
  for (int c = 0; c < 3; c++) {
    int w = c ? cfg_.width / 2 : cfg_.width, h = c ? cfg_.height / 2 : cfg_.height, pw = c ? cw_ / 2 : cw_;
    if (w == pw) HIP_CHECK(hipMemcpyAsync(dst[c], rec_[out_idx_][c], (size_t)w * h, hipMemcpyDeviceToHost, stream_rec_));
    else HIP_CHECK(hipMemcpy2DAsync(dst[c], (size_t)w, rec_[out_idx_][c], (size_t)pw, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, stream_rec_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_rec_));
  return true;
}

bool Encoder::debug_copy(const char *what, void *dst, size_t bytes)
{
  const size_t npx = (size_t)cw_ * ch_, nb8 = npx / 64;
  const void *src = nullptr; size_t have = 0;
  std::string w(what);
  static const char *names[7] = {"cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf"};
  for (int i = 0; i < 7; i++) if (w == names[i]) { src = cu_bytes_[out_set_] + i * nb8; have = nb8; }
  if (w == "cu_mv") { src = cu_mv_[out_set_]; have = nb8 * 4; }
  if (w == "cu_mvd") { src = cu_mvd_[out_set_]; have = nb8 * 4; }
  if (w == "trace" && trace_) { src = trace_; have = sizeof(unsigned long long) * (rows_ * (cw_ / 64) * 72); }
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    if (w == std::string("coef") + char('0' + c)) { src = coef_[out_set_][c]; have = n * 2; }
    if (w == std::string("rec") + char('0' + c)) { src = rec_[out_idx_][c]; have = n; }
    if (w == std::string("src") + char('0' + c)) { src = src_[out_set_][c]; have = n; }
  }
  if (!src || bytes > have) return false;
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return true;
}

}  // namespace kvzx
