// kvazzup_amd/csrc/encoder.hip -- see encoder.h
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstring>
#include "encoder.h"
#include "enc_kernels.h"

namespace kvzx {

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (error) *error = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "kvazzup_amd: %s failed: %s\n", #expr, hipGetErrorString(e_)); return false; } } while (0)

Encoder *Encoder::create(const EncoderConfig &cfg, std::string *error)
{
  Encoder *e = new Encoder();
  if (!e->init(cfg, error)) { delete e; return nullptr; }
  return e;
}

bool Encoder::init(const EncoderConfig &cfg, std::string *error)
{
  if (cfg.width < 16 || cfg.height < 16 || (cfg.width & 1) || (cfg.height & 1) || cfg.width > 16384 || cfg.height > 16384) {
    if (error) *error = "unsupported picture size"; return false;
  }
  if (cfg.qp < 0 || cfg.qp > 51 || cfg.me_range < 1 || cfg.me_range > 32) { if (error) *error = "qp or me-range out of range"; return false; }
  cfg_ = cfg;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg.device) {
    if (error) *error = "no usable HIP device (this library has no CPU fallback)"; return false;
  }
  HIP_OK(hipSetDevice(cfg.device));
  cw_ = (cfg.width + 63) & ~63; ch_ = (cfg.height + 63) & ~63;
  if (cw_ < 128) cw_ = 128;
  rows_ = ch_ / 64;
  HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  const size_t npx = (size_t)cw_ * ch_, nb8 = npx / 64, in_bytes = (size_t)cfg.width * cfg.height * 3 / 2;
  HIP_OK(hipMalloc(&d_in_, in_bytes));
  HIP_OK(hipHostMalloc(&h_in_, in_bytes, hipHostMallocDefault));
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    HIP_OK(hipMalloc(&src_[c], n));
    HIP_OK(hipMalloc(&rec_[0][c], n)); HIP_OK(hipMalloc(&rec_[1][c], n));
    HIP_OK(hipMemset(rec_[0][c], 0, n)); HIP_OK(hipMemset(rec_[1][c], 0, n));
    HIP_OK(hipMalloc(&coef_[c], n * sizeof(int16_t)));
    HIP_OK(hipMemset(coef_[c], 0, n * sizeof(int16_t)));
  }
  HIP_OK(hipMalloc(&cu_bytes_, nb8 * 7)); HIP_OK(hipMemset(cu_bytes_, 0, nb8 * 7));
  HIP_OK(hipMalloc(&cu_mv_, nb8 * 2 * sizeof(int16_t))); HIP_OK(hipMemset(cu_mv_, 0, nb8 * 2 * sizeof(int16_t)));
  HIP_OK(hipMalloc(&cu_mvd_, nb8 * 2 * sizeof(int16_t))); HIP_OK(hipMemset(cu_mvd_, 0, nb8 * 2 * sizeof(int16_t)));
  // intra scratch: ic8 (nb8 u32) | ic16 (nb8/4 u32) | ic32 (nb8/16 u32) | im8 | im16 | im32
  size_t isz = nb8 * 4 + nb8 + nb8 / 4 + nb8 + nb8 / 4 + nb8 / 16 + 64;
  HIP_OK(hipMalloc(&intra_scratch_, isz));
  const int nctu = (cw_ / 64) * rows_;
  tok_cap_ = 49152;                                // tokens per CTU slot (worst case of a 64x64 CTU is ~43k)
  HIP_OK(hipMalloc(&tok_buf_, (size_t)nctu * tok_cap_ * sizeof(uint16_t)));
  HIP_OK(hipMalloc(&tok_count_, sizeof(uint32_t) * nctu));
  HIP_OK(hipMalloc(&tok_seg_, sizeof(uint32_t) * nctu * 32));
  HIP_OK(hipMalloc(&tok_off_, sizeof(uint32_t) * (nctu + 1)));
  tok_dense_cap_ = (size_t)nctu * tok_cap_;
  if (tok_dense_cap_ > ((size_t)1 << 27)) tok_dense_cap_ = (size_t)1 << 27;
  HIP_OK(hipHostMalloc(&h_tok_dense_, tok_dense_cap_ * sizeof(uint16_t), hipHostMallocMapped));
  HIP_OK(hipHostMalloc(&h_tok_count_, sizeof(int32_t) * nctu, hipHostMallocMapped));
  HIP_OK(hipMalloc(&sync_, sizeof(uint32_t) * rows_));
  HIP_OK(hipMalloc(&err_, sizeof(uint32_t))); HIP_OK(hipMemset(err_, 0, sizeof(uint32_t)));
  HIP_OK(hipHostMalloc(&h_err_, sizeof(uint32_t), hipHostMallocDefault));
  entropy_ = new EntropyHost(cfg.entropy_threads < rows_ ? cfg.entropy_threads : rows_);

  memset(&f_, 0, sizeof(f_));
  f_.cw = cw_; f_.ch = ch_; f_.b8w = cw_ / 8; f_.b8h = ch_ / 8;
  f_.qp = cfg.qp; f_.qpc = kChromaQp[cfg.qp]; f_.lambda_q4 = kLambdaQ4[cfg.qp]; f_.range = cfg.me_range;
  f_.wpp = cfg.wpp;
  for (int c = 0; c < 3; c++) { f_.src[c] = src_[c]; f_.coef[c] = coef_[c]; }
  f_.cu_log2 = cu_bytes_; f_.cu_intra = cu_bytes_ + nb8; f_.cu_flags = cu_bytes_ + 2 * nb8; f_.cu_merge_idx = cu_bytes_ + 3 * nb8;
  f_.cu_mvp_idx = cu_bytes_ + 4 * nb8; f_.cu_intra_mode = cu_bytes_ + 5 * nb8; f_.cu_cbf = cu_bytes_ + 6 * nb8;
  f_.cu_mv = cu_mv_; f_.cu_mvd = cu_mvd_;
  uint8_t *p = intra_scratch_;
  f_.ic8 = (uint32_t *)p; p += nb8 * 4; f_.ic16 = (uint32_t *)p; p += nb8; f_.ic32 = (uint32_t *)p; p += nb8 / 4;
  f_.im8 = p; p += nb8; f_.im16 = p; p += nb8 / 4; f_.im32 = p;
  f_.tok_buf = tok_buf_; f_.tok_cap = tok_cap_; f_.tok_cursor = (uint32_t *)tok_count_; f_.tok_seg = tok_seg_; f_.tok_off = tok_off_;
  void *dp = nullptr;
  HIP_OK(hipHostGetDevicePointer(&dp, h_tok_dense_, 0)); f_.tok_dense = (uint16_t *)dp; f_.tok_dense_cap = (uint32_t)tok_dense_cap_;
  HIP_OK(hipHostGetDevicePointer(&dp, h_tok_count_, 0)); f_.tok_count_out = (int32_t *)dp;
  f_.sync = sync_; f_.err = err_;

  sp_.cw = cw_; sp_.ch = ch_; sp_.width = cfg.width; sp_.height = cfg.height; sp_.qp = cfg.qp; sp_.wpp = cfg.wpp;
  sp_.deblock = cfg.deblock; sp_.fps_num = cfg.fps_num; sp_.fps_den = cfg.fps_den;
  HIP_OK(hipStreamSynchronize(stream_));
  HIP_OK(hipDeviceSynchronize());
  return true;
}

Encoder::~Encoder()
{
  if (stream_) hipStreamSynchronize(stream_);
  for (auto &e : ev_pool_) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  hipFree(d_in_); hipHostFree(h_in_);
  for (int c = 0; c < 3; c++) { hipFree(src_[c]); hipFree(rec_[0][c]); hipFree(rec_[1][c]); hipFree(coef_[c]); }
  hipFree(cu_bytes_); hipFree(cu_mv_); hipFree(cu_mvd_); hipFree(intra_scratch_);
  delete entropy_;
  hipFree(tok_buf_); hipFree(tok_count_); hipFree(tok_seg_); hipFree(tok_off_); hipFree(sync_); hipFree(err_);
  hipHostFree(h_tok_dense_); hipHostFree(h_tok_count_); hipHostFree(h_err_);
  if (stream_) hipStreamDestroy(stream_);
}

void Encoder::timed(KernelId id, const std::function<void()> &launch)
{
  if (!profiling_) { launch(); return; }
  if (ev_used_ == ev_pool_.size()) {
    EvPair p; hipEventCreate(&p.a); hipEventCreate(&p.b); p.id = id; ev_pool_.push_back(p);
  }
  EvPair &p = ev_pool_[ev_used_++]; p.id = id;
  hipEventRecord(p.a, stream_);
  launch();
  hipEventRecord(p.b, stream_);
}

void Encoder::get_kernel_times(double *ms, uint64_t *launches, bool reset)
{
  for (int i = 0; i < K_COUNT; i++) { if (ms) ms[i] = k_ms_[i]; if (launches) launches[i] = k_n_[i]; }
  if (reset) for (int i = 0; i < K_COUNT; i++) { k_ms_[i] = 0; k_n_[i] = 0; }
}

bool Encoder::encode_host(const uint8_t *y, const uint8_t *u, const uint8_t *v, EncodedPicture *out)
{
  const size_t ny = (size_t)cfg_.width * cfg_.height;
  memcpy(h_in_, y, ny); memcpy(h_in_ + ny, u, ny / 4); memcpy(h_in_ + ny + ny / 4, v, ny / 4);
  HIP_CHECK(hipMemcpyAsync(d_in_, h_in_, ny * 3 / 2, hipMemcpyHostToDevice, stream_));
  return encode_device(d_in_, out);
}

bool Encoder::encode_device(const uint8_t *d_i420, EncodedPicture *out)
{
  HIP_CHECK(hipSetDevice(cfg_.device));
  const size_t ny = (size_t)cfg_.width * cfg_.height;
  const int w = cfg_.width, h = cfg_.height;
  timed(K_PAD, [&] {
    launch_pad_input(d_i420, w, h, src_[0], cw_, ch_, stream_);
    launch_pad_input(d_i420 + ny, w / 2, h / 2, src_[1], cw_ / 2, ch_ / 2, stream_);
    launch_pad_input(d_i420 + ny + ny / 4, w / 2, h / 2, src_[2], cw_ / 2, ch_ / 2, stream_);
  });
  return run_picture(out);
}

bool Encoder::run_picture(EncodedPicture *out)
{
  const int period = cfg_.intra_period;
  const bool intra = (frame_idx_ == 0) || (period > 0 && (frame_idx_ % period) == 0);
  if (intra) poc_ = 0; else poc_++;
  f_.is_intra = intra; f_.poc = poc_;
  for (int c = 0; c < 3; c++) { f_.rec[c] = rec_[cur_idx_][c]; f_.ref[c] = rec_[ref_idx_][c]; }
  const EncFrame f = f_;
  if (intra) {
    timed(K_INTRA_ANALYSE, [&] { launch_intra_analyse(f, stream_); });
    HIP_CHECK(hipMemsetAsync(sync_, 0, sizeof(uint32_t) * rows_, stream_));
    timed(K_INTRA_RECON, [&] { launch_intra_recon(f, stream_); });
  } else {
    timed(K_ME, [&] { launch_me(f, stream_); });
    timed(K_INTER_RECON, [&] { launch_inter_recon(f, stream_); });
    timed(K_INTER_SIGNAL, [&] { launch_inter_signal(f, stream_); });
  }
  if (cfg_.deblock) timed(K_DEBLOCK, [&] { launch_deblock(f, stream_); });
  timed(K_TOKENIZE, [&] { launch_tokenize(f, stream_); });
  HIP_CHECK(hipMemcpyAsync(h_err_, err_, sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  if (*h_err_) { fprintf(stderr, "kvazzup_amd: device error flags 0x%x (8/16/32: token buffer overflow)\n", *h_err_); return false; }
  if (profiling_) {
    for (size_t i = 0; i < ev_used_; i++) {
      float ms = 0; hipEventElapsedTime(&ms, ev_pool_[i].a, ev_pool_[i].b);
      k_ms_[ev_pool_[i].id] += ms; k_n_[ev_pool_[i].id]++;
    }
    ev_used_ = 0;
  }
  // ---- serial half of entropy coding: host threads turn the bins into the WPP substreams
  const int nsub = cfg_.wpp ? rows_ : 1;
  uint64_t bins = 0;
  auto t0 = std::chrono::steady_clock::now();
  entropy_->code_picture(h_tok_dense_, h_tok_count_, cw_ / 64, rows_, cfg_.wpp != 0, intra ? 0 : 1, cfg_.qp, rows_out_, &bins);
  if (profiling_) { k_ms_[K_HOST_ARITH] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); k_n_[K_HOST_ARITH]++; }
  // ---- access unit assembly (host): parameter sets with IDR pictures, then the slice NAL
  out->poc = poc_; out->is_intra = intra; out->bins = bins;
  bool write_ps = false;
  if (intra) {
    write_ps = (intra_count_ == 0) || (cfg_.vps_period > 0 && (intra_count_ % cfg_.vps_period) == 0);
    intra_count_++;
  }
  assemble_access_unit(out->au, sp_, intra, poc_, write_ps, rows_out_, nsub);
  frame_idx_++;
  int t = cur_idx_; cur_idx_ = ref_idx_; ref_idx_ = t;     // rec_[ref_idx_] now holds the picture just coded
  return true;
}

bool Encoder::download_recon(uint8_t *y, uint8_t *u, uint8_t *v)
{
  uint8_t *dst[3] = {y, u, v};
  for (int c = 0; c < 3; c++) {
    int w = c ? cfg_.width / 2 : cfg_.width, h = c ? cfg_.height / 2 : cfg_.height, pw = c ? cw_ / 2 : cw_;
    HIP_CHECK(hipMemcpy2DAsync(dst[c], (size_t)w, rec_[ref_idx_][c], (size_t)pw, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

bool Encoder::debug_copy(const char *what, void *dst, size_t bytes)
{
  const size_t npx = (size_t)cw_ * ch_, nb8 = npx / 64;
  const void *src = nullptr; size_t have = 0;
  std::string w(what);
  static const char *names[7] = {"cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf"};
  for (int i = 0; i < 7; i++) if (w == names[i]) { src = cu_bytes_ + i * nb8; have = nb8; }
  if (w == "cu_mv") { src = cu_mv_; have = nb8 * 4; }
  if (w == "cu_mvd") { src = cu_mvd_; have = nb8 * 4; }
  for (int c = 0; c < 3; c++) {
    size_t n = c ? npx / 4 : npx;
    if (w == std::string("coef") + char('0' + c)) { src = coef_[c]; have = n * 2; }
    if (w == std::string("rec") + char('0' + c)) { src = rec_[ref_idx_][c]; have = n; }
    if (w == std::string("src") + char('0' + c)) { src = src_[c]; have = n; }
  }
  if (!src || bytes > have) return false;
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return true;
}

}  // namespace kvzx
