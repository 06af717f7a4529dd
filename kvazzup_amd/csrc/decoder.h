// kvazzup_amd/csrc/decoder.h -- host engine of the HIP decoder behind libOpenHevc*
// (/root/reference/src/media/processing/openhevcfilter.cpp:36-56,145-146,195-199).
//
// Division of labour: the host parses NAL units and runs CABAC decoding (bit-serial, one
// substream per CTU row), which yields per-CU records, motion vectors and the levels of the coded
// transform blocks; those are uploaded and the GPU does everything that touches samples:
// motion-compensated / intra prediction, dequantisation + inverse DCT, reconstruction and
// deblocking (dec variants of the encoder kernels, enc_kernels.hip).
//
// Supported streams (round 1): the tool set this project's encoder emits -- Main profile 8-bit
// 4:2:0, CTB 64, coded size a multiple of 64 and at least 128 wide, 2Nx2N CUs of 8/16/32 (intra)
// and 16/32 (inter) with one transform unit each, I and P slices with the previous picture as
// the only reference, arbitrary quarter-sample motion vectors, merge/AMVP without TMVP, WPP or
// plain slice data, one slice per picture, deblocking on/off.  Anything else is rejected with
// a negative return value (oracle/hevc_dec.c is the general CPU checker).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "hevc_core.h"
#include "host_pool.h"
#include "enc_kernels.h"

namespace kvzx {

// DK_HOST_PARSE is not a kernel: wall time of the host CABAC parsing stage
enum DecKernelId { DK_SCATTER = 0, DK_INTER_RECON, DK_INTRA_RECON, DK_DEBLOCK, DK_HOST_PARSE, DK_SAO, DK_COUNT };

struct DecSps {
  bool valid = false;
  int width = 0, height = 0;          // coded size
  int crop_r = 0, crop_b = 0, crop_l = 0, crop_t = 0;   // luma samples
  int log2_max_poc_lsb = 8;
  int num_st_rps = 0; int rps_neg[64], rps_used[64];      // first negative entry of each SPS RPS (subset check)
  uint32_t fps_num = 0, fps_den = 0;
  int strong_intra = 0;
  int sao = 0;                        // sample_adaptive_offset_enabled_flag
};
struct DecPps {
  bool valid = false;
  int qp_in_cu = 0;                   // cu_qp_delta_enabled_flag with diff_cu_qp_delta_depth == 0 (quantisation group = CTU)
  int tile_rows = 1;                  // full-width tile rows with uniform spacing (everything else about tiles is rejected)
  int init_qp = 26, wpp = 0, deblock_control = 0, deblock_disabled = 0, loop_filter_across_slices = 1, cabac_init_present = 0;
};

struct DecodedPicture {
  int width = 0, height = 0;          // cropped
  int coded_w = 0, coded_h = 0;
  const uint8_t *host[3] = {nullptr, nullptr, nullptr}; int host_pitch[3] = {0, 0, 0};
  const uint8_t *dev[3] = {nullptr, nullptr, nullptr}; int dev_pitch[3] = {0, 0, 0};
  int poc = 0; int64_t pts = 0; uint32_t fps_num = 0, fps_den = 0; bool is_intra = false;
};

// worker threads that parse whole pictures concurrently (frame threading)
class FrameWorkers {
 public:
  explicit FrameWorkers(int n) { for (int i = 0; i < n; i++) t_.emplace_back([this] { name_this_thread("kvzx-parse"); run(); }); }
  ~FrameWorkers() { { std::lock_guard<std::mutex> l(m_); quit_ = true; } cv_.notify_all(); for (auto &t : t_) t.join(); }
  void submit(std::function<void()> f) { { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(f)); } cv_.notify_one(); }
 private:
  void run()
  {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [this] { return quit_ || !q_.empty(); }); if (q_.empty()) return; f = std::move(q_.front()); q_.pop_front(); }
      f();
    }
  }
  std::vector<std::thread> t_; std::mutex m_; std::condition_variable cv_; std::deque<std::function<void()>> q_; bool quit_ = false;
};

class Decoder {
 public:
  explicit Decoder(int device) : device_(device) {}
  // frame threading: up to n pictures are parsed concurrently while one more is reconstructed on the GPU; the output is delayed by n pictures
  // (libOpenHevcInit(nb_threads, OH_THREAD_FRAME / OH_THREAD_FRAMESLICE)); call before the first picture
  void set_frame_threads(int n) { if (jobs_.empty()) frame_threads_ = n < 1 ? 1 : (n > 16 ? 16 : n); }
  int frame_threads() const { return frame_threads_; }
  ~Decoder();
  bool start(std::string *error);           // checks the HIP device; no CPU fallback
  // One NAL unit (with or without start code).  <0 error/unsupported, 0 nothing to output, 1 picture ready.
  int decode_nal(const uint8_t *data, size_t len, int64_t pts);
  bool get_picture(DecodedPicture *out);   // the picture announced by the last decode_nal() == 1
  void set_download(bool on) { download_ = on; }
  void set_profiling(int every) { profiling_ = every > 0; prof_every_ = every > 0 ? every : 1; }   // n: every n-th picture
  void set_parse_threads(int n) { if (!pool_) parse_threads_ = n < 1 ? 1 : n; }
  void get_kernel_times(double *ms, uint64_t *launches, bool reset);
  bool debug_copy(const char *what, void *dst, size_t bytes);
  int last_error() const { return last_error_; }
  void flush() {}
  int pending() const { return (int)(job_head_ - job_tail_) + (gpu_job_ ? 1 : 0); }

 private:
  bool ensure_buffers(int cw, int ch);
  void free_buffers();
  int decode_slice(const uint8_t *rbsp, size_t len, int nal_type, int64_t pts);
  struct RowState { std::vector<uint32_t> levels; std::vector<TuDesc> tus; int rc = 0; };     // levels: (position << 16 | level) words
  struct alignas(64) Progress { std::atomic<int> v{0}; char pad[60]; };   // one cache line per row: no false sharing between pollers
  // everything one picture needs between its slice header and its reconstruction
  struct PicJob {
    std::vector<uint8_t> rbsp; size_t data_off = 0, data_len = 0;
    std::vector<size_t> sub_start;
    int tile_rows = 1, qp_in_cu = 0;
    int sao_luma = 0, sao_chroma = 0;                          // slice_sao_luma_flag / slice_sao_chroma_flag
    int slice_qp = 0, max_merge = 5, poc = 0; bool is_intra = false, deblock = true; int64_t pts = 0;
    int crop[4] = {0, 0, 0, 0}; uint32_t fps_num = 0, fps_den = 0;
    // Everything the GPU needs for the picture, in one pinned block that goes over in one copy:
    // [ CU records: 7 byte arrays of b8 entries | motion vectors: b8 x 2 int16 | per CTU: QpY, delta, first coded CU | per CTU: SaoParams | TuDesc x ntu | level words x nlev ]
    uint8_t *h_in = nullptr; size_t h_in_cap = 0; size_t ntu = 0, nlev = 0;
    EncFrame hf{};                                             // host view of the CU / motion arrays inside h_in
    std::vector<RowState> rows; std::vector<uint8_t> wpp_saved;
    std::unique_ptr<Progress[]> row_progress; int row_progress_n = 0;
    std::atomic<int> state{0}; int rc = 0; double parse_ms = 0; int rec_idx = 0;
    PicJob() {}
    PicJob(const PicJob &) {}                                  // (vector<PicJob> construction only)
  };
  int parse_job(PicJob &job, bool row_parallel);
  int parse_row(PicJob &job, int row, const uint8_t *data, size_t len, RowState &rs);
  int finish_oldest();
  void drop_pending();
  size_t sao_offset() const { return (size_t)cw_ * ch_ / 64 * 11 + (size_t)(cw_ / 64) * (ch_ / 64) * 3; }
  size_t fixed_bytes() const { return sao_offset() + (size_t)(cw_ / 64) * (ch_ / 64) * sizeof(SaoParams); }   // CU records + motion vectors + per-CTU QpY / delta / first coded CU + per-CTU SAO parameters
  bool grow_job_input(PicJob &job, size_t bytes);
  void bind_views(EncFrame &f, uint8_t *base);
  int launch_gpu(PicJob &job);
  int complete_gpu();

  int device_; bool started_ = false;
  hipStream_t stream_ = nullptr;
  DecSps sps_[16]; DecPps pps_[64]; uint32_t vps_fps_num_ = 0, vps_fps_den_ = 0;
  int cw_ = 0, ch_ = 0;
  std::vector<PicJob> jobs_; int frame_threads_ = 1; long job_head_ = 0, job_tail_ = 0;
  std::unique_ptr<FrameWorkers> workers_;
  // device side
  EncFrame f_{};
  uint8_t *d_in_ = nullptr; size_t d_in_cap_ = 0;          // device copy of PicJob::h_in
  int16_t *d_mvd_ = nullptr;
  uint8_t *rec_[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  uint8_t *work_[3] = {nullptr, nullptr, nullptr};          // pictures with SAO: reconstruction and deblocking happen here, the filter writes into rec_
  int16_t *coef_[3] = {nullptr, nullptr, nullptr};
  uint32_t *sync_ = nullptr, *err_ = nullptr; uint32_t *h_err_ = nullptr;
  uint8_t *h_out_ = nullptr; size_t h_out_cap_ = 0;
  long launched_ = 0; int out_idx_ = 0; bool have_ref_ = false;
  double t_parse_max_ = 0;                // trace: the longest parse of one picture (an IDR), ms
  bool spin_wait_ = false;                // KVAZZUP_AMD_SPIN: poll the GPU instead of napping between queries
  PicJob *gpu_job_ = nullptr;             // picture whose kernels are in flight (frame-threaded mode)
  int poc_ = 0, prev_poc_ = 0;
  bool download_ = true, profiling_ = false, prof_now_ = false; int prof_every_ = 1;
  bool pic_ready_ = false; DecodedPicture out_;
  const DecSps *active_sps_ = nullptr;
  int last_error_ = 0;
  double t_nal_ = 0, t_wait_ = 0, t_stage_ = 0, t_api_ = 0, t_sync_ = 0;     // decoder-thread time split (KVAZZUP_AMD_TRACE)
  std::vector<uint8_t> rbsp_;
  std::vector<size_t> epb_;                // unescaped payload offset of every removed emulation prevention byte
  std::vector<size_t> sub_start_;          // start of every WPP substream inside the unescaped slice data
  std::unique_ptr<OrderedPool> pool_; int parse_threads_ = 16;
  struct EvPair { hipEvent_t a, b; int id; };
  std::vector<EvPair> ev_pool_; size_t ev_used_ = 0;
  double k_ms_[DK_COUNT] = {0}; uint64_t k_n_[DK_COUNT] = {0};
  template <class F> void timed(int id, F &&launch);
};

}  // namespace kvzx
