// kvazzup_amd/csrc/decoder.h -- host engine of the HIP decoder behind libOpenHevc*
// (/root/reference/src/media/processing/openhevcfilter.cpp:36-56,145-146,195-199).
//
// Division of labour: the host parses NAL units and runs CABAC decoding (bit-serial, one substream per CTU row or tile),
// which yields per-4x4 records (motion, edges, QpY), the transform blocks in decoding order and their non-zero levels
// (dec_frame.h); those go to the GPU in one copy and the GPU does everything that touches samples: motion compensation,
// intra prediction, dequantisation + inverse transforms, reconstruction, deblocking, SAO (dec_kernels.hip).
//
// Supported streams: what a Main-profile encoder in a video call produces and OpenHEVC would be asked to decode -- 8-bit 4:2:0,
// CTB 64 (Kvazaar's fixed geometry), 32 or 16 / minimum CB 8 (Kvazaar's), 16 or 32 / transform blocks 4..min(32, CTB), coded sizes that are multiples of 8, I, P and B
// slices (both reference lists, bi-prediction, pictures handed out in POC order), every CU size and partitioning (AMP included), intra CUs in P pictures, NxN intra, all chroma
// prediction modes, transform trees, several reference pictures (RPS in the SPS or the slice header, inter RPS prediction), long-term reference pictures,
// temporal motion vector prediction, merge levels, cu_qp_delta at any quantisation-group size, chroma QP offsets, sign data
// hiding, transform skip, scaling lists (default, SPS and PPS scaling_list_data), cu_transquant_bypass (lossless coding units), PCM coding units, constrained intra prediction, deblocking offsets /
// overrides, SAO, WPP, tile grids up to the level limit of 20 columns x 22 rows (uniform or
// explicit spacing; in-loop filtering across tile and slice boundaries on or off -- Kvazaar switches it off), pictures in several slice segments: the two ways Kvazaar cuts them (a dependent slice segment per CTU row with WPP, an
// independent slice per tile) and -- one-tile pictures -- segments that begin at ANY coding tree block, independent slices (own SliceQpY) and dependent
// segments mixed (an MTU per slice, N row groups: PicJob::ctb_cut).  Random access (8.1.3, C.5.2.2): decoding may begin at a CRA picture -- its RASL pictures are
// dropped, its RADL pictures decoded --, BLA pictures and end of sequence NAL units start a coded video sequence, pic_output_flag = 0 keeps a picture in,
// no_output_of_prior_pics_flag discards what still waits (Decoder::vwait_).  Rejected with a negative return value (kvzx_decoder_last_error): several slices
// inside a tile of a picture with tiles, slices of one picture that differ in more than SliceQpY and the loop filter flag,
// > 255 slices in a picture.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "hevc_core.h"
#include "host_pool.h"
#include "dec_frame.h"
#include "dec_kernels.h"
#include "batch.h"

namespace kvzx {

// DK_HOST_PARSE is not a kernel: wall time of the host CABAC parsing stage
enum DecKernelId { DK_INTER = 0, DK_INTRA, DK_DEBLOCK, DK_HOST_PARSE, DK_SAO, DK_INTRA_P, DK_COUNT };      // (DK_INTRA_P: the intra blocks of a picture that also has inter blocks)

// short-term reference picture set (7.4.8): negative deltas first (closest first), then positive ones
struct StRps { int n_neg = 0, n_pos = 0; int dpoc[16]; uint8_t used[16]; };

struct DecSps {
  bool valid = false;
  int width = 0, height = 0;          // coded size
  int crop_r = 0, crop_b = 0, crop_l = 0, crop_t = 0;   // luma samples
  int log2_max_poc_lsb = 8;
  int num_st_rps = 0; StRps st_rps[65];
  uint32_t fps_num = 0, fps_den = 0;
  int num_reorder = 0;                // sps_max_num_reorder_pics of the highest sub-layer: pictures that may precede a picture in decoding order and follow it in output order
  int strong_intra = 0, sao = 0, tmvp = 0, amp = 0, th_depth_inter = 0, th_depth_intra = 0;
  int ctb_log2 = 6;                   // CtbLog2SizeY: 6 (every Kvazaar stream), 5 or 4 (round 6: other encoders' streams); transform blocks 4 .. min(32, CTB)
  int min_cb_log2 = 3;                // MinCbLog2SizeY: 3 (every Kvazaar stream), 4 or 5
  int num_lt_sps = -1; uint16_t lt_lsb_sps[32] = {}; uint8_t lt_used_sps[32] = {};      // long_term_ref_pics_present_flag (-1: not set): the SPS's candidates
  int pcm_depth[2] = {0, 0}, pcm_min_log2 = 0, pcm_max_log2 = 0, pcm_no_filter = 0;      // pcm_enabled_flag: PcmBitDepthY / C (0: no PCM), Log2MinIpcmCbSizeY .. Log2MaxIpcmCbSizeY, pcm_loop_filter_disabled_flag
  // scaling_list_enabled_flag: the scaling factors (dec_frame.h KVZ_SCALING_BYTES) of the SPS's lists -- the default ones (Tables 7-5 / 7-6) without
  // sps_scaling_list_data; NULL: flat.  What uvgComm's "scaling list" checkbox switches on in a peer's Kvazaar (kvazaarfilter.cpp:235-242).
  std::shared_ptr<const std::vector<uint8_t>> scaling;
};
struct DecPps {
  bool valid = false;
  int sps_id = 0;
  int sign_hiding = 0, cabac_init_present = 0, num_ref_idx_default = 1, num_ref_idx1_default = 1, init_qp = 26, tskip = 0;
  int dependent_slices = 0;
  int cu_qp_delta = 0, qp_delta_depth = 0, cb_qp_offset = 0, cr_qp_offset = 0, slice_chroma_offsets = 0;
  int weighted_pred = 0, weighted_bipred = 0, lists_mod = 0;
  int output_flag_present = 0, extra_header_bits = 0, header_extension = 0;
  int wpp = 0, tile_rows = 1, row_bd[34];   // tile row i covers CTB rows [row_bd[i], row_bd[i + 1]); filled at slice time when uniform
  int tile_cols = 1, col_bd[34];            // tile column j covers CTB columns [col_bd[j], col_bd[j + 1])
  int uniform_tiles = 1, row_height[33], col_width[33];
  int deblock_control = 0, deblock_override = 0, deblock_disabled = 0, beta_offset_div2 = 0, tc_offset_div2 = 0, loop_filter_across_slices = 1, across_tiles = 1, cip = 0;
  int par_mrg_level = 2;
  int tq_bypass = 0;                                   // transquant_bypass_enabled_flag (a peer's Kvazaar with `lossless`, kvazaarfilter.cpp:244)
  std::shared_ptr<const std::vector<uint8_t>> scaling; // pps_scaling_list_data: these factors instead of the SPS's
};

struct DecodedPicture {
  int width = 0, height = 0;          // cropped
  int coded_w = 0, coded_h = 0;
  const uint8_t *host[3] = {nullptr, nullptr, nullptr}; int host_pitch[3] = {0, 0, 0};
  const uint8_t *dev[3] = {nullptr, nullptr, nullptr}; int dev_pitch[3] = {0, 0, 0};
  int poc = 0; int64_t pts = 0; uint32_t fps_num = 0, fps_den = 0; bool is_intra = false;
  int cvs = 0, num_reorder = 0;       // coded video sequence the picture belongs to (a running count); its SPS's sps_max_num_reorder_pics
  uint64_t serial = 0;                // the picture's number in decoding order (Decoder::vwait_: which waiting pictures an IDR / BLA picture discards)
};

// a finished picture that owns its samples: what is still in the frame-threaded ring when the stream changes its resolution is
// completed and kept like this until the following calls have handed it out (the decoder's own buffers are re-sized meanwhile)
struct OwnedPic {
  DecodedPicture pic;
  std::vector<uint8_t> host;          // download mode: the three planes, pitches as in pic.host_pitch
  uint8_t *dev = nullptr;             // device-resident mode: one allocation holding the three planes
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

// motion of a decoded picture as later pictures see it for temporal prediction (8.5.3.2.8): one entry per 16x16 luma block, the
// motion of its top-left 4x4.  Written by the picture's parser row by row, read by the parsers of following pictures (which may
// run concurrently under frame threading: row_done[] orders them).
struct ColMotion {
  struct Mv { int16_t mv[2][2]; int32_t ref_poc[2]; uint8_t used; uint8_t lt; uint8_t pad[2]; };      // per list: vector, POC of the picture it points into; used: bit L = list L predicts the block (0: intra); lt: bit L = that picture was a long-term reference picture then (8.5.3.2.9)
  int w16 = 0, h16 = 0, hc = 0, poc = 0;
  std::vector<Mv> mv;
  std::unique_ptr<std::atomic<uint8_t>[]> row_done;      // per CTU row
  std::unique_ptr<std::atomic<uint8_t>[]> row_cols;      // tile columns of the row that have been parsed (row_done follows the last one)
  int cols = 1;
};

class Decoder {
 public:
  explicit Decoder(int device) : device_(device) {}
  // frame threading: up to n pictures are parsed concurrently while one more is reconstructed on the GPU; the output is delayed by n pictures
  // (libOpenHevcInit(nb_threads, OH_THREAD_FRAME / OH_THREAD_FRAMESLICE)); call before the first picture
  void set_frame_threads(int n) { if (jobs_.empty()) frame_threads_ = n < 1 ? 1 : (n > 32 ? 32 : n); }
  int frame_threads() const { return frame_threads_; }
  ~Decoder();
  bool start(std::string *error);           // checks the HIP device; no CPU fallback
  // One NAL unit (with or without start code).  <0 error/unsupported, 0 nothing to output, 1 picture ready.
  int decode_nal(const uint8_t *data, size_t len, int64_t pts);
  bool get_picture(DecodedPicture *out);   // the picture announced by the last decode_nal() == 1
  void set_download(bool on) { download_ = on; }
  // Test / measurement hook, never a decoding mode: the host half alone -- NAL units, parameter sets, slice headers and the CABAC slice-data parser run,
  // nothing is allocated on or sent to a device, no picture ever comes out (decode_nal returns 0 for every picture).  What the parser produced is
  // summarised by parse_probe_stats: pictures, transform blocks, level words, a 64-bit FNV-1a digest over every picture's records / tables / blocks /
  // levels (the bytes launch_gpu would upload, fixed part minus the descriptor) and the time spent in parse_job.  CPU tests pin the parser's output with it
  // (tests/test_parser_probe.py); tools/measure/parse_rate.py times the parser where there is no GPU.
  bool set_parse_only() { if (jobs_.empty() && !started_) parse_only_ = true; return parse_only_; }      // (false: declined -- the decoder has been started)
  struct ProbeStats { uint64_t pictures = 0, tus = 0, levels = 0, digest = 0xcbf29ce484222325ull, bins = 0; double parse_ms = 0; };
  ProbeStats parse_probe_stats() const { return probe_; }
  // device-resident output (set_download(false)): the planes handed out stay untouched while the next `n` pictures are decoded
  // (they may be read as reference pictures, never written); default 2
  void set_output_hold(int n) { output_hold_ = n < 1 ? 1 : (n > 8 ? 8 : n); }
  void set_profiling(int every) { profiling_ = every > 0; prof_every_ = every > 0 ? every : 1; }   // n: every n-th picture
  void set_parse_threads(int n) { if (!pool_) parse_threads_ = n < 1 ? 1 : n; }
  void get_kernel_times(double *ms, uint64_t *launches, bool reset);
  bool debug_copy(const char *what, void *dst, size_t bytes);
  int last_error() const { return last_error_; }
  // libOpenHevcSetCheckMD5: decoded picture hash SEI messages (MD5 or checksum) are compared with the pictures as decoded here
  void set_check_hash(bool on) { check_hash_ = on; }
  // libOpenHevcSetTemporalLayer_id: the highest temporal sub-layer that is decoded -- slice NAL units with a higher TemporalId are dropped (nothing of the layers below
  // predicts from them, 8.1.3 sub-bitstream extraction).  OpenHEVC's default is 7 (everything); uvgComm passes 0 (openhevcfilter.cpp:54), and its own sender codes one layer.
  void set_no_cropping(bool on) { no_crop_ = on; }      // libOpenHevcSetNoCropping: pictures are handed out at their coded size, the conformance window is not applied
  void set_max_temporal_id(int t) { max_tid_ = t < 0 ? 0 : (t > 7 ? 7 : t); }
  void hash_stats(int *checked, int *mismatch) const { if (checked) *checked = hash_checked_; if (mismatch) *mismatch = hash_mismatch_; }
  void flush() {}
  int pending() const { return (int)(job_head_ - job_tail_) + (gpu_job_ ? 1 : 0) + (int)gpu_q_.size(); }

  // ---- per-picture state shared with the slice-data parser (decoder.hip)
  // one per substream.  Two cache lines each: the rows of a picture are parsed side by side, every one appending to ITS vectors all the time -- the
  // vectors' headers of neighbouring rows do not share a cache line.  (What limits the row-parallel parse is WPP itself: a unit can start when its left neighbour
  // and the unit above-right are done, and a picture takes as long as the heaviest path through that graph.  The benchmark's moving objects are a few very heavy
  // units stacked over several rows -- ten of a 1080p P picture's 510 units hold a quarter of its parse -- and the heaviest path is two thirds of the whole: 1.4-1.6x
  // at best with any number of threads, tools/measure/wpp_critical_path.py, profiles/r06_wpp_critical_path.txt; seventeen rows end 55 us apart, owf0_timeline.py.)
  struct alignas(128) SubOut { std::vector<uint32_t> levels; std::vector<DecTu> tus; int rc = 0; };
  struct alignas(64) Progress { std::atomic<int> v{0}; char pad[60]; };   // one cache line per row: no false sharing between pollers
  struct SliceHdr {
    bool is_intra = false, is_b = false; int poc = 0;
    bool no_output = false;                                              // pic_output_flag = 0: decoded and kept as a reference, never handed out
    int num_ref_idx1 = 0, mvd_l1_zero = 0, collocated_from_l0 = 1;      // B slices: num_ref_idx_l1_active, mvd_l1_zero_flag, collocated_from_l0_flag
    int tmvp = 0, collocated_ref_idx = 0, sao_luma = 0, sao_chroma = 0, num_ref_idx = 1, cabac_init_flag = 0, max_merge = 5;
    int slice_qp = 26, cb_qp_offset = 0, cr_qp_offset = 0;      // offsets: PPS + slice
    int deblock_disabled = 0, beta_offset_div2 = 0, tc_offset_div2 = 0;
    uint8_t list_mod[2] = {0, 0}, list_entry[2][16] = {};      // ref_pic_lists_modification() (7.3.6.2): entries of the temporary lists (8.3.4)
    bool weighted = false; uint8_t wt_log2[2] = {0, 0}; DecWt wt[32] = {};
    uint32_t wt_explicit = 0;                                        // bit list * 16 + index: the entry's weights or offsets differ from the defaults (with the defaults the explicit formulas ARE the default ones)      // pred_weight_table() (7.3.6.3) as derived by 7.4.7.3: entry list * 16 + index
  };
  // everything one picture needs between its slice header and its reconstruction
  struct PicJob {
    std::vector<uint8_t> rbsp; size_t data_off = 0, data_len = 0;
    std::vector<size_t> sub_start;
    // slice segments (a picture in several NAL units): whether a segment ends with this CTU row (always for the last row); where, inside a
    // substream that spans several rows (no WPP), a new segment's data starts at this row (SIZE_MAX: the row continues the stream)
    std::vector<uint8_t> seg_end_row; std::vector<size_t> row_restart;
    // the picture's substreams in decoding order: CTB rows [cy0, cy1) x columns [cx0, cx1) of one tile (one row with WPP), the tile's first row and id;
    // with tile columns: whether a slice segment ends with substream k (seg_end_row is per CTU row and only used without tile columns)
    struct SubGeom { int cy0, cy1, cx0, cx1, tile_cy0, tile_cy1, tc; };
    std::vector<SubGeom> geom; std::vector<uint8_t> seg_end_sub;
    // FREE slices (round 6; one tile): slice segments that begin at any coding tree block, independent slices inside the tile -- what an encoder that cuts
    // slices by bytes or block counts sends.  Per coding tree block (raster): ctb_cut bit 0 an independent slice begins here, bit 1 a dependent segment,
    // bit 2 a segment ENDS with this block; ctb_data: where the beginning segment's bytes start in rbsp; ctb_slice: the block's slice (counted from 0; it
    // also rides in ctu_tile[] for the kernels' availability tests); slice_qps: SliceQpY by slice; ds_saved: the context states a segment left behind at the
    // end of a CTB row (WPP: the next row's parser may need them, 9.3.1).  All empty: one slice, or Kvazaar's slices (whole rows / whole tiles).
    std::vector<uint8_t> ctb_cut, ctb_slice; std::vector<size_t> ctb_data; std::vector<int8_t> slice_qps; std::vector<uint8_t> ds_saved;
    // a one-tile picture submitted because its segments COVER it row by row: whether the last one really ends with the picture is not in any header.  The parser
    // says so when it does not (DEC_SEG_ENDS_EARLY), and the synchronous decoder then takes the picture back and waits for the rest of its access unit (submit_job).
    bool ambiguous_end = false;
    // some boundary inside the picture is closed to the in-loop filters (loop_filter_across_tiles_enabled_flag = 0 with tiles; a slice with
    // slice_loop_filter_across_slices_enabled_flag = 0): lf_slices = the picture's independent slices (first block, flag)
    bool lf_restricted = false; std::vector<std::pair<int, uint8_t>> lf_slices;
    struct Undo { int poc = 0, prev_poc = 0; bool is_ref = false, used = false, seen_irap = false; long decode_idx = 0; std::shared_ptr<ColMotion> motion; std::vector<std::pair<uint64_t, int>> vwait; } undo;      // what submit_job changed
    SliceHdr sh; std::shared_ptr<const DecSps> sps; DecPps pps;  // (a later SPS / PPS NAL may replace the table entry while this picture is still being parsed: the job keeps the SPS it was coded with alive, the PPS by value)
    int64_t pts = 0; int crop[4] = {0, 0, 0, 0}; uint32_t fps_num = 0, fps_den = 0;
    int slot = 0;                                                // picture buffer this picture is reconstructed into
    int cvs = 0;                                                 // the coded video sequence it belongs to (output order)
    uint64_t serial = 0;                                         // number in decoding order
    std::vector<std::pair<int, int>> conceal;                    // (buffer, source buffer or -1): buffers that stand in for reference pictures that never arrived, filled before this picture's kernels (launch_gpu)
    bool starts_cvs = false, discard_prior = false;              // NoRaslOutputFlag (8.1.3); ... and no_output_of_prior_pics_flag in force (C.5.2.2)
    int nref = 0; int ref_poc[16]; uint8_t ref_slot[16];        // RefPicList0
    int nref1 = 0; int ref_poc1[16]; uint8_t ref_slot1[16];     // RefPicList1 (B slices)
    uint8_t ref_lt[16] = {}, ref_lt1[16] = {};                  // the entry is a long-term reference picture (8.3.2: no vector scaling, 8.5.3.2.7 / 8.5.3.2.9)
    bool no_backward = true;                                     // NoBackwardPredFlag (8.5.3.2.9): no entry of either list follows the picture in output order
    // B slices: the two-list motion of every 4x4 block as the parser's own derivations read it (merge, AMVP); what the kernels need of it goes
    // into the B4Rec (the first used list's vector and picture) and, for bi-predicted blocks (B4_BI), the second vector here
    struct MvF { int16_t mv[2][2]; int8_t ref[2]; };
    std::vector<MvF> mvf; std::vector<B4L1> b4x; std::atomic<int> any_bi{0};
    // One picture at a time (uvgComm's default "Slice" threading): a CTU row's 4x4 records go up as soon as the row is parsed, beside the parsing of the rows
    // below -- the megabyte (4K: four) of records is then on the device when the parse ends, and what submit_job uploads is the tables, transform blocks and
    // levels.  early_dst: the input buffer this picture will be launched from (nullptr: no early upload); early_rows: rows whose copy was queued.
    uint8_t *early_dst = nullptr; std::atomic<int> early_rows{0};
    std::shared_ptr<ColMotion> col, own;                         // collocated picture's motion (NULL: no temporal candidates); this picture's
    // the pinned input block (dec_frame.h) and the host views into it
    uint8_t *h_in = nullptr; size_t h_in_cap = 0; size_t ntu = 0, nlev = 0;
    B4Rec *b4 = nullptr; TuRange *region = nullptr, *ctu = nullptr; uint8_t *ctu_tile = nullptr; SaoParams *sao = nullptr;
    // host-only syntax state per 4x4: prediction mode (0 inter, 1 intra, 2 skip, 255 not decoded yet), coding quadtree depth, intra mode
    std::vector<uint8_t> pred_mode, ct_depth, intra_mode;
    std::vector<SubOut> subs; std::vector<uint8_t> wpp_saved;
    std::unique_ptr<Progress[]> row_progress; int row_progress_n = 0;
    alignas(64) bool any_intra = false; bool any_inter = false;        // (set by the rows' parsers: a line of their own, written once -- the parsers test before they store)
    alignas(64) bool across_slices = true;
    std::atomic<int> state{0}; int rc = 0; double parse_ms = 0;
    hipEvent_t done = nullptr;                                   // recorded behind the picture's last kernel
    std::atomic<int> launched{1};                                // 0 while the picture waits in the device's submission layer (batch.h): `done` has not been recorded yet
    std::vector<uint8_t> expect_hash;                            // payload of the picture's decoded picture hash SEI (libOpenHevcSetCheckMD5), empty: none
    long launch_idx = 0;                                         // count of pictures launched before this one
    hipEvent_t dl_done = nullptr; int dl_buf = -1;               // download mode: the picture's copy into host buffer dl_buf, queued behind `done` on the download stream
    struct EvPair { hipEvent_t a, b; int id; }; std::vector<EvPair> ev; size_t ev_used = 0;     // kernel timing (set_profiling)
    PicJob() {}
    PicJob(const PicJob &) {}                                  // (vector<PicJob> construction only)
  };
  // ---- tile-row split of ONE stream over several decoders, one per GPU (the decoding side of kvazzup_amd/tilesplit.py): this decoder
  // reconstructs CTU rows [row0, row0 + nrows) only -- whole tile rows of a stream whose motion vectors stay inside their tiles (what the
  // split ENCODER writes) -- and deblocking across the band's boundaries goes through two small exchanges with the neighbouring decoders.
  // Synchronous decoder only (one frame thread), no SAO, no temporal motion prediction.  Per picture: decode_nal (returns 0: the band is
  // reconstructed) -> band_export(0) -> [to rank + 1 / from rank - 1] -> band_import(0) -> band_deblock -> band_export(1) ->
  // [to rank - 1 / from rank + 1] -> band_import(1) -> band_finish (the picture is the output; its band's rows are valid).
  void set_band(int row0, int nrows) { if (jobs_.empty()) { band_row0_ = row0; band_nrows_ = nrows; } }
  size_t band_halo_bytes() const { return (size_t)pw_ * 8; }
  bool band_export(int stage, uint8_t *d_buf);     // stage 0: this band's LAST four luma rows (+ chroma) before deblocking and their 4x4 records, for the band below; stage 1: the four rows above this band, final, for the band above
  bool band_import(int stage, const uint8_t *d_buf);   // stage 0: from the band above, into the rows above this band; stage 1: from the band below, this band's last four rows, final
  bool band_ready() const { return band_nrows_ > 0 && gpu_job_ != nullptr; }     // a picture's band is reconstructed and waits for the exchange
  bool band_deblock();
  int band_finish();
  int pw() const { return pw_; }
  int ph() const { return ph_; }

 private:
  bool ensure_buffers(int w, int h, int ctb_log2);
  void free_buffers();
  int decode_slice(const uint8_t *rbsp, size_t len, int nal_type, int64_t pts);
  int hash_sei(const uint8_t *rbsp, size_t len);
  int verify_hash(const PicJob &job, const std::vector<uint8_t> &want);
  bool check_hash_ = false; int hash_checked_ = 0, hash_mismatch_ = 0;
  int parse_job(PicJob &job, bool row_parallel);
  int parse_substream(PicJob &job, int sub, const uint8_t *data, size_t len, SubOut &out);
  int close_open_picture();
  int close_free_picture(PicJob &job);
  void take_back_job(PicJob &job);
  // the slice segments of the picture being assembled, as they arrived (one tile): kept beside the row bookkeeping of Kvazaar's forms, which is dropped the moment a
  // segment turns up that those forms do not have (asm_free_) -- the picture is then put together from this list when the access unit ends
  struct FreeSeg { int address; bool dependent; int slice_qp; std::vector<size_t> subs; };
  std::vector<FreeSeg> asm_segs_; bool asm_free_ = false, free_stream_ = false; bool asm_cur_dependent_ = false; int asm_cur_qp_ = 26;
  // in-loop filtering across slice and tile boundaries: the independent slices of the picture being assembled -- first coding tree block, slice_loop_filter_across_
  // slices_enabled_flag -- in every form a picture's slices come in; build_lf_map turns them (and the PPS's tile flag) into PicJob::lf_restricted / the map the
  // parser's last step writes for the kernels (DecFrame::ctu_nb)
  struct LfSlice { int address; bool across; };
  std::vector<LfSlice> asm_lf_;
  void note_lf_restrictions(PicJob &job);
  int finish_oldest();
  void drop_pending();
  // layout of the input block
  size_t off_region() const { return (size_t)(pw_ / 4) * (ph_ / 4) * sizeof(B4Rec); }
  size_t off_ctu() const { return off_region() + (size_t)(pw_ / 32) * (ph_ / 32) * sizeof(TuRange); }
  // (the tables per coding tree block -- transform-block range, tile id, SAO parameters -- have one entry per CTB of the stream's size: ctbl_ = CtbLog2SizeY)
  size_t nctb() const { return (size_t)(pw_ >> ctbl_) * (ph_ >> ctbl_); }
  size_t off_tile() const { return off_ctu() + nctb() * sizeof(TuRange); }
  size_t off_sao() const { return (off_tile() + nctb() + 15) & ~(size_t)15; }
  size_t off_scaling() const { return (off_sao() + nctb() * sizeof(SaoParams) + 63) & ~(size_t)63; }     // KVZ_SCALING_BYTES scaling factors (pictures with scaling lists)
  size_t off_wt() const { return (off_scaling() + KVZ_SCALING_BYTES + 63) & ~(size_t)63; }       // 32 DecWt (pictures with pred_weight_table())
  size_t off_frame() const { return (off_wt() + 32 * sizeof(DecWt) + 63) & ~(size_t)63; }     // the picture's DecFrame, for launches that read it from device memory (batch.h)
  size_t off_nb() const { return (off_frame() + sizeof(DecFrame) + 15) & ~(size_t)15; }      // per CTB: the neighbouring CTBs the in-loop filters may use (DecFrame::ctu_nb; pictures with closed boundaries)
  size_t fixed_bytes() const { return (off_nb() + nctb() + 15) & ~(size_t)15; }
  bool grow_job_input(PicJob &job, size_t bytes);
  void bind_job(PicJob &job);
  int launch_gpu(PicJob &job);
  int complete_gpu(PicJob &job);
  int alloc_slot();
  void sync_main();                       // everything this decoder has submitted has run (the submission layer's queue included)
  bool batch_attached_ = false, batch_used_ = false;

  int device_; bool started_ = false;
  bool parse_only_ = false; ProbeStats probe_; void probe_book(PicJob &job, double ms);
  hipStream_t stream_up_ = nullptr; hipEvent_t up_done_[9] = {};   // upload of the next picture's input block beside the current picture's kernels
  hipStream_t stream_ = nullptr, stream_dl_ = nullptr;       // reconstruction; download of finished pictures (behind the picture's event, beside the next picture's kernels)
  std::shared_ptr<const DecSps> sps_[16]; DecPps pps_[64]; uint32_t vps_fps_num_ = 0, vps_fps_den_ = 0;
  int w_ = 0, h_ = 0, pw_ = 0, ph_ = 0; int ctbl_ = 6;      // ctbl_: CtbLog2SizeY of the active sequence (the padded size stays a multiple of 64: the kernels that tile the picture do so in 64x64 / 32x32 pieces whatever the CTB)
  std::vector<PicJob> jobs_; int frame_threads_ = 1; long job_head_ = 0, job_tail_ = 0;
  // a picture arriving in several slice segment NAL units: its job is filled segment by segment and submitted with the last one
  bool asm_active_ = false, asm_guessed_one_row_ = false; int asm_subs_ = 0, asm_rows_ = 0, asm_pps_id_ = 0, asm_nal_type_ = 0; bool asm_irap_ = false;
  int submit_job(PicJob &job, int nal_type, bool irap);
  int append_segment_tiles(PicJob &job, size_t bitpos, const uint8_t *rbsp, size_t len, const DecPps &p, const DecPps &pp, int wc, int hc, int address);
  int append_segment(PicJob &job, size_t bitpos, const uint8_t *rbsp, size_t len, const DecPps &p, const DecPps &pp, int wc, int hc, int address, int64_t pts);
  std::deque<OwnedPic> ready_q_; OwnedPic cur_owned_;       // pictures completed ahead of their turn (resolution change), the one last handed out
  // Output order (C.5.2): a stream whose SPS allows reordering (sps_max_num_reorder_pics > 0: B pictures in groups, Kvazaar gop=8) has its pictures
  // copied out as they are completed and handed on by POC -- the smallest of those waiting once more than the SPS's count wait, all of a coded
  // video sequence before the next one's first, the rest one per call when the stream ends.  A low-delay stream (the count is 0) never enters this.
  struct Waiting { OwnedPic pic; int cvs; };
  std::deque<Waiting> reorder_q_; int cvs_ = 0, reorder_ = 0;
  bool pop_reordered(bool flush);
  // Frame memory handed out by get_picture stays valid while kOutHold further NAL units are decoded -- a caller may copy it out on a stage of its
  // own (OpenHEVCFilter's output thread) -- also across a resolution change: the host output buffers and the owned pictures that such a change
  // retires are only freed kOutHold calls later.
  static constexpr int kOutHold = 8;
  long nal_calls_ = 0;
  std::deque<std::pair<long, uint8_t *>> retired_out_;      // (call count at retirement, page-locked buffer)
  std::deque<std::pair<long, OwnedPic>> retired_owned_;
  // device buffers of OwnedPic copies, kept for the next picture of the same size (queue_current_output)
  static constexpr size_t kOwnedPoolMax = 24;
  std::vector<std::pair<size_t, uint8_t *>> owned_pool_; std::map<const uint8_t *, size_t> owned_bytes_; hipEvent_t owned_ev_ = nullptr;
  uint8_t *owned_alloc(size_t bytes); void owned_release(uint8_t *p); void owned_free(uint8_t *p); bool stash_current_output();
  void free_retired(bool all);
  int decode_nal_inner(const uint8_t *data, size_t len, int64_t pts);
  bool queue_current_output();
  std::unique_ptr<FrameWorkers> workers_;
  // decoded picture buffer: slot = device planes + what reference marking needs
  struct DpbPic { uint8_t *plane[3] = {nullptr, nullptr, nullptr}; int poc = 0; bool is_ref = false, used = false, is_lt = false;      // is_lt: marked "used for long-term reference" (8.3.2)
                  long decode_idx = -1000; std::shared_ptr<ColMotion> motion;
                  hipEvent_t last_dl = nullptr;
                  hipEvent_t last_use = nullptr; bool last_use_alt = false; };      // (two chains, below: the last picture that read or wrote the buffer, and the stream it ran on)     // download mode: the copy of the picture last reconstructed here (a later picture's kernels wait for it before they write the buffer)
  DpbPic dpb_[KVZ_DEC_MAX_REFS];
  // device side
  // Frame-threaded mode keeps up to gpu_depth_ pictures queued on the GPU (launched, not yet completed): a picture's way through upload, kernels
  // and the copy back to the host is ~0.2 ms at 1080p, and with only one picture launched ahead of the one being waited for the calling thread
  // sat out most of that for every picture.  The output lag stays `frame_threads` pictures: the parse ring gives up what the GPU queue takes.
  static constexpr int kMaxGpuDepth = 8;
  int gpu_depth_ = 1; std::deque<PicJob *> gpu_q_;
  uint8_t *d_in_[kMaxGpuDepth + 1] = {}; size_t d_in_cap_[kMaxGpuDepth + 1] = {};   // device copies of PicJob::h_in: the pictures in flight take turns
  int16_t *resid_[3] = {nullptr, nullptr, nullptr};         // intra residuals between k_dec_intra_resid and k_dec_intra
  uint8_t *work_[3] = {nullptr, nullptr, nullptr};          // pictures with SAO: reconstruction and deblocking happen here, the filter writes into the slot
  uint32_t *edge_col_ = nullptr; unsigned long long *edge_row_ = nullptr; uint32_t chain_gen_ = 0;      // k_dec_intra's tagged hand-off words (dec_frame.h), the generation of the last launch
  uint32_t *progress_ = nullptr, *intra_order_ = nullptr, *err_ = nullptr; uint32_t *h_err_ = nullptr;
  // A picture without inter blocks depends on no other picture, and its chain (k_dec_intra: 0.55 ms at 1080p) keeps a few dozen compute units busy: with the
  // frame-threaded decoder such pictures ALTERNATE between the decoder's stream and a second one with chain arrays of its own, so that two chains run side by
  // side (an all-intra stream, BASELINE configs[0]: the decoder's rate was 1 / chain).  Pictures that read or overwrite a buffer last used on the other stream
  // wait for that picture's event.
  hipStream_t stream_alt_ = nullptr; char alt_prio_ = 'n'; long intra_seq_ = 0; bool alt_failed_ = false;      // (stream_alt_ is set LAST by ensure_alt: non-NULL = every array of the second chain exists)
  uint32_t *progress_alt_ = nullptr, *edge_col_alt_ = nullptr; unsigned long long *edge_row_alt_ = nullptr; int16_t *resid_alt_[3] = {nullptr, nullptr, nullptr}; uint8_t *work_alt_[3] = {nullptr, nullptr, nullptr};
  bool ensure_alt();
  // download mode: page-locked output buffers take turns -- one is what libOpenHevcGetOutput last handed out (valid until the next
  // decode call, openhevcfilter.cpp:218-229 copies at once), one receives the picture whose kernels are running, queued behind them on
  // the download stream at launch, so the copy over PCIe overlaps the next picture's kernels instead of stalling the calling thread
  static constexpr int kOutRing = 16;     // (pictures queued on the GPU + the one handed out + the one being launched + six the caller may still be copying out of: OpenHEVCFilter's output stage)
  uint8_t *h_out_[kOutRing] = {}; size_t h_out_cap_ = 0;
  void describe_output(const PicJob &job, DecodedPicture &o, int buf) const;
  int queue_download(PicJob &job);
  int start_ready_downloads();
  long launched_ = 0; int out_slot_ = 0; int output_hold_ = 2;
  char prio_dl_ = 'n', prio_up_ = 'n';
  char prio_ = 'n';                                       // priority level of the main stream (stream_pool.h key)
  int band_row0_ = 0, band_nrows_ = 0; uint8_t *band_din_ = nullptr; DecFrame band_f_{};      // band mode: the picture between its reconstruction and band_finish
  double t_parse_max_ = 0;                // trace: the longest parse of one picture (an IDR), ms
  bool spin_wait_ = false;                // KVAZZUP_AMD_SPIN: poll the GPU instead of napping between queries
  PicJob *gpu_job_ = nullptr;             // picture whose kernels are in flight (frame-threaded mode)
  int prev_poc_ = 0, cur_tid_ = 0; bool seen_irap_ = false;
  int max_tid_ = 7; bool no_crop_ = false;
  bool after_eos_ = false;       // an end of sequence / end of bitstream NAL unit came: the next picture starts a coded video sequence (a CRA picture then has NoRaslOutputFlag = 1, 8.1.3)
  // C.5.2.2's bookkeeping in DECODING order, kept at submit time: the pictures that are "needed for output" and have not had their turn -- (serial, POC).  The pictures
  // themselves complete later (frame threads) and leave through reorder_q_ by the same counting rule; this list exists so that an IDR / BLA picture with
  // no_output_of_prior_pics_flag discards exactly the pictures the standard's process would still hold at that instant, whatever the threads' timing.
  std::vector<std::pair<uint64_t, int>> vwait_; std::vector<uint64_t> discarded_; uint64_t pic_serial_ = 0;
  // Concealment v2: a picture the reference picture set says the current one predicts from is not there (its access unit was lost on the way): a buffer with its
  // picture order count stands in -- no motion, never output -- and decoding goes on.  Its samples: a copy of the reference picture nearest in output order among
  // those the DPB held when the current picture arrived (of two equally near the earlier one), mid-grey when there is none (libavcodec's generate_missing_ref always takes grey).
  // The checker follows the same rule (oracle/hevc_dec.c missing_ref).  The buffers wait here for the picture that needed them; its launch fills them.
  int conceal_ref(int poc, bool is_lt); std::vector<std::pair<int, int>> pending_conceal_; uint64_t concealed_ = 0;
  bool cur_discard_ = false;     // the picture whose headers are being read empties the buffer without output (decided with its first segment)
  bool cur_no_rasl_ = false;     // NoRaslOutputFlag of the picture whose slice headers are being read (decided with its first segment)
  bool skip_rasl_ = false;       // NoRaslOutputFlag of the last IRAP picture: the RASL pictures that belong to it refer to pictures that are not there -- their NAL units are dropped
  bool download_ = true, profiling_ = false, prof_now_ = false; int prof_every_ = 1;
  bool pic_ready_ = false; DecodedPicture out_;
  int last_error_ = 0;
  double t_nal_ = 0, t_wait_ = 0, t_stage_ = 0, t_api_ = 0, t_sync_ = 0;     // decoder-thread time split (KVAZZUP_AMD_TRACE)
  std::vector<uint8_t> rbsp_;
  std::vector<size_t> epb_;                // unescaped payload offset of every removed emulation prevention byte
  std::vector<size_t> sub_start_;          // start of every substream inside the unescaped slice data
  std::unique_ptr<OrderedPool> pool_; int parse_threads_ = 16; std::mutex pool_mutex_;   // (frame workers: one picture at a time on the row pool)
  PicJob *timed_job_ = nullptr;
  double k_ms_[DK_COUNT] = {0}; uint64_t k_n_[DK_COUNT] = {0};
  template <class F> void timed(int id, F &&launch, hipStream_t st = nullptr);
};

}  // namespace kvzx
