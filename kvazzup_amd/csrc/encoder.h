// kvazzup_amd/csrc/encoder.h -- host engine of the HIP encoder: owns the device buffers and the
// HIP stream, runs the per-picture kernel pipeline and assembles the access unit.
// This is what sits behind kvz_api->encoder_open / encoder_encode / encoder_close
// (/root/reference/src/media/processing/kvazaarfilter.cpp:291,435-448,317).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "hevc_core.h"
#include "hevc_headers.h"
#include "entropy_host.h"
#include "enc_kernels.h"

namespace kvzx {

// K_HOST_ARITH is not a kernel: wall time of the host arithmetic-coding stage (entropy_host.h)
enum KernelId { K_PAD = 0, K_ME, K_INTER_RECON, K_INTER_SIGNAL, K_INTRA_ANALYSE, K_INTRA_RECON, K_DEBLOCK, K_TOKENIZE, K_HOST_ARITH, K_SAO, K_TOK_COMPACT, K_CABAC_ROWS, K_SUBPEL, K_INTRA_ANALYSE_P, K_INTRA_RECON_P, K_COUNT };      // (.._P: the intra units of a P picture, intra-in-p)

struct EncoderConfig {
  int width = 0, height = 0;
  int qp = 32, intra_period = 64, vps_period = 1;
  int me_range = 16;
  int fps_num = 30, fps_den = 1;
  int wpp = 1, deblock = 1;
  // Tile-row split of one picture over several encoder instances (one per GPU): this instance codes CTU
  // rows [band_row0, band_row0 + band_rows) only -- whole tiles -- through band_phase1 / halo exchange / band_phase2.
  int band_row0 = 0, band_rows = 0;   // band_rows == 0: the whole picture (normal operation)
  int tile_rows = 1;          // tile rows (kvazaar "tiles" CxR: R), uniform spacing, loop filter across tiles on
  int tile_cols = 1;          // tile columns (C); band mode needs 1
  int device = 0;
  int entropy_threads = 16;   // host threads of the arithmetic-coding stage
  int qp_in_cu = 0;           // kvazaar "set-qp-in-cu": cu_qp_delta_enabled_flag; a delta-QP map (set_roi, kvz_picture.roi) then gives every CTU its own QP
  int bitrate = 0;            // bits per second; 0 = constant QP, > 0 = "uvgx rate control v1" (oracle/hevc_enc.c rate_control())
  int slices = 0;             // kvazaar "slices": 1 = wpp (a dependent slice segment per CTU row; needs wpp), 2 = tiles (an independent slice per tile); one NAL unit per segment
  int lossless = 0;           // kvz_config.lossless: cu_transquant_bypass in every coding unit (no deblocking, SAO, RDOQ, sign hiding or rate control with it)
  int rc_bands = 0;           // with bitrate > 0: "uvgx rate control v2" -- a P picture's CTU rows are reconstructed in this many groups and the QP follows the level cost
                              // between them, on the device (rc_kernels.hip; statement rc_band_decide() in oracle/hevc_enc.c); 0 = picture level only (v1); implies qp_in_cu
  int satd = 1;               // intra mode search cost: 8x8 Hadamard sums (1) or SAD (0)
  int subme = 0;              // kvazaar "subme" 0..4: fractional-sample refinement of the searched vectors, "uvgx subme v1" (oracle/hevc_enc.c subme_refine())
  int me_early = 1;           // kvazaar "me-early-termination" (on / sensitive: 1, off: 0): static 32x32 blocks skip the motion search
  int vaq = 0;                // kvazaar "vaq" 1..20: "uvgx VAQ v1" (oracle/hevc_enc.c vaq_deltas()); implies qp_in_cu
  int mv_frame = 0;           // kvazaar "mv-constraint" frame / frametile (1), frametilemargin (2): vectors keep the block inside the picture
  int sao = 0;                // kvazaar "sao": sample adaptive offset, parameters by "uvgx SAO decision v1" (oracle/hevc_sao.c)
  int input_hold = 0;         // "input-hold" (extension): 1 = the caller leaves a DEVICE input picture unchanged until that picture's access unit has been returned --
                              // what the kvz_api contract already demands of host pictures (kvazaarfilter.cpp:76-88); encode_device then returns without waiting for the input stage
  int intra_chain = 1;        // "intra-chain": left-edge blocks / the above-right corner block of a CTU keep to the modes that do not read the neighbouring CTU's below-left / above-right samples (oracle/hevc_enc.c intra_analyse_size)
  int scaling_list = 0;       // kvazaar "scaling-list default": scaling_list_enabled_flag with the default lists (oracle/hevc_scaling.c); quantiser scale per position (qscale << 4) / m
  int rdoq = 0;               // kvazaar "rdoq": "uvgx RDOQ v1" -- sparse high-frequency coefficient groups are dropped when that is cheaper (oracle/hevc_transform.h orc_adjust_levels)
  int intra_in_p = 0;         // "intra-in-p" 0 / 1 (16x16 units only) / 2 (16x16 and 8x8): intra coding units in P pictures ("uvgx intra-in-P v1", oracle/hevc_enc.c me_block32); not in band mode
  int me_source = 0;          // "me-source" ("uvgx search pipelining v1", oracle/hevc_enc.c build_refpad): the integer search looks at the previous input picture; k_me (and k_intra_analyse<P>) of picture t + 1 then run on the input stream beside picture t's chain; not in band mode
  int signhide = 0;           // kvazaar "signhide": sign_data_hiding_enabled_flag; the quantiser makes the parity of every eligible coefficient group say the hidden sign
  int hash = 0;               // kvazaar "hash": 1 checksum, 2 md5 -- a decoded picture hash SEI (D.2.19) behind every picture's slices, from the reconstruction downloaded for it
  int entropy_gpu = 0;        // arithmetic coder: 1 = on the GPU (k_cabac_rows, cabac_kernels.hip), 0 = host thread pool (entropy_host.h); band mode always uses the host pool
  int owf = 0;                // kvazaar "owf": 0 = encode() returns its own picture; 1 = output lags one picture and the host
                              // coding of picture t overlaps the kernels of t + 1; >= 2 = output lags two pictures and the host
                              // coding runs on a background thread, so the calling thread only launches kernels
};

struct EncodedPicture {
  bool valid = false;         // false: nothing was output by this call (pipeline filling, owf >= 1)
  std::vector<uint8_t> au;
  int poc = 0, qp = 0; bool is_intra = false;
  uint64_t bins = 0;
  bool recon_delivered = false;   // the reconstruction has been copied into the planes given to set_recon_sink
};

class Encoder {
 public:
  static Encoder *create(const EncoderConfig &cfg, std::string *error);
  ~Encoder();
  // picture as three host planes (stride = width).  pinned: the three planes lie back to back in page-locked memory (a kvz_picture from
  // picture_alloc) that the caller leaves alone until this picture's access unit has been returned (the kvz_api contract,
  // kvazaarfilter.cpp:76-88): the upload then reads the caller's picture itself, without a staging copy
  bool encode_host(const uint8_t *y, const uint8_t *u, const uint8_t *v, EncodedPicture *out, bool pinned = false);
  // picture as packed I420 in device memory (w*h*3/2 bytes)
  bool encode_device(const uint8_t *d_i420, EncodedPicture *out);
  // owf >= 1: outputs the picture still in flight, if any (kvz_api encoder_encode with pic_in == NULL)
  bool flush(EncodedPicture *out);
  // like flush, but never waits: outputs the oldest picture in flight only if it has already been finished
  bool poll(EncodedPicture *out);
  // delta-QP map for the following pictures (kvz_picture.roi, kvazaarfilter.cpp:423-431): w x h int8 cells spread uniformly over
  // the picture, clamped to [-12, 12]; w == 0 removes it.  Needs cfg.qp_in_cu.  Statement: roi_targets() in oracle/hevc_enc.c.
  void set_roi(int w, int h, const int8_t *map);
  // ---- band mode (cfg.band_rows > 0); every call is synchronous.  Per picture: band_phase1, export the halos, exchange them
  // with the neighbouring bands' encoders (rank - 1 gets `up`, rank + 1 gets `down`), import theirs, band_phase2.
  void band_report_au(long picture, uint32_t bytes);              // rate control in band mode: size of the assembled access unit of picture `picture` (every band's encoder is told; needed before picture + 3 starts)
  bool band_phase1(const uint8_t *d_i420);                       // input, decisions, reconstruction, vertical-edge deblocking of the band
  size_t halo_bytes() const;                                     // size of one halo block
  bool band_export_halo(uint8_t *d_up, uint8_t *d_down);         // the band's first / last 4 luma + 2 x 2 chroma rows (vertical edges filtered) and CU records of its first / last 8x8 row
  bool band_import_halo(const uint8_t *d_from_up, const uint8_t *d_from_down);   // nullptr: no neighbour on that side
  // horizontal-edge deblocking (boundary edges included), tokenizer, arithmetic coding: one substream per CTU row (WPP) or tile of the band
  bool band_phase2(std::vector<std::vector<uint8_t>> *substreams, EncodedPicture *info);      // = band_phase2a + band_phase2b
  bool band_phase2a();                                            // what needs no halo: inner horizontal edges, tokenizer, arithmetic coder (may run beside the exchange, BEFORE band_import_halo)
  bool band_phase2b(std::vector<std::vector<uint8_t>> *substreams, EncodedPicture *info);     // after band_import_halo: the band's two boundary edges; hands out the substreams
  int rc_delay() const { return rc_delay_; }
  int pending() const { return (int)(accepted_ - collected_); }
  // pictures taken from the caller / pictures whose turn to be returned has come (also when collecting one FAILED): what the C ABI pairs its queues of source and
  // reconstruction pictures with (kvz_api.hip encoder_encode)
  long accepted_count() const { return accepted_; }
  long collected_count() const { return collected_; }
  // cropped reconstruction of the last coded picture -> host planes (stride = width)
  bool download_recon(uint8_t *y, uint8_t *u, uint8_t *v);
  // Where the reconstruction of the NEXT picture handed to encode_host goes (page-locked planes, width x height dense; kvz_api's pic_out): the copy is
  // queued behind the picture's chain when the picture is submitted and runs beside its tokenizer and arithmetic coder -- collect() returns when both
  // are done -- instead of as a synchronous download after the access unit is finished (0.13 ms of a 0.52 ms encoding delay at 1080p, 0.5 of 4.7 at 4K
  // with uvgComm's default OWF 0).  Call before encode_host; nullptr: no copy.
  void set_recon_sink(uint8_t *y, uint8_t *u, uint8_t *v) { sink_[0] = y; sink_[1] = u; sink_[2] = v; }
  // debug: copy an internal device array of the last coded picture to the host
  //   "cu_log2","cu_intra","cu_flags","cu_merge_idx","cu_mvp_idx","cu_intra_mode","cu_cbf" (b8 bytes),
  //   "cu_mv" (b8 * 2 int16), "coef0..2" (int16 planes), "rec0..2" (coded planes), "src0..2"
  bool debug_copy(const char *what, void *dst, size_t bytes);
  int coded_width() const { return cw_; }
  int coded_height() const { return ch_; }
  const EncoderConfig &config() const { return cfg_; }
  // every = 0: off; 1: every picture; n: every n-th picture (sampling keeps the event overhead out of the throughput)
  void set_profiling(int every) { profiling_ = every > 0; prof_every_ = every > 0 ? every : 1; }
  // accumulated kernel time (ms) and launch count per KernelId since the last reset
  void get_kernel_times(double *ms, uint64_t *launches, bool reset);
  const uint8_t *device_recon(int plane) const { return rec_[out_idx_][plane]; }   // picture last output
  hipStream_t stream() const { return stream_; }

 private:
  Encoder() {}
  bool init(const EncoderConfig &cfg, std::string *error);
  bool submit(const uint8_t *d_i420, int in_ring);   // in_ring >= 0: the picture is being uploaded into d_in_[in_ring] (encode_host)
  bool collect(EncodedPicture *out);
  struct Slot;
  bool finish_slot(Slot &sl, EncodedPicture *out, int worker = 0);   // wait for the slot's kernels, arithmetic coding, access unit
  void background(int worker);
  void timed(KernelId id, hipStream_t st, const std::function<void()> &launch);

  EncoderConfig cfg_;
  int cw_ = 0, ch_ = 0, rows_ = 0;
  hipStream_t stream_ = nullptr;
  EncFrame f_{};
  // Host pictures (kvz_api->encoder_encode, kvazaarfilter.cpp:435-438): a ring of packed device buffers filled by the copy engine on a
  // stream of its own -- picture t + 1 travels over PCIe while the kernels of picture t run -- and, for callers whose planes are not
  // page-locked, a ring of pinned staging buffers (allocated on first use).  Twelve entries: more than the pictures that can be in flight
  // (owf <= 8 plus the submitter's hand), so a copy never waits in the copy engine's queue for its buffer's previous reader.
  static constexpr int kInRing = 20;
  uint8_t *d_in_[kInRing] = {};          // packed input (device)
  uint8_t *h_in_[kInRing] = {};          // pinned host staging
  hipStream_t stream_h2d_ = nullptr;
  hipEvent_t ev_h2d_[kInRing] = {}, ev_pad_[kInRing] = {}; bool pad_pending_[kInRing] = {}, h2d_pending_[kInRing] = {};
  long in_count_ = 0;
  hipStream_t stream_rec_ = nullptr;     // download of reconstructions the caller asks for (encoder_encode's pic_out)
  uint32_t *probe_words_ = nullptr;      // KVAZZUP_AMD_PARSE_PROBE: {wrong bins, bins} of k_cabac_decode_probe
  // Per-picture working sets (padded source planes, level planes, CU arrays, per-CTU QP arrays, SAO parameters): kSets of them take turns, so the
  // host can queue kSets - 1 pictures' kernels ahead of the one the GPU is working on without waiting for a set to come free (with two sets
  // the input stage of picture t waited for the reconstruction of t - 2, and the calling thread with it: the main stream ran dry between pictures)
  static constexpr int kSets = 8;
  uint8_t *src_[kSets][3] = {};         // padded source planes
  // reconstruction ring: the picture being coded, its reference, and (owf >= 2) the one still waiting to be output
  static constexpr int kMaxDepth = 16;    // pictures in flight behind the one being submitted (owf), at most
  uint8_t *rec_[kMaxDepth + 4][3] = {};
  bool spin_wait_ = false;      // KVAZZUP_AMD_SPIN: poll the GPU instead of napping between queries
  int rc_delay_ = 3;            // rate control: pictures between a picture and the access unit size booked before it (3 .. 7)
  int nrec_ = 3;                // reconstruction ring: the picture being written, its reference, and the ones whose output is still owed (owf)
  int cur_idx_ = 0, ref_idx_ = 2, out_idx_ = 2;
  // Two sets of everything the tokenizer reads (levels and CU records): picture t is tokenised on the second stream
  // from set t & 1 while the kernels of picture t + 1 fill the other one.
  int16_t *coef_[kSets][3] = {};
  uint8_t *cu_bytes_[kSets] = {};          // 7 byte arrays back to back
  int16_t *cu_mv_[kSets] = {}, *cu_mvd_[kSets] = {};
  int set_ = 0, out_set_ = 0;
  char prio_[3] = {'h', 'n', 'n'};                      // priority levels of the main, tokenizer and input streams (stream_pool.h keys)
  std::vector<int8_t> roi_; int roi_w_ = 0, roi_h_ = 0;           // as set by the caller (set_roi)
  std::vector<int8_t> roi_sub_; int roi_sub_w_ = 0, roi_sub_h_ = 0;   // the map of the picture being submitted (it travels with the picture to the submitter thread)
  int8_t *ctu_qt_[kSets] = {}, *ctu_qy_[kSets] = {}, *ctu_delta_[kSets] = {}; uint8_t *ctu_first_[kSets] = {};   // per set
  int8_t *h_ctu_qt_[kSets] = {};   // pinned staging of the picture's ROI deltas, one per CTU
  int8_t *ctu_roi_[kSets] = {};    // ... and their device copies (uploaded on the input stream when the picture brings a map)
  bool stage_roi(hipStream_t st);  // the picture's ROI deltas -> ctu_roi_[set_] on `st`; roi_dev_ = that array or NULL (no map)
  const int8_t *roi_dev_ = nullptr;
  int *vaq_act_ = nullptr, *vaq_sum_ = nullptr; // VAQ: activity of every CTU, its sum over the picture
  uint32_t next_chain_gen();
  bool picture_begin(hipStream_t qt_stream, EncFrame *fold = nullptr, bool zero = false);     // fold: a P picture without VAQ -- the work rides in the picture's k_me launch (EncFrame::pb_*) instead of a launch of its own   // launch_picture_begin: the rate control state on the main stream (picture order), the per-CTU targets on the picture's own stream (+ the VAQ kernels)
  int qp_cur_ = 0; int64_t rc_debt_ = 0; uint32_t rc_bytes_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint32_t rc_known_ = 0;   // rate control state (calling thread)
  void rate_control();
  RcState *rc_state_ = nullptr;                 // rate control v2: device-side state
  bool band_picture_setup();
  bool band_intra_ = false;
  hipStream_t stream_tok_ = nullptr;     // signalling decisions, tokenizer, compaction
  hipStream_t stream_in_ = nullptr;      // input padding (runs ahead of the previous picture's kernels)
  hipEvent_t ev_src_free_[kSets] = {}; bool src_busy_[kSets] = {};
  uint8_t *work_[3] = {nullptr, nullptr, nullptr};   // SAO on: the picture up to deblocking (rec_[] then holds the filtered pictures)
  SaoParams *sao_[kSets] = {};            // per CTU, one array per set

  // An intra picture depends on no other picture: its chain (1.6 ms at 1080p, twenty picture intervals) is queued on a stream of its own the moment the
  // picture is accepted, beside the P pictures in front of it that the main stream is still working through; the next P picture waits for ev_idr_done_.
  hipStream_t stream_idr_ = nullptr; hipEvent_t ev_idr_done_ = nullptr; bool idr_pending_ = false, idr_side_ = false, all_intra_alt_ = false;
  // ... with its own copies of what the P pictures' kernels also use while it runs beside them (round 4: the side chain also with SAO, intra units in P
  // pictures, per-CTU QPs and rate control v2 -- uvgComm's default mode): progress counters + ticket word, the CTUs' edge columns, the SAO work picture
  uint32_t *sync_idr_ = nullptr; uint32_t *edge_col_idr_ = nullptr; uint32_t chain_gen_ = 0; unsigned long long *edge_row_ = nullptr, *edge_row_idr_ = nullptr; uint8_t *work_idr_[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_signalled_ = nullptr, ev_tok_done_[kSets] = {}; bool tok_pending_[kSets] = {};
  void bind_set(int k);
  uint8_t *intra_scratch_ = nullptr;
  uint8_t *d_scaling_ = nullptr;          // scaling-list default: KVZ_SCALING_BYTES scaling factors
  uint16_t *tok_buf_ = nullptr; int tok_cap_ = 0; int32_t *tok_count_ = nullptr; uint32_t *tok_seg_ = nullptr; uint32_t *tok_list_ = nullptr; int tok_nctu_ = 0;
  size_t tok_dense_cap_ = 0;
  std::vector<std::vector<uint8_t>> band_subs_; uint64_t band_bins_ = 0; bool band_coded_ = false;   // band mode: between phase 2a and 2b
  // me-source: k_me and k_intra_analyse<P> of the pictures ahead run on the input stream while the main stream is still in an earlier picture's chain, so what
  // they write -- the 16x16 costs, the candidate list, the quarters' scratch, the P pictures' progress counters with the "has intra units" word -- exists per working set
  bool me_ahead_ = false; uint32_t *me_block_[kSets] = {}; uint32_t *sync_set_[kSets] = {}; int prev_set_ = 0;
  uint32_t *sync_ = nullptr; uint32_t *me_cost16_ = nullptr; uint32_t *edge_col_ = nullptr; uint32_t *intra_order_ = nullptr; uint32_t *err_ = nullptr; unsigned long long *trace_ = nullptr;
  // one slot per picture in flight: the host-visible results of its kernels and what collect() needs to finish it
  struct EvPair { hipEvent_t a, b; KernelId id; };
  struct Slot {
    uint16_t *h_tok_dense = nullptr, *d_tok_dense = nullptr; int32_t *h_tok_count = nullptr, *d_tok_count = nullptr;   // host-mapped pinned
    uint32_t *h_tok_off = nullptr, *d_tok_off = nullptr;
    uint32_t *h_err = nullptr, *d_err = nullptr;
    // GPU arithmetic coder (cfg.entropy_gpu): dense tokens stay in device memory (d_tok_dense .. d_tok_off point there), the coder runs on the slot's own
    // stream -- a substream is a ~0.1-1 ms serial chain, so several pictures' coders must be able to run side by side
    uint16_t *g_tok = nullptr; int32_t *g_count = nullptr; uint32_t *g_off = nullptr;
    uint8_t *g_stage = nullptr; uint32_t *g_cursors = nullptr, *g_ctx_save = nullptr, *g_ctx_ready = nullptr;
    uint8_t *h_out = nullptr, *d_out = nullptr; uint32_t *h_sub = nullptr, *d_sub = nullptr;   // host-mapped: substream bytes; [3][nsub] offset, length, bins
    hipStream_t ent_stream = nullptr; hipEvent_t tok_ev = nullptr; uint32_t gen = 0;
    hipEvent_t done = nullptr, rec_done = nullptr;       // tokens / substreams delivered (stream_tok_ / ent_stream) / reconstruction final (stream_)
    hipEvent_t sink_done = nullptr; bool has_sink = false; // set_recon_sink: the reconstruction's copy into the caller's picture (stream_rec_)
    int poc = 0, rec_idx = 0, set = 0, qp = 0; bool intra = false, write_ps = false; long pic_idx = 0;
    std::vector<EvPair> ev; size_t ev_used = 0;
    EncodedPicture result; bool ready = false, ok = true;   // owf >= 2: filled by the background thread
    EncFrame f_tok{}; bool prof = false, tok_failed = false;                    // tok_deferred_: what the tokenizer's launches need, kept until the launcher thread makes them
  };
  Slot slot_[kMaxDepth + 2]; Slot *cur_slot_ = nullptr; int nslots_ = 1, depth_ = 0;
  size_t stage_cap_ = 0, out_cap_ = 0;
  std::thread bg_[2]; std::mutex bm_; std::condition_variable bcv_; std::deque<int> bq_; bool bquit_ = false;
  // SAO (owf 2..7): the tokenizer of a picture needs the filter's parameters, i.e. the END of the picture's chain.  An event on the main stream that another
  // stream is already waiting for costs the stream's next kernel -- the next picture's k_me -- 12-25 us; one that only the host looks at costs 1-2 us
  // (tools/measure/record_after_writer.hip).  So a thread of its own watches the chains' events and launches a picture's tokenizer when its chain is done.
  bool tok_deferred_ = false; std::thread tok_thread_; std::mutex tm_; std::condition_variable tcv_; std::deque<int> tq_; bool tquit_ = false;
  void tok_launcher();
  bool launch_tokenizer(Slot &sl, const EncFrame &f, bool intra, bool wait_on_stream, bool prof);
  void timed_slot(Slot &sl, bool prof, KernelId id, hipStream_t st, const std::function<void()> &launch);
  std::mutex stat_m_;
  long submitted_ = 0, collected_ = 0, accepted_ = 0;    // pictures whose kernels have been queued / whose access unit has been returned / taken from the caller
  // owf >= 2: the output lags anyway, so the calling thread (uvgComm's encoder filter thread) only hands the picture over; a submitter thread makes the
  // ~20 HIP calls that queue its copy and kernels (~0.1 ms per picture, which at 1080p was most of what that thread had time for).  Applies where the
  // input stays valid without the call waiting for it: page-locked host pictures (contract) and device pictures with input-hold.
  struct SubmitJob { const uint8_t *src = nullptr; bool host = false; std::vector<int8_t> roi; int roi_w = 0, roi_h = 0; int slot = 0; uint8_t *sink[3] = {nullptr, nullptr, nullptr}; };
  uint8_t *sink_[3] = {nullptr, nullptr, nullptr}, *sink_sub_[3] = {nullptr, nullptr, nullptr};      // as set by the caller / of the picture being submitted
  std::thread sub_thread_; std::mutex sm_; std::condition_variable scv_; std::deque<SubmitJob> sq_; bool squit_ = false, sbusy_ = false;
  void submitter();
  void drain_submitter();
  bool enqueue(const uint8_t *src, bool host, EncodedPicture *out);
  bool upload_and_submit(const uint8_t *y, const uint8_t *u, const uint8_t *v, bool pinned);
  hipEvent_t in_done_ = nullptr; bool in_pending_ = false;   // input picture consumed (staging buffer / caller's device buffer reusable)
  // owf >= 2: two background workers, each with its own coder pool, finish two pictures side by side (the substreams of one
  // picture start one after the other -- WPP context hand-over -- so one picture alone cannot keep the pool busy)
  EntropyHost *entropy_ = nullptr, *entropy2_ = nullptr;
  std::vector<std::vector<uint8_t>> rows_out_, rows_out2_;
  int frame_idx_ = 0, poc_ = 0, intra_count_ = 0;
  bool profiling_ = false, prof_now_ = false; int prof_every_ = 1;
  double t_submit_ = 0, t_wait_ = 0, t_arith_ = 0, t_asm_ = 0, t_in_ = 0;   // encoder-thread time split (KVAZZUP_AMD_TRACE)
  double k_ms_[K_COUNT] = {0}; uint64_t k_n_[K_COUNT] = {0};
  StreamParams sp_{};
};

}  // namespace kvzx
