// kvazzup_amd/csrc/batch.h -- the process-wide submission layer of the decoders of one device.
//
// uvgComm's multi-party topology is one OpenHEVCFilter per peer in ONE process (/root/reference/src/media/processing/filtergraph.cpp:561-589),
// each on its own filter thread.  A 1080p picture is 15-40 us of kernel time in launches of 500-2000 workgroups: one picture per launch leaves
// most of the 256 compute units idle and pays the ~6 us launch floor per picture and kernel.  So when more than one decoder is open on a
// device, their pictures are not launched by their own threads: each thread uploads its picture's input block (with the DecFrame descriptor
// inside it) and posts the picture here; one submitter thread per device launches the same kernel for ALL pictures that are waiting --
// grid = the sum of their workgroups, the kernel argument a table of descriptor pointers (dec_frame.h DecBatch) -- on the device's one decoder
// stream (stream_pool.h role 'D'), and records every picture's own completion event behind the batch.
//
// Batching policy: no picture is ever held back to wait for company.  The submitter launches whatever is queued as soon as fewer than two
// batches are in flight on the stream; while two are in flight (the GPU is the busy side) arrivals accumulate and leave together.  At low load
// a picture is launched alone at once; as the load on the GPU grows, so do the batches.  Two pictures of the same decoder are never in one
// batch (the later one may predict from the earlier one); per decoder the launch order is the submission order.
//
// With one decoder open on the device nothing of this is used: Decoder::launch_gpu launches its kernels itself, frame by value, as before.
// KVAZZUP_AMD_BATCH=0 switches the layer off (every decoder launches for itself on the shared stream).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include "dec_frame.h"

namespace kvzx {

enum BatchKernelId { BK_INTER = 0, BK_INTRA, BK_DEBLOCK, BK_SAO, BK_COUNT };

struct DecBatchItem {
  DecFrame f;                            // host copy (grid geometry)
  const DecFrame *d_f = nullptr;         // the descriptor in the picture's input block, device memory
  bool inter = false, intra = false, deblock = false, sao = false;
  hipEvent_t wait0 = nullptr, wait1 = nullptr;      // the decoder stream waits for these first: the input block's upload; the picture buffer's last download
  hipEvent_t done = nullptr;             // recorded behind the picture's last kernel
  std::atomic<int> *launched = nullptr;  // becomes 1 once `done` has been recorded (an event that was never recorded reads as finished)
  const void *owner = nullptr;
  bool profile = false;                  // time this picture's batch with events
};

struct BatchStats {
  uint64_t batches = 0, pictures = 0;                // launched batches and the pictures in them
  uint64_t by_size[KVZ_DEC_BATCH_MAX + 1] = {};      // batches of n pictures
  double ms[BK_COUNT] = {};                          // profiled batches: kernel time, launches and pictures per kernel
  uint64_t launches[BK_COUNT] = {}, frames[BK_COUNT] = {};
};

class DecBatcher {
 public:
  static DecBatcher &get(int device);
  void attach(hipStream_t st);           // a decoder opens on the device (st: the device's decoder stream)
  void detach();
  bool active() const { return enabled_ && users_.load(std::memory_order_relaxed) > 1; }
  void submit(const DecBatchItem &it);
  void drain(const void *owner);         // returns when nothing of `owner` is waiting to be launched
  void get_stats(BatchStats *out, bool reset);
  void hold(bool on);                    // measurement aid: while held nothing is launched (pictures pile up), release launches them in full batches

 private:
  explicit DecBatcher(int device) : device_(device) {}
  void run();
  void launch(DecBatchItem *items, int n);
  int device_;
  bool enabled_ = true, configured_ = true;    // configured_: KVAZZUP_AMD_BATCH and shared role streams; enabled_: that, and every attached decoder on one stream
  std::mutex life_;                            // attach / detach, whole calls
  std::atomic<int> users_{0};
  hipStream_t stream_ = nullptr;
  std::mutex m_; std::condition_variable cv_, idle_cv_;
  std::deque<DecBatchItem> q_; bool quit_ = false, busy_ = false, hold_ = false;
  std::thread th_; bool running_ = false;
  std::deque<hipEvent_t> inflight_;      // last event of the batches launched and not yet seen finished
  hipEvent_t ring_[8] = {}; int ring_at_ = 0;
  struct Prof { hipEvent_t a[BK_COUNT], b[BK_COUNT]; bool used[BK_COUNT]; int frames[BK_COUNT]; bool pending = false; } prof_[4] = {};
  int prof_at_ = 0;
  void collect_prof(Prof &p);
  BatchStats stats_;
};

}  // namespace kvzx
