// kvazzup_amd/csrc/batch.hip -- the decoders' submission layer (batch.h): one submitter thread per device launches the pictures of all
// open decoder instances that are waiting, same kernels of different pictures as ONE launch.
#include "batch.h"
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include "dec_kernels.h"
#include "host_pool.h"
#include "stream_pool.h"

namespace kvzx {

DecBatcher &DecBatcher::get(int device)
{
  static std::mutex m;
  static std::map<int, DecBatcher *> all;              // (never freed: decoders may close from static destructors)
  std::lock_guard<std::mutex> l(m);
  DecBatcher *&b = all[device];
  if (!b) {
    b = new DecBatcher(device);
    const char *e = getenv("KVAZZUP_AMD_BATCH");
    b->configured_ = !(e && e[0] == '0') && StreamPool::get().share;      // (without shared role streams every decoder has a stream of its own: nothing to batch on)
    b->enabled_ = b->configured_;
  }
  return *b;
}

// attach / detach are serialised by life_ from their first to their last line: a decoder that opens while the last one is closing waits until the
// old submitter thread has been joined and quit_ is clear again (it used to be able to start a thread that still saw quit_, launched what was queued
// and left -- running_ true, nobody behind it, every later picture waiting for ever in complete_gpu).
void DecBatcher::attach(hipStream_t st)
{
  std::lock_guard<std::mutex> life(life_);
  std::lock_guard<std::mutex> l(m_);
  if (users_.load() == 0) { stream_ = st; enabled_ = configured_; }      // (a mismatch among earlier users is forgotten once all of them have closed)
  else if (st != stream_) enabled_ = false;            // decoders on different streams (priority levels set apart by hand): every one launches for itself
  users_.fetch_add(1);
}

void DecBatcher::detach()
{
  std::lock_guard<std::mutex> life(life_);
  std::thread t;
  {
    std::unique_lock<std::mutex> l(m_);
    if (users_.fetch_sub(1) != 1) return;
    // the last decoder of the device closes: the submitter thread goes with it (nobody can submit: no decoder is attached, none can attach before this returns)
    quit_ = true; cv_.notify_all();
    t = std::move(th_);
  }
  if (t.joinable()) t.join();
  std::lock_guard<std::mutex> l(m_);
  quit_ = false; running_ = false;
  hipSetDevice(device_);
  for (auto &p : prof_) for (int k = 0; k < BK_COUNT; k++) { if (p.a[k]) { hipEventDestroy(p.a[k]); hipEventDestroy(p.b[k]); p.a[k] = p.b[k] = nullptr; } p.pending = false; }
  inflight_.clear();
}

void DecBatcher::submit(const DecBatchItem &it)
{
  {
    std::lock_guard<std::mutex> l(m_);
    q_.push_back(it);
    if (!running_) { running_ = true; th_ = std::thread([this] { name_this_thread("kvzx-batch"); run(); }); }
  }
  cv_.notify_one();
}

void DecBatcher::drain(const void *owner)
{
  std::unique_lock<std::mutex> l(m_);
  idle_cv_.wait(l, [&] {
    if (busy_) return false;
    for (auto &it : q_) if (it.owner == owner) return false;
    return true;
  });
}

void DecBatcher::hold(bool on)
{
  { std::lock_guard<std::mutex> l(m_); hold_ = on; }
  cv_.notify_all();
}

void DecBatcher::get_stats(BatchStats *out, bool reset)
{
  std::lock_guard<std::mutex> l(m_);
  hipSetDevice(device_);
  for (auto &p : prof_) if (p.pending) collect_prof(p);
  if (out) *out = stats_;
  if (reset) stats_ = BatchStats();
}

void DecBatcher::collect_prof(Prof &p)
{
  for (int k = 0; k < BK_COUNT; k++) {
    if (!p.used[k]) continue;
    float ms = 0;
    if (hipEventSynchronize(p.b[k]) == hipSuccess && hipEventElapsedTime(&ms, p.a[k], p.b[k]) == hipSuccess) { stats_.ms[k] += ms; stats_.launches[k]++; stats_.frames[k] += (uint64_t)p.frames[k]; }
    p.used[k] = false;
  }
  p.pending = false;
}

void DecBatcher::launch(DecBatchItem *items, int n)
{
  const DecFrame *h[KVZ_DEC_BATCH_MAX], *d[BK_COUNT][KVZ_DEC_BATCH_MAX];
  int cnt[BK_COUNT] = {0, 0, 0, 0};
  bool prof = false;
  for (int i = 0; i < n; i++) {
    DecBatchItem &it = items[i];
    if (it.wait0) hipStreamWaitEvent(stream_, it.wait0, 0);
    if (it.wait1) hipStreamWaitEvent(stream_, it.wait1, 0);
    h[i] = &it.f;
    const bool on[BK_COUNT] = {it.inter, it.intra, it.deblock, it.sao};
    for (int k = 0; k < BK_COUNT; k++) { d[k][i] = on[k] ? it.d_f : nullptr; cnt[k] += on[k] ? 1 : 0; }
    prof |= it.profile;
  }
  Prof *p = nullptr;
  if (prof) {
    p = &prof_[prof_at_++ & 3];
    if (p->pending) { std::lock_guard<std::mutex> l(m_); collect_prof(*p); }
    for (int k = 0; k < BK_COUNT; k++) if (!p->a[k]) { hipEventCreate(&p->a[k]); hipEventCreate(&p->b[k]); }
  }
  for (int k = 0; k < BK_COUNT; k++) {
    if (!cnt[k]) continue;
    if (p) { hipEventRecord(p->a[k], stream_); p->used[k] = true; p->frames[k] = cnt[k]; }
    switch (k) {
      case BK_INTER: launch_dec_inter_n(h, d[k], n, stream_); break;
      case BK_INTRA: launch_dec_intra_n(h, d[k], n, stream_); break;
      case BK_DEBLOCK: launch_dec_deblock_n(h, d[k], n, stream_); break;
      default: launch_dec_sao_n(h, d[k], n, stream_); break;
    }
    if (p) hipEventRecord(p->b[k], stream_);
  }
  if (p) { std::lock_guard<std::mutex> l(m_); p->pending = true; }
  for (int i = 0; i < n; i++) {
    hipEventRecord(items[i].done, stream_);
    items[i].launched->store(1, std::memory_order_release);
    futex_wake_all(*items[i].launched);
  }
  hipEvent_t &e = ring_[ring_at_++ & 7];
  if (!e) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEventRecord(e, stream_);
  inflight_.push_back(e);
}

void DecBatcher::run()
{
  hipSetDevice(device_);
  DecBatchItem batch[KVZ_DEC_BATCH_MAX];
  for (;;) {
    {
      std::unique_lock<std::mutex> l(m_);
      busy_ = false; idle_cv_.notify_all();
      cv_.wait(l, [&] { return quit_ || (!q_.empty() && !hold_); });
      if (q_.empty() || (hold_ && quit_)) { if (quit_) return; continue; }
      busy_ = true;
    }
    // no more than two batches in flight: while the GPU is the busy side, arrivals accumulate and leave together
    while (!inflight_.empty() && hipEventQuery(inflight_.front()) != hipErrorNotReady) inflight_.pop_front();
    if (inflight_.size() >= 2) {
      hipEvent_t e = inflight_.front();
      nap_until([&] { return hipEventQuery(e) != hipErrorNotReady ? 1 : 0; }, 15);
      inflight_.pop_front();
    }
    int n = 0;
    {
      std::lock_guard<std::mutex> l(m_);
      const void *taken[KVZ_DEC_BATCH_MAX];
      for (auto it = q_.begin(); it != q_.end() && n < KVZ_DEC_BATCH_MAX;) {
        bool dup = false;
        for (int k = 0; k < n; k++) dup |= taken[k] == it->owner;
        // (an owner's later picture stays behind its earlier one: once one of its pictures is in the batch or was left behind, the rest wait for the next batch)
        if (dup) { ++it; continue; }
        taken[n] = it->owner; batch[n++] = *it; it = q_.erase(it);
      }
      stats_.batches++; stats_.pictures += (uint64_t)n; stats_.by_size[n]++;
    }
    launch(batch, n);
  }
}

}  // namespace kvzx
