// kvazzup_amd/csrc/host_pool.h -- a small persistent pool of host threads that hands out the
// tasks of one job in increasing index order (task r may wait for task r-1, which is therefore
// always already running: the CTU-row wavefront of WPP, H.265 9.3.2.2).  Used by the host halves
// of entropy coding (entropy_host.h) and entropy decoding (decoder.hip).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace kvzx {

// thread names show up in top -H / perf / /proc/<pid>/task/*/comm (15 characters)
inline void name_this_thread(const char *name) { pthread_setname_np(pthread_self(), name); }

// Sleeping on an atomic word (Linux futex) instead of spinning through sched_yield(): the waiter costs nothing while it waits and
// the waker pays one system call.  wait: returns when woken or when the word no longer holds `seen`; callers re-check in a loop.
inline void futex_wait(std::atomic<int> &a, int seen) { syscall(SYS_futex, reinterpret_cast<int *>(&a), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
inline void futex_wake_all(std::atomic<int> &a) { syscall(SYS_futex, reinterpret_cast<int *>(&a), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
static_assert(sizeof(std::atomic<int>) == sizeof(int), "futex word");

// Waiting for the GPU without burning a core: hipEventSynchronize / hipStreamSynchronize poll (the blocking-sync event flag makes
// no difference on this stack), which costs a full core per waiting thread.  Where a lag hides the wake-up -- the encoder's
// background workers, the frame-threaded decoder -- the thread sleeps in short naps between queries instead.
template <class Query> inline bool nap_until(Query &&done, int nap_us = 25)
{
  for (;;) {
    const int r = done();                    // 1 done, 0 not yet, < 0 error
    if (r) return r > 0;
    std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
  }
}

class OrderedPool {
 public:
  explicit OrderedPool(int threads)
  {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && threads > hw) threads = hw;
    if (threads < 1) threads = 1;
    for (int i = 0; i + 1 < threads; i++) workers_.emplace_back([this] { name_this_thread("kvzx-pool"); worker(); });   // the caller is the last worker
  }
  ~OrderedPool()
  {
    { std::lock_guard<std::mutex> l(m_); quit_ = true; gen_++; }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  // Runs fn(0) .. fn(n - 1); returns when all have finished.
  void run(int n, const std::function<void(int)> &fn)
  {
    while (active_.load(std::memory_order_acquire) != 0) std::this_thread::yield();   // stragglers of the previous job
    fn_ = &fn;
    done_.store(0, std::memory_order_relaxed);
    total_.store(n, std::memory_order_relaxed);
    next_.store(0, std::memory_order_release);
    if (n > 1 && !workers_.empty()) { { std::lock_guard<std::mutex> l(m_); gen_++; } cv_.notify_all(); }
    drain();
    for (int d; (d = done_.load(std::memory_order_acquire)) != n;) futex_wait(done_, d);      // (the worker that finishes the last task wakes it)
  }

 private:
  void worker()
  {
    uint64_t seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return gen_ != seen; }); seen = gen_; if (quit_) return; }
      drain();
    }
  }
  void drain()
  {
    active_.fetch_add(1, std::memory_order_acq_rel);
    for (;;) {
      int r = next_.fetch_add(1, std::memory_order_acq_rel);
      if (r >= total_.load(std::memory_order_acquire)) break;
      (*fn_)(r);
      if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == total_.load(std::memory_order_acquire)) futex_wake_all(done_);
    }
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }
  std::vector<std::thread> workers_;
  std::mutex m_; std::condition_variable cv_;
  uint64_t gen_ = 0; bool quit_ = false;
  const std::function<void(int)> *fn_ = nullptr;
  std::atomic<int> next_{1 << 30}, total_{0}, done_{0}, active_{0};
};

}  // namespace kvzx
