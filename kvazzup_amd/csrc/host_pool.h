// kvazzup_amd/csrc/host_pool.h -- a small persistent pool of host threads that hands out the
// tasks of one job in increasing index order (task r may wait for task r-1, which is therefore
// always already running: the CTU-row wavefront of WPP, H.265 9.3.2.2).  Used by the host halves
// of entropy coding (entropy_host.h) and entropy decoding (decoder.hip).
#pragma once
#include <atomic>
#include <emmintrin.h>
#include <cstdint>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include <climits>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <linux/futex.h>
#include <sys/prctl.h>
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
  // (a thread's sleeps are rounded up by its timer slack, 50 us by default -- three times the nap; 2 us makes a nap a nap)
  static thread_local const bool slack_set = (prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0), true);
  (void)slack_set;
  static const int nap_env = [] { const char *e = getenv("KVAZZUP_AMD_NAP_US"); return e ? atoi(e) : 0; }();      // (measurement aid)
  if (nap_env > 0) nap_us = nap_env;
  for (;;) {
    const int r = done();                    // 1 done, 0 not yet, < 0 error
    if (r) return r > 0;
    std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
  }
}

// KVAZZUP_AMD_TIMELINE=<file>: a host-side timeline -- (time, thread, what, picture) records from the encoder's and decoder's threads,
// written out when the process ends; tools/host_timeline.py turns it into per-stage latencies.  Off: one predictable branch per call.
struct Timeline {
  struct Rec { uint64_t ns; uint32_t tid; char what[12]; long pic; };
  std::vector<Rec> recs; std::atomic<size_t> n{0}; const char *path = nullptr;
  static Timeline &get() { static Timeline *t = new Timeline(); return *t; }
  Timeline() { path = getenv("KVAZZUP_AMD_TIMELINE"); if (path) { recs.resize(1 << 21); atexit([] { Timeline::get().dump(); }); } }
  void add(const char *what, long pic)
  {
    if (!path) return;
    const size_t i = n.fetch_add(1, std::memory_order_relaxed);
    if (i >= recs.size()) return;
    Rec &r = recs[i];
    r.ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    r.tid = (uint32_t)syscall(SYS_gettid); strncpy(r.what, what, sizeof(r.what) - 1); r.what[sizeof(r.what) - 1] = 0; r.pic = pic;
  }
  void dump()
  {
    FILE *f = fopen(path, "w"); if (!f) return;
    const size_t m = n.load() < recs.size() ? n.load() : recs.size();
    for (size_t i = 0; i < m; i++) fprintf(f, "%llu %u %s %ld\n", (unsigned long long)recs[i].ns, recs[i].tid, recs[i].what, recs[i].pic);
    fclose(f);
  }
};
inline void tl(const char *what, long pic) { Timeline::get().add(what, pic); }

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
    quit_.store(true, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel); futex_wake_all(gen_);
    for (auto &t : workers_) t.join();
  }
  // Runs fn(0) .. fn(n - 1); returns when all have finished.
  void run(int n, const std::function<void(int)> &fn)
  {
    // A helper woken for the previous job may not have counted itself into active_ yet (late helpers are expected: they start from generation 0).  The task
    // counter is therefore made invalid FIRST and the stragglers waited for SECOND: a helper that counted itself in before the check below is waited for (what
    // it fetched is compared with the OLD total, still in place); one that counts itself in after it can only fetch values of at least 1 << 30, beyond any
    // total.  (The other order left a window: a helper between the check and the store fetched the old counter value -- at least the old total -- and
    // compared it with the NEW total; a job with more tasks than that, e.g. after a resolution change, ran one of its tasks twice.)  Published (next_ = 0) last.
    next_.store(1 << 30, std::memory_order_seq_cst);
    while (active_.load(std::memory_order_seq_cst) != 0) std::this_thread::yield();   // stragglers of the previous job
    fn_ = &fn;
    done_.store(0, std::memory_order_relaxed);
    total_.store(n, std::memory_order_relaxed);
    next_.store(0, std::memory_order_release);
    // The helpers sleep on the generation word itself and are woken together by one futex call (until round 5: a condition variable; measured on the
    // MI355X box's EPYC 9575F, tools/measure/pool_wake, no difference -- seventeen chained 50 us tasks take ~150 us with sixteen threads either way,
    // the last helper starts ~60 us after the call -- this form is the simpler one and spins briefly before it sleeps).
    if (n > 1 && !workers_.empty()) { gen_.fetch_add(1, std::memory_order_acq_rel); futex_wake_all(gen_); }
    drain();
    for (int d; (d = done_.load(std::memory_order_acquire)) < n;) futex_wait(done_, d);      // (the worker that finishes the last task wakes it)
  }

 private:
  void worker()
  {
    // (the generation the pool was BUILT with, not the one this thread finds when it first runs: a helper that starts after run() -- or after the destructor --
    // has already moved the word on would otherwise sleep on a value nobody changes again, and the destructor's join would never return.  Seen with a pool that
    // lived for one all-skip picture on a slow machine: tests/test_parser_probe.py.)
    int seen = 0;
    for (;;) {
      if (quit_.load(std::memory_order_acquire)) return;
      // (a short spin first: at thousands of pictures per second the next job is tens of microseconds away and a futex wake-up costs as much)
      for (int spins = 0; gen_.load(std::memory_order_acquire) == seen && spins < 200; spins++) __builtin_ia32_pause();
      while (gen_.load(std::memory_order_acquire) == seen) futex_wait(gen_, seen);
      seen = gen_.load(std::memory_order_acquire);
      if (quit_.load(std::memory_order_acquire)) return;
      drain();
    }
  }
  void drain()
  {
    active_.fetch_add(1, std::memory_order_seq_cst);      // (seq_cst with the fetch below and run()'s store / check: either run() sees this helper or the helper sees run()'s invalid counter)
    for (;;) {
      int r = next_.fetch_add(1, std::memory_order_seq_cst);
      if (r >= total_.load(std::memory_order_acquire)) break;
      (*fn_)(r);
      if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == total_.load(std::memory_order_acquire)) futex_wake_all(done_);
    }
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }
  std::vector<std::thread> workers_;
  std::atomic<int> gen_{0}; std::atomic<bool> quit_{false};
  const std::function<void(int)> *fn_ = nullptr;
  std::atomic<int> next_{1 << 30}, total_{0}, done_{0}, active_{0};
};

// Picture-sized host copies on several cores.  The reference's filters copy every picture once on their own thread (kvazaarfilter.cpp:
// 410-418 into the kvz_picture, openhevcfilter.cpp:212-229 out of the decoder's frame): ~0.17 ms per 1080p picture on one core, which at
// this library's rates is more than everything else the filter thread does.  A job is a list of 2-D pieces (rows x width bytes); helpers
// and the caller take pieces off a shared counter, so a helper that wakes late just finds less to do.
class CopyPool {
 public:
  struct Piece { uint8_t *dst; const uint8_t *src; size_t width, rows, dpitch, spitch; };
  explicit CopyPool(int threads)
  {
    for (int i = 0; i + 1 < threads; i++) workers_.emplace_back([this] { name_this_thread("kvzx-copy"); worker(); });
  }
  ~CopyPool()
  {
    quit_.store(true, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel); futex_wake_all(gen_);
    for (auto &t : workers_) t.join();
  }
  // one plane (rows x width, any pitches) cut into pieces of about `grain` bytes
  static void add_plane(std::vector<Piece> &v, uint8_t *dst, const uint8_t *src, size_t width, size_t rows, size_t dpitch, size_t spitch, size_t grain = 256 << 10)
  {
    if (dpitch == width && spitch == width) { width *= rows; rows = 1; dpitch = spitch = width; }      // one run
    if (rows == 1) { for (size_t o = 0; o < width; o += grain) v.push_back({dst + o, src + o, width - o < grain ? width - o : grain, 1, 0, 0}); return; }
    const size_t step = grain / width ? grain / width : 1;
    for (size_t r = 0; r < rows; r += step) v.push_back({dst + r * dpitch, src + r * spitch, width, rows - r < step ? rows - r : step, dpitch, spitch});
  }
  void run(const std::vector<Piece> &pieces)
  {
    if (workers_.empty() || pieces.size() < 2) { for (const Piece &p : pieces) one(p); return; }
    next_.store(1 << 30, std::memory_order_seq_cst);                                   // (as in OrderedPool::run: the counter invalid FIRST, then the stragglers -- see there)
    while (active_.load(std::memory_order_seq_cst) != 0) __builtin_ia32_pause();      // stragglers of the previous job
    job_ = &pieces;
    done_.store(0, std::memory_order_relaxed);
    total_.store((int)pieces.size(), std::memory_order_relaxed);
    next_.store(0, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel); futex_wake_all(gen_);
    drain();
    const int n = (int)pieces.size();
    for (int spins = 0; done_.load(std::memory_order_acquire) < n; spins++) { if (spins < 4000) __builtin_ia32_pause(); else std::this_thread::yield(); }
  }

 private:
  // A picture copy touches every byte once: the destination lines are written with non-temporal stores (no read-for-ownership of lines that are
  // about to be overwritten whole, and nothing of the caches' contents is pushed out for data nobody on this core reads again).  At this library's
  // rates the host boundary moves ~30 MB per 1080p picture through memory; a third of the copies' share of that was the ownership reads.
  static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n)
  {
    if (n < 4096) { memcpy(dst, src, n); return; }
    const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
      const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
      const __m128i c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
      _mm_stream_si128((__m128i *)(dst + i), a); _mm_stream_si128((__m128i *)(dst + i + 16), b);
      _mm_stream_si128((__m128i *)(dst + i + 32), c); _mm_stream_si128((__m128i *)(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
  }
  static void one(const Piece &p)
  {
    static const bool nt = getenv("KVAZZUP_AMD_COPY_PLAIN") == nullptr;
    for (size_t r = 0; r < p.rows; r++) { if (nt) stream_copy(p.dst + r * p.dpitch, p.src + r * p.spitch, p.width); else memcpy(p.dst + r * p.dpitch, p.src + r * p.spitch, p.width); }
    if (nt) _mm_sfence();
  }
  void worker()
  {
    // (the generation the pool was BUILT with, not the one this thread finds when it first runs: a helper that starts after run() -- or after the destructor --
    // has already moved the word on would otherwise sleep on a value nobody changes again, and the destructor's join would never return.  Seen with a pool that
    // lived for one all-skip picture on a slow machine: tests/test_parser_probe.py.)
    int seen = 0;
    for (;;) {
      if (quit_.load(std::memory_order_acquire)) return;
      // a short spin first -- at several thousand pictures per second the next job is ~100 us away and a futex wake-up costs a good part of that -- but SHORT:
      // the host is what this pipeline runs out of, and helpers that spin half the time between jobs took cores from the parsers (2000 pauses: 4 500-5 000
      // frames/s through the host boundary on one box, 200: 4 900-5 400, none: 4 700-4 900)
      static const int spin_max = [] { const char *e = getenv("KVAZZUP_AMD_COPY_SPIN"); return e ? atoi(e) : 200; }();
      for (int spins = 0; gen_.load(std::memory_order_acquire) == seen && spins < spin_max; spins++) __builtin_ia32_pause();
      while (gen_.load(std::memory_order_acquire) == seen) futex_wait(gen_, seen);
      seen = gen_.load(std::memory_order_acquire);
      if (quit_.load(std::memory_order_acquire)) return;
      drain();
    }
  }
  void drain()
  {
    active_.fetch_add(1, std::memory_order_seq_cst);
    for (;;) {
      const int r = next_.fetch_add(1, std::memory_order_seq_cst);
      if (r >= total_.load(std::memory_order_acquire)) break;
      one((*job_)[(size_t)r]);
      done_.fetch_add(1, std::memory_order_acq_rel);
    }
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }
  std::vector<std::thread> workers_;
  const std::vector<Piece> *job_ = nullptr;
  std::atomic<int> gen_{0}, next_{1 << 30}, total_{0}, done_{0}, active_{0};
  std::atomic<bool> quit_{false};
};

}  // namespace kvzx
