// kvazzup_amd/csrc/stream_pool.h -- the process's HIP streams, one per (device, role), shared by every encoder / decoder instance.
// HIP maps streams onto four hardware queues per priority level, handed out in creation order, and which of this library's streams end
// up sharing a queue decides 15-20 % of the pipeline's rate ("Hardware queues").  uvgComm runs several codec
// instances in ONE process -- one OpenHEVCFilter per peer beside the shared KvazaarFilter (filtergraph.cpp:347-351,561-589) -- and
// re-initialises them on every settings change (kvazaarfilter.cpp:91-119).  If every instance created streams of its own, the second
// instance's streams would alias the first one's hardware queues and be serialised behind its event waits (round 3: two pipelines in a
// process ran at 3 100 frames/s together against 8 400 for one).  So a stream belongs to a ROLE, not to an instance:
//   * an instance that asks for a role another open instance already holds gets THE SAME stream (use count up) -- at most one stream per
//     role and device however many instances are open; work of different instances is ordered on it like work of one instance,
//     cross-stream dependencies stay events (an event wait only ever names work queued earlier, so the order of enqueueing is a
//     topological order of every dependency: sharing cannot deadlock);
//   * when the last user closes, the stream is kept and handed to the next instance that asks for the role: a re-created encoder /
//     decoder inherits the queue layout of its predecessor.
// Exclusive roles (never shared while open): 'E', the GPU arithmetic coder's per-picture streams -- their kernels last milliseconds and
// must run side by side.  KVAZZUP_AMD_SHARE_STREAMS=0 makes every role exclusive (round 3's behaviour; measurement aid).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

namespace kvzx {

struct StreamPool {
  struct Entry { int device; char role, level; hipStream_t st; int users; };
  std::mutex m;
  std::vector<Entry> all;
  bool share;
  StreamPool() { const char *e = getenv("KVAZZUP_AMD_SHARE_STREAMS"); share = !(e && e[0] == '0'); }
  static StreamPool &get() { static StreamPool p; return p; }
};

// role: 'M' encoder main, 'T' tokenizer, 'I' input, 'E' GPU arithmetic coder; 'D' decoder main, 'U' upload, 'L' download.
// level: 'h' most urgent of the device's priority levels, 'l' least, anything else the default.
inline hipError_t stream_acquire(hipStream_t *st, int device, char role, char level)
{
  StreamPool &p = StreamPool::get();
  const bool exclusive = !p.share || role == 'E';
  std::lock_guard<std::mutex> l(p.m);
  for (auto &e : p.all)
    if (e.device == device && e.role == role && e.level == level && (e.users == 0 || !exclusive)) { e.users++; *st = e.st; return hipSuccess; }
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);            // (hi is the numerically smallest = most urgent)
  hipError_t rc;
  if (level == 'h') rc = hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
  else if (level == 'l') rc = hipStreamCreateWithPriority(st, hipStreamNonBlocking, lo);
  else rc = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  if (rc == hipSuccess) p.all.push_back({device, role, level, *st, 1});
  return rc;
}
inline void stream_release(hipStream_t st, int device, char role, char level)
{
  if (!st) return;
  (void)role; (void)level;
  hipSetDevice(device);
  hipStreamSynchronize(st);                             // (the closing instance's work; other users' work queued before now as well)
  StreamPool &p = StreamPool::get();
  std::lock_guard<std::mutex> l(p.m);
  for (auto &e : p.all)
    if (e.st == st && e.users > 0) { e.users--; return; }
}

// streams of a role currently open by more than one instance on this device?  (the batched-launch layer only engages then)
inline int stream_users(hipStream_t st)
{
  StreamPool &p = StreamPool::get();
  std::lock_guard<std::mutex> l(p.m);
  for (auto &e : p.all) if (e.st == st) return e.users;
  return 0;
}

}  // namespace kvzx
