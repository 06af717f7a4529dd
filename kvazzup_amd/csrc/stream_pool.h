// kvazzup_amd/csrc/stream_pool.h -- HIP streams that outlive the encoder / decoder instance that used them.
// HIP maps streams onto four hardware queues per priority level, handed out in creation order, and which of this library's streams end
// up sharing a queue decides 15-20 % of the pipeline's rate ("Hardware queues").  A process that closes an
// instance and opens another -- uvgComm re-initialises its filters on every settings change (kvazaarfilter.cpp:91-119) -- would get
// streams one turn further round the queues each time.  So streams are kept by role when an instance closes and handed to the next
// instance that asks for the same role on the same device: it inherits the queue layout of its predecessor.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

namespace kvzx {

struct StreamPool {
  struct Idle { int device; char role, level; hipStream_t st; };
  std::mutex m;
  std::vector<Idle> idle;
  static StreamPool &get() { static StreamPool p; return p; }
};

// role: 'M' encoder main, 'T' tokenizer, 'I' input, 'E' GPU arithmetic coder; 'D' decoder main, 'U' upload, 'L' download.
// level: 'h' most urgent of the device's priority levels, 'l' least, anything else the default.
inline hipError_t stream_acquire(hipStream_t *st, int device, char role, char level)
{
  StreamPool &p = StreamPool::get();
  {
    std::lock_guard<std::mutex> l(p.m);
    for (size_t i = 0; i < p.idle.size(); i++)
      if (p.idle[i].device == device && p.idle[i].role == role && p.idle[i].level == level) { *st = p.idle[i].st; p.idle.erase(p.idle.begin() + (long)i); return hipSuccess; }
  }
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);            // (hi is the numerically smallest = most urgent)
  if (level == 'h') return hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
  if (level == 'l') return hipStreamCreateWithPriority(st, hipStreamNonBlocking, lo);
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
inline void stream_release(hipStream_t st, int device, char role, char level)
{
  if (!st) return;
  hipSetDevice(device);
  hipStreamSynchronize(st);
  StreamPool &p = StreamPool::get();
  std::lock_guard<std::mutex> l(p.m);
  p.idle.push_back({device, role, level, st});
}

}  // namespace kvzx
