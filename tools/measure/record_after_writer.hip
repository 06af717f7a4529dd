// What an event record behind a kernel that has just WRITTEN a picture costs the stream's next kernel: the record is an agent-scope release -- the L2s'
// dirty lines go out first.  A chain of (writer kernel, [record], reader kernel) on one stream, another stream waiting for the record;
// plain stores against nontemporal ones.  hipcc --offload-arch=gfx950 -O2 -o tools/measure/record_after_writer tools/measure/record_after_writer.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <atomic>
template <bool NT> __global__ void writer(uint4 *p, size_t n, unsigned v)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 x = {v, v + 1, v + 2, (unsigned)i};
    if (NT) __builtin_nontemporal_store(x, (v4 *)&p[i]); else *(v4 *)&p[i] = x;
  }
}
__global__ void reader(const uint4 *p, size_t n, unsigned *out)
{
  unsigned a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i].x;
  if (a == 0x12345678u) *out = a;
}
// the writer that says "done" by itself: every workgroup releases its stores (agent scope) and counts itself; the last one stores the flag (system scope) --
// no packet between this kernel and the stream's next one
__global__ void writer_flag(uint4 *p, size_t n, unsigned v, unsigned *count, uint32_t *flag, unsigned value)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 x = {v, v + 1, v + 2, (unsigned)i};
    *(v4 *)&p[i] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // (no fence per workgroup: on gfx950 an agent-scope release writes the XCD's whole L2 back -- 2048 of them made this kernel 54 us; what the waiter reads
    // must leave with write-through stores instead, or be final before this kernel started)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) { __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}
// a gate: keeps stream b busy for `ticks` of the 100 MHz clock while the host queues everything behind it -- what is timed after it is the GPU's time alone
__global__ void gate(unsigned long long ticks, unsigned *out) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); if (threadIdx.x == 9999) *out = 1; }
// "done" for the host as a one-thread kernel that stores a number into host-mapped memory: the host reads memory, it makes no HIP call
__global__ void say(volatile unsigned *host_word, unsigned v) { __hip_atomic_store((unsigned *)host_word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }      // (relaxed: the kernels in front of this one have ended -- a release here would write the L2 back once more)
__global__ void tiny(unsigned *out) { if (threadIdx.x == 9999) *out = 1; }
int main()
{
  hipStream_t b, c; hipStreamCreate(&b); hipStreamCreate(&c);
  const size_t bytes = 6u << 20, n = bytes / 16;
  uint4 *p; unsigned *out; hipMalloc(&p, bytes); hipMalloc(&out, 4);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
  const int N = 300;
  uint32_t *flag = nullptr; if (hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory\n"); hipMalloc((void **)&flag, 8); } hipMemset(flag, 0, 8);
  const char *names[17] = {"no record", "record, nobody waits", "record + another stream waiting behind it", "hipExtLaunchKernelGGL stop event + waiter", "record with default flags + waiter", "record, waiter enqueued one iteration later", "hipStreamWriteValue32 + hipStreamWaitValue32 on another stream", "hipStreamWriteValue32, nobody waits", "flag stored by the writer's last workgroup + hipStreamWaitValue32 on another stream", "flag stored by the writer's last workgroup, nobody waits", "512 workgroups: no record", "512 workgroups: record + waiter", "512 workgroups: flag by the last workgroup + hipStreamWaitValue32", "record, a host thread polls it with hipEventQuery", "no record, but a wait for a long-complete event of another stream in front of the reader", "record (host polls) + wait for a long-complete event", "one-thread kernel stores a number in host-mapped memory, a host thread polls the memory"};
  unsigned *h_word = nullptr; hipHostMalloc((void **)&h_word, 64, hipHostMallocMapped); *h_word = 0; unsigned *d_word = nullptr; hipHostGetDevicePointer((void **)&d_word, h_word, 0);
  hipEvent_t old_ev; hipEventCreateWithFlags(&old_ev, hipEventDisableTiming | hipEventDisableSystemFence); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); hipEventRecord(old_ev, c); hipStreamSynchronize(c);
  unsigned *count; hipMalloc(&count, 4); hipMemset(count, 0, 4);
  hipEvent_t evd; hipEventCreateWithFlags(&evd, hipEventDisableTiming);
  hipEvent_t ring[4]; for (auto &e : ring) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  for (int mode = 0; mode < 17; mode++) {
    hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::thread poller;
    if (mode == 13 || mode == 15) poller = std::thread([&] { while (!stop.load()) { hipEventQuery(ev); std::this_thread::sleep_for(std::chrono::microseconds(25)); } });
    const bool gated = getenv("GATE") != nullptr;
    if (gated) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, b, 4000000ull, out);      // 40 ms
    std::thread poller2;
    unsigned long seen = 0;
    if (mode == 16) poller2 = std::thread([&] { unsigned last = 0; while (!stop.load()) { const unsigned v = *(volatile unsigned *)h_word; if (v != last) { last = v; seen++; } std::this_thread::sleep_for(std::chrono::microseconds(25)); } });
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      if (mode == 10 || mode == 11) hipLaunchKernelGGL(writer<false>, dim3(512), dim3(256), 0, b, p, n, (unsigned)i);
      else if (mode == 12) hipLaunchKernelGGL(writer_flag, dim3(512), dim3(256), 0, b, p, n, (unsigned)i, count, flag, 1000u * mode + (unsigned)i + 1u);
      else if (mode == 8 || mode == 9) hipLaunchKernelGGL(writer_flag, dim3(2048), dim3(256), 0, b, p, n, (unsigned)i, count, flag, 1000u * mode + (unsigned)i + 1u);
      else if (mode == 3) hipExtLaunchKernelGGL(writer<false>, dim3(2048), dim3(256), 0, b, nullptr, ev, 0, p, n, (unsigned)i);
      else hipLaunchKernelGGL(writer<false>, dim3(2048), dim3(256), 0, b, p, n, (unsigned)i);
      if (mode == 1 || mode == 2 || mode == 11 || mode == 13 || mode == 15) hipEventRecord(ev, b);
      if (mode == 14 || mode == 15) hipStreamWaitEvent(b, old_ev, 0);
      if (mode == 4) hipEventRecord(evd, b);
      if (mode == 5) hipEventRecord(ring[i & 3], b);
      if (mode == 2 || mode == 3 || mode == 11) { hipStreamWaitEvent(c, ev, 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); }
      if (mode == 4) { hipStreamWaitEvent(c, evd, 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); }
      if (mode == 5 && i) { hipStreamWaitEvent(c, ring[(i - 1) & 3], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); }
      if (mode == 8 || mode == 12) { if (hipStreamWaitValue32(c, flag, 1000u * mode + (unsigned)i + 1u, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { printf("wait value failed\n"); return 1; } hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); }
      if (mode == 6 || mode == 7) { if (hipStreamWriteValue32(b, flag, 1000u * mode + (unsigned)i + 1u, 0) != hipSuccess) { printf("write value failed\n"); return 1; } }
      if (mode == 6) { if (hipStreamWaitValue32(c, flag, 1000u * mode + (unsigned)i + 1u, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { printf("wait value failed\n"); return 1; } hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, out); }
      if (mode == 16) hipLaunchKernelGGL(say, dim3(1), dim3(1), 0, b, d_word, (unsigned)(i + 1));
      hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, b, p, n, out);
    }
    hipStreamSynchronize(b); hipStreamSynchronize(c);
    stop.store(true); if (poller.joinable()) poller.join(); if (poller2.joinable()) { poller2.join(); printf("(the poller saw %lu of %d values)\n", seen, N); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - (gated ? 40000.0 : 0.0);
    printf("%-50s %7.2f us per (writer + reader)\n", names[mode], us / N);
  }
  return 0;
}
