// What a hipStreamWaitEvent in front of a kernel costs a serial chain of short kernels, by the state of the event when the wait is ENQUEUED:
// (a) no wait at all, (b) the event already complete at enqueue time, (c) recorded on another stream but not complete yet when the wait is
// enqueued (the usual case of a pipeline that enqueues several pictures ahead).  hipcc --offload-arch=gfx950 -O2 -o /tmp/wec tools/measure/wait_event_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void spin(long long cycles, int *sink) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (sink && threadIdx.x == 12345) *sink = 1; }
int main()
{
  hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
  const int N = 200; const long long K = 20000;                      // ~10 us kernels
  std::vector<hipEvent_t> ev(N); for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  for (int mode = 0; mode < 4; mode++) {
    hipDeviceSynchronize();
    if (mode == 1) { for (int i = 0; i < N; i++) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000, nullptr); hipEventRecord(ev[i], a); } hipStreamSynchronize(a); }   // complete before the waits are enqueued
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      if (mode == 2) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000, nullptr); hipEventRecord(ev[i], a); }   // recorded just now: not complete at enqueue time, complete long before the chain gets there
      if (mode == 3) { if (i + 1 < N) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000, nullptr); hipEventRecord(ev[i + 1], a); } if (i == 0) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000, nullptr); hipEventRecord(ev[0], a); hipStreamSynchronize(a); } else hipEventSynchronize(ev[i]); }   // the host makes sure it is complete first
      if (mode) hipStreamWaitEvent(b, ev[i], 0);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, K, nullptr);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, K, nullptr);
    }
    hipStreamSynchronize(b);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const char *names[4] = {"no wait", "event complete when the wait is enqueued", "event recorded, not complete, when the wait is enqueued", "host waits for the event, then enqueues the wait"};
    printf("%-62s %7.2f us per (wait + 2 kernels)\n", names[mode], us / N);
  }
  return 0;
}
