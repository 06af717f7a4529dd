// tools/measure/queue_share/queue_share.hip -- several PROCESSES on one GPU, each with streams at several priority levels: what a dependent chain of small
// kernels hopping between two streams costs per hop as the processes' hardware queues add up.  (The two-rank bench beside pytest's three workers crawled
// at 10-30 frames/s in round 4: every picture is a dozen such hops.)  HIP gives a process up to GPU_MAX_HW_QUEUES (default 4) hardware queues PER PRIORITY
// LEVEL it uses; the driver maps a limited number of user queues at a time and time-slices the rest by process.
//   hipcc --offload-arch=gfx950 -O2 -o queue_share queue_share.hip;  ./queue_share <streams> <priority levels 1..3> <hops> [tag]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <unistd.h>
__global__ void k_hop(unsigned *p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
int main(int argc, char **argv)
{
  const int ns = argc > 1 ? atoi(argv[1]) : 8, np = argc > 2 ? atoi(argv[2]) : 3, hops = argc > 3 ? atoi(argv[3]) : 2000;
  const char *tag = argc > 4 ? argv[4] : "";
  int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);        // lo = least urgent (numerically greatest)
  std::vector<hipStream_t> st(ns);
  for (int i = 0; i < ns; i++) { const int pr = np <= 1 ? lo : lo - (i % np) * ((lo - hi) / (np > 1 ? np - 1 : 1)); hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr < hi ? hi : pr); }
  unsigned *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  for (auto s : st) hipLaunchKernelGGL(k_hop, dim3(1), dim3(64), 0, s, d);      // every stream has had work: its hardware queue exists
  hipDeviceSynchronize();
  hipEvent_t e[2]; hipEventCreateWithFlags(&e[0], hipEventDisableTiming); hipEventCreateWithFlags(&e[1], hipEventDisableTiming);
  std::vector<double> rounds;
  for (int rep = 0; rep < 5; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int h = 0; h < hops; h++) {
      hipStream_t a = st[(h & 1) ? 1 % ns : 0], b = st[(h & 1) ? 0 : 1 % ns];
      hipLaunchKernelGGL(k_hop, dim3(1), dim3(64), 0, a, d);
      hipEventRecord(e[h & 1], a);
      hipStreamWaitEvent(b, e[h & 1], 0);
    }
    hipDeviceSynchronize();
    rounds.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / hops);
  }
  std::sort(rounds.begin(), rounds.end());
  printf("%s pid %d: %d streams at %d priority level(s), GPU_MAX_HW_QUEUES=%s: %.1f us per hop (median of 5 x %d hops; worst %.1f)\n", tag, (int)getpid(), ns, np,
         getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default", rounds[2], hops, rounds[4]);
  return 0;
}
