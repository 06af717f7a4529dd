// tools/measure/pool_wake/pool_wake.cpp -- how long a pool of sleeping helper threads takes to get going: OrderedPool as it was until round 4 (helpers asleep
// on a condition variable, host_pool_r04.h) against round 5's (helpers asleep on the generation word, woken together by one futex call).  The job is the
// shape of a picture's row-parallel parse: 17 tasks of ~50 us, every task waiting until the one before it is 10 % done.
//   g++ -O2 -std=c++17 -pthread -DOLD -o pool_old pool_wake.cpp; g++ -O2 -std=c++17 -pthread -o pool_new pool_wake.cpp
#include <atomic>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
#include <unistd.h>
#ifdef OLD
#include "host_pool_r04.h"
#else
#include "../../../kvazzup_amd/csrc/host_pool.h"
#endif
using namespace kvzx;
using Clock = std::chrono::steady_clock;
static double us_since(Clock::time_point t) { return std::chrono::duration<double, std::micro>(Clock::now() - t).count(); }
int main(int argc, char **argv)
{
  const int threads = argc > 1 ? atoi(argv[1]) : 16, gap_us = argc > 2 ? atoi(argv[2]) : 16000, task_us = argc > 3 ? atoi(argv[3]) : 50;
  OrderedPool pool(threads);
  std::vector<double> total, last_start;
  for (int rep = 0; rep < 60; rep++) {
    usleep(gap_us);                                        // the helpers are asleep when the next picture arrives (a 60 pictures/s source)
    std::atomic<int> progress[17]; for (auto &p : progress) p.store(0);
    double start[17];
    const auto t0 = Clock::now();
    pool.run(17, [&](int r) {
      start[r] = us_since(t0);
      for (int step = 0; step < 10; step++) {
        if (r > 0) while (progress[r - 1].load(std::memory_order_acquire) <= step) __builtin_ia32_pause();      // a row follows the row above
        const auto a = Clock::now(); while (us_since(a) < task_us / 10.0) {}
        progress[r].store(step + 1, std::memory_order_release);
      }
    });
    total.push_back(us_since(t0)); last_start.push_back(*std::max_element(start, start + 17));
  }
  std::sort(total.begin(), total.end()); std::sort(last_start.begin(), last_start.end());
  printf("%s pool, %d threads, %d us between jobs, 17 chained tasks of %d us (serial %d us, ideal %d us): job median %.0f us (p90 %.0f), last task started at median %.0f us\n",
#ifdef OLD
         "r04 condition-variable",
#else
         "r05 futex",
#endif
         threads, gap_us, task_us, 17 * task_us, task_us + 16 * task_us / 10, total[total.size() / 2], total[total.size() * 9 / 10], last_start[last_start.size() / 2]);
  return 0;
}
