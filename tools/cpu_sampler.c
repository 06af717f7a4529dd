/* tools/cpu_sampler.c -- a sampling CPU profiler for boxes without perf: LD_PRELOAD this library and every thread's program
 * counter is recorded on SIGPROF (ITIMER_PROF: delivered to running threads in proportion to the CPU time they use).  At exit the
 * samples go to $CPU_SAMPLER_OUT (default gpurun_out/cpu_samples.txt) as "pc thread-name" lines preceded by /proc/self/maps;
 * tools/cpu_sampler_report.py turns them into a per-source-line histogram of a library built with -g.
 *   gcc -O2 -shared -fPIC -o tools/libcpusampler.so tools/cpu_sampler.c -lpthread */
#define _GNU_SOURCE
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

#define MAX_SAMPLES (1 << 21)
static struct { uint64_t pc; int tid; } *g_s;
static volatile long g_n;

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
  (void)sig; (void)si;
  ucontext_t *uc = (ucontext_t *)uc_;
  long i = __sync_fetch_and_add(&g_n, 1);
  if (i < MAX_SAMPLES) { g_s[i].pc = (uint64_t)uc->uc_mcontext.gregs[REG_RIP]; g_s[i].tid = (int)syscall(SYS_gettid); }
}

/* cpu_sampler_begin() / cpu_sampler_end(): bracket a region (bench.py calls them around its timed region when CPU_SAMPLER_REGION is
 * set): begin drops what was sampled so far and re-arms handler and timer (a library may have replaced either), end stops the timer */
void cpu_sampler_end(void) { struct itimerval it; memset(&it, 0, sizeof(it)); setitimer(ITIMER_PROF, &it, NULL); }
void cpu_sampler_begin(void)
{
  if (!g_s) return;
  g_n = 0;
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  const char *us = getenv("CPU_SAMPLER_US");
  struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = us ? atoi(us) : 500; it.it_value = it.it_interval;
  setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((constructor)) static void sampler_start(void)
{
  if (getenv("CPU_SAMPLER_OFF")) return;
  g_s = calloc(MAX_SAMPLES, sizeof(*g_s));
  if (getenv("CPU_SAMPLER_REGION")) return;      /* armed by cpu_sampler_begin() */
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  const char *us = getenv("CPU_SAMPLER_US");
  struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = us ? atoi(us) : 500; it.it_value = it.it_interval;
  setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((destructor)) static void sampler_stop(void)
{
  if (!g_s) return;
  cpu_sampler_end();
  const char *path = getenv("CPU_SAMPLER_OUT"); if (!path) path = "gpurun_out/cpu_samples.txt";
  FILE *o = fopen(path, "w"); if (!o) return;
  FILE *m = fopen("/proc/self/maps", "r");
  if (m) { char line[1024]; while (fgets(line, sizeof(line), m)) if (strstr(line, " r-xp ") || strstr(line, "r-xp")) fprintf(o, "M %s", line); fclose(m); }
  long n = g_n < MAX_SAMPLES ? g_n : MAX_SAMPLES;
  /* thread names while the threads may still exist; samples of threads that are gone keep their id */
  for (long i = 0; i < n; i++) fprintf(o, "S %llx %d\n", (unsigned long long)g_s[i].pc, g_s[i].tid);
  fclose(o);
}
