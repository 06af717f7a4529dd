/* tools/cpu_sampler.c -- a sampling CPU profiler for boxes without perf: LD_PRELOAD this library and every thread's program
 * counter is recorded on SIGPROF.  Every thread gets a timer of its own on its CPU-time clock (pthread_create is interposed;
 * timer_create with SIGEV_THREAD_ID), so a thread is sampled in proportion to the CPU time IT uses -- a process-wide ITIMER_PROF
 * signal lands on the main thread nearly every time, whoever burnt the CPU.  At exit the
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
#include <dlfcn.h>
#include <time.h>

#define MAX_SAMPLES (1 << 21)
static struct { uint64_t pc, caller; int tid; } *g_s;      /* caller: for a sample outside the measured library, the first word on the thread's stack that points into it (0: none) */
static volatile uint64_t g_lib_lo, g_lib_hi;               /* the measured library's executable mapping (CPU_SAMPLER_LIB, default "libkvazzup_amd"): found by cpu_sampler_begin() */
static volatile long g_n;
static volatile int g_on;                  /* samples are kept only between begin and end (or always, without CPU_SAMPLER_REGION) */

static __thread uint64_t t_stack_hi __attribute__((tls_model("initial-exec")));
static void on_prof(int sig, siginfo_t *si, void *uc_)
{
  (void)sig; (void)si;
  ucontext_t *uc = (ucontext_t *)uc_;
  if (!g_on) return;
  long i = __sync_fetch_and_add(&g_n, 1);
  if (i >= MAX_SAMPLES) return;
  const uint64_t pc = (uint64_t)uc->uc_mcontext.gregs[REG_RIP], lo = g_lib_lo, hi = g_lib_hi;
  uint64_t caller = 0;
  if (hi && !(pc >= lo && pc < hi)) {
    /* no unwinder in a signal handler: the innermost return address into the library is, nearly always, the first stack word that points into its code */
    const uint64_t *sp = (const uint64_t *)uc->uc_mcontext.gregs[REG_RSP];
    const uint64_t top = t_stack_hi;                                    /* (this thread's stack ends there: arm_this_thread()) */
    if ((uint64_t)sp < top && top - (uint64_t)sp < (64u << 20))
    for (int k = 0; k < 2048 && (uint64_t)(sp + k + 1) <= top; k++) { const uint64_t v = sp[k]; if (v >= lo && v < hi) { caller = v; break; } }
  }
  g_s[i].pc = pc; g_s[i].caller = caller; g_s[i].tid = (int)syscall(SYS_gettid);
}

/* ---- per-thread timers */
static long sample_ns(void) { const char *us = getenv("CPU_SAMPLER_US"); return (us ? atol(us) : 500) * 1000L; }
static void arm_this_thread(void)
{
  pthread_attr_t at;
  if (pthread_getattr_np(pthread_self(), &at) == 0) { void *lo = NULL; size_t sz = 0; if (pthread_attr_getstack(&at, &lo, &sz) == 0) t_stack_hi = (uint64_t)lo + sz; pthread_attr_destroy(&at); }
  struct sigevent sev; memset(&sev, 0, sizeof(sev));
  sev.sigev_notify = SIGEV_THREAD_ID; sev.sigev_signo = SIGPROF;
  sev._sigev_un._tid = (int)syscall(SYS_gettid);
  timer_t t;
  if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t) != 0) return;
  struct itimerspec its; its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = sample_ns(); its.it_value = its.it_interval;
  timer_settime(t, 0, &its, NULL);
}
struct start_arg { void *(*fn)(void *); void *arg; };
static void *thread_trampoline(void *p)
{
  struct start_arg a = *(struct start_arg *)p; free(p);
  sigset_t ss; sigemptyset(&ss); sigaddset(&ss, SIGPROF); pthread_sigmask(SIG_UNBLOCK, &ss, NULL);
  arm_this_thread();
  return a.fn(a.arg);
}
int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg)
{
  static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
  if (!real) real = (int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *))dlsym(RTLD_NEXT, "pthread_create");
  if (!g_s) return real(th, attr, fn, arg);
  struct start_arg *a = malloc(sizeof(*a)); a->fn = fn; a->arg = arg;
  return real(th, attr, thread_trampoline, a);
}

/* cpu_sampler_begin() / cpu_sampler_end(): bracket a region (bench.py calls them around its timed region when CPU_SAMPLER_REGION is
 * set): begin drops what was sampled so far and re-arms handler and timer (a library may have replaced either), end stops the timer */
void cpu_sampler_end(void) { g_on = 0; }
void cpu_sampler_begin(void)
{
  if (!g_s) return;
  struct sigaction sa; memset(&sa, 0, sizeof(sa));          /* (a library may have replaced the handler) */
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  const char *want = getenv("CPU_SAMPLER_LIB"); if (!want) want = "libkvazzup_amd";
  FILE *m = fopen("/proc/self/maps", "r");
  if (m) {
    char line[1024];
    while (fgets(line, sizeof(line), m)) if (strstr(line, "r-xp") && strstr(line, want)) { unsigned long long a, b; if (sscanf(line, "%llx-%llx", &a, &b) == 2) { g_lib_lo = a; g_lib_hi = b; } break; }
    fclose(m);
  }
  g_n = 0; g_on = 1;
}

__attribute__((constructor)) static void sampler_start(void)
{
  if (getenv("CPU_SAMPLER_OFF")) return;
  g_s = calloc(MAX_SAMPLES, sizeof(*g_s));
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  arm_this_thread();                                       /* the main thread; the others as they are created */
  g_on = getenv("CPU_SAMPLER_REGION") ? 0 : 1;             /* with a region: armed by cpu_sampler_begin() */
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
  for (long i = 0; i < n; i++) fprintf(o, "S %llx %d %llx\n", (unsigned long long)g_s[i].pc, g_s[i].tid, (unsigned long long)g_s[i].caller);
  fclose(o);
}
