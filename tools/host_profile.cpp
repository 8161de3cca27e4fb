// PC-sampling profile of the host part of create_proof (circuit synthesis through the C++ mirror), no GPU and no perf(1)
// needed: SIGPROF every 0.5 ms of CPU time, instruction pointers dumped with /proc/self/maps; tools/host_profile.py
// attributes them to the functions (and, on request, the instructions) of libbellman_hip.so / libbellman_hip_test.so.
//   g++ -O2 -o tools/_build/host_profile tools/host_profile.cpp -ldl
//   tools/_build/host_profile bellman_amd/lib/libbellman_hip_test.so <mode 2|3|9> <log2 rounds> <repeats> <out prefix>
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

static uint64_t g_pc[1 << 20];
static volatile size_t g_n = 0;
static void on_prof(int, siginfo_t *, void *uc) {
  if (g_n < (1 << 20)) g_pc[g_n++] = ((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}
int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s libbellman_hip_test.so mode log2_rounds repeats out_prefix\n", argv[0]); return 2; }
  void *h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  typedef double (*fn_t)(int, size_t, uint64_t, int);
  typedef double (*cap_t)(int, size_t, uint64_t, size_t *);
  static cap_t capture = (cap_t)dlsym(h, "bh_test_capture_check");
  const int mode = atoi(argv[2]), reps = atoi(argv[4]);
  // mode 9: the structure capture (R1cs's constructor without the upload) followed by its self check
  fn_t synth = mode == 9 ? (fn_t)[](int k, size_t n, uint64_t seed, int) { size_t o[4]; return capture(k, n, seed, o); }
                         : (fn_t)dlsym(h, "bh_test_synthesis_ms");
  const size_t n = ((size_t)1 << atoi(argv[3])) - 1;
  synth(1, n, 99, mode);   // grows the recycled vectors
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, 0);
  struct itimerval it = {{0, 500}, {0, 500}}, off = {{0, 0}, {0, 0}};
  setitimer(ITIMER_PROF, &it, 0);
  double best = 1e30;
  for (int r = 0; r < reps; r++) { const double ms = synth(1, n, 99, mode); if (ms < best) best = ms; }
  setitimer(ITIMER_PROF, &off, 0);
  char path[512];
  snprintf(path, sizeof path, "%s.samples", argv[5]);
  FILE *f = fopen(path, "w");
  for (size_t i = 0; i < g_n; i++) fprintf(f, "%llx\n", (unsigned long long)g_pc[i]);
  fclose(f);
  snprintf(path, sizeof path, "%s.maps", argv[5]);
  f = fopen(path, "w");
  FILE *m = fopen("/proc/self/maps", "r");
  char line[1024];
  while (fgets(line, sizeof line, m)) fputs(line, f);
  fclose(m); fclose(f);
  printf("mode %d, 2^%s rounds: min %.1f ms over %d runs, %zu samples\n", mode, argv[3], best, reps, (size_t)g_n);
  return 0;
}
