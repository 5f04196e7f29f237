/* nbody_cpu.c — BASELINE config 1 ("N=4096 fp32, 10 iters, reference nbody.c on host CPU: plumbing, no GPU").
 *
 * TEST INFRASTRUCTURE / CPU BASELINE, not product code: the reference tree has no nbody.c, so this is
 * the oracle (nbody_ref.c) behind the same command line, initial conditions, loop and report line as the GPU host
 * program mini_nbody_amd/host/nbody.c — which makes the two directly comparable (same checksum in --strict mode).
 *
 * Summation order: --sum seq (default: what a plain CPU nbody.c does, one accumulator per body) or
 * --sum blocked [--block K] [--segments S] [--wsplit W]: the GPU engine's order (blocks of K sources, S source segments
 * of W pieces each).
 *
 * usage: nbody_cpu [N] [iters] [--fp64] [--seed S] [--divsqrt] [--rtl] [--threads T] [--sum seq|blocked] [--block K] [--segments S] [--wsplit W]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nbody_ref.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
  int n = 4096, iters = 10, fp64 = 0, npos = 0, rsq = REF_RSQRT_F64, rtl = 0, threads = 0, sum = REF_SUM_SEQ, block = 1024, segments = 1, wsplit = 1;
  unsigned long long seed = 42ull;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--fp64")) fp64 = 1;
    else if (!strcmp(argv[a], "--divsqrt")) rsq = REF_RSQRT_DIVSQRT;
    else if (!strcmp(argv[a], "--rtl")) rtl = 1;   /* the RTL's five roundings for d2 and its 16 partials + adder tree over one stream of all N sources */
    else if (!strcmp(argv[a], "--seed") && a + 1 < argc) seed = strtoull(argv[++a], NULL, 10);
    else if (!strcmp(argv[a], "--threads") && a + 1 < argc) threads = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--sum") && a + 1 < argc) { ++a; sum = !strcmp(argv[a], "blocked") ? REF_SUM_BLOCKED : REF_SUM_SEQ; }
    else if (!strcmp(argv[a], "--block") && a + 1 < argc) block = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--segments") && a + 1 < argc) segments = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--wsplit") && a + 1 < argc) wsplit = atoi(argv[++a]);
    else if (argv[a][0] != '-' && npos == 0) { n = atoi(argv[a]); npos++; }
    else if (argv[a][0] != '-' && npos == 1) { iters = atoi(argv[a]); npos++; }
    else { fprintf(stderr, "usage: %s [N] [iters] [--fp64] [--seed S] [--divsqrt] [--rtl] [--threads T] [--sum seq|blocked] [--block K] [--segments S] [--wsplit W]\n", argv[0]); return 2; }
  }
  if (n <= 0 || iters < 2) { fprintf(stderr, "need N > 0 and iters >= 2 (iteration 1 is warm-up)\n"); return 2; }
  if (threads <= 0) {   /* small problems: do not wake more threads than there are 256-body chunks of rows */
    int cap = n / 256 > 0 ? n / 256 : 1;
    threads = ref_num_threads() < cap ? ref_num_threads() : cap;
  }
  ref_set_num_threads(threads);
  const float dt = 0.01f;
  double total = 0.0, cx = 0, cy = 0, cz = 0;
  if (!fp64) {
    float *pos = (float *)malloc(sizeof(float) * 4 * (size_t)n), *vel = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    ref_ic_f32(pos, vel, n, 0, n, seed);
    const ref_order_t order = {REF_D2_FMA3, rsq, sum, block, 1, segments > 0 ? segments : 1, wsplit > 0 ? wsplit : 1};
    for (int it = 1; it <= iters; ++it) {
      double t0 = now_s();
      if (rtl) ref_step_f32(pos, vel, dt, n, 1, REF_D2_REFERENCE, rsq, REF_SUM_FPGA16);
      else ref_step_f32_order(pos, vel, dt, n, 1, &order);   /* bodyForce (kick) + integrate (drift) */
      if (it > 1) total += now_s() - t0;
    }
    for (int i = 0; i < n; ++i) { cx += pos[4 * i]; cy += pos[4 * i + 1]; cz += pos[4 * i + 2]; }
    printf("checksum (sum of positions): %.9g %.9g %.9g\n", cx, cy, cz);
    free(pos); free(vel);
  } else {
    double *pos = (double *)malloc(sizeof(double) * 4 * (size_t)n), *vel = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    ref_ic_f64(pos, vel, n, 0, n, seed);
    /* the engine's fp64 order: segments x pieces of the wave split, one sequential sum per piece (1 x 1: the plain sequential sum) */
    const ref_order_t order64 = {REF_D2_FMA3, REF_RSQRT_F64, REF_SUM_SEQ, block, 1, segments > 0 ? segments : 1, wsplit > 0 ? wsplit : 1};
    for (int it = 1; it <= iters; ++it) {
      double t0 = now_s();
      ref_step_f64_order(pos, vel, (double)dt, n, 1, &order64);
      if (it > 1) total += now_s() - t0;
    }
    for (int i = 0; i < n; ++i) { cx += pos[4 * i]; cy += pos[4 * i + 1]; cz += pos[4 * i + 2]; }
    printf("checksum (sum of positions): %.17g %.17g %.17g\n", cx, cy, cz);
    free(pos); free(vel);
  }
  double avg = total / (double)(iters - 1);
  printf("%d Bodies (%s, host CPU, %d threads): average %0.3f Billion Interactions / second (%.3f ms / step)\n", n,
         fp64 ? "fp64" : "fp32", ref_num_threads(), 1e-9 * (double)n * (double)n / avg, 1e3 * avg);
  return 0;
}
