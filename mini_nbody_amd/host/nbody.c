/* nbody.c — the host program, in C, over the C-ABI of include/nbody.h
 * (north_star: "host code stays in C ... calling hand-written HIP kernels
 * through a thin C-ABI shim").  The reference tree holds no host program
 * (SURVEY.md §0); this is the build's own: deterministic initial conditions,
 * the bodyForce()/integrate() loop or the device-resident step loop, timing
 * with the first iteration excluded as warm-up, and the metric of BASELINE.md
 * §2: billion pair-interactions/s = N^2 * timed_steps / seconds / 1e9.
 *
 * usage: nbody [N] [iters] [--gpus P] [--fp64] [--tile T] [--host-loop] [--seed S] [--strict] [--rtl] [--jsub K]
 *              [--sum seq|blocked] [--block K] [--one-launch | --two-launch] [--long-buffers 0|1] [--overlap 0|1|2] [--wsplit 1|4|16]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nbody.h"
#include "nbody_ic.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed: %s\n", #call, nbody_error_string(rc_)); return 1; } } while (0)

int main(int argc, char **argv) {
  int n = 30000, iters = 10, gpus = 1, fp64 = 0, tile = 0, host_loop = 0, strict = 0, rtl = 0, npos = 0, jsub = 0, sum = -1, block = 0, two_launch = -1, long_buffers = -1, overlap = -1, wsplit = 0;
  unsigned long long seed = NBODY_IC_DEFAULT_SEED;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--tile") && a + 1 < argc) tile = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--jsub") && a + 1 < argc) jsub = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--seed") && a + 1 < argc) seed = strtoull(argv[++a], NULL, 10);
    else if (!strcmp(argv[a], "--fp64")) fp64 = 1;
    else if (!strcmp(argv[a], "--host-loop")) host_loop = 1;
    else if (!strcmp(argv[a], "--strict")) strict = 1;
    else if (!strcmp(argv[a], "--rtl")) rtl = 1;
    else if (!strcmp(argv[a], "--two-launch")) two_launch = 1;
    else if (!strcmp(argv[a], "--one-launch")) two_launch = 0;
    else if (!strcmp(argv[a], "--overlap") && a + 1 < argc) overlap = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--long-buffers") && a + 1 < argc) long_buffers = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--wsplit") && a + 1 < argc) wsplit = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--sum") && a + 1 < argc) { ++a; sum = !strcmp(argv[a], "seq") ? NBODY_SUM_SEQ : NBODY_SUM_BLOCKED; }
    else if (!strcmp(argv[a], "--block") && a + 1 < argc) block = atoi(argv[++a]);
    else if (argv[a][0] != '-' && npos == 0) { n = atoi(argv[a]); npos++; }
    else if (argv[a][0] != '-' && npos == 1) { iters = atoi(argv[a]); npos++; }
    else { fprintf(stderr, "usage: %s [N] [iters] [--gpus P] [--fp64] [--tile T] [--host-loop] [--seed S] [--strict] [--rtl] [--jsub K] [--sum seq|blocked] [--block K] [--one-launch|--two-launch] [--long-buffers 0|1] [--overlap 0|1|2] [--wsplit 1|4|16]\n", argv[0]); return 2; }
  }
  if (n <= 0 || iters < 2) { fprintf(stderr, "need N > 0 and iters >= 2 (iteration 1 is warm-up)\n"); return 2; }
  const float dt = 0.01f;
  const size_t words = (size_t)n * 4;
  CHECK(nbody_init(n, gpus, fp64, tile));
  /* (a strict binary32 arithmetic is granted by the library only after every device of the context has proved the strict 1/sqrt for
     every positive normal binary32 — nbody_strict_proof, 10 ms per device; otherwise the option call fails and CHECK reports why) */
  if (strict) CHECK(nbody_set_option(NBODY_OPT_ARITH, NBODY_ARITH_STRICT));   /* IEEE-exact: bit-identical to the CPU oracle */
  if (rtl) {
    /* the RTL-faithful result: the reference's own five roundings for d2 (S/dxy.vhd:113-122, S/dzsoft.vhd:201-202,
       S/dxyz_soft.vhd:149-150), 1/sqrt rounded once, and its own summation — 16 interleaved partial sums per axis over ONE stream of all
       N sources, latched rotated, joined by the adder tree (S/fxyz.vhd:129-184, S/final_adder.vhd:88-104, S/top_level.vhd:233-254) */
    CHECK(nbody_set_option(NBODY_OPT_ARITH, NBODY_ARITH_REFERENCE_STRICT));
    CHECK(nbody_set_option(NBODY_OPT_SUM_ORDER, NBODY_SUM_FPGA16));
    CHECK(nbody_set_option(NBODY_OPT_JSUB, 1));
  }
  if (jsub > 0) CHECK(nbody_set_option(NBODY_OPT_JSUB, jsub));                 /* source segments (summation order) */
  if (sum >= 0) CHECK(nbody_set_option(NBODY_OPT_SUM_ORDER, sum));             /* one sequential sum per segment, or blocks */
  if (block > 0) CHECK(nbody_set_option(NBODY_OPT_SUM_BLOCK, block));
  if (two_launch >= 0) CHECK(nbody_set_option(NBODY_OPT_FUSE_COMBINE, !two_launch));   /* in-launch combine or separate kernel (same bits) */
  if (long_buffers >= 0) CHECK(nbody_set_option(NBODY_OPT_ISA_LONG_BUFFERS, long_buffers));
  if (wsplit > 0) CHECK(nbody_set_option(NBODY_OPT_WSPLIT, wsplit));              /* waves of a workgroup share rows and split the segment (4) or not (1) */
  if (overlap >= 0) CHECK(nbody_set_option(NBODY_OPT_OVERLAP, overlap));        /* multi-GPU: 0 gather first, 1 own slice then the rest, 2 per arriving slice */

  double total = 0.0;
  if (!fp64) {
    float *buf = (float *)malloc(2 * words * sizeof(float));
    if (!buf) return 3;
    BodySystem p = { buf, buf + words };
    nbody_ic_fill_f32(p.pos, p.vel, (size_t)n, 0, (size_t)n, seed);
    if (host_loop) {
      for (int it = 1; it <= iters; ++it) {
        double t0 = now_s();
        CHECK(bodyForce(p.pos, p.vel, dt, n));   /* compute interbody forces, kick */
        CHECK(integrate(p.pos, p.vel, dt, n));   /* drift */
        double t = now_s() - t0;
        if (it > 1) total += t;                  /* first iteration is warm-up */
      }
    } else {
      CHECK(nbody_upload(&p));
      CHECK(nbody_step(dt, 1));
      CHECK(nbody_sync());
      double t0 = now_s();
      CHECK(nbody_step(dt, iters - 1));
      CHECK(nbody_sync());
      total = now_s() - t0;
      CHECK(nbody_download(&p));
    }
    double cx = 0, cy = 0, cz = 0;
    for (int i = 0; i < n; ++i) { cx += p.pos[4 * i]; cy += p.pos[4 * i + 1]; cz += p.pos[4 * i + 2]; }
    printf("checksum (sum of positions): %.9g %.9g %.9g\n", cx, cy, cz);
    free(buf);
  } else {
    double *buf = (double *)malloc(2 * words * sizeof(double));
    if (!buf) return 3;
    BodySystemD p = { buf, buf + words };
    nbody_ic_fill_f64(p.pos, p.vel, (size_t)n, 0, (size_t)n, seed);
    if (host_loop) {
      for (int it = 1; it <= iters; ++it) {
        double t0 = now_s();
        CHECK(bodyForce_d(p.pos, p.vel, dt, n));
        CHECK(integrate_d(p.pos, p.vel, dt, n));
        double t = now_s() - t0;
        if (it > 1) total += t;
      }
    } else {
      CHECK(nbody_upload_d(&p));
      CHECK(nbody_step_d(dt, 1));
      CHECK(nbody_sync());
      double t0 = now_s();
      CHECK(nbody_step_d(dt, iters - 1));
      CHECK(nbody_sync());
      total = now_s() - t0;
      CHECK(nbody_download_d(&p));
    }
    double cx = 0, cy = 0, cz = 0;
    for (int i = 0; i < n; ++i) { cx += p.pos[4 * i]; cy += p.pos[4 * i + 1]; cz += p.pos[4 * i + 2]; }
    printf("checksum (sum of positions): %.17g %.17g %.17g\n", cx, cy, cz);
    free(buf);
  }
  double avg = total / (double)(iters - 1);
  long long info_r = 0, info_s = 0, info_b = 0, info_l = 0, info_w = 0;
  nbody_get_info(NBODY_INFO_WSPLIT, &info_w);
  nbody_get_info(NBODY_INFO_IBLOCK, &info_r);
  nbody_get_info(NBODY_INFO_NSEG, &info_s);
  nbody_get_info(NBODY_INFO_SUM_BLOCK, &info_b);
  nbody_get_info(NBODY_INFO_LAUNCHES_PER_STEP, &info_l);
  printf("%d Bodies (%s, %d GPU%s, %s loop, %lld bodies/lane, %lld segments x %lld pieces, sum block %lld, %lld launch%s/step): average %0.3f Billion Interactions / second (%.3f ms / step)\n",
         n, fp64 ? "fp64" : "fp32", gpus, gpus > 1 ? "s" : "", host_loop ? "host" : "device", info_r, info_s, info_w, info_b, info_l,
         info_l > 1 ? "es" : "", 1e-9 * (double)n * (double)n / avg, 1e3 * avg);
  nbody_shutdown();
  return 0;
}
