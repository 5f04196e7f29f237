/* mailbox_driver.c — the PS-side driver of the reference's memory-mapped mailbox, in plain C over the C-ABI (include/nbody.h): what a
 * maintainer of the reference would write to use the MI355X engine INSTEAD OF the PL block (compiled and tested).
 *
 * The reference (/root/reference, "S/" = vec_add.srcs/sources_1/new/) has no host program; its PS side would do exactly this:
 *   write the bodies to words 1..N of RAM A        {x, y, z, ignored}, 16 bytes each     S/top_level.vhd:206-208
 *   write word 0 = {bit 0 BEGIN, bits 46:32 NUM_PTS}                                     S/top_level.vhd:184-185
 *   poll word 0 until BEGIN reads 0; bits 63:32 then hold the elapsed 1000-clock ticks   S/top_level.vhd:146, 255-263
 *   read the forces {Fx, Fy, Fz, 0} of bodies 1..N from words 1..N of RAM B (word 0 is never written)   S/compute_store.vhd:213, 221-242
 * ONE power-up, then any number of requests of any size (NUM_PTS is sampled per request, S/top_level.vhd:180-186).
 *
 * usage: mailbox_driver [--served] [--timed] [--seed S] N1 [N2 ...]
 *   --served  no call per request: nbody_mailbox_serve() lets a library thread play the FSM; the driver only writes and polls memory
 *   --timed   the engine's timed arithmetic instead of the PL block's own bits (faithful = 0)
 * prints, per request, NUM_PTS, the tick word and a checksum (sum of the force components in double), e.g. for the tests to compare with
 * the oracle in the same mode.  Bodies: the repository's seeded generator (include/nbody_ic.h), the first N of the seed's stream. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody.h"
#include "nbody_ic.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed: %s\n", #call, nbody_error_string(rc_)); return 1; } } while (0)

static uint32_t *ram_a;   /* the two block-RAM images of S/top_level.vhd:100-117, as the PS sees them */
static float *ram_b;

int main(int argc, char **argv) {
  int served = 0, faithful = 1, nreq = 0, sizes[64];
  uint64_t seed = NBODY_IC_DEFAULT_SEED;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--served")) served = 1;
    else if (!strcmp(argv[a], "--timed")) faithful = 0;
    else if (!strcmp(argv[a], "--seed") && a + 1 < argc) seed = strtoull(argv[++a], NULL, 0);
    else if (argv[a][0] != '-' && nreq < 64) sizes[nreq++] = atoi(argv[a]);
    else { fprintf(stderr, "usage: %s [--served] [--timed] [--seed S] N1 [N2 ...]\n", argv[0]); return 2; }
  }
  if (nreq == 0) { sizes[nreq++] = 9; sizes[nreq++] = 100; sizes[nreq++] = 32767; sizes[nreq++] = 0; sizes[nreq++] = 40; }
  int capacity = 0;
  /* power-up of the PL block, once: RAMs of ram_depth - 1 = 32767 body words (S/top_level.vhd:45) */
  CHECK(nbody_mailbox_open(/*capacity*/0, faithful));
  CHECK(nbody_mailbox_rams((void **)&ram_a, (void **)&ram_b, &capacity));
  if (served) CHECK(nbody_mailbox_serve(1, /*clock_khz*/300000));
  float *pos = (float *)malloc((size_t)capacity * 4 * sizeof(float)), *vel = (float *)malloc((size_t)capacity * 4 * sizeof(float));
  if (!pos || !vel) return 3;
  for (int r = 0; r < nreq; ++r) {
    const int n = sizes[r];
    if (n < 0 || n > capacity) { fprintf(stderr, "NUM_PTS %d is outside 0..%d\n", n, capacity); return 2; }
    if (n > 0) nbody_ic_fill_f32(pos, vel, (size_t)n, 0, (size_t)n, seed);
    for (int k = 0; k < (capacity + 1) * 4; ++k) ((uint32_t *)ram_b)[k] = 0xDEADBEEFu;   /* to show that word 0 and the words > N are left alone */
    memcpy(ram_a + 4, pos, (size_t)n * 16);                  /* bodies: words 1..N                  S/top_level.vhd:206-208 */
    ram_a[1] = (uint32_t)n; ram_a[2] = 0; ram_a[3] = 0;      /* NUM_PTS in bits [46:32]             S/top_level.vhd:185 */
    if (served) {
      __atomic_store_n(&ram_a[0], 1u, __ATOMIC_RELEASE);                   /* BEGIN, last          S/top_level.vhd:184 */
      while (__atomic_load_n(&ram_a[0], __ATOMIC_ACQUIRE) & 1u) { }        /* poll word 0          S/top_level.vhd:255-263 */
      if (ram_a[3]) { fprintf(stderr, "request refused: %s\n", nbody_error_string((int)ram_a[3])); return 1; }
    } else {
      ram_a[0] = 1u;
      CHECK(nbody_mailbox_run(ram_a, ram_b, /*clock_khz*/300000));
    }
    double cx = 0, cy = 0, cz = 0;
    for (int k = 1; k <= n; ++k) { cx += ram_b[4 * k]; cy += ram_b[4 * k + 1]; cz += ram_b[4 * k + 2]; }   /* force of body k: word k */
    int untouched = 1;
    for (int k = 0; k < 4; ++k) untouched &= ((uint32_t *)ram_b)[k] == 0xDEADBEEFu;
    for (int k = (n + 1) * 4; k < (capacity + 1) * 4; ++k) untouched &= ((uint32_t *)ram_b)[k] == 0xDEADBEEFu;
    printf("NUM_PTS %d  BEGIN %u  ticks %u  checksum (sum of forces): %.9g %.9g %.9g  RAM B word 0 and beyond word N untouched: %s\n",
           n, ram_a[0] & 1u, ram_a[1], cx, cy, cz, untouched ? "yes" : "NO");
  }
  free(pos); free(vel);
  nbody_shutdown();   /* (stops the service thread first) */
  return 0;
}
