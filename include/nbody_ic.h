/* nbody_ic.h — deterministic initial conditions for the all-pairs N-body path.
 *
 * The reference tree (/root/reference, 14 VHDL files) holds no host program and
 * therefore no initial-condition generator (SURVEY.md §0, §2.2 last row).  This
 * header is the build's own definition (SURVEY.md §7 step 1, §8(d) "Synthetic
 * inputs"): 3N position components then 3N velocity components, each uniform in
 * [-1, 1), pos.w = 1, vel.w = 0; the 16-byte body word {x, y, z, .} is the
 * reference's RAM word (S/top_level.vhd:206-208).
 *
 * The stream is SplitMix64 (counter-based: element k depends only on seed and
 * k), so C host code, the oracle and the numpy mirror in
 * mini_nbody_amd/bodies.py agree bit-for-bit and any shard of the bodies can be
 * generated without generating the rest.  It does not depend on libc rand().
 *
 * Header-only C99; safe to include from C and C++.
 */
#ifndef NBODY_IC_H
#define NBODY_IC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_IC_DEFAULT_SEED 42ull

/* k-th output of SplitMix64 started at `seed` (k = 0, 1, ...). */
static inline uint64_t nbody_ic_splitmix64(uint64_t seed, uint64_t k) {
  uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* Top 24 bits -> m in [0, 2^24); value (m - 2^23) / 2^23 in [-1, 1).
 * Exactly representable in binary32, hence identical in binary64. */
static inline float nbody_ic_uniform(uint64_t seed, uint64_t k) {
  int32_t m = (int32_t)(nbody_ic_splitmix64(seed, k) >> 40) - (1 << 23);
  return (float)m * (1.0f / 8388608.0f);
}

/* Fill bodies [first, first + count) of an n-body system into pos/vel
 * (count x 4 floats each).  Stream index of pos component c of body i is
 * 3*i + c; of vel component c it is 3*n + 3*i + c. */
static inline void nbody_ic_fill_f32(float *pos, float *vel, size_t n, size_t first, size_t count, uint64_t seed) {
  for (size_t b = 0; b < count; ++b) {
    size_t i = first + b;
    for (int c = 0; c < 3; ++c) {
      pos[4 * b + c] = nbody_ic_uniform(seed, 3ull * i + (uint64_t)c);
      vel[4 * b + c] = nbody_ic_uniform(seed, 3ull * n + 3ull * i + (uint64_t)c);
    }
    pos[4 * b + 3] = 1.0f;
    vel[4 * b + 3] = 0.0f;
  }
}

static inline void nbody_ic_fill_f64(double *pos, double *vel, size_t n, size_t first, size_t count, uint64_t seed) {
  for (size_t b = 0; b < count; ++b) {
    size_t i = first + b;
    for (int c = 0; c < 3; ++c) {
      pos[4 * b + c] = (double)nbody_ic_uniform(seed, 3ull * i + (uint64_t)c);
      vel[4 * b + c] = (double)nbody_ic_uniform(seed, 3ull * n + 3ull * i + (uint64_t)c);
    }
    pos[4 * b + 3] = 1.0;
    vel[4 * b + 3] = 0.0;
  }
}

#ifdef __cplusplus
}
#endif
#endif /* NBODY_IC_H */
