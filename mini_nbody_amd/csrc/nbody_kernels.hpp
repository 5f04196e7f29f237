// nbody_kernels.hpp — CDNA4 (gfx950) kernels of the all-pairs force path.
//
// One lane owns R "this" bodies i (the reference's 12 spatial lanes,
// S/top_level.vhd:44, 187-232, become 64 x R per wave); every lane of a wave
// sees the same "target" body j at the same time (the reference broadcasts
// TRGT(0..2) to all lanes, S/top_level.vhd:233-254, 284).  Per pair, exactly
// the arithmetic of S/fxyz.vhd:97-127:
//     d = r_j - r_i                      3 v_sub_f32            S/dxy.vhd:94-98, S/dzsoft.vhd:186-187
//     d2 = dx*dx + dy*dy + dz*dz + eps   3 v_fma_f32            S/dxy.vhd:113-122, S/dzsoft.vhd:201-202, S/dxyz_soft.vhd:149-150
//     inv = d2^(-1/2)                    1 v_rsq_f32            S/fxyz.vhd:101-102
//     inv3 = inv * (inv * inv)           2 v_mul_f32            S/cube.vhd:66-70
//     F += d * inv3                      3 v_fma_f32            S/fxyz.vhd:120-127
// = 11 full-rate VALU (2 cycles per wave64 on a SIMD) + 1 quarter-rate
// transcendental (8 cycles): 30 cycles per 64 pairs per SIMD by the instruction
// costs of profiles/r01_microbench_valu_issue.txt; inside the real kernel 33.0-33.4
// (the full-rate instructions that read an SGPR cost 2.27, the transcendental 8.4:
// profiles/r02_loop_diagnostics.md).  That issue count, not HBM and not MFMA, bounds
// the kernel.  v_pk_*_f32 cost 4 cycles on gfx950, so packed math buys nothing and is
// kept out (-fno-slp-vectorize).
//
// The three variants differ only in how r_j reaches the lanes:
//   SMEM      wave-uniform scalar loads into SGPRs; VALU reads them as scalar operands (no instruction spent)
//   LDS       TILE bodies staged in LDS, every lane reads the same address (broadcast ds_read_b128)
//   READLANE  each lane holds one body of a 64-body wave tile; v_readlane_b32 x3 per source (4 cycles each)
// All of them add the sources of a segment in the same order, so they return
// identical bits.  The order inside a segment (ForceArgs::sum_block = K > 0, the default K = 1024): level 1 sums a
// block of K consecutive sources from zero with one fma per term (S/fxyz.vhd:120-127), level 2 adds the finished
// block sums in ascending order into a second accumulator.  The reference does not run one long sequential sum
// either — it keeps 16 partial sums and joins them with an adder tree (S/fxyz.vhd:129-145, S/final_adder.vhd:88-104) —
// and for the same reason: at N = 2^20 a single fp32 accumulator is off by 1e-4 of the force, the two-level sum by
// 3e-7 (profiles/r02_error_budget.md).  Cost: 3 v_add + 7 v_mov per 12288 VALU instructions.  K = 0 keeps the plain
// sequential sum (study mode; what a CPU nbody.c does).
//
// Who walks what (ForceArgs::wsplit, round 3): a workgroup owns 64 rows and its 4 or 16 waves walk one piece of the segment
// each for those same rows; the piece sums are joined through LDS in ascending source order — a third level of the sum —
// so the same waves need a quarter (a sixteenth) of the global partial sums.  wsplit = 1 is round 2's layout (256*R rows per
// workgroup, every wave walks the whole segment), which the LDS and READLANE kernels keep.  In the FPGA order a 16-wave workgroup
// means something else: wave k holds the reference's partial sum k of the workgroup's 64 rows (force_fpga16w_f32).
//
// How a row's force is finished (ForceArgs::finish): with one segment the kernel applies kick and drift itself;
// with several, every 64-row unit stores its partial sums and the LAST to arrive for its rows (an agent-scope ticket
// per 64 rows) adds the partials in ascending segment order and applies kick and drift — one launch per
// step, and the result does not depend on who came last; or (small launches) a combine kernel does, same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nbody_args.hpp"

namespace nbk {

// v_readlane_b32: the value lane k holds, as a wave-uniform scalar.  Takes the float BY VALUE:
// __builtin_bit_cast applied directly to an ext-vector element (v.y) reads element 0 with this compiler.
__device__ __forceinline__ float lane_bcast(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

__device__ __forceinline__ float soft_f32() { return __builtin_bit_cast(float, kSoftBits); }

// ---------------------------------------------------------------------------
// one pair, fp32.  ARITH bit 0: the RTL's five roundings for d2 instead of three fused ones;
// bit 1: "strict" 1/sqrt, rounded once from an fp64 evaluation with IEEE sqrt and divide — the same
// value oracle/nbody_ref.c computes (REF_RSQRT_F64), so a strict run matches the oracle bit for bit.
constexpr int kArithRef = 1, kArithStrict = 2;

// The strict 1/sqrt as the oracle writes it (oracle/nbody_ref.c REF_RSQRT_F64): IEEE square root and divide in binary64, rounded once to
// binary32.  About 150 issue cycles per wave.
__device__ __forceinline__ float rsqrt_ieee_f32(float x) { return (float)(1.0 / __builtin_sqrt((double)x)); }

// The same VALUE from eight binary32 operations for all but ~2^-16 of the arguments (round 4).  y = v_rsq_f32(x) is within 1 ulp, so the
// answer is y or a neighbour: with e = 1 - x y^2 (|e| < 2^-21), x^(-1/2) = y (1 - e)^(-1/2) = y + y e/2 + y 3e^2/8 + ..., and ONE fma
// y + (y/2) e rounds the corrected value to binary32 correctly — provided e is known to ~2^-43, which binary32 delivers because the
// product x y is carried as an exact pair (hi + lo = x y, an fma's error term) and 1 - hi y, - lo y are both tiny: each fma rounds at
// <= 2^-46.  What is neglected (3e^2/8 <= 2^-44.4, the roundings of e, the oracle's own two binary64 roundings <= 2^-51.5) moves the
// value by < 2^-43 y, which can only change the rounding if it lies that close to the midpoint of two binary32 neighbours.  That case is
// DETECTED, not assumed away: the final fma is evaluated twice, with the factor 1/2 widened and narrowed by 2^-16 (a band of >= 2^-41 y about the
// correction where a midpoint can matter, 4x what is neglected even for a 2-ulp seed); rounding is monotone, so if both give the same
// binary32 every value between them does, the oracle's included.  Where they differ (about 2^-16 of arguments; also every NaN, since a NaN
// compares unequal to itself: x = inf, NaN or negative) the wave computes the IEEE form — one wave in ~1000 per source.  The seed's
// accuracy is part of the contract (a seed off by 2^-20 ... 2^-9 would pass the band with a wrong value: tests/test_strict_rsqrt.py
// test_model_limits), so it is proved on the hardware rather than on paper: nbody_rsqrt_selftest() compares this with rsqrt_ieee_f32
// for EVERY binary32 bit pattern (tests/test_strict_rsqrt.py, -m gpu).
constexpr float kStrictBand = 0x1p-16f;
__device__ __forceinline__ bool rsqrt_fast_f32(float x, float& r) {   // true: r is the strict value; false: r is unspecified
  float y = __builtin_amdgcn_rsqf(x);
  float hi = x * y;
  float lo = __builtin_fmaf(x, y, -hi);
  float e = __builtin_fmaf(-hi, y, 1.0f);
  e = __builtin_fmaf(-lo, y, e);
  float t = y * e;                                    // rounded once more (2^-24 of the correction: far inside the band)
  r = __builtin_fmaf(t, 0.5f + kStrictBand, y);
  float r2 = __builtin_fmaf(t, 0.5f - kStrictBand, y);
  return r == r2;
}
__device__ __forceinline__ float rsqrt_strict_f32(float x) {
  float r;
  bool ok = rsqrt_fast_f32(x, r);
  if (__builtin_amdgcn_ballot_w64(!ok) != 0) {      // wave-uniform: no lane pays for the IEEE form unless one of the 64 needs it
    float xs = x;
    asm volatile("" : "+v"(xs));                      // pins the IEEE form inside the branch: without it the compiler evaluates it
    float s = rsqrt_ieee_f32(xs);                     // speculatively for every pair and selects (seen in force_lds_f32, force_readlane_f32)
    r = ok ? r : s;
  }
  return r;
}

// nbody_rsqrt_selftest(): bit patterns first .. first+count-1, one per thread per round; out[0] = patterns where the eight-operation result
// was accepted and differs from the IEEE form (must stay 0), out[1] = patterns sent to the IEEE form, out[2] = smallest offending pattern + 1
__global__ void __launch_bounds__(256) rsqrt_selftest_kernel(unsigned first, unsigned long long count, unsigned long long* out) {
  unsigned long long bad = 0, slow = 0, worst = ~0ull;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (unsigned long long)gridDim.x * blockDim.x) {
    unsigned bits = first + (unsigned)k;
    float x = __builtin_bit_cast(float, bits), r;
    bool ok = rsqrt_fast_f32(x, r);
    float s = rsqrt_ieee_f32(x);
    if (!ok) { ++slow; continue; }
    if (__builtin_bit_cast(unsigned, r) != __builtin_bit_cast(unsigned, s)) { ++bad; if ((unsigned long long)bits < worst) worst = bits; }
  }
  if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[2], worst); }
  if (slow) atomicAdd(&out[1], slow);
}
// the strict 1/sqrt of an array, as the force kernels evaluate it (which = 0) or in the IEEE form alone (which = 1)
__global__ void __launch_bounds__(256) rsqrt_array_kernel(const float* x, float* y, int n, int which) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? x[i] : 1.0f;      // every lane stays in the wave-uniform branch
  float r = which ? rsqrt_ieee_f32(v) : rsqrt_strict_f32(v);
  if (i < n) y[i] = r;
}

template <int ARITH>
__device__ __forceinline__ void pair_f32(float xj, float yj, float zj, float xi, float yi, float zi, float eps,
                                         float& ax, float& ay, float& az) {
  float dx = xj - xi;   // S/dxy.vhd:94-95: a = target, b = this
  float dy = yj - yi;
  float dz = zj - zi;
  float d2;
  if constexpr (ARITH & kArithRef) {
    float sxy = dx * dx + dy * dy;                 // S/dxy.vhd:113-122 (compiled with -ffp-contract=off)
    float sz = __builtin_fmaf(dz, dz, eps);        // S/dzsoft.vhd:201-202
    d2 = sxy + sz;                                 // S/dxyz_soft.vhd:149-150
  } else {
    d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, __builtin_fmaf(dz, dz, eps)));
  }
  float inv;
  if constexpr (ARITH & kArithStrict) inv = rsqrt_strict_f32(d2);
  else inv = __builtin_amdgcn_rsqf(d2);            // v_rsq_f32, 1 ulp; d2 >= eps is never subnormal
  float inv2 = inv * inv;                          // S/cube.vhd:66-67
  float inv3 = inv * inv2;                         // S/cube.vhd:69-70
  ax = __builtin_fmaf(dx, inv3, ax);               // S/fxyz.vhd:120-127
  ay = __builtin_fmaf(dy, inv3, ay);
  az = __builtin_fmaf(dz, inv3, az);
}

// K pairs of one lane against K sources at once, accumulated IN THE ORDER k = 0 .. K-1 — the same values and the same sums as K calls of
// pair_f32.  Why it exists (round 6): the strict 1/sqrt ends in a wave-uniform branch (rsqrt_strict_f32), and a branch per pair is a
// scheduling barrier per pair — the compiler cannot interleave the K independent chains, and a launch with one wave per SIMD runs each
// pair's ~18 dependent operations back to back (N = 4096 in the 16-row FPGA kernel: 220 cycles per pair against 57 of issue).  Here the
// fast path of all K pairs is straight-line code, the K "not decided" flags are gathered in one mask, and ONE wave-uniform branch per K
// pairs sends the rare undecided arguments (3e-5 of them) to the IEEE form.  Measured in the 16-row FPGA kernel, one wave per SIMD
// (profiles/r06_mailbox_kernel_trace.txt): N = 4096 26.3 -> 21.7 us, N = 1024 8.6 -> 7.3 us.  Forcing all K pairs into lock-step with
// scheduling barriers between the stages returned nothing more (21.2 us): a wave alone on its SIMD issues one VALU instruction per ~8
// cycles whatever their dependences.  At full occupancy grouping LOSES (the sixteen-wave kernel at N = 32767 in two groups of four:
// 487 -> 583 us; eight waves per SIMD hide the chains anyway, profiles/r05_strict_loop.md): only the 16-row kernel uses it.
template <int ARITH, int K>
__device__ __forceinline__ void pairs_f32(const f4 (&p)[K], float xi, float yi, float zi, float eps, float& ax, float& ay, float& az) {
  if constexpr (!(ARITH & kArithStrict)) {
#pragma unroll
    for (int k = 0; k < K; ++k) pair_f32<ARITH>(p[k].x, p[k].y, p[k].z, xi, yi, zi, eps, ax, ay, az);
  } else {
    float dx[K], dy[K], dz[K], d2[K], inv[K];
    unsigned undecided = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      dx[k] = p[k].x - xi; dy[k] = p[k].y - yi; dz[k] = p[k].z - zi;
      if constexpr (ARITH & kArithRef) {
        const float sxy = dx[k] * dx[k] + dy[k] * dy[k];
        const float sz = __builtin_fmaf(dz[k], dz[k], eps);
        d2[k] = sxy + sz;
      } else {
        d2[k] = __builtin_fmaf(dx[k], dx[k], __builtin_fmaf(dy[k], dy[k], __builtin_fmaf(dz[k], dz[k], eps)));
      }
      if (!rsqrt_fast_f32(d2[k], inv[k])) undecided |= 1u << k;
    }
    if (__builtin_amdgcn_ballot_w64(undecided != 0u) != 0) {      // wave-uniform, once per K pairs
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool mine = (undecided >> k) & 1u;
        if (__builtin_amdgcn_ballot_w64(mine) != 0) {
          float xs = d2[k];
          asm volatile("" : "+v"(xs));                            // keeps the IEEE form inside the branch (see rsqrt_strict_f32)
          const float s = rsqrt_ieee_f32(xs);
          inv[k] = mine ? s : inv[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float inv2 = inv[k] * inv[k];
      const float inv3 = inv[k] * inv2;
      ax = __builtin_fmaf(dx[k], inv3, ax);
      ay = __builtin_fmaf(dy[k], inv3, ay);
      az = __builtin_fmaf(dz[k], inv3, az);
    }
  }
}

// fp64: x^(-3/2) straight from the v_rsq_f64 seed y (about 2^-24 relative) by ONE third-order step on the cube: with e = 1 - x*y^2,
//   x^(-3/2) = y^3 (1 - e)^(-3/2) = y^3 (1 + 3/2 e + 15/8 e^2 + 35/16 e^3 + ...),   inv3 = y^3 + y^3 * e * (3/2 + 15/8 e)
// leaves (35/16) e^3 < 2^-70: full binary64 in SIX operations (round 1: two Newton steps on y, then the cube: nine; rounds 2-3: one
// third-order step on y, then the cube: seven).  Same expression tree as oracle/nbody_ref.c ref_forces_f64 apart from how the inverse
// cube is obtained, and the same operations in the same order as the hand-scheduled fp64 loop (tools/gen_force_loop.py body_f64).
__device__ __forceinline__ double inv3_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double y2 = y * y;
  double e = __builtin_fma(-x, y2, 1.0);
  double y3 = y2 * y;
  double p = __builtin_fma(e, 1.875, 1.5);
  double q = e * p;
  return __builtin_fma(y3, q, y3);
}
// STRICT (NBODY_ARITH_STRICT in an fp64 context): 1/sqrt as IEEE square root and divide, both correctly rounded — the expression
// oracle/nbody_ref.c evaluates — so that an fp64 run matches the oracle BIT FOR BIT in whatever summation order is configured
// (the fp32 strict mode relies on the same two operations).  About 5x the cost of the seeded form; a parity mode, not the timed one.
template <int STRICT = 0>
__device__ __forceinline__ void pair_f64(double xj, double yj, double zj, double xi, double yi, double zi, double eps,
                                         double& ax, double& ay, double& az) {
  double dx = xj - xi, dy = yj - yi, dz = zj - zi;
  double d2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, __builtin_fma(dz, dz, eps)));
  double inv3;
  if constexpr (STRICT) {
    double inv = 1.0 / __builtin_sqrt(d2);
    double inv2 = inv * inv;
    inv3 = inv * inv2;
  } else {
    inv3 = inv3_f64(d2);
  }
  ax = __builtin_fma(dx, inv3, ax);
  ay = __builtin_fma(dy, inv3, ay);
  az = __builtin_fma(dz, inv3, az);
}

// S/final_adder.vhd:88-104: pairwise tree over 16 leaves
__device__ __forceinline__ float tree16(const float* p) {
  float l3[8], l2[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) l3[j] = p[2 * j] + p[2 * j + 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) l2[j] = l3[2 * j] + l3[2 * j + 1];
  return (l2[0] + l2[1]) + (l2[2] + l2[3]);
}

// ---------------------------------------------------------------------------
// Level-1 / level-2 accumulators of R rows.
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int R>
struct Sums {
  T ax[R], ay[R], az[R];   // level 1: the running block
  T bx[R], by[R], bz[R];   // level 2: finished blocks
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int r = 0; r < R; ++r) { ax[r] = ay[r] = az[r] = (T)0; bx[r] = by[r] = bz[r] = (T)0; }
  }
  __device__ __forceinline__ void fold() {   // a block is finished
#pragma unroll
    for (int r = 0; r < R; ++r) {
      bx[r] = bx[r] + ax[r]; by[r] = by[r] + ay[r]; bz[r] = bz[r] + az[r];
      ax[r] = ay[r] = az[r] = (T)0;
    }
  }
  // the segment's sum: sequential mode = level 1 itself; blocked mode = level 2 after the last (partial) block
  __device__ __forceinline__ void close(bool blocked, bool open_block) {
    if (!blocked) {
#pragma unroll
      for (int r = 0; r < R; ++r) { bx[r] = ax[r]; by[r] = ay[r]; bz[r] = az[r]; }
    } else if (open_block) {
      fold();
    }
  }
};

// Walks a segment [jb, je) block by block: body(j0, j1) must add the sources [j0, j1) in ascending order to level 1.
template <typename T, int R, typename Body>
__device__ __forceinline__ void walk_blocks(const ForceArgs& a, int jb, int je, Sums<T, R>& s, Body body) {
  if (a.sum_block <= 0) { body(jb, je); s.close(false, false); return; }
  int j0 = jb;
  for (; j0 + a.sum_block <= je; j0 += a.sum_block) { body(j0, j0 + a.sum_block); s.fold(); }
  if (j0 < je) body(j0, je);
  s.close(true, j0 < je);
}

// ---------------------------------------------------------------------------
// What happens to a finished force (S/compute_store.vhd:203-242 writes {Fx,Fy,Fz,0}; kick and drift are the
// north_star's bodyForce()/integrate(), one rounding each).
// (v = the row's velocity word, already loaded: the last arriver of the in-launch combine fetches it together with the
// partial sums instead of after them)
template <typename T, typename V4, typename Args>
__device__ __forceinline__ void apply_force(const Args& a, int i, V4 me, T fx, T fy, T fz, V4 v) {
  if (a.force_out) { V4 o = {fx, fy, fz, (T)0}; ((V4*)a.force_out)[i] = o; }
  if (a.do_kick) {
    const T dt = sizeof(T) == 8 ? (T)a.dt64 : (T)a.dt;
    V4* vel = (V4*)a.vel;
    v.x = fma_t(dt, fx, v.x); v.y = fma_t(dt, fy, v.y); v.z = fma_t(dt, fz, v.z);
    vel[i] = v;
    if (a.do_drift) {
      V4 p;
      p.x = fma_t(v.x, dt, me.x); p.y = fma_t(v.y, dt, me.y); p.z = fma_t(v.z, dt, me.z);
      p.w = me.w;
      ((V4*)a.pos_next_rows)[i] = p;
    }
  }
}

template <typename T, typename V4, typename Args>
__device__ __forceinline__ void apply_force(const Args& a, int i, V4 me, T fx, T fy, T fz) {
  V4 v = me;
  if (a.do_kick) v = ((const V4*)a.vel)[i];
  apply_force<T, V4>(a, i, me, fx, fy, fz, v);
}

// (row block, segment row) of this workgroup: see block_segment() for the XCD-aware mapping
template <typename Args>   // ForceArgs by value or in the kernel-argument segment
__device__ __forceinline__ void wg_coords(const Args& a, int* rb, int* y) {
  *y = blockIdx.y;
  *rb = blockIdx.x;
  if (a.xcd_map) {   // the host sets it only when gridDim.y % 8 == 0, or 8 % gridDim.y == 0 and gridDim.x % (8 / gridDim.y) == 0
    const unsigned X = gridDim.x, Y = gridDim.y;
    const unsigned linear = blockIdx.y * X + blockIdx.x, slot = linear >> 3, xcd = linear & 7u;
    if ((Y & 7u) == 0) {
      const unsigned m = slot / X;
      *y = (int)(xcd + 8u * m);
      *rb = (int)(slot - m * X);
    } else {           // Y = 1, 2 or 4 segment rows: XCD x takes row y = x mod Y; the 8 / Y XCDs of a row deal its row blocks
      const unsigned per = 8u / Y;
      *y = (int)(xcd % Y);
      *rb = (int)(slot * per + xcd / Y);
    }
  }
}

// 16-B write-through (sc1) stores and L1-bypassing (sc1) loads of one {x,y,z,w} word through a buffer descriptor:
// the hand-off between workgroups below moves its payload with nothing else (MI355X_MICROARCH.md, inter-workgroup
// visibility: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by other CUs' stores).
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct u4x2 { u4 lo, hi; };
template <typename V4>
__device__ __forceinline__ void store_word_sc1(__amdgpu_buffer_rsrc_t rs, int off, V4 v) {
  if constexpr (sizeof(V4) == 16) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, off, 0, 16);
  } else {
    u4x2 t = __builtin_bit_cast(u4x2, v);
    __builtin_amdgcn_raw_buffer_store_b128(t.lo, rs, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(t.hi, rs, off + 16, 0, 16);
  }
}
template <typename V4>
__device__ __forceinline__ V4 load_word_sc1(__amdgpu_buffer_rsrc_t rs, int off) {
  if constexpr (sizeof(V4) == 16) {
    return __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
  } else {
    u4x2 t;
    t.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
    t.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16);
    return __builtin_bit_cast(V4, t);
  }
}

// Finish the R rows of a lane (rows lane_row + r*kBlock, valid below row_end) given the sums of what this wave walked.
//   WS = 4 first (ForceArgs::wsplit): the four waves of the workgroup hold the sums of the four pieces of ONE segment for the
//                  same 64 rows.  Waves 1..3 put theirs into LDS, one barrier, and wave 0 adds them to its own in ascending
//                  source order, ((w0 + w1) + w2) + w3 — the third level of the sum, restated by the oracle (ref_order_t::wsplit).
//                  Only wave 0 goes on; the workgroup then is one 64-row unit with one partial sum per segment.
//   kFinishDirect  apply them.
//   kFinishStore   store them as this segment's partial; combine_kernel adds the segments later.
//   kFinishLast    the split-reduction hand-off inside one launch, wave by wave (no workgroup barrier: with WS = 1 the four waves
//                  of a workgroup own disjoint rows).  Every wave stores its partial sums with write-through (sc1) stores, drains
//                  them (s_waitcnt vmcnt(0)), and ONE lane takes a ticket with an agent-scope atomic add on the counter of its
//                  64 rows.  The wave whose ticket is the last of the nseg (over all launches of the step) learns
//                  that from the returned value, and its lanes read the nseg partials of their rows with sc1 loads IN
//                  ASCENDING SEGMENT ORDER — so the result is the same whichever wave arrives last — apply them and zero
//                  the ticket for the next step.  Nothing ever waits on another wave.
//                  Visibility relied on (MI355X_MICROARCH.md, inter-workgroup visibility; the compiled sequence is LLVM's own
//                  agent-scope release/acquire for gfx942+: `buffer_store ... sc1` -> `s_waitcnt vmcnt(0)` -> `global_atomic_add`
//                  (agent scope: performed at memory, past every L2) -> `buffer_inv sc1` -> `buffer_load ... sc1`):
//                    (1) an sc1 store is written through the issuing XCD's L2 to memory, and vmcnt reaches 0 only when that
//                        write is acknowledged — so a partial sum is in memory before its wave's ticket is;
//                    (2) the ticket is one memory location for all XCDs (device-scope atomics are not cached in a non-coherent
//                        L2), so exactly one wave sees nseg - 1 and every other wave's add — hence its stores — precedes it;
//                    (3) `buffer_inv sc1` drops the reader's L1 and its XCD's non-coherent L2 lines, and an sc1 load misses
//                        both, so the reader fetches what (1) wrote.  The regions of different tickets never share a cache line
//                        (ForceArgs::part_stride, launch-relative rows: 1 KiB-aligned), so no line can be half stale.
//                  None of this depends on how many workgroups share a CU (the guide measured one per CU; here eight):
//                  the guarantees are per store/atomic/load, not per CU.  tests: test_one_launch_combine_equals_combine_kernel.
//
// The arguments are re-read here from the kernel-argument segment (every force kernel takes one ForceArgs by value, at
// offset 0 of it) instead of being kept in SGPRs through the source loop, where the two scalar-load buffers of the
// delivery need the registers.
//
// CF > 0: partial sums in flight per row in the last arriver's read (default by R).  That read is the tail of a launch —
// every other wave has left — and it is latency-bound: nseg / CF rounds of ~0.8 us (per-wave timestamps at N = 16384: all
// source loops over at 62-63 us, the last arrivers done at 70.5 with 8 in flight; profiles/r02_small_n.md).  The kernel's
// VGPR budget decides how many fit, 4 per partial in fp32 (hipcc splits a 12-byte load into dword loads in three passes).
// results(T0) ... results(T0 + N - 1) of the FPGA order added as final_adder's tree adds them (leaves 2J, 2J+1 first; tree16 above is the
// same association on an array): results(t) = partial[(count + t) mod 16], or zero where no item existed; partial 0 is the caller's own,
// partial k > 0 is wave k's word in LDS.  Written as a recursion with the halves kept apart so that at most four words are in flight:
// read all sixteen at once and the 64-VGPR budget of a 16-wave workgroup spills.
template <int T0, int N>
__device__ __forceinline__ f4 fpga_tree(f4 (*ws)[64], int lane, int count, const f4& own) {
  if constexpr (N == 1) {
    const int idx = (count + T0) & 15;                  // wave-uniform
    f4 p = ws[idx > 0 ? idx - 1 : 0][lane];
    if (idx == 0) p = own;
    if (count - 16 + T0 < 0) { p.x = 0.f; p.y = 0.f; p.z = 0.f; }
    return p;
  } else {
    f4 lo = fpga_tree<T0, N / 2>(ws, lane, count, own);
    if constexpr (N >= 8) asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z) : : "memory");
    const f4 hi = fpga_tree<T0 + N / 2, N / 2>(ws, lane, count, own);
    f4 r = {lo.x + hi.x, lo.y + hi.y, lo.z + hi.z, 0.f};
    return r;
  }
}

template <typename T, typename V4, int R, int WS, int CF = 0, int FPGA = 0>
__device__ __forceinline__ void finish_rows(int seg, int lane_row, int row_end, const V4 (&me)[R], Sums<T, R>& s, V4 (*ws)[64], int fpga_count = 0) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  if constexpr (WS > 1 && FPGA) {
    // force_fpga16w_f32: wave k holds partial sum k of the reference's sixteen (S/fxyz.vhd:129-145).  results(t) = partial[(count + t)
    // mod 16], zero where no item existed (S/fxyz.vhd:147-184), then the adder tree (S/final_adder.vhd:88-104): the rotation is the
    // order in which wave 0 reads the other waves' words (count is the same for the whole workgroup)
    static_assert(R == 1 && WS == 16, "sixteen partial sums on sixteen waves");
    if (wave > 0) { V4 o = {s.bx[0], s.by[0], s.bz[0], (T)0}; ws[wave - 1][lane] = o; }
    __syncthreads();
    if (wave > 0) return;
    const f4 own = {s.bx[0], s.by[0], s.bz[0], 0.f};
    const f4 sum = fpga_tree<0, 16>(ws, lane, fpga_count, own);
    s.bx[0] = sum.x; s.by[0] = sum.y; s.bz[0] = sum.z;
  } else if constexpr (WS > 1) {
    static_assert(R == 1, "the wave split is for one body per lane");
    if (wave > 0) { V4 o = {s.bx[0], s.by[0], s.bz[0], (T)0}; ws[wave - 1][lane] = o; }
    __syncthreads();
    if (wave > 0) return;
    // ascending source order: piece 1, 2, 3, ... onto piece 0; three LDS reads in flight at a time (all fifteen of a
    // 16-wave workgroup at once would cost the kernel its 64-VGPR budget: the next chunk's address is made to depend on
    // this chunk's sum, which is the only ordering the scheduler respects here)
    int ln = lane;
#pragma unroll
    for (int k0 = 0; k0 < WS - 1; k0 += 3) {
      V4 p[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) if (k0 + k < WS - 1) p[k] = ws[k0 + k][ln];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k0 + k < WS - 1) { s.bx[0] = s.bx[0] + p[k].x; s.by[0] = s.by[0] + p[k].y; s.bz[0] = s.bz[0] + p[k].z; }
      if constexpr (WS > 4) asm volatile("" : "+v"(ln) : "v"(s.bx[0]), "v"(s.by[0]), "v"(s.bz[0]));
    }
  }
  const NB_CONST ForceArgs* ka = (const NB_CONST ForceArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));   // opaque from here on: the loads below cannot move above the source loop
  const NB_CONST ForceArgs& a = *ka;
  if (a.finish == kFinishDirect) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane_row + r * kBlock;
      if (i < row_end) apply_force<T, V4>(a, i, me[r], s.bx[r], s.by[r], s.bz[r]);
    }
    return;
  }
  if (a.finish == kFinishStore) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane_row + r * kBlock;
      V4 o = {s.bx[r], s.by[r], s.bz[r], (T)0};   // S/compute_store.vhd:242 {0, Fz, Fy, Fx}
      if (i < row_end) ((V4*)a.partial)[(size_t)seg * a.part_stride + (i - a.row0)] = o;
    }
    return;
  }
  constexpr int kWgRows = WS > 1 ? 64 : kBlock * R;
  int row_block, y_unused;
  wg_coords(a, &row_block, &y_unused);   // recomputed from the re-read arguments rather than kept in SGPRs through the source loop
  const int wg_rel0 = row_block * kWgRows;                              // wave-uniform: first row of this workgroup, from row0
  const int wg_rows = min(kWgRows, a.row_count - wg_rel0);
  const int lane_off = (WS > 1 ? lane : (int)threadIdx.x) * (int)sizeof(V4);
  {
    char* base = (char*)a.partial + ((size_t)seg * a.part_stride + wg_rel0) * sizeof(V4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, wg_rows * (int)sizeof(V4), 0x00020000);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      V4 o = {s.bx[r], s.by[r], s.bz[r], (T)0};
      store_word_sc1<V4>(rs, lane_off + r * kBlock * (int)sizeof(V4), o);   // rows past row_end fall outside the descriptor: dropped
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached memory before its ticket is taken
  unsigned* ticket = a.tickets + (size_t)(wg_rel0 >> 6) + (WS > 1 ? 0 : wave);
  unsigned t = 0;
  if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  if (t != (unsigned)(a.nseg - 1)) return;
  // agent-scope acquire (buffer_inv sc1), then the partials with sc1 loads only: see (3) above
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  T fx[R], fy[R], fz[R];
  // CF > 0 (the hand-scheduled fp32 kernels, which have the registers): the rows' velocities travel with the first round of
  // partial sums — one memory latency off the launch's tail
  V4 v0[R];
  if constexpr (CF > 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane_row + r * kBlock;
      v0[r] = me[r];
      if (a.do_kick) v0[r] = ((const V4*)a.vel)[i < row_end ? i : row_end - 1];
    }
  }
  constexpr int C = CF > 0 ? CF : (R >= 4 ? 2 : (R == 2 ? 4 : 8));   // partials in flight per row (loads first, then the adds in order)
  for (int sg0 = 0; sg0 < a.nseg; sg0 += C) {
    V4 p[C][R];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (sg0 + k < a.nseg) {
        char* base = (char*)a.partial + ((size_t)(sg0 + k) * a.part_stride + wg_rel0) * sizeof(V4);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, wg_rows * (int)sizeof(V4), 0x00020000);
        if (sg0 + k == seg) {   // (wave-uniform) this wave's own partial sum is still in its registers: not read back
#pragma unroll
          for (int r = 0; r < R; ++r) { V4 o = {s.bx[r], s.by[r], s.bz[r], (T)0}; p[k][r] = o; }
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) p[k][r] = load_word_sc1<V4>(rs, lane_off + r * kBlock * (int)sizeof(V4));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (sg0 + k < a.nseg) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (sg0 + k == 0) { fx[r] = p[k][r].x; fy[r] = p[k][r].y; fz[r] = p[k][r].z; }
          else { fx[r] = fx[r] + p[k][r].x; fy[r] = fy[r] + p[k][r].y; fz[r] = fz[r] + p[k][r].z; }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane_row + r * kBlock;
    if (i < row_end) {
      if constexpr (CF > 0) apply_force<T, V4>(a, i, me[r], fx[r], fy[r], fz[r], v0[r]);
      else apply_force<T, V4>(a, i, me[r], fx[r], fy[r], fz[r]);
    }
  }
  if (lane == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Which rows and which source segment a workgroup takes.  The hardware deals workgroups round-robin over the 8 XCDs in
// launch order (blocks b and b + 8 share one; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement"; checked here with
// HW_REG_XCC_ID), and each XCD has its own 4 MiB L2.  With the plain (blockIdx.x, blockIdx.y) = (row block, segment)
// mapping all 8 XCDs walk the same segment at the same time and each pulls it over the fabric: N = 1M, 8 segments of
// 2 MiB, 16 resident sets -> 268 MB of source reads per step for 16 MiB of sources.  xcd_map (when the launch has a
// multiple of 8 segment rows): XCD x takes the segment rows y = x (mod 8) for every row block, in launch order
// slot = linear / 8 -> (y = x + 8 * (slot / X), row block = slot % X): its resident workgroups read ONE segment, from its
// own L2, fetched once.  Only who computes what changes; sums, tickets and results are the same.
__device__ __forceinline__ void block_segment(const ForceArgs& a, int* seg, int* jb, int* je, int* rb) {
  int y;
  wg_coords(a, rb, &y);
  int q = a.slice_start - y / a.sub;
  q %= a.nslices; if (q < 0) q += a.nslices;
  int t = y % a.sub;
  *seg = q * a.sub + t;
  segment_bounds(q, t, a.n_src, a.nslices, a.sub, jb, je);
}

// ---------------------------------------------------------------------------
// Shared prologue: the R rows of a lane.
template <typename T, typename V4, int R>
__device__ __forceinline__ void load_rows(const ForceArgs& a, int lane_row, int row_end, V4 (&me)[R]) {
  const V4* rows = (const V4*)a.rows;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane_row + r * kBlock;
    me[r] = rows[i < row_end ? i : row_end - 1];
  }
}

// What a wave works on: its lane's first row and the sources it walks.  WS = 1: the workgroup's 256*R rows, the whole
// segment.  WS = 4: the workgroup's 64 rows (lane l of every wave owns the same row), piece `wave` of the segment.
template <int R, int WS>
__device__ __forceinline__ void wave_work(const ForceArgs& a, int* seg, int* jb, int* je, int* lane_row) {
  int rb;
  block_segment(a, seg, jb, je, &rb);
  if constexpr (WS > 1) {
    static_assert(R == 1, "the wave split is for one body per lane");
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    piece_bounds(*jb, *je, wave, WS, jb, je);
    *lane_row = a.row0 + rb * 64 + (int)(threadIdx.x & 63);
  } else {
    *lane_row = a.row0 + rb * (kBlock * R) + (int)threadIdx.x;
  }
}
// LDS words of the wave split's hand-over (3 waves x 64 rows); one dummy word when the kernel does not split
#define NB_WS_LDS(V4, WS) __shared__ V4 ws_sums[(WS) > 1 ? (WS) - 1 : 1][64]

// ---------------------------------------------------------------------------
// SMEM variant.  The source words are read with scalar loads (8 bodies = two
// s_load_dwordx16 per group), land in SGPRs and feed the VALU as scalar
// operands: no LDS traffic, no barrier, no VALU instruction spent on the
// broadcast.  Groups are double-buffered by hand (load group g+1, compute g).
template <int R, int ARITH, int WS>
__global__ void __launch_bounds__(wg_threads(WS)) force_smem_f32(ForceArgs a) {
  NB_WS_LDS(f4, WS);
  int seg, jb, je, lane_row;
  wave_work<R, WS>(a, &seg, &jb, &je, &lane_row);
  const float eps = soft_f32();
  const int row_end = a.row0 + a.row_count;
  f4 me[R];
  load_rows<float, f4, R>(a, lane_row, row_end, me);
  Sums<float, R> s;
  s.clear();
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  walk_blocks<float, R>(a, jb, je, s, [&](int j, int j1) {
    constexpr int G = 8;
    if (j + G <= j1) {
      f4 cur[G];
#pragma unroll
      for (int k = 0; k < G; ++k) cur[k] = src[j + k];
      for (; j + 2 * G <= j1; j += G) {
        f4 nxt[G];
#pragma unroll
        for (int k = 0; k < G; ++k) nxt[k] = src[j + G + k];
#pragma unroll
        for (int k = 0; k < G; ++k) {
#pragma unroll
          for (int r = 0; r < R; ++r) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
        }
#pragma unroll
        for (int k = 0; k < G; ++k) cur[k] = nxt[k];
      }
#pragma unroll
      for (int k = 0; k < G; ++k) {
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
      }
      j += G;
    }
    for (; j < j1; ++j) {
      f4 p = src[j];
#pragma unroll
      for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
    }
  });
  finish_rows<float, f4, R, WS>(seg, lane_row, row_end, me, s, ws_sums);
}

// ---------------------------------------------------------------------------
// ISA variant (the default for the timed arithmetic): scalar delivery as in force_smem_f32, with the inner loop
// written instruction by instruction (force_loop_gfx950.inc, generated by tools/gen_force_loop.py, which
// explains the hardware facts it is built on).  One body per lane; same arithmetic, same order (blocks of
// sum_block sources folded into the level-2 accumulators inside the loop) and hence the same bits as
// force_smem_f32<1, 0>.  PLACEMENT = 1 is the product loop, 0 the same instructions one 4-byte phase off (kept to
// re-measure the code-placement effect).  LONG = 1: scalar buffers of 8 bodies for launches with few waves per SIMD
// (small N), where a 4-body buffer's 48 instructions no longer cover the ~280 ns of a scalar load.
#include "force_loop_gfx950.inc"
// SGPR budgets: 81-96 SGPRs leave room for 7 waves per SIMD, <= 80 for 8 (MI355X_MICROARCH.md, "Occupancy API" row).  The
// product loop's scalars end at s71 (78-79 with VCC etc.: 8 waves); the long-buffer loop holds 64 buffer SGPRs (106: 6-7 waves),
// which is why it is a kernel of its own.
template <int PLACEMENT, int LONG, int WS>
__device__ __forceinline__ void force_isa_f32_body(const ForceArgs& a, f4 (*ws_sums)[64]) {
  int seg, jb, je, i;
  wave_work<1, WS>(a, &seg, &jb, &je, &i);
  const float eps = soft_f32();
  const int row_end = a.row0 + a.row_count;
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  const int count = je - jb;
  // Touch the scalar-cache lines of the wave's first sources now, so that they travel together with the row load:
  // the loop's own first s_load is issued only after the wait for xi, yi, zi (one memory latency saved per wave,
  // which is what a short segment at small N feels).
  // (one dword each, unconditional: jb + 4 stays inside the arrays' 64-word pad; no branch and no register reuse, so
  //  hipcc has no reason to wait for them before the row load)
  const NB_CONST float* srcw = (const NB_CONST float*)(uintptr_t)a.src;
  const float warm0 = srcw[4 * (size_t)jb], warm1 = srcw[4 * (size_t)jb + 16];
  f4 me[1];
  load_rows<float, f4, 1>(a, i, row_end, me);
  const float xi = me[0].x, yi = me[0].y, zi = me[0].z;
  Sums<float, 1> s;
  s.clear();
  float ax = 0.0f, ay = 0.0f, az = 0.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
  int j = jb;
  constexpr int group = LONG ? NB_FORCE_LOOP_LONG_GROUP : NB_FORCE_LOOP_GROUP;
  const int groups = count / group;
  const bool blocked = a.sum_block > 0;
  // groups per block: sum_block is a multiple of the group (nbody_set_option); sequential = one block never finished
  const int blk = blocked ? a.sum_block / group : 0x7fffffff;
  if (groups > 0) {
    const uint64_t p = (uint64_t)(uintptr_t)a.src + (uint64_t)jb * sizeof(f4);
#define NB_RUN_LOOP(TEXT, CLOBBERS)                                                                                     \
      asm volatile(TEXT                                                                                                  \
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az), [bx] "+v"(bx), [by] "+v"(by), [bz] "+v"(bz)            \
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups), [blk] "s"(blk) \
                   : CLOBBERS)
    if constexpr (LONG) {   // its own kernel: the 64 buffer SGPRs would cost the product loop its 8th resident wave
      NB_RUN_LOOP(NB_FORCE_LOOP_LONG, NB_FORCE_LOOP_LONG_CLOBBERS);
    } else if constexpr (PLACEMENT == 0) {   // the product loop one 4-byte placement phase off (kept to re-measure that effect)
      NB_RUN_LOOP(NB_FORCE_LOOP_V0, NB_FORCE_LOOP_CLOBBERS);
#ifdef NBODY_DIAG_LOOPS
    // `make diag` only (libnbody_hip_diag.so): experiment encodings of the same operations (2, 9..13, 16..20: bit-identical)
    // and TIMING-ONLY forms with WRONG RESULTS (3..8, 14, 15) that price one part of the loop inside the real kernel
    // (tools/gen_force_loop.py, profiles/r02_loop_diagnostics.md).  The product library does not contain them.
    } else if constexpr (PLACEMENT == 2) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V2, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 3) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V3, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 4) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V4, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 5) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V5, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 6) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V6, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 7) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V7, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 8) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V8, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 9) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V9, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 10) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V10, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 11) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V11, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 12) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V12, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 13) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V13, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 14) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V14, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 15) {
      __shared__ f4 diag_tile[64];
      if (threadIdx.x < 64) diag_tile[threadIdx.x] = me[0];
      __syncthreads();
      NB_RUN_LOOP(NB_FORCE_LOOP_V15, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 16) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V16, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 17) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V17, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 18) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V18, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 19) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V19, NB_FORCE_LOOP_DIAG_CLOBBERS);
    } else if constexpr (PLACEMENT == 20) {
      NB_RUN_LOOP(NB_FORCE_LOOP_V20, NB_FORCE_LOOP_DIAG_CLOBBERS);
#endif
    } else {
      static_assert(PLACEMENT == 1 || LONG, "this loop form exists in the diagnostic build only (make diag)");
      NB_RUN_LOOP(NB_FORCE_LOOP_V1, NB_FORCE_LOOP_CLOBBERS);
    }
#undef NB_RUN_LOOP
    j += groups * group;
  }
  asm volatile("" :: "s"(warm0), "s"(warm1));   // the warm-up loads must not be dropped as dead
  // the (< 8 or < 16) sources left over, with the compiled pair function: identical operations.  They belong to the
  // last, unfinished block (sum_block is a multiple of 64, so a block never ends inside them).
  for (; j < je; ++j) {
    f4 q = src[j];
    pair_f32<0>(q.x, q.y, q.z, xi, yi, zi, eps, ax, ay, az);
  }
  s.ax[0] = ax; s.ay[0] = ay; s.az[0] = az; s.bx[0] = bx; s.by[0] = by; s.bz[0] = bz;
  s.close(blocked, blocked && (count % a.sum_block) != 0);
  // partial sums in flight (+ the velocity word): as many as keep the kernel's resident waves — 11 (63 VGPRs, 8 waves per
  // SIMD) for the product kernel, 13 (71 VGPRs; its SGPRs allow 7 waves) for the long-buffer kernel of small launches
  finish_rows<float, f4, 1, WS, LONG ? 13 : 11>(seg, i, row_end, me, s, ws_sums);
}
template <int PLACEMENT, int WS>
__global__ void __launch_bounds__(wg_threads(WS), WS == 16 ? 8 : 1)   // (threads, min waves per SIMD): two 16-wave workgroups per CU need <= 64 VGPRs
force_isa_f32(ForceArgs a) { NB_WS_LDS(f4, WS); force_isa_f32_body<PLACEMENT, 0, WS>(a, ws_sums); }
template <int WS>
__global__ void __launch_bounds__(wg_threads(WS)) force_isa_long_f32(ForceArgs a) { NB_WS_LDS(f4, WS); force_isa_f32_body<1, 1, WS>(a, ws_sums); }

// ---------------------------------------------------------------------------
// LDS variant (the north_star's "source bodies tiled into LDS", tile = 256 by
// default).  Double-buffered: the global loads of tile t+1 are issued before
// the compute on tile t and written to the other buffer after it, one barrier
// per tile.  Every lane reads the same LDS address: a broadcast, conflict-free.
// The tile pipeline runs over the whole segment; block ends are found by counting.
template <int R, int ARITH, int TILE>
__global__ void __launch_bounds__(kBlock) force_lds_f32(ForceArgs a) {
  static_assert(TILE % kBlock == 0, "tile is a multiple of the workgroup");
  constexpr int LPT = TILE / kBlock;   // loads per thread per tile
  __shared__ f4 tile[2][TILE];
  int seg, jb, je, lane_row;
  wave_work<R, 1>(a, &seg, &jb, &je, &lane_row);
  const float eps = soft_f32();
  const f4* src = (const f4*)a.src;
  const int row_end = a.row0 + a.row_count;
  f4 me[R];
  load_rows<float, f4, R>(a, lane_row, row_end, me);
  Sums<float, R> s;
  s.clear();
  const bool blocked = a.sum_block > 0;
  int until_fold = blocked ? a.sum_block : 0x7fffffff;   // sources left in the running block
  const int last = a.n_src - 1;
  f4 stage[LPT];
#pragma unroll
  for (int l = 0; l < LPT; ++l) { int j = jb + l * kBlock + threadIdx.x; stage[l] = src[j < last ? j : last]; }
#pragma unroll
  for (int l = 0; l < LPT; ++l) tile[0][l * kBlock + threadIdx.x] = stage[l];
  int buf = 0;
  for (int j0 = jb; j0 < je; j0 += TILE) {
    __syncthreads();
    const int nxt0 = j0 + TILE;
    if (nxt0 < je) {
#pragma unroll
      for (int l = 0; l < LPT; ++l) { int j = nxt0 + l * kBlock + threadIdx.x; stage[l] = src[j < last ? j : last]; }
    }
    const int cnt = je - j0 < TILE ? je - j0 : TILE;
    if (cnt == TILE && until_fold >= TILE) {
#pragma unroll 8
      for (int k = 0; k < TILE; ++k) {
        f4 p = tile[buf][k];
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
      }
      until_fold -= TILE;
      if (until_fold == 0) { s.fold(); until_fold = a.sum_block; }
    } else {
      int k = 0;
      while (k < cnt) {
        const int m = cnt - k < until_fold ? cnt - k : until_fold;
        for (int e = k + m; k < e; ++k) {
          f4 p = tile[buf][k];
#pragma unroll
          for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
        }
        until_fold -= m;
        if (until_fold == 0) { s.fold(); until_fold = a.sum_block; }
      }
    }
    if (nxt0 < je) {
#pragma unroll
      for (int l = 0; l < LPT; ++l) tile[buf ^ 1][l * kBlock + threadIdx.x] = stage[l];
    }
    buf ^= 1;
  }
  s.close(blocked, blocked && ((je - jb) % a.sum_block) != 0);
  finish_rows<float, f4, R, 1>(seg, lane_row, row_end, me, s, nullptr);
}

// ---------------------------------------------------------------------------
// READLANE variant (the north_star's "one 64-lane wavefront per tile, positions
// broadcast via __shfl"): lane l of a wave holds body j0+l; v_readlane_b32
// moves one body's x, y, z to SGPRs.  3 x 4 cycles of VALU issue per source,
// amortised over the R bodies of the lane.
template <int R, int ARITH>
__global__ void __launch_bounds__(kBlock) force_readlane_f32(ForceArgs a) {
  int seg, jb, je, lane_row;
  wave_work<R, 1>(a, &seg, &jb, &je, &lane_row);
  const float eps = soft_f32();
  const f4* src = (const f4*)a.src;
  const int lane = threadIdx.x & 63;
  const int row_end = a.row0 + a.row_count;
  f4 me[R];
  load_rows<float, f4, R>(a, lane_row, row_end, me);
  Sums<float, R> s;
  s.clear();
  const bool blocked = a.sum_block > 0;
  int until_fold = blocked ? a.sum_block : 0x7fffffff;
  const int last = a.n_src - 1;
  f4 nxt = src[jb + lane < last ? jb + lane : last];
  for (int j0 = jb; j0 < je; j0 += 64) {
    f4 cur = nxt;
    int jn = j0 + 64 + lane;
    if (j0 + 64 < je) nxt = src[jn < last ? jn : last];
    const int cnt = je - j0 < 64 ? je - j0 : 64;
    if (cnt == 64 && until_fold >= 64) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        float xj = lane_bcast(cur.x, k);
        float yj = lane_bcast(cur.y, k);
        float zj = lane_bcast(cur.z, k);
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(xj, yj, zj, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
      }
      until_fold -= 64;
      if (until_fold == 0) { s.fold(); until_fold = a.sum_block; }
    } else {
      for (int k = 0; k < cnt; ++k) {
        float xj = lane_bcast(cur.x, k);
        float yj = lane_bcast(cur.y, k);
        float zj = lane_bcast(cur.z, k);
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(xj, yj, zj, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
        if (--until_fold == 0) { s.fold(); until_fold = a.sum_block; }
      }
    }
  }
  s.close(blocked, blocked && ((je - jb) % a.sum_block) != 0);
  finish_rows<float, f4, R, 1>(seg, lane_row, row_end, me, s, nullptr);
}

// ---------------------------------------------------------------------------
// FPGA summation order inside a segment: 16 partial
// sums per axis, source j (counted from the segment start) into partial j mod
// 16 (S/fxyz.vhd:129-145), latched rotated by (count mod 16) (S/fxyz.vhd:147-184)
// and summed by the pairwise tree (S/final_adder.vhd:88-104).  One body per
// lane; 48 accumulators live in VGPRs.  A study mode, not the timed path.
// p[t] <- p[(t + rot) mod 16] for a wave-uniform rot, as a barrel shifter
template <int B>
__device__ __forceinline__ void rotate16_stage(float (&p)[16], bool on) {
  if (on) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = p[(k + B) & 15];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = t[k];
  }
}
__device__ __forceinline__ void rotate16(float (&p)[16], int rot) {
  rotate16_stage<8>(p, (rot & 8) != 0);
  rotate16_stage<4>(p, (rot & 4) != 0);
  rotate16_stage<2>(p, (rot & 2) != 0);
  rotate16_stage<1>(p, (rot & 1) != 0);
}

template <int ARITH>
__global__ void __launch_bounds__(kBlock) force_fpga16_f32(ForceArgs a) {
  int seg, jb, je, i;
  wave_work<1, 1>(a, &seg, &jb, &je, &i);
  const float eps = soft_f32();
  const int row_end = a.row0 + a.row_count;
  f4 me[1];
  load_rows<float, f4, 1>(a, i, row_end, me);
  float px[16], py[16], pz[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) px[k] = py[k] = pz[k] = 0.0f;
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  int j = jb;
  if constexpr (ARITH & kArithStrict) {
    // Strict arithmetic: pair_f32 ends in a wave-uniform branch (rsqrt_strict_f32), which the compiler neither moves loads across nor
    // keeps the accumulations in front of.  So the sources are loaded by hand, eight ahead (two s_load_dwordx16 in flight while the other
    // eight are evaluated), and every partial sum is pinned where it is formed — left alone, all 48 fmas of an iteration sink behind the
    // last branch and hold sixteen sources' differences and inverse cubes live (134 VGPRs, 3 waves per SIMD: 911 G pairs/s at N = 262144
    // in the RTL-faithful mode; pinned: 71 VGPRs in this loop; profiles/r04_sweep_arith_n262144.txt).  Same operations in the same order.
    auto eight = [&](const f4 (&p)[8], int h) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        pair_f32<ARITH>(p[k].x, p[k].y, p[k].z, me[0].x, me[0].y, me[0].z, eps, px[h + k], py[h + k], pz[h + k]);
        asm volatile("" : "+v"(px[h + k]), "+v"(py[h + k]), "+v"(pz[h + k]));
      }
    };
    if (j + 16 <= je) {
      f4 lo[8], hi[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) lo[k] = src[j + k];
      for (; j + 16 <= je; j += 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) hi[k] = src[j + 8 + k];
        eight(lo, 0);
        if (j + 32 <= je) {
#pragma unroll
          for (int k = 0; k < 8; ++k) lo[k] = src[j + 16 + k];
        }
        eight(hi, 8);
      }
    }
  } else {
    for (; j + 16 <= je; j += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        f4 p = src[j + k];
        pair_f32<ARITH>(p.x, p.y, p.z, me[0].x, me[0].y, me[0].z, eps, px[k], py[k], pz[k]);
      }
    }
  }
  const int tail = je - j;   // < 16
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < tail) {
      f4 p = src[j + k];
      pair_f32<ARITH>(p.x, p.y, p.z, me[0].x, me[0].y, me[0].z, eps, px[k], py[k], pz[k]);
    }
  }
  // results(t) = partial[(count + t) mod 16], zero where no item existed.  count is the same for the whole wave, so the rotation is four
  // wave-uniform stages of register moves (by 8, 4, 2, 1) on one axis at a time: 48 partial sums + 16 temporaries live, where selecting
  // every result from all sixteen partials held 96 registers and set the kernel's occupancy (4 waves per SIMD instead of 7).
  const int count = je - jb;
  const int rot = count & 15;
  rotate16(px, rot); rotate16(py, rot); rotate16(pz, rot);
  if (count < 16) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (count - 16 + t < 0) { px[t] = 0.f; py[t] = 0.f; pz[t] = 0.f; }
    }
  }
  Sums<float, 1> s;
  s.clear();
  s.bx[0] = tree16(px); s.by[0] = tree16(py); s.bz[0] = tree16(pz);
  finish_rows<float, f4, 1, 1>(seg, i, row_end, me, s, nullptr);
}

// ---------------------------------------------------------------------------
// The FPGA order with the sixteen partial sums of a row on sixteen WAVES (round 4; ForceArgs::wsplit = 16).  The reference's partial k
// is the fma chain over the sources k, k + 16, k + 32, ... of the stream (S/fxyz.vhd:120-145: sixteen accumulations in flight): sixteen
// independent chains.  force_fpga16_f32 keeps all sixteen in one lane, so a launch has one wave per 64 rows — at the mailbox's maximum
// N = 32767 that is 512 waves on a chip with 8192 wave slots (2.56 ms per pass).  Here a workgroup of 16 waves owns 64 rows, wave k walks
// the sources congruent to k with scalar delivery like every other kernel (one chain per lane, 8 sources loaded ahead), and wave 0 reads
// the other fifteen sums from LDS in rotated order and adds the tree (finish_rows FPGA).  Same operations in the same order per chain,
// same rotation, same tree: same bits as force_fpga16_f32 (tests: the rtl_n*.json fixtures, test_fpga16_order with NBODY_OPT_WSPLIT 1
// and 16).
template <int ARITH>
__global__ void __launch_bounds__(wg_threads(16), 8) force_fpga16w_f32(ForceArgs a) {
  NB_WS_LDS(f4, 16);
  int seg, jb, je, rb;
  block_segment(a, &seg, &jb, &je, &rb);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane_row = a.row0 + rb * 64 + (int)(threadIdx.x & 63);
  const float eps = soft_f32();
  const int row_end = a.row0 + a.row_count;
  f4 me[1];
  load_rows<float, f4, 1>(a, lane_row, row_end, me);
  Sums<float, 1> s;
  s.clear();
  float px = 0.0f, py = 0.0f, pz = 0.0f;
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  constexpr int G = 8;
  int j = jb + wave;
  if (j + 16 * (G - 1) < je) {
    f4 cur[G];
#pragma unroll
    for (int k = 0; k < G; ++k) cur[k] = src[j + 16 * k];
    for (; j + 16 * (2 * G - 1) < je; j += 16 * G) {
      f4 nxt[G];
#pragma unroll
      for (int k = 0; k < G; ++k) nxt[k] = src[j + 16 * (G + k)];
#pragma unroll
      for (int k = 0; k < G; ++k) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, me[0].x, me[0].y, me[0].z, eps, px, py, pz);
#pragma unroll
      for (int k = 0; k < G; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, me[0].x, me[0].y, me[0].z, eps, px, py, pz);
    j += 16 * G;
  }
  for (; j < je; j += 16) {
    f4 p = src[j];
    pair_f32<ARITH>(p.x, p.y, p.z, me[0].x, me[0].y, me[0].z, eps, px, py, pz);
  }
  s.bx[0] = px; s.by[0] = py; s.bz[0] = pz;
  finish_rows<float, f4, 1, 16, 0, 1>(seg, lane_row, row_end, me, s, ws_sums, je - jb);
}

// The same sixteen chains with the sources STAGED THROUGH LDS (round 6; the default of the FPGA order on sixteen waves).  Scalar delivery
// suits a wave that walks CONSECUTIVE sources (one s_load_dwordx16 brings four); wave k of this kernel wants the sources k, k + 16, ...: one
// 16-byte scalar load each, eight in flight, every one a scalar-cache miss from a launch's cold start.  Here the workgroup's 1024
// threads fetch 1024 consecutive sources with ONE coalesced 16-byte load each into an LDS tile (double-buffered: the next tile's loads are
// in flight while this one is walked, one barrier per tile), and wave k reads its sources k, k + 16, ... of the tile with broadcast
// ds_read_b128 (every lane the same address: conflict-free).  Same chains — wave k still adds the sources congruent to k (1024 is a multiple
// of 16) in ascending order with the same pair_f32 —, same rotation, same tree: the same bits as the scalar form (NBODY_OPT_VARIANT =
// NBODY_VARIANT_SMEM keeps that one selectable; tests: test_fpga16_order, the rtl_n*.json fixtures).  LDS: 2 x 16 KiB + the 15 KiB join.
// Measured (profiles/r06_mailbox_rate.txt, block C's header): LEVEL with the scalar form at every size — N = 1024 11.3 us either way: the
// launch was not load-bound but issue-bound inside its sixteen CUs, which is what force_fpga16r_f32 below is for.  Kept as the default because
// the 16-row kernel shares its tile code and a launch's first loads no longer serialise on the scalar cache.
template <int ARITH>
__global__ void __launch_bounds__(wg_threads(16), 8) force_fpga16w_lds_f32(ForceArgs a) {
  NB_WS_LDS(f4, 16);
  constexpr int TILE = 1024;
  __shared__ f4 tile[2][TILE];
  int seg, jb, je, rb;
  block_segment(a, &seg, &jb, &je, &rb);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int t = (int)threadIdx.x;
  const int lane_row = a.row0 + rb * 64 + (t & 63);
  const float eps = soft_f32();
  const int row_end = a.row0 + a.row_count;
  f4 me[1];
  load_rows<float, f4, 1>(a, lane_row, row_end, me);
  Sums<float, 1> s;
  s.clear();
  float px = 0.0f, py = 0.0f, pz = 0.0f;
  const f4* gsrc = (const f4*)a.src;
  const f4 none = {0.f, 0.f, 0.f, 0.f};
  tile[0][t] = (jb + t < je) ? gsrc[jb + t] : none;
  __syncthreads();
  int buf = 0;
  for (int base = jb; base < je; base += TILE) {
    const int nxt = base + TILE;
    const bool more = nxt < je;                        // workgroup-uniform
    f4 pre = none;
    if (more && nxt + t < je) pre = gsrc[nxt + t];     // in flight while this tile is walked
    const int cnt = je - base < TILE ? je - base : TILE;
    const f4* tl = tile[buf];
    int m = wave;
    for (; m + 16 * 7 < cnt; m += 16 * 8) {
      f4 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = tl[m + 16 * k];
#pragma unroll
      for (int k = 0; k < 8; ++k) pair_f32<ARITH>(p[k].x, p[k].y, p[k].z, me[0].x, me[0].y, me[0].z, eps, px, py, pz);
    }
    for (; m < cnt; m += 16) {
      const f4 p = tl[m];
      pair_f32<ARITH>(p.x, p.y, p.z, me[0].x, me[0].y, me[0].z, eps, px, py, pz);
    }
    if (more) tile[buf ^ 1][t] = pre;
    __syncthreads();
    buf ^= 1;
  }
  s.bx[0] = px; s.by[0] = py; s.bz[0] = pz;
  finish_rows<float, f4, 1, 16, 0, 1>(seg, lane_row, row_end, me, s, ws_sums, je - jb);
}

// The FPGA order for SMALL launches (round 6): sixteen ROWS and sixteen chains per workgroup.  With 64 rows per workgroup a pass over N
// bodies has N / 64 workgroups — N = 1024: sixteen, on sixteen of the 256 CUs, four waves per SIMD; the launch is issue-bound INSIDE those
// CUs (N = 1024: 5.5 us of issue per SIMD, 11.3 us measured) while 240 CUs idle.  Here a workgroup of 256 threads owns 16 rows: lane l of
// wave w holds row l mod 16 and chain 4 w + l / 16 — the reference's twelve bodies x sixteen partial sums (S/top_level.vhd:44,
// S/fxyz.vhd:129-145) with sixteen bodies, one (body, partial sum) per lane.  N / 16 workgroups of four waves: four times the CUs, a quarter
// of the walk per SIMD.  Sources come through the same double-buffered 1024-body LDS tile (four coalesced loads per thread); a wave reads
// four consecutive words per step (its four chains), each broadcast to sixteen lanes.  The sixteen sums of a row meet in LDS
// (part[chain][row]); lane r of wave 0 takes results(t) = part[(count + t) mod 16][r] — zero where no item existed — and adds final_adder's
// tree.  Same chains, rotation, tree: the same bits.  One segment only (the launch finishes its rows itself): what the mailbox's faithful
// mode runs (NBODY_OPT_JSUB 1); several segments keep the 64-row kernels, whose tickets count 64-row units.
// a.src / a.rows may be RAM A itself (pinned host memory): for a handful of bodies the mailbox skips its ingest launch and this kernel's
// coalesced tile loads are the PCIe reads (a few workgroups x N x 16 bytes); its first wave then stamps the tick count's start (t0_stamp).
template <int ARITH>
__global__ void __launch_bounds__(256) force_fpga16r_f32(ForceArgs a) {
  constexpr int TILE = 1024;
  __shared__ f4 tile[2][TILE];
  __shared__ f4 part[16][16];
  const int t = (int)threadIdx.x;
  if (a.t0_stamp && blockIdx.x == 0 && t == 0) *a.t0_stamp = __builtin_amdgcn_s_memrealtime();   // a mailbox request without an ingest launch starts its ticks here
  const int r = t & 15, c = t >> 4;
  const int row_end = a.row0 + a.row_count;
  const int i = a.row0 + (int)blockIdx.x * 16 + r;
  const f4 me = ((const f4*)a.rows)[i < row_end ? i : row_end - 1];
  const float eps = soft_f32();
  int jb, je;
  segment_bounds(0, 0, a.n_src, 1, 1, &jb, &je);        // the one segment: every source, ascending (S/top_level.vhd:233-254)
  const f4* gsrc = (const f4*)a.src;
  const f4 none = {0.f, 0.f, 0.f, 0.f};
  float px = 0.0f, py = 0.0f, pz = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int j = jb + q * 256 + t; tile[0][q * 256 + t] = j < je ? gsrc[j] : none; }
  __syncthreads();
  int buf = 0;
  for (int base = jb; base < je; base += TILE) {
    const int nxt = base + TILE;
    const bool more = nxt < je;                          // workgroup-uniform
    f4 pre[4] = {none, none, none, none};
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = nxt + q * 256 + t; if (j < je) pre[q] = gsrc[j]; }
    }
    const int cnt = je - base < TILE ? je - base : TILE;
    const f4* tl = tile[buf];
    int m = c;
    for (; m + 16 * 7 < cnt; m += 16 * 8) {              // (cnt is uniform, c differs by at most 15: the lanes leave this loop within one round)
      f4 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = tl[m + 16 * k];
      pairs_f32<ARITH, 8>(p, me.x, me.y, me.z, eps, px, py, pz);
    }
    for (; m < cnt; m += 16) {
      const f4 p = tl[m];
      pair_f32<ARITH>(p.x, p.y, p.z, me.x, me.y, me.z, eps, px, py, pz);
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[buf ^ 1][q * 256 + t] = pre[q];
    }
    __syncthreads();
    buf ^= 1;
  }
  const f4 mine = {px, py, pz, 0.f};
  part[c][r] = mine;
  __syncthreads();
  if (t < 16 && i < row_end) {
    const int count = je - jb;
    float rx[16], ry[16], rz[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const f4 v = part[(count + k) & 15][t];
      const bool item = count - 16 + k >= 0;             // results(k) is written as 0.0 when no item reached it (S/fxyz.vhd:177-181)
      rx[k] = item ? v.x : 0.f; ry[k] = item ? v.y : 0.f; rz[k] = item ? v.z : 0.f;
    }
    apply_force<float, f4>(a, i, me, tree16(rx), tree16(ry), tree16(rz));
  }
}

// ---------------------------------------------------------------------------
// fp64 (BASELINE config 5).  SMEM delivery, R bodies per lane.  One sequential sum per segment (fp64 has 29 more
// bits than the 1e-5 target needs; sum_block is ignored).
template <int R, int WS, int STRICT = 0>
__global__ void __launch_bounds__(wg_threads(WS)) force_smem_f64(ForceArgs a) {
  NB_WS_LDS(d4, WS);
  int seg, jb, je, lane_row;
  wave_work<R, WS>(a, &seg, &jb, &je, &lane_row);
  const double eps = (double)soft_f32();
  const int row_end = a.row0 + a.row_count;
  d4 me[R];
  load_rows<double, d4, R>(a, lane_row, row_end, me);
  Sums<double, R> s;
  s.clear();
  const NB_CONST d4* src = (const NB_CONST d4*)(uintptr_t)a.src;
  constexpr int G = 4;
  int j = jb;
  for (; j + G <= je; j += G) {
    d4 cur[G];
#pragma unroll
    for (int k = 0; k < G; ++k) cur[k] = src[j + k];
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
      for (int r = 0; r < R; ++r) pair_f64<STRICT>(cur[k].x, cur[k].y, cur[k].z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
    }
  }
  for (; j < je; ++j) {
    d4 p = src[j];
#pragma unroll
    for (int r = 0; r < R; ++r) pair_f64<STRICT>(p.x, p.y, p.z, me[r].x, me[r].y, me[r].z, eps, s.ax[r], s.ay[r], s.az[r]);
  }
  s.close(false, false);
  finish_rows<double, d4, R, WS>(seg, lane_row, row_end, me, s, ws_sums);
}

// ---------------------------------------------------------------------------
// fp64 with the hand-scheduled loop (force_loop_gfx950.inc, NB_FORCE_LOOP_F64_*): one body per lane, 4 sources
// per iteration, every instruction 8 bytes (v_rsq_f64 in its _e64 encoding).  Same bits as force_smem_f64<1>.
template <int PLACEMENT, int WS>
__global__ void __launch_bounds__(wg_threads(WS)) force_isa_f64(ForceArgs a) {
  NB_WS_LDS(d4, WS);
  int seg, jb, je, i;
  wave_work<1, WS>(a, &seg, &jb, &je, &i);
  const double eps = (double)soft_f32();
  const int row_end = a.row0 + a.row_count;
  d4 me[1];
  load_rows<double, d4, 1>(a, i, row_end, me);
  const double xi = me[0].x, yi = me[0].y, zi = me[0].z;
  double ax = 0.0, ay = 0.0, az = 0.0;
  int j = jb;
  const int groups = (je - jb) / NB_FORCE_LOOP_F64_GROUP;
  if (groups > 0) {
    const uint64_t p = (uint64_t)(uintptr_t)a.src + (uint64_t)jb * sizeof(d4);
    if constexpr (PLACEMENT == 2) {   // experiment: eps and 3/8 from VGPR pairs instead of SGPR pairs
      asm volatile(NB_FORCE_LOOP_F64_V2
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups)
                   : NB_FORCE_LOOP_F64_CLOBBERS);
    } else if constexpr (PLACEMENT == 1) {
      asm volatile(NB_FORCE_LOOP_F64_V1
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups)
                   : NB_FORCE_LOOP_F64_CLOBBERS);
    } else {
      asm volatile(NB_FORCE_LOOP_F64_V0
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups)
                   : NB_FORCE_LOOP_F64_CLOBBERS);
    }
    j += groups * NB_FORCE_LOOP_F64_GROUP;
  }
  const NB_CONST d4* src = (const NB_CONST d4*)(uintptr_t)a.src;
  for (; j < je; ++j) {
    d4 q = src[j];
    pair_f64(q.x, q.y, q.z, xi, yi, zi, eps, ax, ay, az);
  }
  Sums<double, 1> s;
  s.clear();
  s.bx[0] = ax; s.by[0] = ay; s.bz[0] = az;
  finish_rows<double, d4, 1, WS>(seg, i, row_end, me, s, ws_sums);
}

// ---------------------------------------------------------------------------
// combine (the two-launch form of a step, NBODY_OPT_FUSE_COMBINE = 0): F_i = ((p_0 + p_1) + p_2) + ... over the
// segments in ascending source order (so the result does not depend on the order in which slices arrived), then
// what apply_force does.  HBM-bound, nseg x 16 B per body.  Same operations as the last-arriver path of finish_rows.
template <typename T, typename V4>
__global__ void __launch_bounds__(kBlock) combine_kernel(ForceArgs a) {
  const int i = a.row0 + blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.row0 + a.row_count) return;
  const V4* part = (const V4*)a.partial;
  const V4 me = ((const V4*)a.rows)[i];
  // 16 partials in flight (this kernel is latency: at small N it is a handful of workgroups), added in ascending order
  constexpr int C = 16;
  T fx = (T)0, fy = (T)0, fz = (T)0;
  for (int sg0 = 0; sg0 < a.nseg; sg0 += C) {
    V4 p[C];
#pragma unroll
    for (int k = 0; k < C; ++k)
      if (sg0 + k < a.nseg) p[k] = part[(size_t)(sg0 + k) * a.part_stride + (i - a.row0)];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (sg0 + k < a.nseg) {
        if (sg0 + k == 0) { fx = p[k].x; fy = p[k].y; fz = p[k].z; }
        else { fx = fx + p[k].x; fy = fy + p[k].y; fz = fz + p[k].z; }
      }
    }
  }
  apply_force<T, V4>(a, i, me, fx, fy, fz);
}

// The mailbox's RAM A read port (S/top_level.vhd:206-208, 238-240): body words 1..N of the host's RAM image — pinned host memory
// the device reads over PCIe — into the resident source array, one 16-byte word per lane.  A kernel rather than a copy command:
// it sits on the same queue as the force launch that follows, so a request is launches only, with no hand-over between engines.
// Its first wave also starts the request's tick counter (S/top_level.vhd:138-139: clk_ctr leaves 0 on BEGIN's rising edge): *t0 = the
// constant-rate real-time counter (s_memrealtime, 100 MHz), read back by mailbox_done_kernel.
__global__ void __launch_bounds__(kBlock) ingest_kernel(f4* dst, const f4* ram_a_bodies, int n, unsigned long long* t0) {
  if (t0 && blockIdx.x == 0 && threadIdx.x == 0) *t0 = __builtin_amdgcn_s_memrealtime();
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] = ram_a_bodies[i];
}

// `complete` (S/top_level.vhd:255-263) done by the device, as the PL block does it: the last launch of a request — one wave on the
// same queue, behind the force pass that stored RAM B — rewrites word 0 of RAM A (pinned host memory) with {ticks in bits 63:32, 0
// elsewhere}.  ticks = 1 + elapsed 1000-clock units of a `clock_khz` clock between the ingest kernel's first wave and this store
// (:121-146), from the real-time counter (rt_khz, 100 MHz on gfx950).  BEGIN (bit 0) is cleared LAST and with system-scope release: whoever
// reads BEGIN = 0 also reads the ticks and every word of RAM B.  After it, `seq` goes into a second pinned word that the PS never writes: the
// library's own completion signal (the driver may raise BEGIN again the moment it has seen it fall).
__global__ void __launch_bounds__(64) mailbox_done_kernel(unsigned* word0, unsigned* seq_word, const unsigned long long* t0, unsigned seq,
                                                          unsigned clock_khz, unsigned rt_khz) {
  if (threadIdx.x != 0) return;
  const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - *t0;          // in 1 / rt_khz ms
  const unsigned long long ticks = 1ull + dt * clock_khz / ((unsigned long long)rt_khz * 1000ull);
  word0[1] = (unsigned)(ticks > 0xFFFFFFFFull ? 0xFFFFFFFFull : ticks); word0[2] = 0u; word0[3] = 0u;
  __hip_atomic_store(&word0[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// integrate(): r += v * dt for the rank's bodies, in place.
template <typename T, typename V4>
__global__ void __launch_bounds__(kBlock) drift_kernel(V4* pos_rows, const V4* vel, int n_rows, float dt32, double dt64) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_rows) return;
  const T dt = sizeof(T) == 8 ? (T)dt64 : (T)dt32;
  V4 p = pos_rows[i];
  V4 v = vel[i];
  p.x = fma_t(v.x, dt, p.x); p.y = fma_t(v.y, dt, p.y); p.z = fma_t(v.z, dt, p.z);
  pos_rows[i] = p;
}

}  // namespace nbk
