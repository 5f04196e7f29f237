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
// transcendental (8 cycles): 30 cycles per 64 pairs per SIMD, measured
// (profiles/r01_microbench_valu_issue.txt).  That issue count, not HBM and not
// MFMA, bounds the kernel.  v_pk_*_f32 cost 4 cycles on gfx950 (same file), so
// packed math buys nothing and is kept out (-fno-slp-vectorize).
//
// The three variants differ only in how r_j reaches the lanes:
//   SMEM      wave-uniform scalar loads into SGPRs; VALU reads them as scalar operands (free)
//   LDS       TILE bodies staged in LDS, every lane reads the same address (broadcast ds_read_b128)
//   READLANE  each lane holds one body of a 64-body wave tile; v_readlane_b32 x3 per source (4 cycles each)
// All of them add the sources of a segment in ascending j into one accumulator
// per axis, so they return identical bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;            // 4 waves, one per SIMD
constexpr uint32_t kSoftBits = 0x3089705Fu;  // S/dzsoft.vhd:177

// Address space 4 = constant: a load through it with a wave-uniform address is
// always selected as s_load_* (scalar cache), never as a vector load.
#define NB_CONST __attribute__((address_space(4)))

struct ForceArgs {
  const void* src;      // all N source bodies (16-B or 32-B words), ascending
  const void* rows;     // the rank's own bodies: rows[i] = src[first_body + i]
  void* partial;        // [nseg][n_rows] words {Fx,Fy,Fz,0}
  void* vel;            // [n_rows] (fused epilogue only)
  void* pos_next_rows;  // [n_rows] (fused epilogue only)
  int n_src;            // N
  int n_rows;           // bodies owned by this rank
  int row0;             // first row handled by this launch (nbody_forces_rows)
  int row_count;        // rows handled by this launch
  int nslices, sub;     // segmentation of the sources: nslices slices (one per rank), `sub` pieces each
  int slice_start;      // blockIdx.y / sub = 0 maps to this slice; then descending modulo nslices (ring arrival order)
  int fused;            // 1: nseg == 1, apply kick and drift here
  int fpga16;           // 1: S/fxyz.vhd:129-184 + S/final_adder.vhd:88-104 summation order inside a segment
  float dt;
  double dt64;
};

// slice q of P over n: [first(q), first(q+1)), balanced
__device__ __host__ inline int slice_first(int q, int n, int P) {
  int base = n / P, rem = n % P;
  return q * base + (q < rem ? q : rem);
}
// segment (q, t): piece t of `sub` of slice q
__device__ __host__ inline void segment_bounds(int q, int t, int n, int P, int sub, int* jb, int* je) {
  int f0 = slice_first(q, n, P), f1 = slice_first(q + 1, n, P);
  int len = f1 - f0;
  int piece = (len + sub - 1) / sub;
  int b = f0 + t * piece;
  int e = b + piece;
  if (b > f1) b = f1;
  if (e > f1) e = f1;
  *jb = b; *je = e;
}

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
  if constexpr (ARITH & kArithStrict) inv = (float)(1.0 / __builtin_sqrt((double)d2));
  else inv = __builtin_amdgcn_rsqf(d2);            // v_rsq_f32, 1 ulp; d2 >= eps is never subnormal
  float inv2 = inv * inv;                          // S/cube.vhd:66-67
  float inv3 = inv * inv2;                         // S/cube.vhd:69-70
  ax = __builtin_fmaf(dx, inv3, ax);               // S/fxyz.vhd:120-127
  ay = __builtin_fmaf(dy, inv3, ay);
  az = __builtin_fmaf(dz, inv3, az);
}

// fp64: v_rsq_f64 seed (about 2^-27 relative) + two Newton steps y <- y + y*(1/2 - (x/2)*y*y), 7 operations;
// same expression tree as oracle/nbody_ref.c ref_forces_f64 apart from how 1/sqrt is obtained, and the same
// operations in the same order as the hand-scheduled fp64 loop (tools/gen_force_loop.py body_f64).
__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double hx = x * 0.5;
  double r = hx * y;
  double e = __builtin_fma(-r, y, 0.5);
  y = __builtin_fma(y, e, y);
  r = hx * y;
  e = __builtin_fma(-r, y, 0.5);
  y = __builtin_fma(y, e, y);
  return y;
}
__device__ __forceinline__ void pair_f64(double xj, double yj, double zj, double xi, double yi, double zi, double eps,
                                         double& ax, double& ay, double& az) {
  double dx = xj - xi, dy = yj - yi, dz = zj - zi;
  double d2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, __builtin_fma(dz, dz, eps)));
  double inv = rsqrt_f64(d2);
  double inv2 = inv * inv;
  double inv3 = inv * inv2;
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
// epilogue shared by the fp32 kernels: either store the segment's partial sum or
// (single segment) apply the kick v += dt*F and the drift r += v*dt in place.
__device__ __forceinline__ void epilogue_f32(const ForceArgs& a, int seg, int i, float xi, float yi, float zi, float wi,
                                             float ax, float ay, float az) {
  if (a.fused) {
    f4* vel = (f4*)a.vel;
    f4* pn = (f4*)a.pos_next_rows;
    f4 v = vel[i];
    v.x = __builtin_fmaf(a.dt, ax, v.x);
    v.y = __builtin_fmaf(a.dt, ay, v.y);
    v.z = __builtin_fmaf(a.dt, az, v.z);
    vel[i] = v;
    f4 p;
    p.x = __builtin_fmaf(v.x, a.dt, xi);
    p.y = __builtin_fmaf(v.y, a.dt, yi);
    p.z = __builtin_fmaf(v.z, a.dt, zi);
    p.w = wi;
    pn[i] = p;
  } else {
    f4 o = {ax, ay, az, 0.0f};   // S/compute_store.vhd:242 {0, Fz, Fy, Fx}
    ((f4*)a.partial)[(size_t)seg * a.n_rows + i] = o;
  }
}

__device__ __forceinline__ void block_segment(const ForceArgs& a, int* seg, int* jb, int* je) {
  int y = blockIdx.y;
  int q = a.slice_start - y / a.sub;
  q %= a.nslices; if (q < 0) q += a.nslices;
  int t = y % a.sub;
  *seg = q * a.sub + t;
  segment_bounds(q, t, a.n_src, a.nslices, a.sub, jb, je);
}

// ---------------------------------------------------------------------------
// SMEM variant.  The source words are read with scalar loads (8 bodies = two
// s_load_dwordx16 per group), land in SGPRs and feed the VALU as scalar
// operands: no LDS traffic, no barrier, no VALU instruction spent on the
// broadcast.  Groups are double-buffered by hand (load group g+1, compute g).
template <int R, int ARITH>
__global__ void __launch_bounds__(kBlock) force_smem_f32(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const float eps = soft_f32();
  const f4* rows = (const f4*)a.rows;
  const int lane_row = a.row0 + blockIdx.x * (kBlock * R) + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  float xi[R], yi[R], zi[R], wi[R], ax[R], ay[R], az[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    f4 p = rows[i < row_end ? i : row_end - 1];
    xi[r] = p.x; yi[r] = p.y; zi[r] = p.z; wi[r] = p.w;
    ax[r] = ay[r] = az[r] = 0.0f;
  }
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  constexpr int G = 8;
  int j = jb;
  if (j + G <= je) {
    f4 cur[G];
#pragma unroll
    for (int k = 0; k < G; ++k) cur[k] = src[j + k];
    for (; j + 2 * G <= je; j += G) {
      f4 nxt[G];
#pragma unroll
      for (int k = 0; k < G; ++k) nxt[k] = src[j + G + k];
#pragma unroll
      for (int k = 0; k < G; ++k) {
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
      }
#pragma unroll
      for (int k = 0; k < G; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
      for (int r = 0; r < R; ++r) pair_f32<ARITH>(cur[k].x, cur[k].y, cur[k].z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
    }
    j += G;
  }
  for (; j < je; ++j) {
    f4 p = src[j];
#pragma unroll
    for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    if (i < row_end) epilogue_f32(a, seg, i, xi[r], yi[r], zi[r], wi[r], ax[r], ay[r], az[r]);
  }
}

// ---------------------------------------------------------------------------
// ISA variant (the default for the timed arithmetic): scalar delivery as in force_smem_f32, with the inner loop
// written instruction by instruction (force_loop_gfx950.inc, generated by tools/gen_force_loop.py, which
// explains the hardware facts it is built on).  One body per lane; same arithmetic, same order and hence the
// same bits as force_smem_f32<1, 0>.  PLACEMENT = 1 is the product loop, 0 the same instructions one 4-byte
// phase off (kept to re-measure the code-placement effect).
#include "force_loop_gfx950.inc"
template <int PLACEMENT>
__global__ void __launch_bounds__(kBlock) force_isa_f32(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const float eps = soft_f32();
  const f4* rows = (const f4*)a.rows;
  const int i = a.row0 + blockIdx.x * kBlock + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  const f4 me = rows[i < row_end ? i : row_end - 1];
  const float xi = me.x, yi = me.y, zi = me.z;
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  int j = jb;
  const int groups = (je - jb) / NB_FORCE_LOOP_GROUP;
  if (groups > 0) {
    const uint64_t p = (uint64_t)(uintptr_t)a.src + (uint64_t)jb * sizeof(f4);
    if constexpr (PLACEMENT == 1) {
      asm volatile(NB_FORCE_LOOP_V1
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups)
                   : NB_FORCE_LOOP_CLOBBERS);
    } else {
      asm volatile(NB_FORCE_LOOP_V0
                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
                   : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [eps] "s"(eps), [p] "s"(p), [groups] "s"(groups)
                   : NB_FORCE_LOOP_CLOBBERS);
    }
    j += groups * NB_FORCE_LOOP_GROUP;
  }
  // the (< 8) sources left over, with the compiled pair function: identical operations
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  for (; j < je; ++j) {
    f4 q = src[j];
    pair_f32<0>(q.x, q.y, q.z, xi, yi, zi, eps, ax, ay, az);
  }
  if (i < row_end) epilogue_f32(a, seg, i, xi, yi, zi, me.w, ax, ay, az);
}

// ---------------------------------------------------------------------------
// LDS variant (the north_star's "source bodies tiled into LDS", tile = 256 by
// default).  Double-buffered: the global loads of tile t+1 are issued before
// the compute on tile t and written to the other buffer after it, one barrier
// per tile.  Every lane reads the same LDS address: a broadcast, conflict-free.
template <int R, int ARITH, int TILE>
__global__ void __launch_bounds__(kBlock) force_lds_f32(ForceArgs a) {
  static_assert(TILE % kBlock == 0, "tile is a multiple of the workgroup");
  constexpr int LPT = TILE / kBlock;   // loads per thread per tile
  __shared__ f4 tile[2][TILE];
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const float eps = soft_f32();
  const f4* rows = (const f4*)a.rows;
  const f4* src = (const f4*)a.src;
  const int lane_row = a.row0 + blockIdx.x * (kBlock * R) + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  float xi[R], yi[R], zi[R], wi[R], ax[R], ay[R], az[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    f4 p = rows[i < row_end ? i : row_end - 1];
    xi[r] = p.x; yi[r] = p.y; zi[r] = p.z; wi[r] = p.w;
    ax[r] = ay[r] = az[r] = 0.0f;
  }
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
    if (cnt == TILE) {
#pragma unroll 8
      for (int k = 0; k < TILE; ++k) {
        f4 p = tile[buf][k];
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
      }
    } else {
      for (int k = 0; k < cnt; ++k) {
        f4 p = tile[buf][k];
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(p.x, p.y, p.z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
      }
    }
    if (nxt0 < je) {
#pragma unroll
      for (int l = 0; l < LPT; ++l) tile[buf ^ 1][l * kBlock + threadIdx.x] = stage[l];
    }
    buf ^= 1;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    if (i < row_end) epilogue_f32(a, seg, i, xi[r], yi[r], zi[r], wi[r], ax[r], ay[r], az[r]);
  }
}

// ---------------------------------------------------------------------------
// READLANE variant (the north_star's "one 64-lane wavefront per tile, positions
// broadcast via __shfl"): lane l of a wave holds body j0+l; v_readlane_b32
// moves one body's x, y, z to SGPRs.  3 x 4 cycles of VALU issue per source,
// amortised over the R bodies of the lane.
template <int R, int ARITH>
__global__ void __launch_bounds__(kBlock) force_readlane_f32(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const float eps = soft_f32();
  const f4* rows = (const f4*)a.rows;
  const f4* src = (const f4*)a.src;
  const int lane = threadIdx.x & 63;
  const int lane_row = a.row0 + blockIdx.x * (kBlock * R) + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  float xi[R], yi[R], zi[R], wi[R], ax[R], ay[R], az[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    f4 p = rows[i < row_end ? i : row_end - 1];
    xi[r] = p.x; yi[r] = p.y; zi[r] = p.z; wi[r] = p.w;
    ax[r] = ay[r] = az[r] = 0.0f;
  }
  const int last = a.n_src - 1;
  f4 nxt = src[jb + lane < last ? jb + lane : last];
  for (int j0 = jb; j0 < je; j0 += 64) {
    f4 cur = nxt;
    int jn = j0 + 64 + lane;
    if (j0 + 64 < je) nxt = src[jn < last ? jn : last];
    const int cnt = je - j0 < 64 ? je - j0 : 64;
    if (cnt == 64) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        float xj = lane_bcast(cur.x, k);
        float yj = lane_bcast(cur.y, k);
        float zj = lane_bcast(cur.z, k);
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(xj, yj, zj, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
      }
    } else {
      for (int k = 0; k < cnt; ++k) {
        float xj = lane_bcast(cur.x, k);
        float yj = lane_bcast(cur.y, k);
        float zj = lane_bcast(cur.z, k);
#pragma unroll
        for (int r = 0; r < R; ++r) pair_f32<ARITH>(xj, yj, zj, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    if (i < row_end) epilogue_f32(a, seg, i, xi[r], yi[r], zi[r], wi[r], ax[r], ay[r], az[r]);
  }
}

// ---------------------------------------------------------------------------
// FPGA summation order inside a segment: 16 partial
// sums per axis, source j (counted from the segment start) into partial j mod
// 16 (S/fxyz.vhd:129-145), latched rotated by (count mod 16) (S/fxyz.vhd:147-184)
// and summed by the pairwise tree (S/final_adder.vhd:88-104).  One body per
// lane; 48 accumulators live in VGPRs.  A study mode, not the timed path.
template <int ARITH>
__global__ void __launch_bounds__(kBlock) force_fpga16_f32(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const float eps = soft_f32();
  const f4* rows = (const f4*)a.rows;
  const int i = a.row0 + blockIdx.x * kBlock + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  f4 me = rows[i < row_end ? i : row_end - 1];
  float px[16], py[16], pz[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) px[k] = py[k] = pz[k] = 0.0f;
  const NB_CONST f4* src = (const NB_CONST f4*)(uintptr_t)a.src;
  int j = jb;
  for (; j + 16 <= je; j += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      f4 p = src[j + k];
      pair_f32<ARITH>(p.x, p.y, p.z, me.x, me.y, me.z, eps, px[k], py[k], pz[k]);
    }
  }
  const int tail = je - j;   // < 16
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < tail) {
      f4 p = src[j + k];
      pair_f32<ARITH>(p.x, p.y, p.z, me.x, me.y, me.z, eps, px[k], py[k], pz[k]);
    }
  }
  // results(t) = partial[(count + t) mod 16], zero where no item existed
  const int count = je - jb;
  const int rot = count & 15;
  float rx[16], ry[16], rz[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (((rot + t) & 15) == k) { vx = px[k]; vy = py[k]; vz = pz[k]; }
    }
    if (count - 16 + t < 0) { vx = vy = vz = 0.f; }
    rx[t] = vx; ry[t] = vy; rz[t] = vz;
  }
  float fx = tree16(rx), fy = tree16(ry), fz = tree16(rz);
  if (i < row_end) epilogue_f32(a, seg, i, me.x, me.y, me.z, me.w, fx, fy, fz);
}

// ---------------------------------------------------------------------------
// fp64 (BASELINE config 5).  SMEM delivery, R bodies per lane.
template <int R>
__global__ void __launch_bounds__(kBlock) force_smem_f64(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const double eps = (double)soft_f32();
  const d4* rows = (const d4*)a.rows;
  const int lane_row = a.row0 + blockIdx.x * (kBlock * R) + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  double xi[R], yi[R], zi[R], wi[R], ax[R], ay[R], az[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    d4 p = rows[i < row_end ? i : row_end - 1];
    xi[r] = p.x; yi[r] = p.y; zi[r] = p.z; wi[r] = p.w;
    ax[r] = ay[r] = az[r] = 0.0;
  }
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
      for (int r = 0; r < R; ++r) pair_f64(cur[k].x, cur[k].y, cur[k].z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
    }
  }
  for (; j < je; ++j) {
    d4 p = src[j];
#pragma unroll
    for (int r = 0; r < R; ++r) pair_f64(p.x, p.y, p.z, xi[r], yi[r], zi[r], eps, ax[r], ay[r], az[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int i = lane_row + r * kBlock;
    if (i >= row_end) continue;
    if (a.fused) {
      d4* vel = (d4*)a.vel;
      d4* pn = (d4*)a.pos_next_rows;
      d4 v = vel[i];
      v.x = __builtin_fma(a.dt64, ax[r], v.x);
      v.y = __builtin_fma(a.dt64, ay[r], v.y);
      v.z = __builtin_fma(a.dt64, az[r], v.z);
      vel[i] = v;
      d4 p;
      p.x = __builtin_fma(v.x, a.dt64, xi[r]);
      p.y = __builtin_fma(v.y, a.dt64, yi[r]);
      p.z = __builtin_fma(v.z, a.dt64, zi[r]);
      p.w = wi[r];
      pn[i] = p;
    } else {
      d4 o = {ax[r], ay[r], az[r], 0.0};
      ((d4*)a.partial)[(size_t)seg * a.n_rows + i] = o;
    }
  }
}

// ---------------------------------------------------------------------------
// fp64 with the hand-scheduled loop (force_loop_gfx950.inc, NB_FORCE_LOOP_F64_*): one body per lane, 4 sources
// per iteration, every instruction 8 bytes (v_rsq_f64 in its _e64 encoding).  Same bits as force_smem_f64<1>.
template <int PLACEMENT>
__global__ void __launch_bounds__(kBlock) force_isa_f64(ForceArgs a) {
  int seg, jb, je;
  block_segment(a, &seg, &jb, &je);
  const double eps = (double)soft_f32();
  const d4* rows = (const d4*)a.rows;
  const int i = a.row0 + blockIdx.x * kBlock + threadIdx.x;
  const int row_end = a.row0 + a.row_count;
  const d4 me = rows[i < row_end ? i : row_end - 1];
  const double xi = me.x, yi = me.y, zi = me.z;
  double ax = 0.0, ay = 0.0, az = 0.0;
  int j = jb;
  const int groups = (je - jb) / NB_FORCE_LOOP_F64_GROUP;
  if (groups > 0) {
    const uint64_t p = (uint64_t)(uintptr_t)a.src + (uint64_t)jb * sizeof(d4);
    if constexpr (PLACEMENT == 1) {
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
  if (i < row_end) {
    if (a.fused) {
      d4* vel = (d4*)a.vel;
      d4* pn = (d4*)a.pos_next_rows;
      d4 v = vel[i];
      v.x = __builtin_fma(a.dt64, ax, v.x);
      v.y = __builtin_fma(a.dt64, ay, v.y);
      v.z = __builtin_fma(a.dt64, az, v.z);
      vel[i] = v;
      d4 q;
      q.x = __builtin_fma(v.x, a.dt64, xi);
      q.y = __builtin_fma(v.y, a.dt64, yi);
      q.z = __builtin_fma(v.z, a.dt64, zi);
      q.w = me.w;
      pn[i] = q;
    } else {
      d4 o = {ax, ay, az, 0.0};
      ((d4*)a.partial)[(size_t)seg * a.n_rows + i] = o;
    }
  }
}

// ---------------------------------------------------------------------------
// combine: F_i = ((p_0 + p_1) + p_2) + ... over the segments in ascending source
// order (so the result does not depend on the order in which slices arrived),
// then kick and/or drift.  HBM-bound, nseg x 16 B per body.
struct CombineArgs {
  const void* partial;   // [nseg][n_rows]
  const void* pos_rows;  // current positions of the rank's bodies
  void* pos_next_rows;   // may alias nothing in pos_rows' buffer
  void* vel;
  void* force_out;       // nullable
  int nseg, n_rows, row0, row_count;
  int do_kick, do_drift;
  float dt;
  double dt64;
};

template <typename T, typename V4>
__global__ void __launch_bounds__(kBlock) combine_kernel(CombineArgs a) {
  int i = a.row0 + blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.row0 + a.row_count) return;
  const V4* part = (const V4*)a.partial;
  V4 f = part[i];
  for (int s = 1; s < a.nseg; ++s) {
    V4 p = part[(size_t)s * a.n_rows + i];
    f.x = f.x + p.x; f.y = f.y + p.y; f.z = f.z + p.z;
  }
  f.w = (T)0;
  if (a.force_out) ((V4*)a.force_out)[i] = f;
  const T dt = sizeof(T) == 8 ? (T)a.dt64 : (T)a.dt;
  if (a.do_kick) {
    V4* vel = (V4*)a.vel;
    V4 v = vel[i];
    if constexpr (sizeof(T) == 8) {
      v.x = __builtin_fma(dt, f.x, v.x); v.y = __builtin_fma(dt, f.y, v.y); v.z = __builtin_fma(dt, f.z, v.z);
    } else {
      v.x = __builtin_fmaf(dt, f.x, v.x); v.y = __builtin_fmaf(dt, f.y, v.y); v.z = __builtin_fmaf(dt, f.z, v.z);
    }
    vel[i] = v;
    if (a.do_drift) {
      V4 p = ((const V4*)a.pos_rows)[i];
      if constexpr (sizeof(T) == 8) {
        p.x = __builtin_fma(v.x, dt, p.x); p.y = __builtin_fma(v.y, dt, p.y); p.z = __builtin_fma(v.z, dt, p.z);
      } else {
        p.x = __builtin_fmaf(v.x, dt, p.x); p.y = __builtin_fmaf(v.y, dt, p.y); p.z = __builtin_fmaf(v.z, dt, p.z);
      }
      ((V4*)a.pos_next_rows)[i] = p;
    }
  }
}

// integrate(): r += v * dt for the rank's bodies, in place.
template <typename T, typename V4>
__global__ void __launch_bounds__(kBlock) drift_kernel(V4* pos_rows, const V4* vel, int n_rows, float dt32, double dt64) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_rows) return;
  const T dt = sizeof(T) == 8 ? (T)dt64 : (T)dt32;
  V4 p = pos_rows[i];
  V4 v = vel[i];
  if constexpr (sizeof(T) == 8) {
    p.x = __builtin_fma(v.x, dt, p.x); p.y = __builtin_fma(v.y, dt, p.y); p.z = __builtin_fma(v.z, dt, p.z);
  } else {
    p.x = __builtin_fmaf(v.x, dt, p.x); p.y = __builtin_fmaf(v.y, dt, p.y); p.z = __builtin_fmaf(v.z, dt, p.z);
  }
  pos_rows[i] = p;
}

}  // namespace nbk
