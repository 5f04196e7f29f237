// Probe: can the matrix pipe carry the three coordinate differences of the pair interaction?
//
// Measurement tool, not product code.  The force loop is bound by VALU issue (11 full-rate + 1 quarter-rate
// instruction per pair); the matrix cores idle.  v_mfma_f32_32x32x2_f32 computes, with K = 2,
//     D[m][n] = A[m][0]*B[0][n] + A[m][1]*B[1][n]
// and with A[m] = {x_j[m], 1}, B[.][n] = {1, -x_i[n]} that is x_j[m]*1 + 1*(-x_i[n]): both products exact, one
// rounding of their sum — the IEEE subtraction S/dxy.vhd:94-98 asks for, 1024 of them per instruction, with no VALU
// issue spent.  This program checks on the hardware
//   (1) the register layout of A, B and D,
//   (2) that the result is bit-identical to v_sub_f32 (random operands, wide exponent range, denormal results,
//       equal operands, infinities),
//   (3) what a force loop built on it sustains: 3 MFMA + 16 x (3 fma, rsq, 2 mul, 3 fma) per 1024 pairs, against a plain
//       VALU kernel that adds the same sources in the same order (bitwise comparison of the forces).
// A wave owns 32 rows; lanes 0-31 and 32-63 receive rows m = 8q + c and 8q + 4 + c of D, so the sources are dealt
// so that lane half h walks piece h of the wave's sources in ascending order — the order of the engine's wave split.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off microbench_mfma.hip -o microbench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// (1) + (2): one MFMA per wave, D written out register by register
__global__ void __launch_bounds__(64) probe_layout(const float* a_in, const float* b_in, float* d_out) {
  const int lane = threadIdx.x;
  const float a = a_in[blockIdx.x * 64 + lane], b = b_in[blockIdx.x * 64 + lane];
  f16v c = {0};
  f16v d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d_out[((size_t)blockIdx.x * 16 + r) * 64 + lane] = d[r];
}

// ---------------------------------------------------------------------------------------------------------------
// (3) the force tile loop
__device__ __forceinline__ void pair(float dx, float dy, float dz, float& ax, float& ay, float& az) {
  const float eps = __builtin_bit_cast(float, 0x3089705Fu);
  float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, __builtin_fmaf(dz, dz, eps)));
  float inv = __builtin_amdgcn_rsqf(d2);
  float inv2 = inv * inv;
  float inv3 = inv * inv2;
  ax = __builtin_fmaf(dx, inv3, ax);
  ay = __builtin_fmaf(dy, inv3, ay);
  az = __builtin_fmaf(dz, inv3, az);
}

// every lane loads (lanes 32-63 the word of lane - 32: no divergent branch around the prefetch), the K = 1 half is then set to one
__device__ __forceinline__ f4 sel(int h, f4 v) { f4 o = {h ? 1.0f : v.x, h ? 1.0f : v.y, h ? 1.0f : v.z, 0.f}; return o; }

// DB = 1: the differences of tile t+1 are produced while the VALU works on tile t (two sets of 48 registers)
template <int DB, int WPS>
__global__ void __launch_bounds__(256, WPS) force_mfma(const f4* __restrict__ src, int n_src, const f4* __restrict__ rows, f4* __restrict__ out, int n_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n32 = lane & 31, h = lane >> 5;
  const int row = (blockIdx.x * 4 + wave) * 32 + n32;
  const f4 me = rows[row < n_rows ? row : n_rows - 1];
  // B[0][n] = 1 (lanes 0-31), B[1][n] = -r_i[n] (lanes 32-63)
  const float bx = h ? -me.x : 1.0f, by = h ? -me.y : 1.0f, bz = h ? -me.z : 1.0f;
  // A[m][0] = r_j[sigma(m)] (lanes 0-31, m = lane), A[m][1] = 1 (lanes 32-63).  D's row m = 8q + 4h' + c lands in lane half h',
  // register 4q + c: source sigma(m) = piece h', tile offset 4q + c, so a lane half walks its piece in ascending order.
  const int half_len = n_src / 2;                       // (the probe takes n_src a multiple of 32)
  const int m = n32, q = m >> 3, hp = (m >> 2) & 1, c = m & 3;
  const f4* pa = src + (size_t)hp * half_len + 4 * q + c;
  const int tiles = half_len / 16;
  float ax = 0.f, ay = 0.f, az = 0.f;
  const f16v zero = {0};
  f4 a_cur = sel(h, pa[0]);
  if constexpr (DB) {
    f16v dx = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, bx, zero, 0, 0, 0);
    f16v dy = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, by, zero, 0, 0, 0);
    f16v dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, bz, zero, 0, 0, 0);
    f4 a_nxt = sel(h, pa[16 * (tiles > 1 ? 1 : 0)]);
    for (int t = 0; t < tiles; ++t) {
      const int t2 = t + 2 < tiles ? t + 2 : tiles - 1;
      f4 a_nn = sel(h, pa[16 * (size_t)t2]);
      f16v ex = __builtin_amdgcn_mfma_f32_32x32x2f32(a_nxt.x, bx, zero, 0, 0, 0);
      f16v ey = __builtin_amdgcn_mfma_f32_32x32x2f32(a_nxt.y, by, zero, 0, 0, 0);
      f16v ez = __builtin_amdgcn_mfma_f32_32x32x2f32(a_nxt.z, bz, zero, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) pair(dx[r], dy[r], dz[r], ax, ay, az);
      dx = ex; dy = ey; dz = ez;
      a_nxt = a_nn;
    }
  } else {
    f4 a_nxt = sel(h, pa[16 * (tiles > 1 ? 1 : 0)]);
    for (int t = 0; t < tiles; ++t) {
      f16v dx = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, bx, zero, 0, 0, 0);
      f16v dy = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, by, zero, 0, 0, 0);
      f16v dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, bz, zero, 0, 0, 0);
      a_cur = a_nxt;
      const int t2 = t + 2 < tiles ? t + 2 : tiles - 1;
      a_nxt = sel(h, pa[16 * (size_t)t2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) pair(dx[r], dy[r], dz[r], ax, ay, az);
    }
  }
  // F = S(piece 0) + S(piece 1)
  const float ox = __shfl_xor(ax, 32), oy = __shfl_xor(ay, 32), oz = __shfl_xor(az, 32);
  if (h == 0 && row < n_rows) { f4 o = {ax + ox, ay + oy, az + oz, 0.f}; out[row] = o; }
}

// the hand-scheduled form of the same loop (tools/gen_mfma_loop.py): two sets of difference registers, MFMAs of tile k+1
// beside the VALU work on tile k, every instruction placed.  V: 0 = vector instructions at 0 mod 8 bytes, 1 = at 4 mod 8,
// 2 = at 4 mod 8 with the three MFMAs grouped at the head of a tile
#include "force_loop_mfma_gfx950.inc"
template <int V>
__global__ void __launch_bounds__(256, 4) force_mfma_asm(const f4* __restrict__ src, int n_src, const f4* __restrict__ rows, f4* __restrict__ out, int n_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n32 = lane & 31, h = lane >> 5;
  const int row = (blockIdx.x * 4 + wave) * 32 + n32;
  const f4 me = rows[row < n_rows ? row : n_rows - 1];
  const float bx = h ? -me.x : 1.0f, by = h ? -me.y : 1.0f, bz = h ? -me.z : 1.0f;
  const int half_len = n_src / 2;
  const int m = n32, q = m >> 3, hp = (m >> 2) & 1, c = m & 3;
  const unsigned voff = (unsigned)(hp * half_len + 4 * q + c) * 16u;
  const int iters = half_len / 32;                 // tile pairs (the probe takes n_src a multiple of 64)
  float ax = 0.f, ay = 0.f, az = 0.f;
  const uint64_t p = (uint64_t)(uintptr_t)src;
#define NB_RUN(TEXT) asm volatile(TEXT : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az) \
                                  : [bx] "v"(bx), [by] "v"(by), [bz] "v"(bz), [voff] "v"(voff), [p] "s"(p), [iters] "s"(iters) : NB_MFMA_LOOP_CLOBBERS)
  if constexpr (V == 0) NB_RUN(NB_MFMA_LOOP_V0);
  else if constexpr (V == 1) NB_RUN(NB_MFMA_LOOP_V1);
  else if constexpr (V == 2) NB_RUN(NB_MFMA_LOOP_V2);
  else if constexpr (V == 3) NB_RUN(NB_MFMA_LOOP_V3);
  else NB_RUN(NB_MFMA_LOOP_V4);
  if constexpr (V >= 3) { asm volatile("" :: "v"(bx), "v"(by), "v"(bz)); }
#undef NB_RUN
  const float ox = __shfl_xor(ax, 32), oy = __shfl_xor(ay, 32), oz = __shfl_xor(az, 32);
  if (h == 0 && row < n_rows) { f4 o = {ax + ox, ay + oy, az + oz, 0.f}; out[row] = o; }
}

// the same sums with VALU subtractions: lane = row, two sequential sums (first half, second half of the sources), joined
__global__ void __launch_bounds__(256) force_valu(const f4* __restrict__ src, int n_src, const f4* __restrict__ rows, f4* __restrict__ out, int n_rows) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const f4 me = rows[row < n_rows ? row : n_rows - 1];
  const int half_len = n_src / 2;
  float s[2][3];
  for (int p = 0; p < 2; ++p) {
    float ax = 0.f, ay = 0.f, az = 0.f;
    const f4* sp = src + (size_t)p * half_len;
#pragma unroll 8
    for (int j = 0; j < half_len; ++j) {
      const f4 b = sp[j];
      pair(b.x - me.x, b.y - me.y, b.z - me.z, ax, ay, az);
    }
    s[p][0] = ax; s[p][1] = ay; s[p][2] = az;
  }
  if (row < n_rows) { f4 o = {s[0][0] + s[1][0], s[0][1] + s[1][1], s[0][2] + s[1][2], 0.f}; out[row] = o; }
}

// ---------------------------------------------------------------------------------------------------------------
// (4) do the fp32 MFMA and the fp32 VALU run beside each other?  Every wave: ITER x [1 v_mfma_f32_32x32x2_f32 + K independent
// v_fma_f32]; W waves per SIMD.  Separate pipes would give max(64, ~2K W) cycles per block and SIMD, one shared datapath 64 + 2K.
template <int K>
__global__ void __launch_bounds__(256) mix_mfma_valu(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = seed * 0.5f, c = seed * 0.25f;
  f16v d = {0};
  for (int it = 0; it < iters; ++it) {
    if constexpr (K >= 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(d) : "v"(b), "v"(c));
#define X8 asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
                        "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    constexpr int KK = K < 0 ? -K : K;
    if constexpr (KK >= 8) { X8 }
    if constexpr (KK >= 16) { X8 }
    if constexpr (KK >= 24) { X8 }
    if constexpr (KK >= 32) { X8 }
    if constexpr (KK >= 48) { X8 X8 }
    if constexpr (KK >= 64) { X8 X8 }
#undef X8
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d[0] + d[15];
}

// ---------------------------------------------------------------------------------------------------------------
// (5) the bf16 matrix pipe DOES run beside the VALU (MI355X_MICROARCH.md); can it return the fp32 difference?  A binary32 number
// is the exact sum of three bf16 numbers (8 + 8 + 8 significant bits, by truncation), so x_j - x_i = sum over six exact products
// {x_j.h, x_j.m, x_j.l} * 1 + 1 * {-x_i.h, -x_i.m, -x_i.l}; whether v_mfma_f32_32x32x16_bf16 returns the correctly rounded sum depends on
// how it accumulates its 16 products, which nothing documents: measured here, for two arrangements of the six terms over k.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float x, __bf16* h, __bf16* m, __bf16* l) {
  const float fh = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
  const float r1 = x - fh;                                                   // exact
  const float fm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
  const float fl = r1 - fm;                                                  // exact, <= 8 significant bits
  *h = (__bf16)fh; *m = (__bf16)fm; *l = (__bf16)fl;                         // all three conversions are exact
}
template <int ARR>
__global__ void __launch_bounds__(64) probe_bf16(const float* xj_in, const float* xi_in, float* d_out) {
  const int lane = threadIdx.x, half = lane >> 5;
  const float xj = xj_in[blockIdx.x * 32 + (lane & 31)], xi = xi_in[blockIdx.x * 32 + (lane & 31)];
  __bf16 jh, jm, jl, ih, im, il;
  split3(xj, &jh, &jm, &jl);
  split3(-xi, &ih, &im, &il);
  const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = zero; b[k] = zero; }
  if constexpr (ARR == 0) {   // k 0-2: x_j's parts (times 1), k 8-10: 1 times -x_i's parts
    if (half == 0) { a[0] = jh; a[1] = jm; a[2] = jl; b[0] = one; b[1] = one; b[2] = one; }
    else { a[0] = one; a[1] = one; a[2] = one; b[0] = ih; b[1] = im; b[2] = il; }
  } else {                    // k 0-5: x_j.h, -x_i.h, x_j.m, -x_i.m, x_j.l, -x_i.l
    if (half == 0) { a[0] = jh; a[1] = one; a[2] = jm; a[3] = one; a[4] = jl; a[5] = one;
                     b[0] = one; b[1] = ih; b[2] = one; b[3] = im; b[4] = one; b[5] = il; }
  }
  f16v c = {0};
  f16v d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d_out[((size_t)blockIdx.x * 16 + r) * 64 + lane] = d[r];
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static float uni() { return (float)((double)(rng() >> 11) / 9007199254740992.0 * 2.0 - 1.0); }

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("# device %s  CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);

  // ---- (1) layout: A lanes 0-31 = 100 + m, lanes 32-63 = 1000 (k = 1); B lanes 0-31 = 1 (k = 0), lanes 32-63 = n (k = 1)
  //      => D[m][n] = (100 + m) * 1 + 1000 * n
  {
    std::vector<float> a(64), b(64), d(16 * 64);
    for (int l = 0; l < 64; ++l) { a[l] = l < 32 ? 100.f + l : 1000.f; b[l] = l < 32 ? 1.f : (float)(l - 32); }
    float *da, *db, *dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 16 * 256));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    probe_layout<<<1, 64>>>(da, db, dd);
    CK(hipMemcpy(d.data(), dd, 16 * 256, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
      const int m = 8 * (r / 4) + 4 * (l / 32) + (r % 4), n = l % 32;
      const float want = 100.f + m + 1000.f * n;
      if (d[r * 64 + l] != want) { if (bad < 8) printf("layout: reg %d lane %d holds %g, expected D[%d][%d] = %g\n", r, l, d[r * 64 + l], m, n, want); ++bad; }
    }
    printf("layout v_mfma_f32_32x32x2_f32: A[m][k] in lane 32k+m, B[k][n] in lane 32k+n, D[8(r/4)+4(lane/32)+r%%4][lane%%32] in register r: %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dd));
  }

  // ---- (2) exactness: x_j*1 + 1*(-x_i) against the host's IEEE subtraction
  {
    const int blocks = 4096;
    std::vector<float> a((size_t)blocks * 64), b((size_t)blocks * 64), d((size_t)blocks * 1024);
    for (int k = 0; k < blocks; ++k) {
      const int mode = k % 8;
      for (int l = 0; l < 64; ++l) {
        float v;
        if (mode == 0) v = uni();                                         // the bodies' own range
        else if (mode == 1) v = std::ldexp(uni(), (int)(rng() % 60) - 30);  // wide exponent range
        else if (mode == 2) v = std::ldexp(uni(), -120 - (int)(rng() % 28)); // tiny: results and operands subnormal
        else if (mode == 3) v = (float)((int)(rng() % 5) - 2) * 0.25f;     // many equal operands (x - x = +0)
        else if (mode == 4) v = 1.0f + std::ldexp((float)(rng() % 64), -23);   // nearly equal: exact small differences
        else if (mode == 5) v = std::ldexp(uni(), 100 + (int)(rng() % 27));  // huge: overflow of the difference
        else if (mode == 6) v = (rng() % 16 == 0) ? INFINITY : uni() * 3.0e38f;
        else v = uni() * ((rng() & 1) ? 1e-3f : 1e3f);
        const bool is_k1 = l >= 32;
        // A: lanes 0-31 x_j, lanes 32-63 one;  B: lanes 0-31 one, lanes 32-63 -x_i
        a[(size_t)k * 64 + l] = is_k1 ? 1.0f : v;
        b[(size_t)k * 64 + l] = is_k1 ? -v : 1.0f;
      }
    }
    float *da, *db, *dd;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    probe_layout<<<blocks, 64>>>(da, db, dd);
    CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
    long bad[8] = {0}, tot[8] = {0};
    for (int k = 0; k < blocks; ++k) for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
      const int m = 8 * (r / 4) + 4 * (l / 32) + (r % 4), n = l % 32;
      const float xj = a[(size_t)k * 64 + m], xi = -b[(size_t)k * 64 + 32 + n];
      volatile float want = xj - xi;
      float got = d[((size_t)k * 16 + r) * 64 + l];
      float w = want;
      uint32_t gw, ww; memcpy(&gw, &got, 4); memcpy(&ww, &w, 4);
      const bool both_nan = (got != got) && (w != w);
      ++tot[k % 8];
      if (gw != ww && !both_nan) {
        if (bad[k % 8] < 3) printf("exactness mode %d: %a - %a: mfma %a (0x%08x), v_sub %a (0x%08x)\n", k % 8, xj, xi, got, gw, w, ww);
        ++bad[k % 8];
      }
    }
    const char* names[8] = {"uniform [-1,1)", "exponents 2^-30..2^30", "subnormal range", "equal operands", "nearly equal", "huge (overflow)", "inf / 3e38", "1e-3 | 1e3"};
    for (int k = 0; k < 8; ++k) printf("exactness %-24s %ld of %ld differ from IEEE x_j - x_i\n", names[k], bad[k], tot[k]);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dd));
  }

  // ---- (3) the loop
  const int n = argc > 1 ? atoi(argv[1]) : 262144;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<f4> pos(n);
  rng_state = 42;
  for (int i = 0; i < n; ++i) { f4 p = {uni(), uni(), uni(), 1.0f}; pos[i] = p; }
  f4 *dsrc, *dout0, *dout1;
  CK(hipMalloc(&dsrc, (size_t)(n + 256) * 16)); CK(hipMemset(dsrc, 0, (size_t)(n + 256) * 16)); CK(hipMalloc(&dout0, (size_t)n * 16)); CK(hipMalloc(&dout1, (size_t)n * 16));
  CK(hipMemcpy(dsrc, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<f4> ref(n), got(n);
  auto timeit = [&](const char* name, auto launch, f4* dout, std::vector<f4>& host) {
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best; sum += ms;
    }
    CK(hipMemcpy(host.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost));
    const double pairs = (double)n * n;
    printf("%-44s n=%d  best %.3f ms  avg %.3f ms  %.1f G pairs/s (best)  %.2f cycles per wave-pair at 2.4 GHz\n", name, n, best, sum / reps,
           pairs / best / 1e6, best * 1e-3 * 2.4e9 * 1024.0 / (pairs / 64.0));
  };
  timeit("VALU subtractions (compiled, lane = row)", [&] { force_valu<<<(n + 255) / 256, 256>>>(dsrc, n, dsrc, dout0, n); }, dout0, ref);
  auto cmp = [&](const char* name) {
    long bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
      if (memcmp(&ref[i], &got[i], 12) != 0) {
        ++bad;
        for (int k = 0; k < 3; ++k) { double e = std::fabs((double)ref[i][k] - got[i][k]) / (std::fabs((double)ref[i][k]) + 1e-30); worst = e > worst ? e : worst; }
      }
    }
    printf("  %s vs VALU kernel: %ld of %d rows differ bitwise (worst component rel. diff %.3g)\n", name, bad, n, worst);
  };
  const int grid = (n + 127) / 128;
  timeit("MFMA differences, one D tile, >= 4 waves/SIMD", [&] { force_mfma<0, 4><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("single 4");
  timeit("MFMA differences, one D tile, >= 6 waves/SIMD", [&] { force_mfma<0, 6><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("single 6");
  timeit("MFMA differences, one D tile, >= 8 waves/SIMD", [&] { force_mfma<0, 8><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("single 8");
  timeit("MFMA differences, two D tiles, >= 4 waves/SIMD", [&] { force_mfma<1, 4><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("double 4");
  timeit("MFMA differences, two D tiles, >= 2 waves/SIMD", [&] { force_mfma<1, 2><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("double 2");
  timeit("MFMA differences, hand-scheduled, 0 mod 8", [&] { force_mfma_asm<0><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("asm V0");
  timeit("MFMA differences, hand-scheduled, 4 mod 8", [&] { force_mfma_asm<1><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("asm V1");
  timeit("MFMA differences, hand-sched., MFMAs grouped", [&] { force_mfma_asm<2><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got); cmp("asm V2");
  timeit("TIMING ONLY: bf16 MFMAs in that loop", [&] { force_mfma_asm<3><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got);
  timeit("TIMING ONLY: bf16 MFMAs, grouped", [&] { force_mfma_asm<4><<<grid, 256>>>(dsrc, n, dsrc, dout1, n); }, dout1, got);


  // ---- (5) the fp32 difference from three bf16 parts on the bf16 matrix pipe
  {
    const int blocks = 4096;
    std::vector<float> xj((size_t)blocks * 32), xi((size_t)blocks * 32), d((size_t)blocks * 1024);
    for (int k = 0; k < blocks; ++k) for (int l = 0; l < 32; ++l) {
      const int mode = k % 4;
      float a = uni(), b = uni();
      if (mode == 1) { b = a + std::ldexp(uni(), -8 - (int)(rng() % 12)); }          // close pairs: the case the force depends on
      if (mode == 2) { a = std::ldexp(a, (int)(rng() % 16) - 8); b = std::ldexp(b, (int)(rng() % 16) - 8); }
      if (mode == 3) { a = uni() * 4.0f; b = uni() * 4.0f; }
      xj[(size_t)k * 32 + l] = a; xi[(size_t)k * 32 + l] = b;
    }
    float *dj, *di, *dd;
    CK(hipMalloc(&dj, xj.size() * 4)); CK(hipMalloc(&di, xi.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
    CK(hipMemcpy(dj, xj.data(), xj.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(di, xi.data(), xi.size() * 4, hipMemcpyHostToDevice));
    for (int arr = 0; arr < 2; ++arr) {
      if (arr == 0) probe_bf16<0><<<blocks, 64>>>(dj, di, dd); else probe_bf16<1><<<blocks, 64>>>(dj, di, dd);
      CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
      long bad[4] = {0}, tot[4] = {0}; double worst[4] = {0};
      int shown = 0;
      for (int k = 0; k < blocks; ++k) for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
        const int m = 8 * (r / 4) + 4 * (l / 32) + (r % 4), n = l % 32;
        volatile float want = xj[(size_t)k * 32 + m] - xi[(size_t)k * 32 + n];
        const float w = want, got = d[((size_t)k * 16 + r) * 64 + l];
        ++tot[k % 4];
        if (memcmp(&w, &got, 4) != 0) {
          ++bad[k % 4];
          const double ulp = std::ldexp(1.0, std::ilogb(w == 0 ? 1e-30f : w) - 23);
          const double e = std::fabs((double)got - (double)w) / ulp;
          worst[k % 4] = e > worst[k % 4] ? e : worst[k % 4];
          if (shown < 4) { printf("bf16 arrangement %d: %a - %a: mfma %a, v_sub %a\n", arr, xj[(size_t)k * 32 + m], xi[(size_t)k * 32 + n], got, w); ++shown; }
        }
      }
      const char* names[4] = {"uniform [-1,1)", "close pairs", "exponents 2^-8..2^8", "uniform [-4,4)"};
      for (int q = 0; q < 4; ++q) printf("bf16 x 3 difference, arrangement %d, %-20s %ld of %ld differ from IEEE x_j - x_i (worst %.2f ulp)\n", arr, names[q], bad[q], tot[q], worst[q]);
    }
    CK(hipFree(dj)); CK(hipFree(di)); CK(hipFree(dd));
  }

  // ---- (4) MFMA beside VALU
  {
    float* dout;
    const int cus = prop.multiProcessorCount;
    CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4));
    const int iters = 20000;
    auto one = [&](const char* name, auto kernel, int k_valu, int mfma, int wpc) {
      kernel<<<cus * wpc, 256>>>(dout, 100, 1.5f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      kernel<<<cus * wpc, 256>>>(dout, iters, 1.5f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // cycles per block of (1 MFMA + K VALU) per SIMD at 2.4 GHz: wpc waves share a SIMD
      printf("mix %-34s W=%d  %.1f cycles per block and SIMD at 2.4 GHz (separate pipes: max(%d, %d); one datapath: %d)\n", name, wpc,
             ms * 1e-3 * 2.4e9 / iters / wpc * 1.0, mfma ? 64 : 0, 2 * k_valu, (mfma ? 64 : 0) + 2 * k_valu);
    };
    for (int wpc : {1, 2, 4}) {
      one("1 MFMA f32 32x32x2 alone", mix_mfma_valu<0>, 0, 1, wpc);
      one("32 v_fma_f32 alone", mix_mfma_valu<-32>, 32, 0, wpc);
      one("1 MFMA + 16 v_fma_f32", mix_mfma_valu<16>, 16, 1, wpc);
      one("1 MFMA + 32 v_fma_f32", mix_mfma_valu<32>, 32, 1, wpc);
      one("1 MFMA + 48 v_fma_f32", mix_mfma_valu<48>, 48, 1, wpc);
      one("1 MFMA + 64 v_fma_f32", mix_mfma_valu<64>, 64, 1, wpc);
    }
    CK(hipFree(dout));
  }
  return 0;
}
