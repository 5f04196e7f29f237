/* nbody_ref.c — CPU ORACLE (test infrastructure, see nbody_ref.h).
 *
 * A plain-C restatement of the reference FPGA pipeline's arithmetic
 * (/root/reference/vec_add.srcs/sources_1/new/, "S/").  Every function cites
 * the reference lines it follows.  The reference has no host program, no
 * integrator and no CPU path; bodyForce()/integrate() follow the
 * north_star text and are this build's own definitions.
 *
 * PINNING.  **Parity unpinned at the reference level**: the reference's testbenches assert no values (T/tb_dxy.vhd:907-918,
 * T/tb_sqrt.vhd:562-573 check only "not X"), it holds no expected outputs, and it cannot be built here (VHDL-2008 + seven vendor IP
 * cores that are not in the tree: there is no oracle/_ref).  What pins this restatement instead: the testbenches' own stimuli with
 * analytically derived outputs (tests/golden/kat_*.json), a cycle model of the scatter logic that IS in the tree
 * (tests/test_fpga_scatter_model.py), and independent exact-rational third statements of the whole pipeline
 * (tests/golden/make_system.py -> system_*.json, rtl_*.json; make_fp64.py -> fp64_n64.json), all reproduced bit for bit (fp64: bounded).
 *
 * Build flavours (oracle/Makefile):
 *   libnbody_ref.so       -O2 -ffp-contract=off            bit-reproducible parity oracle
 *   libnbody_ref_fast.so  -O3 -ffp-contract=off -fopenmp   same source, same results, timed as cpu_baseline
 * -ffp-contract=off everywhere: a product is fused only where the source says fmaf().
 */
#include "nbody_ref.h"
#include "../include/nbody_ic.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* S/dzsoft.vhd:177  SOFT = real_to_flt(1.0E-9, normal, 32, 24) = 0x3089705F */
float ref_soft(void) { return bits_to_float(REF_SOFT_BITS); }

/* S/dxy.vhd:94-98   X_DIFF: a = x_target, b = x_this  ->  dx = x_target - x_this
 * S/dxy.vhd:113-117 dx*dx, dy*dy   (each rounded)
 * S/dxy.vhd:121-122 sum = dx^2 + dy^2 (rounded) */
static inline float dxy_(float x_this, float x_target, float y_this, float y_target, float *dx, float *dy) {
  float ddx = x_target - x_this;
  float ddy = y_target - y_this;
  float dx_sq = ddx * ddx;
  float dy_sq = ddy * ddy;
  *dx = ddx; *dy = ddy;
  return dx_sq + dy_sq;
}
float ref_dxy(float x_this, float x_target, float y_this, float y_target, float *dx, float *dy) {
  return dxy_(x_this, x_target, y_this, y_target, dx, dy);
}

/* S/dzsoft.vhd:186-187 dz = z_target - z_this
 * S/dzsoft.vhd:201-202 sum = fma(dz, dz, SOFTENING): one rounding */
static inline float dzsoft_(float z_this, float z_target, float soft, float *dz) {
  float ddz = z_target - z_this;
  *dz = ddz;
  return fmaf(ddz, ddz, soft);
}
float ref_dzsoft(float z_this, float z_target, float *dz) { return dzsoft_(z_this, z_target, ref_soft(), dz); }

/* S/dxyz_soft.vhd:87-93 the two paths; :149-150 PATH_CONV: dist_sqr = (dx^2+dy^2) + (dz^2+eps) */
static inline float dxyz_soft_(float xi, float xt, float yi, float yt, float zi, float zt, float soft,
                               float *dx, float *dy, float *dz) {
  float sxy = dxy_(xi, xt, yi, yt, dx, dy);
  float sz = dzsoft_(zi, zt, soft, dz);
  return sxy + sz;
}
float ref_dxyz_soft(float x_this, float x_target, float y_this, float y_target, float z_this, float z_target,
                    float *dx, float *dy, float *dz) {
  return dxyz_soft_(x_this, x_target, y_this, y_target, z_this, z_target, ref_soft(), dx, dy, dz);
}

/* The contraction the GPU kernel uses for the same quantity:
 * fma(dx,dx, fma(dy,dy, fma(dz,dz,eps))) — 3 roundings instead of 5. */
static inline float d2_fma3_(float dx, float dy, float dz, float soft) {
  return fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, soft)));
}
float ref_d2_fma3(float dx, float dy, float dz) { return d2_fma3_(dx, dy, dz, ref_soft()); }

/* S/fxyz.vhd:101-102 STAGE2: rsqrt IP.  Its rounding is unpinned (no .xci in the
 * tree; T/tb_sqrt.vhd checks no values).  F64: the correctly rounded value up to
 * a double-rounding event; DIVSQRT: what plain C code computes in fp32. */
static inline float rsqrt_f64_(float d2) { return (float)(1.0 / sqrt((double)d2)); }
static inline float rsqrt_divsqrt_(float d2) { return 1.0f / sqrtf(d2); }
float ref_rsqrt(float d2, int rsqrt_mode) { return rsqrt_mode == REF_RSQRT_DIVSQRT ? rsqrt_divsqrt_(d2) : rsqrt_f64_(d2); }

/* S/cube.vhd:66-67 INVDIST2 = inv*inv; :69-70 INVDIST3 = inv_delayed * INVDIST2 */
static inline float cube_(float inv) { float inv2 = inv * inv; return inv * inv2; }
float ref_cube(float inv) { return cube_(inv); }

/* S/final_adder.vhd:88-100: level I node J = node(I+1, 2J) + node(I+1, 2J+1);
 * :102 leaves = buff(0..15); :104 sum = root.  For 16 leaves every node is a
 * real_add (adder_structure, :42-68), each an IP add (S/adder_choose.vhd:81-82). */
static inline float tree16_(const float p[16]) {
  float l3[8], l2[4], l1[2];
  for (int j = 0; j < 8; ++j) l3[j] = p[2 * j] + p[2 * j + 1];
  for (int j = 0; j < 4; ++j) l2[j] = l3[2 * j] + l3[2 * j + 1];
  for (int j = 0; j < 2; ++j) l1[j] = l2[2 * j] + l2[2 * j + 1];
  return l1[0] + l1[1];
}
float ref_tree16(const float p[16]) { return tree16_(p); }

/* ------------------------------------------------------------------------- */
/* One pair: returns inv3, fills d.  S/fxyz.vhd:97-106 (STAGE1..3).            */
#define PAIR_F32(D2MODE, RSQMODE)                                                     \
  float dx, dy, dz, d2;                                                               \
  if ((D2MODE) == REF_D2_REFERENCE) {                                                 \
    d2 = dxyz_soft_(xi, xt, yi, yt, zi, zt, soft, &dx, &dy, &dz);                     \
  } else {                                                                            \
    dx = xt - xi; dy = yt - yi; dz = zt - zi;                                         \
    d2 = d2_fma3_(dx, dy, dz, soft);                                                  \
  }                                                                                   \
  float inv = (RSQMODE) == REF_RSQRT_DIVSQRT ? rsqrt_divsqrt_(d2) : rsqrt_f64_(d2);   \
  float inv3 = cube_(inv);

/* Sequential order: one accumulator per axis, sources ascending
 * (S/top_level.vhd:233-254 streams TRGT_PTR = 1..N, self included), each term
 * added by an fma as in S/fxyz.vhd:120-127.  Rows are processed LANES at a time
 * so the compiler can vectorise over rows; per-row order is unchanged. */
#define LANES 16
#define DEFINE_SEQ_KERNEL(NAME, D2MODE, RSQMODE)                                                        \
  static void NAME(const float *rows, int n_rows, const float *src, int n_src, const float *acc_in,     \
                   float *acc) {                                                                        \
    const float soft = bits_to_float(REF_SOFT_BITS);                                                    \
    const int nblk = (n_rows + LANES - 1) / LANES;                                                      \
    _Pragma("omp parallel for schedule(dynamic, 4)")                                                    \
    for (int b = 0; b < nblk; ++b) {                                                                    \
      float xs[LANES], ys[LANES], zs[LANES], fx[LANES], fy[LANES], fz[LANES];                           \
      for (int l = 0; l < LANES; ++l) {                                                                 \
        int i = b * LANES + l; if (i >= n_rows) i = n_rows - 1;                                         \
        xs[l] = rows[4 * i]; ys[l] = rows[4 * i + 1]; zs[l] = rows[4 * i + 2];                          \
        fx[l] = acc_in ? acc_in[4 * i] : 0.0f;                                                          \
        fy[l] = acc_in ? acc_in[4 * i + 1] : 0.0f;                                                      \
        fz[l] = acc_in ? acc_in[4 * i + 2] : 0.0f;                                                      \
      }                                                                                                 \
      for (int j = 0; j < n_src; ++j) {                                                                 \
        const float xt = src[4 * j], yt = src[4 * j + 1], zt = src[4 * j + 2];                          \
        _Pragma("omp simd")                                                                             \
        for (int l = 0; l < LANES; ++l) {                                                               \
          const float xi = xs[l], yi = ys[l], zi = zs[l];                                               \
          PAIR_F32(D2MODE, RSQMODE)                                                                     \
          fx[l] = fmaf(dx, inv3, fx[l]);                                                                \
          fy[l] = fmaf(dy, inv3, fy[l]);                                                                \
          fz[l] = fmaf(dz, inv3, fz[l]);                                                                \
        }                                                                                               \
      }                                                                                                 \
      for (int l = 0; l < LANES; ++l) {                                                                 \
        int i = b * LANES + l; if (i >= n_rows) break;                                                  \
        acc[4 * i] = fx[l]; acc[4 * i + 1] = fy[l]; acc[4 * i + 2] = fz[l]; acc[4 * i + 3] = 0.0f;      \
      }                                                                                                 \
    }                                                                                                   \
  }

DEFINE_SEQ_KERNEL(seq_ref_f64, REF_D2_REFERENCE, REF_RSQRT_F64)
DEFINE_SEQ_KERNEL(seq_ref_div, REF_D2_REFERENCE, REF_RSQRT_DIVSQRT)
DEFINE_SEQ_KERNEL(seq_fma_f64, REF_D2_FMA3, REF_RSQRT_F64)
DEFINE_SEQ_KERNEL(seq_fma_div, REF_D2_FMA3, REF_RSQRT_DIVSQRT)

/* FPGA order.
 * S/fxyz.vhd:129-145: the fma's c input is 0.0 for the first fma_latency (16)
 *   items of a stream and the fma's own output (16 cycles old) afterwards, so
 *   partial k sums the sources j == k (mod 16), each in ascending order.
 * S/fxyz.vhd:147-184: when the stream ends the 16 in-flight outputs are latched
 *   in the order they leave the pipe: results(t) = partial of item n_src-16+t,
 *   i.e. partial[(n_src + t) mod 16]; slots with no item hold 0 (:177-181).
 * S/compute_store.vhd:139-173 feeds results(0..15) of one axis to the tree;
 * S/final_adder.vhd:88-104 sums them pairwise.
 * The rotation by n_src mod 16 is not taken on trust: tests/test_fpga_scatter_model.py simulates the control logic of
 * S/fxyz.vhd:129-184 cycle by cycle (FLUSH_CNT, the feedback mux, SCTTR_CNT, VALID_FMA_PREV, the results latch; the fma IP
 * as the 16-stage pipeline S/top_level.vhd:40 says it is) for streams of 1..257 items and finds results(t) = the final
 * value of partial (n_src + t) mod 16, zero where no item exists; the same model with this file's arithmetic plus the
 * tree reproduces fpga16_f32() bit for bit. */
static void fpga16_f32(const float *rows, int n_rows, const float *src, int n_src, float *acc, int d2_mode,
                       int rsqrt_mode) {
  const float soft = bits_to_float(REF_SOFT_BITS);
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n_rows; ++i) {
    const float xi = rows[4 * i], yi = rows[4 * i + 1], zi = rows[4 * i + 2];
    float px[16], py[16], pz[16];
    for (int k = 0; k < 16; ++k) px[k] = py[k] = pz[k] = 0.0f;
    for (int j = 0; j < n_src; ++j) {
      const float xt = src[4 * j], yt = src[4 * j + 1], zt = src[4 * j + 2];
      PAIR_F32(d2_mode, rsqrt_mode)
      const int k = j & 15;
      px[k] = fmaf(dx, inv3, px[k]);
      py[k] = fmaf(dy, inv3, py[k]);
      pz[k] = fmaf(dz, inv3, pz[k]);
    }
    float rx[16], ry[16], rz[16];
    for (int t = 0; t < 16; ++t) {
      int item = n_src - 16 + t;
      if (item < 0) { rx[t] = ry[t] = rz[t] = 0.0f; }
      else { int k = item & 15; rx[t] = px[k]; ry[t] = py[k]; rz[t] = pz[k]; }
    }
    acc[4 * i] = tree16_(rx); acc[4 * i + 1] = tree16_(ry); acc[4 * i + 2] = tree16_(rz);
    acc[4 * i + 3] = 0.0f; /* S/compute_store.vhd:242 {0,Fz,Fy,Fx} */
  }
}

void ref_forces_f32(const float *rows, int n_rows, const float *src, int n_src, const float *acc_in, float *acc,
                    int d2_mode, int rsqrt_mode, int sum_mode) {
  if (n_rows <= 0) return;
  if (sum_mode == REF_SUM_FPGA16) { fpga16_f32(rows, n_rows, src, n_src, acc, d2_mode, rsqrt_mode); return; }
  if (sum_mode == REF_SUM_BLOCKED) {   /* one segment, blocks of 1024 (the engine's default block) */
    ref_order_t o = {d2_mode, rsqrt_mode, REF_SUM_BLOCKED, 1024, 1, 1, 1};
    ref_forces_f32_order(rows, n_rows, src, n_src, acc, &o);
    return;
  }
  if (d2_mode == REF_D2_REFERENCE) {
    if (rsqrt_mode == REF_RSQRT_DIVSQRT) seq_ref_div(rows, n_rows, src, n_src, acc_in, acc);
    else seq_ref_f64(rows, n_rows, src, n_src, acc_in, acc);
  } else {
    if (rsqrt_mode == REF_RSQRT_DIVSQRT) seq_fma_div(rows, n_rows, src, n_src, acc_in, acc);
    else seq_fma_f64(rows, n_rows, src, n_src, acc_in, acc);
  }
}

/* ------------------------------------------------------------------------- */
/* The engine's summation order (ref_order_t in nbody_ref.h).
 * Slices and pieces as mini_nbody_amd/csrc/nbody_args.hpp slice_first()/segment_bounds(): slice q of P is balanced
 * (the first n % P slices have one body more); a slice is cut into `sub` pieces of ceil(len / sub) sources. */
static int slice_first_(int q, int n, int P) {
  int base = n / P, rem = n % P;
  return q * base + (q < rem ? q : rem);
}
void ref_segment_bounds(int q, int t, int n, int nslices, int sub, int *jb, int *je) {
  int f0 = slice_first_(q, n, nslices), f1 = slice_first_(q + 1, n, nslices);
  int len = f1 - f0;
  int piece = (len + sub - 1) / sub;
  int b = f0 + t * piece;
  int e = b + piece;
  if (b > f1) b = f1;
  if (e > f1) e = f1;
  *jb = b; *je = e;
}

/* piece w of `ws` of the segment [jb, je): ceil(len / ws) sources each, the last ones shorter or empty */
void ref_piece_bounds(int jb, int je, int w, int ws, int *pb, int *pe) {
  int piece = (je - jb + ws - 1) / ws;
  int b = jb + w * piece;
  int e = b + piece;
  if (b > je) b = je;
  if (e > je) e = je;
  *pb = b; *pe = e;
}

/* Per row: for every segment in ascending order { for every piece w of the segment's `wsplit` (1 or 4: the four waves of a
 * workgroup, mini_nbody_amd/csrc/nbody_args.hpp piece_bounds()) in ascending order { level 1: a = 0, a = fma(d, inv3, a)
 * over a block of sources counted from the piece's first (S/fxyz.vhd:120-127); level 2: b = b + a per finished block };
 * level 3: the segment's sum = ((b_0 + b_1) + b_2) + b_3 — the reference's own remedy, partial sums joined by an adder
 * (S/fxyz.vhd:129-145, S/final_adder.vhd:88-104), in the shape a SIMT workgroup has }, then F = p_0, F = F + p_s.
 * In REF_SUM_SEQ a piece is one block and its sum is `a` itself (no addition of zero); an empty piece contributes +0.
 * Rows are processed LANES at a time for the vectoriser; the order of operations per row is exactly the one written here. */
#define DEFINE_ORDER_KERNEL(NAME, D2MODE, RSQMODE)                                                       \
  static void NAME(const float *rows, int n_rows, const float *src, int n_src, float *acc,               \
                   int blocked, int block, int nslices, int sub, int wsplit) {                           \
    const float soft = bits_to_float(REF_SOFT_BITS);                                                     \
    const int nblk = (n_rows + LANES - 1) / LANES;                                                       \
    _Pragma("omp parallel for schedule(dynamic, 4)")                                                     \
    for (int b = 0; b < nblk; ++b) {                                                                     \
      float xs[LANES], ys[LANES], zs[LANES], tx[LANES], ty[LANES], tz[LANES];                            \
      for (int l = 0; l < LANES; ++l) {                                                                  \
        int i = b * LANES + l; if (i >= n_rows) i = n_rows - 1;                                          \
        xs[l] = rows[4 * i]; ys[l] = rows[4 * i + 1]; zs[l] = rows[4 * i + 2];                           \
        tx[l] = ty[l] = tz[l] = 0.0f;                                                                    \
      }                                                                                                  \
      for (int seg = 0; seg < nslices * sub; ++seg) {                                                    \
        int sb, se;                                                                                      \
        ref_segment_bounds(seg / sub, seg % sub, n_src, nslices, sub, &sb, &se);                         \
        float gx[LANES], gy[LANES], gz[LANES];                                                           \
        for (int l = 0; l < LANES; ++l) gx[l] = gy[l] = gz[l] = 0.0f;                                    \
        for (int w = 0; w < wsplit; ++w) {                                                               \
          int jb, je;                                                                                    \
          ref_piece_bounds(sb, se, w, wsplit, &jb, &je);                                                 \
          float px[LANES], py[LANES], pz[LANES];                                                         \
          for (int l = 0; l < LANES; ++l) px[l] = py[l] = pz[l] = 0.0f;                                  \
          const int step = blocked ? block : (je - jb > 0 ? je - jb : 1);                                \
          for (int j0 = jb; j0 < je; j0 += step) {                                                       \
            const int j1 = j0 + step < je ? j0 + step : je;                                              \
            float fx[LANES], fy[LANES], fz[LANES];                                                       \
            for (int l = 0; l < LANES; ++l) fx[l] = fy[l] = fz[l] = 0.0f;                                \
            for (int j = j0; j < j1; ++j) {                                                              \
              const float xt = src[4 * j], yt = src[4 * j + 1], zt = src[4 * j + 2];                     \
              _Pragma("omp simd")                                                                        \
              for (int l = 0; l < LANES; ++l) {                                                          \
                const float xi = xs[l], yi = ys[l], zi = zs[l];                                          \
                PAIR_F32(D2MODE, RSQMODE)                                                                \
                fx[l] = fmaf(dx, inv3, fx[l]);                                                           \
                fy[l] = fmaf(dy, inv3, fy[l]);                                                           \
                fz[l] = fmaf(dz, inv3, fz[l]);                                                           \
              }                                                                                          \
            }                                                                                            \
            if (blocked) { for (int l = 0; l < LANES; ++l) { px[l] = px[l] + fx[l]; py[l] = py[l] + fy[l]; pz[l] = pz[l] + fz[l]; } } \
            else { for (int l = 0; l < LANES; ++l) { px[l] = fx[l]; py[l] = fy[l]; pz[l] = fz[l]; } }    \
          }                                                                                              \
          if (w == 0) { for (int l = 0; l < LANES; ++l) { gx[l] = px[l]; gy[l] = py[l]; gz[l] = pz[l]; } } \
          else { for (int l = 0; l < LANES; ++l) { gx[l] = gx[l] + px[l]; gy[l] = gy[l] + py[l]; gz[l] = gz[l] + pz[l]; } } \
        }                                                                                                \
        if (seg == 0) { for (int l = 0; l < LANES; ++l) { tx[l] = gx[l]; ty[l] = gy[l]; tz[l] = gz[l]; } } \
        else { for (int l = 0; l < LANES; ++l) { tx[l] = tx[l] + gx[l]; ty[l] = ty[l] + gy[l]; tz[l] = tz[l] + gz[l]; } } \
      }                                                                                                  \
      for (int l = 0; l < LANES; ++l) {                                                                  \
        int i = b * LANES + l; if (i >= n_rows) break;                                                   \
        acc[4 * i] = tx[l]; acc[4 * i + 1] = ty[l]; acc[4 * i + 2] = tz[l]; acc[4 * i + 3] = 0.0f;       \
      }                                                                                                  \
    }                                                                                                    \
  }

DEFINE_ORDER_KERNEL(ord_ref_f64, REF_D2_REFERENCE, REF_RSQRT_F64)
DEFINE_ORDER_KERNEL(ord_ref_div, REF_D2_REFERENCE, REF_RSQRT_DIVSQRT)
DEFINE_ORDER_KERNEL(ord_fma_f64, REF_D2_FMA3, REF_RSQRT_F64)
DEFINE_ORDER_KERNEL(ord_fma_div, REF_D2_FMA3, REF_RSQRT_DIVSQRT)

void ref_forces_f32_order(const float *rows, int n_rows, const float *src, int n_src, float *acc, const ref_order_t *o) {
  if (n_rows <= 0) return;
  const int blocked = o->sum_mode == REF_SUM_BLOCKED;
  const int block = o->block > 0 ? o->block : 1024;
  const int nsl = o->nslices > 0 ? o->nslices : 1, sub = o->sub > 0 ? o->sub : 1, ws = o->wsplit > 0 ? o->wsplit : 1;
  if (o->d2_mode == REF_D2_REFERENCE) {
    if (o->rsqrt_mode == REF_RSQRT_DIVSQRT) ord_ref_div(rows, n_rows, src, n_src, acc, blocked, block, nsl, sub, ws);
    else ord_ref_f64(rows, n_rows, src, n_src, acc, blocked, block, nsl, sub, ws);
  } else {
    if (o->rsqrt_mode == REF_RSQRT_DIVSQRT) ord_fma_div(rows, n_rows, src, n_src, acc, blocked, block, nsl, sub, ws);
    else ord_fma_f64(rows, n_rows, src, n_src, acc, blocked, block, nsl, sub, ws);
  }
}

/* fp64: the same expression tree as the GPU's fp64 kernel:
 * d2 = fma(dx,dx,fma(dy,dy,fma(dz,dz,eps))), inv = 1/sqrt(d2), inv3 = inv*(inv*inv),
 * F = fma(d, inv3, F); eps = (double)(float)1e-9 so both precisions soften alike. */
#define DEFINE_F64_KERNEL(NAME, T_IN)                                                                   \
  void NAME(const T_IN *rows, int n_rows, const T_IN *src, int n_src, double *acc) {                    \
    const double soft = (double)bits_to_float(REF_SOFT_BITS);                                           \
    const int nblk = (n_rows + LANES - 1) / LANES;                                                      \
    _Pragma("omp parallel for schedule(dynamic, 4)")                                                    \
    for (int b = 0; b < nblk; ++b) {                                                                    \
      double xs[LANES], ys[LANES], zs[LANES], fx[LANES], fy[LANES], fz[LANES];                          \
      for (int l = 0; l < LANES; ++l) {                                                                 \
        int i = b * LANES + l; if (i >= n_rows) i = n_rows - 1;                                         \
        xs[l] = (double)rows[4 * i]; ys[l] = (double)rows[4 * i + 1]; zs[l] = (double)rows[4 * i + 2];  \
        fx[l] = fy[l] = fz[l] = 0.0;                                                                    \
      }                                                                                                 \
      for (int j = 0; j < n_src; ++j) {                                                                 \
        const double xt = (double)src[4 * j], yt = (double)src[4 * j + 1], zt = (double)src[4 * j + 2]; \
        _Pragma("omp simd")                                                                             \
        for (int l = 0; l < LANES; ++l) {                                                               \
          double dx = xt - xs[l], dy = yt - ys[l], dz = zt - zs[l];                                     \
          double d2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, soft)));                                      \
          double inv = 1.0 / sqrt(d2);                                                                  \
          double inv2 = inv * inv;                                                                      \
          double inv3 = inv * inv2;                                                                     \
          fx[l] = fma(dx, inv3, fx[l]); fy[l] = fma(dy, inv3, fy[l]); fz[l] = fma(dz, inv3, fz[l]);     \
        }                                                                                               \
      }                                                                                                 \
      for (int l = 0; l < LANES; ++l) {                                                                 \
        int i = b * LANES + l; if (i >= n_rows) break;                                                  \
        acc[4 * i] = fx[l]; acc[4 * i + 1] = fy[l]; acc[4 * i + 2] = fz[l]; acc[4 * i + 3] = 0.0;       \
      }                                                                                                 \
    }                                                                                                   \
  }
DEFINE_F64_KERNEL(ref_forces_f64, double)
DEFINE_F64_KERNEL(ref_forces_f64_from_f32, float)

/* fp64 in the engine's summation order (NBODY_ARITH_STRICT in an fp64 context matches this bit for bit): per row, for every segment
 * in ascending order { for every piece of the wave split in ascending order: ONE sequential sum from +0 (fp64 contexts do not block);
 * the segment's sum = ((p_0 + p_1) + p_2) + ... } then F = g_0, F = F + g_s — ref_order_t's nslices, sub and wsplit; block and the
 * d2 / rsqrt modes do not apply (one d2 form, IEEE 1.0 / sqrt). */
void ref_forces_f64_order(const double *rows, int n_rows, const double *src, int n_src, double *acc, const ref_order_t *o) {
  if (n_rows <= 0) return;
  const double soft = (double)bits_to_float(REF_SOFT_BITS);
  const int nslices = o->nslices > 0 ? o->nslices : 1, sub = o->sub > 0 ? o->sub : 1, wsplit = o->wsplit > 0 ? o->wsplit : 1;
  const int nblk = (n_rows + LANES - 1) / LANES;
  _Pragma("omp parallel for schedule(dynamic, 4)")
  for (int b = 0; b < nblk; ++b) {
    double xs[LANES], ys[LANES], zs[LANES], tx[LANES], ty[LANES], tz[LANES];
    for (int l = 0; l < LANES; ++l) {
      int i = b * LANES + l; if (i >= n_rows) i = n_rows - 1;
      xs[l] = rows[4 * i]; ys[l] = rows[4 * i + 1]; zs[l] = rows[4 * i + 2];
      tx[l] = ty[l] = tz[l] = 0.0;
    }
    for (int seg = 0; seg < nslices * sub; ++seg) {
      int sb, se;
      ref_segment_bounds(seg / sub, seg % sub, n_src, nslices, sub, &sb, &se);
      double gx[LANES], gy[LANES], gz[LANES];
      for (int l = 0; l < LANES; ++l) gx[l] = gy[l] = gz[l] = 0.0;
      for (int w = 0; w < wsplit; ++w) {
        int jb, je;
        ref_piece_bounds(sb, se, w, wsplit, &jb, &je);
        double fx[LANES], fy[LANES], fz[LANES];
        for (int l = 0; l < LANES; ++l) fx[l] = fy[l] = fz[l] = 0.0;
        for (int j = jb; j < je; ++j) {
          const double xt = src[4 * j], yt = src[4 * j + 1], zt = src[4 * j + 2];
          _Pragma("omp simd")
          for (int l = 0; l < LANES; ++l) {
            double dx = xt - xs[l], dy = yt - ys[l], dz = zt - zs[l];
            double d2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, soft)));
            double inv = 1.0 / sqrt(d2);
            double inv2 = inv * inv;
            double inv3 = inv * inv2;
            fx[l] = fma(dx, inv3, fx[l]); fy[l] = fma(dy, inv3, fy[l]); fz[l] = fma(dz, inv3, fz[l]);
          }
        }
        if (w == 0) { for (int l = 0; l < LANES; ++l) { gx[l] = fx[l]; gy[l] = fy[l]; gz[l] = fz[l]; } }
        else { for (int l = 0; l < LANES; ++l) { gx[l] = gx[l] + fx[l]; gy[l] = gy[l] + fy[l]; gz[l] = gz[l] + fz[l]; } }
      }
      if (seg == 0) { for (int l = 0; l < LANES; ++l) { tx[l] = gx[l]; ty[l] = gy[l]; tz[l] = gz[l]; } }
      else { for (int l = 0; l < LANES; ++l) { tx[l] = tx[l] + gx[l]; ty[l] = ty[l] + gy[l]; tz[l] = tz[l] + gz[l]; } }
    }
    for (int l = 0; l < LANES; ++l) {
      int i = b * LANES + l; if (i >= n_rows) break;
      acc[4 * i] = tx[l]; acc[4 * i + 1] = ty[l]; acc[4 * i + 2] = tz[l]; acc[4 * i + 3] = 0.0;
    }
  }
}

/* bodyForce(): kick.  v += dt * F with one rounding per component.  No reference
 * source (SURVEY.md §8(c) last sentence of the "must follow" row). */
void ref_bodyForce_f32(const float *pos, float *vel, float dt, int n, int d2_mode, int rsqrt_mode, int sum_mode) {
  float *acc = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
  ref_forces_f32(pos, n, pos, n, NULL, acc, d2_mode, rsqrt_mode, sum_mode);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) vel[4 * i + c] = fmaf(dt, acc[4 * i + c], vel[4 * i + c]);
  free(acc);
}
/* integrate(): drift.  r += v * dt with one rounding per component. */
void ref_integrate_f32(float *pos, const float *vel, float dt, int n) {
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) pos[4 * i + c] = fmaf(vel[4 * i + c], dt, pos[4 * i + c]);
}
void ref_bodyForce_f64(const double *pos, double *vel, double dt, int n) {
  double *acc = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
  ref_forces_f64(pos, n, pos, n, acc);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) vel[4 * i + c] = fma(dt, acc[4 * i + c], vel[4 * i + c]);
  free(acc);
}
void ref_integrate_f64(double *pos, const double *vel, double dt, int n) {
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) pos[4 * i + c] = fma(vel[4 * i + c], dt, pos[4 * i + c]);
}
void ref_step_f32(float *pos, float *vel, float dt, int n, int nsteps, int d2_mode, int rsqrt_mode, int sum_mode) {
  for (int s = 0; s < nsteps; ++s) {
    ref_bodyForce_f32(pos, vel, dt, n, d2_mode, rsqrt_mode, sum_mode);
    ref_integrate_f32(pos, vel, dt, n);
  }
}
void ref_step_f64_order(double *pos, double *vel, double dt, int n, int nsteps, const ref_order_t *order) {
  double *acc = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
  for (int s = 0; s < nsteps; ++s) {
    ref_forces_f64_order(pos, n, pos, n, acc, order);
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) vel[4 * i + c] = fma(dt, acc[4 * i + c], vel[4 * i + c]);
    ref_integrate_f64(pos, vel, dt, n);
  }
  free(acc);
}

void ref_step_f32_order(float *pos, float *vel, float dt, int n, int nsteps, const ref_order_t *order) {
  float *acc = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
  for (int s = 0; s < nsteps; ++s) {
    ref_forces_f32_order(pos, n, pos, n, acc, order);
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) vel[4 * i + c] = fmaf(dt, acc[4 * i + c], vel[4 * i + c]);
    ref_integrate_f32(pos, vel, dt, n);
  }
  free(acc);
}
void ref_step_f64(double *pos, double *vel, double dt, int n, int nsteps) {
  for (int s = 0; s < nsteps; ++s) {
    ref_bodyForce_f64(pos, vel, dt, n);
    ref_integrate_f64(pos, vel, dt, n);
  }
}

void ref_ic_f32(float *pos, float *vel, int n, int first, int count, uint64_t seed) {
  nbody_ic_fill_f32(pos, vel, (size_t)n, (size_t)first, (size_t)count, seed);
}
void ref_ic_f64(double *pos, double *vel, int n, int first, int count, uint64_t seed) {
  nbody_ic_fill_f64(pos, vel, (size_t)n, (size_t)first, (size_t)count, seed);
}

int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ref_set_num_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}
