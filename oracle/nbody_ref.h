/* nbody_ref.h — CPU ORACLE for the all-pairs softened-gravity force path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link, load or call anything in oracle/.
 * The product path (mini_nbody_amd/, include/nbody.h) never routes through it.
 *
 * What it restates: the arithmetic of the reference's FPGA force pipeline
 * (/root/reference/vec_add.srcs/sources_1/new/ (*.vhd), "S/" below), rounding
 * point by rounding point.  The reference cannot be compiled or simulated in
 * this environment (VHDL-2008 + seven absent Xilinx Floating-Point Operator IP
 * cores), so there is no oracle/_ref build.
 *
 * PARITY PINNING: the reference's own testbenches assert no numerical values
 * (T/tb_dxy.vhd:907-918, T/tb_sqrt.vhd:562-573 check only "not X when valid").
 * Their STIMULI have analytic expected outputs; those are committed as
 * tests/golden/kat_*.json and this oracle is checked against every one of
 * them (tests/test_oracle_kat.py).  Whole-pipeline outputs (forces) are pinned
 * by no reference fixture: at that level parity is "unpinned" in the sense of
 * the task statement, and is anchored on (i) the stage KATs above, (ii) an
 * independent fp64 evaluation, (iii) a pure-Python/numpy restatement
 * (tests/test_oracle_vs_numpy.py), (iv) whole-pipeline known answers computed
 * by a third, exact-rational Python statement of the arithmetic and of the
 * summation orders (tests/golden/make_system.py -> system_*.json), which this
 * oracle and the HIP engine (strict arithmetic) both reproduce bit for bit
 * (tests/test_golden_system.py).
 */
#ifndef NBODY_REF_H
#define NBODY_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* eps = real_to_flt(1.0E-9) = 0x3089705F            S/dzsoft.vhd:177 */
#define REF_SOFT_BITS 0x3089705Fu

/* how dist^2 is rounded */
enum { REF_D2_REFERENCE = 0, /* (rn(dx*dx)+rn(dy*dy)) + fma(dz,dz,eps): S/dxy.vhd:113-122, S/dzsoft.vhd:201-202, S/dxyz_soft.vhd:149-150 */
       REF_D2_FMA3 = 1       /* fma(dx,dx,fma(dy,dy,fma(dz,dz,eps))): the GPU kernel's contraction, SURVEY.md §8(a) a6 */ };
/* how 1/sqrt is rounded (the IP's rounding is unpinned) */
enum { REF_RSQRT_F64 = 0,     /* (float)(1.0/sqrt((double)d2)): one rounding from an fp64 evaluation */
       REF_RSQRT_DIVSQRT = 1  /* 1.0f/sqrtf(d2): two fp32 roundings (what a plain C nbody does) */ };
/* in which order the per-source terms are summed */
enum { REF_SUM_SEQ = 0,      /* one accumulator, sources in ascending order (S/top_level.vhd:233-254) */
       REF_SUM_FPGA16 = 1,   /* 16 interleaved partials + pairwise tree (S/fxyz.vhd:129-184, S/final_adder.vhd:88-104) */
       REF_SUM_BLOCKED = 2   /* two levels: blocks of `block` consecutive sources are summed sequentially from zero and the
                                block sums are added, in ascending order, into a second accumulator.  The reference never
                                runs one long sequential sum either: it keeps 16 partial sums and adds them with a tree
                                (S/fxyz.vhd:129-145, S/final_adder.vhd:88-104); this is the same remedy shaped for a SIMT
                                lane (one fold per `block` sources instead of 16 live partials) */ };

/* The engine's full summation order (mini_nbody_amd/csrc/nbody_kernels.hpp): the sources are cut into `nslices` balanced
 * slices (one per rank) of `sub` pieces each; every segment is summed on its own from zero (sum_mode, `block` sources per
 * block when sum_mode == REF_SUM_BLOCKED, counted from the segment's first source), and the segment sums are added in
 * ascending source order: F = ((p_0 + p_1) + p_2) + ...  nslices = sub = wsplit = 1 and REF_SUM_SEQ is the plain sequential sum. */
typedef struct {
  int d2_mode, rsqrt_mode;
  int sum_mode;      /* REF_SUM_SEQ or REF_SUM_BLOCKED */
  int block;         /* sources per block (REF_SUM_BLOCKED) */
  int nslices, sub;  /* segmentation of the sources */
  int wsplit;        /* pieces per segment (1, or 4 = the four waves of a workgroup, NBODY_OPT_WSPLIT): each piece is summed on
                        its own as above and the piece sums are added in ascending order to give the segment's sum; 0 = 1 */
} ref_order_t;

/* ---- pipeline stages, one function per reference entity ---- */
float ref_soft(void);
/* S/dxy.vhd:94-122   returns rn(rn(dx*dx)+rn(dy*dy)); dx = x_target - x_this */
float ref_dxy(float x_this, float x_target, float y_this, float y_target, float *dx, float *dy);
/* S/dzsoft.vhd:186-202   returns fma(dz,dz,eps) */
float ref_dzsoft(float z_this, float z_target, float *dz);
/* S/dxyz_soft.vhd:87-93,149-150   returns dist_sqr */
float ref_dxyz_soft(float x_this, float x_target, float y_this, float y_target, float z_this, float z_target,
                    float *dx, float *dy, float *dz);
/* contraction used on the GPU */
float ref_d2_fma3(float dx, float dy, float dz);
/* S/fxyz.vhd:101-102 (IP rsqrt) */
float ref_rsqrt(float d2, int rsqrt_mode);
/* S/cube.vhd:66-70   inv * (inv * inv) */
float ref_cube(float inv);
/* S/final_adder.vhd:88-104   pairwise tree over 16 leaves */
float ref_tree16(const float p[16]);

/* ---- whole passes ----
 * rows:  the "this" bodies, n_rows x 4 floats {x,y,z,.}   (S/top_level.vhd:206-208)
 * src:   the "target" bodies streamed past every row, n_src x 4 floats (S/top_level.vhd:238-240)
 * acc:   n_rows x 4 floats {Fx,Fy,Fz,0}                   (S/compute_store.vhd:213,242)
 * acc_in (may be NULL): starting value of the accumulators (sequential mode only) */
void ref_forces_f32(const float *rows, int n_rows, const float *src, int n_src, const float *acc_in, float *acc,
                    int d2_mode, int rsqrt_mode, int sum_mode);
/* forces in the engine's order (segments, blocks): rows vs ALL n_src sources */
void ref_forces_f32_order(const float *rows, int n_rows, const float *src, int n_src, float *acc, const ref_order_t *order);
/* piece w of ws of a segment */
void ref_piece_bounds(int jb, int je, int w, int ws, int *pb, int *pe);
/* segment (q, t) of the order: [*jb, *je) */
void ref_segment_bounds(int q, int t, int n, int nslices, int sub, int *jb, int *je);
/* fp64 arithmetic on fp64 inputs (the fp64 config's oracle) */
void ref_forces_f64(const double *rows, int n_rows, const double *src, int n_src, double *acc);
/* fp64 arithmetic on fp32 inputs, fp64 outputs: the arbiter between fp32 orders */
void ref_forces_f64_from_f32(const float *rows, int n_rows, const float *src, int n_src, double *acc);

/* bodyForce(): v_i = fma(dt, F_i, v_i); integrate(): r_i = fma(v_i, dt, r_i).
 * Neither exists in the reference; they follow the north_star text. */
void ref_bodyForce_f32(const float *pos, float *vel, float dt, int n, int d2_mode, int rsqrt_mode, int sum_mode);
void ref_integrate_f32(float *pos, const float *vel, float dt, int n);
void ref_bodyForce_f64(const double *pos, double *vel, double dt, int n);
void ref_integrate_f64(double *pos, const double *vel, double dt, int n);
/* nsteps x { bodyForce; integrate } */
void ref_step_f32(float *pos, float *vel, float dt, int n, int nsteps, int d2_mode, int rsqrt_mode, int sum_mode);
void ref_step_f64(double *pos, double *vel, double dt, int n, int nsteps);
/* fp64 in the engine's summation order (segments x pieces of the wave split, one sequential sum per piece): what an fp64 context in
 * NBODY_ARITH_STRICT reproduces bit for bit */
void ref_forces_f64_order(const double *rows, int n_rows, const double *src, int n_src, double *acc, const ref_order_t *order);
void ref_step_f64_order(double *pos, double *vel, double dt, int n, int nsteps, const ref_order_t *order);
/* the same loop with the forces summed in the engine's order */
void ref_step_f32_order(float *pos, float *vel, float dt, int n, int nsteps, const ref_order_t *order);

/* deterministic initial conditions (include/nbody_ic.h) */
void ref_ic_f32(float *pos, float *vel, int n, int first, int count, uint64_t seed);
void ref_ic_f64(double *pos, double *vel, int n, int first, int count, uint64_t seed);

int ref_num_threads(void);
void ref_set_num_threads(int t);

#ifdef __cplusplus
}
#endif
#endif
