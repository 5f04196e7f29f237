// VALU issue-rate microbenchmarks for gfx950 (MI355X).
//
// Measurement tool, not product code: it prices the instructions of the pair
// interaction (3 sub + 3 fma + v_rsq_f32 + 2 mul + 3 fma)
// so that DESIGN.md can state the instruction-issue bound next to the 20-flop
// roofline (SURVEY.md §8(d) "Secondary, more honest bound").
//
// Each test runs a loop of ITERS x BODY independent instructions of one kind
// in every wave of 256-thread workgroups, W workgroups per CU (W waves per
// SIMD, enforced through the dynamic-LDS size), and reports (a) wall time x
// in-kernel clock / instructions per SIMD and (b) per-wave s_memtime cycles.
//
// Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

enum Test { T_FMA = 0, T_PKFMA, T_RSQ, T_MIX11_1, T_FMA_SGPR, T_READLANE, T_PKMUL, T_PKADD,
            T_MIX_PK, T_SUB_SGPR, T_MIX22_2, T_RCP, T_SQRT, T_FMA_F64, T_RSQ_F64, T_MUL_F64, T_BANK_SAME3, T_BANK_DIFF3, T_BANK_SAME2, T_FMAC_SAME, T_FMAC_DIFF, T_DEP_CHAIN, T_NTESTS };
static const char* kNames[T_NTESTS] = {
  "v_fma_f32", "v_pk_fma_f32", "v_rsq_f32", "mix 11 fma + 1 rsq", "v_fma_f32 (sgpr src)",
  "v_readlane_b32", "v_pk_mul_f32", "v_pk_add_f32", "mix 11 pk_fma + 2 rsq (2 pairs)",
  "v_sub_f32 (sgpr src)", "mix 22 fma + 2 rsq (interleaved)", "v_rcp_f32", "v_sqrt_f32",
  "v_fma_f64", "v_rsq_f64", "v_mul_f64", "v_fma_f32 3 srcs same VGPR bank", "v_fma_f32 3 srcs different banks",
  "v_fma_f32 2 of 3 srcs same bank", "v_fmac_f32 (vop2) srcs same bank", "v_fmac_f32 (vop2) srcs diff banks", "pair chain (12 dependent ops, 1 chain)" };
// instructions per BODY for cycle accounting
static const int kInstr[T_NTESTS] = { 64, 64, 64, 48, 64, 64, 64, 64, 52, 64, 96, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 48 };

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int TEST>
__global__ void __launch_bounds__(256) ubench(float* out, unsigned long long* cyc, unsigned long long* rt,
                                               int iters, float seed, float sseed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f, c = seed * 0.25f;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  float2v pb = {b, c};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = b;
  float s = __builtin_amdgcn_readfirstlane(sseed);
  int si = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == T_FMA) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_PKFMA) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##k) : "v"(pb));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_PKMUL) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##k) : "v"(pb));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_PKADD) {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##k) : "v"(pb));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_RSQ) {
#define X(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_RCP) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_SQRT) {
#define X(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_MIX11_1) {
      // the issue mix of one pair interaction: 11 full-rate + 1 transcendental
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define M1 X(0) X(1) X(2) X(3) X(4) X(5) asm volatile("v_rsq_f32 %0, %0" : "+v"(a6)); X(7) X(0) X(1) X(2) X(3)
      M1 M1 M1 M1
#undef M1
#undef X
    } else if constexpr (TEST == T_MIX22_2) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define M2 X(0) X(1) X(2) X(3) X(4) X(5) asm volatile("v_rsq_f32 %0, %0" : "+v"(a6)); X(0) X(1) X(2) X(3) X(4) X(5) \
           asm volatile("v_rsq_f32 %0, %0" : "+v"(a7)); X(0) X(1) X(2) X(3) X(4) X(5) X(0) X(1) X(2) X(3)
      M2 M2 M2 M2
#undef M2
#undef X
    } else if constexpr (TEST == T_MIX_PK) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##k) : "v"(pb));
#define M3 X(0) X(1) X(2) X(3) X(4) X(5) asm volatile("v_rsq_f32 %0, %0" : "+v"(a6)); \
           asm volatile("v_rsq_f32 %0, %0" : "+v"(a7)); X(6) X(7) X(0) X(1) X(2)
      M3 M3 M3 M3
#undef M3
#undef X
    } else if constexpr (TEST == T_FMA_SGPR) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "s"(s), "v"(c));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_SUB_SGPR) {
#define X(k) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a##k) : "s"(s));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_READLANE) {
#define X(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(si) : "v"(a##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_FMA_F64) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##k) : "v"(db));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_MUL_F64) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##k) : "v"(db));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    } else if constexpr (TEST == T_BANK_SAME3) {
      // all three sources (and the destination) in VGPR bank 0 (register number mod 4)
#define B8 "v_fma_f32 v32, v36, v40, v32\n v_fma_f32 v44, v36, v40, v44\n v_fma_f32 v48, v36, v40, v48\n v_fma_f32 v52, v36, v40, v52\n" \
           "v_fma_f32 v56, v36, v40, v56\n v_fma_f32 v60, v36, v40, v60\n v_fma_f32 v64, v36, v40, v64\n v_fma_f32 v68, v36, v40, v68\n"
      asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 ::: "v32", "v36", "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
#undef B8
    } else if constexpr (TEST == T_BANK_DIFF3) {
#define B8 "v_fma_f32 v32, v37, v42, v32\n v_fma_f32 v44, v37, v42, v44\n v_fma_f32 v48, v37, v42, v48\n v_fma_f32 v52, v37, v42, v52\n" \
           "v_fma_f32 v56, v37, v42, v56\n v_fma_f32 v60, v37, v42, v60\n v_fma_f32 v64, v37, v42, v64\n v_fma_f32 v68, v37, v42, v68\n"
      asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 ::: "v32", "v37", "v42", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
#undef B8
    } else if constexpr (TEST == T_BANK_SAME2) {
#define B8 "v_fma_f32 v32, v36, v41, v32\n v_fma_f32 v44, v36, v41, v44\n v_fma_f32 v48, v36, v41, v48\n v_fma_f32 v52, v36, v41, v52\n" \
           "v_fma_f32 v56, v36, v41, v56\n v_fma_f32 v60, v36, v41, v60\n v_fma_f32 v64, v36, v41, v64\n v_fma_f32 v68, v36, v41, v68\n"
      asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 ::: "v32", "v36", "v41", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
#undef B8
    } else if constexpr (TEST == T_FMAC_SAME) {
#define B8 "v_fmac_f32 v32, v36, v40\n v_fmac_f32 v44, v36, v40\n v_fmac_f32 v48, v36, v40\n v_fmac_f32 v52, v36, v40\n" \
           "v_fmac_f32 v56, v36, v40\n v_fmac_f32 v60, v36, v40\n v_fmac_f32 v64, v36, v40\n v_fmac_f32 v68, v36, v40\n"
      asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 ::: "v32", "v36", "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
#undef B8
    } else if constexpr (TEST == T_FMAC_DIFF) {
#define B8 "v_fmac_f32 v32, v37, v42\n v_fmac_f32 v44, v37, v42\n v_fmac_f32 v48, v37, v42\n v_fmac_f32 v52, v37, v42\n" \
           "v_fmac_f32 v56, v37, v42\n v_fmac_f32 v60, v37, v42\n v_fmac_f32 v64, v37, v42\n v_fmac_f32 v68, v37, v42\n"
      asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 ::: "v32", "v37", "v42", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
#undef B8
    } else if constexpr (TEST == T_DEP_CHAIN) {
      // one pair interaction as the real kernel issues it for ONE body per lane: a serial dependent chain
#define P1 "v_sub_f32 v33, %0, v37\n v_sub_f32 v34, %0, v38\n v_sub_f32 v35, %0, v39\n v_fma_f32 v36, v35, v35, v41\n v_fmac_f32 v36, v34, v34\n v_fmac_f32 v36, v33, v33\n" \
           "v_rsq_f32 v36, v36\n v_mul_f32 v42, v36, v36\n v_mul_f32 v36, v36, v42\n v_fmac_f32 v43, v33, v36\n v_fmac_f32 v44, v34, v36\n v_fmac_f32 v45, v35, v36\n"
      asm volatile(P1 P1 P1 P1 :: "s"(s) : "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v41", "v42", "v43", "v44", "v45");
#undef P1
    } else if constexpr (TEST == T_RSQ_F64) {
#define X(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(d##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    }
  }
  // wait for the last VALU results before the closing stamp
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)si
            + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cyc[w] = t1 - t0;
    rt[w] = r1 - r0;
  }
}

template <int TEST>
static void run(FILE* f, int ncu, int iters) {
  // 256-thread workgroups (one wave per SIMD each); W workgroups per CU are made
  // co-resident by giving each 160 KiB / W of dynamic LDS.  ROUNDS x (ncu x W)
  // workgroups, so wall time / ROUNDS is the steady-state time of one full chip.
  const int ws[] = {1, 2, 3, 4, 8};
  const int ROUNDS = 3;
  for (int wi = 0; wi < 5; ++wi) {
    int W = ws[wi];
    int threads = 256, grid = ncu * W * ROUNDS;
    size_t lds = (160 * 1024) / W - (W == 1 ? 0 : 512);
    int nwaves = grid * threads / 64;
    float* out; unsigned long long *cyc, *rt;
    CK(hipMalloc(&out, sizeof(float) * grid * threads));
    CK(hipMalloc(&cyc, sizeof(unsigned long long) * nwaves));
    CK(hipMalloc(&rt, sizeof(unsigned long long) * nwaves));
    CK(hipFuncSetAttribute((const void*)ubench<TEST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(ubench<TEST>, dim3(grid), dim3(threads), lds, 0, out, cyc, rt, iters, 1.0f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(ubench<TEST>, dim3(grid), dim3(threads), lds, 0, out, cyc, rt, iters, 1.0f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwaves), hr(nwaves);
    CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nwaves, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, sizeof(unsigned long long) * nwaves, hipMemcpyDeviceToHost));
    std::vector<double> ghz(nwaves);
    for (int i = 0; i < nwaves; ++i) ghz[i] = hr[i] ? (double)h[i] / (double)hr[i] * 0.1 : 0.0;
    std::sort(h.begin(), h.end()); std::sort(ghz.begin(), ghz.end());
    double instr = (double)iters * kInstr[TEST];
    double clk = ghz[nwaves / 2];
    // per SIMD: W waves per round, ROUNDS rounds
    double wall_cyc = (ms * 1e-3) * clk * 1e9 / (instr * W * ROUNDS);
    fprintf(f, "%-34s W=%d  wall cyc/instr/SIMD=%6.3f  | per-wave cyc/instr min/med/max=%6.2f/%6.2f/%6.2f (x1/W: %5.2f) clk=%.3f GHz kernel=%.3f ms\n",
            kNames[TEST], W, wall_cyc, h[0] / instr, h[nwaves / 2] / instr, h[nwaves - 1] / instr,
            h[nwaves / 2] / instr / W, clk, ms);
    fflush(f);
    CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(rt));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
}

int main(int argc, char** argv) {
  FILE* f = stdout;
  if (argc > 1) { f = fopen(argv[1], "w"); if (!f) { perror("fopen"); return 1; } }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  fprintf(f, "# device %s  CUs=%d  clockRate=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount, iters = 4000;
  if (!getenv("NB_ONLY_NEW")) {
  run<T_FMA>(f, ncu, iters);      run<T_PKFMA>(f, ncu, iters);   run<T_PKMUL>(f, ncu, iters);
  run<T_PKADD>(f, ncu, iters);    run<T_RSQ>(f, ncu, iters);     run<T_RCP>(f, ncu, iters);
  run<T_SQRT>(f, ncu, iters);     run<T_MIX11_1>(f, ncu, iters); run<T_MIX22_2>(f, ncu, iters);
  run<T_MIX_PK>(f, ncu, iters);   run<T_FMA_SGPR>(f, ncu, iters); run<T_SUB_SGPR>(f, ncu, iters);
  run<T_READLANE>(f, ncu, iters); run<T_FMA_F64>(f, ncu, iters); run<T_MUL_F64>(f, ncu, iters);
  run<T_RSQ_F64>(f, ncu, iters);
  }
  run<T_BANK_SAME3>(f, ncu, iters); run<T_BANK_DIFF3>(f, ncu, iters); run<T_BANK_SAME2>(f, ncu, iters);
  run<T_FMAC_SAME>(f, ncu, iters); run<T_FMAC_DIFF>(f, ncu, iters); run<T_DEP_CHAIN>(f, ncu, iters);
  if (f != stdout) fclose(f);
  return 0;
}
