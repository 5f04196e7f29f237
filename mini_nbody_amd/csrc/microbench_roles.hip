// Do transcendental and ordinary VALU instructions of DIFFERENT waves overlap on a gfx950 SIMD?
// Measurement tool, not product code.  Three launches of the same grid (many short workgroups, so placement is
// dynamic and the chip is in steady state):
//   mode 0: every wave runs I x 64 v_rsq_f32                    (8 cycles per instruction per SIMD when alone)
//   mode 1: every wave runs 4I x 64 v_fma_f32                   (2 cycles per instruction: the same time as mode 0)
//   mode 2: even workgroups as mode 0, odd workgroups as mode 1
// If the transcendental unit and the main VALU serialise, T2 = (T0 + T1) / 2; if they run side by side, T2 -> T0 / 2.
// Build: hipcc --offload-arch=gfx950 -O3 microbench_roles.hip -o microbench_roles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void __launch_bounds__(256) roles(float* out, int iters, int mode, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f, c = seed * 0.25f;
  const bool trans = mode == 0 || (mode == 2 && (blockIdx.x & 1) == 0);
  if (trans) {
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##k));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    }
  } else {
    for (int it = 0; it < 4 * iters; ++it) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
      R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#undef X
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main(int argc, char** argv) {
  FILE* f = argc > 1 ? fopen(argv[1], "w") : stdout;
  const int iters = 60, grid = 256 * 8 * 24;
  float* out; CK(hipMalloc(&out, sizeof(float) * grid * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[3];
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(roles, dim3(grid), dim3(256), 0, 0, out, iters, mode, 1.0f);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
  fprintf(f, "# %d workgroups x 4 waves; mode 0 (all v_rsq_f32) %.3f ms, mode 1 (all v_fma_f32, 4x the instructions) %.3f ms, mode 2 (half and half) %.3f ms\n",
          grid, ms[0], ms[1], ms[2]);
  fprintf(f, "serialised would be %.3f ms, fully overlapped %.3f ms  ->  overlap fraction %.2f\n", 0.5 * (ms[0] + ms[1]),
          0.5 * (ms[0] > ms[1] ? ms[0] : ms[1]), (0.5 * (ms[0] + ms[1]) - ms[2]) / (0.5 * (ms[0] + ms[1]) - 0.5 * (ms[0] > ms[1] ? ms[0] : ms[1])));
  if (f != stdout) fclose(f);
  return 0;
}
