// How fast does the GPU start workgroups?  Times launches of a kernel that does nothing (and of one that sleeps a fixed
// number of cycles) for growing grids: the slope is the cost per workgroup of dispatch alone.
// build: hipcc --offload-arch=gfx950 -O3 -o build/dispatch_rate tools/microbench/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int SLEEPS> __global__ void __launch_bounds__(256) sleep_kernel(int* p) {
  for (int k = 0; k < SLEEPS; ++k) __builtin_amdgcn_s_sleep(16);     // 16 x 64 cycles each
  if (p && threadIdx.x == 9999) *p = 1;
}

template <typename K> static int time_it(const char* name, K kernel, int wgs, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 30; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), 0, 0, (int*)nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-28s workgroups=%6d x %3d threads  median %8.2f us  min %8.2f us\n", name, wgs, threads, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 0;
}

int main() {
  for (int threads : {64, 256})
    for (int wgs : {256, 1024, 2048, 4096, 8192, 16384, 65536}) {
      if (time_it("empty", empty_kernel, wgs, threads)) return 1;
    }
  for (int wgs : {256, 1024, 2048, 4096, 8192, 16384})
    if (time_it("sleep 10 x 1024 cycles", sleep_kernel<10>, wgs, 256)) return 1;
  for (int wgs : {256, 1024, 2048, 4096, 8192, 16384})
    if (time_it("sleep 30 x 1024 cycles", sleep_kernel<30>, wgs, 256)) return 1;
  return 0;
}
