// What does a step boundary cost INSIDE a launch, against the kernel boundary it would replace?
//
// Measurement tool, not product code (VERDICT r03 "next" item 7: a launch that spans steps for N <= 16384).  At N = 16384 a step
// takes 68.3 us against 57.0 us of pure issue time; the one lever not tried was K steps in ONE launch, an
// agent-scope arrival counter between steps in place of the kernel boundary.  Whether that can pay is decided by what the
// boundary costs in each form, which this program measures with the step's own traffic pattern and nothing else:
//   every workgroup writes its 64 rows' new words (16 B per lane of one wave, write-through `sc1` stores, drained), signals, waits
//   until ALL workgroups of the step have signalled, then reads words other workgroups wrote (as the next step's sources) and
//   checks them — a stale word is an error, counted.
// Forms:
//   graph     one kernel per step, replayed as a HIP graph of 32 steps (what nbody_step does today)
//   persist   ONE cooperative launch (hipLaunchCooperativeKernel: the grid is refused unless it is co-resident), per step an
//             agent-scope atomic add per workgroup and an `sc1` poll of the counter; the data read back with `sc1` vector loads
//             (MI355X_MICROARCH.md, inter-workgroup visibility, third table row)
//   persist+acquire   the same followed by an agent-scope acquire (buffer_inv sc1) in every workgroup — what PLAIN or SCALAR loads
//             of the new positions would need (the product loop delivers sources by s_load)
// Every wait is bounded: a poll that does not see its target within SPIN_LIMIT rounds sets an error flag and the workgroup
// goes on, so the grid always drains.
//
// Build: hipcc --offload-arch=gfx950 -O3 microbench_gridsync.hip -o microbench_gridsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ void store_sc1(u4* p, u4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u4 load_sc1(const u4* p) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned poll_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// one step's work of one workgroup: write the 64 words of its rows for `step` into `dst`, (after the boundary) read 64 words that
// OTHER workgroups wrote for `step` and compare
__device__ __forceinline__ void write_rows(u4* dst, int wg, unsigned step) {
  if (threadIdx.x < 64) {
    const unsigned i = (unsigned)wg * 64u + threadIdx.x;
    u4 v = {i, step, i ^ step, 0x5A5A5A5Au};
    store_sc1(dst + i, v);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <bool PLAIN>
__device__ __forceinline__ int check_rows(const u4* src, int wg, int nwg, unsigned step) {
  int bad = 0;
  if (threadIdx.x < 64) {
    // rows of a workgroup dealt to another XCD (the neighbour in launch order) and of one half the grid away
    for (int hop : {1, nwg / 2 + 3}) {
      const unsigned i = (unsigned)((wg + hop) % nwg) * 64u + threadIdx.x;
      u4 v;
      if constexpr (PLAIN) v = src[i]; else v = load_sc1(src + i);
      if (v.x != i || v.y != step || v.z != (i ^ step)) ++bad;
    }
  }
  return bad;
}

__global__ void __launch_bounds__(256) step_kernel(u4* buf0, u4* buf1, unsigned step, int* errors) {
  // reads what the previous launch wrote (step - 1, the other buffer), writes this step's words
  u4* dst = (step & 1) ? buf1 : buf0;
  const u4* src = (step & 1) ? buf0 : buf1;
  int bad = 0;
  if (step > 0) bad = check_rows<true>(src, blockIdx.x, gridDim.x, step - 1);
  write_rows(dst, blockIdx.x, step);
  if (bad) atomicAdd(errors, bad);
}

// the same with the arrivals sharded: workgroup w adds to shard w mod SHARDS (256 B apart); the workgroup whose add completes a
// shard's count for the step adds to the top counter, which everybody polls — same-address atomics serialise at the memory side
// (~25 ns each), so a single counter costs workgroups x 25 ns per step and the tree SHARDS x 25 ns + workgroups / SHARDS x 25 ns
constexpr int SHARDS = 64;
__global__ void __launch_bounds__(256) persist_tree_kernel(u4* buf0, u4* buf1, unsigned* counter, int steps, int* errors, int* timeouts) {
  const int nwg = gridDim.x;
  const int shard = blockIdx.x % SHARDS;
  const unsigned in_shard = (unsigned)((nwg - shard + SHARDS - 1) / SHARDS);
  const unsigned nshards = (unsigned)(nwg < SHARDS ? nwg : SHARDS);
  unsigned* top = counter + SHARDS * 64;
  int bad = 0;
  for (int s = 0; s < steps; ++s) {
    u4* dst = (s & 1) ? buf1 : buf0;
    write_rows(dst, blockIdx.x, (unsigned)s);
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned t = __hip_atomic_fetch_add(counter + shard * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == in_shard * (unsigned)(s + 1) - 1u) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = nshards * (unsigned)(s + 1);
      int spins = 0;
      while (poll_sc1(top) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { atomicAdd(timeouts, 1); break; }
      }
    }
    __syncthreads();
    bad += check_rows<false>(dst, blockIdx.x, nwg, (unsigned)s);
  }
  if (bad) atomicAdd(errors, bad);
}

template <int ACQUIRE>
__global__ void __launch_bounds__(256) persist_kernel(u4* buf0, u4* buf1, unsigned* counter, int steps, int* errors, int* timeouts) {
  const int nwg = gridDim.x;
  int bad = 0;
  for (int s = 0; s < steps; ++s) {
    u4* dst = (s & 1) ? buf1 : buf0;
    write_rows(dst, blockIdx.x, (unsigned)s);        // stores drained by the storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)nwg * (unsigned)(s + 1);
      int spins = 0;
      while (poll_sc1(counter) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT) { atomicAdd(timeouts, 1); break; }     // never wait forever: the grid must drain
      }
      if constexpr (ACQUIRE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    bad += ACQUIRE ? check_rows<true>(dst, blockIdx.x, nwg, (unsigned)s) : check_rows<false>(dst, blockIdx.x, nwg, (unsigned)s);
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s  CUs=%d  cooperativeLaunch=%d\n", prop.gcnArchName, cus, prop.cooperativeLaunch);
  const int steps = argc > 1 ? atoi(argv[1]) : 2048;
  u4 *b0, *b1;
  unsigned* counter;
  int *errors, *timeouts;
  const int max_wg = cus * 8;
  CK(hipMalloc(&b0, (size_t)max_wg * 64 * 16)); CK(hipMalloc(&b1, (size_t)max_wg * 64 * 16));
  CK(hipMalloc(&counter, (SHARDS + 1) * 256)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&timeouts, 4));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, int nwg, float ms) {
    int he = 0, ht = 0;
    CK(hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, timeouts, 4, hipMemcpyDeviceToHost));
    printf("%-34s %5d workgroups (%d per CU)  %7.3f us per step boundary + 1 KiB per workgroup written and 2 KiB read   stale words %d, timed-out waits %d\n",
           name, nwg, nwg / cus, ms * 1e3 / steps, he, ht);
  };
  for (int per_cu : {1, 2, 4, 8}) {
    const int nwg = cus * per_cu;
    // ---- graph of 32 one-step kernels
    {
      CK(hipMemset(errors, 0, 4)); CK(hipMemset(timeouts, 0, 4));
      hipGraph_t g; hipGraphExec_t ge;
      step_kernel<<<nwg, 256, 0, st>>>(b0, b1, 0u, errors);       // step 0 eagerly (nothing to read yet), then graphs of steps 1..32 pattern
      CK(hipStreamSynchronize(st));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (unsigned s = 1; s <= 32; ++s) step_kernel<<<nwg, 256, 0, st>>>(b0, b1, s, errors);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      // (a replay re-runs steps 1..32: step 1 then reads what step 32 left in the other buffer as "step 0" — the check is only
      //  meaningful for the first replay, so errors are read after ONE replay and the timing taken over the rest)
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      int he = 0; CK(hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < steps / 32; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(errors, &he, 4, hipMemcpyHostToDevice));
      report("graph of 32 one-step kernels", nwg, ms);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // ---- one cooperative launch
    for (int acq = 0; acq < 3; ++acq) {
      CK(hipMemset(errors, 0, 4)); CK(hipMemset(timeouts, 0, 4)); CK(hipMemset(counter, 0, (SHARDS + 1) * 256));
      int nsteps = steps;
      void* args[] = {&b0, &b1, &counter, &nsteps, &errors, &timeouts};
      const void* fn = acq == 2 ? (const void*)persist_tree_kernel : acq ? (const void*)persist_kernel<1> : (const void*)persist_kernel<0>;
      int occ = 0;
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, 0));
      if (nwg > occ * cus) { printf("persist: %d workgroups exceed the co-resident capacity %d: not launched\n", nwg, occ * cus); continue; }
      CK(hipEventRecord(e0, st));
      hipError_t le = hipLaunchCooperativeKernel(fn, dim3(nwg), dim3(256), args, 0, st);
      if (le != hipSuccess) { printf("cooperative launch refused: %s\n", hipGetErrorString(le)); (void)hipGetLastError(); continue; }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      report(acq == 2 ? "one launch, 64 shards + top counter" : acq ? "one launch, counter + agent acquire" : "one launch, counter, sc1 loads", nwg, ms);
    }
  }
  return 0;
}
