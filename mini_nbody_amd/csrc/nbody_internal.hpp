// nbody_internal.hpp — what the four parts of libnbody_hip.so share (none of it crosses the C-ABI of include/nbody.h):
//   kernels.hip   the nbk kernels (nbody_kernels.hpp) and the thin launch functions that pick an instantiation (namespace nbl)
//   context.cpp   the context: options, launch configuration, buffers, the step and its HIP graph, state transfer, the strict gate
//   comm.cpp      RCCL (resolved with dlopen), the transfer plans, the all-gather of a step, probes and self-tests
//   mailbox.cpp   the reference's mailbox: RAM images, one request, the service thread
// Only kernels.hip is device code (a minute of hipcc); the other three are host C++ (seconds).  gfx950 only, no CPU fallback anywhere.
//
// Data layout in HBM (per rank; N bodies in total, the rank owns n_local of them starting at first_body):
//   pos[2]   2 x N words      full position set, double-buffered: a step reads pos[cur] and writes the
//                             rank's slice of pos[cur^1]; the other slices of pos[cur^1] arrive over xGMI
//   vel      n_local words    never leaves the rank
//   partial  nseg x rows'     per-source-segment partial forces (unused when nseg == 1); rows' = the launch's rows rounded up to 64
//   tickets  1 per 64 rows    arrival counters of the in-launch combine (zero between steps)
//   force    n_local words    last combined forces (parity entry points)
// word = {x,y,z,w}: 16 B (fp32) or 32 B (fp64) — the reference's RAM word, S/top_level.vhd:206-208.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only; the library is resolved with dlopen when nranks > 1
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/nbody.h"
#pragma GCC visibility pop
#include "nbody_args.hpp"

namespace nbi {

using nbk::ForceArgs;

// where the last HIP / RCCL error was seen (nbody_error_string); atomics: the mailbox's service thread may be the one that saw it
extern std::atomic<const char*> g_last_file;
extern std::atomic<int> g_last_line;
#define NB_MARK() do { ::nbi::g_last_file.store(__FILE__, std::memory_order_relaxed); ::nbi::g_last_line.store(__LINE__, std::memory_order_relaxed); } while (0)
// While nbody_mailbox_serve(1, .) is in effect the service thread owns the context: every entry point that launches, copies or
// reconfigures answers NBODY_ERR_STATE (nbody_get_info, nbody_error_string, nbody_mailbox_rams, nbody_mailbox_serve and nbody_shutdown do not)
#define NB_REFUSE_WHILE_SERVED() do { if (::nbi::mailbox_serving()) return NBODY_ERR_STATE; } while (0)
#define HIPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { NB_MARK(); return (int)e_; } } while (0)
#define NBC(expr) do { int e_ = (expr); if (e_ != NBODY_OK) return e_; } while (0)
#define NCCLC(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { NB_MARK(); return 2000 + (int)r_; } } while (0)

constexpr int kMaxLocal = 16;
constexpr int kMaxRanks = 64;
constexpr int kTimerRing = 256;
constexpr int kGraphSteps = 32;    // steps per replayed HIP graph once a call brings at least twice as many

// a ring of HIP event pairs whose durations are summed lazily (no host sync while a step is being enqueued)
struct EventTimer {
  hipEvent_t t0[kTimerRing] = {}, t1[kTimerRing] = {};
  int head = 0, count = 0;
  double ms = 0.0;
  long long n = 0;
};

struct Local {
  int device = 0, rank = 0;
  int first = 0, n_local = 0;          // owned bodies
  hipStream_t compute = nullptr, comm = nullptr;
  void* pos[2] = {nullptr, nullptr};
  void* vel = nullptr;
  void* partial = nullptr;
  size_t partial_words = 0;            // capacity of `partial`
  unsigned* tickets = nullptr;         // arrival counters: one per wave of every block of 256 rows
  void* force = nullptr;
  void* force_dst = nullptr;           // where a launch stores {Fx,Fy,Fz,0} instead of `force` (a mailbox request: RAM B itself)
  const void* src_direct = nullptr;    // a mailbox request of a handful of bodies: sources and rows read from RAM A itself (no ingest launch) ...
  unsigned long long* t0_stamp = nullptr;   // ... and the launch's first wave stamps the tick count's start here (ForceArgs::t0_stamp)
  void* full_scratch = nullptr;        // N words: all-gather of a sharded array for the host (multi-process)
  int cur = 0;
  bool all_present = true;             // pos[cur] holds every slice
  hipEvent_t ev_own_ready = nullptr;   // the rank's slice of pos[cur] is written
  hipEvent_t ev_comm_go = nullptr;     // the transfer stream has seen ev_own_ready: its RCCL kernel is next on its queue
  hipEvent_t ev_gather[kMaxRanks] = {};
  ncclComm_t comm_h = nullptr;
  EventTimer kern;   // force kernels (NBODY_OPT_TIMING)
  EventTimer wait;   // how long the compute stream sat waiting for arriving slices: exposed communication
};

struct Options {
  int variant = NBODY_VARIANT_AUTO, iblock = 0, jsub = 0, jslices = 0;
  int arith = NBODY_ARITH_FMA3, sum_order = NBODY_SUM_BLOCKED, sum_block = 1024, fuse = -1;
  int timing = 0, comm = NBODY_COMM_AUTO, overlap = 1, isa_phase = 1, waves_per_simd = 0, graph = 1, long_buffers = -1, xcd_map = -1;
  int wsplit = -1;
};

// what happens to the force of a row once all its segments are summed
struct Finish { bool kick, drift, store_force; };

typedef int (*host_gather_fn)(void* user, void* host_words, int n_total, int word_bytes, int rank, int nranks);

struct Global {
  host_gather_fn host_gather = nullptr;   // multi-process fallback transport: slices exchanged through host memory
  void* host_gather_user = nullptr;
  void* host_stage = nullptr;             // pinned staging buffer, N words
  // HIP graph of an even number of consecutive steps (the position buffers swap every step, so a pair returns to the same state):
  // replayed by nbody_step when one GPU runs many short steps (launch-bound regime)
  hipGraphExec_t step_graph = nullptr;
  bool stepped_eagerly = false;   // a step has been launched outside a capture since nbody_init
  float graph_dt = 0.f; double graph_dt64 = 0.0; int graph_cur = -1, graph_len = 0;
  bool init = false;
  int n = 0, fp64 = 0, tile = 256;
  int cap = 0;                    // body words the buffers were allocated for (= the n of nbody_init; a mailbox request may bring fewer)
  int nranks = 1, nlocal = 0;
  bool multiprocess = false;
  Local loc[kMaxLocal];
  Options opt;
  // resolved launch configuration
  int variant = NBODY_VARIANT_SMEM, R = 4, sub = 1, nslices = 1, nseg = 1, fuse = 1;
  int wsplit = 1;                 // 4: a workgroup owns 64 rows, its four waves walk a quarter of the segment each (ForceArgs::wsplit)
  bool comm_go_armed = false;     // the gather just enqueued recorded ev_comm_go (RCCL transport)
  bool tickets_dirty = false;     // a step failed after some of its launches: the arrival counters may be non-zero
  int cu_count = 0, clock_khz = 0;
  int comm_priority = 0;          // HIP priority of the transfer streams (0 = default)
  long long steps_done = 0;
  // The CONTEXT's N and resolved configuration as nbody_get_info reports them: published by reconfigure(), never touched by a mailbox
  // request (which switches the fields above for its own duration) — so the caller's thread may read them while the service thread works
  struct View { int n = 0, n_local = 0, variant = 0, R = 0, sub = 0, nseg = 1, fuse = 1, wsplit = 1; } view;
};
extern Global g;

inline size_t word_bytes() { return g.fp64 ? 32 : 16; }
inline char* word_ptr(void* base, size_t word) { return (char*)base + word * word_bytes(); }
inline int ring_slice(int rank, int s) { int q = (rank - s) % g.nranks; return q < 0 ? q + g.nranks : q; }
// arrival counters: one per 64 rows (a wave's rows), with slack for the row blocks of 256*R rows whose waves count in
// strides of 4*R, padded to a multiple of 256 bytes
inline size_t ticket_words(int n_local) { return ((size_t)(n_local + 63) / 64 + 32 + 63) / 64 * 64; }

// ---- context.cpp ----
void resolve_config();
int reconfigure();
int ensure_partial(Local& L);
void drop_step_graph();
int timer_begin(EventTimer& T, hipStream_t stream, int* slot);
int timer_end(EventTimer& T, hipStream_t stream, int slot);
int timer_drain(EventTimer& T, int keep);
// the force kernel of local L for rows [row0, row0 + row_count) against `nsl` source slices starting at slice_start and descending
int launch_force(Local& L, int row0, int row_count, int slice_start, int nsl, const Finish& fin, float dt, double dt64);
int launch_combine(Local& L, int row0, int row_count, const Finish& fin, float dt, double dt64);
// launch_force would take the 16-row FPGA kernel (force_fpga16r_f32) for a launch of row_count rows in the configuration as it stands
bool takes_rows16(int row_count);
int sync_all();
int complete_positions();
int forces_impl(const void* pos_words, void* force_words, int n);
int device_count(int* ndev);
int pick_device(int rank, int ndev);

// ---- comm.cpp ----
int rccl_load();
int comm_create(Local& L, int nranks, int rank, const void* uid128);   // ncclCommInitRank on L.device
void comm_destroy(Local& L);
int resolved_comm_form();
int enqueue_gather(int buf);                                    // bring the other ranks' slices of pos[buf] to every local
int gather_sharded_multiprocess(Local& L, const void* own_rows);   // a rank-sharded array into L.full_scratch

// ---- mailbox.cpp ----
void mailbox_shutdown();          // stops the service thread, frees the RAM images
bool mailbox_serving();           // nbody_mailbox_serve(1, .) is in effect
long long mailbox_served();       // requests the service thread has completed in this process

}  // namespace nbi

// ---- kernels.hip: every kernel launch of the library ----
namespace nbl {
// which instantiation of the force kernel a launch takes (the rest travels in ForceArgs: wsplit, fpga16, long_buffers)
struct KernelSel {
  int fp64, variant, R, arith, tile, isa_phase;
  int fpga_lds;     // the FPGA order on sixteen waves: 1 = sources staged through LDS (default), 0 = scalar delivery (NBODY_VARIANT_SMEM asked for)
  int fpga_rows16;  // the FPGA order, one segment, a small launch: 16 rows x 16 chains per 256-thread workgroup (grid.x = 16-row units)
  size_t dyn_lds;   // dynamic LDS per workgroup: the occupancy cap of NBODY_OPT_WAVES_PER_SIMD (0 = none)
};
// all return a hipError_t as int (0 = launched)
int launch_force_kernel(const KernelSel& k, hipStream_t stream, dim3 grid, const nbk::ForceArgs& a);
int launch_combine_kernel(int fp64, hipStream_t stream, dim3 grid, const nbk::ForceArgs& a);
int launch_drift_kernel(int fp64, hipStream_t stream, void* pos_rows, const void* vel, int n_rows, float dt, double dt64);
int launch_ingest_kernel(hipStream_t stream, void* dst_words, const void* ram_a_bodies, int n, unsigned long long* t0);
int launch_mailbox_done_kernel(hipStream_t stream, void* word0, unsigned* seq_word, const unsigned long long* t0, unsigned seq,
                               unsigned clock_khz, unsigned rt_khz);
int launch_rsqrt_selftest_kernel(unsigned first_bits, unsigned long long count, unsigned long long* out3);
int launch_rsqrt_array_kernel(const float* x, float* y, int n, int ieee_only);
bool diag_build();   // this is libnbody_hip_diag.so (-DNBODY_DIAG_LOOPS): experiment encodings and timing-only loop forms present
}  // namespace nbl
