// nbody_hip.hip — the C-ABI of include/nbody.h on top of the kernels in
// nbody_kernels.hpp.  gfx950 only; no CPU fallback anywhere in this file.
//
// Data layout in HBM (per rank; N bodies in total, the rank owns n_local of
// them starting at first_body):
//   pos[2]   2 x N words      full position set, double-buffered: a step reads pos[cur] and writes the
//                             rank's slice of pos[cur^1]; the other slices of pos[cur^1] arrive over xGMI
//   vel      n_local words    never leaves the rank
//   partial  nseg x n_local   per-source-segment partial forces (unused when nseg == 1)
//   tickets  4 per 256 rows   arrival counters (one per wave) of the in-launch combine (zero between steps)
//   force    n_local words    last combined forces (mailbox / parity entry points)
// word = {x,y,z,w}: 16 B (fp32) or 32 B (fp64) — the reference's RAM word, S/top_level.vhd:206-208.
//
// Multi-GPU: bodies are sharded by i; every step each rank
// needs all N positions.  Sources are cut into one slice per rank (x `sub`
// pieces); a step first runs on the rank's own slice while the other slices
// travel (ring of ncclSend/ncclRecv on a second stream, or peer copies when one
// process drives all GPUs), then on the arrived slices.  Partial sums are kept
// per segment and combined in ascending source order — by the last workgroup to
// arrive for a block of rows, inside the force launch (finish_rows in
// nbody_kernels.hpp) — so the result is bit-identical for every arrival order
// and for a single GPU configured with the same segmentation.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only; the library is resolved with dlopen when nranks > 1
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/nbody.h"
#include "nbody_kernels.hpp"

using namespace nbk;

namespace {

#define HIPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_line = __LINE__; return (int)e_; } } while (0)
#define NBC(expr) do { int e_ = (expr); if (e_ != NBODY_OK) return e_; } while (0)
#define NCCLC(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { g_last_line = __LINE__; return 2000 + (int)r_; } } while (0)

int g_last_line = 0;

constexpr int kMaxLocal = 16;
constexpr int kMaxRanks = 64;
constexpr int kTimerRing = 256;

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.handle) return NBODY_OK;
  // librccl.so.1 already mapped by the host framework (e.g. torch) is reused: same SONAME.
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) return NBODY_ERR_RCCL_LOAD;
#define SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return NBODY_ERR_RCCL_LOAD; } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllGather, "ncclAllGather");
#undef SYM
  g_rccl.handle = h;
  return NBODY_OK;
}

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
  void* full_scratch = nullptr;        // N words: all-gather of a sharded array for the host (multi-process)
  int cur = 0;
  bool all_present = true;             // pos[cur] holds every slice
  hipEvent_t ev_own_ready = nullptr;   // the rank's slice of pos[cur] is written
  hipEvent_t ev_gather[kMaxRanks] = {};
  ncclComm_t comm_h = nullptr;
  // force-kernel timing
  hipEvent_t t0[kTimerRing] = {}, t1[kTimerRing] = {};
  int t_head = 0, t_count = 0;
  double t_ms = 0.0;
  long long t_launches = 0;
};

struct Options {
  int variant = NBODY_VARIANT_AUTO, iblock = 0, jsub = 0, jslices = 0;
  int arith = NBODY_ARITH_FMA3, sum_order = NBODY_SUM_BLOCKED, sum_block = 1024, fuse = -1;
  int timing = 0, comm = NBODY_COMM_AUTO, overlap = 1, isa_phase = 1, waves_per_simd = 0, graph = 1, long_buffers = -1, xcd_map = -1;
};

// what happens to the force of a row once all its segments are summed
struct Finish { bool kick, drift, store_force; };

typedef int (*host_gather_fn)(void* user, void* host_words, int n_total, int word_bytes, int rank, int nranks);

struct Global {
  host_gather_fn host_gather = nullptr;   // multi-process fallback transport: slices exchanged through host memory
  void* host_gather_user = nullptr;
  void* host_stage = nullptr;             // pinned staging buffer, N words
  // HIP graph of TWO consecutive steps (the position buffers swap every step, so a pair returns to the same state):
  // replayed by nbody_step when one GPU runs many short steps (launch-bound regime)
  hipGraphExec_t step_graph = nullptr;
  bool stepped_eagerly = false;   // a step has been launched outside a capture since nbody_init
  float graph_dt = 0.f; double graph_dt64 = 0.0; int graph_cur = -1;
  bool init = false;
  int n = 0, fp64 = 0, tile = 256;
  int nranks = 1, nlocal = 0;
  bool multiprocess = false;
  Local loc[kMaxLocal];
  Options opt;
  // resolved launch configuration
  int variant = NBODY_VARIANT_SMEM, R = 4, sub = 1, nslices = 1, nseg = 1, fuse = 1;
  int cu_count = 0, clock_khz = 0;
  long long steps_done = 0;
};
Global g;

inline size_t word_bytes() { return g.fp64 ? 32 : 16; }
inline char* word_ptr(void* base, size_t word) { return (char*)base + word * word_bytes(); }

int blocks_for(int rows, int R) { return (rows + kBlock * R - 1) / (kBlock * R); }

// Choose R (bodies per lane) and sub (pieces per source slice).  Measured at N = 1M on MI355X (profiles/r01_sweep.txt):
// one body per lane (16 waves per SIMD worth of work, 8 resident) beats 2/4/8 bodies per lane — hipcc software-pipelines
// the 8 sources of a scalar-load group across the single chain, and more resident waves hide the transcendental — and
// cutting the sources into pieces so that a launch has >= 16k workgroups adds ~8 % (load balance across the 256 CUs).
void resolve_config() {
  // the largest slice (ceil(N / P)): every rank of a multi-process job resolves the same segmentation from it
  const int n_local = std::max(1, (g.n + g.nranks - 1) / g.nranks);
  g.nslices = g.nranks > 1 ? g.nranks : (g.opt.jslices > 0 ? g.opt.jslices : 1);
  // AUTO: the hand-scheduled ISA loop (+6 % over hipcc's schedule of the same operations, profiles/r01_sweep_isa.txt)
  g.variant = g.opt.variant == NBODY_VARIANT_AUTO ? NBODY_VARIANT_ISA : g.opt.variant;
  if (g.fp64 && g.variant != NBODY_VARIANT_ISA) g.variant = NBODY_VARIANT_SMEM;   // fp64: ISA loop or the compiled SMEM kernel
  // the hand-scheduled loops exist for the timed arithmetic only; the study modes use the C++ kernels
  if (g.variant == NBODY_VARIANT_ISA && (g.opt.arith != NBODY_ARITH_FMA3 || g.opt.sum_order == NBODY_SUM_FPGA16)) g.variant = NBODY_VARIANT_SMEM;
  int R = g.opt.iblock;
  if (R == 0) R = (g.variant == NBODY_VARIANT_LDS || g.variant == NBODY_VARIANT_READLANE) ? 2 : 1;
  if (g.variant == NBODY_VARIANT_ISA) R = 1;
  if (g.fp64 && R > 4) R = 4;
  if (g.opt.sum_order == NBODY_SUM_FPGA16 && !g.fp64) R = 1;
  g.R = R;
  // Small launches and large ones want different things (profiles/r02_small_n.md, one process, wall clock per step):
  //   large (even 64 segments give >= 16 workgroups per CU; N >= 16384 on one GPU): many short segments for load
  //     balance over the 256 CUs — 128 workgroups per CU in the launch, up to 64 segments of >= 128 sources (N = 65536:
  //     64 segments 4547 G/s, 8: 3981; N = 1M: 8 segments 4688, 4: 4660) — and the partial sums added inside the launch
  //     by the last wave to arrive (one launch per step): its store drain and atomic round trip hide behind other
  //     workgroups.
  //   small: the step is latency, not issue: ~2 workgroups per CU (N = 4096: 32 segments 16.0 us per step, 16: 18.4,
  //     64: 19.4; N = 8192: 16 segments 27.2, 64: 31.8) and the sums added by a second small kernel — in one launch the
  //     hand-off is exposed (N = 4096: 22.7 us, N = 8192: 36.4).
  const int cus = g.cu_count > 0 ? g.cu_count : 256;
  const int blocks = blocks_for(n_local, R);
  const bool small = (long long)blocks * 64 < 16LL * cus;
  int sub = g.opt.jsub;
  if (sub == 0) {
    // workgroups per launch-slice: the step's launches together have 128 (2) per CU whatever the rank count, so that
    // P GPUs see the same segment length as one (N = 1M: 8 pieces per slice for P = 1, 2, 4, 8; two virtual ranks with
    // 2 pieces of 262144 sources ran 2.3 % behind one rank, with 8 pieces level)
    const int target_blocks = std::max(1, (small ? 2 : 128) * cus / g.nslices);
    sub = (target_blocks + blocks - 1) / blocks;
    int slice_len = g.n / g.nslices;
    int max_sub = std::max(1, slice_len / 128);    // keep >= 128 sources per segment (a wave walks its segment serially)
    sub = std::max(sub, (slice_len + 131071) / 131072);   // and <= 131072 sources (a workgroup's lifetime: the launch's tail)
    // ... unless the partial sums (nseg words per body) would then exceed 1 GiB per rank: at that size (N >= 8M fp32,
    // 4M fp64) a workgroup's lifetime is a negligible part of a step of many seconds anyway; never below 8 segments a step
    const long long words_cap = (1LL << 30) / (long long)word_bytes() / n_local;
    const int mem_sub = (int)std::max(1LL, std::max(8LL, words_cap) / g.nslices);
    sub = std::min(sub, std::max(mem_sub, (target_blocks + blocks - 1) / blocks));
    sub = std::max(1, std::min(std::min(sub, 64), max_sub));
  }
  g.sub = sub;
  g.nseg = g.nslices * g.sub;
  g.fuse = g.opt.fuse < 0 ? (small ? 0 : 1) : g.opt.fuse;
}

// arrival counters: one per wave (4) of every block of 256 rows, padded to a multiple of 16 bytes
size_t ticket_words(const Local& L) { return ((size_t)blocks_for(L.n_local, 1) * 4 + 63) / 64 * 64; }

int alloc_local(Local& L) {
  HIPC(hipSetDevice(L.device));
  HIPC(hipStreamCreateWithFlags(&L.compute, hipStreamNonBlocking));
  HIPC(hipStreamCreateWithFlags(&L.comm, hipStreamNonBlocking));
  const size_t wb = word_bytes();
  const size_t pad = 64;   // words of slack after the arrays (never read by the kernels; keeps SMEM groups in-bounds by construction anyway)
  for (int b = 0; b < 2; ++b) { HIPC(hipMalloc(&L.pos[b], (g.n + pad) * wb)); HIPC(hipMemset(L.pos[b], 0, (g.n + pad) * wb)); }
  HIPC(hipMalloc(&L.vel, (L.n_local + pad) * wb));
  HIPC(hipMalloc(&L.force, (L.n_local + pad) * wb));
  const size_t nt = ticket_words(L);
  HIPC(hipMalloc((void**)&L.tickets, nt * sizeof(unsigned)));
  HIPC(hipMemset(L.tickets, 0, nt * sizeof(unsigned)));
  HIPC(hipMemset(L.vel, 0, (L.n_local + pad) * wb));
  HIPC(hipMemset(L.force, 0, (L.n_local + pad) * wb));
  HIPC(hipEventCreateWithFlags(&L.ev_own_ready, hipEventDisableTiming));
  for (int s = 0; s < g.nranks && s < kMaxRanks; ++s) HIPC(hipEventCreateWithFlags(&L.ev_gather[s], hipEventDisableTiming));
  for (int k = 0; k < kTimerRing; ++k) { HIPC(hipEventCreate(&L.t0[k])); HIPC(hipEventCreate(&L.t1[k])); }
  return NBODY_OK;
}

int ensure_partial(Local& L) {
  const size_t need = (size_t)g.nseg * (size_t)L.n_local;
  if (L.partial && need <= L.partial_words) return NBODY_OK;
  HIPC(hipSetDevice(L.device));
  if (L.partial) { HIPC(hipFree(L.partial)); L.partial = nullptr; L.partial_words = 0; }
  HIPC(hipMalloc(&L.partial, (need + 64) * word_bytes()));
  L.partial_words = need;
  return NBODY_OK;
}

void drop_step_graph() {
  if (g.step_graph) { (void)hipGraphExecDestroy(g.step_graph); g.step_graph = nullptr; }
  g.graph_cur = -1;
}

int reconfigure() {
  const int o_variant = g.variant, o_R = g.R, o_sub = g.sub, o_nsl = g.nslices, o_fuse = g.fuse;
  resolve_config();
  const bool changed = o_variant != g.variant || o_R != g.R || o_sub != g.sub || o_nsl != g.nslices || o_fuse != g.fuse;
  if (changed) drop_step_graph();
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    NBC(ensure_partial(L));
    if (changed) {
      // the arrival counters are zero between steps by construction (the last arriver resets its own); a change of
      // the row-block shape is the one moment to re-zero them all (stream-ordered with the kernels that use them).
      HIPC(hipSetDevice(L.device));
      HIPC(hipMemsetAsync(L.tickets, 0, ticket_words(L) * sizeof(unsigned), L.compute));
    }
  }
  return NBODY_OK;
}

// ---- force-kernel timing ring ----
int timer_drain(Local& L, int keep) {
  while (L.t_count > keep) {
    int idx = (L.t_head - L.t_count + 2 * kTimerRing) % kTimerRing;
    HIPC(hipEventSynchronize(L.t1[idx]));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, L.t0[idx], L.t1[idx]));
    L.t_ms += ms;
    L.t_launches += 1;
    L.t_count--;
  }
  return NBODY_OK;
}

template <typename K>
int launch_timed(Local& L, K kernel, dim3 grid, const ForceArgs& a) {
  int idx = -1;
  if (g.opt.timing) {
    if (L.t_count == kTimerRing) NBC(timer_drain(L, kTimerRing / 2));
    idx = L.t_head;
    HIPC(hipEventRecord(L.t0[idx], L.compute));
  }
  // optional occupancy cap: k workgroups (= k waves per SIMD) per CU by giving each 160 KiB / k of dynamic LDS
  size_t dyn_lds = 0;
  if (g.opt.waves_per_simd > 0 && g.opt.waves_per_simd < 8) {
    dyn_lds = (size_t)(160 * 1024) / (size_t)g.opt.waves_per_simd - 512 - (g.variant == NBODY_VARIANT_LDS ? (size_t)g.tile * 32 : 0);
    if (dyn_lds > 64 * 1024) HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kernel, grid, dim3(kBlock), dyn_lds, L.compute, a);
  HIPC(hipGetLastError());
  if (idx >= 0) {
    HIPC(hipEventRecord(L.t1[idx], L.compute));
    L.t_head = (L.t_head + 1) % kTimerRing;
    L.t_count++;
  }
  return NBODY_OK;
}

template <int R, int ARITH>
int launch_f32_RA(Local& L, dim3 grid, const ForceArgs& a) {
  if constexpr (R == 8) {   // 8 bodies per lane only exists for the SMEM variant
    return launch_timed(L, force_smem_f32<R, ARITH>, grid, a);
  } else {
    switch (g.variant) {
      case NBODY_VARIANT_LDS:
        if (g.tile >= 1024) return launch_timed(L, force_lds_f32<R, ARITH, 1024>, grid, a);
        if (g.tile >= 512) return launch_timed(L, force_lds_f32<R, ARITH, 512>, grid, a);
        return launch_timed(L, force_lds_f32<R, ARITH, 256>, grid, a);
      case NBODY_VARIANT_READLANE:
        return launch_timed(L, force_readlane_f32<R, ARITH>, grid, a);
      default:
        return launch_timed(L, force_smem_f32<R, ARITH>, grid, a);
    }
  }
}

template <int R>
int launch_f32_R(Local& L, dim3 grid, const ForceArgs& a) {
  switch (g.opt.arith) {
    case NBODY_ARITH_REFERENCE: return launch_f32_RA<R, 1>(L, grid, a);
    case NBODY_ARITH_STRICT: return launch_f32_RA<R, 2>(L, grid, a);
    case NBODY_ARITH_REFERENCE_STRICT: return launch_f32_RA<R, 3>(L, grid, a);
    default: return launch_f32_RA<R, 0>(L, grid, a);
  }
}

// how a launch finishes its rows: directly (one segment), by the last-arriving workgroup, or by combine_kernel
inline int finish_mode() { return g.nseg == 1 ? kFinishDirect : (g.fuse ? kFinishLast : kFinishStore); }

void fill_args(Local& L, ForceArgs& a, int row0, int row_count, const Finish& fin, float dt, double dt64) {
  memset(&a, 0, sizeof(a));
  a.src = L.pos[L.cur];
  a.rows = word_ptr(L.pos[L.cur], (size_t)L.first);
  a.partial = L.partial;
  a.vel = L.vel;
  a.pos_next_rows = word_ptr(L.pos[L.cur ^ 1], (size_t)L.first);
  a.force_out = fin.store_force ? L.force : nullptr;
  a.tickets = L.tickets;
  a.n_src = g.n; a.n_rows = L.n_local; a.row0 = row0; a.row_count = row_count;
  a.nslices = g.nslices; a.sub = g.sub; a.nseg = g.nseg;
  a.finish = finish_mode();
  a.do_kick = fin.kick; a.do_drift = fin.drift;
  a.sum_block = (!g.fp64 && g.opt.sum_order == NBODY_SUM_BLOCKED) ? g.opt.sum_block : 0;
  a.fpga16 = g.opt.sum_order == NBODY_SUM_FPGA16;
  a.dt = dt; a.dt64 = dt64;
}

// Launch the force kernel of local L for rows [row0, row0+row_count) against `nsl` source slices
// starting at slice_start and descending (ring arrival order).  A step may take several launches (own slice, then
// arrived slices); the rows are finished when the LAST of a row block's nseg segments has been summed.
int launch_force(Local& L, int row0, int row_count, int slice_start, int nsl, const Finish& fin, float dt, double dt64) {
  if (row_count <= 0 || nsl <= 0) return NBODY_OK;
  HIPC(hipSetDevice(L.device));
  ForceArgs a;
  fill_args(L, a, row0, row_count, fin, dt, dt64);
  a.slice_start = slice_start;
  const int R = g.R;
  dim3 grid(blocks_for(row_count, R), nsl * g.sub, 1);
  // few workgroups per CU = few waves per SIMD and short segments: the scalar loads are no longer hidden by other waves
  const int cus = g.cu_count > 0 ? g.cu_count : 256;
  // (measured at N = 16384, 16 workgroups per CU: +4 % with the long buffers; N = 65536, 64 per CU: -2 %)
  // XCD-aware placement of segments (block_segment): needs a multiple of 8 segment rows in the launch.  Automatic: for
  // launches of >= 4096 row blocks (N = 1M on one GPU: sources fetched once per XCD, 477 MB of memory-side traffic per step
  // instead of 788, time level); with fewer row blocks it measured slower (N = 262144: -2 %, N = 16384: -18 %)
  a.xcd_map = ((g.opt.xcd_map > 0 || (g.opt.xcd_map < 0 && grid.x >= 4096)) && grid.y % 8 == 0) ? 1 : 0;
  a.long_buffers = g.opt.long_buffers < 0 ? ((long long)grid.x * grid.y < 32LL * cus ? 1 : 0) : g.opt.long_buffers;
  if (g.fp64 && g.variant == NBODY_VARIANT_ISA) {
    if (g.opt.isa_phase == 2) return launch_timed(L, force_isa_f64<2>, grid, a);
    return g.opt.isa_phase == 0 ? launch_timed(L, force_isa_f64<0>, grid, a) : launch_timed(L, force_isa_f64<1>, grid, a);
  }
  if (g.fp64) {
    switch (R) {
      case 1: return launch_timed(L, force_smem_f64<1>, grid, a);
      case 2: return launch_timed(L, force_smem_f64<2>, grid, a);
      default: return launch_timed(L, force_smem_f64<4>, grid, a);
    }
  }
  if (a.fpga16) {
    grid.x = blocks_for(row_count, 1);
    switch (g.opt.arith) {
      case NBODY_ARITH_REFERENCE: return launch_timed(L, force_fpga16_f32<1>, grid, a);
      case NBODY_ARITH_STRICT: return launch_timed(L, force_fpga16_f32<2>, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_timed(L, force_fpga16_f32<3>, grid, a);
      default: return launch_timed(L, force_fpga16_f32<0>, grid, a);
    }
  }
  if (g.variant == NBODY_VARIANT_ISA) {   // one body per lane, fast arithmetic only (resolve_config guarantees both)
    if (a.long_buffers) return launch_timed(L, force_isa_long_f32, grid, a);
    switch (g.opt.isa_phase) {
      case 2: return launch_timed(L, force_isa_f32<2>, grid, a);
      case 3: return launch_timed(L, force_isa_f32<3>, grid, a);   // 3..5: timing-only diagnostics, wrong results
      case 4: return launch_timed(L, force_isa_f32<4>, grid, a);
      case 5: return launch_timed(L, force_isa_f32<5>, grid, a);
      case 6: return launch_timed(L, force_isa_f32<6>, grid, a);
      case 7: return launch_timed(L, force_isa_f32<7>, grid, a);
      case 8: return launch_timed(L, force_isa_f32<8>, grid, a);
      case 9: return launch_timed(L, force_isa_f32<9>, grid, a);    // 9..13: correct loops, other encodings (experiments)
      case 10: return launch_timed(L, force_isa_f32<10>, grid, a);
      case 11: return launch_timed(L, force_isa_f32<11>, grid, a);
      case 12: return launch_timed(L, force_isa_f32<12>, grid, a);
      case 13: return launch_timed(L, force_isa_f32<13>, grid, a);
      case 14: return launch_timed(L, force_isa_f32<14>, grid, a);   // 14, 15: timing-only diagnostics, wrong results
      case 15: return launch_timed(L, force_isa_f32<15>, grid, a);
      case 16: return launch_timed(L, force_isa_f32<16>, grid, a);   // correct: SGPR operand in src1
      case 17: return launch_timed(L, force_isa_f32<17>, grid, a);   // correct: dx, dy in one packed subtraction
      case 18: return launch_timed(L, force_isa_f32<18>, grid, a);   // correct: eps from a VGPR instead of the v_fmaak_f32 literal
      default: break;
    }
    return g.opt.isa_phase == 0 ? launch_timed(L, force_isa_f32<0>, grid, a) : launch_timed(L, force_isa_f32<1>, grid, a);
  }
  switch (R) {
    case 1: return launch_f32_R<1>(L, grid, a);
    case 2: return launch_f32_R<2>(L, grid, a);
    case 8: return launch_f32_R<8>(L, grid, a);
    default: return launch_f32_R<4>(L, grid, a);
  }
}

// the two-launch form (NBODY_OPT_FUSE_COMBINE = 0): after the step's last force launch, add the partials
int launch_combine(Local& L, int row0, int row_count, const Finish& fin, float dt, double dt64) {
  if (row_count <= 0 || finish_mode() != kFinishStore) return NBODY_OK;
  HIPC(hipSetDevice(L.device));
  ForceArgs c;
  fill_args(L, c, row0, row_count, fin, dt, dt64);
  dim3 grid((row_count + kBlock - 1) / kBlock);
  if (g.fp64) hipLaunchKernelGGL((combine_kernel<double, d4>), grid, dim3(kBlock), 0, L.compute, c);
  else hipLaunchKernelGGL((combine_kernel<float, f4>), grid, dim3(kBlock), 0, L.compute, c);
  HIPC(hipGetLastError());
  return NBODY_OK;
}

inline int ring_slice(int rank, int s) { int q = (rank - s) % g.nranks; return q < 0 ? q + g.nranks : q; }

// Host-staged all-gather of one sharded device array (words [first, first+count) are this rank's): D2H own part,
// callback (the host framework's all-gather fills the rest of g.host_stage), H2D everything else on the comm stream.
int host_exchange(Local& L, void* dev_full, int first, int count, bool wait_own_ready) {
  const size_t wb = word_bytes();
  if (!g.host_stage) HIPC(hipHostMalloc(&g.host_stage, (size_t)(g.n + 64) * 32, hipHostMallocDefault));
  if (wait_own_ready) HIPC(hipEventSynchronize(L.ev_own_ready));
  HIPC(hipMemcpy(word_ptr(g.host_stage, first), word_ptr(dev_full, first), (size_t)count * wb, hipMemcpyDeviceToHost));
  int rc = g.host_gather(g.host_gather_user, g.host_stage, g.n, (int)wb, L.rank, g.nranks);
  if (rc) return NBODY_ERR_STATE;
  if (first > 0) HIPC(hipMemcpyAsync(dev_full, g.host_stage, (size_t)first * wb, hipMemcpyHostToDevice, L.comm));
  const int after = first + count;
  if (after < g.n)
    HIPC(hipMemcpyAsync(word_ptr(dev_full, after), word_ptr(g.host_stage, after), (size_t)(g.n - after) * wb, hipMemcpyHostToDevice, L.comm));
  return NBODY_OK;
}

// One ring step on the comm stream: send `send_bytes` at `send_ptr` to the next rank, receive `recv_bytes` at `recv_ptr`
// from the previous one, as one RCCL group (so neither side blocks the other).  With one rank next = prev = self and the
// pair is a device-local copy through RCCL (what nbody_comm_selftest runs on a one-GPU box).
int ring_step(Local& L, const void* send_ptr, size_t send_bytes, void* recv_ptr, size_t recv_bytes) {
  const int P = g.nranks;
  const int next = (L.rank + 1) % P, prev = (L.rank + P - 1) % P;
  NCCLC(g_rccl.GroupStart());
  NCCLC(g_rccl.Send(send_ptr, send_bytes, ncclChar, next, L.comm_h, L.comm));
  NCCLC(g_rccl.Recv(recv_ptr, recv_bytes, ncclChar, prev, L.comm_h, L.comm));
  NCCLC(g_rccl.GroupEnd());
  return NBODY_OK;
}

// RCCL all-gather of one sharded device array in place on the comm stream (multi-process).  Default (NBODY_COMM_AUTO,
// NBODY_COMM_RING): the north_star's ring — P-1 steps, step s forwards the slice that arrived at step s-1 (the rank's
// own at s = 1) and receives slice (rank - s) mod P; ev[s], if given, is recorded as soon as that slice has landed, so
// the force kernel over it can start while the next one travels.  NBODY_COMM_ALLGATHER: one in-place ncclAllGather
// (equal slices only; all events after the collective).
int rccl_gather(Local& L, void* dev_full, hipEvent_t* ev) {
  const int P = g.nranks;
  const size_t wb = word_bytes();
  const bool even = (g.n % P) == 0;
  if (even && g.opt.comm == NBODY_COMM_ALLGATHER) {
    NCCLC(g_rccl.AllGather(word_ptr(dev_full, L.first), dev_full, (size_t)L.n_local * wb, ncclChar, L.comm_h, L.comm));
    if (ev) for (int s = 1; s < P; ++s) HIPC(hipEventRecord(ev[s], L.comm));
    return NBODY_OK;
  }
  if (g.opt.comm == NBODY_COMM_DIRECT) {
    // fully connected: the own slice goes straight to every peer and every peer's slice comes straight back, all in one
    // RCCL group — one hop over the 7 xGMI links at once instead of P-1 ring hops
    NCCLC(g_rccl.GroupStart());
    for (int s = 1; s < P; ++s) {
      const int to = (L.rank + s) % P, from = ring_slice(L.rank, s);
      const int fr = slice_first(from, g.n, P), lr = slice_first(from + 1, g.n, P) - fr;
      NCCLC(g_rccl.Send(word_ptr(dev_full, L.first), (size_t)L.n_local * wb, ncclChar, to, L.comm_h, L.comm));
      NCCLC(g_rccl.Recv(word_ptr(dev_full, fr), (size_t)lr * wb, ncclChar, from, L.comm_h, L.comm));
    }
    NCCLC(g_rccl.GroupEnd());
    if (ev) for (int s = 1; s < P; ++s) HIPC(hipEventRecord(ev[s], L.comm));
    return NBODY_OK;
  }
  for (int s = 1; s < P; ++s) {
    const int qs = ring_slice(L.rank, s - 1);   // forward what arrived last (own slice at s = 1)
    const int qr = ring_slice(L.rank, s);
    const int fs = slice_first(qs, g.n, P), ls = slice_first(qs + 1, g.n, P) - fs;
    const int fr = slice_first(qr, g.n, P), lr = slice_first(qr + 1, g.n, P) - fr;
    NBC(ring_step(L, word_ptr(dev_full, fs), (size_t)ls * wb, word_ptr(dev_full, fr), (size_t)lr * wb));
    if (ev) HIPC(hipEventRecord(ev[s], L.comm));
  }
  return NBODY_OK;
}

// Bring the other ranks' slices of pos[buf] to every local.  Enqueued on the comm streams; records
// ev_gather[s] (s = 1..P-1) as slices arrive.  Sources are valid after their owner's ev_own_ready.
int enqueue_gather(int buf) {
  const int P = g.nranks;
  if (P == 1) return NBODY_OK;
  const size_t wb = word_bytes();
  if (!g.multiprocess) {
    // one process, P devices: every local pulls each remote slice straight from its owner (xGMI is
    // fully connected: one hop, all links busy), in ring order so arrival order matches the RCCL path.
    for (int l = 0; l < g.nlocal; ++l) {
      Local& L = g.loc[l];
      HIPC(hipSetDevice(L.device));
      for (int s = 1; s < P; ++s) {
        Local& O = g.loc[ring_slice(L.rank, s)];
        HIPC(hipStreamWaitEvent(L.comm, O.ev_own_ready, 0));
        HIPC(hipMemcpyPeerAsync(word_ptr(L.pos[buf], O.first), L.device, word_ptr(O.pos[buf], O.first), O.device,
                                (size_t)O.n_local * wb, L.comm));
        HIPC(hipEventRecord(L.ev_gather[s], L.comm));
      }
    }
    return NBODY_OK;
  }
  Local& L = g.loc[0];
  HIPC(hipSetDevice(L.device));
  if (!g.host_gather && !L.comm_h) return NBODY_ERR_STATE;   // neither RCCL nor a host transport was set up
  if (g.host_gather) {
    // host-staged transport (no RCCL): own slice down, exchange on the host, the other slices up
    NBC(host_exchange(L, L.pos[buf], L.first, L.n_local, true));
    for (int s = 1; s < P; ++s) HIPC(hipEventRecord(L.ev_gather[s], L.comm));
    return NBODY_OK;
  }
  HIPC(hipStreamWaitEvent(L.comm, L.ev_own_ready, 0));
  return rccl_gather(L, L.pos[buf], L.ev_gather);
}

// One step on every local: forces on pos[cur], kick, drift into pos[cur^1], swap.
int enqueue_step(float dt, double dt64) {
  const int P = g.nranks;
  const Finish fin = {true, true, false};
  const bool need_gather = !g.loc[0].all_present;
  if (need_gather && g.opt.overlap) {
    // The own-slice kernels run while the other slices travel on the second stream.  Device-side transports (RCCL,
    // peer copies) are enqueued FIRST: they only wait for the previous step's end, and their small kernels/copies get
    // onto the device ahead of the force launch that fills every CU; the host-staged exchange blocks the host, so
    // there the force launch goes first.
    const bool host_staged = g.multiprocess && g.host_gather;
    if (!host_staged) NBC(enqueue_gather(g.loc[0].cur));
    for (int l = 0; l < g.nlocal; ++l) NBC(launch_force(g.loc[l], 0, g.loc[l].n_local, g.loc[l].rank, 1, fin, dt, dt64));
    if (host_staged) NBC(enqueue_gather(g.loc[0].cur));
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    if (need_gather && !g.opt.overlap) {
      // gather first, then one launch over everything
      if (l == 0) NBC(enqueue_gather(L.cur));
    }
    if (P == 1) {
      NBC(launch_force(L, 0, L.n_local, g.nslices - 1, g.nslices, fin, dt, dt64));
    } else if (!need_gather) {
      NBC(launch_force(L, 0, L.n_local, L.rank, P, fin, dt, dt64));
    } else if (g.opt.overlap == 2) {
      // one launch per arriving slice, each released by that slice's event (ring arrival order)
      for (int s = 1; s < P; ++s) {
        HIPC(hipStreamWaitEvent(L.compute, L.ev_gather[s], 0));
        NBC(launch_force(L, 0, L.n_local, ring_slice(L.rank, s), 1, fin, dt, dt64));
      }
    } else if (g.opt.overlap) {
      // the other slices in one launch once they have all arrived (at N = 1M the ring takes ~0.4 ms against
      // ~4 ms of own-slice work already running)
      HIPC(hipStreamWaitEvent(L.compute, L.ev_gather[P - 1], 0));
      NBC(launch_force(L, 0, L.n_local, ring_slice(L.rank, 1), P - 1, fin, dt, dt64));
    } else {
      HIPC(hipStreamWaitEvent(L.compute, L.ev_gather[P - 1], 0));
      NBC(launch_force(L, 0, L.n_local, L.rank, P, fin, dt, dt64));
    }
    NBC(launch_combine(L, 0, L.n_local, fin, dt, dt64));
    HIPC(hipEventRecord(L.ev_own_ready, L.compute));
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    // (the NEXT gather writes into the buffer this step read; it waits on ev_own_ready, recorded after this
    //  step's last kernel, inside enqueue_gather)
    L.cur ^= 1;
    L.all_present = (P == 1);
  }
  g.steps_done++;
  return NBODY_OK;
}

int sync_all() {
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipStreamSynchronize(L.comm));
    HIPC(hipStreamSynchronize(L.compute));
  }
  return NBODY_OK;
}

// make pos[cur] complete on every local (after a step only the own slice is there)
int complete_positions() {
  if (g.nranks == 1 || g.loc[0].all_present) return NBODY_OK;
  NBC(enqueue_gather(g.loc[0].cur));
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipStreamWaitEvent(L.compute, L.ev_gather[g.nranks - 1], 0));
    L.all_present = true;
  }
  return sync_all();
}

int pick_device(int rank, int ndev) {
  const char* e = getenv("NBODY_DEVICE");
  if (e && *e) return atoi(e) % ndev;
  e = getenv("LOCAL_RANK");
  if (e && *e) return atoi(e) % ndev;
  return rank % ndev;
}

int init_common(int n, int fp64, int tile) {
  if (n <= 0 || n > (1 << 30)) return NBODY_ERR_ARG;
  if (tile == 0) tile = 256;
  if (tile < 64 || tile > 1024 || tile % 64) return NBODY_ERR_ARG;
  g.n = n; g.fp64 = fp64 ? 1 : 0; g.tile = tile;
  g.steps_done = 0; g.stepped_eagerly = false;
  return NBODY_OK;
}

int device_count(int* ndev) {
  hipError_t e = hipGetDeviceCount(ndev);
  if (e != hipSuccess || *ndev <= 0) return NBODY_ERR_NO_DEVICE;
  return NBODY_OK;
}

void free_local(Local& L) {
  if (L.compute == nullptr && L.pos[0] == nullptr) return;
  (void)hipSetDevice(L.device);
  if (L.compute) (void)hipStreamSynchronize(L.compute);
  if (L.comm) (void)hipStreamSynchronize(L.comm);
  if (L.comm_h && g_rccl.CommDestroy) g_rccl.CommDestroy(L.comm_h);
  for (int b = 0; b < 2; ++b) if (L.pos[b]) (void)hipFree(L.pos[b]);
  if (L.vel) (void)hipFree(L.vel);
  if (L.partial) (void)hipFree(L.partial);
  if (L.force) (void)hipFree(L.force);
  if (L.tickets) (void)hipFree(L.tickets);
  if (L.full_scratch) (void)hipFree(L.full_scratch);
  if (L.ev_own_ready) (void)hipEventDestroy(L.ev_own_ready);
  for (int s = 0; s < kMaxRanks; ++s) if (L.ev_gather[s]) (void)hipEventDestroy(L.ev_gather[s]);
  for (int k = 0; k < kTimerRing; ++k) { if (L.t0[k]) (void)hipEventDestroy(L.t0[k]); if (L.t1[k]) (void)hipEventDestroy(L.t1[k]); }
  if (L.compute) (void)hipStreamDestroy(L.compute);
  if (L.comm) (void)hipStreamDestroy(L.comm);
  L = Local();
}

int upload_impl(const void* pos, const void* vel) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!pos || !vel) return NBODY_ERR_ARG;
  const size_t wb = word_bytes();
  NBC(sync_all());
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpyAsync(L.pos[L.cur], pos, (size_t)g.n * wb, hipMemcpyHostToDevice, L.compute));
    HIPC(hipMemcpyAsync(L.vel, (const char*)vel + (size_t)L.first * wb, (size_t)L.n_local * wb, hipMemcpyHostToDevice, L.compute));
    HIPC(hipEventRecord(L.ev_own_ready, L.compute));
    L.all_present = true;
  }
  return sync_all();
}

// Multi-process: all-gather a rank-sharded array (n_local words on every rank: velocities, forces) into
// L.full_scratch (N words) with the transport in use.  The compute stream must be idle.
int gather_sharded_multiprocess(Local& L, const void* own_rows) {
  const size_t wb = word_bytes();
  HIPC(hipSetDevice(L.device));
  if (!L.full_scratch) HIPC(hipMalloc(&L.full_scratch, (size_t)(g.n + 64) * wb));
  HIPC(hipMemcpyAsync(word_ptr(L.full_scratch, L.first), own_rows, (size_t)L.n_local * wb, hipMemcpyDeviceToDevice, L.comm));
  if (g.host_gather) {
    HIPC(hipStreamSynchronize(L.comm));
    NBC(host_exchange(L, L.full_scratch, L.first, L.n_local, false));
    HIPC(hipStreamSynchronize(L.comm));
    return NBODY_OK;
  }
  if (!L.comm_h) return NBODY_ERR_STATE;
  NBC(rccl_gather(L, L.full_scratch, nullptr));
  HIPC(hipStreamSynchronize(L.comm));
  return NBODY_OK;
}

int download_impl(void* pos, void* vel) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!pos || !vel) return NBODY_ERR_ARG;
  const size_t wb = word_bytes();
  NBC(sync_all());
  if (g.multiprocess && g.nranks > 1) {
    Local& L = g.loc[0];
    NBC(complete_positions());
    NBC(gather_sharded_multiprocess(L, L.vel));
    HIPC(hipMemcpy(pos, L.pos[L.cur], (size_t)g.n * wb, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(vel, L.full_scratch, (size_t)g.n * wb, hipMemcpyDeviceToHost));
    return NBODY_OK;
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpy((char*)pos + (size_t)L.first * wb, word_ptr(L.pos[L.cur], L.first), (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy((char*)vel + (size_t)L.first * wb, L.vel, (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  }
  return NBODY_OK;
}

// forces of rows [row0, row0+count) of every local's slice, from pos[cur] (must be complete)
int forces_on_device(int row0, int count_or_all) {
  NBC(reconfigure());
  NBC(complete_positions());
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    int cnt = count_or_all < 0 ? L.n_local : count_or_all;
    int r0 = count_or_all < 0 ? 0 : row0;
    const Finish fin = {false, false, true};
    NBC(launch_force(L, r0, cnt, g.nslices - 1, g.nslices, fin, 0.f, 0.0));
    NBC(launch_combine(L, r0, cnt, fin, 0.f, 0.0));
  }
  return sync_all();
}

int step_impl(float dt, double dt64, int nsteps) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (nsteps < 0) return NBODY_ERR_ARG;
  NBC(reconfigure());
  int s = 0;
  // One GPU, no per-kernel timing events, enough steps: replay a captured pair of steps.  A step is 1-2 kernel
  // launches + an event; below N ~ 10^5 the host launch path, not the GPU, sets the pace.
  if (g.opt.graph && g.nranks == 1 && g.nlocal == 1 && !g.opt.timing && nsteps >= 4) {
    Local& L = g.loc[0];
    HIPC(hipSetDevice(L.device));
    // A graph captured before the step's kernels have ever really run replays slower for good — measured at N = 4096:
    // 15.1 us per step when the capture is the first thing after nbody_upload, 13.1 us when one eager step came first,
    // whichever buffer is current and however often the graph is reused (profiles/r02_small_n.md).  So the first step of
    // an engine's life is always launched eagerly.
    if (!g.stepped_eagerly) { NBC(enqueue_step(dt, dt64)); ++s; g.stepped_eagerly = true; }
    if (!g.step_graph || g.graph_cur != L.cur || g.graph_dt != dt || g.graph_dt64 != dt64) {
      drop_step_graph();
      hipGraph_t graph = nullptr;
      const long long done = g.steps_done;
      const int cur0 = L.cur;
      const bool present0 = L.all_present;
      HIPC(hipStreamBeginCapture(L.compute, hipStreamCaptureModeThreadLocal));
      int rc = enqueue_step(dt, dt64);
      if (!rc) rc = enqueue_step(dt, dt64);
      hipError_t e = hipStreamEndCapture(L.compute, &graph);
      g.steps_done = done;                       // capturing executes nothing
      L.cur = cur0; L.all_present = present0;    // two steps return to the same buffer; a failed capture may have toggled once
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); g_last_line = __LINE__; return (int)e; }
      e = hipGraphInstantiate(&g.step_graph, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIPC(e);
      g.graph_cur = L.cur; g.graph_dt = dt; g.graph_dt64 = dt64;
    }
    for (; s + 2 <= nsteps; s += 2) { HIPC(hipGraphLaunch(g.step_graph, L.compute)); g.steps_done += 2; }
  }
  if (s < nsteps) g.stepped_eagerly = true;
  for (; s < nsteps; ++s) NBC(enqueue_step(dt, dt64));
  return NBODY_OK;
}

int body_force_impl(void* pos, void* vel, float dt, double dt64, int n) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (n != g.n) return NBODY_ERR_ARG;
  NBC(upload_impl(pos, vel));
  NBC(reconfigure());
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    const Finish fin = {true, false, true};
    NBC(launch_force(L, 0, L.n_local, g.nslices - 1, g.nslices, fin, dt, dt64));
    NBC(launch_combine(L, 0, L.n_local, fin, dt, dt64));
  }
  NBC(sync_all());
  // vel back (pos is read-only for bodyForce)
  const size_t wb = word_bytes();
  if (g.multiprocess && g.nranks > 1) {
    Local& L = g.loc[0];
    NBC(gather_sharded_multiprocess(L, L.vel));
    HIPC(hipMemcpy(vel, L.full_scratch, (size_t)g.n * wb, hipMemcpyDeviceToHost));
    return NBODY_OK;
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpy((char*)vel + (size_t)L.first * wb, L.vel, (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  }
  return NBODY_OK;
}

int integrate_impl(void* pos, const void* vel, float dt, double dt64, int n) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (n != g.n) return NBODY_ERR_ARG;
  NBC(upload_impl(pos, vel));
  const size_t wb = word_bytes();
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    dim3 grid((L.n_local + kBlock - 1) / kBlock);
    if (L.n_local > 0) {
      if (g.fp64) hipLaunchKernelGGL((drift_kernel<double, d4>), grid, dim3(kBlock), 0, L.compute, (d4*)word_ptr(L.pos[L.cur], L.first), (const d4*)L.vel, L.n_local, dt, dt64);
      else hipLaunchKernelGGL((drift_kernel<float, f4>), grid, dim3(kBlock), 0, L.compute, (f4*)word_ptr(L.pos[L.cur], L.first), (const f4*)L.vel, L.n_local, dt, dt64);
      HIPC(hipGetLastError());
    }
    HIPC(hipEventRecord(L.ev_own_ready, L.compute));
    L.all_present = (g.nranks == 1);
  }
  NBC(sync_all());
  if (g.multiprocess && g.nranks > 1) {
    NBC(complete_positions());
    Local& L = g.loc[0];
    HIPC(hipMemcpy(pos, L.pos[L.cur], (size_t)g.n * wb, hipMemcpyDeviceToHost));
    return NBODY_OK;
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpy((char*)pos + (size_t)L.first * wb, word_ptr(L.pos[L.cur], L.first), (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  }
  // the other locals' copies of pos are now stale: refresh lazily
  for (int l = 0; l < g.nlocal; ++l) g.loc[l].all_present = (g.nranks == 1);
  return NBODY_OK;
}

int forces_impl(const void* pos_words, void* force_words, int n) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (n != g.n || !pos_words || !force_words) return NBODY_ERR_ARG;
  const size_t wb = word_bytes();
  NBC(sync_all());
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpyAsync(L.pos[L.cur], pos_words, (size_t)g.n * wb, hipMemcpyHostToDevice, L.compute));
    L.all_present = true;
  }
  NBC(forces_on_device(0, -1));
  if (g.multiprocess && g.nranks > 1) {   // every process returns all N force words: gather the other ranks' rows
    Local& L = g.loc[0];
    NBC(gather_sharded_multiprocess(L, L.force));
    HIPC(hipMemcpy(force_words, L.full_scratch, (size_t)g.n * wb, hipMemcpyDeviceToHost));
    return NBODY_OK;
  }
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpy((char*)force_words + (size_t)L.first * wb, L.force, (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  }
  return NBODY_OK;
}

int forces_rows_impl(int first_row, int n_rows, void* force_words) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.nlocal != 1 || !force_words) return NBODY_ERR_UNSUPPORTED;
  Local& L = g.loc[0];
  if (first_row < 0 || n_rows <= 0 || first_row + n_rows > L.n_local) return NBODY_ERR_ARG;
  NBC(forces_on_device(first_row, n_rows));
  HIPC(hipSetDevice(L.device));
  HIPC(hipMemcpy(force_words, word_ptr(L.force, first_row), (size_t)n_rows * word_bytes(), hipMemcpyDeviceToHost));
  return NBODY_OK;
}

// two HIP events that are destroyed on every way out
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

}  // namespace

// ============================================================================
extern "C" {

int nbody_init(int n, int ngpus, int fp64, int tile) {
  if (g.init) nbody_shutdown();
  if (ngpus <= 0 || ngpus > kMaxLocal) return NBODY_ERR_ARG;
  NBC(init_common(n, fp64, tile));
  int ndev = 0;
  NBC(device_count(&ndev));
  // NBODY_OVERSUBSCRIBE=1 lets several virtual ranks share a device (bring-up of the multi-GPU
  // schedule on a one-GPU box; the data path is identical apart from the copies staying on-device).
  const char* ov = getenv("NBODY_OVERSUBSCRIBE");
  if (ngpus > ndev && !(ov && atoi(ov))) return NBODY_ERR_NO_DEVICE;
  if (n < ngpus) return NBODY_ERR_ARG;
  g.nranks = ngpus; g.nlocal = ngpus; g.multiprocess = false;
  hipDeviceProp_t prop;
  for (int r = 0; r < ngpus; ++r) {
    Local& L = g.loc[r];
    L = Local();
    L.rank = r;
    L.device = ngpus == 1 ? pick_device(0, ndev) : r % ndev;
    L.first = slice_first(r, n, ngpus);
    L.n_local = slice_first(r + 1, n, ngpus) - L.first;
    int e = alloc_local(L);
    if (e) { nbody_shutdown(); return e; }
  }
  {
    hipError_t pe = hipGetDeviceProperties(&prop, g.loc[0].device);
    if (pe != hipSuccess) { g_last_line = __LINE__; nbody_shutdown(); return (int)pe; }
  }
  g.cu_count = prop.multiProcessorCount; g.clock_khz = prop.clockRate;
  if (ngpus > 1) {
    for (int a = 0; a < ngpus; ++a)
      for (int b = 0; b < ngpus; ++b) {
        if (g.loc[a].device == g.loc[b].device) continue;
        (void)hipSetDevice(g.loc[a].device);
        hipError_t e = hipDeviceEnablePeerAccess(g.loc[b].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); }
      }
  }
  g.init = true;
  g.opt = Options();
  int e = reconfigure();
  if (e) { nbody_shutdown(); return e; }
  return NBODY_OK;
}

int nbody_unique_id(void* uid128) {
  if (!uid128) return NBODY_ERR_ARG;
  NBC(rccl_load());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(uid128, &id, sizeof(id));
  return NBODY_OK;
}

int nbody_init_rank(int n, int fp64, int tile, int rank, int nranks, const void* uid128) {
  if (g.init) nbody_shutdown();
  if (nranks <= 0 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return NBODY_ERR_ARG;
  NBC(init_common(n, fp64, tile));   // uid128 == NULL with nranks > 1: no RCCL, nbody_set_host_gather() must follow
  if (n < nranks) return NBODY_ERR_ARG;
  int ndev = 0;
  NBC(device_count(&ndev));
  g.nranks = nranks; g.nlocal = 1; g.multiprocess = true;
  Local& L = g.loc[0];
  L = Local();
  L.rank = rank;
  L.device = pick_device(rank, ndev);
  L.first = slice_first(rank, n, nranks);
  L.n_local = slice_first(rank + 1, n, nranks) - L.first;
  int e = alloc_local(L);
  if (e) { nbody_shutdown(); return e; }
  hipDeviceProp_t prop;
  {
    hipError_t pe = hipGetDeviceProperties(&prop, L.device);
    if (pe != hipSuccess) { g_last_line = __LINE__; nbody_shutdown(); return (int)pe; }
  }
  g.cu_count = prop.multiProcessorCount; g.clock_khz = prop.clockRate;
  if (uid128) {
    // a communicator is created whenever an id is given — also for nranks = 1, where it carries no traffic in a step
    // but lets nbody_comm_selftest() push bytes through the same RCCL calls the multi-GPU job makes
    e = rccl_load();
    if (e) { nbody_shutdown(); return e; }
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof(id));
    hipError_t de = hipSetDevice(L.device);
    if (de != hipSuccess) { g_last_line = __LINE__; nbody_shutdown(); return (int)de; }
    ncclResult_t r = g_rccl.CommInitRank(&L.comm_h, nranks, id, rank);
    if (r != ncclSuccess) { g_last_line = __LINE__; L.comm_h = nullptr; nbody_shutdown(); return 2000 + (int)r; }
  }
  g.init = true;
  g.opt = Options();
  e = reconfigure();
  if (e) { nbody_shutdown(); return e; }
  if (nranks > 1 && L.comm_h) {
    // One all-gather of the (zeroed) position buffer now: RCCL sets up its rings/channels lazily on the first
    // collective, and that must not land in a caller's first timed step.
    hipError_t he = hipSetDevice(L.device);
    if (he == hipSuccess) he = hipEventRecord(L.ev_own_ready, L.compute);
    e = he != hipSuccess ? (int)he : enqueue_gather(L.cur);
    if (!e) e = sync_all();
    if (e) { nbody_shutdown(); return e; }
  }
  return NBODY_OK;
}

// Transport self-test on the communicator of nbody_init_rank: (1) an in-place all-gather of a patterned scratch array
// through rccl_gather() in the configured NBODY_OPT_COMM form, (2) one ring step (ncclSend to rank+1, ncclRecv from
// rank-1, grouped) of a patterned block — with one rank both are device-local, which is how a one-GPU box exercises
// the library's RCCL calls (symbols, argument order, byte counts).  Every received word is checked on the host.
int nbody_comm_selftest(long long* bytes_moved) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!g.multiprocess) return NBODY_ERR_STATE;
  Local& L = g.loc[0];
  if (!L.comm_h) return NBODY_ERR_STATE;
  NBC(sync_all());
  const size_t wb = word_bytes();
  const int P = g.nranks;
  HIPC(hipSetDevice(L.device));
  if (!L.full_scratch) HIPC(hipMalloc(&L.full_scratch, (size_t)(g.n + 64) * wb));
  // (1) all-gather: word w of rank q's slice = q * 2^24 + (w mod 2^24), in every 4-byte lane of the word
  std::vector<uint32_t> host((size_t)g.n * (wb / 4));
  HIPC(hipMemset(L.full_scratch, 0xff, (size_t)g.n * wb));
  for (int w = L.first; w < L.first + L.n_local; ++w)
    for (size_t k = 0; k < wb / 4; ++k) host[(size_t)w * (wb / 4) + k] = ((uint32_t)L.rank << 24) + ((uint32_t)w & 0xffffffu);
  HIPC(hipMemcpy(word_ptr(L.full_scratch, L.first), &host[(size_t)L.first * (wb / 4)], (size_t)L.n_local * wb, hipMemcpyHostToDevice));
  NBC(rccl_gather(L, L.full_scratch, nullptr));
  HIPC(hipStreamSynchronize(L.comm));
  HIPC(hipMemcpy(host.data(), L.full_scratch, (size_t)g.n * wb, hipMemcpyDeviceToHost));
  for (int q = 0; q < P; ++q)
    for (int w = slice_first(q, g.n, P); w < slice_first(q + 1, g.n, P); ++w)
      for (size_t k = 0; k < wb / 4; ++k)
        if (host[(size_t)w * (wb / 4) + k] != ((uint32_t)q << 24) + ((uint32_t)w & 0xffffffu)) { g_last_line = __LINE__; return NBODY_ERR_STATE; }
  long long moved = (long long)(g.n - L.n_local) * (long long)wb;
  // (2) one ring step: the first half of the scratch array goes to rank+1, the second half is received from rank-1
  const int half = g.n / 2;
  if (half > 0) {
    const int prev = (L.rank + P - 1) % P;
    for (int w = 0; w < half; ++w)
      for (size_t k = 0; k < wb / 4; ++k) host[(size_t)w * (wb / 4) + k] = 0xA5000000u + ((uint32_t)L.rank << 20) + ((uint32_t)w & 0xfffffu);
    HIPC(hipMemcpy(L.full_scratch, host.data(), (size_t)half * wb, hipMemcpyHostToDevice));
    HIPC(hipMemset(word_ptr(L.full_scratch, half), 0, (size_t)half * wb));
    NBC(ring_step(L, L.full_scratch, (size_t)half * wb, word_ptr(L.full_scratch, half), (size_t)half * wb));
    HIPC(hipStreamSynchronize(L.comm));
    HIPC(hipMemcpy(host.data(), word_ptr(L.full_scratch, half), (size_t)half * wb, hipMemcpyDeviceToHost));
    for (int w = 0; w < half; ++w)
      for (size_t k = 0; k < wb / 4; ++k)
        if (host[(size_t)w * (wb / 4) + k] != 0xA5000000u + ((uint32_t)prev << 20) + ((uint32_t)w & 0xfffffu)) { g_last_line = __LINE__; return NBODY_ERR_STATE; }
    moved += (long long)half * (long long)wb;
  }
  if (bytes_moved) *bytes_moved = moved;
  return NBODY_OK;
}

void nbody_shutdown(void) {
  drop_step_graph();
  for (int l = 0; l < kMaxLocal; ++l) free_local(g.loc[l]);
  if (g.host_stage) { (void)hipHostFree(g.host_stage); g.host_stage = nullptr; }
  g.host_gather = nullptr; g.host_gather_user = nullptr;
  g.init = false; g.nlocal = 0; g.nranks = 1;
}

int nbody_set_option(int key, int value) {
  switch (key) {
    case NBODY_OPT_VARIANT: if (value < 0 || value > 4) return NBODY_ERR_ARG; g.opt.variant = value; break;
    case NBODY_OPT_IBLOCK: if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return NBODY_ERR_ARG; g.opt.iblock = value; break;
    case NBODY_OPT_JSUB: if (value < 0 || value > 256) return NBODY_ERR_ARG; g.opt.jsub = value; break;
    case NBODY_OPT_JSLICES: if (value < 0 || value > kMaxRanks) return NBODY_ERR_ARG; g.opt.jslices = value; break;
    case NBODY_OPT_ARITH: if (value < 0 || value > 3) return NBODY_ERR_ARG; g.opt.arith = value; break;
    case NBODY_OPT_SUM_ORDER: if (value < 0 || value > 2) return NBODY_ERR_ARG; g.opt.sum_order = value; break;
    case NBODY_OPT_SUM_BLOCK: if (value < 8 || value > (1 << 24) || value % 64) return NBODY_ERR_ARG; g.opt.sum_block = value; break;
    case NBODY_OPT_FUSE_COMBINE: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.fuse = value; break;
    case NBODY_OPT_ISA_LONG_BUFFERS: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.long_buffers = value; break;
    case NBODY_OPT_XCD_MAP: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.xcd_map = value; break;
    case NBODY_OPT_TIMING: g.opt.timing = value ? 1 : 0; break;
    case NBODY_OPT_COMM: if (value < 0 || value > 3) return NBODY_ERR_ARG; g.opt.comm = value; break;
    case NBODY_OPT_OVERLAP: if (value < 0 || value > 2) return NBODY_ERR_ARG; g.opt.overlap = value; break;
    case NBODY_OPT_GRAPH: g.opt.graph = value ? 1 : 0; break;
    case NBODY_OPT_WAVES_PER_SIMD: if (value < 0 || value > 8) return NBODY_ERR_ARG; g.opt.waves_per_simd = value; break;
    case NBODY_OPT_ISA_PHASE: if (value < 0 || value > 18) return NBODY_ERR_ARG; g.opt.isa_phase = value; break;
    default: return NBODY_ERR_ARG;
  }
  if (g.init) { NBC(sync_all()); drop_step_graph(); return reconfigure(); }
  return NBODY_OK;
}

int nbody_get_info(int key, long long* value) {
  if (!value) return NBODY_ERR_ARG;
  if (!g.init) return NBODY_ERR_NOT_INIT;
  const Local& L = g.loc[0];
  switch (key) {
    case NBODY_INFO_N: *value = g.n; break;
    case NBODY_INFO_N_LOCAL: *value = L.n_local; break;
    case NBODY_INFO_FIRST_BODY: *value = L.first; break;
    case NBODY_INFO_RANK: *value = L.rank; break;
    case NBODY_INFO_NRANKS: *value = g.nranks; break;
    case NBODY_INFO_VARIANT: *value = g.variant; break;
    case NBODY_INFO_IBLOCK: *value = g.R; break;
    case NBODY_INFO_JSUB: *value = g.sub; break;
    case NBODY_INFO_NSEG: *value = g.nseg; break;
    case NBODY_INFO_DEVICE: *value = L.device; break;
    case NBODY_INFO_CU_COUNT: *value = g.cu_count; break;
    case NBODY_INFO_CLOCK_KHZ: *value = g.clock_khz; break;
    case NBODY_INFO_FP64: *value = g.fp64; break;
    case NBODY_INFO_TILE: *value = g.tile; break;
    case NBODY_INFO_STEPS_DONE: *value = g.steps_done; break;
    case NBODY_INFO_SUM_ORDER: *value = g.fp64 ? NBODY_SUM_SEQ : g.opt.sum_order; break;
    case NBODY_INFO_SUM_BLOCK: *value = (!g.fp64 && g.opt.sum_order == NBODY_SUM_BLOCKED) ? g.opt.sum_block : 0; break;
    case NBODY_INFO_LAUNCHES_PER_STEP: {
      const int force = g.nranks == 1 ? 1 : (g.opt.overlap == 2 ? g.nranks : (g.opt.overlap ? 2 : 1));
      *value = force + (finish_mode() == kFinishStore ? 1 : 0);
      break;
    }
    case NBODY_INFO_HAS_COMM: *value = L.comm_h ? 1 : 0; break;
    default: return NBODY_ERR_ARG;
  }
  return NBODY_OK;
}

const char* nbody_error_string(int code) {
  static char buf[160];
  switch (code) {
    case NBODY_OK: return "ok";
    case NBODY_ERR_NOT_INIT: return "nbody: not initialised";
    case NBODY_ERR_ARG: return "nbody: bad argument";
    case NBODY_ERR_NO_DEVICE: return "nbody: no usable HIP device (this library has no CPU path)";
    case NBODY_ERR_RCCL_LOAD: return "nbody: could not load librccl.so.1";
    case NBODY_ERR_STATE: return "nbody: wrong state for this call";
    case NBODY_ERR_UNSUPPORTED: return "nbody: not supported in this configuration";
    default: break;
  }
  if (code > 0 && code < 1000) { snprintf(buf, sizeof(buf), "HIP error %d (%s) near nbody_hip.hip:%d", code, hipGetErrorString((hipError_t)code), g_last_line); return buf; }
  if (code >= 2000) { snprintf(buf, sizeof(buf), "RCCL error %d near nbody_hip.hip:%d", code - 2000, g_last_line); return buf; }
  snprintf(buf, sizeof(buf), "nbody: unknown error %d", code);
  return buf;
}

int nbody_download_slice(void* pos_words, void* vel_words) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!pos_words || !vel_words) return NBODY_ERR_ARG;
  if (g.nlocal != 1) return NBODY_ERR_UNSUPPORTED;      // one process driving several devices owns every slice: nbody_download
  NBC(sync_all());
  Local& L = g.loc[0];
  const size_t wb = word_bytes();
  HIPC(hipSetDevice(L.device));
  HIPC(hipMemcpy(pos_words, word_ptr(L.pos[L.cur], L.first), (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(vel_words, L.vel, (size_t)L.n_local * wb, hipMemcpyDeviceToHost));
  return NBODY_OK;
}

int nbody_upload(const BodySystem* host) { if (!host) return NBODY_ERR_ARG; if (g.init && g.fp64) return NBODY_ERR_STATE; return upload_impl(host->pos, host->vel); }
int nbody_download(BodySystem* host) { if (!host) return NBODY_ERR_ARG; if (g.init && g.fp64) return NBODY_ERR_STATE; return download_impl(host->pos, host->vel); }
int nbody_upload_d(const BodySystemD* host) { if (!host) return NBODY_ERR_ARG; if (g.init && !g.fp64) return NBODY_ERR_STATE; return upload_impl(host->pos, host->vel); }
int nbody_download_d(BodySystemD* host) { if (!host) return NBODY_ERR_ARG; if (g.init && !g.fp64) return NBODY_ERR_STATE; return download_impl(host->pos, host->vel); }

int bodyForce(float* pos, float* vel, float dt, int n) { if (g.init && g.fp64) return NBODY_ERR_STATE; return body_force_impl(pos, vel, dt, (double)dt, n); }
int integrate(float* pos, const float* vel, float dt, int n) { if (g.init && g.fp64) return NBODY_ERR_STATE; return integrate_impl(pos, vel, dt, (double)dt, n); }
int bodyForce_d(double* pos, double* vel, double dt, int n) { if (g.init && !g.fp64) return NBODY_ERR_STATE; return body_force_impl(pos, vel, (float)dt, dt, n); }
int integrate_d(double* pos, const double* vel, double dt, int n) { if (g.init && !g.fp64) return NBODY_ERR_STATE; return integrate_impl(pos, vel, (float)dt, dt, n); }

int nbody_step(float dt, int nsteps) { if (g.init && g.fp64) return NBODY_ERR_STATE; return step_impl(dt, (double)dt, nsteps); }
int nbody_step_d(double dt, int nsteps) { if (g.init && !g.fp64) return NBODY_ERR_STATE; return step_impl((float)dt, dt, nsteps); }
int nbody_sync(void) { if (!g.init) return NBODY_ERR_NOT_INIT; return sync_all(); }

int nbody_forces(const float* pos_words, float* force_words, int n) { if (g.init && g.fp64) return NBODY_ERR_STATE; return forces_impl(pos_words, force_words, n); }
int nbody_forces_d(const double* pos_words, double* force_words, int n) { if (g.init && !g.fp64) return NBODY_ERR_STATE; return forces_impl(pos_words, force_words, n); }

int nbody_forces_rows(int first_row, int n_rows, float* force_words) { if (g.init && g.fp64) return NBODY_ERR_STATE; return forces_rows_impl(first_row, n_rows, force_words); }
int nbody_forces_rows_d(int first_row, int n_rows, double* force_words) { if (g.init && !g.fp64) return NBODY_ERR_STATE; return forces_rows_impl(first_row, n_rows, force_words); }

int nbody_mailbox_run(void* ram_a, void* ram_b, int clock_khz) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || !ram_a || !ram_b) return NBODY_ERR_ARG;
  // word 0: bit 0 BEGIN, bits [46:32] NUM_PTS                         S/top_level.vhd:184-185
  uint32_t* w0 = (uint32_t*)ram_a;
  if (!(w0[0] & 1u)) return NBODY_ERR_STATE;
  const int num_pts = (int)(w0[1] & 0x7FFFu);
  if (num_pts != g.n) return NBODY_ERR_ARG;
  EventPair ev;   // destroyed on every exit
  HIPC(hipSetDevice(g.loc[0].device));
  HIPC(hipEventCreate(&ev.e0)); HIPC(hipEventCreate(&ev.e1));
  HIPC(hipEventRecord(ev.e0, g.loc[0].compute));
  // bodies are words 1..N                                              S/top_level.vhd:55, 206-208
  NBC(forces_impl((const float*)ram_a + 4, (float*)ram_b, num_pts));
  HIPC(hipEventRecord(ev.e1, g.loc[0].compute));
  HIPC(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  // completion: word 0 <- {ticks in [63:32], 0 elsewhere}: BEGIN reads 0  S/top_level.vhd:146, 255-263
  // one tick = 1000 clocks (S/top_level.vhd:121-144); the counter starts at 1 on BEGIN's rising edge (:138-139)
  const double khz = clock_khz > 0 ? (double)clock_khz : 300000.0;
  uint32_t ticks = 1u + (uint32_t)((double)ms * khz / 1000.0);
  w0[0] = 0; w0[1] = ticks; w0[2] = 0; w0[3] = 0;
  return NBODY_OK;
}

int nbody_set_host_gather(nbody_host_gather_fn fn, void* user) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!g.multiprocess) return NBODY_ERR_STATE;
  g.host_gather = (host_gather_fn)fn;
  g.host_gather_user = user;
  return NBODY_OK;
}

int nbody_kernel_time(double* ms_total, long long* launches, int reset) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  double ms = 0.0; long long n = 0;
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    NBC(timer_drain(L, 0));
    ms = std::max(ms, L.t_ms);   // locals run concurrently: report the slowest device
    n += L.t_launches;
    if (reset) { L.t_ms = 0.0; L.t_launches = 0; }
  }
  if (ms_total) *ms_total = ms;
  if (launches) *launches = n;
  return NBODY_OK;
}

int nbody_device_ptr(int which, void** ptr, size_t* bytes) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!ptr) return NBODY_ERR_ARG;
  Local& L = g.loc[0];
  const size_t wb = word_bytes();
  switch (which) {
    case 0: *ptr = L.pos[L.cur]; if (bytes) *bytes = (size_t)g.n * wb; break;
    case 1: *ptr = L.vel; if (bytes) *bytes = (size_t)L.n_local * wb; break;
    case 2: *ptr = L.force; if (bytes) *bytes = (size_t)L.n_local * wb; break;
    default: return NBODY_ERR_ARG;
  }
  return NBODY_OK;
}

}  // extern "C"
