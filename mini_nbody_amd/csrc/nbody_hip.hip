// nbody_hip.hip — the C-ABI of include/nbody.h on top of the kernels in
// nbody_kernels.hpp.  gfx950 only; no CPU fallback anywhere in this file.
//
// Data layout in HBM (per rank; N bodies in total, the rank owns n_local of
// them starting at first_body):
//   pos[2]   2 x N words      full position set, double-buffered: a step reads pos[cur] and writes the
//                             rank's slice of pos[cur^1]; the other slices of pos[cur^1] arrive over xGMI
//   vel      n_local words    never leaves the rank
//   partial  nseg x rows'     per-source-segment partial forces (unused when nseg == 1); rows' = the launch's rows rounded up to 64
//   tickets  1 per 64 rows    arrival counters of the in-launch combine (zero between steps)
//   force    n_local words    last combined forces (mailbox / parity entry points)
// word = {x,y,z,w}: 16 B (fp32) or 32 B (fp64) — the reference's RAM word, S/top_level.vhd:206-208.
//
// Multi-GPU: bodies are sharded by i; every step each rank
// needs all N positions.  Sources are cut into one slice per rank (x `sub`
// pieces); a step first runs on the rank's own slice while the other slices
// travel (ring of ncclSend/ncclRecv on a second stream, or peer copies when one
// process drives all GPUs), then on the arrived slices.  Partial sums are kept
// per segment and combined in ascending source order — by the last 64-row unit to
// arrive, inside the force launch (finish_rows in nbody_kernels.hpp) — so the
// result is bit-identical for every arrival order and for a single GPU configured
// with the same segmentation (NBODY_OPT_JSLICES, _JSUB, _WSPLIT).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only; the library is resolved with dlopen when nranks > 1
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
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
constexpr int kGraphSteps = 32;    // steps per replayed HIP graph once a call brings at least twice as many

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
  // HIP graph of TWO consecutive steps (the position buffers swap every step, so a pair returns to the same state):
  // replayed by nbody_step when one GPU runs many short steps (launch-bound regime)
  hipGraphExec_t step_graph = nullptr;
  bool stepped_eagerly = false;   // a step has been launched outside a capture since nbody_init
  float graph_dt = 0.f; double graph_dt64 = 0.0; int graph_cur = -1, graph_len = 0;
  bool init = false;
  int n = 0, fp64 = 0, tile = 256;
  int cap = 0;                    // body words the buffers were allocated for (= the n of nbody_init; a mailbox request may bring fewer)
  // the mailbox's two RAMs as the PS sees them (S/top_level.vhd:100-117, 148-163): pinned host memory the device reads (RAM A)
  // and writes (RAM B) itself; allocated on the first request or by nbody_mailbox_open
  void* mb_a = nullptr; void* mb_b = nullptr;
  void* mb_a_dev = nullptr; void* mb_b_dev = nullptr;   // the same memory as the device addresses it
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
};
Global g;

inline size_t word_bytes() { return g.fp64 ? 32 : 16; }
inline char* word_ptr(void* base, size_t word) { return (char*)base + word * word_bytes(); }

inline int rows_per_wg(int R, int wsplit) { return wsplit > 1 ? 64 : kBlock * R; }
int blocks_for(int rows, int R, int wsplit) { const int w = rows_per_wg(R, wsplit); return (rows + w - 1) / w; }
// words between two segments' partial sums: the launch's rows rounded up to 64 (ForceArgs::part_stride)
inline size_t part_stride(int rows) { return ((size_t)rows + 63) / 64 * 64; }

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
  // fp64 strict arithmetic (IEEE sqrt and divide: bit-identical to the oracle) exists in the compiled kernel only
  if (g.fp64 && (g.opt.arith & 2)) g.variant = NBODY_VARIANT_SMEM;
  // the hand-scheduled loops exist for the timed arithmetic only; the study modes use the C++ kernels
  if (g.variant == NBODY_VARIANT_ISA && (g.opt.arith != NBODY_ARITH_FMA3 || g.opt.sum_order == NBODY_SUM_FPGA16)) g.variant = NBODY_VARIANT_SMEM;
  int R = g.opt.iblock;
  if (R == 0) R = (g.variant == NBODY_VARIANT_LDS || g.variant == NBODY_VARIANT_READLANE) ? 2 : 1;
  if (g.variant == NBODY_VARIANT_ISA) R = 1;
  if (g.fp64 && R > 4) R = 4;
  if (g.opt.sum_order == NBODY_SUM_FPGA16 && !g.fp64) R = 1;
  g.R = R;
  // The wave split (ForceArgs::wsplit) exists in the scalar-delivery kernels with one body per lane; the LDS and READLANE
  // deliveries stage sources for the whole workgroup.  (The FPGA order has its own use of 16-wave workgroups, below.)
  const bool fpga32 = !g.fp64 && g.opt.sum_order == NBODY_SUM_FPGA16;
  const bool can_split = (g.variant == NBODY_VARIANT_ISA || g.variant == NBODY_VARIANT_SMEM) && R == 1;
  // automatic: wherever it exists, except for NBODY_SUM_SEQ in fp32, whose meaning is ONE sequential sum per segment (what a CPU
  // nbody.c does); fp64 contexts, which always sum sequentially and have 29 bits to spare, take the split
  const bool auto_split = g.fp64 || g.opt.sum_order != NBODY_SUM_SEQ;
  // ... with 16 waves per workgroup where a rank's bodies fill at most half the CUs with 64-row workgroups (n_local <= 8192):
  // there a step is latency and the 16-wave form needs the fewest global partial sums for the same number of waves (measured
  // per step, profiles/r03_small_n.md: N = 2048 7.5 us against 9.1 with 4 waves, N = 4096 9.7 / 10.2, N = 8192 22.0 / 22.1; from
  // N = 16384 up the two are level in fp32 and 4 waves win by 4 % in fp64, so 4 it is)
  const int cus_ = g.cu_count > 0 ? g.cu_count : 256;
  // (one-rank contexts only: 8 virtual ranks of 8192 bodies each ran 2335 G pairs/s with 16 waves, 2553 with 4)
  const int auto_ws = (!g.fp64 && g.nslices == 1 && (n_local + 63) / 64 <= cus_ / 2) ? 16 : 4;
  g.wsplit = !can_split ? 1 : (g.opt.wsplit == 4 || g.opt.wsplit == 16) ? g.opt.wsplit : (g.opt.wsplit < 0 && auto_split) ? auto_ws : 1;
  // The FPGA order's "split" is of another kind: its sixteen partial sums per row go to the sixteen waves of a workgroup
  // (force_fpga16w_f32) — the same chains, rotation and tree, hence the same bits as one lane holding all sixteen (NBODY_OPT_WSPLIT 1,
  // force_fpga16_f32), with sixteen times the waves: automatic, since the mode's home is N <= 32767 (the mailbox), where one wave per
  // 64 rows leaves the chip empty
  if (fpga32) g.wsplit = (can_split && g.opt.wsplit != 1) ? 16 : 1;
  if (g.wsplit == 16 && g.variant == NBODY_VARIANT_ISA && !g.fp64 && g.opt.isa_phase > 1) g.wsplit = 4;   // diagnostic loop forms: 4 waves
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
  const int blocks = blocks_for(n_local, R, 1);   // in workgroups of 256*R rows: `sub` below counts pieces of a slice as round 2 did
  // "small" = the latency regime: fp32: where the 16-wave workgroups are the automatic choice (n_local <= 8192); between there
  // and N = 16384 round 2's small-launch rule (2 workgroups per CU, combine kernel) measured 21-30 % behind the large-launch one
  // (profiles/r03_sweep_boundary_n*.txt: N = 12288 59.5 us per step against 41.6); fp64, which has no 16-wave regime: never
  // (small-launch rule against large: N = 512 13.7 / 12.8 us per step, 1024: 14.4 / 13.0, 2048: 15.5 / 13.6, 4096: 24.6 / 19.9,
  //  8192: 63.6 / 49.6, 12288: 130.8 / 97.6)
  const bool small = g.fp64 ? false : (n_local + 63) / 64 <= cus / 2;
  int sub = g.opt.jsub;
  if (sub == 0) {
    // workgroups per launch-slice: the step's launches together have 128 (2) per CU whatever the rank count, so that
    // P GPUs see the same segment length as one (N = 1M: 8 pieces per slice for P = 1, 2, 4, 8; two virtual ranks with
    // 2 pieces of 262144 sources ran 2.3 % behind one rank, with 8 pieces level)
    const int target_blocks = std::max(1, (small ? 2 : 128) * cus / g.nslices);
    sub = (target_blocks + blocks - 1) / blocks;
    int slice_len = g.n / g.nslices;
    // keep >= 128 sources per piece of a slice (a wave walks its piece serially); fp64, whose loop takes 4 sources per
    // iteration, >= 64 (N = 4096 fp64: 16 segments x 4 pieces of 64 sources 19.9 us per step, 8 x 4 of 128: 23.1)
    int max_sub = std::max(1, slice_len / (g.fp64 ? 64 : 128));
    sub = std::max(sub, (slice_len + 131071) / 131072);   // and <= 131072 sources (a workgroup's lifetime: the launch's tail)
    // ... unless the partial sums (nseg words per body) would then exceed 16 GiB per rank (288 GB are there to be used): at that
    // size (N >= 32M fp32, 16M fp64) a workgroup's lifetime is a negligible part of a step of minutes anyway; never below 8 segments
    const long long words_cap = (16LL << 30) / (long long)word_bytes() / n_local;
    const int mem_sub = (int)std::max(1LL, std::max(8LL, words_cap) / g.nslices);
    sub = std::min(sub, std::max(mem_sub, (target_blocks + blocks - 1) / blocks));
    sub = std::max(1, std::min(std::min(sub, 64), max_sub));
    // (the FPGA order's sixteen waves are not a split of the segment: its segmentation stays what one lane per body resolves to, so
    //  that NBODY_OPT_WSPLIT changes no bit there with NBODY_OPT_JSUB automatic either)
    if (fpga32) {
    } else if (g.wsplit == 16) {
      // 16-wave workgroups: about one workgroup per CU over the step's launches (N = 4096: 4 segments = 256 workgroups 9.7 us
      // per step, 2: 13.6, 8: 11.9; N = 2048: 4: 7.5, 2: 9.5; N = 8192: 2: 22.0, 4: 23.2, 1: 36.6), pieces of >= 32 sources
      const int blocks64 = (n_local + 63) / 64;
      sub = std::max(1, (cus / g.nslices + blocks64 - 1) / std::max(1, blocks64));
      sub = std::max(1, std::min(sub, slice_len / 512));
    } else if (g.wsplit > 1) {
      // With the wave split a workgroup has a quarter of the rows and its waves a quarter of the segment each: the same
      // number of workgroups and the same walk per wave come from a QUARTER of the global segments (partial sums, tickets,
      // last-arriver rounds).  Two corrections, both measured (profiles/r03_traffic_wsplit.md, r03_sweep_segments_*.txt):
      //  - never fewer segments than keep one inside an XCD's L2 share (2 MiB): N = 1M fp32 stays at 8, one per XCD, each
      //    fetched once — with 4 the positions are re-fetched per resident set (+1.9 GB per step), with 2 every workgroup streams
      //    its 8 MiB from the Infinity Cache (43 GB); N = 4M fp64 on one GPU: 64 segments instead of 8 (4.9 TB per step);
      //  - a P-rank job halves instead of quartering: its launches are P times smaller and want the finer grain (8 virtual
      //    ranks at N = 1M: 1 / 2 / 4 segments per slice 4631 / 4643 / 4661 G pairs/s, one rank 4660)
      const long long slice_bytes = (long long)slice_len * (long long)word_bytes();
      const int l2_sub = (int)std::min<long long>(std::min(64, std::max(1, mem_sub)), (slice_bytes + (2LL << 20) - 1) / (2LL << 20));
      const int div = g.nslices > 1 ? 2 : g.wsplit;
      sub = std::max((sub + div - 1) / div, l2_sub);
      // ... and a body never has more than 64 partial sums (8 virtual ranks at N = 262144 with 16 segments per slice, 128 in
      // all, ran 3.2 % behind one rank)
      if (g.nslices > 1) sub = std::max(1, std::min(sub, std::max(l2_sub, 64 / g.nslices)));
    }
  }
  g.sub = sub;
  g.nseg = g.nslices * g.sub;
  g.fuse = g.opt.fuse < 0 ? (small ? 0 : 1) : g.opt.fuse;
}

// arrival counters: one per 64 rows (a wave's rows), with slack for the row blocks of 256*R rows whose waves count in
// strides of 4*R, padded to a multiple of 256 bytes
size_t ticket_words(const Local& L) { return ((size_t)(L.n_local + 63) / 64 + 32 + 63) / 64 * 64; }

int alloc_local(Local& L) {
  HIPC(hipSetDevice(L.device));
  HIPC(hipStreamCreateWithFlags(&L.compute, hipStreamNonBlocking));
  {
    // The transfers' kernels (RCCL) and copies are small and the force launch beside them fills every wave slot of every
    // CU: the second stream gets the highest priority the device offers, so that its work is dispatched ahead of the
    // force kernel's remaining workgroups (profiles/r03_comm_under_load.md).  NBODY_COMM_PRIORITY=0 turns it off (A/B).
    int least = 0, greatest = 0;
    const char* pe = getenv("NBODY_COMM_PRIORITY");
    const bool want = !(pe && *pe && atoi(pe) == 0);
    if (want && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      HIPC(hipStreamCreateWithPriority(&L.comm, hipStreamNonBlocking, greatest));
      g.comm_priority = greatest;
    } else {
      (void)hipGetLastError();
      HIPC(hipStreamCreateWithFlags(&L.comm, hipStreamNonBlocking));
      g.comm_priority = 0;
    }
  }
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
  HIPC(hipEventCreateWithFlags(&L.ev_comm_go, hipEventDisableTiming));
  for (int s = 0; s < g.nranks && s < kMaxRanks; ++s) HIPC(hipEventCreateWithFlags(&L.ev_gather[s], hipEventDisableTiming));
  for (EventTimer* T : {&L.kern, &L.wait})
    for (int k = 0; k < kTimerRing; ++k) { HIPC(hipEventCreate(&T->t0[k])); HIPC(hipEventCreate(&T->t1[k])); }
  return NBODY_OK;
}

void drop_step_graph() {
  if (g.step_graph) { (void)hipGraphExecDestroy(g.step_graph); g.step_graph = nullptr; }
  g.graph_cur = -1;
}

int ensure_partial(Local& L) {
  const size_t need = (size_t)g.nseg * part_stride(L.n_local);
  if (L.partial && need <= L.partial_words) return NBODY_OK;
  HIPC(hipSetDevice(L.device));
  drop_step_graph();   // captured launches hold the old buffer's address (a mailbox request of another size may be what grows it)
  if (L.partial) { HIPC(hipFree(L.partial)); L.partial = nullptr; L.partial_words = 0; }
  HIPC(hipMalloc(&L.partial, (need + 64) * word_bytes()));
  L.partial_words = need;
  return NBODY_OK;
}

int reconfigure() {
  const int o_variant = g.variant, o_R = g.R, o_sub = g.sub, o_nsl = g.nslices, o_fuse = g.fuse, o_ws = g.wsplit;
  resolve_config();
  // (a step that failed after some of its launches leaves arrival counters at a partial count: the next step would combine
  //  early.  Every failing path sets tickets_dirty; the counters are re-zeroed here, before anything else is launched.)
  const bool changed = o_variant != g.variant || o_R != g.R || o_sub != g.sub || o_nsl != g.nslices || o_fuse != g.fuse || o_ws != g.wsplit ||
                       g.tickets_dirty;
  g.tickets_dirty = false;
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

// ---- event timers ----
int timer_drain(EventTimer& T, int keep) {
  while (T.count > keep) {
    int idx = (T.head - T.count + 2 * kTimerRing) % kTimerRing;
    HIPC(hipEventSynchronize(T.t1[idx]));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, T.t0[idx], T.t1[idx]));
    T.ms += ms;
    T.n += 1;
    T.count--;
  }
  return NBODY_OK;
}
// begin/end of one timed span on `stream`; begin returns the slot (or -1 when timing is off)
int timer_begin(EventTimer& T, hipStream_t stream, int* slot) {
  *slot = -1;
  if (!g.opt.timing) return NBODY_OK;
  if (T.count == kTimerRing) NBC(timer_drain(T, kTimerRing / 2));
  *slot = T.head;
  HIPC(hipEventRecord(T.t0[*slot], stream));
  return NBODY_OK;
}
int timer_end(EventTimer& T, hipStream_t stream, int slot) {
  if (slot < 0) return NBODY_OK;
  HIPC(hipEventRecord(T.t1[slot], stream));
  T.head = (T.head + 1) % kTimerRing;
  T.count++;
  return NBODY_OK;
}

template <typename K>
int launch_timed(Local& L, K kernel, dim3 grid, const ForceArgs& a) {
  int slot;
  NBC(timer_begin(L.kern, L.compute, &slot));
  // optional occupancy cap: k workgroups (= k waves per SIMD) per CU by giving each 160 KiB / k of dynamic LDS
  size_t dyn_lds = 0;
  if (g.opt.waves_per_simd > 0 && g.opt.waves_per_simd < 8) {
    const size_t static_lds = (g.variant == NBODY_VARIANT_LDS ? (size_t)g.tile * 32 : 0) + (a.wsplit > 1 ? (size_t)(a.wsplit - 1) * 64 * word_bytes() : 0);
    // (a workgroup of WS waves holds WS / 4 wave slots per SIMD: the cap is on workgroups per CU = waves_per_simd / (WS / 4))
    const size_t budget = (size_t)(160 * 1024) / (size_t)g.opt.waves_per_simd;
    if (static_lds + 512 > budget) return NBODY_ERR_ARG;   // the kernel's own LDS (16-wave fp64 join: 30 KiB) already exceeds that share: no such cap exists
    dyn_lds = budget - 512 - static_lds;
    if (dyn_lds > 64 * 1024) HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kernel, grid, dim3(wg_threads(a.wsplit)), dyn_lds, L.compute, a);
  HIPC(hipGetLastError());
  return timer_end(L.kern, L.compute, slot);
}

template <int R, int ARITH>
int launch_f32_RA(Local& L, dim3 grid, const ForceArgs& a) {
  if constexpr (R == 8) {   // 8 bodies per lane only exists for the SMEM variant
    return launch_timed(L, force_smem_f32<R, ARITH, 1>, grid, a);
  } else {
    switch (g.variant) {
      case NBODY_VARIANT_LDS:
        if (g.tile >= 1024) return launch_timed(L, force_lds_f32<R, ARITH, 1024>, grid, a);
        if (g.tile >= 512) return launch_timed(L, force_lds_f32<R, ARITH, 512>, grid, a);
        return launch_timed(L, force_lds_f32<R, ARITH, 256>, grid, a);
      case NBODY_VARIANT_READLANE:
        return launch_timed(L, force_readlane_f32<R, ARITH>, grid, a);
      default:
        if constexpr (R == 1) {
          if (a.wsplit == 16) return launch_timed(L, force_smem_f32<1, ARITH, 16>, grid, a);
          if (a.wsplit == 4) return launch_timed(L, force_smem_f32<1, ARITH, 4>, grid, a);
        }
        return launch_timed(L, force_smem_f32<R, ARITH, 1>, grid, a);
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

// the hand-scheduled fp32 loop in form PH (NBODY_OPT_ISA_PHASE), with or without the wave split
template <int PH>
int launch_isa_f32(Local& L, dim3 grid, const ForceArgs& a) {
  if constexpr (PH <= 1) {   // the 16-wave form exists for the product loop and its placement twin (resolve_config sees to it)
    if (a.wsplit == 16) return launch_timed(L, force_isa_f32<PH, 16>, grid, a);
  }
  return a.wsplit == 4 ? launch_timed(L, force_isa_f32<PH, 4>, grid, a) : launch_timed(L, force_isa_f32<PH, 1>, grid, a);
}
template <int PH>
int launch_isa_f64(Local& L, dim3 grid, const ForceArgs& a) {
  if (a.wsplit == 16) return launch_timed(L, force_isa_f64<PH, 16>, grid, a);
  return a.wsplit == 4 ? launch_timed(L, force_isa_f64<PH, 4>, grid, a) : launch_timed(L, force_isa_f64<PH, 1>, grid, a);
}

// loop forms that exist in the diagnostic build only (make diag): experiment encodings and timing-only forms
inline bool isa_phase_is_diag(int ph) { return ph >= 2; }

// how a launch finishes its rows: directly (one segment), by the last-arriving workgroup, or by combine_kernel
inline int finish_mode() { return g.nseg == 1 ? kFinishDirect : (g.fuse ? kFinishLast : kFinishStore); }

void fill_args(Local& L, ForceArgs& a, int row0, int row_count, const Finish& fin, float dt, double dt64) {
  memset(&a, 0, sizeof(a));
  a.src = L.pos[L.cur];
  a.rows = word_ptr(L.pos[L.cur], (size_t)L.first);
  a.partial = L.partial;
  a.vel = L.vel;
  a.pos_next_rows = word_ptr(L.pos[L.cur ^ 1], (size_t)L.first);
  a.force_out = fin.store_force ? (L.force_dst ? L.force_dst : L.force) : nullptr;
  a.tickets = L.tickets;
  a.n_src = g.n; a.n_rows = L.n_local; a.row0 = row0; a.row_count = row_count;
  a.nslices = g.nslices; a.sub = g.sub; a.nseg = g.nseg;
  a.finish = finish_mode();
  a.do_kick = fin.kick; a.do_drift = fin.drift;
  a.sum_block = (!g.fp64 && g.opt.sum_order == NBODY_SUM_BLOCKED) ? g.opt.sum_block : 0;
  a.fpga16 = g.opt.sum_order == NBODY_SUM_FPGA16;
  a.wsplit = g.wsplit;
  a.part_stride = (int)part_stride(row_count);
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
  dim3 grid(blocks_for(row_count, R, g.wsplit), nsl * g.sub, 1);
  // few waves per SIMD and short pieces: the scalar loads are no longer hidden by other waves
  const int cus = g.cu_count > 0 ? g.cu_count : 256;
  // XCD-aware placement of segments (block_segment): needs a multiple of 8 segment rows in the launch, or 1/2/4 of them and a
  // row-block count the 8 / rows XCDs of a segment can deal evenly.  Automatic: for launches whose source set is larger than one
  // XCD's L2 share (N = 1M on one GPU: sources fetched once per XCD, 477 MB of memory-side traffic per step
  // instead of 788, time level); for smaller ones it measured slower (N = 262144: -2 %, N = 16384: -18 %: the last arrivers
  // of every row block then sit on one XCD, profiles/r02_small_n.md)
  const bool xcd_ok = grid.y % 8 == 0 || ((grid.y == 1 || grid.y == 2 || grid.y == 4) && grid.x % (8 / grid.y) == 0);
  const bool xcd_auto = (long long)blocks_for(row_count, 1, 1) >= 4096;
  a.xcd_map = ((g.opt.xcd_map > 0 || (g.opt.xcd_map < 0 && xcd_auto)) && xcd_ok) ? 1 : 0;
  // (r03, wall clock per step: 16384 waves in the launch (N = 16384) 68.0 us with the long buffers against 69.5, 20480 waves
  //  101.7 / 102.9, 24576 waves 144.2 / 142.2, 32768 waves 249.6 / 246.6: the switch sits at 88 waves per CU)
  a.long_buffers = g.opt.long_buffers < 0 ? ((long long)grid.x * grid.y * (wg_threads(a.wsplit) / 64) < 88LL * cus ? 1 : 0) : g.opt.long_buffers;
  if (g.fp64 && g.variant == NBODY_VARIANT_ISA) {
    if (g.opt.isa_phase == 2) return launch_isa_f64<2>(L, grid, a);
    return g.opt.isa_phase == 0 ? launch_isa_f64<0>(L, grid, a) : launch_isa_f64<1>(L, grid, a);
  }
  if (g.fp64 && (g.opt.arith & 2)) {   // NBODY_ARITH_STRICT / _REFERENCE_STRICT: IEEE 1/sqrt (fp64 has one d2 form: the bit-0 distinction is fp32's)
    switch (R) {
      case 1:
        if (a.wsplit == 16) return launch_timed(L, force_smem_f64<1, 16, 1>, grid, a);
        return a.wsplit == 4 ? launch_timed(L, force_smem_f64<1, 4, 1>, grid, a) : launch_timed(L, force_smem_f64<1, 1, 1>, grid, a);
      case 2: return launch_timed(L, force_smem_f64<2, 1, 1>, grid, a);
      default: return launch_timed(L, force_smem_f64<4, 1, 1>, grid, a);
    }
  }
  if (g.fp64) {
    switch (R) {
      case 1:
        if (a.wsplit == 16) return launch_timed(L, force_smem_f64<1, 16>, grid, a);
        return a.wsplit == 4 ? launch_timed(L, force_smem_f64<1, 4>, grid, a) : launch_timed(L, force_smem_f64<1, 1>, grid, a);
      case 2: return launch_timed(L, force_smem_f64<2, 1>, grid, a);
      default: return launch_timed(L, force_smem_f64<4, 1>, grid, a);
    }
  }
  if (a.fpga16 && a.wsplit == 16) {
    switch (g.opt.arith) {
      case NBODY_ARITH_REFERENCE: return launch_timed(L, force_fpga16w_f32<1>, grid, a);
      case NBODY_ARITH_STRICT: return launch_timed(L, force_fpga16w_f32<2>, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_timed(L, force_fpga16w_f32<3>, grid, a);
      default: return launch_timed(L, force_fpga16w_f32<0>, grid, a);
    }
  }
  if (a.fpga16) {
    switch (g.opt.arith) {
      case NBODY_ARITH_REFERENCE: return launch_timed(L, force_fpga16_f32<1>, grid, a);
      case NBODY_ARITH_STRICT: return launch_timed(L, force_fpga16_f32<2>, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_timed(L, force_fpga16_f32<3>, grid, a);
      default: return launch_timed(L, force_fpga16_f32<0>, grid, a);
    }
  }
  if (g.variant == NBODY_VARIANT_ISA) {   // one body per lane, fast arithmetic only (resolve_config guarantees both)
    if (a.long_buffers) {
      if (a.wsplit == 16) return launch_timed(L, force_isa_long_f32<16>, grid, a);
      return a.wsplit == 4 ? launch_timed(L, force_isa_long_f32<4>, grid, a) : launch_timed(L, force_isa_long_f32<1>, grid, a);
    }
    switch (g.opt.isa_phase) {
      case 0: return launch_isa_f32<0>(L, grid, a);
#ifdef NBODY_DIAG_LOOPS
      case 2: return launch_isa_f32<2>(L, grid, a);     // 2, 9..13, 16..18: the same operations in other encodings (bit-identical)
      case 3: return launch_isa_f32<3>(L, grid, a);     // 3..8, 14, 15: TIMING-ONLY forms, WRONG RESULTS
      case 4: return launch_isa_f32<4>(L, grid, a);
      case 5: return launch_isa_f32<5>(L, grid, a);
      case 6: return launch_isa_f32<6>(L, grid, a);
      case 7: return launch_isa_f32<7>(L, grid, a);
      case 8: return launch_isa_f32<8>(L, grid, a);
      case 9: return launch_isa_f32<9>(L, grid, a);
      case 10: return launch_isa_f32<10>(L, grid, a);
      case 11: return launch_isa_f32<11>(L, grid, a);
      case 12: return launch_isa_f32<12>(L, grid, a);
      case 13: return launch_isa_f32<13>(L, grid, a);
      case 14: return launch_isa_f32<14>(L, grid, a);
      case 15: return launch_isa_f32<15>(L, grid, a);
      case 16: return launch_isa_f32<16>(L, grid, a);
      case 17: return launch_isa_f32<17>(L, grid, a);
      case 18: return launch_isa_f32<18>(L, grid, a);
      case 19: return launch_isa_f32<19>(L, grid, a);
      case 20: return launch_isa_f32<20>(L, grid, a);
#endif
      default: return launch_isa_f32<1>(L, grid, a);
    }
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

inline int ring_slice(int rank, int s) { int q = (rank - s) % g.nranks; return q < 0 ? q + g.nranks : q; }

// ---- the transfer plan of one rank: which words go to / come from whom, in which RCCL group ----
// A pure function of (form, rank, P, N): rccl_gather() executes it, nbody_comm_plan() exports it so that the CPU tests can
// check it (every word received exactly once, pair s of rank r matches pair s of its peer, ragged N) without a GPU, and
// nbody_comm_selftest() runs the plans of P virtual ranks through real ncclSend/ncclRecv on one device.
//   RING (the north_star's form): P-1 groups; group s forwards the slice that arrived in group s-1 (the rank's own at
//     s = 1) to rank+1 and receives slice (rank - s) mod P from rank-1; an event after each group releases that slice.
//   DIRECT: one group of P-1 pairs; pair s sends the own slice to rank+s and receives slice (rank - s) mod P from its owner
//     — one hop over all 7 xGMI links at once.
struct CommOp {
  int group;                       // 1-based RCCL group the pair belongs to
  int send_peer; long long send_first, send_count;   // words [send_first, send_first + send_count) of the array go to send_peer
  int recv_peer; long long recv_first, recv_count;   // words [recv_first, ...) are received from recv_peer
};
inline int ring_slice_of(int rank, int s, int P) { int q = (rank - s) % P; return q < 0 ? q + P : q; }
int comm_plan(int form, int rank, int P, int n, std::vector<CommOp>& ops) {
  ops.clear();
  if (P < 1 || rank < 0 || rank >= P || n < P) return NBODY_ERR_ARG;
  if (form != NBODY_COMM_RING && form != NBODY_COMM_DIRECT) return NBODY_ERR_ARG;
  for (int s = 1; s < P; ++s) {
    CommOp o;
    const int qr = ring_slice_of(rank, s, P);      // the slice this pair brings in
    o.recv_first = slice_first(qr, n, P); o.recv_count = slice_first(qr + 1, n, P) - o.recv_first;
    if (form == NBODY_COMM_RING) {
      const int qs = ring_slice_of(rank, s - 1, P);   // forward what arrived last (own slice at s = 1)
      o.group = s;
      o.send_peer = (rank + 1) % P; o.recv_peer = (rank + P - 1) % P;
      o.send_first = slice_first(qs, n, P); o.send_count = slice_first(qs + 1, n, P) - o.send_first;
    } else {
      o.group = 1;
      o.send_peer = (rank + s) % P; o.recv_peer = qr;
      o.send_first = slice_first(rank, n, P); o.send_count = slice_first(rank + 1, n, P) - o.send_first;
    }
    ops.push_back(o);
  }
  return NBODY_OK;
}

// which form NBODY_COMM_AUTO means (profiles/r03_comm_under_load.md): ONE RCCL kernel per step enqueued ahead of the force
// launch — ncclAllGather (whose algorithm over xGMI is a ring) when the slices are equal, the DIRECT group when they are
// not — rather than P-1 dependent ring groups, each of which would have to win wave slots from a force kernel that fills
// every CU.  NBODY_COMM_RING remains the north_star's literal form, one event per arriving slice (NBODY_OPT_OVERLAP 2).
inline int resolved_comm_form() {
  const bool even = (g.n % g.nranks) == 0;
  if (g.opt.comm == NBODY_COMM_AUTO) return even ? NBODY_COMM_ALLGATHER : NBODY_COMM_DIRECT;
  if (g.opt.comm == NBODY_COMM_ALLGATHER && !even) return NBODY_COMM_RING;
  return g.opt.comm;
}

// ncclGroupStart ... ncclGroupEnd with the end guaranteed on every way out (an error between the two must not leave the
// library inside an open group)
struct RcclGroup {
  bool open = false;
  int begin() { NCCLC(g_rccl.GroupStart()); open = true; return NBODY_OK; }
  int end() { open = false; NCCLC(g_rccl.GroupEnd()); return NBODY_OK; }
  ~RcclGroup() { if (open) (void)g_rccl.GroupEnd(); }
};

// one RCCL group of a plan: every send and receive of group `grp`, on the comm stream
int run_plan_group(Local& L, void* dev_full, const std::vector<CommOp>& ops, int grp) {
  const size_t wb = word_bytes();
  RcclGroup grpguard;
  NBC(grpguard.begin());
  for (const CommOp& o : ops) {
    if (o.group != grp) continue;
    NCCLC(g_rccl.Send(word_ptr(dev_full, (size_t)o.send_first), (size_t)o.send_count * wb, ncclChar, o.send_peer, L.comm_h, L.comm));
    NCCLC(g_rccl.Recv(word_ptr(dev_full, (size_t)o.recv_first), (size_t)o.recv_count * wb, ncclChar, o.recv_peer, L.comm_h, L.comm));
  }
  return grpguard.end();
}

// One ring step on the comm stream: send `send_bytes` at `send_ptr` to the next rank, receive `recv_bytes` at `recv_ptr`
// from the previous one, as one RCCL group (so neither side blocks the other).  With one rank next = prev = self and the
// pair is a device-local copy through RCCL (nbody_comm_selftest, nbody_comm_probe on a one-GPU box).
int ring_step(Local& L, const void* send_ptr, size_t send_bytes, void* recv_ptr, size_t recv_bytes) {
  const int P = g.nranks;
  const int next = (L.rank + 1) % P, prev = (L.rank + P - 1) % P;
  RcclGroup grpguard;
  NBC(grpguard.begin());
  NCCLC(g_rccl.Send(send_ptr, send_bytes, ncclChar, next, L.comm_h, L.comm));
  NCCLC(g_rccl.Recv(recv_ptr, recv_bytes, ncclChar, prev, L.comm_h, L.comm));
  return grpguard.end();
}

// RCCL all-gather of one sharded device array in place on the comm stream (multi-process), in the resolved form:
// one in-place ncclAllGather (equal slices), or the plan above group by group.  ev[s] (s = 1..P-1), if given, is
// recorded as soon as ring slice s has landed (RING: after its group, so the force kernel over it can start while the
// next one travels; the single-kernel forms: all after the collective).
int rccl_gather(Local& L, void* dev_full, hipEvent_t* ev) {
  const int P = g.nranks;
  const size_t wb = word_bytes();
  const int form = resolved_comm_form();
  if (form == NBODY_COMM_ALLGATHER) {
    NCCLC(g_rccl.AllGather(word_ptr(dev_full, L.first), dev_full, (size_t)L.n_local * wb, ncclChar, L.comm_h, L.comm));
    if (ev) for (int s = 1; s < P; ++s) HIPC(hipEventRecord(ev[s], L.comm));
    return NBODY_OK;
  }
  std::vector<CommOp> ops;
  NBC(comm_plan(form, L.rank, P, g.n, ops));
  const int groups = ops.empty() ? 0 : ops.back().group;
  for (int grp = 1; grp <= groups; ++grp) {
    NBC(run_plan_group(L, dev_full, ops, grp));
    if (ev && form == NBODY_COMM_RING) HIPC(hipEventRecord(ev[grp], L.comm));
  }
  if (ev && form != NBODY_COMM_RING) for (int s = 1; s < P; ++s) HIPC(hipEventRecord(ev[s], L.comm));
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
  // "the transfer stream has got this far": the own-slice force launch waits for it (enqueue_step), so that the RCCL kernel's
  // packet is at the head of its queue when that launch is released.  Without it both are released by the previous step's
  // end, the force launch wins and fills every wave slot, and the RCCL kernel starts only when that launch has drained:
  // measured on one GPU (profiles/r03_comm_under_load.md) 236 ms after the release without the hand-shake, 0.03 ms with it;
  // the stream's priority changes neither.
  HIPC(hipEventRecord(L.ev_comm_go, L.comm));
  g.comm_go_armed = true;
  return rccl_gather(L, L.pos[buf], L.ev_gather);
}

// the compute stream waits for an arriving slice: timed (NBODY_OPT_TIMING) as exposed communication — the span between
// the moment the stream has nothing else to do and the moment the slice's event fires
int wait_for_slice(Local& L, hipEvent_t ev) {
  int slot;
  NBC(timer_begin(L.wait, L.compute, &slot));
  HIPC(hipStreamWaitEvent(L.compute, ev, 0));
  return timer_end(L.wait, L.compute, slot);
}

int enqueue_step_impl(float dt, double dt64);
// One step on every local: forces on pos[cur], kick, drift into pos[cur^1], swap.
int enqueue_step(float dt, double dt64) {
  const int rc = enqueue_step_impl(dt, dt64);
  if (rc) g.tickets_dirty = true;   // some launches of the step may have run: reconfigure() re-zeroes the arrival counters
  return rc;
}
int enqueue_step_impl(float dt, double dt64) {
  const int P = g.nranks;
  const Finish fin = {true, true, false};
  const bool need_gather = !g.loc[0].all_present;
  if (need_gather && g.opt.overlap) {
    // The own-slice kernels run while the other slices travel on the second stream.  Device-side transports (RCCL,
    // peer copies) are enqueued FIRST: they only wait for the previous step's end, and their small kernels/copies get
    // onto the device ahead of the force launch that fills every CU; the host-staged exchange blocks the host, so
    // there the force launch goes first.
    const bool host_staged = g.multiprocess && g.host_gather;
    g.comm_go_armed = false;
    if (!host_staged) NBC(enqueue_gather(g.loc[0].cur));
    if (g.comm_go_armed) HIPC(hipStreamWaitEvent(g.loc[0].compute, g.loc[0].ev_comm_go, 0));   // RCCL transport: see enqueue_gather
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
        NBC(wait_for_slice(L, L.ev_gather[s]));
        NBC(launch_force(L, 0, L.n_local, ring_slice(L.rank, s), 1, fin, dt, dt64));
      }
    } else if (g.opt.overlap) {
      // the other slices in one launch once they have all arrived (N = 1M, P = 8: 14 MiB of transfers against
      // ~3.7 ms of own-slice work already running; what is not hidden shows up in nbody_comm_time)
      NBC(wait_for_slice(L, L.ev_gather[P - 1]));
      NBC(launch_force(L, 0, L.n_local, ring_slice(L.rank, 1), P - 1, fin, dt, dt64));
    } else {
      NBC(wait_for_slice(L, L.ev_gather[P - 1]));
      NBC(launch_force(L, 0, L.n_local, L.rank, P, fin, dt, dt64));
    }
    NBC(launch_combine(L, 0, L.n_local, fin, dt, dt64));
    // "own slice of pos[cur^1] written": what the next step's transfers wait for.  With one rank nothing does, and inside a
    // captured graph the record would be a node between two kernels.
    if (P > 1) HIPC(hipEventRecord(L.ev_own_ready, L.compute));
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
  g.n = n; g.cap = n; g.fp64 = fp64 ? 1 : 0; g.tile = tile;
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
  if (L.ev_comm_go) (void)hipEventDestroy(L.ev_comm_go);
  for (int s = 0; s < kMaxRanks; ++s) if (L.ev_gather[s]) (void)hipEventDestroy(L.ev_gather[s]);
  for (EventTimer* T : {&L.kern, &L.wait})
    for (int k = 0; k < kTimerRing; ++k) { if (T->t0[k]) (void)hipEventDestroy(T->t0[k]); if (T->t1[k]) (void)hipEventDestroy(T->t1[k]); }
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
    // a fresh state starts from clean arrival counters whatever happened before (a failed step leaves them part-counted)
    HIPC(hipMemsetAsync(L.tickets, 0, ticket_words(L) * sizeof(unsigned), L.compute));
    HIPC(hipEventRecord(L.ev_own_ready, L.compute));
    L.all_present = true;
  }
  g.tickets_dirty = false;
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

// forces of the GLOBAL bodies [g0, g0 + count) (count < 0: all), each local for the part that lies in its slice, from
// pos[cur] (made complete first)
int forces_on_device_impl(int g0, int count) {
  NBC(reconfigure());
  NBC(complete_positions());
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    int r0 = 0, cnt = L.n_local;
    if (count >= 0) {
      const int b = std::max(g0, L.first), e = std::min(g0 + count, L.first + L.n_local);
      if (e <= b) continue;
      r0 = b - L.first; cnt = e - b;
    }
    const Finish fin = {false, false, true};
    NBC(launch_force(L, r0, cnt, g.nslices - 1, g.nslices, fin, 0.f, 0.0));
    NBC(launch_combine(L, r0, cnt, fin, 0.f, 0.0));
  }
  return sync_all();
}
int forces_on_device(int g0, int count) {
  const int rc = forces_on_device_impl(g0, count);
  if (rc) g.tickets_dirty = true;
  return rc;
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
    // steps per graph: an even number (the position buffers swap every step, so an even count returns to the same state).
    // Every graph launch is a boundary of its own on the queue and a host call, and a step at N = 4096 is only 10 us:
    // measured per step (profiles/r03_small_n.md) N = 1024: 11.0 us with 2 steps per graph, 9.2 with 8, 8.8 with 64;
    // N = 4096: 12.5 / 10.7 / 10.2; N = 16384: 71.9 / 69.6 / 69.1.  NBODY_OPT_GRAPH = 1: 32 steps per graph when the call
    // brings >= 64, 16 from 32, 8 from 16, else 2; k >= 2: k steps per graph.
    const int left = nsteps - s;
    int len = g.opt.graph >= 2 ? (g.opt.graph & ~1) : (left >= 2 * kGraphSteps ? kGraphSteps : left >= kGraphSteps ? kGraphSteps / 2 : left >= kGraphSteps / 2 ? kGraphSteps / 4 : 2);
    if (len > nsteps - s) len = (nsteps - s) & ~1;
    if (len >= 2 && (!g.step_graph || g.graph_len != len || g.graph_cur != L.cur || g.graph_dt != dt || g.graph_dt64 != dt64)) {
      drop_step_graph();
      hipGraph_t graph = nullptr;
      const long long done = g.steps_done;
      const int cur0 = L.cur;
      const bool present0 = L.all_present;
      HIPC(hipStreamBeginCapture(L.compute, hipStreamCaptureModeThreadLocal));
      int rc = 0;
      for (int k = 0; k < len && !rc; ++k) rc = enqueue_step(dt, dt64);
      hipError_t e = hipStreamEndCapture(L.compute, &graph);
      g.steps_done = done;                       // capturing executes nothing
      L.cur = cur0; L.all_present = present0;    // an even number of steps returns to the same buffer; a failed capture may have toggled
      g.tickets_dirty = false;                   // ... and has launched nothing
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); g_last_line = __LINE__; return (int)e; }
      e = hipGraphInstantiate(&g.step_graph, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIPC(e);
      g.graph_cur = L.cur; g.graph_dt = dt; g.graph_dt64 = dt64; g.graph_len = len;
    }
    if (len >= 2) for (; s + len <= nsteps; s += len) { HIPC(hipGraphLaunch(g.step_graph, L.compute)); g.steps_done += len; }
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
    int rc = launch_force(L, 0, L.n_local, g.nslices - 1, g.nslices, fin, dt, dt64);
    if (!rc) rc = launch_combine(L, 0, L.n_local, fin, dt, dt64);
    if (rc) { g.tickets_dirty = true; return rc; }
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

// first_row: nbody_init contexts (one process, one or several devices): GLOBAL body index, the range may span devices;
// nbody_init_rank contexts: row of this rank's own slice.
int forces_rows_impl(int first_row, int n_rows, void* force_words) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!force_words) return NBODY_ERR_ARG;
  const int base = g.multiprocess ? g.loc[0].first : 0;
  const int limit = g.multiprocess ? g.loc[0].n_local : g.n;
  if (first_row < 0 || n_rows <= 0 || first_row + n_rows > limit) return NBODY_ERR_ARG;
  const int g0 = base + first_row;
  NBC(forces_on_device(g0, n_rows));
  const size_t wb = word_bytes();
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    const int b = std::max(g0, L.first), e = std::min(g0 + n_rows, L.first + L.n_local);
    if (e <= b) continue;
    HIPC(hipSetDevice(L.device));
    HIPC(hipMemcpy((char*)force_words + (size_t)(b - g0) * wb, word_ptr(L.force, b - L.first), (size_t)(e - b) * wb, hipMemcpyDeviceToHost));
  }
  return NBODY_OK;
}

// ---- the reference's mailbox (S/top_level.vhd:176-272) ----
// One context serves requests of ANY NUM_PTS up to its capacity, as the RTL samples NUM_PTS with every BEGIN (:180-186) against a RAM
// sized once (:45).  The buffers are sized for the capacity; a request switches N and the launch configuration for its own duration
// (resolve_config is host arithmetic) and leaves the context's N, state options and captured step graph as they were.
constexpr int kMailboxMaxPoints = 32767;   // ram_depth - 1, S/top_level.vhd:45

int mailbox_rams() {   // RAM A: capacity + 1 words, RAM B: capacity words (+ slack), pinned, mapped, coherent
  if (g.mb_a && g.mb_b) return NBODY_OK;
  Local& L = g.loc[0];
  HIPC(hipSetDevice(L.device));
  const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
  if (!g.mb_a) { HIPC(hipHostMalloc(&g.mb_a, ((size_t)g.cap + 1 + 64) * 16, flags)); memset(g.mb_a, 0, ((size_t)g.cap + 1 + 64) * 16); }
  if (!g.mb_b) { HIPC(hipHostMalloc(&g.mb_b, ((size_t)g.cap + 64) * 16, flags)); memset(g.mb_b, 0, ((size_t)g.cap + 64) * 16); }
  HIPC(hipHostGetDevicePointer(&g.mb_a_dev, g.mb_a, 0));
  HIPC(hipHostGetDevicePointer(&g.mb_b_dev, g.mb_b, 0));
  return NBODY_OK;
}

// N and the launch configuration of a one-rank context switched for the duration of one request
struct ActiveN {
  bool armed = false;
  int n = 0, n_local = 0, variant = 0, R = 0, sub = 0, nslices = 0, nseg = 0, fuse = 0, wsplit = 0;
  int enter(int n_new) {
    Local& L = g.loc[0];
    n = g.n; n_local = L.n_local; variant = g.variant; R = g.R; sub = g.sub; nslices = g.nslices; nseg = g.nseg; fuse = g.fuse; wsplit = g.wsplit;
    armed = true;
    g.n = n_new; L.n_local = n_new;
    resolve_config();
    NBC(ensure_partial(L));
    if (g.tickets_dirty) {   // a failed launch sequence left arrival counters part-counted (they are zero between requests otherwise)
      HIPC(hipMemsetAsync(L.tickets, 0, ((size_t)(g.cap + 63) / 64 + 32 + 63) / 64 * 64 * sizeof(unsigned), L.compute));
      g.tickets_dirty = false;
    }
    return NBODY_OK;
  }
  ~ActiveN() {
    if (!armed) return;
    Local& L = g.loc[0];
    g.n = n; L.n_local = n_local; g.variant = variant; g.R = R; g.sub = sub; g.nslices = nslices; g.nseg = nseg; g.fuse = fuse; g.wsplit = wsplit;
  }
};

// completion of everything on `stream`: polled for the first 200 us (a request at the mailbox's sizes takes 7-500 us of device time and an
// interrupt-driven wait adds tens of us of wake-up), then a blocking wait
int wait_stream(hipStream_t stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipStreamQuery(stream);
    if (e == hipSuccess) return NBODY_OK;
    if (e != hipErrorNotReady) { g_last_line = __LINE__; return (int)e; }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
  }
  HIPC(hipStreamSynchronize(stream));
  return NBODY_OK;
}

// the launches of one request on the compute stream: RAM A's read port, the force pass storing into RAM B (and its combine)
int mailbox_launches(Local& L, int num_pts) {
  // bodies are words 1..N                                              S/top_level.vhd:55, 206-208
  hipLaunchKernelGGL(ingest_kernel, dim3((num_pts + kBlock - 1) / kBlock), dim3(kBlock), 0, L.compute,
                     (f4*)L.pos[L.cur], (const f4*)((const char*)g.mb_a_dev + 16), num_pts);
  HIPC(hipGetLastError());
  // RAM B's write port: the force launch (or its combine) stores {Fx, Fy, Fz, 0} of body k at word k-1 itself, words >= N are never
  // written                                                             S/compute_store.vhd:213, 227-242
  const Finish fin = {false, false, true};
  L.force_dst = g.mb_b_dev;
  int rc = launch_force(L, 0, num_pts, g.nslices - 1, g.nslices, fin, 0.f, 0.0);
  if (!rc) rc = launch_combine(L, 0, num_pts, fin, 0.f, 0.0);
  L.force_dst = nullptr;
  return rc;
}

int mailbox_request(const void* ram_a, void* ram_b, int num_pts) {
  Local& L = g.loc[0];
  HIPC(hipSetDevice(L.device));
  NBC(mailbox_rams());
  ActiveN scope;
  NBC(scope.enter(num_pts));
  // RAM A: the library's own pinned image is read in place; any other host buffer is copied into it first
  if (ram_a != g.mb_a) memcpy((char*)g.mb_a + 16, (const char*)ram_a + 16, (size_t)num_pts * 16);
  L.all_present = true;
  // (replaying the request's launches from a captured HIP graph was measured in round 5 and not kept: 21.1 against 23.7 us at N = 9,
  //  29.8 against 30.1 at N = 1024, level above — gpurun_out/r05/mailbox_rate_b*.txt)
  const int rc = mailbox_launches(L, num_pts);
  if (rc) { g.tickets_dirty = true; return rc; }
  NBC(wait_stream(L.compute));
  if (ram_b != g.mb_b) memcpy(ram_b, g.mb_b, (size_t)num_pts * 16);
  return NBODY_OK;
}


// One request from the RAM images (the body of nbody_mailbox_run and of the service thread).  `served`: an error has no return value to
// travel in, so it is written into word 0 (bits 127:96, which the RTL always writes as 0) with BEGIN cleared.
int mailbox_run_impl(void* ram_a, void* ram_b, int clock_khz, bool served) {
  const auto t0 = std::chrono::steady_clock::now();
  // word 0: bit 0 BEGIN, bits [46:32] NUM_PTS, sampled with every request        S/top_level.vhd:180-186
  uint32_t* w0 = (uint32_t*)ram_a;
  if (!(w0[0] & 1u)) return NBODY_ERR_STATE;   // the FSM stays in `waiting`: nothing is read, nothing is written
  const int num_pts = (int)(w0[1] & 0x7FFFu);
  int rc = NBODY_OK;
  if (g.nranks == 1) {
    if (num_pts > g.cap) rc = NBODY_ERR_ARG;   // (the RTL's RAM always holds 32767 bodies; a smaller capacity is this library's notion)
    // NUM_PTS = 0: block_setup finds THIS_PTR > NUM_PTS at once and goes to `complete` (S/top_level.vhd:189-192): RAM B untouched
    else if (num_pts > 0) rc = mailbox_request(ram_a, ram_b, num_pts);
  } else {
    // a context over several devices / ranks keeps its fixed N: every rank brings the same images (nbody_forces)
    if (num_pts != g.n) rc = NBODY_ERR_ARG;
    else rc = forces_impl((const float*)ram_a + 4, (float*)ram_b, num_pts);
  }
  if (rc && !served) return rc;
  // completion: word 0 <- {ticks in [63:32], 0 elsewhere}: BEGIN reads 0          S/top_level.vhd:146, 255-263
  // one tick = 1000 clocks (S/top_level.vhd:121-144); the counter goes to 1 on BEGIN's rising edge (:138-139); BEGIN-to-done as this
  // host sees it (the device's reads of RAM A and writes of RAM B included)
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const double khz = clock_khz > 0 ? (double)clock_khz : 300000.0;
  const uint32_t ticks = rc ? 0u : 1u + (uint32_t)(ms * khz / 1000.0);
  w0[1] = ticks; w0[2] = 0; w0[3] = (uint32_t)rc;
  __atomic_store_n(&w0[0], 0u, __ATOMIC_RELEASE);   // BEGIN is cleared LAST: whoever sees it cleared sees the ticks and RAM B
  return rc;
}

// The PL block serves the PS without being called: its FSM samples word 0 of RAM A every clock (S/top_level.vhd:180-186).  The same on
// a host: a library thread polls word 0 of the context's own RAM A, runs every request it finds and rewrites word 0 — the driver only
// writes and reads memory.  Idle polling backs off: `pause` for the first ~ms, then yields, then 50-us naps after ~0.1 s without work.
std::thread g_serve_thread;
std::atomic<int> g_serve_on{0};
std::atomic<long long> g_served{0};
int g_serve_khz = 0;

void serve_loop() {
  uint32_t* w0 = (uint32_t*)g.mb_a;
  unsigned idle = 0;
  while (g_serve_on.load(std::memory_order_acquire)) {
    if (!(__atomic_load_n(&w0[0], __ATOMIC_ACQUIRE) & 1u)) {
      ++idle;
      if (idle < 20000) __builtin_ia32_pause();
      else if (idle < 400000) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(50));
      continue;
    }
    idle = 0;
    (void)mailbox_run_impl(g.mb_a, g.mb_b, g_serve_khz, true);
    g_served.fetch_add(1, std::memory_order_relaxed);
  }
}

void serve_stop() {
  g_serve_on.store(0, std::memory_order_release);
  if (g_serve_thread.joinable()) g_serve_thread.join();
}
// a process that exits while the thread serves (no nbody_shutdown): stop and join it before g_serve_thread is destroyed — a joinable
// std::thread reaching its destructor ends the process with std::terminate (declared after the thread, hence destroyed before it)
struct ServeGuard { ~ServeGuard() { serve_stop(); } } g_serve_guard;

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

// The transfer plans of `vp` VIRTUAL ranks (an N-body job of vp ranks over g.n bodies, ragged slices included) executed
// through real ncclSend/ncclRecv on this one-rank communicator: every virtual rank has its own N-word array on the device
// holding only its own slice; group by group, each receive of each virtual rank is issued together with the send its peer's
// plan pairs with it (same group, send_peer = the receiver) — with one real rank all peers are "self" and RCCL matches
// the k-th send with the k-th receive of a group, so issuing them pairwise reproduces exactly the P-rank exchange.
// Afterwards every array must hold all N words.  This runs the plan's offsets, byte counts and send/recv pairing of both
// forms on hardware, which a one-rank job's own plan (P - 1 = 0 pairs) never does.
int nbody_comm_selftest_virtual(int vp, int form, long long* bytes_moved) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!g.multiprocess || g.nranks != 1) return NBODY_ERR_STATE;
  Local& L = g.loc[0];
  if (!L.comm_h) return NBODY_ERR_STATE;
  if (vp < 2 || vp > 16 || g.n < vp || (form != NBODY_COMM_RING && form != NBODY_COMM_DIRECT)) return NBODY_ERR_ARG;
  NBC(sync_all());
  const size_t wb = word_bytes(), lanes = wb / 4;
  HIPC(hipSetDevice(L.device));
  struct Bufs {   // freed on every way out
    std::vector<void*> d;
    ~Bufs() { for (void* p : d) if (p) (void)hipFree(p); }
  } bufs;
  bufs.d.assign(vp, nullptr);
  std::vector<std::vector<CommOp>> plan(vp);
  std::vector<uint32_t> host((size_t)g.n * lanes);
  auto pattern = [](int w, size_t k) { return 0x5A000000u ^ ((uint32_t)w * 4u + (uint32_t)k) * 2654435761u; };
  for (int r = 0; r < vp; ++r) {
    NBC(comm_plan(form, r, vp, g.n, plan[r]));
    HIPC(hipMalloc(&bufs.d[r], (size_t)(g.n + 64) * wb));
    HIPC(hipMemset(bufs.d[r], 0xff, (size_t)g.n * wb));
    const int f = slice_first(r, g.n, vp), c = slice_first(r + 1, g.n, vp) - f;
    for (int w = f; w < f + c; ++w) for (size_t k = 0; k < lanes; ++k) host[(size_t)w * lanes + k] = pattern(w, k);
    HIPC(hipMemcpy(word_ptr(bufs.d[r], f), &host[(size_t)f * lanes], (size_t)c * wb, hipMemcpyHostToDevice));
  }
  long long moved = 0;
  const int groups = plan[0].empty() ? 0 : plan[0].back().group;
  for (int grp = 1; grp <= groups; ++grp) {
    RcclGroup grpguard;
    NBC(grpguard.begin());
    for (int r = 0; r < vp; ++r) {
      for (const CommOp& o : plan[r]) {
        if (o.group != grp) continue;
        // the send that rank o.recv_peer's plan pairs with this receive: same group, addressed to r, same word range
        const CommOp* snd = nullptr;
        for (const CommOp& q : plan[o.recv_peer])
          if (q.group == grp && q.send_peer == r && q.send_first == o.recv_first && q.send_count == o.recv_count) { snd = &q; break; }
        if (!snd) { g_last_line = __LINE__; return NBODY_ERR_STATE; }   // the plans do not pair up
        NCCLC(g_rccl.Send(word_ptr(bufs.d[o.recv_peer], (size_t)snd->send_first), (size_t)snd->send_count * wb, ncclChar, 0, L.comm_h, L.comm));
        NCCLC(g_rccl.Recv(word_ptr(bufs.d[r], (size_t)o.recv_first), (size_t)o.recv_count * wb, ncclChar, 0, L.comm_h, L.comm));
        moved += o.recv_count * (long long)wb;
      }
    }
    NBC(grpguard.end());
  }
  HIPC(hipStreamSynchronize(L.comm));
  for (int r = 0; r < vp; ++r) {
    HIPC(hipMemcpy(host.data(), bufs.d[r], (size_t)g.n * wb, hipMemcpyDeviceToHost));
    for (int w = 0; w < g.n; ++w)
      for (size_t k = 0; k < lanes; ++k)
        if (host[(size_t)w * lanes + k] != pattern(w, k)) { g_last_line = __LINE__; return NBODY_ERR_STATE; }
  }
  if (bytes_moved) *bytes_moved = moved;
  return NBODY_OK;
}

// The plan of rank `rank` of `nranks` over n bodies in form NBODY_COMM_RING or NBODY_COMM_DIRECT, 7 values per pair:
// {group, send_peer, send_first_word, send_words, recv_peer, recv_first_word, recv_words}.  Pure host arithmetic (no GPU,
// no context): what rccl_gather() executes.  *n_ops = pairs (nranks - 1); ops may be NULL to ask for the count.
int nbody_comm_plan(int form, int rank, int nranks, int n, long long* ops, int max_ops, int* n_ops) {
  std::vector<CommOp> v;
  NBC(comm_plan(form, rank, nranks, n, v));
  if (n_ops) *n_ops = (int)v.size();
  if (!ops) return NBODY_OK;
  if ((int)v.size() > max_ops) return NBODY_ERR_ARG;
  for (size_t k = 0; k < v.size(); ++k) {
    long long* o = ops + 7 * k;
    o[0] = v[k].group; o[1] = v[k].send_peer; o[2] = v[k].send_first; o[3] = v[k].send_count;
    o[4] = v[k].recv_peer; o[5] = v[k].recv_first; o[6] = v[k].recv_count;
  }
  return NBODY_OK;
}

// How long one RCCL ring step of `bytes` (ncclSend to rank+1 / ncclRecv from rank-1, one group) takes on the transfer
// stream beside a force pass that fills every wave slot of every CU.  when =
//   0  alone;                                  *comm_ms = enqueue -> done of the ring step
//   1  enqueued just BEFORE a full force pass;  "
//   2  enqueued just AFTER it;                  "
//   3  the steady state of a multi-GPU step: force pass A, then — both released by A's end — the ring step on the
//      transfer stream and force pass B on the compute stream; *comm_ms = end of A -> ring step done (B's duration when
//      the transfer loses the race for the chip, microseconds when it wins);
//   4  the same with the hand-shake enqueue_step() uses: pass B waits for an event the transfer stream records right
//      before its RCCL kernel (L.ev_comm_go), so the RCCL kernel's packet is at the head of its queue when B is released.
// *force_ms: the duration of the (last) force pass.
int nbody_comm_probe(long long bytes, int when, double* comm_ms, double* force_ms) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!g.multiprocess) return NBODY_ERR_STATE;
  Local& L = g.loc[0];
  if (!L.comm_h) return NBODY_ERR_STATE;
  const size_t wb = word_bytes();
  if (bytes <= 0 || when < 0 || when > 4 || (size_t)bytes * 2 > (size_t)g.n * wb) return NBODY_ERR_ARG;
  NBC(reconfigure());
  NBC(complete_positions());
  NBC(sync_all());
  HIPC(hipSetDevice(L.device));
  if (!L.full_scratch) HIPC(hipMalloc(&L.full_scratch, (size_t)(g.n + 64) * wb));
  struct Evs { hipEvent_t e[6] = {}; ~Evs() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); } } ev;
  for (hipEvent_t& x : ev.e) HIPC(hipEventCreate(&x));
  const Finish fin = {false, false, true};
  auto force_pass = [&](hipEvent_t begin, hipEvent_t end) -> int {
    if (begin) HIPC(hipEventRecord(begin, L.compute));
    int rc = launch_force(L, 0, L.n_local, g.nslices - 1, g.nslices, fin, 0.f, 0.0);
    if (!rc) rc = launch_combine(L, 0, L.n_local, fin, 0.f, 0.0);
    if (rc) { g.tickets_dirty = true; return rc; }
    HIPC(hipEventRecord(end, L.compute));
    return NBODY_OK;
  };
  auto comm_step = [&]() -> int {
    HIPC(hipEventRecord(ev.e[0], L.comm));
    NBC(ring_step(L, L.full_scratch, (size_t)bytes, (char*)L.full_scratch + bytes, (size_t)bytes));
    HIPC(hipEventRecord(ev.e[1], L.comm));
    return NBODY_OK;
  };
  hipEvent_t from = ev.e[0];
  if (when == 1) { NBC(comm_step()); NBC(force_pass(ev.e[2], ev.e[3])); }
  else if (when == 2) { NBC(force_pass(ev.e[2], ev.e[3])); NBC(comm_step()); }
  else if (when >= 3) {
    NBC(force_pass(nullptr, ev.e[4]));                       // pass A; e[4] = "own slice ready"
    HIPC(hipStreamWaitEvent(L.comm, ev.e[4], 0));
    if (when == 4) HIPC(hipEventRecord(L.ev_comm_go, L.comm));
    NBC(comm_step());
    if (when == 4) HIPC(hipStreamWaitEvent(L.compute, L.ev_comm_go, 0));
    NBC(force_pass(ev.e[2], ev.e[3]));                       // pass B
    from = ev.e[4];
  } else NBC(comm_step());
  NBC(sync_all());
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, from, ev.e[1]));
  if (comm_ms) *comm_ms = ms;
  if (force_ms) {
    *force_ms = 0.0;
    if (when) { HIPC(hipEventElapsedTime(&ms, ev.e[2], ev.e[3])); *force_ms = ms; }
  }
  return NBODY_OK;
}

// The strict 1/sqrt (NBODY_ARITH_STRICT, fp32) checked against its own definition on the device that will run it: needs no context.
static int rsqrt_device() {
  if (g.init) return g.loc[0].device;
  int ndev = 0;
  if (device_count(&ndev)) return -1;
  return pick_device(0, ndev);
}
struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } };

static int rsqrt_selftest_on(int dev, unsigned first_bits, unsigned long long count, unsigned long long* res3) {
  HIPC(hipSetDevice(dev));
  DevBuf out;
  HIPC(hipMalloc(&out.p, 3 * sizeof(unsigned long long)));
  const unsigned long long zero[3] = {0, 0, ~0ull};
  HIPC(hipMemcpy(out.p, zero, sizeof(zero), hipMemcpyHostToDevice));
  const unsigned long long wgs = (count + 255) / 256;
  rsqrt_selftest_kernel<<<dim3((unsigned)(wgs < 16384 ? wgs : 16384)), dim3(256)>>>(first_bits, count, (unsigned long long*)out.p);
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(res3, out.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return NBODY_OK;
}

int nbody_rsqrt_selftest(unsigned first_bits, unsigned long long count, unsigned long long* mismatches, unsigned long long* ieee_lanes, unsigned* first_bad) {
  if (count == 0 || count > (1ull << 32) || (unsigned long long)first_bits + count > (1ull << 32)) return NBODY_ERR_ARG;
  const int dev = rsqrt_device();
  if (dev < 0) return NBODY_ERR_NO_DEVICE;
  unsigned long long res[3];
  NBC(rsqrt_selftest_on(dev, first_bits, count, res));
  if (mismatches) *mismatches = res[0];
  if (ieee_lanes) *ieee_lanes = res[1];
  if (first_bad) *first_bad = res[0] ? (unsigned)res[2] : 0u;
  return NBODY_OK;
}

// The proof the strict binary32 arithmetic rests on, per DEVICE and once per process: every positive normal binary32 through the
// eight-operation 1/sqrt and through its IEEE definition on device `dev`, no accepted value differing (about 10 ms).  The library
// runs it itself before it lets a context use NBODY_ARITH_STRICT / _REFERENCE_STRICT (nbody_set_option, nbody_mailbox_open): on
// every device of the context, so that no host layer has to remember it.  NBODY_STRICT_PROOF_FAIL=1 makes it fail (the refusal's test).
static int g_strict_proved[64] = {};   // 0 unknown, 1 proved, -1 refuted
static unsigned long long g_strict_bad = 0; static unsigned g_strict_first_bad = 0;
static int prove_strict_on(int dev) {
  if (dev < 0 || dev >= 64) return NBODY_ERR_ARG;
  if (g_strict_proved[dev] == 0) {
    unsigned long long res[3];
    NBC(rsqrt_selftest_on(dev, 0x00800000u, 0x7F800000ull - 0x00800000ull, res));
    const char* fail = getenv("NBODY_STRICT_PROOF_FAIL");
    if (fail && atoi(fail)) { res[0] = 1; res[2] = 0x00800000u; }
    g_strict_proved[dev] = res[0] ? -1 : 1;
    if (res[0]) { g_strict_bad = res[0]; g_strict_first_bad = (unsigned)res[2]; }
  }
  return g_strict_proved[dev] > 0 ? NBODY_OK : NBODY_ERR_UNSUPPORTED;
}
static int prove_strict_context() {
  for (int l = 0; l < g.nlocal; ++l) NBC(prove_strict_on(g.loc[l].device));
  return NBODY_OK;
}

int nbody_strict_proof(unsigned long long* mismatches, unsigned* first_bad) {
  int rc;
  if (g.init) rc = prove_strict_context();
  else { const int dev = rsqrt_device(); if (dev < 0) return NBODY_ERR_NO_DEVICE; rc = prove_strict_on(dev); }
  if (mismatches) *mismatches = rc == NBODY_ERR_UNSUPPORTED ? g_strict_bad : 0;
  if (first_bad) *first_bad = rc == NBODY_ERR_UNSUPPORTED ? g_strict_first_bad : 0;
  return rc;
}

int nbody_rsqrt_strict(const float* x, float* y, int n, int ieee_only) {
  if (!x || !y || n <= 0) return NBODY_ERR_ARG;
  const int dev = rsqrt_device();
  if (dev < 0) return NBODY_ERR_NO_DEVICE;
  HIPC(hipSetDevice(dev));
  DevBuf dx, dy;
  HIPC(hipMalloc(&dx.p, (size_t)n * sizeof(float)));
  HIPC(hipMalloc(&dy.p, (size_t)n * sizeof(float)));
  HIPC(hipMemcpy(dx.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  rsqrt_array_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>((const float*)dx.p, (float*)dy.p, n, ieee_only ? 1 : 0);
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(y, dy.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return NBODY_OK;
}

void nbody_shutdown(void) {
  serve_stop();   // the mailbox's service thread, if one runs, ends before anything it uses is freed
  drop_step_graph();
  for (int l = 0; l < kMaxLocal; ++l) free_local(g.loc[l]);
  if (g.host_stage) { (void)hipHostFree(g.host_stage); g.host_stage = nullptr; }
  if (g.mb_a) { (void)hipHostFree(g.mb_a); g.mb_a = nullptr; g.mb_a_dev = nullptr; }
  if (g.mb_b) { (void)hipHostFree(g.mb_b); g.mb_b = nullptr; g.mb_b_dev = nullptr; }
  g.host_gather = nullptr; g.host_gather_user = nullptr;
  g.init = false; g.nlocal = 0; g.nranks = 1;
}

int nbody_set_option(int key, int value) {
  switch (key) {
    case NBODY_OPT_VARIANT: if (value < 0 || value > 4) return NBODY_ERR_ARG; g.opt.variant = value; break;
    case NBODY_OPT_IBLOCK: if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return NBODY_ERR_ARG; g.opt.iblock = value; break;
    case NBODY_OPT_JSUB: if (value < 0 || value > 256) return NBODY_ERR_ARG; g.opt.jsub = value; break;
    case NBODY_OPT_JSLICES: if (value < 0 || value > kMaxRanks) return NBODY_ERR_ARG; g.opt.jslices = value; break;
    case NBODY_OPT_ARITH:
      if (value < 0 || value > 3) return NBODY_ERR_ARG;
      // the strict binary32 1/sqrt is used only on devices that have proved it (once per device and process, ~10 ms)
      if ((value & 2) && g.init && !g.fp64) NBC(prove_strict_context());
      g.opt.arith = value; break;
    case NBODY_OPT_SUM_ORDER: if (value < 0 || value > 2) return NBODY_ERR_ARG; g.opt.sum_order = value; break;
    case NBODY_OPT_SUM_BLOCK: if (value < 8 || value > (1 << 24) || value % 64) return NBODY_ERR_ARG; g.opt.sum_block = value; break;
    case NBODY_OPT_FUSE_COMBINE: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.fuse = value; break;
    case NBODY_OPT_ISA_LONG_BUFFERS: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.long_buffers = value; break;
    case NBODY_OPT_XCD_MAP: if (value < -1 || value > 1) return NBODY_ERR_ARG; g.opt.xcd_map = value; break;
    case NBODY_OPT_TIMING: g.opt.timing = value ? 1 : 0; break;
    case NBODY_OPT_COMM: if (value < 0 || value > 3) return NBODY_ERR_ARG; g.opt.comm = value; break;
    case NBODY_OPT_OVERLAP: if (value < 0 || value > 2) return NBODY_ERR_ARG; g.opt.overlap = value; break;
    case NBODY_OPT_GRAPH: if (value < 0 || value > 256) return NBODY_ERR_ARG; g.opt.graph = value; break;
    case NBODY_OPT_WAVES_PER_SIMD: if (value < 0 || value > 8) return NBODY_ERR_ARG; g.opt.waves_per_simd = value; break;
    case NBODY_OPT_ISA_PHASE:
      if (value < 0 || value > 20) return NBODY_ERR_ARG;
#ifndef NBODY_DIAG_LOOPS
      // experiment encodings and timing-only forms (wrong results) are not in the product library: `make diag`
      if (isa_phase_is_diag(value) && !(g.init && g.fp64 && value == 2)) return NBODY_ERR_UNSUPPORTED;
#endif
      g.opt.isa_phase = value; break;
    case NBODY_OPT_WSPLIT: if (value != -1 && value != 1 && value != 4 && value != 16) return NBODY_ERR_ARG; g.opt.wsplit = value; break;
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
    case NBODY_INFO_WSPLIT: *value = g.wsplit; break;
    case NBODY_INFO_ISA_PHASE: *value = g.opt.isa_phase; break;
    case NBODY_INFO_LONG_BUFFERS: *value = g.opt.long_buffers; break;
    case NBODY_INFO_XCD_MAP: *value = g.opt.xcd_map; break;
    case NBODY_INFO_FUSE_COMBINE: *value = g.fuse; break;
    case NBODY_INFO_COMM_FORM: *value = g.nranks > 1 ? resolved_comm_form() : -1; break;
    case NBODY_INFO_COMM_PRIORITY: *value = g.comm_priority; break;
    case NBODY_INFO_MAILBOX_SERVED: *value = g_served.load(std::memory_order_relaxed); break;
    case NBODY_INFO_MAILBOX_SERVING: *value = g_serve_on.load(std::memory_order_acquire); break;
    case NBODY_INFO_DIAG_BUILD:
#ifdef NBODY_DIAG_LOOPS
      *value = 1; break;
#else
      *value = 0; break;
#endif
    default: return NBODY_ERR_ARG;
  }
  return NBODY_OK;
}

const char* nbody_error_string(int code) {
  static char buf[320];
  switch (code) {
    case NBODY_OK: return "ok";
    case NBODY_ERR_NOT_INIT: return "nbody: not initialised";
    case NBODY_ERR_ARG: return "nbody: bad argument";
    case NBODY_ERR_NO_DEVICE: return "nbody: no usable HIP device (this library has no CPU path)";
    case NBODY_ERR_RCCL_LOAD: return "nbody: could not load librccl.so.1";
    case NBODY_ERR_STATE: return "nbody: wrong state for this call";
    case NBODY_ERR_UNSUPPORTED:
      if (g_strict_bad) {
        snprintf(buf, sizeof(buf), "nbody: not supported in this configuration (strict arithmetic refused: the eight-operation 1/sqrt differs from "
                 "(float)(1.0/sqrt((double)x)) for %llu arguments on a device, first 0x%08x)", g_strict_bad, g_strict_first_bad);
        return buf;
      }
      return "nbody: not supported in this configuration";
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

int nbody_mailbox_open(int capacity, int faithful) {
  if (capacity == 0) capacity = kMailboxMaxPoints;
  if (capacity < 1 || capacity > kMailboxMaxPoints) return NBODY_ERR_ARG;
  NBC(nbody_init(capacity, 1, 0, 0));
  int rc = mailbox_rams();
  // the partial sums of the largest segmentation any request can resolve to (64 segments), so that no request allocates
  if (!rc) { const int nseg = g.nseg; g.nseg = 64; rc = ensure_partial(g.loc[0]); g.nseg = nseg; }
  if (!rc && faithful) {
    // the PL block's own bits: its rounding points (S/dxy.vhd:113-122, S/dzsoft.vhd:201-202, S/dxyz_soft.vhd:149-150) with 1/sqrt rounded
    // once — after this device has proved that 1/sqrt —, its sixteen partial sums, rotation and adder tree (S/fxyz.vhd:129-184,
    // S/final_adder.vhd:88-104) over ONE stream of all N sources per body (S/top_level.vhd:233-254)
    rc = nbody_set_option(NBODY_OPT_ARITH, NBODY_ARITH_REFERENCE_STRICT);
    if (!rc) rc = nbody_set_option(NBODY_OPT_SUM_ORDER, NBODY_SUM_FPGA16);
    if (!rc) rc = nbody_set_option(NBODY_OPT_JSUB, 1);
  }
  if (rc) { nbody_shutdown(); return rc; }
  return NBODY_OK;
}

int nbody_mailbox_rams(void** ram_a, void** ram_b, int* capacity) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || g.nranks != 1) return NBODY_ERR_UNSUPPORTED;
  NBC(mailbox_rams());
  if (ram_a) *ram_a = g.mb_a;
  if (ram_b) *ram_b = g.mb_b;
  if (capacity) *capacity = g.cap < kMailboxMaxPoints ? g.cap : kMailboxMaxPoints;
  return NBODY_OK;
}

int nbody_mailbox_run(void* ram_a, void* ram_b, int clock_khz) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || !ram_a || !ram_b) return NBODY_ERR_ARG;
  if (g_serve_on.load(std::memory_order_acquire)) return NBODY_ERR_STATE;   // the service thread owns the mailbox: write BEGIN, poll word 0
  return mailbox_run_impl(ram_a, ram_b, clock_khz, false);
}

int nbody_mailbox_serve(int on, int clock_khz) {
  if (!on) { serve_stop(); return NBODY_OK; }
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || g.nranks != 1) return NBODY_ERR_UNSUPPORTED;
  if (g_serve_on.load(std::memory_order_acquire)) { g_serve_khz = clock_khz; return NBODY_OK; }
  NBC(mailbox_rams());
  NBC(sync_all());
  g_serve_khz = clock_khz;
  g_serve_on.store(1, std::memory_order_release);
  g_serve_thread = std::thread(serve_loop);
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
    NBC(timer_drain(L.kern, 0));
    ms = std::max(ms, L.kern.ms);   // locals run concurrently: report the slowest device
    n += L.kern.n;
    if (reset) { L.kern.ms = 0.0; L.kern.n = 0; }
  }
  if (ms_total) *ms_total = ms;
  if (launches) *launches = n;
  return NBODY_OK;
}

int nbody_comm_time(double* wait_ms_total, long long* waits, int reset) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  double ms = 0.0; long long n = 0;
  for (int l = 0; l < g.nlocal; ++l) {
    Local& L = g.loc[l];
    HIPC(hipSetDevice(L.device));
    NBC(timer_drain(L.wait, 0));
    ms = std::max(ms, L.wait.ms);
    n += L.wait.n;
    if (reset) { L.wait.ms = 0.0; L.wait.n = 0; }
  }
  if (wait_ms_total) *wait_ms_total = ms;
  if (waits) *waits = n;
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
