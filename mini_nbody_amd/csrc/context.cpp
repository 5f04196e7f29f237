// context.cpp — the context behind include/nbody.h: options, the launch configuration (resolve_config), buffers, the step and its HIP
// graph, state transfer, the force entry points, the strict-arithmetic gate.  Host C++ only: every kernel launch goes through the nbl::
// functions of kernels.hip; RCCL lives in comm.cpp, the reference's mailbox in mailbox.cpp (nbody_internal.hpp says who owns what).
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
#include "nbody_internal.hpp"

using namespace nbk;

namespace nbi {

std::atomic<const char*> g_last_file{""};
std::atomic<int> g_last_line{0};
Global g;

namespace {

inline int rows_per_wg(int R, int wsplit) { return wsplit > 1 ? 64 : kBlock * R; }
int blocks_for(int rows, int R, int wsplit) { const int w = rows_per_wg(R, wsplit); return (rows + w - 1) / w; }
// words between two segments' partial sums: the launch's rows rounded up to 64 (ForceArgs::part_stride)
inline size_t part_stride(int rows) { return ((size_t)rows + 63) / 64 * 64; }

}  // namespace

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

namespace {

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
  const size_t nt = ticket_words(L.n_local);
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

}  // namespace

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
      HIPC(hipMemsetAsync(L.tickets, 0, ticket_words(L.n_local) * sizeof(unsigned), L.compute));
    }
  }
  g.view.n = g.n; g.view.n_local = g.loc[0].n_local; g.view.variant = g.variant; g.view.R = g.R; g.view.sub = g.sub; g.view.nseg = g.nseg;
  g.view.fuse = g.fuse; g.view.wsplit = g.wsplit;
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

namespace {

// loop forms that exist in the diagnostic build only (make diag): experiment encodings and timing-only forms
inline bool isa_phase_is_diag(int ph) { return ph >= 2; }

// how a launch finishes its rows: directly (one segment), by the last-arriving workgroup, or by combine_kernel
inline int finish_mode() { return g.nseg == 1 ? kFinishDirect : (g.fuse ? kFinishLast : kFinishStore); }

void fill_args(Local& L, ForceArgs& a, int row0, int row_count, const Finish& fin, float dt, double dt64) {
  memset(&a, 0, sizeof(a));
  a.src = L.pos[L.cur];
  a.rows = word_ptr(L.pos[L.cur], (size_t)L.first);
  if (L.src_direct) { a.src = L.src_direct; a.rows = L.src_direct; }   // (one-rank mailbox request: first = 0)
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

}  // namespace

bool takes_rows16(int row_count) {
  if (g.fp64 || g.opt.sum_order != NBODY_SUM_FPGA16 || g.wsplit != 16 || g.nseg != 1 || g.opt.variant == NBODY_VARIANT_SMEM) return false;
  static const int force_rows16 = [] { const char* e = getenv("NBODY_FPGA_ROWS16"); return (e && *e) ? atoi(e) : -1; }();
  const int cus = g.cu_count > 0 ? g.cu_count : 256;
  return force_rows16 > 0 || (force_rows16 < 0 && (long long)row_count < 64LL * cus);
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
  nbl::KernelSel sel = {g.fp64, g.variant, R, g.opt.arith, g.tile, g.opt.isa_phase, g.opt.variant != NBODY_VARIANT_SMEM ? 1 : 0, 0, 0};
  // The FPGA order with ONE segment (the mailbox's faithful mode) in a launch that would leave CUs idle with 64 rows per workgroup:
  // sixteen rows x sixteen chains per workgroup instead (force_fpga16r_f32) — the same bits from four times the workgroups.  Up to four
  // 16-row workgroups per CU (rows < 64 x CUs: there the 64-row form fills every CU too, with a quarter of the source fetches).
  // NBODY_FPGA_ROWS16 = 0 / 1 overrides (A/B).
  if (takes_rows16(row_count)) {
    sel.fpga_rows16 = 1;
    grid = dim3((row_count + 15) / 16, 1, 1);
    a.xcd_map = 0;
    a.t0_stamp = L.t0_stamp;
  }
  // optional occupancy cap: k workgroups (= k waves per SIMD) per CU by giving each 160 KiB / k of dynamic LDS
  if (g.opt.waves_per_simd > 0 && g.opt.waves_per_simd < 8) {
    const size_t static_lds = (g.variant == NBODY_VARIANT_LDS ? (size_t)g.tile * 32 : 0) + (a.wsplit > 1 ? (size_t)(a.wsplit - 1) * 64 * word_bytes() : 0) +
                              ((a.fpga16 && a.wsplit == 16 && sel.fpga_lds) ? (size_t)32 * 1024 : 0);
    // (a workgroup of WS waves holds WS / 4 wave slots per SIMD: the cap is on workgroups per CU = waves_per_simd / (WS / 4))
    const size_t budget = (size_t)(160 * 1024) / (size_t)g.opt.waves_per_simd;
    if (static_lds + 512 > budget) return NBODY_ERR_ARG;   // the kernel's own LDS (16-wave fp64 join: 30 KiB) already exceeds that share: no such cap exists
    sel.dyn_lds = budget - 512 - static_lds;
  }
  int slot;
  NBC(timer_begin(L.kern, L.compute, &slot));
  HIPC((hipError_t)nbl::launch_force_kernel(sel, L.compute, grid, a));   // which instantiation: kernels.hip
  return timer_end(L.kern, L.compute, slot);
}

// the two-launch form (NBODY_OPT_FUSE_COMBINE = 0): after the step's last force launch, add the partials
int launch_combine(Local& L, int row0, int row_count, const Finish& fin, float dt, double dt64) {
  if (row_count <= 0 || finish_mode() != kFinishStore) return NBODY_OK;
  HIPC(hipSetDevice(L.device));
  ForceArgs c;
  fill_args(L, c, row0, row_count, fin, dt, dt64);
  HIPC((hipError_t)nbl::launch_combine_kernel(g.fp64, L.compute, dim3((row_count + kBlock - 1) / kBlock), c));
  return NBODY_OK;
}

namespace {

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

}  // namespace

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

namespace {

void free_local(Local& L) {
  if (L.compute == nullptr && L.pos[0] == nullptr) return;
  (void)hipSetDevice(L.device);
  if (L.compute) (void)hipStreamSynchronize(L.compute);
  if (L.comm) (void)hipStreamSynchronize(L.comm);
  comm_destroy(L);
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
    HIPC(hipMemsetAsync(L.tickets, 0, ticket_words(L.n_local) * sizeof(unsigned), L.compute));
    HIPC(hipEventRecord(L.ev_own_ready, L.compute));
    L.all_present = true;
  }
  g.tickets_dirty = false;
  return sync_all();
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
      if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); NB_MARK(); return (int)e; }
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
    HIPC((hipError_t)nbl::launch_drift_kernel(g.fp64, L.compute, word_ptr(L.pos[L.cur], L.first), L.vel, L.n_local, dt, dt64));
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

}  // namespace

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

namespace {

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

}  // namespace

}  // namespace nbi

using namespace nbi;

// ============================================================================
extern "C" {

int nbody_init(int n, int ngpus, int fp64, int tile) { NB_REFUSE_WHILE_SERVED();
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
    if (pe != hipSuccess) { NB_MARK(); nbody_shutdown(); return (int)pe; }
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

int nbody_init_rank(int n, int fp64, int tile, int rank, int nranks, const void* uid128) { NB_REFUSE_WHILE_SERVED();
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
    if (pe != hipSuccess) { NB_MARK(); nbody_shutdown(); return (int)pe; }
  }
  g.cu_count = prop.multiProcessorCount; g.clock_khz = prop.clockRate;
  if (uid128) {
    // a communicator is created whenever an id is given — also for nranks = 1, where it carries no traffic in a step
    // but lets nbody_comm_selftest() push bytes through the same RCCL calls the multi-GPU job makes
    e = comm_create(L, nranks, rank, uid128);
    if (e) { nbody_shutdown(); return e; }
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
  HIPC((hipError_t)nbl::launch_rsqrt_selftest_kernel(first_bits, count, (unsigned long long*)out.p));
  HIPC(hipMemcpy(res3, out.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return NBODY_OK;
}

int nbody_rsqrt_selftest(unsigned first_bits, unsigned long long count, unsigned long long* mismatches, unsigned long long* ieee_lanes, unsigned* first_bad) { NB_REFUSE_WHILE_SERVED();
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

int nbody_strict_proof(unsigned long long* mismatches, unsigned* first_bad) { NB_REFUSE_WHILE_SERVED();
  int rc;
  if (g.init) rc = prove_strict_context();
  else { const int dev = rsqrt_device(); if (dev < 0) return NBODY_ERR_NO_DEVICE; rc = prove_strict_on(dev); }
  if (mismatches) *mismatches = rc == NBODY_ERR_UNSUPPORTED ? g_strict_bad : 0;
  if (first_bad) *first_bad = rc == NBODY_ERR_UNSUPPORTED ? g_strict_first_bad : 0;
  return rc;
}

int nbody_rsqrt_strict(const float* x, float* y, int n, int ieee_only) { NB_REFUSE_WHILE_SERVED();
  if (!x || !y || n <= 0) return NBODY_ERR_ARG;
  const int dev = rsqrt_device();
  if (dev < 0) return NBODY_ERR_NO_DEVICE;
  HIPC(hipSetDevice(dev));
  DevBuf dx, dy;
  HIPC(hipMalloc(&dx.p, (size_t)n * sizeof(float)));
  HIPC(hipMalloc(&dy.p, (size_t)n * sizeof(float)));
  HIPC(hipMemcpy(dx.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPC((hipError_t)nbl::launch_rsqrt_array_kernel((const float*)dx.p, (float*)dy.p, n, ieee_only));
  HIPC(hipMemcpy(y, dy.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return NBODY_OK;
}

void nbody_shutdown(void) {
  mailbox_shutdown();   // the mailbox's service thread, if one runs, ends before anything it uses is freed; then its RAM images go
  drop_step_graph();
  for (int l = 0; l < kMaxLocal; ++l) free_local(g.loc[l]);
  if (g.host_stage) { (void)hipHostFree(g.host_stage); g.host_stage = nullptr; }
  g.host_gather = nullptr; g.host_gather_user = nullptr;
  g.init = false; g.nlocal = 0; g.nranks = 1;
}

int nbody_set_option(int key, int value) { NB_REFUSE_WHILE_SERVED();
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
      // experiment encodings and timing-only forms (wrong results) are not in the product library: `make diag`
      if (!nbl::diag_build() && isa_phase_is_diag(value) && !(g.init && g.fp64 && value == 2)) return NBODY_ERR_UNSUPPORTED;
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
    case NBODY_INFO_N: *value = g.view.n; break;
    case NBODY_INFO_N_LOCAL: *value = g.view.n_local; break;
    case NBODY_INFO_FIRST_BODY: *value = L.first; break;
    case NBODY_INFO_RANK: *value = L.rank; break;
    case NBODY_INFO_NRANKS: *value = g.nranks; break;
    case NBODY_INFO_VARIANT: *value = g.view.variant; break;
    case NBODY_INFO_IBLOCK: *value = g.view.R; break;
    case NBODY_INFO_JSUB: *value = g.view.sub; break;
    case NBODY_INFO_NSEG: *value = g.view.nseg; break;
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
      *value = force + ((g.view.nseg > 1 && !g.view.fuse) ? 1 : 0);   // (finish_mode() of the context's own configuration)
      break;
    }
    case NBODY_INFO_HAS_COMM: *value = L.comm_h ? 1 : 0; break;
    case NBODY_INFO_WSPLIT: *value = g.view.wsplit; break;
    case NBODY_INFO_ISA_PHASE: *value = g.opt.isa_phase; break;
    case NBODY_INFO_LONG_BUFFERS: *value = g.opt.long_buffers; break;
    case NBODY_INFO_XCD_MAP: *value = g.opt.xcd_map; break;
    case NBODY_INFO_FUSE_COMBINE: *value = g.view.fuse; break;
    case NBODY_INFO_COMM_FORM: *value = g.nranks > 1 ? resolved_comm_form() : -1; break;
    case NBODY_INFO_COMM_PRIORITY: *value = g.comm_priority; break;
    case NBODY_INFO_MAILBOX_SERVED: *value = mailbox_served(); break;
    case NBODY_INFO_MAILBOX_SERVING: *value = mailbox_serving() ? 1 : 0; break;
    case NBODY_INFO_DIAG_BUILD: *value = nbl::diag_build() ? 1 : 0; break;
    default: return NBODY_ERR_ARG;
  }
  return NBODY_OK;
}

static const char* base_name(const char* path) { const char* s = strrchr(path, '/'); return s ? s + 1 : path; }

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
  if (code > 0 && code < 1000) { snprintf(buf, sizeof(buf), "HIP error %d (%s) near %s:%d", code, hipGetErrorString((hipError_t)code), base_name(g_last_file.load(std::memory_order_relaxed)), g_last_line.load(std::memory_order_relaxed)); return buf; }
  if (code >= 2000) { snprintf(buf, sizeof(buf), "RCCL error %d near %s:%d", code - 2000, base_name(g_last_file.load(std::memory_order_relaxed)), g_last_line.load(std::memory_order_relaxed)); return buf; }
  snprintf(buf, sizeof(buf), "nbody: unknown error %d", code);
  return buf;
}

int nbody_download_slice(void* pos_words, void* vel_words) { NB_REFUSE_WHILE_SERVED();
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

int nbody_upload(const BodySystem* host) { NB_REFUSE_WHILE_SERVED(); if (!host) return NBODY_ERR_ARG; if (g.init && g.fp64) return NBODY_ERR_STATE; return upload_impl(host->pos, host->vel); }
int nbody_download(BodySystem* host) { NB_REFUSE_WHILE_SERVED(); if (!host) return NBODY_ERR_ARG; if (g.init && g.fp64) return NBODY_ERR_STATE; return download_impl(host->pos, host->vel); }
int nbody_upload_d(const BodySystemD* host) { NB_REFUSE_WHILE_SERVED(); if (!host) return NBODY_ERR_ARG; if (g.init && !g.fp64) return NBODY_ERR_STATE; return upload_impl(host->pos, host->vel); }
int nbody_download_d(BodySystemD* host) { NB_REFUSE_WHILE_SERVED(); if (!host) return NBODY_ERR_ARG; if (g.init && !g.fp64) return NBODY_ERR_STATE; return download_impl(host->pos, host->vel); }

int bodyForce(float* pos, float* vel, float dt, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && g.fp64) return NBODY_ERR_STATE; return body_force_impl(pos, vel, dt, (double)dt, n); }
int integrate(float* pos, const float* vel, float dt, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && g.fp64) return NBODY_ERR_STATE; return integrate_impl(pos, vel, dt, (double)dt, n); }
int bodyForce_d(double* pos, double* vel, double dt, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && !g.fp64) return NBODY_ERR_STATE; return body_force_impl(pos, vel, (float)dt, dt, n); }
int integrate_d(double* pos, const double* vel, double dt, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && !g.fp64) return NBODY_ERR_STATE; return integrate_impl(pos, vel, (float)dt, dt, n); }

int nbody_step(float dt, int nsteps) { NB_REFUSE_WHILE_SERVED(); if (g.init && g.fp64) return NBODY_ERR_STATE; return step_impl(dt, (double)dt, nsteps); }
int nbody_step_d(double dt, int nsteps) { NB_REFUSE_WHILE_SERVED(); if (g.init && !g.fp64) return NBODY_ERR_STATE; return step_impl((float)dt, dt, nsteps); }
int nbody_sync(void) { NB_REFUSE_WHILE_SERVED(); if (!g.init) return NBODY_ERR_NOT_INIT; return sync_all(); }

int nbody_forces(const float* pos_words, float* force_words, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && g.fp64) return NBODY_ERR_STATE; return forces_impl(pos_words, force_words, n); }
int nbody_forces_d(const double* pos_words, double* force_words, int n) { NB_REFUSE_WHILE_SERVED(); if (g.init && !g.fp64) return NBODY_ERR_STATE; return forces_impl(pos_words, force_words, n); }

int nbody_forces_rows(int first_row, int n_rows, float* force_words) { NB_REFUSE_WHILE_SERVED(); if (g.init && g.fp64) return NBODY_ERR_STATE; return forces_rows_impl(first_row, n_rows, force_words); }
int nbody_forces_rows_d(int first_row, int n_rows, double* force_words) { NB_REFUSE_WHILE_SERVED(); if (g.init && !g.fp64) return NBODY_ERR_STATE; return forces_rows_impl(first_row, n_rows, force_words); }

int nbody_kernel_time(double* ms_total, long long* launches, int reset) { NB_REFUSE_WHILE_SERVED();
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

int nbody_comm_time(double* wait_ms_total, long long* waits, int reset) { NB_REFUSE_WHILE_SERVED();
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

int nbody_device_ptr(int which, void** ptr, size_t* bytes) { NB_REFUSE_WHILE_SERVED();
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
