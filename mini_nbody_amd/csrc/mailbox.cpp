// mailbox.cpp — the reference's only interface: the memory-mapped mailbox of S/top_level.vhd:176-272 and
// S/compute_store.vhd:175-242 on pinned host RAM images, one request at a time, called (nbody_mailbox_run) or served by a library thread
// that stands in for the FSM's `waiting` state (nbody_mailbox_serve).  Host C++ only.
//
// THE ADDRESS MAP is the RTL's, established by a cycle model of its store side (tests/rtl_model.py, tests/test_fpga_store_model.py):
//   RAM A  word 0 = control {bit 0 BEGIN, bits 46:32 NUM_PTS}; words 1..N = bodies {x, y, z, ignored}       S/top_level.vhd:184-185, 206-208
//   RAM B  word k = {Fx, Fy, Fz, 0} of body k — the index the body has in RAM A; word 0 is NEVER written      S/compute_store.vhd:221-242
// (:221-238 register write_we and STORE_PTR + 1 on the same edge and :241 forms write_addr from STORE_PTR combinationally, so the RAM
// samples we = 1 with the incremented address; the name ZERO_PTR, :76-77, suggests the author meant word k - 1 — the RTL does not do it.)
// Both RAMs are capacity + 1 words.  What the RTL as written does that this file deliberately does not: INTEGRATION.md, "Departures".
#include <atomic>
#include <chrono>
#include <thread>

#include "nbody_internal.hpp"

using namespace nbk;

namespace nbi {

namespace {

// One context serves requests of ANY NUM_PTS up to its capacity, as the RTL samples NUM_PTS with every BEGIN (:180-186) against a RAM
// sized once (:45).  The buffers are sized for the capacity; a request switches N and the launch configuration for its own duration
// (resolve_config is host arithmetic) and leaves the context's N, options and captured step graph as they were.
constexpr int kMailboxMaxPoints = 32767;   // ram_depth - 1, S/top_level.vhd:45

// the mailbox's two RAMs as the PS sees them (S/top_level.vhd:100-117, 148-163): pinned host memory the device reads (RAM A)
// and writes (RAM B, and word 0 of RAM A on completion) itself; allocated on the first request or by nbody_mailbox_open
void* mb_a = nullptr; void* mb_b = nullptr;
void* mb_a_dev = nullptr; void* mb_b_dev = nullptr;   // the same memory as the device addresses it
// completion as the device signals it (mailbox_done_kernel): a sequence word in pinned memory that the PS never writes, and the
// device-side start stamp of the tick counter
unsigned* mb_seq = nullptr; unsigned* mb_seq_dev = nullptr;
unsigned long long* mb_t0_dev = nullptr;
unsigned mb_seq_next = 0;       // sequence number of the last request whose completion the device was asked to write
int mb_rt_khz = 100000;         // rate of s_memrealtime (100 MHz on gfx950; hipDeviceAttributeWallClockRate)
int mb_done_by_device = 1;      // NBODY_MAILBOX_DONE=host: the host thread writes word 0 after hipStreamQuery says so (round 5's form; A/B)
unsigned mb_since_query = 0;
int mb_direct_max = 256;        // requests of at most this many bodies whose force launch is the 16-row FPGA kernel read RAM A from that kernel: no
                                // ingest launch (NBODY_MAILBOX_DIRECT_MAX; 0 = always ingest).  Measured on one box, faithful mode
                                // (profiles/r06_mailbox_rate.txt block G): N = 9 15.5 -> 12.5 us, N = 100 16.5 -> 13.8; N = 1024 20.0 -> 21.2 (every one of
                                // its 64 workgroups would read 16 KiB over PCIe): hence 256

int mailbox_rams() {   // RAM A and RAM B: capacity + 1 words each (+ slack), pinned, mapped, coherent
  if (mb_a && mb_b && mb_seq && mb_t0_dev) return NBODY_OK;
  Local& L = g.loc[0];
  HIPC(hipSetDevice(L.device));
  const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
  const size_t bytes = ((size_t)g.cap + 1 + 64) * 16;
  if (!mb_a) { HIPC(hipHostMalloc(&mb_a, bytes, flags)); memset(mb_a, 0, bytes); }
  if (!mb_b) { HIPC(hipHostMalloc(&mb_b, bytes, flags)); memset(mb_b, 0, bytes); }
  if (!mb_seq) { HIPC(hipHostMalloc((void**)&mb_seq, 64, flags)); memset(mb_seq, 0, 64); mb_seq_next = 0; }
  if (!mb_t0_dev) { HIPC(hipMalloc((void**)&mb_t0_dev, 64)); HIPC(hipMemset(mb_t0_dev, 0, 64)); }
  HIPC(hipHostGetDevicePointer(&mb_a_dev, mb_a, 0));
  HIPC(hipHostGetDevicePointer(&mb_b_dev, mb_b, 0));
  HIPC(hipHostGetDevicePointer((void**)&mb_seq_dev, mb_seq, 0));
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, L.device) == hipSuccess && khz > 0) mb_rt_khz = khz;
  else (void)hipGetLastError();
  const char* e = getenv("NBODY_MAILBOX_DONE");
  mb_done_by_device = !(e && !strcmp(e, "host"));
  const char* dm = getenv("NBODY_MAILBOX_DIRECT_MAX");
  if (dm && *dm) mb_direct_max = atoi(dm);
  return NBODY_OK;
}

// N and the launch configuration of a one-rank context switched for the duration of one request.  What nbody_get_info reports is the
// CONTEXT's configuration (Global::view, published by reconfigure()), which a request never touches.
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
      HIPC(hipMemsetAsync(L.tickets, 0, ticket_words(g.cap) * sizeof(unsigned), L.compute));
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

// completion of everything on `stream` as the HOST sees it: polled for the first 200 us (a request at the mailbox's sizes takes 7-500 us
// of device time and an interrupt-driven wait adds tens of us of wake-up), then a blocking wait
int wait_stream(hipStream_t stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipStreamQuery(stream);
    if (e == hipSuccess) return NBODY_OK;
    if (e != hipErrorNotReady) { NB_MARK(); return (int)e; }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
  }
  HIPC(hipStreamSynchronize(stream));
  return NBODY_OK;
}

// completion as the DEVICE wrote it: the sequence word mailbox_done_kernel stores after it has cleared BEGIN.  Memory is polled, no runtime
// call; after 300 us (N = 32767 takes 500) a blocking hipStreamSynchronize takes over, so nothing can spin for ever.
int wait_seq(hipStream_t stream, unsigned seq) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    if (__atomic_load_n(mb_seq, __ATOMIC_ACQUIRE) == seq) break;
    __builtin_ia32_pause();
    if ((++spins & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
      HIPC(hipStreamSynchronize(stream));
      if (__atomic_load_n(mb_seq, __ATOMIC_ACQUIRE) != seq) { NB_MARK(); return NBODY_ERR_STATE; }   // the queue drained and the word is not there
      mb_since_query = 0;
      return NBODY_OK;
    }
  }
  // the runtime retires finished commands when it is asked about the stream: every 256 requests, not every request
  if (++mb_since_query >= 256) { mb_since_query = 0; hipError_t e = hipStreamQuery(stream); if (e != hipSuccess && e != hipErrorNotReady) { NB_MARK(); return (int)e; } }
  return NBODY_OK;
}

// the launches of one request on the compute stream: RAM A's read port (which starts the tick count), the force pass storing into RAM B
// (and its combine), and — device-written completion — the one-wave launch that rewrites word 0
int mailbox_launches(Local& L, int num_pts, bool done_by_device, unsigned seq, int clock_khz) {
  // bodies are words 1..N                                              S/top_level.vhd:55, 206-208
  // A handful of bodies in the faithful mode: the 16-row kernel reads RAM A itself (its tile loads ARE the PCIe reads) and stamps the tick
  // count's start; everything else goes through the ingest launch, which reads RAM A once for all workgroups.
  const bool direct = done_by_device && num_pts <= mb_direct_max && takes_rows16(num_pts);
  if (direct) { L.src_direct = (const char*)mb_a_dev + 16; L.t0_stamp = mb_t0_dev; }
  else HIPC((hipError_t)nbl::launch_ingest_kernel(L.compute, L.pos[L.cur], (const char*)mb_a_dev + 16, num_pts, done_by_device ? mb_t0_dev : nullptr));
  // RAM B's write port: the force launch (or its combine) stores {Fx, Fy, Fz, 0} of body k at word k itself — row k - 1 of the launch
  // goes to force_dst[k - 1] and force_dst is word 1 —; word 0 and the words beyond N are never written       S/compute_store.vhd:213, 221-242
  const Finish fin = {false, false, true};
  L.force_dst = (char*)mb_b_dev + 16;
  int rc = launch_force(L, 0, num_pts, g.nslices - 1, g.nslices, fin, 0.f, 0.0);
  if (!rc) rc = launch_combine(L, 0, num_pts, fin, 0.f, 0.0);
  L.force_dst = nullptr;
  L.src_direct = nullptr; L.t0_stamp = nullptr;
  if (rc || !done_by_device) return rc;
  HIPC((hipError_t)nbl::launch_mailbox_done_kernel(L.compute, mb_a_dev, mb_seq_dev, mb_t0_dev, seq, (unsigned)clock_khz, (unsigned)mb_rt_khz));
  return NBODY_OK;
}

// One request of NUM_PTS > 0 on a one-rank context.  *device_done: word 0 of the library's RAM A image has been rewritten by the device.
int mailbox_request(const void* ram_a, void* ram_b, int num_pts, int clock_khz, bool* device_done) {
  Local& L = g.loc[0];
  *device_done = false;
  HIPC(hipSetDevice(L.device));
  NBC(mailbox_rams());
  ActiveN scope;
  NBC(scope.enter(num_pts));
  // RAM A: the library's own pinned image is read in place; any other host buffer is copied into it first
  if (ram_a != mb_a) memcpy((char*)mb_a + 16, (const char*)ram_a + 16, (size_t)num_pts * 16);
  L.all_present = true;
  const bool by_device = mb_done_by_device != 0;
  const unsigned seq = by_device ? ++mb_seq_next : 0u;
  if (by_device && ram_a != mb_a) ((uint32_t*)mb_a)[0] = 1u;   // (the device clears THIS image's BEGIN; the caller's word 0 follows below)
  const int rc = mailbox_launches(L, num_pts, by_device, seq, clock_khz);
  if (rc) {
    g.tickets_dirty = true;
    if (by_device) { (void)hipStreamSynchronize(L.compute); --mb_seq_next; }   // nothing of this request may still be writing when word 0 is rewritten
    return rc;
  }
  if (by_device) { NBC(wait_seq(L.compute, seq)); *device_done = true; }
  else NBC(wait_stream(L.compute));
  if (ram_b != mb_b) memcpy((char*)ram_b + 16, (char*)mb_b + 16, (size_t)num_pts * 16);   // words 1..N; the caller's word 0 is not written either
  return NBODY_OK;
}

// One request from the RAM images (the body of nbody_mailbox_run and of the service thread).  `served`: an error has no return value to
// travel in, so it is written into word 0 (bits 127:96, which the RTL always writes as 0) with BEGIN cleared.
int mailbox_run_impl(void* ram_a, void* ram_b, int clock_khz, bool served) {
  const auto t0 = std::chrono::steady_clock::now();
  // word 0: bit 0 BEGIN, bits [46:32] NUM_PTS, sampled with every request        S/top_level.vhd:180-186
  uint32_t* w0 = (uint32_t*)ram_a;
  if (!(__atomic_load_n(&w0[0], __ATOMIC_ACQUIRE) & 1u)) return NBODY_ERR_STATE;   // the FSM stays in `waiting`: nothing is read, nothing is written
  const int num_pts = (int)(w0[1] & 0x7FFFu);
  const int khz = clock_khz > 0 ? clock_khz : 300000;
  int rc = NBODY_OK;
  bool device_done = false;
  if (g.nranks == 1) {
    if (num_pts > g.cap) rc = NBODY_ERR_ARG;   // (the RTL's RAM always holds 32767 bodies; a smaller capacity is this library's notion)
    // NUM_PTS = 0: block_setup finds THIS_PTR > NUM_PTS at once and goes to `complete` (S/top_level.vhd:189-192): RAM B untouched
    else if (num_pts > 0) rc = mailbox_request(ram_a, ram_b, num_pts, khz, &device_done);
  } else {
    // a context over several devices / ranks keeps its fixed N: every rank brings the same images (nbody_forces); force k at word k
    if (num_pts != g.n) rc = NBODY_ERR_ARG;
    else rc = forces_impl((const float*)ram_a + 4, (float*)ram_b + 4, num_pts);
  }
  if (rc && !served) return rc;
  if (device_done) {
    // `complete` was the device's: word 0 of the library's image already reads {ticks, BEGIN = 0}.  A caller's own image gets that word.
    if (ram_a != mb_a) {
      const uint32_t* d0 = (const uint32_t*)mb_a;
      w0[1] = d0[1]; w0[2] = 0; w0[3] = 0;
      __atomic_store_n(&w0[0], 0u, __ATOMIC_RELEASE);
    }
    return NBODY_OK;
  }
  // completion by the host (NUM_PTS = 0, a refused request of the served form, several ranks, NBODY_MAILBOX_DONE=host):
  // word 0 <- {ticks in [63:32], 0 elsewhere}: BEGIN reads 0                      S/top_level.vhd:146, 255-263
  // one tick = 1000 clocks (S/top_level.vhd:121-144); the counter goes to 1 on BEGIN's rising edge (:138-139); BEGIN-to-done as this
  // host sees it
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const uint32_t ticks = rc ? 0u : 1u + (uint32_t)(ms * (double)khz / 1000.0);
  w0[1] = ticks; w0[2] = 0; w0[3] = (uint32_t)rc;
  __atomic_store_n(&w0[0], 0u, __ATOMIC_RELEASE);   // BEGIN is cleared LAST: whoever sees it cleared sees the ticks and RAM B
  return rc;
}

// The PL block serves the PS without being called: its FSM samples word 0 of RAM A every clock (S/top_level.vhd:180-186).  The same on
// a host: a library thread polls word 0 of the context's own RAM A, runs every request it finds and rewrites word 0 (or has the device
// rewrite it) — the driver only writes and reads memory.  Idle polling backs off: `pause` for the first ~ms, then yields, then 50-us
// naps after ~0.1 s without work.
std::thread g_serve_thread;
std::atomic<int> g_serve_on{0};
std::atomic<long long> g_served{0};
std::atomic<int> g_serve_khz{0};

void serve_loop() {
  uint32_t* w0 = (uint32_t*)mb_a;
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
    (void)mailbox_run_impl(mb_a, mb_b, g_serve_khz.load(std::memory_order_relaxed), true);
    g_served.fetch_add(1, std::memory_order_release);
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

void mailbox_shutdown() {
  serve_stop();
  if (mb_a) { (void)hipHostFree(mb_a); mb_a = nullptr; mb_a_dev = nullptr; }
  if (mb_b) { (void)hipHostFree(mb_b); mb_b = nullptr; mb_b_dev = nullptr; }
  if (mb_seq) { (void)hipHostFree(mb_seq); mb_seq = nullptr; mb_seq_dev = nullptr; }
  if (mb_t0_dev) { (void)hipFree(mb_t0_dev); mb_t0_dev = nullptr; }
  mb_seq_next = 0; mb_since_query = 0;
}
bool mailbox_serving() { return g_serve_on.load(std::memory_order_acquire) != 0; }
long long mailbox_served() { return g_served.load(std::memory_order_acquire); }

}  // namespace nbi

using namespace nbi;

// ============================================================================
extern "C" {

int nbody_mailbox_open(int capacity, int faithful) {
  NB_REFUSE_WHILE_SERVED();
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
  if (!mailbox_serving()) NBC(mailbox_rams());   // (a served mailbox has them already; nothing is allocated beside the service thread)
  if (ram_a) *ram_a = mb_a;
  if (ram_b) *ram_b = mb_b;
  if (capacity) *capacity = g.cap < kMailboxMaxPoints ? g.cap : kMailboxMaxPoints;
  return NBODY_OK;
}

int nbody_mailbox_run(void* ram_a, void* ram_b, int clock_khz) {
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || !ram_a || !ram_b) return NBODY_ERR_ARG;
  NB_REFUSE_WHILE_SERVED();   // the service thread owns the mailbox: write BEGIN, poll word 0
  return mailbox_run_impl(ram_a, ram_b, clock_khz, false);
}

int nbody_mailbox_serve(int on, int clock_khz) {
  if (!on) { serve_stop(); return NBODY_OK; }
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (g.fp64 || g.nranks != 1) return NBODY_ERR_UNSUPPORTED;
  if (mailbox_serving()) { g_serve_khz.store(clock_khz, std::memory_order_relaxed); return NBODY_OK; }
  NBC(mailbox_rams());
  NBC(sync_all());
  g_serve_khz.store(clock_khz, std::memory_order_relaxed);
  g_serve_on.store(1, std::memory_order_release);
  g_serve_thread = std::thread(serve_loop);
  return NBODY_OK;
}

}  // extern "C"
