// comm.cpp — how the other ranks' position slices reach a rank: RCCL resolved with dlopen, the transfer
// plans (ring / direct), the all-gather of a step on the second stream, the host-staged fallback transport, probes and self-tests.
// Host C++ only.
#include <dlfcn.h>

#include "nbody_internal.hpp"

using namespace nbk;

namespace nbi {

namespace {

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

}  // namespace

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

// ncclCommInitRank for local L (whenever an id is given — also for nranks = 1, where the communicator carries no traffic in a step
// but lets nbody_comm_selftest() push bytes through the same RCCL calls the multi-GPU job makes)
int comm_create(Local& L, int nranks, int rank, const void* uid128) {
  NBC(rccl_load());
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof(id));
  HIPC(hipSetDevice(L.device));
  ncclResult_t r = g_rccl.CommInitRank(&L.comm_h, nranks, id, rank);
  if (r != ncclSuccess) { NB_MARK(); L.comm_h = nullptr; return 2000 + (int)r; }
  return NBODY_OK;
}
void comm_destroy(Local& L) {
  if (L.comm_h && g_rccl.CommDestroy) g_rccl.CommDestroy(L.comm_h);
  L.comm_h = nullptr;
}

namespace {

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

}  // namespace

// which form NBODY_COMM_AUTO means (profiles/r03_comm_under_load.md): ONE RCCL kernel per step enqueued ahead of the force
// launch — ncclAllGather (whose algorithm over xGMI is a ring) when the slices are equal, the DIRECT group when they are
// not — rather than P-1 dependent ring groups, each of which would have to win wave slots from a force kernel that fills
// every CU.  NBODY_COMM_RING remains the north_star's literal form, one event per arriving slice (NBODY_OPT_OVERLAP 2).
int resolved_comm_form() {
  const bool even = (g.n % g.nranks) == 0;
  if (g.opt.comm == NBODY_COMM_AUTO) return even ? NBODY_COMM_ALLGATHER : NBODY_COMM_DIRECT;
  if (g.opt.comm == NBODY_COMM_ALLGATHER && !even) return NBODY_COMM_RING;
  return g.opt.comm;
}

namespace {

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

}  // namespace

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

}  // namespace nbi

using namespace nbi;

// ============================================================================
extern "C" {

int nbody_unique_id(void* uid128) {
  if (!uid128) return NBODY_ERR_ARG;
  NBC(rccl_load());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(uid128, &id, sizeof(id));
  return NBODY_OK;
}

// Transport self-test on the communicator of nbody_init_rank: (1) an in-place all-gather of a patterned scratch array
// through rccl_gather() in the configured NBODY_OPT_COMM form, (2) one ring step (ncclSend to rank+1, ncclRecv from
// rank-1, grouped) of a patterned block — with one rank both are device-local, which is how a one-GPU box exercises
// the library's RCCL calls (symbols, argument order, byte counts).  Every received word is checked on the host.
int nbody_comm_selftest(long long* bytes_moved) { NB_REFUSE_WHILE_SERVED();
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
        if (host[(size_t)w * (wb / 4) + k] != ((uint32_t)q << 24) + ((uint32_t)w & 0xffffffu)) { NB_MARK(); return NBODY_ERR_STATE; }
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
        if (host[(size_t)w * (wb / 4) + k] != 0xA5000000u + ((uint32_t)prev << 20) + ((uint32_t)w & 0xfffffu)) { NB_MARK(); return NBODY_ERR_STATE; }
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
int nbody_comm_selftest_virtual(int vp, int form, long long* bytes_moved) { NB_REFUSE_WHILE_SERVED();
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
        if (!snd) { NB_MARK(); return NBODY_ERR_STATE; }   // the plans do not pair up
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
        if (host[(size_t)w * lanes + k] != pattern(w, k)) { NB_MARK(); return NBODY_ERR_STATE; }
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
int nbody_comm_probe(long long bytes, int when, double* comm_ms, double* force_ms) { NB_REFUSE_WHILE_SERVED();
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

int nbody_set_host_gather(nbody_host_gather_fn fn, void* user) { NB_REFUSE_WHILE_SERVED();
  if (!g.init) return NBODY_ERR_NOT_INIT;
  if (!g.multiprocess) return NBODY_ERR_STATE;
  g.host_gather = (host_gather_fn)fn;
  g.host_gather_user = user;
  return NBODY_OK;
}

}  // extern "C"
