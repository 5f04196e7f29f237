// hip_stub.cpp — TEST INFRASTRUCTURE (tests/test_host_sanitizers.py): a host-only stand-in for the HIP runtime and for kernels.hip, so that
// the library's three HOST translation units (context.cpp, comm.cpp, mailbox.cpp — everything that is not device code) can run on a
// machine without a GPU under AddressSanitizer / UBSan / ThreadSanitizer (sanitizers run on the CPU build only; the GPU pool has none).
// "Device" memory is malloc'd (so ASan sees every copy size and every index the host computed), streams execute synchronously, a stream
// capture records closures that hipGraphLaunch replays, and the nbl:: launch functions run a SIMPLE stand-in force (F_i = sum over the
// launch's segments of (r_j - r_i), float or double, segment by segment) through the REAL data flow of a launch: ForceArgs, segment
// bounds, per-segment partial sums, one arrival counter per 64 rows, ascending combine, apply (store / kick / drift), the mailbox's
// ingest and device-written completion.  It says nothing about the kernels' arithmetic (the GPU tests do); it checks the host's logic:
// buffer sizes, offsets, state switching per request, the service thread's hand-over, the guard, lifetimes.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <vector>

#include "../../mini_nbody_amd/csrc/nbody_internal.hpp"

using nbk::ForceArgs;

namespace {
struct FakeEvent { double ms; };
struct FakeGraph { std::vector<std::function<void()>> ops; };
thread_local bool t_capturing = false;
thread_local FakeGraph* t_capture = nullptr;
int g_devices = [] { const char* e = getenv("STUB_DEVICES"); return e && *e ? atoi(e) : 1; }();

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int enqueue(std::function<void()> f) {
  if (t_capturing) t_capture->ops.push_back(std::move(f));
  else f();
  return 0;
}

template <typename T>
struct W4 { T x, y, z, w; };

template <typename T>
void apply(const ForceArgs& a, int i, T fx, T fy, T fz) {
  typedef W4<T> V;
  const V me = ((const V*)a.rows)[i];
  if (a.force_out) ((V*)a.force_out)[i] = V{fx, fy, fz, (T)0};
  if (a.do_kick) {
    const T dt = sizeof(T) == 8 ? (T)a.dt64 : (T)a.dt;
    V v = ((V*)a.vel)[i];
    v.x += dt * fx; v.y += dt * fy; v.z += dt * fz;
    ((V*)a.vel)[i] = v;
    if (a.do_drift) ((V*)a.pos_next_rows)[i] = V{me.x + v.x * dt, me.y + v.y * dt, me.z + v.z * dt, me.w};
  }
}

template <typename T>
void combine_rows(const ForceArgs& a, int i0, int i1) {
  typedef W4<T> V;
  for (int i = i0; i < i1; ++i) {
    T fx = 0, fy = 0, fz = 0;
    for (int sg = 0; sg < a.nseg; ++sg) {
      const V p = ((const V*)a.partial)[(size_t)sg * a.part_stride + (i - a.row0)];
      fx += p.x; fy += p.y; fz += p.z;
    }
    apply<T>(a, i, fx, fy, fz);
  }
}

template <typename T>
void force(ForceArgs a, int ny) {
  typedef W4<T> V;
  const V* src = (const V*)a.src;
  const V* rows = (const V*)a.rows;
  const int row_end = a.row0 + a.row_count;
  for (int y = 0; y < ny; ++y) {
    int q = a.slice_start - y / a.sub;
    q %= a.nslices; if (q < 0) q += a.nslices;
    const int t = y % a.sub, seg = q * a.sub + t;
    int jb, je;
    nbk::segment_bounds(q, t, a.n_src, a.nslices, a.sub, &jb, &je);
    for (int i = a.row0; i < row_end; ++i) {
      const V me = rows[i];
      T fx = 0, fy = 0, fz = 0;
      for (int j = jb; j < je; ++j) { fx += src[j].x - me.x; fy += src[j].y - me.y; fz += src[j].z - me.z; }
      if (a.finish == nbk::kFinishDirect) apply<T>(a, i, fx, fy, fz);
      else ((V*)a.partial)[(size_t)seg * a.part_stride + (i - a.row0)] = V{fx, fy, fz, (T)0};
    }
  }
  if (a.finish == nbk::kFinishLast) {
    for (int u = 0; u * 64 < a.row_count; ++u) {
      a.tickets[u] += (unsigned)ny;
      if ((int)a.tickets[u] == a.nseg) {
        combine_rows<T>(a, a.row0 + u * 64, std::min(row_end, a.row0 + (u + 1) * 64));
        a.tickets[u] = 0;
      }
    }
  }
}
}  // namespace

namespace nbl {
bool diag_build() { return false; }
int launch_force_kernel(const KernelSel& k, hipStream_t, dim3 grid, const ForceArgs& a) {
  const int fp64 = k.fp64, ny = k.fpga_rows16 ? 1 : (int)grid.y;
  return enqueue([=] { if (a.t0_stamp) *a.t0_stamp = (unsigned long long)(now_ms() * 1e5); if (fp64) force<double>(a, ny); else force<float>(a, ny); });
}
int launch_combine_kernel(int fp64, hipStream_t, dim3, const ForceArgs& a) {
  return enqueue([=] { if (fp64) combine_rows<double>(a, a.row0, a.row0 + a.row_count); else combine_rows<float>(a, a.row0, a.row0 + a.row_count); });
}
int launch_drift_kernel(int fp64, hipStream_t, void* pos_rows, const void* vel, int n_rows, float dt, double dt64) {
  return enqueue([=] {
    for (int i = 0; i < n_rows; ++i) {
      if (fp64) { auto* p = (W4<double>*)pos_rows; auto* v = (const W4<double>*)vel; p[i].x += v[i].x * dt64; p[i].y += v[i].y * dt64; p[i].z += v[i].z * dt64; }
      else { auto* p = (W4<float>*)pos_rows; auto* v = (const W4<float>*)vel; p[i].x += v[i].x * dt; p[i].y += v[i].y * dt; p[i].z += v[i].z * dt; }
    }
  });
}
int launch_ingest_kernel(hipStream_t, void* dst_words, const void* ram_a_bodies, int n, unsigned long long* t0) {
  return enqueue([=] { memcpy(dst_words, ram_a_bodies, (size_t)n * 16); if (t0) *t0 = (unsigned long long)(now_ms() * 1e5); });
}
int launch_mailbox_done_kernel(hipStream_t, void* word0, unsigned* seq_word, const unsigned long long* t0, unsigned seq, unsigned clock_khz, unsigned rt_khz) {
  return enqueue([=] {
    unsigned* w0 = (unsigned*)word0;
    const unsigned long long dt = (unsigned long long)(now_ms() * 1e5) - *t0;
    w0[1] = (unsigned)(1ull + dt * clock_khz / ((unsigned long long)rt_khz * 1000ull)); w0[2] = 0; w0[3] = 0;
    __atomic_store_n(&w0[0], 0u, __ATOMIC_RELEASE);
    __atomic_store_n(seq_word, seq, __ATOMIC_RELEASE);
  });
}
int launch_rsqrt_selftest_kernel(unsigned, unsigned long long, unsigned long long* out3) { out3[0] = 0; out3[1] = 0; return 0; }
int launch_rsqrt_array_kernel(const float* x, float* y, int n, int) { for (int i = 0; i < n; ++i) y[i] = x[i]; return 0; }
}  // namespace nbl

// ---- the HIP runtime, as far as the three host files use it ----
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = g_devices; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d >= 0 && d < g_devices ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; p->clockRate = 2400000; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeWallClockRate ? 100000 : 0; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(64, (n + 63) / 64 * 64); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { enqueue([=] { memset(p, v, n); }); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { enqueue([=] { memcpy(d, s, n); }); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { enqueue([=] { memcpy(d, s, n); }); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new FakeEvent{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (FakeEvent*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!t_capturing) ((FakeEvent*)e)->ms = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(((FakeEvent*)b)->ms - ((FakeEvent*)a)->ms); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { t_capture = new FakeGraph; t_capturing = true; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { t_capturing = false; *g = (hipGraph_t)t_capture; t_capture = nullptr; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *x = (hipGraphExec_t) new FakeGraph(*(FakeGraph*)g); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete (FakeGraph*)g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete (FakeGraph*)x; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t) { for (auto& f : ((FakeGraph*)x)->ops) f(); return hipSuccess; }
}
