// host_sanity.cpp — TEST INFRASTRUCTURE: drives the C-ABI of include/nbody.h on top of hip_stub.cpp (no GPU) so that the library's HOST
// code runs under sanitizers.  usage: host_sanity [serve_requests [mailbox-only]]   exit code 0 = every check held.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/nbody.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "host_sanity: line %d: %s\n", __LINE__, #cond); exit(1); } } while (0)
#define OK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "host_sanity: line %d: %s = %d (%s)\n", __LINE__, #call, rc_, nbody_error_string(rc_)); exit(1); } } while (0)

static void bodies(std::vector<float>& p, int n, unsigned seed) {
  p.resize((size_t)n * 4);
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& v : p) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.0f - 1.0f; }
}

static long long info(int key) { long long v = -1; OK(nbody_get_info(key, &v)); return v; }

int main(int argc, char** argv) {
  const int serve_requests = argc > 1 ? atoi(argv[1]) : 10000;
  const bool mailbox_only = argc > 2;
  std::vector<float> pos, vel, p2, v2, f, f2;
  // ---- a context, the step loop with its graphs, forces, row windows, options that re-segment ----
  for (int n : {1, 63, 1000, 2085}) {
    if (mailbox_only) break;
    bodies(pos, n, 1); bodies(vel, n, 2);
    OK(nbody_init(n, 1, 0, 0));
    BodySystem b = {pos.data(), vel.data()};
    OK(nbody_upload(&b));
    OK(nbody_step(0.01f, 70));                     // one eager step, graphs of 32 + the rest
    OK(nbody_sync());
    p2.assign((size_t)n * 4, 0.f); v2.assign((size_t)n * 4, 0.f);
    BodySystem o = {p2.data(), v2.data()};
    OK(nbody_download(&o));
    CHECK(info(NBODY_INFO_STEPS_DONE) == 70);
    f.assign((size_t)n * 4, 0.f);
    OK(nbody_forces(pos.data(), f.data(), n));
    for (int jsub : {1, 3, 64}) for (int fuse : {0, 1}) for (int ws : {1, 4, 16}) {
      OK(nbody_set_option(NBODY_OPT_JSUB, jsub)); OK(nbody_set_option(NBODY_OPT_FUSE_COMBINE, fuse)); OK(nbody_set_option(NBODY_OPT_WSPLIT, ws));
      f2.assign((size_t)n * 4, -1.f);
      OK(nbody_forces(pos.data(), f2.data(), n));
      if (n > 1) { const int r0 = n / 3, nr = n - r0 < 100 ? n - r0 : 100; std::vector<float> w((size_t)nr * 4); OK(nbody_forces_rows(r0, nr, w.data())); }
      OK(nbody_upload(&b)); OK(nbody_step(0.01f, 5));
    }
    OK(nbody_set_option(NBODY_OPT_SUM_ORDER, NBODY_SUM_FPGA16)); OK(nbody_set_option(NBODY_OPT_JSUB, 1)); OK(nbody_set_option(NBODY_OPT_WSPLIT, -1));
    OK(nbody_forces(pos.data(), f2.data(), n));
    CHECK(bodyForce(pos.data(), vel.data(), 0.01f, n) == 0 && integrate(pos.data(), vel.data(), 0.01f, n) == 0);
    nbody_shutdown();
  }
  if (!mailbox_only) {  // fp64, and one process driving three (stub) devices: peer copies, ragged slices
    const int n = 1001;
    std::vector<double> dp((size_t)n * 4, 0.25), dv((size_t)n * 4, 0.0);
    for (size_t k = 0; k < dp.size(); ++k) dp[k] = (double)((k * 2654435761u) % 1000) / 1000.0;
    OK(nbody_init(n, 1, 1, 0));
    BodySystemD b = {dp.data(), dv.data()};
    OK(nbody_upload_d(&b)); OK(nbody_step_d(0.01, 9)); OK(nbody_download_d(&b));
    CHECK(nbody_step(0.01f, 1) == NBODY_ERR_STATE);
    nbody_shutdown();
    if (getenv("STUB_DEVICES") && atoi(getenv("STUB_DEVICES")) >= 3) {
      bodies(pos, n, 5); bodies(vel, n, 6);
      OK(nbody_init(n, 3, 0, 0));
      BodySystem bb = {pos.data(), vel.data()};
      for (int ov : {0, 1, 2}) { OK(nbody_set_option(NBODY_OPT_OVERLAP, ov)); OK(nbody_upload(&bb)); OK(nbody_step(0.01f, 4)); OK(nbody_download(&bb)); }
      f.assign((size_t)n * 4, 0.f);
      OK(nbody_forces(pos.data(), f.data(), n));
      nbody_shutdown();
    }
  }
  // ---- the mailbox: the address map, every form, the guard ----
  const int cap = 700;
  OK(nbody_mailbox_open(cap, 0));
  uint32_t* ram_a; float* ram_b; int c = 0;
  OK(nbody_mailbox_rams((void**)&ram_a, (void**)&ram_b, &c));
  CHECK(c == cap);
  bodies(pos, cap, 9);
  std::vector<std::vector<uint32_t>> first(cap + 1);
  auto post = [&](int n) { memcpy(ram_a + 4, pos.data(), (size_t)n * 16); ram_a[1] = (uint32_t)n; ram_a[2] = ram_a[3] = 0; };
  auto check_b = [&](const float* rb, int n, int words) {
    const uint32_t* w = (const uint32_t*)rb;
    for (int k = 0; k < 4; ++k) CHECK(w[k] == 0xDEADBEEFu);                                   // word 0 of RAM B is never written
    for (int k = (n + 1) * 4; k < words * 4; ++k) CHECK(w[k] == 0xDEADBEEFu);                  // nor the words beyond N
    std::vector<uint32_t> got(w + 4, w + 4 + (size_t)n * 4);
    if (first[n].empty()) first[n] = got;
    CHECK(got == first[n]);
  };
  auto fill = [&](float* rb, int words) { for (int k = 0; k < words * 4; ++k) ((uint32_t*)rb)[k] = 0xDEADBEEFu; };
  std::vector<uint32_t> own_a((size_t)(cap + 1) * 4);
  std::vector<float> own_b((size_t)(cap + 1) * 4);
  for (int round = 0; round < 3; ++round)
    for (int n : {9, 700, 0, 1, 64, 65, 300, 2, 3}) {
      fill(ram_b, cap + 1); post(n); ram_a[0] = 1;
      OK(nbody_mailbox_run(ram_a, ram_b, 300000));
      CHECK((ram_a[0] & 1u) == 0 && ram_a[1] >= 1 && ram_a[2] == 0 && ram_a[3] == 0);
      check_b(ram_b, n, cap + 1);
      fill(own_b.data(), cap + 1);
      own_a[0] = 1; own_a[1] = (uint32_t)n; own_a[2] = own_a[3] = 0; memcpy(own_a.data() + 4, pos.data(), (size_t)n * 16);
      OK(nbody_mailbox_run(own_a.data(), own_b.data(), 300000));                               // the caller's own images
      CHECK((own_a[0] & 1u) == 0 && own_a[1] >= 1);
      check_b(own_b.data(), n, cap + 1);
      CHECK(info(NBODY_INFO_N) == cap);
    }
  ram_a[0] = 0; CHECK(nbody_mailbox_run(ram_a, ram_b, 0) == NBODY_ERR_STATE);
  ram_a[0] = 1; ram_a[1] = cap + 1; CHECK(nbody_mailbox_run(ram_a, ram_b, 0) == NBODY_ERR_ARG);
  ram_a[0] = 0;                                                                                // (a refused call leaves BEGIN as the caller set it: nothing may be pending when the thread starts)
  // served: this thread posts and polls, a second thread hammers nbody_get_info and the refused entry points
  const long long n0 = info(NBODY_INFO_N), nseg0 = info(NBODY_INFO_NSEG), sub0 = info(NBODY_INFO_JSUB), ws0 = info(NBODY_INFO_WSPLIT);
  OK(nbody_mailbox_serve(1, 300000));
  std::atomic<int> stop{0}, bad{0};
  std::atomic<long long> looked{0};
  std::thread hammer([&] {
    std::vector<float> buf((size_t)cap * 4);
    BodySystem bs = {buf.data(), buf.data()};
    int k = 0;
    while (!stop.load(std::memory_order_acquire)) {
      long long v;
      if (nbody_get_info(NBODY_INFO_N, &v) || v != n0) bad++;
      if (nbody_get_info(NBODY_INFO_NSEG, &v) || v != nseg0) bad++;
      if (nbody_get_info(NBODY_INFO_JSUB, &v) || v != sub0) bad++;
      if (nbody_get_info(NBODY_INFO_WSPLIT, &v) || v != ws0) bad++;
      if (nbody_get_info(NBODY_INFO_MAILBOX_SERVING, &v) || v != 1) bad++;
      int rc;
      switch (k++ % 9) {
        case 0: rc = nbody_step(0.01f, 1); break;
        case 1: rc = nbody_set_option(NBODY_OPT_JSUB, 2); break;
        case 2: rc = nbody_upload(&bs); break;
        case 3: rc = nbody_download(&bs); break;
        case 4: rc = nbody_forces(buf.data(), buf.data(), cap); break;
        case 5: rc = nbody_sync(); break;
        case 6: rc = nbody_mailbox_open(64, 0); break;
        case 7: rc = nbody_init(64, 1, 0, 0); break;
        default: rc = nbody_mailbox_run(ram_a, ram_b, 0); break;
      }
      if (rc != NBODY_ERR_STATE) bad++;
      looked++;
    }
  });
  const int sizes[] = {9, 700, 40, 1, 333, 0, 64};
  for (int k = 0; k < serve_requests; ++k) {
    const int n = sizes[k % 7];
    if (k % 97 == 0) fill(ram_b, cap + 1);
    post(n);
    __atomic_store_n(&ram_a[0], 1u, __ATOMIC_RELEASE);
    while (__atomic_load_n(&ram_a[0], __ATOMIC_ACQUIRE) & 1u) { }
    CHECK(ram_a[3] == 0 && ram_a[1] >= 1);
    if (k % 97 == 0) check_b(ram_b, n, cap + 1);
    else { std::vector<uint32_t> got((uint32_t*)ram_b + 4, (uint32_t*)ram_b + 4 + (size_t)n * 4); CHECK(first[n].empty() || got == first[n]); }
  }
  stop.store(1, std::memory_order_release);
  hammer.join();
  CHECK(bad.load() == 0 && looked.load() > 0);
  long long served = 0;
  for (int spin = 0; spin < 1000000 && served < serve_requests; ++spin) served = info(NBODY_INFO_MAILBOX_SERVED);
  CHECK(served == serve_requests);
  OK(nbody_mailbox_serve(0, 0));
  OK(nbody_sync());                                                                            // the context is the caller's again
  fill(ram_b, cap + 1); post(300); ram_a[0] = 1;
  OK(nbody_mailbox_run(ram_a, ram_b, 0));
  check_b(ram_b, 300, cap + 1);
  OK(nbody_mailbox_serve(1, 0));                                                               // shutdown with the thread still serving
  nbody_shutdown();
  printf("host_sanity ok: %d served requests, %lld looks by the second thread\n", serve_requests, looked.load());
  return 0;
}
