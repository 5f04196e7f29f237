#!/usr/bin/env python3
"""What one request through the reference's mailbox costs on the GPU (nbody_mailbox_run, S/top_level.vhd:184-263), per size and
arithmetic, in ONE context of the RTL's capacity that serves every size in turn (NUM_PTS sampled with every BEGIN):
  own RAMs     the context's pinned RAM A / RAM B (nbody_mailbox_rams): the device reads and writes them itself, no host copy;
               the time of the C call alone (BEGIN raised -> word 0 rewritten)
  any buffers  the caller's own (pageable) images: one host copy each way inside the call
  served       nbody_mailbox_serve: no call at all — BEGIN written into the pinned RAM A, word 0 polled until BEGIN reads 0 (a library
               thread plays the FSM); the time from the store of BEGIN to the load that sees it cleared, polled from Python
and beside it what the RTL itself would take — N + ~250 clocks per 12 bodies (S/top_level.vhd:187-254) at the
300 MHz the tick word is quoted at.
usage (on the GPU box): python tools/mailbox_rate.py [--calls 200]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_call_us(fn, calls):
    fn()
    best = float("inf")
    for _ in range(3):                      # best of three batches: a request is tens of us, the host's noise is not smaller
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        best = min(best, (time.perf_counter() - t0) / calls)
    return 1e6 * best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    args = ap.parse_args()
    import mini_nbody_amd as nb
    print("# nbody_mailbox_run, wall time per request (us); RTL estimate = ceil(N/12) x (N + 250) clocks at 300 MHz")
    print("# one context of capacity 32767 per arithmetic, every size served by it in turn")
    sizes = (9, 100, 1024, 4096, 16384, 32767)
    for name, faithful in (("RTL-faithful (nbody_mailbox_open(., 1))", True), ("timed arithmetic (nbody_mailbox_open(., 0))", False)):
        with nb.Mailbox(faithful=faithful) as mb:
            vp = C.c_void_p
            a_own, b_own = mb.ram_a.ctypes.data_as(vp), mb.ram_b.ctypes.data_as(vp)
            run = mb.lib.nbody_mailbox_run
            for n in sizes:
                pos, _ = nb.make_bodies(n)
                rtl_us = -(-n // 12) * (n + 250) / 300e6 * 1e6
                mb.post(pos)

                def own():
                    mb.ram_a[0, 0] = 1
                    mb.ram_a[0, 1] = n
                    rc = run(a_own, b_own, 300000)
                    assert rc == 0, rc
                us_own = per_call_us(own, args.calls)
                ticks = int(mb.ram_a[0, 1])
                ram_a = nb.mailbox.encode_request(pos)
                ram_b = np.zeros((n + 1, 4), np.float32)
                a_any, b_any = ram_a.ctypes.data_as(vp), ram_b.ctypes.data_as(vp)

                def anyb():
                    ram_a[0, 0] = 1
                    ram_a[0, 1] = n
                    rc = run(a_any, b_any, 300000)
                    assert rc == 0, rc
                us_any = per_call_us(anyb, args.calls)
                assert np.array_equal(ram_b[1:].view(np.uint32), mb.ram_b[1:n + 1].view(np.uint32))
                mb.serve(True, 300000)
                w0 = mb.ram_a[0]

                def served():
                    w0[1] = n
                    w0[0] = 1
                    while w0[0] & 1:
                        pass
                us_srv = per_call_us(served, args.calls)
                mb.serve(False)
                assert np.array_equal(ram_b[1:].view(np.uint32), mb.ram_b[1:n + 1].view(np.uint32)) and int(w0[3]) == 0
                print("N = %5d  %-44s own RAMs %8.1f us  any buffers %8.1f us  served %8.1f us  (%7.1f G pairs/s; ticks word %d)   RTL estimate %10.1f us = %6.0fx" %
                      (n, name, us_own, us_any, us_srv, float(n) * n / us_own / 1e3, ticks, rtl_us, rtl_us / us_own))


if __name__ == "__main__":
    main()
