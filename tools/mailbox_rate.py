#!/usr/bin/env python3
"""What one request through the reference's mailbox costs on the GPU (nbody_mailbox_run: RAM A image in host memory -> RAM B image in
host memory, S/top_level.vhd:184-263), per size and arithmetic: wall time per call (host copies included: the boundary hands over host
buffers), and beside it what the RTL itself would take — N + ~250 clocks per 12 bodies (S/top_level.vhd:187-254)
at the 300 MHz the testbenches' tick counter assumes.
usage (on the GPU box): python tools/mailbox_rate.py [--calls 20]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=20)
    args = ap.parse_args()
    nb = importlib.import_module("mini-nbody_amd")
    print("# nbody_mailbox_run, wall time per request incl. host copies; RTL estimate = ceil(N/12) x (N + 250) clocks at 300 MHz")
    for n in (1024, 4096, 16384, 32767):
        pos, _ = nb.make_bodies(n)
        rtl_ms = -(-n // 12) * (n + 250) / 300e6 * 1e3
        eng = nb.NBody(n)
        try:
            for name, opts in (("RTL-faithful (REFERENCE_STRICT + FPGA16 + JSUB 1)", ((nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT), (nb.OPT_SUM_ORDER, nb.SUM_FPGA16), (nb.OPT_JSUB, 1))),
                               ("the same, sixteen partial sums in one lane (rounds 1-3)", ((nb.OPT_WSPLIT, 1),)),
                               ("context defaults (timed arithmetic)", ((nb.OPT_WSPLIT, -1), (nb.OPT_ARITH, nb.ARITH_FMA3), (nb.OPT_SUM_ORDER, nb.SUM_BLOCKED), (nb.OPT_JSUB, 0)))):
                for k, v in opts:
                    eng.set_option(k, v)
                nb.mailbox.run(eng, nb.mailbox.encode_request(pos))
                t0 = time.perf_counter()
                for _ in range(args.calls):
                    ram_a = nb.mailbox.encode_request(pos)
                    nb.mailbox.run(eng, ram_a, clock_khz=300000)
                ms = 1e3 * (time.perf_counter() - t0) / args.calls
                ticks = nb.mailbox.decode_control(ram_a)["ticks"]
                print("N = %5d  %-58s %8.3f ms per request (%7.1f G pairs/s; ticks word %d)   RTL estimate %9.2f ms = %5.0fx" %
                      (n, name, ms, float(n) * n / ms / 1e6, ticks, rtl_ms, rtl_ms / ms))
        finally:
            eng.close()


if __name__ == "__main__":
    main()
