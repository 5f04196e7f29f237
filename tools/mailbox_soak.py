#!/usr/bin/env python3
"""Soak of the mailbox (nbody_mailbox_open / _run / _serve) on the GPU box: for --seconds seconds, requests of random size (0..capacity,
small sizes favoured) through a random form — called on the context's own RAMs, called on the caller's buffers, served by the library's
thread — every answer compared bit for bit with the first answer for that size (and, for the sizes of the exact-rational fixtures, with
the fixture), a sentinel pattern in RAM B checked at word 0 and beyond word N, the service thread switched on and off between bursts.
usage: python tools/mailbox_soak.py [--seconds 20] [--capacity 4096] [--timed]"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--capacity", type=int, default=4096)
    ap.add_argument("--timed", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import mini_nbody_amd as nb
    fixtures = {}
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "rtl_*.json")):
        d = json.load(open(f))
        w = lambda k: np.array([int(x, 16) for x in d[k]], np.uint32).view(np.float32).reshape(-1, 4)   # noqa: E731
        fixtures[d["n"]] = (w("pos0"), w("forces0"))
    rng = np.random.default_rng(args.seed)
    pos_all, _ = nb.make_bodies(args.capacity, seed=77)
    first, count, by_form = {}, 0, {"own": 0, "any": 0, "served": 0}
    sentinel = np.uint32(0xDEADBEEF)
    t_end = time.time() + args.seconds
    with nb.Mailbox(capacity=args.capacity, faithful=not args.timed) as mb:
        ram_b_any = np.empty((args.capacity + 1, 4), np.float32)
        serving = False
        while time.time() < t_end:
            if rng.random() < 0.02:                       # switch the service thread on / off
                serving = not serving
                mb.serve(serving, 300000)
            n = int(rng.choice([0, 9, 40, 100, int(rng.integers(1, 64)), int(rng.integers(1, 1025)), int(rng.integers(1, args.capacity + 1))]))
            use_fixture = n in fixtures and not args.timed
            pos = fixtures[n][0] if use_fixture else pos_all[:n]
            key = ("fx", n) if use_fixture else n
            form = "served" if serving else ("own" if rng.random() < 0.5 else "any")
            if form == "any":
                ram_b_any.view(np.uint32)[...] = sentinel
                out = nb.mailbox.run(mb, nb.mailbox.encode_request(pos), clock_khz=300000, ram_b=ram_b_any)
                rb = ram_b_any.view(np.uint32)
                tail_ok = bool(np.all(rb[0] == sentinel) and np.all(rb[n + 1:] == sentinel))
            else:
                mb.ram_b.view(np.uint32)[...] = sentinel
                mb.post(pos)
                out, ticks = mb.wait() if serving else mb.run(300000)
                rb = mb.ram_b.view(np.uint32)
                tail_ok = bool(np.all(rb[0] == sentinel) and np.all(rb[n + 1:] == sentinel)) and ticks >= 1 and int(mb.ram_a[0, 0]) == 0
            got = out.view(np.uint32).copy()
            if key not in first:
                first[key] = got
                if use_fixture:
                    assert np.array_equal(got, fixtures[n][1].view(np.uint32)), ("fixture", n)
            assert np.array_equal(got, first[key]), (form, n, count)
            assert tail_ok, (form, n, count)
            count += 1
            by_form[form] += 1
        if serving:
            mb.serve(False)
        served = mb.served()
    print("mailbox soak: %d requests in %.0f s (%s arithmetic, capacity %d): %d distinct sizes, by form %s, %d completed by the service thread — "
          "every RAM B image bit-identical to the size's first, fixtures reproduced, RAM B word 0 and the words beyond N untouched, BEGIN cleared, ticks >= 1"
          % (count, args.seconds, "timed" if args.timed else "RTL-faithful", args.capacity, len(first), by_form, served))


if __name__ == "__main__":
    main()
