#!/usr/bin/env python3
"""What a transfer costs beside a force launch that fills the chip (VERDICT r02 "next" 2), on ONE GPU: the RCCL ring step of a
one-rank communicator (ncclSend/ncclRecv to self: a real RCCL kernel on the transfer stream) timed from enqueue to done —
alone, enqueued just before an N-body force pass on the compute stream, and just after it — with the transfer stream at the
device's highest priority and at default priority (NBODY_COMM_PRIORITY=0, a second process).
usage: python tools/comm_probe.py [--n N] [--reps R]      -> markdown table on stdout"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(n, reps, sizes):
    import mini_nbody_amd as nb
    eng = nb.NBody(n, rank=0, nranks=1, uid=nb.unique_id())
    pos, vel = nb.make_bodies(n)
    eng.upload(pos, vel)
    prio = eng.info(nb._lib.INFO_COMM_PRIORITY)
    rows = []
    for nbytes in sizes:
        for when in (0, 1, 2, 3, 4):
            eng.comm_probe(nbytes, when)          # warm (RCCL channel set-up, kernels loaded)
            c, f = zip(*[eng.comm_probe(nbytes, when) for _ in range(reps)])
            rows.append((prio, nbytes, when, min(c), sorted(c)[len(c) // 2], max(c), sorted(f)[len(f) // 2]))
    eng.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    sizes = [2 << 20]     # one rank's slice at N = 1M, P = 8 (2 MiB); about what a rank receives per step (14 MiB) / 2
    rows = measure(args.n, args.reps, sizes)
    if args.child:
        for r in rows:
            print("ROW " + " ".join(repr(x) for x in r))
        return
    env = dict(os.environ, NBODY_COMM_PRIORITY="0")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--n", str(args.n), "--reps", str(args.reps)],
                         env=env, capture_output=True, text=True, timeout=600)
    for l in out.stdout.splitlines():
        if l.startswith("ROW "):
            rows.append(tuple(eval(x) for x in l[4:].split()))
    names = {0: "alone (idle chip)", 1: "enqueued just BEFORE the force pass", 2: "enqueued just AFTER the force pass",
             3: "steady state: released together with the next force pass by the previous one's end (ms from that end)",
             4: "steady state + hand-shake: the next force pass waits until the transfer stream has reached its RCCL kernel"}
    print("| transfer stream priority | bytes | when | ring step min / median / max (ms, enqueue -> done) | force pass beside it (ms) |")
    print("|---|---|---|---|---|")
    for prio, nbytes, when, lo, med, hi, f in rows:
        print("| %s | %d MiB | %s | %.3f / %.3f / %.3f | %s |" % ("highest (%d)" % prio if prio else "default", nbytes >> 20, names[when], lo, med, hi,
                                                             "%.1f" % f if when else "-"))


if __name__ == "__main__":
    main()
