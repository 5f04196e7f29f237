#!/usr/bin/env python3
"""Kernel configuration sweep on one GPU: (variant, bodies per lane, source sub-segments) -> G pairs/s from the
HIP-event time of the force kernels.  One process, interleaved rounds (cdna guide §5.4 rule 24).
usage: python tools/sweep.py [--n N] [--steps K] [--rounds M] [--configs "smem:4:1,lds:2:4,isa1:1:8:0:sum=seq:fuse=0,..."]
config = variant:bodies-per-lane:jsub[:waves-per-SIMD cap][:sum=seq|blocked|fpga16][:blk=K][:fuse=0|1][:long=0|1][:xcd=-1|0|1][:ws=1|4|16][:graph=K][:jsl=P (source slices, as a P-rank job cuts them)][:arith=fma3|reference|strict|refstrict]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--configs", default="")
    ap.add_argument("--wall", action="store_true", help="time whole steps (graph replay, no per-kernel events) instead of the force kernels")
    args = ap.parse_args()
    import mini_nbody_amd as nb
    import numpy as np
    n = args.n
    if args.configs:
        cfgs = []
        for c in args.configs.split(","):
            f = c.split(":")
            kv = tuple(sorted(x for x in f[3:] if "=" in x))
            w = [int(x) for x in f[3:] if "=" not in x]
            cfgs.append((f[0], int(f[1]), int(f[2]), w[0] if w else 0, kv))
    else:
        cfgs = [(v, r, s, 0, ()) for v in ("smem", "lds") for r in (1, 2, 4) for s in (1, 2, 4, 8)]
    pos, vel = nb.make_bodies(n, dtype=np.float64 if args.fp64 else np.float32)
    eng = nb.NBody(n, fp64=args.fp64, tile=args.tile)
    eng.set_option(nb.OPT_TIMING, 0 if args.wall else 1)
    import time
    vmap = {"smem": nb.VARIANT_SMEM, "lds": nb.VARIANT_LDS, "readlane": nb.VARIANT_READLANE, "isa0": nb.VARIANT_ISA, "isa1": nb.VARIANT_ISA, "isa2": nb.VARIANT_ISA,
            "isa3": nb.VARIANT_ISA, "isa4": nb.VARIANT_ISA, "isa5": nb.VARIANT_ISA,   # isa3..8: timing-only diagnostics
            "isa6": nb.VARIANT_ISA, "isa7": nb.VARIANT_ISA, "isa8": nb.VARIANT_ISA, "isa9": nb.VARIANT_ISA, "isa10": nb.VARIANT_ISA,
            "isa11": nb.VARIANT_ISA, "isa12": nb.VARIANT_ISA, "isa13": nb.VARIANT_ISA, "isa14": nb.VARIANT_ISA, "isa15": nb.VARIANT_ISA, "isa16": nb.VARIANT_ISA, "isa17": nb.VARIANT_ISA, "isa18": nb.VARIANT_ISA, "isa19": nb.VARIANT_ISA, "isa20": nb.VARIANT_ISA}
    res = {c: [] for c in cfgs}
    for rnd in range(args.rounds):
        for c in cfgs:
            v, r, s, w, kv = c
            opts = dict(x.split("=") for x in kv)
            eng.set_option(nb.OPT_SUM_ORDER, {"seq": nb.SUM_SEQ, "blocked": nb.SUM_BLOCKED, "fpga16": nb.SUM_FPGA16}[opts.get("sum", "blocked")])
            eng.set_option(nb.OPT_SUM_BLOCK, int(opts.get("blk", 1024)))
            eng.set_option(nb.OPT_FUSE_COMBINE, int(opts.get("fuse", -1)))
            eng.set_option(nb.OPT_ISA_LONG_BUFFERS, int(opts.get("long", -1)))
            eng.set_option(nb.OPT_XCD_MAP, int(opts.get("xcd", -1)))
            eng.set_option(nb.OPT_WSPLIT, int(opts.get("ws", -1)))
            eng.set_option(nb.OPT_GRAPH, int(opts.get("graph", 1)))
            eng.set_option(nb.OPT_JSLICES, int(opts.get("jsl", 0)))
            eng.set_option(nb.OPT_ARITH, {"fma3": nb.ARITH_FMA3, "reference": nb.ARITH_REFERENCE, "strict": nb.ARITH_STRICT,
                                          "refstrict": nb.ARITH_REFERENCE_STRICT}[opts.get("arith", "fma3")])
            eng.set_option(nb.OPT_WAVES_PER_SIMD, w)
            eng.set_option(nb.OPT_VARIANT, vmap[v])
            eng.set_option(nb.OPT_ISA_PHASE, int(v[3:]) if v.startswith("isa") else 1)
            eng.set_option(nb.OPT_IBLOCK, r)
            eng.set_option(nb.OPT_JSUB, s)
            eng.upload(pos, vel)
            eng.step(0.01, 4 if args.wall else 1)
            eng.sync()
            if args.wall:
                t0 = time.perf_counter()
                eng.step(0.01, args.steps)
                eng.sync()
                ms = 1e3 * (time.perf_counter() - t0)
            else:
                eng.kernel_time(reset=True)
                eng.step(0.01, args.steps)
                eng.sync()
                ms, cnt = eng.kernel_time(reset=True)
            res[c].append(float(n) * n * args.steps / (ms * 1e-3) / 1e9)
    bound = 256 * 4 * 64 / 30 * 2.4
    print("# n=%d steps=%d rounds=%d; issue bound %.0f G/s at 2.4 GHz" % (n, args.steps, args.rounds, bound))
    for c in sorted(cfgs, key=lambda c: -max(res[c])):
        print("%-9s R=%d jsub=%-3d waves/SIMD<=%d %-32s best %7.1f  median %7.1f G pairs/s  (%.1f %% of issue bound)  %9.2f us/step"
              % (c[0], c[1], c[2], c[3] or 8, " ".join(c[4]), max(res[c]), sorted(res[c])[len(res[c]) // 2], 100 * max(res[c]) / bound,
                 1e6 * float(n) * n / (max(res[c]) * 1e9)), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
