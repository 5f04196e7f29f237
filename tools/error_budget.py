#!/usr/bin/env python3
"""Re-association error budget: force error against an fp64 evaluation for the summation
orders the engine offers — one sequential fp32 sum per body (S/top_level.vhd:233-254; what a plain CPU nbody.c does),
segmented sums combined in ascending order, the FPGA's 16 interleaved partials + pairwise tree (S/fxyz.vhd:129-184,
S/final_adder.vhd:88-104) and the engine's default: blocks of 1024 sources, two levels (NBODY_SUM_BLOCKED).
Row-sampled above N = 65536 (all N sources, `--rows` bodies in windows spread over the shards).  Two norms per line:
max-norm (largest component error over the sample / largest force component of the sample) and the worst row
(|dF_i| / |F_i|).  Uses the oracle as the checker (tools are not product code); GPU lines need a GPU.
usage: python tools/error_budget.py [--rows R] [--no-gpu] [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def windows(n, rows):
    """(first, count) windows: the first and last rows of each of 8 shards + the middle, `rows` bodies in all"""
    if rows >= n:
        return [(0, n)]
    w = max(16, rows // 17)
    per = n // 8
    out = []
    for q in range(8):
        out += [(q * per, w), ((q + 1) * per - w, w)]
    out.append((n // 2 - w // 2, w))
    return out


def main():
    import oracle as O
    import mini_nbody_amd as nb
    ora = O.Oracle(fast=True)
    args = sys.argv[1:]
    rows = 2048
    use_gpu = True
    if "--rows" in args:
        rows = int(args[args.index("--rows") + 1])
        del args[args.index("--rows"):args.index("--rows") + 2]
    if "--no-gpu" in args:
        use_gpu = False
        args.remove("--no-gpu")
    sizes = [int(a) for a in args] or [4096, 65536]
    print("| N | rows checked | order / arithmetic | max-norm rel. error vs fp64 | worst row |dF|/|F| | median row |")
    print("|---|---|---|---|---|---|")
    for n in sizes:
        pos, _ = nb.make_bodies(n)
        win = windows(n, rows if n > 65536 else n)
        idx = np.concatenate([np.arange(f, f + c) for f, c in win])
        sample = pos[idx]
        f64 = ora.forces_f64_from_f32(sample, pos)[:, :3]
        scale = np.abs(f64).max()
        rowmag = np.sqrt((f64 ** 2).sum(1))

        def row(name, f):
            d = f[:, :3].astype(np.float64) - f64
            per = np.sqrt((d ** 2).sum(1)) / rowmag
            print("| %d | %d | %s | %.2e | %.2e | %.2e |" % (n, len(idx), name, np.abs(d).max() / scale, per.max(), np.median(per)), flush=True)

        row("CPU oracle, ONE sequential fp32 sum (plain nbody.c), 1/sqrt from fp64", ora.forces_f32(sample, pos))
        row("CPU oracle, sequential, 1.0f/sqrtf", ora.forces_f32(sample, pos, rsqrt=O.RSQRT_DIVSQRT))
        row("CPU oracle, FPGA order (16 partials + tree)", ora.forces_f32(sample, pos, summ=O.SUM_FPGA16))
        row("CPU oracle, sequential in 8 segments", ora.forces_order(sample, pos, summ=O.SUM_SEQ, sub=8))
        row("CPU oracle, blocked 1024, 1 segment", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED))
        row("CPU oracle, blocked 1024, 8 segments", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, sub=8))
        row("CPU oracle, blocked 1024, 8 x 8 segments (the 8-GPU order)", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, nslices=8, sub=8))
        row("CPU oracle, blocked 256, 8 segments", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, block=256, sub=8))
        # round 3: the wave split adds a level — segments x pieces (4 or 16 waves of a workgroup), blocks inside a piece
        row("CPU oracle, blocked 1024, 8 segments x 4 pieces (round 3's order at N = 1M)", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, sub=8, wsplit=4))
        row("CPU oracle, blocked 1024, 16 segments x 4 pieces (N = 16384 ... 65536)", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, sub=16, wsplit=4))
        row("CPU oracle, blocked 1024, 4 segments x 16 pieces (N <= 4096)", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, sub=4, wsplit=16))
        row("CPU oracle, blocked 1024, 8 x 4 segments x 4 pieces (round 3's 8-GPU order at N = 1M)", ora.forces_order(sample, pos, summ=O.SUM_BLOCKED, nslices=8, sub=4, wsplit=4))
        if not use_gpu:
            continue
        eng = nb.NBody(n)
        try:
            eng.upload(pos, np.zeros_like(pos))

            def gpu(opts):
                for k, v in ((nb.OPT_JSUB, 0), (nb.OPT_JSLICES, 1), (nb.OPT_SUM_ORDER, nb.SUM_BLOCKED), (nb.OPT_ARITH, nb.ARITH_FMA3), (nb.OPT_WSPLIT, -1)):
                    eng.set_option(k, v)
                for k, v in opts.items():
                    eng.set_option(k, v)
                return np.concatenate([eng.forces_rows(f, c) for f, c in win])

            for name, opts in (
                ("GPU timed mode (v_rsq_f32), DEFAULT configuration", {}),
                ("GPU v_rsq_f32, blocked, 8 slices x 4 segments x 4 pieces (the 8-GPU order at N = 1M)", {nb.OPT_JSLICES: 8, nb.OPT_JSUB: 4, nb.OPT_WSPLIT: 4}),
                ("GPU v_rsq_f32, round 2's layout (every wave walks the whole segment), default segmentation", {nb.OPT_WSPLIT: 1}),
                ("GPU v_rsq_f32, ONE sequential sum", {nb.OPT_JSUB: 1, nb.OPT_SUM_ORDER: nb.SUM_SEQ}),
                ("GPU v_rsq_f32, sequential, default segmentation (round 1's timed path)", {nb.OPT_SUM_ORDER: nb.SUM_SEQ}),
                ("GPU v_rsq_f32, FPGA order, 1 segment", {nb.OPT_JSUB: 1, nb.OPT_SUM_ORDER: nb.SUM_FPGA16}),
                ("GPU RTL roundings for d2 (REFERENCE), default configuration", {nb.OPT_ARITH: nb.ARITH_REFERENCE}),
                ("GPU strict, default configuration (== CPU oracle bitwise)", {nb.OPT_ARITH: nb.ARITH_STRICT}),
            ):
                f = gpu(opts)
                row(name + " [%d seg x %d pieces, %s]" % (eng.config["nseg"], eng.config["wsplit"], eng.config["sum_order"]), f)
        finally:
            eng.close()


if __name__ == "__main__":
    main()
