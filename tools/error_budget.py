#!/usr/bin/env python3
"""Re-association error budget: force error against the fp64 arbiter for the summation
orders the engine offers — one sequential sum (S/top_level.vhd:233-254), segmented sums combined in ascending order
(the multi-GPU / load-balance decomposition), and the FPGA's 16 interleaved partials + pairwise tree
(S/fxyz.vhd:129-184, S/final_adder.vhd:88-104).  Uses the oracle as the checker (tools are not product code).
usage: python tools/error_budget.py [N ...]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import oracle as O
    nb = importlib.import_module("mini-nbody_amd")
    ora = O.Oracle(fast=True)
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 65536]
    print("| N | order / arithmetic | max-norm rel. error vs fp64 | rms rel. error |")
    print("|---|---|---|---|")
    for n in sizes:
        pos, _ = nb.make_bodies(n)
        f64 = ora.forces_f64_from_f32(pos)[:, :3]
        scale = np.abs(f64).max()
        rms = np.sqrt((f64 ** 2).mean())

        def row(name, f):
            d = f[:, :3].astype(np.float64) - f64
            print("| %d | %s | %.2e | %.2e |" % (n, name, np.abs(d).max() / scale, np.sqrt((d ** 2).mean()) / rms), flush=True)

        row("CPU oracle, sequential, 1/sqrt from fp64", ora.forces_f32(pos))
        row("CPU oracle, FPGA order (16 partials + tree)", ora.forces_f32(pos, summ=O.SUM_FPGA16))
        eng = nb.NBody(n)
        try:
            for name, opts in (
                ("GPU fast (v_rsq_f32), 1 segment", {nb.OPT_JSUB: 1}),
                ("GPU fast, default segmentation", {nb.OPT_JSUB: 0}),
                ("GPU fast, 8 slices x 8 (the 8-GPU order)", {nb.OPT_JSUB: 8, nb.OPT_JSLICES: 8}),
                ("GPU fast, FPGA order", {nb.OPT_JSUB: 1, nb.OPT_SUM_ORDER: nb.SUM_FPGA16}),
                ("GPU reference roundings (RTL d2), 1 segment", {nb.OPT_JSUB: 1, nb.OPT_ARITH: nb.ARITH_REFERENCE}),
                ("GPU strict, 1 segment (== CPU oracle bitwise)", {nb.OPT_JSUB: 1, nb.OPT_ARITH: nb.ARITH_STRICT}),
            ):
                for k, v in ((nb.OPT_JSUB, 0), (nb.OPT_JSLICES, 1), (nb.OPT_SUM_ORDER, nb.SUM_SEQ), (nb.OPT_ARITH, nb.ARITH_FMA3)):
                    eng.set_option(k, v)
                for k, v in opts.items():
                    eng.set_option(k, v)
                row(name + " [%d seg]" % eng.config["nseg"], eng.forces(pos))
        finally:
            eng.close()


if __name__ == "__main__":
    main()
