#!/usr/bin/env python3
"""Generate tests/golden/fp64_n64.json — a checked fixture for the fp64 path (BASELINE configs[4]'s arithmetic).

The timed fp64 arithmetic is held to tolerances (its inverse cube is v_rsq_f64 + one third-order step, not the oracle's IEEE chain;
the fp64 NBODY_ARITH_STRICT mode of round 4 IS bit-identical to the oracle, tests/test_golden_fp64.py), and an oracle-vs-engine
comparison in which both round alike cannot see a slip both share.  This fixture is a third statement: F_i = sum_j (r_j - r_i) (|r_j - r_i|^2 + eps)^(-3/2) over all j including i (S/top_level.vhd:233-254,
S/fxyz.vhd:97-127, eps = (double)1e-9f = 0x3089705F widened, S/dzsoft.vhd:177) evaluated in 60-digit decimal arithmetic — every
binary64 input is exact in it — and rounded ONCE to binary64 per component.  Consumers (tests/test_golden_fp64.py) bound the engine's
and the oracle's error per row in ulps of that row's largest component.

Inputs: full-width doubles, x = u1 + u2 * 2^-25 with u1, u2 from the repository's seeded generator (seeds s, s + 1): exact sums, so
the low 29 bits of the significand are exercised (the generator's own values have 24).  Stored as hex words.
Run: python tests/golden/make_fp64.py
"""
import json
import os
import struct
import sys
from decimal import Decimal, getcontext

HERE = os.path.dirname(os.path.abspath(__file__))
SOFT_BITS = 0x3089705F


def f64_hex(x):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", x))[0]


def main(n=64, seed=21):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import mini_nbody_amd as nb
    import numpy as np
    p1, _ = nb.make_bodies(n, seed=seed, dtype=np.float64)
    p2, _ = nb.make_bodies(n, seed=seed + 1, dtype=np.float64)
    pos = p1 + p2 * 2.0 ** -25
    pos[:, 3] = 1.0
    getcontext().prec = 60
    eps = Decimal(struct.unpack("<f", struct.pack("<I", SOFT_BITS))[0])      # exact: a binary32 value
    P = [[Decimal(float(v)) for v in row[:3]] for row in pos]
    forces = []
    for i in range(n):
        acc = [Decimal(0)] * 3
        for j in range(n):
            d = [P[j][c] - P[i][c] for c in range(3)]
            d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + eps
            inv3 = 1 / (d2 * d2.sqrt())
            acc = [acc[c] + d[c] * inv3 for c in range(3)]
        forces.append([float(a) for a in acc] + [0.0])                       # float(Decimal) rounds correctly, once
    fx = {"n": n, "seed": seed, "what": "F_i over all j incl. i, 60-digit decimal evaluation rounded once to binary64",
          "pos0": [f64_hex(float(v)) for row in pos for v in row], "forces0": [f64_hex(v) for row in forces for v in row]}
    json.dump(fx, open(os.path.join(HERE, "fp64_n%d.json" % n), "w"), indent=0)
    print("wrote fp64_n%d.json" % n)


if __name__ == "__main__":
    main()
