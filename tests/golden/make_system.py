#!/usr/bin/env python3
"""Generate tests/golden/system_*.json — whole-pipeline known answers: seeded bodies -> forces -> positions and
velocities after a few leapfrog steps, in the summation orders the engine and the oracle implement.

The reference holds no expected outputs for its pipeline (its testbenches assert only "not X",
T/tb_dxy.vhd:907-918), so these are NOT reference fixtures; they pin the C oracle (oracle/nbody_ref.c) and the HIP
engine (strict arithmetic) to a THIRD statement of the same arithmetic that shares no code with either: pure Python,
every operation evaluated exactly with fractions.Fraction and rounded once to binary32 where the RTL rounds
(S/dxy.vhd:94-98 sub, S/dzsoft.vhd:201-202 fma, S/cube.vhd:66-70 two mul, S/fxyz.vhd:120-127 fma accumulate; d2 in
the fma-contracted form the timed kernels use; 1/sqrt rounded once from a binary64 evaluation).
Summation order: nslices x sub segments of the sources (balanced slices, ceil pieces), blocks of `block` sources summed
from zero and added in ascending order (block = 0: one sequential sum per segment), segments added in ascending
order — include/nbody.h NBODY_SUM_BLOCKED / oracle/nbody_ref.h ref_order_t.  With wsplit = 4 (NBODY_OPT_WSPLIT, the four
waves of a workgroup) a segment is first cut into 4 pieces of ceil(len / 4) sources, each summed as above on its own, and
the piece sums are added in ascending order to give the segment's sum (a third level).

The RTL-faithful mode (round 4: `rtl_*.json`): d2 with the RTL's own five roundings — dx*dx and dy*dy rounded, their sum
rounded (S/dxy.vhd:113-122), fma(dz, dz, eps) rounded once (S/dzsoft.vhd:201-202), the two joined by one add
(S/dxyz_soft.vhd:149-150) — and the RTL's summation over ONE stream of all N sources (S/top_level.vhd:233-254): partial k
accumulates the sources j = k mod 16 from 0.0 with one fma each (the fma's 16-deep feedback, S/fxyz.vhd:120-145), the
16 in-flight sums are latched rotated, results(t) = partial (n + t) mod 16, or 0.0 where that slot never received an item
(n < 16) (S/fxyz.vhd:147-184; the rotation is what tests/test_fpga_scatter_model.py derives from the control logic), and
joined by the pairwise adder tree ((r0+r1)+(r2+r3))+... (S/final_adder.vhd:88-104).  This is what the engine runs with
NBODY_ARITH_REFERENCE_STRICT + NBODY_SUM_FPGA16 + one segment, and what nbody_mailbox_run returns in that mode.

Inputs are generated here with the repository's seeded generator formula restated (SplitMix64 -> uniform [-1, 1)),
and are stored in the fixture as hex words, so consumers need nothing but the JSON.
Run:  python tests/golden/make_system.py      (about a minute)
"""
import json
import math
import os
import struct
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
SOFT_BITS = 0x3089705F   # S/dzsoft.vhd:177


def bits_f32(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def rnd(q):
    """Fraction -> nearest binary32 (ties to even), as a Python float holding that value"""
    if q == 0:
        return 0.0
    d = float(q)                       # correctly rounded to binary64
    r = bits_f32(f32_bits(d))          # second rounding; may be off by one ulp on a double-rounding pattern
    u = f32_bits(r)
    cands = [r]
    for du in (-1, 1):
        v = u + du
        if 0 <= (v & 0x7FFFFFFF) < 0x7F800000:
            cands.append(bits_f32(v))
    best = min(abs(Fraction(c) - q) for c in cands)
    ties = [c for c in cands if abs(Fraction(c) - q) == best]
    ties.sort(key=lambda c: f32_bits(c) & 1)
    return ties[0]


def fma(a, b, c):
    return rnd(Fraction(a) * Fraction(b) + Fraction(c))


def add(a, b):
    return rnd(Fraction(a) + Fraction(b))


def mul(a, b):
    return rnd(Fraction(a) * Fraction(b))


def rsqrt(d2):
    return rnd(Fraction(1.0 / math.sqrt(d2)))      # binary64 sqrt and divide are correctly rounded; then one rounding to binary32


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def uniform_words(n, seed):
    """n bodies x {x,y,z,w} positions then velocities, as the repository's generator makes them (checked against it by
    tests/test_golden_system.py: the fixture's own input words are compared with make_bodies)"""
    import sys
    root = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, root)
    import mini_nbody_amd as nb
    pos, vel = nb.make_bodies(n, seed=seed)
    return [[float(v) for v in row] for row in pos], [[float(v) for v in row] for row in vel]


def slice_first(q, n, P):
    base, rem = divmod(n, P)
    return q * base + min(q, rem)


def segments(n, nslices, sub):
    out = []
    for q in range(nslices):
        f0, f1 = slice_first(q, n, nslices), slice_first(q + 1, n, nslices)
        piece = (f1 - f0 + sub - 1) // sub
        for t in range(sub):
            b = min(f0 + t * piece, f1)
            out.append((b, min(b + piece, f1)))
    return out


def pieces(sb, se, wsplit):
    piece = (se - sb + wsplit - 1) // wsplit
    out = []
    for w in range(wsplit):
        b = min(sb + w * piece, se)
        out.append((b, min(b + piece, se)))
    return out


def forces(pos, nslices, sub, block, wsplit=1):
    soft = bits_f32(SOFT_BITS)
    n = len(pos)
    out = []
    for i in range(n):
        xi, yi, zi = pos[i][0], pos[i][1], pos[i][2]
        total = None
        for (sb, se) in segments(n, nslices, sub):
            seg = None
            for (jb, je) in pieces(sb, se, wsplit):
                pc = [0.0, 0.0, 0.0]
                step = block if block > 0 else max(1, je - jb)
                blocked = block > 0
                for j0 in range(jb, je, step):
                    a = [0.0, 0.0, 0.0]
                    for j in range(j0, min(j0 + step, je)):
                        dx, dy, dz = add(pos[j][0], -xi), add(pos[j][1], -yi), add(pos[j][2], -zi)
                        d2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, soft)))
                        inv = rsqrt(d2)
                        inv3 = mul(inv, mul(inv, inv))
                        a = [fma(dx, inv3, a[0]), fma(dy, inv3, a[1]), fma(dz, inv3, a[2])]
                    pc = [add(s, x) for s, x in zip(pc, a)] if blocked else a
                seg = pc if seg is None else [add(s, x) for s, x in zip(seg, pc)]
            total = seg if total is None else [add(t, s) for t, s in zip(total, seg)]
        out.append(total + [0.0])
    return out


def d2_reference(dx, dy, dz, soft):
    """the RTL's own rounding points for the softened squared distance"""
    sxy = add(mul(dx, dx), mul(dy, dy))      # S/dxy.vhd:113-122: two products rounded, then their sum
    sz = fma(dz, dz, soft)                   # S/dzsoft.vhd:201-202: fused, one rounding
    return add(sxy, sz)                      # S/dxyz_soft.vhd:149-150


def tree16(r):
    """S/final_adder.vhd:88-104: four levels of pairwise adds over 16 leaves"""
    while len(r) > 1:
        r = [add(r[2 * k], r[2 * k + 1]) for k in range(len(r) // 2)]
    return r[0]


def forces_rtl(pos):
    """every body against ONE stream of all n sources in the RTL's order (module docstring)"""
    soft = bits_f32(SOFT_BITS)
    n = len(pos)
    out = []
    for i in range(n):
        xi, yi, zi = pos[i][0], pos[i][1], pos[i][2]
        part = [[0.0, 0.0, 0.0] for _ in range(16)]
        for j in range(n):                                               # S/top_level.vhd:233-254: all sources, self included, ascending
            dx, dy, dz = add(pos[j][0], -xi), add(pos[j][1], -yi), add(pos[j][2], -zi)     # S/dxy.vhd:94-98: target - this
            inv = rsqrt(d2_reference(dx, dy, dz, soft))
            inv3 = mul(inv, mul(inv, inv))                               # S/cube.vhd:66-70
            k = j % 16                                                   # S/fxyz.vhd:129-145
            part[k] = [fma(dx, inv3, part[k][0]), fma(dy, inv3, part[k][1]), fma(dz, inv3, part[k][2])]
        res = [part[(n + t) % 16] if n - 16 + t >= 0 else [0.0, 0.0, 0.0] for t in range(16)]   # S/fxyz.vhd:147-184
        out.append([tree16([res[t][c] for t in range(16)]) for c in range(3)] + [0.0])
    return out


def step(pos, vel, dt, nslices, sub, block, wsplit=1, rtl=False):
    f = forces_rtl(pos) if rtl else forces(pos, nslices, sub, block, wsplit)
    vel = [[fma(dt, f[i][c], vel[i][c]) for c in range(3)] + [vel[i][3]] for i in range(len(pos))]
    pos = [[fma(vel[i][c], dt, pos[i][c]) for c in range(3)] + [pos[i][3]] for i in range(len(pos))]
    return pos, vel, f


def hexwords(rows):
    return ["%08x" % f32_bits(v) for row in rows for v in row]


def make(name, n, seed, steps, nslices, sub, block, wsplit=1):
    pos, vel = uniform_words(n, seed)
    dt = bits_f32(f32_bits(0.01))
    fx = {"n": n, "seed": seed, "steps": steps, "dt_bits": "%08x" % f32_bits(dt),
          "order": {"nslices": nslices, "sub": sub, "block": block, "summ": "blocked" if block > 0 else "seq", "wsplit": wsplit},
          "arith": "d2 = fma(dx,dx,fma(dy,dy,fma(dz,dz,eps))), inv = (float)(1.0/sqrt((double)d2)), inv3 = inv*(inv*inv), F = fma(d, inv3, F)",
          "pos0": hexwords(pos), "vel0": hexwords(vel)}
    p, v = pos, vel
    for s in range(steps):
        p, v, f = step(p, v, dt, nslices, sub, block, wsplit)
        if s == 0:
            fx["forces0"] = hexwords(f)
    fx["pos"] = hexwords(p)
    fx["vel"] = hexwords(v)
    json.dump(fx, open(os.path.join(HERE, name), "w"), indent=0)
    print("wrote", name)


def make_rtl(name, n, seed, steps):
    pos, vel = uniform_words(n, seed)
    dt = bits_f32(f32_bits(0.01))
    fx = {"n": n, "seed": seed, "steps": steps, "dt_bits": "%08x" % f32_bits(dt),
          "order": {"summ": "fpga16", "d2": "reference", "nslices": 1, "sub": 1},
          "arith": "d2 = ((dx*dx)+(dy*dy)) + fma(dz,dz,eps), inv = (float)(1.0/sqrt((double)d2)), inv3 = inv*(inv*inv), "
                   "partial[j mod 16] = fma(d, inv3, partial[j mod 16]), F = tree16(partial rotated by n mod 16)",
          "pos0": hexwords(pos), "vel0": hexwords(vel)}
    p, v = pos, vel
    for s in range(steps):
        p, v, f = step(p, v, dt, 1, 1, 0, rtl=True)
        if s == 0:
            fx["forces0"] = hexwords(f)
    fx["pos"] = hexwords(p)
    fx["vel"] = hexwords(v)
    json.dump(fx, open(os.path.join(HERE, name), "w"), indent=0)
    print("wrote", name)


if __name__ == "__main__":
    import sys
    which = sys.argv[1] if len(sys.argv) > 1 else "all"      # "wsplit" / "wsplit16" / "rtl": only those fixtures (the others are unchanged)
    if which == "all":
        make("system_n64_seq.json", 64, 42, 10, 1, 1, 0)           # the plain sequential sum, 10 steps (BASELINE config 1's loop, tiny)
        make("system_n200_blocked.json", 200, 7, 3, 1, 3, 64)      # 3 segments of 67/67/66 sources, blocks of 64 + a remainder
        make("system_n150_sharded.json", 150, 9, 2, 4, 2, 64)      # 4 rank slices (38/38/37/37) x 2 pieces: the multi-GPU order
    if which in ("all", "wsplit"):
        make("system_n560_wsplit4.json", 560, 5, 2, 1, 2, 64, wsplit=4)        # 2 segments of 280 = 4 pieces of 70: a block fold + 6 in each
        make("system_n90_wsplit4_sharded.json", 90, 3, 3, 3, 2, 64, wsplit=4)  # 3 rank slices x 2 segments of 15 = pieces of 4, 4, 4, 3
    if which in ("all", "rtl"):
        make_rtl("rtl_n9.json", 9, 11, 2)        # fewer than 16 sources: seven result slots never receive an item
        make_rtl("rtl_n40.json", 40, 12, 2)      # n mod 16 = 8
        make_rtl("rtl_n100.json", 100, 13, 2)    # n mod 16 = 4
    if which in ("all", "wsplit16"):
        make("system_n130_wsplit16.json", 130, 2, 2, 1, 1, 64, wsplit=16)      # one segment, 16 pieces: 14 of 9 sources, one of 4, one empty
