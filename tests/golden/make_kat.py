#!/usr/bin/env python3
"""Generate tests/golden/kat_*.json — known-answer vectors for the pair stages.

INPUTS are the stimuli of the reference's own testbenches
(/root/reference/vec_add.srcs/sim_1/new/, "T/"), restated here as data:
  kat_dxy        T/tb_dxy.vhd:450,458 (x_this), :575,584 (x_target), :701,710 (y_this), :827,836 (y_target)
  kat_dxyz_soft  T/tb_dxyz_soft.vhd:525 (single), :532 with increments :509-511 (ramp of 5); T/tb_dxyz_soft_new.vhd:520 (single), :527 with
                 increments :505-507 (ramp of 100) — the stimuli of the work-in-progress testbench that does not parse, as written
  kat_rsqrt      T/tb_sqrt.vhd:494 (single), :503 (ramp 0.1 .. 10.0), :528-541 (special values)
The reals are converted to binary32 exactly as the testbench helper does
(real_to_flt, e.g. S/dzsoft.vhd:62-147: normalise, integer(mant * 2**23) with
VHDL's round-to-nearest).  The ramps accumulate `value := value + step` in VHDL
`real` (binary64), restated with Python floats.

EXPECTED OUTPUTS are NOT in the reference (its checkers only assert "not X",
T/tb_dxy.vhd:907-918, T/tb_sqrt.vhd:562-573).  They are computed here from the
IEEE-754 definition of each operation with exact rational arithmetic
(fractions.Fraction + one correctly-rounded conversion per rounding point that
the RTL has), independently of oracle/nbody_ref.c.  For the rsqrt IP the
expected value is the correctly rounded 1/sqrt(x); the IP's own rounding is
unpinned (no .xci in the tree), so consumers compare within 1 ulp.

This script does not read /root/reference at run time; the stimuli are
transcribed above.  Run:  python tests/golden/make_kat.py
"""
import json
import math
import os
import struct
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))


def f32_bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def bits_f32(u: int) -> float:
    return struct.unpack("<f", struct.pack("<I", u))[0]


def hexs(u: int) -> str:
    return "0x%08X" % u


def real_to_flt(x: float) -> int:
    """Restatement of the testbench helper real_to_flt(x, normal, 32, 24)."""
    assert x != 0.0
    sign = 1 if x < 0 else 0
    mant = abs(x)
    exp = 0
    mant_max = 2.0 - 1.0 / float(2 ** 23)
    while mant < 1.0:
        exp -= 1
        mant *= 2.0
    while mant > mant_max:
        exp += 1
        mant /= 2.0
    mant -= 1.0
    v = mant * float(2 ** 23)
    mant_int = int(math.floor(v + 0.5))  # VHDL integer(): round to nearest, ties away from zero
    mant_int &= (1 << 23) - 1            # to_unsigned(mant_int, 23) truncates
    return (sign << 31) | ((exp + 127) << 23) | mant_int


def frac_of_bits(u: int) -> Fraction:
    s = -1 if u >> 31 else 1
    e = (u >> 23) & 0xFF
    m = u & 0x7FFFFF
    assert e != 0xFF
    if e == 0:
        return s * Fraction(m, 1 << 149)
    return s * Fraction((1 << 23) | m, 1) * Fraction(2) ** (e - 150)


def round_f32(q: Fraction) -> int:
    """Correctly rounded (nearest-even) binary32 bit pattern of a rational."""
    if q == 0:
        return 0
    sign = 1 if q < 0 else 0
    q = abs(q)
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    assert Fraction(2) ** e <= q < Fraction(2) ** (e + 1)
    e = max(e, -126)
    scaled = q / Fraction(2) ** (e - 23)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    if n >= (1 << 24):
        n >>= 1
        e += 1
    if n < (1 << 23):  # subnormal
        return (sign << 31) | n
    assert e <= 127
    return (sign << 31) | ((e + 127) << 23) | (n & 0x7FFFFF)


SOFT = 0x3089705F
assert real_to_flt(1.0e-9) == SOFT  # S/dzsoft.vhd:177


def op(fn, *bits):
    return round_f32(fn(*[frac_of_bits(b) for b in bits]))


def sub(a, b):
    return op(lambda x, y: x - y, a, b)


def add(a, b):
    return op(lambda x, y: x + y, a, b)


def mul(a, b):
    return op(lambda x, y: x * y, a, b)


def fma(a, b, c):
    return op(lambda x, y, z: x * y + z, a, b, c)


def rsqrt_cr(u: int) -> int:
    """Correctly rounded 1/sqrt(x) for a positive finite binary32 x."""
    x = frac_of_bits(u)
    K = 240
    num = x.denominator << (2 * K)
    s = math.isqrt(num // x.numerator)
    q = Fraction(s, 1 << K)
    if Fraction(s * s * x.numerator, 1) != Fraction(num, 1):
        q += Fraction(1, 1 << (K + 2))  # irrational: nudge off any representable point, far below 1/2 ulp
    return round_f32(q)


def kat_dxy():
    cases = []

    def case(label, xi, xt, yi, yt):
        dx, dy = sub(xt, xi), sub(yt, yi)        # S/dxy.vhd:94-98 (a = target, b = this)
        s = add(mul(dx, dx), mul(dy, dy))         # S/dxy.vhd:113-122
        cases.append(dict(label=label, x_this=hexs(xi), x_target=hexs(xt), y_this=hexs(yi), y_target=hexs(yt),
                          dx=hexs(dx), dy=hexs(dy), sum=hexs(s)))

    case("single", real_to_flt(2.0), real_to_flt(1.0), real_to_flt(1.0), real_to_flt(1.0))
    xi, xt = 1.0, 0.0
    for k in range(100):
        # k = 0 drives x_target = 0.0 through real_to_flt(.., normal), which the helper itself rejects
        # (assert x /= 0.0); the vector keeps it as +0.
        xtb = real_to_flt(xt) if xt != 0.0 else 0
        case("ramp[%d]" % k, real_to_flt(xi), xtb, real_to_flt(1.0), real_to_flt(1.0))
        xi += 2.0
        xt += 1.0
    for c in cases[1:]:
        k = int(c["label"][5:-1])
        assert bits_f32(int(c["sum"], 16)) == float((k + 1) ** 2)
    return dict(source="T/tb_dxy.vhd:450,458,575,584,701,710,827,836", entity="S/dxy.vhd:94-122", cases=cases)


def kat_dxyz_soft():
    cases = []

    def case(label, this, target):
        xi, yi, zi = this
        xt, yt, zt = target
        dx, dy, dz = sub(xt, xi), sub(yt, yi), sub(zt, zi)
        sxy = add(mul(dx, dx), mul(dy, dy))                  # S/dxy.vhd:113-122
        sz = fma(dz, dz, SOFT)                               # S/dzsoft.vhd:201-202
        d2_ref = add(sxy, sz)                                # S/dxyz_soft.vhd:149-150
        d2_fma3 = fma(dx, dx, fma(dy, dy, fma(dz, dz, SOFT)))  # the GPU kernel's contraction
        cases.append(dict(label=label, this=[hexs(v) for v in this], target=[hexs(v) for v in target],
                          dx=hexs(dx), dy=hexs(dy), dz=hexs(dz), dist_sqr=hexs(d2_ref), dist_sqr_fma3=hexs(d2_fma3)))

    r = real_to_flt
    case("single", (r(2.0), r(3.0), r(4.0)), (r(1.0), r(1.0), r(1.0)))
    v1, v3, v5 = 1.0, 1.0, 1.0
    for k in range(5):
        case("ramp[%d]" % k, (r(v1), r(v3), r(v5)), (r(1.0), r(1.0), r(1.0)))
        v1 += 1.0
        v3 += 2.0
        v5 += 3.0
    # The work-in-progress copy of that testbench, T/tb_dxyz_soft_new.vhd — it does not parse —, carries the stimuli its
    # author was about to run: one procedure driving all six operands (:451-466; it assigns y_this twice and z_this never: read as meant,
    # this = (data1, data3, data5), target = (data2, data4, data6)), a single operation this = (1, 2, 3), target = (0, 0, 0) (:520) and a
    # ramp of 100 with the three `this` components incremented by 2.0 (:527 with :505-507).  The zero targets go through real_to_flt(..,
    # normal), which the helper itself rejects; the vectors keep them as +0, as for tb_dxy.
    case("new:single", (r(1.0), r(2.0), r(3.0)), (0, 0, 0))
    n1, n3, n5 = 1.0, 2.0, 3.0
    for k in range(100):
        case("new:ramp[%d]" % k, (r(n1), r(n3), r(n5)), (0, 0, 0))
        n1 += 2.0
        n3 += 2.0
        n5 += 2.0
    for c in cases[7:]:
        k = int(c["label"][9:-1])
        # small integers: every product and sum is exact, eps vanishes in the rounding of dz^2 + eps
        assert bits_f32(int(c["dist_sqr"], 16)) == float((1 + 2 * k) ** 2 + (2 + 2 * k) ** 2 + (3 + 2 * k) ** 2) == bits_f32(int(c["dist_sqr_fma3"], 16))
    assert bits_f32(int(cases[6]["dist_sqr"], 16)) == 14.0
    want = [14.0, None, 14.0, 56.0, 126.0, 224.0]
    for c, w in zip(cases, want):
        if w is None:
            assert int(c["dist_sqr"], 16) == SOFT          # the self-interaction case: d2 = eps exactly
        else:
            assert bits_f32(int(c["dist_sqr"], 16)) == w
    return dict(source="T/tb_dxyz_soft.vhd:509-511,525,532; T/tb_dxyz_soft_new.vhd:451-466,505-507,520,527", entity="S/dxyz_soft.vhd:87-93,149-150", cases=cases)


def kat_rsqrt():
    cases = []
    one = real_to_flt(1.0)
    cases.append(dict(label="single", a=hexs(one), result=hexs(rsqrt_cr(one)), tol_ulp=1))
    v = 0.1
    for k in range(100):
        a = real_to_flt(v)
        cases.append(dict(label="ramp[%d]" % k, a=hexs(a), result=hexs(rsqrt_cr(a)), tol_ulp=1))
        v += 0.1
    special = [("plus_zero", 0x00000000, 0x7F800000), ("minus_zero", 0x80000000, 0xFF800000),
               ("plus_inf", 0x7F800000, 0x00000000), ("minus_inf", 0xFF800000, "nan"),
               ("nan", 0x7FC00000, "nan"), ("plus_one", one, one), ("minus_one", real_to_flt(-1.0), "nan")]
    for label, a, res in special:
        cases.append(dict(label="special:" + label, a=hexs(a), result=res if res == "nan" else hexs(res), tol_ulp=0))
    assert cases[0]["result"] == hexs(one)
    return dict(source="T/tb_sqrt.vhd:494,503,528-541", entity="S/fxyz.vhd:101-102 (IP rsqrt)", cases=cases)


def kat_cube_tree():
    """No reference testbench drives cube or final_adder ("No testbench for cube, fxyz,
    final_adder ...").  These vectors are this build's own, from the RTL's structure."""
    cases = []
    for a in (1.0, 0.5, 31622.776, 0.26726124, 3.0):
        u = f32_bits(a)
        cases.append(dict(inv=hexs(u), inv3=hexs(mul(u, mul(u, u)))))          # S/cube.vhd:66-70
    leaves = [f32_bits(float(v)) for v in
              (1.0, 1e-8, -1.0, 3.5, 1e8, -1e8, 0.1, 0.2, 0.3, 7.0, -2.5, 1e-3, 16777216.0, 1.0, -16777216.0, 1.0)]
    lvl = leaves
    while len(lvl) > 1:
        lvl = [add(lvl[2 * j], lvl[2 * j + 1]) for j in range(len(lvl) // 2)]   # S/final_adder.vhd:88-104
    return dict(source="build-defined (no reference testbench)", entity="S/cube.vhd:66-70; S/final_adder.vhd:88-104",
                cube=cases, tree16=dict(leaves=[hexs(v) for v in leaves], sum=hexs(lvl[0])))


def main():
    for name, fn in (("kat_dxy", kat_dxy), ("kat_dxyz_soft", kat_dxyz_soft), ("kat_rsqrt", kat_rsqrt),
                     ("kat_cube_tree", kat_cube_tree)):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1)
            f.write("\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
