"""The reference's own testbench stimuli for the distance stage, THROUGH THE GPU: T/tb_dxy.vhd (1 + 100 vectors), T/tb_dxyz_soft.vhd (1 + 5,
incl. the self pair d2 = eps) and the 1 + 100 its unfinished successor T/tb_dxyz_soft_new.vhd was about to drive (tests/golden/kat_dxy.json,
kat_dxyz_soft.json; the testbenches assert only "not X", the expected values are analytic).  The library has no entry point for d2 alone —
the stage is inside the pair — so each vector becomes a two-body system {this, target} and the force on `this` is compared, bit for bit, in
both strict arithmetics (the RTL's five roundings for d2 and the kernel's three fused ones) and in the RTL's summation order, with
  (a) the oracle, and
  (b) a value built from the fixture's own d2 and dx, dy, dz — exact rationals, one rounding per operation the RTL rounds
      (tests/golden/make_kat.py), sharing no code with the oracle or the kernels: inv = the once-rounded binary64 1/sqrt of the FIXTURE's d2,
      inv3 = inv * (inv * inv) (S/cube.vhd:66-70), F = fma(d, inv3, 0) (S/fxyz.vhd:120-127; the self pair adds an exact zero first).
So a GPU d2 that differed from the fixture's by one bit would show."""
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_kat", os.path.join(HERE, "golden", "make_kat.py"))
K = importlib.util.module_from_spec(spec)
spec.loader.exec_module(K)


def f32(h):
    return np.array([int(h, 16)], np.uint32).view(np.float32)[0]


def vectors():
    out = []
    for c in json.load(open(os.path.join(HERE, "golden", "kat_dxy.json")))["cases"]:
        out.append(("dxy:" + c["label"], (c["x_this"], c["y_this"], "0x00000000"), (c["x_target"], c["y_target"], "0x00000000"), None))
    for c in json.load(open(os.path.join(HERE, "golden", "kat_dxyz_soft.json")))["cases"]:
        out.append(("dxyz_soft:" + c["label"], tuple(c["this"]), tuple(c["target"]), c))
    return out


def expected_from_fixture(c, ref_d2):
    """F on `this` from the fixture's own numbers: exact rationals, one rounding where the RTL rounds"""
    d2 = int(c["dist_sqr" if ref_d2 else "dist_sqr_fma3"], 16)
    inv = np.float32(1.0 / np.sqrt(np.float64(f32("0x%08X" % d2))))            # the strict 1/sqrt: binary64 sqrt and divide, rounded once
    ib = int(np.array([inv], np.float32).view(np.uint32)[0])
    inv3 = K.mul(ib, K.mul(ib, ib))
    return [K.fma(int(c[k], 16), inv3, 0) for k in ("dx", "dy", "dz")]


def test_the_testbench_stimuli_as_two_body_systems(nb, oracle):
    vecs = vectors()
    assert len(vecs) == 101 + 6 + 101
    checked_fixture = 0
    with nb.NBody(2) as eng:
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
        eng.set_option(nb.OPT_JSUB, 1)
        for arith, d2mode, ref_d2 in ((nb.ARITH_REFERENCE_STRICT, O.D2_REFERENCE, True), (nb.ARITH_STRICT, O.D2_FMA3, False)):
            eng.set_option(nb.OPT_ARITH, arith)
            for label, this, target, c in vecs:
                pos = np.zeros((2, 4), np.float32)
                pos[0, :3] = [f32(h) for h in this]
                pos[1, :3] = [f32(h) for h in target]
                pos[:, 3] = 1.0
                got = eng.forces(pos)
                want = oracle.forces_f32(pos, pos, d2=d2mode, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (label, arith)
                if c is not None:
                    exp = expected_from_fixture(c, ref_d2)
                    assert [int(v) for v in got[0, :3].view(np.uint32)] == exp, (label, arith)
                    checked_fixture += 1
                    if c["label"] == "ramp[0]":                      # this == target: the self-interaction case, d2 = eps exactly, force exactly zero
                        assert not got.any()
    assert checked_fixture == 2 * 107
