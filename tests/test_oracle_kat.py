"""The oracle against the known-answer vectors built from the reference's own
testbench stimuli (tests/golden/make_kat.py; SURVEY.md §4 last row, §8(c)).
Bit-exact for the add/sub/mul/fma stages; rsqrt within 1 ulp of the correctly
rounded value (the IP's rounding is unpinned)."""
import json
import os

import numpy as np
import pytest

import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name + ".json")) as f:
        return json.load(f)


def f(h):
    return np.array([int(h, 16)], np.uint32).view(np.float32)[0]


def bits(x):
    return int(np.array([x], np.float32).view(np.uint32)[0])


def test_soft_constant(oracle):
    # S/dzsoft.vhd:177
    assert bits(oracle.soft()) == 0x3089705F
    assert oracle.soft() == np.float32(1e-9)


def test_kat_dxy(oracle):
    d = load("kat_dxy")
    assert len(d["cases"]) == 101
    for c in d["cases"]:
        s, dx, dy = oracle.dxy(f(c["x_this"]), f(c["x_target"]), f(c["y_this"]), f(c["y_target"]))
        assert (bits(s), bits(dx), bits(dy)) == (int(c["sum"], 16), int(c["dx"], 16), int(c["dy"], 16)), c["label"]


def test_kat_dxyz_soft(oracle):
    d = load("kat_dxyz_soft")
    assert len(d["cases"]) == 6 + 101        # tb_dxyz_soft's 1 + 5, and the 1 + 100 its unfinished successor was about to drive
    for c in d["cases"]:
        this = [f(h) for h in c["this"]]
        tgt = [f(h) for h in c["target"]]
        s, (dx, dy, dz) = oracle.dxyz_soft(this, tgt)
        assert bits(s) == int(c["dist_sqr"], 16), c["label"]
        assert [bits(dx), bits(dy), bits(dz)] == [int(c[k], 16) for k in ("dx", "dy", "dz")]
        assert bits(oracle.d2_fma3(dx, dy, dz)) == int(c["dist_sqr_fma3"], 16)
    # the self-interaction case: d2 == eps exactly, force term exactly zero
    self_case = d["cases"][1]
    assert int(self_case["dist_sqr"], 16) == 0x3089705F


@pytest.mark.parametrize("mode", [O.RSQRT_F64, O.RSQRT_DIVSQRT])
def test_kat_rsqrt(oracle, mode):
    d = load("kat_rsqrt")
    worst = 0
    for c in d["cases"]:
        with np.errstate(all="ignore"):
            r = oracle.rsqrt(f(c["a"]), mode)
        if c["result"] == "nan":
            assert np.isnan(r), c["label"]
            continue
        want = int(c["result"], 16)
        ulp = abs(bits(r) - want)
        worst = max(worst, ulp)
        assert ulp <= c["tol_ulp"], (c["label"], hex(bits(r)), c["result"])
    if mode == O.RSQRT_F64:
        assert worst == 0  # one rounding from an fp64 evaluation == correctly rounded on these inputs


def test_kat_cube_tree(oracle):
    d = load("kat_cube_tree")
    for c in d["cube"]:
        assert bits(oracle.cube(f(c["inv"]))) == int(c["inv3"], 16)
    leaves = np.array([f(h) for h in d["tree16"]["leaves"]], np.float32)
    assert bits(oracle.tree16(leaves)) == int(d["tree16"]["sum"], 16)


def test_self_pair_contributes_zero(oracle):
    # dx=dy=dz=0 => d2 = eps, inv finite, fma(0, inv3, acc) = acc
    p = np.array([[0.3, -0.2, 0.9, 1.0]], np.float32)
    for summ in (O.SUM_SEQ, O.SUM_FPGA16):
        a = oracle.forces_f32(p, summ=summ)
        assert np.all(a == 0) and np.all(np.isfinite(a))


def test_fast_build_is_bit_identical(oracle, oracle_fast):
    pos, _ = oracle.ic(777)
    for d2 in (O.D2_REFERENCE, O.D2_FMA3):
        for rs in (O.RSQRT_F64, O.RSQRT_DIVSQRT):
            for sm in (O.SUM_SEQ, O.SUM_FPGA16):
                a = oracle.forces_f32(pos, d2=d2, rsqrt=rs, summ=sm)
                b = oracle_fast.forces_f32(pos, d2=d2, rsqrt=rs, summ=sm)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (d2, rs, sm)
    p64 = pos.astype(np.float64)
    assert np.array_equal(oracle.forces_f64(p64), oracle_fast.forces_f64(p64))


def test_acc_in_continues_the_sum(oracle):
    # a sequential sum split at any point and carried through memory is the same sum
    pos, _ = oracle.ic(300)
    full = oracle.forces_f32(pos)
    part = oracle.forces_f32(pos, pos[:123])
    both = oracle.forces_f32(pos, pos[123:], acc_in=part)
    assert np.array_equal(full.view(np.uint32), both.view(np.uint32))


def test_config1_cpu_program(oracle_fast):
    """BASELINE config 1: N = 4096 fp32, 10 iterations on the host CPU, as a program (oracle/nbody_cpu.c).  Its checksum
    equals the oracle library's; tests/test_gpu_c_host.py checks that the GPU host program prints the same in --strict."""
    import re
    import subprocess
    import oracle as O
    exe = os.path.join(os.path.dirname(O.__file__), "nbody_cpu")
    assert os.path.exists(exe), "make -C oracle"
    out = subprocess.run([exe, "4096", "10"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"checksum \(sum of positions\): (\S+) (\S+) (\S+)", out.stdout)
    got = np.array([float(m.group(k)) for k in (1, 2, 3)])
    pos, vel = oracle_fast.ic(4096)
    oracle_fast.step(pos, vel, 0.01, 10)
    want = pos[:, :3].astype(np.float64).sum(0)
    assert np.allclose(got, want, rtol=0, atol=1e-6 * np.abs(pos[:, :3]).sum())
    assert re.search(r"4096 Bodies \(fp32, host CPU, \d+ threads\): average \S+ Billion Interactions / second", out.stdout)
