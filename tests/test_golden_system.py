"""Whole-pipeline known answers (tests/golden/system_*.json, made by tests/golden/make_system.py: a pure-Python
exact-rational statement of the arithmetic and of the engine's summation order that shares no code with the C oracle
or the HIP kernels).  On CPU the oracle must reproduce them bit for bit; on the GPU the engine in strict arithmetic
must — with no oracle in the loop."""
import glob
import json
import os

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "system_*.json")))


def words(hexlist):
    return np.array([int(h, 16) for h in hexlist], np.uint32).view(np.float32).reshape(-1, 4)


def load(path):
    fx = json.load(open(path))
    fx["dt"] = float(np.array([int(fx["dt_bits"], 16)], np.uint32).view(np.float32)[0])
    for k in ("pos0", "vel0", "forces0", "pos", "vel"):
        fx[k] = words(fx[k])
    return fx


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_fixtures_present():
    assert len(FIXTURES) >= 6
    assert sum(1 for f in FIXTURES if json.load(open(f))["order"].get("wsplit", 1) == 4) >= 2   # the wave-split order too
    assert sum(1 for f in FIXTURES if json.load(open(f))["order"].get("wsplit", 1) == 16) >= 1  # and 16-wave workgroups


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_inputs_are_the_seeded_generator(nb, path):
    fx = load(path)
    pos, vel = nb.make_bodies(fx["n"], seed=fx["seed"])
    assert np.array_equal(bits(pos), bits(fx["pos0"])) and np.array_equal(bits(vel), bits(fx["vel0"]))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_the_fixture(oracle, oracle_fast, path):
    fx = load(path)
    o = fx["order"]
    for ora in (oracle, oracle_fast):
        order = O.order(summ=O.SUM_BLOCKED if o["summ"] == "blocked" else O.SUM_SEQ, block=o["block"] or 1024,
                        nslices=o["nslices"], sub=o["sub"], wsplit=o.get("wsplit", 1))
        f = ora.forces_order(fx["pos0"], order_=order)
        assert np.array_equal(bits(f), bits(fx["forces0"]))
        p, v = fx["pos0"].copy(), fx["vel0"].copy()
        ora.step_order(p, v, fx["dt"], fx["steps"], order_=order)
        assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_engine_reproduces_the_fixture(nb, path):
    """strict arithmetic, the fixture's order, every delivery variant and both combine forms; then the device loop"""
    fx = load(path)
    o = fx["order"]
    eng = nb.NBody(fx["n"])
    try:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED if o["summ"] == "blocked" else nb.SUM_SEQ)
        if o["block"]:
            eng.set_option(nb.OPT_SUM_BLOCK, o["block"])
        eng.set_option(nb.OPT_JSLICES, o["nslices"])
        eng.set_option(nb.OPT_JSUB, o["sub"])
        ws = o.get("wsplit", 1)
        eng.set_option(nb.OPT_WSPLIT, ws)
        # the wave-split order exists in the scalar-delivery kernel with one body per lane; the plain one in all of them
        shapes = ((nb.VARIANT_SMEM, 1),) if ws > 1 else ((nb.VARIANT_SMEM, 1), (nb.VARIANT_SMEM, 4), (nb.VARIANT_LDS, 2), (nb.VARIANT_READLANE, 2))
        for variant, iblock in shapes:
            for fuse in (1, 0):
                eng.set_option(nb.OPT_VARIANT, variant)
                eng.set_option(nb.OPT_IBLOCK, iblock)
                eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
                assert eng.config["wsplit"] == ws
                assert np.array_equal(bits(eng.forces(fx["pos0"])), bits(fx["forces0"])), (variant, iblock, fuse)
                eng.upload(fx["pos0"], fx["vel0"])
                eng.step(fx["dt"], fx["steps"])
                p, v = eng.download()
                assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"])), (variant, iblock, fuse)
        # host-pointer bodyForce()/integrate()
        p, v = fx["pos0"].copy(), fx["vel0"].copy()
        for _ in range(fx["steps"]):
            eng.bodyForce(p, v, fx["dt"])
            eng.integrate(p, v, fx["dt"])
        assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"]))
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["system_n150_sharded.json", "system_n90_wsplit4_sharded.json"])
def test_virtual_ranks_reproduce_the_sharded_fixture(nb, monkeypatch, name):
    """the sharded fixtures are the orders of a 4-rank job (4 slices x 2 pieces, waves walking whole segments) and of a
    3-rank job with the wave split: that many virtual ranks on this GPU must give them"""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    fx = load(os.path.join(HERE, "golden", name))
    o = fx["order"]
    for overlap in (1, 2, 0):
        eng = nb.NBody(fx["n"], ngpus=o["nslices"])
        try:
            eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
            eng.set_option(nb.OPT_SUM_BLOCK, o["block"])
            eng.set_option(nb.OPT_JSUB, o["sub"])
            eng.set_option(nb.OPT_WSPLIT, o.get("wsplit", 1))
            eng.set_option(nb.OPT_OVERLAP, overlap)
            assert eng.config["wsplit"] == o.get("wsplit", 1)
            assert np.array_equal(bits(eng.forces(fx["pos0"])), bits(fx["forces0"]))
            eng.upload(fx["pos0"], fx["vel0"])
            eng.step(fx["dt"], fx["steps"])
            p, v = eng.download()
            assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"])), overlap
        finally:
            eng.close()


# ---- the RTL-faithful mode (round 4): the reference's own five roundings for d2 and its own summation — 16 interleaved
# partial sums per axis over ONE stream of all N sources, latched rotated, joined by the adder tree — pinned by the third
# statement (tests/golden/make_system.py forces_rtl: exact rationals, no code shared with oracle or kernels)
RTL_FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "rtl_*.json")))


def test_rtl_fixtures_present():
    ns = sorted(json.load(open(f))["n"] for f in RTL_FIXTURES)
    assert ns == [9, 40, 100]                       # below 16 (empty result slots), and two with n mod 16 != 0
    assert all(json.load(open(f))["order"] == {"summ": "fpga16", "d2": "reference", "nslices": 1, "sub": 1} for f in RTL_FIXTURES)


@pytest.mark.parametrize("path", RTL_FIXTURES, ids=[os.path.basename(p) for p in RTL_FIXTURES])
def test_oracle_reproduces_the_rtl_fixture(nb, oracle, oracle_fast, path):
    fx = load(path)
    pos, vel = nb.make_bodies(fx["n"], seed=fx["seed"])
    assert np.array_equal(bits(pos), bits(fx["pos0"])) and np.array_equal(bits(vel), bits(fx["vel0"]))
    for ora in (oracle, oracle_fast):
        f = ora.forces_f32(fx["pos0"], d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(bits(f), bits(fx["forces0"]))
        p, v = fx["pos0"].copy(), fx["vel0"].copy()
        ora.step(p, v, fx["dt"], fx["steps"], d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"]))
        # and the fma-contracted d2 of the timed kernels is a DIFFERENT function: the fixture distinguishes the two modes
        g = ora.forces_f32(fx["pos0"], d2=O.D2_FMA3, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert not np.array_equal(bits(g), bits(fx["forces0"]))


@pytest.mark.gpu
@pytest.mark.parametrize("wsplit", [-1, 1], ids=["sixteen-waves", "one-lane"])
@pytest.mark.parametrize("path", RTL_FIXTURES, ids=[os.path.basename(p) for p in RTL_FIXTURES])
def test_engine_and_mailbox_reproduce_the_rtl_fixture(nb, path, wsplit):
    """NBODY_ARITH_REFERENCE_STRICT + NBODY_SUM_FPGA16 + one segment — the three options INTEGRATION.md §1 lists as "RTL-faithful
    result" — through nbody_forces, through the reference's own RAM images (nbody_mailbox_run) and through the step loop:
    bit for bit the third statement's answer, no oracle in the loop"""
    fx = load(path)
    eng = nb.NBody(fx["n"])
    try:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
        eng.set_option(nb.OPT_JSUB, 1)
        eng.set_option(nb.OPT_WSPLIT, wsplit)     # the sixteen partial sums on sixteen waves (automatic) or in one lane: same bits
        assert eng.config["nseg"] == 1 and eng.config["sum_order"] == "fpga16" and eng.config["wsplit"] == (16 if wsplit < 0 else 1)
        assert np.array_equal(bits(eng.forces(fx["pos0"])), bits(fx["forces0"]))
        ram_a = nb.mailbox.encode_request(fx["pos0"])
        ram_b = nb.mailbox.run(eng, ram_a, clock_khz=300000)
        assert nb.mailbox.decode_control(ram_a)["begin"] == 0
        assert np.array_equal(bits(ram_b), bits(fx["forces0"])) and np.all(ram_b[:, 3] == 0)
        for fuse in (1, 0):
            eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
            eng.upload(fx["pos0"], fx["vel0"])
            eng.step(fx["dt"], fx["steps"])
            p, v = eng.download()
            assert np.array_equal(bits(p), bits(fx["pos"])) and np.array_equal(bits(v), bits(fx["vel"])), fuse
        # the mailbox with the context's DEFAULT options is the timed arithmetic (fma-contracted d2, v_rsq_f32, blocked sums):
        # within tolerance of the RTL-faithful answer, not equal to it — which is why INTEGRATION.md names the three options
        eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED)
        eng.set_option(nb.OPT_JSUB, 0)
        fast = nb.mailbox.run(eng, nb.mailbox.encode_request(fx["pos0"]))
        d = np.abs(fast[:, :3].astype(np.float64) - fx["forces0"][:, :3]).max() / np.abs(fx["forces0"][:, :3]).max()
        assert 0 < d < 1e-5
    finally:
        eng.close()
