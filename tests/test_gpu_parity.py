"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

The engine's summation order is part of the contract: sources are cut into segments, a segment is summed in
blocks of K = 1024 sources (two levels), segments are added in ascending order (include/nbody.h NBODY_SUM_BLOCKED;
the reference's own remedy for long fp32 sums is 16 partials + a tree, S/fxyz.vhd:129-145).  The oracle restates
exactly that order (oracle.order(...), REF_SUM_BLOCKED), so every test below runs the configuration the engine
chooses BY ITSELF — the one bench.py times — unless it says otherwise.

Bars, all written out here:
  * STRICT arithmetic (NBODY_ARITH_STRICT / _REFERENCE_STRICT): every operation is IEEE-exact, so the GPU must
    equal the oracle BIT FOR BIT — forces, and positions/velocities after any number of steps, for every delivery
    variant, register blocking, segmentation and block length.
  * FAST arithmetic (v_rsq_f32, <= 1 ulp; the timed mode): 1e-5 (the north_star's tolerance) on the force of EVERY
    row, relative to that row's own force, against the same-order oracle AND against an fp64 evaluation; after one
    step on every body's velocity and position relative to the terms of its own update.  After many steps the
    dynamics (eps = 1e-9, close pairs) amplify any last-bit difference: there the bar is bit-exactness in strict
    mode, and for the fast mode a comparison with the spread of two CPU runs whose 1/sqrt differ by <= 1 ulp.
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: "within 1e-5 relative fp32 tolerance"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def maxnorm_rel(a, b):
    return float(np.abs(a[:, :3].astype(np.float64) - b[:, :3]).max() / np.abs(b[:, :3]).max())


def row_rel(a, b):
    """per row: |a_i - b_i|_2 / |b_i|_2 — the error of each body's force relative to that body's own force"""
    a64, b64 = a[:, :3].astype(np.float64), b[:, :3].astype(np.float64)
    return np.sqrt(((a64 - b64) ** 2).sum(1)) / np.maximum(np.sqrt((b64 ** 2).sum(1)), 1e-300)


def oracle_forces(ora, eng, rows, src=None, d2=O.D2_FMA3, rsqrt=O.RSQRT_F64):
    """the oracle in the order the engine is configured for"""
    return ora.forces_order(rows, src, order_=O.order(d2=d2, rsqrt=rsqrt, **eng.order))


def oracle_step(ora, eng, pos, vel, dt, steps, rsqrt=O.RSQRT_F64):
    ora.step_order(pos, vel, dt, steps, order_=O.order(rsqrt=rsqrt, **eng.order))


@pytest.fixture()
def engine_factory(nb):
    made = []

    def make(n, **kw):
        for e in made:
            e.close()
        e = nb.NBody(n, **kw)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()


VARIANTS = ["smem", "lds", "readlane"]


def set_variant(nb, eng, variant, iblock, jsub=0, jslices=1, arith=None, wsplit=-1):
    """wsplit: -1 = the engine's choice (4 where the kernel has the wave split: smem / isa with one body per lane), 1 = every
    wave walks the whole segment (the only layout of the lds / readlane kernels and of 2+ bodies per lane)"""
    eng.set_option(nb.OPT_WSPLIT, wsplit)
    eng.set_option(nb.OPT_VARIANT, {"smem": nb.VARIANT_SMEM, "lds": nb.VARIANT_LDS, "readlane": nb.VARIANT_READLANE,
                                    "isa": nb.VARIANT_ISA, "auto": nb.VARIANT_AUTO}[variant])
    eng.set_option(nb.OPT_IBLOCK, iblock)
    eng.set_option(nb.OPT_JSUB, jsub)
    eng.set_option(nb.OPT_JSLICES, jslices)
    if arith is not None:
        eng.set_option(nb.OPT_ARITH, arith)


def test_random_configurations_strict_bit_exact(nb, oracle_fast, engine_factory):
    """A seeded walk through the configuration space — size (1 … 6000, ragged), delivery variant and bodies per lane, pieces per slice,
    source slices, wave split, summation order and block length, combine form, XCD placement, d2 form — 48 draws: in strict arithmetic
    forces, a window of rows and three steps of the device loop equal the oracle in the order the engine reports, bit for bit.  Whatever
    combination the options resolve to (some are overridden: the wave split exists for one body per lane only), the engine's `order` must
    describe what it did."""
    import os
    rng = np.random.default_rng(int(os.environ.get("NBODY_TEST_SEED", "20241004")))
    blocks = (64, 256, 1024, 4096)
    for draw in range(int(os.environ.get("NBODY_TEST_DRAWS", "48"))):      # (a soak run sets these: profiles/r04_random_configurations.txt)
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 1024, 1025, 2085, 4096, 5000, 6000])) if draw % 3 else int(rng.integers(1, 6001))
        variant = str(rng.choice(["auto", "smem", "lds", "readlane"]))
        iblock = int(rng.choice({"auto": [0], "smem": [1, 2, 4, 8], "lds": [1, 2, 4], "readlane": [1, 2, 4]}[variant]))
        jsub, jsl = int(rng.choice([0, 1, 2, 3, 5, 8])), int(rng.choice([1, 1, 2, 3]))
        if n < jsl:
            jsl = 1
        wsplit = int(rng.choice([-1, 1, 4, 16]))
        summ = str(rng.choice(["blocked", "blocked", "seq"]))
        block, fuse, xcd = int(rng.choice(blocks)), int(rng.choice([1, 0])), int(rng.choice([0, 1]))
        arith, d2 = ((nb.ARITH_STRICT, O.D2_FMA3), (nb.ARITH_REFERENCE_STRICT, O.D2_REFERENCE))[int(rng.integers(0, 2))]
        pos, vel = nb.make_bodies(n, seed=1000 + draw)
        eng = engine_factory(n)
        set_variant(nb, eng, variant, iblock, jsub=jsub, jslices=jsl, arith=arith, wsplit=wsplit)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED if summ == "blocked" else nb.SUM_SEQ)
        eng.set_option(nb.OPT_SUM_BLOCK, block)
        eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
        eng.set_option(nb.OPT_XCD_MAP, xcd)
        what = (draw, n, variant, iblock, jsub, jsl, wsplit, summ, block, fuse, xcd, arith, eng.config)
        order = O.order(d2=d2, **eng.order)
        f = eng.forces(pos)
        assert np.array_equal(bits(f), bits(oracle_fast.forces_order(pos, order_=order))), what
        r0, cnt = (n // 3, max(1, min(200, n - n // 3)))
        eng.upload(pos, vel)
        assert np.array_equal(bits(eng.forces_rows(r0, cnt)), bits(f[r0:r0 + cnt])), what
        eng.step(0.01, 3)
        gp, gv = eng.download()
        op, ov = pos.copy(), vel.copy()
        oracle_fast.step_order(op, ov, 0.01, 3, order_=order)
        assert np.array_equal(bits(gp), bits(op)) and np.array_equal(bits(gv), bits(ov)), what


def test_random_configurations_fpga_order(nb, oracle_fast, engine_factory):
    """The same walk for the reference's own summation order (NBODY_SUM_FPGA16): size, pieces per slice, source slices, combine form, XCD
    placement and all four arithmetics, 32 draws.  The sixteen partial sums on sixteen waves (automatic) and in one lane (NBODY_OPT_WSPLIT
    1) are the same operations in the same order: forces, a window of rows and three steps of the device loop must agree bit for bit in
    EVERY arithmetic, the timed one included; with one segment and strict arithmetic both equal the oracle's REF_SUM_FPGA16."""
    import os
    rng = np.random.default_rng(int(os.environ.get("NBODY_TEST_SEED", "20241004")) + 1)
    for draw in range(int(os.environ.get("NBODY_TEST_DRAWS", "32"))):
        n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 257, 1000, 1024, 1025, 2085, 4096, 5000])) if draw % 3 else int(rng.integers(1, 6001))
        jsub, jsl = int(rng.choice([1, 1, 2, 3, 5])), int(rng.choice([1, 1, 2, 3]))      # explicit: the automatic count depends on the kernel's geometry
        if n < jsl:
            jsl = 1
        fuse, xcd = int(rng.choice([1, 0])), int(rng.choice([0, 1]))
        arith = int(rng.choice([nb.ARITH_FMA3, nb.ARITH_REFERENCE, nb.ARITH_STRICT, nb.ARITH_REFERENCE_STRICT]))
        pos, vel = nb.make_bodies(n, seed=3000 + draw)
        eng = engine_factory(n)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
        eng.set_option(nb.OPT_ARITH, arith)
        eng.set_option(nb.OPT_JSUB, jsub)
        eng.set_option(nb.OPT_JSLICES, jsl)
        eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
        eng.set_option(nb.OPT_XCD_MAP, xcd)
        got, nsegs = {}, {}
        for ws in (-1, 1):
            eng.set_option(nb.OPT_WSPLIT, ws)
            what = (draw, n, jsub, jsl, fuse, xcd, arith, ws, eng.config)
            nsegs[ws] = eng.config["nseg"]
            assert eng.config["wsplit"] == (16 if ws < 0 else 1) and eng.config["sum_order"] == "fpga16", what
            f = eng.forces(pos)
            r0, cnt = (n // 3, max(1, min(200, n - n // 3)))
            eng.upload(pos, vel)
            assert np.array_equal(bits(eng.forces_rows(r0, cnt)), bits(f[r0:r0 + cnt])), what
            eng.step(0.01, 3)
            got[ws] = (f,) + tuple(eng.download())
            if eng.config["nseg"] == 1 and arith in (nb.ARITH_STRICT, nb.ARITH_REFERENCE_STRICT):
                d2 = O.D2_REFERENCE if arith == nb.ARITH_REFERENCE_STRICT else O.D2_FMA3
                assert np.array_equal(bits(f), bits(oracle_fast.forces_f32(pos, d2=d2, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16))), what
        assert nsegs[-1] == nsegs[1], what
        for a, b in zip(got[-1], got[1]):
            assert np.array_equal(bits(a), bits(b)), what


def test_defaults_are_the_timed_configuration(nb, engine_factory):
    eng = engine_factory(1 << 16)
    cfg = eng.config
    assert cfg["variant"] == "isa" and cfg["iblock"] == 1 and cfg["sum_order"] == "blocked" and cfg["sum_block"] == 1024
    assert cfg["wsplit"] == 4 and eng.order["wsplit"] == 4            # 64-row workgroups, four waves split the segment
    assert cfg["launches_per_step"] == 1 and cfg["nseg"] > 1          # several segments, still one launch per step
    small = engine_factory(4096)                                      # few workgroups per CU: the hand-off would be exposed
    assert small.config["launches_per_step"] == 2 and small.config["variant"] == "isa"
    small.set_option(nb.OPT_FUSE_COMBINE, 1)
    assert small.config["launches_per_step"] == 1


@pytest.mark.parametrize("n", [1, 2, 63, 64, 257, 1000, 1024, 1025, 2085, 5000])
def test_strict_forces_bit_exact_ragged_sizes(nb, oracle_fast, engine_factory, n):
    """auto segmentation, blocked sums: every delivery variant, 1 and 4 bodies per lane"""
    pos, _ = nb.make_bodies(n, seed=n + 1)
    eng = engine_factory(n)
    for variant in VARIANTS:
        for iblock in (1, 4):
            set_variant(nb, eng, variant, iblock, arith=nb.ARITH_STRICT)
            want = oracle_forces(oracle_fast, eng, pos)
            assert np.array_equal(bits(eng.forces(pos)), bits(want)), (variant, iblock, eng.config)
    set_variant(nb, eng, "smem", 2, arith=nb.ARITH_REFERENCE_STRICT)
    want_ref = oracle_forces(oracle_fast, eng, pos, d2=O.D2_REFERENCE)
    assert np.array_equal(bits(eng.forces(pos)), bits(want_ref))
    # the plain sequential sum (what a CPU nbody.c does) stays available: one segment, one accumulator
    eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_SEQ)
    eng.set_option(nb.OPT_JSUB, 1)
    assert eng.config["sum_order"] == "seq"
    assert np.array_equal(bits(eng.forces(pos)), bits(oracle_fast.forces_f32(pos, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64)))


@pytest.mark.parametrize("block", [64, 256, 1024, 4096])
def test_blocked_sum_block_lengths(nb, oracle_fast, engine_factory, block):
    """block boundaries inside tiles (K < tile), across tiles, K > segment: all variants, strict bit-exact"""
    n = 6000 + 13
    pos, _ = nb.make_bodies(n, seed=17)
    eng = engine_factory(n, tile=512)
    eng.set_option(nb.OPT_SUM_BLOCK, block)
    for variant, iblock, jsub in (("smem", 1, 1), ("smem", 4, 3), ("lds", 2, 2), ("readlane", 2, 5)):
        set_variant(nb, eng, variant, iblock, jsub=jsub, arith=nb.ARITH_STRICT)
        assert eng.config["sum_block"] == block
        want = oracle_forces(oracle_fast, eng, pos)
        assert np.array_equal(bits(eng.forces(pos)), bits(want)), (variant, iblock, jsub)
    # the hand-scheduled loop folds its blocks inside the asm loop: same bits as the compiled kernel
    set_variant(nb, eng, "isa", 0, jsub=2, arith=nb.ARITH_FMA3)
    assert eng.config["variant"] == "isa"
    a = eng.forces(pos)
    set_variant(nb, eng, "smem", 1, jsub=2, arith=nb.ARITH_FMA3)
    assert np.array_equal(bits(a), bits(eng.forces(pos)))


@pytest.mark.parametrize("tile", [256, 512, 1024])
def test_strict_lds_tiles(nb, oracle_fast, engine_factory, tile):
    n = 3000
    pos, _ = nb.make_bodies(n, seed=5)
    eng = engine_factory(n, tile=tile)
    set_variant(nb, eng, "lds", 2, arith=nb.ARITH_STRICT)
    assert np.array_equal(bits(eng.forces(pos)), bits(oracle_forces(oracle_fast, eng, pos)))


def test_fast_variants_agree_bitwise_and_within_tolerance(nb, oracle_fast, engine_factory):
    """All delivery variants and register blockings run the same operations in the same order (with every wave walking the
    whole segment, the one layout they all have); and with the wave split the two kernels that have it agree."""
    n = 4096 + 37
    pos, _ = nb.make_bodies(n, seed=11)
    eng = engine_factory(n)
    ref = None
    for variant in VARIANTS:
        for iblock in (1, 2, 4):
            set_variant(nb, eng, variant, iblock, jsub=1, arith=nb.ARITH_FMA3, wsplit=1)
            assert eng.config["wsplit"] == 1
            got = eng.forces(pos)
            if ref is None:
                ref = got
            assert np.array_equal(bits(got), bits(ref)), (variant, iblock)
    set_variant(nb, eng, "smem", 8, jsub=1, arith=nb.ARITH_FMA3, wsplit=1)
    assert np.array_equal(bits(eng.forces(pos)), bits(ref))
    # the hand-scheduled ISA loop (both code-placement phases): same operations, same order, same bits
    for phase in (0, 1):
        for ws in (1, 4):
            set_variant(nb, eng, "isa", 0, jsub=1, wsplit=ws)
            eng.set_option(nb.OPT_ISA_PHASE, phase)
            assert eng.config["variant"] == "isa" and eng.config["iblock"] == 1 and eng.config["wsplit"] == ws
            if ws == 1:
                assert np.array_equal(bits(eng.forces(pos)), bits(ref)), phase
            for jsub, jsl in ((1, 1), (3, 1), (2, 5)):
                set_variant(nb, eng, "isa", 0, jsub=jsub, jslices=jsl, wsplit=ws)
                got = eng.forces(pos)
                set_variant(nb, eng, "smem", 1, jsub=jsub, jslices=jsl, wsplit=ws)
                assert eng.config["wsplit"] == ws
                assert np.array_equal(bits(got), bits(eng.forces(pos))), (phase, ws, jsub, jsl)
    eng.set_option(nb.OPT_ISA_PHASE, 1)
    set_variant(nb, eng, "smem", 1, jsub=1, wsplit=1)
    want = oracle_forces(oracle_fast, eng, pos)
    f64 = oracle_fast.forces_f64_from_f32(pos)
    assert row_rel(ref, want).max() < TOL and row_rel(ref, f64).max() < TOL
    assert np.all(ref[:, 3] == 0)   # S/compute_store.vhd:242: the 4th word is 0


@pytest.mark.parametrize("ws", [4, 16])
def test_wave_split_third_level_bit_exact(nb, oracle_fast, engine_factory, ws):
    """NBODY_OPT_WSPLIT = 4 / 16: a workgroup owns 64 rows, wave w walks piece w of the segment, the sums are added through
    LDS in ascending order.  Strict arithmetic against the oracle's order with that wsplit (ref_order_t), bit for bit: ragged
    sizes incl. segments shorter than the wave count (empty pieces), several slices, both combine forms, block folds inside
    pieces; the hand-scheduled loop agrees with the compiled kernel; and the forces differ from the wsplit = 1 order only by
    re-association."""
    for n in (1, 3, 5, 64, 65, 257, 1000, 6013):        # (2, 63, 4099 too until round 4: the random-configuration test covers such sizes now)
        pos, vel = nb.make_bodies(n, seed=300 + n)
        eng = engine_factory(n)
        eng.set_option(nb.OPT_SUM_BLOCK, 64)
        for jsub, jsl in ((0, 1), (1, 1), (3, 1), (2, 3)):
            if n < jsl:
                continue
            for fuse in (1, 0):
                set_variant(nb, eng, "smem", 1, jsub=jsub, jslices=jsl, arith=nb.ARITH_STRICT, wsplit=ws)
                eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
                assert eng.config["wsplit"] == ws and eng.order["wsplit"] == ws
                got = eng.forces(pos)
                want = oracle_forces(oracle_fast, eng, pos)
                assert np.array_equal(bits(got), bits(want)), (n, jsub, jsl, fuse)
                eng.upload(pos, vel)
                eng.step(0.01, 3)
                gp, gv = eng.download()
                op, ov = pos.copy(), vel.copy()
                oracle_step(oracle_fast, eng, op, ov, 0.01, 3)
                assert np.array_equal(bits(gp), bits(op)) and np.array_equal(bits(gv), bits(ov)), (n, jsub, jsl, fuse)
            # the timed arithmetic: hand-scheduled loop (both buffer lengths) == compiled kernel
            out = []
            for variant, long_buffers in (("smem", -1), ("isa", 0), ("isa", 1)):
                set_variant(nb, eng, variant, 1 if variant == "smem" else 0, jsub=jsub, jslices=jsl, arith=nb.ARITH_FMA3, wsplit=ws)
                eng.set_option(nb.OPT_ISA_LONG_BUFFERS, long_buffers)
                assert eng.config["wsplit"] == ws
                eng.upload(pos, vel)
                eng.step(0.01, 2)
                out.append((eng.forces(pos),) + eng.download())
            for o in out[1:]:
                for x, y in zip(o, out[0]):
                    assert np.array_equal(bits(x), bits(y)), (n, jsub, jsl)
            eng.set_option(nb.OPT_ISA_LONG_BUFFERS, -1)
        if n >= 1000:
            set_variant(nb, eng, "smem", 1, jsub=2, arith=nb.ARITH_STRICT, wsplit=ws)
            a = eng.forces(pos)
            set_variant(nb, eng, "smem", 1, jsub=2, arith=nb.ARITH_STRICT, wsplit=1)
            assert eng.order["wsplit"] == 1
            b = eng.forces(pos)
            assert not np.array_equal(bits(a), bits(b)) and row_rel(a, b).max() < 1e-5
    # fp64: the same layouts, hand-scheduled loop == compiled kernel, and both within fp64 tolerance of the oracle
    n = 3001
    pos, vel = nb.make_bodies(n, seed=9, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    eng.set_option(nb.OPT_WSPLIT, ws)
    eng.set_option(nb.OPT_JSUB, 2)
    assert eng.config["wsplit"] == ws and eng.config["variant"] == "isa"
    a = eng.forces(pos)
    eng.set_option(nb.OPT_VARIANT, nb.VARIANT_SMEM)
    eng.set_option(nb.OPT_IBLOCK, 1)
    assert eng.config["wsplit"] == ws and eng.config["variant"] == "smem"
    assert np.array_equal(bits(a), bits(eng.forces(pos)))
    assert maxnorm_rel(a, oracle_fast.forces_f64(pos)) < 1e-13


def test_diagnostic_library_encodings_are_bit_identical(nb, tmp_path):
    """`make diag` (libnbody_hip_diag.so, -DNBODY_DIAG_LOOPS): the experiment encodings of the hand-scheduled loop (2 staggered
    loads, 9..13 32-bit encodings, 12 round 1's loop, 16 v_subrev, 17 packed subtraction, 18 eps from a VGPR) run the same
    operations in the same order as the product loop — same bits, with and without the wave split.  Skipped when the
    diagnostic library has not been built (it is not part of the product build)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "mini_nbody_amd", "libnbody_hip_diag.so")
    if not os.path.exists(lib):
        pytest.fail("libnbody_hip_diag.so is missing: __graft_entry__.build() and tests/conftest.py both run `make diag`")
    script = tmp_path / "diag.py"
    script.write_text("""
import sys
import numpy as np
sys.path.insert(0, %r)
import mini_nbody_amd as nb
n = 4096 + 37
pos, _ = nb.make_bodies(n, seed=11)
eng = nb.NBody(n)
assert eng.info(nb._lib.INFO_DIAG_BUILD) == 1
for ws in (1, 4, 16):
    eng.set_option(nb.OPT_WSPLIT, ws)
    for jsub in (1, 3):
        eng.set_option(nb.OPT_JSUB, jsub)
        eng.set_option(nb.OPT_ISA_PHASE, 1)
        ref = eng.forces(pos)
        # (16-wave workgroups exist for the product loop and its placement twin; the other forms would run with 4 waves,
        #  which is another summation order)
        for phase in ((0,) if ws == 16 else (0, 2, 9, 10, 11, 12, 13, 16, 17, 18, 19, 20)):
            eng.set_option(nb.OPT_ISA_PHASE, phase)
            assert eng.config["variant"] == "isa" and eng.config["isa_phase"] == phase
            assert np.array_equal(eng.forces(pos).view(np.uint32), ref.view(np.uint32)), (ws, jsub, phase)
for phase in (3, 4, 5, 6, 7, 8, 14, 15):      # the timing-only forms are accepted here (and compute garbage)
    eng.set_option(nb.OPT_ISA_PHASE, phase)
eng.close()
print("diag ok")
""" % root)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, NBODY_LIB=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "diag ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_diagnostic_loop_forms_are_not_in_the_product_library(nb, engine_factory):
    """NBODY_OPT_ISA_PHASE 2..18 (experiment encodings; 3..8, 14, 15 are timing-only forms with WRONG results) exist only in
    the diagnostic build (make diag): the product library refuses them and keeps running the product loop."""
    n = 2000
    pos, _ = nb.make_bodies(n, seed=1)
    eng = engine_factory(n)
    assert eng.info(nb._lib.INFO_DIAG_BUILD) == 0
    ref = eng.forces(pos)
    for phase in range(2, 19):
        with pytest.raises(nb.NBodyError) as e:
            eng.set_option(nb.OPT_ISA_PHASE, phase)
        assert e.value.code == nb._lib.ERR_UNSUPPORTED, phase
        assert eng.config["isa_phase"] == 1
    assert np.array_equal(bits(eng.forces(pos)), bits(ref))
    eng.set_option(nb.OPT_ISA_PHASE, 0)
    assert np.array_equal(bits(eng.forces(pos)), bits(ref))


@pytest.mark.parametrize("n", [1, 7, 8, 9, 15, 16, 17, 31, 33, 100, 1031, 1024, 2047, 2056, 3000])
@pytest.mark.parametrize("summ", ["blocked", "seq"])
@pytest.mark.parametrize("long_buffers", [0, 1])
def test_isa_loop_equals_compiled_kernel_ragged(nb, engine_factory, n, summ, long_buffers):
    """Default (hand-scheduled ISA loop, groups of 8 sources + scalar tail, block folds inside the loop; and its
    long-buffer form for launches with few waves per SIMD, groups of 16) vs the hipcc-scheduled kernel: same bits —
    forces and three steps, one segment and the auto segmentation."""
    pos, vel = nb.make_bodies(n, seed=100 + n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ISA_LONG_BUFFERS, long_buffers)
    eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED if summ == "blocked" else nb.SUM_SEQ)
    eng.set_option(nb.OPT_SUM_BLOCK, 64)            # several blocks even at these sizes
    for jsub in (1, 0):
        set_variant(nb, eng, "auto", 0, jsub=jsub)
        assert eng.config["variant"] == "isa"
        a = eng.forces(pos)
        eng.upload(pos, vel)
        eng.step(0.01, 3)
        pa, va = eng.download()
        set_variant(nb, eng, "smem", 1, jsub=jsub)
        assert eng.config["variant"] == "smem"
        assert np.array_equal(bits(a), bits(eng.forces(pos)))
        eng.upload(pos, vel)
        eng.step(0.01, 3)
        pb, vb = eng.download()
        assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(va), bits(vb))


def test_segmentation_matches_host_mirror_bitwise(nb, oracle_fast, engine_factory):
    """jslices x jsub segments combined in ascending order == the Python mirror of the decomposition
    (mini_nbody_amd/sharding.py) driven by the oracle segment by segment, and == the oracle's own order function."""
    n = 5000
    pos, _ = nb.make_bodies(n, seed=3)
    eng = engine_factory(n)
    for nsl, sub in ((1, 4), (3, 1), (8, 2), (7, 3)):
        set_variant(nb, eng, "smem", 4, jsub=sub, jslices=nsl, arith=nb.ARITH_STRICT)
        assert eng.config["nseg"] == nsl * sub
        got = eng.forces(pos)
        parts = []
        for q in range(nsl):
            for t in range(sub):
                b, e = nb.sharding.segment_bounds(q, t, n, nsl, sub)
                parts.append(oracle_fast.forces_f32(pos, pos[b:e], summ=O.SUM_BLOCKED))
        want = nb.sharding.combine_ascending(parts)
        assert np.array_equal(bits(got), bits(want)), (nsl, sub)
        assert np.array_equal(bits(got), bits(oracle_forces(oracle_fast, eng, pos))), (nsl, sub)


def test_xcd_aware_segment_placement_changes_no_bit(nb, oracle_fast, engine_factory, monkeypatch):
    """NBODY_OPT_XCD_MAP (default on for launches with a multiple of 8 segment rows): workgroups that share an XCD take
    the same source segments, so each XCD's L2 fetches them once (block_segment in nbody_kernels.hpp).  Only who computes
    what changes: forces, steps (both launch forms), row windows and the 8-virtual-rank schedule give the bits of the plain
    (blockIdx.x, blockIdx.y) placement, for every delivery variant, ragged sizes and fp64."""
    for n, fp64 in ((257, False), (5000, False), (20000, False), (70001, False), (5000, True)):
        dtype = np.float64 if fp64 else np.float32
        pos, vel = nb.make_bodies(n, seed=3 * n, dtype=dtype)
        eng = engine_factory(n, fp64=fp64)
        shapes = (("auto", 0, 8, 1), ("auto", 0, 64, 1), ("auto", 0, 3, 8), ("smem", 2, 16, 1), ("lds", 1, 8, 2), ("readlane", 2, 8, 1))
        for variant, iblock, jsub, jsl in shapes if not fp64 else shapes[:3]:
            for fuse in (1, 0):
                out = {}
                for xcd in (1, 0):
                    set_variant(nb, eng, variant, iblock, jsub=jsub, jslices=jsl)
                    eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
                    eng.set_option(nb.OPT_XCD_MAP, xcd)
                    f = eng.forces(pos)
                    eng.upload(pos, vel)
                    w = eng.forces_rows(n // 3, min(300, n - n // 3))
                    eng.step(0.01, 7)
                    out[xcd] = (f, w) + eng.download()
                for x, y in zip(out[1], out[0]):
                    assert np.array_equal(bits(x), bits(y)), (n, fp64, variant, iblock, jsub, jsl, fuse)
        if not fp64 and n == 20000:     # and it is what the oracle says, in the engine's order
            set_variant(nb, eng, "auto", 0, jsub=8, jslices=1)
            eng.set_option(nb.OPT_XCD_MAP, 1)
            eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
            assert np.array_equal(bits(eng.forces(pos)), bits(oracle_forces(oracle_fast, eng, pos)))
    # the multi-GPU schedule on 8 virtual ranks: own-slice launch (8 segment rows) + the launch over the 7 others (56)
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n = 40000
    pos, vel = nb.make_bodies(n, seed=11)
    out = {}
    for xcd in (1, 0):
        eng = engine_factory(n, ngpus=8)
        eng.set_option(nb.OPT_XCD_MAP, xcd)
        assert eng.config["nseg"] % 8 == 0
        eng.upload(pos, vel)
        eng.step(0.01, 5)
        out[xcd] = eng.download()
    assert np.array_equal(bits(out[1][0]), bits(out[0][0])) and np.array_equal(bits(out[1][1]), bits(out[0][1]))


def test_xcd_map_with_one_two_or_four_segment_rows(nb, engine_factory):
    """wg_coords' second branch — launches with 1, 2 or 4 segment rows, where XCD x takes row x mod Y and the 8 / Y XCDs
    of a row deal its row blocks (launch_force allows it when the row-block count divides evenly).  These are the shapes
    wsplit = 16 with 1, 2, 4 segments produces by itself.  Sizes with a multiple of 8 row blocks of 64 rows, ragged last block;
    a 512-row window (8 row blocks) for nbody_forces_rows; both combine forms; 4 and 16 waves; fp64."""
    for n, fp64 in ((1020, False), (4090, False), (20470, False), (4090, True)):
        assert ((n + 63) // 64) % 8 == 0
        dtype = np.float64 if fp64 else np.float32
        pos, vel = nb.make_bodies(n, seed=n, dtype=dtype)
        eng = engine_factory(n, fp64=fp64)
        for wsplit in (4, 16):
            for jsub in (1, 2, 4):
                for fuse in (1, 0):
                    out = {}
                    for xcd in (1, 0):
                        set_variant(nb, eng, "auto", 0, jsub=jsub, jslices=1)
                        eng.set_option(nb.OPT_WSPLIT, wsplit)
                        eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
                        eng.set_option(nb.OPT_XCD_MAP, xcd)
                        cfg = eng.config
                        assert cfg["wsplit"] == wsplit and cfg["nseg"] == jsub
                        f = eng.forces(pos)
                        eng.upload(pos, vel)
                        w = eng.forces_rows(192, 512)
                        eng.step(0.01, 5)
                        out[xcd] = (f, w) + eng.download()
                    for x, y in zip(out[1], out[0]):
                        assert np.array_equal(bits(x), bits(y)), (n, fp64, wsplit, jsub, fuse)
                    assert np.array_equal(bits(out[1][1]), bits(out[1][0][192:704]))


def test_one_launch_combine_equals_combine_kernel(nb, engine_factory):
    """The in-launch combine (last-arriving workgroup adds the partial sums, finish_rows in nbody_kernels.hpp) against the
    two-launch form (combine_kernel): same bits for every variant and shape, and over many steps — the partial buffers are
    rewritten every step, so a stale read of another workgroup's partial would show as a mismatch."""
    for n, steps in ((300, 20), (4099, 30), (20000, 40), (70001, 12)):
        pos, vel = nb.make_bodies(n, seed=n)
        eng = engine_factory(n)
        for variant, iblock, jsub, jsl in (("auto", 0, 0, 1), ("smem", 4, 5, 3), ("lds", 2, 7, 1), ("readlane", 2, 2, 8), ("auto", 0, 64, 1)):
            out = {}
            for fuse in (1, 0):
                set_variant(nb, eng, variant, iblock, jsub=jsub, jslices=jsl)
                eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
                assert eng.config["launches_per_step"] == (1 if fuse or eng.config["nseg"] == 1 else 2)
                f = eng.forces(pos)
                eng.upload(pos, vel)
                eng.step(0.01, steps)
                p1, v1 = eng.download()
                eng.step(0.01, 3)                      # odd continuation: eager launches after the graph pairs
                out[fuse] = (f,) + (p1, v1) + eng.download()
            for x, y in zip(out[1], out[0]):
                assert np.array_equal(bits(x), bits(y)), (n, variant, iblock, jsub, jsl)
    # a long run under load (many resident workgroups per CU, thousands of hand-offs per step, partial buffers and
    # tickets reused every step): any stale or torn read of a partial sum changes the final bits
    for n, steps, ws, nseg in ((20000, 3000, 4, 16), (20000, 1500, 1, 64), (65536 + 77, 300, 4, 16), (65536 + 77, 150, 1, 64),
                               (6000 + 5, 3000, 16, 4), (16384, 1000, 16, 1)):
        pos, vel = nb.make_bodies(n, seed=7)
        eng = engine_factory(n)
        eng.set_option(nb.OPT_WSPLIT, ws)
        if ws == 16:
            eng.set_option(nb.OPT_JSUB, nseg)       # 16-wave workgroups: 4 segments (hand-off) and 1 (no global partial sums at all)
        out = {}
        for fuse in (1, 0):
            eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
            eng.upload(pos, vel)
            eng.step(1e-4, steps)
            out[fuse] = eng.download()
        assert eng.config["nseg"] == nseg and eng.config["wsplit"] == ws
        assert np.array_equal(bits(out[1][0]), bits(out[0][0])) and np.array_equal(bits(out[1][1]), bits(out[0][1])), n
    # fp64 words (two 16-byte stores per partial)
    n = 5000
    pos, vel = nb.make_bodies(n, seed=2, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    out = {}
    for fuse in (1, 0):
        eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
        eng.set_option(nb.OPT_JSUB, 6)
        eng.upload(pos, vel)
        eng.step(0.01, 9)
        out[fuse] = eng.download()
    assert np.array_equal(bits(out[1][0]), bits(out[0][0])) and np.array_equal(bits(out[1][1]), bits(out[0][1]))


@pytest.mark.parametrize("wsplit", [1, 16, -1])
def test_fpga16_order(nb, oracle_fast, engine_factory, wsplit):
    """16 strided partials + pairwise tree, S/fxyz.vhd:129-184, S/final_adder.vhd:88-104 — with the sixteen
    partial sums in one lane (NBODY_OPT_WSPLIT 1, force_fpga16_f32) and on the sixteen waves of a workgroup (16 = the automatic
    choice, force_fpga16w_f32): the same chains, rotation and tree, hence the same bits.  Sizes: below 16 (results without items),
    ragged tails of every kind, several 64-row workgroups, more than 8 items per chain (the loads-ahead loop) and N = 32767, the
    mailbox's maximum."""
    for n in (1, 5, 15, 16, 17, 100, 129, 257, 1031, 4099, 32767):
        pos, _ = nb.make_bodies(n, seed=2)
        eng = engine_factory(n)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
        eng.set_option(nb.OPT_JSUB, 1)
        eng.set_option(nb.OPT_WSPLIT, wsplit)
        assert eng.config["wsplit"] == (1 if wsplit == 1 else 16) and eng.config["nseg"] == 1
        eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT)
        want = oracle_fast.forces_f32(pos, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(bits(eng.forces(pos)), bits(want)), n
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        want3 = oracle_fast.forces_f32(pos, d2=O.D2_FMA3, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(bits(eng.forces(pos)), bits(want3)), n
        eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
        fast = eng.forces(pos)
        assert (maxnorm_rel(fast, want) < TOL) if n > 1 else not fast.any()      # one body: the self pair alone, force exactly zero
        if n >= 100:
            # several segments: every segment its own sixteen partial sums, rotation and tree; the two kernels agree bit for bit
            eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT)
            eng.set_option(nb.OPT_JSUB, 3)
            got = eng.forces(pos)
            eng.set_option(nb.OPT_WSPLIT, 1)
            assert np.array_equal(bits(got), bits(eng.forces(pos))), n
            # ... and cut as a 2-rank job cuts the sources (two slices x two segments, one launch per slice)
            eng.set_option(nb.OPT_JSLICES, 2)
            eng.set_option(nb.OPT_JSUB, 2)
            one_lane = eng.forces(pos)
            eng.set_option(nb.OPT_WSPLIT, wsplit)
            assert eng.config["nseg"] == 4 and np.array_equal(bits(eng.forces(pos)), bits(one_lane)), n
        if wsplit != 1:
            # round 6: the sixteen-wave form stages its sources through LDS by default (force_fpga16w_lds_f32); NBODY_VARIANT_SMEM keeps
            # round 4's scalar delivery (force_fpga16w_f32): the same chains, the same bits — one segment and several
            for jsl, jsub in ((0, 1), (2, 2)):
                eng.set_option(nb.OPT_JSLICES, jsl)
                eng.set_option(nb.OPT_JSUB, jsub)
                eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT)
                eng.set_option(nb.OPT_VARIANT, nb.VARIANT_AUTO)
                staged = eng.forces(pos)
                eng.set_option(nb.OPT_VARIANT, nb.VARIANT_SMEM)
                assert eng.config["wsplit"] == 16 and np.array_equal(bits(eng.forces(pos)), bits(staged)), (n, jsl, jsub)
                if jsub == 1:
                    assert np.array_equal(bits(staged), bits(want)), n


def test_bodyForce_integrate_config1_shape(nb, oracle_fast, engine_factory):
    """BASELINE config 1's shape (N = 4096, 10 iterations) through the host-pointer entry points, the engine's own
    configuration; and with one sequential sum against the plain CPU loop."""
    n, dt, iters = 4096, 0.01, 10
    for seq in (False, True):
        pos, vel = nb.make_bodies(n)
        opos, ovel = pos.copy(), vel.copy()
        eng = engine_factory(n)
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        if seq:
            eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_SEQ)
            eng.set_option(nb.OPT_JSUB, 1)
        for _ in range(iters):
            eng.bodyForce(pos, vel, dt)
            eng.integrate(pos, vel, dt)
        if seq:
            for _ in range(iters):
                oracle_fast.bodyForce(opos, ovel, dt)
                oracle_fast.integrate(opos, ovel, dt)
        else:
            oracle_step(oracle_fast, eng, opos, ovel, dt, iters)
        assert np.array_equal(bits(pos), bits(opos)), seq
        assert np.array_equal(bits(vel), bits(ovel)), seq


@pytest.mark.parametrize("jsub", [0, 1, 4])
def test_step_loop_strict_bit_exact(nb, oracle_fast, engine_factory, jsub):
    n, dt, steps = 4096, 0.01, 10
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    eng.set_option(nb.OPT_JSUB, jsub)
    eng.upload(pos, vel)
    eng.step(dt, steps)
    gp, gv = eng.download()
    op, ov = pos.copy(), vel.copy()
    oracle_step(oracle_fast, eng, op, ov, dt, steps)
    assert np.array_equal(bits(gp), bits(op))
    assert np.array_equal(bits(gv), bits(ov))
    assert np.all(gp[:, 3] == 1) and np.all(gv[:, 3] == 0)


def test_step_loop_fast_within_tolerance(nb, oracle_fast, engine_factory):
    """The timed mode (v_rsq_f32), the engine's own configuration, N = 4096.

    One step — no amplification yet — is held to 1e-5 on EVERY body, relative to the terms of that body's own update:
    |dv_i| <= 1e-5 max(|v_i|, dt |F_i|) and |dr_i| <= 1e-5 max(|r_i|, dt |v'_i|).  (A component-wise relative error
    |a-b|/|b| is the wrong norm here: a coordinate that happens to lie near zero makes it arbitrarily large for an
    absolute error far below the body's own scale; round 1 worked around that with quantiles.)
    Ten steps: close pairs (eps = 1e-9) amplify last-bit differences — two CPU runs whose 1/sqrt differ by <= 1 ulp
    disagree by far more than 1e-5 on some bodies — so the fixed-step-count bar is bit-exactness in strict mode
    (test_step_loop_strict_bit_exact, test_config2_n65536) and here: median within 1e-5 and the tails no worse than
    4x that CPU-vs-CPU envelope."""
    n, dt, steps = 4096, 0.01, 10
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n)
    eng.upload(pos, vel)
    f0 = eng.forces(pos)
    eng.upload(pos, vel)
    eng.step(dt, 1)
    g1, gv1 = eng.download()
    eng.step(dt, steps - 1)
    gp, gv = eng.download()
    o1, ov1 = pos.copy(), vel.copy()
    oracle_step(oracle_fast, eng, o1, ov1, dt, 1)
    op, ov = pos.copy(), vel.copy()
    oracle_step(oracle_fast, eng, op, ov, dt, steps)
    ep, ev = pos.copy(), vel.copy()
    oracle_step(oracle_fast, eng, ep, ev, dt, steps, rsqrt=O.RSQRT_DIVSQRT)

    def inf(a):
        return np.abs(a[:, :3].astype(np.float64)).max(1)

    dtf = float(np.float32(dt))
    assert np.all(inf(gv1 - ov1.astype(np.float64)) <= TOL * np.maximum(inf(vel), dtf * inf(f0)))
    assert np.all(inf(g1 - o1.astype(np.float64)) <= TOL * np.maximum(inf(pos), dtf * inf(ov1)))
    assert maxnorm_rel(g1, o1) < TOL

    def elementwise(a, b):
        return np.abs(a[:, :3] - b[:, :3]) / np.maximum(np.abs(b[:, :3]), 1e-30)

    el, env = elementwise(gp, op), elementwise(ep, op)
    assert np.median(el) < TOL
    assert np.isfinite(gp).all() and np.isfinite(gv).all()
    assert maxnorm_rel(gp, op) < 4 * max(maxnorm_rel(ep, op), 1e-6)
    assert np.quantile(el, 0.99) < 4 * max(np.quantile(env, 0.99), TOL)


def test_config2_n65536(nb, oracle_fast, engine_factory, capsys):
    """BASELINE config 2 in full: N = 65536, 100 steps, one GPU.  Strict arithmetic in the engine's own configuration
    (64 source segments, blocked sums, one launch per step) and with the LDS tile = 256 delivery the config names: bit-exact
    against the oracle after all 100 steps (4.3e11 pairs on the host cores).  Fast mode (the timed arithmetic, ISA loop):
    forces of the initial state within 1e-5 on every row against the same-order oracle and against fp64; the 100-step
    deviation from the strict run is printed for the record (the dynamics amplify last-bit differences)."""
    n, dt, steps = 65536, 0.01, 100
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n, tile=256)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    cfg = eng.config
    assert cfg["nseg"] > 1 and cfg["sum_order"] == "blocked" and cfg["launches_per_step"] == 1
    order0 = dict(eng.order)                  # the engine's own summation order: what every CPU run below restates
    op, ov = pos.copy(), vel.copy()
    oracle_step(oracle_fast, eng, op, ov, dt, steps)
    eng.upload(pos, vel)
    eng.step(dt, steps)
    gp, gv = eng.download()
    assert np.array_equal(bits(gp), bits(op)) and np.array_equal(bits(gv), bits(ov))
    set_variant(nb, eng, "lds", 2, jsub=cfg["jsub"], arith=nb.ARITH_STRICT)      # LDS tile = 256, same segmentation
    assert eng.config["variant"] == "lds" and eng.config["tile"] == 256
    eng.upload(pos, vel)
    eng.step(dt, steps)
    lp, lv = eng.download()
    assert np.array_equal(bits(lp), bits(op)) and np.array_equal(bits(lv), bits(ov))
    # the timed arithmetic
    set_variant(nb, eng, "auto", 0, jsub=0, arith=nb.ARITH_FMA3)
    assert eng.config["variant"] == "isa" and eng.config["nseg"] == cfg["nseg"] and dict(eng.order) == order0
    f = eng.forces(pos)
    want = oracle_forces(oracle_fast, eng, pos)
    f64 = oracle_fast.forces_f64_from_f32(pos)
    r_same, r_f64 = row_rel(f, want), row_rel(f, f64)
    assert r_same.max() < TOL and r_f64.max() < TOL, (r_same.max(), r_f64.max())
    eng.upload(pos, vel)
    eng.step(dt, steps)
    fp, fv = eng.download()
    assert np.isfinite(fp).all() and np.isfinite(fv).all()
    set_variant(nb, eng, "lds", 2, jsub=cfg["jsub"], arith=nb.ARITH_FMA3)
    eng.upload(pos, vel)
    eng.step(dt, steps)
    assert np.array_equal(bits(eng.download()[0]), bits(fp))                     # LDS tile = 256 == ISA loop, fast mode
    el = np.abs(fp[:, :3] - op[:, :3]) / np.maximum(np.abs(op[:, :3]), 1e-30)
    # The fixed-step-count envelope, asserted where the config lives: a SECOND CPU run in the same
    # summation order whose 1/sqrt is 1.0f/sqrtf instead of the once-rounded fp64 value — it differs from the first by <= 1 ulp
    # per pair, as v_rsq_f32 does.  With eps = 1e-9 and dt = 0.01 close pairs amplify such last-bit differences (a pair closer
    # than (2 dt^2)^(1/3) ~ 0.06 does, nearly every body at this density), so after 100 steps two CPU runs disagree by far more
    # than 1e-5 on many components; the GPU's fast mode must stay inside that CPU-vs-CPU spread: max-norm and 99 % quantile
    # within 4x, and no more than 1.5x as many components beyond 1e-5.
    ep, ev = pos.copy(), vel.copy()
    oracle_fast.step_order(ep, ev, dt, steps, order_=O.order(rsqrt=O.RSQRT_DIVSQRT, **order0))
    env = np.abs(ep[:, :3] - op[:, :3]) / np.maximum(np.abs(op[:, :3]), 1e-30)
    gpu_max, cpu_max = maxnorm_rel(fp, op), maxnorm_rel(ep, op)
    gpu_q99, cpu_q99 = float(np.quantile(el, 0.99)), float(np.quantile(env, 0.99))
    gpu_cnt, cpu_cnt = int((el > TOL).sum()), int((env > TOL).sum())
    with capsys.disabled():
        print("\n[config 2] N=65536: forces fast vs same-order oracle max row rel %.2e, vs fp64 %.2e; after 100 steps against the strict run: "
              "GPU fast mode max-norm %.2e, median component %.2e, 99%% %.2e, components beyond 1e-5: %d  |  CPU run with 1.0f/sqrtf (its "
              "1/sqrt differs by <= 1 ulp): max-norm %.2e, median %.2e, 99%% %.2e, beyond 1e-5: %d"
              % (r_same.max(), r_f64.max(), gpu_max, np.median(el), gpu_q99, gpu_cnt, cpu_max, np.median(env), cpu_q99, cpu_cnt))
    assert np.median(el) < TOL
    assert gpu_max <= 4 * max(cpu_max, 1e-6), (gpu_max, cpu_max)
    assert gpu_q99 <= 4 * max(cpu_q99, TOL), (gpu_q99, cpu_q99)
    assert gpu_cnt <= 1.5 * cpu_cnt + 16, (gpu_cnt, cpu_cnt)


def shard_edge_rows(n, shards, width):
    """(first, count) windows holding the first and last `width` rows of each of `shards` equal shards"""
    out = []
    per = n // shards
    for q in range(shards):
        out.append((q * per, width))
        out.append(((q + 1) * per - width, width))
    return out


_HEADLINE = {}


def headline_reference(nb, oracle_fast, eng):
    """ONE CPU pass over the headline size for the whole session (1.1e12 pairs: under a minute on the GPU box's host): the oracle's forces on
    every row of the seeded N = 1,048,576 state in the order the engine's timed configuration sums in.  Both headline-size tests read it."""
    import time
    if "want" not in _HEADLINE:
        n = 1 << 20
        pos, vel = nb.make_bodies(n)
        t0 = time.time()
        _HEADLINE.update(pos=pos, vel=vel, order=eng.order, want=oracle_forces(oracle_fast, eng, pos))
        _HEADLINE["t_cpu"] = time.time() - t0
    assert _HEADLINE["order"] == eng.order
    return _HEADLINE


def test_headline_size_every_row(nb, oracle_fast, engine_factory, capsys):
    """N = 1,048,576 in the configuration bench.py times, EVERY row (round 4: the host of the GPU box has the cores for one full CPU pass
    — 1.1e12 pairs, under a minute on 256 threads — so the headline size need not be row-sampled): strict arithmetic bit-identical to the
    oracle on all 1,048,576 rows; the timed arithmetic within 1e-5 of the same-order oracle on every row, relative to that row's own force."""
    n = 1 << 20
    eng = engine_factory(n)
    cfg = eng.config
    assert cfg["variant"] == "isa" and cfg["nseg"] == 8 and cfg["sum_block"] == 1024 and cfg["launches_per_step"] == 1 and cfg["wsplit"] == 4
    ref = headline_reference(nb, oracle_fast, eng)
    pos, want = ref["pos"], ref["want"]
    fast = eng.forces(pos)
    r = row_rel(fast, want)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    assert eng.config["nseg"] == 8 and eng.order == ref["order"]
    strict = eng.forces(pos)
    with capsys.disabled():
        print("\n[config 3, every row] N=1048576: strict (%s loop) == oracle on all rows: %s; timed arithmetic worst row %.2e, 99.9 %% %.2e (CPU pass %.0f s)"
              % (eng.config["variant"], np.array_equal(bits(strict), bits(want)), r.max(), np.quantile(r, 0.999), ref["t_cpu"]))
    assert np.array_equal(bits(strict), bits(want))
    assert r.max() < TOL


def test_headline_size_row_windows_and_properties(nb, oracle_fast, engine_factory, capsys):
    """N = 1,048,576 (BASELINE config 3) in the configuration bench.py times, through the ROW-WINDOW entry point (nbody_forces_rows: what a
    sharded job checks itself with) on 1280 rows — the first and last 64 rows of each of 8 shards of 131072 (the 8-GPU shard edges) plus
    256 in the middle, all N sources: strict windows bit-identical to the session's one CPU pass (headline_reference), the timed arithmetic
    within 1e-5 of it and of fp64.  Then size-independent properties of one timed-mode step on the full state."""
    n = 1 << 20
    eng = engine_factory(n)
    cfg = eng.config
    assert cfg["variant"] == "isa" and cfg["nseg"] == 8 and cfg["sum_block"] == 1024 and cfg["launches_per_step"] == 1
    ref = headline_reference(nb, oracle_fast, eng)
    pos, vel, want_all = ref["pos"], ref["vel"], ref["want"]
    eng.upload(pos, vel)
    sample = shard_edge_rows(n, 8, 64) + [(n // 2 - 128, 256)]
    assert sum(c for _, c in sample) >= 1024
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    assert eng.config["nseg"] == 8
    for first, cnt in sample:
        assert np.array_equal(bits(eng.forces_rows(first, cnt)), bits(want_all[first:first + cnt])), first
    eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
    assert eng.config == cfg
    worst_same = worst_f64 = 0.0
    for first, cnt in sample:
        got = eng.forces_rows(first, cnt)
        f64 = oracle_fast.forces_f64_from_f32(pos[first:first + cnt], pos)
        worst_same = max(worst_same, row_rel(got, want_all[first:first + cnt]).max())
        worst_f64 = max(worst_f64, row_rel(got, f64).max())
    with capsys.disabled():
        print("\n[config 3] N=1048576 timed configuration %s: worst sampled row vs same-order oracle %.2e, vs fp64 %.2e" % (cfg, worst_same, worst_f64))
    assert worst_same < TOL and worst_f64 < TOL
    # properties: a step moves exactly r' = r + v'*dt (checked from the downloaded state), w carried
    eng.step(0.01, 1)
    p1, v1 = eng.download()
    dt = np.float64(np.float32(0.01))
    assert np.array_equal(bits(p1[:, :3]), bits((v1[:, :3].astype(np.float64) * dt + pos[:, :3]).astype(np.float32)))
    assert np.all(p1[:, 3] == 1) and np.all(v1[:, 3] == 0) and np.isfinite(v1).all()
    # the kick used the same forces the row sample saw: v' = fma(dt, F, v) exactly, on the sampled rows
    first, cnt = sample[-1]
    eng.upload(pos, vel)
    f = eng.forces_rows(first, cnt)
    assert np.array_equal(bits(v1[first:first + cnt, :3]),
                          bits((dt * f[:, :3].astype(np.float64) + vel[first:first + cnt, :3]).astype(np.float32)))
    # momentum: unit masses and antisymmetric pair terms -> sum_i F_i ~ 0 relative to sum_i |F_i|
    dv = (v1[:, :3].astype(np.float64) - vel[:, :3]) / dt
    assert np.abs(dv.sum(0)).max() / np.abs(dv).sum() < 1e-4


def test_config4_workload_eight_virtual_ranks(nb, oracle_fast, engine_factory, monkeypatch):
    """BASELINE config 4's workload on this box's one GPU: N = 1,048,576 sharded over 8 (virtual) ranks of 131072 bodies,
    each rank's own-slice launch first, then the arrived slices — the schedule of the 8-GPU job with peer copies in place
    of xGMI.  Two steps must equal one GPU configured with the same segmentation (8 slices x the same sub) bit for bit, and
    the forces on rows at the shard edges must be within 1e-5 of the oracle in that order and of fp64."""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n, P, dt, steps = 1 << 20, 8, 0.01, 2
    pos, vel = nb.make_bodies(n)
    multi = engine_factory(n, ngpus=P)
    cfg = multi.config
    assert cfg["nranks"] == P and cfg["n_local"] == n // P and cfg["launches_per_step"] == 2
    multi.upload(pos, vel)
    multi.step(dt, steps)
    mp, mv = multi.download()
    one = engine_factory(n)
    one.set_option(nb.OPT_JSLICES, P)
    one.set_option(nb.OPT_JSUB, cfg["jsub"])
    one.set_option(nb.OPT_WSPLIT, cfg["wsplit"])
    assert one.config["nseg"] == cfg["nseg"]
    one.upload(pos, vel)
    one.step(dt, steps)
    wp, wv = one.download()
    assert np.array_equal(bits(mp), bits(wp)) and np.array_equal(bits(mv), bits(wv))
    # the row sample on the SHARDED engine itself (global body indices; windows that straddle two ranks' slices included)
    multi = engine_factory(n, ngpus=P)
    multi.upload(pos, vel)
    worst = 0.0
    per = n // P
    for first, cnt in shard_edge_rows(n, P, 32) + [(q * per - 16, 32) for q in range(1, P)]:
        got = multi.forces_rows(first, cnt)
        want = oracle_forces(oracle_fast, multi, pos[first:first + cnt], pos)
        f64 = oracle_fast.forces_f64_from_f32(pos[first:first + cnt], pos)
        worst = max(worst, row_rel(got, want).max(), row_rel(got, f64).max())
    assert worst < TOL, worst
    multi.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    for first, cnt in ((per - 16, 32), (5 * per - 3, 40)):
        assert np.array_equal(bits(multi.forces_rows(first, cnt)), bits(oracle_forces(oracle_fast, multi, pos[first:first + cnt], pos))), first


def test_config5_workload_fp64(nb, oracle_fast, engine_factory):
    """BASELINE config 5's workload on one GPU: N = 4,194,304 fp64, one step (1.8e13 pairs, ~11 s).  Row-sampled forces
    (nbody_forces_rows_d) against the fp64 oracle at the 8-GPU shard edges, then the size-independent properties of the step."""
    n = 1 << 22
    pos, vel = nb.make_bodies(n, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    cfg = eng.config
    assert cfg["variant"] == "isa" and cfg["sum_order"] == "seq" and cfg["launches_per_step"] == 1
    eng.upload(pos, vel)
    for first, cnt in shard_edge_rows(n, 8, 16) + [(n // 2 - 64, 128)]:
        got = eng.forces_rows(first, cnt)
        want = oracle_fast.forces_f64(pos[first:first + cnt], pos)
        assert row_rel(got, want).max() < 1e-11, first        # different summation orders in fp64: ~sqrt(N) * 2^-53
        assert np.all(got[:, 3] == 0)
    # the same windows in strict arithmetic (round 4): bit-identical to the oracle in the engine's order — the segmentation of config 5's
    # size (64 L2-resident segments x 4 pieces), pinned to the last bit at full size
    order0 = eng.order
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    assert eng.order == order0 and eng.config["nseg"] == cfg["nseg"] and eng.config["variant"] == "smem"
    order = O.order(nslices=order0["nslices"], sub=order0["sub"], wsplit=order0["wsplit"])
    for first, cnt in shard_edge_rows(n, 8, 16)[:4] + [(n // 2 - 64, 128)]:
        got = eng.forces_rows(first, cnt)
        assert np.array_equal(bits(got), bits(oracle_fast.forces_f64_order(pos[first:first + cnt], pos, order_=order))), first
    eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
    eng.step(0.01, 1)
    p1, v1 = eng.download()
    assert np.isfinite(p1).all() and np.isfinite(v1).all() and np.all(p1[:, 3] == 1) and np.all(v1[:, 3] == 0)
    # r' = fma(v', dt, r) exactly: check with exact rational arithmetic on a sample (fp64 has no wider host type)
    from fractions import Fraction
    for i in (0, 1, n // 3, n - 1):
        for c in range(3):
            exact = Fraction(float(v1[i, c])) * Fraction(0.01) + Fraction(float(pos[i, c]))
            assert float(exact) == p1[i, c], (i, c)
    dv = (v1[:, :3] - vel[:, :3]) / 0.01
    assert np.abs(dv.sum(0)).max() / np.abs(dv).sum() < 1e-9


def test_translation_property(nb, engine_factory):
    """Forces depend on differences only: shifting every body by a power-of-two-exact offset that
    keeps all coordinates exactly representable leaves them bit-identical."""
    n = 2048
    pos, _ = nb.make_bodies(n, seed=9)
    pos[:, :3] = np.round(pos[:, :3] * 1024) / 1024      # 11 fractional bits
    eng = engine_factory(n)
    f0 = eng.forces(pos)
    shifted = pos.copy()
    shifted[:, :3] += np.float32(8.0)
    assert np.array_equal(bits(eng.forces(shifted)), bits(f0))


def test_coincident_bodies_stay_finite(nb, engine_factory):
    """S/dzsoft.vhd:177, 201-202: eps keeps d2 > 0 for coincident bodies; their mutual term is exactly 0."""
    n = 512
    pos, _ = nb.make_bodies(n, seed=4)
    pos[1] = pos[0]
    pos[100:110] = pos[99]
    eng = engine_factory(n)
    f = eng.forces(pos)
    assert np.isfinite(f).all()
    assert np.array_equal(bits(f[0]), bits(f[1]))


def test_extreme_values_strict_bit_exact(nb, oracle_fast, engine_factory):
    """Edge cases of the arithmetic, strict mode bit for bit against the oracle (NaN compared as NaN):
    separations that underflow (d2 == eps), coordinates whose squares overflow (d2 = inf, inv = 0), huge and tiny
    magnitudes mixed, exact duplicates, signed zeros."""
    n = 600
    pos, _ = nb.make_bodies(n, seed=77)
    pos[10, :3] = pos[11, :3] + np.float32(1e-30)          # dx*dx underflows: d2 == eps exactly
    pos[20, :3] = [3e19, -2e19, 1e19]                      # squares overflow to inf against everything else
    pos[21, :3] = [1e-38, -1e-38, 0.0]                     # near the subnormal range
    pos[22, :3] = [-0.0, 0.0, -0.0]
    pos[23, :3] = [0.0, -0.0, 0.0]
    pos[30:34, :3] = pos[29, :3]                           # five coincident bodies
    pos[40, :3] = [1e10, 1e10, 1e10]
    eng = engine_factory(n)
    eng.set_option(nb.OPT_SUM_BLOCK, 64)                   # several blocks and several segments at this size
    for arith, d2 in ((nb.ARITH_STRICT, O.D2_FMA3), (nb.ARITH_REFERENCE_STRICT, O.D2_REFERENCE)):
        eng.set_option(nb.OPT_ARITH, arith)
        with np.errstate(all="ignore"):
            want = oracle_forces(oracle_fast, eng, pos, d2=d2)
        got = eng.forces(pos)
        nan_w, nan_g = np.isnan(want), np.isnan(got)
        assert np.array_equal(nan_w, nan_g)
        assert np.array_equal(bits(got)[~nan_g], bits(want)[~nan_w])
    # the timed mode stays finite wherever the oracle is finite
    eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
    got = eng.forces(pos)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))


def test_nan_input_propagates(nb, engine_factory):
    """A NaN coordinate poisons every force (all pairs see it), as in the RTL where NaN flows through the IP cores
    (T/tb_sqrt.vhd:536-537: recip_sqrt(NaN) = NaN)."""
    n = 300
    pos, _ = nb.make_bodies(n, seed=5)
    pos[7, 1] = np.nan
    eng = engine_factory(n)
    f = eng.forces(pos)
    assert np.isnan(f[:, :3]).all() and np.all(f[:, 3] == 0)


def test_mailbox_maximum_points(nb, oracle_fast, engine_factory):
    """NUM_PTS is a 15-bit field and the RAM holds 32768 words (S/top_level.vhd:45, 185): N = 32767 is the largest
    request the reference accepts; same image in, same layout out."""
    n = nb.mailbox.MAX_POINTS
    pos, _ = nb.make_bodies(n, seed=12)
    ram_a = nb.mailbox.encode_request(pos)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    ram_b = nb.mailbox.run(eng, ram_a)
    assert ram_b.shape == (n, 4) and nb.mailbox.decode_control(ram_a)["begin"] == 0
    rows = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
    want = oracle_forces(oracle_fast, eng, pos[rows], pos)
    assert np.array_equal(bits(ram_b[rows]), bits(want))


def test_mailbox_front_end(nb, oracle_fast, engine_factory):
    """The reference's RAM images verbatim."""
    n = 1000
    pos, _ = nb.make_bodies(n, seed=8)
    pos[:, 3] = np.float32(123.0)                         # bits 127:96 are ignored, S/top_level.vhd:206-208
    ram_a = nb.mailbox.encode_request(pos)
    assert nb.mailbox.decode_control(ram_a) == dict(begin=1, num_pts=n, ticks=n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    ram_b = nb.mailbox.run(eng, ram_a, clock_khz=300000)
    ctl = nb.mailbox.decode_control(ram_a)
    assert ctl["begin"] == 0 and ctl["ticks"] >= 1       # S/top_level.vhd:146, 255-263
    assert np.array_equal(bits(ram_b), bits(oracle_forces(oracle_fast, eng, pos)))
    assert np.all(ram_b[:, 3] == 0)
    with pytest.raises(nb.NBodyError):                    # BEGIN not set -> nothing to do
        nb.mailbox.run(eng, ram_a)


def test_fp64_strict_bit_exact(nb, oracle_fast, engine_factory):
    """Round 4: NBODY_ARITH_STRICT in an fp64 context — 1/sqrt as IEEE sqrt and divide — is bit-identical to the oracle's fp64
    evaluation in the engine's summation order (segments x pieces of the wave split, one sequential sum per piece): forces, row
    windows, the device loop and the host-pointer loop, ragged sizes, 1/2/4 bodies per lane, 1/4/16 waves, both combine forms.
    fp64 parity was tolerance-only before: a summation-order slip of last-bit size is now visible."""
    for n in (1, 2, 63, 65, 257, 1000, 4099):
        pos, vel = nb.make_bodies(n, seed=n, dtype=np.float64)
        pos[:, :3] += nb.make_bodies(n, seed=n + 1, dtype=np.float64)[0][:, :3] * 2.0 ** -25       # full-width significands
        eng = engine_factory(n, fp64=True)
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        for iblock, wsplit, jsub, jsl, fuse in ((0, -1, 0, 1, -1), (1, 1, 1, 1, 1), (1, 4, 3, 1, 1), (1, 16, 2, 1, 0), (2, 1, 5, 1, 1), (4, 1, 2, 3, 0), (1, 4, 2, 3, 1)):
            if n < jsl:
                continue
            eng.set_option(nb.OPT_IBLOCK, iblock)
            eng.set_option(nb.OPT_WSPLIT, wsplit)
            eng.set_option(nb.OPT_JSUB, jsub)
            eng.set_option(nb.OPT_JSLICES, jsl)
            eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
            cfg = eng.config
            assert cfg["variant"] == "smem", cfg                  # the strict arithmetic lives in the compiled kernel
            o = eng.order
            order = O.order(nslices=o["nslices"], sub=o["sub"], wsplit=o["wsplit"])
            what = (n, iblock, wsplit, jsub, jsl, fuse, cfg)
            f = eng.forces(pos)
            assert np.array_equal(bits(f), bits(oracle_fast.forces_f64_order(pos, order_=order))), what
            r0, cnt = n // 3, max(1, min(100, n - n // 3))
            eng.upload(pos, vel)
            assert np.array_equal(bits(eng.forces_rows(r0, cnt)), bits(f[r0:r0 + cnt])), what
            eng.step(0.01, 4)
            gp, gv = eng.download()
            op, ov = pos.copy(), vel.copy()
            oracle_fast.step_f64_order(op, ov, 0.01, 4, order_=order)
            assert np.array_equal(bits(gp), bits(op)) and np.array_equal(bits(gv), bits(ov)), what
        # host-pointer bodyForce_d()/integrate_d()
        p2, v2 = pos.copy(), vel.copy()
        eng.bodyForce(p2, v2, 0.01)
        eng.integrate(p2, v2, 0.01)
        o = eng.order
        op, ov = pos.copy(), vel.copy()
        oracle_fast.step_f64_order(op, ov, 0.01, 1, order_=O.order(nslices=o["nslices"], sub=o["sub"], wsplit=o["wsplit"]))
        assert np.array_equal(bits(p2), bits(op)) and np.array_equal(bits(v2), bits(ov))
        # the timed fp64 arithmetic (the inverse cube from the v_rsq_f64 seed by one third-order step) differs from it by a few ulp per pair only
        eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
        fast = eng.forces(pos)
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        strict = eng.forces(pos)
        assert np.array_equal(bits(fast), bits(strict)) or maxnorm_rel(fast, strict) < 1e-14       # (n = 1: the self pair alone, exactly zero)


def test_fp64_strict_every_row_at_the_profiled_size(nb, oracle_fast, engine_factory):
    """fp64 N = 262144 — the size the fp64 rocprofv3 profiles and bench lines are taken at — in the engine's own configuration (8 segments x 4
    pieces): strict arithmetic bit-identical to the oracle on every row (6.9e10 pairs on the host), and the timed arithmetic (the 16-instruction
    loop) within 1e-13 of it on every row, relative to that row's own force"""
    n = 262144
    pos, _ = nb.make_bodies(n, dtype=np.float64)
    pos[:, :3] += nb.make_bodies(n, seed=7, dtype=np.float64)[0][:, :3] * 2.0 ** -25
    eng = engine_factory(n, fp64=True)
    assert eng.config["variant"] == "isa"
    fast = eng.forces(pos)
    o = eng.order
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    assert eng.order == o and eng.config["variant"] == "smem"
    want = oracle_fast.forces_f64_order(pos, order_=O.order(nslices=o["nslices"], sub=o["sub"], wsplit=o["wsplit"]))
    assert np.array_equal(bits(eng.forces(pos)), bits(want))
    assert row_rel(fast, want).max() < 1e-13


def test_fp64_strict_virtual_ranks_bit_exact(nb, oracle_fast, engine_factory, monkeypatch):
    """config 5's decomposition in small: an fp64 job of 3 and of 8 virtual ranks on this GPU (ragged slices), strict arithmetic, the three
    overlap modes and both combine forms — forces and four steps bit-identical to the oracle in the job's order (one slice per rank x
    pieces per slice x waves), i.e. the multi-GPU fp64 schedule pinned to the last bit"""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    for ranks, n in ((3, 5000 + 1), (8, 9000 + 5)):
        pos, vel = nb.make_bodies(n, seed=n, dtype=np.float64)
        pos[:, :3] += nb.make_bodies(n, seed=n + 1, dtype=np.float64)[0][:, :3] * 2.0 ** -25
        for overlap, fuse in ((1, 1), (2, 1), (0, 0), (1, 0)):
            eng = engine_factory(n, fp64=True, ngpus=ranks)
            eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
            eng.set_option(nb.OPT_OVERLAP, overlap)
            eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
            o = eng.order
            assert o["nslices"] == ranks
            order = O.order(nslices=o["nslices"], sub=o["sub"], wsplit=o["wsplit"])
            assert np.array_equal(bits(eng.forces(pos)), bits(oracle_fast.forces_f64_order(pos, order_=order))), (ranks, overlap, fuse)
            eng.upload(pos, vel)
            eng.step(0.01, 4)
            gp, gv = eng.download()
            op, ov = pos.copy(), vel.copy()
            oracle_fast.step_f64_order(op, ov, 0.01, 4, order_=order)
            assert np.array_equal(bits(gp), bits(op)) and np.array_equal(bits(gv), bits(ov)), (ranks, overlap, fuse)


def test_fp64_path(nb, oracle_fast, engine_factory):
    """BASELINE config 5's arithmetic at a size the host can check: fp64 forces and 5 steps."""
    n = 4096
    pos, vel = nb.make_bodies(n, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    f = eng.forces(pos)
    want = oracle_fast.forces_f64(pos)
    assert maxnorm_rel(f, want) < 1e-13
    assert np.array_equal(bits(eng.forces_rows(100, 300)), bits(f[100:400]))
    for iblock, jsub in ((1, 1), (2, 4), (4, 2)):
        eng.set_option(nb.OPT_IBLOCK, iblock)
        eng.set_option(nb.OPT_JSUB, jsub)
        eng.upload(pos, vel)
        eng.step(0.01, 5)
        gp, gv = eng.download()
        op, ov = pos.copy(), vel.copy()
        oracle_fast.step(op, ov, 0.01, 5)
        assert np.abs(gp[:, :3] - op[:, :3]).max() / np.abs(op[:, :3]).max() < 1e-10
    p2, v2 = pos.copy(), vel.copy()
    eng.bodyForce(p2, v2, 0.01)
    eng.integrate(p2, v2, 0.01)
    o2, ov2 = pos.copy(), vel.copy()
    oracle_fast.bodyForce(o2, ov2, 0.01)
    oracle_fast.integrate(o2, ov2, 0.01)
    assert np.abs(p2 - o2).max() < 1e-12 * np.abs(o2).max()


# (the pair-level accuracy of the fp64 inverse cube — 150 separations against the oracle until round 4, 19 s of OpenMP wake-ups for two-body
#  calls — is now tests/test_golden_fp64.py::test_inverse_cube_of_the_timed_fp64_arithmetic_over_the_whole_range_of_d2: 3000 pairs, 0.2 s)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 9, 250, 1031])
def test_fp64_isa_loop_equals_compiled_kernel(nb, oracle_fast, engine_factory, n):
    """fp64 default (hand-scheduled loop, 4 sources per iteration + scalar tail) vs the hipcc-scheduled kernel: same bits;
    and both within fp64 tolerance of the oracle."""
    pos, vel = nb.make_bodies(n, seed=200 + n, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    eng.set_option(nb.OPT_JSUB, 1)
    assert eng.config["variant"] == "isa"
    a = eng.forces(pos)
    for phase in (0, 2):        # the same loop one placement phase off; the constants from VGPR pairs
        eng.set_option(nb.OPT_ISA_PHASE, phase)
        assert np.array_equal(bits(a), bits(eng.forces(pos))), phase
    eng.set_option(nb.OPT_ISA_PHASE, 1)
    eng.upload(pos, vel)
    eng.step(0.01, 3)
    pa, va = eng.download()
    eng.set_option(nb.OPT_VARIANT, nb.VARIANT_SMEM)
    eng.set_option(nb.OPT_IBLOCK, 1)
    assert eng.config["variant"] == "smem"
    assert np.array_equal(bits(a), bits(eng.forces(pos)))
    eng.upload(pos, vel)
    eng.step(0.01, 3)
    pb, vb = eng.download()
    assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(va), bits(vb))
    want = oracle_fast.forces_f64(pos)
    scale = max(np.abs(want[:, :3]).max(), 1e-300)
    assert np.abs(a[:, :3] - want[:, :3]).max() / scale < 1e-13 or n == 1


def test_virtual_multi_gpu_schedule_bitwise(nb, oracle_fast, engine_factory, monkeypatch):
    """The multi-GPU schedule (i-sharding, ring-ordered arrival, per-slice partials, ascending combine by the last
    arriver across the step's launches, double-buffered positions) with P virtual ranks sharing this box's one GPU;
    the transfers are peer copies, everything else is the code the 8-GPU run executes.  Must equal one GPU configured
    with the same segmentation bit for bit, in fast mode, over several steps — for every overlap mode (gather first /
    own slice then the rest / one launch per arriving slice) and for the two-launch combine."""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n, dt, steps = 8192 + 5, 0.01, 4
    pos, vel = nb.make_bodies(n, seed=21)
    for P in (1, 2, 3, 4):
        one = engine_factory(n)
        one.set_option(nb.OPT_IBLOCK, 2)
        one.set_option(nb.OPT_JSUB, 2)
        one.set_option(nb.OPT_JSLICES, P)
        one.set_option(nb.OPT_WSPLIT, 4)      # (the engine's own choice depends on a rank's body count: pinned on both sides)
        one.upload(pos, vel)
        one.step(dt, steps)
        wp, wv = one.download()
        if P == 1:
            # and against the oracle in that order (strict)
            one.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
            one.upload(pos, vel)
            one.step(dt, steps)
            sp, sv = one.download()
            op, ov = pos.copy(), vel.copy()
            oracle_step(oracle_fast, one, op, ov, dt, steps)
            assert np.array_equal(bits(sp), bits(op)) and np.array_equal(bits(sv), bits(ov))
        for overlap, fuse in (((1, 1),) if P == 1 else ((1, 1), (0, 1), (2, 1), (1, 0), (2, 0))):
            eng = engine_factory(n, ngpus=P)
            eng.set_option(nb.OPT_IBLOCK, 2)
            eng.set_option(nb.OPT_JSUB, 2)
            eng.set_option(nb.OPT_WSPLIT, 4)
            eng.set_option(nb.OPT_OVERLAP, overlap)
            eng.set_option(nb.OPT_FUSE_COMBINE, fuse)
            eng.upload(pos, vel)
            eng.step(dt, steps)
            gp, gv = eng.download()
            assert np.array_equal(bits(gp), bits(wp)), (P, overlap, fuse)
            assert np.array_equal(bits(gv), bits(wv)), (P, overlap, fuse)


@pytest.mark.parametrize("fp64", [False, True])
def test_virtual_eight_ranks_all_entry_points(nb, engine_factory, monkeypatch, fp64):
    """P = 8 virtual ranks (the driver's 8-GPU shape) on a ragged N: device loop, host-pointer bodyForce()/integrate()
    and the force-only entry point all equal the one-GPU run with the same segmentation, bit for bit."""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n, P, dt = 4099, 8, 0.01
    dtype = np.float64 if fp64 else np.float32
    pos, vel = nb.make_bodies(n, seed=61, dtype=dtype)

    def run(eng):
        eng.set_option(nb.OPT_JSUB, 2)
        eng.set_option(nb.OPT_WSPLIT, 16 if not fp64 else 4)
        out = {}
        out["forces"] = eng.forces(pos)
        p, v = pos.copy(), vel.copy()
        for _ in range(2):
            eng.bodyForce(p, v, dt)
            eng.integrate(p, v, dt)
        out["host_loop"] = (p, v)
        eng.upload(pos, vel)
        eng.step(dt, 3)
        out["device_loop"] = eng.download()
        return out

    multi = run(engine_factory(n, ngpus=P, fp64=fp64))
    one_eng = engine_factory(n, fp64=fp64)
    one_eng.set_option(nb.OPT_JSLICES, P)
    one = run(one_eng)
    assert np.array_equal(bits(multi["forces"]), bits(one["forces"]))
    for key in ("host_loop", "device_loop"):
        for a, b in zip(multi[key], one[key]):
            assert np.array_equal(bits(a), bits(b)), key


def test_step_graph_replay_equals_eager(nb, engine_factory):
    """nbody_step on one GPU replays a HIP graph of two steps (launch-bound regime); results must not depend on it,
    for even and odd step counts, after an upload in between, and after an option change."""
    n = 3000
    pos, vel = nb.make_bodies(n, seed=8)
    eng = engine_factory(n)
    out = {}
    for graph in (1, 0):
        eng.set_option(nb.OPT_GRAPH, graph)
        eng.upload(pos, vel)
        eng.step(0.01, 7)
        a = eng.download()
        eng.step(0.01, 4)
        eng.set_option(nb.OPT_JSUB, 3)
        eng.step(0.005, 6)
        b = eng.download()
        eng.set_option(nb.OPT_JSUB, 0)
        out[graph] = (a, b)
    for k in (0, 1):
        for x, y in zip(out[1][k], out[0][k]):
            assert np.array_equal(bits(x), bits(y))


@pytest.mark.parametrize("n", [3000, 20000])
def test_long_step_graphs_equal_eager_launches(nb, engine_factory, n):
    """the 8-, 16- and 32-step captures (NBODY_OPT_GRAPH = 1 picks them from the call's step count: 8 from 16, 16 from
    32, 32 from 64) and explicit steps-per-graph values, each against eager launches of the same steps — both launch regimes
    (n = 3000: 16-wave workgroups; n = 20000: in-launch combine), with a leftover tail after the graphs and a second call that
    re-uses or re-captures the graph"""
    pos, vel = nb.make_bodies(n, seed=9)
    eng = engine_factory(n)

    def run(graph, calls):
        eng.set_option(nb.OPT_GRAPH, graph)
        eng.upload(pos, vel)
        for k in calls:
            eng.step(0.01, k)
        return eng.download()

    for calls in ((20,), (40,), (70,), (33, 17), (65, 3, 64)):
        want = run(0, calls)
        for graph in (1, 8, 16, 32, 6):
            got = run(graph, calls)
            assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(bits(got[1]), bits(want[1])), (calls, graph)


def test_rccl_calls_on_a_one_rank_communicator(nb, engine_factory):
    """The RCCL data path of the multi-GPU job — dlopen/dlsym of librccl, ncclCommInitRank, ncclAllGather in place,
    grouped ncclSend/ncclRecv ring step: argument order and byte counts — executed on this box's one GPU through a
    one-rank communicator (nbody_init_rank with nranks = 1 and a unique id; a second process on the same device is
    refused by RCCL, so this is as far as one GPU goes).  nbody_comm_selftest checks every received word."""
    n = 10000 + 3
    uid = nb.unique_id()
    assert len(uid) == 128 and any(uid)
    eng = engine_factory(n, rank=0, nranks=1, uid=uid)
    assert eng.info(nb._lib.INFO_HAS_COMM) == 1
    for comm in (nb.COMM_RING, nb.COMM_ALLGATHER, nb.COMM_AUTO, nb.COMM_DIRECT):
        eng.set_option(nb.OPT_COMM, comm)
        moved = eng.comm_selftest()
        assert moved == (n // 2) * 16            # one rank: the all-gather moves nothing, the ring step half the array
    # the P > 1 transfer plans (offsets, byte counts, send/receive pairing; ragged slices: n = 10003) of 2, 3, 5 and 8 virtual
    # ranks through real ncclSend/ncclRecv: afterwards every virtual rank's array holds all N words
    for vp in (2, 3, 5, 8):
        for form in (nb.COMM_RING, nb.COMM_DIRECT):
            assert eng.comm_selftest_virtual(vp, form) == (vp - 1) * n * 16, (vp, form)
    with pytest.raises(nb.NBodyError):
        eng.comm_selftest_virtual(8, nb.COMM_ALLGATHER)      # a collective has no plan
    # one ring step timed beside a force pass (tools/comm_probe.py makes the table of profiles/r03_comm_under_load.md)
    for when in (0, 1, 2):
        c_ms, f_ms = eng.comm_probe(64 << 10, when)
        assert c_ms > 0 and (f_ms > 0) == (when != 0)
    # the context computes as usual
    pos, vel = nb.make_bodies(n, seed=1)
    eng.upload(pos, vel)
    eng.step(0.01, 2)
    gp, gv = eng.download()
    plain = engine_factory(n)
    plain.upload(pos, vel)
    plain.step(0.01, 2)
    wp, wv = plain.download()
    assert np.array_equal(bits(gp), bits(wp)) and np.array_equal(bits(gv), bits(wv))
    with pytest.raises(nb.NBodyError):           # no communicator on a plain context
        plain.comm_selftest()
    f64 = engine_factory(2001, fp64=True, rank=0, nranks=1, uid=nb.unique_id())
    assert f64.comm_selftest() == (2001 // 2) * 32


def test_rate_floors_of_the_timed_kernels(nb, engine_factory, capsys):
    """A guard against silent performance regressions (a loop assembled one placement phase off loses 28 %, an extra VGPR an eighth
    of the resident waves): HIP-event time of the force kernel in the engine's own configuration.  Floors sit 8-10 % under the slowest
    box met in four rounds (fp32 N = 1M: 4476 G pairs/s; fp32 N = 65536: 4480 untimed, ~4270 with an event pair around every 0.94-ms launch; fp64 N = 262144 with the 16-instruction loop: 1926)."""
    seen = {}
    for name, n, fp64, steps, floor in (("fp32 N=1048576", 1 << 20, False, 2, 4100.0), ("fp32 N=65536", 65536, False, 50, 3800.0),
                                        ("fp64 N=262144", 262144, True, 3, 1750.0)):
        pos, vel = nb.make_bodies(n, dtype=np.float64 if fp64 else np.float32)
        eng = engine_factory(n, fp64=fp64)
        eng.upload(pos, vel)
        eng.set_option(nb.OPT_TIMING, 1)
        eng.step(0.01, 1)
        eng.sync()
        eng.kernel_time(reset=True)
        eng.step(0.01, steps)
        eng.sync()
        ms, launches = eng.kernel_time(reset=True)
        rate = float(n) * n * steps / (ms * 1e-3) / 1e9
        seen[name] = rate
        assert launches == steps and rate >= floor, (name, rate, floor)
    with capsys.disabled():
        print("\n[rate floors] " + "; ".join("%s: %.0f G pairs/s" % kv for kv in seen.items()))
