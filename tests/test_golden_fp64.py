"""The fp64 path against a checked fixture (tests/golden/fp64_n64.json, made by tests/golden/make_fp64.py: the force sum
evaluated in 60-digit decimal arithmetic and rounded once to binary64 — no code shared with the oracle or the kernels).

The timed fp64 arithmetic takes the inverse cube from the v_rsq_f64 seed by one third-order step (a few ulp; the strict mode — IEEE sqrt
and divide, bit-identical to the oracle — is tested at the end of this file and in test_gpu_parity.py), so its bar is a bound, written here:
every row's force within 8 ulp of that row's largest component (measured on MI355X: 4.0 worst with one wave per segment, 3.0 with 4 or 16).  What a correctly rounded evaluation in sequential order
costs is measured beside it (the oracle: 7 ulp worst row, 2 median at N = 64) — the bound is about that, far below what a
missing, doubled or misplaced source would do (a single term is ~2^52 ulp), in every segmentation the engine can take."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BOUND_ULP = 8.0


def load():
    fx = json.load(open(os.path.join(HERE, "golden", "fp64_n64.json")))
    w = lambda k: np.array([int(h, 16) for h in fx[k]], np.uint64).view(np.float64).reshape(-1, 4)   # noqa: E731
    return fx, w("pos0"), w("forces0")


def row_ulps(got, want):
    ulp = np.spacing(np.abs(want[:, :3]).max(1))
    return np.abs(got[:, :3] - want[:, :3]).max(1) / ulp


def test_fixture_inputs_are_full_width_doubles(nb):
    fx, pos, f = load()
    p1, _ = nb.make_bodies(fx["n"], seed=fx["seed"], dtype=np.float64)
    p2, _ = nb.make_bodies(fx["n"], seed=fx["seed"] + 1, dtype=np.float64)
    want = p1 + p2 * 2.0 ** -25
    assert np.array_equal(pos[:, :3], want[:, :3]) and np.all(pos[:, 3] == 1.0)
    assert (pos[:, :3].view(np.uint64) & np.uint64(0x1FFFFFFF)).any()         # the low 29 bits of the significand are in use
    assert np.isfinite(f).all() and np.all(f[:, 3] == 0)


def test_oracle_fp64_within_bound(oracle, oracle_fast):
    fx, pos, f = load()
    for ora in (oracle, oracle_fast):
        e = row_ulps(ora.forces_f64(pos), f)
        assert e.max() <= 8.0, e.max()          # sequential order, IEEE sqrt and divide: measured 7.0 worst, 2.0 median


@pytest.mark.gpu
def test_engine_fp64_within_bound_in_every_segmentation(nb, capsys):
    """the engine's own configuration, the three wave splits and two explicit segmentations, both delivery kernels"""
    fx, pos, f = load()
    eng = nb.NBody(fx["n"], fp64=True)
    seen = []
    try:
        for variant in (nb.VARIANT_AUTO, nb.VARIANT_SMEM):
            for wsplit in (-1, 1, 4, 16):
                for jsub in (0, 1, 3):
                    eng.set_option(nb.OPT_VARIANT, variant)
                    eng.set_option(nb.OPT_WSPLIT, wsplit)
                    eng.set_option(nb.OPT_JSUB, jsub)
                    cfg = eng.config
                    e = row_ulps(eng.forces(pos), f)
                    seen.append((cfg["variant"], cfg["wsplit"], cfg["nseg"], float(e.max()), float(np.median(e))))
                    assert e.max() <= BOUND_ULP, seen[-1]
        assert {s[1] for s in seen} == {1, 4, 16}
    finally:
        eng.close()
    with capsys.disabled():
        print("\n[fp64 fixture] worst row / median in ulp of the row's largest component: " +
              "; ".join("%s W=%d nseg=%d: %.1f / %.1f" % s for s in seen[:12]))


@pytest.mark.gpu
def test_engine_fp64_strict_equals_the_oracle_and_the_fixture_bounds_both(nb, oracle_fast):
    """NBODY_ARITH_STRICT in fp64 (IEEE sqrt and divide): bit-identical to the oracle in the engine's order — and both, being the same
    numbers, sit inside the fixture's bound"""
    import oracle as O
    fx, pos, f = load()
    eng = nb.NBody(fx["n"], fp64=True)
    try:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        for wsplit, jsub in ((1, 1), (4, 1), (16, 1), (4, 3), (1, 5)):
            eng.set_option(nb.OPT_WSPLIT, wsplit)
            eng.set_option(nb.OPT_JSUB, jsub)
            o = eng.order
            got = eng.forces(pos)
            want = oracle_fast.forces_f64_order(pos, order_=O.order(nslices=o["nslices"], sub=o["sub"], wsplit=o["wsplit"]))
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (wsplit, jsub)
            assert row_ulps(got, f).max() <= BOUND_ULP
    finally:
        eng.close()


@pytest.mark.gpu
def test_inverse_cube_of_the_timed_fp64_arithmetic_over_the_whole_range_of_d2(nb, capsys):
    """inv3_f64 (csrc/nbody_kernels.hpp: v_rsq_f64 seed + ONE third-order step on the cube, six operations) was pinned only
    by the N = 64 fixture's 8-ulp bound.  Here per PAIR, over d2 from the softening (1e-9) to 1e12 with random significands: a two-body
    system at separation d along a random direction gives F_0 = d * inv3(d2) (the self term adds 0), so the timed arithmetic and the
    strict one (IEEE square root and divide, then inv * (inv * inv): the oracle's expression) must agree to a few ulps of the largest
    component.  The strict chain's own roundings are the larger part of the allowance: inv carries two (<= 2 x 2^-53), its cube three
    times that plus two more, the product with d another — up to 9 x 2^-53 relative against < 3 x 2^-53 for the timed form, i.e. up to
    12 half-ulps = 6-12 ulps of the result depending on where its significand lies.  Measured worst: 6.0; bound 8."""
    rng = np.random.default_rng(5)
    m = 3000
    mag = 10.0 ** rng.uniform(-6.0, 6.0, m)
    direction = rng.normal(size=(m, 3))
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    sep = direction * mag[:, None] * rng.uniform(1.0, 2.0, (m, 1))
    worst = 0.0
    with nb.NBody(2, fp64=True) as eng:
        got = {}
        for arith in (nb.ARITH_FMA3, nb.ARITH_STRICT):
            eng.set_option(nb.OPT_ARITH, arith)
            out = np.empty((m, 3))
            pos = np.zeros((2, 4))
            pos[:, 3] = 1.0
            for k in range(m):
                pos[1, :3] = sep[k]
                out[k] = eng.forces(pos)[0, :3]
            got[arith] = out
        a, b = got[nb.ARITH_FMA3], got[nb.ARITH_STRICT]
        assert np.isfinite(a).all() and np.isfinite(b).all()
        ulp = np.spacing(np.abs(b).max(axis=1))
        worst = float((np.abs(a - b).max(axis=1) / ulp).max())
    with capsys.disabled():
        print("\n[fp64 inverse cube] %d pairs, d2 in [1e-12, 4e12]: timed arithmetic vs IEEE chain, worst %.2f ulp of the pair's largest component" % (m, worst))
    assert worst <= 8.0
