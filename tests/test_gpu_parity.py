"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

Two bars, both written out here:
  * STRICT arithmetic (NBODY_ARITH_STRICT / _REFERENCE_STRICT): every operation is
    IEEE-exact, so the GPU must equal the oracle BIT FOR BIT — forces, and
    positions/velocities after any number of steps, for every delivery variant,
    register blocking and segmentation.
  * FAST arithmetic (v_rsq_f32, <= 1 ulp; the timed mode): the north_star's 1e-5
    relative fp32 tolerance on single-pass forces (max-norm), and on positions after
    the fixed step count in max-norm and in the median; element-wise it is compared
    with the spread two CPU restatements show when their 1/sqrt differs by 1 ulp
    (the dynamics with eps = 1e-9 amplify any last-bit difference — see DESIGN.md
    "Parity").
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


def set_variant(nb, eng, variant, iblock, jsub=1, jslices=1, arith=None, tile=None):
    eng.set_option(nb.OPT_VARIANT, {"smem": nb.VARIANT_SMEM, "lds": nb.VARIANT_LDS, "readlane": nb.VARIANT_READLANE}[variant])
    eng.set_option(nb.OPT_IBLOCK, iblock)
    eng.set_option(nb.OPT_JSUB, jsub)
    eng.set_option(nb.OPT_JSLICES, jslices)
    if arith is not None:
        eng.set_option(nb.OPT_ARITH, arith)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 257, 1000, 2085])
def test_strict_forces_bit_exact_ragged_sizes(nb, oracle_fast, engine_factory, n):
    pos, _ = nb.make_bodies(n, seed=n + 1)
    want = oracle_fast.forces_f32(pos, d2=O.D2_FMA3, rsqrt=O.RSQRT_F64)
    eng = engine_factory(n)
    for variant in VARIANTS:
        for iblock in (1, 4):
            set_variant(nb, eng, variant, iblock, arith=nb.ARITH_STRICT)
            got = eng.forces(pos)
            assert np.array_equal(bits(got), bits(want)), (variant, iblock)
    set_variant(nb, eng, "smem", 2, arith=nb.ARITH_REFERENCE_STRICT)
    want_ref = oracle_fast.forces_f32(pos, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64)
    assert np.array_equal(bits(eng.forces(pos)), bits(want_ref))


@pytest.mark.parametrize("tile", [256, 512, 1024])
def test_strict_lds_tiles(nb, oracle_fast, engine_factory, tile):
    n = 3000
    pos, _ = nb.make_bodies(n, seed=5)
    want = oracle_fast.forces_f32(pos)
    eng = engine_factory(n, tile=tile)
    set_variant(nb, eng, "lds", 2, arith=nb.ARITH_STRICT)
    assert np.array_equal(bits(eng.forces(pos)), bits(want))


def test_fast_variants_agree_bitwise_and_within_tolerance(nb, oracle_fast, engine_factory):
    """All delivery variants and register blockings run the same operations in the same order."""
    n = 4096 + 37
    pos, _ = nb.make_bodies(n, seed=11)
    eng = engine_factory(n)
    ref = None
    for variant in VARIANTS:
        for iblock in (1, 2, 4):
            set_variant(nb, eng, variant, iblock, arith=nb.ARITH_FMA3)
            got = eng.forces(pos)
            if ref is None:
                ref = got
            assert np.array_equal(bits(got), bits(ref)), (variant, iblock)
    set_variant(nb, eng, "smem", 8, arith=nb.ARITH_FMA3)
    assert np.array_equal(bits(eng.forces(pos)), bits(ref))
    # the hand-scheduled ISA loop (both code-placement phases): same operations, same order, same bits
    for phase in (0, 1):
        eng.set_option(nb.OPT_VARIANT, nb.VARIANT_ISA)
        eng.set_option(nb.OPT_ISA_PHASE, phase)
        assert eng.config["variant"] == "isa" and eng.config["iblock"] == 1
        assert np.array_equal(bits(eng.forces(pos)), bits(ref)), phase
        for jsub, jsl in ((3, 1), (2, 5)):
            eng.set_option(nb.OPT_JSUB, jsub)
            eng.set_option(nb.OPT_JSLICES, jsl)
            got = eng.forces(pos)
            eng.set_option(nb.OPT_VARIANT, nb.VARIANT_SMEM)
            assert np.array_equal(bits(got), bits(eng.forces(pos))), (phase, jsub, jsl)
            eng.set_option(nb.OPT_VARIANT, nb.VARIANT_ISA)
        eng.set_option(nb.OPT_JSUB, 1)
        eng.set_option(nb.OPT_JSLICES, 1)
    want = oracle_fast.forces_f32(pos)
    f64 = oracle_fast.forces_f64_from_f32(pos)
    assert maxnorm_rel(ref, want) < TOL
    # against the fp64 arbiter the GPU is as good as the CPU fp32 path
    assert maxnorm_rel(ref, f64) < max(2 * maxnorm_rel(want, f64), 2e-6)
    assert np.all(ref[:, 3] == 0)   # S/compute_store.vhd:242: the 4th word is 0


@pytest.mark.parametrize("n", [1, 7, 8, 9, 15, 16, 17, 100, 1031])
def test_isa_loop_equals_compiled_kernel_ragged(nb, engine_factory, n):
    """Default (hand-scheduled ISA loop, groups of 8 sources + scalar tail) vs the hipcc-scheduled kernel: same bits."""
    pos, vel = nb.make_bodies(n, seed=100 + n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_JSUB, 1)
    assert eng.config["variant"] == "isa"
    a = eng.forces(pos)
    eng.upload(pos, vel)
    eng.step(0.01, 3)
    pa, va = eng.download()
    eng.set_option(nb.OPT_VARIANT, nb.VARIANT_SMEM)
    assert eng.config["variant"] == "smem"
    assert np.array_equal(bits(a), bits(eng.forces(pos)))
    eng.upload(pos, vel)
    eng.step(0.01, 3)
    pb, vb = eng.download()
    assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(va), bits(vb))


def test_segmentation_matches_host_mirror_bitwise(nb, oracle_fast, engine_factory):
    """jslices x jsub segments combined in ascending order == the Python mirror of the decomposition
    (mini-nbody_amd/sharding.py) driven by the oracle, bit for bit in strict mode."""
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
                parts.append(oracle_fast.forces_f32(pos, pos[b:e]))
        want = nb.sharding.combine_ascending(parts)
        assert np.array_equal(bits(got), bits(want)), (nsl, sub)


def test_fpga16_order(nb, oracle_fast, engine_factory):
    """16 strided partials + pairwise tree, S/fxyz.vhd:129-184, S/final_adder.vhd:88-104."""
    for n in (5, 16, 100, 1031):
        pos, _ = nb.make_bodies(n, seed=2)
        eng = engine_factory(n)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
        eng.set_option(nb.OPT_JSUB, 1)
        eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE_STRICT)
        want = oracle_fast.forces_f32(pos, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(bits(eng.forces(pos)), bits(want)), n
        eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
        assert maxnorm_rel(eng.forces(pos), want) < TOL


def test_bodyForce_integrate_config1_shape(nb, oracle_fast, engine_factory):
    """BASELINE config 1's shape (N = 4096, 10 iterations) through the host-pointer entry points."""
    n, dt, iters = 4096, 0.01, 10
    pos, vel = nb.make_bodies(n)
    opos, ovel = pos.copy(), vel.copy()
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    eng.set_option(nb.OPT_JSUB, 1)          # one source segment == the oracle's single sequential sum
    for _ in range(iters):
        eng.bodyForce(pos, vel, dt)
        eng.integrate(pos, vel, dt)
        oracle_fast.bodyForce(opos, ovel, dt)
        oracle_fast.integrate(opos, ovel, dt)
    assert np.array_equal(bits(pos), bits(opos))
    assert np.array_equal(bits(vel), bits(ovel))


@pytest.mark.parametrize("jsub", [1, 4])
def test_step_loop_strict_bit_exact(nb, oracle_fast, engine_factory, jsub):
    n, dt, steps = 4096, 0.01, 10
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    eng.set_option(nb.OPT_JSUB, jsub)
    eng.upload(pos, vel)
    eng.step(dt, steps)
    gp, gv = eng.download()
    if jsub == 1:
        op, ov = pos.copy(), vel.copy()
        oracle_fast.step(op, ov, dt, steps)
    else:
        op, ov = pos.copy(), vel.copy()
        dtf = np.float32(dt)
        for _ in range(steps):
            parts = []
            for t in range(jsub):
                b, e = nb.sharding.segment_bounds(0, t, n, 1, jsub)
                parts.append(oracle_fast.forces_f32(op, op[b:e]))
            acc = nb.sharding.combine_ascending(parts)
            # kick and drift with one rounding each (fma): emulate in float64, exact for fp32 inputs
            ov[:, :3] = (dtf.astype(np.float64) * acc[:, :3].astype(np.float64) + ov[:, :3]).astype(np.float32)
            op[:, :3] = (ov[:, :3].astype(np.float64) * dtf.astype(np.float64) + op[:, :3]).astype(np.float32)
    assert np.array_equal(bits(gp), bits(op))
    assert np.array_equal(bits(gv), bits(ov))
    assert np.all(gp[:, 3] == 1) and np.all(gv[:, 3] == 0)


def test_step_loop_fast_within_tolerance(nb, oracle_fast, engine_factory):
    """The timed mode (v_rsq_f32) after the fixed step count, N = 4096, 10 steps, same summation order
    as the oracle (one source segment) so that the only difference is the 1-ulp 1/sqrt."""
    n, dt, steps = 4096, 0.01, 10
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n)
    eng.set_option(nb.OPT_JSUB, 1)
    eng.upload(pos, vel)
    eng.step(dt, 1)
    g1, _ = eng.download()
    eng.step(dt, steps - 1)
    gp, gv = eng.download()
    o1, ov1 = pos.copy(), vel.copy()
    oracle_fast.step(o1, ov1, dt, 1)
    op, ov = pos.copy(), vel.copy()
    oracle_fast.step(op, ov, dt, steps)
    # a second CPU restatement whose 1/sqrt differs by <= 1 ulp: the envelope of "equally right" answers
    e1, ev1 = pos.copy(), vel.copy()
    oracle_fast.step(e1, ev1, dt, 1, rsqrt=O.RSQRT_DIVSQRT)
    ep, ev = pos.copy(), vel.copy()
    oracle_fast.step(ep, ev, dt, steps, rsqrt=O.RSQRT_DIVSQRT)

    def elementwise(a, b):
        return np.abs(a[:, :3] - b[:, :3]) / np.maximum(np.abs(b[:, :3]), 1e-30)

    # one step: no amplification yet -> the plain tolerance, in max-norm and for 99 % of the components
    assert maxnorm_rel(g1, o1) < TOL
    el1, env1 = elementwise(g1, o1), elementwise(e1, o1)
    assert np.quantile(el1, 0.99) < TOL
    assert np.quantile(el1, 0.999) < 4 * max(np.quantile(env1, 0.999), TOL)
    # fixed step count: median within tolerance, and no worse than the CPU-vs-CPU envelope
    el, env = elementwise(gp, op), elementwise(ep, op)
    assert np.median(el) < TOL
    assert np.isfinite(gp).all() and np.isfinite(gv).all()
    assert maxnorm_rel(gp, op) < 4 * max(maxnorm_rel(ep, op), 1e-6)
    assert np.quantile(el, 0.99) < 4 * max(np.quantile(env, 0.99), TOL)


def test_config2_n65536(nb, oracle_fast, engine_factory):
    """BASELINE config 2: N = 65536, LDS tile = 256 and the default SMEM variant.
    Strict mode bit-exact over 3 steps (1.3e10 pairs on the host); 100 steps on the GPU stay finite
    and agree between the two variants bit for bit."""
    n, dt = 65536, 0.01
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n, tile=256)
    op, ov = pos.copy(), vel.copy()
    oracle_fast.step(op, ov, dt, 3)
    for variant in ("lds", "smem"):
        set_variant(nb, eng, variant, 4, jsub=1, arith=nb.ARITH_STRICT)
        eng.upload(pos, vel)
        eng.step(dt, 3)
        gp, gv = eng.download()
        assert np.array_equal(bits(gp), bits(op)), variant
        assert np.array_equal(bits(gv), bits(ov)), variant
    out = []
    for variant in ("lds", "smem"):
        set_variant(nb, eng, variant, 0, jsub=0, arith=nb.ARITH_FMA3)
        eng.set_option(nb.OPT_JSUB, 8)
        eng.upload(pos, vel)
        eng.step(dt, 100)
        out.append(eng.download())
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(bits(out[0][0]), bits(out[1][0]))
    # fast-mode forces of the initial state: same summation order as the oracle (one segment) -> only the
    # 1-ulp 1/sqrt differs; and against the fp64 arbiter the GPU is as accurate as the CPU fp32 path
    # (at this N two fp32 sums in DIFFERENT orders already differ by ~1e-5 of the largest force).
    set_variant(nb, eng, "smem", 4, jsub=1, arith=nb.ARITH_FMA3)
    f = eng.forces(pos)
    want = oracle_fast.forces_f32(pos)
    f64 = oracle_fast.forces_f64_from_f32(pos)
    assert maxnorm_rel(f, want) < TOL
    assert maxnorm_rel(f, f64) < 2 * maxnorm_rel(want, f64) + 1e-6
    set_variant(nb, eng, "smem", 0, jsub=0, arith=nb.ARITH_FMA3)      # the auto configuration (segmented)
    assert maxnorm_rel(eng.forces(pos), f64) < 2 * maxnorm_rel(want, f64) + 1e-6


def test_headline_size_row_sample_and_properties(nb, oracle_fast, engine_factory):
    """N = 1,048,576 (BASELINE config 3): one CPU pass is 1.1e12 pairs, so parity is row-sampled
    (SURVEY.md §7 "Hard parts"): 1024 rows incl. the first and last body, all N sources, strict mode
    bit-exact and fast mode within tolerance; plus size-independent properties on the full state."""
    n = 1 << 20
    pos, vel = nb.make_bodies(n)
    eng = engine_factory(n)
    eng.upload(pos, vel)
    sample = [(0, 256), (n // 2 - 128, 256), (n - 512, 512)]
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    eng.set_option(nb.OPT_JSUB, 1)
    for first, cnt in sample:
        got = eng.forces_rows(first, cnt)
        want = oracle_fast.forces_f32(pos[first:first + cnt], pos)
        assert np.array_equal(bits(got), bits(want)), first
    eng.set_option(nb.OPT_ARITH, nb.ARITH_FMA3)
    for first, cnt in sample[:1]:
        got = eng.forces_rows(first, cnt)
        want = oracle_fast.forces_f32(pos[first:first + cnt], pos)
        assert maxnorm_rel(got, want) < TOL
    # properties: a step moves exactly r' = r + v'*dt (checked from the downloaded state), w carried
    eng.step(0.01, 1)
    p1, v1 = eng.download()
    dt = np.float64(np.float32(0.01))
    assert np.array_equal(bits(p1[:, :3]), bits((v1[:, :3].astype(np.float64) * dt + pos[:, :3]).astype(np.float32)))
    assert np.all(p1[:, 3] == 1) and np.all(v1[:, 3] == 0) and np.isfinite(v1).all()
    # momentum: unit masses and antisymmetric pair terms -> sum_i F_i ~ 0 relative to sum_i |F_i|
    dv = (v1[:, :3].astype(np.float64) - vel[:, :3]) / dt
    assert np.abs(dv.sum(0)).max() / np.abs(dv).sum() < 1e-4


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
    for arith, d2 in ((nb.ARITH_STRICT, O.D2_FMA3), (nb.ARITH_REFERENCE_STRICT, O.D2_REFERENCE)):
        eng.set_option(nb.OPT_ARITH, arith)
        eng.set_option(nb.OPT_JSUB, 1)
        with np.errstate(all="ignore"):
            want = oracle_fast.forces_f32(pos, d2=d2, rsqrt=O.RSQRT_F64)
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
    eng.set_option(nb.OPT_JSUB, 1)
    ram_b = nb.mailbox.run(eng, ram_a)
    assert ram_b.shape == (n, 4) and nb.mailbox.decode_control(ram_a)["begin"] == 0
    rows = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
    want = oracle_fast.forces_f32(pos[rows], pos)
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
    eng.set_option(nb.OPT_JSUB, 1)
    ram_b = nb.mailbox.run(eng, ram_a, clock_khz=300000)
    ctl = nb.mailbox.decode_control(ram_a)
    assert ctl["begin"] == 0 and ctl["ticks"] >= 1       # S/top_level.vhd:146, 255-263
    assert np.array_equal(bits(ram_b), bits(oracle_fast.forces_f32(pos)))
    assert np.all(ram_b[:, 3] == 0)
    with pytest.raises(nb.NBodyError):                    # BEGIN not set -> nothing to do
        nb.mailbox.run(eng, ram_a)


def test_fp64_path(nb, oracle_fast, engine_factory):
    """BASELINE config 5's arithmetic at a size the host can check: fp64 forces and 5 steps."""
    n = 4096
    pos, vel = nb.make_bodies(n, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    f = eng.forces(pos)
    want = oracle_fast.forces_f64(pos)
    assert maxnorm_rel(f, want) < 1e-13
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


@pytest.mark.parametrize("n", [1, 3, 4, 5, 9, 250, 1031])
def test_fp64_isa_loop_equals_compiled_kernel(nb, oracle_fast, engine_factory, n):
    """fp64 default (hand-scheduled loop, 4 sources per iteration + scalar tail) vs the hipcc-scheduled kernel: same bits;
    and both within fp64 tolerance of the oracle."""
    pos, vel = nb.make_bodies(n, seed=200 + n, dtype=np.float64)
    eng = engine_factory(n, fp64=True)
    eng.set_option(nb.OPT_JSUB, 1)
    assert eng.config["variant"] == "isa"
    a = eng.forces(pos)
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
    """The multi-GPU schedule (i-sharding, ring-ordered arrival, per-slice partials, ascending combine,
    double-buffered positions) with P virtual ranks sharing this box's one GPU; the transfers are peer
    copies, everything else is the code the 8-GPU run executes.  Must equal one GPU configured with
    the same segmentation bit for bit, in fast mode, over several steps."""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n, dt, steps = 8192 + 5, 0.01, 4
    pos, vel = nb.make_bodies(n, seed=21)
    results = {}
    for P in (1, 2, 3, 4):
        for overlap in ((1,) if P == 1 else (1, 0)):
            eng = engine_factory(n, ngpus=P)
            eng.set_option(nb.OPT_IBLOCK, 2)
            eng.set_option(nb.OPT_JSUB, 2)
            eng.set_option(nb.OPT_OVERLAP, overlap)
            eng.upload(pos, vel)
            eng.step(dt, steps)
            results[(P, overlap)] = eng.download()
            # reference: one GPU, same segmentation
            one = engine_factory(n)
            one.set_option(nb.OPT_IBLOCK, 2)
            one.set_option(nb.OPT_JSUB, 2)
            one.set_option(nb.OPT_JSLICES, P)
            one.upload(pos, vel)
            one.step(dt, steps)
            wp, wv = one.download()
            gp, gv = results[(P, overlap)]
            assert np.array_equal(bits(gp), bits(wp)), (P, overlap)
            assert np.array_equal(bits(gv), bits(wv)), (P, overlap)


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
