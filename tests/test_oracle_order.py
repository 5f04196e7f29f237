"""The oracle's statement of the engine's summation order (REF_SUM_BLOCKED, ref_order_t in oracle/nbody_ref.h):
checked on CPU against an independent numpy emulation built from the plain sequential kernel, against the Python
mirror of the segmentation (mini_nbody_amd/sharding.py), and for what it is for — the error against fp64."""
import numpy as np
import pytest

import oracle as O


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def emulate(ora, pos, rows, nslices, sub, block, nb, wsplit=1):
    """segments in ascending order; a segment is cut into `wsplit` pieces (ceil(len / wsplit) sources each); inside a piece
    blocks of `block` sources, each summed by the sequential kernel from zero, block sums added in ascending order starting
    from +0; piece sums: S = w0, S = S + w_k; segment sums: F = p0, F = F + p_s"""
    n = len(pos)
    total = None
    for q in range(nslices):
        for t in range(sub):
            sb, se = nb.sharding.segment_bounds(q, t, n, nslices, sub)
            seg = None
            for w in range(wsplit):
                b, e = nb.sharding.piece_bounds(sb, se, w, wsplit)
                pc = np.zeros_like(rows)
                for j0 in range(b, e, block):
                    part = ora.forces_f32(rows, pos[j0:min(j0 + block, e)])
                    pc = (pc + part).astype(np.float32)
                seg = pc if seg is None else (seg + pc).astype(np.float32)
            total = seg if total is None else (total + seg).astype(np.float32)
    total[:, 3] = 0
    return total


@pytest.mark.parametrize("n,nslices,sub,block,wsplit", [(700, 1, 1, 64, 1), (700, 1, 1, 1024, 1), (1500, 3, 2, 128, 1), (1029, 8, 1, 64, 1),
                                                        (64, 1, 4, 64, 1), (5, 2, 2, 64, 1),
                                                        (700, 1, 1, 64, 4), (1500, 3, 2, 64, 4), (1029, 8, 1, 64, 4), (5, 2, 2, 64, 4), (3, 1, 1, 64, 4)])
def test_blocked_order_equals_emulation(nb, oracle, oracle_fast, n, nslices, sub, block, wsplit):
    pos, _ = nb.make_bodies(n, seed=n)
    for ora in (oracle, oracle_fast):
        got = ora.forces_order(pos, summ=O.SUM_BLOCKED, block=block, nslices=nslices, sub=sub, wsplit=wsplit)
        assert np.array_equal(bits(got), bits(emulate(ora, pos, pos, nslices, sub, block, nb, wsplit)))
    # the two builds agree bit for bit (the fast one is what the GPU tests use)
    assert np.array_equal(bits(oracle.forces_order(pos, block=block, nslices=nslices, sub=sub, wsplit=wsplit)),
                          bits(oracle_fast.forces_order(pos, block=block, nslices=nslices, sub=sub, wsplit=wsplit)))


def test_piece_bounds_tile_the_segment(nb):
    """the four waves' pieces: contiguous, ascending, ceil(len / 4) each, the last ones shorter or empty"""
    for sb, se in ((0, 0), (0, 1), (5, 8), (0, 4), (10, 15), (100, 380), (7, 131079)):
        prev = sb
        for w in range(4):
            b, e = nb.sharding.piece_bounds(sb, se, w, 4)
            assert b == prev and b <= e <= se and e - b <= -(-(se - sb) // 4)
            prev = e
        assert prev == se


def test_sequential_order_is_the_classic_entry_point(nb, oracle):
    pos, _ = nb.make_bodies(900, seed=1)
    for d2 in (O.D2_FMA3, O.D2_REFERENCE):
        for rs in (O.RSQRT_F64, O.RSQRT_DIVSQRT):
            a = oracle.forces_order(pos, order_=O.order(d2=d2, rsqrt=rs, summ=O.SUM_SEQ))
            assert np.array_equal(bits(a), bits(oracle.forces_f32(pos, d2=d2, rsqrt=rs)))
    # one block longer than the segment: blocked == sequential except that a sum of -0.0 becomes +0.0 (0 + x)
    a = oracle.forces_order(pos, summ=O.SUM_BLOCKED, block=4096)
    b = oracle.forces_f32(pos)
    assert np.array_equal(a, b)


def test_segment_bounds_agree_with_the_host_mirror(nb, oracle):
    for n, P, sub in ((10, 3, 2), (4099, 8, 2), (1 << 20, 8, 8), (7, 8, 1)):
        for q in range(P):
            for t in range(sub):
                assert oracle.segment_bounds(q, t, n, P, sub) == nb.sharding.segment_bounds(q, t, n, P, sub)


def test_step_order_is_forces_kick_drift(nb, oracle):
    n, dt = 600, np.float32(0.01)
    pos, vel = nb.make_bodies(n, seed=4)
    p, v = pos.copy(), vel.copy()
    oracle.step_order(p, v, 0.01, 2, summ=O.SUM_BLOCKED, block=64, sub=3)
    for _ in range(2):
        f = oracle.forces_order(pos, summ=O.SUM_BLOCKED, block=64, sub=3)
        vel[:, :3] = (np.float64(dt) * f[:, :3].astype(np.float64) + vel[:, :3]).astype(np.float32)     # one rounding: fma
        pos[:, :3] = (vel[:, :3].astype(np.float64) * np.float64(dt) + pos[:, :3]).astype(np.float32)
    assert np.array_equal(bits(p), bits(pos)) and np.array_equal(bits(v), bits(vel))


def test_blocked_sum_is_what_meets_the_tolerance(nb, oracle_fast):
    """N = 262144, 512 sampled rows, all sources: every row of the two-level sum within 1e-5 of fp64; the single
    sequential fp32 sum is not (this is why the engine does not use it)."""
    n = 1 << 18
    pos, _ = nb.make_bodies(n)
    rows = pos[np.r_[0:256, n - 256:n]]
    f64 = oracle_fast.forces_f64_from_f32(rows, pos)[:, :3]

    def worst(f):
        d = f[:, :3].astype(np.float64) - f64
        return float((np.sqrt((d ** 2).sum(1)) / np.sqrt((f64 ** 2).sum(1))).max())

    assert worst(oracle_fast.forces_order(rows, pos, summ=O.SUM_BLOCKED, sub=8)) < 5e-6
    assert worst(oracle_fast.forces_order(rows, pos, summ=O.SUM_BLOCKED, nslices=8, sub=8)) < 5e-6
    assert worst(oracle_fast.forces_f32(rows, pos)) > 1e-5


def emulate_f64(ora, pos, rows, nslices, sub, nb, wsplit):
    """the fp64 order: one sequential sum per piece (the plain fp64 kernel over that piece), piece sums S = w0, S = S + w_k,
    segment sums F = g0, F = F + g_s; an empty piece contributes +0"""
    n = len(pos)
    total = None
    for q in range(nslices):
        for t in range(sub):
            sb, se = nb.sharding.segment_bounds(q, t, n, nslices, sub)
            seg = None
            for w in range(wsplit):
                b, e = nb.sharding.piece_bounds(sb, se, w, wsplit)
                pc = ora.forces_f64(rows, pos[b:e]) if e > b else np.zeros_like(rows)
                seg = pc if seg is None else seg + pc
            total = seg if total is None else total + seg
    total[:, 3] = 0
    return total


@pytest.mark.parametrize("n,nslices,sub,wsplit", [(700, 1, 1, 1), (1500, 3, 2, 1), (1029, 8, 1, 4), (700, 1, 1, 16), (5, 2, 2, 4), (3, 1, 1, 16), (300, 1, 3, 4)])
def test_fp64_order_equals_emulation(nb, oracle, oracle_fast, n, nslices, sub, wsplit):
    """ref_forces_f64_order (round 4: what an fp64 context in NBODY_ARITH_STRICT reproduces bit for bit) against an emulation built
    from the plain sequential fp64 kernel; with one segment and one piece it IS that kernel"""
    pos, _ = nb.make_bodies(n, seed=n, dtype=np.float64)
    pos[:, :3] += nb.make_bodies(n, seed=n + 1, dtype=np.float64)[0][:, :3] * 2.0 ** -25      # full-width significands
    for ora in (oracle, oracle_fast):
        got = ora.forces_f64_order(pos, nslices=nslices, sub=sub, wsplit=wsplit)
        assert np.array_equal(got.view(np.uint64), emulate_f64(ora, pos, pos, nslices, sub, nb, wsplit).view(np.uint64))
    assert np.array_equal(oracle.forces_f64_order(pos, nslices=nslices, sub=sub, wsplit=wsplit).view(np.uint64),
                          oracle_fast.forces_f64_order(pos, nslices=nslices, sub=sub, wsplit=wsplit).view(np.uint64))
    if nslices == sub == wsplit == 1:
        assert np.array_equal(oracle.forces_f64_order(pos, nslices=1, sub=1, wsplit=1).view(np.uint64), oracle.forces_f64(pos).view(np.uint64))
