"""The strict 1/sqrt of the fp32 force kernels (NBODY_ARITH_STRICT): the VALUE is (float)(1.0 / sqrt((double)x)) — what the oracle computes
(oracle/nbody_ref.c REF_RSQRT_F64; the reference's own rsqrt is a Xilinx IP instance, S/fxyz.vhd:101-102, whose rounding is unpinned) —
but since round 4 it is obtained from eight binary32 operations on the v_rsq_f32 seed, with the IEEE expression only for arguments too
close to a rounding boundary (csrc/nbody_kernels.hpp rsqrt_strict_f32).  Three statements are tied together here:
  * a numpy model of the eight operations (CPU, seeds off by -2 ... +2 ulp): whatever it accepts equals the IEEE value;
  * on the GPU, the kernel's evaluation against the IEEE expression for EVERY binary32 bit pattern (nbody_rsqrt_selftest);
  * on the GPU, both against numpy's binary64 sqrt and divide and against the oracle's ref_rsqrt.
"""
import numpy as np
import pytest

LD = np.longdouble
BAND = np.float32(2.0 ** -16)


def fma32(a, b, c):
    # binary32 fma through the 64-bit significand of x87 long double: a*b is exact (48 bits), the sum is rounded at 64 bits and again
    # at 24 — a double rounding that can matter for ~2^-40 of operands
    return (a.astype(LD) * b.astype(LD) + c.astype(LD)).astype(np.float32)


def eight_operations(x, y):
    """rsqrt_fast_f32 of csrc/nbody_kernels.hpp, operation for operation; y is the seed.  Returns (value, accepted)."""
    one = np.ones_like(x)
    hi = x * y
    lo = fma32(x, y, -hi)
    e = fma32(-hi, y, one)
    e = fma32(-lo, y, e)
    t = y * e
    r1 = fma32(t, np.full_like(x, np.float32(0.5) + BAND), y)
    r2 = fma32(t, np.full_like(x, np.float32(0.5) - BAND), y)
    return r1, r1 == r2


def ieee(x):
    return (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)


def arguments(n, seed):
    rng = np.random.default_rng(seed)
    x = np.exp(rng.uniform(np.log(1e-9), np.log(1e9), n)).astype(np.float32)      # d2 >= eps = 1e-9f in the kernels
    x[: n // 8] = rng.uniform(1e-9, 12.0, n // 8).astype(np.float32)               # the range of d2 in a [-1, 1)^3 box
    return x


@pytest.mark.skipif(np.finfo(LD).nmant < 63, reason="needs the x87 64-bit significand for the fma model")
@pytest.mark.parametrize("ulps", [0, 1, -1, 2, -2])
def test_model_accepts_only_the_ieee_value(ulps):
    x = arguments(400_000, 7 + ulps)
    want = ieee(x)
    seed = want.copy()
    for _ in range(abs(ulps)):
        seed = np.nextafter(seed, np.float32(np.inf if ulps > 0 else 0.0))
    got, ok = eight_operations(x, seed)
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    rejected = 1.0 - ok.mean()
    print("seed off by %+d ulp: %.2e of the arguments go to the IEEE form" % (ulps, rejected))
    assert rejected < 2.0 ** -12          # ~2^-16 expected: the band is 2^-16 of the correction on either side of a midpoint


@pytest.mark.skipif(np.finfo(LD).nmant < 63, reason="needs the x87 64-bit significand for the fma model")
def test_model_limits():
    """What the acceptance test does and does not protect against.  Non-finite intermediates are always rejected (a NaN compares unequal
    to itself) and so is a seed that is no approximation at all (|e| > 2^-8 moves the two evaluations apart by more than an ulp).  A seed
    that is merely poor (error between ~2^-20 and 2^-9) is NOT caught: one Newton step then leaves 3/8 e^2 > the band.  v_rsq_f32's 1-ulp
    accuracy is therefore part of the contract — and is what the exhaustive GPU test below establishes, pattern by pattern."""
    x = arguments(100_000, 3)
    want = ieee(x)
    for factor in (1.5, 0.5, 1.0 + 2.0 ** -6):
        got, ok = eight_operations(x, (want * np.float32(factor)).astype(np.float32))
        assert not ok.any(), factor
    with np.errstate(all="ignore"):
        for bad in (np.inf, np.nan):
            got, ok = eight_operations(np.full(4, bad, np.float32), np.zeros(4, np.float32))
            assert not ok.any()
    got, ok = eight_operations(x, (want * np.float32(1.0 + 2.0 ** -12)).astype(np.float32))
    assert (ok & (got.view(np.uint32) != want.view(np.uint32))).any()      # the documented limit, kept visible


@pytest.mark.gpu
def test_every_binary32_bit_pattern(nb):
    rsqrt_selftest = nb.rsqrt_selftest
    # positive normal numbers: the arguments a force kernel can produce (d2 >= 1e-9f)
    bad, slow, first = rsqrt_selftest(0x00800000, 0x7F800000 - 0x00800000)
    n = 0x7F800000 - 0x00800000
    print("positive normals: %d patterns, %d mismatches, %d (%.3e) evaluated by the IEEE form" % (n, bad, slow, slow / n))
    assert bad == 0, hex(first)
    assert 0 < slow < n * 2.0 ** -13
    # ... and every other pattern (zeros, subnormals, infinities, NaNs, negatives)
    bad, slow, first = rsqrt_selftest(0, 1 << 32)
    print("all 2^32 patterns: %d mismatches, %d evaluated by the IEEE form" % (bad, slow))
    assert bad == 0, hex(first)


@pytest.mark.gpu
def test_kernel_evaluation_equals_numpy_and_oracle(nb):
    rsqrt_strict = nb.rsqrt_strict
    import oracle as O
    x = arguments(2_000_003, 11)
    x[:12] = np.array([1e-9, float.fromhex('0x1.12e0be826d695p-30'), 14.0, 56.0, 126.0, 224.0, 1.0, 0.1, 10.0, 4.0, 3.0, 2.0], np.float32)   # the testbench stimuli (tests/golden/kat_*.json)
    want = ieee(x)
    for only in (False, True):
        got = rsqrt_strict(x, ieee_only=only)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), only
    ref = O.Oracle()
    for k in range(0, 4096):
        assert np.float32(ref.rsqrt(float(x[k]), O.RSQRT_F64)).view(np.uint32) == want[k].view(np.uint32)
    special = np.array([np.inf, np.nan, 0.0, -0.0, -1.0, 1e-45, 1e-39, 3.4028235e38, 1.17549435e-38], np.float32)
    a, b = rsqrt_strict(special), rsqrt_strict(special, ieee_only=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with np.errstate(all="ignore"):
        w = ieee(special)
    assert np.array_equal(a[~np.isnan(w)].view(np.uint32), w[~np.isnan(w)].view(np.uint32)) and np.isnan(a[np.isnan(w)]).all()


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not __import__("os").path.exists(OBJDUMP), reason="needs ROCm's llvm-objdump")
def test_ieee_form_stays_behind_its_branch_in_the_shipped_library(tmp_path):
    """The point of the eight operations is lost if the compiler evaluates the IEEE form for every pair and selects (it did, in the LDS
    and lane-broadcast kernels, until the branch body was pinned): in the code object of libnbody_hip.so every strict fp32 kernel
    (one that holds both v_rsq_f32 and v_rsq_f64) must reach each v_rsq_f64 through a branch taken after the preceding v_rsq_f32."""
    import os
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = shutil.copy(os.path.join(root, "mini_nbody_amd", "libnbody_hip.so"), tmp_path)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp_path, check=True, capture_output=True)
    obj = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(obj) == 1, os.listdir(tmp_path)
    text = subprocess.run([OBJDUMP, "-d", os.path.join(tmp_path, obj[0])], check=True, capture_output=True, text=True).stdout
    kernels = re.split(r"^[0-9a-f]+ <([^>]+)>:$", text, flags=re.M)
    strict = 0
    for name, body in zip(kernels[1::2], kernels[2::2]):
        if "v_rsq_f32" not in body or "v_rsq_f64" not in body:
            continue
        strict += 1
        branched = False
        for line in body.splitlines():
            if "v_rsq_f32" in line:
                branched = False
            elif "s_cbranch" in line:
                branched = True
            elif "v_rsq_f64" in line:
                assert branched, name
    assert strict >= 20, strict          # smem / lds / readlane / fpga16 kernels x STRICT, REFERENCE_STRICT (+ the two self-test kernels)


@pytest.mark.gpu
def test_testbench_stimuli_through_the_kernels_evaluation(nb):
    """The reference's own test inputs at this boundary — T/tb_sqrt.vhd:494, 503, 528-541: 1.0, the ramp 0.1 ... 10.0, seven special values
    (tests/golden/kat_rsqrt.json; the testbench asserts only not-X, the outputs are the analytic values) — through the strict 1/sqrt as the
    force kernels evaluate it: every finite case equals the fixture bit for bit (the fixture's 1-ulp allowance is for the fast v_rsq_f32),
    NaN where the fixture says NaN.  The fast arithmetic's seed itself stays inside that allowance."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_rsqrt.json")))
    x = np.array([int(c["a"], 16) for c in d["cases"]], np.uint32).view(np.float32)
    got = nb.rsqrt_strict(x)
    for c, g in zip(d["cases"], got):
        if c["result"] == "nan":
            assert np.isnan(g), c["label"]
        else:
            assert int(g.view(np.uint32)) == int(c["result"], 16), (c["label"], hex(int(g.view(np.uint32))), c["result"])


def test_the_python_mirror_has_no_gate_of_its_own(nb):
    """The proof that admits the strict binary32 arithmetic lives in the LIBRARY (nbody_strict_proof, called by nbody_set_option and
    nbody_mailbox_open on every device of the context; tests/test_gpu_mailbox.py test_strict_arithmetic_is_refused_...): the mirror
    passes the option through and raises what the library answers."""
    import mini_nbody_amd.engine as eng_mod
    calls = []

    class Lib:
        def nbody_set_option(self, k, v):
            calls.append((k, v))
            return 0

    e = eng_mod.NBody.__new__(eng_mod.NBody)
    e.fp64, e.lib = False, Lib()
    e.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
    assert calls == [(nb.OPT_ARITH, nb.ARITH_STRICT)] and not hasattr(eng_mod, "_check_strict_rsqrt_once")
