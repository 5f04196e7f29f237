"""The reference's ONLY interface — the memory-mapped mailbox — as one long-lived context:
NUM_PTS sampled with every BEGIN (S/top_level.vhd:180-186) against RAMs sized once (:45), N = 0 completing at once with RAM B
untouched (:189-192), the force of body k at word k of RAM B, word 0 and the words beyond N never written (S/compute_store.vhd:221-242;
the cycle model: tests/test_fpga_store_model.py), word 0 of RAM A rewritten with the tick count and BEGIN = 0 (S/top_level.vhd:146,
255-263) — by the device itself.  All through the C-ABI (nbody_mailbox_open / _rams / _run / _serve)."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

SENTINEL = np.uint32(0xDEADBEEF)
RTL = {json.load(open(f))["n"]: f for f in glob.glob(os.path.join(HERE, "golden", "rtl_*.json"))}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def fixture(n):
    d = json.load(open(RTL[n]))
    words = lambda key: np.array([int(x, 16) for x in d[key]], np.uint32).view(np.float32).reshape(-1, 4)   # noqa: E731
    return words("pos0"), words("forces0")


def untouched(ram_b, n):
    """word 0 of RAM B and every word beyond N still hold the sentinel: S/compute_store.vhd:221-242 never writes them"""
    w = ram_b.view(np.uint32)
    return bool(np.all(w[0] == SENTINEL) and np.all(w[n + 1:] == SENTINEL))


def rtl_oracle(ora, rows, src):
    """the RTL's rounding points, 1/sqrt rounded once, sixteen partial sums + rotation + tree over ONE stream of all N"""
    return ora.forces_f32(rows, src, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)


def test_one_context_serves_every_num_pts_bit_for_bit(nb, oracle_fast):
    """ONE faithful context of the RTL's capacity replays rtl_n9 -> rtl_n100 -> rtl_n40 (exact-rational third statements of the RTL,
    tests/golden/make_system.py) -> N = 32767 (row samples against the oracle) -> N = 0 -> rtl_n9 again, on the context's own RAMs
    (no host copy) — every RAM B image bit for bit at words 1..N, word 0 and every word beyond N left exactly as they were; the tick
    word is the DEVICE's count of the request (ingest kernel's first wave to the store of word 0), not the host's latency."""
    big = nb.mailbox.MAX_POINTS
    pos_big, _ = nb.make_bodies(big, seed=12)
    with nb.Mailbox() as mb:                                   # capacity 32767 = ram_depth - 1, faithful
        assert mb.capacity == big and mb.ram_a.shape == (big + 1, 4) and mb.ram_b.shape == (big + 1, 4)
        for n in (9, 100, 40, big, 0, 9):
            mb.ram_b.view(np.uint32)[...] = SENTINEL
            pos = pos_big if n == big else (np.zeros((0, 4), np.float32) if n == 0 else fixture(n)[0])
            assert mb.post(pos) == n
            out, ticks = mb.run(clock_khz=300000)
            assert len(out) == n
            ctl = nb.mailbox.decode_control(mb.ram_a)
            assert ctl["begin"] == 0 and ctl["ticks"] == ticks >= 1                 # S/top_level.vhd:146, 255-263
            assert np.all(mb.ram_a[0, [0, 2, 3]] == 0)                              # {ticks in 63:32, 0 elsewhere}
            assert untouched(mb.ram_b, n), n                                        # S/compute_store.vhd:221-242
            if n == 0:
                assert ticks <= 2                                                   # straight to `complete`, S/top_level.vhd:189-192
            elif n == big:
                rows = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
                assert np.array_equal(bits(out[rows]), bits(rtl_oracle(oracle_fast, pos[rows], pos)))
                assert np.all(out[:, 3] == 0)
                assert 30 < ticks < 3000                                            # ~0.5 ms of device time at 300 MHz = ~150 ticks
            else:
                assert np.array_equal(bits(out), bits(fixture(n)[1])), n
                assert np.all(out[:, 3] == 0)                                       # {Fx, Fy, Fz, 0}, S/compute_store.vhd:242
                assert ticks <= 30                                                  # < 100 us of DEVICE time (the host's 20 us of latency would be 7)


def test_callers_own_ram_images(nb):
    """Any host buffers do as RAM A / RAM B (one host copy each way): same bits at words 1..N, and the caller's RAM B word 0 and beyond
    word N is untouched."""
    with nb.Mailbox(capacity=128) as mb:
        ram_b = np.full((129, 4), SENTINEL, np.uint32).view(np.float32)
        for n in (100, 9, 0, 40):
            ram_b.view(np.uint32)[...] = SENTINEL
            pos = fixture(n)[0] if n else np.zeros((0, 4), np.float32)
            ram_a = nb.mailbox.encode_request(pos)
            out = nb.mailbox.run(mb, ram_a, clock_khz=300000, ram_b=ram_b)
            assert nb.mailbox.decode_control(ram_a)["begin"] == 0
            assert untouched(ram_b, n)
            if n:
                assert np.array_equal(bits(out[:n]), bits(fixture(n)[1]))


def test_requests_the_fsm_does_not_take(nb):
    lib = nb._lib.load()
    with nb.Mailbox(capacity=64) as mb:
        # BEGIN not set: the FSM stays in `waiting` (S/top_level.vhd:180-186): nothing read, nothing written
        mb.ram_b.view(np.uint32)[...] = SENTINEL
        mb.post(fixture(9)[0])
        mb.ram_a[0, 0] = 0
        before = mb.ram_a.copy()
        with pytest.raises(nb.NBodyError) as e:
            mb.run()
        assert e.value.code == nb._lib.ERR_STATE
        assert np.array_equal(mb.ram_a, before) and np.all(mb.ram_b.view(np.uint32) == SENTINEL)
        # NUM_PTS beyond this context's capacity (the RTL's RAM always holds 32767: the capacity is the library's notion)
        mb.ram_a[0] = (1, 65, 0, 0)
        with pytest.raises(nb.NBodyError) as e:
            mb.run()
        assert e.value.code == nb._lib.ERR_ARG and mb.ram_a[0, 0] == 1
        # ... and the context still serves
        assert np.array_equal(bits(mb.forces(fixture(40)[0])), bits(fixture(40)[1]))
    assert lib.nbody_mailbox_open(40000, 1) == nb._lib.ERR_ARG                    # 15-bit NUM_PTS, S/top_level.vhd:45
    assert lib.nbody_mailbox_open(-1, 1) == nb._lib.ERR_ARG
    v = C.c_longlong()
    assert lib.nbody_get_info(nb._lib.INFO_N, C.byref(v)) == nb._lib.ERR_NOT_INIT   # a refused open leaves no context


def test_default_arithmetic_mailbox_is_within_tolerance(nb):
    """faithful = 0: the engine's timed arithmetic (fma-contracted d2, v_rsq_f32, blocked sums) — within 1e-5 of the RTL's answer and
    not equal to it, for every size in one context."""
    with nb.Mailbox(capacity=4096, faithful=False) as mb:
        for n in (100, 9, 40):
            want = fixture(n)[1][:, :3].astype(np.float64)
            got = mb.forces(fixture(n)[0])[:, :3].astype(np.float64)
            d = np.abs(got - want).max() / np.abs(want).max()
            assert 0 < d < 1e-5, (n, d)


def test_requests_inside_an_nbody_init_context_leave_it_as_it_was(nb, oracle_fast):
    """Any one-GPU fp32 context is a mailbox of capacity n.  A request of another size switches N for its own duration only: the
    context's N, options and captured step graph survive, and after a fresh upload the step loop gives the bits it gave before."""
    n = 1000
    pos, vel = nb.make_bodies(n, seed=5)
    small, _ = nb.make_bodies(300, seed=6)
    with nb.NBody(n) as eng:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        eng.upload(pos, vel)
        eng.step(0.01, 8)
        p0, v0 = eng.download()
        order0, cfg0 = eng.order, eng.config
        out = nb.mailbox.run(eng, nb.mailbox.encode_request(small))
        assert eng.info(nb._lib.INFO_N) == n and eng.order == order0 and eng.config == cfg0
        eng.upload(pos, vel)
        eng.step(0.01, 8)
        p1, v1 = eng.download()
        assert np.array_equal(bits(p0), bits(p1)) and np.array_equal(bits(v0), bits(v1))
    # the 300-body request itself: strict arithmetic in the order a 300-body context resolves to
    with nb.NBody(300) as e300:
        e300.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        want = oracle_fast.forces_order(small, None, order_=O.order(rsqrt=O.RSQRT_F64, **e300.order))
    assert np.array_equal(bits(out), bits(want))


def test_fpga_order_segmentation_does_not_depend_on_the_wave_split(nb):
    """with NBODY_SUM_FPGA16 the sixteen waves are not a split of the segment, so with NBODY_OPT_JSUB automatic the
    segmentation — hence every bit — is the same for NBODY_OPT_WSPLIT 1, -1 and 16."""
    for n in (1000, 4096, 20000):
        pos, _ = nb.make_bodies(n, seed=n)
        with nb.NBody(n) as eng:
            eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
            got = {}
            for ws in (1, -1, 16):
                eng.set_option(nb.OPT_WSPLIT, ws)
                got[ws] = (eng.config["nseg"], bits(eng.forces(pos)).copy())
            assert got[1][0] == got[-1][0] == got[16][0], n
            assert np.array_equal(got[1][1], got[-1][1]) and np.array_equal(got[1][1], got[16][1]), n


PROOF_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import mini_nbody_amd as nb
L = nb._lib
try:
    nb.Mailbox(capacity=64, faithful=True)
    print("mailbox: granted")
except nb.NBodyError as e:
    print("mailbox:", e.code, "refused" in str(e))
import ctypes as C
v = C.c_longlong()
print("context after refusal:", L.load().nbody_get_info(L.INFO_N, C.byref(v)))
with nb.NBody(64) as eng:
    try:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        print("option: granted")
    except nb.NBodyError as e:
        print("option:", e.code)
    eng.set_option(nb.OPT_ARITH, nb.ARITH_REFERENCE)          # not a strict mode: no proof needed
    print("proof:", nb.strict_proof())
with nb.NBody(64, fp64=True) as eng:
    eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)              # fp64 strict is IEEE sqrt and divide themselves
    print("fp64 strict: granted")
with nb.Mailbox(capacity=64, faithful=False) as mb:
    print("plain mailbox: granted")
"""


def test_strict_arithmetic_is_refused_on_a_device_that_fails_the_proof(nb):
    """The library itself gates the strict binary32 arithmetic on nbody_strict_proof (every positive normal binary32 on every device
    of the context): a device that fails it — simulated with NBODY_STRICT_PROOF_FAIL=1 in a fresh process — gets NBODY_ERR_UNSUPPORTED
    from nbody_set_option and from nbody_mailbox_open(., 1), which then leaves no context; everything that needs no proof still works."""
    env = dict(os.environ, NBODY_STRICT_PROOF_FAIL="1")
    p = subprocess.run([sys.executable, "-c", PROOF_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "mailbox: %d True" % nb._lib.ERR_UNSUPPORTED, lines
    assert lines[1] == "context after refusal: %d" % nb._lib.ERR_NOT_INIT, lines
    assert lines[2] == "option: %d" % nb._lib.ERR_UNSUPPORTED, lines
    assert lines[3].startswith("proof: (1,"), lines
    assert lines[4:] == ["fp64 strict: granted", "plain mailbox: granted"], lines
    # and on the real device the proof holds
    assert nb.strict_proof() == (0, 0)


def test_repeated_and_alternating_requests_stay_bit_exact(nb, oracle_fast):
    """The steady state of a PS driver: the same sizes again and again, in any order.  Every answer must be the first answer — twelve sizes
    repeated and alternated in both arithmetics, and inside an nbody_init context whose position buffer toggles with every step."""
    pos_all, _ = nb.make_bodies(4096, seed=31)
    sizes = [9, 100, 40, 777, 1024, 63, 65, 2085, 4096, 300, 1, 2]
    for faithful in (True, False):
        with nb.Mailbox(capacity=4096, faithful=faithful) as mb:
            first = {}
            for rnd in range(4):
                for n in (sizes if rnd % 2 == 0 else sizes[::-1]):
                    got = mb.forces(pos_all[:n])
                    if n not in first:
                        first[n] = got
                        if faithful and n <= 1024:
                            assert np.array_equal(bits(got), bits(rtl_oracle(oracle_fast, pos_all[:n], pos_all[:n]))), n
                    assert np.array_equal(bits(got), bits(first[n])), (faithful, rnd, n)
            for _ in range(5):                                             # the steady state of a PS driver: one size, again and again
                assert np.array_equal(bits(mb.forces(pos_all[:777])), bits(first[777]))
    pos, vel = nb.make_bodies(1000, seed=5)
    with nb.NBody(1000) as eng:
        eng.upload(pos, vel)
        want = None
        for k in range(6):
            out = nb.mailbox.run(eng, nb.mailbox.encode_request(pos_all[:300]))
            want = out if want is None else want
            assert np.array_equal(bits(out), bits(want)), k
            if k % 2 == 1:
                eng.upload(pos, vel)
                eng.step(0.01, 1)                                          # the current position buffer toggles
                eng.sync()


def test_served_mailbox_needs_no_call_per_request(nb, oracle_fast):
    """nbody_mailbox_serve: a library thread plays the PL block's FSM (S/top_level.vhd:180-186, 255-263) — the driver writes the bodies,
    then word 0 with NUM_PTS and BEGIN, and polls word 0 until BEGIN reads 0; nothing is called per request.  Same bits as the called
    form, same protocol: RAM B untouched from word N on, NUM_PTS = 0 completes, a request the library cannot take comes back with its
    error code in bits 127:96 of word 0 (where the RTL always writes 0) and BEGIN cleared."""
    with nb.Mailbox(capacity=2048, faithful=True) as mb:
        called = {n: mb.forces(fixture(n)[0]) for n in (9, 100, 40)}
        before = mb.served()
        mb.serve(True, clock_khz=300000)
        try:
            with pytest.raises(nb.NBodyError) as e:                # the service thread owns the mailbox now: the called form is refused
                nb.mailbox.run(mb, nb.mailbox.encode_request(fixture(9)[0]))
            assert e.value.code == nb._lib.ERR_STATE
            count = 0
            for rnd in range(3):
                for n in (9, 100, 0, 40):
                    mb.ram_b.view(np.uint32)[...] = SENTINEL
                    mb.post(fixture(n)[0] if n else np.zeros((0, 4), np.float32))        # BEGIN is the last thing post() writes
                    out, ticks = mb.wait()
                    count += 1
                    assert ticks >= 1 and np.all(mb.ram_a[0, [0, 2, 3]] == 0)
                    assert untouched(mb.ram_b, n), n
                    if n:
                        assert np.array_equal(bits(out), bits(called[n])) and np.array_equal(bits(out), bits(fixture(n)[1])), n
            big, _ = nb.make_bodies(2048, seed=3)
            mb.post(big)
            out, _ = mb.wait()
            count += 1
            assert np.array_equal(bits(out[:64]), bits(rtl_oracle(oracle_fast, big[:64], big)))
            mb.ram_a[0] = (1, 2049, 0, 0)                          # beyond the capacity
            with pytest.raises(nb.NBodyError) as e:
                mb.wait()
            count += 1
            assert e.value.code == nb._lib.ERR_ARG and mb.ram_a[0, 0] == 0 and mb.ram_a[0, 1] == 0
            for _ in range(1000):                                  # (the device clears BEGIN itself; the thread's counter follows within microseconds)
                if mb.served() - before == count:
                    break
                time.sleep(0.001)
            assert mb.served() - before == count
        finally:
            mb.serve(False)
        assert np.array_equal(bits(mb.forces(fixture(40)[0])), bits(called[40]))           # the called form works again
    # shutdown with the thread still serving: it is stopped and joined first
    mb = nb.Mailbox(capacity=64)
    mb.serve(True)
    mb.post(fixture(9)[0])
    mb.wait()
    mb.close()


@pytest.mark.parametrize("faithful", [True, False])
def test_num_pts_1_2_3_are_answered_a_stated_departure_from_the_rtl(nb, oracle_fast, faithful):
    """The RTL as written never raises TRGT_VALID for NUM_PTS <= uram_latency = 3 (S/top_level.vhd:42, 234-254; the cycle model:
    tests/test_fpga_fsm_model.py test_i_*): the pass completes with RAM B untouched.  That is a defect of the sequencer, not protocol; the
    library computes the forces for 1, 2 and 3 bodies like for any other N ("Departures from the RTL as written")."""
    pos, _ = nb.make_bodies(3, seed=9)
    with nb.Mailbox(capacity=64, faithful=faithful) as mb:
        for n in (1, 2, 3):
            mb.ram_b.view(np.uint32)[...] = SENTINEL
            mb.post(pos[:n])
            out, ticks = mb.run(300000)
            assert ticks >= 1 and mb.ram_a[0, 0] == 0 and untouched(mb.ram_b, n)
            want = rtl_oracle(oracle_fast, pos[:n], pos[:n])
            if faithful:
                assert np.array_equal(bits(out), bits(want)), n
            else:
                assert np.abs(out[:, :3].astype(np.float64) - want[:, :3]).max() <= 1e-5 * max(1e-30, np.abs(want[:, :3]).max()), n
            if n == 1:
                assert np.all(out == 0)                                             # one body: only the self pair, exactly zero force


def test_the_served_context_is_guarded_against_the_callers_thread(nb):
    """While nbody_mailbox_serve(1, .) is in effect the service thread owns the context: every entry point that
    launches, copies or reconfigures answers NBODY_ERR_STATE, and nbody_get_info reports the CONTEXT's N and configuration — never the N a
    request has switched in for its own duration.  A second thread hammers both while 10^4 served requests of alternating sizes stay
    bit-exact."""
    L = nb._lib
    lib = L.load()
    cap = 1024
    pos_all, _ = nb.make_bodies(cap, seed=21)
    sizes = (9, 700, 40, 100, 1, 333)
    with nb.Mailbox(capacity=cap, faithful=True) as mb:
        first = {n: mb.forces(pos_all[:n]) for n in sizes}
        cfg0 = {k: C.c_longlong() for k in (L.INFO_N, L.INFO_N_LOCAL, L.INFO_NSEG, L.INFO_JSUB, L.INFO_WSPLIT, L.INFO_VARIANT, L.INFO_FUSE_COMBINE)}
        for k, v in cfg0.items():
            assert lib.nbody_get_info(k, C.byref(v)) == 0
        assert cfg0[L.INFO_N].value == cap
        mb.serve(True, 300000)
        stop, seen = threading.Event(), {"info": 0, "refused": 0, "bad": []}
        buf = np.zeros((cap, 4), np.float32)
        bs = L.BodySystem(buf.ctypes.data_as(C.POINTER(C.c_float)), buf.ctypes.data_as(C.POINTER(C.c_float)))

        def hammer():
            v = C.c_longlong()
            refused = [lambda: lib.nbody_step(C.c_float(0.01), 1), lambda: lib.nbody_set_option(L.OPT_JSUB, 2), lambda: lib.nbody_upload(C.byref(bs)),
                       lambda: lib.nbody_download(C.byref(bs)), lambda: lib.nbody_forces(buf.ctypes.data_as(C.POINTER(C.c_float)), buf.ctypes.data_as(C.POINTER(C.c_float)), cap),
                       lambda: lib.bodyForce(buf.ctypes.data_as(C.POINTER(C.c_float)), buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(0.01), cap),
                       lambda: lib.nbody_sync(), lambda: lib.nbody_mailbox_open(64, 0), lambda: lib.nbody_init(64, 1, 0, 0),
                       lambda: lib.nbody_mailbox_run(mb.ram_a.ctypes.data_as(C.c_void_p), mb.ram_b.ctypes.data_as(C.c_void_p), 0)]
            k = 0
            while not stop.is_set():
                for key, want in cfg0.items():
                    rc = lib.nbody_get_info(key, C.byref(v))
                    if rc != 0 or v.value != want.value:
                        seen["bad"].append(("info", key, rc, v.value, want.value))
                    seen["info"] += 1
                rc = refused[k % len(refused)]()
                if rc != L.ERR_STATE:
                    seen["bad"].append(("call", k % len(refused), rc))
                seen["refused"] += 1
                k += 1

        th = threading.Thread(target=hammer)
        th.start()
        try:
            for k in range(10000):
                n = sizes[k % len(sizes)]
                mb.post(pos_all[:n])
                out, ticks = mb.wait()
                if ticks < 1 or not np.array_equal(bits(out), bits(first[n])):
                    raise AssertionError("served request %d (N = %d) differs from the called form" % (k, n))
        finally:
            stop.set()
            th.join()
            mb.serve(False)
        assert not seen["bad"], seen["bad"][:5]
        assert seen["info"] > 1000 and seen["refused"] > 100, seen
        # ... and the context is the caller's again
        assert lib.nbody_sync() == 0 and np.array_equal(bits(mb.forces(pos_all[:40])), bits(first[40]))


def test_mailbox_on_a_context_over_several_devices_keeps_the_address_map(nb, monkeypatch):
    """A context over several devices keeps its fixed N (NUM_PTS must equal it) and answers through nbody_forces' path; the map is the
    same: force of body k at word k of the caller's RAM B, word 0 and the words beyond N untouched, word 0 of RAM A rewritten by the host."""
    monkeypatch.setenv("NBODY_OVERSUBSCRIBE", "1")
    n = 1000
    pos, _ = nb.make_bodies(n, seed=17)
    with nb.NBody(n, ngpus=3) as eng:
        want = eng.forces(pos)
        ram_b = np.full((n + 3, 4), SENTINEL, np.uint32).view(np.float32)
        ram_a = nb.mailbox.encode_request(pos)
        out = nb.mailbox.run(eng, ram_a, clock_khz=300000, ram_b=ram_b)
        ctl = nb.mailbox.decode_control(ram_a)
        assert ctl["begin"] == 0 and ctl["ticks"] >= 1 and untouched(ram_b, n)
        assert np.array_equal(bits(out), bits(want)) and np.array_equal(bits(ram_b[1:n + 1]), bits(want))
        bad = nb.mailbox.encode_request(pos[:999])
        with pytest.raises(nb.NBodyError) as e:
            nb.mailbox.run(eng, bad, ram_b=ram_b)
        assert e.value.code == nb._lib.ERR_ARG


HOST_DONE_SCRIPT = r"""
import sys, glob, json, os
sys.path.insert(0, %r)
import numpy as np
import mini_nbody_amd as nb
fx = {}
for f in glob.glob(os.path.join(%r, "tests", "golden", "rtl_*.json")):
    d = json.load(open(f))
    w = lambda k: np.array([int(x, 16) for x in d[k]], np.uint32).view(np.float32).reshape(-1, 4)
    fx[d["n"]] = (w("pos0"), w("forces0"))
with nb.Mailbox(capacity=512, faithful=True) as mb:
    for served in (False, True):
        if served:
            mb.serve(True, 300000)
        for n in (9, 100, 40, 0, 9):
            mb.ram_b.view(np.uint32)[...] = 0xDEADBEEF
            mb.post(fx[n][0] if n else np.zeros((0, 4), np.float32))
            out, ticks = mb.wait() if served else mb.run(300000)
            rb = mb.ram_b.view(np.uint32)
            ok = ticks >= 1 and mb.ram_a[0, 0] == 0 and np.all(rb[0] == 0xDEADBEEF) and np.all(rb[n + 1:] == 0xDEADBEEF)
            ok = ok and (n == 0 or np.array_equal(out.view(np.uint32), fx[n][1].view(np.uint32)))
            print("served" if served else "called", n, "ok" if ok else "BAD", ticks)
        if served:
            mb.serve(False)
"""


def test_host_written_completion_is_still_a_working_form(nb):
    """NBODY_MAILBOX_DONE=host keeps round 5's completion — the host thread waits on the stream and writes word 0 itself, ticks from its
    own clock — for A/B (profiles/r06_mailbox_rate.txt) and as what serves the requests the device cannot complete.  Same bits, same map."""
    env = dict(os.environ, NBODY_MAILBOX_DONE="host")
    p = subprocess.run([sys.executable, "-c", HOST_DONE_SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 10 and all(l.split()[2] == "ok" for l in lines), lines
    # the host's ticks are its own latency (>= 3 at 300 MHz for >= 7 us), the device's at N = 9 are 2-3: the two forms are told apart by the tick word
    assert all(int(l.split()[3]) >= 2 for l in lines if l.split()[1] == "9"), lines
