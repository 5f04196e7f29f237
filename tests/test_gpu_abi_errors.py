"""Error behaviour of the C-ABI on a live context (the reference reports no errors at all, S/top_level.vhd:50; the
conventions are include/nbody.h's): wrong sizes, wrong precision, wrong state — and that a failed call leaves the
context usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_error_codes_and_recovery(nb):
    L = nb._lib
    lib = L.load()
    lib.nbody_shutdown()
    fp = C.POINTER(C.c_float)
    buf = np.zeros((128, 4), np.float32)
    p = buf.ctypes.data_as(fp)
    assert lib.nbody_step(0.01, 1) == L.ERR_NOT_INIT
    assert lib.nbody_init(0, 1, 0, 0) == L.ERR_ARG                 # n <= 0
    assert lib.nbody_init(128, 0, 0, 0) == L.ERR_ARG               # ngpus <= 0
    assert lib.nbody_init(128, 1, 0, 100) == L.ERR_ARG             # tile not a multiple of 64
    assert lib.nbody_init(128, 99, 0, 0) in (L.ERR_ARG, L.ERR_NO_DEVICE)
    assert lib.nbody_init(128, 1, 0, 0) == 0
    try:
        assert lib.bodyForce(p, p, 0.01, 64) == L.ERR_ARG          # n differs from nbody_init's
        assert lib.nbody_forces(p, p, 127) == L.ERR_ARG
        assert lib.nbody_upload(None) == L.ERR_ARG
        assert lib.nbody_step(0.01, -1) == L.ERR_ARG
        assert lib.nbody_step_d(0.01, 1) == L.ERR_STATE            # fp64 call on an fp32 context
        assert lib.bodyForce_d(None, None, 0.01, 128) == L.ERR_STATE
        assert lib.nbody_forces_rows(0, 129, p) == L.ERR_ARG
        assert lib.nbody_forces_rows_d(0, 1, buf.ctypes.data_as(C.POINTER(C.c_double))) == L.ERR_STATE   # fp64 call, fp32 context
        assert lib.nbody_comm_selftest(None) == L.ERR_STATE        # not a multi-process context: no communicator
        assert lib.nbody_set_option(nb.OPT_SUM_BLOCK, 100) == L.ERR_ARG   # not a multiple of 64
        assert lib.nbody_set_option(nb.OPT_SUM_ORDER, 3) == L.ERR_ARG
        assert lib.nbody_set_option(nb.OPT_OVERLAP, 3) == L.ERR_ARG
        assert lib.nbody_set_option(nb.OPT_VARIANT, 99) == L.ERR_ARG
        ram_a = np.zeros((129, 4), np.uint32)                      # BEGIN not set
        assert lib.nbody_mailbox_run(ram_a.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), 0) == L.ERR_STATE
        ram_a[0, 0], ram_a[0, 1] = 1, 129                          # NUM_PTS beyond the context's capacity (its n)
        assert lib.nbody_mailbox_run(ram_a.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), 0) == L.ERR_ARG
        assert lib.nbody_set_host_gather(L.HOST_GATHER_FN(lambda *a: 0), None) == L.ERR_STATE   # not a multi-process context
        # still usable
        pos, vel = nb.make_bodies(128)
        out = np.empty_like(pos)
        assert lib.nbody_forces(pos.ctypes.data_as(fp), out.ctypes.data_as(fp), 128) == 0
        assert np.isfinite(out).all() and np.abs(out).max() > 0
        assert b"bad argument" in lib.nbody_error_string(L.ERR_ARG)
        v = C.c_longlong()
        assert lib.nbody_get_info(L.INFO_N, C.byref(v)) == 0 and v.value == 128
        assert lib.nbody_get_info(999, C.byref(v)) == L.ERR_ARG
    finally:
        lib.nbody_shutdown()
    assert lib.nbody_sync() == L.ERR_NOT_INIT
    lib.nbody_shutdown()                                           # idempotent


def test_a_failed_step_does_not_poison_the_next_one(nb):
    """A multi-rank step that fails after its own-slice launch has run (here: the host transport's exchange callback fails
    once) leaves the arrival counters of the in-launch combine part-counted; the library must re-zero them, or the next step
    would combine before all partial sums are written.  Rank 0 of a two-rank job in this one process, the other rank's
    positions supplied by the callback from a one-GPU run with the same segmentation: after the failure, the retried step
    gives that run's bits."""
    import ctypes as C
    n, dt, jsub = 20000, 0.01, 4
    pos, vel = nb.make_bodies(n, seed=15)
    ref = nb.NBody(n)
    ref.set_option(nb.OPT_JSLICES, 2)
    ref.set_option(nb.OPT_JSUB, jsub)
    ref.set_option(nb.OPT_WSPLIT, 4)
    ref.set_option(nb.OPT_FUSE_COMBINE, 1)
    ref.upload(pos, vel)
    ref.step(dt, 1)
    p1, _ = ref.download()
    ref.step(dt, 1)
    p2, v2 = ref.download()
    ref.close()
    state = {"fail": False, "calls": 0}

    def gather(host_ptr, n_total, word_bytes, rank, nranks):
        state["calls"] += 1
        if state["fail"]:
            return 1
        buf = np.ctypeslib.as_array((C.c_float * (n_total * 4)).from_address(host_ptr)).reshape(n_total, 4)
        buf[n_total // 2:] = p1[n_total // 2:]          # rank 1's slice of the positions after step 1
        return 0

    eng = nb.NBody(n, rank=0, nranks=2, uid=None)
    try:
        eng.set_host_gather(gather)
        eng.set_option(nb.OPT_JSUB, jsub)
        eng.set_option(nb.OPT_WSPLIT, 4)
        eng.set_option(nb.OPT_FUSE_COMBINE, 1)
        eng.set_option(nb.OPT_OVERLAP, 1)
        assert eng.config["launches_per_step"] == 2 and eng.config["nseg"] == 2 * jsub
        eng.upload(pos, vel)
        eng.step(dt, 1)                  # every slice present after the upload: no exchange yet
        eng.sync()
        state["fail"] = True
        with pytest.raises(nb.NBodyError):
            eng.step(dt, 1)              # own-slice launch enqueued, then the exchange fails
        eng.sync()
        assert state["calls"] == 1
        state["fail"] = False
        eng.step(dt, 1)                  # the same step again
        eng.sync()
        gp, gv = eng.download_slice()
        half = n // 2
        assert np.array_equal(gp.view(np.uint32), p2[:half].view(np.uint32))
        assert np.array_equal(gv.view(np.uint32), v2[:half].view(np.uint32))
    finally:
        eng.close()


def test_occupancy_cap_that_cannot_exist_is_refused_not_underflowed(nb):
    """NBODY_OPT_WAVES_PER_SIMD caps the occupancy by giving each workgroup 160 KiB / k of dynamic LDS minus what the kernel holds
    itself.  The 16-wave fp64 kernel's join already holds 30 KiB: with k = 7 that share is 22.9 KiB and the subtraction used to
    wrap around into a 160-KiB request and an opaque HIP launch error.  Now: NBODY_ERR_ARG, the context stays usable,
    and a cap that does exist changes no bit."""
    n = 3000
    pos, vel = nb.make_bodies(n, seed=4, dtype=np.float64)
    eng = nb.NBody(n, fp64=True)
    try:
        eng.set_option(nb.OPT_WSPLIT, 16)
        want = eng.forces(pos)
        eng.set_option(nb.OPT_WAVES_PER_SIMD, 7)
        with pytest.raises(nb.NBodyError) as e:
            eng.forces(pos)
        assert e.value.code == nb._lib.ERR_ARG
        eng.set_option(nb.OPT_WAVES_PER_SIMD, 2)          # 80 KiB per workgroup: one 16-wave workgroup per CU
        assert np.array_equal(eng.forces(pos).view(np.uint64), want.view(np.uint64))
        eng.set_option(nb.OPT_WAVES_PER_SIMD, 0)
        eng.upload(pos, vel)
        eng.step(0.01, 3)                                  # and the step loop is intact after the refused launch
        assert np.isfinite(eng.download()[0]).all()
    finally:
        eng.close()
