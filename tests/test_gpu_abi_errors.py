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
        ram_a[0, 0], ram_a[0, 1] = 1, 77                           # NUM_PTS != n of the context
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
