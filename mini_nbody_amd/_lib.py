"""ctypes binding of the C-ABI in include/nbody.h (libnbody_hip.so).

There is no fallback: if the shared library is missing or no GPU is usable the
calls raise.  Nothing here imports or links oracle/.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_LIB selects another build of the same C-ABI (the diagnostic library of `make diag`); there is still no fallback
LIB_PATH = os.environ.get("NBODY_LIB") or os.path.join(HERE, "libnbody_hip.so")

# mirrors of the enums in include/nbody.h
(OPT_VARIANT, OPT_IBLOCK, OPT_JSUB, OPT_JSLICES, OPT_ARITH, OPT_SUM_ORDER, OPT_TIMING, OPT_COMM, OPT_OVERLAP, OPT_ISA_PHASE,
 OPT_WAVES_PER_SIMD, OPT_GRAPH, OPT_SUM_BLOCK, OPT_FUSE_COMBINE, OPT_ISA_LONG_BUFFERS, OPT_XCD_MAP, OPT_WSPLIT) = range(1, 18)
VARIANT_AUTO, VARIANT_SMEM, VARIANT_LDS, VARIANT_READLANE, VARIANT_ISA = range(5)
ARITH_FMA3, ARITH_REFERENCE, ARITH_STRICT, ARITH_REFERENCE_STRICT = 0, 1, 2, 3
SUM_SEQ, SUM_FPGA16, SUM_BLOCKED = 0, 1, 2
COMM_RING, COMM_ALLGATHER, COMM_AUTO, COMM_DIRECT = 0, 1, 2, 3
(INFO_N, INFO_N_LOCAL, INFO_FIRST_BODY, INFO_RANK, INFO_NRANKS, INFO_VARIANT, INFO_IBLOCK, INFO_JSUB, INFO_NSEG,
 INFO_DEVICE, INFO_CU_COUNT, INFO_CLOCK_KHZ, INFO_FP64, INFO_TILE, INFO_STEPS_DONE, INFO_SUM_ORDER, INFO_SUM_BLOCK,
 INFO_LAUNCHES_PER_STEP, INFO_HAS_COMM, INFO_WSPLIT, INFO_ISA_PHASE, INFO_LONG_BUFFERS, INFO_XCD_MAP, INFO_FUSE_COMBINE,
 INFO_COMM_FORM, INFO_COMM_PRIORITY, INFO_DIAG_BUILD, INFO_MAILBOX_SERVED, INFO_MAILBOX_SERVING) = range(1, 30)

ERR_NOT_INIT, ERR_ARG, ERR_NO_DEVICE, ERR_RCCL_LOAD, ERR_STATE, ERR_UNSUPPORTED = 1001, 1002, 1003, 1004, 1005, 1006

# every symbol include/nbody.h declares (tests/test_abi.py checks the library exports exactly these)
SYMBOLS = [
    "nbody_init", "nbody_unique_id", "nbody_init_rank", "nbody_shutdown", "nbody_set_option", "nbody_get_info",
    "nbody_error_string", "nbody_upload", "nbody_download", "nbody_upload_d", "nbody_download_d", "bodyForce",
    "integrate", "bodyForce_d", "integrate_d", "nbody_step", "nbody_step_d", "nbody_sync", "nbody_forces",
    "nbody_forces_d", "nbody_forces_rows", "nbody_mailbox_run", "nbody_kernel_time", "nbody_device_ptr",
    "nbody_set_host_gather", "nbody_download_slice", "nbody_comm_selftest", "nbody_forces_rows_d",
    "nbody_comm_selftest_virtual", "nbody_comm_plan", "nbody_comm_probe", "nbody_comm_time",
    "nbody_rsqrt_selftest", "nbody_rsqrt_strict", "nbody_strict_proof", "nbody_mailbox_open", "nbody_mailbox_rams",
    "nbody_mailbox_serve",
]


HOST_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)


class BodySystem(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_float)), ("vel", C.POINTER(C.c_float))]


class BodySystemD(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_double)), ("vel", C.POINTER(C.c_double))]


class NBodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (code %d)" % (msg, code))
        self.code = code


_lib = None


def load():
    """Load libnbody_hip.so (built by `make lib` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `make lib` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    fp, dp, vp, i, f, d = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p, C.c_int, C.c_float, C.c_double
    sig = {
        "nbody_init": [i, i, i, i], "nbody_unique_id": [vp], "nbody_init_rank": [i, i, i, i, i, vp],
        "nbody_set_option": [i, i], "nbody_get_info": [i, C.POINTER(C.c_longlong)],
        "nbody_upload": [C.POINTER(BodySystem)], "nbody_download": [C.POINTER(BodySystem)],
        "nbody_upload_d": [C.POINTER(BodySystemD)], "nbody_download_d": [C.POINTER(BodySystemD)],
        "bodyForce": [fp, fp, f, i], "integrate": [fp, fp, f, i], "bodyForce_d": [dp, dp, d, i],
        "integrate_d": [dp, dp, d, i], "nbody_step": [f, i], "nbody_step_d": [d, i], "nbody_sync": [],
        "nbody_forces": [fp, fp, i], "nbody_forces_d": [dp, dp, i], "nbody_forces_rows": [i, i, fp], "nbody_forces_rows_d": [i, i, dp],
        "nbody_comm_selftest": [C.POINTER(C.c_longlong)],
        "nbody_comm_selftest_virtual": [i, i, C.POINTER(C.c_longlong)],
        "nbody_comm_plan": [i, i, i, i, C.POINTER(C.c_longlong), i, C.POINTER(i)],
        "nbody_comm_probe": [C.c_longlong, i, C.POINTER(d), C.POINTER(d)],
        "nbody_comm_time": [C.POINTER(d), C.POINTER(C.c_longlong), i],
        "nbody_mailbox_run": [vp, vp, i], "nbody_kernel_time": [C.POINTER(d), C.POINTER(C.c_longlong), i],
        "nbody_device_ptr": [i, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "nbody_set_host_gather": [HOST_GATHER_FN, vp], "nbody_download_slice": [vp, vp],
        "nbody_rsqrt_selftest": [C.c_uint, C.c_ulonglong, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)],
        "nbody_rsqrt_strict": [fp, fp, i, i],
        "nbody_strict_proof": [C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)],
        "nbody_mailbox_open": [i, i], "nbody_mailbox_rams": [C.POINTER(vp), C.POINTER(vp), C.POINTER(i)],
        "nbody_mailbox_serve": [i, i],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.nbody_shutdown.argtypes = []
    L.nbody_shutdown.restype = None
    L.nbody_error_string.argtypes = [i]
    L.nbody_error_string.restype = C.c_char_p
    _lib = L
    return L


def check(code):
    if code != 0:
        raise NBodyError(code, load().nbody_error_string(code).decode())
