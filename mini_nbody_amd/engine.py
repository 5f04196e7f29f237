"""Host-side mirror of the path's interface: bodyForce()/integrate() on a
{pos, vel} structure of arrays, the device-resident step loop and the
force-only (mailbox) entry point — thin Python over the C-ABI of
include/nbody.h.  All compute happens in libnbody_hip.so (HIP, gfx950).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _as(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("expected an (n, 4) array of %s words" % np.dtype(dtype).name)
    return a


class NBody:
    """One context per process (the reference handles one request at a time, S/top_level.vhd:180-186)."""

    def __init__(self, n, fp64=False, tile=0, ngpus=1, rank=None, nranks=None, uid=None):
        self.lib = L.load()
        self.n, self.fp64 = int(n), bool(fp64)
        self.dtype = np.float64 if fp64 else np.float32
        if rank is None:
            L.check(self.lib.nbody_init(self.n, int(ngpus), int(self.fp64), int(tile)))
        else:
            buf = C.create_string_buffer(bytes(uid), 128) if uid is not None else None
            L.check(self.lib.nbody_init_rank(self.n, int(self.fp64), int(tile), int(rank), int(nranks), buf))
        self._open = True

    # ---- options / info ----
    def set_option(self, key, value):
        """(A strict binary32 arithmetic is granted by the library only after every device of the context has proved its 1/sqrt —
        nbody_strict_proof in include/nbody.h — and refused with ERR_UNSUPPORTED otherwise.)"""
        L.check(self.lib.nbody_set_option(int(key), int(value)))

    def info(self, key):
        v = C.c_longlong()
        L.check(self.lib.nbody_get_info(int(key), C.byref(v)))
        return v.value

    @property
    def config(self):
        names = {L.VARIANT_SMEM: "smem", L.VARIANT_LDS: "lds", L.VARIANT_READLANE: "readlane", L.VARIANT_ISA: "isa"}
        sums = {L.SUM_SEQ: "seq", L.SUM_FPGA16: "fpga16", L.SUM_BLOCKED: "blocked"}
        return dict(variant=names.get(self.info(L.INFO_VARIANT), "?"), iblock=self.info(L.INFO_IBLOCK),
                    jsub=self.info(L.INFO_JSUB), nseg=self.info(L.INFO_NSEG), tile=self.info(L.INFO_TILE),
                    sum_order=sums.get(self.info(L.INFO_SUM_ORDER), "?"), sum_block=self.info(L.INFO_SUM_BLOCK),
                    launches_per_step=self.info(L.INFO_LAUNCHES_PER_STEP), wsplit=self.info(L.INFO_WSPLIT),
                    isa_phase=self.info(L.INFO_ISA_PHASE), long_buffers=self.info(L.INFO_LONG_BUFFERS),
                    xcd_map=self.info(L.INFO_XCD_MAP), fuse=self.info(L.INFO_FUSE_COMBINE),
                    n_local=self.info(L.INFO_N_LOCAL), first_body=self.info(L.INFO_FIRST_BODY),
                    rank=self.info(L.INFO_RANK), nranks=self.info(L.INFO_NRANKS))

    # ---- state ----
    def _bs(self, pos, vel):
        if self.fp64:
            return L.BodySystemD(pos.ctypes.data_as(C.POINTER(C.c_double)), vel.ctypes.data_as(C.POINTER(C.c_double)))
        return L.BodySystem(pos.ctypes.data_as(C.POINTER(C.c_float)), vel.ctypes.data_as(C.POINTER(C.c_float)))

    def upload(self, pos, vel):
        pos, vel = _as(pos, self.dtype), _as(vel, self.dtype)
        if len(pos) != self.n or len(vel) != self.n:
            raise ValueError("upload expects the full %d bodies" % self.n)
        bs = self._bs(pos, vel)
        L.check((self.lib.nbody_upload_d if self.fp64 else self.lib.nbody_upload)(C.byref(bs)))

    def download(self):
        pos = np.empty((self.n, 4), self.dtype)
        vel = np.empty((self.n, 4), self.dtype)
        bs = self._bs(pos, vel)
        L.check((self.lib.nbody_download_d if self.fp64 else self.lib.nbody_download)(C.byref(bs)))
        return pos, vel

    def download_slice(self):
        """This rank's own bodies only (n_local words each), no collective."""
        cnt = self.info(L.INFO_N_LOCAL)
        pos = np.empty((cnt, 4), self.dtype)
        vel = np.empty((cnt, 4), self.dtype)
        L.check(self.lib.nbody_download_slice(pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p)))
        return pos, vel

    # ---- the path: same names and argument meaning as the C entry points ----
    def bodyForce(self, pos, vel, dt):
        """v += dt * F(pos), in place on vel; pos is read-only."""
        p, v = _as(pos, self.dtype), _as(vel, self.dtype)
        ct = C.c_double if self.fp64 else C.c_float
        fn = self.lib.bodyForce_d if self.fp64 else self.lib.bodyForce
        L.check(fn(p.ctypes.data_as(C.POINTER(ct)), v.ctypes.data_as(C.POINTER(ct)), dt, len(p)))
        if v is not vel:
            vel[...] = v
        return vel

    def integrate(self, pos, vel, dt):
        """r += v * dt, in place on pos."""
        p, v = _as(pos, self.dtype), _as(vel, self.dtype)
        ct = C.c_double if self.fp64 else C.c_float
        fn = self.lib.integrate_d if self.fp64 else self.lib.integrate
        L.check(fn(p.ctypes.data_as(C.POINTER(ct)), v.ctypes.data_as(C.POINTER(ct)), dt, len(p)))
        if p is not pos:
            pos[...] = p
        return pos

    def step(self, dt, nsteps=1):
        """nsteps x {bodyForce; integrate} on the uploaded state; asynchronous."""
        L.check((self.lib.nbody_step_d if self.fp64 else self.lib.nbody_step)(dt, int(nsteps)))

    def sync(self):
        L.check(self.lib.nbody_sync())

    def forces(self, pos):
        """{Fx, Fy, Fz, 0} per body from {x, y, z, .} words (the reference's RAM images)."""
        p = _as(pos, self.dtype)
        out = np.empty_like(p)
        ct = C.c_double if self.fp64 else C.c_float
        fn = self.lib.nbody_forces_d if self.fp64 else self.lib.nbody_forces
        L.check(fn(p.ctypes.data_as(C.POINTER(ct)), out.ctypes.data_as(C.POINTER(ct)), len(p)))
        return out

    def forces_rows(self, first_row, n_rows):
        """Forces on a row sample (rows of this rank's slice) from the state on the device."""
        out = np.empty((n_rows, 4), self.dtype)
        if self.fp64:
            L.check(self.lib.nbody_forces_rows_d(int(first_row), int(n_rows), out.ctypes.data_as(C.POINTER(C.c_double))))
        else:
            L.check(self.lib.nbody_forces_rows(int(first_row), int(n_rows), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def comm_selftest(self):
        """Push a patterned array through the RCCL calls of the multi-GPU path (all-gather + one ring step); returns
        the bytes this rank received.  Needs the communicator of NBody(..., rank=, nranks=, uid=)."""
        moved = C.c_longlong()
        L.check(self.lib.nbody_comm_selftest(C.byref(moved)))
        return moved.value

    def comm_selftest_virtual(self, vranks, form):
        """The transfer plans of `vranks` virtual ranks run through real ncclSend/ncclRecv on a one-rank communicator."""
        moved = C.c_longlong()
        L.check(self.lib.nbody_comm_selftest_virtual(int(vranks), int(form), C.byref(moved)))
        return moved.value

    def comm_probe(self, nbytes, when):
        """(ms of one RCCL ring step of nbytes from enqueue to done, ms of the force pass beside it); when = 0 alone,
        1 enqueued just before a full force pass, 2 just after it."""
        c, f = C.c_double(), C.c_double()
        L.check(self.lib.nbody_comm_probe(int(nbytes), int(when), C.byref(c), C.byref(f)))
        return c.value, f.value

    def comm_time(self, reset=False):
        """(ms the compute stream waited for arriving slices, number of waits) since the last reset (OPT_TIMING = 1)."""
        ms, cnt = C.c_double(), C.c_longlong()
        L.check(self.lib.nbody_comm_time(C.byref(ms), C.byref(cnt), int(reset)))
        return ms.value, cnt.value

    @property
    def order(self):
        """The summation order of the current configuration as keyword arguments of the oracle's order()
        (tests mirror the engine's order with it)."""
        cfg = self.config
        nsl = cfg["nseg"] // cfg["jsub"]
        return dict(nslices=nsl, sub=cfg["jsub"], block=cfg["sum_block"] or 1024,
                    summ={"seq": 0, "fpga16": 1, "blocked": 2}[cfg["sum_order"]], wsplit=cfg["wsplit"])

    def kernel_time(self, reset=False):
        ms, cnt = C.c_double(), C.c_longlong()
        L.check(self.lib.nbody_kernel_time(C.byref(ms), C.byref(cnt), int(reset)))
        return ms.value, cnt.value

    def device_ptr(self, which):
        p, b = C.c_void_p(), C.c_size_t()
        L.check(self.lib.nbody_device_ptr(int(which), C.byref(p), C.byref(b)))
        return p.value, b.value

    def set_host_gather(self, fn):
        """Multi-process transport through host memory: fn(host_ptr, n_total, word_bytes, rank, nranks) -> 0 must
        all-gather the array in place (see nbody_set_host_gather in include/nbody.h)."""
        def tramp(user, host_ptr, n_total, word_bytes, rank, nranks):
            try:
                return int(fn(host_ptr, n_total, word_bytes, rank, nranks) or 0)
            except Exception:   # an exception must not unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._gather_cb = L.HOST_GATHER_FN(tramp)      # keep the thunk alive as long as the engine
        L.check(self.lib.nbody_set_host_gather(self._gather_cb, None))

    def close(self):
        if self._open:
            self.lib.nbody_shutdown()
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def comm_plan(form, rank, nranks, n):
    """The transfer plan of one rank as a list of dicts (pure host arithmetic in the library: no GPU needed)."""
    lib = L.load()
    cnt = C.c_int()
    L.check(lib.nbody_comm_plan(int(form), int(rank), int(nranks), int(n), None, 0, C.byref(cnt)))
    buf = (C.c_longlong * (7 * max(1, cnt.value)))()
    L.check(lib.nbody_comm_plan(int(form), int(rank), int(nranks), int(n), buf, cnt.value, C.byref(cnt)))
    keys = ("group", "send_peer", "send_first", "send_count", "recv_peer", "recv_first", "recv_count")
    return [dict(zip(keys, buf[7 * k:7 * k + 7])) for k in range(cnt.value)]


def strict_proof():
    """The library's gate for the strict binary32 arithmetic (nbody_strict_proof): every positive normal binary32 through the
    eight-operation 1/sqrt and through its IEEE definition on every device of the open context (else on the device a one-GPU context
    would take), cached per device.  Returns (mismatches, first mismatching pattern) — (0, 0) when proved."""
    bad, first = C.c_ulonglong(), C.c_uint()
    rc = L.load().nbody_strict_proof(C.byref(bad), C.byref(first))
    if rc not in (0, L.ERR_UNSUPPORTED):
        L.check(rc)
    return bad.value, first.value


def rsqrt_selftest(first_bits=0, count=1 << 32):
    """The strict 1/sqrt against its definition for `count` binary32 bit patterns from first_bits (default: every float), on the
    device: (mismatches — must be 0, patterns evaluated by the IEEE form, smallest mismatching pattern)."""
    bad, slow, first = C.c_ulonglong(), C.c_ulonglong(), C.c_uint()
    L.check(L.load().nbody_rsqrt_selftest(int(first_bits), int(count), C.byref(bad), C.byref(slow), C.byref(first)))
    return bad.value, slow.value, first.value


def rsqrt_strict(x, ieee_only=False):
    """y = the strict 1/sqrt of the float32 array x, as the force kernels evaluate it (or by the IEEE expression alone)."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    L.check(L.load().nbody_rsqrt_strict(x.ctypes.data_as(fp), y.ctypes.data_as(fp), int(x.size), 1 if ieee_only else 0))
    return y


def unique_id():
    buf = C.create_string_buffer(128)
    L.check(L.load().nbody_unique_id(buf))
    return bytes(buf.raw)
