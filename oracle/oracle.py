"""ctypes binding of the CPU oracle (oracle/nbody_ref.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

D2_REFERENCE, D2_FMA3 = 0, 1
RSQRT_F64, RSQRT_DIVSQRT = 0, 1
SUM_SEQ, SUM_FPGA16, SUM_BLOCKED = 0, 1, 2
DEFAULT_BLOCK = 1024

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_fp = C.POINTER(C.c_float)


class Order(C.Structure):
    """ref_order_t: the engine's summation order (segments of the sources, blocks inside a segment)."""
    _fields_ = [("d2_mode", C.c_int), ("rsqrt_mode", C.c_int), ("sum_mode", C.c_int), ("block", C.c_int),
                ("nslices", C.c_int), ("sub", C.c_int), ("wsplit", C.c_int)]


def order(d2=D2_FMA3, rsqrt=RSQRT_F64, summ=SUM_BLOCKED, block=DEFAULT_BLOCK, nslices=1, sub=1, wsplit=1):
    return Order(d2, rsqrt, summ, block, nslices, sub, wsplit)


def build(force=False):
    if force or not all(os.path.exists(os.path.join(HERE, f)) for f in ("libnbody_ref.so", "libnbody_ref_fast.so", "nbody_cpu")):
        subprocess.run(["make", "-C", HERE] + (["-B"] if force else []), check=True, capture_output=True)


class Oracle:
    """fast=False: libnbody_ref.so (-O2, serial).  fast=True: libnbody_ref_fast.so
    (-O3, OpenMP); tests/test_oracle_kat.py checks that both give identical bits."""

    def __init__(self, fast=False, path=None):
        build()
        self.path = path or os.path.join(HERE, "libnbody_ref_fast.so" if fast else "libnbody_ref.so")
        L = self.lib = C.CDLL(self.path)
        f = C.c_float
        L.ref_soft.restype = f
        L.ref_dxy.restype = f
        L.ref_dxy.argtypes = [f, f, f, f, _fp, _fp]
        L.ref_dzsoft.restype = f
        L.ref_dzsoft.argtypes = [f, f, _fp]
        L.ref_dxyz_soft.restype = f
        L.ref_dxyz_soft.argtypes = [f] * 6 + [_fp] * 3
        L.ref_d2_fma3.restype = f
        L.ref_d2_fma3.argtypes = [f, f, f]
        L.ref_rsqrt.restype = f
        L.ref_rsqrt.argtypes = [f, C.c_int]
        L.ref_cube.restype = f
        L.ref_cube.argtypes = [f]
        L.ref_tree16.restype = f
        L.ref_tree16.argtypes = [_f32p]
        L.ref_forces_f32.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int]
        L.ref_forces_f64.argtypes = [_f64p, C.c_int, _f64p, C.c_int, _f64p]
        L.ref_forces_f64_from_f32.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f64p]
        L.ref_bodyForce_f32.argtypes = [_f32p, _f32p, f, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_integrate_f32.argtypes = [_f32p, _f32p, f, C.c_int]
        L.ref_bodyForce_f64.argtypes = [_f64p, _f64p, C.c_double, C.c_int]
        L.ref_integrate_f64.argtypes = [_f64p, _f64p, C.c_double, C.c_int]
        L.ref_step_f32.argtypes = [_f32p, _f32p, f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_step_f64.argtypes = [_f64p, _f64p, C.c_double, C.c_int, C.c_int]
        L.ref_forces_f32_order.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.POINTER(Order)]
        L.ref_step_f32_order.argtypes = [_f32p, _f32p, f, C.c_int, C.c_int, C.POINTER(Order)]
        L.ref_forces_f64_order.argtypes = [_f64p, C.c_int, _f64p, C.c_int, _f64p, C.POINTER(Order)]
        L.ref_step_f64_order.argtypes = [_f64p, _f64p, C.c_double, C.c_int, C.c_int, C.POINTER(Order)]
        L.ref_segment_bounds.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)] * 2
        L.ref_ic_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.ref_ic_f64.argtypes = [_f64p, _f64p, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.ref_num_threads.restype = C.c_int
        L.ref_set_num_threads.argtypes = [C.c_int]

        self._max_threads = None      # the thread count this object may use at most (None: whatever OpenMP would take)

    def _size_team(self, pairs):
        """Size the OpenMP team to the work of the next call: a pass over a few thousand pairs on a 256-thread host spends ~0.1 s waking
        its team and microseconds computing (the GPU suite makes thousands of such calls).  One thread per 250 000 pairs, at most the
        host's (or set_num_threads()'s) count.  The result does not depend on the team: rows are independent."""
        if self._max_threads is None:
            self._max_threads = max(1, self.lib.ref_num_threads())
        self.lib.ref_set_num_threads(int(max(1, min(self._max_threads, pairs // 250000))))

    # ---- stages ----
    def soft(self):
        return np.float32(self.lib.ref_soft())

    def dxy(self, x_this, x_target, y_this, y_target):
        dx, dy = C.c_float(), C.c_float()
        s = self.lib.ref_dxy(x_this, x_target, y_this, y_target, C.byref(dx), C.byref(dy))
        return np.float32(s), np.float32(dx.value), np.float32(dy.value)

    def dzsoft(self, z_this, z_target):
        dz = C.c_float()
        s = self.lib.ref_dzsoft(z_this, z_target, C.byref(dz))
        return np.float32(s), np.float32(dz.value)

    def dxyz_soft(self, this, target):
        dx, dy, dz = C.c_float(), C.c_float(), C.c_float()
        s = self.lib.ref_dxyz_soft(this[0], target[0], this[1], target[1], this[2], target[2],
                                   C.byref(dx), C.byref(dy), C.byref(dz))
        return np.float32(s), (np.float32(dx.value), np.float32(dy.value), np.float32(dz.value))

    def d2_fma3(self, dx, dy, dz):
        return np.float32(self.lib.ref_d2_fma3(dx, dy, dz))

    def rsqrt(self, d2, mode=RSQRT_F64):
        return np.float32(self.lib.ref_rsqrt(d2, mode))

    def cube(self, inv):
        return np.float32(self.lib.ref_cube(inv))

    def tree16(self, p):
        return np.float32(self.lib.ref_tree16(np.ascontiguousarray(p, np.float32)))

    # ---- passes ----
    def forces_f32(self, rows, src=None, acc_in=None, d2=D2_FMA3, rsqrt=RSQRT_F64, summ=SUM_SEQ):
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
        src = rows if src is None else np.ascontiguousarray(src, np.float32).reshape(-1, 4)
        acc = np.zeros_like(rows)
        if acc_in is not None:
            acc_in = np.ascontiguousarray(acc_in, np.float32)
            ptr = acc_in.ctypes.data_as(C.c_void_p)
        else:
            ptr = None
        self._size_team(len(rows) * len(src))
        self.lib.ref_forces_f32(rows, len(rows), src, len(src), ptr, acc, d2, rsqrt, summ)
        return acc

    def forces_order(self, rows, src=None, order_=None, **kw):
        """Forces in the engine's summation order: order_ = oracle.order(...) or keyword arguments of it."""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
        src = rows if src is None else np.ascontiguousarray(src, np.float32).reshape(-1, 4)
        o = order_ if order_ is not None else order(**kw)
        acc = np.zeros_like(rows)
        self._size_team(len(rows) * len(src))
        self.lib.ref_forces_f32_order(rows, len(rows), src, len(src), acc, C.byref(o))
        return acc

    def step_order(self, pos, vel, dt, nsteps, order_=None, **kw):
        o = order_ if order_ is not None else order(**kw)
        self._size_team(len(pos) * len(pos))
        self.lib.ref_step_f32_order(pos, vel, dt, len(pos), nsteps, C.byref(o))

    def forces_f64_order(self, rows, src=None, order_=None, **kw):
        """fp64 forces in the engine's order (segments x wave-split pieces, one sequential sum per piece): the strict fp64 mode's twin"""
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, 4)
        src = rows if src is None else np.ascontiguousarray(src, np.float64).reshape(-1, 4)
        o = order_ if order_ is not None else order(**kw)
        acc = np.zeros_like(rows)
        self._size_team(len(rows) * len(src))
        self.lib.ref_forces_f64_order(rows, len(rows), src, len(src), acc, C.byref(o))
        return acc

    def step_f64_order(self, pos, vel, dt, nsteps, order_=None, **kw):
        o = order_ if order_ is not None else order(**kw)
        self._size_team(len(pos) * len(pos))
        self.lib.ref_step_f64_order(pos, vel, dt, len(pos), nsteps, C.byref(o))

    def segment_bounds(self, q, t, n, nslices, sub):
        b, e = C.c_int(), C.c_int()
        self.lib.ref_segment_bounds(q, t, n, nslices, sub, C.byref(b), C.byref(e))
        return b.value, e.value

    def forces_f64(self, rows, src=None):
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, 4)
        src = rows if src is None else np.ascontiguousarray(src, np.float64).reshape(-1, 4)
        acc = np.zeros_like(rows)
        self._size_team(len(rows) * len(src))
        self.lib.ref_forces_f64(rows, len(rows), src, len(src), acc)
        return acc

    def forces_f64_from_f32(self, rows, src=None):
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
        src = rows if src is None else np.ascontiguousarray(src, np.float32).reshape(-1, 4)
        acc = np.zeros(rows.shape, np.float64)
        self._size_team(len(rows) * len(src))
        self.lib.ref_forces_f64_from_f32(rows, len(rows), src, len(src), acc)
        return acc

    def bodyForce(self, pos, vel, dt, d2=D2_FMA3, rsqrt=RSQRT_F64, summ=SUM_SEQ):
        """In place on vel (float32 or float64 arrays, n x 4)."""
        self._size_team(len(pos) * len(pos))
        if pos.dtype == np.float64:
            self.lib.ref_bodyForce_f64(pos, vel, dt, len(pos))
        else:
            self.lib.ref_bodyForce_f32(pos, vel, dt, len(pos), d2, rsqrt, summ)

    def integrate(self, pos, vel, dt):
        if pos.dtype == np.float64:
            self.lib.ref_integrate_f64(pos, vel, dt, len(pos))
        else:
            self.lib.ref_integrate_f32(pos, vel, dt, len(pos))

    def step(self, pos, vel, dt, nsteps, d2=D2_FMA3, rsqrt=RSQRT_F64, summ=SUM_SEQ):
        self._size_team(len(pos) * len(pos))
        if pos.dtype == np.float64:
            self.lib.ref_step_f64(pos, vel, dt, len(pos), nsteps)
        else:
            self.lib.ref_step_f32(pos, vel, dt, len(pos), nsteps, d2, rsqrt, summ)

    def ic(self, n, seed=42, first=0, count=None, dtype=np.float32):
        count = n - first if count is None else count
        pos = np.empty((count, 4), dtype)
        vel = np.empty((count, 4), dtype)
        (self.lib.ref_ic_f64 if dtype == np.float64 else self.lib.ref_ic_f32)(pos, vel, n, first, count, seed)
        return pos, vel

    def num_threads(self):
        """the most threads a call of this object will use"""
        if self._max_threads is None:
            self._max_threads = max(1, self.lib.ref_num_threads())
        return self._max_threads

    def set_num_threads(self, t):
        self._max_threads = max(1, int(t))
        self.lib.ref_set_num_threads(self._max_threads)
