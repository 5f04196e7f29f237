"""The reference's memory-mapped mailbox as data.

RAM A: word 0 = control {bit 0 BEGIN, bits 46:32 NUM_PTS}        S/top_level.vhd:184-185
       words 1..N = {x, y, z, ignored}, 16 bytes each             S/top_level.vhd:206-208, 238-240
RAM B: word k = {Fx, Fy, Fz, 0} of body k (the index the body has in RAM A); word 0 is never written
                                                                  S/compute_store.vhd:213, 221-242 (tests/test_fpga_store_model.py)
done:  word 0 of RAM A <- {ticks in bits 63:32}, BEGIN reads 0    S/top_level.vhd:146, 255-263
       one tick = 1000 clocks                                     S/top_level.vhd:121-144
max N = ram_depth - 1 = 32767                                     S/top_level.vhd:45
"""
import ctypes as C

import numpy as np

from . import _lib as L

MAX_POINTS = 32767
WORD = 16


def encode_request(pos):
    """(N+1) x 4 uint32 image of RAM A with BEGIN set (N = 0: the control word alone)."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 4)
    n = len(pos)
    if not 0 <= n <= MAX_POINTS:
        raise ValueError("NUM_PTS is a 15-bit field: 0..%d bodies" % MAX_POINTS)
    ram = np.zeros((n + 1, 4), np.uint32)
    ram[0, 0] = 1          # BEGIN
    ram[0, 1] = n          # bits [46:32]
    ram[1:] = pos.view(np.uint32)
    return ram


def decode_control(ram_a):
    w = np.asarray(ram_a, np.uint32).reshape(-1, 4)[0]
    return dict(begin=int(w[0] & 1), num_pts=int(w[1] & 0x7FFF), ticks=int(w[1]))


def run(engine, ram_a, clock_khz=0, ram_b=None):
    """Execute one request in place on the open context (`engine`: an NBody or a Mailbox) and rewrite word 0 of ram_a.  `ram_b`: the
    caller's own RAM B image of at least N + 1 words, of which only words 1..N are written (a fresh one if None).  Returns the forces
    of bodies 1..N = words 1..N of RAM B, as a view."""
    ram_a = np.asarray(ram_a)
    assert ram_a.dtype == np.uint32 and ram_a.flags.c_contiguous
    n = decode_control(ram_a)["num_pts"]
    if ram_b is None:
        ram_b = np.zeros((n + 1, 4), np.float32)
    assert ram_b.dtype == np.float32 and ram_b.flags.c_contiguous and len(ram_b) >= n + 1
    L.check(engine.lib.nbody_mailbox_run(ram_a.ctypes.data_as(C.c_void_p), ram_b.ctypes.data_as(C.c_void_p), int(clock_khz)))
    return ram_b[1:n + 1]


class Mailbox:
    """The PL block as the PS sees it (nbody_mailbox_open): ONE context serving requests of any NUM_PTS in 0..capacity, NUM_PTS sampled
    with every BEGIN (S/top_level.vhd:180-186).  faithful=True: the RTL's own rounding points and summation order, bit for bit
    (granted only after the device has proved the strict 1/sqrt).  `ram_a` / `ram_b` are the context's own RAM images — pinned host
    memory the device reads and writes directly — as numpy views: (capacity + 1, 4) uint32 and (capacity + 1, 4) float32 (word 0 of RAM B
    exists and is never written)."""

    def __init__(self, capacity=MAX_POINTS, faithful=True):
        self.lib = L.load()
        L.check(self.lib.nbody_mailbox_open(int(capacity), 1 if faithful else 0))
        self._open = True
        a, b, cap = C.c_void_p(), C.c_void_p(), C.c_int()
        L.check(self.lib.nbody_mailbox_rams(C.byref(a), C.byref(b), C.byref(cap)))
        self.capacity, self.faithful, self._posted = cap.value, bool(faithful), 0
        self.ram_a = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint32)), shape=(self.capacity + 1, 4))
        self.ram_b = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_float)), shape=(self.capacity + 1, 4))

    def post(self, pos):
        """What the PS does before it raises BEGIN: bodies into words 1..N of RAM A, NUM_PTS and BEGIN into word 0."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 4)
        n = len(pos)
        if n > self.capacity:
            raise ValueError("%d bodies exceed the mailbox's capacity %d" % (n, self.capacity))
        self.ram_a[1:n + 1] = pos.view(np.uint32)
        self._posted = n
        self.ram_a[0, 1:] = (n, 0, 0)
        self.ram_a[0, 0] = 1                  # BEGIN is written last (a served mailbox may take the request at once)
        return n

    def run(self, clock_khz=0):
        """One request on the context's own RAMs (no host copy): returns (view of RAM B words 1..N = the forces of bodies 1..N, ticks)."""
        n = int(self.ram_a[0, 1] & 0x7FFF)
        L.check(self.lib.nbody_mailbox_run(self.ram_a.ctypes.data_as(C.c_void_p), self.ram_b.ctypes.data_as(C.c_void_p), int(clock_khz)))
        return self.ram_b[1:n + 1], int(self.ram_a[0, 1])

    def forces(self, pos, clock_khz=0):
        """post + run; the result is a copy."""
        self.post(pos)
        out, _ = self.run(clock_khz)
        return out.copy()

    # ---- the mailbox without a call per request (nbody_mailbox_serve): a library thread plays the PL block's FSM ----
    def serve(self, on=True, clock_khz=0):
        """Start (or stop) the service thread: from then on a request is `post()` followed by polling word 0 — memory only."""
        L.check(self.lib.nbody_mailbox_serve(1 if on else 0, int(clock_khz)))
        self._serving = bool(on)

    def wait(self, timeout=10.0):
        """What the PS does after raising BEGIN: poll word 0 of RAM A until BEGIN reads 0.  Returns (RAM B words 1..N (view), ticks);
        raises if the library flagged the request (bits 127:96 of word 0, which the RTL always writes as 0)."""
        import time
        n = self._posted                       # (word 0 may already hold the tick count: NUM_PTS is what post() wrote)
        w0 = self.ram_a[0]
        t0 = time.perf_counter()
        while w0[0] & 1:
            if time.perf_counter() - t0 > timeout:
                raise TimeoutError("the mailbox did not complete within %.1f s" % timeout)
        if int(w0[3]):
            raise L.NBodyError(int(w0[3]), "mailbox request refused: " + self.lib.nbody_error_string(int(w0[3])).decode())
        return self.ram_b[1:n + 1], int(w0[1])

    def served(self):
        v = C.c_longlong()
        L.check(self.lib.nbody_get_info(L.INFO_MAILBOX_SERVED, C.byref(v)))
        return v.value

    def close(self):
        if self._open:
            self.ram_a = self.ram_b = None
            self.lib.nbody_shutdown()
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
