"""The reference's memory-mapped mailbox as data.

RAM A: word 0 = control {bit 0 BEGIN, bits 46:32 NUM_PTS}        S/top_level.vhd:184-185
       words 1..N = {x, y, z, ignored}, 16 bytes each             S/top_level.vhd:206-208, 238-240
RAM B: word k-1 = {Fx, Fy, Fz, 0} of body k                      S/compute_store.vhd:213, 227-242
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
    """(N+1) x 4 uint32 image of RAM A with BEGIN set."""
    pos = np.ascontiguousarray(pos, np.float32)
    n = len(pos)
    if not 0 < n <= MAX_POINTS:
        raise ValueError("NUM_PTS is a 15-bit field: 1..%d bodies" % MAX_POINTS)
    ram = np.zeros((n + 1, 4), np.uint32)
    ram[0, 0] = 1          # BEGIN
    ram[0, 1] = n          # bits [46:32]
    ram[1:] = pos.view(np.uint32)
    return ram


def decode_control(ram_a):
    w = np.asarray(ram_a, np.uint32).reshape(-1, 4)[0]
    return dict(begin=int(w[0] & 1), num_pts=int(w[1] & 0x7FFF), ticks=int(w[1]))


def run(engine, ram_a, clock_khz=0):
    """Execute one request in place: returns RAM B (N x 4 float32) and rewrites word 0 of ram_a."""
    ram_a = np.asarray(ram_a)
    assert ram_a.dtype == np.uint32 and ram_a.flags.c_contiguous
    n = decode_control(ram_a)["num_pts"]
    ram_b = np.zeros((n, 4), np.float32)
    L.check(engine.lib.nbody_mailbox_run(ram_a.ctypes.data_as(C.c_void_p), ram_b.ctypes.data_as(C.c_void_p), int(clock_khz)))
    return ram_b
