"""Deterministic initial conditions — numpy mirror of include/nbody_ic.h.

The reference has no host program and so no IC generator; this is
the build's own definition: SplitMix64, 3N position
components then 3N velocity components, uniform in [-1, 1), pos.w = 1,
vel.w = 0.  Counter-based, so any shard [first, first+count) can be produced
on its own and matches the C generator bit for bit (tests/test_host_logic.py).
"""
import numpy as np

DEFAULT_SEED = 42
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, k):
    """k-th output (k = 0, 1, ...) of SplitMix64 started at `seed`; k is a uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(seed, k):
    m = (splitmix64(seed, k) >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (m.astype(np.float32) * np.float32(1.0 / 8388608.0)).astype(np.float32)


def make_bodies(n, seed=DEFAULT_SEED, first=0, count=None, dtype=np.float32):
    """(pos, vel), each (count, 4), for bodies [first, first+count) of an n-body system."""
    count = n - first if count is None else count
    i = np.arange(first, first + count, dtype=np.uint64)
    pos = np.empty((count, 4), dtype)
    vel = np.empty((count, 4), dtype)
    for c in range(3):
        pos[:, c] = uniform(seed, np.uint64(3) * i + np.uint64(c))
        vel[:, c] = uniform(seed, np.uint64(3 * n) + np.uint64(3) * i + np.uint64(c))
    pos[:, 3] = 1
    vel[:, 3] = 0
    return pos, vel
