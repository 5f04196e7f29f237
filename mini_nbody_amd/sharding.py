"""Host-side decomposition of the path — Python mirror of the logic in
csrc/nbody_args.hpp (slice_first, segment_bounds) and csrc/context.cpp
(ring order, ascending combine), so that it can be exercised on CPU
(tests/test_sharding_gloo.py) without a GPU.

Bodies are sharded by i (the reference processes i in independent blocks
against the full j set, S/top_level.vhd:187-254); sources are cut into one
slice per rank, each into `sub` pieces; the partial sums of the segments are
added in ascending source order.
"""
import numpy as np


def slice_first(q, n, nslices):
    base, rem = divmod(n, nslices)
    return q * base + min(q, rem)


def slice_bounds(q, n, nslices):
    return slice_first(q, n, nslices), slice_first(q + 1, n, nslices)


def segment_bounds(q, t, n, nslices, sub):
    f0, f1 = slice_bounds(q, n, nslices)
    piece = (f1 - f0 + sub - 1) // sub
    b = min(f0 + t * piece, f1)
    e = min(b + piece, f1)
    return b, e


def piece_bounds(jb, je, w, wsplit):
    """Piece w of `wsplit` of the segment [jb, je): what wave w of a workgroup walks (NBODY_OPT_WSPLIT = 4)."""
    piece = (je - jb + wsplit - 1) // wsplit
    b = min(jb + w * piece, je)
    return b, min(b + piece, je)


def ring_slice(rank, s, nranks):
    """Slice that reaches `rank` at ring step s (s = 0: its own)."""
    return (rank - s) % nranks


def ring_schedule(rank, nranks):
    """[(step, send_slice, recv_slice)] for s = 1..P-1: forward what arrived last, receive from prev."""
    return [(s, ring_slice(rank, s - 1, nranks), ring_slice(rank, s, nranks)) for s in range(1, nranks)]


def direct_schedule(rank, nranks):
    """[(send_to, recv_from)] of the fully connected form (NBODY_COMM_DIRECT): one group, the own slice to every peer,
    every peer's own slice back; pair s of rank r matches pair s of rank recv_from (who sends to r)."""
    return [((rank + s) % nranks, (rank - s) % nranks) for s in range(1, nranks)]


def combine_ascending(partials):
    """((p0 + p1) + p2) + ... in the dtype of the partials (combine_kernel in nbody_kernels.hpp)."""
    acc = partials[0].copy()
    for p in partials[1:]:
        acc = (acc + p).astype(partials[0].dtype)
    return acc


def sharded_forces(rank, nranks, pos, sub, force_fn, arrival=None):
    """Forces on rank's bodies computed segment by segment in `arrival` order of the slices
    (default: ring order, own slice first) and combined in ascending order.  force_fn(rows, src)
    returns the sequential-order partial force of `rows` against `src`."""
    n = len(pos)
    f0, f1 = slice_bounds(rank, n, nranks)
    rows = pos[f0:f1]
    order = arrival if arrival is not None else [ring_slice(rank, s, nranks) for s in range(nranks)]
    parts = {}
    for q in order:
        for t in range(sub):
            b, e = segment_bounds(q, t, n, nranks, sub)
            parts[q * sub + t] = force_fn(rows, pos[b:e]) if e > b else np.zeros_like(rows)
    return combine_ascending([parts[c] for c in sorted(parts)])
