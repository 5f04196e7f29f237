"""One process per GPU: rendezvous through torch.distributed (plumbing only), data path through RCCL
inside libnbody_hip.so.  torch is imported lazily so that single-GPU use does not need it."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Join the job described by RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT.  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_unique_id(make_uid):
    """Rank 0 creates the 128-byte RCCL id with make_uid(); everyone gets it."""
    import torch.distributed as dist
    rank, world, _ = env_rank()
    if world == 1:
        return None
    box = [make_uid() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def gloo_host_gather(host_ptr, n_total, word_bytes, rank, nranks):
    """All-gather of a sharded host array through torch.distributed (any backend that moves CPU tensors)."""
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    from .sharding import slice_bounds
    buf = np.ctypeslib.as_array((C.c_ubyte * (n_total * word_bytes)).from_address(host_ptr))
    maxlen = (-(-n_total // nranks)) * word_bytes
    f0, f1 = slice_bounds(rank, n_total, nranks)
    mine = torch.zeros(maxlen, dtype=torch.uint8)
    mine[:(f1 - f0) * word_bytes] = torch.from_numpy(buf[f0 * word_bytes:f1 * word_bytes].copy())
    outs = [torch.empty(maxlen, dtype=torch.uint8) for _ in range(nranks)]
    dist.all_gather(outs, mine)
    for q in range(nranks):
        if q == rank:
            continue
        b0, b1 = slice_bounds(q, n_total, nranks)
        buf[b0 * word_bytes:b1 * word_bytes] = outs[q][:(b1 - b0) * word_bytes].numpy()
    return 0


def make_engine(n, fp64=False, tile=0, transport="auto"):
    """NBody engine for this process: single GPU when WORLD_SIZE is 1, else rank `RANK` of the job.
    transport: "rccl" (positions travel GPU to GPU inside the library), "host" (staged through host memory and
    torch.distributed), "auto" (rccl, and if any rank cannot set it up, every rank falls back to host)."""
    import torch
    import torch.distributed as dist
    from .engine import NBody, unique_id
    from ._lib import NBodyError
    rank, world, _ = env_rank()
    if world == 1:
        return NBody(n, fp64=fp64, tile=tile)
    eng, err = None, ""
    if transport in ("auto", "rccl"):
        try:
            uid = broadcast_unique_id(unique_id)
            eng = NBody(n, fp64=fp64, tile=tile, rank=rank, nranks=world, uid=uid)
        except (NBodyError, OSError) as e:
            eng, err = None, str(e)
            if transport == "rccl":
                raise
        ok = torch.tensor([1 if eng is not None else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 1:
            eng.transport = "rccl"
            return eng
        if eng is not None:
            eng.close()
    eng = NBody(n, fp64=fp64, tile=tile, rank=rank, nranks=world, uid=None)
    eng.set_host_gather(gloo_host_gather)
    eng.transport = "host-staged via torch.distributed" + (" (RCCL unavailable: %s)" % err if err else "")
    return eng


# the transfer forms a multi-rank RCCL job can take for its per-step all-gather of positions, as (name, NBODY_OPT_COMM, NBODY_OPT_OVERLAP):
# the library's default first (ties keep it)
def comm_candidates(L):
    return [("allgather", L.COMM_ALLGATHER, 1), ("direct", L.COMM_DIRECT, 1), ("ring", L.COMM_RING, 2)]


def autotune_comm(eng, dt=0.01, steps=2, margin=0.01, candidates=None, set_option=None, restore=None):
    """Choose the transfer form by MEASUREMENT on the job's own hardware, during warm-up: every candidate runs one untimed and
    `steps` timed steps on the live engine, the slowest rank's time counts (all-reduce MAX over the control plane), and the
    fastest form is kept — the library's default unless another one is more than `margin` faster.  Every rank makes the same
    choice from the same all-reduced numbers (the forms are different RCCL call sequences: a disagreement would deadlock).
    The result of a step does not depend on the form, only its duration does.
    Why: which form hides best behind the own-slice kernel was decided on ONE GPU with a one-rank communicator
    (profiles/r03_comm_under_load.md); the first job on real xGMI links measures it instead of trusting that.
    The candidates advance the simulation by 1 + `steps` steps each: pass restore=(pos, vel) — the full arrays, as upload() takes them —
    to have the state put back afterwards.  A form whose options or warm-up step raise on ANY rank is dropped by every rank together
    (its entry reads inf).
    Returns (chosen name, {name: ms per step}); the engine is left configured with the choice."""
    import time
    import torch
    import torch.distributed as dist
    from . import _lib as L
    cands = candidates or comm_candidates(L)
    if candidates is None and eng.n % max(1, env_rank()[1]) != 0:
        # slices of different lengths: ncclAllGather does not apply (the library would run the ring for it); its default is DIRECT
        cands = [c for c in cands if c[0] != "allgather"]
    setopt = set_option or eng.set_option
    ms = {}

    def agreed_failure(failed):
        """MAX over the ranks of a failure flag: every rank drops a form together (a rank that went on alone would wait in the next
        barrier for the ones that raised)"""
        t = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t[0] > 0)

    for name, comm, overlap in cands:
        failed = False
        try:
            setopt(L.OPT_COMM, comm)
            setopt(L.OPT_OVERLAP, overlap)
            eng.step(dt, 1)
            eng.sync()
        except Exception:          # (a rank that raises INSIDE a collective leaves its peers in it: that case is the supervisor's deadline)
            failed = True
        if agreed_failure(failed):
            ms[name] = float("inf")
            continue
        dist.barrier()
        t0 = time.perf_counter()
        try:
            eng.step(dt, steps)
            eng.sync()
            el = time.perf_counter() - t0
        except Exception:
            el = float("inf")
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms[name] = float(t[0]) * 1e3 / steps
    ok = [c for c in cands if ms[c[0]] != float("inf")]
    if not ok:
        raise RuntimeError("no transfer form completed its warm-up steps: %r" % (ms,))
    best = ok[0][0]
    for name, _, _ in ok[1:]:
        if ms[name] < ms[best] * (1.0 - margin):
            best = name
    for name, comm, overlap in cands:
        if name == best:
            setopt(L.OPT_COMM, comm)
            setopt(L.OPT_OVERLAP, overlap)
    if restore is not None:
        # the candidates ran (1 + steps) steps each on the live engine: put the caller's state back, so that an autotuned run enters its
        # timed region from the same state as a default one
        eng.upload(*restore)
    return best, ms
