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


def make_engine(n, fp64=False, tile=0):
    """NBody engine for this process: single GPU when WORLD_SIZE is 1, else rank `RANK` of the job."""
    from .engine import NBody, unique_id
    rank, world, _ = env_rank()
    if world == 1:
        return NBody(n, fp64=fp64, tile=tile)
    uid = broadcast_unique_id(unique_id)
    return NBody(n, fp64=fp64, tile=tile, rank=rank, nranks=world, uid=uid)
