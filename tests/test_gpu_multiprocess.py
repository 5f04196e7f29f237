"""Process-per-GPU flow on hardware: two (three) OS processes, each with its own HIP context on this box's one GPU,
rendezvous over gloo, nbody_init_rank(), positions exchanged every step through the host-staged transport
(nbody_set_host_gather) — i.e. everything the 8-GPU job does except that the slices travel through host memory
instead of RCCL/xGMI.  Result must equal one process configured with the same segmentation, bit for bit."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch
    nb = importlib.import_module("mini-nbody_amd")
    D = importlib.import_module("mini-nbody_amd.distributed")
    rank, world, local = D.init_process_group("gloo")
    n, steps = {n}, {steps}
    eng = D.make_engine(n, transport="host")
    eng.set_option(nb.OPT_JSUB, {jsub})
    eng.set_option(nb.OPT_OVERLAP, {overlap})
    pos, vel = nb.make_bodies(n, seed=33)
    f = eng.forces(pos)                      # every process gets all N force words (the other ranks' rows are gathered)
    eng.upload(pos, vel)
    eng.step(0.01, steps)
    p, v = eng.download()
    cfg = eng.config
    assert cfg["nranks"] == world and cfg["rank"] == rank
    if rank == world - 1:
        np.save({out!r} + "_force.npy", f)
    if rank == 0:
        np.save({out!r} + "_pos.npy", p); np.save({out!r} + "_vel.npy", v)
    eng.close()
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,overlap", [(2, 1), (3, 0), (3, 2)])
def test_two_processes_one_gpu_host_transport(nb, tmp_path, world, overlap):
    n, steps, jsub = 12000 + 7, 3, 2
    out = str(tmp_path / "mp")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n=n, steps=steps, jsub=jsub, overlap=overlap, out=out))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NBODY_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    gp, gv = np.load(out + "_pos.npy"), np.load(out + "_vel.npy")
    pos, vel = nb.make_bodies(n, seed=33)
    one = nb.NBody(n)
    try:
        one.set_option(nb.OPT_JSUB, jsub)
        one.set_option(nb.OPT_JSLICES, world)
        wf = one.forces(pos)
        one.upload(pos, vel)
        one.step(0.01, steps)
        wp, wv = one.download()
    finally:
        one.close()
    assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    assert np.array_equal(gv.view(np.uint32), wv.view(np.uint32))
    assert np.array_equal(np.load(out + "_force.npy").view(np.uint32), wf.view(np.uint32))
