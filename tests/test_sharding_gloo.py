"""The N > 1 path on CPU: world_size-2 (and 3) gloo jobs run the host-side decomposition
(mini_nbody_amd/sharding.py — slices, ring exchange, ascending combine, kick, drift) with the oracle as
the per-segment force function (tests may; the product's force function is the HIP kernel) and real
torch.distributed send/recv for the ring, and must reproduce the single-process result bit for bit."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import oracle as O
    import mini_nbody_amd as nb
    S = nb.sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = O.Oracle(fast=False)
    n, sub, steps, dt = {n}, {sub}, {steps}, np.float32(0.01)
    pos, vel = nb.make_bodies(n, seed=5)
    f0, f1 = S.slice_bounds(rank, n, world)
    buf = [pos.copy(), np.zeros_like(pos)]          # double-buffered full position set
    my_vel = vel[f0:f1].copy()
    cur = 0
    have_all = True
    for step in range(steps):
        src = buf[cur]
        if not have_all:                             # ring all-gather of the other slices into buf[cur]
            for s, snd, rcv in S.ring_schedule(rank, world):
                a0, a1 = S.slice_bounds(snd, n, world); b0, b1 = S.slice_bounds(rcv, n, world)
                out = torch.from_numpy(src[a0:a1].copy()); inc = torch.empty((b1 - b0, 4), dtype=torch.float32)
                reqs = [dist.isend(out, (rank + 1) % world), dist.irecv(inc, (rank - 1) % world)]
                [r.wait() for r in reqs]
                src[b0:b1] = inc.numpy()
        # per-segment force function: the engine's default order inside a segment (blocks of {block} sources, two levels)
        acc = S.sharded_forces(rank, world, src, sub, lambda rows, s_: ora.forces_order(rows, s_, summ=O.SUM_BLOCKED, block={block}))
        kick = (np.float64(dt) * acc[:, :3].astype(np.float64) + my_vel[:, :3]).astype(np.float32)
        my_vel[:, :3] = kick
        nxt = buf[cur ^ 1]
        nxt[f0:f1, :3] = (my_vel[:, :3].astype(np.float64) * np.float64(dt) + src[f0:f1, :3]).astype(np.float32)
        nxt[f0:f1, 3] = src[f0:f1, 3]
        cur ^= 1
        have_all = world == 1
    np.save({out!r} + "_pos%d.npy" % rank, buf[cur][f0:f1]); np.save({out!r} + "_vel%d.npy" % rank, my_vel)
    dist.barrier(); dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def single_process(nb, oracle, n, P, sub, steps, block):
    """the oracle's own statement of the engine's order (ref_step_f32_order): P slices x sub pieces, blocked sums"""
    import oracle as O
    pos, vel = nb.make_bodies(n, seed=5)
    oracle.step_order(pos, vel, 0.01, steps, summ=O.SUM_BLOCKED, block=block, nslices=P, sub=sub)
    return pos, vel


@pytest.mark.parametrize("world,n,sub,block", [(2, 301, 2, 64), (3, 200, 1, 1024), (2, 1500, 3, 128)])
def test_gloo_ring_matches_single_process(nb, oracle, tmp_path, world, n, sub, block):
    steps = 3
    out = str(tmp_path / "r")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n=n, sub=sub, steps=steps, out=out, block=block))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    want_p, want_v = single_process(nb, oracle, n, world, sub, steps, block)
    got_p = np.concatenate([np.load(out + "_pos%d.npy" % r) for r in range(world)])
    got_v = np.concatenate([np.load(out + "_vel%d.npy" % r) for r in range(world)])
    assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))
    assert np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32))
