"""mini_nbody_amd.distributed.autotune_comm on CPU: a world-size-2 gloo job with a stub engine whose steps take a time that
depends on the transfer form AND the rank.  What must hold: the slowest rank's time decides, every rank makes the same choice
(the forms are different RCCL call sequences — a disagreement would deadlock the real job), the library's default is kept unless
another form is faster by more than the margin, ragged slice lengths drop the all-gather form, and the engine is left configured
with the choice."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import mini_nbody_amd as nb
    import mini_nbody_amd.distributed as D
    L = nb._lib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cost = {cost!r}          # {{form name: [seconds per step on rank 0, on rank 1]}}
    names = {{L.COMM_ALLGATHER: "allgather", L.COMM_DIRECT: "direct", L.COMM_RING: "ring"}}

    class Stub:
        n = {n}
        def __init__(self): self.opt = {{}}; self.log = []
        def set_option(self, k, v): self.opt[k] = v
        def upload(self, pos, vel): self.log.append(("upload", pos, vel))
        def step(self, dt, k):
            form = names[self.opt[L.OPT_COMM]]
            if {fail!r} == [form, rank]:
                raise RuntimeError("RCCL error 5 in form %s" % form)
            self.log.append((form, self.opt[L.OPT_OVERLAP], k))
            time.sleep(cost[form][rank] * k)
        def sync(self): pass

    e = Stub()
    best, ms = D.autotune_comm(e, 0.01, steps=2, margin={margin}, restore={restore!r})
    json.dump({{"best": best, "ms": ms, "left": [names[e.opt[L.OPT_COMM]], e.opt[L.OPT_OVERLAP]], "log": e.log}}, open({out!r} + "%d.json" % rank, "w"))
    dist.barrier(); dist.destroy_process_group()
""")


def run(tmp_path, cost, n=1000, margin=0.01, fail=None, restore=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r")
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, cost=cost, n=n, margin=margin, out=out, fail=fail, restore=restore))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-2000:]
    return [json.load(open(out + "%d.json" % r)) for r in range(2)]


def test_the_slowest_rank_decides_and_all_ranks_agree(tmp_path):
    # rank 0 alone would pick "direct" (0.02 s), rank 1 alone "ring" (0.03): by the MAX over ranks allgather 0.20, direct 0.12, ring 0.06
    a, b = run(tmp_path, {"allgather": [0.05, 0.20], "direct": [0.02, 0.12], "ring": [0.06, 0.03]})
    assert a["best"] == b["best"] == "ring" and a["left"] == b["left"] == ["ring", 2]
    assert a["ms"] == b["ms"] and a["ms"]["ring"] < a["ms"]["direct"] < a["ms"]["allgather"]
    # one untimed + two timed steps per form, in the same order on both ranks
    assert [x[:2] for x in a["log"]] == [x[:2] for x in b["log"]] == [["allgather", 1]] * 2 + [["direct", 1]] * 2 + [["ring", 2]] * 2
    assert [x[2] for x in a["log"]] == [1, 2] * 3


def test_ties_keep_the_default_and_ragged_slices_drop_the_all_gather(tmp_path):
    a, b = run(tmp_path, {"allgather": [0.050, 0.050], "direct": [0.048, 0.048], "ring": [0.049, 0.049]}, margin=0.25)
    assert a["best"] == b["best"] == "allgather" and a["left"] == ["allgather", 1]          # 4 % faster is inside the margin
    a, b = run(tmp_path, {"allgather": [0.001, 0.001], "direct": [0.05, 0.05], "ring": [0.02, 0.02]}, n=1001)
    assert "allgather" not in a["ms"] and a["best"] == b["best"] == "ring"                  # 1001 bodies over 2 ranks: no equal slices


def test_a_form_that_fails_on_one_rank_is_dropped_by_all_and_the_state_is_put_back(tmp_path):
    """the "direct" form raises on rank 1 only; rank 0 must not go on to time it alone (it would wait in the barrier for a rank
    that left) — both drop it, agree on the best of the rest, and both put the caller's state back after the candidates' steps."""
    a, b = run(tmp_path, {"allgather": [0.05, 0.06], "direct": [0.001, 0.001], "ring": [0.02, 0.03]}, fail=["direct", 1], restore=("POS", "VEL"))
    assert a["best"] == b["best"] == "ring" and a["ms"]["direct"] == b["ms"]["direct"] == float("inf")
    assert a["left"] == b["left"] == ["ring", 2]
    for r in (a, b):
        assert r["log"][-1] == ["upload", "POS", "VEL"]
        assert [x[0] for x in r["log"][:-1]].count("direct") == (1 if r is a else 0)     # rank 0 ran the untimed step only; rank 1 raised in it
