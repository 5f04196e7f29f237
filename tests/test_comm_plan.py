"""The transfer plan the library executes for P > 1 ranks (nbody_comm_plan in include/nbody.h: the pure host function
rccl_gather() in csrc/comm.cpp walks), checked without a GPU for P = 2..8 and ragged N:
  * against the Python mirror of the schedule (mini_nbody_amd/sharding.py ring_schedule / direct_schedule),
  * every word of every other rank's slice is received exactly once, nothing lands in the rank's own slice,
  * pair s of rank r matches a pair of its peer in the same RCCL group: same word range, peer addressed back,
  * a ring rank only forwards what it already holds (its own slice, or a slice received in an earlier group).
The same plans are run through real ncclSend/ncclRecv on one GPU by nbody_comm_selftest_virtual (tests/test_gpu_parity.py)."""
import pytest

SIZES = [8, 64, 1000, 1001, 4099, 12007, 1 << 20, (1 << 20) + 5]


def plans(nb, form, P, n):
    return [nb.comm_plan(form, r, P, n) for r in range(P)]


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7, 8])
def test_plan_matches_the_host_mirror(nb, P):
    S = nb.sharding
    for n in SIZES:
        for r in range(P):
            ring = nb.comm_plan(nb.COMM_RING, r, P, n)
            assert len(ring) == P - 1
            for op, (s, q_send, q_recv) in zip(ring, S.ring_schedule(r, P)):
                assert op["group"] == s
                assert (op["send_peer"], op["recv_peer"]) == ((r + 1) % P, (r - 1) % P)
                assert (op["send_first"], op["send_first"] + op["send_count"]) == S.slice_bounds(q_send, n, P)
                assert (op["recv_first"], op["recv_first"] + op["recv_count"]) == S.slice_bounds(q_recv, n, P)
            direct = nb.comm_plan(nb.COMM_DIRECT, r, P, n)
            assert len(direct) == P - 1
            for op, (to, frm) in zip(direct, S.direct_schedule(r, P)):
                assert op["group"] == 1 and (op["send_peer"], op["recv_peer"]) == (to, frm)
                assert (op["send_first"], op["send_first"] + op["send_count"]) == S.slice_bounds(r, n, P)
                assert (op["recv_first"], op["recv_first"] + op["recv_count"]) == S.slice_bounds(frm, n, P)


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("form", ["ring", "direct"])
def test_every_word_received_once_and_pairs_match(nb, P, form):
    S = nb.sharding
    f = nb.COMM_RING if form == "ring" else nb.COMM_DIRECT
    for n in SIZES:
        all_plans = plans(nb, f, P, n)
        for r, plan in enumerate(all_plans):
            own = S.slice_bounds(r, n, P)
            got = sorted((op["recv_first"], op["recv_first"] + op["recv_count"]) for op in plan)
            # the received ranges tile [0, n) minus the own slice, each exactly once
            want = sorted(S.slice_bounds(q, n, P) for q in range(P) if q != r)
            assert got == want, (n, r)
            have = {own}
            for op in plan:
                rng = (op["send_first"], op["send_first"] + op["send_count"])
                assert rng in have, "rank %d sends words it does not hold yet in group %d" % (r, op["group"])
                # the receive is matched by exactly one send of the peer, in the same group, of the same words
                peer = all_plans[op["recv_peer"]]
                match = [q for q in peer if q["group"] == op["group"] and q["send_peer"] == r
                         and (q["send_first"], q["send_count"]) == (op["recv_first"], op["recv_count"])]
                assert len(match) == 1, (n, r, op)
                # and the send is matched by exactly one receive of its destination
                dest = all_plans[op["send_peer"]]
                match = [q for q in dest if q["group"] == op["group"] and q["recv_peer"] == r
                         and (q["recv_first"], q["recv_count"]) == (op["send_first"], op["send_count"])]
                assert len(match) == 1, (n, r, op)
                if form == "ring":      # what arrived in this group may be forwarded in the next one
                    have.add((op["recv_first"], op["recv_first"] + op["recv_count"]))
            # per peer and group, sends and receives come in equal numbers (RCCL matches them in order)
            for grp in {op["group"] for op in plan}:
                for peer in range(P):
                    ns = sum(1 for op in plan if op["group"] == grp and op["send_peer"] == peer)
                    nr = sum(1 for op in all_plans[peer] if op["group"] == grp and op["recv_peer"] == r)
                    assert ns == nr


def test_plan_argument_errors(nb):
    lib = nb._lib.load()
    import ctypes as C
    cnt = C.c_int()
    assert lib.nbody_comm_plan(nb.COMM_RING, 0, 1, 100, None, 0, C.byref(cnt)) == 0 and cnt.value == 0
    assert lib.nbody_comm_plan(nb.COMM_RING, 2, 2, 100, None, 0, C.byref(cnt)) == nb._lib.ERR_ARG      # rank out of range
    assert lib.nbody_comm_plan(nb.COMM_ALLGATHER, 0, 2, 100, None, 0, C.byref(cnt)) == nb._lib.ERR_ARG  # a collective has no plan
    assert lib.nbody_comm_plan(nb.COMM_RING, 0, 4, 3, None, 0, C.byref(cnt)) == nb._lib.ERR_ARG        # fewer bodies than ranks
    buf = (C.c_longlong * 7)()
    assert lib.nbody_comm_plan(nb.COMM_RING, 0, 4, 100, buf, 1, C.byref(cnt)) == nb._lib.ERR_ARG       # buffer too small
