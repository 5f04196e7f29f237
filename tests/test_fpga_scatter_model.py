"""Rows a7/a8 of SURVEY.md §8: which partial sum lands in which slot of `results`, checked by simulating the control logic
that IS in the reference tree, cycle by cycle — S/fxyz.vhd:129-184 (FLUSH_CNT, FLUSH_ACTV, the feedback mux, SCTTR_CNT,
VALID_FMA_PREV, SCTTR_ACTV, the results latch), with the fma IP (absent from the tree) modelled as what its component
declaration and the design's own constant say it is: a pipeline of fma_latency = 16 stages with a valid bit
(S/fxyz.vhd:51-63, S/top_level.vhd:40), carrying SYMBOLIC values (the tuple of item indices summed so far).

What the oracle (oracle/nbody_ref.c fpga16_f32) and the kernel (force_fpga16_f32) assume, and this proves from the RTL:
  * partial k accumulates the items j = k (mod 16) in ascending order, each chain starting from 0.0 (S/fxyz.vhd:129-145);
  * when a stream of n items ends, results(t) latches the fma output of item n - 16 + t, i.e. the FINAL value of partial
    (n + t) mod 16, and 0 where no such item exists (S/fxyz.vhd:147-184);
  * a following stream starts from zero again.
Then the same model with NUMBERS (every fma evaluated by the oracle's own pair + accumulate on one source) followed by the
pairwise tree over results(0..15) (S/compute_store.vhd:157-168 feeds results(16 d .. 16 d + 15) as buff(0..15);
S/final_adder.vhd:88-104 adds leaves 2J, 2J+1) must give the oracle's REF_SUM_FPGA16 forces bit for bit."""
import numpy as np
import pytest

import oracle as O

from rtl_model import FMA_LATENCY, ZERO, FxyzControl, expected_slot     # the cycle model of S/fxyz.vhd:129-184 (tests/rtl_model.py)


@pytest.mark.parametrize("n", list(range(1, 41)) + [47, 48, 49, 100, 257])
def test_results_slot_t_holds_partial_n_plus_t_mod_16(n):
    got = FxyzControl().run_stream(n)
    for t in range(FMA_LATENCY):
        assert got[t] == expected_slot(n, t), (n, t)
    # every item is in exactly one slot, each slot's items are one residue class in ascending order
    items = sorted(j for slot in got for j in slot)
    assert items == list(range(n))
    for t, slot in enumerate(got):
        if slot:
            assert len({j % FMA_LATENCY for j in slot}) == 1 and list(slot) == sorted(slot)
            assert slot[-1] % FMA_LATENCY == (n + t) % FMA_LATENCY      # results(t) = partial[(n + t) mod 16]


def test_a_second_stream_restarts_from_zero():
    """block-group after block-group (S/top_level.vhd:188-196): the flush counter re-arms while VALID_FMA is low, so the
    next stream's first 16 items again take c = 0.0 and nothing of the previous stream is summed in"""
    m = FxyzControl()
    m.run_stream(37)
    got = m.run_stream(21, first_item=1000)
    for t in range(FMA_LATENCY):
        assert got[t] == expected_slot(21, t, first_item=1000)


@pytest.mark.parametrize("latency", [4, 8, 16])
def test_holds_for_other_fma_latencies(latency):
    """fma_latency is a generic (S/fxyz.vhd:35): the number of partial sums is the pipeline depth"""
    for n in (1, latency - 1, latency, latency + 1, 3 * latency + 2):
        got = FxyzControl(latency).run_stream(n)
        assert got == [expected_slot(n, t, latency) for t in range(latency)]


@pytest.mark.parametrize("n", [1, 5, 15, 16, 17, 31, 32, 33, 40, 100])
def test_numeric_model_plus_tree_equals_the_oracle_fpga16_order(nb, oracle, n):
    """the cycle model with real arithmetic — fma(d, inv3, c) evaluated by the oracle's own sequential kernel on ONE source
    with c as the starting accumulator — then the adder tree over results(0..15): the oracle's REF_SUM_FPGA16, bit for bit"""
    pos, _ = nb.make_bodies(n + 3, seed=50 + n)
    row, src = pos[:1], pos[3:]

    def fma_num(item, c):
        acc_in = np.zeros((1, 4), np.float32) if c is ZERO else c
        return oracle.forces_f32(row, src[item:item + 1], acc_in=acc_in, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_SEQ)

    slots = FxyzControl(fma=fma_num).run_stream(n)
    axis = np.zeros((3, FMA_LATENCY), np.float32)
    for t, v in enumerate(slots):
        if v is not ZERO:
            axis[:, t] = v[0, :3]
    want = oracle.forces_f32(row, src, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
    got = np.array([oracle.tree16(axis[d]) for d in range(3)], np.float32)     # S/final_adder.vhd:88-104 on results(16 d ..)
    assert np.array_equal(got.view(np.uint32), want[0, :3].view(np.uint32)), n
    assert want[0, 3] == 0                                                      # S/compute_store.vhd:242


@pytest.mark.parametrize("n", [1, 5, 15, 16, 17, 31, 40, 100, 257])
def test_sixteen_waves_decomposition_equals_the_oracle_fpga16_order(nb, oracle, n):
    """What force_fpga16w_f32 does (round 4), restated on the CPU: "wave" k runs ONE sequential fma chain from zero over the sources
    k, k + 16, k + 32, ... (the oracle's plain sequential kernel on that subset), "wave 0" takes results(t) = chain[(n + t) mod 16] — zero
    where n - 16 + t < 0 — and adds final_adder's tree.  Bit for bit the oracle's REF_SUM_FPGA16 over the whole stream, for every row:
    the sixteen partial sums of the reference are independent chains, which is all the kernel relies on."""
    pos, _ = nb.make_bodies(n, seed=70 + n)
    for d2 in (O.D2_REFERENCE, O.D2_FMA3):
        chains = np.zeros((16, n, 4), np.float32)
        for k in range(min(16, n)):
            chains[k] = oracle.forces_f32(pos, pos[k::16], d2=d2, rsqrt=O.RSQRT_F64, summ=O.SUM_SEQ)
        got = np.zeros((n, 4), np.float32)
        for i in range(n):
            for axis in range(3):
                leaves = np.array([chains[(n + t) % 16, i, axis] if n - 16 + t >= 0 else 0.0 for t in range(16)], np.float32)
                got[i, axis] = oracle.tree16(leaves)
        want = oracle.forces_f32(pos, d2=d2, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, d2)
