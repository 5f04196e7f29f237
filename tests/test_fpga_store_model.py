"""Row b of SURVEY.md §8 (the drop-in boundary), RAM B's address map: WHERE the force of body k is written, established by
simulating the store side of the reference cycle by cycle — S/compute_store.vhd:117-242 (tests/rtl_model.py ComputeStore).

The point (Missing 1): the process at :221-238 registers `write_we` and `STORE_PTR <= STORE_PTR + 1` on the SAME edge, and
:241 forms `write_addr` combinationally from STORE_PTR.  When the RAM samples we = 1 — the edge after — the address is already the
incremented one.  So the first lane of the first block-group lands at word 1, not word 0:
    force of body k  ->  RAM B word k   (the index body k has in RAM A),   word 0 is never written,
    masked lanes (index > NUM_PTS) are skipped but counted (:227-233),   RESET_STORE brings the pointer back to 0 (:224-226).
(The name ZERO_PTR, :76-77, suggests the author may have meant k - 1; the RTL does not do it.  The library follows the RTL.)"""
import pytest

from rtl_model import ComputeStore, expected_slot, ZERO

NB = 12


def run_group(cs, first_word, n_targets, mask, first_target=1):
    """one block-group as the sequencer drives it (S/top_level.vhd:233-254): n targets on consecutive clocks for the twelve resident
    bodies first_word .. first_word + 11, then idle until STORE_BUSY falls (S/top_level.vhd:193)"""
    this = [first_word + b for b in range(NB)]
    for j in range(n_targets):
        cs.edge(1, this, first_target + j, mask, 0)
    seen_busy = 0
    for _ in range(2000):
        cs.edge(0, this, None, mask, 0)
        seen_busy |= cs.store_busy
        if seen_busy and not cs.store_busy:
            return
    raise AssertionError("STORE_BUSY never fell")


def check_word(cs, word, this_word, n_targets, first_target=1):
    din = cs.ram_b[word]
    lane = (word - 1) % NB
    for dim in range(3):
        block, d, slots = din[dim]
        assert (block, d) == (lane, dim), (word, din[dim][:2])                       # {Fx, Fy, Fz} of ONE lane, in that order: :213
        for t, slot in enumerate(slots):
            want = expected_slot(n_targets, t, first_item=first_target)
            assert tuple(tgt for tgt, _ in slot) == want, (word, dim, t)               # results(t) = partial[(n + t) mod 16]
            assert all(snap[lane] == this_word for _, snap in slot), (word, dim, t)   # summed for THIS body and no other


@pytest.mark.parametrize("n", [16, 17, 40])
def test_first_block_group_lands_at_words_1_to_12_and_word_0_is_never_written(n):
    cs = ComputeStore()
    run_group(cs, 1, n, [1] * NB)
    assert [w for _, w, _ in cs.writes] == list(range(1, 13))
    assert 0 not in cs.ram_b
    for k in range(1, 13):
        check_word(cs, k, k, n)
    # twelve writes on every third clock (x, y, z gathered per lane: :204-218), the first 48 clocks after SCATTER_COMPLETE
    cyc = [c for c, _, _ in cs.writes]
    assert [b - a for a, b in zip(cyc, cyc[1:])] == [3] * 11


def test_two_block_groups_and_a_tail_mask():
    """N = 17: the second group holds bodies 13..24 of which 13..17 exist.  Its lanes write words 13..17; the masked lanes are skipped
    BUT COUNTED, so a third group (had there been one) would start at word 25"""
    n = 17
    cs = ComputeStore()
    run_group(cs, 1, n, [1] * NB)
    run_group(cs, 13, n, [1 if 13 + b <= n else 0 for b in range(NB)])
    assert [w for _, w, _ in cs.writes] == list(range(1, 18))
    assert cs.store_ptr == 24                                                      # :233 counts every lane, masked or not
    for k in range(1, 18):
        check_word(cs, k, k, n)
    assert 0 not in cs.ram_b and all(w <= n for w in cs.ram_b)
    run_group(cs, 25, n, [1] * NB)                                                 # (no such group exists at N = 17; it shows the counter)
    assert [w for _, w, _ in cs.writes][17:] == list(range(25, 37))


def test_reset_store_returns_the_pointer_to_zero():
    """`complete` pulses RESET_STORE (S/top_level.vhd:255-263): the next request's forces start at word 1 again"""
    cs = ComputeStore()
    run_group(cs, 1, 20, [1] * NB)
    assert cs.store_ptr == 12
    cs.edge(0, [0] * NB, None, [1] * NB, 1)
    assert cs.store_ptr == 0 and cs.write_we == 0
    n0 = len(cs.writes)
    run_group(cs, 1, 33, [1] * 5 + [0] * 7)
    assert [w for _, w, _ in cs.writes][n0:] == [1, 2, 3, 4, 5]
    check_word(cs, 5, 5, 33)


def test_a_stream_shorter_than_the_fma_pipeline_leaves_zero_slots():
    """fewer than 16 targets: the slots whose item never existed are written as 0.0 (S/fxyz.vhd:177-181) and the tree adds them"""
    cs = ComputeStore()
    run_group(cs, 1, 5, [1] * NB)
    _, _, slots = cs.ram_b[1][0]
    assert sum(1 for s in slots if s is ZERO) == 11 and sorted(t for s in slots for t, _ in s) == [1, 2, 3, 4, 5]


@pytest.mark.parametrize("pipe_depth", [40, 83, 120])
def test_the_address_map_does_not_depend_on_the_arithmetic_pipelines_depth(pipe_depth):
    """the latency from valid_in to VALID_FMA (83 clocks with the constants of S/top_level.vhd:37-42) moves WHEN things happen, not where"""
    cs = ComputeStore(pipe_depth=pipe_depth)
    run_group(cs, 1, 30, [1] * NB)
    run_group(cs, 13, 30, [1] * 6 + [0] * 6)
    assert [w for _, w, _ in cs.writes] == list(range(1, 19))


def test_data_and_address_of_a_write_belong_to_the_same_lane():
    """WRITE_INT_DIN is complete (z latched) at the very edge that raises write_we and increments STORE_PTR; the RAM samples all three one
    edge later, before lane b + 1's x overwrites bits 31:0 on that same edge — a registered signal is read as it was (:204-218, 221-242)"""
    cs = ComputeStore()
    run_group(cs, 100, 16, [1] * NB)
    for k in range(1, 13):
        lanes = {cs.ram_b[k][d][0] for d in range(3)}
        assert lanes == {k - 1}
