"""Rows a10 / b of SURVEY.md §8: the reference's sequencer (S/top_level.vhd:121-146, 165-280) on top of compute_store, simulated cycle by
cycle (tests/rtl_model.py TopLevel) — the protocol the mailbox of include/nbody.h stands in for, AND the four places where the RTL
as written does not implement that protocol (Missing 2; INTEGRATION.md "Departures from the RTL as written"):
  (i)   NUM_PTS <= uram_latency (3): TRGT_VALID never rises, nothing is streamed, RAM B is never written         :42, 234-254
  (ii)  THIS_PTR / TRGT_PTR have no initial value: the first request after power-up takes its bodies one word low   :56, 58
  (iii) `waiting` polls the word THIS_PTR points at (= word 1 after the first pass), not word 0                    :192, 276
  (iv)  BEGIN_SIGNAL is assigned only in `waiting`: still '1' on return from `complete`, the FSM starts a second,
        spurious pass whose NUM_PTS is the tick count just written, and whose completion writes ticks = 0           :184, 265, 138
and two more this model found:
  (v)   THIS_PTR is log_ram_depth bits wide: a request whose last block-group reaches lane index 2^15 wraps it, `THIS_PTR > NUM_PTS`
        never becomes true and the pass never ends — NUM_PTS >= 32761 at the RTL's ram_depth                        :45-46, 189, 194
  (vi)  block_setup tests `THIS_PTR > NUM_PTS` BEFORE it tests STORE_BUSY: `complete` — ticks, BEGIN cleared, RESET_STORE — happens
        while the LAST block-group is still in the arithmetic pipeline; its forces are stored 145-170 clocks after word 0 was rewritten, and at
        words 1..12 (STORE_PTR was just reset), on top of the first block-group's                                   :189-193, 224-225
With four one-line repairs ("ptr_init", "poll_word0", "clear_begin", "drain_first") the same RTL runs the protocol the header describes: bodies at
words 1..N of RAM A, forces at words 1..N of RAM B (word 0 never written), every body summed over targets 1..N in ascending order with
the sixteen partial sums latched in rotation, one completion, ticks >= 1, BEGIN cleared."""
import pytest

from rtl_model import TopLevel, expected_slot

ALL_FIXES = ("ptr_init", "poll_word0", "clear_begin", "drain_first")


def ram_image(n_words=64, num_pts=0, begin=0):
    """bodies whose control-looking fields are harmless when (iii) makes the FSM read one as a control word: field 0 even (BEGIN = 0)"""
    ram = [(2 * k, 0x10000 * k, 0, 0) for k in range(n_words)]
    ram[0] = (begin, num_pts, 0, 0)
    return ram


def lanes_of(top, word):
    """(this word, [targets...] per results slot) of the force written at RAM B `word`"""
    din = top.cs.ram_b[word]
    lane = din[0][0]
    slots = din[0][2]
    this = {snap[lane] for s in slots for _, snap in s}
    return lane, this, [tuple(t for t, _ in s) for s in slots]


@pytest.mark.parametrize("n", [4, 9, 12, 13, 24, 40, 100])
def test_the_intended_protocol_with_the_four_repairs(n):
    top = TopLevel(ram_image(n + 20, n, 1), fixes=ALL_FIXES)
    cycles = top.run()
    assert cycles is not None and len(top.events("done")) == 1
    # RAM B: exactly the words 1..N, word 0 untouched, nothing beyond N (the masked lanes)
    assert sorted(top.cs.ram_b) == list(range(1, n + 1))
    groups = (n + 11) // 12
    assert top.streamed == [list(range(1, n + 1))] * groups                         # every group streams ALL targets 1..N, ascending: :233-254
    for k in range(1, n + 1):
        lane, this, slots = lanes_of(top, k)
        assert lane == (k - 1) % 12 and this == {k}                                 # the force at word k is body k's
        assert slots == [expected_slot(n, t, first_item=1) for t in range(16)]      # ... over targets 1..N, rotated: S/fxyz.vhd:147-184
    # completion: word 0 <- {ticks in bits 63:32, 0 elsewhere}; BEGIN reads 0                                    :146, 255-263
    assert top.ram_a[0][0] == 0 and top.ram_a[0][2:] == (0, 0)
    assert top.ram_a[0][1] == 1 + (cycles // 1000)                                   # one tick per 1000 clocks, 1 at BEGIN's rising edge: :121-144
    assert top.cs.store_ptr == 0                                                     # RESET_STORE: the next request starts at word 1 again
    # the schedule SURVEY.md §6 derives: N + ~250 clocks per block-group
    assert groups * (n + 150) < cycles < groups * (n + 300) + 50


def test_request_after_request_on_one_power_up():
    """NUM_PTS is sampled with every BEGIN (:180-186): 40, then 9, then 13 on the same machine; words beyond the new N keep the old pass's
    forces (nothing clears RAM B)"""
    top = TopLevel(ram_image(64, 40, 1), fixes=ALL_FIXES)
    assert top.run() is not None
    for n in (9, 13):
        before = dict(top.cs.ram_b)
        top.post(n)
        assert top.run() is not None
        for k in range(1, n + 1):
            assert lanes_of(top, k)[1] == {k} and lanes_of(top, k)[2] == [expected_slot(n, t, first_item=1) for t in range(16)]
        assert all(top.cs.ram_b[k] == before[k] for k in before if k > n)
    assert len(top.events("done")) == 3


def test_num_pts_zero_completes_at_once():
    top = TopLevel(ram_image(32, 0, 1), fixes=ALL_FIXES)
    cycles = top.run()
    assert cycles is not None and cycles < 40 and top.cs.ram_b == {} and top.ram_a[0][:2] == (0, 1)


def test_no_begin_no_action():
    top = TopLevel(ram_image(32, 9, 0), fixes=ALL_FIXES)
    for _ in range(500):
        top.edge()
    assert top.state == "waiting" and top.cs.ram_b == {} and top.ram_a[0] == (0, 9, 0, 0) and not top.events("start")


# ---- what the RTL as written does instead ----

@pytest.mark.parametrize("n", [1, 2, 3])
def test_i_three_or_fewer_points_are_never_streamed(n):
    """(i) `target_cnt < uram_latency` consumes counts 0..2, and `target_cnt < NUM_PTS` is then false for NUM_PTS <= 3: the branch that
    raises TRGT_VALID is never taken (:234-243).  The pass completes, ticks are written — and RAM B holds nothing.  The library writes
    the forces for NUM_PTS = 1, 2, 3 (a stated departure: tests/test_gpu_mailbox.py)."""
    top = TopLevel(ram_image(32, n, 1), fixes=ALL_FIXES)
    assert top.run() is not None
    assert top.cs.ram_b == {} and top.streamed == [[]] and top.ram_a[0][0] == 0 and top.ram_a[0][1] >= 1
    assert len(top.events("done")) == 1


def test_i_four_points_is_the_smallest_request_that_works():
    top = TopLevel(ram_image(32, 4, 1), fixes=ALL_FIXES)
    assert top.run() is not None and sorted(top.cs.ram_b) == [1, 2, 3, 4]


def test_ii_the_first_request_after_power_up_is_one_word_low():
    """(ii) THIS_PTR and TRGT_PTR power up as 0 in fabric (no initial value in the source; 'U' in simulation): the first pass loads its
    resident bodies from words 0..11 — the CONTROL word as a body — and streams targets 0..N-1; the force written at word L is that of
    the body at word L - 1.  From the second pass on both pointers are BASE_PTR (:192, 245)."""
    n = 9
    top = TopLevel(ram_image(32, n, 1), fixes=("poll_word0", "clear_begin"), this_ptr_init=0, trgt_ptr_init=0)
    assert top.run() is not None
    assert top.streamed[0] == list(range(0, n))                                     # the control word is a target, body N is not
    assert sorted(top.cs.ram_b) == list(range(1, n + 2))                            # lanes whose WORD index <= N are unmasked: words 0..9 -> 10 writes
    for k in sorted(top.cs.ram_b):
        assert lanes_of(top, k)[1] == {k - 1}
    top.post(n)
    assert top.run() is not None
    assert top.streamed[-1] == list(range(1, n + 1)) and all(lanes_of(top, k)[1] == {k} for k in range(1, n + 1))


def test_iii_waiting_polls_the_word_this_ptr_points_at():
    """(iii) READ_INT_ADDR is THIS_PTR in `waiting` (:276).  After a pass THIS_PTR = BASE_PTR = 1 (:192): BEGIN and NUM_PTS of the NEXT
    request are sampled from body 1's word — bit 0 of its x, the low 15 bits of its y — and word 0 is not looked at again."""
    top = TopLevel(ram_image(32, 9, 1), fixes=("ptr_init", "clear_begin"))
    # with THIS_PTR = 1 from power-up even the FIRST request is read from word 1: body 1 as posted has x even -> BEGIN = 0 -> nothing happens
    for _ in range(300):
        top.edge()
    assert not top.events("start") and top.ram_a[0] == (1, 9, 0, 0)
    # a body 1 whose x has bit 0 set and whose y ends in ...0101 starts a pass of 5 points
    top.ram_a[1] = (0x3F800001, 0x40000005, 0, 0)
    top.ram_a[0] = (0, 0, 0, 0)
    for _ in range(3000):
        top.edge()
        if top.events("done"):
            break
    assert top.events("begin_sampled_from_word")[0][2:] == (1, 5) and top.events("start")[0][2] == 5


def test_iv_begin_is_stale_on_return_from_complete():
    """(iv) one BEGIN, two passes: back in `waiting`, `if BEGIN_SIGNAL` still sees the '1' sampled before the pass (:181, 184) and enters
    block_setup again, with NUM_PTS <- bits 46:32 of the word on PL_READ_dout at that edge: word 0 as just rewritten, i.e. the TICK COUNT.
    The second `complete` then writes ticks = 0: RESET_STORE cleared clk_ctr and there was no rising edge of BEGIN to restart it (:136-139)."""
    top = TopLevel(ram_image(60, 40, 1), fixes=("ptr_init", "poll_word0", "drain_first"))
    top.clk_div = 990              # the free-running divider (:124-128) is about to wrap: ticks = 2 after ~1000 clocks
    assert top.run() is not None
    starts, writes = top.events("start"), top.events("ram_a_write")
    assert len(top.events("done")) == 2 and len(starts) == 2
    first_ticks = writes[0][3]
    assert starts[0][2] == 40 and first_ticks >= 2
    assert starts[1][2] == first_ticks                                              # the spurious pass's NUM_PTS is the first pass's tick count
    assert writes[1][3] == 0 and top.ram_a[0] == (0, 0, 0, 0)                       # ... and what the PS finally reads is ticks = 0


def test_v_a_request_that_reaches_the_top_of_the_ram_never_ends():
    """(v) at ram_depth = 2^6 (same RTL, smaller generic) the last block-group of NUM_PTS >= 61 pushes THIS_PTR past 63: it wraps,
    `THIS_PTR > NUM_PTS` (:189) stays false, and block-groups follow one another for ever; NUM_PTS = 60 ends.  At the RTL's 2^15 words the
    same arithmetic — ceil(N / 12) * 12 + 1 >= 2^15 — puts the limit at NUM_PTS >= 32761."""
    top = TopLevel(ram_image(64, 60, 1), fixes=ALL_FIXES, ptr_bits=6)
    assert top.run(max_cycles=20000) is not None and sorted(top.cs.ram_b) == list(range(1, 61))
    top = TopLevel(ram_image(64, 61, 1), fixes=ALL_FIXES, ptr_bits=6)
    assert top.run(max_cycles=40000) is None and len(top.streamed) > 12             # more block-groups than 61 bodies can fill
    first_bad = next(n for n in range(1, 1 << 15) if ((n + 11) // 12) * 12 + 1 >= (1 << 15))
    assert first_bad == 32761


@pytest.mark.parametrize("n", [9, 13, 30])
def test_vi_completion_overtakes_the_last_block_group(n):
    """(vi) :189-192 go to `complete` as soon as THIS_PTR > NUM_PTS, without the STORE_BUSY test of :193.  The PS reads "done" and the tick
    count while the last block-group's sums are still in the pipeline; RESET_STORE has put STORE_PTR back to 0 by the time they are
    stored, so they land at words 1.. — for N > 12 on top of the first block-group's forces, which are lost."""
    top = TopLevel(ram_image(n + 20, n, 1), fixes=("ptr_init", "poll_word0", "clear_begin"))
    assert top.run() is not None
    done_at = top.events("done")[0][0]
    last_write = top.cs.writes[-1][0]
    assert last_write > done_at + 100                                                # RAM B is still being written after BEGIN reads 0
    groups = (n + 11) // 12
    tail = n - 12 * (groups - 1)                                                     # bodies of the last block-group
    first_of_last = 12 * (groups - 1) + 1
    for k in range(1, tail + 1):
        assert lanes_of(top, k)[1] == {first_of_last + k - 1}                        # word k holds the force of body 12 (G - 1) + k
    if groups > 1:
        assert sorted(top.cs.ram_b) == list(range(1, 12 * (groups - 1) + 1))         # nothing was ever written beyond the first G - 1 groups
        for k in range(tail + 1, 13):
            assert lanes_of(top, k)[1] == {k}                                        # what is left of the first block-group
    # with the STORE_BUSY test first, the same request is the protocol's
    top = TopLevel(ram_image(n + 20, n, 1), fixes=ALL_FIXES)
    assert top.run() is not None
    assert top.cs.writes[-1][0] < top.events("done")[0][0] and all(lanes_of(top, k)[1] == {k} for k in range(1, n + 1))


@pytest.mark.parametrize("n", [9, 40, 100])
def test_the_modelled_rtl_with_numbers_writes_the_fixtures_ram_b_image(oracle, n):
    """The whole chain in one test: the sequencer + compute_store cycle model decides WHO is summed into WHAT and WHERE it is written
    (which RAM A word each lane holds, which targets reach which of the sixteen partial sums in which order, which RAM B word takes the
    result); the oracle's pair arithmetic and final_adder's tree supply the numbers; and the RAM B image that comes out — words 1..N —
    is, bit for bit, the `forces0` of tests/golden/rtl_n*.json, the exact-rational third statement of the RTL (tests/golden/make_system.py
    forces_rtl) that the GPU's faithful mailbox reproduces in tests/test_gpu_mailbox.py.  Nothing about the order of summation or the address
    of a result is assumed here: both are read off the model's RAM B."""
    import glob
    import json
    import os

    import numpy as np
    import oracle as O
    path = [f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rtl_*.json")) if json.load(open(f))["n"] == n][0]
    d = json.load(open(path))
    words = lambda key: np.array([int(x, 16) for x in d[key]], np.uint32).view(np.float32).reshape(-1, 4)   # noqa: E731
    pos, want = words("pos0"), words("forces0")
    top = TopLevel(ram_image(n + 20, n, 1), fixes=ALL_FIXES)
    assert top.run() is not None and sorted(top.cs.ram_b) == list(range(1, n + 1))
    image = np.zeros((n + 1, 4), np.float32)
    for k in range(1, n + 1):
        din = top.cs.ram_b[k]
        for dim in range(3):                                            # {Fx, Fy, Fz}: S/compute_store.vhd:213
            lane, d_, slots = din[dim]
            assert d_ == dim
            leaves = np.zeros(16, np.float32)
            for t, slot in enumerate(slots):
                if not slot:
                    continue                                             # no item reached results(t): 0.0 (S/fxyz.vhd:177-181)
                this_word = slot[0][1][lane]
                src = np.ascontiguousarray(pos[[j - 1 for j, _ in slot]])               # RAM A word j holds body j (1-based)
                row = pos[this_word - 1:this_word]
                leaves[t] = oracle.forces_f32(row, src, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_SEQ)[0, dim]   # one fma chain from 0.0
            image[k, dim] = oracle.tree16(leaves)                        # S/final_adder.vhd:88-104
    assert np.array_equal(image[1:].view(np.uint32), want.view(np.uint32))
    assert not image[0].any()
