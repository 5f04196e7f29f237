"""Cycle models of the control logic that IS in the reference tree — test infrastructure, CPU only, no arithmetic: every value that
travels through the datapath is SYMBOLIC (which RAM A word it came from), so what these models establish is WHO is summed into WHAT
and WHERE it is written, never a number.

  FxyzControl    S/fxyz.vhd:129-215     flush / feedback mux, scatter counter, results latch, fma_busy, scatter_complete (one lane)
  ComputeStore   S/compute_store.vhd:117-242   store_busy, gather iterators, the countdown through the adder tree, store iterators,
                                        STORE_PTR / write_we / write_addr, and the RAM B port sampling (we, addr, din) at the edge
  TopLevel       S/top_level.vhd:121-146, 165-280   tick counter, THIS_PTR shift register, the four-state FSM, the read-address mux,
                                        the RAM A port (read latency uram_latency, the word-0 write-back)

VHDL semantics kept: inside one clocked process every signal read is the value BEFORE the edge; concurrent assignments
(write_addr, READ_INT_ADDR, FLUSH_ACTV, ...) are functions of the current register values.  The vendor IP that is not in the tree is
modelled as what its declaration and the design's own constants say it is: a pipeline of `latency` stages with a valid bit
(S/top_level.vhd:37-42) — a value on an IP's input during cycle c is on its output during cycle c + latency, exactly as the design's
own `translate` delay line (S/adder_choose.vhd:66-78) stands in for an adder.  The ps_pl block RAM (absent: S/top_level.vhd:100-117)
is modelled as the design assumes it: data for the address of cycle c on PL_READ_dout during cycle c + uram_latency (the FSM pairs
PL_READ_dout with THIS_PTR_SHR(uram_latency), S/top_level.vhd:165-174, 201-208); writes sampled at the edge after (we, addr, din) are
driven.  Registers WITHOUT an initial value in the RTL (THIS_PTR, TRGT_PTR: S/top_level.vhd:56, 58) take `this_ptr_init` /
`trgt_ptr_init` (0 = what FPGA fabric powers up with)."""
from collections import deque

FMA_LATENCY = 16     # S/top_level.vhd:40
ZERO = ()            # symbolic 0.0: the empty sum


class FxyzControl:
    """One axis of S/fxyz.vhd:120-184 (+ lane 0's fma_busy / scatter_complete, :186-208).  step() is one clock: combinational signals
    from the registers' current values, then the rising edge.  `fma` maps (a_item, c_value) -> value; the default builds the tuple of
    summed items."""

    def __init__(self, latency=FMA_LATENCY, fma=None):
        self.L = latency
        self.pipe = [(0, None)] * latency        # the fma IP: pipe[-1] is on m_axis_result this cycle         :120-127
        self.flush_cnt = latency                 # signal FLUSH_CNT ... := fma_latency                          :86
        self.scttr_cnt = 0                       # signal SCTTR_CNT ... := 0                                    :87
        self.valid_prev = 0                      # VALID_FMA_PREV                                               :79
        self.results = ["U"] * latency           # results(0 .. fma_latency - 1) of this axis, 'U' until written
        self.fma = fma or (lambda item, c: c + (item,))
        self.scatter_complete = 0
        self.fma_busy = 0

    def step(self, valid_fma, item=None, valid_in=0):
        L = self.L
        valid_fx, fx_out = self.pipe[-1]
        flush_actv = valid_fma == 0 or self.flush_cnt != 0                                                   # :142
        fx_in = ZERO if flush_actv else fx_out                                                                # :143
        scttr_actv = (valid_fma == 0 and self.valid_prev == 1 and self.scttr_cnt == 0) or self.scttr_cnt != 0  # :167
        # ---- rising edge ----
        if valid_fma:
            assert fx_in is not None, "the feedback mux selected an fma output that carries no item"
            entering = (1, self.fma(item, fx_in))
        else:
            entering = (0, None)
        if valid_fma == 0:                                                                                    # :133-137
            flush_next = L
        elif self.flush_cnt != 0:
            flush_next = self.flush_cnt - 1
        else:
            flush_next = self.flush_cnt
        scttr_next = self.scttr_cnt
        if scttr_actv:                                                                                        # :150-156, :172-182
            self.results[self.scttr_cnt] = fx_out if valid_fx else ZERO
            scttr_next = 0 if self.scttr_cnt == L - 1 else self.scttr_cnt + 1
        self.scatter_complete = 1 if self.scttr_cnt == L - 1 else 0                                           # :191-195
        if self.scttr_cnt == L - 1:                                                                           # :202-206
            self.fma_busy = 0
        elif valid_in:
            self.fma_busy = 1
        self.valid_prev = valid_fma                                                                           # :163
        self.pipe = [entering] + self.pipe[:-1]
        self.flush_cnt, self.scttr_cnt = flush_next, scttr_next

    def run_stream(self, n, idle_before=3, first_item=0):
        """the sequencer's compute state: n items on consecutive clocks (S/top_level.vhd:233-254), then idle until the
        scatter is over (the FSM waits for STORE_BUSY, S/top_level.vhd:193)"""
        for _ in range(idle_before):
            self.step(0)
        for k in range(n):
            self.step(1, first_item + k)
        done = 0
        for _ in range(4 * self.L):
            self.step(0)
            done |= self.scatter_complete
        assert done and self.scttr_cnt == 0
        return list(self.results)


def expected_slot(n, t, L=FMA_LATENCY, first_item=0):
    k = n - L + t                      # the item whose fma output is on the bus when results(t) is latched
    if k < 0:
        return ZERO
    return tuple(first_item + j for j in range(k % L, k + 1, L))


class ComputeStore:
    """S/compute_store.vhd.  The twelve fxyz lanes share every control signal (only lane 0's busy / complete flags are wired,
    :98-108), so ONE FxyzControl carries the stream; an item is (target word, snapshot of the twelve `this` words), and lane b's
    result vector is that control's results read with snapshot[b].  edge() is one rising edge of aclk."""

    def __init__(self, num_blocks=12, fma_latency=FMA_LATENCY, add_final_latency=11, pipe_depth=83, ptr_bits=15):
        self.nb, self.L = num_blocks, fma_latency
        levels = (fma_latency - 1).bit_length()                       # ceil_log2(fma_latency)                          :66
        self.add_pipeline_latency = levels * add_final_latency        # 4 x 11 = 44
        self.ptr_mask = (1 << ptr_bits) - 1
        self.fx = FxyzControl(fma_latency)
        # valid_in -> VALID_FMA: diff 11 + (mult 6 + add 11 | fma 16 + 1 balancing) + add 11 + rsqrt 32 + 2 x mult 6 = 83 clocks
        # (S/dxy.vhd:94-122, S/dxyz_soft.vhd:70-73, 119-150, S/fxyz.vhd:97-106); only the timing depends on it
        self.front = deque([(0, None)] * pipe_depth, maxlen=pipe_depth)
        self.tree = deque([(0, None)] * self.add_pipeline_latency, maxlen=self.add_pipeline_latency)   # final_adder: BUFF_SUM(c) = FMA_RES_CUR(c - 44)
        # registers, with the RTL's initial values                                                                          :69-86
        self.store_busy = 0
        self.write_ongoing = 0
        self.block_iter_add = self.dim_iter_add = 0
        self.block_iter_store = self.dim_iter_store = 0
        self.fma_res_cur, self.fma_valid_cur = None, 0
        self.gather_iter = 0
        self.store_ptr = 0                                            # STORE_PTR := ZERO_PTR := 0                           :76-77
        self.write_we = 0
        self.write_int_din = [None, None, None]
        self.ram_b = {}                                               # word -> {Fx, Fy, Fz} as written
        self.writes = []                                              # (cycle, word, din) in the order the RAM saw them
        self.cycle = 0

    def edge(self, valid_in, this_words, target_word, mask, reset_store):
        nb, L, lat = self.nb, self.L, self.add_pipeline_latency
        fx = self.fx
        # ---- concurrent signals, from the registers as they stand ----
        scatter_complete, fma_busy = fx.scatter_complete, fx.fma_busy
        write_addr = self.store_ptr                                   # :241  a function of STORE_PTR, no register in between
        buff_sum = self.tree[0][1]                                    # the adder tree's output this cycle
        valid_fma, item = self.front[0]
        # ---- the RAM B port: what it samples at this edge is what was driven during the cycle that ends here ----
        if self.write_we:                                             # :240-242
            word = tuple(self.write_int_din)
            self.ram_b[write_addr] = word
            self.writes.append((self.cycle, write_addr, word))
        # ---- clocked processes (all read the OLD values) ----
        n = {}
        n["store_busy"] = 1 if (fma_busy or scatter_complete or self.write_ongoing) else 0                    # :117-126
        if scatter_complete:                                                                                  # :128-137
            n["write_ongoing"] = 1
        elif self.dim_iter_store == 2 and self.block_iter_store == nb - 1:
            n["write_ongoing"] = 0
        if self.dim_iter_add == 2:                                                                            # :140-151
            n["block_iter_add"] = 0 if self.block_iter_add == nb - 1 else self.block_iter_add + 1
        if scatter_complete or self.dim_iter_add != 0 or self.block_iter_add != 0:                            # :154-173
            n["fma_res_cur"] = (self.block_iter_add, self.dim_iter_add, tuple(fx.results))                    # FMA_RES(block)(dim x 16 ..)
            n["dim_iter_add"] = 0 if self.dim_iter_add == 2 else self.dim_iter_add + 1
            n["fma_valid_cur"] = 1
        else:
            n["fma_valid_cur"] = 0
        if scatter_complete:                                                                                  # :176-187
            n["gather_iter"] = self.gather_iter + 1
        elif self.gather_iter == 0 or self.gather_iter == lat + 1:
            n["gather_iter"] = 0
        else:
            n["gather_iter"] = self.gather_iter + 1
        if self.dim_iter_store == 2:                                                                          # :190-201
            n["block_iter_store"] = 0 if self.block_iter_store == nb - 1 else self.block_iter_store + 1
        if self.gather_iter == lat + 1 or self.dim_iter_store != 0 or self.block_iter_store != 0:             # :204-218
            n["dim_iter_store"] = 0 if self.dim_iter_store == 2 else self.dim_iter_store + 1
            din = list(self.write_int_din)
            din[self.dim_iter_store] = buff_sum
            n["write_int_din"] = din
        if reset_store:                                                                                       # :221-238
            n["store_ptr"] = 0
            n["write_we"] = 0
        elif self.dim_iter_store == 2:
            n["write_we"] = 1 if mask[self.block_iter_store] else 0
            n["store_ptr"] = (self.store_ptr + 1) & self.ptr_mask
        else:
            n["write_we"] = 0
        # the IP pipelines and the lanes' control
        self.tree.append((self.fma_valid_cur, self.fma_res_cur))
        self.front.append((1, (target_word, tuple(this_words))) if valid_in else (0, None))
        fx.step(valid_fma, item, valid_in=valid_in)
        for k, v in n.items():
            setattr(self, k, v)
        self.cycle += 1


WAITING, BLOCK_SETUP, COMPUTE, COMPLETE = "waiting", "block_setup", "compute", "complete"


class TopLevel:
    """S/top_level.vhd.  ram_a: list of words, word = (w0, w1, w2, w3) 32-bit fields, field 0 = bits 31:0.  A body's fields are opaque
    here (the lanes carry the WORD INDEX they were loaded from); only the control fields are interpreted: BEGIN = bit 0 of field 0,
    NUM_PTS = bits 46:32 = the low 15 bits of field 1 (:184-185).
    fixes: which of the four one-line repairs the protocol needs are applied (none = the RTL as written):
      "ptr_init"    THIS_PTR, TRGT_PTR := BASE_PTR at power-up                         (:56, 58 have no initial value)
      "poll_word0"  READ_INT_ADDR = 0 in `waiting`                                     (:276 reads THIS_PTR there)
      "clear_begin" BEGIN_SIGNAL <= '0' when `complete` is entered                     (:184 is its only assignment)
      "drain_first" block_setup waits for STORE_BUSY = '0' before it goes to `complete` (:189-192 test THIS_PTR > NUM_PTS first)"""

    def __init__(self, ram_a, num_blocks=12, uram_latency=3, ptr_bits=15, this_ptr_init=0, trgt_ptr_init=0, fixes=(), **cs_kw):
        self.ram_a = [tuple(w) for w in ram_a]
        self.nb, self.ul = num_blocks, uram_latency
        self.mask_ptr = (1 << ptr_bits) - 1
        self.fixes = set(fixes)
        self.cs = ComputeStore(num_blocks=num_blocks, ptr_bits=ptr_bits, **cs_kw)
        self.state = WAITING
        self.base_ptr = 1                                                                                     # :55
        self.this_ptr = self.base_ptr if "ptr_init" in self.fixes else this_ptr_init
        self.trgt_ptr = self.base_ptr if "ptr_init" in self.fixes else trgt_ptr_init
        self.shr = [self.this_ptr] * (uram_latency + 1)                # THIS_PTR_SHR; (0) is THIS_PTR itself           :165-174
        self.rd = [None] * uram_latency                                # the RAM's read pipeline: rd[-1] is PL_READ_dout
        self.block_cnt = self.target_cnt = self.complete_cnt = 0
        self.pl_read_we = 0
        self.this_words = [None] * num_blocks                          # X/Y/Z_THIS(b): the word index they were loaded from
        self.write_mask = [0] * num_blocks
        self.trgt, self.trgt_valid = None, 0
        self.num_pts, self.begin_signal, self.begin_prev = 0, 0, 0
        self.reset_store = 0
        self.clk_ctr, self.clk_div = 0, 0
        self.cycle = 0
        self.log = []                                                  # (cycle, event, ...) for the tests
        self.streamed = []                                             # per block-group: the target words that entered with TRGT_VALID
        self._stream = None

    def _control(self, word):
        return (word[0] & 1, word[1] & self.mask_ptr) if word is not None else (0, 0)

    def edge(self):
        nb, ul, M = self.nb, self.ul, self.mask_ptr
        st = self.state
        # ---- concurrent signals ----
        if st in (WAITING, BLOCK_SETUP):                                                                      # :276-278
            read_addr = 0 if (st == WAITING and "poll_word0" in self.fixes) else self.this_ptr
        elif st == COMPUTE:
            read_addr = self.trgt_ptr
        else:
            read_addr = 0
        dout_word, dout_addr = self.rd[-1] if self.rd[-1] is not None else (None, None)
        shr_last = self.shr[ul]
        store_busy = self.cs.store_busy
        # ---- the RAM A port at this edge: write first (word-0 write-back), then the read that enters the pipeline ----
        if self.pl_read_we:                                                                                   # :146, 258
            self.ram_a[read_addr] = (0, self.clk_ctr, 0, 0)
            self.log.append((self.cycle, "ram_a_write", read_addr, self.clk_ctr))
        rd_next = [(self.ram_a[read_addr] if read_addr < len(self.ram_a) else (0, 0, 0, 0), read_addr)] + self.rd[:-1]
        # ---- compute_store sees the registers as they stand ----
        if self.trgt_valid:
            self._stream.append(self.trgt)
        self.cs.edge(self.trgt_valid, self.this_words, self.trgt, self.write_mask, self.reset_store)
        # ---- clocked processes ----
        n = {}
        n["clk_div"] = 0 if self.clk_div == 999 else self.clk_div + 1                                        # :121-131
        n["begin_prev"] = self.begin_signal
        if self.reset_store:                                                                                  # :133-144
            n["clk_ctr"] = 0
        elif self.begin_signal and not self.begin_prev:
            n["clk_ctr"] = self.clk_ctr + 1
        elif self.clk_div == 999 and self.clk_ctr != 0:
            n["clk_ctr"] = self.clk_ctr + 1
        shr_next = [None] + self.shr[:-1]                                                                     # :168-173
        if st == WAITING:                                                                                     # :180-186
            b, npts = self._control(dout_word)
            n["begin_signal"], n["num_pts"] = b, npts
            if self.begin_signal:
                n["state"] = BLOCK_SETUP
                self.log.append((self.cycle, "start", npts))          # the NUM_PTS the pass runs with is the one latched at this very edge
                self._stream = None
            if dout_addr is not None and b:
                self.log.append((self.cycle, "begin_sampled_from_word", dout_addr, npts))
            n["pl_read_we"] = 0
        elif st == BLOCK_SETUP:                                                                               # :187-232
            bc = self.block_cnt
            lane = bc - ul
            if bc == 0:
                if self.this_ptr > self.num_pts and store_busy and "drain_first" in self.fixes:
                    pass                                               # (repair: the last block-group's forces are stored before `complete`)
                elif self.this_ptr > self.num_pts:
                    n["state"] = COMPLETE
                    n["this_ptr"] = self.base_ptr
                    if "clear_begin" in self.fixes:
                        n["begin_signal"] = 0
                elif not store_busy:
                    n["this_ptr"] = (self.this_ptr + 1) & M
                    n["block_cnt"] = bc + 1
            elif bc < ul:
                n["this_ptr"] = (self.this_ptr + 1) & M
                n["block_cnt"] = bc + 1
            else:
                mask = list(n.get("write_mask", self.write_mask))
                mask[lane] = 0 if shr_last > self.num_pts else 1                                              # :201-205
                words = list(self.this_words)
                words[lane] = dout_addr                                                                       # :206-208
                n["write_mask"], n["this_words"] = mask, words
                if bc < nb:
                    n["this_ptr"] = (self.this_ptr + 1) & M
                    n["block_cnt"] = bc + 1
                elif bc < nb - 1 + ul:
                    n["block_cnt"] = bc + 1
                else:
                    n["state"] = COMPUTE
                    n["block_cnt"] = 0
                    self._stream = []
                    self.streamed.append(self._stream)
        elif st == COMPUTE:                                                                                   # :233-254
            tc = self.target_cnt
            if tc < ul:
                n["trgt_ptr"] = (self.trgt_ptr + 1) & M
                n["target_cnt"] = tc + 1
            elif tc < self.num_pts:
                n["trgt"], n["trgt_valid"] = dout_addr, 1
                n["trgt_ptr"] = (self.trgt_ptr + 1) & M
                n["target_cnt"] = tc + 1
            elif tc < self.num_pts + ul:
                n["trgt_ptr"] = self.base_ptr
                n["trgt"] = dout_addr
                n["target_cnt"] = tc + 1
            else:
                n["trgt_valid"] = 0
                n["state"] = BLOCK_SETUP
                n["target_cnt"] = 0
        else:                                                                                                 # :255-269
            cc = self.complete_cnt
            if cc == 0:
                n["reset_store"], n["pl_read_we"], n["complete_cnt"] = 1, 1, cc + 1
            elif cc == 1:
                n["reset_store"], n["pl_read_we"], n["complete_cnt"] = 0, 0, cc + 1
            elif cc == 15:
                n["state"], n["complete_cnt"] = WAITING, 0
                self.log.append((self.cycle, "done"))
            else:
                n["complete_cnt"] = cc + 1
        for k, v in n.items():
            setattr(self, k, v)
        shr_next[0] = self.this_ptr                                    # THIS_PTR_SHR(0) <= THIS_PTR, concurrent             :165
        self.shr = shr_next
        self.rd = rd_next
        self.cycle += 1

    # ---- the PS side ----
    def post(self, num_pts, begin=1):
        self.ram_a[0] = (begin, num_pts, 0, 0)

    def run(self, max_cycles=200000, until_done=1):
        """clock until `until_done` passes have completed and the FSM sits in `waiting` with nothing pending; returns the cycles taken,
        or None when max_cycles went by first (a pass that never ends)"""
        c0, done0 = self.cycle, sum(1 for e in self.log if e[1] == "done")
        while self.cycle - c0 < max_cycles:
            self.edge()
            done = sum(1 for e in self.log if e[1] == "done") - done0
            if done >= until_done and self.state == WAITING and not self.begin_signal and not self.cs.store_busy:
                return self.cycle - c0
        return None

    def events(self, kind):
        return [e for e in self.log if e[1] == kind]
