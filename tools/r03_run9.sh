#!/bin/bash
set -u
out=gpurun_out/r03_run9
mkdir -p $out
export TMPDIR=/tmp
# what the IEEE-exact arithmetic (bit-identical to the CPU oracle) costs, and the RTL's rounding points for d2
timeout -k 10 300 python3 tools/sweep.py --n 262144 --steps 1 --rounds 2 --configs "isa1:1:0,smem:1:0,smem:1:0:arith=reference,smem:1:0:arith=strict,smem:1:0:arith=refstrict,smem:2:0:arith=strict,smem:4:0:arith=strict" > $out/arith_n262144.txt 2>&1; cat $out/arith_n262144.txt
# error budget with the wave-split orders
timeout -k 10 500 python3 tools/error_budget.py 4096 65536 262144 1048576 > $out/error_budget.md 2> $out/error_budget.err; cat $out/error_budget.md | cut -c1-220
