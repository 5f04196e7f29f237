#!/bin/bash
set -u
out=gpurun_out/r03_run13
mkdir -p $out
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
for n in 2048 4096 8192 12288; do
  sw --fp64 --wall --n $n --steps 512 --rounds 3 --configs "isa1:1:0,isa1:1:16:ws=4:fuse=1" > $out/f64_n$n.txt 2>&1; cat $out/f64_n$n.txt
done
