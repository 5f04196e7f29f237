#!/bin/bash
set -u
out=gpurun_out/r03_run12
mkdir -p $out
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
for n in 8192 9216 12288 16384 20480 24576 32768; do
  sw --wall --n $n --steps 2048 --rounds 3 --configs "isa1:1:0,isa1:1:16:ws=4:fuse=1" > $out/n$n.txt 2>&1; cat $out/n$n.txt
done
for n in 4096 8192 12288 16384; do
  sw --fp64 --wall --n $n --steps 512 --rounds 3 --configs "isa1:1:0,isa1:1:16:ws=4:fuse=1,isa1:1:8:ws=4:fuse=1,isa1:1:8:ws=4:fuse=0,isa1:1:4:ws=16:fuse=0" > $out/f64_n$n.txt 2>&1; cat $out/f64_n$n.txt
done
