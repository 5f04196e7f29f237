#!/bin/bash
# between the 16-wave regime (n_local <= 8192) and N = 16384: which layout, how many segments, which combine
set -u
out=gpurun_out/r03_run11
mkdir -p $out
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
for n in 8192 9216 10240 12288 14336 16384 20480; do
  sw --wall --n $n --steps 2048 --rounds 3 --configs "isa1:1:0,isa1:1:16:ws=4:fuse=1,isa1:1:16:ws=4:fuse=0,isa1:1:8:ws=4:fuse=1,isa1:1:2:ws=16:fuse=0,isa1:1:2:ws=16:fuse=1,isa1:1:1:ws=16,isa1:1:4:ws=16:fuse=1,isa1:1:16:ws=4:fuse=1:long=0" > $out/n$n.txt 2>&1; cat $out/n$n.txt
done
