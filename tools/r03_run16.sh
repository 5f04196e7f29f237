#!/bin/bash
# does the engine's own choice hold up off the powers of two, and for the per-rank shapes of multi-GPU jobs?
set -u
out=gpurun_out/r03_run16
mkdir -p $out
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
for n in 30000 50000 100000 300000; do
  steps=256; [ $n -ge 100000 ] && steps=32; [ $n -ge 300000 ] && steps=6
  sw --wall --n $n --steps $steps --rounds 3 --configs "isa1:1:0,isa1:1:8:ws=4,isa1:1:16:ws=4,isa1:1:32:ws=4,isa1:1:4:ws=4" > $out/n$n.txt 2>&1; cat $out/n$n.txt
done
# per-rank shapes: jsl = P slices on one GPU (the summation order and launch shapes of a P-rank job's rank, all slices in one launch)
sw --n 262144 --steps 3 --rounds 3 --configs "isa1:1:0:jsl=8,isa1:1:1:ws=4:jsl=8,isa1:1:2:ws=4:jsl=8,isa1:1:4:ws=4:jsl=8,isa1:1:8:ws=4:jsl=8" > $out/n262144_p8.txt 2>&1; cat $out/n262144_p8.txt
for P in 2 4 8; do NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus $P | tail -1; done
for P in 2 8; do NBODY_OVERSUBSCRIBE=1 ./build/nbody 262144 21 --gpus $P | tail -1; done
./build/nbody 262144 21 | tail -1
for P in 8; do NBODY_OVERSUBSCRIBE=1 ./build/nbody 65536 101 --gpus $P | tail -1; NBODY_OVERSUBSCRIBE=1 ./build/nbody 65536 101 --gpus $P --wsplit 4 | tail -1; NBODY_OVERSUBSCRIBE=1 ./build/nbody 65536 101 --gpus $P --wsplit 4 --jsub 4 | tail -1; done
./build/nbody 65536 101 | tail -1
