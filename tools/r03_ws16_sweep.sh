#!/bin/bash
# 16-wave workgroups (NBODY_OPT_WSPLIT = 16) against 4-wave ones, wall clock per step on the graph path
set -u
out=gpurun_out/r03_ws16
mkdir -p $out
run() { n=$1; steps=$2; cfgs=$3; timeout -k 10 150 python3 tools/sweep.py --wall --n $n --steps $steps --rounds 3 --configs "$cfgs" > $out/n$n.txt 2>&1; cat $out/n$n.txt; }
run 2048 4096 "isa1:1:0,isa1:1:2:ws=16,isa1:1:4:ws=16,isa1:1:1:ws=16"
run 4096 4096 "isa1:1:0,isa1:1:4:ws=16,isa1:1:2:ws=16,isa1:1:8:ws=16,isa1:1:4:ws=16:fuse=1,isa1:1:2:ws=16:fuse=1"
run 8192 3072 "isa1:1:0,isa1:1:2:ws=16,isa1:1:4:ws=16,isa1:1:1:ws=16,isa1:1:2:ws=16:fuse=1"
run 16384 2048 "isa1:1:0,isa1:1:1:ws=16:long=0,isa1:1:1:ws=16:long=1,isa1:1:2:ws=16:long=0,isa1:1:2:ws=16:long=1"
run 32768 768 "isa1:1:0,isa1:1:1:ws=16:long=0,isa1:1:1:ws=16:long=1,isa1:1:2:ws=16:long=0"
run 65536 256 "isa1:1:0,isa1:1:1:ws=16:long=0,isa1:1:2:ws=16:long=0,isa1:1:4:ws=16:long=0"
run 131072 64 "isa1:1:0,isa1:1:1:ws=16,isa1:1:2:ws=16,isa1:1:4:ws=16"
run 262144 32 "isa1:1:0,isa1:1:1:ws=16,isa1:1:2:ws=16,isa1:1:4:ws=16,isa1:1:8:ws=16"
timeout -k 10 200 python3 tools/sweep.py --n 1048576 --steps 2 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:8:ws=16,isa1:1:16:ws=16" > $out/n1m.txt 2>&1; cat $out/n1m.txt
timeout -k 10 200 python3 tools/sweep.py --fp64 --n 262144 --steps 2 --rounds 2 --configs "isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:8:ws=16,isa1:1:4:ws=16,isa1:1:2:ws=16" > $out/fp64_n262144.txt 2>&1; cat $out/fp64_n262144.txt
