#!/bin/bash
# do resident sets that divide the work evenly pay?  (long-buffer fp32 kernel: 7 waves per SIMD; fp64: 5)
set -u
out=gpurun_out/r03_run10
mkdir -p $out
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
sw --wall --n 16384 --steps 2048 --rounds 3 --configs "isa1:1:16:ws=4,isa1:1:14:ws=4:long=1,isa1:1:7:ws=4:long=1,isa1:1:21:ws=4:long=1,isa1:1:28:ws=4:long=1,isa1:1:8:ws=4:long=0,isa1:1:16:ws=4:long=0" > $out/n16384.txt 2>&1; cat $out/n16384.txt
sw --wall --n 12288 --steps 2048 --rounds 3 --configs "isa1:1:0,isa1:1:14:ws=4:long=1,isa1:1:7:ws=4:long=1,isa1:1:16:ws=4:long=1" > $out/n12288.txt 2>&1; cat $out/n12288.txt
sw --wall --n 24576 --steps 1024 --rounds 3 --configs "isa1:1:0,isa1:1:14:ws=4:long=1,isa1:1:7:ws=4:long=1,isa1:1:16:ws=4:long=0,isa1:1:16:ws=4:long=1" > $out/n24576.txt 2>&1; cat $out/n24576.txt
sw --wall --n 32768 --steps 768 --rounds 3 --configs "isa1:1:0,isa1:1:7:ws=4:long=1,isa1:1:14:ws=4:long=1,isa1:1:16:ws=4:long=1,isa1:1:8:ws=4:long=0" > $out/n32768.txt 2>&1; cat $out/n32768.txt
sw --fp64 --n 131072 --steps 3 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:5:ws=4,isa1:1:10:ws=4,isa1:1:15:ws=4" > $out/f64_n131072.txt 2>&1; cat $out/f64_n131072.txt
sw --fp64 --n 262144 --steps 2 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:5:ws=4,isa1:1:10:ws=4" > $out/f64_n262144.txt 2>&1; cat $out/f64_n262144.txt
