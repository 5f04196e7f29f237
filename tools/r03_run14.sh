#!/bin/bash
set -u
out=gpurun_out/r03_run14
mkdir -p $out
export TMPDIR=/tmp
sw() { timeout -k 10 300 python3 tools/sweep.py "$@"; }
sw --fp64 --wall --n 512 --steps 1024 --rounds 3 --configs "isa1:1:0,isa1:1:2:ws=4:fuse=1,isa1:1:1:ws=4,isa1:1:2:ws=4:fuse=0" > $out/f64_n512.txt 2>&1; cat $out/f64_n512.txt
sw --fp64 --wall --n 1024 --steps 1024 --rounds 3 --configs "isa1:1:0,isa1:1:4:ws=4:fuse=1,isa1:1:2:ws=4:fuse=1,isa1:1:4:ws=4:fuse=0" > $out/f64_n1024.txt 2>&1; cat $out/f64_n1024.txt
sw --fp64 --wall --n 2048 --steps 1024 --rounds 3 --configs "isa1:1:0,isa1:1:8:ws=4:fuse=1,isa1:1:4:ws=4:fuse=1,isa1:1:8:ws=4:fuse=0" > $out/f64_n2048.txt 2>&1; cat $out/f64_n2048.txt
sw --n 1572864 --steps 1 --rounds 3 --configs "isa1:1:0,isa1:1:12:ws=4,isa1:1:16:ws=4:xcd=1,isa1:1:16:ws=4:xcd=0,isa1:1:8:ws=4:xcd=1" > $out/f32_n1572864.txt 2>&1; cat $out/f32_n1572864.txt
for cfg in "f32_1p5m_auto --bodies 1572864 --steps 1" "f32_1p5m_sub16x --bodies 1572864 --jsub 16 --xcd-map 1 --steps 1"; do
  set -- $cfg; tag=$1; shift
  echo "== traffic $tag"; timeout -k 10 300 tools/pmc_traffic.sh $tag "$@" 2>&1 | tail -6
done
