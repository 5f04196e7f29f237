#!/bin/bash
# how many global segments when the sources no longer fit one L2: time (sweep.py) and traffic (pmc_traffic.sh)
set -u
out=gpurun_out/r03_run7
mkdir -p $out
export TMPDIR=/tmp
sw() { timeout -k 10 400 python3 tools/sweep.py "$@"; }
sw --n 262144 --steps 3 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:4:ws=4,isa1:1:4:ws=4:xcd=1,isa1:1:2:ws=4:xcd=1,isa1:1:2:ws=4,isa1:1:8:ws=4:xcd=1" > $out/f32_n262144.txt 2>&1; cat $out/f32_n262144.txt
sw --n 524288 --steps 2 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:4:ws=4,isa1:1:4:ws=4:xcd=1,isa1:1:8:ws=4:xcd=1,isa1:1:16:ws=4:xcd=1" > $out/f32_n524288.txt 2>&1; cat $out/f32_n524288.txt
sw --fp64 --n 131072 --steps 3 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:4:ws=4,isa1:1:2:ws=4,isa1:1:2:ws=4:xcd=1,isa1:1:4:ws=4:xcd=1" > $out/f64_n131072.txt 2>&1; cat $out/f64_n131072.txt
sw --fp64 --n 524288 --steps 1 --rounds 2 --configs "isa1:1:8:ws=4,isa1:1:8:ws=4:xcd=1,isa1:1:16:ws=4:xcd=1,isa1:1:4:ws=4:xcd=1" > $out/f64_n524288.txt 2>&1; cat $out/f64_n524288.txt
sw --fp64 --n 1048576 --steps 1 --rounds 2 --configs "isa1:1:8:ws=4,isa1:1:16:ws=4:xcd=1,isa1:1:32:ws=4:xcd=1,isa1:1:8:ws=1" > $out/f64_n1m.txt 2>&1; cat $out/f64_n1m.txt
for cfg in "f32_262k_sub4x --bodies 262144 --jsub 4 --xcd-map 1" "f32_262k_sub8 --bodies 262144 --jsub 8" "f64_1m_sub8 --fp64 --bodies 1048576 --jsub 8 --steps 1" "f64_1m_sub16x --fp64 --bodies 1048576 --jsub 16 --xcd-map 1 --steps 1"; do
  set -- $cfg; tag=$1; shift
  echo "== traffic $tag"; timeout -k 10 300 tools/pmc_traffic.sh $tag "$@" 2>&1 | tail -6
done
# the 8-rank decomposition of N = 1M on this one GPU (virtual ranks): segments per slice
for js in 1 2 4; do NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus 8 --jsub $js | tail -1; done
NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus 2 --jsub 4 | tail -1
NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus 2 --jsub 8 | tail -1
./build/nbody 1048576 5 | tail -1
