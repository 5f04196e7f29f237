#!/bin/bash
# round 3, GPU call 3: steady-state comm probe; steps per HIP graph at small N; the wave split at N = 1M (time + traffic)
set -u
out=gpurun_out/r03_run2
mkdir -p $out
timeout -k 10 240 python3 tools/comm_probe.py > $out/comm_probe.md 2> $out/comm_probe.err; echo "probe rc=$?"; cat $out/comm_probe.md
for n in 1024 4096 16384; do
  timeout -k 10 120 python3 tools/sweep.py --wall --n $n --steps 4000 --rounds 3 --configs "isa1:1:0:graph=2,isa1:1:0:graph=4,isa1:1:0:graph=8,isa1:1:0:graph=16,isa1:1:0:graph=64" > $out/graph_n$n.txt 2>&1
  cat $out/graph_n$n.txt
done
timeout -k 10 300 python3 tools/sweep.py --n 1048576 --steps 2 --rounds 3 --configs "isa1:1:8:ws=1,isa1:1:8:ws=4,isa1:1:4:ws=4,isa1:1:2:ws=4,isa1:1:16:ws=4,isa1:1:4:ws=4:xcd=0,isa1:1:4:ws=1" > $out/ws_n1m.txt 2>&1
cat $out/ws_n1m.txt
for cfg in "ws1_sub8 --wsplit 1 --jsub 8" "ws4_sub8 --wsplit 4 --jsub 8" "ws4_sub4 --wsplit 4 --jsub 4" "ws4_sub2 --wsplit 4 --jsub 2"; do
  set -- $cfg; tag=$1; shift
  echo "== traffic $tag"; timeout -k 10 200 tools/pmc_traffic.sh $tag "$@" 2>&1 | tail -12
done
