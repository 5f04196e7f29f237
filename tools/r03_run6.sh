#!/bin/bash
set -u
out=gpurun_out/r03_run6
mkdir -p $out
export TMPDIR=/tmp
# torch.distributed.run path on one GPU: every rank process supervises its own worker
NBODY_OVERSUBSCRIBE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --bodies 262144 --steps 3 --no-cpu-baseline > $out/torchrun_2.json 2> $out/torchrun_2.err
echo "torchrun rc=$?"; grep '^{' $out/torchrun_2.json | cut -c1-900; tail -3 $out/torchrun_2.err
# fp64: 4 against 8 global segments (2 MiB segments fit an XCD's L2)
timeout -k 10 200 python3 tools/sweep.py --fp64 --n 262144 --steps 2 --rounds 3 --configs "isa1:1:8:ws=4,isa1:1:4:ws=4,isa1:1:4:ws=4:xcd=1,isa1:1:16:ws=4" > $out/fp64_sub.txt 2>&1; cat $out/fp64_sub.txt
for cfg in "f64_sub4 --fp64 --bodies 262144 --jsub 4" "f64_sub4x --fp64 --bodies 262144 --jsub 4 --xcd-map 1" "f64_sub8 --fp64 --bodies 262144 --jsub 8"; do
  set -- $cfg; tag=$1; shift
  echo "== traffic $tag"; timeout -k 10 200 tools/pmc_traffic.sh $tag "$@" 2>&1 | tail -6
done
# config 5's size on one GPU
timeout -k 10 500 tools/profile.sh fp64_n4194304 --fp64 --bodies 4194304 --steps 1 --warmup 1 > $out/prof_4m.log 2>&1; tail -3 $out/prof_4m.log
