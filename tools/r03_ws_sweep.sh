#!/bin/bash
# wave split (NBODY_OPT_WSPLIT) against round 2's layout, wall clock per step on the graph path, at the sizes the verdict names
set -u
out=gpurun_out/r03_ws
mkdir -p $out
for n in 4096 8192 16384 32768 65536 131072; do
  case $n in
    4096)  cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:0:ws=4:fuse=1,isa1:1:4:ws=4,isa1:1:4:ws=4:fuse=1,isa1:1:16:ws=4,isa1:1:16:ws=4:fuse=1"; steps=4000;;
    8192)  cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:0:ws=4:fuse=1,isa1:1:8:ws=4,isa1:1:8:ws=4:fuse=1"; steps=3000;;
    16384) cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:8:ws=4,isa1:1:32:ws=4,isa1:1:16:ws=4:long=0,isa1:1:16:ws=4:long=1,isa1:1:16:ws=4:fuse=0"; steps=2000;;
    32768) cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:8:ws=4,isa1:1:32:ws=4"; steps=800;;
    65536) cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:8:ws=4,isa1:1:32:ws=4,isa1:1:64:ws=4"; steps=300;;
    131072) cfgs="isa1:1:0:ws=1,isa1:1:0:ws=4,isa1:1:8:ws=4,isa1:1:32:ws=4"; steps=80;;
  esac
  python3 tools/sweep.py --wall --n $n --steps $steps --rounds 3 --configs "$cfgs" > $out/n$n.txt 2>&1
  echo "== n=$n"; cat $out/n$n.txt
done
