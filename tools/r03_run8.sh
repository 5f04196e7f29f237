#!/bin/bash
# round 3, GPU call 8: profiles at head after the segment-rule change
set -u
out=gpurun_out/r03_run8
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-400 $out/bench_default.json
tools/profile.sh fp32_n1048576 > $out/prof_1m.log 2>&1; tail -1 $out/prof_1m.log
tools/profile.sh fp64_n262144 --fp64 --bodies 262144 > $out/prof_fp64.log 2>&1; tail -1 $out/prof_fp64.log
tools/profile.sh fp32_n65536 --bodies 65536 --steps 40 --events inline > $out/prof_65536.log 2>&1; tail -1 $out/prof_65536.log
tools/profile.sh fp32_n16384 --bodies 16384 --steps 100 --events inline > $out/prof_16384.log 2>&1; tail -1 $out/prof_16384.log
timeout -k 10 500 tools/profile.sh fp64_n4194304 --fp64 --bodies 4194304 --steps 1 --warmup 1 > $out/prof_4m.log 2>&1; tail -1 $out/prof_4m.log
python3 bench.py --fp64 --bodies 262144 > $out/bench_fp64_n262144.json 2> $out/bench_fp64.err; cut -c1-300 $out/bench_fp64_n262144.json
NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus 8 | tail -1
./build/nbody 1048576 5 | tail -1
