#!/bin/bash
# round 3, GPU call 5: profiles at head (rocprofv3 trace + PMC passes), kernel trace of the C host program at small N, bench line
set -u
out=gpurun_out/r03_run5
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cat $out/bench_default.json
tools/profile.sh fp32_n1048576 > $out/prof_1m.log 2>&1; tail -3 $out/prof_1m.log
tools/profile.sh fp64_n262144 --fp64 --bodies 262144 > $out/prof_fp64.log 2>&1; tail -3 $out/prof_fp64.log
tools/profile.sh fp32_n65536 --bodies 65536 --steps 40 --events inline > $out/prof_65536.log 2>&1; tail -3 $out/prof_65536.log
tools/profile.sh fp32_n16384 --bodies 16384 --steps 100 --events inline > $out/prof_16384.log 2>&1; tail -3 $out/prof_16384.log
for n in 4096 16384 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_n$n -- ./build/nbody $n 513 > $out/ktrace_n$n.txt 2>&1
  tail -2 $out/ktrace_n$n.txt
done
for n in 1024 4096 16384 65536 1048576; do
  it=2001; [ $n -ge 65536 ] && it=301; [ $n -ge 1048576 ] && it=6
  ./build/nbody $n $it | tail -1
done > $out/host_program.txt 2>&1; cat $out/host_program.txt
NBODY_OVERSUBSCRIBE=1 ./build/nbody 1048576 5 --gpus 8 | tail -1
./build/nbody 1048576 5 | tail -1
