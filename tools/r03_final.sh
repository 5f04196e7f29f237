#!/bin/bash
# round 3, final GPU call: profiles at head (rocprofv3 trace + PMC passes for the five configurations), then the bench lines
set -u
out=gpurun_out/r03_final
mkdir -p $out
export TMPDIR=/tmp
tools/profile.sh fp32_n1048576 > $out/prof_1m.log 2>&1; tail -1 $out/prof_1m.log
tools/profile.sh fp64_n262144 --fp64 --bodies 262144 > $out/prof_fp64.log 2>&1; tail -1 $out/prof_fp64.log
tools/profile.sh fp32_n65536 --bodies 65536 --steps 40 --events inline > $out/prof_65536.log 2>&1; tail -1 $out/prof_65536.log
tools/profile.sh fp32_n16384 --bodies 16384 --steps 100 --events inline > $out/prof_16384.log 2>&1; tail -1 $out/prof_16384.log
timeout -k 10 500 tools/profile.sh fp64_n4194304 --fp64 --bodies 4194304 --steps 1 --warmup 1 > $out/prof_4m.log 2>&1; tail -1 $out/prof_4m.log
