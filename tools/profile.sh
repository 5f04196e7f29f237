#!/bin/bash
# Profiles one bench.py configuration with rocprofv3 on the GPU box: kernel trace + stats, then PMC passes
# (separate runs, never combined with trace domains other than --kernel-trace).  Output under gpurun_out/prof_<tag>/.
# usage: tools/profile.sh <tag> [bench.py args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_trace.json 2> $out/trace.err
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --no-cpu-baseline --steps 2 "$@" > $out/bench_pmc_$name.json 2> $out/pmc_$name.err
done
find $out -name "*.csv" | head -40
