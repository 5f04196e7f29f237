#!/bin/bash
# Profiles one bench.py configuration with rocprofv3 on the GPU box: kernel trace + stats, then PMC passes
# (separate runs, never combined with trace domains other than --kernel-trace).  Output under gpurun_out/prof_<tag>/;
# tools/parse_prof.py turns it into profiles/<round>_<tag>_summary.txt and profiles/pmc_<tag>.json.
# usage: tools/profile.sh <tag> [bench.py args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --in-process --configs-pass never --strict-pass never --no-cpu-baseline "$@" > $out/bench_trace.json 2> $out/trace.err
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_COUNT" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS SQ_INSTS_BRANCH" \
  "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py --in-process --configs-pass never --strict-pass never --no-cpu-baseline --steps 2 "$@" > $out/bench_pmc_$i.json 2> $out/pmc_$i.err
  echo "pass $i rc=$? : $set"
done
find $out -name "*.csv" | wc -l
