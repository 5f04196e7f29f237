#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) of one bench.py configuration, two PMC passes.
# usage: tools/pmc_traffic.sh <tag> [bench.py args...]   -> gpurun_out/traffic_<tag>/
set -u
tag=$1; shift
out=gpurun_out/traffic_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_5 -- python3 bench.py --in-process --configs-pass never --strict-pass never --no-cpu-baseline --steps 2 "$@" > $out/bench_trace.json 2> $out/p5.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_6 -- python3 bench.py --in-process --configs-pass never --strict-pass never --no-cpu-baseline --steps 2 "$@" > $out/bench6.json 2> $out/p6.err
python3 tools/parse_prof.py $out | grep -E "fetch_bytes|write_bytes|hbm_bytes|l2_hit|FETCH_SIZE|WRITE_SIZE|TCC"
