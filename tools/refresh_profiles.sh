#!/bin/bash
# The round's rocprofv3 evidence in two steps (what rounds 2-4 did by hand):
#   on the GPU box:   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh box'
#                     the five configurations the documents quote — fp32 N = 1M (BASELINE's headline), 65536, 16384, fp64 N = 262144 and
#                     4194304 — each through tools/profile.sh: one `rocprofv3 --kernel-trace --stats` run of bench.py + six separate --pmc passes
#   here, afterwards: bash tools/refresh_profiles.sh parse r05
#                     tools/parse_prof.py per configuration -> profiles/<round>_rocprofv3_<tag>_summary.txt, _kernel_stats.csv and
#                     profiles/pmc_<tag>.json (what bench.py attaches as roofline.traffic / roofline.pmc when configuration and
#                     product-kernel-source hash match), plus the traced run's own bench line
#   then, so that the committed bench lines carry the new profile: gpurun -- 'bash tools/refresh_profiles.sh lines'  and
#                     bash tools/refresh_profiles.sh keep r05
set -u
TAGS="fp32_n1048576 fp32_n65536 fp32_n16384 fp64_n262144 fp64_n4194304"
args_of() { case $1 in fp32_n1048576) echo "";; fp32_n65536) echo "--bodies 65536";; fp32_n16384) echo "--bodies 16384";;
            fp64_n262144) echo "--fp64 --bodies 262144";; fp64_n4194304) echo "--fp64 --bodies 4194304 --steps 2";; esac; }
case ${1:-} in
  box)
    for t in $TAGS; do bash tools/profile.sh $t $(args_of $t) > gpurun_out/prof_$t.log 2>&1 || exit 1; echo "profiled $t"; done ;;
  parse)
    r=${2:?round tag, e.g. r05}
    for t in $TAGS; do
      python3 tools/parse_prof.py gpurun_out/prof_$t --json profiles/pmc_$t.json > profiles/${r}_rocprofv3_${t}_summary.txt
      cp "$(ls -t gpurun_out/prof_$t/trace/*/*_kernel_stats.csv | head -1)" profiles/${r}_rocprofv3_${t}_kernel_stats.csv
    done
    cp gpurun_out/prof_fp32_n1048576/bench_trace.json profiles/${r}_bench_n1_under_rocprof_trace.json
    grep -H -E "^cycles_per_wave_pair|^clock_ghz|^hbm_bytes_per_launch|force kernel avg" profiles/${r}_rocprofv3_*_summary.txt ;;
  lines)
    mkdir -p gpurun_out/lines
    python3 bench.py > gpurun_out/lines/bench_n1.json 2> gpurun_out/lines/bench_n1.err &&
    python3 bench.py --steps 20 --warmup 5 > gpurun_out/lines/bench_n1_driver_form.json 2>/dev/null &&
    python3 bench.py --fp64 --bodies 262144 > gpurun_out/lines/bench_fp64_n262144.json 2>/dev/null &&
    python3 bench.py --bodies 65536 --steps 200 > gpurun_out/lines/bench_n65536.json 2>/dev/null; echo "lines done" ;;
  keep)
    r=${2:?round tag}
    for f in bench_n1 bench_n1_driver_form bench_fp64_n262144 bench_n65536; do cp gpurun_out/lines/$f.json profiles/${r}_$f.json; done ;;
  *) sed -n 2,13p "$0"; exit 2 ;;
esac
