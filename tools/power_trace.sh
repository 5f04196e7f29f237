#!/bin/bash
# Socket power, shader clock and temperature sampled with rocm-smi while bench.py runs the headline workload (and fp64 N = 262144): the
# direct telemetry behind "the chip is power-limited" (so far inferred from GRBM_GUI_ACTIVE / time).  Run ON the GPU box:
#   bash tools/power_trace.sh gpurun_out/r04/power      -> <prefix>_{idle,fp32,fp64}.txt (raw rocm-smi samples) + <prefix>_{fp32,fp64}.json (bench lines)
# rocm-smi only reads sysfs; it is never put under rocprofv3 and never touches the queue the bench uses.
set -u
P=${1:?prefix}
mkdir -p "$(dirname "$P")"
sample() { rocm-smi --showpower --showmaxpower --showclocks --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" ; echo "--- $(date +%s.%N)"; }
for i in 1 2 3; do sample; sleep 0.5; done > "${P}_idle.txt"
run() {  # tag, bench args...
  tag=$1; shift
  python3 bench.py "$@" --cpu-baseline never > "${P}_${tag}.json" 2> "${P}_${tag}.err" &
  bp=$!
  while kill -0 $bp 2>/dev/null; do sample; sleep 0.4; done > "${P}_${tag}.txt"
  wait $bp
}
run fp32 --steps 60 --warmup 5
run fp64 --fp64 --bodies 262144 --steps 400 --warmup 5
echo "power trace done"
