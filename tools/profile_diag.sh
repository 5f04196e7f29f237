#!/bin/bash
# Cycles per wave-pair of the product loop and of its timing-only diagnostic forms (NBODY_OPT_ISA_PHASE 1, 3, 4, 5),
# each under rocprofv3 --pmc GRBM_GUI_ACTIVE so that the clock each form actually runs at is known.
# The diagnostic forms are not in the product library: `make diag` first (here, before gpurun: built .so files travel);
# this script points the package at libnbody_hip_diag.so through NBODY_LIB.
# usage: tools/profile_diag.sh  (on the GPU box; output gpurun_out/prof_diag/)
set -u
export NBODY_LIB="${NBODY_LIB:-$PWD/mini_nbody_amd/libnbody_hip_diag.so}"
[ -f "$NBODY_LIB" ] || { echo "missing $NBODY_LIB: run make diag"; exit 1; }
export PHASES="${PHASES:-1 3 4 5 2 0}"
out=gpurun_out/prof_diag
mkdir -p $out
export TMPDIR=/tmp
for ph in $PHASES; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/ph$ph -- python3 bench.py --in-process --configs-pass never --strict-pass never --no-cpu-baseline --steps 3 --isa-phase $ph > $out/bench_ph$ph.json 2> $out/ph$ph.err
  echo "phase $ph rc=$?"
done
python3 - <<'PY'
import csv, glob, json
print("| loop form | kernel ms | clock GHz | SIMD cycles per wave-pair | wave lifetime: issuing VALU / waiting turn / parked |")
print("|---|---|---|---|---|")
names = {9: "e32 subtractions + s_nop, eps in a VGPR", 10: "e32 subtractions, no filler", 11: "all 32-bit encodings", 12: "product loop with eps in a VGPR", 13: "e32 subtractions + s_nop", 14: "TIMING-ONLY: coordinates from VGPRs, transcendental kept", 15: "TIMING-ONLY: the same + one broadcast ds_read_b128 per source", 6: "12 independent v_fma_f32", 7: "pair interaction in 32-bit encodings, no transcendental", 8: "source coordinates from VGPRs, no transcendental", 1: "product loop", 3: "v_rsq_f32 -> v_mov_b32 (no transcendental)", 4: "loop's s_load -> s_nop (no scalar loads)", 5: "neither", 2: "staggered s_load_dwordx8", 0: "product loop one 4-byte phase off"}
import os
for ph in [int(x) for x in os.environ.get("PHASES", "1 3 4 5 2 0").split()]:
    rows = []
    for f in glob.glob("gpurun_out/prof_diag/ph%d/*/*_counter_collection.csv" % ph):
        rows += [r for r in csv.DictReader(open(f)) if "force_" in r["Kernel_Name"]]
    c = {}
    for r in rows:
        c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    if not dur:
        print("| %s | no data | | | |" % names[ph]); continue
    m = lambda k: sum(c[k]) / len(c[k])
    t = sum(dur) / len(dur) * 1e-9
    cyc = m("GRBM_GUI_ACTIVE") / 8.0
    wp = (1 << 20) * float(1 << 20) / 64.0
    wc = m("SQ_WAVE_CYCLES")
    print("| %s | %.2f | %.3f | %.2f | %.1f %% / %.1f %% / %.1f %% |" % (names[ph], t * 1e3, cyc / t / 1e9, cyc * 1024.0 / wp,
          100 * m("SQ_ACTIVE_INST_VALU") / wc, 100 * m("SQ_WAIT_INST_ANY") / wc, 100 * m("SQ_WAIT_ANY") / wc))
PY
