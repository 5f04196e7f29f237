#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time from the kernel trace, PMC counters of the
force kernel averaged per launch, and the derived figures DESIGN.md quotes (clock, VALU busy, HBM bytes).
usage: python tools/parse_prof.py gpurun_out/prof_<tag> [--json profiles/latest_pmc.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    lines = []
    stats = glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))
    force_avg_ns = None
    force_name = None
    if stats:
        lines.append("## rocprofv3 --kernel-trace --stats (bench.py, default steps)")
        for r in csv.DictReader(open(stats[0])):
            lines.append("%-70s calls=%-3s avg=%12.3f us  total=%5.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
            if "force_" in r["Name"] and force_avg_ns is None:
                force_avg_ns, force_name = float(r["AverageNs"]), r["Name"]
    counters = defaultdict(list)
    durs = []
    meta = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "force_" not in r["Kernel_Name"]:
                continue
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVES"):
                durs.append((r["Counter_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
            meta = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"], kernel=r["Kernel_Name"])
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    lines.append("")
    lines.append("## PMC counters of %s, mean per launch (separate --pmc passes)" % meta.get("kernel", "?"))
    lines.append("launch: grid=%s wg=%s VGPR=%s SGPR=%s LDS=%s" % (meta.get("grid"), meta.get("wg"), meta.get("vgpr"), meta.get("sgpr"), meta.get("lds")))
    for k in sorted(avg):
        lines.append("%-28s %20.1f   (%d launches)" % (k, avg[k], len(counters[k])))
    derived = {}
    dur_gui = [x for n, x in durs if n == "GRBM_GUI_ACTIVE"]
    if dur_gui and "GRBM_GUI_ACTIVE" in avg:
        t = sum(dur_gui) / len(dur_gui) * 1e-9
        derived["kernel_s_profiled"] = t
        derived["clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9     # counter is summed over the 8 XCDs
    if "SQ_BUSY_CYCLES" in avg and "SQ_ACTIVE_INST_VALU" in avg:
        # SQ_* cycle counters are in quad-cycles summed over SEs/CUs as rocprofv3 reports them; ratios are unit-free
        derived["valu_busy_of_wave_cycles"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"] if avg.get("SQ_WAVE_CYCLES") else None
    if "SQ_ACTIVE_INST_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        # SQ_ACTIVE_INST_VALU: quad-cycles during which a wave has a VALU instruction executing, summed over waves;
        # a CDNA4 SIMD keeps two wave64 VALU instructions in flight (4 cycles each, 2-cycle issue), so 100 % busy is
        # 2 x SIMDs x cycles.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
        cycles = avg["GRBM_GUI_ACTIVE"] / 8.0
        derived["valu_busy_frac"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (2.0 * 1024.0 * cycles)
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg:
        derived["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
    if "FETCH_SIZE" in avg:
        # FETCH_SIZE is in KiB of 64-B requests; gfx950 tallies 128-B requests of wide coalesced reads at 64 B (x2), MI355X_MICROARCH.md §HBM
        derived["fetch_bytes_raw"] = avg["FETCH_SIZE"] * 1024.0
        derived["fetch_bytes_x2"] = avg["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in avg:
        derived["write_bytes"] = avg["WRITE_SIZE"] * 1024.0
    if "fetch_bytes_x2" in derived and "write_bytes" in derived:
        derived["hbm_bytes_per_launch"] = derived["fetch_bytes_x2"] + derived["write_bytes"]
    if "hbm_bytes_per_launch" in derived and "kernel_s_profiled" in derived:
        derived["hbm_gb_per_s"] = derived["hbm_bytes_per_launch"] / derived["kernel_s_profiled"] / 1e9
        derived["hbm_frac_of_8tbs"] = derived["hbm_gb_per_s"] / 8000.0
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        derived["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
    lines.append("")
    lines.append("## derived")
    for k, v in derived.items():
        lines.append("%-28s %s" % (k, ("%.6g" % v) if isinstance(v, float) else v))
    if force_avg_ns:
        lines.append("%-28s %.3f ms (%s)" % ("force kernel avg (trace)", force_avg_ns / 1e6, force_name))
    print("\n".join(lines))
    if out_json:
        json.dump({"source": d, "kernel": meta.get("kernel"), "counters_mean_per_launch": avg, **derived,
                   "force_kernel_avg_ms_trace": force_avg_ns / 1e6 if force_avg_ns else None}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
