#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time from the kernel trace, PMC counters of the
force kernel averaged per launch, and the derived figures DESIGN.md quotes — clock, cycles per wave-pair (needs no
assumption: GRBM cycles x SIMDs / wave-pairs), the wave-lifetime breakdown, HBM-side bytes.  Writes the JSON that
bench.py attaches to a run of the SAME configuration (`config` is taken from the profiled bench.py line).
usage: python tools/parse_prof.py gpurun_out/prof_<tag> [--json profiles/pmc_<tag>.json]"""
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
    bench = None
    try:
        txt = [l for l in open(os.path.join(d, "bench_trace.json")).read().splitlines() if l.startswith("{")]
        bench = json.loads(txt[-1])
    except Exception:
        pass
    # gpurun merges a call's files into what an earlier call left in the same directory: keep the newest file per pass
    stats = sorted(glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    force_avg_ns = None
    force_name = None
    if stats:
        lines.append("## rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline ...")
        for r in csv.DictReader(open(stats[0])):
            lines.append("%-70s calls=%-3s avg=%12.3f us  total=%5.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
            if "force_" in r["Name"] and force_avg_ns is None:
                force_avg_ns, force_name = float(r["AverageNs"]), r["Name"]
    if bench:
        lines.append("bench.py line of that run: value %.1f %s, ms_per_step %.3f, kernel_ms_avg (HIP events) %.4f, config %s"
                     % (bench["value"], bench["unit"], bench["ms_per_step"], bench["roofline"]["kernel_ms_avg"], json.dumps(bench["config"]["kernel"])))
    counters = defaultdict(list)
    durs = []
    meta = {}
    newest = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        key = f.split(os.sep)[-3]
        if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
            newest[key] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            if "force_" not in r["Kernel_Name"]:
                continue
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
                durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            meta = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"], kernel=r["Kernel_Name"])
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    lines.append("")
    lines.append("## PMC counters of %s, mean per launch (separate --pmc passes, bench.py --steps 2)" % meta.get("kernel", "?"))
    lines.append("launch: grid=%s wg=%s VGPR=%s SGPR=%s LDS=%s" % (meta.get("grid"), meta.get("wg"), meta.get("vgpr"), meta.get("sgpr"), meta.get("lds")))
    for k in sorted(avg):
        lines.append("%-28s %20.1f   (%d launches)" % (k, avg[k], len(counters[k])))
    derived = {}
    cfg = None
    wave_pairs = None
    if bench:
        k = bench["config"]["kernel"]
        cfg = {"n": bench["config"]["n_bodies"], "dtype": bench["dtype"], "n_gpus": bench["n_gpus"], "variant": k["variant"], "iblock": k["iblock"],
               "nseg": k["nseg"], "sum_order": k["sum_order"], "sum_block": k["sum_block"], "launches_per_step": k["launches_per_step"],
               "wsplit": k.get("wsplit", 1), "isa_phase": k.get("isa_phase", 1), "long_buffers": k.get("long_buffers", -1), "xcd_map": k.get("xcd_map", -1),
               "kernel_source_sha": bench["config"].get("kernel_source_sha")}
        if bench["config"].get("arith", "fma3") != "fma3":
            cfg["arith"] = bench["config"]["arith"]        # bench.py --arith: a study run (absent = the timed arithmetic)
        wave_pairs = float(k["n_local"]) * bench["config"]["n_bodies"] / 64.0
        # The profiled run recorded its hash with the bench.py it carried.  Until round 4 that was a hash of the three kernel files whole;
        # now bench.kernel_source_sha() leaves the `#ifdef NBODY_DIAG_LOOPS` text out.  If the tree here still holds exactly the files that
        # run was made from (their whole-file hash equals the recorded one), store the hash in today's definition instead.
        try:
            import hashlib
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            h = hashlib.sha1()
            for f in ("nbody_kernels.hpp", "nbody_args.hpp", "force_loop_gfx950.inc", "kernels.hip"):
                h.update(open(os.path.join(root, "mini_nbody_amd", "csrc", f), "rb").read())
            if h.hexdigest()[:12] == cfg["kernel_source_sha"]:
                sys.path.insert(0, root)
                import bench as _bench
                cfg["kernel_source_sha"] = _bench.kernel_source_sha()
        except Exception:
            pass
    if durs and "GRBM_GUI_ACTIVE" in avg:
        t = sum(durs) / len(durs) * 1e-9
        cycles = avg["GRBM_GUI_ACTIVE"] / 8.0            # the counter is summed over the 8 XCDs
        derived["kernel_s_profiled"] = t
        derived["clock_ghz"] = cycles / t / 1e9
        if wave_pairs:
            derived["cycles_per_wave_pair"] = cycles * 1024.0 / wave_pairs      # 1024 SIMDs
    if "SQ_ACTIVE_INST_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        # rocprof's classic VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE; the counter is summed over the waves
        # whose VALU instruction is in the pipe, so it passes 1 when instructions of several waves overlap there
        derived["sq_valu_busy"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0)
    if avg.get("SQ_WAVE_CYCLES"):
        wc = avg["SQ_WAVE_CYCLES"]
        # SQ_* cycle counters are in quad-cycles summed over waves; MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
            if name in avg:
                derived["wave_lifetime_frac_" + name[3:].lower()] = avg[name] / wc
        if wave_pairs:
            derived["wave_cycles_per_wave_pair"] = wc * 4.0 / wave_pairs
    if wave_pairs:
        for name, key in (("SQ_INSTS_VALU", "valu_insts_per_wave_pair"), ("SQ_INSTS_SALU", "salu_insts_per_wave_pair"),
                          ("SQ_INSTS_SMEM", "smem_insts_per_wave_pair"), ("SQ_INSTS_VALU_TRANS_F32", "trans_f32_insts_per_wave_pair"),
                          ("SQ_INSTS_VALU_TRANS_F64", "trans_f64_insts_per_wave_pair"), ("SQ_IFETCH", "ifetch_per_wave_pair")):
            if name in avg:
                derived[key] = avg[name] / wave_pairs
        if "SQ_THREAD_CYCLES_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
            derived["thread_cycles_valu_per_wave_pair_div64"] = avg["SQ_THREAD_CYCLES_VALU"] / 64.0 / wave_pairs
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
        lines.append("%-42s %s" % (k, ("%.6g" % v) if isinstance(v, float) else v))
    if force_avg_ns:
        lines.append("%-42s %.3f ms (%s)" % ("force kernel avg (trace)", force_avg_ns / 1e6, force_name))
    print("\n".join(lines))
    if out_json:
        json.dump({"source": d, "config": cfg, "kernel": meta.get("kernel"), "launch": meta, "counters_mean_per_launch": avg, **derived,
                   "force_kernel_avg_ms_trace": force_avg_ns / 1e6 if force_avg_ns else None,
                   "note": "FETCH_SIZE/WRITE_SIZE are L2 memory-side (fabric) request bytes; Infinity-Cache hits are counted too "
                           "(MI355X_MICROARCH.md §HBM); FETCH_SIZE doubled as that section prescribes for 64-B scalar / 16-B-per-lane reads."},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
