"""tools/results_table.py: the north_star's one table (absolute rate, % roofline, the 1/2/4/8 curve, host CPU) generated from the
driver's records.  CPU only: the committed BENCH_r01..r03 / SCALE_r01..r03 records, plus a synthetic SCALE record of the shape a
multi-GPU run will leave (headline lines for N = 1, 2, 4, 8; the N = 8 line carrying the extras)."""
import importlib.util
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("results_table", os.path.join(ROOT, "tools", "results_table.py"))
rt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rt)


def line(n, value, **kw):
    d = {"metric": "billion pair-interactions/s at N=1M fp32; 1/2/4/8 GPUs + % FP32 roofline", "value": value, "unit": "billion pair-interactions/s",
         "n_gpus": n, "steps": 5, "dtype": "f32", "config": {"workload": "N=1048576 fp32 all-pairs softened gravity, leapfrog kick-drift, dt=0.01, seed 42"},
         "roofline": {"frac": 0.59, "frac_of_issue_bound": 0.885}, "cpu_baseline": {"value": 60.0, "cores": 256}}
    if n > 1:
        d.update({"comm_exposed_ms_per_step": 0.02, "transport_used": "rccl", "fallback_from": None})
    d.update(kw)
    return d


def test_committed_records_give_one_row_each():
    recs = rt.records(ROOT)
    names = [r[0] for r in recs]
    for k in (1, 2, 3):
        assert "BENCH_r%02d.json" % k in names and "SCALE_r%02d.json" % k in names
    by = {r[0]: r for r in recs}
    assert len(by["BENCH_r03.json"][2]) == 1 and by["BENCH_r03.json"][2][0]["value"] == 4614.52
    assert by["SCALE_r03.json"][2] is None and "8-GPU" in by["SCALE_r03.json"][3]
    t = rt.table(ROOT)
    assert "| BENCH_r03.json | 1048576 | fp32 | 1 | 4615 | 58.7 | 88.0 |" in t and "55.6 (256 threads)" in t
    assert "| SCALE_r03.json | — | — | — | not measured:" in t


def test_scaling_record_with_extras(tmp_path):
    for f in ("BENCH_r03.json",):
        shutil.copy(os.path.join(ROOT, f), tmp_path / f)
    extras = {"comm_forms": {"ring": {"ms_per_step": 31.2, "comm_exposed_ms_per_step": 1.25, "value": 35240.0, "overlap": 2},
                             "direct": {"ms_per_step": 30.1, "comm_exposed_ms_per_step": 0.03, "value": 36530.0, "overlap": 1},
                             "allgather": {"ms_per_step": 30.2, "comm_exposed_ms_per_step": 0.04, "value": 36400.0, "overlap": 1}},
              "config5": {"workload": "N=4194304 fp64 over 8 GPU(s), 2 timed steps", "value": 14900.0, "ms_per_step": 1180.0,
                          "roofline": {"frac": 0.474, "frac_of_issue_bound": 0.95}, "hbm_gb_per_s": 0.057, "hbm_frac_of_peak": 7.1e-6, "comm_exposed_ms_per_step": 0.1}}
    strict = {"strict_mode": {"arith": "NBODY_ARITH_STRICT", "value": 2844.0, "steps": 3, "ms_per_step": 386.6}}
    scale = {"runs": [{"n": 1, "parsed": line(1, 4600.0, **strict)}, {"n": 2, "parsed": line(2, 9100.0)}, {"n": 4, "parsed": line(4, 18000.0)},
                      {"n": 8, "parsed": line(8, 35500.0, **extras), "tail": json.dumps(line(8, 35500.0)) + "\n"}]}
    json.dump(scale, open(tmp_path / "SCALE_r04.json", "w"))
    t = rt.table(str(tmp_path))
    rows = [r for r in t.splitlines() if r.startswith("| SCALE_r04.json")]
    assert len(rows) == 4 + 1 + 3 + 1                                 # four headline lines (the raw copy of N = 8 de-duplicated), strict_mode, three forms, config 5
    assert "| 1 | 2844 | 36.2 |" in t and "strict_mode: NBODY_ARITH_STRICT, bit-identical to the CPU oracle (3 steps" in t
    assert "| 8 | 35500 | 59.0 | 88.5 | 0.965 | 0.020 | rccl |" in t     # 35500 / (8 x 4600)
    assert "| 2 | 9100 | 59.0 | 88.5 | 0.989 |" in t
    assert "extras: NBODY_COMM_RING, overlap 2, 31.20 ms/step" in t and "| 1.250 |" in t
    assert "| 4194304 | fp64 | 8 | 14900 | 47.4 | 95.0 |" in t and "BASELINE configs[4]" in t
    # a fallback is visible in the table: a peer-copy number cannot pass for the RCCL point
    json.dump({"parsed": line(8, 30000.0, transport_used="peer", fallback_from="rccl", extras="timed out 90 s after the headline line")},
              open(tmp_path / "SCALE_r05.json", "w"))
    t = rt.table(str(tmp_path))
    assert "peer (fell back from rccl)" in t and "extras: timed out 90 s" in t


def test_baseline_md_block_is_generated():
    """BASELINE.md §5's table is the tool's output for the committed records (regenerate with --write after a round's records land)"""
    s = open(os.path.join(ROOT, "BASELINE.md")).read()
    assert rt.BEGIN in s and rt.END in s
    block = s[s.index(rt.BEGIN) + len(rt.BEGIN):s.index(rt.END)].strip()
    rows = [r for r in block.splitlines() if r.startswith("| ")]
    have = {r.split("|")[1].strip() for r in rows}
    want = {r[0] for r in rt.records(ROOT)}
    assert want <= have | {n for n in want if int(n.split("_r")[1][:2]) > 3}    # records newer than the commit are the driver's
    assert {"1", "2", "4", "8"} <= have                                          # the short form: one row per GPU count
    for r in rows:
        name = r.split("|")[1].strip()
        if name in ("BENCH_r01.json", "BENCH_r02.json", "BENCH_r03.json"):
            assert r in rt.table(ROOT)


def test_latest_table_picks_the_newest_line_per_gpu_count(tmp_path):
    json.dump({"parsed": line(1, 4600.0)}, open(tmp_path / "BENCH_r03.json", "w"))
    json.dump({"parsed": line(1, 4700.0)}, open(tmp_path / "BENCH_r04.json", "w"))
    json.dump({"runs": [{"parsed": line(1, 4710.0)}, {"parsed": line(2, 9300.0)}, {"parsed": line(8, 36000.0)}]}, open(tmp_path / "SCALE_r04.json", "w"))
    t = rt.latest_table(str(tmp_path)).splitlines()
    assert t[2].startswith("| 1 | SCALE_r04.json | 4710 |") and t[3].startswith("| 2 | SCALE_r04.json | 9300 | 59.0 | 0.987 |")
    assert "not measured yet" in t[4] and t[5].startswith("| 8 | SCALE_r04.json | 36000 | 59.0 | 0.955 |")


def test_kernel_source_sha_ignores_the_diagnostic_build(tmp_path, monkeypatch):
    """bench.kernel_source_sha() keys the committed profiles to the PRODUCT kernel source: text under `#ifdef NBODY_DIAG_LOOPS` (the
    experiment encodings of `make diag`) is left out, the product's side of an #else is not, and any other change moves the hash"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "mini_nbody_amd" / "csrc"
    d.mkdir(parents=True)
    base = {"nbody_kernels.hpp": "int a;\n#ifdef NBODY_DIAG_LOOPS\nint diag1;\n#if X\nint nested;\n#endif\n#endif\nint b;\n",
            "nbody_args.hpp": "struct ForceArgs { int n; };\n",
            "force_loop_gfx950.inc": "#define P 1\n#ifdef NBODY_DIAG_LOOPS\n#define V19 2\n#endif  // NBODY_DIAG_LOOPS\n",
            "kernels.hip": "#ifdef NBODY_DIAG_LOOPS\nreturn 1;\n#else\nreturn 0;\n#endif\n#ifndef NBODY_DIAG_LOOPS\nint product_only;\n#endif\n"}

    def sha(files):
        for k, v in files.items():
            (d / k).write_text(v)
        monkeypatch.setattr(bench, "ROOT", str(tmp_path))
        return bench.kernel_source_sha()

    h0 = sha(base)
    assert sha(dict(base, **{"nbody_kernels.hpp": base["nbody_kernels.hpp"].replace("int diag1;", "int diag1; int diag2;").replace("int nested;", "")})) == h0
    assert sha(dict(base, **{"force_loop_gfx950.inc": base["force_loop_gfx950.inc"].replace("#define V19 2", "#define V19 2\n#define V20 3")})) == h0
    assert sha(dict(base, **{"kernels.hip": base["kernels.hip"].replace("return 1;", "return 2;")})) == h0
    assert sha(dict(base, **{"kernels.hip": base["kernels.hip"].replace("return 0;", "return 3;")})) != h0          # the product's side of the #else
    assert sha(dict(base, **{"kernels.hip": base["kernels.hip"].replace("int product_only;", "int product_only2;")})) != h0   # #ifndef: product code
    assert sha(dict(base, **{"nbody_kernels.hpp": base["nbody_kernels.hpp"].replace("int b;", "int c;")})) != h0


def test_one_gpu_line_with_the_other_baseline_configs(tmp_path):
    """round 5: the driver's N = 1 line also carries `configs` (BASELINE configs 1 and 2 and the fp64 arithmetic, measured after the headline);
    an N > 1 line carries a numeric cpu_baseline of its own"""
    cpu2 = {"value": 61.25, "unit": "billion pair-interactions/s", "cores": 256, "kind": "port", "sample": "first 65536 of 65536 rows"}
    configs = {"config2": {"workload": "N=65536 fp32, 100 steps, 1 GPU", "value": 4547.0, "ms_per_step": 0.9446, "frac": 0.5781, "cpu_baseline": cpu2},
               "config2_lds_tile256": {"workload": "N=65536 fp32, 100 steps, 1 GPU", "value": 4306.0, "ms_per_step": 0.9974, "frac": 0.5475, "cpu_baseline": cpu2,
                                       "kernel": {"variant": "lds", "tile": 256, "iblock": 4, "nseg": 32}},
               "fp64": {"workload": "N=262144 fp64, 3 steps, 1 GPU", "value": 1830.0, "ms_per_step": 37.55, "frac": 0.4656, "frac_of_issue_bound": 0.93,
                        "cpu_baseline": {"value": 30.5, "cores": 256}},
               "config1": {"workload": "N=4096 fp32, 10 iterations (the first is warm-up), one sequential sum per body", "value": 0.82, "ms_per_step": 20.4,
                           "cpu_value": 1.913, "checksums_equal": True},
               "seconds": 4.2}
    json.dump({"parsed": line(1, 4700.0, configs=configs)}, open(tmp_path / "BENCH_r05.json", "w"))
    json.dump({"runs": [{"parsed": line(1, 4700.0)}, {"parsed": line(2, 9300.0, cpu_baseline={"value": 58.2, "cores": 256})}]}, open(tmp_path / "SCALE_r05.json", "w"))
    t = rt.table(str(tmp_path))
    assert "| BENCH_r05.json | 65536 | fp32 | 1 | 4547 | 57.8 |" in t and "BASELINE configs[1], the engine's default kernel, 0.945 ms/step" in t
    assert "| BENCH_r05.json | 65536 | fp32 | 1 | 4306 | 54.8 |" in t and "tile 256; 4 bodies per lane, 32 segments" in t
    # round 6: every configs entry fills the host-CPU column from its own cpu_baseline
    assert t.count("| 61.2 (256 threads) |") == 2 and "| 30.5 (256 threads) |" in t
    assert "| BENCH_r05.json | 262144 | fp64 | 1 | 1830 | 46.6 | 93.0 |" in t
    assert "| BENCH_r05.json | 4096 | fp32 | 1 | 0.82 |" in t and "1.913 G pairs/s on the host), checksum lines EQUAL" in t
    assert "| 2 | 9300 | 59.0 | 88.5 | 0.989 | 0.020 | rccl | 58.2 (256 threads) |" in t
