"""The one-GPU bench line end to end on the GPU box (a small size: the line's shape is what is checked, the headline is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_gpu_line_carries_roofline_cpu_baseline_and_strict_mode():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "65536", "--steps", "8", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["dtype"] == "f32" and j["value"] > 1000 and j["config"]["finite"]
    roof, cpu, strict = j["roofline"], j["cpu_baseline"], j["strict_mode"]
    assert roof["bound"] == "valu" and 0.3 < roof["frac"] < 0.7 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    # the bit-exact arithmetic on the same box, after the timed region: slower than the timed mode, far faster than rounds 1-3's 0.19 of it
    assert strict["arith"] == "NBODY_ARITH_STRICT" and strict["kernel"]["variant"] == "smem"
    assert 0.4 * j["value"] < strict["value"] < 0.9 * j["value"], (strict, j["value"])
    # ... and switched off on request, and absent from an fp64 line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "16384", "--steps", "4", "--strict-pass", "never", "--cpu-baseline", "never"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "strict_mode" not in json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_default_line_carries_the_other_baseline_configs():
    """the driver's own N = 1 invocation (default bodies; few steps here) yields ONE line whose `configs` hold BASELINE
    configs 1 and 2 and the fp64 arithmetic, measured after the headline line was out; `value` is untouched by them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["config"]["n_bodies"] == 1 << 20 and j["value"] > 3000 and "extras" not in j, j.get("extras")
    assert j["cpu_baseline"]["value"] > 0 and j["strict_mode"]["value"] > 0
    c = j["configs"]
    assert set(c) >= {"config2", "config2_lds_tile256", "fp64", "config1", "seconds"} and c["seconds"] < 60
    # every entry carries its own measured host-CPU column, and the LDS entry runs the sweep's best LDS form and names it
    for key in ("config2", "config2_lds_tile256", "fp64", "config1"):
        cb = c[key]["cpu_baseline"]
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"], (key, cb)
    assert c["config2_lds_tile256"]["kernel"]["iblock"] == 4 and c["config2_lds_tile256"]["kernel"]["nseg"] == 32 and "best measured" in c["config2_lds_tile256"]["form"]
    for key in ("config2", "config2_lds_tile256"):
        assert c[key]["value"] > 2000 and 0.2 < c[key]["frac"] < 0.7 and c[key]["ms_per_step"] < 3, (key, c[key])
    assert c["config2"]["kernel"]["variant"] == "isa" and c["config2_lds_tile256"]["kernel"]["variant"] == "lds" and c["config2_lds_tile256"]["kernel"]["tile"] == 256
    assert c["fp64"]["value"] > 500 and c["fp64"]["peak_tflops"] == 78.6 and 0.5 < c["fp64"]["frac_of_issue_bound"] <= 1.05, c["fp64"]
    assert c["config1"]["checksums_equal"] is True and c["config1"]["cpu_value"] > 0 and c["config1"]["value"] > 0, c["config1"]
