"""Every repository path the documents cite exists: DESIGN.md, BASELINE.md, README.md, INTEGRATION.md and profiles/README.md name
their evidence by file (`profiles/...`, `tests/...`, `tools/...`, `mini_nbody_amd/...`, `oracle/...`, `include/...`); a stale name is
a claim without its evidence."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "BASELINE.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]
PREFIXES = ("profiles/", "tests/", "tools/", "mini_nbody_amd/", "oracle/", "include/")
BUILT = ("mini_nbody_amd/libnbody_hip.so", "mini_nbody_amd/libnbody_hip_diag.so", "oracle/libnbody_ref", "oracle/nbody_cpu", "oracle/_ref",
         "mini_nbody_amd/csrc/microbench_streams.inc", "mini_nbody_amd/csrc/force_loop_mfma_gfx950.inc")     # made by `make`, git-ignored


def cited_paths(text, doc):
    out = set()
    for tok in re.findall(r"`([^`\n]+)`", text):
        tok = tok.strip()
        if doc.startswith("profiles") and re.match(r"^(r0\d_|pmc_)[\w.*…{}-]+$", tok):
            tok = "profiles/" + tok                    # profiles/README.md names its own files without the directory
        if not tok.startswith(PREFIXES):
            continue
        tok = tok.split("::")[0].split(" ")[0].rstrip(".,;:)")
        if "…" in tok or "<" in tok or "{" in tok or "$" in tok or tok.endswith("/"):
            continue                                   # elided or templated names
        out.add(tok)
    return out


def test_cited_paths_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for tok in sorted(cited_paths(text, doc)):
            if tok.startswith(BUILT):
                continue
            if not glob.glob(os.path.join(ROOT, tok)):
                missing.append((doc, tok))
    assert not missing, missing


def test_every_profile_of_the_current_round_is_indexed():
    """... and the other way round for the evidence directory: every file of the newest round under profiles/ (and every pmc_*.json) has
    its row in profiles/README.md (a `…_summary.txt` / `…_kernel_stats.csv` pair is indexed by its common stem)."""
    index = open(os.path.join(ROOT, "profiles", "README.md")).read()
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "*")))
    rounds = sorted({m.group(1) for f in files for m in [re.match(r"(r\d\d)_", f)] if m})
    newest = rounds[-1]
    missing = []
    for f in files:
        if not (f.startswith(newest + "_") or f.startswith("pmc_")):
            continue
        stem = re.sub(r"_(summary\.txt|kernel_stats\.csv)$", "", f)      # r04_rocprofv3_fp32_n65536 is indexed as `r04_rocprofv3_fp32_n65536_*`
        if not (f in index or stem in index or (f.startswith("pmc_") and "pmc_*.json" in index)):
            missing.append(f)
    assert not missing, missing


def test_design_prose_stays_within_120_columns():
    """VERDICT r04 housekeeping: DESIGN.md's prose is wrapped at 120 columns (tables, headings and code fences cannot be);
    `python tools/wrap_md.py DESIGN.md` re-flows it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wrap_md", os.path.join(ROOT, "tools", "wrap_md.py"))
    wm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wm)
    for doc in ("DESIGN.md", os.path.join("profiles", "README.md")):
        src = open(os.path.join(ROOT, doc)).read()
        assert wm.too_long(src, 120) == [], doc
        if doc == "DESIGN.md":
            assert wm.wrap(src, 120) == src          # re-flowing it changes nothing: it IS the tool's output
