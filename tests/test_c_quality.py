"""Build-level checks that need no GPU: the C-ABI header is plain C99 usable from a C translation unit as it stands, and
the CPU-side C code (oracle, IC generator, CPU program) runs clean under AddressSanitizer + UBSan (sanitizers run on the
CPU build only; the GPU pool has none)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_strict_c99(tmp_path):
    src = tmp_path / "use_header.c"
    src.write_text('#include "nbody.h"\n#include "nbody_ic.h"\n'
                   "int use(void) { BodySystem b = {0, 0}; BodySystemD d = {0, 0}; (void)b; (void)d;\n"
                   "  return NBODY_OK + NBODY_SUM_BLOCKED + NBODY_COMM_DIRECT + NBODY_INFO_HAS_COMM + (int)sizeof(nbody_host_gather_fn); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("prog", ["nbody.c", "mailbox_driver.c"])
def test_host_programs_are_warning_free_c11(prog):
    """the two C host programs above the C-ABI (the north_star's bodyForce()/integrate() program and the PS-side mailbox driver of
    INTEGRATION.md §1) compile as strict C11 with every warning an error (syntax only: no GPU, no library needed)"""
    r = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "mini_nbody_amd", "host", prog)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_cpu_code_is_sanitizer_clean(tmp_path):
    exe = tmp_path / "nbody_cpu_san"
    flags = ["-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
             "-ffp-contract=off", "-fopenmp"]
    r = subprocess.run(["gcc"] + flags + ["-o", str(exe), os.path.join(ROOT, "oracle", "nbody_cpu.c"), os.path.join(ROOT, "oracle", "nbody_ref.c"), "-lm"],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2")
    for args in (["257", "3"], ["300", "3", "--sum", "blocked", "--block", "64", "--segments", "5"], ["64", "2", "--fp64"], ["1", "2"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "checksum" in out.stdout
