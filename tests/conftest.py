import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored; on a fresh checkout build them once (hipcc cross-compiles without a GPU).
    A failing build fails the session loudly: there is no fallback to test instead."""
    import subprocess
    need = [os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip.so"), os.path.join(ROOT, "build", "nbody"), os.path.join(ROOT, "build", "mailbox_driver"),
            os.path.join(ROOT, "oracle", "libnbody_ref.so"), os.path.join(ROOT, "oracle", "libnbody_ref_fast.so"),
            os.path.join(ROOT, "oracle", "nbody_cpu"), os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip_diag.so")]
    # ... and again whenever a source is newer than what was built from it: a stale library must not be what the tests (or a gpurun
    # snapshot, which ships the built files) exercise
    src = [os.path.join(ROOT, "mini_nbody_amd", "csrc", f) for f in ("kernels.hip", "context.cpp", "comm.cpp", "mailbox.cpp", "nbody_internal.hpp", "nbody_args.hpp", "nbody_kernels.hpp",
                                                                      "force_loop_gfx950.inc")] + \
          [os.path.join(ROOT, "include", "nbody.h"), os.path.join(ROOT, "oracle", "nbody_ref.c"), os.path.join(ROOT, "oracle", "nbody_ref.h"),
           os.path.join(ROOT, "oracle", "nbody_cpu.c"), os.path.join(ROOT, "mini_nbody_amd", "host", "nbody.c"),
           os.path.join(ROOT, "mini_nbody_amd", "host", "mailbox_driver.c")]
    if all(os.path.exists(p) for p in need) and max(os.path.getmtime(p) for p in src) <= min(os.path.getmtime(p) for p in need):
        return
    r = subprocess.run(["make", "lib", "diag", "host", "oracle"], cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0:
        raise pytest.UsageError("`make lib diag host oracle` failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


@pytest.fixture(scope="session")
def oracle():
    """Serial, bit-reproducible oracle build (test infrastructure)."""
    import oracle as O
    return O.Oracle(fast=False)


@pytest.fixture(scope="session")
def oracle_fast(tmp_path_factory):
    """OpenMP/vectorised oracle build: same source, same bits (checked in test_oracle_kat.py).  Compiled for THIS host when a compiler
    is here (-march=native: on an AVX-512 host the strict 1/sqrt — binary64 sqrt and divide per pair — runs several times faster than
    in the shipped x86-64-v3 build, which is what keeps the headline-size CPU pass inside the GPU suite's budget); the shipped build
    otherwise.  Every operation is IEEE-exact, so the vector width changes no bit."""
    import subprocess
    import oracle as O
    path = None
    try:
        path = str(tmp_path_factory.mktemp("oracle_native") / "libnbody_ref_native.so")
        subprocess.run(["gcc", "-std=c11", "-fPIC", "-shared", "-O3", "-march=native", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
                        "-fno-math-errno", "-fno-trapping-math", "-o", path, os.path.join(ROOT, "oracle", "nbody_ref.c"), "-lm"],
                       check=True, capture_output=True, timeout=180)
    except Exception:
        path = None
    return O.Oracle(fast=True, path=path)


@pytest.fixture(scope="session")
def nb():
    """The product package."""
    import mini_nbody_amd
    return mini_nbody_amd


def has_gpu():
    try:
        return os.path.exists("/dev/kfd") and len(os.listdir("/sys/class/kfd/kfd/topology/nodes")) > 1
    except OSError:
        return False
