"""The library's HOST code — context.cpp, comm.cpp, mailbox.cpp: everything in libnbody_hip.so that is not device code — under
AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU (sanitizers run on the CPU build only; the GPU pool has none).
tests/host_stub/hip_stub.cpp stands in for the HIP runtime ("device" memory is malloc'd, streams run synchronously, a capture
records closures) and for kernels.hip (a simple stand-in force pushed through the REAL data flow of a launch: ForceArgs, segment
bounds, partial sums, arrival counters, ascending combine, apply, the mailbox's ingest and device-written completion);
tests/host_stub/host_sanity.cpp drives the C-ABI: contexts of several sizes, 70-step loops (graph capture and replay), every
segmentation / combine form / wave split, row windows, fp64, one process over three stub devices with ragged slices, the mailbox in
every form with the RTL's address map checked against a sentinel, and the SERVED mailbox with a second thread hammering
nbody_get_info and the refused entry points ("a TSan build of a host-only stub of the serve loop").
What this checks is the host's logic — buffer sizes and offsets (ASan sees every index the host computed), the per-request switch of
N, the service thread's hand-over, the guard, lifetimes at shutdown; the kernels' arithmetic is the GPU tests' business."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = "/opt/rocm/lib/llvm/bin/clang++"
SRC = [os.path.join(ROOT, "mini_nbody_amd", "csrc", f) for f in ("context.cpp", "comm.cpp", "mailbox.cpp")] + \
      [os.path.join(ROOT, "tests", "host_stub", f) for f in ("hip_stub.cpp", "host_sanity.cpp")]


def build(tmp_path, name, san):
    if not os.path.exists(CXX):
        pytest.skip("no clang++ under /opt/rocm")
    exe = str(tmp_path / name)
    r = subprocess.run([CXX, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wall",
                        "-Wno-unused-function"] + san + SRC + ["-o", exe, "-ldl", "-lpthread"], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "sanitize" in r.stderr and "unsupported" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def run(exe, args, extra_env):
    env = dict(os.environ, STUB_DEVICES="3", NBODY_OVERSUBSCRIBE="1", **extra_env)
    out = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "host_sanity ok" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]


@pytest.mark.skipif(shutil.which("make") is None, reason="no toolchain")
def test_host_code_is_address_and_ub_clean(tmp_path):
    exe = build(tmp_path, "host_sanity_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    run(exe, ["2000"], {"ASAN_OPTIONS": "detect_leaks=1"})


def test_served_mailbox_is_race_free_under_tsan(tmp_path):
    """the service thread against a second thread that calls nbody_get_info and every refused entry point while 4000 served requests of
    alternating sizes are posted and polled from memory"""
    exe = build(tmp_path, "host_sanity_tsan", ["-fsanitize=thread"])
    run(exe, ["4000", "mailbox-only"], {"TSAN_OPTIONS": "halt_on_error=1"})
