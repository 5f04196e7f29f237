"""The C host program (mini_nbody_amd/host/nbody.c: plain C over the C-ABI) end to end on the GPU: its checksum of
positions after 10 strict-mode iterations must equal the oracle's, for the host-pointer bodyForce()/integrate() loop
and for the device-resident loop."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "nbody")


@pytest.mark.parametrize("extra", [["--host-loop"], []])
def test_c_host_program_matches_oracle(nb, oracle_fast, extra):
    if not os.path.exists(EXE):
        pytest.fail("build/nbody is missing: run `make host`")
    import oracle as O
    n, iters = 4096, 10
    out = subprocess.run([EXE, str(n), str(iters), "--strict"] + extra, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"checksum \(sum of positions\): (\S+) (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    got = np.array([float(m.group(k)) for k in (1, 2, 3)])
    cfg = re.search(r"(\d+) segments x (\d+) pieces, sum block (\d+), (\d+) launch", out.stdout)
    assert cfg, out.stdout
    segments, pieces, block, launches = (int(cfg.group(k)) for k in (1, 2, 3, 4))
    assert block == 1024 and launches in (1, 2) and segments > 1 and pieces in (4, 16)      # the engine's own configuration
    pos, vel = nb.make_bodies(n)
    oracle_fast.step_order(pos, vel, 0.01, iters, summ=O.SUM_BLOCKED, block=block, sub=segments, wsplit=pieces)
    want = pos[:, :3].astype(np.float64).sum(0)
    assert np.allclose(got, want, rtol=0, atol=1e-6 * np.abs(pos[:, :3]).sum()), (got, want)   # printed with %.9g
    assert re.search(r"%d Bodies .* Billion Interactions / second" % n, out.stdout)


def test_gpu_host_program_and_cpu_program_print_the_same_checksum():
    """BASELINE config 1 (CPU program, oracle/nbody_cpu) next to the GPU host program in --strict mode: identical
    initial conditions, identical arithmetic, identical checksum line — with one sequential sum per body (what a plain
    CPU nbody.c does) and in the engine's orders (blocked sums; 16 segments walked whole, or the default 8 segments of four
    pieces each)."""
    cpu = os.path.join(ROOT, "oracle", "nbody_cpu")
    assert os.path.exists(cpu) and os.path.exists(EXE)
    line = lambda out: [l for l in out.splitlines() if l.startswith("checksum")][0]
    for cpu_args, gpu_args in (([], ["--sum", "seq", "--jsub", "1", "--wsplit", "1"]),
                               (["--sum", "blocked", "--segments", "16"], ["--jsub", "16", "--one-launch", "--wsplit", "1"]),
                               (["--sum", "blocked", "--segments", "8", "--wsplit", "4"], ["--jsub", "8", "--one-launch", "--wsplit", "4"]),
                               (["--sum", "blocked", "--segments", "4", "--wsplit", "16"], []),      # the engine's own choice at N = 4096
                               (["--rtl"], ["--rtl"]),                                             # the RTL-faithful result
                               (["--rtl"], ["--rtl", "--host-loop"]),
                               (["--sum", "blocked", "--block", "256", "--segments", "3", "--wsplit", "4"], ["--jsub", "3", "--block", "256", "--two-launch", "--wsplit", "4"])):
        a = subprocess.run([cpu, "4096", "10"] + cpu_args, capture_output=True, text=True, timeout=300)
        b = subprocess.run([EXE, "4096", "10"] + ([] if "--rtl" in gpu_args else ["--strict"]) + gpu_args, capture_output=True, text=True, timeout=300)
        assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
        assert line(a.stdout) == line(b.stdout), (cpu_args, gpu_args)


def test_fp64_host_program_and_cpu_program_print_the_same_checksum():
    """Round 4: `build/nbody --fp64 --strict` (IEEE sqrt and divide on the GPU) prints the 17-digit checksum of `oracle/nbody_cpu --fp64`
    run in the order the engine reports — device loop and host-pointer loop"""
    cpu = os.path.join(ROOT, "oracle", "nbody_cpu")
    line = lambda out: [l for l in out.splitlines() if l.startswith("checksum")][0]
    for extra in ([], ["--host-loop"], ["--jsub", "3", "--wsplit", "1"]):
        b = subprocess.run([EXE, "4096", "6", "--fp64", "--strict"] + extra, capture_output=True, text=True, timeout=300)
        assert b.returncode == 0, b.stderr
        cfg = re.search(r"(\d+) segments x (\d+) pieces", b.stdout)
        assert cfg, b.stdout
        a = subprocess.run([cpu, "4096", "6", "--fp64", "--segments", cfg.group(1), "--wsplit", cfg.group(2)], capture_output=True, text=True, timeout=300)
        assert a.returncode == 0, a.stderr
        assert line(a.stdout) == line(b.stdout), (extra, cfg.groups())


_RTL_SUMS = {}


@pytest.mark.parametrize("mode", [[], ["--served"], ["--timed"], ["--timed", "--served"]])
def test_ps_side_mailbox_driver_in_c(nb, oracle_fast, mode):
    """mini_nbody_amd/host/mailbox_driver.c — the PS-side driver INTEGRATION.md §1 shows, as a compiled C program over the C-ABI: ONE
    power-up, then requests of several sizes incl. the RTL's maximum and an empty one, called (nbody_mailbox_run) or served (the driver
    only writes RAM A and polls word 0).  Its force checksums must be the oracle's in the RTL-faithful mode (bit-exact forces, so the
    double-precision sums agree to the last printed digit), within 1e-5 in the timed arithmetic; BEGIN cleared, ticks >= 1, the force of body k at
    word k of RAM B, word 0 and the words beyond N untouched (S/compute_store.vhd:221-242)."""
    import oracle as O
    exe = os.path.join(ROOT, "build", "mailbox_driver")
    if not os.path.exists(exe):
        pytest.fail("build/mailbox_driver is missing: run `make host`")
    sizes = [9, 1000, 32767, 0, 40]
    out = subprocess.run([exe] + mode + [str(n) for n in sizes], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("NUM_PTS")]
    assert len(lines) == len(sizes), out.stdout
    for n, line in zip(sizes, lines):
        m = re.match(r"NUM_PTS (\d+)  BEGIN (\d)  ticks (\d+)  checksum \(sum of forces\): (\S+) (\S+) (\S+)  RAM B word 0 and beyond word N untouched: (\w+)", line)
        assert m, line
        assert int(m.group(1)) == n and m.group(2) == "0" and int(m.group(3)) >= 1 and m.group(7) == "yes", line
        got = np.array([float(m.group(k)) for k in (4, 5, 6)])
        if n == 0:
            assert np.all(got == 0)
            continue
        if n not in _RTL_SUMS:                    # one oracle pass per size for the four modes
            pos, _ = nb.make_bodies(n)
            f = oracle_fast.forces_f32(pos, pos, d2=O.D2_REFERENCE, rsqrt=O.RSQRT_F64, summ=O.SUM_FPGA16)
            _RTL_SUMS[n] = (f[:, :3].astype(np.float64).sum(0), np.abs(f[:, :3].astype(np.float64)).sum())
        want, scale = _RTL_SUMS[n]
        if "--timed" in mode:
            assert np.abs(got - want).max() <= 1e-5 * scale, (n, got, want)
        else:
            assert np.allclose(got, want, rtol=0, atol=2e-9 * scale), (n, got, want)          # printed with %.9g
