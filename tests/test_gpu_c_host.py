"""The C host program (mini-nbody_amd/host/nbody.c: plain C over the C-ABI) end to end on the GPU: its checksum of
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
    n, iters = 4096, 10
    env = dict(os.environ, NBODY_JSUB="1")
    out = subprocess.run([EXE, str(n), str(iters), "--strict", "--jsub", "1"] + extra, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"checksum \(sum of positions\): (\S+) (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    got = np.array([float(m.group(k)) for k in (1, 2, 3)])
    pos, vel = nb.make_bodies(n)
    oracle_fast.step(pos, vel, 0.01, iters)
    want = pos[:, :3].astype(np.float64).sum(0)
    assert np.allclose(got, want, rtol=0, atol=1e-6 * np.abs(pos[:, :3]).sum()), (got, want)   # printed with %.9g
    assert re.search(r"%d Bodies .* Billion Interactions / second" % n, out.stdout)


def test_gpu_host_program_and_cpu_program_print_the_same_checksum():
    """BASELINE config 1 (CPU program, oracle/nbody_cpu) next to the GPU host program in --strict mode, one source
    segment: identical initial conditions, identical arithmetic, identical checksum line."""
    cpu = os.path.join(ROOT, "oracle", "nbody_cpu")
    assert os.path.exists(cpu) and os.path.exists(EXE)
    a = subprocess.run([cpu, "4096", "10"], capture_output=True, text=True, timeout=300)
    b = subprocess.run([EXE, "4096", "10", "--strict", "--jsub", "1"], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    line = lambda out: [l for l in out.splitlines() if l.startswith("checksum")][0]
    assert line(a.stdout) == line(b.stdout)
