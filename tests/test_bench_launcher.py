"""bench.py's supervision of its worker processes (`python bench.py --gpus N` without torch.distributed.run, and one
supervisor per rank under it), with a stub in place of the GPU worker: environment handed to the workers, a failing
worker -> one retry on the host transport with the reason recorded, a hang -> killed at the deadline and retried, and
the supervisors of a torch.distributed.run job agreeing through their shared directory.  No GPU, no torch."""
import json
import os
import subprocess
import sys
import textwrap
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = textwrap.dedent('''
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    transport = sys.argv[sys.argv.index("--transport") + 1]
    mode = os.environ.get("STUB_MODE", "ok")
    assert os.environ["NBODY_BENCH_WORKER"] == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert "TORCHELASTIC_USE_AGENT_STORE" not in os.environ and "OMP_NUM_THREADS" not in os.environ
    open(os.environ["NBODY_BENCH_IMPORTED_FILE"], "w").close()      # "import torch has returned"
    if transport not in ("host", "peer") or (transport == "peer" and mode.endswith("twice")):
        mode = mode.replace("twice", "")
        if mode == "fail" and rank == world - 1:
            sys.stderr.write("RCCL error 5 near comm.cpp:123\\n")
            sys.exit(3)
        if mode in ("hang", "fail"):
            time.sleep(3600)      # a hang; or the others wait in a collective for the rank that died
    if os.environ.get("NBODY_BENCH_READY_FILE") and mode != "noready":
        open(os.environ["NBODY_BENCH_READY_FILE"], "w").close()
    if mode == "noready" and transport not in ("host", "peer"):
        time.sleep(3600)
    line = {"metric": "stub", "n_gpus": world, "config": {"comm": "allgather / %s / overlap 1" % {"host": "host-staged", "peer": "peer copies"}.get(transport, "rccl")},
            "env": {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "NCCL_HOSTID", "NCCL_SOCKET_IFNAME", "NBODY_OVERSUBSCRIBE") if k in os.environ},
            "argv": sys.argv[1:]}
    if os.environ.get("STUB_HOSTID_DIR"):
        open(os.path.join(os.environ["STUB_HOSTID_DIR"], "rank%d" % rank), "w").write(os.environ.get("NCCL_HOSTID", ""))
    if rank == 0:
        print("noise before the line")
        print(json.dumps(line), flush=True)
    # the extras pass: after the headline line, in the same processes
    if mode == "extrashang":
        time.sleep(3600)
    if mode == "extrascrash" and rank == world - 1:
        sys.stderr.write("HIP error 719 in the extras pass\\n")
        sys.exit(5)
    if mode == "extrascrash":
        time.sleep(3600)          # the others sit in the collective the crashed rank never joins
    if mode == "extraspartial":
        if rank == 0:
            line["comm_forms"] = {"allgather": {"ms_per_step": 30.5}}
            print(json.dumps(line), flush=True)
            sys.stdout.write('{"metric": "stub", "n_gpus": 3, "comm_forms": {"allgather": {"ms_pe')      # a line cut off mid-write
            sys.stdout.flush()
        time.sleep(3600)          # the second form hangs
    if mode == "extras" and rank == 0:
        time.sleep(0.3)
        line["comm_forms"] = {"ring": {"ms_per_step": 31.0}, "direct": {"ms_per_step": 30.0}, "allgather": {"ms_per_step": 30.5}}
        print(json.dumps(line), flush=True)
''')


@pytest.fixture()
def stub(tmp_path):
    p = tmp_path / "stub_worker.py"
    p.write_text(STUB)
    return [sys.executable, str(p), "--gpus", "3", "--steps", "2"]


def test_bare_launch_starts_one_worker_per_gpu(stub):
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extra_env={"STUB_MODE": "ok"})
    assert code == 0 and obj["n_gpus"] == 3
    assert obj["env"]["RANK"] == "0" and obj["env"]["WORLD_SIZE"] == "3" and int(obj["env"]["MASTER_PORT"]) > 0
    assert obj["argv"][-2:] == ["--transport", "rccl"] and obj["config"]["comm"].startswith("allgather / rccl")


def test_failing_worker_is_retried_without_rccl(stub):
    """first fallback: one process driving all GPUs with peer copies; second: the host-staged transport"""
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extra_env={"STUB_MODE": "fail"})
    assert code == 0
    assert "peer copies" in obj["config"]["comm"] and "rccl attempt: worker exited with code 3" in obj["config"]["comm"]
    assert "RCCL error 5" in obj["config"]["comm"]
    assert obj["argv"][-2:] == ["--transport", "peer"]
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extra_env={"STUB_MODE": "failtwice"})
    assert code == 0 and "host-staged" in obj["config"]["comm"] and obj["argv"][-2:] == ["--transport", "host"]
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "host", deadline_s=60, extra_env={"STUB_MODE": "ok"})     # asked for host: nothing else is tried
    assert code == 0 and obj["argv"][-2:] == ["--transport", "host"]


def test_hang_is_killed_at_the_deadline_and_retried(stub):
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=1.5, extra_env={"STUB_MODE": "hang"})
    assert code == 0 and "rccl attempt: timed out after" in obj["config"]["comm"]


def test_transport_that_never_comes_up_is_noticed_at_the_startup_deadline(stub):
    """a hang in the first collective: no ready file within --startup-deadline -> killed long before the overall deadline"""
    import time
    t0 = time.time()
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=600, startup_s=1.5, extra_env={"STUB_MODE": "noready"})
    assert code == 0 and time.time() - t0 < 60
    assert "peer copies" in obj["config"]["comm"] and "before the transport was up" in obj["config"]["comm"]


def test_failure_on_the_host_transport_too_is_an_error(stub, tmp_path):
    bad = tmp_path / "bad.py"
    bad.write_text("import sys; sys.exit(7)")
    code, obj = bench.supervise([sys.executable, str(bad)], 2, [0, 1], "auto", deadline_s=30)
    assert code == 1 and obj is None


@pytest.mark.parametrize("mode", ["ok", "fail", "hang"])
def test_one_supervisor_per_rank_agree_through_the_shared_directory(stub, tmp_path, mode):
    """the torch.distributed.run shape: three supervisors, each owning one rank, sharing rdzv_dir"""
    rdzv = tmp_path / "rdzv"
    rdzv.mkdir()
    res = {}

    def run(rank):
        res[rank] = bench.supervise(stub, 3, [rank], "auto", deadline_s=60 if mode != "hang" else 2.0, rdzv_dir=str(rdzv),
                                    extra_env={"STUB_MODE": mode})

    threads = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert all(res[r][0] == 0 for r in range(3)), res
    assert res[1][1] is None and res[2][1] is None and res[0][1]["n_gpus"] == 3
    if mode == "ok":
        assert "rccl attempt" not in res[0][1]["config"]["comm"]
    else:
        assert "peer copies" in res[0][1]["config"]["comm"] and "rccl attempt:" in res[0][1]["config"]["comm"]


def test_command_line_entry_without_world_size(tmp_path):
    """python bench.py --gpus 2 (no WORLD_SIZE): the parent only supervises — it must not import torch or need a GPU; with
    the real worker on a box without GPUs both attempts fail and the exit code says so"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # (deadlines sized for a fresh container, where the first `import torch` alone can take a minute or two)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--bodies", "4096", "--deadline", "300",
                        "--startup-deadline", "300", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present: covered by the -m gpu test")
    assert r.returncode == 1
    assert "attempt 0 (--transport rccl) failed" in r.stderr and "attempt 1 (--transport peer) failed" in r.stderr
    assert "attempt 2 (--transport host) failed" in r.stderr
    assert "needs a GPU" in r.stderr


def test_sigterm_to_the_supervisor_ends_the_workers(tmp_path):
    """a launcher's time limit (SIGTERM to `python bench.py --gpus N`) must not leave workers behind holding GPUs: the
    supervisor unwinds through its clean-up, and a supervisor killed outright takes its workers with it (parent-death signal)"""
    import signal
    import time
    pidfile = tmp_path / "pids"
    stub = tmp_path / "sleeper.py"
    stub.write_text("import os, time\nopen(%r, 'a').write(str(os.getpid()) + '\\n')\ntime.sleep(600)\n" % str(pidfile))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent('''
        import signal, sys
        sys.path.insert(0, %r)
        import bench
        for s in (signal.SIGTERM, signal.SIGINT):
            signal.signal(s, bench._raise_terminated)
        bench.supervise([sys.executable, %r], 2, [0, 1], "auto", deadline_s=300)
    ''') % (ROOT, str(stub)))

    def alive(pid):
        try:
            os.kill(pid, 0)
            return open("/proc/%d/stat" % pid).read().split()[2] != "Z"
        except (OSError, IOError):
            return False

    for sig in (signal.SIGTERM, signal.SIGKILL):
        if pidfile.exists():
            pidfile.unlink()
        p = subprocess.Popen([sys.executable, str(driver)])
        t0 = time.time()
        while time.time() - t0 < 30 and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
            time.sleep(0.1)
        pids = [int(x) for x in pidfile.read_text().split()]
        assert len(pids) == 2 and all(alive(q) for q in pids)
        p.send_signal(sig)
        p.wait(30)
        t0 = time.time()
        while time.time() - t0 < 15 and any(alive(q) for q in pids):
            time.sleep(0.1)
        assert not any(alive(q) for q in pids), "workers survived %s of their supervisor" % sig


# ---- round 4: the extras pass, the time budget, machine-readable fallback fields -------------------------------------------

def test_extras_line_is_preferred_when_the_extras_finish(stub):
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extras_s=30, extra_env={"STUB_MODE": "extras"})
    assert code == 0 and set(obj["comm_forms"]) == {"ring", "direct", "allgather"} and "extras" not in obj
    assert obj["transport_used"] == "rccl" and obj["fallback_from"] is None and obj["fallback_reason"] is None
    assert [a["transport"] for a in obj["attempts"]] == ["rccl"] and obj["attempts"][0]["result"] == "ok"


@pytest.mark.parametrize("mode,what", [("extrashang", "timed out"), ("extrascrash", "failed: worker exited with code 5")])
def test_extras_that_hang_or_crash_keep_the_headline(stub, mode, what):
    """the headline line is out before any extra is attempted: a hang in the extras is killed at ITS deadline (not the
    attempt's), a crash ends it at once, and either way the first line is returned with exit code 0 and no retry"""
    import time
    t0 = time.time()
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=600, extras_s=1.5, extra_env={"STUB_MODE": mode})
    assert code == 0 and time.time() - t0 < 60
    assert obj["metric"] == "stub" and obj["extras"].startswith(what) and "comm_forms" not in obj
    assert obj["transport_used"] == "rccl" and len(obj["attempts"]) == 1 and obj["attempts"][0]["extras"].startswith(what)
    if mode == "extrascrash":
        assert "HIP error 719" in obj["extras"]


@pytest.mark.parametrize("shape", ["bare", "per-rank"])
def test_extras_hang_under_one_supervisor_per_rank(stub, tmp_path, shape):
    """the torch.distributed.run shape: only the lead sees rank 0's stdout; the others learn of the headline through the marker"""
    if shape == "bare":
        pytest.skip("covered above")
    rdzv = tmp_path / "rdzv"
    rdzv.mkdir()
    res = {}

    def run(rank):
        res[rank] = bench.supervise(stub, 3, [rank], "auto", deadline_s=600, extras_s=1.5, rdzv_dir=str(rdzv), extra_env={"STUB_MODE": "extrashang"})

    threads = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert all(res[r][0] == 0 for r in range(3)), res
    assert res[0][1]["metric"] == "stub" and res[0][1]["extras"].startswith("timed out")


def test_hang_in_a_timed_step_leaves_room_for_the_fallback(stub):
    """one budget for the whole command: attempt 0's deadline = budget - reserve for the attempts still to come"""
    import time
    t0 = time.time()
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=600, budget_s=8.0, warm_import_s=0.5, reserve_s=1.5, min_attempt_s=0.5,
                                min_start_s=1.0, extra_env={"STUB_MODE": "hang"})
    took = time.time() - t0
    assert code == 0 and took < 8.0 + 6.0, took                      # (+ the kill grace of the hung group)
    assert obj["transport_used"] == "peer" and obj["fallback_from"] == "rccl" and obj["fallback_reason"].startswith("timed out after 5 s")
    assert [a["transport"] for a in obj["attempts"]] == ["rccl", "peer"]
    assert 3.5 <= obj["attempts"][0]["seconds"] <= 7.0 and obj["attempts"][0]["result"].startswith("timed out")
    assert obj["attempts"][1]["result"] == "ok" and "rccl attempt: timed out" in obj["config"]["comm"]


def test_hang_at_start_up_is_noticed_soon_after_the_import(stub):
    """two-stage readiness: `import torch` returned, but the transport is not up ready_after_import_s later"""
    import time
    t0 = time.time()
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=600, startup_s=300, ready_after_import_s=1.0, budget_s=540,
                                extra_env={"STUB_MODE": "noready"})
    assert code == 0 and time.time() - t0 < 30
    assert obj["transport_used"] == "peer" and "transport not up 1 s after `import torch` returned" in obj["fallback_reason"]
    assert obj["attempts"][0]["import_s"] is not None and obj["attempts"][0]["seconds"] < 15


def test_no_fallback_turns_a_failed_transport_into_exit_code_3(stub):
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, no_fallback=True, extra_env={"STUB_MODE": "fail"})
    assert code == bench.EXIT_NO_FALLBACK == 3 and obj is None
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, no_fallback=True, extra_env={"STUB_MODE": "ok"})
    assert code == 0 and obj["transport_used"] == "rccl"


def test_the_fallback_label_names_the_transport_that_was_tried_first(stub):
    """--transport peer that fails falls back to host, and the text says "peer attempt", not "rccl attempt" """
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "peer", deadline_s=60, extra_env={"STUB_MODE": "failtwice"})
    assert code == 0 and obj["transport_used"] == "host" and obj["fallback_from"] == "peer"
    assert "(peer attempt: worker exited with code 3" in obj["config"]["comm"]


def test_budget_spent_before_a_fallback_starts():
    """nothing is started with less than 20 s of the budget left"""
    import time
    code, obj = bench.supervise([sys.executable, "-c", "import time; time.sleep(3600)"], 2, [0, 1], "auto", deadline_s=600, budget_s=21.5,
                                warm_import_s=0.0, reserve_s=9.5, min_attempt_s=0.5, t_start=time.time())
    assert code == 1 and obj is None      # attempt 0: 21.5 - 2 x 9.5 = 2.5 s; then < 20 s are left: peer and host are not started


def test_an_extra_that_hangs_keeps_the_extras_finished_before_it(stub):
    """rank 0 prints one more line after every finished extra; the supervisor takes the last COMPLETE line"""
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=600, extras_s=1.5, extra_env={"STUB_MODE": "extraspartial"})
    assert code == 0 and obj["metric"] == "stub" and "argv" in obj
    assert obj["comm_forms"] == {"allgather": {"ms_per_step": 30.5}} and obj["extras"].startswith("timed out")


def test_virtual_hosts_give_every_rank_its_own_host_identity(stub, tmp_path, monkeypatch):
    """NBODY_VIRTUAL_HOSTS=1 (rehearsal of an N-rank RCCL job on fewer GPUs): every worker gets a different NCCL_HOSTID, loopback as
    RCCL's socket interface and leave to share a device; without it none of these is set"""
    monkeypatch.delenv("NCCL_HOSTID", raising=False)
    monkeypatch.delenv("NBODY_VIRTUAL_HOSTS", raising=False)
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extra_env={"STUB_MODE": "ok"})
    assert code == 0 and "NCCL_HOSTID" not in obj["env"]
    monkeypatch.setenv("NBODY_VIRTUAL_HOSTS", "1")
    code, obj = bench.supervise(stub, 3, [0, 1, 2], "auto", deadline_s=60, extra_env={"STUB_MODE": "ok", "STUB_HOSTID_DIR": str(tmp_path)})
    assert code == 0 and obj["env"]["NCCL_HOSTID"] == "nbody-virtual-host-0" and obj["env"]["NCCL_SOCKET_IFNAME"] == "lo" and obj["env"]["NBODY_OVERSUBSCRIBE"] == "1"
    ids = {open(tmp_path / ("rank%d" % r)).read() for r in range(3)}
    assert ids == {"nbody-virtual-host-0", "nbody-virtual-host-1", "nbody-virtual-host-2"}


# ---- round 5: one GPU runs in a supervised worker too; every line carries a measured cpu_baseline ---------------------------------

STUB1 = textwrap.dedent('''
    import json, os, signal, sys, time
    assert os.environ["NBODY_BENCH_WORKER"] == "1" and os.environ["WORLD_SIZE"] == "1" and os.environ["RANK"] == "0"
    assert "--transport" not in sys.argv
    mode = os.environ.get("STUB_MODE", "ok")
    if mode == "early":
        sys.stderr.write("no usable HIP device\\n")
        sys.exit(7)
    line = {"metric": "stub", "n_gpus": 1, "value": 4700.0, "cpu_baseline": {"value": 59.7, "cores": 256}}
    print("noise before the line")
    print(json.dumps(line), flush=True)                  # the headline: complete before any study pass
    line["strict_mode"] = {"value": 2840.0}
    print(json.dumps(line), flush=True)
    if mode == "fault":
        os.kill(os.getpid(), signal.SIGSEGV)             # a study kernel takes the process down
    if mode == "hang":
        time.sleep(3600)
    line["configs"] = {"config2": {"value": 4500.0}}
    print(json.dumps(line), flush=True)
''')


@pytest.fixture()
def stub1(tmp_path):
    p = tmp_path / "stub_worker1.py"
    p.write_text(STUB1)
    return [sys.executable, str(p), "--gpus", "1"]


def test_one_gpu_worker_is_supervised_and_the_last_line_wins(stub1, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "ok")
    code, obj = bench.supervise_single(stub1, 60, 30)
    assert code == 0 and obj["value"] == 4700.0 and obj["strict_mode"]["value"] == 2840.0 and obj["configs"]["config2"]["value"] == 4500.0
    assert "extras" not in obj


@pytest.mark.parametrize("mode,what", [("fault", "failed: worker exited with code -11"), ("hang", "timed out")])
def test_a_study_pass_that_faults_or_hangs_keeps_the_headline_and_the_passes_before_it(stub1, monkeypatch, mode, what):
    import time
    monkeypatch.setenv("STUB_MODE", mode)
    t0 = time.time()
    code, obj = bench.supervise_single(stub1, 600, 1.5)
    assert code == 0 and time.time() - t0 < 60
    assert obj["value"] == 4700.0 and obj["cpu_baseline"]["value"] == 59.7 and obj["strict_mode"]["value"] == 2840.0
    assert "configs" not in obj and obj["extras"].startswith(what)


def test_a_worker_that_dies_before_its_headline_is_an_error(stub1, monkeypatch):
    import io
    monkeypatch.setenv("STUB_MODE", "early")
    log = io.StringIO()
    code, obj = bench.supervise_single(stub1, 60, 30, log=log)
    assert code == 7 and obj is None and "no usable HIP device" in log.getvalue()


def test_one_gpu_command_line_goes_through_the_supervisor(tmp_path):
    """`python bench.py` (the driver's N = 1 invocation) starts a worker and prints exactly ONE line: the worker's last"""
    stub = tmp_path / "bench.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the real supervisor code with a stub where the GPU benchmark would run
    stub.write_text(src.replace('    if not os.path.exists(os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip.so")):',
                                '    print(json.dumps({"metric": "stub", "n_gpus": 1, "worker": os.environ.get(WORKER_ENV), "argv": argv})); '
                                'print(json.dumps({"metric": "stub", "n_gpus": 1, "second": True, "argv": argv})); return\n'
                                '    if not os.path.exists(os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip.so")):', 1))
    env = dict(os.environ, PYTHONPATH=ROOT)     # (the copy lives outside the repository: mini_nbody_amd/launcher.py is found through the path)
    r = subprocess.run([sys.executable, str(stub), "--steps", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["second"] is True and lines[0]["argv"] == ["--steps", "3"] and "supervisor_seconds" in lines[0]
    # --in-process: no worker, both lines come from this very process
    r = subprocess.run([sys.executable, str(stub), "--in-process"], capture_output=True, text=True, timeout=120, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 2 and lines[0]["worker"] is None


def test_every_line_gets_a_measured_cpu_baseline():
    """an N > 1 line used to carry cpu_baseline.value null.  Now: a 3-second row sample on rank 0 (15 s at N = 1 or on request),
    and the leg itself returns a number and the cores it used (here: a fraction of a second on this container's cores)."""
    assert bench.cpu_leg_seconds(1, "auto") == 15.0 and bench.cpu_leg_seconds(2, "auto") == 3.0 and bench.cpu_leg_seconds(8, "auto") == 3.0
    assert bench.cpu_leg_seconds(8, "always") == 15.0 and bench.cpu_leg_seconds(8, "never") is None and bench.cpu_leg_seconds(1, "auto", True) is None
    cb = bench.cpu_baseline(8192, 42, False, target_s=0.3)
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "8192 sources" in cb["sample"]
