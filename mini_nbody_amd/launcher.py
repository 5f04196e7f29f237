"""The supervision of bench.py's worker processes (moved out of bench.py in round 6; bench.py re-exports every name).

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts and SUPERVISES its own N workers (one per GPU; the parent never
touches a GPU); under torch.distributed.run every rank process supervises its own worker and the N supervisors agree through a directory
under /tmp; one GPU runs in ONE supervised worker the same way.  A worker that fails, a transport that is not up in time, or an attempt that
passes its share of the command's one time budget is killed — exactly the process group this file started, never a pattern — and the job is
started again with the next transport (rccl -> peer -> host).  Nothing here imports torch or touches a GPU.  Covered on the CPU with stub
workers: tests/test_bench_launcher.py."""
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

WORKER_ENV = "NBODY_BENCH_WORKER"
READY_ENV = "NBODY_BENCH_READY_FILE"          # the worker touches it once its engine and transport are up (and self-tested)
IMPORTED_ENV = "NBODY_BENCH_IMPORTED_FILE"    # ... and this one as soon as `import torch` has returned
WARM_IMPORT_S = 15.0      # what a LATER attempt's `import torch` costs: the first one paged the image in (1-2 min), the rest come from the page cache
RUN_RESERVE_S = 90.0      # what a fallback attempt needs after its import: transport + self-test, warm-up + timed steps, the CPU leg
EXIT_NO_FALLBACK = 3      # --no-fallback and the requested transport failed


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def kill_group(proc, grace=5.0):
    """end exactly the process group this supervisor started (start_new_session), never a pattern"""
    if proc.poll() is not None:
        return
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            return
        t0 = time.time()
        while time.time() - t0 < grace:
            if proc.poll() is not None:
                return
            time.sleep(0.05)


def tail(path, n=320):
    try:
        return open(path, errors="replace").read()[-n:].strip().replace("\n", " | ")
    except OSError:
        return ""


def _worker_preexec():
    """in the child, before exec: its own session (so that the supervisor can end the whole group, and only that group) and
    a parent-death signal (a supervisor that is killed outright must not leave workers holding GPUs)"""
    os.setsid()
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)      # PR_SET_PDEATHSIG
    except Exception:
        pass


class Terminated(BaseException):
    """SIGTERM / SIGINT reached the supervisor: unwind through the clean-up that ends the workers"""


def _raise_terminated(signum, frame):
    raise Terminated("signal %d" % signum)


def start_worker(cmd, rank, local, world, port, transport, logdir, attempt, extra_env=None):
    env = dict(os.environ)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(local), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), WORKER_ENV: "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    # the workers hold their own rendezvous (rank 0 hosts the store on `port`), whatever launched the supervisors
    for k in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)
    env.pop("OMP_NUM_THREADS", None)      # torch.distributed.run pins it to 1; the CPU-baseline leg wants the host's cores
    if env.get("NBODY_VIRTUAL_HOSTS"):
        # rehearsal on a box with fewer GPUs than ranks: every rank poses as a host of its own (RCCL compares host id and PCI bus id before
        # it refuses a second rank on one device) and RCCL connects the ranks through its socket transport over loopback — the real multi-rank
        # RCCL path (communicator of N ranks, ncclAllGather, grouped ncclSend/ncclRecv, streams, events, the hand-shake) without a second GPU
        env.update({"NCCL_HOSTID": "nbody-virtual-host-%d" % rank, "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NBODY_OVERSUBSCRIBE": "1"})
    if extra_env:
        env.update(extra_env)
    env[READY_ENV] = os.path.join(logdir, "a%d_rank%d.ready" % (attempt, rank))
    env[IMPORTED_ENV] = os.path.join(logdir, "a%d_rank%d.imported" % (attempt, rank))
    full = list(cmd) + (["--transport", transport] if transport else [])
    with open(os.path.join(logdir, "a%d_rank%d.out" % (attempt, rank)), "w") as out, \
         open(os.path.join(logdir, "a%d_rank%d.err" % (attempt, rank)), "w") as err:
        return subprocess.Popen(full, env=env, stdout=out, stderr=err, preexec_fn=_worker_preexec), out.name, err.name


def json_lines(path):
    """every line of `path` that parses as a JSON object, in order (a line still being written is skipped)"""
    out = []
    try:
        for l in open(path).read().splitlines():
            if l.startswith("{"):
                try:
                    out.append(json.loads(l))
                except ValueError:
                    pass
    except OSError:
        pass
    return out


def json_line(path):
    lines = json_lines(path)
    return lines[-1] if lines else None


def wait_workers(procs, deadline_s, peers_failed=lambda: None, poll=0.1, startup_s=None, ready_files=(), imported_files=(),
                 ready_after_import_s=None, line_seen=lambda: False, extras_s=None, info=None):
    """-> None when the attempt counts as a success, else a reason string.
    deadline_s: seconds, or a function of the import time measured so far (None until every worker has imported torch).
    startup_s: every worker must have touched its ready file (engine created, transport self-test passed) within that many
      seconds; ready_after_import_s: ... and within that many after the last worker's `import torch` returned — where a
      transport hangs it hangs in its first collective, and that is noticed a minute after the import, not at the deadline.
    line_seen(): rank 0 has printed its headline line.  From then on nothing can fail the attempt: the workers get extras_s
      more seconds for the extras pass, and a time-out, a crash or a peer's complaint after that moment only sets
      info["extras"] (the caller then reports the FIRST line).
    info (dict, filled in): import_s, ready_s, first_line_s, extras."""
    info = {} if info is None else info
    info.update({"import_s": None, "ready_s": None, "first_line_s": None, "extras": None})
    t0 = time.time()
    started = not ready_files or (startup_s is None and ready_after_import_s is None)
    while True:
        now = time.time() - t0
        if info["import_s"] is None and imported_files and all(os.path.exists(f) for f in imported_files):
            info["import_s"] = now
        if info["first_line_s"] is None and line_seen():
            info["first_line_s"] = now
        headline = info["first_line_s"] is not None
        if not started and not headline:
            if all(os.path.exists(f) for f in ready_files):
                started = True
                info["ready_s"] = now
            elif ready_after_import_s is not None and info["import_s"] is not None and now - info["import_s"] > ready_after_import_s:
                return "transport not up %.0f s after `import torch` returned (%.0f s)" % (ready_after_import_s, info["import_s"])
            elif startup_s is not None and now > startup_s:
                return "timed out after %.0f s before the transport was up" % startup_s
        codes = [p.poll() for p, _, _ in procs]
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            i, c = bad[0]
            why = "worker exited with code %d: %s" % (c, tail(procs[i][2]) or tail(procs[i][1]))
            if headline or line_seen():
                info["extras"] = "failed: " + why
                return None
            return why
        if all(c == 0 for c in codes):
            return None
        why = peers_failed()
        if why:
            if headline:
                info["extras"] = "failed: " + why
                return None
            return why
        limit = deadline_s(info["import_s"]) if callable(deadline_s) else deadline_s
        if headline and extras_s is not None and now - info["first_line_s"] > extras_s:
            info["extras"] = "timed out %.0f s after the headline line" % extras_s
            return None
        if now > limit:
            if headline:
                info["extras"] = "timed out (attempt deadline %.0f s)" % limit
                return None
            return "timed out after %.0f s" % limit
        time.sleep(poll)


def supervise(worker_cmd, world, my_ranks, transport, deadline_s, rdzv_dir=None, log=sys.stderr, extra_env=None, startup_s=None,
              budget_s=None, ready_after_import_s=None, extras_s=None, no_fallback=False, t_start=None,
              warm_import_s=WARM_IMPORT_S, reserve_s=RUN_RESERVE_S, min_attempt_s=10.0, min_start_s=20.0):
    """Run the job as `world` worker processes, of which this supervisor owns `my_ranks` (all of them when it was started
    bare; one when torch.distributed.run started one supervisor per rank — then rdzv_dir, shared by the supervisors, carries
    the worker port, every supervisor's verdict on an attempt and the "headline printed" marker).  Attempt 0 uses
    `transport`; if a worker fails or a deadline passes, everything is killed and the next attempt runs with --transport
    peer, then host — unless no_fallback.  budget_s bounds the whole call: attempt k may take
    min(deadline_s, budget left - (attempts still to come) x (import + reserve_s)), where import = the time this attempt's
    `import torch` took, capped at warm_import_s: the first import of a fresh box pages the image in (1-2 min), a later
    attempt's comes from the page cache.
    Returns (exit code, JSON object of rank 0's line or None)."""
    logdir = tempfile.mkdtemp(prefix="nbody_bench_logs_")
    lead = 0 in my_ranks
    t_start = time.time() if t_start is None else t_start
    # what is tried, in order: the requested transport (RCCL: one process per GPU, the intended path); then ONE process driving all
    # the GPUs with peer copies over xGMI (needs nothing from RCCL, costs the host ~0.4 ms of launches per step at 8 GPUs); then
    # positions staged through host memory (needs nothing from the GPU fabric at all)
    # ("auto" would fall back to the host transport inside the worker when RCCL cannot be set up; under supervision the better
    #  fallback is the next attempt's, so the first attempt insists on RCCL)
    first = "rccl" if transport == "auto" else transport
    attempts = [first] + [t for t in ("peer", "host") if t != first and not (first == "host" and t == "peer")]
    if no_fallback:
        attempts = attempts[:1]
    records = []
    first_reason = None
    try:
        for attempt, tr in enumerate(attempts):
            t_a = time.time()
            to_come = len(attempts) - attempt - 1
            if budget_s is not None and attempt > 0 and budget_s - (t_a - t_start) < min_start_s:   # no fallback is started on a spent budget
                records.append({"transport": tr, "seconds": 0.0, "result": "not started: %.0f s of the budget left" % (budget_s - (t_a - t_start))})
                print("[bench supervisor] attempt %d (--transport %s) not started: budget spent" % (attempt, tr), file=log, flush=True)
                break

            def attempt_deadline(import_s, t_a=t_a, to_come=to_come):
                if budget_s is None:
                    return deadline_s
                left = budget_s - (t_a - t_start)
                warm = warm_import_s if import_s is None else min(import_s, warm_import_s)
                return max(min_attempt_s, min(deadline_s, left - to_come * (warm + reserve_s)))

            # ---- the workers' rendezvous port: chosen by the supervisor of rank 0, published through rdzv_dir
            if rdzv_dir is None:
                port = free_port()
            else:
                pfile = os.path.join(rdzv_dir, "port.%d" % attempt)
                if lead:
                    port = free_port()
                    with open(pfile + ".tmp", "w") as f:
                        f.write(str(port))
                    os.replace(pfile + ".tmp", pfile)
                else:
                    t0 = time.time()
                    while not os.path.exists(pfile):
                        if time.time() - t0 > attempt_deadline(None):
                            return 1, None
                        time.sleep(0.05)
                    port = int(open(pfile).read())
            procs = []
            try:
                for r in my_ranks:
                    procs.append(start_worker(worker_cmd, r, r, world, port, tr, logdir, attempt, extra_env))
            except BaseException:       # a worker could not be started: do not leave the ones that were behind
                for p, _, _ in procs:
                    kill_group(p)
                raise

            def peers_failed():
                if rdzv_dir is None:
                    return None
                for f in glob.glob(os.path.join(rdzv_dir, "verdict.%d.*" % attempt)):
                    if f.endswith(".tmp"):
                        continue
                    try:
                        txt = open(f).read()
                    except OSError:
                        continue
                    if txt and not txt.startswith("ok") and not txt.startswith("a peer supervisor reported"):
                        return "a peer supervisor reported: " + txt
                return None

            marker = os.path.join(rdzv_dir, "line.%d" % attempt) if rdzv_dir is not None else None

            def line_seen():
                """rank 0's headline line is out (the lead sees the worker's stdout; the others a marker the lead leaves)"""
                if lead:
                    if json_line(procs[0][1]) is None:
                        return False
                    if marker and not os.path.exists(marker):
                        open(marker, "w").close()
                    return True
                return bool(marker) and os.path.exists(marker)

            info = {}
            try:
                ready = [os.path.join(logdir, "a%d_rank%d.ready" % (attempt, r)) for r in my_ranks]
                imported = [os.path.join(logdir, "a%d_rank%d.imported" % (attempt, r)) for r in my_ranks]
                reason = wait_workers(procs, attempt_deadline, peers_failed, startup_s=startup_s, ready_files=ready, imported_files=imported,
                                      ready_after_import_s=ready_after_import_s, line_seen=line_seen, extras_s=extras_s, info=info)
            except BaseException:       # interrupted (Ctrl-C, the launcher's SIGTERM): the workers go with the supervisor
                for p, _, _ in procs:
                    kill_group(p)
                raise
            if rdzv_dir is not None:
                for r in my_ranks:      # this supervisor's verdict, then everybody's
                    vf = os.path.join(rdzv_dir, "verdict.%d.%d" % (attempt, r))
                    with open(vf + ".tmp", "w") as f:
                        f.write("ok" if reason is None else reason)
                    os.replace(vf + ".tmp", vf)
                t0 = time.time()
                while reason is None and info.get("extras") is None:
                    files = glob.glob(os.path.join(rdzv_dir, "verdict.%d.*" % attempt))
                    reason = peers_failed()
                    if reason and line_seen():          # the headline is out: a peer's late complaint concerns the extras only
                        info["extras"], reason = "failed: " + reason, None
                        break
                    if len([f for f in files if not f.endswith(".tmp")]) >= world or reason:
                        break
                    if time.time() - t0 > attempt_deadline(info.get("import_s")):
                        reason = "peer supervisors did not report within %.0f s" % attempt_deadline(info.get("import_s"))
                    time.sleep(0.05)
            for p, _, _ in procs:
                kill_group(p)
            rec = {"transport": tr, "seconds": round(time.time() - t_a, 1), "result": "ok" if reason is None else reason[:400],
                   "import_s": None if info.get("import_s") is None else round(info["import_s"], 1)}
            if info.get("extras"):
                rec["extras"] = info["extras"][:400]
            records.append(rec)
            if reason is None:
                lines = json_lines(procs[0][1]) if lead else []
                # every line rank 0 prints is the headline plus the extras finished so far: the LAST complete one is the fullest account,
                # also when the extras pass ended badly (then "extras" says how)
                obj = lines[-1] if lines else None
                if lead and obj is None:
                    reason = "rank 0 printed no JSON line: " + tail(procs[0][2])
                    records[-1]["result"] = reason[:400]
                else:
                    if obj is not None:
                        if info.get("extras"):
                            obj["extras"] = info["extras"]
                        obj["transport_used"] = tr
                        obj["fallback_from"] = attempts[0] if attempt > 0 else None
                        obj["fallback_reason"] = first_reason if attempt > 0 else None
                        obj["attempts"] = records
                        if first_reason is not None:
                            obj.setdefault("config", {})["comm"] = "%s (%s attempt: %s)" % (obj.get("config", {}).get("comm"), attempts[0], first_reason)
                    return 0, obj
            print("[bench supervisor] attempt %d (--transport %s) failed after %.0f s: %s" % (attempt, tr, time.time() - t_a, reason), file=log, flush=True)
            if first_reason is None:
                first_reason = reason if len(reason) <= 420 else reason[:60] + " ... " + reason[-340:]
        return (EXIT_NO_FALLBACK if no_fallback else 1), None
    finally:
        shutil.rmtree(logdir, ignore_errors=True)


def supervise_single(cmd, budget_s, extras_s, log=sys.stderr, poll=0.05):
    """One GPU: the benchmark runs in ONE worker process and this parent, which never touches a GPU, returns the LAST complete JSON
    line the worker printed.  The worker prints the headline — complete, cpu_baseline included — before it attempts any study pass
    (strict arithmetic, the other BASELINE configurations) and one more line after each; so a study kernel that faults, aborts or hangs
    costs neither the headline nor the passes finished before it: the line then says so in "extras".  Returns (exit code, object)."""
    logdir = tempfile.mkdtemp(prefix="nbody_bench_logs_")
    t0 = time.time()
    proc = None
    try:
        proc, out_path, err_path = start_worker(cmd, 0, int(os.environ.get("LOCAL_RANK", "0")), 1, free_port(), None, logdir, 0)
        first_line_t, extras = None, None
        while True:
            code = proc.poll()
            if first_line_t is None and json_line(out_path) is not None:
                first_line_t = time.time()
            if code is not None:
                if code != 0:
                    extras = "failed: worker exited with code %d: %s" % (code, tail(err_path) or tail(out_path))
                break
            if first_line_t is not None and extras_s is not None and time.time() - first_line_t > extras_s:
                extras = "timed out %.0f s after the headline line" % extras_s
                break
            if budget_s is not None and time.time() - t0 > budget_s:
                extras = "timed out (budget %.0f s)" % budget_s
                break
            time.sleep(poll)
        kill_group(proc)
        lines = json_lines(out_path)
        if not lines:
            print("[bench supervisor] no JSON line from the worker (%s): %s" % (extras or "exit code %s" % proc.returncode, tail(err_path, 2000)), file=log, flush=True)
            return (proc.returncode if proc.returncode not in (None, 0) else 1), None
        obj = lines[-1]
        if extras:
            obj["extras"] = extras[:600]
            print("[bench supervisor] headline kept; after it: %s" % extras, file=log, flush=True)
        return 0, obj
    except BaseException:
        if proc is not None:
            kill_group(proc)
        raise
    finally:
        shutil.rmtree(logdir, ignore_errors=True)
