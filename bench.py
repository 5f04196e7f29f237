#!/usr/bin/env python3
"""bench.py — billion pair-interactions/s of the all-pairs force path (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over all N bodies: force accumulation over all N^2 ordered pairs
(self included, S/top_level.vhd:237-243), kick, drift — nbody_step() of include/nbody.h, state resident
in HBM before the timed region starts.  Workload: N = 1,048,576 fp32 (BASELINE configs[2]/[3], the
configuration the metric is quoted on); with N GPUs the same N is sharded by body (strong scaling) and
the positions travel by RCCL inside libnbody_hip.so.  torch is used for the rendezvous, the barrier,
the max-over-ranks and torch.cuda.synchronize() only.  `--fp64 --bodies 4194304` is BASELINE configs[4].

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts and SUPERVISES its own N
worker processes (one per GPU; the parent never touches a GPU); under torch.distributed.run every rank process supervises
its own worker and the N supervisors agree through a directory under /tmp.  A worker that fails, a transport that is not up
--ready-deadline seconds after `import torch` returned, or an attempt that passes its deadline (an RCCL hang has no other
symptom) is killed and the job is started again — with --transport peer (rank 0 drives all the GPUs from one process with
peer copies over xGMI: nothing of RCCL is needed), and if that fails too with --transport host (positions staged through host
memory and torch.distributed).  The whole command has ONE time budget (--budget, 540 s): an attempt's deadline is what is
left of it minus a reserve for the attempts still to come, so that every fallback can finish inside the launcher's own limit.
The line says what ran in machine-readable fields — transport_used, fallback_from, fallback_reason, attempts[{transport,
seconds, result}] — and --no-fallback turns a failure of the requested transport into exit code 3 instead of a retry (for
scaling runs where a peer-copy number must not pass as the RCCL point).  Workers run the library's transport self-test
(nbody_comm_selftest: every received word checked) before the warm-up.  NBODY_VIRTUAL_HOSTS=1 in the environment rehearses an N-rank RCCL
job on a box with fewer GPUs: every rank poses as its own host (NCCL_HOSTID) and RCCL connects them over loopback sockets.

Extras (N > 1 only; the N = 1 path is untouched).  Rank 0 prints the headline line FIRST; then, in the same worker
processes, (a) three steps each of NBODY_COMM_RING with one launch per arriving slice, NBODY_COMM_DIRECT and
NBODY_COMM_ALLGATHER -> "comm_forms": {form: {ms_per_step, comm_exposed_ms_per_step, value}} (ring
against direct over xGMI) plus the all-gather with no overlap, whose exposed time is the transfer's own duration, and (b) at N = 8 BASELINE configs[4]: an fp64 N = 4,194,304 engine, 1 warm-up + 2 timed steps ->
"config5": {value, ms_per_step, roofline, hbm_gb_per_s, comm_exposed_ms_per_step}; then a second line = headline + extras,
after every finished extra rank 0 prints one more line = headline + extras so far, and the supervisor takes the LAST complete line.
The extras have their own deadline (--extras-deadline, 90 s after the first line appeared): past it the workers are killed and that
line is returned with "extras": "timed out ..." and exit code 0 — an extra that hangs costs neither the headline nor the extras
finished before it.

Rank 0 prints ONE JSON line, always with:
  roofline      the force kernel priced at 20 flop per pair against the fp32 (157.3 TFLOP/s) or
                fp64 (78.6: measured v_fma_f64 issue rate, profiles/r01_microbench_valu_issue.txt) VECTOR peak — no contraction for the matrix cores — with the kernel's
                duration measured live by HIP events on the library's compute stream; beside it the instruction-issue
                bound (30 cycles per wave-pair in fp32, 76 in fp64) and cycles per wave-pair.  `traffic` and the
                other *_pmc fields come from the committed rocprofv3 passes of THIS configuration
                (profiles/pmc_*.json, tools/profile.sh) and are attached only when that profile's recorded
                configuration equals the run's; otherwise `traffic` is null.
  cpu_baseline  the oracle (oracle/nbody_ref.c, kind "port": the reference is VHDL and has no CPU path) timed on
                this box's host cores, rank 0, after the timed region, on a bounded row sample: about 15 s of host work at N = 1,
                3 s on an N > 1 line (the other ranks wait at a barrier) — every line carries a measured value and its core count.
One GPU: the benchmark runs in a supervised worker process (the parent never touches a GPU; --in-process for profilers).  The worker
prints the headline line — complete, cpu_baseline included — BEFORE any study pass, then one more line after each of
  strict_mode   (fp32) the rate of NBODY_ARITH_STRICT — the arithmetic in which the GPU equals the CPU oracle bit for bit —
                from up to 3 steps after the timed region: what the parity claim costs on this box; never `value`.
  configs       (the default line only) the other BASELINE configurations one GPU reaches, each {value, ms_per_step, frac}: config2 (N = 65536,
                100 steps: the engine's default and --variant lds --tile 256), fp64 (N = 262144, 3 steps, frac against 78.6 TFLOP/s and
                of the issue bound), config1 (N = 4096, 10 iterations: oracle/nbody_cpu beside build/nbody --strict, same checksum line);
                at most --configs-budget (40) seconds; every entry carries its own cpu_baseline (a <= 3-s row sample at its N and precision).
and the parent prints the LAST complete line: a study kernel that faults or hangs costs neither the headline nor the passes before it
("extras" then says what happened).
and for N > 1: comm_exposed_ms_per_step — how long per step the compute stream sat waiting for arriving position slices
(HIP events around every such wait; 0 = the transfers hid behind the own-slice kernel).
"""
import argparse
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process device memory on this pool needs dmabuf IPC (the launcher normally exports it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FLOP_PER_PAIR = 20                # SURVEY.md §8(d) convention (literal count: 18)
# fp32: MI355X_MICROARCH.md "Peak FP32 (vector) 157.3 TFLOPS" = 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz.
# fp64: the guide states NO fp64 vector figure; 78.6 is this repository's own measurement — v_fma_f64 issues at 4.0 cycles per wave64
# on a SIMD (profiles/r01_microbench_valu_issue.txt: v_fma_f64 / v_mul_f64 W=4, W=8), i.e. 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_VECTOR_TFLOPS = {"f32": 157.3, "f64": 78.6}
# cycles of VALU issue per wave64 pair: fp32 11 x 2 + 8 (v_rsq_f32); fp64 15 x 4 + 16 (v_rsq_f64; round 4: the inverse cube in six operations, was 16 x 4 + 16)
# measured: profiles/r01_microbench_valu_issue.txt, DESIGN.md §3
ISSUE_CYCLES_PER_WAVE_PAIR = {"f32": 30, "f64": 76}
METRIC = "billion pair-interactions/s at N=1M fp32; 1/2/4/8 GPUs + % FP32 roofline"
COMM_NAMES = {0: "ring", 1: "allgather", 2: "auto", 3: "direct"}


_CPU_ORACLE = {}      # the host build of the oracle, made once per process (the configs entries time their own samples with it)


def cpu_oracle():
    """(Oracle, its library path or None, its thread count): oracle/nbody_ref.c built for THIS host — the cpu_baseline legs' checker-turned-baseline"""
    if _CPU_ORACLE:
        return _CPU_ORACLE["ora"], _CPU_ORACLE["path"], _CPU_ORACLE["cores"]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    path = None
    # same source, tuned for this host if the compiler is here (falls back to the prebuilt x86-64-v3 build)
    try:
        tmp = tempfile.mkdtemp(prefix="nbody_ref_native_")
        path = os.path.join(tmp, "libnbody_ref_native.so")
        subprocess.run(["gcc", "-std=c11", "-fPIC", "-shared", "-O3", "-march=native", "-fopenmp", "-ffp-contract=off",
                        "-fno-math-errno", "-fno-trapping-math", "-o", path, os.path.join(ROOT, "oracle", "nbody_ref.c"),
                        "-lm"], check=True, capture_output=True, timeout=120)
    except Exception:
        path = None
    ora = O.Oracle(fast=True, path=path)
    # all the cores this process may run on (torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the other
    # ranks are idle at a barrier while this runs)
    try:
        ora.set_num_threads(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    _CPU_ORACLE.update(ora=ora, path=path, cores=ora.num_threads())
    return ora, path, _CPU_ORACLE["cores"]


def cpu_baseline(n, seed, fp64, target_s=15.0):
    """The oracle timed on the host cores: a row sample (first rows x all N sources), sized to take
    roughly target_s seconds (15 at N = 1; 3 on an N > 1 line, where the other ranks wait at a barrier, and per `configs` entry).
    Returns the JSON object."""
    import numpy as np
    ora, path, cores = cpu_oracle()
    import oracle as O
    import mini_nbody_amd as nb
    pos, _ = nb.make_bodies(n, seed=seed, dtype=np.float64 if fp64 else np.float32)

    def run(rows):
        if fp64:
            ora.forces_f64(pos[:rows], pos)
        else:
            ora.forces_f32(pos[:rows], pos, rsqrt=O.RSQRT_DIVSQRT)

    rows = min(n, 4096)
    run(256)      # warm the thread pool
    t0 = time.perf_counter()
    run(rows)
    t = time.perf_counter() - t0
    rate = rows * n / t
    rows2 = int(min(n, max(rows, (rate * target_s / n) // 16 * 16)))
    if rows2 > rows:
        t0 = time.perf_counter()
        run(rows2)
        t = time.perf_counter() - t0
        rows = rows2
    # a small N fits whole in a fraction of the time asked for (N = 65536: 4.3e9 pairs, ~0.1 s — as long as waking 256 threads): the
    # whole pass is then repeated until about a third of target_s has gone by, and the passes are timed together
    passes = 1
    if rows == n and t < target_s / 6.0:
        passes = int(max(2, min(64, (target_s / 3.0) / max(t, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(passes):
            run(rows)
        t = time.perf_counter() - t0
    what = "fp64, sequential-j, 1.0/sqrt" if fp64 else "fp32, sequential-j, 1.0f/sqrtf"
    return {"value": round(passes * rows * n / t / 1e9, 3), "unit": "billion pair-interactions/s", "cores": cores, "kind": "port",
            "sample": "oracle/nbody_ref.c (%s), first %d of %d rows x all %d sources%s, %.1f s, gcc -O3 %s -fopenmp"
                      % (what, rows, n, n, " x %d passes" % passes if passes > 1 else "", t, "-march=native" if path else "-march=x86-64-v3")}


def cpu_baseline_program(argv, timeout_s):
    """The cpu_baseline leg of BASELINE configs[0]: the oracle run as a program (`oracle/nbody_cpu N iters ...`: test infrastructure, timed
    on this box's host cores as a REPORTED baseline, never a product path) -> (G pairs/s, ms per step, its checksum line, threads)."""
    import re
    o = subprocess.run([os.path.join(ROOT, "oracle", "nbody_cpu")] + list(argv), capture_output=True, text=True, timeout=timeout_s)
    if o.returncode:
        raise RuntimeError("oracle/nbody_cpu exited with %d: %s" % (o.returncode, o.stderr[-300:]))
    rate = re.search(r"(\d+) threads\): average ([0-9.]+) Billion Interactions / second \(([0-9.]+) ms / step\)", o.stdout)
    chk = [l for l in o.stdout.splitlines() if l.startswith("checksum")]
    return float(rate.group(2)), float(rate.group(3)), (chk[0] if chk else None), int(rate.group(1))


def cpu_leg_seconds(world, mode, disabled=False):
    """How long the host-CPU leg may take on a line of `world` GPUs (None: no leg, no `cpu_baseline` key): about 15 s of host work on a
    one-GPU line; on an N > 1 line a 3-second row sample — every line carries a MEASURED baseline, and the other ranks sit at a barrier
    for 3 s, not 15 — unless --cpu-baseline always asks for the long sample there too."""
    if disabled or mode == "never":
        return None
    return 15.0 if (world == 1 or mode == "always") else 3.0


def kernel_source_sha():
    """identifies the PRODUCT kernel code a profile was taken on (a profile of an older loop must not be attached to a new one): the
    four kernel files (the device translation unit kernels.hip, the kernels, the launch argument block and the hand-scheduled loop; the host-side context.cpp / comm.cpp / mailbox.cpp are not kernel code) with everything under `#ifdef NBODY_DIAG_LOOPS` left out — the experiment encodings of `make diag`
    (libnbody_hip_diag.so) are not in the product library, and adding one must not orphan the product's profiles"""
    import hashlib
    h = hashlib.sha1()
    for f in ("nbody_kernels.hpp", "nbody_args.hpp", "force_loop_gfx950.inc", "kernels.hip"):
        skip = 0
        for line in open(os.path.join(ROOT, "mini_nbody_amd", "csrc", f), "rb").read().splitlines(True):
            t = line.strip()
            if skip:
                if t.startswith(b"#if"):
                    skip += 1
                elif t.startswith(b"#endif"):
                    skip -= 1
                elif skip == 1 and (t.startswith(b"#else") or t.startswith(b"#elif")):
                    skip = 0                     # (the product's side of an #ifdef ... #else: hashed)
                continue
            if t.startswith(b"#ifdef NBODY_DIAG_LOOPS"):
                skip = 1
                continue
            h.update(line)
    return h.hexdigest()[:12]


def matching_pmc(run_cfg):
    """profiles/pmc_*.json written by tools/parse_prof.py for the SAME configuration (bodies, precision, ranks, variant,
    segments, block length, launches per step), or None: counters cannot be read from inside the timed process."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_*.json"))):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("config") == run_cfg:
            pj["_file"] = os.path.relpath(f, ROOT)
            return pj
    return None



# ---------------------------------------------------------------------------------------------------------------------
# Supervision of the worker processes.  Nothing in this section imports torch or touches a GPU.
# the supervision of the worker processes lives in mini_nbody_amd/launcher.py (no torch, no GPU); every name stays reachable as bench.<name>
from mini_nbody_amd.launcher import (  # noqa: E402,F401
    WORKER_ENV, READY_ENV, IMPORTED_ENV, WARM_IMPORT_S, RUN_RESERVE_S, EXIT_NO_FALLBACK, free_port, kill_group, tail,
    _worker_preexec, Terminated, _raise_terminated, start_worker, json_lines, json_line, wait_workers, supervise, supervise_single)


def supervisor_main(args, argv):
    """-> exit code, or None when this process should run the benchmark itself (it IS a worker, or --in-process)."""
    if os.environ.get(WORKER_ENV) or (args.gpus <= 1 and args.in_process):
        return None
    t_start = time.time()
    world_env = os.environ.get("WORLD_SIZE")
    for sig in (signal.SIGTERM, signal.SIGINT):      # a launcher's time limit must end the workers too, not orphan them
        try:
            signal.signal(sig, _raise_terminated)
        except ValueError:                            # not the main thread (tests call supervise() directly)
            pass
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    if args.gpus <= 1:
        if world_env is not None and int(world_env) != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, world_env))
        code, obj = supervise_single(cmd, args.budget if args.budget > 0 else None, args.extras_deadline)
        if obj is not None:
            obj["supervisor_seconds"] = round(time.time() - t_start, 1)
            print(json.dumps(obj), flush=True)
        return code
    # drop a --transport given on the command line: the supervisor passes the one of the attempt
    clean = []
    skip = False
    for a in cmd:
        if skip:
            skip = False
            continue
        if a == "--transport":
            skip = True
            continue
        if a.startswith("--transport="):
            continue
        clean.append(a)
    kw = dict(startup_s=args.startup_deadline, budget_s=args.budget if args.budget > 0 else None, ready_after_import_s=args.ready_deadline,
              extras_s=args.extras_deadline, no_fallback=args.no_fallback, t_start=t_start)
    if world_env is None:
        code, obj = supervise(clean, args.gpus, list(range(args.gpus)), args.transport, args.deadline, **kw)
    else:
        world = int(world_env)
        if world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        rank = int(os.environ.get("RANK", "0"))
        # every supervisor of this job has the same parent (the torch.distributed.run agent) and the same MASTER_PORT
        rdzv = os.path.join(tempfile.gettempdir(), "nbody_bench_rdzv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
        os.makedirs(rdzv, exist_ok=True)
        code, obj = supervise(clean, world, [rank], args.transport, args.deadline, rdzv_dir=rdzv, **kw)
        if rank == 0:
            time.sleep(0.5)
            shutil.rmtree(rdzv, ignore_errors=True)
    if obj is not None:
        obj["supervisor_seconds"] = round(time.time() - t_start, 1)
        print(json.dumps(obj), flush=True)
    return code


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=1 << 20, help="bodies (default: the metric N = 1,048,576)")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--variant", choices=["auto", "smem", "lds", "readlane", "isa"], default="auto")
    ap.add_argument("--isa-phase", type=int, default=-1)
    ap.add_argument("--iblock", type=int, default=0)
    ap.add_argument("--jsub", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--sum", choices=["blocked", "seq"], default="blocked")
    ap.add_argument("--sum-block", type=int, default=0)
    ap.add_argument("--fuse", type=int, default=-1, help="1: one launch per step (in-launch combine), 0: two, -1: auto")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline never")
    ap.add_argument("--arith", choices=["fma3", "reference", "strict", "refstrict", "rtl"], default="fma3",
                    help="study runs only (the headline is fma3, the timed arithmetic): reference = the RTL's five roundings for d2; strict / "
                         "refstrict = IEEE-exact, bit-identical to the CPU oracle; rtl = refstrict + the reference's 16 partial sums and adder "
                         "tree over ONE stream of all N (the mailbox drop-in's arithmetic).  The line's metric and config say which")
    ap.add_argument("--strict-pass", choices=["auto", "never"], default="auto",
                    help="auto: a one-GPU fp32 line also carries `strict_mode` — the rate of NBODY_ARITH_STRICT, the arithmetic that is "
                         "bit-identical to the CPU oracle, from up to 3 steps AFTER the timed region (not the timed mode, not `value`)")
    ap.add_argument("--cpu-baseline", choices=["auto", "always", "never"], default="auto",
                    help="the oracle timed on this box's host cores after the timed region, rank 0: auto = a ~15-s row sample at N = 1, a 3-s one "
                         "at N > 1 (the other ranks wait at a barrier meanwhile); always: the 15-s sample at N > 1 too; never: no cpu_baseline key")
    ap.add_argument("--comm", choices=["auto", "ring", "allgather", "direct"], default="auto")
    ap.add_argument("--transport", choices=["auto", "rccl", "peer", "host"], default="auto",
                    help="how positions travel between GPUs: rccl (one process per GPU, RCCL over xGMI; auto = rccl, falling back to host when "
                         "RCCL cannot be set up), peer (rank 0 drives all the GPUs from one process with peer copies, the other ranks idle), "
                         "host (staged through host memory and torch.distributed)")
    ap.add_argument("--xcd-map", type=int, default=-1, help="XCD-aware placement of source segments: 1 on, 0 off, -1 the engine's default")
    ap.add_argument("--overlap", type=int, default=1, help="0 gather first, 1 own slice then the rest, 2 one launch per arriving slice")
    ap.add_argument("--wsplit", type=int, default=-1, help="4: 64-row workgroups whose waves split the segment, 1: round 2's layout, -1: the engine's choice")
    ap.add_argument("--events", choices=["auto", "inline", "separate"], default="auto",
                    help="HIP events around every force kernel inside the timed region (inline), or the timed region on the path "
                         "nbody_step() users get (HIP-graph replay, no events) and the kernel duration from a second pass (separate); "
                         "auto: separate when a step is short (one rank's share < 1e10 pairs), where events would change the path")
    ap.add_argument("--deadline", type=float, default=600.0, help="N > 1: upper bound in seconds on one supervised attempt (the budget usually binds first)")
    ap.add_argument("--budget", type=float, default=540.0,
                    help="N > 1: seconds the WHOLE command may take, fallbacks included: an attempt's deadline is what is left minus a reserve "
                         "for the attempts still to come (0: no budget, --deadline per attempt)")
    ap.add_argument("--startup-deadline", type=float, default=300.0,
                    help="N > 1: seconds within which every worker must have its engine and transport up (first import of torch on a fresh box: 1-2 min)")
    ap.add_argument("--ready-deadline", type=float, default=90.0,
                    help="N > 1: seconds after `import torch` returned within which the transport must be up and self-tested (a hang in the first collective)")
    ap.add_argument("--extras-deadline", type=float, default=90.0,
                    help="N > 1: seconds the extras pass may take after the headline line appeared; past it the headline is returned alone")
    ap.add_argument("--autotune", action="store_true",
                    help="N > 1, RCCL, --comm auto: measure the three transfer forms in warm-up and time the fastest (off by default: the headline "
                         "runs the library's default form — one ncclAllGather — and the other forms are measured by the extras pass, AFTER the "
                         "headline line is out, so that a form that has never run on real links cannot cost the headline)")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N > 1: if the requested transport fails, exit with code 3 instead of retrying on peer copies / the host")
    ap.add_argument("--extras", choices=["auto", "off", "forms", "all"], default="auto",
                    help="N > 1: after the headline line, time the three transfer forms (forms) and BASELINE configs[4] in fp64 (all); "
                         "auto = forms, plus config 5 when N = 8; never at N = 1")
    ap.add_argument("--in-process", action="store_true",
                    help="one GPU: run the benchmark in this very process instead of in a supervised worker (profilers: rocprofv3 then sees "
                         "the kernels in the process it started)")
    ap.add_argument("--configs-pass", choices=["auto", "never"], default="auto",
                    help="auto: a default one-GPU fp32 line (N = 1,048,576, timed arithmetic) also carries `configs` — BASELINE configs 1 and 2 and "
                         "the fp64 arithmetic, measured after the headline line is out, at most --configs-budget seconds")
    ap.add_argument("--configs-budget", type=float, default=40.0)
    ap.add_argument("--config5-bodies", type=int, default=4194304, help="bodies of the extras pass's fp64 run (BASELINE configs[4]: 4,194,304)")
    args = ap.parse_args(argv)
    code = supervisor_main(args, argv)
    if code is not None:
        raise SystemExit(code)

    if not os.path.exists(os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip.so")):
        # fresh checkout (built files are git-ignored): build the HIP library; a failure is fatal, there is no fallback
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            subprocess.run(["make", "lib"], cwd=ROOT, check=True, capture_output=True)
        else:
            for _ in range(600):
                if os.path.exists(os.path.join(ROOT, "mini_nbody_amd", "libnbody_hip.so")):
                    break
                time.sleep(0.5)
            time.sleep(2.0)
    import torch
    if os.environ.get(IMPORTED_ENV):
        open(os.environ[IMPORTED_ENV], "w").close()   # tells the supervisor: the image is paged in, what follows is the transport
    import mini_nbody_amd as nb
    import mini_nbody_amd.distributed as D

    rank, world, local = D.env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    peer = world > 1 and args.transport == "peer"    # one process (rank 0) drives all the GPUs; the other ranks keep the barriers only
    if not (peer and rank != 0):                     # (those never touch a GPU, not even to ask whether one is there)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not peer:
        os.environ.setdefault("NBODY_DEVICE", str(local % torch.cuda.device_count()))
    import torch.distributed as dist
    if world > 1:
        D.init_process_group(backend="gloo")   # control plane only; the data path is RCCL inside the library

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.n
    dt = 0.01
    import numpy as np
    MAXOP = dist.ReduceOp.MAX if world > 1 else None
    # extras (N > 1 only): the three transfer forms, and BASELINE configs[4] in fp64 at N = 8 (or when asked for)
    want_forms = world > 1 and args.extras != "off"
    want_c5 = world > 1 and (args.extras == "all" or (args.extras == "auto" and world == 8))

    def allmax(vals):
        """MAX over the ranks of a few numbers (every rank calls it, also a peer-transport rank that holds no engine)"""
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64)
        dist.all_reduce(t, op=MAXOP)
        return [float(x) for x in t]

    def run_timed(e, steps, warmup, inline):
        """W untimed steps, then EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() on both sides; MAX over ranks.
        e is None on the ranks of a peer-transport job that hold no GPU: they keep the collectives only."""
        if e is not None:
            e.step(dt, warmup)
            e.sync()
            e.set_option(nb.OPT_TIMING, 1 if inline else 0)
            e.kernel_time(reset=True)
            e.comm_time(reset=True)
        barrier()
        if e is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if e is not None:
            e.step(dt, steps)                     # EXACTLY K steps
            e.sync()
            torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_steps, kernel_ms, launches, wait_ms, waits = steps, 0.0, 0, 0.0, 0
        if e is not None:
            if not inline:
                # the timed region ran the path nbody_step() users get (one GPU: HIP-graph replay); the kernel's duration comes
                # from a second pass with events around every launch
                kernel_steps = min(steps, 50)
                e.set_option(nb.OPT_TIMING, 1)
                e.kernel_time(reset=True)
                e.comm_time(reset=True)
                e.step(dt, kernel_steps)
                e.sync()
            kernel_ms, launches = e.kernel_time(reset=True)
            wait_ms, waits = e.comm_time(reset=True)
        elapsed, kernel_ms, wait_ms = allmax([elapsed, kernel_ms, wait_ms])
        return {"elapsed": elapsed, "kernel_ms": kernel_ms, "launches": launches, "wait_ms": wait_ms, "waits": waits, "kernel_steps": kernel_steps}

    def open_engine(bodies, fp64):
        """this rank's engine for `bodies` bodies (None: a peer-transport rank other than 0, which never touches a GPU)"""
        if peer:
            if rank != 0:
                return None
            e = nb.NBody(bodies, fp64=fp64, tile=args.tile, ngpus=world)
            e.transport = "peer copies over xGMI, one process driving all %d GPUs" % world
            return e
        return D.make_engine(bodies, fp64=fp64, tile=args.tile, transport=args.transport)

    def roofline_of(e, cfg, r, bodies, fp64, inline):
        """the force kernel priced at 20 flop per pair against the vector peak; duration from HIP events on the compute stream"""
        dtype = "f64" if fp64 else "f32"
        n_local = cfg["n_local"]
        launches = r["launches"] // world if peer else r["launches"]   # one process timed the launches of all its devices; durations are the slowest device's
        launches_per_step = max(1, launches // max(1, r["kernel_steps"]))
        pairs_per_launch = float(n_local) * float(bodies) / launches_per_step
        avg_launch_s = r["kernel_ms"] * 1e-3 / max(1, launches)
        kernel_rate = pairs_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0   # pairs/s on one GPU
        achieved_tflops = kernel_rate * FLOP_PER_PAIR / 1e12
        peak = PEAK_VECTOR_TFLOPS[dtype]
        cu, clk = e.info(nb._lib.INFO_CU_COUNT), e.info(nb._lib.INFO_CLOCK_KHZ) * 1e3
        simds = cu * 4
        issue_bound = simds * 64.0 / ISSUE_CYCLES_PER_WAVE_PAIR[dtype] * clk   # pairs/s at the nominal clock
        wave_pairs_per_launch = pairs_per_launch / 64.0
        return {"bound": "valu", "achieved": round(achieved_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved_tflops / peak, 4), "traffic": None, "flop_per_pair": FLOP_PER_PAIR,
                "kernel_ms_avg": round(avg_launch_s * 1e3, 4), "kernel_launches": launches,
                "kernel_gpairs_per_s": round(kernel_rate / 1e9, 1),
                "algorithmic_flops_per_launch": pairs_per_launch * FLOP_PER_PAIR,
                "algorithmic_hbm_bytes_per_launch": n_local * (32 if fp64 else 16) * 4 / launches_per_step,
                "issue_cycles_per_wave_pair_model": ISSUE_CYCLES_PER_WAVE_PAIR[dtype],
                "issue_bound_gpairs_per_s": round(issue_bound / 1e9, 1),
                "frac_of_issue_bound": round(kernel_rate / issue_bound, 4) if issue_bound else None,
                "cycles_per_wave_pair_at_nominal_clock": round(avg_launch_s * clk * simds / wave_pairs_per_launch, 2) if wave_pairs_per_launch else None,
                "kernel_events": "inline (inside the timed region)" if inline else "separate pass of %d steps after the timed region (the timed region ran without events%s)" % (r["kernel_steps"], ", HIP-graph replay" if world == 1 else ""),
                "note": "VALU-issue-bound: per pair 11 full-rate + 1 quarter-rate instruction in fp32 (30 cycles per wave64), "
                        "15 + 1 in fp64 (76); neither HBM nor MFMA bounds it (no contraction; HBM traffic is 64 B per body per step)"}

    eng = open_engine(n, args.fp64)
    if eng is not None:
        eng.set_option(nb.OPT_VARIANT, {"auto": nb.VARIANT_AUTO, "smem": nb.VARIANT_SMEM, "lds": nb.VARIANT_LDS,
                                        "readlane": nb.VARIANT_READLANE, "isa": nb.VARIANT_ISA}[args.variant])
        if args.isa_phase >= 0:
            eng.set_option(nb.OPT_ISA_PHASE, args.isa_phase)
        eng.set_option(nb.OPT_IBLOCK, args.iblock)
        eng.set_option(nb.OPT_JSUB, args.jsub)
        eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED if args.sum == "blocked" else nb.SUM_SEQ)
        if args.arith != "fma3":
            eng.set_option(nb.OPT_ARITH, {"reference": nb.ARITH_REFERENCE, "strict": nb.ARITH_STRICT, "refstrict": nb.ARITH_REFERENCE_STRICT,
                                          "rtl": nb.ARITH_REFERENCE_STRICT}[args.arith])
            if args.arith == "rtl":
                eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_FPGA16)
                eng.set_option(nb.OPT_JSUB, 1)
        if args.sum_block > 0:
            eng.set_option(nb.OPT_SUM_BLOCK, args.sum_block)
        eng.set_option(nb.OPT_FUSE_COMBINE, args.fuse)
        if args.xcd_map >= 0:
            eng.set_option(nb.OPT_XCD_MAP, args.xcd_map)
        eng.set_option(nb.OPT_COMM, {"auto": nb.COMM_AUTO, "ring": nb.COMM_RING, "allgather": nb.COMM_ALLGATHER, "direct": nb.COMM_DIRECT}[args.comm])
        eng.set_option(nb.OPT_OVERLAP, args.overlap)
        eng.set_option(nb.OPT_WSPLIT, args.wsplit)
    transport = (getattr(eng, "transport", "rccl") if eng is not None else "peer") if world > 1 else None
    if world > 1 and transport == "rccl":
        # every word of a patterned all-gather checked, in the form the steps will use, before anything is timed; a failure
        # ends this worker with a non-zero code and the supervisor starts the job again on the next transport
        eng.comm_selftest()
    if os.environ.get(READY_ENV):
        open(os.environ[READY_ENV], "w").close()     # tells the supervisor: engine created, transport up (and self-tested)
    if eng is not None:
        pos, vel = nb.make_bodies(n, seed=args.seed, dtype=np.float64 if args.fp64 else np.float32)
        eng.upload(pos, vel)                      # inputs resident in HBM before the timed region
    autotuned = None
    if world > 1 and transport == "rccl" and args.comm == "auto" and args.overlap == 1 and args.autotune:
        # warm-up, untimed: the three transfer forms measured on THIS job's links, the fastest kept (ties: the library's default)
        chosen, form_ms = D.autotune_comm(eng, dt, restore=(pos, vel))
        args.overlap = {"ring": 2}.get(chosen, 1)
        autotuned = "autotuned in warm-up (ms per step: %s) -> %s" % (", ".join("%s %.3f" % kv for kv in form_ms.items()), chosen)

    share = float(n) * float(n) / world
    inline = args.events == "inline" or (args.events == "auto" and share >= 1e10)
    r = run_timed(eng, args.steps, args.warmup, inline)
    elapsed, wait_ms, waits, kernel_steps = r["elapsed"], r["wait_ms"], r["waits"], r["kernel_steps"]

    out = None
    if rank == 0:
        cfg = eng.config
        # sanity of the state the timed steps produced: this rank's own slice, without any collective (nothing after the
        # timed region may stall the report)
        p_own, _ = eng.download() if peer else eng.download_slice()
        finite = bool(np.isfinite(p_own).all())
        dtype = "f64" if args.fp64 else "f32"
        pairs_per_step = float(n) * float(n)
        value = pairs_per_step * args.steps / elapsed / 1e9
        run_cfg = {"n": n, "dtype": dtype, "n_gpus": world, "variant": cfg["variant"], "iblock": cfg["iblock"], "nseg": cfg["nseg"],
                   "sum_order": cfg["sum_order"], "sum_block": cfg["sum_block"], "launches_per_step": cfg["launches_per_step"],
                   "wsplit": cfg["wsplit"], "isa_phase": cfg["isa_phase"], "long_buffers": cfg["long_buffers"], "xcd_map": cfg["xcd_map"],
                   "kernel_source_sha": kernel_source_sha()}
        if args.arith != "fma3":
            run_cfg["arith"] = args.arith       # (absent = the timed arithmetic: the profiles of the headline keep matching)
        roof = roofline_of(eng, cfg, r, n, args.fp64, inline)
        if args.arith != "fma3":
            roof["note"] = ("study arithmetic %s, not the timed mode: the issue model beside it (30 cycles per wave-pair) is the timed arithmetic's; "
                            "strict adds 8 full-rate operations and a compare per pair (DESIGN.md 3.6)" % args.arith)
        pj = matching_pmc(run_cfg)
        if pj:
            roof["traffic"] = pj.get("hbm_bytes_per_launch")
            roof["pmc"] = {"source": "%s (rocprofv3 passes of this configuration, not this run)" % pj["_file"],
                           "clock_ghz": round(pj.get("clock_ghz", 0.0), 3),
                           "cycles_per_wave_pair": round(pj.get("cycles_per_wave_pair", 0.0), 2),
                           "hbm_gb_per_s": round(pj.get("hbm_gb_per_s", 0.0), 2),
                           "hbm_frac_of_peak": round(pj.get("hbm_frac_of_8tbs", 0.0), 5),
                           "kernel_ms_avg_trace": pj.get("force_kernel_avg_ms_trace"),
                           # "VALU-busy" as the north_star means it: (sum over the VALU instructions a wave issues for one pair of
                           # their issue cycles: 11 x 2 + 8 in fp32, 15 x 4 + 16 in fp64, profiles/r01_microbench_valu_issue.txt)
                           # / the SIMD cycles the kernel actually took per wave-pair (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / wave-pairs)
                           "valu_issue_busy": round(ISSUE_CYCLES_PER_WAVE_PAIR[dtype] / pj["cycles_per_wave_pair"], 4) if pj.get("cycles_per_wave_pair") else None,
                           "valu_issue_busy_formula": "(11 x 2 + 8 | 15 x 4 + 16 issue cycles per wave-pair) / measured SIMD cycles per wave-pair",
                           "sq_active_inst_valu_over_busy_cycles": pj.get("sq_valu_busy")}
        out = {
            "metric": METRIC if (n == (1 << 20) and not args.fp64 and args.arith == "fma3") else
                      "billion pair-interactions/s at N=%d %s%s" % (n, "fp64" if args.fp64 else "fp32", "" if args.arith == "fma3" else " (--arith %s)" % args.arith),
            "value": round(value, 2), "unit": "billion pair-interactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "N=%d %s all-pairs softened gravity, leapfrog kick-drift, dt=0.01, seed %d"
                                   % (n, "fp64" if args.fp64 else "fp32", args.seed),
                       "n_bodies": n, "pairs_per_step": pairs_per_step, "parallelism": "bodies sharded over %d GPU(s)" % world,
                       "kernel": cfg, "arith": args.arith, "kernel_source_sha": kernel_source_sha(),
                       "comm": ("%s / %s / overlap %d / stream priority %d%s" % ("hipMemcpyPeerAsync" if peer else COMM_NAMES.get(eng.info(nb._lib.INFO_COMM_FORM), "?"),
                                                                          transport, args.overlap, eng.info(nb._lib.INFO_COMM_PRIORITY),
                                                                          " / " + autotuned if autotuned else "")) if world > 1 else None,
                       "graph_replay": bool(world == 1 and not inline and args.steps >= 4),
                       "finite": finite},
            "roofline": roof,
        }
        if world > 1:
            out["comm_exposed_ms_per_step"] = round(wait_ms / max(1, kernel_steps), 4)
            out["comm_waits_per_step"] = round(waits / max(1, kernel_steps), 2)
            out["transport_used"] = "peer" if peer else ("rccl" if transport == "rccl" else "host")
    extras = want_forms or want_c5
    if rank == 0:
        target_s = cpu_leg_seconds(world, args.cpu_baseline, args.no_cpu_baseline)
        if target_s is not None:
            # after the timed region, on rank 0 (the other ranks wait at the barrier below, idle)
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.seed, args.fp64, target_s)
            except Exception as e:    # the GPU result is reported in any case
                out["cpu_baseline"] = {"value": None, "unit": "billion pair-interactions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)            # THE HEADLINE LINE — complete, and out before any study pass or extra is attempted
    barrier()

    def publish(key, value):
        """rank 0: one more line = headline + the passes finished so far (the supervisor takes the last complete line, so a later
        pass that faults or hangs cannot take the headline or an earlier pass with it)"""
        if rank == 0:
            out[key] = value
            print(json.dumps(out), flush=True)

    if world == 1:
        # one GPU: the study passes, each published as it finishes
        if not args.fp64 and args.strict_pass == "auto" and args.arith == "fma3":
            publish("strict_mode", strict_pass(eng, nb, n, dt, args.steps))
        eng.close()
        default_line = (n == (1 << 20) and not args.fp64 and args.arith == "fma3" and args.variant == "auto" and args.sum == "blocked"
                        and args.jsub == 0 and args.iblock == 0 and args.wsplit == -1)
        if args.configs_pass == "auto" and default_line:
            configs_pass(nb, args, np, dt, run_timed, roofline_of, publish, args.configs_budget)
    elif extras:
        if want_forms:
            comm_forms_pass(eng, nb, args, n, transport, run_timed, publish)
        if eng is not None:
            eng.close()
        if want_c5:
            publish("config5", config5_pass(nb, args, world, rank, open_engine, run_timed, roofline_of, np))
        barrier()
    elif eng is not None:
        eng.close()
    if world > 1:
        dist.destroy_process_group()


def configs_pass(nb, args, np, dt, run_timed, roofline_of, publish, budget_s):
    """The BASELINE configurations one GPU can reach besides the headline's, on the box and in the process that timed the headline,
    AFTER the headline line is out (never `value`):
      config2              N = 65536 fp32, 100 steps, the engine's default kernel                     BASELINE configs[1]
      config2_lds_tile256  the same through the north_star's form: sources tiled into LDS, tile 256   BASELINE configs[1] verbatim
      fp64                 N = 262144 fp64, 3 steps: the arithmetic of BASELINE configs[4] (whose N = 4,194,304 over 8 GPUs is 524,288 per GPU)
      config1              N = 4096, 10 iterations: the CPU program (oracle/nbody_cpu, through cpu_baseline_program(): this configuration's
                           cpu_baseline leg) beside the GPU host program in strict arithmetic
                           (build/nbody --strict, one sequential sum per body): same checksum line          BASELINE configs[0]
    Each entry {value, ms_per_step, frac (of the 20-flop vector roofline), ...}; the whole pass stops starting new entries once
    budget_s seconds are spent.  A failure is recorded in the entry, not raised."""
    t0 = time.perf_counter()
    res = {}

    def left():
        return budget_s - (time.perf_counter() - t0)

    def entry(key, make):
        if left() < 1.0:
            res[key] = {"value": None, "skipped": "the pass's %.0f-s budget was spent" % budget_s}
        else:
            try:
                res[key] = make()
            except Exception as ex:
                res[key] = {"value": None, "error": repr(ex)}
        publish("configs", dict(res))

    cpu_cache = {}

    def cpu_leg(n_, fp64_):
        """this entry's own host-CPU column: a <= 3-s row sample at the entry's N and precision (entries of one N and precision share it)"""
        key = (n_, bool(fp64_))
        if key not in cpu_cache:
            try:
                cpu_cache[key] = cpu_baseline(n_, args.seed, fp64_, 3.0)
            except Exception as ex:
                cpu_cache[key] = {"value": None, "unit": "billion pair-interactions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        return cpu_cache[key]

    def config2(variant, tile, iblock=0, jsub=0):
        def make():
            n2, steps = 65536, 100
            with nb.NBody(n2, tile=tile) as e:
                e.set_option(nb.OPT_VARIANT, variant)
                if iblock:
                    e.set_option(nb.OPT_IBLOCK, iblock)
                if jsub:
                    e.set_option(nb.OPT_JSUB, jsub)
                pos, vel = nb.make_bodies(n2, seed=args.seed)
                e.upload(pos, vel)
                r = run_timed(e, steps, 50, False)      # 50 untimed steps (~50 ms): clocks up after the engine switch
                cfg = e.config
                value = float(n2) * n2 * steps / r["elapsed"] / 1e9
                return {"workload": "N=%d fp32, %d steps, 1 GPU" % (n2, steps), "value": round(value, 2), "unit": "billion pair-interactions/s",
                        "ms_per_step": round(1e3 * r["elapsed"] / steps, 4), "frac": round(value * FLOP_PER_PAIR / 1e3 / PEAK_VECTOR_TFLOPS["f32"], 4),
                        "kernel": {k: cfg[k] for k in ("variant", "tile", "iblock", "nseg", "wsplit", "launches_per_step")},
                        "form": ("the engine's default" if variant == nb.VARIANT_AUTO else
                                 "the north_star's form — sources tiled into LDS, tile %d — in its best measured shape: %d bodies per lane, %d segments "
                                 "(profiles/r05_sweep_lds_n65536.txt)" % (tile, iblock, jsub)),
                        "cpu_baseline": cpu_leg(n2, False)}
        return make

    def fp64():
        n5, steps = 262144, 3
        with nb.NBody(n5, fp64=True) as e:
            pos, vel = nb.make_bodies(n5, seed=args.seed, dtype=np.float64)
            e.upload(pos, vel)
            r = run_timed(e, steps, 1, True)
            cfg = e.config
            roof = roofline_of(e, cfg, r, n5, True, True)
            value = float(n5) * n5 * steps / r["elapsed"] / 1e9
            return {"workload": "N=%d fp64, %d steps, 1 GPU" % (n5, steps), "value": round(value, 2), "unit": "billion pair-interactions/s",
                    "ms_per_step": round(1e3 * r["elapsed"] / steps, 3), "frac": roof["frac"], "peak_tflops": roof["peak"],
                    "frac_of_issue_bound": roof["frac_of_issue_bound"], "kernel_ms_avg": roof["kernel_ms_avg"],
                    "kernel": {k: cfg[k] for k in ("variant", "nseg", "wsplit", "launches_per_step")},
                    "cpu_baseline": cpu_leg(n5, True)}

    def config1():
        import re
        cpu, gpu = os.path.join(ROOT, "oracle", "nbody_cpu"), os.path.join(ROOT, "build", "nbody")
        missing = [os.path.relpath(x, ROOT) for x in (cpu, gpu) if not os.path.exists(x)]
        if missing:
            return {"value": None, "error": "missing %s: run `make host oracle`" % ", ".join(missing)}

        def run(cmd):
            o = subprocess.run(cmd, capture_output=True, text=True, timeout=max(5.0, left() + 5.0))
            if o.returncode:
                raise RuntimeError("%s exited with %d: %s" % (os.path.basename(cmd[0]), o.returncode, o.stderr[-300:]))
            rate = re.search(r"average ([0-9.]+) Billion Interactions / second \(([0-9.]+) ms / step\)", o.stdout)
            chk = [l for l in o.stdout.splitlines() if l.startswith("checksum")]
            return float(rate.group(1)), float(rate.group(2)), (chk[0] if chk else None), o.stdout
        # (a) the engine's own configuration in strict arithmetic, and the CPU program told to sum in the order the GPU program reports
        g_rate, g_ms, g_chk, g_out = run([gpu, "4096", "10", "--strict"])
        m = re.search(r"(\d+) segments x (\d+) pieces, sum block (\d+)", g_out)
        order = ["--sum", "blocked", "--segments", m.group(1), "--wsplit", m.group(2), "--block", m.group(3)]
        c_rate, c_ms, c_chk, c_threads = cpu_baseline_program(["4096", "10"] + order, max(5.0, left() + 5.0))
        # (b) one sequential sum per body — what a plain CPU nbody.c does — on both sides
        s_c_rate, s_c_ms, s_c_chk, _ = cpu_baseline_program(["4096", "10"], max(5.0, left() + 5.0))
        s_g_rate, s_g_ms, s_g_chk, _ = run([gpu, "4096", "10", "--strict", "--sum", "seq", "--jsub", "1", "--wsplit", "1"])
        return {"workload": "N=4096 fp32, 10 iterations (the first is warm-up), strict arithmetic, the engine's own summation order (%s segments x %s pieces, blocks of %s)" % m.groups(),
                "value": g_rate, "unit": "billion pair-interactions/s", "ms_per_step": g_ms,
                "gpu_program": "build/nbody 4096 10 --strict", "cpu_program": "oracle/nbody_cpu 4096 10 " + " ".join(order),
                "cpu_value": c_rate, "cpu_ms_per_step": c_ms, "checksum_gpu": g_chk, "checksum_cpu": c_chk,
                "cpu_baseline": {"value": c_rate, "unit": "billion pair-interactions/s", "cores": c_threads, "kind": "port",
                                 "sample": "oracle/nbody_cpu 4096 10 %s: the whole configuration, 9 timed iterations" % " ".join(order)},
                "sequential_sum": {"gpu_program": "build/nbody 4096 10 --strict --sum seq --jsub 1 --wsplit 1", "cpu_program": "oracle/nbody_cpu 4096 10",
                                   "value": s_g_rate, "ms_per_step": s_g_ms, "cpu_value": s_c_rate, "cpu_ms_per_step": s_c_ms,
                                   "checksum_gpu": s_g_chk, "checksum_cpu": s_c_chk, "checksums_equal": bool(s_g_chk and s_g_chk == s_c_chk)},
                "checksums_equal": bool(g_chk and g_chk == c_chk and s_g_chk and s_g_chk == s_c_chk)}

    entry("config2", config2(nb.VARIANT_AUTO, 0))
    entry("config2_lds_tile256", config2(nb.VARIANT_LDS, 256, iblock=4, jsub=32))
    entry("fp64", fp64)
    entry("config1", config1)
    res["seconds"] = round(time.perf_counter() - t0, 2)
    publish("configs", dict(res))
    return res


def strict_pass(eng, nb, n, dt, steps):
    """What the parity claim costs, on the box that timed the headline: the same engine switched to NBODY_ARITH_STRICT — every operation
    IEEE-exact, 1/sqrt = (float)(1.0 / sqrt((double)d2)): the arithmetic in which the GPU equals the CPU oracle bit for bit (tests -m gpu,
    smoke()) — for up to 3 steps after the timed region.  Never the headline; a failure is reported in the object, not raised."""
    try:
        eng.set_option(nb.OPT_ARITH, nb.ARITH_STRICT)
        eng.set_option(nb.OPT_TIMING, 0)
        # the GPU has sat idle through the host-CPU leg (10-15 s): at least one step and at least 0.1 s of steps bring its clocks back up
        # before anything is timed (one box measured 31 ms for a 1.4-ms step right after the idle period)
        t_w, t_step = time.perf_counter(), 0.0
        while True:
            t1 = time.perf_counter()
            eng.step(dt, 1)
            eng.sync()
            t_step = time.perf_counter() - t1
            if time.perf_counter() - t_w >= 0.1:
                break
        # up to 3 steps at the headline's size; short steps are repeated to fill ~30 ms
        k = max(1, min(int(steps), 3), min(200, int(0.03 / max(t_step, 1e-6))))
        t0 = time.perf_counter()
        eng.step(dt, k)
        eng.sync()
        el = time.perf_counter() - t0
        cfg = eng.config
        return {"arith": "NBODY_ARITH_STRICT", "value": round(float(n) * float(n) * k / el / 1e9, 2), "unit": "billion pair-interactions/s",
                "steps": k, "ms_per_step": round(1e3 * el / k, 3),
                "kernel": {key: cfg[key] for key in ("variant", "nseg", "wsplit", "sum_order", "sum_block", "launches_per_step")},
                "note": "bit-identical to the CPU oracle in this summation order (tests/test_gpu_parity.py); not the timed mode"}
    except Exception as e:       # the headline is reported in any case
        return {"arith": "NBODY_ARITH_STRICT", "value": None, "error": repr(e)}


def comm_forms_pass(eng, nb, args, n, transport, run_timed, publish, steps=3):
    """SURVEY.md §8(f) rank 4 measured where it can be: three steps each of the north_star's ring (P-1 dependent groups, one
    force launch per arriving slice), the DIRECT group (one hop over all links) and ncclAllGather, same engine, same state.
    Transports without forms (peer copies, host-staged) give one entry.  Every rank takes part (eng None: collectives only)."""
    if transport == "rccl":
        # in ascending order of novelty: the all-gather is what the headline just ran; the ring's P-1 dependent groups come last
        forms = [("allgather", nb.COMM_ALLGATHER, 1), ("direct", nb.COMM_DIRECT, 1), ("ring", nb.COMM_RING, 2),
                 # ... and the all-gather with NO overlap (gather first, then one launch): its comm_exposed_ms_per_step IS the transfer's
                 # duration on this job's links, the number the overlapped forms are hiding
                 ("allgather_gather_first", nb.COMM_ALLGATHER, 0)]
    else:
        forms = [("peer" if transport.startswith("peer") else "host", None, args.overlap)]
    res = {}
    for name, comm, overlap in forms:
        if eng is not None and comm is not None:
            eng.set_option(nb.OPT_COMM, comm)
            eng.set_option(nb.OPT_OVERLAP, overlap)
        r = run_timed(eng, steps, 1, True)
        entry = {"ms_per_step": round(1e3 * r["elapsed"] / steps, 3), "comm_exposed_ms_per_step": round(r["wait_ms"] / steps, 4),
                 "value": round(float(n) * float(n) * steps / r["elapsed"] / 1e9, 2), "overlap": overlap, "steps": steps}
        if eng is not None and comm is not None:
            entry["form_resolved"] = COMM_NAMES.get(eng.info(nb._lib.INFO_COMM_FORM), "?")     # allgather needs equal slices: ring otherwise
        res[name] = entry
        publish("comm_forms", dict(res))
    return res        # (the engine is closed right after this pass: its options are not restored)


def config5_pass(nb, args, world, rank, open_engine, run_timed, roofline_of, np, steps=2):
    """BASELINE configs[4]: N = 4,194,304 fp64 sharded over the job's GPUs, 1 warm-up + 2 timed steps, in the engine's own
    configuration.  "HBM GB/s vs peak" (the config's own words) = the algorithmic bytes of a step — 128 B per owned body: its
    position and velocity read and written — over the step time; the path stays VALU-bound."""
    n5 = args.config5_bodies
    e5 = open_engine(n5, True)
    try:
        if e5 is not None:
            pos, vel = nb.make_bodies(n5, seed=args.seed, dtype=np.float64)
            e5.upload(pos, vel)
            del pos, vel
        r = run_timed(e5, steps, 1, True)
        res = None
        if rank == 0:
            cfg = e5.config
            roof = roofline_of(e5, cfg, r, n5, True, True)
            ms = 1e3 * r["elapsed"] / steps
            bytes_per_step = cfg["n_local"] * 128.0
            res = {"workload": "N=%d fp64 over %d GPU(s), %d timed steps" % (n5, world, steps),
                   "value": round(float(n5) * float(n5) * steps / r["elapsed"] / 1e9, 2), "unit": "billion pair-interactions/s",
                   "ms_per_step": round(ms, 3),
                   "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms_avg", "kernel_launches", "frac_of_issue_bound")},
                   "hbm_gb_per_s": round(bytes_per_step / (ms * 1e-3) / 1e9, 3), "hbm_frac_of_peak": round(bytes_per_step / (ms * 1e-3) / 8e12, 7),
                   "hbm_note": "algorithmic bytes per GPU and step (128 B per owned body) over the step time; peak 8 TB/s",
                   "comm_exposed_ms_per_step": round(r["wait_ms"] / steps, 4),
                   "kernel": {k: cfg[k] for k in ("variant", "nseg", "wsplit", "launches_per_step", "n_local")}}
        return res
    finally:
        if e5 is not None:
            e5.close()


if __name__ == "__main__":
    main()
