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

Rank 0 prints ONE JSON line, always with:
  roofline      the force kernel priced at 20 flop per pair against the fp32 (157.3 TFLOP/s) or
                fp64 (78.6) VECTOR peak — the path has no contraction for the matrix cores — with the kernel's
                duration measured live by HIP events on the library's compute stream; beside it the instruction-issue
                bound (30 cycles per wave-pair in fp32, 80 in fp64) and cycles per wave-pair.  `traffic` and the
                other *_pmc fields come from the committed rocprofv3 passes of THIS configuration
                (profiles/pmc_*.json, tools/profile.sh) and are attached only when that profile's recorded
                configuration equals the run's; otherwise `traffic` is null.
  cpu_baseline  the oracle (oracle/nbody_ref.c, kind "port": the reference is VHDL and has no CPU path) timed on
                this box's host cores, rank 0, after the timed region, on a bounded row sample.
"""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process device memory on this pool needs dmabuf IPC (the launcher normally exports it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FLOP_PER_PAIR = 20                # SURVEY.md §8(d) convention (literal count: 18)
PEAK_VECTOR_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 (32) flop/clk x 2.4 GHz
# cycles of VALU issue per wave64 pair: fp32 11 x 2 + 8 (v_rsq_f32); fp64 16 x 4 + 16 (v_rsq_f64)
# measured: profiles/r01_microbench_valu_issue.txt, DESIGN.md §3
ISSUE_CYCLES_PER_WAVE_PAIR = {"f32": 30, "f64": 80}
METRIC = "billion pair-interactions/s at N=1M fp32; 1/2/4/8 GPUs + % FP32 roofline"


def cpu_baseline(n, seed, fp64):
    """The oracle timed on the host cores: a row sample (first rows x all N sources), sized to take
    roughly 10-20 s.  Returns the JSON object."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
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
    cores = ora.num_threads()
    nb = importlib.import_module("mini-nbody_amd")
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
    target_s = 15.0
    rows2 = int(min(n, max(rows, (rate * target_s / n) // 16 * 16)))
    if rows2 > rows:
        t0 = time.perf_counter()
        run(rows2)
        t = time.perf_counter() - t0
        rows = rows2
    what = "fp64, sequential-j, 1.0/sqrt" if fp64 else "fp32, sequential-j, 1.0f/sqrtf"
    return {"value": round(rows * n / t / 1e9, 3), "unit": "billion pair-interactions/s", "cores": cores, "kind": "port",
            "sample": "oracle/nbody_ref.c (%s), first %d of %d rows x all %d sources, %.1f s, gcc -O3 %s -fopenmp"
                      % (what, rows, n, n, t, "-march=native" if path else "-march=x86-64-v3")}


def kernel_source_sha():
    """identifies the kernel code a profile was taken on (a profile of an older loop must not be attached to a new one)"""
    import hashlib
    h = hashlib.sha1()
    for f in ("nbody_kernels.hpp", "force_loop_gfx950.inc", "nbody_hip.hip"):
        h.update(open(os.path.join(ROOT, "mini-nbody_amd", "csrc", f), "rb").read())
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


def main():
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", choices=["auto", "ring", "allgather", "direct"], default="auto")
    ap.add_argument("--transport", choices=["auto", "rccl", "host"], default="auto")
    ap.add_argument("--xcd-map", type=int, default=-1, help="XCD-aware placement of source segments: 1 on, 0 off, -1 the engine's default")
    ap.add_argument("--overlap", type=int, default=1, help="0 gather first, 1 own slice then the rest, 2 one launch per arriving slice")
    args = ap.parse_args()

    if not os.path.exists(os.path.join(ROOT, "mini-nbody_amd", "libnbody_hip.so")):
        # fresh checkout (built files are git-ignored): build the HIP library; a failure is fatal, there is no fallback
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            subprocess.run(["make", "lib"], cwd=ROOT, check=True, capture_output=True)
        else:
            for _ in range(600):
                if os.path.exists(os.path.join(ROOT, "mini-nbody_amd", "libnbody_hip.so")):
                    break
                time.sleep(0.5)
            time.sleep(2.0)
    import torch
    nb = importlib.import_module("mini-nbody_amd")
    D = importlib.import_module("mini-nbody_amd.distributed")

    rank, world, local = D.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local % torch.cuda.device_count())
    os.environ.setdefault("NBODY_DEVICE", str(local % torch.cuda.device_count()))
    import torch.distributed as dist
    if world > 1:
        D.init_process_group(backend="gloo")   # control plane only; the data path is RCCL inside the library

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.n
    eng = D.make_engine(n, fp64=args.fp64, tile=args.tile, transport=args.transport)
    eng.set_option(nb.OPT_VARIANT, {"auto": nb.VARIANT_AUTO, "smem": nb.VARIANT_SMEM, "lds": nb.VARIANT_LDS,
                                    "readlane": nb.VARIANT_READLANE, "isa": nb.VARIANT_ISA}[args.variant])
    if args.isa_phase >= 0:
        eng.set_option(nb.OPT_ISA_PHASE, args.isa_phase)
    eng.set_option(nb.OPT_IBLOCK, args.iblock)
    eng.set_option(nb.OPT_JSUB, args.jsub)
    eng.set_option(nb.OPT_SUM_ORDER, nb.SUM_BLOCKED if args.sum == "blocked" else nb.SUM_SEQ)
    if args.sum_block > 0:
        eng.set_option(nb.OPT_SUM_BLOCK, args.sum_block)
    eng.set_option(nb.OPT_FUSE_COMBINE, args.fuse)
    if args.xcd_map >= 0:
        eng.set_option(nb.OPT_XCD_MAP, args.xcd_map)
    eng.set_option(nb.OPT_COMM, {"auto": nb.COMM_AUTO, "ring": nb.COMM_RING, "allgather": nb.COMM_ALLGATHER, "direct": nb.COMM_DIRECT}[args.comm])
    eng.set_option(nb.OPT_OVERLAP, args.overlap)
    import numpy as np
    pos, vel = nb.make_bodies(n, seed=args.seed, dtype=np.float64 if args.fp64 else np.float32)
    eng.upload(pos, vel)                      # inputs resident in HBM before the timed region
    dt = 0.01

    eng.step(dt, args.warmup)
    eng.sync()
    eng.set_option(nb.OPT_TIMING, 1)
    eng.kernel_time(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(dt, args.steps)                  # EXACTLY K steps
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_time(reset=True)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    cfg = eng.config
    # sanity of the state the timed steps produced: this rank's own slice, without any collective (nothing after the
    # timed region may stall the report)
    p_own, _ = eng.download_slice()
    finite = bool(np.isfinite(p_own).all())

    if rank == 0:
        dtype = "f64" if args.fp64 else "f32"
        pairs_per_step = float(n) * float(n)
        value = pairs_per_step * args.steps / elapsed / 1e9
        n_local = cfg["n_local"]
        # force kernel: per launch this rank's share of the pairs; duration from HIP events on the compute stream
        launches_per_step = max(1, launches // max(1, args.steps))
        pairs_per_launch = float(n_local) * float(n) / launches_per_step
        avg_launch_s = kernel_ms * 1e-3 / max(1, launches)
        kernel_rate = pairs_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0   # pairs/s on one GPU
        achieved_tflops = kernel_rate * FLOP_PER_PAIR / 1e12
        peak = PEAK_VECTOR_TFLOPS[dtype]
        cu, clk = eng.info(nb._lib.INFO_CU_COUNT), eng.info(nb._lib.INFO_CLOCK_KHZ) * 1e3
        simds = cu * 4
        issue_bound = simds * 64.0 / ISSUE_CYCLES_PER_WAVE_PAIR[dtype] * clk   # pairs/s at the nominal clock
        wave_pairs_per_launch = pairs_per_launch / 64.0
        run_cfg = {"n": n, "dtype": dtype, "n_gpus": world, "variant": cfg["variant"], "iblock": cfg["iblock"], "nseg": cfg["nseg"],
                   "sum_order": cfg["sum_order"], "sum_block": cfg["sum_block"], "launches_per_step": cfg["launches_per_step"],
                   "kernel_source_sha": kernel_source_sha()}
        roof = {"bound": "valu", "achieved": round(achieved_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved_tflops / peak, 4), "traffic": None, "flop_per_pair": FLOP_PER_PAIR,
                "kernel_ms_avg": round(avg_launch_s * 1e3, 4), "kernel_launches": launches,
                "kernel_gpairs_per_s": round(kernel_rate / 1e9, 1),
                "algorithmic_flops_per_launch": pairs_per_launch * FLOP_PER_PAIR,
                "algorithmic_hbm_bytes_per_launch": n_local * (32 if args.fp64 else 16) * 4 / launches_per_step,
                "issue_cycles_per_wave_pair_model": ISSUE_CYCLES_PER_WAVE_PAIR[dtype],
                "issue_bound_gpairs_per_s": round(issue_bound / 1e9, 1),
                "frac_of_issue_bound": round(kernel_rate / issue_bound, 4) if issue_bound else None,
                "cycles_per_wave_pair_at_nominal_clock": round(avg_launch_s * clk * simds / wave_pairs_per_launch, 2) if wave_pairs_per_launch else None,
                "note": "VALU-issue-bound: per pair 11 full-rate + 1 quarter-rate instruction in fp32 (30 cycles per wave64), "
                        "16 + 1 in fp64 (80); neither HBM nor MFMA bounds it (no contraction; HBM traffic is 64 B per body per step)"}
        pj = matching_pmc(run_cfg)
        if pj:
            roof["traffic"] = pj.get("hbm_bytes_per_launch")
            roof["pmc"] = {"source": "%s (rocprofv3 passes of this configuration, not this run)" % pj["_file"],
                           "clock_ghz": round(pj.get("clock_ghz", 0.0), 3),
                           "cycles_per_wave_pair": round(pj.get("cycles_per_wave_pair", 0.0), 2),
                           "hbm_gb_per_s": round(pj.get("hbm_gb_per_s", 0.0), 2),
                           "hbm_frac_of_peak": round(pj.get("hbm_frac_of_8tbs", 0.0), 5),
                           "kernel_ms_avg_trace": pj.get("force_kernel_avg_ms_trace")}
        out = {
            "metric": METRIC if (n == (1 << 20) and not args.fp64) else "billion pair-interactions/s at N=%d %s" % (n, "fp64" if args.fp64 else "fp32"),
            "value": round(value, 2), "unit": "billion pair-interactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "N=%d %s all-pairs softened gravity, leapfrog kick-drift, dt=0.01, seed %d"
                                   % (n, "fp64" if args.fp64 else "fp32", args.seed),
                       "n_bodies": n, "pairs_per_step": pairs_per_step, "parallelism": "bodies sharded over %d GPU(s)" % world,
                       "kernel": cfg, "kernel_source_sha": kernel_source_sha(), "comm": (args.comm + " / " + getattr(eng, "transport", "rccl") + " / overlap %d" % args.overlap) if world > 1 else None,
                       "finite": finite},
            "roofline": roof,
        }
    eng.close()
    if rank == 0:
        if not args.no_cpu_baseline:
            # after the timed region and with the GPU context closed; the other ranks wait at the barrier below, idle
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.seed, args.fp64)
            except Exception as e:    # the GPU result is reported in any case
                out["cpu_baseline"] = {"value": None, "unit": "billion pair-interactions/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
