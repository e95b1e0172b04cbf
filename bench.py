#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the device-resident loop on BASELINE.json's configs; the default is the headline one
(dense n = 4096, m = 8192, fp64, one QP per GPU; ranks run independent replicas, no data-path collective).

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU, RCCL for the timing gather)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one qps_solve() of --iters ADMM iterations (default 500; ϵAbs = ϵRel = 0 so exactly that many run) on a problem already
resident in HBM with its factorisation cached (setup is reported separately).  Rank 0 prints ONE JSON line carrying
`roofline` (dominant kernel, HIP events attached to its dispatches on the solver's stream, inside the timed region) and
`cpu_baseline` (the oracle's C restatement on this box's host cores, bounded sample; rank 0 at N = 1 only).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEEP = {}                  # the headline problem, kept for the fp32 leg of `side_configs`
LAST_TRAFFIC_META = {}     # filled by pmc_traffic(): build identity of the PMC summary the last traffic figure came from
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
HBM_ACHIEVABLE_GBS = 6290.0

CONFIGS = {
    "c2": dict(n=4096, m=8192, dtype="f64", label="dense n=4096 m=8192 fp64 single QP (BASELINE configs[1])"),
    "c5": dict(n=4096, m=8192, dtype="f32", label="dense n=4096 m=8192 fp32, rho-update + refactor every 50 its (configs[4])"),
    "c1": dict(n=64, m=128, dtype="f64", label="dense n=64 m=128 fp64 (configs[0] shape, plumbing)"),
    "c3": dict(n=50000, m=100000, dtype="f64", label="sparse P,A n=50k m=100k ~0.1% nnz fp64, CSR SpMV + matrix-free CG (configs[2])"),
    "c4": dict(n=1024, m=2048, dtype="f64", batch=256, label="batch of 256 dense n=1024 m=2048 QPs sharded across the ranks (configs[3])"),
}
SWEEP_VARIANTS = {0: None, 1: "blocked substitution (2n/nb-1 dependent phases per sweep)", 2: "explicit inverse, both sweeps fused into one pass over the triangle",
                  3: "explicit inverse, two triangular GEMVs", 4: "single-launch small-problem loop",
                  5: "blocked substitution, one launch per sweep (n/nb dependent phases handed over inside the launch)"}


def spawn_ranks_if_needed(args, argv):
    """`python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed.run on 127.0.0.1) when it was
    not launched under torchrun.  Runs BEFORE torch or the HIP library is imported: a process that has touched the GPU is never
    re-executed; the ranks are children and this process only waits and exits with their code.  Rank 0's JSON line reaches stdout
    through the inherited descriptor."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: launch N ranks for --gpus N")
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=0, help="ADMM iterations per step (default 500; 100 for c3 / c4; 2000 for c1)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--trsv-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-eps", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="default command only (config c2, one GPU): skip the GPU legs of c5 / c4 / c3 that are appended as `side_configs`")
    ap.add_argument("--profile-level", type=int, default=1,
                    help="1: HIP events on sampled dispatches of the loop kernels (well under 1 %% overhead); 2: every launch of every kernel")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="rank plumbing only (rendezvous, barrier, timing reduction) with a sleep in place of the solver: runs without a GPU")
    args = ap.parse_args()
    spawn_ranks_if_needed(args, sys.argv[1:])

    import torch  # first: keeps a single HIP runtime in the process (torch bundles its own libamdhip64)
    import numpy as np
    from quadraticprogramsolver_amd import dist as qd

    info = qd.rank_info_from_env()
    if info.world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={info.world_size}")
    ndev = torch.cuda.device_count()       # does not initialise the GPU
    # one rank per GPU -> the barrier and the timing reduction go through RCCL (backend "nccl"); ranks that have to share a card
    # (more ranks than devices, e.g. a rehearsal on a 1-GPU box) or a CPU-only selftest use gloo
    backend = qd.init_process_group(info, prefer=os.environ.get("QPS_DIST_BACKEND") or ("nccl" if ndev >= info.world_size and ndev > 0 else "gloo"))
    device = info.local_rank % ndev if ndev > 0 else 0
    if args.launcher_selftest:
        return launcher_selftest(args, info, backend, device, ndev, qd)
    import quadraticprogramsolver_amd as qps
    cfg = CONFIGS[args.config]
    if args.iters <= 0:
        args.iters = {"c3": 100, "c4": 100, "c1": 2000}.get(args.config, 500)
    if torch.cuda.is_available():
        torch.cuda.set_device(device)      # torch.cuda.synchronize() below must act on this rank's GPU, not on GPU 0
    sync = (lambda: torch.cuda.synchronize()) if torch.cuda.is_available() else (lambda: None)
    run = {"c2": run_dense, "c5": run_dense, "c1": run_dense, "c3": run_sparse, "c4": run_batch}[args.config]
    out = run(args, cfg, info, device, qps, qd, np, sync)
    if args.config == "c2" and info.world_size == 1 and not args.no_side_configs and args.trsv_block == 0:
        # every BASELINE config in the one driver-run line (RunBenchmarks.jl:88-104: one run covers every test): GPU legs only, after the
        # headline's timed region, each a shortened run of exactly what `bench.py --config cX` measures
        out["side_configs"] = side_configs(args, info, device, qps, qd, np, sync)
    if info.rank == 0:
        out["config"]["dist_backend"] = backend
        out["config"]["ranks"] = info.world_size
        out["config"]["devices_visible"] = ndev
        print(json.dumps(out), flush=True)
    qd.shutdown(info)


def base_line(args, cfg, info, value, tmax, scaling, metric="ADMM iterations/sec"):
    return {"metric": metric, "value": round(value, 2), "unit": "iterations/s", "n_gpus": info.world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(tmax / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic", "config": {"workload": cfg["label"], "n": cfg["n"], "m": cfg["m"], "admm_iterations_per_step": args.iters}}


# Translation units that hold only the sparse solvers (their own object files: no kernel of a dense handle is compiled from them).  A summary of a DENSE config
# measured on a tree that differs from the running one in these files alone is still this build's figure: the object files with the dense kernels were built
# from byte-identical sources, headers and flags.  Such a match is reported as what it is (`traffic_match`: "dense translation units").
SPARSE_ONLY_UNITS = ("k_sparse.hip", "k_ldl.hip", "ldl_symbolic.cpp", "spmv_layout.cpp")
DENSE_CONFIGS = ("c1", "c2", "c2_trsv1024", "c4", "c5")


def csrc_digest(exclude=(), read=None, listing=None):
    """Identity of the kernel sources the running library was built from: sha256 over the sorted files of csrc/ and include/qps.h (`exclude`: file names left out).
    The PMC summaries under profiles/ carry the digest of the tree they were measured on; a figure measured on other sources is not reported.
    `read` / `listing` let a tool hash another tree (a git commit) with the same rule."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "quadraticprogramsolver_amd", "csrc")
    names = os.listdir(base) if listing is None else listing
    read = read or (lambda rel: open(os.path.join(base, rel), "rb").read())
    files = sorted(f for f in names if (f.endswith((".hip", ".h", ".cpp")) or f == "Makefile") and f not in exclude)
    for f in files + [os.path.join("..", "..", "include", "qps.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(read(f))
    return h.hexdigest()[:16]


def digest_matches(meta, config):
    """None when the summary's build is not the running one; else how it matches: "whole tree", or -- dense configs only -- "dense translation units"."""
    if meta.get("csrc_sha16") == csrc_digest():
        return "whole tree"
    if config in DENSE_CONFIGS and meta.get("csrc_dense_sha16") and meta.get("csrc_dense_sha16") == csrc_digest(exclude=SPARSE_ONLY_UNITS):
        return "dense translation units (the running tree differs from the measured one at most in " + ", ".join(SPARSE_ONLY_UNITS) + ": sparse solvers only)"
    return None


def build_head():
    """git HEAD of the tree (the GPU box has no .git: tests/tools/gpu.sh stamps .build_head before shipping the snapshot)."""
    try:
        import subprocess
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()[:12]
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, ".build_head")).read().strip()[:12] or None
    except Exception:
        return None


def roofline_of(kernel_label, seconds, launches, algo_bytes_per_launch, traffic, traffic_src, extra=None):
    dur = seconds / launches
    ach = algo_bytes_per_launch / dur / 1e9
    r = {"bound": "hbm", "kernel": kernel_label, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(dur * 1e6, 2), "algo_bytes_per_launch": algo_bytes_per_launch,
         "launches_timed": launches, "timing": "HIP events attached to the sampled dispatches on the solver's stream, inside the timed region"}
    if traffic:
        r["traffic_over_algorithmic"] = round(traffic / algo_bytes_per_launch, 3)
        r["traffic_GBs"] = round(traffic / dur / 1e9, 1)
    r["traffic_head"] = LAST_TRAFFIC_META.get("head")            # build the PMC passes ran on (null: no matching summary)
    r["traffic_csrc_sha16"] = LAST_TRAFFIC_META.get("csrc_sha16")
    r["running_csrc_sha16"] = csrc_digest()
    if LAST_TRAFFIC_META.get("match"):
        r["traffic_match"] = LAST_TRAFFIC_META["match"]
    if LAST_TRAFFIC_META.get("stale"):
        r["traffic_note"] = LAST_TRAFFIC_META["stale"]
    if extra:
        r.update(extra)
    return r


# rocprofv3 kernel names of the profiler categories (for the side-by-side averages in `kernels[]`)
ROCPROF_NAME = [(r"^apass\(fused A-pass", r"k_apass<.*false, 0>"), (r"^apass\(check variant", r"k_apass<.*true, 0>"),
                (r"^sweeps\(fused", r"k_sweep_fused(_wave)?<"), (r"^colsum\(", r"k_colsum<"), (r"^trsv_forward", r"k_trsv_blocked<.*false, \d+>"),
                (r"^trsv_backward", r"k_trsv_blocked<.*true, \d+>"), (r"^spmv_blk\(", r"k_spmv_(sell|blk)<"), (r"^admm_small", r"k_admm_small")]


def rocprof_stats(config):
    """Average kernel durations from the committed `rocprofv3 --kernel-trace --stats` summary of this config's command
    (profiles/rNN_*bench_<config>_kernel_stats.csv, newest), only when it was taken on the sources now running (side-car .meta.json)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*bench_{config}_kernel_stats.csv")))
    for f in reversed(files):
        try:
            meta = json.load(open(f[:-4] + ".meta.json"))
        except Exception:
            continue
        if not digest_matches(meta, config):
            continue
        rows = {}
        for r in csv.DictReader(open(f)):
            rows[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
        return rows, os.path.basename(f)
    return {}, None


def kernel_list(ktimes, config=None, iters_per_s=None, grid_y=None):
    """Per-kernel averages.  `avg_us` = HIP events attached to the sampled dispatches (they read ~1-2 us longer per launch than the kernel's own
    duration: the sum over an iteration's launches can exceed the wall time per iteration); `rocprof_avg_us` = the same kernel in the committed
    rocprofv3 kernel-trace summary of this command, when one exists for the running sources.  The summaries are made by tests/tools/trace_stats_split.py:
    a kernel launched with several grids has one row per grid and the dispatches that returned at the CG `done` flag sit in a row of their own, so a figure
    printed here is the average of launches that did this category's work.  grid_y: {category prefix: gridDim.y} picks the row of a category whose kernel
    serves several categories (the two products of a CG iteration); a category that cannot be told apart gets no rocprof figure rather than a joint one."""
    rows, src = rocprof_stats(config) if config else ({}, None)
    out = []
    for k in ktimes:
        e = {"name": k["name"], "avg_us": round(k["seconds"] / k["launches"] * 1e6, 2), "launches": k["launches"],
             "GBs": round(k["algo_bytes"] / (k["seconds"] / k["launches"]) / 1e9, 1) if k["algo_bytes"] > 0 else None, "rocprof_avg_us": None}
        pat = next((rp for cp, rp in ROCPROF_NAME if re.search(cp, k["name"])), None)
        if pat and rows:
            hit = [(n, v) for n, v in rows.items() if re.search(pat, n) and "[returned at the done flag]" not in n]
            gy = next((v for pre, v in (grid_y or {}).items() if k["name"].startswith(pre)), None)
            if gy is not None and len(hit) > 1:
                hit = [(n, v) for n, v in hit if re.search(r"\[grid \d+ x %d x \d+\]" % gy, n)]
            if hit:
                name, (avg, calls) = max(hit, key=lambda t: t[1][1])
                e["rocprof_avg_us"] = round(avg, 2)
                e["_rocprof_kernel"] = name
        out.append(e)
    # several categories left on ONE row (no per-grid rows in the summary, or equal grids): that row is their joint average -- not printed as either's
    seen = {}
    for e in out:
        if e.get("_rocprof_kernel"):
            seen.setdefault(e["_rocprof_kernel"], []).append(e)
    for name, es in seen.items():
        if len(es) > 1:
            for e in es:
                e["rocprof_avg_us"] = None
                e["rocprof_note"] = f"{len(es)} categories run the same kernel with the same grid ({name.split('(')[0][:40]}): the summary cannot tell them apart"
    for e in out:
        e.pop("_rocprof_kernel", None)
    if src:
        out.append({"name": "_rocprof_source", "file": f"profiles/{src}", "made_by": "tests/tools/trace_stats_split.py (per-grid rows, done-flag no-ops apart)"})
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
# dense single-QP configs: c2 (headline), c5 (fp32 + refactor per check), c1 (plumbing shape, single-launch loop)
# ----------------------------------------------------------------------------------------------------------------------------------
def run_dense(args, cfg, info, device, qps, qd, np, sync, problem=None):
    n, m = cfg["n"], cfg["m"]
    s = 8 if cfg["dtype"] == "f64" else 4
    # synthetic input: randomQp at density 1.0, seed 1234, one independent stream per rank (weak scaling: replicas)
    P, q, A, l, u = problem if problem is not None else qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=info.rank)
    if args.config == "c2":
        KEEP["c2_problem"] = (P, q, A, l, u)       # c5 of `side_configs` is the same problem in fp32
    prob = qps.QuadraticProgram(P, q, A, l, u, dtype=cfg["dtype"], device=device)
    solve_kw = dict(numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, trsvBlock=args.trsv_block, reuseFactor=True)
    if args.config == "c5":
        solve_kw.update(adptΡ=True, fctrΡ=1.0, numItrConv=50, ρ=0.1)   # ρ proposal applied (=> refactor) at every check
    setup_info = {}
    x = np.zeros(n)
    prob.solve(x, **dict(solve_kw, numIterations=min(args.iters, 25), reuseFactor=False), info=setup_info)   # builds + caches the factor
    for _ in range(args.warmup):
        x = np.zeros(n)
        prob.solve(x, **solve_kw)
    prob.set_profiling(args.profile_level)
    iters_done, refactors, t_refactor, last = 0, 0, 0.0, {}
    qd.barrier(info); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = np.zeros(n); si = {}
        prob.solve(x, **solve_kw, info=si)
        iters_done += si["iterations"]; refactors += si["numRefactor"]; t_refactor += si["tRefactor"]; last = si
    sync(); qd.barrier(info)
    elapsed = time.perf_counter() - t0
    ktimes = prob.kernel_times()   # HIP-event times accumulated over the timed steps (reset by set_profiling)
    prob.set_profiling(0)
    value, tmax = qd.gather_timings(info, elapsed, iters_done)
    out = None
    if info.rank == 0:
        out = base_line(args, cfg, info, value, tmax, "weak")
        b_iter = s * (m * n + n * n) + s * (6 * n + 10 * m)       # SURVEY §8d algorithmic bytes per ADMM iteration
        small = next((k for k in ktimes if k["name"].startswith("admm_small")), None)
        dom = next((k for k in ktimes if k["name"].startswith("apass(fused")), None)
        if small:      # c1: the whole loop is one launch; its category carries bytes per ITERATION
            out["roofline"] = roofline_of(small["name"], small["seconds"], small["launches"], b_iter * iters_done / small["launches"], None, None,
                                          {"note": "latency-bound single-workgroup launch (one CU): the HBM roofline is quoted for completeness, the working set lives in registers / LDS"})
        elif dom:
            traffic, src = pmc_traffic(args.config, r"k_apass<.*false, 0>$")
            out["roofline"] = roofline_of(dom["name"], dom["seconds"], dom["launches"], dom["algo_bytes"], traffic, src)
        else:
            out["roofline"] = None
        per_gpu = value / info.world_size
        out["config"].update({"qps_per_gpu": 1, "parallelism": f"replicas x{info.world_size} (independent QPs, no collective)",
                              "trsv_block": last.get("trsvBlock"), "sweep_variant": SWEEP_VARIANTS.get(last.get("sweepVariant")),
                              "params": ("eps=0, adptRho off, rho=1, sigma=1e-6, alpha=1.6, numItrConv=25 (SolveQuadraticProgram.jl:15-17)" if args.config != "c5"
                                         else "fp32, eps=0, adptRho on, fctrRho=1, numItrConv=50, rho0=0.1: the proposal is applied (re-factorisation) at every check")})
        out["loop_roofline"] = {"algo_bytes_per_iteration": b_iter, "achieved_GBs": round(b_iter * per_gpu / 1e9, 1),
                                "frac_of_8TBs": round(b_iter * per_gpu / 1e9 / HBM_PEAK_GBS, 4), "frac_of_6.29TBs": round(b_iter * per_gpu / 1e9 / HBM_ACHIEVABLE_GBS, 4)}
        out["setup_ms"] = round(setup_info.get("tSetup", 0.0) * 1e3, 2)
        if refactors:
            out["refactor"] = {"count": refactors, "ms_each": round(t_refactor / refactors * 1e3, 3), "share_of_loop_time": round(t_refactor / elapsed, 3)}
        out["kernels"] = kernel_list(ktimes, args.config + (f"_trsv{args.trsv_block}" if args.trsv_block else ""))
        sw = next((k for k in ktimes if k["name"].startswith("sweeps(fused")), None)
        if sw:   # the fused kernel reads the triangle once; SURVEY §8d's figure for the two sweeps it replaces is s*(n(n+1) + 4n)
            dur = sw["seconds"] / sw["launches"]
            t_sw, src_sw = pmc_traffic(args.config, r"k_sweep_fused<")
            out["sweep_roofline"] = {"kernel": sw["name"], "avg_launch_us": round(dur * 1e6, 2), "kernel_bytes": sw["algo_bytes"],
                                     "GBs": round(sw["algo_bytes"] / dur / 1e9, 1), "frac_of_8TBs": round(sw["algo_bytes"] / dur / 1e9 / HBM_PEAK_GBS, 4),
                                     "two_sweep_algo_bytes": s * (n * (n + 1) + 4 * n), "two_sweep_algo_GBs": round(s * (n * (n + 1) + 4 * n) / dur / 1e9, 1),
                                     "traffic": t_sw, "traffic_frac_of_8TBs": round(t_sw / dur / 1e9 / HBM_PEAK_GBS, 4) if t_sw else None, "traffic_source": src_sw}
        bl = [k for k in ktimes if k["name"] in ("trsv_forward", "trsv_backward") and k["launches"] > 0]
        if bl and last.get("sweepVariant") == 5:
            # --trsv-block 1024 (fp64) / 2048 (fp32): the back-substitution kernel north_star names.  One launch per sweep; SURVEY §8d's
            # per-sweep figure s*(n(n+1)/2 + 2n) is what each launch has to move.  PMC traffic: profiles/r*pmc_traffic_<config>_trsv<nb>.json
            ent = []
            for k in bl:
                dur = k["seconds"] / k["launches"]
                t_k, src_k = pmc_traffic(f"{args.config}_trsv{last.get('trsvBlock')}", r"k_trsv_blocked<.*" + ("true" if k["name"].endswith("backward") else "false") + r", \d+>")
                ent.append({"kernel": f"k_trsv_blocked ({k['name']})", "avg_launch_us": round(dur * 1e6, 2), "launches_timed": k["launches"], "algo_bytes": k["algo_bytes"],
                            "GBs": round(k["algo_bytes"] / dur / 1e9, 1), "frac_of_8TBs": round(k["algo_bytes"] / dur / 1e9 / HBM_PEAK_GBS, 4),
                            "traffic": t_k, "traffic_frac_of_8TBs": round(t_k / dur / 1e9 / HBM_PEAK_GBS, 4) if t_k else None, "traffic_source": src_k})
            out["sweep_roofline"] = {"variant": SWEEP_VARIANTS[5], "trsv_block": last.get("trsvBlock"), "dependent_phases_per_sweep": -(-n // last.get("trsvBlock")),
                                     "sweeps": ent, "target": ">= 0.40 of the 8 TB/s HBM roofline on the back-substitution kernel (BASELINE north_star)"}
        if not args.no_time_to_eps and args.config in ("c2", "c5"):
            # time-to-eps on the feasible variant (the plain m = 2n draw is primal infeasible: see generator docstring); fp32 to 1e-4
            Pf, qf, Af, lf, uf = qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=info.rank, feasible=True)
            eps = 1e-6 if cfg["dtype"] == "f64" else 1e-4
            with qps.QuadraticProgram(Pf, qf, Af, lf, uf, dtype=cfg["dtype"], device=device) as pf:
                xf = np.zeros(n); ti = {}
                t1 = time.perf_counter()
                flag = pf.solve(xf, numIterations=50000, ϵAbs=eps, ϵRel=eps, ρ=0.1, adptΡ=True, trsvBlock=args.trsv_block, info=ti)
                out["time_to_eps"] = {"eps": eps, "rho0": 0.1, "adptRho": True, "flag": int(flag), "iterations": ti["iterations"],
                                      "refactorisations": ti["numRefactor"], "ms_total": round((time.perf_counter() - t1) * 1e3, 2),
                                      "ms_setup": round(ti["tSetup"] * 1e3, 2), "ms_loop": round(ti["tLoop"] * 1e3, 2),
                                      "problem": "randomQp density 1.0, bounds centred on A*x0 (feasible variant)"}
            if not args.no_cpu_baseline and info.world_size == 1:
                # BASELINE.md §3 / RunTests.jl:50-58,93: the CPU restatement runs the identical problem with the identical parameters to the same
                # tolerance; its wall time stands beside the GPU's, its x / flag / counts are the parity check of the run just timed
                out["time_to_eps"].update(cpu_time_to_eps(np, (Pf, qf, Af, lf, uf), eps, xf, int(flag), ti, fp32=cfg["dtype"] != "f64"))
            del Pf, Af
        out["cpu_baseline"], out["parity"] = None, None
        if not args.no_cpu_baseline and info.world_size == 1:
            # the oracle solved this very problem for a fixed K (the cpu_baseline sample): run the HIP path for the same K once, outside the
            # timed region, and put the two side by side (SolveQuadraticProgram.jl:45-71; iterate-level tolerance of SURVEY §8c, fp32: 1e-3)
            out["cpu_baseline"], ref = cpu_baseline_dense(P, q, A, l, u, args.config)
            xg = np.zeros(n); gi = {}
            prob.solve(xg, **dict(solve_kw, numIterations=ref["K"]), info=gi)
            zg, yg = prob.dual()
            tol = 1e-9 if cfg["dtype"] == "f64" else 1e-3
            out["parity"] = parity_block(np, ref, xg, zg, yg, gi, tol, "oracle/qps_oracle.c (fp64 C restatement of the reference loop), same problem, same parameters"
                                         + (", same refactor-per-check schedule; HIP path in fp32" if args.config == "c5" else ""))
    prob.close()
    return out


def side_configs(args, info, device, qps, qd, np, sync):
    """GPU legs of BASELINE configs[4] (c5), configs[3] (c4, all 256 QPs on this GPU) and configs[2] (c3), run after the headline's timed
    region with the very functions `bench.py --config cX` runs (shorter: fewer steps, no CPU baseline, no time-to-eps)."""
    res = {"note": "GPU legs only, measured after the headline timed region by the code path of `bench.py --config cX` (fewer steps); the full lines "
                   "with cpu_baseline / parity / time_to_eps are profiles/rNN_*_bench_cX.json"}
    t_all = time.perf_counter()
    for name, steps in (("c5", 6), ("c4", 3), ("c3", 5)):
        t0 = time.perf_counter()
        try:
            sa = argparse.Namespace(**vars(args))
            sa.config, sa.steps, sa.warmup, sa.iters = name, steps, 1, {"c3": 100, "c4": 100}.get(name, 500)
            sa.no_cpu_baseline = sa.no_time_to_eps = True
            cfg = CONFIGS[name]
            if name == "c5":
                o = run_dense(sa, cfg, info, device, qps, qd, np, sync, problem=KEEP.pop("c2_problem", None))
            elif name == "c4":
                o = run_batch(sa, cfg, info, device, qps, qd, np, sync)
            else:
                o = run_sparse(sa, cfg, info, device, qps, qd, np, sync)
            rf, lr = o.get("roofline") or {}, o.get("loop_roofline") or {}
            e = {"workload": cfg["label"], "value": o["value"], "unit": o["unit"], "metric": o["metric"], "dtype": o["dtype"], "steps": steps, "warmup": 1,
                 "admm_iterations_per_step": sa.iters, "ms_per_step": o["ms_per_step"],
                 "roofline": {k: rf.get(k) for k in ("kernel", "frac", "achieved", "avg_launch_us", "algo_bytes_per_launch", "traffic", "traffic_head")},
                 "loop_roofline_frac_of_8TBs": lr.get("frac_of_8TBs"), "refactor": o.get("refactor"), "sweep_variant": o["config"].get("sweep_variant"),
                 "params": o["config"].get("params")}
            if name == "c3":
                e["cg_iterations_per_s"] = o.get("cg_iterations_per_s")
            if name == "c4":
                e["qps_on_this_gpu"] = o["config"].get("qps_on_rank0")
            e["wall_s"] = round(time.perf_counter() - t0, 1)
            res[name] = e
        except Exception as ex:          # a side config must never take the headline line down with it
            res[name] = {"error": f"{type(ex).__name__}: {ex}"[:300], "wall_s": round(time.perf_counter() - t0, 1)}
    KEEP.clear()
    res["wall_s"] = round(time.perf_counter() - t_all, 1)
    return res


def parity_block(np, ref, x, z, y, gi, tol, against, extra=None):
    """max relative deviation (max|a - b| / max(1, max|b|)) of the HIP path's x / z / y after K iterations from the oracle's."""
    rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) if b.size else 0.0
    dx, dz, dy = rel(x, ref["x"]), rel(z, ref["z"]), rel(y, ref["y"])
    same = (gi.get("iterations") == ref["iterations"]) and (gi.get("numRefactor", 0) == ref["numRefactor"])
    blk = {"K": ref["K"], "max_rel_dev_x": dx, "z": dz, "y": dy, "tolerance": tol, "ok": bool(max(dx, dz, dy) <= tol and same),
           "iterations": [gi.get("iterations"), ref["iterations"]], "refactorisations": [gi.get("numRefactor", 0), ref["numRefactor"]], "against": against}
    if extra:
        blk.update(extra)
    return blk


# ----------------------------------------------------------------------------------------------------------------------------------
# c3: CSR / matrix-free CG, replicas
# ----------------------------------------------------------------------------------------------------------------------------------
def run_sparse(args, cfg, info, device, qps, qd, np, sync):
    n, m = cfg["n"], cfg["m"]
    P, q, A, l, u = qps.GenerateSparseBenchmarkQP(n, m, seed=1234 + info.rank)
    solver = qps.QuadraticProgram(P, q, A, l, u, linsys="cg", dtype=cfg["dtype"], device=device)
    cg = {"n": 0}

    def run():
        x = np.zeros(n); si = {}
        solver.solve(x, numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, info=si)
        cg["n"] += si["cgIterations"]
    run()
    for _ in range(args.warmup):
        run()
    cg["n"] = 0
    solver.set_profiling(args.profile_level)
    qd.barrier(info); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync(); qd.barrier(info)
    elapsed = time.perf_counter() - t0
    ktimes = solver.kernel_times()
    solver.set_profiling(0)
    value, tmax = qd.gather_timings(info, elapsed, args.iters * args.steps)
    out = None
    if info.rank == 0:
        out = base_line(args, cfg, info, value, tmax, "weak")
        pnnz, annz = P.nnz, A.nnz
        spmv = lambda nnz, rows, cols: nnz * 12 + rows * 4 + 8 * (rows + cols)
        cg_bytes = spmv(annz, m, n) + spmv(annz, n, m) + spmv(pnnz, n, n) + 8 * (10 * n + 2 * m)
        cg_per_admm = cg["n"] / (args.iters * args.steps)
        out["config"].update({"parallelism": f"replicas x{info.world_size}", "nnz_P": int(pnnz), "nnz_A": int(annz),
                              "params": "eps=0, rho=1, sigma=1e-6, epsPcg=1e-6, numItrPcg=1000 (LinearSystemSolvers.jl:164), no preconditioner"})
        blk = [k for k in ktimes if k["name"].startswith("spmv_blk(")]
        if blk:   # both column-blocked products of a CG iteration run the same kernel: they are timed in pairs and reported together
            secs, launches = sum(k["seconds"] for k in blk), sum(k["launches"] for k in blk)
            bytes_avg = sum(k["algo_bytes"] * k["launches"] for k in blk) / launches
            traffic, src = pmc_traffic("c3", r"k_spmv_(sell|blk)<")
            out["roofline"] = roofline_of("column-blocked SpMV k_spmv_sell ([P;A] u and A' v of a CG iteration, x block in LDS; QPS_SPMV_SELL=0: k_spmv_blk)", secs, launches, bytes_avg, traffic, src,
                                          {"note": "the ~155 MB working set is Infinity-Cache resident (256 MiB): fractions are against the HBM peak all the same"})
        else:
            out["roofline"] = None
        out["cg_iterations_per_s"] = round(cg["n"] / elapsed, 1)
        out["cg_iterations_per_admm_iteration"] = round(cg_per_admm, 2)
        out["loop_roofline"] = {"algo_bytes_per_cg_iteration": cg_bytes, "achieved_GBs": round(cg_bytes * cg["n"] / elapsed / 1e9, 1),
                                "frac_of_8TBs": round(cg_bytes * cg["n"] / elapsed / 1e9 / HBM_PEAK_GBS, 4)}
        cb = 57344 // 8                                                   # columns per LDS-resident x block (spmv_layout.h), fp64
        out["kernels"] = kernel_list(ktimes, "c3", grid_y={"spmv_blk([P;A]": -(-n // cb), "spmv_blk(A'": -(-m // cb)})
        out["cpu_baseline"], out["parity"] = None, None
        if not args.no_cpu_baseline and info.world_size == 1:
            out["cpu_baseline"], ref = cpu_baseline_sparse(P, q, A, l, u)
            xg = np.zeros(n); gi = {}
            solver.solve(xg, numIterations=ref["K"], ϵAbs=0.0, ϵRel=0.0, ϵPcg=1e-12, numItrPcg=5000, info=gi)
            zg, yg = solver.dual()
            out["parity"] = parity_block(np, ref, xg, zg, yg, gi, 1e-7, "oracle/qps_oracle.c, matrix-free CG plugin (LinearSystemSolvers.jl:145-186), same problem; inner CG "
                                         "driven to epsPcg = 1e-12 on both sides so that the inexact solve does not separate them",
                                         {"cg_iterations": [gi.get("cgIterations"), ref["cgIterations"]]})
    solver.close()
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
# c4: the 256-QP batch cut into contiguous slabs, one per rank, no collective
# ----------------------------------------------------------------------------------------------------------------------------------
def run_batch(args, cfg, info, device, qps, qd, np, sync):
    n, m = cfg["n"], cfg["m"]
    begin, end = qd.shard_range(cfg["batch"], info.rank, info.world_size)      # QP b -> rank b // ceil(256 / world)
    probs = generate_batch(qps, n, m, range(begin, end))
    solver = qps.QuadraticProgramBatch(probs, dtype=cfg["dtype"], device=device)
    first = probs[0]
    del probs
    run = lambda: solver.solve(numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True)
    run()
    for _ in range(args.warmup):
        run()
    set_batch_profiling(solver, args.profile_level)
    qd.barrier(info); sync()
    t0 = time.perf_counter()
    infos = None
    for _ in range(args.steps):
        _, _, infos = run()
    sync(); qd.barrier(info)
    elapsed = time.perf_counter() - t0
    ktimes = batch_kernel_times(solver)
    set_batch_profiling(solver, 0)
    units = (end - begin) * args.iters * args.steps
    value, tmax = qd.gather_timings(info, elapsed, units)
    out = None
    if info.rank == 0:
        out = base_line(args, cfg, info, value, tmax, "strong", "ADMM iterations/sec (QP-iterations, summed over the batch)")
        b_iter = 8 * (m * n + n * n) + 8 * (6 * n + 10 * m)
        per_gpu = value / info.world_size
        out["config"].update({"batch": cfg["batch"], "qps_on_rank0": end - begin, "parallelism": f"256 QPs in {info.world_size} contiguous slab(s), no collective",
                              "trsv_block": infos[0].get("trsvBlock"), "sweep_variant": SWEEP_VARIANTS.get(infos[0].get("sweepVariant")),
                              "params": "eps=0, adptRho off, rho=1, sigma=1e-6, alpha=1.6, numItrConv=25"})
        dom = next((k for k in ktimes if k["name"].startswith("apass(fused")), None)
        if dom:
            traffic, src = pmc_traffic("c4", r"k_apass<.*false, 0>$")
            out["roofline"] = roofline_of(dom["name"], dom["seconds"], dom["launches"], dom["algo_bytes"], traffic, src)
        else:
            out["roofline"] = None
        out["loop_roofline"] = {"algo_bytes_per_qp_iteration": b_iter, "achieved_GBs": round(b_iter * per_gpu / 1e9, 1),
                                "frac_of_8TBs": round(b_iter * per_gpu / 1e9 / HBM_PEAK_GBS, 4), "frac_of_6.29TBs": round(b_iter * per_gpu / 1e9 / HBM_ACHIEVABLE_GBS, 4)}
        out["kernels"] = kernel_list(ktimes, "c4")
        out["cpu_baseline"], out["parity"] = None, None
        if not args.no_cpu_baseline and info.world_size == 1:
            out["cpu_baseline"], ref = cpu_baseline_batch(first, cfg["batch"])
            Xg, _, gis = solver.solve(numIterations=ref["K"], ϵAbs=0.0, ϵRel=0.0, reuseFactor=True)
            Zg, Yg = solver.dual()
            out["parity"] = parity_block(np, ref, Xg[0], Zg[0], Yg[0], gis[0], 1e-9, "oracle/qps_oracle.c on QP 0 of the batch (the batched launches carried all "
                                         f"{end - begin} QPs of the rank), same parameters")
    solver.close()
    if info.rank == 0 and not args.no_time_to_eps:
        # time-to-eps of this rank's slab on the feasible variant (every QP runs to its own stopping iteration, per-QP rho switches)
        # runs to a tolerance take a different number of iterations per QP: this leg hands the QPs out round-robin (dist.shard_assign) instead of in
        # slabs; on the recorded counts of this set (tests/golden/c4_time_to_eps_iterations.json) that is 9.1 % against 9.8 % spread at 8 ranks, LPT 0.8 %
        mine = qd.shard_assign(cfg["batch"], info.rank, info.world_size, assign="round_robin")[:32]
        probs = generate_batch(qps, n, m, mine, feasible=True)
        with qps.QuadraticProgramBatch(probs, dtype=cfg["dtype"], device=device) as sb:
            t1 = time.perf_counter()
            Xe, flags, infos2 = sb.solve(numIterations=50000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
            out["time_to_eps"] = {"eps": 1e-6, "rho0": 0.1, "adptRho": True, "qps": len(probs), "ms_total": round((time.perf_counter() - t1) * 1e3, 2),
                                  "ms_setup": round(infos2[0]["tSetup"] * 1e3, 2), "ms_loop": round(infos2[0]["tLoop"] * 1e3, 2),
                                  "iterations_min_max": [min(i["iterations"] for i in infos2), max(i["iterations"] for i in infos2)],
                                  "flags": sorted(set(int(f) for f in flags)), "refactorisations_max": max(i["numRefactor"] for i in infos2),
                                  "iterations": [i["iterations"] for i in infos2], "assignment": "round_robin", "qp_indices": mine,
                                  "problem": "randomQp density 1.0, bounds centred on A*x0 (feasible variant), the first 32 QPs of rank 0's round-robin share"}
            if not args.no_cpu_baseline and info.world_size == 1:
                out["time_to_eps"].update(cpu_time_to_eps_batch(np, probs, 1e-6, Xe, flags, infos2))
    return out


def usable_cores():
    """min(affinity mask, cgroup CPU quota): a GPU box shows every host core in the mask while the job's share is 16."""
    try:
        c = len(os.sched_getaffinity(0))
    except Exception:
        c = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            c = min(c, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, c)


def generate_batch(qps, n, m, streams, feasible=False):
    """The QPs of a slab, generated on a few host threads (one counter-based stream per QP: the order of generation does not matter)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, usable_cores()))) as ex:
        return list(ex.map(lambda b: qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=b, feasible=feasible), streams))


def set_batch_profiling(solver, level):
    from quadraticprogramsolver_amd import _lib
    _lib.check(_lib.lib().qps_set_profiling(solver._h, int(level)), solver._h)


def batch_kernel_times(solver):
    import ctypes as C
    from quadraticprogramsolver_amd import _lib
    buf = (_lib.QpsKernelTime * 32)()
    cnt = C.c_int32(0)
    _lib.check(_lib.lib().qps_kernel_times(solver._h, buf, 32, C.byref(cnt)), solver._h)
    return [dict(name=buf[i].name.decode(), seconds=buf[i].seconds, launches=buf[i].launches, algo_bytes=buf[i].algo_bytes) for i in range(cnt.value)]


def launcher_selftest(args, info, backend, device, ndev, qd):
    """The N-rank plumbing without the solver (no GPU needed): every rank sleeps 1 ms per "step"; the line is marked as a selftest
    and carries no throughput claim."""
    qd.barrier(info)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    qd.barrier(info)
    elapsed = time.perf_counter() - t0
    value, tmax = qd.gather_timings(info, elapsed, float(args.steps))
    (ranks,) = qd.all_sum(info, 1.0)
    if info.rank == 0:
        print(json.dumps({"metric": "launcher selftest (no solver call)", "value": None, "unit": None, "n_gpus": info.world_size, "steps": args.steps,
                          "warmup": args.warmup, "selftest": True, "ranks_counted": int(ranks), "steps_counted": round(value * tmax),
                          "config": {"dist_backend": backend, "devices_visible": ndev}}), flush=True)
    qd.shutdown(info)


def pmc_traffic(config, kernel_regex):
    """HBM bytes per launch of a kernel from the committed PMC passes of this config (rocprofv3 --pmc FETCH_SIZE and, in a separate
    pass, --pmc WRITE_SIZE on this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; tests/tools/pmc_summary.py).  PMC
    counters cannot be read from inside the timed process, so the figure comes from the newest profiles/rNN_*pmc_traffic_<config>.json."""
    LAST_TRAFFIC_META.clear()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*pmc_traffic_{config}.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        meta = d.get("_meta") or {}
        how = digest_matches(meta, config)
        if not how:
            # measured on other kernel sources (or before summaries carried their build): not this build's traffic
            LAST_TRAFFIC_META.update(stale=f"profiles/{os.path.basename(files[-1])} was measured on csrc {meta.get('csrc_sha16')} (HEAD {meta.get('head')}), "
                                           f"the running build is csrc {csrc_digest()}: traffic not reported")
            return None, None
        LAST_TRAFFIC_META.update(head=meta.get("head"), csrc_sha16=meta.get("csrc_sha16"), match=how)
        best = None
        for name, v in d.items():
            if name != "_meta" and re.search(kernel_regex, name) and (best is None or v.get("launches", v.get("launches_sampled", 0)) > best[1].get("launches", best[1].get("launches_sampled", 0))):
                best = (name, v)
        if best:
            return best[1]["hbm_bytes_per_launch"], f"profiles/{os.path.basename(files[-1])}: {best[0]} (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    except Exception:
        pass
    return None, None


# ----------------------------------------------------------------------------------------------------------------------------------
# CPU baselines: the oracle's C restatement (kind "port": the Julia reference cannot run in this pipeline) on a bounded sample
# ----------------------------------------------------------------------------------------------------------------------------------
CPU_SAMPLES = 3            # BASELINE.md §3 / RunBenchmarks.jl:65-67,98-101: min over several samples, fresh x = 0 (and a fresh factorisation) per sample


def cpu_baseline_dense(P, q, A, l, u, config):
    """Same problem as the GPU run, CPU_SAMPLES samples of: full setup + 100 iterations on all cores (OpenMP: GEMVs over rows, blocked
    triangular solves with shared panels), every sample from x = 0; the reported rate is the best sample (min time, as BenchmarkTools
    reports), the slowest is kept beside it.  Then 20 iterations on one core (the reference loop and its LDL' solves are single-threaded).
    c5 runs the same refactor-per-check schedule (fp64: the oracle has no fp32 path)."""
    from oracle import c_oracle as co
    cores = co.available_cores()   # min(affinity mask, cgroup CPU quota)
    kw = dict(epsAbs=0.0, epsRel=0.0)
    if config == "c5":
        kw.update(adptRho=True, fctrRho=1.0, numItrConv=50, rho=0.1)
    its = 100 if config != "c1" else 20000
    rates, setups, ref = [], [], None
    for _ in range(CPU_SAMPLES):
        x, i_all = co.solve(P, q, A, l, u, numIterations=its, numThreads=cores, **kw)
        rates.append(its / i_all["tLoop"]); setups.append(i_all["tSetup"])
        if ref is None:
            ref = {"K": its, "x": x, "z": i_all["z"], "y": i_all["y"], "iterations": i_all["iterations"], "numRefactor": i_all["numRefactor"]}
    its1 = max(its // 5, 10)
    _, i_one = co.solve(P, q, A, l, u, numIterations=its1, numThreads=cores, loopThreads=1, **kw)
    sample = (f"same problem as the GPU run: {CPU_SAMPLES} samples of full setup + {its} ADMM iterations from x = 0 on all cores (OpenMP; blocked triangular solves), "
              f"value = best sample; single-thread loop rate from {its1} more iterations"
              + ("; refactor at every check (fp64 oracle: no fp32 CPU path)" if config == "c5" else ""))
    return {"value": round(max(rates), 3), "unit": "iterations/s", "cores": cores, "kind": "port", "sample": sample, "samples": CPU_SAMPLES,
            "value_min": round(min(rates), 3), "value_max": round(max(rates), 3), "values": [round(r, 2) for r in rates],
            "setup_s": round(min(setups), 3), "single_thread_iterations_per_s": round(its1 / i_one["tLoop"], 3),
            "refactorisations_in_sample": ref["numRefactor"],
            "note": "CPU restatement of the reference algorithm (oracle/qps_oracle.c); Julia is absent on this box"}, ref


def cpu_time_to_eps(np, problem, eps, x_gpu, flag_gpu, info_gpu, fp32=False, num_itr_conv=25):
    """Time-to-eps of the CPU restatement on the problem the GPU leg just solved, same parameters (RunTests.jl:50-58: rho0 = 0.1, adaptive), and
    the parity of the two runs: equal flag / iteration count / refactorisation count and max|x - x_oracle| <= 1e-5 (RunTests.jl:93); a fp32 GPU
    run against the fp64 oracle: 1e-3 and the stopping iteration within one check interval."""
    from oracle import c_oracle as co
    cores = co.available_cores()
    P, q, A, l, u = problem
    t0 = time.perf_counter()
    xo, io = co.solve(P, q, A, l, u, numIterations=50000, epsAbs=eps, epsRel=eps, rho=0.1, adptRho=True, numItrConv=num_itr_conv, numThreads=cores)
    wall = time.perf_counter() - t0
    dev = float(np.abs(np.asarray(x_gpu, dtype=np.float64) - xo).max())
    tol = 1e-3 if fp32 else 1e-5
    same_flag = flag_gpu == io["convFlag"]
    if fp32:
        same_counts = abs(info_gpu["iterations"] - io["iterations"]) <= num_itr_conv
    else:
        same_counts = info_gpu["iterations"] == io["iterations"] and info_gpu["numRefactor"] == io["numRefactor"]
    cpu = {"iterations": io["iterations"], "flag": io["convFlag"], "refactorisations": io["numRefactor"], "ms_setup": round(io["tSetup"] * 1e3, 2),
           "ms_loop": round(io["tLoop"] * 1e3, 2), "ms_total": round(wall * 1e3, 2), "cores": cores, "kind": "port",
           "note": "oracle/qps_oracle.c (CPU restatement; Julia is absent on this box), one sample"}
    par = {"ok": bool(same_flag and same_counts and dev <= tol), "flag": [flag_gpu, io["convFlag"]], "iterations": [info_gpu["iterations"], io["iterations"]],
           "refactorisations": [info_gpu["numRefactor"], io["numRefactor"]], "max_abs_dev_x": dev, "tolerance": tol,
           "rule": ("flag equal, stopping iteration within one check interval, max|x - x_oracle| <= 1e-3 (fp32 path against the fp64 oracle)" if fp32 else
                    "flag, iterations and refactorisations equal; max|x - x_oracle| <= 1e-5 (RunTests.jl:93)")}
    return {"cpu": cpu, "parity": par, "speedup_loop": round(io["tLoop"] / max(info_gpu["tLoop"], 1e-12), 1),
            "speedup_total": round((io["tSetup"] + io["tLoop"]) / max(info_gpu["tSetup"] + info_gpu["tLoop"], 1e-12), 1)}


def cpu_baseline_sparse(P, q, A, l, u):
    """c3: the oracle's CSC + matrix-free CG plugin (LinearSystemSolvers.jl:145-186) on the same problem, CPU_SAMPLES samples of 10 ADMM iterations."""
    from oracle import c_oracle as co
    cores = co.available_cores()
    rates, cg_rates = [], []
    for _ in range(CPU_SAMPLES):
        x, i_all = co.solve(P, q, A, l, u, numIterations=10, epsAbs=0.0, epsRel=0.0, numThreads=cores, linsys=co.KIND_CG_MATFREE)
        rates.append(10 / i_all["tLoop"]); cg_rates.append(i_all["cgIterations"] / i_all["tLoop"])
    x, i_one = co.solve(P, q, A, l, u, numIterations=4, epsAbs=0.0, epsRel=0.0, numThreads=cores, loopThreads=1, linsys=co.KIND_CG_MATFREE)
    # the parity reference: K = 5 with the inner CG driven to 1e-12 (at the benchmark's epsPcg = 1e-6 two correct CG codes differ by ~1e-6)
    xr, ir = co.solve(P, q, A, l, u, numIterations=5, epsAbs=0.0, epsRel=0.0, numThreads=cores, linsys=co.KIND_CG_MATFREE, epsPcg=1e-12, numItrPcg=5000)
    ref = {"K": 5, "x": xr, "z": ir["z"], "y": ir["y"], "iterations": ir["iterations"], "numRefactor": ir["numRefactor"], "cgIterations": ir["cgIterations"]}
    return {"value": round(max(rates), 3), "unit": "iterations/s", "cores": cores, "kind": "port", "samples": CPU_SAMPLES,
            "value_min": round(min(rates), 3), "value_max": round(max(rates), 3), "values": [round(r, 3) for r in rates],
            "sample": f"same problem as the GPU run: {CPU_SAMPLES} samples of 10 ADMM iterations from x = 0 (matrix-free CG, epsPcg 1e-6) on all cores (OpenMP over the CSC columns), "
                      "value = best sample; single-thread rate from 4 more",
            "cg_iterations_per_s": round(max(cg_rates), 1), "single_thread_iterations_per_s": round(4 / i_one["tLoop"], 3),
            "note": "CPU restatement of the reference algorithm (oracle/qps_oracle.c); Julia is absent on this box"}, ref


def cpu_baseline_batch(first, batch):
    """c4: a CPU runs the QPs one after the other, so its QP-iterations/s is the rate of one QP (n = 1024): 200 iterations of QP 0, 5 samples."""
    from oracle import c_oracle as co
    cores = co.available_cores()
    P, q, A, l, u = first
    rates, setups, ref = [], [], None
    for _ in range(5):
        x, i_all = co.solve(P, q, A, l, u, numIterations=200, epsAbs=0.0, epsRel=0.0, numThreads=cores)
        rates.append(200 / i_all["tLoop"]); setups.append(i_all["tSetup"])
        if ref is None:
            ref = {"K": 200, "x": x, "z": i_all["z"], "y": i_all["y"], "iterations": i_all["iterations"], "numRefactor": i_all["numRefactor"]}
    _, i_one = co.solve(P, q, A, l, u, numIterations=200, epsAbs=0.0, epsRel=0.0, numThreads=cores, loopThreads=1)
    return {"value": round(max(rates), 3), "unit": "iterations/s", "cores": cores, "kind": "port", "samples": 5,
            "value_min": round(min(rates), 3), "value_max": round(max(rates), 3), "values": [round(r, 1) for r in rates],
            "sample": f"QP 0 of the batch: 5 samples of full setup + 200 ADMM iterations from x = 0 on all cores, value = best sample; the {batch} QPs would run back to back at this QP-iteration rate",
            "setup_s": round(min(setups), 3), "single_thread_iterations_per_s": round(200 / i_one["tLoop"], 3),
            "note": "CPU restatement of the reference algorithm (oracle/qps_oracle.c); Julia is absent on this box"}, ref


def cpu_time_to_eps_batch(np, probs, eps, X_gpu, flags_gpu, infos_gpu):
    """c4 slab: the CPU restatement solves the same QPs one after the other (RunBenchmarks.jl:88-104 is that loop) to the same tolerance with the
    same parameters; per-QP parity as in cpu_time_to_eps."""
    from oracle import c_oracle as co
    cores = co.available_cores()
    t0 = time.perf_counter()
    t_setup = t_loop = 0.0
    its, bad, dev_max = [], [], 0.0
    for b, (P, q, A, l, u) in enumerate(probs):
        xo, io = co.solve(P, q, A, l, u, numIterations=50000, epsAbs=eps, epsRel=eps, rho=0.1, adptRho=True, numThreads=cores)
        t_setup += io["tSetup"]; t_loop += io["tLoop"]; its.append(io["iterations"])
        dev = float(np.abs(X_gpu[b] - xo).max()); dev_max = max(dev_max, dev)
        if not (int(flags_gpu[b]) == io["convFlag"] and infos_gpu[b]["iterations"] == io["iterations"] and infos_gpu[b]["numRefactor"] == io["numRefactor"] and dev <= 1e-5):
            bad.append(b)
    wall = time.perf_counter() - t0
    return {"cpu": {"qps": len(probs), "ms_setup": round(t_setup * 1e3, 2), "ms_loop": round(t_loop * 1e3, 2), "ms_total": round(wall * 1e3, 2), "iterations": its,
                    "cores": cores, "kind": "port", "note": "oracle/qps_oracle.c, the QPs one after the other (each on all cores), one sample"},
            "parity": {"ok": not bad, "qps_checked": len(probs), "mismatching_qps": bad, "max_abs_dev_x": dev_max, "tolerance": 1e-5,
                       "rule": "per QP: flag, iterations and refactorisations equal; max|x - x_oracle| <= 1e-5 (RunTests.jl:93)"}}


if __name__ == "__main__":
    main()
