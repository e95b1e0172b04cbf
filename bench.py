#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the device-resident loop on BASELINE.json's headline config
(dense n = 4096, m = 8192, fp64, one QP per GPU; ranks run independent replicas, no data-path collective).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one qps_solve() of --iters (default 500) ADMM iterations (ϵAbs = ϵRel = 0, adptΡ off, reference defaults otherwise) on a
problem already resident in HBM with its factorisation cached (setup is reported separately).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
HBM_ACHIEVABLE_GBS = 6290.0

CONFIGS = {
    "c2": dict(n=4096, m=8192, dtype="f64", label="dense n=4096 m=8192 fp64 single QP (BASELINE configs[1])"),
    "c5": dict(n=4096, m=8192, dtype="f32", label="dense n=4096 m=8192 fp32, rho-update + refactor every 50 its (configs[4])"),
    "c1": dict(n=64, m=128, dtype="f64", label="dense n=64 m=128 fp64 (configs[0] shape, plumbing)"),
    "c3": dict(n=50000, m=100000, dtype="f64", label="sparse P,A n=50k m=100k ~0.1% nnz fp64, CSR SpMV + matrix-free CG (configs[2])"),
    "c4": dict(n=1024, m=2048, dtype="f64", batch=256, label="batch of 256 dense n=1024 m=2048 QPs sharded across the ranks (configs[3])"),
}


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
    ap.add_argument("--iters", type=int, default=500, help="ADMM iterations per step")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--trsv-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-eps", action="store_true")
    ap.add_argument("--profile-level", type=int, default=1,
                    help="1: HIP-event bracket the dominant kernel on every 50th iteration; 2: every launch of every kernel")
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
    n, m = cfg["n"], cfg["m"]
    s = 8 if cfg["dtype"] == "f64" else 4

    if torch.cuda.is_available():
        torch.cuda.set_device(device)      # torch.cuda.synchronize() below must act on this rank's GPU, not on GPU 0
    if args.config in ("c3", "c4"):
        return side_config(args, cfg, info, backend, device, qps, qd, np, torch)
    # synthetic input: randomQp at density 1.0, seed 1234, one independent stream per rank (weak scaling)
    P, q, A, l, u = qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=info.rank)
    prob = qps.QuadraticProgram(P, q, A, l, u, dtype=cfg["dtype"], device=device)
    solve_kw = dict(numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, trsvBlock=args.trsv_block, reuseFactor=True)
    if args.config == "c5":
        solve_kw.update(adptΡ=True, fctrΡ=1.0, numItrConv=50, ρ=0.1)   # ρ proposal applied (=> refactor) at every check

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    setup_info = {}
    x = np.zeros(n)
    prob.solve(x, **dict(solve_kw, numIterations=min(args.iters, 25), reuseFactor=False), info=setup_info)   # builds + caches the factor
    for _ in range(args.warmup):
        x = np.zeros(n)
        prob.solve(x, **solve_kw)
    prob.set_profiling(args.profile_level)
    iters_done = 0
    loop_device_s = 0.0
    qd.barrier(info); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = np.zeros(n); si = {}
        prob.solve(x, **solve_kw, info=si)
        iters_done += si["iterations"]; loop_device_s += si["tLoop"]
    sync(); qd.barrier(info)
    elapsed = time.perf_counter() - t0
    ktimes = prob.kernel_times()   # HIP-event times accumulated over the timed steps (reset by set_profiling)
    prob.set_profiling(0)
    value, tmax = qd.gather_timings(info, elapsed, iters_done)

    out = None
    if info.rank == 0:
        b_iter = s * (m * n + n * n) + s * (6 * n + 10 * m)       # SURVEY §8d algorithmic bytes per ADMM iteration
        # dominant kernel of the loop: the fused A-pass (sampled at level 1); fall back to the largest accumulated time
        dom = next((k for k in ktimes if k["name"].startswith("apass(fused")), None) or (max(ktimes, key=lambda k: k["seconds"]) if ktimes else None)
        roofline = None
        if dom:
            dur = dom["seconds"] / dom["launches"]
            ach = dom["algo_bytes"] / dur / 1e9
            traffic, traffic_src = pmc_traffic(dom["name"], cfg["dtype"])
            roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_us": round(dur * 1e6, 2), "algo_bytes_per_launch": dom["algo_bytes"],
                        "launches_timed": dom["launches"]}
        per_gpu = value / info.world_size
        out = {
            "metric": "ADMM iterations/sec", "value": round(value, 2), "unit": "iterations/s", "n_gpus": info.world_size,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tmax / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["label"], "n": n, "m": m, "admm_iterations_per_step": args.iters,
                       "qps_per_gpu": 1, "parallelism": f"replicas x{info.world_size} (independent QPs, no collective)",
                       "params": "eps=0, adptRho off, rho=1, sigma=1e-6, alpha=1.6, numItrConv=25 (SolveQuadraticProgram.jl:15-17)"
                                 if args.config != "c5" else "fp32, adptRho on, fctrRho=1, numItrConv=50 (refactor at every check)",
                       "dist_backend": backend},
            "roofline": roofline,
            "loop_roofline": {"algo_bytes_per_iteration": b_iter, "achieved_GBs": round(b_iter * per_gpu / 1e9, 1),
                              "frac_of_8TBs": round(b_iter * per_gpu / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_of_6.29TBs": round(b_iter * per_gpu / 1e9 / HBM_ACHIEVABLE_GBS, 4)},
            "setup_ms": round(setup_info.get("tSetup", 0.0) * 1e3, 2),
            "kernels": [{"name": k["name"], "avg_us": round(k["seconds"] / k["launches"] * 1e6, 2), "launches": k["launches"],
                         "algo_GBs": round(k["algo_bytes"] / (k["seconds"] / k["launches"]) / 1e9, 1)} for k in ktimes],
        }
        if not args.no_time_to_eps and args.config == "c2":
            # time-to-eps on the feasible variant (the plain m = 2n draw is primal infeasible: see generator docstring)
            Pf, qf, Af, lf, uf = qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=info.rank, feasible=True)
            with qps.QuadraticProgram(Pf, qf, Af, lf, uf, dtype=cfg["dtype"], device=device) as pf:
                xf = np.zeros(n); ti = {}
                t1 = time.perf_counter()
                flag = pf.solve(xf, numIterations=50000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True, trsvBlock=args.trsv_block, info=ti)
                out["time_to_eps"] = {"eps": 1e-6, "rho0": 0.1, "adptRho": True, "flag": int(flag), "iterations": ti["iterations"],
                                      "refactorisations": ti["numRefactor"], "ms_total": round((time.perf_counter() - t1) * 1e3, 2),
                                      "ms_setup": round(ti["tSetup"] * 1e3, 2), "ms_loop": round(ti["tLoop"] * 1e3, 2),
                                      "problem": "randomQp density 1.0, bounds centred on A*x0 (feasible variant)"}
            del Pf, Af
        if not args.no_cpu_baseline and info.world_size == 1:
            out["cpu_baseline"] = cpu_baseline(P, q, A, l, u, args.config)
    prob.close()
    if info.rank == 0:
        print(json.dumps(out), flush=True)
    qd.shutdown(info)


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


def side_config(args, cfg, info, backend, device, qps, qd, np, torch):
    """Configs that are not the headline line: c3 (CSR/CG, replicas) and c4 (the 256-QP batch cut into contiguous slabs,
    one per rank, no collective).  Same timing contract: W warm-up steps, K timed steps between barrier + synchronize."""
    n, m = cfg["n"], cfg["m"]
    sync = (lambda: torch.cuda.synchronize()) if torch.cuda.is_available() else (lambda: None)
    if args.iters == 500:
        args.iters = 100          # side configs: shorter steps
    if args.config == "c4":
        begin, end = qd.shard_range(cfg["batch"], info.rank, info.world_size)      # QP b -> rank b // ceil(256 / world)
        probs = [qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=b) for b in range(begin, end)]
        solver = qps.QuadraticProgramBatch(probs, dtype=cfg["dtype"], device=device)
        run = lambda: solver.solve(numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True)
        units_per_step = (end - begin) * args.iters
        algo_bytes_per_unit = 8 * (m * n + n * n) + 8 * (6 * n + 10 * m)
    else:
        P, q, A, l, u = qps.GenerateSparseBenchmarkQP(n, m, seed=1234 + info.rank)
        solver = qps.QuadraticProgram(P, q, A, l, u, linsys="cg", dtype=cfg["dtype"], device=device)
        cg = {"n": 0}
        def run():
            x = np.zeros(n); si = {}
            solver.solve(x, numIterations=args.iters, ϵAbs=0.0, ϵRel=0.0, info=si)
            cg["n"] += si["cgIterations"]
        units_per_step = args.iters
        algo_bytes_per_unit = None
    run()
    for _ in range(args.warmup):
        run()
    if args.config == "c3":
        cg["n"] = 0
    qd.barrier(info); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync(); qd.barrier(info)
    elapsed = time.perf_counter() - t0
    value, tmax = qd.gather_timings(info, elapsed, units_per_step * args.steps)
    if info.rank == 0:
        out = {"metric": "ADMM iterations/sec" + (" (QP-iterations, summed over the batch)" if args.config == "c4" else ""),
               "value": round(value, 2), "unit": "iterations/s", "n_gpus": info.world_size, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(tmax / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "strong" if args.config == "c4" else "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
               "config": {"workload": cfg["label"], "n": n, "m": m, "admm_iterations_per_step": args.iters, "dist_backend": backend,
                          "parallelism": (f"256 QPs in {info.world_size} contiguous slab(s), no collective" if args.config == "c4"
                                          else f"replicas x{info.world_size}")},
               "roofline": None, "cpu_baseline": None}
        if algo_bytes_per_unit:
            per_gpu = value / info.world_size
            out["loop_roofline"] = {"algo_bytes_per_iteration": algo_bytes_per_unit, "achieved_GBs": round(algo_bytes_per_unit * per_gpu / 1e9, 1),
                                    "frac_of_8TBs": round(algo_bytes_per_unit * per_gpu / 1e9 / HBM_PEAK_GBS, 4)}
        if args.config == "c3":
            out["cg_iterations_per_s"] = round(cg["n"] / elapsed, 1)
            out["cg_iterations_per_admm_iteration"] = round(cg["n"] / (units_per_step * args.steps), 2)
        print(json.dumps(out), flush=True)
    solver.close()
    qd.shutdown(info)


def pmc_traffic(kernel_label, dtype):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and, in a
    separate pass, --pmc WRITE_SIZE on this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  PMC counters
    cannot be read from inside the timed process, so the figure comes from profiles/ (or null when absent)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not (os.path.exists(path) and kernel_label.startswith("apass(fused") and dtype == "f64"):
        return None, None
    try:
        d = json.load(open(path))
        for name, v in d.items():
            if name.startswith("k_apass<double") and (name.endswith("false>") or name.endswith("false, 0>")):   # the plain ADMM variant
                return v["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    except Exception:
        pass
    return None, None


def cpu_baseline(P, q, A, l, u, config):
    """The oracle's C restatement (kind "port": the Julia reference cannot run in this pipeline) timed on this box's
    host cores on a bounded sample of the same workload: the full setup + 50 iterations on all cores, then 10
    iterations of the loop on one core (the reference loop and its LDL solves are single-threaded)."""
    from oracle import c_oracle as co
    cores = co.available_cores()   # min(affinity mask, cgroup CPU quota)
    x, i_all = co.solve(P, q, A, l, u, numIterations=50, epsAbs=0.0, epsRel=0.0, numThreads=cores)
    x, i_one = co.solve(P, q, A, l, u, numIterations=10, epsAbs=0.0, epsRel=0.0, numThreads=cores, loopThreads=1)
    return {"value": round(50 / i_all["tLoop"], 3), "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": "same problem as the GPU run: full setup + 50 ADMM iterations on all cores (OpenMP over rows); "
                      "single-thread loop rate from 10 more iterations",
            "setup_s": round(i_all["tSetup"], 3), "single_thread_iterations_per_s": round(10 / i_one["tLoop"], 3),
            "note": "CPU restatement of the reference algorithm (oracle/qps_oracle.c); Julia is absent on this box"}


if __name__ == "__main__":
    main()
