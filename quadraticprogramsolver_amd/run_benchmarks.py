"""Counterpart of the reference benchmark driver RunBenchmarks.jl (SURVEY §3.4, §8b "what calls it").

Same protocol: for every ProblemClass x 2 sizes x numSimulations problems (RunBenchmarks.jl:88-91), solve with a fresh
``vX = zeros(n)`` per sample (:98 setup), keep the MIN time over the samples (:101), record whether the run converged
(``convFlag != convNumItr``, :104), and append ONE row to a CSV whose header must match the existing file (:125-137).
The reference's BenchmarkTools allocation columns have no GPU meaning; the iteration count takes their place.

    python -m quadraticprogramsolver_amd.run_benchmarks --csv QPSBenchmark.csv [--sizes 10 100] [--sims 5] [--samples 15]
"""
from __future__ import annotations

import argparse
import csv
import datetime
import os
import time

import numpy as np

from . import ConvergenceFlag, GenerateRandomQP, ProblemClass, QuadraticProgram, make_rng
from . import _lib


def run(csv_path: str, sizes=(2, 20), num_simulations: int = 5, samples: int = 15, num_iterations: int = 50000,
        seed: int = 1234, linsys: str = "ldl", device: int = 0):
    """``linsys``: "ldl" = the sparse L D L' plugin, the counterpart of the FacLdlInit / FacLdl! pair RunBenchmarks.jl:54-55 selects
    (default); "cholesky" = the dense reduced form; "cg" = matrix-free CG."""
    header = ["Solver Label", "Solver Version", "System Info", "Test Date Time"]       # RunBenchmarks.jl:79-82
    row = [f"QPS HIP {linsys}", _lib.lib().qps_version().decode(), "AMD Instinct MI355X (gfx950)",
           datetime.datetime.utcnow().strftime("%Y_%m_%d_%S_%M_%H")]                    # :61 date format kept
    test_idx = 1
    for pc in ProblemClass:                                                            # :88
        for n in sizes:                                                                # :89
            for sim in range(num_simulations):                                         # :90
                mP, vQ, mA, vL, vU = GenerateRandomQP(pc, n, rng=make_rng(seed, test_idx))   # :91
                best, flag, its = float("inf"), ConvergenceFlag.convNumItr, 0
                with QuadraticProgram(mP, vQ, mA, vL, vU, linsys=linsys, device=device) as prob:
                    for _ in range(samples):                                           # :99 samples, evals = 1
                        vX = np.zeros(mP.shape[0])                                     # :98 setup = (vX = copy(vXX))
                        info = {}
                        t0 = time.perf_counter()
                        flag = prob.solve(vX, numIterations=num_iterations, info=info)  # :57 defaults otherwise
                        best = min(best, time.perf_counter() - t0)                     # :101 min time
                        its = info["iterations"]
                tag = f"Test {test_idx:04d}"
                header += [f"{tag} Run Time", f"{tag} # Iterations", f"{tag} Problem Size", f"{tag} Convergence"]   # :106-109
                row += [int(best * 1e9), its, f"{mP.shape[0]}x{mA.shape[0]}", flag != ConvergenceFlag.convNumItr]  # :101-104
                test_idx += 1
    if os.path.isfile(csv_path):                                                       # :125-133
        with open(csv_path, newline="") as f:
            existing = next(csv.reader(f))
        if existing != header:
            raise RuntimeError("The Header of the tests doesn't match the header of the CSV file")
        with open(csv_path, "a", newline="") as f:
            csv.writer(f).writerow(row)
    else:                                                                              # :134-137
        with open(csv_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(header)
            w.writerow(row)
    return header, row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default="QPSBenchmark.csv")                                # RunBenchmarks.jl:49
    ap.add_argument("--sizes", type=int, nargs="+", default=[2, 20])                    # :29-37 ([10 100] .÷ 5)
    ap.add_argument("--sims", type=int, default=5)                                      # :28
    ap.add_argument("--samples", type=int, default=15)                                  # :65
    ap.add_argument("--linsys", default="ldl", choices=["ldl", "cholesky", "cg"])        # :54-55 hLinSolInit = FacLdlInit
    a = ap.parse_args()
    header, row = run(a.csv, tuple(a.sizes), a.sims, a.samples, linsys=a.linsys)
    print(f"appended {len(row)} columns to {a.csv}")


if __name__ == "__main__":
    main()
