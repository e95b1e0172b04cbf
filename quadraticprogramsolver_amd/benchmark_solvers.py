"""Counterpart of the reference's plugin comparison driver BenchmarkSolvers.jl / BenchmarkSolver.jl (SURVEY §8b "what calls it").

Same protocol: one problem class (``randomQp``, BenchmarkSolvers.jl:19), ``numDims`` sizes between ``numElementsMin`` and
``numElementsMax`` (linear or log spaced, :20-25, :64-65), every plugin pair solved with the keyword set of :48-57
(``numIterations = 5000, ϵAbs = ϵRel = 1e-6, ρ = 1, σ = 1e-6, α = 1.6, adptΡ = true, fctrΡ = 5, numItrConv = 25``) from a fresh
``vX = zeros(n)`` per sample, and the min / max / median run time per (size, plugin) collected in ``tR`` (:67, :78-80).
The reference's allocation columns (:81-82) have no device meaning; the ADMM iteration count and the convergence flag take
their place.  The plots (:88-96) become a CSV table.

    python -m quadraticprogramsolver_amd.benchmark_solvers [--min 200 --max 1200 --dims 5 --samples 5 --csv solvers.csv]
"""
from __future__ import annotations

import argparse
import csv
import time

import numpy as np

from . import (ConvergenceFlag, GenerateRandomQP, HipCg, HipCgInit, HipChol, HipCholF32, HipCholF32Init, HipCholInit, HipLdl, HipLdlInit, ProblemClass,
               QuadraticProgram, make_rng)

SOLVERS = (("HipLdl (sparse KKT L D L', fp64)", HipLdlInit, HipLdl), ("HipChol (dense Cholesky, fp64)", HipCholInit, HipChol),
           ("HipCholF32 (dense Cholesky, fp32)", HipCholF32Init, HipCholF32), ("HipCg (CSR matrix-free CG, fp64)", HipCgInit, HipCg))
REF_KW = dict(numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=1, σ=1e-6, α=1.6, δ=1e-6, adptΡ=True, fctrΡ=5, numItrConv=25)   # BenchmarkSolvers.jl:48-57


def GenerateElementsVector(minVal: int, maxVal: int, numDims: int, logSpace: bool = False):
    """BenchmarkSolvers.jl:64-65: numDims sizes from minVal to maxVal (rounded), linear or logarithmic spacing."""
    if numDims <= 1:
        return [int(minVal)]
    v = np.logspace(np.log10(max(minVal, 1)), np.log10(max(maxVal, 1)), numDims) if logSpace else np.linspace(minVal, maxVal, numDims)
    return [int(round(x)) for x in v]


def BenchmarkSolver(pair, problemClass, vNumElements, vNumConstraints, samples: int = 5, seed: int = 1234, device: int = 0):
    """BenchmarkSolver.jl: one plugin pair over the size vector; per size the list of sample times [s], iterations, flag."""
    _, init, sol = pair
    out = []
    for ii, (n, m) in enumerate(zip(vNumElements, vNumConstraints)):
        mP, vQ, mA, vL, vU = GenerateRandomQP(problemClass, n, numConstraints=m, rng=make_rng(seed, ii))
        times, its, flag = [], 0, ConvergenceFlag.convNumItr
        with QuadraticProgram(mP, vQ, mA, vL, vU, linsys=init._qps_linsys, dtype=init._qps_dtype, device=device) as prob:
            for _ in range(samples):
                vX = np.zeros(mP.shape[0]); info = {}
                t0 = time.perf_counter()
                flag = prob.solve(vX, info=info, **REF_KW)
                times.append(time.perf_counter() - t0)
                its = info["iterations"]
        out.append({"times": times, "iterations": its, "flag": flag, "size": (mP.shape[0], mA.shape[0])})
    return out


def run(numElementsMin=200, numElementsMax=1200, numConstraintsMin=0, numConstraintsMax=0, numDims=5, logSpace=False,
        problemClass=ProblemClass.randomQp, samples=5, solvers=SOLVERS, csv_path=None):
    vN = GenerateElementsVector(numElementsMin, numElementsMax, numDims, logSpace)
    vM = GenerateElementsVector(numConstraintsMin, numConstraintsMax, numDims, logSpace)
    tR = np.zeros((numDims, len(solvers), 5))            # min, max, median time [s], iterations, converged   (BenchmarkSolvers.jl:67)
    rows = []
    for jj, pair in enumerate(solvers):
        res = BenchmarkSolver(pair, problemClass, vN, vM, samples)
        for ii, r in enumerate(res):
            tR[ii, jj] = (min(r["times"]), max(r["times"]), float(np.median(r["times"])), r["iterations"], r["flag"] != ConvergenceFlag.convNumItr)
            rows.append([pair[0], r["size"][0], r["size"][1], *tR[ii, jj, :3], int(tR[ii, jj, 3]), bool(tR[ii, jj, 4])])
    if csv_path:
        with open(csv_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Solver", "numElements", "numConstraints", "Min Run Time [s]", "Max Run Time [s]", "Median Run Time [s]", "# Iterations", "Converged"])
            w.writerows(rows)
    return vN, tR, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=200); ap.add_argument("--max", type=int, default=1200)      # BenchmarkSolvers.jl:20-21
    ap.add_argument("--dims", type=int, default=5); ap.add_argument("--log", action="store_true")          # :24-25
    ap.add_argument("--samples", type=int, default=5); ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    vN, tR, rows = run(a.min, a.max, 0, 0, a.dims, a.log, samples=a.samples, csv_path=a.csv)
    for r in rows:
        print(f"{r[0]:36s} n={r[1]:5d} m={r[2]:5d}  min {r[3]*1e3:8.2f} ms  max {r[4]*1e3:8.2f} ms  median {r[5]*1e3:8.2f} ms  its {r[6]:5d}  converged {r[7]}")


if __name__ == "__main__":
    main()
