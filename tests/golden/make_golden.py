"""Generates the committed golden fixtures under tests/golden/*.npz.

The Julia reference cannot run in this pipeline and holds no golden vectors of its own, so these
fixtures are produced by the numpy/LAPACK mirror of the reference algorithm (oracle/qps_oracle_np.py), never by the
reference.  They pin (a) the C restatement oracle/qps_oracle.c and (b) the HIP path against an independently written
implementation.  PARITY UNPINNED with respect to the Julia code itself.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import qps_oracle_np as onp  # noqa: E402
from quadraticprogramsolver_amd.generator import GenerateDenseBenchmarkQP, GenerateRandomQP, ProblemClass, make_rng  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(P, q, A, l, u, K=None, **kw):
    x = np.zeros(P.shape[0])
    info = {}
    if K is not None:
        kw = dict(kw, numIterations=K, ϵAbs=0.0, ϵRel=0.0)
    flag = onp.SolveQuadraticProgramRefLoop(x, P, q, A, l, u, onp.KktLdlInit, onp.KktLdl, info=info, **kw)
    return x, int(flag), info


def save_problem(name, P, q, A, l, u, solver_kw):
    P = np.asarray(P.toarray() if hasattr(P, "toarray") else P)
    A = np.asarray(A.toarray() if hasattr(A, "toarray") else A)
    d = dict(P=P, q=q, A=A, l=l, u=u)
    for K in (25, 50, 100):
        x, _, info = run(P, q, A, l, u, K=K, **{k: v for k, v in solver_kw.items() if k in ("ρ",)})   # fixed-K iterates: adptΡ off
        d[f"x_K{K}"], d[f"z_K{K}"], d[f"y_K{K}"] = x, info["z"], info["y"]
    x, flag, info = run(P, q, A, l, u, **solver_kw)
    d.update(x_final=x, flag=np.int64(flag), iterations=np.int64(info["iterations"]), rho_final=np.float64(info["rho_final"]),
             n_refactor=np.int64(info["n_refactor"]), z_final=info["z"], y_final=info["y"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, P.shape, A.shape, "flag", flag, "iterations", info["iterations"], "rho", info["rho_final"])


def main():
    tests_kw = dict(numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)   # RunTests.jl:50-54
    # C1-sized problems (BASELINE.json config 1: n = 64, m = 128) for three classes
    # BASELINE config 1 shape (n = 64, m = 128).  The plain randomQp draw is primal infeasible at m = 2n (see
    # GenerateDenseBenchmarkQP), so its long-run end state is not a meaningful fixture: iterates only (final = 200 its).
    P, q, A, l, u = GenerateDenseBenchmarkQP(64, 128, stream=1)
    save_problem("c1_randomQp_n64_m128", P, q, A, l, u, dict(tests_kw, numIterations=200))
    P, q, A, l, u = GenerateDenseBenchmarkQP(64, 128, stream=1, feasible=True)
    save_problem("c1_randomQp_feasible_n64_m128", P, q, A, l, u, tests_kw)
    P, q, A, l, u = GenerateRandomQP(ProblemClass.randomQp, 64, rng=make_rng(1234, 5), densityFctr=1.0, dense=True)
    save_problem("c1_randomQp_n64_m32", P, q, A, l, u, tests_kw)
    P, q, A, l, u = GenerateRandomQP(ProblemClass.equalityConstrainedQp, 64, numConstraints=32, rng=make_rng(1234, 2), densityFctr=1.0, dense=True)
    save_problem("c1_equalityConstrainedQp_n64_m32", P, q, A, l, u, tests_kw)
    P, q, A, l, u = GenerateRandomQP(ProblemClass.isotonicRegression, 64, rng=make_rng(1234, 3))
    save_problem("c1_isotonicRegression_n64", P, q, A, l, u, tests_kw)
    P, q, A, l, u = GenerateRandomQP(ProblemClass.supportVectorMachine, 4, numConstraints=40, rng=make_rng(1234, 4))
    save_problem("svm_n4_m40_infbounds", P, q, A, l, u, tests_kw)
    # analytic known-answer problems (no solver needed for the expected x*)
    rng = make_rng(1234, 9)
    for n in (4, 16, 64):
        M = rng.standard_normal((n, n))
        P = M.T @ M + 1e-2 * np.eye(n)
        q = rng.standard_normal(n)
        A = rng.standard_normal((2 * n, n))
        inf = np.inf * np.ones(2 * n)
        np.savez_compressed(os.path.join(OUT, f"kat_unconstrained_n{n}.npz"), P=P, q=q, A=A, l=-inf, u=inf,
                            x_star=np.linalg.solve(P, -q))
        Ae = rng.standard_normal((n // 2, n))
        b = rng.standard_normal(n // 2)
        K = np.block([[P, Ae.T], [Ae, np.zeros((n // 2, n // 2))]])
        sol = np.linalg.solve(K, np.concatenate([-q, b]))
        np.savez_compressed(os.path.join(OUT, f"kat_equality_n{n}.npz"), P=P, q=q, A=Ae, l=b, u=b, x_star=sol[:n], y_star=sol[n:])
        p = rng.random(n) + 0.5
        lo, hi = -rng.random(n), rng.random(n)
        np.savez_compressed(os.path.join(OUT, f"kat_box_diag_n{n}.npz"), P=np.diag(p), q=q, A=np.eye(n), l=lo, u=hi,
                            x_star=np.clip(-q / p, lo, hi))


if __name__ == "__main__":
    main()
