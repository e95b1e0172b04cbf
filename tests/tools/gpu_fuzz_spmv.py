"""Randomised differential run of the column-blocked SpMV kernels (sliced form k_spmv_sell and task form k_spmv_blk) through the C ABI (not a test; prints
every mismatch).  Each case: a random sparse A (m x n) and P = M'M + 0.01 I with a structure drawn to stress the layout (tests/spmv_cases.py), then
  PRIMARY, operator level (qps_operator_apply): [P; A] u, A' v, A u, P u and the reduced operator against scipy at 1e-13 (fp64) / 2e-5 (fp32) of sum_j |a_ij u_j| --
           a missing, duplicated or misplaced entry cannot hide here;
  SECONDARY, through CG (qps_linsys_solve = LinOpCg!, LinearSystemSolvers.jl:164-181): z~ == A x~ to rounding and the fp64-recomputed residual within a band of the
           IterativeSolvers stopping rule (1.2x fp64; 10x fp32, where the recurrence residual CG stops on drifts from the true one over ~1000 iterations).
usage: python tests/tools/gpu_fuzz_spmv.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
import quadraticprogramsolver_amd as q
from spmv_cases import draw_case, spd_companion

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
os.environ["QPS_SPMV_BLOCKED"] = "1"
bad, drift, t0 = 0, 0, time.time()
scale_of = lambda M, v: float((abs(M) @ np.abs(v)).max()) + 1e-300
for c in range(cases):
    dtype = "f64" if rng.random() < 0.6 else "f32"
    sell = "1" if rng.random() < 0.7 else "0"
    os.environ["QPS_SPMV_SELL"] = sell
    A, tag = draw_case(rng)
    m, n = A.shape
    tag = f"case {c}: {dtype} sell={sell} " + tag
    P = spd_companion(rng, n, c)
    qv = rng.standard_normal(n)
    Ac, At, Pc = sp.csr_matrix(A), sp.csr_matrix(A.T), sp.csr_matrix(P)
    rho, sigma = float(rng.choice([0.1, 1.0, 7.0])), 1e-6
    x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    try:
        with q.QuadraticProgram(P, qv, sp.csc_matrix(A), np.zeros(m), np.zeros(m), linsys="cg", dtype=dtype) as prob:
            otol = 1e-13 if dtype == "f64" else 2e-5
            Au = Ac @ x
            errs = {"PA": np.abs(prob.apply("PA", x) - np.concatenate([Pc @ x, Au])).max() / max(scale_of(Pc, x), scale_of(Ac, x)),
                    "At": np.abs(prob.apply("At", z) - At @ z).max() / scale_of(At, z),
                    "A": np.abs(prob.apply("A", x) - Au).max() / scale_of(Ac, x), "P": np.abs(prob.apply("P", x) - Pc @ x).max() / scale_of(Pc, x)}
            red = prob.apply("reduced", x, ρ=rho, σ=sigma)
            errs["reduced"] = np.abs(red - (Pc @ x + rho * (At @ Au) + sigma * x)).max() / (scale_of(Pc, x) + rho * scale_of(At, np.abs(Ac) @ np.abs(x)) + sigma) / 4
            op_ok = all(np.isfinite(v) and v <= otol for v in errs.values())
            prob.linsys_init(rho, sigma)
            xx, zz = np.zeros(n), np.zeros(m)
            eps = 1e-10 if dtype == "f64" else 1e-4
            prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz, ϵPcg=eps, numItrPcg=20000)
        rhs = sigma * x - qv + At @ (rho * z - y)
        res = np.linalg.norm(Pc @ xx + sigma * xx + rho * (At @ (Ac @ xx)) - rhs)
        reltol = 1.4901161193847656e-08 if dtype == "f64" else 3.4526698300124393e-04
        tol = max(reltol * np.linalg.norm(rhs), eps)                      # x0 = 0 after Init: r0 = rhs
        zerr = np.abs(zz - Ac @ xx).max() / max(1.0, np.abs(zz).max())
        cg_ok = np.all(np.isfinite(xx)) and res <= (1.2 if dtype == "f64" else 10.0) * tol and zerr <= (1e-11 if dtype == "f64" else 3e-4)
        worst = max(errs, key=errs.get)
        if not op_ok:
            bad += 1
            print("MISMATCH (operator level)", tag, " ".join(f"{k}={v:.2e}" for k, v in errs.items()), f"tolerance {otol:g}", flush=True)
        elif not cg_ok:
            bad += 1
            print("MISMATCH (CG level only: the products are exact to rounding)", tag, f"res={res:.3e} tol={tol:.3e} zerr={zerr:.3e}", flush=True)
        else:
            if dtype == "f32" and res > 1.2 * tol:
                drift += 1
                print("drift", tag, f"fp32 CG residual {res / tol:.1f}x its stopping tolerance with every product within {errs[worst]:.1e} ({worst}) of scipy", flush=True)
            elif c % 10 == 0:
                print("ok", tag, f"worst operator error {errs[worst]:.1e} ({worst}), res/tol={res / tol:.2f} zerr={zerr:.1e}", flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300], flush=True)
print(f"{cases} cases, {bad} mismatches, {drift} fp32 cases inside the 10x drift band (products exact), {time.time() - t0:.0f} s")
