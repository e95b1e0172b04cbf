"""CG-iteration time of the explicit reduced-matrix plugin (ItrSolCg, QPS_LINSYS_CG_EXPLICIT: one product per CG iteration) beside the matrix-free one (LinOpCg: three
products) on the structured classes where a plain CG request picks the explicit matrix by itself (round-3 review item 6): isotonic regression (the reference generator's
class 9: dense-ish P, bidiagonal A) and banded / control-like A at two sizes.  Fixed 40 ADMM iterations, eps = 0, inner CG at its reference defaults (epsPcg = 1e-6).
usage: python tests/tools/gpu_cg_explicit_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import quadraticprogramsolver_amd as q


def banded(n, m, bw, seed):
    rng = np.random.default_rng(seed)
    A = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-bw, bw + 1)], list(range(-bw, bw + 1)), shape=(n, n), format="csc")[:m, :]
    T = sp.diags([0.1 * np.ones(n - 1), 1.0 + rng.random(n), 0.1 * np.ones(n - 1)], [-1, 0, 1], format="csc")
    x0 = rng.standard_normal(n); c = A @ x0
    return sp.csc_matrix(T), rng.standard_normal(n), sp.csc_matrix(A), c - rng.random(m), c + rng.random(m)


cases = [("isotonicRegression n=800", q.GenerateRandomQP(q.ProblemClass.isotonicRegression, 800, rng=q.make_rng(5, 1))),
         ("isotonicRegression n=2000", q.GenerateRandomQP(q.ProblemClass.isotonicRegression, 2000, rng=q.make_rng(5, 2))),
         ("banded n=20000 m=19000 bw=3", banded(20000, 19000, 3, 1)),
         ("banded n=400000 m=390000 bw=5", banded(400000, 390000, 5, 2))]
K = 40
for name, (P, qq, A, l, u) in cases:
    n = P.shape[0]
    row = []
    for linsys, env in (("cg_explicit", None), ("cg", "0")):
        if env is None:
            os.environ.pop("QPS_CG_EXPLICIT", None)
        else:
            os.environ["QPS_CG_EXPLICIT"] = env
        t0 = time.perf_counter()
        with q.QuadraticProgram(P, qq, A, l, u, linsys=linsys) as prob:
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=5, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)          # builds the explicit matrix on first use
            t1 = time.perf_counter()
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
        row.append((linsys, info["cgExplicit"], info["cgIterations"], info["tLoop"], t1 - t0, x))
        print(f"   ... {name} {linsys} done: {info['cgIterations']} CG its, loop {info['tLoop']:.3f} s", flush=True)
    os.environ.pop("QPS_CG_EXPLICIT", None)
    (la, ea, ca, ta, sa, xa), (lb, eb, cb, tb, sb, xb) = row
    print(f"{name}: nnz P {P.nnz}, nnz A {A.nnz}", flush=True)
    print(f"   explicit matrix (ItrSolCg):  cgExplicit={ea} {ca:6d} CG its in {K} ADMM its, {1e6 * ta / max(ca, 1):7.1f} us per CG iteration, {K / ta:8.0f} ADMM it/s; create + first solve {1e3 * sa:7.1f} ms")
    print(f"   matrix-free (LinOpCg):       cgExplicit={eb} {cb:6d} CG its in {K} ADMM its, {1e6 * tb / max(cb, 1):7.1f} us per CG iteration, {K / tb:8.0f} ADMM it/s; create + first solve {1e3 * sb:7.1f} ms")
    print(f"   x agree to {np.abs(xa - xb).max() / max(1.0, np.abs(xb).max()):.1e} (two inexact inner solves at epsPcg = 1e-6)", flush=True)
