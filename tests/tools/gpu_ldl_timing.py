"""Per-iteration time of the sparse KKT plugin (eager launches vs hipGraph replay: run once with QPS_GRAPH=0, once without)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as q
from quadraticprogramsolver_amd import _lib
import ctypes as C
cases = [(q.ProblemClass.lassoOptimization, 100), (q.ProblemClass.huberFitting, 100), (q.ProblemClass.supportVectorMachine, 100),
         (q.ProblemClass.portfolioOptimization, 1000), (q.ProblemClass.isotonicRegression, 800), (q.ProblemClass.randomQp, 100)]
print("QPS_GRAPH =", os.environ.get("QPS_GRAPH", "(default)"))
for pc, n in cases:
    P, qq, A, l, u = q.GenerateRandomQP(pc, n, rng=q.make_rng(4321, 1000 * int(pc) + 1))
    with q.QuadraticProgram(P, qq, A, l, u, linsys="ldl") as prob:
        x = np.zeros(P.shape[0]); info = {}
        prob.solve(x, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)          # symbolic + first factorisation
        t_setup = info["tSetup"]
        best = 1e9
        for rep in range(3):
            x = np.zeros(P.shape[0]); info = {}
            prob.solve(x, numIterations=1000, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, reuseFactor=True, info=info)
            best = min(best, info["tLoop"])
        x = np.zeros(P.shape[0]); info = {}
        t0 = time.perf_counter()
        flag = prob.solve(x, numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=info)
        dt = time.perf_counter() - t0
    print(f"{pc.name:24s} n={n:5d} N={P.shape[0]:6d} M={A.shape[0]:6d}: setup {t_setup*1e3:7.2f} ms; {best/1000*1e6:7.2f} us/iteration; "
          f"RunTests solve: flag {int(flag)} {info['iterations']} its {info['numRefactor']} refactors {dt*1e3:7.2f} ms (refactor {info['tRefactor']*1e3:.2f} ms)")
