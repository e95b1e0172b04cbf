"""Time the ProxQP.jl-form loop on the GPU: fused single pass over [A; C] (loopVariant 0) vs the unfused loop (1).
usage: python tests/tools/gpu_proxqp_timing.py [n me mi iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as qps
from quadraticprogramsolver_amd.generator import make_rng

n, me, mi, iters = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 1024, 7168, 500)))
rng = make_rng(1220, 3)
M = rng.standard_normal((n, n)) / np.sqrt(n); P = M.T @ M + 0.01 * np.eye(n); P = 0.5 * (P + P.T)
q = rng.standard_normal(n); A = rng.standard_normal((me, n)) / np.sqrt(n); C = rng.standard_normal((mi, n)) / np.sqrt(n)
x0 = rng.standard_normal(n); b = A @ x0; d = C @ x0 + 0.3 * np.abs(rng.standard_normal(mi)) - 0.1
z = np.zeros
res = {}
for variant in (1, 0, 1, 0):
    with qps.ProxQP(P, q, A, b, C, d, z(n), z(me), z(mi), z(mi)) as prob:
        qps.SolveQuadraticProgramProxQP(prob, numIterations=10, numItrConv=1000, adptΡ=False, loopVariant=variant)   # warm up + factorise
        prob2 = prob
        t0 = time.perf_counter()
        rep = qps.SolveQuadraticProgramProxQP(prob, numIterations=iters, numItrConv=50, adptΡ=False, loopVariant=variant)
        dt = time.perf_counter() - t0
        res[variant] = (dt, prob.vX.copy(), rep)
        print(f"n={n} me={me} mi={mi} variant={variant}: {iters} iterations (incl. factorisation) {dt*1e3:.1f} ms -> {iters/dt:.0f} it/s  resP={rep['PrimalResidual']:.2e} resD={rep['DualResidual']:.2e}", flush=True)
print("max |x_fused - x_unfused| =", np.abs(res[0][1] - res[1][1]).max())
