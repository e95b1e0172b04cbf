"""Soak run of the single-launch blocked sweeps (not a test): thousands of linear solves through the literal plugin pair, EVERY one checked on the host
against (P + sigma I + rho A'A) x~ = sigma x - q + A'(rho z - y) -- a stale or torn hand-off granule in any launch would show as a residual.
usage: python tests/tools/gpu_trsv_blocked_soak.py [solves] [n] [nb]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
m = 64
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=77)
rng = np.random.default_rng(1)
rho, sigma = 0.1, 1e-6
M = P + sigma * np.eye(n) + rho * (A.T @ A)
worst, bad, t0 = 0.0, 0, time.time()
with q.QuadraticProgram(P, qq, A, l, u) as prob:
    prob.linsys_init(rho, sigma, trsvBlock=nb)
    xx, zz = np.zeros(n), np.zeros(m)
    for k in range(solves):
        x, z, y = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4), rng.standard_normal(m), rng.standard_normal(m)
        prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz)
        rhs = sigma * x - qq + A.T @ (rho * z - y)
        res = np.abs(M @ xx - rhs).max() / max(np.abs(rhs).max(), 1e-300)
        worst = max(worst, res)
        if not (res <= 1e-9) or not np.all(np.isfinite(xx)):
            bad += 1
            print(f"solve {k}: relative residual {res:.3e}", flush=True)
        if (k + 1) % 500 == 0:
            print(f"{k + 1} solves ({2 * (k + 1)} sweep launches), worst relative residual {worst:.2e}, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    xk = np.zeros(n); info = {}
    prob.solve(xk, numIterations=10, ϵAbs=0.0, ϵRel=0.0, ρ=rho, trsvBlock=nb, info=info)
print(f"n={n} trsvBlock={nb}: {solves} solves, worst relative residual {worst:.2e}, {bad} bad; sweep variant still {info['sweepVariant']}")
