"""Ad-hoc check + timing of the single-launch blocked sweeps (k_trsv_blocked.hip); not a test.
usage: python tests/tools/gpu_trsv_blocked.py [n] [m] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
nbs = [int(v) for v in os.environ.get("QPS_NBS", "4096,1024,512,2048").split(",")]
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m)
prob = q.QuadraticProgram(P, qq, A, l, u, dtype=dtype)
rng = np.random.default_rng(5)
rho, sigma = 0.1, 1e-6
x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
rhs = sigma * x - qq + A.T @ (rho * z - y)
ref = None
for nb in nbs:
    prob.linsys_init(rho, sigma, trsvBlock=nb)
    xx, zz = np.zeros(n), np.zeros(m)
    prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz)
    lhs = P @ xx + sigma * xx + rho * (A.T @ (A @ xx))
    res = np.abs(lhs - rhs).max() / np.abs(rhs).max()
    if ref is None: ref = xx.copy()
    print(f"nb={nb}: linear-solve residual {res:.2e}, max|x - x_first|/max|x| {np.abs(xx - ref).max() / np.abs(ref).max():.2e}", flush=True)
for nb in nbs:
    xk = np.zeros(n); info = {}
    prob.solve(xk, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=rho, trsvBlock=nb, info=info)
    if nb == nbs[0]: x0 = xk.copy()
    best = 0
    for rep in range(3):
        xk2 = np.zeros(n)
        prob.solve(xk2, numIterations=300, ϵAbs=0.0, ϵRel=0.0, ρ=rho, trsvBlock=nb, reuseFactor=True, info=info)
        best = max(best, info['iterations'] / info['tLoop'])
    print(f"nb={nb} variant={info['sweepVariant']}: setup {info['tSetup']*1e3:.1f} ms, {best:.0f} it/s ({1e6/best:.1f} us/it), iterates vs first {np.abs(xk - x0).max() / np.abs(x0).max():.2e}", flush=True)
    prob.set_profiling(2)
    xk2 = np.zeros(n)
    prob.solve(xk2, numIterations=100, ϵAbs=0.0, ϵRel=0.0, ρ=rho, trsvBlock=nb, reuseFactor=True, info=info)
    for k in prob.kernel_times():
        if k['launches'] == 0: continue
        us = k['seconds'] / k['launches'] * 1e6
        if 'trsv' in k['name'] or 'sweep' in k['name'] or 'colsum' in k['name']:
            print(f"   {k['name']:46s} {us:9.1f} us/launch  {k['algo_bytes']/us/1e6:8.3f} TB/s algorithmic  ({k['launches']} launches)")
    prob.set_profiling(0)
