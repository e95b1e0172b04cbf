"""Ad-hoc C2 timing: one configuration, all-kernel profile (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
n, m = 4096, 8192
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m)
dtype = os.environ.get("QPS_DTYPE", "f64")
prob = q.QuadraticProgram(P, qq, A, l, u, dtype=dtype)
nb = int(os.environ.get("QPS_NB", "0"))
x = np.zeros(n); info = {}
prob.solve(x, numIterations=25, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, info=info)
best = 0
for rep in range(3):
    x = np.zeros(n)
    prob.solve(x, numIterations=400, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, reuseFactor=True, info=info)
    best = max(best, info['iterations'] / info['tLoop'])
print(f"{dtype} nb={nb} PASS_THREADS={os.environ.get('QPS_PASS_THREADS','512')}: setup {info['tSetup']*1e3:.1f} ms, best {best:.0f} it/s ({1e6/best:.1f} us/it)", flush=True)
prob.set_profiling(2)
x = np.zeros(n)
prob.solve(x, numIterations=100, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, reuseFactor=True, info=info)
for k in prob.kernel_times():
    us = k['seconds'] / k['launches'] * 1e6
    print(f"   {k['name']:24s} {us:9.1f} us/launch  {k['algo_bytes']/us/1e6:8.3f} TB/s algorithmic  ({k['launches']} launches)")
