"""Ad-hoc timing of BASELINE config 3 (sparse n=50k, m=100k, ~0.1 % nnz, CSR/CG path).  Not a test."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
n, m = int(os.environ.get("QPS_N", 50000)), int(os.environ.get("QPS_M", 100000))
t = time.time(); P, qq, A, l, u = q.GenerateSparseBenchmarkQP(n, m); print(f"gen {time.time()-t:.1f}s nnz(P)={P.nnz} nnz(A)={A.nnz}", flush=True)
t = time.time(); prob = q.QuadraticProgram(P, qq, A, l, u, linsys="cg"); print(f"create {time.time()-t:.2f}s", flush=True)
for K in (25, 200):
    x = np.zeros(n); info = {}
    prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, info=info)
    print(f"K={K}: loop {info['tLoop']*1e3:.1f} ms -> {info['iterations']/info['tLoop']:.1f} ADMM it/s, CG its {info['cgIterations']} ({info['cgIterations']/info['tLoop']:.0f} CG it/s)", flush=True)
prob.set_profiling(2)
x = np.zeros(n); prob.solve(x, numIterations=100, ϵAbs=0.0, ϵRel=0.0, info=info)
for k in prob.kernel_times():
    us = k['seconds'] / k['launches'] * 1e6
    print(f"   {k['name']:52s} {us:9.1f} us/launch  {k['algo_bytes']/us/1e6:8.3f} TB/s algorithmic  ({k['launches']} launches)")
prob.set_profiling(0)
if os.environ.get("QPS_SKIP_EPS"): sys.exit(0)
x = np.zeros(n); t = time.time()
flag = prob.solve(x, numIterations=20000, ρ=0.1, adptΡ=True, info=info)
print(f"time-to-eps(1e-6): flag {int(flag)} its {info['iterations']} cg {info['cgIterations']} loop {info['tLoop']*1e3:.1f} ms")
if os.environ.get("QPS_ORACLE"):
    from oracle import c_oracle as co
    t = time.time(); xo, io = co.solve(P, qq, A, l, u, numIterations=25, epsAbs=0.0, epsRel=0.0, linsys=3); print("oracle 25 its", time.time()-t, io['tLoop'], io['cgIterations'])
