"""Ad-hoc: per-call cost of the CSR/CG handles and of ProxQP handles at small sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import quadraticprogramsolver_amd as q
for (n, m) in ((64, 128), (500, 1000)):
    P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3, feasible=True)
    Ps, As = sp.csc_matrix(P), sp.csc_matrix(A)
    q.QuadraticProgram(Ps, qq, As, l, u, linsys="cg").close()
    tc = ts = td = 0.0; reps = 10
    for _ in range(reps):
        t0 = time.perf_counter(); prob = q.QuadraticProgram(Ps, qq, As, l, u, linsys="cg"); t1 = time.perf_counter()
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info); t2 = time.perf_counter()
        prob.close(); t3 = time.perf_counter()
        tc += t1 - t0; ts += t2 - t1; td += t3 - t2
    print(f"CSR/CG n={n:5d} m={m:5d}: create {tc/reps*1e3:7.3f} ms  solve(50 its, {info['cgIterations']} CG its) {ts/reps*1e3:7.3f} ms  destroy {td/reps*1e3:7.3f} ms", flush=True)
