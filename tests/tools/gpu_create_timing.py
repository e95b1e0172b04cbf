"""Ad-hoc: handle creation / destruction / first-solve cost at small sizes (what one RunTests.jl-style call pays besides the loop)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
for (n, m) in ((10, 5), (64, 128), (100, 200), (1024, 2048)):
    P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3, feasible=True)
    q.QuadraticProgram(P, qq, A, l, u).close()
    tc = ts = td = 0.0; reps = 20
    for _ in range(reps):
        t0 = time.perf_counter(); prob = q.QuadraticProgram(P, qq, A, l, u); t1 = time.perf_counter()
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=200, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info); t2 = time.perf_counter()
        prob.close(); t3 = time.perf_counter()
        tc += t1 - t0; ts += t2 - t1; td += t3 - t2
    print(f"n={n:5d} m={m:5d}: create {tc/reps*1e3:7.3f} ms  solve(200 its) {ts/reps*1e3:7.3f} ms (setup {info['tSetup']*1e3:.3f} loop {info['tLoop']*1e3:.3f})  destroy {td/reps*1e3:7.3f} ms", flush=True)
