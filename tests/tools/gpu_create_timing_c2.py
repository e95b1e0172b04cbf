"""Ad-hoc: what the boundary's host-buffer hand-over costs at BASELINE's headline size (dense n = 4096, m = 8192): qps_create_dense (upload of P 128 MiB + A 256 MiB from pageable
host memory + layout import), first solve (setup + 725 iterations to eps = 1e-6), destroy -- the PCIe-inclusive counterpart of bench.py's resident-data figures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
n, m = 4096, 8192
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, seed=1234, feasible=True)
Pf, Af = np.asfortranarray(P), np.asfortranarray(A)
q.QuadraticProgram(Pf, qq, Af, l, u).close()
for dtype in ("f64", "f32"):
    tc, ts, td, reps = 0.0, 0.0, 0.0, 5
    for _ in range(reps):
        t0 = time.perf_counter(); prob = q.QuadraticProgram(Pf, qq, Af, l, u, dtype=dtype); t1 = time.perf_counter()
        x = np.zeros(n); info = {}
        eps = 1e-6 if dtype == "f64" else 1e-4
        prob.solve(x, numIterations=50000, ϵAbs=eps, ϵRel=eps, ρ=0.1, adptΡ=True, info=info); t2 = time.perf_counter()
        prob.close(); t3 = time.perf_counter()
        tc += t1 - t0; ts += t2 - t1; td += t3 - t2
    gb = (n * n + m * n) * 8 / 1e9
    print(f"{dtype}: create {tc / reps * 1e3:7.2f} ms ({gb / (tc / reps):5.1f} GB/s of host arrays)  solve to eps {ts / reps * 1e3:7.2f} ms (setup {info['tSetup'] * 1e3:.2f}, loop {info['tLoop'] * 1e3:.2f}, "
          f"{info['iterations']} iterations)  destroy {td / reps * 1e3:6.2f} ms  => create + solve {(tc + ts) / reps * 1e3:7.2f} ms", flush=True)
