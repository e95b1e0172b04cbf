"""Ad-hoc A/B of the register-resident single-launch kernel against the LDS-vector one (QPS_SMALL_REG is read per process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
for dtype in ("f64", "f32"):
    for (n, m) in ((64, 128), (100, 50), (100, 100), (128, 128), (100, 200), (64, 256)):
        P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3)
        with q.QuadraticProgram(P, qq, A, l, u, dtype=dtype) as prob:
            best = 0.0
            for rep in range(3):
                x = np.zeros(n); info = {}
                prob.solve(x, numIterations=2000, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True, info=info)
                best = max(best, info["iterations"] / info["tLoop"])
        print(f"QPS_SMALL_REG={os.environ.get('QPS_SMALL_REG', 'default')} {dtype} n={n:4d} m={m:4d}: {best:9.0f} it/s ({1e6/best:5.2f} us/it)", flush=True)
