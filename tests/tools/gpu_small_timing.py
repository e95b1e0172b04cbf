"""Ad-hoc: iterations/s at the reference's own test sizes, GPU loop vs the CPU port (1 thread)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
from oracle import c_oracle as co
for (n, m) in ((10, 5), (64, 128), (100, 50), (256, 512), (512, 1024), (1024, 2048)):
    P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3)
    with q.QuadraticProgram(P, qq, A, l, u) as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=100, ϵAbs=0.0, ϵRel=0.0, info=info)
        x = np.zeros(n)
        prob.solve(x, numIterations=2000, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True, info=info)
        g = info["iterations"] / info["tLoop"]
    xo, io = co.solve(P, qq, A, l, u, numIterations=2000 if n <= 256 else 200, epsAbs=0.0, epsRel=0.0, numThreads=1)
    c = io["iterations"] / io["tLoop"]
    print(f"n={n:5d} m={m:5d}: GPU {g:10.0f} it/s ({1e6/g:6.1f} us/it)   CPU port 1 thread {c:10.0f} it/s   ratio {g/c:6.2f}", flush=True)
