"""Ad-hoc: run a few refactorisations at C2 size (for rocprofv3 --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(4096, 8192)
prob = q.QuadraticProgram(P, qq, A, l, u, dtype=os.environ.get("QPS_DTYPE", "f64"))
info = {}
x = np.zeros(4096)
prob.solve(x, numIterations=400, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=50, info=info)
print(info)
