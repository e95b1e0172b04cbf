"""Ad-hoc: single-QP iteration rate at launch-bound sizes (hipGraph replay on/off is chosen by QPS_GRAPH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
for (n, m) in ((256, 512), (1024, 2048), (2048, 4096)):
    P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3)
    with q.QuadraticProgram(P, qq, A, l, u) as prob:
        best = 0.0
        for rep in range(4):
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=2000, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True, info=info)
            best = max(best, info["iterations"] / info["tLoop"])
    print(f"QPS_GRAPH={os.environ.get('QPS_GRAPH', 'default')} n={n:5d} m={m:5d}: {best:9.0f} it/s ({1e6/best:5.1f} us/it)", flush=True)
