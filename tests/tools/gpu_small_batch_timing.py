"""Ad-hoc: throughput of a batch of small QPs (one workgroup per QP, register-resident kernel) vs solving them one after the other."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
for (cnt, n, m) in ((32, 64, 128), (256, 64, 128), (256, 100, 50)):
    probs = [q.GenerateDenseBenchmarkQP(n, m, stream=b) for b in range(cnt)]
    with q.QuadraticProgramBatch(probs) as batch:
        batch.solve(numIterations=50, ϵAbs=0.0, ϵRel=0.0)
        best = 0.0
        for rep in range(3):
            X, flags, infos = batch.solve(numIterations=2000, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True)
            best = max(best, cnt * 2000 / infos[0]["tLoop"])
    with q.QuadraticProgram(*probs[0]) as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=2000, ϵAbs=0.0, ϵRel=0.0, info=info)
        prob.solve(x, numIterations=2000, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True, info=info)
        single = 2000 / info["tLoop"]
    print(f"QPS_SMALL_REG={os.environ.get('QPS_SMALL_REG', 'default')} {cnt:4d} x (n={n}, m={m}): batch {best:12.0f} QP-it/s   one QP alone {single:9.0f} it/s   ratio {best/single:6.1f}", flush=True)
