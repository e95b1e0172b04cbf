"""Ad-hoc timing of BASELINE config 4 on one GPU: 32 dense QPs (n=1024, m=2048, fp64) = one rank's slab of 256/8."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
cnt, n, m = int(os.environ.get("QPS_CNT", 32)), 1024, 2048
t = time.time(); probs = [q.GenerateDenseBenchmarkQP(n, m, stream=b) for b in range(cnt)]; print(f"gen {time.time()-t:.1f}s", flush=True)
t = time.time(); batch = q.QuadraticProgramBatch(probs); print(f"create {time.time()-t:.2f}s", flush=True)
for K in (25, 500, 500):
    X, flags, infos = batch.solve(numIterations=K, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True)
    i0 = infos[0]
    print(f"K={K}: setup {i0['tSetup']*1e3:.1f} ms loop {i0['tLoop']*1e3:.1f} ms -> {K/i0['tLoop']:.0f} batched it/s = {cnt*K/i0['tLoop']:.0f} QP-it/s; "
          f"bytes/QP-it 25.4MB -> {cnt*K/i0['tLoop']*25.4e6/1e12:.2f} TB/s", flush=True)
