"""QP-iterations/s of the in-process multi-device batch driver against the chunk size, on ONE card (round-3 review item 4: "report QP-it/s vs chunk size so the cost of
small chunks is known"): BASELINE config 4's shape (dense n = 1024, m = 2048, fp64), 64 QPs, fixed K = 100 iterations and a run to eps = 1e-6.  Workers share device 0.
Every range pays its own handle: upload of its QPs (25 MB each), A'A, Cholesky, inverse -- that is the cost of a small chunk, beside the lower rate of a batched launch
that carries fewer QPs.  usage: python tests/tools/gpu_batch_multi_timing.py [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import quadraticprogramsolver_amd as q

cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, m = 1024, 2048
probs = [q.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=b, feasible=True) for b in range(cnt)]
print(f"{cnt} QPs n={n} m={m} fp64, one MI355X; rates include handle creation (upload + setup) of every range")
with q.QuadraticProgramBatch(probs) as one:
    one.solve(numIterations=5, ϵAbs=0.0, ϵRel=0.0)
    t0 = time.perf_counter(); one.solve(numIterations=100, ϵAbs=0.0, ϵRel=0.0, reuseFactor=True); t1 = time.perf_counter() - t0
print(f"single resident handle, all {cnt} QPs, factor cached, K=100: {cnt * 100 / t1:10.0f} QP-it/s  (the bench.py figure: data resident, setup excluded)")
for workers in (1, 2):
    for chunk in (8, 16, 32, 0):
        for tag, kw, unit in (("K=100", dict(numIterations=100, ϵAbs=0.0, ϵRel=0.0), None), ("eps=1e-6", dict(numIterations=50000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True), None)):
            with q.QuadraticProgramBatch(probs, devices=[0] * workers, chunk=chunk) as multi:
                t0 = time.perf_counter()
                X, flags, infos = multi.solve(**kw)
                dt = time.perf_counter() - t0
                its = sum(i["iterations"] for i in infos)
                print(f"workers={workers} chunk={chunk:2d} {tag:9s}: {its / dt:10.0f} QP-it/s  wall {dt * 1e3:7.1f} ms  worker busy {['%.0f ms' % (s * 1e3) for s in multi.worker_seconds]}", flush=True)
