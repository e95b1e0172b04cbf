"""Run-to-run determinism of the CSR/CG path on the C3-shaped problem: same handle, same call, bitwise-equal x and equal CG counts?"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
n, m = int(os.environ.get("QPS_N", 50000)), int(os.environ.get("QPS_M", 100000))
P, qq, A, l, u = q.GenerateSparseBenchmarkQP(n, m)
for mode, env in (("stream", {"QPS_SPMV_BLOCKED": "0"}), ("blocked, separate P/A/A'", {"QPS_SPMV_BLOCKED": "1", "QPS_SPMV_FUSEPA": "0"}),
                  ("blocked, stacked [P;A] + fused u", {"QPS_SPMV_BLOCKED": "1", "QPS_SPMV_FUSEPA": "1"})):
    os.environ.update(env)
    with q.QuadraticProgram(P, qq, A, l, u, linsys="cg") as prob:
        for rep in range(4):
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=int(os.environ.get("QPS_K", 12)), ϵAbs=0.0, ϵRel=0.0, info=info)
            print(f"{mode:36s} rep {rep}: cg {info['cgIterations']:6d}  x sha {hashlib.sha1(x.tobytes()).hexdigest()[:12]}  |x| {np.abs(x).max():.15e}", flush=True)
