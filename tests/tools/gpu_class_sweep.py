"""Every problem class of the reference generator (GenerateQuadraticProgram.jl:8-115) at RunTests.jl's two sizes, through each plugin family of the device path, with
RunTests.jl's parameters (eps = 1e-7, rho0 = 0.1, adaptive; RunTests.jl:50-58): wall time of one whole SolveQuadraticProgram!-style call (handle creation + solve),
iterations, flag -- a sweep to spot a class on which some plugin is pathologically slow.  usage: python tests/tools/gpu_class_sweep.py [size ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import quadraticprogramsolver_amd as q

sizes = [int(a) for a in sys.argv[1:]] or [10, 100]
print("class, numElements, N x M, nnz P / A | plugin: ms per call (create + solve), iterations, flag, cg iterations")
for pc in q.ProblemClass:
    for ne in sizes:
        P, qq, A, l, u = q.GenerateRandomQP(pc, ne, rng=q.make_rng(1234, int(pc) * 10 + ne))
        n, m = P.shape[0], A.shape[0]
        line = f"{pc.name:26s} {ne:4d} {n:6d} x {m:6d} nnz {P.nnz:8d} / {A.nnz:8d} |"
        ref = None
        for linsys in ("ldl", "cg", "cholesky"):
            if linsys == "cholesky" and (n > 12000 or m > 40000):
                line += "  cholesky: (skipped: dense copy too large)"; continue
            try:
                best, info, flag = 1e9, {}, 0
                for rep in range(2):
                    t0 = time.perf_counter()
                    with q.QuadraticProgram(P, qq, A, l, u, linsys=linsys) as prob:
                        x = np.zeros(n); info = {}
                        flag = prob.solve(x, numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=info)
                    best = min(best, time.perf_counter() - t0)
                if ref is None:
                    ref = x.copy()
                dev = float(np.abs(x - ref).max())
                line += f"  {linsys}: {best * 1e3:8.1f} ms {info['iterations']:6d} its flag {int(flag)}" + (f" cg {info['cgIterations']} ({'explicit' if info['cgExplicit'] else 'matrix-free'})" if linsys == "cg" else "") + (f" dx {dev:.0e}" if linsys != "ldl" else "")
            except Exception as e:
                line += f"  {linsys}: {type(e).__name__} {str(e)[:60]}"
            print("   ...", pc.name, ne, linsys, flush=True)
        print(line, flush=True)
