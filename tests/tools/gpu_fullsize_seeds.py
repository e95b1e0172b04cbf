"""Ad-hoc: the full-size parity checks of tests/test_gpu_fullsize.py over several seeds and several sizes at which the tile grids of the setup GEMMs are large
(mirror-tile pairing, XCD-aware lower-tile order, fused Cholesky with the XCD rule): fp64 iterates against the oracle at 1e-9, fp32 on the refactor-per-check
schedule against the fp64 oracle at 1e-3, adaptive rho to eps with equal flag / iterations / refactor counts.  Not a test.
usage: python tests/tools/gpu_fullsize_seeds.py [seeds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
from oracle import c_oracle as co
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
bad = 0; t0 = time.time()
for (n, m) in ((4096, 8192), (4032, 2000), (2112, 4300), (6208, 1500)):
    for s in range(seeds):
        P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, seed=4000 + s, feasible=True)
        xo, io = co.solve(P, qq, A, l, u, numIterations=60, epsAbs=0.0, epsRel=0.0)
        xa, ia = co.solve(P, qq, A, l, u, numIterations=4000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True)
        x5, i5 = co.solve(P, qq, A, l, u, numIterations=100, epsAbs=0.0, epsRel=0.0, rho=0.1, adptRho=True, fctrRho=1.0, numItrConv=50)
        with q.QuadraticProgram(P, qq, A, l, u) as prob:
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=60, ϵAbs=0.0, ϵRel=0.0, info=info); z, y = prob.dual()
            d64 = max(rel(x, xo), rel(z, io["z"]), rel(y, io["y"]))
            xb = np.zeros(n); ib = {}
            prob.solve(xb, numIterations=60, ϵAbs=0.0, ϵRel=0.0, trsvBlock=1024, info=ib)
            dblk = rel(xb, xo)
            x = np.zeros(n); ie = {}
            fl = prob.solve(x, numIterations=4000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True, info=ie)
            deps = rel(x, xa); same = int(fl) == ia["convFlag"] and ie["iterations"] == ia["iterations"] and ie["numRefactor"] == ia["numRefactor"]
        with q.QuadraticProgram(P, qq, A, l, u, dtype="f32") as p32:
            x = np.zeros(n); i3 = {}
            p32.solve(x, numIterations=100, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=50, info=i3)
            d32 = rel(x, x5); same32 = i3["numRefactor"] == i5["numRefactor"]
        ok = d64 <= 1e-9 and dblk <= 1e-9 and deps <= 1e-5 and same and d32 <= 1e-3 and same32
        bad += not ok
        print(f"{'ok' if ok else 'MISMATCH'} n={n} m={m} seed={4000 + s}: fp64 K=60 {d64:.1e} (trsvBlock 1024, variant {ib['sweepVariant']}: {dblk:.1e}); to eps: flag {int(fl)}/{ia['convFlag']} "
              f"its {ie['iterations']}/{ia['iterations']} refactors {ie['numRefactor']}/{ia['numRefactor']} x {deps:.1e}; fp32 refactor schedule K=100: {d32:.1e}, refactors "
              f"{i3['numRefactor']}/{i5['numRefactor']}", flush=True)
print(f"{4 * seeds} cases, {bad} bad, {time.time() - t0:.0f} s")
