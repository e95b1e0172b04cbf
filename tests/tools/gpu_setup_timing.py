"""Ad-hoc: setup / refactor timing at C2 size for fp64 and fp32 (config 5 style run).  Not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
P, qq, A, l, u = q.GenerateDenseBenchmarkQP(4096, 8192)
for dt in ("f64", "f32"):
    prob = q.QuadraticProgram(P, qq, A, l, u, dtype=dt)
    for rep in range(3):
        x = np.zeros(4096); info = {}
        prob.solve(x, numIterations=25, ϵAbs=0.0, ϵRel=0.0, info=info)
    print(dt, f"setup {info['tSetup']*1e3:.2f} ms")
    for nb in (0, 2048, 1024):
        x = np.zeros(4096)
        prob.solve(x, numIterations=500, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=50, trsvBlock=nb, info=info)
        print(dt, f"c5-style nb={nb}: iters {info['iterations']} refactors {info['numRefactor']} loop {info['tLoop']*1e3:.1f} ms "
              f"(refactor {info['tRefactor']*1e3:.1f} ms = {info['tRefactor']/max(info['numRefactor'],1)*1e3:.2f} ms each) -> {info['iterations']/info['tLoop']:.0f} it/s")
    prob.close()
