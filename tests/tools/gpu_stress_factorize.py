"""Bounded stress of one parity-test scenario (isotonic n=100 through the CSC dense path, five solves per handle) to
characterise an intermittent factorisation breakdown: reports which solve failed and whether a retry on the same handle fails too."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as qps
from quadraticprogramsolver_amd.generator import GenerateRandomQP, ProblemClass, make_rng

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
pc = ProblemClass.isotonicRegression
P, q, A, l, u = GenerateRandomQP(pc, 100, numConstraints=0, rng=make_rng(1234, 40 + int(pc)), dense=False, densityFctr=None)
fails = 0
ref = None
for rep in range(reps):
    with qps.QuadraticProgram(P, q, A, l, u) as prob:
        for si, (K, nb, variant) in enumerate(((25, 0, 0), (100, 64, 0), (50, 256, 1), (75, 0, 2), (60, 0, 0))):
            x = np.zeros(P.shape[0])
            try:
                prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, loopVariant=variant)
            except qps.QpsError as e:
                fails += 1
                again = "ok"
                try:
                    prob.solve(np.zeros(P.shape[0]), numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, loopVariant=variant)
                except qps.QpsError as e2:
                    again = "fails again: " + e2.message[-60:]
                print(f"rep {rep} solve {si} (K={K} nb={nb} variant={variant}): {e.message[-70:]} | retry {again}", flush=True)
                break
            if si == 0:
                if ref is None:
                    ref = x.copy()
                elif not np.array_equal(ref, x):
                    print(f"rep {rep}: first solve differs from rep 0 by {np.abs(ref - x).max():.3e}", flush=True)
print(f"{fails} failures in {reps} handles", flush=True)
