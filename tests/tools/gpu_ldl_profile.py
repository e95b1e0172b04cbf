"""Kernel trace target for the sparse KKT plugin: lasso numElements = 100 (N = M = 10 200), 1000 plain iterations + a RunTests.jl solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as q
P, qq, A, l, u = q.GenerateRandomQP(q.ProblemClass.lassoOptimization, 100, rng=q.make_rng(4321, 6001))
with q.QuadraticProgram(P, qq, A, l, u, linsys="ldl") as prob:
    x = np.zeros(P.shape[0]); info = {}
    prob.solve(x, numIterations=1000, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
    print("1000 iterations:", info["tLoop"] * 1e3, "ms; setup", info["tSetup"] * 1e3, "ms")
    x = np.zeros(P.shape[0]); info = {}
    flag = prob.solve(x, numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=info)
    print("RunTests solve: flag", int(flag), info["iterations"], "iterations", info["numRefactor"], "refactors", info["tLoop"] * 1e3, "ms")
