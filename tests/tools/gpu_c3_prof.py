import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
P, qq, A, l, u = q.GenerateSparseBenchmarkQP(50000, 100000)
prob = q.QuadraticProgram(P, qq, A, l, u, linsys="cg")
x = np.zeros(50000); info = {}
prob.solve(x, numIterations=60, ϵAbs=0.0, ϵRel=0.0, info=info)
print(info["iterations"] / info["tLoop"], info["cgIterations"] / info["tLoop"], info["cgIterations"])
