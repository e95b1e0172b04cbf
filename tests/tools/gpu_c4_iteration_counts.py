"""Per-QP iteration counts of BASELINE config 4's problem set run to a tolerance (run on the GPU box through gpurun).
256 QPs (n = 1024, m = 2048, seed 1234, stream b, feasible variant), eps = 1e-6, rho0 = 0.1, adaptive rho -- the parameters of bench.py's time_to_eps leg --
solved in slabs of 32 through qps_solve_batch.  Writes gpurun_out/c4_time_to_eps_iterations.json; the file is committed as
tests/golden/c4_time_to_eps_iterations.json and is what tests/test_dist_cpu.py balances the ranks on."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as qps

n, m, total, slab = 1024, 2048, 256, 32
its, flags, refac = [], [], []
for b0 in range(0, total, slab):
    probs = [qps.GenerateDenseBenchmarkQP(n, m, seed=1234, stream=b, feasible=True) for b in range(b0, b0 + slab)]
    with qps.QuadraticProgramBatch(probs) as sb:
        _, fl, infos = sb.solve(numIterations=50000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
    its += [int(i["iterations"]) for i in infos]; flags += [int(f) for f in fl]; refac += [int(i["numRefactor"]) for i in infos]
    print(f"QPs {b0}..{b0 + slab - 1}: iterations {min(its[b0:])}..{max(its[b0:])}", flush=True)
out = {"workload": "BASELINE configs[3]: 256 dense QPs n=1024 m=2048, GenerateDenseBenchmarkQP(seed=1234, stream=b, feasible=True)",
       "params": "numIterations=50000, epsAbs=epsRel=1e-6, rho=0.1, adptRho=True, fp64, batched solver, slabs of 32",
       "command": "python tests/tools/gpu_c4_iteration_counts.py (one MI355X, through gpurun)", "iterations": its, "flags": flags, "refactorisations": refac}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c4_time_to_eps_iterations.json", "w"))
print("min", min(its), "max", max(its), "sum", sum(its), "flags", sorted(set(flags)))
