"""Ad-hoc: follows ONE case of tests/tools/gpu_fuzz_proxqp.py over growing iteration counts and prints, per K, the deviation of the device state from the
numpy restatement and the two reported rho -- a defect shows as a jump at one event, amplified rounding as a smooth growth.  Not a test.
usage: python tests/tools/gpu_proxqp_case_trace.py <case> <seed> [dense|sparse n me mi rho numItrConv feasible explicit]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import quadraticprogramsolver_amd as q
from oracle import proxqp_oracle_np as po
from test_gpu_proxqp import make_problem, make_sparse_problem, rel
c = int(sys.argv[1]); sparse = sys.argv[2] == "sparse"; n, me, mi = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rho, nic, feas, explicit, Kmax = float(sys.argv[6]), int(sys.argv[7]), sys.argv[8] == "1", sys.argv[9] == "1", int(sys.argv[10])
if sparse:
    P, qv, A, b, C, d = make_sparse_problem(n, me, mi, 500 + c, density=min(0.3, 8.0 / n), feasible=feas); Pd, Ad, Cd = P.toarray(), A.toarray(), C.toarray()
else:
    P, qv, A, b, C, d = make_problem(n, me, mi, 500 + c, feas); Pd, Ad, Cd = P, A, C
w = np.linalg.eigvalsh(Pd); print(f"case {c}: cond(P) = {w[-1] / w[0]:.2e}")
for K in sorted(set(list(range(nic, Kmax + 1, nic)))):
    ref = po.ProxQP.from_problem(Pd, qv, Ad, b, Cd, d); init = po.ProxQP.from_problem(Pd, qv, Ad, b, Cd, d)
    rr = po.SolveQuadraticProgramProxQP(ref, numIterations=K, ρ=rho, σ=1e-2, adptΡ=True, τ=10.0, numItrConv=nic)
    args = (P, qv, A, b, C, d) + ((init.vX, init.vY, init.vZ, init.vS) if explicit else ())
    with q.ProxQP(*args) as prob:
        rg = q.SolveQuadraticProgramProxQP(prob, numIterations=K, ρ=rho, σ=1e-2, adptΡ=True, τ=10.0, numItrConv=nic)
        dev = max(rel(prob.vX, ref.vX), rel(prob.vZ, ref.vZ) if mi else 0.0, rel(prob.vS, ref.vS) if mi else 0.0)
    w = np.linalg.eigvalsh(Pd + 1e-2 * np.eye(n) + rr["ρ"] * (Ad.T @ Ad + Cd.T @ Cd))
    print(f"K={K:4d} dev={dev:.2e} rho gpu={rg['ρ']:.10g} ref={rr['ρ']:.10g} rel={abs(rg['ρ'] - rr['ρ']) / rr['ρ']:.1e} rp={rr['PrimalResidual']:.2e} rd={rr['DualResidual']:.2e} "
          f"its {rg['Iterations']}/{rr['Iterations']} cond(K_rho)={w[-1] / w[0]:.1e}", flush=True)
