"""Ad-hoc timing of the sparse ProxQP solver against the densified one (not a test).  usage: python tests/tools/gpu_proxqp_sparse_timing.py [n] [me] [mi] [density]
QPS_PROXQP_SPARSE=0 in the environment selects the densified path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import quadraticprogramsolver_amd as q
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
me = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
dens = float(sys.argv[4]) if len(sys.argv) > 4 else 0.004
rng = np.random.default_rng(3)
if dens > 0:
    M = sp.random(n, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    A = (sp.random(me, n, density=dens, random_state=rng, data_rvs=rng.standard_normal) + sp.eye(me, n)).tocsc()
    C = sp.random(mi, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
elif dens < 0:   # separable structure: 16 x 16 SPD blocks of P, one equality per block (sum = 1), x >= 0: a wide, shallow elimination tree at any size
    nb = n // 16; n = nb * 16; me = nb; mi = n
    M = sp.block_diag([rng.standard_normal((16, 16)) for _ in range(nb)], format="csc")
    A = sp.kron(sp.identity(nb), np.ones((1, 16)), format="csc")
    C = (-sp.identity(n)).tocsc()
else:   # banded structure (bandwidth 5): the factor stays sparse at any size
    M = sp.diags([rng.standard_normal(n - k) for k in range(3)], [0, 1, 2], format="csc")
    A = sp.diags([np.ones(me), rng.standard_normal(me)], [0, 3], shape=(me, n), format="csc")
    C = sp.diags([rng.standard_normal(mi), rng.standard_normal(mi), rng.standard_normal(mi)], [0, 1, 4], shape=(mi, n), format="csc") if mi <= n else \
        sp.vstack([sp.diags([rng.standard_normal(n), rng.standard_normal(n - 1)], [0, 1], format="csc")] * (mi // n)).tocsc()
    mi = C.shape[0]
P = (M.T @ M + 0.01 * sp.identity(n)).tocsc(); P = (0.5 * (P + P.T)).tocsc()
qv = rng.standard_normal(n); x0 = np.abs(rng.standard_normal(n)) if dens < 0 else rng.standard_normal(n); b = A @ x0; d = C @ x0 + 0.3
t0 = time.perf_counter()
prob = q.ProxQP(P, qv, A, b, C, d)
t1 = time.perf_counter()
rep = q.SolveQuadraticProgramProxQP(prob, numIterations=50)
t2 = time.perf_counter()
K = 2000
rep = q.SolveQuadraticProgramProxQP(prob, numIterations=K, adptΡ=False)
t3 = time.perf_counter()
print(f"n={n} me={me} mi={mi} nnz(P)={P.nnz} nnz(A)+nnz(C)={A.nnz + C.nnz} sparse={os.environ.get('QPS_PROXQP_SPARSE', '1')}: create+init {1e3*(t1-t0):.1f} ms, "
      f"first solve (50 its, incl. analysis) {1e3*(t2-t1):.1f} ms, {K} iterations {1e3*(t3-t2):.1f} ms = {1e6*(t3-t2)/K:.1f} us/iteration; report {rep}", flush=True)
