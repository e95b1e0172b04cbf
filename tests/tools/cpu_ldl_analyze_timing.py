"""Ad-hoc: host-side cost of the symbolic analysis of the sparse KKT plugin (qps_ldl_analyze: ordering + elimination tree + pattern), CPU only."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from quadraticprogramsolver_amd import _lib
from quadraticprogramsolver_amd.generator import GenerateRandomQP, ProblemClass, make_rng
lib = _lib.lib() if hasattr(_lib, "lib") else _lib.load()
P64 = ctypes.POINTER(ctypes.c_int64)
for pc in (ProblemClass.lassoOptimization, ProblemClass.huberFitting, ProblemClass.supportVectorMachine, ProblemClass.portfolioOptimization):
    P, qv, A, l, u = GenerateRandomQP(pc, 100, rng=make_rng(4321, 6001))
    Pc = sp.csc_matrix(P); Ac = sp.csc_matrix(A); n = P.shape[0]; m = A.shape[0]
    arrs = [a.astype(np.int64) for a in (Pc.indptr, Pc.indices, Ac.indptr, Ac.indices)]
    perm = np.zeros(n + m, dtype=np.int64); rep = _lib.QpsLdlReport()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        rc = lib.qps_ldl_analyze(n, m, *[a.ctypes.data_as(P64) for a in arrs], 0, perm.ctypes.data_as(P64), ctypes.byref(rep))
        ts.append(time.perf_counter() - t)
    print(f"{pc.name:24s} rc {rc} N {n + m:6d} nnz(L) {rep.nnzL:8d} analyze ms {[round(x * 1e3, 1) for x in ts]}")
