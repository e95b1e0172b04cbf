"""Randomised check of the symbolic analysis of the sparse direct KKT plugin (quadraticprogramsolver_amd/csrc/ldl_symbolic.cpp, through tests/capi/layout_shim.cpp) against an
independent count: for the permutation the analysis returns, the strictly-lower non-zeros of L by Liu's row-subtree walk on the permuted KKT pattern
[P + sigma I, A'; A, -I / rho] (LinearSystemSolvers.jl:18), written here from the textbook and sharing no code with the product.  CPU only (not a test).
usage: python tests/tools/cpu_fuzz_ldl_symbolic.py [cases] [seed]"""
import os, sys, time, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ctypes as C
import numpy as np, scipy.sparse as sp
import test_layout_cpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
path = os.environ.get("QPS_HOST_TEST_LIB")
if not path:
    subprocess.check_call(["make", "-C", T.CSRC, "-s", "host-test"])
    path = os.path.join(root, "quadraticprogramsolver_amd", "libqps_host_test.so")
shim = C.CDLL(path)


def fill_of(K, perm):
    """(strictly-lower nnz of L, elimination-tree height) of the Cholesky pattern of K[perm][:, perm] (K symmetric pattern with a full diagonal)."""
    N = K.shape[0]
    B = sp.csr_matrix(K[perm][:, perm]); B.sort_indices()
    parent = np.full(N, -1); mark = np.full(N, -1); count = 0
    for i in range(N):
        mark[i] = i
        for j in B.indices[B.indptr[i]:B.indptr[i + 1]]:
            if j >= i: break
            while mark[j] != i:                                             # walk up from j until a node already reached from row i
                if parent[j] < 0: parent[j] = i
                mark[j] = i; count += 1
                j = parent[j]
    depth = np.zeros(N, dtype=np.int64)
    for j in range(N - 1, -1, -1):
        if parent[j] >= 0: depth[j] = depth[parent[j]] + 1
    return count, int(depth.max()) + 1


bad = 0; t0 = time.time()
for c in range(cases):
    n = int(rng.choice([5, 40, 200, 700, 1500])); m = int(rng.choice([0, 1, n // 3, n, 2 * n]))
    kind = rng.choice(["random", "banded", "diagP", "arrow"])
    if kind == "random":
        Mx = sp.random(n, n, density=min(0.5, 3.0 / n), random_state=rng, format="csc"); P = (Mx.T @ Mx + sp.identity(n)).tocsc()
        A = sp.random(m, n, density=min(0.5, 4.0 / n), random_state=rng, format="csc")
    elif kind == "banded":
        P = sp.diags([np.ones(n - abs(k)) for k in (-2, -1, 0, 1, 2) if n - abs(k) > 0], [k for k in (-2, -1, 0, 1, 2) if n - abs(k) > 0], format="csc")
        A = sp.diags([np.ones(min(m, n)), np.ones(max(0, min(m, n - 3)))], [0, 3], shape=(m, n), format="csc") if m else sp.csc_matrix((0, n))
    elif kind == "diagP":
        P = sp.identity(n, format="csc"); A = sp.random(m, n, density=min(0.5, 6.0 / n), random_state=rng, format="csc")
    else:                                                                   # a dense row / column in P and in A
        P = sp.lil_matrix(sp.identity(n)); P[0, :] = 1.0; P[:, 0] = 1.0; P = P.tocsc()
        A = sp.lil_matrix(sp.random(m, n, density=min(0.5, 2.0 / n), random_state=rng)); 
        if m: A[m - 1, :] = 1.0
        A = A.tocsc()
    P.sort_indices(); A.sort_indices()
    max_tail = int(rng.choice([0, 64, 8192])); min_level = int(rng.choice([1, 64])); max_levels = 1 << 20
    N = n + m
    perm = np.zeros(N, dtype=np.int64); rep = np.zeros(8, dtype=np.int64)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    Pcp, Pri, Acp, Ari = P.indptr.astype(np.int64), P.indices.astype(np.int64), A.indptr.astype(np.int64), A.indices.astype(np.int64)
    if Pri.size == 0: Pri = np.zeros(1, np.int64)
    if Ari.size == 0: Ari = np.zeros(1, np.int64)
    rc = shim.lt_ldl_analyze(n, m, ip(Pcp), ip(Pri), ip(Acp), ip(Ari), 0, max_tail, min_level, max_levels, ip(perm), ip(rep))
    tag = f"case {c}: {kind} n={n} m={m} nnz(P)={P.nnz} nnz(A)={A.nnz} max_tail={max_tail} min_level={min_level}"
    if rc != 0:
        bad += 1; print(f"ERROR {tag}: analysis raised (rc {rc})", flush=True); continue
    msgs = []
    if sorted(perm.tolist()) != list(range(N)): msgs.append("perm is not a permutation")
    else:
        Ab = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=(n, m)).T if m else sp.csr_matrix((0, n))   # pattern of A (m x n) from its CSC arrays
        Pb = sp.csr_matrix(abs(P)); Pb.data[:] = 1.0
        K = sp.bmat([[Pb + sp.identity(n), Ab.T], [Ab, sp.identity(m)]], format="csr") if m else sp.csr_matrix(Pb + sp.identity(n))
        K.data[:] = 1.0
        cnt, height = fill_of(K, perm)
        nnzK_lower = (sp.tril(K, -1)).nnz
        if rep[0] != N: msgs.append(f"N {rep[0]} != {N}")
        if rep[1] + rep[2] != N: msgs.append(f"Ns + Nt = {rep[1] + rep[2]} != {N}")
        if rep[6] not in (cnt, cnt + N): msgs.append(f"nnzL_exact {rep[6]} vs independent {cnt} (strictly lower; + N = {cnt + N})")
        if rep[5] not in (nnzK_lower, nnzK_lower + N): msgs.append(f"nnzK {rep[5]} vs {nnzK_lower} (+ N = {nnzK_lower + N})")
        if rep[7] < rep[6]: msgs.append(f"nnzL with the dense tail {rep[7]} < exact {rep[6]}")
    if msgs:
        bad += 1; print(f"MISMATCH {tag}: " + "; ".join(msgs) + f" report {rep.tolist()}", flush=True)
    else:
        print(f"ok {tag}: nnz(L) {rep[6]} levels {rep[3]} Ns {rep[1]} Nt {rep[2]}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
