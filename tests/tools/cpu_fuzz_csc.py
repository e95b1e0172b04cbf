"""Randomised run of the CSC entry code of quadraticprogramsolver_amd/csrc/spmv_layout.cpp (through tests/capi/layout_shim.cpp) against scipy: a caller's CSC with unsorted
rows, duplicate entries, empty columns / rows and index base 0 or 1 -> canonical CSC -> CSR of the matrix and of its transpose; and the matrices of ItrSolCgInit
(LinearSystemSolvers.jl:112-114: mPI, mAA on one frozen pattern) against P + sigma I + rho A'A.  CPU only (not a test).  usage: python tests/tools/cpu_fuzz_csc.py [cases] [seed]"""
import os, sys, time, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ctypes as C
import numpy as np, scipy.sparse as sp
import test_layout_cpu as T
from test_layout_cpu import _ip32, _ip64, _dp

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
path = os.environ.get("QPS_HOST_TEST_LIB")
if not path:
    subprocess.check_call(["make", "-C", T.CSRC, "-s", "host-test"])
    path = os.path.join(root, "quadraticprogramsolver_amd", "libqps_host_test.so")
L = C.CDLL(path)
L.lt_csc_to_csr.restype = C.c_int64; L.lt_reduced_matrix.restype = C.c_int64
i64, p64, pd, p32 = C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)
L.lt_reduced_matrix.argtypes = [i64, i64, p64, p64, pd, p64, p64, pd, i64, p32, p32, pd, pd, pd, p64]
L.lt_csc_to_csr.argtypes = [i64, i64, p64, p64, pd, C.c_int, p32, p32, pd, p32, p32, pd]
bad = 0; t0 = time.time()
for c in range(cases):
    m = int(rng.choice([1, 2, 17, 300, 4000])); n = int(rng.choice([1, 3, 170, 2500])); base = int(rng.integers(0, 2))
    dens = float(rng.choice([0.0, 0.002, 0.03, 0.4]))
    A = sp.random(m, n, density=dens, random_state=rng, format="coo", dtype=np.float64)
    ndup = int(min(A.nnz, rng.choice([0, 1, 50])))
    rows = np.concatenate([A.row, A.row[:ndup]]).astype(np.int64); cols = np.concatenate([A.col, A.col[:ndup]]).astype(np.int64)
    vals = np.concatenate([A.data, rng.standard_normal(ndup)])
    order = rng.permutation(rows.size); rows, cols, vals = rows[order], cols[order], vals[order]
    key = np.argsort(cols, kind="stable"); rows, cols, vals = rows[key], cols[key], vals[key]       # grouped by column, rows unsorted inside a column
    cp = np.zeros(n + 1, dtype=np.int64); np.add.at(cp, cols + 1, 1); cp = np.cumsum(cp) + base
    ri = (rows + base).astype(np.int64); nz = np.ascontiguousarray(vals)
    tag = f"case {c}: {m}x{n} nnz={rows.size} dup={ndup} base={base}"
    msgs = []
    ref = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); ref.sum_duplicates(); ref.sort_indices()
    cap = max(1, rows.size)
    rp, ci, va = np.zeros(m + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    trp, tci, tva = np.zeros(n + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    rix, nzx = (ri, nz) if ri.size else (np.zeros(1, np.int64), np.zeros(1))
    got = L.lt_csc_to_csr(m, n, _ip64(cp), _ip64(rix), _dp(nzx), base, _ip32(rp), _ip32(ci), _dp(va), _ip32(trp), _ip32(tci), _dp(tva))
    # (a duplicate pair may sum to an explicit zero, which scipy keeps as well: counts compare as they are)
    if got != ref.nnz: msgs.append(f"nnz {got} vs {ref.nnz}")
    else:
        refT = sp.csr_matrix(ref.T); refT.sort_indices()
        if not (np.array_equal(rp, ref.indptr) and np.array_equal(ci[:got], ref.indices) and np.allclose(va[:got], ref.data, rtol=0, atol=1e-14)): msgs.append("CSR of the matrix differs")
        if not (np.array_equal(trp, refT.indptr) and np.array_equal(tci[:got], refT.indices) and np.allclose(tva[:got], refT.data, rtol=0, atol=1e-14)): msgs.append("CSR of the transpose differs")
    # reduced matrix on canonical inputs (square P of order n, A m x n)
    if n >= 2:
        Pm = sp.random(n, n, density=min(0.5, 2.0 / n), random_state=rng, format="csc"); P = sp.csc_matrix((Pm + Pm.T) * 0.5 + sp.diags(rng.random(n) + 0.5)); P.sort_indices()
        Ac = sp.csc_matrix(ref); Ac.sort_indices()
        AtA = (Ac.T @ Ac); capr = 4 * (P.nnz + AtA.nnz) + n + 8
        rrp, rci = np.zeros(n + 1, np.int32), np.zeros(capr, np.int32); vP, vAA, dg = np.zeros(capr), np.zeros(capr), np.zeros(capr); work = C.c_int64(0)
        Ari, Anz = (Ac.indices.astype(np.int64), Ac.data.copy()) if Ac.nnz else (np.zeros(1, np.int64), np.zeros(1))
        nnzL = L.lt_reduced_matrix(n, m, _ip64(P.indptr.astype(np.int64)), _ip64(P.indices.astype(np.int64)), _dp(P.data.copy()), _ip64(Ac.indptr.astype(np.int64)), _ip64(Ari), _dp(Anz),
                                   capr, _ip32(rrp), _ip32(rci), _dp(vP), _dp(vAA), _dp(dg), C.byref(work))
        if nnzL <= 0: msgs.append(f"reduced matrix: rc {nnzL}")
        else:
            rho, sigma = 0.37, 1e-3
            Lm = sp.csr_matrix((vP[:nnzL] + sigma * dg[:nnzL] + rho * vAA[:nnzL], rci[:nnzL], rrp), shape=(n, n))
            refL = (P + sigma * sp.identity(n) + rho * AtA)
            err = abs(Lm - refL).max() if (Lm - refL).nnz else 0.0
            if err > 1e-12 * max(1.0, abs(refL).max()): msgs.append(f"reduced matrix differs by {err:.2e}")
            if work.value != int((np.diff(sp.csr_matrix(Ac).indptr).astype(np.int64) ** 2).sum()): msgs.append("work estimate differs")
    if msgs: bad += 1; print(f"MISMATCH {tag}: " + "; ".join(msgs), flush=True)
    else: print(f"ok {tag}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
