"""CPU-only checks of the drop-in boundary: libqps_hip.so loads, exports every symbol include/qps.h declares, mirrors the
reference defaults, validates arguments like SolveQuadraticProgram.m:158-184, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "qps.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qps_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(qps):
    from quadraticprogramsolver_amd import _lib
    names = header_functions()
    assert names == sorted(_lib.EXPORTED_SYMBOLS)
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/qps.h but not exported"
    assert b"gfx950" in _lib.lib().qps_version()


def test_default_params_are_the_reference_defaults(qps):
    """SolveQuadraticProgram.jl:15-17 and LinearSystemSolvers.jl:125."""
    from quadraticprogramsolver_amd import _lib
    p = _lib.default_params()
    assert (p.numIterations, p.epsAbs, p.epsRel) == (5000, 1e-6, 1e-6)
    assert (p.rho, p.sigma, p.alpha, p.delta, p.adptRho) == (1.0, 1e-6, 1.6, 1e-6, 0)
    assert (p.fctrRho, p.numItrConv, p.numItrPolish, p.epsMinres, p.numItrMinres) == (5.0, 25, 10, 1e-6, 500)
    assert (p.epsPcg, p.numItrPcg) == (1e-6, 1000)
    assert C.sizeof(_lib.QpsParams) == 8 * 4 + 9 * 8 + 4 * 4
    assert C.sizeof(_lib.QpsInfo) == 4 * 4 + 7 * 8 + 2 * 4 + 8 + 4 * 4
    assert C.sizeof(_lib.QpsPolishReport) == 6 * 4 + 2 * 8


def test_enums_match_reference(qps):
    assert [int(f) for f in qps.ConvergenceFlag] == [1, 2, 3]          # SolveQuadraticProgram.jl:12
    assert [f.name for f in qps.ConvergenceFlag] == ["convNumItr", "convAdmm", "convPrimDual"]
    assert [f.name for f in qps.LinearSolverMode] == ["modeAuto", "modeItertaive", "modeDirect"]   # :11 (sic)
    assert [int(c) for c in qps.ProblemClass] == list(range(1, 10))    # GenerateQuadraticProgram.jl:6


def test_dimension_validation_mirrors_matlab_checks(qps):
    P, q, A, l, u = np.eye(4), np.zeros(4), np.ones((3, 4)), -np.ones(3), np.ones(3)
    with pytest.raises(ValueError):
        qps.QuadraticProgram(np.ones((4, 3)), q, A, l, u)
    with pytest.raises(ValueError):
        qps.QuadraticProgram(P, np.zeros(5), A, l, u)
    with pytest.raises(ValueError):
        qps.QuadraticProgram(P, q, np.ones((3, 5)), l, u)
    with pytest.raises(ValueError):
        qps.QuadraticProgram(P, q, A, np.ones(2), u)
    with pytest.raises(TypeError):   # an arbitrary CPU plugin pair is refused: there is no host loop in the product
        qps.SolveQuadraticProgramInplace(np.zeros(4), P, q, A, l, u, lambda *a: None, lambda *a: None)


def test_c_abi_status_codes_without_touching_the_gpu(qps):
    from quadraticprogramsolver_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    dp = C.POINTER(C.c_double)
    P = np.eye(3); q = np.zeros(3); A = np.ones((2, 3), order="F"); l = -np.ones(2); u = np.ones(2)
    g = lambda a: a.ctypes.data_as(dp)
    assert L.qps_create_dense(0, 2, g(P), 3, g(A), 2, g(q), g(l), g(u), 0, 0, C.byref(h)) == 2    # BAD_DIMENSION
    assert L.qps_create_dense(3, 2, g(P), 2, g(A), 2, g(q), g(l), g(u), 0, 0, C.byref(h)) == 2    # ldp < n
    assert L.qps_create_dense(3, 2, None, 3, g(A), 2, g(q), g(l), g(u), 0, 0, C.byref(h)) == 1    # BAD_ARGUMENT
    assert L.qps_create_dense(3, 2, g(P), 3, g(A), 2, g(q), g(l), g(u), 7, 0, C.byref(h)) == 1    # unknown dtype
    assert b"dtype" in L.qps_last_error(None)
    assert L.qps_solve(None, g(q), None, None) == 1
    assert L.qps_destroy(None) == 0


def test_asymmetric_p_is_refused_like_issymmetric(qps):
    """SolveQuadraticProgram.m:166-168: `if(~issymmetric(mP)) error(...)`, tolerance 0.  Checked before the device is
    touched, for the dense, the CSC and the batch constructors."""
    import scipy.sparse as sp
    from quadraticprogramsolver_amd import _lib
    n, m = 70, 5
    rng = np.random.default_rng(0)
    M = rng.standard_normal((n, n)); P = M.T @ M + np.eye(n); P = 0.5 * (P + P.T)
    q, A, l, u = np.zeros(n), rng.standard_normal((m, n)), -np.ones(m), np.ones(m)
    Pbad = P.copy(); Pbad[66, 3] = np.nextafter(Pbad[66, 3], np.inf)       # one ulp off, far from the diagonal
    for bad in (Pbad, sp.csc_matrix(Pbad)):
        with pytest.raises(qps.QpsError) as e:
            qps.QuadraticProgram(bad, q, A, l, u)
        assert e.value.status == 1 and "symmetric" in str(e.value)
    with pytest.raises(qps.QpsError) as e:
        qps.QuadraticProgramBatch([(P, q, A, l, u), (Pbad, q, A, l, u)])
    assert e.value.status == 1 and "QP 1" in str(e.value)
    # structurally asymmetric sparse pattern (an entry without its mirror image)
    S = sp.lil_matrix(sp.eye(n)); S[2, 40] = 0.5
    with pytest.raises(qps.QpsError) as e:
        qps.QuadraticProgram(S.tocsc(), q, sp.csc_matrix(A), l, u, linsys="cg")
    assert e.value.status == 1
    # symmetric inputs pass the check and reach the device test (status 7 without a GPU); unsorted CSC columns are accepted
    if _lib.lib().qps_device_count() == 0:
        for good in (P, sp.csc_matrix(P)):
            with pytest.raises(qps.QpsError) as e:
                qps.QuadraticProgram(good, q, A, l, u)
            assert e.value.status == 7


def test_mode_auto_rule_is_the_reference_rule(qps):
    """SolveQuadraticProgram.jl:129-130, :143-151 (SolveQuadraticProgram.m:190-199): direct iff rows(P) + rows(A) <= 5000 and
    (nnz(P) + nnz(A)) / rows^2 <= 0.4 -- evaluated by the library, no device needed."""
    from quadraticprogramsolver_amd import _lib
    L = _lib.lib()
    CG, CHOL, LDL = _lib.QPS_LINSYS_CG, _lib.QPS_LINSYS_CHOLESKY, _lib.QPS_LINSYS_KKT_LDL
    assert L.qps_linsys_auto(1000, 500, 150000, 75000, 1) == LDL            # 1500 rows, density 0.1
    assert L.qps_linsys_auto(1000, 500, 150000, 75000, 0) == CHOL           # same problem held as dense arrays
    assert L.qps_linsys_auto(3000, 2000, 10, 10, 1) == LDL                  # exactly 5000 rows: still direct (<=)
    assert L.qps_linsys_auto(3000, 2001, 10, 10, 1) == CG                   # 5001 rows
    assert L.qps_linsys_auto(100, 100, 16000, 0, 1) == LDL                  # density exactly 0.4
    assert L.qps_linsys_auto(100, 100, 16001, 0, 1) == CG
    assert L.qps_linsys_auto(4096, 8192, 4096 * 4096, 4096 * 8192, 0) == CG  # BASELINE's dense config by the literal rule
    import scipy.sparse as sp
    assert qps.AutoLinearSolverMode(sp.eye(100, format="csc"), sp.eye(100, format="csc")) == qps.LinearSolverMode.modeDirect
    assert qps.AutoLinearSolverMode(np.ones((100, 100)), np.ones((50, 100))) == qps.LinearSolverMode.modeItertaive


def _analyze(P, A, base=0):
    import scipy.sparse as sp
    from quadraticprogramsolver_amd import _lib
    Pc, Ac = sp.csc_matrix(P), sp.csc_matrix(A)
    Pc.sum_duplicates(); Ac.sum_duplicates()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    arrs = [np.ascontiguousarray(a.astype(np.int64) + base) for a in (Pc.indptr, Pc.indices, Ac.indptr, Ac.indices)]
    perm = np.zeros(P.shape[0] + A.shape[0], dtype=np.int64)
    rep = _lib.QpsLdlReport()
    _lib.check(_lib.lib().qps_ldl_analyze(P.shape[0], A.shape[0], ip(arrs[0]), ip(arrs[1]), ip(arrs[2]), ip(arrs[3]), base, ip(perm), C.byref(rep)))
    return perm - base, rep.as_dict()


def _symbolic_nnz(K, perm):
    """Independent count of nnz(L) (strictly lower) for the ordering `perm`: boolean elimination on the permuted pattern."""
    N = K.shape[0]
    B = (K[perm][:, perm].toarray() != 0)
    B = B | B.T
    cnt = 0
    for k in range(N):
        rows = np.nonzero(B[k + 1:, k])[0] + k + 1
        cnt += rows.size
        if rows.size:
            B[np.ix_(rows, rows)] = True
    return cnt


@pytest.mark.parametrize("pc,n", [("randomQp", 30), ("equalityConstrainedQp", 40), ("portfolioOptimization", 100), ("lassoOptimization", 6),
                                  ("huberFitting", 4), ("supportVectorMachine", 6), ("isotonicRegression", 60)])
def test_ldl_symbolic_analysis(qps, pc, n):
    """The host half of QPS_LINSYS_KKT_LDL (ordering, elimination tree, symbolic factor) runs without a device: the ordering is a
    permutation, the predicted nnz(L) equals an independent boolean elimination under that ordering, the minimum-degree ordering
    beats the natural one, the level / tail split covers every column, and the 1-based (Julia) entry gives the same answer."""
    import scipy.sparse as sp
    P, q, A, l, u = qps.GenerateRandomQP(getattr(qps.ProblemClass, pc), n, rng=qps.make_rng(31, n))
    nn, m = P.shape[0], A.shape[0]
    perm, rep = _analyze(P, A)
    assert sorted(perm.tolist()) == list(range(nn + m))
    K = sp.bmat([[sp.csc_matrix(P) + sp.eye(nn), sp.csc_matrix(A).T], [sp.csc_matrix(A), -sp.eye(m)]], format="csr")
    assert rep["nnzL"] == _symbolic_nnz(K, perm)
    assert rep["nnzL"] <= _symbolic_nnz(K, np.arange(nn + m))
    assert rep["numRows"] == nn + m == rep["numSparseColumns"] + rep["tailSize"]
    assert rep["nnzK"] == sp.tril(K, -1).nnz and rep["nnzStored"] >= rep["nnzL"] >= rep["nnzK"]
    perm1, rep1 = _analyze(P, A, base=1)
    assert np.array_equal(perm, perm1) and rep == rep1


def _banded_qp(n, seed=0):
    """Chain-structured QP (smoothing / trend-filtering shape): tridiagonal SPD P, first-difference constraints -- a banded KKT matrix."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    P = sp.diags([-np.ones(n - 1), 2.5 + rng.random(n), -np.ones(n - 1)], [-1, 0, 1], format="csc")
    A = sp.diags([-np.ones(n - 1), np.ones(n - 1)], [0, 1], shape=(n - 1, n), format="csc")
    return P, rng.standard_normal(n), A, -0.1 * np.ones(n - 1), 0.1 * np.ones(n - 1)


def test_ldl_analysis_of_chain_like_problems_falls_back_to_line_dissection(qps, monkeypatch):
    """A banded KKT matrix has a minimum-degree elimination tree about as deep as the matrix is long.  While the chain fits the dense tail
    (<= 8192 columns) it is simply factorised densely; beyond that every level would be a dependent launch in each triangular sweep, and the
    analysis dissects the breadth-first line order instead: still a permutation, a factor that matches an independent boolean elimination, a
    few hundred levels where minimum degree gave thousands; same through the 1-based entry.  (The small instance runs with a 512-column tail
    limit so that the fallback is reached at a size the boolean elimination can check.)"""
    import scipy.sparse as sp
    monkeypatch.setenv("QPS_LDL_MAX_TAIL", "512")
    P, q, A, l, u = _banded_qp(3000)
    nn, m = P.shape[0], A.shape[0]
    perm, rep = _analyze(P, A)
    assert sorted(perm.tolist()) == list(range(nn + m))
    K = sp.bmat([[sp.csc_matrix(P) + sp.eye(nn), sp.csc_matrix(A).T], [sp.csc_matrix(A), -sp.eye(m)]], format="csr")
    assert rep["nnzL"] == _symbolic_nnz(K, perm)
    assert rep["numRows"] == nn + m == rep["numSparseColumns"] + rep["tailSize"]
    assert rep["treeHeight"] <= 400 and rep["numSparseLevels"] <= 400
    perm1, rep1 = _analyze(P, A, base=1)
    assert np.array_equal(perm, perm1) and rep == rep1
    # a large instance analyses with the default limits too (minimum degree alone: refused, "elimination tree too deep")
    monkeypatch.delenv("QPS_LDL_MAX_TAIL")
    P, q, A, l, u = _banded_qp(100000, 1)
    perm, rep = _analyze(P, A)
    assert rep["numRows"] == 199999 and rep["numSparseLevels"] <= 600 and rep["nnzL"] <= 12 * rep["nnzK"]


def test_ldl_analysis_randomised_patterns_with_forced_line_dissection(qps, monkeypatch):
    """Random band / sparse / block-diagonal / path patterns (disconnected graphs and m = 0 included) with limits that push the analysis through its
    line-dissection fallback: always a permutation, and nnz(L) equals an independent boolean elimination under that ordering."""
    import scipy.sparse as sp
    for k, v in {"QPS_LDL_MAX_TAIL": "32", "QPS_LDL_MIN_LEVEL": "4", "QPS_LDL_DISSECT_LEVELS": "8", "QPS_LDL_MAX_LEVELS": "100000"}.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(3)
    for c in range(32):
        n = int(rng.integers(20, 200)); m = int(rng.integers(0, 2 * n))
        kind = ["band", "rand", "blocks", "path2"][c % 4]
        if kind == "band":
            bw = int(rng.integers(1, 5)); m = min(m, n - 1)
            P = sp.diags([np.ones(n - k) for k in range(bw + 1)] + [np.ones(n - k) for k in range(1, bw + 1)],
                         list(range(bw + 1)) + [-k for k in range(1, bw + 1)], format="csc")
            A = sp.diags([np.ones(m), np.ones(m)], [0, 1], shape=(m, n), format="csc") if m > 0 else sp.csc_matrix((0, n))
        elif kind == "rand":
            M = sp.random(n, n, density=2.0 / n, random_state=rng, format="csc"); P = (M + M.T + sp.identity(n)).tocsc()
            A = sp.random(m, n, density=1.5 / n, random_state=rng, format="csc")
        elif kind == "blocks":
            b = max(2, n // int(rng.integers(2, 9)))
            P = sp.block_diag([np.ones((min(b, n - i), min(b, n - i))) for i in range(0, n, b)], format="csc")
            A = sp.random(m, n, density=0.5 / n, random_state=rng, format="csc")
        else:
            P = sp.identity(n, format="csc"); rows = np.arange(m) % max(n - 2, 1)
            A = sp.csc_matrix((np.ones(2 * m), (np.r_[np.arange(m), np.arange(m)], np.r_[rows, rows + 2])), shape=(m, n)) if m > 0 else sp.csc_matrix((0, n))
        perm, rep = _analyze(P, A)
        N = n + m
        assert sorted(perm.tolist()) == list(range(N)), (c, kind)
        K = sp.bmat([[sp.csc_matrix(P) + sp.eye(n), sp.csc_matrix(A).T], [sp.csc_matrix(A), -sp.eye(m)]], format="csr") if m > 0 else (sp.csc_matrix(P) + sp.eye(n)).tocsr()
        assert rep["nnzL"] == _symbolic_nnz(K, perm) and rep["numRows"] == N == rep["numSparseColumns"] + rep["tailSize"], (c, kind, rep)


def test_fails_loudly_without_a_gpu(qps):
    from quadraticprogramsolver_amd import _lib
    if _lib.lib().qps_device_count() > 0:
        pytest.skip("a GPU is visible")
    P, q, A, l, u = np.eye(4), np.zeros(4), np.ones((3, 4)), -np.ones(3), np.ones(3)
    with pytest.raises(qps.QpsError) as e:
        qps.SolveQuadraticProgram(P, q, A, l, u)
    assert e.value.status == 7
    for pair in ((qps.HipCholInit, qps.HipChol), (qps.HipCgInit, qps.HipCg), (qps.HipLdlInit, qps.HipLdl)):
        with pytest.raises(qps.QpsError) as e:
            qps.SolveQuadraticProgramInplace(np.zeros(4), P, q, A, l, u, *pair)
        assert e.value.status == 7 and "no CPU fallback" in str(e.value)


def test_proxqp_defaults_are_the_reference_defaults(qps):
    """ProxQP.jl:118 keyword defaults."""
    from quadraticprogramsolver_amd import _lib
    p = _lib.QpsProxQpParams()
    _lib.check(_lib.lib().qps_proxqp_default_params(C.byref(p)))
    assert (p.numIterations, p.epsAbs, p.epsRel, p.numItrConv, p.rho, p.sigma, p.adptRho, p.tau) == (2000, 1e-7, 1e-6, 50, 1e2, 1e-2, 1, 10.0)


def test_sparse_proxqp_constructor_validates_before_it_needs_a_device(qps):
    """qps_proxqp_create_csc (SparseProxQP, ProxQP.jl:71, :95-115) checks the CSC fields on the host: a malformed colptr, a row index out of
    range or a NaN is refused with its own status whether or not a GPU is visible; well-formed input without a GPU fails loudly with NO_DEVICE."""
    import scipy.sparse as sp
    from quadraticprogramsolver_amd import _lib
    n, me, mi = 6, 2, 3
    P = sp.identity(n, format="csc"); A = sp.csc_matrix(np.ones((me, n))); Cm = sp.csc_matrix(np.ones((mi, n)))
    q, b, d = np.zeros(n), np.zeros(me), np.ones(mi)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

    def create(Pm=P, Am=A, Cmm=Cm, edit=None, base=0):
        arrs = []
        for M in (Pm, Am, Cmm):
            arrs.append([M.indptr.astype(np.int64) + base, M.indices.astype(np.int64) + base, M.data.astype(np.float64).copy()])
        if edit:
            edit(arrs)
        h = C.c_void_p()
        (Pc, Pr, Pv), (Ac, Ar, Av), (Cc, Cr, Cv) = arrs
        return _lib.lib().qps_proxqp_create_csc(n, me, mi, ip(Pc), ip(Pr), dp(Pv), dp(q), ip(Ac), ip(Ar), dp(Av), dp(b), ip(Cc), ip(Cr), dp(Cv), dp(d), base, 0, 0, C.byref(h)), h

    def bad_colptr(a): a[1][0][2] = a[1][0][1] - 1
    def bad_row(a): a[2][1][0] = mi + 5
    def nan_val(a): a[0][2][0] = np.nan
    def bad_start(a): a[0][0][0] = 1
    assert create(edit=bad_colptr)[0] == 1      # QPS_ERR_BAD_ARGUMENT
    assert create(edit=bad_row)[0] == 2         # QPS_ERR_BAD_DIMENSION
    assert create(edit=nan_val)[0] == 3         # QPS_ERR_NOT_FINITE
    assert create(edit=bad_start)[0] == 1
    rc, h = create(base=1)                      # Julia's 1-based fields
    if _lib.lib().qps_device_count() > 0:
        assert rc == 0
        _lib.lib().qps_destroy(h)
    else:
        assert rc == 7                          # QPS_ERR_NO_DEVICE: there is no CPU fallback


def test_no_null_stream_fills_or_uploads_in_the_library():
    """Stream-ordering rule (qps_internal.h): handles work on non-blocking streams, so nothing their kernels read may be written
    through the null stream.  Synchronous hipMemset / hipMemcpy, waits on the null stream and device-wide synchronisations that
    would paper over such a write are banned from the sources altogether."""
    csrc = os.path.join(ROOT, "quadraticprogramsolver_amd", "csrc")
    banned = re.compile(r"\bhipMemset\s*\(|\bhipMemcpy\s*\(|hipStreamSynchronize\s*\(\s*(nullptr|0|NULL)\s*\)|\bhipDeviceSynchronize\s*\(|\bhipMemcpy2D\s*\(")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            code = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f), errors="ignore").read())
            hit = banned.search(code)
            assert hit is None, f"{f}: {hit.group(0)}"


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    pkg = os.path.join(ROOT, "quadraticprogramsolver_amd")
    for dp_, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dp_, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("test infrastructure", ""), f"{f} mentions the oracle"


def test_no_function_local_static_guards_per_device_hip_state():
    """include/qps.h:19 promises handles on distinct devices of one process.  A function attribute (hipFuncSetAttribute) is a setting of ONE
    device's code object and device properties describe ONE device: neither may hide behind a plain function-local `static` that remembers the
    first device the process touched.  The library keys them by device ordinal (PerDeviceOnce, device_cu_count, launch_is_co_resident)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "quadraticprogramsolver_amd", "csrc")
    offenders = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))):
        text = open(path).read()
        text_nc = re.sub(r"//[^\n]*", "", text)
        lines = text_nc.split("\n")
        for i, line in enumerate(lines):
            if "hipFuncSetAttribute" in line:
                ctx = " ".join(lines[max(0, i - 3):i + 1])
                if ".once(" not in ctx and "launch_is_co_resident" not in " ".join(lines[max(0, i - 12):i + 1]):
                    offenders.append(f"{os.path.basename(path)}:{i + 1}: hipFuncSetAttribute not behind a per-device once-flag")
            if re.search(r"\bstatic\b(?!\s+(constexpr|inline|__device__|__global__|PerDeviceOnce))", line) and re.search(
                    r"hipGetDeviceProperties|hipDeviceGetAttribute|hipOccupancyMaxActiveBlocks|multiProcessorCount|hipGetDevice\(", " ".join(lines[i:i + 3])):
                offenders.append(f"{os.path.basename(path)}:{i + 1}: device query cached in a function-local static")
            if re.search(r"\bstatic\s+bool\s+attr", line):
                offenders.append(f"{os.path.basename(path)}:{i + 1}: static bool attr* flag")
    assert not offenders, offenders


# ---------------------------------------------------------------------------------------------------------------------
# The Julia binding cannot run in this pipeline (no Julia): guard it against drifting away from the C ABI by reading it.
# _lib.py's Structures / argtypes are tied to include/qps.h by the tests above; here the .jl file is tied to _lib.py.
# ---------------------------------------------------------------------------------------------------------------------
JL = os.path.join(ROOT, "julia", "QuadraticProgramSolverHIP.jl")
JL_SCALAR = {"Int32": C.c_int32, "Int64": C.c_int64, "Float64": C.c_double}


def _jl_source():
    return re.sub(r"#[^\n]*", "", open(JL, encoding="utf-8").read())


def _jl_structs(src):
    out = {}
    for mm in re.finditer(r"(?:mutable\s+)?struct\s+(\w+)\s*\n(.*?)\nend", src, re.S):
        fields = re.findall(r"(\w+)\s*::\s*(\w+)", mm.group(2))
        out[mm.group(1)] = fields
    return out


def test_julia_structs_match_the_ctypes_structures():
    from quadraticprogramsolver_amd import _lib
    structs = _jl_structs(_jl_source())
    pairs = {"QpsParams": _lib.QpsParams, "QpsInfo": _lib.QpsInfo, "QpsProxQpParams": _lib.QpsProxQpParams, "QpsProxQpReport": _lib.QpsProxQpReport,
             "QpsPolishReport": _lib.QpsPolishReport}
    for name, ct in pairs.items():
        assert name in structs, name
        jl = [(f, JL_SCALAR[t]) for f, t in structs[name]]
        assert jl == [(f, t) for f, t in ct._fields_], (name, jl, ct._fields_)


def _jl_ccalls(src):
    """(symbol, return type, [argument types]) of every ccall((:sym, LIBQPS), Ret, (types...), args...)."""
    calls = []
    for mm in re.finditer(r"ccall\(\(:(\w+),\s*LIBQPS\),\s*(\w+),\s*\(", src):
        i, depth, start = mm.end(), 1, mm.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        body = src[start:i - 1]
        types, cur, d = [], "", 0
        for ch in body:
            if ch == "{":
                d += 1
            elif ch == "}":
                d -= 1
            if ch == "," and d == 0:
                types.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            types.append(cur.strip())
        calls.append((mm.group(1), mm.group(2), types))
    return calls


def test_julia_ccall_signatures_match_the_ctypes_argtypes():
    from quadraticprogramsolver_amd import _lib
    L = _lib.lib()
    structs = {"QpsParams": _lib.QpsParams, "QpsInfo": _lib.QpsInfo, "QpsProxQpParams": _lib.QpsProxQpParams, "QpsProxQpReport": _lib.QpsProxQpReport,
               "QpsPolishReport": _lib.QpsPolishReport}

    def ctype_of(t, sym, pos):
        if t in JL_SCALAR:
            return JL_SCALAR[t]
        if t == "Ptr{Cvoid}":
            return C.c_void_p
        if t == "Ref{Ptr{Cvoid}}":
            return C.POINTER(C.c_void_p)
        mm = re.fullmatch(r"(?:Ptr|Ref)\{(\w+)\}", t)
        assert mm, (sym, pos, t)
        inner = mm.group(1)
        if inner == "UInt8":                       # the batch's array of qps_info records is handed over as raw bytes
            return getattr(L, sym).argtypes[pos]
        return C.POINTER(structs[inner] if inner in structs else JL_SCALAR[inner])

    calls = _jl_ccalls(_jl_source())
    seen = set()
    for sym, ret, types in calls:
        fn = getattr(L, sym)
        want = list(fn.argtypes)
        got = [ctype_of(t, sym, k) for k, t in enumerate(types)]
        assert got == want, (sym, types, want)
        assert (ret == "Cstring") == (fn.restype is C.c_char_p) and (ret == "Int32") == (fn.restype is C.c_int32), (sym, ret)
        seen.add(sym)
    # every entry point a Julia user needs is bound (host-only analysis / profiling helpers are not part of the wrapper)
    need = {"qps_create_dense", "qps_create_csc", "qps_solve", "qps_linsys_init", "qps_linsys_solve", "qps_linsys_set_cg", "qps_linsys_auto", "qps_destroy",
            "qps_last_error", "qps_create_dense_batch", "qps_solve_batch", "qps_polish", "qps_proxqp_create_dense", "qps_proxqp_create_csc",
            "qps_proxqp_init_kkt", "qps_proxqp_get_state", "qps_proxqp_solve"}
    assert need <= seen, need - seen


def test_julia_binding_covers_the_reference_interface():
    """The names a user of the reference types: three plugin pairs callable with the literal plugin signature (LinearSystemSolvers.jl:16,28,145,164 --
    the CG pair with its `ϵPcg, numItrPcg` kwargs), the modeAuto rule in the convenience form (SolveQuadraticProgram.jl:143-151), ProxQP{T} (ProxQP.jl:8)."""
    src = _jl_source()
    for pat in (r"function \(::HipCholInitT\)\(vX, mP, vQ, mA, ρ, ρ¹, σ, numElements, numConstraints\)",
                r"function \(::HipCgInitT\)\(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, ρ, ρ¹, σ, numElements, numConstraints\)",
                r"function \(::HipLdlInitT\)\(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, ρ, ρ¹, σ, numElements, numConstraints\)",
                r"function \(::HipCgT\)\(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ; ϵPcg = 1e-6, numItrPcg = 1000\)",
                r"function \(::HipCholT\)\(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ\)",
                r"linSolverMode::LinearSolverMode = modeAuto", r"modeItertaive", r"AutoLinearSystemPair\(mP, mA\)",
                r"ProxQPHip\{T <: AbstractFloat\}", r"where \{T <: AbstractFloat\}", r"_dtype\(::Type\{Float32\}\) = Int32\(1\)",
                r"ϵPcg = 1e-6, numItrPcg = 1000", r"dtype::Type = Float64, trsvBlock = 0"):
        assert re.search(pat, src), pat
    # the kwargs of SolveQuadraticProgram.jl:15-17, names and defaults
    for kw in ("numIterations = 5000", "ϵAbs = 1e-6", "ϵRel = 1e-6", "ρ = 1", "σ = 1e-6", "α = 1.6", "δ = 1e-6", "adptΡ::Bool = false", "fctrΡ = 5",
               "numItrConv = 25", "numItrPolish = 10", "ϵMinres = 1e-6", "numItrMinres = 500"):
        assert kw in src, kw
