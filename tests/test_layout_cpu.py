"""CPU tests of the product's host-only code (round-3 review item 5): the SpMV layout builders of quadraticprogramsolver_amd/csrc/spmv_layout.cpp -- sliced form
(k_spmv_sell), task form (k_spmv_blk), CSR-stream row blocks, CSC canonicalisation / validation / CSC -> CSR -- and the symbolic analysis of ldl_symbolic.cpp, built
with g++ behind tests/capi/layout_shim.cpp (no device, no HIP).  Every layout is expanded back into y = M x by a host interpreter that walks the arrays the way the
kernel does (partial sums per column block, blocks added in order; a (block, row) written twice or never poisons y with NaN) and compared with scipy.
QPS_HOST_TEST_LIB selects another build of the same library (tests/tools/run_sanitizers.sh points it at the AddressSanitizer + UBSan one).
Reference: the three products of the CG operator, LinearSystemSolvers.jl:152-157."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from spmv_cases import draw_case, moderately_dense, spd_companion

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quadraticprogramsolver_amd", "csrc")


@pytest.fixture(scope="module")
def shim():
    path = os.environ.get("QPS_HOST_TEST_LIB")
    if not path:
        subprocess.check_call(["make", "-C", CSRC, "-s", "host-test"])
        path = os.path.join(ROOT, "quadraticprogramsolver_amd", "libqps_host_test.so")
    L = C.CDLL(path)
    L.lt_csc_asymmetry.restype = C.c_int64
    L.lt_csc_to_csr.restype = C.c_int64
    L.lt_reduced_matrix.restype = C.c_int64
    i64, p64, pd, p32 = C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.lt_reduced_matrix.argtypes = [i64, i64, p64, p64, pd, p64, p64, pd, i64, p32, p32, pd, pd, pd, p64]
    L.lt_csc_to_csr.argtypes = [i64, i64, p64, p64, pd, C.c_int, p32, p32, pd, p32, p32, pd]
    L.lt_validate_csc.argtypes = [i64, i64, p64, p64, pd, C.c_int, C.c_char_p, C.c_int]
    L.lt_csc_asymmetry.argtypes = [i64, p64, p64, pd, C.c_int]
    return L


def _ip32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _ip64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def apply(shim, form, dtype, M, x, wgs=512):
    """y = M x through the layout `form` ("sell" / "tasks"); returns (status, y, stats)."""
    M = sp.csr_matrix(M); M.sort_indices()
    rp, ci, va = M.indptr.astype(np.int32), M.indices.astype(np.int32), np.ascontiguousarray(M.data, dtype=np.float64)
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1)
    y = np.full(M.shape[0], np.nan)
    stats = np.zeros(8, dtype=np.int64)
    fn = shim.lt_sell_apply if form == "sell" else shim.lt_tasks_apply
    rc = fn(0 if dtype == "f64" else 1, M.shape[0], M.shape[1], _ip32(rp), _ip32(ci), _dp(va), wgs, _dp(np.ascontiguousarray(x)), _dp(y), _ip64(stats))
    return rc, y, stats


def check_product(shim, M, rng, tag, expect_sell=None):
    x = rng.standard_normal(M.shape[1])
    for dtype, tol in (("f64", 1e-13), ("f32", 2e-5)):
        if dtype == "f32":
            ref = sp.csr_matrix(M).astype(np.float32) @ x.astype(np.float32)
            scale = (abs(sp.csr_matrix(M)).astype(np.float64) @ np.abs(x)).max() + 1e-300       # rounding scales with sum |a_ij x_j|, not with |y_i|
        else:
            ref = sp.csr_matrix(M) @ x
            scale = (abs(sp.csr_matrix(M)) @ np.abs(x)).max() + 1e-300
        for form in ("sell", "tasks"):
            rc, y, stats = apply(shim, form, dtype, M, x)
            assert rc >= 0, (tag, form, dtype)
            if form == "sell" and expect_sell is not None:
                assert rc == (1 if expect_sell else 0), (tag, dtype, "sliced form built" if rc else "sliced form declined", stats[:4])
            if rc == 0:
                continue                                                   # the builder declined the sliced form: the handle would use the task form
            assert np.all(np.isfinite(y)), (tag, form, dtype, "a (block, row) was written twice or never", int(np.isnan(y).sum()))
            err = np.abs(y - ref).max() / scale
            assert err <= tol, (tag, form, dtype, err)


@pytest.mark.parametrize("seed", range(12))
def test_layouts_expand_back_to_the_matrix(shim, seed):
    """The fuzz generator's shapes: A, A' and the stacked [P; A] of a CG iteration, both forms, both precisions, against scipy."""
    rng = np.random.default_rng(1000 + seed)
    A, tag = draw_case(rng)
    check_product(shim, A, rng, tag + " (A)")
    check_product(shim, sp.csr_matrix(A.T), rng, tag + " (A')")
    if seed % 3 == 0:
        P = spd_companion(rng, A.shape[1], seed)
        check_product(shim, sp.vstack([P, A]).tocsr(), rng, tag + " ([P; A])")


def test_layout_edge_shapes(shim):
    """Exact block / window boundaries, one row, one column, an empty matrix, an all-empty block in the middle, a row that is long in the LAST block only."""
    rng = np.random.default_rng(5)
    for (m, n) in ((1, 1), (1, 9000), (9000, 1), (64, 7168), (65, 7169), (2048, 14336), (2049, 14337), (4096, 300)):
        A = sp.random(m, n, density=min(1.0, 6.0 / max(n, 1) + 0.002), random_state=np.random.RandomState(m + n), format="csr")
        check_product(shim, A, rng, f"edge m={m} n={n}")
    check_product(shim, sp.csr_matrix((500, 20000)), rng, "empty matrix")
    A = sp.random(3000, 22000, density=2e-4, random_state=np.random.RandomState(3), format="lil")
    A[:, 7168:14336] = 0                                                        # fp64: the whole second column block is empty
    A[17, 15000:22000] = 1.0                                                    # 7000 entries in the last block: beyond SLONG and beyond BCHUNK
    check_product(shim, sp.csr_matrix(A), rng, "empty middle block + long last-block row")


def test_moderately_dense_matrix_goes_to_the_task_form(shim):
    """Advisor (round 3): at ~5 % density every row holds far more than SLONG = 96 entries per column block; the sliced form would be all long rows (its slices empty,
    lci / lva a second copy of the matrix), so the builder declines it and the handle takes the task form, which must reproduce the product; one row beyond BCHUNK = 2048
    entries in a block exercises the task form's long-row path."""
    rng = np.random.default_rng(6)
    A, tag = moderately_dense(rng, 700, 15000, 0.05)
    check_product(shim, A, rng, tag, expect_sell=False)
    B = sp.lil_matrix(A); B[5, :] = rng.standard_normal(15000)
    check_product(shim, sp.csr_matrix(B), rng, tag + " +dense_row", expect_sell=False)
    # a matrix whose rows are ALL longer than a task (no task at all, long rows only): the block must still get workgroups for them (stats[6] = workgroups per block)
    D = sp.csr_matrix(rng.standard_normal((300, 2500)))
    rc, y, stats = apply(shim, "tasks", "f64", D, np.ones(2500))
    assert rc == 1 and stats[0] == 0 and stats[3] == 300 and stats[6] >= 256, stats
    check_product(shim, D, rng, "dense 300 x 2500: long rows only", expect_sell=False)
    # ... while a few long rows in an otherwise sparse matrix stay in the sliced form (SLONG path of k_spmv_sell)
    S, tag2 = draw_case(rng, n=15000, m=4100, avg=8.0, dense_row=True, dense_cols=False, empty_run=True)
    check_product(shim, S, rng, tag2, expect_sell=True)


def test_csc_canonicalisation_and_csr_conversion(shim):
    """A caller's CSC with unsorted rows, duplicate entries and base 1 -> canonical CSC -> CSR of the matrix and of its transpose, against scipy."""
    rng = np.random.default_rng(7)
    for base in (0, 1):
        m, n = 300, 170
        A = sp.random(m, n, density=0.03, random_state=np.random.RandomState(9 + base), format="coo")
        rows = np.concatenate([A.row, A.row[:50]]); cols = np.concatenate([A.col, A.col[:50]]); vals = np.concatenate([A.data, rng.standard_normal(50)])   # 50 duplicates
        order = rng.permutation(rows.size)
        rows, cols, vals = rows[order], cols[order], vals[order]
        key = np.argsort(cols, kind="stable")                                   # group by column, rows left unsorted inside a column
        rows, cols, vals = rows[key], cols[key], vals[key]
        cp = np.zeros(n + 1, dtype=np.int64); np.add.at(cp, cols + 1, 1); cp = np.cumsum(cp) + base
        ri = (rows + base).astype(np.int64); nz = np.ascontiguousarray(vals)
        ref = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); ref.sum_duplicates(); ref.sort_indices()
        nnz = rows.size
        rp, ci, va = np.zeros(m + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        trp, tci, tva = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        got = shim.lt_csc_to_csr(m, n, _ip64(cp), _ip64(ri), _dp(nz), base, _ip32(rp), _ip32(ci), _dp(va), _ip32(trp), _ip32(tci), _dp(tva))
        assert got == ref.nnz
        assert np.array_equal(rp, ref.indptr) and np.array_equal(ci[:got], ref.indices) and np.allclose(va[:got], ref.data, rtol=0, atol=1e-15)
        refT = sp.csr_matrix(ref.T); refT.sort_indices()
        assert np.array_equal(trp, refT.indptr) and np.array_equal(tci[:got], refT.indices) and np.allclose(tva[:got], refT.data, rtol=0, atol=1e-15)


def test_csc_validation_and_symmetry(shim):
    """What qps_create_csc refuses before a device is touched (the same function serves the library and this test)."""
    P = sp.csc_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 3.0, 0.5], [0.0, 0.5, 4.0]]))
    cp, ri, nz = P.indptr.astype(np.int64), P.indices.astype(np.int64), P.data.copy()
    msg = C.create_string_buffer(200)
    assert shim.lt_validate_csc(3, 3, _ip64(cp), _ip64(ri), _dp(nz), 0, msg, 200) == 0
    assert shim.lt_csc_asymmetry(3, _ip64(cp), _ip64(ri), _dp(nz), 0) == -1
    bad = ri.copy(); bad[2] = 3
    assert shim.lt_validate_csc(3, 3, _ip64(cp), _ip64(bad), _dp(nz), 0, msg, 200) == 2 and b"row index out of range" in msg.value
    bad = cp.copy(); bad[1] = 5; bad[2] = 3
    assert shim.lt_validate_csc(3, 3, _ip64(bad), _ip64(ri), _dp(nz), 0, msg, 200) == 1 and b"monotone" in msg.value
    assert shim.lt_validate_csc(3, 3, _ip64(cp), _ip64(ri), _dp(nz), 1, msg, 200) == 1 and b"index_base" in msg.value
    bad = nz.copy(); bad[1] = np.inf
    assert shim.lt_validate_csc(3, 3, _ip64(cp), _ip64(ri), _dp(bad), 0, msg, 200) == 3
    asym = nz.copy(); asym[1] = 1.5                                             # P[1, 0] != P[0, 1]
    assert shim.lt_csc_asymmetry(3, _ip64(cp), _ip64(ri), _dp(asym), 0) == 0
    # an explicit zero on one side only is still symmetric
    Z = sp.csc_matrix((np.array([1.0, 0.0, 2.0]), (np.array([0, 1, 1]), np.array([0, 0, 1]))), shape=(2, 2))
    assert shim.lt_csc_asymmetry(2, _ip64(Z.indptr.astype(np.int64)), _ip64(Z.indices.astype(np.int64)), _dp(Z.data.copy()), 0) == -1


def test_reduced_matrix_of_itrsolcg(shim):
    """ItrSolCgInit (LinearSystemSolvers.jl:112-114): mAA = mA' * mA and mPI = mP + sigma I on ONE frozen pattern = pattern(P) U pattern(A'A) U diagonal, so that
    mL = vP + sigma diag + rho vAA elementwise (:114, :128).  Against scipy on a banded A (control-like), a bidiagonal A (isotonic regression) and a random one;
    the size cap makes the builder decline."""
    rng = np.random.default_rng(12)
    n = 900
    banded = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-3, 4)], list(range(-3, 4)), shape=(n, n), format="csc")[: n - 50, :]
    bidiag = sp.diags([-np.ones(n - 1), np.ones(n - 1)], [0, 1], shape=(n - 1, n), format="csc")
    rnd = sp.random(700, n, density=0.01, random_state=np.random.RandomState(2), format="csc")
    for A, name in ((sp.csc_matrix(banded), "banded"), (bidiag, "bidiagonal"), (rnd, "random")):
        m = A.shape[0]
        P = sp.csc_matrix(sp.diags(rng.random(n) + 0.5) + (sp.random(n, n, density=0.002, random_state=np.random.RandomState(3), format="csc") if name == "random" else 0))
        P = sp.csc_matrix((P + P.T) * 0.5); A.sort_indices(); P.sort_indices()
        cap = 4 * (P.nnz + (A.T @ A).nnz) + n
        rp, ci = np.zeros(n + 1, np.int32), np.zeros(cap, np.int32)
        vP, vAA, dg = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        work = C.c_int64(0)
        args = (n, m, _ip64(P.indptr.astype(np.int64)), _ip64(P.indices.astype(np.int64)), _dp(P.data.copy()), _ip64(A.indptr.astype(np.int64)), _ip64(A.indices.astype(np.int64)),
                _dp(A.data.copy()))
        nnz = shim.lt_reduced_matrix(*args, cap, _ip32(rp), _ip32(ci), _dp(vP), _dp(vAA), _dp(dg), C.byref(work))
        assert nnz > 0, (name, nnz)
        assert work.value == int((np.diff(sp.csr_matrix(A).indptr).astype(np.int64) ** 2).sum())
        rho, sigma = 0.37, 1e-3
        L = sp.csr_matrix((vP[:nnz] + sigma * dg[:nnz] + rho * vAA[:nnz], ci[:nnz], rp), shape=(n, n))
        ref = (P + sigma * sp.identity(n) + rho * (A.T @ A)).toarray()
        assert np.abs(L.toarray() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()), name
        assert np.all(np.diff(rp) >= 1) and all(np.all(np.diff(ci[rp[j]:rp[j + 1]]) > 0) for j in range(n))          # a diagonal entry everywhere, sorted columns
        assert np.array_equal(sp.csr_matrix((dg[:nnz], ci[:nnz], rp), shape=(n, n)).toarray(), np.eye(n))
        assert shim.lt_reduced_matrix(*args, nnz - 1, _ip32(rp), _ip32(ci), _dp(vP), _dp(vAA), _dp(dg), C.byref(work)) == -1   # over the cap: declined


def test_stream_row_blocks(shim):
    """CSR-stream kernel: consecutive rows of <= 1024 non-zeros per workgroup, a longer row alone; the blocks must tile [0, nrows)."""
    rng = np.random.default_rng(8)
    A, _ = draw_case(rng, n=9000, m=4100, avg=20.0, dense_row=True, dense_cols=False, empty_run=True)
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
    rb = np.zeros(A.shape[0] + 2, np.int32); nb = C.c_int(0)
    shim.lt_stream_blocks(A.shape[0], A.shape[1], _ip32(rp), _ip32(ci), _dp(va), _ip32(rb), C.byref(nb))
    rb = rb[:nb.value + 1]
    assert rb[0] == 0 and rb[-1] == A.shape[0] and np.all(np.diff(rb) > 0)
    for a, b in zip(rb[:-1], rb[1:]):
        assert (rp[b] - rp[a] <= 1024 or b - a == 1) and b - a <= 256
    # a matrix that is mostly empty rows (the zero block of a lasso / Huber P): the blocks stay short in ROWS too
    E = sp.csr_matrix((np.ones(50), (np.arange(50), np.arange(50))), shape=(30000, 30000))
    rb = np.zeros(30002, np.int32)
    shim.lt_stream_blocks(30000, 30000, _ip32(E.indptr.astype(np.int32)), _ip32(E.indices.astype(np.int32)), _dp(E.data.copy()), _ip32(rb), C.byref(nb))
    rb = rb[:nb.value + 1]
    assert rb[0] == 0 and rb[-1] == 30000 and np.diff(rb).max() <= 256


def test_ldl_symbolic_analysis_on_the_host_build(shim):
    """ldl_symbolic.cpp in the host-only build: a valid permutation, fill no worse than natural order on a KKT pattern with structure (the sanitizer run walks the
    quotient-graph code through this)."""
    rng = np.random.default_rng(9)
    n, m = 400, 700
    A = sp.random(m, n, density=0.01, random_state=np.random.RandomState(4), format="csc")
    P = spd_companion(rng, n, 11)
    perm = np.zeros(n + m, dtype=np.int64); rep = np.zeros(8, dtype=np.int64)
    rc = shim.lt_ldl_analyze(n, m, _ip64(P.indptr.astype(np.int64)), _ip64(P.indices.astype(np.int64)), _ip64(A.indptr.astype(np.int64)), _ip64(A.indices.astype(np.int64)), 0,
                             8192, 64, 4096, _ip64(perm), _ip64(rep))
    assert rc == 0 and sorted(perm.tolist()) == list(range(n + m))
    assert rep[0] == n + m and rep[1] + rep[2] == n + m and rep[6] >= rep[5] - (n + m)          # nnz(L) >= strictly-lower nnz(K) (diagonal aside)


def test_tile_order_deals_every_lower_tile_exactly_once(shim):
    """tile_order.h: the workgroup id -> tile maps of the MFMA GEMM's lower-tile launch (k_gemm, XCD-aware order) and of the Cholesky's fused trailing update
    (k_chol_update_diag, with and without the rule that keeps ids on the diagonal workgroup's XCD idle) deal every lower tile exactly once for every grid size,
    and the GEMM order keeps the first 64 tiles of every XCD within few operand panels (what it exists for)."""
    shim.lt_tile_order.restype = C.c_int
    shim.lt_tile_order.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for nt in list(range(1, 41)) + [62, 63, 64, 65, 100, 128, 256]:
        for kind in (0, 1, 2):
            tiles = np.zeros((nt, nt), dtype=np.int32); panels = np.zeros(8, dtype=np.int32)
            nids = shim.lt_tile_order(kind, nt, tiles.ctypes.data, panels.ctypes.data)
            assert nids > 0, (kind, nt, nids)
            want = np.tril(np.ones((nt, nt), dtype=np.int32))
            if kind != 0:
                want[0, 0] = 0                                               # the diagonal workgroup's own tile
            assert np.array_equal(tiles, want), (kind, nt)
            if kind == 0:
                assert nids % 8 == 0 and nids - nt * (nt + 1) // 2 < 8
                if nt >= 32:
                    # 64 consecutive tiles of the super-block order touch 8 + 8 panels per block they straddle (the plain 2-D grid: ~45 at the top of the triangle);
                    # a ragged last super-row (nt = 33: one tile row of 33 tiles) is the exception
                    assert np.median(panels) <= 26 and panels.max() <= 48, (nt, panels)


def test_staged_sliced_form_owns_whole_windows(shim):
    """The staged variant of the sliced form (spmv_layout.h, SellLayout): once a workgroup's share of the rows reaches 1024 the sorting window IS that share, a
    workgroup owns whole windows, sums their long rows itself and stores a window as one run.  The interpreter walks it the way the kernel does (a window row
    nobody sums, or summed twice, poisons y; a workgroup range that is not window-aligned raises).  Uniform matrices go staged; a matrix whose windows cost very
    different amounts keeps the slice-granular ranges."""
    rng = np.random.default_rng(12)
    x = None
    cases = []
    # (rows, cols, density, wgs, expect staged): share = ceil(rows / (wgs // nblk))
    for rows, cols, avg, wgs, want in ((40000, 9000, 6.0, 64, True), (52001, 20011, 4.0, 64, True), (150000, 5000, 12.0, 64, True), (9000, 9000, 6.0, 64, False)):
        M = sp.random(rows, cols, density=avg / cols, random_state=rng, format="csr", dtype=np.float64)
        cases.append((f"uniform {rows}x{cols}", M, wgs, want))
    # long rows inside staged windows (dense rows: more than SLONG entries per block) and an empty stretch of rows
    M = sp.random(40000, 8000, density=5.0 / 8000, random_state=rng, format="lil", dtype=np.float64)
    for r in (0, 2111, 2112, 17000, 39999):
        M[r, :] = rng.standard_normal(8000)
    M[20000:23000, :] = 0
    cases.append(("long rows + empty stretch", M.tocsr(), 64, None))
    # skewed: the first tenth of the rows carries most of the entries -> windows of very different cost -> slice-granular ranges
    top = sp.random(4000, 8000, density=80.0 / 8000, random_state=rng, format="csr", dtype=np.float64)
    rest = sp.random(36000, 8000, density=1.0 / 8000, random_state=rng, format="csr", dtype=np.float64)
    cases.append(("skewed", sp.vstack([top, rest]).tocsr(), 64, False))
    for tag, M, wgs, want in cases:
        x = rng.standard_normal(M.shape[1])
        ref = M @ x
        scale = (abs(M) @ np.abs(x)).max() + 1e-300
        for dtype, tol in (("f64", 1e-13), ("f32", 2e-5)):
            rc, y, stats = apply(shim, "sell", dtype, M, x, wgs=wgs)
            assert rc == 1, (tag, dtype, rc)
            staged, win = divmod(int(stats[7]), 100000)
            assert win % 64 == 0 and 64 <= win <= 2304, (tag, win)
            if want is not None and dtype == "f64":
                assert bool(staged) == want, (tag, dtype, "staged" if staged else "lane-by-lane", win, stats)
            if staged:
                assert win >= 1024
            assert np.all(np.isfinite(y)), (tag, dtype, "a (block, row) was written twice or never", int(np.isnan(y).sum()))
            assert np.abs(y - ref).max() / scale <= tol, (tag, dtype)
