"""ONE application of the operator's matrices on the device against scipy (round-3 review item 2 / advisor): the column-blocked SpMV kernels -- sliced form
k_spmv_sell and task form k_spmv_blk, incl. both long-row paths (more than SLONG = 96 entries of a row in a block: a wave per row; more than BCHUNK = 2048: a workgroup
per row) -- the stacked [P; A] product of a CG iteration, A' v, the CSR-stream kernel and the dense GEMVs, through qps_operator_apply.  Until now every check of these
kernels went through a whole CG solve, where a 10x band on the fp32 residual could not separate recurrence drift from a small wrong contribution
(profiles/r03_y_fuzz_spmv.log, case 24); here a missing, duplicated or misplaced entry shows at 1e-13 (fp64) / 2e-5 (fp32) of sum_j |a_ij x_j|.
Reference: the three products of the operator, LinearSystemSolvers.jl:152-157 (mul!(vZZ, mA, vW); mul!(vU, mA', vZZ); mul!(vU, mP, vW, 1.0, rho))."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from spmv_cases import draw_case, moderately_dense, spd_companion

pytestmark = pytest.mark.gpu
TOL = {"f64": 1e-13, "f32": 2e-5}


def scale_of(M, v):
    return float((abs(sp.csr_matrix(M)) @ np.abs(v)).max()) + 1e-300            # rounding scales with sum |a_ij v_j|


def check_operator(gpu, P, A, rng, tag, dtype, monkeypatch, sell, blocked="1"):
    monkeypatch.setenv("QPS_SPMV_BLOCKED", blocked)                              # read per handle, at creation
    monkeypatch.setenv("QPS_SPMV_SELL", sell)
    n, m = P.shape[0], A.shape[0]
    u, v = rng.standard_normal(n), rng.standard_normal(m)
    Pc, Ac, At = sp.csr_matrix(P), sp.csr_matrix(A), sp.csr_matrix(A.T)
    rho, sigma = 0.7, 1e-3
    with gpu.QuadraticProgram(P, rng.standard_normal(n), A, np.zeros(m), np.zeros(m), linsys="cg", dtype=dtype) as prob:
        got = {"P": prob.apply("P", u), "A": prob.apply("A", u), "At": prob.apply("At", v), "PA": prob.apply("PA", u), "reduced": prob.apply("reduced", u, ρ=rho, σ=sigma)}
    Au = Ac @ u
    ref = {"P": Pc @ u, "A": Au, "At": At @ v, "PA": np.concatenate([Pc @ u, Au]), "reduced": Pc @ u + rho * (At @ Au) + sigma * u}
    sc = {"P": scale_of(Pc, u), "A": scale_of(Ac, u), "At": scale_of(At, v), "PA": max(scale_of(Pc, u), scale_of(Ac, u)),
          "reduced": scale_of(Pc, u) + rho * scale_of(At, np.abs(Ac) @ np.abs(u)) + sigma * np.abs(u).max()}
    for k in got:
        assert np.all(np.isfinite(got[k])), (tag, dtype, sell, k)
        err = np.abs(got[k] - ref[k]).max() / sc[k]
        assert err <= TOL[dtype] * (4 if k == "reduced" else 1), (tag, dtype, f"sell={sell}", k, err)


@pytest.mark.parametrize("sell", ["1", "0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_one_product_against_scipy_over_the_fuzz_shapes(gpu, monkeypatch, dtype, sell):
    """Five draws of the fuzz generator per (precision, form): heavy tails, empty runs, dense rows and columns, ragged last slices, 1-3 column blocks."""
    rng = np.random.default_rng(300 + (dtype == "f32") + 2 * (sell == "0"))
    for c in range(5):
        A, tag = draw_case(rng)
        P = spd_companion(rng, A.shape[1], c)
        check_operator(gpu, P, sp.csc_matrix(A), rng, f"draw {c}: {tag}", dtype, monkeypatch, sell)


@pytest.mark.parametrize("sell", ["1", "0"])
def test_the_shape_of_fuzz_case_24(gpu, monkeypatch, sell):
    """The case the round-3 fuzz log flagged (fp32, task form, n = 22 000, a dense row of 2216+ entries per block -- the lr_ptr / lr_desc path of k_spmv_blk, and the SLONG
    path of k_spmv_sell) at the operator level, both precisions: the products themselves are exact to rounding, so what that log saw was CG drift."""
    rng = np.random.default_rng(24)
    A, tag = draw_case(rng, n=22000, m=12345, avg=8.0, dense_row=True, dense_cols=True, empty_run=True)
    P = spd_companion(rng, 22000, 24)
    for dtype in ("f32", "f64"):
        check_operator(gpu, P, sp.csc_matrix(A), rng, "case-24 shape: " + tag, dtype, monkeypatch, sell)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_moderately_dense_matrix_on_the_device(gpu, monkeypatch, dtype):
    """~5 % density: every row is longer than SLONG in every block, the builder declines the sliced form (tests/test_layout_cpu.py) and the handle's task form
    must carry the product -- with QPS_SPMV_SELL at its default."""
    rng = np.random.default_rng(31)
    A, tag = moderately_dense(rng, 700, 15000, 0.05)
    B = sp.lil_matrix(A); B[5, :] = rng.standard_normal(15000)                    # + one row beyond BCHUNK entries per block
    P = spd_companion(rng, 15000, 31)
    monkeypatch.delenv("QPS_SPMV_SELL", raising=False)
    check_operator(gpu, P, sp.csc_matrix(B), rng, tag, dtype, monkeypatch, sell="1")


def test_csr_stream_kernel_and_dense_gemvs(gpu, monkeypatch):
    """The same entry on the un-blocked CSR-stream kernel (QPS_SPMV_BLOCKED = 0) and on a dense handle (row / column GEMVs of the loop)."""
    rng = np.random.default_rng(41)
    A, tag = draw_case(rng, n=5000, m=4100, avg=8.0, dense_row=True, dense_cols=False, empty_run=True)
    P = spd_companion(rng, 5000, 41)
    for dtype in ("f64", "f32"):
        check_operator(gpu, P, sp.csc_matrix(A), rng, "stream: " + tag, dtype, monkeypatch, sell="1", blocked="0")
    n, m = 300, 500
    Pd = rng.standard_normal((n, n)); Pd = Pd.T @ Pd / n + 0.1 * np.eye(n); Ad = rng.standard_normal((m, n))
    u, v = rng.standard_normal(n), rng.standard_normal(m)
    for dtype in ("f64", "f32"):
        with gpu.QuadraticProgram(Pd, rng.standard_normal(n), Ad, np.zeros(m), np.zeros(m), dtype=dtype) as prob:
            tol = 1e-13 if dtype == "f64" else 2e-5
            assert np.abs(prob.apply("P", u) - Pd @ u).max() <= tol * (np.abs(Pd) @ np.abs(u)).max()
            assert np.abs(prob.apply("A", u) - Ad @ u).max() <= tol * (np.abs(Ad) @ np.abs(u)).max()
            assert np.abs(prob.apply("At", v) - Ad.T @ v).max() <= tol * (np.abs(Ad.T) @ np.abs(v)).max()
            assert np.abs(prob.apply("PA", u) - np.concatenate([Pd @ u, Ad @ u])).max() <= tol * max((np.abs(Pd) @ np.abs(u)).max(), (np.abs(Ad) @ np.abs(u)).max())
            r = prob.apply("reduced", u, ρ=0.3, σ=1e-2)
            rr = Pd @ u + 0.3 * (Ad.T @ (Ad @ u)) + 1e-2 * u
            assert np.abs(r - rr).max() <= 8 * tol * ((np.abs(Pd) @ np.abs(u)).max() + 0.3 * (np.abs(Ad.T) @ (np.abs(Ad) @ np.abs(u))).max())


def test_unsorted_duplicated_csc_input_at_the_c_abi(gpu):
    """A C caller's CSC need not be sorted or free of duplicates (Julia's sparse() guarantees both; include/qps.h accepts either): the handle canonicalises before it
    builds its layouts, so the products equal those of the summed, sorted matrix."""
    import ctypes as C
    from quadraticprogramsolver_amd import _lib
    rng = np.random.default_rng(51)
    n, m = 9000, 2500
    A, _ = draw_case(rng, n=n, m=m, avg=8.0, dense_row=True, dense_cols=False, empty_run=False)
    P = spd_companion(rng, n, 51)
    coo = sp.coo_matrix(A)
    rows = np.concatenate([coo.row, coo.row[:400]]); cols = np.concatenate([coo.col, coo.col[:400]]); vals = np.concatenate([coo.data, rng.standard_normal(400)])
    order = rng.permutation(rows.size); rows, cols, vals = rows[order], cols[order], vals[order]
    key = np.argsort(cols, kind="stable"); rows, cols, vals = rows[key], cols[key], vals[key]        # grouped by column, rows unsorted inside
    cp = np.zeros(n + 1, dtype=np.int64); np.add.at(cp, cols + 1, 1); cp = np.cumsum(cp) + 1          # index_base 1
    ri = (rows + 1).astype(np.int64); nz = np.ascontiguousarray(vals)
    Aref = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); Aref.sum_duplicates()
    Pc = sp.csc_matrix(P); Pc.sort_indices()
    Pcp, Pri, Pnz = (Pc.indptr + 1).astype(np.int64), (Pc.indices + 1).astype(np.int64), np.ascontiguousarray(Pc.data)
    q, l, u = rng.standard_normal(n), np.zeros(m), np.zeros(m)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    h = C.c_void_p()
    L = _lib.lib()
    _lib.check(L.qps_create_csc(n, m, ip(Pcp), ip(Pri), dp(Pnz), ip(cp), ip(ri), dp(nz), dp(q), dp(l), dp(u), 1, 0, _lib.QPS_F64, 0, C.byref(h)))
    try:
        uvec, vvec = rng.standard_normal(n), rng.standard_normal(m)
        out = np.zeros(n + m)
        _lib.check(L.qps_operator_apply(h, _lib.QPS_OP_PA, dp(uvec), dp(out), 1.0, 0.0), h)
        ref = np.concatenate([sp.csr_matrix(P) @ uvec, Aref @ uvec])
        assert np.abs(out - ref).max() <= 1e-13 * max(scale_of(P, uvec), scale_of(Aref, uvec))
        out2 = np.zeros(n)
        _lib.check(L.qps_operator_apply(h, _lib.QPS_OP_AT, dp(vvec), dp(out2), 1.0, 0.0), h)
        assert np.abs(out2 - Aref.T @ vvec).max() <= 1e-13 * scale_of(Aref.T, vvec)
    finally:
        L.qps_destroy(h)


_C3_SIZE = []


def c3_size_matrices():
    """(P, A) at BASELINE config 3's size, built once for both parametrisations and without scipy's lil format or a legacy RandomState (`sp.random` draws its positions
    by permuting ALL m * n candidates with those: minutes and 20 GB at this size -- 190 of the suite's 570 s before)."""
    if not _C3_SIZE:
        rng = np.random.default_rng(77)
        n, m = 50000, 100000
        A = sp.random(m, n, density=1e-3, random_state=rng, format="csr", dtype=np.float64)
        dense_rows = np.array([0, 2111, 2112, 54321, m - 1])
        keep = np.ones(m); keep[60000:63000] = 0.0; keep[dense_rows] = 0.0                          # an empty stretch of constraint rows
        R = sp.random(dense_rows.size, n, density=0.05, random_state=rng, format="csr", dtype=np.float64)   # ~2500 entries: ~360 per column block (> SLONG)
        S = sp.csr_matrix((np.ones(dense_rows.size), (dense_rows, np.arange(dense_rows.size))), shape=(m, dense_rows.size))
        A = (sp.diags(keep) @ A + S @ R).tocsc(); A.eliminate_zeros(); A.sort_indices()
        M = sp.random(n, n, density=3.0 / n, random_state=rng, data_rvs=rng.standard_normal, format="csc")
        P = (M.T @ M + 1e-2 * sp.identity(n)).tocsc()                                               # spd_companion's recipe with a Generator
        _C3_SIZE.append((P, A))
    return _C3_SIZE[0]


@pytest.mark.parametrize("staged", ["1", "0"])
def test_staged_sliced_form_at_baseline_config_3_size(gpu, monkeypatch, staged):
    """The staged variant of the sliced form (a workgroup owns whole sorting windows, collects their row sums in LDS and stores a window as one run; spmv_layout.h) only
    comes into play when a workgroup's share of the rows reaches 1024 -- BASELINE config 3's size (n = 50 000, m = 100 000): every product of the operator against scipy,
    both precisions, with a few dense rows (long rows INSIDE staged windows, summed by the window's owner) and an empty stretch of constraint rows, staged and
    QPS_SPMV_STAGED=0 (the lane-by-lane stores) on the same matrices."""
    monkeypatch.setenv("QPS_SPMV_STAGED", staged)                                # read per handle, at creation
    P, A = c3_size_matrices()
    rng = np.random.default_rng(79)
    for dtype in ("f64", "f32"):
        check_operator(gpu, P, A, rng, f"c3-size staged={staged}", dtype, monkeypatch, "1")


def test_staged_form_with_several_windows_per_workgroup(gpu, monkeypatch):
    """A workgroup's share of the rows beyond the largest window (WMAX = 2304 rows): the share is cut into equal windows and a workgroup walks several of them, reusing
    its LDS array behind a second barrier -- tall thin A (1.3 M x 5000: [P; A] u has 1020 windows of 1280 rows on 512 workgroups) and, transposed, a product whose 182
    column blocks leave two workgroups per block with two windows each."""
    rng = np.random.default_rng(78)
    n, m = 5000, 1300000
    A = sp.random(m, n, density=9.0 / n, random_state=rng, format="csc", dtype=np.float64)   # about as many entries per row as P: the windows of [P; A] cost the same
    P = spd_companion(rng, n, 5)
    check_operator(gpu, P, A, rng, "several windows per workgroup", "f64", monkeypatch, "1")
