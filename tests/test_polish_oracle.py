"""CPU checks of the polishing restatement (oracle/polish_oracle_np.py; SolveQuadraticProgram.m:289-325).  PARITY UNPINNED:
MATLAB cannot run here, so the restatement is pinned by the linear algebra it states -- MINRES against scipy's independent
implementation and a direct solve, the polished primal against a direct solve of the reduced KKT system."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from quadraticprogramsolver_amd.generator import GenerateDenseBenchmarkQP, make_rng


@pytest.fixture(scope="module")
def pol():
    from oracle import polish_oracle_np
    return polish_oracle_np


def admm_state(np_oracle, P, q, A, l, u, eps):
    x = np.zeros(P.shape[0]); info = {}
    np_oracle.SolveQuadraticProgramRefLoop(x, P, q, A, l, u, np_oracle.RedCholInit, np_oracle.RedChol, numIterations=4000, εAbs=eps, εRel=eps,
                                           ρ=0.1, adptΡ=True, info=info)
    return x, info["y"]


@pytest.mark.parametrize("n,seed", [(30, 1), (120, 2)])
def test_minres_matches_direct_solve_and_scipy(pol, n, seed):
    rng = make_rng(77, seed)
    B = rng.standard_normal((n, n)); K = B + B.T + np.diag(np.linspace(-3, 3, n))          # symmetric indefinite
    b = rng.standard_normal(n)
    x, flag, relres, it = pol.minres(lambda v: K @ v, b, 1e-10, 10 * n, np.zeros(n))
    assert flag == 0 and relres <= 1e-10 and it <= 10 * n
    assert np.linalg.norm(K @ x - b) <= 2e-10 * np.linalg.norm(b)
    xs, info = spl.minres(K, b, rtol=1e-12, maxiter=20 * n)
    assert np.abs(x - xs).max() <= 1e-6 * max(1.0, np.abs(xs).max())
    # warm start: from the solution itself nothing is left to do
    x2, flag2, _, it2 = pol.minres(lambda v: K @ v, b, 1e-8, 50, x)
    assert flag2 == 0 and it2 == 0 and np.array_equal(x2, x)
    # b = 0 -> x = 0, converged (MATLAB's documented behaviour)
    x3, flag3, _, _ = pol.minres(lambda v: K @ v, np.zeros(n), 1e-8, 50, x)
    assert flag3 == 0 and not x3.any()
    # iteration cap -> flag 1
    assert pol.minres(lambda v: K @ v, b, 1e-14, 3, np.zeros(n))[1] == 1


@pytest.mark.parametrize("n,m,stream", [(40, 80, 1), (64, 128, 2)])
def test_polish_with_a_correct_active_set_reaches_the_kkt_point(np_oracle, pol, n, m, stream):
    P, q, A, l, u = (np.asarray(a) for a in GenerateDenseBenchmarkQP(n, m, feasible=True, stream=stream))
    x, y = admm_state(np_oracle, P, q, A, l, u, 1e-4)
    y = np.where(np.abs(y) > 1e-7, y, 0.0)            # drop the rounding noise on inactive rows: "identified correctly" (:290-291)
    xp, flag, info = pol.Polish(P, q, A, l, u, x, y, 10, 1e-6, 1e-9, 2000)
    assert flag == 0 and info["numActiveLower"] + info["numActiveUpper"] <= n
    L, U = y < 0, y > 0
    Aa = np.vstack([A[L], A[U]]); g = np.concatenate([-q, l[L], u[U]])
    t = np.linalg.solve(np.block([[P, Aa.T], [Aa, np.zeros((Aa.shape[0],) * 2)]]), g)
    assert np.abs(xp - t[:n]).max() <= 1e-7
    xs = np.zeros(n)
    np_oracle.SolveQuadraticProgramRefLoop(xs, P, q, A, l, u, np_oracle.RedCholInit, np_oracle.RedChol, numIterations=50000, εAbs=1e-11, εRel=1e-11, ρ=0.1, adptΡ=True)
    assert np.abs(xp - xs).max() <= 1e-7 < np.abs(x - xs).max()      # polishing gained three digits on the 1e-4 iterate


def test_polish_semantics_of_the_flag(np_oracle, pol):
    P, q, A, l, u = (np.asarray(a) for a in GenerateDenseBenchmarkQP(40, 80, feasible=True, stream=1))
    x, y = admm_state(np_oracle, P, q, A, l, u, 1e-4)
    xp, flag, _ = pol.Polish(P, q, A, l, u, x, y, 0)                 # numPolishItr = 0: nothing runs, minresFlag stays -1 (:311)
    assert flag == -1 and np.array_equal(xp, x)
    xp, flag, info = pol.Polish(P, q, A, l, u, x, y, 10, 1e-6, 1e-6, 2)   # MINRES cannot converge in 2 iterations: x kept (:316-325)
    assert flag == 1 and info["refinements"] == 1 and np.array_equal(xp, x)
    # the literal sign test picks up rounding noise of y on inactive rows (more "active" rows than variables here)
    xp, flag, info = pol.Polish(P, q, A, l, u, x, y)
    assert info["numActiveLower"] + info["numActiveUpper"] > 40
