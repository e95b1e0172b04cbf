"""GPU parity of the sparse direct KKT plugin (QPS_LINSYS_KKT_LDL; HipLdlInit / HipLdl) -- the counterpart of the plugin the
reference's own tests and benchmarks select: FacLdlInit / FacLdl! (RunTests.jl:55-56, RunBenchmarks.jl:54-55), QDLdlInit / QDLdl!
(SolveQuadraticProgramUnitTest.jl:65-66), LaLdlInit / LaLdl! (LinearSystemSolvers.jl:16-107).

Oracle: the C restatement with its sparse CSC L D L' plugin (QDLDL's published algorithm, oracle/qps_oracle.c kind 4; ordering from
SuperLU's MMD, i.e. shared with nothing in the product), itself checked against the dense KKT oracle in tests/test_oracle.py.
Tolerances: iterate level 1e-9 relative where K is well conditioned (classes 1-4, 9); the lasso / Huber / SVM / portfolio classes
have zero blocks in P, so K carries pivots of size sigma = 1e-6 and two correct factorisations with different orderings differ by
~1e-16 * cond(K) ~ 1e-8 (the sparse and the dense KKT ORACLES differ by 7e-9 on them): 1e-6 there; solution level 1e-5
(RunTests.jl:58) with the oracle's flag, iteration and re-factorisation counts everywhere."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from quadraticprogramsolver_amd.generator import GenerateRandomQP, ProblemClass, make_rng

pytestmark = pytest.mark.gpu
ABS_DEV_THR = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ILL = (ProblemClass.portfolioOptimization, ProblemClass.lassoOptimization, ProblemClass.huberFitting, ProblemClass.supportVectorMachine)


def note(line):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "ldl_parity.log"), "a") as f:
        f.write(line + "\n")


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if b.size else 0.0


def kkt(P, A, rho, sigma):
    n, m = P.shape[0], A.shape[0]
    return sp.bmat([[sp.csc_matrix(P) + sigma * sp.eye(n), sp.csc_matrix(A).T], [sp.csc_matrix(A), -sp.eye(m) / rho]], format="csc")


PAIR_CASES = [(ProblemClass.randomQp, 100), (ProblemClass.equalityConstrainedQp, 100), (ProblemClass.portfolioOptimization, 100),
              (ProblemClass.lassoOptimization, 20), (ProblemClass.huberFitting, 10), (ProblemClass.supportVectorMachine, 20),
              (ProblemClass.isotonicRegression, 300), (ProblemClass.randomQp, 10), (ProblemClass.inequalityConstrainedQp, 120)]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("pc,n", PAIR_CASES)
def test_ldl_plugin_pair_in_isolation(gpu, pc, n, dtype):
    """LinSysSolInit / LinSysSol! of the direct plugins (LinearSystemSolvers.jl:16-44): K [x~; nu] = [sigma x - q; z - y / rho],
    z~ = z + (nu - y) / rho, including the changedRho re-factorisation -- against SuperLU on the same K."""
    P, q, A, l, u = GenerateRandomQP(pc, n, rng=make_rng(77, int(pc)))
    nn, m = P.shape[0], A.shape[0]
    rng = make_rng(78, int(pc))
    sigma = 1e-6
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="ldl", dtype=dtype) as prob:
        prob.linsys_init(0.3, sigma)
        for changed, rho in ((False, 0.3), (True, 40.0), (False, 40.0), (True, 1e-3)):
            x, z, y = rng.standard_normal(nn), rng.standard_normal(m), rng.standard_normal(m)
            xx, zz = np.zeros(nn), np.zeros(m)
            prob.linsys_solve(x, z, y, rho, sigma, changed, xx, zz)
            sol = spla.splu(kkt(P, A, rho, sigma)).solve(np.concatenate([sigma * x - q, z - y / rho]))
            zr = z + (sol[nn:] - y) / rho
            tol = (1e-9 if pc not in ILL else 1e-6) if dtype == "f64" else 5e-2
            if dtype == "f64":
                assert rel(xx, sol[:nn]) <= tol and rel(zz, zr) <= tol, (pc, rho, rel(xx, sol[:nn]), rel(zz, zr))
            else:   # fp32: pivots of size sigma = 1e-6 next to O(1) entries -- check the residual of the well-scaled rows instead
                K = kkt(P, A, rho, sigma)
                nu = (zz - z) * rho + y
                r = K @ np.concatenate([xx, nu]) - np.concatenate([sigma * x - q, z - y / rho])
                assert np.abs(r).max() <= 5e-3 * max(1.0, np.abs(np.concatenate([sigma * x - q, z - y / rho])).max()), (pc, rho, np.abs(r).max())


ITER_CASES = [(ProblemClass.randomQp, 100), (ProblemClass.inequalityConstrainedQp, 100), (ProblemClass.equalityConstrainedQp, 100),
              (ProblemClass.optimalControl, 100), (ProblemClass.portfolioOptimization, 100), (ProblemClass.lassoOptimization, 10),
              (ProblemClass.huberFitting, 10), (ProblemClass.supportVectorMachine, 10), (ProblemClass.isotonicRegression, 100)]


@pytest.mark.parametrize("pc,n", ITER_CASES)
def test_ldl_iterates_match_oracle_all_classes(gpu, c_oracle, pc, n):
    m = (n // 2) if pc == ProblemClass.equalityConstrainedQp else 0
    P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(1234, 40 + int(pc)))
    tol = 1e-9 if pc not in ILL else 1e-6
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="ldl") as prob:
        for K in (25, 60):
            x = np.zeros(P.shape[0]); info = {}
            prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
            z, y = prob.dual()
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
            assert rel(x, xo) <= tol and rel(z, io["z"]) <= tol and rel(y, io["y"]) <= 10 * tol, (pc, K, rel(x, xo), rel(z, io["z"]), rel(y, io["y"]))
            assert abs(info["resPrim"] - io["resPrim"]) <= tol * max(1.0, io["resPrim"])
            assert abs(info["resDual"] - io["resDual"]) <= 10 * tol * max(1.0, io["resDual"])


RUNTESTS_SIZES = {pc: (10, 100) for pc in ProblemClass}                               # RunTests.jl:30-38: every class at both sizes


@pytest.mark.parametrize("pc", list(ProblemClass))
def test_ldl_runtests_sweep(gpu, c_oracle, np_oracle, pc):
    """RunTests.jl:62-99 with the plugin the reference itself selects there (:55-56 FacLdlInit / FacLdl!): every ProblemClass, both
    sizes -- lasso / Huber / SVM at numElements = 100 (10 200 / 30 100 / 10 100 variables) included -- numIterations = 50000,
    eps = 1e-7, rho = 0.1, adptRho; the assertion of :93 with the CPU oracle in OSQP's role."""
    checked = 0
    for sim in range(3):
        for n in RUNTESTS_SIZES[pc]:
            m = (n // 2) if pc == ProblemClass.equalityConstrainedQp else 0           # RunTests.jl:39-47
            P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(4321, 1000 * int(pc) + 10 * sim + (n > 10)))
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=50000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
            if io["convFlag"] == 1 or (io["convFlag"] == 2 and io["resPrim"] > 1e-4):
                continue   # infeasible tiny draw: ends by the stall test with rho at its clamp, where the stopping iteration is round-off sensitive
            x = np.zeros(P.shape[0]); info = {}
            t0 = time.perf_counter()
            flag = gpu.SolveQuadraticProgramInplace(x, P, q, A, l, u, gpu.HipLdlInit, gpu.HipLdl, numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7,
                                                    ρ=0.1, adptΡ=True, info=info)                  # RunTests.jl:85
            dt = time.perf_counter() - t0
            dev = np.abs(x - xo).max()
            note(f"runtests {pc.name} n={n} N={P.shape[0]} M={A.shape[0]} sim={sim}: flag {int(flag)}/{io['convFlag']} iterations {info['iterations']}/{io['iterations']} "
                 f"refactor {info['numRefactor']}/{io['numRefactor']} max|x-x_oracle| {dev:.2e}; setup {info['tSetup']*1e3:.1f} ms loop {info['tLoop']*1e3:.1f} ms call {dt*1e3:.1f} ms")
            assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"], (pc, n, sim, info, io["iterations"])
            assert dev <= ABS_DEV_THR                                                   # RunTests.jl:93
            checked += 1
    assert checked >= 3


def test_ldl_isotonic_800_unit_test_script(gpu, c_oracle):
    """SolveQuadraticProgramUnitTest.jl:31-43, :65-66: isotonicRegression, numElements = 800, rho = 0.1, adptRho, the QDLDL plugin.
    Time-to-eps of the three device plugins side by side (logged, not asserted)."""
    P, q, A, l, u = GenerateRandomQP(ProblemClass.isotonicRegression, 800, rng=make_rng(1234, 9))
    kw = dict(numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=5000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
    out = {}
    for name, pair in (("ldl", (gpu.HipLdlInit, gpu.HipLdl)), ("cholesky", (gpu.HipCholInit, gpu.HipChol)), ("cg", (gpu.HipCgInit, gpu.HipCg))):
        for rep in range(2):                                                            # second call: warm caches / recycled handles
            x = np.zeros(800); info = {}
            t0 = time.perf_counter()
            flag = gpu.SolveQuadraticProgramInplace(x, P, q, A, l, u, *pair, info=info, **kw)
            out[name] = (x, int(flag), info, time.perf_counter() - t0)
        note(f"isotonic n=800 {name}: flag {out[name][1]} iterations {info['iterations']} refactor {info['numRefactor']} setup {info['tSetup']*1e3:.2f} ms "
             f"loop {info['tLoop']*1e3:.2f} ms whole call {out[name][3]*1e3:.2f} ms max|x-x_oracle| {np.abs(x - xo).max():.2e}")
    x, flag, info, _ = out["ldl"]
    assert flag == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
    assert np.abs(x - xo).max() <= ABS_DEV_THR
    assert np.abs(out["cholesky"][0] - xo).max() <= ABS_DEV_THR


def test_mode_auto_follows_the_reference_rule(gpu, c_oracle):
    """SolveQuadraticProgram.jl:143-151: direct when n + m <= 5000 and the density is <= 0.4, else iterative.  Both branches are
    reached through `linearSolverMode = modeAuto` and solve to the reference tolerance."""
    P, q, A, l, u = GenerateRandomQP(ProblemClass.lassoOptimization, 10, rng=make_rng(5, 1))  # sparse, 2040 rows, density < 0.001: direct -> L D L'
    assert gpu.AutoLinearSolverMode(P, A) == gpu.LinearSolverMode.modeDirect
    info = {}
    x, flag = gpu.SolveQuadraticProgram(P, q, A, l, u, numIterations=20000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=info)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
    assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["cgIterations"] == 0 and np.abs(x - xo).max() <= ABS_DEV_THR
    Pr, qr, Ar, lr, ur = GenerateRandomQP(ProblemClass.randomQp, 200, rng=make_rng(5, 1))     # P = M'M is 99 % full: density 0.47 > 0.4 -> iterative
    assert gpu.AutoLinearSolverMode(Pr, Ar) == gpu.LinearSolverMode.modeItertaive
    P, q, A, l, u = GenerateRandomQP(ProblemClass.randomQp, 4000, numConstraints=2000, densityFctr=0.003, rng=make_rng(5, 2))   # 6000 rows: iterative
    assert gpu.AutoLinearSolverMode(P, A) == gpu.LinearSolverMode.modeItertaive
    info = {}
    x, flag = gpu.SolveQuadraticProgram(P, q, A, l, u, numIterations=4000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True, info=info)
    assert info["cgIterations"] > 0 and int(flag) in (2, 3)
    xd, fd = gpu.SolveQuadraticProgram(P, q, A, l, u, linearSolverMode=gpu.LinearSolverMode.modeDirect, numIterations=4000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
    assert int(fd) == int(flag) and np.abs(x - xd).max() <= 1e-4 * max(1.0, np.abs(xd).max())
    # dense arrays: n = 64, m = 128 has (64^2 + 128 * 64) / 192^2 = 0.33 <= 0.4 -> direct (dense Cholesky); n = 64, m = 32 has 0.67 -> iterative
    from quadraticprogramsolver_amd.generator import GenerateDenseBenchmarkQP
    Pd, qd, Ad, ld, ud = GenerateDenseBenchmarkQP(64, 128, stream=3, feasible=True)
    assert gpu.AutoLinearSolverMode(Pd, Ad) == gpu.LinearSolverMode.modeDirect
    Pd, qd, Ad, ld, ud = GenerateDenseBenchmarkQP(64, 32, stream=3, feasible=True)
    assert gpu.AutoLinearSolverMode(Pd, Ad) == gpu.LinearSolverMode.modeItertaive


def test_auto_at_the_c_abi_on_a_csc_handle(gpu, c_oracle, monkeypatch):
    """A C consumer that leaves `qps_params.linsys` at QPS_LINSYS_AUTO on a CSC handle gets the reference's modeAuto rule (SolveQuadraticProgram.jl:143-151)
    evaluated on the handle's own sizes: the direct KKT L D L' for a small sparse problem (no inner CG iterations, the direct oracle's iterates), CG beyond 5000
    rows -- and CG too when the direct factor is refused by the analysis (forced here with QPS_LDL_MAX_LEVELS = 1), while an EXPLICIT request for the direct
    plugin still fails with QPS_ERR_UNSUPPORTED."""
    from quadraticprogramsolver_amd import _lib
    P, q, A, l, u = GenerateRandomQP(ProblemClass.lassoOptimization, 10, rng=make_rng(5, 1))          # 2040 rows, density < 0.001: direct
    kw = dict(numIterations=20000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:                                    # a CSC handle ...
        prob.linsys = _lib.QPS_LINSYS_AUTO                                                            # ... driven with linsys = AUTO
        x = np.zeros(P.shape[0]); info = {}
        flag = prob.solve(x, info=info, **kw)
        assert info["cgIterations"] == 0 and int(flag) == io["convFlag"] and info["iterations"] == io["iterations"]
        assert np.abs(x - xo).max() <= ABS_DEV_THR
    monkeypatch.setenv("QPS_LDL_MAX_LEVELS", "1")                                                     # the analysis now refuses every pattern
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
        prob.linsys = _lib.QPS_LINSYS_AUTO
        x = np.zeros(P.shape[0]); info = {}
        flag = prob.solve(x, info=info, numIterations=20000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
        assert info["cgIterations"] > 0 and int(flag) in (2, 3)                                       # fell back to CG, solved all the same
        assert np.abs(x - xo).max() <= 1e-3 * max(1.0, np.abs(xo).max())
        prob.linsys = _lib.QPS_LINSYS_KKT_LDL
        with pytest.raises(_lib.QpsError) as ei:
            prob.solve(np.zeros(P.shape[0]), **kw)
        assert ei.value.status == 8                                                                   # QPS_ERR_UNSUPPORTED
    monkeypatch.delenv("QPS_LDL_MAX_LEVELS")
    P, q, A, l, u = GenerateRandomQP(ProblemClass.randomQp, 4000, numConstraints=2000, densityFctr=0.003, rng=make_rng(5, 2))   # 6000 rows: iterative
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
        prob.linsys = _lib.QPS_LINSYS_AUTO
        x = np.zeros(P.shape[0]); info = {}
        flag = prob.solve(x, info=info, numIterations=4000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
        assert info["cgIterations"] > 0 and int(flag) in (2, 3)


def test_ldl_refuses_what_it_cannot_hold(gpu):
    """A dense handle has no sparse KKT plugin; the error is explicit (no silent fall back to another plugin)."""
    from quadraticprogramsolver_amd import _lib
    import ctypes as C
    P, q, A, l, u = np.eye(8), np.zeros(8), np.ones((2, 8)), -np.ones(2), np.ones(2)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        prob.linsys = _lib.QPS_LINSYS_KKT_LDL
        with pytest.raises(gpu.QpsError) as e:
            prob.solve(np.zeros(8))
        assert e.value.status == 8


def _banded_problem(n, bw, rng):
    """SPD band matrix P (half bandwidth bw) with chain constraints x_{i+1} - x_i in [-1, 1]: a deep, narrow elimination tree."""
    diags = [rng.standard_normal(n - k) * 0.3 for k in range(1, bw + 1)]
    P = sp.diags([np.full(n, 2.0 * bw)] + diags + diags, [0] + list(range(1, bw + 1)) + [-k for k in range(1, bw + 1)], format="csc")
    A = sp.diags([-np.ones(n - 1), np.ones(n - 1)], [0, 1], shape=(n - 1, n), format="csc")
    return P, rng.standard_normal(n), A, -np.ones(n - 1), np.ones(n - 1)


@pytest.mark.parametrize("n,bw,env", [(3000, 2, {}), (3000, 2, {"QPS_LDL_MAX_TAIL": "256", "QPS_LDL_MIN_LEVEL": "8"}), (600, 5, {"QPS_LDL_MAX_TAIL": "64", "QPS_LDL_MIN_LEVEL": "2"})])
def test_ldl_deep_elimination_tree(gpu, c_oracle, monkeypatch, n, bw, env):
    """Banded P + chain constraints: the elimination tree is a few long chains, so most levels are narrow.  Default limits put them
    into the dense tail; with a small tail limit the narrow levels stay sparse (one launch per level and sweep, hundreds of levels).
    Both layouts must reproduce the oracle's iterates and solve."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)                         # read when the handle analyses its KKT matrix (first use of the plugin)
    P, q, A, l, u = _banded_problem(n, bw, make_rng(91, n + bw))
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="ldl") as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=40, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=40, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
        assert rel(x, xo) <= 1e-9
        x = np.zeros(n); info = {}
        flag = prob.solve(x, numIterations=20000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=info)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
    assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
    assert np.abs(x - xo).max() <= ABS_DEV_THR


def test_ldl_chain_problem_beyond_the_dense_tail(gpu, c_oracle):
    """n = 30 000 chain-structured problem (N = 59 999 KKT rows): the minimum-degree elimination tree is ~60 000 levels deep and no longer fits the
    dense tail -- the plugin used to refuse it.  The analysis now dissects the breadth-first line order (ldl_symbolic.cpp: line_dissection_order),
    a few hundred levels, and the iterates / the RunTests-style solve match the oracle's sparse L D L' (which orders with SuperLU's MMD)."""
    n = 30000
    P, q, A, l, u = _banded_problem(n, 2, make_rng(94, 0))
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="ldl") as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=30, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=30, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
        assert rel(x, xo) <= 1e-9
        x = np.zeros(n); info = {}
        t0 = time.perf_counter()
        flag = prob.solve(x, numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True, info=info)
        note(f"chain n={n}: flag {int(flag)} iterations {info['iterations']} refactor {info['numRefactor']} setup {info['tSetup']*1e3:.1f} ms loop {info['tLoop']*1e3:.1f} ms "
             f"({info['tLoop']/max(info['iterations'],1)*1e6:.0f} us/iteration) call {(time.perf_counter()-t0)*1e3:.1f} ms")
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=5000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True, linsys=c_oracle.KIND_KKT_LDL_SPARSE)
    assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
    assert np.abs(x - xo).max() <= ABS_DEV_THR


def test_ldl_refuses_a_tree_that_is_too_deep(gpu, monkeypatch):
    """More sparse levels than QPS_LDL_MAX_LEVELS: explicit QPS_ERR_UNSUPPORTED naming the CG plugin (no silent fallback)."""
    monkeypatch.setenv("QPS_LDL_MAX_TAIL", "64"); monkeypatch.setenv("QPS_LDL_MIN_LEVEL", "2"); monkeypatch.setenv("QPS_LDL_MAX_LEVELS", "16")
    P, q, A, l, u = _banded_problem(2000, 1, make_rng(92, 0))
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="ldl") as prob:
        with pytest.raises(gpu.QpsError) as e:
            prob.solve(np.zeros(2000), numIterations=10)
        assert e.value.status == 8 and "CG" in str(e.value)


def test_ldl_large_sparse_problems(gpu, c_oracle):
    """Beyond the reference's test sizes: portfolioOptimization n = 5000 (N = 5050, M = 5051, one dense constraint row, dense tail of 51)
    and a sparse inequalityConstrainedQp (n = 1500, m = 3000) whose fill leaves a dense tail of 1400 columns -- solution-level parity
    with the oracle's sparse L D L' (same flags, iteration and re-factorisation counts)."""
    for pc, n, m, dens in ((ProblemClass.portfolioOptimization, 5000, 0, None), (ProblemClass.inequalityConstrainedQp, 1500, 3000, 0.004)):
        P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(93, int(pc)), densityFctr=dens)
        kw = dict(numIterations=20000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True)
        xo, io = c_oracle.solve(P, q, A, l, u, linsys=c_oracle.KIND_KKT_LDL_SPARSE, **kw)
        x = np.zeros(P.shape[0]); info = {}
        t0 = time.perf_counter()
        flag = gpu.SolveQuadraticProgramInplace(x, P, q, A, l, u, gpu.HipLdlInit, gpu.HipLdl, numIterations=20000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True, info=info)
        note(f"large {pc.name} N={P.shape[0]} M={A.shape[0]}: flag {int(flag)}/{io['convFlag']} iterations {info['iterations']}/{io['iterations']} refactor {info['numRefactor']}/{io['numRefactor']} "
             f"max|x-x_oracle| {np.abs(x - xo).max():.2e}; setup {info['tSetup']*1e3:.1f} ms loop {info['tLoop']*1e3:.1f} ms call {(time.perf_counter()-t0)*1e3:.1f} ms")
        assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
        assert np.abs(x - xo).max() <= ABS_DEV_THR
