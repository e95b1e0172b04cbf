"""GPU parity tests proper: every case calls libqps_hip.so through its C ABI and compares with the CPU oracle, the
committed golden vectors, or a size-independent property.  Tolerances:
  (i)   iterate level, fixed K, adptΡ off:        |x_gpu − x_ref|∞ ≤ 1e-9 · max(1, |x_ref|∞)      (fp64)
  (ii)  solution level, RunTests.jl:50-58 params: |x_gpu − x_ref|∞ ≤ 1e-5 and the same ConvergenceFlag
  (iii) KKT certificate recomputed on the host in fp64
  fp32 (BASELINE config 5): 1e-3 relative at iterate level."""
import numpy as np
import pytest

from conftest import load_golden
from quadraticprogramsolver_amd.generator import (GenerateDenseBenchmarkQP, GenerateRandomQP, ProblemClass, make_rng)

pytestmark = pytest.mark.gpu
ABS_DEV_THR = 1e-5   # RunTests.jl:58
REF_KW = dict(numIterations=50000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)   # RunTests.jl:50-54

GOLDEN_PROBLEMS = ["c1_randomQp_n64_m128", "c1_randomQp_feasible_n64_m128", "c1_randomQp_n64_m32",
                   "c1_equalityConstrainedQp_n64_m32", "c1_isotonicRegression_n64", "svm_n4_m40_infbounds"]


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if b.size else 0.0


@pytest.mark.parametrize("variant", [0, 2])   # 0: automatic (these sizes: whole loop in one launch); 2: multi-launch fused loop
@pytest.mark.parametrize("name", GOLDEN_PROBLEMS)
def test_golden_iterates(gpu, name, variant):
    g = load_golden(name)
    with gpu.QuadraticProgram(g["P"], g["q"], g["A"], g["l"], g["u"]) as prob:
        for K in (25, 50, 100):
            x = np.zeros(g["P"].shape[0]); info = {}
            flag = prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, loopVariant=variant, info=info)
            z, y = prob.dual()
            assert flag == gpu.ConvergenceFlag.convNumItr and info["iterations"] == K
            assert rel(x, g[f"x_K{K}"]) <= 1e-9 and rel(z, g[f"z_K{K}"]) <= 1e-9 and rel(y, g[f"y_K{K}"]) <= 1e-8


@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("name", GOLDEN_PROBLEMS)
def test_golden_solutions(gpu, name, variant):
    g = load_golden(name)
    kw = dict(REF_KW)
    if name == "c1_randomQp_n64_m128":
        kw["numIterations"] = 200
    x = np.zeros(g["P"].shape[0]); info = {}
    flag = gpu.SolveQuadraticProgramInplace(x, g["P"], g["q"], g["A"], g["l"], g["u"], gpu.HipCholInit, gpu.HipChol, info=info,
                                            loopVariant=variant, **kw)
    assert int(flag) == int(g["flag"]) and info["iterations"] == int(g["iterations"]) and info["numRefactor"] == int(g["n_refactor"])
    assert np.abs(x - g["x_final"]).max() <= ABS_DEV_THR


@pytest.mark.parametrize("n", [4, 16, 64])
@pytest.mark.parametrize("kat", ["unconstrained", "equality", "box_diag"])
def test_known_answers(gpu, kat, n):
    g = load_golden(f"kat_{kat}_n{n}")
    x, flag = gpu.SolveQuadraticProgram(g["P"], g["q"], g["A"], g["l"], g["u"], linearSolverMode=gpu.LinearSolverMode.modeDirect, **REF_KW)
    assert int(flag) in (2, 3)
    assert np.abs(x - g["x_star"]).max() <= ABS_DEV_THR


CASES = [(ProblemClass.randomQp, 100, 0, False), (ProblemClass.inequalityConstrainedQp, 100, 0, False),
         (ProblemClass.equalityConstrainedQp, 100, 50, True), (ProblemClass.optimalControl, 100, 0, False),
         (ProblemClass.portfolioOptimization, 100, 0, False), (ProblemClass.lassoOptimization, 10, 0, False),
         (ProblemClass.huberFitting, 4, 0, False), (ProblemClass.supportVectorMachine, 10, 0, False),
         (ProblemClass.isotonicRegression, 100, 0, False), (ProblemClass.randomQp, 300, 700, True),
         (ProblemClass.randomQp, 1000, 500, True), (ProblemClass.randomQp, 1100, 2300, True)]


@pytest.mark.parametrize("pc,n,m,dense", CASES)
def test_iterates_match_oracle_all_classes(gpu, c_oracle, pc, n, m, dense):
    """RunTests.jl:62-99 shape (every ProblemClass, sizes 10/100 and beyond) at iterate level against the C oracle."""
    P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(1234, 40 + int(pc)), dense=dense,
                                     densityFctr=1.0 if dense else None)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        for K, nb, variant in ((25, 0, 0), (100, 64, 0), (50, 256, 1), (75, 0, 2), (60, 0, 0)):   # 1 = unfused kernels, 2 = multi-launch
            x = np.zeros(P.shape[0]); info = {}
            prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, loopVariant=variant, info=info)
            z, y = prob.dual()
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=0.1)
            assert rel(x, xo) <= 1e-9 and rel(z, io["z"]) <= 1e-9 and rel(y, io["y"]) <= 1e-8
            assert abs(info["resPrim"] - io["resPrim"]) <= 1e-9 * max(1.0, io["resPrim"])
            assert abs(info["resDual"] - io["resDual"]) <= 1e-9 * max(1.0, io["resDual"])


@pytest.mark.parametrize("pc,n,m,dense", CASES[:9])
def test_solutions_match_oracle_and_kkt(gpu, c_oracle, np_oracle, pc, n, m, dense):
    P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(1234, 60 + int(pc)), dense=dense,
                                     densityFctr=1.0 if dense else None)
    x = np.zeros(P.shape[0]); info = {}
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        flag = prob.solve(x, info=info, **REF_KW)
        z, y = prob.dual()
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=50000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
    assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
    assert np.abs(x - xo).max() <= ABS_DEV_THR
    if int(flag) == 3:
        prim, dual, comp = np_oracle.kkt_certificate(x, y, P, q, A, l, u)
        assert prim <= 1e-5 * max(1.0, np.abs(z).max()) and dual <= 1e-4 and comp <= 1e-4


@pytest.mark.parametrize("name", ["c1_randomQp_feasible_n64_m128", "c1_isotonicRegression_n64"])
def test_fused_and_unfused_loops_agree(gpu, name):
    """The single-launch small-problem loop, the fused multi-launch loop and the literal LinearSystemSolvers.jl:134-139 kernel order give the same run,
    including the rho switches (the slabs of A'(rho z - y) are rebuilt on changedΡ)."""
    g = load_golden(name)
    out = []
    with gpu.QuadraticProgram(g["P"], g["q"], g["A"], g["l"], g["u"]) as prob:
        for variant in (0, 1, 2):
            x = np.zeros(g["P"].shape[0]); info = {}
            flag = prob.solve(x, loopVariant=variant, info=info, **REF_KW)
            out.append((x, int(flag), info["iterations"], info["numRefactor"], info["resPrim"], info["resDual"]))
    assert out[0][1:4] == out[1][1:4] == out[2][1:4] == (int(g["flag"]), int(g["iterations"]), int(g["n_refactor"]))
    for o in out[1:]:
        assert rel(out[0][0], o[0]) <= 1e-9
        assert abs(out[0][4] - o[4]) <= 1e-9 * max(1.0, o[4]) and abs(out[0][5] - o[5]) <= 1e-9 * max(1.0, o[5])


def test_plugin_pair_drives_reference_loop(gpu, np_oracle):
    """The literal plugin pair (LinearSystemSolvers.jl:16,28 signature) inside the oracle's reference-shaped loop:
    CPU loop + GPU linear solve == CPU loop + CPU linear solve, including a changedΡ re-factorisation."""
    g = load_golden("c1_randomQp_feasible_n64_m128")
    xa = np.zeros(64); ia = {}
    fa = np_oracle.SolveQuadraticProgramRefLoop(xa, g["P"], g["q"], g["A"], g["l"], g["u"], gpu.HipCholInit, gpu.HipChol,
                                                numIterations=300, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=ia)
    xb = np.zeros(64); ib = {}
    fb = np_oracle.SolveQuadraticProgramRefLoop(xb, g["P"], g["q"], g["A"], g["l"], g["u"], np_oracle.RedCholInit, np_oracle.RedChol,
                                                numIterations=300, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, info=ib)
    assert fa == fb and ia["iterations"] == ib["iterations"] and ia["n_refactor"] == ib["n_refactor"] >= 1
    assert rel(xa, xb) <= 1e-9


def test_linear_solve_in_isolation(gpu, c_oracle):
    rng = make_rng(11, 0)
    n, m = 200, 333
    M = rng.standard_normal((n, n)); P = M.T @ M + 1e-2 * np.eye(n); A = rng.standard_normal((m, n)); q = rng.standard_normal(n)
    ref = c_oracle.LinSys(c_oracle.KIND_RED_CHOL, P, q, A, 0.3, 1e-6)
    with gpu.QuadraticProgram(P, q, A, np.zeros(m), np.zeros(m)) as prob:
        prob.linsys_init(0.3, 1e-6, trsvBlock=64)
        for changed, rho in ((False, 0.3), (True, 40.0), (False, 40.0), (True, 1e-3)):
            x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
            xx, zz = np.zeros(n), np.zeros(m)
            prob.linsys_solve(x, z, y, rho, 1e-6, changed, xx, zz)
            xr, zr = ref.solve(x, z, y, rho, 1e-6, changed)
            assert rel(xx, xr) <= 1e-9 and rel(zz, zr) <= 1e-9


def test_edge_cases(gpu, c_oracle):
    rng = make_rng(5, 5)
    n = 8
    M = rng.standard_normal((n, n)); P = M.T @ M + np.eye(n); q = rng.standard_normal(n)
    # no constraints at all (empty A, l, u)
    x, flag = gpu.SolveQuadraticProgram(P, q, np.zeros((0, n)), np.zeros(0), np.zeros(0), linearSolverMode=gpu.LinearSolverMode.modeDirect, numIterations=5000, ϵAbs=1e-9, ϵRel=1e-9)
    assert np.abs(x - np.linalg.solve(P, -q)).max() <= 1e-6
    # n = 1
    x, flag = gpu.SolveQuadraticProgram(np.array([[2.0]]), np.array([-4.0]), np.array([[1.0]]), np.array([-1.0]), np.array([1.0]), linearSolverMode=gpu.LinearSolverMode.modeDirect, **REF_KW)
    assert abs(x[0] - 1.0) <= ABS_DEV_THR
    # warm start is honoured for x and z, y restart at 0 (SolveQuadraticProgram.jl:39-40)
    g = load_golden("c1_randomQp_n64_m32")
    x0 = make_rng(1, 2).standard_normal(64)
    xg = x0.copy()
    gpu.SolveQuadraticProgramInplace(xg, g["P"], g["q"], g["A"], g["l"], g["u"], numIterations=25, ϵAbs=0.0, ϵRel=0.0)
    xo, _ = c_oracle.solve(g["P"], g["q"], g["A"], g["l"], g["u"], vX=x0, numIterations=25, epsAbs=0.0, epsRel=0.0)
    assert rel(xg, xo) <= 1e-9
    # numIterations not a multiple of numItrConv, numItrConv = 1
    xg = np.zeros(64); info = {}
    gpu.SolveQuadraticProgramInplace(xg, g["P"], g["q"], g["A"], g["l"], g["u"], numIterations=37, numItrConv=1, ϵAbs=1e-3, ϵRel=1e-3, info=info)
    xo, io = c_oracle.solve(g["P"], g["q"], g["A"], g["l"], g["u"], numIterations=37, numItrConv=1, epsAbs=1e-3, epsRel=1e-3)
    assert info["iterations"] == io["iterations"] and info["convFlag"] == io["convFlag"] and rel(xg, xo) <= 1e-9


def test_error_behaviour(gpu):
    # factorisation breakdown is reported, not hidden (reference: the library throws, LinearSystemSolvers.jl:18)
    n = 70
    P = -np.eye(n); A = np.zeros((1, n)); A[0, 0] = 1.0
    with pytest.raises(gpu.QpsError) as e:
        gpu.SolveQuadraticProgram(P, np.zeros(n), A, np.array([-1.0]), np.array([1.0]), linearSolverMode=gpu.LinearSolverMode.modeDirect)
    assert e.value.status == 4 and "pivot" in str(e.value)
    with pytest.raises(gpu.QpsError) as e:
        gpu.QuadraticProgram(np.full((3, 3), np.nan), np.zeros(3), np.zeros((1, 3)), np.zeros(1), np.zeros(1))
    assert e.value.status == 3
    with pytest.raises(gpu.QpsError):
        gpu.SolveQuadraticProgram(np.eye(3), np.zeros(3), np.eye(3), -np.ones(3), np.ones(3), linearSolverMode=gpu.LinearSolverMode.modeDirect, ρ=-1.0)


def test_fp32_path(gpu, c_oracle):
    """BASELINE config 5 arithmetic: fp32 loop + re-factorisation on every check (fctrΡ = 1, numItrConv = 50)."""
    P, q, A, l, u = GenerateDenseBenchmarkQP(256, 512, stream=3, feasible=True)
    x = np.zeros(256); info = {}
    with gpu.QuadraticProgram(P, q, A, l, u, dtype="f32") as prob:
        prob.solve(x, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=50, epsAbs=0.0, epsRel=0.0, rho=0.1)
        assert rel(x, xo) <= 1e-3
        x = np.zeros(256)
        flag = prob.solve(x, numIterations=2000, ϵAbs=1e-4, ϵRel=1e-4, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=50, info=info)
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=2000, epsAbs=1e-4, epsRel=1e-4, rho=0.1, adptRho=True, fctrRho=1.0, numItrConv=50)
        assert info["numRefactor"] >= 1 and int(flag) in (2, 3)
        assert np.abs(x - xo).max() <= 1e-3 * max(1.0, np.abs(xo).max())   # fp32 1e-3 relative


def test_no_refactor_after_the_last_iteration(gpu, c_oracle):
    """numIterations a multiple of numItrConv with adptRho: when the FINAL check proposes a rho outside the band the reference
    never re-enters the loop top (SolveQuadraticProgram.jl:45-51), so no re-factorisation happens and rho stays.  All three loop
    variants (single launch / multi-launch fused / unfused) must report the oracle's numRefactor and rhoFinal."""
    g = load_golden("c1_randomQp_feasible_n64_m128")
    for K in (25, 50, 75):
        xo, io = c_oracle.solve(g["P"], g["q"], g["A"], g["l"], g["u"], numIterations=K, epsAbs=1e-12, epsRel=1e-12, rho=1e-3, adptRho=True)
        with gpu.QuadraticProgram(g["P"], g["q"], g["A"], g["l"], g["u"]) as prob:
            for variant in (0, 1, 2):
                x = np.zeros(64); info = {}
                prob.solve(x, numIterations=K, ϵAbs=1e-12, ϵRel=1e-12, ρ=1e-3, adptΡ=True, loopVariant=variant, info=info)
                assert info["iterations"] == io["iterations"] == K
                assert info["numRefactor"] == io["numRefactor"] and info["rhoFinal"] == pytest.approx(io["rhoFinal"], rel=1e-9), (K, variant, info, io)
                assert rel(x, xo) <= 1e-9
    assert io["rhoProposed"] != io["rhoFinal"]      # the last check did propose a switch that must not be applied


def test_batch_polish_with_a_count_that_does_not_divide_the_pass_grid(gpu):
    """count = 3, n > 1024: the batched pass plans 3 x 82 slabs while the polishing step's single-QP pass writes 256
    (the slab buffer must cover both).  batch + polish == single + polish for every QP."""
    cnt, n, m = 3, 1100, 2300
    probs = [GenerateDenseBenchmarkQP(n, m, stream=70 + b, feasible=True) for b in range(cnt)]
    kw = dict(numIterations=4000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, polish=True)
    with gpu.QuadraticProgramBatch(probs) as batch:
        X, flags, infos = batch.solve(**kw)
    for b in range(cnt):
        P, q, A, l, u = probs[b]
        x = np.zeros(n); info = {}
        with gpu.QuadraticProgram(P, q, A, l, u) as prob:
            flag = prob.solve(x, loopVariant=2, info=info, **kw)
        assert int(flag) == int(flags[b]) and info["iterations"] == infos[b]["iterations"]
        assert info["polishFlag"] == infos[b]["polishFlag"]
        assert rel(X[b], x) <= 1e-8


@pytest.mark.parametrize("n", [100, 130, 200, 300, 448, 1000])
def test_cholesky_steps_at_every_block_count(gpu, n):
    """The 128-column Cholesky steps (k_setup.hip: k_chol_step) over matrices of 2, 3, 4, 5, 7 and 16 blocks of 64: one step only, a last
    odd 64-column step, several steps; fp64 against the host solve of M x~ = rhs (LinearSystemSolvers.jl:117-121), fp32 at its own
    precision, and the factorisation must survive a rho switch."""
    rng = make_rng(41, n)
    m = n + 37
    G = rng.standard_normal((n, n)); P = G.T @ G / n + 1e-2 * np.eye(n); A = rng.standard_normal((m, n)); q = rng.standard_normal(n)
    for dtype, tol in (("f64", 1e-9), ("f32", 2e-3)):
        with gpu.QuadraticProgram(P, q, A, np.zeros(m), np.zeros(m), dtype=dtype) as prob:
            prob.linsys_init(0.3, 1e-6)
            for changed, rho in ((False, 0.3), (True, 25.0)):
                x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
                xx, zz = np.zeros(n), np.zeros(m)
                prob.linsys_solve(x, z, y, rho, 1e-6, changed, xx, zz)
                Mh = P + 1e-6 * np.eye(n) + rho * (A.T @ A)
                xr = np.linalg.solve(Mh, 1e-6 * x - q + A.T @ (rho * z - y))
                assert rel(xx, xr) <= tol, (n, dtype, rho, rel(xx, xr))
                assert rel(zz, A @ xr) <= tol


def test_batched_cholesky_steps(gpu, c_oracle):
    """A batch whose padded size is three blocks of 64 (a 128-column step and a last odd block, every launch carrying all QPs)."""
    cnt, n, m = 3, 150, 170
    probs = [GenerateDenseBenchmarkQP(n, m, stream=30 + b, feasible=True) for b in range(cnt)]
    with gpu.QuadraticProgramBatch(probs) as batch:
        X, flags, infos = batch.solve(numIterations=80, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        for b in range(cnt):
            xo, io = c_oracle.solve(*probs[b], numIterations=80, epsAbs=0.0, epsRel=0.0, rho=0.1)
            assert rel(X[b], xo) <= 1e-9 and infos[b]["iterations"] == 80


def test_batch_api(gpu, c_oracle):
    """BASELINE config 4 shape in miniature: a batch advanced in lock step must reproduce per-QP independent runs --
    fixed-K iterates, and (adaptive rho) per-QP flags, stopping iterations and refactor counts."""
    cnt, n, m = 5, 96, 160
    probs = [GenerateDenseBenchmarkQP(n, m, stream=10 + b, feasible=(b != 2)) for b in range(cnt)]   # QP 2: infeasible draw
    with gpu.QuadraticProgramBatch(probs) as batch:
        X, flags, infos = batch.solve(numIterations=100, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        for b in range(cnt):
            xo, io = c_oracle.solve(*probs[b], numIterations=100, epsAbs=0.0, epsRel=0.0, rho=0.1)
            assert rel(X[b], xo) <= 1e-9 and infos[b]["iterations"] == 100 and int(flags[b]) == 1
            assert abs(infos[b]["resPrim"] - io["resPrim"]) <= 1e-9 * max(1.0, io["resPrim"])
        X, flags, infos = batch.solve(numIterations=3000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)
        its = set()
        for b in range(cnt):
            xo, io = c_oracle.solve(*probs[b], numIterations=3000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
            assert int(flags[b]) == io["convFlag"] and infos[b]["iterations"] == io["iterations"], (b, infos[b], io["iterations"])
            assert infos[b]["numRefactor"] == io["numRefactor"]
            assert np.abs(X[b] - xo).max() <= ABS_DEV_THR
            its.add(io["iterations"])
        assert len(its) > 1      # the QPs really stop at different iterations
    # warm starts + the fallback path (m = 0 is outside the fused pass: independent solvers behind the same API)
    rng = make_rng(8, 8)
    free = []
    for b in range(3):
        Mx = rng.standard_normal((20, 20)); free.append((Mx.T @ Mx + np.eye(20), rng.standard_normal(20), np.zeros((0, 20)), np.zeros(0), np.zeros(0)))
    with gpu.QuadraticProgramBatch(free) as batch:
        X, flags, infos = batch.solve(numIterations=5000, ϵAbs=1e-9, ϵRel=1e-9)
        for b in range(3):
            assert np.abs(X[b] - np.linalg.solve(free[b][0], -free[b][1])).max() <= 1e-6


@pytest.mark.parametrize("n,m,dtype", [(60, 100, "f64"), (100, 60, "f64"), (64, 250, "f32")])
def test_small_batch_one_workgroup_per_qp(gpu, c_oracle, n, m, dtype):
    """Batches of small QPs take the register-resident kernel with one workgroup per QP (each QP runs its own loop; the host only
    refactors on rho switches): fixed-K iterates, and with adaptive rho the per-QP flags, stopping iterations and refactor counts,
    must equal independent runs of the oracle."""
    cnt = 9
    probs = [GenerateDenseBenchmarkQP(n, m, stream=30 + b, feasible=(b != 4)) for b in range(cnt)]   # QP 4: infeasible draw (stalls)
    tol = 1e-9 if dtype == "f64" else 2e-3
    with gpu.QuadraticProgramBatch(probs, dtype=dtype) as batch:
        X, flags, infos = batch.solve(numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        for b in range(cnt):
            xo, io = c_oracle.solve(*probs[b], numIterations=60, epsAbs=0.0, epsRel=0.0, rho=0.1)
            assert rel(X[b], xo) <= tol and infos[b]["iterations"] == 60 and int(flags[b]) == 1
            assert abs(infos[b]["resPrim"] - io["resPrim"]) <= tol * max(1.0, io["resPrim"])
        eps = 1e-7 if dtype == "f64" else 1e-4
        X, flags, infos = batch.solve(numIterations=4000, ϵAbs=eps, ϵRel=eps, ρ=0.1, adptΡ=True)
        its = set()
        for b in range(cnt):
            xo, io = c_oracle.solve(*probs[b], numIterations=4000, epsAbs=eps, epsRel=eps, rho=0.1, adptRho=True)
            assert np.abs(X[b] - xo).max() <= (ABS_DEV_THR if dtype == "f64" else 5e-3)
            if dtype == "f64":
                assert int(flags[b]) == io["convFlag"] and infos[b]["iterations"] == io["iterations"], (b, infos[b], io["iterations"])
                assert infos[b]["numRefactor"] == io["numRefactor"]
            its.add(infos[b]["iterations"])
        assert min(its) >= 25


def test_full_size_properties_c2(gpu):
    """BASELINE config 2 (n = 4096, m = 8192, fp64) through size-independent properties: the linear solve satisfies
    (P + σI + ρA'A) x~ = σx − q + A'(ρz − y) and z~ = A x~ to fp64 accuracy, the reported residuals are the true ones,
    and every block size of the triangular sweep gives the same iterates."""
    n, m = 4096, 8192
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m)
    rng = make_rng(77, 0)
    rho, sigma = 0.1, 1e-6
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        prob.linsys_init(rho, sigma, trsvBlock=2048)
        x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
        xx, zz = np.zeros(n), np.zeros(m)
        prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz)
        rhs = sigma * x - q + A.T @ (rho * z - y)
        lhs = P @ xx + sigma * xx + rho * (A.T @ (A @ xx))
        assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(rhs).max()
        assert np.abs(zz - A @ xx).max() <= 1e-11 * max(1.0, np.abs(zz).max())
        xs = []
        for nb, variant in ((512, 0), (2048, 1), (4096, 0)):
            xk = np.zeros(n); info = {}
            prob.solve(xk, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=rho, trsvBlock=nb, loopVariant=variant, info=info)
            xs.append(xk)
            zk, yk = prob.dual()
            assert abs(info["resPrim"] - np.abs(A @ xk - zk).max()) <= 1e-9 * max(1.0, info["resPrim"])
            assert abs(info["resDual"] - np.abs(P @ xk + q + A.T @ yk).max()) <= 1e-8 * max(1.0, info["resDual"])
            assert np.all(zk >= l - 1e-12) and np.all(zk <= u + 1e-12)
        assert rel(xs[0], xs[1]) <= 1e-9 and rel(xs[2], xs[1]) <= 1e-9


# ------------------------------------------------------------------------------------------------------------------
# CSR / matrix-free CG path (BASELINE config 3; LinearSystemSolvers.jl:145-186)
# ------------------------------------------------------------------------------------------------------------------
SPARSE_CASES = [(ProblemClass.randomQp, 200, 0), (ProblemClass.isotonicRegression, 300, 0),
                (ProblemClass.portfolioOptimization, 200, 0), (ProblemClass.supportVectorMachine, 10, 0),
                (ProblemClass.randomQp, 2000, 3000)]


# CSR-stream SpMV (gathers through L1/L2) / column-blocked SpMV with the x block in LDS: sliced form (k_spmv_sell, the default) / task form (k_spmv_blk)
@pytest.mark.parametrize("blocked", ["0", "1", "tasks"])
@pytest.mark.parametrize("pc,n,m", SPARSE_CASES)
def test_cg_path_iterates_match_oracle(gpu, c_oracle, monkeypatch, pc, n, m, blocked):
    """With the inner tolerance driven to 1e-13 both CG implementations solve the linear system to fp64 accuracy, so the
    ADMM iterates must agree tightly (the summation order inside the SpMVs differs)."""
    monkeypatch.setenv("QPS_SPMV_BLOCKED", "0" if blocked == "0" else "1")   # read when the handle is created
    monkeypatch.setenv("QPS_SPMV_SELL", "0" if blocked == "tasks" else "1")
    P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(1234, 80 + int(pc)))
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
        for K in (25, 50):
            x = np.zeros(P.shape[0]); info = {}
            prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, ϵPcg=1e-13, numItrPcg=5000, info=info)
            z, y = prob.dual()
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_CG_MATFREE,
                                    epsPcg=1e-13, numItrPcg=5000)
            assert info["cgIterations"] > 0
            assert rel(x, xo) <= 1e-7 and rel(z, io["z"]) <= 1e-7 and rel(y, io["y"]) <= 1e-6


@pytest.mark.parametrize("sell", ["1", "0"])           # sliced form (rows > 96 entries per block leave the slices) / task form (rows > 2048 leave the tasks)
def test_cg_path_blocked_spmv_with_long_empty_and_ragged_rows(gpu, c_oracle, monkeypatch, sell):
    """The column-blocked SpMV on a matrix built to hit every branch of its layout: rows with more entries in one column block than a task holds
    (> 2048: a dense row across the wide block, dense columns of A = long rows of A'), rows of several hundred entries, a run of empty rows longer than
    a task may hold, a narrow last column block, a row count that is no multiple of anything -- against the oracle's matrix-free CG at iterate level,
    and the plugin pair against the host's products."""
    import scipy.sparse as sp
    monkeypatch.setenv("QPS_SPMV_BLOCKED", "1")
    monkeypatch.setenv("QPS_SPMV_SELL", sell)
    rng = make_rng(91, 0)
    n, m = 8000, 5003                                      # fp64: column blocks of 7168 -> A has blocks of 7168 + 832 columns, A' one block
    A = sp.random(m, n, density=4e-3, random_state=np.random.RandomState(5), data_rvs=rng.standard_normal, format="lil")
    A[17, :] = rng.standard_normal(n) * 0.1                # dense row: long in both column blocks
    A[m - 1, 7168:] = rng.standard_normal(n - 7168) * 0.1  # long in the narrow block only (832 entries), last row of a ragged slice
    A[100:900, :] = 0.0                                    # 800 empty rows in a row (a task holds at most 512 rows)
    A[:, 4000:4003] = rng.standard_normal((m, 3)) * 0.05   # dense columns of A = long rows of A'
    A = sp.csc_matrix(A)
    M = sp.random(n, n, density=1e-3, random_state=np.random.RandomState(6), data_rvs=rng.standard_normal, format="csc")
    P = (M.T @ M + 1e-2 * sp.eye(n)).tocsc()
    q = rng.standard_normal(n); l = -rng.random(m); u = rng.random(m)
    Ac, At, Pc = sp.csr_matrix(A), sp.csr_matrix(A.T), sp.csr_matrix(P)
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=10, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, ϵPcg=1e-13, numItrPcg=5000, info=info)
        z, y = prob.dual()
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=10, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_CG_MATFREE, epsPcg=1e-13, numItrPcg=5000)
        assert rel(x, xo) <= 1e-7 and rel(z, io["z"]) <= 1e-7 and rel(y, io["y"]) <= 1e-6
        # the products themselves: z~ = A x~ of the plugin pair, and the reported residuals (A x, P x, A'y of CheckConvergence)
        prob.linsys_init(0.3, 1e-6)
        xv, zv, yv = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
        xx, zz = np.zeros(n), np.zeros(m)
        prob.linsys_solve(xv, zv, yv, 0.3, 1e-6, False, xx, zz, ϵPcg=1e-13, numItrPcg=5000)
        assert np.abs(zz - Ac @ xx).max() <= 1e-11 * max(1.0, np.abs(zz).max())
        rhs = 1e-6 * xv - q + At @ (0.3 * zv - yv)          # IterativeSolvers rule: ||r|| <= max(sqrt(eps) ||r0||, abstol) with a warm-started x0: a loose bound here
        assert np.linalg.norm(Pc @ xx + 1e-6 * xx + 0.3 * (At @ (Ac @ xx)) - rhs) <= 1e-6 * np.linalg.norm(rhs)
        xk = np.zeros(n); info = {}
        prob.solve(xk, numIterations=25, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
        zk, yk = prob.dual()
        assert abs(info["resPrim"] - np.abs(Ac @ xk - zk).max()) <= 1e-9 * max(1.0, info["resPrim"])
        assert abs(info["resDual"] - np.abs(Pc @ xk + q + At @ yk).max()) <= 1e-8 * max(1.0, info["resDual"])


@pytest.mark.parametrize("sell", ["1", "0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cg_path_column_blocked_spmv_with_several_blocks(gpu, c_oracle, monkeypatch, dtype, sell):
    """n = 9000, m = 16000: A has two column blocks, A' three (fp64: 7168 columns per block; one / two in fp32), so the partial sums per block,
    the fused sum-while-loading of A u inside the A' product and the one-launch combine are all exercised."""
    from quadraticprogramsolver_amd.generator import GenerateSparseBenchmarkQP
    monkeypatch.setenv("QPS_SPMV_BLOCKED", "1")       # (auto-selected from 200 k non-zeros per matrix; P has 169 k here)
    monkeypatch.setenv("QPS_SPMV_SELL", sell)
    n, m = 9000, 16000
    P, q, A, l, u = GenerateSparseBenchmarkQP(n, m, densityA=2e-3, seed=99)
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg", dtype=dtype) as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=20, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, ϵPcg=1e-13 if dtype == "f64" else 1e-5, numItrPcg=3000, info=info)
        z, y = prob.dual()
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20, epsAbs=0.0, epsRel=0.0, rho=0.1, linsys=c_oracle.KIND_CG_MATFREE, epsPcg=1e-13, numItrPcg=3000)
    tol = 1e-7 if dtype == "f64" else 2e-3
    assert info["cgIterations"] > 0
    assert rel(x, xo) <= tol and rel(z, io["z"]) <= tol and rel(y, io["y"]) <= 10 * tol


@pytest.mark.parametrize("pc,n,m", SPARSE_CASES[:4])
def test_cg_path_solutions(gpu, c_oracle, np_oracle, pc, n, m):
    """Reference defaults of the CG plugins (ϵPcg = 1e-6, numItrPcg = 1000, LinearSystemSolvers.jl:164): inexact inner
    solves, so parity is at solution level: same x* as the direct oracle within 1e-4 (inner abstol 1e-6 amplified by the
    conditioning of the reduced operator)."""
    P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(1234, 80 + int(pc)))
    x = np.zeros(P.shape[0]); info = {}
    flag = gpu.SolveQuadraticProgramInplace(x, P, q, A, l, u, gpu.HipCgInit, gpu.HipCg, numIterations=20000, ϵAbs=1e-6, ϵRel=1e-6,
                                            ρ=0.1, adptΡ=True, info=info)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True)
    assert int(flag) in (2, 3) and io["convFlag"] in (2, 3)
    assert np.abs(x - xo).max() <= 1e-4 * max(1.0, np.abs(xo).max())


def test_cg_plugin_pair_in_isolation(gpu, c_oracle):
    P, q, A, l, u = GenerateRandomQP(ProblemClass.randomQp, 500, numConstraints=800, rng=make_rng(4, 4))
    rng = make_rng(12, 0)
    n, m = P.shape[0], A.shape[0]
    Pd, Ad = P.toarray(), A.toarray()
    with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
        prob.linsys_init(0.5, 1e-6)
        x0 = np.zeros(n)                      # LinOpCgInit: vXX = zeros (:147); afterwards the previous x~ (warm start, :179)
        for rho in (0.5, 7.0):
            x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
            xx, zz = np.zeros(n), np.zeros(m)
            prob.linsys_solve(x, z, y, rho, 1e-6, rho != 0.5, xx, zz)
            M = Pd + 1e-6 * np.eye(n) + rho * Ad.T @ Ad
            rhs = 1e-6 * x - q + Ad.T @ (rho * z - y)
            # IterativeSolvers stopping rule: ||r|| <= max(sqrt(eps) ||r0||, abstol = 1e-6), r0 = b - M x0
            tol = max(1.4901161193847656e-08 * np.linalg.norm(rhs - M @ x0), 1e-6)
            assert np.linalg.norm(M @ xx - rhs) <= 1.1 * tol
            assert np.abs(zz - Ad @ xx).max() <= 1e-10 * max(1.0, np.abs(zz).max())
            x0 = xx.copy()


def test_run_benchmarks_driver(gpu, tmp_path):
    """RunBenchmarks.jl counterpart: 9 classes x 2 sizes x sims -> one CSV row; a second run appends under the same header;
    a different layout is refused (RunBenchmarks.jl:125-137)."""
    import csv
    from quadraticprogramsolver_amd import run_benchmarks
    f = str(tmp_path / "QPSBenchmark.csv")
    header, row = run_benchmarks.run(f, sizes=(2, 20), num_simulations=1, samples=2)
    assert len(header) == len(row) == 4 + 4 * 9 * 2
    run_benchmarks.run(f, sizes=(2, 20), num_simulations=1, samples=2)
    rows = list(csv.reader(open(f)))
    assert len(rows) == 3 and rows[0] == header
    conv = [rows[1][i] for i in range(7, len(header), 4)]
    assert conv.count("True") >= 12          # tiny random instances may be infeasible; most converge
    with pytest.raises(RuntimeError):
        run_benchmarks.run(f, sizes=(2,), num_simulations=1, samples=1)


def test_benchmark_solvers_driver(gpu, tmp_path):
    """BenchmarkSolvers.jl counterpart: every plugin pair over a size sweep, min / max / median time per (size, plugin)."""
    import csv
    from quadraticprogramsolver_amd import benchmark_solvers as bs
    assert bs.GenerateElementsVector(200, 1200, 5) == [200, 450, 700, 950, 1200]        # BenchmarkSolvers.jl:20-25,64
    assert bs.GenerateElementsVector(10, 1000, 3, logSpace=True) == [10, 100, 1000]
    f = str(tmp_path / "solvers.csv")
    vN, tR, rows = bs.run(20, 60, 0, 0, numDims=2, samples=2, csv_path=f)
    assert vN == [20, 60] and tR.shape == (2, 4, 5) and len(rows) == 8
    assert np.all(tR[:, :, 0] <= tR[:, :, 2]) and np.all(tR[:, :, 2] <= tR[:, :, 1]) and np.all(tR[:, :, 0] > 0)   # min <= median <= max
    assert np.all(tR[:, 0, 3] > 0)
    table = list(csv.reader(open(f)))
    assert len(table) == 9 and table[0][0] == "Solver"


def test_large_n_keeps_the_fused_kernels(gpu):
    """n beyond 8192 (fp64): the fused A-pass switches to its 1024-thread instantiation and the fused forward+backward sweep to its
    wide single-buffered one (NP <= 16384), with one inverted block over the whole factor (trsvBlock = 16384, ragged doubling).
    Checked through size-independent properties (linear-solve residual, reported residuals), against the unfused kernel order
    (loopVariant = 1: A read twice, blocked substitution with three sweep blocks) and against plain blocked substitution."""
    n, m = 9000, 3000
    rng = make_rng(31, 0)
    d = rng.random(n) + 0.5
    U = rng.standard_normal((n, 8)) / np.sqrt(n)
    P = np.diag(d) + U @ U.T                       # SPD, cheap to build and to multiply
    P = 0.5 * (P + P.T)
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    q = rng.standard_normal(n); l = -rng.random(m); u = rng.random(m)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        rho, sigma = 0.5, 1e-6
        prob.linsys_init(rho, sigma)
        x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
        xx, zz = np.zeros(n), np.zeros(m)
        prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz)
        rhs = sigma * x - q + A.T @ (rho * z - y)
        lhs = P @ xx + sigma * xx + rho * (A.T @ (A @ xx))
        assert np.abs(lhs - rhs).max() <= 1e-10 * max(1.0, np.abs(rhs).max())
        runs = {}
        for tag, kw in (("fused", dict()), ("unfused", dict(loopVariant=1, trsvBlock=4096)), ("substitution", dict(trsvBlock=64))):
            xk = np.zeros(n); info = {}
            prob.solve(xk, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=rho, info=info, **kw)
            runs[tag] = (xk, info)
        assert runs["fused"][1]["sweepVariant"] == 2 and runs["fused"][1]["trsvBlock"] == 16384
        assert runs["unfused"][1]["sweepVariant"] == 1 and runs["substitution"][1]["trsvBlock"] == 64
        assert rel(runs["fused"][0], runs["unfused"][0]) <= 1e-9 and rel(runs["fused"][0], runs["substitution"][0]) <= 1e-9
        xk = np.zeros(n); info = {}
        flag = prob.solve(xk, numIterations=2000, ϵAbs=1e-6, ϵRel=1e-6, ρ=rho, adptΡ=True, info=info)
        zk, yk = prob.dual()
        assert int(flag) == 3
        assert abs(info["resPrim"] - np.abs(A @ xk - zk).max()) <= 1e-9 and abs(info["resDual"] - np.abs(P @ xk + q + A.T @ yk).max()) <= 1e-8
        assert np.all(zk >= l - 1e-12) and np.all(zk <= u + 1e-12)


# ------------------------------------------------------------------------------------------------------------------
# The reference's own test loop (RunTests.jl:62-99): every ProblemClass x sizes {10, 100} x several simulations with
# numIterations = 50000, eps = 1e-7, rho = 0.1, adptRho = true and the assertion max|x_ref - x| <= 1e-5.  The
# reference solver's role (OSQP/Gurobi) is played by the CPU oracle.  Classes 6-8 grow to (n + 100 n) variables, so their
# second size is trimmed to what the dense CPU oracle can factor in a few seconds.
# ------------------------------------------------------------------------------------------------------------------
RUNTESTS_SIZES = {ProblemClass.randomQp: (10, 100), ProblemClass.inequalityConstrainedQp: (10, 100),
                  ProblemClass.equalityConstrainedQp: (10, 100), ProblemClass.optimalControl: (10, 100),
                  ProblemClass.portfolioOptimization: (10, 100), ProblemClass.lassoOptimization: (4, 10),
                  ProblemClass.huberFitting: (2, 5), ProblemClass.supportVectorMachine: (4, 10),
                  ProblemClass.isotonicRegression: (10, 100)}


@pytest.mark.parametrize("pc", list(ProblemClass))
def test_runtests_sweep(gpu, c_oracle, np_oracle, pc):
    checked = 0
    for sim in range(10):                                                       # RunTests.jl:29 numSimulations = 10
        for n in RUNTESTS_SIZES[pc]:
            m = (n // 2) if pc == ProblemClass.equalityConstrainedQp else 0     # RunTests.jl:39-47
            P, q, A, l, u = GenerateRandomQP(pc, n, numConstraints=m, rng=make_rng(4321, 1000 * int(pc) + 10 * sim + (n > 10)))
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=50000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
            if io["convFlag"] == 1 or (io["convFlag"] == 2 and io["resPrim"] > 1e-4):
                continue   # infeasible tiny draw (empty rows in A): ends by numIterations or by the stall test with rho at
                           # its clamp, where the stopping iteration is roundoff-sensitive -- not a parity case
            x = np.zeros(P.shape[0]); info = {}
            with gpu.QuadraticProgram(P, q, A, l, u) as prob:
                flag = prob.solve(x, info=info, **REF_KW)                       # RunTests.jl:85
                z, y = prob.dual()
            assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"], (pc, n, sim, info, io["iterations"])
            assert np.abs(x - xo).max() <= ABS_DEV_THR                          # RunTests.jl:93
            if int(flag) == 3:
                prim, dual, comp = np_oracle.kkt_certificate(x, y, P, q, A, l, u)
                assert prim <= 1e-5 * max(1.0, np.abs(z).max()) and dual <= 1e-4 and comp <= 1e-4
            checked += 1
    assert checked >= 10


def test_pure_c_consumer_of_the_abi(gpu, tmp_path):
    """The boundary is a C ABI: a C program built with gcc against include/qps.h and linked to libqps_hip.so (no Python,
    no HIP headers) solves a known-answer problem, drives the plugin pair and sees the factorisation error code."""
    import os, subprocess
    from quadraticprogramsolver_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "capi_example")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "capi", "example.c"),
                           "-o", exe, "-L", libdir, "-lqps_hip", "-lm", f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "C ABI example OK" in out.stdout, out.stdout


def test_concurrent_handles_from_two_host_threads(gpu, c_oracle):
    """Threading contract of include/qps.h: distinct handles (each with its own stream) may be driven from distinct host
    threads at the same time.  ctypes releases the GIL inside the foreign call, so the two solves really overlap."""
    import threading
    cases = [GenerateDenseBenchmarkQP(300, 500, stream=50, feasible=True), GenerateDenseBenchmarkQP(700, 900, stream=51, feasible=True)]
    results = [None, None]

    def work(k):
        P, q, A, l, u = cases[k]
        with gpu.QuadraticProgram(P, q, A, l, u) as prob:
            outs = []
            for rep in range(3):
                x = np.zeros(P.shape[0]); info = {}
                flag = prob.solve(x, info=info, **REF_KW)
                outs.append((x, int(flag), info["iterations"]))
            results[k] = outs

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for k in range(2):
        assert results[k] is not None
        xo, io = c_oracle.solve(*cases[k], numIterations=50000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
        for x, flag, its in results[k]:
            assert flag == io["convFlag"] and its == io["iterations"] and np.abs(x - xo).max() <= ABS_DEV_THR


@pytest.mark.parametrize("dtype,n,m", [("f64", 10, 5), ("f64", 64, 128), ("f64", 100, 50), ("f64", 100, 100), ("f64", 128, 128),
                                         ("f32", 64, 256), ("f32", 100, 200), ("f32", 128, 128)])
def test_register_resident_single_launch_kernel(gpu, c_oracle, dtype, n, m):
    """Every instantiation of the register-resident single-launch kernel (n <= 128; one or two column blocks, one to four row
    blocks) against the oracle: iterates and residuals after fixed K, and a full adaptive-rho solve (the kernel returns to the
    host for every rho switch and is relaunched) with the same flag, iteration and refactor counts."""
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=21, feasible=True)
    tol = 1e-9 if dtype == "f64" else 2e-3
    with gpu.QuadraticProgram(P, q, A, l, u, dtype=dtype) as prob:
        for K in (25, 60):
            x = np.zeros(n); info = {}
            prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, info=info)
            z, y = prob.dual()
            xo, io = c_oracle.solve(P, q, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=0.1)
            assert rel(x, xo) <= tol and rel(z, io["z"]) <= tol and rel(y, io["y"]) <= 10 * tol
            assert abs(info["resPrim"] - io["resPrim"]) <= tol * max(1.0, io["resPrim"])
            assert abs(info["resDual"] - io["resDual"]) <= 10 * tol * max(1.0, io["resDual"])
        x = np.zeros(n); info = {}
        flag = prob.solve(x, numIterations=20000, ϵAbs=1e-7 if dtype == "f64" else 1e-4, ϵRel=1e-7 if dtype == "f64" else 1e-4, ρ=0.1, adptΡ=True, info=info)
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=20000, epsAbs=1e-7 if dtype == "f64" else 1e-4, epsRel=1e-7 if dtype == "f64" else 1e-4, rho=0.1, adptRho=True)
        assert rel(x, xo) <= (1e-6 if dtype == "f64" else 5e-3)
        if dtype == "f64":
            assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]


KNOBS = [{"QPS_GRAPH": "0"}, {"QPS_SWEEP_MODE": "0"}, {"QPS_PASS_THREADS": "1024"}, {"QPS_SWEEP_RB": "4"}, {"QPS_SWEEP_WGS": "128", "QPS_PASS_WGS": "128"},
         {"QPS_SMALL_REG": "0"}, {"QPS_SMALL_REG": "0", "QPS_SMALL_LDSMAT": "0"}, {"QPS_SMALL_REG": "0", "QPS_SMALL_THREADS": "256"}, {"QPS_SPMV_BLOCKED": "1", "QPS_SPMV_FUSEPA": "0", "QPS_SPMV_WGS": "96"},
         {"QPS_CHOL_STEP": "64"}, {"QPS_CHOL_FUSED": "0"}, {"QPS_CHOL_AVOID": "1"}, {"QPS_SWEEP_WAVE": "0"}, {"QPS_GEMM_PAIR": "0", "QPS_GEMM_LOWER_MAP": "0"}]


@pytest.mark.parametrize("knob", KNOBS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_tuning_knobs_do_not_change_results(gpu, knob, tmp_path):
    """Every QPS_* environment knob only selects a launch geometry / kernel variant: a child process with the knob set must
    reproduce the iterates of the default configuration (the knobs are read once per process, hence the subprocess)."""
    import json, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent('''
        import sys, json, numpy as np
        sys.path.insert(0, sys.argv[1])
        import quadraticprogramsolver_amd as q
        out = {}
        for tag, (n, m) in {"small": (64, 128), "mid": (1100, 2300), "narrow": (1000, 700)}.items():
            P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=7, feasible=True)
            x = np.zeros(n); q.SolveQuadraticProgramInplace(x, P, qq, A, l, u, numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
            out[tag] = x.tolist()
        # n = 4096: the tile grids of the setup GEMMs are large enough for the mirror-tile pairing and the XCD-aware order of the lower tiles
        P, qq, A, l, u = q.GenerateDenseBenchmarkQP(4096, 256, stream=11, feasible=True)
        x = np.zeros(4096); q.SolveQuadraticProgramInplace(x, P, qq, A, l, u, numIterations=20, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        out["wide"] = x.tolist()
        Ps, qs, As, ls, us = q.GenerateSparseBenchmarkQP(3000, 5000, densityA=4e-3, seed=5)
        x = np.zeros(3000)
        with q.QuadraticProgram(Ps, qs, As, ls, us, linsys="cg") as prob:
            prob.solve(x, numIterations=15, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, ϵPcg=1e-12, numItrPcg=3000)
        out["cg"] = x.tolist()
        Pl, ql, Al, ll, ul = q.GenerateRandomQP(q.ProblemClass.lassoOptimization, 10, rng=q.make_rng(9, 9))
        x = np.zeros(Pl.shape[0])
        with q.QuadraticProgram(Pl, ql, Al, ll, ul, linsys="ldl") as prob:
            prob.solve(x, numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        out["ldl"] = x.tolist()
        print(json.dumps(out))
    ''')
    f = tmp_path / "knob.py"; f.write_text(script)

    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("QPS_")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, str(f), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return {k: np.array(v) for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}

    global _KNOB_BASE
    try:
        base = _KNOB_BASE
    except NameError:
        base = _KNOB_BASE = run({})
    got = run(knob)
    for tag in base:
        assert np.abs(got[tag] - base[tag]).max() <= 1e-9 * max(1.0, np.abs(base[tag]).max()), (knob, tag)
