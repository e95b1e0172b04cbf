"""ItrSolCgInit / ItrSolCg! on the device (round-3 review item 6; LinearSystemSolvers.jl:110-142): cg! on the EXPLICIT reduced matrix mL = mPI + rho mAA -- mAA = mA'mA and
mPI = mP + sigma I formed once (:112-114), mL rebuilt from the cached parts on changedRho (:127-129), one product per CG iteration (:137) -- against the oracle's
restatement of that very plugin (linsys kind 2) at iterate level, against the matrix-free plugin of the same handle type, and the rule by which a plain QPS_LINSYS_CG
request takes it.  Classes: isotonic regression (the reference generator's class 9: bidiagonal mA) and a banded, control-like mA."""
import numpy as np
import pytest
import scipy.sparse as sp

from quadraticprogramsolver_amd.generator import GenerateRandomQP, GenerateSparseBenchmarkQP, ProblemClass, make_rng

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if b.size else 0.0


def banded_problem(n, m, bw, seed):
    """A control-like QP: banded mA (2 bw + 1 diagonals), diagonal-plus-tridiagonal mP, box on A x."""
    rng = np.random.default_rng(seed)
    A = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-bw, bw + 1)], list(range(-bw, bw + 1)), shape=(n, n), format="csc")[:m, :]
    T = sp.diags([0.1 * np.ones(n - 1), 1.0 + rng.random(n), 0.1 * np.ones(n - 1)], [-1, 0, 1], format="csc")
    x0 = rng.standard_normal(n)
    c = A @ x0
    return sp.csc_matrix(T), rng.standard_normal(n), sp.csc_matrix(A), c - rng.random(m), c + rng.random(m)


def problems():
    P, q, A, l, u = GenerateRandomQP(ProblemClass.isotonicRegression, 600, rng=make_rng(77, 1))
    yield "isotonicRegression n=600", (P, q, A, l, u)
    yield "banded n=1500 m=1400 bw=3", banded_problem(1500, 1400, 3, 5)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_explicit_cg_iterates_against_the_oracle_plugin(gpu, c_oracle, dtype):
    """K iterations with the inner CG driven to 1e-12 on both sides (so that the inexact solve does not separate them): x / z / y of the HIP ItrSolCg plugin vs the
    oracle's kind 2 on the identical inputs, 1e-9 (fp64) / 2e-3 (fp32: the reduced matrix itself is rounded to fp32); the same iterates from the matrix-free plugin."""
    K = 50
    for name, (P, q, A, l, u) in problems():
        n = P.shape[0]
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=0.3, linsys=c_oracle.KIND_CG_EXPLICIT, epsPcg=1e-12, numItrPcg=5000)
        eps = 1e-12 if dtype == "f64" else 1e-6
        tol = 1e-9 if dtype == "f64" else 2e-3
        res = {}
        for linsys in ("cg_explicit", "cg"):
            with gpu.QuadraticProgram(P, q, A, l, u, linsys=linsys, dtype=dtype) as prob:
                x = np.zeros(n); info = {}
                import os
                if linsys == "cg":
                    os.environ["QPS_CG_EXPLICIT"] = "0"                       # the matrix-free operator, for comparison (read per request)
                try:
                    prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, ρ=0.3, ϵPcg=eps, numItrPcg=5000, info=info)
                finally:
                    os.environ.pop("QPS_CG_EXPLICIT", None)
                z, y = prob.dual()
            assert info["cgExplicit"] == (1 if linsys == "cg_explicit" else 0), (name, linsys, info)
            assert info["iterations"] == io["iterations"] == K
            d = (rel(x, xo), rel(z, io["z"]), rel(y, io["y"]))
            assert max(d) <= tol, (name, dtype, linsys, d)
            res[linsys] = (x, info)
        assert rel(res["cg_explicit"][0], res["cg"][0]) <= tol


def test_explicit_cg_adaptive_rho_to_a_tolerance(gpu, c_oracle):
    """RunTests.jl:50-58 parameters (eps = 1e-7, rho0 = 0.1, adaptive): every rho switch rebuilds mL from the cached parts (:127-129); flag, stopping iteration and
    refactor count equal the oracle's plugin, solutions within RunTests.jl:93's 1e-5."""
    for name, (P, q, A, l, u) in problems():
        n = P.shape[0]
        kw = dict(numIterations=20000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
        xo, io = c_oracle.solve(P, q, A, l, u, linsys=c_oracle.KIND_CG_EXPLICIT, epsPcg=1e-12, numItrPcg=5000, **kw)
        with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg_explicit") as prob:
            x = np.zeros(n); info = {}
            flag = prob.solve(x, numIterations=20000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True, ϵPcg=1e-12, numItrPcg=5000, info=info)
        assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"], (name, info, io["iterations"], io["numRefactor"])
        assert io["numRefactor"] >= 1, name                                        # the rebuild path was exercised
        assert np.abs(x - xo).max() <= 1e-5


def test_plugin_pair_itrsolcg_literally(gpu, c_oracle):
    """The literal pair (LinearSystemSolvers.jl:110-142): Init, Sol! with changedRho = false / true, against the host solve of (mP + sigma I + rho mA'mA) x~ = rhs."""
    import scipy.sparse.linalg as spla
    P, q, A, l, u = banded_problem(1200, 1100, 2, 9)
    n, m = P.shape[0], A.shape[0]
    rng = np.random.default_rng(3)
    vXX, vZZ, tu = gpu.HipItrSolCgInit(np.zeros(n), P, q, A, 0.5, 2.0, 1e-6, n, m)
    for changed, rho in ((False, 0.5), (True, 40.0), (False, 40.0)):
        x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
        gpu.HipItrSolCg(tu, vXX, vZZ, x, P, q, A, z, y, rho, 1.0 / rho, 1e-6, n, m, changed, ϵPcg=1e-13, numItrPcg=5000)
        Mh = sp.csc_matrix(P + 1e-6 * sp.identity(n) + rho * (A.T @ A))
        xr = spla.spsolve(Mh, 1e-6 * x - q + A.T @ (rho * z - y))
        # cg! stops at ||r|| <= max(sqrt(eps) ||r0||, abstol) with r0 taken from the WARM start (the previous x~, another right-hand side): the relative
        # tolerance governs, so the check is on the residual -- a wrong matrix (a stale rho after changedRho, a missing entry) leaves a residual of order ||rhs||
        rhs = 1e-6 * x - q + A.T @ (rho * z - y)
        assert np.linalg.norm(Mh @ vXX - rhs) <= 1e-5 * np.linalg.norm(rhs), (changed, rho, np.linalg.norm(Mh @ vXX - rhs) / np.linalg.norm(rhs))
        assert rel(vXX, xr) <= 1e-4, (changed, rho, rel(vXX, xr))
        assert rel(vZZ, A @ vXX) <= 1e-12                                        # z~ = mA * x~ (:139)
    tu[0].close()


def test_when_a_plain_cg_request_takes_the_explicit_matrix(gpu):
    """QPS_LINSYS_CG chooses by itself: isotonic regression and banded mA -> explicit (mA'mA is cheap and mL no larger than 1.5 x what the matrix-free operator streams);
    an unstructured sparse mA with ~20 entries per row (BASELINE config 3 in miniature) -> matrix-free, where an explicit request is still honoured."""
    for name, (P, q, A, l, u) in problems():
        with gpu.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
            x = np.zeros(P.shape[0]); info = {}
            prob.solve(x, numIterations=25, ϵAbs=0.0, ϵRel=0.0, info=info)
        assert info["cgExplicit"] == 1, name
    P, q, A, l, u = GenerateSparseBenchmarkQP(3000, 6000, densityA=7e-3, seed=3)
    xs = {}
    for linsys, want in (("cg", 0), ("cg_explicit", 1)):
        with gpu.QuadraticProgram(P, q, A, l, u, linsys=linsys) as prob:
            x = np.zeros(3000); info = {}
            prob.solve(x, numIterations=10, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, ϵPcg=1e-12, numItrPcg=5000, info=info)
        assert info["cgExplicit"] == want, (linsys, info)
        xs[linsys] = x
    assert rel(xs["cg_explicit"], xs["cg"]) <= 1e-8
