"""Problem generator: drop-in for the reference test harness ``GenerateRandomQP``.

Mirrors GenerateQuadraticProgram.jl:6-115 (the nine OSQP-paper problem classes): same argument meaning
(``problemClass``, ``numElements``, ``numConstraints`` with 0 = the class default), same block structure, same
distributions, same return tuple ``(mP, vQ, mA, vL, vU)`` with sparse CSC matrices and dense vectors.

Differences, all additive:
  * Julia's global RNG stream (Random.seed!(1234), RunTests.jl:11-12) cannot be reproduced without Julia, so the
    draws come from numpy's counter-based Philox generator, seed 1234 by default (``rng=`` / ``seed=``).
  * ``densityFctr`` is an argument (the reference hard-codes 0.15 / 0.5 / 0.25, :11,:38,:94); ``dense=True`` returns
    ndarray matrices (density 1.0 is how BASELINE.json's dense configs are produced).
"""
from __future__ import annotations

import enum

import numpy as np
import scipy.sparse as sp


class ProblemClass(enum.IntEnum):
    """GenerateQuadraticProgram.jl:6"""
    randomQp = 1
    inequalityConstrainedQp = 2
    equalityConstrainedQp = 3
    optimalControl = 4
    portfolioOptimization = 5
    lassoOptimization = 6
    huberFitting = 7
    supportVectorMachine = 8
    isotonicRegression = 9


def make_rng(seed: int = 1234, stream: int = 0) -> np.random.Generator:
    """Counter-based PRNG (Philox4x64); ``stream`` selects an independent substream (problem index in a batch)."""
    return np.random.Generator(np.random.Philox(key=[seed, stream]))


def sprandn(rng: np.random.Generator, m: int, n: int, density: float) -> sp.csc_matrix:
    """Julia ``sprandn(m, n, p)``: each entry is non-zero independently with probability p, values N(0,1)."""
    total = m * n
    if total == 0 or density <= 0.0:
        return sp.csc_matrix((m, n))
    if density >= 1.0:
        return sp.csc_matrix(rng.standard_normal((m, n)))
    k = int(rng.binomial(total, density))
    pos = np.unique(rng.integers(0, total, size=int(k * 1.02) + 16, dtype=np.int64))
    while pos.size < k:  # top up after de-duplication (rare)
        pos = np.unique(np.concatenate([pos, rng.integers(0, total, size=k - pos.size + 16, dtype=np.int64)]))
    if pos.size > k:
        pos = rng.permutation(pos)[:k]
    rows, cols = pos % m, pos // m
    return sp.csc_matrix((rng.standard_normal(k), (rows, cols)), shape=(m, n))


def _eye(n, v=1.0):
    return sp.identity(n, format="csc") * v


def GenerateRandomQP(problemClass, numElements: int = 1000, *, numConstraints: int = 0, rng=None, seed: int = 1234,
                     densityFctr=None, dense: bool = False):
    """GenerateQuadraticProgram.jl:8-115.  Returns (mP, vQ, mA, vL, vU)."""
    problemClass = ProblemClass(problemClass)
    if rng is None:
        rng = make_rng(seed)
    n = int(numElements)
    PC = ProblemClass
    if problemClass in (PC.randomQp, PC.inequalityConstrainedQp, PC.equalityConstrainedQp, PC.optimalControl):
        d = 0.15 if densityFctr is None else densityFctr                       # :11
        α = 1e-2                                                               # :12
        if dense or d >= 1.0:
            mM = rng.standard_normal((n, n))
            mP = mM.T @ mM + α * np.eye(n)                                     # :15
            mP = 0.5 * (mP + mP.T)
        else:
            mM = sprandn(rng, n, n, d)                                         # :14
            mP = (mM.T @ mM + _eye(n, α)).tocsc()                              # :15
        vQ = rng.standard_normal(n)                                            # :16
        if problemClass == PC.inequalityConstrainedQp:
            m = numConstraints or 10 * n                                       # :18
            mA = _sprandn_or_dense(rng, m, n, d, dense)                        # :19
            vL = -rng.random(m)                                                # :20
            vU = rng.random(m)                                                 # :21
        elif problemClass == PC.equalityConstrainedQp:
            m = numConstraints or n // 2                                       # :23
            mA = _sprandn_or_dense(rng, m, n, d, dense)                        # :24
            vL = rng.standard_normal(m)                                        # :25
            vU = vL.copy()                                                     # :26
        else:
            m = numConstraints or n // 2                                       # :28
            mA = _sprandn_or_dense(rng, m, n, d, dense)                        # :29
            vL = -rng.random(m)                                                # :30
            vU = rng.random(m)                                                 # :31
            vI = rng.random(m) <= 0.15                                         # :32
            vL[vI] = vU[vI]                                                    # :33
            vI = rng.random(m) <= 0.15                                         # :34
            vU[vI] = 1.0  # :35 `vU[vI] .= vI[vI]` assigns `true` == 1.0 (reference quirk, kept)
    elif problemClass == PC.portfolioOptimization:
        d = 0.5 if densityFctr is None else densityFctr                        # :38
        k = numConstraints or max(5, n // 100)                                 # :40
        mD = sp.diags(rng.random(n) * np.sqrt(k), format="csc")                # :41
        mP = sp.block_diag([mD, _eye(k)], format="csc")                        # :42
        vQ = np.concatenate([rng.standard_normal(n), np.zeros(k)])             # :43
        mF = sprandn(rng, n, k, d)                                             # :44
        mA = sp.bmat([[mF.T, -_eye(k)],
                      [sp.csc_matrix(np.ones((1, n))), sp.csc_matrix((1, k))],
                      [_eye(n), sp.csc_matrix((n, k))]], format="csc")        # :45
        vL = np.concatenate([np.zeros(k), [1.0], np.zeros(n)])                 # :46
        vU = np.concatenate([np.zeros(k), [1.0], np.ones(n)])                  # :47
    elif problemClass == PC.lassoOptimization:
        d = 0.15 if densityFctr is None else densityFctr                       # :49
        m = numConstraints or n * 100                                          # :51
        mAd = sprandn(rng, m, n, d)                                            # :52
        vXX = (rng.standard_normal(n) / np.sqrt(n)) * (rng.random(n) > 0.5)    # :53
        vB = mAd @ vXX + rng.standard_normal(m)                                # :54
        λ = np.max(np.abs(mAd.T @ vB)) / 5.0                                   # :55
        mP = sp.block_diag([sp.csc_matrix((n, n)), _eye(m, 2.0), sp.csc_matrix((n, n))], format="csc")  # :57
        vQ = np.concatenate([np.zeros(n + m), λ * np.ones(n)])                 # :58
        mA = sp.bmat([[mAd, -_eye(m), sp.csc_matrix((m, n))],
                      [_eye(n), sp.csc_matrix((n, m)), -_eye(n)],
                      [_eye(n), sp.csc_matrix((n, m)), _eye(n)]], format="csc")  # :59
        vL = np.concatenate([vB, -np.inf * np.ones(n), np.zeros(n)])           # :60
        vU = np.concatenate([vB, np.zeros(n), np.inf * np.ones(n)])            # :61
    elif problemClass == PC.huberFitting:
        d = 0.15 if densityFctr is None else densityFctr                       # :63
        m = numConstraints or n * 100                                          # :65
        mAd = sprandn(rng, m, n, d)                                            # :66
        vXX = rng.standard_normal(n) / np.sqrt(n)                              # :67
        vI = rng.random(m) < 0.95                                              # :68
        vB = (mAd @ vXX) + 0.5 * vI * rng.standard_normal(m) + 10.0 * (~vI) * rng.random(m)  # :69
        mP = sp.block_diag([sp.csc_matrix((n, n)), _eye(m, 2.0), sp.csc_matrix((2 * m, 2 * m))], format="csc")  # :71
        vQ = np.concatenate([np.zeros(n + m), 2.0 * np.ones(2 * m)])           # :72
        mIm = _eye(m)                                                          # :73
        mA = sp.vstack([sp.hstack([mAd, -mIm, -mIm, mIm]),
                        sp.hstack([sp.csc_matrix((m, n + m)), mIm, sp.csc_matrix((m, m))]),
                        sp.hstack([sp.csc_matrix((m, n + m + m)), mIm])], format="csc")  # :74
        vL = np.concatenate([vB, np.zeros(2 * m)])                             # :75
        vU = np.concatenate([vB, np.inf * np.ones(2 * m)])                     # :76
    elif problemClass == PC.supportVectorMachine:
        d = 0.15 if densityFctr is None else densityFctr                       # :78
        m = numConstraints or n * 100                                          # :80
        numClassA = m // 2                                                     # :81
        m = 2 * numClassA  # the reference silently requires an even count (vB has 2*numClassA entries, :83)
        λ = 1.0                                                                # :82
        vB = np.concatenate([np.ones(numClassA), -np.ones(numClassA)])         # :83
        mAu = sprandn(rng, numClassA, n, d)                                    # :84
        mAl = sprandn(rng, numClassA, n, d)                                    # :85
        up = mAu / np.sqrt(m) + (mAu != 0).astype(np.float64) / m              # :86
        lo = mAl / np.sqrt(m) - (mAl != 0).astype(np.float64) / m
        mAd = sp.vstack([up, lo], format="csc")
        mP = sp.block_diag([_eye(n, 2.0), sp.csc_matrix((m, m))], format="csc")  # :88
        vQ = λ * np.concatenate([np.zeros(n), np.ones(m)])                     # :89
        mA = sp.bmat([[sp.diags(vB) @ mAd, -_eye(m)],
                      [sp.csc_matrix((m, n)), _eye(m)]], format="csc")         # :90
        vL = np.concatenate([-np.inf * np.ones(m), np.zeros(m)])               # :91
        vU = np.concatenate([-np.ones(m), np.inf * np.ones(m)])                # :92
    elif problemClass == PC.isotonicRegression:
        d = 0.25 if densityFctr is None else densityFctr                       # :94
        α = 1e-2                                                               # :95
        mM = sprandn(rng, n, n, d)                                             # :97
        mP = (mM.T @ mM + _eye(n, α)).tocsc()                                  # :98
        vQ = rng.standard_normal(n)                                            # :99
        o = np.ones(n - 1)
        if rng.random() >= 0.5:                                                # :101
            mA = sp.diags([o, -o], [0, 1], shape=(n - 1, n), format="csc")     # :103 non-increasing
        else:
            mA = sp.diags([-o, o], [0, 1], shape=(n - 1, n), format="csc")     # :106 non-decreasing
        vL = np.zeros(n - 1)                                                   # :108
        vU = 10.0 * np.ones(n - 1)                                             # :109
    else:  # pragma: no cover
        raise ValueError(problemClass)
    if dense:
        mP = mP.toarray() if sp.issparse(mP) else mP
        mA = mA.toarray() if sp.issparse(mA) else mA
    else:
        mP = sp.csc_matrix(mP)
        mA = sp.csc_matrix(mA)
    return mP, vQ, mA, vL, vU                                                  # :112


def _sprandn_or_dense(rng, m, n, d, dense):
    if dense or d >= 1.0:
        return rng.standard_normal((m, n))
    return sprandn(rng, m, n, d)


def GenerateDenseBenchmarkQP(numElements: int, numConstraints: int, *, seed: int = 1234, stream: int = 0,
                             feasible: bool = False):
    """BASELINE.json dense configs: ``randomQp`` of GenerateQuadraticProgram.jl:10-16,27-35 at density
    1.0, returned as dense column-major (Fortran-order) float64 arrays.

    With m = 2n the reference distribution is primal infeasible with overwhelming probability (15 % of the rows become
    equalities pinned at their *upper* bound, :32-33, on top of 2n slabs around 0), so ADMM ends by the stall test
    (convAdmm) and "time-to-eps" is undefined.  ``feasible=True`` keeps every draw but centres the bounds on A*x0 for a
    random x0 and pins the equality rows at (A*x0)_i, which makes x0 feasible; it is used for time-to-eps only."""
    rng = make_rng(seed, stream)
    mP, vQ, mA, vL, vU = GenerateRandomQP(ProblemClass.randomQp, numElements, numConstraints=numConstraints, rng=rng,
                                          densityFctr=1.0, dense=True)
    if feasible:
        x0 = rng.standard_normal(numElements) / np.sqrt(numElements)
        s = mA @ x0
        eq = vL == vU
        vL = np.where(eq, s, vL + s)
        vU = np.where(eq, s, vU + s)
    return np.asfortranarray(mP), vQ, np.asfortranarray(mA), vL, vU


def GenerateSparseBenchmarkQP(numElements: int, numConstraints: int, *, densityA: float = 1e-3, seed: int = 1234):
    """BASELINE.json sparse config: M = sprandn(n, n, sqrt(1e-3/n)) so nnz(P)/n^2 ~ 1e-3,
    A = sprandn(m, n, 1e-3), bounds as randomQp."""
    rng = make_rng(seed)
    n, m = numElements, numConstraints
    mM = sprandn(rng, n, n, float(np.sqrt(densityA / n)))
    mP = (mM.T @ mM + _eye(n, 1e-2)).tocsc()
    vQ = rng.standard_normal(n)
    mA = sprandn(rng, m, n, densityA)
    vL = -rng.random(m)
    vU = rng.random(m)
    vI = rng.random(m) <= 0.15
    vL[vI] = vU[vI]
    vI = rng.random(m) <= 0.15
    vU[vI] = 1.0
    return mP, vQ, mA, vL, vU


# ---------------------------------------------------------------------------------------------------------------------
# Problem interchange.  The reference's only fixture hook is ``QpModel.mat`` with the keys
# mP, vQ, mA, vL, vU (SolveQuadraticProgramUnitTest.m:83-85 writes it, SolveQuadraticProgramUnitTest.jl:49-54 reads it).
# Writing the same MAT-v5 container lets anyone with Julia (MAT.jl) or MATLAB run the reference on exactly the inputs
# this build was measured on, which is what closes the "parity unpinned" gap outside this pipeline.
# ---------------------------------------------------------------------------------------------------------------------
def SaveQpModel(fileName: str, mP, vQ, mA, vL, vU) -> None:
    """Write ``QpModel.mat`` (MAT v5; sparse matrices stay sparse CSC, vectors are n x 1 columns as MATLAB/Julia expect)."""
    import scipy.io as sio
    col = lambda v: np.asarray(v, dtype=np.float64).reshape(-1, 1)
    mat = lambda M: sp.csc_matrix(M, dtype=np.float64) if sp.issparse(M) else np.asarray(M, dtype=np.float64)
    sio.savemat(fileName, {"mP": mat(mP), "vQ": col(vQ), "mA": mat(mA), "vL": col(vL), "vU": col(vU)}, do_compression=True)


def LoadQpModel(fileName: str):
    """Read a ``QpModel.mat`` written by SaveQpModel, by the reference's MATLAB script or by MAT.jl."""
    import scipy.io as sio
    d = sio.loadmat(fileName)
    vec = lambda k: np.asarray(d[k], dtype=np.float64).reshape(-1)
    mat = lambda k: sp.csc_matrix(d[k]) if sp.issparse(d[k]) else np.asarray(d[k], dtype=np.float64)
    return mat("mP"), vec("vQ"), mat("mA"), vec("vL"), vec("vU")
