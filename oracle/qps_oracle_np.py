"""numpy/LAPACK mirror of the reference ADMM QP path.  TEST INFRASTRUCTURE ONLY.

Independently written second restatement (the first is oracle/qps_oracle.c) of
  SolveQuadraticProgram.jl:14-112 (loop + CheckConvergence) and LinearSystemSolvers.jl:16-186 (plugin pairs).
It keeps the reference's *shape*: the loop takes the linear-system solver as a pair of function arguments
``(LinSysSolInit, LinSysSol)`` with the reference's positional signature, so a test can plug either a CPU pair
from this file or the GPU pair (``HipCholInit``/``HipChol`` of quadraticprogramsolver_amd) into the same loop.

PARITY UNPINNED: the Julia reference cannot run here and holds no golden vectors; see oracle/qps_oracle.c header.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import enum
import math

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


class ConvergenceFlag(enum.IntEnum):
    """SolveQuadraticProgram.jl:12"""
    convNumItr = 1
    convAdmm = 2
    convPrimDual = 3


def _norm_inf(v) -> float:
    """Julia norm(v, Inf): 0.0 for empty, NaN-propagating."""
    v = np.asarray(v)
    if v.size == 0:
        return 0.0
    return float(np.max(np.abs(v)))  # np.max propagates NaN


def _jmax(*a) -> float:
    r = a[0]
    for b in a[1:]:
        r = math.nan if (math.isnan(r) or math.isnan(b)) else max(r, b)
    return r


def _jclamp(x, lo, hi):
    """Julia clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)) (NaN passes through)."""
    return np.where(x > hi, hi, np.where(x < lo, lo, x))


def _dense(M):
    return M.toarray() if sp.issparse(M) else np.asarray(M, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------------------
# Plugin pairs (LinearSystemSolvers.jl).  tuSolver is a mutable list exactly as in the reference.
# ---------------------------------------------------------------------------------------------------------------

def KktLdlInit(vX, mP, vQ, mA, ρ, ρ1, σ, numElements, numConstraints):
    """LinearSystemSolvers.jl:16-26 (LaLdlInit / QDLdlInit / FacLdlInit are textually identical up to the
    library call).  Dense symmetric-indefinite factorisation of the quasi-definite KKT matrix."""
    n, m = numElements, numConstraints
    K = np.zeros((n + m, n + m))
    K[:n, :n] = _dense(mP) + σ * np.eye(n)
    K[n:, :n] = _dense(mA)
    K[:n, n:] = _dense(mA).T
    K[n:, n:] = -ρ1 * np.eye(m)
    hDL = sla.lu_factor(K)  # any exact factorisation of K serves; LAPACK getrf keeps this independent of the C LDLt
    vV = np.zeros(n + m)
    vXX = vV[:n]  # views, :21-22
    vZZ = vV[n:]
    return vXX, vZZ, [hDL, vV]


def KktLdl(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElements, numConstraints, changedΡ):
    """LinearSystemSolvers.jl:28-44"""
    n, m = numElements, numConstraints
    if changedΡ:  # :30-32
        tuSolver[0] = KktLdlInit(vX, mP, vQ, mA, ρ, ρ1, σ, n, m)[2][0]
    hDL, vV = tuSolver
    vXX[:] = σ * vX - vQ           # :37
    vZZ[:] = vZ - ρ1 * vY          # :38
    vV[:] = sla.lu_solve(hDL, vV)  # :39
    vZZ[:] = vZ + ρ1 * (vZZ - vY)  # :40


def RedCholInit(vX, mP, vQ, mA, ρ, ρ1, σ, numElements, numConstraints):
    """Reduced form of LinearSystemSolvers.jl:110-122 with cg! replaced by a Cholesky solve
    (ProxQP.jl:175-206 is the in-repo precedent)."""
    n, m = numElements, numConstraints
    Ad = _dense(mA)
    mAA = Ad.T @ Ad                       # :112
    mPI = _dense(mP) + σ * np.eye(n)      # :113
    mL = mPI + ρ * mAA                    # :114
    vT = np.zeros(n)
    vXX = np.zeros(n)
    vZZ = np.zeros(m)
    return vXX, vZZ, [sla.cho_factor(mL, lower=True), mPI, mAA, vT]


def RedChol(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElements, numConstraints, changedΡ):
    """LinearSystemSolvers.jl:125-142 with the cg! call (:137) replaced by cho_solve."""
    if changedΡ:  # :127-129
        tuSolver[0] = sla.cho_factor(tuSolver[1] + ρ * tuSolver[2], lower=True)
    vT = tuSolver[3]
    vZZ[:] = ρ * vZ - vY              # :134
    vT[:] = mA.T @ vZZ                # :135
    vT[:] = σ * vX - vQ + vT          # :136
    vXX[:] = sla.cho_solve(tuSolver[0], vT)
    vZZ[:] = mA @ vXX                 # :139


def _cg(x, op, b, abstol, maxiter):
    """IterativeSolvers.cg! (un-vendored, unpinned; v0.9 published algorithm), x warm-started."""
    u = np.zeros_like(x)
    r = b - op(x)
    residual = float(np.linalg.norm(r))
    prev = 1.0
    tol = max(math.sqrt(np.finfo(np.float64).eps) * residual, abstol)
    it = 0
    while not (residual <= tol) and it < maxiter:
        β = residual ** 2 / prev ** 2
        u = r + β * u
        c = op(u)
        α = residual ** 2 / float(u @ c)
        x += α * u
        r -= α * c
        prev = residual
        residual = float(np.linalg.norm(r))
        it += 1
    return it


def LinOpCgInit(vX, mP, vQ, mA, ρ, ρ1, σ, numElements, numConstraints):
    """LinearSystemSolvers.jl:145-162.  The operator reads the *current* ρ from tuSolver[2]."""
    vXX = np.zeros(numElements)
    vZZ = np.zeros(numConstraints)
    vT = np.zeros(numElements)
    return vXX, vZZ, [None, vT, {"cg_iterations": 0}]


def LinOpCg(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElements, numConstraints, changedΡ,
            ϵPcg=1e-6, numItrPcg=1000):
    """LinearSystemSolvers.jl:164-186"""
    def op(w):  # :152-157  u = P w + ρ A'(A w) + σ w
        return mP @ w + ρ * (mA.T @ (mA @ w)) + σ * w
    vT = tuSolver[1]
    vZZ[:] = ρ * vZ - vY              # :176
    vT[:] = mA.T @ vZZ                # :177
    vT[:] = σ * vX - vQ + vT          # :178
    tuSolver[2]["cg_iterations"] += _cg(vXX, op, vT, ϵPcg, numItrPcg)  # :179 (warm start: vXX holds the previous x̃)
    vZZ[:] = mA @ vXX                 # :181


# ---------------------------------------------------------------------------------------------------------------
# Loop + convergence check (SolveQuadraticProgram.jl)
# ---------------------------------------------------------------------------------------------------------------

def CheckConvergence(vX, mP, vQ, mA, vZ, vY, vXP, vZP, ρ, ρρ, adptΡ, ϵAbs, ϵRel, ϵAdmm, convFlag):
    """SolveQuadraticProgram.jl:79-112"""
    MIN_VAL_Ρ, MAX_VAL_Ρ = 1e-3, 1e6                                        # :81-82
    Ax = mA @ vX
    Px = mP @ vX
    Aty = mA.T @ vY
    normResPrim = _norm_inf(Ax - vZ)                                        # :85
    normResDual = _norm_inf(Px + vQ + Aty)                                  # :86
    maxNormPrim = _jmax(_norm_inf(Ax), _norm_inf(vZ))                       # :88
    maxNormDual = _jmax(_norm_inf(Px), _norm_inf(Aty), _norm_inf(vQ))       # :89
    if adptΡ:                                                               # :92-96
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.float64(normResPrim * maxNormDual) / np.float64(normResDual * maxNormPrim)
            ρρ = float(_jclamp(ρ * np.sqrt(ratio), MIN_VAL_Ρ, MAX_VAL_Ρ))
    epsPrim = ϵAbs + ϵRel * maxNormPrim                                     # :99
    epsDual = ϵAbs + ϵRel * maxNormDual                                     # :100
    if (normResPrim < epsPrim) and (normResDual < epsDual):                 # :102-104
        convFlag = ConvergenceFlag.convPrimDual
    if (_norm_inf(vX - vXP) <= ϵAdmm) and (_norm_inf(vZ - vZP) <= ϵAdmm):   # :105-107
        convFlag = ConvergenceFlag.convAdmm
    return ρρ, convFlag, (normResPrim, normResDual)


def SolveQuadraticProgramRefLoop(vX, mP, vQ, mA, vL, vU, LinSysSolInit, LinSysSol, *,
                                 numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=1, σ=1e-6, α=1.6, δ=1e-6, adptΡ=False,
                                 fctrΡ=5, numItrConv=25, numItrPolish=10, ϵMinres=1e-6, numItrMinres=500, info=None):
    """SolveQuadraticProgram.jl:14-76 (`SolveQuadraticProgram!`).  Mutates vX, returns the ConvergenceFlag.
    ``info`` (optional dict) receives iterations / final ρ / z / y -- additive, the reference returns only the flag."""
    numElementsX = vX.shape[0]
    numRowsA = mA.shape[0]
    ρ = float(ρ)
    ρ1 = 1 / ρ                                                               # :30
    α1 = 1 - α                                                               # :31
    convFlag = ConvergenceFlag.convNumItr                                    # :33
    ϵAdmm = min(ϵAbs, ϵRel) * 1e-2                                           # :34
    vXX, vZZ, tuSolver = LinSysSolInit(vX, mP, vQ, mA, ρ, ρ1, σ, numElementsX, numRowsA)  # :36
    vXP = np.zeros(numElementsX)                                             # :38
    vZ = np.zeros(numRowsA)                                                  # :39
    vY = np.zeros(numRowsA)                                                  # :40
    vZP = np.zeros(numRowsA)                                                 # :41
    ρρ = ρ                                                                   # :43
    nref = 0
    res = (math.nan, math.nan)
    ii = 0
    for ii in range(1, numIterations + 1):                                   # :45
        changedΡ = False
        if adptΡ and ((ρρ * fctrΡ < ρ) or (ρρ > fctrΡ * ρ)):                 # :47
            ρ = ρρ
            ρ1 = 1 / ρ
            changedΡ = True
            nref += 1
        LinSysSol(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElementsX, numRowsA, changedΡ)  # :54
        vXP[:] = vX                                                          # :56
        vX[:] = α * vXX + α1 * vX                                            # :57
        vZP[:] = vZ                                                          # :59
        vZ[:] = _jclamp(α * vZZ + α1 * vZ + ρ1 * vY, vL, vU)                 # :60
        vY[:] = vY + ρ * (α * vZZ + α1 * vZP - vZ)                           # :61
        if ii % numItrConv == 0:                                             # :63
            ρρ, convFlag, res = CheckConvergence(vX, mP, vQ, mA, vZ, vY, vXP, vZP, ρ, ρρ, adptΡ, ϵAbs, ϵRel, ϵAdmm, convFlag)
            if convFlag != ConvergenceFlag.convNumItr:                       # :66-68
                break
    if info is not None:
        info.update(iterations=ii, rho_final=ρ, rho_proposed=ρρ, n_refactor=nref, z=vZ.copy(), y=vY.copy(),
                    res_prim=res[0], res_dual=res[1], tuSolver=tuSolver)
    return convFlag                                                          # :73


def kkt_certificate(x, y, mP, vQ, mA, vL, vU):
    """Oracle-independent optimality measure: returns (primal infeasibility, dual residual, complementarity gap)
    all in the ∞-norm.  x is optimal iff all three are 0 for some y (strictly convex P ⇒ unique x*)."""
    Ax = mA @ x
    prim = _norm_inf(np.maximum(Ax - vU, 0) + np.maximum(vL - Ax, 0))
    dual = _norm_inf(mP @ x + vQ + mA.T @ y)
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    with np.errstate(invalid="ignore"):
        gu = np.where(np.isfinite(vU), yp * (vU - Ax), yp)   # infinite bound: the multiplier itself must vanish
        gl = np.where(np.isfinite(vL), ym * (Ax - vL), ym)
    comp = _jmax(_norm_inf(gu), _norm_inf(gl))
    return prim, dual, comp
