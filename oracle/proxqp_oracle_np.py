"""numpy restatement of the reference's second solver form, ProxQP.jl.  TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED
(the Julia reference cannot run in this pipeline and ships no vectors; see oracle/qps_oracle.c header).

    min 1/2 x'Px + q'x   s.t.  A x = b,  C x <= d

Follows ProxQP.jl statement by statement (citations are file:line under /root/reference):
  struct + inner constructor   ProxQP.jl:8-66       -> ProxQP.__init__ (10-argument form)
  dense convenience ctor       ProxQP.jl:73-93      -> ProxQP.from_problem (KKT initialisation of x, y; s = max(d - Cx, 0); z = 0)
  SolveQuadraticProgram!       ProxQP.jl:118-173    -> SolveQuadraticProgramProxQP
  UpdateM!/UpdateDecomposition ProxQP.jl:175-199    -> _update_decomposition
  CalculateRhs!, UpdateX/S/Y/Z ProxQP.jl:208-249
  CheckConvergence!            ProxQP.jl:252-298
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


def _norm_inf(v):
    v = np.asarray(v)
    return float(np.max(np.abs(v))) if v.size else 0.0


class ProxQP:
    """ProxQP.jl:8-66 (dense case)."""

    def __init__(self, mP, vQ, mA, vB, mC, vD, vX, vY, vZ, vS):
        self.mP, self.vQ, self.mA, self.vB, self.mC, self.vD = (np.asarray(a, dtype=np.float64) for a in (mP, vQ, mA, vB, mC, vD))
        self.vX, self.vY, self.vZ, self.vS = (np.array(a, dtype=np.float64) for a in (vX, vY, vZ, vS))
        self.dataDim, self.numEq, self.numInEq = self.mP.shape[0], self.mA.shape[0], self.mC.shape[0]   # :39-41
        mAA = self.mA.T @ self.mA
        mAA = 0.5 * (mAA.T + mAA)                                   # :42-43
        mCC = self.mC.T @ self.mC
        mCC = 0.5 * (mCC.T + mCC)                                   # :44-45
        self.mK = mAA + mCC                                         # :46
        self.sC = None

    @classmethod
    def from_problem(cls, mP, vQ, mA, vB, mC, vD):
        """ProxQP.jl:73-93: x, y from the equality-constrained KKT system, s = max(d - C x, 0), z = 0."""
        mP, vQ, mA, vB, mC, vD = (np.asarray(a, dtype=np.float64) for a in (mP, vQ, mA, vB, mC, vD))
        n, me = mP.shape[0], mA.shape[0]
        mK = np.block([[mP, mA.T], [mA, np.zeros((me, me))]])       # :80
        vK = np.linalg.solve(mK, np.concatenate([-vQ, vB]))         # :81-83
        vX, vY = vK[:n], vK[n:]
        vS = np.maximum(vD - mC @ vX, 0.0)                          # :88
        vZ = np.zeros(mC.shape[0])                                  # :89
        return cls(mP, vQ, mA, vB, mC, vD, vX, vY, vZ, vS)

    def _update_decomposition(self, ρ, σ):
        mM = self.mP + ρ * self.mK                                  # :178
        mM[np.diag_indices_from(mM)] += σ                           # :180
        self.sC = sla.cho_factor(mM, lower=True)                    # :196


def CheckConvergence(p: ProxQP, ϵAbs, ϵRel, ρ, adptΡ, τ):
    """ProxQP.jl:252-298"""
    MIN_VAL_Ρ, MAX_VAL_Ρ = 1e-5, 1e5                                # :255-256
    vX1 = p.mP @ p.vX                                               # :261
    vX2 = p.mA.T @ p.vY                                             # :262
    vX3 = p.mC.T @ p.vZ                                             # :263
    vBb = p.mA @ p.vX                                               # :264
    vDb = p.mC @ p.vX                                               # :265
    normResPrim = max(_norm_inf(vBb - p.vB), _norm_inf(vDb - p.vD + p.vS))            # :266
    normResDual = _norm_inf(vX1 + vX2 + vX3 + p.vQ)                                    # :267
    maxNormPrim = max(_norm_inf(vBb), _norm_inf(p.vB), _norm_inf(vDb), _norm_inf(p.vD), _norm_inf(p.vS))   # :269
    maxNormDual = max(_norm_inf(vX1), _norm_inf(vX2), _norm_inf(vX3), _norm_inf(p.vQ))                      # :270
    updatedΡ, scaleRatio = False, 1.0
    if adptΡ:                                                       # :277-286
        with np.errstate(divide="ignore", invalid="ignore"):
            resRatio = np.float64(normResPrim * maxNormDual) / np.float64(normResDual * maxNormPrim)
            if (resRatio > τ) or (1.0 / resRatio > τ):
                updatedΡ = True
                t = ρ * np.sqrt(np.sqrt(resRatio))
                ρρ = float(np.where(t > MAX_VAL_Ρ, MAX_VAL_Ρ, np.where(t < MIN_VAL_Ρ, MIN_VAL_Ρ, t)))
                scaleRatio = ρ / ρρ
                ρ = ρρ
    epsPrim = ϵAbs + ϵRel * maxNormPrim                             # :289
    epsDual = ϵAbs + ϵRel * maxNormDual                             # :290
    convFlag = bool((normResPrim < epsPrim) and (normResDual < epsDual))              # :292-294
    return convFlag, normResPrim, normResDual, ρ, scaleRatio, updatedΡ


def SolveQuadraticProgramProxQP(p: ProxQP, *, numIterations=2000, ϵAbs=1e-7, ϵRel=1e-6, numItrConv=50, ρ=1e2, σ=1e-2, adptΡ=True, τ=10.0):
    """ProxQP.jl:118-173 (`SolveQuadraticProgram!(::ProxQP)`).  Mutates p.vX/vY/vZ/vS, returns the report dict.
    Note the reference does NOT stop at convergence (the `break` is commented out, :156): it always runs numIterations."""
    dReport = {"Converged": False, "Iterations": numIterations, "ρ": ρ, "σ": σ, "PrimalResidual": np.inf, "DualResidual": np.inf}   # :127
    ρ1 = 1.0 / ρ                                                    # :129
    p._update_decomposition(ρ, σ)                                   # :131
    convFlag = False
    for ii in range(1, numIterations + 1):                          # :135
        vR = -p.vQ + σ * p.vX                                       # :211
        vBb = ρ * p.vB - p.vY                                       # :212
        vR = vR + p.mA.T @ vBb                                      # :213
        vDb = ρ * (p.vD - p.vS) - p.vZ                              # :215
        vR = vR + p.mC.T @ vDb                                      # :216
        p.vX[:] = sla.cho_solve(p.sC, vR)                           # :224
        p.vS[:] = np.maximum(p.vD - ρ1 * p.vZ - p.mC @ p.vX, 0.0)   # :230-232
        p.vY[:] = p.vY - ρ * p.vB + ρ * (p.mA @ p.vX)               # :238-239
        p.vZ[:] = np.maximum(p.vZ + ρ * (p.vS - p.vD) + ρ * (p.mC @ p.vX), 0.0)   # :246-248
        if ii % numItrConv == 0:                                    # :151
            convFlag, normResPrim, normResDual, ρ, scaleRatio, updatedΡ = CheckConvergence(p, ϵAbs, ϵRel, ρ, adptΡ, τ)   # :152
            dReport["PrimalResidual"] = normResPrim                 # :153
            dReport["DualResidual"] = normResDual
            if convFlag:
                dReport["Iterations"] = ii                          # :156 (no break)
            if updatedΡ:                                            # :159-165
                ρ1 = 1.0 / ρ
                p._update_decomposition(ρ, σ)
                dReport["ρ"] = ρ
    dReport["Converged"] = convFlag                                 # :169
    return dReport
