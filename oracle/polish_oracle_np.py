"""numpy restatement of the polishing step of the MATLAB reference.  TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED
(MATLAB cannot run in this pipeline and the reference ships no vectors; see oracle/qps_oracle.c header).

Follows SolveQuadraticProgram.m:289-325 statement by statement (citations are file:line under /root/reference):
  active sets from the sign of the multiplier              :293-294
  right-hand side  g = [-q; l(L); u(U)]                    :299
  K  = [P A_L' A_U'; A_L 0 0; A_U 0 0]                     :304
  KK = K + blkdiag(delta I, -delta I, -delta I)            :305
  iterative refinement  tt = minres(KK, g - K t, eps, itr, [], [], tt);  t += tt     :314-320
  x = t(1:n) only when the last minres call converged      :322-325

`minres` is MathWorks' function (third party, not under /root/reference, version unpinned).  It is restated here from the
published algorithm (Paige & Saunders, "Solution of sparse indefinite systems of linear equations", SIAM J. Numer. Anal.
12 (1975); the same Lanczos / Givens recurrences as the public SOL implementation): unpreconditioned, initial guess x0,
convergence test  ||r|| <= tol ||b||  on the recurrence's residual norm (MATLAB documents `tol` as the bound on
norm(b-A*x)/norm(b)), flag 0 = converged, 1 = maxit reached.  Iteration counts are therefore unpinned; the polished x is
pinned by the linear system itself (tests compare with a direct solve of K t = g).

The vector layout used here keeps the multiplier block at full length m with a 0/1 mask instead of compacting the active
rows (the product path does the same): inactive entries stay exactly 0 throughout, and inner products are unchanged by
that embedding, so the iterates equal MATLAB's compacted ones up to summation order.
"""
from __future__ import annotations

import numpy as np


def minres(matvec, b, tol, maxit, x0):
    """Paige-Saunders MINRES for symmetric (indefinite) A.  Returns (x, flag, relres, iters)."""
    x = np.array(x0, dtype=np.float64)
    bnorm = float(np.linalg.norm(b))
    if bnorm == 0.0:                                   # MATLAB: b == 0 -> x = 0, flag 0
        return np.zeros_like(x), 0, 0.0, 0
    r1 = b - matvec(x)
    y = r1.copy()
    beta1 = float(np.sqrt(r1 @ y))
    if not np.isfinite(beta1):
        return x, 1, np.nan, 0
    if beta1 <= tol * bnorm:                           # initial guess already good enough
        return x, 0, beta1 / bnorm, 0
    oldb, beta, dbar, epsln, phibar, cs, sn = 0.0, beta1, 0.0, 0.0, beta1, -1.0, 0.0
    w = np.zeros_like(x); w2 = np.zeros_like(x); r2 = r1.copy()
    flag, itn = 1, 0
    tiny = np.finfo(np.float64).eps
    for itn in range(1, maxit + 1):
        s = 1.0 / beta
        v = s * y
        y = matvec(v)
        if itn >= 2:
            y = y - (beta / oldb) * r1
        alfa = float(v @ y)
        y = y - (alfa / beta) * r2
        r1 = r2
        r2 = y
        oldb = beta
        beta = float(np.sqrt(r2 @ y))
        oldeps = epsln
        delta = cs * dbar + sn * alfa
        gbar = sn * dbar - cs * alfa
        epsln = sn * beta
        dbar = -cs * beta
        gamma = max(float(np.sqrt(gbar * gbar + beta * beta)), tiny)
        cs = gbar / gamma
        sn = beta / gamma
        phi = cs * phibar
        phibar = sn * phibar
        w1 = w2
        w2 = w
        w = (v - oldeps * w1 - delta * w2) / gamma
        x = x + phi * w
        if not np.isfinite(phibar):
            break
        if phibar <= tol * bnorm:
            flag = 0
            break
        if beta == 0.0:                                # Lanczos breakdown: exact solution reached
            flag = 0
            break
    return x, flag, phibar / bnorm, itn


def Polish(mP, vQ, mA, vL, vU, vX, vY, numPolishItr=10, paramDelta=1e-6, minresEps=1e-6, minresItr=500):
    """SolveQuadraticProgram.m:289-325.  Returns (vX_polished, minresFlag, info dict).  minresFlag = -1: never ran."""
    mP, vQ, mA, vL, vU, vX, vY = (np.asarray(a, dtype=np.float64) for a in (mP, vQ, mA, vL, vU, vX, vY))
    n, m = mP.shape[0], mA.shape[0]
    vLi = vY < 0                                                          # :293
    vUi = vY > 0                                                          # :294
    mask = (vLi | vUi).astype(np.float64)
    bound = np.where(vLi, vL, np.where(vUi, vU, 0.0))
    vG = np.concatenate([-vQ, bound])                                     # :299 (embedded at full length m)

    def K(t, delta):
        tx, tl = t[:n], t[n:] * mask
        return np.concatenate([mP @ tx + mA.T @ tl + delta * tx, mask * (mA @ tx) - delta * tl])   # :304-305

    vT = np.zeros(n + m)                                                  # :307
    vTT = np.zeros(n + m)                                                 # :308
    minresFlag, total, outer = -1, 0, 0                                   # :311
    for jj in range(numPolishItr):                                        # :314
        rhs = vG - K(vT, 0.0)
        vTT, minresFlag, relres, it = minres(lambda t: K(t, paramDelta), rhs, minresEps, minresItr, vTT)   # :315
        total += it
        outer = jj + 1
        if minresFlag:                                                    # :316-318
            break
        vT = vT + vTT                                                     # :319
    out = vX.copy()
    if minresFlag == 0:                                                   # :322-325
        out = vT[:n].copy()
    return out, minresFlag, {"numActiveLower": int(vLi.sum()), "numActiveUpper": int(vUi.sum()), "minresIterations": total,
                             "refinements": outer, "multipliers": vT[n:].copy()}
