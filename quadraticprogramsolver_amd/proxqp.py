"""Host mirror of the reference's second solver form, ProxQP.jl, over the C ABI (qps_proxqp_* in include/qps.h).

    min 1/2 x'Px + q'x   s.t.  A x = b,  C x <= d

Reference (file:line under RoyiAvital/QuadraticProgramSolver):
  struct ProxQP + constructors                  ProxQP.jl:8-115
  SolveQuadraticProgram!(sQpProb; ...) -> Dict  ProxQP.jl:118-173
Field names (vX, vY, vZ, vS, dataDim, numEq, numInEq), keyword names (numIterations, ϵAbs, ϵRel, numItrConv, ρ, σ, adptΡ, τ)
and the report keys are the reference's.  All arithmetic runs on the device; this file only marshals.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import QPS_F32, QPS_F64, QpsProxQpParams, QpsProxQpReport


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class ProxQP:
    """``ProxQP(mP, vQ, mA, vB, mC, vD)`` initialises x, y from the equality-constrained KKT system, s = max(d - Cx, 0), z = 0
    (ProxQP.jl:73-93); ``ProxQP(mP, vQ, mA, vB, mC, vD, vX, vY, vZ, vS)`` takes the state explicitly (ProxQP.jl:36)."""

    def __init__(self, mP, vQ, mA, vB, mC, vD, vX=None, vY=None, vZ=None, vS=None, *, dtype="f64", device=0):
        self.dataDim, self.numEq, self.numInEq = mP.shape[0], mA.shape[0], mC.shape[0]        # ProxQP.jl:39-41
        n, me, mi = self.dataDim, self.numEq, self.numInEq
        if mP.shape != (n, n) or (me and mA.shape[1] != n) or (mi and mC.shape[1] != n):
            raise ValueError("dimension mismatch between mP, mA and mC")
        vec = lambda v, k, name: self._vec(v, k, name)
        q, b, d = vec(vQ, n, "vQ"), vec(vB, me, "vB"), vec(vD, mi, "vD")
        h = C.c_void_p()
        dt = {"f64": QPS_F64, "f32": QPS_F32}[dtype]
        if sp.issparse(mP) and sp.issparse(mA) and sp.issparse(mC):                            # SparseProxQP (ProxQP.jl:71, :95-115): CSC fields as they are
            ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
            csc = []
            for M in (mP, mA, mC):
                Mc = sp.csc_matrix(M, dtype=np.float64); Mc.sum_duplicates()
                csc.append((Mc.indptr.astype(np.int64), Mc.indices.astype(np.int64), np.ascontiguousarray(Mc.data)))
            (Pc, Pr, Pv), (Ac, Ar, Av), (Cc, Cr, Cv) = csc
            _lib.check(_lib.lib().qps_proxqp_create_csc(n, me, mi, ip(Pc), ip(Pr), _dp(Pv), _dp(q), ip(Ac), ip(Ar), _dp(Av), _dp(self._pad(b)), ip(Cc), ip(Cr),
                                                        _dp(Cv), _dp(self._pad(d)), 0, dt, device, C.byref(h)))
        else:
            dense = lambda M: np.asfortranarray(M.toarray() if sp.issparse(M) else M, dtype=np.float64)
            P, A, Cm = dense(mP), dense(mA), dense(mC)
            _lib.check(_lib.lib().qps_proxqp_create_dense(n, me, mi, _dp(P), max(n, 1), _dp(q), _dp(A), max(me, 1), _dp(b), _dp(Cm), max(mi, 1),
                                                          _dp(d), dt, device, C.byref(h)))
        self._h = h
        self.vX, self.vY, self.vZ, self.vS = np.zeros(n), np.zeros(me), np.zeros(mi), np.zeros(mi)
        if vX is None:
            _lib.check(_lib.lib().qps_proxqp_init_kkt(self._h), self._h)                      # ProxQP.jl:80-89
        else:
            x, y, z, s = vec(vX, n, "vX"), vec(vY, me, "vY"), vec(vZ, mi, "vZ"), vec(vS, mi, "vS")
            _lib.check(_lib.lib().qps_proxqp_set_state(self._h, _dp(x), _dp(self._pad(y)), _dp(self._pad(z)), _dp(self._pad(s))), self._h)
        self._pull()

    @staticmethod
    def _vec(v, k, name):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
        if a.shape[0] != k:
            raise ValueError(f"dimension mismatch: {name} has {a.shape[0]} elements, expected {k}")
        return a

    @staticmethod
    def _pad(a):
        return a if a.size else np.zeros(1)

    def _pull(self):
        y, z, s = (np.zeros(max(k, 1)) for k in (self.numEq, self.numInEq, self.numInEq))
        _lib.check(_lib.lib().qps_proxqp_get_state(self._h, _dp(self.vX), _dp(y), _dp(z), _dp(s)), self._h)
        self.vY[:], self.vZ[:], self.vS[:] = y[:self.numEq], z[:self.numInEq], s[:self.numInEq]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().qps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def SolveQuadraticProgramProxQP(sQpProb: ProxQP, *, numIterations=2000, ϵAbs=1e-7, ϵRel=1e-6, numItrConv=50, ρ=1e2, σ=1e-2, adptΡ=True, τ=10.0, loopVariant=0):
    """``SolveQuadraticProgram!(sQpProb :: ProxQP; ...)`` (ProxQP.jl:118-173): updates sQpProb.vX/vY/vZ/vS, returns the report
    dict with the reference's keys.  Like the reference it always runs ``numIterations`` iterations."""
    p = QpsProxQpParams()
    _lib.check(_lib.lib().qps_proxqp_default_params(C.byref(p)))
    p.numIterations, p.numItrConv, p.adptRho, p.loopVariant = int(numIterations), int(numItrConv), int(bool(adptΡ)), int(loopVariant)
    p.epsAbs, p.epsRel, p.rho, p.sigma, p.tau = float(ϵAbs), float(ϵRel), float(ρ), float(σ), float(τ)
    rep = QpsProxQpReport()
    _lib.check(_lib.lib().qps_proxqp_solve(sQpProb._h, C.byref(p), C.byref(rep)), sQpProb._h)
    sQpProb._pull()
    return {"Converged": bool(rep.converged), "Iterations": rep.iterations, "ρ": rep.rho, "σ": rep.sigma,
            "PrimalResidual": rep.resPrim, "DualResidual": rep.resDual}                          # ProxQP.jl:127
