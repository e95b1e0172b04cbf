"""Host-side mirror of the reference interface for the ADMM QP path, driving libqps_hip.so through its C ABI.

Reference (file:line under RoyiAvital/QuadraticProgramSolver):
  SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, LinSysSolInit, LinSysSol!; kw...)   SolveQuadraticProgram.jl:14-76
  @enum LinearSolverMode / ConvergenceFlag                                          SolveQuadraticProgram.jl:11-12
  plugin pair Init / Sol!                                                            LinearSystemSolvers.jl:16-229

Python cannot spell ``!`` in an identifier, so the mutating function is ``SolveQuadraticProgramInplace`` (alias
``SolveQuadraticProgram_b``); positional order, keyword names (including the Unicode ones) and the returned enum are
the reference's.  The Julia ``ccall`` wrapper that keeps the exact spelling lives in julia/QuadraticProgramSolverHIP.jl.

Passing the sentinel pair ``(HipCholInit, HipChol)`` (dense reduced-form Cholesky) or ``(HipCgInit, HipCg)`` (CSR
matrix-free CG) routes the whole loop to the device-resident implementation.  The pairs are also callable literally
with the reference plugin signature (host vectors in, device solve, host vectors out) so that an unmodified
reference-style loop can drive the GPU linear solve.  There is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import QPS_F32, QPS_F64, QPS_LINSYS_AUTO, QPS_LINSYS_CG, QPS_LINSYS_CG_EXPLICIT, QPS_LINSYS_CHOLESKY, QPS_LINSYS_KKT_LDL, QpsInfo


class LinearSolverMode(enum.IntEnum):
    """SolveQuadraticProgram.jl:11 (the reference's spelling ``modeItertaive`` is kept)."""
    modeAuto = 1
    modeItertaive = 2
    modeDirect = 3


class ConvergenceFlag(enum.IntEnum):
    """SolveQuadraticProgram.jl:12"""
    convNumItr = 1
    convAdmm = 2
    convPrimDual = 3


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _vec(v, name, length=None):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if length is not None and a.shape[0] != length:
        raise ValueError(f"dimension mismatch: {name} has {a.shape[0]} elements, expected {length}")
    return a


def _validate_dims(numElementsX, mP, vQ, mA, vL, vU):
    """The reference Julia loop never validates (SolveQuadraticProgram.jl:19-24); the MATLAB original does
    (SolveQuadraticProgram.m:158-184) and this mirrors those checks."""
    if mP.shape[0] != mP.shape[1]:
        raise ValueError("The matrix mP must be square")
    if mP.shape[0] != numElementsX:
        raise ValueError("The matrix mP dimensions must match the vector vX")
    if np.asarray(vQ).reshape(-1).shape[0] != numElementsX:
        raise ValueError("The vector vQ dimensions must match the vector vX")
    if mA.shape[1] != numElementsX:
        raise ValueError("The number of columns of mA must match the vector vX")
    if np.asarray(vL).reshape(-1).shape[0] != mA.shape[0] or np.asarray(vU).reshape(-1).shape[0] != mA.shape[0]:
        raise ValueError("The vectors vL, vU dimensions must match the rows of mA")


class QuadraticProgram:
    """A problem resident in HBM (qps_create_dense / qps_create_csc ... qps_destroy).

    ``linsys``: "cholesky" (dense reduced form; sparse inputs are densified on the device), "cg" (CSR; CG on the reduced operator: matrix-free
    -- LinOpCg / LinMapsCg -- unless the explicit reduced matrix pays, see include/qps.h), "cg_explicit" (CSR; ItrSolCg: CG on the explicit
    reduced matrix mPI + rho mAA, LinearSystemSolvers.jl:110-142) or "ldl" (CSR, sparse L D L' of the KKT matrix: the reference's direct plugins).
    """

    def __init__(self, mP, vQ, mA, vL, vU, *, linsys="cholesky", dtype="f64", device=0):
        n = mP.shape[0]
        m = mA.shape[0]
        _validate_dims(n, mP, vQ, mA, vL, vU)
        self.n, self.m = n, m
        self.linsys = {"cholesky": QPS_LINSYS_CHOLESKY, "cg": QPS_LINSYS_CG, "cg_explicit": QPS_LINSYS_CG_EXPLICIT, "ldl": QPS_LINSYS_KKT_LDL, "auto": QPS_LINSYS_AUTO}[linsys]
        csr = linsys in ("cg", "cg_explicit", "ldl")
        dt = {"f64": QPS_F64, "f32": QPS_F32}[dtype]
        q, l, u = _vec(vQ, "vQ", n), _vec(vL, "vL", m), _vec(vU, "vU", m)
        h = C.c_void_p()
        L = _lib.lib()
        if sp.issparse(mP) or sp.issparse(mA) or csr:
            Pc = sp.csc_matrix(mP, dtype=np.float64)
            Ac = sp.csc_matrix(mA, dtype=np.float64)
            Pc.sum_duplicates()
            Ac.sum_duplicates()
            Pcp, Pri, Pnz = Pc.indptr.astype(np.int64), Pc.indices.astype(np.int64), np.ascontiguousarray(Pc.data)
            Acp, Ari, Anz = Ac.indptr.astype(np.int64), Ac.indices.astype(np.int64), np.ascontiguousarray(Ac.data)
            st = L.qps_create_csc(n, m, _ip(Pcp), _ip(Pri), _dp(Pnz), _ip(Acp), _ip(Ari), _dp(Anz), _dp(q), _dp(l), _dp(u),
                                  0, 0 if csr else 1, dt, device, C.byref(h))
        else:
            Pd = np.asfortranarray(mP, dtype=np.float64)
            Ad = np.asfortranarray(mA, dtype=np.float64)
            st = L.qps_create_dense(n, m, _dp(Pd), max(n, 1), _dp(Ad), max(m, 1), _dp(q), _dp(l), _dp(u), dt, device, C.byref(h))
        _lib.check(st)
        self._h = h

    # -- SolveQuadraticProgram! -----------------------------------------------------------------------------------
    def solve(self, vX, *, numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=1, σ=1e-6, α=1.6, δ=1e-6, adptΡ=False, fctrΡ=5,
              numItrConv=25, numItrPolish=10, ϵMinres=1e-6, numItrMinres=500, ϵPcg=1e-6, numItrPcg=1000,
              trsvBlock=0, reuseFactor=False, loopVariant=0, polish=False, info=None):
        """Mutates ``vX`` (warm start in, solution out) and returns the ConvergenceFlag.  ``polish=True`` appends the
        polishing step of SolveQuadraticProgram.m:289-325 (off by default: the Julia loop reserves its kwargs unused)."""
        if not isinstance(vX, np.ndarray) or vX.dtype != np.float64 or not vX.flags.c_contiguous or vX.shape != (self.n,):
            raise ValueError("vX must be a contiguous float64 vector of length numElements (it is updated in place)")
        p = _lib.default_params()
        p.numIterations, p.epsAbs, p.epsRel = int(numIterations), float(ϵAbs), float(ϵRel)
        p.rho, p.sigma, p.alpha, p.delta = float(ρ), float(σ), float(α), float(δ)
        p.adptRho, p.fctrRho, p.numItrConv = int(bool(adptΡ)), float(fctrΡ), int(numItrConv)
        p.numItrPolish, p.epsMinres, p.numItrMinres = int(numItrPolish), float(ϵMinres), int(numItrMinres)
        p.epsPcg, p.numItrPcg = float(ϵPcg), int(numItrPcg)
        p.linsys, p.trsvBlock, p.reuseFactor = self.linsys, int(trsvBlock), int(bool(reuseFactor))
        p.loopVariant, p.polish = int(loopVariant), int(bool(polish))
        inf = QpsInfo()
        _lib.check(_lib.lib().qps_solve(self._h, _dp(vX), C.byref(p), C.byref(inf)), self._h)
        if info is not None:
            info.update(inf.as_dict())
        return ConvergenceFlag(inf.convFlag)

    def polish(self, vX, vY, *, numItrPolish=10, δ=1e-6, ϵMinres=1e-6, numItrMinres=500):
        """The polishing step alone (SolveQuadraticProgram.m:289-325) from a primal ``vX`` (updated in place when MINRES
        converged) and a multiplier ``vY``.  Returns the report dict (flag 0 = polished, 1 = kept, -1 = did not run)."""
        if not isinstance(vX, np.ndarray) or vX.dtype != np.float64 or not vX.flags.c_contiguous or vX.shape != (self.n,):
            raise ValueError("vX must be a contiguous float64 vector of length numElements (it is updated in place)")
        y = np.ascontiguousarray(vY, dtype=np.float64)
        if y.shape != (self.m,):
            raise ValueError("vY must have length numConstraints")
        p = _lib.default_params()
        p.numItrPolish, p.delta, p.epsMinres, p.numItrMinres, p.polish = int(numItrPolish), float(δ), float(ϵMinres), int(numItrMinres), 1
        rep = _lib.QpsPolishReport()
        _lib.check(_lib.lib().qps_polish(self._h, _dp(vX), _dp(y if self.m else np.zeros(1)), C.byref(p), C.byref(rep)), self._h)
        return rep.as_dict()

    def dual(self):
        """(z, y) of the last solve (additive: the reference discards them)."""
        z = np.zeros(max(self.m, 1))
        y = np.zeros(max(self.m, 1))
        _lib.check(_lib.lib().qps_get_dual(self._h, _dp(z), _dp(y)), self._h)
        return z[:self.m], y[:self.m]

    # -- the literal plugin pair ------------------------------------------------------------------------------------
    def linsys_init(self, ρ, σ, trsvBlock=0):
        _lib.check(_lib.lib().qps_linsys_init(self._h, float(ρ), float(σ), self.linsys, int(trsvBlock)), self._h)

    def linsys_solve(self, vX, vZ, vY, ρ, σ, changedΡ, vXX, vZZ, ϵPcg=None, numItrPcg=None):
        if ϵPcg is not None or numItrPcg is not None:                       # LinOpCg!(...; ϵPcg = 1e-6, numItrPcg = 1000), LinearSystemSolvers.jl:164
            _lib.check(_lib.lib().qps_linsys_set_cg(self._h, float(1e-6 if ϵPcg is None else ϵPcg), int(1000 if numItrPcg is None else numItrPcg)), self._h)
        x, z, y = _vec(vX, "vX", self.n), _vec(vZ, "vZ", self.m), _vec(vY, "vY", self.m)
        xx = np.zeros(self.n)
        zz = np.zeros(max(self.m, 1))
        _lib.check(_lib.lib().qps_linsys_solve(self._h, _dp(x), _dp(z), _dp(y), float(ρ), float(σ), int(bool(changedΡ)),
                                               _dp(xx), _dp(zz)), self._h)
        vXX[:] = xx
        vZZ[:] = zz[:self.m]

    # -- one application of the operator's matrices (qps_operator_apply) ----------------------------------------------
    def apply(self, op, v, ρ=1.0, σ=0.0):
        """``op`` in "P", "A", "At", "PA", "reduced": mP v, mA v, mA' v, [mP; mA] v, (mP + ρ mA'mA + σ I) v (LinearSystemSolvers.jl:152-157) through the
        device kernels this handle's solves use for those products."""
        kind = {"P": _lib.QPS_OP_P, "A": _lib.QPS_OP_A, "At": _lib.QPS_OP_AT, "PA": _lib.QPS_OP_PA, "reduced": _lib.QPS_OP_REDUCED}[op]
        vin = _vec(v, "v", self.m if op == "At" else self.n)
        out = np.zeros(max({"P": self.n, "A": self.m, "At": self.n, "PA": self.n + self.m, "reduced": self.n}[op], 1))
        _lib.check(_lib.lib().qps_operator_apply(self._h, kind, _dp(vin if vin.size else np.zeros(1)), _dp(out), float(ρ), float(σ)), self._h)
        return out[:{"P": self.n, "A": self.m, "At": self.n, "PA": self.n + self.m, "reduced": self.n}[op]]

    # -- profiling ----------------------------------------------------------------------------------------------------
    def set_profiling(self, level: int):
        _lib.check(_lib.lib().qps_set_profiling(self._h, int(level)), self._h)

    def kernel_times(self):
        buf = (_lib.QpsKernelTime * 32)()
        cnt = C.c_int32(0)
        _lib.check(_lib.lib().qps_kernel_times(self._h, buf, 32, C.byref(cnt)), self._h)
        return [dict(name=buf[i].name.decode(), seconds=buf[i].seconds, launches=buf[i].launches, algo_bytes=buf[i].algo_bytes)
                for i in range(cnt.value)]

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


class QuadraticProgramBatch:
    """A batch of independent dense QPs of one shape resident in HBM (qps_create_dense_batch / qps_solve_batch):
    the per-problem loop of RunBenchmarks.jl:88-104 advanced in lock step by batched launches.  Every QP keeps its own
    rho, proposed rho, convergence flag and stopping iteration, exactly as if solved alone."""

    def __init__(self, problems, *, dtype="f64", device=0, devices=None, chunk=0):
        """``problems``: sequence of (mP, vQ, mA, vL, vU) tuples with identical shapes (dense or scipy sparse).

        ``devices`` (a list of device ordinals, one entry per worker; a device may appear more than once) selects the in-process multi-device driver
        (qps_solve_batch_multi): one host thread per entry, each taking ranges of at most ``chunk`` QPs from a shared counter (``chunk`` <= 0: one contiguous slab
        per worker).  The problem data then stays on the host and every ``solve`` builds and drops the per-range batch handles; ``dual`` is not available."""
        self.count = len(problems)
        self.devices = None if devices is None else [int(d) for d in devices]
        self.chunk = int(chunk)
        self.worker_of, self.worker_seconds = None, None
        mP0, _, mA0, _, _ = problems[0]
        self.n, self.m = mP0.shape[0], mA0.shape[0]
        dense = lambda M: np.asarray(M.toarray() if sp.issparse(M) else M, dtype=np.float64)
        for (mP, vQ, mA, vL, vU) in problems:
            _validate_dims(self.n, mP, vQ, mA, vL, vU)
            if mA.shape[0] != self.m:
                raise ValueError("all problems of a batch must have the same number of constraints")
        P = np.ascontiguousarray(np.stack([dense(p[0]).ravel(order="F") for p in problems]))
        A = np.ascontiguousarray(np.stack([dense(p[2]).ravel(order="F") for p in problems])) if self.m > 0 else np.zeros((self.count, 1))
        q = np.ascontiguousarray(np.stack([_vec(p[1], "vQ", self.n) for p in problems]))
        l = np.ascontiguousarray(np.stack([_vec(p[3], "vL", self.m) for p in problems])) if self.m > 0 else np.zeros((self.count, 1))
        u = np.ascontiguousarray(np.stack([_vec(p[4], "vU", self.m) for p in problems])) if self.m > 0 else np.zeros((self.count, 1))
        h = C.c_void_p()
        dt = {"f64": QPS_F64, "f32": QPS_F32}[dtype]
        if self.devices is not None:
            if not self.devices:
                raise ValueError("devices must list at least one worker")
            self._host = (P, A, q, l, u, dt)
            self._h = None
            return
        _lib.check(_lib.lib().qps_create_dense_batch(self.count, self.n, self.m, _dp(P), _dp(A), _dp(q), _dp(l), _dp(u), dt, device, C.byref(h)))
        self._h = h

    def solve(self, mX=None, *, numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=1, σ=1e-6, α=1.6, adptΡ=False, fctrΡ=5, numItrConv=25,
              trsvBlock=0, reuseFactor=False, polish=False, numItrPolish=10, δ=1e-6, ϵMinres=1e-6, numItrMinres=500):
        """Returns (mX [count x n], list of ConvergenceFlag, list of info dicts).  ``mX`` (optional) holds the warm starts."""
        X = np.zeros((self.count, self.n)) if mX is None else np.ascontiguousarray(mX, dtype=np.float64).copy()
        p = _lib.default_params()
        p.numIterations, p.epsAbs, p.epsRel = int(numIterations), float(ϵAbs), float(ϵRel)
        p.rho, p.sigma, p.alpha = float(ρ), float(σ), float(α)
        p.adptRho, p.fctrRho, p.numItrConv = int(bool(adptΡ)), float(fctrΡ), int(numItrConv)
        p.trsvBlock, p.reuseFactor = int(trsvBlock), int(bool(reuseFactor))
        p.polish, p.numItrPolish, p.delta, p.epsMinres, p.numItrMinres = int(bool(polish)), int(numItrPolish), float(δ), float(ϵMinres), int(numItrMinres)
        infos = (QpsInfo * self.count)()
        if self.devices is not None:
            P, A, q, l, u, dt = self._host
            W = len(self.devices)
            devs = (C.c_int32 * W)(*self.devices)
            owner = (C.c_int32 * self.count)()
            secs = (C.c_double * W)()
            _lib.check(_lib.lib().qps_solve_batch_multi(self.count, self.n, self.m, _dp(P), _dp(A), _dp(q), _dp(l), _dp(u), dt, devs, W, self.chunk, _dp(X), C.byref(p), infos,
                                                        owner, secs))
            self.worker_of, self.worker_seconds = list(owner), list(secs)
            return X, [ConvergenceFlag(i.convFlag) for i in infos], [i.as_dict() for i in infos]
        _lib.check(_lib.lib().qps_solve_batch(self._h, _dp(X), C.byref(p), infos), self._h)
        return X, [ConvergenceFlag(i.convFlag) for i in infos], [i.as_dict() for i in infos]

    def dual(self):
        """(mZ, mY) [count x m] of the last solve (additive: the reference discards them)."""
        if self.devices is not None:
            raise RuntimeError("dual() is not available in multi-device mode (the per-range handles are dropped after their solve)")
        Z = np.zeros((self.count, max(self.m, 1)))
        Y = np.zeros((self.count, max(self.m, 1)))
        if self.m > 0:
            _lib.check(_lib.lib().qps_get_dual(self._h, _dp(Z), _dp(Y)), self._h)
        return Z[:, :self.m], Y[:, :self.m]

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


# ------------------------------------------------------------------------------------------------------------------
# Plugin pairs with the reference signature (LinearSystemSolvers.jl:16,28)
# ------------------------------------------------------------------------------------------------------------------
def _make_pair(linsys: str, dtype: str = "f64"):
    def Init(vX, mP, vQ, mA, ρ, ρ1, σ, numElements, numConstraints):
        # vL/vU are not part of the plugin signature and the linear solve does not need them
        prob = QuadraticProgram(mP, vQ, mA, np.zeros(numConstraints), np.zeros(numConstraints), linsys=linsys, dtype=dtype)
        prob.linsys_init(ρ, σ)
        vXX = np.zeros(numElements)
        vZZ = np.zeros(numConstraints)
        return vXX, vZZ, [prob]

    if linsys in ("cg", "cg_explicit"):
        def Sol(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElements, numConstraints, changedΡ, *, ϵPcg=1e-6, numItrPcg=1000):
            tuSolver[0].linsys_solve(vX, vZ, vY, ρ, σ, changedΡ, vXX, vZZ, ϵPcg=ϵPcg, numItrPcg=numItrPcg)    # kwargs of LinearSystemSolvers.jl:164
    else:
        def Sol(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ1, σ, numElements, numConstraints, changedΡ):
            tuSolver[0].linsys_solve(vX, vZ, vY, ρ, σ, changedΡ, vXX, vZZ)

    Init._qps_linsys = Sol._qps_linsys = linsys
    Init._qps_dtype = Sol._qps_dtype = dtype
    return Init, Sol


HipCholInit, HipChol = _make_pair("cholesky")
HipCgInit, HipCg = _make_pair("cg")
HipItrSolCgInit, HipItrSolCg = _make_pair("cg_explicit")   # ItrSolCgInit / ItrSolCg!: cg! on the explicit reduced matrix (LinearSystemSolvers.jl:110-142)
HipLdlInit, HipLdl = _make_pair("ldl")          # sparse L D L' of the KKT matrix: LaLdl / QDLdl / FacLdl (LinearSystemSolvers.jl:16-107)
HipCholF32Init, HipCholF32 = _make_pair("cholesky", "f32")


def SolveQuadraticProgramInplace(vX, mP, vQ, mA, vL, vU, LinSysSolInit=HipCholInit, LinSysSol=HipChol, *,
                                 numIterations=5000, ϵAbs=1e-6, ϵRel=1e-6, ρ=1, σ=1e-6, α=1.6, δ=1e-6, adptΡ=False,
                                 fctrΡ=5, numItrConv=25, numItrPolish=10, ϵMinres=1e-6, numItrMinres=500, info=None,
                                 device=0, trsvBlock=0, loopVariant=0, polish=False):
    """``SolveQuadraticProgram!`` (SolveQuadraticProgram.jl:14-76): mutates ``vX``, returns the ConvergenceFlag.

    ``δ, numItrPolish, ϵMinres, numItrMinres`` are accepted and, as in the reference (:16-17, no polish), unused unless
    ``polish=True`` asks for the polishing step of the MATLAB implementation (SolveQuadraticProgram.m:289-325).
    ``info`` (optional dict) receives iterations, final ρ, residuals, timings -- additive."""
    linsys = getattr(LinSysSolInit, "_qps_linsys", None)
    if linsys is None or getattr(LinSysSol, "_qps_linsys", None) != linsys:
        raise TypeError("LinSysSolInit/LinSysSol must be one of this package's pairs (HipCholInit, HipChol) / "
                        "(HipCgInit, HipCg) / (HipItrSolCgInit, HipItrSolCg) / (HipLdlInit, HipLdl): the device-resident loop has no CPU path")
    with QuadraticProgram(mP, vQ, mA, vL, vU, linsys=linsys, dtype=LinSysSolInit._qps_dtype, device=device) as prob:
        return prob.solve(vX, numIterations=numIterations, ϵAbs=ϵAbs, ϵRel=ϵRel, ρ=ρ, σ=σ, α=α, δ=δ, adptΡ=adptΡ,
                          fctrΡ=fctrΡ, numItrConv=numItrConv, numItrPolish=numItrPolish, ϵMinres=ϵMinres,
                          numItrMinres=numItrMinres, trsvBlock=trsvBlock, loopVariant=loopVariant, polish=polish, info=info)


SolveQuadraticProgram_b = SolveQuadraticProgramInplace


def AutoLinearSolverMode(mP, mA):
    """The ``modeAuto`` rule of the reference (SolveQuadraticProgram.jl:129-130, :143-151; SolveQuadraticProgram.m:190-199),
    literally: direct when ``numRowsL = rows(P) + rows(A) <= 5000`` and ``(nnz(P) + nnz(A)) / numRowsL^2 <= 0.4``, else iterative.
    Evaluated by the library (``qps_linsys_auto``) so that every binding shares one rule.  ``nnz`` of a dense array counts its
    non-zero entries, as Julia's / MATLAB's ``nnz`` does."""
    nnz = lambda M: int(M.nnz) if sp.issparse(M) else int(np.count_nonzero(M))
    kind = _lib.lib().qps_linsys_auto(mP.shape[0], mA.shape[0], nnz(mP), nnz(mA), int(sp.issparse(mP) and sp.issparse(mA)))
    return LinearSolverMode.modeItertaive if kind == QPS_LINSYS_CG else LinearSolverMode.modeDirect


def SolveQuadraticProgram(mP, vQ, mA, vL, vU, *, linearSolverMode=LinearSolverMode.modeAuto, **kw):
    """Convenience form spelled in BASELINE.json's north_star: ``SolveQuadraticProgram(P, q, A, l, u; ...) -> (x, flag)``.

    ``linearSolverMode`` (SolveQuadraticProgram.jl:11): ``modeDirect`` = a factorisation on the device -- the sparse L D L' of the
    KKT matrix when both matrices are scipy-sparse (what the reference's direct branch does, :168-172), the dense reduced Cholesky
    otherwise; ``modeItertaive`` = matrix-free CG (:153-157); ``modeAuto`` = the reference's size / density rule (see
    ``AutoLinearSolverMode``).  Note that the rule was tuned for a CPU: it sends every problem with more than 5000 rows -- BASELINE's
    dense n = 4096, m = 8192 included -- to CG; pass ``modeDirect`` (or a plugin pair to ``SolveQuadraticProgramInplace``) to keep
    such a problem on the factorisation path."""
    mode = LinearSolverMode(linearSolverMode)
    n = mP.shape[0]
    if mode == LinearSolverMode.modeAuto:
        mode = AutoLinearSolverMode(mP, mA)
    if mode == LinearSolverMode.modeDirect:
        pair = (HipLdlInit, HipLdl) if (sp.issparse(mP) and sp.issparse(mA)) else (HipCholInit, HipChol)
    else:
        pair = (HipCgInit, HipCg)
    vX = np.zeros(n)
    flag = SolveQuadraticProgramInplace(vX, mP, vQ, mA, vL, vU, *pair, **kw)
    return vX, flag
