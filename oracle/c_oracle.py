"""ctypes binding of oracle/libqps_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("QPS_ORACLE_LIB") or os.path.join(_HERE, "libqps_oracle.so")   # QPS_ORACLE_LIB: another build of the same file (the sanitizer one)

KIND_RED_CHOL, KIND_KKT_LDL, KIND_CG_EXPLICIT, KIND_CG_MATFREE, KIND_KKT_LDL_SPARSE = 0, 1, 2, 3, 4


class OqParams(C.Structure):
    _fields_ = [("numIterations", C.c_int32), ("adptRho", C.c_int32), ("numItrConv", C.c_int32), ("linsys", C.c_int32),
                ("epsAbs", C.c_double), ("epsRel", C.c_double), ("rho", C.c_double), ("sigma", C.c_double),
                ("alpha", C.c_double), ("fctrRho", C.c_double), ("epsPcg", C.c_double), ("numItrPcg", C.c_int32),
                ("numThreads", C.c_int32), ("loopThreads", C.c_int32), ("reserved", C.c_int32)]


class OqInfo(C.Structure):
    _fields_ = [("convFlag", C.c_int32), ("iterations", C.c_int32), ("numRefactor", C.c_int32), ("cgIterations", C.c_int32),
                ("rhoFinal", C.c_double), ("rhoProposed", C.c_double), ("resPrim", C.c_double), ("resDual", C.c_double),
                ("maxNormPrim", C.c_double), ("maxNormDual", C.c_double), ("tSetup", C.c_double), ("tLoop", C.c_double)]


def available_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  A GPU box exposes every host core
    in the mask while the job's share is far smaller; OpenMP must not spawn a thread per visible core."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except Exception:
            continue
    return max(1, n)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "qps_oracle.c")
    if os.environ.get("QPS_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("OMP_NUM_THREADS", str(min(available_cores(), 16)))
        _lib = C.CDLL(_LIB_PATH)
        _lib.oq_version.restype = C.c_char_p
        _lib.oq_linsys_init_dense.restype = C.c_void_p
        _lib.oq_max_threads.restype = C.c_int32
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def make_params(numIterations=5000, epsAbs=1e-6, epsRel=1e-6, rho=1.0, sigma=1e-6, alpha=1.6, adptRho=False, fctrRho=5.0,
                numItrConv=25, linsys=KIND_RED_CHOL, epsPcg=1e-6, numItrPcg=1000, numThreads=0, loopThreads=0) -> OqParams:
    return OqParams(numIterations, int(bool(adptRho)), numItrConv, linsys, epsAbs, epsRel, float(rho), sigma, alpha,
                    float(fctrRho), epsPcg, numItrPcg, numThreads, loopThreads, 0)


def kkt_ordering(mP, mA):
    """A fill-reducing ordering of the KKT pattern for linsys kind 4, from a third-party code that shares nothing with the
    product: SuperLU's multiple-minimum-degree ordering of A' + A (scipy.sparse.linalg.splu, symmetric mode).  perm[new] = old."""
    import scipy.sparse.linalg as spla
    n, m = mP.shape[0], mA.shape[0]
    Pp = sp.csc_matrix(mP).astype(bool).astype(np.float64)
    Ap = sp.csc_matrix(mA).astype(bool).astype(np.float64)
    K = sp.bmat([[Pp + sp.eye(n), Ap.T], [Ap, sp.eye(m)]], format="csc") + (n + m) * sp.eye(n + m, format="csc")   # pattern only, diagonally dominant
    lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    return np.ascontiguousarray(np.argsort(lu.perm_c), dtype=np.int64)      # (perm_c itself is the inverse map: 240x the fill)


def solve(mP, vQ, mA, vL, vU, vX=None, perm=None, **kw):
    """Run the C restatement.  Dense (ndarray) or sparse (scipy) inputs.  Returns (x, info dict incl. z, y).
    linsys=KIND_KKT_LDL_SPARSE: ``perm`` (perm[new] = old over [x; nu]) orders the sparse L D L'; default ``kkt_ordering``."""
    n = mP.shape[0]
    m = mA.shape[0]
    keep = None
    if kw.get("linsys") == KIND_KKT_LDL_SPARSE:
        keep = np.ascontiguousarray(kkt_ordering(mP, mA) if perm is None else perm, dtype=np.int64)
        assert sorted(keep.tolist()) == list(range(n + m))
        lib().oq_set_kkt_perm(_ip(keep))
    else:
        lib().oq_set_kkt_perm(None)
    if not kw.get("numThreads"):
        kw["numThreads"] = min(available_cores(), 16)
    prm = make_params(**kw)
    info = OqInfo()
    x = np.zeros(n) if vX is None else np.array(vX, dtype=np.float64)
    q = np.ascontiguousarray(vQ, dtype=np.float64)
    l = np.ascontiguousarray(vL, dtype=np.float64)
    u = np.ascontiguousarray(vU, dtype=np.float64)
    z = np.zeros(max(m, 1))
    y = np.zeros(max(m, 1))
    if sp.issparse(mP) or sp.issparse(mA):
        Pc = sp.csc_matrix(mP)
        Ac = sp.csc_matrix(mA)
        Pc.sum_duplicates(); Ac.sum_duplicates()
        Pcp, Pri, Pnz = Pc.indptr.astype(np.int64), Pc.indices.astype(np.int64), Pc.data.astype(np.float64)
        Acp, Ari, Anz = Ac.indptr.astype(np.int64), Ac.indices.astype(np.int64), Ac.data.astype(np.float64)
        rc = lib().oq_solve_csc(C.c_int64(n), C.c_int64(m), _ip(Pcp), _ip(Pri), _dp(Pnz), _ip(Acp), _ip(Ari), _dp(Anz),
                                _dp(q), _dp(l), _dp(u), _dp(x), C.byref(prm), C.byref(info), _dp(z), _dp(y))
    else:
        Pd = np.asfortranarray(mP, dtype=np.float64)
        Ad = np.asfortranarray(mA, dtype=np.float64)
        rc = lib().oq_solve_dense(C.c_int64(n), C.c_int64(m), _dp(Pd), _dp(q), _dp(Ad), _dp(l), _dp(u), _dp(x),
                                  C.byref(prm), C.byref(info), _dp(z), _dp(y))
    d = {f: getattr(info, f) for f, _ in OqInfo._fields_}
    d["z"] = z[:m]
    d["y"] = y[:m]
    d["rc"] = rc
    return x, d


class LinSys:
    """The C plugin pair on its own (dense inputs): Init at construction, ``solve`` = Sol!."""

    def __init__(self, kind, mP, vQ, mA, rho, sigma):
        self.n, self.m = mP.shape[0], mA.shape[0]
        self._P = np.asfortranarray(mP, dtype=np.float64)
        self._A = np.asfortranarray(mA, dtype=np.float64)
        self._q = np.ascontiguousarray(vQ, dtype=np.float64)
        self._h = C.c_void_p(lib().oq_linsys_init_dense(C.c_int32(kind), C.c_int64(self.n), C.c_int64(self.m), _dp(self._P),
                                                        _dp(self._q), _dp(self._A), C.c_double(rho), C.c_double(sigma)))

    def solve(self, x, z, y, rho, sigma, changed_rho):
        xx = np.zeros(self.n)
        zz = np.zeros(max(self.m, 1))
        x = np.ascontiguousarray(x, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        lib().oq_linsys_solve_dense(self._h, _dp(x), _dp(z), _dp(y), C.c_double(rho), C.c_double(sigma),
                                    C.c_int32(int(changed_rho)), _dp(xx), _dp(zz))
        return xx, zz[:self.m]

    def __del__(self):
        try:
            if self._h:
                lib().oq_linsys_free(self._h)
                self._h = None
        except Exception:
            pass
