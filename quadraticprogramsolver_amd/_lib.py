"""ctypes binding of libqps_hip.so (C ABI declared in include/qps.h).

The library is the product: there is no Python/CPU fallback.  If the shared object is missing or no MI355X is
visible, every call fails loudly (``QpsLibraryError`` / ``QpsError``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libqps_hip.so")
CSRC = os.path.join(_PKG, "csrc")

# every symbol include/qps.h declares (tests check that the library exports exactly these)
EXPORTED_SYMBOLS = [
    "qps_default_params", "qps_device_count", "qps_create_dense", "qps_create_csc", "qps_solve", "qps_get_dual",
    "qps_linsys_init", "qps_linsys_solve", "qps_create_dense_batch", "qps_solve_batch", "qps_kernel_times",
    "qps_set_profiling", "qps_destroy", "qps_last_error", "qps_version",
    "qps_proxqp_default_params", "qps_proxqp_create_dense", "qps_proxqp_init_kkt", "qps_proxqp_set_state", "qps_proxqp_get_state",
    "qps_proxqp_solve", "qps_polish", "qps_linsys_auto", "qps_ldl_analyze", "qps_proxqp_create_csc", "qps_linsys_set_cg", "qps_operator_apply", "qps_solve_batch_multi",
]

QPS_OK = 0
STATUS_NAMES = {0: "QPS_OK", 1: "QPS_ERR_BAD_ARGUMENT", 2: "QPS_ERR_BAD_DIMENSION", 3: "QPS_ERR_NOT_FINITE",
                4: "QPS_ERR_FACTORIZATION", 5: "QPS_ERR_HIP", 6: "QPS_ERR_OUT_OF_MEMORY", 7: "QPS_ERR_NO_DEVICE",
                8: "QPS_ERR_UNSUPPORTED"}
QPS_F64, QPS_F32 = 0, 1
QPS_LINSYS_AUTO, QPS_LINSYS_CHOLESKY, QPS_LINSYS_CG, QPS_LINSYS_KKT_LDL, QPS_LINSYS_CG_EXPLICIT = 0, 1, 2, 3, 4
QPS_OP_P, QPS_OP_A, QPS_OP_AT, QPS_OP_PA, QPS_OP_REDUCED = 0, 1, 2, 3, 4


class QpsLibraryError(RuntimeError):
    """libqps_hip.so is missing / cannot be loaded."""


class QpsError(RuntimeError):
    """A libqps_hip call returned a non-zero qps_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


class QpsParams(C.Structure):
    _fields_ = [("numIterations", C.c_int32), ("adptRho", C.c_int32), ("numItrConv", C.c_int32),
                ("numItrPolish", C.c_int32), ("numItrMinres", C.c_int32), ("linsys", C.c_int32),
                ("trsvBlock", C.c_int32), ("reuseFactor", C.c_int32),
                ("epsAbs", C.c_double), ("epsRel", C.c_double), ("rho", C.c_double), ("sigma", C.c_double),
                ("alpha", C.c_double), ("delta", C.c_double), ("fctrRho", C.c_double), ("epsMinres", C.c_double),
                ("epsPcg", C.c_double), ("numItrPcg", C.c_int32), ("loopVariant", C.c_int32),
                ("polish", C.c_int32), ("reserved0", C.c_int32)]


class QpsInfo(C.Structure):
    _fields_ = [("convFlag", C.c_int32), ("iterations", C.c_int32), ("numRefactor", C.c_int32),
                ("cgIterations", C.c_int32), ("rhoFinal", C.c_double), ("rhoProposed", C.c_double),
                ("resPrim", C.c_double), ("resDual", C.c_double), ("tSetup", C.c_double), ("tLoop", C.c_double),
                ("tRefactor", C.c_double), ("polishFlag", C.c_int32), ("polishIterations", C.c_int32), ("tPolish", C.c_double),
                ("trsvBlock", C.c_int32), ("sweepVariant", C.c_int32), ("sweepGaveUp", C.c_int32), ("cgExplicit", C.c_int32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class QpsPolishReport(C.Structure):
    _fields_ = [("flag", C.c_int32), ("refinements", C.c_int32), ("minresIterations", C.c_int32), ("numActiveLower", C.c_int32),
                ("numActiveUpper", C.c_int32), ("reserved0", C.c_int32), ("relres", C.c_double), ("seconds", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "reserved0"}


class QpsProxQpParams(C.Structure):
    _fields_ = [("numIterations", C.c_int32), ("numItrConv", C.c_int32), ("adptRho", C.c_int32), ("loopVariant", C.c_int32),
                ("epsAbs", C.c_double), ("epsRel", C.c_double), ("rho", C.c_double), ("sigma", C.c_double), ("tau", C.c_double)]


class QpsProxQpReport(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("rho", C.c_double), ("sigma", C.c_double),
                ("resPrim", C.c_double), ("resDual", C.c_double)]


class QpsLdlReport(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("numRows", "numSparseColumns", "tailSize", "numSparseLevels", "treeHeight", "nnzK", "nnzL", "nnzStored")]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class QpsKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("seconds", C.c_double), ("launches", C.c_int64), ("algo_bytes", C.c_double)]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise QpsLibraryError("building libqps_hip.so failed (see compiler output above)")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:
            build()          # compile in-tree for gfx950 (hipcc cross-compiles without a GPU); never a CPU substitute
        except Exception as e:
            raise QpsLibraryError(f"{LIB_PATH} not found and building it failed ({e}): run "
                                  "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)") from e
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise QpsLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    i32, i64, dbl, hp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
    L.qps_default_params.argtypes = [C.POINTER(QpsParams)]
    L.qps_device_count.argtypes = []
    L.qps_create_dense.argtypes = [i64, i64, dp, i64, dp, i64, dp, dp, dp, i32, i32, C.POINTER(hp)]
    L.qps_create_csc.argtypes = [i64, i64, ip, ip, dp, ip, ip, dp, dp, dp, dp, i32, i32, i32, i32, C.POINTER(hp)]
    L.qps_solve.argtypes = [hp, dp, C.POINTER(QpsParams), C.POINTER(QpsInfo)]
    L.qps_polish.argtypes = [hp, dp, dp, C.POINTER(QpsParams), C.POINTER(QpsPolishReport)]
    L.qps_get_dual.argtypes = [hp, dp, dp]
    L.qps_linsys_init.argtypes = [hp, dbl, dbl, i32, i32]
    L.qps_linsys_solve.argtypes = [hp, dp, dp, dp, dbl, dbl, i32, dp, dp]
    L.qps_linsys_set_cg.argtypes = [hp, dbl, i32]
    L.qps_operator_apply.argtypes = [hp, i32, dp, dp, dbl, dbl]
    L.qps_create_dense_batch.argtypes = [i64, i64, i64, dp, dp, dp, dp, dp, i32, i32, C.POINTER(hp)]
    L.qps_solve_batch.argtypes = [hp, dp, C.POINTER(QpsParams), C.POINTER(QpsInfo)]
    L.qps_solve_batch_multi.argtypes = [i64, i64, i64, dp, dp, dp, dp, dp, i32, C.POINTER(i32), i32, i32, dp, C.POINTER(QpsParams), C.POINTER(QpsInfo), C.POINTER(i32), dp]
    L.qps_kernel_times.argtypes = [hp, C.POINTER(QpsKernelTime), i32, C.POINTER(i32)]
    L.qps_set_profiling.argtypes = [hp, i32]
    L.qps_proxqp_default_params.argtypes = [C.POINTER(QpsProxQpParams)]
    L.qps_proxqp_create_dense.argtypes = [i64, i64, i64, dp, i64, dp, dp, i64, dp, dp, i64, dp, i32, i32, C.POINTER(hp)]
    L.qps_proxqp_create_csc.argtypes = [i64, i64, i64, ip, ip, dp, dp, ip, ip, dp, dp, ip, ip, dp, dp, i32, i32, i32, C.POINTER(hp)]
    L.qps_proxqp_init_kkt.argtypes = [hp]
    L.qps_proxqp_set_state.argtypes = [hp, dp, dp, dp, dp]
    L.qps_proxqp_get_state.argtypes = [hp, dp, dp, dp, dp]
    L.qps_proxqp_solve.argtypes = [hp, C.POINTER(QpsProxQpParams), C.POINTER(QpsProxQpReport)]
    L.qps_linsys_auto.argtypes = [i64, i64, i64, i64, i32]
    L.qps_ldl_analyze.argtypes = [i64, i64, ip, ip, ip, ip, i32, ip, C.POINTER(QpsLdlReport)]
    L.qps_destroy.argtypes = [hp]
    L.qps_last_error.argtypes = [hp]
    L.qps_last_error.restype = C.c_char_p
    L.qps_version.argtypes = []
    L.qps_version.restype = C.c_char_p
    for name in EXPORTED_SYMBOLS:
        if name not in ("qps_last_error", "qps_version"):
            getattr(L, name).restype = C.c_int32
    _lib = L
    return L


def check(status: int, handle=None):
    if status != QPS_OK:
        msg = lib().qps_last_error(handle if handle else None)
        raise QpsError(status, (msg or b"").decode("utf-8", "replace"))


def default_params() -> QpsParams:
    p = QpsParams()
    check(lib().qps_default_params(C.byref(p)))
    return p
