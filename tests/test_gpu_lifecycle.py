"""Handle life cycle: create -> solve -> destroy, repeated, must give back what it took -- device memory (hipMemGetInfo through the same HIP runtime the
library uses) and host memory (resident set size).  Per-device caches (streams, pinned staging rings, the sweep gate) are filled by the first cycle of every
handle kind and are not counted.  Reference: the plugin pairs allocate their factorisation once per solve and drop it with the closure
(LinearSystemSolvers.jl:16-40, :110-142); a handle that outlives `SolveQuadraticProgram!` must do the same at qps_destroy."""
import ctypes
import gc

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

MiB = 1 << 20


def _hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.hipMemGetInfo.restype = ctypes.c_int
    lib.hipDeviceSynchronize.restype = ctypes.c_int
    return lib


def _free_bytes(hip):
    assert hip.hipDeviceSynchronize() == 0
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _rss_bytes():
    import psutil
    return psutil.Process().memory_info().rss


def _proxqp_problem(q, n, me, mi, stream):
    rng = q.make_rng(1220, stream)
    M = rng.standard_normal((n, n)); P = M.T @ M + 0.01 * np.eye(n); P = 0.5 * (P + P.T)
    qq = rng.standard_normal(n); A = rng.standard_normal((me, n)); C = rng.standard_normal((mi, n))
    x0 = rng.standard_normal(n)
    return P, qq, A, A @ x0, C, C @ x0 + 0.3 * np.abs(rng.standard_normal(mi)) - 0.1


def _cycles(q):
    """One create -> solve -> destroy cycle per handle kind; every kind allocates differently (dense factor + inverse, batch slabs, CSR pairs + sliced layouts,
    explicit reduced matrix, sparse L D L' levels, ProxQP KKT)."""
    dense = q.GenerateDenseBenchmarkQP(700, 1300, stream=3, feasible=True)
    wide = q.GenerateDenseBenchmarkQP(2100, 300, stream=4, feasible=True)
    batch = [q.GenerateDenseBenchmarkQP(200, 400, stream=20 + b, feasible=True) for b in range(6)]
    sparse = q.GenerateSparseBenchmarkQP(6000, 9000, densityA=2e-3, seed=5)
    lasso = q.GenerateRandomQP(q.ProblemClass.lassoOptimization, 10, rng=q.make_rng(9, 9))
    iso = q.GenerateRandomQP(q.ProblemClass.isotonicRegression, 20, rng=q.make_rng(9, 10))
    pq = _proxqp_problem(q, 150, 20, 120, 1)
    pqs = tuple(sp.csc_matrix(a) if a.ndim == 2 else a for a in pq)

    def run_dense(problem, dtype, **kw):
        P, qq, A, l, u = problem
        with q.QuadraticProgram(P, qq, A, l, u, dtype=dtype) as prob:
            x = np.zeros(P.shape[0])
            prob.solve(x, numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=25, **kw)

    def run_batch(dtype):
        with q.QuadraticProgramBatch(batch, dtype=dtype) as b:
            b.solve(numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True)

    def run_csc(problem, linsys):
        P, qq, A, l, u = problem
        with q.QuadraticProgram(P, qq, A, l, u, linsys=linsys) as prob:
            x = np.zeros(P.shape[0])
            prob.solve(x, numIterations=12, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=5, ϵPcg=1e-10, numItrPcg=500)

    def run_proxqp(problem):
        with q.ProxQP(*problem) as prob:
            q.SolveQuadraticProgramProxQP(prob, numIterations=60, ρ=200.0, adptΡ=True)

    def run_polish():
        P, qq, A, l, u = dense
        with q.QuadraticProgram(P, qq, A, l, u) as prob:
            x = np.zeros(P.shape[0])
            prob.solve(x, numIterations=400, ρ=0.1, adptΡ=True)
            z, y = prob.dual()
            prob.polish(x, y)

    return {
        "dense f64": lambda: run_dense(dense, "f64"),
        "dense f32": lambda: run_dense(dense, "f32"),
        "dense f64, blocked sweeps": lambda: run_dense(wide, "f64", trsvBlock=512),
        "batch f64": lambda: run_batch("f64"),
        "batch f32": lambda: run_batch("f32"),
        "csc cg (matrix-free)": lambda: run_csc(sparse, "cg"),
        "csc cg (explicit reduced matrix)": lambda: run_csc(iso, "cg_explicit"),
        "csc ldl": lambda: run_csc(lasso, "ldl"),
        "proxqp dense": lambda: run_proxqp(pq),
        "proxqp sparse": lambda: run_proxqp(pqs),
        "polish": run_polish,
    }


def test_create_solve_destroy_cycles_give_back_device_and_host_memory(gpu):
    hip = _hip()
    reps = 12
    report = {}
    for name, cycle in _cycles(gpu).items():
        cycle(); cycle()                                         # per-device caches, allocator pools of the runtime, code objects
        gc.collect()
        dev0, rss0 = _free_bytes(hip), _rss_bytes()
        for _ in range(reps):
            cycle()
        gc.collect()
        dev1, rss1 = _free_bytes(hip), _rss_bytes()
        report[name] = ((dev0 - dev1) / MiB, (rss1 - rss0) / MiB)
    print({k: (round(a, 2), round(b, 1)) for k, (a, b) in report.items()})
    for name, (dev_lost, rss_grown) in report.items():
        # a handle of these sizes holds 10-150 MiB on the device: one leaked allocation per cycle would show as >= 12 x its size
        assert dev_lost <= 4.0, (name, "device MiB not given back over %d cycles" % reps, report)
        assert rss_grown <= 64.0, (name, "host MiB grown over %d cycles" % reps, report)


def test_destroy_is_idempotent_and_a_closed_handle_refuses_work(gpu):
    P, qq, A, l, u = gpu.GenerateDenseBenchmarkQP(64, 128, stream=1, feasible=True)
    prob = gpu.QuadraticProgram(P, qq, A, l, u)
    x = np.zeros(64)
    prob.solve(x, numIterations=50)
    prob.close(); prob.close()                                   # second close: no double free
    with pytest.raises(Exception):
        prob.solve(x, numIterations=50)
