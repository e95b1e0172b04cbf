"""The in-process multi-device batch driver (round-3 review item 4; qps_solve_batch_multi, SURVEY 8e: "one host thread + one stream per device", hand-out at chunk
boundaries; RunBenchmarks.jl:88-104 is the per-problem loop being sharded).  A 1-GPU box lists the one card twice: two host threads, two streams, two handles alive at a
time on the same device -- the per-device state (stream / pinned-block recycling, one-time kernel attributes, the blocked-sweep gate) is driven from two threads at once."""
import numpy as np
import pytest

from quadraticprogramsolver_amd import dist as qd
from quadraticprogramsolver_amd.generator import GenerateDenseBenchmarkQP

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def test_two_workers_on_one_card_equal_single_handle_batches_bit_for_bit(gpu, c_oracle):
    """24 QPs (n = 160, m = 230: three 64-blocks, the fused batched pass), run to eps = 1e-6 with adaptive rho, devices = [0, 0], chunk = 8: the ranges the driver cuts are
    those of dist.handout_ranges whoever solves them, and every QP's x / flag / iteration count / refactor count equals -- bit for bit -- a single-handle
    QuadraticProgramBatch over exactly that range; QP 0 and the last QP against the oracle.  Then the static slabs (chunk = 0) of a fixed-K run."""
    cnt, n, m = 24, 160, 230
    probs = [GenerateDenseBenchmarkQP(n, m, stream=60 + b, feasible=True) for b in range(cnt)]
    kw = dict(numIterations=20000, ϵAbs=1e-6, ϵRel=1e-6, ρ=0.1, adptΡ=True)
    with gpu.QuadraticProgramBatch(probs, devices=[0, 0], chunk=8) as multi:
        X, flags, infos = multi.solve(**kw)
        owner = list(multi.worker_of)
        assert multi.worker_seconds is not None and len(multi.worker_seconds) == 2
    assert set(owner) <= {0, 1}
    ranges = qd.handout_ranges(cnt, 2, 8)
    assert sum(k for _, k in ranges) == cnt
    for b0, k in ranges:
        assert len(set(owner[b0:b0 + k])) == 1                                   # a range is one worker's
        with gpu.QuadraticProgramBatch(probs[b0:b0 + k]) as one:
            X1, f1, i1 = one.solve(**kw)
        assert np.array_equal(X[b0:b0 + k], X1), (b0, k, np.abs(X[b0:b0 + k] - X1).max())
        assert [int(f) for f in flags[b0:b0 + k]] == [int(f) for f in f1]
        for a, b in zip(infos[b0:b0 + k], i1):
            assert (a["iterations"], a["numRefactor"], a["rhoFinal"]) == (b["iterations"], b["numRefactor"], b["rhoFinal"])
    for b in (0, cnt - 1):
        xo, io = c_oracle.solve(*probs[b], numIterations=20000, epsAbs=1e-6, epsRel=1e-6, rho=0.1, adptRho=True)
        assert int(flags[b]) == io["convFlag"] and infos[b]["iterations"] == io["iterations"] and np.abs(X[b] - xo).max() <= 1e-5
    # fixed K, static contiguous slabs: QP b -> worker b // 12; equal to the 12-QP single-handle batches
    with gpu.QuadraticProgramBatch(probs, devices=[0, 0], chunk=0) as multi:
        Xs, _, infs = multi.solve(numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        assert list(multi.worker_of) == [b // 12 for b in range(cnt)]
    for w in range(2):
        with gpu.QuadraticProgramBatch(probs[12 * w:12 * w + 12]) as one:
            X1, _, _ = one.solve(numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1)
        assert np.array_equal(Xs[12 * w:12 * w + 12], X1)
    assert all(i["iterations"] == 60 for i in infs)


def test_three_workers_on_one_card_with_the_blocked_sweep(gpu):
    """Three host threads on device 0, trsvBlock = 512 on n = 1024 (the single-launch blocked sweeps: the co-residency gate of k_trsv_blocked.hip chains the persistent launches of
    different streams) -- per-QP handles (a shape the fused batched pass does not cover is not needed for that: the batch path falls back to per-QP solvers when m = 0)."""
    cnt, n = 6, 1024
    rng = np.random.default_rng(5)
    probs = []
    for b in range(cnt):
        G = rng.standard_normal((n, n)); P = G.T @ G / n + 0.05 * np.eye(n)
        probs.append((P, rng.standard_normal(n), np.zeros((0, n)), np.zeros(0), np.zeros(0)))
    with gpu.QuadraticProgramBatch(probs, devices=[0, 0, 0], chunk=1) as multi:
        X, flags, infos = multi.solve(numIterations=30, ϵAbs=0.0, ϵRel=0.0, trsvBlock=512)
    for b in range(cnt):
        P, q = probs[b][0], probs[b][1]
        x = np.zeros(n)                                                         # unconstrained: x_{k+1} = 1.6 x~ - 0.6 x_k with (P + sigma I) x~ = sigma x_k - q
        for _ in range(30):
            x = 1.6 * np.linalg.solve(P + 1e-6 * np.eye(n), 1e-6 * x - q) - 0.6 * x
        assert rel(X[b], x) <= 1e-9, (b, rel(X[b], x))


def test_errors_of_a_worker_come_back(gpu):
    """An asymmetric P in the middle of the batch: the range that holds it fails at handle creation, the call returns that error."""
    cnt, n, m = 6, 64, 80
    probs = [list(GenerateDenseBenchmarkQP(n, m, stream=90 + b, feasible=True)) for b in range(cnt)]
    probs[3][0] = probs[3][0].copy(); probs[3][0][1, 0] += 1.0
    from quadraticprogramsolver_amd._lib import QpsError
    with gpu.QuadraticProgramBatch([tuple(p) for p in probs], devices=[0, 0], chunk=2) as multi:
        with pytest.raises(QpsError) as e:
            multi.solve(numIterations=10)
    assert "symmetric" in str(e.value)
