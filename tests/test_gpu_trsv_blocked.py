"""GPU parity tests of the single-launch blocked triangular substitution (k_trsv_blocked.hip; qps_info.sweepVariant = 5): the kernel
BASELINE's north_star names.  Each case goes through the C ABI with an explicit trsvBlock and is compared with the CPU oracle
(iterate level, 1e-9 relative in fp64 / 1e-3 in fp32), with the linear system itself (host fp64 residual) or with the explicit-inverse
sweep of the same handle."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from quadraticprogramsolver_amd.generator import GenerateDenseBenchmarkQP, make_rng

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


# (n, m, trsvBlock): two full blocks; three blocks with a ragged last one (NP = 2112: 1024 + 1024 + 64); nb = 512 with five blocks
# (NP = 2112: 4 x 512 + 64); NP = 1088 = 2 x 512 + 64
CASES64 = [(2048, 300, 1024), (2100, 700, 1024), (2100, 700, 512), (1080, 2000, 512)]


@pytest.mark.parametrize("n,m,nb", CASES64)
def test_blocked_sweep_iterates_match_oracle(gpu, c_oracle, n, m, nb):
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=21, feasible=True)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=25, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, info=info)
        z, y = prob.dual()
        assert info["sweepVariant"] == 5 and info["trsvBlock"] == nb
        xo, io = c_oracle.solve(P, q, A, l, u, numIterations=25, epsAbs=0.0, epsRel=0.0, rho=0.1)
        assert rel(x, xo) <= 1e-9 and rel(z, io["z"]) <= 1e-9 and rel(y, io["y"]) <= 1e-8


def test_blocked_sweep_adaptive_rho_same_counts_as_oracle(gpu, c_oracle):
    """ρ switches re-build the pre-multiplied sweep matrix; flag, iteration and refactor counts as the oracle's (RunTests.jl:50-58 parameters)."""
    n, m = 2100, 1500
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=22, feasible=True)
    kw = dict(numIterations=4000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        x = np.zeros(n); info = {}
        flag = prob.solve(x, trsvBlock=1024, info=info, **kw)
    xo, io = c_oracle.solve(P, q, A, l, u, numIterations=4000, epsAbs=1e-7, epsRel=1e-7, rho=0.1, adptRho=True)
    assert info["sweepVariant"] == 5
    assert int(flag) == io["convFlag"] and info["iterations"] == io["iterations"] and info["numRefactor"] == io["numRefactor"]
    assert np.abs(x - xo).max() <= 1e-5


@pytest.mark.parametrize("nb", [1024, 2048])
def test_blocked_sweep_fp32(gpu, c_oracle, nb):
    """fp32: trsvBlock 1024 (256-thread rows) and 2048 (512-thread rows); n = 2300 gives 3 / 2 blocks with a ragged last one."""
    n, m = 2300, 900
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=23, feasible=True)
    with gpu.QuadraticProgram(P, q, A, l, u, dtype="f32") as prob:
        x = np.zeros(n); info = {}
        prob.solve(x, numIterations=25, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, info=info)
        assert info["sweepVariant"] == 5
        xo, _ = c_oracle.solve(P, q, A, l, u, numIterations=25, epsAbs=0.0, epsRel=0.0, rho=0.1)
        assert rel(x, xo) <= 1e-3


def test_blocked_sweep_plugin_pair_solves_the_system(gpu):
    """The literal plugin pair (LinearSystemSolvers.jl:134-139) through the blocked sweeps: (P + σI + ρA'A) x~ = σx − q + A'(ρz − y), z~ = A x~,
    at ρ = 0.1 and at the ρ = 1e6 clamp, and the same x~ as the explicit inverse of the same handle."""
    n, m = 3000, 1200
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=24)
    rng = make_rng(78, 0)
    x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        for rho in (0.1, 1e6):
            sigma = 1e-6
            got = {}
            for nb in (1024, 4096):
                prob.linsys_init(rho, sigma, trsvBlock=nb)
                xx, zz = np.zeros(n), np.zeros(m)
                prob.linsys_solve(x, z, y, rho, sigma, False, xx, zz)
                rhs = sigma * x - q + A.T @ (rho * z - y)
                lhs = P @ xx + sigma * xx + rho * (A.T @ (A @ xx))
                assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(rhs).max()
                assert np.abs(zz - A @ xx).max() <= 1e-11 * max(1.0, np.abs(zz).max())
                got[nb] = xx
            assert rel(got[1024], got[4096]) <= 1e-9


def test_blocked_sweep_full_size_c2(gpu):
    """BASELINE config 2 (n = 4096, m = 8192, fp64) with trsvBlock = 1024: same iterates as the explicit inverse, reported residuals are the
    true ones; also n = 6000 (six blocks, ragged) against the multi-launch substitution with 64-column blocks."""
    n, m = 4096, 8192
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        xs = {}
        for nb in (1024, 512, 4096):
            xk = np.zeros(n); info = {}
            prob.solve(xk, numIterations=50, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, info=info)
            assert info["sweepVariant"] == (2 if nb == 4096 else 5)
            zk, yk = prob.dual()
            assert abs(info["resPrim"] - np.abs(A @ xk - zk).max()) <= 1e-9 * max(1.0, info["resPrim"])
            assert abs(info["resDual"] - np.abs(P @ xk + q + A.T @ yk).max()) <= 1e-8 * max(1.0, info["resDual"])
            xs[nb] = xk
        assert rel(xs[1024], xs[4096]) <= 1e-9 and rel(xs[512], xs[4096]) <= 1e-9
    n, m = 6000, 500
    P, q, A, l, u = GenerateDenseBenchmarkQP(n, m, stream=25)
    with gpu.QuadraticProgram(P, q, A, l, u) as prob:
        xs = {}
        for nb in (1024, 64):
            xk = np.zeros(n); info = {}
            prob.solve(xk, numIterations=30, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=nb, info=info)
            assert info["sweepVariant"] == (5 if nb == 1024 else 1)
            xs[nb] = xk
        assert rel(xs[1024], xs[64]) <= 1e-9


def _child(script, env_extra, tmp_path, name):
    f = tmp_path / name; f.write_text(textwrap.dedent(script))
    env = {k: v for k, v in os.environ.items() if not k.startswith("QPS_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, str(f), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


SCRIPT = '''
    import sys, json, numpy as np
    sys.path.insert(0, sys.argv[1])
    import quadraticprogramsolver_amd as q
    P, qq, A, l, u = q.GenerateDenseBenchmarkQP(2100, 900, stream=26, feasible=True)
    out = {}
    with q.QuadraticProgram(P, qq, A, l, u) as prob:
        for tag, kw in {"fixed": dict(numIterations=40, ϵAbs=0.0, ϵRel=0.0, ρ=0.1), "adaptive": dict(numIterations=3000, ϵAbs=1e-7, ϵRel=1e-7, ρ=0.1, adptΡ=True)}.items():
            x = np.zeros(2100); info = {}
            prob.solve(x, trsvBlock=1024, info=info, **kw)
            out[tag] = {"x": x.tolist(), "variant": info["sweepVariant"], "iterations": info["iterations"], "refactor": info["numRefactor"], "gaveUp": info["sweepGaveUp"]}
        xx, zz = np.zeros(2100), np.zeros(900)
        prob.linsys_init(0.3, 1e-6, trsvBlock=1024)
        prob.linsys_solve(np.ones(2100), np.ones(900), -np.ones(900), 0.3, 1e-6, False, xx, zz)
        out["pair"] = {"x": xx.tolist()}
    print(json.dumps(out))
'''


def test_blocked_sweep_gives_up_cleanly(gpu, tmp_path):
    """The backstop: a launch whose workgroups are not all resident (another PROCESS running the same kernel on the card -- launches of one
    process are chained by the sweep gate) must not hang.  With the spin limit at zero every courier gives up at its first unanswered poll, the
    launch drains, the host sees the give-up word at its next read-back and repeats THAT solve on the multi-launch substitution
    (qps_info.sweepGaveUp = 1, sweepVariant = 1); the next solve tries the blocked sweeps again (and, with the limit still at zero, gives up
    again) -- same answers as the normal run (and as the window knob, which only changes the prefetch depth)."""
    base = _child(SCRIPT, {}, tmp_path, "base.py")
    assert base["fixed"]["variant"] == 5 and base["adaptive"]["variant"] == 5 and base["fixed"]["gaveUp"] == 0 and base["adaptive"]["gaveUp"] == 0
    gave_up = _child(SCRIPT, {"QPS_SWEEP_SPIN_LIMIT": "0"}, tmp_path, "giveup.py")
    assert gave_up["fixed"]["variant"] == 1 and gave_up["adaptive"]["variant"] == 1
    assert gave_up["fixed"]["gaveUp"] == 1 and gave_up["adaptive"]["gaveUp"] == 1      # every solve retried the blocked sweeps first
    deep = _child(SCRIPT, {"QPS_SWEEP_WINDOW": "3"}, tmp_path, "deep.py")
    off = _child(SCRIPT, {"QPS_SWEEP_BLOCKED": "0"}, tmp_path, "off.py")
    assert deep["fixed"]["variant"] == 5 and off["fixed"]["variant"] == 1
    for other in (gave_up, deep, off):
        for tag in ("fixed", "adaptive", "pair"):
            a, b = np.array(other[tag]["x"]), np.array(base[tag]["x"])
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), tag
        assert other["adaptive"]["iterations"] == base["adaptive"]["iterations"] and other["adaptive"]["refactor"] == base["adaptive"]["refactor"]


def test_blocked_sweeps_from_two_host_threads(gpu, c_oracle):
    """Two handles driven from two host threads, both on the single-launch blocked sweeps (SURVEY §8b "Threading", include/qps.h:19).  Two
    persistent 256-workgroup launches must never hold parts of the chip while waiting for the rest: the per-device sweep gate chains the
    pairs of the two streams, so EVERY solve of both threads stays on variant 5, none gives up, nobody waits for a spin limit (the whole
    two-thread run takes seconds) -- and every solve returns the oracle's iterates."""
    import threading
    import time
    cases = [GenerateDenseBenchmarkQP(2100, 400, stream=60, feasible=True), GenerateDenseBenchmarkQP(2500, 300, stream=61, feasible=True)]
    results = [None, None]

    def work(k):
        P, q, A, l, u = cases[k]
        with gpu.QuadraticProgram(P, q, A, l, u) as prob:
            outs = []
            for rep in range(4):
                x = np.zeros(P.shape[0]); info = {}
                prob.solve(x, numIterations=200, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, trsvBlock=1024, info=info)
                outs.append((x, info["sweepVariant"], info["sweepGaveUp"]))
            results[k] = outs

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    elapsed = time.perf_counter() - t0
    for k in range(2):
        assert results[k] is not None
        xo, _ = c_oracle.solve(*cases[k], numIterations=200, epsAbs=0.0, epsRel=0.0, rho=0.1)
        for x, variant, gave_up in results[k]:
            assert variant == 5 and gave_up == 0 and rel(x, xo) <= 1e-9
        print("sweep variants of thread", k, [v for _, v, _ in results[k]])
    assert elapsed < 5.0, elapsed          # 2 x 4 solves of 200 iterations incl. handle creation and factorisation; a give-up alone used to cost 0.4 s
