"""Randomised differential run of the C ABI against the CPU oracle over shapes, dtypes, plugins and loop variants (not a test; prints every
mismatch).  usage: python tests/tools/gpu_fuzz.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
from oracle import c_oracle as co
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
t0 = time.time()
for c in range(cases):
    kind = rng.choice(os.environ.get("FUZZ_KINDS", "dense,dense,dense,cg,ldl,batch").split(","))
    dtype = "f64" if rng.random() < 0.75 else "f32"
    if kind == "dense":
        n = int(rng.choice([3, 17, 64, 65, 100, 129, 300, 511, 513, 1000, 1024, 1025, 1500, 2049, 2300]))
        m = int(rng.choice([0, 1, 5, n // 2 + 1, n, 2 * n])) if n < 1200 else int(rng.choice([1, 200, n // 2]))
    elif kind == "batch":
        n = int(rng.choice([16, 64, 100, 200, 1000])); m = int(rng.choice([n // 2 + 1, 2 * n]))
    else:
        n = int(rng.choice([20, 150, 400, 1200])); m = int(rng.choice([5, n // 2 + 1, 2 * n]))
    adpt = bool(rng.random() < 0.5)
    K = int(rng.choice([25, 60, 200])) if not adpt else int(rng.choice([200, 1000]))
    eps, rho0 = (0.0 if not adpt else 1e-7), float(rng.choice([0.1, 1.0, 10.0]))
    kw = dict(numIterations=K, ϵAbs=eps, ϵRel=eps, ρ=rho0, adptΡ=adpt)
    okw = dict(numIterations=K, epsAbs=eps, epsRel=eps, rho=rho0, adptRho=adpt)
    tol = 1e-8 if dtype == "f64" else (2e-3 if m >= n else 5e-2)   # fp32 with few constraints: cond(P) ~ 1e6 is not tamed by rho A'A
    tag = f"case {c}: {kind} {dtype} n={n} m={m} K={K} adpt={adpt} rho={rho0}"
    try:
        if kind in ("dense", "batch"):
            P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, max(m, 0), stream=100 + c, feasible=bool(rng.random() < 0.7)) if m > 0 else \
                q.GenerateDenseBenchmarkQP(n, 1, stream=100 + c)
            if m == 0: A, l, u = A[:0], l[:0], u[:0]
        else:
            dens = 0.05 if n <= 400 else 0.01
            P, qq, A, l, u = q.GenerateSparseBenchmarkQP(n, m, densityA=dens, seed=200 + c)
        xo, io = co.solve(P, qq, A, l, u, **okw) if kind != "cg" else co.solve(P, qq, A, l, u, linsys=co.KIND_CG_MATFREE, epsPcg=1e-13, numItrPcg=5000, **okw)
        if kind == "dense":
            extra = {}
            if n >= 1025 and rng.random() < 0.6: extra["trsvBlock"] = int(rng.choice([64, 256, 512, 1024]))
            if rng.random() < 0.3: extra["loopVariant"] = int(rng.choice([1, 2]))
            tag += f" {extra}"
            with q.QuadraticProgram(P, qq, A, l, u, dtype=dtype) as prob:
                x = np.zeros(n); info = {}
                flag = prob.solve(x, info=info, **kw, **extra)
        elif kind == "batch":
            cnt = int(rng.choice([2, 3, 5]))
            probs = [(P, qq, A, l, u)] + [q.GenerateDenseBenchmarkQP(n, m, stream=1000 + 10 * c + b, feasible=True) for b in range(1, cnt)]
            with q.QuadraticProgramBatch(probs, dtype=dtype) as batch:
                X, flags, infos = batch.solve(**kw)
            x, flag, info = X[0], flags[0], infos[0]
        else:
            with q.QuadraticProgram(P, qq, A, l, u, dtype=dtype, linsys=kind) as prob:
                x = np.zeros(n); info = {}
                flag = prob.solve(x, info=info, ϵPcg=1e-13, numItrPcg=5000, **kw) if kind == "cg" else prob.solve(x, info=info, **kw)
        dev = np.abs(x - xo).max() / max(1.0, np.abs(xo).max())
        same = (int(flag) == io["convFlag"] and info["iterations"] == io["iterations"]) if dtype == "f64" else True
        loose = 1e-5 if (adpt or kind == "cg") and dtype == "f64" else tol
        note = ""
        if not same and eps == 0.0 and dev <= 1e-14 and {int(flag), io["convFlag"]} == {1, 2}:
            # eps = 0 => epsAdmm = 0 (SolveQuadraticProgram.jl:34): the stall test (:105) fires only on a bit-exact repeat of x and z, and whether a converged sequence
            # produces one is decided by the last bit of every operation (seed 53 case 166; tests/test_oracle.py::test_bit_exact_stagnation_...: the C and the numpy
            # restatement differ in exactly this way).  Same x to rounding, flags {convNumItr, convAdmm}: accepted and named.
            same = True; note = f" [bit-exact stall is rounding-dependent at eps = 0: flag {int(flag)}/{io['convFlag']} its {info['iterations']}/{io['iterations']}]"
        if not (dev <= loose) or not same or not np.all(np.isfinite(x)):
            bad += 1
            print(f"MISMATCH {tag}: dev={dev:.2e} flag {int(flag)}/{io['convFlag']} its {info['iterations']}/{io['iterations']} ref {info.get('numRefactor')}/{io.get('numRefactor')}", flush=True)
        else:
            print(f"ok {tag}: dev={dev:.1e}{note}", flush=True)
    except Exception as e:
        bad += 1
        print(f"ERROR {tag}: {type(e).__name__}: {e}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
