"""Ad-hoc GPU bring-up script (not a test): compares libqps_hip with the CPU oracle on a few sizes and prints timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
from oracle import c_oracle as co

def run_case(pc, n, m=0, dense=True, K=50, nb=0, dtype="f64", **kw):
    P, qq, A, l, u = q.GenerateRandomQP(pc, n, numConstraints=m, dense=dense)
    n, m = P.shape[0], A.shape[0]
    t = time.time()
    prob = q.QuadraticProgram(P, qq, A, l, u, dtype=dtype)
    tc = time.time() - t
    x = np.zeros(n); info = {}
    flag = prob.solve(x, numIterations=K, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, info=info, **kw)
    z, y = prob.dual()
    xo, io = co.solve(P, qq, A, l, u, numIterations=K, epsAbs=0.0, epsRel=0.0, rho=kw.get("ρ", 1.0), adptRho=kw.get("adptΡ", False))
    sc = max(1.0, np.abs(xo).max())
    print(f"{q.ProblemClass(pc).name} n={n} m={m} K={K} nb={nb} {dtype}: flag {int(flag)}/{io['convFlag']} it {info['iterations']}/{io['iterations']} "
          f"dx={np.abs(x-xo).max()/sc:.2e} dz={np.abs(z-io['z']).max():.2e} dy={np.abs(y-io['y']).max():.2e} "
          f"res {info['resPrim']:.3e}/{io['resPrim']:.3e} {info['resDual']:.3e}/{io['resDual']:.3e} create {tc:.2f}s setup {info['tSetup']*1e3:.1f}ms loop {info['tLoop']*1e3:.1f}ms", flush=True)
    prob.close()

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what in ("small", "all"):
        run_case(1, 10, 5)
        run_case(1, 64, 128)
        run_case(2, 50)
        run_case(1, 100, 50, ρ=0.1, adptΡ=True, K=200)
        run_case(9, 100, dense=False, ρ=0.1, adptΡ=True, K=200)
        run_case(1, 300, 700, nb=64)
        run_case(1, 300, 700, nb=128)
        run_case(1, 1000, 2000, nb=256)
        run_case(1, 1000, 2000, nb=1024)
        run_case(1, 1100, 500, nb=512)
        run_case(1, 64, 128, dtype="f32")
        run_case(1, 1000, 2000, dtype="f32")
    if what in ("c2", "all"):
        n, m = 4096, 8192
        t = time.time(); P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m); print("gen", time.time() - t, flush=True)
        t = time.time(); prob = q.QuadraticProgram(P, qq, A, l, u); print("create", time.time() - t, flush=True)
        for nb, variant in ((4096, 1), (1024, 0), (2048, 0), (4096, 0)):
            x = np.zeros(n); info = {}
            prob.set_profiling(0)
            prob.solve(x, numIterations=200, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, loopVariant=variant, info=info)
            print(f"nb={nb} variant={variant} setup {info['tSetup']*1e3:.1f} ms loop {info['tLoop']*1e3:.1f} ms -> {info['iterations']/info['tLoop']:.1f} it/s", flush=True)
            prob.set_profiling(2)
            x2 = np.zeros(n)
            prob.solve(x2, numIterations=100, ϵAbs=0.0, ϵRel=0.0, trsvBlock=nb, reuseFactor=True, loopVariant=variant, info=info)
            for k in prob.kernel_times():
                us = k['seconds'] / k['launches'] * 1e6
                print(f"   {k['name']:24s} {us:9.1f} us/launch  {k['algo_bytes']/us/1e6:8.3f} TB/s algorithmic  ({k['launches']} launches)")
        if os.environ.get("QPS_ORACLE_C2"):
            t = time.time(); xo, io = co.solve(P, qq, A, l, u, numIterations=200, epsAbs=0.0, epsRel=0.0); print("oracle", time.time() - t, io['tSetup'], io['tLoop'])
            print("dx", np.abs(x - xo).max() / max(1, np.abs(xo).max()))
