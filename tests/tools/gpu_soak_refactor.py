"""Ad-hoc soak: the in-launch hand-offs of the fused Cholesky chain (tile workgroups -> diagonal workgroup, agent-scope counter) and of the blocked sweeps, exercised
thousands of times under UNEVEN load from a second host thread, checking every word: a solve is deterministic, so every repetition of the same solve must return the
same bits; one stale tile read would change them.  Not a test.  usage: python tests/tools/gpu_soak_refactor.py [seconds]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quadraticprogramsolver_amd as q
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
stop = threading.Event()
load_runs = [0]
def load():
    # unrelated work on the same card: a batch of mid-size QPs (fills the chip in bursts) and a sparse CG handle
    probs = [q.GenerateDenseBenchmarkQP(512, 1024, stream=50 + b, feasible=True) for b in range(24)]
    with q.QuadraticProgramBatch(probs) as batch:
        while not stop.is_set():
            batch.solve(numIterations=150, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=25)
            load_runs[0] += 1
t = threading.Thread(target=load); t.start()
bad = 0; total = 0; refactors = 0; gave_up = 0
t0 = time.time()
try:
    cases = ((4096, 2048, "f64", 0), (4096, 2048, "f32", 0), (4032, 1000, "f64", 1024), (2112, 3000, "f32", 0))
    if os.environ.get("SOAK_ONLY_BLOCKED"): cases = ((4032, 1000, "f64", 1024), (4096, 2048, "f32", 1024), (4032, 1000, "f64", 1024), (4096, 2048, "f32", 2048))
    for (n, m, dtype, nb) in cases:
        P, qq, A, l, u = q.GenerateDenseBenchmarkQP(n, m, stream=3, feasible=True)
        with q.QuadraticProgram(P, qq, A, l, u, dtype=dtype) as prob:
            first = None; reps = 0
            t1 = time.time()
            while time.time() - t1 < budget / 4:
                x = np.zeros(n); info = {}
                prob.solve(x, numIterations=60, ϵAbs=0.0, ϵRel=0.0, ρ=0.1, adptΡ=True, fctrΡ=1.0, numItrConv=5, trsvBlock=nb, info=info)   # a re-factorisation every 5 iterations
                z, y = prob.dual()
                key = (x.tobytes(), z.tobytes(), y.tobytes())
                if first is None: first = key
                elif key != first:
                    bad += 1
                    print(f"MISMATCH n={n} {dtype} trsvBlock={nb} repetition {reps}: max |dx| {np.abs(x - np.frombuffer(first[0])).max():.3e}; this solve: sweepVariant "
                          f"{info['sweepVariant']} sweepGaveUp {info.get('sweepGaveUp')}", flush=True)
                reps += 1; total += 1; refactors += info["numRefactor"]; gave_up += int(info.get("sweepGaveUp", 0) != 0)
            print(f"n={n} m={m} {dtype} trsvBlock={nb} (sweep variant {info['sweepVariant']}): {reps} identical solves, {info['numRefactor']} refactorisations each, load thread at {load_runs[0]} batch solves", flush=True)
finally:
    stop.set(); t.join()
print(f"{total} solves, {refactors} refactorisations (32-63 in-launch hand-offs each), {bad} mismatches, {gave_up} solves whose blocked sweep gave up and was repeated on variant 1, {time.time() - t0:.0f} s")
