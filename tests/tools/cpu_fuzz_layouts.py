"""Randomised run of the SpMV layout builders + their host interpreter (tests/capi/layout_shim.cpp) against scipy over shapes wider than tests/spmv_cases.py draws:
random launch widths (a workgroup's share of the rows decides sliced / staged / several windows per workgroup), up to 200 000 rows or columns, skewed row costs,
dense rows and columns, empty stretches, both forms, both precisions, A and A'.  CPU only (not a test; prints every mismatch).
usage: python tests/tools/cpu_fuzz_layouts.py [cases] [seed]     QPS_HOST_TEST_LIB selects another build (the sanitizer one: tests/tools/run_sanitizers.sh)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ctypes as C, subprocess
import numpy as np, scipy.sparse as sp
import test_layout_cpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = os.environ.get("QPS_HOST_TEST_LIB")
if not path:
    subprocess.check_call(["make", "-C", T.CSRC, "-s", "host-test"])
    path = os.path.join(root, "quadraticprogramsolver_amd", "libqps_host_test.so")
shim = C.CDLL(path)
rng = np.random.default_rng(seed)
bad = 0; t0 = time.time(); forms = {}


def draw(rng):
    rows = int(rng.choice([1, 63, 64, 65, 1000, 2304, 2305, 9000, 40000, 52001, 120000, 200000]))
    cols = int(rng.choice([1, 300, 7168, 7169, 14336, 14337, 20011, 50000, 200000]))
    avg = float(rng.choice([0.3, 1.5, 4.0, 9.0, 30.0]))
    avg = min(avg, cols)
    if rows * avg > 3e6: avg = 3e6 / rows
    kind = rng.choice(["uniform", "skewed", "pareto"])
    if kind == "uniform":
        M = sp.random(rows, cols, density=min(1.0, avg / cols), random_state=rng, format="csr", dtype=np.float64)
    elif kind == "skewed":
        top = max(1, rows // 10)
        M = sp.vstack([sp.random(top, cols, density=min(1.0, 8 * avg / cols), random_state=rng, format="csr", dtype=np.float64),
                       sp.random(rows - top, cols, density=min(1.0, 0.2 * avg / cols), random_state=rng, format="csr", dtype=np.float64)]).tocsr()
    else:
        lens = np.minimum(rng.pareto(1.3, rows) * avg * 0.5 + rng.poisson(avg * 0.5, rows), cols).astype(np.int64)
        r = np.repeat(np.arange(rows), lens)
        c = (rng.random(r.size) * cols).astype(np.int64)                     # duplicates are summed by csr_matrix: fine
        M = sp.csr_matrix((rng.standard_normal(r.size), (r, c)), shape=(rows, cols))
    tag = f"{kind} {rows}x{cols} avg={avg:.2g}"
    extra = []
    if rng.random() < 0.4 and cols > 1:                                       # dense rows (long in every column block)
        k = int(rng.choice([1, 2, 5]))
        rr = np.unique(rng.integers(0, rows, k))
        D = sp.random(rr.size, cols, density=float(rng.choice([0.05, 0.5, 1.0])), random_state=rng, format="csr", dtype=np.float64)
        S = sp.csr_matrix((np.ones(rr.size), (rr, np.arange(rr.size))), shape=(rows, rr.size))
        M = (M + S @ D).tocsr(); extra.append(f"+{rr.size} dense rows")
    if rng.random() < 0.3 and rows > 1:                                       # dense columns (long rows of the transpose)
        j = int(rng.integers(0, cols))
        Dc = sp.csr_matrix((rng.standard_normal(rows), (np.arange(rows), np.full(rows, j))), shape=(rows, cols))
        M = (M + Dc).tocsr(); extra.append("+dense col")
    if rng.random() < 0.4 and rows > 10:                                      # an empty stretch of rows
        a = int(rng.integers(0, rows - 1)); b = min(rows, a + int(rng.choice([3, 70, 3000])))
        keep = np.ones(rows); keep[a:b] = 0.0
        M = (sp.diags(keep) @ M).tocsr(); extra.append(f"+empty {a}:{b}")
    M.eliminate_zeros(); M.sort_indices()
    return M, tag + " " + " ".join(extra) + f" nnz={M.nnz}"


for c in range(cases):
    M, tag = draw(rng)
    wgs = int(rng.choice([16, 64, 96, 256, 512, 1000]))
    transposed = bool(rng.random() < 0.4)
    if transposed: M = sp.csr_matrix(M.T); M.sort_indices()
    x = rng.standard_normal(M.shape[1])
    ref = M @ x
    scale = (abs(M) @ np.abs(x)).max() + 1e-300 if M.nnz else 1.0
    tag = f"case {c}: {tag}{' (transposed)' if transposed else ''} wgs={wgs}"
    try:
        msgs = []
        for dtype, tol in (("f64", 1e-13), ("f32", 2e-5)):
            for form in ("sell", "tasks"):
                rc, y, stats = T.apply(shim, form, dtype, M, x, wgs=wgs)
                if rc < 0: msgs.append(f"{form}/{dtype}: builder raised"); continue
                if rc == 0: forms[(form, "declined")] = forms.get((form, "declined"), 0) + 1; continue
                if form == "sell":
                    staged = int(stats[7]) // 100000
                    forms[("sell", "staged" if staged else "lane")] = forms.get(("sell", "staged" if staged else "lane"), 0) + 1
                if not np.all(np.isfinite(y)): msgs.append(f"{form}/{dtype}: {int(np.isnan(y).sum())} rows written twice or never"); continue
                err = np.abs(y - ref).max() / scale
                if err > tol: msgs.append(f"{form}/{dtype}: err {err:.2e}")
        if msgs:
            bad += 1; print(f"MISMATCH {tag}: " + "; ".join(msgs), flush=True)
        else:
            print(f"ok {tag}", flush=True)
    except Exception as e:
        bad += 1; print(f"ERROR {tag}: {type(e).__name__}: {e}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s; forms applied: {dict(sorted((' '.join(k), v) for k, v in forms.items()))}")
