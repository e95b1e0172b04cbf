"""Sparse matrices whose structure stresses the SpMV layouts of quadraticprogramsolver_amd/csrc/spmv_layout.cpp (shared by the CPU layout tests, the GPU
operator tests and tests/tools/gpu_fuzz_spmv.py): heavy-tailed row lengths, runs of empty rows, a dense row (long in every column block), dense columns
(= long rows of the transpose), row counts that leave a ragged last slice and sorting window, 1-3 column blocks in either precision, and a moderately
dense regime in which EVERY row holds more than SLONG = 96 entries per column block."""
import numpy as np
import scipy.sparse as sp

NS = [300, 2047, 5000, 7168, 7169, 9000, 15000, 22000]       # columns: 7168 = one fp64 x block exactly, 14336 = one fp32 block
MS = [64, 65, 1000, 2048, 2049, 4100, 12345, 30000]          # rows: 2048 = one sorting window exactly


def draw_case(rng, n=None, m=None, avg=None, dense_row=None, dense_cols=None, empty_run=None):
    """Returns (A csr m x n, tag).  Unset arguments are drawn."""
    n = int(rng.choice(NS)) if n is None else n
    m = int(rng.choice(MS)) if m is None else m
    avg = float(rng.choice([1.5, 4.0, 8.0, 20.0])) if avg is None else avg
    lens = np.minimum(rng.pareto(1.5, m) * avg * 0.5 + rng.poisson(avg * 0.5, m), n).astype(int)   # heavy tail
    tag = f"n={n} m={m} avg={avg}"
    if (rng.random() < 0.5) if empty_run is None else empty_run:
        a = int(rng.integers(0, m)); lens[a:a + int(rng.choice([3, 70, 700]))] = 0                # a run of empty rows
        tag += " +empty_run"
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens]) if rows.size else np.zeros(0, int)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n)).tolil()
    if (rng.random() < 0.5) if dense_row is None else dense_row:
        A[int(rng.integers(0, m)), :] = rng.standard_normal(n) * 0.05; tag += " +dense_row"
    if (rng.random() < 0.4) if dense_cols is None else dense_cols:
        j = int(rng.integers(0, n - 2)); A[:, j:j + 2] = rng.standard_normal((m, 2)) * 0.05; tag += " +dense_cols"
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A, tag + f" nnz={A.nnz} maxrow={int(np.diff(A.indptr).max()) if m else 0}"


def moderately_dense(rng, m, n, density):
    """Bernoulli pattern at `density` (5 % of a 7168-column block = ~350 entries per row and block: every row is a 'long row' of the sliced form)."""
    A = sp.random(m, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=rng.standard_normal, format="csr")
    A.sort_indices()
    return A, f"moderately dense n={n} m={m} density={density} nnz={A.nnz}"


def spd_companion(rng, n, seed):
    """P = M'M + 0.01 I with ~3 entries per column of M (what the fuzz tool pairs an A with)."""
    M = sp.random(n, n, density=min(3.0 / n, 0.5), random_state=np.random.RandomState(seed), data_rvs=rng.standard_normal, format="csc")
    return (M.T @ M + 1e-2 * sp.identity(n)).tocsc()
