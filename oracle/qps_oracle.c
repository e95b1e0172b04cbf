/*
 * qps_oracle.c -- CPU restatement of the reference ADMM QP path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X build.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The product library
 * (quadraticprogramsolver_amd/csrc) never links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (Julia) cannot run in this pipeline, ships no golden vectors, and its
 * third-party numerics (SuiteSparse/QDLDL/LDLFactorizations LDLt, IterativeSolvers.cg!) are un-vendored and
 * unpinned (no Manifest.toml, Project.toml lists none of them).  This restatement follows the reference
 * source text line by line (citations below are file:line under /root/reference) and is pinned only by
 * (a) analytic known-answer tests, (b) an independently written numpy/LAPACK mirror (oracle/qps_oracle_np.py),
 * (c) KKT optimality certificates, (d) the reference's own tolerance contract (RunTests.jl:58,93).
 *
 * What is restated
 *   loop            SolveQuadraticProgram.jl:14-76   -> oq_solve_*()
 *   convergence     SolveQuadraticProgram.jl:79-112  -> check_convergence()
 *   KKT plugins     LinearSystemSolvers.jl:16-107    -> linsys kind 1 (dense LDLt of the quasi-definite KKT matrix) and kind 4
 *                                                       (sparse CSC LDLt in the manner of QDLDL: elimination tree, column counts,
 *                                                       up-looking numeric factorisation, :47-75; the ordering is an input)
 *   reduced plugins LinearSystemSolvers.jl:110-142   -> linsys kind 0 (Cholesky instead of cg!, cf. ProxQP.jl:175-206,221-225)
 *                                                       and kind 2 (cg! on the explicit reduced matrix)
 *   matrix-free     LinearSystemSolvers.jl:145-186   -> linsys kind 3 (operator P w + rho A'(A w) + sigma w)
 *   cg!             IterativeSolvers.jl (un-vendored, unpinned; v0.9 published algorithm: stop when
 *                   ||r||_2 <= max(reltol*||r0||_2, abstol), reltol = sqrt(eps), x warm-started)
 *
 * All matrices are column-major (Julia layout).  Sparse inputs are CSC with 0-based int64 indices.
 * Build: see oracle/Makefile (gcc -O3 -march=native -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OQ_EXPORT __attribute__((visibility("default")))
#define OQ_PAR_MIN ((int64_t)1 << 18)   /* below this many multiply-adds a loop stays serial */

/* SolveQuadraticProgram.jl:12  @enum ConvergenceFlag convNumItr = 1 convAdmm convPrimDual */
enum { convNumItr = 1, convAdmm = 2, convPrimDual = 3 };

typedef struct {
    int32_t numIterations;   /* SolveQuadraticProgram.jl:15 default 5000 */
    int32_t adptRho;         /* :16 adptΡ::Bool = false */
    int32_t numItrConv;      /* :17 default 25 */
    int32_t linsys;          /* 0 reduced Cholesky, 1 KKT LDLt, 2 cg! explicit reduced matrix, 3 cg! matrix-free */
    double epsAbs, epsRel;   /* :15 default 1e-6 */
    double rho, sigma, alpha;/* :16 defaults 1, 1e-6, 1.6 */
    double fctrRho;          /* :17 default 5 */
    double epsPcg;           /* LinearSystemSolvers.jl:125 default 1e-6 */
    int32_t numItrPcg;       /* LinearSystemSolvers.jl:125 default 1000 */
    int32_t numThreads;      /* 0 = leave OpenMP default */
    int32_t loopThreads;     /* 0 = same as numThreads; else thread count for the iteration loop only (setup keeps numThreads) */
    int32_t reserved;
} oq_params;

typedef struct {
    int32_t convFlag;
    int32_t iterations;      /* number of loop bodies executed (the reference never reports this) */
    int32_t numRefactor;     /* number of changedΡ events */
    int32_t cgIterations;    /* total inner CG iterations (kinds 2,3) */
    double rhoFinal;         /* ρ in force at exit */
    double rhoProposed;      /* ρρ at exit */
    double resPrim, resDual; /* from the last CheckConvergence call */
    double maxNormPrim, maxNormDual;
    double tSetup, tLoop;    /* seconds */
} oq_info;

static double now_sec(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

/* Julia norm(v, Inf): max |v_i|, NaN-propagating, 0 for empty vectors. */
static double norm_inf(const double *v, int64_t n) {
    double r = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double a = fabs(v[i]);
        if (a > r || isnan(a)) r = a;
        if (isnan(r)) return r;
    }
    return r;
}
static double norm_inf_diff(const double *a, const double *b, int64_t n) {
    double r = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double d = fabs(a[i] - b[i]);
        if (d > r || isnan(d)) r = d;
        if (isnan(r)) return r;
    }
    return r;
}
/* Julia max(a,b) propagates NaN */
static double jmax(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }
/* Julia clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)); NaN passes through */
static double jclamp(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

/* ------------------------------------------------------------------------------------------------
 * Problem operator abstraction: dense column-major or CSC.  y = A x, y = A' x, y = P x.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n, m;
    int sparse;
    /* dense */
    const double *P, *A;
    /* CSC */
    const int64_t *Pcp, *Pri; const double *Pnz;
    const int64_t *Acp, *Ari; const double *Anz;
} oq_prob;

static void mul_A(const oq_prob *p, const double *x, double *y) { /* y[m] = A x */
    int64_t n = p->n, m = p->m;
    if (!p->sparse) {
#pragma omp parallel for schedule(static) if (n * m > OQ_PAR_MIN)
        for (int64_t i0 = 0; i0 < m; i0 += 256) {
            int64_t i1 = i0 + 256 < m ? i0 + 256 : m;
            for (int64_t i = i0; i < i1; ++i) y[i] = 0.0;
            for (int64_t j = 0; j < n; ++j) {
                const double *col = p->A + j * m; double xj = x[j];
                for (int64_t i = i0; i < i1; ++i) y[i] += col[i] * xj;
            }
        }
    } else {
        for (int64_t i = 0; i < m; ++i) y[i] = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            double xj = x[j];
            for (int64_t k = p->Acp[j]; k < p->Acp[j + 1]; ++k) y[p->Ari[k]] += p->Anz[k] * xj;
        }
    }
}
static void mul_At(const oq_prob *p, const double *v, double *y) { /* y[n] = A' v */
    int64_t n = p->n, m = p->m;
    if (!p->sparse) {
#pragma omp parallel for schedule(static) if (n * m > OQ_PAR_MIN)
        for (int64_t j = 0; j < n; ++j) {
            const double *col = p->A + j * m; double s = 0.0;
            for (int64_t i = 0; i < m; ++i) s += col[i] * v[i];
            y[j] = s;
        }
    } else {
#pragma omp parallel for schedule(static) if (p->Acp[n] > OQ_PAR_MIN)
        for (int64_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (int64_t k = p->Acp[j]; k < p->Acp[j + 1]; ++k) s += p->Anz[k] * v[p->Ari[k]];
            y[j] = s;
        }
    }
}
static void mul_P(const oq_prob *p, const double *x, double *y) { /* y[n] = P x (P symmetric, stored in full) */
    int64_t n = p->n;
    if (!p->sparse) {
        /* P symmetric: row i of P == column i, so use contiguous columns for a dot product */
#pragma omp parallel for schedule(static) if (n * n > OQ_PAR_MIN)
        for (int64_t i = 0; i < n; ++i) {
            const double *col = p->P + i * n; double s = 0.0;
            for (int64_t j = 0; j < n; ++j) s += col[j] * x[j];
            y[i] = s;
        }
    } else {
#pragma omp parallel for schedule(static) if (p->Pcp[n] > OQ_PAR_MIN)
        for (int64_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (int64_t k = p->Pcp[j]; k < p->Pcp[j + 1]; ++k) s += p->Pnz[k] * x[p->Pri[k]];
            y[j] = s;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Dense kernels for the factorisations (column-major, lower triangle).
 * ---------------------------------------------------------------------------------------------- */
/* C[n x n] (lower triangle incl. diagonal, mirrored at the end) = A' A, A is m x n column-major.
 * LinearSystemSolvers.jl:112  mAA = mA' * mA */
static void dense_AtA(int64_t m, int64_t n, const double *A, double *C) {
    const int64_t TB = 48;
#pragma omp parallel for schedule(dynamic, 1) if (n * n * m > 64 * OQ_PAR_MIN)
    for (int64_t jb = 0; jb < n; jb += TB) {
        int64_t je = jb + TB < n ? jb + TB : n;
        for (int64_t ib = jb; ib < n; ib += TB) {
            int64_t ie = ib + TB < n ? ib + TB : n;
            for (int64_t j = jb; j < je; ++j) {
                const double *cj = A + j * m;
                for (int64_t i = (ib > j ? ib : j); i < ie; ++i) {
                    const double *ci = A + i * m; double s = 0.0;
                    for (int64_t k = 0; k < m; ++k) s += ci[k] * cj[k];
                    C[i + j * n] = s;
                }
            }
        }
    }
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = j + 1; i < n; ++i) C[j + i * n] = C[i + j * n];
}

/* In-place lower Cholesky M = L L' (column-major, lower triangle overwritten by L).  Returns 0 or 1+index of
 * the failing pivot.  Right-looking blocked so that the n=4096 case is usable as a CPU baseline.
 * ProxQP.jl:196-197 cholesky!(...) is the in-repo precedent for a dense Cholesky of P + sigma I + rho A'A. */
static int64_t dense_cholesky(int64_t n, double *M) {
    const int64_t NB = 64;
    for (int64_t kb = 0; kb < n; kb += NB) {
        int64_t ke = kb + NB < n ? kb + NB : n;
        /* factor the diagonal block and the panel below it, column by column */
        for (int64_t k = kb; k < ke; ++k) {
            double d = M[k + k * n];
            for (int64_t p = kb; p < k; ++p) d -= M[k + p * n] * M[k + p * n];
            if (!(d > 0.0)) return k + 1;
            d = sqrt(d);
            M[k + k * n] = d;
            double inv = 1.0 / d;
#pragma omp parallel for schedule(static) if (n - k > 512)
            for (int64_t i = k + 1; i < n; ++i) {
                double s = M[i + k * n];
                for (int64_t p = kb; p < k; ++p) s -= M[i + p * n] * M[k + p * n];
                M[i + k * n] = s * inv;
            }
        }
        /* trailing update: M[i,j] -= sum_{p in block} L[i,p] L[j,p], j >= ke, i >= j */
        int64_t nbk = ke - kb;
#pragma omp parallel for schedule(dynamic, 8) if ((n - ke) * (n - ke) > OQ_PAR_MIN)
        for (int64_t j = ke; j < n; ++j) {
            double lj[64];
            for (int64_t p = 0; p < nbk; ++p) lj[p] = M[j + (kb + p) * n];
            for (int64_t p = 0; p < nbk; ++p) {
                const double *colp = M + (kb + p) * n; double ljp = lj[p];
                double *cj = M + j * n;
                for (int64_t i = j; i < n; ++i) cj[i] -= colp[i] * ljp;
            }
        }
    }
    return 0;
}
/* Solve L L' x = b in place (two triangular sweeps).  ProxQP.jl:224 ldiv!(vX, sC, vR).
 * Blocked so that the all-cores CPU baseline is not one core's TRSV: 256-wide diagonal blocks are substituted serially
 * (n * 256 / 2 multiply-adds per sweep), the panels below / beside them -- all but 1/16 of the triangle at n = 4096 -- are
 * GEMV-shaped and shared by the OpenMP threads over contiguous column segments. */
static void dense_chol_solve(int64_t n, const double *L, double *b) {
    const int64_t NB = 256;
    /* forward: L y = b */
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        int64_t j1 = j0 + NB < n ? j0 + NB : n;
        for (int64_t j = j0; j < j1; ++j) {                      /* diagonal block, column-oriented (axpy) */
            double xj = b[j] / L[j + j * n];
            b[j] = xj;
            const double *col = L + j * n;
            for (int64_t i = j + 1; i < j1; ++i) b[i] -= col[i] * xj;
        }
        if (j1 < n) {                                            /* b[j1:] -= L[j1:, j0:j1] * y[j0:j1], rows shared by the threads */
#pragma omp parallel for schedule(static) if ((n - j1) * (j1 - j0) > OQ_PAR_MIN / 4)
            for (int64_t i0 = j1; i0 < n; i0 += 512) {
                int64_t i1 = i0 + 512 < n ? i0 + 512 : n;
                for (int64_t j = j0; j < j1; ++j) {
                    const double *col = L + j * n; double xj = b[j];
                    for (int64_t i = i0; i < i1; ++i) b[i] -= col[i] * xj;
                }
            }
        }
    }
    /* backward: L' x = y, dot products over the columns of L */
    for (int64_t j1 = n; j1 > 0; j1 -= NB) {
        int64_t j0 = j1 - NB > 0 ? j1 - NB : 0;
        if (j1 < n) {                                            /* b[j] -= L[j1:, j]' x[j1:] for the columns of this block */
#pragma omp parallel for schedule(static) if ((n - j1) * (j1 - j0) > OQ_PAR_MIN / 4)
            for (int64_t j = j0; j < j1; ++j) {
                const double *col = L + j * n; double s = 0.0;
                for (int64_t i = j1; i < n; ++i) s += col[i] * b[i];
                b[j] -= s;
            }
        }
        for (int64_t j = j1 - 1; j >= j0; --j) {                 /* diagonal block */
            const double *col = L + j * n; double s = b[j];
            for (int64_t i = j + 1; i < j1; ++i) s -= col[i] * b[i];
            b[j] = s / L[j + j * n];
        }
    }
}
/* Dense LDL' without pivoting of a symmetric quasi-definite matrix K (N x N, column-major, lower triangle):
 * on exit the strict lower triangle holds L (unit diagonal implied) and the diagonal holds D.
 * Any symmetric permutation of a quasi-definite matrix has an LDL' factorisation, which is what
 * LinearSystemSolvers.jl:18 ldlt(...), :49 QDLDL.qdldl(...), :81 ldl(...) rely on. */
static int64_t dense_ldlt(int64_t N, double *K) {
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    for (int64_t k = 0; k < N; ++k) {
        double d = K[k + k * N];
        for (int64_t p = 0; p < k; ++p) { w[p] = K[k + p * N] * K[p + p * N]; d -= K[k + p * N] * w[p]; }
        if (d == 0.0 || isnan(d)) { free(w); return k + 1; }
        K[k + k * N] = d;
#pragma omp parallel for schedule(static) if (N - k > 512)
        for (int64_t i = k + 1; i < N; ++i) {
            double s = K[i + k * N];
            for (int64_t p = 0; p < k; ++p) s -= K[i + p * N] * w[p];
            K[i + k * N] = s / d;
        }
    }
    free(w);
    return 0;
}
static void dense_ldlt_solve(int64_t N, const double *K, double *b) {
    for (int64_t j = 0; j < N; ++j) { double xj = b[j]; const double *c = K + j * N; for (int64_t i = j + 1; i < N; ++i) b[i] -= c[i] * xj; }
    for (int64_t j = 0; j < N; ++j) b[j] /= K[j + j * N];
    for (int64_t j = N - 1; j >= 0; --j) { const double *c = K + j * N; double s = b[j]; for (int64_t i = j + 1; i < N; ++i) s -= c[i] * b[i]; b[j] = s; }
}

/* ------------------------------------------------------------------------------------------------
 * Sparse L D L' of a symmetric quasi-definite matrix given by its upper triangle in CSC (0-based), the algorithm QDLDL
 * publishes (Stellato et al., "OSQP", Math. Prog. Comp. 2020, sec. 4; QDLDL.jl is what QDLdlInit calls,
 * LinearSystemSolvers.jl:49): (1) elimination tree and column counts from the upper triangle, (2) up-looking numeric
 * factorisation -- row k of L solves a sparse triangular system whose pattern is the reach of column k in the tree,
 * (3) L x = b, x ./= D, L' x = x.  No pivoting: a quasi-definite matrix has an L D L' for every symmetric permutation.
 * QDLDL.jl is un-vendored and unpinned; its ordering (AMD) is replaced by a caller-supplied permutation, which changes
 * round-off only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t N;
    int64_t *Ap, *Ai; double *Ax;      /* upper triangle (incl. diagonal) of the PERMUTED matrix, CSC, sorted rows */
    int64_t *etree, *Lnz, *Lp, *Li; double *Lx, *D, *Dinv;
    int64_t *perm, *iperm;             /* perm[new] = old */
    double *work;                      /* N doubles: permuted right-hand side */
} sp_ldl;

static void sp_ldl_free(sp_ldl *f) {
    if (!f) return;
    free(f->Ap); free(f->Ai); free(f->Ax); free(f->etree); free(f->Lnz); free(f->Lp); free(f->Li); free(f->Lx); free(f->D); free(f->Dinv);
    free(f->perm); free(f->iperm); free(f->work); free(f);
}
/* elimination tree and the number of off-diagonal entries of every column of L */
static int64_t sp_ldl_etree(sp_ldl *f) {
    int64_t N = f->N, total = 0;
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) { f->etree[j] = -1; f->Lnz[j] = 0; mark[j] = -1; }
    for (int64_t j = 0; j < N; ++j) {
        mark[j] = j;
        for (int64_t p = f->Ap[j]; p < f->Ap[j + 1]; ++p) {
            int64_t i = f->Ai[p];
            while (i < j && mark[i] != j) {           /* walk from i towards the root until a node already seen for column j */
                if (f->etree[i] == -1) f->etree[i] = j;
                f->Lnz[i]++; total++;                  /* L_ji is a non-zero of column i */
                mark[i] = j;
                i = f->etree[i];
            }
        }
    }
    free(mark);
    return total;
}
/* numeric factorisation; returns 0 or 1 + the index of a zero pivot */
static int64_t sp_ldl_numeric(sp_ldl *f) {
    int64_t N = f->N;
    int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);      /* next free slot of every column of L */
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    double *y = (double *)calloc((size_t)N, sizeof(double));
    int64_t fail = 0;
    for (int64_t j = 0; j < N; ++j) { next[j] = f->Lp[j]; mark[j] = -1; }
    for (int64_t k = 0; k < N && !fail; ++k) {
        /* pattern of row k of L = reach of the non-zeros of column k (upper triangle) in the tree, in topological order */
        int64_t top = N; double d = 0.0;
        mark[k] = k;
        for (int64_t p = f->Ap[k]; p < f->Ap[k + 1]; ++p) {
            int64_t i = f->Ai[p];
            if (i == k) { d = f->Ax[p]; continue; }
            y[i] = f->Ax[p];
            int64_t len = 0;
            for (; mark[i] != k; i = f->etree[i]) { path[len++] = i; mark[i] = k; }
            while (len > 0) stack[--top] = path[--len];
        }
        for (int64_t t = top; t < N; ++t) {
            int64_t j = stack[t];
            double yj = y[j]; y[j] = 0.0;
            for (int64_t p = f->Lp[j]; p < next[j]; ++p) y[f->Li[p]] -= f->Lx[p] * yj;   /* rows of column j computed so far */
            double lkj = yj * f->Dinv[j];
            d -= yj * lkj;
            f->Li[next[j]] = k; f->Lx[next[j]] = lkj; next[j]++;
        }
        if (d == 0.0 || isnan(d)) { fail = k + 1; break; }
        f->D[k] = d; f->Dinv[k] = 1.0 / d;
    }
    free(next); free(stack); free(path); free(mark); free(y);
    return fail;
}
static void sp_ldl_solve(const sp_ldl *f, double *b) {     /* b in the ORIGINAL numbering, solved in place */
    int64_t N = f->N; double *x = f->work;
    for (int64_t k = 0; k < N; ++k) x[k] = b[f->perm[k]];
    for (int64_t j = 0; j < N; ++j) { double xj = x[j]; for (int64_t p = f->Lp[j]; p < f->Lp[j + 1]; ++p) x[f->Li[p]] -= f->Lx[p] * xj; }
    for (int64_t j = 0; j < N; ++j) x[j] *= f->Dinv[j];
    for (int64_t j = N - 1; j >= 0; --j) { double s = x[j]; for (int64_t p = f->Lp[j]; p < f->Lp[j + 1]; ++p) s -= f->Lx[p] * x[f->Li[p]]; x[j] = s; }
    for (int64_t k = 0; k < N; ++k) b[f->perm[k]] = x[k];
}
/* symbolic part for a matrix given as (row, col, value) triplets of its upper triangle in the original numbering */
static sp_ldl *sp_ldl_symbolic(int64_t N, int64_t nt, const int64_t *tr, const int64_t *tc, const int64_t *perm) {
    sp_ldl *f = (sp_ldl *)calloc(1, sizeof(sp_ldl));
    f->N = N;
    f->perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)N); f->iperm = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t k = 0; k < N; ++k) { f->perm[k] = perm ? perm[k] : k; f->iperm[f->perm[k]] = k; }
    f->Ap = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t)); f->Ai = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt > 0 ? nt : 1));
    f->Ax = (double *)calloc((size_t)(nt > 0 ? nt : 1), sizeof(double));
    for (int64_t e = 0; e < nt; ++e) { int64_t a = f->iperm[tr[e]], b = f->iperm[tc[e]]; f->Ap[(a > b ? a : b) + 1]++; }
    for (int64_t j = 0; j < N; ++j) f->Ap[j + 1] += f->Ap[j];
    f->etree = (int64_t *)malloc(sizeof(int64_t) * (size_t)N); f->Lnz = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    f->Lp = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
    f->D = (double *)malloc(sizeof(double) * (size_t)N); f->Dinv = (double *)malloc(sizeof(double) * (size_t)N);
    f->work = (double *)malloc(sizeof(double) * (size_t)N);
    return f;
}
/* (re)load the values: triplet e goes to slot pos[e] of the permuted CSC (pos is built on the first call) */
static void sp_ldl_load(sp_ldl *f, int64_t nt, const int64_t *tr, const int64_t *tc, const double *tv, int64_t **pos_io) {
    int64_t N = f->N;
    if (!*pos_io) {
        int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt > 0 ? nt : 1));
        int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
        for (int64_t j = 0; j < N; ++j) next[j] = f->Ap[j];
        for (int64_t e = 0; e < nt; ++e) {
            int64_t a = f->iperm[tr[e]], b = f->iperm[tc[e]]; int64_t col = a > b ? a : b, row = a > b ? b : a;
            pos[e] = next[col]++; f->Ai[pos[e]] = row;
        }
        free(next);
        *pos_io = pos;
        int64_t total = sp_ldl_etree(f);
        for (int64_t j = 0; j < N; ++j) f->Lp[j + 1] = f->Lp[j] + f->Lnz[j];
        f->Li = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
        f->Lx = (double *)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    }
    for (int64_t e = 0; e < nt; ++e) f->Ax[(*pos_io)[e]] = tv[e];
}

/* ------------------------------------------------------------------------------------------------
 * Linear-system plugins.  Interface mirrors the reference plugin pair
 *   Init(vX, mP, vQ, mA, rho, rho1, sigma, n, m) -> (vXX, vZZ, tuSolver)        LinearSystemSolvers.jl:16,24
 *   Sol!(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, rho, rho1, sigma, n, m, changedRho)   :28,42
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int kind;
    const oq_prob *p;
    double *vV;            /* kind 1: (n+m) buffer; vXX/vZZ are views into it (LinearSystemSolvers.jl:19-22) */
    double *vXX, *vZZ;
    double *F;             /* factor: n*n Cholesky (kind 0) or (n+m)^2 LDLt (kind 1) or explicit mL (kind 2) */
    double *mPI, *mAA;     /* cached P + sigma I and A'A (LinearSystemSolvers.jl:112-114) */
    double *vT;            /* RHS buffer (LinearSystemSolvers.jl:115) */
    double *cg_u, *cg_r, *cg_c, *cg_t, *cg_tmp_m; /* CG state (IterativeSolvers CGStateVariables u, r, c) */
    double epsPcg; int32_t numItrPcg;
    int32_t cgIterations;
    int64_t fail;
    sp_ldl *sp; int64_t sp_nt; int64_t *sp_tr, *sp_tc, *sp_pos; double *sp_tv; int64_t sp_rho_first;   /* kind 4: triplets of triu(K); the last m are the -1/rho diagonal */
} oq_linsys;

static const int64_t *g_kkt_perm = NULL;   /* ordering for kind 4 (perm[new] = old over [x; nu]); NULL = natural order */

static void dense_P_dense(const oq_prob *p, double *out) { /* materialise P as dense column-major */
    int64_t n = p->n;
    if (!p->sparse) { memcpy(out, p->P, sizeof(double) * (size_t)(n * n)); return; }
    memset(out, 0, sizeof(double) * (size_t)(n * n));
    for (int64_t j = 0; j < n; ++j) for (int64_t k = p->Pcp[j]; k < p->Pcp[j + 1]; ++k) out[p->Pri[k] + j * n] += p->Pnz[k];
}
static void dense_A_dense(const oq_prob *p, double *out) {
    int64_t n = p->n, m = p->m;
    if (!p->sparse) { memcpy(out, p->A, sizeof(double) * (size_t)(n * m)); return; }
    memset(out, 0, sizeof(double) * (size_t)(n * m));
    for (int64_t j = 0; j < n; ++j) for (int64_t k = p->Acp[j]; k < p->Acp[j + 1]; ++k) out[p->Ari[k] + j * m] += p->Anz[k];
}

/* (Re)build and factor the kind-specific matrix for the given rho. */
static void linsys_factor(oq_linsys *s, double rho, double rho1, double sigma) {
    const oq_prob *p = s->p; int64_t n = p->n, m = p->m;
    if (s->kind == 0 || s->kind == 2) {
        /* LinearSystemSolvers.jl:114 / :128   mL = mPI + rho * mAA  (rebuilt from cached parts on changedRho) */
        for (int64_t k = 0; k < n * n; ++k) s->F[k] = s->mPI[k] + rho * s->mAA[k];
        if (s->kind == 0) s->fail = dense_cholesky(n, s->F);
    } else if (s->kind == 1) {
        /* LinearSystemSolvers.jl:18  [mP + sigma I  mA'; mA  -rho1 I] */
        int64_t N = n + m; double *K = s->F;
        memset(K, 0, sizeof(double) * (size_t)(N * N));
        double *Pd = (double *)malloc(sizeof(double) * (size_t)(n * n)); dense_P_dense(p, Pd);
        double *Ad = (double *)malloc(sizeof(double) * (size_t)(n * m > 0 ? n * m : 1)); dense_A_dense(p, Ad);
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t i = j; i < n; ++i) K[i + j * N] = Pd[i + j * n];
            K[j + j * N] += sigma;
            for (int64_t i = 0; i < m; ++i) K[(n + i) + j * N] = Ad[i + j * m];
        }
        for (int64_t i = 0; i < m; ++i) K[(n + i) + (n + i) * N] = -rho1;
        free(Pd); free(Ad);
        s->fail = dense_ldlt(N, K);
    } else if (s->kind == 4) {
        /* LinearSystemSolvers.jl:49 / :62  QDLDL.qdldl([mP + sigma I  mA'; mA  -rho1 I]): only the last m triplets change with rho */
        for (int64_t i = 0; i < m; ++i) s->sp_tv[s->sp_rho_first + i] = -rho1;
        sp_ldl_load(s->sp, s->sp_nt, s->sp_tr, s->sp_tc, s->sp_tv, &s->sp_pos);
        s->fail = sp_ldl_numeric(s->sp);
    }
    (void)sigma;
}

static oq_linsys *linsys_init(int kind, const oq_prob *p, double rho, double rho1, double sigma, double epsPcg, int32_t numItrPcg) {
    oq_linsys *s = (oq_linsys *)calloc(1, sizeof(oq_linsys));
    int64_t n = p->n, m = p->m;
    s->kind = kind; s->p = p; s->epsPcg = epsPcg; s->numItrPcg = numItrPcg;
    if (kind == 4) {
        /* buffers are allocated with the triplets below */
    } else if (kind == 1) {
        s->vV = (double *)calloc((size_t)(n + m), sizeof(double));              /* :19 zeros(m + n) */
        s->vXX = s->vV; s->vZZ = s->vV + n;                                       /* :21-22 views */
        s->F = (double *)malloc(sizeof(double) * (size_t)((n + m) * (n + m)));
    } else {
        s->vXX = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));     /* :117 zeros(n) */
        s->vZZ = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));     /* :118 zeros(m) */
        s->vT = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));      /* :115 */
    }
    if (kind == 0 || kind == 2) {
        s->mPI = (double *)malloc(sizeof(double) * (size_t)(n * n));
        s->mAA = (double *)malloc(sizeof(double) * (size_t)(n * n));
        s->F = (double *)malloc(sizeof(double) * (size_t)(n * n));
        dense_P_dense(p, s->mPI);
        for (int64_t j = 0; j < n; ++j) s->mPI[j + j * n] += sigma;              /* :113 mPI = mP + sigma I */
        if (m > 0) {
            if (!p->sparse) dense_AtA(m, n, p->A, s->mAA);                         /* :112 mAA = mA' mA */
            else { double *Ad = (double *)malloc(sizeof(double) * (size_t)(n * m)); dense_A_dense(p, Ad); dense_AtA(m, n, Ad, s->mAA); free(Ad); }
        } else memset(s->mAA, 0, sizeof(double) * (size_t)(n * n));
    }
    if (kind == 4) {
        /* triplets of the upper triangle of K in the [x; nu] numbering: triu(P) + sigma I, A' (rows < n, columns >= n), -rho1 I */
        int64_t cap = n + m + 1; 
        if (p->sparse) cap += p->Pcp[n] + p->Acp[n]; else cap += n * n + n * m;
        s->sp_tr = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap); s->sp_tc = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        s->sp_tv = (double *)malloc(sizeof(double) * (size_t)cap);
        int64_t nt = 0;
        double *dg = (double *)calloc((size_t)n, sizeof(double));
        if (p->sparse) {
            for (int64_t j = 0; j < n; ++j) for (int64_t k = p->Pcp[j]; k < p->Pcp[j + 1]; ++k) {
                int64_t i = p->Pri[k];
                if (i == j) dg[j] += p->Pnz[k]; else if (i < j) { s->sp_tr[nt] = i; s->sp_tc[nt] = j; s->sp_tv[nt] = p->Pnz[k]; nt++; }
            }
            for (int64_t j = 0; j < n; ++j) for (int64_t k = p->Acp[j]; k < p->Acp[j + 1]; ++k) { s->sp_tr[nt] = j; s->sp_tc[nt] = n + p->Ari[k]; s->sp_tv[nt] = p->Anz[k]; nt++; }
        } else {
            for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i <= j; ++i) {
                double v = p->P[i + j * n];
                if (i == j) dg[j] = v; else if (v != 0.0) { s->sp_tr[nt] = i; s->sp_tc[nt] = j; s->sp_tv[nt] = v; nt++; }
            }
            for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < m; ++i) { double v = p->A[i + j * m]; if (v != 0.0) { s->sp_tr[nt] = j; s->sp_tc[nt] = n + i; s->sp_tv[nt] = v; nt++; } }
        }
        for (int64_t j = 0; j < n; ++j) { s->sp_tr[nt] = j; s->sp_tc[nt] = j; s->sp_tv[nt] = dg[j] + sigma; nt++; }
        s->sp_rho_first = nt;
        for (int64_t i = 0; i < m; ++i) { s->sp_tr[nt] = n + i; s->sp_tc[nt] = n + i; s->sp_tv[nt] = -rho1; nt++; }
        free(dg);
        s->sp_nt = nt;
        s->sp = sp_ldl_symbolic(n + m, nt, s->sp_tr, s->sp_tc, g_kkt_perm);
        s->vV = (double *)calloc((size_t)(n + m), sizeof(double));               /* :50 zeros(m + n); vXX / vZZ are views (:52-53) */
        s->vXX = s->vV; s->vZZ = s->vV + n;
    }
    if (kind == 2 || kind == 3) {
        s->cg_u = (double *)calloc((size_t)n, sizeof(double));
        s->cg_r = (double *)calloc((size_t)n, sizeof(double));
        s->cg_c = (double *)calloc((size_t)n, sizeof(double));
        s->cg_t = (double *)calloc((size_t)n, sizeof(double));
        s->cg_tmp_m = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    }
    linsys_factor(s, rho, rho1, sigma);
    return s;
}
static void linsys_free(oq_linsys *s) {
    if (!s) return;
    if (s->kind == 1 || s->kind == 4) free(s->vV); else { free(s->vXX); free(s->vZZ); free(s->vT); }
    sp_ldl_free(s->sp); free(s->sp_tr); free(s->sp_tc); free(s->sp_tv); free(s->sp_pos);
    free(s->F); free(s->mPI); free(s->mAA); free(s->cg_u); free(s->cg_r); free(s->cg_c); free(s->cg_t); free(s->cg_tmp_m);
    free(s);
}

/* u = mL w.  kind 2: explicit symmetric matrix.  kind 3: LinearSystemSolvers.jl:152-157
 *   mul!(vZZ, mA, vW); mul!(vU, mA', vZZ); mul!(vU, mP, vW, 1.0, rho)  [vU = P w + rho vU]; vU .+= sigma .* vW */
static void reduced_op(oq_linsys *s, double rho, double sigma, const double *w, double *u) {
    const oq_prob *p = s->p; int64_t n = p->n;
    if (s->kind == 2) {
#pragma omp parallel for schedule(static) if (n * n > OQ_PAR_MIN)
        for (int64_t i = 0; i < n; ++i) { const double *col = s->F + i * n; double acc = 0.0; for (int64_t j = 0; j < n; ++j) acc += col[j] * w[j]; u[i] = acc; }
    } else {
        mul_A(p, w, s->cg_tmp_m);
        mul_At(p, s->cg_tmp_m, u);
        /* P w accumulated on top of rho * A'(A w) */
        double *t = s->cg_t;
        mul_P(p, w, t);
        for (int64_t i = 0; i < n; ++i) u[i] = t[i] + rho * u[i] + sigma * w[i];
    }
}
/* IterativeSolvers.cg!(x, A, b; abstol, maxiter) with x warm-started (initially_zero = false):
 *   r = b - A x; residual = ||r||; tol = max(reltol * residual, abstol), reltol = sqrt(eps(Float64));
 *   while residual > tol && it < maxiter: beta = residual^2 / prev_residual^2 (prev = 1, u = 0 initially);
 *   u = r + beta u; c = A u; alpha = residual^2 / dot(u, c); x += alpha u; r -= alpha c; residual = ||r||.
 * Call site: LinearSystemSolvers.jl:137 (and :181, :224). */
static void cg_solve(oq_linsys *s, double rho, double sigma, double *x, const double *b) {
    int64_t n = s->p->n; double *u = s->cg_u, *r = s->cg_r, *c = s->cg_c;
    for (int64_t i = 0; i < n; ++i) u[i] = 0.0;
    reduced_op(s, rho, sigma, x, c);
    double res2 = 0.0;
    for (int64_t i = 0; i < n; ++i) { r[i] = b[i] - c[i]; res2 += r[i] * r[i]; }
    double residual = sqrt(res2), prev = 1.0;
    double reltol = sqrt(2.220446049250313e-16);
    double tol = fmax(reltol * residual, s->epsPcg);
    int32_t it = 0;
    while (!(residual <= tol) && it < s->numItrPcg) {
        double beta = (residual * residual) / (prev * prev);
        for (int64_t i = 0; i < n; ++i) u[i] = r[i] + beta * u[i];
        reduced_op(s, rho, sigma, u, c);
        double uc = 0.0; for (int64_t i = 0; i < n; ++i) uc += u[i] * c[i];
        double alpha = (residual * residual) / uc;
        res2 = 0.0;
        for (int64_t i = 0; i < n; ++i) { x[i] += alpha * u[i]; r[i] -= alpha * c[i]; res2 += r[i] * r[i]; }
        prev = residual; residual = sqrt(res2);
        ++it;
    }
    s->cgIterations += it;
}

/* The Sol! half of the plugin pair. */
static void linsys_solve(oq_linsys *s, const double *vX, const double *vQ, const double *vZ, const double *vY,
                         double rho, double rho1, double sigma, int changedRho) {
    const oq_prob *p = s->p; int64_t n = p->n, m = p->m;
    if (changedRho) linsys_factor(s, rho, rho1, sigma);   /* :30-32, :61-63, :93-95, :127-129 */
    if (s->kind == 1 || s->kind == 4) {
        for (int64_t i = 0; i < n; ++i) s->vXX[i] = sigma * vX[i] - vQ[i];        /* :37 */
        for (int64_t i = 0; i < m; ++i) s->vZZ[i] = vZ[i] - rho1 * vY[i];         /* :38 */
        if (s->kind == 1) dense_ldlt_solve(n + m, s->F, s->vV);                   /* :39 / :70 / :102 */
        else sp_ldl_solve(s->sp, s->vV);                                          /* :70 QDLDL.solve!(hDL, vV) */
        for (int64_t i = 0; i < m; ++i) s->vZZ[i] = vZ[i] + rho1 * (s->vZZ[i] - vY[i]); /* :40 */
    } else {
        for (int64_t i = 0; i < m; ++i) s->vZZ[i] = rho * vZ[i] - vY[i];          /* :134 vZZ used as buffer */
        mul_At(p, s->vZZ, s->vT);                                                 /* :135 */
        for (int64_t i = 0; i < n; ++i) s->vT[i] = sigma * vX[i] - vQ[i] + s->vT[i]; /* :136 */
        if (s->kind == 0) { memcpy(s->vXX, s->vT, sizeof(double) * (size_t)n); dense_chol_solve(n, s->F, s->vXX); }
        else cg_solve(s, rho, sigma, s->vXX, s->vT);                              /* :137 warm-started from previous vXX */
        mul_A(p, s->vXX, s->vZZ);                                                 /* :139 */
    }
}

/* SolveQuadraticProgram.jl:79-112 */
static void check_convergence(const oq_prob *p, const double *vX, const double *vQ, const double *vZ, const double *vY,
                              const double *vXP, const double *vZP, double rho, double *rhorho, int adptRho,
                              double epsAbs, double epsRel, double epsAdmm, int32_t *convFlag, oq_info *info,
                              double *wAx, double *wPx, double *wAty, double *wn) {
    const double MIN_VAL_RHO = 1e-3, MAX_VAL_RHO = 1e6;             /* :81-82 */
    int64_t n = p->n, m = p->m;
    mul_A(p, vX, wAx); mul_P(p, vX, wPx); mul_At(p, vY, wAty);
    double normResPrim = norm_inf_diff(wAx, vZ, m);                  /* :85 */
    for (int64_t i = 0; i < n; ++i) wn[i] = wPx[i] + vQ[i] + wAty[i];
    double normResDual = norm_inf(wn, n);                            /* :86 */
    double maxNormPrim = jmax(norm_inf(wAx, m), norm_inf(vZ, m));    /* :88 */
    double maxNormDual = jmax(jmax(norm_inf(wPx, n), norm_inf(wAty, n)), norm_inf(vQ, n)); /* :89 */
    if (adptRho) {                                                   /* :92-96 */
        double numeratorVal = normResPrim * maxNormDual;
        double denominatorVal = normResDual * maxNormPrim;
        *rhorho = jclamp(rho * sqrt(numeratorVal / denominatorVal), MIN_VAL_RHO, MAX_VAL_RHO);
    }
    double epsPrim = epsAbs + epsRel * maxNormPrim;                  /* :99 */
    double epsDual = epsAbs + epsRel * maxNormDual;                  /* :100 */
    if ((normResPrim < epsPrim) && (normResDual < epsDual)) *convFlag = convPrimDual;       /* :102-104 */
    if ((norm_inf_diff(vX, vXP, n) <= epsAdmm) && (norm_inf_diff(vZ, vZP, m) <= epsAdmm)) *convFlag = convAdmm; /* :105-107, not else */
    info->resPrim = normResPrim; info->resDual = normResDual; info->maxNormPrim = maxNormPrim; info->maxNormDual = maxNormDual;
}

/* SolveQuadraticProgram.jl:14-76 */
static int32_t solve_core(const oq_prob *p, const double *vQ, const double *vL, const double *vU, double *vX,
                          const oq_params *prm, oq_info *info, double *z_out, double *y_out) {
    int64_t n = p->n, m = p->m;
#ifdef _OPENMP
    if (prm->numThreads > 0) omp_set_num_threads(prm->numThreads);
#endif
    double rho = prm->rho, sigma = prm->sigma, alpha = prm->alpha;
    double rho1 = 1.0 / rho;                                   /* :30 */
    double alpha1 = 1.0 - alpha;                               /* :31 */
    int32_t convFlag = convNumItr;                             /* :33 */
    double epsAdmm = fmin(prm->epsAbs, prm->epsRel) * 1e-2;    /* :34 */
    memset(info, 0, sizeof(*info));
    double t0 = now_sec();
    oq_linsys *s = linsys_init(prm->linsys, p, rho, rho1, sigma, prm->epsPcg, prm->numItrPcg); /* :36 */
    double t1 = now_sec();
#ifdef _OPENMP
    if (prm->loopThreads > 0) omp_set_num_threads(prm->loopThreads);
#endif
    if (s->fail) { info->convFlag = -(int32_t)s->fail; linsys_free(s); return -1; }
    size_t mm = (size_t)(m > 0 ? m : 1), nn = (size_t)(n > 0 ? n : 1);
    double *vXP = (double *)calloc(nn, sizeof(double));        /* :38 */
    double *vZ = (double *)calloc(mm, sizeof(double));         /* :39 */
    double *vY = (double *)calloc(mm, sizeof(double));         /* :40 */
    double *vZP = (double *)calloc(mm, sizeof(double));        /* :41 */
    double *wAx = (double *)calloc(mm, sizeof(double)), *wPx = (double *)calloc(nn, sizeof(double));
    double *wAty = (double *)calloc(nn, sizeof(double)), *wn = (double *)calloc(nn, sizeof(double));
    double rhorho = rho;                                       /* :43 */
    int32_t ii = 0, nref = 0;
    for (ii = 1; ii <= prm->numIterations; ++ii) {             /* :45 */
        int changedRho = 0;
        if (prm->adptRho && ((rhorho * prm->fctrRho < rho) || (rhorho > prm->fctrRho * rho))) {  /* :47 */
            rho = rhorho; rho1 = 1.0 / rho; changedRho = 1; ++nref;                              /* :48-51 */
        }
        linsys_solve(s, vX, vQ, vZ, vY, rho, rho1, sigma, changedRho);                           /* :54 */
        if (s->fail) { convFlag = -(int32_t)s->fail; break; }
        const double *vXX = s->vXX, *vZZ = s->vZZ;
        memcpy(vXP, vX, sizeof(double) * (size_t)n);                                             /* :56 */
        for (int64_t i = 0; i < n; ++i) vX[i] = alpha * vXX[i] + alpha1 * vX[i];                 /* :57 */
        memcpy(vZP, vZ, sizeof(double) * (size_t)m);                                             /* :59 */
        for (int64_t i = 0; i < m; ++i) vZ[i] = jclamp(alpha * vZZ[i] + alpha1 * vZ[i] + rho1 * vY[i], vL[i], vU[i]); /* :60 */
        for (int64_t i = 0; i < m; ++i) vY[i] = vY[i] + rho * (alpha * vZZ[i] + alpha1 * vZP[i] - vZ[i]);            /* :61 */
        if (ii % prm->numItrConv == 0) {                                                         /* :63 */
            check_convergence(p, vX, vQ, vZ, vY, vXP, vZP, rho, &rhorho, prm->adptRho, prm->epsAbs, prm->epsRel,
                              epsAdmm, &convFlag, info, wAx, wPx, wAty, wn);                     /* :64 */
            if (convFlag != convNumItr) break;                                                   /* :66-68 */
        }
    }
    double t2 = now_sec();
    info->convFlag = convFlag;
    info->iterations = ii > prm->numIterations ? prm->numIterations : ii;
    info->numRefactor = nref; info->cgIterations = s->cgIterations;
    info->rhoFinal = rho; info->rhoProposed = rhorho; info->tSetup = t1 - t0; info->tLoop = t2 - t1;
    if (z_out) memcpy(z_out, vZ, sizeof(double) * (size_t)m);
    if (y_out) memcpy(y_out, vY, sizeof(double) * (size_t)m);
    free(vXP); free(vZ); free(vY); free(vZP); free(wAx); free(wPx); free(wAty); free(wn);
    linsys_free(s);
    return convFlag;                                           /* :73 */
}

OQ_EXPORT int32_t oq_solve_dense(int64_t n, int64_t m, const double *P, const double *q, const double *A,
                                 const double *l, const double *u, double *x, const oq_params *prm, oq_info *info,
                                 double *z_out, double *y_out) {
    oq_prob p; memset(&p, 0, sizeof(p)); p.n = n; p.m = m; p.sparse = 0; p.P = P; p.A = A;
    return solve_core(&p, q, l, u, x, prm, info, z_out, y_out);
}
OQ_EXPORT int32_t oq_solve_csc(int64_t n, int64_t m, const int64_t *Pcp, const int64_t *Pri, const double *Pnz,
                               const int64_t *Acp, const int64_t *Ari, const double *Anz, const double *q,
                               const double *l, const double *u, double *x, const oq_params *prm, oq_info *info,
                               double *z_out, double *y_out) {
    oq_prob p; memset(&p, 0, sizeof(p)); p.n = n; p.m = m; p.sparse = 1;
    p.Pcp = Pcp; p.Pri = Pri; p.Pnz = Pnz; p.Acp = Acp; p.Ari = Ari; p.Anz = Anz;
    return solve_core(&p, q, l, u, x, prm, info, z_out, y_out);
}

/* The plugin pair exposed on its own (dense inputs), so a test can drive Init/Sol! step by step. */
typedef struct { oq_prob p; oq_linsys *s; double *q; } oq_plugin;
OQ_EXPORT void *oq_linsys_init_dense(int32_t kind, int64_t n, int64_t m, const double *P, const double *q, const double *A,
                                     double rho, double sigma) {
    oq_plugin *h = (oq_plugin *)calloc(1, sizeof(oq_plugin));
    h->p.n = n; h->p.m = m; h->p.P = P; h->p.A = A;
    h->q = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); memcpy(h->q, q, sizeof(double) * (size_t)n);
    h->s = linsys_init(kind, &h->p, rho, 1.0 / rho, sigma, 1e-6, 1000);
    return h;
}
OQ_EXPORT int32_t oq_linsys_solve_dense(void *hh, const double *x, const double *z, const double *y, double rho, double sigma,
                                        int32_t changedRho, double *xx_out, double *zz_out) {
    oq_plugin *h = (oq_plugin *)hh;
    linsys_solve(h->s, x, h->q, z, y, rho, 1.0 / rho, sigma, changedRho);
    memcpy(xx_out, h->s->vXX, sizeof(double) * (size_t)h->p.n);
    memcpy(zz_out, h->s->vZZ, sizeof(double) * (size_t)h->p.m);
    return (int32_t)h->s->fail;
}
OQ_EXPORT void oq_linsys_free(void *hh) { oq_plugin *h = (oq_plugin *)hh; if (!h) return; linsys_free(h->s); free(h->q); free(h); }

/* ordering used by linsys kind 4 (perm[new] = old over [x; nu], n + m entries; NULL = natural order); the array must stay alive */
OQ_EXPORT void oq_set_kkt_perm(const int64_t *perm) { g_kkt_perm = perm; }

OQ_EXPORT int32_t oq_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
OQ_EXPORT const char *oq_version(void) { return "qps-oracle 0.1 (CPU restatement; parity unpinned)"; }
