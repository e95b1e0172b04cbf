/* Pure-C consumer of the drop-in boundary (include/qps.h): no Python, no HIP headers, no C++.
 * Solves   min 1/2 x'Px + q'x  s.t. l <= Ax <= u   for a small diagonal-box problem with a known answer
 * (x* = clamp(-q./p, l, u)), once through qps_solve and once through the literal plugin pair
 * qps_linsys_init / qps_linsys_solve, and prints the result.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qps.h"

int main(void) {
    enum { N = 6 };
    double P[N * N] = {0}, A[N * N] = {0}, q[N], l[N], u[N], x[N] = {0}, xstar[N];
    for (int i = 0; i < N; ++i) {
        const double p = 1.0 + 0.5 * i;
        P[i + i * N] = p; A[i + i * N] = 1.0;                       /* column-major, like a Julia Matrix{Float64} */
        q[i] = (i % 2 ? 1.0 : -1.0) * (1.0 + i); l[i] = -0.75; u[i] = 0.5;
        double t = -q[i] / p; xstar[i] = t > u[i] ? u[i] : (t < l[i] ? l[i] : t);
    }
    if (qps_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", qps_version()); return 2; }
    qps_handle h = NULL;
    int rc = qps_create_dense(N, N, P, N, A, N, q, l, u, QPS_F64, 0, &h);
    if (rc != QPS_OK) { fprintf(stderr, "create failed: %s\n", qps_last_error(NULL)); return 1; }
    qps_params prm; qps_info info;
    qps_default_params(&prm);
    prm.numIterations = 50000; prm.epsAbs = 1e-7; prm.epsRel = 1e-7; prm.rho = 0.1; prm.adptRho = 1;   /* RunTests.jl:50-54 */
    rc = qps_solve(h, x, &prm, &info);
    if (rc != QPS_OK) { fprintf(stderr, "solve failed: %s\n", qps_last_error(h)); return 1; }
    double dev = 0.0;
    for (int i = 0; i < N; ++i) dev = fmax(dev, fabs(x[i] - xstar[i]));
    printf("flag=%d iterations=%d max|x-x*|=%.3e rho_final=%g\n", info.convFlag, info.iterations, dev, info.rhoFinal);
    int ok = (info.convFlag == QPS_CONV_PRIM_DUAL || info.convFlag == QPS_CONV_ADMM) && dev <= 1e-5;
    /* the plugin pair: x~ solves (P + sigma I + rho A'A) x~ = sigma x - q + A'(rho z - y), z~ = A x~ */
    double z[N], y[N], xx[N], zz[N], xin[N];
    for (int i = 0; i < N; ++i) { xin[i] = 0.1 * i; z[i] = 0.2 - 0.05 * i; y[i] = 0.03 * i; }
    rc = qps_linsys_init(h, 0.7, 1e-6, QPS_LINSYS_CHOLESKY, 0);
    if (rc == QPS_OK) rc = qps_linsys_solve(h, xin, z, y, 0.7, 1e-6, 0, xx, zz);
    if (rc != QPS_OK) { fprintf(stderr, "linsys failed: %s\n", qps_last_error(h)); return 1; }
    double res = 0.0;
    for (int i = 0; i < N; ++i) {
        const double lhs = (P[i + i * N] + 1e-6 + 0.7) * xx[i];                 /* A = I, P diagonal */
        const double rhs = 1e-6 * xin[i] - q[i] + (0.7 * z[i] - y[i]);
        res = fmax(res, fabs(lhs - rhs)); res = fmax(res, fabs(zz[i] - xx[i]));
    }
    printf("plugin pair residual=%.3e\n", res);
    ok = ok && res <= 1e-12;
    /* polishing step alone (SolveQuadraticProgram.m:289-325): multiplier signs of the known answer -> exact KKT point */
    {
        double yk[N], xp[N]; qps_polish_report prep;
        for (int i = 0; i < N; ++i) {                                          /* y = -(P x* + q) on clamped rows, 0 elsewhere (A = I) */
            const double g = P[i + i * N] * xstar[i] + q[i];
            yk[i] = (xstar[i] == u[i] || xstar[i] == l[i]) ? -g : 0.0; xp[i] = x[i];
        }
        qps_params pp; qps_default_params(&pp); pp.epsMinres = 1e-12; pp.numItrMinres = 200;
        rc = qps_polish(h, xp, yk, &pp, &prep);
        if (rc != QPS_OK) { fprintf(stderr, "polish failed: %s\n", qps_last_error(h)); return 1; }
        double dp = 0.0;
        for (int i = 0; i < N; ++i) dp = fmax(dp, fabs(xp[i] - xstar[i]));
        printf("polish flag=%d active=%d+%d minres=%d max|x-x*|=%.3e\n", prep.flag, prep.numActiveLower, prep.numActiveUpper, prep.minresIterations, dp);
        ok = ok && prep.flag == 0 && dp <= 1e-6;   /* delta = 1e-6 regularisation, ten refinement steps */
    }
    /* second solver form (ProxQP.jl): min 1/2 x'Px + q'x  s.t.  x_0 + x_1 = 1 (A x = b),  x <= 0.5 (C x <= d) */
    {
        double Aeq[N] = {1.0, 1.0, 0, 0, 0, 0}, beq[1] = {1.0}, dd[N], xs[N], ys[1], zs[N], ss[N];
        for (int i = 0; i < N; ++i) dd[i] = 0.5;
        qps_handle hp = NULL; qps_proxqp_params qp; qps_proxqp_report qr;
        rc = qps_proxqp_create_dense(N, 1, N, P, N, q, Aeq, 1, beq, A, N, dd, QPS_F64, 0, &hp);
        if (rc == QPS_OK) rc = qps_proxqp_init_kkt(hp);
        qps_proxqp_default_params(&qp); qp.numIterations = 3000;
        if (rc == QPS_OK) rc = qps_proxqp_solve(hp, &qp, &qr);
        if (rc == QPS_OK) rc = qps_proxqp_get_state(hp, xs, ys, zs, ss);
        if (rc != QPS_OK) { fprintf(stderr, "proxqp failed: %s\n", qps_last_error(hp)); return 1; }
        double viol = fabs(xs[0] + xs[1] - 1.0);
        for (int i = 0; i < N; ++i) viol = fmax(viol, xs[i] - 0.5);
        printf("proxqp converged=%d iterations=%d violation=%.3e\n", qr.converged, qr.iterations, viol);
        ok = ok && qr.converged && viol <= 1e-5;
        qps_destroy(hp);
    }
    /* the same known-answer problem as Julia-style CSC arrays (1-based Int64 colptr / rowval) through the sparse direct KKT plugin
     * (QPS_LINSYS_KKT_LDL: LaLdl / QDLdl / FacLdl of LinearSystemSolvers.jl:16-107), after the host-only symbolic analysis */
    {
        int64_t cp[N + 1], ri[N], perm[2 * N]; double pv[N], av[N], xs[N] = {0};
        for (int i = 0; i < N; ++i) { cp[i] = i + 1; ri[i] = i + 1; pv[i] = 1.0 + 0.5 * i; av[i] = 1.0; }
        cp[N] = N + 1;
        qps_ldl_report lr;
        rc = qps_ldl_analyze(N, N, cp, ri, cp, ri, 1, perm, &lr);
        if (rc != QPS_OK) { fprintf(stderr, "ldl_analyze failed: %s\n", qps_last_error(NULL)); return 1; }
        ok = ok && lr.numRows == 2 * N && lr.numSparseColumns + lr.tailSize == 2 * N && lr.nnzK == N && lr.nnzL == N;   /* K = [D I; I -I/rho]: no fill */
        ok = ok && qps_linsys_auto(N, N, N, N, 1) == QPS_LINSYS_KKT_LDL && qps_linsys_auto(3000, 2001, N, N, 1) == QPS_LINSYS_CG;
        qps_handle hs = NULL;
        rc = qps_create_csc(N, N, cp, ri, pv, cp, ri, av, q, l, u, 1, /*dense_path*/0, QPS_F64, 0, &hs);
        if (rc != QPS_OK) { fprintf(stderr, "create_csc failed: %s\n", qps_last_error(NULL)); return 1; }
        qps_params ps = prm; ps.linsys = QPS_LINSYS_KKT_LDL; ps.rho = 0.1; ps.adptRho = 1;
        rc = qps_solve(hs, xs, &ps, &info);
        if (rc != QPS_OK) { fprintf(stderr, "ldl solve failed: %s\n", qps_last_error(hs)); return 1; }
        double dl = 0.0;
        for (int i = 0; i < N; ++i) dl = fmax(dl, fabs(xs[i] - xstar[i]));
        printf("kkt ldl: flag=%d iterations=%d refactor=%d max|x-x*|=%.3e (sparse columns %lld, dense tail %lld)\n", info.convFlag, info.iterations,
               info.numRefactor, dl, (long long)lr.numSparseColumns, (long long)lr.tailSize);
        ok = ok && dl <= 1e-5 && info.cgIterations == 0;
        /* an asymmetric P is refused like MATLAB's issymmetric check (SolveQuadraticProgram.m:166-168) */
        double Pbad[N * N]; for (int i = 0; i < N * N; ++i) Pbad[i] = P[i];
        Pbad[1] = 1e-3;                                                       /* (1, 0) without its mirror image */
        qps_handle hx = NULL;
        ok = ok && qps_create_dense(N, N, Pbad, N, A, N, q, l, u, QPS_F64, 0, &hx) == QPS_ERR_BAD_ARGUMENT && hx == NULL;
        qps_destroy(hs);
    }
    /* error path: a non-positive-definite problem must be reported, not hidden */
    qps_handle hb = NULL;
    for (int i = 0; i < N; ++i) P[i + i * N] = -1.0;
    rc = qps_create_dense(N, N, P, N, A, N, q, l, u, QPS_F64, 0, &hb);
    if (rc == QPS_OK) { prm.rho = 1e-3; prm.adptRho = 0; rc = qps_solve(hb, x, &prm, NULL); printf("indefinite P -> status %d (%s)\n", rc, qps_last_error(hb)); ok = ok && rc == QPS_ERR_FACTORIZATION; }
    qps_destroy(hb);
    qps_destroy(h);
    printf(ok ? "C ABI example OK\n" : "C ABI example FAILED\n");
    return ok ? 0 : 1;
}
