/*
 * qps.h -- C ABI of the MI355X-native ADMM QP solver (libqps_hip.so).
 *
 * Drop-in boundary for ONE path of RoyiAvital/QuadraticProgramSolver (citations are file:line under the
 * reference repository):
 *
 *   SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, LinSysSolInit, LinSysSol!; kw...) -> ConvergenceFlag
 *                                                     SolveQuadraticProgram.jl:14-76   -> qps_create_* + qps_solve
 *   CheckConvergence(...)                             SolveQuadraticProgram.jl:79-112  -> inside qps_solve (device)
 *   LinSysSolInit(vX,mP,vQ,mA,rho,rho1,sigma,n,m)     LinearSystemSolvers.jl:16,47,78,110,145 -> qps_linsys_init
 *   LinSysSol!(tuSolver,vXX,vZZ,vX,...,changedRho)    LinearSystemSolvers.jl:28,59,91,125,164 -> qps_linsys_solve
 *   @enum ConvergenceFlag                             SolveQuadraticProgram.jl:12      -> qps_conv_flag
 *   polishing block (MATLAB implementation only)      SolveQuadraticProgram.m:289-325  -> qps_polish, qps_params.polish
 *   ProxQP(mP,vQ,mA,vB,mC,vD[,vX,vY,vZ,vS]) + SolveQuadraticProgram!(sQpProb; kw...) -> dReport
 *                                                     ProxQP.jl:36,73-93,118-173      -> qps_proxqp_*
 *
 * Everything crossing this boundary is a plain pointer, size or scalar.  Host arrays stay owned by the caller and may
 * be freed as soon as the call that received them returns (qps_create_* copies the problem into HBM).
 * A handle is bound to one device and one HIP stream; distinct handles -- on one device or on several devices of the process --
 * may be driven from distinct host threads (per-device state of the library is keyed by device ordinal), a
 * single handle is not thread-safe.  All functions return a qps_status (0 = ok).
 */
#ifndef QPS_H
#define QPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPS_VERSION_MAJOR 0
#define QPS_VERSION_MINOR 1

typedef struct qps_solver_s *qps_handle;

typedef enum {
    QPS_OK = 0,
    QPS_ERR_BAD_ARGUMENT = 1,     /* null pointer, negative size, unknown enum value                             */
    QPS_ERR_BAD_DIMENSION = 2,    /* mirrors the dimension checks of SolveQuadraticProgram.m:158-184             */
    QPS_ERR_NOT_FINITE = 3,       /* NaN/Inf in P, q, A or NaN in l/u (l,u may be +-Inf)                         */
    QPS_ERR_FACTORIZATION = 4,    /* Cholesky breakdown (non-positive pivot); qps_last_error names the column    */
    QPS_ERR_HIP = 5,              /* a HIP runtime call failed; qps_last_error carries hipGetErrorString         */
    QPS_ERR_OUT_OF_MEMORY = 6,
    QPS_ERR_NO_DEVICE = 7,        /* no gfx950 device visible: the library never falls back to the CPU           */
    QPS_ERR_UNSUPPORTED = 8
} qps_status;

/* SolveQuadraticProgram.jl:12  @enum ConvergenceFlag convNumItr = 1 convAdmm convPrimDual */
typedef enum { QPS_CONV_NUM_ITR = 1, QPS_CONV_ADMM = 2, QPS_CONV_PRIM_DUAL = 3 } qps_conv_flag;

/* arithmetic type of the device-resident loop */
typedef enum { QPS_F64 = 0, QPS_F32 = 1 } qps_dtype;

/* which linear-system path replaces the reference plugin pair */
typedef enum {
    QPS_LINSYS_AUTO = 0,        /* dense handle -> QPS_LINSYS_CHOLESKY; CSC handle -> the reference's modeAuto rule on the handle's
                                   sizes (qps_linsys_auto: KKT_LDL or CG), CG when the direct factor does not fit the device plugin */
    QPS_LINSYS_CHOLESKY = 1,    /* reduced form P + sigma I + rho A'A, one Cholesky + two triangular sweeps / it   */
    QPS_LINSYS_CG = 2,          /* matrix-free CG on the reduced operator (LinearSystemSolvers.jl:145-186)         */
    QPS_LINSYS_KKT_LDL = 3,     /* sparse L D L' of the quasi-definite KKT matrix [P + sigma I  A'; A  -I/rho], the reference's
                                   direct plugins LaLdl / QDLdl / FacLdl (LinearSystemSolvers.jl:16-107): minimum-degree ordering
                                   and symbolic factor once per handle, numeric-only re-factorisation on a rho switch, two
                                   level-scheduled sparse triangular solves per iteration.  CSC handles created with dense_path = 0 */
    QPS_LINSYS_CG_EXPLICIT = 4  /* ItrSolCgInit / ItrSolCg! (LinearSystemSolvers.jl:110-142): cg! on the EXPLICIT reduced matrix mL = mPI + rho mAA,
                                   mAA = mA'mA and mPI = mP + sigma I formed once per handle (:112-114), mL rebuilt from the cached parts on a rho switch
                                   (:127-129), ONE product per CG iteration (:137).  CSC handles.  A QPS_LINSYS_CG (or AUTO -> CG) request takes this
                                   plugin by itself when the matrix pays: mA'mA cheap to form and mL no larger than 1.5 x (nnz P + 2 nnz A) -- isotonic
                                   regression, banded / control-like mA; otherwise, and always with QPS_CG_EXPLICIT=0 in the environment, it stays
                                   matrix-free.  Same iterates up to rounding either way (the oracle restates both: linsys kinds 2 and 3) */
} qps_linsys_kind;

/* Keyword arguments of SolveQuadraticProgram! (SolveQuadraticProgram.jl:15-17), same names, same defaults.
 * delta, numItrPolish, epsMinres, numItrMinres are accepted and, exactly as in the reference, unused by the loop; they drive the
 * polishing step of the MATLAB implementation when `polish` is set (see qps_polish below). */
typedef struct {
    int32_t numIterations;   /* 5000 */
    int32_t adptRho;         /* adptΡ, 0/1, default 0 */
    int32_t numItrConv;      /* 25 */
    int32_t numItrPolish;    /* 10; used only when polish != 0 (the Julia loop reserves the kwarg without using it) */
    int32_t numItrMinres;    /* 500; polishing only */
    int32_t linsys;          /* qps_linsys_kind, additive, default AUTO */
    int32_t trsvBlock;       /* additive: size of the inverted diagonal blocks of the blocked triangular sweep; a
                                power-of-two multiple of 64, 0 = library default: one inverted block over the whole factor
                                while the fused forward+backward sweep covers n (n padded <= 16384 fp64 / 32768 fp32), else 4096.
                                1024 or 512 (fp64) / 2048 or 1024 (fp32) with n above it: blocked substitution, ONE launch per
                                sweep (qps_info.sweepVariant = 5); other sizes below n: one launch per block phase (1)         */
    int32_t reuseFactor;     /* additive: 1 = keep the factorisation of a previous qps_solve/linsys_init when
                                (rho, sigma) are unchanged; 0 = factorise on every call like the reference (:36)      */
    double epsAbs;           /* ϵAbs 1e-6 */
    double epsRel;           /* ϵRel 1e-6 */
    double rho;              /* ρ 1 */
    double sigma;            /* σ 1e-6 */
    double alpha;            /* α 1.6 */
    double delta;            /* δ 1e-6; polishing only */
    double fctrRho;          /* fctrΡ 5 */
    double epsMinres;        /* ϵMinres 1e-6; polishing only */
    double epsPcg;           /* CG plugins' ϵPcg 1e-6 (LinearSystemSolvers.jl:125) */
    int32_t numItrPcg;       /* CG plugins' numItrPcg 1000 */
    int32_t loopVariant;     /* additive: 0 = automatic: small problems run the whole loop in one single-workgroup launch,
                                larger ones use the fused single pass over A per iteration when the shape allows (default);
                                1 = unfused kernels (A read twice; the literal order of LinearSystemSolvers.jl:134-139);
                                2 = multi-launch fused loop even for small problems */
    int32_t polish;          /* additive: 0 = no polishing (what SolveQuadraticProgram.jl does, default); 1 = run the polishing step
                                of SolveQuadraticProgram.m:289-325 after the loop with numItrPolish, delta, epsMinres, numItrMinres */
    int32_t reserved0;
} qps_params;

/* Additive out-of-band report (the reference returns only the flag, SolveQuadraticProgram.jl:73). */
typedef struct {
    int32_t convFlag;        /* qps_conv_flag */
    int32_t iterations;      /* loop bodies executed */
    int32_t numRefactor;     /* changedΡ events (each one re-factorises) */
    int32_t cgIterations;    /* total inner CG iterations (QPS_LINSYS_CG) */
    double rhoFinal;         /* ρ in force at exit */
    double rhoProposed;      /* ρρ at exit */
    double resPrim;          /* ||A x - z||_inf at the last check (:85) */
    double resDual;          /* ||P x + q + A'y||_inf at the last check (:86) */
    double tSetup;           /* seconds: LinSysSolInit (assembly + factorisation), device time incl. sync */
    double tLoop;            /* seconds: the iteration loop, device time incl. sync */
    double tRefactor;        /* seconds spent in changedΡ re-factorisations (part of tLoop) */
    int32_t polishFlag;      /* minresFlag of SolveQuadraticProgram.m:311-325: -1 polishing did not run, 0 converged (x replaced),
                                1 the last MINRES call did not converge (x kept) */
    int32_t polishIterations;/* total MINRES iterations of the polishing step */
    double tPolish;          /* seconds spent polishing (not part of tLoop) */
    int32_t trsvBlock;       /* dense handles: size of the inverted diagonal blocks the triangular sweeps ran with (0: not a dense Cholesky run) */
    int32_t sweepVariant;    /* dense handles: 1 = blocked substitution over the sweep matrix (2 n / trsvBlock - 1 dependent phases per sweep),
                                2 = explicit inverse, both sweeps fused into one pass over the triangle (trsvBlock >= n),
                                3 = explicit inverse, two triangular GEMVs, 4 = single-launch small-problem loop,
                                5 = blocked substitution, ONE launch per sweep (n / trsvBlock dependent phases handed from workgroup to
                                    workgroup inside the launch; trsvBlock = 1024 or 512 fp64, 2048 or 1024 fp32); 0 otherwise */
    int32_t sweepGaveUp;     /* times a variant-5 launch of this solve gave up waiting for its workgroups (only another PROCESS running the
                                same kernel on the card can cause that) and the solve was repeated on variant 1; its time is in tLoop */
    int32_t cgExplicit;      /* CSC handles, CG plugins: 1 = the solve ran cg! on the explicit reduced matrix (ItrSolCg, one product per CG iteration),
                                0 = on the matrix-free operator (LinOpCg / LinMapsCg) or no CG at all */
} qps_info;

/* Fill *p with the reference defaults (SolveQuadraticProgram.jl:15-17). */
int32_t qps_default_params(qps_params *p);

/* Number of HIP devices visible (0 when there is none; never an error). */
int32_t qps_device_count(void);

/* Dense problem, column-major (Julia Matrix{Float64}): P is n x n (leading dimension ldp), A is m x n (lda).
 * Replaces the data half of the call SolveQuadraticProgram.jl:14 for dense inputs. */
int32_t qps_create_dense(int64_t n, int64_t m, const double *P, int64_t ldp, const double *A, int64_t lda,
                         const double *q, const double *l, const double *u, int32_t dtype, int32_t device,
                         qps_handle *out);

/* Sparse problem, CSC (Julia SparseMatrixCSC{Float64,Int64}: colptr, rowval, nzval), index_base 1 for Julia, 0 for C.
 * P must hold the full symmetric matrix (as GenerateQuadraticProgram.jl produces).  dense_path != 0 densifies the
 * problem on the device and uses the Cholesky path; 0 keeps CSR storage and uses the CG path. */
int32_t qps_create_csc(int64_t n, int64_t m,
                       const int64_t *P_colptr, const int64_t *P_rowval, const double *P_nzval,
                       const int64_t *A_colptr, const int64_t *A_rowval, const double *A_nzval,
                       const double *q, const double *l, const double *u, int32_t index_base, int32_t dense_path,
                       int32_t dtype, int32_t device, qps_handle *out);

/* SolveQuadraticProgram! (SolveQuadraticProgram.jl:14-76): x_inout is vX (warm start in, solution out, length n);
 * z and y restart at 0 (:39-40).  info may be NULL.  Blocks until the result is in x_inout. */
int32_t qps_solve(qps_handle h, double *x_inout, const qps_params *params, qps_info *info);

/* The polishing step alone (SolveQuadraticProgram.m:289-325; MATLAB only -- the Julia loop reserves its kwargs,
 * SolveQuadraticProgram.jl:16-17): active sets from the sign of the multiplier y (length m), reduced KKT system, iterative
 * refinement with MINRES (numItrPolish, delta, epsMinres, numItrMinres of *params).  x_inout (length n) is replaced by the
 * polished primal only when report->flag == 0.  Dense and CSR handles (batch handles: through qps_params.polish). */
typedef struct {
    int32_t flag;              /* minresFlag: -1 did not run (numItrPolish <= 0), 0 converged, 1 not converged (x kept) */
    int32_t refinements;       /* bodies of the refinement loop executed (<= numItrPolish) */
    int32_t minresIterations;  /* total MINRES iterations */
    int32_t numActiveLower;    /* numL = sum(vY < 0) */
    int32_t numActiveUpper;    /* numU = sum(vY > 0) */
    int32_t reserved0;
    double relres;             /* ||r|| / ||b|| of the last MINRES call */
    double seconds;
} qps_polish_report;
int32_t qps_polish(qps_handle h, double *x_inout, const double *y, const qps_params *params, qps_polish_report *report);

/* Final z and y of the last qps_solve (length m each; batch handles: [count][m] each, of the last qps_solve_batch; either pointer
 * may be NULL). Additive. */
int32_t qps_get_dual(qps_handle h, double *z_out, double *y_out);

/* The reference plugin pair, literally (host vectors in, device solve, host vectors out):
 *   qps_linsys_init  == LinSysSolInit(vX, mP, vQ, mA, rho, 1/rho, sigma, n, m)       LinearSystemSolvers.jl:110-122
 *   qps_linsys_solve == LinSysSol!(tuSolver, vXX, vZZ, vX, ..., vZ, vY, rho, 1/rho, sigma, n, m, changedRho)  :125-142
 * Post-condition as in the reference: xx_out == x-tilde (n), zz_out == z-tilde (m). */
int32_t qps_linsys_init(qps_handle h, double rho, double sigma, int32_t linsys, int32_t trsvBlock);
int32_t qps_linsys_solve(qps_handle h, const double *x, const double *z, const double *y, double rho, double sigma,
                         int32_t changed_rho, double *xx_out, double *zz_out);
/* The keyword arguments of the CG plugins' Sol!  --  ItrSolCg! / LinOpCg! / LinMapsCg!(...; ϵPcg = 1e-6, numItrPcg = 1000)
 * (LinearSystemSolvers.jl:125, :164, :207): inner tolerance (abstol of IterativeSolvers.cg!) and iteration cap used by the following
 * qps_linsys_solve calls of a CG handle.  Accepted and without effect on the direct plugins (they have no inner iteration). */
int32_t qps_linsys_set_cg(qps_handle h, double epsPcg, int32_t numItrPcg);

/* ONE application of the matrices behind the reduced operator of the CG plugins -- LinOpCgInit / LinMapsCgInit build it from three products,
 *     mul!(vZZ, mA, vW); mul!(vU, mA', vZZ); mul!(vU, mP, vW, 1.0, rho); vU .+= sigma .* vW          LinearSystemSolvers.jl:152-157, :195-200
 * and CheckConvergence applies the same three matrices to x and y (SolveQuadraticProgram.jl:85-89) -- through the device kernels qps_solve itself
 * uses for them on this handle (CSC handles: the column-blocked SpMV when the handle built one, else the CSR-stream kernel; dense handles: the
 * row / column GEMVs).  Host vector in, host vector out, in the handle's arithmetic type on the device:
 *     QPS_OP_P        out[n]     = mP  * in[n]
 *     QPS_OP_A        out[m]     = mA  * in[n]
 *     QPS_OP_AT       out[n]     = mA' * in[m]
 *     QPS_OP_PA       out[n + m] = [mP; mA] * in[n]   (the stacked product of a CG iteration: one pass over `in`)
 *     QPS_OP_REDUCED  out[n]     = mP in + rho mA'(mA in) + sigma in                                    (:152-157; rho, sigma used by this one only)
 * Additive (the reference has no such call): it lets a binding or a test compare the device copies of its matrices with its own product, one
 * operator at a time.  The solver state (x, z, y of the last solve) is not touched. */
typedef enum { QPS_OP_P = 0, QPS_OP_A = 1, QPS_OP_AT = 2, QPS_OP_PA = 3, QPS_OP_REDUCED = 4 } qps_operator_kind;
int32_t qps_operator_apply(qps_handle h, int32_t op, const double *in, double *out, double rho, double sigma);

/* Batch of `count` independent dense QPs of identical shape (BASELINE config 4).  Problem b uses
 * P + b*ldp*n ... i.e. arrays are stacked along a leading batch axis: P[count][n*n], A[count][m*n] (column-major
 * each), q[count][n], l/u[count][m].  x_inout is [count][n]; infos (may be NULL) is [count]. */
int32_t qps_create_dense_batch(int64_t count, int64_t n, int64_t m, const double *P, const double *A, const double *q,
                               const double *l, const double *u, int32_t dtype, int32_t device, qps_handle *out);
int32_t qps_solve_batch(qps_handle h, double *x_inout, const qps_params *params, qps_info *infos);

/* The per-problem loop of RunBenchmarks.jl:88-104 sharded over the devices of ONE process ("one host thread + one stream per device", "work-stealing at
 * chunk boundaries").  `count` independent dense QPs of one shape, arrays stacked as for qps_create_dense_batch; `devices[num_workers]` lists the device of every worker
 * -- one host thread each; a device may be listed more than once (two workers sharing a card).  Every worker repeatedly takes the next range of QPs from a shared
 * counter -- `chunk` of them at first, fewer towards the end of the batch: clamp(ceil(remaining / (2 W)), max(1, chunk / 4), chunk) -- builds a batch handle for them on
 * its device, solves them (qps_solve_batch) and drops the handle: a run to a tolerance, in which every QP stops at its own iteration, balances itself.  chunk <= 0:
 * static contiguous slabs, worker w takes QPs [w * ceil(count / W), ...) -- what a fixed-K run wants (equal work per QP, no hand-out).  The size of a range depends
 * only on where it starts, so the ranges are cut the same way whoever solves them: the results do not depend on timing or on which worker took a range, and they
 * equal qps_solve_batch on batches of those same ranges bit for bit.  No data-path collective, no exchange between devices.
 * x_inout [count][n] (warm starts in, solutions out), infos [count] (may be NULL), worker_of [count] (may be NULL: which worker solved QP b), worker_seconds
 * [num_workers] (may be NULL: busy time per worker, handle creation included).  On an error the other workers stop at their next chunk boundary and the first error is
 * returned (qps_last_error(NULL) on the calling thread). */
int32_t qps_solve_batch_multi(int64_t count, int64_t n, int64_t m, const double *P, const double *A, const double *q, const double *l, const double *u,
                              int32_t dtype, const int32_t *devices, int32_t num_workers, int32_t chunk, double *x_inout, const qps_params *p, qps_info *infos,
                              int32_t *worker_of, double *worker_seconds);

/* Device-time of the loop kernels of the last qps_solve, measured with HIP events attached to the kernel dispatches
 * themselves (hipExtLaunchKernelGGL start / stop events on the solver's stream: the dispatch's own begin / end timestamps)
 * (used by bench.py's roofline block).  names is a caller buffer of `cap` entries; returns the number filled. */
typedef struct { char name[48]; double seconds; int64_t launches; double algo_bytes; } qps_kernel_time;
int32_t qps_kernel_times(qps_handle h, qps_kernel_time *out, int32_t cap, int32_t *count);
/* level 0 = off; 1 = sampling: once per 50 iterations ONE launch of each loop kernel (fused pass, fused sweeps, the two slab
 * reductions; each on a different iteration) is bracketed with HIP events -- cheap enough for a timed region (an event pair
 * per launch would cost ~7 % of the loop);
 * 2 = bracket every loop kernel (diagnostic).  Setting the level also resets the accumulated times. */
int32_t qps_set_profiling(qps_handle h, int32_t on);

/* ---- The reference's second solver form (ProxQP.jl):  min 1/2 x'Px + q'x  s.t.  A x = b,  C x <= d ---------------------------
 * qps_proxqp_create_dense  == the data half of  ProxQP(mP, vQ, mA, vB, mC, vD [, vX, vY, vZ, vS])        ProxQP.jl:36,73
 * qps_proxqp_init_kkt      == the initialisation of the 6-argument constructor (x, y from the equality-constrained KKT
 *                             system, s = max(d - C x, 0), z = 0)                                        ProxQP.jl:80-89
 * qps_proxqp_set_state     == the explicit vX, vY, vZ, vS of the 10-argument constructor                 ProxQP.jl:36
 * qps_proxqp_solve         == SolveQuadraticProgram!(sQpProb; numIterations, ϵAbs, ϵRel, numItrConv, ρ, σ, adptΡ, τ) -> dReport
 *                             (always runs numIterations: the reference's `break` is commented out)      ProxQP.jl:118-173
 * qps_proxqp_get_state     == reading sQpProb.vX / vY / vZ / vS afterwards
 * All matrices column-major (Julia Matrix{Float64}); numEq or numInEq may be 0.  Same handle type, qps_destroy frees it. */
typedef struct {
    int32_t numIterations;   /* 2000 */
    int32_t numItrConv;      /* 50 */
    int32_t adptRho;         /* adptΡ, default 1 */
    int32_t loopVariant;     /* additive: 0 = auto (one fused pass over [A; C] per iteration), 1 = unfused (two passes) */
    double epsAbs;           /* ϵAbs 1e-7 */
    double epsRel;           /* ϵRel 1e-6 */
    double rho;              /* ρ 1e2 */
    double sigma;            /* σ 1e-2 */
    double tau;              /* τ 10 */
} qps_proxqp_params;
typedef struct {             /* dReport (ProxQP.jl:127): "Converged", "Iterations", "ρ", "σ", "PrimalResidual", "DualResidual" */
    int32_t converged; int32_t iterations; double rho, sigma, resPrim, resDual;
} qps_proxqp_report;
int32_t qps_proxqp_default_params(qps_proxqp_params *p);
int32_t qps_proxqp_create_dense(int64_t n, int64_t numEq, int64_t numInEq, const double *P, int64_t ldp, const double *q,
                                const double *A, int64_t lda, const double *b, const double *C, int64_t ldc, const double *d,
                                int32_t dtype, int32_t device, qps_handle *out);
/* SparseProxQP (ProxQP.jl:71, :95-115): the same with the colptr / rowval / nzval fields of three SparseMatrixCSC inputs (index_base 1 for
 * Julia).  The matrices stay sparse: the linear system of UpdateX! (:221-225) is solved in its KKT form [P + sigma I, G'; G, -I/rho], G = [A; C],
 * by the sparse L D L' plugin -- ordering and symbolic factor once per handle, a rho update re-factorises numerically on the frozen pattern
 * (the role of AlignSparsePattern / GetNzvalDiagIdxs / UpdateM! / the pattern-reusing cholesky!, :184-190, :201-206, :335-372).
 * qps_proxqp_init_kkt on such a handle: sparse factor of [P A'; A -1e-8 I] + iterative refinement against [P A'; A 0] (:95-115).
 * QPS_ERR_UNSUPPORTED at the first solve / init when the factor does not fit the level-scheduled plugin (then: the dense constructor). */
int32_t qps_proxqp_create_csc(int64_t n, int64_t numEq, int64_t numInEq, const int64_t *P_colptr, const int64_t *P_rowval, const double *P_nzval,
                              const double *q, const int64_t *A_colptr, const int64_t *A_rowval, const double *A_nzval, const double *b,
                              const int64_t *C_colptr, const int64_t *C_rowval, const double *C_nzval, const double *d, int32_t index_base,
                              int32_t dtype, int32_t device, qps_handle *out);
int32_t qps_proxqp_init_kkt(qps_handle h);
int32_t qps_proxqp_set_state(qps_handle h, const double *x, const double *y, const double *z, const double *s);
int32_t qps_proxqp_get_state(qps_handle h, double *x, double *y, double *z, double *s);
int32_t qps_proxqp_solve(qps_handle h, const qps_proxqp_params *params, qps_proxqp_report *report);

/* The modeAuto rule of SolveQuadraticProgramRef! (SolveQuadraticProgram.jl:129-130, :143-151; SolveQuadraticProgram.m:190-199):
 * numRowsL = n + m, nnzDensity = (nnz(P) + nnz(A)) / numRowsL^2; direct when numRowsL <= 5000 and nnzDensity <= 0.4, else
 * iterative.  Returns the qps_linsys_kind to use: QPS_LINSYS_CG (iterative), or for "direct" QPS_LINSYS_KKT_LDL when the caller
 * holds sparse matrices (sparse_input != 0; the reference's ldlt / decomposition(...,'ldl') of the sparse KKT matrix) and
 * QPS_LINSYS_CHOLESKY for dense arrays.  Pure function: no handle, no device. */
int32_t qps_linsys_auto(int64_t n, int64_t m, int64_t nnzP, int64_t nnzA, int32_t sparse_input);

/* Symbolic analysis of the KKT matrix for QPS_LINSYS_KKT_LDL on its own (host only, no device needed): what LaLdlInit / QDLdlInit /
 * FacLdlInit do before the numeric factorisation (LinearSystemSolvers.jl:18, :49, :81).  perm_out (n + m entries, may be NULL)
 * receives the elimination order over [x; nu] in the caller's index base; the report sizes the factor. */
typedef struct {
    int64_t numRows;            /* n + m */
    int64_t numSparseColumns;   /* columns kept as scalar sparse columns (wide elimination-tree levels) */
    int64_t tailSize;           /* columns near the root handled as one dense block */
    int64_t numSparseLevels;    /* launches per triangular sweep over the sparse part */
    int64_t treeHeight;         /* height of the elimination tree */
    int64_t nnzK;               /* strictly lower triangle of K */
    int64_t nnzL;               /* strictly lower triangle of L under the minimum-degree ordering */
    int64_t nnzStored;          /* entries actually stored (the tail counted as dense) */
} qps_ldl_report;
int32_t qps_ldl_analyze(int64_t n, int64_t m, const int64_t *P_colptr, const int64_t *P_rowval, const int64_t *A_colptr,
                        const int64_t *A_rowval, int32_t index_base, int64_t *perm_out, qps_ldl_report *report);

int32_t qps_destroy(qps_handle h);
/* Human-readable description of the last failure on this handle (or of the last failed create when h == NULL). */
const char *qps_last_error(qps_handle h);
const char *qps_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QPS_H */
