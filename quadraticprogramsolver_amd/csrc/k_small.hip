// k_small.hip -- small-problem path: the WHOLE ADMM loop of SolveQuadraticProgram.jl:45-71 inside one kernel launch.
//
// For the reference's own test sizes (n = 10..100, RunTests.jl:30-38) an iteration moves a few hundred KB, so a loop built
// from separate launches is bound by launch latency (~17 us/iteration measured = 4-5 kernel boundaries), slower than one CPU
// thread.  Here ONE workgroup of 1024 threads keeps every vector in LDS, reads the (L2-resident) matrices with unit stride
// and runs iterations back to back: linear solve, x/z/y update, and every numItrConv iterations CheckConvergence
// (:79-112) with the rho proposal.  The kernel returns when a termination test fires, when the iteration budget is spent,
// or when the proposed rho leaves the fctrRho band (:47) -- the host then re-factorises and relaunches.
//
// All four matrix-vector products are written as column accumulations  out[j] = sum_i Mat[i][j] v[i]  over row-major
// storage (lanes along the contiguous index j): A'w uses A, z~ = A x~ uses a transposed copy At, the forward sweep uses
// the upper triangle of S (= W'), the backward sweep the lower triangle (= W).  Requires one inverted block (nb >= n).
#include <cstdlib>

#include "qps_kernels.h"
#include "wave_reduce.h"

namespace qps {

namespace {


__device__ __forceinline__ unsigned long long absbits_s(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__device__ __forceinline__ double jmax_s(double a, double b) { return (isnan(a) || isnan(b)) ? (double)NAN : (a > b ? a : b); }

// out[j] = pre(j) + sum_{i in rows, mask(i,j)} Mat[i*ld + j] * v[i],  j < ncols.  tri: 0 none, 1 keep i <= j, 2 keep i >= j.
// Lanes run along j (J = power of two >= ncols, <= ST), the ST / J row groups split i; partial sums meet in LDS `scr`.
template <typename T, int ST>
__device__ __forceinline__ void gemv_cols_lds(const T* __restrict__ Mat, int ld, int nrows, int ncols, const T* v, T* out, T* scr, int tri) {
    const int tid = threadIdx.x;
    for (int j0 = 0; j0 < ncols; j0 += ST) {
        const int nc = min(ncols - j0, ST);
        int J = 64; while (J < nc) J <<= 1;
        const int G = ST / J, jj = tid % J, g = tid / J, j = j0 + jj;
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
        if (jj < nc) {
            int ibeg = g, iend = nrows;
            if (tri == 1) iend = min(nrows, j + 1);
            if (tri == 2) { const int first = j; ibeg = first + ((g - first % G) % G + G) % G; }
            const T* mp = Mat + j;
            int i = ibeg;
            for (; i + 3 * G < iend; i += 4 * G) {
                a0 += mp[(int64_t)i * ld] * v[i];
                a1 += mp[(int64_t)(i + G) * ld] * v[i + G];
                a2 += mp[(int64_t)(i + 2 * G) * ld] * v[i + 2 * G];
                a3 += mp[(int64_t)(i + 3 * G) * ld] * v[i + 3 * G];
            }
            for (; i < iend; i += G) a0 += mp[(int64_t)i * ld] * v[i];
        }
        scr[g * J + jj] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (g == 0 && jj < nc) {
            T s = T(0);
            for (int k = 0; k < G; ++k) s += scr[k * J + jj];   // fixed order
            out[j] = s;
        }
        __syncthreads();
    }
}

// out[r] = sum_c Al[r * lda + c] * v[c] for an LDS-resident matrix with an ODD row stride in 8-byte words (lda = NP + 1):
// PARTS threads share a row (contiguous column ranges), lanes of a wave sit on different rows => conflict-free reads.
template <typename T, int ST>
__device__ __forceinline__ void gemv_rows_ldsmat(const T* Al, int lda, int nrows, int ncols, const T* v, T* out, T* scr) {
    const int tid = threadIdx.x;
    int R = 64; while (R < nrows && R < ST) R <<= 1;          // lanes along rows
    const int PARTS = ST / R, r = tid % R, part = tid / R;
    const int span = (ncols + PARTS - 1) / PARTS, c0 = part * span, c1 = min(ncols, c0 + span);
    for (int rb = 0; rb < nrows; rb += R) {
        T s = T(0);
        const int row = rb + r;
        if (row < nrows) { const T* ap = Al + (size_t)row * lda; for (int c = c0; c < c1; ++c) s += ap[c] * v[c]; }
        scr[part * R + r] = s;
        __syncthreads();
        if (part == 0 && row < nrows) { T tsum = T(0); for (int p = 0; p < PARTS; ++p) tsum += scr[p * R + r]; out[row] = tsum; }
        __syncthreads();
    }
}

struct SmallArgs {
    int n, m, NP, MP;
    int it_begin, it_end, numItrConv, adptRho;     // iterations it_begin+1 .. it_end are run (1-based, as `ii` in the reference)
    double rho, rhorho, sigma, alpha, epsAbs, epsRel, epsAdmm, fctrRho;
};
// status[0] last iteration executed, [1] convFlag, [2] need_rho (1: proposed rho left the band), doubles: res[0..7] as the check kernels
struct SmallOut { int last_it, convFlag, need_rho, pad; double res[8]; };

template <typename T, int ST, bool LM>
__global__ __launch_bounds__(ST) void k_admm_small(SmallArgs a, const T* __restrict__ A, const T* __restrict__ At, const T* __restrict__ P,
                                                   const T* __restrict__ S, const T* __restrict__ q, const T* __restrict__ l,
                                                   const T* __restrict__ u, T* __restrict__ gx, T* __restrict__ gxp, T* __restrict__ gz,
                                                   T* __restrict__ gy, SmallOut* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = a.NP, MP = a.MP, tid = threadIdx.x;
    T* x = reinterpret_cast<T*>(smem);          // NP
    T* xp = x + NP; T* xx = xp + NP; T* t = xx + NP; T* yv = t + NP; T* qv = yv + NP;     // 6 NP
    T* z = qv + NP; T* zp = z + MP; T* y = zp + MP; T* zz = y + MP; T* w = zz + MP; T* lv = w + MP; T* uv = lv + MP;   // 7 MP
    T* scr = uv + MP;                           // 1024
    // LM: A (row stride NP + 1, conflict-free row walks) and S are copied into LDS once per launch
    T* Al = scr + 1024; const int lda = NP + 1;
    T* Sl = Al + (LM ? (size_t)MP * lda : 0);
    if (LM) {
        for (int i = tid; i < MP * NP; i += ST) { const int r = i / NP, c = i - r * NP; Al[(size_t)r * lda + c] = A[i]; }
        for (int i = tid; i < NP * NP; i += ST) Sl[i] = S[i];
    }
    const T* Am = LM ? Al : A; const int ldA = LM ? lda : NP;
    const T* Sm = LM ? Sl : S;
    __shared__ unsigned long long nrm[9];
    __shared__ int sh_flag, sh_need;
    __shared__ double sh_rhorho;

    for (int i = tid; i < NP; i += ST) { x[i] = gx[i]; xp[i] = gxp[i]; qv[i] = q[i]; xx[i] = T(0); }
    for (int i = tid; i < MP; i += ST) { z[i] = gz[i]; y[i] = gy[i]; lv[i] = l[i]; uv[i] = u[i]; zp[i] = T(0); }
    if (tid == 0) { sh_flag = 1; sh_need = 0; sh_rhorho = a.rhorho; }
    __syncthreads();
    const T rho = (T)a.rho, rho1 = T(1) / rho, sigma = (T)a.sigma, alpha = (T)a.alpha, alpha1 = T(1) - alpha;
    int it = a.it_begin;
    double res[8] = {NAN, NAN, NAN, NAN, a.rhorho, 1.0, NAN, NAN};
    while (it < a.it_end) {
        ++it;
        // LinearSystemSolvers.jl:134-139 (reduced form, Cholesky instead of cg!)
        for (int i = tid; i < MP; i += ST) w[i] = rho * z[i] - y[i];                        // :134
        __syncthreads();
        gemv_cols_lds<T, ST>(Am, ldA, a.m, NP, w, t, scr, 0);                                     // :135  A' w
        for (int i = tid; i < NP; i += ST) t[i] = sigma * x[i] - qv[i] + t[i];                // :136
        __syncthreads();
        gemv_cols_lds<T, ST>(Sm, NP, NP, NP, t, yv, scr, 1);                                      // forward sweep:  y = W t   (rows of W' = upper part)
        gemv_cols_lds<T, ST>(Sm, NP, NP, NP, yv, xx, scr, 2);                                     // backward sweep: x~ = W' y (rows of W = lower part)
        if (LM) gemv_rows_ldsmat<T, ST>(Al, lda, MP, NP, xx, zz, scr);                            // :139  z~ = A x~
        else gemv_cols_lds<T, ST>(At, MP, a.n, MP, xx, zz, scr, 0);
        // SolveQuadraticProgram.jl:56-61
        for (int i = tid; i < NP; i += ST) { const T xo = x[i]; xp[i] = xo; x[i] = alpha * xx[i] + alpha1 * xo; }
        for (int i = tid; i < MP; i += ST) {
            const T zo = z[i], yo = y[i], zt = zz[i];
            zp[i] = zo;
            const T tt = alpha * zt + alpha1 * zo + rho1 * yo;
            const T zn = tt > uv[i] ? uv[i] : (tt < lv[i] ? lv[i] : tt);
            z[i] = zn;
            y[i] = yo + rho * (alpha * zt + alpha1 * zo - zn);
        }
        __syncthreads();
        if (it % a.numItrConv == 0) {                                                         // :63  CheckConvergence :79-112
            T* Ax = zz; T* Px = t; T* Aty = yv;                                               // scratch vectors are free here
            if (LM) gemv_rows_ldsmat<T, ST>(Al, lda, MP, NP, x, Ax, scr);
            else gemv_cols_lds<T, ST>(At, MP, a.n, MP, x, Ax, scr, 0);
            gemv_cols_lds<T, ST>(P, NP, a.n, NP, x, Px, scr, 0);
            gemv_cols_lds<T, ST>(Am, ldA, a.m, NP, y, Aty, scr, 0);
            if (tid < 9) nrm[tid] = 0ull;
            __syncthreads();
            unsigned long long v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = tid; i < max(a.n, a.m); i += ST) {
                if (i < a.m) {
                    v[0] = max(v[0], absbits_s((double)(Ax[i] - z[i]))); v[2] = max(v[2], absbits_s((double)Ax[i]));
                    v[3] = max(v[3], absbits_s((double)z[i])); v[8] = max(v[8], absbits_s((double)(z[i] - zp[i])));
                }
                if (i < a.n) {
                    v[1] = max(v[1], absbits_s((double)(Px[i] + qv[i] + Aty[i]))); v[4] = max(v[4], absbits_s((double)Px[i]));
                    v[5] = max(v[5], absbits_s((double)Aty[i])); v[6] = max(v[6], absbits_s((double)qv[i]));
                    v[7] = max(v[7], absbits_s((double)(x[i] - xp[i])));
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long tv = __shfl_xor(v[k], o, 64); v[k] = tv > v[k] ? tv : v[k]; }
                if ((tid & 63) == 0 && v[k]) atomicMax(&nrm[k], v[k]);
            }
            __syncthreads();
            if (tid == 0) {
                double nv[9];
                for (int k = 0; k < 9; ++k) nv[k] = __longlong_as_double((long long)nrm[k]);
                const double normResPrim = nv[0], normResDual = nv[1];                        // :85-86
                const double maxNormPrim = jmax_s(nv[2], nv[3]);                              // :88
                const double maxNormDual = jmax_s(jmax_s(nv[4], nv[5]), nv[6]);               // :89
                double rr = sh_rhorho;
                if (a.adptRho) {                                                              // :92-96
                    const double tv = a.rho * sqrt((normResPrim * maxNormDual) / (normResDual * maxNormPrim));
                    rr = tv > 1e6 ? 1e6 : (tv < 1e-3 ? 1e-3 : tv);
                }
                int flag = 1;
                if ((normResPrim < a.epsAbs + a.epsRel * maxNormPrim) && (normResDual < a.epsAbs + a.epsRel * maxNormDual)) flag = 3;   // :102-104
                if ((nv[7] <= a.epsAdmm) && (nv[8] <= a.epsAdmm)) flag = 2;                   // :105-107 (not else)
                sh_rhorho = rr; sh_flag = flag;
                sh_need = (flag == 1 && a.adptRho && ((rr * a.fctrRho < a.rho) || (rr > a.fctrRho * a.rho))) ? 1 : 0;   // :47
                res[0] = normResPrim; res[1] = normResDual; res[2] = maxNormPrim; res[3] = maxNormDual; res[4] = rr; res[5] = flag;
                res[6] = nv[7]; res[7] = nv[8];
            }
            __syncthreads();
            if (sh_flag != 1 || sh_need) break;
        }
    }
    for (int i = tid; i < NP; i += ST) { gx[i] = x[i]; gxp[i] = xp[i]; }
    for (int i = tid; i < MP; i += ST) { gz[i] = z[i]; gy[i] = y[i]; }
    if (tid == 0) {
        out->last_it = it; out->convFlag = sh_flag; out->need_rho = sh_need;
        for (int k = 0; k < 8; ++k) out->res[k] = res[k];
        out->res[4] = sh_rhorho;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// n <= 128, small m: the matrices live in REGISTERS.  512 threads = 8 waves; lane = column (n-vectors) or row inside a block
// of 64 (m-vectors), wave g owns the matrix rows g, g + 8, g + 16, ...  Every product of the iteration is a column
// accumulation over register-resident entries whose vector operand comes by v_readlane from the lane that owns it
// (wave-uniform index), followed by ONE cross-wave sum through LDS -- four barriers per iteration, no LDS read per matrix
// entry (the LDS-resident version above reads every entry of A twice and of S once per iteration through the LDS pipe).
//   a[r]     = A[g + 8 r][lane]           A' w  :  sum_r a[r] * w[g + 8 r]
//   at[b][c] = A[64 b + lane][g + 8 c]    A x~  :  sum_c at[b][c] * x~[g + 8 c]
//   s[r]     = S[g + 8 r][lane]           y = W t:  rows j = g + 8 r <= lane (upper part = W');  x~ = W' y: rows >= lane (lower part = W)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int NB, int MB>      // NB = n padded / 64 (1 or 2), MB = m padded / 64
__device__ __forceinline__ void small_reg_body(const SmallArgs& a, const T* __restrict__ A, const T* __restrict__ P, const T* __restrict__ S,
                                               const T* __restrict__ q, const T* __restrict__ l, const T* __restrict__ u,
                                               T* __restrict__ gx, T* __restrict__ gxp, T* __restrict__ gz, T* __restrict__ gy,
                                               SmallOut* __restrict__ out) {
    constexpr int NPc = 64 * NB, RA = 8 * MB, RS = 8 * NB;          // rows of A / S per wave
    __shared__ T scrN[2][8][NPc];
    __shared__ T scrM[8][64 * MB];
    __shared__ unsigned long long nrm[9];
    __shared__ int sh_flag, sh_need;
    __shared__ double sh_rhorho;
    __shared__ double sres[8];                 // written by thread 0 only: kept out of everybody's registers
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    T ar[RA][NB], at[MB][RS], sr[RS][NB];
#pragma unroll
    for (int r = 0; r < RA; ++r)
#pragma unroll
        for (int c = 0; c < NB; ++c) ar[r][c] = A[(int64_t)(g + 8 * r) * NPc + 64 * c + lane];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int c = 0; c < RS; ++c) at[b][c] = A[(int64_t)(64 * b + lane) * NPc + g + 8 * c];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int c = 0; c < NB; ++c) sr[r][c] = S[(int64_t)(g + 8 * r) * NPc + 64 * c + lane];
    T x[NB], xp[NB], xt[NB], qv[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) { x[c] = gx[64 * c + lane]; xp[c] = gxp[64 * c + lane]; xt[c] = T(0); qv[c] = q[64 * c + lane]; }
    T z[MB], y[MB], zp[MB], lv[MB], uv[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) { z[b] = gz[64 * b + lane]; y[b] = gy[64 * b + lane]; zp[b] = T(0); lv[b] = l[64 * b + lane]; uv[b] = u[64 * b + lane]; }
    if (tid == 0) { sh_flag = 1; sh_need = 0; sh_rhorho = a.rhorho; }
    if (tid < 8) sres[tid] = (tid == 4) ? a.rhorho : (tid == 5 ? 1.0 : (double)NAN);
    __syncthreads();
    const T rho = (T)a.rho, rho1 = T(1) / rho, sigma = (T)a.sigma, alpha = (T)a.alpha, alpha1 = T(1) - alpha;
    // entry i of a vector held as lane-blocks of 64 (i is wave-uniform)
    auto nget = [&](const T (&v)[NB], int i) -> T { return NB == 1 ? lane_get(v[0], i) : ((i < 64) ? lane_get(v[0], i) : lane_get(v[NB - 1], i - 64)); };
    // out (n-vector) = sum over the 8 waves of sum_r ar[r][.] * v[g + 8 r]   (v: an m-vector held as MB lane-blocks)
    auto atv = [&](const T (&v)[MB], T (*scr)[NPc], T (&outv)[NB]) {
        T p[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) p[c] = T(0);
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const T vi = lane_get(v[r >> 3], g + 8 * (r & 7));
#pragma unroll
            for (int c = 0; c < NB; ++c) p[c] += ar[r][c] * vi;
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) scr[g][64 * c + lane] = p[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) t += scr[k][64 * c + lane];
            outv[c] = t;
        }
    };
    // out (m-vector) = sum over the waves of sum_c at[b][c] * v[g + 8 c]   (v: an n-vector)
    auto av = [&](const T (&v)[NB], T (&outv)[MB]) {
        T pm[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) pm[b] = T(0);
#pragma unroll
        for (int c = 0; c < RS; ++c) {
            const T vj = nget(v, g + 8 * c);
#pragma unroll
            for (int b = 0; b < MB; ++b) pm[b] += at[b][c] * vj;
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) scrM[g][64 * b + lane] = pm[b];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) t += scrM[k][64 * b + lane];
            outv[b] = t;
        }
    };
    // out (n-vector) = cross-wave sum of per-thread partials p[.]
    auto nsum = [&](const T (&p)[NB], T (*scr)[NPc], T (&outv)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; ++c) scr[g][64 * c + lane] = p[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) t += scr[k][64 * c + lane];
            outv[c] = t;
        }
    };
    int it = a.it_begin;
    while (it < a.it_end) {
        ++it;
        T w[MB], t[NB], y1[NB], p[NB], zt[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) w[b] = rho * z[b] - y[b];                                    // LinearSystemSolvers.jl:134
        atv(w, scrN[0], t);                                                                       // :135  A' w
#pragma unroll
        for (int c = 0; c < NB; ++c) { t[c] = sigma * x[c] - qv[c] + t[c]; p[c] = T(0); }         // :136
#pragma unroll
        for (int r = 0; r < RS; ++r) {                                                            // forward sweep  y = W t  (upper part = W')
            const int j = g + 8 * r; const T tj = nget(t, j);
#pragma unroll
            for (int c = 0; c < NB; ++c) if (j <= 64 * c + lane) p[c] += sr[r][c] * tj;
        }
        nsum(p, scrN[1], y1);
#pragma unroll
        for (int c = 0; c < NB; ++c) p[c] = T(0);
#pragma unroll
        for (int r = 0; r < RS; ++r) {                                                            // backward sweep  x~ = W' y  (lower part = W)
            const int i = g + 8 * r; const T yi = nget(y1, i);
#pragma unroll
            for (int c = 0; c < NB; ++c) if (i >= 64 * c + lane) p[c] += sr[r][c] * yi;
        }
        nsum(p, scrN[0], xt);
        av(xt, zt);                                                                               // :139  z~ = A x~
#pragma unroll
        for (int c = 0; c < NB; ++c) { xp[c] = x[c]; x[c] = alpha * xt[c] + alpha1 * x[c]; }      // SolveQuadraticProgram.jl:56-57
#pragma unroll
        for (int b = 0; b < MB; ++b) {                                                            // :59-61
            const T zo = z[b], yo = y[b];
            zp[b] = zo;
            const T tt = alpha * zt[b] + alpha1 * zo + rho1 * yo;
            const T zn = tt > uv[b] ? uv[b] : (tt < lv[b] ? lv[b] : tt);
            z[b] = zn;
            y[b] = yo + rho * (alpha * zt[b] + alpha1 * zo - zn);
        }
        if (it % a.numItrConv == 0) {                                                             // :63  CheckConvergence :79-112
            T Ax[MB], Px[NB], Aty[NB];
            // av / atv / nsum read every wave's row of their scratch buffer after ONE barrier and have no trailing barrier: two
            // consecutive uses of the SAME buffer need a barrier in between (scrN ping-pongs [0]/[1]; scrM was last read by
            // av(xt, zt) above with only register work since) or a fast wave overwrites rows a slower wave is still summing
            __syncthreads();
            av(x, Ax);
#pragma unroll
            for (int c = 0; c < NB; ++c) p[c] = T(0);
#pragma unroll 2
            for (int r = 0; r < RS; ++r) {                                                        // P x (P symmetric: column accumulation over its rows); rare path: keep its register footprint small
                const T xi = nget(x, g + 8 * r);
#pragma unroll
                for (int c = 0; c < NB; ++c) p[c] += P[(int64_t)(g + 8 * r) * NPc + 64 * c + lane] * xi;
            }
            nsum(p, scrN[1], Px);
            atv(y, scrN[0], Aty);
            if (tid < 9) nrm[tid] = 0ull;
            __syncthreads();
            if (g == 0) {                                                                         // every wave holds the same vectors: one wave reduces
                // one norm at a time (the nine maxima alive together cost 18 VGPRs of a kernel that is at its register limit)
                auto put = [&](int k, unsigned long long v) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const unsigned long long tv = __shfl_xor(v, o, 64); v = tv > v ? tv : v; }
                    if (lane == 0) nrm[k] = v;
                };
                unsigned long long v = 0;
#pragma unroll
                for (int b = 0; b < MB; ++b) if (64 * b + lane < a.m) v = max(v, absbits_s((double)(Ax[b] - z[b])));
                put(0, v); v = 0;
#pragma unroll
                for (int b = 0; b < MB; ++b) if (64 * b + lane < a.m) v = max(v, absbits_s((double)Ax[b]));
                put(2, v); v = 0;
#pragma unroll
                for (int b = 0; b < MB; ++b) if (64 * b + lane < a.m) v = max(v, absbits_s((double)z[b]));
                put(3, v); v = 0;
#pragma unroll
                for (int b = 0; b < MB; ++b) if (64 * b + lane < a.m) v = max(v, absbits_s((double)(z[b] - zp[b])));
                put(8, v); v = 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) if (64 * c + lane < a.n) v = max(v, absbits_s((double)(Px[c] + qv[c] + Aty[c])));
                put(1, v); v = 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) if (64 * c + lane < a.n) v = max(v, absbits_s((double)Px[c]));
                put(4, v); v = 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) if (64 * c + lane < a.n) v = max(v, absbits_s((double)Aty[c]));
                put(5, v); v = 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) if (64 * c + lane < a.n) v = max(v, absbits_s((double)qv[c]));
                put(6, v); v = 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) if (64 * c + lane < a.n) v = max(v, absbits_s((double)(x[c] - xp[c])));
                put(7, v);
            }
            __syncthreads();
            if (tid == 0) {
                double nv[9];
                for (int k = 0; k < 9; ++k) nv[k] = __longlong_as_double((long long)nrm[k]);
                const double normResPrim = nv[0], normResDual = nv[1];                            // :85-86
                const double maxNormPrim = jmax_s(nv[2], nv[3]);                                  // :88
                const double maxNormDual = jmax_s(jmax_s(nv[4], nv[5]), nv[6]);                   // :89
                double rr = sh_rhorho;
                if (a.adptRho) {                                                                  // :92-96
                    const double tv = a.rho * sqrt((normResPrim * maxNormDual) / (normResDual * maxNormPrim));
                    rr = tv > 1e6 ? 1e6 : (tv < 1e-3 ? 1e-3 : tv);
                }
                int flag = 1;
                if ((normResPrim < a.epsAbs + a.epsRel * maxNormPrim) && (normResDual < a.epsAbs + a.epsRel * maxNormDual)) flag = 3;   // :102-104
                if ((nv[7] <= a.epsAdmm) && (nv[8] <= a.epsAdmm)) flag = 2;                       // :105-107 (not else)
                sh_rhorho = rr; sh_flag = flag;
                sh_need = (flag == 1 && a.adptRho && ((rr * a.fctrRho < a.rho) || (rr > a.fctrRho * a.rho))) ? 1 : 0;   // :47
                sres[0] = normResPrim; sres[1] = normResDual; sres[2] = maxNormPrim; sres[3] = maxNormDual; sres[4] = rr; sres[5] = flag;
                sres[6] = nv[7]; sres[7] = nv[8];
            }
            __syncthreads();
            if (sh_flag != 1 || sh_need) break;
        }
    }
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) { gx[64 * c + lane] = x[c]; gxp[64 * c + lane] = xp[c]; }
#pragma unroll
        for (int b = 0; b < MB; ++b) { gz[64 * b + lane] = z[b]; gy[64 * b + lane] = y[b]; }
    }
    if (tid == 0) {
        out->last_it = it; out->convFlag = sh_flag; out->need_rho = sh_need;
        for (int k = 0; k < 8; ++k) out->res[k] = sres[k];
        out->res[4] = sh_rhorho;
    }
}

template <typename T, int NB, int MB>
__global__ __launch_bounds__(512) void k_admm_small_reg(SmallArgs a, const T* __restrict__ A, const T* __restrict__ P, const T* __restrict__ S,
                                                        const T* __restrict__ q, const T* __restrict__ l, const T* __restrict__ u,
                                                        T* __restrict__ gx, T* __restrict__ gxp, T* __restrict__ gz, T* __restrict__ gy,
                                                        SmallOut* __restrict__ out) {
    small_reg_body<T, NB, MB>(a, A, P, S, q, l, u, gx, gxp, gz, gy, out);
}
// batch of independent QPs of one shape: workgroup b runs QP b with its own arguments (iteration window, rho, ...); a QP whose
// window is empty (finished, or waiting for nothing) returns at once.  Matrices NP*NP / MP*NP apart, vectors NP / MP apart.
template <typename T, int NB, int MB>
__global__ __launch_bounds__(512) void k_admm_small_reg_batch(const SmallArgs* __restrict__ args, const T* __restrict__ A, const T* __restrict__ P,
                                                              const T* __restrict__ S, const T* __restrict__ q, const T* __restrict__ l,
                                                              const T* __restrict__ u, T* __restrict__ gx, T* __restrict__ gxp,
                                                              T* __restrict__ gz, T* __restrict__ gy, SmallOut* __restrict__ out) {
    const int64_t b = blockIdx.x;
    const SmallArgs a = args[b];
    if (a.it_begin >= a.it_end) return;
    constexpr int64_t NPc = 64 * NB, MPc = 64 * MB;
    small_reg_body<T, NB, MB>(a, A + b * MPc * NPc, P + b * NPc * NPc, S + b * NPc * NPc, q + b * NPc, l + b * MPc, u + b * MPc,
                              gx + b * NPc, gxp + b * NPc, gz + b * MPc, gy + b * MPc, out + b);
}

// At[c][r] = A[r][c]  (A row-major MP x NP, At row-major NP x MP); sizes are small here
template <typename T> __global__ void k_transpose_small(const T* __restrict__ A, int NP, int MP, T* __restrict__ At) {
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < MP) At[(int64_t)c * MP + r] = A[(int64_t)r * NP + c];
}

template <typename T> size_t small_lds_bytes(int NP, int MP) { return sizeof(T) * ((size_t)6 * NP + (size_t)7 * MP + 1024); }
template <typename T> size_t small_lds_mat_bytes(int NP, int MP) { return sizeof(T) * ((size_t)MP * (NP + 1) + (size_t)NP * NP); }

}  // namespace

template <typename T> bool admm_small_supported(int n, int m, int NP, int MP) {
    if (m < 1 || NP > 512 || MP > 2048) return false;
    if (small_lds_bytes<T>(NP, MP) > 150 * 1024) return false;
    const double bytes = ((double)2 * MP * NP + (double)NP * NP) * sizeof(T);
    return bytes <= 1.25 * 1024 * 1024;       // beyond ~1 MiB per iteration one CU's L2 bandwidth loses to the multi-CU loop
}

template <typename T>
void admm_small(hipStream_t st, int n, int m, int NP, int MP, int it_begin, int it_end, int numItrConv, int adptRho, double rho,
                double rhorho, double sigma, double alpha, double epsAbs, double epsRel, double epsAdmm, double fctrRho, const T* A,
                const T* At, const T* P, const T* S, const T* q, const T* l, const T* u, T* x, T* xp, T* z, T* y, void* out_dev) {
    SmallArgs a{n, m, NP, MP, it_begin, it_end, numItrConv, adptRho, rho, rhorho, sigma, alpha, epsAbs, epsRel, epsAdmm, fctrRho};
    static const int reg_env = [] { const char* e = getenv("QPS_SMALL_REG"); return e ? atoi(e) : 1; }();
    // matrices in registers: n <= 64 with m <= 128 (fp64) / 256 (fp32); n <= 128 with m <= 64 (fp64) / 128 (fp32) -- beyond that the
    // per-thread share of A (twice) and S no longer fits the 256 VGPRs of an 8-wave workgroup without spilling
    if (reg_env && MP >= 64 && (NP == 64 || NP == 128)) {
        SmallOut* o = reinterpret_cast<SmallOut*>(out_dev);
        const int mb = MP / 64, mb_max = sizeof(T) == 8 ? 2 : 4;
#define QPS_REG(NBv, MBv) hipLaunchKernelGGL((k_admm_small_reg<T, NBv, MBv>), dim3(1), dim3(512), 0, st, a, A, P, S, q, l, u, x, xp, z, y, o)
        if (mb <= mb_max) {
            if (NP == 64) {
                if (mb == 1) QPS_REG(1, 1); else if (mb == 2) QPS_REG(1, 2);
                else if constexpr (sizeof(T) == 4) { if (mb == 3) QPS_REG(1, 3); else QPS_REG(1, 4); }
            } else {
                if (mb == 1) QPS_REG(2, 1); else if (mb == 2) QPS_REG(2, 2);
                else if constexpr (sizeof(T) == 4) { if (mb == 3) QPS_REG(2, 3); else QPS_REG(2, 4); }
            }
            return;
        }
#undef QPS_REG
    }
    size_t lds = small_lds_bytes<T>(NP, MP);
    static const int lm_env = [] { const char* e = getenv("QPS_SMALL_LDSMAT"); return e ? atoi(e) : 1; }();
    const bool lm = lm_env && (lds + small_lds_mat_bytes<T>(NP, MP) <= 158 * 1024);
    if (lm) lds += small_lds_mat_bytes<T>(NP, MP);
    static const int th_env = [] { const char* e = getenv("QPS_SMALL_THREADS"); return e ? atoi(e) : 0; }();
    // few waves for tiny problems (a workgroup barrier costs with the number of waves), 1024 threads once there is work for them
    const int th = th_env > 0 ? th_env : ((int64_t)MP * NP <= 65536 ? 512 : 1024);   // measured at n = 10 / 64 / 100: 512 beats 256 and 1024
    static PerDeviceOnce attr_set[2][3][2];   // the LDS attribute is a setting of the DEVICE's code object: once per device ordinal
    const int ti = sizeof(T) == 8 ? 0 : 1, dev_ = current_device();
#define QPS_SMALL(THN, IDX, LMV)                                                                                                             \
    do {                                                                                                                                     \
        attr_set[ti][IDX][LMV].once(dev_, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_admm_small<T, THN, (LMV) != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((k_admm_small<T, THN, (LMV) != 0>), dim3(1), dim3(THN), lds, st, a, A, At, P, S, q, l, u, x, xp, z, y, reinterpret_cast<SmallOut*>(out_dev)); \
    } while (0)
#define QPS_SMALL2(THN, IDX) do { if (lm) QPS_SMALL(THN, IDX, 1); else QPS_SMALL(THN, IDX, 0); } while (0)
    if (th <= 256) QPS_SMALL2(256, 0); else if (th <= 512) QPS_SMALL2(512, 1); else QPS_SMALL2(1024, 2);
#undef QPS_SMALL2
#undef QPS_SMALL
}

template <typename T> void transpose_small(hipStream_t st, const T* A, int NP, int MP, T* At) {
    if (NP <= 0 || MP <= 0) return;
    hipLaunchKernelGGL((k_transpose_small<T>), dim3((MP + 255) / 256, NP), dim3(256), 0, st, A, NP, MP, At);
}

template <typename T> bool admm_small_batch_supported(int NP, int MP) {
    static const int reg_env = [] { const char* e = getenv("QPS_SMALL_REG"); return e ? atoi(e) : 1; }();
    return reg_env && MP >= 64 && (NP == 64 || NP == 128) && MP / 64 <= (sizeof(T) == 8 ? 2 : 4);
}
size_t admm_small_args_bytes() { return sizeof(SmallArgs); }
void admm_small_args_set(void* host_array, int idx, int n, int m, int NP, int MP, int it_begin, int it_end, int numItrConv, int adptRho, double rho,
                         double rhorho, double sigma, double alpha, double epsAbs, double epsRel, double epsAdmm, double fctrRho) {
    reinterpret_cast<SmallArgs*>(host_array)[idx] = SmallArgs{n, m, NP, MP, it_begin, it_end, numItrConv, adptRho, rho, rhorho, sigma, alpha, epsAbs, epsRel, epsAdmm, fctrRho};
}
template <typename T>
void admm_small_batch(hipStream_t st, int count, int NP, int MP, const void* args_dev, const T* A, const T* P, const T* S, const T* q, const T* l,
                      const T* u, T* x, T* xp, T* z, T* y, void* outs_dev) {
    const SmallArgs* a = reinterpret_cast<const SmallArgs*>(args_dev);
    SmallOut* o = reinterpret_cast<SmallOut*>(outs_dev);
    const int mb = MP / 64;
#define QPS_REGB(NBv, MBv) hipLaunchKernelGGL((k_admm_small_reg_batch<T, NBv, MBv>), dim3(count), dim3(512), 0, st, a, A, P, S, q, l, u, x, xp, z, y, o)
    if (NP == 64) {
        if (mb == 1) QPS_REGB(1, 1); else if (mb == 2) QPS_REGB(1, 2);
        else if constexpr (sizeof(T) == 4) { if (mb == 3) QPS_REGB(1, 3); else QPS_REGB(1, 4); }
    } else {
        if (mb == 1) QPS_REGB(2, 1); else if (mb == 2) QPS_REGB(2, 2);
        else if constexpr (sizeof(T) == 4) { if (mb == 3) QPS_REGB(2, 3); else QPS_REGB(2, 4); }
    }
#undef QPS_REGB
}
size_t admm_small_out_bytes() { return sizeof(SmallOut); }
void admm_small_read(const void* host_copy, int* last_it, int* convFlag, int* need_rho, double* res8) {
    const SmallOut* o = reinterpret_cast<const SmallOut*>(host_copy);
    *last_it = o->last_it; *convFlag = o->convFlag; *need_rho = o->need_rho;
    for (int k = 0; k < 8; ++k) res8[k] = o->res[k];
}

#define INST(T)                                                                                                                    \
    template bool admm_small_batch_supported<T>(int, int);                                                                         \
    template void admm_small_batch<T>(hipStream_t, int, int, int, const void*, const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, T*, T*, void*); \
    template void transpose_small<T>(hipStream_t, const T*, int, int, T*);                                                         \
    template bool admm_small_supported<T>(int, int, int, int);                                                                     \
    template void admm_small<T>(hipStream_t, int, int, int, int, int, int, int, int, double, double, double, double, double, double, \
                                double, double, const T*, const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, T*, T*, void*);
INST(double)
INST(float)
#undef INST

}  // namespace qps
