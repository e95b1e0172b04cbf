// k_ldl.hip -- sparse direct KKT plugin on the device: numeric L D L' of
//     K = [P + sigma I  A'; A  -rho^-1 I]                                   (LinearSystemSolvers.jl:18 / :49 / :81)
// and the two triangular solves of every ADMM iteration (:39 / :70 / :102), on the layout ldl_symbolic.h prepares:
//
//   sparse part (columns of the wide elimination-tree levels): scalar L in CSR and CSC.  The columns of a level are independent,
//     so a level is ONE launch -- numeric: every entry L_ij = (K_ij - sum_k L_ik D_k L_jk) / D_j is a sparse dot product of two
//     rows of L (sorted-list intersection, the shorter list drives a binary search in the longer), 16 lanes per entry, fixed
//     summation order; solves: forward = row gathers over the CSR, backward = column gathers over the CSC (no atomics anywhere:
//     results are bitwise reproducible);
//   tail (the narrow levels near the root: long dependency chains, dense fill): one dense matrix.  Its Schur complement
//     K_tt - L_ts D_s L_ts' is formed by the MFMA GEMM on densified column chunks of L_ts, factorised by the dense (signed)
//     Cholesky, explicitly inverted by recursive doubling, and every iteration applies it as two triangular GEMVs -- the same
//     kernels the dense reduced-form path uses (k_setup.hip, k_loop.hip).
//
// K is quasi-definite: pivots of the n variables are positive, those of the m constraint rows negative, for every ordering, so the
// tail is factorised as Lt J Lt' with J = diag(+-1) known beforehand and no pivoting is needed (what QDLDL relies on).
#include <algorithm>
#include <cstdlib>

#include "qps_kernels.h"
#include "qps_ldl.h"
#include "wave_reduce.h"

namespace qps {

namespace {

constexpr int GL = 16;   // lanes per entry / column in the numeric kernels (one DPP row)

template <typename T>
__global__ __launch_bounds__(256) void k_ldl_scatter(int64_t cnt, const int64_t* __restrict__ dst, const int* __restrict__ src, const T* __restrict__ kval,
                                                     T* __restrict__ vr, T* __restrict__ Mt) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    const int64_t d = dst[e];
    const T v = kval[src[e]];
    if (d >= 0) vr[d] = v; else Mt[-(d + 1)] = v;
}
// diagonal of K: P_ii + sigma for a variable, -1/rho for a constraint row; padding of the dense tail gets a unit diagonal
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_diag0(int N, int Ns, int ldt, const int* __restrict__ dposP, const signed char* __restrict__ sign,
                                                   const T* __restrict__ kval, T sigma, T neg_rho1, T* __restrict__ D0, T* __restrict__ Mt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ns + ldt) return;
    if (i >= N) { Mt[(int64_t)(i - Ns) * ldt + (i - Ns)] = T(1); return; }
    const T d = sign[i] > 0 ? ((dposP[i] >= 0 ? kval[dposP[i]] : T(0)) + sigma) : neg_rho1;
    if (i < Ns) D0[i] = d; else Mt[(int64_t)(i - Ns) * ldt + (i - Ns)] = d;
}
// D_j = K_jj - sum_k L_jk^2 D_k for the columns [c0, c1) of one level (their rows of L only hold columns of earlier levels)
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_level_diag(int c0, int c1, const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ vr,
                                                        const T* __restrict__ D0, T* __restrict__ D, T* __restrict__ Dinv,
                                                        const signed char* __restrict__ sign, int* __restrict__ fail) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GL; const int lane = threadIdx.x % GL;   // 64-bit: a level may hold more than 2^28 entries
    const bool valid = (int64_t)c0 + g < (int64_t)c1;              // tested before narrowing
    const int j = valid ? (int)(c0 + g) : c0;
    T s = T(0);
    if (valid)
        for (int k = rp[j] + lane; k < rp[j + 1]; k += GL) { const T l = vr[k]; s += l * l * D[ci[k]]; }
    s = row16_sum_last(s);
    if (valid && lane == GL - 1) {
        const T d = D0[j] - s;
        if (!((T)sign[j] * d > T(0))) atomicCAS(fail, 0, j + 1);
        D[j] = d; Dinv[j] = T(1) / d;
    }
}
// L_ij for the CSC entries [q0, q1) = the columns of one level
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_level_offdiag(int q0, int q1, const int* __restrict__ rp, const int* __restrict__ ci, const int* __restrict__ ri,
                                                           const int* __restrict__ cj, const int* __restrict__ csc2csr, const T* __restrict__ D,
                                                           const T* __restrict__ Dinv, T* __restrict__ vr, T* __restrict__ vc) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GL; const int lane = threadIdx.x % GL;   // 64-bit: a level may hold more than 2^28 entries
    const bool valid = (int64_t)q0 + g < (int64_t)q1;              // tested before narrowing
    const int q = valid ? (int)(q0 + g) : q0;
    T s = T(0); int p = 0, j = 0;
    if (valid) {
        const int i = ri[q]; j = cj[q]; p = csc2csr[q];
        int a0 = rp[j], a1 = rp[j + 1], b0 = rp[i], b1 = p;              // row j (all of it), row i up to column j (exclusive)
        if (a1 - a0 > b1 - b0) { const int t0 = a0, t1 = a1; a0 = b0; a1 = b1; b0 = t0; b1 = t1; }   // a = the shorter list drives
        for (int t = a0 + lane; t < a1; t += GL) {
            const int k = ci[t];
            int lo = b0, hi = b1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ci[mid] < k) lo = mid + 1; else hi = mid; }
            if (lo < b1 && ci[lo] == k) s += vr[t] * D[k] * vr[lo];
        }
    }
    s = row16_sum_last(s);
    if (valid && lane == GL - 1) { const T v = (vr[p] - s) * Dinv[j]; vr[p] = v; vc[q] = v; }
}
// dense panels of L_ts for one chunk of tail-touching columns: Ld[i][kk] = L_{Ns+i, col(kk)}, Wd = Ld * D
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_densify(int e0, int e1, const int* __restrict__ ent_q, const int* __restrict__ ent_dst, const int* __restrict__ cj,
                                                     const T* __restrict__ vc, const T* __restrict__ D, T* __restrict__ Ld, T* __restrict__ Wd) {
    const int e = e0 + blockIdx.x * 256 + threadIdx.x;
    if (e >= e1) return;
    const int q = ent_q[e]; const T v = vc[q];
    Ld[ent_dst[e]] = v; Wd[ent_dst[e]] = v * D[cj[q]];
}
// Mt -= sum over the K-slices of one group of the partial products (fixed order: bitwise reproducible), lower tiles only
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_sub_slabs(int ldt, int nslab, const T* __restrict__ slabs, T* __restrict__ Mt) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= ldt || j > (i | 63)) return;
    const int64_t e = (int64_t)i * ldt + j, tt = (int64_t)ldt * ldt;
    T s = T(0);
    for (int c = 0; c < nslab; ++c) s += slabs[(int64_t)c * tt + e];
    Mt[e] -= s;
}
// u <- J u between the two sweeps of a tail with constraint rows in it (K_tt = Lt J Lt': x = inv(Lt)' J inv(Lt) t).  The signs
// cannot live in the sweep matrix: its diagonal is shared by the lower (forward) and the upper (backward) triangle.
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_mul_sign(int ldt, const T* __restrict__ sgn, T* __restrict__ u) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ldt && sgn[i] < T(0)) u[i] = -u[i];
}

// ---- per-iteration kernels ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_rhs(int N, int n, const int* __restrict__ perm, const T* __restrict__ x, const T* __restrict__ q,
                                                 const T* __restrict__ z, const T* __restrict__ y, T sigma, T rho1, T* __restrict__ b) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int o = perm[k];
    b[k] = o < n ? sigma * x[o] - q[o] : z[o - n] - rho1 * y[o - n];          // LinearSystemSolvers.jl:37-38
}
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_rhs_raw(int N, int n, const int* __restrict__ perm, const T* __restrict__ r1, const T* __restrict__ r2, T* __restrict__ b) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int o = perm[k];
    b[k] = o < n ? r1[o] : r2[o - n];
}
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_post_raw(int n, int m, int Ns, const int* __restrict__ iperm, const T* __restrict__ b, const T* __restrict__ tx,
                                                      T* __restrict__ ox, T* __restrict__ onu) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n + m) return;
    const int k = iperm[o];
    const T v = k >= Ns ? tx[k - Ns] : b[k];
    if (o < n) ox[o] = v; else onu[o - n] = v;
}
template <typename T, int LPR> __device__ __forceinline__ T group_sum(T s) {
    if (LPR == 4) return quad_sum_all(s);
    if (LPR == 16) return row16_sum_last(s);       // valid in the last lane of the group
    if (LPR == 64) return wave_sum_all(s);
    if (LPR == 256) {                              // one workgroup per row / column (long tail rows): wave sums, then a fixed-order sum of the four
        __shared__ T wsum4[4];
        s = wave_sum_all(s);
        if ((threadIdx.x & 63) == 0) wsum4[threadIdx.x >> 6] = s;
        __syncthreads();
        return (wsum4[0] + wsum4[1]) + (wsum4[2] + wsum4[3]);
    }
    return s;
}
// forward substitution for the rows [r0, r1) of one level (or of the tail): b_r -= sum_k L_rk b_k, all k in earlier levels
template <typename T, int LPR>
__global__ __launch_bounds__(256) void k_ldl_fwd(int r0, int r1, int Ns, const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ vr,
                                                 T* __restrict__ b, T* __restrict__ tb) {
    const int g = (blockIdx.x * 256 + threadIdx.x) / LPR, lane = threadIdx.x % LPR;
    const int r = r0 + g;
    const bool valid = r < r1;
    T s = T(0);
    if (valid) {
        // four independent gathers in flight per lane (a single accumulator makes every load wait for the previous add); the partial sums
        // are added in a fixed order
        T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
        int k = rp[r] + lane; const int ke = rp[r + 1];
        for (; k + 3 * LPR < ke; k += 4 * LPR) {
            s0 += vr[k] * b[ci[k]]; s1 += vr[k + LPR] * b[ci[k + LPR]]; s2 += vr[k + 2 * LPR] * b[ci[k + 2 * LPR]]; s3 += vr[k + 3 * LPR] * b[ci[k + 3 * LPR]];
        }
        for (; k < ke; k += LPR) s0 += vr[k] * b[ci[k]];
        s = (s0 + s1) + (s2 + s3);
    }
    s = group_sum<T, LPR>(s);
    if (valid && lane == LPR - 1) { const T v = b[r] - s; if (r >= Ns) tb[r - Ns] = v; else b[r] = v; }
}
// backward substitution for the columns [c0, c1) of one level: x_j = y_j / D_j - sum_i L_ij x_i, all i in later levels or the tail
template <typename T, int LPR>
__global__ __launch_bounds__(256) void k_ldl_bwd(int c0, int c1, int Ns, const int* __restrict__ cp, const int* __restrict__ ri, const T* __restrict__ vc,
                                                 const T* __restrict__ Dinv, T* __restrict__ b, const T* __restrict__ tx) {
    const int g = (blockIdx.x * 256 + threadIdx.x) / LPR, lane = threadIdx.x % LPR;
    const bool valid = (int64_t)c0 + g < (int64_t)c1;              // tested before narrowing
    const int j = valid ? (int)(c0 + g) : c0;
    T s = T(0);
    if (valid) {
        T s0 = T(0), s1 = T(0);
        auto xv = [&](int i) { return i >= Ns ? tx[i - Ns] : b[i]; };
        int q = cp[j] + lane; const int qe = cp[j + 1];
        for (; q + LPR < qe; q += 2 * LPR) { s0 += vc[q] * xv(ri[q]); s1 += vc[q + LPR] * xv(ri[q + LPR]); }
        for (; q < qe; q += LPR) s0 += vc[q] * xv(ri[q]);
        s = s0 + s1;
    }
    s = group_sum<T, LPR>(s);
    if (valid && lane == LPR - 1) b[j] = b[j] * Dinv[j] - s;
}
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_post(int n, int m, int Ns, const int* __restrict__ iperm, const T* __restrict__ b, const T* __restrict__ tx,
                                                  const T* __restrict__ z, const T* __restrict__ y, T rho1, T* __restrict__ xx, T* __restrict__ zz) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n + m) return;
    const int k = iperm[o];
    const T v = k >= Ns ? tx[k - Ns] : b[k];
    if (o < n) xx[o] = v;
    else { const int r = o - n; zz[r] = z[r] + rho1 * (v - y[r]); }           // LinearSystemSolvers.jl:40  nu -> z~
}

// un-permute, nu -> z~ (LinearSystemSolvers.jl:40), the ADMM updates (SolveQuadraticProgram.jl:56-61, same expressions as k_admm_update) and
// the permuted right-hand side of the NEXT iteration (LinearSystemSolvers.jl:37-38) in one launch; thread o owns entry o of [x; nu]
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_post_update(int n, int m, int Ns, const int* __restrict__ iperm, T* __restrict__ b, const T* __restrict__ tx,
                                                         const T* __restrict__ q, T* __restrict__ x, T* __restrict__ xp, T* __restrict__ z, T* __restrict__ zp,
                                                         T* __restrict__ y, const T* __restrict__ l, const T* __restrict__ u, T alpha, T rho, T sigma) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n + m) return;
    const int k = iperm[o];
    const T v = k >= Ns ? tx[k - Ns] : b[k];                                   // entry o of the solution: every thread reads its own b[k] before it overwrites it
    const T alpha1 = T(1) - alpha, rho1 = T(1) / rho;
    if (o < n) {
        const T xo = x[o];
        xp[o] = xo;                                                           // :56
        const T xn = alpha * v + alpha1 * xo;                                 // :57
        x[o] = xn;
        b[k] = sigma * xn - q[o];                                             // next :37
    } else {
        const int r = o - n;
        const T zo = z[r], yo = y[r];
        const T zt = zo + rho1 * (v - yo);                                    // LinearSystemSolvers.jl:40  nu -> z~
        zp[r] = zo;                                                           // :59
        const T t = alpha * zt + alpha1 * zo + rho1 * yo;                     // :60
        const T lo = l[r], hi = u[r];
        const T zn = t > hi ? hi : (t < lo ? lo : t);
        z[r] = zn;
        const T yn = yo + rho * (alpha * zt + alpha1 * zo - zn);              // :61
        y[r] = yn;
        b[k] = zn - rho1 * yn;                                                // next :38
    }
}

// ---- dense signed Cholesky of the tail (one 64-column step = diagonal block, panel, MFMA trailing update) -----------------
// 64 x 64 block: M = Lt J Lt' with J = diag(sgn).  Column k: d = sgn_k a_kk > 0, Lt_kk = sqrt(d), Lt_ik = sgn_k a_ik / sqrt(d);
// trailing a_ij -= Lt_ik sgn_k Lt_jk.  The inverse of Lt (lower) follows by forward substitution on the identity.
template <typename T>
__global__ __launch_bounds__(256) void k_spotrf64(T* __restrict__ M, int64_t ld, int kb, const T* __restrict__ sgn, T* __restrict__ dinv, int* __restrict__ fail) {
    __shared__ T a[64][65];
    __shared__ T x[64][65];
    __shared__ T sg[64];
    const int t = threadIdx.x, c = t & 63, g = t >> 6;
    T* blk = M + (int64_t)kb * 64 * ld + kb * 64;
    for (int r = g; r < 64; r += 4) { a[r][c] = blk[(int64_t)r * ld + c]; x[r][c] = (r == c) ? T(1) : T(0); }
    if (t < 64) sg[t] = sgn[kb * 64 + t];
    __syncthreads();
    for (int k = 0; k < 64; ++k) {
        T d = sg[k] * a[k][k];
        if (!(d > T(0))) { if (t == 0) atomicCAS(fail, 0, kb * 64 + k + 1); d = T(1); }
        const T rs = T(1) / sqrt(d);
        __syncthreads();
        if (t >= k && t < 64) a[t][k] = (t == k) ? d * rs : sg[k] * a[t][k] * rs;
        __syncthreads();
        // trailing update of the lower triangle: entry (i, j), k < j <= i
        for (int e = t; e < 64 * 64; e += 256) {
            const int i = e >> 6, j = e & 63;
            if (j > k && i >= j) a[i][j] -= a[i][k] * sg[k] * a[j][k];
        }
        __syncthreads();
    }
    // inverse: row i of X = (e_i - sum_{p<i} L_ip X_p) / L_ii, rows in order; thread (c, g) owns columns c of rows handled by all
    for (int i = 0; i < 64; ++i) {
        if (t < 64) {
            T s = (i == c) ? T(1) : T(0);
            for (int p = 0; p < i; ++p) s -= a[i][p] * x[p][c];
            x[i][c] = (c <= i) ? s / a[i][i] : T(0);
        }
        __syncthreads();
    }
    for (int r = g; r < 64; r += 4) {
        blk[(int64_t)r * ld + c] = (c <= r) ? a[r][c] : T(0);
        dinv[(int64_t)kb * 4096 + r * 64 + c] = x[r][c];
    }
}
// panel: Lt21 = A21 inv(Lt11)' J (in place), W21 = Lt21 J for the trailing update  A22 -= Lt21 J Lt21' = W21 Lt21'
template <typename T>
__global__ __launch_bounds__(256) void k_spanel(T* __restrict__ M, int64_t ld, int kb, const T* __restrict__ sgn, const T* __restrict__ dinv, T* __restrict__ W) {
    __shared__ T a[64][65];
    __shared__ T iv[64][65];
    const int t = threadIdx.x, c = t & 63, g = t >> 6;
    const int64_t row0 = (int64_t)(kb + 1 + blockIdx.x) * 64;
    T* tile = M + row0 * ld + kb * 64;
    T* wt = W + row0 * ld + kb * 64;
    for (int r = g; r < 64; r += 4) { a[r][c] = tile[(int64_t)r * ld + c]; iv[r][c] = dinv[(int64_t)kb * 4096 + r * 64 + c]; }
    __syncthreads();
    const T sj = sgn[kb * 64 + c];
    for (int r = g; r < 64; r += 4) {
        T s = T(0);
        for (int k = 0; k <= c; ++k) s += a[r][k] * iv[c][k];       // (A21 inv(Lt11)')[r][c], inv lower triangular
        const T v = s * sj;
        tile[(int64_t)r * ld + c] = v; wt[(int64_t)r * ld + c] = v * sj;
    }
}

template <typename T> struct DevVec {
    T* p = nullptr; int64_t n = 0;
    void alloc(int64_t count, hipStream_t st) { n = count; p = dalloc<T>(count, st); }
    void upload(const std::vector<T>& h, StagedUploader& up) { alloc((int64_t)h.size(), up.st); if (!h.empty()) up.copy(p, h.data(), sizeof(T) * h.size()); }
    ~DevVec() { if (p) (void)hipFree(p); }
};

int pick_lpr(int64_t nnz, int rows) {
    if (rows <= 0) return 1;
    const double avg = (double)nnz / rows;
    return avg <= 2.0 ? 1 : (avg <= 12.0 ? 4 : (avg <= 96.0 ? 16 : (avg <= 1024.0 || rows > 2048 ? 64 : 256)));
}

template <typename T> struct SparseLdlImpl : SparseLdl<T> {
    hipStream_t st; LdlSymbolic S;
    DevVec<int> rp, ci, cp, ri, cj, csc2csr, ksrc, dposP, perm, iperm, ent_q, ent_dst;
    DevVec<int64_t> kdst; DevVec<signed char> sign;
    DevVec<T> kval, vr, vc, D0, D, Dinv, b, tb, tu, tx, tsgn, Mt, St, tmp, dinv, Ld, Wd;
    DevVec<int> fail;
    std::vector<int> group_ptr; int KC = 16, GS = 1, nb = 64; bool tail_signed = false;   // Schur complement: groups of GS K-slices of KC columns
    DevVec<T> slabs;
    std::vector<int> lpr_fwd, lpr_bwd; int lpr_tail = 1;

    SparseLdlImpl(hipStream_t st_, LdlSymbolic&& sym, const double* Pv, int64_t pnnz, const double* Av, int64_t annz) : st(st_), S(std::move(sym)) {
        const int N = S.N, Ns = S.Ns, Nt = S.Nt, ldt = S.ldt;
        StagedUploader up(st);   // host arrays -> device on the handle's own stream (qps_internal.h, stream-ordering rule)
        rp.upload(S.rp, up); ci.upload(S.ci, up); cp.upload(S.cp, up); ri.upload(S.ri, up);
        {
            std::vector<int> cjv(S.ri.size()), inv(S.ri.size());
            for (int j = 0; j < Ns; ++j) for (int q = S.cp[j]; q < S.cp[j + 1]; ++q) cjv[q] = j;
            for (size_t k = 0; k < S.csr2csc.size(); ++k) inv[S.csr2csc[k]] = (int)k;
            cj.upload(cjv, up); csc2csr.upload(inv, up);
        }
        kdst.upload(S.k_dst, up); ksrc.upload(S.k_src, up); dposP.upload(S.dpos_P, up); sign.upload(S.sign, up); perm.upload(S.perm, up); iperm.upload(S.iperm, up);
        {
            std::vector<T> kv((size_t)(pnnz + annz));
            for (int64_t k = 0; k < pnnz; ++k) kv[k] = (T)Pv[k];
            for (int64_t k = 0; k < annz; ++k) kv[pnnz + k] = (T)Av[k];
            kval.upload(kv, up);
        }
        const int64_t nz = (int64_t)S.ci.size();
        vr.alloc(nz, st); vc.alloc(nz, st); D0.alloc(N, st); D.alloc(N, st); Dinv.alloc(N, st); b.alloc(N, st); fail.alloc(4, st);
        tb.alloc(ldt + 64, st); tu.alloc(ldt + 64, st); tx.alloc(ldt + 64, st);
        if (ldt > 0) {
            const int64_t tt = (int64_t)ldt * ldt;
            Mt.alloc(tt, st); St.alloc(tt, st); tmp.alloc(tt, st); dinv.alloc((int64_t)(ldt / 64) * 4096, st);
            std::vector<T> sg(ldt, T(1));
            for (int i = 0; i < Nt; ++i) { sg[i] = (T)S.sign[Ns + i]; if (S.sign[Ns + i] < 0) tail_signed = true; }
            tsgn.upload(sg, up);
            nb = 64; while (nb < ldt) nb *= 2;                                     // one inverted block covers the whole tail
            // chunks of tail-touching sparse columns for the Schur complement GEMM
            std::vector<int> tcol_first(Ns, -1); int ntc = 0;
            for (int j = 0; j < Ns; ++j) {
                const int* lo = S.ri.data() + S.cp[j]; const int* hi = S.ri.data() + S.cp[j + 1];
                const int* it = std::lower_bound(lo, hi, Ns);
                if (it != hi) { tcol_first[j] = (int)(it - S.ri.data()); ++ntc; }
            }
            // Schur complement of the tail on the MFMA GEMM, split along K: the tail has few 64 x 64 tiles (3 at ldt = 128) and tens of thousands
            // of columns to sum over, so K is cut into slices of KC columns that run as ONE batched launch (blockIdx.z = slice) into per-slice
            // slabs, which k_ldl_sub_slabs subtracts in fixed order; a group = the slices that fit a 1 GiB budget (usually all of them)
            const int64_t tiles = (int64_t)(ldt / 64) * (ldt / 64 + 1) / 2;
            int nsplit = (int)std::min<int64_t>(256, std::max<int64_t>(1, 768 / tiles));
            KC = std::max(64, ((ntc + nsplit - 1) / nsplit + 15) / 16 * 16);
            KC = std::min(KC, 4096);
            nsplit = std::max(1, (ntc + KC - 1) / KC);
            const int64_t budget = ((int64_t)1 << 30) / (int64_t)sizeof(T);                      // elements
            GS = (int)std::max<int64_t>(1, std::min<int64_t>(nsplit, budget / ((int64_t)2 * ldt * KC + (int64_t)ldt * ldt)));
            std::vector<int> eq, ed; group_ptr.assign(1, 0);
            int kk = 0;                                                                        // column index inside the current group
            for (int j = 0; j < Ns; ++j) {
                if (tcol_first[j] < 0) continue;
                for (int q = tcol_first[j]; q < S.cp[j + 1]; ++q) { eq.push_back(q); ed.push_back((S.ri[q] - Ns) * (GS * KC) + kk); }
                if (++kk == GS * KC) { group_ptr.push_back((int)eq.size()); kk = 0; }
            }
            if (kk > 0) group_ptr.push_back((int)eq.size());
            ent_q.upload(eq, up); ent_dst.upload(ed, up);
            if (group_ptr.size() > 1) { Ld.alloc((int64_t)ldt * GS * KC, st); Wd.alloc((int64_t)ldt * GS * KC, st); slabs.alloc((int64_t)GS * ldt * ldt, st); }
        }
        const int L = (int)S.level_ptr.size() - 1;
        lpr_fwd.assign(std::max(L, 0), 1); lpr_bwd.assign(std::max(L, 0), 1);
        for (int l = 0; l < L; ++l) {
            const int c0 = S.level_ptr[l], c1 = S.level_ptr[l + 1];
            lpr_fwd[l] = pick_lpr((int64_t)S.rp[c1] - S.rp[c0], c1 - c0);
            lpr_bwd[l] = pick_lpr((int64_t)S.cp[c1] - S.cp[c0], c1 - c0);
        }
        lpr_tail = pick_lpr((int64_t)S.rp[N] - S.rp[Ns], Nt);
    }
    const LdlSymbolic& symbolic() const override { return S; }
    int launches_per_solve() const override { const int L = (int)S.level_ptr.size() - 1; return 2 + std::max(L - 1, 0) + L + (S.Nt > 0 ? 3 + (tail_signed ? 1 : 0) : 0); }
    double bytes_per_solve() const override {
        const double s = sizeof(T);
        return 2.0 * (double)S.ci.size() * (s + 4) + (double)S.ldt * S.ldt * s + 8.0 * S.N * s;
    }

    void factorize(double rho, double sigma) override {
        const int N = S.N, Ns = S.Ns, ldt = S.ldt;
        const int64_t nz = (int64_t)S.ci.size();
        HIPC(hipMemsetAsync(fail.p, 0, 4 * sizeof(int), st));
        if (nz > 0) HIPC(hipMemsetAsync(vr.p, 0, sizeof(T) * nz, st));
        if (ldt > 0) HIPC(hipMemsetAsync(Mt.p, 0, sizeof(T) * (size_t)ldt * ldt, st));
        const int64_t ne = (int64_t)S.k_dst.size();
        if (ne > 0) hipLaunchKernelGGL((k_ldl_scatter<T>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, ne, kdst.p, ksrc.p, kval.p, vr.p, Mt.p);
        hipLaunchKernelGGL((k_ldl_diag0<T>), dim3((Ns + ldt + 255) / 256), dim3(256), 0, st, N, Ns, ldt, dposP.p, sign.p, kval.p, (T)sigma, (T)(-1.0 / rho), D0.p, Mt.p);
        const int L = (int)S.level_ptr.size() - 1;
        for (int l = 0; l < L; ++l) {
            const int c0 = S.level_ptr[l], c1 = S.level_ptr[l + 1];
            hipLaunchKernelGGL((k_ldl_level_diag<T>), dim3((unsigned)(((int64_t)(c1 - c0) * GL + 255) / 256)), dim3(256), 0, st, c0, c1, rp.p, ci.p, vr.p, D0.p, D.p, Dinv.p, sign.p, fail.p);
            const int q0 = S.cp[c0], q1 = S.cp[c1];
            if (q1 > q0) hipLaunchKernelGGL((k_ldl_level_offdiag<T>), dim3((unsigned)(((int64_t)(q1 - q0) * GL + 255) / 256)), dim3(256), 0, st, q0, q1, rp.p, ci.p, ri.p, cj.p,
                                            csc2csr.p, D.p, Dinv.p, vr.p, vc.p);
        }
        if (ldt > 0) {
            for (size_t gI = 0; gI + 1 < group_ptr.size(); ++gI) {                  // Mt -= (L_ts D) L_ts', group by group (lower tiles)
                const int e0 = group_ptr[gI], e1 = group_ptr[gI + 1];
                const int64_t Kg = (int64_t)GS * KC;
                HIPC(hipMemsetAsync(Ld.p, 0, sizeof(T) * (size_t)ldt * Kg, st));
                HIPC(hipMemsetAsync(Wd.p, 0, sizeof(T) * (size_t)ldt * Kg, st));
                hipLaunchKernelGGL((k_ldl_densify<T>), dim3((e1 - e0 + 255) / 256), dim3(256), 0, st, e0, e1, ent_q.p, ent_dst.p, cj.p, vc.p, D.p, Ld.p, Wd.p);
                gemm<T>(st, ldt, ldt, KC, T(1), Wd.p, Kg, true, Ld.p, Kg, true, T(0), slabs.p, ldt, true, GS, KC, KC, (int64_t)ldt * ldt);
                hipLaunchKernelGGL((k_ldl_sub_slabs<T>), dim3((ldt + 255) / 256, ldt), dim3(256), 0, st, ldt, GS, slabs.p, Mt.p);
            }
            if (tail_signed) cholesky_signed<T>(st, ldt, Mt.p, dinv.p, fail.p + 1, tsgn.p, tmp.p);
            else cholesky<T>(st, ldt, Mt.p, dinv.p, fail.p + 1, 1, chol_scratch_fits(ldt) ? St.p : nullptr);
            build_sweep_matrix<T>(st, ldt, nb, Mt.p, dinv.p, St.p, tmp.p);
        }
        int f[2] = {0, 0};
        HIPC(hipMemcpyAsync(f, fail.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        if (f[0] != 0 || f[1] != 0) {
            char bf[256];
            snprintf(bf, sizeof bf, "L D L' of the KKT matrix broke down: pivot %d of the %s has the wrong sign or is zero (rho=%g, sigma=%g)",
                     f[0] ? f[0] : f[1], f[0] ? "sparse part" : "dense tail", rho, sigma);
            throw QpsError(QPS_ERR_FACTORIZATION, bf);
        }
    }

    template <int LPR> void fwd_launch(int r0, int r1) {
        hipLaunchKernelGGL((k_ldl_fwd<T, LPR>), dim3((unsigned)(((int64_t)(r1 - r0) * LPR + 255) / 256)), dim3(256), 0, st, r0, r1, S.Ns, rp.p, ci.p, vr.p, b.p, tb.p);
    }
    void fwd(int r0, int r1, int lpr) { if (r1 <= r0) return; if (lpr == 1) fwd_launch<1>(r0, r1); else if (lpr == 4) fwd_launch<4>(r0, r1); else if (lpr == 16) fwd_launch<16>(r0, r1); else if (lpr == 64) fwd_launch<64>(r0, r1); else fwd_launch<256>(r0, r1); }
    template <int LPR> void bwd_launch(int c0, int c1) {
        hipLaunchKernelGGL((k_ldl_bwd<T, LPR>), dim3((unsigned)(((int64_t)(c1 - c0) * LPR + 255) / 256)), dim3(256), 0, st, c0, c1, S.Ns, cp.p, ri.p, vc.p, Dinv.p, b.p, tx.p);
    }
    void bwd(int c0, int c1, int lpr) { if (c1 <= c0) return; if (lpr == 1) bwd_launch<1>(c0, c1); else if (lpr == 4) bwd_launch<4>(c0, c1); else if (lpr == 16) bwd_launch<16>(c0, c1); else if (lpr == 64) bwd_launch<64>(c0, c1); else bwd_launch<256>(c0, c1); }

    void sweeps() {   // b (permuted right-hand side) -> solution in b (sparse part) and tx (tail)
        const int N = S.N, Ns = S.Ns, Nt = S.Nt, ldt = S.ldt;
        const int L = (int)S.level_ptr.size() - 1;
        for (int l = 1; l < L; ++l) fwd(S.level_ptr[l], S.level_ptr[l + 1], lpr_fwd[l]);          // level 0: leaves, nothing to subtract
        if (Nt > 0) {
            fwd(Ns, N, lpr_tail);                                                                   // tb = b_t - L_ts y_s
            gemv_rows<T>(st, St.p, ldt, tb.p, tu.p, nullptr, T(1), T(0), 0, ldt, 0, ldt, 1);        // inv(Lt) tb
            if (tail_signed) hipLaunchKernelGGL((k_ldl_mul_sign<T>), dim3((ldt + 255) / 256), dim3(256), 0, st, ldt, tsgn.p, tu.p);   // J .
            gemv_rows<T>(st, St.p, ldt, tu.p, tx.p, nullptr, T(1), T(0), 0, ldt, 0, ldt, 2);        // inv(Lt)' .
        }
        for (int l = L - 1; l >= 0; --l) bwd(S.level_ptr[l], S.level_ptr[l + 1], lpr_bwd[l]);
    }
    void iterate(T* x, T* xp, const T* q, T* z, T* zp, T* y, const T* l, const T* u, double alpha, double rho, double sigma, bool rhs_ready) override {
        const int N = S.N;
        if (!rhs_ready) hipLaunchKernelGGL((k_ldl_rhs<T>), dim3((N + 255) / 256), dim3(256), 0, st, N, S.n, perm.p, x, q, z, y, (T)sigma, (T)(1.0 / rho), b.p);
        sweeps();
        hipLaunchKernelGGL((k_ldl_post_update<T>), dim3((N + 255) / 256), dim3(256), 0, st, S.n, S.m, S.Ns, iperm.p, b.p, tx.p, q, x, xp, z, zp, y, l, u,
                           (T)alpha, (T)rho, (T)sigma);
    }
    void solve_raw(const T* r1, const T* r2, T* out_x, T* out_nu) override {
        const int N = S.N;
        hipLaunchKernelGGL((k_ldl_rhs_raw<T>), dim3((N + 255) / 256), dim3(256), 0, st, N, S.n, perm.p, r1, r2, b.p);
        sweeps();
        hipLaunchKernelGGL((k_ldl_post_raw<T>), dim3((N + 255) / 256), dim3(256), 0, st, S.n, S.m, S.Ns, iperm.p, b.p, tx.p, out_x, out_nu);
    }
    void solve(const T* x, const T* q, const T* z, const T* y, double rho, double sigma, T* xx, T* zz) override {
        const int N = S.N;
        const T rho1 = (T)(1.0 / rho);
        hipLaunchKernelGGL((k_ldl_rhs<T>), dim3((N + 255) / 256), dim3(256), 0, st, N, S.n, perm.p, x, q, z, y, (T)sigma, rho1, b.p);
        sweeps();
        hipLaunchKernelGGL((k_ldl_post<T>), dim3((N + 255) / 256), dim3(256), 0, st, S.n, S.m, S.Ns, iperm.p, b.p, tx.p, z, y, rho1, xx, zz);
    }
};

}  // namespace

template <typename T> void cholesky_signed(hipStream_t st, int NP, T* M, T* dinv, int* fail_dev, const T* sgn, T* W) {
    const int nblk = NP / 64;
    for (int kb = 0; kb < nblk; ++kb) {
        hipLaunchKernelGGL((k_spotrf64<T>), dim3(1), dim3(256), 0, st, M, (int64_t)NP, kb, sgn, dinv, fail_dev);
        const int rem = NP - (kb + 1) * 64;
        if (rem <= 0) break;
        hipLaunchKernelGGL((k_spanel<T>), dim3(rem / 64), dim3(256), 0, st, M, (int64_t)NP, kb, sgn, dinv, W);
        const T* W21 = W + (int64_t)(kb + 1) * 64 * NP + kb * 64;
        const T* L21 = M + (int64_t)(kb + 1) * 64 * NP + kb * 64;
        T* A22 = M + (int64_t)(kb + 1) * 64 * NP + (kb + 1) * 64;
        gemm<T>(st, rem, rem, 64, T(-1), W21, NP, true, L21, NP, true, T(1), A22, NP, true);
    }
}

template <typename T>
std::unique_ptr<SparseLdl<T>> make_sparse_ldl(hipStream_t st, LdlSymbolic&& sym, const double* Pvals, int64_t pnnz, const double* Avals, int64_t annz) {
    return std::unique_ptr<SparseLdl<T>>(new SparseLdlImpl<T>(st, std::move(sym), Pvals, pnnz, Avals, annz));
}

#define INST(T)                                                                                                            \
    template void cholesky_signed<T>(hipStream_t, int, T*, T*, int*, const T*, T*);                                        \
    template std::unique_ptr<SparseLdl<T>> make_sparse_ldl<T>(hipStream_t, LdlSymbolic&&, const double*, int64_t, const double*, int64_t);
INST(double)
INST(float)
#undef INST

}  // namespace qps
