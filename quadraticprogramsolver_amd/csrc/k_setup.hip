// k_setup.hip -- one-off / per-refactor kernels: layout import, A'A (SYRK) and the blocked Cholesky on the fp64 / fp32
// MFMA pipe (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32), inversion of the diagonal sweep blocks.
// Reference: LinearSystemSolvers.jl:112-114 (mAA, mPI, mL), :127-129 (rebuild on changedRho), ProxQP.jl:175-206
// (dense Cholesky + in-place re-factorisation precedent).
#include "qps_kernels.h"
#include "wave_reduce.h"

namespace qps {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
    using acc_t = d4;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma<float> {
    using acc_t = f4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

constexpr int GT = 64;    // C tile edge per workgroup
constexpr int GK = 16;    // K depth per LDS stage
constexpr int GLD = 80;   // LDS row stride (elements): k-groups of a fragment read land 32 banks apart

// 64 x 64 C tile per 256-thread workgroup; waves in a 2 x 2 grid, each wave 2 x 2 MFMA 16 x 16 tiles.
// tile != nullptr: the finished C tile goes to that LDS array (row stride 65) INSTEAD of global memory
template <typename T, bool AK, bool BK>
__device__ __forceinline__ void gemm_tile(int K, T alpha, const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, T beta,
                                          T* __restrict__ C, int64_t ldc, int ktri, int bi, int bj, T (*As)[GK][GLD], T (*Bs)[GK][GLD],
                                          T (*tile)[65]) {
    // As, Bs: the caller's double-buffered LDS staging arrays [2][GK][GLD] (one barrier per k-step); `tile` may alias them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    using acc_t = typename Mfma<T>::acc_t;
    acc_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
    const int i0 = bi * GT, j0 = bj * GT;
    // ktri 1: opB is lower triangular (opB(k,j) = 0 for k < j) -> start at k = j0;  ktri 2: opA is lower triangular
    // (opA(i,k) = 0 for k > i) -> stop after the diagonal tile of row block i0
    const int kbeg = (ktri == 1) ? j0 : 0;
    const int kend = (ktri == 2) ? min(K, i0 + GT) : K;
    // opA(i0.., k0..) goes to As[k][i], opB(k0.., j0..) to Bs[k][j], 4 elements per thread each; the slab of the NEXT k-step is
    // fetched into registers before the MFMAs of the current one (the global-load latency overlaps the multiply instead of
    // relying on other workgroups of the CU to cover it)
    T ga[4], gb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (AK) { const int k = tid & 15, i = (tid >> 4) + 16 * r; ga[r] = A[(int64_t)(i0 + i) * lda + k0 + k]; }
            else    { const int i = tid & 63, k = (tid >> 6) + 4 * r;  ga[r] = A[(int64_t)(k0 + k) * lda + i0 + i]; }
            if (BK) { const int k = tid & 15, j = (tid >> 4) + 16 * r; gb[r] = B[(int64_t)(j0 + j) * ldb + k0 + k]; }
            else    { const int j = tid & 63, k = (tid >> 6) + 4 * r;  gb[r] = B[(int64_t)(k0 + k) * ldb + j0 + j]; }
        }
    };
    if (kbeg < kend) gload(kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GK, buf ^= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (AK) { const int k = tid & 15, i = (tid >> 4) + 16 * r; As[buf][k][i] = ga[r]; }
            else    { const int i = tid & 63, k = (tid >> 6) + 4 * r;  As[buf][k][i] = ga[r]; }
            if (BK) { const int k = tid & 15, j = (tid >> 4) + 16 * r; Bs[buf][k][j] = gb[r]; }
            else    { const int j = tid & 63, k = (tid >> 6) + 4 * r;  Bs[buf][k][j] = gb[r]; }
        }
        __syncthreads();   // also orders this step's reads of `buf` after the stores above, and the stores of step k+2 into `buf` after them
        if (k0 + GK < kend) gload(k0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            const int kr = kk + (lane >> 4), cl = lane & 15;
            const T a0 = As[buf][kr][wm * 32 + cl], a1 = As[buf][kr][wm * 32 + 16 + cl];
            const T b0 = Bs[buf][kr][wn * 32 + cl], b1 = Bs[buf][kr][wn * 32 + 16 + cl];
            acc[0][0] = Mfma<T>::run(a0, b0, acc[0][0]);
            acc[0][1] = Mfma<T>::run(a0, b1, acc[0][1]);
            acc[1][0] = Mfma<T>::run(a1, b0, acc[1][0]);
            acc[1][1] = Mfma<T>::run(a1, b1, acc[1][1]);
        }
    }
    if (tile) __syncthreads();   // the tile may live in the staging buffers: every wave must be done reading them
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm * 32 + a * 16 + Mfma<T>::row(lane, r);
                const int col = j0 + wn * 32 + b * 16 + (lane & 15);
                T* cp = C + (int64_t)row * ldc + col;
                T v = alpha * acc[a][b][r];
                if (beta != T(0)) v += beta * (*cp);
                if (tile) tile[row - i0][col - j0] = v;
                else *cp = v;
            }
}
template <typename T, bool AK, bool BK>
__global__ __launch_bounds__(256) void k_gemm(int K, T alpha, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                              int64_t ldb, T beta, T* __restrict__ C, int64_t ldc, int lower_only,
                                              int64_t sA, int64_t sB, int64_t sC, int ktri) {
    const int bj = blockIdx.x, bi = blockIdx.y;
    if (lower_only && bj > bi) return;
    A += (int64_t)blockIdx.z * sA; B += (int64_t)blockIdx.z * sB; C += (int64_t)blockIdx.z * sC;
    __shared__ T As[2][GK][GLD];
    __shared__ T Bs[2][GK][GLD];
    gemm_tile<T, AK, BK>(K, alpha, A, lda, B, ldb, beta, C, ldc, ktri, bi, bj, As, Bs, nullptr);
}

// column-major double src (rows x cols) -> row-major T dst (ld ldd), via a 64 x 64 LDS transpose tile; dst padding is
// left untouched (the caller zero-fills the allocation first).
template <typename T>
__global__ __launch_bounds__(256) void k_import_colmajor(const double* __restrict__ src, int64_t lds, int rows, int cols,
                                                         T* __restrict__ dst, int64_t ldd) {
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {   // lanes along rows: contiguous in column-major src
        const int r = r0 + tx, c = c0 + cc;
        tile[cc][tx] = (r < rows && c < cols) ? (T)src[(int64_t)c * lds + r] : T(0);
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {   // lanes along columns: contiguous in row-major dst
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows && c < cols) dst[(int64_t)r * ldd + c] = tile[tx][rr];
    }
}

template <typename T>
__global__ void k_make_PI(int n, int NP, const T* __restrict__ P, T sigma, T* __restrict__ PI) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= NP) return;
    P += (int64_t)blockIdx.z * NP * NP; PI += (int64_t)blockIdx.z * NP * NP;
    T v = T(0);
    if (i < n && j < n) v = P[(int64_t)i * NP + j] + (i == j ? sigma : T(0));   // LinearSystemSolvers.jl:113
    else if (i == j) v = T(1);
    PI[(int64_t)i * NP + j] = v;
}
template <typename T>
__global__ void k_assemble_M(int NP, const T* __restrict__ PI, const T* __restrict__ AA, T rho, T* __restrict__ M,
                             const double* __restrict__ rho_arr) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= NP || j > (i | 63)) return;                                          // lower tiles incl. whole diagonal tile
    if (rho_arr) rho = (T)rho_arr[blockIdx.z];                                    // batched: every QP has its own rho
    const int64_t o = (int64_t)blockIdx.z * NP * NP + (int64_t)i * NP + j;
    // AA holds the lower tiles of A'A; inside a diagonal tile both halves are present
    M[o] = PI[o] + rho * AA[o];                                                   // LinearSystemSolvers.jl:114 / :128
}

// Cholesky of one 64 x 64 diagonal block (factor only).  256 threads: lane i = row, wave g = column group of 16, so
// thread (i, g) keeps the 16 entries a[i][16g .. 16g+15] in registers.  The 16 columns of a panel live in ONE wave, so
// the panel is factorised with wave64 shuffles only (no LDS, no barrier); the finished panel is then published through
// LDS and the waves to its right apply the rank-16 update.  4 panels => 8 barriers per block instead of 128.
// Broadcast of lane `L` (a compile-time constant once the panel loops are unrolled) through v_readlane: a few cycles, where
// __shfl's ds_bpermute costs an LDS round trip -- the factorisation of a diagonal block is one long dependent chain of them.
template <int L> __device__ __forceinline__ double lane_bcast(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), L);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int L> __device__ __forceinline__ float lane_bcast(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); }
// 1/sqrt(d) to working precision: hardware estimate + two Newton steps (no division, no sqrt sequence on the critical path)
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}
__device__ __forceinline__ float rsqrt_nr(float d) {
    float y = __builtin_amdgcn_rsqf(d);
    y = y * (1.5f - 0.5f * d * y * y);
    return y;
}

template <typename T, int P, int K> struct PotrfCol {
    static __device__ __forceinline__ void run(T (&a)[16], int i, int kb, int* fail) {
        T d = lane_bcast<16 * P + K>(a[K]);
        if (!(d > T(0))) { if (i == 0) atomicCAS(fail, 0, kb * 64 + 16 * P + K + 1); d = T(1); }
        const T rs = rsqrt_nr(d);
        a[K] *= rs;                                          // l_ik for i >= k (row k itself: d * rs = sqrt(d))
        PotrfCol<T, P, K>::template update<K + 1>(a);
        PotrfCol<T, P, K + 1>::run(a, i, kb, fail);
    }
    template <int J> static __device__ __forceinline__ void update(T (&a)[16]) {
        if constexpr (J < 16) {
            const T ljk = lane_bcast<16 * P + J>(a[K]);
            a[J] -= a[K] * ljk;
            update<J + 1>(a);
        }
    }
};
template <typename T, int P> struct PotrfCol<T, P, 16> { static __device__ __forceinline__ void run(T (&)[16], int, int, int*) {} };

template <typename T, int P>
__device__ __forceinline__ void potrf_panel(T (&a)[16], T (*Lp)[17], int i, int g, int kb, int* fail) {
    if (g == P) {                                            // wave-uniform
        PotrfCol<T, P, 0>::run(a, i, kb, fail);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lp[i][k] = a[k];
    }
    __syncthreads();
    if (g > P) {
        T li[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) li[k] = Lp[i][k];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) s += li[k] * Lp[16 * g + jj][k];   // same address for all lanes: broadcast
            a[jj] -= s;
        }
    }
    __syncthreads();
}

// src: the block to factorise, either in place in global memory (src == nullptr) or in an LDS tile (row stride 65)
template <typename T>
__device__ __forceinline__ void potrf_block(T* __restrict__ blk, int64_t ld, int kb, int* __restrict__ fail, T (*src)[65]) {
    __shared__ T Lp[64][17];
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    T a[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) a[jj] = src ? src[i][16 * g + jj] : blk[(int64_t)i * ld + 16 * g + jj];
    potrf_panel<T, 0>(a, Lp, i, g, kb, fail);
    potrf_panel<T, 1>(a, Lp, i, g, kb, fail);
    potrf_panel<T, 2>(a, Lp, i, g, kb, fail);
    potrf_panel<T, 3>(a, Lp, i, g, kb, fail);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = 16 * g + jj;
        blk[(int64_t)i * ld + j] = (j <= i) ? a[jj] : T(0);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_potrf64(T* __restrict__ M, int64_t ld, int kb, int* __restrict__ fail, int64_t sM) {
    M += (int64_t)blockIdx.x * sM; fail += blockIdx.x;                            // batched: one workgroup per QP
    potrf_block<T>(M + (int64_t)kb * 64 * ld + kb * 64, ld, kb, fail, nullptr);
}
// Trailing update of step kb (A22 -= L21 L21', lower tiles) with the factorisation of the NEXT diagonal block fused in: the
// workgroup that owns tile (0, 0) keeps its finished tile in LDS and factorises it while the other tiles are still being
// updated -- one launch less on the chain of dependent launches per 64-column step.
template <typename T>
__global__ __launch_bounds__(256) void k_update_potrf(T* __restrict__ M, int64_t ld, int kb, int* __restrict__ fail, int64_t sM) {
    const int bj = blockIdx.x, bi = blockIdx.y;
    if (bj > bi) return;
    M += (int64_t)blockIdx.z * sM; fail += blockIdx.z;
    __shared__ T stage[2][2][GK][GLD];                                            // As | Bs; reused as the 64 x 65 tile of the (0, 0) workgroup
    static_assert(sizeof(T) * 64 * 65 <= sizeof(T) * 2 * 2 * GK * GLD, "tile must fit into the staging buffers");
    T (*tile)[65] = reinterpret_cast<T (*)[65]>(&stage[0][0][0][0]);
    const T* A21 = M + (int64_t)(kb + 1) * 64 * ld + kb * 64;
    T* A22 = M + (int64_t)(kb + 1) * 64 * ld + (kb + 1) * 64;
    const bool diag0 = (bi == 0 && bj == 0);                                      // workgroup-uniform
    gemm_tile<T, true, true>(64, T(-1), A21, ld, A21, ld, T(1), A22, ld, 0, bi, bj, stage[0], stage[1], diag0 ? tile : nullptr);
    if (diag0) {
        __syncthreads();
        potrf_block<T>(A22, ld, kb + 1, fail, tile);
    }
}

// Inverses of ALL 64 x 64 diagonal blocks of L in one launch (blockIdx.x = block): forward elimination on the identity,
// row i accumulates -sum_{p<i} L[i][p] X[p][:] and is scaled by 1/L[i][i] when k reaches i.  Off the Cholesky critical path.
template <typename T>
__global__ __launch_bounds__(256) void k_inv64(const T* __restrict__ L, int64_t ld, T* __restrict__ dinv, int64_t sM, int64_t sD) {
    __shared__ T Ls[64][65];
    __shared__ T rowk[64];
    const int kb = blockIdx.x;
    L += (int64_t)blockIdx.y * sM; dinv += (int64_t)blockIdx.y * sD;
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    const T* blk = L + (int64_t)kb * 64 * ld + kb * 64;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Ls[i][16 * g + jj] = blk[(int64_t)i * ld + 16 * g + jj];
    __syncthreads();
    T x[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) x[jj] = (i == 16 * g + jj) ? T(1) : T(0);
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        if (i == k) {
            const T inv = T(1) / Ls[k][k];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) { x[jj] *= inv; rowk[16 * g + jj] = x[jj]; }
        }
        __syncthreads();
        if (i > k) {
            const T lik = Ls[i][k];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) x[jj] -= lik * rowk[16 * g + jj];
        }
        __syncthreads();
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) dinv[(int64_t)kb * 4096 + i * 64 + 16 * g + jj] = x[jj];
}

template <typename T, int J> struct TrsmStep {
    // one column of the substitution for a row owned by 4 lanes (lane part p holds x_k for k = p mod 4): partial sums over the
    // lane's k < J, two DPP-class shuffles, scale; everything indexed at compile time
    static __device__ __forceinline__ void run(T (&xs)[16], const T (*Ls)[65], const T* rdiag, T (*Xs)[65], int r, int p) {
        T s = (p == (J & 3)) ? Xs[r][J] : T(0);
#pragma unroll
        for (int i = 0; i < (J + 3) / 4; ++i) {
            const int k = 4 * i + p;
            if (k < J) s -= xs[i] * Ls[J][k];
        }
        s = quad_sum_all(s);                                 // DPP quad_perm: no LDS round trip inside the 64-step chain
        const T xj = s * rdiag[J];
        if (p == (J & 3)) { xs[J >> 2] = xj; Xs[r][J] = xj; }
        TrsmStep<T, J + 1>::run(xs, Ls, rdiag, Xs, r, p);
    }
};
template <typename T> struct TrsmStep<T, 64> { static __device__ __forceinline__ void run(T (&)[16], const T (*)[65], const T*, T (*)[65], int, int) {} };

// Panel solve X * L11' = A21 (in place) for one 64 x 64 tile of A21 per workgroup.  256 threads: the two tiles travel through
// LDS with lanes along a row (one 512-B segment per load instruction, four row groups in parallel); then 4 lanes share a row
// of the substitution  x_j = (a_j - sum_{k<j} x_k L11[j][k]) / L11[j][j].
template <typename T>
__global__ __launch_bounds__(256) void k_trsm_panel(T* __restrict__ M, int64_t ld, int kb, int nrows, int64_t sM) {
    __shared__ T Ls[64][65];
    __shared__ T Xs[64][65];
    __shared__ T rdiag[64];
    const int t = threadIdx.x, c = t & 63, g = t >> 6;
    M += (int64_t)blockIdx.y * sM;
    const T* L11 = M + (int64_t)kb * 64 * ld + kb * 64;
    T* a0 = M + (int64_t)((kb + 1) * 64 + blockIdx.x * 64) * ld + kb * 64;      // this workgroup's tile of A21 (nrows is a multiple of 64)
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int r = g + 4 * i; Ls[r][c] = L11[(int64_t)r * ld + c]; Xs[r][c] = a0[(int64_t)r * ld + c]; }
    __syncthreads();
    if (t < 64) rdiag[t] = T(1) / Ls[t][t];
    __syncthreads();
    {
        const int r = t >> 2, p = t & 3;
        T xs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[i] = T(0);
        TrsmStep<T, 0>::run(xs, Ls, rdiag, Xs, r, p);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int r = g + 4 * i; a0[(int64_t)r * ld + c] = Xs[r][c]; }
    (void)nrows;
}

// S lower <- L lower with the 64-blocks on the diagonal replaced by their inverses; S upper <- 0
template <typename T>
__global__ void k_init_sweep(int NP, const T* __restrict__ L, const T* __restrict__ dinv, T* __restrict__ S, int64_t sD) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= NP) return;
    L += (int64_t)blockIdx.z * NP * NP; S += (int64_t)blockIdx.z * NP * NP; dinv += (int64_t)blockIdx.z * sD;
    T v = T(0);
    if ((i >> 6) == (j >> 6)) v = dinv[(int64_t)(i >> 6) * 4096 + (i & 63) * 64 + (j & 63)];
    else if (j < i) v = L[(int64_t)i * NP + j];
    S[(int64_t)i * NP + j] = v;
}
// S[j][i] = S[i][j] for j < i, 64 x 64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void k_mirror(int NP, T* __restrict__ S) {
    const int bj = blockIdx.x, bi = blockIdx.y;
    if (bj > bi) return;
    S += (int64_t)blockIdx.z * NP * NP;
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = S[(int64_t)(bi * 64 + r) * NP + bj * 64 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        // destination element (row = bj*64 + r, col = bi*64 + tx) = source (bi*64 + tx, bj*64 + r)
        if (bi != bj || tx > r) S[(int64_t)(bj * 64 + r) * NP + bi * 64 + tx] = tile[tx][r];
    }
}

}  // namespace

template <typename T>
void import_colmajor(hipStream_t st, const double* src, int64_t lds, int rows, int cols, T* dst, int64_t ldd) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL((k_import_colmajor<T>), dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, lds, rows, cols, dst, ldd);
}

template <typename T>
void gemm(hipStream_t st, int M, int N, int K, T alpha, const T* A, int64_t lda, bool ak, const T* B, int64_t ldb, bool bk,
          T beta, T* C, int64_t ldc, bool lower_only, int batch, int64_t sA, int64_t sB, int64_t sC, int ktri) {
    if (M <= 0 || N <= 0 || batch <= 0) return;
    dim3 grid(N / GT, M / GT, batch), block(256);
    const int lo = lower_only ? 1 : 0;
    if (ak && bk) hipLaunchKernelGGL((k_gemm<T, true, true>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri);
    else if (ak && !bk) hipLaunchKernelGGL((k_gemm<T, true, false>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri);
    else if (!ak && bk) hipLaunchKernelGGL((k_gemm<T, false, true>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri);
    else hipLaunchKernelGGL((k_gemm<T, false, false>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri);
}

template <typename T> void make_PI(hipStream_t st, int n, int NP, const T* P, T sigma, T* PI, int batch) {
    hipLaunchKernelGGL((k_make_PI<T>), dim3((NP + 255) / 256, NP, batch), dim3(256), 0, st, n, NP, P, sigma, PI);
}
template <typename T> void assemble_M(hipStream_t st, int NP, const T* PI, const T* AA, T rho, T* M, int batch, const double* rho_arr) {
    hipLaunchKernelGGL((k_assemble_M<T>), dim3((NP + 255) / 256, NP, batch), dim3(256), 0, st, NP, PI, AA, rho, M, rho_arr);
}

// batch > 1: the same factorisation for `batch` matrices NP*NP apart (dinv blocks (NP/64)*4096 apart, fail flags 1 apart):
// every launch carries all QPs, so the launch-latency-bound panel chain is paid once instead of `batch` times.
template <typename T> void cholesky(hipStream_t st, int NP, T* M, T* dinv, int* fail_dev, int batch) {
    (void)hipMemsetAsync(fail_dev, 0, sizeof(int) * batch, st);
    const int nblk = NP / 64;
    const int64_t sM = (int64_t)NP * NP, sD = (int64_t)nblk * 4096;
    hipLaunchKernelGGL((k_potrf64<T>), dim3(batch), dim3(256), 0, st, M, (int64_t)NP, 0, fail_dev, sM);
    for (int kb = 0; kb < nblk; ++kb) {
        const int rem = NP - (kb + 1) * 64;
        if (rem <= 0) break;
        // L21 = A21 * inv(L11)'  by forward substitution, one thread per row
        hipLaunchKernelGGL((k_trsm_panel<T>), dim3(rem / 64, batch), dim3(256), 0, st, M, (int64_t)NP, kb, rem, sM);
        // A22 -= L21 * L21'  (lower tiles only) + factorisation of diagonal block kb + 1
        hipLaunchKernelGGL((k_update_potrf<T>), dim3(rem / 64, rem / 64, batch), dim3(256), 0, st, M, (int64_t)NP, kb, fail_dev, sM);
    }
    hipLaunchKernelGGL((k_inv64<T>), dim3(nblk, batch), dim3(256), 0, st, M, (int64_t)NP, dinv, sM, sD);
}

// batch > 1: L, S and tmp hold `batch` matrices NP*NP apart.  The doubling GEMMs are batched over the QPs (blockIdx.z) and
// looped over the pairs of a level on the host (pairs x QPs would need two batch strides).
template <typename T> void build_sweep_matrix(hipStream_t st, int NP, int nb, const T* L, const T* dinv, T* S, T* tmp, int batch) {
    const int64_t sM = (int64_t)NP * NP, sD = (int64_t)(NP / 64) * 4096;
    hipLaunchKernelGGL((k_init_sweep<T>), dim3((NP + 255) / 256, NP, batch), dim3(256), 0, st, NP, L, dinv, S, sD);
    // recursive doubling: inv([L00 0; L10 L11]) = [inv00 0; -inv11 L10 inv00, inv11]
    for (int s = 64; s < nb; s *= 2) {
        const int64_t pstride = (int64_t)2 * s * (NP + 1);
        int nfull = 0;
        while ((nfull + 1) * 2 * s <= NP) ++nfull;
        if (nfull > 0 && batch == 1) {
            // tmp10 = L10 * inv00 ; S10 = -inv11 * tmp10     (batched over the pairs)
            gemm<T>(st, s, s, s, T(1), L + (int64_t)s * NP, NP, true, S, NP, false, T(0), tmp + (int64_t)s * NP, NP, false, nfull, pstride, pstride, pstride, 1);
            gemm<T>(st, s, s, s, T(-1), S + (int64_t)s * (NP + 1), NP, true, tmp + (int64_t)s * NP, NP, false, T(0), S + (int64_t)s * NP, NP, false, nfull, pstride, pstride, pstride, 2);
        } else {
            for (int p = 0; p < nfull; ++p) {   // batched over the QPs, one pair per launch
                const int64_t o = (int64_t)p * pstride;
                gemm<T>(st, s, s, s, T(1), L + o + (int64_t)s * NP, NP, true, S + o, NP, false, T(0), tmp + o + (int64_t)s * NP, NP, false, batch, sM, sM, sM, 1);
                gemm<T>(st, s, s, s, T(-1), S + o + (int64_t)s * (NP + 1), NP, true, tmp + o + (int64_t)s * NP, NP, false, T(0), S + o + (int64_t)s * NP, NP, false, batch, sM, sM, sM, 2);
            }
        }
        const int o = nfull * 2 * s;
        const int s2 = NP - o - s;
        if (s2 > 0) {   // ragged last pair: second block has s2 < s rows
            const T* L10 = L + (int64_t)(o + s) * NP + o;
            T* t10 = tmp + (int64_t)(o + s) * NP + o;
            gemm<T>(st, s2, s, s, T(1), L10, NP, true, S + (int64_t)o * (NP + 1), NP, false, T(0), t10, NP, false, batch, sM, sM, sM, 1);
            gemm<T>(st, s2, s, s2, T(-1), S + (int64_t)(o + s) * (NP + 1), NP, true, t10, NP, false, T(0), S + (int64_t)(o + s) * NP + o, NP, false, batch, sM, sM, sM, 2);
        }
    }
    hipLaunchKernelGGL((k_mirror<T>), dim3(NP / 64, NP / 64, batch), dim3(256), 0, st, NP, S);
}

#define INST(T)                                                                                                        \
    template void import_colmajor<T>(hipStream_t, const double*, int64_t, int, int, T*, int64_t);                      \
    template void gemm<T>(hipStream_t, int, int, int, T, const T*, int64_t, bool, const T*, int64_t, bool, T, T*, int64_t, \
                          bool, int, int64_t, int64_t, int64_t, int);                                                  \
    template void make_PI<T>(hipStream_t, int, int, const T*, T, T*, int);                                             \
    template void assemble_M<T>(hipStream_t, int, const T*, const T*, T, T*, int, const double*);                      \
    template void cholesky<T>(hipStream_t, int, T*, T*, int*, int);                                                    \
    template void build_sweep_matrix<T>(hipStream_t, int, int, const T*, const T*, T*, T*, int);
INST(double)
INST(float)
#undef INST

}  // namespace qps
