// k_setup.hip -- one-off / per-refactor kernels: layout import, A'A (SYRK) and the blocked Cholesky on the fp64 / fp32
// MFMA pipe (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32), inversion of the diagonal sweep blocks.
// Reference: LinearSystemSolvers.jl:112-114 (mAA, mPI, mL), :127-129 (rebuild on changedRho), ProxQP.jl:175-206
// (dense Cholesky + in-place re-factorisation precedent).
#include "tile_order.h"
#include "qps_kernels.h"
#include "wave_reduce.h"
#include <algorithm>
#include <type_traits>

namespace qps {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
template <typename T> struct Vec16;                    // 16 bytes of T as a native vector
template <> struct Vec16<double> { using type = d2; };
template <> struct Vec16<float> { using type = f4; };

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
// k-steps of operand slabs in flight per workgroup: two for fp32 (a few per cent on every product); the fp64 kernel loses a wave of
// occupancy to the extra registers and runs slower with more than one (tests/tools/micro/chol_chain.hip)
template <typename T> struct GemmDepth { static constexpr int v = sizeof(T) == 4 ? 2 : 1; };
constexpr int GLD = 80;   // LDS row stride (elements): k-groups of a fragment read land 32 banks apart
constexpr int LPS = 20;   // row stride of the published 64 x 16 Cholesky panel: rows 16-byte aligned (b128 LDS reads in the rank-16 update)

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
    const int kbeg = (ktri == 1) ? j0 : (ktri == 3 ? i0 : 0);   // ktri 3: opA is upper triangular (opA(i,k) = 0 for k < i)
    const int kend = (ktri == 2) ? min(K, i0 + GT) : K;
    // opA(i0.., k0..) goes to As[k][i], opB(k0.., j0..) to Bs[k][j], 4 elements per thread each.  The slabs of the next GemmDepth k-steps are
    // in flight (in registers) while the current one is multiplied.
    T ga[GemmDepth<T>::v][4], gb[GemmDepth<T>::v][4];
    // Addresses: everything that changes from slab to slab (k0, the row group r) is workgroup-uniform and stays in scalar registers; a thread adds only its own
    // constant 32-bit offset.  (With the whole index formed per thread, the slabs of an operand that is NOT k-contiguous cost a 64-bit multiply per load: 97 vector
    // instructions beside 32 MFMAs in the loop of the A'A product -- measured 41 % MFMA-pipe busy, 71 % of the wave cycles in issue stalls:
    // profiles/r04_n_gemm_f32_counters_before.txt.)
    const int toA = AK ? (tid >> 4) * (int)lda + (tid & 15) : (tid >> 6) * (int)lda + (tid & 63);
    const int toB = BK ? (tid >> 4) * (int)ldb + (tid & 15) : (tid >> 6) * (int)ldb + (tid & 63);
    const T* const uA = AK ? A + (int64_t)i0 * lda : A + i0;                      // (uniform)
    const T* const uB = BK ? B + (int64_t)j0 * ldb : B + j0;
    auto gload = [&](T (&xa)[4], T (&xb)[4], int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T* sa = AK ? uA + (int64_t)(16 * r) * lda + k0 : uA + (int64_t)(k0 + 4 * r) * lda;     // (uniform: scalar arithmetic)
            const T* sb = BK ? uB + (int64_t)(16 * r) * ldb + k0 : uB + (int64_t)(k0 + 4 * r) * ldb;
            xa[r] = sa[toA];
            xb[r] = sb[toB];
        }
    };
    // the C tile of an accumulating product is fetched first, not after the last multiply (older than every slab load: the conditional
    // does not disturb the counting below)
    T cv[2][2][4];
    const bool have_c = beta != T(0);                                              // workgroup-uniform
    if (have_c) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cv[a][b][r] = C[(int64_t)(i0 + wm * 32 + a * 16 + Mfma<T>::row(lane, r)) * ldc + j0 + wn * 32 + b * 16 + (lane & 15)];
    }
    // (every slab load is unconditional -- past the end the last slab is fetched again and never used -- so that the compiler's count of
    // outstanding loads stays exact across the steps; a conditional load anywhere makes it wait for all of them at every step)
    const int klast = max(kend - GK, 0);
#pragma unroll
    for (int s = 0; s < GemmDepth<T>::v; ++s) gload(ga[s], gb[s], min(kbeg + s * GK, klast));
    int buf = 0;
    // LDS layout of a slab: an operand that is NOT k-contiguous in memory keeps [k][i] (row stride GLD = 80: stores of 64 consecutive i and the fragment reads of
    // 16 i x 4 k are both conflict free).  A k-contiguous operand is stored the way it is loaded, [i][k] with row stride SK = 16 + 4 words (fp64: + 2 double words):
    // its stores (16 consecutive k of four rows per wave) and its fragment reads (address (row cl) SK + k, cl SK mod 64 = sixteen distinct multiples of 4) are conflict
    // free too -- in the [k][i] layout the stores of such an operand hit four banks sixteen lanes deep (counters: 54-70 % of the LDS cycles were bank conflicts).
    constexpr int SK = GK + (sizeof(T) == 4 ? 4 : 2);
    static_assert(64 * SK <= GK * GLD, "the transposed slab fits the staging buffer");
    auto step = [&](T (&xa)[4], T (&xb)[4], int kc) {
        T* const At_ = &As[buf][0][0];
        T* const Bt_ = &Bs[buf][0][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (AK) { const int k = tid & 15, i = (tid >> 4) + 16 * r; At_[i * SK + k] = xa[r]; }
            else    { const int i = tid & 63, k = (tid >> 6) + 4 * r;  As[buf][k][i] = xa[r]; }
            if (BK) { const int k = tid & 15, j = (tid >> 4) + 16 * r; Bt_[j * SK + k] = xb[r]; }
            else    { const int j = tid & 63, k = (tid >> 6) + 4 * r;  Bs[buf][k][j] = xb[r]; }
        }
        __syncthreads();   // also orders this step's reads of `buf` after the stores above, and the stores of step k+2 into `buf` after them
        gload(xa, xb, min(kc + GemmDepth<T>::v * GK, klast));
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            const int kr = kk + (lane >> 4), cl = lane & 15;
            const T a0 = AK ? At_[(wm * 32 + cl) * SK + kr] : As[buf][kr][wm * 32 + cl], a1 = AK ? At_[(wm * 32 + 16 + cl) * SK + kr] : As[buf][kr][wm * 32 + 16 + cl];
            const T b0 = BK ? Bt_[(wn * 32 + cl) * SK + kr] : Bs[buf][kr][wn * 32 + cl], b1 = BK ? Bt_[(wn * 32 + 16 + cl) * SK + kr] : Bs[buf][kr][wn * 32 + 16 + cl];
            acc[0][0] = Mfma<T>::run(a0, b0, acc[0][0]);
            acc[0][1] = Mfma<T>::run(a0, b1, acc[0][1]);
            acc[1][0] = Mfma<T>::run(a1, b0, acc[1][0]);
            acc[1][1] = Mfma<T>::run(a1, b1, acc[1][1]);
        }
        buf ^= 1;
    };
    int k0 = kbeg;
    for (; k0 + GemmDepth<T>::v * GK <= kend; k0 += GemmDepth<T>::v * GK) {
#pragma unroll
        for (int s = 0; s < GemmDepth<T>::v; ++s) step(ga[s], gb[s], k0 + s * GK);
    }
    // remainder (depth not a multiple of GemmDepth<T>::v * 16): the registers of slot s hold step k0 + s * GK
#pragma unroll
    for (int s = 0; s < GemmDepth<T>::v - 1; ++s)
        if (k0 + s * GK < kend) step(ga[s], gb[s], k0 + s * GK);
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
                if (have_c) v += beta * cv[a][b][r];
                if (tile) tile[row - i0][col - j0] = v;
                else *cp = v;
            }
}
// lower_only >= 2 (= tiles per side): the lower tiles on a 1-D grid in the XCD-aware order of lower_tile_of (tile_order.h)
template <typename T, bool AK, bool BK>
__global__ __launch_bounds__(256) void k_gemm(int K, T alpha, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                              int64_t ldb, T beta, T* __restrict__ C, int64_t ldc, int lower_only,
                                              int64_t sA, int64_t sB, int64_t sC, int ktri, int pair) {
    int bj = blockIdx.x, bi = blockIdx.y;
    if (lower_only >= 2) { if (!lower_tile_of((int)blockIdx.x, (int)gridDim.x, lower_only, bi, bj)) return; }
    else if (lower_only && bj > bi) return;
    A += (int64_t)blockIdx.z * sA; B += (int64_t)blockIdx.z * sB; C += (int64_t)blockIdx.z * sC;
    __shared__ T As[2][GK][GLD];
    __shared__ T Bs[2][GK][GLD];
    gemm_tile<T, AK, BK>(K, alpha, A, lda, B, ldb, beta, C, ldc, ktri, bi, bj, As, Bs, nullptr);
    if constexpr (!(AK && BK)) if (pair) {   // (no triangular-operand product of this library has both operands k-contiguous: that instantiation keeps its registers)
        // A triangular operand makes the depth of a tile grow along one axis of the tile grid (ktri 1: from K - 64 j down to 64; ktri 2 / 3: along i).  With one tile
        // per workgroup and the whole grid resident at once, the launch lasted as long as its deepest tiles -- the 2048^3 products of the inverse doubling took as long
        // as the FULL product although half their depth is skipped.  Here a workgroup takes a tile and its mirror image along that axis: every workgroup has the
        // same depth in total.
        __syncthreads();                                                           // the staging buffers are reused
        if (ktri == 1) gemm_tile<T, AK, BK>(K, alpha, A, lda, B, ldb, beta, C, ldc, ktri, bi, 2 * (int)gridDim.x - 1 - bj, As, Bs, nullptr);
        else gemm_tile<T, AK, BK>(K, alpha, A, lda, B, ldb, beta, C, ldc, ktri, 2 * (int)gridDim.y - 1 - bi, bj, As, Bs, nullptr);
    }
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
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4, i = blockIdx.y;           // four consecutive columns per thread (NP is a multiple of 64)
    if (j >= NP || j > (i | 63)) return;                                          // lower tiles incl. whole diagonal tile
    if (rho_arr) rho = (T)rho_arr[blockIdx.z];                                    // batched: every QP has its own rho
    const int64_t o = (int64_t)blockIdx.z * NP * NP + (int64_t)i * NP + j;
    // AA holds the lower tiles of A'A; inside a diagonal tile both halves are present
    typedef T v4 __attribute__((ext_vector_type(4)));
    const v4 p = *reinterpret_cast<const v4*>(PI + o), a = *reinterpret_cast<const v4*>(AA + o);
    *reinterpret_cast<v4*>(M + o) = p + rho * a;                                   // LinearSystemSolvers.jl:114 / :128
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

// phase clock of the step kernel for tests/tools/micro/chol_chain.hip (compiled out of the library)
#ifdef QPS_CHOL_TIMING
__device__ long long g_chol_clock[16];
__device__ long long g_potrf_clock[16];
#define CHOL_T(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_chol_clock[k] = wall_clock64(); } while (0)
#define POTRF_T(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_potrf_clock[k] = wall_clock64(); } while (0)
#else
#define CHOL_T(k) do { } while (0)
#define POTRF_T(k) do { } while (0)
#endif
// 1/d to working precision (hardware estimate + Newton steps on the FMA pipe)
__device__ __forceinline__ double recip_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float recip_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}
// One column of a 16-column panel, lane = row.  The 64 columns of a diagonal block are ONE chain of dependent instructions, so what counts is the length of the
// chain from one pivot to the next: pivot d (readlane) -> 1/d -> t = u_i / d -> a[K+1] -= t u_{K+1} -> next pivot.  The rank-1 update works on the UNSCALED
// column u (its broadcasts do not wait for the pivot's reciprocal) and the scaling l_ik = u_i / sqrt(d) is computed beside the chain, not on it (the scaled
// column is only needed when the panel is published).  A pivot that is not positive is recorded without a branch (`bad`: first offending column + 1; the
// columns after it are then garbage, and the caller reports the failure).
template <typename T, int P, int K> struct PotrfCol {
    static __device__ __forceinline__ void run(T (&a)[16], int col0, int& bad) {
        const T d = lane_bcast<16 * P + K>(a[K]);
        bad = (bad == 0 && !(d > T(0))) ? col0 + 16 * P + K + 1 : bad;
        const T t = a[K] * recip_nr(d);
        PotrfCol<T, P, K>::template update<K + 1>(a, t);
        a[K] *= rsqrt_nr(d);                                 // l_ik for i >= k (row k itself: d / sqrt(d) = sqrt(d))
        PotrfCol<T, P, K + 1>::run(a, col0, bad);
    }
    template <int J> static __device__ __forceinline__ void update(T (&a)[16], T t) {
        if constexpr (J < 16) {
            const T ujk = lane_bcast<16 * P + J>(a[K]);
            a[J] -= t * ujk;
            update<J + 1>(a, t);
        }
    }
};
template <typename T, int P> struct PotrfCol<T, P, 16> { static __device__ __forceinline__ void run(T (&)[16], int, int&) {} };

template <typename T, int P>
__device__ __forceinline__ void potrf_panel(T (&a)[16], T (*Lp)[LPS], int i, int g, int col0, int* fail) {
    if (g == P) {                                            // wave-uniform
        POTRF_T(3 * P);
        int bad = 0;
        PotrfCol<T, P, 0>::run(a, col0, bad);
        if (bad && i == 0 && fail) atomicCAS(fail, 0, bad);
        POTRF_T(3 * P + 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lp[i][k] = a[k];
    }
    __syncthreads();
    if (g == 3) POTRF_T(3 * P + 2);
    if (g > P) {
        T li[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) li[k] = Lp[i][k];
        if constexpr (sizeof(T) == 4) {
            // v_mfma_f32_4x4x1 (16 blocks of 4 x 4, one per 4 lanes): lane 4b + n receives D_b[m][n] = sum_k A_b[m][k] B_b[k][n] in register m,
            // A_b[m][k] supplied by lane 4b + m, B_b[k][n] by lane 4b + n.  With B = this lane's own panel row and A = panel row 16g + 4q + m,
            // lane i ends up with the corrections of its own a[4q .. 4q+3]: the lane = row layout of the factorisation is kept.
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T* ar = &Lp[16 * g + 4 * q + (i & 3)][0];
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[k], li[k], acc, 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) a[4 * q + m] -= acc[m];
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) s += li[k] * Lp[16 * g + jj][k];   // same address for all lanes: broadcast
                a[jj] -= s;
            }
        }
    }
    __syncthreads();
}

// src: the block to factorise, either in place in global memory (src == nullptr) or in an LDS tile (row stride 65)
template <typename T>
__device__ __forceinline__ void potrf_block(T* __restrict__ blk, int64_t ld, int kb, int* __restrict__ fail, T (*src)[65]) {
    __shared__ __attribute__((aligned(16))) T Lp[64][LPS];
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    T a[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) a[jj] = src ? src[i][16 * g + jj] : blk[(int64_t)i * ld + 16 * g + jj];
    potrf_panel<T, 0>(a, Lp, i, g, kb * 64, fail);
    potrf_panel<T, 1>(a, Lp, i, g, kb * 64, fail);
    potrf_panel<T, 2>(a, Lp, i, g, kb * 64, fail);
    potrf_panel<T, 3>(a, Lp, i, g, kb * 64, fail);
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

// ---------------------------------------------------------------------------------------------------------------------------
// 128-column steps: ONE panel launch + ONE plain GEMM launch per 128 columns (the 64-column chain above costs two launches per 64).
// Every workgroup of the panel launch factorises the 128 x 128 diagonal block of the step by itself, in LDS (redundantly: no
// hand-off between workgroups), inverts the factor (W = inv(L), recursive doubling on the MFMA pipe) and forms its own 64 rows of
// the panel as the product L21 = A21 W' instead of a 64-step substitution.  Workgroup 0 publishes the factor of the diagonal
// block (to `stash`, because the other workgroups may still be reading the unfactorised block from M) and the inverses of the
// two 64 x 64 diagonal blocks (dinv).
// ---------------------------------------------------------------------------------------------------------------------------
// Row stride of the LDS tiles of the step kernel: the MFMA fragment reads of mm64 (16 rows x 4 consecutive k per wave) are free of bank
// conflicts when the stride is 4 mod 64 words (fp32: 68) resp. 2 mod 32 double words (fp64: 66)
template <typename T> struct TS { static constexpr int v = sizeof(T) == 4 ? 68 : 66; };
template <typename T> struct LV {   // view of a sub-block of an LDS tile: element (i, j) at p[i * rs + j * cs]
    T* p; int rs, cs;
    __device__ __forceinline__ T& at(int i, int j) const { return p[i * rs + j * cs]; }
};
template <typename T> __device__ __forceinline__ LV<T> lv(T (*t)[TS<T>::v], int r0 = 0, int c0 = 0) { return LV<T>{&t[r0][c0], TS<T>::v, 1}; }
template <typename T> __device__ __forceinline__ LV<T> lvT(T (*t)[TS<T>::v], int r0 = 0, int c0 = 0) { return LV<T>{&t[r0][c0], 1, TS<T>::v}; }   // (i, j) -> t[r0 + j][c0 + i]

// 64 x 64 product held in registers: waves in a 2 x 2 grid, each wave the row tiles {wm, 3 - wm} and the column tiles {wn, 3 - wn}
// (16 x 16 MFMA tiles; the interleaving balances the waves when a triangular operand lets a product skip part of the depth)
template <typename T> struct Acc64 {
    typename Mfma<T>::acc_t v[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[a][b][r] = T(0);
    }
    static __device__ __forceinline__ int tile_of(int w, int a) { return a == 0 ? w : 3 - w; }
    // f(row, col, value) for the 16 entries this lane holds
    template <typename F> __device__ __forceinline__ void foreach(F&& f) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) f(tile_of(wm, a) * 16 + Mfma<T>::row(lane, r), tile_of(wn, b) * 16 + (lane & 15), v[a][b][r]);
    }
    // f(row, col, entry &): fill / modify the entries in place (vector element references cannot be bound: through a scalar)
    template <typename F> __device__ __forceinline__ void foreach_ref(F&& f) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) { T e = v[a][b][r]; f(tile_of(wm, a) * 16 + Mfma<T>::row(lane, r), tile_of(wn, b) * 16 + (lane & 15), e); v[a][b][r] = e; }
    }
};
// c += A * B, all 64 x 64, operands in LDS.  TRI 1: B(k, j) = 0 for k > j (the transpose of a lower-triangular tile): the depth of a
// column tile stops at its last column
template <typename T, int TRI = 0> __device__ __forceinline__ void mm64(Acc64<T>& c, LV<T> A, LV<T> B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int cl = lane & 15, kq = lane >> 4;
    const int r0 = wm * 16 + cl, r1 = (3 - wm) * 16 + cl, c0 = wn * 16 + cl, c1 = (3 - wn) * 16 + cl;
    const int kend0 = TRI == 1 ? (wn + 1) * 16 : 64;                               // column tile wn <= column tile 3 - wn
#pragma unroll 8
    for (int kk = 0; kk < kend0; kk += 4) {
        const int kr = kk + kq;
        const T a0 = A.at(r0, kr), a1 = A.at(r1, kr);
        const T b0 = B.at(kr, c0), b1 = B.at(kr, c1);
        c.v[0][0] = Mfma<T>::run(a0, b0, c.v[0][0]);
        c.v[0][1] = Mfma<T>::run(a0, b1, c.v[0][1]);
        c.v[1][0] = Mfma<T>::run(a1, b0, c.v[1][0]);
        c.v[1][1] = Mfma<T>::run(a1, b1, c.v[1][1]);
    }
    if (TRI == 1) {
        const int kend1 = (4 - wn) * 16;
#pragma unroll 4
        for (int kk = kend0; kk < kend1; kk += 4) {
            const int kr = kk + kq;
            const T a0 = A.at(r0, kr), a1 = A.at(r1, kr);
            const T b1 = B.at(kr, c1);
            c.v[0][1] = Mfma<T>::run(a0, b1, c.v[0][1]);
            c.v[1][1] = Mfma<T>::run(a1, b1, c.v[1][1]);
        }
    }
}
// D = alpha * A * B for an MS x NS x KS sub-block product (multiples of 16), 16 x 16 output tiles dealt to the four waves; D must not
// alias A or B
template <typename T> __device__ __forceinline__ void mm_sub(int MS, int NS, int KS, T alpha, LV<T> A, LV<T> B, LV<T> D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 15, kq = lane >> 4;
    const int tn = NS >> 4, tiles = (MS >> 4) * tn;
    for (int t = wave; t < tiles; t += 4) {
        const int ti = t / tn, tj = t - ti * tn;
        typename Mfma<T>::acc_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = T(0);
        for (int k0 = 0; k0 < KS; k0 += 4) acc = Mfma<T>::run(A.at(ti * 16 + cl, k0 + kq), B.at(k0 + kq, tj * 16 + cl), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) D.at(ti * 16 + Mfma<T>::row(lane, r), tj * 16 + cl) = alpha * acc[r];
    }
}

template <typename T, int I> struct TriInv16 {
    // row I of the inverse of a 16 x 16 lower-triangular block, one column per lane: x_I = (e_I - sum_{p<I} L[I][p] x_p) / L[I][I]
    static __device__ __forceinline__ void run(T (&x)[16], const T (*X)[TS<T>::v], int o, int j, T rd) {
        T s = (j == I) ? T(1) : T(0);
#pragma unroll
        for (int p = 0; p < I; ++p) s -= X[o + I][o + p] * x[p];          // same address in every lane: LDS broadcast
        x[I] = s * lane_bcast<I>(rd);
        TriInv16<T, I + 1>::run(x, X, o, j, rd);
    }
};
template <typename T> struct TriInv16<T, 16> { static __device__ __forceinline__ void run(T (&)[16], const T (*)[TS<T>::v], int, int, T) {} };

// X (64 x 64 lower triangular, upper part zero) <- inv(X) in place; Y: scratch tile.  Ends with a barrier.
template <typename T> __device__ __forceinline__ void tri_inv64(T (*X)[TS<T>::v], T (*Y)[TS<T>::v]) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (g == 0) POTRF_T(12);
    {   // the four 16 x 16 diagonal blocks, one per wave
        const int o = 16 * g, j = lane & 15;
        const T rd = T(1) / X[o + j][o + j];
        T x[16];
        TriInv16<T, 0>::run(x, X, o, j, rd);
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) X[o + i][o + j] = x[i];
        }
    }
    __syncthreads();
    if (g == 0) POTRF_T(13);
    // inv([L00 0; L10 L11]) = [W00 0; -W11 L10 W00, W11]: 16 -> 32 (two pairs), 32 -> 64
    for (int p = 0; p < 2; ++p) mm_sub<T>(16, 16, 16, T(1), lv(X, 32 * p + 16, 32 * p), lv(X, 32 * p, 32 * p), lv(Y, 32 * p + 16, 32 * p));
    __syncthreads();
    for (int p = 0; p < 2; ++p) mm_sub<T>(16, 16, 16, T(-1), lv(X, 32 * p + 16, 32 * p + 16), lv(Y, 32 * p + 16, 32 * p), lv(X, 32 * p + 16, 32 * p));
    __syncthreads();
    if (g == 0) POTRF_T(14);
    mm_sub<T>(32, 32, 32, T(1), lv(X, 32, 0), lv(X, 0, 0), lv(Y, 32, 0));
    __syncthreads();
    mm_sub<T>(32, 32, 32, T(-1), lv(X, 32, 32), lv(Y, 32, 0), lv(X, 32, 0));
    __syncthreads();
    if (g == 0) POTRF_T(15);
}

// Cholesky factor of the 64 x 64 tile in LDS, in place (upper part zeroed).  Ends with a barrier.
template <typename T> __device__ __forceinline__ void potrf_tile(T (*tile)[TS<T>::v], T (*Lp)[LPS], int col0, int* fail) {
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    T a[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) a[jj] = tile[i][16 * g + jj];
    potrf_panel<T, 0>(a, Lp, i, g, col0, fail);
    potrf_panel<T, 1>(a, Lp, i, g, col0, fail);
    potrf_panel<T, 2>(a, Lp, i, g, col0, fail);
    potrf_panel<T, 3>(a, Lp, i, g, col0, fail);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { const int j = 16 * g + jj; tile[i][j] = (j <= i) ? a[jj] : T(0); }
    __syncthreads();
}
template <typename T> __device__ __forceinline__ void tile_load(T (*tile)[TS<T>::v], const T* __restrict__ src, int64_t ld) {
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int r = g + 4 * i; tile[r][c] = src[(int64_t)r * ld + c]; }
}
template <typename T> __device__ __forceinline__ void tile_store(T* __restrict__ dst, int64_t ld, const T (*tile)[TS<T>::v]) {
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int r = g + 4 * i; dst[(int64_t)r * ld + c] = tile[r][c]; }
}

// One step of the factorisation: columns [64 cb, 64 (cb + w)), w = 1 or 2 blocks (w = 1 only for a last odd block: nrb = 0 then).
// grid (max(nrb, 1), batch); workgroup b < nrb owns rows 64 (cb + w + b) .. + 63 of the panel.
// stash: 3 tiles (L00, L10, L11; row-major 64 x 64 each) per step and matrix.
template <typename T>
__global__ __launch_bounds__(256) void k_chol_step(T* __restrict__ M, int64_t ld, int cb, int w, int nrb, T* __restrict__ dinv, T* __restrict__ stash,
                                                   int* __restrict__ fail, int64_t sM, int64_t sD, int64_t sS) {
    __shared__ T B0[64][TS<T>::v];
    __shared__ T B1[64][TS<T>::v];
    __shared__ T B2[64][TS<T>::v];
    __shared__ T B3[64][TS<T>::v];
    __shared__ __attribute__((aligned(16))) T Lp[64][LPS];
    M += (int64_t)blockIdx.y * sM; dinv += (int64_t)blockIdx.y * sD; stash += (int64_t)blockIdx.y * sS + (int64_t)(cb >> 1) * 3 * 4096;
    const bool lead = blockIdx.x == 0;                                             // workgroup-uniform
    int* myfail = lead ? fail + blockIdx.y : nullptr;
    const int c0 = cb * 64;
    const T* D = M + (int64_t)c0 * ld + c0;
    tile_load<T>(B0, D, ld);
    if (w == 2) { tile_load<T>(B1, D + 64 * ld, ld); tile_load<T>(B2, D + 64 * ld + 64, ld); }
    __syncthreads();
    CHOL_T(0);
    potrf_tile<T>(B0, Lp, c0, myfail);
    CHOL_T(1);
    if (lead) tile_store<T>(stash, 64, B0);                                        // L00
    tri_inv64<T>(B0, B3);                                                          // B0 = W00
    CHOL_T(2);
    if (lead) tile_store<T>(dinv + (int64_t)cb * 4096, 64, B0);
    if (w == 2) {
        Acc64<T> c;
        c.zero(); mm64<T, 1>(c, lv(B1), lvT(B0));                                  // L10 = D10 W00'
        __syncthreads();
        c.foreach([&](int r, int cc, T v) { B1[r][cc] = v; });
        __syncthreads();
        if (lead) tile_store<T>(stash + 4096, 64, B1);
        CHOL_T(3);
        c.zero(); mm64<T>(c, lv(B1), lvT(B1));                                     // D11 -= L10 L10'
        c.foreach([&](int r, int cc, T v) { B2[r][cc] -= v; });
        __syncthreads();
        CHOL_T(4);
        potrf_tile<T>(B2, Lp, c0 + 64, myfail);
        CHOL_T(5);
        if (lead) tile_store<T>(stash + 8192, 64, B2);                             // L11
        tri_inv64<T>(B2, B3);                                                      // B2 = W11
        CHOL_T(6);
        if (lead) tile_store<T>(dinv + (int64_t)(cb + 1) * 4096, 64, B2);
        CHOL_T(7);
    }
    if ((int)blockIdx.x < nrb) {
        // this workgroup's 64 rows of L21 = A21 inv(L11)' with inv(L) = [W00 0; -W11 L10 W00, W11]:
        //   X0 = A0 W00',  X1 = (A1 - X0 L10') W11'      (the off-diagonal block of the inverse is never formed)
        T* Ab = M + (int64_t)(c0 + w * 64 + blockIdx.x * 64) * ld + c0;
        tile_load<T>(B3, Ab, ld);
        __syncthreads();
        Acc64<T> x0, x1;
        x0.zero(); mm64<T, 1>(x0, lv(B3), lvT(B0));
        __syncthreads();                                                           // every wave is done with A0 (B3) and W00 (B0)
        x0.foreach([&](int r, int cc, T v) { B3[r][cc] = v; });                    // X0 becomes an operand
        tile_load<T>(B0, Ab + 64, ld);                                             // A1
        __syncthreads();
        x1.zero(); mm64<T>(x1, lv(B3), lvT(B1));                                   // X0 L10'
        __syncthreads();
        x1.foreach([&](int r, int cc, T v) { B3[r][cc] = B0[r][cc] - v; });        // T = A1 - X0 L10'
        __syncthreads();
        x1.zero(); mm64<T, 1>(x1, lv(B3), lvT(B2));
        x0.foreach([&](int r, int cc, T v) { Ab[(int64_t)r * ld + cc] = v; });
        x1.foreach([&](int r, int cc, T v) { Ab[(int64_t)r * ld + 64 + cc] = v; });
        CHOL_T(8);
    }
}
// ---------------------------------------------------------------------------------------------------------------------------
// 128-column steps with the diagonal block OFF the chain of dependent launches (round 4).  In the chain above every step launch spends
// ~25 us (fp32) / ~45 us (fp64) factorising and inverting its 128 x 128 diagonal block before any panel row is touched, and the trailing
// update waits behind it.  Here the diagonal block of step k + 1 is factorised INSIDE the trailing-update launch of step k: workgroup
// (0, 0) of that launch -- the first one dispatched -- forms the leading tile of the updated A22 itself (one round trip for its operands),
// factorises and inverts it, takes the tiles (1, 0) and (1, 1) from the two workgroups that own them in the trailing GEMM (a counter in
// global memory tells it they are stored; by then it has been busy for ~10 us, they need ~5) and finishes the 128 x 128 block while the
// other workgroups run the trailing GEMM.  It leaves L00 / L10 / L11 (stash) and W00 / W11 (dinv) behind; the step launch (k_chol_panel)
// then only loads those and forms its panel rows -- three 64^3 products.
// LDS of the fused launch: the GEMM staging buffers + ONE tile.  The diagonal workgroup reuses the staging buffers as its second tile, as
// the published panel of potrf_tile and as the scratch of tri_inv64, so the trailing GEMM keeps (almost) the occupancy of the plain k_gemm.
// Forward progress: nobody waits for the diagonal workgroup inside the launch, and the two workgroups it waits for wait for nobody, so the
// wait ends as soon as they have been dispatched and run; it is bounded all the same (a fixed number of polls, then the failure word is set
// and the factorisation reports a breakdown instead of hanging).
// ---------------------------------------------------------------------------------------------------------------------------
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {   // compile-time loop: the bodies index register arrays
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// t00 = -(C - R0 R0') for the first 64 rows R0 of the panel A21 (64 x K, k contiguous), C = the leading tile of A22.  K = 0: -C.
// The operand -- 64 rows x 128 columns of the panel, L2 hits -- is requested in ONE go with 16-byte loads and then fed through LDS 32 columns
// at a time (row-major, row stride 4 x odd words: the MFMA fragment reads are conflict free): one memory round trip for the whole product.
// (A staged loop with one slab of prefetch paid a round trip per 16 columns: 8 us, more beside the trailing GEMM.)
template <typename T>
__device__ __forceinline__ void diag_tile00(int K, const T* __restrict__ A21, const T* __restrict__ A22, int64_t ld, T* __restrict__ buf, Acc64<T>& t00) {
    using V = typename Vec16<T>::type;             // (a native vector: arrays of HIP's float4 / double2 structs are not promoted to registers)
    constexpr int VN = 16 / sizeof(T), KQ = 32, S = KQ + (sizeof(T) == 4 ? 4 : 2), VPT = 8 / VN, NQ = 4, VPR = KQ / VN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    t00.foreach_ref([&](int r, int c, T& v) { v = -A22[(int64_t)r * ld + c]; });   // the C tile first (older than every operand load)
    if (K <= 0) return;                                                            // (K is 0 or NQ * KQ = 128: the fused chain has no other depth)
    V x0[NQ * VPT];
    static_for<0, NQ>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        static_for<0, VPT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int f = tid + 256 * j, i = f / VPR, kv = f % VPR;
            x0[p * VPT + j] = *reinterpret_cast<const V*>(A21 + (int64_t)i * ld + p * KQ + kv * VN);
        });
    });
    T* Aq = buf;
    const int cl = lane & 15, kq = lane >> 4;
    const int r0 = (wm * 16 + cl) * S, r1 = ((3 - wm) * 16 + cl) * S, c0 = (wn * 16 + cl) * S, c1 = ((3 - wn) * 16 + cl) * S;   // Acc64's tile interleaving
    static_for<0, NQ>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if (p) __syncthreads();                                                    // the previous quarter is done being read
        static_for<0, VPT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int f = tid + 256 * j, i = f / VPR, kv = f % VPR;
            *reinterpret_cast<V*>(Aq + i * S + kv * VN) = x0[p * VPT + j];
        });
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KQ; kk += 4) {
            const int kr = kk + kq;
            const T a0 = Aq[r0 + kr], a1 = Aq[r1 + kr], b0 = Aq[c0 + kr], b1 = Aq[c1 + kr];
            t00.v[0][0] = Mfma<T>::run(a0, b0, t00.v[0][0]); t00.v[0][1] = Mfma<T>::run(a0, b1, t00.v[0][1]);
            t00.v[1][0] = Mfma<T>::run(a1, b0, t00.v[1][0]); t00.v[1][1] = Mfma<T>::run(a1, b1, t00.v[1][1]);
        }
    });
}

constexpr int DIAG_POLLS = 1 << 22;   // polls of the hand-off counter before the diagonal workgroup gives up (~1 s: never reached by a live launch)

// Factor + inverse of the 128 x 128 (two) or 64 x 64 diagonal block: L00, L10, L11 -> stash (3 tiles), W00 = inv(L00), W11 = inv(L11) ->
// dinv[cb], dinv[cb + 1].  t00: the NEGATED leading tile (diag_tile00); the tiles (1, 0) and (1, 1) are read from d10 / d11 (row stride ld) once
// *ready == 2 (ready == nullptr: they are in place already).  tile0: one LDS tile; spare: 64 * TS elements of LDS that nobody else uses any
// more (second tile / published potrf panel / tri_inv64 scratch in turn).
template <typename T>
__device__ __forceinline__ void diag_factor(Acc64<T>& t00, const T* __restrict__ d10, const T* __restrict__ d11, int64_t ld, int* ready, bool two, int cb,
                                            T (*tile0)[TS<T>::v], T* spare, T* __restrict__ dinv, T* __restrict__ stash, int* fail) {
    T (*B1)[TS<T>::v] = reinterpret_cast<T (*)[TS<T>::v]>(spare);
    T (*Lp)[LPS] = reinterpret_cast<T (*)[LPS]>(spare);
    const int pc = threadIdx.x & 63, pg = threadIdx.x >> 6;                        // tile_load's element order
    __syncthreads();                                                               // the staging buffers (spare) are done being read
    t00.foreach([&](int r, int c, T v) { tile0[r][c] = -v; });
    __syncthreads();
    CHOL_T(1);
    potrf_tile<T>(tile0, Lp, cb * 64, fail);
    CHOL_T(2);
    tile_store<T>(stash, 64, tile0);                                               // L00
    if (two && ready) {
        // tiles (1, 0) and (1, 1) stored by their owners?  One lane polls (acquire at agent scope: this CU's L1 is invalidated, so the loads
        // below, issued behind the barrier, see the owners' stores), then the counter is cleared for the next launch on this matrix.
        if (threadIdx.x == 0) {
            int polls = 0;
            while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 2 && ++polls < DIAG_POLLS) __builtin_amdgcn_s_sleep(2);
            if (polls >= DIAG_POLLS && fail) atomicCAS(fail, 0, -1);               // never seen: reported as a breakdown (the data read below is then stale)
            __hip_atomic_store(ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();                                                               // (also: the inverse below overwrites tile0 in place)
    CHOL_T(3);
    T pre[16];
    if (two) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pre[i] = d10[(int64_t)(pg + 4 * i) * ld + pc];    // D10, in flight during the inverse
    }
    tri_inv64<T>(tile0, B1);                                                       // tile0 = W00 (the published panel in `spare` is dead)
    CHOL_T(4);
    tile_store<T>(dinv + (int64_t)cb * 4096, 64, tile0);
    if (!two) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) B1[pg + 4 * i][pc] = pre[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) pre[i] = d11[(int64_t)(pg + 4 * i) * ld + pc];        // D11, in flight during the product
    __syncthreads();
    Acc64<T> c;
    c.zero(); mm64<T, 1>(c, lv(B1), lvT(tile0));                                   // L10 = D10 W00'
    __syncthreads();
    c.foreach([&](int r, int cc, T v) { B1[r][cc] = v; });
#pragma unroll
    for (int i = 0; i < 16; ++i) tile0[pg + 4 * i][pc] = pre[i];                   // D11 (W00 is not needed any more)
    __syncthreads();
    CHOL_T(5);
    tile_store<T>(stash + 4096, 64, B1);                                           // L10
    c.zero(); mm64<T>(c, lv(B1), lvT(B1));                                         // D11 -= L10 L10'
    c.foreach([&](int r, int cc, T v) { tile0[r][cc] -= v; });
    __syncthreads();                                                               // also: every read of B1 is done before potrf_tile publishes into it
    CHOL_T(6);
    potrf_tile<T>(tile0, Lp, cb * 64 + 64, fail);
    CHOL_T(7);
    tile_store<T>(stash + 8192, 64, tile0);                                        // L11
    __syncthreads();
    tri_inv64<T>(tile0, B1);                                                       // tile0 = W11
    CHOL_T(8);
    tile_store<T>(dinv + (int64_t)(cb + 1) * 4096, 64, tile0);
    CHOL_T(9);
}

// Trailing update of the step that ended at block column cbn (A22 -= L21 L21', lower tiles, depth K = 128) + diagonal block of the NEXT
// step (block columns cbn, cbn + 1) by workgroup 0.  K = 0 with a grid of one: the first diagonal block alone (it also clears the hand-off
// counter of its matrix).  grid (ids, batch): id 0 = the diagonal workgroup, the ids behind it take the lower tiles of the g x g tile grid
// (g = rows below the finished columns / 64) in the order (1, 0), (1, 1), (2, 0), ...  -- the two tiles the diagonal workgroup waits for first.
// avoid = 1: ids that are multiples of 8 take no tile.  Workgroups are dealt round-robin over the 8 XCDs (observed placement, used for speed only), so
// these are the ones that would share the diagonal workgroup's XCD: a trailing GEMM next to it (2 of its workgroups on the same CU, its traffic in the same
// L2) doubles the time of the factorisation chain (potrf 7.8 -> 15.5 us fp32, 14.8 -> 31 us fp64; profiles/r04_b_chol_fused_microbench.log), which costs
// the launch more than the eighth of the chip the GEMM gives up.
// ready: one int per matrix (behind the stash tiles of the matrix).
template <typename T>
__global__ __launch_bounds__(256, 2) void k_chol_update_diag(T* __restrict__ M, int64_t ld, int cbn, int K, int nblk, int g, int avoid, T* __restrict__ dinv,
                                                             T* __restrict__ stash, int64_t ready_off, int* __restrict__ fail, int64_t sM, int64_t sD, int64_t sS) {
    const int L = blockIdx.x;
    const bool two = nblk - cbn >= 2;                                              // the next step is two blocks wide (workgroup-uniform)
    M += (int64_t)blockIdx.y * sM;
    __shared__ __attribute__((aligned(16))) T stage[2][2][GK][GLD];
    __shared__ T tile0[64][TS<T>::v];
    static_assert(2 * 2 * GK * GLD >= 64 * TS<T>::v && 2 * 2 * GK * GLD >= 64 * LPS && 2 * 2 * GK * GLD >= 64 * 36, "the staging buffers double as a tile");
    const T* A21 = M + (int64_t)cbn * 64 * ld + (int64_t)cbn * 64 - K;             // rows of the finished panel below it: columns [64 cbn - K, 64 cbn)
    T* A22 = M + (int64_t)cbn * 64 * ld + (int64_t)cbn * 64;
    T* smat = stash + (int64_t)blockIdx.y * sS;
    int* ready = reinterpret_cast<int*>(smat + ready_off);
    if (L == 0) {                                                                  // the diagonal workgroup
#if defined(QPS_DIAG_DEBUG) && QPS_DIAG_DEBUG == 1
        if (K > 0) return;                                                         // (micro-benchmark only: the trailing GEMM of the fused launch alone)
#endif
        if (K == 0 && threadIdx.x == 0) __hip_atomic_store(ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CHOL_T(0);
        Acc64<T> t00;
        diag_tile00<T>(K, A21, A22, ld, &stage[0][0][0][0], t00);
        diag_factor<T>(t00, A22 + 64 * ld, A22 + 64 * ld + 64, ld, K > 0 ? ready : nullptr, two, cbn, tile0, &stage[0][0][0][0], dinv + (int64_t)blockIdx.y * sD,
                       smat + (int64_t)(cbn >> 1) * 3 * 4096, fail + blockIdx.y);
        return;
    }
    int bi, bj;
    if (!chol_update_tile_of(L, avoid, g, bi, bj)) return;                         // (tile_order.h)
    gemm_tile<T, true, true>(K, T(-1), A21, ld, A21, ld, T(1), A22, ld, 0, bi, bj, stage[0], stage[1], nullptr);
    if (bi == 1) {                                                                 // tiles (1, 0) and (1, 1): the diagonal workgroup is waiting for them
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Panel rows of the step at block columns cb, cb + 1 (both present), diagonal block already factorised by k_chol_update_diag:
// X0 = A0 W00', X1 = (A1 - X0 L10') W11' for this workgroup's 64 rows.  grid (nrb, batch).
template <typename T>
__global__ __launch_bounds__(256) void k_chol_panel(T* __restrict__ M, int64_t ld, int cb, const T* __restrict__ dinv, const T* __restrict__ stash,
                                                    int64_t sM, int64_t sD, int64_t sS) {
    __shared__ T B0[64][TS<T>::v];
    __shared__ T B1[64][TS<T>::v];
    __shared__ T B2[64][TS<T>::v];
    __shared__ T B3[64][TS<T>::v];
    M += (int64_t)blockIdx.y * sM; dinv += (int64_t)blockIdx.y * sD; stash += (int64_t)blockIdx.y * sS + (int64_t)(cb >> 1) * 3 * 4096;
    const int c0 = cb * 64;
    T* Ab = M + (int64_t)(c0 + 128 + blockIdx.x * 64) * ld + c0;
    tile_load<T>(B3, Ab, ld);                                                      // A0
    tile_load<T>(B0, dinv + (int64_t)cb * 4096, 64);                               // W00
    tile_load<T>(B1, stash + 4096, 64);                                            // L10
    tile_load<T>(B2, dinv + (int64_t)(cb + 1) * 4096, 64);                         // W11
    T a1[16];                                                                      // A1 in flight behind them (tile_load's element order)
    {
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) a1[i] = Ab[(int64_t)(g + 4 * i) * ld + 64 + c];
    }
    __syncthreads();
    Acc64<T> x0, x1;
    x0.zero(); mm64<T, 1>(x0, lv(B3), lvT(B0));
    __syncthreads();                                                               // every wave is done with A0 (B3) and W00 (B0)
    x0.foreach([&](int r, int cc, T v) { B3[r][cc] = v; });                        // X0 becomes an operand
    {
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) B0[g + 4 * i][c] = a1[i];                     // A1
    }
    __syncthreads();
    x1.zero(); mm64<T>(x1, lv(B3), lvT(B1));                                       // X0 L10'
    __syncthreads();
    x1.foreach([&](int r, int cc, T v) { B3[r][cc] = B0[r][cc] - v; });            // T = A1 - X0 L10'
    __syncthreads();
    x1.zero(); mm64<T, 1>(x1, lv(B3), lvT(B2));
    x0.foreach([&](int r, int cc, T v) { Ab[(int64_t)r * ld + cc] = v; });
    x1.foreach([&](int r, int cc, T v) { Ab[(int64_t)r * ld + 64 + cc] = v; });
}

// the factors of the diagonal blocks, stashed by the step kernels, go to their place in M
template <typename T>
__global__ __launch_bounds__(256) void k_chol_unstash(T* __restrict__ M, int64_t ld, const T* __restrict__ stash, int nblk, int64_t sM, int64_t sS) {
    const int step = blockIdx.x, which = blockIdx.y;                               // which: 0 = L00, 1 = L10, 2 = L11
    const int cb = 2 * step;
    if (which > 0 && cb + 1 >= nblk) return;
    M += (int64_t)blockIdx.z * sM; stash += (int64_t)blockIdx.z * sS;
    const T* src = stash + ((int64_t)step * 3 + which) * 4096;
    T* dst = M + (int64_t)(cb + (which > 0)) * 64 * ld + (cb + (which == 2)) * 64;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int r = g + 4 * i; dst[(int64_t)r * ld + c] = src[r * 64 + c]; }
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
__global__ __launch_bounds__(256) void k_mirror(int NP, T* __restrict__ S, int blk_tiles) {
    const int bj = blockIdx.x, bi = blockIdx.y;
    if (bj > bi) return;
    if (blk_tiles > 0 && bi / blk_tiles != bj / blk_tiles) return;   // premultiplied form: only the diagonal blocks are mirrored
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
    int lo = lower_only ? 1 : 0;
    // triangular operand: pair every tile with its mirror image along the axis its depth varies on (see k_gemm) when that axis has an even number of tiles and the
    // halved grid still gives every CU work
    static const bool pair_ok = [] { const char* e = getenv("QPS_GEMM_PAIR"); return !(e && atoi(e) == 0); }();
    const int nj = N / GT, ni = M / GT;
    int pair = 0;
    if (pair_ok && !lower_only && ktri != 0 && !(ak && bk) && (int64_t)ni * nj * batch >= 512) pair = (ktri == 1) ? (nj % 2 == 0) : (ni % 2 == 0);
    dim3 grid(pair && ktri == 1 ? nj / 2 : nj, pair && ktri != 1 ? ni / 2 : ni, batch), block(256);
    // lower tiles of a square, chip-filling tile grid: 1-D grid in the XCD-aware order of lower_tile_of (QPS_GEMM_LOWER_MAP=0: the plain 2-D grid)
    static const bool map_ok = [] { const char* e = getenv("QPS_GEMM_LOWER_MAP"); return !(e && atoi(e) == 0); }();
    if (map_ok && lower_only && ni == nj && ni >= 2 && (int64_t)ni * (ni + 1) / 2 * batch >= 512) {   // (batches: the ids of one QP are a multiple of 8, every QP is dealt the same way)
        lo = ni;
        grid = dim3(lower_tile_ids(ni), 1, batch);
    }
    if (ak && bk) hipLaunchKernelGGL((k_gemm<T, true, true>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri, pair);
    else if (ak && !bk) hipLaunchKernelGGL((k_gemm<T, true, false>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri, pair);
    else if (!ak && bk) hipLaunchKernelGGL((k_gemm<T, false, true>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri, pair);
    else hipLaunchKernelGGL((k_gemm<T, false, false>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, lo, sA, sB, sC, ktri, pair);
}

template <typename T> void make_PI(hipStream_t st, int n, int NP, const T* P, T sigma, T* PI, int batch) {
    hipLaunchKernelGGL((k_make_PI<T>), dim3((NP + 255) / 256, NP, batch), dim3(256), 0, st, n, NP, P, sigma, PI);
}
template <typename T> void assemble_M(hipStream_t st, int NP, const T* PI, const T* AA, T rho, T* M, int batch, const double* rho_arr) {
    hipLaunchKernelGGL((k_assemble_M<T>), dim3((NP + 1023) / 1024, NP, batch), dim3(256), 0, st, NP, PI, AA, rho, M, rho_arr);
}

// batch > 1: the same factorisation for `batch` matrices NP*NP apart (dinv blocks (NP/64)*4096 apart, fail flags 1 apart):
// every launch carries all QPs, so the launch-latency-bound panel chain is paid once instead of `batch` times.
// scratch (optional, NP * 96 elements per matrix, batch * NP * NP apart... see chol_scratch_elems): selects the 128-column steps.
static int chol_step_width() {
    static const int w = [] { const char* e = getenv("QPS_CHOL_STEP"); return (e && atoi(e) == 64) ? 64 : 128; }();
    return w;
}
// QPS_CHOL_FUSED=0: the round-3 chain (every step launch factorises its own diagonal block before its panel rows)
static bool chol_fused() { const char* e = getenv("QPS_CHOL_FUSED"); return !(e && atoi(e) == 0); }   // read per factorisation (a test switches it)

// trailing updates of at least this many tiles leave the diagonal workgroup's XCD alone (QPS_CHOL_AVOID: 0 = never)
static int chol_avoid_tiles() { const char* e = getenv("QPS_CHOL_AVOID"); const int v = e ? atoi(e) : 256; return v > 0 ? v : (1 << 30); }

template <typename T> void cholesky(hipStream_t st, int NP, T* M, T* dinv, int* fail_dev, int batch, T* scratch) {
    (void)hipMemsetAsync(fail_dev, 0, sizeof(int) * batch, st);
    const int nblk = NP / 64;
    const int64_t sM = (int64_t)NP * NP, sD = (int64_t)nblk * 4096;
    if (scratch && chol_step_width() == 128 && chol_fused()) {
        // diagonal block of step k + 1 inside the trailing-update launch of step k (k_chol_update_diag); the step launch forms panel rows only
        const int64_t sS = chol_scratch_elems(NP);
        const int64_t roff = sS - 64;                                              // the hand-off counter of a matrix sits behind its stash tiles
        hipLaunchKernelGGL((k_chol_update_diag<T>), dim3(1, batch), dim3(256), 0, st, M, (int64_t)NP, 0, 0, nblk, 0, 0, dinv, scratch, roff, fail_dev, sM, sD, sS);
        for (int cb = 0; cb + 2 <= nblk; cb += 2) {
            const int rem = NP - (cb + 2) * 64;
            if (rem <= 0) break;
            hipLaunchKernelGGL((k_chol_panel<T>), dim3(rem / 64, batch), dim3(256), 0, st, M, (int64_t)NP, cb, dinv, scratch, sM, sD, sS);
            const int g = rem / 64, nt = g * (g + 1) / 2 - 1;                      // lower tiles behind (0, 0)
            // a trailing GEMM that fills the chip is kept off the diagonal workgroup's XCD -- unless it is so long that 8/7 of it exceeds the contended
            // factorisation (fp64 beyond ~56 x 56 tiles: 102.7 against 94.2 us at 62 x 62)
            const int avoid = (batch == 1 && nt >= chol_avoid_tiles() && (sizeof(T) == 4 || nt <= 1500)) ? 1 : 0;
            const int ids = chol_update_ids(g, avoid);
            hipLaunchKernelGGL((k_chol_update_diag<T>), dim3(ids, batch), dim3(256), 0, st, M, (int64_t)NP, cb + 2, 128, nblk, g, avoid, dinv, scratch, roff,
                               fail_dev, sM, sD, sS);
        }
        hipLaunchKernelGGL((k_chol_unstash<T>), dim3((nblk + 1) / 2, 3, batch), dim3(256), 0, st, M, (int64_t)NP, scratch, nblk, sM, sS);
        return;
    }
    if (scratch && chol_step_width() == 128) {
        const int64_t sS = chol_scratch_elems(NP);
        for (int cb = 0; cb < nblk; cb += 2) {
            const int w = (nblk - cb >= 2) ? 2 : 1;
            const int rem = NP - (cb + w) * 64;
            // diagonal block (every workgroup, redundantly) + panel L21 = A21 inv(L11)' (64 rows per workgroup)
            hipLaunchKernelGGL((k_chol_step<T>), dim3(rem > 0 ? rem / 64 : 1, batch), dim3(256), 0, st, M, (int64_t)NP, cb, w, rem / 64, dinv, scratch,
                               fail_dev, sM, sD, sS);
            if (rem > 0) {   // A22 -= L21 L21' (lower tiles), depth 128
                const T* A21 = M + (int64_t)(cb + w) * 64 * NP + cb * 64;
                T* A22 = M + (int64_t)(cb + w) * 64 * NP + (cb + w) * 64;
                gemm<T>(st, rem, rem, w * 64, T(-1), A21, NP, true, A21, NP, true, T(1), A22, NP, true, batch, sM, sM, sM, 0);
            }
        }
        hipLaunchKernelGGL((k_chol_unstash<T>), dim3((nblk + 1) / 2, 3, batch), dim3(256), 0, st, M, (int64_t)NP, scratch, nblk, sM, sS);
        return;
    }
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
template <typename T> void build_sweep_matrix(hipStream_t st, int NP, int nb, const T* L, const T* dinv, T* S, T* tmp, int batch, bool premul) {
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
    if (premul && batch == 1 && nb < NP) {
        // Block rows pre-multiplied by their inverted diagonal block (k_trsv_blocked.hip): one dependent phase per block row.
        //   upper blocks of block row J:  -(L[(J+1)nb.., J] W_JJ)'  = -W_JJ' L[.., J]'   (W_JJ' upper triangular: ktri 3; S upper is still zero)
        //   lower blocks of block row J:  -W_JJ L[J, 0 .. J nb)                          (W_JJ lower triangular: ktri 2)
        const int nblk = (NP + nb - 1) / nb;
        for (int J = 0; J + 1 < nblk; ++J) {
            const int64_t r0 = (int64_t)J * nb, c0 = r0 + nb;
            gemm<T>(st, nb, (int)(NP - c0), nb, T(-1), S + r0 * (NP + 1), NP, false, L + c0 * NP + r0, NP, true, T(0), S + r0 * NP + c0, NP, false, 1, 0, 0, 0, 3);
        }
        for (int J = 1; J < nblk; ++J) {
            const int64_t r0 = (int64_t)J * nb;
            const int rows = (int)std::min<int64_t>(nb, NP - r0);
            gemm<T>(st, rows, (int)r0, rows, T(-1), S + r0 * (NP + 1), NP, true, L + r0 * NP, NP, false, T(0), S + r0 * NP, NP, false, 1, 0, 0, 0, 2);
        }
        hipLaunchKernelGGL((k_mirror<T>), dim3(NP / 64, NP / 64, batch), dim3(256), 0, st, NP, S, nb / 64);
        return;
    }
    hipLaunchKernelGGL((k_mirror<T>), dim3(NP / 64, NP / 64, batch), dim3(256), 0, st, NP, S, 0);
}

#define INST(T)                                                                                                        \
    template void import_colmajor<T>(hipStream_t, const double*, int64_t, int, int, T*, int64_t);                      \
    template void gemm<T>(hipStream_t, int, int, int, T, const T*, int64_t, bool, const T*, int64_t, bool, T, T*, int64_t, \
                          bool, int, int64_t, int64_t, int64_t, int);                                                  \
    template void make_PI<T>(hipStream_t, int, int, const T*, T, T*, int);                                             \
    template void assemble_M<T>(hipStream_t, int, const T*, const T*, T, T*, int, const double*);                      \
    template void cholesky<T>(hipStream_t, int, T*, T*, int*, int, T*);                                                    \
    template void build_sweep_matrix<T>(hipStream_t, int, int, const T*, const T*, T*, T*, int, bool);
INST(double)
INST(float)
#undef INST

}  // namespace qps
