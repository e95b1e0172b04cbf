// k_sparse.hip -- CSR path: SpMV kernels and the device-resident matrix-free CG plugin.
//
// Reference: LinOpCgInit / LinOpCg! (LinearSystemSolvers.jl:145-186): the reduced operator
//     u = P w + rho A'(A w) + sigma w                                   (:152-157)
// is applied matrix-free inside IterativeSolvers.cg!(vXX, mL, vT, abstol = 1e-6, maxiter = 1000) (:179), x~ warm
// started from the previous ADMM iteration, then z~ = A x~ (:181).  cg! itself is un-vendored and unpinned; the
// published algorithm is restated (stop when ||r||_2 <= max(sqrt(eps) ||r0||_2, abstol), r0 = b - A x0).
//
// Storage: A twice as CSR (rows of A for A w; rows of A' -- which *is* the caller's CSC of A, taken over unchanged apart
// from Int64 -> int32 and the index base -- for A' v) and P as CSR (symmetric, so its CSC is its CSR).  fp values,
// int32 indices.  CSR-stream SpMV: a workgroup streams one contiguous slice of (col, val) pairs with unit-stride loads;
// x is gathered (measured: at ~50 nnz/row the kernel is bound by the gather rate of x, not by the CSR streams).
// CG scalars (alpha, beta, residual, done flag) live in device memory: the host enqueues CG iterations in batches and
// reads 32 bytes per batch; iterations enqueued past convergence return at their first instruction.
#include <algorithm>

#include "qps_internal.h"
#include "qps_kernels.h"
#include "qps_ldl.h"
#include "qps_polish.h"
#include "qps_proxqp.h"
#include "k_proxqp_rows.h"
#include "wave_reduce.h"
#include "spmv_layout.h"
#include <hip/hip_ext.h>

namespace qps {

namespace {

struct CgState { double res2, prev2, tol, uc; int iters, done, maxiter, pad; };   // two copies, ping-ponged per CG iteration
using layout::STREAM_NNZ;           // non-zeros streamed per workgroup (256 threads x 4); the layouts themselves are built in spmv_layout.cpp

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    v = wave_sum_all(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// CSR-stream SpMV: a workgroup owns the consecutive rows [rb[b], rb[b+1]) whose non-zeros (<= STREAM_NNZ) form ONE contiguous
// slice of (col, val): the slice is read with unit-stride coalesced loads, 4 gathers of x in flight per thread, products go
// to LDS, then 8 lanes per row sum that row's segment.  A row longer than STREAM_NNZ gets a workgroup of its own (strided
// walk + block reduction).  out[row] = a * (M x)[row] + b0 v0[row] + b1 v1[row]; optional partial[b] = sum_rows dotv[row] * out[row].
template <typename T>
__global__ __launch_bounds__(256) void k_spmv_stream(const int* __restrict__ rb, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     const T* __restrict__ va, const T* __restrict__ x, T* __restrict__ out, T a,
                                                     const T* __restrict__ v0, T b0, const T* __restrict__ v1, T b1,
                                                     const T* __restrict__ dotv, double* __restrict__ partial,
                                                     const CgState* __restrict__ st) {
    if (st && st->done) return;
    __shared__ T prod[STREAM_NNZ];
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
    const int base = rp[r0], end = rp[r1];
    double dot = 0.0;
    if (r1 - r0 == 1 && end - base > STREAM_NNZ) {           // one long row
        T s = T(0);
        for (int k = base + tid; k < end; k += 256) s += va[k] * x[ci[k]];
        const double tot = block_sum_256((double)s, sh);
        if (tid == 0) {
            T r = a * (T)tot;
            if (v0) r += b0 * v0[r0];
            if (v1) r += b1 * v1[r0];
            out[r0] = r;
            if (dotv) dot = (double)dotv[r0] * (double)r;
        }
    } else {
        T p[4]; int c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = base + tid + 256 * j; c[j] = (k < end) ? ci[k] : -1; p[j] = (k < end) ? va[k] : T(0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c[j] >= 0) p[j] *= x[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) prod[tid + 256 * j] = p[j];
        __syncthreads();
        const int lane = tid & 7;
        for (int row = r0 + (tid >> 3); row < r1; row += 32) {
            const int s0 = rp[row] - base, s1 = rp[row + 1] - base;
            T s = T(0);
            for (int k = s0 + lane; k < s1; k += 8) s += prod[k];
            s = oct_sum_all(s);
            if (lane == 0) {
                T r = a * s;
                if (v0) r += b0 * v0[row];
                if (v1) r += b1 * v1[row];
                out[row] = r;
                if (dotv) dot += (double)dotv[row] * (double)r;
            }
        }
    }
    if (partial) {
        const double tot = block_sum_256(dot, sh);
        if (tid == 0) partial[blockIdx.x] = tot;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Column-blocked SpMV with the block of x resident in LDS.
// The CSR-stream kernel above gathers x[col] through L1/L2: at 8 B per gather and 128-B lines the gathers of a 5 M-nnz
// matrix move ~640 MB between L2 and L1 -- ten times the 60 MB of (col, val) that HBM has to deliver -- and bound the
// kernel at ~2 TB/s.  Here the columns are cut into blocks of CB = 56 KiB / sizeof(T) entries; block b of the matrix is its
// own CSR (16-bit local column indices: 10 instead of 12 bytes per non-zero) and a workgroup of block b first copies
// x[b*CB .. (b+1)*CB) into LDS with full-line loads, then streams its tasks (consecutive rows holding <= BCHUNK non-zeros
// of the block) with unit-stride loads and gathers from LDS.  Row sums per block go to partial[b][row]; k_spmv_combine
// adds the blocks in fixed order and applies the epilogue (scale, two axpys, optional dot partials).
// ---------------------------------------------------------------------------------------------------------------------
using layout::BCHUNK;                 // non-zeros per task (512 threads x 4)
using layout::BTHREADS;
template <typename T> struct BlkOf { static constexpr int CB = layout::cb_of<T>(); };   // 56 KiB of x + 16 KiB of products: two workgroups per CU


// four consecutive values as aligned 16-byte accesses, register to register (no pointer casts of local arrays: those end up in scratch)
__device__ __forceinline__ void load4(const double* p, double (&o)[4]) { const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2); o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; }
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) { const float4 a = *reinterpret_cast<const float4*>(p); o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }
__device__ __forceinline__ void store4(double* p, double a, double b, double c, double d) { *reinterpret_cast<double2*>(p) = make_double2(a, b); *reinterpret_cast<double2*>(p + 2) = make_double2(c, d); }
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ double2 vaxpy(double2 r, double beta, double2 u) { return make_double2(r.x + beta * u.x, r.y + beta * u.y); }
__device__ __forceinline__ float4 vaxpy(float4 r, float beta, float4 u) { return make_float4(r.x + beta * u.x, r.y + beta * u.y, r.z + beta * u.z, r.w + beta * u.w); }
__device__ __forceinline__ double2 vmask_tail(double2 v, int c, int cw) { return make_double2(c < cw ? v.x : 0.0, c + 1 < cw ? v.y : 0.0); }
__device__ __forceinline__ float4 vmask_tail(float4 v, int c, int cw) { return make_float4(c < cw ? v.x : 0.f, c + 1 < cw ? v.y : 0.f, c + 2 < cw ? v.z : 0.f, c + 3 < cw ? v.w : 0.f); }
__device__ __forceinline__ double2 vload_tail(const double* p, int c, int cw) { return make_double2(c < cw ? p[c] : 0.0, c + 1 < cw ? p[c + 1] : 0.0); }
__device__ __forceinline__ float4 vload_tail(const float* p, int c, int cw) { return make_float4(c < cw ? p[c] : 0.f, c + 1 < cw ? p[c + 1] : 0.f, c + 2 < cw ? p[c + 2] : 0.f, c + 3 < cw ? p[c + 3] : 0.f); }
__device__ __forceinline__ void vstore_tail(double* p, int c, int cw, double2 v) { if (c < cw) p[c] = v.x; if (c + 1 < cw) p[c + 1] = v.y; }
__device__ __forceinline__ void vstore_tail(float* p, int c, int cw, float4 v) { if (c < cw) p[c] = v.x; if (c + 1 < cw) p[c + 1] = v.y; if (c + 2 < cw) p[c + 2] = v.z; if (c + 3 < cw) p[c + 3] = v.w; }

// In-kernel stamps (diagnostic builds only: -DQPS_SPMV_STAMPS, tests/tools/gpu_c3_stamps.sh): thread 0 of every workgroup records s_memtime at
// the phase boundaries of k_spmv_blk; qps_debug_spmv_stamps() hands the table to the host.  The product is built without it.
#ifdef QPS_SPMV_STAMPS
constexpr int STAMP_WGS = 1024, STAMP_SLOTS = 64;
__device__ long long g_spmv_stamps[STAMP_WGS * STAMP_SLOTS];
__device__ int g_spmv_stamp_rows = 0;
#define QPS_STAMP(k)                                                                                                      \
    do {                                                                                                                  \
        const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                              \
        if (threadIdx.x == 0 && wg_ < STAMP_WGS && (k) < STAMP_SLOTS && (g_spmv_stamp_rows == 0 || g_spmv_stamp_rows == nrows)) \
            g_spmv_stamps[wg_ * STAMP_SLOTS + (k)] = (long long)__builtin_amdgcn_s_memtime();                             \
    } while (0)
#else
#define QPS_STAMP(k) do { } while (0)
#endif

// (rows per task: at most 4 passes x (BTHREADS / LPR) rows = 512 or 256 -- the builder's `max_rows`; their row pointers are staged in LDS)
using layout::BMAXT;                  // tasks per workgroup at most (their descriptors are staged in LDS)
// Optional fusion of the CG direction update into the x-block load of the operator's first product:  u_new = r + beta u_old,
// beta from the ||r||^2 partials of the previous iteration; block (0,0) publishes the scalars (what k_cg_next_u does).
// u_new goes to a second buffer (other workgroups of the same column block still read u_old).
template <typename T> struct CgFuse {
    const T* r = nullptr; const T* uold = nullptr; T* unew = nullptr; const double* part_rr = nullptr; int nparts = 0;
    const CgState* cur = nullptr; CgState* nxt = nullptr; int first = 0;
};
// task descriptor: rows [x, y) of the block, non-zeros [z, w) of the blocked arrays
template <typename T, int LPR>        // LPR lanes add up one row segment (4 when a row holds <= ~8 non-zeros per block)
__global__ __launch_bounds__(BTHREADS) void k_spmv_blk(int nrows, int ncols, const int* __restrict__ task_ptr, const int4* __restrict__ tasks, int per,
                                                       const int* __restrict__ rp, const unsigned short* __restrict__ ci, const T* __restrict__ va,
                                                       const int* __restrict__ lr_ptr, const int4* __restrict__ lr_desc,
                                                       const T* __restrict__ xin, T* __restrict__ partial, const CgState* __restrict__ st, CgFuse<T> fu) {
    if (st && st->done) return;
    constexpr int CB = BlkOf<T>::CB;
    __shared__ __align__(16) T xs[CB + 8];                      // xs[CB] = 0: what an entry past the end of a task is multiplied by
    __shared__ __align__(16) T prod[BCHUNK];
    __shared__ int rps[BTHREADS + 2];
    __shared__ int4 meta[BMAXT];
    __shared__ double sh[BTHREADS / 64];
    const int tid = threadIdx.x, b = blockIdx.y;
    T beta = T(0);
    if (fu.r) {
        if (fu.cur->done) { if (blockIdx.x == 0 && b == 0 && tid == 0) *fu.nxt = *fu.cur; return; }
        double res2 = fu.cur->res2, prev2 = fu.cur->prev2; int iters = fu.cur->iters, done = 0;
        if (!fu.first) {
            double d = 0.0;
            for (int i = tid; i < fu.nparts; i += BTHREADS) d += fu.part_rr[i];            // same order in every workgroup
            d = wave_sum_all(d);
            if ((tid & 63) == 0) sh[tid >> 6] = d;
            __syncthreads();
            d = 0.0;
            for (int w = 0; w < BTHREADS / 64; ++w) d += sh[w];
            __syncthreads();
            prev2 = res2; res2 = d; iters += 1;
            if (sqrt(d) <= fu.cur->tol || iters >= fu.cur->maxiter) done = 1;
        }
        if (blockIdx.x == 0 && b == 0 && tid == 0) { CgState s_ = *fu.cur; s_.res2 = res2; s_.prev2 = prev2; s_.iters = iters; s_.done = done; *fu.nxt = s_; }
        if (done) return;
        beta = (T)(res2 / prev2);
    }
    // this workgroup's tasks: a contiguous run [ts, te) of block b, so their descriptors come with one coalesced load and
    // no data load ever waits on a chain of dependent loads
    const int ts = task_ptr[b] + blockIdx.x * per, te = min(task_ptr[b + 1], ts + per), nt = max(te - ts, 0);
    const int lr0 = lr_ptr[b] + blockIdx.x, lr1 = lr_ptr[b + 1];   // rows longer than a task: dealt round-robin over the block's workgroups
    if (nt <= 0 && lr0 >= lr1) return;
    QPS_STAMP(0);
    // the descriptors of the first two tasks at workgroup-uniform addresses (scalar loads): their data loads leave together with the x block
    // instead of behind the descriptor table's trip through LDS (one dependent memory round trip less in front of the first task).
    // (A workgroup without tasks -- it has long rows only -- reads descriptor 0 of the table and never uses it.)
    const int4 m0 = tasks[nt > 0 ? ts : 0], m1 = tasks[nt > 0 ? min(ts + 1, te - 1) : 0];
    int4 mt = make_int4(0, 0, 0, 0);
    if (tid < nt) mt = tasks[ts + tid];
    const int c0 = b * CB, cw = min(CB, ncols - c0);
    // x block with 16-byte loads (the kernel is bound by the NUMBER of vector-memory instructions a CU can retire -- ~20-30 cycles each whatever
    // their width, profiles/r03_b_c3_counters_before.json -- so every stream is read with the widest load its alignment allows)
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N, XV = CB / (BTHREADS * VN);
    static_assert(XV * BTHREADS * VN == CB, "x block = whole 16-byte loads per thread");
    V xr[XV];
    const bool xal = ((reinterpret_cast<uintptr_t>(xin + c0) & 15u) == 0) && (!fu.r || (((reinterpret_cast<uintptr_t>(fu.uold + c0) | reinterpret_cast<uintptr_t>(fu.unew + c0)) & 15u) == 0));   // (uniform)
    if (xal) {
        // every load first, unconditional, at a clamped position (vectors are allocated 64 elements past their length, so the vector that straddles the
        // end of the last block is readable); elements past the block are zeroed afterwards and the direction update stores behind all loads
        const int clast = ((cw + VN - 1) / VN - 1) * VN;
        V uo[XV];
#pragma unroll
        for (int j = 0; j < XV; ++j) xr[j] = *reinterpret_cast<const V*>(xin + c0 + min((tid + BTHREADS * j) * VN, clast));
        if (fu.r) {
#pragma unroll
            for (int j = 0; j < XV; ++j) uo[j] = *reinterpret_cast<const V*>(fu.uold + c0 + min((tid + BTHREADS * j) * VN, clast));
#pragma unroll
            for (int j = 0; j < XV; ++j) xr[j] = vaxpy(xr[j], beta, uo[j]);                              // xin = r
        }
#pragma unroll
        for (int j = 0; j < XV; ++j) xr[j] = vmask_tail(xr[j], (tid + BTHREADS * j) * VN, cw);
        if (fu.r && blockIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < XV; ++j) { const int c = (tid + BTHREADS * j) * VN; if (c < cw) *reinterpret_cast<V*>(fu.unew + c0 + c) = xr[j]; }
        }
    } else {
#pragma unroll
        for (int j = 0; j < XV; ++j) xr[j] = vload_tail(xin + c0, (tid + BTHREADS * j) * VN, cw);
        if (fu.r) {
#pragma unroll
            for (int j = 0; j < XV; ++j) {
                const int c = (tid + BTHREADS * j) * VN;
                xr[j] = vaxpy(xr[j], beta, vload_tail(fu.uold + c0, c, cw));
                if (blockIdx.x == 0) vstore_tail(fu.unew + c0, c, cw, xr[j]);
            }
        }
    }
    QPS_STAMP(1);
    const int* rpb = rp + (int64_t)b * (nrows + 1);
    T* pout = partial + (int64_t)b * nrows;
    // software pipeline, two tasks deep: (col, val) slices and row pointers of the next two tasks are in flight while one is reduced.
    // A thread owns FOUR CONSECUTIVE entries of a task: their 16-bit columns are one 8-byte load, their values one (fp32) or two (fp64) 16-byte
    // loads -- the builder starts every task at a multiple of four entries (zero entries in between), so these loads are aligned.
    struct Regs { uint2 cp; T v[4]; int r; int r0, r1, base, end; };   // fetched values stay RAW until their task: arithmetic on them here would wait for them here
    Regs RA, RB;
    // UNCONDITIONAL loads at clamped positions (a thread past the slice re-reads the slice's first entries and marks them c = -1 afterwards): a
    // predicated load cannot be counted by the compiler, which then waits for every outstanding load (vmcnt(0)) before it touches ANY fetched
    // register -- the two-tasks-deep pipeline below waited for the slices it had just requested before reducing the one it had
    auto fetch_desc = [&](Regs& R, const int4 m) {
        R.r0 = m.x; R.r1 = m.y; R.base = m.z; R.end = m.w;
        const int k = R.base + 4 * tid;
        const bool valid = k < R.end;
        const int kk = valid ? k : R.base;
        R.cp = *reinterpret_cast<const uint2*>(ci + kk);
        load4(va + kk, R.v);
        R.r = rpb[min(R.r0 + tid, R.r1)];                       // a task holds at most RPP * PG <= BTHREADS rows: one row pointer per thread
    };
    auto fetch = [&](Regs& R, int i) { fetch_desc(R, meta[i]); };
    constexpr int RPP = BTHREADS / LPR;                         // rows per pass
    constexpr int PG = 4, NIT = 4;                              // passes per task, entries per lane read without looking at the row length
    // Every task issues the SAME number of vector-memory operations -- 3 or 4 loads, one store, no store inside a loop or a branch -- so that the
    // compiler's in-order count of outstanding operations is exact and the wait in front of a task's data is for THAT data (fetched two tasks ago),
    // not for whatever was requested last.  (With the row sums stored from inside the pass loop the count was unknown and every task waited for
    // part of the fetch issued a thousand cycles earlier: the two-tasks-deep pipeline had an effective depth of one.)
    T* const dump = partial + (int64_t)gridDim.y * nrows;       // 64 spare elements behind the partial sums: where a lane without a row stores
    auto process = [&](Regs& R, int inext, int ti) {
        (void)ti;
        const int r0 = R.r0, r1 = R.r1, base = R.base, end = R.end;
        {   // four unconditional gathers in flight together (a select around a gather becomes a branch with a wait of its own)
            const bool valid = base + 4 * tid < end;            // past the end: the zero slot of xs
            const int q0 = valid ? (int)(R.cp.x & 0xffffu) : CB, q1 = valid ? (int)(R.cp.x >> 16) : CB;
            const int q2 = valid ? (int)(R.cp.y & 0xffffu) : CB, q3 = valid ? (int)(R.cp.y >> 16) : CB;
            const T x0 = xs[q0], x1 = xs[q1], x2 = xs[q2], x3 = xs[q3];
            store4(prod + 4 * tid, R.v[0] * x0, R.v[1] * x1, R.v[2] * x2, R.v[3] * x3);
        }
        QPS_STAMP(3 + 4 * ti);
        {
            const int nr_ = r1 - r0;
            if (tid < nr_) rps[tid] = R.r - base;
            if (tid == 0) rps[nr_] = end - base;                // (the zero entries in front of the next task belong to the last row: harmless)
        }
        fetch(R, min(inext, nt - 1));                           // this register set is free again (past the end: the last task again, never used)
        __syncthreads();
        QPS_STAMP(4 + 4 * ti);
        // Row sums.  The reads of a pass depend on each other (row pointers -> product segment -> sum): one pass at a time is a chain of LDS round
        // trips (measured with in-kernel stamps: 2.4 k cycles of a 5.7 k-cycle task).  PG passes are walked together: all their row pointers first,
        // then NIT entries per lane of every pass at clamped positions (no branch, all reads in flight), leftovers of longer rows in a loop, and
        // ONE store per group: lane q of a row's lane group keeps the sum of pass q, so a wave writes four runs of consecutive rows at once.
        const int nr = r1 - r0, lane = tid & (LPR - 1), row_in_pass = tid / LPR;
        {
            int s0[PG], s1[PG];
#pragma unroll
            for (int q = 0; q < PG; ++q) {
                const int i = row_in_pass + q * RPP, ic = min(i, max(nr - 1, 0));
                s0[q] = rps[ic]; s1[q] = rps[ic + 1];
                if (i >= nr) s1[q] = s0[q];                      // no such row: an empty segment
            }
            T e[PG][NIT];
#pragma unroll
            for (int q = 0; q < PG; ++q)
#pragma unroll
                for (int t = 0; t < NIT; ++t) e[q][t] = prod[min(s0[q] + lane + LPR * t, BCHUNK - 1)];
            T sum[PG];
#pragma unroll
            for (int q = 0; q < PG; ++q) {
                sum[q] = T(0);
#pragma unroll
                for (int t = 0; t < NIT; ++t) sum[q] += (s0[q] + lane + LPR * t < s1[q]) ? e[q][t] : T(0);
            }
#pragma unroll
            for (int q = 0; q < PG; ++q)
                for (int k = s0[q] + lane + LPR * NIT; k < s1[q]; k += LPR) sum[q] += prod[k];     // rows with more than LPR * NIT entries in this block
            T mine = T(0);
#pragma unroll
            for (int q = 0; q < PG; ++q) {
                const T tot = (LPR == 8) ? oct_sum_all(sum[q]) : quad_sum_all(sum[q]);   // DPP: the LDS pipe is busy with the x gathers
                mine = (lane == q) ? tot : mine;
            }
            const int i = row_in_pass + lane * RPP;
            T* dst = (lane < PG && i < nr) ? pout + r0 + i : dump + (tid & 63);
            *dst = mine;                                         // ONE unconditional store per task
        }
        QPS_STAMP(5 + 4 * ti);
        __syncthreads();
        QPS_STAMP(6 + 4 * ti);
    };
    fetch_desc(RA, m0);
    fetch_desc(RB, m1);
#pragma unroll
    for (int j = 0; j < XV; ++j) *reinterpret_cast<V*>(xs + (tid + BTHREADS * j) * VN) = xr[j];
    if (tid < 8) xs[CB + tid] = T(0);
    if (tid < nt) meta[tid] = mt;
    __syncthreads();                                            // xs and the descriptor table complete
    QPS_STAMP(2);
    // pairs of tasks: inside this loop BOTH register sets are always walked, so the count of operations between a fetch and its use is the same on
    // every path (with the second task behind an `if` the compiler assumed the path without it and waited for the fetch it had just issued)
    int i = 0;
    for (; i + 1 < nt; i += 2) {
        process(RA, i + 2, i);
        process(RB, i + 3, i + 1);
    }
    if (i < nt) process(RA, i + 2, i);
    // rows with more than BCHUNK entries in this block: strided walk by the whole workgroup + block reduction
    for (int li = lr0; li < lr1; li += gridDim.x) {
        const int4 d = lr_desc[li];
        T s = T(0);
        for (int k = d.y + tid; k < d.z; k += BTHREADS) s += va[k] * xs[ci[k]];
        double dd = (double)s;
        dd = wave_sum_all(dd);
        __syncthreads();
        if ((tid & 63) == 0) sh[tid >> 6] = dd;
        __syncthreads();
        if (tid == 0) { double tot = 0.0; for (int w = 0; w < BTHREADS / 64; ++w) tot += sh[w]; pout[d.x] = (T)tot; }
    }
    QPS_STAMP(63);
}


// ---------------------------------------------------------------------------------------------------------------------
// Column-blocked SpMV, sliced form (round 3).  What the task form above costs per non-zero, measured (counters: profiles/r03_b_c3_counters_before.json;
// in-kernel stamps: tests/tools/gpu_c3_stamps.sh): the x gather from LDS, the product written to LDS, the product read back from LDS by the lanes of
// its row -- three LDS accesses per entry -- and two workgroup barriers per 2048 entries; a task lasts 5.7 k cycles of which 2.2 k are the row sums,
// however the reads are batched: the LDS pipe is the shared resource the two workgroups of a CU fight for.
//
// Here a LANE owns a ROW, so a product never leaves its register: the only LDS access per entry is the gather of x.  To keep the loads coalesced the
// rows of a column block are cut into slices of 64 (one per lane) and stored unit by unit: unit u of a slice holds entries u*E .. u*E+E-1 of all 64
// rows, lane after lane (E = 2 in fp64, 4 in fp32: one 16-byte load of values and one 4- / 8-byte load of 16-bit columns per lane and unit).  A slice
// is padded to the length of its longest row, so the rows are first sorted by their length in this block inside windows of SIGMA rows (sliced ELL with
// a sorting window): the 64 rows of a slice then have nearly the same length and the padding stays small, while a row's partial sum still
// lands within SIGMA rows of where its neighbours' do (the permutation is 16 bits per row and block).  Padding entries point at xs[CB] = 0 with value 0.
// Rows with more than SLONG entries in one block (a dense constraint row) would pad their whole slice: they are left out of the slices and summed by a
// wave each.  No barrier after the x block is in place, no reduction, no row pointers; the waves of a workgroup walk their slices independently.
// (Tried and dropped: slices handed out one ticket at a time from a per-block atomic counter, to even out the workgroups' lifetimes -- ~600 waves
// adding to one address serialise at ~80 ns each: 240 us per product instead of 27.  Non-temporal loads of the matrix: no difference.)
// ---------------------------------------------------------------------------------------------------------------------
using layout::SIGMA;                  // sorting window (rows), 2048.  8192 measured: slice padding on BASELINE config 3 5.2 % -> 1.4 % of the entries (pair rounding: 7 % either
                                      // way), products 27.9 / 17.1 -> 33.4 / 18.9 us -- the row sums of a slice then scatter over 64 KB of the partial sums instead of 16 KB
using layout::SLONG;                  // more entries than this in one block: the row is summed by a wave of its own
constexpr int SU = 4;                 // units of a slice in flight per batch (8: same time -- the launch is not bound by round trips per slice)
struct SellDims { int nrows, ncols, nsl, wpb, win, nwin, staged; };   // nsl = slices per column block; win / nwin / staged: the sorting window (spmv_layout.h, SellLayout)
template <typename T> struct SellOf;
template <> struct SellOf<double> { static constexpr int E = layout::sell_e<double>(); using CV = unsigned; };
template <> struct SellOf<float> { static constexpr int E = layout::sell_e<float>(); using CV = uint2; };
static_assert(sizeof(int4) == sizeof(layout::Int4), "descriptors are uploaded as the host builder lays them out");
__device__ __forceinline__ void sell_cols(unsigned p, int (&c)[2]) { c[0] = (int)(p & 0xffffu); c[1] = (int)(p >> 16); }
__device__ __forceinline__ void sell_cols(uint2 p, int (&c)[4]) { c[0] = (int)(p.x & 0xffffu); c[1] = (int)(p.x >> 16); c[2] = (int)(p.y & 0xffffu); c[3] = (int)(p.y >> 16); }
__device__ __forceinline__ void sell_vals(double2 v, double (&o)[2]) { o[0] = v.x; o[1] = v.y; }
__device__ __forceinline__ void sell_vals(float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
// wg_ptr [nblk][wpb + 1] slice range of every workgroup; sl_off [nblk * nsl + 1] first unit of a slice; perm [nblk * nsl * 64] row of a lane
// relative to its window (0xffff: no row); cols / vals: unit-major, 64 lanes x E entries per unit; lr_ptr / lr_desc: long rows (row, first, end)
template <typename T>
__global__ __launch_bounds__(BTHREADS) void k_spmv_sell(SellDims a, const int* __restrict__ wg_ptr, const int* __restrict__ sl_off,
                                                        const unsigned short* __restrict__ perm, const typename SellOf<T>::CV* __restrict__ cols,
                                                        const typename VecOf<T>::type* __restrict__ vals, const unsigned short* __restrict__ lci,
                                                        const T* __restrict__ lva, const int* __restrict__ lr_ptr, const int4* __restrict__ lr_desc,
                                                        const T* __restrict__ xin, T* __restrict__ partial, const CgState* __restrict__ st, CgFuse<T> fu) {
    if (st && st->done) return;
    constexpr int CB = BlkOf<T>::CB, NW = BTHREADS / 64, E = SellOf<T>::E;
    using V = typename VecOf<T>::type;
    using CV = typename SellOf<T>::CV;
    constexpr int VN = VecOf<T>::N, XV = CB / (BTHREADS * VN);
    __shared__ __align__(16) T xs[CB + 8];                      // xs[CB] = 0: what a padding entry is multiplied by
    __shared__ double sh[NW];
    __shared__ T wsum[layout::WMAX];                            // staged form: the row sums of the window in progress (xs + wsum: two workgroups per CU still fit)
    const int nrows = a.nrows;
    const int tid = threadIdx.x, b = blockIdx.y, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: slice offsets stay scalar
    T beta = T(0);
    if (fu.r) {                                                 // same prologue as k_spmv_blk: u = r + beta u_old while the x block is loaded
        if (fu.cur->done) { if (blockIdx.x == 0 && b == 0 && tid == 0) *fu.nxt = *fu.cur; return; }
        double res2 = fu.cur->res2, prev2 = fu.cur->prev2; int iters = fu.cur->iters, done = 0;
        if (!fu.first) {
            double d = 0.0;
            for (int i = tid; i < fu.nparts; i += BTHREADS) d += fu.part_rr[i];            // same order in every workgroup
            d = wave_sum_all(d);
            if (lane == 0) sh[wave] = d;
            __syncthreads();
            d = 0.0;
            for (int w = 0; w < NW; ++w) d += sh[w];
            prev2 = res2; res2 = d; iters += 1;
            if (sqrt(d) <= fu.cur->tol || iters >= fu.cur->maxiter) done = 1;
        }
        if (blockIdx.x == 0 && b == 0 && tid == 0) { CgState s_ = *fu.cur; s_.res2 = res2; s_.prev2 = prev2; s_.iters = iters; s_.done = done; *fu.nxt = s_; }
        if (done) return;
        beta = (T)(res2 / prev2);
    }
    const int s_begin = wg_ptr[b * (a.wpb + 1) + blockIdx.x], s_end = wg_ptr[b * (a.wpb + 1) + blockIdx.x + 1];
    const int lr0 = lr_ptr[b * a.nwin], lr1 = lr_ptr[(b + 1) * a.nwin];
    if (s_begin >= s_end && (a.staged || lr0 + (int)blockIdx.x * NW >= lr1)) return;   // (workgroup-uniform)
    QPS_STAMP(0);
    const int c0 = b * CB, cw = min(CB, a.ncols - c0);
    const int64_t sb = (int64_t)b * a.nsl;
    {
        V xr[XV];
        const bool xal = ((reinterpret_cast<uintptr_t>(xin + c0) & 15u) == 0) && (!fu.r || (((reinterpret_cast<uintptr_t>(fu.uold + c0) | reinterpret_cast<uintptr_t>(fu.unew + c0)) & 15u) == 0));   // (uniform)
        if (xal) {
            // every load first, unconditional, at a clamped position (vectors are allocated 64 elements past their length, so the vector that straddles
            // the end of the last block is readable); elements past the block are zeroed afterwards and the direction update stores behind all loads
            const int clast = ((cw + VN - 1) / VN - 1) * VN;
            V uo[XV];
#pragma unroll
            for (int j = 0; j < XV; ++j) xr[j] = *reinterpret_cast<const V*>(xin + c0 + min((tid + BTHREADS * j) * VN, clast));
            if (fu.r) {
#pragma unroll
                for (int j = 0; j < XV; ++j) uo[j] = *reinterpret_cast<const V*>(fu.uold + c0 + min((tid + BTHREADS * j) * VN, clast));
#pragma unroll
                for (int j = 0; j < XV; ++j) xr[j] = vaxpy(xr[j], beta, uo[j]);                          // xin = r
            }
#pragma unroll
            for (int j = 0; j < XV; ++j) xr[j] = vmask_tail(xr[j], (tid + BTHREADS * j) * VN, cw);
            if (fu.r && blockIdx.x == 0) {
#pragma unroll
                for (int j = 0; j < XV; ++j) { const int c = (tid + BTHREADS * j) * VN; if (c < cw) *reinterpret_cast<V*>(fu.unew + c0 + c) = xr[j]; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < XV; ++j) xr[j] = vload_tail(xin + c0, (tid + BTHREADS * j) * VN, cw);
            if (fu.r) {
#pragma unroll
                for (int j = 0; j < XV; ++j) {
                    const int c = (tid + BTHREADS * j) * VN;
                    xr[j] = vaxpy(xr[j], beta, vload_tail(fu.uold + c0, c, cw));
                    if (blockIdx.x == 0) vstore_tail(fu.unew + c0, c, cw, xr[j]);
                }
            }
        }
        QPS_STAMP(1);
#pragma unroll
        for (int j = 0; j < XV; ++j) *reinterpret_cast<V*>(xs + (tid + BTHREADS * j) * VN) = xr[j];
        if (tid < 8) xs[CB + tid] = T(0);
    }
    __syncthreads();                                            // xs complete; the only barrier of the kernel
    QPS_STAMP(2);
    T* pout = partial + (int64_t)b * nrows;
    const int spw = a.win >> 6;                                 // slices per sorting window
    // One slice: the lane's row sum.  (UNCONDITIONAL loads: a unit past the slice re-reads its last one and is not used)
    auto slice_sum = [&](int s) {
        const int u0 = sl_off[sb + s], nu = sl_off[sb + s + 1] - u0;            // (scalar loads)
        T acc = T(0);
        for (int ub = 0; ub < nu; ub += SU) {
            CV cp[SU]; V vv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int64_t at = ((int64_t)(u0 + min(ub + u, nu - 1))) * 64 + lane;
                cp[u] = cols[at]; vv[u] = vals[at];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                if (ub + u < nu) {                              // (scalar condition: no vector-memory operation behind it)
                    int c[E]; T v[E];
                    sell_cols(cp[u], c); sell_vals(vv[u], v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc += v[e] * xs[c[e]];
                }
            }
        }
        return acc;
    };
    auto long_sum = [&](int i) {                                // a long row: one wave, every lane returns the sum
        const int4 d = lr_desc[i];
        T sum = T(0);
        for (int k = d.y + lane; k < d.z; k += 64) sum += lva[k] * xs[lci[k]];
        return wave_sum_all(sum);
    };
    if (a.staged) {
        // The workgroup owns whole windows (wg_ptr boundaries are multiples of spw).  A lane's row sum goes to its row's place in wsum (LDS), the long rows of the
        // window are summed by the same workgroup, and the window leaves as ONE contiguous run of stores.  Stored lane by lane (below), the 64 row sums of a slice are
        // 64 partial writes of 64-byte lines scattered over the window: on BASELINE config 3 the partial sums were 40 % of the launch's L2 transactions and 2.5 x
        // their size in HBM write traffic (profiles/r04_z_pmc_traffic_c3.txt); with stores as coalesced as these the products take 10 / 13 % less
        // (profiles/r04_q_c3_coalesced_store_experiment.log).
        for (int w = s_begin / spw; w * spw < s_end; ++w) {
            const int s1 = min((w + 1) * spw, a.nsl);
            for (int s = w * spw + wave; s < s1; s += NW) {
                const unsigned pm = perm[(sb + s) * 64 + lane];
                const T acc = slice_sum(s);
                if (pm != 0xffffu) wsum[pm] = acc;
            }
            for (int i = lr_ptr[b * a.nwin + w] + wave; i < lr_ptr[b * a.nwin + w + 1]; i += NW) {
                const T sum = long_sum(i);
                if (lane == 0) wsum[lr_desc[i].x - w * a.win] = sum;
            }
            __syncthreads();
            const int r0 = w * a.win, wr = min(a.win, nrows - r0);
            for (int i = tid; i < wr; i += BTHREADS) pout[r0 + i] = wsum[i];
            if ((w + 1) * spw < s_end) __syncthreads();         // (uniform) the next window reuses wsum
        }
        QPS_STAMP(63);
        return;
    }
    for (int s = s_begin + wave; s < s_end; s += NW) {
        const int u0 = sl_off[sb + s], nu = sl_off[sb + s + 1] - u0;            // (scalar loads)
        const unsigned pm = perm[(sb + s) * 64 + lane];
        T acc = T(0);
        for (int ub = 0; ub < nu; ub += SU) {
            CV cp[SU]; V vv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {                      // UNCONDITIONAL loads (a unit past the slice re-reads its last one and is not used)
                const int64_t at = ((int64_t)(u0 + min(ub + u, nu - 1))) * 64 + lane;
                cp[u] = cols[at]; vv[u] = vals[at];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                if (ub + u < nu) {                              // (scalar condition: no vector-memory operation behind it)
                    int c[E]; T v[E];
                    sell_cols(cp[u], c); sell_vals(vv[u], v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc += v[e] * xs[c[e]];
                }
            }
        }
        if (pm != 0xffffu) pout[(s / spw) * a.win + (int)pm] = acc;             // spw slices per sorting window
        QPS_STAMP(3 + (s - s_begin) / NW);
    }
    // long rows of this block: one wave per row
    for (int i = lr0 + blockIdx.x * NW + wave; i < lr1; i += a.wpb * NW) {
        const int4 d = lr_desc[i];
        T sum = T(0);
        for (int k = d.y + lane; k < d.z; k += 64) sum += lva[k] * xs[lci[k]];
        sum = wave_sum_all(sum);
        if (lane == 0) pout[d.x] = sum;
    }
    QPS_STAMP(63);
}

// out[row] = a0 * sum_b p0[b][row] + a1 * sum_b p1[b][row] + b0 v0[row] + b1 v1[row]; optional partials of dot(dotv, out)
template <typename T>
__global__ __launch_bounds__(256) void k_spmv_combine(int nrows, const T* __restrict__ p0, int n0, int64_t s0, T a0, const T* __restrict__ p1, int n1, int64_t s1_, T a1,
                                                      const T* __restrict__ v0, T b0, const T* __restrict__ v1, T b1, T* __restrict__ out,
                                                      const T* __restrict__ dotv, double* __restrict__ partial, const CgState* __restrict__ st) {
    if (st && st->done) return;
    __shared__ double sh[4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    double dot = 0.0;
    if (row < nrows) {
        T s = T(0);
#pragma unroll 8
        for (int b = 0; b < n0; ++b) s += p0[(int64_t)b * s0 + row];                      // independent loads, several in flight
        T r = a0 * s;
        if (p1) {
            T s1 = T(0);
#pragma unroll 8
            for (int b = 0; b < n1; ++b) s1 += p1[(int64_t)b * s1_ + row];
            r += a1 * s1;
        }
        if (v0) r += b0 * v0[row];
        if (v1) r += b1 * v1[row];
        out[row] = r;
        if (dotv) dot = (double)dotv[row] * (double)r;
    }
    if (partial) {
        const double tot = block_sum_256(dot, sh);
        if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    }
}

// CG iteration on the blocked products, second launch: blocks [0, nbn) add up the P u partials of their 256 rows (pu), blocks beyond add up the
// A u partials (v = A u).  Since u'(P u + sigma u + rho A'A u) = u'Pu + sigma u'u + rho v'v, the partials of dot(u, c) are complete HERE, before
// the A' product: the x / r update then rides in the last combine of the iteration instead of a launch of its own.
template <typename T>
__global__ __launch_bounds__(256) void k_cg_combine_pa(int n, int m, int nbn, const T* __restrict__ pp, int nblk, int64_t stride, const T* __restrict__ u,
                                                       T sigma, T rho, T* __restrict__ pu, T* __restrict__ v, double* __restrict__ part_uc,
                                                       const CgState* __restrict__ st) {
    if (st->done) return;
    __shared__ double sh[4];
    const bool prow = (int)blockIdx.x < nbn;                                       // workgroup-uniform
    const int row = (prow ? blockIdx.x : blockIdx.x - nbn) * 256 + threadIdx.x;
    double d = 0.0;
    if (row < (prow ? n : m)) {
        const T* p0 = pp + (prow ? 0 : n) + row;
        T s = T(0);
#pragma unroll 8
        for (int b = 0; b < nblk; ++b) s += p0[(int64_t)b * stride];               // independent loads, several in flight
        if (prow) { const T ui = u[row]; pu[row] = s; d = (double)ui * ((double)s + (double)sigma * (double)ui); }
        else { v[row] = s; d = (double)rho * (double)s * (double)s; }
    }
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) part_uc[blockIdx.x] = d;
}
// ... fourth and last launch: c = pu + rho sum_b (A' v)_b + sigma u is formed in registers, alpha = res2 / dot(u, c) from the partials above (same
// order in every block -> same alpha), x += alpha u, r -= alpha c, partials of ||r||^2 (LinearSystemSolvers.jl:179: the body of cg!)
template <typename T>
__global__ __launch_bounds__(256) void k_cg_combine_update(int n, const T* __restrict__ pu, const T* __restrict__ pat, int nblk, int64_t stride, T rho, T sigma,
                                                           const T* __restrict__ u, int nparts_uc, const double* __restrict__ part_uc, T* __restrict__ x,
                                                           T* __restrict__ r, double* __restrict__ part_rr, const CgState* __restrict__ st) {
    if (st->done) return;
    __shared__ double sh[4];
    double uc = 0.0;
    for (int i = threadIdx.x; i < nparts_uc; i += 256) uc += part_uc[i];
    uc = block_sum_256(uc, sh);
    const T alpha = (T)(st->res2 / uc);
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0;
    if (i < n) {
        T s = T(0);
#pragma unroll 8
        for (int b = 0; b < nblk; ++b) s += pat[(int64_t)b * stride + i];
        const T ui = u[i];
        const T c = pu[i] + rho * s + sigma * ui;
        x[i] += alpha * ui;
        const T ri = r[i] - alpha * c;
        r[i] = ri;
        d = (double)ri * (double)ri;
    }
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = d;
}

// r = b - c, u = 0, partial ||r||^2
template <typename T>
__global__ __launch_bounds__(256) void k_cg_init(int n, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ r,
                                                 T* __restrict__ u, double* __restrict__ partial) {
    __shared__ double sh[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0;
    if (i < n) { const T ri = b[i] - c[i]; r[i] = ri; u[i] = T(0); d = (double)ri * (double)ri; }
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = d;
}
__global__ __launch_bounds__(256) void k_cg_init_final(int nparts, const double* __restrict__ partial, CgState* st, double abstol, double reltol, int maxiter) {
    __shared__ double sh[4];
    double d = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) d += partial[i];
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) {
        const double residual = sqrt(d);
        st->res2 = d; st->prev2 = 1.0;
        st->tol = fmax(reltol * residual, abstol);   // reltol = sqrt(eps(real(eltype(b)))): the handle's arithmetic type
        st->iters = 0; st->maxiter = maxiter;
        st->done = (!(residual <= st->tol) && 0 < maxiter) ? 0 : 1;
    }
}
// alpha = res2 / dot(u, c); x += alpha u; r -= alpha c; partial ||r||^2
template <typename T>
__global__ __launch_bounds__(256) void k_cg_update_xr(int n, int nparts_uc, const double* __restrict__ partial_uc,
                                                      const T* __restrict__ u, const T* __restrict__ c, T* __restrict__ x,
                                                      T* __restrict__ r, double* __restrict__ partial_rr, const CgState* __restrict__ st) {
    if (st->done) return;
    __shared__ double sh[4];
    double uc = 0.0;
    for (int i = threadIdx.x; i < nparts_uc; i += 256) uc += partial_uc[i];   // same order in every block -> same alpha
    uc = block_sum_256(uc, sh);
    const T alpha = (T)(st->res2 / uc);
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0;
    if (i < n) {
        x[i] += alpha * u[i];
        const T ri = r[i] - alpha * c[i];
        r[i] = ri;
        d = (double)ri * (double)ri;
    }
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) partial_rr[blockIdx.x] = d;
}
__global__ __launch_bounds__(256) void k_cg_finish(int nparts, const double* __restrict__ partial_rr, CgState* st) {
    if (st->done) return;
    __shared__ double sh[4];
    double d = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) d += partial_rr[i];
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) {
        st->prev2 = st->res2; st->res2 = d; st->iters += 1;
        const double residual = sqrt(d);
        if (residual <= st->tol || st->iters >= st->maxiter) st->done = 1;
    }
}
// Merged "finish previous iteration + start this one": every block reduces the ||r||^2 partials of the previous
// iteration in the same order (same value in every block), block 0 publishes the new scalars into the OTHER state slot
// (nobody reads that slot during this launch), then u = r + beta u with beta = res2_new / res2_old.
template <typename T>
__global__ __launch_bounds__(256) void k_cg_next_u(int n, int nparts, const double* __restrict__ partial_rr, const T* __restrict__ r,
                                                   T* __restrict__ u, const CgState* __restrict__ cur, CgState* __restrict__ nxt, int first) {
    __shared__ double sh[4];
    if (cur->done) { if (blockIdx.x == 0 && threadIdx.x == 0) *nxt = *cur; return; }
    double res2 = cur->res2, prev2 = cur->prev2; int iters = cur->iters, done = 0;
    if (!first) {
        double d = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 256) d += partial_rr[i];
        d = block_sum_256(d, sh);
        prev2 = res2; res2 = d; iters += 1;
        if (sqrt(d) <= cur->tol || iters >= cur->maxiter) done = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CgState s = *cur; s.res2 = res2; s.prev2 = prev2; s.iters = iters; s.done = done; *nxt = s;
    }
    if (done) return;
    const T beta = (T)(res2 / prev2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) u[i] = r[i] + beta * u[i];
}
template <typename T>
__global__ __launch_bounds__(256) void k_axpby(int n, T a, const T* __restrict__ x, T b, const T* __restrict__ y, T* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}

// mL = mPI + rho * mAA on the frozen pattern of the explicit reduced matrix: out = vP + sigma * diag + rho * vAA   (LinearSystemSolvers.jl:113-114, :128)
template <typename T>
__global__ __launch_bounds__(256) void k_build_reduced(int64_t nnz, const T* __restrict__ vP, const T* __restrict__ vAA, const T* __restrict__ dg, T sigma, T rho, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nnz) out[i] = vP[i] + sigma * dg[i] + rho * vAA[i];
}
// values of a column-blocked copy refreshed from the CSR values they were laid out from: dst[i] = va[src[i]] (src < 0: padding, stays 0)
template <typename T>
__global__ __launch_bounds__(256) void k_refresh_values(int64_t count, const int* __restrict__ src, const T* __restrict__ va, T* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) { const int k = src[i]; dst[i] = k >= 0 ? va[k] : T(0); }
}

struct Csr {
    int nrows = 0; int64_t nnz = 0; int* rp = nullptr; int* ci = nullptr; void* va = nullptr; int* rb = nullptr; int nblocks = 0;
    // column-blocked copy (k_spmv_blk): nblk CSR blocks back to back; used when `blocked`
    bool blocked = false; int ncols = 0, nblk = 0, wpb = 0; int* brp = nullptr; unsigned short* bci = nullptr; void* bva = nullptr;
    int* task_ptr = nullptr; int4* tasks = nullptr; int per = 1, lpr4 = 0; void* partial = nullptr;
    int* lr_ptr = nullptr; int4* lr_desc = nullptr;   // rows with more than BCHUNK entries in one block: (row, first entry, end entry, 0), [nblk + 1] ranges
    // sliced form (k_spmv_sell): units of 64 lanes x E entries; bci / bva then hold only the long rows' entries
    bool sell = false; int nsl = 0, win = layout::SIGMA, nwin = 0, staged = 0; int* wg_ptr = nullptr; int* sl_off = nullptr; unsigned short* sl_perm = nullptr; void* s_cols = nullptr; void* s_vals = nullptr;
    // value refresh of a matrix with a frozen pattern (the explicit reduced matrix): where every slot of s_vals / bva came from in `va`
    int* src = nullptr; int64_t src_n = 0; int* lsrc = nullptr; int64_t lsrc_n = 0;
};

template <typename T> struct SparseSolver : SolverBase {
    HandleResources res; std::unique_ptr<StagedUploader> up;   // `up` lives for the duration of the constructor only
    Csr A, At, P, PA;   // PA = [P; A] stacked, column-blocked only: P u and A u of the CG operator from ONE pass over u
    // ItrSolCgInit / ItrSolCg! (LinearSystemSolvers.jl:110-142): the explicit reduced matrix mL = mPI + rho mAA on a frozen pattern; ONE product per CG iteration
    Csr Lm; T *L_vP = nullptr, *L_vAA = nullptr, *L_dg = nullptr;
    int explicit_state = 0;          // 0 undecided, 1 built, -1 declined (too dense to pay, or its inputs were released)
    bool cg_explicit = false; bool L_valid = false; double L_rho = 0, L_sigma = 0; int num_L_builds = 0;
    T *q = nullptr, *l = nullptr, *u = nullptr, *x = nullptr, *xp = nullptr, *z = nullptr, *zp = nullptr, *y = nullptr;
    T *xx = nullptr, *zz = nullptr, *w = nullptr, *tt = nullptr, *cu = nullptr, *cu2 = nullptr, *cr = nullptr, *cc = nullptr, *tm = nullptr;
    T *Ax = nullptr, *Px = nullptr, *Aty = nullptr;
    double *part_uc = nullptr, *part_rr = nullptr; int part_uc_cap = 0; CgState* state = nullptr; CgState* state_host = nullptr;
    unsigned long long* scratch = nullptr; double* res_dev = nullptr; double* res_host = nullptr; double* stage = nullptr;
    int nb_n = 0, nb_op = 0; int64_t cg_total = 0; int last_cg = 4;
    double eps_pcg = 1e-6; int itr_pcg = 1000;
    int cat_spmv, cat_op, cat_vec, cat_chk, cat_ldl, cat_pa = 0, cat_at = 0; int prof_calls = 0;
    // sparse direct KKT plugin (LinearSystemSolvers.jl:16-107): built on first use from canonical host copies of the caller's CSC
    std::vector<int64_t> hPcp, hPri, hAcp, hAri; std::vector<double> hPnz, hAnz;
    std::unique_ptr<SparseLdl<T>> ldl; bool ldl_valid = false; double ldl_rho = 0, ldl_sigma = 0; int plugin_kind = QPS_LINSYS_CG;
    int num_factorizations = 0; bool ldl_unfit = false;   // ldl_unfit: the analysis refused this pattern once (AUTO then goes to CG without asking again)
    // hipGraph replay of ONE plain iteration of the direct plugin (a dozen launches of 2-3 us of work each: the eager loop is bound by
    // the host's launch rate).  Keyed by the scalars baked into the kernel arguments; every pointer of the loop is fixed.
    struct IterGraph { double rho, sigma, alpha; hipGraphExec_t exec; };
    std::vector<IterGraph> graphs; bool graphs_disabled = false;
    void drop_graphs() { for (auto& gr : graphs) (void)hipGraphExecDestroy(gr.exec); graphs.clear(); }
    hipGraphExec_t iter_graph(double rho, double sigma, double alpha) {
        for (auto& gr : graphs) if (gr.rho == rho && gr.sigma == sigma && gr.alpha == alpha) return gr.exec;
        if (graphs_disabled) return nullptr;
        if (graphs.size() >= 8) drop_graphs();
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        ldl->iterate(x, xp, q, z, zp, y, l, u, alpha, rho, sigma, true);                            // SolveQuadraticProgram.jl:54-61, right-hand side already in place
        // nothing was enqueued while capturing, so a failure here just means: run this handle's iterations eagerly from now on
        if (hipStreamEndCapture(st, &graph) != hipSuccess || graph == nullptr) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        graphs.push_back({rho, sigma, alpha, exec});
        return exec;
    }

    // Column-blocked copy of M: the sliced form (k_spmv_sell) when the host builder produces one, else the task form (k_spmv_blk).  The layouts are
    // built by plain host code (spmv_layout.cpp, tested on the CPU against scipy); here they are only uploaded.
    template <typename V> V* upload_array(const std::vector<V>& v, int64_t min_count = 1) {
        V* d = dalloc<V>(std::max<int64_t>((int64_t)v.size(), min_count), st);
        if (!v.empty()) up->copy(d, v.data(), sizeof(V) * v.size());
        return d;
    }
    void build_blocked(Csr& M, const layout::CsrHost& H, bool with_src = false) {
        static const int wgs_env = [] { const char* e = getenv("QPS_SPMV_WGS"); return e ? atoi(e) : 0; }();
        const int wgs = wgs_env > 0 ? wgs_env : 512;
        M.ncols = H.ncols;
        const char* e = getenv("QPS_SPMV_SELL");                     // read per handle: 0 = the task form (k_spmv_blk)
        if (!(e && atoi(e) == 0)) {
            layout::SellLayout<T> L;
            const char* es = getenv("QPS_SPMV_STAGED");                 // read per handle: 0 = row sums stored lane by lane also when whole windows could be staged
            if (layout::build_sell<T>(H, wgs, L, with_src, !(es && atoi(es) == 0))) {
                M.nblk = L.nblk; M.nsl = L.nsl; M.wpb = L.wpb; M.win = L.win; M.nwin = L.nwin; M.staged = L.staged;
                if (with_src) { M.src = upload_array(L.src); M.src_n = (int64_t)L.src.size(); M.lsrc = upload_array(L.lsrc); M.lsrc_n = (int64_t)L.lsrc.size(); }
                M.s_cols = upload_array(L.cols); M.s_vals = upload_array(L.vals);
                M.bci = upload_array(L.lci); M.bva = upload_array(L.lva);
                M.sl_off = upload_array(L.sl_off); M.sl_perm = upload_array(L.perm); M.wg_ptr = upload_array(L.wg_ptr);
                M.lr_ptr = upload_array(L.lr_ptr); M.lr_desc = reinterpret_cast<int4*>(upload_array(L.lr));
                M.partial = dalloc<T>((int64_t)L.nblk * H.nrows + 64, st);
                M.blocked = true; M.sell = true;
                return;
            }
        }
        layout::TaskLayout<T> L;
        try { layout::build_tasks<T>(H, wgs, L, with_src); }
        catch (const std::length_error& ex) { throw QpsError(QPS_ERR_BAD_DIMENSION, ex.what()); }
        M.nblk = L.nblk; M.wpb = L.wpb; M.per = L.per; M.lpr4 = L.lpr4;
        if (with_src) { M.src = upload_array(L.src); M.src_n = (int64_t)L.src.size(); }
        M.brp = upload_array(L.brp); M.bci = upload_array(L.bci); M.bva = upload_array(L.bva);
        M.task_ptr = upload_array(L.task_ptr); M.tasks = reinterpret_cast<int4*>(upload_array(L.tasks));
        M.lr_ptr = upload_array(L.lr_ptr); M.lr_desc = reinterpret_cast<int4*>(upload_array(L.lr));
        M.partial = dalloc<T>((int64_t)L.nblk * H.nrows + 64, st);    // + 64: the dump slots of lanes without a row
        M.blocked = true;
    }
    void upload_csr(Csr& M, const layout::CsrHost& H, bool with_src = false) {
        M.nrows = H.nrows; M.nnz = (int64_t)H.ci.size();
        {   // LDS-resident x pays once the gathers dominate; QPS_SPMV_BLOCKED = 1 / 0 forces the choice
            // ... and only while the per-block partial sums (one per row and column block, written by the product and read back by its combine launch) stay below
            // the matrix itself: a long banded matrix (n = 400 000: 56 column blocks) has 2.7 partial sums per entry -- 165 us per product where the CSR-stream
            // kernel, whose gathers of x stay inside the band and hit the cache, needs a third of that
            const char* e = getenv("QPS_SPMV_BLOCKED");
            const int64_t nblk_ = (H.ncols + BlkOf<T>::CB - 1) / BlkOf<T>::CB;
            const bool want = e ? atoi(e) != 0 : (M.nnz >= 200000 && nblk_ * (int64_t)H.nrows <= M.nnz);
            if (want && H.nrows > 0 && H.ncols > 0) build_blocked(M, H, with_src);
        }
        M.rp = upload_array(H.rp); M.ci = upload_array(H.ci);
        {
            std::vector<T> v(H.va.begin(), H.va.end());
            M.va = upload_array(v);
        }
        const std::vector<int> rbv = layout::stream_row_blocks(H);   // row blocks of the CSR-stream kernel
        M.nblocks = (int)rbv.size() - 1;
        M.rb = upload_array(rbv);
    }
    void upload_vec(const double* h, T* d, int64_t count) {
        if (count <= 0) return;
        HIPC(hipMemcpyAsync(stage, h, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st));
        convert_copy<T>(st, stage, d, count);
        HIPC(hipStreamSynchronize(st));
    }
    void download_vec(const T* d, double* h, int64_t count) {
        if (count <= 0) return;
        convert_back<T>(st, d, stage, count);
        HIPC(hipMemcpyAsync(h, stage, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    // ItrSolCgInit (LinearSystemSolvers.jl:110-122): mAA = mA' * mA, mPI = mP + sigma I, mL = mPI + rho * mAA, formed on the host ONCE per handle as three value
    // arrays on the frozen pattern of mL (spmv_layout.cpp: reduced_matrix); returns whether the handle has it.  forced = an explicit QPS_LINSYS_CG_EXPLICIT request;
    // otherwise the matrix is only formed when it pays -- A'A cheap to form (sum of squared row lengths) and mL no larger than 1.5 x the entries the matrix-free
    // operator streams per application (nnz P + 2 nnz A).  BASELINE config 3 (unstructured, ~50 entries per row) fails the first test in O(m) and stays matrix-free.
    bool explicit_prepare(bool forced) {
        if (explicit_state != 0) {
            if (forced && explicit_state < 0) throw QpsError(QPS_ERR_UNSUPPORTED, "QPS_LINSYS_CG_EXPLICIT: the reduced matrix of this handle cannot be formed (too many entries, or the handle's "
                                                                                  "host copies were released to the direct plugin before the first CG request)");
            return explicit_state > 0;
        }
        explicit_state = -1;
        if (hPcp.empty() || hAcp.empty()) { if (forced) return explicit_prepare(true); return false; }   // released (the L D L' plugin was built first)
        const int64_t pnnz = (int64_t)hPri.size(), annz = (int64_t)hAri.size();
        layout::CsrHost Ph, Ah, Ath, Lh; std::vector<double> vAA, dg;
        try {
            Ph = layout::csc_as_transposed_csr(n, n, hPcp, hPri, hPnz);
            layout::csc_to_csr_pair(m, n, hAcp, hAri, hAnz, Ah, Ath);
            if (!forced && layout::ata_work(Ah) > 16 * (pnnz + annz)) return false;
            const int64_t cap = forced ? 1900000000LL : (3 * (pnnz + 2 * annz)) / 2 + n;
            if (!layout::reduced_matrix(Ph, Ah, Ath, cap, Lh, vAA, dg)) { if (forced) return explicit_prepare(true); return false; }
        } catch (const std::length_error& ex) { throw QpsError(QPS_ERR_BAD_DIMENSION, ex.what()); }
        up.reset(new StagedUploader(st));
        upload_csr(Lm, Lh, true);
        std::vector<T> t1(Lh.va.begin(), Lh.va.end()), t2(vAA.begin(), vAA.end()), t3(dg.begin(), dg.end());
        L_vP = upload_array(t1); L_vAA = upload_array(t2); L_dg = upload_array(t3);
        up.reset();
        HIPC(hipStreamSynchronize(st));
        if (dot_parts(Lm) + 64 > part_uc_cap) {                                                     // partials of dot(u, mL u): one per workgroup of the product's last launch
            (void)hipFree(part_uc);
            part_uc_cap = dot_parts(Lm) + 64; part_uc = dalloc<double>(part_uc_cap, st);
        }
        explicit_state = 1; L_valid = false;
        return true;
    }
    // the values of mL for (rho, sigma): LinearSystemSolvers.jl:114 at Init, :127-129 on changedRho -- elementwise on the frozen pattern, then the column-blocked copy refreshed
    void reduced_values(double rho, double sigma) {
        if (L_valid && L_rho == rho && L_sigma == sigma) return;
        const int64_t nnz = Lm.nnz;
        if (nnz > 0) hipLaunchKernelGGL((k_build_reduced<T>), dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, L_vP, L_vAA, L_dg, (T)sigma, (T)rho, static_cast<T*>(Lm.va));
        if (Lm.blocked) {
            T* main_vals = static_cast<T*>(Lm.sell ? Lm.s_vals : Lm.bva);
            if (Lm.src_n > 0) hipLaunchKernelGGL((k_refresh_values<T>), dim3((unsigned)((Lm.src_n + 255) / 256)), dim3(256), 0, st, Lm.src_n, Lm.src, static_cast<const T*>(Lm.va), main_vals);
            if (Lm.sell && Lm.lsrc_n > 0)
                hipLaunchKernelGGL((k_refresh_values<T>), dim3((unsigned)((Lm.lsrc_n + 255) / 256)), dim3(256), 0, st, Lm.lsrc_n, Lm.lsrc, static_cast<const T*>(Lm.va), static_cast<T*>(Lm.bva));
        }
        L_valid = true; L_rho = rho; L_sigma = sigma; ++num_L_builds;
    }
    // Which plugin a request means on a CSC handle, with the direct one factorised.  QPS_LINSYS_AUTO is the reference's modeAuto rule
    // (SolveQuadraticProgram.jl:143-151, qps_linsys_auto) on this handle's sizes and non-zero counts, so that a C caller passing AUTO reaches the direct
    // KKT plugin exactly when the reference would; when the factor then does not fit the level-scheduled plugin (QPS_ERR_UNSUPPORTED from the analysis) an
    // AUTO request falls back to CG -- an EXPLICIT QPS_LINSYS_KKT_LDL request still fails loudly.
    // A CG request is served by the explicit reduced matrix (ItrSolCg, one product per CG iteration) when the handle has one -- always for
    // QPS_LINSYS_CG_EXPLICIT, for QPS_LINSYS_CG / AUTO when explicit_prepare finds that it pays (QPS_CG_EXPLICIT = 0 keeps those matrix-free) -- else by the
    // matrix-free operator (LinOpCg / LinMapsCg, three products).
    int resolve_cg(int linsys, double rho, double sigma) {
        const char* e = getenv("QPS_CG_EXPLICIT");                                                  // read per request
        cg_explicit = linsys == QPS_LINSYS_CG_EXPLICIT ? explicit_prepare(true) : (!(e && atoi(e) == 0) && explicit_prepare(false));
        if (cg_explicit) reduced_values(rho, sigma);
        return QPS_LINSYS_CG;
    }
    int resolve_kind(int linsys, double rho, double sigma, bool force) {
        int kind = linsys;
        if (linsys == QPS_LINSYS_AUTO) kind = qps_linsys_auto(n, m, P.nnz, A.nnz, 1);
        if (kind != QPS_LINSYS_KKT_LDL) return resolve_cg(linsys, rho, sigma);
        if (ldl_unfit && linsys == QPS_LINSYS_AUTO) return resolve_cg(linsys, rho, sigma);
        try { ldl_prepare(rho, sigma, force); }
        catch (const QpsError& e) {
            if (linsys == QPS_LINSYS_AUTO && e.code == QPS_ERR_UNSUPPORTED) { ldl_unfit = true; return resolve_cg(linsys, rho, sigma); }
            throw;
        }
        return QPS_LINSYS_KKT_LDL;
    }
    // LaLdlInit / QDLdlInit / FacLdlInit (LinearSystemSolvers.jl:16-24, :47-55, :78-86): ordering + symbolic once per handle, numeric per (rho, sigma)
    void ldl_prepare(double rho, double sigma, bool force) {
        if (!ldl) {
            if (n + m > 2000000000LL) throw QpsError(QPS_ERR_BAD_DIMENSION, "KKT matrix too large for the sparse direct plugin");
            // read at every analysis (once per handle), not once per process: the limits are part of a handle's layout
            const char *e1 = getenv("QPS_LDL_MAX_TAIL"), *e2 = getenv("QPS_LDL_MIN_LEVEL"), *e3 = getenv("QPS_LDL_MAX_LEVELS");
            const int max_tail = e1 ? atoi(e1) : 8192, min_level = e2 ? atoi(e2) : 64, max_levels = e3 ? atoi(e3) : 4096;
            LdlSymbolic sym;
            try { sym = ldl_analyze((int)n, (int)m, hPcp.data(), hPri.data(), hAcp.data(), hAri.data(), 0, max_tail, min_level, max_levels); }
            catch (const std::runtime_error& e) { throw QpsError(QPS_ERR_UNSUPPORTED, e.what()); }
            ldl = make_sparse_ldl<T>(st, std::move(sym), hPnz.data(), (int64_t)hPnz.size(), hAnz.data(), (int64_t)hAnz.size());
            ldl_valid = false;
            for (auto* v : {&hPcp, &hPri, &hAcp, &hAri}) std::vector<int64_t>().swap(*v);           // the canonical host copies have served their purpose
            std::vector<double>().swap(hPnz); std::vector<double>().swap(hAnz);                     // (a later CG request stays matrix-free: explicit_prepare)
        }
        if (force || !ldl_valid || ldl_rho != rho || ldl_sigma != sigma) {
            ldl_valid = false;
            ++num_factorizations;
            ldl->factorize(rho, sigma);
            ldl_valid = true; ldl_rho = rho; ldl_sigma = sigma;
        }
    }

    SparseSolver(int dev, int64_t n_, int64_t m_, int dt, const int64_t* Pcp, const int64_t* Pri, const double* Pnz,
                 const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* qh, const double* lh, const double* uh, int base) {
        device = dev; n = n_; m = m_; dtype = dt; sparse = true;
        HIPC(hipSetDevice(device));
        res = acquire_resources(device, 0);   // recycled stream + pinned block (qps_internal.h)
        st = res.st;
        prof.st = st;
        up.reset(new StagedUploader(st));
        if (Pcp[n] - base > 2000000000LL || Acp[n] - base > 2000000000LL) throw QpsError(QPS_ERR_BAD_DIMENSION, "more than 2^31 non-zeros");
        // canonical host copies first (sorted rows, duplicates summed, 0-based: a C caller's CSC need not be what Julia's sparse() guarantees);
        // they feed the CSR copies below and, later, the analysis of the L D L' plugin
        layout::canonical_csc(n, Pcp, Pri, Pnz, base, hPcp, hPri, hPnz);
        layout::canonical_csc(n, Acp, Ari, Anz, base, hAcp, hAri, hAnz);
        const int64_t pnnz = (int64_t)hPri.size(), annz = (int64_t)hAri.size();
        layout::CsrHost Ph, Ah, Ath;
        try {
            Ph = layout::csc_as_transposed_csr(n, n, hPcp, hPri, hPnz);      // P: CSC == CSR (symmetric, full storage): its columns are its rows
            layout::csc_to_csr_pair(m, n, hAcp, hAri, hAnz, Ah, Ath);        // rows of A by a counting sort; rows of A' are the columns of A == the CSC
        } catch (const std::length_error& ex) { throw QpsError(QPS_ERR_BAD_DIMENSION, ex.what()); }
        upload_csr(P, Ph);
        upload_csr(At, Ath);
        upload_csr(A, Ah);
        const char* fuse_env = getenv("QPS_SPMV_FUSEPA");                 // 0: keep P, A, A' as three separate blocked products
        if (m > 0 && P.blocked && A.blocked && At.blocked && pnnz + annz < 2000000000LL && !(fuse_env && atoi(fuse_env) == 0)) {
            const layout::CsrHost Sh = layout::stack_rows(Ph, Ah);
            PA.nrows = (int)(n + m); PA.nnz = pnnz + annz;
            build_blocked(PA, Sh);
        }
        const int64_t nn = n + 64, mm = m + 64;
        q = dalloc<T>(nn, st); x = dalloc<T>(nn, st); xp = dalloc<T>(nn, st); xx = dalloc<T>(nn, st); tt = dalloc<T>(nn, st);
        cu = dalloc<T>(nn, st); cu2 = dalloc<T>(nn, st); cr = dalloc<T>(nn, st); cc = dalloc<T>(nn, st); Px = dalloc<T>(nn, st); Aty = dalloc<T>(nn, st);
        l = dalloc<T>(mm, st); u = dalloc<T>(mm, st); z = dalloc<T>(mm, st); zp = dalloc<T>(mm, st); y = dalloc<T>(mm, st); zz = dalloc<T>(mm, st);
        w = dalloc<T>(mm, st); tm = dalloc<T>(mm, st); Ax = dalloc<T>(mm, st);
        nb_n = (int)((n + 255) / 256);
        part_uc_cap = std::max(std::max(At.nblocks, P.nblocks), nb_n + (int)((m + 255) / 256)) + 64;
        part_uc = dalloc<double>(part_uc_cap, st); part_rr = dalloc<double>(nb_n + 64, st);
        state = reinterpret_cast<CgState*>(dalloc<double>(16, st));
        state_host = reinterpret_cast<CgState*>(res.pinned);                          // pinned block: CG state | check results
        scratch = dalloc<unsigned long long>(16, st); res_dev = dalloc<double>(16, st);
        res_host = reinterpret_cast<double*>(reinterpret_cast<char*>(res.pinned) + 128);
        stage = dalloc<double>(std::max(nn, mm) + 64, st);
        upload_vec(qh, q, n); upload_vec(lh, l, m); upload_vec(uh, u, m);
        const double s = sizeof(T);
        auto spmv_bytes = [&](const Csr& M, int cols) { return (double)M.nnz * (s + 4) + M.nrows * 4.0 + s * (M.nrows + cols); };
        cat_spmv = prof.category("spmv(A / A')", 0.5 * (spmv_bytes(A, (int)n) + spmv_bytes(At, (int)m)));
        cat_op = prof.category("cg_iteration(A u, P u + rho A'. + sigma u, axpys)", spmv_bytes(A, (int)n) + spmv_bytes(At, (int)m) + spmv_bytes(P, (int)n) + s * 10.0 * n);
        cat_vec = prof.category("admm_update", s * (3.0 * n + 7.0 * m));
        cat_chk = prof.category("check_convergence", spmv_bytes(A, (int)n) + spmv_bytes(At, (int)m) + spmv_bytes(P, (int)n));
        cat_ldl = prof.category("kkt_ldl_solve(rhs, level sweeps, dense tail, post)", 0.0);   // bytes are known once the factor exists
        // the dominant kernel of the CG path: the stacked [P; A] product (nnz * 12 + rows * 4 + s * (rows + cols))
        cat_pa = prof.category("spmv_blk([P;A] u, x block in LDS)", (double)(pnnz + annz) * (s + 4) + (double)(n + m) * 4.0 + s * (double)(n + m + n));
        cat_at = prof.category("spmv_blk(A' v, x block in LDS)", spmv_bytes(At, (int)m));
        // every fill and every upload above was enqueued on `st` (qps_internal.h, stream-ordering rule): nothing to wait for here
        up.reset();
    }
    ~SparseSolver() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        drop_graphs();
        if (cu2) (void)hipFree(cu2);
        for (Csr* M_ : {&A, &At, &P, &PA, &Lm}) {
            void* bp[] = {M_->brp, M_->bci, M_->bva, M_->task_ptr, M_->tasks, M_->partial, M_->lr_ptr, M_->lr_desc, M_->wg_ptr, M_->sl_off, M_->sl_perm, M_->s_cols, M_->s_vals, M_->src, M_->lsrc};
            for (void* p : bp) if (p) (void)hipFree(p);
        }
        { void* lp[] = {Lm.rp, Lm.ci, Lm.va, Lm.rb, L_vP, L_vAA, L_dg}; for (void* p : lp) if (p) (void)hipFree(p); }
        void* ptrs[] = {A.rp, A.ci, A.va, A.rb, At.rp, At.ci, At.va, At.rb, P.rp, P.ci, P.va, P.rb, q, l, u, x, xp, z, zp, y, xx, zz, w, tt, cu, cr, cc, tm,
                        Ax, Px, Aty, part_uc, part_rr, state, scratch, res_dev, stage};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        prof.release_events();
        if (res.st) recycle_resources(device, res);
    }

    void spmv(const Csr& M, const T* xin, T* out, T a, const T* v0, T b0, const T* v1, T b1, const CgState* stt,
              const T* dotv = nullptr, double* partial = nullptr) {
        if (M.nblocks <= 0) return;
        if (M.blocked) {
            spmv_blk(M, xin, stt);
            hipLaunchKernelGGL((k_spmv_combine<T>), dim3((M.nrows + 255) / 256), dim3(256), 0, st, M.nrows, static_cast<const T*>(M.partial), M.nblk, (int64_t)M.nrows, a,
                               (const T*)nullptr, 0, (int64_t)0, T(0), v0, b0, v1, b1, out, dotv, partial, stt);
            return;
        }
        hipLaunchKernelGGL((k_spmv_stream<T>), dim3(M.nblocks), dim3(256), 0, st, M.rb, M.rp, M.ci, static_cast<const T*>(M.va), xin, out, a,
                           v0, b0, v1, b1, dotv, partial, stt);
    }
    // partial[b][row] of M * x
    void spmv_blk(const Csr& M, const T* xin, const CgState* stt, CgFuse<T> fu = CgFuse<T>()) {
        const LaunchTiming lt = g_launch_timing;   // profiled launch: the dispatch's own begin / end timestamps (qps_kernels.h)
        g_launch_timing = LaunchTiming();
        if (M.sell) {
            const SellDims sd{M.nrows, M.ncols, M.nsl, M.wpb, M.win, M.nwin, M.staged};
            using CV = typename SellOf<T>::CV; using V = typename VecOf<T>::type;
            if (lt.start) hipExtLaunchKernelGGL((k_spmv_sell<T>), dim3(M.wpb, M.nblk), dim3(BTHREADS), 0, st, lt.start, lt.stop, 0, sd, M.wg_ptr, M.sl_off, M.sl_perm,
                                                static_cast<const CV*>(M.s_cols), static_cast<const V*>(M.s_vals), M.bci, static_cast<const T*>(M.bva), M.lr_ptr, M.lr_desc, xin,
                                                static_cast<T*>(M.partial), stt, fu);
            else hipLaunchKernelGGL((k_spmv_sell<T>), dim3(M.wpb, M.nblk), dim3(BTHREADS), 0, st, sd, M.wg_ptr, M.sl_off, M.sl_perm, static_cast<const CV*>(M.s_cols),
                                    static_cast<const V*>(M.s_vals), M.bci, static_cast<const T*>(M.bva), M.lr_ptr, M.lr_desc, xin, static_cast<T*>(M.partial), stt, fu);
            return;
        }
#define QPS_BLK(LPR)                                                                                                                        \
        do {                                                                                                                                \
            if (lt.start) hipExtLaunchKernelGGL((k_spmv_blk<T, LPR>), dim3(M.wpb, M.nblk), dim3(BTHREADS), 0, st, lt.start, lt.stop, 0, M.nrows, M.ncols, M.task_ptr, \
                                                M.tasks, M.per, M.brp, M.bci, static_cast<const T*>(M.bva), M.lr_ptr, M.lr_desc, xin, static_cast<T*>(M.partial), stt, fu); \
            else hipLaunchKernelGGL((k_spmv_blk<T, LPR>), dim3(M.wpb, M.nblk), dim3(BTHREADS), 0, st, M.nrows, M.ncols, M.task_ptr, M.tasks, M.per, M.brp, M.bci, \
                                    static_cast<const T*>(M.bva), M.lr_ptr, M.lr_desc, xin, static_cast<T*>(M.partial), stt, fu);             \
        } while (0)
        if (M.lpr4) QPS_BLK(4); else QPS_BLK(8);
#undef QPS_BLK
    }
    int dot_parts(const Csr& M) const { return M.blocked ? (M.nrows + 255) / 256 : M.nblocks; }
    // c = P u + rho A'(A u) + sigma u (LinearSystemSolvers.jl:152-157) as three streamed SpMVs; optional partials of dot(u, c)
    void op_reduced(const T* uin, T* cout, double rho, double sigma, double* partial, const CgState* stt) {
        if (m > 0 && P.blocked && A.blocked && At.blocked) {
            // three blocked products; P u and A'(A u) share one combine launch that also forms the dot partials
            spmv_blk(P, uin, stt);
            spmv(A, uin, tm, T(1), nullptr, T(0), nullptr, T(0), stt);
            spmv_blk(At, tm, stt);
            hipLaunchKernelGGL((k_spmv_combine<T>), dim3(nb_n), dim3(256), 0, st, (int)n, static_cast<const T*>(P.partial), P.nblk, (int64_t)n, T(1),
                               static_cast<const T*>(At.partial), At.nblk, (int64_t)n, (T)rho, uin, (T)sigma, (const T*)nullptr, T(0), cout, partial ? uin : nullptr, partial, stt);
            return;
        }
        spmv(P, uin, cout, T(1), uin, (T)sigma, nullptr, T(0), stt, (m > 0) ? nullptr : uin, (m > 0) ? nullptr : partial);
        if (m > 0) {
            spmv(A, uin, tm, T(1), nullptr, T(0), nullptr, T(0), stt);
            spmv(At, tm, cout, (T)rho, cout, T(1), nullptr, T(0), stt, uin, partial);
        }
    }
    int op_parts() const { return (m > 0 && P.blocked && A.blocked && At.blocked) ? nb_n : (m > 0 ? dot_parts(At) : dot_parts(P)); }

    // IterativeSolvers.cg!(xx, Op, tt; abstol = eps_pcg, maxiter = itr_pcg), xx warm started
    int cg(double rho, double sigma) {
        CgState* slot[2] = {state, state + 1};
        int cur = 0;
        // c = mL u: ONE product with the explicit matrix (ItrSolCg!, LinearSystemSolvers.jl:137) or the three of the matrix-free operator (:152-157)
        auto apply_op = [&](const T* uin, T* cout, double* partial, const CgState* stt) {
            if (cg_explicit) spmv(Lm, uin, cout, T(1), nullptr, T(0), nullptr, T(0), stt, partial ? uin : nullptr, partial);
            else op_reduced(uin, cout, rho, sigma, partial, stt);
        };
        const int parts_op = cg_explicit ? dot_parts(Lm) : op_parts();
        const bool fused_pa = PA.blocked && !cg_explicit;
        apply_op(xx, cc, nullptr, nullptr);
        T* ub[2] = {cu, cu2}; int ui = 0;                            // ub[ui] = current direction u
        hipLaunchKernelGGL((k_cg_init<T>), dim3(nb_n), dim3(256), 0, st, (int)n, tt, cc, cr, ub[ui], part_rr);
        hipLaunchKernelGGL(k_cg_init_final, dim3(1), dim3(256), 0, st, nb_n, part_rr, slot[cur], eps_pcg,
                           sizeof(T) == 8 ? 1.4901161193847656e-08 : 3.4526698300124393e-04, itr_pcg);
        // Launch batches: the first one is the previous call's iteration count (consecutive ADMM iterations need nearly the same number); every further
        // one is what the measured contraction says is left (res_end / res_init over the iterations run so far), at least one and never more than doubles
        // the total.  An iteration enqueued past convergence still costs four dispatches that return at the done flag: with "last + 1, then double" a
        // sixth of the product dispatches of BASELINE config 3 were such no-ops (profiles/r03_z_bench_c3_kernel_stats.csv).
        HIPC(hipMemcpyAsync(state_host + 1, slot[cur], sizeof(CgState), hipMemcpyDeviceToHost, st));   // the initial residual, read at the first synchronisation
        int launched = 0, batch = std::max(1, std::min(last_cg, 64));
        for (;;) {
            for (int b = 0; b < batch; ++b) {
                ProfScope ps(prof, cat_op, 2);
                // fold the previous iteration's ||r||^2, publish the scalars into the other slot, u = r + beta u
                if (fused_pa) {
                    // [P; A] u with u = r + beta u_old formed while the x blocks are loaded; then A'(A u); ONE combine for c and dot(u, c)
                    CgFuse<T> fu; fu.r = cr; fu.uold = ub[ui]; fu.unew = ub[ui ^ 1]; fu.part_rr = part_rr; fu.nparts = nb_n;
                    fu.cur = slot[cur]; fu.nxt = slot[cur ^ 1]; fu.first = b == 0 ? 1 : 0;
                    // level 1: the two products of the first CG iteration of every 8th cg() call are timed by their own dispatch timestamps
                    const bool sample = prof.level == 2 || (prof.level == 1 && b == 0 && launched == 0 && prof_calls % 8 == 0);
                    { ProfLaunchScope ps2(prof, cat_pa, sample ? 1 : 3); spmv_blk(PA, cr, nullptr, fu); }
                    cur ^= 1; ui ^= 1;
                    const T* pp = static_cast<const T*>(PA.partial);
                    const int nb_m = (int)((m + 255) / 256);
                    hipLaunchKernelGGL((k_cg_combine_pa<T>), dim3((unsigned)(nb_n + nb_m)), dim3(256), 0, st, (int)n, (int)m, nb_n, pp, PA.nblk, (int64_t)(n + m), ub[ui],
                                       (T)sigma, (T)rho, cc, tm, part_uc, slot[cur]);                       // cc = P u, tm = A u, partials of dot(u, c)
                    { ProfLaunchScope ps2(prof, cat_at, sample ? 1 : 3); spmv_blk(At, tm, slot[cur]); }
                    hipLaunchKernelGGL((k_cg_combine_update<T>), dim3(nb_n), dim3(256), 0, st, (int)n, cc, static_cast<const T*>(At.partial), At.nblk, (int64_t)n, (T)rho,
                                       (T)sigma, ub[ui], nb_n + nb_m, part_uc, xx, cr, part_rr, slot[cur]);
                    continue;
                } else {
                    hipLaunchKernelGGL((k_cg_next_u<T>), dim3(nb_n), dim3(256), 0, st, (int)n, nb_n, part_rr, cr, ub[ui], slot[cur], slot[cur ^ 1], b == 0 ? 1 : 0);
                    cur ^= 1;
                    apply_op(ub[ui], cc, part_uc, slot[cur]);
                }
                hipLaunchKernelGGL((k_cg_update_xr<T>), dim3(nb_n), dim3(256), 0, st, (int)n, parts_op, part_uc, ub[ui], cc, xx, cr, part_rr, slot[cur]);
            }
            hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(256), 0, st, nb_n, part_rr, slot[cur]);   // fold the last iteration of the batch
            launched += batch;
            HIPC(hipMemcpyAsync(state_host, slot[cur], sizeof(CgState), hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            prof.harvest();
            if (state_host->done || launched >= itr_pcg) break;
            {
                const double r0 = std::sqrt(state_host[1].res2), r1 = std::sqrt(state_host->res2), tol = state_host->tol;
                int left = std::max(launched, 1);                                                   // no usable rate: as many again
                if (r1 > 0 && r0 > r1 && tol > 0 && r1 > tol) {
                    const double per_it = std::log(r0 / r1) / std::max(1, state_host->iters);       // contraction per iteration so far
                    left = (int)std::ceil(std::log(r1 / tol) / per_it);
                }
                batch = std::max(1, std::min(std::min(std::max(left, 2), std::max(launched, 2)), std::min(64, itr_pcg - launched)));   // (two at least: a synchronisation costs about one iteration)
            }
        }
        last_cg = state_host->iters;
        ++prof_calls;
        cg_total += state_host->iters;
        return state_host->iters;
    }

    // LinOpCg! body (LinearSystemSolvers.jl:176-181), or the direct KKT plugins' Sol! body (:37-40)
    void linear_solve(double rho, double sigma) {
        if (plugin_kind == QPS_LINSYS_KKT_LDL) {
            ProfScope ps(prof, cat_ldl, 2);
            ldl->solve(x, q, z, y, rho, sigma, xx, zz);
            return;
        }
        if (m > 0) hipLaunchKernelGGL((k_axpby<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (int)m, (T)rho, z, T(-1), y, w);   // :176
        {
            ProfScope ps(prof, cat_spmv, 2);
            if (m > 0) spmv(At, w, tt, T(1), x, (T)sigma, q, T(-1), nullptr);                        // :177-178
            else hipLaunchKernelGGL((k_axpby<T>), dim3(nb_n), dim3(256), 0, st, (int)n, (T)sigma, x, T(-1), q, tt);
        }
        cg(rho, sigma);                                                                             // :179
        {
            ProfScope ps(prof, cat_spmv, 2);
            if (m > 0) spmv(A, xx, zz, T(1), nullptr, T(0), nullptr, T(0), nullptr);                 // :181
        }
    }

    void solve(double* xh, const qps_params& p, qps_info* info) override {
        HIPC(hipSetDevice(device));
        const double t0 = now_s();
        double rho = p.rho; const double sigma = p.sigma, alpha = p.alpha;
        eps_pcg = p.epsPcg; itr_pcg = p.numItrPcg;
        const double epsAdmm = std::fmin(p.epsAbs, p.epsRel) * 1e-2;
        int convFlag = QPS_CONV_NUM_ITR;
        if (p.linsys == QPS_LINSYS_CHOLESKY) throw QpsError(QPS_ERR_UNSUPPORTED, "CSR handles offer QPS_LINSYS_CG, QPS_LINSYS_CG_EXPLICIT and QPS_LINSYS_KKT_LDL (create with dense_path=1 for the reduced Cholesky path)");
        plugin_kind = resolve_kind(p.linsys, rho, sigma, !p.reuseFactor);                           // SolveQuadraticProgram.jl:36 LinSysSolInit (incl. the factorisation)
        upload_vec(xh, x, n);
        const size_t nb_ = sizeof(T) * (size_t)(n + 64), mb_ = sizeof(T) * (size_t)(m + 64);
        HIPC(hipMemsetAsync(xp, 0, nb_, st)); HIPC(hipMemsetAsync(xx, 0, nb_, st));                 // LinOpCgInit: vXX = zeros (:147)
        HIPC(hipMemsetAsync(z, 0, mb_, st)); HIPC(hipMemsetAsync(y, 0, mb_, st)); HIPC(hipMemsetAsync(zp, 0, mb_, st));
        HIPC(hipStreamSynchronize(st));
        const double t1 = now_s();
        double rhorho = rho; int ii = 0, nref = 0; double resP = NAN, resD = NAN, tref = 0;
        cg_total = 0; last_cg = 4;
        const int NPv = (int)n, MPv = (int)m;
        static const int graph_env = [] { const char* e = getenv("QPS_GRAPH"); return e ? atoi(e) : -1; }();
        const bool use_graph = plugin_kind == QPS_LINSYS_KKT_LDL && prof.level == 0 && graph_env != 0;
        bool ldl_rhs_ready = false;   // the fused post/update launch leaves the next right-hand side behind; stale after a rho switch
        for (ii = 1; ii <= p.numIterations; ++ii) {
            if (p.adptRho && ((rhorho * p.fctrRho < rho) || (rhorho > p.fctrRho * rho))) {
                rho = rhorho; ++nref;                                                               // matrix-free CG: nothing to rebuild
                if (plugin_kind == QPS_LINSYS_CG && cg_explicit) { const double ta = now_s(); reduced_values(rho, sigma); tref += now_s() - ta; }   // changedΡ: LinearSystemSolvers.jl:127-129
                if (plugin_kind == QPS_LINSYS_KKT_LDL) { const double ta = now_s(); ldl_prepare(rho, sigma, true); tref += now_s() - ta; ldl_rhs_ready = false; }   // changedΡ: numeric refactor only
            }
            if (plugin_kind == QPS_LINSYS_KKT_LDL && prof.level < 2) {
                // direct plugin: sweeps + ONE launch for nu -> z~, the x / z / y updates and the next right-hand side (k_ldl_post_update)
                if (use_graph && ldl_rhs_ready && ii % p.numItrConv != 0) {
                    hipGraphExec_t ge = iter_graph(rho, sigma, alpha);
                    if (ge) { HIPC(hipGraphLaunch(ge, st)); continue; }                             // a plain iteration: same kernels, same order
                }
                ldl->iterate(x, xp, q, z, zp, y, l, u, alpha, rho, sigma, ldl_rhs_ready);
                ldl_rhs_ready = true;
            } else {
            linear_solve(rho, sigma);
            {
                ProfScope ps(prof, cat_vec, 2);
                admm_update<T>(st, NPv, MPv, xx, zz, x, xp, z, zp, y, l, u, (T)alpha, (T)rho);
            }
            }
            if (ii % p.numItrConv == 0) {
                {
                    ProfScope ps(prof, cat_chk, 2);
                    if (m > 0) spmv(A, x, Ax, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                    spmv(P, x, Px, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                    if (m > 0) spmv(At, y, Aty, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                    else HIPC(hipMemsetAsync(Aty, 0, nb_, st));
                    CheckScalars cs{p.epsAbs, p.epsRel, epsAdmm, rho, rhorho, p.adptRho, convFlag};
                    check_convergence<T>(st, (int)n, (int)m, Ax, Px, Aty, q, x, xp, z, zp, scratch, res_dev, cs);
                }
                HIPC(hipMemcpyAsync(res_host, res_dev, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                prof.harvest();
                resP = res_host[0]; resD = res_host[1]; rhorho = res_host[4]; convFlag = (int)res_host[5];
                if (convFlag != QPS_CONV_NUM_ITR) break;
            }
        }
        HIPC(hipStreamSynchronize(st));
        prof.harvest();
        const double t2 = now_s();
        PolishReport pr;
        if (p.polish) polish_device(p, &pr);
        download_vec(x, xh, n);
        if (info) {
            info->convFlag = convFlag; info->iterations = ii > p.numIterations ? p.numIterations : ii;
            info->numRefactor = nref; info->cgIterations = (int)cg_total; info->rhoFinal = rho; info->rhoProposed = rhorho;
            info->resPrim = resP; info->resDual = resD; info->tSetup = t1 - t0; info->tLoop = t2 - t1; info->tRefactor = tref;
            info->trsvBlock = 0; info->sweepVariant = 0; info->sweepGaveUp = 0; info->cgExplicit = (plugin_kind == QPS_LINSYS_CG && cg_explicit) ? 1 : 0;
            info->polishFlag = pr.flag; info->polishIterations = pr.minresIterations; info->tPolish = pr.seconds;
        }
    }
    void get_dual(double* zh, double* yh) override {
        HIPC(hipSetDevice(device));
        if (zh) download_vec(z, zh, m);
        if (yh) download_vec(y, yh, m);
    }
    // SolveQuadraticProgram.m:289-325 with the products of K as SpMVs (x block of length n, multiplier block of length m)
    void polish_device(const qps_params& p, PolishReport* pr) {
        PolishProduct<T> kmat = [&](const T* v, T delta, T* out, const T* mask, T* scratch) {
            spmv(P, v, out, T(1), v, delta, nullptr, T(0), nullptr);                                // P v_x + delta v_x
            if (m <= 0) return;
            spmv(A, v, out + n, T(1), nullptr, T(0), nullptr, T(0), nullptr);                       // A v_x
            polish_mask_rows<T>(st, (int)m, mask, v + n, delta, out + n, scratch);                  // mask . - delta mask v_lambda; scratch = mask v_lambda
            spmv(At, scratch, out, T(1), out, T(1), nullptr, T(0), nullptr);                        // + A'(mask v_lambda)
        };
        polish_with<T>(st, n, m, (int)n, (int)m, q, l, u, y, x, p, pr, kmat);
    }
    void polish(double* xh, const double* yh, const qps_params& p, qps_polish_report* rep) override {
        HIPC(hipSetDevice(device));
        upload_vec(xh, x, n); upload_vec(yh, y, m);
        PolishReport pr;
        polish_device(p, &pr);
        download_vec(x, xh, n);
        if (rep) {
            rep->flag = pr.flag; rep->refinements = pr.refinements; rep->minresIterations = pr.minresIterations;
            rep->numActiveLower = pr.numLower; rep->numActiveUpper = pr.numUpper; rep->reserved0 = 0; rep->relres = pr.relres; rep->seconds = pr.seconds;
        }
    }
    // qps_operator_apply: one application of P / A / A' / [P; A] / the reduced operator (LinearSystemSolvers.jl:152-157) through the SpMV kernels the loop
    // uses for them; work buffers only (cu, cc: n; w, tm: m), the solver state stays
    void operator_apply(int op, const double* in, double* out, double rho, double sigma) override {
        HIPC(hipSetDevice(device));
        if (op == QPS_OP_AT) {
            upload_vec(in, w, m);
            if (m > 0) spmv(At, w, cc, T(1), nullptr, T(0), nullptr, T(0), nullptr); else HIPC(hipMemsetAsync(cc, 0, sizeof(T) * (size_t)(n + 64), st));
            download_vec(cc, out, n);
            return;
        }
        upload_vec(in, cu, n);
        if (op == QPS_OP_P) { spmv(P, cu, cc, T(1), nullptr, T(0), nullptr, T(0), nullptr); download_vec(cc, out, n); }
        else if (op == QPS_OP_A) { if (m > 0) spmv(A, cu, tm, T(1), nullptr, T(0), nullptr, T(0), nullptr); download_vec(tm, out, m); }
        else if (op == QPS_OP_PA) {
            if (PA.blocked) {                                                                       // the stacked product of a CG iteration, then its blocks added up
                spmv_blk(PA, cu, nullptr);
                T* both = dalloc<T>(n + m + 64, st);
                hipLaunchKernelGGL((k_spmv_combine<T>), dim3((unsigned)((n + m + 255) / 256)), dim3(256), 0, st, (int)(n + m), static_cast<const T*>(PA.partial), PA.nblk, (int64_t)(n + m),
                                   T(1), (const T*)nullptr, 0, (int64_t)0, T(0), (const T*)nullptr, T(0), (const T*)nullptr, T(0), both, (const T*)nullptr, (double*)nullptr, (const CgState*)nullptr);
                download_vec(both, out, n); download_vec(both + n, out + n, m);
                (void)hipFree(both);
            } else {
                spmv(P, cu, cc, T(1), nullptr, T(0), nullptr, T(0), nullptr); download_vec(cc, out, n);
                if (m > 0) spmv(A, cu, tm, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                download_vec(tm, out + n, m);
            }
        } else if (op == QPS_OP_REDUCED) { op_reduced(cu, cc, rho, sigma, nullptr, nullptr); download_vec(cc, out, n); }
        else throw QpsError(QPS_ERR_BAD_ARGUMENT, "unknown qps_operator_kind");
    }
    void linsys_set_cg(double eps, int itr) override { eps_pcg = eps; itr_pcg = itr; }             // LinOpCg!(...; ϵPcg, numItrPcg): LinearSystemSolvers.jl:164
    void linsys_init(double rho, double sigma, int linsys, int) override {
        HIPC(hipSetDevice(device));
        if (linsys != QPS_LINSYS_AUTO && linsys != QPS_LINSYS_CG && linsys != QPS_LINSYS_KKT_LDL && linsys != QPS_LINSYS_CG_EXPLICIT)
            throw QpsError(QPS_ERR_UNSUPPORTED, "CSR handles offer QPS_LINSYS_CG, QPS_LINSYS_CG_EXPLICIT and QPS_LINSYS_KKT_LDL (create with dense_path=1 for the reduced Cholesky path)");
        plugin_kind = resolve_kind(linsys, rho, sigma, true);                                       // LinearSystemSolvers.jl:18 / :49 / :81
        if (plugin_kind == QPS_LINSYS_KKT_LDL) return;
        HIPC(hipMemsetAsync(xx, 0, sizeof(T) * (size_t)(n + 64), st));                              // LinOpCgInit (:147)
        cg_total = 0; last_cg = 4;
    }
    void linsys_solve(const double* xh, const double* zh, const double* yh, double rho, double sigma, int changed, double* xxh, double* zzh) override {
        HIPC(hipSetDevice(device));
        if (plugin_kind == QPS_LINSYS_KKT_LDL) {
            if (!ldl) throw QpsError(QPS_ERR_BAD_ARGUMENT, "qps_linsys_solve called before qps_linsys_init");
            if (changed) ldl_prepare(rho, sigma, true);                                             // :30-32 / :61-63 / :93-95
        }
        if (plugin_kind == QPS_LINSYS_CG && cg_explicit) reduced_values(rho, sigma);                 // :127-129 (a no-op while rho and sigma are the cached ones)
        upload_vec(xh, x, n); upload_vec(zh, z, m); upload_vec(yh, y, m);
        linear_solve(rho, sigma);
        HIPC(hipStreamSynchronize(st));
        download_vec(xx, xxh, n); download_vec(zz, zzh, m);
    }
};


// =================================================================================================================
// SparseProxQP (ProxQP.jl:71, :95-115; sparse branches :184-190, :201-206, :335-372): the second solver form on SparseMatrixCSC inputs.
//
// The reference keeps M = P + rho (A'A + C'C) + sigma I as a sparse matrix with a frozen pattern (AlignSparsePattern :351-372, diagonal
// positions :335-349), rewrites its values on a rho update (UpdateM! :184-190) and re-factorises with the pattern-reusing `cholesky!`
// (:201-206).  Here the same linear system is solved in its KKT form
//     [P + sigma I   G' ; G   -I / rho] [x ; nu] = [sigma x - q ; h - dual / rho],     G = [A; C],  h = [b ; d - s],  dual = [y ; z]
// (eliminating nu gives exactly M x = sigma x - q + G'[rho b - y ; rho (d - s) - z], CalculateRhs! :208-219) with the sparse L D L' plugin
// of the main path: ordering + symbolic factor once per handle, a rho update is numeric only -- the pattern-reusing re-factorisation --
// and G x, which the row updates (:227-249) need, comes back with the solve (z~ = z + (nu - y) / rho = G x).  G'G is never formed: the
// reduced matrix of a sparse G is much denser than [.. G'; G ..].  CheckConvergence! (:252-298) runs on the CSR SpMVs of the main path.
// The six-argument constructor's start (:95-115: [P A'; A 0] [x; y] = [-q; b]) uses a second factor, of [P A'; A -delta I], and iterative
// refinement against the unperturbed system.
// =================================================================================================================
template <typename T> struct SparseProxQpSolver : ProxQpBase {
    std::unique_ptr<SparseSolver<T>> ss;          // P, G = [A; C] as CSR (G, G'), the L D L' plugin, the staging buffers
    hipStream_t st = nullptr;
    int mtot = 0;
    T *g = nullptr, *dual = nullptr, *slack = nullptr, *hvec = nullptr, *x = nullptr, *xx = nullptr, *v = nullptr, *de = nullptr, *di = nullptr;
    T *X1 = nullptr, *X2 = nullptr, *X3 = nullptr, *r1 = nullptr, *r2 = nullptr, *dx = nullptr, *dnu = nullptr, *zero_m = nullptr;
    unsigned long long* slots = nullptr; unsigned long long* slots_host = nullptr;
    // canonical 0-based host copies of P and A for the factor of the initialisation (dropped after init_kkt / set_state)
    std::vector<int64_t> kPcp, kPri, kAcp, kAri; std::vector<double> kPnz, kAnz;

    static dim3 g1(int64_t c) { return dim3((unsigned)((std::max<int64_t>(c, 1) + 255) / 256)); }

    SparseProxQpSolver(int dev, int64_t n_, int64_t me_, int64_t mi_, int dt, const int64_t* Pcp, const int64_t* Pri, const double* Pnz, const double* qh,
                       const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* bh, const int64_t* Ccp, const int64_t* Cri, const double* Cnz,
                       const double* dh, int base) {
        device = dev; n = n_; me = me_; mi = mi_; mtot = (int)(me + mi);
        // G = [A; C] column by column (rows of C shifted by numEq), 0-based; P re-based
        std::vector<int64_t> Gcp(n + 1, 0), Gri, P0cp(n + 1), P0ri((size_t)(Pcp[n] - base));
        std::vector<double> Gnz;
        for (int64_t j = 0; j <= n; ++j) P0cp[j] = Pcp[j] - base;
        for (size_t k = 0; k < P0ri.size(); ++k) P0ri[k] = Pri[k] - base;
        for (int64_t j = 0; j < n; ++j) {
            if (me > 0) for (int64_t k = Acp[j] - base; k < Acp[j + 1] - base; ++k) { Gri.push_back(Ari[k] - base); Gnz.push_back(Anz[k]); }
            if (mi > 0) for (int64_t k = Ccp[j] - base; k < Ccp[j + 1] - base; ++k) { Gri.push_back(Cri[k] - base + me); Gnz.push_back(Cnz[k]); }
            Gcp[j + 1] = (int64_t)Gri.size();
        }
        std::vector<double> zeros((size_t)std::max(mtot, 1), 0.0), gh((size_t)std::max(mtot, 1), 0.0);
        for (int64_t i = 0; i < me; ++i) gh[i] = bh[i];
        for (int64_t i = 0; i < mi; ++i) gh[me + i] = dh[i];
        static const int64_t dummy_i[1] = {0}; static const double dummy_d[1] = {0.0};
        ss.reset(new SparseSolver<T>(dev, n, mtot, dt, P0cp.data(), P0ri.empty() ? dummy_i : P0ri.data(), Pnz, Gcp.data(), Gri.empty() ? dummy_i : Gri.data(),
                                     Gnz.empty() ? dummy_d : Gnz.data(), qh, zeros.data(), zeros.data(), 0));
        st = ss->st;
        const int64_t nn = n + 64, mm = mtot + 64;
        g = dalloc<T>(mm, st); dual = dalloc<T>(mm, st); slack = dalloc<T>(mm, st); hvec = dalloc<T>(mm, st); v = dalloc<T>(mm, st); de = dalloc<T>(mm, st); di = dalloc<T>(mm, st);
        r2 = dalloc<T>(mm, st); dnu = dalloc<T>(mm, st); zero_m = dalloc<T>(mm, st);
        x = dalloc<T>(nn, st); xx = dalloc<T>(nn, st); X1 = dalloc<T>(nn, st); X2 = dalloc<T>(nn, st); X3 = dalloc<T>(nn, st); r1 = dalloc<T>(nn, st); dx = dalloc<T>(nn, st);
        slots = dalloc<unsigned long long>(16, st);
        slots_host = reinterpret_cast<unsigned long long*>(ss->res.pinned);
        ss->upload_vec(gh.data(), g, mtot);
        layout::canonical_csc(n, P0cp.data(), P0ri.empty() ? dummy_i : P0ri.data(), Pnz, 0, kPcp, kPri, kPnz);
        if (me > 0) layout::canonical_csc(n, Acp, Ari, Anz, base, kAcp, kAri, kAnz);
        else kAcp.assign(n + 1, 0);
    }
    ~SparseProxQpSolver() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        void* ptrs[] = {g, dual, slack, hvec, x, xx, v, de, di, X1, X2, X3, r1, r2, dx, dnu, zero_m, slots};
        for (void* p_ : ptrs) if (p_) (void)hipFree(p_);
    }
    void drop_init_copies() {
        for (auto* vv : {&kPcp, &kPri, &kAcp, &kAri}) std::vector<int64_t>().swap(*vv);
        std::vector<double>().swap(kPnz); std::vector<double>().swap(kAnz);
    }
    void set_state(const double* xh, const double* yh, const double* zh, const double* sh) override {
        HIPC(hipSetDevice(device));
        ss->upload_vec(xh, x, n); ss->upload_vec(yh, dual, me); ss->upload_vec(zh, dual + me, mi);
        HIPC(hipMemsetAsync(slack, 0, sizeof(T) * (size_t)(mtot + 64), st));
        ss->upload_vec(sh, slack + me, mi);
    }
    void get_state(double* xh, double* yh, double* zh, double* sh) override {
        HIPC(hipSetDevice(device));
        if (xh) ss->download_vec(x, xh, n);
        if (yh) ss->download_vec(dual, yh, me);
        if (zh) ss->download_vec(dual + me, zh, mi);
        if (sh) ss->download_vec(slack + me, sh, mi);
    }
    void axpby(int64_t c, T a, const T* xa, T b, const T* ya, T* out) { if (c > 0) hipLaunchKernelGGL((k_axpby<T>), g1(c), dim3(256), 0, st, (int)c, a, xa, b, ya, out); }

    // ProxQP.jl:95-115: [P A'; A 0] [x; y] = [-q; b], s = max(d - C x, 0), z = 0
    void init_kkt() override {
        HIPC(hipSetDevice(device));
        if (kPcp.empty()) throw QpsError(QPS_ERR_BAD_ARGUMENT, "qps_proxqp_init_kkt: the initialisation data of this handle was released (call it once, before set_state)");
        const double delta = sizeof(T) == 8 ? 1e-8 : 1e-3;
        LdlSymbolic sym;
        try { sym = ldl_analyze((int)n, (int)me, kPcp.data(), kPri.data(), kAcp.data(), kAri.data(), 0, 8192, 64, 4096); }
        catch (const std::runtime_error& e) { throw QpsError(QPS_ERR_UNSUPPORTED, e.what()); }
        std::unique_ptr<SparseLdl<T>> kkt = make_sparse_ldl<T>(st, std::move(sym), kPnz.data(), (int64_t)kPnz.size(), kAnz.data(), (int64_t)kAnz.size());
        // [P + delta I, A'; A, -delta I]: quasi-definite for every positive SEMI-definite P, so a singular P with a non-singular KKT matrix -- which the
        // reference's `mK \\ vR` (an LU, ProxQP.jl:103-106) accepts -- factorises too; the refinement below runs against the UNSHIFTED system
        kkt->factorize(1.0 / delta, delta);
        HIPC(hipMemsetAsync(x, 0, sizeof(T) * (size_t)(n + 64), st));
        HIPC(hipMemsetAsync(dual, 0, sizeof(T) * (size_t)(mtot + 64), st));
        HIPC(hipMemsetAsync(de, 0, sizeof(T) * (size_t)(mtot + 64), st));
        // Iterative refinement against the UNSHIFTED system until the residual stops falling (at most 50 steps).  In the directions of P that A does not pin
        // a step contracts the error only by delta / (lambda + delta): with fp32's delta = 1e-3 and lambda_min(P) = 5e-4 that is 0.67 per step, which a fixed
        // count of 7 left at 6e-2 and the handle was refused although the reference's `mK \ vR` (ProxQP.jl:103-106) solves the system.  The start need not be
        // exact (ProxQP iterates from it): it is refused only when the residual stops falling above the tolerance -- a singular [P A'; A 0], on which the reference's
        // backslash fails as well: the inconsistent part of the right-hand side stays while |x| grows by ~1 / delta per step -- or turns non-finite.
        std::vector<double> hr((size_t)n), hq((size_t)n), hr2((size_t)std::max<int64_t>(me, 1), 0.0), hb((size_t)std::max<int64_t>(me, 1), 0.0);
        auto amax = [](const std::vector<double>& a_) { double m_ = 0; for (double t : a_) { if (std::isnan(t)) return (double)INFINITY; m_ = std::fmax(m_, std::fabs(t)); } return m_; };
        ss->download_vec(ss->q, hq.data(), n);
        if (me > 0) ss->download_vec(g, hb.data(), me);
        // against the right-hand side ONLY: a singular system answers the shifted solve with |x| ~ 1 / delta, and a scale that included |x| would
        // normalise the residual it leaves behind away
        const double scale = std::fmax(1.0, std::fmax(amax(hq), amax(hb)));
        const double target = sizeof(T) == 8 ? 1e-12 : 1e-6;
        double res0 = INFINITY, best = INFINITY, res = INFINITY; int stalled = 0;
        for (int it = 0; it <= 50; ++it) {
            // residual of the UNPERTURBED system: r1 = -q - P x - A'y, r2 = b - A x (the rows of C carry zeros in `de`, so G'de = A'y)
            ss->spmv(ss->P, x, r1, T(-1), ss->q, T(-1), nullptr, T(0), nullptr);
            if (me > 0) {
                HIPC(hipMemcpyAsync(de, dual, sizeof(T) * (size_t)me, hipMemcpyDeviceToDevice, st));
                ss->spmv(ss->At, de, r1, T(-1), r1, T(1), nullptr, T(0), nullptr);
                ss->spmv(ss->A, x, v, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                axpby(me, T(1), g, T(-1), v, r2);
            }
            ss->download_vec(r1, hr.data(), n);
            if (me > 0) ss->download_vec(r2, hr2.data(), me);
            res = std::fmax(amax(hr), amax(hr2)) / scale;
            if (it == 0) res0 = res;
            if (!std::isfinite(res)) break;
            if (res <= target || it == 50) break;
            if (res < 0.99 * best) stalled = 0; else if (++stalled >= 3) break;  // three steps without a 1 % gain: as good as this factor gets
            best = std::fmin(best, res);
            kkt->solve_raw(r1, r2, dx, dnu);
            axpby(n, T(1), x, T(1), dx, x);
            axpby(me, T(1), dual, T(1), dnu, dual);
        }
        {
            const double accept = sizeof(T) == 8 ? 1e-6 : 1e-2;
            // refused: non-finite, or a residual that has stopped falling ABOVE the tolerance -- the floor an inconsistent (singular) system leaves, |r| = the part of
            // [-q; b] outside the range; a residual that is still falling after 50 steps is an approximate start, and a start is all ProxQP needs
            if (!std::isfinite(res) || (res > accept && stalled >= 3)) {
                char bmsg[320]; snprintf(bmsg, sizeof bmsg, "qps_proxqp_init_kkt: the iterative refinement of [P A'; A 0] [x; y] = [-q; b] did not converge (relative residual %.3g after starting at %.3g): "
                                                          "the KKT matrix is singular or too ill-conditioned", res, res0);
                throw QpsError(QPS_ERR_FACTORIZATION, bmsg);
            }
        }
        if (mtot > 0) {
            ss->spmv(ss->A, x, v, T(1), nullptr, T(0), nullptr, T(0), nullptr);                    // G x
            hipLaunchKernelGGL((pqrows::k_pq_init_s<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, v, dual, slack);   // :110-111
        }
        HIPC(hipStreamSynchronize(st));
        drop_init_copies();
    }

    void solve(const qps_proxqp_params& p, qps_proxqp_report* rep) override {
        HIPC(hipSetDevice(device));
        drop_init_copies();
        double rho = p.rho; const double sigma = p.sigma;
        int converged = 0, conv_it = p.numIterations; double resP = INFINITY, resD = INFINITY, rho_rep = p.rho;
        ss->ldl_prepare(rho, sigma, true);                                                          // UpdateDecomposition! (ProxQP.jl:131, :201-206)
        const char* gxs = getenv("QPS_PROXQP_GX_SOLVE");                                            // read per solve
        const bool gx_product = !(gxs && gxs[0] == '1');
        for (int ii = 1; ii <= p.numIterations; ++ii) {                                             // :135
            const bool check = (ii % p.numItrConv == 0);
            if (mtot > 0) hipLaunchKernelGGL((pqrows::k_pq_h<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, slack, hvec);
            ss->ldl->solve(x, ss->q, hvec, dual, rho, sigma, xx, v);                                // CalculateRhs! + UpdateX! (:208-225)
            std::swap(x, xx);
            // `mA * vX`, `mC * vX` of the row updates (:230-248) by the product itself, as the reference forms them -- not the G x that falls out of the KKT solve
            // (same value up to the un-refined, un-pivoted solve's error).  The reference's check (:264-266) repeats the SAME product, so `C x - d + s` with
            // s = d - C x cancels EXACTLY on rows with z = 0; a primal residual of exactly 0 sends rho to its lower clamp (:281-283), a residual of 1e-16 to
            // rho * 1e-4: with G x from the solve in the updates and from the product in the check the two runs part for good (ProxQP fuzz, seed 43, case 184).
            if (mtot > 0 && gx_product) ss->spmv(ss->A, x, v, T(1), nullptr, T(0), nullptr, T(0), nullptr);
            if (mtot > 0) hipLaunchKernelGGL((pqrows::k_pq_update<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, v, dual, slack, (T)rho);   // :227-249
            if (check) {                                                                            // :151  CheckConvergence! :252-298
                ss->spmv(ss->P, x, X1, T(1), nullptr, T(0), nullptr, T(0), nullptr);                // :261
                HIPC(hipMemsetAsync(X2, 0, sizeof(T) * (size_t)n, st)); HIPC(hipMemsetAsync(X3, 0, sizeof(T) * (size_t)n, st));
                if (mtot > 0) {
                    hipLaunchKernelGGL((pqrows::k_pq_split<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, dual, de, di);
                    ss->spmv(ss->At, de, X2, T(1), nullptr, T(0), nullptr, T(0), nullptr);          // A'y (:262)
                    ss->spmv(ss->At, di, X3, T(1), nullptr, T(0), nullptr, T(0), nullptr);          // C'z (:263)
                }
                // mA * vX, mC * vX of CheckConvergence! (:264-265): the product the row updates have just used (QPS_PROXQP_GX_SOLVE=1: recomputed here -- the
                // solve's G x carries the error of the un-refined L D L' solve, which must not go into the reported primal residual)
                if (mtot > 0 && !gx_product) ss->spmv(ss->A, x, v, T(1), nullptr, T(0), nullptr, T(0), nullptr);
                HIPC(hipMemsetAsync(slots, 0, 16 * sizeof(unsigned long long), st));
                hipLaunchKernelGGL((pqrows::k_pq_norms<T>), dim3(64), dim3(256), 0, st, (int)n, (int)me, mtot, v, g, slack, X1, X2, X3, ss->q, slots);
                HIPC(hipMemcpyAsync(slots_host, slots, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                const pqrows::CheckOutcome co = pqrows::decide(slots_host, p, rho);                 // :266-294
                rho = co.rho; converged = co.converged ? 1 : 0; resP = co.resPrim; resD = co.resDual;
                if (converged) conv_it = ii;                                                        // :155-157 (no break)
                if (co.updated) { ss->ldl_prepare(rho, sigma, true); rho_rep = rho; }               // :159-165: UpdateM! + pattern-reusing cholesky!
            }
        }
        HIPC(hipStreamSynchronize(st));
        if (rep) { rep->converged = converged; rep->iterations = conv_it; rep->rho = rho_rep; rep->sigma = sigma; rep->resPrim = resP; rep->resDual = resD; }
    }
};

}  // namespace

ProxQpBase* make_proxqp_sparse(int device, int64_t n, int64_t me, int64_t mi, int dtype, const int64_t* Pcp, const int64_t* Pri, const double* Pnz, const double* q,
                               const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* b, const int64_t* Ccp, const int64_t* Cri, const double* Cnz,
                               const double* d, int index_base) {
    if (dtype == QPS_F64) return new SparseProxQpSolver<double>(device, n, me, mi, dtype, Pcp, Pri, Pnz, q, Acp, Ari, Anz, b, Ccp, Cri, Cnz, d, index_base);
    return new SparseProxQpSolver<float>(device, n, me, mi, dtype, Pcp, Pri, Pnz, q, Acp, Ari, Anz, b, Ccp, Cri, Cnz, d, index_base);
}

SolverBase* make_sparse_solver(int device, int64_t n, int64_t m, int dtype, const int64_t* Pcp, const int64_t* Pri,
                               const double* Pnz, const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* q,
                               const double* l, const double* u, int index_base) {
    if (dtype == QPS_F64) return new SparseSolver<double>(device, n, m, dtype, Pcp, Pri, Pnz, Acp, Ari, Anz, q, l, u, index_base);
    return new SparseSolver<float>(device, n, m, dtype, Pcp, Pri, Pnz, Acp, Ari, Anz, q, l, u, index_base);
}

}  // namespace qps

#ifdef QPS_SPMV_STAMPS
extern "C" __attribute__((visibility("default"))) int qps_debug_spmv_stamps(long long* out, int count, int filter_rows) {
    using namespace qps;
    // (called between blocking qps_solve calls: the handle's stream is idle, nothing to wait for here)
    if (count < 0) {   // clear the table
        static const std::vector<long long> zeros((size_t)STAMP_WGS * STAMP_SLOTS, 0);
        return hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_stamps), zeros.data(), sizeof(long long) * zeros.size()) == hipSuccess ? 0 : -4;
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_stamp_rows), &filter_rows, sizeof(int)) != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spmv_stamps), sizeof(long long) * (size_t)count) != hipSuccess) return -2;
    return 0;
}
#endif
