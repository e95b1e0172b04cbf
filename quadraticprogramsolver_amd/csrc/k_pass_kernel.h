// k_pass_kernel.h -- kernel template shared by k_pass.hip (ADMM rows) and k_pass_pq.hip (ProxQP rows): the fused single pass over A (SURVEY.md §7 hard part 2, §8d "A streamed once").
//
// Per ADMM iteration the reference touches A twice: z~ = A x~ (LinearSystemSolvers.jl:139) and, at the top of the next
// iteration, A'(rho z - y) (:134-135).  Row i of the second product needs only row i of the first, so one workgroup
// can hold a tile of R full rows of A in registers and do, per tile:
//     z~_i = A_i . x~                     (row dot, wave64 shuffle + LDS reduction)
//     z_i, y_i update                     (SolveQuadraticProgram.jl:59-61, by the thread that owns row i)
//     w_i = rho z_i - y_i                 (LinearSystemSolvers.jl:134)
//     acc  += A_i' w_i                    (column accumulation in registers, every thread owns NP/THREADS columns)
// A is read exactly once (algorithmic m*n*s bytes).  Each workgroup writes its column sums to one slab; colsum()
// adds the slabs in fixed order while forming the next right-hand side (deterministic, no float atomics).
// The x update (:56-57) rides along (workgroup 0; x is ping-ponged so no workgroup ever reads a buffer being written).
// CHECK variant (every numItrConv-th iteration): additionally A x_new (second dot on the same tile), A'y_new (second
// accumulator) and the primal-side inf-norms of CheckConvergence (:85,88,105), so the check re-reads neither A nor A'.
#pragma once
#include <cstdlib>

#include "qps_kernels.h"
#include "wave_reduce.h"
#include <hip/hip_ext.h>

namespace qps {

namespace {

__device__ __forceinline__ unsigned long long absbits(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__device__ __forceinline__ unsigned long long umax(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

template <typename T, int THREADS, int KC, int R, bool CHECK>
struct Pass {
    using V = typename VecOf<T>::type;
    static constexpr int VN = VecOf<T>::N;
    static constexpr int CHUNK = THREADS * VN;
    static constexpr int WAVES = THREADS / 64;

    static __device__ __forceinline__ void load_tile(V (&a)[R][KC], const T* __restrict__ A, int64_t ld, int row, int NP, int tid) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = tid * VN + k * CHUNK;
                if (c < NP) {
                    // A is streamed exactly once per iteration by exactly one CU: non-temporal, so that it does not evict the
                    // sweep triangle and the slabs from L2 / Infinity Cache (measured on C2: pass 49.9 -> 46.7 us, fused sweep
                    // 23.5 -> 21.8 us; a run-time selectable hint was slower than either fixed choice, fp32 is insensitive).
                    // (Unconditional, clamped loads -- which let the compiler count the loads in flight instead of waiting for all of them in
                    // front of every other tile, and do help the triangular sweeps -- make THIS kernel slower: C2 45.3 -> 58.4 us.  With both
                    // tiles' requests counted the FMAs are woven between 32 single-register waits; the coarse wait keeps the tile whole.)
                    typedef T NV __attribute__((ext_vector_type(VN)));
                    const NV t = __builtin_nontemporal_load(reinterpret_cast<const NV*>(A + (int64_t)(row + i) * ld + c));
                    T* p = reinterpret_cast<T*>(&a[i][k]);
#pragma unroll
                    for (int e = 0; e < VN; ++e) p[e] = t[e];
                }
                else {
                    T* p = reinterpret_cast<T*>(&a[i][k]);
#pragma unroll
                    for (int e = 0; e < VN; ++e) p[e] = T(0);
                }
            }
    }
};

template <typename T, int THREADS, int KC, int R, bool CHECK, int MODE>
__global__ __launch_bounds__(THREADS) void k_apass(const T* __restrict__ A, int64_t ld, int NP, int MP, int rows_per_wg,
                                                   const T* __restrict__ xx, const T* __restrict__ x_old, T* __restrict__ x_new,
                                                   T* __restrict__ z, T* __restrict__ y, const T* __restrict__ l,
                                                   const T* __restrict__ u, T alpha, T rho, T* __restrict__ part,
                                                   T* __restrict__ part2, int64_t part_ld, unsigned long long* __restrict__ slots,
                                                   PassBatch pb) {
    using PS = Pass<T, THREADS, KC, R, CHECK>;
    if (pb.count > 1 || pb.active) {                       // batched launch: blockIdx.y = QP index
        const int b = blockIdx.y;
        if (pb.active && !pb.active[b]) return;
        A += (int64_t)b * MP * ld; xx += (int64_t)b * NP; x_old += (int64_t)b * NP; x_new += (int64_t)b * NP;
        z += (int64_t)b * MP; y += (int64_t)b * MP; l += (int64_t)b * MP; u += (int64_t)b * MP;
        part += (int64_t)b * pb.slabs * part_ld; part2 += (int64_t)b * pb.slabs * part_ld; slots += (int64_t)b * 16;
        if (pb.rho_arr) rho = (T)pb.rho_arr[b];
    }
    using V = typename PS::V;
    constexpr int VN = PS::VN, CHUNK = PS::CHUNK, WAVES = PS::WAVES;
    const int tid = threadIdx.x, g = blockIdx.x;
    const int row0 = g * rows_per_wg;
    const int row1 = min(MP, row0 + rows_per_wg);
    const T alpha1 = T(1) - alpha, rho1 = T(1) / rho;

    __shared__ T red[WAVES][R];
    __shared__ T red2[WAVES][R];
    __shared__ T wsh[R];
    __shared__ T ysh[R];

    // this thread's slice of x~ (and of the relaxed x, SolveQuadraticProgram.jl:57)
    V xv[KC];
    V xn[CHECK ? KC : 1];
    unsigned long long mdx = 0ull;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = tid * VN + k * CHUNK;
        T* xp_ = reinterpret_cast<T*>(&xv[k]);
        if (c < NP) {
            xv[k] = *reinterpret_cast<const V*>(xx + c);
            // (plain variant: workgroup 0 writes the relaxed x at the END of the kernel -- here its four load / wait / store round trips made it
            // start streaming A ~3 us after the other 255 workgroups, and the launch lasts as long as its slowest workgroup)
            if (CHECK) {
                const V xo = *reinterpret_cast<const V*>(x_old + c);
                const T* xop = reinterpret_cast<const T*>(&xo);
                V xnv;
                T* xnp = reinterpret_cast<T*>(&xnv);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    xnp[e] = alpha * xp_[e] + alpha1 * xop[e];                       // :57
                    if (CHECK) mdx = umax(mdx, absbits((double)(xnp[e] - xop[e])));  // :105 norm(vX - vXP, Inf)
                }
                if (g == 0) *reinterpret_cast<V*>(x_new + c) = xnv;
                if (CHECK) xn[k] = xnv;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VN; ++e) xp_[e] = T(0);
            if (CHECK) { T* q_ = reinterpret_cast<T*>(&xn[k]);
#pragma unroll
                for (int e = 0; e < VN; ++e) q_[e] = T(0); }
        }
    }
    if (CHECK && g == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mdx = umax(mdx, (unsigned long long)__shfl_xor(mdx, o, 64));
        if ((tid & 63) == 0 && mdx) atomicMax(&slots[7], mdx);
    }

    T acc[KC][VN];
    T acc2[CHECK ? KC : 1][VN];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int e = 0; e < VN; ++e) { acc[k][e] = T(0); if (CHECK) acc2[k][e] = T(0); }

    unsigned long long m_res = 0ull, m_ax = 0ull, m_z = 0ull, m_dz = 0ull;   // owned by threads tid < R

    V bufA[R][KC], bufB[R][KC];
    // per-row scalars of the tile, prefetched by the R owner threads
    T zA = T(0), yA = T(0), lA = T(0), uA = T(0), zB = T(0), yB = T(0), lB = T(0), uB = T(0);

    auto prefetch_rows = [&](int row, T& zo, T& yo, T& lo, T& hi) {
        if (tid < R) { zo = z[row + tid]; yo = y[row + tid]; lo = l[row + tid]; hi = u[row + tid]; }
    };
    auto process = [&](V (&a)[R][KC], int row, T zo, T yo, T lo, T hi) {
        T d[R], d2[CHECK ? R : 1];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            T s = T(0), s2 = T(0);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const T* ap = reinterpret_cast<const T*>(&a[i][k]);
                const T* xp_ = reinterpret_cast<const T*>(&xv[k]);
                const T* xq_ = reinterpret_cast<const T*>(&xn[CHECK ? k : 0]);
#pragma unroll
                for (int e = 0; e < VN; ++e) { s += ap[e] * xp_[e]; if (CHECK) s2 += ap[e] * xq_[e]; }
            }
            d[i] = wave_sum_all(s);                       // DPP + readlane, no LDS-pipe instruction (wave_reduce.h)
            if (CHECK) d2[i] = wave_sum_all(s2);
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) { red[tid >> 6][i] = d[i]; if (CHECK) red2[tid >> 6][i] = d2[i]; }
        }
        __syncthreads();
        if (tid < R) {
            T zt = T(0), ax = T(0);
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { zt += red[w][tid]; if (CHECK) ax += red2[w][tid]; }   // fixed order
            T zn, yn;
            if (MODE == 0) {
                const T t = alpha * zt + alpha1 * zo + rho1 * yo;                  // :60
                zn = t > hi ? hi : (t < lo ? lo : t);                               //     clamp(., vL, vU)
                yn = yo + rho * (alpha * zt + alpha1 * zo - zn);                    // :61
                wsh[tid] = rho * zn - yn;                                           // LinearSystemSolvers.jl:134
            } else if (MODE == 2) {
                // masked KKT product of the polishing step (SolveQuadraticProgram.m:304-305): l-array = 0/1 mask of the active
                // rows, y-array = multiplier block of the input (read only), alpha = delta; out = mask (A v_x) - delta mask v_lambda
                zn = lo * zt - alpha * (lo * yo);
                yn = yo;
                wsh[tid] = lo * yo;
            } else {
                // ProxQP.jl rows over G = [A; C]: z-array = slack s, y-array = dual (y | z), l-array = bound (b | d), zt = (G x)_row
                const T gb = lo, vr = zt;
                if (row + tid < pb.pq_me) {                                         // equality row: y += rho (A x - b)   ProxQP.jl:238-239
                    zn = T(0);
                    yn = yo - rho * gb; yn += rho * vr;
                    wsh[tid] = rho * gb - yn;                                       // :212  rho b - y
                } else {                                                            // inequality row                     ProxQP.jl:230-232, 246-248
                    T sv = gb - rho1 * yo; sv += -vr; zn = sv > T(0) ? sv : T(0);
                    T zz_ = yo + rho * (zn - gb); zz_ += rho * vr; yn = zz_ > T(0) ? zz_ : T(0);
                    wsh[tid] = rho * (gb - zn) - yn;                                // :215  rho (d - s) - z
                }
            }
            z[row + tid] = zn;
            if (MODE != 2) y[row + tid] = yn;
            if (CHECK) {
                ysh[tid] = yn;
                m_res = umax(m_res, absbits((double)(ax - zn)));                    // :85
                m_ax = umax(m_ax, absbits((double)ax));                             // :88
                m_z = umax(m_z, absbits((double)zn));
                m_dz = umax(m_dz, absbits((double)(zn - zo)));                      // :105
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const T w = wsh[i];
            const T yw = CHECK ? ysh[i] : T(0);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const T* ap = reinterpret_cast<const T*>(&a[i][k]);
#pragma unroll
                for (int e = 0; e < VN; ++e) { acc[k][e] += ap[e] * w; if (CHECK) acc2[k][e] += ap[e] * yw; }
            }
        }
    };

    if (row0 < row1) {
        PS::load_tile(bufA, A, ld, row0, NP, tid);
        prefetch_rows(row0, zA, yA, lA, uA);
    }
    for (int row = row0; row < row1; row += 2 * R) {
        const bool haveB = row + R < row1;
        if (haveB) { PS::load_tile(bufB, A, ld, row + R, NP, tid); prefetch_rows(row + R, zB, yB, lB, uB); }
        process(bufA, row, zA, yA, lA, uA);
        if (haveB) {
            if (row + 2 * R < row1) { PS::load_tile(bufA, A, ld, row + 2 * R, NP, tid); prefetch_rows(row + 2 * R, zA, yA, lA, uA); }
            process(bufB, row + R, zB, yB, lB, uB);
        }
    }

    if (!CHECK && MODE != 2 && g == 0) {
        // x_new = alpha x~ + (1 - alpha) x_old (SolveQuadraticProgram.jl:57) by workgroup 0, all loads in flight together
        V xo[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) xo[k] = *reinterpret_cast<const V*>(x_old + min(tid * VN + k * CHUNK, NP - VN));
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = tid * VN + k * CHUNK;
            if (c < NP) {
                const T* xp_ = reinterpret_cast<const T*>(&xv[k]);
                const T* xop = reinterpret_cast<const T*>(&xo[k]);
                V xnv; T* xnp = reinterpret_cast<T*>(&xnv);
#pragma unroll
                for (int e = 0; e < VN; ++e) xnp[e] = alpha * xp_[e] + alpha1 * xop[e];
                *reinterpret_cast<V*>(x_new + c) = xnv;
            }
        }
    }

    // slab of column sums (zeros for an idle workgroup, so colsum may add every slab unconditionally)
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = tid * VN + k * CHUNK;
        if (c < NP) {
            V o; T* op = reinterpret_cast<T*>(&o);
#pragma unroll
            for (int e = 0; e < VN; ++e) op[e] = acc[k][e];
            // write-through slab stores: the slabs are read by the NEXT kernel, nothing in this one; leaving them dirty in L2 costs
            // a flush at the kernel boundary (measured on C2: 72.3 -> 71.6 us per iteration)
#if QPS_NT_SLABS
            { typedef T NVS __attribute__((ext_vector_type(VN))); NVS ov;
#pragma unroll
              for (int e = 0; e < VN; ++e) ov[e] = op[e];
              __builtin_nontemporal_store(ov, reinterpret_cast<NVS*>(part + (int64_t)g * part_ld + c)); }
#else
            *reinterpret_cast<V*>(part + (int64_t)g * part_ld + c) = o;
#endif
            if (CHECK) {
#pragma unroll
                for (int e = 0; e < VN; ++e) op[e] = acc2[k][e];
                *reinterpret_cast<V*>(part2 + (int64_t)g * part_ld + c) = o;
            }
        }
    }
    if (CHECK && tid < R) {
        if (m_res) atomicMax(&slots[0], m_res);
        if (m_ax) atomicMax(&slots[2], m_ax);
        if (m_z) atomicMax(&slots[3], m_z);
        if (m_dz) atomicMax(&slots[8], m_dz);
    }
}

template <typename T, int THREADS, int KC, int R, int RC, int MODE>
void launch_pass(hipStream_t st, bool check, int G, const T* A, int64_t ld, int NP, int MP, int rows_per_wg, const T* xx,
                 const T* x_old, T* x_new, T* z, T* y, const T* l, const T* u, T alpha, T rho, T* part, T* part2,
                 int64_t part_ld, unsigned long long* slots, PassBatch pb) {
    if constexpr (MODE == 0) {
        if (check) {
            if (g_launch_timing.start) {
                const LaunchTiming lt = g_launch_timing;
                g_launch_timing = LaunchTiming();
                hipExtLaunchKernelGGL((k_apass<T, THREADS, KC, RC, true, 0>), dim3(G, pb.count), dim3(THREADS), 0, st, lt.start, lt.stop, 0, A, ld, NP, MP, rows_per_wg, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
            } else {
                hipLaunchKernelGGL((k_apass<T, THREADS, KC, RC, true, 0>), dim3(G, pb.count), dim3(THREADS), 0, st, A, ld, NP, MP, rows_per_wg, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
            }
            return;
        }
    }
    if (g_launch_timing.start) {   // profiled launch: timestamps of the dispatch itself
        const LaunchTiming lt = g_launch_timing;
        g_launch_timing = LaunchTiming();
        hipExtLaunchKernelGGL((k_apass<T, THREADS, KC, R, false, MODE>), dim3(G, pb.count), dim3(THREADS), 0, st, lt.start, lt.stop, 0, A, ld, NP, MP, rows_per_wg, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
        return;
    }
    hipLaunchKernelGGL((k_apass<T, THREADS, KC, R, false, MODE>), dim3(G, pb.count), dim3(THREADS), 0, st, A, ld, NP, MP, rows_per_wg, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
}

}  // namespace

}  // namespace qps
