// k_trsv.hip -- the triangular solve for the case where one inverted diagonal block covers the whole factor (nb >= n):
// both sweeps of x~ = L'^{-1} (L^{-1} t) in one pass over W = inv(L).  (A persistent, software-pipelined kernel per
// *separate* sweep was measured and dropped: 17-20 us per sweep against 18.8 us for the plain row-dot launch.)
#include <algorithm>
#include <cstdlib>

#include "qps_kernels.h"
#include "wave_reduce.h"
#include <hip/hip_ext.h>

namespace qps {

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Both sweeps in ONE pass over the lower triangle (nb >= n, S lower = W = inv(L)):
//     y_r = sum_{c<=r} W[r][c] t[c]            (forward sweep, row dot)
//     x~_c = sum_{r>=c} W[r][c] y_r            (backward sweep = column accumulation of the SAME entries)
// A tile of RB rows sits in registers; after its row dots are reduced (one barrier, red[] double-buffered, every thread
// sums the per-wave partials itself so no second barrier / broadcast is needed) the tile is reused for the column
// accumulation.  The triangle is read once (n(n+1)/2 elements instead of n(n+1)); each workgroup leaves a slab of
// column sums that colsum() adds in fixed order.  Same structure as the fused A-pass (k_pass.hip).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int THREADS, int KC, int RB, bool DB = true>   // DB: double-buffered tiles (the wide variants KC > 8 keep one tile: registers)
__global__ __launch_bounds__(THREADS) void k_sweep_fused(const T* __restrict__ S, int64_t ld, int NP, const T* __restrict__ v,
                                                         T* __restrict__ part, int64_t part_ld, BatchStride bs) {
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N, CHUNK = THREADS * VN, WAVES = THREADS / 64;
    if (bs.active && !bs.active[blockIdx.y]) return;
    S += (int64_t)blockIdx.y * bs.mat; v += (int64_t)blockIdx.y * bs.vin; part += (int64_t)blockIdx.y * bs.vout;
    const int tid = threadIdx.x, G = gridDim.x, g = blockIdx.x;
    const int ntiles = NP / RB;
    __shared__ T red[2][WAVES][RB];

    V xv[KC];
    T acc[KC][VN];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = tid * VN + k * CHUNK;
        if (c < NP) xv[k] = *reinterpret_cast<const V*>(v + c);
        else { T* p = reinterpret_cast<T*>(&xv[k]);
#pragma unroll
            for (int e = 0; e < VN; ++e) p[e] = T(0); }
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[k][e] = T(0);
    }
    auto tile_row = [&](int it) { return (ntiles - 1 - (g + it * G)) * RB; };   // long rows first, cyclic over workgroups
    auto load = [&](V (&a)[RB][KC], int row) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                // UNCONDITIONAL loads (threads right of the diagonal re-read the diagonal's own vector -- a line the row needs anyway; process() zeroes
                // what lies above the diagonal): with a predicated load here the compiler could not count the loads in flight and waited for ALL of them
                // (vmcnt(0)) before every tile -- the tile it had just requested included, so the double buffer never overlapped anything
                const int c = tid * VN + k * CHUNK;
                const int cc = min(min(c, (row + i) & ~(VN - 1)), NP - VN);
                a[i][k] = *reinterpret_cast<const V*>(S + (int64_t)(row + i) * ld + cc);
            }
    };
    // A wave whose first column lies right of the diagonal holds only zeros of this row: it skips the dot, the reduction and the
    // accumulation (wave-uniform test) and only meets the barrier -- on the narrow batched problems half the waves of a row.
    constexpr bool SKIP = KC <= 2;          // wide problems (KC = 4: C2) lose more to the extra branches than the few idle waves cost
    const int wave_col0 = (tid & ~63) * VN;
    auto process = [&](V (&a)[RB][KC], int row, int par) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            T s = T(0);
            if (SKIP && wave_col0 > row + i) { if ((tid & 63) == 0) red[par][tid >> 6][i] = T(0); continue; }
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = tid * VN + k * CHUNK;
                T* ap = reinterpret_cast<T*>(&a[i][k]);
                const T* xp = reinterpret_cast<const T*>(&xv[k]);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    if (c + e > row + i) ap[e] = T(0);            // entries above the diagonal belong to the mirrored half
                    s += ap[e] * xp[e];
                }
            }
            s = wave_sum_all(s);                                   // DPP + readlane: no LDS-pipe instruction
            if ((tid & 63) == 0) red[par][tid >> 6][i] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            // lanes 0..WAVES-1 fetch one per-wave partial each (one LDS read per row and wave instead of WAVES broadcast reads per
            // thread); fixed order, identical in every wave
            static_assert(WAVES <= 8, "one partial per lane of the first eight");
            if (SKIP && wave_col0 > row + i) continue;
            const T y = lanes8_sum_all(((tid & 63) < WAVES) ? red[par][tid & 63][i] : T(0));
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const T* ap = reinterpret_cast<const T*>(&a[i][k]);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[k][e] += ap[e] * y;
            }
        }
    };
    const int my_tiles = (ntiles - g + G - 1) / G;
    if constexpr (DB) {
        if (my_tiles > 0) {
            V bufA[RB][KC], bufB[RB][KC];
            load(bufA, tile_row(0));
            for (int it = 0; it < my_tiles; it += 2) {
                const bool haveB = it + 1 < my_tiles;
                load(bufB, tile_row(min(it + 1, my_tiles - 1)));            // (past the end: the last tile again, never used -- no branch around a load)
                process(bufA, tile_row(it), 0);
                load(bufA, tile_row(min(it + 2, my_tiles - 1)));
                if (haveB) process(bufB, tile_row(it + 1), 1);
            }
        }
    } else {
        V buf[RB][KC];
        for (int it = 0; it < my_tiles; ++it) { load(buf, tile_row(it)); process(buf, tile_row(it), it & 1); }
    }
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
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Narrow factors (a row fits ONE wave: NP <= 64 * VN * KCW = 1024 fp64 / 2048 fp32 -- BASELINE config 4's n = 1024): a WAVE owns a row.
// In the kernel above 512 threads share a row of at most 8 KB: one 16-byte load per thread, then a wave reduction, an LDS exchange and a
// barrier per row pair -- on the batched n = 1024 problems that bookkeeping, not the bytes, set the pace (38 us for 134 MB).  Here every wave
// streams whole rows (KCW 16-byte loads per lane and row, the next row in flight), the row dot ends in one DPP wave sum with no LDS traffic
// and no barrier, the column accumulation stays in the wave's registers, and the eight waves of a workgroup add their column sums through LDS
// once, at the end.  Rows are dealt cyclically over all waves of a QP, longest first.  Same slab interface as k_sweep_fused.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int KCW, bool DB = true>
__global__ __launch_bounds__(512) void k_sweep_fused_wave(const T* __restrict__ S, int64_t ld, int NP, const T* __restrict__ v,
                                                          T* __restrict__ part, int64_t part_ld, BatchStride bs) {
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N, CH = 64 * VN, WAVES = 8, COLS = KCW * CH;
    if (bs.active && !bs.active[blockIdx.y]) return;
    S += (int64_t)blockIdx.y * bs.mat; v += (int64_t)blockIdx.y * bs.vin; part += (int64_t)blockIdx.y * bs.vout;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    const int W = gridDim.x * WAVES, w = wave * gridDim.x + g;        // global wave index: neighbouring rows go to different workgroups
    __shared__ T sh[WAVES][COLS];

    V xv[KCW];
    T acc[KCW][VN];
#pragma unroll
    for (int k = 0; k < KCW; ++k) {
        const int c = k * CH + lane * VN;
        xv[k] = *reinterpret_cast<const V*>(v + min(c, NP - VN));
        if (c >= NP) { T* p = reinterpret_cast<T*>(&xv[k]);
#pragma unroll
            for (int e = 0; e < VN; ++e) p[e] = T(0); }
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[k][e] = T(0);
    }
    auto row_of = [&](int it) { return NP - 1 - (w + it * W); };
    auto load = [&](V (&a)[KCW], int r) {
#pragma unroll
        for (int k = 0; k < KCW; ++k) {
            // unconditional loads (exact load counting): lanes right of the diagonal re-read the diagonal's own vector -- a line the row needs anyway
            const int c = k * CH + lane * VN;
            const int cc = min(min(c, r & ~(VN - 1)), NP - VN);
            a[k] = *reinterpret_cast<const V*>(S + (int64_t)r * ld + cc);
        }
    };
    auto process = [&](V (&a)[KCW], int r) {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < KCW; ++k) {
            const int c = k * CH + lane * VN;
            T* ap = reinterpret_cast<T*>(&a[k]);
            const T* xp = reinterpret_cast<const T*>(&xv[k]);
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                if (c + e > r) ap[e] = T(0);                       // entries above the diagonal belong to the mirrored half
                s += ap[e] * xp[e];
            }
        }
        const T y = wave_sum_all(s);                               // y_r = W_r . t
#pragma unroll
        for (int k = 0; k < KCW; ++k) {
            const T* ap = reinterpret_cast<const T*>(&a[k]);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[k][e] += ap[e] * y;   // x~ += W_r' y_r
        }
    };
    const int mine = w < NP ? (NP - w + W - 1) / W : 0;
    if constexpr (!DB) {
        V buf[KCW];
        for (int it = 0; it < mine; ++it) { load(buf, row_of(it)); process(buf, row_of(it)); }
    } else if (mine > 0) {
        V bufA[KCW], bufB[KCW];
        load(bufA, row_of(0));
        for (int it = 0; it < mine; it += 2) {
            const int rb = row_of(min(it + 1, mine - 1));          // past the end: the last row again, never used
            load(bufB, rb);
            process(bufA, row_of(it));
            load(bufA, row_of(min(it + 2, mine - 1)));
            if (it + 1 < mine) process(bufB, rb);
        }
    }
#pragma unroll
    for (int k = 0; k < KCW; ++k)
#pragma unroll
        for (int e = 0; e < VN; ++e) sh[wave][k * CH + lane * VN + e] = acc[k][e];
    __syncthreads();
    for (int c = tid; c < NP; c += 512) {
        T t = T(0);
#pragma unroll
        for (int q = 0; q < WAVES; ++q) t += sh[q][c];                 // fixed order
        part[(int64_t)g * part_ld + c] = t;
    }
}

static bool sweep_wave() { static const bool on = [] { const char* e = getenv("QPS_SWEEP_WAVE"); return !(e && atoi(e) == 0); }(); return on; }
template <typename T> static bool sweep_wave_covers(int NP) { return sweep_wave() && NP <= 64 * VecOf<T>::N * 8; }

}  // namespace

template <typename T> bool sweep_fused_supported(int NP) { return NP <= 16 * 512 * VecOf<T>::N && NP >= 1024; }   // 16384 fp64 / 32768 fp32

// Rows per tile of the double-buffered variants: about 64 B per thread and tile (two tiles = 64 KB in flight per CU).  Measured with the exact-wait
// loads (one MI355X, dispatch events): fp64 n = 4096 (KC 4) 18.0 / 16.0 / 15.6 us at 4 / 2 / 1 rows, fp32 n = 4096 (KC 2) 12.0 / 11.0 / 11.1 us --
// more bytes in flight per CU make the sweep slower, not faster.  QPS_SWEEP_RB forces a value (1, 2, 4).
static int sweep_rb_env() { static int rb = [] { const char* e = getenv("QPS_SWEEP_RB"); return e ? atoi(e) : 0; }(); return rb; }
static int sweep_rb_for(int kc) {
    int rb = sweep_rb_env() > 0 ? sweep_rb_env() : (kc >= 4 ? 1 : (kc == 2 ? 2 : 4));
    if (rb != 1 && rb != 2 && rb != 4) rb = 2;
    if (kc > 4 && rb == 4) rb = 2;        // (8 chunks x 4 rows x 2 tiles does not fit the register file)
    return rb;
}

template <typename T> int sweep_fused_slabs(int NP, int count) {
    if (sweep_wave_covers<T>(NP)) {      // wave-per-row kernel: about one workgroup per CU over the whole launch, at least two rows per wave
        const int per = count >= 256 ? 1 : 256 / (count < 1 ? 1 : count);
        return std::max(1, std::min(per, NP / 16));
    }
    const int kc0 = (NP + 512 * VecOf<T>::N - 1) / (512 * VecOf<T>::N);
    int RB = sweep_rb_for(kc0);
    if (NP > 8 * 512 * VecOf<T>::N) RB = 1;   // wide single-buffered variants
    static const int total_env = [] { const char* e = getenv("QPS_SWEEP_WGS"); return e ? atoi(e) : 0; }();
    const int kc = (NP + 512 * VecOf<T>::N - 1) / (512 * VecOf<T>::N);
    const int total = total_env > 0 ? total_env : (count <= 1 ? 256 : (kc <= 1 ? 1024 : (kc == 2 ? 512 : 256)));   // batches: by register footprint
    const int per = count >= total ? 1 : total / count;
    const int ntiles = NP / RB;
    return ntiles < per ? ntiles : per;
}

// slab set written at part + qp * bs.vout (bs.vout = slabs * part_ld for a batch); returns the number of slabs per QP
template <typename T>
int sweep_fused(hipStream_t st, const T* S, int64_t ld, int NP, const T* v, T* part, int64_t part_ld, BatchStride bs) {
    constexpr int TH = 512;
    const int chunk = TH * VecOf<T>::N;
    const int kc = (NP + chunk - 1) / chunk;
    const int G = sweep_fused_slabs<T>(NP, bs.count);
    dim3 grid(G, bs.count);
    const int rb = sweep_rb_for(kc);
    const LaunchTiming lt = g_launch_timing;   // profiled launch: the dispatch's own timestamps (qps_kernels.h)
    g_launch_timing = LaunchTiming();
    if (sweep_wave_covers<T>(NP)) {
#define QPS_WV(KCW)                                                                                                                \
    do {                                                                                                                           \
        if (lt.start) hipExtLaunchKernelGGL((k_sweep_fused_wave<T, KCW>), grid, dim3(TH), 0, st, lt.start, lt.stop, 0, S, ld, NP, v, part, part_ld, bs); \
        else hipLaunchKernelGGL((k_sweep_fused_wave<T, KCW>), grid, dim3(TH), 0, st, S, ld, NP, v, part, part_ld, bs);              \
    } while (0)
        // one row per wave in flight (no second row prefetched): 250-252 k -> 254.3 k QP-it/s on the 32-QP slab of C4 -- the eight independent waves of a
        // workgroup already overlap each other, a second row per wave only adds requests in flight (QPS_SWEEP_WAVE_DB=1 prefetches it)
        static const int wave_db = [] { const char* e = getenv("QPS_SWEEP_WAVE_DB"); return e ? atoi(e) : 0; }();
        if (NP <= 64 * VecOf<T>::N * 4) QPS_WV(4);
        else if (wave_db == 0) { if (lt.start) hipExtLaunchKernelGGL((k_sweep_fused_wave<T, 8, false>), grid, dim3(TH), 0, st, lt.start, lt.stop, 0, S, ld, NP, v, part, part_ld, bs); else hipLaunchKernelGGL((k_sweep_fused_wave<T, 8, false>), grid, dim3(TH), 0, st, S, ld, NP, v, part, part_ld, bs); }
        else QPS_WV(8);
#undef QPS_WV
        return G;
    }
#define QPS_S(KC, RB)                                                                                                              \
    do {                                                                                                                           \
        if (lt.start) hipExtLaunchKernelGGL((k_sweep_fused<T, TH, KC, RB>), grid, dim3(TH), 0, st, lt.start, lt.stop, 0, S, ld, NP, v, part, part_ld, bs); \
        else hipLaunchKernelGGL((k_sweep_fused<T, TH, KC, RB>), grid, dim3(TH), 0, st, S, ld, NP, v, part, part_ld, bs);            \
    } while (0)
#define QPS_SW(KC)                                                                                                                 \
    do {                                                                                                                           \
        if (lt.start) hipExtLaunchKernelGGL((k_sweep_fused<T, TH, KC, 1, false>), grid, dim3(TH), 0, st, lt.start, lt.stop, 0, S, ld, NP, v, part, part_ld, bs); \
        else hipLaunchKernelGGL((k_sweep_fused<T, TH, KC, 1, false>), grid, dim3(TH), 0, st, S, ld, NP, v, part, part_ld, bs);      \
    } while (0)
#define QPS_SR(KC) do { if (rb == 1) QPS_S(KC, 1); else if (rb == 2) QPS_S(KC, 2); else QPS_S(KC, 4); } while (0)
    if (kc > 8) { if (kc <= 12) QPS_SW(12); else QPS_SW(16); }
    else if (kc <= 1) QPS_SR(1);
    else if (kc <= 2) QPS_SR(2);
    else if (kc <= 4) QPS_SR(4);
    else { if (rb == 1) QPS_S(8, 1); else QPS_S(8, 2); }
#undef QPS_SR
#undef QPS_SW
#undef QPS_S
    return G;
}

#define INST(T)                                                     \
    template int sweep_fused_slabs<T>(int, int);                    \
    template int sweep_fused<T>(hipStream_t, const T*, int64_t, int, const T*, T*, int64_t, BatchStride); \
    template bool sweep_fused_supported<T>(int);
INST(double)
INST(float)
#undef INST

}  // namespace qps
