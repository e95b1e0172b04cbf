// k_trsv_blocked.hip -- blocked triangular substitution (trsvBlock < n) as ONE launch per sweep.
//
// This is the kernel BASELINE's north_star names ("repeated triangular back-substitution ... LDS-staged dense tiles"): the factor is cut
// into nb x nb blocks (nb = 1024 fp64 / 2048 fp32 columns = one 16-byte load per thread of a 512-thread row), the diagonal blocks are
// inverted at factorisation time and every block row is pre-multiplied by its inverted diagonal block (build_sweep_matrix, premul):
//
//     forward   y_J = W_JJ t_J - sum_{K<J} (W_JJ L_JK) y_K           W_JJ = inv(L_JJ)
//     backward  x_J = W_JJ' y_J - sum_{K>J} (L_KJ W_JJ)' x_K
//
// so that one block row is ONE dependent phase (n / nb phases per sweep instead of 2 n / nb - 1) and every row of the sweep matrix is a
// plain row dot whose operand is the right-hand side inside the diagonal block and the already finished part of the solution left
// (forward) / right (backward) of it.  Same bytes as any triangular sweep: n (n + 1) / 2 elements.
//
// One launch = 256 persistent workgroups (one per CU), each 8 streaming waves + 1 courier wave:
//   * every workgroup owns nb / 256 rows of EVERY block row (dealt from alternating ends, so that the triangular diagonal blocks balance);
//   * the streaming waves only ever issue loads of the sweep matrix: a window of W row tiles (nb/256 rows x 16 B per thread each) is in
//     flight at all times, ordered block row by block row with the tile that depends on the previous phase last.  They never wait on a
//     vector-memory counter for anything but their own tiles: operands come from LDS (`vsh`), hand-offs from LDS flags;
//   * the courier wave does all the talking between workgroups.  It adds the streaming waves' partial sums, stores the rows of y and
//     publishes them as 8-byte {epoch, 32 bits of the value} granules (write-through, data = flag: guide, Guideline 16 R2), and it
//     polls the granules of the previous block (every workgroup needs all nb values, produced four per workgroup), unpacks them into
//     LDS and raises the LDS flag the streaming waves wait on before their last tile of the block row.
// The epoch is a per-launch argument that only ever grows, so no word has to be cleared between launches and a launch that gave up
// leaves nothing behind.
//
// Co-residency.  The hand-offs need all 256 workgroups of a launch on the chip at once.  That is GUARANTEED, not hoped for:
//   * trsv_blocked_supported() admits the kernel on a device only when that device has >= 256 CUs and
//     hipOccupancyMaxActiveBlocksPerMultiprocessor says a workgroup with its 144 KiB of LDS fits a CU (facts cached per device ordinal);
//   * every other kernel of the library terminates without waiting for another launch, so it can only delay these workgroups;
//   * the one thing that could starve a launch for good is a SECOND launch of this kernel holding some of the CUs while waiting for
//     its own missing workgroups.  SweepGate rules that out inside a process: all blocked-sweep launches of a device are chained
//     across streams (the pair forward + backward is enqueued under a per-device mutex; a pair that follows a pair of ANOTHER stream
//     first waits -- hipStreamWaitEvent -- for an event recorded behind that one), so at most one of them is ever on the chip.  A handle
//     alone on its device pays one uncontended mutex per pair and no event.
// The bounded spin stays as a backstop for what the process cannot see (another PROCESS running the same kernel on the card): a courier
// that waited `spin_limit` polls sets `abort_word`, every courier checks that word while it spins, and the launch drains with garbage in
// `out`; the host sees the word at its next read-back, repeats that solve on the one-launch-per-phase substitution (qps_info.sweepGaveUp
// counts it) and tries the blocked sweeps again at its next solve (qps_capi.hip).
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "qps_kernels.h"
#include "wave_reduce.h"
#include <hip/hip_ext.h>

namespace qps {

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
constexpr int TBG = 256;   // workgroups of a launch = rows-per-block divisor (one workgroup per CU)

__device__ __forceinline__ unsigned lds_get(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_set(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T> __device__ __forceinline__ T elem(const typename VecOf<T>::type& v, int e);
template <> __device__ __forceinline__ double elem<double>(const double2& v, int e) { return e == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float elem<float>(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// phase p works on block row blk(p); its tiles in consumption order: the blocks of phases 0 .. p-2 (finished long ago), the diagonal
// block (operand = right-hand side), and last the block of phase p-1 (the dependency).  `idx` in [0, p] (p = 0: the diagonal tile only).
__device__ __forceinline__ int tile_phase(int p, int idx) { return p == 0 ? 0 : (idx == p ? p - 1 : (idx == p - 1 ? p : idx)); }
__device__ __forceinline__ void tile_next(int& p, int& idx) { if (idx >= p) { ++p; idx = 0; } else ++idx; }

template <typename T, int STREAM, bool BWD, int W>
__global__ __launch_bounds__(STREAM + 64) void k_trsv_blocked(const T* __restrict__ S, int64_t ld, int NP, const T* __restrict__ v,
                                                              T* __restrict__ out, unsigned long long* pub, int64_t plane, unsigned epoch,
                                                              unsigned* abort_word, unsigned spin_limit, int mode) {
    // mode (measurements only, tests/tools/micro/trsv_blocked_bench.hip): bit 0 = the courier does not wait for the other workgroups'
    // granules (one poll pass), bit 1 = every matrix load hits one cached line (the chain without the stream); results are garbage then
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N, NB = STREAM * VN, RB = NB / TBG, SW = STREAM / 64;
    static_assert(RB >= 1 && RB * TBG == NB, "block rows are dealt evenly over the workgroups");
    extern __shared__ __align__(16) unsigned char trsv_smem[];
    T* vsh = reinterpret_cast<T*>(trsv_smem);          // nblk * NB operands: chunk K holds t_K until y_K (x_K) has been gathered
    __shared__ T red[2][SW][RB];
    __shared__ unsigned flags[2];                       // [0] gathered phases + 1, [1] partial-sum arrivals of the streaming waves
    const int tid = threadIdx.x, g = blockIdx.x;
    const int nblk = (NP + NB - 1) / NB;
    if (tid == 0) { flags[0] = 1u; flags[1] = 0u; }
    __syncthreads();
    auto blk = [&](int q) { return BWD ? nblk - 1 - q : q; };
    auto row0 = [&](int p) { return blk(p) * NB + ((p & 1) ? (TBG - 1 - g) : g) * RB; };

    if (tid >= STREAM) {
        // ------------------------------------------------------------------------------------------------ courier wave
        const int lane = tid - STREAM;
        constexpr int NG = NB / 64;                    // operands per lane of one gathered block
        bool dead = false;
        for (int p = 0; p < nblk; ++p) {
            if (p >= 1) {
                const int K = blk(p - 1);
                unsigned long long glo[NG], ghi[sizeof(T) == 8 ? NG : 1];
                for (unsigned spins = 0;; ++spins) {
                    // every load of the pass first (clamped addresses, no branch in between: they overlap), then the tags
#pragma unroll
                    for (int k = 0; k < NG; ++k) {
                        const int c = min(K * NB + k * 64 + lane, NP - 1);
                        glo[k] = __hip_atomic_load((gu64*)(pub + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (sizeof(T) == 8) ghi[sizeof(T) == 8 ? k : 0] = __hip_atomic_load((gu64*)(pub + plane + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NG; ++k) {
                        const bool beyond = K * NB + k * 64 + lane >= NP;
                        bool okk = (unsigned)(glo[k] >> 32) == epoch;
                        if (sizeof(T) == 8) okk = okk & ((unsigned)(ghi[sizeof(T) == 8 ? k : 0] >> 32) == epoch);
                        ok = ok & (okk | beyond);
                    }
                    if (__all(ok) || dead || (mode & 1)) break;
                    if (spins >= spin_limit) {                                  // give up: every courier of the launch follows (bounded)
                        dead = true;
                        if (lane == 0) __hip_atomic_store((gu32*)(abort_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if ((spins & 63u) == 63u) {
                        if (__hip_atomic_load((gu32*)(abort_word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) dead = true;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int c = K * NB + k * 64 + lane;
                    T val;
                    if (sizeof(T) == 8) {
                        const unsigned long long bits = (ghi[sizeof(T) == 8 ? k : 0] << 32) | (glo[k] & 0xffffffffull);
                        val = (T)__longlong_as_double((long long)bits);
                    } else {
                        val = (T)__uint_as_float((unsigned)glo[k]);
                    }
                    vsh[c] = c < NP ? val : T(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the operands are in LDS before the flag is
                if (lane == 0) lds_set(&flags[0], (unsigned)(p + 1));
            }
            // partial sums of this phase from the streaming waves
            while (lds_get(&flags[1]) < (unsigned)(SW * (p + 1))) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            if (lane < RB) {
                T s = T(0);
#pragma unroll
                for (int w = 0; w < SW; ++w) s += red[p & 1][w][lane];           // fixed order
                const int r = row0(p) + lane;
                if (r < NP) {
                    if (p + 1 < nblk) {                                          // the last block row has no consumer in this launch
                        if (sizeof(T) == 8) {
                            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)s);
                            __hip_atomic_store((gu64*)(pub + r), ((unsigned long long)epoch << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((gu64*)(pub + plane + r), ((unsigned long long)epoch << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            __hip_atomic_store((gu64*)(pub + r), ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint((float)s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    out[r] = s;
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- streaming waves
    const int lane = tid & 63, wave = tid >> 6;
    const int ttot = nblk * (nblk + 1) / 2;
    auto load = [&](V (&b)[RB], int p, int idx) {
        const int q = tile_phase(p, idx);
        const int c = blk(q) * NB + tid * VN, r0 = row0(p);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            int r = min(r0 + i, NP - 1);
            int cc = c;
            // diagonal tile: lanes beyond the diagonal re-read the diagonal's own vector (same line: no extra traffic, no branch)
            if (q == p) cc = BWD ? max(c, r & ~(VN - 1)) : min(c, r & ~(VN - 1));
            cc = min(cc, NP - VN);
            if (mode & 2) { r = 0; cc = tid * VN; }
            b[i] = *reinterpret_cast<const V*>(S + (int64_t)r * ld + cc);
        }
    };
    T acc[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i] = T(0);
    auto consume = [&](const V (&b)[RB], int p, int idx) {
        const int q = tile_phase(p, idx);
        if (p > 0 && idx == p) {                                                 // the dependent tile: operand gathered by the courier
            while (lds_get(&flags[0]) < (unsigned)(p + 1)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        const int c = blk(q) * NB + tid * VN, r0 = row0(p);
        const V xv = *reinterpret_cast<const V*>(vsh + c);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                T a = elem<T>(b[i], e);
                if (q == p) { const bool keep = BWD ? (c + e >= r0 + i) : (c + e <= r0 + i); a = keep ? a : T(0); }
                acc[i] += a * elem<T>(xv, e);
            }
        if (idx >= p) {                                                          // last tile of the block row
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const T s = wave_sum_all(acc[i]);
                if (lane == 0) red[p & 1][wave][i] = s;
                acc[i] = T(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&flags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    V buf[W][RB];
    int lp = 0, lidx = 0, issued = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) { load(buf[w], lp, lidx); if (++issued < ttot) tile_next(lp, lidx); }
    // right-hand side into this thread's own LDS slots (no other thread reads them before the courier has replaced the chunk); issued behind the
    // first tiles so that both travel in the same memory round trip (the first tile needs t_0 anyway)
    for (int K0 = 0; K0 < nblk; K0 += 4) {
        V tv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = min((K0 + k) * NB + tid * VN, NP - VN);               // unconditional (clamped) loads: exact load counting
            tv[k] = *reinterpret_cast<const V*>(v + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (K0 + k) * NB + tid * VN;
            if (K0 + k < nblk) {
                V x = tv[k];
                if (c >= NP) { T* xp = reinterpret_cast<T*>(&x);
#pragma unroll
                    for (int e = 0; e < VN; ++e) xp[e] = T(0); }
                *reinterpret_cast<V*>(vsh + c) = x;
            }
        }
    }
    int cp = 0, cidx = 0;
    for (int s0 = 0; s0 < ttot; s0 += W) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (s0 + w < ttot) { consume(buf[w], cp, cidx); tile_next(cp, cidx); }
            load(buf[w], lp, lidx);                                              // past the end: the last tile again, never used
            if (++issued < ttot) tile_next(lp, lidx);
        }
    }
}

// Per-device facts, keyed by device ordinal (a process may hold handles on several devices, include/qps.h:19): 0 = not asked yet,
// 1 = a launch of 256 workgroups with 144 KiB of LDS each is co-resident on this device, 2 = it is not.
std::atomic<int> g_resident[kMaxDevices];
bool launch_is_co_resident(int dev) {
    if (dev < 0 || dev >= kMaxDevices) return false;
    int v = g_resident[dev].load(std::memory_order_acquire);
    if (v == 0) {
        int per_cu = 0;
        const void* kern = reinterpret_cast<const void*>(k_trsv_blocked<double, 512, false, 1>);   // 576 threads, the largest LDS request
        const bool attr = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess;
        const bool occ = attr && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512 + 64, (size_t)144 * 1024) == hipSuccess && per_cu >= 1;
        v = (occ && device_cu_count(dev) >= TBG) ? 1 : 2;
        (void)hipGetLastError();
        g_resident[dev].store(v, std::memory_order_release);
    }
    return v == 1;
}

// All blocked-sweep launches of one device in one chain (see the header comment).
struct SweepGate { std::mutex mu; hipStream_t last = nullptr; hipEvent_t ev = nullptr; };
SweepGate g_gate[kMaxDevices];

}  // namespace

TrsvBlockedPair::TrsvBlockedPair(int device, hipStream_t st) : dev(device >= 0 && device < kMaxDevices ? device : 0), stream(st) {
    SweepGate& gt = g_gate[dev];
    gt.mu.lock();
    if (gt.last && gt.last != st) {
        // the previous pair went to another stream: order this one behind everything that stream holds (the pair included)
        if (!gt.ev && hipEventCreateWithFlags(&gt.ev, hipEventDisableTiming) != hipSuccess) { gt.ev = nullptr; (void)hipGetLastError(); }
        if (gt.ev && hipEventRecord(gt.ev, gt.last) == hipSuccess) (void)hipStreamWaitEvent(st, gt.ev, 0);
        else { (void)hipGetLastError(); (void)hipStreamSynchronize(gt.last); }      // no event to be had: wait on the host instead
    }
}
TrsvBlockedPair::~TrsvBlockedPair() { SweepGate& gt = g_gate[dev]; gt.last = stream; gt.mu.unlock(); }
void trsv_blocked_forget_stream(int device, hipStream_t st) {
    if (device < 0 || device >= kMaxDevices) return;
    SweepGate& gt = g_gate[device];
    std::lock_guard<std::mutex> lk(gt.mu);
    if (gt.last == st) gt.last = nullptr;      // the stream is idle (its owner synchronised it) and may be destroyed
}

// nb must be the width one row of 512 (256) threads covers with 16-byte loads; at least two blocks; operands fit the 160 KiB of LDS
template <typename T> bool trsv_blocked_supported(int NP, int nb) {
    static const int off = [] { const char* e = getenv("QPS_SWEEP_BLOCKED"); return (e && atoi(e) == 0) ? 1 : 0; }();
    if (off) return false;
    constexpr int VN = VecOf<T>::N;
    if (nb != 512 * VN && nb != 256 * VN) return false;
    const int nblk = (NP + nb - 1) / nb;
    if (nblk < 2 || NP < 2 * VN) return false;
    if ((size_t)nblk * nb * sizeof(T) > (size_t)144 * 1024) return false;
    return launch_is_co_resident(current_device());
}
template <typename T> int64_t trsv_blocked_pub_words(int NP) { return (int64_t)(sizeof(T) / 4) * NP; }

template <typename T>
void trsv_blocked(hipStream_t st, bool bwd, const T* S, int64_t ld, int NP, int nb, const T* v, T* out, unsigned long long* pub,
                  unsigned epoch, unsigned* abort_word, int mode) {
    constexpr int VN = VecOf<T>::N;
    const int nblk = (NP + nb - 1) / nb;
    const size_t lds = (size_t)nblk * nb * sizeof(T);
    static const unsigned spin_limit = [] { const char* e = getenv("QPS_SWEEP_SPIN_LIMIT"); return e ? (unsigned)atol(e) : 200000u; }();
    const LaunchTiming lt = g_launch_timing;
    g_launch_timing = LaunchTiming();
    const int dev_ = current_device();
#define QPS_TB(STREAM, BWDV, WIN)                                                                                                       \
    do {                                                                                                                                \
        auto kern = k_trsv_blocked<T, STREAM, BWDV, WIN>;                                                                               \
        static PerDeviceOnce attr_done;   /* the attribute lives in the device's code object: once per device ordinal */                \
        attr_done.once(dev_, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); }); \
        if (lt.start) hipExtLaunchKernelGGL(kern, dim3(TBG), dim3(STREAM + 64), lds, st, lt.start, lt.stop, 0, S, ld, NP, v, out, pub, (int64_t)NP, epoch, abort_word, spin_limit, mode); \
        else hipLaunchKernelGGL(kern, dim3(TBG), dim3(STREAM + 64), lds, st, S, ld, NP, v, out, pub, (int64_t)NP, epoch, abort_word, spin_limit, mode); \
    } while (0)
    // window: tiles in flight per streaming thread (RB rows x 16 B each); RB = nb / 256
    // window = row tiles in flight per streaming thread.  Deeper is NOT better: the courier's polls queue behind the CU's own outstanding
    // matrix loads, so every tile of window lengthens every hop (n = 4096 fp64, nb = 1024: 22.4 / 19.6 / 19.3 / 18.0 / 16.1 us per sweep at
    // 6 / 4 / 3 / 2 / 1 tiles of 64 B per thread; with one tile the eight waves of a CU still keep 32 KB in flight).  Long sweeps (many
    // tiles per hop) want a little more: n = 16384 fp32 144 / 127 / 114 us at 1 / 2 / 3 tiles (tests/tools/micro/trsv_blocked_bench.hip).
    static const int wenv = [] { const char* e = getenv("QPS_SWEEP_WINDOW"); return e ? atoi(e) : 0; }();
    int win = nblk <= 4 ? 1 : (VN == 2 ? 2 : 3);
    if (nb == 256 * VN && VN == 2) win = 3;          // fp64 nb = 512: 32-byte tiles
    if (wenv > 0) win = wenv;
#define QPS_TBW(STREAM, WIN) do { if (bwd) QPS_TB(STREAM, true, WIN); else QPS_TB(STREAM, false, WIN); } while (0)
#define QPS_TBS(STREAM) do { if (win <= 1) QPS_TBW(STREAM, 1); else if (win == 2) QPS_TBW(STREAM, 2); else if (win == 3) QPS_TBW(STREAM, 3); else QPS_TBW(STREAM, 4); } while (0)
    if (nb == 512 * VN) QPS_TBS(512); else QPS_TBS(256);
#undef QPS_TBS
#undef QPS_TBW
#undef QPS_TB
}

#define INST(T)                                                     \
    template bool trsv_blocked_supported<T>(int, int);              \
    template int64_t trsv_blocked_pub_words<T>(int);                \
    template void trsv_blocked<T>(hipStream_t, bool, const T*, int64_t, int, int, const T*, T*, unsigned long long*, unsigned, unsigned*, int);
INST(double)
INST(float)
#undef INST

}  // namespace qps
