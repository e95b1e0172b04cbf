// qps_internal.h -- shared host-side plumbing of libqps_hip (not part of the public boundary).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/qps.h"

namespace qps {

struct QpsError { int code; std::string msg; QpsError(int c, std::string m) : code(c), msg(std::move(m)) {} };

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int roundup(int64_t v, int q) { return (int)(((v + q - 1) / q) * q); }

struct KernelStat { double seconds = 0; int64_t launches = 0; double algo_bytes = 0; };

// Brackets launches of one category with HIP events on the solver stream; elapsed times are harvested at the next
// host synchronisation (every numItrConv iterations), so the loop itself is not stalled.
struct Profiler {
    int level = 0;
    hipStream_t st = nullptr;
    struct Pending { int cat; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::vector<std::string> names;
    std::vector<KernelStat> stats;
    int category(const char* name, double bytes) {
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == name) return (int)i;
        names.push_back(name); KernelStat s; s.algo_bytes = bytes; stats.push_back(s); return (int)names.size() - 1;
    }
    hipEvent_t get() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; (void)hipEventCreate(&e); return e; }
    bool on(int cat_level) const { return level >= cat_level; }
    void begin(int cat, hipEvent_t& a) { a = get(); (void)hipEventRecord(a, st); (void)cat; }
    void end(int cat, hipEvent_t a) { hipEvent_t b = get(); (void)hipEventRecord(b, st); pending.push_back({cat, a, b}); }
    void harvest() {   // call after a stream synchronisation
        for (auto& p : pending) {
            float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
            stats[p.cat].seconds += ms * 1e-3; stats[p.cat].launches += 1;
            pool.push_back(p.a); pool.push_back(p.b);
        }
        pending.clear();
    }
    void reset() { for (auto& s : stats) { s.seconds = 0; s.launches = 0; } }
    void release_events() {   // the stream may outlive this profiler (recycled): leave no event behind on it
        for (auto e : pool) (void)hipEventDestroy(e);
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        pool.clear(); pending.clear();
    }
    ~Profiler() { for (auto e : pool) (void)hipEventDestroy(e); for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); } }
};
// Times ONE kernel dispatch by its own timestamps: the launcher called inside the scope consumes g_launch_timing (qps_kernels.h).
// If no launcher consumed it (a code path without support), the sample is dropped.
struct ProfLaunchScope {
    Profiler& p; int cat; hipEvent_t a, b; bool active;
    ProfLaunchScope(Profiler& pr, int c, int lvl);
    ~ProfLaunchScope();
};
struct ProfScope {
    Profiler& p; int cat; hipEvent_t a; bool active;
    ProfScope(Profiler& pr, int c, int lvl) : p(pr), cat(c), a(nullptr), active(pr.on(lvl)) { if (active) p.begin(cat, a); }
    ~ProfScope() { if (active) p.end(cat, a); }
};

#define HIPC(expr)                                                                                            \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw QpsError(e_ == hipErrorOutOfMemory ? QPS_ERR_OUT_OF_MEMORY : QPS_ERR_HIP, b_);               \
        }                                                                                                     \
    } while (0)

// Stream-ordering rule of the library: a handle works on its own NON-BLOCKING stream, which is not ordered against the null stream.
// Nothing a handle's kernels read is therefore ever written through the null stream: zero fills are hipMemsetAsync on the
// handle's stream, host data arrives through StagedUploader (pinned staging + hipMemcpyAsync on the same stream), and every later
// kernel of the handle is enqueued on that stream too -- the order "fill -> upload -> first use" is the stream's own order, no
// device-wide synchronisation stands in for it.  (Round 1 zero-filled with hipMemset and uploaded with synchronous hipMemcpy
// from pageable memory: both run on the null stream and may still be in flight when the call returns.  Without the waits that
// were bolted on afterwards, a fill could land after an import kernel of the solver stream had written the same buffer -- the
// "Cholesky breakdown on a well-conditioned matrix" seen once -- and the first SpMV could read CSR arrays whose DMA had not
// finished -- the run-to-run differences of CG iteration counts.)
template <typename T> inline T* dalloc(int64_t count, hipStream_t st) {
    T* p = nullptr; if (count < 64) count = 64;
    HIPC(hipMalloc((void**)&p, sizeof(T) * (size_t)count));
    HIPC(hipMemsetAsync(p, 0, sizeof(T) * (size_t)count, st));
    return p;
}

// Pageable host memory -> device, ordered on `st`: the bytes travel through a pinned double buffer; a half is reused only after the
// copy that read it has completed (its event), so the caller's source may be freed as soon as copy() returns.
struct StagedUploader {
    hipStream_t st; char* pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
    size_t cap; int cur = 0;
    explicit StagedUploader(hipStream_t s, size_t cap_ = (size_t)8 << 20) : st(s), cap(cap_) {}
    StagedUploader(const StagedUploader&) = delete;
    ~StagedUploader() {
        for (int k = 0; k < 2; ++k) {
            if (busy[k]) (void)hipEventSynchronize(ev[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (pin[k]) (void)hipHostFree(pin[k]);
        }
    }
    void copy(void* dst, const void* src, size_t bytes) {
        const char* s_ = static_cast<const char*>(src); char* d_ = static_cast<char*>(dst);
        while (bytes > 0) {
            const int k = cur; cur ^= 1;
            if (!pin[k]) { HIPC(hipHostMalloc((void**)&pin[k], cap)); HIPC(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
            if (busy[k]) { HIPC(hipEventSynchronize(ev[k])); busy[k] = false; }
            const size_t nb = bytes < cap ? bytes : cap;
            memcpy(pin[k], s_, nb);
            HIPC(hipMemcpyAsync(d_, pin[k], nb, hipMemcpyHostToDevice, st));
            HIPC(hipEventRecord(ev[k], st)); busy[k] = true;
            s_ += nb; d_ += nb; bytes -= nb;
        }
    }
};

// ---- the boundary's host-buffer hand-over (qps_create_*): large pageable arrays -> device -----------------------------------------------------------------
// hipMemcpyAsync from pageable memory moves ~3-10 GB/s (the runtime stages through its own small pinned buffers on one thread): at BASELINE's headline size the
// hand-over of P and A (0.4 GB) plus their validation took 145 ms, against 64 ms for the whole solve to eps.  Here several host threads copy slices of the source
// into a per-device pinned ring (two halves, cached per device like streams are) and every half travels as ONE hipMemcpyAsync on the handle's stream while the
// threads fill the other half.
int host_threads();                                                            // QPS_HOST_THREADS, default min(8, hardware threads)
// f(t, begin, end) on `parts` slices of [0, count) run on host threads (the calling thread takes the first slice); small jobs stay on the caller
template <typename F> inline void host_parallel(int64_t count, int64_t min_per_thread, F&& f) {
    int nt = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, count / std::max<int64_t>(1, min_per_thread)));
    if (nt <= 1) { f(0, (int64_t)0, count); return; }
    std::vector<std::thread> th;
    const int64_t per = (count + nt - 1) / nt;
    int started = 1;                                     // slices [started, nt) that found no thread are run by the caller (thread creation may be refused)
    try { for (int t = 1; t < nt; ++t) { th.emplace_back([&, t] { f(t, std::min(count, t * per), std::min(count, (t + 1) * per)); }); started = t + 1; } }
    catch (const std::system_error&) {}
    f(0, (int64_t)0, std::min(count, per));
    for (int t = started; t < nt; ++t) f(t, std::min(count, t * per), std::min(count, (t + 1) * per));
    for (auto& x : th) x.join();
}
struct PinnedRing { char* base = nullptr; size_t half = 0; };
PinnedRing acquire_ring(int device);                                           // a cached 2 x 32 MiB pinned block (allocated on first use)
void recycle_ring(int device, PinnedRing r);
struct FastUploader {
    hipStream_t st; int device; PinnedRing ring; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; int cur = 0;
    FastUploader(hipStream_t s, int dev) : st(s), device(dev) { ring = acquire_ring(dev); }
    FastUploader(const FastUploader&) = delete;
    ~FastUploader() {
        for (int k = 0; k < 2; ++k) { if (busy[k]) (void)hipEventSynchronize(ev[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
        recycle_ring(device, ring);
    }
    // contiguous bytes; returns when the last slice has been handed to the stream (the source may then be freed: every byte sits in pinned memory or on the device)
    void copy(void* dst, const void* src, size_t bytes) {
        const char* s_ = static_cast<const char*>(src); char* d_ = static_cast<char*>(dst);
        while (bytes > 0) {
            const int k = cur; cur ^= 1;
            if (!ev[k]) HIPC(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
            if (busy[k]) { HIPC(hipEventSynchronize(ev[k])); busy[k] = false; }
            const size_t nb = std::min(bytes, ring.half);
            char* pin = ring.base + (size_t)k * ring.half;
            host_parallel((int64_t)nb, (int64_t)2 << 20, [&](int, int64_t b, int64_t e) { if (e > b) memcpy(pin + b, s_ + b, (size_t)(e - b)); });
            HIPC(hipMemcpyAsync(d_, pin, nb, hipMemcpyHostToDevice, st));
            HIPC(hipEventRecord(ev[k], st)); busy[k] = true;
            s_ += nb; d_ += nb; bytes -= nb;
        }
    }
};

// One device allocation per handle: thirty-odd hipMalloc + zero-fill + hipFree calls cost ~4.5 ms per handle, ten times the
// solve itself at the reference's test sizes.  Lay the buffers out twice: once against a null base to add up the sizes, once for real.
struct Arena {
    char* base = nullptr; size_t off = 0; bool planning = true;
    template <typename T> T* take(int64_t count) {
        if (count < 64) count = 64;
        off = (off + 255) & ~(size_t)255;
        T* p = planning ? nullptr : reinterpret_cast<T*>(base + off);
        off += sizeof(T) * (size_t)count;
        return p;
    }
    size_t bytes = 0;
    size_t planned() const { return ((off + 255) & ~(size_t)255) + 256; }
    // allocate what the planning pass added up (or adopt a recycled block of at least that size), zero it, restart the layout for real
    void commit(hipStream_t st, char* recycled = nullptr, size_t recycled_bytes = 0) {
        const size_t need = planned();
        if (recycled) { base = recycled; bytes = recycled_bytes; }
        else { HIPC(hipMalloc((void**)&base, need)); bytes = need; }
        HIPC(hipMemsetAsync(base, 0, need, st));   // on the handle's stream: ordered before everything the handle enqueues later
        off = 0; planning = false;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; }
};

// Per-device recycling of what a dense handle needs besides its data: the stream, the pinned read-back block and (for small
// problems) the device block itself.  hipStreamCreate / hipHostMalloc / hipMalloc and their counterparts cost ~3 ms per handle --
// five times a whole solve at the reference's test sizes, where every SolveQuadraticProgram! call builds and drops a handle.
struct HandleResources { hipStream_t st = nullptr; void* pinned = nullptr; char* block = nullptr; size_t block_bytes = 0; };
HandleResources acquire_resources(int device, size_t block_need);      // block may come back null (caller allocates)
void recycle_resources(int device, HandleResources r);                  // stream must be idle

struct SolverBase {
    int device = 0; hipStream_t st = nullptr; int dtype = 0; int64_t n = 0, m = 0; bool sparse = false;
    std::string err; Profiler prof;
    virtual ~SolverBase() {}
    virtual void solve(double* x, const qps_params& p, qps_info* info) = 0;
    virtual void get_dual(double* z, double* y) = 0;
    virtual void polish(double* x, const double* y, const qps_params& p, qps_polish_report* rep) { (void)x; (void)y; (void)p; (void)rep; throw QpsError(QPS_ERR_UNSUPPORTED, "polishing is not implemented for this handle type"); }
    virtual void linsys_init(double rho, double sigma, int linsys, int nb) = 0;
    virtual void linsys_set_cg(double eps_pcg, int num_itr_pcg) { (void)eps_pcg; (void)num_itr_pcg; }   // direct plugins: no inner iteration
    virtual void linsys_solve(const double* x, const double* z, const double* y, double rho, double sigma, int changed,
                              double* xx, double* zz) = 0;
    virtual void operator_apply(int op, const double* in, double* out, double rho, double sigma) {   // qps_operator_apply
        (void)op; (void)in; (void)out; (void)rho; (void)sigma; throw QpsError(QPS_ERR_UNSUPPORTED, "qps_operator_apply is not implemented for this handle type");
    }
};


SolverBase* make_sparse_solver(int device, int64_t n, int64_t m, int dtype, const int64_t* Pcp, const int64_t* Pri,
                               const double* Pnz, const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* q,
                               const double* l, const double* u, int index_base);

}  // namespace qps
