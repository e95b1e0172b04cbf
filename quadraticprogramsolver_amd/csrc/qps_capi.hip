// qps_capi.hip -- the C ABI of include/qps.h: handle table, problem import, the device-resident ADMM loop driver.
//
// Loop structure follows SolveQuadraticProgram.jl:36-73; the linear solve is the reduced form of
// LinearSystemSolvers.jl:110-142 with cg! replaced by Cholesky + two triangular sweeps (ProxQP.jl:175-206,221-225).
// There is NO CPU fallback: without a HIP device every entry point fails with QPS_ERR_NO_DEVICE.
#include "ldl_symbolic.h"
#include <atomic>
#include <thread>

#include "batch_schedule.h"
#include "qps_internal.h"
#include "spmv_layout.h"
#include "qps_kernels.h"
#include "qps_polish.h"
#include "qps_proxqp.h"

using namespace qps;

namespace qps {
namespace {
constexpr size_t kPinnedBytes = 512, kRecycleBlockMax = (size_t)512 << 20;   // blocks above 512 MiB go back to the driver (at most 4 x that per device stay cached, of 288 GB)
constexpr int kRecyclePerDevice = 4;
std::mutex g_res_mu;
std::vector<HandleResources> g_res[kMaxDevices];
std::atomic<int> g_cus[kMaxDevices];   // 0 = not asked yet
}  // namespace
int current_device() { int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; } return dev; }
int device_cu_count(int device) {
    if (device < 0 || device >= kMaxDevices) return 0;
    int c = g_cus[device].load(std::memory_order_acquire);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); c = -1; }
        g_cus[device].store(c, std::memory_order_release);
    }
    return c > 0 ? c : 0;
}
HandleResources acquire_resources(int device, size_t block_need) {
    HandleResources r;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (device >= 0 && device < kMaxDevices && !g_res[device].empty()) { r = g_res[device].back(); g_res[device].pop_back(); }
    }
    if (!r.st) HIPC(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
    if (!r.pinned) HIPC(hipHostMalloc(&r.pinned, kPinnedBytes));
    if (r.block && (r.block_bytes < block_need || r.block_bytes > 4 * block_need + ((size_t)1 << 20))) { (void)hipFree(r.block); r.block = nullptr; r.block_bytes = 0; }
    return r;
}
void recycle_resources(int device, HandleResources r) {
    if (r.block && r.block_bytes > kRecycleBlockMax) { (void)hipFree(r.block); r.block = nullptr; r.block_bytes = 0; }
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (device >= 0 && device < kMaxDevices && (int)g_res[device].size() < kRecyclePerDevice) { g_res[device].push_back(r); return; }
    }
    if (r.block) (void)hipFree(r.block);
    if (r.pinned) (void)hipHostFree(r.pinned);
    if (r.st) (void)hipStreamDestroy(r.st);
}
int host_threads() {
    static const int n = [] {
        const char* e = getenv("QPS_HOST_THREADS");
        const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
        return std::max(1, std::min(e ? atoi(e) : 8, hw));
    }();
    return n;
}
namespace {
constexpr size_t kRingHalf = (size_t)32 << 20;
std::vector<PinnedRing> g_rings[kMaxDevices];
}  // namespace
PinnedRing acquire_ring(int device) {
    PinnedRing r;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (device >= 0 && device < kMaxDevices && !g_rings[device].empty()) { r = g_rings[device].back(); g_rings[device].pop_back(); }
    }
    if (!r.base) { HIPC(hipHostMalloc((void**)&r.base, 2 * kRingHalf)); r.half = kRingHalf; }
    return r;
}
void recycle_ring(int device, PinnedRing r) {
    if (!r.base) return;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (device >= 0 && device < kMaxDevices && (int)g_rings[device].size() < kRecyclePerDevice) { g_rings[device].push_back(r); return; }
    }
    (void)hipHostFree(r.base);
}
thread_local LaunchTiming g_launch_timing;
ProfLaunchScope::ProfLaunchScope(Profiler& pr, int c, int lvl) : p(pr), cat(c), a(nullptr), b(nullptr), active(pr.on(lvl)) {
    if (active) { a = p.get(); b = p.get(); g_launch_timing.start = a; g_launch_timing.stop = b; }
}
ProfLaunchScope::~ProfLaunchScope() {
    if (!active) return;
    const bool consumed = (g_launch_timing.start == nullptr);
    g_launch_timing = LaunchTiming();
    if (consumed) p.pending.push_back({cat, a, b}); else { p.pool.push_back(a); p.pool.push_back(b); }
}
}  // namespace qps

namespace {

thread_local std::string g_last_error;

int pick_nb(int requested, int NP, bool cover = false) {
    // default: one inverted block over the whole factor while the fused forward+backward sweep covers NP (`cover`), else 4096
    int nb = requested > 0 ? requested : (cover ? 32768 : 4096);
    int p = 64; while (p * 2 <= nb) p *= 2;   // power-of-two multiple of 64
    nb = p;
    while (nb > 64 && nb / 2 >= NP) nb /= 2;
    return nb;
}

// =================================================================================================================
// Dense problem, Cholesky path
// =================================================================================================================
template <typename T> struct DenseSolver : SolverBase {
    int NP = 0, MP = 0;
    T *A = nullptr, *P = nullptr, *q = nullptr, *l = nullptr, *u = nullptr;
    T *PI = nullptr, *AA = nullptr, *M = nullptr, *S = nullptr, *tmp = nullptr, *dinv = nullptr; int* fail = nullptr;
    T *x = nullptr, *xp = nullptr, *z = nullptr, *zp = nullptr, *y = nullptr, *xx = nullptr, *zz = nullptr, *tt = nullptr, *yv = nullptr;
    T *part = nullptr, *part2 = nullptr, *sw_part = nullptr, *Ax = nullptr, *Px = nullptr, *Aty = nullptr;
    T* At = nullptr; void* small_out = nullptr; void* small_out_host = nullptr; bool small_ok = false;   // small-problem path
    Arena arena; HandleResources res; double* stage_mat = nullptr; int64_t stage_mat_count = 0;
    int pass_slabs = 0, pass_rpw = 0;   // fused-pass plan (0 slabs: shape not supported, unfused loop only)
    unsigned long long* scratch = nullptr; double* res_dev = nullptr; double* res_host = nullptr; double* stage = nullptr;
    bool have_AA = false, factor_valid = false; double fac_rho = 0, fac_sigma = 0; int fac_nb = 0;
    int nb = 2048; int part_tiles = 0; int num_factorizations = 0;
    // single-launch blocked sweeps (k_trsv_blocked.hip): hand-off granules, launch epoch, give-up word; `blocked_off` while the solve whose
    // launch gave up is being repeated on the multi-launch substitution (the next solve tries the blocked sweeps again)
    unsigned long long* pub = nullptr; unsigned* abort_dev = nullptr; unsigned sweep_epoch = 0; bool blocked_off = false, fac_premul = false;
    bool use_blocked() const { return !blocked_off && trsv_blocked_supported<T>(NP, nb); }
    // hipGraph replay of runs of plain iterations (no check, no rho switch) for problems small enough to be launch bound
    struct IterGraph { int count; const void* xa; double rho, sigma, alpha; int nb; hipGraphExec_t exec; };
    std::vector<IterGraph> graphs;
    bool graphs_disabled = false;
    void drop_graphs() { for (auto& gr : graphs) (void)hipGraphExecDestroy(gr.exec); graphs.clear(); }
    // `count` (even) plain iterations of the fused loop starting with x in `xa`, xp in `xb`; captured once per (count, roles, scalars)
    hipGraphExec_t iter_graph(int count, T* xa, T* xb, double rho, double sigma, double alpha) {
        for (auto& gr : graphs) if (gr.count == count && gr.xa == xa && gr.rho == rho && gr.sigma == sigma && gr.alpha == alpha && gr.nb == nb) return gr.exec;
        if (graphs_disabled) return nullptr;
        if (graphs.size() >= 8) drop_graphs();
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        for (int k = 0; k < count; ++k) {
            colsum<T>(st, part, NP, pass_slabs, xa, (T)sigma, q, T(-1), tt, NP);                    // LinearSystemSolvers.jl:136
            sweeps();                                                                               // :137 (profiling is off here: plain launches)
            apass<T>(st, false, A, NP, NP, MP, xx, xa, xb, z, y, l, u, (T)alpha, (T)rho, part, part2, NP, scratch);   // :56-61, :139, next :134-135
            std::swap(xa, xb);
        }
        // nothing was enqueued while capturing, so a failure here just means: run this handle's iterations eagerly from now on
        if (hipStreamEndCapture(st, &graph) != hipSuccess || graph == nullptr) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { (void)hipGetLastError(); graphs_disabled = true; return nullptr; }
        graphs.push_back({count, xa, rho, sigma, alpha, nb, exec});   // (xa is back in its original role: count is even)
        return exec;
    }
    int prof_iter = 0;   // iteration index seen by the level-1 sampler (one bracketed launch per kernel kind per 50 iterations)
    int sample_lvl(int slot) const { return (prof.level == 1 && prof_iter % 50 == slot) ? 1 : 2; }
    int cat_atw, cat_colsum, cat_fwd, cat_bwd, cat_ax, cat_upd, cat_chk, cat_pass, cat_passchk, cat_sweep, cat_xsum, cat_small;

    DenseSolver(int dev, int64_t n_, int64_t m_, int dt) {
        device = dev; n = n_; m = m_; dtype = dt;
        HIPC(hipSetDevice(device));
        NP = roundup(n, 64); MP = roundup(m, 64);
        const int64_t nn = (int64_t)NP * NP;
        part_tiles = gemv_cols_tiles(MP);
        pass_slabs = apass_plan<T>(NP, MP, &pass_rpw);
        const int slabs = std::max(std::max(part_tiles, pass_slabs), 1);
        small_ok = admm_small_supported<T>((int)n, (int)m, NP, MP);
        // staging area of the matrix import (column-major doubles): inside the arena when the matrices are small
        stage_mat_count = std::max((int64_t)n * n, (int64_t)m * n);
        if (stage_mat_count > ((int64_t)1 << 20)) stage_mat_count = 0;
        auto layout = [&](Arena& ar) {
            A = ar.take<T>((int64_t)MP * NP); P = ar.take<T>(nn); q = ar.take<T>(NP); l = ar.take<T>(MP); u = ar.take<T>(MP);
            PI = ar.take<T>(nn); AA = ar.take<T>(nn); M = ar.take<T>(nn); S = ar.take<T>(nn); tmp = ar.take<T>(nn);
            dinv = ar.take<T>((int64_t)(NP / 64) * 4096); fail = ar.take<int>(4);
            x = ar.take<T>(NP); xp = ar.take<T>(NP); xx = ar.take<T>(NP); tt = ar.take<T>(NP); yv = ar.take<T>(NP);
            z = ar.take<T>(MP); zp = ar.take<T>(MP); y = ar.take<T>(MP); zz = ar.take<T>(MP);
            part = ar.take<T>((int64_t)slabs * NP);
            part2 = ar.take<T>((int64_t)slabs * NP);
            sw_part = ar.take<T>((int64_t)std::max(sweep_fused_slabs<T>(NP), 1) * NP);
            Ax = ar.take<T>(MP); Px = ar.take<T>(NP); Aty = ar.take<T>(NP);
            scratch = ar.take<unsigned long long>(16); res_dev = ar.take<double>(16);
            pub = ar.take<unsigned long long>(trsv_blocked_pub_words<T>(NP));
            abort_dev = reinterpret_cast<unsigned*>(res_dev + 14);   // rides in the check's read-back (slots 8..15 are unused by the check kernels)
            stage = ar.take<double>((int64_t)NP + 2 * (int64_t)MP + 64);
            if (small_ok) { At = ar.take<T>((int64_t)NP * MP); small_out = ar.take<double>(32); }
            if (stage_mat_count > 0) stage_mat = ar.take<double>(stage_mat_count);
        };
        layout(arena);
        res = acquire_resources(device, arena.planned());                    // stream, pinned block and maybe a recycled device block
        st = res.st; prof.st = st;
        try { arena.commit(st, res.block, res.block_bytes); }
        catch (...) { recycle_resources(device, res); res = HandleResources(); st = nullptr; prof.st = nullptr; throw; }
        res.block = nullptr;                                                  // owned by the arena from here on
        layout(arena);
        res_host = reinterpret_cast<double*>(res.pinned);                     // one pinned block: check results | small-kernel report
        small_out_host = small_ok ? reinterpret_cast<char*>(res.pinned) + 16 * sizeof(double) : nullptr;
        const double s = sizeof(T);
        // algorithmic bytes per launch (SURVEY §8d "per-kernel algorithmic bytes")
        cat_atw = prof.category("gemv_cols(A'w)", s * ((double)m * n + m + n));
        cat_colsum = prof.category("colsum(rhs)", s * ((double)part_tiles * n + 3.0 * n));
        cat_fwd = prof.category("trsv_forward", s * ((double)n * (n + 1) / 2 + 2.0 * n));
        cat_bwd = prof.category("trsv_backward", s * ((double)n * (n + 1) / 2 + 2.0 * n));
        cat_ax = prof.category("gemv_rows(Ax~)", s * ((double)m * n + m + n));
        cat_upd = prof.category("admm_update", s * (3.0 * n + 7.0 * m));
        cat_chk = prof.category("check_convergence", s * (2.0 * m * n + (double)n * n + 4.0 * n + 4.0 * m));
        // fused pass: A once + x~, x, z, y, l, u in, x, z, y out (s*m*n + vector traffic)
        cat_pass = prof.category("apass(fused A-pass)", s * ((double)m * n + 3.0 * n + 6.0 * m));
        cat_passchk = prof.category("apass(check variant)", s * ((double)m * n + 4.0 * n + 6.0 * m));
        // fused forward+backward sweep: the bytes this kernel has to move -- the triangle ONCE (n(n+1)/2) plus its slabs; SURVEY §8d's
        // figure for the two separate sweeps it replaces is twice the triangle (bench.py reports that one as `two_sweep_algo_GBs`)
        cat_sweep = prof.category("sweeps(fused fwd+bwd, triangle read once)", s * ((double)n * (n + 1) / 2 + 2.0 * n + (double)sweep_fused_slabs<T>(NP) * n));
        cat_xsum = prof.category("colsum(x~ slabs)", s * ((double)sweep_fused_slabs<T>(NP) * n + n));
        // single-launch small-problem loop: bytes PER ITERATION (one launch runs many; bench.py multiplies by the iterations it timed)
        cat_small = prof.category("admm_small(whole loop in one launch; bytes per iteration)", s * ((double)m * n + (double)n * n) + s * (6.0 * n + 10.0 * m));
    }
    ~DenseSolver() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        if (st) trsv_blocked_forget_stream(device, st);   // the stream goes back to the pool (or is destroyed): the sweep gate must not record on it
        drop_graphs();
        prof.release_events();
        res.block = arena.base; res.block_bytes = arena.bytes; arena.base = nullptr;
        if (res.st) recycle_resources(device, res); else arena.release();
    }

    // host double array -> device T vector (zero padded allocation is preserved beyond `count`)
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
    // column-major host matrix -> row-major device T (streamed through a bounded staging buffer, column panels)
    void upload_matrix(const double* h, int64_t ldh, int rows, int cols, T* d, int64_t ldd) {
        if (rows <= 0 || cols <= 0) return;
        if (stage_mat && (int64_t)rows * cols <= stage_mat_count) {   // small matrix: one copy through the arena's staging area
            HIPC(hipMemcpy2DAsync(stage_mat, sizeof(double) * (size_t)rows, h, sizeof(double) * (size_t)ldh, sizeof(double) * (size_t)rows, (size_t)cols, hipMemcpyHostToDevice, st));
            import_colmajor<T>(st, stage_mat, rows, rows, cols, d, ldd);
            HIPC(hipStreamSynchronize(st));
            return;
        }
        const int64_t budget = (int64_t)32 << 20;   // doubles per panel (256 MiB)
        int pc = (int)std::max<int64_t>(64, (budget / std::max(rows, 1)) / 64 * 64);
        double* buf = nullptr;
        HIPC(hipMalloc((void**)&buf, sizeof(double) * (size_t)rows * (size_t)std::min(pc, cols)));
        FastUploader fu(st, device);                 // pinned ring filled by several host threads (qps_internal.h)
        for (int c0 = 0; c0 < cols; c0 += pc) {
            const int nc = std::min(pc, cols - c0);
            if (ldh == rows) fu.copy(buf, h + (int64_t)c0 * ldh, sizeof(double) * (size_t)rows * (size_t)nc);
            else for (int c = 0; c < nc; ++c) fu.copy(buf + (int64_t)c * rows, h + (int64_t)(c0 + c) * ldh, sizeof(double) * (size_t)rows);
            import_colmajor<T>(st, buf, rows, rows, nc, d + c0, ldd);
            HIPC(hipStreamSynchronize(st));
        }
        HIPC(hipFree(buf));
    }

    // LinSysSolInit: LinearSystemSolvers.jl:110-122 (mAA, mPI, mL) + factorisation (ProxQP.jl:196)
    void factorize(double rho, double sigma, bool rebuild_all) {
        if (rebuild_all || !have_AA) {
            make_PI<T>(st, (int)n, NP, P, (T)sigma, PI);                                             // :113
            if (MP > 0) gemm<T>(st, NP, NP, MP, T(1), A, NP, false, A, NP, false, T(0), AA, NP, true); // :112 mAA = mA' mA
            else HIPC(hipMemsetAsync(AA, 0, sizeof(T) * (size_t)NP * NP, st));
            have_AA = true;
        }
        assemble_M<T>(st, NP, PI, AA, (T)rho, M);                                                    // :114 / :128
        cholesky<T>(st, NP, M, dinv, fail, 1, chol_scratch_fits(NP) ? S : nullptr);   // S is rebuilt right after: free as scratch
        build_sweep_matrix<T>(st, NP, nb, M, dinv, S, tmp, 1, use_blocked());
        int f = 0;
        HIPC(hipMemcpyAsync(&f, fail, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        ++num_factorizations;
        if (f != 0) {
            factor_valid = false;
            char b[256];
            if (f < 0) snprintf(b, sizeof b, "Cholesky of P + sigma I + rho A'A: the diagonal workgroup of a fused update launch gave up waiting for its two tiles (k_chol_update_diag; rho=%g, factorisation #%d of this handle)", rho, num_factorizations);
            else snprintf(b, sizeof b, "Cholesky of P + sigma I + rho A'A broke down: non-positive pivot at column %d (rho=%g, sigma=%g; factorisation #%d of this handle)", f, rho, sigma, num_factorizations);
            throw QpsError(QPS_ERR_FACTORIZATION, b);
        }
        factor_valid = true; fac_rho = rho; fac_sigma = sigma; fac_nb = nb; fac_premul = use_blocked();
    }
    // Did a blocked-sweep launch give up waiting (its workgroups were not co-resident: only another PROCESS running the same kernel on the card
    // can do that, launches of this process are chained by the sweep gate)?  Then the iterates are garbage: the caller repeats its work on the
    // multi-launch sweeps (`blocked_off` until the next solve / linsys_init).  Synchronises the stream.
    bool sweep_gave_up(bool fetched = false) {
        if (!fac_premul) return false;
        unsigned* h = reinterpret_cast<unsigned*>(res_host + 14);
        if (!fetched) {
            HIPC(hipMemcpyAsync(h, abort_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
        }
        if (*h == 0u) return false;
        HIPC(hipMemsetAsync(abort_dev, 0, sizeof(unsigned), st));
        blocked_off = true; factor_valid = false;
        return true;
    }

    int sweep_variant() const {
        const int nblk = (NP + nb - 1) / nb;
        if (nblk > 1 && fac_premul) return 5;
        static const int sweep_mode = getenv("QPS_SWEEP_MODE") ? atoi(getenv("QPS_SWEEP_MODE")) : 2;
        if (nblk == 1) return (sweep_mode == 2 && sweep_fused_supported<T>(NP)) ? 2 : 3;
        return 1;
    }
    // x~ = (L L')^{-1} tt via the blocked sweeps over S (tt is consumed)
    void sweeps() {
        const int nblk = (NP + nb - 1) / nb;
        static const int sweep_mode = getenv("QPS_SWEEP_MODE") ? atoi(getenv("QPS_SWEEP_MODE")) : 2;
        if (nblk == 1 && sweep_mode == 2 && sweep_fused_supported<T>(NP)) {
            // one inverted block: forward and backward sweep read the same entries -> one fused pass over the triangle
            int G;
            { ProfLaunchScope ps(prof, cat_sweep, sample_lvl(13)); G = sweep_fused<T>(st, S, NP, NP, tt, sw_part, NP); }
            { ProfLaunchScope ps(prof, cat_xsum, sample_lvl(21)); colsum<T>(st, sw_part, NP, G, nullptr, T(0), nullptr, T(0), xx, NP); }
            return;
        }
        if (fac_premul) {
            // blocked substitution, one launch per sweep: n / nb dependent phases handed from workgroup to workgroup inside the launch
            if (sweep_epoch >= 0xfffffff0u) {   // the granule tags are 32-bit launch counters: start over on a cleared buffer
                HIPC(hipMemsetAsync(pub, 0, sizeof(unsigned long long) * (size_t)trsv_blocked_pub_words<T>(NP), st));
                sweep_epoch = 0;
            }
            TrsvBlockedPair gate(device, st);   // never two of these persistent launches on the chip at once (k_trsv_blocked.hip, "Co-residency")
            { ProfLaunchScope ps(prof, cat_fwd, sample_lvl(13)); trsv_blocked<T>(st, false, S, NP, NP, nb, tt, yv, pub, ++sweep_epoch, abort_dev); }
            { ProfLaunchScope ps(prof, cat_bwd, sample_lvl(21)); trsv_blocked<T>(st, true, S, NP, NP, nb, yv, xx, pub, ++sweep_epoch, abort_dev); }
            return;
        }
        {
            ProfScope ps(prof, cat_fwd, 2);
            for (int J = 0; J < nblk; ++J) {
                const int r0 = J * nb, r1 = std::min(NP, r0 + nb);
                gemv_rows<T>(st, S, NP, tt, yv, nullptr, T(1), T(0), r0, r1, r0, r1, 1);
                if (r1 < NP) gemv_rows<T>(st, S, NP, yv, tt, tt, T(-1), T(1), r1, NP, r0, r1, 0);
            }
        }
        {
            ProfScope ps(prof, cat_bwd, 2);
            for (int J = nblk - 1; J >= 0; --J) {
                const int r0 = J * nb, r1 = std::min(NP, r0 + nb);
                gemv_rows<T>(st, S, NP, yv, xx, nullptr, T(1), T(0), r0, r1, r0, r1, 2);
                if (r0 > 0) gemv_rows<T>(st, S, NP, xx, yv, yv, T(-1), T(1), 0, r0, r0, r1, 0);
            }
        }
    }
    // LinSysSol! body: LinearSystemSolvers.jl:134-139
    void linear_solve(double rho, double sigma) {
        {
            ProfScope ps(prof, cat_atw, 1);
            gemv_cols_partial<T>(st, A, NP, z, y, (T)rho, T(-1), part, NP, MP, NP);                  // :134-135
        }
        {
            ProfScope ps(prof, cat_colsum, 2);
            colsum<T>(st, part, NP, part_tiles, x, (T)sigma, q, T(-1), tt, NP);                      // :136
        }
        sweeps();                                                                                   // :137 (Cholesky instead of cg!)
        {
            ProfScope ps(prof, cat_ax, 2);
            gemv_rows<T>(st, A, NP, xx, zz, nullptr, T(1), T(0), 0, MP, 0, NP, 0);                   // :139
        }
    }

    void solve(double* xh, const qps_params& p, qps_info* info) override {
        HIPC(hipSetDevice(device));
        blocked_off = false;                  // a handle whose launch gave up once tries the blocked sweeps again at every new solve
        int gave_up = 0; double t_lost = 0;
        const double t_begin = now_s();
        while (solve_once(xh, p, info)) {     // true: a blocked-sweep launch gave up; x on the host is still the caller's
            ++gave_up; t_lost = now_s() - t_begin;
            const int lvl = prof.level; prof.reset(); prof.level = lvl;   // the aborted attempt's launches are not this solve's kernels
        }
        if (info) { info->sweepGaveUp = gave_up; info->cgExplicit = 0; info->tLoop += t_lost; }   // the repeated work is loop time
    }
    bool solve_once(double* xh, const qps_params& p, qps_info* info) {
        if (p.linsys != QPS_LINSYS_AUTO && p.linsys != QPS_LINSYS_CHOLESKY)
            throw QpsError(QPS_ERR_UNSUPPORTED, "dense handles offer QPS_LINSYS_CHOLESKY only (qps_create_csc with dense_path = 0 for the CG and the sparse L D L' plugins)");
        const double t0 = now_s();
        nb = pick_nb(p.trsvBlock, NP, sweep_fused_supported<T>(NP));
        double rho = p.rho, sigma = p.sigma; const double alpha = p.alpha;
        const double epsAdmm = std::fmin(p.epsAbs, p.epsRel) * 1e-2;                                // SolveQuadraticProgram.jl:34
        int convFlag = QPS_CONV_NUM_ITR;                                                            // :33
        const bool reuse = p.reuseFactor && factor_valid && fac_rho == rho && fac_sigma == sigma && fac_nb == nb && fac_premul == use_blocked();
        if (!reuse) factorize(rho, sigma, !p.reuseFactor || !have_AA || fac_sigma != sigma);        // :36
        upload_vec(xh, x, n);
        HIPC(hipMemsetAsync(xp, 0, sizeof(T) * NP, st));                                            // :38
        HIPC(hipMemsetAsync(z, 0, sizeof(T) * std::max(MP, 64), st));                               // :39
        HIPC(hipMemsetAsync(y, 0, sizeof(T) * std::max(MP, 64), st));                               // :40
        HIPC(hipMemsetAsync(zp, 0, sizeof(T) * std::max(MP, 64), st));                              // :41
        HIPC(hipStreamSynchronize(st));
        const double t1 = now_s();
        double rhorho = rho;                                                                        // :43
        int ii = 0, nref = 0; double tref = 0, resP = NAN, resD = NAN;
        if (small_ok && p.loopVariant == 0 && (NP + nb - 1) / nb == 1) {
            // Small problem: the whole loop :45-71 runs inside one single-workgroup launch; the host only steps in for a rho switch.
            transpose_small<T>(st, A, NP, MP, At);
            int it = 0;
            while (it < p.numIterations) {
                {
                    ProfScope ps(prof, cat_small, 1);   // events around the one launch (a ~3 us launch gap against a kernel of hundreds of us)
                    admm_small<T>(st, (int)n, (int)m, NP, MP, it, p.numIterations, p.numItrConv, p.adptRho, rho, rhorho, sigma, alpha, p.epsAbs,
                                  p.epsRel, epsAdmm, p.fctrRho, A, At, P, S, q, l, u, x, xp, z, y, small_out);
                }
                HIPC(hipMemcpyAsync(small_out_host, small_out, admm_small_out_bytes(), hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                prof.harvest();
                int last = 0, flag = 1, need = 0; double r8[8];
                admm_small_read(small_out_host, &last, &flag, &need, r8);
                it = last; convFlag = flag; rhorho = r8[4];
                if (!std::isnan(r8[0]) || !std::isnan(r8[1])) { resP = r8[0]; resD = r8[1]; }
                if (flag != QPS_CONV_NUM_ITR) break;                                                // :66-68
                if (need && it < p.numIterations) {                                                 // :47-51 (the loop top is never re-entered after the last iteration)
                    rho = rhorho; ++nref;
                    const double ta = now_s();
                    factorize(rho, sigma, false);
                    tref += now_s() - ta;
                }
            }
            ii = it;
            const double t2s = now_s();
            PolishReport prs;
            if (p.polish) polish_dense<T>(st, n, m, NP, MP, P, A, q, l, u, y, x, part, p, &prs);   // SolveQuadraticProgram.m:289-325
            download_vec(x, xh, n);
            if (info) {
                info->convFlag = convFlag; info->iterations = ii; info->numRefactor = nref; info->cgIterations = 0;
                info->rhoFinal = rho; info->rhoProposed = rhorho; info->resPrim = resP; info->resDual = resD;
                info->tSetup = t1 - t0; info->tLoop = t2s - t1; info->tRefactor = tref;
                info->polishFlag = prs.flag; info->polishIterations = prs.minresIterations; info->tPolish = prs.seconds;
                info->trsvBlock = nb; info->sweepVariant = 4;
            }
            return false;
        }
        const bool fused = pass_slabs > 0 && p.loopVariant != 1;
        int rhs_slabs = 0;   // z = y = 0: A'(rho z - y) = 0, no slab to add for the first right-hand side
        // Launch-bound sizes (an iteration of four to eight kernels lasts less than the host needs to enqueue it): replay graphs.
        // Profiling brackets launches and stays eager.
        static const int graph_env = [] { const char* e = getenv("QPS_GRAPH"); return e ? atoi(e) : -1; }();
        // (the blocked sweeps take a fresh epoch argument per launch: no replay)
        const bool use_graph = fused && prof.level == 0 && p.numItrConv >= 3 && !fac_premul && (graph_env >= 0 ? graph_env != 0 : NP <= 2048);
        for (ii = 1; ii <= p.numIterations; ++ii) {                                                 // :45
            bool changed = false;
            if (p.adptRho && ((rhorho * p.fctrRho < rho) || (rhorho > p.fctrRho * rho))) {          // :47
                rho = rhorho; ++nref; changed = true;                                               // :48-51
                const double ta = now_s();
                factorize(rho, sigma, false);                                                       // changedΡ: LinearSystemSolvers.jl:127-129
                tref += now_s() - ta;
            }
            const bool check = (ii % p.numItrConv == 0);                                            // :63
            if (use_graph && !changed && !check && rhs_slabs == pass_slabs) {
                // the plain iterations up to the next check (an even number of them, so x / xp end in the same roles) as ONE graph launch
                int run = std::min(p.numItrConv - ii % p.numItrConv, p.numIterations - ii + 1) & ~1;
                hipGraphExec_t ge = run >= 2 ? iter_graph(run, x, xp, rho, sigma, alpha) : nullptr;
                if (ge) {
                    HIPC(hipGraphLaunch(ge, st));
                    ii += run - 1;
                    continue;
                }
            }
            if (!fused) {
                linear_solve(rho, sigma);                                                           // :54
                {
                    ProfScope ps(prof, cat_upd, 2);
                    admm_update<T>(st, NP, MP, xx, zz, x, xp, z, zp, y, l, u, (T)alpha, (T)rho);     // :56-61
                }
                if (check) {
                    ProfScope ps(prof, cat_chk, 2);
                    gemv_rows<T>(st, A, NP, x, Ax, nullptr, T(1), T(0), 0, MP, 0, NP, 0);            // mA * vX
                    gemv_rows<T>(st, P, NP, x, Px, nullptr, T(1), T(0), 0, NP, 0, NP, 0);            // mP * vX
                    gemv_cols_partial<T>(st, A, NP, y, nullptr, T(1), T(0), part, NP, MP, NP);       // mA' * vY
                    colsum<T>(st, part, NP, part_tiles, nullptr, T(0), nullptr, T(0), Aty, NP);
                    CheckScalars cs{p.epsAbs, p.epsRel, epsAdmm, rho, rhorho, p.adptRho, convFlag};
                    check_convergence<T>(st, (int)n, (int)m, Ax, Px, Aty, q, x, xp, z, zp, scratch, res_dev, cs);   // :64
                }
            } else {
                // Fused loop: the pass of iteration ii-1 already left the slabs of A'(rho z - y); after a rho switch they
                // are stale (w depends on rho) and are rebuilt by the plain column GEMV.
                if (changed && rhs_slabs > 0) {
                    ProfScope ps(prof, cat_atw, 2);
                    rhs_slabs = gemv_cols_partial<T>(st, A, NP, z, y, (T)rho, T(-1), part, NP, MP, NP);
                }
                prof_iter = ii;
                {
                    ProfLaunchScope ps(prof, cat_colsum, sample_lvl(29));
                    colsum<T>(st, part, NP, rhs_slabs, x, (T)sigma, q, T(-1), tt, NP);              // LinearSystemSolvers.jl:136
                }
                sweeps();                                                                           // :137
                if (check) HIPC(hipMemsetAsync(scratch, 0, 16 * sizeof(unsigned long long), st));
                {
                    // level 1 samples the dominant kernel on every 50th iteration only (an event pair per launch costs ~7 % of
                    // the loop, one in ten still ~3 %); level 2 brackets every launch of every kernel
                    const int lvl = (prof.level == 1 && (check || ii % 50 != 38)) ? 3 : 1;   // mid-chunk sample of the plain variant only
                    ProfLaunchScope ps(prof, check ? cat_passchk : cat_pass, lvl);   // the dispatch's own begin / end timestamps
                    apass<T>(st, check, A, NP, NP, MP, xx, x, xp, z, y, l, u, (T)alpha, (T)rho, part, part2, NP, scratch);
                }
                std::swap(x, xp);   // x now holds the relaxed iterate, xp the previous one (SolveQuadraticProgram.jl:56-57)
                rhs_slabs = pass_slabs;
                if (check) {
                    ProfScope ps(prof, cat_chk, 2);
                    colsum<T>(st, part2, NP, pass_slabs, nullptr, T(0), nullptr, T(0), Aty, NP);     // mA' * vY
                    gemv_rows<T>(st, P, NP, x, Px, nullptr, T(1), T(0), 0, NP, 0, NP, 0);            // mP * vX
                    CheckScalars cs{p.epsAbs, p.epsRel, epsAdmm, rho, rhorho, p.adptRho, convFlag};
                    check_convergence<T>(st, (int)n, (int)m, Ax, Px, Aty, q, x, xp, z, zp, scratch, res_dev, cs, 1);
                }
            }
            if (check) {
                HIPC(hipMemcpyAsync(res_host, res_dev, 15 * sizeof(double), hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                prof.harvest();
                if (sweep_gave_up(true)) return true;
                resP = res_host[0]; resD = res_host[1]; rhorho = res_host[4]; convFlag = (int)res_host[5];
                if (convFlag != QPS_CONV_NUM_ITR) break;                                            // :66-68
            }
        }
        HIPC(hipStreamSynchronize(st));
        prof.harvest();
        if (sweep_gave_up()) return true;
        const double t2 = now_s();
        PolishReport pr;
        if (p.polish) polish_dense<T>(st, n, m, NP, MP, P, A, q, l, u, y, x, part, p, &pr);        // SolveQuadraticProgram.m:289-325
        download_vec(x, xh, n);
        if (info) {
            info->convFlag = convFlag; info->iterations = ii > p.numIterations ? p.numIterations : ii;
            info->numRefactor = nref; info->cgIterations = 0; info->rhoFinal = rho; info->rhoProposed = rhorho;
            info->resPrim = resP; info->resDual = resD; info->tSetup = t1 - t0; info->tLoop = t2 - t1; info->tRefactor = tref;
            info->polishFlag = pr.flag; info->polishIterations = pr.minresIterations; info->tPolish = pr.seconds;
            info->trsvBlock = nb; info->sweepVariant = sweep_variant();
        }
        return false;
    }
    void polish(double* xh, const double* yh, const qps_params& p, qps_polish_report* rep) override {
        HIPC(hipSetDevice(device));
        upload_vec(xh, x, n); upload_vec(yh, y, m);
        PolishReport pr;
        polish_dense<T>(st, n, m, NP, MP, P, A, q, l, u, y, x, part, p, &pr);
        download_vec(x, xh, n);
        if (rep) {
            rep->flag = pr.flag; rep->refinements = pr.refinements; rep->minresIterations = pr.minresIterations;
            rep->numActiveLower = pr.numLower; rep->numActiveUpper = pr.numUpper; rep->reserved0 = 0; rep->relres = pr.relres; rep->seconds = pr.seconds;
        }
    }
    void get_dual(double* zh, double* yh) override {
        HIPC(hipSetDevice(device));
        if (zh) download_vec(z, zh, m);
        if (yh) download_vec(y, yh, m);
    }
    void linsys_init(double rho, double sigma, int linsys, int nbreq) override {
        HIPC(hipSetDevice(device));
        if (linsys != QPS_LINSYS_AUTO && linsys != QPS_LINSYS_CHOLESKY) throw QpsError(QPS_ERR_UNSUPPORTED, "dense handles support QPS_LINSYS_CHOLESKY only");
        nb = pick_nb(nbreq, NP, sweep_fused_supported<T>(NP));
        blocked_off = false;
        factorize(rho, sigma, true);
    }
    void linsys_solve(const double* xh, const double* zh, const double* yh, double rho, double sigma, int changed,
                      double* xxh, double* zzh) override {
        HIPC(hipSetDevice(device));
        if (!factor_valid && !changed) throw QpsError(QPS_ERR_BAD_ARGUMENT, "qps_linsys_solve called before qps_linsys_init");
        if (changed) factorize(rho, sigma, false);                                                  // LinearSystemSolvers.jl:127-129
        upload_vec(xh, x, n); upload_vec(zh, z, m); upload_vec(yh, y, m);
        linear_solve(rho, sigma);
        if (sweep_gave_up()) { factorize(rho, sigma, false); linear_solve(rho, sigma); }
        download_vec(xx, xxh, n); download_vec(zz, zzh, m);
    }
    // qps_operator_apply: the GEMVs the loop and CheckConvergence use (SolveQuadraticProgram.jl:85-89), on work buffers only (x, z, y stay)
    void operator_apply(int op, const double* in, double* out, double rho, double sigma) override {
        HIPC(hipSetDevice(device));
        auto Pin = [&](T* dst) { gemv_rows<T>(st, P, NP, xx, dst, nullptr, T(1), T(0), 0, NP, 0, NP, 0); };
        auto Ain = [&](T* dst) { if (m > 0) gemv_rows<T>(st, A, NP, xx, dst, nullptr, T(1), T(0), 0, MP, 0, NP, 0); };
        auto Atv = [&](const T* v, T* dst) {
            if (m > 0) { gemv_cols_partial<T>(st, A, NP, v, nullptr, T(1), T(0), part, NP, MP, NP); colsum<T>(st, part, NP, part_tiles, nullptr, T(0), nullptr, T(0), dst, NP); }
            else HIPC(hipMemsetAsync(dst, 0, sizeof(T) * NP, st));
        };
        if (op == QPS_OP_AT) { upload_vec(in, zz, m); Atv(zz, Aty); download_vec(Aty, out, n); return; }
        upload_vec(in, xx, n);
        if (op == QPS_OP_P) { Pin(Px); download_vec(Px, out, n); }
        else if (op == QPS_OP_A) { Ain(Ax); download_vec(Ax, out, m); }
        else if (op == QPS_OP_PA) { Pin(Px); Ain(Ax); download_vec(Px, out, n); download_vec(Ax, out + n, m); }
        else if (op == QPS_OP_REDUCED) {                                                            // LinearSystemSolvers.jl:152-157
            Pin(Px); Ain(Ax); Atv(Ax, Aty);
            std::vector<double> a((size_t)n), b((size_t)n);
            download_vec(Px, a.data(), n); download_vec(Aty, b.data(), n);
            for (int64_t i = 0; i < n; ++i) out[i] = a[(size_t)i] + rho * b[(size_t)i] + sigma * in[i];
        } else throw QpsError(QPS_ERR_BAD_ARGUMENT, "unknown qps_operator_kind");
    }
};

// =================================================================================================================
// Batch of independent dense QPs of one shape (BASELINE config 4).  All QPs advance in lock step through batched launches
// (blockIdx.y = QP); rho, the proposed rho, the convergence flag and the active mask are per QP, exactly as if each QP
// ran its own SolveQuadraticProgram! (the tests compare every QP of a batch with its own stand-alone run).
// =================================================================================================================
struct BatchSolverBase {
    int device = 0; int64_t n = 0, m = 0; int count = 0; std::string err; Profiler prof;
    virtual ~BatchSolverBase() {}
    virtual void solve_batch(double* x, const qps_params& p, qps_info* infos) = 0;
    virtual void get_dual(double* z, double* y) = 0;   // [count][m] each
};

template <typename T> struct BatchedDenseSolver : BatchSolverBase {
    hipStream_t st = nullptr;
    int NP = 0, MP = 0, nb = 0, slabs = 0, rpw = 0, part_tiles = 0, sw_slabs = 0;
    T *A = nullptr, *P = nullptr, *q = nullptr, *l = nullptr, *u = nullptr, *PI = nullptr, *AA = nullptr, *M = nullptr, *S = nullptr,
      *tmp = nullptr, *dinv = nullptr;
    T *x = nullptr, *xp = nullptr, *xres = nullptr, *z = nullptr, *y = nullptr, *xx = nullptr, *tt = nullptr, *yv = nullptr, *part = nullptr,
      *part2 = nullptr, *part_tmp = nullptr, *sw_part = nullptr, *Px = nullptr, *Aty = nullptr;
    int* fail = nullptr; int* d_active = nullptr; double *d_rho = nullptr, *d_rhorho = nullptr;
    unsigned long long* scratch = nullptr; double* res_dev = nullptr; double* res_host = nullptr; double* stage = nullptr;
    int* h_int = nullptr; double* h_dbl = nullptr;   // pinned staging for the small per-QP arrays
    bool have_AA = false; double fac_sigma = -1; int fac_nb = -1; std::vector<double> fac_rho;   // per-QP factor cache
    char* sb_args = nullptr; void* sb_host = nullptr;                                              // small-batch path: argument / report slots
    int cat_pass = 0, cat_sweep = 0;

    BatchedDenseSolver(int dev, int cnt, int64_t n_, int64_t m_) {
        device = dev; n = n_; m = m_; count = cnt;
        HIPC(hipSetDevice(device));
        HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        prof.st = st;
        NP = roundup(n, 64); MP = roundup(m, 64);
        slabs = apass_plan<T>(NP, MP, &rpw, count);
        {   // algorithmic bytes per launch of the batched kernels: `count` times the single-QP figures
            const double s = sizeof(T), c = cnt;
            cat_pass = prof.category("apass(fused A-pass, batched)", c * s * ((double)m * n + 3.0 * n + 6.0 * m));
            cat_sweep = prof.category("sweeps(fused fwd+bwd, batched, triangle read once)", c * s * ((double)n * (n + 1) / 2 + 2.0 * n));
        }
        part_tiles = gemv_cols_tiles(MP);
        const int64_t nn = (int64_t)NP * NP, c = count;
        A = dalloc<T>(c * MP * NP, st); P = dalloc<T>(c * nn, st); PI = dalloc<T>(c * nn, st); AA = dalloc<T>(c * nn, st); M = dalloc<T>(c * nn, st);
        S = dalloc<T>(c * nn, st); tmp = dalloc<T>(c * nn, st); dinv = dalloc<T>(c * (int64_t)(NP / 64) * 4096, st);
        q = dalloc<T>(c * NP, st); l = dalloc<T>(c * MP, st); u = dalloc<T>(c * MP, st);
        x = dalloc<T>(c * NP, st); xp = dalloc<T>(c * NP, st); xres = dalloc<T>(c * NP, st); xx = dalloc<T>(c * NP, st); tt = dalloc<T>(c * NP, st);
        yv = dalloc<T>(c * NP, st); Px = dalloc<T>(c * NP, st); Aty = dalloc<T>(c * NP, st); z = dalloc<T>(c * MP, st); y = dalloc<T>(c * MP, st);
        // polishing (one QP at a time) writes the slabs of a count = 1 pass plan, or the tiles of the column GEMV, into `part`:
        // count * slabs can be smaller than either (count = 3, NP = 2048: 246 < 256)
        int rpw1 = 0;
        const int64_t part_slabs = std::max<int64_t>(std::max<int64_t>(c * slabs, apass_plan<T>(NP, MP, &rpw1, 1)), std::max(part_tiles, 1));
        part = dalloc<T>(part_slabs * NP, st); part2 = dalloc<T>(c * slabs * NP, st); part_tmp = dalloc<T>((int64_t)std::max(part_tiles, 1) * NP, st);
        sw_slabs = sweep_fused_slabs<T>(NP, count); sw_part = dalloc<T>(c * std::max(sw_slabs, 1) * NP, st);
        fail = dalloc<int>(count + 4, st); d_active = dalloc<int>(count + 4, st); d_rho = dalloc<double>(count + 4, st); d_rhorho = dalloc<double>(count + 4, st);
        scratch = dalloc<unsigned long long>(16 * c, st); res_dev = dalloc<double>(8 * c, st);
        HIPC(hipHostMalloc((void**)&res_host, 8 * c * sizeof(double)));
        HIPC(hipHostMalloc((void**)&h_int, (count + 4) * sizeof(int)));
        HIPC(hipHostMalloc((void**)&h_dbl, 2 * (count + 4) * sizeof(double)));
        stage = dalloc<double>(std::max<int64_t>((int64_t)MP * NP, nn) + 64, st);
    }
    ~BatchedDenseSolver() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        prof.release_events();
        void* ptrs[] = {A, P, q, l, u, PI, AA, M, S, tmp, dinv, x, xp, xres, z, y, xx, tt, yv, part, part2, part_tmp, sw_part, Px, Aty, fail, d_active,
                        d_rho, d_rhorho, scratch, res_dev, stage};
        for (void* p_ : ptrs) if (p_) (void)hipFree(p_);
        if (res_host) (void)hipHostFree(res_host);
        if (h_int) (void)hipHostFree(h_int);
        if (h_dbl) (void)hipHostFree(h_dbl);
        if (sb_args) (void)hipFree(sb_args);
        if (sb_host) (void)hipHostFree(sb_host);
        if (st) (void)hipStreamDestroy(st);
    }
    // Loading a batch: everything goes through ONE pinned ring (several host threads fill a half while the other half travels) and ONE device staging buffer,
    // all on the handle's stream -- the copy into `stage` for matrix k + 1 is ordered behind the import kernel that read matrix k, so nothing waits on the host until
    // finish_loading().  (A synchronisation per matrix left the DMA idle while the host copied and the host idle while the DMA ran: 64 QPs of n = 1024 took 150 ms to load.)
    std::unique_ptr<FastUploader> loader;
    void put_vec(const double* h, T* d, int64_t cnt_) {
        if (cnt_ <= 0) return;
        if (!loader) loader.reset(new FastUploader(st, device));
        loader->copy(stage, h, sizeof(double) * (size_t)cnt_);
        convert_copy<T>(st, stage, d, cnt_);
    }
    void put_matrix(const double* h, int rows, int cols, T* d) {   // column-major rows x cols (ld = rows) -> row-major, ld NP
        if (rows <= 0 || cols <= 0) return;
        if (!loader) loader.reset(new FastUploader(st, device));
        loader->copy(stage, h, sizeof(double) * (size_t)rows * cols);
        import_colmajor<T>(st, stage, rows, rows, cols, d, NP);
    }
    void finish_loading() { HIPC(hipStreamSynchronize(st)); loader.reset(); }
    void get_dual(double* zh, double* yh) override {
        HIPC(hipSetDevice(device));
        for (int b = 0; b < count && m > 0; ++b) {
            if (zh) { convert_back<T>(st, z + (int64_t)b * MP, stage, m); HIPC(hipMemcpyAsync(zh + (int64_t)b * m, stage, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st)); }
            if (yh) { convert_back<T>(st, y + (int64_t)b * MP, stage, m); HIPC(hipMemcpyAsync(yh + (int64_t)b * m, stage, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st)); }
        }
    }
    void load_problem(int b, const double* Ph, const double* Ah, const double* qh, const double* lh, const double* uh) {
        put_matrix(Ph, (int)n, (int)n, P + (int64_t)b * NP * NP);
        put_matrix(Ah, (int)m, (int)n, A + (int64_t)b * MP * NP);
        put_vec(qh, q + (int64_t)b * NP, n); put_vec(lh, l + (int64_t)b * MP, m); put_vec(uh, u + (int64_t)b * MP, m);
    }
    // LinearSystemSolvers.jl:112-114 / :127-129 for QP b (no host synchronisation; fail[b] is read later)
    void factorize_one(int b, double rho, double sigma, bool rebuild) {
        const int64_t nn = (int64_t)NP * NP;
        T *Pb = P + b * nn, *Ab = A + (int64_t)b * MP * NP, *PIb = PI + b * nn, *AAb = AA + b * nn, *Mb = M + b * nn, *Sb = S + b * nn;
        if (rebuild) {
            make_PI<T>(st, (int)n, NP, Pb, (T)sigma, PIb);
            gemm<T>(st, NP, NP, MP, T(1), Ab, NP, false, Ab, NP, false, T(0), AAb, NP, true);
        }
        assemble_M<T>(st, NP, PIb, AAb, (T)rho, Mb);
        T* dinvb = dinv + (int64_t)b * (NP / 64) * 4096;
        cholesky<T>(st, NP, Mb, dinvb, fail + b, 1, chol_scratch_fits(NP) ? Sb : nullptr);
        build_sweep_matrix<T>(st, NP, nb, Mb, dinvb, Sb, tmp + b * nn);
    }
    // all QPs at once (same rho): every launch of the panel chain carries the whole batch
    void factorize_all(double rho, double sigma, bool rebuild, const double* rho_arr = nullptr) {
        const int64_t nn = (int64_t)NP * NP;
        if (rebuild) {
            make_PI<T>(st, (int)n, NP, P, (T)sigma, PI, count);
            gemm<T>(st, NP, NP, MP, T(1), A, NP, false, A, NP, false, T(0), AA, NP, true, count, (int64_t)MP * NP, (int64_t)MP * NP, nn);
        }
        assemble_M<T>(st, NP, PI, AA, (T)rho, M, count, rho_arr);
        cholesky<T>(st, NP, M, dinv, fail, count, chol_scratch_fits(NP) ? S : nullptr);
        build_sweep_matrix<T>(st, NP, nb, M, dinv, S, tmp, count);
    }
    void check_fail(const std::vector<int>& which) {
        if (which.empty()) return;
        HIPC(hipMemcpyAsync(h_int, fail, sizeof(int) * count, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        for (int b : which) if (h_int[b] != 0) {
            char buf[256]; snprintf(buf, sizeof buf, "QP %d of the batch: Cholesky of P + sigma I + rho A'A broke down at column %d", b, h_int[b]);
            throw QpsError(QPS_ERR_FACTORIZATION, buf);
        }
    }
    void push_state(const std::vector<double>& rho, const std::vector<double>& rhorho, const std::vector<int>& active) {
        for (int b = 0; b < count; ++b) { h_dbl[b] = rho[b]; h_dbl[count + 4 + b] = rhorho[b]; h_int[b] = active[b]; }
        HIPC(hipMemcpyAsync(d_rho, h_dbl, sizeof(double) * count, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(d_rhorho, h_dbl + count + 4, sizeof(double) * count, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(d_active, h_int, sizeof(int) * count, hipMemcpyHostToDevice, st));
        HIPC(hipStreamSynchronize(st));   // the pinned staging is reused
    }

    void solve_batch(double* xh, const qps_params& p, qps_info* infos) override {
        HIPC(hipSetDevice(device));
        const double t0 = now_s();
        if (slabs <= 0) throw QpsError(QPS_ERR_UNSUPPORTED, "batched path needs m >= 1 and n within the fused-pass limit");
        nb = pick_nb(p.trsvBlock, NP, sweep_fused_supported<T>(NP));
        const double sigma = p.sigma, alpha = p.alpha;
        const double epsAdmm = std::fmin(p.epsAbs, p.epsRel) * 1e-2;
        std::vector<double> rho(count, p.rho), rhorho(count, p.rho), resP(count, NAN), resD(count, NAN), tref(count, 0.0);
        std::vector<int> active(count, 1), conv(count, QPS_CONV_NUM_ITR), iters(count, p.numIterations), nref(count, 0), all(count);
        for (int b = 0; b < count; ++b) all[b] = b;
        HIPC(hipMemsetAsync(fail, 0, sizeof(int) * count, st));
        const bool rebuild = !(p.reuseFactor && have_AA && fac_sigma == sigma);
        if ((int)fac_rho.size() != count) fac_rho.assign(count, -1.0);
        std::vector<int> todo;
        for (int b = 0; b < count; ++b)                                                             // SolveQuadraticProgram.jl:36
            if (rebuild || fac_rho[b] != p.rho || fac_nb != nb) todo.push_back(b);
        if ((int)todo.size() == count) factorize_all(p.rho, sigma, rebuild);
        else for (int b : todo) factorize_one(b, p.rho, sigma, rebuild);
        check_fail(todo);
        have_AA = true; fac_sigma = sigma; fac_nb = nb;
        for (int b : todo) fac_rho[b] = p.rho;
        for (int b = 0; b < count; ++b) put_vec(xh + (int64_t)b * n, x + (int64_t)b * NP, n);
        loader.reset();                                                                             // (gives the pinned ring back)
        HIPC(hipMemsetAsync(z, 0, sizeof(T) * (size_t)count * MP, st));                             // :39
        HIPC(hipMemsetAsync(y, 0, sizeof(T) * (size_t)count * MP, st));                             // :40
        push_state(rho, rhorho, active);
        const double t1 = now_s();
        BatchStride bsS; bsS.count = count; bsS.mat = (int64_t)NP * NP; bsS.vin = NP; bsS.vout = NP; bsS.active = d_active;
        BatchStride bsC; bsC.count = count; bsC.mat = (int64_t)slabs * NP; bsC.vin = NP; bsC.vout = NP; bsC.active = d_active;
        BatchStride bsK; bsK.count = count; bsK.vin = NP; bsK.vout = MP; bsK.active = d_active;
        PassBatch pb; pb.count = count; pb.slabs = slabs; pb.rho_arr = d_rho; pb.active = d_active;
        int rhs_slabs = 0, nactive = count, ii = 0;
        const int nblk = (NP + nb - 1) / nb;
        if (p.loopVariant == 0 && nblk == 1 && admm_small_batch_supported<T>(NP, MP)) {
            // Small shapes: ONE workgroup per QP runs that QP's whole loop (register-resident kernel, k_small.hip); the host only
            // steps in when some QP wants a rho switch (refactor, relaunch from its own iteration) -- SolveQuadraticProgram.jl:45-71 per QP.
            const size_t ab = admm_small_args_bytes(), ob = admm_small_out_bytes();
            if (!sb_args) { sb_args = dalloc<char>((int64_t)(ab + ob) * count + 64, st); HIPC(hipHostMalloc(&sb_host, (ab + ob) * (size_t)count + 64)); }
            char* args_h = static_cast<char*>(sb_host); char* outs_h = args_h + ab * count;
            char* args_d = sb_args; char* outs_d = sb_args + ab * count;
            std::vector<int> it(count, 0);
            HIPC(hipMemsetAsync(xp, 0, sizeof(T) * (size_t)count * NP, st));                        // :38
            while (nactive > 0) {
                for (int b = 0; b < count; ++b)
                    admm_small_args_set(args_h, b, (int)n, (int)m, NP, MP, it[b], active[b] ? p.numIterations : it[b], p.numItrConv, p.adptRho, rho[b], rhorho[b],
                                        sigma, alpha, p.epsAbs, p.epsRel, epsAdmm, p.fctrRho);
                HIPC(hipMemcpyAsync(args_d, args_h, ab * count, hipMemcpyHostToDevice, st));
                admm_small_batch<T>(st, count, NP, MP, args_d, A, P, S, q, l, u, x, xp, z, y, outs_d);
                HIPC(hipMemcpyAsync(outs_h, outs_d, ob * count, hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                std::vector<int> changed;
                for (int b = 0; b < count; ++b) {
                    if (!active[b]) continue;
                    int last = 0, flag = QPS_CONV_NUM_ITR, need = 0; double r8[8];
                    admm_small_read(outs_h + ob * b, &last, &flag, &need, r8);
                    it[b] = last; rhorho[b] = r8[4];
                    if (!std::isnan(r8[0]) || !std::isnan(r8[1])) { resP[b] = r8[0]; resD[b] = r8[1]; }
                    if (flag != QPS_CONV_NUM_ITR) { conv[b] = flag; iters[b] = last; active[b] = 0; --nactive; }   // :66-68
                    else if (need && last < p.numIterations) { rho[b] = rhorho[b]; ++nref[b]; changed.push_back(b); }   // :47-51
                    else { iters[b] = p.numIterations; active[b] = 0; --nactive; }
                }
                if (!changed.empty()) {
                    const double ta = now_s();
                    const bool together = (int)changed.size() * 2 >= count;
                    if (together) { push_state(rho, rhorho, active); factorize_all(0.0, sigma, false, d_rho); for (int b = 0; b < count; ++b) fac_rho[b] = rho[b]; }
                    else for (int b : changed) { factorize_one(b, rho[b], sigma, false); fac_rho[b] = rho[b]; }
                    if (together) check_fail(all); else check_fail(changed);
                    const double dt = (now_s() - ta) / changed.size();
                    for (int b : changed) tref[b] += dt;
                }
            }
            HIPC(hipMemcpyAsync(xres, x, sizeof(T) * (size_t)count * NP, hipMemcpyDeviceToDevice, st));
        }
        for (ii = 1; ii <= p.numIterations && nactive > 0; ++ii) {                                  // :45
            std::vector<int> changed;
            if (p.adptRho)
                for (int b = 0; b < count; ++b)
                    if (active[b] && ((rhorho[b] * p.fctrRho < rho[b]) || (rhorho[b] > p.fctrRho * rho[b]))) {   // :47
                        rho[b] = rhorho[b]; ++nref[b]; changed.push_back(b);
                    }
            if (!changed.empty()) {
                const double ta = now_s();
                const bool together = (int)changed.size() * 2 >= count;   // most QPs switch at the same check: one batched refactor
                if (together) {
                    push_state(rho, rhorho, active);                      // d_rho must hold the new values before assembly
                    factorize_all(0.0, sigma, false, d_rho);
                    for (int b = 0; b < count; ++b) fac_rho[b] = rho[b];
                }
                for (int b : changed) {
                    if (!together) { factorize_one(b, rho[b], sigma, false); fac_rho[b] = rho[b]; }
                    if (rhs_slabs > 0) {   // slabs of A'(rho z - y) depend on rho: rebuild them for this QP (slab 0 = the sum, rest 0)
                        T* pb_ = part + (int64_t)b * slabs * NP;
                        gemv_cols_partial<T>(st, A + (int64_t)b * MP * NP, NP, z + (int64_t)b * MP, y + (int64_t)b * MP, (T)rho[b], T(-1), part_tmp, NP, MP, NP);
                        colsum<T>(st, part_tmp, NP, part_tiles, nullptr, T(0), nullptr, T(0), pb_, NP);
                        if (slabs > 1) HIPC(hipMemsetAsync(pb_ + NP, 0, sizeof(T) * (size_t)(slabs - 1) * NP, st));
                    }
                }
                if (together) check_fail(all); else check_fail(changed);
                push_state(rho, rhorho, active);
                const double dt = (now_s() - ta) / changed.size();
                for (int b : changed) tref[b] += dt;
            }
            const bool check = (ii % p.numItrConv == 0);
            colsum<T>(st, part, NP, rhs_slabs, x, (T)sigma, q, T(-1), tt, NP, bsC);                  // LinearSystemSolvers.jl:136
            if (nblk == 1 && sweep_fused_supported<T>(NP)) {                                       // both sweeps in one pass
                BatchStride bsW = bsS; bsW.vout = (int64_t)sw_slabs * NP;
                { ProfLaunchScope ps(prof, cat_sweep, (prof.level == 1 && ii % 50 == 13) ? 1 : 2); sweep_fused<T>(st, S, NP, NP, tt, sw_part, NP, bsW); }
                BatchStride bsX = bsS; bsX.mat = (int64_t)sw_slabs * NP;
                colsum<T>(st, sw_part, NP, sw_slabs, nullptr, T(0), nullptr, T(0), xx, NP, bsX);
            } else
            for (int J = 0; J < nblk; ++J) {                                                        // forward sweep
                const int r0 = J * nb, r1 = std::min(NP, r0 + nb);
                gemv_rows<T>(st, S, NP, tt, yv, nullptr, T(1), T(0), r0, r1, r0, r1, 1, bsS);
                if (r1 < NP) gemv_rows<T>(st, S, NP, yv, tt, tt, T(-1), T(1), r1, NP, r0, r1, 0, bsS);
            }
            if (!(nblk == 1 && sweep_fused_supported<T>(NP)))
            for (int J = nblk - 1; J >= 0; --J) {                                                   // backward sweep
                const int r0 = J * nb, r1 = std::min(NP, r0 + nb);
                gemv_rows<T>(st, S, NP, yv, xx, nullptr, T(1), T(0), r0, r1, r0, r1, 2, bsS);
                if (r0 > 0) gemv_rows<T>(st, S, NP, xx, yv, yv, T(-1), T(1), 0, r0, r0, r1, 0, bsS);
            }
            if (check) HIPC(hipMemsetAsync(scratch, 0, 16 * sizeof(unsigned long long) * count, st));
            {
                ProfLaunchScope ps(prof, cat_pass, (prof.level == 1 && !check && ii % 50 == 38) ? 1 : (check ? 3 : 2));   // level 1: one plain launch in 50
                apass<T>(st, check, A, NP, NP, MP, xx, x, xp, z, y, l, u, (T)alpha, T(1), part, part2, NP, scratch, pb);   // :56-61 + next rhs
            }
            std::swap(x, xp);
            rhs_slabs = slabs;
            if (check) {                                                                            // :63-69
                colsum<T>(st, part2, NP, slabs, nullptr, T(0), nullptr, T(0), Aty, NP, bsC);
                BatchStride bsP = bsS; bsP.mat = (int64_t)NP * NP;
                gemv_rows<T>(st, P, NP, x, Px, nullptr, T(1), T(0), 0, NP, 0, NP, 0, bsP);
                CheckScalars cs{p.epsAbs, p.epsRel, epsAdmm, 0.0, 0.0, p.adptRho, QPS_CONV_NUM_ITR};
                check_convergence<T>(st, (int)n, (int)m, (const T*)nullptr, Px, Aty, q, x, xp, z, z, scratch, res_dev, cs, 1, bsK, d_rho, d_rhorho);
                HIPC(hipMemcpyAsync(res_host, res_dev, 8 * sizeof(double) * count, hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                prof.harvest();
                bool any_done = false;
                for (int b = 0; b < count; ++b) {
                    if (!active[b]) continue;
                    const double* r = res_host + 8 * b;
                    resP[b] = r[0]; resD[b] = r[1]; rhorho[b] = r[4]; conv[b] = (int)r[5];
                    if (conv[b] != QPS_CONV_NUM_ITR) {
                        active[b] = 0; iters[b] = ii; --nactive; any_done = true;
                        HIPC(hipMemcpyAsync(xres + (int64_t)b * NP, x + (int64_t)b * NP, sizeof(T) * NP, hipMemcpyDeviceToDevice, st));
                    }
                }
                push_state(rho, rhorho, active);
                (void)any_done;
            }
        }
        for (int b = 0; b < count; ++b)
            if (active[b]) HIPC(hipMemcpyAsync(xres + (int64_t)b * NP, x + (int64_t)b * NP, sizeof(T) * NP, hipMemcpyDeviceToDevice, st));
        HIPC(hipStreamSynchronize(st));
        prof.harvest();
        const double t2 = now_s();
        std::vector<PolishReport> pol(count);
        if (p.polish)                                                                               // SolveQuadraticProgram.m:289-325, one QP after the other
            for (int b = 0; b < count; ++b)
                polish_dense<T>(st, n, m, NP, MP, P + (int64_t)b * NP * NP, A + (int64_t)b * MP * NP, q + (int64_t)b * NP, l + (int64_t)b * MP,
                                u + (int64_t)b * MP, y + (int64_t)b * MP, xres + (int64_t)b * NP, part, p, &pol[b]);
        for (int b = 0; b < count; ++b) {
            convert_back<T>(st, xres + (int64_t)b * NP, stage, n);
            HIPC(hipMemcpyAsync(xh + (int64_t)b * n, stage, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (infos) {
                qps_info& in = infos[b];
                in.convFlag = conv[b]; in.iterations = iters[b]; in.numRefactor = nref[b]; in.cgIterations = 0;
                in.rhoFinal = rho[b]; in.rhoProposed = rhorho[b]; in.resPrim = resP[b]; in.resDual = resD[b];
                in.tSetup = t1 - t0; in.tLoop = t2 - t1; in.tRefactor = tref[b];   // wall time of the whole batch
                in.polishFlag = pol[b].flag; in.polishIterations = pol[b].minresIterations; in.tPolish = pol[b].seconds;
                in.trsvBlock = nb; in.sweepVariant = (nblk == 1 && sweep_fused_supported<T>(NP)) ? 2 : (nblk == 1 ? 3 : 1);
                in.sweepGaveUp = 0; in.cgExplicit = 0;
            }
        }
    }
};

// =================================================================================================================
// handle plumbing
// =================================================================================================================
struct Handle { SolverBase* impl = nullptr; std::vector<SolverBase*> batch; BatchSolverBase* fused_batch = nullptr; ProxQpBase* proxqp = nullptr; int64_t n = 0, m = 0; std::string err; };

int fail_with(Handle* h, int code, const std::string& msg) {
    g_last_error = msg;
    if (h) h->err = msg;
    return code;
}
bool all_finite(const double* p, int64_t count, bool allow_inf) {
    // branch-free over blocks (vectorises), an early exit between blocks; large arrays on several host threads
    std::atomic<int> bad{0};
    host_parallel(count, (int64_t)1 << 20, [&](int, int64_t b, int64_t e) {
        for (int64_t i0 = b; i0 < e && !bad.load(std::memory_order_relaxed); i0 += 4096) {
            const int64_t i1 = std::min(e, i0 + 4096);
            int flag = 0;
            if (allow_inf) { for (int64_t i = i0; i < i1; ++i) flag |= (p[i] != p[i]); }
            else { for (int64_t i = i0; i < i1; ++i) flag |= !(std::fabs(p[i]) <= 1.7976931348623157e308); }
            if (flag) bad.store(1, std::memory_order_relaxed);
        }
    });
    return bad.load() == 0;
}
// a column-major matrix (every column `rows` long, `ld` apart): contiguous storage is checked as one array
bool all_finite_matrix(const double* p, int64_t rows, int64_t cols, int64_t ld) {
    if (ld == rows) return all_finite(p, rows * cols, false);
    std::atomic<int> bad{0};
    host_parallel(cols, std::max<int64_t>(1, ((int64_t)1 << 20) / std::max<int64_t>(rows, 1)), [&](int, int64_t b, int64_t e) {
        for (int64_t j = b; j < e && !bad.load(std::memory_order_relaxed); ++j) {
            int flag = 0;
            for (int64_t i = 0; i < rows; ++i) flag |= !(std::fabs(p[i + j * ld]) <= 1.7976931348623157e308);
            if (flag) bad.store(1, std::memory_order_relaxed);
        }
    });
    return bad.load() == 0;
}
// issymmetric(mP) with tolerance 0, as SolveQuadraticProgram.m:166-168 (the MATLAB implementation raises; the CSR path reads the
// caller's CSC of P as its CSR and the dense check reads rows of P, so an asymmetric P would silently solve a different problem).
// Tiled so that both the (i, j) and the (j, i) walk stay inside a cached 64 x 64 block.  Returns -1 or the first offending column.
int64_t dense_asymmetry(const double* P, int64_t n, int64_t ldp) {
    const int64_t nb = (n + 63) / 64;
    std::atomic<int64_t> first{INT64_MAX};
    // block columns dealt cyclically over the host threads (block column b holds nb - b tiles); the smallest offending column wins, as in a serial sweep
    const int nt = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n * n / ((int64_t)1 << 20)));
    auto work = [&](int t) {
        for (int64_t b = t; b < nb; b += nt) {
            const int64_t j0 = b * 64;
            if (j0 >= first.load(std::memory_order_relaxed)) break;
            for (int64_t i0 = j0; i0 < n; i0 += 64) {
                const int64_t j1 = std::min(n, j0 + 64), i1 = std::min(n, i0 + 64);
                for (int64_t j = j0; j < j1; ++j)
                    for (int64_t i = std::max(i0, j + 1); i < i1; ++i)
                        if (P[i + j * ldp] != P[j + i * ldp]) { int64_t cur = first.load(); while (j < cur && !first.compare_exchange_weak(cur, j)) {} }
            }
        }
    };
    std::vector<std::thread> th;
    int started = 1;
    try { for (int t = 1; t < nt; ++t) { th.emplace_back(work, t); started = t + 1; } } catch (const std::system_error&) {}
    work(0);
    for (int t = started; t < nt; ++t) work(t);        // (thread creation refused: the caller takes those block columns)
    for (auto& x : th) x.join();
    const int64_t f = first.load();
    return f == INT64_MAX ? -1 : f;
}
int check_device(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return QPS_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt) return QPS_ERR_BAD_ARGUMENT;
    return QPS_OK;
}

template <typename F> int guarded(Handle* h, F&& f) {
    try { f(); return QPS_OK; }
    catch (const QpsError& e) { return fail_with(h, e.code, e.msg); }
    catch (const std::bad_alloc&) { return fail_with(h, QPS_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    catch (const std::exception& e) { return fail_with(h, QPS_ERR_HIP, e.what()); }
}

int validate_params(Handle* h, const qps_params* p) {
    if (!p) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "params is NULL");
    if (p->numIterations < 0) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "numIterations must be >= 0");
    if (p->numItrConv <= 0) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "numItrConv must be positive");
    if (!(p->rho > 0) || !std::isfinite(p->rho)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "rho must be positive and finite");
    if (!(p->sigma >= 0) || !std::isfinite(p->sigma)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "sigma must be non-negative and finite");
    if (!std::isfinite(p->alpha)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "alpha must be finite");
    if (std::isnan(p->epsAbs) || std::isnan(p->epsRel)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "epsAbs/epsRel must not be NaN");
    if (p->adptRho && !(p->fctrRho > 0)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "fctrRho must be positive");
    return QPS_OK;
}

SolverBase* make_dense(int device, int64_t n, int64_t m, int dtype, const double* P, int64_t ldp, const double* A, int64_t lda,
                       const double* q, const double* l, const double* u) {
    if (dtype == QPS_F64) {
        auto* s = new DenseSolver<double>(device, n, m, dtype);
        try {
            s->upload_matrix(P, ldp, (int)n, (int)n, s->P, s->NP);
            s->upload_matrix(A, lda, (int)m, (int)n, s->A, s->NP);
            s->upload_vec(q, s->q, n); s->upload_vec(l, s->l, m); s->upload_vec(u, s->u, m);
        } catch (...) { delete s; throw; }
        return s;
    }
    auto* s = new DenseSolver<float>(device, n, m, dtype);
    try {
        s->upload_matrix(P, ldp, (int)n, (int)n, s->P, s->NP);
        s->upload_matrix(A, lda, (int)m, (int)n, s->A, s->NP);
        s->upload_vec(q, s->q, n); s->upload_vec(l, s->l, m); s->upload_vec(u, s->u, m);
    } catch (...) { delete s; throw; }
    return s;
}

}  // namespace

extern "C" {

#define QPS_API __attribute__((visibility("default")))

QPS_API int32_t qps_default_params(qps_params* p) {
    if (!p) return QPS_ERR_BAD_ARGUMENT;
    memset(p, 0, sizeof(*p));
    p->numIterations = 5000; p->epsAbs = 1e-6; p->epsRel = 1e-6;            // SolveQuadraticProgram.jl:15
    p->rho = 1.0; p->sigma = 1e-6; p->alpha = 1.6; p->delta = 1e-6; p->adptRho = 0;   // :16
    p->fctrRho = 5.0; p->numItrConv = 25; p->numItrPolish = 10; p->epsMinres = 1e-6; p->numItrMinres = 500;   // :17
    p->linsys = QPS_LINSYS_AUTO; p->trsvBlock = 0; p->reuseFactor = 0; p->loopVariant = 0;
    p->epsPcg = 1e-6; p->numItrPcg = 1000;                                  // LinearSystemSolvers.jl:125
    return QPS_OK;
}

QPS_API int32_t qps_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

QPS_API int32_t qps_create_dense(int64_t n, int64_t m, const double* P, int64_t ldp, const double* A, int64_t lda,
                                 const double* q, const double* l, const double* u, int32_t dtype, int32_t device,
                                 qps_handle* out) {
    if (!out) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "out handle pointer is NULL");
    *out = nullptr;
    if (n <= 0 || m < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1 and m >= 0");
    if (n > (1 << 20) || m > (1 << 24)) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "dense problem too large");
    if (!P || !q || (m > 0 && (!A || !l || !u))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (ldp < n || (m > 0 && lda < m)) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "leading dimension smaller than the row count");
    if (dtype != QPS_F64 && dtype != QPS_F32) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "unknown dtype");
    if (!all_finite_matrix(P, n, n, ldp)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "P contains NaN/Inf");
    if (m > 0 && !all_finite_matrix(A, m, n, lda)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "A contains NaN/Inf");
    if (!all_finite(q, n, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "q contains NaN/Inf");
    if (m > 0 && (!all_finite(l, m, true) || !all_finite(u, m, true))) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "l/u contain NaN");
    {
        const int64_t bad = dense_asymmetry(P, n, ldp);                              // SolveQuadraticProgram.m:166-168
        if (bad >= 0) { char b[160]; snprintf(b, sizeof b, "The matrix mP must be a symmetric positive definite matrix (asymmetric entry in column %lld)", (long long)bad); return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, b); }
    }
    // argument validation first (a CPU-only caller sees the same errors), the device last
    int dc = check_device(device);
    if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
    if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range");
    Handle* h = new Handle(); h->n = n; h->m = m;
    int rc = guarded(nullptr, [&] { h->impl = make_dense(device, n, m, dtype, P, ldp, A, lda, q, l, u); });
    if (rc != QPS_OK) { delete h; return rc; }
    *out = reinterpret_cast<qps_handle>(h);
    return QPS_OK;
}

QPS_API int32_t qps_create_csc(int64_t n, int64_t m, const int64_t* Pcp, const int64_t* Pri, const double* Pnz,
                               const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* q, const double* l,
                               const double* u, int32_t index_base, int32_t dense_path, int32_t dtype, int32_t device,
                               qps_handle* out) {
    if (!out) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "out handle pointer is NULL");
    *out = nullptr;
    if (n <= 0 || m < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1 and m >= 0");
    if (!Pcp || !q || !Acp || (m > 0 && (!l || !u))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (index_base != 0 && index_base != 1) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "index_base must be 0 or 1");
    if (dtype != QPS_F64 && dtype != QPS_F32) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "unknown dtype");
    {   // colptr / rowval / nzval of both matrices (plain host code shared with the CPU tests: spmv_layout.cpp)
        std::string why;
        int vc = layout::validate_csc(n, n, Pcp, Pri, Pnz, index_base, "P", &why);
        if (vc == 0) vc = layout::validate_csc(m, n, Acp, Ari, Anz, index_base, "A", &why);
        if (vc != 0) return fail_with(nullptr, vc, why);
    }
    if (!all_finite(q, n, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "q contains NaN/Inf");
    if (m > 0 && (!all_finite(l, m, true) || !all_finite(u, m, true))) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "l/u contain NaN");
    {
        int64_t bad = -1;
        int src = guarded(nullptr, [&] { bad = layout::csc_asymmetry(n, Pcp, Pri, Pnz, index_base); });   // SolveQuadraticProgram.m:166-168
        if (src != QPS_OK) return src;
        if (bad >= 0) { char b[160]; snprintf(b, sizeof b, "The matrix mP must be a symmetric positive definite matrix (asymmetric entry in column %lld)", (long long)bad); return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, b); }
    }
    // argument validation first (a CPU-only caller sees the same errors), the device last
    int dc = check_device(device);
    if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
    if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range");
    Handle* h = new Handle(); h->n = n; h->m = m;
    int rc;
    if (dense_path) {
        if ((double)n * n > 4e9 || (double)m * n > 8e9) { delete h; return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "problem too large to densify"); }
        rc = guarded(nullptr, [&] {
            std::vector<double> Pd((size_t)n * n, 0.0), Ad((size_t)std::max<int64_t>(m, 1) * n, 0.0);
            for (int64_t j = 0; j < n; ++j) {
                for (int64_t k = Pcp[j] - index_base; k < Pcp[j + 1] - index_base; ++k) Pd[(size_t)(Pri[k] - index_base) + (size_t)j * n] += Pnz[k];
                for (int64_t k = Acp[j] - index_base; k < Acp[j + 1] - index_base; ++k) Ad[(size_t)(Ari[k] - index_base) + (size_t)j * m] += Anz[k];
            }
            h->impl = make_dense(device, n, m, dtype, Pd.data(), n, Ad.data(), std::max<int64_t>(m, 1), q, l, u);
        });
    } else {
        rc = guarded(nullptr, [&] { h->impl = make_sparse_solver(device, n, m, dtype, Pcp, Pri, Pnz, Acp, Ari, Anz, q, l, u, index_base); });
    }
    if (rc != QPS_OK) { delete h; return rc; }
    *out = reinterpret_cast<qps_handle>(h);
    return QPS_OK;
}

QPS_API int32_t qps_solve(qps_handle hh, double* x, const qps_params* p, qps_info* info) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (!x) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "x_inout is NULL");
    int rc = validate_params(h, p);
    if (rc != QPS_OK) return rc;
    if (!all_finite(x, h->n, false)) return fail_with(h, QPS_ERR_NOT_FINITE, "x_inout (warm start) contains NaN/Inf");
    return guarded(h, [&] { h->impl->solve(x, *p, info); });
}

QPS_API int32_t qps_polish(qps_handle hh, double* x, const double* y, const qps_params* p, qps_polish_report* rep) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (!x || (h->m > 0 && !y)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "NULL vector");
    int rc = validate_params(h, p);
    if (rc != QPS_OK) return rc;
    if (!all_finite(x, h->n, false) || !all_finite(y, h->m, false)) return fail_with(h, QPS_ERR_NOT_FINITE, "x or y contains NaN/Inf");
    return guarded(h, [&] { h->impl->polish(x, y, *p, rep); });
}

QPS_API int32_t qps_get_dual(qps_handle hh, double* z, double* y) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || (!h->impl && !h->fused_batch && h->batch.empty())) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (h->fused_batch) return guarded(h, [&] { h->fused_batch->get_dual(z, y); });
    if (!h->impl)       // a batch of independent solvers: [count][m] like the fused batch
        return guarded(h, [&] { for (size_t b = 0; b < h->batch.size(); ++b) h->batch[b]->get_dual(z ? z + (int64_t)b * h->m : nullptr, y ? y + (int64_t)b * h->m : nullptr); });
    return guarded(h, [&] { h->impl->get_dual(z, y); });
}

QPS_API int32_t qps_linsys_init(qps_handle hh, double rho, double sigma, int32_t linsys, int32_t trsvBlock) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (!(rho > 0) || !(sigma >= 0)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "rho must be positive, sigma non-negative");
    return guarded(h, [&] { h->impl->linsys_init(rho, sigma, linsys, trsvBlock); });
}

QPS_API int32_t qps_linsys_solve(qps_handle hh, const double* x, const double* z, const double* y, double rho, double sigma,
                                 int32_t changed, double* xx, double* zz) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (!x || !xx || (h->m > 0 && (!z || !y || !zz))) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "NULL vector");
    if (!(rho > 0) || !(sigma >= 0)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "rho must be positive, sigma non-negative");
    return guarded(h, [&] { h->impl->linsys_solve(x, z, y, rho, sigma, changed, xx, zz); });
}

QPS_API int32_t qps_operator_apply(qps_handle hh, int32_t op, const double* in, double* out, double rho, double sigma) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle (qps_operator_apply takes a single-problem handle)");
    if (op < QPS_OP_P || op > QPS_OP_REDUCED) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "unknown qps_operator_kind");
    if (!in || !out) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "NULL vector");
    if (op == QPS_OP_REDUCED && (!(rho >= 0) || !(sigma >= 0))) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "rho and sigma must be non-negative");
    if (!all_finite(in, op == QPS_OP_AT ? h->m : h->n, false)) return fail_with(h, QPS_ERR_NOT_FINITE, "the input vector contains NaN/Inf");
    return guarded(h, [&] { h->impl->operator_apply(op, in, out, rho, sigma); });
}

QPS_API int32_t qps_linsys_set_cg(qps_handle hh, double epsPcg, int32_t numItrPcg) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->impl) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "invalid handle");
    if (std::isnan(epsPcg) || epsPcg < 0 || numItrPcg < 0) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "need epsPcg >= 0 and numItrPcg >= 0");
    return guarded(h, [&] { h->impl->linsys_set_cg(epsPcg, numItrPcg); });
}

QPS_API int32_t qps_create_dense_batch(int64_t count, int64_t n, int64_t m, const double* P, const double* A, const double* q,
                                       const double* l, const double* u, int32_t dtype, int32_t device, qps_handle* out) {
    if (!out) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "out handle pointer is NULL");
    *out = nullptr;
    if (count <= 0 || count > 65535) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "batch count must be in 1..65535");
    if (n <= 0 || m < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1 and m >= 0");
    if (!P || !q || (m > 0 && (!A || !l || !u))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (dtype != QPS_F64 && dtype != QPS_F32) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "unknown dtype");
    if (!all_finite(P, count * n * n, false) || !all_finite(q, count * n, false) || (m > 0 && !all_finite(A, count * m * n, false)))
        return fail_with(nullptr, QPS_ERR_NOT_FINITE, "P/A/q contain NaN/Inf");
    if (m > 0 && (!all_finite(l, count * m, true) || !all_finite(u, count * m, true))) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "l/u contain NaN");
    {   // issymmetric(mP) for every QP of the batch, the QPs dealt to the host threads (64 QPs of n = 1024 took 130 ms one after the other); the first offender is reported
        std::atomic<int64_t> first_bad{INT64_MAX};
        host_parallel(count, std::max<int64_t>(1, ((int64_t)1 << 20) / (n * n)), [&](int, int64_t b0, int64_t b1) {
            for (int64_t b = b0; b < b1 && b < first_bad.load(std::memory_order_relaxed); ++b)
                if (dense_asymmetry(P + b * n * n, n, n) >= 0) { int64_t cur = first_bad.load(); while (b < cur && !first_bad.compare_exchange_weak(cur, b)) {} }
        });
        const int64_t b = first_bad.load();
        if (b != INT64_MAX) { char bf[160]; snprintf(bf, sizeof bf, "QP %lld of the batch: the matrix mP must be a symmetric positive definite matrix", (long long)b); return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, bf); }
    }
    // argument validation first (a CPU-only caller sees the same errors), the device last
    int dc = check_device(device);
    if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
    if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range");
    Handle* h = new Handle(); h->n = n; h->m = m;
    int rpw = 0;
    const int NPb = roundup(n, 64), MPb = roundup(m, 64);
    const bool fusable = count > 1 && m > 0 && (dtype == QPS_F64 ? apass_plan<double>(NPb, MPb, &rpw, (int)count) : apass_plan<float>(NPb, MPb, &rpw, (int)count)) > 0;
    if (fusable) {
        int rc = guarded(nullptr, [&] {
            auto load = [&](auto* s) {
                h->fused_batch = s;
                for (int64_t b = 0; b < count; ++b) s->load_problem((int)b, P + b * n * n, A + b * m * n, q + b * n, l + b * m, u + b * m);
                s->finish_loading();
            };
            if (dtype == QPS_F64) load(new BatchedDenseSolver<double>(device, (int)count, n, m));
            else load(new BatchedDenseSolver<float>(device, (int)count, n, m));
        });
        if (rc != QPS_OK) { delete h->fused_batch; delete h; return rc; }
        *out = reinterpret_cast<qps_handle>(h);
        return QPS_OK;
    }
    for (int64_t b = 0; b < count; ++b) {   // shapes the fused pass does not cover: independent solvers, one after the other
        qps_handle one = nullptr;
        int rc = qps_create_dense(n, m, P + b * n * n, n, A ? A + b * m * n : nullptr, m, q + b * n, l ? l + b * m : nullptr,
                                  u ? u + b * m : nullptr, dtype, device, &one);
        if (rc != QPS_OK) { for (auto* s : h->batch) delete s; delete h; return rc; }
        Handle* oh = reinterpret_cast<Handle*>(one);
        h->batch.push_back(oh->impl); oh->impl = nullptr; delete oh;
    }
    *out = reinterpret_cast<qps_handle>(h);
    return QPS_OK;
}

QPS_API int32_t qps_solve_batch(qps_handle hh, double* x, const qps_params* p, qps_info* infos) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || (h->batch.empty() && !h->fused_batch)) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "not a batch handle");
    if (!x) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "x_inout is NULL");
    int rc = validate_params(h, p);
    if (rc != QPS_OK) return rc;
    if (h->fused_batch) return guarded(h, [&] { h->fused_batch->solve_batch(x, *p, infos); });
    for (size_t b = 0; b < h->batch.size(); ++b) {
        rc = guarded(h, [&] { h->batch[b]->solve(x + (int64_t)b * h->n, *p, infos ? infos + b : nullptr); });
        if (rc != QPS_OK) return rc;
    }
    return QPS_OK;
}

QPS_API int32_t qps_solve_batch_multi(int64_t count, int64_t n, int64_t m, const double* P, const double* A, const double* q, const double* l, const double* u,
                                      int32_t dtype, const int32_t* devices, int32_t num_workers, int32_t chunk, double* x, const qps_params* p, qps_info* infos,
                                      int32_t* worker_of, double* worker_seconds) {
    if (count <= 0 || n <= 0 || m < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need count >= 1, n >= 1 and m >= 0");
    if (!P || !q || !x || (m > 0 && (!A || !l || !u))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (!devices || num_workers <= 0 || num_workers > 256) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "need 1..256 workers and their device list");
    if (chunk > 65535) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "chunk must be at most 65535 (the batch handle's limit)");
    {
        const int rc = validate_params(nullptr, p);
        if (rc != QPS_OK) return rc;
    }
    for (int w = 0; w < num_workers; ++w) {
        const int dc = check_device(devices[w]);
        if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
        if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range in the worker list");
    }
    std::mutex err_mu; std::string err_msg; int err_code = QPS_OK;
    auto solve_range = [&](int w, int64_t b0, int64_t cnt) -> int {
        qps_handle h = nullptr;
        int rc = qps_create_dense_batch(cnt, n, m, P + b0 * n * n, A ? A + b0 * m * n : nullptr, q + b0 * n, l ? l + b0 * m : nullptr, u ? u + b0 * m : nullptr, dtype,
                                        devices[w], &h);
        if (rc == QPS_OK) rc = qps_solve_batch(h, x + b0 * n, p, infos ? infos + b0 : nullptr);
        if (rc != QPS_OK) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (err_code == QPS_OK) { err_code = rc; const char* msg = qps_last_error(h); err_msg = msg ? msg : ""; }   // (the message lives on this worker's thread)
        } else if (worker_of) {
            for (int64_t b = b0; b < b0 + cnt; ++b) worker_of[b] = w;
        }
        if (h) (void)qps_destroy(h);
        return rc;
    };
    // one host thread per worker, ranges dealt as batch_schedule.h describes (65535: the batch handle's limit)
    (void)run_batch_workers(count, num_workers, chunk, 65535, solve_range, worker_seconds);
    if (err_code != QPS_OK) return fail_with(nullptr, err_code, err_msg);
    return QPS_OK;
}

QPS_API int32_t qps_set_profiling(qps_handle hh, int32_t on) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h) return QPS_ERR_BAD_ARGUMENT;
    if (h->impl) { h->impl->prof.level = on; h->impl->prof.reset(); }
    if (h->fused_batch) { h->fused_batch->prof.level = on; h->fused_batch->prof.reset(); }
    for (auto* s : h->batch) { s->prof.level = on; s->prof.reset(); }
    return QPS_OK;
}

QPS_API int32_t qps_kernel_times(qps_handle hh, qps_kernel_time* out, int32_t cap, int32_t* count) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !count) return QPS_ERR_BAD_ARGUMENT;
    SolverBase* s = h->impl ? h->impl : (h->batch.empty() ? nullptr : h->batch[0]);
    Profiler* pf = s ? &s->prof : (h->fused_batch ? &h->fused_batch->prof : nullptr);
    if (!pf) return QPS_ERR_BAD_ARGUMENT;
    int k = 0;
    for (size_t i = 0; i < pf->names.size() && k < cap; ++i) {
        if (pf->stats[i].launches == 0) continue;
        if (out) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, pf->names[i].c_str(), sizeof(out[k].name) - 1);
            out[k].seconds = pf->stats[i].seconds; out[k].launches = pf->stats[i].launches; out[k].algo_bytes = pf->stats[i].algo_bytes;
        }
        ++k;
    }
    *count = k;
    return QPS_OK;
}

QPS_API int32_t qps_proxqp_default_params(qps_proxqp_params* p) {
    if (!p) return QPS_ERR_BAD_ARGUMENT;
    memset(p, 0, sizeof(*p));
    p->numIterations = 2000; p->epsAbs = 1e-7; p->epsRel = 1e-6; p->numItrConv = 50; p->rho = 1e2; p->sigma = 1e-2; p->adptRho = 1; p->tau = 10.0;   // ProxQP.jl:118
    return QPS_OK;
}

QPS_API int32_t qps_proxqp_create_dense(int64_t n, int64_t me, int64_t mi, const double* P, int64_t ldp, const double* q, const double* A,
                                        int64_t lda, const double* b, const double* C, int64_t ldc, const double* d, int32_t dtype,
                                        int32_t device, qps_handle* out) {
    if (!out) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "out handle pointer is NULL");
    *out = nullptr;
    if (n <= 0 || me < 0 || mi < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1, numEq >= 0, numInEq >= 0");
    if (n > (1 << 16) || me + mi > (1 << 20)) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "problem too large");
    if (!P || !q || (me > 0 && (!A || !b)) || (mi > 0 && (!C || !d))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (ldp < n || (me > 0 && lda < me) || (mi > 0 && ldc < mi)) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "leading dimension smaller than the row count");
    if (dtype != QPS_F64 && dtype != QPS_F32) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "unknown dtype");
    int dc = check_device(device);
    if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
    if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range");
    for (int64_t j = 0; j < n; ++j) {
        if (!all_finite(P + j * ldp, n, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "P contains NaN/Inf");
        if (me > 0 && !all_finite(A + j * lda, me, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "A contains NaN/Inf");
        if (mi > 0 && !all_finite(C + j * ldc, mi, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, "C contains NaN/Inf");
    }
    if (!all_finite(q, n, false) || (me > 0 && !all_finite(b, me, false)) || (mi > 0 && !all_finite(d, mi, false)))
        return fail_with(nullptr, QPS_ERR_NOT_FINITE, "q/b/d contain NaN/Inf");
    Handle* h = new Handle(); h->n = n; h->m = me + mi;
    int rc = guarded(nullptr, [&] { h->proxqp = make_proxqp(device, n, me, mi, dtype, P, ldp, A, lda, b, C, ldc, d, q); });
    if (rc != QPS_OK) { delete h; return rc; }
    *out = reinterpret_cast<qps_handle>(h);
    return QPS_OK;
}
// SparseProxQP (ProxQP.jl:71, :95-115): the CSC fields of the three SparseMatrixCSC inputs, kept sparse.  The linear system of UpdateX! is solved
// in its KKT form by the sparse L D L' plugin (k_sparse.hip: SparseProxQpSolver): symbolic analysis once, a rho update re-factorises numerically
// on the frozen pattern -- the role of AlignSparsePattern / GetNzvalDiagIdxs / UpdateM! / the pattern-reusing cholesky! (:184-190, :201-206, :335-372).
// QPS_PROXQP_SPARSE=0 densifies the inputs onto the dense solver instead (in-place dense re-factorisation, :193-199).
QPS_API int32_t qps_proxqp_create_csc(int64_t n, int64_t me, int64_t mi, const int64_t* Pcp, const int64_t* Pri, const double* Pnz, const double* q,
                                      const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* b, const int64_t* Ccp, const int64_t* Cri,
                                      const double* Cnz, const double* d, int32_t index_base, int32_t dtype, int32_t device, qps_handle* out) {
    if (!out) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "out handle pointer is NULL");
    *out = nullptr;
    if (n <= 0 || me < 0 || mi < 0) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1, numEq >= 0, numInEq >= 0");
    if (!Pcp || !q || (me > 0 && (!Acp || !b)) || (mi > 0 && (!Ccp || !d))) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL problem array");
    if (index_base != 0 && index_base != 1) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "index_base must be 0 or 1");
    if (dtype != QPS_F64 && dtype != QPS_F32) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "unknown dtype");
    auto validate = [&](const int64_t* cp, const int64_t* ri, const double* nz, int64_t rows, const char* name) -> int {
        if (rows == 0) return QPS_OK;
        if (cp[0] != index_base) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, std::string(name) + ": colptr does not start at index_base");
        for (int64_t j = 0; j < n; ++j) if (cp[j + 1] < cp[j]) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, std::string(name) + ": colptr not monotone");
        const int64_t nnz = cp[n] - index_base;
        if (nnz > 0 && (!ri || !nz)) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, std::string(name) + ": NULL index / value array");
        for (int64_t k = 0; k < nnz; ++k) {
            const int64_t i = ri[k] - index_base;
            if (i < 0 || i >= rows) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, std::string(name) + ": row index out of range");
        }
        if (!all_finite(nz, nnz, false)) return fail_with(nullptr, QPS_ERR_NOT_FINITE, std::string(name) + " contains NaN/Inf");
        return QPS_OK;
    };
    int rv = validate(Pcp, Pri, Pnz, n, "mP"); if (rv == QPS_OK) rv = validate(Acp, Ari, Anz, me, "mA"); if (rv == QPS_OK) rv = validate(Ccp, Cri, Cnz, mi, "mC");
    if (rv != QPS_OK) return rv;
    static const bool keep_sparse = [] { const char* e = getenv("QPS_PROXQP_SPARSE"); return !(e && atoi(e) == 0); }();
    if (keep_sparse && me + mi > 0) {
        if (n > 2000000000LL || me + mi > 2000000000LL) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "problem too large");
        if (!all_finite(q, n, false) || (me > 0 && !all_finite(b, me, false)) || (mi > 0 && !all_finite(d, mi, false)))
            return fail_with(nullptr, QPS_ERR_NOT_FINITE, "q/b/d contain NaN/Inf");
        int dc = check_device(device);
        if (dc == QPS_ERR_NO_DEVICE) return fail_with(nullptr, dc, "no HIP device visible: libqps_hip has no CPU fallback");
        if (dc != QPS_OK) return fail_with(nullptr, dc, "device index out of range");
        Handle* h = new Handle(); h->n = n; h->m = me + mi;
        int rc = guarded(nullptr, [&] { h->proxqp = make_proxqp_sparse(device, n, me, mi, dtype, Pcp, Pri, Pnz, q, Acp, Ari, Anz, b, Ccp, Cri, Cnz, d, index_base); });
        if (rc != QPS_OK) { delete h; return rc; }
        *out = reinterpret_cast<qps_handle>(h);
        return QPS_OK;
    }
    if (n > (1 << 16) || me + mi > (1 << 20)) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "problem too large to densify");
    std::vector<double> Pd, Ad, Cd;
    auto densify = [&](const int64_t* cp, const int64_t* ri, const double* nz, int64_t rows, std::vector<double>& D) {
        D.assign((size_t)std::max<int64_t>(rows, 1) * n, 0.0);
        if (rows == 0) return;
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = cp[j] - index_base; k < cp[j + 1] - index_base; ++k) D[(size_t)(ri[k] - index_base) + (size_t)j * rows] += nz[k];
    };
    densify(Pcp, Pri, Pnz, n, Pd); densify(Acp, Ari, Anz, me, Ad); densify(Ccp, Cri, Cnz, mi, Cd);
    return qps_proxqp_create_dense(n, me, mi, Pd.data(), n, q, me > 0 ? Ad.data() : nullptr, std::max<int64_t>(me, 1), b, mi > 0 ? Cd.data() : nullptr,
                                   std::max<int64_t>(mi, 1), d, dtype, device, out);
}
QPS_API int32_t qps_proxqp_init_kkt(qps_handle hh) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->proxqp) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "not a ProxQP handle");
    return guarded(h, [&] { h->proxqp->init_kkt(); });
}
QPS_API int32_t qps_proxqp_set_state(qps_handle hh, const double* x, const double* y, const double* z, const double* s) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->proxqp) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "not a ProxQP handle");
    if (!x || (h->proxqp->me > 0 && !y) || (h->proxqp->mi > 0 && (!z || !s))) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "NULL state vector");
    return guarded(h, [&] { h->proxqp->set_state(x, y, z, s); });
}
QPS_API int32_t qps_proxqp_get_state(qps_handle hh, double* x, double* y, double* z, double* s) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->proxqp) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "not a ProxQP handle");
    return guarded(h, [&] { h->proxqp->get_state(x, y, z, s); });
}
QPS_API int32_t qps_proxqp_solve(qps_handle hh, const qps_proxqp_params* p, qps_proxqp_report* rep) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h || !h->proxqp) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "not a ProxQP handle");
    if (!p) return fail_with(h, QPS_ERR_BAD_ARGUMENT, "params is NULL");
    if (p->numIterations < 0 || p->numItrConv <= 0 || !(p->rho > 0) || !(p->sigma >= 0) || !(p->tau > 0))
        return fail_with(h, QPS_ERR_BAD_ARGUMENT, "need numIterations >= 0, numItrConv > 0, rho > 0, sigma >= 0, tau > 0");
    return guarded(h, [&] { h->proxqp->solve(*p, rep); });
}

QPS_API int32_t qps_linsys_auto(int64_t n, int64_t m, int64_t nnzP, int64_t nnzA, int32_t sparse_input) {
    // SolveQuadraticProgram.jl:129-130 MAX_NUM_ROWS_L = 5000, MAX_DENSITY = 0.4; :143-151 (SolveQuadraticProgram.m:190-199)
    const double numRowsL = (double)n + (double)m;
    const double nnzDensity = ((double)nnzP + (double)nnzA) / (numRowsL * numRowsL);
    const bool directSol = (numRowsL <= 5000.0) && (nnzDensity <= 0.4);
    if (!directSol) return QPS_LINSYS_CG;
    return sparse_input ? QPS_LINSYS_KKT_LDL : QPS_LINSYS_CHOLESKY;
}

QPS_API int32_t qps_ldl_analyze(int64_t n, int64_t m, const int64_t* Pcp, const int64_t* Pri, const int64_t* Acp, const int64_t* Ari,
                                int32_t index_base, int64_t* perm_out, qps_ldl_report* rep) {
    if (n <= 0 || m < 0 || n + m > 2000000000LL) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "need n >= 1, m >= 0, n + m < 2^31");
    if (!Pcp || !Acp || (!Pri && Pcp[n] > index_base) || (!Ari && Acp[n] > index_base)) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "NULL index array");
    if (index_base != 0 && index_base != 1) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "index_base must be 0 or 1");
    for (int64_t j = 0; j < n; ++j) if (Pcp[j + 1] < Pcp[j] || Acp[j + 1] < Acp[j]) return fail_with(nullptr, QPS_ERR_BAD_ARGUMENT, "colptr not monotone");
    for (int64_t k = 0; k < Pcp[n] - index_base; ++k) if (Pri[k] - index_base < 0 || Pri[k] - index_base >= n) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "P row index out of range");
    for (int64_t k = 0; k < Acp[n] - index_base; ++k) if (Ari[k] - index_base < 0 || Ari[k] - index_base >= m) return fail_with(nullptr, QPS_ERR_BAD_DIMENSION, "A row index out of range");
    return guarded(nullptr, [&] {
        const char* e1 = getenv("QPS_LDL_MAX_TAIL"); const char* e2 = getenv("QPS_LDL_MIN_LEVEL"); const char* e3 = getenv("QPS_LDL_MAX_LEVELS");
        LdlSymbolic s;
        try { s = ldl_analyze((int)n, (int)m, Pcp, Pri, Acp, Ari, index_base, e1 ? atoi(e1) : 8192, e2 ? atoi(e2) : 64, e3 ? atoi(e3) : 4096); }
        catch (const std::runtime_error& e) { throw QpsError(QPS_ERR_UNSUPPORTED, e.what()); }
        if (perm_out) for (int64_t k = 0; k < n + m; ++k) perm_out[k] = (int64_t)s.perm[k] + index_base;
        if (rep) {
            rep->numRows = s.N; rep->numSparseColumns = s.Ns; rep->tailSize = s.Nt; rep->numSparseLevels = (int64_t)s.level_ptr.size() - 1;
            rep->treeHeight = s.levels_total; rep->nnzK = s.nnzK; rep->nnzL = s.nnzL_exact; rep->nnzStored = s.nnzL;
        }
    });
}

QPS_API int32_t qps_destroy(qps_handle hh) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!h) return QPS_OK;
    delete h->impl;
    delete h->fused_batch;
    delete h->proxqp;
    for (auto* s : h->batch) delete s;
    delete h;
    return QPS_OK;
}

QPS_API const char* qps_last_error(qps_handle hh) {
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (h) return h->err.c_str();
    return g_last_error.c_str();
}

QPS_API const char* qps_version(void) { return "qps-hip 0.1 (gfx950)"; }

}  // extern "C"
