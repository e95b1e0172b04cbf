// qps_proxqp.hip -- the reference's second solver form (ProxQP.jl) on the same device kernels.
//
//     min 1/2 x'Px + q'x   s.t.  A x = b,  C x <= d                               (ProxQP.jl:118-124)
//
// With the stacked constraint matrix G = [A; C] the iteration of ProxQP.jl:135-149 is the ADMM skeleton of the main path:
//     M = P + sigma I + rho G'G  (UpdateM! :175-181)      ->  same assembly + Cholesky + sweep matrix as DenseSolver
//     r = sigma x - q + G'w,  w = [rho b - y ; rho (d - s) - z]   (CalculateRhs! :208-219)
//     x = M^{-1} r                                          (UpdateX! :221-225)   ->  fused sweeps
//     v = G x;  s = max(d - z/rho - v, 0) (:227-233);  y += rho (v - b) (:235-240);  z = max(z + rho (s - d) + rho v, 0) (:242-249)
// CheckConvergence! (:252-298) runs every numItrConv iterations; the loop never breaks on convergence (:156).
// The dense convenience constructor's initialisation (:73-93: x, y from the equality-constrained KKT system) is done on the
// device by the range-space method: x = -P^{-1}(q + A'y), (A P^{-1} A') y = -(b + A P^{-1} q).
#include <algorithm>

#include "qps_internal.h"
#include "qps_kernels.h"
#include "qps_proxqp.h"
#include "k_proxqp_rows.h"

namespace qps {

namespace {

using namespace pqrows;

template <typename T> __global__ void k_pq_scale_add(int n, T a, const T* __restrict__ x, T b, const T* __restrict__ y, T* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a * x[i] + (y ? b * y[i] : T(0));
}
// dst = lower triangle of src (upper zero): the sweep matrix without its mirrored half, usable as a plain GEMM operand
template <typename T> __global__ void k_pq_lower(int NP, const T* __restrict__ src, T* __restrict__ dst) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j < NP) dst[(int64_t)i * NP + j] = (j <= i) ? src[(int64_t)i * NP + j] : T(0);
}

inline dim3 g1(int n) { return dim3((unsigned)((std::max(n, 1) + 255) / 256)); }

template <typename T> struct ProxQpSolver : ProxQpBase {
    hipStream_t st = nullptr; HandleResources res;
    int NP = 0, MP = 0, MEP = 0, mtot = 0, nb = 0, part_tiles = 0;
    T *G = nullptr, *Aonly = nullptr, *P = nullptr, *q = nullptr, *g = nullptr, *dual = nullptr, *slack = nullptr, *x = nullptr;
    T *w = nullptr, *v = nullptr, *de = nullptr, *di = nullptr, *tt = nullptr, *yv = nullptr, *xx = nullptr, *part = nullptr, *sw_part = nullptr;
    T *PI = nullptr, *KK = nullptr, *M = nullptr, *S = nullptr, *tmp = nullptr, *dinv = nullptr, *X1 = nullptr, *X2 = nullptr, *X3 = nullptr;
    int* fail = nullptr; unsigned long long* slots = nullptr; unsigned long long* slots_host = nullptr; double* stage = nullptr;
    bool have_K = false;

    ProxQpSolver(int dev, int64_t n_, int64_t me_, int64_t mi_) {
        device = dev; n = n_; me = me_; mi = mi_;
        HIPC(hipSetDevice(device));
        res = acquire_resources(device, 0);   // recycled stream + pinned block (qps_internal.h)
        st = res.st;
        mtot = (int)(me + mi);
        NP = roundup(n, 64); MP = roundup(mtot, 64); MEP = roundup(std::max<int64_t>(me, 1), 64);
        const int64_t nn = (int64_t)NP * NP;
        G = dalloc<T>((int64_t)MP * NP, st); Aonly = dalloc<T>((int64_t)MEP * NP, st); P = dalloc<T>(nn, st); q = dalloc<T>(NP, st); x = dalloc<T>(NP, st);
        g = dalloc<T>(MP, st); dual = dalloc<T>(MP, st); slack = dalloc<T>(MP, st); w = dalloc<T>(MP, st); v = dalloc<T>(MP, st); de = dalloc<T>(MP, st); di = dalloc<T>(MP, st);
        tt = dalloc<T>(NP, st); yv = dalloc<T>(NP, st); xx = dalloc<T>(NP, st); X1 = dalloc<T>(NP, st); X2 = dalloc<T>(NP, st); X3 = dalloc<T>(NP, st);
        part_tiles = std::max(gemv_cols_tiles(MP), apass_proxqp_slabs<T>(NP, MP));
        part = dalloc<T>((int64_t)std::max(part_tiles, 1) * NP, st);
        sw_part = dalloc<T>((int64_t)std::max(sweep_fused_slabs<T>(NP), 1) * NP, st);
        PI = dalloc<T>(nn, st); KK = dalloc<T>(nn, st); M = dalloc<T>(nn, st); S = dalloc<T>(nn, st); tmp = dalloc<T>(nn, st); dinv = dalloc<T>((int64_t)(NP / 64) * 4096, st);
        fail = dalloc<int>(4, st); slots = dalloc<unsigned long long>(16, st);
        slots_host = reinterpret_cast<unsigned long long*>(res.pinned);
        stage = dalloc<double>(std::max<int64_t>((int64_t)MP * NP, nn) + 64, st);
    }
    ~ProxQpSolver() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        void* ptrs[] = {G, Aonly, P, q, g, dual, slack, x, w, v, de, di, tt, yv, xx, part, sw_part, PI, KK, M, S, tmp, dinv, X1, X2, X3, fail, slots, stage};
        for (void* p_ : ptrs) if (p_) (void)hipFree(p_);
        if (res.st) recycle_resources(device, res);
    }
    void put_vec(const double* h, T* d, int64_t c) {
        if (c <= 0) return;
        HIPC(hipMemcpyAsync(stage, h, sizeof(double) * (size_t)c, hipMemcpyHostToDevice, st));
        convert_copy<T>(st, stage, d, c);
        HIPC(hipStreamSynchronize(st));
    }
    void get_vec(const T* d, double* h, int64_t c) {
        if (c <= 0) return;
        convert_back<T>(st, d, stage, c);
        HIPC(hipMemcpyAsync(h, stage, sizeof(double) * (size_t)c, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    void put_matrix(const double* h, int64_t ldh, int rows, int cols, T* d) {   // column-major host -> row-major device rows (ld NP)
        if (rows <= 0 || cols <= 0) return;
        HIPC(hipMemcpy2DAsync(stage, sizeof(double) * (size_t)rows, h, sizeof(double) * (size_t)ldh, sizeof(double) * (size_t)rows, (size_t)cols, hipMemcpyHostToDevice, st));
        import_colmajor<T>(st, stage, rows, rows, cols, d, NP);
        HIPC(hipStreamSynchronize(st));
    }
    void load(const double* Ph, int64_t ldp, const double* Ah, int64_t lda, const double* bh, const double* Ch, int64_t ldc, const double* dh, const double* qh) {
        put_matrix(Ph, ldp, (int)n, (int)n, P);
        put_matrix(Ah, lda, (int)me, (int)n, G);
        put_matrix(Ah, lda, (int)me, (int)n, Aonly);
        put_matrix(Ch, ldc, (int)mi, (int)n, G + (int64_t)me * NP);
        put_vec(qh, q, n); put_vec(bh, g, me); put_vec(dh, g + me, mi);
    }
    void check_fail(const char* what) {
        int f = 0;
        HIPC(hipMemcpyAsync(&f, fail, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        if (f != 0) { char b_[256]; snprintf(b_, sizeof b_, "Cholesky of %s broke down: non-positive pivot at column %d", what, f); throw QpsError(QPS_ERR_FACTORIZATION, b_); }
    }
    // tt -> xx = (L L')^{-1} tt with the sweep matrix Sm (NPm x NPm)
    void sweeps(const T* Sm, int NPm, int nbm, T* rhs, T* out, T* work) {
        const int nblk = (NPm + nbm - 1) / nbm;
        if (nblk == 1 && sweep_fused_supported<T>(NPm)) {
            const int Gs = sweep_fused<T>(st, Sm, NPm, NPm, rhs, sw_part, NPm);
            colsum<T>(st, sw_part, NPm, Gs, nullptr, T(0), nullptr, T(0), out, NPm);
            return;
        }
        for (int J = 0; J < nblk; ++J) {
            const int r0 = J * nbm, r1 = std::min(NPm, r0 + nbm);
            gemv_rows<T>(st, Sm, NPm, rhs, work, nullptr, T(1), T(0), r0, r1, r0, r1, 1);
            if (r1 < NPm) gemv_rows<T>(st, Sm, NPm, work, rhs, rhs, T(-1), T(1), r1, NPm, r0, r1, 0);
        }
        for (int J = nblk - 1; J >= 0; --J) {
            const int r0 = J * nbm, r1 = std::min(NPm, r0 + nbm);
            gemv_rows<T>(st, Sm, NPm, work, out, nullptr, T(1), T(0), r0, r1, r0, r1, 2);
            if (r0 > 0) gemv_rows<T>(st, Sm, NPm, out, work, work, T(-1), T(1), 0, r0, r0, r1, 0);
        }
    }
    // one inverted block over the whole factor while the fused forward+backward sweep covers it (as DenseSolver does), else 4096-blocks
    int pick(int NPm) { const int cap = sweep_fused_supported<T>(NPm) ? 32768 : 4096; int p = 64; while (p < NPm && p < cap) p *= 2; return p; }

    // UpdateDecomposition! (ProxQP.jl:193-199): M = P + rho K + sigma I, Cholesky, sweep matrix
    void update_decomposition(double rho, double sigma) {
        if (!have_K) {
            gemm<T>(st, NP, NP, MP, T(1), G, NP, false, G, NP, false, T(0), KK, NP, true);          // mK = A'A + C'C (:42-46)
            have_K = true;
        }
        make_PI<T>(st, (int)n, NP, P, (T)sigma, PI);
        assemble_M<T>(st, NP, PI, KK, (T)rho, M);                                                   // :178-180
        cholesky<T>(st, NP, M, dinv, fail, 1, chol_scratch_fits(NP) ? S : nullptr);                 // :196
        nb = pick(NP);
        build_sweep_matrix<T>(st, NP, nb, M, dinv, S, tmp);
        check_fail("P + rho (A'A + C'C) + sigma I");
    }

    void set_state(const double* xh, const double* yh, const double* zh, const double* sh) override {
        HIPC(hipSetDevice(device));
        put_vec(xh, x, n); put_vec(yh, dual, me); put_vec(zh, dual + me, mi);
        HIPC(hipMemsetAsync(slack, 0, sizeof(T) * MP, st));
        put_vec(sh, slack + me, mi);
    }
    void get_state(double* xh, double* yh, double* zh, double* sh) override {
        HIPC(hipSetDevice(device));
        if (xh) get_vec(x, xh, n);
        if (yh) get_vec(dual, yh, me);
        if (zh) get_vec(dual + me, zh, mi);
        if (sh) get_vec(slack + me, sh, mi);
    }

    // ProxQP.jl:73-93 on the device (range-space method; P is SPD as the reference requires, A needs full row rank)
    void init_kkt() override {
        HIPC(hipSetDevice(device));
        const int nbP = pick(NP);
        make_PI<T>(st, (int)n, NP, P, T(0), PI);                     // P with identity on the padding
        HIPC(hipMemcpyAsync(M, PI, sizeof(T) * (size_t)NP * NP, hipMemcpyDeviceToDevice, st));
        cholesky<T>(st, NP, M, dinv, fail, 1, chol_scratch_fits(NP) ? S : nullptr);
        build_sweep_matrix<T>(st, NP, nbP, M, dinv, S, tmp);         // S: sweep matrix of P
        check_fail("P (KKT initialisation)");
        HIPC(hipMemsetAsync(dual, 0, sizeof(T) * MP, st));
        if (me > 0) {
            if (nbP < NP) throw QpsError(QPS_ERR_UNSUPPORTED, "KKT initialisation of the dense solver needs n <= 16384 fp64 / 32768 fp32 (pass an explicit state, or CSC inputs for the sparse solver)");
            // B = W_P A'  (NP x MEP), W_P = inv(L_P) = lower triangle of S ; Schur = B'B = A P^{-1} A'
            T* Wl = tmp;                                             // lower-only copy of the sweep matrix
            hipLaunchKernelGGL((k_pq_lower<T>), dim3((NP + 255) / 256, NP), dim3(256), 0, st, NP, S, Wl);
            T* B = PI;                                               // reuse: NP x MEP (<= NP x NP needs MEP <= NP)
            T* Sch = KK; have_K = false;                             // MEP x MEP, ld MEP
            if (MEP > NP) throw QpsError(QPS_ERR_UNSUPPORTED, "KKT initialisation needs numEq <= n");
            gemm<T>(st, NP, MEP, NP, T(1), Wl, NP, true, Aonly, NP, true, T(0), B, MEP, false, 1, 0, 0, 0, 2);
            gemm<T>(st, MEP, MEP, NP, T(1), B, MEP, false, B, MEP, false, T(0), Sch, MEP, false);
            T* SchPI = M;                                            // Schur + identity on its padding
            make_PI<T>(st, (int)me, MEP, Sch, T(0), SchPI);
            T* dinv2 = dinv; T* S2 = Wl;                             // Wl no longer needed after the two GEMMs
            cholesky<T>(st, MEP, SchPI, dinv2, fail, 1, chol_scratch_fits(MEP) ? S2 : nullptr);
            const int nb2 = pick(MEP);
            build_sweep_matrix<T>(st, MEP, nb2, SchPI, dinv2, S2, B);
            check_fail("A P^{-1} A' (KKT initialisation: A must have full row rank)");
            // u = P^{-1} q ; t = b + A u ; y = -Schur^{-1} t
            HIPC(hipMemcpyAsync(tt, q, sizeof(T) * NP, hipMemcpyDeviceToDevice, st));
            sweeps(S, NP, nbP, tt, xx, yv);
            gemv_rows<T>(st, Aonly, NP, xx, w, g, T(1), T(1), 0, MEP, 0, NP, 0);                    // w = A u + b  (padding rows: 0 + g? g holds d there)
            HIPC(hipMemsetAsync(w + me, 0, sizeof(T) * (size_t)(MEP - me), st));
            sweeps(S2, MEP, nb2, w, de, di);                                                        // de = Schur^{-1} t
            hipLaunchKernelGGL((k_pq_scale_add<T>), g1((int)me), dim3(256), 0, st, (int)me, T(-1), de, T(0), (const T*)nullptr, dual);   // y
        }
        // x = -P^{-1}(q + A'y)
        HIPC(hipMemsetAsync(de, 0, sizeof(T) * MP, st));
        if (me > 0) HIPC(hipMemcpyAsync(de, dual, sizeof(T) * (size_t)me, hipMemcpyDeviceToDevice, st));
        const int tiles = gemv_cols_partial<T>(st, Aonly, NP, de, nullptr, T(1), T(0), part, NP, MEP, NP);
        colsum<T>(st, part, NP, tiles, q, T(1), nullptr, T(0), tt, NP);
        sweeps(S, NP, nbP, tt, xx, yv);
        hipLaunchKernelGGL((k_pq_scale_add<T>), g1(NP), dim3(256), 0, st, NP, T(-1), xx, T(0), (const T*)nullptr, x);
        // s = max(d - C x, 0), z = 0                                                               (:88-89)
        gemv_rows<T>(st, G, NP, x, v, nullptr, T(1), T(0), 0, MP, 0, NP, 0);
        hipLaunchKernelGGL((k_pq_init_s<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, v, dual, slack);
        HIPC(hipStreamSynchronize(st));
        have_K = false;   // KK / PI were used as scratch
    }

    void solve(const qps_proxqp_params& p, qps_proxqp_report* rep) override {
        HIPC(hipSetDevice(device));
        double rho = p.rho; const double sigma = p.sigma;
        int converged = 0, conv_it = p.numIterations; double resP = INFINITY, resD = INFINITY, rho_rep = p.rho;
        update_decomposition(rho, sigma);                                                           // ProxQP.jl:131
        // Fused iteration (default): one read of G per iteration -- the pass leaves the slabs of G'w for the next right-hand
        // side, so only the first iteration, the iteration after a check and a rho change form them with the two-kernel path.
        const bool fused = p.loopVariant != 1 && apass_proxqp_slabs<T>(NP, MP) > 0;
        int slabs = 0;                                                                              // > 0: part holds the slabs of G'w for the current (s, y, z)
        for (int ii = 1; ii <= p.numIterations; ++ii) {                                             // :135
            const bool check = (ii % p.numItrConv == 0);
            if (slabs == 0) {
                hipLaunchKernelGGL((k_pq_w<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, dual, slack, (T)rho, w);
                slabs = gemv_cols_partial<T>(st, G, NP, w, nullptr, T(1), T(0), part, NP, MP, NP);        // G'w (:213,216)
            }
            colsum<T>(st, part, NP, slabs, x, (T)sigma, q, T(-1), tt, NP);                            // :211
            sweeps(S, NP, nb, tt, x, yv);                                                           // :224 (no relaxation: x = M^{-1} r)
            if (fused && !check) {
                slabs = apass_proxqp<T>(st, G, NP, NP, MP, (int)me, x, xx, slack, dual, g, (T)rho, part, NP);   // :227-249 + next :212-216
            } else {
                gemv_rows<T>(st, G, NP, x, v, nullptr, T(1), T(0), 0, MP, 0, NP, 0);                 // A x and C x (kept: the check reads v)
                hipLaunchKernelGGL((k_pq_update<T>), g1(mtot), dim3(256), 0, st, (int)me, mtot, g, v, dual, slack, (T)rho);   // :227-249
                slabs = 0;
            }
            if (check) {                                                                            // :151  CheckConvergence! :252-298
                gemv_rows<T>(st, P, NP, x, X1, nullptr, T(1), T(0), 0, NP, 0, NP, 0);                // :261
                hipLaunchKernelGGL((k_pq_split<T>), g1(MP), dim3(256), 0, st, (int)me, MP, dual, de, di);
                int t2 = gemv_cols_partial<T>(st, G, NP, de, nullptr, T(1), T(0), part, NP, MP, NP);
                colsum<T>(st, part, NP, t2, nullptr, T(0), nullptr, T(0), X2, NP);                   // A'y (:262)
                t2 = gemv_cols_partial<T>(st, G, NP, di, nullptr, T(1), T(0), part, NP, MP, NP);
                colsum<T>(st, part, NP, t2, nullptr, T(0), nullptr, T(0), X3, NP);                   // C'z (:263)
                HIPC(hipMemsetAsync(slots, 0, 16 * sizeof(unsigned long long), st));
                hipLaunchKernelGGL((k_pq_norms<T>), dim3(64), dim3(256), 0, st, (int)n, (int)me, mtot, v, g, slack, X1, X2, X3, q, slots);
                HIPC(hipMemcpyAsync(slots_host, slots, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                const CheckOutcome co = decide(slots_host, p, rho);                                 // :266-294
                const bool updated = co.updated; rho = co.rho; converged = co.converged ? 1 : 0; resP = co.resPrim; resD = co.resDual;
                if (converged) conv_it = ii;                                                        // :155-157 (no break)
                if (updated) { update_decomposition(rho, sigma); rho_rep = rho; }                   // :159-165
            }
        }
        HIPC(hipStreamSynchronize(st));
        if (rep) { rep->converged = converged; rep->iterations = conv_it; rep->rho = rho_rep; rep->sigma = sigma; rep->resPrim = resP; rep->resDual = resD; }
    }
};

}  // namespace

ProxQpBase* make_proxqp(int device, int64_t n, int64_t me, int64_t mi, int dtype, const double* P, int64_t ldp, const double* A, int64_t lda,
                        const double* b, const double* C, int64_t ldc, const double* d, const double* q) {
    if (dtype == QPS_F64) { auto* s = new ProxQpSolver<double>(device, n, me, mi); try { s->load(P, ldp, A, lda, b, C, ldc, d, q); } catch (...) { delete s; throw; } return s; }
    auto* s = new ProxQpSolver<float>(device, n, me, mi);
    try { s->load(P, ldp, A, lda, b, C, ldc, d, q); } catch (...) { delete s; throw; }
    return s;
}

}  // namespace qps
