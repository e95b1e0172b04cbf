// qps_polish.hip -- the polishing step of the MATLAB reference on the device.
//
// SolveQuadraticProgram.m:289-325: guess the active set from the sign of the multiplier y (:293-294), form the reduced KKT
// system  K t = g,  K = [P A_L' A_U'; A_L 0 0; A_U 0 0],  g = [-q; l(L); u(U)]  (:299-304), and refine  t  with MINRES on
// the regularised  KK = K + blkdiag(delta I, -delta I)  (:305, :314-320);  x = t(1:n) only if the last MINRES call
// converged (:322-325).  The Julia loop reserves the same kwargs (numItrPolish, delta, epsMinres, numItrMinres;
// SolveQuadraticProgram.jl:16-17) without using them, hence the step is opt-in here (qps_params.polish, qps_polish()).
//
// Device layout: the multiplier block stays at full length m with a 0/1 mask instead of compacting the active rows, so
// the products with K are the loop's own HBM-bound kernels: one GEMV over P and ONE masked pass over A per MINRES
// iteration (k_pass_pq.hip, MODE 2: row dots mask.(A v_x) and the column sums A'(mask.v_lambda) from the same read of A).
// MINRES (Paige & Saunders 1975; MathWorks' minres is not part of the reference tree) runs device-resident: the Lanczos
// and Givens scalars live in a ping-ponged state block, every vector kernel re-derives the scalars it needs from
// per-block partial sums added in a fixed order, and the host only polls the `done` word every few iterations.
#include <algorithm>
#include <functional>

#include "qps_internal.h"
#include "qps_kernels.h"
#include "qps_polish.h"
#include "wave_reduce.h"

namespace qps {

namespace {

struct MrState { double beta, oldb, dbar, epsln, phibar, cs, sn, bnorm, tol, alfa; int itn, done, flag, maxit; };

__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum_all(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double sum_partials(const double* p, int np, double* sh) {   // same order in every block
    double d = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) d += p[i];
    return block_sum(d, sh);
}

// mask = (y < 0 || y > 0), g = [-q ; y < 0 ? l : (y > 0 ? u : 0)]            SolveQuadraticProgram.m:293-299
template <typename T>
__global__ void k_pol_setup(int n, int NP, int m, int MP, const T* __restrict__ q, const T* __restrict__ l, const T* __restrict__ u,
                            const T* __restrict__ y, T* __restrict__ mask, T* __restrict__ g, int* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NP) g[i] = i < n ? -q[i] : T(0);
    if (i < MP) {
        T mk = T(0), gv = T(0);
        if (i < m) {
            const T yi = y[i];
            if (yi < T(0)) { mk = T(1); gv = l[i]; atomicAdd(&counts[0], 1); }
            else if (yi > T(0)) { mk = T(1); gv = u[i]; atomicAdd(&counts[1], 1); }
        }
        mask[i] = mk; g[NP + i] = gv;
    }
}
// unfused product only: out_lam = mask (A v_x) - delta mask v_lam (in place over A v_x), wl = mask v_lam
template <typename T> __global__ void k_pol_maskrow(int MP, const T* __restrict__ mask, const T* __restrict__ vlam, T delta, T* __restrict__ out_lam, T* __restrict__ wl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < MP) { const T mk = mask[i], w = mk * vlam[i]; out_lam[i] = mk * out_lam[i] - delta * w; wl[i] = w; }
}
template <typename T> __global__ void k_pol_axpby(int N, T a, const T* __restrict__ x, T b, const T* __restrict__ y, T* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) out[i] = a * x[i] + b * y[i];
}
// r1 = b - c, r2 = r1, w = w2 = 0; partials of ||r1||^2 and ||b||^2
template <typename T>
__global__ __launch_bounds__(256) void k_mr_begin(int N, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ r1, T* __restrict__ r2,
                                                  T* __restrict__ w0, T* __restrict__ w1, double* __restrict__ pa, double* __restrict__ pb) {
    __shared__ double sh[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0, e = 0.0;
    if (i < N) { const T bi = b[i], ri = bi - c[i]; r1[i] = ri; r2[i] = ri; w0[i] = T(0); w1[i] = T(0); d = (double)ri * (double)ri; e = (double)bi * (double)bi; }
    d = block_sum(d, sh); e = block_sum(e, sh);
    if (threadIdx.x == 0) { pa[blockIdx.x] = d; pb[blockIdx.x] = e; }
}
__global__ __launch_bounds__(256) void k_mr_begin_final(int np, const double* __restrict__ pa, const double* __restrict__ pb, MrState* st, double tol, int maxit) {
    __shared__ double sh[4];
    const double d = sum_partials(pa, np, sh), e = sum_partials(pb, np, sh);
    if (threadIdx.x == 0) {
        MrState s;
        const double beta1 = sqrt(d), bnorm = sqrt(e);
        s.beta = beta1; s.oldb = 0.0; s.dbar = 0.0; s.epsln = 0.0; s.phibar = beta1; s.cs = -1.0; s.sn = 0.0; s.bnorm = bnorm; s.tol = tol; s.alfa = 0.0;
        s.itn = 0; s.maxit = maxit; s.done = 0; s.flag = 1;
        if (!(beta1 == beta1) || isinf(beta1) || !(bnorm == bnorm) || isinf(bnorm)) { s.done = 1; s.flag = 1; }
        else if (beta1 <= tol * bnorm) { s.done = 1; s.flag = 0; }
        else if (maxit <= 0) { s.done = 1; s.flag = 1; }
        st[0] = s; st[1] = s;
    }
}
// partial of  alfa = v'(A v - (beta/oldb) r1),  v = y / beta,  A v = c / beta   (y is r2)
template <typename T>
__global__ __launch_bounds__(256) void k_mr_alfa(int N, const MrState* __restrict__ st, const T* __restrict__ y, const T* __restrict__ c,
                                                 const T* __restrict__ r1, double* __restrict__ pa) {
    if (st->done) return;
    __shared__ double sh[4];
    const T s = (T)(1.0 / st->beta), f = st->itn >= 1 ? (T)(st->beta / st->oldb) : T(0);
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0;
    if (i < N) { const T t = s * c[i] - f * r1[i]; d = (double)(s * y[i]) * (double)t; }
    d = block_sum(d, sh);
    if (threadIdx.x == 0) pa[blockIdx.x] = d;
}
// y_new = A v - (beta/oldb) r1 - (alfa/beta) r2, written over r1; partial ||y_new||^2
template <typename T>
__global__ __launch_bounds__(256) void k_mr_lanczos(int N, MrState* __restrict__ st, int np, const double* __restrict__ pa, const T* __restrict__ c,
                                                    T* __restrict__ r1, const T* __restrict__ r2, double* __restrict__ pb) {
    if (st->done) return;
    __shared__ double sh[4];
    const double alfa = sum_partials(pa, np, sh);
    const T s = (T)(1.0 / st->beta), f = st->itn >= 1 ? (T)(st->beta / st->oldb) : T(0), h = (T)(alfa / st->beta);
    const int i = blockIdx.x * 256 + threadIdx.x;
    double d = 0.0;
    if (i < N) { T t = s * c[i] - f * r1[i]; t -= h * r2[i]; r1[i] = t; d = (double)t * (double)t; }
    d = block_sum(d, sh);
    if (threadIdx.x == 0) { pb[blockIdx.x] = d; if (blockIdx.x == 0) st->alfa = alfa; }
}
// Givens rotation, w = (v - oldeps w1 - delta w2) / gamma (over w1's buffer), x += phi w; block 0 publishes the next state
template <typename T>
__global__ __launch_bounds__(256) void k_mr_update(int N, const MrState* __restrict__ cur, MrState* __restrict__ nxt, int np, const double* __restrict__ pb,
                                                   const T* __restrict__ vsrc, T* __restrict__ w1, const T* __restrict__ w2, T* __restrict__ x) {
    if (cur->done) { if (blockIdx.x == 0 && threadIdx.x == 0) *nxt = *cur; return; }
    __shared__ double sh[4];
    const double beta_new = sqrt(sum_partials(pb, np, sh));
    const double beta = cur->beta, alfa = cur->alfa, cs0 = cur->cs, sn0 = cur->sn, dbar0 = cur->dbar;
    const double oldeps = cur->epsln;
    const double delta = cs0 * dbar0 + sn0 * alfa;
    const double gbar = sn0 * dbar0 - cs0 * alfa;
    const double epsln = sn0 * beta_new;
    const double dbar = -cs0 * beta_new;
    const double gamma = fmax(sqrt(gbar * gbar + beta_new * beta_new), 2.220446049250313e-16);
    const double cs = gbar / gamma, sn = beta_new / gamma;
    const double phi = cs * cur->phibar, phibar = sn * cur->phibar;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        const T v = (T)(1.0 / beta) * vsrc[i];
        const T wn = (v - (T)oldeps * w1[i] - (T)delta * w2[i]) / (T)gamma;
        w1[i] = wn;
        x[i] += (T)phi * wn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        MrState s = *cur;
        s.oldb = beta; s.beta = beta_new; s.dbar = dbar; s.epsln = epsln; s.phibar = phibar; s.cs = cs; s.sn = sn; s.itn = cur->itn + 1;
        if (!(phibar == phibar) || isinf(phibar)) { s.done = 1; s.flag = 1; }
        else if (phibar <= s.tol * s.bnorm) { s.done = 1; s.flag = 0; }
        else if (beta_new == 0.0) { s.done = 1; s.flag = 0; }
        else if (s.itn >= s.maxit) { s.done = 1; s.flag = 1; }
        *nxt = s;
    }
}

inline dim3 blocks(int n) { return dim3((unsigned)((std::max(n, 1) + 255) / 256)); }

// Work space of one polishing call (freed on every exit path)
template <typename T> struct PolishWork {
    T* base = nullptr; double* pa = nullptr; double* pb = nullptr; MrState* state = nullptr; MrState* state_host = nullptr; int* counts = nullptr;
    ~PolishWork() {
        if (base) (void)hipFree(base); if (pa) (void)hipFree(pa); if (pb) (void)hipFree(pb); if (state) (void)hipFree(state);
        if (counts) (void)hipFree(counts); if (state_host) (void)hipHostFree(state_host);
    }
};

// tt = minres(KK, b, tol, maxit, x0 = tt): returns the flag (0 converged, 1 not)
template <typename T>
int minres_device(hipStream_t st, int N, const std::function<void(const T*, T, T*)>& matvec, T delta, const T* b, T* xsol, T* c, T* Ra, T* Rb,
                  T* W[2], PolishWork<T>& wk, double tol, int maxit, int* iters, double* relres) {
    const int nb = (N + 255) / 256;
    matvec(xsol, delta, c);                                                                        // KK x0
    hipLaunchKernelGGL((k_mr_begin<T>), dim3(nb), dim3(256), 0, st, N, b, c, Ra, Rb, W[0], W[1], wk.pa, wk.pb);
    hipLaunchKernelGGL(k_mr_begin_final, dim3(1), dim3(256), 0, st, nb, wk.pa, wk.pb, wk.state, tol, maxit);
    HIPC(hipMemcpyAsync(wk.state_host, wk.state, sizeof(MrState), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (wk.state_host->bnorm == 0.0) {                                                             // b == 0: x = 0, converged
        HIPC(hipMemsetAsync(xsol, 0, sizeof(T) * (size_t)N, st));
        *iters = 0; *relres = 0.0; return 0;
    }
    T* r1 = Ra; T* r2 = Rb;        // r2 doubles as the Lanczos vector y
    int cur = 0, launched = 0;
    const int poll = 8;
    while (!wk.state_host->done && launched < maxit) {
        const int burst = std::min(poll, maxit - launched);
        for (int k = 0; k < burst; ++k) {
            MrState* sc = wk.state + cur; MrState* sn = wk.state + (cur ^ 1);
            matvec(r2, delta, c);                                                                  // KK y  (= beta KK v)
            hipLaunchKernelGGL((k_mr_alfa<T>), dim3(nb), dim3(256), 0, st, N, sc, r2, c, r1, wk.pa);
            hipLaunchKernelGGL((k_mr_lanczos<T>), dim3(nb), dim3(256), 0, st, N, sc, nb, wk.pa, c, r1, r2, wk.pb);
            // r1's buffer now holds y_new; the old y (r2) is beta v
            hipLaunchKernelGGL((k_mr_update<T>), dim3(nb), dim3(256), 0, st, N, sc, sn, nb, wk.pb, r2, W[0], W[1], xsol);
            std::swap(r1, r2);                                                                     // r1 = old r2, r2 = y_new
            std::swap(W[0], W[1]);                                                                 // W[0] = w2 (the old w), W[1] = w (written over the old w2)
            cur ^= 1; ++launched;
        }
        HIPC(hipMemcpyAsync(wk.state_host, wk.state + cur, sizeof(MrState), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    *iters = wk.state_host->itn;
    *relres = wk.state_host->phibar / wk.state_host->bnorm;
    return wk.state_host->done ? wk.state_host->flag : 1;
}

}  // namespace

template <typename T> void polish_mask_rows(hipStream_t st, int MP, const T* mask, const T* vlam, T delta, T* out_lam, T* wl) {
    hipLaunchKernelGGL((k_pol_maskrow<T>), blocks(MP), dim3(256), 0, st, MP, mask, vlam, delta, out_lam, wl);
}

template <typename T>
void polish_dense(hipStream_t st, int64_t n, int64_t m, int NP, int MP, const T* P, const T* A, const T* q, const T* l, const T* u, const T* y,
                  T* x, T* part, const qps_params& p, PolishReport* rep) {
    const char* unf = getenv("QPS_POLISH_UNFUSED");                               // 1: force the two-GEMV product (the path of shapes the pass kernel does not cover)
    const bool fusedA = MP > 0 && apass_proxqp_slabs<T>(NP, MP) > 0 && !(unf && atoi(unf) != 0);
    // out = K v + delta blkdiag(I, -I) v  with the multiplier block masked                         :304-305
    PolishProduct<T> kmat = [&](const T* v, T delta, T* out, const T* mask, T* scratch) {
        gemv_rows<T>(st, P, NP, v, out, v, T(1), delta, 0, NP, 0, NP, 0);                           // P v_x + delta v_x
        if (MP <= 0) return;
        if (fusedA) {
            const int G = apass_kkt<T>(st, A, NP, NP, MP, v, v + NP, mask, delta, out + NP, part, NP);
            colsum<T>(st, part, NP, G, out, T(1), nullptr, T(0), out, NP);                          // + A'(mask v_lambda)
        } else {
            gemv_rows<T>(st, A, NP, v, out + NP, nullptr, T(1), T(0), 0, MP, 0, NP, 0);             // A v_x
            polish_mask_rows<T>(st, MP, mask, v + NP, delta, out + NP, scratch);
            const int tiles = gemv_cols_partial<T>(st, A, NP, scratch, nullptr, T(1), T(0), part, NP, MP, NP);
            colsum<T>(st, part, NP, tiles, out, T(1), nullptr, T(0), out, NP);
        }
    };
    polish_with<T>(st, n, m, NP, MP, q, l, u, y, x, p, rep, kmat);
}

template <typename T>
void polish_with(hipStream_t st, int64_t n, int64_t m, int NP, int MP, const T* q, const T* l, const T* u, const T* y, T* x, const qps_params& p,
                 PolishReport* rep, const PolishProduct<T>& kmat) {
    PolishReport r;
    const double t0 = now_s();
    if (p.numItrPolish <= 0) { if (rep) *rep = r; return; }                                        // :292
    const int N = NP + MP;
    PolishWork<T> wk;
    wk.base = dalloc<T>((int64_t)10 * N + MP, st);
    T* g = wk.base; T* t = g + N; T* tt = t + N; T* rhs = tt + N; T* c = rhs + N; T* Ra = c + N; T* Rb = Ra + N;
    T* W[3] = {Rb + N, Rb + 2 * (int64_t)N, Rb + 3 * (int64_t)N}; T* mask = Rb + 4 * (int64_t)N;
    const int nb = (N + 255) / 256;
    wk.pa = dalloc<double>(nb, st); wk.pb = dalloc<double>(nb, st); wk.state = dalloc<MrState>(2, st); wk.counts = dalloc<int>(4, st);
    HIPC(hipHostMalloc((void**)&wk.state_host, sizeof(MrState)));
    hipLaunchKernelGGL((k_pol_setup<T>), blocks(std::max(NP, MP)), dim3(256), 0, st, (int)n, NP, (int)m, MP, q, l, u, y, mask, g, wk.counts);
    int counts[2] = {0, 0};
    HIPC(hipMemcpyAsync(counts, wk.counts, sizeof(counts), hipMemcpyDeviceToHost, st));

    std::function<void(const T*, T, T*)> matvec = [&](const T* v, T delta, T* out) { kmat(v, delta, out, mask, W[2]); };
    HIPC(hipMemsetAsync(t, 0, sizeof(T) * (size_t)N, st));                                          // :307
    HIPC(hipMemsetAsync(tt, 0, sizeof(T) * (size_t)N, st));                                         // :308
    int flag = -1;                                                                                  // :311
    for (int jj = 0; jj < p.numItrPolish; ++jj) {                                                   // :314
        matvec(t, T(0), c);
        hipLaunchKernelGGL((k_pol_axpby<T>), blocks(N), dim3(256), 0, st, N, T(1), g, T(-1), c, rhs);   // g - K t
        int it = 0; double rr = NAN;
        flag = minres_device<T>(st, N, matvec, (T)p.delta, rhs, tt, c, Ra, Rb, W, wk, p.epsMinres, p.numItrMinres, &it, &rr);   // :315
        r.minresIterations += it; r.refinements = jj + 1; r.relres = rr;
        if (flag) break;                                                                            // :316-318
        hipLaunchKernelGGL((k_pol_axpby<T>), blocks(N), dim3(256), 0, st, N, T(1), t, T(1), tt, t);     // :319
    }
    if (flag == 0) HIPC(hipMemcpyAsync(x, t, sizeof(T) * (size_t)NP, hipMemcpyDeviceToDevice, st));     // :322-325
    HIPC(hipStreamSynchronize(st));
    r.flag = flag; r.numLower = counts[0]; r.numUpper = counts[1]; r.seconds = now_s() - t0;
    if (rep) *rep = r;
}

#define INST(T)                                                                                                               \
    template void polish_mask_rows<T>(hipStream_t, int, const T*, const T*, T, T*, T*);                                       \
    template void polish_with<T>(hipStream_t, int64_t, int64_t, int, int, const T*, const T*, const T*, const T*, T*, const qps_params&, \
                                 PolishReport*, const PolishProduct<T>&);                                                     \
    template void polish_dense<T>(hipStream_t, int64_t, int64_t, int, int, const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, \
                                  const qps_params&, PolishReport*);
INST(double)
INST(float)
#undef INST

}  // namespace qps
