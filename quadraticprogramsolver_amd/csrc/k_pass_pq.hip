// k_pass_pq.hip -- the fused single pass over G = [A; C] with the ProxQP.jl row updates (kernel template: k_pass_kernel.h,
// MODE 1).  One read of G per iteration does  v = G x  (UpdateS!/UpdateY!/UpdateZ! inputs, ProxQP.jl:230,239,248), the
// slack / dual updates (:227-249) and the column sums  G'[rho b - y ; rho (d - s) - z]  of the NEXT iteration's
// CalculateRhs! (:212-216).  Kept in its own translation unit so the ADMM instantiations of k_pass.hip compile exactly
// as they did before this mode existed.
#include "k_pass_kernel.h"

namespace qps {

template <typename T> int apass_proxqp_slabs(int NP, int MP) {
    if (NP > 8 * 512 * VecOf<T>::N) return 0;
    int rpw = 0;
    return apass_plan<T>(NP, MP, &rpw, 1);
}

template <typename T>
int apass_proxqp(hipStream_t st, const T* G, int64_t ld, int NP, int MP, int me, const T* x, T* x_scratch, T* slack, T* dual,
                 const T* g, T rho, T* part, int64_t part_ld) {
    int rpw = 0;
    const int W = apass_plan<T>(NP, MP, &rpw, 1);
    if (W <= 0 || NP > 8 * 512 * VecOf<T>::N) return 0;
    PassBatch pb;
    pb.slabs = W;
    pb.pq_me = me;
    const int chunk = 512 * VecOf<T>::N;
    const int kc = (NP + chunk - 1) / chunk;
#define QPS_PASS(KC, R) launch_pass<T, 512, KC, R, R, 1>(st, false, W, G, ld, NP, MP, rpw, x, x, x_scratch, slack, dual, g, g, T(1), rho, part, part, part_ld, nullptr, pb)
    if (kc <= 1) QPS_PASS(1, 4);
    else if (kc <= 2) QPS_PASS(2, 4);
    else if (kc <= 4) QPS_PASS(4, 4);
    else QPS_PASS(8, 2);
#undef QPS_PASS
    return W;
}

// Masked KKT product of the polishing step in one read of A:  out_lambda = mask (A v_x) - delta mask v_lambda  and the slabs of
// A'(mask v_lambda) (SolveQuadraticProgram.m:304-305 with the multiplier block kept at full length m).
template <typename T>
int apass_kkt(hipStream_t st, const T* A, int64_t ld, int NP, int MP, const T* vx, const T* vlam, const T* mask, T delta, T* out_lam,
              T* part, int64_t part_ld) {
    int rpw = 0;
    const int W = apass_plan<T>(NP, MP, &rpw, 1);
    if (W <= 0 || NP > 8 * 512 * VecOf<T>::N) return 0;
    PassBatch pb;
    pb.slabs = W;
    const int chunk = 512 * VecOf<T>::N;
    const int kc = (NP + chunk - 1) / chunk;
#define QPS_PASS(KC, R) launch_pass<T, 512, KC, R, R, 2>(st, false, W, A, ld, NP, MP, rpw, vx, vx, nullptr, out_lam, const_cast<T*>(vlam), mask, mask, delta, T(1), part, part, part_ld, nullptr, pb)
    if (kc <= 1) QPS_PASS(1, 4);
    else if (kc <= 2) QPS_PASS(2, 4);
    else if (kc <= 4) QPS_PASS(4, 4);
    else QPS_PASS(8, 2);
#undef QPS_PASS
    return W;
}

#define INST(T)                                          \
    template int apass_kkt<T>(hipStream_t, const T*, int64_t, int, int, const T*, const T*, const T*, T, T*, T*, int64_t); \
    template int apass_proxqp_slabs<T>(int, int);        \
    template int apass_proxqp<T>(hipStream_t, const T*, int64_t, int, int, int, const T*, T*, T*, T*, const T*, T, T*, int64_t);
INST(double)
INST(float)
#undef INST

}  // namespace qps
