// k_proxqp_rows.h -- row-wise kernels and the host decision of the ProxQP.jl iteration, shared by the dense (qps_proxqp.hip) and the sparse
// (k_sparse.hip: SparseProxQpSolver) solver of that form.  G = [A; C] stacked, g = [b; d], dual = [y; z], s on the inequality rows.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

#include <hip/hip_runtime.h>

#include "../../include/qps.h"

namespace qps {
namespace pqrows {

__device__ __forceinline__ unsigned long long absb(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }

// w = [rho b - y ; rho (d - s) - z]                                                  ProxQP.jl:212,215
template <typename T> __global__ void k_pq_w(int me, int mtot, const T* __restrict__ g, const T* __restrict__ dual, const T* __restrict__ s, T rho, T* __restrict__ w) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= mtot) return;
    w[r] = (r < me) ? rho * g[r] - dual[r] : rho * (g[r] - s[r]) - dual[r];
}
// s, y, z updates from v = G x                                                       ProxQP.jl:227-249
template <typename T> __global__ void k_pq_update(int me, int mtot, const T* __restrict__ g, const T* __restrict__ v, T* __restrict__ dual, T* __restrict__ s, T rho) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= mtot) return;
    const T rho1 = T(1) / rho, vr = v[r], gr = g[r];
    if (r < me) { T y = dual[r]; y -= rho * gr; y += rho * vr; dual[r] = y; }          // :238-239 (same operation order)
    else {
        T sv = gr - rho1 * dual[r]; sv += -vr; sv = sv > T(0) ? sv : T(0);             // :230-232
        s[r] = sv;
        T z = dual[r] + rho * (sv - gr); z += rho * vr; dual[r] = z > T(0) ? z : T(0); // :246-248
    }
}
// s = max(d - C x, 0), z = 0 on the inequality rows (ProxQP.jl:88-89); eq rows: slack unused (0)
template <typename T> __global__ void k_pq_init_s(int me, int mtot, const T* __restrict__ g, const T* __restrict__ v, T* __restrict__ dual, T* __restrict__ s) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= mtot) return;
    if (r < me) s[r] = T(0);
    else { const T t = g[r] - v[r]; s[r] = t > T(0) ? t : T(0); dual[r] = T(0); }
}
// masked copies of the dual: eq part / ineq part (for A'y and C'z, ProxQP.jl:262-263)
template <typename T> __global__ void k_pq_split(int me, int mtot, const T* __restrict__ dual, T* __restrict__ de, T* __restrict__ di) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= mtot) return;
    de[r] = r < me ? dual[r] : T(0);
    di[r] = r < me ? T(0) : dual[r];
}
// the twelve inf-norms of CheckConvergence! (:266-270): slots 0 |Ax-b| 1 |Cx-d+s| 2 |Ax| 3 |b| 4 |Cx| 5 |d| 6 |s|
//                                                        7 |Px+A'y+C'z+q| 8 |Px| 9 |A'y| 10 |C'z| 11 |q|
template <typename T>
__global__ void k_pq_norms(int n, int me, int mtot, const T* __restrict__ v, const T* __restrict__ g, const T* __restrict__ s, const T* __restrict__ X1,
                           const T* __restrict__ X2, const T* __restrict__ X3, const T* __restrict__ q, unsigned long long* __restrict__ slots) {
    unsigned long long m[12] = {0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < max(n, mtot); i += gridDim.x * 256) {
        if (i < mtot) {
            if (i < me) { m[0] = max(m[0], absb((double)(v[i] - g[i]))); m[2] = max(m[2], absb((double)v[i])); m[3] = max(m[3], absb((double)g[i])); }
            else { m[1] = max(m[1], absb((double)(v[i] - g[i] + s[i]))); m[4] = max(m[4], absb((double)v[i])); m[5] = max(m[5], absb((double)g[i])); m[6] = max(m[6], absb((double)s[i])); }
        }
        if (i < n) {
            m[7] = max(m[7], absb((double)(X1[i] + X2[i] + X3[i] + q[i]))); m[8] = max(m[8], absb((double)X1[i]));
            m[9] = max(m[9], absb((double)X2[i])); m[10] = max(m[10], absb((double)X3[i])); m[11] = max(m[11], absb((double)q[i]));
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m[k], o, 64); m[k] = t > m[k] ? t : m[k]; }
        if ((threadIdx.x & 63) == 0 && m[k]) atomicMax(&slots[k], m[k]);
    }
}
// h = [b ; d - s]: the constraint part of the KKT right-hand side [sigma x - q ; h - dual / rho] whose reduced form is CalculateRhs! (ProxQP.jl:208-219)
template <typename T> __global__ void k_pq_h(int me, int mtot, const T* __restrict__ g, const T* __restrict__ s, T* __restrict__ h) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= mtot) return;
    h[r] = (r < me) ? g[r] : g[r] - s[r];
}

// Host part of CheckConvergence! (ProxQP.jl:266-294) on the twelve norms of k_pq_norms (bit patterns of non-negative doubles)
struct CheckOutcome { double resPrim, resDual, rho; bool converged, updated; };
inline CheckOutcome decide(const unsigned long long* slots_host, const qps_proxqp_params& p, double rho) {
    double nv[12];
    for (int k = 0; k < 12; ++k) { long long bits = (long long)slots_host[k]; std::memcpy(&nv[k], &bits, sizeof(double)); }
    CheckOutcome o;
    const double normResPrim = std::max(nv[0], nv[1]);                                  // :266
    const double normResDual = nv[7];                                                   // :267
    const double maxNormPrim = std::max(std::max(std::max(nv[2], nv[3]), std::max(nv[4], nv[5])), nv[6]);   // :269
    const double maxNormDual = std::max(std::max(nv[8], nv[9]), std::max(nv[10], nv[11]));                  // :270
    o.updated = false; o.rho = rho;
    if (p.adptRho) {                                                                    // :277-286
        const double resRatio = (normResPrim * maxNormDual) / (normResDual * maxNormPrim);
        if ((resRatio > p.tau) || (1.0 / resRatio > p.tau)) {
            o.updated = true;
            const double t = rho * std::sqrt(std::sqrt(resRatio));
            o.rho = t > 1e5 ? 1e5 : (t < 1e-5 ? 1e-5 : t);
        }
    }
    o.converged = (normResPrim < p.epsAbs + p.epsRel * maxNormPrim) && (normResDual < p.epsAbs + p.epsRel * maxNormDual);   // :289-294
    o.resPrim = normResPrim; o.resDual = normResDual;
    return o;
}

}  // namespace pqrows
}  // namespace qps
