// qps_proxqp.h -- internal interface of the ProxQP.jl form (device solver in qps_proxqp.hip).
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/qps.h"

namespace qps {

struct ProxQpBase {
    int device = 0; int64_t n = 0, me = 0, mi = 0; std::string err;
    virtual ~ProxQpBase() {}
    virtual void init_kkt() = 0;                                                           // ProxQP.jl:73-93
    virtual void set_state(const double* x, const double* y, const double* z, const double* s) = 0;   // inner constructor :36
    virtual void get_state(double* x, double* y, double* z, double* s) = 0;
    virtual void solve(const qps_proxqp_params& p, qps_proxqp_report* rep) = 0;            // ProxQP.jl:118-173
};

ProxQpBase* make_proxqp(int device, int64_t n, int64_t me, int64_t mi, int dtype, const double* P, int64_t ldp, const double* A, int64_t lda,
                        const double* b, const double* C, int64_t ldc, const double* d, const double* q);

// SparseProxQP (ProxQP.jl:71, :95-115): CSC inputs kept sparse, the linear system solved in its KKT form by the sparse L D L' plugin (k_sparse.hip)
ProxQpBase* make_proxqp_sparse(int device, int64_t n, int64_t me, int64_t mi, int dtype, const int64_t* Pcp, const int64_t* Pri, const double* Pnz, const double* q,
                               const int64_t* Acp, const int64_t* Ari, const double* Anz, const double* b, const int64_t* Ccp, const int64_t* Cri, const double* Cnz,
                               const double* d, int index_base);

}  // namespace qps
