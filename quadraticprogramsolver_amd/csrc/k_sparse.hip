// k_sparse.hip -- CSR SpMV + matrix-free CG path (placeholder until the kernels land in this round).
#include "qps_internal.h"
namespace qps {
SolverBase* make_sparse_solver(int, int64_t, int64_t, int, const int64_t*, const int64_t*, const double*, const int64_t*,
                               const int64_t*, const double*, const double*, const double*, const double*, int) {
    throw QpsError(QPS_ERR_UNSUPPORTED, "CSR/CG path not built yet");
}
}  // namespace qps
