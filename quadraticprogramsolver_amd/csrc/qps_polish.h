// qps_polish.h -- polishing step (SolveQuadraticProgram.m:289-325) on device-resident problem data.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/qps.h"

namespace qps {

struct PolishReport {
    int flag = -1;               // minresFlag of the last MINRES call (:311): -1 never ran, 0 converged (x replaced), 1 not converged (x kept)
    int refinements = 0;         // executed bodies of the refinement loop :314
    int minresIterations = 0;    // total MINRES iterations
    int numLower = 0, numUpper = 0;   // sizes of the guessed active sets (:296-297)
    double relres = NAN;         // ||r|| / ||b|| of the last MINRES call
    double seconds = 0;
};

// x (device, padded NP) is replaced by the polished primal when the last MINRES call converged.  y: the multiplier of the
// ADMM loop (device).  part: slab scratch of at least max(gemv_cols_tiles(MP), apass slabs) * NP elements.
template <typename T>
void polish_dense(hipStream_t st, int64_t n, int64_t m, int NP, int MP, const T* P, const T* A, const T* q, const T* l, const T* u, const T* y,
                  T* x, T* part, const qps_params& p, PolishReport* rep);

}  // namespace qps
