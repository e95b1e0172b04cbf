// qps_polish.h -- polishing step (SolveQuadraticProgram.m:289-325) on device-resident problem data.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <functional>

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

// out = K v + delta blkdiag(I, -I) v on vectors [x block (NP) | multiplier block (MP)], multiplier block masked (:304-305);
// `scratch` holds at least MP elements
template <typename T> using PolishProduct = std::function<void(const T* v, T delta, T* out, const T* mask, T* scratch)>;
// the refinement loop :307-325 around a caller-supplied product (dense handles: GEMV + masked pass; CSR handles: SpMVs)
template <typename T>
void polish_with(hipStream_t st, int64_t n, int64_t m, int NP, int MP, const T* q, const T* l, const T* u, const T* y, T* x, const qps_params& p,
                 PolishReport* rep, const PolishProduct<T>& kmat);
// out_lam = mask out_lam - delta mask vlam (in place), wl = mask vlam
template <typename T> void polish_mask_rows(hipStream_t st, int MP, const T* mask, const T* vlam, T delta, T* out_lam, T* wl);

// x (device, padded NP) is replaced by the polished primal when the last MINRES call converged.  y: the multiplier of the
// ADMM loop (device).  part: slab scratch of at least max(gemv_cols_tiles(MP), apass slabs) * NP elements.
template <typename T>
void polish_dense(hipStream_t st, int64_t n, int64_t m, int NP, int MP, const T* P, const T* A, const T* q, const T* l, const T* u, const T* y,
                  T* x, T* part, const qps_params& p, PolishReport* rep);

}  // namespace qps
