// qps_ldl.h -- device-side sparse direct KKT plugin (internal).  Counterpart of the reference's direct plugin pairs
// LaLdlInit/LaLdl!, QDLdlInit/QDLdl!, FacLdlInit/FacLdl! (LinearSystemSolvers.jl:16-107).
#pragma once
#include <cstdint>
#include <memory>

#include "ldl_symbolic.h"
#include "qps_internal.h"

namespace qps {

template <typename T> struct SparseLdl {
    virtual ~SparseLdl() {}
    // numeric L D L' of K(rho, sigma) on the device; symbolic data is reused (the changedRho branch :30-32 / :61-63 / :93-95)
    virtual void factorize(double rho, double sigma) = 0;
    // LinSysSol! body (:37-40): rhs [sigma x - q; z - y / rho], solve in place, xx = x~, zz = z + (nu - y) / rho
    virtual void solve(const T* x, const T* q, const T* z, const T* y, double rho, double sigma, T* xx, T* zz) = 0;
    // One whole ADMM iteration around the solve (SolveQuadraticProgram.jl:54-61): the sweeps, then ONE launch that un-permutes,
    // forms z~ (:40), applies the x / z / y updates (:56-61) and writes the NEXT iteration's permuted right-hand side (:37-38) --
    // two launches fewer per iteration than solve() + admm_update.  rhs_ready: the previous call of iterate() with the same
    // (rho, sigma) already left the right-hand side in place.
    virtual void iterate(T* x, T* xp, const T* q, T* z, T* zp, T* y, const T* l, const T* u, double alpha, double rho, double sigma, bool rhs_ready) = 0;
    // K(rho, sigma) [out_x; out_nu] = [r1; r2] with the factor of the last factorize(): the multiplier block comes back as it is
    // (solve() folds it into z~ = z + (nu - y) / rho, which loses nu when rho is huge)
    virtual void solve_raw(const T* r1, const T* r2, T* out_x, T* out_nu) = 0;
    virtual const LdlSymbolic& symbolic() const = 0;
    virtual int launches_per_solve() const = 0;
    virtual double bytes_per_solve() const = 0;
};

// Pvals / Avals: the caller's CSC value arrays (P full symmetric storage), same order as the index arrays given to ldl_analyze
template <typename T>
std::unique_ptr<SparseLdl<T>> make_sparse_ldl(hipStream_t st, LdlSymbolic&& sym, const double* Pvals, int64_t pnnz, const double* Avals, int64_t annz);

// Dense signed Cholesky M = Lt J Lt' (J = diag(sgn), sgn = +-1 known beforehand: quasi-definite matrices) of an NP x NP row-major
// matrix (lower), in place; dinv receives the inverses of the 64 x 64 diagonal blocks of Lt, W (NP x NP scratch) the panels Lt J.
template <typename T> void cholesky_signed(hipStream_t st, int NP, T* M, T* dinv, int* fail_dev, const T* sgn, T* W);

}  // namespace qps
