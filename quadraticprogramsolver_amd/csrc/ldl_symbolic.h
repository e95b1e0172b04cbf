// ldl_symbolic.h -- host-side symbolic analysis for the sparse direct KKT plugin (internal; the boundary is include/qps.h).
//
// Reference: the three direct plugins LaLdlInit / QDLdlInit / FacLdlInit (LinearSystemSolvers.jl:16-24, :47-55, :78-86) factorise
//     K = [mP + sigma I   mA' ;  mA   -rho^-1 I]                                                  (:18, :49, :81)
// with SuiteSparse ldlt / QDLDL.qdldl / LDLFactorizations.ldl -- all three: fill-reducing (approximate minimum degree) ordering,
// elimination tree, symbolic factor, numeric L D L'.  Only the (2,2) block depends on rho, so everything in this file is done
// once per handle and a changedRho re-factorisation (:30-32, :61-63, :93-95) is numeric only.
//
// K is symmetric quasi-definite: every symmetric permutation has an L D L' factorisation without pivoting, and the sign of a
// pivot is known beforehand (+ for the n variables, - for the m constraint rows).
//
// Layout produced here (device-friendly):
//   * columns are reordered by elimination-tree level (leaves first), which is an equivalent reordering (same fill);
//     the columns of one level are mutually independent, both in the numeric factorisation and in the triangular solves;
//   * the wide levels form the SPARSE part (columns [0, Ns)): scalar CSR (for the forward gather) and CSC (backward gather) of
//     the strictly lower triangle of L, restricted to those columns;
//   * the remaining Nt = N - Ns columns (narrow levels near the root: long dependency chains, dense fill) form the TAIL, kept as
//     one dense Nt x Nt matrix and factorised / solved by the dense kernels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace qps {

struct LdlSymbolic {
    int n = 0, m = 0, N = 0;
    int Ns = 0, Nt = 0;                       // sparse columns, tail size (Ns + Nt == N)
    std::vector<int> perm, iperm;             // perm[new] = old index in [x; nu] numbering, iperm[old] = new
    std::vector<int> level_ptr;               // sparse levels: columns [level_ptr[l], level_ptr[l + 1]); level_ptr.back() == Ns
    std::vector<signed char> sign;            // per new index: +1 variable (pivot > 0), -1 constraint row (pivot < 0)
    // strictly lower triangle of L restricted to columns < Ns
    std::vector<int> rp, ci;                  // CSR over the N rows (sorted columns)
    std::vector<int> cp, ri;                  // CSC over the Ns columns (sorted rows, rows up to N - 1)
    std::vector<int> csr2csc;                 // position of CSR entry k in the CSC arrays
    // the entries of K (lower triangle, new numbering) and where they start the numeric factorisation:
    //   dst >= 0          : CSR position of an L entry (column < Ns)
    //   dst <  0          : -(1 + (i - Ns) * ldt + (j - Ns)) position in the dense tail (row-major, leading dimension ldt)
    // src indexes the value table built by the caller: P lower entries first (in CSC order of the input), then the entries of A
    std::vector<int64_t> k_dst; std::vector<int> k_src;
    std::vector<int> dpos_P;                  // per new index: position of P_ii in the value table or -1 (structural zero); constraint rows -1
    int ldt = 0;                              // leading dimension of the dense tail (Nt rounded up to 64)
    int64_t nnzL = 0;                         // entries of the strictly lower triangle of L, tail counted as dense
    int64_t nnzL_exact = 0;                   // structural non-zeros of L under this ordering (before the tail is densified)
    int64_t nnzK = 0;                         // entries of the strictly lower triangle of K
    int levels_total = 0;                     // height of the elimination tree
};

// Symmetric pattern as adjacency lists without self loops -> elimination order (approximate minimum degree, Amestoy / Davis / Duff).
std::vector<int> amd_order(int N, const std::vector<std::vector<int>>& adj);

// Fallback for chain-like graphs whose minimum-degree elimination tree is as deep as the matrix is long: recursive dissection of the breadth-first
// line order (empty result: the line order is too wide for it to help).
std::vector<int> line_dissection_order(int N, const std::vector<std::vector<int>>& adj);

// P, A: CSC with 0-based 32-bit-safe indices (P full symmetric storage; only its lower triangle is read).
// max_tail: upper bound for Nt; min_level_width: a level narrower than this goes to the tail when the tail has room.
// Throws std::runtime_error when the factor would not fit (nnz(L) beyond 2^31 - 1 or the level count beyond max_levels).
LdlSymbolic ldl_analyze(int n, int m, const int64_t* Pcp, const int64_t* Pri, const int64_t* Acp, const int64_t* Ari, int index_base,
                        int max_tail, int min_level_width, int max_levels);

}  // namespace qps
