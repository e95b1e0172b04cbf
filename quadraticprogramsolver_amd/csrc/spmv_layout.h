// spmv_layout.h -- host-side construction of the sparse layouts the SpMV kernels of k_sparse.hip read.  Plain C++ (no device code,
// no HIP headers): the same translation unit is built into libqps_hip.so and, with g++ (optionally -fsanitize=address,undefined), into the
// host-only test library of tests/capi/layout_shim.cpp, whose CPU tests expand every layout back into a matrix-vector product and
// compare it with scipy.
//
// Reference: the operator of LinOpCgInit / LinMapsCgInit (LinearSystemSolvers.jl:152-157, :195-200) is three products per application --
// mA * w, mA' * (.), mP * w -- on SparseMatrixCSC inputs; everything here is how those three matrices are laid out for the device.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace qps {
namespace layout {

struct Int4 { int x, y, z, w; };                 // layout-compatible with HIP's int4 (descriptors are uploaded as they are)

// ---- shared constants of the builders and the kernels -------------------------------------------------------------------------------
constexpr int STREAM_NNZ = 1024;                 // k_spmv_stream: non-zeros per workgroup (256 threads x 4)
constexpr int STREAM_ROWS = 256;                 // k_spmv_stream: rows per workgroup at most (eight passes of 32 rows)
constexpr int X_BLOCK_BYTES = 57344;             // column-blocked forms: 56 KiB of x per block in LDS (+ 16 KiB of products: two workgroups per CU)
constexpr int BCHUNK = 2048;                     // k_spmv_blk: non-zeros per task (512 threads x 4)
constexpr int BTHREADS = 512;
constexpr int BMAXT = 64;                        // k_spmv_blk: tasks per workgroup at most (their descriptors are staged in LDS)
constexpr int SIGMA = 2048;                      // k_spmv_sell: sorting window (rows)
constexpr int SLONG = 96;                        // k_spmv_sell: more entries than this in one block -> the row is summed by a wave of its own
constexpr int WMAX = 2304;                        // k_spmv_sell: rows of a sorting window at most (staged form: the window's row sums sit in LDS, 18 KiB in fp64, before they are stored)
template <typename T> constexpr int cb_of() { return X_BLOCK_BYTES / (int)sizeof(T); }   // columns per block
template <typename T> constexpr int sell_e() { return sizeof(T) == 8 ? 2 : 4; }   // entries per lane and unit: one 16-byte load of values

// ---- the caller's CSC arrays ------------------------------------------------------------------------------------------------------------
// What qps_create_csc checks before anything touches a device.  0 = fine; otherwise the qps_status to return (QPS_ERR_BAD_ARGUMENT = 1,
// QPS_ERR_BAD_DIMENSION = 2, QPS_ERR_NOT_FINITE = 3) and a message naming the matrix.
int validate_csc(int64_t nrows, int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, const char* name, std::string* msg);
// Sorted rows inside every column, duplicates summed, 0-based (what Julia's sparse() guarantees; C callers may not).
void canonical_csc(int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, std::vector<int64_t>& ocp, std::vector<int64_t>& ori,
                   std::vector<double>& onz);
// issymmetric(mP) with tolerance 0 (SolveQuadraticProgram.m:166-168) on a CSC matrix: -1 or the first offending column.
int64_t csc_asymmetry(int64_t n, const int64_t* cp, const int64_t* ri, const double* nz, int base);

struct CsrHost { int nrows = 0, ncols = 0; std::vector<int> rp, ci; std::vector<double> va; };
// A canonical CSC matrix (nrows x ncols) as two CSR matrices: `cols` = its transpose (the CSC arrays themselves, narrowed to int32) and `rows` =
// the matrix by rows (counting sort over the columns: column indices come out sorted inside every row).
void csc_to_csr_pair(int64_t nrows, int64_t ncols, const std::vector<int64_t>& cp, const std::vector<int64_t>& ri, const std::vector<double>& nz, CsrHost& rows, CsrHost& cols);
// The CSC arrays of a canonical matrix (nrows x ncols) read as the CSR of its transpose (ncols x nrows) -- for a symmetric P that is P itself.
CsrHost csc_as_transposed_csr(int64_t nrows, int64_t ncols, const std::vector<int64_t>& cp, const std::vector<int64_t>& ri, const std::vector<double>& nz);
// [top; bottom] stacked (same column count)
CsrHost stack_rows(const CsrHost& top, const CsrHost& bottom);
// k_spmv_stream: consecutive rows (at most STREAM_ROWS) holding <= STREAM_NNZ non-zeros per workgroup; a longer row stands alone.  Returns the row-block boundaries.
std::vector<int> stream_row_blocks(const CsrHost& M);

// ---- sliced form (k_spmv_sell) -------------------------------------------------------------------------------------------------------------
// Columns cut into blocks of CB; inside a block the rows are sorted by their length in the block within windows of SIGMA rows, cut into slices
// of 64 (one row per lane), a slice padded to its longest row and stored unit by unit (unit u = entries u*E .. u*E+E-1 of all 64 rows, lane after
// lane).  Padding entries carry column CB (the kernel keeps xs[CB] = 0) and value 0.  Rows with more than SLONG entries in a block are left
// out of the slices (`lr`, their entries in lci / lva).
// Sorting window: `win` rows (a multiple of 64; `nwin` windows per block, win / 64 slices each, the last one shorter).  Small launches keep win = SIGMA and hand
// slices out at slice granularity (a row sum is stored by the lane that owns the row: 8-byte stores scattered over the window).  When a workgroup's share of the rows
// reaches 1024, win is that share (<= WMAX) and -- if the windows cost about the same -- the layout is STAGED: a workgroup owns whole windows (wg_ptr boundaries are
// multiples of win / 64), sums the long rows of its windows itself, collects a window's row sums in LDS and stores them as one contiguous run.
template <typename T> struct SellLayout {
    int nrows = 0, ncols = 0, nblk = 0, nsl = 0, wpb = 0;    // nsl = slices per block, wpb = workgroups per block
    int win = SIGMA, nwin = 0, staged = 0;
    std::vector<int> sl_off;                                 // [nblk * nsl + 1] first unit of a slice
    std::vector<unsigned short> perm;                        // [nblk * nsl * 64] row of a lane relative to its window (0xffff: none)
    std::vector<unsigned short> cols; std::vector<T> vals;   // unit-major, 64 lanes x E entries per unit (+ one spare unit)
    std::vector<int> wg_ptr;                                 // [nblk * (wpb + 1)] slice range of every workgroup
    std::vector<int> lr_ptr; std::vector<Int4> lr;           // long rows per (block, window), [nblk * nwin + 1]: (row, first, end, 0) into lci / lva
    std::vector<unsigned short> lci; std::vector<T> lva;
    std::vector<int> src, lsrc;                              // (with_src) position in the CSR arrays of every slot of vals / lva, -1 for padding
    int64_t entries = 0, padded = 0, long_entries = 0;       // statistics: stored entries, slots incl. padding, entries of long rows
};
// false: not representable (too many units / slices) or not worth it (most entries sit in long rows) -> use the task form
// with_src: also record where every stored value came from, so that the values of a matrix whose pattern is fixed (the explicit reduced matrix of
// ItrSolCg, rebuilt on every rho switch) can be refreshed on the device without rebuilding the layout
template <typename T> bool build_sell(const CsrHost& M, int wgs, SellLayout<T>& out, bool with_src = false, bool allow_staged = true);

// ---- task form (k_spmv_blk) ------------------------------------------------------------------------------------------------------------------
// Per column block a CSR with 16-bit local column indices; a task = consecutive rows holding <= BCHUNK entries (<= 512 / 256 rows), every task
// starts at a multiple of four entries (zero entries in the gap); a row longer than BCHUNK is summed by a whole workgroup (`lr`).
template <typename T> struct TaskLayout {
    int nrows = 0, ncols = 0, nblk = 0, wpb = 0, per = 1, lpr4 = 0;
    std::vector<int> brp;                                    // [nblk * (nrows + 1)] entry offsets of the rows of a block
    std::vector<unsigned short> bci; std::vector<T> bva;     // blocks back to back (+ 64 spare entries)
    std::vector<int> task_ptr; std::vector<Int4> tasks;      // tasks per block: (row begin, row end, entry begin, entry end)
    std::vector<int> lr_ptr; std::vector<Int4> lr;           // long rows per block: (row, first, end, 0)
    std::vector<int> src;                                    // (with_src) position in the CSR arrays of every slot of bva, -1 for the gaps
};
template <typename T> void build_tasks(const CsrHost& M, int wgs, TaskLayout<T>& out, bool with_src = false);   // throws std::length_error beyond 2^31 entries

// ---- the explicit reduced matrix of ItrSolCgInit (LinearSystemSolvers.jl:112-114): mAA = mA' * mA, mPI = mP + sigma I, mL = mPI + rho * mAA -------------
// sum over the rows of A of (row length)^2: the multiply-adds of A'A and an upper bound of its non-zero count (O(rows) to evaluate)
int64_t ata_work(const CsrHost& Arows);
// Pattern of mL = pattern(P) U pattern(A'A) U diagonal as a CSR (symmetric: also its CSC) with sorted columns; on that ONE pattern the three value arrays
// the rebuild of :127-129 needs: vP (mP, 0 where absent), vAA (mA' * mA), diag (1 on the diagonal entries).  mL's values are vP + sigma diag + rho vAA.
// P: n x n CSR (= CSC), Arows: A by rows (m x n), Acols: A by columns (n x m).  max_nnz: give up (return false, L untouched) beyond this many entries.
bool reduced_matrix(const CsrHost& P, const CsrHost& Arows, const CsrHost& Acols, int64_t max_nnz, CsrHost& L, std::vector<double>& vAA, std::vector<double>& diag);

// ---- host interpreters: y = M x by walking the arrays the way the kernels do (partial sums per block, then the blocks added in order) ----
template <typename T> void apply_sell(const SellLayout<T>& L, const T* x, T* y);
template <typename T> void apply_tasks(const TaskLayout<T>& L, const T* x, T* y);

}  // namespace layout
}  // namespace qps
