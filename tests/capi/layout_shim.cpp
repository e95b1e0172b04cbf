// layout_shim.cpp -- TEST INFRASTRUCTURE: a C ABI over the host-only code of the product (spmv_layout.cpp, ldl_symbolic.cpp) so that CPU tests can
// expand every SpMV layout back into a matrix-vector product and run the symbolic analysis without a GPU.  Built with g++ by
// `make -C quadraticprogramsolver_amd/csrc host-test` (SAN=1: -fsanitize=address,undefined) into libqps_host_test[_san].so; never part of libqps_hip.so.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../quadraticprogramsolver_amd/csrc/batch_schedule.h"
#include "../../quadraticprogramsolver_amd/csrc/ldl_symbolic.h"
#include "../../quadraticprogramsolver_amd/csrc/spmv_layout.h"
#include "../../quadraticprogramsolver_amd/csrc/tile_order.h"

using namespace qps;
using namespace qps::layout;

namespace {
CsrHost make_csr(int nrows, int ncols, const int* rp, const int* ci, const double* va) {
    CsrHost H; H.nrows = nrows; H.ncols = ncols;
    H.rp.assign(rp, rp + nrows + 1); H.ci.assign(ci, ci + rp[nrows]); H.va.assign(va, va + rp[nrows]);
    return H;
}
template <typename T> int sell_apply_t(const CsrHost& H, int wgs, const double* x, double* y, int64_t* stats) {
    SellLayout<T> L;
    if (!build_sell<T>(H, wgs, L, true)) return 0;
    // the source map (value refresh of a frozen pattern): every stored value is the CSR value it points at, padding points nowhere
    if (L.src.size() != L.vals.size() || L.lsrc.size() != L.lva.size()) return -2;
    for (size_t i = 0; i < L.vals.size(); ++i) if (L.src[i] >= 0 ? L.vals[i] != (T)H.va[(size_t)L.src[i]] : L.vals[i] != T(0)) return -2;
    for (size_t i = 0; i < L.lva.size(); ++i) if (L.lsrc[i] >= 0 ? L.lva[i] != (T)H.va[(size_t)L.lsrc[i]] : L.lva[i] != T(0)) return -2;
    std::vector<T> xt(x, x + H.ncols), yt((size_t)H.nrows);
    apply_sell<T>(L, xt.data(), yt.data());
    for (int r = 0; r < H.nrows; ++r) y[r] = (double)yt[(size_t)r];
    if (stats) { stats[0] = L.entries; stats[1] = L.padded; stats[2] = L.long_entries; stats[3] = (int64_t)L.lr.size(); stats[4] = L.nblk; stats[5] = L.nsl; stats[6] = L.wpb; stats[7] = (int64_t)L.staged * 100000 + L.win; }
    return 1;
}
template <typename T> int tasks_apply_t(const CsrHost& H, int wgs, const double* x, double* y, int64_t* stats) {
    TaskLayout<T> L;
    build_tasks<T>(H, wgs, L, true);
    if (L.src.size() != L.bva.size()) return -2;
    for (size_t i = 0; i < L.bva.size(); ++i) if (L.src[i] >= 0 ? L.bva[i] != (T)H.va[(size_t)L.src[i]] : L.bva[i] != T(0)) return -2;
    std::vector<T> xt(x, x + H.ncols), yt((size_t)H.nrows);
    apply_tasks<T>(L, xt.data(), yt.data());
    for (int r = 0; r < H.nrows; ++r) y[r] = (double)yt[(size_t)r];
    if (stats) { stats[0] = (int64_t)L.tasks.size(); stats[1] = (int64_t)L.bci.size(); stats[2] = 0; stats[3] = (int64_t)L.lr.size(); stats[4] = L.nblk; stats[5] = L.per; stats[6] = L.wpb; }
    return 1;
}
}  // namespace

extern "C" {
#define SHIM_API __attribute__((visibility("default")))

// y = M x through the sliced layout (dtype 0 = fp64, 1 = fp32).  1 = built and applied, 0 = the builder declined (task form instead), -1 = exception
SHIM_API int lt_sell_apply(int dtype, int nrows, int ncols, const int* rp, const int* ci, const double* va, int wgs, const double* x, double* y, int64_t* stats) {
    try { const CsrHost H = make_csr(nrows, ncols, rp, ci, va); return dtype == 0 ? sell_apply_t<double>(H, wgs, x, y, stats) : sell_apply_t<float>(H, wgs, x, y, stats); }
    catch (const std::exception&) { return -1; }
}
SHIM_API int lt_tasks_apply(int dtype, int nrows, int ncols, const int* rp, const int* ci, const double* va, int wgs, const double* x, double* y, int64_t* stats) {
    try { const CsrHost H = make_csr(nrows, ncols, rp, ci, va); return dtype == 0 ? tasks_apply_t<double>(H, wgs, x, y, stats) : tasks_apply_t<float>(H, wgs, x, y, stats); }
    catch (const std::exception&) { return -1; }
}
// the row blocks of the CSR-stream kernel: count written to *nblocks, boundaries to rb (capacity nrows + 2)
SHIM_API int lt_stream_blocks(int nrows, int ncols, const int* rp, const int* ci, const double* va, int* rb, int* nblocks) {
    const CsrHost H = make_csr(nrows, ncols, rp, ci, va);
    const std::vector<int> v = stream_row_blocks(H);
    std::memcpy(rb, v.data(), sizeof(int) * v.size());
    *nblocks = (int)v.size() - 1;
    return 0;
}
SHIM_API int lt_validate_csc(int64_t nrows, int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, char* msg, int cap) {
    std::string why;
    const int rc = validate_csc(nrows, ncols, cp, ri, nz, base, "M", &why);
    if (msg && cap > 0) { std::strncpy(msg, why.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; }
    return rc;
}
SHIM_API int64_t lt_csc_asymmetry(int64_t n, const int64_t* cp, const int64_t* ri, const double* nz, int base) { return csc_asymmetry(n, cp, ri, nz, base); }
// caller's CSC (any order, duplicates, base 0 / 1) -> canonical CSC -> CSR of the matrix (rows) and of its transpose (cols).  Output arrays sized by the caller:
// rp[nrows + 1], ci / va [nnz], trp[ncols + 1], tci / tva [nnz]; returns the canonical non-zero count (<= the input's)
SHIM_API int64_t lt_csc_to_csr(int64_t nrows, int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, int* rp, int* ci, double* va, int* trp, int* tci,
                               double* tva) {
    std::vector<int64_t> ocp, ori; std::vector<double> onz;
    canonical_csc(ncols, cp, ri, nz, base, ocp, ori, onz);
    CsrHost rows, cols;
    csc_to_csr_pair(nrows, ncols, ocp, ori, onz, rows, cols);
    std::memcpy(rp, rows.rp.data(), sizeof(int) * rows.rp.size());
    if (!rows.ci.empty()) { std::memcpy(ci, rows.ci.data(), sizeof(int) * rows.ci.size()); std::memcpy(va, rows.va.data(), sizeof(double) * rows.va.size()); }
    std::memcpy(trp, cols.rp.data(), sizeof(int) * cols.rp.size());
    if (!cols.ci.empty()) { std::memcpy(tci, cols.ci.data(), sizeof(int) * cols.ci.size()); std::memcpy(tva, cols.va.data(), sizeof(double) * cols.va.size()); }
    const CsrHost st = stack_rows(rows, rows);               // exercised for its own sake: [M; M]
    if (st.nrows != 2 * rows.nrows || st.ci.size() != 2 * rows.ci.size()) return -1;
    return (int64_t)ori.size();
}
// ItrSolCgInit's matrices (LinearSystemSolvers.jl:112-114) from canonical CSC inputs: pattern of mL = mPI + rho mAA as a CSR + the three value arrays on it.
// Output arrays sized by the caller (cap entries); returns nnz(mL), -1 when it exceeds cap, -2 on an exception.  *work = sum of squared row lengths of A.
SHIM_API int64_t lt_reduced_matrix(int64_t n, int64_t m, const int64_t* Pcp, const int64_t* Pri, const double* Pnz, const int64_t* Acp, const int64_t* Ari, const double* Anz,
                                   int64_t cap, int* rp, int* ci, double* vP, double* vAA, double* dg, int64_t* work) {
    try {
        std::vector<int64_t> pcp, pri, acp, ari; std::vector<double> pnz, anz;
        canonical_csc(n, Pcp, Pri, Pnz, 0, pcp, pri, pnz); canonical_csc(n, Acp, Ari, Anz, 0, acp, ari, anz);
        const CsrHost Ph = csc_as_transposed_csr(n, n, pcp, pri, pnz);
        CsrHost Ah, Ath; csc_to_csr_pair(m, n, acp, ari, anz, Ah, Ath);
        if (work) *work = ata_work(Ah);
        CsrHost L; std::vector<double> aa, d;
        if (!reduced_matrix(Ph, Ah, Ath, cap, L, aa, d)) return -1;
        std::memcpy(rp, L.rp.data(), sizeof(int) * L.rp.size());
        if (!L.ci.empty()) {
            std::memcpy(ci, L.ci.data(), sizeof(int) * L.ci.size()); std::memcpy(vP, L.va.data(), sizeof(double) * L.va.size());
            std::memcpy(vAA, aa.data(), sizeof(double) * aa.size()); std::memcpy(dg, d.data(), sizeof(double) * d.size());
        }
        return (int64_t)L.ci.size();
    } catch (const std::exception&) { return -2; }
}
// The hand-out of qps_solve_batch_multi (batch_schedule.h) with a stand-in for the solve: "solving" QP b takes cost_us[b] microseconds of wall time (sleep), a range
// takes the sum of its QPs.  worker_of[count] / solved[count] (how often each QP was solved) / worker_seconds[workers] out.  fail_at >= 0: the range holding that QP
// reports an error (the hand-out must stop).  Returns what run_batch_workers returns.
SHIM_API int lt_schedule(int64_t count, int workers, int64_t chunk, const double* cost_us, int* worker_of, int* solved, double* worker_seconds, int64_t fail_at) {
    auto solve = [&](int w, int64_t b0, int64_t cnt) -> int {
        double us = 0;
        for (int64_t b = b0; b < b0 + cnt; ++b) { us += cost_us[b]; worker_of[b] = w; __atomic_fetch_add(&solved[b], 1, __ATOMIC_RELAXED); }
        std::this_thread::sleep_for(std::chrono::duration<double, std::micro>(us));
        return (fail_at >= b0 && fail_at < b0 + cnt) ? 7 : 0;
    };
    return run_batch_workers(count, workers, chunk, 65535, solve, worker_seconds);
}
// the symbolic analysis of the sparse direct KKT plugin (what qps_ldl_analyze exports from the product library): perm[new] = old, report[8]
SHIM_API int lt_ldl_analyze(int n, int m, const int64_t* Pcp, const int64_t* Pri, const int64_t* Acp, const int64_t* Ari, int base, int max_tail, int min_level, int max_levels,
                            int64_t* perm_out, int64_t* report) {
    try {
        const LdlSymbolic s = ldl_analyze(n, m, Pcp, Pri, Acp, Ari, base, max_tail, min_level, max_levels);
        if (perm_out) for (int k = 0; k < n + m; ++k) perm_out[k] = s.perm[(size_t)k];
        if (report) { report[0] = s.N; report[1] = s.Ns; report[2] = s.Nt; report[3] = (int64_t)s.level_ptr.size() - 1; report[4] = s.levels_total; report[5] = s.nnzK; report[6] = s.nnzL_exact; report[7] = s.nnzL; }
        return 0;
    } catch (const std::exception&) { return -1; }
}
// The id -> tile maps of tile_order.h walked over a whole launch: tiles[(bi * nt + bj)] counts how often tile (bi, bj) was dealt.  kind 0: k_gemm's lower tiles of an
// nt x nt grid (returns the number of ids of the launch, ids_per_xcd_span[x] = number of distinct tile ROWS + COLUMNS XCD x touches in its first 64 tiles);
// kind 1 / 2: k_chol_update_diag without / with the XCD rule (id 0, the diagonal workgroup, takes tile (0, 0) itself and is not dealt).
SHIM_API int lt_tile_order(int kind, int nt, int* tiles, int* panels_first64) {
    const int nids = kind == 0 ? lower_tile_ids(nt) : chol_update_ids(nt, kind == 2);
    std::vector<std::vector<char>> rows(8, std::vector<char>((size_t)nt, 0)), cols(8, std::vector<char>((size_t)nt, 0));
    std::vector<int> dealt(8, 0);
    for (int id = 0; id < nids; ++id) {
        int bi = -1, bj = -1;
        const bool got = kind == 0 ? lower_tile_of(id, nids, nt, bi, bj) : chol_update_tile_of(id, kind == 2, nt, bi, bj);
        if (!got) continue;
        if (bi < 0 || bi >= nt || bj < 0 || bj > bi) return -1;
        if (kind == 2 && (id & 7) == 0) return -2;                                 // an id on the diagonal workgroup's XCD was given a tile
        ++tiles[(size_t)bi * nt + bj];
        const int x = id & 7;
        if (dealt[x]++ < 64) { rows[x][(size_t)bi] = 1; cols[x][(size_t)bj] = 1; }
    }
    if (panels_first64) for (int x = 0; x < 8; ++x) { int c = 0; for (int k = 0; k < nt; ++k) c += rows[x][(size_t)k] + cols[x][(size_t)k]; panels_first64[x] = c; }
    return nids;
}
}
