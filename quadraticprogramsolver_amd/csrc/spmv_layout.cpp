// spmv_layout.cpp -- see spmv_layout.h.  Host code only.
#include "spmv_layout.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <utility>

namespace qps {
namespace layout {

// ---------------------------------------------------------------------------------------------------------------------------------------
// the caller's CSC arrays
// ---------------------------------------------------------------------------------------------------------------------------------------
int validate_csc(int64_t nrows, int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, const char* name, std::string* msg) {
    auto fail = [&](int code, const char* what) {
        if (msg) { *msg = std::string(name) + " " + what; }
        return code;
    };
    if (!cp) return fail(1, "colptr is NULL");
    if (ncols < 0 || nrows < 0) return fail(2, "has a negative dimension");
    if (cp[0] != base) return fail(1, "colptr does not start at index_base");
    for (int64_t j = 0; j < ncols; ++j) if (cp[j + 1] < cp[j]) return fail(1, "colptr not monotone");
    const int64_t nnz = cp[ncols] - base;
    if (nnz < 0) return fail(1, "colptr does not start at index_base");
    if (nnz > 0 && (!ri || !nz)) return fail(1, "rowval / nzval is NULL");
    for (int64_t k = 0; k < nnz; ++k) if (ri[k] - base < 0 || ri[k] - base >= nrows) return fail(2, "row index out of range");
    for (int64_t k = 0; k < nnz; ++k) if (!std::isfinite(nz[k])) return fail(3, "contains NaN/Inf");
    return 0;
}

void canonical_csc(int64_t ncols, const int64_t* cp, const int64_t* ri, const double* nz, int base, std::vector<int64_t>& ocp, std::vector<int64_t>& ori,
                   std::vector<double>& onz) {
    ocp.assign((size_t)ncols + 1, 0); ori.clear(); onz.clear();
    const int64_t nnz = cp[ncols] - base;
    ori.reserve((size_t)nnz); onz.reserve((size_t)nnz);
    std::vector<std::pair<int64_t, double>> col;
    auto less = [](const std::pair<int64_t, double>& a, const std::pair<int64_t, double>& b) { return a.first < b.first; };
    for (int64_t j = 0; j < ncols; ++j) {
        col.clear();
        for (int64_t k = cp[j] - base; k < cp[j + 1] - base; ++k) col.emplace_back(ri[k] - base, nz[k]);
        if (!std::is_sorted(col.begin(), col.end(), less)) std::stable_sort(col.begin(), col.end(), less);
        for (size_t k = 0; k < col.size(); ++k) {
            if ((int64_t)ori.size() > ocp[j] && ori.back() == col[k].first) onz.back() += col[k].second;
            else { ori.push_back(col[k].first); onz.push_back(col[k].second); }
        }
        ocp[j + 1] = (int64_t)ori.size();
    }
}

int64_t csc_asymmetry(int64_t n, const int64_t* cp, const int64_t* ri, const double* nz, int base) {
    std::vector<int64_t> scp, sri; std::vector<double> snz;
    canonical_csc(n, cp, ri, nz, base, scp, sri, snz);
    const int64_t snnz = (int64_t)sri.size();
    // transpose by a counting sort
    std::vector<int64_t> tcp((size_t)n + 1, 0), tri((size_t)snnz); std::vector<double> tnz((size_t)snnz);
    for (int64_t k = 0; k < snnz; ++k) tcp[(size_t)sri[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) tcp[i + 1] += tcp[i];
    {
        std::vector<int64_t> pos(tcp.begin(), tcp.end() - 1);
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = scp[j]; k < scp[j + 1]; ++k) { const int64_t r = sri[k]; tri[pos[r]] = j; tnz[pos[r]] = snz[k]; pos[r]++; }
    }
    // explicit zeros on one side only are still symmetric values: compare through a merge that treats a missing entry as 0
    for (int64_t j = 0; j < n; ++j) {
        int64_t a = scp[j], b = tcp[j];
        while (a < scp[j + 1] || b < tcp[j + 1]) {
            const int64_t ra = a < scp[j + 1] ? sri[a] : n, rb = b < tcp[j + 1] ? tri[b] : n;
            if (ra == rb) { if (snz[a] != tnz[b]) return j; ++a; ++b; }
            else if (ra < rb) { if (snz[a] != 0.0) return j; ++a; }
            else { if (tnz[b] != 0.0) return j; ++b; }
        }
    }
    return -1;
}

void csc_to_csr_pair(int64_t nrows, int64_t ncols, const std::vector<int64_t>& cp, const std::vector<int64_t>& ri, const std::vector<double>& nz, CsrHost& rows, CsrHost& cols) {
    const int64_t nnz = cp[(size_t)ncols];
    if (nnz > 2000000000LL || nrows > 2000000000LL || ncols > 2000000000LL) throw std::length_error("more than 2^31 non-zeros / rows / columns");
    cols.nrows = (int)ncols; cols.ncols = (int)nrows;
    cols.rp.resize((size_t)ncols + 1); cols.ci.resize((size_t)nnz); cols.va.assign(nz.begin(), nz.begin() + nnz);
    for (int64_t j = 0; j <= ncols; ++j) cols.rp[(size_t)j] = (int)cp[(size_t)j];
    rows.nrows = (int)nrows; rows.ncols = (int)ncols;
    rows.rp.assign((size_t)nrows + 1, 0); rows.ci.resize((size_t)nnz); rows.va.resize((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) { cols.ci[(size_t)k] = (int)ri[(size_t)k]; rows.rp[(size_t)ri[(size_t)k] + 1]++; }
    for (int64_t i = 0; i < nrows; ++i) rows.rp[(size_t)i + 1] += rows.rp[(size_t)i];
    std::vector<int> pos(rows.rp.begin(), rows.rp.end() - 1);
    for (int64_t j = 0; j < ncols; ++j)          // columns visited in order -> sorted column indices inside every row
        for (int64_t k = cp[(size_t)j]; k < cp[(size_t)j + 1]; ++k) {
            const int r = (int)ri[(size_t)k];
            rows.ci[(size_t)pos[(size_t)r]] = (int)j; rows.va[(size_t)pos[(size_t)r]] = nz[(size_t)k]; pos[(size_t)r]++;
        }
}

CsrHost csc_as_transposed_csr(int64_t nrows, int64_t ncols, const std::vector<int64_t>& cp, const std::vector<int64_t>& ri, const std::vector<double>& nz) {
    const int64_t nnz = cp[(size_t)ncols];
    if (nnz > 2000000000LL || nrows > 2000000000LL || ncols > 2000000000LL) throw std::length_error("more than 2^31 non-zeros / rows / columns");
    CsrHost C; C.nrows = (int)ncols; C.ncols = (int)nrows;
    C.rp.resize((size_t)ncols + 1); C.ci.resize((size_t)nnz); C.va.assign(nz.begin(), nz.begin() + nnz);
    for (int64_t j = 0; j <= ncols; ++j) C.rp[(size_t)j] = (int)cp[(size_t)j];
    for (int64_t k = 0; k < nnz; ++k) C.ci[(size_t)k] = (int)ri[(size_t)k];
    return C;
}

CsrHost stack_rows(const CsrHost& top, const CsrHost& bottom) {
    if (top.ncols != bottom.ncols) throw std::invalid_argument("stack_rows: column counts differ");
    const int64_t tn = (int64_t)top.ci.size(), bn = (int64_t)bottom.ci.size();
    if (tn + bn > 2000000000LL || (int64_t)top.nrows + bottom.nrows > 2000000000LL) throw std::length_error("more than 2^31 entries in the stacked matrix");
    CsrHost S; S.nrows = top.nrows + bottom.nrows; S.ncols = top.ncols;
    S.rp.resize((size_t)S.nrows + 1);
    for (int i = 0; i <= top.nrows; ++i) S.rp[(size_t)i] = top.rp[(size_t)i];
    for (int i = 1; i <= bottom.nrows; ++i) S.rp[(size_t)top.nrows + i] = (int)tn + bottom.rp[(size_t)i];
    S.ci.reserve((size_t)(tn + bn)); S.va.reserve((size_t)(tn + bn));
    S.ci.insert(S.ci.end(), top.ci.begin(), top.ci.end()); S.ci.insert(S.ci.end(), bottom.ci.begin(), bottom.ci.end());
    S.va.insert(S.va.end(), top.va.begin(), top.va.end()); S.va.insert(S.va.end(), bottom.va.begin(), bottom.va.end());
    return S;
}

std::vector<int> stream_row_blocks(const CsrHost& M) {
    std::vector<int> rb(1, 0);
    int start = 0;
    for (int r = 0; r < M.nrows; ++r) {
        // ... and at most STREAM_ROWS rows: the workgroup walks its rows 32 at a time, and a block of thousands of EMPTY rows (the zero block of a lasso / Huber P:
        // 20 100 of 30 100 rows) kept one workgroup looping for ~200 us per product
        if (r - start >= STREAM_ROWS) { rb.push_back(r); start = r; }
        if (M.rp[(size_t)r + 1] - M.rp[(size_t)start] > STREAM_NNZ && r > start) { rb.push_back(r); start = r; }
        if (M.rp[(size_t)r + 1] - M.rp[(size_t)start] > STREAM_NNZ) { rb.push_back(r + 1); start = r + 1; }   // single long row
    }
    if (start < M.nrows) rb.push_back(M.nrows);
    return rb;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// sliced form
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct RowRun { int len, row, first; };           // a row's entries inside one column block: [first, first + len) of the CSR arrays
}

template <typename T> bool build_sell(const CsrHost& M, int wgs, SellLayout<T>& out, bool with_src, bool allow_staged) {
    constexpr int CB = cb_of<T>(), E = sell_e<T>();
    const int nrows = M.nrows, ncols = M.ncols;
    const int64_t nnz = (int64_t)M.ci.size();
    if (nrows <= 0 || ncols <= 0) return false;
    const int nblk = (ncols + CB - 1) / CB;
    // the sorting window: SIGMA rows, or a workgroup's share of the rows once that share is large enough to be worth staging in LDS (see SellLayout)
    const int target = std::max(1, std::max(1, wgs) / nblk);         // workgroups per block of a launch that is resident at once
    const int share = (nrows + target - 1) / target;
    const bool want_staged = allow_staged && share >= 1024;
    const int per_wg = want_staged ? (share + WMAX - 1) / WMAX : 1;   // windows per workgroup when its share exceeds the largest window: equal windows, not one full + a rest
    const int win = want_staged ? std::min(WMAX, ((share + per_wg - 1) / per_wg + 63) / 64 * 64) : SIGMA, spw = win / 64;
    const int nwin = (nrows + win - 1) / win;
    const int nsl = (nwin - 1) * spw + ((nrows - (nwin - 1) * win) + 63) / 64;                 // slices per block (only the last window is short)
    if ((int64_t)nblk * nsl * 64 > 2000000000LL) return false;
    out = SellLayout<T>();
    out.nrows = nrows; out.ncols = ncols; out.nblk = nblk; out.nsl = nsl; out.win = win; out.nwin = nwin;
    out.sl_off.assign((size_t)nblk * nsl + 1, 0);
    out.perm.assign((size_t)nblk * nsl * 64, (unsigned short)0xffff);
    out.lr_ptr.assign((size_t)nblk * nwin + 1, 0);
    std::vector<int64_t> wcost((size_t)nblk * nwin, 0);              // per (block, window): units + a fixed part per slice + the wave steps of its long rows
    out.cols.reserve((size_t)(nnz + nnz / 8)); out.vals.reserve((size_t)(nnz + nnz / 8));
    // the blocks are visited in order and the column indices of a row are sorted, so the entries of (block b, row r) are the run that starts
    // where the run of (b - 1, r) ended: one cursor per row instead of a table per block
    std::vector<int> cur(M.rp.begin(), M.rp.begin() + nrows);
    std::vector<RowRun> ord;
    int64_t units = 0;
    for (int b = 0; b < nblk; ++b) {
        const int64_t cend = std::min<int64_t>((int64_t)(b + 1) * CB, ncols);
        for (int w = 0; w < nwin; ++w) {
            const int r0 = w * win, r1 = std::min(nrows, r0 + win);
            int64_t cost = 0;
            ord.clear();
            for (int r = r0; r < r1; ++r) {
                const int k0 = cur[(size_t)r], kend = M.rp[(size_t)r + 1];
                int k = k0;
                while (k < kend && M.ci[(size_t)k] < cend) ++k;
                cur[(size_t)r] = k;
                const int L = k - k0;
                if (L > SLONG) {                                     // summed by a wave of its own, not stored by a slice
                    out.lr.push_back(Int4{r, (int)out.lci.size(), (int)out.lci.size() + L, 0});
                    for (int q = 0; q < L; ++q) {
                        out.lci.push_back((unsigned short)(M.ci[(size_t)(k0 + q)] - b * CB)); out.lva.push_back((T)M.va[(size_t)(k0 + q)]);
                        if (with_src) out.lsrc.push_back(k0 + q);
                    }
                    out.long_entries += L;
                    cost += (L + 63) / 64 + 8;
                    if (out.lci.size() > 2000000000ULL) return false;
                    continue;
                }
                ord.push_back(RowRun{L, r, k0});
            }
            std::stable_sort(ord.begin(), ord.end(), [](const RowRun& x_, const RowRun& y_) { return x_.len > y_.len; });
            const int wsl = (r1 - r0 + 63) / 64;                     // slices of this window (long rows leave lanes without a row at its end)
            for (int sl = 0; sl < wsl; ++sl) {
                const int64_t g = (int64_t)b * nsl + (int64_t)w * spw + sl;
                out.sl_off[(size_t)g] = (int)units;
                const int p0 = sl * 64;
                const int L = p0 < (int)ord.size() ? ord[(size_t)p0].len : 0, nu = (L + E - 1) / E;
                for (int lane = 0; lane < 64; ++lane)
                    if (p0 + lane < (int)ord.size()) out.perm[(size_t)(g * 64 + lane)] = (unsigned short)(ord[(size_t)(p0 + lane)].row - r0);
                const size_t base = out.cols.size();
                out.cols.resize(base + (size_t)nu * 64 * E, (unsigned short)CB); out.vals.resize(base + (size_t)nu * 64 * E, T(0));
                if (with_src) out.src.resize(base + (size_t)nu * 64 * E, -1);
                for (int lane = 0; lane < 64 && p0 + lane < (int)ord.size(); ++lane) {
                    const RowRun& rr = ord[(size_t)(p0 + lane)];
                    for (int j = 0; j < rr.len; ++j) {
                        const size_t at = base + ((size_t)(j / E) * 64 + (size_t)lane) * E + (size_t)(j % E);
                        out.cols[at] = (unsigned short)(M.ci[(size_t)(rr.first + j)] - b * CB); out.vals[at] = (T)M.va[(size_t)(rr.first + j)];
                        if (with_src) out.src[at] = rr.first + j;
                    }
                    out.entries += rr.len;
                }
                units += nu; cost += nu + 2;
                if (units > 30000000LL) return false;                // int32 unit offsets x 64 lanes: leave such matrices to the task form
            }
            out.lr_ptr[(size_t)b * nwin + w + 1] = (int)out.lr.size();
            wcost[(size_t)b * nwin + w] = cost;
        }
    }
    out.sl_off[(size_t)nblk * nsl] = (int)units;
    out.padded = units * 64 * E;
    // most of the matrix in long rows (a moderately dense matrix: every row holds more than SLONG entries per block): the slices would be empty
    // and lci / lva a second copy of the matrix -- the task form streams such rows in chunks of BCHUNK instead
    if (out.long_entries * 2 > nnz) return false;
    if (want_staged) {
        // whole windows per workgroup: the same number of consecutive windows each (the last workgroups may go without -- they return at once); kept only if no
        // workgroup ends up with more than 9/8 of the mean cost (a skewed matrix is better served by the slice-granular ranges below, whose row sums are stored
        // lane by lane).  (Balancing the window ranges by cost instead handed single workgroups a third window where two were due: with whole windows as the unit
        // a workgroup can only be under its share, never over it, if it takes exactly its count.)
        const int wpb = std::min(nwin, target), per = (nwin + wpb - 1) / wpb;
        std::vector<int> wg((size_t)nblk * (wpb + 1), 0);
        bool balanced = true;
        for (int b = 0; b < nblk && balanced; ++b) {
            const int64_t* wc = &wcost[(size_t)b * nwin];
            int* wp = &wg[(size_t)b * (wpb + 1)];
            int64_t tot = 0, worst = 0;
            for (int k = 0; k < wpb; ++k) {
                const int w0 = std::min(nwin, k * per), w1 = std::min(nwin, (k + 1) * per);
                int64_t mine = 0;
                for (int w = w0; w < w1; ++w) mine += wc[w];
                tot += mine; worst = std::max(worst, mine);
                wp[k] = std::min(nsl, w0 * spw);
            }
            wp[wpb] = nsl;
            if (worst * wpb * 8 > tot * 9 + 64 * (int64_t)wpb) balanced = false;
        }
        if (balanced) { out.staged = 1; out.wpb = wpb; out.wg_ptr = std::move(wg); }
    }
    if (!out.staged) {
    out.wpb = std::max(1, std::min(nsl, std::max(1, wgs) / nblk));  // floor: a launch of at most `wgs` workgroups is resident at once (2 per CU)
    out.wg_ptr.assign((size_t)nblk * (out.wpb + 1), 0);
    for (int b = 0; b < nblk; ++b) {                                 // slice ranges of equal cost (units + a fixed part per slice)
        const int* so = &out.sl_off[(size_t)b * nsl];
        const int64_t tot = (int64_t)(so[nsl] - so[0]) + 2 * (int64_t)nsl;
        int* wp = &out.wg_ptr[(size_t)b * (out.wpb + 1)];
        int w = 1;
        for (int sl = 0; sl < nsl; ++sl) {
            const int64_t run_ = (int64_t)(so[sl + 1] - so[0]) + 2 * (int64_t)(sl + 1);
            while (w < out.wpb && run_ * out.wpb >= tot * w) wp[w++] = sl + 1;
        }
        while (w <= out.wpb) wp[w++] = nsl;
    }
    }
    out.cols.resize(out.cols.size() + 64 * E, (unsigned short)CB); out.vals.resize(out.vals.size() + 64 * E, T(0));   // a slice without units still reads one
    out.lci.resize(out.lci.size() + 64, 0); out.lva.resize(out.lva.size() + 64, T(0));
    if (with_src) { out.src.resize(out.vals.size(), -1); out.lsrc.resize(out.lva.size(), -1); }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// task form
// ---------------------------------------------------------------------------------------------------------------------------------------
template <typename T> void build_tasks(const CsrHost& M, int wgs, TaskLayout<T>& out, bool with_src) {
    constexpr int CB = cb_of<T>();
    const int nrows = M.nrows, ncols = M.ncols;
    const int64_t nnz = (int64_t)M.ci.size();
    const int nblk = std::max(1, (ncols + CB - 1) / CB);
    if ((int64_t)nblk * ((int64_t)nrows + 1) > 2000000000LL) throw std::length_error("row-pointer table of the column-blocked copy exceeds 2^31 entries");
    out = TaskLayout<T>();
    out.nrows = nrows; out.ncols = ncols; out.nblk = nblk;
    // per block: entries per row, then the tasks (consecutive rows holding <= BCHUNK entries, at most max_rows rows; a longer row stands alone),
    // then the offsets -- every TASK starts at a multiple of four entries (the kernel reads four consecutive entries per thread with aligned 8- and
    // 16-byte loads); the gap in front of it holds zero entries, which the last row of the previous task sums up harmlessly
    std::vector<int>& brp = out.brp;
    brp.assign((size_t)nblk * ((size_t)nrows + 1), 0);
    for (int r = 0; r < nrows; ++r)
        for (int k = M.rp[(size_t)r]; k < M.rp[(size_t)r + 1]; ++k) brp[(size_t)(M.ci[(size_t)k] / CB) * ((size_t)nrows + 1) + (size_t)r + 1]++;
    out.task_ptr.assign((size_t)nblk + 1, 0); out.lr_ptr.assign((size_t)nblk + 1, 0);
    // lanes per row segment: 4 while a row holds <= ~8 entries per block on average, else 8; a task's rows are summed in ONE group of four passes
    out.lpr4 = (nnz <= (int64_t)8 * nrows * nblk) ? 1 : 0;
    const int max_rows = 4 * (BTHREADS / (out.lpr4 ? 4 : 8));       // 512 / 256 rows per task at most (k_spmv_blk: RPP * PG)
    int64_t run = 0;                                                 // blocks back to back, rows in order inside a block
    auto align4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
    std::vector<int> cnt((size_t)nrows);
    for (int b = 0; b < nblk; ++b) {
        int* q_ = &brp[(size_t)b * ((size_t)nrows + 1)];
        for (int r = 0; r < nrows; ++r) cnt[(size_t)r] = q_[r + 1];
        int64_t acc = align4(run); int start = 0; int64_t tnnz = 0;
        q_[0] = (int)acc;
        for (int r = 0; r < nrows; ++r) {
            const bool alone = cnt[(size_t)r] > BCHUNK;
            if (r > start && (alone || tnnz + cnt[(size_t)r] > BCHUNK || r - start >= max_rows)) {        // close [start, r)
                const int64_t e_ = align4(acc);
                out.tasks.push_back(Int4{start, r, q_[start], (int)e_});
                acc = e_; start = r; tnnz = 0;
            }
            if (acc + cnt[(size_t)r] + 64 > 2000000000LL) throw std::length_error("more than 2^31 entries in the column-blocked copy");
            q_[r] = (int)acc; acc += cnt[(size_t)r]; tnnz += cnt[(size_t)r];
            if (alone) {                                             // a row longer than a task: summed by a whole workgroup behind the tasks
                const int64_t e_ = align4(acc);
                out.lr.push_back(Int4{r, q_[r], (int)acc, 0});
                acc = e_; start = r + 1; tnnz = 0;
            }
        }
        const int64_t e_ = align4(acc);
        if (start < nrows) out.tasks.push_back(Int4{start, nrows, q_[start], (int)e_});
        q_[nrows] = (int)e_;
        run = e_;
        out.task_ptr[(size_t)b + 1] = (int)out.tasks.size(); out.lr_ptr[(size_t)b + 1] = (int)out.lr.size();
    }
    if (run + 64 > 2000000000LL) throw std::length_error("more than 2^31 entries in the column-blocked copy");
    out.bci.assign((size_t)run + 64, 0);
    out.bva.assign((size_t)run + 64, T(0));
    if (with_src) out.src.assign((size_t)run + 64, -1);
    {
        std::vector<int> pos((size_t)nblk * (size_t)nrows);
        for (int b = 0; b < nblk; ++b) for (int r = 0; r < nrows; ++r) pos[(size_t)b * nrows + r] = brp[(size_t)b * ((size_t)nrows + 1) + r];
        for (int r = 0; r < nrows; ++r)
            for (int k = M.rp[(size_t)r]; k < M.rp[(size_t)r + 1]; ++k) {
                const int b = M.ci[(size_t)k] / CB; int& w_ = pos[(size_t)b * nrows + r];
                out.bci[(size_t)w_] = (unsigned short)(M.ci[(size_t)k] - b * CB); out.bva[(size_t)w_] = (T)M.va[(size_t)k];
                if (with_src) out.src[(size_t)w_] = k;
                ++w_;
            }
    }
    // workgroups per block: enough for the tasks AND for the long rows (dealt round-robin over the block's workgroups behind the tasks) -- a block whose rows are
    // all longer than a task (a dense-ish matrix: isotonic regression's P at n = 3000 is 3000 rows of 3000 entries) has no task at all, and one workgroup walking
    // its 3000 rows one after the other took 15 ms per product
    int maxt = 1, maxu = 1;
    for (int b = 0; b < nblk; ++b) {
        const int nt_b = out.task_ptr[(size_t)b + 1] - out.task_ptr[(size_t)b], nl_b = out.lr_ptr[(size_t)b + 1] - out.lr_ptr[(size_t)b];
        maxt = std::max(maxt, nt_b); maxu = std::max(maxu, nt_b + nl_b);
    }
    out.wpb = std::max(1, std::min(maxu, (std::max(1, wgs) + nblk - 1) / nblk));   // about two workgroups per CU over the launch
    out.wpb = std::max(out.wpb, (maxt + BMAXT - 1) / BMAXT);        // at most BMAXT tasks per workgroup
    out.per = (maxt + out.wpb - 1) / out.wpb;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the explicit reduced matrix (ItrSolCgInit, LinearSystemSolvers.jl:112-114)
// ---------------------------------------------------------------------------------------------------------------------------------------
int64_t ata_work(const CsrHost& Arows) {
    int64_t w = 0;
    for (int r = 0; r < Arows.nrows; ++r) { const int64_t len = Arows.rp[(size_t)r + 1] - Arows.rp[(size_t)r]; w += len * len; }
    return w;
}

bool reduced_matrix(const CsrHost& P, const CsrHost& Arows, const CsrHost& Acols, int64_t max_nnz, CsrHost& L, std::vector<double>& vAA, std::vector<double>& diag) {
    const int n = P.nrows;
    if (Acols.nrows != n || Arows.ncols != n || P.ncols != n) throw std::invalid_argument("reduced_matrix: shapes do not match");
    // Gustavson, one row (= column: the matrix is symmetric) of A'A at a time: (A'A)[j, i] = sum over the rows r of A that hold column j of A[r, j] * A[r, i]
    std::vector<double> accA((size_t)n, 0.0), accP((size_t)n, 0.0);
    std::vector<int> mark((size_t)n, -1), touched;
    CsrHost out; out.nrows = n; out.ncols = n; out.rp.assign((size_t)n + 1, 0);
    std::vector<double> oAA, odiag;
    for (int j = 0; j < n; ++j) {
        touched.clear();
        auto touch = [&](int i) { if (mark[(size_t)i] != j) { mark[(size_t)i] = j; accA[(size_t)i] = 0.0; accP[(size_t)i] = 0.0; touched.push_back(i); } };
        touch(j);                                                                  // the diagonal is always stored (sigma lives there)
        for (int k = P.rp[(size_t)j]; k < P.rp[(size_t)j + 1]; ++k) { const int i = P.ci[(size_t)k]; touch(i); accP[(size_t)i] += P.va[(size_t)k]; }
        for (int k = Acols.rp[(size_t)j]; k < Acols.rp[(size_t)j + 1]; ++k) {
            const int r = Acols.ci[(size_t)k]; const double arj = Acols.va[(size_t)k];
            for (int q = Arows.rp[(size_t)r]; q < Arows.rp[(size_t)r + 1]; ++q) { const int i = Arows.ci[(size_t)q]; touch(i); accA[(size_t)i] += arj * Arows.va[(size_t)q]; }
        }
        if ((int64_t)out.ci.size() + (int64_t)touched.size() > max_nnz) return false;
        std::sort(touched.begin(), touched.end());
        for (int i : touched) { out.ci.push_back(i); out.va.push_back(accP[(size_t)i]); oAA.push_back(accA[(size_t)i]); odiag.push_back(i == j ? 1.0 : 0.0); }
        out.rp[(size_t)j + 1] = (int)out.ci.size();
    }
    L = std::move(out); vAA = std::move(oAA); diag = std::move(odiag);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// host interpreters
// ---------------------------------------------------------------------------------------------------------------------------------------
template <typename T> void apply_sell(const SellLayout<T>& L, const T* x, T* y) {
    constexpr int CB = cb_of<T>(), E = sell_e<T>();
    const T nan = std::numeric_limits<T>::quiet_NaN();
    std::vector<T> partial((size_t)L.nblk * L.nrows, nan);           // NaN: a (block, row) nobody writes shows up in y
    std::vector<T> xs((size_t)CB + 8, T(0));
    std::vector<T> wsum((size_t)WMAX);
    const int spw = L.win / 64;
    if (L.win % 64 != 0 || L.win > WMAX || L.nwin != (L.nrows + L.win - 1) / L.win) throw std::logic_error("apply_sell: window geometry");
    auto put = [&](T& dst, T v) { if (dst == dst) dst = nan; else dst = v; };   // a slot written twice is poisoned: it shows up in y
    auto slice_sums = [&](int64_t sb, int s, T* out_lane) {
        const int u0 = L.sl_off[(size_t)(sb + s)], nu = L.sl_off[(size_t)(sb + s + 1)] - u0;
        for (int lane = 0; lane < 64; ++lane) {
            T acc = T(0);
            for (int u = 0; u < nu; ++u)
                for (int e = 0; e < E; ++e) {
                    const size_t at = ((size_t)(u0 + u) * 64 + (size_t)lane) * E + (size_t)e;
                    acc += L.vals.at(at) * xs.at(L.cols.at(at));
                }
            out_lane[lane] = acc;
        }
    };
    auto long_sum = [&](int i) { const Int4 d = L.lr.at((size_t)i); T sum = T(0); for (int k = d.y; k < d.z; ++k) sum += L.lva.at((size_t)k) * xs.at(L.lci.at((size_t)k)); return sum; };
    for (int b = 0; b < L.nblk; ++b) {
        const int c0 = b * CB, cw = std::min(CB, L.ncols - c0);
        std::fill(xs.begin(), xs.end(), T(0));
        for (int c = 0; c < cw; ++c) xs[(size_t)c] = x[c0 + c];
        T* pout = &partial[(size_t)b * L.nrows];
        const int64_t sb = (int64_t)b * L.nsl;
        T acc[64];
        for (int wg = 0; wg < L.wpb; ++wg) {                         // the slice ranges of the workgroups must tile [0, nsl)
            const int s_begin = L.wg_ptr[(size_t)b * (L.wpb + 1) + wg], s_end = L.wg_ptr[(size_t)b * (L.wpb + 1) + wg + 1];
            if (!L.staged) {
                for (int s = s_begin; s < s_end; ++s) {
                    slice_sums(sb, s, acc);
                    for (int lane = 0; lane < 64; ++lane) {
                        const unsigned pm = L.perm[(size_t)((sb + s) * 64 + lane)];
                        if (pm != 0xffffu) put(pout[(size_t)(s / spw) * L.win + pm], acc[lane]);
                    }
                }
                continue;
            }
            // staged: whole windows; the window's row sums (slices, then its long rows) are collected in wsum and stored as one run
            if (s_begin >= s_end) continue;                                 // a workgroup without a window (the kernel returns at once)
            if (s_begin % spw != 0 || (s_end % spw != 0 && s_end != L.nsl)) throw std::logic_error("apply_sell: a staged workgroup does not own whole windows");
            for (int w = s_begin / spw; w * spw < s_end; ++w) {
                std::fill(wsum.begin(), wsum.end(), nan);            // a row of the window nobody sums shows up in y
                for (int s = w * spw; s < std::min((w + 1) * spw, L.nsl); ++s) {
                    slice_sums(sb, s, acc);
                    for (int lane = 0; lane < 64; ++lane) {
                        const unsigned pm = L.perm[(size_t)((sb + s) * 64 + lane)];
                        if (pm != 0xffffu) put(wsum.at(pm), acc[lane]);
                    }
                }
                for (int i = L.lr_ptr[(size_t)b * L.nwin + w]; i < L.lr_ptr[(size_t)b * L.nwin + w + 1]; ++i) put(wsum.at((size_t)(L.lr.at((size_t)i).x - w * L.win)), long_sum(i));
                for (int r = w * L.win; r < std::min(L.nrows, (w + 1) * L.win); ++r) put(pout[(size_t)r], wsum[(size_t)(r - w * L.win)]);
            }
        }
        if (!L.staged)
            for (int i = L.lr_ptr[(size_t)b * L.nwin]; i < L.lr_ptr[(size_t)(b + 1) * L.nwin]; ++i) put(pout[(size_t)L.lr.at((size_t)i).x], long_sum(i));
    }
    for (int r = 0; r < L.nrows; ++r) { T s = T(0); for (int b = 0; b < L.nblk; ++b) s += partial[(size_t)b * L.nrows + r]; y[r] = s; }
}

template <typename T> void apply_tasks(const TaskLayout<T>& L, const T* x, T* y) {
    constexpr int CB = cb_of<T>();
    const T nan = std::numeric_limits<T>::quiet_NaN();
    std::vector<T> partial((size_t)L.nblk * L.nrows, nan);
    std::vector<T> xs((size_t)CB + 8, T(0));
    for (int b = 0; b < L.nblk; ++b) {
        const int c0 = b * CB, cw = std::max(0, std::min(CB, L.ncols - c0));
        std::fill(xs.begin(), xs.end(), T(0));
        for (int c = 0; c < cw; ++c) xs[(size_t)c] = x[c0 + c];
        const int* rpb = &L.brp[(size_t)b * ((size_t)L.nrows + 1)];
        T* pout = &partial[(size_t)b * L.nrows];
        const int nt = L.task_ptr[(size_t)b + 1] - L.task_ptr[(size_t)b];
        if (nt > L.wpb * L.per || L.per > BMAXT) throw std::logic_error("a block's tasks do not fit its workgroups");
        for (int ti = L.task_ptr[(size_t)b]; ti < L.task_ptr[(size_t)b + 1]; ++ti) {
            const Int4 m = L.tasks.at((size_t)ti);
            if (m.z % 4 != 0 || m.w - m.z > BCHUNK || m.y - m.x > 4 * (BTHREADS / (L.lpr4 ? 4 : 8))) throw std::logic_error("task violates the kernel's limits");
            for (int r = m.x; r < m.y; ++r) {
                const int s0 = rpb[r], s1 = (r + 1 < m.y) ? rpb[r + 1] : m.w;   // (the zero entries in front of the next task belong to the last row)
                T acc = T(0);
                for (int k = s0; k < s1; ++k) acc += L.bva.at((size_t)k) * xs.at(L.bci.at((size_t)k));
                T& dst = pout[(size_t)r];
                if (dst == dst) dst = nan; else dst = acc;
            }
        }
        for (int i = L.lr_ptr[(size_t)b]; i < L.lr_ptr[(size_t)b + 1]; ++i) {
            const Int4 d = L.lr.at((size_t)i);
            T sum = T(0);
            for (int k = d.y; k < d.z; ++k) sum += L.bva.at((size_t)k) * xs.at(L.bci.at((size_t)k));
            T& dst = pout[(size_t)d.x];
            if (dst == dst) dst = nan; else dst = sum;
        }
    }
    for (int r = 0; r < L.nrows; ++r) { T s = T(0); for (int b = 0; b < L.nblk; ++b) s += partial[(size_t)b * L.nrows + r]; y[r] = s; }
}

template bool build_sell<double>(const CsrHost&, int, SellLayout<double>&, bool, bool);
template bool build_sell<float>(const CsrHost&, int, SellLayout<float>&, bool, bool);
template void build_tasks<double>(const CsrHost&, int, TaskLayout<double>&, bool);
template void build_tasks<float>(const CsrHost&, int, TaskLayout<float>&, bool);
template void apply_sell<double>(const SellLayout<double>&, const double*, double*);
template void apply_sell<float>(const SellLayout<float>&, const float*, float*);
template void apply_tasks<double>(const TaskLayout<double>&, const double*, double*);
template void apply_tasks<float>(const TaskLayout<float>&, const float*, float*);

}  // namespace layout
}  // namespace qps
