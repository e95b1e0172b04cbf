// ldl_symbolic.cpp -- host-side symbolic analysis of the KKT matrix for the sparse direct plugin (see ldl_symbolic.h).
// Plain C++ (no HIP): ordering, elimination tree, level sets, pattern of L, scatter map of K.
#include "ldl_symbolic.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

namespace qps {

// =====================================================================================================================
// Approximate minimum degree on a quotient graph (Amestoy, Davis, Duff: "An approximate minimum degree ordering algorithm",
// SIMAX 1996 -- the algorithm behind the `amd` the reference's LDL' packages call).  Variables carry a weight nv (super-
// variables); an eliminated pivot becomes an element whose variable list stands for the clique it created.
//   * element absorption: the elements adjacent to the pivot are merged into the new element;
//   * approximate external degree  d_i = min(remaining, d_i + |Lp \ i|, |A_i| + |Lp \ i| + sum_e |Le \ Lp|);
//   * aggressive absorption (|Le \ Lp| = 0), mass elimination (no connection outside Lp), indistinguishable variables
//     found by hashing and merged; nearly dense rows (more than 4 sqrt(N) entries; AMD's customary 10 sqrt(N) leaves the
//     100 n-by-15 % columns of the lasso / Huber / SVM generators in the graph, each hanging on ~1500 two-variable elements that
//     every pivot rescans: 1 s instead of 40 ms of ordering at numElements = 100) are ordered last.
// Lists are std::vectors cleaned lazily (stale members are skipped by their state), no in-place garbage collection.
// =====================================================================================================================
std::vector<int> amd_order(int N, const std::vector<std::vector<int>>& adj) {
    enum : signed char { VAR = 0, ELEM = 1, DEAD_ELEM = 2, MERGED = 3, DENSE = 4 };
    std::vector<signed char> state(N, VAR);
    std::vector<int> nv(N, 1), deg(N, 0);
    std::vector<std::vector<int>> A(N), E(N), Le(N), members(N);
    std::vector<int64_t> wsize(N, 0);                          // weighted |Le| of a live element
    std::vector<int> order; order.reserve(N);
    std::vector<int> dense_nodes;
    static const double dense_fac = [] { const char* e = getenv("QPS_LDL_DENSE_FACTOR"); return e ? atof(e) : 4.0; }();
    const int dense_thr = std::max(16, (int)(dense_fac * std::sqrt((double)std::max(N, 1))));
    int live = 0;
    for (int i = 0; i < N; ++i)
        if ((int)adj[i].size() > dense_thr) { state[i] = DENSE; dense_nodes.push_back(i); }
    for (int i = 0; i < N; ++i) {
        if (state[i] != VAR) continue;
        ++live;
        A[i].reserve(adj[i].size());
        for (int j : adj[i]) if (j != i && state[j] == VAR) A[i].push_back(j);
        deg[i] = (int)A[i].size();
    }
    // degree buckets (doubly linked)
    std::vector<int> head(N + 1, -1), nxt(N, -1), prv(N, -1);
    auto bucket_insert = [&](int i) { const int d = std::min(std::max(deg[i], 0), N); nxt[i] = head[d]; prv[i] = -1; if (head[d] >= 0) prv[head[d]] = i; head[d] = i; };
    auto bucket_remove = [&](int i) {
        const int d = std::min(std::max(deg[i], 0), N);
        if (prv[i] >= 0) nxt[prv[i]] = nxt[i]; else if (head[d] == i) head[d] = nxt[i];
        if (nxt[i] >= 0) prv[nxt[i]] = prv[i];
        nxt[i] = prv[i] = -1;
    };
    for (int i = 0; i < N; ++i) if (state[i] == VAR) bucket_insert(i);
    std::vector<int> mark(N, -1), wstamp(N, -1); std::vector<int64_t> w(N, 0);
    std::vector<int> Lp, hbucket_head, hnext(N, -1); std::vector<unsigned> hval(N, 0);
    int mindeg = 0, eliminated = 0, tag = 0;
    while (eliminated < live) {
        while (mindeg <= N && head[mindeg] < 0) ++mindeg;
        if (mindeg > N) throw std::runtime_error("amd_order: degree lists exhausted before every variable was ordered");
        const int p = head[mindeg];
        bucket_remove(p);
        ++tag;
        // ---- the new element Lp = (A_p  U  union of the elements of p) \ {p}
        Lp.clear();
        mark[p] = tag;
        for (int i : A[p]) if (state[i] == VAR && mark[i] != tag) { mark[i] = tag; Lp.push_back(i); }
        for (int e : E[p]) {
            if (state[e] != ELEM) continue;
            for (int i : Le[e]) if (state[i] == VAR && mark[i] != tag) { mark[i] = tag; Lp.push_back(i); }
            state[e] = DEAD_ELEM; std::vector<int>().swap(Le[e]);                 // absorbed into p
        }
        std::vector<int>().swap(A[p]); std::vector<int>().swap(E[p]);
        state[p] = ELEM;
        int nvpiv = nv[p];
        for (int i : Lp) bucket_remove(i);
        // ---- |Le \ Lp| for every element adjacent to a member of Lp
        for (int i : Lp)
            for (int e : E[i]) {
                if (state[e] != ELEM) continue;
                if (wstamp[e] != tag) { wstamp[e] = tag; w[e] = wsize[e]; }
                w[e] -= nv[i];
            }
        // ---- clean the lists of every i in Lp, external degree, mass elimination
        std::vector<int64_t> dext(Lp.size(), 0);
        int64_t degme = 0;
        for (size_t t = 0; t < Lp.size(); ++t) {
            const int i = Lp[t];
            int64_t d = 0; size_t k = 0;
            for (int e : E[i]) {
                if (state[e] != ELEM) continue;
                if (w[e] <= 0) { state[e] = DEAD_ELEM; std::vector<int>().swap(Le[e]); continue; }   // aggressive absorption: Le is inside Lp
                E[i][k++] = e; d += w[e];
            }
            E[i].resize(k);
            k = 0;
            for (int j : A[i]) if (state[j] == VAR && mark[j] != tag) { A[i][k++] = j; d += nv[j]; }
            A[i].resize(k);
            if (d == 0) {                                                          // indistinguishable from the pivot: eliminate with it
                state[i] = MERGED; members[p].push_back(i);
                nvpiv += nv[i]; mark[i] = -2;
                std::vector<int>().swap(A[i]); std::vector<int>().swap(E[i]);
                dext[t] = -1;
            } else { dext[t] = d; degme += nv[i]; }
        }
        eliminated += nvpiv;
        const int remaining = live - eliminated;
        // ---- approximate degrees and hash keys of the survivors
        std::vector<int> surv; surv.reserve(Lp.size());
        for (size_t t = 0; t < Lp.size(); ++t) {
            const int i = Lp[t];
            if (dext[t] < 0) continue;
            const int64_t lpi = degme - nv[i];
            int64_t d = std::min<int64_t>((int64_t)deg[i] + lpi, dext[t] + lpi);
            d = std::min<int64_t>(d, remaining - nv[i]);
            deg[i] = (int)std::max<int64_t>(d, 0);
            E[i].push_back(p);
            unsigned h = 0;
            for (int j : A[i]) h += (unsigned)j;
            for (int e : E[i]) h += (unsigned)e;
            hval[i] = h;
            surv.push_back(i);
        }
        // ---- indistinguishable variables: same hash, same element set, same variable set -> one supervariable
        if (surv.size() > 1) {
            const unsigned hsize = (unsigned)surv.size() * 2 + 1;
            hbucket_head.assign(hsize, -1);
            for (int i : surv) { const unsigned b = hval[i] % hsize; hnext[i] = hbucket_head[b]; hbucket_head[b] = i; }
            for (unsigned b = 0; b < hsize; ++b) {
                for (int i = hbucket_head[b]; i >= 0; i = hnext[i]) {
                    if (state[i] != VAR || hnext[i] < 0) continue;                 // (the last of a bucket has nobody left to compare with)
                    ++tag;                                                         // fresh stamp for the comparison marks
                    for (int j : A[i]) mark[j] = tag;
                    for (int e : E[i]) mark[e] = tag;
                    int prevj = i;
                    for (int j = hnext[i]; j >= 0; j = hnext[j]) {
                        bool same = state[j] == VAR && hval[j] == hval[i] && A[j].size() == A[i].size() && E[j].size() == E[i].size();
                        if (same) for (int k : A[j]) if (mark[k] != tag) { same = false; break; }
                        if (same) for (int e : E[j]) if (mark[e] != tag) { same = false; break; }
                        if (same) {
                            nv[i] += nv[j]; deg[i] = std::max(deg[i] - nv[j], 0);
                            state[j] = MERGED; members[i].push_back(j);
                            std::vector<int>().swap(A[j]); std::vector<int>().swap(E[j]);
                            hnext[prevj] = hnext[j];                               // unlink j from the bucket
                        } else prevj = j;
                    }
                }
            }
            // the marks used `tag` values; make sure the pivot stamp of the next round is fresh
        }
        // ---- the element p and the degree lists
        Le[p].clear(); int64_t wp = 0;
        for (int i : surv) if (state[i] == VAR) { Le[p].push_back(i); wp += nv[i]; bucket_insert(i); if (deg[i] < mindeg) mindeg = deg[i]; }
        wsize[p] = wp;
        if (Le[p].empty()) state[p] = DEAD_ELEM;
        order.push_back(p);
    }
    // pivots in elimination order, each followed by the variables merged into it (recursively); dense rows last, lightest first
    std::vector<int> out; out.reserve(N);
    std::vector<int> stack;
    for (int p : order) {
        stack.push_back(p);
        while (!stack.empty()) {
            const int v = stack.back(); stack.pop_back();
            out.push_back(v);
            for (auto it = members[v].rbegin(); it != members[v].rend(); ++it) stack.push_back(*it);
        }
    }
    std::stable_sort(dense_nodes.begin(), dense_nodes.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    for (int i : dense_nodes) out.push_back(i);
    if ((int)out.size() != N) throw std::runtime_error("amd_order: internal error (ordering is not a permutation)");
    std::vector<char> seen(N, 0);
    for (int v : out) { if (v < 0 || v >= N || seen[v]) throw std::runtime_error("amd_order: internal error (duplicate in ordering)"); seen[v] = 1; }
    return out;
}

namespace {

// lower-triangular pattern of the permuted KKT matrix as sorted column lists (strictly lower), plus the origin of every entry
struct KEntry { int row, col, src; };

}  // namespace

// Fallback ordering for chain-like graphs (banded KKT systems: isotonic regression, trend filtering, control problems), where minimum degree
// peels the band from its ends and leaves an elimination tree as deep as the matrix is long -- one dependent launch per column.  The vertices are
// laid out on a line by breadth-first search (every edge then spans at most `bw` positions), and the line is dissected recursively: the `bw`
// positions in the middle of an interval separate its halves, the halves are ordered first (recursively; short intervals keep their line order),
// the separator last.  Depth of the elimination tree: leaf length + bw * log2(N / leaf) instead of N.  Returns an empty vector when the line
// order is not narrow enough for this to help (bw > N / 16).
std::vector<int> line_dissection_order(int N, const std::vector<std::vector<int>>& adj) {
    std::vector<int> pos(N, -1), line; line.reserve(N);
    std::vector<int> degree_order(N);
    for (int i = 0; i < N; ++i) degree_order[i] = i;
    std::stable_sort(degree_order.begin(), degree_order.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    auto bfs = [&](int start, std::vector<int>& out, std::vector<int>& mark, int stamp) {
        out.clear(); out.push_back(start); mark[start] = stamp;
        for (size_t h = 0; h < out.size(); ++h)
            for (int w : adj[out[h]]) if (mark[w] != stamp && pos[w] < 0) { mark[w] = stamp; out.push_back(w); }
    };
    std::vector<int> mark(N, 0), comp; int stamp = 0;
    for (int s0 : degree_order) {
        if (pos[s0] >= 0) continue;
        bfs(s0, comp, mark, ++stamp);                       // pseudo-peripheral start: the last vertex of a search is far from its root
        bfs(comp.back(), comp, mark, ++stamp);
        bfs(comp.back(), comp, mark, ++stamp);
        for (int v : comp) { pos[v] = (int)line.size(); line.push_back(v); }
    }
    int bw = 1;
    for (int v = 0; v < N; ++v) for (int w : adj[v]) bw = std::max(bw, std::abs(pos[v] - pos[w]));
    if ((int64_t)bw * 16 > N) return std::vector<int>();
    const int leaf = std::max(4 * bw, 128);
    std::vector<int> order; order.reserve(N);
    // explicit stack: (lo, hi, separator range to emit after both halves)
    struct Item { int lo, hi, state, slo, shi; };
    std::vector<Item> st; st.push_back({0, N, 0, 0, 0});
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.state == 1) { for (int k = it.slo; k < it.shi; ++k) order.push_back(line[k]); continue; }
        if (it.hi - it.lo <= leaf) { for (int k = it.lo; k < it.hi; ++k) order.push_back(line[k]); continue; }
        const int mid = it.lo + (it.hi - it.lo) / 2, slo = mid - bw / 2, shi = slo + bw;
        st.push_back({0, 0, 1, slo, shi});                  // emitted last (popped last)
        st.push_back({shi, it.hi, 0, 0, 0});
        st.push_back({it.lo, slo, 0, 0, 0});
    }
    return order;
}

LdlSymbolic ldl_analyze(int n, int m, const int64_t* Pcp, const int64_t* Pri, const int64_t* Acp, const int64_t* Ari, int base,
                        int max_tail, int min_level_width, int max_levels) {
    LdlSymbolic S;
    S.n = n; S.m = m; S.N = n + m;
    const int N = S.N;
    // QPS_LDL_TIMING=1: phase times of the analysis on stderr (tests/tools/cpu_ldl_analyze_timing.py)
    const bool timing = [] { const char* e = getenv("QPS_LDL_TIMING"); return e && atoi(e) != 0; }();
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = timing ? now() : 0.0;
    auto lap = [&](const char* what) { if (timing) { const double t = now(); fprintf(stderr, "[ldl_analyze] %-34s %7.2f ms\n", what, t - t_last); t_last = t; } };
    const int64_t pnnz = Pcp[n] - base, annz = Acp[n] - base;
    // ---- entries of the strictly lower triangle of K in the ORIGINAL numbering ([x; nu]); value table = P entries, then A entries
    std::vector<KEntry> ent; ent.reserve((size_t)(pnnz / 2 + annz));
    std::vector<int> dposP(N, -1);
    for (int j = 0; j < n; ++j)
        for (int64_t k = Pcp[j] - base; k < Pcp[j + 1] - base; ++k) {
            const int i = (int)(Pri[k] - base);
            if (i > j) ent.push_back({i, j, (int)k});
            else if (i == j && dposP[j] < 0) dposP[j] = (int)k;                    // (duplicates of a diagonal entry: first one wins; the caller sums duplicates beforehand)
        }
    for (int j = 0; j < n; ++j)
        for (int64_t k = Acp[j] - base; k < Acp[j + 1] - base; ++k) ent.push_back({n + (int)(Ari[k] - base), j, (int)(pnnz + k)});
    S.nnzK = (int64_t)ent.size();
    lap("entries of K");
    // ---- ordering
    std::vector<std::vector<int>> adj(N);
    {
        std::vector<int> cnt(N, 0);
        for (const KEntry& e : ent) { cnt[e.row]++; cnt[e.col]++; }
        for (int i = 0; i < N; ++i) adj[i].reserve(cnt[i]);
        for (const KEntry& e : ent) { adj[e.row].push_back(e.col); adj[e.col].push_back(e.row); }
        for (int i = 0; i < N; ++i) { std::sort(adj[i].begin(), adj[i].end()); adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end()); }
        lap("adjacency lists");
        S.perm = amd_order(N, adj);
        lap("minimum-degree ordering");
    }
    std::vector<int> ip(N);
    // ordering pass: 0 = minimum degree, 1 = line dissection on trial, 2 = final (whichever was kept)
    int pass = 0, md_levels = 0; std::vector<int> md_perm;
    const int dissect_from = [] { const char* e = getenv("QPS_LDL_DISSECT_LEVELS"); return e ? atoi(e) : 256; }();
retry_with_other_ordering:
    for (int k = 0; k < N; ++k) ip[S.perm[k]] = k;
    // ---- elimination tree of the permuted matrix (Liu's algorithm with path compression on the row subtrees)
    auto lower_cols = [&](const std::vector<int>& iperm, std::vector<int>& cptr, std::vector<int>& rows) {
        // strictly-lower pattern by column in the permuted numbering (unsorted rows)
        cptr.assign(N + 1, 0);
        for (const KEntry& e : ent) { const int a = iperm[e.row], b = iperm[e.col]; cptr[std::min(a, b) + 1]++; }
        for (int j = 0; j < N; ++j) cptr[j + 1] += cptr[j];
        rows.resize(ent.size());
        std::vector<int> pos(cptr.begin(), cptr.end() - 1);
        for (const KEntry& e : ent) { const int a = iperm[e.row], b = iperm[e.col]; rows[pos[std::min(a, b)]++] = std::max(a, b); }
    };
    auto etree = [&](const std::vector<int>& cptr, const std::vector<int>& rows, std::vector<int>& parent) {
        // row-oriented sweep needs, for every row i, the columns j < i with K_ij != 0: transpose of the column lists
        std::vector<int> rptr(N + 1, 0), cols(rows.size());
        for (int r : rows) rptr[r + 1]++;
        for (int i = 0; i < N; ++i) rptr[i + 1] += rptr[i];
        { std::vector<int> pos(rptr.begin(), rptr.end() - 1); for (int j = 0; j < N; ++j) for (int k = cptr[j]; k < cptr[j + 1]; ++k) cols[pos[rows[k]]++] = j; }
        parent.assign(N, -1);
        std::vector<int> anc(N, -1);
        for (int i = 0; i < N; ++i)
            for (int k = rptr[i]; k < rptr[i + 1]; ++k) {
                int j = cols[k];
                while (j != -1 && j < i) { const int nx = anc[j]; anc[j] = i; if (nx == -1) parent[j] = i; j = nx; }
            }
    };
    std::vector<int> cptr, rows, parent;
    lower_cols(ip, cptr, rows);
    etree(cptr, rows, parent);
    lap("elimination tree");
    // ---- levels (leaves = 0) and the equivalent reordering "by level, then by position"
    std::vector<int> level(N, 0);
    for (int j = 0; j < N; ++j) if (parent[j] >= 0) level[parent[j]] = std::max(level[parent[j]], level[j] + 1);   // children precede parents
    int height = 0; for (int j = 0; j < N; ++j) height = std::max(height, level[j] + 1);
    S.levels_total = height;
    std::vector<int> lcount(height + 1, 0);
    for (int j = 0; j < N; ++j) lcount[level[j] + 1]++;
    for (int l = 0; l < height; ++l) lcount[l + 1] += lcount[l];
    std::vector<int> newpos(N);
    { std::vector<int> pos(lcount.begin(), lcount.end() - 1); for (int j = 0; j < N; ++j) newpos[j] = pos[level[j]]++; }
    // tail: the longest suffix of levels none of which is wide, capped by max_tail; if the narrow suffix does not fit, the widest
    // prefix of it stays sparse (one launch per level) -- bounded by max_levels
    int first_tail_level = height;
    while (first_tail_level > 0 && (lcount[first_tail_level] - lcount[first_tail_level - 1]) < min_level_width && (N - lcount[first_tail_level - 1]) <= max_tail)
        --first_tail_level;
    if (pass == 0 && first_tail_level > dissect_from) {
        // every sparse level is a dependent launch in each triangular sweep: minimum degree left a chain (banded system) -- try the dissected
        // breadth-first line (line_dissection_order above) and keep it when it at least halves the number of levels
        std::vector<int> alt = line_dissection_order(N, adj);
        lap("line dissection ordering (deep elimination tree under minimum degree)");
        if (!alt.empty()) { pass = 1; md_levels = first_tail_level; md_perm = S.perm; S.perm.swap(alt); goto retry_with_other_ordering; }
    }
    if (pass == 1) {
        pass = 2;
        if (2 * first_tail_level > md_levels) { S.perm = md_perm; goto retry_with_other_ordering; }   // no better: back to minimum degree
    }
    if (first_tail_level > max_levels) throw std::runtime_error("sparse KKT LDL': elimination tree too deep for the level-scheduled solves (" + std::to_string(first_tail_level) +
                                                               " sparse levels); use the CG plugin for this problem");
    { std::vector<std::vector<int>>().swap(adj); }
    S.Ns = lcount[first_tail_level]; S.Nt = N - S.Ns;
    S.level_ptr.assign(lcount.begin(), lcount.begin() + first_tail_level + 1);
    // compose the permutations
    {
        std::vector<int> perm2(N);
        for (int j = 0; j < N; ++j) perm2[newpos[j]] = S.perm[j];
        S.perm.swap(perm2);
        S.iperm.resize(N);
        for (int k = 0; k < N; ++k) S.iperm[S.perm[k]] = k;
    }
    S.sign.resize(N);
    for (int k = 0; k < N; ++k) S.sign[k] = S.perm[k] < n ? 1 : -1;
    S.dpos_P.resize(N);
    for (int k = 0; k < N; ++k) S.dpos_P[k] = S.perm[k] < n ? dposP[S.perm[k]] : -1;
    lap("levels + composed permutation");
    // ---- pattern of L by rows in the final numbering: row i = union of the etree paths from the non-zeros of row i of K
    lower_cols(S.iperm, cptr, rows);
    etree(cptr, rows, parent);
    std::vector<int> rptr(N + 1, 0), kcols(rows.size());
    for (int r : rows) rptr[r + 1]++;
    for (int i = 0; i < N; ++i) rptr[i + 1] += rptr[i];
    { std::vector<int> pos(rptr.begin(), rptr.end() - 1); for (int j = 0; j < N; ++j) for (int k = cptr[j]; k < cptr[j + 1]; ++k) kcols[pos[rows[k]]++] = j; }
    const int Ns = S.Ns;
    // Two sweeps over the row subtrees and one linear pass, no sorting: the first sweep counts, the second fills the CSC arrays (rows are
    // visited in ascending order, so every column comes out sorted by row), the linear pass over the columns in ascending order then
    // fills the CSR arrays (every row comes out sorted by column) together with the CSR -> CSC position map.
    S.rp.assign(N + 1, 0);
    S.cp.assign(Ns + 1, 0);
    std::vector<int> stamp(N, -1);
    int64_t total = 0, exact = 0;
    for (int i = 0; i < N; ++i) {
        stamp[i] = i;
        int cnt = 0;
        for (int k = rptr[i]; k < rptr[i + 1]; ++k)
            for (int j = kcols[k]; j != -1 && j < i && stamp[j] != i; j = parent[j]) { stamp[j] = i; if (j < Ns) { ++cnt; S.cp[j + 1]++; } else ++exact; }
        total += cnt;
        S.rp[i + 1] = (int)std::min<int64_t>(total, 2147483647LL);
    }
    if (total > 2000000000LL) throw std::runtime_error("sparse KKT LDL': the factor has more than 2^31 non-zeros under the minimum-degree ordering; use the CG plugin");
    S.nnzL_exact = total + exact;
    S.nnzL = total + (int64_t)S.Nt * (S.Nt - 1) / 2;
    for (int j = 0; j < Ns; ++j) S.cp[j + 1] += S.cp[j];
    S.ri.resize((size_t)total); S.ci.resize((size_t)total); S.csr2csc.resize((size_t)total);
    {
        std::vector<int> pos(S.cp.begin(), S.cp.end() - 1);
        std::fill(stamp.begin(), stamp.end(), -1);
        for (int i = 0; i < N; ++i) {
            stamp[i] = i;
            for (int k = rptr[i]; k < rptr[i + 1]; ++k)
                for (int j = kcols[k]; j != -1 && j < i && stamp[j] != i; j = parent[j]) { stamp[j] = i; if (j < Ns) S.ri[pos[j]++] = i; }
        }
    }
    {
        std::vector<int> rpos(S.rp.begin(), S.rp.end() - 1);
        for (int j = 0; j < Ns; ++j)
            for (int q = S.cp[j]; q < S.cp[j + 1]; ++q) { const int k = rpos[S.ri[q]]++; S.ci[k] = j; S.csr2csc[k] = q; }
    }
    lap("pattern of L (CSC + CSR)");
    // ---- where the entries of K start: CSR position (binary search in the sorted row) or dense tail position
    S.ldt = ((S.Nt + 63) / 64) * 64;
    S.k_dst.resize(ent.size()); S.k_src.resize(ent.size());
    for (size_t e = 0; e < ent.size(); ++e) {
        const int a = S.iperm[ent[e].row], b = S.iperm[ent[e].col];
        const int i = std::max(a, b), j = std::min(a, b);
        S.k_src[e] = ent[e].src;
        if (j < Ns) {
            const int* lo = S.ci.data() + S.rp[i]; const int* hi = S.ci.data() + S.rp[i + 1];
            const int* it = std::lower_bound(lo, hi, j);
            if (it == hi || *it != j) throw std::runtime_error("sparse KKT LDL': internal error (an entry of K is missing from the pattern of L)");
            S.k_dst[e] = (int64_t)(it - S.ci.data());
        } else S.k_dst[e] = -(1 + (int64_t)(i - Ns) * S.ldt + (j - Ns));
    }
    lap("scatter map of K");
    return S;
}

}  // namespace qps
