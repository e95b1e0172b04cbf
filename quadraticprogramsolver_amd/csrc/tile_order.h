// tile_order.h -- workgroup id -> tile of a lower-triangular tile grid, for the two launches that enumerate such a grid on a 1-D id range (k_gemm with
// lower_only >= 2, k_chol_update_diag).  Plain index arithmetic, usable from device code and from the host-only test library (tests/capi/layout_shim.cpp,
// tests/test_layout_cpu.py: every lower tile is dealt exactly once, for every grid size).
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define QPS_TILE_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define QPS_TILE_FN inline
#endif

namespace qps {

// Lower tiles of an nt x nt tile grid on a 1-D grid of `nids` workgroup ids (a multiple of 8).  Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of
// its own, and the ~64-96 tiles an XCD works on at one time are the ones whose operand panels it can share.  On the plain 2-D grid (upper tiles returning at once)
// an XCD's tiles are every eighth of a tile row: in the short rows at the top of the triangle ~96 resident tiles span ~40 row panels and 5 column panels -- every
// panel byte is used by two tiles (A'A, n = 4096: L2 hit rate 48 %, 4.7 GB fetched for a 134 MB operand, profiles/r04_n_gemm_f32_counters_after.txt).  Here XCD x
// takes the x-th eighth of the tiles in an order that walks the triangle in 8 x 8 super-blocks (row-major inside a block, blocks row-major inside the triangle):
// 64 consecutive tiles share 16 panels.  A'A 4096 x 4096 x 8192: 1.80 -> 1.39 ms fp32 (99 TFLOP/s), 3.39 -> 2.61 ms fp64.
constexpr int lower_tile_ids(int nt) { return 8 * ((nt * (nt + 1) / 2 + 7) / 8); }
QPS_TILE_FN bool lower_tile_of(int id, int nids, int nt, int& bi, int& bj) {
    constexpr int S = 8;                                                           // (4 ... 16 measure the same: profiles/r04_o_gemm_lower_map.log)
    int t = (id & 7) * (nids >> 3) + (id >> 3);
    const int nsb = (nt + S - 1) / S;
    for (int I = 0; I < nsb; ++I) {
        const int rows = (nt - I * S < S) ? nt - I * S : S, full = rows * I * S, cnt = full + rows * (rows + 1) / 2;
        if (t < cnt) {
            if (t < full) { const int J = t / (rows * S), r = t - J * rows * S; bi = I * S + r / S; bj = J * S + r % S; }
            else { int r = t - full, a = 0; while (r > a) { r -= a + 1; ++a; } bi = I * S + a; bj = I * S + r; }
            return true;
        }
        t -= cnt;
    }
    return false;                                                                  // padding ids of the last XCD
}

// k_chol_update_diag: id 0 is the diagonal workgroup (tile (0, 0) of the trailing matrix and the factorisation of the next diagonal block); ids >= 1 take the other
// lower tiles of the g x g tile grid in row-major order of the triangle.  avoid = 1: ids that are multiples of 8 take no tile (they would share an XCD with id 0).
constexpr int chol_update_ids(int g, int avoid) { return 1 + (g * (g + 1) / 2 - 1) + (avoid ? (g * (g + 1) / 2 - 1) / 7 + 2 : 0); }
QPS_TILE_FN bool chol_update_tile_of(int id, int avoid, int g, int& bi, int& bj) {
    if (id <= 0 || (avoid && (id & 7) == 0)) return false;
    const int t = avoid ? id - (id >> 3) : id;                                     // 1-based index into the lower tiles behind (0, 0)
    if (t >= g * (g + 1) / 2) return false;
    bi = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);                        // t = bi (bi + 1) / 2 + bj, 0 <= bj <= bi
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    bj = t - bi * (bi + 1) / 2;
    return true;
}

}  // namespace qps
