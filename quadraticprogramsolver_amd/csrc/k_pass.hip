// k_pass.hip -- launchers of the fused single pass over A for the ADMM rows (kernel template: k_pass_kernel.h).
#include "k_pass_kernel.h"

namespace qps {

static int pass_threads() {
    static int th = [] { const char* e = getenv("QPS_PASS_THREADS"); int v = e ? atoi(e) : 512; return v == 1024 ? 1024 : 512; }();
    return th;
}

// 512-thread workgroups (one per CU, up to 8 column chunks per thread) cover NP <= 8 * 512 * VN (8192 fp64 / 16384 fp32); wider problems take
// the 1024-thread instantiations (half the register tile per thread): NP <= 16384 fp64 / 32768 fp32
template <typename T> static int pass_threads_for(int NP) { return (pass_threads() == 1024 || NP > 8 * 512 * VecOf<T>::N) ? 1024 : 512; }
template <typename T> int apass_max_np() { return 8 * 1024 * VecOf<T>::N; }

template <typename T> int apass_plan(int NP, int MP, int* rows_per_wg, int count) {
    if (NP > apass_max_np<T>() || MP <= 0) { *rows_per_wg = 0; return 0; }
    const int R = 4;
    // about one workgroup per CU over the whole launch (256 CUs), split evenly over the QPs of a batch
    // One workgroup per CU when the register tile is large (KC >= 2: 138+ VGPRs, one 8-wave workgroup fits per CU); narrow
    // problems (KC == 1, <= 98 VGPRs) take four per CU (measured on 256 x n=1024: 164 k -> 213 k QP-iterations/s)
    static const int total_env = [] { const char* e = getenv("QPS_PASS_WGS"); return e ? atoi(e) : 0; }();
    const int kc1 = NP <= pass_threads_for<T>(NP) * VecOf<T>::N;
    const int total = total_env > 0 ? total_env : ((kc1 && count > 1) ? 1024 : 256);   // the wider launch pays for batches only
    const int target = count >= total ? 1 : total / (count < 1 ? 1 : count);
    int rpw = (MP + target - 1) / target;
    rpw = ((rpw + R - 1) / R) * R;
    *rows_per_wg = rpw;
    return (MP + rpw - 1) / rpw;
}

template <typename T, int TH>
void apass_th(hipStream_t st, bool check, const T* A, int64_t ld, int NP, int MP, const T* xx, const T* x_old, T* x_new, T* z,
              T* y, const T* l, const T* u, T alpha, T rho, T* part, T* part2, int64_t part_ld, unsigned long long* slots,
              PassBatch pb) {
    int rpw = 0;
    const int G = apass_plan<T>(NP, MP, &rpw, pb.count);
    pb.slabs = G;
    const int chunk = TH * VecOf<T>::N;
    const int kc = (NP + chunk - 1) / chunk;
#define QPS_PASS(KC, R, RC) launch_pass<T, TH, KC, R, RC, 0>(st, check, G, A, ld, NP, MP, rpw, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb)
    // (row tile of the plain variant, row tile of the check variant): sized so that neither spills
    if (TH == 512) {
        if (kc <= 1) QPS_PASS(1, 4, 4);
        else if (kc <= 2) QPS_PASS(2, 4, 4);
        else if (kc <= 4) QPS_PASS(4, 4, 2);
        else QPS_PASS(8, 2, 1);
    } else {   // 1024 threads: 4 waves per SIMD, 128 VGPRs per lane
        if (kc <= 1) QPS_PASS(1, 4, 4);
        else if (kc <= 2) QPS_PASS(2, 4, 2);
        else if (kc <= 4) QPS_PASS(4, 2, 1);
        else QPS_PASS(8, 1, 1);
    }
#undef QPS_PASS
}

template <typename T>
void apass(hipStream_t st, bool check, const T* A, int64_t ld, int NP, int MP, const T* xx, const T* x_old, T* x_new, T* z,
           T* y, const T* l, const T* u, T alpha, T rho, T* part, T* part2, int64_t part_ld, unsigned long long* slots,
           PassBatch pb) {
    if (pass_threads_for<T>(NP) == 1024) apass_th<T, 1024>(st, check, A, ld, NP, MP, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
    else apass_th<T, 512>(st, check, A, ld, NP, MP, xx, x_old, x_new, z, y, l, u, alpha, rho, part, part2, part_ld, slots, pb);
}

#define INST(T)                                                                                                            \
    template int apass_max_np<T>();                                                                                        \
    template int apass_plan<T>(int, int, int*, int);                                                                       \
    template void apass<T>(hipStream_t, bool, const T*, int64_t, int, int, const T*, const T*, T*, T*, T*, const T*, const T*, \
                           T, T, T*, T*, int64_t, unsigned long long*, PassBatch);
INST(double)
INST(float)
#undef INST

}  // namespace qps
