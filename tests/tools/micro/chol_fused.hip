// Ad-hoc micro-benchmark of the round-4 Cholesky chain (diagonal block factorised inside the trailing-update launch): numerics against the
// 64-column chain at several sizes, per-launch times, whole factorisation with QPS_CHOL_FUSED = 0 / 1.  Not part of the library.
//   bash tests/tools/micro/build.sh chol_fused   (add -DQPS_DIAG_PARK32=1 through EXTRA)
#define QPS_CHOL_TIMING 1
#include "../../../quadraticprogramsolver_amd/csrc/k_setup.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace qps;
namespace qps { int current_device() { int d = 0; (void)hipGetDevice(&d); return d; } }

template <typename F> static double time_chain(hipStream_t st, int reps, F&& f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3 * ms / reps;
}

template <typename T> static void fill_spd(std::vector<T>& h, int NP) {
    for (int i = 0; i < NP; ++i) for (int j = 0; j <= i; ++j) {
        const T v = (i == j) ? T(NP) : T(((i * 131 + j * 71) % 97) / 97.0 - 0.5);
        h[(size_t)i * NP + j] = v; h[(size_t)j * NP + i] = v;
    }
}

template <typename T> static void numerics(hipStream_t st, int NP, int batch) {
    const size_t nn = (size_t)NP * NP;
    std::vector<T> h(nn * batch);
    { std::vector<T> one(nn); fill_spd(one, NP); for (int b = 0; b < batch; ++b) { for (size_t k = 0; k < nn; ++k) h[b * nn + k] = one[k]; for (int i = 0; i < NP; ++i) h[b * nn + (size_t)i * NP + i] += T(b); } }
    T *M, *M0, *dinv, *S; int* fail;
    hipMalloc(&M, sizeof(T) * nn * batch); hipMalloc(&M0, sizeof(T) * nn * batch); hipMalloc(&S, sizeof(T) * nn * batch);
    hipMalloc(&dinv, sizeof(T) * NP * 64 * batch); hipMalloc(&fail, 4 * batch);
    hipMemcpy(M0, h.data(), sizeof(T) * nn * batch, hipMemcpyHostToDevice);
    std::vector<T> La(nn * batch), Lb(nn * batch), da((size_t)NP * 64 * batch), db((size_t)NP * 64 * batch);
    hipMemcpyAsync(M, M0, sizeof(T) * nn * batch, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, batch); hipStreamSynchronize(st);
    hipMemcpy(La.data(), M, sizeof(T) * nn * batch, hipMemcpyDeviceToHost); hipMemcpy(da.data(), dinv, sizeof(T) * da.size(), hipMemcpyDeviceToHost);
    setenv("QPS_CHOL_FUSED", "1", 1);
    hipMemcpyAsync(M, M0, sizeof(T) * nn * batch, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, batch, chol_scratch_fits(NP) ? S : nullptr); hipStreamSynchronize(st);
    hipMemcpy(Lb.data(), M, sizeof(T) * nn * batch, hipMemcpyDeviceToHost); hipMemcpy(db.data(), dinv, sizeof(T) * db.size(), hipMemcpyDeviceToHost);
    std::vector<int> hf(batch); hipMemcpy(hf.data(), fail, 4 * batch, hipMemcpyDeviceToHost);
    double eL = 0, mL = 0, eD = 0, mD = 0;
    for (int b = 0; b < batch; ++b) for (int i = 0; i < NP; ++i) for (int j = 0; j <= i; ++j) { const size_t k = b * nn + (size_t)i * NP + j; eL = std::max(eL, (double)std::fabs(La[k] - Lb[k])); mL = std::max(mL, (double)std::fabs(La[k])); }
    for (size_t k = 0; k < da.size(); ++k) { eD = std::max(eD, (double)std::fabs(da[k] - db[k])); mD = std::max(mD, (double)std::fabs(da[k])); }
    printf("  NP %4d batch %2d: fused vs 64-column chain: max |dL| %.3e (max |L| %.3e), max |d dinv| %.3e (max %.3e), fail %d %s\n", NP, batch, eL, mL, eD, mD, hf[0],
           (eL <= (sizeof(T) == 8 ? 1e-11 : 2e-3) * mL && !(eL != eL)) ? "ok" : "MISMATCH");
    hipFree(M); hipFree(M0); hipFree(S); hipFree(dinv); hipFree(fail);
}

template <typename T> static void run(const char* name) {
    const int NP = 4096;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    printf("== %s\n", name);
    for (int np : {64, 128, 192, 256, 1024, 4032, 4096}) numerics<T>(st, np, 1);
    numerics<T>(st, 1024, 3); numerics<T>(st, 192, 5);
    std::vector<T> h((size_t)NP * NP); fill_spd(h, NP);
    T *M, *M0, *dinv, *S, *tmp; int* fail;
    hipMalloc(&M, sizeof(T) * NP * NP); hipMalloc(&M0, sizeof(T) * NP * NP); hipMalloc(&S, sizeof(T) * NP * NP); hipMalloc(&tmp, sizeof(T) * NP * NP);
    hipMalloc(&dinv, sizeof(T) * NP * 64); hipMalloc(&fail, 64);
    hipMemcpy(M0, h.data(), sizeof(T) * NP * NP, hipMemcpyHostToDevice);
    hipMemcpy(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice);
    const int64_t z = 0, roff = chol_scratch_elems(NP) - 64;
    printf("k_chol_update_diag K=0 1x1     %7.2f us\n", time_chain(st, 64, [&] { hipLaunchKernelGGL((k_chol_update_diag<T>), dim3(1, 1), dim3(256), 0, st, M0, (int64_t)NP, 0, 0, 64, 0, 0, dinv, S, roff, fail, z, z, z); }));
    auto launch_ud = [&](int g, int avoid) {
        const int nt = g * (g + 1) / 2 - 1, ids = 1 + nt + (avoid ? nt / 7 + 2 : 0);
        hipLaunchKernelGGL((k_chol_update_diag<T>), dim3(ids, 1), dim3(256), 0, st, M, (int64_t)NP, 2, 128, 64, g, avoid, dinv, S, roff, fail, z, z, z);
    };
    for (int g : {2, 4, 8, 16, 24, 32, 48, 62}) {
        const double tav = time_chain(st, 64, [&] { launch_ud(g, 1); });
        const double tf = time_chain(st, 64, [&] { launch_ud(g, 0); });
        const double tg = time_chain(st, 64, [&] { const T* A21 = M + (int64_t)128 * NP; T* A22 = M + (int64_t)128 * NP + 128;
                                                   gemm<T>(st, g * 64, g * 64, 128, T(-1), A21, NP, true, A21, NP, true, T(1), A22, NP, true, 1, 0, 0, 0, 0); });
        const double tp = time_chain(st, 64, [&] { hipLaunchKernelGGL((k_chol_panel<T>), dim3(g, 1), dim3(256), 0, st, M, (int64_t)NP, 0, dinv, S, z, z, z); });
        const double ts = time_chain(st, 64, [&] { hipLaunchKernelGGL((k_chol_step<T>), dim3(g, 1), dim3(256), 0, st, M, (int64_t)NP, 0, 2, g, dinv, S, fail, z, z, z); });
        {
            launch_ud(g, 0);
            hipStreamSynchronize(st);
            long long ck[16]; hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_chol_clock), sizeof(ck));
            printf("   diag phases (us): tile00 %.2f potrf0 %.2f store+wait %.2f inv0 %.2f L10 %.2f D11 %.2f potrf1 %.2f inv1 %.2f store %.2f | total %.2f\n", (ck[1] - ck[0]) * 0.01, (ck[2] - ck[1]) * 0.01,
                   (ck[3] - ck[2]) * 0.01, (ck[4] - ck[3]) * 0.01, (ck[5] - ck[4]) * 0.01, (ck[6] - ck[5]) * 0.01, (ck[7] - ck[6]) * 0.01, (ck[8] - ck[7]) * 0.01, (ck[9] - ck[8]) * 0.01, (ck[9] - ck[0]) * 0.01);
        }
        printf("grid %2d: update+diag %7.2f us (XCD left alone: %7.2f us) | plain update %7.2f us | panel only %7.2f us | old step %7.2f us\n", g, tf, tav, tg, tp, ts);
    }
    for (const char* mode : {"0", "1", "2"}) {
        setenv("QPS_CHOL_FUSED", mode[0] == '0' ? "0" : "1", 1);
        setenv("QPS_CHOL_AVOID", mode[0] == '2' ? "0" : "256", 1);
        printf("cholesky 4096 mode %s (0: round-3 chain, 1: fused, 2: fused, XCD not left alone)  %7.2f us (incl. %7.2f us copy)\n", mode,
               time_chain(st, 6, [&] { hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, 1, S); }),
               time_chain(st, 6, [&] { hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); }));
    }
    int hf[16]; hipMemcpy(hf, fail, 64, hipMemcpyDeviceToHost); printf("fail flag %d\n", hf[0]);
    hipFree(M); hipFree(M0); hipFree(S); hipFree(tmp); hipFree(dinv); hipFree(fail);
    hipStreamDestroy(st);
}

int main() {
    run<float>("fp32");
    run<double>("fp64");
    return 0;
}
