// Ad-hoc micro-benchmark of the blocked-Cholesky launch chain (not part of the library): times back-to-back launches of the
// pieces of one 64-column step, to see what the ~37 us per step are made of.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../../quadraticprogramsolver_amd/csrc -I../../../include chol_chain.hip -o chol_chain
#define QPS_CHOL_TIMING 1
#include "../../../quadraticprogramsolver_amd/csrc/k_setup.hip"

#include <cmath>
#include <cstdio>
#include <vector>

using namespace qps;

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <typename F> static double time_chain(hipStream_t st, int reps, F&& f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 8; ++i) f();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3 * ms / reps;
}

template <typename T> static void run(const char* name) {
    const int NP = 4096;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<T> h((size_t)NP * NP);
    for (int i = 0; i < NP; ++i) for (int j = 0; j < NP; ++j) h[(size_t)i * NP + j] = (i == j) ? T(NP) : T(((i * 131 + j * 71) % 97) / 97.0 - 0.5);
    for (int i = 0; i < NP; ++i) for (int j = 0; j < i; ++j) h[(size_t)j * NP + i] = h[(size_t)i * NP + j];
    T *M, *M0, *dinv, *S, *tmp; int* fail;
    hipMalloc(&M, sizeof(T) * NP * NP); hipMalloc(&M0, sizeof(T) * NP * NP); hipMalloc(&S, sizeof(T) * NP * NP); hipMalloc(&tmp, sizeof(T) * NP * NP);
    hipMalloc(&dinv, sizeof(T) * NP * 64); hipMalloc(&fail, 64);
    hipMemcpy(M0, h.data(), sizeof(T) * NP * NP, hipMemcpyHostToDevice);
    hipMemcpy(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice);
    hipMemset(fail, 0, 64);
    printf("== %s\n", name);
    printf("empty kernel chain            %7.2f us\n", time_chain(st, 256, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr); }));
    printf("empty 2016 x 256 chain        %7.2f us\n", time_chain(st, 256, [&] { hipLaunchKernelGGL(k_empty, dim3(63, 32), dim3(256), 0, st, nullptr); }));
    printf("k_potrf64 chain               %7.2f us\n", time_chain(st, 256, [&] { hipLaunchKernelGGL((k_potrf64<T>), dim3(1), dim3(256), 0, st, M, (int64_t)NP, 0, fail, (int64_t)0); }));
    for (int wg : {1, 8, 32, 63})
        printf("k_trsm_panel %2d wgs           %7.2f us\n", wg, time_chain(st, 128, [&] { hipLaunchKernelGGL((k_trsm_panel<T>), dim3(wg, 1), dim3(256), 0, st, M, (int64_t)NP, 0, wg * 64, (int64_t)0); }));
    for (int g : {1, 4, 8, 16, 32, 48, 63})
        printf("k_update_potrf %2d x %2d         %7.2f us\n", g, g, time_chain(st, 64, [&] { hipLaunchKernelGGL((k_update_potrf<T>), dim3(g, g, 1), dim3(256), 0, st, M, (int64_t)NP, 0, fail, (int64_t)0); }));
    for (int g : {1, 8, 32, 63})
        printf("plain update (k_gemm) %2d x %2d  %7.2f us\n", g, g, time_chain(st, 64, [&] {
                   const T* A21 = M + (int64_t)64 * NP; T* A22 = M + (int64_t)64 * NP + 64;
                   gemm<T>(st, g * 64, g * 64, 64, T(-1), A21, NP, true, A21, NP, true, T(1), A22, NP, true, 1, 0, 0, 0, 0); }));
    printf("k_inv64 (64 blocks)           %7.2f us\n", time_chain(st, 64, [&] { hipLaunchKernelGGL((k_inv64<T>), dim3(64, 1), dim3(256), 0, st, M, (int64_t)NP, dinv, (int64_t)0, (int64_t)0); }));
    {   // numerical agreement of the two chains: lower triangle of L and the 64-block inverses
        std::vector<T> La((size_t)NP * NP), Lb((size_t)NP * NP), da((size_t)NP * 64), db((size_t)NP * 64);
        hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, 1); hipStreamSynchronize(st);
        hipMemcpy(La.data(), M, sizeof(T) * NP * NP, hipMemcpyDeviceToHost); hipMemcpy(da.data(), dinv, sizeof(T) * NP * 64, hipMemcpyDeviceToHost);
        hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, 1, S); hipStreamSynchronize(st);
        hipMemcpy(Lb.data(), M, sizeof(T) * NP * NP, hipMemcpyDeviceToHost); hipMemcpy(db.data(), dinv, sizeof(T) * NP * 64, hipMemcpyDeviceToHost);
        double eL = 0, mL = 0, eD = 0, mD = 0;
        for (int i = 0; i < NP; ++i) for (int j = 0; j <= i; ++j) { eL = std::max(eL, (double)std::fabs(La[(size_t)i * NP + j] - Lb[(size_t)i * NP + j])); mL = std::max(mL, (double)std::fabs(La[(size_t)i * NP + j])); }
        for (size_t k = 0; k < da.size(); ++k) { eD = std::max(eD, (double)std::fabs(da[k] - db[k])); mD = std::max(mD, (double)std::fabs(da[k])); }
        printf("128-column steps vs 64: max |dL| %.3e (max |L| %.3e), max |d dinv| %.3e (max %.3e)\n", eL, mL, eD, mD);
    }
    printf("cholesky 4096 (128 steps)     %7.2f us\n", time_chain(st, 4, [&] { hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, 1, S); }));
    for (int nrb : {0, 1, 16, 62})
        printf("k_chol_step nrb %2d            %7.2f us\n", nrb, time_chain(st, 64, [&] { hipLaunchKernelGGL((k_chol_step<T>), dim3(nrb > 0 ? nrb : 1, 1), dim3(256), 0, st, M0, (int64_t)NP, 0, 2, nrb, dinv, S, fail, (int64_t)0, (int64_t)0, (int64_t)0); }));
    {
        hipLaunchKernelGGL((k_chol_step<T>), dim3(1, 1), dim3(256), 0, st, M0, (int64_t)NP, 0, 2, 1, dinv, S, fail, (int64_t)0, (int64_t)0, (int64_t)0);
        hipStreamSynchronize(st);
        long long ck[16]; hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_chol_clock), sizeof(ck));
        const char* nm[] = {"potrf 0", "tri_inv 0", "L10", "D11 update", "potrf 1", "tri_inv 1", "(dinv store)", "panel"};
        for (int k = 0; k < 8; ++k) printf("   phase %-12s %6.2f us\n", nm[k], (ck[k + 1] - ck[k]) * 0.01);
        long long pk[16]; hipMemcpyFromSymbol(pk, HIP_SYMBOL(g_potrf_clock), sizeof(pk));
        printf("   tri_inv: 16 x 16 diagonal inverses %5.2f us, level 32 %5.2f us, level 64 %5.2f us\n", (pk[13] - pk[12]) * 0.01, (pk[14] - pk[13]) * 0.01, (pk[15] - pk[14]) * 0.01);
        for (int P = 0; P < 4; ++P) printf("   potrf panel %d: factorise %5.2f us, publish+barrier %5.2f us, update+barrier to next %5.2f us\n", P, (pk[3 * P + 1] - pk[3 * P]) * 0.01,
                                           (pk[3 * P + 2] - pk[3 * P + 1]) * 0.01, P < 3 ? (pk[3 * P + 3] - pk[3 * P + 2]) * 0.01 : 0.0);
    }
    for (int g : {1, 8, 32, 62})
        printf("plain update K=128 %2d x %2d     %7.2f us\n", g, g, time_chain(st, 64, [&] {
                   const T* A21 = M + (int64_t)128 * NP; T* A22 = M + (int64_t)128 * NP + 128;
                   gemm<T>(st, g * 64, g * 64, 128, T(-1), A21, NP, true, A21, NP, true, T(1), A22, NP, true, 1, 0, 0, 0, 0); }));
    {   // GEMM shapes of the refactorisation (QPS_GEMM_TILE=64 forces the 64-tile kernel): checksum + time
        auto checksum = [&](const T* d, size_t cnt) { std::vector<T> hh(cnt); hipMemcpy(hh.data(), d, sizeof(T) * cnt, hipMemcpyDeviceToHost); double a = 0; for (size_t k = 0; k < cnt; k += 7) a += (double)hh[k] * (double)((k % 13) + 1); return a; };
        hipMemcpy(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice);
        auto g1 = [&] { gemm<T>(st, 2048, 2048, 2048, T(1), M + (int64_t)2048 * NP, NP, true, M0, NP, false, T(0), tmp + (int64_t)2048 * NP, NP, false, 1, 0, 0, 0, 1); };
        g1(); hipStreamSynchronize(st);
        printf("gemm 2048^3 (L10 * W00, ktri 1)  checksum %.9e  %7.2f us\n", checksum(tmp + (int64_t)2048 * NP, (size_t)2048 * NP), time_chain(st, 8, g1));
        auto g2 = [&] { gemm<T>(st, 2048, 2048, 2048, T(-1), M0 + (int64_t)2048 * (NP + 1), NP, true, tmp + (int64_t)2048 * NP, NP, false, T(0), S + (int64_t)2048 * NP, NP, false, 1, 0, 0, 0, 2); };
        g2(); hipStreamSynchronize(st);
        printf("gemm 2048^3 (W11 * tmp, ktri 2)  checksum %.9e  %7.2f us\n", checksum(S + (int64_t)2048 * NP, (size_t)2048 * NP), time_chain(st, 8, g2));
        auto g3 = [&] { gemm<T>(st, NP, NP, NP, T(1), M0, NP, false, M0, NP, false, T(0), S, NP, true); };
        g3(); hipStreamSynchronize(st);
        printf("syrk 4096 x 4096 x 4096 (A'A form, lower) checksum %.9e  %7.2f us\n", checksum(S + (int64_t)3000 * NP, (size_t)64 * NP), time_chain(st, 4, g3));
    }
    printf("cholesky 4096                 %7.2f us\n", time_chain(st, 4, [&] { hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); cholesky<T>(st, NP, M, dinv, fail, 1); }));
    printf("  (copy alone                 %7.2f us)\n", time_chain(st, 4, [&] { hipMemcpyAsync(M, M0, sizeof(T) * NP * NP, hipMemcpyDeviceToDevice, st); }));
    printf("build_sweep_matrix 4096       %7.2f us\n", time_chain(st, 4, [&] { build_sweep_matrix<T>(st, NP, NP, M, dinv, S, tmp, 1); }));
    int hf[16]; hipMemcpy(hf, fail, 64, hipMemcpyDeviceToHost); printf("fail flag %d\n", hf[0]);
    hipFree(M); hipFree(M0); hipFree(S); hipFree(tmp); hipFree(dinv); hipFree(fail);
    hipStreamDestroy(st);
}

int main() {
    run<float>("fp32");
    run<double>("fp64");
    return 0;
}
