// Ad-hoc: the fp32 GEMM shapes of a re-factorisation at n = 4096 (inverse-doubling top level 2048^3 with a triangular operand, A'A-shaped SYRK), launched
// back to back for counter collection (rocprofv3 --pmc) and timing.  Not part of the library.
#include "../../../quadraticprogramsolver_amd/csrc/k_setup.hip"
#include <cstdio>
#include <vector>
using namespace qps;
namespace qps { int current_device() { int d = 0; (void)hipGetDevice(&d); return d; } }
template <typename T> static void run(int reps, const char* tname) {
    printf("== %s\n", tname);
    const int NP = 4096, MP = 8192;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    T *A, *L, *S, *tmp;
    hipMalloc(&A, sizeof(T) * (size_t)MP * NP); hipMalloc(&L, sizeof(T) * (size_t)NP * NP); hipMalloc(&S, sizeof(T) * (size_t)NP * NP); hipMalloc(&tmp, sizeof(T) * (size_t)NP * NP);
    std::vector<T> h((size_t)MP * NP);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (T)((i * 2654435761u) % 1000) / T(1000) - T(0.5);
    hipMemcpy(A, h.data(), sizeof(T) * (size_t)MP * NP, hipMemcpyHostToDevice);
    hipMemcpy(L, h.data(), sizeof(T) * (size_t)NP * NP, hipMemcpyHostToDevice); hipMemcpy(S, h.data() + 1000, sizeof(T) * (size_t)NP * NP, hipMemcpyHostToDevice);
    auto timeit = [&](const char* name, double flops, auto&& f) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        f(); hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f us  %6.1f TFLOP/s\n", name, 1e3 * ms / reps, flops / (1e-3 * ms / reps) / 1e12);
    };
    const int s = 2048;
    timeit("2048^3 L10 * W00 (TF, ktri 1)", 2.0 * s * s * s * 0.5, [&] { gemm<T>(st, s, s, s, T(1), L + (int64_t)s * NP, NP, true, S, NP, false, T(0), tmp + (int64_t)s * NP, NP, false, 1, 0, 0, 0, 1); });
    timeit("2048^3 W11 * tmp (TF, ktri 2)", 2.0 * s * s * s * 0.5, [&] { gemm<T>(st, s, s, s, -T(1), S + (int64_t)s * (NP + 1), NP, true, tmp + (int64_t)s * NP, NP, false, T(0), S + (int64_t)s * NP, NP, false, 1, 0, 0, 0, 2); });
    timeit("A'A 4096 x 4096 x 8192 (FF, lower)", 2.0 * NP * NP * MP * 0.5, [&] { gemm<T>(st, NP, NP, MP, T(1), A, NP, false, A, NP, false, T(0), S, NP, true); });
    timeit("2048^3 plain (TT)", 2.0 * s * s * s, [&] { gemm<T>(st, s, s, s, T(1), L, NP, true, S, NP, true, T(0), tmp, NP, false); });
    hipFree(A); hipFree(L); hipFree(S); hipFree(tmp); hipStreamDestroy(st);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    run<float>(reps, "fp32");
    run<double>(reps, "fp64");
    return 0;
}
