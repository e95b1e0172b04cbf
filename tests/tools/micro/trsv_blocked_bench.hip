// Ad-hoc micro-benchmark of the single-launch blocked sweep (not part of the library): the kernel alone on a synthetic sweep matrix,
// checked against the host recurrence, timed in its normal form and with the hand-off waits / the matrix stream switched off.
//   tests/tools/micro/build.sh trsv_blocked_bench && tests/tools/micro/trsv_blocked_bench [n] [nb]
#include "../../../quadraticprogramsolver_amd/csrc/k_trsv_blocked.hip"

#include <cmath>
#include <cstdio>
#include <vector>

namespace qps { thread_local LaunchTiming g_launch_timing; }
using namespace qps;

template <typename T> static void run(const char* name, int NP, int nb) {
    if (!trsv_blocked_supported<T>(NP, nb)) { printf("%s n=%d nb=%d: unsupported\n", name, NP, nb); return; }
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<T> S((size_t)NP * NP), t(NP);
    unsigned rs = 12345u;
    auto rnd = [&] { rs = rs * 1664525u + 1013904223u; return (double)(rs >> 8) / (1 << 24) - 0.5; };
    for (auto& x : S) x = (T)(rnd() * 2.0 / NP * 8);
    for (int i = 0; i < NP; ++i) { S[(size_t)i * NP + i] = (T)(1.0 + 0.1 * rnd()); t[i] = (T)rnd(); }
    T *dS, *dt, *dy; unsigned long long* pub; unsigned* ab;
    hipMalloc(&dS, sizeof(T) * S.size()); hipMalloc(&dt, sizeof(T) * NP); hipMalloc(&dy, sizeof(T) * NP);
    hipMalloc(&pub, 8 * trsv_blocked_pub_words<T>(NP)); hipMalloc(&ab, 16);
    hipMemcpy(dS, S.data(), sizeof(T) * S.size(), hipMemcpyHostToDevice); hipMemcpy(dt, t.data(), sizeof(T) * NP, hipMemcpyHostToDevice);
    hipMemset(pub, 0, 8 * trsv_blocked_pub_words<T>(NP)); hipMemset(ab, 0, 16); hipMemset(dy, 0, sizeof(T) * NP);
    unsigned epoch = 0;
    const int nblk = (NP + nb - 1) / nb;
    for (int bwd = 0; bwd < 2; ++bwd) {
        trsv_blocked<T>(st, bwd != 0, dS, NP, NP, nb, dt, dy, pub, ++epoch, ab);
        hipStreamSynchronize(st);
        std::vector<T> y(NP); unsigned abh = 0;
        hipMemcpy(y.data(), dy, sizeof(T) * NP, hipMemcpyDeviceToHost); hipMemcpy(&abh, ab, 4, hipMemcpyDeviceToHost);
        std::vector<double> ref(NP);
        double err = 0, mx = 0;
        for (int jj = 0; jj < nblk; ++jj) {
            const int J = bwd ? nblk - 1 - jj : jj;
            for (int r = J * nb; r < std::min(NP, (J + 1) * nb); ++r) {
                double s = 0;
                if (!bwd) for (int c = 0; c <= r; ++c) s += (double)S[(size_t)r * NP + c] * (c >= J * nb ? (double)t[c] : ref[c]);
                else for (int c = r; c < NP; ++c) s += (double)S[(size_t)r * NP + c] * (c < (J + 1) * nb ? (double)t[c] : ref[c]);
                ref[r] = s;
            }
        }
        for (int r = 0; r < NP; ++r) { err = std::max(err, std::fabs(ref[r] - (double)y[r])); mx = std::max(mx, std::fabs(ref[r])); }
        printf("%s n=%d nb=%d %s: max err %.2e (max |y| %.2e) abort=%u\n", name, NP, nb, bwd ? "backward" : "forward", err, mx, abh);
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 5; ++i) trsv_blocked<T>(st, bwd != 0, dS, NP, NP, nb, dt, dy, pub, ++epoch, ab, mode);
            hipStreamSynchronize(st);
            const int reps = 50;
            hipEventRecord(e0, st);
            for (int i = 0; i < reps; ++i) trsv_blocked<T>(st, bwd != 0, dS, NP, NP, nb, dt, dy, pub, ++epoch, ab, mode);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / reps, bytes = sizeof(T) * ((double)NP * (NP + 1) / 2 + 2.0 * NP);
            printf("    mode %d (%s): %.1f us per launch back to back, %.2f TB/s of the triangle\n", mode,
                   mode == 0 ? "normal" : mode == 1 ? "no hand-off wait" : mode == 2 ? "no matrix stream" : "neither", us, bytes / us / 1e6);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    hipFree(dS); hipFree(dt); hipFree(dy); hipFree(pub); hipFree(ab); hipStreamDestroy(st);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int nb = argc > 2 ? atoi(argv[2]) : 0;
    if (nb) { run<double>("fp64", n, nb); run<float>("fp32", n, nb); }
    else { run<double>("fp64", n, 1024); run<double>("fp64", n, 512); run<float>("fp32", n, 2048); run<float>("fp32", n, 1024); }
    return 0;
}
