// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (independent accumulators, operands in registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = d4{0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        k<4><<<grid, 256>>>(d, 100); hipDeviceSynchronize();
        hipEventRecord(a); k<4><<<grid, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 2048.0;
        printf("f64 mfma 16x16x4: %d WG/CU x 4 waves, 4 accumulators: %.1f TFLOP/s (%.3f ms)\n", wgs_per_cu, flops / ms / 1e9, ms);
    }
    return 0;
}
