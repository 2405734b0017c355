// Microbenchmark: sustained f64 MFMA (v_mfma_f64_16x16x4_f64) and f64 FMA (v_fma_f64) rates on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_loop(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double a = a0, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv) {
    double *out;
    (void)hipMalloc(&out, 256 * 4096 * 256 * sizeof(double));
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        int grid = 256 * wg_per_cu;
        float ms = timeit([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        double flops = (double)grid * 4 /*waves*/ * iters * 4 * 2048.0;
        printf("MFMA f64 16x16x4, %d WG/CU (=%d waves/SIMD), 4 acc: %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wg_per_cu, wg_per_cu,
               flops / ms / 1e9, 2.4e9 * (ms * 1e-3) / ((double)iters * 4 * wg_per_cu));
        ms = timeit([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        printf("   same, 1 dependent accumulator: %.2f TFLOP/s (%.1f cycles per dependent MFMA)\n", (double)grid * 4 * iters * 2048.0 / ms / 1e9,
               2.4e9 * (ms * 1e-3) / ((double)iters * wg_per_cu));
        ms = timeit([&] { hipLaunchKernelGGL(fma_loop<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3); });
        printf("v_fma_f64, %d waves/SIMD, 16 independent: %.2f TFLOP/s (%.2f cycles/instr/SIMD)\n", wg_per_cu,
               (double)grid * 256 * iters * 16 * 2.0 / ms / 1e9, 2.4e9 * (ms * 1e-3) / ((double)iters * 16 * wg_per_cu));
        ms = timeit([&] { hipLaunchKernelGGL(fma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3); });
        printf("   dependent chain: %.2f cycles per dependent v_fma_f64\n", 2.4e9 * (ms * 1e-3) / ((double)iters));
    }
    return 0;
}
