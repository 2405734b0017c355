// Do f64 MFMA and f64 VALU instructions overlap on gfx950?  Three kernels with the same launch shape (every CU, W waves per
// SIMD): NM independent MFMAs per iteration, NF independent FMAs per iteration, and both in one loop body (the compiler is
// free to interleave them).  If the matrix and the vector pipes were separate the mixed kernel would take max(t_mfma, t_fma);
// it takes their sum.  A fourth launch runs MFMA-only and FMA-only WORKGROUPS side by side on every CU (different waves of a
// SIMD): the same.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_shared tools/ubench/f64_shared_units.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NF>
__global__ __launch_bounds__(256) void mixed(double *out, int iters, double a0, double b0, int split) {
    d4 acc[NM > 0 ? NM : 1];
    double f[NF > 0 ? NF : 1];
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int i = 0; i < (NF > 0 ? NF : 1); ++i) f[i] = threadIdx.x * 1e-9 + i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    // split != 0: even workgroups do the MFMAs only, odd ones the FMAs only (side by side on a CU)
    const bool do_m = !split || (blockIdx.x & 1) == 0, do_f = !split || (blockIdx.x & 1) == 1;
    for (int it = 0; it < iters; ++it) {
        if (NM > 0 && do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (NF > 0 && do_f) {
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i] = fma(f[i], a0, b0);
        }
    }
    double s = 0;
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < (NF > 0 ? NF : 1); ++i) s += f[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char **argv) {
    double *out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(double));
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    constexpr int NM = 4, NF = 64;   // per iteration: 4 MFMAs (4 x 64 issue cycles) and 64 FMAs (64 x 4 issue cycles): equal work at the datasheet rates
    for (int w = 1; w <= 2; ++w) {
        const int grid = 256 * w;
        const float tm = timeit([&] { hipLaunchKernelGGL((mixed<NM, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3, 0); });
        const float tf = timeit([&] { hipLaunchKernelGGL((mixed<0, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999999, 1e-3, 0); });
        const float tb = timeit([&] { hipLaunchKernelGGL((mixed<NM, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999999, 1e-3, 0); });
        const double fl_m = (double)grid * 4 * iters * NM * 2048.0, fl_f = (double)grid * 256 * (double)iters * NF * 2.0;
        printf("%d wave(s) per SIMD:  MFMA only %.3f ms (%.1f TFLOP/s)   FMA only %.3f ms (%.1f TFLOP/s)   both in one loop %.3f ms  = %.2f x (MFMA + FMA), %.2f x max\n",
               w, tm, fl_m / tm / 1e9, tf, fl_f / tf / 1e9, tb, tb / (tm + tf), tb / (tm > tf ? tm : tf));
    }
    {   // MFMA workgroups and FMA workgroups side by side: 2 workgroups per CU, one of each kind
        const int grid = 512;
        const float tm = timeit([&] { hipLaunchKernelGGL((mixed<NM, 0>), dim3(256), dim3(256), 0, 0, out, iters, 1.0, 1e-3, 0); });
        const float tf = timeit([&] { hipLaunchKernelGGL((mixed<0, NF>), dim3(256), dim3(256), 0, 0, out, iters, 0.999999, 1e-3, 0); });
        const float ts = timeit([&] { hipLaunchKernelGGL((mixed<NM, NF>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999999, 1e-3, 1); });
        printf("side by side (one MFMA and one FMA workgroup per CU): alone %.3f / %.3f ms, together %.3f ms = %.2f x (MFMA + FMA), %.2f x max\n",
               tm, tf, ts, ts / (tm + tf), ts / (tm > tf ? tm : tf));
    }
    return 0;
}
