// Microbenchmark: steady-state rate of the f64 MFMA tile engine used by the factorisation's bulk tasks
//   C(tile) -= sum over nkb 64-blocks  L_kb R_kb^T
// V0: 64x64 output tile per 256-thread workgroup (wave = 16 rows x 64 columns)      -- what csrc/chol.hip does
// V1: 128x64 output tile per workgroup (wave = 32 rows x 64 columns: 6 LDS fragment reads per 8 MFMAs instead of 5 per 4)
// build: hipcc --offload-arch=gfx950 -O3 -I../../dgp_amd/csrc tile_engine.hip -o tile_engine
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "tile.hpp"

__global__ __launch_bounds__(256, 2) void v0_kernel(double *A, int64_t ld, int64_t stride, const int2 *tiles, int ntiles,
                                                     int nkb) {
    __shared__ double As[64 * LDM], Bs[64 * LDM];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x / ntiles;
    const int2 tk = tiles[blockIdx.x % ntiles];
    double *M = A + (int64_t)b * stride;
    double *C = M + (int64_t)tk.x * 64 * ld + (int64_t)tk.y * 64;
    const double *Lp = M + (int64_t)tk.x * 64 * ld, *Rp = M + (int64_t)tk.y * 64 * ld;
    const int c2 = (tid & 15) * 2, r0 = tid >> 4;
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    d4 acc[4];
    double2 pa[4], pb[4];
    const int nh = 2 * nkb;
    auto fetch = [&](int hh) {
        const int64_t off = 32 * hh;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            pa[it] = *reinterpret_cast<const double2 *>(Lp + (int64_t)(r0 + 16 * it) * ld + off + c2);
            pb[it] = *reinterpret_cast<const double2 *>(Rp + (int64_t)(r0 + 16 * it) * ld + off + c2);
        }
    };
    fetch(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol];
    for (int hh = 0; hh < nh; ++hh) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = r0 + 16 * it;
            As[r * LDM + c2] = pa[it].x; As[r * LDM + c2 + 1] = pa[it].y;
            Bs[r * LDM + c2] = pb[it].x; Bs[r * LDM + c2 + 1] = pb[it].y;
        }
        if (hh + 1 < nh) fetch(hh + 1);
        __syncthreads();
        mfma_tile<OP_MK, OP_MK>(As, Bs, acc, wave, lane, -1.0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol] = acc[t][r];
}

// 128x64: tiles[].x is the first of two vertically adjacent 64-row blocks
template <int WGS>
__global__ __launch_bounds__(256, WGS) void v1_kernel(double *A, int64_t ld, int64_t stride, const int2 *tiles, int ntiles,
                                                       int nkb) {
    __shared__ double As[128 * LDM], Bs[64 * LDM];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x / ntiles;
    const int2 tk = tiles[blockIdx.x % ntiles];
    double *M = A + (int64_t)b * stride;
    double *C = M + (int64_t)tk.x * 64 * ld + (int64_t)tk.y * 64;
    const double *Lp = M + (int64_t)tk.x * 64 * ld, *Rp = M + (int64_t)tk.y * 64 * ld;
    const int c2 = (tid & 15) * 2, r0 = tid >> 4;
    const int m = lane & 15, kk = lane >> 4;
    d4 acc[2][4];
    double2 pa[8], pb[4];
    const int nh = 2 * nkb;
    auto fetch = [&](int hh) {
        const int64_t off = 32 * hh;
#pragma unroll
        for (int it = 0; it < 8; ++it) pa[it] = *reinterpret_cast<const double2 *>(Lp + (int64_t)(r0 + 16 * it) * ld + off + c2);
#pragma unroll
        for (int it = 0; it < 4; ++it) pb[it] = *reinterpret_cast<const double2 *>(Rp + (int64_t)(r0 + 16 * it) * ld + off + c2);
    };
    fetch(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[h][t][r] = C[(int64_t)(32 * wave + 16 * h + kk + 4 * r) * ld + 16 * t + m];
    for (int hh = 0; hh < nh; ++hh) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = r0 + 16 * it;
            As[r * LDM + c2] = pa[it].x; As[r * LDM + c2 + 1] = pa[it].y;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = r0 + 16 * it;
            Bs[r * LDM + c2] = pb[it].x; Bs[r * LDM + c2 + 1] = pb[it].y;
        }
        if (hh + 1 < nh) fetch(hh + 1);
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += 4) {
            const double a0 = -As[(32 * wave + m) * LDM + k0 + kk], a1 = -As[(32 * wave + 16 + m) * LDM + k0 + kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = Bs[(16 * t + m) * LDM + k0 + kk];
                acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(int64_t)(32 * wave + 16 * h + kk + 4 * r) * ld + 16 * t + m] = acc[h][t][r];
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int Np = 2048, nb = Np / 64, B = argc > 1 ? atoi(argv[1]) : 12;
    double *A;
    (void)hipMalloc(&A, (size_t)B * Np * Np * 8);
    {   // random data (all-zero operands toggle nothing and overstate the clock)
        std::vector<double> h((size_t)Np * Np);
        for (auto &v : h) v = 1e-3 * ((double)rand() / RAND_MAX - 0.5);
        for (int b = 0; b < B; ++b) (void)hipMemcpy(A + (size_t)b * Np * Np, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    // lower-triangle tiles below block row/column 4 .. nb (so that 4..8 panels to the left exist)
    for (int nkb = 4; nkb <= 16; nkb *= 2) {
        std::vector<int2> t0, t1;
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j <= i; ++j) t0.push_back(make_int2(i, j));
        for (int i = 0; i + 1 < nb; i += 2)
            for (int j = 0; j <= i; ++j) t1.push_back(make_int2(i, j));
        int2 *d0, *d1;
        (void)hipMalloc(&d0, t0.size() * sizeof(int2)); (void)hipMalloc(&d1, t1.size() * sizeof(int2));
        (void)hipMemcpy(d0, t0.data(), t0.size() * sizeof(int2), hipMemcpyHostToDevice);
        (void)hipMemcpy(d1, t1.data(), t1.size() * sizeof(int2), hipMemcpyHostToDevice);
        const int64_t ld = Np, stride = (int64_t)Np * Np;
        float ms = 0;
        for (int dyn = 0; dyn <= 45000; dyn += 22500) {   // extra dynamic LDS caps the occupancy at 4 / 2 / 2 workgroups per CU
            ms = timeit([&] { hipLaunchKernelGGL(v0_kernel, dim3(B * t0.size()), dim3(256), dyn, 0, A, ld, stride, d0, (int)t0.size(), nkb); }, 5);
            printf("B=%d nkb=%2d  V0  64x64 (+%d B LDS): %5d WGs %.3f ms  %.1f TFLOP/s\n", B, nkb, dyn, (int)(B * t0.size()), ms,
                   (double)B * t0.size() * nkb * 2.0 * 64 * 64 * 64 / ms / 1e9);
        }
        ms = timeit([&] { hipLaunchKernelGGL(v1_kernel<2>, dim3(B * t1.size()), dim3(256), 0, 0, A, ld, stride, d1, (int)t1.size(), nkb); }, 5);
        printf("B=%d nkb=%2d  V1 128x64 (2 WG/CU): %5d WGs %.3f ms  %.1f TFLOP/s\n", B, nkb, (int)(B * t1.size()), ms,
               (double)B * t1.size() * nkb * 2.0 * 128 * 64 * 64 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(v1_kernel<3>, dim3(B * t1.size()), dim3(256), 0, 0, A, ld, stride, d1, (int)t1.size(), nkb); }, 5);
        printf("B=%d nkb=%2d  V1 128x64 (3 WG/CU): %5d WGs %.3f ms  %.1f TFLOP/s\n", B, nkb, (int)(B * t1.size()), ms,
               (double)B * t1.size() * nkb * 2.0 * 128 * 64 * 64 / ms / 1e9);
    }
    return 0;
}
