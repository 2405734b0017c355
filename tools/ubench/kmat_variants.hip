// K-assembly store/compute variants, timed with HIP events (n x n full symmetric matrix from lower 64x64 tiles).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/kmat_variants tools/ubench/kmat_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline void tri_decode(int t, int &bi, int &bj) {
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    bi = i; bj = t - i * (i + 1) / 2;
}

// flags: 1 = compute distances, 2 = exp, 4 = mirrored store (naive), 8 = mirrored via LDS transpose, 16 = paired-column mapping + 16B stores,
//        32 = row-major sweep of all tiles (no symmetry: every WG computes one tile, no mirrored store)
template <int F>
__global__ __launch_bounds__(256) void kvar(const double *X, int n, int D, double *K, int nb) {
    extern __shared__ double lds[];
    double *XiT = lds, *XjT = lds + D * 64, *TT = lds + 2 * D * 64;
    int bi, bj;
    if (F & 32) { bi = blockIdx.x / nb; bj = blockIdx.x % nb; } else tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64;
    for (int idx = tid; idx < 64 * D; idx += 256) {
        int row = idx / D, d = idx - row * D;
        XiT[d * 64 + row] = X[(i0 + row) * D + d];
        XjT[d * 64 + row] = X[(j0 + row) * D + d];
    }
    __syncthreads();
    int cq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = (F & 16) ? (2 * tx + (q & 1) + 32 * (q >> 1)) : (tx + 16 * q);
    double s[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[p][q] = 0.0;
    if (F & 1)
        for (int d = 0; d < D; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) { xi[p] = XiT[d * 64 + ty + 16 * p]; xj[p] = XjT[d * 64 + cq[p]]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) { double df = xi[p] - xj[q]; s[p][q] = fma(df, df, s[p][q]); }
        }
    double v[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double val = (F & 2) ? exp(-s[p][q]) : 1.0 - s[p][q];
            if (i0 + ty + 16 * p == j0 + cq[q]) val = 1.0 + 1e-8;
            v[p][q] = val;
        }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double *row = K + (i0 + ty + 16 * p) * n + j0;
        if (F & 16) {
            *reinterpret_cast<double2 *>(row + cq[0]) = make_double2(v[p][0], v[p][1]);
            *reinterpret_cast<double2 *>(row + cq[2]) = make_double2(v[p][2], v[p][3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) row[cq[q]] = v[p][q];
        }
    }
    if (bi == bj) return;
    if (F & 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) K[(j0 + cq[q]) * n + i0 + ty + 16 * p] = v[p][q];
    }
    if (F & 8) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) TT[(ty + 16 * p) * 65 + cq[q]] = v[p][q];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double *row = K + (j0 + ty + 16 * p) * n + i0;
            if (F & 16) {
                *reinterpret_cast<double2 *>(row + cq[0]) = make_double2(TT[cq[0] * 65 + ty + 16 * p], TT[cq[1] * 65 + ty + 16 * p]);
                *reinterpret_cast<double2 *>(row + cq[2]) = make_double2(TT[cq[2] * 65 + ty + 16 * p], TT[cq[3] * 65 + ty + 16 * p]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) row[cq[q]] = TT[cq[q] * 65 + ty + 16 * p];
            }
        }
    }
}

template <int F>
int run(const char *name, const double *X, int n, int D, double *K) {
    int nb = n / 64, nt = (F & 32) ? nb * nb : nb * (nb + 1) / 2;
    size_t shm = (2 * D * 64 + 64 * 65) * sizeof(double);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) kvar<F><<<nt, 256, shm>>>(X, n, D, K, nb);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) kvar<F><<<nt, 256, shm>>>(X, n, D, K, nb);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    double bytes = (F & (4 | 8 | 32)) ? 8.0 * n * n : 4.0 * n * (n + 64);
    printf("%-46s n=%5d D=%2d  %7.3f ms  %6.0f GB/s written\n", name, n, D, ms, bytes / ms / 1e6);
    return 0;
}

int main() {
    for (int n : {8192, 16384}) for (int D : {5}) {
        std::vector<double> h((size_t)n * D);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 10007) / 10007.0;
        double *X, *K;
        CK(hipMalloc(&X, h.size() * 8)); CK(hipMalloc(&K, (size_t)n * n * 8));
        CK(hipMemcpy(X, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        run<1 | 2 | 4>("current: dist+exp, naive mirrored", X, n, D, K);
        run<1 | 4>("dist, no exp, naive mirrored", X, n, D, K);
        run<4>("stores only, naive mirrored", X, n, D, K);
        run<8>("stores only, LDS-transposed mirrored", X, n, D, K);
        run<8 | 16>("stores only, LDS-transposed, 16B", X, n, D, K);
        run<0>("stores only, lower tiles only", X, n, D, K);
        run<16>("stores only, lower tiles only, 16B", X, n, D, K);
        run<1 | 2>("dist+exp, lower tiles only", X, n, D, K);
        run<1 | 2 | 8>("dist+exp, LDS-transposed mirrored", X, n, D, K);
        run<1 | 2 | 8 | 16>("dist+exp, LDS-transposed, 16B", X, n, D, K);
        run<32>("stores only, all tiles row-major sweep", X, n, D, K);
        run<32 | 16>("stores only, all tiles row-major sweep, 16B", X, n, D, K);
        run<1 | 2 | 32>("dist+exp, all tiles row-major (no symmetry)", X, n, D, K);
        run<1 | 2 | 32 | 16>("dist+exp, all tiles row-major, 16B", X, n, D, K);
        CK(hipFree(X)); CK(hipFree(K));
    }
    return 0;
}
