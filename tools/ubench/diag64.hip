// Stand-alone check + timing of the 64x64 diagonal-block factorisation (dgp_amd/csrc/diagfac.hpp) on one workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_ubench/diag64 tools/ubench/diag64.hip
#include "../../dgp_amd/csrc/diagfac.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// column-block layout from the LOWER triangle of a row-major tile
__device__ __forceinline__ void load_cb_lower(d4 (&X)[4], const double *C, int64_t ld, int w, int lm, int lu) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = 16 * t + lu + 4 * r, Cc = 16 * w + lm;
            X[t][r] = (t <= w) ? (R >= Cc ? C[(int64_t)R * ld + Cc] : C[(int64_t)Cc * ld + R]) : 0.0;
        }
}

__global__ __launch_bounds__(256, 2) void k_diag(const double *A, double *L, double *W, int ncol, int reps, long long *t, int *bad,
                                                 long long *stamps) {
    __shared__ DiagShared sh;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    long long tot = 0, tot_c = 0;
    int b = 0;
    for (int rep = 0; rep < reps; ++rep) {
        Tile64 tile;
        load_cb_lower(tile.v, A, 64, w, l & 15, l >> 4);
        __syncthreads();
        int nc = ncol;
        asm volatile("" : "+s"(nc));   // keep the compiler from hoisting the per-pivot conditions out of the timing loop
        const long long t0 = wall_clock64(), c0 = clock64();
        Tile64 winv;
        b = diag_factor(tile, winv, sh, nc, rep == reps - 1 ? stamps : nullptr);
        diag_store_inverse<false>(winv, W);
        diag_store_factor(tile, L, 64, nc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tot += wall_clock64() - t0;
        tot_c += clock64() - c0;
    }
    if (tid == 0) { t[0] = tot; t[1] = tot_c; bad[0] = b; }
}

int main() {
    const int N = 64;
    std::vector<double> A(N * N), Asym(N * N);
    srand(7);
    std::vector<double> B(N * N);
    for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < N; ++k) s += B[i * N + k] * B[j * N + k];
            Asym[i * N + j] = s + (i == j ? 2.0 : 0.0);
        }
    double *dA, *dL, *dW; long long *dt, *dst; int *dbad;
    CK(hipMalloc(&dA, N * N * 8)); CK(hipMalloc(&dL, N * N * 8)); CK(hipMalloc(&dW, N * N * 8));
    CK(hipMalloc(&dt, 16)); CK(hipMalloc(&dbad, 4)); CK(hipMalloc(&dst, 128));
    const int ncols[] = {64, 16, 37, 1, 48, 63, 0};
    int fails = 0;
    for (int ncol : ncols) {
        // carried rows: make the carried corner something indefinite (like -y^T K^-1 y): subtract
        std::vector<double> Ain = Asym;
        for (int i = ncol; i < N; ++i)
            for (int j = ncol; j < N; ++j) Ain[i * N + j] = (i == j) ? 0.0 : 0.0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) A[i * N + j] = (j <= i) ? Ain[i * N + j] : NAN;   // the upper triangle must not be read
        // reference: pivots < ncol only
        std::vector<double> S = Ain, E(N * N, 0.0);
        for (int i = 0; i < N; ++i) E[i * N + i] = 1.0;
        for (int j = 0; j < ncol; ++j) {
            double d = S[j * N + j];
            for (int i = j + 1; i < N; ++i) {
                double m = S[i * N + j] / d;
                for (int c = 0; c < N; ++c) { S[i * N + c] -= m * S[j * N + c]; E[i * N + c] -= m * E[j * N + c]; }
            }
        }
        std::vector<double> Lref(N * N, 0.0), Wref(N * N, 0.0);
        for (int j = 0; j < N; ++j) {
            double sc = j < ncol ? 1.0 / sqrt(S[j * N + j]) : 1.0;
            for (int c = 0; c < N; ++c) {
                double u = S[j * N + c] * sc;   // U[j][c]
                if (j < ncol) { if (c >= j) Lref[c * N + j] = u; }
                else if (c >= ncol) Lref[j * N + c] = S[j * N + c];   // carried corner (full)
                Wref[j * N + c] = E[j * N + c] * sc;
            }
        }
        CK(hipMemcpy(dA, A.data(), N * N * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dL, 0xff, N * N * 8)); CK(hipMemset(dW, 0xff, N * N * 8));
        const int reps = 200;
        hipLaunchKernelGGL(k_diag, dim3(1), dim3(256), 0, 0, dA, dL, dW, ncol, reps, dt, dbad, dst);
        CK(hipDeviceSynchronize());
        std::vector<double> L(N * N), W(N * N);
        long long t[2], st[16]; int bad;
        CK(hipMemcpy(L.data(), dL, N * N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(W.data(), dW, N * N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(st, dst, 128, hipMemcpyDeviceToHost));
        double eL = 0, eW = 0;
        for (int i = 0; i < N * N; ++i) {
            double a = fabs(L[i] - Lref[i]), b = fabs(W[i] - Wref[i]);
            if (!(a <= eL)) eL = a;
            if (!(b <= eW)) eW = b;
        }
        printf("ncol %2d: max|L-Lref| %.3e  max|W-Wref| %.3e  bad %d   %.2f us (%.0f cycles) per factorisation\n", ncol, eL, eW, bad,
               t[0] / 100.0 / reps, (double)t[1] / reps);
        printf("   stamps (us from start): ");
        for (int i = 1; i < 9; ++i) printf("%.2f ", (st[i] - st[0]) / 100.0);
        printf("\n");
        if (!(eL < 1e-11) || !(eW < 1e-9) || bad) ++fails;
    }
    // a non-positive pivot is reported
    {
        std::vector<double> Ain = Asym;
        Ain[21 * N + 21] = -5.0;
        CK(hipMemcpy(dA, Ain.data(), N * N * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_diag, dim3(1), dim3(256), 0, 0, dA, dL, dW, 64, 1, dt, dbad, dst);
        CK(hipDeviceSynchronize());
        int bad;
        CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
        printf("indefinite block: bad = %d (expected 22)\n", bad);
        if (bad != 22) ++fails;
    }
    printf(fails ? "FAILED\n" : "OK\n");
    return fails;
}
