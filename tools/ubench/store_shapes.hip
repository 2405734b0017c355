// Store-only ceilings for K assembly (gate (i)): a full symmetric n x n f64 matrix written from LOWER square
// tiles (direct tile + mirrored tile per workgroup, triangle in row-major order), constant data, 16 bytes per lane.
// What changes is the SHAPE of what one wave store instruction writes and the tile edge T:
//   shape 0: linear fill (1 KB contiguous per instruction, workgroups consecutive) -- the ceiling
//   shape 1: T = 64,  an instruction writes 4 rows x 256 B (the library's kernel since round 3)
//   shape 2: T = 128, an instruction writes 1 row x 1 KB
//   shape 3: T = 256, an instruction writes 1 row x 1 KB (rows of 2 KB from two instructions of the same wave)
//   shape 4: T = 128, an instruction writes 2 rows x 512 B
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/store_shapes tools/ubench/store_shapes.hip ; run: build_ubench/store_shapes
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
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fill_lin(double *K, int64_t total) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < total) *reinterpret_cast<d2 *>(K + i) = (d2){1.5, 2.5};
}

// one square tile (r0.., c0..) of edge T: rows handed to the waves round-robin; RPI rows per instruction (64 / RPI lanes per row)
template <int T, int RPI>
__device__ inline void write_tile(double *K, int64_t ld, int64_t r0, int64_t c0, int n, int tid, double val) {
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int LPR = 64 / RPI;            // lanes per row
    constexpr int SEG = 2 * LPR;             // doubles a row gets from one instruction
    const int rsub = lane / LPR, cl = (lane % LPR) * 2;
    for (int cseg = 0; cseg < T; cseg += SEG)
        for (int r = wave * RPI + rsub; r < T; r += 4 * RPI) {
            const int64_t gr = r0 + r, gc = c0 + cseg + cl;
            if (gr < n && gc + 1 < n) *reinterpret_cast<d2 *>(K + gr * ld + gc) = (d2){val, val + 1.0};
        }
}
// 1 row per instruction but the wave walks ALONG the row first (two instructions complete a 2-KB row back to back)
template <int T, int RPI>
__global__ __launch_bounds__(256) void fill_tiles(double *K, int n, int64_t ld) {
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x;
    write_tile<T, RPI>(K, ld, (int64_t)bi * T, (int64_t)bj * T, n, tid, 1.5);
    if (bi != bj) write_tile<T, RPI>(K, ld, (int64_t)bj * T, (int64_t)bi * T, n, tid, 2.5);
}
template <int T>
__global__ __launch_bounds__(256) void fill_tiles_rowfirst(double *K, int n, int64_t ld) {
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int side = 0; side < (bi != bj ? 2 : 1); ++side) {
        const int64_t r0 = (int64_t)(side ? bj : bi) * T, c0 = (int64_t)(side ? bi : bj) * T;
        for (int r = wave; r < T; r += 4)
            for (int cseg = 0; cseg < T; cseg += 128) {
                const int64_t gr = r0 + r, gc = c0 + cseg + 2 * lane;
                if (gr < n && gc + 1 < n) *reinterpret_cast<d2 *>(K + gr * ld + gc) = (d2){1.5, 2.5};
            }
    }
}

int main(int argc, char **argv) {
    std::vector<int> sizes = {5120, 8192, 16384};
    int pad = 0;   // (usage: store_shapes [pad n1 n2 ...]: leading dimension n + pad)
    if (argc > 2) {
        pad = atoi(argv[1]);
        sizes.clear();
        for (int i = 2; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("store-only ceilings, full symmetric n x n f64 from lower tiles; GB/s of 8 n^2 bytes (of 8 TB/s)\n");
    printf("%6s | %-18s %-18s %-18s %-18s %-18s\n", "n", "linear fill", "T=64 4x256B", "T=128 1x1KB", "T=256 1x1KB(2KB)", "T=128 2x512B");
    for (int n : sizes) {
        const int64_t ld = n + pad, total = (int64_t)ld * n;
        const int nbuf = total * 8 > 1000000000LL ? 2 : 3;
        std::vector<double *> bufs(nbuf);
        for (auto &b : bufs) CK(hipMalloc((void **)&b, total * 8));
        double res[5];
        for (int shape = 0; shape < 5; ++shape) {
            auto launch = [&](double *K) {
                const int T = shape == 1 ? 64 : (shape == 3 ? 256 : 128);
                const int nb = (n + T - 1) / T, nt = nb * (nb + 1) / 2;
                switch (shape) {
                    case 0: hipLaunchKernelGGL(fill_lin, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, 0, K, total); break;
                    case 1: hipLaunchKernelGGL((fill_tiles<64, 4>), dim3(nt), dim3(256), 0, 0, K, n, ld); break;
                    case 2: hipLaunchKernelGGL((fill_tiles<128, 1>), dim3(nt), dim3(256), 0, 0, K, n, ld); break;
                    case 3: hipLaunchKernelGGL((fill_tiles_rowfirst<256>), dim3(nt), dim3(256), 0, 0, K, n, ld); break;
                    default: hipLaunchKernelGGL((fill_tiles<128, 2>), dim3(nt), dim3(256), 0, 0, K, n, ld); break;
                }
            };
            for (auto b : bufs) launch(b);
            CK(hipDeviceSynchronize());
            const int reps = 24;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) launch(bufs[r % nbuf]);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            res[shape] = 8.0 * total / (ms / reps) / 1e6;
        }
        printf("%6d |", n);   // (ld = n + pad)
        for (int s = 0; s < 5; ++s) printf(" %7.0f (%.3f)    ", res[s], res[s] / 8000.0);
        printf("\n");
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
