// Store-only ceilings for K assembly, part 2: does the ORDER in which the 64 x 64 tiles are written matter?  ("order the
// grid so co-resident workgroups write the same row block".)  Constant data, 16 bytes per lane, a wave instruction writes 4 rows x 256 B.
//   0: linear fill                        1: lower tiles row-major + their mirrors (the library's order)
//   2: ALL tiles directly, row-major over the full matrix (no mirror writes)       3: lower tiles only, row-major (half the bytes)
//   4: the mirrors only, in the lower tiles' order (a column of tiles per tile row) 5: ALL tiles directly, column-major over the full matrix
//   6: strips of 8 rows x 512 columns (4 KB of every row from ONE workgroup), row-major     7: strips of 2 rows x 2048 columns (16 KB per row)
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/store_orders tools/ubench/store_orders.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ inline void tri_decode(int t, int &bi, int &bj) {
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    bi = i; bj = t - i * (i + 1) / 2;
}
__global__ __launch_bounds__(256) void fill_lin(double *K, int64_t total) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < total) *reinterpret_cast<d2 *>(K + i) = (d2){1.5, 2.5};
}
__device__ inline void write_tile(double *K, int64_t ld, int64_t r0, int64_t c0, int n, int tid) {
    const int wave = tid >> 6, lane = tid & 63, rsub = lane >> 4, cl = (lane & 15) * 2;
    for (int cseg = 0; cseg < 64; cseg += 32)
        for (int r = wave * 4 + rsub; r < 64; r += 16) {
            const int64_t gr = r0 + r, gc = c0 + cseg + cl;
            if (gr < n && gc + 1 < n) *reinterpret_cast<d2 *>(K + gr * ld + gc) = (d2){1.5, 2.5};
        }
}
__global__ __launch_bounds__(256) void fill_order(double *K, int n, int64_t ld, int nb, int mode) {
    const int tid = threadIdx.x;
    int bi, bj;
    if (mode == 2) { bi = blockIdx.x / nb; bj = blockIdx.x % nb; write_tile(K, ld, 64LL * bi, 64LL * bj, n, tid); return; }
    if (mode == 5) { bj = blockIdx.x / nb; bi = blockIdx.x % nb; write_tile(K, ld, 64LL * bi, 64LL * bj, n, tid); return; }
    tri_decode(blockIdx.x, bi, bj);
    if (mode == 1 || mode == 3) write_tile(K, ld, 64LL * bi, 64LL * bj, n, tid);
    if ((mode == 1 || mode == 4) && bi != bj) write_tile(K, ld, 64LL * bj, 64LL * bi, n, tid);
}
// a strip of R rows x C columns per workgroup: every wave instruction writes 1 KB of one row
template <int R, int C>
__global__ __launch_bounds__(256) void fill_strip(double *K, int n, int64_t ld, int ncs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t r0 = (int64_t)(blockIdx.x / ncs) * R, c0 = (int64_t)(blockIdx.x % ncs) * C;
    for (int r = 0; r < R; ++r)
        for (int cseg = wave * 128; cseg < C; cseg += 512) {
            const int64_t gr = r0 + r, gc = c0 + cseg + 2 * lane;
            if (gr < n && gc + 1 < n) *reinterpret_cast<d2 *>(K + gr * ld + gc) = (d2){1.5, 2.5};
        }
}
int main(int argc, char **argv) {
    std::vector<int> sizes = {5120, 8192, 16384};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("store-only, 64 x 64 tiles, GB/s of the bytes each variant writes (of 8 TB/s)\n%6s | %-16s %-16s %-16s %-16s %-16s %-16s\n", "n", "linear", "lower+mirror", "all row-major", "lower only",
           "mirrors only", "all col-major");
    printf("       (+ strips 8 x 512, 2 x 2048)\n");
    for (int n : sizes) {
        const int64_t ld = n, total = (int64_t)ld * n;
        const int nbuf = total * 8 > 1000000000LL ? 2 : 3, nb = (n + 63) / 64, nt = nb * (nb + 1) / 2;
        std::vector<double *> bufs(nbuf);
        for (auto &b : bufs) hipMalloc((void **)&b, total * 8);
        printf("%6d |", n);
        for (int mode = 0; mode < 8; ++mode) {
            auto launch = [&](double *K) {
                if (mode == 6) { const int ncs = (n + 511) / 512; hipLaunchKernelGGL((fill_strip<8, 512>), dim3(((n + 7) / 8) * ncs), dim3(256), 0, 0, K, n, ld, ncs); }
                else if (mode == 7) { const int ncs = (n + 2047) / 2048; hipLaunchKernelGGL((fill_strip<2, 2048>), dim3(((n + 1) / 2) * ncs), dim3(256), 0, 0, K, n, ld, ncs); }
                else if (mode == 0) hipLaunchKernelGGL(fill_lin, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, 0, K, total);
                else hipLaunchKernelGGL(fill_order, dim3((mode == 2 || mode == 5) ? nb * nb : nt), dim3(256), 0, 0, K, n, ld, nb, mode);
            };
            for (auto b : bufs) launch(b);
            hipDeviceSynchronize();
            const int reps = 24;
            hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r) launch(bufs[r % nbuf]);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (mode == 3 || mode == 4) ? 4.0 * total : 8.0 * total;
            const double gbs = bytes / (ms / reps) / 1e6;
            printf(" %7.0f (%.3f) ", gbs, gbs / 8000.0);
        }
        printf("\n");
        for (auto b : bufs) hipFree(b);
    }
    return 0;
}
