// Single-workgroup latencies on an otherwise idle MI355X (the regime of the factorisation's pivot chain).
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/latency tools/ubench/latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 4096

__global__ void k_fma(double *out, long long *t, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(x, a, b);
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_rcp(double *out, long long *t) {
    double x = out[threadIdx.x];
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_rsq(double *out, long long *t) {
    double x = out[threadIdx.x];
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_mfma(double *out, long long *t) {
    d4 acc = {0, 0, 0, 0};
    double a = out[threadIdx.x], b = out[threadIdx.x + 64];
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    long long t1 = wall_clock64();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_mfma_dep_valu(double *out, long long *t) {   // mfma -> valu on result -> mfma (operand dependency)
    d4 acc = {0, 0, 0, 0};
    double a = out[threadIdx.x], b = out[threadIdx.x + 64];
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        a = acc[0] * 0.5;
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_lds(double *out, long long *t) {   // write -> barrier -> read (other lane) round trip, 256 threads
    __shared__ double s[256];
    double x = out[threadIdx.x];
    long long t0 = wall_clock64();
    for (int i = 0; i < N; ++i) {
        s[threadIdx.x] = x;
        __syncthreads();
        x = s[(threadIdx.x + 65) & 255] + 1.0;
        __syncthreads();
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_bar(double *out, long long *t) {
    long long t0 = wall_clock64();
    for (int i = 0; i < N; ++i) __syncthreads();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_ldsread(double *out, long long *t) {   // dependent ds_read chain (pointer chasing)
    __shared__ int s[256];
    s[threadIdx.x] = (threadIdx.x * 7 + 3) & 255;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) p = s[p];
    long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}

int main() {
    double *out; long long *t;
    CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&t, 64));
    CK(hipMemset(out, 0, 4096 * 8));
    long long h;
    auto rep = [&](const char *name, double per) { printf("%-52s %8.2f ns\n", name, per); };
    for (int pass = 0; pass < 2; ++pass) {
        k_fma<<<1, 64>>>(out, t, 0.999, 1e-3); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent v_fma_f64 (1 wave)", h * 10.0 / N);
        k_fma<<<1, 256>>>(out, t, 0.999, 1e-3); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent v_fma_f64 (4 waves, 1 per SIMD)", h * 10.0 / N);
        CK(hipMemset(out, 0x3f, 4096 * 8));
        k_rcp<<<1, 64>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent v_rcp_f64", h * 10.0 / N);
        k_rsq<<<1, 64>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent v_rsq_f64", h * 10.0 / N);
        CK(hipMemset(out, 0, 4096 * 8));
        k_mfma<<<1, 64>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent mfma_f64_16x16x4 (same accumulator)", h * 10.0 / N);
        k_mfma_dep_valu<<<1, 64>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("mfma -> v_mul on result -> mfma operand", h * 10.0 / N);
        k_lds<<<1, 256>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("LDS write -> barrier -> read -> barrier (4 waves)", h * 10.0 / N);
        k_bar<<<1, 256>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("__syncthreads alone (4 waves)", h * 10.0 / N);
        k_ldsread<<<1, 64>>>(out, t); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); if (pass) rep("dependent ds_read_b32", h * 10.0 / N);
    }
    return 0;
}
