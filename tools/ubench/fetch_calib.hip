// Calibration of rocprofv3's FETCH_SIZE on gfx950 per load width (MI355X_MICROARCH.md, HBM: "exactly 1/2 of the bytes of a wide coalesced
// streaming read; other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels stream
// the same 1 GiB (> the 256-MiB Infinity Cache) once each with the load forms of the library's hand-off code:
//   calib_b128_plain : global_load_dwordx4 (16 B per lane)
//   calib_b128_sc1   : buffer_load_dwordx4 ... sc1 (16 B per lane: operand tiles of the one-launch factorisation, tile.hpp fetch_mk_sc1)
//   calib_b64_sc1    : global_load_dwordx2 sc1 (8 B per lane, agent-scope relaxed atomic loads: accumulator tiles, chol.hip load_acc_sc1)
// run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> --output-format csv -- build_ubench/fetch_calib
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/fetch_calib tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_b128_plain(const double2 *p, int64_t n16, double *sink) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_b128_sc1(const double *p, int64_t n16, double *sink) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        // (a resource per 1-GiB window: 32-bit offsets)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p + 2 * (i & ~(int64_t)0x3ffffff)), 0, 0x7fffffff, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i & 0x3ffffff) * 16), 0, 16);
        acc += __hiloint2double(v.y, v.x) + __hiloint2double(v.w, v.z);
    }
    if (acc == 1.2345) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_b64_sc1(const double *p, int64_t n8, double *sink) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256)
        acc += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 1.2345) sink[0] = acc;
}

int main() {
    const int64_t bytes = 1ll << 30;
    double *buf, *sink;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_b128_plain, dim3(8192), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_b128_sc1, dim3(8192), dim3(256), 0, 0, (const double *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_b64_sc1, dim3(8192), dim3(256), 0, 0, (const double *)buf, bytes / 8, sink);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel read %lld bytes once per launch, 3 launches each\n", (long long)bytes);
    return 0;
}
