// Which CU does a workgroup run on?  HW_ID / XCC_ID hardware registers per workgroup (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/hwid tools/ubench/hwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
    __shared__ double pad[4096];   // 32 KB: a few workgroups per CU
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        pad[0] = hw;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 1 && pad[0] < 0) out[0] = 0;
}

int main() {
    const int nwg = 1024;
    unsigned *d;
    CK(hipMalloc(&d, nwg * 8));
    probe<<<nwg, 256>>>(d, 2000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * nwg);
    CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 20; ++i) printf("wg %3d  hw_id %08x  xcc_id %08x\n", i, h[2 * i], h[2 * i + 1]);
    // which bit fields vary, and how many workgroups share a (masked hw_id, xcc) key
    unsigned orv = 0, andv = ~0u, xo = 0, xa = ~0u;
    for (int i = 0; i < nwg; ++i) { orv |= h[2 * i]; andv &= h[2 * i]; xo |= h[2 * i + 1]; xa &= h[2 * i + 1]; }
    printf("hw_id varying bits %08x   xcc_id varying bits %08x\n", orv & ~andv, xo & ~xa);
    for (unsigned mask : {0x0000ff00u, 0x0000ff30u, 0x000fff00u, 0x0000f300u}) {
        std::map<unsigned long long, int> cnt;
        for (int i = 0; i < nwg; ++i) cnt[((unsigned long long)(h[2 * i + 1] & 0xf) << 32) | (h[2 * i] & mask)]++;
        int mx = 0;
        for (auto &kv : cnt) mx = kv.second > mx ? kv.second : mx;
        printf("mask %08x: %zu distinct keys, max %d workgroups per key\n", mask, cnt.size(), mx);
    }
    for (int i = 0; i < 8; ++i) printf("wg %3d  hw_id %08x  xcc %x | wg %3d hw_id %08x xcc %x\n", i, h[2 * i], h[2 * i + 1], 256 + i, h[2 * (256 + i)], h[2 * (256 + i) + 1]);
    return 0;
}
