// Ticket atomics of the one-launch factorisation's queues: throughput and latency of returning agent-scope fetch-adds from one lane of every
// workgroup (two workgroups per CU), all on ONE word, on one word per XCD group (different lines), and with a polled word on the same line.
// build: hipcc --offload-arch=gfx950 -O3 -o build_ubench/atomic_tickets tools/ubench/atomic_tickets.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void tickets(int *heads, int stride_ints, int nheads, int iters, int gap, long long *cyc, int *sink) {
    if (threadIdx.x != 0) return;
    int *h = heads + (blockIdx.x % nheads) * stride_ints;
    long long t0 = wall_clock64();
    int acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += __hip_atomic_fetch_add(h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);   // ~64 x 8 cycles between a workgroup's pulls
    }
    cyc[blockIdx.x] = wall_clock64() - t0;
    sink[blockIdx.x] = acc;
}

int main() {
    int *heads, *sink;
    long long *cyc;
    const int wgs = 512;
    hipMalloc(&heads, 1 << 16);
    hipMalloc(&sink, wgs * 4);
    hipMalloc(&cyc, wgs * 8);
    long long hc[512];
    printf("# %d workgroups, one lane each: returning agent-scope fetch-add, `gap` x s_sleep(8) between a workgroup's atomics\n", wgs);
    printf("# heads stride(B) gap  iters |  launch us   atomics/us   mean latency per atomic (us)\n");
    for (int gap : {0, 8, 32}) {
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int nheads = cfg == 0 ? 1 : 8, stride = cfg == 2 ? 128 / 4 : (cfg == 3 ? 4096 / 4 : 1);   // 0: one word; 1: 8 words of one line; 2: 8 lines; 3: 8 words 4 KB apart
            const int iters = 200;
            hipMemset(heads, 0, 1 << 16);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tickets, dim3(wgs), dim3(256), 0, 0, heads, stride, nheads, iters, gap, cyc, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
            double mean = 0;
            for (int i = 0; i < wgs; ++i) mean += hc[i];
            mean /= wgs;   // 100-MHz ticks for `iters` atomics + gaps
            const double gap_us = gap * 8 * 64 / 2100.0;
            printf("%5d %9d %4d %6d | %9.1f %12.1f %10.2f\n", nheads, stride * 4, gap, iters, ms * 1e3, wgs * (double)iters / (ms * 1e3), mean / 100.0 / iters - gap_us);
        }
    }
    return 0;
}
