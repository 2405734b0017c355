// Context, stream and timing entry points of libdgp_amd.
#include "common.hpp"

#include <new>
#include <stdlib.h>

extern "C" const char *dgpamd_version(void) { return "dgp_amd 0.1 (gfx950)"; }

extern "C" int64_t dgpamd_padded_dim(int64_t n) { return padded_dim(n); }

extern "C" int dgpamd_create(int device, void *stream, dgpamd_ctx **out) {
    if (!out) return DGPAMD_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DGPAMD_HIP_ERROR;
    if (device < 0 || device >= count) return DGPAMD_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return DGPAMD_HIP_ERROR;
    dgpamd_ctx *ctx = new (std::nothrow) dgpamd_ctx;
    if (!ctx) return DGPAMD_HIP_ERROR;
    ctx->device = device;
    ctx->err[0] = 0;
    ctx->prof_class = PROF_NONE;
    ctx->prof_work = 0.0;
    ctx->use_graphs = 1;
    ctx->linkgp_direct = 0;
    ctx->potrf_mode = 1;
    ctx->pred = nullptr;
    if (const char *pm = getenv("DGPAMD_POTRF_MODE")) ctx->potrf_mode = (pm[0] == '0') ? 0 : 1;   // (0: per-block-step launches, e.g. on a shared device)
    ctx->trace = nullptr;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
        ctx->num_cu = ncu;
    }
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    ctx->host_seq = 0;
    ctx->args_inflight = 0;
    ctx->devargs = ctx->hostargs = nullptr;
    ctx->devargs_bytes = 0;
    // NULL = the device's default (null) stream, which is also torch's default current stream
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
    *out = ctx;
    return DGPAMD_OK;
}

extern "C" int dgpamd_destroy(dgpamd_ctx *ctx) {
    if (!ctx) return DGPAMD_BAD_ARG;
    for (auto &kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->devargs) (void)hipFree(ctx->devargs);
    if (ctx->hostargs) (void)hipHostFree(ctx->hostargs);
    delete ctx;
    return DGPAMD_OK;
}

extern "C" const char *dgpamd_last_error(const dgpamd_ctx *ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int dgpamd_sync(dgpamd_ctx *ctx) {
    if (!ctx) return DGPAMD_BAD_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DGPAMD_OK;
}

int ensure_pinned(dgpamd_ctx *ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return DGPAMD_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    const size_t want = bytes < 65536 ? 65536 : bytes;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return DGPAMD_OK;
}

int ensure_devargs(dgpamd_ctx *ctx, size_t bytes) {
    if (ctx->devargs_bytes >= bytes) return DGPAMD_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // nothing may still read the old arrays
    if (ctx->devargs) (void)hipFree(ctx->devargs);
    if (ctx->hostargs) (void)hipHostFree(ctx->hostargs);
    ctx->devargs = ctx->hostargs = nullptr;
    ctx->devargs_bytes = 0;
    const size_t want = bytes < 65536 ? 65536 : bytes;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->devargs, want));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->hostargs, want, hipHostMallocDefault));
    ctx->devargs_bytes = want;
    return DGPAMD_OK;
}

extern "C" int dgpamd_fetch(dgpamd_ctx *ctx, const void *device_src, void *host_dst, size_t bytes) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!device_src || !host_dst) BAD_ARG(ctx, "null pointer");
    if (bytes == 0) return DGPAMD_OK;
    int rc = ensure_pinned(ctx, bytes);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host_dst, ctx->pinned, bytes);
    return DGPAMD_OK;
}

extern "C" int dgpamd_fetch2(dgpamd_ctx *ctx, const void *src_a, size_t bytes_a, const void *src_b, size_t bytes_b,
                             void *host_dst) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!src_a || !src_b || !host_dst) BAD_ARG(ctx, "null pointer");
    int rc = ensure_pinned(ctx, bytes_a + bytes_b);
    if (rc) return rc;
    char *pin = reinterpret_cast<char *>(ctx->pinned);
    HIP_TRY(ctx, hipMemcpyAsync(pin, src_a, bytes_a, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(pin + bytes_a, src_b, bytes_b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host_dst, pin, bytes_a + bytes_b);
    return DGPAMD_OK;
}

extern "C" int dgpamd_event_create(dgpamd_ctx *ctx, void **ev) {
    if (!ctx || !ev) return DGPAMD_BAD_ARG;
    hipEvent_t e;
    HIP_TRY(ctx, hipEventCreate(&e));
    *ev = (void *)e;
    return DGPAMD_OK;
}

extern "C" int dgpamd_event_record(dgpamd_ctx *ctx, void *ev) {
    if (!ctx || !ev) return DGPAMD_BAD_ARG;
    HIP_TRY(ctx, hipEventRecord((hipEvent_t)ev, ctx->stream));
    return DGPAMD_OK;
}

extern "C" int dgpamd_event_elapsed_ms(dgpamd_ctx *ctx, void *start, void *stop, float *ms_h) {
    if (!ctx || !start || !stop || !ms_h) return DGPAMD_BAD_ARG;
    HIP_TRY(ctx, hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(ctx, hipEventElapsedTime(ms_h, (hipEvent_t)start, (hipEvent_t)stop));
    return DGPAMD_OK;
}

extern "C" int dgpamd_event_destroy(dgpamd_ctx *ctx, void *ev) {
    if (!ctx || !ev) return DGPAMD_BAD_ARG;
    HIP_TRY(ctx, hipEventDestroy((hipEvent_t)ev));
    return DGPAMD_OK;
}

// ---- launch timing of one kernel class (bench.py: roofline of the dominant kernel) -----------------
extern "C" int dgpamd_prof_enable(dgpamd_ctx *ctx, int kernel_class) {
    if (!ctx || kernel_class < 0 || kernel_class > PROF_GP_QUAD) return DGPAMD_BAD_ARG;
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    ctx->prof_pair_work.clear();
    ctx->prof_pair_pred.clear();
    ctx->prof_work = 0.0;
    ctx->prof_class = kernel_class;
    return DGPAMD_OK;
}

// Elapsed time reported by an event pair with NOTHING between the two records: the fixed cost that
// bracketing a launch adds to its measured duration (bench.py subtracts it per launch).
extern "C" int dgpamd_prof_event_overhead_us(dgpamd_ctx *ctx, double *us_h) {
    if (!ctx || !us_h) return DGPAMD_BAD_ARG;
    const int reps = 64;
    hipEvent_t ev[2 * reps];
    for (int i = 0; i < 2 * reps; ++i) HIP_TRY(ctx, hipEventCreate(&ev[i]));
    for (int i = 0; i < reps; ++i) {
        HIP_TRY(ctx, hipEventRecord(ev[2 * i], ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ev[2 * i + 1], ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    for (int i = 0; i < 2 * reps; ++i) (void)hipEventDestroy(ev[i]);
    *us_h = 1e3 * tot / reps;
    return DGPAMD_OK;
}

extern "C" int dgpamd_prof_collect(dgpamd_ctx *ctx, int64_t *launches_h, double *total_ms_h, double *work_h) {
    if (!ctx || !launches_h || !total_ms_h || !work_h) return DGPAMD_BAD_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0, work = 0.0;
    const size_t pairs = ctx->prof_events.size() / 2;
    int64_t counted = 0;
    for (size_t i = 0; i < pairs; ++i) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
        // a predicated launch that found its predicate set did nothing (an empty event pair reads ~5 us, such a launch
        // 6-8, the smallest real one of the bench 25): neither its time nor the work it would have done is counted
        if (i < ctx->prof_pair_pred.size() && ctx->prof_pair_pred[i] && ms < 0.012f) continue;
        total += ms;
        work += i < ctx->prof_pair_work.size() ? ctx->prof_pair_work[i] : 0.0;
        ++counted;
    }
    *launches_h = counted;
    *total_ms_h = total;
    *work_h = work;
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    ctx->prof_pair_work.clear();
    ctx->prof_pair_pred.clear();
    ctx->prof_work = 0.0;
    ctx->prof_class = PROF_NONE;
    return DGPAMD_OK;
}

extern "C" int dgpamd_set_graphs(dgpamd_ctx *ctx, int enable) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->use_graphs = enable ? 1 : 0;
    return DGPAMD_OK;
}

extern "C" int dgpamd_debug_trace(dgpamd_ctx *ctx, long long *device_buf) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->trace = device_buf;
    return DGPAMD_OK;
}

extern "C" int dgpamd_set_linkgp_direct(dgpamd_ctx *ctx, int enable) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->linkgp_direct = enable ? 1 : 0;
    return DGPAMD_OK;
}
