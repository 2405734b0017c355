// Context, stream and timing entry points of libdgp_amd.
#include "common.hpp"
#include <sched.h>

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
    if (ctx->llik_pinned) (void)hipHostFree(ctx->llik_pinned);
    for (auto &mb : ctx->mail) {
        if (mb.host) (void)hipHostFree(mb.host);
        if (mb.snap) (void)hipFree(mb.snap);
        if (mb.ev) (void)hipEventDestroy(mb.ev);
    }
    if (ctx->devargs) (void)hipFree(ctx->devargs);
    for (void *q : ctx->scratch)
        if (q) (void)hipFree(q);
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

int ctx_scratch(dgpamd_ctx *ctx, int which, size_t bytes, void **p) {
    if (ctx->scratch_bytes[which] < bytes) {
        if (ctx->scratch[which]) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // nothing may still use the old buffer
            (void)hipFree(ctx->scratch[which]);
            ctx->scratch[which] = nullptr;
            ctx->scratch_bytes[which] = 0;
        }
        const size_t want = bytes + bytes / 4;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch[which], want));
        ctx->scratch_bytes[which] = want;
    }
    *p = ctx->scratch[which];
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

// ---------------------------------------------------------------------------
// Mailboxes: results to the host without a stream synchronisation.  A small result is written into host-coherent pinned
// memory by a one-block kernel, followed by a sequence word the host spins on (a wake-up from hipStreamSynchronize costs
// tens of microseconds, per round of an optimiser that is the turn-around time); a large one goes through the copy engine
// path of hipMemcpyAsync and the host polls the event recorded behind it.  Either way the wait is for THIS result: later
// launches on the stream do not delay it.
// ---------------------------------------------------------------------------
#define MAIL_KERNEL_MAX 16384   // bytes; above this the blit path is faster than one workgroup writing across the bus

// (`snap`: the same words once more in device memory -- what a collect falls back on when the host memory turns out not to be
//  coherent: the state AT THE POST, whatever has been queued behind it since)
__global__ void mail_publish_kernel(const uint32_t *src, uint32_t *host, uint32_t *snap, int words, unsigned long long *flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const uint32_t v = src[i];
        host[i] = v;
        snap[i] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int mail_post(dgpamd_ctx *ctx, dgpamd_ctx::Mailbox &mb, const void *src, size_t bytes) {
    if (mb.pending) BAD_ARG(ctx, "the mailbox still holds a result nobody collected");
    if (mb.cap < bytes + 8) {
        if (mb.host) (void)hipHostFree(mb.host);
        mb.host = nullptr; mb.cap = 0;
        size_t want = bytes + 8 < 65536 ? 65536 : ((bytes + 8 + 4095) & ~(size_t)4095);
        if (hipHostMalloc((void **)&mb.host, want, hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(ctx, hipHostMalloc((void **)&mb.host, want, hipHostMallocDefault));
        }
        mb.cap = want;
    }
    if (!mb.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&mb.ev, hipEventDisableTiming));
    unsigned long long *flag = reinterpret_cast<unsigned long long *>(mb.host + mb.cap - 8);
    mb.by_kernel = bytes <= MAIL_KERNEL_MAX && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
    mb.seq = ++ctx->host_seq;
    mb.src = src; mb.bytes = bytes;
    if (mb.by_kernel) {
        if (!mb.snap) HIP_TRY(ctx, hipMalloc((void **)&mb.snap, MAIL_KERNEL_MAX));
        __atomic_store_n(flag, 0ull, __ATOMIC_RELEASE);
        hipLaunchKernelGGL(mail_publish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t *)src, (uint32_t *)mb.host, (uint32_t *)mb.snap,
                           (int)(bytes / 4), flag, mb.seq);
        LAUNCH_CHECK(ctx);
    } else
        HIP_TRY(ctx, hipMemcpyAsync(mb.host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(mb.ev, ctx->stream));
    mb.pending = 1;
    return DGPAMD_OK;
}

static int mail_collect(dgpamd_ctx *ctx, dgpamd_ctx::Mailbox &mb, void *host_dst, size_t bytes) {
    if (!mb.pending) BAD_ARG(ctx, "nothing was posted to this mailbox");
    if (host_dst && bytes != mb.bytes) BAD_ARG(ctx, "size differs from the posted one");
    mb.pending = 0;
    const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(mb.host + mb.cap - 8);
    for (unsigned long long it = 1;; ++it) {
        if (mb.by_kernel) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == mb.seq) break;
            __builtin_ia32_pause();
            if ((it & 0xffff) != 0) continue;   // now and then: is the stream still alive? (a fault would leave the word unwritten)
            if (it > (1ull << 22)) sched_yield();   // (a stalled stream must not burn a core flat out)
        }
        const hipError_t q = hipEventQuery(mb.ev);
        if (q == hipSuccess) {
            // (this memory is not coherent: the slow way -- from the snapshot the publishing kernel left in device memory, which is
            //  the state at the post; the event says that kernel has finished, so the blocking copy needs no further ordering)
            if (mb.by_kernel && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != mb.seq)
                HIP_TRY(ctx, hipMemcpy(mb.host, mb.snap, mb.bytes, hipMemcpyDeviceToHost));
            break;
        }
        if (q != hipErrorNotReady) HIP_TRY(ctx, q);
        if (!mb.by_kernel) {
            __builtin_ia32_pause();
            if (it > 200000) {   // (~0.1 s of polling: sleep instead)
                HIP_TRY(ctx, hipEventSynchronize(mb.ev));
                break;
            }
        }
    }
    if (host_dst) memcpy(host_dst, mb.host, mb.bytes);
    return DGPAMD_OK;
}

extern "C" int dgpamd_set_reduce_hook(dgpamd_ctx *ctx, dgpamd_reduce_hook hook, void *user) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->reduce_hook = hook;
    ctx->reduce_user = user;
    return DGPAMD_OK;
}

extern "C" int dgpamd_post(dgpamd_ctx *ctx, const void *device_src, size_t bytes, int slot) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!device_src || bytes == 0) BAD_ARG(ctx, "null pointer or nothing to copy");
    if (slot < 0 || slot >= DGPAMD_MAILBOXES) BAD_ARG(ctx, "no such mailbox");
    return mail_post(ctx, ctx->mail[slot], device_src, bytes);
}

extern "C" int dgpamd_collect(dgpamd_ctx *ctx, int slot, void *host_dst, size_t bytes) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (slot < 0 || slot >= DGPAMD_MAILBOXES) BAD_ARG(ctx, "no such mailbox");
    return mail_collect(ctx, ctx->mail[slot], host_dst, bytes);
}

extern "C" int dgpamd_fetch(dgpamd_ctx *ctx, const void *device_src, void *host_dst, size_t bytes) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!device_src || !host_dst) BAD_ARG(ctx, "null pointer");
    if (bytes == 0) return DGPAMD_OK;
    static const bool spin = !(getenv("DGPAMD_FETCH_SPIN") && getenv("DGPAMD_FETCH_SPIN")[0] == '0');
    if (spin && bytes <= MAIL_KERNEL_MAX) {   // the few words a sampler / optimiser step returns: no sleep, no wake-up
        int rc = mail_post(ctx, ctx->mail[DGPAMD_MAILBOXES], device_src, bytes);
        if (rc) return rc;
        return mail_collect(ctx, ctx->mail[DGPAMD_MAILBOXES], host_dst, bytes);
    }
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

extern "C" int dgpamd_debug_tasklog(dgpamd_ctx *ctx, long long *device_buf, long long words) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->tlog = words > 0 ? device_buf : nullptr;
    ctx->tlog_words = device_buf ? words : 0;
    return DGPAMD_OK;
}

extern "C" int dgpamd_set_linkgp_direct(dgpamd_ctx *ctx, int enable) {
    if (!ctx) return DGPAMD_BAD_ARG;
    ctx->linkgp_direct = enable ? 1 : 0;
    return DGPAMD_OK;
}
