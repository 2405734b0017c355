// Shared host/device helpers of libdgp_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dgp_amd.h"

#define NB 64          // factorisation block / tile edge
#define LDT 66         // LDS leading dimension of a 64-wide f64 tile (conflict-free ds_read_b64 fragments)

#include <array>
#include <map>
#include <vector>

// kernel classes for the launch-timing facility (dgpamd_prof_*)
enum { PROF_NONE = 0, PROF_KMATRIX = 1, PROF_POTRF_DIAG = 2, PROF_TRSM = 3, PROF_SYRK = 4, PROF_TRTRI = 5,
       PROF_LAUUM = 6, PROF_GRAD = 7, PROF_LINKGP_J = 8, PROF_GP_QUAD = 9 };

struct dgpamd_ctx {
    int device;
    int num_cu;                                       // compute units of the device (workgroups i and i + num_cu share one)
    hipStream_t stream;
    bool own_stream;
    char err[512];
    int prof_class;                                   // PROF_* being timed (0 = off)
    double prof_work;                                 // algorithmic flops (or bytes) of the timed launches
    std::vector<hipEvent_t> prof_events;              // start/stop pairs
    std::vector<double> prof_pair_work;               // algorithmic work of each pair
    std::vector<char> prof_pair_pred;                 // 1: the launch was predicated (ctx->pred set): it may have done nothing
    int use_graphs;                                   // replay static launch sequences as hipGraphs
    int linkgp_direct;                                // 1: evaluate the Matern J factor in the reference's direct form
    const int32_t *pred;                              // device word: kernels launched while it is set return at once when it is non-zero (dgpamd_ess_queue)
    int potrf_mode;                                   // 1: factorisation as one persistent dataflow launch; 0: one launch per block step
    long long *trace;                                 // device buffer for in-kernel timestamps (diagnostics), or null
    long long *tlog = nullptr;                        // device buffer for the one-launch factorisation's full task log (dgpamd_debug_tasklog)
    long long tlog_words = 0;
    double *pinned;                                   // small pinned staging buffer for result copies (lazy)
    int args_inflight;                                // 1 while a copy of `hostargs` may still be running (a call left early)
    unsigned long long host_seq;                      // sequence number of the last result a kernel published into `pinned`
    char *devargs, *hostargs;                         // argument arrays of the multi-node launches (device / pinned host)
    size_t devargs_bytes;
    size_t pinned_bytes;
    std::map<std::array<uint64_t, 10>, hipGraphExec_t> graphs;
    struct Mailbox {                                  // dgpamd_post / dgpamd_collect: a result on its way to the host
        char *host = nullptr;                         // pinned, host-coherent; the last 8 bytes are the sequence word
        size_t cap = 0, bytes = 0;
        hipEvent_t ev = nullptr;
        const void *src = nullptr;
        void *snap = nullptr;                         // device copy of a small result as it was at the post (mail_collect's fallback)
        unsigned long long seq = 0;
        int pending = 0, by_kernel = 0;
    } mail[DGPAMD_MAILBOXES + 1];                     // (the last one is dgpamd_fetch's own)
    dgpamd_reduce_hook reduce_hook = nullptr;         // sum over ranks of a device vector, queued on the stream (dgpamd_set_reduce_hook)
    void *reduce_user = nullptr;
    // dgpamd_llik_batch_launch / _wait: the evaluation in flight and its own pinned staging buffer
    double *llik_pinned = nullptr;
    size_t llik_pinned_bytes = 0, llik_bytes = 0;
    int llik_pending = 0;
    unsigned long long llik_seq = 0;
    const double *llik_dev_out = nullptr;
    void *scratch[2] = {nullptr, nullptr};            // device scratch grown on demand (ctx_scratch): [0] per-row partial results of the
    size_t scratch_bytes[2] = {0, 0};                 // Vecchia row kernels, [1] per-chunk candidate lists of the neighbour searches
};

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,  \
                     hipGetErrorString(e__));                                                      \
            return DGPAMD_HIP_ERROR;                                                               \
        }                                                                                          \
    } while (0)

#define LAUNCH_CHECK(ctx) HIP_TRY(ctx, hipGetLastError())

#define BAD_ARG(ctx, msg)                                                     \
    do {                                                                      \
        snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s", __func__, msg);    \
        return DGPAMD_BAD_ARG;                                                \
    } while (0)

// Run `body` (a sequence of launches on ctx->stream with a static shape) through a cached hipGraph.
// Falls back to direct launches on the null stream (not capturable), while a kernel class is being
// timed, or if capture fails.
template <typename F>
static inline int graph_run(dgpamd_ctx *ctx, const std::array<uint64_t, 10> &key, F body) {
    if (!ctx->use_graphs || ctx->stream == nullptr || ctx->prof_class != PROF_NONE) return body();
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return body();
        }
        int rc = body();
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (rc != DGPAMD_OK || e != hipSuccess || !graph) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            return rc != DGPAMD_OK ? rc : body();
        }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return body();
        }
        if (ctx->graphs.size() >= 64) {   // callers that keep changing buffers: do not let the cache grow without bound
            (void)hipStreamSynchronize(ctx->stream);   // (the host runs ahead: a replay may still be executing)
            for (auto &kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
            ctx->graphs.clear();
        }
        it = ctx->graphs.emplace(key, exec).first;
    }
    HIP_TRY(ctx, hipGraphLaunch(it->second, ctx->stream));
    return DGPAMD_OK;
}

// bracket one launch with HIP events on the launching stream when its class is being timed
#define PROF_BEGIN(ctx, cls, work)                                            \
    do {                                                                      \
        if ((ctx)->prof_class == (cls)) {                                     \
            hipEvent_t e0__;                                                  \
            if (hipEventCreate(&e0__) == hipSuccess) {                        \
                (void)hipEventRecord(e0__, (ctx)->stream);                    \
                (ctx)->prof_events.push_back(e0__);                           \
                (ctx)->prof_pair_work.push_back((double)(work));              \
                (ctx)->prof_pair_pred.push_back((ctx)->pred ? 1 : 0);         \
            }                                                                 \
        }                                                                     \
    } while (0)
#define PROF_END(ctx, cls)                                                    \
    do {                                                                      \
        if ((ctx)->prof_class == (cls) && ((ctx)->prof_events.size() & 1)) {  \
            hipEvent_t e1__;                                                  \
            if (hipEventCreate(&e1__) == hipSuccess) {                        \
                (void)hipEventRecord(e1__, (ctx)->stream);                    \
                (ctx)->prof_events.push_back(e1__);                           \
            }                                                                 \
        }                                                                     \
    } while (0)


static inline int64_t padded_dim(int64_t n) { return ((n + 1 + NB - 1) / NB) * NB; }

typedef double d4 __attribute__((ext_vector_type(4)));

// Kernel parameters of one GP node's correlation function, passed by value.
struct KernParams {
    int kind;                  // DGPAMD_SEXP / DGPAMD_MATERN25
    int Dl, Dg;                // local (gathered) and global columns
    int colmap[DGPAMD_MAXD];   // gathered column of Xloc for d < Dl
    double inv_len[DGPAMD_MAXD];
    double nugget;
};

static inline int fill_kern_params(dgpamd_ctx *ctx, KernParams &kp, int kind, const int32_t *colmap_h, int Dl, int Dg,
                                   const double *length_h, int nlen, double nugget) {
    int D = Dl + Dg;
    if (kind != DGPAMD_SEXP && kind != DGPAMD_MATERN25) BAD_ARG(ctx, "kind must be 0 (sexp) or 1 (matern2.5)");
    if (D <= 0 || D > DGPAMD_MAXD) BAD_ARG(ctx, "need 1 <= Dl+Dg <= DGPAMD_MAXD");
    if (nlen != 1 && nlen != D) BAD_ARG(ctx, "nlen must be 1 or Dl+Dg");
    kp.kind = kind;
    kp.Dl = Dl;
    kp.Dg = Dg;
    for (int d = 0; d < D; ++d) {
        kp.colmap[d] = (d < Dl) ? (colmap_h ? colmap_h[d] : d) : 0;
        kp.inv_len[d] = 1.0 / length_h[nlen == 1 ? 0 : d];
    }
    kp.nugget = nugget;
    return DGPAMD_OK;
}

#define SQRT5 2.23606797749978969641

// one-dimensional factors of the correlation functions on SCALED differences
__device__ __forceinline__ void corr_accum_sexp(double d, double &s) { s = fma(d, d, s); }
__device__ __forceinline__ void corr_accum_matern(double d, double &prod, double &s) {
    double r = fabs(d);
    prod *= fma(r, fma(r, 5.0 / 3.0, SQRT5), 1.0);
    s += r;
}

// exp(-x) for x >= ~-1 (the arguments of the correlation functions and of the linked-GP pair loop are sums of squares / distances), with full-rate instructions only: the rounding to the nearest integer is the
// "1.5 * 2^52" addition (the integer then sits in the low word of the sum: no v_rndne / v_cvt), the scaling by 2^k a
// multiplication by a double whose exponent field is written with integer arithmetic (no v_ldexp).  On gfx950 v_rndne_f64,
// v_cvt_i32_f64 and v_ldexp_f64 issue at a quarter of v_fma_f64's rate: three of them cost as much as the twelve
// multiply-adds of the polynomial (PMC: 31 VALU instructions per pair but 78 % of the issue cycles).  k is clamped at -1022:
// arguments beyond ~708 give ~1e-308 instead of 0, which the weights multiply into nothing.
__device__ __forceinline__ double exp_negated(double x) {
    const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
    const double kf = fma(x, -1.44269504088896338700e+00, MAGIC);
    const double k = kf - MAGIC;
    double r = fma(k, -6.93147180369123816490e-01, -x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ki = __double2loint(kf);               // k as a two's-complement integer
    ki = ki < -1022 ? -1022 : ki;
    return p * __hiloint2double((ki + 1023) << 20, 0);
}

// exp_negated with every Horner step pinned to ONE three-operand v_fma_f64 whose constant sits in an SGPR pair.  Left to itself the
// compiler (inside vecchia_row4_kernel's pair loop) kept the nine polynomial constants in VGPRs and emitted each step as
// v_mov_b64 tmp, C ; v_fmac_f64 tmp, p, r -- the two-address form needs the constant copied into its destination first -- nine moves
// per exponential, 36 of the 290 VALU instructions of a pass.  The same operations in the same order: the same bits.
#define EXPN_STEP(P, R, C) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(P) : "v"(P), "v"(R), "s"((double)(C)))
__device__ __forceinline__ double exp_negated_v3(double x) {
    const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
    const double kf = fma(x, -1.44269504088896338700e+00, MAGIC);
    const double k = kf - MAGIC;
    double r = fma(k, -6.93147180369123816490e-01, -x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;
    EXPN_STEP(p, r, 2.50521083854417187751e-08);
    EXPN_STEP(p, r, 2.75573192239858906526e-07);
    EXPN_STEP(p, r, 2.75573192239858906526e-06);
    EXPN_STEP(p, r, 2.48015873015873015873e-05);
    EXPN_STEP(p, r, 1.98412698412698412698e-04);
    EXPN_STEP(p, r, 1.38888888888888888889e-03);
    EXPN_STEP(p, r, 8.33333333333333333333e-03);
    EXPN_STEP(p, r, 4.16666666666666666667e-02);
    EXPN_STEP(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ki = __double2loint(kf);               // k as a two's-complement integer
    ki = ki < -1022 ? -1022 : ki;
    return p * __hiloint2double((ki + 1023) << 20, 0);
}

// The same with a table tab[j] = 2^(j/N) (in LDS): exp(-x) = 2^e tab[j] exp(r), k = round(-N x / ln 2) = N e + j,
// |r| <= ln 2 / 2N, so a short polynomial replaces the degree-12 one (N = 64: degree 5, truncation 3.5e-17, 10 double-precision
// instructions instead of 17; N = 256, since round 5: degree 4, nine), plus five integer ones and one LDS read that do not occupy the double-precision units.  For a
// kernel whose inner loop leaves the LDS pipe idle (linkgp_Jsexp2_kernel; in one that feeds MFMA operands from LDS the table
// reads cost more than they saved, round 2).
#define EXPN_TAB 256   // entries of exp_negated_tab's table (tab[j] = 2^(j / EXPN_TAB))
#ifndef KM_EXP_TAB
#define KM_EXP_TAB 1   // K assembly and the gradient reductions take their exponential from the table form (0: the library's exp, rounds 1-5)
#endif
__device__ __forceinline__ double exp_negated_tab(double x, const double *tab) {
    // (round 5: 256 entries instead of 64 -- |r| <= ln 2 / 512, a degree-4 polynomial (truncation r^5 / 120 <= 3.8e-17) instead of the degree-5 one:
    //  nine double-precision instructions; the kernel that uses it is bound by exactly those)
    const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
    const double kf = fma(x, -3.69329930467574632284e+02, MAGIC);   // 256 / ln 2
    const double k = kf - MAGIC;
    double r = fma(k, -6.93147180369123816490e-01 / 256.0, -x);
    r = fma(k, -1.90821492927058770002e-10 / 256.0, r);
    double p = 4.16666666666666666667e-02;
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ki = __double2loint(kf);
    // k >> 8 from bits 8..39 of kf's mantissa (v_alignbit_b32: one instruction, as the 32-bit shift it replaces): the low word alone wraps
    // from x = 2^31 ln 2 / 256 = 5.8e6 on and the clamp below was passed by; this form is right up to |k| < 2^39, x = 1.5e9
    int e = (int)__builtin_amdgcn_alignbit((unsigned)__double2hiint(kf), (unsigned)ki, 8);
    e = e < -1021 ? -1021 : e;
    const double v = tab[ki & 255] * p;
    return __hiloint2double(__double2hiint(v) + (e << 20), __double2loint(v));
}

// exp_negated_tab in two halves, for a caller that issues the table read well ahead of its use (the LDS round trip sat in front of every
// exponential of linkgp_Jsexp2_kernel otherwise: the scheduler sinks a load to its use): the same operations, the same bits.
__device__ __forceinline__ void exp_negated_tab_begin(double x, const double *tab, double &kf, double &t) {
    kf = fma(x, -3.69329930467574632284e+02, 6755399441055744.0);
    t = tab[__double2loint(kf) & (EXPN_TAB - 1)];
}
__device__ __forceinline__ double exp_negated_tab_end(double x, double kf, double t) {
    const double k = kf - 6755399441055744.0;
    double r = fma(k, -6.93147180369123816490e-01 / 256.0, -x);
    r = fma(k, -1.90821492927058770002e-10 / 256.0, r);
    double p = 4.16666666666666666667e-02;
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int e = (int)__builtin_amdgcn_alignbit((unsigned)__double2hiint(kf), (unsigned)__double2loint(kf), 8);   // (k >> 8 for |k| < 2^39: see exp_negated_tab)
    e = e < -1021 ? -1021 : e;
    const double v = t * p;
    return __hiloint2double(__double2hiint(v) + (e << 20), __double2loint(v));
}

// lower-triangle tile index t -> (bi, bj), bi >= bj, t = bi(bi+1)/2 + bj
__device__ __forceinline__ void tri_decode(int t, int &bi, int &bj) {
    int b = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) / 2 <= t) ++b;
    while (b * (b + 1) / 2 > t) --b;
    bi = b;
    bj = t - b * (b + 1) / 2;
}

// ---- cross-file internals -------------------------------------------------
struct KmatArgs {
    KernParams kp;
    int64_t n;
    const double *Xloc;
    int64_t ldloc, stride_loc;
    const double *Xglob;
    const double *W;
    double *K;
    int64_t ldk, stride_k;
    int full;
    const double *Y;
    int64_t ldy, stride_y;
    int r;
    const int32_t *pred;   // null, or a device word: the launch does nothing when it is non-zero
    int32_t *zero_ptr;     // null, or words to clear on the way (the factorisation's synchronisation block: saves its memset launch)
    int zero_words;
    int nbk;               // 64-blocks per dimension of the launch's tile grid (set by the launchers)
};
int launch_kmatrix(dgpamd_ctx *ctx, const KmatArgs &a, int batch);
int launch_kmatrix_multi(dgpamd_ctx *ctx, const KmatArgs *dev_args, const KmatArgs *host_args, int count);   // same n / mode
int build_kmat_args(dgpamd_ctx *ctx, KmatArgs &a, int kind, int64_t n, const double *Xloc, int64_t ldloc,
                    int64_t stride_loc, const int32_t *colmap_h, int Dl, const double *Xglob, int Dg,
                    const double *length_h, int nlen, double nugget, const double *W, double *K, int64_t ldk,
                    int64_t stride_k, int full, const double *Y, int64_t ldy, int64_t stride_y, int r, int batch);
// When `post` is given and the factorisation ran as the one-launch kernel, its two tiny follow-up launches (results out of
// the workspace, -alpha into row n of the inverse) are left to the caller, who folds them into a kernel of its own:
// post->pending = 1 and the pointers say where the results are.
struct PotrfPost {
    int pending = 0;
    const double *ld_ws = nullptr;
    const int32_t *info_ws = nullptr;
    const int32_t *status = nullptr;
    int32_t *spare = nullptr;   // a word of the (cleared) synchronisation block that the factorisation does not use
};
int run_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet, int32_t *info,
              double *ws, double *T = nullptr, double *S = nullptr, PotrfPost *post = nullptr, bool sync_cleared = false);
// The words run_potrf would clear before its launch for this call (null / 0: none -- the per-step launches clear their own
// flags inside their graph): a kernel that runs just before on the same stream may clear them instead (KmatArgs::zero_ptr)
// and pass sync_cleared = true.
void potrf_sync_area(dgpamd_ctx *ctx, int64_t n, int batch, bool inv, double *ws, int32_t **ptr, int *words);   // T, S: fused inverse (dgpamd_potrf_inv)
size_t potrf_ws_doubles(int64_t n, int batch);
// vecchia_llik for `batch` input sets into caller-provided buffers (partial: batch x n x 2 doubles, out: batch x 2 = (quad,
// logdet) per set), no allocation; launches made while ctx->pred is set are predicated on it (dgpamd_ess_queue)
int vecchia_llik_batch_into(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, int64_t x_stride, int batch,
                            const double *y, const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                            const double *nugget_diag, double *partial, double *out);
int ensure_pinned(dgpamd_ctx *ctx, size_t bytes);   // grow the context's pinned staging buffer
int ensure_devargs(dgpamd_ctx *ctx, size_t bytes);  // grow the device / pinned argument arrays
// One scratch buffer per purpose and context: its users follow each other on the context's stream, so the next call may overwrite
// it (a stream-ordered hipMallocAsync / hipFreeAsync pair per call cost ~0.2 ms of host time, milliseconds for 60 MB).
int ctx_scratch(dgpamd_ctx *ctx, int which, size_t bytes, void **p);

