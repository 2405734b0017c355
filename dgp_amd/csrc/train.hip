// Training-path pipelines (SURVEY 8 a4,a5,a8): ESS proposals, the batched ESS
// target log-likelihood, and the in-flight derivative reductions of the M-step.
#include "common.hpp"

#include <math.h>

// ---------------------------------------------------------------------------
// a4  update_f (functions.py:203-208), batched over speculative angles
// ---------------------------------------------------------------------------
struct ProposeArgs {
    const double *F, *NU;
    double *FP;
    int64_t count;   // n*M
    double th[DGPAMD_MAXB];
};
__global__ __launch_bounds__(256) void ess_propose_kernel(ProposeArgs a) {
    const int b = blockIdx.y;
    // (cosine and sine on the device, as the device queue's ess_prepare takes them: the host loop and the queue then propose the same bits for the same
    //  angle -- the host's libm here was one of the last-bit differences between the two paths, profiles/r06_mstep_host_edges.txt)
    const double c = cos(a.th[b]), s = sin(a.th[b]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * 256)
        a.FP[(int64_t)b * a.count + i] = a.F[i] * c + a.NU[i] * s;
}

extern "C" int dgpamd_ess_propose(dgpamd_ctx *ctx, int64_t n, int M, const double *F, const double *NU,
                                  const double *theta_h, int batch, double *FP) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || M <= 0 || !F || !NU || !theta_h || !FP) BAD_ARG(ctx, "null pointer or empty block");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    ProposeArgs a;
    a.F = F; a.NU = NU; a.FP = FP; a.count = n * M;
    for (int b = 0; b < batch; ++b) a.th[b] = theta_h[b];
    int64_t blocks = (a.count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ess_propose_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// a5  log_likelihood_func (kernel_class.py:481-492), batched
// ---------------------------------------------------------------------------
__global__ void loglik_finish_kernel(const double *A, int64_t ld, int64_t stride_a, int64_t n, const double *logdet,
                                     double scale, double *ll, int batch) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double quad = -A[(int64_t)b * stride_a + n * ld + n];
    ll[b] = -0.5 * ((double)n * log(scale) + logdet[b] + quad / scale);
}

extern "C" int dgpamd_loglik(dgpamd_ctx *ctx, int kind, int64_t n, const double *Xloc, int64_t ldloc,
                             int64_t stride_loc, const int32_t *colmap_h, int Dl, const double *Xglob, int Dg,
                             const double *length_h, int nlen, double nugget, const double *W, double scale,
                             const double *y, double *A, int64_t stride_a, int batch, double *ll, int32_t *info,
                             void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!y || !A || !ll || !info || !work) BAD_ARG(ctx, "null pointer");
    if (!(scale > 0.0)) BAD_ARG(ctx, "scale must be positive");
    const int64_t Np = padded_dim(n);
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    KmatArgs a;
    int rc = build_kmat_args(ctx, a, kind, n, Xloc, ldloc, stride_loc, colmap_h, Dl, Xglob, Dg, length_h, nlen, nugget,
                             W, A, Np, stride_a, 0, y, n, 0, 1, batch);
    if (rc) return rc;
    double *ws = (double *)work;
    potrf_sync_area(ctx, n, batch, false, ws, &a.zero_ptr, &a.zero_words);   // (cleared by the assembly kernel on the way)
    rc = launch_kmatrix(ctx, a, batch);
    if (rc) return rc;
    double *logdet = ws + (size_t)batch * (Np / 64) * 4096;
    rc = run_potrf(ctx, n, A, stride_a, batch, logdet, info, ws, nullptr, nullptr, nullptr, a.zero_ptr != nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(loglik_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, A, Np, stride_a, n, logdet, scale, ll,
                       batch);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// The closing arithmetic of a5 alone, for buffers factored elsewhere (e.g. as extra matrices of another batched call):
// ll[b] = -0.5 (n log scale + logdet[b] + y'K^-1y / scale), y'K^-1y = -corner of the augmented buffer.
extern "C" int dgpamd_loglik_finish(dgpamd_ctx *ctx, int64_t n, const double *A, int64_t stride_a, int batch, const double *logdet,
                                    double scale, double *ll) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !A || !logdet || !ll) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > 64) BAD_ARG(ctx, "need 1 <= batch <= 64");
    if (!(scale > 0.0)) BAD_ARG(ctx, "scale must be positive");
    hipLaunchKernelGGL(loglik_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, A, padded_dim(n), stride_a, n, logdet, scale, ll, batch);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// a8  derivative reductions of kernel.llik (kernel_class.py:414-427) without
//     ever storing dK:  tr_p = sum Kinv o dK_p ,  quad_p = alpha^T dK_p alpha.
//     dK/dlog g_d = c_d K: sexp c_d = 2 r_d^2 (functions.py:36-45);
//     matern c_d = (5/3) r^2 (1+sqrt5 r)/(1+sqrt5 r+5/3 r^2) (functions.py:71-93);
//     shared lengthscale: sum over d.  dK/dlog eta = eta diag(W) (kernel_class.py:346-351).
// ---------------------------------------------------------------------------
struct GradArgs {
    KernParams kp;
    int64_t n;
    const double *Xloc;
    int64_t ldloc;
    const double *Xglob;
    const double *W;
    const double *Ainv;
    int64_t ld;
    int shared_len, nugget_est, P;
    double *partial;
    const double *alpha_col;   // null: -alpha is row n of Ainv; else -alpha_i = alpha_col[i * ld] (column n of L^-T, dgpamd_potrf_inv)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int KIND>
__device__ __forceinline__ double dcoef(double df) {
    if (KIND == DGPAMD_SEXP) return 2.0 * df * df;
    double r = fabs(df);
    double e1 = fma(r, SQRT5, 1.0), e2 = (5.0 / 3.0) * r * r;
    // e1 + e2 >= 1: the hardware reciprocal and two Newton rounds (<= 1 ulp) instead of the IEEE division sequence, which
    // was most of this kernel's instructions (five divisions per matrix entry)
    const double d = e1 + e2;
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return e2 * e1 * x;
}

template <int KIND>
__device__ __forceinline__ void grad_reduce_body(const GradArgs &a) {
    extern __shared__ double lds[];
    const int D = a.kp.Dl + a.kp.Dg;
    double *XiT = lds, *XjT = lds + D * 64;
    double *red = lds + 2 * D * 64;   // [4 waves][2P]
#if KM_EXP_TAB
    double *etab = red + 4 * 2 * a.P;   // [EXPN_TAB]: exp_negated_tab's table, as kmatrix_body builds it (the same bits in K here and there)
    for (int j = threadIdx.x; j < EXPN_TAB; j += 256) etab[j] = exp2((double)j * (1.0 / EXPN_TAB));
#endif
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64, n = a.n;
    const int P2 = 2 * a.P;

    for (int idx = tid; idx < 64 * D; idx += 256) {
        int row = idx / D, d = idx - row * D;
        int64_t gi = i0 + row, gj = j0 + row;
        double vi = 0.0, vj = 0.0;
        if (d < a.kp.Dl) {
            int c = a.kp.colmap[d];
            if (gi < n) vi = a.Xloc[gi * a.ldloc + c];
            if (gj < n) vj = a.Xloc[gj * a.ldloc + c];
        } else {
            int c = d - a.kp.Dl;
            if (gi < n) vi = a.Xglob[gi * a.kp.Dg + c];
            if (gj < n) vj = a.Xglob[gj * a.kp.Dg + c];
        }
        XiT[d * 64 + row] = vi * a.kp.inv_len[d];
        XjT[d * 64 + row] = vj * a.kp.inv_len[d];
    }
    __syncthreads();

    double s[4][4], pr[4][4];
    double num[4][4];   // shared lengthscale, Matern: sum_d e1_d e2_d prod_{d' != d} q_d'  (so that c K = num exp(-sqrt5 s), see below)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[p][q] = 0.0;
            pr[p][q] = 1.0;
            num[p][q] = 0.0;
        }
    if (a.shared_len && KIND != DGPAMD_SEXP) {
        // ONE pass for an isotropic Matern node: the derivative coefficient of dimension d is e1 e2 / q with q = e1 + e2 the
        // same polynomial that K's product runs over, so sum_d c_d K = [sum_d e1_d e2_d prod_{d' != d} q_d'] exp(-sqrt5 s):
        // a two-term recurrence beside the product, no division, no second pass (157 -> ~90 VALU instructions per entry).
        for (int d = 0; d < D; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                xi[p] = XiT[d * 64 + ty + 16 * p];
                xj[p] = XjT[d * 64 + tx + 16 * p];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double r = fabs(xi[p] - xj[q]);
                    const double qd_ = fma(r, fma(r, 5.0 / 3.0, SQRT5), 1.0);   // = e1 + e2, as corr_accum_matern forms it
                    const double e1 = fma(r, SQRT5, 1.0), e2 = (5.0 / 3.0) * r * r;
                    num[p][q] = fma(num[p][q], qd_, e1 * e2 * pr[p][q]);
                    pr[p][q] *= qd_;
                    s[p][q] += r;
                }
        }
    } else
    for (int d = 0; d < D; ++d) {
        double xi[4], xj[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            xi[p] = XiT[d * 64 + ty + 16 * p];
            xj[p] = XjT[d * 64 + tx + 16 * p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double df = xi[p] - xj[q];
                if (KIND == DGPAMD_SEXP)
                    corr_accum_sexp(df, s[p][q]);
                else
                    corr_accum_matern(df, pr[p][q], s[p][q]);
            }
    }
    // weights w1 = wt K_ij Kinv_ij , w2 = wt K_ij alpha_i alpha_j  (off-diagonal, in range)
    const double wt = (bi == bj) ? 1.0 : 2.0;
    double w1[4][4], w2[4][4], ai[4], aj[4];
    const double *arow = a.Ainv + n * a.ld;   // row n = -alpha^T
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int64_t gi = i0 + ty + 16 * p, gj = j0 + tx + 16 * p;
        ai[p] = gi < n ? -(a.alpha_col ? a.alpha_col[gi * a.ld] : arow[gi]) : 0.0;
        aj[p] = gj < n ? -(a.alpha_col ? a.alpha_col[gj * a.ld] : arow[gj]) : 0.0;
    }
    double trn = 0.0, qn = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t gi = i0 + ty + 16 * p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t gj = j0 + tx + 16 * q;
            // K_ij -- or, with one shared lengthscale, directly c_ij K_ij (sexp: c = sum_d 2 df^2 = 2 s; Matern: num / prod)
            double kv;
#if KM_EXP_TAB
            if (KIND == DGPAMD_SEXP)
                kv = (a.shared_len ? 2.0 * s[p][q] : 1.0) * exp_negated_tab(s[p][q], etab);
            else
                kv = (a.shared_len ? num[p][q] : pr[p][q]) * exp_negated_tab(SQRT5 * s[p][q], etab);
#else
            if (KIND == DGPAMD_SEXP)
                kv = (a.shared_len ? 2.0 * s[p][q] : 1.0) * exp(-s[p][q]);
            else
                kv = (a.shared_len ? num[p][q] : pr[p][q]) * exp(-SQRT5 * s[p][q]);
#endif
            const bool in = gi < n && gj < n;
            double kinv = in ? a.Ainv[gi * a.ld + gj] : 0.0;
            if (!in || gi == gj) kv = 0.0;
            w1[p][q] = wt * kv * kinv;
            w2[p][q] = wt * kv * ai[p] * aj[q];
            if (a.nugget_est && in && gi == gj) {
                double w = a.W ? a.W[gi] : 1.0;
                trn += a.kp.nugget * w * kinv;
                qn += a.kp.nugget * w * ai[p] * ai[p];
            }
        }
    }
    const int npl = a.shared_len ? 1 : D;
    if (a.shared_len) {   // (the weights already carry the coefficient)
        double tr = 0.0, qd = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tr += w1[p][q];
                qd += w2[p][q];
            }
        tr = wave_sum(tr);
        qd = wave_sum(qd);
        if (lane == 0) {
            red[wave * P2 + 0] = tr;
            red[wave * P2 + a.P] = qd;
        }
    } else {
        for (int d = 0; d < D; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                xi[p] = XiT[d * 64 + ty + 16 * p];
                xj[p] = XjT[d * 64 + tx + 16 * p];
            }
            double tr = 0.0, qd = 0.0;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double c = dcoef<KIND>(xi[p] - xj[q]);
                    tr = fma(c, w1[p][q], tr);
                    qd = fma(c, w2[p][q], qd);
                }
            tr = wave_sum(tr);
            qd = wave_sum(qd);
            if (lane == 0) {
                red[wave * P2 + d] = tr;
                red[wave * P2 + a.P + d] = qd;
            }
        }
    }
    if (a.nugget_est) {
        trn = wave_sum(trn);
        qn = wave_sum(qn);
        if (lane == 0) {
            red[wave * P2 + npl] = trn;
            red[wave * P2 + a.P + npl] = qn;
        }
    }
    __syncthreads();
    if (tid < P2) a.partial[(int64_t)blockIdx.x * P2 + tid] = red[tid] + red[P2 + tid] + red[2 * P2 + tid] + red[3 * P2 + tid];
}

template <int KIND>
__global__ __launch_bounds__(256) void grad_reduce_kernel(GradArgs a) {
    grad_reduce_body<KIND>(a);
}

// Several nodes in one launch (grid.z = node), arguments from a device array; `out` of node c follows its GradArgs.
struct GradMulti {
    GradArgs a;
    double *out;
};
__global__ __launch_bounds__(256) void grad_reduce_multi_kernel(const GradMulti *args) {
    const GradArgs &a = args[blockIdx.z].a;
    if (a.kp.kind == DGPAMD_SEXP)
        grad_reduce_body<DGPAMD_SEXP>(a);
    else
        grad_reduce_body<DGPAMD_MATERN25>(a);
}
// ... and with up to three nodes' arguments by value (see kmatrix_multi_val_kernel)
struct GradMulti3 {
    GradMulti g[3];
};
static_assert(sizeof(GradMulti3) <= 3600, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void grad_reduce_multi_val_kernel(GradMulti3 v) {
    const GradArgs &a = v.g[blockIdx.z].a;
    if (a.kp.kind == DGPAMD_SEXP)
        grad_reduce_body<DGPAMD_SEXP>(a);
    else
        grad_reduce_body<DGPAMD_MATERN25>(a);
}

__global__ __launch_bounds__(256) void grad_final_kernel(const double *partial, int ntiles, int P2, double *out) {
    __shared__ double sm[4];
    const int idx = blockIdx.x, tid = threadIdx.x;
    double v = 0.0;
    for (int t = tid; t < ntiles; t += 256) v += partial[(int64_t)t * P2 + idx];
    v = wave_sum(v);
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) out[idx] = sm[0] + sm[1] + sm[2] + sm[3];
}

// What the LAST block of grad_final_multi_kernel does when the round's other small launches are folded into it (the
// one-launch factorisation ran): the factorisation's results out of its workspace into every node's row of dev_out, then all
// rows into pinned host memory and the sequence word the host spins on.
struct LlikFinish {
    int enable;
    int32_t *counter;        // zero between launches (the last block puts it back)
    const double *ld_ws;
    const int32_t *info_ws, *status;
    const double *A;
    int64_t ld, stride_a, n;
    double *dev_out;
    int64_t stride_out;
    int batch;
    double *host;
    unsigned long long *flag, seq;
};
__device__ __forceinline__ void grad_final_multi_body(const GradMulti &g, int ntiles, const LlikFinish &f);
__global__ __launch_bounds__(256) void grad_final_multi_kernel(const GradMulti *args, int ntiles, LlikFinish f) {
    grad_final_multi_body(args[blockIdx.y], ntiles, f);
}
__global__ __launch_bounds__(256) void grad_final_multi_val_kernel(GradMulti3 v, int ntiles, LlikFinish f) {
    grad_final_multi_body(v.g[blockIdx.y], ntiles, f);
}
__device__ __forceinline__ void grad_final_multi_body(const GradMulti &g, int ntiles, const LlikFinish &f) {
    __shared__ double sm[4];
    __shared__ int last;
    const int idx = blockIdx.x, tid = threadIdx.x, P2 = 2 * g.a.P;
    if (idx < P2) {
        double v = 0.0;
        for (int t = tid; t < ntiles; t += 256) v += g.a.partial[(int64_t)t * P2 + idx];
        v = wave_sum(v);
        if ((tid & 63) == 0) sm[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) g.out[idx] = sm[0] + sm[1] + sm[2] + sm[3];
    }
    if (!f.enable) return;
    if (tid == 0) {
        __threadfence();
        last = atomicAdd(f.counter, 1) == (int)(gridDim.x * gridDim.y) - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (tid < f.batch) {
        const int b = tid;
        f.dev_out[b * f.stride_out] = __hip_atomic_load(f.ld_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f.dev_out[b * f.stride_out + 1] = -f.A[(int64_t)b * f.stride_a + f.n * f.ld + f.n];
        f.dev_out[b * f.stride_out + 2] = (double)(__hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                        ? -1 : __hip_atomic_load(f.info_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (tid == 0) *f.counter = 0;
    __threadfence();
    __syncthreads();
    const int nd = (int)(f.batch * f.stride_out);
    for (int i = tid; i < nd; i += 256) f.host[i] = __hip_atomic_load(f.dev_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(f.flag, f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" size_t dgpamd_grad_workspace(int64_t n, int nparam) {
    int64_t nb = (n + 63) / 64;
    return (size_t)(nb * (nb + 1) / 2) * 2 * (size_t)nparam * sizeof(double);
}

extern "C" int dgpamd_grad_reduce(dgpamd_ctx *ctx, int kind, int64_t n, const double *Xloc, int64_t ldloc,
                                  const int32_t *colmap_h, int Dl, const double *Xglob, int Dg, const double *length_h,
                                  int nlen, double nugget, const double *W, int nugget_est, const double *Ainv,
                                  double *out, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !Ainv || !out || !work || !length_h) BAD_ARG(ctx, "null pointer or n <= 0");
    if ((Dl > 0 && !Xloc) || (Dg > 0 && !Xglob)) BAD_ARG(ctx, "null input pointer");
    GradArgs a;
    int rc = fill_kern_params(ctx, a.kp, kind, colmap_h, Dl, Dg, length_h, nlen, nugget);
    if (rc) return rc;
    const int D = Dl + Dg;
    a.n = n; a.Xloc = Xloc; a.ldloc = ldloc; a.Xglob = Xglob; a.W = W; a.Ainv = Ainv; a.ld = padded_dim(n);
    a.shared_len = (nlen == 1); a.nugget_est = nugget_est ? 1 : 0;
    a.P = (nlen == 1 ? 1 : D) + a.nugget_est;
    a.partial = (double *)work;
    a.alpha_col = nullptr;
    const int nb = (int)((n + 63) / 64), ntiles = nb * (nb + 1) / 2;
    size_t shm = ((size_t)2 * D * 64 + 4 * 2 * a.P + KM_EXP_TAB * EXPN_TAB) * sizeof(double);
    if (kind == DGPAMD_SEXP)
        hipLaunchKernelGGL(grad_reduce_kernel<DGPAMD_SEXP>, dim3(ntiles), dim3(256), shm, ctx->stream, a);
    else
        hipLaunchKernelGGL(grad_reduce_kernel<DGPAMD_MATERN25>, dim3(ntiles), dim3(256), shm, ctx->stream, a);
    hipLaunchKernelGGL(grad_final_kernel, dim3(2 * a.P), dim3(256), 0, ctx->stream, (const double *)work, ntiles,
                       2 * a.P, out);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}


// ----------------------------------------------------------------------------
// lock-step M-step: the device part of kernel.llik for several nodes in one call
// ----------------------------------------------------------------------------
__global__ void llik_pack_kernel(const double *logdet, const int32_t *info, const double *A, int64_t ld, int64_t stride_a,
                                 int64_t n, double *out, int64_t stride_out, int batch) {
    const int b = threadIdx.x;
    if (b < batch) {
        out[b * stride_out] = logdet[b];
        out[b * stride_out + 1] = -A[(int64_t)b * stride_a + n * ld + n];   // y' K^-1 y sits negated in the corner
        out[b * stride_out + 2] = (double)info[b];
    }
}

// The same with the one-launch factorisation's two follow-ups folded in (PotrfPost): results out of the workspace (a lost
// hand-off becomes info = -1), -alpha = column n of T into row n of the inverse.
__global__ void llik_post_kernel(const double *T, double *S, const double *A, int64_t ld, int64_t stride_a, int64_t n,
                                 const double *ld_ws, const int32_t *info_ws, const int32_t *status, double *logdet, int32_t *info,
                                 double *out, int64_t stride_out) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) S[(int64_t)b * stride_a + n * ld + j] = T[(int64_t)b * stride_a + j * ld + n];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double l = __hip_atomic_load(ld_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t f = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              ? -1 : __hip_atomic_load(info_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        logdet[b] = l;
        info[b] = f;
        out[b * stride_out] = l;
        out[b * stride_out + 1] = -A[(int64_t)b * stride_a + n * ld + n];
        out[b * stride_out + 2] = (double)f;
    }
}

// Results straight into pinned host memory, then a sequence word: the host spins on it (microseconds) instead of sleeping in
// a stream synchronisation (tens of microseconds to wake up), once per round of the lock-step M-step.
__global__ void publish_host_kernel(const double *src, double *host, int nd, unsigned long long *flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < nd; i += blockDim.x) host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The two halves of dgpamd_llik_batch: everything queued (launch), then the wait for the results (wait).  Between the two the caller may do host work that
// touches neither the evaluations' inputs nor this context's other blocking calls' staging (the results land in a pinned buffer of their own).
extern "C" int dgpamd_llik_batch_launch(dgpamd_ctx *ctx, int64_t n, int batch, const dgpamd_node *nodes, double *A, double *T,
                                        double *Ainv, int64_t stride_a, void *work, void *grad_work, double *dev_out, int64_t stride_out) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (ctx->llik_pending) BAD_ARG(ctx, "an evaluation is still in flight: dgpamd_llik_batch_wait first");
    if (n <= 0 || !nodes || !A || !T || !Ainv || !work || !grad_work || !dev_out) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    const int64_t Np = padded_dim(n);
    if (n + 1 > Np) BAD_ARG(ctx, "no room for the augmented row");
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    // argument arrays of the two multi-node launches: [KmatArgs x batch | GradMulti x batch], staged in pinned memory
    const size_t need = (size_t)batch * (sizeof(KmatArgs) + sizeof(GradMulti));
    int rc = ensure_devargs(ctx, need);
    if (rc) return rc;
    // (the previous call's argument copy has been consumed: every call ends with the stream drained -- unless it left early)
    if (ctx->args_inflight) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    KmatArgs *ka = reinterpret_cast<KmatArgs *>(ctx->hostargs);
    GradMulti *ga = reinterpret_cast<GradMulti *>(ctx->hostargs + (size_t)batch * sizeof(KmatArgs));
    const int nb = (int)((n + 63) / 64), ntiles = nb * (nb + 1) / 2;
    int Pmax = 0, Dmax = 0;
    for (int b = 0; b < batch; ++b) {
        const dgpamd_node &nd = nodes[b];
        const int D = nd.Dl + nd.Dg;
        const int P = (nd.nlen == 1 ? 1 : D) + (nd.nugget_est ? 1 : 0);
        if (3 + 2 * P > stride_out) BAD_ARG(ctx, "stride_out too small");
        Pmax = P > Pmax ? P : Pmax;
        Dmax = D > Dmax ? D : Dmax;
        rc = build_kmat_args(ctx, ka[b], nd.kind, n, nd.Xloc, nd.ldloc, 0, (const int32_t *)nd.colmap, nd.Dl, nd.Xglob, nd.Dg,
                             nd.length, nd.nlen, nd.nugget, nd.W, A + (int64_t)b * stride_a, Np, 0, 0, nd.y, n, 0, 1, 1);
        if (rc) return rc;
    }
    const size_t gw = dgpamd_grad_workspace(n, Pmax);   // per node (the caller provides batch of them)
    for (int b = 0; b < batch; ++b) {
        const dgpamd_node &nd = nodes[b];
        GradArgs &g = ga[b].a;
        g.kp = ka[b].kp;
        g.n = n; g.Xloc = nd.Xloc; g.ldloc = nd.ldloc; g.Xglob = nd.Xglob; g.W = nd.W;
        g.Ainv = Ainv + (int64_t)b * stride_a; g.ld = Np;
        g.shared_len = (nd.nlen == 1); g.nugget_est = nd.nugget_est ? 1 : 0;
        g.P = (nd.nlen == 1 ? 1 : nd.Dl + nd.Dg) + g.nugget_est;
        g.partial = reinterpret_cast<double *>(reinterpret_cast<char *>(grad_work) + (size_t)b * gw);
        g.alpha_col = nullptr;
        ga[b].out = dev_out + b * stride_out + 3;
    }
    potrf_sync_area(ctx, n, batch, true, (double *)work, &ka[0].zero_ptr, &ka[0].zero_words);   // (node 0's blocks clear it)
    const bool cleared = ka[0].zero_ptr != nullptr;   // = the factorisation will run as the one-launch kernel
    if (cleared)   // (its -alpha is read where it is, column n of L^-T: no copy into the inverse's row n)
        for (int b = 0; b < batch; ++b) ga[b].a.alpha_col = T + (int64_t)b * stride_a + n;
    // one to three nodes (most rounds of an M-step): the arguments ride in the launches themselves, no copy into device memory (DGPAMD_LLIK_ARGS_COPY=1: always copy)
    static const bool always_copy = getenv("DGPAMD_LLIK_ARGS_COPY") != nullptr && atoi(getenv("DGPAMD_LLIK_ARGS_COPY")) != 0;
    const bool by_value = batch <= 3 && !always_copy;
    const KmatArgs *kd = nullptr;
    const GradMulti *gd = nullptr;
    if (!by_value) {
        ctx->args_inflight = 1;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->devargs, ctx->hostargs, need, hipMemcpyHostToDevice, ctx->stream));
        kd = reinterpret_cast<const KmatArgs *>(ctx->devargs);
        gd = reinterpret_cast<const GradMulti *>(ctx->devargs + (size_t)batch * sizeof(KmatArgs));
    }
    rc = launch_kmatrix_multi(ctx, kd, ka, batch);
    if (rc) return rc;
    double *logdet = dev_out + (int64_t)batch * stride_out;
    int32_t *info = reinterpret_cast<int32_t *>(logdet + batch);
    PotrfPost post;
    rc = run_potrf(ctx, n, A, stride_a, batch, logdet, info, (double *)work, T, Ainv, &post, cleared);   // factor + inverse, one sweep
    if (rc) return rc;
    const int nd = (int)(batch * stride_out);
    const size_t bytes = (size_t)nd * sizeof(double);
    if (ctx->llik_pinned_bytes < bytes + 64) {   // (a staging buffer of this call's own: dgpamd_fetch and friends may run between launch and wait)
        if (ctx->llik_pinned) (void)hipHostFree(ctx->llik_pinned);
        ctx->llik_pinned = nullptr;
        ctx->llik_pinned_bytes = 0;
        const size_t want = ((bytes + 64 + 4095) / 4096) * 4096;
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->llik_pinned, want, hipHostMallocDefault));
        ctx->llik_pinned_bytes = want;
    }
    unsigned long long *flag = reinterpret_cast<unsigned long long *>(ctx->llik_pinned + (ctx->llik_pinned_bytes / sizeof(double) - 1));
    *flag = 0;   // (no evaluation is in flight: nobody else touches this buffer)
    const unsigned long long seq = ++ctx->host_seq;
    LlikFinish fin;
    memset(&fin, 0, sizeof(fin));
    const bool fused = post.pending && cleared;
    if (fused) {   // results, packing and publication ride in the last block of the final reduction
        fin.enable = 1;
        fin.counter = post.spare;   // (zero: the synchronisation block was cleared before the factorisation)
        fin.ld_ws = post.ld_ws; fin.info_ws = post.info_ws; fin.status = post.status;
        fin.A = A; fin.ld = Np; fin.stride_a = stride_a; fin.n = n;
        fin.dev_out = dev_out; fin.stride_out = stride_out; fin.batch = batch;
        fin.host = ctx->llik_pinned; fin.flag = flag; fin.seq = seq;
    } else if (post.pending)
        hipLaunchKernelGGL(llik_post_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, ctx->stream, (const double *)T,
                           Ainv, (const double *)A, Np, stride_a, n, post.ld_ws, post.info_ws, post.status, logdet, info, dev_out,
                           stride_out);
    else
        hipLaunchKernelGGL(llik_pack_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)logdet,
                           (const int32_t *)info, (const double *)A, Np, stride_a, n, dev_out, stride_out, batch);
    {
        const size_t shm = ((size_t)2 * Dmax * 64 + 4 * 2 * Pmax + KM_EXP_TAB * EXPN_TAB) * sizeof(double);
        if (by_value) {
            GradMulti3 v;
            for (int c = 0; c < 3; ++c) v.g[c] = ga[c < batch ? c : 0];
            hipLaunchKernelGGL(grad_reduce_multi_val_kernel, dim3(ntiles, 1, batch), dim3(256), shm, ctx->stream, v);
            hipLaunchKernelGGL(grad_final_multi_val_kernel, dim3(2 * Pmax, batch), dim3(256), 0, ctx->stream, v, ntiles, fin);
        } else {
            hipLaunchKernelGGL(grad_reduce_multi_kernel, dim3(ntiles, 1, batch), dim3(256), shm, ctx->stream, gd);
            hipLaunchKernelGGL(grad_final_multi_kernel, dim3(2 * Pmax, batch), dim3(256), 0, ctx->stream, gd, ntiles, fin);
        }
        LAUNCH_CHECK(ctx);
    }
    if (!fused)
        hipLaunchKernelGGL(publish_host_kernel, dim3(1), dim3(128), 0, ctx->stream, (const double *)dev_out, ctx->llik_pinned, nd, flag, seq);
    LAUNCH_CHECK(ctx);
    ctx->llik_pending = 1;
    ctx->llik_seq = seq;
    ctx->llik_bytes = bytes;
    ctx->llik_dev_out = dev_out;
    return DGPAMD_OK;
}

extern "C" int dgpamd_llik_batch_wait(dgpamd_ctx *ctx, double *host_out) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!ctx->llik_pending) BAD_ARG(ctx, "no evaluation in flight");
    if (!host_out) BAD_ARG(ctx, "null pointer");
    const unsigned long long seq = ctx->llik_seq;
    const size_t bytes = ctx->llik_bytes;
    const double *dev_out = ctx->llik_dev_out;
    unsigned long long *flag = reinterpret_cast<unsigned long long *>(ctx->llik_pinned + (ctx->llik_pinned_bytes / sizeof(double) - 1));
    ctx->llik_pending = 0;
    // spin on the sequence word; now and then make sure the stream is still alive (a fault would leave the word unwritten)
    for (unsigned long long it = 1;; ++it) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        __builtin_ia32_pause();
        if ((it & 0xfffff) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {   // everything ran: the word is there, or this memory is not coherent -- settle it the slow way
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                    HIP_TRY(ctx, hipMemcpy(ctx->llik_pinned, dev_out, bytes, hipMemcpyDeviceToHost));
                break;
            }
            if (q != hipErrorNotReady) HIP_TRY(ctx, q);
        }
    }
    ctx->args_inflight = 0;
    memcpy(host_out, ctx->llik_pinned, bytes);
    return DGPAMD_OK;
}

extern "C" int dgpamd_llik_batch(dgpamd_ctx *ctx, int64_t n, int batch, const dgpamd_node *nodes, double *A, double *T,
                                 double *Ainv, int64_t stride_a, void *work, void *grad_work, double *dev_out, double *host_out,
                                 int64_t stride_out) {
    if (ctx && !host_out) BAD_ARG(ctx, "null pointer or n <= 0");
    const int rc = dgpamd_llik_batch_launch(ctx, n, batch, nodes, A, T, Ainv, stride_a, work, grad_work, dev_out, stride_out);
    if (rc) return rc;
    return dgpamd_llik_batch_wait(ctx, host_out);
}


// ----------------------------------------------------------------------------
// a7  one elliptical-slice update, loop and all (imputation.py:81-119)
// ----------------------------------------------------------------------------
extern "C" int dgpamd_ess_update(dgpamd_ctx *ctx, int64_t n, int M, double *F, const double *NU, const dgpamd_node *node,
                                 double scale, double log_y, double *state, const double *uniforms, int nuni, int batch_first,
                                 int batch_next, double *FP, double *A, void *work, double *ll_dev, int32_t *info_dev,
                                 double *out) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || M <= 0 || !F || !NU || !node || !state || !FP || !A || !work || !ll_dev || !info_dev || !out)
        BAD_ARG(ctx, "null pointer or empty block");
    if (nuni < 0 || (nuni > 0 && !uniforms)) BAD_ARG(ctx, "bad uniform stream");
    if (batch_first <= 0 || batch_first > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch_first <= DGPAMD_MAXB");
    if (batch_next <= 0 || batch_next > batch_first) batch_next = batch_first;
    const int64_t Np = padded_dim(n);
    double theta = state[0], lo = state[1], hi = state[2];
    bool pending = state[3] != 0.0;
    int used = 0, proposals = 0, batches = 0;
    auto shrink = [&](double u) {   // imputation.py:115-119; numpy's uniform(lo, hi) = lo + (hi - lo) u
        if (theta < 0.0) lo = theta; else hi = theta;
        theta = lo + (hi - lo) * u;
    };
    auto finish = [&](int status, double ll, int info) {
        state[0] = theta; state[1] = lo; state[2] = hi; state[3] = pending ? 1.0 : 0.0;
        out[0] = status; out[1] = used; out[2] = proposals; out[3] = batches; out[4] = ll; out[5] = info;
        return DGPAMD_OK;
    };
    double th[DGPAMD_MAXB], blo[DGPAMD_MAXB], bhi[DGPAMD_MAXB], llh[DGPAMD_MAXB + DGPAMD_MAXB / 2 + 1];
    int B = batch_first;
    for (;;) {
        if (pending) {   // closing shrink of the previous, fully rejected batch
            if (used >= nuni) return finish(1, 0.0, 0);
            shrink(uniforms[used++]);
            pending = false;
            B = batch_next;
        }
        // speculative angles: theta followed by what the next rejections would produce
        int nb = 1;
        th[0] = theta; blo[0] = lo; bhi[0] = hi;
        {
            double t_ = theta, l_ = lo, h_ = hi;
            while (nb < B && used + nb - 1 < nuni) {
                if (t_ < 0.0) l_ = t_; else h_ = t_;
                t_ = l_ + (h_ - l_) * uniforms[used + nb - 1];
                th[nb] = t_; blo[nb] = l_; bhi[nb] = h_;
                ++nb;
            }
        }
        int rc = dgpamd_ess_propose(ctx, n, M, F, NU, th, nb, FP);
        if (rc) return rc;
        rc = dgpamd_loglik(ctx, node->kind, n, FP, M, n * (int64_t)M, (const int32_t *)node->colmap, node->Dl, node->Xglob,
                           node->Dg, node->length, node->nlen, node->nugget, node->W, scale, node->y, A, Np * Np, nb, ll_dev,
                           info_dev, work);
        if (rc) return rc;
        rc = dgpamd_fetch2(ctx, ll_dev, sizeof(double) * nb, info_dev, sizeof(int32_t) * nb, llh);
        if (rc) return rc;
        const int32_t *infoh = reinterpret_cast<const int32_t *>(llh + nb);
        ++batches;
        for (int b = 0; b < nb; ++b) {
            if (infoh[b] != 0) {
                proposals += b + 1;
                return finish(2, 0.0, infoh[b]);
            }
            if (llh[b] > log_y) {
                used += b;
                proposals += b + 1;
                theta = th[b]; lo = blo[b]; hi = bhi[b];
                HIP_TRY(ctx, hipMemcpyAsync(F, FP + (int64_t)b * n * M, sizeof(double) * n * M, hipMemcpyDeviceToDevice,
                                            ctx->stream));
                return finish(0, llh[b], 0);
            }
        }
        used += nb - 1;
        proposals += nb;
        theta = th[nb - 1]; lo = blo[nb - 1]; hi = bhi[nb - 1];
        pending = true;
    }
}

// ----------------------------------------------------------------------------
// a7  elliptical-slice updates queued WITHOUT host synchronisation (imputation.py:44-119)
//
// The accept / shrink loop of an update depends on the data only through "which proposal is the first above the
// threshold"; everything it needs -- the bracket, the cursor into the sampler's uniform stream, the threshold, the
// current log-likelihood -- lives in a small device state that a one-thread DECIDE kernel advances after every
// speculative batch.  The launches of the later batches of an update are predicated on its `done` word (ctx->pred:
// K assembly and the one-launch factorisation return at once), so a whole sequence of updates is queued back to back and
// the host fetches the state ONCE at the end.  An update that is not accepted within `max_batches` batches, or runs out of
// uniforms, sets a status; every later update of the queue then leaves the latents alone and the host resumes from the
// state (it holds the bracket, the cursor and the pending shrink exactly as the sequential loop would have them).
// ----------------------------------------------------------------------------
enum { ES_THETA = 0, ES_LO, ES_HI, ES_PENDING, ES_CURSOR, ES_STATUS, ES_INFO, ES_LL, ES_LOGY, ES_PROPOSALS, ES_BATCHES,
       ES_UPDATES, ES_NWORDS = DGPAMD_ESS_STATE };
struct EssScratch {   // device scratch of one queue (dgpamd_ess_queue_scratch bytes)
    double th[DGPAMD_MAXB], lo[DGPAMD_MAXB], hi[DGPAMD_MAXB], cs[DGPAMD_MAXB], sn[DGPAMD_MAXB], ll[DGPAMD_MAXB], logdet[DGPAMD_MAXB];
    int32_t info[DGPAMD_MAXB], infomax[DGPAMD_MAXB];
    int32_t nb, done, acc, halt;   // halt: the queue has stopped (status != 0): later updates launch nothing
    double likpart[DGPAMD_MAXB * 32];   // a likelihood node's partial sums (LIK_CHUNKS per candidate)
};
#define ESS_TWO_PI 6.283185307179586

__device__ void ess_prepare(double *st, const double *u, int nuni, int B, EssScratch *sc);
// Opens update number `upd` of the queue: closes the previous one first (not accepted within its queued batches: status 3,
// what ess_end_kernel does after the last update), draws the threshold and the first angle, and prepares the first batch.
__global__ void ess_begin_kernel(double *st, const double *u, const double *logu, int nuni, EssScratch *sc, int upd, int B, int resume) {
    if (threadIdx.x) return;
    if (resume && upd == 0) {   // the OPEN update of an earlier queue goes on: its threshold, angle and bracket are in the state (the closing shrink of its last, fully
        sc->done = 0;           // rejected batch is taken by ess_prepare from the uniforms of THIS queue)
        sc->halt = 0;
        ess_prepare(st, u, nuni, B, sc);
        return;
    }
    if (upd > 0 && !sc->done && st[ES_STATUS] == 0.0) st[ES_STATUS] = 3.0;
    sc->acc = -1;
    sc->nb = 0;
    if (st[ES_STATUS] != 0.0) { sc->done = 1; sc->halt = 1; return; }   // an earlier update of the queue has stopped it
    int cur = (int)st[ES_CURSOR];
    if (cur + 2 > nuni) { st[ES_STATUS] = 4.0; sc->done = 1; sc->halt = 1; return; }   // (before the update has begun)
    st[ES_LOGY] = st[ES_LL] + logu[cur];                   // log_y = ll + log u   (imputation.py:79)
    const double theta = ESS_TWO_PI * u[cur + 1];          // theta ~ U(0, 2 pi), bracket [theta - 2 pi, theta]  (:81-82)
    st[ES_THETA] = theta; st[ES_LO] = theta - ESS_TWO_PI; st[ES_HI] = theta; st[ES_PENDING] = 0.0;
    st[ES_CURSOR] = cur + 2;
    sc->done = 0;
    sc->halt = 0;
    ess_prepare(st, u, nuni, B, sc);
}
// the angles of the next speculative batch: theta, then what consecutive rejections would produce (imputation.py:115-119)
__device__ void ess_prepare(double *st, const double *u, int nuni, int B, EssScratch *sc) {
    sc->acc = -1;
    sc->nb = 0;
    if (sc->done) return;
    int cur = (int)st[ES_CURSOR];
    double theta = st[ES_THETA], lo = st[ES_LO], hi = st[ES_HI];
    if (st[ES_PENDING] != 0.0) {   // closing shrink of the previous, fully rejected batch
        if (cur >= nuni) { st[ES_STATUS] = 1.0; sc->done = 1; return; }
        if (theta < 0.0) lo = theta; else hi = theta;
        theta = __dadd_rn(lo, __dmul_rn(hi - lo, u[cur++]));   // numpy's uniform(lo, hi): a product and a sum, two roundings (no fma: the host loop's and the reference's bits)
        st[ES_THETA] = theta; st[ES_LO] = lo; st[ES_HI] = hi; st[ES_PENDING] = 0.0; st[ES_CURSOR] = cur;
    }
    int nb = 1;
    sc->th[0] = theta; sc->lo[0] = lo; sc->hi[0] = hi;
    while (nb < B && cur + nb - 1 < nuni) {
        if (theta < 0.0) lo = theta; else hi = theta;
        theta = __dadd_rn(lo, __dmul_rn(hi - lo, u[cur + nb - 1]));
        sc->th[nb] = theta; sc->lo[nb] = lo; sc->hi[nb] = hi;
        ++nb;
    }
    for (int b = nb; b < B; ++b) sc->th[b] = sc->th[nb - 1];   // (unused slots: a valid angle keeps their matrices harmless)
    for (int b = 0; b < B; ++b) { sc->cs[b] = cos(sc->th[b]); sc->sn[b] = sin(sc->th[b]); }
    sc->nb = nb;
}
__global__ void ess_prepare_kernel(double *st, const double *u, int nuni, int B, EssScratch *sc) {
    if (threadIdx.x) return;
    ess_prepare(st, u, nuni, B, sc);
}
__global__ __launch_bounds__(256) void ess_propose_dev_kernel(const double *F, const double *NU, double *FP, int64_t count,
                                                              const EssScratch *sc) {
    if (sc->done) return;
    const int b = blockIdx.y;
    const double c = sc->cs[b], s = sc->sn[b];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        FP[(int64_t)b * count + i] = F[i] * c + NU[i] * s;
}
// log-likelihood of one upper node for every proposal of the batch, summed over the nodes (imputation.py:91-106)
// log-likelihood of one upper node for every proposal of the batch, summed over the nodes (imputation.py:91-106).  The
// factorisation's results come straight from its workspace when it ran as the one-launch kernel (ld_ws / info_ws / status:
// PotrfPost; null: sc->logdet / sc->info hold them); `st` non-null (the last node of a batch): the decision follows in the
// same launch.
__device__ void ess_decide(double *st, EssScratch *sc);
__global__ void ess_node_ll_kernel(const double *A, int64_t ld, int64_t stride_a, int64_t n, double scale, int B, int first,
                                   EssScratch *sc, const double *ld_ws, const int32_t *info_ws, const int32_t *status, double *st) {
    const int b = threadIdx.x;
    if (sc->done) return;
    if (b < B) {
        const double logdet = ld_ws ? __hip_atomic_load(ld_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sc->logdet[b];
        const int32_t info = ld_ws ? ((status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                          ? -1 : __hip_atomic_load(info_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                   : sc->info[b];
        const double quad = -A[(int64_t)b * stride_a + n * ld + n];
        const double ll = -0.5 * ((double)n * log(scale) + logdet + quad / scale);
        sc->ll[b] = (first ? 0.0 : sc->ll[b]) + ll;
        sc->infomax[b] = first ? info : (sc->infomax[b] ? sc->infomax[b] : info);
    }
    if (st) {
        __threadfence_block();
        __syncthreads();
        if (b == 0) ess_decide(st, sc);
    }
}
__device__ void ess_decide(double *st, EssScratch *sc) {
    if (sc->done) return;
    const int nb = sc->nb;
    int cur = (int)st[ES_CURSOR];
    st[ES_BATCHES] += 1.0;
    for (int b = 0; b < nb; ++b) {
        if (sc->infomax[b] != 0) {   // (only a proposal the sequential loop would have reached can stop the update)
            st[ES_STATUS] = 2.0; st[ES_INFO] = sc->infomax[b]; st[ES_PROPOSALS] += b + 1; sc->done = 1;
            return;
        }
        if (sc->ll[b] > st[ES_LOGY]) {
            st[ES_CURSOR] = cur + b; st[ES_PROPOSALS] += b + 1;
            st[ES_THETA] = sc->th[b]; st[ES_LO] = sc->lo[b]; st[ES_HI] = sc->hi[b]; st[ES_PENDING] = 0.0;
            st[ES_LL] = sc->ll[b]; st[ES_UPDATES] += 1.0;
            sc->acc = b; sc->done = 1;
            return;
        }
    }
    st[ES_CURSOR] = cur + nb - 1; st[ES_PROPOSALS] += nb;
    st[ES_THETA] = sc->th[nb - 1]; st[ES_LO] = sc->lo[nb - 1]; st[ES_HI] = sc->hi[nb - 1]; st[ES_PENDING] = 1.0;
}
__global__ __launch_bounds__(256) void ess_accept_kernel(double *F, const double *FP, int64_t count, const EssScratch *sc) {
    const int acc = sc->acc;
    if (acc < 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        F[i] = FP[(int64_t)acc * count + i];
}
__global__ void ess_end_kernel(double *st, EssScratch *sc) {
    if (threadIdx.x) return;
    if (!sc->done && st[ES_STATUS] == 0.0) st[ES_STATUS] = 3.0;   // not accepted within the queued batches
    sc->acc = -1;
}
__global__ void ess_set_ll_kernel(double *st, const EssScratch *sc) {
    if (threadIdx.x) return;
    if (st[ES_STATUS] != 0.0) return;   // (a continued queue that has already stopped: its state is what the host resumes from)
    if (sc->infomax[0] != 0) { st[ES_STATUS] = 2.0; st[ES_INFO] = sc->infomax[0]; }
    st[ES_LL] = sc->ll[0];
}
__global__ void ess_note_info_kernel(double *st, const int32_t *info, int count) {
    if (threadIdx.x || st[ES_STATUS] != 0.0) return;
    for (int i = 0; i < count; ++i)
        if (info[i] != 0) { st[ES_STATUS] = 2.0; st[ES_INFO] = info[i]; return; }
}

// ---- Vecchia nodes upstairs (kernel.log_likelihood_func_vecch, kernel_class.py:494-509) ----
// The ordered inputs of every candidate block: out[b][i][d] = [X_b[ord[i]][colmap[d]] | Xglob[ord[i]][d - Dl]]
struct VGatherArgs {
    const double *X;
    int64_t stride_x;   // doubles between the candidate blocks (0: one block)
    int M, Dl, Dg;
    int colmap[DGPAMD_MAXD];
    const double *Xglob;
    const int64_t *ord;
    int64_t n;
    double *out;
    const int32_t *pred;
};
__global__ __launch_bounds__(256) void ess_vgather_kernel(VGatherArgs a) {
    if (a.pred && *a.pred) return;
    const int D = a.Dl + a.Dg, b = blockIdx.y;
    const int64_t total = a.n * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / D;
        const int d = (int)(e - i * D);
        const int64_t src = a.ord[i];
        a.out[(int64_t)b * total + e] = d < a.Dl ? a.X[(int64_t)b * a.stride_x + src * a.M + a.colmap[d]] : a.Xglob[src * a.Dg + (d - a.Dl)];
    }
}
// ... and the node's log-likelihood of every candidate from the row sums o[b] = (quad, logdet): -0.5 (logdet + quad / scale)
__global__ void ess_vnode_ll_kernel(const double *o, double scale, int B, int first, EssScratch *sc, double *st) {
    const int b = threadIdx.x;
    if (sc->done) return;
    if (b < B) {
        const double ll = -0.5 * (o[2 * b + 1] + o[2 * b] / scale);
        sc->ll[b] = (first ? 0.0 : sc->ll[b]) + ll;
        if (first) sc->infomax[b] = 0;
    }
    if (st) {
        __threadfence_block();
        __syncthreads();
        if (b == 0) ess_decide(st, sc);
    }
}

// ---- likelihood node upstairs (the reference's plugin protocol llik(): likelihood_class.py:30-90 Poisson, :245-292 NegBin,
// :470-621 ZIP, :624-812 ZINB, Categorical) ----
// log p(y_i | f_i) summed over the observations of every candidate block: obs i reads latent row rep[i] (replicates) or i,
// columns cols[0..ncol).  LIK_CHUNKS fixed slices of the observations per candidate, each summed by one workgroup in a fixed
// order (thread-strided, then a tree), the slices added in order by lik_sum / ess_liknode_ll_kernel: the same bits from the
// queue and from dgpamd_lik_loglik.
#define LIK_CHUNKS 32
struct LikArgs {
    const double *X;
    int64_t stride_x;
    int M, kind, ncol, classes;
    int cols[DGPAMD_MAXD];
    const int64_t *rep;
    const double *y;
    int64_t nobs;
    double par;
    double *part;   // [B][LIK_CHUNKS]
    const int32_t *pred;
};
__device__ __forceinline__ double lik_logaddexp(double a, double b) {   // numpy.logaddexp
    if (a == b) return a + 0.6931471805599453;   // (also -inf, -inf and +inf, +inf)
    const double d = a - b;
    if (d > 0.0) return a + log1p(exp(-d));
    if (d <= 0.0) return b + log1p(exp(d));
    return a + b;   // NaN
}
__device__ __forceinline__ double lik_log_ndtr(double x) {   // scipy.special.log_ndtr
    if (x > 0.0) return log1p(-0.5 * erfc(x * 0.7071067811865476));
    if (x > -20.0) return log(0.5 * erfc(-x * 0.7071067811865476));
    const double r = 1.0 / (x * x);   // asymptotic series of the Mills ratio
    const double ser = 1.0 + r * (-1.0 + r * (3.0 + r * (-15.0 + r * (105.0 + r * (-945.0)))));
    return -0.5 * x * x - log(-x) - 0.9189385332046727 + log(ser);
}
// NegBin._logpmf (likelihood_class.py:245-292): gammaln(y + size) - gammaln(size) - gammaln(y + 1) + y a - (y + size) logaddexp(0, a),
// size = exp(-f2), a = f1 + f2.  For counts up to 64 the first difference is evaluated as sum_{j<y} log(size + j) -- the same
// number, without the cancellation of two gammaln values of order size log size that makes the expression noise (+-10^3 at
// size = 1e16) when a slice-sampling proposal drives the dispersion latent far negative; a chain fed that noise gets trapped on
// spuriously high values.  Where the reference's expression is well conditioned the two agree to rounding.
__device__ __forceinline__ double lik_negbin(double y, double f1, double f2) {
    const double size = exp(-f2), a = f1 + f2;
    double lg;
    if (y >= 0.0 && y <= 64.0 && y == floor(y)) {
        lg = 0.0;
        for (double j = 0.0; j < y; j += 1.0) lg += log(size + j);
    } else {
        lg = lgamma(y + size) - lgamma(size);
    }
    return lg - lgamma(y + 1.0) + y * a - (y + size) * lik_logaddexp(0.0, a);
}
__device__ __forceinline__ double lik_expit(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ double lik_point(const LikArgs &a, const double *f, double y) {
    const int *c = a.cols;
    switch (a.kind) {
    case DGPAMD_LIK_POISSON: {
        const double v = f[c[0]];
        return y * v - exp(v) - lgamma(y + 1.0);
    }
    case DGPAMD_LIK_NEGBIN:
        return lik_negbin(y, f[c[0]], f[c[1]]);
    case DGPAMD_LIK_ZIP: {
        const double fl = f[c[0]], lam = exp(fl), pi = lik_expit(f[c[1]]);
        if (y == 0.0) return lik_logaddexp(log(pi), log1p(-pi) - lam);
        return log1p(-pi) - lam + y * fl - lgamma(y + 1.0);
    }
    case DGPAMD_LIK_ZINB: {
        const double nb = lik_negbin(y, f[c[0]], f[c[1]]), pi = lik_expit(f[c[2]]);
        if (y == 0.0) return lik_logaddexp(log(pi), log1p(-pi) + nb);
        return log1p(-pi) + nb;
    }
    case DGPAMD_LIK_BIN_LOGIT: {
        const double v = f[c[0]];
        return y * v - lik_logaddexp(0.0, v);
    }
    case DGPAMD_LIK_BIN_PROBIT: {
        const double v = f[c[0]];
        return y * lik_log_ndtr(v) + (1.0 - y) * lik_log_ndtr(-v);
    }
    case DGPAMD_LIK_ROBUSTMAX: {
        int best = 0;
        double top = f[c[0]];
        bool nan = top != top;   // (numpy.argmax returns the first NaN)
        for (int k = 1; k < a.ncol && !nan; ++k) {
            const double v = f[c[k]];
            if (v != v) { best = k; nan = true; }
            else if (v > top) { top = v; best = k; }
        }
        return best == (int)y ? log(1.0 - a.par) : log(a.par / (a.classes - 1));
    }
    default: {   // DGPAMD_LIK_SOFTMAX
        double top = f[c[0]];
        for (int k = 1; k < a.ncol; ++k) {
            const double v = f[c[k]];
            top = (v > top || v != v) ? v : top;
        }
        double s = 0.0;
        for (int k = 0; k < a.ncol; ++k) s += exp(f[c[k]] - top);
        return f[c[(int)y]] - (log(s) + top);
    }
    }
}
__global__ __launch_bounds__(256) void lik_partial_kernel(LikArgs a) {
    if (a.pred && *a.pred) return;
    __shared__ double red[256];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int64_t per = (a.nobs + LIK_CHUNKS - 1) / LIK_CHUNKS, lo = ch * per, hi = lo + per < a.nobs ? lo + per : a.nobs;
    const double *Xb = a.X + (int64_t)b * a.stride_x;
    double acc = 0.0;
    for (int64_t i = lo + tid; i < hi; i += 256) {
        const int64_t row = a.rep ? a.rep[i] : i;
        acc += lik_point(a, Xb + row * a.M, a.y[i]);
    }
    red[tid] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) a.part[b * LIK_CHUNKS + ch] = red[0];
}
__global__ void lik_sum_kernel(const double *part, int B, double *out) {
    const int b = threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int c = 0; c < LIK_CHUNKS; ++c) s += part[b * LIK_CHUNKS + c];
    out[b] = s;
}
__global__ void ess_liknode_ll_kernel(const double *part, int B, int first, EssScratch *sc, double *st) {
    const int b = threadIdx.x;
    if (sc->done) return;
    if (b < B) {
        double s = 0.0;
        for (int c = 0; c < LIK_CHUNKS; ++c) s += part[b * LIK_CHUNKS + c];
        sc->ll[b] = (first ? 0.0 : sc->ll[b]) + s;
        if (first) sc->infomax[b] = 0;
    }
    if (st) {
        __threadfence_block();
        __syncthreads();
        if (b == 0) ess_decide(st, sc);
    }
}
static int lik_launch(dgpamd_ctx *ctx, const dgpamd_node &nd, int64_t n, int M, const double *X, int64_t stride_x, int B, double *part) {
    if (nd.lik_kind < DGPAMD_LIK_POISSON || nd.lik_kind > DGPAMD_LIK_SOFTMAX) BAD_ARG(ctx, "unknown likelihood");
    if (!nd.y || nd.lik_nobs <= 0 || !nd.colmap || nd.Dl <= 0 || nd.Dl > DGPAMD_MAXD) BAD_ARG(ctx, "incomplete likelihood node");
    static const int need[] = {0, 1, 2, 2, 3, 1, 1, 0, 0};
    if (need[nd.lik_kind] && nd.Dl != need[nd.lik_kind]) BAD_ARG(ctx, "wrong number of latent columns for this likelihood");
    if (nd.lik_kind >= DGPAMD_LIK_ROBUSTMAX && (nd.lik_classes != nd.Dl || nd.Dl < 2)) BAD_ARG(ctx, "one latent column per class");
    if (!nd.lik_rep && nd.lik_nobs != n) BAD_ARG(ctx, "observations and latent rows differ without a replicate map");
    LikArgs a;
    a.X = X; a.stride_x = stride_x; a.M = M; a.kind = nd.lik_kind; a.ncol = nd.Dl; a.classes = nd.lik_classes;
    for (int d = 0; d < nd.Dl; ++d) {
        a.cols[d] = ((const int32_t *)nd.colmap)[d];
        if (a.cols[d] < 0 || a.cols[d] >= M) BAD_ARG(ctx, "latent column out of range");
    }
    a.rep = nd.lik_rep; a.y = nd.y; a.nobs = nd.lik_nobs; a.par = nd.lik_par; a.part = part; a.pred = ctx->pred;
    hipLaunchKernelGGL(lik_partial_kernel, dim3(LIK_CHUNKS, B), dim3(256), 0, ctx->stream, a);
    return DGPAMD_OK;
}
extern "C" int dgpamd_lik_loglik(dgpamd_ctx *ctx, const dgpamd_node *node, int64_t n, int M, const double *X, int64_t stride_x, int batch,
                                 double *work, double *out) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!node || !X || !work || !out || n <= 0 || M <= 0 || batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "bad arguments");
    int rc = lik_launch(ctx, *node, n, M, X, stride_x, batch, work);
    if (rc) return rc;
    hipLaunchKernelGGL(lik_sum_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)work, batch, out);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}
extern "C" size_t dgpamd_lik_workspace(int batch) { return (size_t)batch * LIK_CHUNKS * sizeof(double); }

extern "C" size_t dgpamd_ess_queue_scratch(void) { return (sizeof(EssScratch) + 15) / 16 * 16; }
// gathered inputs (batch x n x D) + per-row partials (batch x n x 2) + the row sums (batch x 2)
extern "C" size_t dgpamd_ess_queue_vwork(int64_t n, int D, int batch) {
    return ((size_t)batch * n * (D + 2) + 2 * (size_t)batch + 2) * sizeof(double);
}

// all upper nodes' log-likelihoods of the B candidate blocks X (B x n x M; stride 0: one block) into sc->ll / infomax
static int ess_queue_logliks(dgpamd_ctx *ctx, int64_t n, int M, const double *X, int64_t stride_x, int B, const dgpamd_node *nodes,
                             const double *scales_h, int nnodes, double *A, void *work, EssScratch *sc, double *st_decide, void *vwork,
                             int vbatch) {
    const int64_t Np = padded_dim(n);
    double *ws = (double *)work;
    for (int k = 0; k < nnodes; ++k) {
        const dgpamd_node &nd = nodes[k];
        if (nd.lik_kind) {   // a likelihood node: the summed log-density of the observations
            int rc = lik_launch(ctx, nd, n, M, X, stride_x, B, sc->likpart);
            if (rc) return rc;
            hipLaunchKernelGGL(ess_liknode_ll_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)sc->likpart, B,
                               k == 0 ? 1 : 0, sc, k == nnodes - 1 ? st_decide : nullptr);
            continue;
        }
        if (!(scales_h[k] > 0.0)) BAD_ARG(ctx, "scale must be positive");
        if (nd.vecch_nn) {
            const int D = nd.Dl + nd.Dg;
            if (!vwork || !nd.vecch_ord || !nd.vecch_nd || !nd.vecch_y || nd.vecch_m < 0) BAD_ARG(ctx, "incomplete Vecchia node");
            if (D <= 0 || D > DGPAMD_MAXD) BAD_ARG(ctx, "need 1 <= Dl+Dg <= DGPAMD_MAXD");
            double *Xall = (double *)vwork, *partial = Xall + (size_t)vbatch * n * D, *osum = partial + (size_t)vbatch * n * 2;
            VGatherArgs ga;
            ga.X = X; ga.stride_x = stride_x; ga.M = M; ga.Dl = nd.Dl; ga.Dg = nd.Dg;
            for (int d = 0; d < nd.Dl; ++d) ga.colmap[d] = nd.colmap ? ((const int32_t *)nd.colmap)[d] : d;
            ga.Xglob = nd.Xglob; ga.ord = nd.vecch_ord; ga.n = n; ga.out = Xall; ga.pred = ctx->pred;
            int64_t gb = (n * D + 255) / 256;
            if (gb > 2048) gb = 2048;
            hipLaunchKernelGGL(ess_vgather_kernel, dim3((unsigned)gb, B), dim3(256), 0, ctx->stream, ga);
            // (the rows of this rank -- all of them unless the likelihood's rows are split over processes; a row reads X and y through
            //  its neighbour list, so a block of rows needs nothing else)
            const bool part = nd.vecch_rows > 0;
            if (part && (nd.vecch_row0 < 0 || nd.vecch_row0 + nd.vecch_rows > n)) BAD_ARG(ctx, "row block outside the neighbour array");
            if (part && !ctx->reduce_hook) BAD_ARG(ctx, "a row block needs a reduce hook (dgpamd_set_reduce_hook)");
            const int64_t rows = part ? nd.vecch_rows : n;
            int rc = vecchia_llik_batch_into(ctx, nd.kind, rows, D, nd.vecch_m, Xall, n * (int64_t)D, B, nd.vecch_y,
                                             nd.vecch_nn + (part ? nd.vecch_row0 * (nd.vecch_m + 1) : 0), nd.length, nd.nlen, nd.nugget,
                                             nd.vecch_nd, partial, osum);
            if (rc) return rc;
            if (part) {   // every rank's sums together, on the stream, before the decision below reads them
                const hipStream_t was = ctx->stream;
                if (ctx->reduce_hook(ctx->reduce_user, osum, 2 * B) != 0 || ctx->stream != was) BAD_ARG(ctx, "the reduce hook failed");
            }
            hipLaunchKernelGGL(ess_vnode_ll_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)osum, scales_h[k], B,
                               k == 0 ? 1 : 0, sc, k == nnodes - 1 ? st_decide : nullptr);
            continue;
        }
        if (!A || !work) BAD_ARG(ctx, "dense nodes upstairs need A and work");
        KmatArgs a;
        int rc = build_kmat_args(ctx, a, nd.kind, n, X, M, stride_x, (const int32_t *)nd.colmap, nd.Dl, nd.Xglob, nd.Dg, nd.length,
                                 nd.nlen, nd.nugget, nd.W, A, Np, Np * Np, 0, nd.y, n, 0, 1, B);
        if (rc) return rc;
        potrf_sync_area(ctx, n, B, false, ws, &a.zero_ptr, &a.zero_words);
        rc = launch_kmatrix(ctx, a, B);
        if (rc) return rc;
        PotrfPost post;   // (the factorisation's results are read where they are: no copy-out launch)
        rc = run_potrf(ctx, n, A, Np * Np, B, sc->logdet, sc->info, ws, nullptr, nullptr, &post, a.zero_ptr != nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(ess_node_ll_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)A, Np, Np * Np, n,
                           scales_h[k], B, k == 0 ? 1 : 0, sc, post.pending ? post.ld_ws : nullptr,
                           post.pending ? post.info_ws : nullptr, post.pending ? post.status : nullptr,
                           k == nnodes - 1 ? st_decide : nullptr);
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_ess_queue(dgpamd_ctx *ctx, int64_t n, int M, double *F, const double *NU, int nupd, const dgpamd_node *nodes,
                                const double *scales_h, int nnodes, double *state, const double *uniforms, const double *log_uniforms,
                                int nuni, int batch_first, int batch_next, int max_batches, int compute_ll0, double *FP, double *A,
                                void *work, void *scratch, void *vwork) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || M <= 0 || nupd <= 0 || nnodes <= 0 || !F || !NU || !nodes || !scales_h || !state || !FP || !scratch)
        BAD_ARG(ctx, "null pointer or empty block");
    if (nuni < 0 || (nuni > 0 && (!uniforms || !log_uniforms))) BAD_ARG(ctx, "bad uniform stream");
    if (batch_first <= 0 || batch_first > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch_first <= DGPAMD_MAXB");
    if (batch_next <= 0 || batch_next > batch_first) batch_next = batch_first;
    if (max_batches < 1) max_batches = 1;
    EssScratch *sc = reinterpret_cast<EssScratch *>(scratch);
    const int64_t count = n * (int64_t)M;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    int rc;
    HIP_TRY(ctx, hipMemsetAsync(sc, 0, sizeof(EssScratch), ctx->stream));
    const int resume = compute_ll0 == 2 ? 1 : 0;   // (2: the first update of this queue is the open update the previous queue of the I-step left)
    if (compute_ll0 == 1) {   // log-likelihood of the current state (imputation.py:70-78): the first threshold's base
        rc = ess_queue_logliks(ctx, n, M, F, 0, 1, nodes, scales_h, nnodes, A, work, sc, nullptr, vwork, batch_first);
        if (rc) return rc;
        hipLaunchKernelGGL(ess_set_ll_kernel, dim3(1), dim3(64), 0, ctx->stream, state, (const EssScratch *)sc);
    }
    for (int u = 0; u < nupd; ++u) {
        const double *NUu = NU + (int64_t)u * count;
        hipLaunchKernelGGL(ess_begin_kernel, dim3(1), dim3(64), 0, ctx->stream, state, uniforms, log_uniforms, nuni, sc, u, batch_first, resume);
        for (int j = 0; j < max_batches; ++j) {
            const int B = j == 0 ? batch_first : batch_next;
            if (j > 0) hipLaunchKernelGGL(ess_prepare_kernel, dim3(1), dim3(64), 0, ctx->stream, state, uniforms, nuni, B, sc);
            hipLaunchKernelGGL(ess_propose_dev_kernel, dim3((unsigned)blocks, B), dim3(256), 0, ctx->stream, (const double *)F, NUu, FP,
                               count, (const EssScratch *)sc);
            // every launch is predicated on the update being open (the first batch: on the queue not having stopped)
            ctx->pred = &sc->done;
            rc = ess_queue_logliks(ctx, n, M, FP, count, B, nodes, scales_h, nnodes, A, work, sc, state, vwork, batch_first);   // (decides as well)
            ctx->pred = nullptr;
            if (rc) return rc;
            hipLaunchKernelGGL(ess_accept_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, F, (const double *)FP, count,
                               (const EssScratch *)sc);
        }
        if (u == nupd - 1) hipLaunchKernelGGL(ess_end_kernel, dim3(1), dim3(64), 0, ctx->stream, state, sc);
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_ess_queue_note_info(dgpamd_ctx *ctx, double *state, const int32_t *info, int count) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!state || !info || count <= 0) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(ess_note_info_kernel, dim3(1), dim3(64), 0, ctx->stream, state, info, count);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}
