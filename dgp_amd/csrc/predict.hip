// Imputed-GP prediction.
//   gp      (functions.py:379-394)  m = ry.r , v = |scale(1+eta - r^T Rinv r)|
//           as  cross_corr -> symmetric MFMA quadratic form (lower tiles of Rinv, x2) -> finalize
//   link_gp (functions.py:396-430)  m = I.ry , v = |sum_ij J_ij (ry_i ry_j - scale Rinv_ij) - m^2 + scale(1+eta)|
//           one workgroup per (lower 64x64 tile of C = ry ry^T - scale Rinv) x (chunk of test points);
//           Psexp / R2sexp (functions.py:259-272, kernel_class.py:752-764) are never materialised.
#include "common.hpp"
#include "tile.hpp"
#include "linkfun.hpp"

#include <math.h>

#define MC_MAX 2048   // test points per workspace chunk
#define TCH 32        // test points per linked-GP workgroup

__device__ __forceinline__ double wave_sum_p(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// r[i][t] = k(W_i, x_t)   (K_vec_nb, vecchia.py:244-265)
// ---------------------------------------------------------------------------
struct CrossArgs {
    int kind, D;
    double inv_len[DGPAMD_MAXD];
    int64_t n, M, t0, Mc;
    const double *W, *x;
    double *R;   // [npad][Mc]
};

template <int KIND>
__global__ __launch_bounds__(256) void cross_corr_kernel(CrossArgs a) {
    extern __shared__ double lds[];
    const int D = a.D;
    double *WT = lds, *XT = lds + D * 64;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    for (int idx = tid; idx < 64 * D; idx += 256) {
        int row = idx / D, d = idx - row * D;
        int64_t gi = i0 + row, gt = a.t0 + c0 + row;
        WT[d * 64 + row] = (gi < a.n ? a.W[gi * D + d] : 0.0) * a.inv_len[d];
        XT[d * 64 + row] = (gt < a.M ? a.x[gt * D + d] : 0.0) * a.inv_len[d];
    }
    __syncthreads();
    double s[4][4], pr[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[p][q] = 0.0;
            pr[p][q] = 1.0;
        }
    for (int d = 0; d < D; ++d) {
        double xi[4], xj[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            xi[p] = WT[d * 64 + ty + 16 * p];
            xj[p] = XT[d * 64 + tx + 16 * p];
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
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t gi = i0 + ty + 16 * p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = (KIND == DGPAMD_SEXP) ? exp(-s[p][q]) : pr[p][q] * exp(-SQRT5 * s[p][q]);
            if (gi >= a.n) v = 0.0;
            a.R[gi * a.Mc + c0 + tx + 16 * q] = v;
        }
    }
}

// partial[bi][t] = sum_{i in block bi} r_it * ( Rinv[bi][bi] r_bi + 2 sum_{kb<bi} Rinv[bi][kb] r_kb )_it
struct GpQuadArgs {
    const double *Rinv;
    int64_t ldr, n, Mc;
    const double *R;
    double *partial;
};

__global__ __launch_bounds__(256) void gp_quad_kernel(GpQuadArgs a) {
    __shared__ double As[64 * LDM];
    __shared__ double Bs[KC * LDK];
    __shared__ double red[4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bi = blockIdx.x, tj = blockIdx.y;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int64_t rlim = a.n - (int64_t)bi * 64;
    for (int kb = 0; kb <= bi; ++kb) {
        const double *Ag = a.Rinv + ((int64_t)bi * 64) * a.ldr + (int64_t)kb * 64;
        const double *Bg = a.R + ((int64_t)kb * 64) * a.Mc + (int64_t)tj * 64;
        const int64_t clim = a.n - (int64_t)kb * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            load_mk_masked(Ag, a.ldr, As, tid, h, rlim > 64 ? 64 : (int)rlim, clim > 64 ? 64 : (int)clim);
            load_km(Bg, a.Mc, Bs, tid, h, 64);
            __syncthreads();
            mfma_tile<OP_MK, OP_KM>(As, Bs, acc, wave, lane, kb < bi ? 2.0 : 1.0);
        }
    }
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            v = fma(acc[t][r], a.R[((int64_t)bi * 64 + crow + 4 * r) * a.Mc + (int64_t)tj * 64 + 16 * t + ccol], v);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) red[wave][16 * t + lane] = v;
    }
    __syncthreads();
    if (tid < 64) a.partial[(int64_t)bi * a.Mc + (int64_t)tj * 64 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(256) void gp_finalize_kernel(const double *R, const double *partial, const double *ry,
                                                          int nry, int64_t n, int nb, int64_t Mc, int64_t t0, int64_t M,
                                                          double scale, double nugget, double *mean, double *var) {
    // 64 test points per workgroup (coalesced over t), the training points in 4 interleaved slices (one per wave)
    __shared__ double part[4][64];
    const int slice = threadIdx.x >> 6, tl = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 64 + tl;
    const bool live = t < Mc && t0 + t < M;
    for (int s = blockIdx.y; s < nry; s += gridDim.y) {
        const double *rys = ry + (int64_t)s * n;
        double m = 0.0;
        if (live)
            for (int64_t i = slice; i < n; i += 4) m = fma(rys[i], R[i * Mc + t], m);
        part[slice][tl] = m;
        __syncthreads();
        if (slice == 0 && live) mean[(int64_t)s * M + t0 + t] = (part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]);
        __syncthreads();
    }
    if (blockIdx.y == 0) {
        double q = 0.0;
        if (live)
            for (int b = slice; b < nb; b += 4) q += partial[(int64_t)b * Mc + t];
        part[slice][tl] = q;
        __syncthreads();
        if (slice == 0 && live) var[t0 + t] = fabs(scale * (1.0 + nugget - ((part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]))));
    }
}

extern "C" size_t dgpamd_gp_workspace(int64_t n, int64_t M) {
    int64_t nb = (n + 63) / 64;
    int64_t Mc = ((M + 63) / 64) * 64;
    if (Mc > MC_MAX) Mc = MC_MAX;
    return (size_t)((nb * 64 + nb) * Mc) * sizeof(double);
}

extern "C" int dgpamd_gp_predict(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int D, const double *x,
                                 const double *Wtr, const double *length_h, int nlen, const double *Rinv, int64_t ldr,
                                 const double *ry, int nry, double scale, double nugget, double *mean, double *var,
                                 void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || M <= 0 || nry <= 0 || !x || !Wtr || !length_h || !Rinv || !ry || !mean || !var || !work)
        BAD_ARG(ctx, "null pointer or empty problem");
    if (kind != DGPAMD_SEXP && kind != DGPAMD_MATERN25) BAD_ARG(ctx, "kind must be 0 or 1");
    if (D <= 0 || D > DGPAMD_MAXD || (nlen != 1 && nlen != D)) BAD_ARG(ctx, "bad D / nlen");
    if (ldr < n) BAD_ARG(ctx, "ldr < n");
    const int nb = (int)((n + 63) / 64);
    int64_t Mc = ((M + 63) / 64) * 64;
    if (Mc > MC_MAX) Mc = MC_MAX;
    double *R = (double *)work, *partial = R + (int64_t)nb * 64 * Mc;
    CrossArgs c;
    c.kind = kind; c.D = D; c.n = n; c.M = M; c.Mc = Mc; c.W = Wtr; c.x = x; c.R = R;
    for (int d = 0; d < D; ++d) c.inv_len[d] = 1.0 / length_h[nlen == 1 ? 0 : d];
    GpQuadArgs q;
    q.Rinv = Rinv; q.ldr = ldr; q.n = n; q.Mc = Mc; q.R = R; q.partial = partial;
    const size_t shm = (size_t)2 * D * 64 * sizeof(double);
    for (int64_t t0 = 0; t0 < M; t0 += Mc) {
        c.t0 = t0;
        int64_t mc = M - t0 < Mc ? M - t0 : Mc;
        unsigned tb = (unsigned)((mc + 63) / 64);
        if (kind == DGPAMD_SEXP)
            hipLaunchKernelGGL(cross_corr_kernel<DGPAMD_SEXP>, dim3(nb, tb), dim3(256), shm, ctx->stream, c);
        else
            hipLaunchKernelGGL(cross_corr_kernel<DGPAMD_MATERN25>, dim3(nb, tb), dim3(256), shm, ctx->stream, c);
        // algorithmic flops of r^T R^-1 r for mc test points as the reference forms it (functions.py:379-394: R^-1 r, then the dot
        // product): 2 n^2 mc; the kernel reads the lower tiles of R^-1 only and executes half of that
        PROF_BEGIN(ctx, PROF_GP_QUAD, 2.0 * (double)n * (double)n * (double)mc);
        hipLaunchKernelGGL(gp_quad_kernel, dim3(nb, tb), dim3(256), 0, ctx->stream, q);
        PROF_END(ctx, PROF_GP_QUAD);
        hipLaunchKernelGGL(gp_finalize_kernel, dim3((unsigned)((mc + 63) / 64), (unsigned)(nry < 64 ? nry : 64)),
                           dim3(256), 0, ctx->stream, (const double *)R, (const double *)partial, ry, nry, n, nb, Mc,
                           t0, M, scale, nugget, mean, var);
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// linked GP: closed forms of E[k(x,Z)] and E[k(x,Z) k(x',Z)], Z ~ N(m, v)
// ---------------------------------------------------------------------------
struct LinkArgs {
    int kind, Dw, Dz;
    int64_t n, M, t0, Mc;
    const double *m, *v, *z;
    const double *W, *Wg;
    double len[DGPAMD_MAXD];
    const double *Rinv;
    int64_t ldr;
    const double *ry;
    double scale, nugget;
    double *partial;   // [ntiles][Mc]
    double *recs;      // [Mc][Dw][npad][REC] separable Matern records (linkgp_Jsep)
    double *gfac;      // [Mc][npad] Matern factor of the deterministic global inputs (linkgp_Jsep)
    int64_t npad;
    double *mean, *var;
    const int32_t *drop;   // leave-one-out: training point left out of the conditioning set of test point t (else null)
    int no_order_classes;  // linkgp_Jsep: treat every sub-tile as mixed (DGPAMD_JSEP_NOCLASS: the comparison run of the tools)
    int tch;               // linkgp_Jsep: test points per workgroup
    long long *dbg;        // linkgp_Jsep_kernel<2, true> (DGPAMD_JSEP_LOG=1 with dgpamd_debug_tasklog's buffer set): per-workgroup and per-step stamps, or null
};
#define JSEP_LOG_STEPS 48
#define JSEP_LOG_WORDS (8 + JSEP_LOG_STEPS * 4 * 2)

// mean_t = sum_i I_i(t) ry_i : one workgroup per test point (the training points strided over its 256 threads)
template <int KIND>
__global__ __launch_bounds__(256) void linkgp_mean_kernel(LinkArgs a) {
    __shared__ double part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = a.t0 + (int64_t)blockIdx.x;
    if (t >= a.M || t >= a.t0 + a.Mc) return;
    const double *mt = a.m + t * a.Dw, *vt = a.v + t * a.Dw;
    const double *zt = a.Dz ? a.z + t * a.Dz : nullptr;
    // leave-one-out: (R_-d)^-1 y = Rinv y - u (Rinv y)_d / u_d with u = Rinv[:, d], zero at d itself
    const int64_t dr = a.drop ? a.drop[t] : -1;
    const double *ud = a.drop ? a.Rinv + dr * a.ldr : nullptr;
    const double cd = a.drop ? a.ry[dr] / ud[dr] : 0.0;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < a.n; i += 256) {
        if (i == dr) continue;
        double I;
        if (KIND == DGPAMD_SEXP) {
            double e = 0.0;
            for (int k = 0; k < a.Dw; ++k) {
                double l = a.len[k], d = a.W[i * a.Dw + k] - mt[k];
                e += d * d / (2.0 * vt[k] + l * l);
            }
            for (int g = 0; g < a.Dz; ++g) {
                double d = (a.Wg[i * a.Dz + g] - zt[g]) / a.len[a.Dw + g];
                e += d * d;
            }
            I = exp(-e);
        } else {
            I = 1.0;
            for (int k = 0; k < a.Dw; ++k) I *= matern_I_dim(a.W[i * a.Dw + k], mt[k], vt[k], a.len[k]);
            double pr = 1.0, s = 0.0;
            for (int g = 0; g < a.Dz; ++g) corr_accum_matern((a.Wg[i * a.Dz + g] - zt[g]) / a.len[a.Dw + g], pr, s);
            I *= pr * exp(-SQRT5 * s);
        }
        acc = fma(I, ud ? a.ry[i] - cd * ud[i] : a.ry[i], acc);
    }
    acc = wave_sum_p(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = (part[0] + part[1]) + (part[2] + part[3]);
        if (KIND == DGPAMD_SEXP) {
            double c = 1.0;
            for (int k = 0; k < a.Dw; ++k) c *= 1.0 + 2.0 * vt[k] / (a.len[k] * a.len[k]);
            acc *= 1.0 / sqrt(c);
        }
        a.mean[t] = acc;
    }
}

// partial[tile][t] = sum_{(i,j) in tile} wt (ry_i ry_j - scale Rinv_ij) Jhat_ij(t)
// (sexp: Jhat = J / J_coef1, the prefactor is applied in the finalize kernel)
// LOO: test point t conditions on every training point but d = drop[t].  With u = Rinv[:, d], rho = u_d the inverse of
// the reduced correlation matrix (embedded, zero row / column d) is Rinv - u u^T / rho, so the pair weight becomes
//   wt [ (ry_i - c u_i)(ry_j - c u_j) - scale (Rinv_ij - u_i u_j / rho) ],   c = ry_d / rho.
template <int KIND, bool LOO>
__global__ __launch_bounds__(256) void linkgp_J_kernel(LinkArgs a) {
    extern __shared__ double lds[];
    const int Dw = a.Dw, Dz = a.Dz, DT = Dw + Dz;
    double *WiT = lds;                    // [DT][64]
    double *WjT = WiT + DT * 64;          // [DT][64]
    double *tm = WjT + DT * 64;           // [TCH][Dw]
    double *tv = tm + TCH * Dw;           // [TCH][Dw]
    double *tz = tv + TCH * Dw;           // [TCH][Dz]
    double *red = tz + TCH * Dz;          // [TCH][4]
    double *Cs = red + TCH * 4;           // [64][65]  (matern only)
    double *ui = Cs + (KIND == DGPAMD_MATERN25 ? 64 * 65 : 0);   // LOO only: [TCH][64] u rows of the tile
    double *uj = ui + TCH * 64;           // [TCH][64] u columns of the tile
    double *ryi = uj + TCH * 64, *ryj = ryi + 64;
    double *ct = ryj + 64, *gt = ct + TCH;   // [TCH] c and wt scale / rho
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64, n = a.n;
    const int64_t tbase = a.t0 + (int64_t)blockIdx.y * TCH;
    int nt = TCH;
    if (tbase + nt > a.M) nt = (int)(a.M - tbase);
    if (tbase + nt > a.t0 + a.Mc) nt = (int)(a.t0 + a.Mc - tbase);

    for (int idx = tid; idx < 64 * DT; idx += 256) {
        int row = idx / DT, d = idx - row * DT;
        int64_t gi = i0 + row, gj = j0 + row;
        double vi = 0.0, vj = 0.0;
        if (d < Dw) {
            if (gi < n) vi = a.W[gi * Dw + d];
            if (gj < n) vj = a.W[gj * Dw + d];
        } else {
            if (gi < n) vi = a.Wg[gi * Dz + d - Dw];
            if (gj < n) vj = a.Wg[gj * Dz + d - Dw];
        }
        WiT[d * 64 + row] = vi;
        WjT[d * 64 + row] = vj;
    }
    for (int idx = tid; idx < nt * Dw; idx += 256) {
        tm[idx] = a.m[tbase * Dw + idx];
        tv[idx] = a.v[tbase * Dw + idx];
    }
    for (int idx = tid; idx < nt * Dz; idx += 256) tz[idx] = a.z[tbase * Dz + idx];

    const double wt = (bi == bj) ? 1.0 : 2.0;
    if (LOO) {
        if (tid < 64) ryi[tid] = i0 + tid < n ? a.ry[i0 + tid] : 0.0;
        else if (tid < 128) ryj[tid - 64] = j0 + tid - 64 < n ? a.ry[j0 + tid - 64] : 0.0;
        for (int idx = tid; idx < nt * 64; idx += 256) {
            const int t = idx >> 6, r = idx & 63;
            const double *u = a.Rinv + (int64_t)a.drop[tbase + t] * a.ldr;
            ui[idx] = i0 + r < n ? u[i0 + r] : 0.0;
            uj[idx] = j0 + r < n ? u[j0 + r] : 0.0;
        }
        if (tid < nt) {
            const int64_t d = a.drop[tbase + tid];
            const double rho = a.Rinv[d * a.ldr + d];
            ct[tid] = a.ry[d] / rho;
            gt[tid] = wt * a.scale / rho;
        }
    }
    // Cr: the pair weight (LOO: only its t-independent part wt scale Rinv_ij, subtracted per test point)
    double Cr[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t gi = i0 + ty + 16 * p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t gj = j0 + tx + 16 * q;
            double c = 0.0;
            if (gi < n && gj < n)
                c = LOO ? wt * a.scale * a.Rinv[gi * a.ldr + gj] : wt * (a.ry[gi] * a.ry[gj] - a.scale * a.Rinv[gi * a.ldr + gj]);
            Cr[p][q] = c;
            if (KIND == DGPAMD_MATERN25) Cs[(ty + 16 * p) * 65 + tx + 16 * q] = c;
        }
    }
    __syncthreads();

    if (KIND == DGPAMD_SEXP) {
        // t-independent part of the exponent: sum_k (w_ik - w_jk)^2 / (2 l_k^2)  (= -log R2sexp, kernel_class.py:761-763)
        double base[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) base[p][q] = 0.0;
        for (int k = 0; k < Dw; ++k) {
            const double c2 = 1.0 / (2.0 * a.len[k] * a.len[k]);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double d = WiT[k * 64 + ty + 16 * p] - WjT[k * 64 + tx + 16 * q];
                    base[p][q] = fma(d * d, c2, base[p][q]);
                }
        }
        for (int t = 0; t < nt; ++t) {
            double e[4][4], ei[4], ej[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ei[p] = 0.0;
                ej[p] = 0.0;
            }
            for (int g = 0; g < Dz; ++g) {
                const double il = 1.0 / a.len[Dw + g], zz = tz[t * Dz + g];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    double di = (WiT[(Dw + g) * 64 + ty + 16 * p] - zz) * il;
                    double dj = (WjT[(Dw + g) * 64 + tx + 16 * p] - zz) * il;
                    ei[p] = fma(di, di, ei[p]);
                    ej[p] = fma(dj, dj, ej[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) e[p][q] = base[p][q] + ei[p] + ej[q];
            for (int k = 0; k < Dw; ++k) {
                const double l = a.len[k];
                const double c1 = 1.0 / (8.0 * tv[t * Dw + k] + 2.0 * l * l), m2 = 2.0 * tm[t * Dw + k];
                double wi[4], wj[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    wi[p] = WiT[k * 64 + ty + 16 * p] - m2;
                    wj[p] = WjT[k * 64 + tx + 16 * p];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double sgm = wi[p] + wj[q];
                        e[p][q] = fma(sgm * sgm, c1, e[p][q]);
                    }
            }
            double acc = 0.0;
            if (LOO) {
                double ai[4], aj[4], gi_[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const double u1 = ui[t * 64 + ty + 16 * p], u2 = uj[t * 64 + tx + 16 * p];
                    ai[p] = wt * (ryi[ty + 16 * p] - ct[t] * u1);
                    aj[p] = ryj[tx + 16 * p] - ct[t] * u2;
                    gi_[p] = gt[t] * u1;
                    ej[p] = u2;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc = fma(fma(ai[p], aj[q], fma(gi_[p], ej[q], -Cr[p][q])), exp(-e[p][q]), acc);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = fma(Cr[p][q], exp(-e[p][q]), acc);
            }
            acc = wave_sum_p(acc);
            if (lane == 0) red[t * 4 + wave] = acc;
        }
    } else {
        for (int t = 0; t < nt; ++t) {
            double acc = 0.0;
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                const int r = ty + 16 * (e >> 2), c = tx + 16 * (e & 3);
                double cij = Cs[r * 65 + c];
                if (LOO ? (i0 + r >= n || j0 + c >= n) : cij == 0.0) continue;
                if (LOO) {
                    const double u1 = ui[t * 64 + r], u2 = uj[t * 64 + c];
                    cij = fma(wt * (ryi[r] - ct[t] * u1), ryj[c] - ct[t] * u2, fma(gt[t] * u1, u2, -cij));
                }
                const bool diag = (i0 + r == j0 + c);
                double prod = 1.0;
                for (int k = 0; k < Dw; ++k) {
                    const double l = a.len[k], zm = tm[t * Dw + k], zv = tv[t * Dw + k];
                    const double xi = WiT[k * 64 + r], xj = WjT[k * 64 + c];
                    if (zv != 0.0)
                        prod *= diag ? matern_Jd0(xi, zm, zv, l) : matern_Jd(xj, xi, zm, zv, l);
                    else
                        prod *= matern_point(zm - xi, l) * matern_point(zm - xj, l);
                }
                double pi_ = 1.0, si = 0.0, pj = 1.0, sj = 0.0;
                for (int g = 0; g < Dz; ++g) {
                    const double il = 1.0 / a.len[Dw + g], zz = tz[t * Dz + g];
                    corr_accum_matern((WiT[(Dw + g) * 64 + r] - zz) * il, pi_, si);
                    corr_accum_matern((WjT[(Dw + g) * 64 + c] - zz) * il, pj, sj);
                }
                if (Dz) prod *= pi_ * pj * exp(-SQRT5 * (si + sj));
                acc = fma(cij, prod, acc);
            }
            acc = wave_sum_p(acc);
            if (lane == 0) red[t * 4 + wave] = acc;
        }
    }
    __syncthreads();
    if (tid < nt)
        a.partial[(int64_t)blockIdx.x * a.Mc + (tbase - a.t0) + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
}

// SExp J with the pair loop on MFMA (functions.py:432-451 restated).  For test point t the exponent of pair (i, j) is
//   base_ij + ei_i(t) + ej_j(t) + sum_k c1_k(t) (w_ik - 2 m_k(t) + w_jk)^2,    c1_k = 1 / (8 v_k + 2 l_k^2),
// and the square expands into a row term, a column term and the dot product sum_k [2 c1_k (w_ik - 2 m_k)] w_jk: per
// test point the 64 x 64 x Dw products of a tile are a handful of v_mfma_f64_16x16x4 (accumulated on top of base_ij,
// which sits in the accumulator layout), and the VALU work per pair drops from 3 Dw + exp to 2 adds + exp.  The row /
// column terms and the A operand of test point t+1 are staged (double buffered in LDS) while t is evaluated.
__global__ __launch_bounds__(256, 3) void linkgp_Jsexp_kernel(LinkArgs a) {
    extern __shared__ double lds[];
    const int Dw = a.Dw, Dz = a.Dz, DT = Dw + Dz;
    const int KP = (Dw + 3) & ~3, LDU = KP + 2;   // row stride = 2 (mod 4) doubles: conflict-free ds_read_b64 A fragments
    double *WiT = lds;                    // [DT][64]
    double *WjT = WiT + DT * 64;          // [DT][64]
    double *WjB = WjT + DT * 64;          // [KP][LDK]   B operand: w_jk, k-major, zero padded
    double *tm = WjB + KP * LDK;          // [TCH][Dw]
    double *tv = tm + TCH * Dw;           // [TCH][Dw]
    double *tz = tv + TCH * Dw;           // [TCH][Dz]
    double *red = tz + TCH * Dz;          // [TCH][4]
    double *U = red + TCH * 4;            // [2][64][LDU]  A operand of a test point: 2 c1_k (w_ik - 2 m_k)
    double *R = U + 2 * 64 * LDU;         // [2][64] row terms
    double *S = R + 2 * 64;               // [2][64] column terms
    double *ilg = S + 2 * 64;             // [Dz] reciprocal lengthscales of the global dimensions
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64, n = a.n;
    const int64_t tbase = a.t0 + (int64_t)blockIdx.y * TCH;
    int nt = TCH;
    if (tbase + nt > a.M) nt = (int)(a.M - tbase);
    if (tbase + nt > a.t0 + a.Mc) nt = (int)(a.t0 + a.Mc - tbase);

    for (int idx = tid; idx < 64 * DT; idx += 256) {
        int row = idx / DT, d = idx - row * DT;
        int64_t gi = i0 + row, gj = j0 + row;
        double vi = 0.0, vj = 0.0;
        if (d < Dw) {
            if (gi < n) vi = a.W[gi * Dw + d];
            if (gj < n) vj = a.W[gj * Dw + d];
        } else {
            if (gi < n) vi = a.Wg[gi * Dz + d - Dw];
            if (gj < n) vj = a.Wg[gj * Dz + d - Dw];
        }
        WiT[d * 64 + row] = vi;
        WjT[d * 64 + row] = vj;
    }
    for (int idx = tid; idx < nt * Dw; idx += 256) {
        tm[idx] = a.m[tbase * Dw + idx];
        tv[idx] = a.v[tbase * Dw + idx];
    }
    for (int idx = tid; idx < nt * Dz; idx += 256) tz[idx] = a.z[tbase * Dz + idx];
    for (int g = tid; g < Dz; g += 256) ilg[g] = 1.0 / a.len[Dw + g];
    __syncthreads();
    for (int idx = tid; idx < KP * 64; idx += 256) {
        const int k = idx >> 6, j = idx & 63;
        WjB[k * LDK + j] = k < Dw ? WjT[k * 64 + j] : 0.0;
    }
    // accumulator layout: wave w owns rows 16w..16w+15; element (tile tt, reg r) of a lane = row 16w + (lane>>4) + 4r,
    // column 16tt + (lane&15)
    const int mrow = 16 * wave + (lane >> 4), mcol = lane & 15, kq = lane >> 4, mi = lane & 15;
    const double wt = (bi == bj) ? 1.0 : 2.0;
    d4 Cr[4], base[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = i0 + mrow + 4 * r, gj = j0 + 16 * tt + mcol;
            Cr[tt][r] = (gi < n && gj < n) ? wt * (a.ry[gi] * a.ry[gj] - a.scale * a.Rinv[gi * a.ldr + gj]) : 0.0;
            double b = 0.0;   // t-independent part: sum_k (w_ik - w_jk)^2 / (2 l_k^2)  (= -log R2sexp, kernel_class.py:761-763)
            for (int k = 0; k < Dw; ++k) {
                const double d = WiT[k * 64 + mrow + 4 * r] - WjT[k * 64 + 16 * tt + mcol];
                b = fma(d * d, 1.0 / (2.0 * a.len[k] * a.len[k]), b);
            }
            base[tt][r] = b;
        }
    // per-test-point constants, once: tv <- c1_k(t) = 1 / (8 v_k + 2 l_k^2), tm <- 2 m_k(t); reciprocal lengthscales of the
    // global dimensions (no divisions in the staging below)
    for (int idx = tid; idx < nt * Dw; idx += 256) {
        const double l = a.len[idx % Dw];
        tv[idx] = 1.0 / (8.0 * tv[idx] + 2.0 * l * l);
        tm[idx] = 2.0 * tm[idx];
    }
    // staging of test point t: 128 point tasks (64 row points: A operand + row term; 64 column points: column term)
    // on the even threads, so that the four waves share them evenly
    auto stage = [&](int t, int buf) {
        if (tid & 1) return;
        const int task = tid >> 1;
        const double *c1 = tv + t * Dw, *m2 = tm + t * Dw, *zt = tz + t * Dz;
        if (task < 64) {
            double rr = 0.0;
            double *u = U + (buf * 64 + task) * LDU;
            for (int k = 0; k < Dw; ++k) {
                const double wi = WiT[k * 64 + task] - m2[k], cw = c1[k] * wi;
                u[k] = 2.0 * cw;
                rr = fma(cw, wi, rr);
            }
            for (int k = Dw; k < KP; ++k) u[k] = 0.0;
            for (int g = 0; g < Dz; ++g) {
                const double di = (WiT[(Dw + g) * 64 + task] - zt[g]) * ilg[g];
                rr = fma(di, di, rr);
            }
            R[buf * 64 + task] = rr;
        } else {
            const int j = task - 64;
            double ss = 0.0;
            for (int k = 0; k < Dw; ++k) {
                const double wj = WjT[k * 64 + j];
                ss = fma(c1[k] * wj, wj, ss);
            }
            for (int g = 0; g < Dz; ++g) {
                const double dj = (WjT[(Dw + g) * 64 + j] - zt[g]) * ilg[g];
                ss = fma(dj, dj, ss);
            }
            S[buf * 64 + j] = ss;
        }
    };
    __syncthreads();
    if (nt > 0) stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        __syncthreads();   // stage(t) visible; everybody is done with the other buffer
        if (t + 1 < nt) stage(t + 1, buf ^ 1);
        const vlds_double *Ur = (const vlds_double *)(U + (buf * 64 + 16 * wave + mi) * LDU);
        const vlds_double *Bv = (const vlds_double *)WjB;
        d4 e[4] = {base[0], base[1], base[2], base[3]};
        for (int k0 = 0; k0 < KP; k0 += 4) {
            const double av = Ur[k0 + kq];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                e[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bv[(k0 + kq) * LDK + 16 * tt + mi], e[tt], 0, 0, 0);
        }
        double rr[4], acc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[r] = R[buf * 64 + mrow + 4 * r];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const double ss = S[buf * 64 + 16 * tt + mcol];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fma(Cr[tt][r], exp_negated(e[tt][r] + rr[r] + ss), acc);
        }
        acc = wave_sum_p(acc);
        if (lane == 0) red[t * 4 + wave] = acc;
    }
    __syncthreads();
    if (tid < nt)
        a.partial[(int64_t)blockIdx.x * a.Mc + (tbase - a.t0) + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
}

// ---- SExp J, second form: per-(test point, training point) records + a pair loop without LDS -------------------------
// The exponent of pair (i, j) at test point t is  base_ij + sum_k u_ik(t) w_jk + rr_i(t) + ss_j(t)  (see above).  The t-
// dependent per-point quantities -- u (Dw values), rr, ss -- are computed ONCE per (t, point) by sexp_records_kernel
// instead of once per tile pair inside the pair kernel (where that staging was a quarter of its VALU instructions), and the
// two additive terms ride in the MFMA as two more k-columns:  A_i = [u_i | rr_i | 1 | 0..],  B_j = [w_j ; 1 ; ss_j ; 0..]
// (k padded to a multiple of 4).  The B operand's rows are t-independent except the ss row, so a lane keeps its B fragments
// in REGISTERS for all test points of the chunk and the lanes that hold the ss row load theirs from the records; the A
// fragments of a test point are three or four 8-byte loads per lane from the records (L2 / Infinity Cache resident: the
// grid runs the tiles of a test chunk back to back), prefetched one test point ahead.  The MFMA's C input is base_ij, its
// output the whole exponent: per pair the VALU work is the exponential and one multiply-add.  No LDS traffic, no barrier
// in the loop over test points.  For Dw <= 14 (four k-steps); wider nodes keep linkgp_Jsexp_kernel.
__global__ __launch_bounds__(256) void sexp_records_kernel(LinkArgs a, int KPA) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tt = blockIdx.y, t = a.t0 + tt;
    if (i >= a.npad || t >= a.M) return;
    double *ra = a.recs + (tt * a.npad + i) * KPA;
    double rr = 0.0, ss = 0.0;
    if (i < a.n) {
        for (int k = 0; k < a.Dw; ++k) {
            const double l = a.len[k], c1 = 1.0 / (8.0 * a.v[t * a.Dw + k] + 2.0 * l * l), w = a.W[i * a.Dw + k];
            const double d = w - 2.0 * a.m[t * a.Dw + k], cw = c1 * d;
            ra[k] = 2.0 * cw;
            rr = fma(cw, d, rr);
            ss = fma(c1 * w, w, ss);
        }
        // (a point sees the same global-input term in its row and in its column role; added in the first form's order)
        for (int g = 0; g < a.Dz; ++g) {
            const double di = (a.Wg[i * a.Dz + g] - a.z[t * a.Dz + g]) * (1.0 / a.len[a.Dw + g]);
            rr = fma(di, di, rr);
            ss = fma(di, di, ss);
        }
    } else {
        for (int k = 0; k < a.Dw; ++k) ra[k] = 0.0;
    }
    ra[a.Dw] = rr;
    ra[a.Dw + 1] = 1.0;
    for (int k = a.Dw + 2; k < KPA; ++k) ra[k] = 0.0;
    a.gfac[tt * a.npad + i] = ss;
}

#define SX_KS 4   // k-steps held in registers: Dw + 2 <= 16
// KS = k-steps (compile time: no branches in the loop over test points).  The MFMAs of test point t + 1 are issued between
// the four column-tile groups of test point t's exponentials.  The loop is unrolled by two so that the exponent tiles of
// consecutive test points alternate between two register sets without copies.
#define TCH2 128   // test points per workgroup of the second SExp form (256 where that still leaves >= 32 rounds of workgroups: sexp_tch2): the tile's weights and base exponents (a 32-KB tile of
                   // R^-1 read, 16 x Dw LDS reads per lane) are set up once per TCH2 test points -- at 32 that was 40 % of the run time
template <int KS, bool TAB>
__global__ __launch_bounds__(256, 2) void linkgp_Jsexp2_kernel(LinkArgs a) {
    extern __shared__ double lds[];
    constexpr int KPA = 4 * KS;
    const int Dw = a.Dw;
    double *WiT = lds;                    // [Dw][64]
    double *WjT = WiT + Dw * 64;          // [Dw][64]
    const int tch2 = a.tch;               // (this kernel's test points per workgroup)
    double *red = WjT + Dw * 64;          // [tch2][4]
    double *etab = red + tch2 * 4;        // [EXPN_TAB]: 2^(j/EXPN_TAB), exp_negated_tab's table
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = tid; j < EXPN_TAB; j += 256) etab[j] = exp2((double)j * (1.0 / EXPN_TAB));
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64, n = a.n;
    const int64_t tbase = a.t0 + (int64_t)blockIdx.y * tch2, tt0 = (int64_t)blockIdx.y * tch2;
    int nt = tch2;
    if (tbase + nt > a.M) nt = (int)(a.M - tbase);
    if (tbase + nt > a.t0 + a.Mc) nt = (int)(a.t0 + a.Mc - tbase);
    for (int idx = tid; idx < 64 * Dw; idx += 256) {
        const int row = idx / Dw, d = idx - row * Dw;
        const int64_t gi = i0 + row, gj = j0 + row;
        WiT[d * 64 + row] = gi < n ? a.W[gi * Dw + d] : 0.0;
        WjT[d * 64 + row] = gj < n ? a.W[gj * Dw + d] : 0.0;
    }
    __syncthreads();
    const int mrow = 16 * wave + (lane >> 4), mcol = lane & 15, kq = lane >> 4, mi = lane & 15;
    const int kqS = (Dw + 1) & 3;   // (the ss row: k = Dw + 1, in k-step (Dw + 1) >> 2 == KS - 1)
    const bool dyn = kq == kqS;   // this lane holds the ss row in its B fragment of the last k-step
    const double wt = (bi == bj) ? 1.0 : 2.0;
    d4 Cr[4], base[4];
    // The tile's weights: all 24 loads go out back to back, at clamped indices with the bound applied to the value, and land under the base exponents'
    // arithmetic.  (Written as a conditional load inside the element loop every element was a branch with a load round trip of its own -- s_waitcnt
    // vmcnt(1), vmcnt(0), sixteen times, ~15 us of a workgroup's 130-260.  No LDS-DMA in this kernel: register loads in flight together are safe.)
    {
        double ryr[4], ryc[4];
        int64_t cir[4], cjc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = i0 + mrow + 4 * r;
            cir[r] = gi < n ? gi : n - 1;
            ryr[r] = a.ry[cir[r]];
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int64_t gj = j0 + 16 * tt + mcol;
            cjc[tt] = gj < n ? gj : n - 1;
            ryc[tt] = a.ry[cjc[tt]];
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cr[tt][r] = a.Rinv[cir[r] * a.ldr + cjc[tt]];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double b = 0.0;   // t-independent part: sum_k (w_ik - w_jk)^2 / (2 l_k^2)
                for (int k = 0; k < Dw; ++k) {
                    const double d = WiT[k * 64 + mrow + 4 * r] - WjT[k * 64 + 16 * tt + mcol];
                    b = fma(d * d, 1.0 / (2.0 * a.len[k] * a.len[k]), b);
                }
                base[tt][r] = b;
            }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = i0 + mrow + 4 * r < n && j0 + 16 * tt + mcol < n;
                Cr[tt][r] = in ? wt * (ryr[r] * ryc[tt] - a.scale * Cr[tt][r]) : 0.0;
            }
    }
    // static B fragments: B[k][j], k = 4 ks + kq, j = 16 tt + mi (the ss row's lanes get theirs per test point)
    double bst[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int k = 4 * ks + kq;
            bst[ks][tt] = k < Dw ? WjT[k * 64 + 16 * tt + mi] : (k == Dw ? 1.0 : 0.0);
        }
    const double *recA = a.recs + (tt0 * a.npad + i0 + 16 * wave + mi) * KPA + kq;   // + t * npad * KPA + 4 ks
    const double *recS = a.gfac + tt0 * a.npad + j0 + mi;                               // + t * npad + 16 tt
    const int64_t strA = a.npad * KPA, strS = a.npad;
    struct Frag { double av[KS], bd[4]; };
    auto fetch = [&](int t) {
        Frag f;
        t = t < nt ? t : nt - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f.av[ks] = recA[t * strA + 4 * ks];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) f.bd[tt] = dyn ? recS[t * strS + 16 * tt] : bst[KS - 1][tt];   // (the ss row lies in the LAST k-step: (Dw + 1) >> 2 == KS - 1)
        return f;
    };
    // one k-step of test point f's exponent tiles into `en` (four independent MFMAs)
    auto kstep = [&](const Frag &f, int ks, d4 (&en)[4]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const double bv = (ks == KS - 1) ? f.bd[tt] : bst[ks][tt];   // (selected once per test point, in fetch: three selects per k-step and tile here otherwise)
            en[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.av[ks], bv, ks == 0 ? base[tt] : en[tt], 0, 0, 0);
        }
    };
    // exponentials of column tile g of `e`, weights Cr.  TAB: the table reads of tile g + 1 are issued (begin) before tile g's arithmetic and pinned there,
    // so that their LDS round trip runs under it (they sat, one by one, directly in front of their use).
    double kfa[4], ta[4], kfb[4], tb[4];
    auto begin = [&](const d4 (&e)[4], int g, double (&kf)[4], double (&t)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) exp_negated_tab_begin(e[g][r], etab, kf[r], t[r]);
    };
    auto group = [&](const d4 (&e)[4], int g, double &acc, const double (&kf)[4], const double (&t)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fma(Cr[g][r], TAB ? exp_negated_tab_end(e[g][r], kf[r], t[r]) : exp_negated(e[g][r]), acc);
    };
    // test point t from `e`, while test point t + 1 (fragments f1) goes into `en`; f2 <- fragments of t + 2
    auto step = [&](int t, const d4 (&e)[4], d4 (&en)[4], const Frag &f1, Frag &f2) {
        f2 = fetch(t + 2);
        if (TAB) begin(e, 0, kfa, ta);
        __builtin_amdgcn_sched_barrier(0);
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // k-step g of the NEXT test point (four independent MFMAs) and the four exponentials of column tile g of THIS one
            // in one scheduling region, the scheduler asked for  MFMA, ~20 VALU, MFMA, ~20 VALU, ...  (inside the VALU slots the
            // four exponentials still interleave; a barrier after every exponential serialised their dependent chains: 4x
            // slower).  Measured: the f64 MFMA and the f64 VALU work do NOT overlap on gfx950 whatever the order -- the run time
            // is the sum of the two (41 ms = 29 without the products + 12; 78.6 TFLOP/s is the matrix AND the vector f64 peak:
            // one set of double-precision units) -- so the order only keeps the products' operands off the critical path.
            if (TAB && g < 3) {
                if (g & 1) begin(e, g + 1, kfa, ta);
                else begin(e, g + 1, kfb, tb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g < KS) kstep(f1, g, en);
            if (g & 1) group(e, g, acc, kfb, tb);
            else group(e, g, acc, kfa, ta);
            if (g < KS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, TAB ? 16 : 21, 0);   // the VALU instructions of about one exponential
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        acc = wave_sum_p(acc);
        if (lane == 0) red[t * 4 + wave] = acc;
    };
    if (nt > 0) {
        Frag fa = fetch(0), fb = fetch(1), fc;
        d4 e0[4], e1[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kstep(fa, ks, e0);
        for (int t = 0; t < nt; t += 2) {
            step(t, e0, e1, fb, fc);        // t from e0; t + 1 -> e1 (fragments fb); fc <- t + 2
            if (t + 1 < nt) step(t + 1, e1, e0, fc, fb);   // t + 1 from e1; t + 2 -> e0 (fragments fc); fb <- t + 3
        }
    }
    __syncthreads();
    if (tid < nt)
        a.partial[(int64_t)blockIdx.x * a.Mc + (tbase - a.t0) + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
}

// Matern-2.5 J through the separable S/T form (linkfun.hpp).
//   matern_records_kernel : once per (test point, dimension, training point): S[0..11] T[12..26] f2[27] x[28] 0[29]  (REC = 30)
//   linkgp_Jsep_kernel    : one WG per (lower 64x64 tile of C) x (chunk of TCH test points); per (t, k) the 128
//                           records of its two point blocks are streamed global -> registers -> LDS one step
//                           ahead of the pair phase; every pair costs 30 FMAs (both orientations) and a select
//                           instead of 3 erf + 5 exp + ~300 flops.  The grid runs tiles fastest so that a
//                           test-chunk's records (TCH*Dw*n*240 B) are re-read from the Infinity Cache.
#define REC 30   // S[0..11] T[12..26] f2[27] x[28] 0[29]: the record as the pair kernel holds it in LDS (stride 30 doubles -> conflict-free
                 // column reads; the zero pads the 3-term erf-difference products to an MFMA k-step of 4)
#define PST REC
#define MC_SEP 256
#define TCHS 32    // test points per workgroup of linkgp_Jsep_kernel

__global__ __launch_bounds__(256) void matern_records_kernel(LinkArgs a) {
    // 128 points x (S role | T role) per workgroup; the 128 records are staged in LDS and leave as one contiguous
    // 30.7 KB block (a thread writing its own 240-byte record touched 64 cache lines per store instruction)
    __shared__ double stage[128 * (REC + 1)];
    const int tid = threadIdx.x, pp = tid & 127, role = tid >> 7;
    const int64_t i0 = (int64_t)blockIdx.x * 128, i = i0 + pp;
    const int k = blockIdx.y;
    const int64_t tt = blockIdx.z, t = a.t0 + tt;
    if (t >= a.M) return;
    const double x = (i < a.n) ? a.W[i * a.Dw + k] : 0.0;
    const double l = a.len[k], zm = a.m[t * a.Dw + k], zv = a.v[t * a.Dw + k];
    double *rec = stage + pp * (REC + 1);
    if (zv != 0.0) {
        MaternDimConst kc;
        matern_dim_const(zm, zv, l, kc);
        if (role == 0) {
            double f2;
            double out[12];
            matern_role_S(x, kc, out, f2);
#pragma unroll
            for (int c = 0; c < 12; ++c) rec[c] = out[c];
            rec[27] = f2;
            rec[28] = x;
        } else {
            double out[15];
            matern_role_T(x, kc, out);
#pragma unroll
            for (int c = 0; c < 15; ++c) rec[12 + c] = out[c];
            rec[29] = 0.0;
        }
    } else {   // deterministic input in this dimension: J factor = k(x_i, m) k(x_j, m)  (functions.py:488-491)
        const double pt = matern_point(zm - x, l);
        if (role == 0) {
            rec[0] = pt;
#pragma unroll
            for (int c = 1; c < 12; ++c) rec[c] = 0.0;
            rec[27] = 0.0;
            rec[28] = x;
        } else {
            rec[12] = pt;
#pragma unroll
            for (int c = 13; c < 27; ++c) rec[c] = 0.0;
            rec[29] = 0.0;
        }
    }
    __syncthreads();
    const int64_t npts = a.npad - i0 < 128 ? a.npad - i0 : 128;
    double *dst = a.recs + ((tt * a.Dw + k) * a.npad + i0) * REC;
    for (int e = tid; e < (int)npts * REC; e += 256) dst[e] = stage[(e / REC) * (REC + 1) + e % REC];
}

// Deterministic global inputs: the separable Matern factor k(Wg_i, z_t) of (training point, test point)
// (functions.py:413-420), once per pair of points instead of once per tile pair in the J kernel.
__global__ __launch_bounds__(256) void global_factor_kernel(LinkArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tt = blockIdx.y, t = a.t0 + tt;
    if (i >= a.npad || t >= a.M) return;
    double pr = 1.0, sm = 0.0;
    if (i < a.n)
        for (int g = 0; g < a.Dz; ++g)
            corr_accum_matern((a.Wg[i * a.Dz + g] - a.z[t * a.Dz + g]) / a.len[a.Dw + g], pr, sm);
    a.gfac[tt * a.npad + i] = pr * exp(-SQRT5 * sm);
}

template <int PIPE, bool LOG>
__global__ __launch_bounds__(256, 2) void linkgp_Jsep_kernel(LinkArgs a) {
    extern __shared__ double lds[];
    const long long t_entry = LOG ? wall_clock64() : 0;   // (step log: when the workgroup began, ahead of its set-up)
    const int Dw = a.Dw, Dz = a.Dz, DT = Dw + Dz;
    double *WiT = lds;                    // [DT][64]
    double *WjT = WiT + DT * 64;          // [DT][64]
    const int tch = a.tch;
    double *tz = WjT + DT * 64;           // [tch][Dz]
    double *red = tz + tch * Dz;          // [tch][4]
    double *PT = red + tch * 4;           // [2][128][PST]
    int *smode = reinterpret_cast<int *>(PT + 2 * 128 * PST);   // [Dw][4]: per (dimension, wave) the order class of its rows against the tile's columns
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64, n = a.n;
    const int64_t tbase = a.t0 + (int64_t)blockIdx.y * tch;
    int nt = tch;
    if (tbase + nt > a.M) nt = (int)(a.M - tbase);
    if (tbase + nt > a.t0 + a.Mc) nt = (int)(a.t0 + a.Mc - tbase);

    for (int idx = tid; idx < 64 * DT; idx += 256) {
        int row = idx / DT, d = idx - row * DT;
        int64_t gi = i0 + row, gj = j0 + row;
        double vi = 0.0, vj = 0.0;
        if (d < Dw) {
            if (gi < n) vi = a.W[gi * Dw + d];
            if (gj < n) vj = a.W[gj * Dw + d];
        } else {
            if (gi < n) vi = a.Wg[gi * Dz + d - Dw];
            if (gj < n) vj = a.Wg[gj * Dz + d - Dw];
        }
        WiT[d * 64 + row] = vi;
        WjT[d * 64 + row] = vj;
    }
    for (int idx = tid; idx < nt * Dz; idx += 256) tz[idx] = a.z[tbase * Dz + idx];
    // MFMA accumulator layout of the 64x64 tile: wave w owns rows 16w..16w+15; element (tile tt, reg r) of a lane is
    // row 16w + (lane>>4) + 4r, column 16tt + (lane&15)  (v_mfma_f64_16x16x4_f64 C/D map).
    const int mrow = 16 * wave + (lane >> 4), mcol = lane & 15, kq = lane >> 4, mi = lane & 15;
    const double wt = (bi == bj) ? 1.0 : 2.0;
    d4 Cr[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = i0 + mrow + 4 * r, gj = j0 + 16 * tt + mcol;
            Cr[tt][r] = (gi < n && gj < n) ? wt * (a.ry[gi] * a.ry[gj] - a.scale * a.Rinv[gi * a.ldr + gj]) : 0.0;
        }
    // The 2 x 64 records of step (t, k) go from global memory straight into LDS (global_load_lds_dwordx4: one wave-instruction
    // moves 1 KB; the record in memory IS the LDS record, so the two blocks' 15 KB each are plain copies): 30 wave-loads per
    // step, issued right after the step's barrier into the buffer the previous step read, landed by the next barrier.  No
    // registers, no LDS stores, no address arithmetic per element (the register-staged version spent a fifth of the kernel
    // on its 7 loads + 16 LDS stores per thread and step).
    // (debug, timing only: no_order_classes & 2 makes every workgroup read the records of the first point block -- what the pair phase costs
    //  when its records come from the XCD's L2 instead of the fabric)
    const int64_t irec = (a.no_order_classes & 2) ? 0 : i0, jrec = (a.no_order_classes & 2) ? 0 : j0;
    // (addresses: the records of step s = t Dw + k of this chunk start at rec0 + s x npad x REC, so a step costs one scalar multiply-add; the wave number is made
    //  uniform so that the KB numbers, the LDS destinations (M0) and the `q < 30` tests are scalar -- computed per lane they were 16 VALU instructions, four
    //  readfirstlanes and four exec-masked branches per step, between the barrier and the first fragment read)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const double *rec0 = a.recs + (tbase - a.t0) * Dw * a.npad * REC + lane * 2;
    const int64_t sstride = a.npad * REC;
    const int qlim = (a.no_order_classes & 32) ? 15 : 30;
    auto stage = [&](int s1, double *P, int u0 = 0, int u1 = 8) {
        const double *base = rec0 + s1 * sstride;
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int q = wv + 4 * u;   // KB number q of the 30-KB image: 0..14 the row block, 15..29 the column block
            if (q < qlim) {
                const double *src = base + (q >= 15 ? jrec * REC + (q - 15) * 128 : irec * REC + q * 128);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(P + q * 128), 16, 0, 0);
            }
        }
    };
    const int iSd = kq < 3 ? 6 + kq : 29, iTd = kq < 3 ? 24 + kq : 29;   // (unconditional loads: no exec-masked branches in the pair loop)
    __syncthreads();
    // Order classes.  The J factor of a pair is <S(x_lo), T(x_hi)> + (f2_hi - f2_lo) <S'(x_lo), T'(x_hi)> with lo / hi the pair's
    // smaller / larger coordinate in dimension k: a wave's 16 rows whose coordinates all lie at or below those of the tile's 64
    // columns (class 1) or all above them (class 2) need ONE of the two record products, 4 MFMAs per 16 x 16 sub-tile instead
    // of 8 and no select.  Training points that arrive ordered by cells (dgp_amd.ops.cell_order) make about half of the (rows,
    // tile, dimension) triples such; any other order is class 0 throughout and costs what it did.  (Padding rows / columns are staged as x = 0 and
    // take part in the bounds, so that every element -- also the ones C zeroes -- gets the orientation the select would pick.)
    for (int idx = tid; idx < Dw * 4; idx += 256) {
        const int k = idx >> 2, w = idx & 3;
        double rlo = WiT[k * 64 + 16 * w], rhi = rlo, clo = WjT[k * 64], chi = clo;
        for (int r = 1; r < 16; ++r) {
            const double x = WiT[k * 64 + 16 * w + r];
            rlo = x < rlo ? x : rlo;
            rhi = x > rhi ? x : rhi;
        }
        for (int c = 1; c < 64; ++c) {
            const double x = WjT[k * 64 + c];
            clo = x < clo ? x : clo;
            chi = x > chi ? x : chi;
        }
        smode[idx] = (a.no_order_classes & 4) ? 1 : ((a.no_order_classes & 1) ? 0 : (rhi <= clo ? 1 : (rlo > chi ? 2 : 0)));
    }
    __syncthreads();
    // The records of step (t, k) are double buffered in LDS: ONE barrier per step.
    double *PT1 = PT + 128 * PST;
    const int nstep = nt * Dw;
    if (nstep > 0) stage(0, PT);
    int step = 0;
    int mw_next = __builtin_amdgcn_readfirstlane(smode[wave]);
    // (diagnostics, LOG only: workgroup start / end on the 100-MHz clock and in shader cycles, where it ran, and for its first
    //  JSEP_LOG_STEPS steps every wave's arrival at and departure from the step's barrier -- tools/gpu_pair_steplog.py)
    long long *lg = nullptr;
    long long *slog = reinterpret_cast<long long *>(smode + Dw * 4 + (Dw & 1) * 4);   // (LOG: JSEP_LOG_STEPS x 4 x 2 stamps behind the kernel's own LDS)
    if (LOG && a.dbg) {
        for (int e = tid; e < JSEP_LOG_STEPS * 8; e += 256) slog[e] = 0;
        lg = a.dbg + 64 + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * JSEP_LOG_WORDS;
        if (tid == 0) {
            lg[0] = wall_clock64();
            lg[2] = clock64();
            lg[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            lg[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            lg[6] = ((long long)bi << 16) | bj;
            lg[7] = t_entry;   // (the steps of a full chunk: a.tch x Dw)
        }
    }
    for (int t = 0; t < nt; ++t) {
        d4 prod[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) prod[tt] = (d4){1.0, 1.0, 1.0, 1.0};
#pragma unroll 1
        for (int k = 0; k < Dw; ++k, ++step) {
            double *P = (step & 1) ? PT1 : PT;
            if (LOG && lg && step < JSEP_LOG_STEPS && lane == 0) slog[(step * 4 + wave) * 2] = clock64();   // (kept in LDS: a global store here would be waited for by the barrier's vmcnt(0))
            // records of this step landed; every wave is done with the other buffer.  The drain is explicit: a workgroup-scope fence orders LDS and scalar
            // traffic only (lgkmcnt), and the other waves read this wave's DMA right behind the barrier -- the compiler happens to put its own vmcnt(0)
            // here, nothing obliges it to (tests/test_isa_hazards.py checks the emitted code)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (LOG && lg && step < JSEP_LOG_STEPS && lane == 0) slog[(step * 4 + wave) * 2 + 1] = clock64() | ((long long)mw_next << 60);
            // the next step's records into the other buffer.  Past the end the last step is staged again (harmless).
            const int s1 = step + 1 < nstep ? step + 1 : nstep - 1;
            const bool nostage = (a.no_order_classes & 8) != 0;   // (timing only)
            if (PIPE == 0 && !nostage) stage(s1, (step & 1) ? PT : PT1);   // (comparison build: all eight right behind the barrier)
            // volatile: keeps these fragment reads as ds_read_b64 (2 LDS cycles, 64 banks: conflict-free with PST = 30); merged
            // into ds_read2_b64 by the compiler they take 8 cycles and bank modulo 32 (2-way conflicts here)
            const vlds_double *Arow = (const vlds_double *)(P + (16 * wave + mi) * PST);   // this lane's row record as an MFMA A operand (i = lane&15)
            double f2r[4], xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f2r[r] = P[(mrow + 4 * r) * PST + 27];
                xr[r] = P[(mrow + 4 * r) * PST + 28];
            }
            const int mw = mw_next;   // (read one step ahead: the LDS round trip + readfirstlane sat between the barrier and the first MFMA)
            const int mraw = ((const volatile int *)smode)[(k + 1 < Dw ? k + 1 : 0) * 4 + wave];
            if (mw == 0) {   // mixed: both record products, selected per pair
                // A fragments (rows): S[0..11] and T[0..11] in three k-steps each, the erf-difference pair in one (k=3 padded with 0)
                double aS[3], aT[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    aS[ks] = Arow[4 * ks + kq];
                    aT[ks] = Arow[12 + 4 * ks + kq];
                }
                const double aSd = Arow[iSd], aTd = Arow[iTd];
                // Column tiles software pipelined by hand: the 8 MFMAs of tile tt+1 are issued in pairs between the four
                // row groups of tile tt's (VALU) epilogue; the sched_barrier() fences pin that order (on its own the compiler
                // issues all 32 MFMAs first and all epilogues afterwards).
                d4 o1[2], o2[2], e1[2], e2[2];
                // The column fragments are read TWO tiles ahead (two register sets) and a tile's f2 / x one tile ahead: the LDS round trip of a
                // fragment read sat in front of every MFMA group otherwise (read, wait, MFMA), and with two waves per SIMD nobody hides it.
                double bf[2][8], f2c[2], xc[2];
                auto loadB = [&](int tt, int s) {   // column records as MFMA B operands (j = lane&15)
                    const vlds_double *Bcol = (const vlds_double *)(P + (64 + 16 * tt + mi) * PST);
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        bf[s][ks] = Bcol[12 + 4 * ks + kq];
                        bf[s][3 + ks] = Bcol[4 * ks + kq];
                    }
                    bf[s][6] = Bcol[iTd];
                    bf[s][7] = Bcol[iSd];
                };
                auto loadC = [&](int tt, int s) {
                    const vlds_double *Ccol = (const vlds_double *)(P + (64 + 16 * tt + mcol) * PST);
                    f2c[s] = Ccol[27];
                    xc[s] = Ccol[28];
                };
                auto mm2 = [&](int c, int slot) {   // MFMA pair c of a tile: S_row . T_col and T_row . S_col, then the erf pair
                    const d4 z = {0.0, 0.0, 0.0, 0.0};
                    if (c < 3) {
                        o1[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[c], bf[slot][c], c ? o1[slot] : z, 0, 0, 0);
                        o2[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aT[c], bf[slot][3 + c], c ? o2[slot] : z, 0, 0, 0);
                    } else {
                        e1[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aSd, bf[slot][6], z, 0, 0, 0);
                        e2[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aTd, bf[slot][7], z, 0, 0, 0);
                    }
                };
                loadB(0, 0);
                loadC(0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) mm2(c, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int sl = tt & 1;
                    if (PIPE == 2 && !nostage) stage(s1, (step & 1) ? PT : PT1, 2 * tt, 2 * tt + 2);
                    if (tt < 3) loadC(tt + 1, sl ^ 1);
                    if (tt < 3) loadB(tt + 1, sl ^ 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (tt < 3) mm2(r, sl ^ 1);
                        const double d = f2c[sl] - f2r[r];
                        const double j1 = fma(d, e1[sl][r], o1[sl][r]), j2 = fma(-d, e2[sl][r], o2[sl][r]);
                        prod[tt][r] *= (xr[r] <= xc[sl]) ? j1 : j2;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // one orientation for the whole 16 x 64 strip.  Class 1 (rows at or below columns): <S_row, T_col> + d <S'_row, T'_col>;
                // class 2 (rows above columns): <T_row, S_col> - d <T'_row, S'_col>, d = f2_col - f2_row.  The two differ in record
                // offsets and one sign only; the pipelining is the mixed path's with four MFMAs per sub-tile.
                const bool c1 = mw == 1;
                const int offA = c1 ? 0 : 12, offB = c1 ? 12 : 0, iAd = c1 ? iSd : iTd, iBd = c1 ? iTd : iSd;
                const double aX0 = Arow[offA + kq], aX1 = Arow[offA + 4 + kq], aX2 = Arow[offA + 8 + kq], aD = Arow[iAd];
                double g2r[4];   // (-d = f2_row - f2_col exactly: the signs go onto the two differences' operands)
#pragma unroll
                for (int r = 0; r < 4; ++r) g2r[r] = c1 ? f2r[r] : -f2r[r];
                d4 oo[2], ee[2];
                double bx[2][4], g2c[2];
                auto loadBc = [&](int tt, int s) {
                    const vlds_double *Bcol = (const vlds_double *)(P + (64 + 16 * tt + mi) * PST);
                    bx[s][0] = Bcol[offB + kq];
                    bx[s][1] = Bcol[offB + 4 + kq];
                    bx[s][2] = Bcol[offB + 8 + kq];
                    bx[s][3] = Bcol[iBd];
                };
                auto loadCc = [&](int tt, int s) {
                    const vlds_double *Ccol = (const vlds_double *)(P + (64 + 16 * tt + mcol) * PST);
                    g2c[s] = Ccol[27];
                };
                auto mmc = [&](int c, int slot) {
                    const d4 z = {0.0, 0.0, 0.0, 0.0};
                    if (c == 0) oo[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aX0, bx[slot][0], z, 0, 0, 0);
                    else if (c == 1) ee[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aD, bx[slot][3], z, 0, 0, 0);
                    else if (c == 2) oo[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aX1, bx[slot][1], oo[slot], 0, 0, 0);
                    else oo[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(aX2, bx[slot][2], oo[slot], 0, 0, 0);
                };
                loadBc(0, 0);
                loadCc(0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) mmc(c, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int sl = tt & 1;
                    if (PIPE == 2 && !nostage) stage(s1, (step & 1) ? PT : PT1, 2 * tt, 2 * tt + 2);
                    if (tt < 3) loadCc(tt + 1, sl ^ 1);
                    if (tt < 3) loadBc(tt + 1, sl ^ 1);
                    const double g2 = c1 ? g2c[sl] : -g2c[sl];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (tt < 3) mmc(r, sl ^ 1);
                        prod[tt][r] *= fma(g2 - g2r[r], ee[sl][r], oo[sl][r]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            mw_next = __builtin_amdgcn_readfirstlane(mraw);
        }
        // deterministic global inputs: separable Matern factor (functions.py:413-420), precomputed per point
        double gr[4], gc[4];
        if (Dz) {
            const double *gf = a.gfac + ((tbase - a.t0) + t) * a.npad;
#pragma unroll
            for (int r = 0; r < 4; ++r) gr[r] = gf[i0 + mrow + 4 * r];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) gc[tt] = gf[j0 + 16 * tt + mcol];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) gr[r] = gc[r] = 1.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fma(Cr[tt][r], prod[tt][r] * gr[r] * gc[tt], acc);
        acc = wave_sum_p(acc);
        if (lane == 0) red[t * 4 + wave] = acc;
    }
    // (the compiler's barrier waits for LDS and scalar traffic only -- s_waitcnt lgkmcnt(0); s_barrier; s_endpgm in the emitted code: the last step's redundant
    //  DMA, still in flight, would land in the LDS of whichever workgroup is given that space next)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (LOG && lg) {
        if (tid == 0) {
            lg[1] = wall_clock64();
            lg[3] = clock64();
        }
        for (int e = tid; e < JSEP_LOG_STEPS * 8; e += 256) lg[8 + e] = slog[e];
    }
    if (tid < nt)
        a.partial[(int64_t)blockIdx.x * a.Mc + (tbase - a.t0) + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
}

template <int KIND>
__global__ __launch_bounds__(256) void linkgp_finalize_kernel(LinkArgs a, int ntiles) {   // one wave per test point
    const int lane = threadIdx.x & 63;
    const int64_t tt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t t = a.t0 + tt;
    if (tt >= a.Mc || t >= a.M) return;
    double s = 0.0;
    for (int b = lane; b < ntiles; b += 64) s += a.partial[(int64_t)b * a.Mc + tt];
    s = wave_sum_p(s);
    if (lane) return;
    if (KIND == DGPAMD_SEXP) {
        double c = 1.0;
        for (int k = 0; k < a.Dw; ++k) c *= 1.0 + 4.0 * a.v[t * a.Dw + k] / (a.len[k] * a.len[k]);
        s *= 1.0 / sqrt(c);
    }
    const double mu = a.mean[t];
    a.var[t] = fabs(s - mu * mu + a.scale * (1.0 + a.nugget));
}

// Test points per launch of the record-based pair kernels (Matern: linkgp_Jsep_kernel, `per_wg` = TCH; SExp: linkgp_Jsexp2_kernel,
// TCH2).  A launch is ntiles x (points / per_wg) workgroups on the chip's 512 workgroup slots, and its last, ragged round leaves
// slots idle: at n = 2000 with 256 points per launch that was 8.25 rounds and 10 % of the kernel's time (tools/gpu_pair_steplog.py:
// slot occupancy 0.32 over the last tenth of the launch).  >= 32 rounds per launch bound that loss by 1-2 %; the records of a launch
// (Matern: points x Dw x npad x 240 B) are kept under 4 GiB, and a launch never holds fewer than MC_SEP points.
static int64_t pair_chunk(int64_t nb, int64_t Mc, int Dw, int per_wg, int64_t rec_doubles_per_point) {
    const int64_t ntiles = nb * (nb + 1) / 2;
    int64_t want = ((32 * 512 + ntiles - 1) / ntiles) * per_wg;
    const int64_t cap = (((int64_t)4 << 30) / (rec_doubles_per_point * (int64_t)sizeof(double))) / per_wg * per_wg;
    if (want > cap) want = cap;
    if (want < MC_SEP) want = MC_SEP;
    return Mc < want ? Mc : want;
}
// doubles of records per test point: Matern (Dw x npad records of REC + the global factor), SExp second form (npad x (KPA <= 16) + ss)
static int64_t matern_rec_doubles(int64_t nb, int Dw) { return (int64_t)(Dw * REC + 1) * nb * 64; }
static int64_t sexp_rec_doubles(int64_t nb, int Dw) { return (int64_t)(((Dw + 2 + 3) & ~3) + 1) * nb * 64; }

// Test points per workgroup of linkgp_Jsexp2_kernel: a workgroup's set-up (the tile's weights and base exponents) is amortised over them -- 256 where a launch of
// MC_MAX points still has 32 rounds of workgroups (n = 5000: 28.5 -> 27.4 ms per 2048 points), 128 below (n = 2000: 256 would leave 8 rounds and cost 6 %).
static int sexp_tch2(int64_t nb) {
    if (getenv("DGPAMD_JSEXP_TCH")) {
        const int c = atoi(getenv("DGPAMD_JSEXP_TCH"));
        if (c == 64 || c == 128 || c == 256) return c;
    }
    return nb * (nb + 1) / 2 * (MC_MAX / 256) >= 32 * 512 ? 256 : TCH2;
}

static int jsep_tch() {   // test points per workgroup of linkgp_Jsep_kernel (DGPAMD_JSEP_TCH: comparison runs)
    if (getenv("DGPAMD_JSEP_TCH")) {
        const int c = atoi(getenv("DGPAMD_JSEP_TCH"));
        if (c >= 8 && c <= 256 && c % 8 == 0) return c;
    }
    return TCHS;
}

extern "C" size_t dgpamd_linkgp_workspace(int64_t n, int64_t M, int Dw) {
    int64_t nb = (n + 63) / 64;
    int64_t Mc = ((M + TCH - 1) / TCH) * TCH;
    if (Mc > MC_MAX) Mc = MC_MAX;
    // (the records of one launch: the larger of the Matern and the SExp form's -- the call's kind is not known here)
    int64_t recs = pair_chunk(nb, Mc, Dw, jsep_tch(), matern_rec_doubles(nb, Dw)) * matern_rec_doubles(nb, Dw);
    {
        const int64_t r2 = pair_chunk(nb, Mc, Dw, sexp_tch2(nb), sexp_rec_doubles(nb, Dw)) * sexp_rec_doubles(nb, Dw);
        if (r2 > recs) recs = r2;
    }
    return (size_t)(nb * (nb + 1) / 2 * Mc + recs) * sizeof(double);
}

static int linkgp_run(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int Dw, int Dz, const double *m, const double *v,
                      const double *z, const double *Wtr, const double *Wg, const double *length_h, int nlen,
                      const double *Rinv, int64_t ldr, const double *ry, const int32_t *drop, double scale,
                      double nugget, double *mean, double *var, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || M <= 0 || !m || !v || !Wtr || !length_h || !Rinv || !ry || !mean || !var || !work)
        BAD_ARG(ctx, "null pointer or empty problem");
    if (kind != DGPAMD_SEXP && kind != DGPAMD_MATERN25) BAD_ARG(ctx, "kind must be 0 or 1");
    if (Dw <= 0 || Dz < 0 || Dw + Dz > DGPAMD_MAXD) BAD_ARG(ctx, "bad Dw / Dz");
    if (Dz > 0 && (!z || !Wg)) BAD_ARG(ctx, "Dz > 0 needs z and Wg");
    if (nlen != 1 && nlen != Dw + Dz) BAD_ARG(ctx, "nlen must be 1 or Dw+Dz");
    if (ldr < n) BAD_ARG(ctx, "ldr < n");
    LinkArgs a;
    a.kind = kind; a.Dw = Dw; a.Dz = Dz; a.n = n; a.M = M; a.m = m; a.v = v; a.z = z; a.W = Wtr; a.Wg = Wg;
    for (int d = 0; d < Dw + Dz; ++d) a.len[d] = length_h[nlen == 1 ? 0 : d];   // functions.py:402-410 broadcast
    a.Rinv = Rinv; a.ldr = ldr; a.ry = ry; a.scale = scale; a.nugget = nugget; a.mean = mean; a.var = var;
    a.partial = (double *)work;
    a.drop = drop;
    a.no_order_classes = 0;
    a.dbg = nullptr;
    a.tch = jsep_tch();
    int64_t Mc = ((M + TCH - 1) / TCH) * TCH;
    if (Mc > MC_MAX) Mc = MC_MAX;
    const int nb = (int)((n + 63) / 64), ntiles = nb * (nb + 1) / 2;
    const bool direct = ctx->linkgp_direct || drop;   // the leave-one-out weights change per test point: direct kernels
    const bool sep = (kind == DGPAMD_MATERN25) && !direct;
    a.recs = a.partial + (int64_t)ntiles * Mc;
    a.npad = (int64_t)nb * 64;
    const bool sx2 = (kind == DGPAMD_SEXP) && !direct && Dw + 2 <= 16;
    if (sep) Mc = pair_chunk(nb, Mc, Dw, a.tch, matern_rec_doubles(nb, Dw));   // records of one chunk: Mc*Dw*npad*240 B (Matern), Mc*npad*(Dw+3..6)*8 B (SExp)
    if (sx2) {
        a.tch = sexp_tch2(nb);
        Mc = pair_chunk(nb, Mc, Dw, a.tch, sexp_rec_doubles(nb, Dw));
    }
    if (getenv("DGPAMD_PAIR_CHUNK") && (sep || sx2)) {   // (comparison runs: the fixed 256 points per launch of the earlier builds)
        const int64_t c = atoll(getenv("DGPAMD_PAIR_CHUNK"));
        if (c >= TCH && c < Mc) Mc = c / TCH * TCH;
    }
    a.Mc = Mc;
    a.gfac = sx2 ? a.recs + Mc * a.npad * (int64_t)((Dw + 2 + 3) & ~3)   // (SExp second form: [Mc][npad][KPA] records, then the ss values)
                 : a.recs + Mc * (int64_t)Dw * a.npad * REC;
    const int DT = Dw + Dz;
    size_t shm = ((size_t)2 * DT * 64 + (size_t)TCH * (2 * Dw + Dz) + TCH * 4) * sizeof(double);
    if (kind == DGPAMD_MATERN25) shm += 64 * 65 * sizeof(double);
    if (drop) {
        shm += ((size_t)2 * TCH * 64 + 128 + 2 * TCH) * sizeof(double);
        if (kind == DGPAMD_SEXP)
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)linkgp_J_kernel<DGPAMD_SEXP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        else
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)linkgp_J_kernel<DGPAMD_MATERN25, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    }
    for (int64_t t0 = 0; t0 < M; t0 += Mc) {
        a.t0 = t0;
        const int64_t mc = M - t0 < Mc ? M - t0 : Mc;
        const unsigned tb = (unsigned)((mc + TCH - 1) / TCH);
        if (kind == DGPAMD_SEXP) {
            hipLaunchKernelGGL(linkgp_mean_kernel<DGPAMD_SEXP>, dim3((unsigned)mc), dim3(256), 0, ctx->stream, a);
            if (drop) {
                hipLaunchKernelGGL((linkgp_J_kernel<DGPAMD_SEXP, true>), dim3(ntiles, tb), dim3(256), shm, ctx->stream, a);
            } else if (direct) {
                hipLaunchKernelGGL((linkgp_J_kernel<DGPAMD_SEXP, false>), dim3(ntiles, tb), dim3(256), shm, ctx->stream, a);
            } else if (Dw + 2 <= 4 * SX_KS && !getenv("DGPAMD_SEXP_FORM1")) {
                const int KPA = (Dw + 2 + 3) & ~3;
                const size_t shm2 = ((size_t)2 * Dw * 64 + a.tch * 4 + EXPN_TAB) * sizeof(double);
                const bool poly = getenv("DGPAMD_SEXP_POLY") != nullptr;   // (comparison run: the table-free exponential)
                const unsigned tb2 = (unsigned)((mc + a.tch - 1) / a.tch);
                hipLaunchKernelGGL(sexp_records_kernel, dim3((unsigned)((a.npad + 255) / 256), (unsigned)mc), dim3(256), 0, ctx->stream, a, KPA);
                PROF_BEGIN(ctx, PROF_LINKGP_J, (double)mc * (double)n * (double)(n + 1) * 0.5);   // pair evaluations (one exponential each)
#define JSEXP2(KS_) \
    if (poly) hipLaunchKernelGGL((linkgp_Jsexp2_kernel<KS_, false>), dim3(ntiles, tb2), dim3(256), shm2, ctx->stream, a); \
    else hipLaunchKernelGGL((linkgp_Jsexp2_kernel<KS_, true>), dim3(ntiles, tb2), dim3(256), shm2, ctx->stream, a)
                switch (KPA / 4) {
                    case 1: JSEXP2(1); break;
                    case 2: JSEXP2(2); break;
                    case 3: JSEXP2(3); break;
                    default: JSEXP2(4); break;
                }
#undef JSEXP2
                PROF_END(ctx, PROF_LINKGP_J);
            } else {
                const int KP = (Dw + 3) & ~3, LDU = KP + 2;
                const size_t shm_s = ((size_t)2 * DT * 64 + (size_t)KP * LDK + (size_t)TCH * (2 * Dw + Dz) + TCH * 4 + 2 * 64 * LDU + 4 * 64 + Dz) * sizeof(double);
                if (shm_s > 48 * 1024)
                    HIP_TRY(ctx, hipFuncSetAttribute((const void *)linkgp_Jsexp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_s));
                hipLaunchKernelGGL(linkgp_Jsexp_kernel, dim3(ntiles, tb), dim3(256), shm_s, ctx->stream, a);
            }
            hipLaunchKernelGGL(linkgp_finalize_kernel<DGPAMD_SEXP>, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, ctx->stream, a, ntiles);
        } else {
            hipLaunchKernelGGL(linkgp_mean_kernel<DGPAMD_MATERN25>, dim3((unsigned)mc), dim3(256), 0, ctx->stream, a);
            if (drop) {
                hipLaunchKernelGGL((linkgp_J_kernel<DGPAMD_MATERN25, true>), dim3(ntiles, tb), dim3(256), shm, ctx->stream, a);
            } else if (direct) {
                hipLaunchKernelGGL((linkgp_J_kernel<DGPAMD_MATERN25, false>), dim3(ntiles, tb), dim3(256), shm, ctx->stream, a);
            } else {
                const int pipe = getenv("DGPAMD_JSEP_PIPE") ? atoi(getenv("DGPAMD_JSEP_PIPE")) : 2;
                const size_t shm_sep = ((size_t)2 * DT * 64 + (size_t)a.tch * Dz + a.tch * 4 + 2 * 128 * PST) * sizeof(double) + (size_t)Dw * 4 * sizeof(int) +
                                       (getenv("DGPAMD_JSEP_LOG") ? 16 + JSEP_LOG_STEPS * 8 * sizeof(long long) : 0);
                a.no_order_classes = (getenv("DGPAMD_JSEP_NOCLASS") ? 1 : 0) | (getenv("DGPAMD_JSEP_SAMEBLK") ? 2 : 0) | (getenv("DGPAMD_JSEP_DIAG") ? atoi(getenv("DGPAMD_JSEP_DIAG")) : 0);
                const bool plog = ctx->tlog != nullptr && getenv("DGPAMD_JSEP_LOG") != nullptr;
                auto jsep = plog ? linkgp_Jsep_kernel<2, true> : (pipe == 0 ? linkgp_Jsep_kernel<0, false> : linkgp_Jsep_kernel<2, false>);
                a.dbg = (plog && ctx->tlog_words >= 64 + (long long)ntiles * ((mc + a.tch - 1) / a.tch) * JSEP_LOG_WORDS) ? ctx->tlog : nullptr;
                if (shm_sep > 48 * 1024)
                    HIP_TRY(ctx, hipFuncSetAttribute((const void *)jsep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_sep));
                hipLaunchKernelGGL(matern_records_kernel, dim3((unsigned)((a.npad + 127) / 128), Dw, (unsigned)mc), dim3(256), 0, ctx->stream, a);
                if (Dz)
                    hipLaunchKernelGGL(global_factor_kernel, dim3((unsigned)((a.npad + 255) / 256), (unsigned)mc), dim3(256), 0, ctx->stream, a);
                PROF_BEGIN(ctx, PROF_LINKGP_J, (double)mc * (double)n * (double)n * 0.5 * Dw * 30.0 * 2.0);
                hipLaunchKernelGGL(jsep, dim3(ntiles, (unsigned)((mc + a.tch - 1) / a.tch)), dim3(256), shm_sep, ctx->stream, a);
                PROF_END(ctx, PROF_LINKGP_J);
            }
            hipLaunchKernelGGL(linkgp_finalize_kernel<DGPAMD_MATERN25>, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, ctx->stream, a, ntiles);
        }
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_linkgp_predict(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int Dw, int Dz, const double *m,
                                     const double *v, const double *z, const double *Wtr, const double *Wg,
                                     const double *length_h, int nlen, const double *Rinv, int64_t ldr,
                                     const double *ry, double scale, double nugget, double *mean, double *var,
                                     void *work) {
    return linkgp_run(ctx, kind, n, M, Dw, Dz, m, v, z, Wtr, Wg, length_h, nlen, Rinv, ldr, ry, nullptr, scale, nugget,
                      mean, var, work);
}

extern "C" int dgpamd_linkgp_loo(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int Dw, int Dz, const double *m,
                                 const double *v, const double *z, const double *Wtr, const double *Wg,
                                 const double *length_h, int nlen, const double *Rinv, int64_t ldr, const double *ry,
                                 const int32_t *drop, double scale, double nugget, double *mean, double *var,
                                 void *work) {
    if (ctx && !drop) BAD_ARG(ctx, "drop is null");
    return linkgp_run(ctx, kind, n, M, Dw, Dz, m, v, z, Wtr, Wg, length_h, nlen, Rinv, ldr, ry, drop, scale, nugget, mean,
                      var, work);
}

// ---------------------------------------------------------------------------
// a15  moments over imputations (emulation.py:846-847)
// ---------------------------------------------------------------------------
__global__ void moments_acc_kernel(int64_t count, const double *mu, const double *var, double *s1, double *s2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        double m = mu[i];
        s1[i] += m;
        s2[i] += m * m + var[i];
    }
}
__global__ void moments_fin_kernel(int64_t count, double S, double *s1, double *s2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        double m = s1[i] / S;
        s1[i] = m;
        s2[i] = s2[i] / S - m * m;
    }
}

extern "C" int dgpamd_moments_accumulate(dgpamd_ctx *ctx, int64_t count, const double *mu, const double *var,
                                         double *sum_mu, double *sum_m2) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (count <= 0 || !mu || !var || !sum_mu || !sum_m2) BAD_ARG(ctx, "null pointer or empty");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moments_acc_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, count, mu, var, sum_mu, sum_m2);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_moments_finalize(dgpamd_ctx *ctx, int64_t count, double S, double *sum_mu_to_mu,
                                       double *sum_m2_to_var) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (count <= 0 || !(S > 0.0) || !sum_mu_to_mu || !sum_m2_to_var) BAD_ARG(ctx, "bad arguments");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moments_fin_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, count, S, sum_mu_to_mu, sum_m2_to_var);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}
