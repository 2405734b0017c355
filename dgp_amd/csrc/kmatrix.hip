// Kernel-matrix assembly: K_ij = c(x_i, x_j), diag = 1 + nugget*W_i.
// Replaces kernel.k_matrix (kernel_class.py:304-359): scipy pdist/squareform for
// 'sexp', numba pdist_matern_coef (functions.py:16-34) for 'matern2.5'.
//
// One 256-thread workgroup per 64x64 LOWER tile (bi >= bj); the two 64-row slabs
// of X are gathered ([Xloc[:,colmap] | Xglob]), scaled by 1/lengthscale and
// staged transposed in LDS; each thread owns a 4x4 micro-tile -- rows ty+16a,
// column PAIRS 2tx, 2tx+1 and 32+2tx, 33+2tx -- so that every store is a 16-byte
// global_store_dwordx4 and a wave's store instruction writes four 256-byte row
// segments (round 3: 8-byte stores, four 128-byte lines per instruction).
// Tiles below the diagonal come first in the grid, the diagonal tiles (which
// write no mirrored tile in symmetric mode) last: the tail of the launch is its
// cheapest work.  The kernel is f64-VALU + HBM-write bound.
#include "common.hpp"
#include <algorithm>

#ifndef KM_WGS
#define KM_WGS 8   // workgroups per CU the kernels are compiled for (registers <= 64)
#endif


// Round 5: the tile is computed and stored in TWO HALVES (rows ty, ty + 16, then rows ty + 32, ty + 48 of every thread): 8
// entries per thread at a time instead of 16 halves the registers (108 -> <= 64: eight workgroups per CU instead of four), and a
// workgroup's first stores are on their way while its second half is still being computed.  Why occupancy: a workgroup lives
// ~12 us (inputs 2-5 us behind the CU's queued stores, arithmetic 2.5 us, its own stores' drain), so four per CU deliver a tile
// per 3 us per CU -- exactly what the store stream takes (3.1 us per tile and CU at the 0.68 of HBM that tile-shaped stores reach,
// profiles/r05_kmatrix_store_shapes.txt): any hiccup showed, and at n = 5000 (three rounds of workgroups that start together)
// the arithmetic and the store phases of the whole launch did not overlap at all (26 us + 36 us = the 57 us measured).
template <int KIND>
__device__ __forceinline__ void kmatrix_body(const KmatArgs &a, const int b) {
    extern __shared__ double lds[];
    const int D = a.kp.Dl + a.kp.Dg;
    double *XiT = lds;            // [D][64]
    double *XjT = lds + D * 64;   // [D][64]
    double *TT2 = lds + 2 * D * 64;   // [16][65]: the mirror's transposition buffer (symmetric mode)
#if KM_EXP_TAB
    double *etab = TT2 + (a.full ? 16 * 65 : 0);   // [EXPN_TAB]: 2^(j / EXPN_TAB), exp_negated_tab's table
    for (int j = threadIdx.x; j < EXPN_TAB; j += 256) etab[j] = exp2((double)j * (1.0 / EXPN_TAB));   // (visible behind the staging barrier below)
#endif
    int bi, bj;
    {   // strictly-lower tiles first (row-major in the triangle), then the diagonal
        const int nb = a.nbk, nlow = nb * (nb - 1) / 2, t = blockIdx.x;
        if (t < nlow) {
            tri_decode(t, bi, bj);   // t = b(b+1)/2 + j with j <= b: the strictly-lower tile (b + 1, j)
            ++bi;
        } else {
            bi = bj = t - nlow;
        }
    }
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64;
    const double *Xl = a.Xloc + (int64_t)b * a.stride_loc;

    {
        const float invD = 1.0f / (float)D;
        const bool small = 64 * D < (1 << 20);   // (idx / D through the float reciprocal: exact for these sizes)
        for (int idx = tid; idx < 64 * D; idx += 256) {
            int row = small ? (int)(((float)idx + 0.5f) * invD) : idx / D;
            int d = idx - row * D;
            int64_t gi = i0 + row, gj = j0 + row;
            double vi = 0.0, vj = 0.0;
            if (d < a.kp.Dl) {
                int c = a.kp.colmap[d];
                if (gi < a.n) vi = Xl[gi * a.ldloc + c];
                if (gj < a.n) vj = Xl[gj * a.ldloc + c];
            } else {
                int c = d - a.kp.Dl;
                if (gi < a.n) vi = a.Xglob[gi * a.kp.Dg + c];
                if (gj < a.n) vj = a.Xglob[gj * a.kp.Dg + c];
            }
            double il = a.kp.inv_len[d];
            XiT[d * 64 + row] = vi * il;
            XjT[d * 64 + row] = vj * il;
        }
    }
    __syncthreads();

    double *Kb = a.K + (int64_t)b * a.stride_k;
    const double *Yb = a.Y ? a.Y + (int64_t)b * a.stride_y : nullptr;
    const bool interior = i0 + 64 <= a.n;   // j0 <= i0: the whole tile lies inside the n x n correlation block
    const bool even_ld = (a.ldk & 1) == 0;  // (16-byte stores need 16-byte addresses: base pointers are, rows are when ldk is even)
    const int pc = 2 * (tx & 7), pr0 = 2 * ty + (tx >> 3);   // mirror: column pair / row of this lane inside a 64 x 16 block (rows pr0, pr0 + 32)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // micro-tile of this half: rows ty + 16 (2 h + p), p = 0, 1; columns col(q) = 32 (q >> 1) + 2 tx + (q & 1)
        double s[2][4], pr[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[p][q] = 0.0;
                pr[p][q] = 1.0;
            }
        for (int d = 0; d < D; ++d) {
            double xi[2], xj[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) xi[p] = XiT[d * 64 + ty + 16 * (2 * h + p)];
            {
                const double2 lo = *reinterpret_cast<const double2 *>(XjT + d * 64 + 2 * tx);
                const double2 hi = *reinterpret_cast<const double2 *>(XjT + d * 64 + 32 + 2 * tx);
                xj[0] = lo.x; xj[1] = lo.y; xj[2] = hi.x; xj[3] = hi.y;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double df = xi[p] - xj[q];
                    if (KIND == DGPAMD_SEXP)
                        corr_accum_sexp(df, s[p][q]);
                    else
                        corr_accum_matern(df, pr[p][q], s[p][q]);
                }
        }
        double v[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                // Round 6: the table exponential of the pair kernels (common.hpp: nine f64 + five 32-bit instructions and an LDS read; <= 3.5e-16
                // relative on [0, 700]) instead of the library's exp (22 instructions, a third of this kernel's arithmetic) -- the kernel is bound by its
                // f64 VALU issue, not by its stores.  The gradient reductions that recompute these entries (csrc/train.hip) were
                // switched in the same commit: the objective's K and the gradient's dK hold the same bits.  -DKM_EXP_TAB=0: rounds 1-5.
#if KM_EXP_TAB
                v[p][q] = (KIND == DGPAMD_SEXP) ? exp_negated_tab(s[p][q], etab) : pr[p][q] * exp_negated_tab(SQRT5 * s[p][q], etab);
#else
                v[p][q] = (KIND == DGPAMD_SEXP) ? exp(-s[p][q]) : pr[p][q] * exp(-SQRT5 * s[p][q]);
#endif
        if (bi == bj) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ty + 16 * (2 * h + p) == 32 * (q >> 1) + 2 * tx + (q & 1)) {
                        const int64_t gi = i0 + ty + 16 * (2 * h + p);
                        v[p][q] = 1.0 + a.kp.nugget * (a.W ? a.W[gi < a.n ? gi : 0] : 1.0);
                    }
        }
        if (interior && even_ld) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                double *row = Kb + (i0 + ty + 16 * (2 * h + p)) * a.ldk + j0 + 2 * tx;
                *reinterpret_cast<double2 *>(row) = make_double2(v[p][0], v[p][1]);
                *reinterpret_cast<double2 *>(row + 32) = make_double2(v[p][2], v[p][3]);
            }
        } else if (a.full) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int64_t gi = i0 + ty + 16 * (2 * h + p);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t gj = j0 + 32 * (q >> 1) + 2 * tx + (q & 1);
                    if (gi < a.n && gj < a.n) Kb[gi * a.ldk + gj] = v[p][q];
                }
            }
        } else {
            // augmented factorisation buffer: rows >= n carry right-hand sides, corner zero (ldk = padded dimension: even)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int64_t gi = i0 + ty + 16 * (2 * h + p);
                double w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t gj = j0 + 32 * (q >> 1) + 2 * tx + (q & 1);
                    double val = v[p][q];
                    if (gi >= a.n) {
                        int64_t qy = gi - a.n;
                        val = (gj < a.n && qy < a.r) ? Yb[qy * a.ldy + gj] : 0.0;
                    } else if (gj >= a.n) {
                        val = 0.0;
                    }
                    w[q] = val;
                }
                double *row = Kb + gi * a.ldk + j0 + 2 * tx;
                *reinterpret_cast<double2 *>(row) = make_double2(w[0], w[1]);
                *reinterpret_cast<double2 *>(row + 32) = make_double2(w[2], w[3]);
            }
        }
        if (a.full && bi != bj) {
            // mirrored tile K[j][i]: transposed through LDS so that these stores are row-contiguous too (naive transposed
            // stores write 32-byte fragments and cost 40% of the kernel's bandwidth), in passes of 16 rows through a 16 x 65
            // buffer of its own (the staged inputs are still being read by the second half).  Per pass the mirrored block is
            // 64 rows x 16 columns: eight lanes cover a row's 128-byte line with 16-byte stores.
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c = 2 * h + p;
                __syncthreads();   // (the previous pass has been read)
#pragma unroll
                for (int q = 0; q < 4; ++q) TT2[ty * 65 + 32 * (q >> 1) + 2 * tx + (q & 1)] = v[p][q];   // rows i0 + 16 c + ty of the tile, all 64 columns
                __syncthreads();
                // mirrored block: rows j0 + r (r = 0..63), columns i0 + 16 c + cc (cc = 0..15)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int r = pr0 + 32 * k;
                    const int64_t gc = i0 + 16 * c + pc;
                    const double m0 = TT2[pc * 65 + r], m1 = TT2[(pc + 1) * 65 + r];
                    double *dst = Kb + (j0 + r) * a.ldk + gc;
                    if (interior && even_ld) {
                        *reinterpret_cast<double2 *>(dst) = make_double2(m0, m1);
                    } else {
                        if (interior || gc < a.n) dst[0] = m0;
                        if (interior || gc + 1 < a.n) dst[1] = m1;
                    }
                }
            }
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256, KM_WGS) void kmatrix_kernel(KmatArgs a) {
    if (a.pred && *a.pred) return;   // (a speculative batch that an earlier one has made unnecessary: dgpamd_ess_queue)
    if (a.zero_ptr && blockIdx.z == 0)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.zero_words; i += gridDim.x * 256) a.zero_ptr[i] = 0;
    kmatrix_body<KIND>(a, blockIdx.z);
}

// Several nodes in one launch (grid.z = node): every node brings its own inputs, kernel family and hyper-parameters,
// read from an argument array in device memory (uniform addresses: scalar loads).
__global__ __launch_bounds__(256, 6) void kmatrix_multi_kernel(const KmatArgs *args) {   // (both kernel families inlined: 74 registers)
    const KmatArgs &a = args[blockIdx.z];
    if (a.zero_ptr)   // (set in one node's arguments only)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.zero_words; i += gridDim.x * 256) a.zero_ptr[i] = 0;
    if (a.kp.kind == DGPAMD_SEXP)
        kmatrix_body<DGPAMD_SEXP>(a, 0);
    else
        kmatrix_body<DGPAMD_MATERN25>(a, 0);
}

// The same with the nodes' arguments BY VALUE (up to three nodes: 2.8 KB of kernel arguments): no argument array in device memory, no host-to-device
// copy in front of the launch -- a blit kernel and its launch gap per round of the lock-step M-step, most of whose rounds hold one to three nodes.
struct KmatArgs3 {
    KmatArgs a[3];
};
static_assert(sizeof(KmatArgs3) <= 3800, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256, 6) void kmatrix_multi_val_kernel(KmatArgs3 v) {
    const KmatArgs &a = v.a[blockIdx.z];
    if (a.zero_ptr)   // (set in one node's arguments only)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.zero_words; i += gridDim.x * 256) a.zero_ptr[i] = 0;
    if (a.kp.kind == DGPAMD_SEXP)
        kmatrix_body<DGPAMD_SEXP>(a, 0);
    else
        kmatrix_body<DGPAMD_MATERN25>(a, 0);
}

int launch_kmatrix_multi(dgpamd_ctx *ctx, const KmatArgs *dev_args, const KmatArgs *host_args, int count) {
    int Dmax = 0, full = 0;
    for (int c = 0; c < count; ++c) {
        Dmax = std::max(Dmax, host_args[c].kp.Dl + host_args[c].kp.Dg);
        full |= host_args[c].full;
    }
    int64_t rows = host_args[0].full ? host_args[0].n : padded_dim(host_args[0].n);
    int nbk = (int)((rows + 63) / 64);
    size_t shm = ((size_t)2 * Dmax * 64 + (full ? (size_t)16 * 65 : (size_t)0) + KM_EXP_TAB * EXPN_TAB) * sizeof(double);
    // (D >= 44 in symmetric mode: more dynamic LDS than a launch gets by default; gfx950 has 160 KB per CU)
    if (shm > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)kmatrix_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    if (!dev_args) {   // by value (count <= 3)
        if (count > 3) BAD_ARG(ctx, "by-value launch of more than three nodes");
        KmatArgs3 v;
        for (int c = 0; c < 3; ++c) v.a[c] = host_args[c < count ? c : 0];
        if (shm > 48 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)kmatrix_multi_val_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(kmatrix_multi_val_kernel, dim3(nbk * (nbk + 1) / 2, 1, count), dim3(256), shm, ctx->stream, v);
        LAUNCH_CHECK(ctx);
        return DGPAMD_OK;
    }
    hipLaunchKernelGGL(kmatrix_multi_kernel, dim3(nbk * (nbk + 1) / 2, 1, count), dim3(256), shm, ctx->stream, dev_args);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

int launch_kmatrix(dgpamd_ctx *ctx, const KmatArgs &a_, int batch) {
    KmatArgs a = a_;
    a.pred = ctx->pred;
    const int D = a.kp.Dl + a.kp.Dg;
    int64_t rows = a.full ? a.n : padded_dim(a.n);
    int nbk = (int)((rows + 63) / 64);
    int ntiles = nbk * (nbk + 1) / 2;
    size_t shm = ((size_t)2 * D * 64 + (a.full ? (size_t)16 * 65 : (size_t)0) + KM_EXP_TAB * EXPN_TAB) * sizeof(double);
    a.nbk = nbk;
    dim3 grid(ntiles, 1, batch);
    if (shm > 48 * 1024) {   // (D up to DGPAMD_MAXD = 64: 65.5 KB + the mirror's 8.3 KB -- beyond a launch's default dynamic LDS)
        if (a.kp.kind == DGPAMD_SEXP)
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)kmatrix_kernel<DGPAMD_SEXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        else
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)kmatrix_kernel<DGPAMD_MATERN25>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    }
    // algorithmic bytes: the stored triangle(s) of K (8 n^2, or 4 n^2 for the lower tiles) + X once
    PROF_BEGIN(ctx, PROF_KMATRIX, (double)batch * ((a.full ? 8.0 : 4.0) * (double)rows * (double)rows + 8.0 * (double)a.n * D));
    if (a.kp.kind == DGPAMD_SEXP)
        hipLaunchKernelGGL(kmatrix_kernel<DGPAMD_SEXP>, grid, dim3(256), shm, ctx->stream, a);
    else
        hipLaunchKernelGGL(kmatrix_kernel<DGPAMD_MATERN25>, grid, dim3(256), shm, ctx->stream, a);
    PROF_END(ctx, PROF_KMATRIX);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

int build_kmat_args(dgpamd_ctx *ctx, KmatArgs &a, int kind, int64_t n, const double *Xloc, int64_t ldloc,
                    int64_t stride_loc, const int32_t *colmap_h, int Dl, const double *Xglob, int Dg,
                    const double *length_h, int nlen, double nugget, const double *W, double *K, int64_t ldk,
                    int64_t stride_k, int full, const double *Y, int64_t ldy, int64_t stride_y, int r, int batch) {
    if (n <= 0) BAD_ARG(ctx, "n must be positive");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    if (Dl < 0 || Dg < 0) BAD_ARG(ctx, "negative dimension");
    if ((Dl > 0 && !Xloc) || (Dg > 0 && !Xglob) || !K || !length_h) BAD_ARG(ctx, "null pointer");
    if (r < 0 || (r > 0 && !Y)) BAD_ARG(ctx, "r > 0 needs Y");
    if (!full) {
        if (ldk != padded_dim(n)) BAD_ARG(ctx, "augmented buffer needs ldk == dgpamd_padded_dim(n)");
        if (n + r > ldk) BAD_ARG(ctx, "too many right-hand sides for the padded buffer");
    } else if (ldk < n) {
        BAD_ARG(ctx, "ldk < n");
    }
    int rc = fill_kern_params(ctx, a.kp, kind, colmap_h, Dl, Dg, length_h, nlen, nugget);
    if (rc) return rc;
    a.pred = nullptr;
    a.zero_ptr = nullptr;
    a.zero_words = 0;
    a.n = n;
    a.Xloc = Xloc;
    a.ldloc = ldloc;
    a.stride_loc = stride_loc;
    a.Xglob = Xglob;
    a.W = W;
    a.K = K;
    a.ldk = ldk;
    a.stride_k = stride_k;
    a.full = full;
    a.Y = Y;
    a.ldy = ldy;
    a.stride_y = stride_y;
    a.r = r;
    a.nbk = (int)(((full ? n : padded_dim(n)) + 63) / 64);
    return DGPAMD_OK;
}

extern "C" int dgpamd_kmatrix(dgpamd_ctx *ctx, int kind, int64_t n, const double *Xloc, int64_t ldloc,
                              int64_t stride_loc, const int32_t *colmap_h, int Dl, const double *Xglob, int Dg,
                              const double *length_h, int nlen, double nugget, const double *W, double *K,
                              int64_t ldk, int64_t stride_k, int full, const double *Y, int64_t ldy,
                              int64_t stride_y, int r, int batch) {
    if (!ctx) return DGPAMD_BAD_ARG;
    KmatArgs a;
    int rc = build_kmat_args(ctx, a, kind, n, Xloc, ldloc, stride_loc, colmap_h, Dl, Xglob, Dg, length_h, nlen, nugget,
                             W, K, ldk, stride_k, full, Y, ldy, stride_y, r, batch);
    if (rc) return rc;
    return launch_kmatrix(ctx, a, batch);
}
