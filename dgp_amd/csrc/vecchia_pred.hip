// Register-resident Vecchia prediction kernels (a22 gp_vecch vecchia.py:635-654, a23 link_gp_vecch :758-796 with IJ_nb
// :838-907).  One wave per test point, one row of the conditioning block per lane, rows broadcast entry by entry with
// v_readlane; every register index is a compile-time constant (static_for), which is why this file takes minutes to compile.
#include "vecchia_pred.hpp"
#include "linkfun.hpp"
#include <utility>

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

// The same prediction with the block in REGISTERS (pm <= VG_BC, D <= 16): one wave per test point, one ROW of the
// (b + 2) x (b + 1) block [neighbours ; test point ; y] per lane -- lane r holds row r's b neighbour columns in reg[] and the
// test point's column in `last`.  A column of the block is built by broadcasting point c's scaled coordinates from lane c
// (v_readlane: an SGPR operand for every lane) and evaluating the correlation in all lanes at once; pivot j's elimination
// broadcasts row j entry by entry the same way: reg[c] -= (reg[j] / d_j) * A[j][c] for the rows below j.  No LDS, no barrier,
// no dependent LDS round trips; four test points per workgroup.  (LDL^T without square roots: the Schur complement of the test
// point and the eliminated y row are the same numbers as with gp_vecch's Cholesky, vecchia.py:635-654.)  The LDS kernel above
// ran one wave per point at 3-6 workgroups per CU on chains of dependent LDS reads: 8.3 ms per 100 000 points at pm = 50, D = 8.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    return fma(x, e, x);
}
template <int KIND, int DM>
__global__ __launch_bounds__(256) void vecchia_gp_reg_kernel(VGpArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.M) return;   // (wave-uniform)
    const int pm = a.pm, D = a.vp.D;
    const int64_t nnv = lane < pm ? a.NN[t * pm + lane] : -1;
    const int b = __builtin_amdgcn_readfirstlane(__popcll(__ballot(nnv >= 0)));   // the valid neighbours come first
    const int bb = b + 1;
    double xr[DM];
    {
        const double *src = lane < b ? a.w + nnv * D : a.x + t * D;
#pragma unroll
        for (int d = 0; d < DM; ++d) xr[d] = (d < D && lane <= b) ? src[d] * a.vp.inv_len[d] : 0.0;
    }
    const double yv = lane < b ? a.y[nnv] : 0.0;
    const double dg = 1.0 + a.vp.nugget * (lane < b ? a.nugget_diag[nnv] : 1.0);
    auto column = [&](int c) {   // column c of the block for every row at once
        double s = 0.0, pr = 1.0;
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            const double df = xr[d] - readlane_f64(xr[d], c);
            if (KIND == DGPAMD_SEXP)
                corr_accum_sexp(df, s);
            else
                corr_accum_matern(df, pr, s);
        }
        double v = (KIND == DGPAMD_SEXP) ? exp_negated(s) : pr * exp_negated(SQRT5 * s);
        v = lane == c ? dg : v;
        return lane == bb ? readlane_f64(yv, c) : v;   // (the y row; yv is 0 in the test point's lane)
    };
    // (static_for: every reg[] index is a compile-time constant -- left to `#pragma unroll` the compiler keeps the loops and the
    //  array goes to scratch)
    double reg[VG_BC];
    static_for<0, VG_BC>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        reg[c] = 0.0;
        if (c < b) reg[c] = column(c);
    });
    double last = column(b);
    // elimination of the b neighbour columns (columns beyond b hold zeros and stay zero: skipped in groups of eight)
    static_for<0, VG_BC>([&](auto ij) {
        constexpr int j = decltype(ij)::value;
        if (j < b) {
            const double rd = rcp_newton(readlane_f64(reg[j], j));
            const double mi = lane > j ? reg[j] * rd : 0.0;
            static_for<(j + 1) / 8, (VG_BC + 7) / 8>([&](auto ig) {
                constexpr int c0 = 8 * decltype(ig)::value;
                if (c0 < b) {
                    static_for<0, 8>([&](auto iq) {
                        constexpr int c = c0 + decltype(iq)::value;
                        if constexpr (c > j && c < VG_BC) reg[c] = fma(-mi, readlane_f64(reg[c], j), reg[c]);
                    });
                }
            });
            last = fma(-mi, readlane_f64(last, j), last);
        }
    });
    const double var = readlane_f64(last, b), mean = readlane_f64(last, bb);
    if (lane == 0) {
        a.mean[t] = -mean;
        a.var[t] = a.scale * var;
    }
}

template <int KIND>
static void launch_vgp_reg(dgpamd_ctx *ctx, const VGpArgs &a) {
    const unsigned grid = (unsigned)((a.M + 3) / 4);
    if (a.vp.D <= 8)
        hipLaunchKernelGGL((vecchia_gp_reg_kernel<KIND, 8>), dim3(grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((vecchia_gp_reg_kernel<KIND, 16>), dim3(grid), dim3(256), 0, ctx->stream, a);
}
void launch_vecchia_gp_reg(dgpamd_ctx *ctx, const VGpArgs &a) {
    if (a.vp.kind == DGPAMD_SEXP)
        launch_vgp_reg<DGPAMD_SEXP>(ctx, a);
    else
        launch_vgp_reg<DGPAMD_MATERN25>(ctx, a);
}


// The second half of both register-resident link_gp kernels: Gauss-Jordan elimination of K | N with J, y and I riding along, then
// the mean and the variance (see the squared-exponential kernel below for the algebra).
__device__ __forceinline__ void linkgp_reg_finish(double (&reg)[VL_BC], double (&jm)[VL_BC], double yv, double Iv, const bool act,
                                                  const int b, const int lane, const VLinkArgs &a, const int64_t t) {
    // elimination: rows below the pivot take  row_i -= (a_ij / d_j) row_j  in K | N, in J, in y and in I
    double dmine = 1.0;
    static_for<0, VL_BC>([&](auto ij) {
        constexpr int j = decltype(ij)::value;
        if (j < b) {
            const double d = readlane_f64(reg[j], j);
            dmine = lane == j ? d : dmine;
            const double rd = rcp_newton(d);
            const double mi = (lane > j && act) ? reg[j] * rd : 0.0;
            static_for<0, (VL_BC + 7) / 8>([&](auto ig) {
                constexpr int c0 = 8 * decltype(ig)::value;
                if (c0 < b) {
                    static_for<0, 8>([&](auto iq) {
                        constexpr int c = c0 + decltype(iq)::value;
                        if constexpr (c < VL_BC) {
                            if constexpr (c != j) reg[c] = fma(-mi, readlane_f64(reg[c], j), reg[c]);
                            jm[c] = fma(-mi, readlane_f64(jm[c], j), jm[c]);
                        }
                    });
                }
            });
            reg[j] = lane > j ? -mi : reg[j];
            yv = fma(-mi, readlane_f64(yv, j), yv);
            Iv = fma(-mi, readlane_f64(Iv, j), Iv);
        }
    });
    const double rdi = act ? rcp_newton(dmine) : 0.0;
    const double v = yv * rdi;
    // t = L^-T v (column sums of N scaled by v), then the lane's share of v . (M t) and of the trace
    double mt = 0.0, trp = 0.0;
    static_for<0, VL_BC>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        if (c < b) {
            const double nic = lane > c ? reg[c] : (lane == c ? 1.0 : 0.0);
            const double tc = wsum(act ? nic * v : 0.0);
            mt = fma(jm[c], tc, mt);
            trp = fma(jm[c], nic, trp);
        }
    });
    const double mu = wsum(Iv * v), qd = wsum(v * mt), tr = wsum(trp * rdi);
    if (lane == 0) {
        a.mean[t] = mu;
        a.var[t] = fabs(qd - mu * mu + a.scale * (1.0 + a.nugget - tr));
    }
}

// link_gp_vecch for the squared-exponential kernel with everything in REGISTERS (pm <= VL_BC, Dw <= 8, Dz in {0} or <= 8):
// one wave per test point, lane r = neighbour r.  reg[c] holds row r of K and, in the columns the elimination has passed, of
// N = (unit lower factor)^-1 -- Gauss-Jordan in place: pivot j's row operation  row_i -= (a_ij / d_j) row_j  applied to the
// identity puts N[i][c] where K's eliminated entries were, with the same fma for every column (row j is broadcast entry by
// entry with v_readlane).  jm[c] holds row r of J and takes the same row operations, M = L^-1 J.  With K = L D L^T:
//   R^-1 y = L^-T v,  v = D^-1 L^-1 y              (y and I ride along as per-lane scalars)
//   mean   = (L^-1 I) . v
//   y' R^-1 J R^-1 y = v . (M t),  t = L^-T v      (column sums over the lanes: wave reductions)
//   tr(K^-1 J) = sum_i (1 / d_i) sum_c M[i][c] N[i][c]
// -- the quantities of vecchia.py:758-796 / IJ_nb :838-907 without a transposition and without LDS.  The coordinates are
// held relative to the test point and scaled (u = (w - m) / l, ug = (wg - z) / l): differences are unchanged, and the
// exponents of I, J and of the global factor become sums of squares of u, u_r + u_c and ug.
template <int DGM>
__global__ __launch_bounds__(256) void vecchia_linkgp_sexp_reg_kernel(VLinkArgs a) {
    constexpr int DLM = 8;
    const int lane = threadIdx.x & 63;
    const int64_t t = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (t >= a.M) return;
    const int pm = a.pm, Dw = a.Dw, Dz = a.Dz;
    const int64_t nnv = lane < pm ? a.NN[t * pm + lane] : -1;
    const int b = __builtin_amdgcn_readfirstlane(__popcll(__ballot(nnv >= 0)));
    const bool act = lane < b;
    // per test point (uniform): weights of the exponents
    double wk[DLM], wi[DLM], c1 = 1.0, jc = 1.0;
#pragma unroll
    for (int k = 0; k < DLM; ++k) {
        wk[k] = 0.0; wi[k] = 0.0;
        if (k < Dw) {
            const double l = a.len[k], v = a.v[t * Dw + k], l2 = l * l;
            wk[k] = l2 / (8.0 * v + 2.0 * l2);
            wi[k] = l2 / (2.0 * v + l2);
            c1 *= 1.0 + 2.0 * v / l2;
            jc *= 1.0 + 4.0 * v / l2;
        }
    }
    jc = 1.0 / sqrt(jc);
    double u[DLM], ug[DGM > 0 ? DGM : 1];
#pragma unroll
    for (int k = 0; k < DLM; ++k) u[k] = (k < Dw && act) ? (a.w1[nnv * Dw + k] - a.m[t * Dw + k]) / a.len[k] : 0.0;
#pragma unroll
    for (int g = 0; g < DGM; ++g) ug[g] = (g < Dz && act) ? (a.wg[nnv * Dz + g] - a.z[t * Dz + g]) / a.len[Dw + g] : 0.0;
    // I (times the factor of the global inputs), y, the nugget's weight
    double Iz = 1.0, Iv, yv = act ? a.y[nnv] : 0.0;
    {
        double e = 0.0, sg = 0.0;
#pragma unroll
        for (int k = 0; k < DLM; ++k) e = fma(u[k] * u[k], wi[k], e);
#pragma unroll
        for (int g = 0; g < DGM; ++g) sg = fma(ug[g], ug[g], sg);
        if (DGM > 0) Iz = exp(-sg);
        Iv = act ? exp(-e) / sqrt(c1) * Iz : 0.0;
    }
    const double dg = 1.0 + a.nugget * (act ? a.nugget_diag[nnv] : 1.0);
    double reg[VL_BC], jm[VL_BC];
    static_for<0, VL_BC>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        reg[c] = 0.0;
        jm[c] = 0.0;
        if (c < b) {
            double s = 0.0, ex = 0.0;
#pragma unroll
            for (int k = 0; k < DLM; ++k) {
                const double uc = readlane_f64(u[k], c), df = u[k] - uc, sm = u[k] + uc;
                s = fma(df, df, s);
                ex = fma(sm * sm, wk[k], ex);
            }
            ex = fma(0.5, s, ex);
#pragma unroll
            for (int g = 0; g < DGM; ++g) {
                const double df = ug[g] - readlane_f64(ug[g], c);
                s = fma(df, df, s);
            }
            const double kv = lane == c ? dg : exp_negated(s);
            const double jv = jc * exp_negated(ex) * Iz * readlane_f64(Iz, c);
            reg[c] = act ? kv : 0.0;
            jm[c] = act ? jv : 0.0;
        }
    });
    linkgp_reg_finish(reg, jm, yv, Iv, act, b, lane, a, t);
}

// link_gp_vecch for the Matern-2.5 kernel, the same register design (pm <= VL_BC, Dw <= 8, Dz in {0} or <= 8): one wave per test
// point, lane r = neighbour r.  The J factor of a pair is separable per local dimension (csrc/linkfun.hpp:
// Jd = <S(x_lo), T(x_hi)> + (f2(x_hi) - f2(x_lo)) <S'(x_lo), T'(x_hi)>, lo / hi the pair's smaller / larger coordinate): every lane
// evaluates ITS neighbour's record (S[0..11], T[0..14], f2: ~500 instructions of erf / exp) once per dimension and keeps it in
// registers; column c's record is broadcast entry by entry with v_readlane and the lane forms both orientations' sums (30
// multiply-adds) and selects by the coordinates -- the same multiply-adds in the same order as the LDS kernel's pair loop
// (vecchia_linkgp_kernel, csrc/vecchia.hip), so J holds the same bits.  Only the dimension loop is a run-time loop (the
// coordinate of the dimension is re-read from memory: a run-time index into a register array would put it in scratch); K is
// built from scaled coordinates held in registers, then the elimination of the squared-exponential kernel takes over.
// The LDS kernel spent 68 ms per 100 000 points (pm = 50, Dw = Dz = 8) on dependent LDS round trips in its two
// column-parallel substitutions and its pair loops.
template <int DGM>
__global__ __launch_bounds__(256) void vecchia_linkgp_matern_reg_kernel(VLinkArgs a) {
    constexpr int DLM = 8;
    const int lane = threadIdx.x & 63;
    const int64_t t = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (t >= a.M) return;
    const int pm = a.pm, Dw = a.Dw, Dz = a.Dz;
    const int64_t nnv = lane < pm ? a.NN[t * pm + lane] : -1;
    const int b = __builtin_amdgcn_readfirstlane(__popcll(__ballot(nnv >= 0)));
    const bool act = lane < b;
    const int64_t row = act ? nnv : 0;
    // global inputs: the factor against the test point's z (vecchia.py:771-776), then the scaled coordinates for K
    double ug[DGM > 0 ? DGM : 1], Iz = 1.0;
    {
        double pr = 1.0, sg = 0.0;
#pragma unroll
        for (int g = 0; g < DGM; ++g) {
            ug[g] = 0.0;
            if (g < Dz) {
                const double x = a.wg[row * Dz + g];
                corr_accum_matern((x - a.z[t * Dz + g]) / a.len[Dw + g], pr, sg);
                ug[g] = act ? x / a.len[Dw + g] : 0.0;
            }
        }
        if (DGM > 0) Iz = pr * exp(-SQRT5 * sg);
    }
    double uw[DLM];
#pragma unroll
    for (int k = 0; k < DLM; ++k) uw[k] = (k < Dw && act) ? a.w1[row * Dw + k] / a.len[k] : 0.0;
    double jm[VL_BC];
    static_for<0, VL_BC>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        jm[c] = 0.0;
        if (c < b) jm[c] = act ? Iz * readlane_f64(Iz, c) : 0.0;
    });
    // I and J, dimension by dimension (functions.py:453-494 through the separable form)
    double Iv = 1.0;
#pragma unroll 1
    for (int k = 0; k < Dw; ++k) {
        const double x = a.w1[row * Dw + k], mk = a.m[t * Dw + k], vk = a.v[t * Dw + k], lk = a.len[k];
        Iv *= matern_I_dim(x, mk, vk, lk);
        double so[12], to[15], f2 = 0.0;
        if (vk != 0.0) {   // (uniform over the wave: the test point's input variance in this dimension)
            MaternDimConst kc;
            matern_dim_const(mk, vk, lk, kc);
            matern_role_S(x, kc, so, f2);
            matern_role_T(x, kc, to);
        } else {           // a deterministic input: the product of the two point correlations (functions.py:488-491)
            const double pt = matern_point(mk - x, lk);
#pragma unroll
            for (int q = 0; q < 12; ++q) so[q] = 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q) to[q] = 0.0;
            so[0] = pt;
            to[0] = pt;
        }
        static_for<0, VL_BC>([&](auto ic) {
            constexpr int c = decltype(ic)::value;
            if (c < b) {
                const double xc = readlane_f64(x, c), f2c = readlane_f64(f2, c);
                double o1 = 0.0, o2 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    o1 = fma(so[q], readlane_f64(to[q], c), o1);       // this lane's point is the smaller one
                    o2 = fma(readlane_f64(so[q], c), to[q], o2);       // column c's point is the smaller one
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    e1 = fma(so[6 + q], readlane_f64(to[12 + q], c), e1);
                    e2 = fma(readlane_f64(so[6 + q], c), to[12 + q], e2);
                }
                const double f = x <= xc ? fma(f2c - f2, e1, o1) : fma(f2 - f2c, e2, o2);
                jm[c] *= f;
            }
        });
    }
    Iv = act ? Iv * Iz : 0.0;
    const double yv = act ? a.y[row] : 0.0;
    const double dg = 1.0 + a.nugget * (act ? a.nugget_diag[row] : 1.0);
    double reg[VL_BC];
    static_for<0, VL_BC>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        reg[c] = 0.0;
        if (c < b) {
            double pr = 1.0, sd = 0.0;
#pragma unroll
            for (int k = 0; k < DLM; ++k) corr_accum_matern(uw[k] - readlane_f64(uw[k], c), pr, sd);   // (dimensions past Dw: 0 - 0)
#pragma unroll
            for (int g = 0; g < DGM; ++g) corr_accum_matern(ug[g] - readlane_f64(ug[g], c), pr, sd);
            const double kv = lane == c ? dg : pr * exp_negated(SQRT5 * sd);
            reg[c] = act ? kv : 0.0;
        }
    });
    linkgp_reg_finish(reg, jm, yv, Iv, act, b, lane, a, t);
}

void launch_vecchia_linkgp_matern_reg(dgpamd_ctx *ctx, const VLinkArgs &a) {
    const unsigned grid = (unsigned)((a.M + 3) / 4);
    if (a.Dz == 0)
        hipLaunchKernelGGL(vecchia_linkgp_matern_reg_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(vecchia_linkgp_matern_reg_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, a);
}

void launch_vecchia_linkgp_sexp_reg(dgpamd_ctx *ctx, const VLinkArgs &a) {
    const unsigned grid = (unsigned)((a.M + 3) / 4);
    if (a.Dz == 0)
        hipLaunchKernelGGL(vecchia_linkgp_sexp_reg_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(vecchia_linkgp_sexp_reg_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, a);
}
