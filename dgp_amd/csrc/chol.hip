// Blocked Cholesky / triangular inverse on augmented buffers (SURVEY 8 a3,a5,a8,a10).
// Replaces the LAPACK potrf/potrs call sites of the reference
// (kernel_class.py:417-423,483-487,746-748; functions.py:109,119).
//
// Right-looking, 64-wide block columns.  Per block column k:
//   potrf_diag   one workgroup per matrix: the 64x64 diagonal block is held in
//                registers (4x4 strided micro-tiles), each pivot column is
//                broadcast through LDS with ONE barrier per pivot, and the same
//                loop applies the eliminations to an identity -> the block's
//                inverse comes out for free (used instead of a triangular solve);
//   tile_gemm    TRSM as  P_i = A_ik * Linv_kk^T   (f64 MFMA 16x16x4),
//                SYRK as  A_ij -= P_i P_j^T        (lower tiles only).
// The same tile_gemm engine runs the blocked triangular inverse (recursive
// doubling over block pairs) and K^-1 = L^-T L^-1.
#include "common.hpp"
#include "tile.hpp"

// ----------------------------------------------------------------------------
// diagonal block: Cholesky + inverse of the factor, in registers
// ----------------------------------------------------------------------------
__device__ __forceinline__ double rsqrt_f64(double d) {
    // Goldschmidt: g -> sqrt(d), h -> 1/(2 sqrt(d)) refined in parallel from the hardware estimate (two rounds
    // reach full f64); dependency depth 6 instead of 9 for Newton on the reciprocal alone.  The pivot chain of the
    // factorisation is latency bound (a dependent f64 op is ~13 ns on MI355X), so depth is what counts.
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return h + h;
}

// The 64 pivots are taken FOUR at a time: the 4x4 pivot block is factored redundantly in every thread's
// registers (that is the serial chain), the scaled 64x4 panel and the four finished rows of the running
// inverse go through LDS once, and the trailing update is rank-4: two barriers per four pivots.
struct DiagShared {
    double colraw[4][64];    // raw pivot columns j0..j0+3
    double rowraw[4][64];    // raw rows j0..j0+3 of the running inverse
    double panL[64][4];      // scaled panel  L[r][j0+u]
    double panY[64][4];      // finished rows Linv[j0+u][c]
    double piv[64];
};

// Factor the 64x64 tile held in registers (thread (tx,ty) owns rows ty+16p, columns tx+16q), write the factor to
// Ab (ld) and the inverse of the factor to Wb (64x64), accumulate logdet / info of matrix b.  ncol = pivots in
// this block (rows/columns beyond are carried right-hand sides).
__device__ __forceinline__ void diag_factor(double (&a)[4][4], DiagShared &sh, double *Ab, int64_t ld, double *Wb,
                                            int ncol, int k, int b, double *logdet, int32_t *info) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double y[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[p][q] = (ty + 16 * p == tx + 16 * q) ? 1.0 : 0.0;
    if (tid < 64) sh.piv[tid] = 1.0;
    int bad = 0;

#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        // in block jb only register rows p >= jb / columns q >= jb of A are live; inverse rows have columns q <= jb
        for (int jq = 0; jq < 4; ++jq) {
            const int j0 = jb * 16 + jq * 4;
            if (j0 >= ncol) break;
            const int nact = ncol - j0 >= 4 ? 4 : ncol - j0;   // active pivots of this panel
            if ((tx >> 2) == jq) {
#pragma unroll
                for (int p = jb; p < 4; ++p) sh.colraw[tx & 3][ty + 16 * p] = a[p][jb];
            }
            if ((ty >> 2) == jq) {
#pragma unroll
                for (int q = 0; q <= jb; ++q) sh.rowraw[ty & 3][tx + 16 * q] = y[jb][q];
            }
            __syncthreads();
            // ---- 4x4 pivot block, redundantly in every thread ----
            double d00 = sh.colraw[0][j0], d10 = sh.colraw[0][j0 + 1], d20 = sh.colraw[0][j0 + 2], d30 = sh.colraw[0][j0 + 3];
            double d11 = sh.colraw[1][j0 + 1], d21 = sh.colraw[1][j0 + 2], d31 = sh.colraw[1][j0 + 3];
            double d22 = sh.colraw[2][j0 + 2], d32 = sh.colraw[2][j0 + 3], d33 = sh.colraw[3][j0 + 3];
            if (!(d00 > 0.0)) { if (!bad) bad = j0 + 1; d00 = 1.0; }
            const double i0 = rsqrt_f64(d00);
            double L10 = d10 * i0, L20 = d20 * i0, L30 = d30 * i0;
            double p1 = fma(-L10, L10, d11);
            if (nact < 2) { p1 = 1.0; L10 = 0.0; }
            if (!(p1 > 0.0)) { if (!bad) bad = j0 + 2; p1 = 1.0; }
            const double i1 = rsqrt_f64(p1);
            double L21 = fma(-L20, L10, d21) * i1, L31 = fma(-L30, L10, d31) * i1;
            double p2 = fma(-L21, L21, fma(-L20, L20, d22));
            if (nact < 3) { p2 = 1.0; L20 = 0.0; L21 = 0.0; }
            if (!(p2 > 0.0)) { if (!bad) bad = j0 + 3; p2 = 1.0; }
            const double i2 = rsqrt_f64(p2);
            double L32 = fma(-L31, L21, fma(-L30, L20, d32)) * i2;
            double p3 = fma(-L32, L32, fma(-L31, L31, fma(-L30, L30, d33)));
            if (nact < 4) { p3 = 1.0; L30 = 0.0; L31 = 0.0; L32 = 0.0; }
            if (!(p3 > 0.0)) { if (!bad) bad = j0 + 4; p3 = 1.0; }
            const double i3 = rsqrt_f64(p3);
            if (tid == 0) {
                sh.piv[j0] = d00;
                if (nact > 1) sh.piv[j0 + 1] = p1;
                if (nact > 2) sh.piv[j0 + 2] = p2;
                if (nact > 3) sh.piv[j0 + 3] = p3;
            }
            // ---- scaled panel (one row per thread) and finished inverse rows (one column per thread) ----
            if (tid < 64) {
                const int r = tid;
                const double a0 = sh.colraw[0][r], a1 = sh.colraw[1][r], a2 = sh.colraw[2][r], a3 = sh.colraw[3][r];
                const double x0 = a0 * i0;
                const double x1 = nact > 1 ? fma(-x0, L10, a1) * i1 : 0.0;
                const double x2 = nact > 2 ? fma(-x1, L21, fma(-x0, L20, a2)) * i2 : 0.0;
                const double x3 = nact > 3 ? fma(-x2, L32, fma(-x1, L31, fma(-x0, L30, a3))) * i3 : 0.0;
                sh.panL[r][0] = x0; sh.panL[r][1] = x1; sh.panL[r][2] = x2; sh.panL[r][3] = x3;
            } else if (tid < 128) {
                const int c = tid - 64;
                const bool in = c < 16 * (jb + 1);
                const double y0 = in ? sh.rowraw[0][c] : 0.0, y1 = in ? sh.rowraw[1][c] : 0.0;
                const double y2 = in ? sh.rowraw[2][c] : 0.0, y3 = in ? sh.rowraw[3][c] : 0.0;
                const double f0 = y0 * i0;
                const double f1 = fma(-L10, f0, y1) * i1;
                const double f2 = fma(-L21, f1, fma(-L20, f0, y2)) * i2;
                const double f3 = fma(-L32, f2, fma(-L31, f1, fma(-L30, f0, y3))) * i3;
                sh.panY[c][0] = f0; sh.panY[c][1] = f1; sh.panY[c][2] = f2; sh.panY[c][3] = f3;
            }
            __syncthreads();
            // ---- rank-4 trailing update, assignment of the finished columns / inverse rows ----
            double Lr[4][4], Lc[4][4], Yf[4][4];
#pragma unroll
            for (int p = jb; p < 4; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    Lr[p][u] = sh.panL[ty + 16 * p][u];
                    Lc[p][u] = sh.panL[tx + 16 * p][u];
                }
#pragma unroll
            for (int q = 0; q <= jb; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) Yf[q][u] = sh.panY[tx + 16 * q][u];
            const int ur = ty & 3, uc = tx & 3;
#pragma unroll
            for (int p = jb; p < 4; ++p) {
                const bool below = (p > jb) || (ty >= 4 * jq + nact);            // row beyond the active pivots
                const bool inblk = (p == jb) && ((ty >> 2) == jq) && (ur < nact); // row j0+ur of the pivot block
#pragma unroll
                for (int q = jb; q < 4; ++q) {
                    const bool right = (q > jb) || (tx >= 4 * jq + nact);
                    if (below && right) {
                        double v = a[p][q];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v = fma(-Lr[p][u], Lc[q][u], v);
                        a[p][q] = v;
                    }
                }
#pragma unroll
                for (int q = 0; q <= jb; ++q) {
                    if (below) {
                        double v = y[p][q];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v = fma(-Lr[p][u], Yf[q][u], v);
                        y[p][q] = v;
                    } else if (inblk) {
                        y[p][q] = ur == 0 ? Yf[q][0] : (ur == 1 ? Yf[q][1] : (ur == 2 ? Yf[q][2] : Yf[q][3]));
                    }
                }
                // finished column c = j0+uc: rows r >= c take the panel value (diagonal = sqrt(pivot))
                if ((tx >> 2) == jq && uc < nact) {
                    const bool onorbelow = below || (inblk && (ur >= uc));
                    if (onorbelow) a[p][jb] = uc == 0 ? Lr[p][0] : (uc == 1 ? Lr[p][1] : (uc == 2 ? Lr[p][2] : Lr[p][3]));
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int r = ty + 16 * p, c = tx + 16 * q;
            Ab[(int64_t)r * ld + c] = a[p][q];
            Wb[r * 64 + c] = y[p][q];
        }
    __syncthreads();
    if (tid < 64) {
        double v = (tid < ncol) ? log(sh.piv[tid]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (tid == 0) {
            logdet[b] = (k == 0 ? 0.0 : logdet[b]) + v;
            if (k == 0) info[b] = 0;
            if (bad && info[b] == 0) info[b] = k * 64 + bad;
        }
    }
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(double *A, int64_t ld, int64_t stride_a, int k, int64_t n,
                                                         double *ws, int64_t stride_ws, double *logdet,
                                                         int32_t *info, int32_t *flags) {
    __shared__ DiagShared sh;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double *Ab = A + (int64_t)b * stride_a + ((int64_t)k * 64) * ld + (int64_t)k * 64;
    double *Wb = ws + (int64_t)b * stride_ws + (int64_t)k * 4096;
    int64_t rem = n - (int64_t)k * 64;
    const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
    double a[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[p][q] = Ab[(int64_t)(ty + 16 * p) * ld + tx + 16 * q];
    diag_factor(a, sh, Ab, ld, Wb, ncol, k, b, logdet, info);
    if (tid == 0 && flags) flags[b] = k + 1;   // visible to the next launch (kernel boundary)
}

// ----------------------------------------------------------------------------
// 64x64x64 tile GEMM engine on f64 MFMA
// ----------------------------------------------------------------------------
enum { G_TRSM = 0, G_SYRK = 1, G_TRTRI1 = 2, G_TRTRI2 = 3, G_LAUUM = 4 };

struct GemmArgs {
    double *A;        // Np x Np buffers
    double *B;        // second buffer (temp / inverse)
    const double *ws; // diagonal-block inverses
    int64_t ld, stride_a, stride_ws;
    int64_t n;
    int nbk;   // blocks per dimension
    int k;     // block column (TRSM / SYRK)
    int s;     // half-size of the pair in blocks (TRTRI)
};

template <int MODE>
__global__ __launch_bounds__(256) void tile_gemm_kernel(GemmArgs g) {
    constexpr int OPA = (MODE == G_LAUUM) ? OP_KM : OP_MK;
    constexpr int OPB = (MODE == G_TRSM || MODE == G_SYRK) ? OP_MK : OP_KM;
    __shared__ double As[(OPA == OP_MK) ? 64 * LDM : KC * LDK];
    __shared__ double Bs[(OPB == OP_MK) ? 64 * LDM : KC * LDK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ld = g.ld;
    double *A = g.A + (int64_t)blockIdx.z * g.stride_a;
    double *B = g.B ? g.B + (int64_t)blockIdx.z * g.stride_a : nullptr;

    int bi, bj, kb0, kb1;          // output tile, k-block range [kb0, kb1)
    double *C;                     // output buffer
    double sign = 1.0;
    bool accumulate = false;
    if (MODE == G_TRSM) {
        bi = g.k + 1 + blockIdx.x; bj = g.k; kb0 = g.k; kb1 = g.k + 1; C = A;
    } else if (MODE == G_SYRK) {
        int ti, tj;
        tri_decode(blockIdx.x, ti, tj);
        bi = g.k + 1 + ti; bj = g.k + 1 + tj; kb0 = g.k; kb1 = g.k + 1; C = A;
        sign = -1.0; accumulate = true;
    } else if (MODE == G_TRTRI1 || MODE == G_TRTRI2) {
        const int s = g.s, per = s * s;
        const int p = blockIdx.x / per, rem = blockIdx.x - p * per;
        const int base = 2 * p * s;
        bi = base + s + rem / s; bj = base + rem % s;
        if (bi >= g.nbk) return;
        if (MODE == G_TRTRI1) {           // T = L21 * Linv11   (Linv11 lower: kb >= bj)
            kb0 = bj; kb1 = base + s; C = B;
        } else {                          // X21 = -Linv22 * T  (Linv22 lower: kb <= bi)
            kb0 = base + s; kb1 = bi + 1; C = A; sign = -1.0;
        }
    } else {                              // LAUUM: Kinv_ij = sum_{kb >= bi} Linv[kb][bi]^T Linv[kb][bj]
        tri_decode(blockIdx.x, bi, bj);
        kb0 = bi; kb1 = (int)((g.n + 63) / 64); C = B;
    }

    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    if (accumulate) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[t][r] = C[((int64_t)bi * 64 + crow + 4 * r) * ld + (int64_t)bj * 64 + 16 * t + ccol];
    }

    for (int kb = kb0; kb < kb1; ++kb) {
        const double *Ag, *Bg;
        int64_t lda = ld, ldb = ld;
        int limA = 64, limB = 64;
        if (MODE == G_TRSM) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = g.ws + (int64_t)blockIdx.z * g.stride_ws + (int64_t)kb * 4096; ldb = 64;
        } else if (MODE == G_SYRK) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = A + ((int64_t)bj * 64) * ld + (int64_t)kb * 64;
        } else if (MODE == G_TRTRI1) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = A + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
        } else if (MODE == G_TRTRI2) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = B + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
        } else {
            Ag = A + ((int64_t)kb * 64) * ld + (int64_t)bi * 64;
            Bg = A + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
            int64_t lim = g.n - (int64_t)kb * 64;   // rows >= n (right-hand sides) do not belong to L^-1
            limA = limB = lim >= 64 ? 64 : (int)lim;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if (OPA == OP_MK) load_mk(Ag, lda, As, tid, h); else load_km(Ag, lda, As, tid, h, limA);
            if (OPB == OP_MK) load_mk(Bg, ldb, Bs, tid, h); else load_km(Bg, ldb, Bs, tid, h, limB);
            __syncthreads();
            mfma_tile<OPA, OPB>(As, Bs, acc, wave, lane, sign);
        }
    }

#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gr = (int64_t)bi * 64 + crow + 4 * r, gc = (int64_t)bj * 64 + 16 * t + ccol;
            C[gr * ld + gc] = acc[t][r];
            if (MODE == G_LAUUM && bi != bj) C[gc * ld + gr] = acc[t][r];
        }
}

// ----------------------------------------------------------------------------
// One block step of the factorisation as ONE launch (k >= 1):
//   every trailing tile (i,j), i >= j >= k, takes its update with panel k-1;
//   the workgroup of tile (k,k) then factors it (diag_factor) and publishes the block's inverse with an
//   agent-scope release; the workgroups of column k (tiles (i,k), i > k) wait for that flag and apply
//   P_i = A_ik Linv_k^T.  The serial pivot chain of step k thus overlaps the bulk of the trailing update,
//   and a factorisation is nbk+1 launches instead of 3 nbk.
// Deadlock freedom: only the <= (nbk-k-1)*batch <= 31*64 column-k workgroups ever wait, and they wait for a
// workgroup of the SAME launch that never waits itself; waiting workgroups are far fewer than the resident
// capacity whenever batch <= 8 and otherwise every non-waiting workgroup terminates, so the diagonal workgroup
// is scheduled under any dispatch order.  The spin is bounded (info = -1 on timeout).
// ----------------------------------------------------------------------------
struct StepArgs {
    double *A;
    double *ws;
    int64_t ld, stride_a, stride_ws, n;
    int nbk, k, batch;
    double *logdet;
    int32_t *info;
    int32_t *flags;
};

__global__ __launch_bounds__(256) void potrf_step_kernel(StepArgs g) {
    __shared__ double tiles[2 * 64 * LDM];   // As | Bs ; the diagonal workgroup reuses it as its 64x64 tile
    __shared__ DiagShared sh;
    double *As = tiles, *Bs = tiles + 64 * LDM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // 1-D grid ordered by urgency: the diagonal tiles of ALL matrices first, then their column-k tiles, then the
    // bulk of the trailing update (dispatch follows the block index in practice, so the serial pivot chain of every
    // matrix starts at once and the bulk fills the machine behind it).
    const int k = g.k, kp = g.k - 1, m = g.nbk - g.k, batch = g.batch;
    int b, ti, tj;
    {
        const int idx = blockIdx.x;
        if (idx < batch * m) {
            b = idx % batch; ti = idx / batch; tj = 0;
        } else {
            const int bulk = m * (m - 1) / 2, r = idx - batch * m;
            b = r / bulk;
            tri_decode(r - b * bulk, ti, tj);
            ++ti; ++tj;
        }
    }
    const int bi = k + ti, bj = k + tj;
    const int64_t ld = g.ld;
    double *A = g.A + (int64_t)b * g.stride_a;
    double *C = A + ((int64_t)bi * 64) * ld + (int64_t)bj * 64;
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;

    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol];
    const double *Pi = A + ((int64_t)bi * 64) * ld + (int64_t)kp * 64;
    const double *Pj = A + ((int64_t)bj * 64) * ld + (int64_t)kp * 64;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        load_mk(Pi, ld, As, tid, h);
        load_mk(Pj, ld, Bs, tid, h);
        __syncthreads();
        mfma_tile<OP_MK, OP_MK>(As, Bs, acc, wave, lane, -1.0);
    }
    if (tj != 0) {   // plain trailing tile
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol] = acc[t][r];
        return;
    }
    if (ti == 0) {   // the next diagonal tile: factor it right away
        __syncthreads();
        double *S = tiles;   // [64][64]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(crow + 4 * r) * 64 + 16 * t + ccol] = acc[t][r];
        __syncthreads();
        const int tx = tid & 15, ty = tid >> 4;
        double a[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[p][q] = S[(ty + 16 * p) * 64 + tx + 16 * q];
        int64_t rem = g.n - (int64_t)k * 64;
        const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        diag_factor(a, sh, C, ld, g.ws + (int64_t)b * g.stride_ws + (int64_t)k * 4096, ncol, k, b, g.logdet, g.info);
        // publish the inverse: every storing wave drains its stores, one agent-scope release, then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&g.flags[b], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // panel tile (bi, k): wait for Linv_k, then P_i = A_ik * Linv_k^T
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(&g.flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 24)) {
                g.info[b] = -1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const double *Linv = g.ws + (int64_t)b * g.stride_ws + (int64_t)k * 4096;
    d4 out[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(crow + 4 * r) * LDM + 16 * t + ccol] = acc[2 * h + t][r];
        load_mk(Linv, 64, Bs, tid, h);
        __syncthreads();
        mfma_tile<OP_MK, OP_MK>(As, Bs, out, wave, lane, 1.0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol] = out[t][r];
}

// place the diagonal-block inverses on the diagonal of the (to be inverted) factor
__global__ __launch_bounds__(256) void put_diag_inverse_kernel(double *A, int64_t ld, const double *ws,
                                                               int64_t stride_a, int64_t stride_ws) {
    const int kb = blockIdx.x, tid = threadIdx.x;
    double *Ab = A + (int64_t)blockIdx.y * stride_a + ((int64_t)kb * 64) * ld + (int64_t)kb * 64;
    const double *W = ws + (int64_t)blockIdx.y * stride_ws + (int64_t)kb * 4096;
    for (int idx = tid; idx < 4096; idx += 256) Ab[(int64_t)(idx >> 6) * ld + (idx & 63)] = W[idx];
}

// rows [n, n+r) of L^-1 (columns < n) are -alpha^T: copy them beside K^-1
__global__ void copy_aug_rows_kernel(const double *A, double *B, int64_t ld, int64_t n, int r, int64_t stride_a) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    A += (int64_t)blockIdx.y * stride_a;
    B += (int64_t)blockIdx.y * stride_a;
    for (int q = 0; q < r; ++q) B[(n + q) * ld + j] = A[(n + q) * ld + j];
}

__global__ void aug_quad_kernel(const double *A, int64_t ld, int64_t stride_a, int64_t n, int r, double *quad) {
    int b = blockIdx.x, t = threadIdx.x;
    if (t < r * r) {
        int q = t / r, q2 = t % r;
        quad[(int64_t)b * r * r + t] = -A[(int64_t)b * stride_a + (n + q) * ld + n + q2];
    }
}

// out[b][i] = sqrt(scale_b) * sum_{j<=i} L[i][j] z[b][j]; one wave per row
struct TrmvArgs {
    const double *L;
    int64_t ld, stride_a, n;
    const double *z;
    double *out;
    double sscale[DGPAMD_MAXB];
};
__global__ __launch_bounds__(256) void trmv_lower_kernel(TrmvArgs a) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= a.n) return;
    const double *row = a.L + (int64_t)b * a.stride_a + i * a.ld;
    const double *z = a.z + (int64_t)b * a.n;
    double s = 0.0;
    for (int64_t j = lane; j <= i; j += 64) s = fma(row[j], z[j], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) a.out[(int64_t)b * a.n + i] = a.sscale[b] * s;
}

// ----------------------------------------------------------------------------
// host drivers
// ----------------------------------------------------------------------------
size_t potrf_ws_doubles(int64_t n, int batch) {
    int64_t nbk = padded_dim(n) / 64;
    return (size_t)batch * nbk * 4096 + 2 * DGPAMD_MAXB;   // diagonal inverses + {logdet (loglik), logdet (graph)} scratch; info words follow
}

extern "C" size_t dgpamd_potrf_workspace(int64_t n, int batch) {
    return potrf_ws_doubles(n, batch) * sizeof(double) + 2 * DGPAMD_MAXB * sizeof(int32_t);   // + info, step flags
}

static int potrf_launches(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet,
                          int32_t *info, double *ws, int32_t *flags) {
    const int64_t Np = padded_dim(n);
    const int nbk = (int)(Np / 64);
    const int64_t stride_ws = (int64_t)nbk * 4096;
    const double tile_flops = 2.0 * 64.0 * 64.0 * 64.0;
    // step 0: diagonal block and first panel as two launches
    PROF_BEGIN(ctx, PROF_POTRF_DIAG, (double)batch * (64.0 * 64.0 * 64.0));
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(batch), dim3(256), 0, ctx->stream, A, Np, stride_a, 0, n, ws, stride_ws,
                       logdet, info, flags);
    PROF_END(ctx, PROF_POTRF_DIAG);
    if (nbk > 1) {
        GemmArgs g;
        g.A = A; g.B = nullptr; g.ws = ws; g.ld = Np; g.stride_a = stride_a; g.stride_ws = stride_ws;
        g.n = n; g.nbk = nbk; g.k = 0; g.s = 0;
        PROF_BEGIN(ctx, PROF_TRSM, (double)batch * (nbk - 1) * tile_flops);
        hipLaunchKernelGGL(tile_gemm_kernel<G_TRSM>, dim3(nbk - 1, 1, batch), dim3(256), 0, ctx->stream, g);
        PROF_END(ctx, PROF_TRSM);
    }
    // steps 1..nbk-1: trailing update with panel k-1 + factorisation of block k + panel k, fused
    StepArgs st;
    st.A = A; st.ws = ws; st.ld = Np; st.stride_a = stride_a; st.stride_ws = stride_ws; st.n = n; st.nbk = nbk;
    st.logdet = logdet; st.info = info; st.flags = flags; st.batch = batch;
    for (int k = 1; k < nbk; ++k) {
        const int m = nbk - k;
        st.k = k;
        PROF_BEGIN(ctx, PROF_SYRK, (double)batch * (m * (m + 1) / 2 + (m - 1)) * tile_flops);
        hipLaunchKernelGGL(potrf_step_kernel, dim3((unsigned)(batch * (m * (m + 1) / 2))), dim3(256), 0, ctx->stream, st);
        PROF_END(ctx, PROF_SYRK);
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

__global__ void potrf_copy_out_kernel(const double *ld_ws, const int32_t *info_ws, double *logdet, int32_t *info, int batch) {
    const int b = threadIdx.x;
    if (b < batch) {
        logdet[b] = ld_ws[b];
        info[b] = info_ws[b];
    }
}

int run_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet, int32_t *info,
              double *ws) {
    // 3 launches per 64-column block step with a static shape: replayed as one hipGraph.  The graph writes
    // logdet/info into the workspace tail (fixed addresses -> the cached graph does not depend on where the
    // caller wants them); a tiny kernel outside the graph copies them out.
    const int64_t nbk = padded_dim(n) / 64;
    double *ld_ws = ws + (size_t)batch * nbk * 4096 + DGPAMD_MAXB;
    int32_t *info_ws = reinterpret_cast<int32_t *>(ws + (size_t)batch * nbk * 4096 + 2 * DGPAMD_MAXB);
    int32_t *flags = info_ws + DGPAMD_MAXB;
    const std::array<uint64_t, 10> key = {1, (uint64_t)n, (uint64_t)batch, (uint64_t)A, (uint64_t)stride_a, (uint64_t)ws,
                                          0, 0, 0, 0};
    int rc = graph_run(ctx, key, [&]() { return potrf_launches(ctx, n, A, stride_a, batch, ld_ws, info_ws, ws, flags); });
    if (rc) return rc;
    hipLaunchKernelGGL(potrf_copy_out_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)ld_ws,
                       (const int32_t *)info_ws, logdet, info, batch);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet,
                            int32_t *info, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !A || !logdet || !info || !work) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    const int64_t Np = padded_dim(n);
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    return run_potrf(ctx, n, A, stride_a, batch, logdet, info, (double *)work);
}

extern "C" int dgpamd_aug_quad(dgpamd_ctx *ctx, int64_t n, const double *A, int64_t stride_a, int batch, int r,
                               double *quad) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!A || !quad || r <= 0 || r * r > 256 || n + r > padded_dim(n)) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(aug_quad_kernel, dim3(batch), dim3(256), 0, ctx->stream, A, padded_dim(n), stride_a, n, r, quad);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_trmv_lower(dgpamd_ctx *ctx, int64_t n, const double *L, int64_t stride_a, const double *scale_h,
                                 const double *z, double *out, int batch) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !L || !z || !out || !scale_h) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    TrmvArgs a;
    a.L = L; a.ld = padded_dim(n); a.stride_a = stride_a; a.n = n; a.z = z; a.out = out;
    for (int b = 0; b < batch; ++b) a.sscale[b] = sqrt(scale_h[b]);
    hipLaunchKernelGGL(trmv_lower_kernel, dim3((unsigned)((n + 3) / 4), batch), dim3(256), 0, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

static int potri_launches(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int64_t stride_a, int r, int batch,
                          void *work);

extern "C" int dgpamd_potri_batched(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int64_t stride_a, int r,
                                    int batch, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !A || !Ainv || !work || r < 0) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    const int64_t Np = padded_dim(n);
    if (n + r > Np) BAD_ARG(ctx, "too many right-hand sides");
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    const std::array<uint64_t, 10> key = {2, (uint64_t)n, (uint64_t)r, (uint64_t)A, (uint64_t)Ainv, (uint64_t)work,
                                          (uint64_t)batch, (uint64_t)stride_a, 0, 0};
    return graph_run(ctx, key, [&]() { return potri_launches(ctx, n, A, Ainv, stride_a, r, batch, work); });
}

extern "C" int dgpamd_potri(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int r, void *work) {
    return dgpamd_potri_batched(ctx, n, A, Ainv, 0, r, 1, work);
}

static int potri_launches(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int64_t stride_a, int r, int batch,
                          void *work) {
    const int64_t Np = padded_dim(n);
    const int nbk = (int)(Np / 64);
    const int64_t stride_ws = (int64_t)nbk * 4096;
    GemmArgs g;
    g.A = A; g.B = Ainv; g.ws = (const double *)work; g.ld = Np; g.stride_a = stride_a; g.stride_ws = stride_ws;
    g.n = n; g.nbk = nbk; g.k = 0; g.s = 0;
    hipLaunchKernelGGL(put_diag_inverse_kernel, dim3(nbk, batch), dim3(256), 0, ctx->stream, A, Np, (const double *)work,
                       stride_a, stride_ws);
    for (int s = 1; s < nbk; s *= 2) {
        const int np = (nbk + 2 * s - 1) / (2 * s);
        g.s = s;
        hipLaunchKernelGGL(tile_gemm_kernel<G_TRTRI1>, dim3(np * s * s, 1, batch), dim3(256), 0, ctx->stream, g);
        hipLaunchKernelGGL(tile_gemm_kernel<G_TRTRI2>, dim3(np * s * s, 1, batch), dim3(256), 0, ctx->stream, g);
    }
    const int nbn = (int)((n + 63) / 64);
    hipLaunchKernelGGL(tile_gemm_kernel<G_LAUUM>, dim3(nbn * (nbn + 1) / 2, 1, batch), dim3(256), 0, ctx->stream, g);
    if (r > 0)
        hipLaunchKernelGGL(copy_aug_rows_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, ctx->stream, A,
                           Ainv, Np, n, r, stride_a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}
