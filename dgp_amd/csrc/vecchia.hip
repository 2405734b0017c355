// Vecchia-approximation kernels (reference dgpsi/vecchia.py).
//
// Thousands of tiny (m+1)x(m+1) problems: ONE WAVE (a 64-thread workgroup) per
// row / test point, the block matrix lives in LDS, the right-hand side rides
// along as an extra row of the factorisation (same trick as chol.hip), so
// forward solves cost nothing extra.  Neighbour search is exact brute force on
// device (multi-pass selection in (distance, index) order -> deterministic).
#include "common.hpp"
#include "linkfun.hpp"
#include "vecchia_pred.hpp"

#include <math.h>
#include <utility>

#define VW 64   // threads per item (one wave)

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}


static int fill_vparams(dgpamd_ctx *ctx, VParams &p, int kind, int D, const double *length_h, int nlen, double nugget) {
    if (kind != DGPAMD_SEXP && kind != DGPAMD_MATERN25) BAD_ARG(ctx, "kind must be 0 or 1");
    if (D <= 0 || D > DGPAMD_MAXD || !length_h || (nlen != 1 && nlen != D)) BAD_ARG(ctx, "bad D / nlen / length");
    p.kind = kind; p.D = D; p.nlen = nlen; p.nugget = nugget;
    for (int d = 0; d < D; ++d) p.inv_len[d] = 1.0 / length_h[nlen == 1 ? 0 : d];
    return DGPAMD_OK;
}

static int set_lds(dgpamd_ctx *ctx, const void *fn, size_t shm) {
    if (shm > 160 * 1024) BAD_ARG(ctx, "conditioning set too large for LDS (160 KiB)");
    if (shm > 48 * 1024) HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    return DGPAMD_OK;
}

// correlation of two SCALED points held in LDS
template <int KIND>
__device__ __forceinline__ double corr_pts(const double *xa, const double *xb, int D) {
    double s = 0.0, pr = 1.0;
    for (int d = 0; d < D; ++d) {
        double df = xa[d] - xb[d];
        if (KIND == DGPAMD_SEXP)
            corr_accum_sexp(df, s);
        else
            corr_accum_matern(df, pr, s);
    }
    return (KIND == DGPAMD_SEXP) ? exp(-s) : pr * exp(-SQRT5 * s);
}

template <int KIND>
__device__ __forceinline__ double dcoef_v(double df) {
    if (KIND == DGPAMD_SEXP) return 2.0 * df * df;
    double r = fabs(df);
    double e1 = fma(r, SQRT5, 1.0), e2 = (5.0 / 3.0) * r * r;
    const double den = e1 + e2;   // >= 1: hardware reciprocal + two Newton rounds instead of the ~30-instruction division
    double x = __builtin_amdgcn_rcp(den);
    x = fma(x, fma(-den, x, 1.0), x);
    x = fma(x, fma(-den, x, 1.0), x);
    return e2 * e1 * x;
}

// In-LDS right-looking Cholesky of the leading npiv columns of a `rows` x `rows` lower matrix (ld = lda).
// Rows beyond npiv are carried (right-hand sides / prediction rows).  Returns 0 or 1+index of a bad pivot.
__device__ int lds_chol(double *A, int lda, int rows, int npiv, int lane) {
    int bad = 0;
    for (int j = 0; j < npiv; ++j) {
        __syncthreads();
        double d = A[j * lda + j];
        if (!(d > 0.0)) {
            if (!bad) bad = j + 1;
            d = 1.0;
        }
        const double sd = sqrt(d), inv = 1.0 / sd;
        __syncthreads();
        for (int r = j + 1 + lane; r < rows; r += VW) A[r * lda + j] *= inv;
        if (lane == 0) A[j * lda + j] = sd;
        __syncthreads();
        for (int r = j + 1 + lane; r < rows; r += VW) {
            const double lr = A[r * lda + j];
            const int cmax = r < rows ? r : rows - 1;
            for (int c = j + 1; c <= cmax; ++c) A[r * lda + c] = fma(-lr, A[c * lda + j], A[r * lda + c]);
        }
    }
    __syncthreads();
    return bad;
}

// x <- L^-T x for nrhs vectors stored as rows X[q][.] (ld = ldx); column-oriented back substitution
__device__ void lds_backsolve_T(const double *L, int lda, int b, double *X, int ldx, int nrhs, int lane) {
    for (int c = b - 1; c >= 0; --c) {
        __syncthreads();
        if (lane < nrhs) X[lane * ldx + c] /= L[c * lda + c];
        __syncthreads();
        for (int q = 0; q < nrhs; ++q) {
            const double xc = X[q * ldx + c];
            for (int r = lane; r < c; r += VW) X[q * ldx + r] = fma(-L[c * lda + r], xc, X[q * ldx + r]);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// a17  neighbour search
// ---------------------------------------------------------------------------
// One 256-thread workgroup per query.  Pass p selects the p-th smallest (dist, index) pair.
__device__ __forceinline__ bool pair_less(double d1, int64_t i1, double d2, int64_t i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

__global__ __launch_bounds__(256) void nn_select_kernel(int64_t nq, int64_t nx, int D, const double *q, const double *x,
                                                        int m_out, int ordered, int64_t *out) {
    extern __shared__ double lds[];
    double *qs = lds;                                  // [D]
    double *rd = lds + D;                              // [4] wave minima (dist)
    int64_t *ri = reinterpret_cast<int64_t *>(rd + 4); // [4] wave minima (index)
    int64_t *sel = ri + 4;                             // [m_out] selected indices
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t iq = blockIdx.x;
    const int64_t ncand = ordered ? iq + 1 : nx;       // ordered: only points with index <= own
    const int k = (int)(ncand < m_out ? ncand : m_out);
    for (int d = tid; d < D; d += 256) qs[d] = q[iq * D + d];
    __syncthreads();
    double pd = -1.0;
    int64_t pi = -1;
    for (int pass = 0; pass < k; ++pass) {
        double bd = INFINITY;
        int64_t bidx = INT64_MAX;
        for (int64_t j = tid; j < ncand; j += 256) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double df = x[j * D + d] - qs[d];
                s = fma(df, df, s);
            }
            if (pair_less(pd, pi, s, j) && pair_less(s, j, bd, bidx)) {
                bd = s;
                bidx = j;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double od = __shfl_down(bd, off, 64);
            int64_t oi = __shfl_down(bidx, off, 64);
            if (pair_less(od, oi, bd, bidx)) {
                bd = od;
                bidx = oi;
            }
        }
        if (lane == 0) {
            rd[wave] = bd;
            ri[wave] = bidx;
        }
        __syncthreads();
        bd = rd[0];
        bidx = ri[0];
        for (int w = 1; w < 4; ++w)
            if (pair_less(rd[w], ri[w], bd, bidx)) {
                bd = rd[w];
                bidx = ri[w];
            }
        pd = bd;
        pi = bidx;
        if (tid == 0) sel[pass] = bidx;
        __syncthreads();
    }
    if (tid == 0) {
        if (ordered) {   // vecchia.py:108  np.fliplr(np.sort(NNarray)): index-descending, -1 padded
            for (int a = 1; a < k; ++a) {
                int64_t v = sel[a];
                int b = a - 1;
                while (b >= 0 && sel[b] < v) {
                    sel[b + 1] = sel[b];
                    --b;
                }
                sel[b + 1] = v;
            }
        }
        for (int a = 0; a < m_out; ++a) out[iq * m_out + a] = a < k ? sel[a] : -1;
    }
}

// Large candidate sets: the m_out selection passes above recompute every distance every pass.  Here a workgroup writes
// the distances of its query ONCE (global scratch row), narrows them with a 1024-bin histogram (zooming into the bin
// that holds the k-th smallest until at most NN_CMAX candidates are left), gathers those into LDS and runs the same
// (distance, index)-ordered selection passes on that short list -- identical output, ~m_out times less arithmetic.
// Workgroups are persistent over the queries (grid-stride), one scratch row each.
#define NN_NB 1024
#define NN_CMAX 2048
__global__ __launch_bounds__(256) void nn_select_big_kernel(int64_t nq, int64_t nx, int D, const double *q, const double *x,
                                                            int m_out, int ordered, int64_t *out, double *scratch) {
    extern __shared__ double lds[];
    double *qs = lds;                                   // [D]
    double *rd = qs + D;                                // [8] wave partials
    double *cd = rd + 8;                                // [NN_CMAX] candidate distances
    int64_t *ci = reinterpret_cast<int64_t *>(cd + NN_CMAX);   // [NN_CMAX] candidate indices
    int64_t *ri = ci + NN_CMAX;                         // [4]
    int64_t *sel = ri + 4;                              // [m_out]
    int *hist = reinterpret_cast<int *>(sel + m_out);   // [NN_NB]
    int *part = hist + NN_NB;                           // [256]
    int *ctrl = part + 256;                             // [4]: bin*, count below bin*, candidate counter
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *dist = scratch + (int64_t)blockIdx.x * nx;
    for (int64_t iq = blockIdx.x; iq < nq; iq += gridDim.x) {
        const int64_t ncand = ordered ? iq + 1 : nx;
        const int k = (int)(ncand < m_out ? ncand : m_out);
        __syncthreads();
        for (int d = tid; d < D; d += 256) qs[d] = q[iq * D + d];
        __syncthreads();
        // ---- distances, once
        double lo = INFINITY, hi = -INFINITY;
        for (int64_t j = tid; j < ncand; j += 256) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                double df = x[j * D + d] - qs[d];
                s = fma(df, df, s);
            }
            dist[j] = s;
            lo = fmin(lo, s);
            hi = fmax(hi, s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fmin(lo, __shfl_down(lo, off, 64));
            hi = fmax(hi, __shfl_down(hi, off, 64));
        }
        if (lane == 0) {
            rd[wave] = lo;
            rd[4 + wave] = hi;
        }
        __syncthreads();
        lo = fmin(fmin(rd[0], rd[1]), fmin(rd[2], rd[3]));
        hi = fmax(fmax(rd[4], rd[5]), fmax(rd[6], rd[7]));
        // ---- zoom: histogram of the values <= hi, find the bin of the k-th smallest
        int ncollect = 0;
        bool listed = false;
        for (int zoom = 0; zoom < 6; ++zoom) {
            const double scale = hi > lo ? (double)NN_NB / (hi - lo) : 0.0;
            __syncthreads();
            for (int b = tid; b < NN_NB; b += 256) hist[b] = 0;
            __syncthreads();
            for (int64_t j = tid; j < ncand; j += 256) {
                const double s = dist[j];
                if (s <= hi) {
                    int b = (int)((s - lo) * scale);
                    atomicAdd(&hist[b < NN_NB ? b : NN_NB - 1], 1);
                }
            }
            __syncthreads();
            part[tid] = hist[4 * tid] + hist[4 * tid + 1] + hist[4 * tid + 2] + hist[4 * tid + 3];
            __syncthreads();
            if (tid == 0) {
                int cum = 0, c = 0;
                while (c < 255 && cum + part[c] < k) cum += part[c++];
                int b = 4 * c;
                while (b < 4 * c + 3 && cum + hist[b] < k) cum += hist[b++];
                ctrl[0] = b;
                ctrl[1] = cum + hist[b];   // everything in the bins <= b
                ctrl[2] = 0;
            }
            __syncthreads();
            const int bstar = ctrl[0];
            ncollect = ctrl[1];
            if (ncollect <= NN_CMAX) {
                for (int64_t j = tid; j < ncand; j += 256) {
                    const double s = dist[j];
                    if (s <= hi) {
                        int b = (int)((s - lo) * scale);
                        b = b < NN_NB ? b : NN_NB - 1;
                        if (b <= bstar) {
                            const int pos = atomicAdd(&ctrl[2], 1);
                            cd[pos] = s;
                            ci[pos] = j;
                        }
                    }
                }
                listed = true;
                break;
            }
            // too many: keep only the bins <= b* and spread them over the histogram again.  The new upper end is the
            // largest VALUE kept (an exact, monotone filter: s <= nhi <=> bin(s) <= b*), not a recomputed bin edge.
            double nhi = -INFINITY;
            for (int64_t j = tid; j < ncand; j += 256) {
                const double s = dist[j];
                if (s <= hi) {
                    int b = (int)((s - lo) * scale);
                    b = b < NN_NB ? b : NN_NB - 1;
                    if (b <= bstar) nhi = fmax(nhi, s);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nhi = fmax(nhi, __shfl_down(nhi, off, 64));
            __syncthreads();
            if (lane == 0) rd[wave] = nhi;
            __syncthreads();
            nhi = fmax(fmax(rd[0], rd[1]), fmax(rd[2], rd[3]));
            if (!(nhi < hi) && bstar == NN_NB - 1) break;   // nothing was cut off
            if (!(nhi > lo)) break;                          // all remaining values equal: ties, stored-distance passes
            hi = nhi;
        }
        __syncthreads();
        // ---- selection passes in (distance, index) order, over the short list or (ties) over the stored distances
        double pd = -1.0;
        int64_t pi = -1;
        for (int pass = 0; pass < k; ++pass) {
            double bd = INFINITY;
            int64_t bidx = INT64_MAX;
            if (listed) {
                for (int c = tid; c < ncollect; c += 256)
                    if (pair_less(pd, pi, cd[c], ci[c]) && pair_less(cd[c], ci[c], bd, bidx)) {
                        bd = cd[c];
                        bidx = ci[c];
                    }
            } else {
                for (int64_t j = tid; j < ncand; j += 256) {
                    const double s = dist[j];
                    if (pair_less(pd, pi, s, j) && pair_less(s, j, bd, bidx)) {
                        bd = s;
                        bidx = j;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                double od = __shfl_down(bd, off, 64);
                int64_t oi = __shfl_down(bidx, off, 64);
                if (pair_less(od, oi, bd, bidx)) {
                    bd = od;
                    bidx = oi;
                }
            }
            if (lane == 0) {
                rd[wave] = bd;
                ri[wave] = bidx;
            }
            __syncthreads();
            bd = rd[0];
            bidx = ri[0];
            for (int w = 1; w < 4; ++w)
                if (pair_less(rd[w], ri[w], bd, bidx)) {
                    bd = rd[w];
                    bidx = ri[w];
                }
            pd = bd;
            pi = bidx;
            if (tid == 0) sel[pass] = bidx;
            __syncthreads();
        }
        if (tid == 0) {
            if (ordered) {
                for (int a = 1; a < k; ++a) {
                    int64_t v = sel[a];
                    int b = a - 1;
                    while (b >= 0 && sel[b] < v) {
                        sel[b + 1] = sel[b];
                        --b;
                    }
                    sel[b + 1] = v;
                }
            }
            for (int a = 0; a < m_out; ++a) out[iq * m_out + a] = a < k ? sel[a] : -1;
        }
    }
}

// ---------------------------------------------------------------------------
// Large candidate sets, few dimensions, at most 64 neighbours: a streaming top-k.  The store-once kernel above reads
// the candidate array once per QUERY (n^2/2 x 64 B = 80 GB of L2 traffic at n = 50 000: it runs at the L2's bandwidth)
// and then passes over n stored distances several times more.  Here a wave takes 64 queries, one per lane, and one chunk
// of the candidate range: 64 candidates at a time are staged in LDS and every lane reads them with wave-uniform
// (broadcast) addresses -- each candidate byte is loaded once per 64 queries.  A lane keeps its K best (distance, index)
// pairs SORTED IN REGISTERS; a candidate that beats the lane's K-th best bubbles in (K compare-and-swap steps, executed
// only when some lane of the wave has one: a quarter of the candidates at n = 50 000, K = 26).  The partial lists of the
// chunks go to global scratch and a second kernel merges them the same way and writes the output in the order the
// selection passes above produce.  Distances use the same fma chain, the order is the same (distance, index) order:
// identical neighbour arrays.
// ---------------------------------------------------------------------------
#include <limits.h>

template <int K>
__device__ __forceinline__ void topk_insert(double (&ld)[K], int (&li)[K], double cd, int ci) {
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const bool sw = cd < ld[t] || (cd == ld[t] && ci < li[t]);
        const double od = ld[t];
        const int oi = li[t];
        ld[t] = sw ? cd : od;
        li[t] = sw ? ci : oi;
        cd = sw ? od : cd;
        ci = sw ? oi : ci;
    }
}

#define NN_PEND 16   // accepted candidates a lane may hold back before the wave merges them into the sorted lists (8: 15 % slower in the query form -- the
                     // passes of a flush are as many as the fullest lane holds, mostly empty for the others; 32: the ordered search loses occupancy to the 24 KB of notes)

template <int DMAX, int K>
__global__ __launch_bounds__(64) void nn_scan_kernel(int64_t nq, int64_t nx, int D, const double *__restrict__ q,
                                                     const double *__restrict__ x, int ordered, int64_t chunk, int nchunk,
                                                     double *__restrict__ sd, int *__restrict__ si) {
    __shared__ __attribute__((aligned(16))) double tile[64 * DMAX];
    __shared__ double pend_d[NN_PEND * 64];
    __shared__ int pend_i[NN_PEND * 64];
    const int lane = threadIdx.x;
    const int64_t nblk = (nq + 63) / 64;
    const int64_t qb = ordered ? nblk - 1 - blockIdx.x : blockIdx.x;   // (ordered: the long scans first)
    const int64_t iq = qb * 64 + lane;
    const bool qlive = iq < nq;
    const int64_t nscan = ordered ? (qb * 64 + 64 < nx ? qb * 64 + 64 : nx) : nx;
    const int64_t lo = (int64_t)blockIdx.y * chunk;
    if (lo >= nscan) return;
    const int64_t hi = lo + chunk < nscan ? lo + chunk : nscan;
    const int64_t mine = ordered ? iq + 1 : nx;   // candidates j < mine
    double qv[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) qv[d] = (qlive && d < D) ? q[iq * D + d] : 0.0;
    double ld[K];
    int li[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        ld[t] = INFINITY;
        li[t] = INT_MAX;
    }
    // A candidate below the lane's K-th best is only NOTED (LDS); when some lane has noted NN_PEND of them the whole wave
    // bubbles its notes into the sorted lists: one pass of K compare-and-swap steps then serves up to 64 insertions
    // instead of one.  Until then the threshold is the old K-th best -- a few notes more than necessary, the same result.
    double tau = INFINITY;
    int cnt = 0;
    auto flush = [&]() {
        // (a LOOP over the passes: unrolled, the NN_PEND copies of the K-step insertion are 40 KB of code per call site, more
        //  than the instruction cache holds, and the waves -- each at another point of it -- ran at the speed of its misses)
#pragma unroll 1
        for (int p = 0; p < NN_PEND; ++p) {
            const bool have = p < cnt;
            if (!__any(have)) break;
            topk_insert<K>(ld, li, have ? pend_d[p * 64 + lane] : INFINITY, have ? pend_i[p * 64 + lane] : INT_MAX);
        }
        cnt = 0;
        tau = ld[K - 1];
    };
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t jc = base + lane;
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DMAX; ++d) tile[lane * DMAX + d] = (jc < hi && d < D) ? x[jc * D + d] : 0.0;
        __syncthreads();
        const int cnt_c = (int)(hi - base < 64 ? hi - base : 64);
        for (int c = 0; c < cnt_c; ++c) {
            double s = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {   // (dimensions past D: 0 - 0, fma(0, 0, s) = s)
                const double df = tile[c * DMAX + d] - qv[d];
                s = fma(df, df, s);
            }
            const int64_t j = base + c;
            const bool take = qlive && j < mine && s < tau;   // (equal distance: the earlier index stays)
            if (__any(take)) {
                if (take) {
                    pend_d[cnt * 64 + lane] = s;
                    pend_i[cnt * 64 + lane] = (int)j;
                    ++cnt;
                }
                if (__any(cnt == NN_PEND)) flush();
            }
        }
    }
    flush();
    const int64_t item = qb * nchunk + blockIdx.y;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        sd[(item * K + t) * 64 + lane] = ld[t];
        si[(item * K + t) * 64 + lane] = li[t];
    }
}

template <int K>
__global__ __launch_bounds__(64) void nn_merge_kernel(int64_t nq, int64_t nx, int m_out, int ordered, int64_t chunk, int nchunk,
                                                      const double *__restrict__ sd, const int *__restrict__ si,
                                                      int64_t *__restrict__ out) {
    const int lane = threadIdx.x;
    const int64_t qb = blockIdx.x, iq = qb * 64 + lane;
    const int64_t nscan = ordered ? (qb * 64 + 64 < nx ? qb * 64 + 64 : nx) : nx;
    const int nch = (int)((nscan + chunk - 1) / chunk);
    double ld[K];
    int li[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        ld[t] = sd[((qb * nchunk) * K + t) * 64 + lane];
        li[t] = si[((qb * nchunk) * K + t) * 64 + lane];
    }
    for (int c = 1; c < nch; ++c) {
        const int64_t item = qb * nchunk + c;
#pragma unroll 1   // (unrolled this was K copies of the K-step insertion: 137 KB of code at K = 51)
        for (int t = 0; t < K; ++t) {
            const double cd = sd[(item * K + t) * 64 + lane];
            const int ci = si[(item * K + t) * 64 + lane];
            if (__any(cd < ld[K - 1] || (cd == ld[K - 1] && ci < li[K - 1]))) topk_insert<K>(ld, li, cd, ci);
        }
    }
    if (iq >= nq) return;
    const int64_t mine = ordered ? iq + 1 : nx;
    const int kk = (int)(mine < m_out ? mine : m_out);   // valid neighbours: the first kk of the sorted list
    if (!ordered) {
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < m_out) out[iq * m_out + t] = t < kk ? li[t] : -1;
        return;
    }
    // vecchia.py:108  np.fliplr(np.sort(NNarray)): index-descending, -1 padded
    for (int t = kk; t < m_out; ++t) out[iq * m_out + t] = -1;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < kk) {
            int pos = 0;
#pragma unroll
            for (int u = 0; u < K; ++u) pos += (u < kk && li[u] > li[t]);
            out[iq * m_out + pos] = li[t];
        }
    }
}

template <int DMAX, int K>
static int launch_nn_stream(dgpamd_ctx *ctx, int64_t nq, int64_t nx, int D, const double *q, const double *x, int m_out, int ordered,
                            int64_t *out) {
    const int64_t nblk = (nq + 63) / 64;
    // chunks of the candidate range: enough (block, chunk) items to fill the chip, at most 8192 candidates each
    int64_t chunk = (nx * nblk / 4096 + 63) / 64 * 64;
    chunk = chunk < 1024 ? 1024 : (chunk > 8192 ? 8192 : chunk);
    // measured (n = 50 000, d = 8): every chunk starts with empty lists, so the query form -- few blocks of queries, each scanning
    // everything -- wants long chunks (10 000 queries: 9.8 ms with 1920-candidate chunks, 4.9 ms with 8192); the ordered search
    // has blocks enough and only gains from 16384 at the largest sizes (4.1 -> 3.7 ms at n = 50 000)
    if (!ordered)
        chunk = nq < 15000 ? 8192 : (nq < 40000 ? 16384 : 25024);
    else if (nx >= 45000)
        chunk = 16384;
    if (const char *ce = getenv("DGPAMD_NN_CHUNK")) chunk = atoll(ce) > 0 ? (atoll(ce) + 63) / 64 * 64 : chunk;   // (tuning aid)
    const int nchunk = (int)((nx + chunk - 1) / chunk);
    const size_t items = (size_t)nblk * nchunk;
    void *both = nullptr;
    int rc = ctx_scratch(ctx, 1, items * K * 64 * (sizeof(double) + sizeof(int)), &both);
    if (rc) return rc;
    double *sd = reinterpret_cast<double *>(both);
    int *si = reinterpret_cast<int *>(sd + items * K * 64);
    hipLaunchKernelGGL((nn_scan_kernel<DMAX, K>), dim3((unsigned)nblk, (unsigned)nchunk), dim3(64), 0, ctx->stream, nq, nx, D, q, x,
                       ordered, chunk, nchunk, sd, si);
    hipLaunchKernelGGL((nn_merge_kernel<K>), dim3((unsigned)nblk), dim3(64), 0, ctx->stream, nq, nx, m_out, ordered, chunk, nchunk,
                       (const double *)sd, (const int *)si, out);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// Round 5, the QUERY form (vecchia.py:20-40) at prediction sizes -- 1e5 queries against 5e4 points, 50 neighbours: filter, then select.
// The streaming kernel above keeps K = 51 sorted pairs per lane in registers: 256 VGPRs, ONE wave per SIMD, so every LDS read
// and every dependent issue of its distance loop was exposed (one instruction per ~11 cycles; 43 ms at D = 16, a third of what its
// own instruction count needs).  The top-k state does not have to ride along the scan:
//   (A) nn_tau_kernel: per query the r-th smallest distance to a strided sample of S candidates (a small sorted list: r <= 32) -- an
//       upper bound tau of its K-th nearest distance unless fewer than K of ALL candidates lie at or below it (r is chosen so that
//       this has probability ~1e-8 per query, and it is CHECKED);
//   (B) nn_collect_kernel: the scan itself -- 256 queries per workgroup share the staged candidates, a lane holds its query and tau
//       (~60 VGPRs: five waves per SIMD) and notes every candidate with distance <= tau, through LDS, in a region of global memory of
//       its own (per query and candidate chunk: no atomics);
//   (C) nn_pick_kernel: per query the K nearest of its few hundred notes, by the same (distance, index) insertion as above.  A query
//       whose notes are fewer than K, or overflowed their region, is scanned exactly by its lane (the rare slow path).
// The distances are the same fma chains, the order the same (distance, index) order: identical neighbour arrays (tests).
// ---------------------------------------------------------------------------
#define NNF_R 32        // sample list length (r <= NNF_R)
#define NNF_PEND 8      // notes a lane holds in LDS before the wave writes them out

template <int DMAX>
__global__ __launch_bounds__(64) void nn_tau_kernel(int64_t nq, int64_t nx, int D, const double *__restrict__ q, const double *__restrict__ x,
                                                    int64_t S, int64_t stride, int r, double *__restrict__ tau) {
    __shared__ __attribute__((aligned(16))) double tile[64 * DMAX];
    const int lane = threadIdx.x;
    const int64_t iq = (int64_t)blockIdx.x * 64 + lane;
    const bool qlive = iq < nq;
    double qv[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) qv[d] = (qlive && d < D) ? q[iq * D + d] : 0.0;
    double ld[NNF_R];
    int li[NNF_R];
#pragma unroll
    for (int t = 0; t < NNF_R; ++t) {
        ld[t] = INFINITY;
        li[t] = INT_MAX;
    }
    for (int64_t base = 0; base < S; base += 64) {
        const int64_t js = (base + lane) * stride;   // the sample: every stride-th candidate
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DMAX; ++d) tile[lane * DMAX + d] = (base + lane < S && js < nx && d < D) ? x[js * D + d] : 0.0;
        __syncthreads();
        const int cnt_c = (int)(S - base < 64 ? S - base : 64);
        for (int c = 0; c < cnt_c; ++c) {
            double sacc = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const double df = tile[c * DMAX + d] - qv[d];
                sacc = fma(df, df, sacc);
            }
            if (__any(sacc < ld[NNF_R - 1])) topk_insert<NNF_R>(ld, li, sacc, (int)(base + c));
        }
    }
    double t = ld[0];
#pragma unroll
    for (int u = 1; u < NNF_R; ++u) t = (u < r) ? ld[u] : t;   // the r-th smallest
    if (qlive) tau[iq] = t;
}

// grid (query blocks of 256, candidate chunks); notes of query iq in chunk ch: nd / ni [(iq * nchunk + ch) * capc ...], count in ncnt
template <int DMAX>
__global__ __launch_bounds__(256) void nn_collect_kernel(int64_t nq, int64_t nx, int D, const double *__restrict__ q, const double *__restrict__ x,
                                                         const double *__restrict__ tau, int64_t chunk, int nchunk, int capc,
                                                         double *__restrict__ nd, int *__restrict__ ni, int *__restrict__ ncnt) {
    __shared__ __attribute__((aligned(16))) double tile[64 * DMAX];
    __shared__ double pend_d[4][NNF_PEND * 64];
    __shared__ int pend_i[4][NNF_PEND * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t iq = (int64_t)blockIdx.x * 256 + tid;
    const bool qlive = iq < nq;
    const int64_t lo = (int64_t)blockIdx.y * chunk, hi = lo + chunk < nx ? lo + chunk : nx;
    double qv[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) qv[d] = (qlive && d < D) ? q[iq * D + d] : 0.0;
    const double tq = qlive ? tau[iq] : -1.0;
    double *mynd = nd + (iq * nchunk + blockIdx.y) * (int64_t)capc;
    int *myni = ni + (iq * nchunk + blockIdx.y) * (int64_t)capc;
    int cnt = 0, tot = 0;
    auto flush = [&]() {
        for (int p = 0; p < cnt; ++p)
            if (tot + p < capc) {
                mynd[tot + p] = pend_d[wave][p * 64 + lane];
                myni[tot + p] = pend_i[wave][p * 64 + lane];
            }
        tot += cnt;   // (beyond capc: counted, not stored -- the pick kernel sees the overflow)
        cnt = 0;
    };
    for (int64_t base = lo; base < hi; base += 64) {
        __syncthreads();
        {   // 64 candidates x DMAX doubles, staged by all 256 threads
            const int c = tid >> 2, d0 = (tid & 3) * (DMAX / 4);
            const int64_t jc = base + c;
#pragma unroll
            for (int d = 0; d < DMAX / 4; ++d) tile[c * DMAX + d0 + d] = (jc < hi && d0 + d < D) ? x[jc * D + d0 + d] : 0.0;
        }
        __syncthreads();
        const int cnt_c = (int)(hi - base < 64 ? hi - base : 64);
        for (int c = 0; c < cnt_c; c += 2) {   // two candidates per pass: independent chains
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const double d0 = tile[c * DMAX + d] - qv[d], d1 = tile[(c + 1) * DMAX + d] - qv[d];   // (c + 1 == cnt_c: a zero-padded row, never taken)
                s0 = fma(d0, d0, s0);
                s1 = fma(d1, d1, s1);
            }
            const bool t0 = s0 <= tq, t1 = c + 1 < cnt_c && s1 <= tq;
            if (__any(t0 || t1)) {
                if (t0) {
                    pend_d[wave][cnt * 64 + lane] = s0;
                    pend_i[wave][cnt * 64 + lane] = (int)(base + c);
                    ++cnt;
                }
                if (__any(cnt == NNF_PEND)) flush();
                if (t1) {
                    pend_d[wave][cnt * 64 + lane] = s1;
                    pend_i[wave][cnt * 64 + lane] = (int)(base + c + 1);
                    ++cnt;
                }
                if (__any(cnt == NNF_PEND)) flush();
            }
        }
    }
    flush();
    if (qlive) ncnt[iq * nchunk + blockIdx.y] = tot;
}

template <int DMAX, int K>
__global__ __launch_bounds__(64) void nn_pick_kernel(int64_t nq, int64_t nx, int D, const double *__restrict__ q, const double *__restrict__ x,
                                                     int nchunk, int capc, const double *__restrict__ nd, const int *__restrict__ ni,
                                                     const int *__restrict__ ncnt, int m_out, int64_t *__restrict__ out, int *__restrict__ slow) {
    const int lane = threadIdx.x;
    const int64_t iq = (int64_t)blockIdx.x * 64 + lane;
    const bool qlive = iq < nq;
    double ld[K];
    int li[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        ld[t] = INFINITY;
        li[t] = INT_MAX;
    }
    int total = 0;
    bool over = false;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = qlive ? ncnt[iq * nchunk + ch] : 0;
        over |= c0 > capc;
        const int cn = c0 < capc ? c0 : capc;
        total += cn;
        const double *mynd = nd + (iq * nchunk + ch) * (int64_t)capc;
        const int *myni = ni + (iq * nchunk + ch) * (int64_t)capc;
        int cmax = cn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(cmax, off, 64);
            cmax = o > cmax ? o : cmax;
        }
#pragma unroll 1
        for (int p = 0; p < cmax; ++p) {
            const bool have = p < cn;
            const double cd = have ? mynd[p] : INFINITY;
            const int ci = have ? myni[p] : INT_MAX;
            if (__any(cd < ld[K - 1] || (cd == ld[K - 1] && ci < li[K - 1]))) topk_insert<K>(ld, li, cd, ci);
        }
    }
    // the rare slow path: too few notes (the sampled bound was below the K-th nearest distance) or an overflowed region -- the lane
    // scans every candidate itself (same chain, same order)
    const int need = (int)(nx < m_out ? nx : m_out);
    const bool bad = qlive && (over || total < need);
    if (__any(bad)) {
        if (bad) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                ld[t] = INFINITY;
                li[t] = INT_MAX;
            }
            atomicAdd(slow, 1);
        }
        double qv[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) qv[d] = (qlive && d < D) ? q[iq * D + d] : 0.0;
#pragma unroll 1
        for (int64_t j = 0; j < nx; ++j) {
            double sacc = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const double df = (d < D ? x[j * D + d] : 0.0) - qv[d];
                sacc = fma(df, df, sacc);
            }
            const bool take = bad && sacc < ld[K - 1];   // (equal distance: the earlier index stays)
            if (__any(take)) topk_insert<K>(ld, li, take ? sacc : INFINITY, take ? (int)j : INT_MAX);
        }
    }
    if (!qlive) return;
    const int kk = (int)(nx < m_out ? nx : m_out);
#pragma unroll
    for (int t = 0; t < K; ++t)
        if (t < m_out) out[iq * m_out + t] = t < kk ? li[t] : -1;
}

// The filter pipeline's plan for (nq, nx, K): sample size, rank, chunking, region capacity; false when the shape does not suit it
struct NnfPlan {
    int64_t S, stride, chunk;
    int r, nchunk, capc;
};
static bool nnf_plan(int64_t nq, int64_t nx, int K, NnfPlan &p) {
    if (nq < 30000 || nx < 20000 || nx >= INT_MAX) return false;   // (measured: below ~25 000 queries the streaming kernel is as fast or faster)
    p.S = nx / 16 < 2048 ? 2048 : (nx / 16 > 8192 ? 8192 : nx / 16);
    p.stride = nx / p.S;
    const double lam = (double)K * (double)p.S / (double)nx;             // sample points expected among the K nearest
    const double rr = lam + 6.0 * sqrt(lam) + 6.0;                         // P(Poisson(lam) >= r) ~ 1e-8: the bound then holds K candidates
    if (rr > NNF_R) return false;
    p.r = (int)ceil(rr);
    const double en = (double)p.r * (double)nx / (double)p.S;            // candidates expected at or below tau, +- en / sqrt(r)
    // two chunks of candidates (the collect kernel is fast enough for 392 x 2 workgroups to fill the chip at 1e5 queries); more for few queries
    p.nchunk = nq >= 60000 ? 2 : (nq >= 25000 ? 4 : 8);
    p.chunk = ((nx + p.nchunk - 1) / p.nchunk + 63) / 64 * 64;
    p.nchunk = (int)((nx + p.chunk - 1) / p.chunk);
    const double per = en / p.nchunk * (1.0 + 5.0 / sqrt((double)p.r)) + 64.0;   // a region: its share at + 5 sigma of the rank's spread
    p.capc = ((int)per + 63) / 64 * 64;
    return (double)nq * p.nchunk * p.capc * 12.0 <= 1.5e9;                 // (scratch: <= 1.5 GB, else the streaming kernel)
}

template <int DMAX, int K>
static int launch_nn_filter(dgpamd_ctx *ctx, int64_t nq, int64_t nx, int D, const double *q, const double *x, int m_out, int64_t *out,
                            const NnfPlan &p) {
    const size_t nreg = (size_t)nq * p.nchunk;
    const size_t bytes = nreg * p.capc * 12 + nreg * 4 + (size_t)nq * 8 + 64;
    void *both = nullptr;
    int rc = ctx_scratch(ctx, 1, bytes, &both);
    if (rc) return rc;
    double *nd = reinterpret_cast<double *>(both);
    double *tau = nd + nreg * p.capc;
    int *ni = reinterpret_cast<int *>(tau + nq);
    int *ncnt = ni + nreg * p.capc;
    int *slow = ncnt + nreg;
    HIP_TRY(ctx, hipMemsetAsync(slow, 0, 4, ctx->stream));
    hipLaunchKernelGGL((nn_tau_kernel<DMAX>), dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, ctx->stream, nq, nx, D, q, x, p.S, p.stride, p.r, tau);
    hipLaunchKernelGGL((nn_collect_kernel<DMAX>), dim3((unsigned)((nq + 255) / 256), (unsigned)p.nchunk), dim3(256), 0, ctx->stream, nq, nx, D, q, x,
                       (const double *)tau, p.chunk, p.nchunk, p.capc, nd, ni, ncnt);
    hipLaunchKernelGGL((nn_pick_kernel<DMAX, K>), dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, ctx->stream, nq, nx, D, q, x, p.nchunk, p.capc,
                       (const double *)nd, (const int *)ni, (const int *)ncnt, m_out, out, slow);
    return DGPAMD_OK;
}

template <int DMAX>
static int launch_nn_stream_k(dgpamd_ctx *ctx, int64_t nq, int64_t nx, int D, const double *q, const double *x, int m_out,
                              int ordered, int64_t *out) {
    // the query form at prediction sizes: filter, then select (DGPAMD_NN_FILTER=0: the streaming kernel, for comparisons)
    const bool filter_on = !(getenv("DGPAMD_NN_FILTER") && getenv("DGPAMD_NN_FILTER")[0] == '0');   // (read per call: the tests flip it)
    NnfPlan plan;
    if (!ordered && filter_on && m_out > 26 && nnf_plan(nq, nx, m_out, plan)) {
        if (m_out <= 32) return launch_nn_filter<DMAX, 32>(ctx, nq, nx, D, q, x, m_out, out, plan);
        if (m_out <= 51) return launch_nn_filter<DMAX, 51>(ctx, nq, nx, D, q, x, m_out, out, plan);
        return launch_nn_filter<DMAX, 64>(ctx, nq, nx, D, q, x, m_out, out, plan);
    }
    if (m_out <= 16) return launch_nn_stream<DMAX, 16>(ctx, nq, nx, D, q, x, m_out, ordered, out);
    if (m_out <= 26) return launch_nn_stream<DMAX, 26>(ctx, nq, nx, D, q, x, m_out, ordered, out);
    if (m_out <= 32) return launch_nn_stream<DMAX, 32>(ctx, nq, nx, D, q, x, m_out, ordered, out);
    if (m_out <= 51) return launch_nn_stream<DMAX, 51>(ctx, nq, nx, D, q, x, m_out, ordered, out);
    return launch_nn_stream<DMAX, 64>(ctx, nq, nx, D, q, x, m_out, ordered, out);
}

#define NN_BIG_MIN 4096   // candidate sets from this size on take the store-once path
static int launch_nn(dgpamd_ctx *ctx, int64_t nq, int64_t nx, int D, const double *q, const double *x, int m_out, int ordered,
                     int64_t *out) {
    if (nx < NN_BIG_MIN) {
        size_t shm = (D + 4) * sizeof(double) + (4 + (size_t)m_out) * sizeof(int64_t);
        hipLaunchKernelGGL(nn_select_kernel, dim3((unsigned)nq), dim3(256), shm, ctx->stream, nq, nx, D, q, x, m_out, ordered, out);
        return DGPAMD_OK;
    }
    // measured crossovers (tools/gpu_nn_bench.py): the streaming kernels win the ordered search from n ~ 12 000 on (5.8x at
    // n = 50 000) and the query form from ~6000 queries on (1.9x at 10 000, 2.6x at 20 000 queries against 50 000 points; their
    // floor is one cold-started chunk scan, ~4 ms).
    // DGPAMD_NN_STORE_ONCE = 1 / 2 forces the store-once / the streaming kernels (the tests compare the two).
    const char *env = getenv("DGPAMD_NN_STORE_ONCE");
    const int force = env ? atoi(env) : 0;
    const bool pays = ordered ? nx >= 12000 : (nq >= 6000 && nx >= 20000);
    if (D <= 16 && m_out <= 64 && nx < INT_MAX && force != 1 && (pays || force == 2))
        return D <= 8 ? launch_nn_stream_k<8>(ctx, nq, nx, D, q, x, m_out, ordered, out)
                      : launch_nn_stream_k<16>(ctx, nq, nx, D, q, x, m_out, ordered, out);
    const unsigned grid = (unsigned)(nq < 2 * (int64_t)ctx->num_cu * 2 ? nq : 2 * (int64_t)ctx->num_cu * 2);
    double *scratch = nullptr;
    HIP_TRY(ctx, hipMallocAsync((void **)&scratch, (size_t)grid * nx * sizeof(double), ctx->stream));
    size_t shm = (D + 8 + NN_CMAX) * sizeof(double) + (NN_CMAX + 4 + (size_t)m_out) * sizeof(int64_t) + (NN_NB + 256 + 4) * sizeof(int);
    if (shm > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)nn_select_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(nn_select_big_kernel, dim3(grid), dim3(256), shm, ctx->stream, nq, nx, D, q, x, m_out, ordered, out, scratch);
    HIP_TRY(ctx, hipFreeAsync(scratch, ctx->stream));
    return DGPAMD_OK;
}

extern "C" int dgpamd_nn_ordered(dgpamd_ctx *ctx, int64_t n, int D, const double *x, int m, int64_t *NNarray) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || D <= 0 || !x || !NNarray || m < 0) BAD_ARG(ctx, "bad arguments");
    if (m > n - 1) m = (int)(n - 1);
    int rc = launch_nn(ctx, n, n, D, x, x, m + 1, 1, NNarray);
    if (rc) return rc;
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

__global__ void nn_cyclic_kernel(int64_t M, int m, int64_t *NN) {   // vecchia.py:23-26 (m == n shortcut)
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < M * m) NN[e] = ((e % m) + (e / m)) % m;
}

extern "C" int dgpamd_nn_query(dgpamd_ctx *ctx, int64_t M, int64_t n, int D, const double *q, const double *x, int m,
                               int64_t *NN) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (M <= 0 || n <= 0 || D <= 0 || !q || !x || !NN || m <= 0) BAD_ARG(ctx, "bad arguments");
    if (m >= n) {
        m = (int)n;
        hipLaunchKernelGGL(nn_cyclic_kernel, dim3((unsigned)((M * m + 255) / 256)), dim3(256), 0, ctx->stream, M, m, NN);
    } else {
        int rc = launch_nn(ctx, M, n, D, q, x, m, 0, NN);
        if (rc) return rc;
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// a19-a21  per-row kernels on the ordered data
// ---------------------------------------------------------------------------
enum { V_LLIK = 0, V_NLLIK = 1, V_LMAT = 2 };

struct VRowArgs {
    VParams vp;
    int64_t n;
    int m;            // NNarray has m+1 columns
    const double *X, *y, *nugget_diag;
    const int64_t *NN;
    int nugget_est, P;
    double *partial;  // [batch][n][2 + 2P] (LLIK: [batch][n][2])
    double *Lmat;     // [n][m+1]
    int64_t x_stride; // doubles between the input sets of a batch (blockIdx.y); LLIK only
    const int32_t *pred;   // null, or a device word: the launch does nothing when it is non-zero (dgpamd_ess_queue)
    long long *trace = nullptr;   // diagnostics (dgpamd_debug_trace): shader-clock stamps of the first 64 row blocks' phases, 16 words each
};
#define VR4_STAMP(slot)                                                                                               \
    do {                                                                                                              \
        if (a.trace && rowblk < 64 && by == 0 && threadIdx.x == 0) a.trace[rowblk * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

// gather the row's conditioning block (ascending index, self last) into LDS; returns its size
__device__ int gather_block(const int64_t *NNrow, int mp1, int *idx, int lane) {
    int b = 0;
    for (int a = 0; a < mp1; ++a) b += (NNrow[a] >= 0);
    for (int a = lane; a < b; a += VW) idx[a] = (int)NNrow[b - 1 - a];   // idx[idx>=0][::-1]
    return b;
}

template <int KIND, int MODE>
__global__ __launch_bounds__(VW) void vecchia_row_kernel(VRowArgs a) {
    extern __shared__ double lds[];
    if (a.pred && *a.pred) return;
    const int mp1 = a.m + 1, D = a.vp.D, lda = mp1 + 2;
    double *A = lds;                         // [(mp1+1)][lda]  block + one right-hand-side row
    double *xs = A + (mp1 + 1) * lda;        // [mp1][D] scaled inputs
    double *V = xs + mp1 * D;                // [2][lda]  u = L^-T e_last, alpha = L^-T w
    int *idx = reinterpret_cast<int *>(V + 2 * lda);
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    const int b = gather_block(a.NN + i * mp1, mp1, idx, lane);
    __syncthreads();
    for (int e = lane; e < b * D; e += VW) {
        int r = e / D, d = e - r * D;
        xs[e] = a.X[(int64_t)idx[r] * D + d] * a.vp.inv_len[d];
    }
    __syncthreads();
    for (int e = lane; e < b * (b + 1) / 2; e += VW) {   // (the lower triangle only: no idle lanes)
        int r, c;
        tri_decode(e, r, c);
        double v;
        if (r == c)
            v = 1.0 + a.vp.nugget * (MODE == V_LMAT ? 1.0 : a.nugget_diag[idx[r]]);
        else
            v = corr_pts<KIND>(xs + r * D, xs + c * D, D);
        A[r * lda + c] = v;
    }
    const int rows = (MODE == V_LMAT) ? b : b + 1;
    if (MODE != V_LMAT)
        for (int c = lane; c <= b; c += VW) A[b * lda + c] = c < b ? a.y[idx[c]] : 0.0;
    lds_chol(A, lda, rows, b, lane);

    if (MODE == V_LLIK) {
        if (lane == 0) {
            const double wl = A[b * lda + b - 1], ll = A[(b - 1) * lda + b - 1];
            a.partial[i * 2] = wl * wl;                 // (L^-1 y)_last^2      vecchia.py:177
            a.partial[i * 2 + 1] = 2.0 * log(fabs(ll)); // 2 log L_last,last   vecchia.py:178
        }
        return;
    }
    // u = L^-T e_last (and alpha = L^-T w for the gradient)
    for (int c = lane; c < b; c += VW) {
        V[c] = (c == b - 1) ? 1.0 : 0.0;
        if (MODE == V_NLLIK) V[lda + c] = A[b * lda + c];
    }
    lds_backsolve_T(A, lda, b, V, lda, MODE == V_NLLIK ? 2 : 1, lane);
    if (MODE == V_LMAT) {
        for (int c = lane; c < mp1; c += VW) a.Lmat[i * mp1 + c] = c < b ? V[b - 1 - c] : 0.0;   // reversed, self first
        return;
    }
    // vecchia.py:216-219 restated: t_last = u^T dK u ; s = alpha^T dK u
    //   dquad_k = 2 s w_last - t_last w_last^2 ; dlogdet_k = t_last
    const int P = a.P, npl = (a.vp.nlen == 1) ? 1 : D;
    const double wl = A[b * lda + b - 1];
    double *out = a.partial + i * (2 + 2 * P);
    if (lane == 0) {
        out[0] = wl * wl;
        out[1] = 2.0 * log(fabs(A[(b - 1) * lda + b - 1]));
    }
    const double *u = V, *al = V + lda;
    for (int k = 0; k < npl; ++k) {
        double tl = 0.0, s = 0.0;
        for (int e = lane; e < b * (b - 1) / 2; e += VW) {   // (strictly lower triangle: row r + 1, column c of the decoded pair)
            int r, c;
            tri_decode(e, r, c);
            ++r;
            double kv = corr_pts<KIND>(xs + r * D, xs + c * D, D), cf = 0.0;
            if (a.vp.nlen == 1)
                for (int d = 0; d < D; ++d) cf += dcoef_v<KIND>(xs[r * D + d] - xs[c * D + d]);
            else
                cf = dcoef_v<KIND>(xs[r * D + k] - xs[c * D + k]);
            const double dk = cf * kv;
            tl = fma(2.0 * dk, u[r] * u[c], tl);
            s = fma(dk, al[r] * u[c] + al[c] * u[r], s);
        }
        tl = wsum(tl);
        s = wsum(s);
        if (lane == 0) {
            out[2 + k] = 2.0 * s * wl - tl * wl * wl;
            out[2 + P + k] = tl;
        }
    }
    if (a.nugget_est) {   // dK/dlog eta = diag(nugget * nugget_diag)   vecchia.py:329-332
        double tl = 0.0, s = 0.0;
        for (int r = lane; r < b; r += VW) {
            const double dk = a.vp.nugget * a.nugget_diag[idx[r]];
            tl = fma(dk, u[r] * u[r], tl);
            s = fma(dk, al[r] * u[r], s);
        }
        tl = wsum(tl);
        s = wsum(s);
        if (lane == 0) {
            out[2 + npl] = 2.0 * s * wl - tl * wl * wl;
            out[2 + P + npl] = tl;
        }
    }
}

// ---------------------------------------------------------------------------
// The same three per-row computations for conditioning sets of at most 31 points (m <= 30: the default m = 25), held in
// REGISTERS: FOUR rows of the likelihood per wave, one 16-lane DPP row each.  Lane t of a group owns rows t and 16 + t of
// the (m+2) x (m+1) block (the right-hand side rides as row m+1).  Column j of the factor is normalised in place and
// every trailing entry takes  a[r][c] -= l[r] * l[c]  in ONE v_fmac_f64_dpp whose first operand is read from lane
// c mod 16 of the group (row_newbcast) -- no LDS round trips, no shuffles and no barriers inside the factorisation.
// (The LDS version above spends its time on three barriers and a dependent LDS read-modify-write chain per column with
// 40 % of the lanes idle; a version with 32-wide ds_bpermute shuffles was bound by the LDS pipe: 2.9x slower than this.)
// The block itself is assembled with its m(m+1)/2 entries spread evenly over the group and handed over through LDS
// once.  Sets shorter than the compiled size (the first m rows; sizes between the compiled ones) are padded IN FRONT
// with identity rows, so that "self" is always the last slot and the loop bounds are compile-time constants.
// ---------------------------------------------------------------------------
#define VR_MAXB 31

__device__ __forceinline__ void tri_decode_small(int t, int &r, int &c) {   // t < 2^16: float is plenty
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((b + 1) * (b + 2) / 2 <= t) ++b;
    while (b * (b + 1) / 2 > t) --b;
    r = b;
    c = t - b * (b + 1) / 2;
}

__device__ __forceinline__ double hsum16(double v) {   // sum over the 16 lanes of a group, in every lane of it
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}

// the same with four DPP rotations of the 16-lane row (every lane pairs the same values: identical sums in all lanes)
__device__ __forceinline__ double row_allreduce_sum(double v) {
#define ROR_ADD(CTRL)                                                                                 \
    {                                                                                                 \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);      \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);      \
        v += __hiloint2double(hi, lo);                                                                \
    }
    ROR_ADD(0x128) ROR_ADD(0x124) ROR_ADD(0x122) ROR_ADD(0x121)
#undef ROR_ADD
    return v;
}

// 1/sqrt(d) and sqrt(d): hardware estimate + two Newton rounds + one correction of the root (a few ulp)
__device__ __forceinline__ void rsqrt_sqrt(double d, double &inv, double &sd) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    double r = d * y;
    r = fma(0.5 * y, fma(-r, r, d), r);
    inv = y;
    sd = r;
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// value of lane C of the caller's 16-lane group.  s_nop 4: a DPP source written by the previous VALU needs two wait states
// and an EXEC written by the previous SALU (the end of a divergent region: the compiler cannot see into the asm) five.
template <int C>
__device__ __forceinline__ double group_bcast(const double &v) {
    double out;
    asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(v), "i"(C));
    return out;
}
// acc += (lane C's src) * mul
template <int C>
__device__ __forceinline__ void fmac_bcast(double &acc, const double &src, const double &mul) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "i"(C));
}

// Row stride of the scaled inputs in LDS: the dimensions padded with zeros to a multiple of eight (the pair loop reads whole chunks of
// eight: no index clamps, no selects -- two of the four VALU instructions per pair and dimension), + 2: rows stay 16-byte aligned
// (ds_read_b128) and the sixteen rows a group reads at once fall on disjoint bank quadruples (stride 10: banks 0, 20, 40, 60, 16, ...;
// stride 18: 0, 36, 8, 44, ...).
#define VR4_DP(D) ((((D) + 7) & ~7) + 2)
static size_t vrow4_lds(int BS, int D, bool grad, bool solves) {
    const size_t asz = (size_t)(BS + 1) * (BS + 2) / 2 + (grad ? (size_t)BS * (BS - 1) / 2 : 0);
    return 4 * (asz + (size_t)BS * VR4_DP(D) + (solves ? 2 * 32 : 0)) * sizeof(double);   // (llik: 19.9 KB at m = 25, d = 8: 8 waves per CU)
}

// four rows (row block rb) of input set `by`: the body of vecchia_row4_kernel
template <int KIND, int MODE, int BS>
__device__ __forceinline__ void vrow4_body(const VRowArgs &a, double *lds, const int64_t rowblk, const int by) {
    constexpr int rows = BS + 1;                      // block rows, then the right-hand side as row BS
    constexpr int T2 = BS * (BS - 1) / 2;
    constexpr int asz = rows * (rows + 1) / 2 + (MODE == V_NLLIK ? T2 : 0);
    constexpr int NBB = BS > 16 ? BS : 16;            // columns kept for the second row of a lane
#define AT(r, c) ((r) * ((r) + 1) / 2 + (c))
    const int mp1 = a.m + 1, D = a.vp.D, DP = VR4_DP(D);
    const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
    double *A = lds + (size_t)g * asz;                                   // packed lower triangle (+ K itself for the gradient)
    double *Kp = A + rows * (rows + 1) / 2;
    double *xs = lds + (size_t)4 * asz + (size_t)g * BS * DP;             // [BS][D] scaled inputs
    double *V = lds + (size_t)4 * asz + (size_t)4 * BS * DP + (size_t)g * 64;   // [2][32] u, alpha
    const int64_t i = rowblk * 4 + g;
    const bool live = i < a.n;
    const double *X = a.X + (MODE == V_LLIK ? (int64_t)by * a.x_stride : 0);
    double *partial = a.partial;
    if (MODE == V_LLIK) partial += (int64_t)by * a.n * 2;

    // conditioning set: the valid entries of the row come first; slot R <- entry b-1-(R-pad)  (ascending, self last).
    // One load of the row, the reversal by shuffle; then every slot fetches its own point (inputs, output, nugget
    // weight) with all loads in flight together: two memory latencies in all before the arithmetic starts.
    VR4_STAMP(0);
    const int nn0 = (live && t < mp1) ? (int)a.NN[i * mp1 + t] : -1;
    const int nn1 = (live && 16 + t < mp1) ? (int)a.NN[i * mp1 + 16 + t] : -1;
    const unsigned m0 = (unsigned)(__ballot(nn0 >= 0) >> (16 * g)) & 0xffffu, m1 = (unsigned)(__ballot(nn1 >= 0) >> (16 * g)) & 0xffffu;
    const int b = __popc(m0) + __popc(m1);
    const int pad = BS - b;
    int my[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int R = 16 * h + t, src = b - 1 - (R - pad);
        const int g0 = __shfl(nn0, src & 15, 16), g1 = __shfl(nn1, src & 15, 16);
        my[h] = (R >= pad && R < BS) ? ((src & 16) ? g1 : g0) : -1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int R = 16 * h + t;
        double yv = 0.0, ndv = 1.0;
        if (MODE != V_LMAT && my[h] >= 0) {
            yv = a.y[my[h]];
            ndv = a.nugget_diag[my[h]];
        }
        const double *xrow = X + (int64_t)(my[h] >= 0 ? my[h] : 0) * D;
        for (int d0 = 0; d0 < D; d0 += 8) {
            double x8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x8[q] = (my[h] >= 0 && d0 + q < D) ? xrow[d0 + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (R < BS) xs[R * DP + d0 + q] = d0 + q < D ? x8[q] * a.vp.inv_len[d0 + q] : 0.0;   // (zeros past D: see VR4_DP)
        }
        if (R < BS) {
            A[AT(R, R)] = my[h] >= 0 ? 1.0 + a.vp.nugget * ndv : 1.0;
            if (MODE != V_LMAT) A[AT(BS, R)] = yv;
        }
    }
    VR4_STAMP(1);
    __syncthreads();
    VR4_STAMP(2);
    // Strictly lower entries in 2 x 2 tiles: rows (2i, 2i + 1) against columns (2j, 2j + 1), j <= i, one tile per lane and pass -- FOUR
    // points read from LDS for four entries.  (Two entries per lane and pass took four points as well: with the selects gone the
    // pair loop ran at the LDS's bandwidth, 2 KB per wave, pass and dimension.)  The diagonal tiles hold one entry below the
    // diagonal; their other three results, and those of the phantom row / column of an odd block size, are not stored.
    {
        constexpr int NP = (BS + 1) / 2, NT = NP * (NP + 1) / 2;
        const bool iso = MODE == V_NLLIK && a.vp.nlen == 1;
        // tile e = the strictly-lower pair (i + 1, j) of an (NP + 1) x (NP + 1) matrix, j <= i: decoded once, then advanced by sixteen
        // entries per pass with integer arithmetic (a table in memory cost a dependent load per pass: ~1000 cycles of a 12 000-cycle loop)
        int trow, tcol;
        tri_decode_small(t, trow, tcol);
        ++trow;
        for (int e = t; e < NT; e += 16) {
            const int ti = trow - 1, tj = tcol;
            tcol += 16;
            while (tcol >= trow) {
                tcol -= trow;
                ++trow;
            }
            const int r0 = 2 * ti, r1 = 2 * ti + 1, c0 = 2 * tj, c1 = 2 * tj + 1;
            const double *pr0 = xs + r0 * DP, *pr1 = xs + (r1 < BS ? r1 : BS - 1) * DP;
            const double *pc0 = xs + c0 * DP, *pc1 = xs + (c1 < BS ? c1 : BS - 1) * DP;
            double sa[4] = {0.0, 0.0, 0.0, 0.0}, pa[4] = {1.0, 1.0, 1.0, 1.0}, cfa[4] = {0.0, 0.0, 0.0, 0.0};   // (r0,c0) (r0,c1) (r1,c0) (r1,c1)
            for (int d0 = 0; d0 < D; d0 += 8) {
                double2 x0[4], x1[4], y0[4], y1[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {   // (16-byte reads; the columns past D hold zeros: 0 - 0, neutral for both kernels)
                    x0[h] = reinterpret_cast<const double2 *>(pr0 + d0)[h];
                    x1[h] = reinterpret_cast<const double2 *>(pr1 + d0)[h];
                    y0[h] = reinterpret_cast<const double2 *>(pc0 + d0)[h];
                    y1[h] = reinterpret_cast<const double2 *>(pc1 + d0)[h];
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const double u0 = (h & 1) ? x0[h >> 1].y : x0[h >> 1].x, u1 = (h & 1) ? x1[h >> 1].y : x1[h >> 1].x;
                    const double v0 = (h & 1) ? y0[h >> 1].y : y0[h >> 1].x, v1 = (h & 1) ? y1[h >> 1].y : y1[h >> 1].x;
                    const double f[4] = {u0 - v0, u0 - v1, u1 - v0, u1 - v1};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (KIND == DGPAMD_SEXP) {
                            corr_accum_sexp(f[w], sa[w]);
                        } else if (!iso) {
                            corr_accum_matern(f[w], pa[w], sa[w]);
                        } else {
                            // one shared lengthscale: dK/dlog(l) = K sum_d n_d / g_d with g_d = 1 + sqrt5 r + 5/3 r^2 the dimension's
                            // factor of K and n_d = (5/3 r^2)(1 + sqrt5 r).  The sum is carried as a fraction over the running product
                            // of the g_d -- which K needs anyway -- so that NO reciprocal is taken: N <- N g_d + n_d G, G <- G g_d, and
                            // at the end dK = exp(-sqrt5 s) N  (K = G exp(-sqrt5 s), the G cancels).  A reciprocal per pair and
                            // dimension (quarter-rate seed + two Newton rounds) was a third of this loop's issue cycles.
                            const double r = fabs(f[w]);
                            const double gk = fma(r, fma(r, 5.0 / 3.0, SQRT5), 1.0);   // (corr_accum_matern's factor: K holds the same bits in every mode)
                            const double nd = ((5.0 / 3.0) * r * r) * fma(r, SQRT5, 1.0);
                            cfa[w] = fma(nd, pa[w], cfa[w] * gk);
                            pa[w] *= gk;
                            sa[w] += r;
                        }
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int r = (w & 2) ? r1 : r0, c = (w & 1) ? c1 : c0;
                if (iso && KIND == DGPAMD_SEXP) cfa[w] = 2.0 * sa[w];   // dcoef = 2 df^2
                // exp_negated: the exponential from full-rate instructions only (the library's exp spends half of its issue cycles on
                // three quarter-rate ones); the three modes of this kernel -- I-step sums, M-step sums + gradients, sparse-factor rows
                // -- share it
                // (the squared-exponential instance gains 3-4 % from the pinned form; in the Matern instances it costs as much, SGPR pressure)
                //  and the sparse-factor mode of the squared exponential loses 4 %)
                const double ex = KIND == DGPAMD_SEXP ? (MODE == V_LMAT ? exp_negated(sa[w]) : exp_negated_v3(sa[w])) : exp_negated(SQRT5 * sa[w]);
                double kv = (KIND == DGPAMD_SEXP) ? ex : pa[w] * ex;
                if (c < pad) kv = 0.0;   // pads come first
                if (r < BS && c < r) {
                    const int el = r * (r - 1) / 2 + c;
                    A[el + r] = kv;   // AT(r, c) = r (r - 1) / 2 + c + r
                    // kept for the derivative sums: dK itself with one shared lengthscale (Matern: the numerator of the fraction
                    // above times the exponential), else the correlation
                    if (MODE == V_NLLIK)
                        Kp[el] = !iso ? kv : (KIND == DGPAMD_SEXP ? cfa[w] * kv : (c < pad ? 0.0 : cfa[w] * ex));
                }
            }
        }
    }
    VR4_STAMP(3);
    __syncthreads();
    VR4_STAMP(4);

    double ra[16], rb[NBB];   // rows t and 16 + t
    constexpr int LROWS = MODE == V_LMAT ? BS : BS + 1;   // (no right-hand-side row in the sparse-factor mode: nothing wrote it)
#pragma unroll
    for (int c = 0; c < 16; ++c) ra[c] = (c <= t && c < BS && t < LROWS) ? A[AT(t, c)] : 0.0;
#pragma unroll
    for (int c = 0; c < NBB; ++c) rb[c] = (c < BS && 16 + t < LROWS && c <= 16 + t) ? A[AT(16 + t, c < BS ? c : 0)] : 0.0;

    double sd_last = 1.0, w_last = 0.0;
    VR4_STAMP(5);
    static_for<BS>([&](auto J) {
        constexpr int j = J;
        double d = j < 16 ? group_bcast<(j & 15)>(ra[j & 15]) : group_bcast<(j & 15)>(rb[j < NBB ? j : 0]);
        if (!(d > 0.0)) d = 1.0;
        double inv, sd;
        rsqrt_sqrt(d, inv, sd);
        double la = 0.0, lb, nla = 0.0, nlb;
        if constexpr (j < 16) {
            la = ra[j] * inv;   // (lane j: d * inv = the diagonal of the factor)
            ra[j] = la;
            nla = -la;
        }
        lb = rb[j] * inv;
        rb[j] = lb;
        nlb = -lb;
        asm volatile("s_nop 4" : "+v"(nla), "+v"(nlb));   // (DPP sources just written; see group_bcast)
        static_for<BS>([&](auto Cc) {
            constexpr int c = Cc;
            if constexpr (c > j) {
                if constexpr (c < 16) {
                    fmac_bcast<(c & 15)>(ra[c & 15], nla, la);
                    if constexpr (BS >= 16) fmac_bcast<(c & 15)>(rb[c], nla, lb);
                } else {
                    fmac_bcast<(c & 15)>(rb[c], nlb, lb);
                }
            }
        });
        if constexpr (j == BS - 1) {
            sd_last = sd;
            w_last = BS < 16 ? la : lb;   // (in the right-hand-side row's lane: (L^-1 y)_last)
        }
    });
    constexpr int WL = BS & 15;   // lane of the right-hand-side row BS (second row of the lane when BS >= 16)
    VR4_STAMP(6);

    if (MODE == V_LLIK) {
        if (t == WL && live) {
            partial[i * 2] = w_last * w_last;            // vecchia.py:177
            partial[i * 2 + 1] = 2.0 * log(sd_last);     // vecchia.py:178
        }
        return;
    }
    // x <- L^-T x for e_last (u) and for w (alpha), in registers.  As soon as x_r is known the lane that owns row r adds
    // L[r][c] x_r to its partial sums of the columns c < r; when column c's turn comes the sum over the group's 16 lanes
    // (four DPP rotations) closes it.  No LDS, no barrier; pads decouple (identity rows in front).
    {
        // (entries above the diagonal are by-products of the factorisation, possibly not finite: they meet zeros below)
#pragma unroll
        for (int c = 0; c < 16; ++c) ra[c] = c <= t ? ra[c] : 0.0;
#pragma unroll
        for (int c = 0; c < NBB; ++c) rb[c] = c <= 16 + t ? rb[c] : 0.0;
        double accu[BS], acca[MODE == V_NLLIK ? BS : 1];
#pragma unroll
        for (int c = 0; c < BS; ++c) accu[c] = 0.0;
#pragma unroll
        for (int c = 0; c < (MODE == V_NLLIK ? BS : 1); ++c) acca[c] = 0.0;
        double u0 = 0.0, u1 = 0.0, a0 = 0.0, a1 = 0.0;   // x of rows t and 16 + t
        static_for<BS>([&](auto RR) {
            constexpr int r = BS - 1 - RR, owner = r & 15;
            constexpr bool sb = r >= 16;
            double diag;
            if constexpr (sb) diag = group_bcast<owner>(rb[r]); else diag = group_bcast<owner>(ra[r & 15]);
            double inv = __builtin_amdgcn_rcp(diag);
            inv = fma(inv, fma(-diag, inv, 1.0), inv);
            inv = fma(inv, fma(-diag, inv, 1.0), inv);
            const double ur = ((r == BS - 1 ? 1.0 : 0.0) - row_allreduce_sum(accu[r])) * inv;
            const double um = (t == owner) ? ur : 0.0;
            if constexpr (sb) u1 += um; else u0 += um;
            double am = 0.0;
            if constexpr (MODE == V_NLLIK) {
                double wr;
                if constexpr (BS < 16) wr = group_bcast<WL>(ra[r & 15]); else wr = group_bcast<WL>(rb[r]);
                const double ar = (wr - row_allreduce_sum(acca[r])) * inv;
                am = (t == owner) ? ar : 0.0;
                if constexpr (sb) a1 += am; else a0 += am;
            }
            static_for<r>([&](auto Cc) {
                constexpr int c = Cc;
                double lrc;
                if constexpr (sb) lrc = rb[c]; else lrc = ra[c & 15];
                accu[c] = fma(lrc, um, accu[c]);
                if constexpr (MODE == V_NLLIK) acca[c] = fma(lrc, am, acca[c]);
            });
        });
        VR4_STAMP(7);
        V[t] = u0;
        if (MODE == V_NLLIK) V[32 + t] = a0;
        if (16 + t < BS) {
            V[16 + t] = u1;
            if (MODE == V_NLLIK) V[48 + t] = a1;
        }
    }
    __syncthreads();
    if (MODE == V_LMAT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int R = 16 * h + t;
            if (R < mp1 && live) a.Lmat[i * mp1 + R] = V[BS - 1 - R];   // reversed, self first; pads give the trailing zeros
        }
        return;
    }
    // vecchia.py:216-219 restated: t_last = u^T dK u ; s = alpha^T dK u
    const int P = a.P, npl = (a.vp.nlen == 1) ? 1 : D;
    const double wl = __shfl(w_last, WL, 16);
    double *out = partial + i * (2 + 2 * P);
    if (t == 0 && live) {
        out[0] = wl * wl;
        out[1] = 2.0 * log(sd_last);
    }
    const double *u = V, *al = V + 32;
    int gr0, gc0;   // the lane's first strictly-lower entry (r, c); advanced by sixteen entries per pass (see the pair loop)
    tri_decode_small(t, gr0, gc0);
    ++gr0;
    for (int k = 0; k < npl; ++k) {
        double tl = 0.0, sm = 0.0;
        int gr = gr0, gc = gc0;
        // (a fixed number of passes, the entries past the end clamped and weighted with zero, unrolled seven at a time: the five
        //  LDS reads of a pass depend on its (r, c) and, issued pass by pass, each pass waited ~450 cycles for them -- 10 000 of
        //  the 47 000 cycles of a wave)
        constexpr int NPASS = (T2 + 15) / 16;
#pragma unroll 7
        for (int p = 0; p < NPASS; ++p) {
            const int e = t + 16 * p;
            const bool in = e < T2;
            const int r = in ? gr : 1, c = in ? gc : 0, ec = in ? e : 0;
            gc += 16;
            while (gc >= gr) {
                gc -= gr;
                ++gr;
            }
            double dk = a.vp.nlen == 1 ? Kp[ec] : dcoef_v<KIND>(xs[r * DP + k] - xs[c * DP + k]) * Kp[ec];
            dk = in ? dk : 0.0;
            tl = fma(2.0 * dk, u[r] * u[c], tl);
            sm = fma(dk, al[r] * u[c] + al[c] * u[r], sm);
        }
        tl = row_allreduce_sum(tl);
        sm = row_allreduce_sum(sm);
        if (t == 0 && live) {
            out[2 + k] = 2.0 * sm * wl - tl * wl * wl;
            out[2 + P + k] = tl;
        }
    }
    VR4_STAMP(8);
    if (a.nugget_est) {   // dK/dlog eta = diag(nugget * nugget_diag)   vecchia.py:329-332
        double tl = 0.0, sm = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int R = 16 * h + t;
            if (R < BS && my[h] >= 0) {
                const double dk = a.vp.nugget * a.nugget_diag[my[h]];
                tl = fma(dk, u[R] * u[R], tl);
                sm = fma(dk, al[R] * u[R], sm);
            }
        }
        tl = row_allreduce_sum(tl);
        sm = row_allreduce_sum(sm);
        if (t == 0 && live) {
            out[2 + npl] = 2.0 * sm * wl - tl * wl * wl;
            out[2 + P + npl] = tl;
        }
    }
#undef AT
}

// A workgroup (one wave) walks row blocks rb = blockIdx.x, blockIdx.x + gridDim.x, ...: the grid is capped (launch_vrow4_nb)
// so that a launch the device-side ESS queue has predicated away costs a few thousand workgroup dispatches, not
// n / 4 x batch of them (at n = 50 000 the no-op launches of an I-step added up to 9 ms: ~7 ns per dispatched workgroup).
template <int KIND, int MODE, int BS>
__global__ __launch_bounds__(64) void vecchia_row4_kernel(VRowArgs a) {
    extern __shared__ double lds[];
    if (a.pred && *a.pred) return;
    const int64_t nrb = (a.n + 3) / 4;
    for (int64_t q = blockIdx.x; q < nrb; q += gridDim.x) {
        vrow4_body<KIND, MODE, BS>(a, lds, q, (int)blockIdx.y);
        __syncthreads();   // (the next row block reuses the LDS)
    }
}

#ifndef VR4_GRID
#define VR4_GRID 2048   // row-block workgroups per input set (each walks n / 4 / 2048 row blocks)
#endif
template <int KIND, int MODE, int BS>
static int launch_vrow4_nb(dgpamd_ctx *ctx, VRowArgs &a, int batch) {
    const size_t shm = vrow4_lds(BS, a.vp.D, MODE == V_NLLIK, MODE != V_LLIK);
    int rc = set_lds(ctx, (const void *)vecchia_row4_kernel<KIND, MODE, BS>, shm);
    if (rc) return rc;
    const int64_t nrb = (a.n + 3) / 4;
    static const int64_t cap_env = getenv("DGPAMD_VR4_GRID") ? atoll(getenv("DGPAMD_VR4_GRID")) : -1;
    // the cap applies to the launches a device-side queue may predicate away (ctx->pred set); the others keep one
    // workgroup per row block, which the hardware balances better (llik x6 688 vs 721 us, nllik 333 vs 380 us at n = 50 000)
    const int64_t cap = cap_env >= 0 ? (cap_env == 0 ? nrb : cap_env) : (a.pred ? VR4_GRID : nrb);
    a.trace = ctx->trace;
    hipLaunchKernelGGL((vecchia_row4_kernel<KIND, MODE, BS>), dim3((unsigned)(nrb < cap ? nrb : cap), (unsigned)batch), dim3(64), shm,
                       ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

template <int KIND, int MODE>
static int launch_vrow4(dgpamd_ctx *ctx, VRowArgs &a, int batch) {
    const int nb = a.m + 1;   // compiled sizes: every fourth, and 26 (the default m = 25) exactly
    if (nb <= 4) return launch_vrow4_nb<KIND, MODE, 4>(ctx, a, batch);
    if (nb <= 8) return launch_vrow4_nb<KIND, MODE, 8>(ctx, a, batch);
    if (nb <= 12) return launch_vrow4_nb<KIND, MODE, 12>(ctx, a, batch);
    if (nb <= 16) return launch_vrow4_nb<KIND, MODE, 16>(ctx, a, batch);
    if (nb <= 20) return launch_vrow4_nb<KIND, MODE, 20>(ctx, a, batch);
    if (nb <= 24) return launch_vrow4_nb<KIND, MODE, 24>(ctx, a, batch);
    if (nb <= 26) return launch_vrow4_nb<KIND, MODE, 26>(ctx, a, batch);
    if (nb <= 28) return launch_vrow4_nb<KIND, MODE, 28>(ctx, a, batch);
    return launch_vrow4_nb<KIND, MODE, VR_MAXB>(ctx, a, batch);
}

// deterministic column sums of partial[batch][n][w] -> out[batch][w]  (grid: w x batch)
__global__ __launch_bounds__(1024) void colsum_kernel(const double *partial, int64_t n, int w, double *out) {
    __shared__ double sm[16];
    const int c = blockIdx.x, tid = threadIdx.x;
    partial += (int64_t)blockIdx.y * n * w;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int64_t r = tid;
    for (; r + 3072 < n; r += 4096) {   // four loads in flight per thread
        v0 += partial[r * w + c];
        v1 += partial[(r + 1024) * w + c];
        v2 += partial[(r + 2048) * w + c];
        v3 += partial[(r + 3072) * w + c];
    }
    for (; r < n; r += 1024) v0 += partial[r * w + c];
    double v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += sm[q];
        out[(int64_t)blockIdx.y * w + c] = t;
    }
}

static size_t vrow_lds(int m, int D) {
    const int mp1 = m + 1, lda = mp1 + 2;
    return ((size_t)(mp1 + 1) * lda + (size_t)mp1 * D + 2 * lda) * sizeof(double) + (size_t)mp1 * sizeof(int);
}

// Debugging aid (DGPAMD_POISON_LDS=1): fill the LDS of every CU with NaNs before a row launch, so that a read of LDS the
// kernel has not written shows up in the results whatever ran on the device before (tests/test_gpu_ops.py uses it).
__global__ __launch_bounds__(256) void lds_poison_kernel(double *sink) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = __builtin_nan("");
    __syncthreads();
    if (lds[(threadIdx.x * 31) & 8191] == 1.0) sink[0] = 1.0;   // (never true: keeps the stores alive)
}

static int maybe_poison_lds(dgpamd_ctx *ctx, const double *any_device_ptr) {
    const char *poison = getenv("DGPAMD_POISON_LDS");
    if (poison && atoi(poison)) {
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        hipLaunchKernelGGL(lds_poison_kernel, dim3(8 * ctx->num_cu), dim3(256), 65536, ctx->stream, (double *)any_device_ptr);
    }
    return DGPAMD_OK;
}

// Fill every CU's LDS with NaNs now (whatever the environment says): Engine._enter calls it before EVERY library call under
// DGPAMD_POISON_LDS=2, which puts the whole test suite under the same check.
extern "C" int dgpamd_debug_poison_lds(dgpamd_ctx *ctx) {
    if (!ctx) return DGPAMD_BAD_ARG;
    double *sink = nullptr;
    HIP_TRY(ctx, hipMallocAsync((void **)&sink, sizeof(double), ctx->stream));
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(lds_poison_kernel, dim3(8 * ctx->num_cu), dim3(256), 65536, ctx->stream, sink);
    HIP_TRY(ctx, hipFreeAsync(sink, ctx->stream));
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

template <int MODE>
static int launch_vrow(dgpamd_ctx *ctx, VRowArgs &a, int batch = 1) {
    {
        int prc = maybe_poison_lds(ctx, a.X);
        if (prc) return prc;
    }
    const char *env = getenv("DGPAMD_VECCHIA_LDS");   // (1: the LDS version for every size -- the tests compare the two)
    if (a.m + 1 <= VR_MAXB && !(env && atoi(env)))   // register-resident factorisation, two rows per wave
        return a.vp.kind == DGPAMD_SEXP ? launch_vrow4<DGPAMD_SEXP, MODE>(ctx, a, batch) : launch_vrow4<DGPAMD_MATERN25, MODE>(ctx, a, batch);
    const size_t shm = vrow_lds(a.m, a.vp.D);
    const void *fn = a.vp.kind == DGPAMD_SEXP ? (const void *)vecchia_row_kernel<DGPAMD_SEXP, MODE>
                                              : (const void *)vecchia_row_kernel<DGPAMD_MATERN25, MODE>;
    int rc = set_lds(ctx, fn, shm);
    if (rc) return rc;
    VRowArgs one = a;
    for (int bb = 0; bb < batch; ++bb) {   // (LDS version: one launch per input set)
        if (a.vp.kind == DGPAMD_SEXP)
            hipLaunchKernelGGL((vecchia_row_kernel<DGPAMD_SEXP, MODE>), dim3((unsigned)a.n), dim3(VW), shm, ctx->stream, one);
        else
            hipLaunchKernelGGL((vecchia_row_kernel<DGPAMD_MATERN25, MODE>), dim3((unsigned)a.n), dim3(VW), shm, ctx->stream, one);
        one.X += a.x_stride;
        one.partial += a.n * 2;
    }
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// scratch for the per-row partials lives behind the public outputs: the caller passes device buffers sized for the reduced
// result only; the context keeps one growable buffer for them (ctx_scratch).
static int with_partials(dgpamd_ctx *ctx, size_t bytes, double **p) {
    return ctx_scratch(ctx, 0, bytes, reinterpret_cast<void **>(p));   // (row kernel writes, column sums read: stream order)
}

int vecchia_llik_batch_into(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, int64_t x_stride, int batch,
                            const double *y, const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                            const double *nugget_diag, double *partial, double *out) {
    VRowArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, nugget);
    if (rc) return rc;
    a.n = n; a.m = m; a.X = X; a.y = y; a.nugget_diag = nugget_diag; a.NN = NNarray; a.nugget_est = 0; a.P = 0;
    a.Lmat = nullptr; a.x_stride = x_stride; a.partial = partial; a.pred = ctx->pred;
    rc = launch_vrow<V_LLIK>(ctx, a, batch);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3(2, (unsigned)batch), dim3(1024), 0, ctx->stream, (const double *)partial, n, 2, out);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_vecchia_llik_batch(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, int64_t x_stride,
                                         int batch, const double *y, const int64_t *NNarray, const double *length_h, int nlen,
                                         double nugget, const double *nugget_diag, double *out_llik) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || batch <= 0 || batch > 65535 || !X || !y || !NNarray || !nugget_diag || !out_llik)
        BAD_ARG(ctx, "bad arguments");
    VRowArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, nugget);
    if (rc) return rc;
    a.n = n; a.m = m; a.X = X; a.y = y; a.nugget_diag = nugget_diag; a.NN = NNarray; a.nugget_est = 0; a.P = 0;
    a.Lmat = nullptr; a.x_stride = x_stride; a.pred = nullptr;
    rc = with_partials(ctx, (size_t)batch * n * 2 * sizeof(double), &a.partial);
    if (rc) return rc;
    rc = launch_vrow<V_LLIK>(ctx, a, batch);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3(2, (unsigned)batch), dim3(1024), 0, ctx->stream, (const double *)a.partial, n, 2, out_llik);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_vecchia_llik(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, const double *y,
                                   const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                                   const double *nugget_diag, double *out_llik) {
    return dgpamd_vecchia_llik_batch(ctx, kind, n, D, m, X, 0, 1, y, NNarray, length_h, nlen, nugget, nugget_diag, out_llik);
}

extern "C" int dgpamd_vecchia_nllik(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X,
                                    const double *y, const int64_t *NNarray, const double *length_h, int nlen,
                                    double nugget, const double *nugget_diag, int nugget_est, double *out_nllik) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || !X || !y || !NNarray || !nugget_diag || !out_nllik) BAD_ARG(ctx, "bad arguments");
    VRowArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, nugget);
    if (rc) return rc;
    a.n = n; a.m = m; a.X = X; a.y = y; a.nugget_diag = nugget_diag; a.NN = NNarray; a.nugget_est = nugget_est ? 1 : 0;
    a.P = (nlen == 1 ? 1 : D) + a.nugget_est;
    a.Lmat = nullptr; a.x_stride = 0; a.pred = nullptr;
    const int w = 2 + 2 * a.P;
    rc = with_partials(ctx, (size_t)n * w * sizeof(double), &a.partial);
    if (rc) return rc;
    rc = launch_vrow<V_NLLIK>(ctx, a);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3(w), dim3(1024), 0, ctx->stream, (const double *)a.partial, n, w, out_nllik);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_vecchia_lmatrix(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X,
                                      const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                                      double *Lmat) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || !X || !NNarray || !Lmat) BAD_ARG(ctx, "bad arguments");
    VRowArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, nugget);
    if (rc) return rc;
    a.n = n; a.m = m; a.X = X; a.y = nullptr; a.nugget_diag = nullptr; a.NN = NNarray; a.nugget_est = 0; a.P = 0;
    a.partial = nullptr; a.Lmat = Lmat; a.x_stride = 0; a.pred = nullptr;
    return launch_vrow<V_LMAT>(ctx, a);
}

// ---------------------------------------------------------------------------
// Hetero likelihood, exact conditional posterior of the mean latent under Vecchia (imputation.py:143-160,
// vecchia.U_matrix :426-446, U_matrix_sp :599-610, Hetero.post_het_vecch likelihood_class.py:166-182).
// Row i of impNN (kernel_class.py:268-274) conditions the LATENT of ordered point i on, in the stacked vector
// [observations 0..n-1 ; latents n..2n-1], its own latent (n+i), its own observation (i) and its nearest other
// points (latent if earlier in the ordering, observation otherwise).  u_i = L_i^-T e_last of the block
// scale*corr + diag(gamma on observation entries + 1e-10) is column i of the sparse factor U; the reference
// assembles U (2n x n) with scipy.sparse and solves with U_l^T.  Here every row is emitted directly in the layout
// dgpamd_vecchia_spsolve consumes: Lrows[i] = [diagonal, latent-neighbour entries, 0..], NNl[i] = [i, their
// columns, 0..], and t_i = sum over the observation entries of u * y  (= (U_ol^T y)_i).
// ---------------------------------------------------------------------------
struct VHetArgs {
    VParams vp;
    int64_t n;
    int m;
    const double *X, *gamma, *y;
    const int64_t *NN;
    double scale;
    double *Lrows, *t;
    int64_t *NNl;
    int32_t *info;
};

template <int KIND>
__global__ __launch_bounds__(VW) void vecchia_het_rows_kernel(VHetArgs a) {
    extern __shared__ double lds[];
    const int mp1 = a.m + 1, D = a.vp.D, lda = mp1 + 2;
    double *A = lds;                         // [(mp1+1)][lda]
    double *xs = A + (mp1 + 1) * lda;        // [mp1][D] scaled inputs
    double *V = xs + mp1 * D;                // [2][lda]  u = L^-T e_last
    int *idx = reinterpret_cast<int *>(V + 2 * lda);
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x, n = a.n;
    const int b = gather_block(a.NN + i * mp1, mp1, idx, lane);   // reversed: own observation, own latent last
    __syncthreads();
    for (int e = lane; e < b * D; e += VW) {
        const int r = e / D, d = e - r * D;
        const int64_t p = idx[r] >= n ? idx[r] - n : idx[r];
        xs[e] = a.X[p * D + d] * a.vp.inv_len[d];
    }
    __syncthreads();
    for (int e = lane; e < b * (b + 1) / 2; e += VW) {
        int r, c;
        tri_decode(e, r, c);
        double v;
        if (r == c)
            v = a.scale + (idx[r] >= n ? 0.0 : a.gamma[idx[r]]) + 1e-10;
        else
            v = a.scale * corr_pts<KIND>(xs + r * D, xs + c * D, D);
        A[r * lda + c] = v;
    }
    const int bad = lds_chol(A, lda, b, b, lane);
    for (int c = lane; c < b; c += VW) V[c] = (c == b - 1) ? 1.0 : 0.0;
    lds_backsolve_T(A, lda, b, V, lda, 1, lane);
    if (lane == 0) {
        if (bad) atomicCAS(a.info, 0, (int)(i + 1));   // a block that is not positive definite (numpy raises LinAlgError)
        double *Lr = a.Lrows + i * mp1;
        int64_t *Nr = a.NNl + i * mp1;
        Lr[0] = V[b - 1];
        Nr[0] = i;
        int slot = 1;
        double t = 0.0;
        for (int c = 0; c < b - 1; ++c) {
            if (idx[c] >= n) {
                Lr[slot] = V[c];
                Nr[slot] = idx[c] - n;
                ++slot;
            } else {
                t = fma(V[c], a.y[idx[c]], t);
            }
        }
        for (; slot < mp1; ++slot) {
            Lr[slot] = 0.0;
            Nr[slot] = 0;
        }
        a.t[i] = t;
    }
}

extern "C" int dgpamd_vecchia_het_rows(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X,
                                       const int64_t *impNN, const double *length_h, int nlen, double scale,
                                       const double *gamma, const double *y, double *Lrows, int64_t *NNl, double *t,
                                       int32_t *info) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 1 || !X || !impNN || !gamma || !y || !Lrows || !NNl || !t || !info) BAD_ARG(ctx, "bad arguments");
    HIP_TRY(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
    VHetArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, 0.0);
    if (rc) return rc;
    a.n = n; a.m = m; a.X = X; a.gamma = gamma; a.y = y; a.NN = impNN; a.scale = scale; a.Lrows = Lrows; a.NNl = NNl; a.t = t; a.info = info;
    const size_t shm = vrow_lds(m, D);
    const void *fn = kind == DGPAMD_SEXP ? (const void *)vecchia_het_rows_kernel<DGPAMD_SEXP>
                                         : (const void *)vecchia_het_rows_kernel<DGPAMD_MATERN25>;
    rc = set_lds(ctx, fn, shm);
    if (rc) return rc;
    {
        int prc = maybe_poison_lds(ctx, a.X);
        if (prc) return prc;
    }
    if (kind == DGPAMD_SEXP)
        hipLaunchKernelGGL(vecchia_het_rows_kernel<DGPAMD_SEXP>, dim3((unsigned)n), dim3(VW), shm, ctx->stream, a);
    else
        hipLaunchKernelGGL(vecchia_het_rows_kernel<DGPAMD_MATERN25>, dim3((unsigned)n), dim3(VW), shm, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// forward_solve_sp (vecchia.py:111-120): x_i = (b_i - sum_j L[i,j] x[NN[i,j]]) / L[i,0], rows in order.
// One persistent 1024-thread workgroup walks the rows in windows; inside a window every row whose
// in-window dependencies are published computes in the same sweep (wavefront over the dependency DAG).
// ---------------------------------------------------------------------------
#define SPW 1024
// Batched form: workgroup p solves with matrix p / nrhs and right-hand side p (independent chains run side by side:
// the prior draws of all Vecchia nodes of a layer, for all sweeps of a sample() call, are ONE launch).
__global__ __launch_bounds__(SPW) void spsolve_kernel(int64_t n, int mp1, const double *L, const int64_t *NN,
                                                      double lscale, const double *lscales, int nrhs, const double *b,
                                                      double *x) {
    __shared__ double xs[SPW];
    __shared__ int rdy[SPW];
    const int tid = threadIdx.x;
    {
        const int64_t p = blockIdx.x, mat = p / nrhs;
        L += mat * n * mp1;
        NN += mat * n * mp1;
        b += p * n;
        x += p * n;
        if (lscales) lscale = lscales[mat];
    }
    constexpr int WMAX = 6;   // in-window dependencies a row keeps in registers (more: re-read from the arrays)
    for (int64_t base = 0; base < n; base += SPW) {
        const int64_t i = base + tid;
        const bool active = i < n;
        rdy[tid] = 0;
        double acc = 0.0;
        int cnt = 0, nw = 0;          // nw: in-window dependencies (all of them counted, the first WMAX kept)
        int wdep[WMAX];
        double wl[WMAX];
#pragma unroll
        for (int q = 0; q < WMAX; ++q) {
            wdep[q] = 0;
            wl[q] = 0.0;
        }
        if (active) {
            cnt = (int)(i + 1 < mp1 ? i + 1 : mp1);
            // dependencies finished in earlier windows, eight at a time: the index loads, then the eight x loads, are in
            // flight together (one at a time the two dependent memory latencies per entry made 50 us of a window's 180);
            // the sum runs in the order of the entries as before
            for (int j0 = 1; j0 < cnt; j0 += 8) {
                int64_t dep[8];
                double lv[8], xv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = j0 + q;
                    dep[q] = j < cnt ? NN[i * mp1 + j] : -1;
                    lv[q] = j < cnt ? L[i * mp1 + j] * lscale : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    xv[q] = (dep[q] >= 0 && dep[q] < base) ? ((const volatile double *)x)[dep[q]] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (dep[q] >= 0 && dep[q] < base) acc = fma(lv[q], xv[q], acc);
                    if (dep[q] >= base) {
#pragma unroll
                        for (int w = 0; w < WMAX; ++w)
                            if (w == nw) {
                                wdep[w] = (int)(dep[q] - base);
                                wl[w] = lv[q];
                            }
                        ++nw;
                    }
                }
            }
        }
        bool done = !active;
        __syncthreads();
        while (true) {
            bool fire = false;
            if (!done) {
                fire = true;
                if (nw <= WMAX) {
#pragma unroll
                    for (int w = 0; w < WMAX; ++w)
                        if (w < nw && !rdy[wdep[w]]) fire = false;
                } else {
                    for (int j = 1; j < cnt; ++j) {
                        const int64_t dep = NN[i * mp1 + j];
                        if (dep >= base && !rdy[dep - base]) {
                            fire = false;
                            break;
                        }
                    }
                }
            }
            double xi = 0.0;
            if (fire) {
                double s = acc;
                if (nw <= WMAX) {
#pragma unroll
                    for (int w = 0; w < WMAX; ++w)
                        if (w < nw) s = fma(wl[w], xs[wdep[w]], s);
                } else {
                    for (int j = 1; j < cnt; ++j) {
                        const int64_t dep = NN[i * mp1 + j];
                        if (dep >= base) s = fma(L[i * mp1 + j] * lscale, xs[dep - base], s);
                    }
                }
                xi = (b[i] - s) / (L[i * mp1] * lscale);
            }
            __syncthreads();
            if (fire) {
                xs[tid] = xi;
                rdy[tid] = 1;
                x[i] = xi;
                done = true;
            }
            if (__syncthreads_and(done)) break;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// The same substitution LEVEL-SCHEDULED.  Row i depends on rows NN[i][1..] only, all earlier in the ordering; its level is
// 1 + the highest level among them, and rows of one level are independent.  With a random ordering and m = 25 the
// dependency DAG of n = 50 000 points is a few hundred levels deep (tens to hundreds of rows each), against 50 000 rows
// walked in windows of 1024 above.  The schedule depends on the neighbour array only, so it is built once per array
// (dgpamd_vecchia_levels: when the ordering is refreshed, kernel_class.py:245-277 / dgp.py:1388) and reused by every draw:
// fmvn_sp for all nodes of a layer and all sweeps of imputer.sample (vecchia.py:133-140, imputation.py:54-63).
// Schedule of one matrix (int32): lev[n] | order[n] | ptr[n + 1] | cursor[n + 1] | nlev.
// ---------------------------------------------------------------------------
static inline size_t splevel_words(int64_t n) { return (size_t)(4 * n + 3); }

// levels of one matrix per workgroup: the window walk of spsolve_kernel on integers
__global__ __launch_bounds__(SPW) void splevel_kernel(int64_t n, int mp1, const int64_t *NN, int32_t *sched, int64_t words) {
    __shared__ int ls[SPW];
    __shared__ int rdy[SPW];
    __shared__ int top;
    const int tid = threadIdx.x;
    NN += (int64_t)blockIdx.x * n * mp1;
    int32_t *lev = sched + (int64_t)blockIdx.x * words;
    if (tid == 0) top = 0;
    for (int64_t base = 0; base < n; base += SPW) {
        const int64_t i = base + tid;
        const bool active = i < n;
        rdy[tid] = 0;
        int acc = -1;
        const int cnt = active ? (int)(i + 1 < mp1 ? i + 1 : mp1) : 0;
        for (int j = 1; j < cnt; ++j) {
            const int64_t dep = NN[i * mp1 + j];
            if (dep >= 0 && dep < base) acc = max(acc, ((const volatile int32_t *)lev)[dep]);
        }
        bool done = !active;
        __syncthreads();
        while (true) {
            bool fire = !done;
            int l = acc;
            if (fire)
                for (int j = 1; j < cnt; ++j) {
                    const int64_t dep = NN[i * mp1 + j];
                    if (dep >= base) {
                        if (!rdy[dep - base]) { fire = false; break; }
                        l = max(l, ls[dep - base]);
                    }
                }
            __syncthreads();
            if (fire) {
                ls[tid] = l + 1;
                rdy[tid] = 1;
                lev[i] = l + 1;
                atomicMax(&top, l + 1);
                done = true;
            }
            if (__syncthreads_and(done)) break;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) lev[4 * n + 2] = top + 1;   // nlev
}
// counting sort of the rows by level (the order inside a level does not matter: its rows are independent, every row's
// sum is formed the same way wherever it sits)
__global__ __launch_bounds__(1024) void splevel_sort_kernel(int64_t n, int32_t *sched, int64_t words) {
    const int tid = threadIdx.x;
    int32_t *lev = sched + (int64_t)blockIdx.x * words, *order = lev + n, *ptr = order + n, *cursor = ptr + n + 1;
    const int nl = lev[4 * n + 2];
    for (int64_t i = tid; i <= n; i += 1024) { ptr[i] = 0; cursor[i] = 0; }
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) atomicAdd(&ptr[lev[i] + 1], 1);   // counts, shifted by one
    __syncthreads();
    if (tid == 0) {   // (a few hundred levels: a serial scan)
        int run = 0;
        for (int l = 0; l <= nl; ++l) { run += ptr[l]; ptr[l] = run; }
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) {
        const int l = lev[i];
        order[ptr[l] + atomicAdd(&cursor[l], 1)] = (int32_t)i;
    }
}

// One workgroup per (matrix, chunk of <= SPL_R right-hand sides): level by level, SPL_LANES lanes per row (128 rows per
// pass of the 1024 threads); a lane takes the row's entries j = 1 + lane, 1 + lane + SPL_LANES, ... (up to SPL_DEPS of them
// kept in registers; wider conditioning sets re-read the arrays) for ALL right-hand sides of the chunk, the partial sums
// are closed by a fixed butterfly.  Three passes are in flight: the row index of pass p + 2 and the matrix entries of pass
// p + 1 are requested together with the x values of pass p, so a pass exposes ONE memory round trip; a barrier (after the
// stores have drained) closes every level.
#define SPL_R 1
#define SPL_LANES 8
#define SPL_DEPS 4
struct SplRow {
    int i, cnt;                  // row (-1: none), entries in its conditioning set
    int dep[SPL_DEPS];
    double lv[SPL_DEPS], diag, rhs;
};
__global__ __launch_bounds__(1024) void spsolve_level_kernel(int64_t n, int mp1, const double *L, const int64_t *NN, const double *lscales,
                                                             int nrhs, const double *b, double *x, const int32_t *sched, int64_t words) {
    const int tid = threadIdx.x, lane = tid & (SPL_LANES - 1), grp = tid / SPL_LANES;
    constexpr int RPP = 1024 / SPL_LANES;   // rows per pass
    const int chunks = (nrhs + SPL_R - 1) / SPL_R;
    const int mat = blockIdx.x / chunks, q0 = (blockIdx.x - mat * chunks) * SPL_R;
    const int R = nrhs - q0 < SPL_R ? nrhs - q0 : SPL_R;
    L += (int64_t)mat * n * mp1;
    NN += (int64_t)mat * n * mp1;
    b += ((int64_t)mat * nrhs + q0) * n;
    x += ((int64_t)mat * nrhs + q0) * n;
    const double lscale = lscales[mat];
    const int32_t *lev = sched + (int64_t)mat * words, *order = lev + n, *ptr = order + n;
    const int nl = lev[4 * n + 2];
    // the passes in order: (level, first row); `adv` moves one pass on
    struct Pos { int lv, rb, rend; };
    auto adv = [&](Pos p) {
        p.rb += RPP;
        while (p.lv < nl && p.rb >= p.rend) {
            ++p.lv;
            if (p.lv < nl) { p.rb = ptr[p.lv]; p.rend = ptr[p.lv + 1]; }
        }
        return p;
    };
    auto row_of = [&](const Pos &p) { return (p.lv < nl && p.rb + grp < p.rend) ? order[p.rb + grp] : -1; };
    auto entries = [&](int i) {
        SplRow d;
        d.i = i;
        d.cnt = 0;
        d.diag = 1.0;
        d.rhs = 0.0;
#pragma unroll
        for (int q = 0; q < SPL_DEPS; ++q) { d.dep[q] = -1; d.lv[q] = 0.0; }
        if (i >= 0) {
            d.cnt = (int)((int64_t)i + 1 < mp1 ? i + 1 : mp1);
#pragma unroll
            for (int q = 0; q < SPL_DEPS; ++q) {
                const int j = 1 + lane + q * SPL_LANES;
                if (j < d.cnt) {
                    d.dep[q] = (int)NN[(int64_t)i * mp1 + j];
                    d.lv[q] = L[(int64_t)i * mp1 + j];
                }
            }
            d.diag = L[(int64_t)i * mp1];
            if (lane < R) d.rhs = b[(int64_t)lane * n + i];
        }
        return d;
    };
    Pos p0{0, nl > 0 ? ptr[0] : 0, nl > 0 ? ptr[1] : 0};
    Pos p1 = adv(p0), p2 = adv(p1);
    SplRow cur = entries(row_of(p0));
    int i1 = row_of(p1);
    while (p0.lv < nl) {
        const int i2 = row_of(p2);            // index of the pass after next
        const SplRow nxt = entries(i1);       // entries of the next pass
        double acc[SPL_R];
#pragma unroll
        for (int q = 0; q < SPL_R; ++q) acc[q] = 0.0;
        if (cur.i >= 0) {
#pragma unroll
            for (int e = 0; e < SPL_DEPS; ++e)
                if (cur.dep[e] >= 0) {
                    const double lv_ = cur.lv[e] * lscale;
#pragma unroll
                    for (int q = 0; q < SPL_R; ++q)
                        if (q < R) acc[q] = fma(lv_, __hip_atomic_load(x + (int64_t)q * n + cur.dep[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), acc[q]);
                }
            for (int j = 1 + lane + SPL_DEPS * SPL_LANES; j < cur.cnt; j += SPL_LANES) {   // (m + 1 > 1 + SPL_DEPS * SPL_LANES only)
                const int64_t dep = NN[(int64_t)cur.i * mp1 + j];
                if (dep >= 0) {
                    const double lv_ = L[(int64_t)cur.i * mp1 + j] * lscale;
#pragma unroll
                    for (int q = 0; q < SPL_R; ++q)
                        if (q < R) acc[q] = fma(lv_, __hip_atomic_load(x + (int64_t)q * n + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < SPL_R; ++q) {
            double v = acc[q];
#pragma unroll
            for (int off = SPL_LANES / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, SPL_LANES);
            acc[q] = v;
        }
        if (cur.i >= 0 && lane < R) {
            double sm = 0.0;
#pragma unroll
            for (int q = 0; q < SPL_R; ++q)
                if (q == lane) sm = acc[q];
            __hip_atomic_store(x + (int64_t)lane * n + cur.i, (cur.rhs - sm) / (cur.diag * lscale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p1.lv != p0.lv) {   // the level ends with this pass: its x must be in memory before anybody reads it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        cur = nxt;
        i1 = i2;
        p0 = p1; p1 = p2; p2 = adv(p2);
    }
}

extern "C" int dgpamd_vecchia_spsolve(dgpamd_ctx *ctx, int64_t n, int m, const double *Lmat, const int64_t *NNarray,
                                      double inv_sqrt_scale, const double *b, double *x) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || !Lmat || !NNarray || !b || !x) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(spsolve_kernel, dim3(1), dim3(SPW), 0, ctx->stream, n, m + 1, Lmat, NNarray, inv_sqrt_scale,
                       (const double *)nullptr, 1, b, x);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_vecchia_spsolve_batch(dgpamd_ctx *ctx, int64_t n, int m, int nmat, int nrhs, const double *Lmat,
                                            const int64_t *NNarray, const double *inv_sqrt_scale, const double *b,
                                            double *x) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || nmat <= 0 || nrhs <= 0 || !Lmat || !NNarray || !inv_sqrt_scale || !b || !x) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(spsolve_kernel, dim3((unsigned)(nmat * nrhs)), dim3(SPW), 0, ctx->stream, n, m + 1, Lmat, NNarray, 1.0,
                       inv_sqrt_scale, nrhs, b, x);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" size_t dgpamd_vecchia_levels_bytes(int64_t n, int nmat) { return (size_t)nmat * splevel_words(n) * sizeof(int32_t); }

extern "C" int dgpamd_vecchia_levels(dgpamd_ctx *ctx, int64_t n, int m, int nmat, const int64_t *NNarray, void *sched) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || nmat <= 0 || !NNarray || !sched) BAD_ARG(ctx, "bad arguments");
    const int64_t words = (int64_t)splevel_words(n);
    hipLaunchKernelGGL(splevel_kernel, dim3((unsigned)nmat), dim3(SPW), 0, ctx->stream, n, m + 1, NNarray, (int32_t *)sched, words);
    hipLaunchKernelGGL(splevel_sort_kernel, dim3((unsigned)nmat), dim3(1024), 0, ctx->stream, n, (int32_t *)sched, words);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_vecchia_spsolve_levels(dgpamd_ctx *ctx, int64_t n, int m, int nmat, int nrhs, const double *Lmat,
                                             const int64_t *NNarray, const double *inv_sqrt_scale, const double *b, double *x,
                                             const void *sched) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || m < 0 || nmat <= 0 || nrhs <= 0 || !Lmat || !NNarray || !inv_sqrt_scale || !b || !x || !sched) BAD_ARG(ctx, "bad arguments");
    const int chunks = (nrhs + SPL_R - 1) / SPL_R;
    hipLaunchKernelGGL(spsolve_level_kernel, dim3((unsigned)(nmat * chunks)), dim3(1024), 0, ctx->stream, n, m + 1, Lmat, NNarray,
                       inv_sqrt_scale, nrhs, b, x, (const int32_t *)sched, (int64_t)splevel_words(n));
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// a22  gp_vecch (vecchia.py:635-654): block = [pm neighbours ; test point], rhs row y
// ---------------------------------------------------------------------------

template <int KIND>
__global__ __launch_bounds__(VW) void vecchia_gp_kernel(VGpArgs a) {
    extern __shared__ double lds[];
    const int pm = a.pm, D = a.vp.D, lda = pm + 3;
    double *A = lds;                       // [(pm+2)][lda]
    double *xs = A + (pm + 2) * lda;       // [pm+1][D]
    int *idx = reinterpret_cast<int *>(xs + (pm + 1) * D);
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    int b = 0;
    for (int c = 0; c < pm; ++c) b += (a.NN[t * pm + c] >= 0);
    for (int c = lane; c < b; c += VW) idx[c] = (int)a.NN[t * pm + c];
    __syncthreads();
    for (int e = lane; e < (b + 1) * D; e += VW) {
        int r = e / D, d = e - r * D;
        xs[e] = (r < b ? a.w[(int64_t)idx[r] * D + d] : a.x[t * D + d]) * a.vp.inv_len[d];
    }
    __syncthreads();
    const int bb = b + 1;
    for (int e = lane; e < bb * (bb + 1) / 2; e += VW) {
        int r, c;
        tri_decode(e, r, c);
        double v;
        if (r == c)
            v = 1.0 + a.vp.nugget * (r < b ? a.nugget_diag[idx[r]] : 1.0);
        else
            v = corr_pts<KIND>(xs + r * D, xs + c * D, D);
        A[r * lda + c] = v;
    }
    for (int c = lane; c <= bb; c += VW) A[bb * lda + c] = c < b ? a.y[idx[c]] : 0.0;
    lds_chol(A, lda, bb + 1, b, lane);
    if (lane == 0) {
        // after eliminating the b neighbour columns: Schur complement of the test point and -l21.w
        a.mean[t] = -A[bb * lda + b];
        a.var[t] = a.scale * A[b * lda + b];
    }
}

extern "C" int dgpamd_vecchia_gp(dgpamd_ctx *ctx, int kind, int64_t M, int64_t n, int D, int pm, const double *x,
                                 const double *w, const int64_t *NN, const double *y, double scale,
                                 const double *length_h, int nlen, double nugget, const double *nugget_diag,
                                 double *mean, double *var) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (M <= 0 || n <= 0 || pm <= 0 || !x || !w || !NN || !y || !nugget_diag || !mean || !var) BAD_ARG(ctx, "bad arguments");
    VGpArgs a;
    int rc = fill_vparams(ctx, a.vp, kind, D, length_h, nlen, nugget);
    if (rc) return rc;
    a.M = M; a.n = n; a.pm = pm; a.x = x; a.w = w; a.y = y; a.nugget_diag = nugget_diag; a.NN = NN; a.scale = scale;
    a.mean = mean; a.var = var;
    {
        const char *env = getenv("DGPAMD_VECCHIA_LDS");   // (1: the LDS version for every size -- the tests compare the two)
        if (pm <= VG_BC && D <= 16 && !(env && atoi(env))) {
            launch_vecchia_gp_reg(ctx, a);
            LAUNCH_CHECK(ctx);
            return DGPAMD_OK;
        }
    }
    const size_t shm = ((size_t)(pm + 2) * (pm + 3) + (size_t)(pm + 1) * D) * sizeof(double) + (size_t)pm * sizeof(int);
    const void *fn = kind == DGPAMD_SEXP ? (const void *)vecchia_gp_kernel<DGPAMD_SEXP> : (const void *)vecchia_gp_kernel<DGPAMD_MATERN25>;
    rc = set_lds(ctx, fn, shm);
    if (rc) return rc;
    {
        int prc = maybe_poison_lds(ctx, a.x);
        if (prc) return prc;
    }
    if (kind == DGPAMD_SEXP)
        hipLaunchKernelGGL(vecchia_gp_kernel<DGPAMD_SEXP>, dim3((unsigned)M), dim3(VW), shm, ctx->stream, a);
    else
        hipLaunchKernelGGL(vecchia_gp_kernel<DGPAMD_MATERN25>, dim3((unsigned)M), dim3(VW), shm, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// ---------------------------------------------------------------------------
// a23  link_gp_vecch (vecchia.py:758-796) with IJ_nb (vecchia.py:838-907)
// ---------------------------------------------------------------------------


#define VREC 29   // doubles per separable Matern record of a neighbour in LDS (28 used; odd stride)
template <int KIND>
__global__ __launch_bounds__(VW) void vecchia_linkgp_kernel(VLinkArgs a) {
    extern __shared__ double lds[];
    const int pm = a.pm, Dw = a.Dw, Dz = a.Dz, DT = Dw + Dz, lda = pm + 2;
    double *A = lds;                      // [(pm+1)][lda]   K block + rhs row y
    double *J = A + (pm + 1) * lda;       // [pm][lda]       J, then L^-1 J, then L^-1 (L^-1 J)^T
    double *xs = J + pm * lda;            // [pm][DT]        raw (unscaled) neighbour inputs
    double *Iv = xs + pm * DT;            // [pm]
    double *Ry = Iv + pm;                 // [lda]
    int *idx = reinterpret_cast<int *>(Ry + lda);
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    int b = 0;
    for (int c = 0; c < pm; ++c) b += (a.NN[t * pm + c] >= 0);
    for (int c = lane; c < b; c += VW) idx[c] = (int)a.NN[t * pm + c];
    __syncthreads();
    for (int e = lane; e < b * DT; e += VW) {
        int r = e / DT, d = e - r * DT;
        xs[e] = d < Dw ? a.w1[(int64_t)idx[r] * Dw + d] : a.wg[(int64_t)idx[r] * Dz + d - Dw];
    }
    __syncthreads();
    const double *mt = a.m + t * Dw, *vt = a.v + t * Dw;
    const double *zt = Dz ? a.z + t * Dz : nullptr;
    // I and the global-input factor
    for (int r = lane; r < b; r += VW) {
        double I, Iz = 1.0;
        if (KIND == DGPAMD_SEXP) {
            double e = 0.0, c1 = 1.0;
            for (int k = 0; k < Dw; ++k) {
                double l = a.len[k], d = xs[r * DT + k] - mt[k];
                e += d * d / (2.0 * vt[k] + l * l);
                c1 *= 1.0 + 2.0 * vt[k] / (l * l);
            }
            I = exp(-e) / sqrt(c1);
            double s = 0.0;
            for (int g = 0; g < Dz; ++g) {
                double d = (xs[r * DT + Dw + g] - zt[g]) / a.len[Dw + g];
                s = fma(d, d, s);
            }
            if (Dz) Iz = exp(-s);
        } else {
            I = 1.0;
            for (int k = 0; k < Dw; ++k) I *= matern_I_dim(xs[r * DT + k], mt[k], vt[k], a.len[k]);
            double pr = 1.0, s = 0.0;
            for (int g = 0; g < Dz; ++g) corr_accum_matern((xs[r * DT + Dw + g] - zt[g]) / a.len[Dw + g], pr, s);
            if (Dz) Iz = pr * exp(-SQRT5 * s);
        }
        Iv[r] = I * Iz;
        Ry[r] = Iz;   // park the global factor
    }
    __syncthreads();
    // K block (all DT columns) and J (local columns, times the global factors)
    double jc1 = 1.0;
    if (KIND == DGPAMD_SEXP) {
        for (int k = 0; k < Dw; ++k) jc1 *= 1.0 + 4.0 * vt[k] / (a.len[k] * a.len[k]);
        jc1 = 1.0 / sqrt(jc1);
    }
    for (int e = lane; e < b * (b + 1) / 2; e += VW) {   // the lower triangle only (over b * b with the upper half skipped, half the lanes idled)
        int r, c;
        tri_decode(e, r, c);
        double s = 0.0, pr = 1.0;
        for (int d = 0; d < DT; ++d) {
            double df = (xs[r * DT + d] - xs[c * DT + d]) / a.len[d];
            if (KIND == DGPAMD_SEXP)
                corr_accum_sexp(df, s);
            else
                corr_accum_matern(df, pr, s);
        }
        double kv = (KIND == DGPAMD_SEXP) ? exp(-s) : pr * exp(-SQRT5 * s);
        if (r == c) kv = 1.0 + a.nugget * a.nugget_diag[idx[r]];
        A[r * lda + c] = kv;
        double jv;
        if (KIND == DGPAMD_SEXP) {
            double ex = 0.0;
            for (int k = 0; k < Dw; ++k) {
                double l = a.len[k], ai = xs[r * DT + k] - mt[k], aj = xs[c * DT + k] - mt[k];
                ex += (ai + aj) * (ai + aj) / (8.0 * vt[k] + 2.0 * l * l) + (ai - aj) * (ai - aj) / (2.0 * l * l);
            }
            jv = jc1 * exp(-ex);
        } else {
            jv = 1.0;   // (the Matern factors follow dimension by dimension, below)
        }
        jv *= Ry[r] * Ry[c];
        J[r * lda + c] = jv;
        J[c * lda + r] = jv;
    }
    if (KIND != DGPAMD_SEXP) {
        // Matern-2.5 J factors through the separable form of csrc/linkfun.hpp (Jd = <S(x_lo), T(x_hi)> + (f2_hi - f2_lo) <S', T'>):
        // per dimension every neighbour's record (S[0..11] T[12..26] f2[27], ~500 instructions of erf / exp) is evaluated ONCE,
        // and a pair costs 30 multiply-adds and a select instead of matern_Jd's 3 erf + 5 exp -- with 50 neighbours 1275 pairs
        // share 50 records.  (A dimension with zero input variance has S[0] = T[0] = k(x, m), the rest zero: the product of the
        // two point correlations, functions.py:488-491.)
        double *rec = reinterpret_cast<double *>(idx + ((pm + 1) & ~1));   // [pm][VREC], behind idx (the launch sizes the LDS for it)
        for (int k = 0; k < Dw; ++k) {
            __syncthreads();
            {
                for (int r = lane; r < b; r += VW) {
                    double *rr = rec + r * VREC;
                    const double x = xs[r * DT + k];
                    if (vt[k] != 0.0) {
                        MaternDimConst kc;
                        matern_dim_const(mt[k], vt[k], a.len[k], kc);
                        double f2, so[12], to[15];
                        matern_role_S(x, kc, so, f2);
                        matern_role_T(x, kc, to);
                        for (int q = 0; q < 12; ++q) rr[q] = so[q];
                        for (int q = 0; q < 15; ++q) rr[12 + q] = to[q];
                        rr[27] = f2;
                    } else {
                        const double pt = matern_point(mt[k] - x, a.len[k]);
                        for (int q = 0; q < 28; ++q) rr[q] = 0.0;
                        rr[0] = pt;
                        rr[12] = pt;
                    }
                }
            }
            __syncthreads();
            for (int e = lane; e < b * (b + 1) / 2; e += VW) {
                int r, c;
                tri_decode(e, r, c);
                const double *ri = rec + r * VREC, *rj = rec + c * VREC;
                const double xi = xs[r * DT + k], xj = xs[c * DT + k];
                const double *lo = xi <= xj ? ri : rj, *hi = xi <= xj ? rj : ri;   // the record of the smaller / larger coordinate
                double o = 0.0, ed = 0.0;
                for (int q = 0; q < 12; ++q) o = fma(lo[q], hi[12 + q], o);
                for (int q = 0; q < 3; ++q) ed = fma(lo[6 + q], hi[24 + q], ed);
                const double f = fma(hi[27] - lo[27], ed, o);
                J[r * lda + c] *= f;
            }
        }
        __syncthreads();
        for (int e = lane; e < b * (b + 1) / 2; e += VW) {
            int r, c;
            tri_decode(e, r, c);
            J[c * lda + r] = J[r * lda + c];
        }
    }
    for (int c = lane; c <= b; c += VW) A[b * lda + c] = c < b ? a.y[idx[c]] : 0.0;
    lds_chol(A, lda, b + 1, b, lane);
    // Rinv_y = L^-T w
    for (int c = lane; c < b; c += VW) Ry[c] = A[b * lda + c];
    lds_backsolve_T(A, lda, b, Ry, lda, 1, lane);
    // mean and Ry^T J Ry on the untouched J
    double mu = 0.0, qd = 0.0;
    for (int r = lane; r < b; r += VW) {
        mu = fma(Iv[r], Ry[r], mu);
        double s = 0.0;
        for (int c = 0; c < b; ++c) s = fma(J[r * lda + c], Ry[c], s);
        qd = fma(Ry[r], s, qd);
    }
    mu = wsum(mu);
    qd = wsum(qd);
    __syncthreads();
    // tr(K^-1 J) = tr(L^-1 (L^-1 J)^T): two rounds of column-parallel forward substitutions
    for (int c = lane; c < b; c += VW)
        for (int r = 0; r < b; ++r) {
            double s = J[r * lda + c];
            for (int k = 0; k < r; ++k) s = fma(-A[r * lda + k], J[k * lda + c], s);
            J[r * lda + c] = s / A[r * lda + r];
        }
    __syncthreads();
    double tr = 0.0;
    for (int c = lane; c < b; c += VW) {
        // column c of G^T is row c of G; solve L h = G[c,:]^T and keep h_c
        double hc = 0.0;
        for (int r = 0; r <= c; ++r) {
            double s = J[c * lda + r];
            for (int k = 0; k < r; ++k) s = fma(-A[r * lda + k], J[c * lda + k], s);
            s /= A[r * lda + r];
            J[c * lda + r] = s;   // row c is private to this lane from here on
            hc = s;
        }
        tr += hc;
    }
    tr = wsum(tr);
    if (lane == 0) {
        a.mean[t] = mu;
        a.var[t] = fabs(qd - mu * mu + a.scale * (1.0 + a.nugget - tr));
    }
}

extern "C" int dgpamd_vecchia_linkgp(dgpamd_ctx *ctx, int kind, int64_t M, int64_t n, int Dw, int Dz, int pm,
                                     const double *m, const double *v, const double *z, const double *w1,
                                     const double *wg, const int64_t *NN, const double *y, double scale,
                                     const double *length_h, int nlen, double nugget, const double *nugget_diag,
                                     double *mean, double *var) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (M <= 0 || n <= 0 || pm <= 0 || !m || !v || !w1 || !NN || !y || !nugget_diag || !mean || !var || !length_h)
        BAD_ARG(ctx, "bad arguments");
    if (kind != DGPAMD_SEXP && kind != DGPAMD_MATERN25) BAD_ARG(ctx, "kind must be 0 or 1");
    if (Dw <= 0 || Dz < 0 || Dw + Dz > DGPAMD_MAXD || (nlen != 1 && nlen != Dw + Dz)) BAD_ARG(ctx, "bad dimensions");
    if (Dz > 0 && (!z || !wg)) BAD_ARG(ctx, "Dz > 0 needs z and wg");
    VLinkArgs a;
    a.kind = kind; a.Dw = Dw; a.Dz = Dz; a.pm = pm; a.M = M; a.n = n; a.m = m; a.v = v; a.z = z; a.w1 = w1; a.wg = wg;
    a.y = y; a.nugget_diag = nugget_diag; a.NN = NN; a.scale = scale; a.nugget = nugget; a.mean = mean; a.var = var;
    for (int d = 0; d < Dw + Dz; ++d) a.len[d] = length_h[nlen == 1 ? 0 : d];
    {
        const char *env = getenv("DGPAMD_VECCHIA_LDS");   // (1: the LDS version for every size -- the tests compare the two)
        if (pm <= VL_BC && Dw <= 8 && Dz <= 8 && !(env && atoi(env))) {
            if (kind == DGPAMD_SEXP)
                launch_vecchia_linkgp_sexp_reg(ctx, a);
            else
                launch_vecchia_linkgp_matern_reg(ctx, a);
            LAUNCH_CHECK(ctx);
            return DGPAMD_OK;
        }
    }
    const int lda = pm + 2;
    const size_t shm = ((size_t)(pm + 1) * lda + (size_t)pm * lda + (size_t)pm * (Dw + Dz) + pm + lda) * sizeof(double) +
                       (size_t)((pm + 1) & ~1) * sizeof(int) + (kind == DGPAMD_SEXP ? 0 : (size_t)pm * VREC * sizeof(double));
    const void *fn = kind == DGPAMD_SEXP ? (const void *)vecchia_linkgp_kernel<DGPAMD_SEXP>
                                         : (const void *)vecchia_linkgp_kernel<DGPAMD_MATERN25>;
    int rc = set_lds(ctx, fn, shm);
    if (rc) return rc;
    {
        int prc = maybe_poison_lds(ctx, a.y);
        if (prc) return prc;
    }
    if (kind == DGPAMD_SEXP)
        hipLaunchKernelGGL(vecchia_linkgp_kernel<DGPAMD_SEXP>, dim3((unsigned)M), dim3(VW), shm, ctx->stream, a);
    else
        hipLaunchKernelGGL(vecchia_linkgp_kernel<DGPAMD_MATERN25>, dim3((unsigned)M), dim3(VW), shm, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}
