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
#include "diagfac.hpp"

// ----------------------------------------------------------------------------
// diagonal block: Cholesky + inverse of the factor (diagfac.hpp)
// ----------------------------------------------------------------------------
// logdet / info of the block just factored (wave 0 knows about failed pivots; the factor's last barrier made the
// pivots visible).  Kept out of diag_factor so that the chain publishes its block BEFORE this reduction.
__device__ __forceinline__ void diag_logdet(DiagShared &sh, int ncol, int k, int b, int bad, double *logdet, int32_t *info) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const double pv = sh.piv[tid];
        double v = (tid < ncol && pv > 0.0) ? log(pv) : 0.0;   // (a failed pivot is reported through info)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (tid == 0) {
            // (device-scope accesses: the words are read-modified by the chain workgroups of consecutive launches, which run
            //  on whatever XCD the dispatcher picks, and the workspace address they live at holds other data -- diagonal-block
            //  inverses -- under another batch size's layout: no per-XCD L2 line of an earlier use may serve them)
            const double ld0 = k == 0 ? 0.0 : __hip_atomic_load(logdet + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(logdet + b, ld0 + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int inf = k == 0 ? 0 : __hip_atomic_load(info + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bad && inf == 0) inf = k * 64 + bad;
            __hip_atomic_store(info + b, inf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ----------------------------------------------------------------------------
// 64x64x64 tile GEMM engine on f64 MFMA
// ----------------------------------------------------------------------------
enum { G_TRTRI1 = 2, G_TRTRI2 = 3, G_LAUUM = 4 };

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
    constexpr int OPB = OP_KM;
    __shared__ double As[(OPA == OP_MK) ? 64 * LDM : KC * LDK];
    __shared__ double Bs[(OPB == OP_MK) ? 64 * LDM : KC * LDK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ld = g.ld;
    double *A = g.A + (int64_t)blockIdx.z * g.stride_a;
    double *B = g.B ? g.B + (int64_t)blockIdx.z * g.stride_a : nullptr;

    int bi, bj, kb0, kb1;          // output tile, k-block range [kb0, kb1)
    double *C;                     // output buffer
    double sign = 1.0;
    if (MODE == G_TRTRI1 || MODE == G_TRTRI2) {
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
    // operands of k-block kb
    auto operands = [&](int kb, const double *&Ag, const double *&Bg, int64_t &lda, int64_t &ldb, int &lim) {
        lda = ld; ldb = ld; lim = 64;
        if (MODE == G_TRTRI1) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = A + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
        } else if (MODE == G_TRTRI2) {
            Ag = A + ((int64_t)bi * 64) * ld + (int64_t)kb * 64;
            Bg = B + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
        } else {
            Ag = A + ((int64_t)kb * 64) * ld + (int64_t)bi * 64;
            Bg = A + ((int64_t)kb * 64) * ld + (int64_t)bj * 64;
            int64_t l = g.n - (int64_t)kb * 64;   // rows >= n (right-hand sides) do not belong to L^-1
            lim = l >= 64 ? 64 : (int)l;
        }
    };
    auto fetch = [&](int st, HalfTile &fa, HalfTile &fb) {
        const double *Ag, *Bg;
        int64_t lda, ldb;
        int lim;
        operands(kb0 + (st >> 1), Ag, Bg, lda, ldb, lim);
        fa = (OPA == OP_MK) ? fetch_mk(Ag, lda, tid, st & 1) : fetch_km(Ag, lda, tid, st & 1, lim);
        fb = (OPB == OP_MK) ? fetch_mk(Bg, ldb, tid, st & 1) : fetch_km(Bg, ldb, tid, st & 1, lim);
    };
    // software pipeline over the 32-deep stages: the loads of stage st+1 are in flight during the MFMAs of stage st
    const int nst = 2 * (kb1 - kb0);
    HalfTile fa, fb;
    if (nst > 0) fetch(0, fa, fb);
    for (int st = 0; st < nst; ++st) {
        __syncthreads();
        if (OPA == OP_MK) commit_mk(fa, As, tid); else commit_km(fa, As, tid);
        if (OPB == OP_MK) commit_mk(fb, Bs, tid); else commit_km(fb, Bs, tid);
        if (st + 1 < nst) fetch(st + 1, fa, fb);
        __syncthreads();
        mfma_tile<OPA, OPB>(As, Bs, acc, wave, lane, sign);
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
// One block step k of the factorisation as ONE launch, driven by a TASK TABLE built on the host (one task per
// workgroup and matrix, in urgency = dispatch order).  A task is a 64x64 tile update
//      C = (first ? 0 : C) +/- sum_{kb0 <= kb < kb0+nkb}  L[li][kb] R[ri][kb]^T          (buffers A / T / S)
// followed by one of
//   STORE   (bulk)   store the tile;
//   CHAIN            factor the diagonal tile (diag_factor) and publish the block's inverse W_k with an agent-scope
//                    release (MI355X_MICROARCH.md hand-off recipe);
//   SOLVE   (panel)  wait for that flag (acquire, bounded spin), tile <- tile * W_k^T  -- the explicit inverse of the
//                    diagonal block turns every triangular solve into an MFMA GEMM;
//   TDIAG            wait, write W_k^T (the diagonal block of L^-T).
// Plain factorisation (buffer A): column k = chain + panel tasks applying panel k-1; the bulk is LAZY -- tiles of the
// columns k+1, k+1+LAZY, ... take LAZY = 4 panels (256 pivots) per visit, so C is read/written once per 256 pivots
// while the serial chain still applies a single panel.
// Fused inverse (dgpamd_potrf_inv): the identity rides along as n extra rows (buffer T, only its non-zero tiles
// exist) with a zero corner (buffer S).  The same right-looking sweep then leaves T = L^-T and the Schur
// complement of K in [[K, I], [I, 0]], i.e. S = K^-1 (accumulated with a + sign), and column n of T = -K^-1 y:
// what potri (TRTRI + LAUUM) computes in ~13 more latency-bound launches becomes bulk work in the shadow of the
// pivot chain.  T tiles follow the same lazy rule as A's; S tile (q, q') is visited by the launches
// k = q+2, q+4, ... (two panels each) and by two tail launches.
// CU guard: with two workgroups per CU, workgroups i and i + 256 of a launch share a CU (measured, tools/ubench/
// hwid.hip).  The pivot chain is issue- and LDS-bound and a bulk workgroup beside it slows it by ~40 %, so workgroups
// 256 .. 256+batch-1 are placeholders that sleep until "their" chain has published its block (potrf_inv at B = 3:
// 1.21 -> 1.00 ms).  They hold no task, check through HW_ID / XCC_ID that they really are on the chain's CU, and
// leave at once otherwise.
// Deadlock freedom: only SOLVE / TDIAG workgroups and the placeholders wait, and only for the chain workgroup of the SAME launch, which
// has a lower block index (dispatched first) and never waits itself.  The spin is bounded (info = -1).
// ----------------------------------------------------------------------------
enum { T_STORE = 0, T_SOLVE = 1, T_CHAIN = 2, T_TDIAG = 3, T_LOOK = 4, T_LOOKD = 5 };
enum { BUF_A = 0, BUF_T = 1, BUF_S = 2 };

struct StepArgs {
    double *buf[3];   // A (Np x Np, K with the right-hand sides as extra rows), T (L^-T), S (K^-1)
    double *ws;
    int64_t ld, stride_a, stride_ws, n;
    int nbk, k, batch;
    const int4 *tasks;   // this launch's tasks
    int nhead, npanel, lead;   // task order in the table: head (chain, tdiag), panel (solve), bulk; the panel workgroups
                               // are dispatched after the first `lead` bulk tasks (they only wait: no slot hogging)
    int guard, nph;      // guard > 0: workgroups m * guard .. m * guard + batch - 1 (guard = number of CUs, m = 1 .. nph) are
                         // placeholders that keep the chain workgroups' CUs to themselves (nph + 1 workgroups fit a CU)
    double *logdet;
    int32_t *info;
    int32_t *flags;
    long long *trace;
};
#define STAMP(slot)                                                                       \
    do {                                                                                  \
        if (g.trace && b == 0 && tid == 0) g.trace[16 * g.k + (slot)] = wall_clock64();   \
    } while (0)

static inline int4 make_task(int post, int first, int plus, int mask_last, int bufC, int ci, int cj, int bufL, int li,
                             int bufR, int ri, int kb0, int nkb) {
    int4 t;
    t.x = post | (first << 4) | (plus << 5) | (mask_last << 6) | (bufC << 8) | (bufL << 10) | (bufR << 12);
    t.y = ci | (cj << 16);
    t.z = li | (ri << 16);
    t.w = kb0 | (nkb << 16);
    return t;
}

__device__ __forceinline__ void load_acc(d4 (&acc)[4], const double *C, int64_t ld, int crow, int ccol) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol];
}
__device__ __forceinline__ void load_acc_sc1(d4 (&acc)[4], const double *C, int64_t ld, int crow, int ccol) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[t][r] = __hip_atomic_load(C + (int64_t)(crow + 4 * r) * ld + 16 * t + ccol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_acc(const d4 (&acc)[4], double *C, int64_t ld, int crow, int ccol) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(int64_t)(crow + 4 * r) * ld + 16 * t + ccol] = acc[t][r];
}
// acc = (C ? C : 0) + sign * sum over nkb consecutive 64-blocks  L_kb R_kb^T.  Software pipelined: the global loads of
// the next half tile are in flight (registers) while the MFMAs of the current one run.  In the 64-block number
// mask_kb (relative to the first) the columns >= klim of both operands read as zero.
// SC1: every global load bypasses L1 (operands handed over inside a launch: the one-launch kernel).
// KEEP: acc is carried over from an earlier call (the panels of a visit applied in two runs, see the one-launch kernel's workers).
template <bool SC1 = false, bool AHEAD = false, bool KEEP = false>
__device__ __forceinline__ void tile_update(d4 (&acc)[4], const double *C, const double *Lp, const double *Rp, int64_t ld,
                                            int nkb, double sign, int mask_kb, int klim, double *As, double *Bs, int tid,
                                            int wave, int lane) {
    const int c2 = (tid & 15) * 2, r0 = tid >> 4;
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    double2 pa[4], pb[4];
    const int nh = 2 * nkb;
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void *)Lp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)Rp, 0, 0x7fffffff, 0x00020000);
    auto fetch = [&](int hh) {
        const int64_t off = 32 * hh;   // consecutive half tiles are consecutive 32-column slabs
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (SC1) {
                const int bo = (int)(((int64_t)(r0 + 16 * it) * ld + off + c2) * 8);
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsL, bo, 0, 16);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsR, bo, 0, 16);
                pa[it] = make_double2(__hiloint2double(va.y, va.x), __hiloint2double(va.w, va.z));
                pb[it] = make_double2(__hiloint2double(vb.y, vb.x), __hiloint2double(vb.w, vb.z));
            } else {
                pa[it] = *reinterpret_cast<const double2 *>(Lp + (int64_t)(r0 + 16 * it) * ld + off + c2);
                pb[it] = *reinterpret_cast<const double2 *>(Rp + (int64_t)(r0 + 16 * it) * ld + off + c2);
            }
        }
    };
    if (nh > 0) fetch(0);
    if (KEEP) {
    } else if (C) {
        if (SC1) load_acc_sc1(acc, C, ld, crow, ccol);
        else load_acc(acc, C, ld, crow, ccol);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    for (int hh = 0; hh < nh; ++hh) {
        __syncthreads();
        if ((hh >> 1) == mask_kb) {
            const int c = 32 * (hh & 1) + c2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (c >= klim) pa[it].x = pb[it].x = 0.0;
                if (c + 1 >= klim) pa[it].y = pb[it].y = 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = r0 + 16 * it;
            As[r * LDM + c2] = pa[it].x; As[r * LDM + c2 + 1] = pa[it].y;
            Bs[r * LDM + c2] = pb[it].x; Bs[r * LDM + c2 + 1] = pb[it].y;
        }
        if (hh + 1 < nh) fetch(hh + 1);
        __syncthreads();
        // (AHEAD: the one-launch kernel's workers, two waves per SIMD -- the operand fragments requested two k-steps ahead gave
        //  the fused inverse 1-2.5 % at 6-12 matrices; the per-step kernel's four waves per SIMD hide the LDS latency themselves)
        if (AHEAD) mfma_tile_ahead<OP_MK, OP_MK>(As, Bs, acc, wave, lane, sign);
        else mfma_tile<OP_MK, OP_MK>(As, Bs, acc, wave, lane, sign);
    }
}
// The pivot chain's diagonal tile in the COLUMN-BLOCK layout of diagfac.hpp (wave w: acc[t][r] = S[16t + lu + 4r][16w + lm]),
// read from the LOWER triangle of the stored tile (the last block's upper part does not mirror the carried rows): the
// 16x16 tiles (w, t), t <= w, are loaded row-contiguously (four full 128-byte lines per load instruction; the transposed
// access touches sixteen) and transposed inside the wave through `scratch` (16 x 16 doubles of LDS per wave).
template <bool SC1 = false>
__device__ __forceinline__ void load_cb_lower(d4 (&acc)[4], const double *C, int64_t ld, int wave, int lane, double *scratch) {
    const int lm = lane & 15, lu = lane >> 4;
    d4 nat[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double *p = C + (int64_t)(16 * wave + lu + 4 * r) * ld + 16 * t + lm;
            nat[t][r] = (t <= wave) ? (SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p) : 0.0;
        }
    vlds_double *S = (vlds_double *)scratch + wave * 256;   // element (i, j) at i * 16 + (j ^ i): both passes conflict-free
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t <= wave) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(lu + 4 * r) * 16 + (lm ^ (lu + 4 * r))] = nat[t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double tr = S[lm * 16 + ((lu + 4 * r) ^ lm)];
                acc[t][r] = (t == wave && lu + 4 * r >= lm) ? nat[t][r] : tr;
            }
        } else {
            acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        }
    }
}
// ... and its update acc = C - P P^T with ONE panel tile P = A[k][k-1] that serves as both operands.  It sits on the
// critical path of every block step and its loads are first touches (the tile was written by another CU in the previous
// launch), so C and BOTH halves of P are requested at once -- one exposed memory latency instead of two.
__device__ __forceinline__ void diag_update_cb(d4 (&acc)[4], const double *C, const double *P, int64_t ld, double *As, double *Bs,
                                               int tid, int wave, int lane) {
    const HalfTile p0 = fetch_mk(P, ld, tid, 0), p1 = fetch_mk(P, ld, tid, 1);
    load_cb_lower(acc, C, ld, wave, lane, Bs);
    const vlds_double *Ap = (const vlds_double *)As;
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        commit_mk(h == 0 ? p0 : p1, As, tid);
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += 4) {
            const double bv = -Ap[(16 * wave + m) * LDM + k0 + kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double av = Ap[(16 * t + m) * LDM + k0 + kk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
}
// out += in * Bg^T   (in: accumulator-layout 64x64 tile, Bg: 64x64 row-major LOWER-TRIANGULAR tile in global memory: the
// inverse W_k of a diagonal block's factor).  Both halves of Bg are requested up front: one exposed load latency instead
// of two on the panel's critical path.  Column tile t of the product takes the 16-blocks kb <= t of the sum only (40
// instead of 64 MFMAs per wave: the blocks above W_k's diagonal are exact zeros, and the f64 MFMA runs at the vector
// rate on gfx950, so the skipped zeros are time).
template <bool SC1 = false>
__device__ __forceinline__ void mul_acc_bt(d4 (&out)[4], const d4 (&in)[4], const double *Bg, int64_t ldb,
                                           double *As, double *Bs, int tid, int wave, int lane) {
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    const HalfTile b0 = SC1 ? fetch_mk_sc1(Bg, ldb, tid, 0) : fetch_mk(Bg, ldb, tid, 0);
    const HalfTile b1 = SC1 ? fetch_mk_sc1(Bg, ldb, tid, 1) : fetch_mk(Bg, ldb, tid, 1);
    const vlds_double *Ap = (const vlds_double *)As, *Bp = (const vlds_double *)Bs;
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(crow + 4 * r) * LDM + 16 * t + ccol] = in[2 * h + t][r];
        commit_mk(h == 0 ? b0 : b1, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += 4) {
            const int kb = 2 * h + (k0 >> 4);
            const double av = Ap[(16 * wave + m) * LDM + k0 + kk];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t >= kb) out[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bp[(16 * t + m) * LDM + k0 + kk], out[t], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void wg_release_store(int32_t *flag, int value, int tid) {
    // every storing wave drains its stores, one agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void wg_wait_acquire(int32_t *flag, int target, int32_t *info, int tid) {
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 24)) {
                *info = -1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// The CU this workgroup runs on: XCC_ID[2:0] and HW_ID[15:8] (SE / SH / CU ids) -- measured on MI355X
// (tools/ubench/hwid.hip): 256 distinct keys, and workgroups i and i + 256 of a launch share one.
__device__ __forceinline__ int cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    return (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xffu)) + 1;
}

__global__ __launch_bounds__(256, 2) void potrf_step_kernel(StepArgs g) {
    __shared__ double tiles[2 * 64 * LDM];   // As | Bs
    double *As = tiles, *Bs = tiles + 64 * LDM;
    // the diagonal factor's LDS lives in Bs (idle while a chain workgroup factors): 35 KB per workgroup, four per CU
    static_assert(sizeof(DiagShared) <= 64 * LDM * sizeof(double), "DiagShared must fit the B staging tile");
    DiagShared &sh = *reinterpret_cast<DiagShared *>(Bs);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    const int k = g.k, batch = g.batch;
    int32_t *cukey = g.flags + DGPAMD_MAXB;   // per matrix: (launch + 1) << 16 | CU key of its chain workgroup
    int bidx = blockIdx.x;
    if (g.guard && bidx >= g.guard) {
        // The pivot chain is issue- and LDS-bound: a bulk workgroup on the same CU slows it by ~40 % (measured).  With nph + 1
        // workgroups per CU, workgroups m * (number of CUs) + b (m = 1 .. nph) land beside chain workgroup b: they are placeholders
        // that sleep until that chain has published its block, so the slots are taken and the chain has the CU to itself.  (If
        // the hardware placed one elsewhere it leaves at once.)
        const int row = bidx / g.guard, pos = bidx - row * g.guard;
        if (row <= g.nph && pos < batch) {
            if (tid == 0) {
                const int c = pos, tag = (k + 1) << 16;
                int key = 0, spins = 0;
                while (((key = __hip_atomic_load(&cukey[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 16) != k + 1 &&
                       ++spins < 64)
                    __builtin_amdgcn_s_sleep(2);
                if (key == (tag | cu_key())) {
                    spins = 0;
                    while (__hip_atomic_load(&g.flags[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1 && ++spins < (1 << 20))
                        __builtin_amdgcn_s_sleep(16);
                }
            }
            return;
        }
        bidx -= batch * (row <= g.nph ? row : g.nph);   // placeholders of this and the earlier rows
    }
    const int b = bidx % batch;
    int slot = bidx / batch;
    if (slot >= g.nhead) {
        if (slot < g.nhead + g.lead) slot += g.npanel;                      // one of the first bulk tasks
        else if (slot < g.nhead + g.lead + g.npanel) slot -= g.lead;        // a panel task
    }
    const int4 tk = g.tasks[slot];
    // debug (dgpamd_debug_trace): per-workgroup start / end / (task kind, panels, CU) of the launch named in trace[4095]
    long long *wtr = (g.trace && tid == 0 && g.trace[4095] == k) ? g.trace + 4096 + 4 * (int64_t)blockIdx.x : nullptr;
    if (wtr) {
        wtr[0] = wall_clock64();
        wtr[2] = (tk.x & 15) | ((tk.w >> 16) << 8) | ((long long)cu_key() << 16);
    }
    const int post = tk.x & 15, first = (tk.x >> 4) & 1, plus = (tk.x >> 5) & 1, mask_last = (tk.x >> 6) & 1;
    const int bufC = (tk.x >> 8) & 3, bufL = (tk.x >> 10) & 3, bufR = (tk.x >> 12) & 3;
    const int ci = tk.y & 0xffff, cj = tk.y >> 16, li = tk.z & 0xffff, ri = tk.z >> 16;
    const int kb0 = tk.w & 0xffff, nkb = tk.w >> 16;
    const int64_t ld = g.ld, mo = (int64_t)b * g.stride_a;
    const int64_t rem_last = g.n - (int64_t)(g.nbk - 1) * 64;
    const int ncol_last = rem_last > 0 ? (int)rem_last : 0;   // pivots of the last block (the rest are carried rows)
    double *C = g.buf[bufC] + mo + ((int64_t)ci * 64) * ld + (int64_t)cj * 64;
    double *Wk = g.ws + (int64_t)b * g.stride_ws + (int64_t)k * 4096;

    if (post == T_TDIAG) {   // T[k][k] = W_k^T, rows of the carried right-hand sides zeroed
        wg_wait_acquire(g.flags + b, k + 1, g.info + b, tid);
        const int nrow = (k == g.nbk - 1) ? ncol_last : 64;
        for (int idx = tid; idx < 4096; idx += 256) tiles[(idx >> 6) * 65 + (idx & 63)] = Wk[idx];
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            C[(int64_t)r * ld + c] = r < nrow ? tiles[c * 65 + r] : 0.0;
        }
        return;
    }
    Tile64 acc;
    if (post == T_CHAIN) {
        STAMP(0);
        __builtin_amdgcn_s_setprio(3);
        if (g.guard && tid == 0)
            __hip_atomic_store(&cukey[b], ((k + 1) << 16) | cu_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (post == T_CHAIN) {
        if (nkb == 1)
            diag_update_cb(acc.v, C, g.buf[bufL] + mo + ((int64_t)li * 64) * ld + (int64_t)kb0 * 64, ld, As, Bs, tid, wave, lane);
        else {
            load_cb_lower(acc.v, C, ld, wave, lane, Bs);
            __syncthreads();   // (the factor's LDS lives in Bs as well: every wave is done with its transposition scratch)
        }
    } else
        tile_update(acc.v, first ? nullptr : C, g.buf[bufL] + mo + ((int64_t)li * 64) * ld + (int64_t)kb0 * 64,
                    g.buf[bufR] + mo + ((int64_t)ri * 64) * ld + (int64_t)kb0 * 64, ld, nkb, plus ? 1.0 : -1.0,
                    mask_last ? g.nbk - 1 - kb0 : -1, ncol_last, As, Bs, tid, wave, lane);
    if (post == T_STORE) {
        store_acc(acc.v, C, ld, crow, ccol);
        if (wtr) wtr[1] = wall_clock64();
        return;
    }
    if (post == T_CHAIN) {
        STAMP(1);
        const int64_t rem = g.n - (int64_t)k * 64;
        const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        Tile64 winv;
        const int bad = diag_factor(acc, winv, sh, ncol, (g.trace && b == 0) ? g.trace + 1024 + 16 * k : nullptr);
        diag_store_inverse<false>(winv, Wk);
        STAMP(2);
        wg_release_store(g.flags + b, k + 1, tid);   // the panel can start: it needs W_k only
        STAMP(3);
        diag_store_factor(acc, C, ld, ncol);         // the factor's own tile and the log-determinant are nobody's input
        diag_logdet(sh, ncol, k, b, bad, g.logdet, g.info);
        if (wtr) wtr[1] = wall_clock64();
        return;
    }
    // T_SOLVE: wait for W_k, then tile <- tile * W_k^T
    const bool stamp = (bufC == BUF_A && ci == k + 1);
    if (stamp) STAMP(8);
    wg_wait_acquire(g.flags + b, k + 1, g.info + b, tid);
    if (stamp) STAMP(9);
    d4 out[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = (d4){0.0, 0.0, 0.0, 0.0};
    mul_acc_bt(out, acc.v, Wk, 64, As, Bs, tid, wave, lane);
    store_acc(out, C, ld, crow, ccol);
    if (stamp) STAMP(10);
    if (wtr) wtr[1] = wall_clock64();
}

// ----------------------------------------------------------------------------
// The whole factorisation (with or without the fused inverse) as ONE persistent launch: dataflow over tile versions.
//
// Workgroups 0 .. batch-1 are the pivot CHAINS, one per matrix, alive for all block steps; every other workgroup is a
// WORKER that pulls 64x64-tile tasks from one queue (an agent-scope counter) in the order of a host-built table.
// A chain never hands its critical data to another workgroup: after factoring block k it publishes W_k, solves the
// first panel tile P = A[k+1][k] W_k^T itself, applies it to the next diagonal tile, D = A[k+1][k+1] - P P^T, keeps D
// in registers and factors on.  What it needs from the workers -- A[k+1][k] and A[k+1][k+1] with the panels <= k-1
// applied -- depends only on block k-1, so it is produced while the chain factors block k (look-ahead of one block,
// no launch boundary, no hand-off on the critical path).
// Every tile has a VERSION word: the number of finished visits, or VER_FINAL once nobody will write it again.  A task
// waits (one wave, one flag per lane, bounded) until its output tile has seen all earlier visits and its operand tiles
// are final, takes ONE agent-scope acquire, works with plain loads, stores its tile WRITE-THROUGH (sc1) and publishes
// the new version after every storing wave has drained (MI355X_MICROARCH.md, hand-off recipe R1).  Panel tasks wait
// for the chain's W_k the same way.  Deadlock freedom: the table is a topological order of the task graph and is
// pulled in order, so the oldest unfinished task always has its inputs finished or in progress on a resident
// workgroup; the chains are the first workgroups of the grid.  Every spin is bounded (status word -> DGPAMD_HANDOFF).
// Workers that share a CU with a chain leave at once (the chain is issue- and LDS-bound; see the CU guard above).
// ----------------------------------------------------------------------------
#define VER_FINAL 0x40000000
#define VER_PLANES 4   // version words per matrix: buffers A, T, S + one plane of auxiliary words
#define MEGA_SPIN_LIMIT (1 << 20)   // polls of ~1 us: a lost hand-off gives up after about a second

// One word on a 128-byte line of its own.  Round 5: the queue heads (one returning atomic per task pulled), the status word and the first sixteen matrices' W_k
// counters used to share ONE line -- every publication of a W_k by a chain queued behind the workers' ticket atomics and the polls of everyone who waited for any
// matrix's W (the "hot line" effects of profiles/r05_mega_table_sweeps.txt: one more read per solve of that line cost 15 %).
struct alignas(128) MegaLine {
    int32_t v;
    int32_t fill[31];
};
struct MegaSync {   // zeroed by a memset node ahead of every launch; the version words follow it
    int32_t qhead;                  // (unused)
    int32_t status;                 // 0, or the code of the first spin that gave up
    int32_t pad[30];                // (pad[0]: the spare word of the deferred post-processing)
    MegaLine ghead[8];              // queue head of each group of matrices (see MegaArgs::ngroups)
    MegaLine uhead[8];              // head of each group's CRITICAL queue (ghead: the bulk queue's), see MegaArgs::nut
    int32_t cukey[DGPAMD_MAXB];     // per matrix: CU of its chain workgroup (cu_key()); read once by every worker
    MegaLine wflag[DGPAMD_MAXB];    // per matrix: blocks factored (W_k is readable for k < wflag)
};
static_assert(sizeof(MegaSync) % 128 == 0, "MegaSync is whole lines");

struct MTask {
    int4 a;   // as the per-launch tables: what / where (make_task)
    int4 b;   // x: visits of the output tile before this one; y: 1 = this visit makes it final; z: block whose inverse a
              // SOLVE / TDIAG task multiplies with
};

struct MegaArgs {
    double *buf[3];
    double *ws;
    int64_t ld, stride_a, stride_ws, n;
    int nbk, batch, inv;
    const MTask *tasks;
    int ntask;               // per matrix
    int nut;                 // the first nut entries of `tasks` form the CRITICAL queue (per block step the look-ahead pair, the first
                             // solve and the catch-up updates: what the chain waits for next), the rest the bulk queue
    int ncrit;               // workers per matrix that serve the critical queue first
    const int2 *chain_need;  // per block k: worker visits of A[k+1][k] and of A[k+1][k+1] the chain waits for
    MegaSync *sync;
    int32_t *ver;            // [batch][VER_PLANES][nbk * nbk]: versions of the tiles of A, T, S; auxiliary words (see T_LOOKD)
    double *logdet;
    int32_t *info;
    long long *trace;
    long long *tlog;         // debug (dgpamd_debug_tasklog): every chain step's and every task's stamps, or null
    const int32_t *pred;     // null, or a device word: the launch does nothing when it is non-zero
    double *piv;             // [batch][ld]: the pivots (second chain form: their logarithms are summed at the end)
    int split;               // 1 (default): a visit whose newest panel has not arrived applies the older ones first (DGPAMD_MEGA_SPLIT=0: off)
    int nowait;              // debug (DGPAMD_MEGA_NOWAIT=1, timing only, results invalid): the chain does not wait for the workers
    int ngroups;             // The matrices are dealt into this many groups (matrix b: group b % ngroups), the XCDs as well
                             // (XCD x: group x % ngroups; 1, 2, 4 or 8), and a worker takes the tasks of its own group's
                             // matrices first: every XCD has its own L2, so with one queue for all a panel tile was fetched
                             // into all eight of them (PMC: ~4x the algorithmic HBM-side traffic at 12 matrices).
};

// Write-through (sc1) store of an accumulator tile, 16 bytes per lane: neighbouring lanes hold neighbouring columns of the
// same rows, so each pair swaps one value (DPP quad_perm [1,0,3,2]) -- the even lane ends up with two columns of row r0, the odd
// lane with two columns of row r1 -- and every store instruction still writes whole 128-byte lines (8-byte sc1 stores cost
// 2.7x the time per byte: MI355X_MICROARCH.md, stores of each flavour).
__device__ __forceinline__ double swap_pair(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void store_acc_sc1(const d4 (&acc)[4], double *C, int64_t ld, int crow, int ccol) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, 0x7fffffff, 0x00020000);
    const bool odd = ccol & 1;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const double a0 = acc[t][2 * rp], a1 = acc[t][2 * rp + 1];
            const double got = swap_pair(odd ? a0 : a1);
            const double lo = odd ? got : a0, hi = odd ? a1 : got;
            const int row = crow + 4 * (2 * rp + (odd ? 1 : 0)), col = 16 * t + (ccol & ~1);
            u32x4 v;
            v.x = __double2loint(lo); v.y = __double2hiint(lo); v.z = __double2loint(hi); v.w = __double2hiint(hi);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((int64_t)row * ld + col) * 8), 0, 16);
        }
    // (eight 16-byte stores in a row: their data registers stay untouched for 16 wait states, as behind the chain's panel-tile stores --
    //  tests/test_isa_hazards.py; the callers used to drain right behind them, now the next ticket is pulled first)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}
// every storing wave drains its stores, then one lane publishes
__device__ __forceinline__ void wg_publish(int32_t *flag, int value, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wave 0: lane l < nflag polls flag[l] until it reaches need[l] (relaxed, bounded), then ONE agent-scope acquire;
// the workgroup's barrier follows.  Returns (to every thread, through LDS word `bc`) 0 or the give-up code.
// ACQ false would rely on sc1 loads alone (L1 bypassed, no acquire); it is measured valid only for one workgroup per CU
// (MI355X_MICROARCH.md, valid forms) and saves ~1 us per task here -- not used: the acquire stays.
template <bool ACQ>
__device__ __forceinline__ void wg_wait_flags(const int32_t *addr, int need, int code, int32_t *status, int tid) {
    if (tid < 64) {
        bool ok = (addr == nullptr);
        int spins = 0;
        for (;;) {
            if (!ok) ok = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            // give up after the limit -- or at once when another workgroup already has (its inputs may never come)
            if (spins > MEGA_SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (tid == 0) atomicCAS(status, 0, code);
                break;
            }
        }
        if (ACQ) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();
}

// The same wait with OPTIONAL words (lanes with opt set): the wait ends when every other lane's word has reached its value; the
// return value (the same in every thread, through LDS word `bcw`) says whether the optional ones had as well at that moment.
__device__ __forceinline__ int wg_wait_flags_opt(const int32_t *addr, int need, bool opt, int code, int32_t *status, int tid, int32_t *bcw) {
    if (tid < 64) {
        bool ok = (addr == nullptr);
        int spins = 0;
        for (;;) {
            if (!ok) ok = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
            if (__all(ok || opt)) break;
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if (spins > MEGA_SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (tid == 0) atomicCAS(status, 0, code);
                break;
            }
        }
        const bool all_ok = __all(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) *bcw = all_ok ? 1 : 0;
    }
    __syncthreads();
    return *bcw;
}

// acc (column-block layout) -= P P^T for the 32-column half of P staged MK in As
__device__ __forceinline__ void mfma_cb_half(const double *As, d4 (&acc)[4], int wave, int lane) {
    const vlds_double *Ap = (const vlds_double *)As;
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
        const double bv = -Ap[(16 * wave + m) * LDM + k0 + kk];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double av = Ap[(16 * t + m) * LDM + k0 + kk];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void mega_chain(const MegaArgs &g, const int b, double *As, double *Bs, DiagShared &sh, double *scratch) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lu = lane >> 4;
    const int crow = 16 * wave + lu, ccol = lm;
    const int64_t ld = g.ld;
    double *A = g.buf[BUF_A] + (int64_t)b * g.stride_a;
    int32_t *ver = g.ver + (int64_t)b * VER_PLANES * g.nbk * g.nbk;   // buffer A's versions first
    int32_t *wflag = &g.sync->wflag[b].v;
    long long *tr = (g.trace && b == 0 && tid == 0) ? g.trace : nullptr;
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) __hip_atomic_store(&g.sync->cukey[b], cu_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tr) __hip_atomic_store(&g.trace[5999], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Tile64 D;
    load_cb_lower(D.v, A, ld, wave, lane, scratch);
    __syncthreads();   // (scratch and the factor's LDS share Bs)
    double logdet = 0.0;
    int info = 0;
    for (int k = 0; k < g.nbk; ++k) {
        if (tr) tr[16 * k + 0] = wall_clock64();
        const int64_t rem = g.n - (int64_t)k * 64;
        const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        double *Wk = g.ws + (int64_t)b * g.stride_ws + (int64_t)k * 4096;
        Tile64 winv;
        // (whether the workers' tiles for the next block have arrived is asked during the factorisation: DiagPoll)
        const int2 need = g.chain_need[k < g.nbk - 1 ? k : 0];
        DiagPoll poll;
        // A[k+1][k] (lane 0), A[k+1][k+1] (lane 1); no poll in the last block (a null word: the struct itself is always handed
        // over -- a select between its address and null kept it in scratch memory)
        poll.f = k + 1 < g.nbk ? ver + (k + 1) * g.nbk + k + (lane == 0 ? 0 : 1) : nullptr;
        poll.need = lane == 0 ? need.x : need.y;
        const int bad = diag_factor(D, winv, sh, ncol, (g.trace && b == 0) ? g.trace + 1024 + 16 * k : nullptr,
                                    &poll);
        const int ready = sh.ready == 3;   // (read before Bs, which holds the factor's LDS, is reused)
        if (tr) tr[16 * k + 1] = wall_clock64();
        diag_store_inverse<true>(winv, Wk);
        wg_publish(wflag, k + 1, tid);   // the panel tasks of this block can run
        if (tr) tr[16 * k + 2] = wall_clock64();
        diag_store_factor(D, A + ((int64_t)k * 64) * ld + (int64_t)k * 64, ld, ncol);   // nobody's input
        if (bad && !info) info = k * 64 + bad;
        if (tid < 64) {
            const double pv = sh.piv[tid];
            double v = (tid < ncol && pv > 0.0) ? log(pv) : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            logdet += v;
        }
        if (k + 1 == g.nbk) break;
        // ---- the next block's inputs from the workers: A[k+1][k] and A[k+1][k+1] with the panels <= k-1 applied ----
        if (!ready) {
            const int32_t *addr = nullptr;
            int nd = 0;
            if (tid == 0) { addr = ver + (k + 1) * g.nbk + k; nd = need.x; }
            if (tid == 1) { addr = ver + (k + 1) * g.nbk + k + 1; nd = need.y; }
            wg_wait_flags<true>(addr, nd, 100 + k, &g.sync->status, tid);
            if (__hip_atomic_load(&g.sync->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // (a lost hand-off ends the launch)
        }
        if (tr) tr[16 * k + 3] = wall_clock64();
        double *Pg = A + ((int64_t)(k + 1) * 64) * ld + (int64_t)k * 64;
        const double *Cn = A + ((int64_t)(k + 1) * 64) * ld + (int64_t)(k + 1) * 64;
        const HalfTile a0 = fetch_mk_sc1(Pg, ld, tid, 0), a1 = fetch_mk_sc1(Pg, ld, tid, 1);
        // ---- P = A[k+1][k] W_k^T, W_k straight from the registers ----
        d4 P[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) P[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            commit_mk(h == 0 ? a0 : a1, As, tid);
            if ((wave >> 1) == h) {
#pragma unroll
                for (int I = 0; I < 4; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Bs[(16 * I + lu + 4 * r) * LDM + 16 * (wave & 1) + lm] = (I >= wave) ? winv.v[I][r] : 0.0;
            }
            __syncthreads();
            // W_k = L_kk^-1 is LOWER triangular: column tile t of P = A W_k^T takes the 16-blocks kb <= t of the sum only
            // (40 instead of 64 MFMAs per wave: the f64 MFMA runs at the vector rate on gfx950, so the skipped zeros are time)
            {
                const vlds_double *Ap = (const vlds_double *)As, *Bp = (const vlds_double *)Bs;
                const int m_ = lane & 15, kk_ = lane >> 4;
#pragma unroll
                for (int k0 = 0; k0 < KC; k0 += 4) {
                    const int kb = 2 * h + (k0 >> 4);
                    const double av = Ap[(16 * wave + m_) * LDM + k0 + kk_];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (t >= kb) P[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bp[(16 * t + m_) * LDM + k0 + kk_], P[t], 0, 0, 0);
                }
            }
        }
        if (tr) tr[16 * k + 4] = wall_clock64();
        store_acc_sc1(P, Pg, ld, crow, ccol);
        __syncthreads();   // (Bs, which holds the transposition scratch, is no longer read)
        Tile64 Dn;         // loaded only now: the registers of W_k and of the panel tile's halves are free again
        load_cb_lower<true>(Dn.v, Cn, ld, wave, lane, scratch);
        // ---- D = A[k+1][k+1] - P P^T in the column-block layout; the panel tile is published between the halves ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // P's stores have drained by now
            __syncthreads();
            if (h == 1 && tid == 0)
                __hip_atomic_store(ver + (k + 1) * g.nbk + k, VER_FINAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) As[(crow + 4 * r) * LDM + 16 * t + ccol] = P[2 * h + t][r];
            __syncthreads();
            mfma_cb_half(As, Dn.v, wave, lane);
        }
        D = Dn;
        if (tr) tr[16 * k + 5] = wall_clock64();
    }
    if (tid == 0) {
        __hip_atomic_store(g.logdet + b, logdet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g.info + b, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the chain, second form (round 4) -------------------------------------------------------------------------------
// The same arithmetic as mega_chain in the same order (bit-identical results), re-arranged around what bounded the block
// step (profiles/r03_potrf_phase_trace.txt: 7.0 factor + 0.8 publish + 2.0 wait + 1.7 solve + 3.5 update):
//   * the solve produces P^T in the accumulator layout, P^T = W_k Q^T (operands swapped: the same products in the same order).
//     An accumulator tile is, register for register, an MFMA operand of the next product (the layout duality diagfac.hpp
//     uses), so the update D -= P P^T takes its B operand straight from the wave's own registers and its A operand from the
//     other waves' registers, exchanged through LDS AS THEY STAND: one barrier, no transposition, no staging by halves;
//   * Q and W_k are staged whole (LDQ = 66), so the 40 MFMAs of the solve run without a barrier in between;
//   * only the tiles t <= w of the next diagonal block are updated (the factor reads no others), and NO barrier separates
//     the update from the next factorisation: wave 0 (16 MFMAs) starts eliminating while wave 3 (64) is still updating --
//     it is not needed before the first 16-block's barrier;
//   * W_k is published behind the wait for the panel tile's loads (vmcnt counts in order: the older W stores have drained
//     when the younger loads have landed), not behind a drain of its own;
//   * the panel tile's version word, the log-determinant of block k and the drain of the panel stores ride in the shadow
//     of the next factorisation's first elimination (DiagShadow).
#define LDQ 66   // LDS leading dimension of a whole 64x64 f64 tile: bank(4 row + 2 k), conflict-free ds_read_b64 fragments
// The lower factor L = U^T of a block factored by diag_factor, stored ROW-contiguously: wave w writes rows 16w .. 16w + 15
// of the tile (the 16x16 tiles (w, t), t <= w, transposed inside the wave through `scratch` -- 16 x 16 doubles of LDS per
// wave -- and zeros right of the diagonal): four full 128-byte lines per store instruction where the transposed 8-byte
// stores of diag_store_factor touch sixteen.  Blocks with carried rows (the last one) take diag_store_factor.
template <int wave>
__device__ __forceinline__ void diag_store_factor_rows(const Tile64 &tile, double *Ab, int64_t ld, int lane, double *scratch) {
    const int lm = lane & 15, lu = lane >> 4;
    vlds_double *S = (vlds_double *)scratch + wave * 256;   // element (i, j) at i * 16 + (j ^ i): both passes conflict-free
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t <= wave) {   // L[16w + lu + 4r][16t + lm] = U[16t + lm][16w + lu + 4r]
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(lu + 4 * r) * 16 + (lm ^ (lu + 4 * r))] = tile.v[t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double trv = S[lm * 16 + ((lu + 4 * r) ^ lm)];   // U[16t + lm][16w + lu + 4r]
                Ab[(int64_t)(16 * wave + lu + 4 * r) * ld + 16 * t + lm] = (t == wave && lm > lu + 4 * r) ? 0.0 : trv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ab[(int64_t)(16 * wave + lu + 4 * r) * ld + 16 * t + lm] = 0.0;
        }
    }
}

// P (64x64, row-major in global memory) from P^T in the accumulator layout -- lane l of wave w holds
// P[16w + lm][16t + lu + 4r] in register (t, r) -- as write-through 16-byte stores: the lanes of DPP rows lu and lu ^ 1 hold
// neighbouring columns, so each pair of registers (r0, r1) is exchanged across the two rows (v_permlane16_swap: odd rows of
// the first operand with even rows of the second): the even row ends up with columns (lu, lu + 1) of r0, the odd row with
// columns (lu - 1, lu) of r1.
__device__ __forceinline__ void store_pt_sc1(const d4 (&P)[4], double *Pg, int64_t ld, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)Pg, 0, 0x7fffffff, 0x00020000);
    const int lm = lane & 15, lu = lane >> 4, odd = lu & 1;
    const int base = (int)(((int64_t)(16 * wave + lm) * ld + (lu & ~1) + 4 * odd) * 8);   // row 16w + lm, column (lu & ~1) + 4 odd
    // All eight 16-byte values first, each in registers of its own, then the eight stores, then wait states before any of the
    // registers may be rewritten.  (The first form built every value in the same four registers; the compiler rewrote them right
    // behind each store -- with the store's offset in an SGPR it inserts no wait state, LLVM's rule for stores of more than 8
    // bytes -- and with the memory pipeline under load from other processes the store had not read all of its data yet: the first
    // double of the lanes read last held the NEXT pair's value.  Two ranks sharing one GPU: a few wrong factors per thousand
    // launches, tools/gpu_mega_stress_shared.sh; never seen with the device to itself.)
    // Round 5: the constant part of the address rides in the instruction's immediate offset field (voffset + constant) and the soffset
    // operand is the literal 0 -- the form for which LLVM's hazard recogniser inserts the wait states itself (it skips a store whose
    // soffset is a register: the constants above 64 used to be materialised in SGPRs).  The registers of their own and the
    // wait states below stay; tests/test_isa_hazards.py reads the shipped code and fails when a data register of any 16-byte store
    // of the kernel is rewritten too early, or when a store's soffset is a register again.
    // (-DMEGA_DIAG_OLD_STORE=1: round 4's first form -- one set of data registers, offsets in SGPRs -- for the one-fix-at-a-time
    //  stress of profiles/r05_hazard_one_fix_at_a_time.txt; never in a shipped build)
#if defined(MEGA_DIAG_OLD_STORE) && MEGA_DIAG_OLD_STORE
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const double x = P[t][2 * rp], y = P[t][2 * rp + 1];
            const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
            u32x4 v1;
            v1.x = lo[0]; v1.y = hi[0]; v1.z = lo[1]; v1.w = hi[1];
            __builtin_amdgcn_raw_buffer_store_b128(v1, rs, base, (16 * t + 8 * rp) * 8, 16);
        }
    return;
#endif
    u32x4 v[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const double x = P[t][2 * rp], y = P[t][2 * rp + 1];
            const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
            v[2 * t + rp].x = lo[0]; v[2 * t + rp].y = hi[0]; v[2 * t + rp].z = lo[1]; v[2 * t + rp].w = hi[1];
        }
#pragma unroll
    for (int q = 0; q < 8; ++q)   // (every value is complete, in registers of its own, before the first store is issued)
        asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
#pragma unroll
    for (int q = 0; q < 8; ++q)
        // (one per-lane base offset; the tile / register-pair part is a constant that folds into the instruction's offset field)
        __builtin_amdgcn_raw_buffer_store_b128(v[q], rs, base + (16 * (q >> 1) + 8 * (q & 1)) * 8, 0, 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)   // (alive until here: four more stores have been issued behind each of these)
        asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
    // (the last four values stay untouched for sixteen more wait states: the registers are operands of the statement that waits)
    asm volatile("s_nop 7\n\ts_nop 7"
                 : "+v"(v[4].x), "+v"(v[4].y), "+v"(v[4].z), "+v"(v[4].w), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[5].z), "+v"(v[5].w),
                   "+v"(v[6].x), "+v"(v[6].y), "+v"(v[6].z), "+v"(v[6].w), "+v"(v[7].x), "+v"(v[7].y), "+v"(v[7].z), "+v"(v[7].w)
                 :: "memory");
}

// ---- the per-wave parts of the chain's block step as static programs (W = wave: every role decision is a compile-time
// constant; left as run-time conditions on the wave index they became a basic block per instruction, each with its own
// full wait) ----
// W_k -> LDS, row-major (the blocks on and below the diagonal: the solve reads no others)
template <int W>
__device__ __forceinline__ void chain_stage_w(const Tile64 &winv, double *Ws, int lm, int lu) {
#pragma unroll
    for (int I = W; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ws[(16 * I + lu + 4 * r) * LDQ + 16 * W + lm] = winv.v[I][r];
}
// The next diagonal tile's request (lower 16x16 tiles of row block W, row-contiguous loads).  Its update is dealt so that
// waves 1-3 carry three tiles each and wave 0, which starts the next elimination, one: tile (0, 3) of wave 3's column
// block is computed by wave 1 -- in wave 3's layout, loaded transposed straight from the lower triangle (H) -- and handed
// over through LDS behind the next factorisation's first barrier.
template <int W>
__device__ __forceinline__ void chain_request_next(d4 (&nat)[4], d4 &H, const double *Cn, int64_t ld, int lm, int lu) {
#pragma unroll
    for (int t = (W == 3 ? 1 : 0); t <= W; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            nat[t][r] = __hip_atomic_load(Cn + (int64_t)(16 * W + lu + 4 * r) * ld + 16 * t + lm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (W == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            H[r] = __hip_atomic_load(Cn + (int64_t)(48 + lm) * ld + lu + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void store_pt_sc1(const d4 (&P)[4], double *Pg, int64_t ld, int wave, int lane);
// D = A[k+1][k+1] - P P^T in the column-block layout: wave W's tiles (see chain_request_next), the operands of the other
// waves from the exchange buffer Xc two k-steps ahead of their MFMAs (one wave per SIMD: a read waited for in front of
// its MFMA exposes the LDS latency, ~90 cycles per 64-cycle MFMA).
template <int W>
__device__ __forceinline__ void chain_update(Tile64 &Dn, const d4 (&P)[4], const d4 (&nat)[4], d4 &H, vlds_double *Xc, double *scratch,
                                             vlds_double *hand, double *Pg, int64_t ld, int lane, int *pcount, int ptarget, int32_t *pflag) {
    const int lm = lane & 15, lu = lane >> 4;
    constexpr int T0 = (W == 3 ? 1 : 0);   // wave 3's tile 0 comes from wave 1
    // The panel tile, final, as write-through stores.  Wave 0 goes straight on to the next elimination and stores nothing:
    // its rows are wave 1's (from the exchange buffer: the same registers, lane for lane).
    if (W != 0) store_pt_sc1(P, Pg, ld, W, lane);
    if (W == 1) {
        d4 P0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) P0[t][r] = Xc[((0 * 4 + t) * 4 + r) * 64 + lane];
        store_pt_sc1(P0, Pg, ld, 0, lane);
    }
    vlds_double *S = (vlds_double *)scratch + W * 256;   // element (i, j) at i * 16 + (j ^ i): both passes conflict-free
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= T0 && t <= W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(lu + 4 * r) * 16 + (lm ^ (lu + 4 * r))] = nat[t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double trv = S[lm * 16 + ((lu + 4 * r) ^ lm)];
                Dn.v[t][r] = (t == W && lu + 4 * r >= lm) ? nat[t][r] : trv;
            }
        } else {
            Dn.v[t] = (d4){0.0, 0.0, 0.0, 0.0};
        }
    }
    if (W != 0) {
        // The tile's version word as soon as the stores have drained: waves 1-3 count themselves in LDS when theirs have (the
        // wait costs them ~0.5 us they have -- wave 0 eliminates for 1.2 us after its single tile), the last one publishes.
        // (Round 4's first form published from the next factorisation's first barrier, 1 us later: the look-ahead task waits
        // for this word before it can update the tile the chain needs next.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && atomicAdd(pcount, 1) == ptarget)
            __hip_atomic_store(pflag, VER_FINAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double xa[3][4], xh[3];   // a ring of three k-steps (sched_barrier: the scheduler must not gather all the reads up front)
    auto rd = [&](int s_) {
        const int o = s_ % 3;
#pragma unroll
        for (int t = T0; t < W; ++t) xa[o][t] = Xc[(t * 16 + s_) * 64 + lane];
        if (W == 1) xh[o] = Xc[(3 * 16 + s_) * 64 + lane];
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {   // k-step (c, r) = (s_ / 4, s_ % 4): columns 16 c + 4 r .. + 3 of the panel tile
        if (s_ + 2 < 16) rd(s_ + 2);
        __builtin_amdgcn_sched_barrier(0);
        const double pv = P[s_ >> 2][s_ & 3], bv = -pv;
#pragma unroll
        for (int t = T0; t <= W; ++t)
            Dn.v[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(t == W ? pv : xa[s_ % 3][t], bv, Dn.v[t], 0, 0, 0);
        if (W == 1)   // tile (0, 3): rows of block 0 against the rows of block 3, both from the exchange buffer
            H = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s_ % 3][0], -xh[s_ % 3], H, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (W == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hand[r * 64 + lane] = H[r];
    }
}

__device__ __forceinline__ void mega_chain2(const MegaArgs &g, const int b, double *Qs, double *Ws, DiagShared &sh) {
    // (wave in a scalar register: the per-wave roles below are then scalar branches, not sixteen exec-masked blocks)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    const int64_t ld = g.ld;
    double *A = g.buf[BUF_A] + (int64_t)b * g.stride_a;
    int32_t *ver = g.ver + (int64_t)b * VER_PLANES * g.nbk * g.nbk;   // buffer A's versions first
    int32_t *wflag = &g.sync->wflag[b].v;
    long long *tr = (g.trace && b == 0 && tid == 0) ? g.trace : nullptr;
    long long *tl = (g.tlog && tid == 0) ? g.tlog + 64 + 8 * (int64_t)b * g.nbk : nullptr;   // (full log: every matrix)
    if (tl && b == 0) g.tlog[0] = wall_clock64();
    double *scratch = &sh.u[0][0][0];   // (wave-private 16 x 16 transposition areas; the factor writes sh.u only behind its first barrier)
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) __hip_atomic_store(&g.sync->cukey[b], cu_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tr) __hip_atomic_store(&g.trace[5999], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Tile64 D;
    load_cb_lower(D.v, A, ld, wave, lane0, scratch);
    DiagShadow shw;
    shw.hand = nullptr; shw.wout = nullptr;
    if (tid == 0) {
        sh.pcount = 0;   // (visible to the other waves behind the first factorisation's barriers)
        sh.look = 0;
    }
    double *pivots = g.piv + (int64_t)b * g.ld;   // every block's pivots: their logarithms are summed when the chain has ended
    int info = 0;
    vlds_double *hand = (vlds_double *)Ws;   // (Ws is idle between the solve and the next block's W_k)
    const vlds_double *Qp = (const vlds_double *)Qs, *Wp = (const vlds_double *)Ws;
    vlds_double *Xc = (vlds_double *)Qs;   // exchange of the P^T registers: Xc[((w * 4 + t) * 4 + r) * 64 + lane] (32 KB, over Qs)
    for (int k = 0; k < g.nbk; ++k) {
        // (opaque to the optimiser: the dozens of per-lane offsets below are loop invariants that would otherwise be hoisted
        //  out of the block loop and held in registers across the factorisation, where the pressure peaks -- spills)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int lm = lane & 15, lu = lane >> 4;
        if (tr) tr[16 * k + 0] = wall_clock64();
        if (tl) tl[8 * k + 0] = wall_clock64();
        const int64_t rem = g.n - (int64_t)k * 64;
        const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        double *Wk = g.ws + (int64_t)b * g.stride_ws + (int64_t)k * 4096;
        Tile64 winv;
        // (whether the workers' tiles for the next block have arrived is asked during the factorisation: DiagPoll)
        const int2 need = g.chain_need[k < g.nbk - 1 ? k : 0];
        DiagPoll poll;
        // A[k+1][k] (lane 0), A[k+1][k+1] (lane 1); no poll in the last block (a null word: the struct itself is always handed
        // over -- a select between its address and null kept it in scratch memory)
        poll.f = k + 1 < g.nbk ? ver + (k + 1) * g.nbk + k + (lane == 0 ? 0 : 1) : nullptr;
        poll.need = lane == 0 ? need.x : need.y;
        const int bad = diag_factor(D, winv, sh, ncol, (g.trace && b == 0) ? g.trace + 1024 + 16 * k : nullptr,
                                    &poll, &shw);
        const int ready = g.nowait ? 3 : sh.ready;
        if (tr) tr[16 * k + 1] = wall_clock64();
        if (tl) tl[8 * k + 1] = wall_clock64();
        if (bad && !info) info = k * 64 + bad;
        if (wave == 3) __hip_atomic_store(pivots + 64 * k + lane, sh.piv[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        shw.hand = nullptr;
        // (W_k row block by row block from inside the factorisation -- DiagShadow::wout -- published it 0.4 us earlier and cost
        //  the factorisation 0.8 us: measured, not used)
        diag_store_inverse<true>(winv, Wk);
        if (k + 1 == g.nbk) {
            diag_store_factor(D, A + ((int64_t)k * 64) * ld + (int64_t)k * 64, ld, ncol);
            wg_publish(wflag, k + 1, tid);   // (the last block's inverse: T tasks of the fused inverse)
            break;
        }
        // ---- the next block's inputs from the workers: A[k+1][k] and A[k+1][k+1] with the panels <= k-1 applied ----
        // (bit 0: the panel tile A[k+1][k] had arrived when the factorisation polled, bit 1: the diagonal tile A[k+1][k+1] --
        //  the second is needed one solve later and is waited for separately)
        // A tile that had not arrived at the factorisation's poll usually has by now (the look-ahead tasks publish the panel tile
        // ~10 us after W_k-1, the poll is 1.7 us before this point): ONE more look -- a write-through load of the word by one
        // lane, handed to the others through LDS -- and the step goes on as if the poll had seen it.  No acquire here: every
        // load of these tiles below bypasses L1 (sc1), every store of them was write-through and drained ahead of the word,
        // and this CU holds the chain alone (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire, first row).
        // (The bits go into a word of their own.  Written into sh.ready they raced with the other waves' read of it above: nothing
        //  orders a slow wave's read before wave 0's update, the waves then disagreed about `ready != 3`, one side took a barrier
        //  the other did not, and from there on every LDS hand-over of the step was off by one barrier.  With the chain alone on
        //  its CU the four waves run in step and it never happened; with another process's waves on the same SIMDs -- two ranks
        //  sharing a GPU -- it gave wrong factors a few times per thousand launches.)
        int ready2 = ready;
        if (ready != 3) {
            if (tid < 2 && !((ready >> tid) & 1)) {
                const int got = __hip_atomic_load(ver + (k + 1) * g.nbk + k + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(MEGA_DIAG_OLD_LOOK) && MEGA_DIAG_OLD_LOOK   // (round 4's first form, for the one-fix-at-a-time stress only)
                if (got >= (tid == 0 ? need.x : need.y)) atomicOr(&sh.ready, 1 << tid);
            }
            lds_barrier();
            ready2 = g.nowait ? 3 : sh.ready;
#else
                if (got >= (tid == 0 ? need.x : need.y)) atomicOr(&sh.look, 1 << tid);
            }
            lds_barrier();
            ready2 = g.nowait ? 3 : (ready | sh.look);
#endif
        }
        bool published = false;
        if (!(ready2 & 1)) {
            wg_publish(wflag, k + 1, tid);   // (the panel tasks of this block must not wait for the chain's own inputs)
            published = true;
            wg_wait_flags<true>(tid == 0 ? ver + (k + 1) * g.nbk + k : nullptr, need.x, 100 + k, &g.sync->status, tid);
            if (__hip_atomic_load(&g.sync->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // (a lost hand-off ends the launch)
        }
        double *Pg = A + ((int64_t)(k + 1) * 64) * ld + (int64_t)k * 64;
        const double *Cn = A + ((int64_t)(k + 1) * 64) * ld + (int64_t)(k + 1) * 64;
        const HalfTile a0 = fetch_mk_sc1(Pg, ld, tid, 0), a1 = fetch_mk_sc1(Pg, ld, tid, 1);
        // while the panel tile is on its way: W_k into LDS, and the factor's own tile (nobody's input) to memory
        {
            double *Akk = A + ((int64_t)k * 64) * ld + (int64_t)k * 64;
            switch (wave) {
                case 0: chain_stage_w<0>(winv, Ws, lm, lu); diag_store_factor_rows<0>(D, Akk, ld, lane, scratch); break;
                case 1: chain_stage_w<1>(winv, Ws, lm, lu); diag_store_factor_rows<1>(D, Akk, ld, lane, scratch); break;
                case 2: chain_stage_w<2>(winv, Ws, lm, lu); diag_store_factor_rows<2>(D, Akk, ld, lane, scratch); break;
                default: chain_stage_w<3>(winv, Ws, lm, lu); diag_store_factor_rows<3>(D, Akk, ld, lane, scratch); break;
            }
        }
        // the panel tile has landed -- and with it, vmcnt counting in order, this wave's older W_k stores have drained
        if (tl) tl[8 * k + 7] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const int c2 = (tid & 15) * 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = (tid >> 4) + 16 * it;
                Qs[r * LDQ + c2] = a0.v[it].x; Qs[r * LDQ + c2 + 1] = a0.v[it].y;
                Qs[r * LDQ + 32 + c2] = a1.v[it].x; Qs[r * LDQ + 32 + c2 + 1] = a1.v[it].y;
            }
        }
        lds_barrier();
        if (tid == 0) sh.look = 0;   // (every wave has read it: the next block's second look starts from zero, three barriers from here)
        if (tid == 0 && !published) __hip_atomic_store(wflag, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the panel tasks of this block can run
        if (tr) tr[16 * k + 2] = tr[16 * k + 3] = wall_clock64();
        if (tl) { tl[8 * k + 2] = wall_clock64(); tl[8 * k + 6] = ready2; }
        if (!(ready2 & 2)) {
            wg_wait_flags<true>(tid == 0 ? ver + (k + 1) * g.nbk + k + 1 : nullptr, need.y, 200 + k, &g.sync->status, tid);
            if (__hip_atomic_load(&g.sync->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        // the next diagonal tile: requested now, in flight during the solve
        d4 nat[4];
        d4 H = (d4){0.0, 0.0, 0.0, 0.0};
        switch (wave) {
            case 0: chain_request_next<0>(nat, H, Cn, ld, lm, lu); break;
            case 1: chain_request_next<1>(nat, H, Cn, ld, lm, lu); break;
            case 2: chain_request_next<2>(nat, H, Cn, ld, lm, lu); break;
            default: chain_request_next<3>(nat, H, Cn, ld, lm, lu); break;
        }
        if (tr) tr[16 * k + 6] = wall_clock64();
        if (tl) tl[8 * k + 3] = wall_clock64();
        // ---- P^T = W_k Q^T: tile t of wave w is P[16w.., 16t..]^T; W_k is LOWER triangular, so tile t sums the 16-blocks kb <= t ----
        d4 P[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) P[t] = (d4){0.0, 0.0, 0.0, 0.0};
        // (one wave per SIMD: an operand read waited for in front of its MFMA exposes the LDS latency, ~90 cycles per 64-cycle
        //  MFMA -- the fragments of k-step s + 2 are requested before the MFMAs of step s are issued)
        {
            double qv[3], wv[3][4];   // a ring of three k-steps (sched_barrier: the scheduler must not gather all the reads up front)
            auto rd = [&](int s) {
                const int k0 = 4 * s, kb = s >> 2, o = s % 3;
                qv[o] = Qp[(16 * wave + lm) * LDQ + k0 + lu];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t >= kb) wv[o][t] = Wp[(16 * t + lm) * LDQ + k0 + lu];
            };
            rd(0);
            rd(1);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 2 < 16) rd(s + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t >= (s >> 2)) P[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[s % 3][t], qv[s % 3], P[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tr) tr[16 * k + 4] = wall_clock64();
        if (tl) tl[8 * k + 4] = wall_clock64();
        lds_barrier();   // (Qs / Ws are no longer read: the exchange buffer lies over Qs)
        if (tr) tr[16 * k + 7] = wall_clock64();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xc[((wave * 4 + t) * 4 + r) * 64 + lane] = P[t][r];
        shw.hand = (const double *)Ws;
        lds_barrier();
        if (tr) tr[16 * k + 8] = wall_clock64();
        // ---- D = A[k+1][k+1] - P P^T in the column-block layout ----
        Tile64 Dn;
        switch (wave) {
            case 0: chain_update<0>(Dn, P, nat, H, Xc, scratch, hand, Pg, ld, lane, &sh.pcount, 3 * k + 2, ver + (k + 1) * g.nbk + k); break;
            case 1: chain_update<1>(Dn, P, nat, H, Xc, scratch, hand, Pg, ld, lane, &sh.pcount, 3 * k + 2, ver + (k + 1) * g.nbk + k); break;
            case 2: chain_update<2>(Dn, P, nat, H, Xc, scratch, hand, Pg, ld, lane, &sh.pcount, 3 * k + 2, ver + (k + 1) * g.nbk + k); break;
            default: chain_update<3>(Dn, P, nat, H, Xc, scratch, hand, Pg, ld, lane, &sh.pcount, 3 * k + 2, ver + (k + 1) * g.nbk + k); break;
        }
        D = Dn;
        if (tr) tr[16 * k + 5] = wall_clock64();
        if (tl) tl[8 * k + 5] = wall_clock64();
    }
    // The log-determinant: per 64-block the logarithms of its pivots summed over the lanes (the same tree, then the same
    // running sum over the blocks as diag_logdet: the same bits).  Kept out of the block steps -- the double-precision
    // logarithm costs the wave that evaluates it ~0.5 us and twenty registers of polynomial constants that the compiler
    // holds across the whole loop -- and done here for all blocks at once, the four waves taking every fourth block.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    vlds_double *vs = (vlds_double *)Qs;
    for (int kk = wave; kk < g.nbk; kk += 4) {
        const int64_t rem = g.n - (int64_t)kk * 64;
        const int ncol = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        const double pv = __hip_atomic_load(pivots + 64 * kk + lane0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double v = (lane0 < ncol && pv > 0.0) ? log(pv) : 0.0;   // (a failed pivot is reported through info)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane0 == 0) vs[kk] = v;
    }
    lds_barrier();
    if (tid == 0) {
        double logdet = 0.0;
        for (int kk = 0; kk < g.nbk; ++kk) logdet += vs[kk];
        __hip_atomic_store(g.logdet + b, logdet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g.info + b, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // (Qs becomes a worker's staging tile next)
}

// One ticket for a worker (called by thread 0 only): of the critical queue, of the bulk queue with bit 30 set, or -1 when both queues
// of the group are empty.  role 0 asks the critical queue first, role 1 the bulk queue.  No look at the heads before the atomic:
// the words are the hottest lines of the launch and a load of one costs as much as the atomic itself; a queue found empty once
// is not asked again (qempty).  Not inlined: six call sites, and its state must not live in the worker loop's registers.
__device__ __noinline__ int mega_pull(MegaSync *sync, int grp, int nbg, int nut, int ntask, int role, int32_t *qempty) {
    const int ucap = nut * nbg, bcap = (ntask - nut) * nbg;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int eu = qempty[0], eb = qempty[1];
        const int pick = (role == 0 ? !eu : eb) ? 0 : 1;   // 0: critical queue
        if (pick == 0 ? eu : eb) break;
        const int t = __hip_atomic_fetch_add(pick == 0 ? &sync->uhead[grp].v : &sync->ghead[grp].v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < (pick == 0 ? ucap : bcap)) return pick == 0 ? t : (t | (1 << 30));
        qempty[pick] = 1;
    }
    return -1;
}

#ifndef MEGA_V2
#define MEGA_V2 1   // 0: the chain of rounds 2-3 (mega_chain), kept for same-box comparisons
#endif
__global__ __launch_bounds__(256, 2) void potrf_mega_kernel(MegaArgs g) {
    __shared__ double tiles[2 * 64 * LDM];   // As | Bs (workers); the chain's whole panel tile Q, then its register exchange
    __shared__ int32_t bc[4];
    double *As = tiles, *Bs = tiles + 64 * LDM;
#if MEGA_V2
    // the chain keeps W_k and the factor's LDS beside the panel tile (79.8 KB per workgroup: two workgroups per CU still fit 160 KB)
    __shared__ double wtile[64 * LDQ];
    __shared__ DiagShared dsh;
    static_assert(64 * LDQ <= 2 * 64 * LDM, "the chain's panel tile must fit the workers' staging tiles");
    static_assert(2 * (sizeof(double) * (2 * 64 * LDM + 64 * LDQ) + sizeof(DiagShared) + 16) <= 160 * 1024, "two workgroups per CU");
#else
    DiagShared &sh = *reinterpret_cast<DiagShared *>(Bs);   // (idle while a chain workgroup factors)
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int crow = 16 * wave + (lane >> 4), ccol = lane & 15;
    const int batch = g.batch;
    if (g.pred && *g.pred) return;   // (a speculative batch that an earlier one has made unnecessary: dgpamd_ess_queue)
    if ((int)blockIdx.x < batch) {
#if MEGA_V2
        mega_chain2(g, blockIdx.x, tiles, wtile, dsh);
#else
        mega_chain(g, blockIdx.x, As, Bs, sh, &sh.u[0][0][0]);
#endif
    } else {
        // a worker beside a chain leaves (bounded wait for the chains' keys: they are dispatched first)
        if (tid < 64) {
            int key = 0, spins = 0;
            const bool mine = tid < batch;
            while (mine && (key = __hip_atomic_load(&g.sync->cukey[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < 2000)
                __builtin_amdgcn_s_sleep(4);
            const bool beside = __any(mine && key == cu_key());
            if (tid == 0) bc[0] = beside ? 1 : 0;
        }
        __syncthreads();
        if (bc[0]) return;
    }
    // ---- worker loop (the chains join it when their matrix is factored) ----
    const int64_t ld = g.ld;
    const int64_t rem_last = g.n - (int64_t)(g.nbk - 1) * 64;
    const int ncol_last = rem_last > 0 ? (int)rem_last : 0;
    // debug (dgpamd_debug_trace): the first 80 tasks of one worker, 5 stamps each, from trace[2048]: pulled, inputs ready,
    // computed, stored, published (+ the task's kind / panels in the sixth word)
    long long *wst = (g.trace && tid == 0 && (int)blockIdx.x == batch + 40) ? g.trace + 2048 : nullptr;
    int nst = 0;
    // this workgroup's group of matrices: those of its XCD's group first, then -- when that queue is empty -- the others'
    const int G = g.ngroups;
    const int home = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) % G;   // HW_REG_XCC_ID
    int grp = home, tried = 0;
    auto group_size = [&](int gi) { return (batch - gi + G - 1) / G; };
    // Two queues per group of matrices.  CRITICAL: per block step the look-ahead pair, the first solve of the panel column and the
    // catch-up updates behind it -- the lane that hands the chain its next inputs (six tasks per block step and matrix).  It is
    // served by workers of its own (`ncrit` per matrix, taken from the group's XCDs), which pull it in order and wait inside
    // their tasks.  With one queue these tasks sat behind what was left of the previous block step's bulk whenever the engine was
    // busy (tools/analyze_tasklog.py: "pulled late", 12 us per look-ahead task at ten matrices, and the chain's loop -- factor,
    // publish, look-ahead -- was 27 us per block step where the chain alone takes 12.6).  BULK: everything else, in the old order,
    // for all other workers; each kind of worker helps with the other queue once its own is empty.  Both queues are topological
    // orders of their own tasks and each has workers that take nothing else first, so the earliest unfinished task of the launch
    // is always held, or about to be pulled, by a worker that can run it.
    // The pull of the NEXT task is issued behind the stores of a task's tile and lands while they drain.
    const int widx = (int)blockIdx.x - batch;
    // (never more than a quarter of an XCD's ~64 workers: with 32-64 matrices `ncrit` per matrix would leave nobody who takes bulk work first,
    //  and critical workers that run ahead wait inside their tasks for bulk results nobody computes)
    const int crit_want = g.nut > 0 ? (g.ncrit * group_size(home) * G + 7) / 8 : 0;
    const int crit_per_xcd = crit_want > 16 ? 16 : crit_want;
    const int role = (widx >= 0 && (widx >> 3) < crit_per_xcd) ? 0 : 1;   // 0: critical queue first, 1: bulk queue first
    __shared__ int32_t qempty[2];   // (thread 0's notes: this group's critical / bulk queue has been found empty)
    if (tid == 0) { qempty[0] = g.nut > 0 ? 0 : 1; qempty[1] = 0; }
    auto pull_take = [&]() -> int { return mega_pull(g.sync, grp, group_size(grp), g.nut, g.ntask, role, qempty); };
    if (tid == 0) bc[1] = pull_take();
    for (;;) {
        __syncthreads();
        int q = __builtin_amdgcn_readfirstlane(bc[1]);
        __syncthreads();
        while (q < 0) {   // this group's queues are empty: on to the next group (all empty: done)
            if (++tried >= G) break;
            grp = grp + 1 == G ? 0 : grp + 1;
            if (tid == 0) {
                qempty[0] = g.nut > 0 ? 0 : 1;
                qempty[1] = 0;
                bc[1] = pull_take();
            }
            __syncthreads();
            q = __builtin_amdgcn_readfirstlane(bc[1]);
            __syncthreads();
        }
        if (tried >= G) break;
        const int nbg = group_size(grp);
        const int qt = q & ~(1 << 30), qs = qt / nbg;
        const int slot = (q >> 30) ? g.nut + qs : qs, b = grp + G * (qt - qs * nbg);
        const MTask mt = g.tasks[slot];
        long long *st = (wst && nst < 80) ? wst + 8 * nst++ : nullptr;
        // debug: every worker adds its waiting / arithmetic / remaining time to 100-us buckets of the launch (trace[6000..6095])
        const bool agg = g.trace != nullptr && tid == 0;
        long long a0 = 0, a1 = 0, a2 = 0;
        if (agg) a0 = wall_clock64();
        // debug (dgpamd_debug_tasklog): pulled, inputs seen, arithmetic done, W_k seen, stored, published, workgroup
        long long *tl = (g.tlog && tid == 0) ? g.tlog + 64 + 8 * ((int64_t)batch * g.nbk + (int64_t)b * g.ntask + slot) : nullptr;
        if (tl) { tl[0] = wall_clock64(); tl[6] = blockIdx.x; }
        if (st) { st[0] = wall_clock64(); st[5] = (mt.a.x & 15) | ((mt.a.w >> 16) << 8) | ((long long)slot << 16); }
        const int4 tk = mt.a;
        const int post = tk.x & 15, first = (tk.x >> 4) & 1, plus = (tk.x >> 5) & 1, mask_last = (tk.x >> 6) & 1;
        const int bufC = (tk.x >> 8) & 3, bufL = (tk.x >> 10) & 3, bufR = (tk.x >> 12) & 3;
        const int ci = tk.y & 0xffff, cj = tk.y >> 16, li = tk.z & 0xffff, ri = tk.z >> 16;
        const int kb0 = tk.w & 0xffff, nkb = tk.w >> 16;
        const int need_c = mt.b.x, fin = mt.b.y & 1, wk = mt.b.z;
        const int64_t mo = (int64_t)b * g.stride_a;
        const int nb2 = g.nbk * g.nbk;
        int32_t *ver = g.ver + (int64_t)b * VER_PLANES * nb2;
        int32_t *vC = ver + bufC * nb2 + ci * g.nbk + cj;
        double *C = g.buf[bufC] + mo + ((int64_t)ci * 64) * ld + (int64_t)cj * 64;
        double *Wk = g.ws + (int64_t)b * g.stride_ws + (int64_t)wk * 4096;
        int32_t *wflag = &g.sync->wflag[b].v;
        const int newver = fin ? VER_FINAL : need_c + 1;
        int qn = 0;
        auto pull_next = [&]() {
            if (tid == 0) qn = pull_take();
        };
        auto finish = [&](int32_t *vflag, int vvalue) {   // drain (the tile's stores and the pull), publish, hand the next task to the loop's head
            wg_publish(vflag, vvalue, tid);
            if (tid == 0) bc[1] = qn;
            if (st) st[4] = wall_clock64();
            if (tl) tl[5] = wall_clock64();
            if (agg) {
                const long long a4 = wall_clock64(), t00 = g.trace[5999];
                long long bk = t00 > 0 ? (a0 - t00) / 10000 : 0;
                bk = bk < 0 ? 0 : (bk > 31 ? 31 : bk);
                if (a1 == 0) a1 = a2 = a0;   // (a task without separate phases: all of it counts as the rest)
                atomicAdd((unsigned long long *)g.trace + 6000 + bk, (unsigned long long)(a1 - a0));
                atomicAdd((unsigned long long *)g.trace + 6032 + bk, (unsigned long long)(a2 - a1));
                atomicAdd((unsigned long long *)g.trace + 6064 + bk, (unsigned long long)(a4 - a2));
                // ... and by kind of task: [buffer A / T / S][update, solve]: waiting, arithmetic, rest, count (trace[6100..6123])
                const int kd = 4 * (2 * bufC + (post == T_SOLVE ? 1 : 0));
                atomicAdd((unsigned long long *)g.trace + 6100 + kd, (unsigned long long)(a1 - a0));
                atomicAdd((unsigned long long *)g.trace + 6101 + kd, (unsigned long long)(a2 - a1));
                atomicAdd((unsigned long long *)g.trace + 6102 + kd, (unsigned long long)(a4 - a2));
                atomicAdd((unsigned long long *)g.trace + 6103 + kd, 100ull);
            }
        };

        if (post == T_TDIAG) {   // T[k][k] = W_k^T, rows of the carried right-hand sides zeroed
            wg_wait_flags<true>(tid == 0 ? wflag : nullptr, wk + 1, 2000 + wk, &g.sync->status, tid);
            if (tl) tl[1] = tl[2] = tl[3] = wall_clock64();
            const int nrow = (wk == g.nbk - 1) ? ncol_last : 64;
            for (int idx = tid; idx < 4096; idx += 256)
                tiles[(idx >> 6) * 65 + (idx & 63)] = __hip_atomic_load(Wk + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            pull_next();
            for (int idx = tid; idx < 4096; idx += 256) {
                const int r = idx >> 6, c = idx & 63;
                __hip_atomic_store(C + (int64_t)r * ld + c, r < nrow ? tiles[c * 65 + r] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            finish(vC, newver);
            continue;
        }
        // ---- inputs: the output tile's earlier visits, the operand tiles final ----
        // The NEWEST panel of a visit is usually the one that arrives last -- its operand tile was solved one block step ago, and
        // along a row of tiles every solve waits for the previous column's (the recursion that bounds the whole launch when the
        // engine is not saturated: 13-20 us per column with all of a visit's panels applied behind the last arrival).  So the
        // newest panel's words are optional in the first wait: if they have not arrived, the older panels are applied first
        // and only the last panel's products follow the arrival (the same products in the same order: the same bits).
        int split = 0;
        {
            const int32_t *addr = nullptr;
            int nd = VER_FINAL;
            if (tid == 0) { addr = vC; nd = need_c; }
            else if (tid <= nkb) addr = ver + bufL * nb2 + li * g.nbk + kb0 + tid - 1;
            else if (tid <= 2 * nkb) addr = ver + bufR * nb2 + ri * g.nbk + kb0 + tid - 1 - nkb;
            if (nkb >= 2 && g.split)
                split = !wg_wait_flags_opt(addr, nd, tid == nkb || tid == 2 * nkb, 1000 + slot % 1000, &g.sync->status, tid, &bc[2]);
            else
                wg_wait_flags<true>(addr, nd, 1000 + slot % 1000, &g.sync->status, tid);
        }
        if (st) st[1] = wall_clock64();
        if (agg) a1 = wall_clock64();
        if (tl) tl[1] = wall_clock64();
        Tile64 acc;
        const double *Lp0 = g.buf[bufL] + mo + ((int64_t)li * 64) * ld + (int64_t)kb0 * 64;
        const double *Rp0 = g.buf[bufR] + mo + ((int64_t)ri * 64) * ld + (int64_t)kb0 * 64;
        tile_update<true, true>(acc.v, first ? nullptr : C, Lp0, Rp0, ld, split ? nkb - 1 : nkb, plus ? 1.0 : -1.0,
                          mask_last ? g.nbk - 1 - kb0 : -1, ncol_last, As, Bs, tid, wave, lane);
        if (split) {
            const int kl = kb0 + nkb - 1;
            const int32_t *addr = tid == 0 ? ver + bufL * nb2 + li * g.nbk + kl : (tid == 1 ? ver + bufR * nb2 + ri * g.nbk + kl : nullptr);
            wg_wait_flags<true>(addr, VER_FINAL, 6000 + slot % 1000, &g.sync->status, tid);
            if (tl) tl[7] = wall_clock64();
            tile_update<true, true, true>(acc.v, nullptr, Lp0 + (int64_t)(nkb - 1) * 64, Rp0 + (int64_t)(nkb - 1) * 64, ld, 1, plus ? 1.0 : -1.0,
                                          mask_last ? g.nbk - 1 - kl : -1, ncol_last, As, Bs, tid, wave, lane);
        }
        if (st) st[2] = wall_clock64();
        if (agg) a2 = wall_clock64();
        if (tl) tl[2] = wall_clock64();
        if (post == T_STORE) {
            store_acc_sc1(acc.v, C, ld, crow, ccol);
            pull_next();   // (behind the stores: the ticket's round trip delayed wave 0's stores, and with them the publication, by 1-2 us)
            if (st) st[3] = wall_clock64();
            if (tl) tl[4] = wall_clock64();
            finish(vC, newver);
            continue;
        }
        // T_SOLVE / T_LOOK / T_LOOKD: wait for W_k, then tile <- tile * W_k^T
        // debug: the look-ahead tasks of every block of matrix 0 stamp their phases into trace[7000 + 8 k ..]
        long long *lk = ((post == T_LOOK || post == T_LOOKD) && g.trace && b == 0 && tid == 0 && wk < 120) ? g.trace + 7000 + 8 * wk : nullptr;
        if (lk && post == T_LOOK) { lk[0] = a0; lk[1] = wall_clock64(); }
        int32_t *vX = ver + 3 * nb2 + ci * g.nbk + cj;   // auxiliary word of the tile: T_LOOK has read it (T_LOOKD may overwrite it)
        if (post == T_LOOK) {   // (acc holds everything this task reads of the tile)
            __syncthreads();
            if (tid == 0) __hip_atomic_store(vX, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wg_wait_flags<true>(tid == 0 ? wflag : nullptr, wk + 1, 3000 + wk, &g.sync->status, tid);
        if (lk && post == T_LOOK) lk[2] = wall_clock64();
        if (tl) tl[3] = wall_clock64();
        d4 out[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t] = (d4){0.0, 0.0, 0.0, 0.0};
        mul_acc_bt<true>(out, acc.v, Wk, 64, As, Bs, tid, wave, lane);
        if (post == T_SOLVE) {
            store_acc_sc1(out, C, ld, crow, ccol);
            pull_next();
            if (st) st[3] = wall_clock64();
            if (tl) tl[4] = wall_clock64();
            finish(vC, newver);
            continue;
        }
        // ---- T_LOOK / T_LOOKD: the chain's look-ahead.  S = A[k+2][k] is final now and stays in registers; panel k is applied
        // with it to one of the two tiles the chain picks up after its next factorisation: T_LOOK  Q = A[k+2][k+1] -= S P^T
        // (P = A[k+1][k], the chain's own panel tile; this task also stores S), T_LOOKD  D = A[k+2][k+2] -= S S^T.  As three
        // tasks (solve, two updates) the chain's inputs were two worker hand-offs behind W_k (store, drain, publish, poll,
        // acquire, reload of S: ~15 us against a 15-us block step); the same products in the same order, so the same bits.
        {
            int32_t *vA = ver + BUF_A * nb2;
            double *Ab = g.buf[BUF_A] + mo;
            const int need2 = mt.b.w;   // earlier visits of the tile this task updates
            if (post == T_LOOK) {
                int32_t *vQ = vA + ci * g.nbk + cj + 1;
                // lanes 0 / 1: the chain's panel tile A[k+1][k] final; the earlier visits of Q (the catch-up tasks of the block
                // before: two hand-offs behind W_k-1, so NOT waited for ahead of the solve)
                const int32_t *addr = tid == 0 ? vA + (ci - 1) * g.nbk + cj : (tid == 1 ? vQ : nullptr);
                const int nd = tid == 0 ? VER_FINAL : need2;
                if (lk) lk[3] = wall_clock64();
                wg_wait_flags<true>(addr, nd, 4000 + wk, &g.sync->status, tid);
                if (lk) lk[4] = wall_clock64();
                if (tl) tl[7] = wall_clock64();
                const double *Pg = Ab + ((int64_t)(ci - 1) * 64) * ld + (int64_t)cj * 64;
                double *Qg = Ab + ((int64_t)ci * 64) * ld + (int64_t)(cj + 1) * 64;
                const HalfTile p0 = fetch_mk_sc1(Pg, ld, tid, 0), p1 = fetch_mk_sc1(Pg, ld, tid, 1);
                Tile64 qt;
                load_acc_sc1(qt.v, Qg, ld, crow, ccol);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) As[(crow + 4 * r) * LDM + 16 * t + ccol] = out[2 * h + t][r];
                    commit_mk(h == 0 ? p0 : p1, Bs, tid);
                    __syncthreads();
                    mfma_tile_ahead<OP_MK, OP_MK>(As, Bs, qt.v, wave, lane, -1.0);
                }
                store_acc_sc1(qt.v, Qg, ld, crow, ccol);
                pull_next();
                if (st) st[3] = wall_clock64();
                if (tl) tl[4] = wall_clock64();
                finish(vQ, need2 + 1);
                if (lk) lk[5] = wall_clock64();
            } else {
                int32_t *vD = vA + ci * g.nbk + ci;
                // lane 0: the earlier visits of D; lane 1: T_LOOK has read the tile that S replaces
                wg_wait_flags<true>(tid == 0 ? vD : (tid == 1 ? vX : nullptr), tid == 0 ? need2 : 1, 5000 + wk, &g.sync->status, tid);
                if (tl) tl[7] = wall_clock64();
                store_acc_sc1(out, C, ld, crow, ccol);   // the solved tile, final
                double *Dg = Ab + ((int64_t)ci * 64) * ld + (int64_t)ci * 64;
                Tile64 d;
                load_acc_sc1(d.v, Dg, ld, crow, ccol);
                // ... published as soon as it has drained (the wait covers D's loads as well): the next block's catch-up tasks need it
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(vC, newver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lk) lk[7] = wall_clock64();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) As[(crow + 4 * r) * LDM + 16 * t + ccol] = out[2 * h + t][r];
                    __syncthreads();
                    mfma_tile_ahead<OP_MK, OP_MK>(As, As, d.v, wave, lane, -1.0);
                }
                store_acc_sc1(d.v, Dg, ld, crow, ccol);
                pull_next();
                if (st) st[3] = wall_clock64();
                if (tl) tl[4] = wall_clock64();
                finish(vD, need2 + 1);
                if (lk) lk[6] = wall_clock64();
            }
        }
    }
}

// column n of T holds -K^-1 y: copy it into row n of S (the layout dgpamd_potri leaves: row n of Ainv = -alpha^T)
__global__ void copy_alpha_kernel(const double *T, double *S, int64_t ld, int64_t n, int64_t stride_a) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    T += (int64_t)blockIdx.y * stride_a;
    S += (int64_t)blockIdx.y * stride_a;
    S[n * ld + j] = T[j * ld + n];
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

static size_t mega_sync_bytes(int64_t nbk, int batch);
// offset of the one-launch kernel's synchronisation block inside the workspace (on a 128-byte line: its hot words have lines of their own)
static size_t mega_sync_offset(int64_t n, int batch) {
    const size_t b = potrf_ws_doubles(n, batch) * sizeof(double) + 3 * DGPAMD_MAXB * sizeof(int32_t);   // + info, step flags, chain CU keys
    return (b + 127) / 128 * 128;
}
// offset of the one-launch kernel's pivot array (batch x Np doubles) behind the synchronisation block
static size_t mega_piv_offset(int64_t n, int batch) {
    return mega_sync_offset(n, batch) + mega_sync_bytes(padded_dim(n) / 64, batch);
}
extern "C" size_t dgpamd_potrf_workspace(int64_t n, int batch) {
    return mega_piv_offset(n, batch) + (size_t)batch * padded_dim(n) * sizeof(double);
}

// Task tables (see potrf_step_kernel): one vector of tasks per launch, cached on the device per (nbk, inverse).
struct TaskTable {
    int4 *dev = nullptr;
    std::vector<int> offset, count, nhead, npanel;
    std::vector<double> tile_ops;   // 64^3 multiply-add units per launch and matrix (for the profiler)
};

#ifndef PANEL_LEAD
#define PANEL_LEAD 1024   // bulk workgroups dispatched before the (waiting) panel workgroups of a launch
#endif
#ifndef LAZY
#define LAZY 4   // panels (64-pivot blocks) applied per visit of a bulk tile
#endif
static void build_tasks(int nbk, bool inv, std::vector<std::vector<int4>> &L, std::vector<double> &ops) {
    const int nl = nbk + (inv ? 2 : 0);
    L.assign(nl, {});
    ops.assign(nl, 0.0);
    for (int k = 0; k < nl; ++k) {
        std::vector<int4> &t = L[k];
        double &w = ops[k];
        if (k < nbk) {
            const int np1 = k >= 1 ? 1 : 0;   // column k takes panel k-1 only
            t.push_back(make_task(T_CHAIN, 0, 0, 0, BUF_A, k, k, BUF_A, k, BUF_A, k, k - np1, np1));
            w += np1 + 1;
            if (inv) t.push_back(make_task(T_TDIAG, 1, 0, 0, BUF_T, k, k, BUF_A, 0, BUF_A, 0, 0, 0));
            for (int i = k + 1; i < nbk; ++i) {
                t.push_back(make_task(T_SOLVE, 0, 0, 0, BUF_A, i, k, BUF_A, i, BUF_A, k, k - np1, np1));
                w += np1 + 1;
            }
            if (inv)
                for (int q = k - 1; q >= 0; --q) {   // T[q][k]: first touched by panel k-1 when q == k-1
                    t.push_back(make_task(T_SOLVE, q == k - 1, 0, 0, BUF_T, q, k, BUF_T, q, BUF_A, k, k - 1, 1));
                    w += 2;
                }
            // lazy bulk of A: columns k+1, k+1+LAZY, ... take the LAZY panels (k-LAZY .. k-1)
            const int kb0 = k >= LAZY ? k - LAZY : 0, nkb = k - kb0;
            if (nkb > 0)
                for (int j = k + 1; j < nbk; j += LAZY)
                    for (int i = j; i < nbk; ++i) {
                        t.push_back(make_task(T_STORE, 0, 0, 0, BUF_A, i, j, BUF_A, i, BUF_A, j, kb0, nkb));
                        w += nkb;
                    }
            if (inv && nkb > 0)   // same rule for the rows of T that already have a panel: q <= k-1
                for (int j = k + 1; j < nbk; j += LAZY)
                    for (int q = 0; q <= k - 1; ++q) {
                        const int f0 = q > kb0 ? q : kb0;   // panels >= q only; the first visit starts from zero
                        t.push_back(make_task(T_STORE, f0 == q, 0, 0, BUF_T, q, j, BUF_T, q, BUF_A, j, f0, k - f0));
                        w += k - f0;
                    }
        }
        if (inv)   // S[q][q'] += Pt_q Pt_q'^T for the panels (k-2, k-1), rows q = k-2, k-4, ...
            for (int q = k - 2; q >= 0; q -= 2) {
                const int nkb = (k - 1 < nbk ? k : nbk) - (k - 2);   // clip to panels < nbk
                if (nkb <= 0) continue;
                const int mask = (k - 2 + nkb - 1 == nbk - 1) ? 1 : 0;   // the last block carries right-hand sides
                for (int q2 = 0; q2 <= q; ++q2) {
                    t.push_back(make_task(T_STORE, q == k - 2, 1, mask, BUF_S, q, q2, BUF_T, q, BUF_T, q2, k - 2, nkb));
                    w += nkb;
                }
            }
    }
}

static int get_tasks(dgpamd_ctx *ctx, int nbk, bool inv, TaskTable *&out) {
    static std::map<std::pair<dgpamd_ctx *, std::pair<int, int>>, TaskTable> cache;   // contexts are few and long-lived
    TaskTable &tt = cache[{ctx, {nbk, inv ? 1 : 0}}];
    if (!tt.dev) {
        std::vector<std::vector<int4>> L;
        build_tasks(nbk, inv, L, tt.tile_ops);
        std::vector<int4> flat;
        for (auto &v : L) {
            tt.offset.push_back((int)flat.size());
            tt.count.push_back((int)v.size());
            int nh = 0, np_ = 0;
            for (auto &t4 : v) {
                const int post = t4.x & 15;
                nh += (post == T_CHAIN || post == T_TDIAG);
                np_ += (post == T_SOLVE);
            }
            tt.nhead.push_back(nh);
            tt.npanel.push_back(np_);
            flat.insert(flat.end(), v.begin(), v.end());
        }
        HIP_TRY(ctx, hipMalloc((void **)&tt.dev, flat.size() * sizeof(int4)));
        HIP_TRY(ctx, hipMemcpy(tt.dev, flat.data(), flat.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    out = &tt;
    return DGPAMD_OK;
}

// ---- task table of the one-launch kernel -------------------------------------------------------------------------
// The same right-looking schedule as build_tasks (lazy bulk: a tile takes LAZY panels per visit), restated per tile:
// `applied` panels so far, `visits` so far.  Differences: no chain tasks (the chain workgroups run on their own); the
// first panel tile A[k+1][k] gets its last worker visit as a plain update (the chain does the solve); what the chain
// waits for is produced one block step ahead of the rest (look-ahead, see below).
// Panels per visit of a trailing A / T tile and of a K^-1 tile, and where the cadence starts (see build_mega_tasks).  The
// per-step kernel's 4 / 2 / next column against 8 / 8 / three columns on gives potrf -12 % at 6 matrices and -13 % at 12,
// potrf_inv -19 % at 6: deeper visits run the tile engine at a higher rate, and a column whose last bulk visit lies three
// block steps before its solves does not have those solves queue up behind a 40-us task.
#define MEGA_LAZY 8
#define MEGA_SLAZY 8
#define MEGA_NEAR 2
struct MegaTable {
    MTask *dev = nullptr;        // urgent tasks first (nut of them), then the bulk tasks
    int2 *need_dev = nullptr;
    int ntask = 0, nut = 0;
};

// tag[i]: (block step << 1) | critical -- critical: the lane that hands the chain its next inputs (the look-ahead pair, the first solve
// of the panel column and the catch-up updates behind it); everything else is bulk.
static void build_mega_tasks(int nbk, bool inv, std::vector<MTask> &out, std::vector<int2> &need, int lazy, int slazy,
                             int near, int lag, int xcatch, int stail, int look, std::vector<int> &tag, int slag = 0) {
    int cur_step = 0, cur_urgent = 0;
    const int nb2 = nbk * nbk;
    std::vector<int> appliedA(nb2, 0), visitsA(nb2, 0), appliedT(nb2, 0), visitsT(nb2, 0), visitsS(nb2, 0), appliedS(nbk, 0);
    for (int q = 0; q < nbk; ++q) appliedS[q] = q;   // row q of K^-1 sums the panels kb >= q
    if (inv)
        for (int q = 0; q < nbk; ++q)
            for (int j = 0; j < nbk; ++j) appliedT[q * nbk + j] = q;   // T[q][j] sums the panels kb >= q
    need.assign(nbk, make_int2(0, 0));
    auto emit = [&](int4 a, int need_c, int fin, int wk) {
        MTask t;
        t.a = a;
        t.b = make_int4(need_c, fin, wk, 0);
        out.push_back(t);
        tag.push_back((cur_step << 1) | (cur_urgent ? 1 : 0));
    };
    // bring A[i][j] up to the panels < upto (plain update, stored)
    auto updA = [&](int i, int j, int upto) {
        int &ap = appliedA[i * nbk + j];
        if (upto <= ap) return;
        emit(make_task(T_STORE, 0, 0, 0, BUF_A, i, j, BUF_A, i, BUF_A, j, ap, upto - ap), visitsA[i * nbk + j]++, 0, 0);
        ap = upto;
    };
    // bring T[q][j] up to the panels < upto
    auto updT = [&](int q, int j, int upto) {
        int &ap = appliedT[q * nbk + j];
        if (upto <= ap) return;
        emit(make_task(T_STORE, ap == q, 0, 0, BUF_T, q, j, BUF_T, q, BUF_A, j, ap, upto - ap), visitsT[q * nbk + j]++, 0, 0);
        ap = upto;
    };
    auto solveA = [&](int i, int k) {   // panel tile (i, k): the remaining updates, then the solve with W_k
        const int ap = appliedA[i * nbk + k];
        emit(make_task(T_SOLVE, 0, 0, 0, BUF_A, i, k, BUF_A, i, BUF_A, k, ap, k - ap), visitsA[i * nbk + k]++, 1, k);
        appliedA[i * nbk + k] = k;
    };
    // The look-ahead of block k as TWO tasks that run side by side, each with the solve of A[k+2][k] in it: T_LOOK applies
    // panel k to A[k+2][k+1], T_LOOKD stores the solved tile and applies panel k to A[k+2][k+2] (each tile lacks exactly that
    // panel: the catch-up tasks of block k-1 have brought them up to the panels < k).  One task for both left the diagonal
    // tile 4 us behind the panel tile, which the chain then waited for; and the solved tile is published by the task that has
    // time for it (the next block's catch-up tasks wait for it: published behind the panel-tile update it was 7 us late).
    auto lookA = [&](int k) -> bool {
        const int i = k + 2;
        if (!look || appliedA[i * nbk + k + 1] != k || appliedA[i * nbk + i] != k) return false;
        const int ap = appliedA[i * nbk + k];
        // T_LOOK first (it waits for nothing of T_LOOKD's; T_LOOKD waits for T_LOOK's "input read" word before it stores the
        // solved tile over the input): the table stays a topological order
        MTask t;
        t.a = make_task(T_LOOK, 0, 0, 0, BUF_A, i, k, BUF_A, i, BUF_A, k, ap, k - ap);
        t.b = make_int4(visitsA[i * nbk + k], 0, k, visitsA[i * nbk + k + 1]);
        out.push_back(t);
        tag.push_back((cur_step << 1) | 1);
        MTask d;
        d.a = make_task(T_LOOKD, 0, 0, 0, BUF_A, i, k, BUF_A, i, BUF_A, k, ap, k - ap);
        d.b = make_int4(visitsA[i * nbk + k]++, 1, k, visitsA[i * nbk + i]);
        out.push_back(d);
        tag.push_back((cur_step << 1) | 1);
        ++visitsA[i * nbk + k + 1];
        ++visitsA[i * nbk + i];
        appliedA[i * nbk + k] = k;
        appliedA[i * nbk + k + 1] = appliedA[i * nbk + i] = k + 1;
        return true;
    };
    const int nl = nbk + (inv ? 1 + lag + slag : 0);   // (one more pass flushes what is left of K^-1)
    for (int k = 0; k < nl; ++k) {
        cur_step = k;
        if (k < nbk) {
            // LOOK-AHEAD: what the chain picks up after factoring block k+1 -- A[k+2][k+1] and A[k+2][k+2] with the panels
            // <= k -- needs only W_k and the chain's own panel tile A[k+1][k]: it comes first in the tasks of block k,
            // ahead of their bulk, so the chain's serial part (solve, update, factor: ~15 us) runs beside that bulk
            // instead of after it.
            if (k + 2 < nbk) {
                cur_urgent = 1;
                if (!lookA(k)) {
                    solveA(k + 2, k);
                    updA(k + 2, k + 2, k + 1);
                    updA(k + 2, k + 1, k + 1);
                }
                cur_urgent = 0;
                need[k + 1] = make_int2(visitsA[(k + 2) * nbk + k + 1], visitsA[(k + 2) * nbk + k + 2]);
            }
            if (inv) emit(make_task(T_TDIAG, 1, 0, 0, BUF_T, k, k, BUF_A, 0, BUF_A, 0, 0, 0), visitsT[k * nbk + k]++, 1, k);
            // panel column k; behind its first tile the catch-up of the NEXT look-ahead's two tiles (panels <= k), so that
            // the visits the chain waits for apply one panel each
            for (int i = k + 3; i < nbk; ++i) {
                cur_urgent = (i == k + 3);
                solveA(i, k);
                cur_urgent = 0;
                if (i == k + 3) {
                    cur_urgent = 1;
                    updA(k + 3, k + 3, k + 1);
                    updA(k + 3, k + 2, k + 1);
                    if (xcatch) updA(k + 3, k + 1, k + 1);   // (the next look-ahead solve then has no panel left to apply)
                    // The next block's look-ahead tile A[k+3][k+1] up to the panels < k: all final by now.  The look-ahead tasks
                    // then apply ONE panel (k, whose operand the solve of A[k+3][k] publishes ~5 us before W_k+1) where they
                    // applied the three the lazy cadence leaves -- 5 us of arithmetic between the arrival of the last operand
                    // (the previous block's solved tile) and the solve, on the path to the tile the chain waits for.
                    else if (look) updA(k + 3, k + 1, k);
                    cur_urgent = 0;
                }
            }
            if (inv)
                for (int q = k - 1; q >= 0; --q) {
                    const int ap = appliedT[q * nbk + k];
                    emit(make_task(T_SOLVE, ap == q, 0, 0, BUF_T, q, k, BUF_T, q, BUF_A, k, ap, k - ap), visitsT[q * nbk + k]++, 1, k);
                    appliedT[q * nbk + k] = k;
                }
            // lazy bulk: columns k+1, k+1+LAZY, ... below the diagonal; diagonal tiles one block step ahead of that
            // (near: the cadence's first column is k+1+near -- a column's last bulk visit is then 1+near block steps before its
            //  solves, which do not queue up behind it; lag: the bulk applies the panels < k-lag only, i.e. nothing that the
            //  solves of the previous `lag` block steps are still producing.  The solves apply what is left: 1+near+lag panels.)
            const int kl = k - lag;
            if (kl > 0) {
                for (int j = k + 2 + near; j < nbk; j += lazy) updA(j, j, kl);
                for (int j = k + 1 + near; j < nbk; j += lazy)
                    for (int i = j + 1; i < nbk; ++i) updA(i, j, kl);
                if (inv)
                    for (int j = k + 1 + near; j < nbk; j += lazy)
                        for (int q = 0; q <= kl - 1; ++q) updT(q, j, kl);
            }
        }
        if (inv) {
            // S[q][q'] += Pt_q Pt_q'^T: row q is visited when `slazy` panels are pending (rows q = k - slazy, k - 2 slazy, ... at
            // block step k).  With `stail` (up to three matrices: the chains bound the time) the threshold shrinks towards the end
            // with the block steps that are left, so that when the last chain has finished every K^-1 tile lacks one or two
            // panels, not eight: the tail after the chains is one round of short tasks behind the last column's solves
            // (potrf_inv of one matrix 0.57 -> 0.55 ms, of three 0.77 -> 0.75; with more matrices the extra shallow tasks cost more than the tail).
            const int ks = k - lag - slag;   // (slag: the K^-1 visits trail by that many block steps -- nothing waits for them)
            const int left = nbk - ks;   // block steps until the flush
            const int thr = ks >= nbk ? 1 : (stail && left < slazy ? (left > 1 ? left : 1) : slazy);
            for (int q = 0; q <= ks - 1 && q < nbk; ++q) {
                const int upto = ks < nbk ? ks : nbk;
                if (upto - appliedS[q] < thr) continue;
                const int ap = appliedS[q], nkb = upto - ap;
                const int mask = (upto - 1 == nbk - 1) ? 1 : 0;
                for (int q2 = 0; q2 <= q; ++q2)
                    emit(make_task(T_STORE, ap == q, 1, mask, BUF_S, q, q2, BUF_T, q, BUF_T, q2, ap, nkb), visitsS[q * nbk + q2]++, 0, 0);
                appliedS[q] = upto;
            }
        }
    }
}

static int mega_wgs_per_cu();
static int get_mega_tasks(dgpamd_ctx *ctx, int nbk, bool inv, int batch, MegaTable *&out) {
    // Panels per visit: 8 (trailing and K^-1 tiles), 10 from six matrices on; the cadence starts three columns to the right
    // of the chain (near = 2).  Measured at n = 2000 (tools/gpu_lazy_sweep.py, profiles/r02_mega_table_sweeps.txt).
    // DGPAMD_MEGA_LAZY / _SLAZY / _NEAR / _LAG / _XCATCH / _STAIL override (tuning only).
    const char *lz = getenv("DGPAMD_MEGA_LAZY"), *sz = getenv("DGPAMD_MEGA_SLAZY");
    const bool deep = batch >= 6;
    // (n = 5000, ten matrices with their inverses: 25.9 ms at 8 panels per visit, 25.3 at 12, 25.4 at 16 -- 49 TFLOP/s; one matrix is
    //  fastest at 8: the chain waits for deeper visits)
    // Round 4, with the 12.5-us chain (profiles/r04_mega_table_sweeps.txt, n = 2000): with the inverse, 10 panels per visit also
    // below six matrices (potrf_inv 0.610 -> 0.587 ms at two, 0.738 -> 0.714 at three, 0.813 -> 0.797 at four); from six on 12 panels
    // and the cadence starting one column further right (near = 3: 1.142 -> 1.124 ms at six, 1.977 -> 1.847 at ten).  Without
    // the inverse nothing moved (near = 3 costs one matrix 0.06 ms).
    const int deep_lazy = (nbk >= 64 || inv) ? 12 : 10;
    const int shallow = inv ? 10 : MEGA_LAZY, sshallow = inv ? 10 : MEGA_SLAZY;
    const int lazy = lz && atoi(lz) > 0 ? atoi(lz) : (deep ? deep_lazy : shallow), slazy = sz && atoi(sz) > 0 ? atoi(sz) : (deep ? deep_lazy : sshallow);
    const char *en = getenv("DGPAMD_MEGA_NEAR"), *el = getenv("DGPAMD_MEGA_LAG"), *ex = getenv("DGPAMD_MEGA_XCATCH");
    const char *es = getenv("DGPAMD_MEGA_STAIL");
    // (a task waits on 1 + 2 nkb version words, one per lane of ONE wave: the overrides are clamped so that the deepest visit --
    //  lazy + 1 + near + lag panels -- stays within 64 flags, and build_mega_tasks' output is checked below)
    // Round 5 (profiles/r05_mega_table_sweeps.txt): with the newest panel of a visit applied on its own (kernel, `split`) the solves can
    // leave more panels to themselves without lengthening the recursion along a row of tiles, and a column's last deep visit
    // then lies four block steps ahead of its solves: three columns' distance (near = 3) at every batch size.
    const int near_raw = en ? atoi(en) : 3, lag_raw = el ? atoi(el) : 0;
    const int near = near_raw < 0 ? 0 : (near_raw > 6 ? 6 : near_raw), lag = lag_raw < 0 ? 0 : (lag_raw > 2 ? 2 : lag_raw), xcatch = (ex ? atoi(ex) : 0) + 2 * (es ? atoi(es) : (inv && batch <= 3 ? 1 : 0));
    const char *elk = getenv("DGPAMD_MEGA_LOOK");   // 0: the look-ahead as three tasks (rounds 2-3)
    const int look = elk ? (atoi(elk) != 0) : 1;
    const char *esl = getenv("DGPAMD_MEGA_SLAG");
    const int slag = esl ? (atoi(esl) < 0 ? 0 : (atoi(esl) > 8 ? 8 : atoi(esl))) : 0;
    // Two queues (critical lane + bulk) up to three matrices, one from four on.  (Re-measured after the synchronisation block's words got lines of their own,
    // profiles/r05_mega_table_sweeps.txt: most of what the critical queue had gained at four and more matrices was that ITS head did not share the line of the W
    // counters -- with every word on its own line one queue is 1-5 % faster there, the two still 1.5-3 % at one to three.)
    const char *eq = getenv("DGPAMD_MEGA_QUEUES");   // 1: one queue per group of matrices (rounds 2-4); 2: critical + bulk
    int queues = eq ? (atoi(eq) == 1 ? 1 : 2) : (batch <= 3 ? 2 : 1);
    {
        // The two-queue form needs workers of BOTH kinds: the kernel makes the first 8 x crit_per_xcd workers critical-first (crit_per_xcd =
        // ceil(ncrit x matrices / 8), at most 16) and everybody else bulk-first.  On a small or CU-masked device every worker would be critical-first,
        // run ahead into later block steps and wait there for bulk tiles that nobody pulls -- until the spin limit trips.  One queue then.
        const int ncrit_env = getenv("DGPAMD_MEGA_NCRIT") ? atoi(getenv("DGPAMD_MEGA_NCRIT")) : 0;
        const int ncrit = ncrit_env > 0 ? ncrit_env : (batch <= 4 ? 12 : 8);
        int crit = (ncrit * batch + 7) / 8;
        if (crit > 16) crit = 16;
        const int64_t workers = (int64_t)mega_wgs_per_cu() * ctx->num_cu - batch;
        if (workers < 2 * 8 * (int64_t)crit) queues = 1;
    }
    static std::map<std::pair<dgpamd_ctx *, std::array<int, 10>>, MegaTable> cache;
    MegaTable &mt = cache[{ctx, {nbk, inv ? 1 : 0, lazy, slazy, near, lag, xcatch, look, queues, slag}}];
    if (!mt.dev) {
        std::vector<MTask> all, tasks;
        std::vector<int2> need;
        std::vector<int> tag;
        build_mega_tasks(nbk, inv, all, need, lazy > 24 ? 24 : lazy, slazy > 24 ? 24 : slazy, near, lag, xcatch & 1, xcatch >> 1, look, tag, slag);
        for (const MTask &t : all)   // wg_wait_flags polls with the 64 lanes of one wave
            if (1 + 2 * (t.a.w >> 16) > 64) BAD_ARG(ctx, "task table: a visit applies more panels than one wave can wait for");
        // urgent tasks first (their block step rides in b.y above the `final` bit), then the bulk tasks: two queues (see the kernel)
        if (queues == 2)
            for (size_t i = 0; i < all.size(); ++i)
                if (tag[i] & 1) {
                    MTask t = all[i];
                    t.b.y |= (tag[i] >> 1) << 8;
                    tasks.push_back(t);
                }
        mt.nut = (int)tasks.size();
        for (size_t i = 0; i < all.size(); ++i)
            if (queues != 2 || !(tag[i] & 1)) tasks.push_back(all[i]);
        mt.ntask = (int)tasks.size();
        HIP_TRY(ctx, hipMalloc((void **)&mt.dev, (tasks.size() + 1) * sizeof(MTask)));
        HIP_TRY(ctx, hipMemcpy(mt.dev, tasks.data(), tasks.size() * sizeof(MTask), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void **)&mt.need_dev, need.size() * sizeof(int2)));
        HIP_TRY(ctx, hipMemcpy(mt.need_dev, need.data(), need.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    out = &mt;
    return DGPAMD_OK;
}

extern "C" int dgpamd_debug_mega_table(dgpamd_ctx *ctx, int64_t n, int inv, int batch, int32_t *host_out, int64_t cap_words) {
    if (!ctx || !host_out || n <= 0 || batch <= 0) return -DGPAMD_BAD_ARG;
    const int nbk = (int)(padded_dim(n) / 64);
    MegaTable *mt = nullptr;
    if (get_mega_tasks(ctx, nbk, inv != 0, batch, mt)) return -DGPAMD_BAD_ARG;
    if ((int64_t)mt->ntask * 8 + 2 * nbk > cap_words) return -DGPAMD_BAD_ARG;
    if (hipMemcpy(host_out, mt->dev, (size_t)mt->ntask * sizeof(MTask), hipMemcpyDeviceToHost) != hipSuccess) return -DGPAMD_HIP_ERROR;
    if (hipMemcpy(host_out + (int64_t)mt->ntask * 8, mt->need_dev, (size_t)nbk * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess) return -DGPAMD_HIP_ERROR;
    return mt->ntask;
}

static int mega_wgs_per_cu() {
    static int n = 0;
    if (!n) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, potrf_mega_kernel, 256, 0) != hipSuccess || q < 1) q = 2;
        n = q > 4 ? 4 : q;
    }
    return n;
}

// bytes of the one-launch kernel's synchronisation block (MegaSync + tile versions), a multiple of 16
static size_t mega_sync_bytes(int64_t nbk, int batch) {
    size_t b = sizeof(MegaSync) + (size_t)batch * VER_PLANES * nbk * nbk * sizeof(int32_t);
    return (b + 15) / 16 * 16;
}

static int potrf_mega_launch(dgpamd_ctx *ctx, int64_t n, double *A, double *T, double *S, int64_t stride_a, int batch,
                             double *logdet, int32_t *info, double *ws, void *syncmem, double *piv, const MegaTable *mt,
                             bool no_post = false, bool sync_cleared = false) {
    const int64_t Np = padded_dim(n);
    const int nbk = (int)(Np / 64);
    if (!sync_cleared) HIP_TRY(ctx, hipMemsetAsync(syncmem, 0, mega_sync_bytes(nbk, batch), ctx->stream));
    MegaArgs g;
    g.buf[BUF_A] = A; g.buf[BUF_T] = T; g.buf[BUF_S] = S;
    g.ws = ws; g.ld = Np; g.stride_a = stride_a; g.stride_ws = (int64_t)nbk * 4096; g.n = n; g.nbk = nbk; g.batch = batch;
    g.inv = T != nullptr;
    g.tasks = mt->dev; g.ntask = mt->ntask; g.chain_need = mt->need_dev;
    g.nut = mt->nut;
    {
        // (workers of the critical queue per matrix: one per task of a block step, and a few more while the engine has room)
        static const int ncrit_env = getenv("DGPAMD_MEGA_NCRIT") ? atoi(getenv("DGPAMD_MEGA_NCRIT")) : 0;
        g.ncrit = ncrit_env > 0 ? ncrit_env : (batch <= 4 ? 12 : 8);
    }
    g.sync = reinterpret_cast<MegaSync *>(syncmem);
    g.ver = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(syncmem) + sizeof(MegaSync));
    g.logdet = logdet; g.info = info; g.trace = ctx->trace; g.pred = ctx->pred; g.piv = piv;
    g.tlog = (ctx->tlog && 64 + 8 * ((int64_t)batch * nbk + (int64_t)batch * mt->ntask) <= ctx->tlog_words) ? ctx->tlog : nullptr;
    {
        static const int nowait = getenv("DGPAMD_MEGA_NOWAIT") ? atoi(getenv("DGPAMD_MEGA_NOWAIT")) : 0;
        g.nowait = nowait;
        static const int split = getenv("DGPAMD_MEGA_SPLIT") ? atoi(getenv("DGPAMD_MEGA_SPLIT")) : 1;
        g.split = split;
    }
    {
        const char *eg = getenv("DGPAMD_MEGA_GROUPS");
        // (measured at n = 2000, tools/gpu_lazy_sweep.py: uneven groups are fine -- an XCD whose own queue has run out takes
        // from the others -- so odd batches use two groups as well: potrf_inv -10 % at 3 matrices, -7 % at 6, -12 % at 12)
        // (round 5, same sweeps: three matrices with the inverse, and odd batches without it, are 3-5 % faster as ONE group -- two groups of 2 + 1 or 3 + 2
        //  matrices leave half the XCDs with the smaller share until their queue runs out)
        const bool one_group = (batch & 1) && (batch == 3 || !g.inv);
        int G = eg ? atoi(eg) : (one_group ? 1 : batch % 8 == 0 ? 8 : batch % 4 == 0 ? 4 : batch >= 2 ? 2 : 1);
        if (G != 1 && G != 2 && G != 4 && G != 8) G = 1;
        while (G > batch) G >>= 1;
        g.ngroups = G;
    }
    // every workgroup resident at once (not needed for progress, but a queued worker would only start late)
    int64_t grid = (int64_t)mega_wgs_per_cu() * ctx->num_cu;
    const int64_t useful = (int64_t)batch * (mt->ntask + 1 + mega_wgs_per_cu());
    if (grid > useful) grid = useful;
    if (grid < batch + 1) grid = batch + 1;
    // algorithmic flops of the launch: n^3/3 per matrix, n^3 with the fused inverse
    PROF_BEGIN(ctx, PROF_SYRK, (double)batch * (double)n * (double)n * (double)n * (T ? 1.0 : 1.0 / 3.0));
    hipLaunchKernelGGL(potrf_mega_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, g);
    PROF_END(ctx, PROF_SYRK);
    if (T && !no_post)
        hipLaunchKernelGGL(copy_alpha_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, ctx->stream,
                           (const double *)T, S, Np, n, stride_a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// workgroups of potrf_step_kernel that one CU holds (registers / LDS): decides how many placeholder rows guard a chain
static int step_kernel_wgs_per_cu() {
    static int n = 0;
    if (!n) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, potrf_step_kernel, 256, 0) != hipSuccess || q < 1) q = 2;
        n = q > 4 ? 4 : q;
    }
    return n;
}

static int potrf_launches(dgpamd_ctx *ctx, int64_t n, double *A, double *T, double *S, int64_t stride_a, int batch,
                          double *logdet, int32_t *info, double *ws, int32_t *flags, const TaskTable *tt) {
    const int64_t Np = padded_dim(n);
    const int nbk = (int)(Np / 64);
    const double tile_flops = 2.0 * 64.0 * 64.0 * 64.0;
    HIP_TRY(ctx, hipMemsetAsync(flags, 0, 2 * DGPAMD_MAXB * sizeof(int32_t), ctx->stream));
    StepArgs st;
    st.buf[BUF_A] = A; st.buf[BUF_T] = T; st.buf[BUF_S] = S;
    st.ws = ws; st.ld = Np; st.stride_a = stride_a; st.stride_ws = (int64_t)nbk * 4096; st.n = n; st.nbk = nbk;
    st.logdet = logdet; st.info = info; st.flags = flags; st.batch = batch; st.trace = ctx->trace;
    for (size_t k = 0; k < tt->count.size(); ++k) {
        if (tt->count[k] == 0) continue;
        st.k = (int)k;
        st.tasks = tt->dev + tt->offset[k];
        st.nhead = tt->nhead[k];
        st.npanel = tt->npanel[k];
        {
            const int nbulk = tt->count[k] - st.nhead - st.npanel, lead = (PANEL_LEAD + batch - 1) / batch;
            st.lead = lead < nbulk ? lead : nbulk;
        }
        const int64_t nwg = (int64_t)batch * tt->count[k];
        st.guard = (k < (size_t)nbk && nwg > ctx->num_cu) ? ctx->num_cu : 0;   // (tail launches: no chain; small ones: no neighbours)
        st.nph = step_kernel_wgs_per_cu() - 1;
        int64_t grid = nwg;
        if (st.guard)   // smallest grid that holds nwg tasks beside the placeholders of the rows it reaches
            for (int m = 1; m <= st.nph && grid > (int64_t)m * st.guard; ++m) grid += batch;
        PROF_BEGIN(ctx, PROF_SYRK, (double)batch * tt->tile_ops[k] * tile_flops);
        hipLaunchKernelGGL(potrf_step_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, st);
        PROF_END(ctx, PROF_SYRK);
    }
    if (T)
        hipLaunchKernelGGL(copy_alpha_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, ctx->stream,
                           (const double *)T, S, Np, n, stride_a);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// status: the one-launch kernel's give-up word (null for the per-step launches, whose spins write info = -1 themselves).
// A lost hand-off is reported as info = -1 for every matrix of the call: not a numerical failure (ops.py raises
// DgpAmdError for it, not LinAlgError).
__global__ void potrf_copy_out_kernel(const double *ld_ws, const int32_t *info_ws, double *logdet, int32_t *info, int batch,
                                      const int32_t *status) {
    const int b = threadIdx.x;
    if (b < batch) {
        logdet[b] = __hip_atomic_load(ld_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int st = status ? __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        info[b] = st ? -1 : __hip_atomic_load(info_ws + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static bool potrf_uses_mega(const dgpamd_ctx *ctx, int batch, bool inv) {
    // mode 1 (default): the one persistent launch; mode 0: one launch per block step (no in-kernel waits between workgroups: the
    // fallback when the device is shared -- HandoffError -- and the same-run reference of the tools).  The device queue's
    // predicated launches exist for the one-launch kernel only.  (Round 2's mode 2, chosen per call by batch size, had no advantage
    // at any batch size and is gone.)
    (void)batch; (void)inv;
    return ctx->potrf_mode == 1;
}

void potrf_sync_area(dgpamd_ctx *ctx, int64_t n, int batch, bool inv, double *ws, int32_t **ptr, int *words) {
    *ptr = nullptr;
    *words = 0;
    if (!potrf_uses_mega(ctx, batch, inv)) return;
    *ptr = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(ws) + mega_sync_offset(n, batch));
    *words = (int)(mega_sync_bytes(padded_dim(n) / 64, batch) / sizeof(int32_t));
}

int run_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet, int32_t *info,
              double *ws, double *T, double *S, PotrfPost *post, bool sync_cleared) {
    // One launch per 64-column block step with a static shape: replayed as one hipGraph.  The graph writes
    // logdet/info into the workspace tail (fixed addresses -> the cached graph does not depend on where the
    // caller wants them); a tiny kernel outside the graph copies them out.
    const int64_t nbk = padded_dim(n) / 64;
    double *ld_ws = ws + (size_t)batch * nbk * 4096 + DGPAMD_MAXB;
    int32_t *info_ws = reinterpret_cast<int32_t *>(ws + (size_t)batch * nbk * 4096 + 2 * DGPAMD_MAXB);
    int32_t *flags = info_ws + DGPAMD_MAXB;
    const bool mega = potrf_uses_mega(ctx, batch, T != nullptr);
    if (mega) {
        MegaTable *mt = nullptr;
        int rc = get_mega_tasks(ctx, (int)nbk, T != nullptr, batch, mt);   // (uploads the table on first use)
        if (rc) return rc;
        (void)mega_wgs_per_cu();
        void *syncmem = reinterpret_cast<char *>(ws) + mega_sync_offset(n, batch);
        double *piv = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + mega_piv_offset(n, batch));
        rc = potrf_mega_launch(ctx, n, A, T, S, stride_a, batch, ld_ws, info_ws, ws, syncmem, piv, mt, post != nullptr, sync_cleared);
        if (rc) return rc;
        if (post) {
            post->pending = 1;
            post->ld_ws = ld_ws;
            post->info_ws = info_ws;
            post->status = &reinterpret_cast<MegaSync *>(syncmem)->status;
            post->spare = &reinterpret_cast<MegaSync *>(syncmem)->pad[0];
            return DGPAMD_OK;
        }
        hipLaunchKernelGGL(potrf_copy_out_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)ld_ws,
                           (const int32_t *)info_ws, logdet, info, batch, (const int32_t *)&reinterpret_cast<MegaSync *>(syncmem)->status);
        LAUNCH_CHECK(ctx);
        return DGPAMD_OK;
    }
    TaskTable *tt = nullptr;
    int rc = get_tasks(ctx, (int)nbk, T != nullptr, tt);   // (uploads the table on first use: outside the capture)
    if (rc) return rc;
    (void)step_kernel_wgs_per_cu();   // (occupancy query: outside the capture as well)
    const std::array<uint64_t, 10> key = {1, (uint64_t)n, (uint64_t)batch, (uint64_t)A, (uint64_t)stride_a, (uint64_t)ws,
                                          (uint64_t)T, (uint64_t)S, 0, 0};
    rc = graph_run(ctx, key, [&]() { return potrf_launches(ctx, n, A, T, S, stride_a, batch, ld_ws, info_ws, ws, flags, tt); });
    if (rc) return rc;
    hipLaunchKernelGGL(potrf_copy_out_kernel, dim3(1), dim3(DGPAMD_MAXB), 0, ctx->stream, (const double *)ld_ws,
                       (const int32_t *)info_ws, logdet, info, batch, (const int32_t *)nullptr);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

extern "C" int dgpamd_set_potrf_mode(dgpamd_ctx *ctx, int mode) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (mode < 0 || mode > 1) BAD_ARG(ctx, "mode must be 0 (one launch per block step) or 1 (one persistent launch)");
    ctx->potrf_mode = mode;
    return DGPAMD_OK;
}

extern "C" int dgpamd_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch, double *logdet,
                            int32_t *info, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !A || !logdet || !info || !work) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    const int64_t Np = padded_dim(n);
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    return run_potrf(ctx, n, A, stride_a, batch, logdet, info, (double *)work, nullptr, nullptr);
}

extern "C" int dgpamd_potrf_inv(dgpamd_ctx *ctx, int64_t n, double *A, double *T, double *S, int64_t stride_a, int batch,
                                double *logdet, int32_t *info, void *work) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (n <= 0 || !A || !T || !S || !logdet || !info || !work) BAD_ARG(ctx, "null pointer or n <= 0");
    if (batch <= 0 || batch > DGPAMD_MAXB) BAD_ARG(ctx, "need 1 <= batch <= DGPAMD_MAXB");
    const int64_t Np = padded_dim(n);
    if (n + 1 > Np) BAD_ARG(ctx, "no room for the right-hand-side row");
    if (batch > 1 && stride_a < Np * Np) BAD_ARG(ctx, "stride_a < Np*Np");
    return run_potrf(ctx, n, A, stride_a, batch, logdet, info, (double *)work, T, S);
}

extern "C" int dgpamd_aug_quad(dgpamd_ctx *ctx, int64_t n, const double *A, int64_t stride_a, int batch, int r,
                               double *quad) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (!A || !quad || r <= 0 || r * r > 256 || n + r > padded_dim(n)) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(aug_quad_kernel, dim3(batch), dim3(256), 0, ctx->stream, A, padded_dim(n), stride_a, n, r, quad);
    LAUNCH_CHECK(ctx);
    return DGPAMD_OK;
}

// out[i] = sum_j A[i][j] x[j]; one wave per row
__global__ __launch_bounds__(256) void gemv_kernel(const double *A, int64_t ld, int64_t rows, int64_t cols, const double *x,
                                                   double *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= rows) return;
    const double *row = A + i * ld;
    double s = 0.0;
    for (int64_t j = lane; j < cols; j += 64) s = fma(row[j], x[j], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[i] = s;
}

extern "C" int dgpamd_gemv(dgpamd_ctx *ctx, int64_t rows, int64_t cols, const double *A, int64_t ld, const double *x,
                           double *out) {
    if (!ctx) return DGPAMD_BAD_ARG;
    if (rows <= 0 || cols <= 0 || !A || !x || !out || ld < cols) BAD_ARG(ctx, "bad arguments");
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, A, ld, rows, cols, x, out);
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
