// 64x64 diagonal block of the blocked Cholesky: factor + inverse of the factor, one 256-thread workgroup.
// (the serial part of scipy's potrf in kernel_class.py:417,483 / numpy's cholesky in functions.py:119)
//
// Layout.  The block is symmetric, so wave w owns COLUMN block w of it as four 16x16 tiles in the f64 MFMA accumulator
// layout:  X[t][r] = S[16t + lu + 4r][16w + lm]   (lm = lane & 15, lu = lane >> 4).  An accumulator tile D is, register
// for register, the B operand of D and the A operand of D^T (k-step r = register r), so with A = U^T U (U upper):
//   solve    U[J][w]  = V_J S[J][w]                 A = V_J (from LDS), B = the tile itself
//   update   S[I][w] -= U[J][I]^T U[J][w]           A = registers of U[J][I] (exchanged through LDS), B = U[J][w]
// run on v_mfma_f64_16x16x4 without any transposition.  V_J = L_JJ^-1 comes from the factorisation of the 16x16
// diagonal tile, the only serial part:
//   wave J turns its tile into one COLUMN per lane (16 registers; the four 16-lane DPP rows hold copies) next to a
//   column of the identity, and runs Gaussian elimination without scaling: per pivot j one broadcast of the pivot
//   (v_mov_b64_dpp row_newbcast), a Newton reciprocal, and per remaining row i two v_fmac_f64_dpp that read lane j's
//   x[i] through the DPP -- no LDS, no barrier, no scalar round trip inside the 16 pivots.  The identity columns end
//   as N = (unit lower factor)^-1; the Cholesky scaling 1/sqrt(d_j) is applied to rows when the results are read.
// The same row operations applied to the identity columns of the other waves give the inverse of the whole 64x64
// factor: wave c owns column block c of W = L^-1 (tiles Y[I], I >= c), solved / updated with the same two MFMA forms.
// Per 16 pivots: one elimination (~0.7 us), two barriers, 8 dependent MFMAs.
//
// Blocks with fewer than 64 pivots (the last one: rows / columns >= ncol are carried right-hand sides): the carried
// rows take part in the row operations but are never pivots.
#pragma once
#include "common.hpp"

__device__ __forceinline__ double rsqrt_f64(double d) {
    // Goldschmidt from the hardware estimate: depth 6
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return h + h;
}

__device__ __forceinline__ double rcp_f64(double d) {
    // hardware estimate + two Newton rounds: depth 5
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    return fma(x, e, x);
}

struct DiagShared {
    double f[16 * 17];      // the 16x16 diagonal tile, f[row * 17 + col]: in, then the elimination's result (upper part: the
                            // unscaled factor, below the diagonal of pivot columns: -pivot_col * (unit lower factor)^-1)
    double s[16];           // 1 / sqrt(pivot) of the current 16-block (1 for carried rows)
    double cs[16];          // -1 / pivot (0 for carried columns)
    double u[4][4][64];     // u[w][r][lane] = U[J][w] in accumulator layout (wave w's tile of the current block row)
    double piv[64];         // pivots (1 for carried rows): the log-determinant's input
    int bad[4];             // per 16-block: 1 + index of its first non-positive pivot, or 0
    int ready;              // DiagPoll: bit i = the word lane i polled beside the last 16-block's elimination had reached its value
    int pcount;             // one-launch chain: waves whose stores of the panel tile have drained (a running count)
    int look;               // one-launch chain: bits of `ready` a second look found set (a word of its own: `ready` itself must not
                            // change between the factorisation's last barrier and the moment the SLOWEST wave has read it)
};

// Two version words (and the values they must reach) that wave 0 polls ONCE while wave 3 eliminates the last 16-block --
// the wave has nothing else to do then, so the ~1 us round trip of the poll costs the factorisation nothing.  The
// one-launch kernel's chain asks this way whether its next inputs have arrived.
struct DiagPoll {
    const int *f;   // per lane: lanes 0 and 1 poll one word each (the caller fills both fields per lane: a select between two
    int need;       // struct members on the lane index kept the struct in scratch memory, a scratch load beside the poll)
};

// What the one-launch kernel's chain hands to the factorisation that follows its block step.
struct DiagShadow {
    double *wout;       // global, or null: W = L^-1 (row-major 64x64) is stored from inside the factorisation, row block J as soon as
                        // it is final (write-through stores; drained before the factorisation's last barrier): the caller can
                        // publish it at once instead of storing and draining 32 KB behind the factorisation
    const double *hand; // LDS, or null: wave 3's tile X[0] as another wave computed it (hand[r * 64 + lane]); read behind the
                        // first barrier, ahead of the tile's first use
};

struct Tile64 {
    d4 v[4];
};

typedef volatile double __attribute__((address_space(3))) vlds_f64;

// ---- the elimination of a full 16x16 tile, hand scheduled -------------------------------------------------------
// One register column z per lane holds BOTH results.  At pivot j the lanes c > j still carry the block (z_c[i] = X[i][c]),
// the lanes c < j carry the inverse of the unit lower factor scaled by -pivot_c (z_c[i] = -d_c N[i][c], i > c), and lane j
// changes sides: its entries below the pivot are the multipliers' numerators X[i][j] and ARE -d_j N[i][j] = X[i][j] as they
// stand.  With nz_c = -z_c[j] / d_j  (0 in lane j) every row i > j takes
//        z_c[i] += (lane j's z[i]) * nz_c
// in ONE v_fmac_f64_dpp (row_newbcast:j reads lane j of the 16-lane row): block and inverse advance together.
// The chain of the NEXT pivot (broadcast, hardware reciprocal, two Newton rounds, nz: seven dependent operations)
// starts as soon as row j+1 is final and is interleaved one operation at a time with the remaining rows of pivot j: the
// wave issues in order, so the placement is fixed here with volatile asm instead of being left to the compiler (which
// puts the chain into one run of dependent instructions and stalls ~50 cycles per pivot).
template <int JN>
__device__ __forceinline__ void chain_op(const int k, const double &zj, const double &zz, double &db, double &r0, double &e,
                                         double &nz) {
    switch (k) {
        case 0:   // s_nop 1: a DPP source written by the previous VALU instruction needs two wait states
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(db) : "v"(zj), "i"(JN));
            break;
        case 1: asm volatile("v_rcp_f64_e32 %0, %1" : "=v"(r0) : "v"(db)); break;
        case 2: asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(db), "v"(r0)); break;   // (trans result: 1 wait state)
        case 3: asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r0) : "v"(e)); break;
        case 4: asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(db), "v"(r0)); break;
        case 5: asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r0) : "v"(e)); break;
        default: asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nz) : "v"(zz), "v"(r0)); break;
    }
}
// row index (0 = row j+1) after which chain operation k is issued (or after the last row when there are fewer)
__host__ __device__ constexpr int chain_slot(int k) { return k == 0 ? 0 : (k == 1 ? 1 : k + 1); }

// zz = (lane == JN) ? 0 : z   (in asm: left to the compiler the sixteen lane compares are hoisted and cost 32 SGPRs)
template <int JN>
__device__ __forceinline__ double zero_in_lane(const double &z, const int lm) {
    const int zlo = __double2loint(z), zhi = __double2hiint(z);
    int lo, hi;
    asm volatile("v_cmp_ne_u32_e32 vcc, %4, %5\n\tv_cndmask_b32_e32 %0, 0, %2, vcc\n\tv_cndmask_b32_e32 %1, 0, %3, vcc"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(zlo), "v"(zhi), "i"(JN), "v"(lm)
                 : "vcc");
    return __hiloint2double(hi, lo);
}

template <int J>
struct ElimFull {
    static __device__ __forceinline__ void run(double (&z)[16], const int lm, const double nz) {
        if constexpr (J < 15) {
            constexpr int P = 15 - J;            // rows below the pivot
            constexpr bool next = (J + 1 < 15);  // pivot 15 eliminates nothing: no reciprocal needed
            constexpr int JN = J + 1 < 16 ? J + 1 : 15;
            double db = 0.0, r0 = 0.0, e = 0.0, nz2 = 0.0, zz = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int i = J + 1 + p;
                asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(z[i]) : "v"(nz), "i"(J));
                if (next) {
                    if (p == 0) zz = zero_in_lane<JN>(z[JN], lm);   // (off the chain: needs row J+1 only)
#pragma unroll
                    for (int k = 0; k < 7; ++k)
                        if ((chain_slot(k) < P - 1 ? chain_slot(k) : P - 1) == p) chain_op<JN>(k, z[JN], zz, db, r0, e, nz2);
                }
            }
            ElimFull<J + 1>::run(z, lm, nz2);
        }
    }
};

__device__ __forceinline__ void lds_barrier() {
    // LDS traffic only: global stores / loads stay in flight across it (__syncthreads would drain vmcnt as well)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The program of wave W (compile-time: every role decision below is static, the four waves run four straight-line
// programs that meet at the same sequence of barriers).  X[t] (t <= W): column block W of the block; Y[I] (I >= W):
// column block W of the inverse.
template <int W>
__device__ __forceinline__ void diag_wave(d4 (&X)[4], d4 (&Y)[4], DiagShared &sh, const int ncol, const int l,
                                          long long *stamp, const DiagPoll *poll, DiagShadow *shadow) {
    const int lm = l & 15, lu = l >> 4;
    vlds_f64 *f = (vlds_f64 *)sh.f, *ss = (vlds_f64 *)sh.s, *cs = (vlds_f64 *)sh.cs, *piv = (vlds_f64 *)sh.piv;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        const int nact = ncol - 16 * J >= 16 ? 16 : (ncol - 16 * J > 0 ? ncol - 16 * J : 0);   // pivots of this 16-block
        if (W == 0 && stamp && l == 0) stamp[2 * J] = wall_clock64();
        if (W == J) {
            // ---- accumulator layout -> LDS (same wave from here to B1: the LDS operations of a wave are in order) ----
#pragma unroll
            for (int r = 0; r < 4; ++r) f[(lu + 4 * r) * 17 + lm] = X[J][r];
            if (nact == 16) {
                // one column per lane (the four DPP rows hold copies)
                double z[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = f[i * 17 + lm];
                double db, r0, e, nz;
                const double zz = zero_in_lane<0>(z[0], lm);
#pragma unroll
                for (int k = 0; k < 7; ++k) chain_op<0>(k, z[0], zz, db, r0, e, nz);
                ElimFull<0>::run(z, lm, nz);
                if (lu == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) f[i * 17 + lm] = z[i];
                }
            } else {
                // the block that holds the end of the matrix (at most one per factorisation), or carried rows only:
                // the same recurrence on the tile in LDS, pivot by pivot, four entries per lane
                double zv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) zv[r] = X[J][r];
                for (int j = 0; j < nact; ++j) {
                    const double rp = rcp_f64(f[j * 18]), zz = (lm == j) ? 0.0 : f[j * 17 + lm];
                    const double nzj = -zz * rp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double m = (lu + 4 * r > j) ? f[(lu + 4 * r) * 17 + j] : 0.0;
                        zv[r] = fma(m, nzj, zv[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[(lu + 4 * r) * 17 + lm] = zv[r];
                }
            }
            // pivots, Cholesky scaling 1/sqrt(pivot) of the rows, first non-positive pivot
            const bool act = lm < nact;
            const double dsel = act ? f[lm * 18] : 1.0;
            const double s = act ? rsqrt_f64(dsel) : 1.0;
            const unsigned long long bm = __ballot(act && !(dsel > 0.0)) & 0xffffull;
            if (lu == 0) {
                ss[lm] = s;
                cs[lm] = act ? -s * s : 0.0;   // column scale of the inverse: -1 / pivot (carried columns: identity)
                piv[16 * J + lm] = dsel;
                if (lm == 0) sh.bad[J] = bm ? 16 * J + __builtin_ctzll(bm) + 1 : 0;
            }
            // U_JJ and V_J in accumulator layout
            const double csl = cs[lm];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lu + 4 * r;
                const double sr = ss[row], fv = f[row * 17 + lm] * sr;
                // below the diagonal of a pivot column the tile holds the inverse
                X[J][r] = (row > lm && lm < nact) ? 0.0 : fv;
                Y[J][r] = (row == lm) ? sr : (row > lm ? fv * csl : 0.0);
            }
        }
        if (W == 0 && J == 3 && poll && poll->f) {
            const bool ok = l >= 2 || __hip_atomic_load(poll->f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= poll->need;
            const unsigned long long arrived = __ballot(ok);
            if (l == 0) sh.ready = (int)(arrived & 3ull);   // bit 0: lane 0's word had reached its value, bit 1: lane 1's
            // the acquire for those inputs is taken here too (it completes beside the elimination); if the poll failed
            // the caller waits and acquires again
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();   // B1: V_J (s, cs, f) visible
        if (J == 0 && W == 3 && shadow && shadow->hand) {
            const vlds_f64 *hd = (const vlds_f64 *)shadow->hand;
#pragma unroll
            for (int r = 0; r < 4; ++r) X[0][r] = hd[r * 64 + l];
        }
        if (W == 0 && stamp && l == 0) stamp[2 * J + 1] = wall_clock64();
        if (W != J) {
            // ---- solve: tile <- V_J * tile   (column block W of A for W > J, of the inverse for W < J) ----
            const double sl = ss[lm];
            double vf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // V[lm][4kk + lu]: s on the diagonal, s_row * f * (-1/pivot_col) below it
                const int col = 4 * kk + lu;
                const double below = f[lm * 17 + col] * sl * cs[col];
                vf[kk] = (lm == col) ? sl : (lm > col ? below : 0.0);
            }
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
            d4 &B = (W > J) ? X[J] : Y[J];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[kk], B[kk], acc, 0, 0, 0);
            B = acc;
            if (W > J) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sh.u[W][r][l] = acc[r];
            }
        }
        if (shadow && shadow->wout) {   // row block J of the inverse is final: wave W holds its columns 16W .. 16W + 15 (zero right of the diagonal)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(shadow->wout + (16 * J + lu + 4 * r) * 64 + 16 * W + lm, (J >= W) ? Y[J][r] : 0.0, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (J == 3) break;
        lds_barrier();   // B2: U[J][w'] of the other waves visible
        // ---- update with block row J (carried rows of the block are no pivots: masked out of the A operand) ----
        const bool whole = ncol >= 16 * J + 16;   // every row of the block row is a pivot (all but the last block step)
        const d4 B = (W > J) ? X[J] : Y[J];
#pragma unroll
        for (int I = J + 1; I < 4; ++I) {
            // W > J: tiles I = J+1 .. W of A's column block;  W <= J: tiles I = J+1 .. 3 of the inverse's column block
            if (W > J && I > W) continue;
            double a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double uv = (I == W) ? X[J][r] : sh.u[I][r][l];
                a[r] = -uv;
            }
            if (!whole) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = (16 * J + lu + 4 * r < ncol) ? a[r] : 0.0;
            }
            d4 &C = (W > J) ? X[I] : Y[I];
#pragma unroll
            for (int r = 0; r < 4; ++r) C = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], B[r], C, 0, 0, 0);
        }
    }
}

// Factor the 64x64 block held as X (column-block layout above; only tiles t <= w are read) in place: X becomes the upper
// factor U = L^T, and Winv column block w of the inverse L^-1 in the same layout (tiles above the diagonal zero).
// ncol = pivots in this block.  Leaves the pivots in sh.piv and returns (every thread) 1 + the index of the first
// non-positive pivot, or 0.  `stamp`: optional 9 slots of wall_clock64 stamps (wave 0, lane 0): start of 16-block J, its
// first barrier, end.
__device__ __forceinline__ int diag_factor(Tile64 &tile, Tile64 &Winv, DiagShared &sh, int ncol_, long long *stamp = nullptr,
                                           const DiagPoll *poll = nullptr, DiagShadow *shadow = nullptr) {
    d4 (&X)[4] = tile.v;
    d4 (&Y)[4] = Winv.v;
    int tid = threadIdx.x;
    // opaque to the optimiser: inside a loop over block steps the dozens of lane predicates below would otherwise be
    // hoisted out of it and held in SGPR pairs for the whole loop (spills)
    asm volatile("" : "+v"(tid));
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int ncol = __builtin_amdgcn_readfirstlane(ncol_);
#pragma unroll
    for (int t = 0; t < 4; ++t) Y[t] = (d4){0.0, 0.0, 0.0, 0.0};
    switch (w) {
        case 0: diag_wave<0>(X, Y, sh, ncol, l, stamp, poll, shadow); break;
        case 1: diag_wave<1>(X, Y, sh, ncol, l, stamp, poll, shadow); break;
        case 2: diag_wave<2>(X, Y, sh, ncol, l, stamp, poll, shadow); break;
        default: diag_wave<3>(X, Y, sh, ncol, l, stamp, poll, shadow); break;
    }
    if (stamp && tid == 0) stamp[8] = wall_clock64();
    if (shadow && shadow->wout) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's stores of the inverse have drained)
    lds_barrier();
    int bad = 0;
#pragma unroll
    for (int J = 3; J >= 0; --J) bad = sh.bad[J] ? sh.bad[J] : bad;
    return bad;
}

// W = L^-1 (row-major 64x64, coalesced rows) from the column blocks diag_factor leaves in registers.
// SC1: write-through stores (MI355X_MICROARCH.md hand-off recipe: payload of an in-launch hand-off).
template <bool SC1>
__device__ __forceinline__ void diag_store_inverse(const Tile64 &Winv, double *Wb) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lm = l & 15, lu = l >> 4;
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *p = Wb + (16 * I + lu + 4 * r) * 64 + 16 * w + lm;
            const double v = (I >= w) ? Winv.v[I][r] : 0.0;
            if (SC1)
                __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *p = v;
        }
}

// The lower factor L = U^T of a block factored by diag_factor to Ab (row stride ld): strictly upper part zeroed, carried
// corner (rows and columns >= ncol) kept symmetric.
__device__ __forceinline__ void diag_store_factor(const Tile64 &tile, double *Ab, int64_t ld, int ncol) {
    const d4 (&X)[4] = tile.v;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lm = l & 15, lu = l >> 4;
    // L = U^T: tile (t, w) of U is tile (w, t) of L, element (row 16w + lm, column 16t + lu + 4r) ----
    const int64_t lrow = 16 * w + lm;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t lcol = 16 * t + lu + 4 * r;
            if (t <= w) {
                Ab[lrow * ld + lcol] = X[t][r];
                // carried corner (rows and columns >= ncol): kept symmetric
                if (t < w && lcol >= ncol) Ab[lcol * ld + lrow] = X[t][r];
            } else if (!(lrow >= ncol && lcol >= ncol)) {
                Ab[lrow * ld + lcol] = 0.0;   // strictly upper part of the factor
            }
        }
}
