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
    double f[16 * 16];      // the 16x16 diagonal tile on its way from accumulator layout to one column per lane
    double nt[16 * 17];     // nt[c * 17 + i] = N[i][c]: unit-lower inverse of the tile's LDL^T factor, unscaled
    double s[16];           // 1 / sqrt(pivot) of the current 16-block (1 for carried rows)
    double u[4][4][64];     // u[w][r][lane] = U[J][w] in accumulator layout (wave w's tile of the current block row)
    double piv[64];         // pivots (1 for carried rows): the log-determinant's input
    double ident[16 * 16];  // identity (the start of the inverse's columns; read instead of 16 compares per lane)
    int bad[4];             // per 16-block: 1 + index of its first non-positive pivot, or 0
};

struct Tile64 {
    d4 v[4];
};

typedef volatile double __attribute__((address_space(3))) vlds_f64;

// ---- the elimination of a full 16x16 tile, hand scheduled -------------------------------------------------------
// Pivot J: rows i > J of the lane's columns x (block) and y (identity) take  row_i += (lane J's x[i]) * (np | nq), with
// np / nq = -x[J]/pivot, -y[J]/pivot of THIS lane (the scaled pivot row) -- two DPP multiply-adds per row.  The chain of
// the NEXT pivot (broadcast, hardware reciprocal, two Newton rounds, its np / nq: eight dependent operations) starts as
// soon as row J+1 is final and is interleaved one operation at a time with the remaining rows of pivot J: the wave
// issues in order, so the placement is fixed here with volatile asm instead of being left to the compiler (which puts
// the chain into one run of dependent instructions and stalls ~50 cycles per pivot).
template <int JN>
__device__ __forceinline__ void chain_op(const int k, const double &xj, const double &yj, double &db, double &r0, double &e,
                                         double &np2, double &nq2) {
    switch (k) {
        case 0:   // s_nop 1: a DPP source written by the previous VALU instruction needs two wait states
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(db) : "v"(xj), "i"(JN));
            break;
        case 1: asm volatile("v_rcp_f64_e32 %0, %1" : "=v"(r0) : "v"(db)); break;
        case 2: asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(db), "v"(r0)); break;   // (trans result: 1 wait state)
        case 3: asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r0) : "v"(e)); break;
        case 4: asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(db), "v"(r0)); break;
        case 5: asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r0) : "v"(e)); break;
        case 6: asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(np2) : "v"(xj), "v"(r0)); break;
        default: asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nq2) : "v"(yj), "v"(r0)); break;
    }
}
// row-pair index (0 = row J+1) after which chain operation k is issued (or after the last row when there are fewer)
__host__ __device__ constexpr int chain_slot(int k) { return k == 0 ? 0 : (k == 1 ? 1 : (k < 7 ? k + 1 : 7)); }

template <int J>
struct ElimFull {
    static __device__ __forceinline__ void run(double (&x)[16], double (&y)[16], const double np, const double nq) {
        if constexpr (J < 15) {
            constexpr int P = 15 - J;            // rows below the pivot
            constexpr bool next = (J + 1 < 15);  // pivot 15 eliminates nothing: no reciprocal needed
            double db = 0.0, r0 = 0.0, e = 0.0, np2 = 0.0, nq2 = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int i = J + 1 + p;
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y[i]) : "v"(x[i]), "v"(nq), "i"(J));
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(x[i]), "v"(np), "i"(J));
                if (next) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if ((chain_slot(k) < P - 1 ? chain_slot(k) : P - 1) == p)
                            chain_op<(J + 1 < 16 ? J + 1 : 15)>(k, x[J + 1 < 16 ? J + 1 : 15], y[J + 1 < 16 ? J + 1 : 15], db, r0, e, np2, nq2);
                }
            }
            ElimFull<J + 1>::run(x, y, np2, nq2);
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
                                          long long *stamp) {
    const int lm = l & 15, lu = l >> 4;
    vlds_f64 *f = (vlds_f64 *)sh.f, *nt = (vlds_f64 *)sh.nt, *ss = (vlds_f64 *)sh.s, *piv = (vlds_f64 *)sh.piv,
             *ident = (vlds_f64 *)sh.ident;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        const int nact = ncol - 16 * J >= 16 ? 16 : (ncol - 16 * J > 0 ? ncol - 16 * J : 0);   // pivots of this 16-block
        if (W == 0 && stamp && l == 0) stamp[2 * J] = wall_clock64();
        if (W == J) {
            // ---- accumulator layout -> LDS (same wave from here to B1: the LDS operations of a wave are in order) ----
#pragma unroll
            for (int r = 0; r < 4; ++r) f[(lu + 4 * r) * 16 + lm] = X[J][r];
            if (nact == 16) {
                // one column per lane (the four DPP rows hold copies), the identity beside it
                double x[16], y[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    x[i] = f[i * 16 + lm];
                    y[i] = ident[i * 16 + lm];
                }
                double db, r0, e, np, nq;
#pragma unroll
                for (int k = 0; k < 8; ++k) chain_op<0>(k, x[0], y[0], db, r0, e, np, nq);
                ElimFull<0>::run(x, y, np, nq);
                if (lu == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        f[i * 16 + lm] = x[i];
                        nt[lm * 17 + i] = y[i];
                    }
                }
            } else {
                // the block that holds the end of the matrix (at most one per factorisation), or carried rows only:
                // the same elimination on the tile in LDS, pivot by pivot, four entries of each matrix per lane
                double xv[4], yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xv[r] = X[J][r];
                    yv[r] = (lu + 4 * r == lm) ? 1.0 : 0.0;
                }
                for (int j = 0; j < nact; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[(lu + 4 * r) * 16 + lm] = xv[r];
                        nt[lm * 17 + lu + 4 * r] = yv[r];
                    }
                    const double rp = rcp_f64(f[j * 17]), px = f[j * 16 + lm], py = nt[lm * 17 + j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double m = (lu + 4 * r > j) ? f[(lu + 4 * r) * 16 + j] * rp : 0.0;
                        xv[r] = fma(-m, px, xv[r]);
                        yv[r] = fma(-m, py, yv[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[(lu + 4 * r) * 16 + lm] = xv[r];
                    nt[lm * 17 + lu + 4 * r] = yv[r];
                }
            }
            // pivots, Cholesky scaling 1/sqrt(pivot) of the rows, first non-positive pivot
            const bool act = lm < nact;
            const double dsel = act ? f[lm * 17] : 1.0;
            const double s = act ? rsqrt_f64(dsel) : 1.0;
            const unsigned long long bm = __ballot(act && !(dsel > 0.0)) & 0xffffull;
            if (lu == 0) {
                ss[lm] = s;
                piv[16 * J + lm] = dsel;
                if (lm == 0) sh.bad[J] = bm ? 16 * J + __builtin_ctzll(bm) + 1 : 0;
            }
            // U_JJ and V_J in accumulator layout
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lu + 4 * r;
                const double sr = ss[row];
                // below the diagonal of a pivot column the eliminated entries are zero up to rounding: exact zeros
                X[J][r] = (row > lm && lm < nact) ? 0.0 : f[row * 16 + lm] * sr;
                Y[J][r] = nt[lm * 17 + row] * sr;
            }
        }
        lds_barrier();   // B1: V_J (nt, s) visible
        if (W == 0 && stamp && l == 0) stamp[2 * J + 1] = wall_clock64();
        if (W != J) {
            // ---- solve: tile <- V_J * tile   (column block W of A for W > J, of the inverse for W < J) ----
            const double sl = ss[lm];
            double vf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) vf[kk] = nt[(4 * kk + lu) * 17 + lm] * sl;   // V[lm][4kk + lu]
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

// Factor the 64x64 block held as X (column-block layout above; only tiles t <= w are read), write the lower factor to
// Ab (row stride ld; strictly upper part zeroed, carried corner kept) and its inverse to Wb (64x64, row-major).
// ncol = pivots in this block.  Leaves the pivots in sh.piv and returns (every thread) 1 + the index of the first
// non-positive pivot, or 0.  `stamp`: optional 9 slots of wall_clock64 stamps (wave 0, lane 0): start of 16-block J, its first barrier, end.
__device__ __forceinline__ int diag_factor(Tile64 &tile, DiagShared &sh, double *Ab, int64_t ld, double *Wb, int ncol_,
                                           long long *stamp = nullptr) {
    d4 (&X)[4] = tile.v;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lm = l & 15, lu = l >> 4;
    const int ncol = __builtin_amdgcn_readfirstlane(ncol_);
    d4 Y[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Y[t] = (d4){0.0, 0.0, 0.0, 0.0};
    sh.ident[tid] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    lds_barrier();
    switch (w) {
        case 0: diag_wave<0>(X, Y, sh, ncol, l, stamp); break;
        case 1: diag_wave<1>(X, Y, sh, ncol, l, stamp); break;
        case 2: diag_wave<2>(X, Y, sh, ncol, l, stamp); break;
        default: diag_wave<3>(X, Y, sh, ncol, l, stamp); break;
    }
    if (stamp && tid == 0) stamp[8] = wall_clock64();

    // ---- results: W (row-major 64x64, coalesced) ----
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) Wb[(16 * I + lu + 4 * r) * 64 + 16 * w + lm] = (I >= w) ? Y[I][r] : 0.0;
    // ---- L = U^T: tile (t, w) of U is tile (w, t) of L, element (row 16w + lm, column 16t + lu + 4r) ----
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
    lds_barrier();
    int bad = 0;
#pragma unroll
    for (int J = 3; J >= 0; --J) bad = sh.bad[J] ? sh.bad[J] : bad;
    return bad;
}
