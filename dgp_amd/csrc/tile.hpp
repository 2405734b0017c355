// f64 MFMA (v_mfma_f64_16x16x4_f64) tile helpers shared by the factorisation and prediction kernels.
//
// Fragment maps on gfx950 (cdna_hip_programming.md section 3): A operand lane l holds A[i = l&15][k = l>>4],
// B operand lane l holds B[k = l>>4][j = l&15], the 4 results of lane l are D[row = (l>>4) + 4*reg][col = l&15].
// A 256-thread workgroup owns a 64x64 output tile: wave w computes rows 16w..16w+15 as four 16x16 MFMA tiles.
#pragma once
#include "common.hpp"

#define LDK 80   // LDS leading dimension for k-major tiles (row = k): conflict-free fragments

enum { OP_MK = 0, OP_KM = 1 };   // operand tile storage: row = m (k contiguous) | row = k (m contiguous)
#define KC 32     // k-depth staged per barrier pair
#define LDM 34    // LDS ld of an MK half-tile [64][KC]: bank(4*row + 2*k) -> conflict-free ds_read_b64 fragments

// stage half h of a 64x64 f64 tile (row stride ldg).  MK: rows = m, columns 32h..32h+31.
__device__ __forceinline__ void load_mk(const double *__restrict__ g, int64_t ldg, double *__restrict__ s, int tid,
                                        int h) {
    const int c2 = (tid & 15) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it;
        double2 v = *reinterpret_cast<const double2 *>(g + (int64_t)r * ldg + 32 * h + c2);
        s[r * LDM + c2] = v.x;
        s[r * LDM + c2 + 1] = v.y;
    }
}
// KM: rows = k (32h..32h+31), 64 columns; k rows >= row_limit are zeroed.
__device__ __forceinline__ void load_km(const double *__restrict__ g, int64_t ldg, double *__restrict__ s, int tid,
                                        int h, int row_limit) {
    const int c2 = (tid & 31) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 5) + 8 * it;
        double2 v = make_double2(0.0, 0.0);
        if (32 * h + r < row_limit) v = *reinterpret_cast<const double2 *>(g + (int64_t)(32 * h + r) * ldg + c2);
        s[r * LDK + c2] = v.x;
        s[r * LDK + c2 + 1] = v.y;
    }
}

// The same staging split in two, for software pipelining: fetch_* issues the global loads of a half tile into
// registers (no wait), commit_* writes them to LDS one stage later, while the MFMAs of the stage between run.
struct HalfTile {
    double2 v[4];
};
__device__ __forceinline__ HalfTile fetch_mk(const double *__restrict__ g, int64_t ldg, int tid, int h) {
    HalfTile f;
    const int c2 = (tid & 15) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        f.v[it] = *reinterpret_cast<const double2 *>(g + (int64_t)((tid >> 4) + 16 * it) * ldg + 32 * h + c2);
    return f;
}
// The same fetch as sc1 loads (L1 bypassed: the way data handed over by another workgroup inside a launch is read when
// no agent-scope acquire is taken; MI355X_MICROARCH.md, hand-off recipe).  16 bytes per lane through a buffer resource
// whose base is the tile itself (offsets stay far below 4 GB).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ HalfTile fetch_mk_sc1(const double *g, int64_t ldg, int tid, int h) {
    HalfTile f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);
    const int c2 = (tid & 15) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((tid >> 4) + 16 * it) * ldg + 32 * h + c2) * 8), 0, 16);
        f.v[it] = make_double2(__hiloint2double(v.y, v.x), __hiloint2double(v.w, v.z));
    }
    return f;
}
__device__ __forceinline__ void commit_mk(const HalfTile &f, double *__restrict__ s, int tid) {
    const int c2 = (tid & 15) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it;
        s[r * LDM + c2] = f.v[it].x;
        s[r * LDM + c2 + 1] = f.v[it].y;
    }
}
__device__ __forceinline__ HalfTile fetch_km(const double *__restrict__ g, int64_t ldg, int tid, int h, int row_limit) {
    HalfTile f;
    const int c2 = (tid & 31) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 5) + 8 * it;
        f.v[it] = make_double2(0.0, 0.0);
        if (32 * h + r < row_limit) f.v[it] = *reinterpret_cast<const double2 *>(g + (int64_t)(32 * h + r) * ldg + c2);
    }
    return f;
}
__device__ __forceinline__ void commit_km(const HalfTile &f, double *__restrict__ s, int tid) {
    const int c2 = (tid & 31) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 5) + 8 * it;
        s[r * LDK + c2] = f.v[it].x;
        s[r * LDK + c2 + 1] = f.v[it].y;
    }
}

// LDS-qualified volatile view of a staged tile.  Volatile keeps every fragment read a single ds_read_b64 (2 LDS cycles
// per wave, banks modulo 64: conflict-free with LDM = 34 / LDK = 80); left alone the compiler pairs them into
// ds_read2_b64, which takes 8 cycles and banks modulo 32 (MI355X_MICROARCH.md, LDS table) -- 2-way conflicts here.
typedef volatile double __attribute__((address_space(3))) vlds_double;

// acc[t] += sign * A(16w.., :) * B(:, 16t..)   over the KC-deep staged half tiles
template <int OPA, int OPB>
__device__ __forceinline__ void mfma_tile(const double *As_, const double *Bs_, d4 acc[4], int wave, int lane,
                                          double sign) {
    const vlds_double *As = (const vlds_double *)As_, *Bs = (const vlds_double *)Bs_;
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
        double av = (OPA == OP_MK) ? As[(16 * wave + m) * LDM + k0 + kk] : As[(k0 + kk) * LDK + 16 * wave + m];
        av *= sign;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double bv = (OPB == OP_MK) ? Bs[(16 * t + m) * LDM + k0 + kk] : Bs[(k0 + kk) * LDK + 16 * t + m];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
        }
    }
}

// The same product with the operand fragments requested two k-steps ahead of their MFMAs (a ring of three steps, held in
// place by scheduling barriers): for code that runs alone on its SIMD, where a fragment read waited for in front of its
// MFMA exposes the LDS latency.  Same operations in the same order as mfma_tile: the same bits.
template <int OPA, int OPB>
__device__ __forceinline__ void mfma_tile_ahead(const double *As_, const double *Bs_, d4 acc[4], int wave, int lane,
                                                double sign) {
    const vlds_double *As = (const vlds_double *)As_, *Bs = (const vlds_double *)Bs_;
    const int m = lane & 15, kk = lane >> 4;
    double av[3], bv[3][4];
    auto rd = [&](int s) {
        const int k0 = 4 * s, o = s % 3;
        av[o] = (OPA == OP_MK) ? As[(16 * wave + m) * LDM + k0 + kk] : As[(k0 + kk) * LDK + 16 * wave + m];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[o][t] = (OPB == OP_MK) ? Bs[(16 * t + m) * LDM + k0 + kk] : Bs[(k0 + kk) * LDK + 16 * t + m];
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {
        if (s + 2 < KC / 4) rd(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        const double a = av[s % 3] * sign;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s % 3][t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// MK half-tile with row/column masking (rows >= row_limit or columns >= col_limit read as zero)
__device__ __forceinline__ void load_mk_masked(const double *__restrict__ g, int64_t ldg, double *__restrict__ s,
                                               int tid, int h, int row_limit, int col_limit) {
    const int c2 = (tid & 15) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it;
        const int c = 32 * h + c2;
        double v0 = 0.0, v1 = 0.0;
        if (r < row_limit) {
            if (c < col_limit) v0 = g[(int64_t)r * ldg + c];
            if (c + 1 < col_limit) v1 = g[(int64_t)r * ldg + c + 1];
        }
        s[r * LDM + c2] = v0;
        s[r * LDM + c2 + 1] = v1;
    }
}
