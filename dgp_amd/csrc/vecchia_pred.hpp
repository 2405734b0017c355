// Argument blocks of the Vecchia prediction kernels (csrc/vecchia.hip: the LDS kernels and the host entries;
// csrc/vecchia_pred.hip: the register-resident kernels -- a translation unit of their own: their fully unrolled
// eliminations take minutes to compile).
#pragma once
#include "common.hpp"

struct VParams {
    int kind, D, nlen;
    double inv_len[DGPAMD_MAXD];
    double nugget;
};

struct VGpArgs {
    VParams vp;
    int64_t M, n;
    int pm;
    const double *x, *w, *y, *nugget_diag;
    const int64_t *NN;
    double scale;
    double *mean, *var;
};

struct VLinkArgs {
    int kind, Dw, Dz, pm;
    int64_t M, n;
    const double *m, *v, *z, *w1, *wg, *y, *nugget_diag;
    const int64_t *NN;
    double len[DGPAMD_MAXD];
    double scale, nugget;
    double *mean, *var;
};

#define VG_BC 51   // neighbours the register-resident gp kernel holds (pm <= VG_BC, D <= 16)
#define VL_BC 50   // neighbours the register-resident link_gp kernel holds (pm <= VL_BC, Dw <= 8, Dz <= 8; both kernels)
void launch_vecchia_gp_reg(dgpamd_ctx *ctx, const VGpArgs &a);
void launch_vecchia_linkgp_sexp_reg(dgpamd_ctx *ctx, const VLinkArgs &a);
void launch_vecchia_linkgp_matern_reg(dgpamd_ctx *ctx, const VLinkArgs &a);
