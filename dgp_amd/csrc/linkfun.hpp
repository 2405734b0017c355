// Closed forms of E[k(x,Z)] and E[k(x,Z)k(x',Z)] for the Matern-2.5 kernel under Z ~ N(m, v):
// the per-dimension factors of the linked-GP I and J (functions.py:453-494, vecchia.py:915-988).
// The expressions keep the reference's term order (cancellation-prone; f64 throughout).
#pragma once
#include "common.hpp"

#include <math.h>

__device__ __forceinline__ double matern_point(double d, double l) {
    double a = fabs(d);
    return (1.0 + SQRT5 * a / l + 5.0 * d * d / (3.0 * l * l)) * exp(-SQRT5 * a / l);
}

// functions.py:463-471 (one dimension of IJ_matern's I)
static __device__ double matern_I_dim(double xk, double zm, double zv, double l) {
    const double zX = zm - xk;
    if (zv == 0.0) return matern_point(zX, l);
    const double muA = zX - SQRT5 * zv / l, muB = zX + SQRT5 * zv / l;
    const double sv = sqrt(0.5 * zv / M_PI) / l, s2 = sqrt(2.0 * zv);
    double t1 = exp((5.0 * zv - 2.0 * SQRT5 * l * zX) / (2.0 * l * l)) *
                ((1.0 + SQRT5 * muA / l + 5.0 * (muA * muA + zv) / (3.0 * l * l)) * 0.5 * (1.0 + erf(muA / s2)) +
                 (SQRT5 + (5.0 * muA) / (3.0 * l)) * sv * exp(-0.5 * muA * muA / zv));
    double t2 = exp((5.0 * zv + 2.0 * SQRT5 * l * zX) / (2.0 * l * l)) *
                ((1.0 - SQRT5 * muB / l + 5.0 * (muB * muB + zv) / (3.0 * l * l)) * 0.5 * (1.0 + erf(-muB / s2)) +
                 (SQRT5 - (5.0 * muB) / (3.0 * l)) * sv * exp(-0.5 * muB * muB / zv));
    return t1 + t2;
}

// vecchia.py:915-959
static __device__ double matern_Jd(double X1, double X2, double z_m, double z_v, double l) {
    const double x1 = fmin(X1, X2), x2 = fmax(X1, X2);
    const double l2 = l * l, l3 = l2 * l, l4 = 9.0 * l2 * l2;
    const double x1s = x1 * x1, x2s = x2 * x2, x12 = x1 * x2, xs = x1 + x2;
    const double sv = sqrt(0.5 * z_v / M_PI), s2 = sqrt(2.0 * z_v);
    const double zv2 = z_v * z_v;

    const double E30 = 1.0 + (25.0 * x1s * x2s - 3.0 * SQRT5 * (3.0 * l3 + 5.0 * l * x12) * xs + 15.0 * l2 * (x1s + x2s + 3.0 * x12)) / l4;
    const double E31 = (18.0 * SQRT5 * l3 + 15.0 * SQRT5 * l * (x1s + x2s) - (75.0 * l2 + 50.0 * x12) * xs + 60.0 * SQRT5 * l * x12) / l4;
    const double E32 = 5.0 * (5.0 * x1s + 5.0 * x2s + 15.0 * l2 - 9.0 * SQRT5 * l * xs + 20.0 * x12) / l4;
    const double E33 = 10.0 * (3.0 * SQRT5 * l - 5.0 * x1 - 5.0 * x2) / l4;
    const double E34 = 25.0 / l4;
    const double muC = z_m - 2.0 * SQRT5 * z_v / l;
    const double c2 = muC * muC, c3 = c2 * muC, c4 = c2 * c2;
    const double E3A31 = E30 + muC * E31 + (c2 + z_v) * E32 + (c3 + 3.0 * z_v * muC) * E33 + (c4 + 6.0 * z_v * c2 + 3.0 * zv2) * E34;
    const double E3A32 = E31 + (muC + x2) * E32 + (c2 + 2.0 * z_v + x2s + muC * x2) * E33 +
                         (c3 + x2s * x2 + x2 * c2 + muC * x2s + 3.0 * z_v * x2 + 5.0 * z_v * muC) * E34;
    const double P1 = exp((10.0 * z_v + SQRT5 * l * (xs - 2.0 * z_m)) / l2) *
                      (0.5 * E3A31 * (1.0 + erf((muC - x2) / s2)) + E3A32 * sv * exp(-0.5 * (x2 - muC) * (x2 - muC) / z_v));

    const double E40 = 1.0 + (25.0 * x1s * x2s + 3.0 * SQRT5 * (3.0 * l3 - 5.0 * l * x12) * (x2 - x1) + 15.0 * l2 * (x1s + x2s - 3.0 * x12)) / l4;
    const double E41 = 5.0 * (3.0 * SQRT5 * l * (x2s - x1s) + 3.0 * l2 * xs - 10.0 * x12 * xs) / l4;
    const double E42 = 5.0 * (5.0 * x1s + 5.0 * x2s - 3.0 * l2 - 3.0 * SQRT5 * l * (x2 - x1) + 20.0 * x12) / l4;
    const double E43 = -50.0 * (X1 + X2) / l4;
    const double E44 = 25.0 / l4;
    const double m2 = z_m * z_m, m3 = m2 * z_m, m4 = m2 * m2;
    const double E4A41 = E40 + z_m * E41 + (m2 + z_v) * E42 + (m3 + 3.0 * z_v * z_m) * E43 + (m4 + 6.0 * z_v * m2 + 3.0 * zv2) * E44;
    const double E4A42 = E41 + (z_m + x1) * E42 + (m2 + 2.0 * z_v + x1s + z_m * x1) * E43 +
                         (m3 + x1s * x1 + x1 * m2 + z_m * x1s + 3.0 * z_v * x1 + 5.0 * z_v * z_m) * E44;
    const double E4A43 = E41 + (z_m + x2) * E42 + (m2 + 2.0 * z_v + x2s + z_m * x2) * E43 +
                         (m3 + x2s * x2 + x2 * m2 + z_m * x2s + 3.0 * z_v * x2 + 5.0 * z_v * z_m) * E44;
    const double P2 = exp(-SQRT5 * (x2 - x1) / l) *
                      (0.5 * E4A41 * (erf((x2 - z_m) / s2) - erf((x1 - z_m) / s2)) +
                       E4A42 * sv * exp(-0.5 * (x1 - z_m) * (x1 - z_m) / z_v) - E4A43 * sv * exp(-0.5 * (x2 - z_m) * (x2 - z_m) / z_v));

    const double E50 = 1.0 + (25.0 * x1s * x2s + 3.0 * SQRT5 * (3.0 * l3 + 5.0 * l * x12) * xs + 15.0 * l2 * (x1s + x2s + 3.0 * x12)) / l4;
    const double E51 = (18.0 * SQRT5 * l3 + 15.0 * SQRT5 * l * (x1s + x2s) + (75.0 * l2 + 50.0 * x12) * xs + 60.0 * SQRT5 * l * x12) / l4;
    const double E52 = 5.0 * (5.0 * x1s + 5.0 * x2s + 15.0 * l2 + 9.0 * SQRT5 * l * xs + 20.0 * x12) / l4;
    const double E53 = 10.0 * (3.0 * SQRT5 * l + 5.0 * x1 + 5.0 * x2) / l4;
    const double E54 = 25.0 / l4;
    const double muD = z_m + 2.0 * SQRT5 * z_v / l;
    const double d2 = muD * muD, d3 = d2 * muD, d4_ = d2 * d2;
    const double E5A51 = E50 - muD * E51 + (d2 + z_v) * E52 - (d3 + 3.0 * z_v * muD) * E53 + (d4_ + 6.0 * z_v * d2 + 3.0 * zv2) * E54;
    const double E5A52 = E51 - (muD + x1) * E52 + (d2 + 2.0 * z_v + x1s + muD * x1) * E53 -
                         (d3 + x1s * x1 + x1 * d2 + muD * x1s + 3.0 * z_v * x1 + 5.0 * z_v * muD) * E54;
    const double P3 = exp((10.0 * z_v - SQRT5 * l * (xs - 2.0 * z_m)) / l2) *
                      (0.5 * E5A51 * (1.0 + erf((x1 - muD) / s2)) + E5A52 * sv * exp(-0.5 * (x1 - muD) * (x1 - muD) / z_v));
    return P1 + P2 + P3;
}

// vecchia.py:961-988
static __device__ double matern_Jd0(double x1, double z_m, double z_v, double l) {
    const double l2 = l * l, l3 = l2 * l, l4 = 9.0 * l2 * l2;
    const double x1s = x1 * x1;
    const double sv = sqrt(0.5 * z_v / M_PI), s2 = sqrt(2.0 * z_v), zv2 = z_v * z_v;
    const double E30 = 1.0 + (25.0 * x1s * x1s - 6.0 * SQRT5 * (3.0 * l3 + 5.0 * l * x1s) * x1 + 75.0 * l2 * x1s) / l4;
    const double E31 = (18.0 * SQRT5 * l3 + 90.0 * SQRT5 * l * x1s - (150.0 * l2 + 100.0 * x1s) * x1) / l4;
    const double E32 = 5.0 * (30.0 * x1s + 15.0 * l2 - 18.0 * SQRT5 * l * x1) / l4;
    const double E33 = 10.0 * (3.0 * SQRT5 * l - 10.0 * x1) / l4;
    const double E34 = 25.0 / l4;
    const double muC = z_m - 2.0 * SQRT5 * z_v / l;
    const double c2 = muC * muC, c3 = c2 * muC, c4 = c2 * c2;
    const double E3A31 = E30 + muC * E31 + (c2 + z_v) * E32 + (c3 + 3.0 * z_v * muC) * E33 + (c4 + 6.0 * z_v * c2 + 3.0 * zv2) * E34;
    const double E3A32 = E31 + (muC + x1) * E32 + (c2 + 2.0 * z_v + x1s + muC * x1) * E33 +
                         (c3 + x1s * x1 + x1 * c2 + muC * x1s + 3.0 * z_v * x1 + 5.0 * z_v * muC) * E34;
    const double P1 = exp((10.0 * z_v + SQRT5 * l * (2.0 * x1 - 2.0 * z_m)) / l2) *
                      (0.5 * E3A31 * (1.0 + erf((muC - x1) / s2)) + E3A32 * sv * exp(-0.5 * (x1 - muC) * (x1 - muC) / z_v));
    const double E50 = 1.0 + (25.0 * x1s * x1s + 6.0 * SQRT5 * (3.0 * l3 + 5.0 * l * x1s) * x1 + 75.0 * l2 * x1s) / l4;
    const double E51 = (18.0 * SQRT5 * l3 + 90.0 * SQRT5 * l * x1s + (150.0 * l2 + 100.0 * x1s) * x1) / l4;
    const double E52 = 5.0 * (30.0 * x1s + 15.0 * l2 + 18.0 * SQRT5 * l * x1) / l4;
    const double E53 = 10.0 * (3.0 * SQRT5 * l + 10.0 * x1) / l4;
    const double E54 = 25.0 / l4;
    const double muD = z_m + 2.0 * SQRT5 * z_v / l;
    const double d2 = muD * muD, d3 = d2 * muD, d4_ = d2 * d2;
    const double E5A51 = E50 - muD * E51 + (d2 + z_v) * E52 - (d3 + 3.0 * z_v * muD) * E53 + (d4_ + 6.0 * z_v * d2 + 3.0 * zv2) * E54;
    const double E5A52 = E51 - (muD + x1) * E52 + (d2 + 2.0 * z_v + x1s + muD * x1) * E53 -
                         (d3 + x1s * x1 + x1 * d2 + muD * x1s + 3.0 * z_v * x1 + 5.0 * z_v * muD) * E54;
    const double P3 = exp((10.0 * z_v - SQRT5 * l * (2.0 * x1 - 2.0 * z_m)) / l2) *
                      (0.5 * E5A51 * (1.0 + erf((x1 - muD) / s2)) + E5A52 * sv * exp(-0.5 * (x1 - muD) * (x1 - muD) / z_v));
    return P1 + P3;
}


// ---------------------------------------------------------------------------------------------------
// Separable form of the same factor.  Every erf/exp in Jd depends on ONE of the two points, and each
// E-polynomial splits by powers of the other point (tools/derive_matern_st.py), hence
//     Jd(x1,x2) = sum_{c<12} S_c(lo) T_c(hi) + (f2(hi) - f2(lo)) * sum_{a<3} S_{6+a}(lo) T_{12+a}(hi)
// with lo = min, hi = max, f2(x) = erf((x-m)/sqrt(2v)).  The erf difference stays a pairwise unit exactly as
// in the reference expression (it cancels when both points lie on one side of a sharply peaked input).
// tools/check_matern_st.py: agrees with the direct expression to 2e-11 relative over 2000 random cases.
// ---------------------------------------------------------------------------------------------------
struct MaternDimConst {   // per (test point, dimension)
    double m, v, l, inv9l4, sv, is2, muC, muD, c5, q5;   // c5 = 5v/l^2, q5 = sqrt5/l
    double C1, C2, C3, C4, D1, D2, D3, D4, M1, M2, M3, M4;   // moment multipliers of muC, muD, m
};

static __device__ __forceinline__ void matern_dim_const(double m, double v, double l, MaternDimConst &k) {
    k.m = m; k.v = v; k.l = l;
    const double l2 = l * l;
    k.inv9l4 = 1.0 / (9.0 * l2 * l2);
    k.sv = sqrt(0.5 * v / M_PI);
    k.is2 = 1.0 / sqrt(2.0 * v);
    k.muC = m - 2.0 * SQRT5 * v / l;
    k.muD = m + 2.0 * SQRT5 * v / l;
    k.c5 = 5.0 * v / l2;
    k.q5 = SQRT5 / l;
    double mu = k.muC;
    k.C1 = mu; k.C2 = mu * mu + v; k.C3 = mu * mu * mu + 3.0 * v * mu; k.C4 = mu * mu * mu * mu + 6.0 * v * mu * mu + 3.0 * v * v;
    mu = k.muD;
    k.D1 = mu; k.D2 = mu * mu + v; k.D3 = mu * mu * mu + 3.0 * v * mu; k.D4 = mu * mu * mu * mu + 6.0 * v * mu * mu + 3.0 * v * v;
    mu = m;
    k.M1 = mu; k.M2 = mu * mu + v; k.M3 = mu * mu * mu + 3.0 * v * mu; k.M4 = mu * mu * mu * mu + 6.0 * v * mu * mu + 3.0 * v * v;
}

// second-kind multipliers (mu + x), (mu^2 + 2v + x^2 + mu x), (mu^3 + x^3 + x mu^2 + mu x^2 + 3 v x + 5 v mu)
#define MATERN_B(mu, x, v, b2, b3, b4)                  \
    const double b2 = (mu) + (x);                       \
    const double b3 = (mu) * (mu) + 2.0 * (v) + (x) * (x) + (mu) * (x); \
    const double b4 = (mu) * (mu) * (mu) + (x) * (x) * (x) + (x) * (mu) * (mu) + (mu) * (x) * (x) + 3.0 * (v) * (x) + 5.0 * (v) * (mu)

// S-role (the point is the SMALLER one): out[0..11], f2
static __device__ __forceinline__ void matern_role_S(double x, const MaternDimConst &k, double *out, double &f2) {
    const double l = k.l, l2 = l * l, l3 = l2 * l, l4 = l2 * l2, v = k.v, dx = x - k.m;
    const double hA = exp(k.c5 + k.q5 * dx), hB = exp(k.c5 - k.q5 * dx), eP = exp(k.q5 * dx);
    f2 = erf(dx * k.is2);
    const double g2 = exp(-0.5 * dx * dx / v);
    const double dd = x - k.muD;
    const double f3 = 1.0 + erf(dd * k.is2), g3 = exp(-0.5 * dd * dd / v);
    // coefficients of hi^a in 9 l^4 E5j (point = lo)
    const double s50[3] = {9.0 * l4 + x * (9.0 * SQRT5 * l3 + 15.0 * l2 * x), 9.0 * SQRT5 * l3 + x * (45.0 * l2 + 15.0 * SQRT5 * l * x), 15.0 * l2 + x * (15.0 * SQRT5 * l + 25.0 * x)};
    const double s51[3] = {18.0 * SQRT5 * l3 + x * (75.0 * l2 + 15.0 * SQRT5 * l * x), 75.0 * l2 + x * (60.0 * SQRT5 * l + 50.0 * x), 15.0 * SQRT5 * l + 50.0 * x};
    const double s52[3] = {75.0 * l2 + x * (45.0 * SQRT5 * l + 25.0 * x), 45.0 * SQRT5 * l + 100.0 * x, 25.0};
    const double s53[3] = {30.0 * SQRT5 * l + 50.0 * x, 50.0, 0.0};
    const double s54[3] = {25.0, 0.0, 0.0};
    const double s41[3] = {x * (15.0 * l2 - 15.0 * SQRT5 * l * x), 15.0 * l2 - 50.0 * x * x, 15.0 * SQRT5 * l - 50.0 * x};
    const double s42[3] = {-15.0 * l2 + x * (15.0 * SQRT5 * l + 25.0 * x), -15.0 * SQRT5 * l + 100.0 * x, 25.0};
    const double s43[3] = {-50.0 * x, -50.0, 0.0};
    const double s44[3] = {25.0, 0.0, 0.0};
    MATERN_B(k.muD, x, v, e2, e3, e4);
    MATERN_B(k.m, x, v, b2, b3, b4);
    double xa = 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double U5 = (s50[a] - k.D1 * s51[a] + k.D2 * s52[a] - k.D3 * s53[a] + k.D4 * s54[a]) * k.inv9l4;
        const double V5 = (s51[a] - e2 * s52[a] + e3 * s53[a] - e4 * s54[a]) * k.inv9l4;
        const double V42 = (s41[a] + b2 * s42[a] + b3 * s43[a] + b4 * s44[a]) * k.inv9l4;
        out[a] = hA * xa;
        out[3 + a] = hB * (0.5 * U5 * f3 + k.sv * V5 * g3);
        out[6 + a] = eP * xa;
        out[9 + a] = eP * (k.sv * V42 * g2);
        xa *= x;
    }
}

// T-role (the point is the LARGER one): out[0..14]
static __device__ __forceinline__ void matern_role_T(double x, const MaternDimConst &k, double *out) {
    const double l = k.l, l2 = l * l, l3 = l2 * l, l4 = l2 * l2, v = k.v, dx = x - k.m;
    const double hA = exp(k.c5 + k.q5 * dx), hB = exp(k.c5 - k.q5 * dx), eM = exp(-k.q5 * dx);
    const double g2 = exp(-0.5 * dx * dx / v);
    const double dc = k.muC - x;
    const double f1 = 1.0 + erf(dc * k.is2), g1 = exp(-0.5 * dc * dc / v);
    // coefficients of lo^a in 9 l^4 E3j / E4j (point = hi)
    const double l30[3] = {9.0 * l4 + x * (-9.0 * SQRT5 * l3 + 15.0 * l2 * x), -9.0 * SQRT5 * l3 + x * (45.0 * l2 - 15.0 * SQRT5 * l * x), 15.0 * l2 + x * (-15.0 * SQRT5 * l + 25.0 * x)};
    const double l31[3] = {18.0 * SQRT5 * l3 + x * (-75.0 * l2 + 15.0 * SQRT5 * l * x), -75.0 * l2 + x * (60.0 * SQRT5 * l - 50.0 * x), 15.0 * SQRT5 * l - 50.0 * x};
    const double l32[3] = {75.0 * l2 + x * (-45.0 * SQRT5 * l + 25.0 * x), -45.0 * SQRT5 * l + 100.0 * x, 25.0};
    const double l33[3] = {30.0 * SQRT5 * l - 50.0 * x, -50.0, 0.0};
    const double l34[3] = {25.0, 0.0, 0.0};
    const double l40[3] = {9.0 * l4 + x * (9.0 * SQRT5 * l3 + 15.0 * l2 * x), -9.0 * SQRT5 * l3 + x * (-45.0 * l2 - 15.0 * SQRT5 * l * x), 15.0 * l2 + x * (15.0 * SQRT5 * l + 25.0 * x)};
    const double l41[3] = {x * (15.0 * l2 + 15.0 * SQRT5 * l * x), 15.0 * l2 - 50.0 * x * x, -15.0 * SQRT5 * l - 50.0 * x};
    const double l42[3] = {-15.0 * l2 + x * (-15.0 * SQRT5 * l + 25.0 * x), 15.0 * SQRT5 * l + 100.0 * x, 25.0};
    const double l43[3] = {-50.0 * x, -50.0, 0.0};
    const double l44[3] = {25.0, 0.0, 0.0};
    MATERN_B(k.muC, x, v, c2, c3, c4);
    MATERN_B(k.m, x, v, b2, b3, b4);
    double xa = 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double U = (l30[a] + k.C1 * l31[a] + k.C2 * l32[a] + k.C3 * l33[a] + k.C4 * l34[a]) * k.inv9l4;
        const double V = (l31[a] + c2 * l32[a] + c3 * l33[a] + c4 * l34[a]) * k.inv9l4;
        const double U4 = (l40[a] + k.M1 * l41[a] + k.M2 * l42[a] + k.M3 * l43[a] + k.M4 * l44[a]) * k.inv9l4;
        const double V43 = (l41[a] + b2 * l42[a] + b3 * l43[a] + b4 * l44[a]) * k.inv9l4;
        out[a] = hA * (0.5 * U * f1 + k.sv * V * g1);
        out[3 + a] = hB * xa;
        out[6 + a] = eM * (-k.sv * V43 * g2);
        out[9 + a] = eM * xa;
        out[12 + a] = eM * (0.5 * U4);
        xa *= x;
    }
}
