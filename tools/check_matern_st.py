"""Numerical check (numpy) of the separable S/T form of the Matern-2.5 J factor against the oracle's Jd."""
import sys
import numpy as np
from scipy.special import erf
sys.path.insert(0, '/root/repo')
from oracle import dgp_oracle as O
SQ5 = np.sqrt(5.0)


def coefL(x, l, which):   # x larger point; coefficient of x1^a
    l2, l3 = l * l, l ** 3
    t = {
        '30': [9 * l**4 + x * (-9 * SQ5 * l3 + 15 * l2 * x), -9 * SQ5 * l3 + x * (45 * l2 - 15 * SQ5 * l * x), 15 * l2 + x * (-15 * SQ5 * l + 25 * x)],
        '31': [18 * SQ5 * l3 + x * (-75 * l2 + 15 * SQ5 * l * x), -75 * l2 + x * (60 * SQ5 * l - 50 * x), 15 * SQ5 * l - 50 * x],
        '32': [75 * l2 + x * (-45 * SQ5 * l + 25 * x), -45 * SQ5 * l + 100 * x, 25 + 0 * x],
        '33': [30 * SQ5 * l - 50 * x, -50 + 0 * x, 0 * x],
        '34': [25 + 0 * x, 0 * x, 0 * x],
        '40': [9 * l**4 + x * (9 * SQ5 * l3 + 15 * l2 * x), -9 * SQ5 * l3 + x * (-45 * l2 - 15 * SQ5 * l * x), 15 * l2 + x * (15 * SQ5 * l + 25 * x)],
        '41': [x * (15 * l2 + 15 * SQ5 * l * x), 15 * l2 - 50 * x * x, -15 * SQ5 * l - 50 * x],
        '42': [-15 * l2 + x * (-15 * SQ5 * l + 25 * x), 15 * SQ5 * l + 100 * x, 25 + 0 * x],
        '43': [-50 * x, -50 + 0 * x, 0 * x],
        '44': [25 + 0 * x, 0 * x, 0 * x]}
    return t[which]


def coefS(x, l, which):   # x smaller point; coefficient of x2^a
    l2, l3 = l * l, l ** 3
    t = {
        '50': [9 * l**4 + x * (9 * SQ5 * l3 + 15 * l2 * x), 9 * SQ5 * l3 + x * (45 * l2 + 15 * SQ5 * l * x), 15 * l2 + x * (15 * SQ5 * l + 25 * x)],
        '51': [18 * SQ5 * l3 + x * (75 * l2 + 15 * SQ5 * l * x), 75 * l2 + x * (60 * SQ5 * l + 50 * x), 15 * SQ5 * l + 50 * x],
        '52': [75 * l2 + x * (45 * SQ5 * l + 25 * x), 45 * SQ5 * l + 100 * x, 25 + 0 * x],
        '53': [30 * SQ5 * l + 50 * x, 50 + 0 * x, 0 * x],
        '54': [25 + 0 * x, 0 * x, 0 * x],
        '40': [9 * l**4 + x * (-9 * SQ5 * l3 + 15 * l2 * x), 9 * SQ5 * l3 + x * (-45 * l2 + 15 * SQ5 * l * x), 15 * l2 + x * (-15 * SQ5 * l + 25 * x)],
        '41': [x * (15 * l2 - 15 * SQ5 * l * x), 15 * l2 - 50 * x * x, 15 * SQ5 * l - 50 * x],
        '42': [-15 * l2 + x * (15 * SQ5 * l + 25 * x), -15 * SQ5 * l + 100 * x, 25 + 0 * x],
        '43': [-50 * x, -50 + 0 * x, 0 * x],
        '44': [25 + 0 * x, 0 * x, 0 * x]}
    return t[which]


def ST(x, m, v, l):
    inv = 1.0 / (9 * l ** 4)
    sv, s2 = np.sqrt(0.5 * v / np.pi), np.sqrt(2 * v)
    muC, muD = m - 2 * SQ5 * v / l, m + 2 * SQ5 * v / l
    hA, hB = np.exp((5 * v + SQ5 * l * (x - m)) / l**2), np.exp((5 * v - SQ5 * l * (x - m)) / l**2)
    eP, eM = np.exp(SQ5 * (x - m) / l), np.exp(-SQ5 * (x - m) / l)
    f1, g1 = 1 + erf((muC - x) / s2), np.exp(-0.5 * (x - muC)**2 / v)
    f2, g2 = erf((x - m) / s2), np.exp(-0.5 * (x - m)**2 / v)
    f3, g3 = 1 + erf((x - muD) / s2), np.exp(-0.5 * (x - muD)**2 / v)

    def mom(mu):   # moment multipliers shared by the A*1 and A*2 combinations
        return (mu, mu**2 + v, mu**3 + 3 * v * mu, mu**4 + 6 * v * mu**2 + 3 * v**2,
                mu + x, mu**2 + 2 * v + x**2 + mu * x, mu**3 + x**3 + x * mu**2 + mu * x**2 + 3 * v * x + 5 * v * mu)
    T, S, Tx = [], [], []
    c1, c2, c3, c4, b2, b3, b4 = mom(muC)
    L = {k: coefL(x, l, k) for k in ('30', '31', '32', '33', '34', '40', '41', '42', '43', '44')}
    for a in range(3):
        U = (L['30'][a] + c1 * L['31'][a] + c2 * L['32'][a] + c3 * L['33'][a] + c4 * L['34'][a]) * inv
        V = (L['31'][a] + b2 * L['32'][a] + b3 * L['33'][a] + b4 * L['34'][a]) * inv
        T.append(hA * (0.5 * U * f1 + sv * V * g1))
    for a in range(3):
        T.append(hB * x**a)
    c1, c2, c3, c4, b2, b3, b4 = mom(m)
    for a in range(3):
        U4 = (L['40'][a] + c1 * L['41'][a] + c2 * L['42'][a] + c3 * L['43'][a] + c4 * L['44'][a]) * inv
        V43 = (L['41'][a] + b2 * L['42'][a] + b3 * L['43'][a] + b4 * L['44'][a]) * inv
        T.append(eM * (-sv * V43 * g2))
        Tx.append(eM * 0.5 * U4)
    for a in range(3):
        T.append(eM * x**a)
    Sc = {k: coefS(x, l, k) for k in ('50', '51', '52', '53', '54', '40', '41', '42', '43', '44')}
    for a in range(3):
        S.append(hA * x**a)
    d1, d2, d3, d4, e2, e3, e4 = mom(muD)
    for a in range(3):
        U5 = (Sc['50'][a] - d1 * Sc['51'][a] + d2 * Sc['52'][a] - d3 * Sc['53'][a] + d4 * Sc['54'][a]) * inv
        V5 = (Sc['51'][a] - e2 * Sc['52'][a] + e3 * Sc['53'][a] - e4 * Sc['54'][a]) * inv
        S.append(hB * (0.5 * U5 * f3 + sv * V5 * g3))
    for a in range(3):
        S.append(eP * x**a)
    for a in range(3):
        U4p = (Sc['40'][a] + c1 * Sc['41'][a] + c2 * Sc['42'][a] + c3 * Sc['43'][a] + c4 * Sc['44'][a]) * inv
        V42 = (Sc['41'][a] + b2 * Sc['42'][a] + b3 * Sc['43'][a] + b4 * Sc['44'][a]) * inv
        S.append(eP * (sv * V42 * g2))
    return np.array(S), np.array(T + Tx), f2


rng = np.random.default_rng(1)
worst = 0
for trial in range(2000):
    m, v, l = rng.uniform(-0.5, 1.5), 10 ** rng.uniform(-4, 0.5), rng.uniform(0.2, 3.0)
    xa, xb = rng.uniform(-1, 2, size=2)
    if trial % 10 == 0:
        xb = xa
    lo, hi = min(xa, xb), max(xa, xb)
    S, _, f2lo = ST(lo, m, v, l)
    _, T, f2hi = ST(hi, m, v, l)
    got = S @ T[:12] + (f2hi - f2lo) * (S[6:9] @ T[12:15])
    with np.errstate(all='ignore'):
        ref = O.Jd(xa, xb, m, v, l)
    if not np.isfinite(ref):
        continue
    err = abs(got - ref) / max(abs(ref), 1e-300)
    worst = max(worst, err)
    if err > 1e-8:
        print('BAD', m, v, l, xa, xb, got, ref, err)
print('worst relative error of <S,T> vs oracle Jd over 2000 random cases: %.3e' % worst)
