"""CPU model of the one-launch factorisation's task table (dgp_amd/csrc/chol.hip, build_mega_tasks + mega_chain):
the same generator restated in Python, executed on small tiles by ONE in-order worker beside the chain (the strictest
schedule: if it never blocks, the table is a topological order), versions checked as the kernel checks them, results
against numpy.  Run: python tools/sim_mega_schedule.py"""
import numpy as np

LAZY = 4
FINAL = 1 << 30
STORE, SOLVE, TDIAG = 0, 1, 3
A_, T_, S_ = 0, 1, 2


def build(nbk, inv):
    tasks, need = [], [(0, 0)] * nbk
    apA = np.zeros((nbk, nbk), int); viA = np.zeros((nbk, nbk), int)
    apT = np.zeros((nbk, nbk), int); viT = np.zeros((nbk, nbk), int); viS = np.zeros((nbk, nbk), int)
    if inv:
        for q in range(nbk):
            apT[q, :] = q

    def emit(post, first, plus, bufC, ci, cj, bufL, li, bufR, ri, kb0, nkb, need_c, fin, wk):
        tasks.append(dict(post=post, first=first, plus=plus, bufC=bufC, ci=ci, cj=cj, bufL=bufL, li=li, bufR=bufR, ri=ri,
                          kb0=kb0, nkb=nkb, need_c=need_c, fin=fin, wk=wk))

    def updA(i, j, upto):
        if upto <= apA[i, j]:
            return
        emit(STORE, 0, 0, A_, i, j, A_, i, A_, j, apA[i, j], upto - apA[i, j], viA[i, j], 0, 0)
        viA[i, j] += 1; apA[i, j] = upto

    def updT(q, j, upto):
        if upto <= apT[q, j]:
            return
        emit(STORE, int(apT[q, j] == q), 0, T_, q, j, T_, q, A_, j, apT[q, j], upto - apT[q, j], viT[q, j], 0, 0)
        viT[q, j] += 1; apT[q, j] = upto

    def solveA(i, k):
        ap = apA[i, k]
        emit(SOLVE, 0, 0, A_, i, k, A_, i, A_, k, ap, k - ap, viA[i, k], 1, k); viA[i, k] += 1; apA[i, k] = k

    nl = nbk + (2 if inv else 0)
    for k in range(nl):
        if k < nbk:
            if k + 2 < nbk:   # look-ahead: the chain's inputs for block k+1
                solveA(k + 2, k)
                updA(k + 2, k + 2, k + 1)
                updA(k + 2, k + 1, k + 1)
                need[k + 1] = (viA[k + 2, k + 1], viA[k + 2, k + 2])
            if inv:
                emit(TDIAG, 1, 0, T_, k, k, A_, 0, A_, 0, 0, 0, viT[k, k], 1, k); viT[k, k] += 1
            for i in range(k + 3, nbk):
                solveA(i, k)
                if i == k + 3:   # catch-up of the next look-ahead's tiles
                    updA(k + 3, k + 3, k + 1)
                    updA(k + 3, k + 2, k + 1)
            if inv:
                for q in range(k - 1, -1, -1):
                    ap = apT[q, k]
                    emit(SOLVE, int(ap == q), 0, T_, q, k, T_, q, A_, k, ap, k - ap, viT[q, k], 1, k); viT[q, k] += 1; apT[q, k] = k
            for j in range(k + 2, nbk, LAZY):
                updA(j, j, k)
            for j in range(k + 1, nbk, LAZY):
                for i in range(j + 1, nbk):
                    updA(i, j, k)
            if inv:
                for j in range(k + 1, nbk, LAZY):
                    for q in range(0, k):
                        updT(q, j, k)
        if inv:
            q = k - 2
            while q >= 0:
                nkb = (k if k - 1 < nbk else nbk) - (k - 2)
                if nkb > 0:
                    for q2 in range(q + 1):
                        emit(STORE, int(q == k - 2), 1, S_, q, q2, T_, q, T_, q2, k - 2, nkb, viS[q, q2], 0, 0); viS[q, q2] += 1
                q -= 2
    return tasks, need


def run(nbk, inv, b=3, seed=0):
    rng = np.random.default_rng(seed)
    n = nbk * b
    M = rng.normal(size=(n, n)); K = M @ M.T + n * np.eye(n)
    buf = [np.tril(K).copy(), np.zeros((n, n)), np.zeros((n, n))]
    buf[0][np.triu_indices(n, 1)] = np.nan          # the upper triangle of A is never read
    for j in range(nbk):                           # ... except inside diagonal tiles (stored whole)
        sl = slice(j * b, (j + 1) * b); buf[0][sl, sl] = K[sl, sl]
    ver = np.zeros((3, nbk, nbk), int)
    W = [None] * nbk
    wflag = 0
    tasks, need = build(nbk, inv)
    tile = lambda B, i, j: buf[B][i * b:(i + 1) * b, j * b:(j + 1) * b]

    def ready(t):
        if ver[t['bufC'], t['ci'], t['cj']] < t['need_c']:
            return False
        for kb in range(t['kb0'], t['kb0'] + t['nkb']):
            if ver[t['bufL'], t['li'], kb] < FINAL or ver[t['bufR'], t['ri'], kb] < FINAL:
                return False
        if t['post'] in (SOLVE, TDIAG) and wflag < t['wk'] + 1:
            return False
        return True

    def execute(t):
        assert ver[t['bufC'], t['ci'], t['cj']] == t['need_c'], ('visit order', t)
        C = tile(t['bufC'], t['ci'], t['cj'])
        if t['post'] == TDIAG:
            C[:] = W[t['wk']].T
        else:
            acc = np.zeros((b, b)) if t['first'] else C.copy()
            for kb in range(t['kb0'], t['kb0'] + t['nkb']):
                acc += (1.0 if t['plus'] else -1.0) * tile(t['bufL'], t['li'], kb) @ tile(t['bufR'], t['ri'], kb).T
            if t['post'] == SOLVE:
                acc = acc @ W[t['wk']].T
            C[:] = acc
        assert not np.isnan(C).any(), t
        ver[t['bufC'], t['ci'], t['cj']] = FINAL if t['fin'] else t['need_c'] + 1

    # chain state machine: phase 0 = factor block k (needs nothing), phase 1 = waits for the workers' tiles
    k, phase, D = 0, 0, tile(0, 0, 0).copy()
    qi = 0
    while True:
        progressed = False
        if k < nbk:
            if phase == 0:
                L = np.linalg.cholesky(D); W[k] = np.linalg.inv(L); wflag = k + 1
                tile(0, k, k)[:] = L
                phase = 1; progressed = True
                if k + 1 == nbk:
                    k = nbk
            elif ver[0, k + 1, k] >= need[k][0] and ver[0, k + 1, k + 1] >= need[k][1]:
                assert ver[0, k + 1, k] == need[k][0] and ver[0, k + 1, k + 1] == need[k][1]
                P = tile(0, k + 1, k) @ W[k].T
                tile(0, k + 1, k)[:] = P; ver[0, k + 1, k] = FINAL
                D = tile(0, k + 1, k + 1) - P @ P.T
                k += 1; phase = 0; progressed = True
        if qi < len(tasks) and ready(tasks[qi]):
            execute(tasks[qi]); qi += 1; progressed = True
        if not progressed:
            break
    assert k >= nbk and qi == len(tasks), 'deadlock: chain at block %d phase %d, queue at %d of %d: %s' % (k, phase, qi, len(tasks), tasks[qi] if qi < len(tasks) else None)
    Lref = np.linalg.cholesky(K)
    eL = np.abs(np.tril(buf[0]) - Lref).max()
    msg = 'nbk=%d inv=%d tasks=%d  |L-Lref| %.1e' % (nbk, inv, len(tasks), eL)
    assert eL < 1e-10, msg
    if inv:
        Tref = np.linalg.inv(Lref).T
        eT = np.abs(np.triu(buf[1]) - Tref).max()
        Sref = np.linalg.inv(K)
        eS = np.abs(np.tril(buf[2]) - np.tril(Sref)).max() / np.abs(Sref).max()
        msg += '  |T-L^-T| %.1e  rel|S-K^-1| %.1e' % (eT, eS)
        assert eT < 1e-10 and eS < 1e-10, msg
    print(msg)


if __name__ == '__main__':
    for nbk in (1, 2, 3, 4, 5, 6, 7, 9, 12, 17, 32):
        for inv in (0, 1):
            run(nbk, inv)
    print('OK')
