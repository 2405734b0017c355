#!/usr/bin/env python3
"""Host-side model of the factorisation's task schedule (csrc/chol.hip build_tasks): compares the fixed lazy rule with
a budgeted earliest-deadline-first rule under a simple cost model measured on MI355X
(tools/gpu_step_wgtrace.py): a bulk task with p panels takes 2 + 4.75 p us, a solve ~9 us after the chain (24.5 us),
512 - 2 B workgroup slots."""
import sys, heapq
import numpy as np

CHAIN, SOLVE = 24.0, 4.0
RNG = np.random.default_rng(0)


def dur(p):
    return 2.0 + 4.75 * p


def launch_time(bulk_panels, nsolve, B, slots=512):
    """list scheduling: bulk tasks (durations by panels) in order on the slots, solves become runnable at CHAIN."""
    s = slots - 2 * B
    tasks = []
    for p in bulk_panels:
        tasks += list(dur(p) * RNG.uniform(0.8, 1.25, B))
    free = [0.0] * s
    heapq.heapify(free)
    end = 0.0
    for d in tasks:
        t = heapq.heappop(free)
        heapq.heappush(free, t + d)
        end = max(end, t + d)
    for _ in range(nsolve * B):
        t = max(heapq.heappop(free), CHAIN)
        heapq.heappush(free, t + SOLVE)
        end = max(end, t + SOLVE)
    return max(end, CHAIN)


def fixed_lazy(nbk, inv, LAZY=4):
    """the current rule: returns per launch (list of bulk panel counts, number of solve tasks)"""
    out = []
    for k in range(nbk + (2 if inv else 0)):
        bulk, ns = [], 0
        if k < nbk:
            ns = nbk - 1 - k + (k if inv else 0)
            kb0 = max(k - LAZY, 0)
            nkb = k - kb0
            if nkb > 0:
                for j in range(k + 1, nbk, LAZY):
                    bulk += [nkb] * (nbk - j)
                    if inv:
                        for q in range(k):
                            f0 = max(q, kb0)
                            bulk.append(k - f0)
        if inv:
            for q in range(k - 2, -1, -2):
                nkb = min(k, nbk) - (k - 2)
                if nkb > 0:
                    bulk += [nkb] * (q + 1)
        out.append((bulk, ns))
    return out


def edf(nbk, inv, B, budget_us, CAP=4, slots=512, ntask=None):
    """budgeted EDF: per launch, mandatory tile visits (deadline = the launch before the column is factored, at most
    CAP panels per visit), then optional full groups in deadline order while the launch's budget (slot-time) lasts;
    S tiles (no deadline) fill what is left, largest backlog first."""
    doneA = np.zeros((nbk, nbk), int)     # panels applied to A[i][j] (i >= j)
    doneT = np.zeros((nbk, nbk), int)     # panels q.. applied to T[q][j]: absolute index of the next panel
    for q in range(nbk):
        doneT[q, :] = q
    doneS = np.zeros((nbk, nbk), int)
    for q in range(nbk):
        doneS[q, :] = q                    # S[q][q2] takes panels p >= q
    out = []
    nl = nbk + (2 if inv else 0)
    k = 0
    while True:
        bulk, ns = [], 0
        if k < nbk:
            ns = nbk - 1 - k + (k if inv else 0)
        room = budget_us * (slots - 2 * B) / B if ntask is None else ntask * dur(CAP) * ((slots - 2 * B) // B)
        avail = min(k, nbk)                        # panels 0..avail-1 exist
        opt = []
        for j in range(k + 1, nbk):
            need_after = (j - 1) - CAP * (j - 1 - k)
            for i in range(j, nbk):
                d = doneA[i, j]
                a = min(avail, j - 1) - d
                if a <= 0:
                    continue
                if d < need_after or j == k + 1:
                    p = min(CAP, a); bulk.append(p); doneA[i, j] += p; room -= dur(p)
                elif a >= CAP:
                    opt.append((j, 0, i))
            if inv:
                for q in range(0, min(k, j)):
                    d = doneT[q, j]
                    a = min(avail, j - 1) - d
                    if a <= 0:
                        continue
                    if d < need_after or j == k + 1:
                        p = min(CAP, a); bulk.append(p); doneT[q, j] += p; room -= dur(p)
                    elif a >= CAP:
                        opt.append((j, 1, q))
        for j, kind, r in opt:     # already in deadline order
            if room < dur(CAP):
                break
            if kind == 0:
                doneA[r, j] += CAP
            else:
                doneT[r, j] += CAP
            bulk.append(CAP); room -= dur(CAP)
        if inv:
            cand = []
            last = k >= nbk
            for q in range(nbk):
                for q2 in range(q + 1):
                    a = avail - doneS[q, q2]
                    if a > 0:
                        cand.append((-a, q, q2))
            cand.sort()
            for na, q, q2 in cand:
                a = -na
                if not last and (a < CAP or room < dur(CAP)):
                    continue
                p = min(CAP, a)
                doneS[q, q2] += p; bulk.append(p); room -= dur(p)
        out.append((bulk, ns))
        k += 1
        if k >= nbk:
            if not inv:
                break
            if all(doneS[q, q2] >= nbk for q in range(nbk) for q2 in range(q + 1)):
                break
    return out


def total(sched, B):
    ts = [launch_time(b, ns, B) for b, ns in sched]
    return sum(ts), ts


if __name__ == '__main__':
    nbk = 32
    for inv in (False, True):
        for B in ((1, 2, 3, 4, 6) if inv else (4, 6, 8, 12)):
            t0, ts0 = total(fixed_lazy(nbk, inv), B)
            print('inv=%d B=%2d  fixed lazy %.0f us' % (inv, B, t0))
            for CAP in (4, 5, 6, 8):
                for nt in (1.0, 1.5, 2.0):
                    t1, ts1 = total(edf(nbk, inv, B, 0, CAP=CAP, ntask=nt), B)
                    print('     EDF cap %d panels, %.1f rounds: %.0f us' % (CAP, nt, t1), ' '.join('%d' % t for t in ts1) if '-v' in sys.argv else '')
