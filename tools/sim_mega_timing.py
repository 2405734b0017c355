#!/usr/bin/env python3
"""Timing model of the one-launch factorisation (csrc/chol.hip potrf_mega_kernel): the task-table generator restated in Python
(build(): the same statements as build_mega_tasks, checked against a table dumped from the library) and a discrete-event simulation
of the launch -- chains, the two ticket queues per group of matrices, workers that pull in order and wait inside their tasks, the
split wait, version words with a visibility latency -- with task durations taken from the task logs (profiles/r05_tasklog_*.txt).
Purpose: try table orders in seconds instead of GPU runs, and say what a change can buy before it is built.
usage: sim_mega_timing.py [B ...]            (prints model time for potrf / potrf_inv at those batch sizes, current table)
       sim_mega_timing.py check file.npz     (compare build() with the table in a task log)"""
import heapq
import sys
import numpy as np

FINAL = 1 << 30
STORE, SOLVE, TDIAG, LOOK, LOOKD = 0, 1, 3, 4, 5
A_, T_, S_ = 0, 1, 2


def build(nbk, inv, lazy=10, slazy=10, near=3, lag=0, xcatch=0, stail=0, look=1, slag=0, far_merge=0, critd=0):
    """-> (tasks, need): tasks = dicts(post, first, bufC, ci, cj, bufL, li, bufR, ri, kb0, nkb, need_c, fin, wk, need2, step, crit)"""
    tasks, need = [], [(0, 0)] * nbk
    apA = np.zeros((nbk, nbk), int); viA = np.zeros((nbk, nbk), int)
    apT = np.zeros((nbk, nbk), int); viT = np.zeros((nbk, nbk), int)
    viS = np.zeros((nbk, nbk), int); apS = np.arange(nbk)
    if inv:
        for q in range(nbk):
            apT[q, :] = q
    cur = dict(step=0, crit=0)

    def emit(post, first, bufC, ci, cj, bufL, li, bufR, ri, kb0, nkb, need_c, fin, wk, need2=0, crit=None):
        tasks.append(dict(post=post, first=first, bufC=bufC, ci=ci, cj=cj, bufL=bufL, li=li, bufR=bufR, ri=ri, kb0=kb0, nkb=nkb,
                          need_c=need_c, fin=fin, wk=wk, need2=need2, step=cur['step'], crit=cur['crit'] if crit is None else crit))

    def updA(i, j, upto):
        if upto <= apA[i, j]:
            return
        emit(STORE, 0, A_, i, j, A_, i, A_, j, apA[i, j], upto - apA[i, j], viA[i, j], 0, 0)
        viA[i, j] += 1; apA[i, j] = upto

    def updT(q, j, upto):
        if upto <= apT[q, j]:
            return
        emit(STORE, int(apT[q, j] == q), T_, q, j, T_, q, A_, j, apT[q, j], upto - apT[q, j], viT[q, j], 0, 0)
        viT[q, j] += 1; apT[q, j] = upto

    def solveA(i, k):
        ap = apA[i, k]
        emit(SOLVE, 0, A_, i, k, A_, i, A_, k, ap, k - ap, viA[i, k], 1, k)
        viA[i, k] += 1; apA[i, k] = k

    def lookA(k):
        i = k + 2
        if not look or apA[i, k + 1] != k or apA[i, i] != k:
            return False
        ap = apA[i, k]
        emit(LOOK, 0, A_, i, k, A_, i, A_, k, ap, k - ap, viA[i, k], 0, k, viA[i, k + 1], crit=1)
        emit(LOOKD, 0, A_, i, k, A_, i, A_, k, ap, k - ap, viA[i, k], 1, k, viA[i, i], crit=1)
        viA[i, k] += 1; viA[i, k + 1] += 1; viA[i, i] += 1
        apA[i, k] = k; apA[i, k + 1] = apA[i, i] = k + 1
        return True

    nl = nbk + ((1 + lag + slag) if inv else 0)
    for k in range(nl):
        cur['step'] = k
        if k < nbk:
            if k + 2 < nbk:
                cur['crit'] = 1
                if not lookA(k):
                    solveA(k + 2, k); updA(k + 2, k + 2, k + 1); updA(k + 2, k + 1, k + 1)
                cur['crit'] = 0
                need[k + 1] = (int(viA[k + 2, k + 1]), int(viA[k + 2, k + 2]))
            if inv:
                emit(TDIAG, 1, T_, k, k, A_, 0, A_, 0, 0, 0, viT[k, k], 1, k); viT[k, k] += 1
            for i in range(k + 3, nbk):
                cur['crit'] = int(i <= k + 3 + critd)
                solveA(i, k)
                cur['crit'] = 0
                if i == k + 3:
                    cur['crit'] = 1
                    updA(k + 3, k + 3, k + 1); updA(k + 3, k + 2, k + 1)
                    if xcatch:
                        updA(k + 3, k + 1, k + 1)
                    elif look:
                        updA(k + 3, k + 1, k)
                    cur['crit'] = 0
            if inv:
                for q in range(k - 1, -1, -1):
                    ap = apT[q, k]
                    emit(SOLVE, int(ap == q), T_, q, k, T_, q, A_, k, ap, k - ap, viT[q, k], 1, k)
                    viT[q, k] += 1; apT[q, k] = k
            kl = k - lag
            if kl > 0:
                # far_merge: a column's visits further than `lazy` block steps from its solves are not made (their panels wait for the
                # one deep visit `lazy` steps out): defers the early, shallow, far-away work
                cols_d = [j for j in range(k + 2 + near, nbk, lazy)]
                cols = [j for j in range(k + 1 + near, nbk, lazy)]
                if far_merge:
                    cols_d, cols = cols_d[:far_merge], cols[:far_merge]
                for j in cols_d:
                    updA(j, j, kl)
                for j in cols:
                    for i in range(j + 1, nbk):
                        updA(i, j, kl)
                if inv:
                    for j in cols:
                        for q in range(0, kl):
                            updT(q, j, kl)
        if inv:
            ks = k - lag - slag
            left = nbk - ks
            thr = 1 if ks >= nbk else ((left if left > 1 else 1) if (stail and left < slazy) else slazy)
            for q in range(0, min(ks, nbk)):
                upto = ks if ks < nbk else nbk
                if upto - apS[q] < thr:
                    continue
                ap, nkb = int(apS[q]), int(upto - apS[q])
                for q2 in range(q + 1):
                    emit(STORE, int(ap == q), S_, q, q2, T_, q, T_, q2, ap, nkb, viS[q, q2], 0, 0); viS[q, q2] += 1
                apS[q] = upto
    return tasks, need


class Sim:
    """One launch.  Model constants (us) from the task logs: a panel 2.8 alone .. 4.8 with every worker computing, the chain's factor 7.5 .. 8.7,
    stores 1.5 .. 2.5, a publication visible 1.0 .. 2.5 later."""

    def __init__(self, tasks, need, B, nbk, inv, queues=2, ncrit=None, groups=None, split=True, window=0):
        self.tasks, self.need, self.B, self.nbk, self.inv = tasks, need, B, nbk, inv
        self.split = split
        self.window = window   # > 0: a worker takes the first task among the next `window` of its queue whose first inputs have arrived (an idealised ready list)
        self.taken = {}
        G = groups or (8 if B % 8 == 0 else 4 if B % 4 == 0 else 2 if B >= 2 else 1)
        while G > B:
            G //= 2
        self.G = G
        self.ncrit = ncrit if ncrit else (12 if B <= 4 else 8)
        crit = [t for t in tasks if t['crit']] if queues == 2 else []
        bulk = [t for t in tasks if not t['crit']] if queues == 2 else list(tasks)
        self.q = [crit, bulk]
        self.heads = [[0, 0] for _ in range(G)]
        self.P = 512 - 2 * B
        self.now = 0.0
        self.heap, self.seq = [], 0
        self.avail = {}       # (b, buf, i, j) -> list of (version, time)
        self.waiters = {}     # (b, buf, i, j) -> list of (version, callback)
        self.wavail = {}      # (b, k) -> time W_k is visible
        self.wwait = {}
        self.computing = 0
        self.busy_time = 0.0
        self.end = 0.0
        self.chain_end = [0.0] * B
        self.chain_steps = [[] for _ in range(B)]

    # ---- event plumbing ----
    def at(self, t, fn, *a):
        self.seq += 1
        heapq.heappush(self.heap, (t, self.seq, fn, a))

    def load(self):
        return min(1.0, self.computing / float(self.P))

    def publish(self, b, buf, i, j, ver, t):
        tv = t + 1.0 + 1.5 * self.load()
        key = (b, buf, i, j)
        self.avail.setdefault(key, []).append((ver, tv))
        ws = self.waiters.pop(key, [])
        keep = []
        for v, cb in ws:
            if ver >= v:
                cb(tv)
            else:
                keep.append((v, cb))
        if keep:
            self.waiters[key] = keep
        self.end = max(self.end, t)

    def when(self, b, buf, i, j, ver, cb):
        """cb(time) once tile (buf, i, j) of matrix b has a version >= ver"""
        if ver <= 0:
            cb(0.0)
            return
        for v, t in self.avail.get((b, buf, i, j), ()):
            if v >= ver:
                cb(t)
                return
        self.waiters.setdefault((b, buf, i, j), []).append((ver, cb))

    def when_w(self, b, k, cb):
        if (b, k) in self.wavail:
            cb(self.wavail[(b, k)])
        else:
            self.wwait.setdefault((b, k), []).append(cb)

    def publish_w(self, b, k, t):
        tv = t + 0.8 + 1.0 * self.load()
        self.wavail[(b, k)] = tv
        for cb in self.wwait.pop((b, k), []):
            cb(tv)

    def wait_all(self, items, t0, cb):
        """items: list of registration functions f(callback); cb(max(t0, all times)) when every one has fired"""
        if not items:
            cb(t0)
            return
        st = dict(n=len(items), t=t0)

        def one(t):
            st['t'] = max(st['t'], t)
            st['n'] -= 1
            if st['n'] == 0:
                cb(st['t'])
        for f in items:
            f(one)

    # ---- the chain of matrix b ----
    def chain_step(self, b, k, t):
        tf = t + 7.5 + 1.2 * self.load() + (1.5 if k == 0 else 0.0)
        if k + 1 == self.nbk:
            self.publish_w(b, k, tf + 0.6)
            self.chain_end[b] = tf
            self.end = max(self.end, tf)
            return
        nq, nd = self.need[k]
        self.publish_w(b, k, tf + 1.0)

        def got_q(tq):
            ts = max(tf + 1.2, tq) + 0.7

            def got_d(td):
                t1 = max(ts, td) + 1.2
                self.at(t1, lambda: self.publish(b, A_, k + 1, k, FINAL, t1 + 0.6))
                t2 = t1 + 1.2
                self.chain_steps[b].append(t2 - t)
                self.at(t2, self.chain_step, b, k + 1, t2)
            self.when(b, A_, k + 1, k + 1, nd, got_d)
        self.when(b, A_, k + 1, k, nq, got_q)

    # ---- a worker ----
    def pull(self, w, t):
        grp0 = w['grp']
        for step in range(self.G):
            grp = (grp0 + step) % self.G
            nbg = (self.B - grp + self.G - 1) // self.G
            order = (0, 1) if w['role'] == 0 else (1, 0)
            for qi in order:
                cap = len(self.q[qi]) * nbg
                h = self.heads[grp][qi]
                if self.window:
                    tk_set = self.taken.setdefault((grp, qi), set())
                    while h < cap and h in tk_set:
                        h += 1
                    self.heads[grp][qi] = h
                    if h < cap:
                        pick, seen = h, 0
                        for c in range(h, cap):
                            if c in tk_set:
                                continue
                            seen += 1
                            if seen > self.window:
                                break
                            if self.ready_now(self.q[qi][c // nbg], grp + self.G * (c % nbg), t):
                                pick = c
                                break
                        tk_set.add(pick)
                        w['grp'] = grp
                        self.run_task(w, self.q[qi][pick // nbg], grp + self.G * (pick % nbg), t)
                        return
                    continue
                if h < cap:
                    self.heads[grp][qi] += 1
                    slot, b = h // nbg, grp + self.G * (h % nbg)
                    w['grp'] = grp
                    self.run_task(w, self.q[qi][slot], b, t)
                    return
        # nothing left anywhere

    def has(self, b, buf, i, j, ver, t):
        if ver <= 0:
            return True
        return any(v >= ver and tv <= t for v, tv in self.avail.get((b, buf, i, j), ()))

    def ready_now(self, tk, b, t):
        if tk['post'] == TDIAG:
            return (b, tk['wk']) in self.wavail and self.wavail[(b, tk['wk'])] <= t
        nkb, kb0 = tk['nkb'], tk['kb0']
        n1 = nkb - 1 if (self.split and nkb >= 2) else nkb
        if not self.has(b, tk['bufC'], tk['ci'], tk['cj'], tk['need_c'], t):
            return False
        for kb in range(kb0, kb0 + n1):
            if not (self.has(b, tk['bufL'], tk['li'], kb, FINAL, t) and self.has(b, tk['bufR'], tk['ri'], kb, FINAL, t)):
                return False
        return True

    def run_task(self, w, tk, b, t):
        post, nkb, kb0 = tk['post'], tk['nkb'], tk['kb0']
        if post == TDIAG:
            def wdone(tw):
                t1 = max(t, tw) + 3.5 + 1.5 * self.load()
                self.at(t1, self.finish, w, tk, b, t1, [(tk['bufC'], tk['ci'], tk['cj'], FINAL)])
            self.when_w(b, tk['wk'], wdone)
            return
        n1 = nkb - 1 if (self.split and nkb >= 2) else nkb
        first = [lambda cb: self.when(b, tk['bufC'], tk['ci'], tk['cj'], tk['need_c'], cb)]
        for kb in range(kb0, kb0 + n1):
            first.append(lambda cb, kb=kb: self.when(b, tk['bufL'], tk['li'], kb, FINAL, cb))
            first.append(lambda cb, kb=kb: self.when(b, tk['bufR'], tk['ri'], kb, FINAL, cb))

        def phase1(tr):
            tr = max(tr, t) + 1.0
            self.computing += 1
            tp = 2.8 + 2.0 * self.load()
            d1 = 1.2 + n1 * tp
            t1 = tr + d1

            def after1():
                if n1 == nkb:
                    phase3(t1)
                    return
                kl = kb0 + nkb - 1
                self.wait_all([lambda cb: self.when(b, tk['bufL'], tk['li'], kl, FINAL, cb),
                               lambda cb: self.when(b, tk['bufR'], tk['ri'], kl, FINAL, cb)], t1, phase2)

            def phase2(tn):
                late = tn > t1 + 1e-9
                if late:
                    self.computing -= 1   # (waiting, not computing)
                    tn += 0.8

                    def go():
                        self.computing += 1
                        t2 = tn + 2.8 + 2.0 * self.load()
                        self.at(t2, phase3, t2)
                    self.at(tn, go)
                else:
                    t2 = t1 + 2.8 + 2.0 * self.load()
                    self.at(t2, phase3, t2)

            def phase3(t2):
                self.computing -= 1
                self.busy_time += 0  # (accounted at finish)
                if post == STORE:
                    ts = t2 + 1.5 + 1.0 * self.load()
                    newver = FINAL if tk['fin'] else tk['need_c'] + 1
                    self.at(ts, self.finish, w, tk, b, ts, [(tk['bufC'], tk['ci'], tk['cj'], newver)])
                    return

                def wdone(tw):
                    t3 = max(t2, tw + 0.5) + 2.0 + 0.8 * self.load()
                    if post == SOLVE:
                        ts = t3 + 1.5 + 1.0 * self.load()
                        self.at(ts, self.finish, w, tk, b, ts, [(tk['bufC'], tk['ci'], tk['cj'], FINAL)])
                    elif post == LOOK:
                        ci, cj = tk['ci'], tk['cj']
                        self.aux[(b, ci, cj)] = t2
                        for cb in self.auxwait.pop((b, ci, cj), []):
                            cb(t2)
                        self.wait_all([lambda cb: self.when(b, A_, ci - 1, cj, FINAL, cb),
                                       lambda cb: self.when(b, A_, ci, cj + 1, tk['need2'], cb)], t3,
                                      lambda t4: self.at(t4 + 0.8 + 3.0 + 1.0 * self.load() + 1.5,
                                                         self.finish, w, tk, b, t4 + 0.8 + 3.0 + 1.0 * self.load() + 1.5,
                                                         [(A_, ci, cj + 1, tk['need2'] + 1)]))
                    else:   # LOOKD
                        ci, cj = tk['ci'], tk['cj']

                        def auxw(cb):
                            if (b, ci, cj) in self.aux:
                                cb(self.aux[(b, ci, cj)])
                            else:
                                self.auxwait.setdefault((b, ci, cj), []).append(cb)

                        def after(t4):
                            t4 += 0.8
                            self.at(t4 + 1.2, lambda: self.publish(b, A_, ci, cj, FINAL, t4 + 1.2))
                            te = t4 + 1.2 + 3.0 + 1.0 * self.load() + 1.5
                            self.at(te, self.finish, w, tk, b, te, [(A_, ci, ci, tk['need2'] + 1)])
                        self.wait_all([lambda cb: self.when(b, A_, ci, ci, tk['need2'], cb), auxw], t3, after)
                self.when_w(b, tk['wk'], wdone)
            self.at(t1, after1)
        self.wait_all(first, t, lambda tr: self.at(max(tr, t), phase1, tr))

    def finish(self, w, tk, b, t, pubs):
        for buf, i, j, ver in pubs:
            self.publish(b, buf, i, j, ver, t)
        self.ntask += 1
        self.at(t + 1.0, self.pull, w, t + 1.0)

    def run(self):
        self.aux, self.auxwait, self.ntask = {}, {}, 0
        for b in range(self.B):
            self.at(0.0, self.chain_step, b, 0, 0.0)
        G = self.G
        for wi in range(self.P):
            xcd = wi % 8
            grp = xcd % G
            nbg = (self.B - grp + G - 1) // G
            cpx = (self.ncrit * nbg * G + 7) // 8 if self.q[0] else 0
            w = dict(id=wi, grp=grp, role=0 if (wi // 8) < cpx else 1)
            self.at(3.0 + 0.01 * wi, self.pull, w, 3.0 + 0.01 * wi)
        while self.heap:
            t, _, fn, a = heapq.heappop(self.heap)
            self.now = t
            fn(*a)
        total = sum(len(q) for q in self.q) * self.B
        assert self.ntask == total, ('deadlock or lost task: %d of %d tasks ran' % (self.ntask, total))
        return self.end


def model(B, inv, nbk=32, **kw):
    simkw = {k: kw.pop(k) for k in list(kw) if k in ('queues', 'ncrit', 'groups', 'split', 'window')}
    deep = B >= 6
    p = dict(lazy=(12 if inv else 10) if deep else (10 if inv else 8), slazy=(12 if inv else 10) if deep else (10 if inv else 8), near=3,
             stail=1 if (inv and B <= 3) else 0)
    p.update(kw)
    tasks, need = build(nbk, inv, **p)
    s = Sim(tasks, need, B, nbk, inv, **simkw)
    end = s.run()
    return end, max(s.chain_end), float(np.median(s.chain_steps[0])), len(tasks)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'check':
        z = np.load(sys.argv[2])
        tab = z['table']
        inv, nbk, B = int(z['inv']), int(z['nbk']), int(z['B'])
        deep = B >= 6
        tasks, need = build(nbk, bool(inv), lazy=(12 if inv else 10) if deep else (10 if inv else 8), slazy=(12 if inv else 10) if deep else (10 if inv else 8),
                            near=3, stail=1 if (inv and B <= 3) else 0)
        mine = [t for t in tasks if t['crit']] + [t for t in tasks if not t['crit']]
        assert len(mine) == len(tab), (len(mine), len(tab))
        bad = 0
        for t, r in zip(mine, tab):
            got = (r[0] & 15, (r[0] >> 8) & 3, r[1] & 0xffff, r[1] >> 16, r[3] & 0xffff, r[3] >> 16, r[4], r[5] & 1, r[6], r[7])
            want = (t['post'], t['bufC'], t['ci'], t['cj'], t['kb0'], t['nkb'], t['need_c'], t['fin'], t['wk'], t['need2'])
            bad += got != want
        print('%d tasks, %d differ from the library\'s table; chain needs equal: %s' % (len(tab), bad, np.array_equal(np.array(need), z['need'])))
        sys.exit(1 if bad else 0)
    for B in [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 6, 10]:
        a = model(B, False)
        b = model(B, True)
        print('B=%2d  potrf %.3f ms (chain ends %.3f, step median %.1f us, %d tasks)   potrf_inv %.3f ms (chain ends %.3f, step %.1f us, %d tasks)' % (
            B, a[0] / 1e3, a[1] / 1e3, a[2], a[3], b[0] / 1e3, b[1] / 1e3, b[2], b[3]))
