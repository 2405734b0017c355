"""Critical-path analysis of a full task log of the one-launch factorisation (tools/gpu_mega_tasklog.py -> .npz).
Rebuilds who produced every tile version, finds for every task the input that arrived last (or the worker it had to wait
for), walks the critical path back from the last publication and prints where its time went; also the chain's stalls per
block step with the task that ended each.  usage: analyze_tasklog.py log.npz [--path] [--stalls] [--matrix b]"""
import sys
import numpy as np

FINAL = 1 << 30
STORE, SOLVE, TDIAG, LOOK, LOOKD = 0, 1, 3, 4, 5
KIND = {STORE: 'upd', SOLVE: 'solve', TDIAG: 'tdiag', LOOK: 'look', LOOKD: 'lookd'}
BUF = 'ATS'


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    B, nbk, ntask = int(d['B']), int(d['nbk']), int(d['ntask'])
    log = d['log'].astype(np.float64)
    t0 = log[0]
    ch = (log[64:64 + 8 * B * nbk].reshape(B, nbk, 8) - t0) / 100.0
    ch[d['log'][64:64 + 8 * B * nbk].reshape(B, nbk, 8) == 0] = np.nan
    tk_raw = d['log'][64 + 8 * B * nbk:64 + 8 * B * nbk + 8 * B * ntask].reshape(B, ntask, 8)
    tk = (tk_raw.astype(np.float64) - t0) / 100.0
    tk[tk_raw == 0] = np.nan
    wg = tk_raw[:, :, 6].copy()
    ready2 = d['log'][64:64 + 8 * B * nbk].reshape(B, nbk, 8)[:, :, 6]
    tab = d['table']
    T = dict(post=tab[:, 0] & 15, first=(tab[:, 0] >> 4) & 1, bufC=(tab[:, 0] >> 8) & 3, bufL=(tab[:, 0] >> 10) & 3, bufR=(tab[:, 0] >> 12) & 3,
             ci=tab[:, 1] & 0xffff, cj=tab[:, 1] >> 16, li=tab[:, 2] & 0xffff, ri=tab[:, 2] >> 16, kb0=tab[:, 3] & 0xffff, nkb=tab[:, 3] >> 16,
             need_c=tab[:, 4], fin=tab[:, 5] & 1, step=tab[:, 5] >> 8, wk=tab[:, 6], need2=tab[:, 7])
    return d, B, nbk, ntask, ch, tk, wg, ready2, T


def describe(T, s):
    p = int(T['post'][s])
    base = '%s %s(%d,%d)' % (KIND[p], BUF[int(T['bufC'][s])], T['ci'][s], T['cj'][s])
    if p != TDIAG:
        base += ' p%d+%d' % (T['kb0'][s], T['nkb'][s])
    if p in (SOLVE, TDIAG, LOOK, LOOKD):
        base += ' W%d' % T['wk'][s]
    return base


def main():
    path = sys.argv[1]
    d, B, nbk, ntask, ch, tk, wg, ready2, T = load(path)
    inv = int(d['inv'])
    print('%s: n=%d B=%d inv=%d nbk=%d tasks/matrix=%d, launch %.3f ms logged (plain: %s)' % (
        path, int(d['n']), B, inv, nbk, ntask, d['ms'][-1], ' '.join('%.3f' % v for v in d['ms'][:-1])))
    # ---- producers: (b, buf, i, j) -> sorted list of (version, time, ('t', slot) | ('c', k)) ----
    prod = {}

    def add(b, buf, i, j, ver, t, who):
        prod.setdefault((b, buf, i, j), []).append((ver, t, who))

    wtime = np.full((B, nbk), np.nan)
    for b in range(B):
        for k in range(nbk):
            # W_k: the flag is stored right before stamp 2 (or, when the panel tile was late, before the wait: then ~stamp 1 + 1 us)
            late = (k + 1 < nbk) and not (int(ready2[b, k]) & 1)
            wtime[b, k] = (ch[b, k, 1] + 1.0) if (late or k + 1 == nbk) else ch[b, k, 2]
            if k + 1 < nbk:
                add(b, 0, k + 1, k, FINAL, ch[b, k, 4] + 0.6, ('c', k))   # the chain's panel tile (published from the update's first part)
        for s in range(ntask):
            p = int(T['post'][s])
            newver = FINAL if T['fin'][s] else int(T['need_c'][s]) + 1
            if np.isnan(tk[b, s, 5]):
                continue
            if p in (STORE, SOLVE, TDIAG):
                add(b, int(T['bufC'][s]), int(T['ci'][s]), int(T['cj'][s]), newver, tk[b, s, 5], ('t', s))
            elif p == LOOK:
                add(b, 0, int(T['ci'][s]), int(T['cj'][s]) + 1, int(T['need2'][s]) + 1, tk[b, s, 5], ('t', s))
            elif p == LOOKD:
                add(b, 0, int(T['ci'][s]), int(T['cj'][s]), FINAL, tk[b, s, 7] + 1.5, ('t', s))
                add(b, 0, int(T['ci'][s]), int(T['ci'][s]), int(T['need2'][s]) + 1, tk[b, s, 5], ('t', s))

    def arrival(b, buf, i, j, ver):
        """time and producer of the first publication of tile (buf, i, j) with a version >= ver (0: initial state)"""
        if ver <= 0:
            return 0.0, None
        best = None
        for v, t, who in prod.get((b, buf, i, j), ()):
            if v >= ver and (best is None or t < best[0]):
                best = (t, who)
        return best if best else (np.nan, None)

    def inputs_of(b, s):
        """first-wait inputs of task s: list of (time, producer, label)"""
        p = int(T['post'][s])
        res = []
        if p == TDIAG:
            return [(wtime[b, int(T['wk'][s])], ('c', int(T['wk'][s])), 'W%d' % T['wk'][s])]
        t, w = arrival(b, int(T['bufC'][s]), int(T['ci'][s]), int(T['cj'][s]), int(T['need_c'][s]))
        res.append((t, w, 'C'))
        for kb in range(int(T['kb0'][s]), int(T['kb0'][s]) + int(T['nkb'][s])):
            for buf, i, nm in ((int(T['bufL'][s]), int(T['li'][s]), 'L'), (int(T['bufR'][s]), int(T['ri'][s]), 'R')):
                t, w = arrival(b, buf, i, kb, FINAL)
                res.append((t, w, '%s%s(%d,%d)' % (nm, BUF[buf], i, kb)))
        return res

    def second_inputs(b, s):
        p = int(T['post'][s])
        res = []
        if p in (SOLVE, LOOK, LOOKD):
            res.append((wtime[b, int(T['wk'][s])], ('c', int(T['wk'][s])), 'W%d' % T['wk'][s]))
        if p == LOOK:
            t, w = arrival(b, 0, int(T['ci'][s]) - 1, int(T['cj'][s]), FINAL)
            res.append((t, w, 'P'))
            t, w = arrival(b, 0, int(T['ci'][s]), int(T['cj'][s]) + 1, int(T['need2'][s]))
            res.append((t, w, 'Q'))
        if p == LOOKD:
            t, w = arrival(b, 0, int(T['ci'][s]), int(T['ci'][s]), int(T['need2'][s]))
            res.append((t, w, 'D'))
        return res

    # previous task of the same workgroup (by pull time)
    prev_on_wg = {}
    for b in range(B):
        pass
    order = {}
    for b in range(B):
        for s in range(ntask):
            if not np.isnan(tk[b, s, 0]):
                order.setdefault(int(wg[b, s]), []).append((tk[b, s, 0], b, s))
    for w_, lst in order.items():
        lst.sort()
        for a, c in zip(lst[:-1], lst[1:]):
            prev_on_wg[(c[1], c[2])] = (a[1], a[2])

    # ---- summary of task phases ----
    end = np.nanmax(tk[:, :, 5])
    chain_end = np.nanmax(ch[:, nbk - 1, 1])
    print('chains end (last block factored) %.1f us; last task published %.1f us; tail %.1f us' % (chain_end, end, end - chain_end))
    for b in range(B):
        steps = ch[b, 1:, 0] - ch[b, :-1, 0]
        print('  matrix %d: chain %.1f us, step median %.1f, mean %.1f, >15us: %d steps, sum of excess over 12.6: %.1f us' % (
            b, ch[b, nbk - 1, 1] - ch[b, 0, 0], np.median(steps), steps.mean(), int((steps > 15).sum()), np.clip(steps - 12.6, 0, None).sum()))
    nworkers = len(order)
    busy = 0.0
    wait_in = 0.0
    for b in range(B):
        for s in range(ntask):
            if np.isnan(tk[b, s, 0]):
                continue
            wait_in += tk[b, s, 1] - tk[b, s, 0]
            busy += tk[b, s, 5] - tk[b, s, 1]
    print('workers seen: %d; time in tasks after their inputs arrived %.0f us (%.1f %% of workers x launch), waiting inside tasks for first inputs %.0f us' % (
        nworkers, busy, 100 * busy / (nworkers * end), wait_in))

    # ---- per-kind latency from the last input's arrival to the publication ----
    print('by kind: count | median (inputs seen - last input published) | compute | W wait | store+publish | pulled after inputs were ready (count, median lateness)')
    for p in (STORE, SOLVE, LOOK, LOOKD, TDIAG):
        for buf in range(3):
            rows = []
            for b in range(B):
                for s in np.nonzero((T['post'] == p) & (T['bufC'] == buf))[0]:
                    if np.isnan(tk[b, s, 5]):
                        continue
                    ins = inputs_of(b, s)
                    last = np.nanmax([t for t, _, _ in ins]) if ins else 0.0
                    rows.append((tk[b, s, 1] - max(last, tk[b, s, 0]), tk[b, s, 2] - tk[b, s, 1],
                                 (tk[b, s, 3] - tk[b, s, 2]) if p in (SOLVE, LOOK, LOOKD) else 0.0,
                                 tk[b, s, 5] - (tk[b, s, 3] if p in (SOLVE, LOOK, LOOKD) else tk[b, s, 2]),
                                 tk[b, s, 0] - last, int(T['nkb'][s])))
            if not rows:
                continue
            r = np.array(rows)
            late = r[:, 4] > 0.5
            print('  %-6s %s: %5d | %5.1f | %5.1f (%.1f panels) | %5.1f | %5.1f | %d, %.1f us' % (
                KIND[p], BUF[buf], len(r), np.median(r[:, 0]), np.median(r[:, 1]), r[:, 5].mean(), np.median(r[:, 2]), np.median(r[:, 3]),
                int(late.sum()), np.median(r[late, 4]) if late.any() else 0.0))

    # ---- chain stalls ----
    if '--stalls' in sys.argv:
        mb = int(sys.argv[sys.argv.index('--matrix') + 1]) if '--matrix' in sys.argv else 0
        need = d['need']
        print('matrix %d, per block step: factor | W+stage (incl. wait for the panel tile) | wait diag | solve | update || panel tile published at (rel. factor end), by; diag tile likewise' % mb)
        for k in range(nbk - 1):
            c = ch[mb, k]
            tq, wq = arrival(mb, 0, k + 1, k, int(need[k][0]))
            td, wd = arrival(mb, 0, k + 1, k + 1, int(need[k][1]))
            def nm(w):
                return '-' if w is None else ('chain %d' % w[1] if w[0] == 'c' else describe(T, w[1]))
            print('%2d | %5.1f | %5.1f (issue %4.1f, wait + stage %4.1f) | %5.1f | %5.1f | %5.1f || Q %+6.1f %-28s D %+6.1f %s' % (
                k, c[1] - c[0], c[2] - c[1], c[7] - c[1], c[2] - c[7], c[3] - c[2], c[4] - c[3], c[5] - c[4], tq - c[1], nm(wq), td - c[1], nm(wd)))

    # ---- critical path, walked back from the last publication ----
    if '--path' in sys.argv:
        b, s = np.unravel_index(np.nanargmax(tk[:, :, 5]), tk[:, :, 5].shape)
        cur = ('t', int(b), int(s))
        cat = {}
        lines = []
        t_cur = end
        guard = 0
        while cur is not None and guard < 4000:
            guard += 1
            if cur[0] == 't':
                _, b, s = cur
                p = int(T['post'][s])
                ins = inputs_of(b, s)
                tin, win, lab = max(ins, key=lambda x: (-1 if np.isnan(x[0]) else x[0]))
                sec = second_inputs(b, s)
                # walk back: published <- (second wait) <- computed <- inputs seen <- {last input | pull <- previous task of the worker}
                seg_end = t_cur
                t_ready = tk[b, s, 1]
                nxt = None
                if sec:
                    t2, w2, l2 = max(sec, key=lambda x: (-1 if np.isnan(x[0]) else x[0]))
                    if t2 > tk[b, s, 2] + 0.3:   # the second wait was binding
                        cat['task after 2nd wait (%s)' % KIND[p]] = cat.get('task after 2nd wait (%s)' % KIND[p], 0.0) + seg_end - t2
                        lines.append('%8.1f  %-34s m%d  <- %s at %.1f (2nd wait), then %.1f us to publish' % (seg_end, describe(T, s), b, l2, t2, seg_end - t2))
                        nxt = (w2, t2)
                if nxt is None:
                    if tin >= tk[b, s, 0] - 0.3 or (b, s) not in prev_on_wg:
                        cat['task (%s %s)' % (KIND[p], BUF[int(T['bufC'][s])])] = cat.get('task (%s %s)' % (KIND[p], BUF[int(T['bufC'][s])]), 0.0) + seg_end - tin
                        lines.append('%8.1f  %-34s m%d  <- %s at %.1f, seen %.1f, %.1f us to publish' % (seg_end, describe(T, s), b, lab, tin, t_ready, seg_end - tin))
                        nxt = (win, tin)
                    else:
                        cat['task (%s %s) pulled late' % (KIND[p], BUF[int(T['bufC'][s])])] = cat.get('task (%s %s) pulled late' % (KIND[p], BUF[int(T['bufC'][s])]), 0.0) + seg_end - tk[b, s, 0]
                        pb, ps = prev_on_wg[(b, s)]
                        lines.append('%8.1f  %-34s m%d  <- pulled at %.1f (inputs ready since %.1f): worker busy with %s m%d' % (seg_end, describe(T, s), b, tk[b, s, 0], tin, describe(T, ps), pb))
                        cur = ('t', pb, ps)
                        t_cur = tk[b, s, 0]
                        continue
                w, t = nxt
                if w is None:
                    break
                t_cur = t
                cur = ('t', b, w[1]) if w[0] == 't' else ('c', b, w[1])
            else:
                _, b, k = cur
                # chain of matrix b produced something in step k at t_cur; the step started at ch[b,k,0]; walk through the step
                c = ch[b, k]
                tq, wq = arrival(b, 0, k + 1, k, int(d['need'][k][0])) if k + 1 < nbk else (np.nan, None)
                td, wd = arrival(b, 0, k + 1, k + 1, int(d['need'][k][1])) if k + 1 < nbk else (np.nan, None)
                # did this step's output at t_cur depend on a late worker tile?  (only when t_cur is after the stage / diag waits)
                if k + 1 < nbk and t_cur >= c[3] - 0.05 and td > c[2] + 0.3 and td >= tq:
                    cat['chain after diag-tile wait'] = cat.get('chain after diag-tile wait', 0.0) + t_cur - td
                    lines.append('%8.1f  chain step %d m%d  <- diagonal tile at %.1f' % (t_cur, k, b, td))
                    t_cur = td
                    cur = ('t', b, wd[1]) if wd and wd[0] == 't' else None
                    continue
                if k + 1 < nbk and t_cur >= c[2] - 0.05 and tq > c[1] + 0.5:
                    cat['chain after panel-tile wait'] = cat.get('chain after panel-tile wait', 0.0) + t_cur - tq
                    lines.append('%8.1f  chain step %d m%d  <- panel tile at %.1f (factor ended %.1f)' % (t_cur, k, b, tq, c[1]))
                    t_cur = tq
                    cur = ('t', b, wq[1]) if wq and wq[0] == 't' else None
                    continue
                cat['chain'] = cat.get('chain', 0.0) + t_cur - c[0]
                lines.append('%8.1f  chain step %d m%d  (started %.1f)' % (t_cur, k, b, c[0]))
                t_cur = c[0]
                if k == 0:
                    break
                # the step started when the previous one ended; the previous step's end depended on its own waits
                cur = ('c', b, k - 1)
        print('critical path back from the last publication (%d hops):' % len(lines))
        for ln in (lines if '--full' in sys.argv else lines[:60]):
            print(ln)
        print('time on the critical path by category (us):')
        for k_, v in sorted(cat.items(), key=lambda x: -x[1]):
            print('  %-40s %8.1f' % (k_, v))
        print('  %-40s %8.1f' % ('sum', sum(cat.values())))


if __name__ == '__main__':
    main()


def look_table(path, mb):
    """the look-ahead tasks of matrix mb, relative to the end of the chain's factorisation of their block"""
    d, B, nbk, ntask, ch, tk, wg, ready2, T = load(path)
    print('matrix %d look-ahead tasks, us after the chain factored block k: kind | pulled | own inputs seen | first run done | W_k seen | 2nd wait done | stored | published | worker' % mb)
    for s in range(ntask):
        p = int(T['post'][s])
        if p not in (LOOK, LOOKD):
            continue
        k = int(T['wk'][s])
        t0 = ch[mb, k, 1]
        r = tk[mb, s]
        print('%2d %-5s | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f | wg %d' % (k, KIND[p], r[0] - t0, r[1] - t0, r[2] - t0, r[3] - t0, r[7] - t0, r[4] - t0, r[5] - t0, int(wg[mb, s])))


if '--look' in sys.argv:
    look_table(sys.argv[1], int(sys.argv[sys.argv.index('--matrix') + 1]) if '--matrix' in sys.argv else 0)
