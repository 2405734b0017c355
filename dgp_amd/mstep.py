"""Lock-step M-step.

The reference fits its GP nodes one after another, each with scipy's L-BFGS-B (dgp.py:1391-1398,
kernel_class.py:516-579); every objective evaluation needs K assembly + Cholesky + inverse + reductions on the
device.  One factorisation is a latency-bound chain that leaves most of an MI355X idle, and independent
factorisations on separate HIP streams overlap only ~3x (the launches serialise on the host side of the runtime).
So the optimisers advance in LOCK-STEP from one thread instead: scipy's reverse-communication L-BFGS-B core is
driven for all nodes at once, and every round the pending objective evaluations run as ONE batched pipeline
(per-node K assembly with its own hyper-parameters, batched potrf / potri graphs, per-node reductions, a single
device-to-host copy).  Every node sees exactly the iterates scipy.optimize.minimize would give it.
"""

import os
import numpy as np

from . import dist as ddist

try:   # scipy 1.15's reverse-communication L-BFGS-B core (the routine scipy.optimize.minimize itself drives)
    import scipy
    from scipy.optimize import _lbfgsb as _core
    _HAVE_CORE = tuple(int(v) for v in scipy.__version__.split('.')[:2]) == (1, 15) and \
        'ln_task' in (_core.setulb.__doc__ or '')
except Exception:   # pragma: no cover
    _core, _HAVE_CORE = None, False


class _Problem:
    """State of one L-BFGS-B run, laid out as scipy's _minimize_lbfgsb does (scipy/optimize/_lbfgsb_py.py)."""

    def __init__(self, x0, lb, ub, maxiter, maxfun, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxls=20):
        n = len(x0)
        self.m, self.maxls, self.maxiter, self.maxfun = maxcor, maxls, maxiter, maxfun
        self.factr, self.pgtol = ftol / np.finfo(float).eps, gtol
        self.nbd = np.zeros(n, np.int32)
        self.low, self.up = np.zeros(n), np.zeros(n)
        x0 = np.asarray(x0, dtype=np.float64).ravel()
        if lb is not None:
            if (lb > ub).any():
                raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
            x0 = np.clip(x0, lb, ub)
            for i in range(n):
                lo, hi = np.isfinite(lb[i]), np.isfinite(ub[i])
                if lo:
                    self.low[i] = lb[i]
                if hi:
                    self.up[i] = ub[i]
                self.nbd[i] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[(lo, hi)]
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0)
        self.g = np.zeros(n)
        self.wa = np.zeros(2 * maxcor * n + 5 * n + 11 * maxcor * maxcor + 8 * maxcor)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29)
        self.nit = self.nfev = 0
        self.finished = False

    def advance(self):
        """Run the core until it asks for f and g at self.x (returns True) or stops (returns False)."""
        while True:
            _core.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa,
                         self.iwa, self.task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if self.task[0] == 3:
                return True
            if self.task[0] == 1:   # new iteration: same stopping tests as scipy
                self.nit += 1
                if self.nit >= self.maxiter:
                    self.task[0], self.task[1] = 5, 504
                elif self.nfev > self.maxfun:
                    self.task[0], self.task[1] = 5, 502
            else:
                self.finished = True
                return False


def minimize_lockstep(problems, evaluate, on_finish=None, after_first_launch=None, groups=1):
    """Run several independent L-BFGS-B minimisations in lock-step from ONE thread: every round, all runs that want an
    objective value get it from one call evaluate([(index, x), ...]) -> [(f, g), ...].  Each run sees exactly the
    sequence of points scipy.optimize.minimize(method='L-BFGS-B') would give it (same core, same options).
    An `evaluate` with .launch(req, slot) -> token and .collect(req, token) halves lets `after_first_launch()` (host work
    of the caller that does not touch what the evaluations read) run between the two halves of the FIRST round, and --
    groups > 1 -- keeps that many groups of runs in flight: while the host waits for one group's results and advances its
    optimisers, the other groups' evaluations keep the device busy (no idle turn-around between rounds).  The runs are
    independent, so the grouping changes the order of the launches, not what any run sees.  Returns the number of rounds
    of the longest-running group."""
    if not _HAVE_CORE:
        raise RuntimeError('scipy L-BFGS-B core not available')
    told = set()

    def advance(members):
        want = [i for i in members if not problems[i].finished and problems[i].advance()]
        if on_finish is not None:   # (before the next round's launches reuse the buffers of the runs that have just ended)
            for i in members:
                if problems[i].finished and i not in told:
                    told.add(i)
                    on_finish(i)
        return [(i, problems[i].x.copy()) for i in want]

    def absorb(req, out):
        for (i, _), (f, g) in zip(req, out):
            p = problems[i]
            p.f = np.array(float(np.asarray(f).reshape(-1)[0]))
            p.g = np.asarray(g, dtype=np.float64).copy()
            p.nfev += 1

    split = hasattr(evaluate, 'launch')
    groups = max(1, min(int(groups), len(problems))) if split else 1
    if groups == 1:
        rounds = 0
        everyone = range(len(problems))
        while True:
            req = advance(everyone)
            if not req:
                return rounds
            if rounds == 0 and after_first_launch is not None and split:
                token = evaluate.launch(req, 0)
                try:
                    after_first_launch()
                except BaseException:   # (leave no evaluation in flight behind an exception of the caller's host work)
                    if hasattr(evaluate, 'abandon'):
                        try:
                            evaluate.abandon(token)
                        except Exception:   # noqa: BLE001
                            pass
                    raise
                out = evaluate.collect(req, token)
            else:
                if rounds == 0 and after_first_launch is not None:
                    after_first_launch()
                out = evaluate(req)
            absorb(req, out)
            rounds += 1
    members = [list(range(g, len(problems), groups)) for g in range(groups)]   # (interleaved: neighbours differ in cost)
    flying = {}
    rounds = [0] * groups
    try:
        for g in range(groups):
            req = advance(members[g])
            if req:
                flying[g] = (req, evaluate.launch(req, g))
        if after_first_launch is not None:
            after_first_launch()
        g = 0
        while flying:
            if g in flying:
                req, token = flying.pop(g)
                absorb(req, evaluate.collect(req, token))
                rounds[g] += 1
                req = advance(members[g])
                if req:
                    flying[g] = (req, evaluate.launch(req, g))
            g = (g + 1) % groups
    finally:
        for req, token in flying.values():   # (an exception: leave no mailbox occupied)
            try:
                evaluate.abandon(token)
            except Exception:   # noqa: BLE001
                pass
    return max(rounds)


def maximise_lockstep(engine, nodes, cache, after_first_launch=None):
    """kernel.maximise() for several dense GP nodes at once (dgp.py:1391-1398 runs them one after another; given
    the imputed latents their objectives are independent): one L-BFGS-B state per node, every round's objective
    evaluations batched on the device by ONE C call (dgpamd_llik_batch).  after_first_launch(): host work of the caller that
    touches nothing the evaluations read (dgp.train: the deferred refresh of the numpy attributes, the R2 diagnostics) -- run
    between the two halves of the FIRST round's call, while the device works on it.  Returns (rounds, evaluations)."""
    setups = [nd._opt_setup() for nd in nodes]
    problems = [_Problem(x0, lb, ub, opts.get('maxiter', 15000), opts.get('maxfun', 15000)) for x0, lb, ub, opts in setups]
    evals = [0]
    for nd in nodes:
        nd._stage()
        nd._in_maximise = True
    groups = {}
    for i, nd in enumerate(nodes):
        groups.setdefault(len(nd.output), []).append(i)
    plans, where = {}, {}
    with engine.stream():
        for n, idxs in groups.items():
            plans[n] = engine.llik_plan(n, [dict(kind=nodes[i].name, Xloc=nodes[i]._staged['Xl'], Xglob=nodes[i]._staged['Xg'],
                                                 nlen=len(nodes[i].length), nugget_est=nodes[i].nugget_est,
                                                 W=nodes[i]._staged['W'], y=nodes[i]._staged['y']) for i in idxs])
            for pos, i in enumerate(idxs):
                where[i] = (n, pos)

    def prepare(req):
        todo = {}
        for i, x in req:
            nd = nodes[i]
            nd.update(x)
            n, pos = where[i]
            plans[n].set(pos, nd.length, nd.nugget[0])
            todo.setdefault(n, []).append((pos, i))
        return list(todo.items())

    def finish(req, groups, first=None):
        host = {}
        for g, (n, lst) in enumerate(groups):   # (inside the engine's stream context entered once below, not once per round)
            res = plans[n].wait(first) if (g == 0 and first is not None) else plans[n].run([pos for pos, _ in lst])
            for r, (pos, i) in enumerate(lst):
                host[i] = res[pos]
                last[i] = (n, r, res[pos])   # where this node's factor is now: row r of the run's buffers
        evals[0] += len(req)
        return [nodes[i]._llik_finish(host[i]) for i, _ in req]

    def evaluate(req):
        return finish(req, prepare(req))

    def launch(req, slot=0):   # the first group of sizes queued (one evaluation in flight per engine); the others follow in collect
        groups = prepare(req)
        n, lst = groups[0]
        return groups, plans[n].launch([pos for pos, _ in lst])

    def collect(req, token):
        return finish(req, token[0], first=token[1])
    if after_first_launch is not None:
        evaluate.launch, evaluate.collect = launch, collect
        evaluate.abandon = lambda token: plans[token[0][0][0]].wait(token[1])   # (wait for the queued evaluation, drop its results)

    last = {}
    imp = getattr(cache, 'imp', None)

    def on_finish(i):
        # The reference keeps the hyper-parameters of the LAST evaluation (kernel_class.py:516-579 discards minimize's
        # result; llik() has updated the node), so the factorisation of that evaluation is the one the next I-step needs:
        # the prior factor of a first-layer node, the log-likelihood of the current latents for a node of the last layer.
        if imp is not None and i in last and hasattr(imp, 'adopt_from_mstep'):
            n, r, host = last[i]
            imp.adopt_from_mstep(nodes[i], plans[n].factor_view(r), host)

    try:
        with engine.stream():
            rounds = minimize_lockstep(problems, evaluate, on_finish, after_first_launch=after_first_launch)
    finally:
        for nd in nodes:
            nd._in_maximise = False
            nd.iter_count = 0
    for nd in nodes:
        nd.add_to_path()
    return rounds, evals[0]


def maximise_lockstep_vecch(engine, nodes, after_first_launch=None):
    """kernel.maximise() for several Vecchia GP nodes at once: the same lock-step driver, every round's objective evaluations
    (vecchia_nllik, one launch per node) queued back to back and fetched with ONE synchronisation -- the reference (and
    kernel.maximise) pays a host round trip per node and evaluation, which at n = 50 000 is two thirds of a 1-ms kernel.
    Only for nodes whose optimiser runs without a callback (kernel_class.py:537-542: DGP nodes, isotropic GP nodes).  With the
    likelihood rows split over ranks (dist.split_training(rows=True)) every rank drives the same optimisers on the same
    all-reduced sums: one collective per round.  Returns (rounds, evaluations)."""
    import torch
    setups = [nd._opt_setup() for nd in nodes]
    problems = [_Problem(x0, lb, ub, opts.get('maxiter', 15000), opts.get('maxfun', 15000)) for x0, lb, ub, opts in setups]
    evals = [0]
    for nd in nodes:
        nd._vecch_fixed = nd._vecch_stage(trust_pre=True)   # inputs, outputs, neighbours: fixed during the run
        nd._in_maximise = True

    def launch(req, slot=0):   # one row launch per node, queued back to back; the sums start their way to the host at once
        outs = []
        for i, x in req:
            nd = nodes[i]
            nd.update(x)
            outs.append(nd._llik_vecch_device())
        buf = torch.cat([o for o, _ in outs])
        if ddist.rows_split():   # every rank evaluated its rows of every node: ONE all-reduce per round
            buf = ddist.allreduce_sum_vector(buf)
        return outs, engine.post(buf, slot)

    def collect(req, token):   # ONE wait for the round -- for these sums, not for what has been queued behind them
        outs, posted = token
        host = engine.collect(posted)
        res, at = [], 0
        for (i, _), (o, P) in zip(req, outs):
            k = o.numel()
            res.append(nodes[i]._llik_vecch_finish(host[at:at + k], P))
            at += k
        evals[0] += len(req)
        return res

    def evaluate(req):
        return collect(req, launch(req))
    evaluate.launch, evaluate.collect = launch, collect   # (the groups hold different nodes: a node's state is its own run's)
    evaluate.abandon = lambda token: engine.discard(token[1])
    # (one mailbox per group in flight, slots 0 .. groups-1: never more than the engine reserves for the M-step -- the deferred
    #  detach of the I-step posts the latents to the slots behind them)
    groups = max(1, min(int(os.environ.get('DGPAMD_MSTEP_GROUPS', '2')), engine.MSTEP_SLOTS)) if len(nodes) >= 4 else 1

    try:
        with engine.stream():
            rounds = minimize_lockstep(problems, evaluate, after_first_launch=after_first_launch, groups=groups)
    finally:
        for nd in nodes:
            nd._in_maximise = False
            nd.iter_count = 0
            nd.__dict__.pop('_vecch_fixed', None)
    for nd in nodes:
        nd.add_to_path()
    return rounds, evals[0]
