"""The query neighbour search's filter-then-select path (round 5: nn_tau / nn_collect / nn_pick) against the streaming top-k kernel
(DGPAMD_NN_FILTER=0) on the same inputs: identical neighbour arrays, and the time of each.  One subprocess per setting (the switch is
read once per process).  usage: gpu_nn_filter_check.py [quick]"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'run':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from dgp_amd.ops import Engine
    eng = Engine(0)
    out = {}
    rng = np.random.default_rng(11)
    cases = []
    for (M, n, D, m) in ((100000, 50000, 16, 50), (100000, 50000, 8, 50), (20000, 50000, 8, 50), (7000, 20000, 3, 31), (100000, 50000, 16, 64)):
        cases.append(('uniform M=%d n=%d D=%d m=%d' % (M, n, D, m), rng.uniform(size=(M, D)), rng.uniform(size=(n, D)), m))
    # clustered candidates: most queries' neighbours lie in a cluster the strided sample barely touches (exercises wide taus and the slow path)
    xc = np.concatenate([rng.normal(size=(400, 4)) * 1e-3, rng.uniform(size=(29600, 4)) * 40.0])
    rng.shuffle(xc)
    cases.append(('clustered n=30000 D=4 m=50', np.concatenate([rng.normal(size=(4000, 4)) * 1e-3, rng.uniform(size=(4000, 4)) * 40.0]), xc, 50))
    g = np.stack(np.meshgrid(np.arange(160.), np.arange(150.)), -1).reshape(-1, 2)   # a grid: many exactly tied distances
    cases.append(('grid with ties n=24000 D=2 m=40', g[rng.integers(0, len(g), 8000)], g, 40))
    cases.append(('all candidates equal n=20000 D=3 m=50', rng.uniform(size=(6500, 3)), np.ones((20000, 3)), 50))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, q, x, m in cases:
        qd, xd = eng.tensor(q), eng.tensor(x)
        eng.nn_query(qd, xd, m)
        torch.cuda.synchronize()
        with eng.stream():
            st = torch.cuda.current_stream()
            ev0.record(st)
            for _ in range(3):
                nn = eng.nn_query(qd, xd, m)
            ev1.record(st)
        torch.cuda.synchronize()
        out[name] = nn.cpu().numpy()
        print('%-44s %8.2f ms' % (name, ev0.elapsed_time(ev1) / 3), flush=True)
    np.savez(sys.argv[2], **{k.replace(' ', '_').replace('=', ''): v for k, v in out.items()})
    sys.exit(0)
res = {}
for flt in ('1', '0'):
    f = '/tmp/nnf_%s.npz' % flt
    r = subprocess.run([sys.executable, __file__, 'run', f], env=dict(os.environ, DGPAMD_NN_FILTER=flt), capture_output=True, text=True, timeout=900)
    print('== DGPAMD_NN_FILTER=%s (%s)' % (flt, 'filter, then select' if flt == '1' else 'streaming top-k'))
    print(r.stdout[-3000:], r.stderr[-1500:] if r.returncode else '')
    res[flt] = np.load(f)
bad = 0
for k in res['1'].files:
    same = np.array_equal(res['1'][k], res['0'][k])
    bad += not same
    print('%-50s identical neighbour arrays: %s' % (k, same))
print('ALL IDENTICAL' if not bad else '%d CASES DIFFER' % bad)
