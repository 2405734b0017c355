"""The linked-GP pair kernels under foreign load (the LDS-DMA / register-load retirement fact was met as wrong sums first; a defect that depends on
co-resident work would pass a single-process suite).  Process A computes Matern and SExp link_gp predictions alone, then again and again while two other processes
keep the GPU busy with factorisations and pair kernels of their own; every repetition must equal the first result BIT FOR BIT.
usage: python tools/gpu_pair_shared.py [repetitions]          (role 'load' is started by the script itself)"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(eng, kind, n, Dw, Dz, M, seed):
    from dgp_amd.ops import cell_order
    rng = np.random.default_rng(seed)
    W, Wg = rng.normal(size=(n, Dw)), rng.uniform(size=(n, Dz))
    if kind == 'matern2.5':
        p = cell_order(W)
        W, Wg = W[p], Wg[p]
    G = rng.normal(size=(n, 8)) / np.sqrt(n)
    Rinv, ry = G @ G.T + np.eye(n), rng.normal(size=n)
    m, v, z = rng.normal(size=(M, Dw)), rng.uniform(0.01, 0.4, size=(M, Dw)), rng.uniform(size=(M, Dz))
    t = eng.tensor
    return (kind, t(m), t(v), t(z), t(W), t(Wg), np.array([2.5]), t(Rinv), n, t(ry), 1.3, 1e-4)


def main():
    role = os.environ.get('PAIR_SHARED_ROLE', 'check')
    import torch
    from dgp_amd.ops import Engine
    eng = Engine(0)
    if role == 'load':   # foreign load: factorisations of several batch sizes and pair kernels, until the file goes away
        rng = np.random.default_rng(int(os.environ.get('SEED', '1')))
        n = 2000
        Np = eng.padded_dim(n)
        X = eng.tensor(rng.uniform(size=(6, n, 5)))
        y = eng.tensor(rng.normal(size=n))
        A, T, S = eng.empty(6, Np, Np), eng.empty(6, Np, Np), eng.empty(6, Np, Np)
        work = eng.potrf_workspace(n, 6)
        pr = problem(eng, 'matern2.5', 1000, 3, 2, 512, 99)
        first = True
        while os.path.exists(os.environ['PAIR_SHARED_FLAG']):
            if not first and not os.path.exists(os.environ['PAIR_SHARED_FLAG'] + '.ready.' + os.environ.get('SEED', '1')):
                open(os.environ['PAIR_SHARED_FLAG'] + '.ready.' + os.environ.get('SEED', '1'), 'w').close()   # (looping: the checker may start)
            first = False
            for B in (1, 3, 6):
                eng.kmatrix('matern2.5', X[:B] if B > 1 else X[0], None, None, [1.0], 1e-6, out=A[:B] if B > 1 else A[0], full=False, Y=y, batch=B)
                eng.potrf_inv(n, A, T, S, batch=B, work=work)
            eng.linkgp_predict(*pr)
            torch.cuda.synchronize()
        return
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    probs = [problem(eng, 'matern2.5', 2000, 5, 5, 2304, 1), problem(eng, 'sexp', 2000, 5, 5, 2304, 2), problem(eng, 'matern2.5', 700, 3, 1, 1100, 3)]
    alone = [[t.cpu().numpy().copy() for t in eng.linkgp_predict(*p)] for p in probs]
    again = [[t.cpu().numpy().copy() for t in eng.linkgp_predict(*p)] for p in probs]
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(alone, again)), 'not even alone'
    t0 = time.perf_counter()
    for p in probs:
        [t.cpu().numpy() for t in eng.linkgp_predict(*p)]
    t_alone = time.perf_counter() - t0
    flag = '/tmp/pair_shared_flag_%d' % os.getpid()
    open(flag, 'w').close()
    env = dict(os.environ, PAIR_SHARED_ROLE='load', PAIR_SHARED_FLAG=flag, DGPAMD_POTRF_MODE='0')   # (the load shares the device: per-block-step factorisation)
    loads = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, SEED=str(s))) for s in (1, 2)]
    try:
        for s_ in (1, 2):   # (wait until both loads have built their problems and are looping)
            w0 = time.perf_counter()
            while not os.path.exists(flag + '.ready.%d' % s_):
                time.sleep(0.2)
                if time.perf_counter() - w0 > 240 or any(p.poll() is not None for p in loads):
                    raise RuntimeError('the load processes did not start')
        bad, t0 = 0, time.perf_counter()
        for r in range(reps):
            for k, p in enumerate(probs):
                got = [t.cpu().numpy() for t in eng.linkgp_predict(*p)]
                if not (np.array_equal(got[0], alone[k][0]) and np.array_equal(got[1], alone[k][1])):
                    bad += 1
                    print('repetition %d problem %d differs: max |d mean| %.3e  max |d var| %.3e' % (r, k, np.abs(got[0] - alone[k][0]).max(), np.abs(got[1] - alone[k][1]).max()), flush=True)
        dt = time.perf_counter() - t0
        alive = sum(p.poll() is None for p in loads)
    finally:
        os.unlink(flag)
        for p in loads:
            p.wait(timeout=120)
        for s_ in (1, 2):
            if os.path.exists(flag + '.ready.%d' % s_):
                os.unlink(flag + '.ready.%d' % s_)
    print('%d repetitions x %d problems under the load of two other processes (%d still running at the end): %d results differ from the ones computed alone; '
          '%.3f s per repetition under load against %.3f s alone' % (reps, len(probs), alive, bad, dt / reps, t_alone))
    sys.exit(1 if bad or alive != 2 else 0)


if __name__ == '__main__':
    main()
