"""First real multi-GPU contact: two ranks on TWO devices over RCCL (backend 'nccl').

Every other multi-rank test of this suite is gloo, or two ranks sharing GPU 0, or an RCCL group of one rank -- all the pool's
one-GPU boxes allow.  This one skips unless at least two devices are visible and otherwise runs the three splits SURVEY 8(e)
names on two devices: emulator.predict with the imputations sharded (one all-reduce of the two moment arrays,
emulation.py:701-779,846-847) and with the test points sharded (one all-gather, emulation.py:603-613), the M-step with the nodes
dealt over the ranks (one all-gather of the fitted hyper-parameters, dgp.py:1455-1467), and a Vecchia I-step / M-step with the
likelihood rows split (one all-reduce per evaluation, vecchia.py:164-242).  The launcher starts the ranks as child processes
before anything here touches the GPU (torch.cuda.device_count() does not initialise it on this image)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, faulthandler, copy
faulthandler.dump_traceback_later(500, exit=True)   # (a stuck rank says where)
sys.path.insert(0, %r)
import numpy as np, torch
from dgp_amd import dgp, kernel, combine, emulator, dist as dd
backend = os.environ.get('TWO_DEV_BACKEND', 'nccl')       # 'gloo' + both ranks on GPU 0: the dry run of this script on a one-GPU box
dd.init_from_env(backend)
import torch.distributed as td
local = int(os.environ['LOCAL_RANK']) if backend == 'nccl' else 0
torch.cuda.set_device(local)
assert td.get_backend() == backend and td.get_world_size() == 2 and torch.cuda.current_device() == local
dev = torch.device('cuda', local)
ids = dd.allgather_vector(np.array([float(dd.rank()), float(torch.cuda.current_device())]), device=dev).reshape(-1, 2)
if backend == 'nccl':   # the two ranks really sit on two devices
    assert sorted(ids[:, 1].tolist()) == [0.0, 1.0], ids
t = torch.full((4,), float(dd.rank() + 1), dtype=torch.float64, device=dev)
dd.allreduce_sum(t)
assert t.tolist() == [3.0] * 4
print('rccl ok', flush=True)

rng = np.random.default_rng(3)
X = rng.uniform(size=(150, 3)); Y = np.sin(4 * X[:, [0]]) + X[:, [1]] * X[:, [2]]
def layers():
    return combine([kernel(length=np.array([1.0]), name='matern2.5') for _ in range(3)],
                   [kernel(length=np.array([1.0]), name='matern2.5', scale_est=True, connect=np.arange(3))])
def hyper(model):
    return np.concatenate([np.concatenate((nd.scale, nd.length, nd.nugget)) for layer in model.all_layer for nd in layer])

# ---- prediction: both ranks train the same model (same seed), then shard ----
m = dgp(X, Y, layers(), seed=7, device=local)
m.train(N=3, ess_burn=2, disable=True)
both = dd.allgather_objects(hyper(m).tobytes())
assert both[0] == both[1], 'the two devices trained different models from the same seed'
est = m.estimate()
xt = rng.uniform(size=(301, 3))
ref = emulator(copy.deepcopy(est), N=4, seed=9, shard=False, device=local).predict(xt)     # all four imputations on this rank
pt = emulator(copy.deepcopy(est), N=4, seed=9, shard='points', device=local)               # the same imputations, half of the points
assert pt.shard_points and pt.N == 4
mu, var = pt.predict(xt)
np.testing.assert_allclose(mu, ref[0], rtol=1e-10, atol=1e-12)
np.testing.assert_allclose(var, ref[1], rtol=1e-9, atol=1e-13)
sh = emulator(copy.deepcopy(est), N=4, seed=9, device=local)                                 # two imputations per rank, one all-reduce
assert sh.shard and sh.N == 2
ms, vs = sh.predict(xt)
box = dd.allgather_objects((ms.tobytes(), vs.tobytes()))
assert box[0] == box[1], 'the ranks hold different reduced moments'
# (the sharded emulator draws other imputations than the unsharded one: same posterior, Monte-Carlo agreement only; the
#  exact check of the reduction: the mixture of the two ranks' own halves)
assert np.all(np.isfinite(ms)) and np.all(np.isfinite(vs)) and np.sqrt(np.mean((ms - ref[0]) ** 2)) < 0.1
mine = tuple(t.cpu().numpy() for t in sh._layer_moments(xt)[-1])      # this rank's (S/2, M, 1) means and variances
parts = dd.allgather_objects((mine[0].tobytes(), mine[1].tobytes(), mine[0].shape))
mu_s = np.concatenate([np.frombuffer(a, dtype=np.float64).reshape(shp) for a, _, shp in parts])
v_s = np.concatenate([np.frombuffer(b, dtype=np.float64).reshape(shp) for _, b, shp in parts])
assert mu_s.shape[0] == 4
np.testing.assert_allclose(ms, mu_s.mean(0), rtol=1e-10, atol=1e-12)                                   # emulation.py:846
np.testing.assert_allclose(vs, (mu_s ** 2 + v_s).mean(0) - mu_s.mean(0) ** 2, rtol=1e-9, atol=1e-12)   # emulation.py:847
print('prediction ok', flush=True)

# ---- M-step nodes split == unsplit, bit for bit ----
def dense(split):
    dd.split_training(rows=False, nodes=split)
    mm = dgp(X, Y, layers(), seed=7, device=local)
    mm.train(N=3, ess_burn=2, disable=True)
    return mm
a, b = dense(True), dense(False)
assert np.array_equal(hyper(a), hyper(b)), (hyper(a), hyper(b))
for la, lb in zip(a.all_layer, b.all_layer):
    for na, nb in zip(la, lb):
        assert np.array_equal(na.para_path, nb.para_path)
dd.split_training(nodes=False)
print('node split ok', flush=True)

# ---- Vecchia rows split: the I-step within 1e-9 of the unsplit one, the same on both ranks ----
Xv = rng.uniform(size=(260, 2)); Yv = np.sin(5 * Xv[:, [0]]) + Xv[:, [1]] ** 2
def vecch():
    np.random.seed(5)
    ls = combine([kernel(length=np.array([1.0]), name='sexp') for _ in range(2)],
                 [kernel(length=np.array([1.0]), name='sexp', scale_est=True, connect=np.arange(2))])
    return dgp(Xv, Yv, ls, seed=9, vecchia=True, m=8, device=local)
e_, f_ = vecch(), vecch()
e_.imp.draws = type(e_.imp.draws)(seed=123); f_.imp.draws = type(f_.imp.draws)(seed=123)
dd.split_training(rows=True)
q0 = e_.imp.queued_calls
e_.imp.sample(burnin=3)
assert e_.imp.queued_calls == q0 + 1        # the device queue runs under the split: RCCL sums every batch's partial sums on the stream
dd.split_training(rows=False)
f_.imp.sample(burnin=3)
Fe = np.concatenate([nd.output for nd in e_.all_layer[0]], 1)
Ff = np.concatenate([nd.output for nd in f_.all_layer[0]], 1)
np.testing.assert_allclose(Fe, Ff, rtol=0, atol=1e-9)
both = dd.allgather_objects(Fe.tobytes())
assert both[0] == both[1], 'the ranks ended the row-split I-step with different latents'
dd.split_training(rows=True)
np.random.seed(11); e_._m_step()
dd.split_training(rows=False)
np.random.seed(11); f_._m_step()
np.testing.assert_allclose(hyper(e_), hyper(f_), rtol=1e-4, atol=1e-8)   # (L-BFGS-B stops on a flat objective: see test_training_splits_two_ranks)
print('rows split ok', flush=True)
dd.barrier()
td.destroy_process_group()
print('rank', local, 'ok')
"""


def _two_ranks(tmp_path, extra_env, port):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK='0' if 'TWO_DEV_BACKEND' in extra_env else str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0].decode())
    except subprocess.TimeoutExpired:   # (a rank that died leaves the other waiting in a collective: end both, show what they said)
        for p in procs:
            p.kill()
        outs = [p.communicate()[0].decode()[-3000:] for p in procs]
        raise AssertionError('two-device worker timed out:\n' + '\n----\n'.join(outs))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
        assert 'ok' in o


def test_two_ranks_on_two_devices_over_rccl(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible devices (this pool\'s boxes have one): the first real multi-GPU contact')
    _two_ranks(tmp_path, {}, 29561)


def test_the_two_device_worker_dry_run_on_one_gpu(tmp_path):
    """The SAME worker script with both ranks on GPU 0 and gloo as the backend: what a one-GPU box can check of it -- every
    statement runs, every comparison holds -- so that the two-device test above does not meet its first execution on the
    driver's multi-GPU box.  (Two processes time-slice the device: the per-block-step factorisation, as in bench.py's gloo mode.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    _two_ranks(tmp_path, {'TWO_DEV_BACKEND': 'gloo', 'DGPAMD_POTRF_MODE': '0'}, 29563)
