"""The driver's N > 1 bench flow, pinned on the one-GPU box: `bench.py --gpus 2` with both ranks on GPU 0
(gloo, per-block-step factorisation) and the one-rank RCCL group -- launcher, barriers, max-over-ranks timing, sharded
prediction, the node-split leg and the two strong-scaling legs must end with ONE JSON line as the last line of stdout."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines, r.stderr[-2000:]
    return json.loads(lines[-1])   # the result must be the LAST line of stdout (the driver parses that)


def _finite(x):
    return isinstance(x, (int, float)) and math.isfinite(x)


def test_bench_two_ranks_on_one_gpu():
    d = _run(['--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--predict-points', '256', '--predict-seconds', '0.1',
              '--sustained-steps', '2', '--prof-kernel', 'none'])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and _finite(d['value']) and d['value'] > 0
    assert d['distributed']['world_size'] == 2 and d['distributed']['backend'] == 'gloo' and len(d['distributed']['ranks']) == 2
    assert d['distributed']['ranks_share_a_device'] is True
    assert d['predict']['finite'] and d['predict']['imputations'] == 20
    assert _finite(d['mstep_nodes_split']['si_it_per_s']), d['mstep_nodes_split']
    s3, s4 = d['strong_scaling']['cfg3_predict_imputations_sharded'], d['strong_scaling']['cfg4_train_rows_split']
    assert s3.get('finite') is True and _finite(s3['point_imputations_per_s']), s3
    assert _finite(s4['si_it_per_s']) and s4['si_it_per_s'] > 0, s4
    assert 'linear in N by construction' in d['config']['parallelism']
    # the line itself names the columns that answer north_star's scaling question (the replica `value` does not)
    assert 'predict.point_imputations_per_s' in d['scaling_metric']['columns'] and 'linear in N' in d['scaling_metric']['note']
    ss = d['step_split']
    assert abs(ss['sum_ms'] - ss['ms_per_step']) <= 0.03 * ss['ms_per_step'], ss
    assert _finite(d['sustained_it_per_s'])
    assert d['cpu_baseline'] is None   # (rank 0 at N = 1 only)


def test_bench_one_rank_through_rccl():
    d = _run(['--gpus', '1', '--backend', 'nccl', '--steps', '2', '--warmup', '1', '--predict-points', '256', '--predict-seconds', '0.1',
              '--sustained-steps', '2', '--prof-kernel', 'none', '--no-cpu-baseline'], env_extra={'DGPAMD_DIST_FORCE': '1'})
    assert d['n_gpus'] == 1 and _finite(d['value']) and d['scaling_metric'] is None
    assert d['distributed']['world_size'] == 1 and d['distributed']['backend'] == 'nccl'
    assert d['predict']['finite']
