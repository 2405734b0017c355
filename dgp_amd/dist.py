"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).  The SI path shards embarrassingly over imputation
samples; the only data-path collective of prediction is the sum of the two predictive-moment arrays
(emulation.py:846-847 aggregated across ranks).

Training one model on several GPUs (every rank holds the whole model and draws the same numbers: same seed) has two
natural splits, both off by default and switched on with split_training():
  * Vecchia rows (vecchia.py:164-242: the `prange` over rows): a rank evaluates the rows vecchia_rows() gives it and
    the (quad, logdet, gradient) sums are all-reduced -- a few doubles per objective evaluation;
  * M-step nodes (dgp.py:1455-1467: pool.map over the nodes of a layer): node i is fitted by rank i mod world, the
    fitted (scale, lengthscales, nugget) are all-gathered once per M-step."""
import os

import torch
import torch.distributed as td


def _forced():
    """DGPAMD_DIST_FORCE=1: treat an initialised process group of ONE rank as active, so that every collective of the library
    really goes through the backend (RCCL on a one-GPU box: the first-contact test of the `nccl` path)."""
    return os.environ.get('DGPAMD_DIST_FORCE') == '1'


def is_active():
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _forced())


def _dev(device=None):
    """Where a collective's tensor lives: the device under nccl / RCCL (it reduces device memory only), the host under gloo."""
    if td.get_backend() == 'nccl':
        return device if device is not None else torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def world():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def share(total, r, w):
    """Number of the `total` imputations (or other independent units) owned by rank r of w."""
    return total // w + (1 if r < total % w else 0)


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    if td.is_initialized() or (int(os.environ.get('WORLD_SIZE', '1')) <= 1 and not _forced()):
        return
    if _forced():   # a group of one needs no launcher: fill in what torch.distributed.run would have set
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('LOCAL_RANK', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    import datetime
    td.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=int(os.environ.get('DGPAMD_DIST_TIMEOUT', '300'))))


def allreduce_sum(*tensors):
    """In-place sum over ranks of each tensor.  nccl / RCCL reduces device tensors, gloo host tensors: a tensor on the other
    side makes the round trip through a copy (under gloo with several ranks on one GPU: the functional checks)."""
    if not is_active():
        return
    want = _dev().type
    for t in tensors:
        if t.device.type == want:
            td.all_reduce(t, op=td.ReduceOp.SUM)
        else:
            h = t.to(_dev())
            td.all_reduce(h, op=td.ReduceOp.SUM)
            t.copy_(h)


def row_range(total, r, w):
    """Rows [lo, hi) of `total` independent rows (test points) owned by rank r of w: equal chunks, the last ones shorter."""
    chunk = -(-total // w)
    return min(total, r * chunk), min(total, (r + 1) * chunk)


def allgather_rows(local, total, device=None):
    """Concatenate over ranks the row blocks of `row_range` (numpy array, rows first): every rank gets all `total` rows.
    Blocks are padded to the common chunk so that one equal-size all-gather does it (emulation.py:607-613 concatenated
    the pool's chunks the same way)."""
    import numpy as np
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not is_active():
        return local
    w = world()
    chunk = -(-total // w)
    pad = np.zeros((chunk,) + local.shape[1:])
    pad[:len(local)] = local
    t = torch.from_numpy(pad).to(_dev(device))
    parts = [torch.empty_like(t) for _ in range(w)]
    td.all_gather(parts, t)
    return np.concatenate([p.cpu().numpy() for p in parts], 0)[:total]


def allreduce_max_scalar(value, device=None):
    """max over ranks of a python float (bench.py: the slowest rank's time)."""
    if not is_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev(device))
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def broadcast_int(value, src=0, device=None):
    """A python int (up to 128 bits: a SeedSequence entropy) from rank `src` to every rank, as four int64 words -- no pickling,
    a device tensor under nccl / RCCL."""
    if not is_active():
        return int(value)
    v = int(value)
    words = [(v >> (32 * i)) & 0xffffffff for i in range(4)]
    t = torch.tensor(words, dtype=torch.int64, device=_dev(device))
    td.broadcast(t, src=src)
    return sum(int(w) << (32 * i) for i, w in enumerate(t.tolist()))


def barrier():
    if not is_active():
        return
    if td.get_backend() == 'nccl':   # name the device: RCCL otherwise guesses it from the rank
        td.barrier(device_ids=[torch.cuda.current_device()])
    else:
        td.barrier()


_SPLIT = {'rows': False, 'nodes': False}


def split_training(rows=None, nodes=None):
    """Switch the Vecchia-row split OR the M-step-node split on or off (they act only in an initialised process
    group of more than one rank).  Returns the current settings.
    The two do not compose -- with the node split every rank fits different nodes, so the row sums of one rank's node
    would be added to another rank's -- and asking for both raises ValueError (nothing is changed then)."""
    new = dict(_SPLIT)
    if rows is not None:
        new['rows'] = bool(rows)
    if nodes is not None:
        new['nodes'] = bool(nodes)
    if new['rows'] and new['nodes']:
        raise ValueError('split_training: the Vecchia-row split and the M-step-node split cannot be combined '
                         '(switch one off first: split_training(rows=False, nodes=True) or (rows=True, nodes=False))')
    _SPLIT.update(new)
    return dict(_SPLIT)


def rows_split():
    return _SPLIT['rows'] and is_active()


def nodes_split():
    return _SPLIT['nodes'] and is_active()


def vecchia_rows(n):
    """Rows [lo, hi) of the n rows of a Vecchia likelihood this rank evaluates (all rows without the split)."""
    return row_range(n, rank(), world()) if rows_split() else (0, n)


def allreduce_sum_vector(t):
    """Sum over ranks of a small device (or host) tensor; returns a tensor on the same device.  gloo reduces host
    tensors only, so under gloo the few doubles make a round trip through the host."""
    if not is_active():
        return t
    if td.get_backend() == 'nccl' or not t.is_cuda:
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t
    h = t.cpu()
    td.all_reduce(h, op=td.ReduceOp.SUM)
    return h.to(t.device)


def allgather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (pickles through the host: diagnostics and tests only -- the
    training and prediction paths exchange fixed-size float64 vectors, allgather_vector)."""
    if not is_active():
        return [obj]
    out = [None] * world()
    td.all_gather_object(out, obj)
    return out


def allgather_vector(vec, device=None):
    """(world, len(vec)) float64 array: every rank's fixed-length vector, on every rank -- ONE equal-size all-gather of
    doubles (no pickling; a device tensor under nccl/RCCL, a host tensor under gloo)."""
    import numpy as np
    vec = np.ascontiguousarray(vec, dtype=np.float64).ravel()
    if not is_active():
        return vec[None, :].copy()
    on_dev = td.get_backend() == 'nccl'
    t = torch.from_numpy(vec)
    if on_dev:
        t = t.to(_dev(device))
    out = torch.empty((world(), len(vec)), dtype=torch.float64, device=t.device)
    td.all_gather_into_tensor(out, t) if on_dev else td.all_gather(list(out.unbind(0)), t)
    return out.cpu().numpy()
