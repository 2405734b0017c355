"""The C-ABI library loads without a GPU and exports every function include/dgp_amd.h declares;
the ctypes signature table covers exactly that set (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'dgp_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dgpamd_[a-z_0-9]+)\s*\(', src)))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ('dgpamd_kmatrix', 'dgpamd_potrf', 'dgpamd_loglik', 'dgpamd_potri', 'dgpamd_grad_reduce',
                 'dgpamd_gp_predict', 'dgpamd_linkgp_predict', 'dgpamd_nn_ordered', 'dgpamd_vecchia_llik'):
        assert must in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, 'dgp_amd', 'libdgp_amd.so'))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.dgpamd_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.dgpamd_version()
    lib.dgpamd_padded_dim.restype = ctypes.c_int64
    lib.dgpamd_padded_dim.argtypes = [ctypes.c_int64]
    assert lib.dgpamd_padded_dim(2000) == 2048 and lib.dgpamd_padded_dim(64) == 128 and lib.dgpamd_padded_dim(1) == 64


def test_linkgp_workspace_covers_a_launch_of_the_pair_kernels(monkeypatch):
    """dgpamd_linkgp_workspace (a host-side size formula: no GPU) holds the per-tile partial sums of a workspace chunk plus the records of ONE launch
    of the record-based pair kernels (csrc/predict.hip pair_chunk): as many test points as make 32 rounds of workgroups on 512 slots, at least
    256, records <= 4 GiB -- 1024 points at the bench's n = 2000, 256 at cfg3's n = 5000, all of a small call; never less than round 4's 256."""
    for v in ('DGPAMD_JSEP_TCH', 'DGPAMD_PAIR_CHUNK'):
        monkeypatch.delenv(v, raising=False)
    lib = ctypes.CDLL(os.path.join(ROOT, 'dgp_amd', 'libdgp_amd.so'))
    lib.dgpamd_linkgp_workspace.restype = ctypes.c_size_t
    lib.dgpamd_linkgp_workspace.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]

    def parts(n, M, Dw, chunk):
        nb = (n + 63) // 64
        mc = min(-(-M // 32) * 32, 2048)
        return 8 * (nb * (nb + 1) // 2 * mc + min(chunk, mc) * (Dw * nb * 64 * 30 + nb * 64))
    assert lib.dgpamd_linkgp_workspace(2000, 16384, 5) == parts(2000, 16384, 5, 1024)
    assert lib.dgpamd_linkgp_workspace(5000, 100000, 10) == parts(5000, 100000, 10, 256)
    assert lib.dgpamd_linkgp_workspace(130, 70, 3) == parts(130, 70, 3, 96)
    big = lib.dgpamd_linkgp_workspace(700, 100000, 16)      # 66 tiles: 32 rounds would be 7968 points -- the workspace chunk's 2048 and the 4-GiB cap bound it
    assert parts(700, 100000, 16, 256) <= big <= parts(700, 100000, 16, 2048) and big - 8 * 66 * 2048 <= 4 << 30


def test_ctypes_table_matches_header():
    from dgp_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert not _lib.MISSING


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product refuses to run instead of computing on the CPU."""
    import pytest
    import torch
    from dgp_amd.ops import Engine, DgpAmdError
    if torch.cuda.is_available():
        pytest.skip('device present')
    with pytest.raises(DgpAmdError):
        Engine(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'dgp_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.replace('the oracle', ''), f
