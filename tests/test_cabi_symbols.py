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
