"""ctypes binding of libdgp_amd.so (include/dgp_amd.h).

The HIP library IS the product: there is no CPU fallback.  If the shared
library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DGPAMD_LIB: another build of the same library, for same-box A/B timing of kernel variants by the tools)
LIB_PATH = os.environ.get('DGPAMD_LIB') or os.path.join(_HERE, 'libdgp_amd.so')

OK, NOT_PD, BAD_ARG, HIP_ERROR = 0, 1, 2, 3
SEXP, MATERN25 = 0, 1
MAXD, MAXB = 64, 64
KIND = {'sexp': SEXP, 'matern2.5': MATERN25}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'dgp_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(or `make -C dgp_amd/csrc`).  dgp_amd has no CPU fallback.' % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_d = C.c_double
_z = C.c_size_t

class Node(C.Structure):
    """dgpamd_node (include/dgp_amd.h)."""
    _fields_ = [('kind', C.c_int), ('Dl', C.c_int), ('Dg', C.c_int), ('nlen', C.c_int), ('nugget_est', C.c_int),
                ('reserved', C.c_int), ('ldloc', C.c_int64), ('Xloc', C.c_void_p), ('colmap', C.c_void_p),
                ('Xglob', C.c_void_p), ('length', C.c_void_p), ('nugget', C.c_double), ('W', C.c_void_p),
                ('y', C.c_void_p), ('vecch_ord', C.c_void_p), ('vecch_nn', C.c_void_p), ('vecch_nd', C.c_void_p),
                ('vecch_y', C.c_void_p), ('vecch_m', C.c_int), ('reserved2', C.c_int),
                ('vecch_row0', C.c_int64), ('vecch_rows', C.c_int64),
                ('lik_kind', C.c_int), ('lik_classes', C.c_int), ('lik_nobs', C.c_int64), ('lik_rep', C.c_void_p),
                ('lik_par', C.c_double)]


SIGNATURES = {
    'dgpamd_create': (_i, [_i, _p, C.POINTER(_p)]),
    'dgpamd_destroy': (_i, [_p]),
    'dgpamd_last_error': (C.c_char_p, [_p]),
    'dgpamd_sync': (_i, [_p]),
    'dgpamd_version': (C.c_char_p, []),
    'dgpamd_padded_dim': (_l, [_l]),
    'dgpamd_event_create': (_i, [_p, C.POINTER(_p)]),
    'dgpamd_event_record': (_i, [_p, _p]),
    'dgpamd_event_elapsed_ms': (_i, [_p, _p, _p, C.POINTER(C.c_float)]),
    'dgpamd_event_destroy': (_i, [_p, _p]),
    'dgpamd_gemv': (_i, [_p, _l, _l, _p, _l, _p, _p]),
    'dgpamd_fetch': (_i, [_p, _p, _p, _z]),
    'dgpamd_fetch2': (_i, [_p, _p, _z, _p, _z, _p]),
    'dgpamd_set_reduce_hook': (_i, [_p, _p, _p]),
    'dgpamd_post': (_i, [_p, _p, _z, _i]),
    'dgpamd_collect': (_i, [_p, _i, _p, _z]),
    'dgpamd_set_graphs': (_i, [_p, _i]),
    'dgpamd_set_linkgp_direct': (_i, [_p, _i]),
    'dgpamd_set_potrf_mode': (_i, [_p, _i]),
    'dgpamd_debug_trace': (_i, [_p, _p]),
    'dgpamd_debug_tasklog': (_i, [_p, _p, C.c_longlong]),
    'dgpamd_debug_mega_table': (_i, [_p, _l, _i, _i, _p, _l]),
    'dgpamd_prof_enable': (_i, [_p, _i]),
    'dgpamd_prof_event_overhead_us': (_i, [_p, C.POINTER(_d)]),
    'dgpamd_prof_collect': (_i, [_p, C.POINTER(_l), C.POINTER(_d), C.POINTER(_d)]),
    'dgpamd_kmatrix': (_i, [_p, _i, _l, _p, _l, _l, _p, _i, _p, _i, _p, _i, _d, _p, _p, _l, _l, _i, _p, _l, _l, _i, _i]),
    'dgpamd_potrf_workspace': (_z, [_l, _i]),
    'dgpamd_potrf': (_i, [_p, _l, _p, _l, _i, _p, _p, _p]),
    'dgpamd_aug_quad': (_i, [_p, _l, _p, _l, _i, _i, _p]),
    'dgpamd_loglik_finish': (_i, [_p, _l, _p, _l, _i, _p, _d, _p]),
    'dgpamd_loglik': (_i, [_p, _i, _l, _p, _l, _l, _p, _i, _p, _i, _p, _i, _d, _p, _d, _p, _p, _l, _i, _p, _p, _p]),
    'dgpamd_trmv_lower': (_i, [_p, _l, _p, _l, _p, _p, _p, _i]),
    'dgpamd_lik_workspace': (_z, [_i]),
    'dgpamd_lik_loglik': (_i, [_p, _p, _l, _i, _p, _l, _i, _p, _p]),
    'dgpamd_ess_propose': (_i, [_p, _l, _i, _p, _p, _p, _i, _p]),
    'dgpamd_potri': (_i, [_p, _l, _p, _p, _i, _p]),
    'dgpamd_potri_batched': (_i, [_p, _l, _p, _p, _l, _i, _i, _p]),
    'dgpamd_grad_workspace': (_z, [_l, _i]),
    'dgpamd_grad_reduce': (_i, [_p, _i, _l, _p, _l, _p, _i, _p, _i, _p, _i, _d, _p, _i, _p, _p, _p]),
    'dgpamd_ess_update': (_i, [_p, _l, _i, _p, _p, _p, _d, _d, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    'dgpamd_ess_queue_scratch': (_z, []),
    'dgpamd_ess_queue': (_i, [_p, _l, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'dgpamd_ess_queue_vwork': (_z, [_l, _i, _i]),
    'dgpamd_ess_queue_note_info': (_i, [_p, _p, _p, _i]),
    'dgpamd_llik_batch': (_i, [_p, _l, _i, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l]),
    'dgpamd_llik_batch_launch': (_i, [_p, _l, _i, _p, _p, _p, _p, _l, _p, _p, _p, _l]),
    'dgpamd_llik_batch_wait': (_i, [_p, _p]),
    'dgpamd_potrf_inv': (_i, [_p, _l, _p, _p, _p, _l, _i, _p, _p, _p]),
    'dgpamd_gp_workspace': (_z, [_l, _l]),
    'dgpamd_gp_predict': (_i, [_p, _i, _l, _l, _i, _p, _p, _p, _i, _p, _l, _p, _i, _d, _d, _p, _p, _p]),
    'dgpamd_linkgp_workspace': (_z, [_l, _l, _i]),
    'dgpamd_linkgp_predict': (_i, [_p, _i, _l, _l, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _l, _p, _d, _d, _p, _p, _p]),
    'dgpamd_linkgp_loo': (_i, [_p, _i, _l, _l, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _l, _p, _p, _d, _d, _p, _p, _p]),
    'dgpamd_moments_accumulate': (_i, [_p, _l, _p, _p, _p, _p]),
    'dgpamd_moments_finalize': (_i, [_p, _l, _d, _p, _p]),
    'dgpamd_nn_ordered': (_i, [_p, _l, _i, _p, _i, _p]),
    'dgpamd_nn_query': (_i, [_p, _l, _l, _i, _p, _p, _i, _p]),
    'dgpamd_debug_poison_lds': (_i, [_p]),
    'dgpamd_vecchia_llik': (_i, [_p, _i, _l, _i, _i, _p, _p, _p, _p, _i, _d, _p, _p]),
    'dgpamd_vecchia_llik_batch': (_i, [_p, _i, _l, _i, _i, _p, _l, _i, _p, _p, _p, _i, _d, _p, _p]),
    'dgpamd_vecchia_nllik': (_i, [_p, _i, _l, _i, _i, _p, _p, _p, _p, _i, _d, _p, _i, _p]),
    'dgpamd_vecchia_lmatrix': (_i, [_p, _i, _l, _i, _i, _p, _p, _p, _i, _d, _p]),
    'dgpamd_vecchia_spsolve': (_i, [_p, _l, _i, _p, _p, _d, _p, _p]),
    'dgpamd_vecchia_spsolve_batch': (_i, [_p, _l, _i, _i, _i, _p, _p, _p, _p, _p]),
    'dgpamd_vecchia_levels_bytes': (_z, [_l, _i]),
    'dgpamd_vecchia_levels': (_i, [_p, _l, _i, _i, _p, _p]),
    'dgpamd_vecchia_spsolve_levels': (_i, [_p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    'dgpamd_vecchia_het_rows': (_i, [_p, _i, _l, _i, _i, _p, _p, _p, _i, _d, _p, _p, _p, _p, _p, _p]),
    'dgpamd_vecchia_gp': (_i, [_p, _i, _l, _l, _i, _i, _p, _p, _p, _p, _d, _p, _i, _d, _p, _p, _p]),
    'dgpamd_vecchia_linkgp': (_i, [_p, _i, _l, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _d, _p, _i, _d, _p, _p, _p]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _f = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _f.restype = _res
    _f.argtypes = _args

if MISSING:
    raise ImportError('dgp_amd: libdgp_amd.so lacks symbols declared in include/dgp_amd.h: %s' % ', '.join(MISSING))
