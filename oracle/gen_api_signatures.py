"""Test infrastructure (not product code): writes tests/golden/api_signatures.json -- for every public class the
reference exports (dgpsi/__init__.py) the names of its public methods and of their parameters, read from the
reference's sources with `ast` (names only, no source text).  tests/test_host_logic.py checks that the classes of
dgp_amd accept the same calls.  Run in the build container:  python oracle/gen_api_signatures.py"""
import ast
import json
import os

REF = '/root/reference/dgpsi'
CLASSES = {'dgp.py': ['dgp'], 'gp.py': ['gp'], 'emulation.py': ['emulator'], 'kernel_class.py': ['kernel'],
           'linkgp.py': ['container', 'lgp'], 'synthetic.py': ['path'],
           'likelihood_class.py': ['Poisson', 'Hetero', 'NegBin', 'Categorical', 'ZIP', 'ZINB']}
FUNCS = {'kernel_class.py': ['combine'], 'utils.py': ['write', 'read', 'summary', 'nb_seed', 'set_thread', 'get_thread']}
# the njit "operator API" (SURVEY.md 8(b)): module -> functions whose ORDERED parameter lists dgp_amd.functions /
# dgp_amd.vecchia reproduce (positional calls must keep working)
OPERATORS = {'functions.py': ['gp', 'link_gp', 'fmvn', 'update_f'],
             'vecchia.py': ['get_pred_nn', 'nn', 'forward_solve_sp', 'vecchia_llik', 'vecchia_nllik', 'L_matrix', 'gp_vecch',
                            'link_gp_vecch']}


def params(fn):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args + a.kwonlyargs]
    return [n for n in names if n != 'self']


def main():
    out = {'classes': {}, 'functions': {}}
    for f, names in CLASSES.items():
        tree = ast.parse(open(os.path.join(REF, f)).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name in names:
                out['classes'][node.name] = {m.name: params(m) for m in node.body if isinstance(m, ast.FunctionDef)
                                             and (not m.name.startswith('_') or m.name == '__init__')}
    for f, names in FUNCS.items():
        tree = ast.parse(open(os.path.join(REF, f)).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                out['functions'][node.name] = params(node)
    out['operators'] = {}
    for f, names in OPERATORS.items():
        tree = ast.parse(open(os.path.join(REF, f)).read())
        out['operators'][f[:-3]] = {node.name: params(node) for node in tree.body if isinstance(node, ast.FunctionDef) and node.name in names}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'api_signatures.json')
    with open(dst, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print('wrote', dst, sum(len(v) for v in out['classes'].values()), 'methods')


if __name__ == '__main__':
    main()
