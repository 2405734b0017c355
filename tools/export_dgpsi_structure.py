#!/usr/bin/env python3
"""Export a structure trained with dgpsi to the arrays-only .npz that dgp_amd.load_structure reads.

Runs where dgpsi is installed; needs numpy only (no dgp_amd, no GPU).  Either import `export` and call it on
`model.estimate()` / `emulator.all_layer`, or point the script at a file written by dgpsi.write():

    python export_dgpsi_structure.py trained_emulator.pkl structure.npz

The file format is the one of dgp_amd/utils.py:save_structure (per node: name, prior, flags [scale_est, nugget_est,
vecchia], m and the arrays length / scale / nugget / input / output / global_input / input_dim / connect / prior_coef /
bds / rep / para_path; a Hetero likelihood node: input / output / input_dim / rep).  No code is pickled."""
import sys
import numpy as np

NODE_ARRAYS = ('length', 'scale', 'nugget', 'input', 'output', 'global_input', 'input_dim', 'connect', 'prior_coef', 'bds',
               'rep', 'para_path')


def export(all_layer, npz_file):
    out = {'n_layer': np.array(len(all_layer))}
    for l, layer in enumerate(all_layer):
        out['l%d_n' % l] = np.array(len(layer))
        for k, nd in enumerate(layer):
            p = 'l%d_k%d_' % (l, k)
            if getattr(nd, 'type', 'gp') != 'gp':
                out[p + 'likelihood'] = np.array(str(nd.name))
                if nd.name == 'Categorical':
                    out[p + 'cat_num_classes'] = np.array(int(nd.num_classes))
                    out[p + 'cat_link'] = np.array(str(nd.link))
                    out[p + 'cat_eps'] = np.array(float(nd.robustmax_eps))
                    if nd.class_encoder is not None:
                        out[p + 'cat_classes'] = np.asarray(nd.class_encoder.classes_)
                arrays = ('input', 'output', 'input_dim', 'rep')
            else:
                out[p + 'name'] = np.array(str(nd.name))
                out[p + 'prior_name'] = np.array('' if nd.prior_name is None else str(nd.prior_name))
                out[p + 'flags'] = np.array([bool(nd.scale_est), bool(nd.nugget_est), bool(getattr(nd, 'vecch', False))])
                out[p + 'm'] = np.array(-1 if getattr(nd, 'm', None) is None else int(nd.m))
                arrays = NODE_ARRAYS
            for a in arrays:
                v = getattr(nd, a, None)
                if v is not None:
                    out[p + a] = np.asarray(v).copy()
    np.savez_compressed(npz_file, **out)


if __name__ == '__main__':
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    import dill
    with open(sys.argv[1], 'rb') as f:
        obj = dill.load(f)
    export(obj.all_layer if hasattr(obj, 'all_layer') else obj, sys.argv[2])
