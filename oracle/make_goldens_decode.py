"""Golden vectors for the harness consumers of the hot path's outputs:
`seperate_outputs` (utils/sample.py:4-30) and `FeaturizeMol.decode_output` (utils/transforms.py:65-122).
BUILD-CONTAINER ONLY (needs /root/reference).

utils/transforms.py cannot be imported here (it pulls torch_geometric.transforms, lmdb, rdkit), so the
reference's OWN `decode_output` source is lifted at run time: the method's AST is extracted from the file
where it lies, compiled, and bound to a minimal stand-in for `self` holding the four attributes it reads.
Nothing of it is copied into this repository; only the inputs/outputs are saved (tests/golden/decode.npz).
"""
import ast
import os
import sys
import types

import numpy as np
from scipy.special import softmax

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')
from oracle import moldiff_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
ATOMIC = [6, 7, 8, 9, 15, 16, 17]


def reference_decode():
    src = open('/root/reference/utils/transforms.py').read()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == 'decode_output':
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {'np': np, 'softmax': softmax}
    exec(compile(mod, '/root/reference/utils/transforms.py', 'exec'), ns)
    self = types.SimpleNamespace(num_element=7, num_bond_types=4, num_edge_types=6,
                                 nodetype_to_ele={i: e for i, e in enumerate(ATOMIC)})
    return lambda **kw: ns['decode_output'](self, **kw)


def main():
    from utils.sample import seperate_outputs  # the reference's own function (pure numpy)
    ref_decode = reference_decode()
    g = np.random.Generator(np.random.PCG64(41))
    sizes = [6, 2, 9, 4]
    bn = np.repeat(np.arange(len(sizes)), sizes)
    hei, bh, off = [], [], 0
    for i, n in enumerate(sizes):
        iu, ju = np.triu_indices(n, k=1)
        hei.append(np.stack([iu + off, ju + off]))
        bh.append(np.full(len(iu), i))
        off += n
    hei, bh = np.concatenate(hei, 1), np.concatenate(bh)
    N, Eh = len(bn), len(bh)
    pred_node = (g.standard_normal((N, 8)) * 2).astype(np.float32)
    pred_node[[2, 6, 11], 7] += 8.0      # force some mask-type atoms (class 7) -> dropped, bonds re-indexed
    pred_pos = g.standard_normal((N, 3)).astype(np.float32)
    pred_half = (g.standard_normal((Eh, 6)) * 2).astype(np.float32)
    pred_half[::5, 5] += 6.0             # some mask-type bonds (class 5) -> not bonds
    outputs = {'pred': [pred_node, pred_pos, pred_half],
               'traj': [pred_node[None], pred_pos[None], pred_half[None]]}
    sep = seperate_outputs(outputs, len(sizes), bn, hei, bh)
    osep = O.separate_outputs(outputs['pred'], len(sizes), bn, hei, bh)
    save = {'sizes': np.array(sizes), 'pred_node': pred_node, 'pred_pos': pred_pos, 'pred_halfedge': pred_half}
    for i, (a, b) in enumerate(zip(sep, osep)):
        for x, y in zip(a['pred'], b['pred']):
            assert np.array_equal(x, y)
        assert np.array_equal(a['halfedge_index'], b['halfedge_index'])
        ref = ref_decode(pred_node=a['pred'][0], pred_pos=a['pred'][1], pred_halfedge=a['pred'][2],
                         halfedge_index=a['halfedge_index'])
        mine = O.decode_output(b['pred'][0], b['pred'][1], b['pred'][2], b['halfedge_index'], ATOMIC, 4)
        for k in ref:
            assert np.array_equal(np.asarray(ref[k]).shape, np.asarray(mine[k]).shape), (i, k)
            if np.asarray(ref[k]).dtype.kind == 'f':
                assert np.abs(np.asarray(ref[k]) - mine[k]).max(initial=0) < 1e-6, (i, k)
            else:
                assert np.array_equal(ref[k], mine[k]), (i, k)
            save[f'mol{i}_{k}'] = np.asarray(ref[k])
        save[f'mol{i}_halfedge_index'] = a['halfedge_index']
    np.savez_compressed(os.path.join(OUT, 'decode.npz'), **save)
    print('decode.npz written;', {k: v.shape for k, v in save.items() if k.startswith('mol2')})


if __name__ == '__main__':
    main()
