"""Accuracy of the two matrix paths against the float64 oracle, on the golden forward batch (8 molecules, N = 204) and on the
full-size batch (256 molecules): max |HIP - fp64| for the exact fp32 path, the split float16 path, and the fp32 CPU oracle itself.
    python tools/split_accuracy.py            (GPU box; prints a table, kept under profiles/)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util as U  # noqa: E402
from oracle import moldiff_oracle as O  # noqa: E402
from moldiff_amd import _lib  # noqa: E402

DEV = 'cuda:0'


def f64(P):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}


def run(kind, sizes, seed, tval):
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, Eh, B = len(bn), len(bh), int(bn.max()) + 1
    r = U.rng(seed)
    xn = F.one_hot(torch.from_numpy(r.integers(0, 8, N)), 8).float()
    xh = F.one_hot(torch.from_numpy(r.integers(0, 6, Eh)), 6).float()
    pos = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 2.5)
    t = torch.full((B,), tval, dtype=torch.long)
    m = U.moldiff(kind, DEV)
    P = U.params(U.moldiff(kind))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        o32 = O.moldiff_forward(P, U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
        o64 = O.moldiff_forward(f64(P), U.CFG, xn.double(), pos.double(), bn, torch.cat([xh, xh]).double(), ei, be, t)
    args = [a.to(DEV) for a in (xn, pos, bn, torch.cat([xh, xh]), ei, be, t)]
    res = {}
    for path in ('exact_f32', 'split_f16'):
        with _lib.default_matrix_path(path), torch.no_grad():
            res[path] = {k: v.cpu() for k, v in m(*args).items()}
    print(f'{kind}: {B} molecules, N = {N}, E = {2 * Eh}, t = {tval}    max |x - fp64| (rms)')
    print(f'  {"quantity":14s} {"oracle fp32 (CPU)":>22s} {"HIP exact fp32":>22s} {"HIP split f16":>22s}   scale')
    for k in ('pred_node', 'pred_pos', 'pred_halfedge'):
        ref = o64[k]
        row = []
        for x in (o32[k], res['exact_f32'][k], res['split_f16'][k]):
            d = (x.double() - ref)
            row.append(f'{float(d.abs().max()):.3e} ({float(d.pow(2).mean().sqrt()):.2e})')
        print(f'  {k:14s} {row[0]:>22s} {row[1]:>22s} {row[2]:>22s}   {float(ref.abs().max()):.2f}')


if __name__ == '__main__':
    g = U.gold('forward.npz')
    run('MolDiff', g['sizes'], 3, 500)
    run('MolDiff', g['sizes'], 4, 20)
    np.random.seed(2920)
    sizes = np.random.normal(24.923464980477522, 5.516291901819105, size=256).astype('int64')
    run('MolDiff_simple', sizes, 5, 700)
