"""Where does the guidance increment of the split path differ from the exact path's?  (development diagnostic, GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util as U  # noqa: E402
from tests.test_gpu_fullsize import _workload, _state, _f64, B  # noqa: E402
from oracle import moldiff_oracle as O  # noqa: E402
from moldiff_amd import _lib  # noqa: E402

DEV = 'cuda:0'
ph, sizes = _workload('MolDiff')
bp = U.bondpred(DEV)
Pb = U.params(U.bondpred())
st = _state(ph, 61)
bn, hei, bh = ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge']
ei, be = torch.cat([hei, hei.flip(0)], 1), torch.cat([bh, bh])
t = torch.full((B,), 500, dtype=torch.long)
torch.set_num_threads(min(32, os.cpu_count() or 1))
d64, lg64 = O.guidance_delta(_f64(Pb), U.CFGB, st['h_node'].double(), st['pos'].double(), bn, ei, be, t, 1e-4)
res = {}
for path in ('exact_f32', 'split_f16'):
    with _lib.default_matrix_path(path):
        pos_in = st['pos'].to(DEV).requires_grad_(True)
        with torch.enable_grad():
            lg = bp(st['h_node'].to(DEV), pos_in, bn.to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
            u = torch.sigmoid(-torch.logsumexp(lg, -1)).log().sum()
            d = -torch.autograd.grad(u, pos_in)[0] * 1e-4
        res[path] = (d.cpu().double(), lg.detach().cpu().double())
for path, (d, lg) in res.items():
    e = (d - d64).abs().max(dim=1).values
    print(f'{path}: logits max err {float((lg - lg64).abs().max()):.3e} rms {float((lg - lg64).pow(2).mean().sqrt()):.3e};  '
          f'delta max err {float(e.max()):.3e} rms {float((d - d64).pow(2).mean().sqrt()):.3e}  (scale {float(d64.abs().max()):.3e})')
es = (res['split_f16'][0] - d64).abs().max(dim=1).values
ee = (res['exact_f32'][0] - d64).abs().max(dim=1).values
top = torch.argsort(es, descending=True)[:8]
pos = st['pos'].double()
for a in top.tolist():
    mol = int(bn[a])
    idx = (bn == mol).nonzero().squeeze(-1)
    dd = (pos[idx] - pos[a]).norm(dim=1)
    dd = dd[dd > 0]
    print(f'  atom {a} (molecule {mol}, n = {len(idx)}): split err {float(es[a]):.3e}  exact err {float(ee[a]):.3e}  |delta| {float(d64[a].abs().max()):.3e}  '
          f'nearest neighbour {float(dd.min()):.3f}')
print('per-molecule max error, worst 5 (split):')
mol_err = torch.zeros(B, dtype=torch.double).scatter_reduce(0, bn, es, 'amax')
mol_ex = torch.zeros(B, dtype=torch.double).scatter_reduce(0, bn, ee, 'amax')
for mth in torch.argsort(mol_err, descending=True)[:5].tolist():
    print(f'  molecule {mth} (n = {int(sizes[mth])}): split {float(mol_err[mth]):.3e} exact {float(mol_ex[mth]):.3e}')
print('median per-atom error: split %.3e exact %.3e' % (float(es.median()), float(ee.median())))
