"""GPU parity: HIP network kernels (through the C-ABI) vs the CPU oracle and the committed goldens.

Tolerances (fp32, stated per SURVEY.md section 8(c)): features/logits <= 2e-5 abs per block-level call at
activations of O(1), positions <= 1e-4 abs; the golden files were produced by the real reference.
"""
import numpy as np
import pytest
import torch

from tests import util as U
from oracle import moldiff_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _two_mol():
    return U.graph_from_sizes([5, 7])


def _inputs_blocks():
    g = U.gold('blocks_full.npz')
    bn, hei, bh, ei, be = _two_mol()
    assert np.array_equal(g['halfedge_index'], hei.numpy())
    tg = torch.from_numpy(g['t'])
    return g, bn, ei, be, U.t32(g['x']), U.t32(g['edge_attr']), U.t32(g['pos']), tg


@pytest.mark.parametrize('i', [0, 3])
def test_node_block_vs_golden(i):
    g, bn, ei, be, x, ea, pos, tg = _inputs_blocks()
    m = U.moldiff('MolDiff', DEV)
    nt = (tg[bn].unsqueeze(-1) / 1000).float()
    out = m.denoiser.node_blocks_with_edge[i](x.to(DEV), ei.to(DEV), ea.to(DEV), nt.to(DEV))
    assert U.maxdiff(out, g[f'nodeblock{i}_out']) < 2e-5


@pytest.mark.parametrize('i', [0, 3])
def test_edge_block_vs_golden(i):
    g, bn, ei, be, x, ea, pos, tg = _inputs_blocks()
    m = U.moldiff('MolDiff', DEV)
    et = (tg[be].unsqueeze(-1) / 1000).float()
    out = m.denoiser.edge_blocks[i](ea.to(DEV), ei.to(DEV), x.to(DEV), et.to(DEV))
    assert U.maxdiff(out, g[f'edgeblock{i}_out']) < 2e-5


@pytest.mark.parametrize('i', [0, 3])
def test_pos_update_vs_golden(i):
    g, bn, ei, be, x, ea, pos, tg = _inputs_blocks()
    m = U.moldiff('MolDiff', DEV)
    et = (tg[be].unsqueeze(-1) / 1000).float()
    rel = pos[ei[0]] - pos[ei[1]]
    dist = torch.norm(rel, dim=-1)
    out = m.denoiser.pos_blocks[i](x.to(DEV), ea.to(DEV), ei.to(DEV), rel.to(DEV), dist.to(DEV), et.to(DEV))
    assert U.maxdiff(out, g[f'posupdate{i}_out']) < 2e-5


@U.both_paths
@pytest.mark.parametrize('tag,sizes', [('n12', [5, 7]), ('n204', None)])
def test_node_edge_net_vs_golden(tag, sizes):
    gd = U.gold('nodeedgenet.npz')
    sizes = gd[f'{tag}_sizes'] if sizes is None else sizes
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, E = len(bn), ei.shape[1]
    r = U.rng(int(gd['input_seed']))
    hn = U.t32(r.standard_normal((N, 256), dtype=np.float32))
    he = U.t32(r.standard_normal((E, 64), dtype=np.float32))
    pos = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 2)
    B = int(bn.max()) + 1
    tg = torch.from_numpy(r.integers(0, 1000, B))
    assert np.array_equal(tg.numpy(), gd[f'{tag}_t'])
    nt, et = (tg[bn].unsqueeze(-1) / 1000).float(), (tg[be].unsqueeze(-1) / 1000).float()
    m = U.moldiff('MolDiff', DEV)
    o = m.denoiser(hn.to(DEV), pos.to(DEV), he.to(DEV), ei.to(DEV), nt.to(DEV), et.to(DEV))
    stride = 1 if tag == 'n12' else 16
    assert U.maxdiff(o[0], gd[f'{tag}_6_h_node']) < 1e-4
    assert U.maxdiff(o[1], gd[f'{tag}_6_pos']) < 1e-4
    assert U.maxdiff(o[2][::stride], gd[f'{tag}_6_h_edge_s{stride}']) < 1e-4


def test_segment_sum_matches_index_add():
    import ctypes
    from moldiff_amd import _lib
    bn, hei, bh, ei, be = U.graph_from_sizes([3, 1, 0, 9, 2])
    N, E = len(bn), ei.shape[1]
    g = _lib.Graph(ei, bn, 5)
    r = U.rng(5)
    for C in (3, 64, 256):
        src = U.t32(r.standard_normal((E, C), dtype=np.float32))
        for by_right in (0, 1):
            out = torch.empty(N, C, device=DEV)
            ws, nb = g.workspace(torch.device(DEV))
            s = src.to(DEV)
            _lib.check(_lib.lib().mdx_segment_sum(g.h, _lib.ptr(s), C, by_right, _lib.ptr(out), ws, nb, _lib.stream()))
            ref = O.seg_sum(src, ei[by_right], N)
            assert U.maxdiff(out, ref) < 1e-5


@U.both_paths
@pytest.mark.parametrize('tval', ['t999', 't500', 't0', 'tmix'])
def test_moldiff_forward_vs_golden(tval):
    g = U.gold('forward.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes(g['sizes'])
    xn = torch.nn.functional.one_hot(torch.from_numpy(g['node_type']), 8).float()
    xh = torch.nn.functional.one_hot(torch.from_numpy(g['halfedge_type']), 6).float()
    pos = U.t32(g['pos'])
    t = torch.from_numpy(g['tmix']) if tval == 'tmix' else torch.full((8,), int(tval[1:]), dtype=torch.long)
    m = U.moldiff('MolDiff', DEV)
    out = m(xn.to(DEV), pos.to(DEV), bn.to(DEV), torch.cat([xh, xh]).to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    assert U.maxdiff(out['pred_node'], g[f'{tval}_pred_node']) < 2e-5
    assert U.maxdiff(out['pred_halfedge'], g[f'{tval}_pred_halfedge']) < 2e-5
    assert U.maxdiff(out['pred_pos'], g[f'{tval}_pred_pos']) < 1e-4
