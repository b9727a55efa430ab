"""GPU: edge cases of the packed-graph path -- ragged / degenerate molecules, empty inputs, molecules larger than a
kernel tile, and graphs that are NOT the fully-connected molecule layout (the kernels only assume a CSR)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from moldiff_amd import _lib
from oracle import moldiff_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _fwd_inputs(sizes, seed):
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, Eh = len(bn), len(bh)
    r = U.rng(seed)
    xn = F.one_hot(torch.from_numpy(r.integers(0, 8, N)), 8).float()
    xh = F.one_hot(torch.from_numpy(r.integers(0, 6, Eh)), 6).float()
    pos = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 2.5)
    t = torch.from_numpy(r.integers(0, 1000, len(sizes)))
    return bn, hei, bh, ei, be, xn, xh, pos, t


@U.both_paths
@pytest.mark.parametrize('sizes', [[1, 7, 2, 1, 0, 5], [2, 2], [3]])
def test_forward_with_degenerate_molecules_matches_oracle(sizes):
    bn, hei, bh, ei, be, xn, xh, pos, t = _fwd_inputs(sizes, 3)
    m = U.moldiff('MolDiff', DEV)
    out = m(xn.to(DEV), pos.to(DEV), bn.to(DEV), torch.cat([xh, xh]).to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    with torch.no_grad():
        ref = O.moldiff_forward(U.params(m), U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
    assert U.maxdiff(out['pred_node'], ref['pred_node']) < 2e-5
    assert U.maxdiff(out['pred_halfedge'], ref['pred_halfedge']) < 2e-5
    assert U.maxdiff(out['pred_pos'], ref['pred_pos']) < 1e-4


@U.both_paths
def test_molecule_larger_than_a_tile_matches_oracle():
    """n = 90: every node's run of 89 outgoing edges spans several 48-edge tiles; n = 64 is the old 'max'."""
    bn, hei, bh, ei, be, xn, xh, pos, t = _fwd_inputs([90, 64], 4)
    pos = pos * 3  # keep atoms apart (1/d terms)
    m = U.moldiff('MolDiff', DEV)
    out = m(xn.to(DEV), pos.to(DEV), bn.to(DEV), torch.cat([xh, xh]).to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    with torch.no_grad():
        ref = O.moldiff_forward(U.params(m), U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
    assert U.maxdiff(out['pred_node'], ref['pred_node']) < 5e-5
    assert U.maxdiff(out['pred_halfedge'], ref['pred_halfedge']) < 5e-5
    assert U.maxdiff(out['pred_pos'], ref['pred_pos']) < 2e-4


@U.both_paths
def test_empty_batch_and_single_atom_batch_do_not_crash():
    m = U.moldiff('MolDiff_simple', DEV)
    for sizes in ([], [1], [1, 1, 1]):
        bn, hei, bh, ei, be = U.graph_from_sizes(sizes, DEV)
        sm = m.sampler(len(sizes), bn, hei, bh, seed=1)
        sm.init()
        sm.step(0)
        sm.step(1)
        torch.cuda.synchronize()
        st = sm.state()
        assert st['pos'].shape == (len(bn), 3) and torch.isfinite(st['pos']).all()
        assert st['h_halfedge'].shape == (0, 6)


@U.both_paths
def test_general_graph_not_molecule_layout():
    """NodeEdgeNet on an arbitrary directed graph given in shuffled edge order (ring + chords, asymmetric degrees):
    the kernels rely only on the CSR plan, not on the fully-connected triangular layout."""
    r = U.rng(8)
    N = 37
    src = np.concatenate([np.arange(N), np.arange(N), r.integers(0, N, 60)])
    dst = np.concatenate([(np.arange(N) + 1) % N, (np.arange(N) + 5) % N, r.integers(0, N, 60)])
    keep = src != dst
    ei = np.stack([src[keep], dst[keep]])
    ei = ei[:, r.permutation(ei.shape[1])]
    ei = torch.from_numpy(ei)
    E = ei.shape[1]
    hn = U.t32(r.standard_normal((N, 256), dtype=np.float32))
    he = U.t32(r.standard_normal((E, 64), dtype=np.float32))
    pos = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 3)
    nt, et = U.t32(r.random((N, 1), dtype=np.float32)), U.t32(r.random((E, 1), dtype=np.float32))
    m = U.moldiff('MolDiff', DEV)
    o = m.denoiser(hn.to(DEV), pos.to(DEV), he.to(DEV), ei.to(DEV), nt.to(DEV), et.to(DEV))
    with torch.no_grad():
        ref = O.node_edge_net(U.params(m), 'denoiser', hn, pos, he, ei, nt, et, num_blocks=6, cutoff=15)
    assert U.maxdiff(o[0], ref[0]) < 1e-4 and U.maxdiff(o[2], ref[2]) < 1e-4 and U.maxdiff(o[1], ref[1]) < 2e-4


@U.both_paths
def test_coincident_atoms_propagate_nan_like_the_reference():
    """G10 of SURVEY.md: w*rel/d/(d+1) with d == 0 gives NaN in the reference; it must not be 'fixed'."""
    bn, hei, bh, ei, be, xn, xh, pos, t = _fwd_inputs([4], 5)
    pos[1] = pos[0]
    m = U.moldiff('MolDiff', DEV)
    out = m(xn.to(DEV), pos.to(DEV), bn.to(DEV), torch.cat([xh, xh]).to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    with torch.no_grad():
        ref = O.moldiff_forward(U.params(m), U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
    assert torch.isnan(ref['pred_pos']).any() and torch.isnan(out['pred_pos']).any()


def test_workspace_and_tape_sizes_scale_linearly():
    L = _lib.lib()
    a = L.mdx_workspace_bytes(6279, 154666)
    b = L.mdx_workspace_bytes(50357, 1250914)   # B = 2048
    assert 0.3e9 < a < 0.6e9 and 7.5 < b / a < 8.5
    tb = L.mdx_bondpred_tape_bytes(50357, 1250914, 8)
    ta = L.mdx_bondpred_tape_bytes(6279, 154666, 8)
    # (E,256) x 3 + the BondFFN intermediates (E,640) per block, 8 blocks (round 3: 5.9 KB per edge and block); 288 GB holds B = 2048 four times
    assert 6e9 < ta < 9e9 and 7.5 < tb / ta < 8.5 and tb < 72e9


def test_wrong_handle_kind_and_small_workspace_are_reported():
    import ctypes
    m = U.moldiff('MolDiff', DEV)
    eng = m._engine()
    bn, hei, bh, ei, be = U.graph_from_sizes([4, 5], DEV)
    g = _lib.Graph(ei, bn, 2)
    x = torch.zeros(9, 8, device=DEV)
    pos = torch.zeros(9, 3, device=DEV)
    t = torch.zeros(2, dtype=torch.int64, device=DEV)
    out = torch.zeros(g.Eh, 5, device=DEV)
    ws, nb = g.workspace(torch.device(DEV))
    rc = _lib.lib().mdx_bondpred_forward(eng.h, g.h, _lib.ptr(x), _lib.ptr(pos), _lib.ptr(t), _lib.ptr(out), ws, nb, None, 0,
                                         _lib.stream())
    assert rc == 3 and b'BondPredictor' in _lib.lib().mdx_last_error()
    rc = _lib.lib().mdx_moldiff_forward(eng.h, g.h, _lib.ptr(x), _lib.ptr(pos), None, _lib.ptr(torch.zeros(g.Eh, 6, device=DEV)),
                                        _lib.ptr(t), None, None, None, ws, ctypes.c_size_t(1024), _lib.stream())
    assert rc == 3 and b'workspace too small' in _lib.lib().mdx_last_error()


@U.both_paths
def test_distance_smearing_clamps_at_the_cutoff_like_the_reference():
    """GaussianSmearing (models/common.py:233-237) clamps d to [0, cutoff] before the Gaussians: pairs beyond the denoiser's
    15 A cutoff (and well inside the first Gaussian, d ~ 1e-3) must give what the reference gives, through the FUSED path
    (edge kernel A's in-register smearing), not only through the host tables."""
    bn, hei, bh, ei, be, xn, xh, pos, t = _fwd_inputs([9, 6], 11)
    pos = pos * 0.6
    pos[0] += torch.tensor([40.0, 0.0, 0.0])            # 40 A from its molecule: every pair with atom 0 is clamped
    pos[3] = pos[2] + torch.tensor([1e-3, 0.0, 0.0])    # nearly coincident pair
    pos[10] += torch.tensor([0.0, 15.0, 0.0])           # right around the cutoff from several partners
    d = (pos[ei[0]] - pos[ei[1]]).norm(dim=-1)
    assert (d > 15).sum() >= 16 and (d < 2e-3).sum() >= 2
    m = U.moldiff('MolDiff', DEV)
    out = m(xn.to(DEV), pos.to(DEV), bn.to(DEV), torch.cat([xh, xh]).to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    with torch.no_grad():
        ref = O.moldiff_forward(U.params(m), U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
    assert U.maxdiff(out['pred_node'], ref['pred_node']) < 5e-5
    assert U.maxdiff(out['pred_halfedge'], ref['pred_halfedge']) < 5e-5
    # positions: the 1/d/(d+1) force of the nearly coincident pair is ~1e3 times the usual scale; compare relative to it
    scale = max(1.0, float(ref['pred_pos'].abs().max()))
    assert U.maxdiff(out['pred_pos'], ref['pred_pos']) < 2e-4 * scale
