"""GPU parity of the bond predictor (models/bond_predictor.py:128-162) and of the guidance gradient
(models/model.py:312-325) against goldens written by the real reference.

Tolerances: logits <= 2e-5 abs; delta = -1e-4 * dU/dpos compared relative to its own scale
(|delta| ~ 1e-4..1e-3): max abs error <= 1e-3 * max|delta| + 1e-9 (fp32 backward through 8 blocks).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util as U
from oracle import moldiff_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@U.both_paths
def test_bondpred_forward_vs_golden():
    g = U.gold('forward.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes(g['sizes'])
    xn = F.one_hot(torch.from_numpy(g['node_type']), 8).float()
    m = U.bondpred(DEV)
    with torch.no_grad():
        out = m(xn.to(DEV), U.t32(g['pos']).to(DEV), bn.to(DEV), ei.to(DEV), be.to(DEV), torch.from_numpy(g['tmix']).to(DEV))
    assert U.maxdiff(out, g['tmix_bond_logits']) < 2e-5


@U.both_paths
@pytest.mark.parametrize('tag', ['n12', 'n101'])
def test_guidance_delta_vs_golden(tag):
    g = U.gold('guidance.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes(g[f'{tag}_sizes'])
    xn = F.one_hot(torch.from_numpy(g[f'{tag}_node_type']), 8).float().to(DEV)
    pos = U.t32(g[f'{tag}_pos']).to(DEV).requires_grad_(True)
    t = torch.full((int(bn.max()) + 1,), int(g['t']), dtype=torch.long, device=DEV)
    m = U.bondpred(DEV)
    # exactly the reference's guidance code (models/model.py:315,322-325), with the drop-in module
    logits = m(xn, pos, bn.to(DEV), ei.to(DEV), be.to(DEV), t)
    uncertainty = torch.sigmoid(-torch.logsumexp(logits, dim=-1)).log().sum()
    delta = -torch.autograd.grad(uncertainty, pos)[0] * 1e-4
    assert U.maxdiff(logits, g[f'{tag}_logits']) < 2e-5
    ref = g[f'{tag}_delta']
    assert U.maxdiff(delta, ref) <= 1e-3 * float(np.abs(ref).max()) + 1e-9


@U.both_paths
def test_backward_of_arbitrary_logit_functional_vs_oracle_autograd():
    """Any scalar of the logits works (the other guidance types of model.py:317-359): random cotangent."""
    bn, hei, bh, ei, be = U.graph_from_sizes([6, 9, 4])
    N, Eh = len(bn), len(bh)
    r = U.rng(31)
    xn = F.one_hot(torch.from_numpy(r.integers(0, 8, N)), 8).float()
    pos0 = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 1.5)
    t = torch.tensor([10, 500, 990])
    cot = U.t32(r.standard_normal((Eh, 5), dtype=np.float32))
    m = U.bondpred(DEV)
    pos = pos0.to(DEV).requires_grad_(True)
    logits = m(xn.to(DEV), pos, bn.to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    got, = torch.autograd.grad((logits * cot.to(DEV)).sum(), pos)
    P = U.params(m)
    p = pos0.clone().requires_grad_(True)
    ref_logits = O.bondpred_forward(P, U.CFGB, xn, p, bn, ei, be, t)
    ref, = torch.autograd.grad((ref_logits * cot).sum(), p)
    assert U.maxdiff(logits, ref_logits) < 2e-5
    assert U.maxdiff(got, ref) <= 1e-3 * float(ref.abs().max())


@U.both_paths
def test_guidance_gradient_with_distances_straddling_the_cutoff_vs_oracle_autograd():
    """The clamp of GaussianSmearing passes the gradient on the closed interval [0, cutoff] and blocks it beyond
    (torch.clamp's backward): pairs beyond the predictor's 20 A cutoff contribute nothing to dL/dpos through their own
    distance, pairs inside do -- checked through the row-owner backward kernel against the oracle's autograd."""
    bn, hei, bh, ei, be = U.graph_from_sizes([7, 5, 8])
    N, Eh = len(bn), len(bh)
    r = U.rng(47)
    xn = F.one_hot(torch.from_numpy(r.integers(0, 8, N)), 8).float()
    pos0 = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 1.5)
    pos0[0] += torch.tensor([30.0, 0.0, 0.0])   # beyond the cutoff from all of molecule 0
    pos0[8] += torch.tensor([0.0, 19.0, 0.0])   # around the cutoff from molecule 1's atoms
    d = (pos0[ei[0]] - pos0[ei[1]]).norm(dim=-1)
    assert (d > 20).sum() >= 12 and ((d > 15) & (d <= 20)).sum() >= 2
    t = torch.tensor([20, 400, 950])
    cot = U.t32(r.standard_normal((Eh, 5), dtype=np.float32))
    m = U.bondpred(DEV)
    pos = pos0.to(DEV).requires_grad_(True)
    logits = m(xn.to(DEV), pos, bn.to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    got, = torch.autograd.grad((logits * cot.to(DEV)).sum(), pos)
    p = pos0.clone().requires_grad_(True)
    ref_logits = O.bondpred_forward(U.params(m), U.CFGB, xn, p, bn, ei, be, t)
    ref, = torch.autograd.grad((ref_logits * cot).sum(), p)
    assert U.maxdiff(logits, ref_logits) < 2e-5
    assert U.maxdiff(got, ref) <= 1e-3 * float(ref.abs().max())


@U.both_paths
def test_guidance_gradient_on_a_sparse_shuffled_graph_vs_oracle_autograd():
    """The backward walks the edges in BY-RIGHT order and sums its by-right payloads over each right node's run inside the kernel
    (round 5): nothing in that may rely on the fully-connected molecule layout.  Two 'molecules' (13 and 20 atoms) whose half-edges
    are a random 35 % subset of all pairs plus a ring, given in shuffled order -- irregular degrees (2..12 edges per atom), runs that
    straddle the 16-row units -- logits and d(sum log sigmoid(-logsumexp))/d pos vs the oracle."""
    r = U.rng(77)
    sizes, off, pairs, bn = [13, 20], 0, [], []
    for b, n in enumerate(sizes):
        iu = np.stack(np.triu_indices(n, 1))
        keep = r.random(iu.shape[1]) < 0.35
        ring = np.stack([np.arange(n - 1), np.arange(1, n)])          # keeps every atom connected
        he = np.unique(np.concatenate([iu[:, keep], ring], axis=1), axis=1)
        he = he[:, r.permutation(he.shape[1])] + off
        pairs.append(he)
        bn += [b] * n
        off += n
    # half-edges grouped by molecule (batch_halfedge is non-decreasing like the reference's), shuffled inside each molecule
    hei = torch.from_numpy(np.concatenate(pairs, axis=1)).long()
    bn = torch.tensor(bn)
    ei = torch.cat([hei, hei.flip(0)], 1)
    be = bn[ei[0]]
    N, Eh = len(bn), hei.shape[1]
    xn = F.one_hot(torch.from_numpy(r.integers(0, 8, N)), 8).float()
    pos0 = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 2.0)
    t = torch.tensor([120, 870])
    m = U.bondpred(DEV)
    pos = pos0.to(DEV).requires_grad_(True)
    logits = m(xn.to(DEV), pos, bn.to(DEV), ei.to(DEV), be.to(DEV), t.to(DEV))
    (got,) = torch.autograd.grad(torch.sigmoid(-torch.logsumexp(logits, -1)).log().sum(), pos)
    P = U.params(m)
    p = pos0.clone().requires_grad_(True)
    ref_logits = O.bondpred_forward(P, U.CFGB, xn, p, bn, ei, be, t)
    (ref,) = torch.autograd.grad(torch.sigmoid(-torch.logsumexp(ref_logits, -1)).log().sum(), p)
    assert logits.shape == (Eh, 5)
    assert U.maxdiff(logits, ref_logits) < 2e-5
    assert U.maxdiff(got, ref) <= 1e-3 * float(ref.abs().max())
