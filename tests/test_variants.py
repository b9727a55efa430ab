"""Constructor variants the reference accepts beyond its shipped configs (VERDICT r3 "missing" 5), against goldens written by the real
reference (oracle/make_goldens_variants.py -> tests/golden/variants.npz).

  * time-free bond predictor: diff.num_timesteps = 0 (models/bond_predictor.py:27-31, :97-102, :141-144)
"""
import copy

import numpy as np
import pytest
import torch

import moldiff_amd as M
from moldiff_amd.harness import default_config
from oracle import moldiff_oracle as O
from tests import util as U

SEED_TIMEFREE = 20230811
RTOL = 2e-5   # loss, as in tests/test_loss.py
GTOL = 1e-4   # parameter gradients, same rule as tests/test_loss.py::_check_param_grads
LTOL = 2e-5   # logits: the golden-forward tolerance of the bond predictor (tests/test_gpu_sampling.py)

_models = {}


def timefree(device='cpu'):
    key = str(device)
    if key not in _models:
        cfg = copy.deepcopy(default_config('bondpred'))
        cfg.diff.num_timesteps = 0
        m = M.BondPredictor(cfg, 8, 5).eval()
        m.load_state_dict(M.recipe_state_dict(m, SEED_TIMEFREE), strict=True)
        _models[key] = m.to(device)
    return _models[key]


def _tf_case(device='cpu'):
    z = U.gold('variants.npz')
    sizes = [int(s) for s in z['tf_sizes']]
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes, device)
    node_type = torch.from_numpy(z['tf_node_type']).to(device)
    node_pos = torch.from_numpy(z['tf_node_pos']).to(device)
    half_type = torch.from_numpy(z['tf_halfedge_type']).to(device)
    return z, sizes, (node_type, node_pos, bn, half_type, hei, bh, len(sizes)), (ei, be)


def test_timefree_predictor_has_the_reference_state_dict():
    """No time embedding, full-width embedders; the key set is the reference's."""
    z = U.gold('variants.npz')
    m = timefree()
    assert sorted(m.state_dict()) == [str(k) for k in z['tf_keys']]
    assert m.node_embedder.weight.shape == (256, 8) and m.edge_embedder.weight.shape == (64, 16)
    assert not hasattr(m, 'time_emb')


def test_oracle_timefree_predictor_matches_reference_golden():
    z, sizes, args, (ei, be) = _tf_case()
    Pb = U.params(timefree())
    cfgb = dict(num_timesteps=0, num_blocks=int(z['tf_num_blocks']), cutoff=float(z['tf_cutoff']))
    with torch.no_grad():
        h = torch.nn.functional.one_hot(args[0], 8).float()
        logits = O.bondpred_forward(Pb, cfgb, h, args[1], args[2], ei, be, None)
        loss = O.bondpred_loss(Pb, cfgb, None, *args, None, {})['loss']
    assert U.maxdiff(logits, z['tf_logits']) <= 1e-6
    assert abs(float(loss) - float(z['tf_loss'])) <= 1e-6


@pytest.mark.gpu
def test_gpu_timefree_predictor_forward_matches_reference_golden():
    z, sizes, args, (ei, be) = _tf_case('cuda')
    m = timefree('cuda')
    with torch.no_grad():
        h = torch.nn.functional.one_hot(args[0], 8).float()
        got = m(h, args[1], args[2], ei, be, None)
        loss = m.get_loss(*args)['loss']
    assert got.shape == z['tf_logits'].shape
    assert U.maxdiff(got, z['tf_logits']) <= LTOL * max(1.0, float(np.abs(z['tf_logits']).max()))
    assert abs(float(loss) - float(z['tf_loss'])) <= RTOL * max(1.0, float(z['tf_loss']))


@pytest.mark.gpu
def test_gpu_timefree_predictor_training_gradients_match_reference():
    z, sizes, args, _ = _tf_case('cuda')
    m = timefree('cuda')
    m.zero_grad(set_to_none=True)
    got = m.get_loss(*args)
    assert abs(float(got['loss'].detach()) - float(z['tf_loss'])) <= RTOL * max(1.0, float(z['tf_loss']))
    got['loss'].backward()
    names = [k[len('tf_grad_norm/'):] for k in z.files if k.startswith('tf_grad_norm/')]
    P = dict(m.named_parameters())
    assert set(names) == {k for k, v in P.items() if v.requires_grad}
    gmax = max(float(z[f'tf_grad_norm/{k}']) for k in names)
    for k in names:
        g = P[k].grad
        assert g is not None, f'no gradient reached {k}'
        want = float(z[f'tf_grad_norm/{k}'])
        scale = max(want, 1e-3 * gmax)
        err = abs(float(g.double().norm()) - want) / scale
        fk = f'tf_grad_full/{k}'
        if fk in z.files:
            err = max(err, float((g - torch.from_numpy(z[fk]).to(g.device)).double().norm()) / scale)
        assert err <= GTOL, (k, err)
    m.zero_grad(set_to_none=True)


@pytest.mark.gpu
def test_gpu_timefree_predictor_guidance_gradient_matches_oracle_autograd():
    """The hand-written d/d pos backward works for the variant too: d sum(log sigmoid(-logsumexp)) / d pos vs autograd through the oracle."""
    z, sizes, args, (ei, be) = _tf_case('cuda')
    m = timefree('cuda')
    h = torch.nn.functional.one_hot(args[0], 8).float()
    pos = args[1].clone().requires_grad_(True)
    logits = m(h, pos, args[2], ei, be, None)
    obj = torch.sigmoid(-torch.logsumexp(logits, -1)).log().sum()
    (g,) = torch.autograd.grad(obj, pos)
    Pb = U.params(timefree())
    cfgb = dict(num_timesteps=0, num_blocks=int(z['tf_num_blocks']), cutoff=float(z['tf_cutoff']))
    pc = args[1].cpu().clone().requires_grad_(True)
    lo = O.bondpred_forward(Pb, cfgb, h.cpu(), pc, args[2].cpu(), ei.cpu(), be.cpu(), None)
    (go,) = torch.autograd.grad(torch.sigmoid(-torch.logsumexp(lo, -1)).log().sum(), pc)
    assert U.maxdiff(g, go) <= 1e-4 * max(1.0, float(go.abs().max()))
