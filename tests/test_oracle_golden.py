"""CPU: the oracle against the golden vectors written by the REAL reference (oracle/make_goldens.py).

This is what pins the oracle on machines where /root/reference does not exist (e.g. the GPU box): every
restated function is re-run here on the committed inputs and must reproduce the reference's outputs
(bit-exact at the generating thread count; <= 2e-5 otherwise -- CPU BLAS blocking depends on thread count,
SURVEY.md section 4).  Weights are regenerated from the recipe seed + the committed key/shape list.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import moldiff_oracle as O
from tests import util as U

TOL = 2e-5
FULL = dict(pos=dict(beta_schedule='advance', scale_start=0.9999, scale_end=0.0001, width=3),
            node=dict(beta_schedule='advance', scale_start=0.9999, scale_end=0.0001, width=3),
            edge=dict(beta_schedule='segment', time_segment=[600, 400],
                      segment_diff=[dict(scale_start=0.9999, scale_end=0.001, width=3),
                                    dict(scale_start=0.001, scale_end=0.0001, width=2)]))
SIMPLE = dict(FULL, edge=FULL['pos'])


def oracle_params(kind):
    """Parameter dict built WITHOUT the product package: recipe weights + the oracle's own tables."""
    which = 'BondPredictor' if kind == 'bondpred' else 'MolDiff'
    shapes = {k: tuple(v) for k, v in U.KEYS[which].items() if not O.is_frozen_key(k)}
    P = O.recipe_state_dict(shapes, U.KEYS['seeds'][which])
    sched = SIMPLE if kind == 'simple' else FULL
    P.update({'pos_transition.' + k: v for k, v in O.pos_tables(O.beta_schedule(sched['pos'], 1000)).items()})
    nt = O.cat_tables(O.beta_schedule(sched['node'], 1000), 8, 'tomask')
    P.update({'node_transition.' + k: nt[k] for k in ('q_mats', 'transpopse_q_onestep_mats')})
    if kind != 'bondpred':
        et = O.cat_tables(O.beta_schedule(sched['edge'], 1000), 6, 'absorb')
        P.update({'edge_transition.' + k: et[k] for k in ('q_mats', 'transpopse_q_onestep_mats')})
        off, co = O.smearing_table(0.0, 15, 16, 'exp')
        P['denoiser.distance_expansion.offset'], P['denoiser.distance_expansion.coeff'] = off, co
        off, co = O.smearing_table(0.0, 1000, 10, 'linear')
        P['time_emb.0.offset'], P['time_emb.0.coeff'] = off, co
    else:
        off, co = O.smearing_table(0.0, 20, 16, 'exp')
        P['encoder.distance_expansion.offset'], P['encoder.distance_expansion.coeff'] = off, co
        off, co = O.smearing_table(0.0, 1000, 20, 'linear')
        P['time_emb.offset'], P['time_emb.coeff'] = off, co
    return P


@pytest.fixture(scope='module')
def P_full():
    return oracle_params('full')


@pytest.fixture(scope='module')
def P_bond():
    return oracle_params('bondpred')


def test_recipe_weights_match_committed_hash(P_full, P_bond):
    for which, P in (('MolDiff', P_full), ('BondPredictor', P_bond)):
        h = hashlib.sha256()
        for k in sorted(P):
            if not O.is_frozen_key(k):
                h.update(P[k].numpy().tobytes())
        assert h.hexdigest() == U.KEYS[f'recipe_sha256_{which}']


def test_schedules_and_tables():
    g = U.gold('schedules.npz')
    probe = g['probe_t']
    for nm, sched in (('full', FULL), ('simple', SIMPLE)):
        for part in ('pos', 'node', 'edge'):
            b = O.beta_schedule(sched[part], 1000)
            assert np.abs(b - g[f'{nm}_{part}_betas']).max() == 0.0
        pt = O.pos_tables(g[f'{nm}_pos_betas'])
        for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar'):
            assert np.array_equal(pt[k].numpy(), g[f'{nm}_pos_{k}'])
        for part, K, kind in (('node', 8, 'tomask'), ('edge', 6, 'absorb')):
            ct = O.cat_tables(g[f'{nm}_{part}_betas'], K, kind)
            for k in ('q_mats', 'transpopse_q_onestep_mats'):
                assert np.array_equal(ct[k][probe].numpy(), g[f'{nm}_{part}_{k}_probe'])
            assert np.array_equal(ct['init_prob'], g[f'{nm}_{part}_init_prob'])


def test_known_answer_constants():
    """SURVEY.md Appendix C (values printed by the reference)."""
    b = O.beta_schedule(FULL['pos'], 1000)
    assert np.allclose(b[[0, 1, 499, 999]], [1.0e-4, 3.005505359e-4, 3.300573892e-3, 0.7503248778], rtol=1e-8)
    bs = O.beta_schedule(FULL['edge'], 1000)
    assert np.allclose(bs[[0, 599, 600, 999]], [6.00821e-4, 0.333698131, 1.245475e-3, 0.012301536], rtol=1e-5)
    pt = O.pos_tables(b)
    assert np.allclose([pt['coef_x0'][1], pt['coef_xt'][1], pt['std'][1]], [0.75036240, 0.24963760, 0.0086625628], rtol=1e-6)
    assert float(pt['coef_x0'][0]) == 1.0 and float(pt['coef_xt'][0]) == 0.0 and float(pt['std'][0]) == 0.0
    off, co = O.smearing_table(0.0, 15, 16, 'exp')
    assert np.allclose(off[[1, 15]].numpy(), [0.20302498, 15.0], rtol=1e-6)
    assert np.allclose(co[[0, 1, 15]].numpy(), [-12.130285, -12.130285, -0.06857728], rtol=1e-5)


def test_smearing():
    g = U.gold('smearing.npz')
    for nm, (start, stop, n, kind) in {'d15': (0.0, 15, 16, 'exp'), 'd20': (0.0, 20, 16, 'exp'),
                                        't10': (0.0, 1000, 10, 'linear'), 't20': (0.0, 1000, 20, 'linear')}.items():
        off, co = O.smearing_table(start, stop, n, kind)
        assert np.array_equal(off.numpy(), g[nm + '_offset']) and np.array_equal(co.numpy(), g[nm + '_coeff'])
        x = torch.from_numpy(g['dist'] if nm[0] == 'd' else g['tid'])
        assert U.maxdiff(O.smear(x, off, co, start, stop), g[nm + '_out']) <= 1e-7


def test_blocks(P_full):
    g = U.gold('blocks_full.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes([5, 7])
    assert np.array_equal(hei.numpy(), g['halfedge_index'])
    x, ea, pos, tg = U.t32(g['x']), U.t32(g['edge_attr']), U.t32(g['pos']), torch.from_numpy(g['t'])
    nt, et = tg[bn].unsqueeze(-1) / 1000, tg[be].unsqueeze(-1) / 1000
    rel = pos[ei[0]] - pos[ei[1]]
    dist = torch.norm(rel, dim=-1)
    with torch.no_grad():
        for i in (0, 3):
            assert U.maxdiff(O.node_block(P_full, f'denoiser.node_blocks_with_edge.{i}', x, ei, ea, nt), g[f'nodeblock{i}_out']) < TOL
            assert U.maxdiff(O.edge_block(P_full, f'denoiser.edge_blocks.{i}', ea, ei, x, et), g[f'edgeblock{i}_out']) < TOL
            assert U.maxdiff(O.pos_update(P_full, f'denoiser.pos_blocks.{i}', x, ea, ei, rel, dist, et), g[f'posupdate{i}_out']) < TOL
            assert U.maxdiff(O.bond_ffn(P_full, f'denoiser.edge_blocks.{i}.bond_ffn_left', ea, x[ei[0]], et),
                             g[f'bondffn_left{i}_out']) < TOL


def test_node_edge_net(P_full, P_bond):
    gd = U.gold('nodeedgenet.npz')
    for tag in ('n12', 'n204'):
        bn, hei, bh, ei, be = U.graph_from_sizes(gd[f'{tag}_sizes'])
        N, E = len(bn), ei.shape[1]
        r = U.rng(int(gd['input_seed']))
        hn, he = U.t32(r.standard_normal((N, 256), dtype=np.float32)), U.t32(r.standard_normal((E, 64), dtype=np.float32))
        pos = U.t32(r.standard_normal((N, 3), dtype=np.float32) * 2)
        tg = torch.from_numpy(r.integers(0, 1000, int(bn.max()) + 1))
        nt, et = tg[bn].unsqueeze(-1) / 1000, tg[be].unsqueeze(-1) / 1000
        stride = 1 if tag == 'n12' else 16
        with torch.no_grad():
            o = O.node_edge_net(P_full, 'denoiser', hn, pos, he, ei, nt, et, num_blocks=6, cutoff=15)
            ob = O.node_edge_net(P_bond, 'encoder', hn, pos, he, ei, nt, et, num_blocks=8, cutoff=20, update_pos=False)
        assert U.maxdiff(o[0], gd[f'{tag}_6_h_node']) < 1e-4 and U.maxdiff(o[1], gd[f'{tag}_6_pos']) < 1e-4
        assert U.maxdiff(o[2][::stride], gd[f'{tag}_6_h_edge_s{stride}']) < 1e-4
        assert U.maxdiff(ob[0], gd[f'{tag}_8_h_node']) < 1e-4
        assert U.maxdiff(ob[2][::stride], gd[f'{tag}_8_h_edge_s{stride}']) < 1e-4


def test_forward_and_bondpred(P_full, P_bond):
    g = U.gold('forward.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes(g['sizes'])
    xn = F.one_hot(torch.from_numpy(g['node_type']), 8).float()
    xh = F.one_hot(torch.from_numpy(g['halfedge_type']), 6).float()
    pos = U.t32(g['pos'])
    with torch.no_grad():
        for tv in ('t999', 't0', 'tmix'):
            t = torch.from_numpy(g['tmix']) if tv == 'tmix' else torch.full((8,), int(tv[1:]), dtype=torch.long)
            o = O.moldiff_forward(P_full, U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
            for k in o:
                assert U.maxdiff(o[k], g[f'{tv}_{k}']) < (1e-4 if k == 'pred_pos' else TOL)
        lb = O.bondpred_forward(P_bond, U.CFGB, xn, pos, bn, ei, be, torch.from_numpy(g['tmix']))
        assert U.maxdiff(lb, g['tmix_bond_logits']) < TOL


def test_guidance_delta(P_bond):
    g = U.gold('guidance.npz')
    for tag in ('n12', 'n101'):
        bn, hei, bh, ei, be = U.graph_from_sizes(g[f'{tag}_sizes'])
        xn = F.one_hot(torch.from_numpy(g[f'{tag}_node_type']), 8).float()
        t = torch.full((int(bn.max()) + 1,), int(g['t']), dtype=torch.long)
        d, lg = O.guidance_delta(P_bond, U.CFGB, xn, U.t32(g[f'{tag}_pos']), bn, ei, be, t, 1e-4)
        assert U.maxdiff(lg, g[f'{tag}_logits']) < TOL
        assert U.maxdiff(d, g[f'{tag}_delta']) < 1e-8 + 1e-3 * float(np.abs(g[f'{tag}_delta']).max())


def _gt_state(g, tag, gt, j):
    """State iteration j of the guidance_types.npz run of objective `gt` starts from (frame j; frame 0 = the prior)."""
    if j == 0:
        nt, ht = g[f'{tag}_init_node_type'], g[f'{tag}_init_halfedge_type']
        oh_n, oh_h = F.one_hot(torch.from_numpy(nt.astype(np.int64)), 8).float(), F.one_hot(torch.from_numpy(ht.astype(np.int64)), 6).float()
        return {'h_node': oh_n, 'pos': U.t32(g[f'{tag}_init_pos']), 'h_halfedge': oh_h,
                'log_node': torch.log(oh_n.clamp(min=1e-30)), 'log_halfedge': torch.log(oh_h.clamp(min=1e-30))}
    nt, ht = g[f'{tag}_0_none_node_type'], g[f'{tag}_0_none_halfedge_type']
    return {'h_node': F.one_hot(torch.from_numpy(nt.astype(np.int64)), 8).float(), 'pos': U.t32(g[f'{tag}_0_{gt}_pos']),
            'h_halfedge': F.one_hot(torch.from_numpy(ht.astype(np.int64)), 6).float(),
            'log_node': U.t32(g[f'{tag}_0_log_node']), 'log_halfedge': U.t32(g[f'{tag}_0_log_halfedge'])}


@pytest.mark.parametrize('gt', O.GUIDANCE_TYPES)
def test_all_eight_guidance_objectives_vs_reference_sample(gt, P_full, P_bond):
    """models/model.py:317-359: guidance_types.npz was written by the reference's own `sample()` (oracle/make_goldens_guidance.py);
    the oracle's step, teacher-forced from the reference's frames, must land on the reference's guided positions -- which
    `halfedge_type_prev` / `log_halfedge_type` feed the logit* / crossent* objectives is exactly what this catches."""
    g = U.gold('guidance_types.npz')
    first, scale = int(g['first']), float(g['scale'])
    tabs = U.tables(P_full)
    for tag in ('n12', 'n101'):
        bn, hei, bh, ei, be = U.graph_from_sizes(g[f'{tag}_sizes'])
        graph = {'batch_node': bn, 'halfedge_index': hei, 'batch_halfedge': bh, 'n_graphs': int(bn.max()) + 1}
        for j in range(int(g['nsteps'])):
            st = _gt_state(g, tag, gt, j)
            noise = {k: U.t32(g[f'{tag}_{j}_{k}']) for k in ('eps_pos', 'u_node', 'u_halfedge')}
            with torch.no_grad():
                new, _ = O.sample_step(P_full, U.CFG, tabs, st, graph, 999 - (first + j), noise, Pb=P_bond, cfgb=U.CFGB,
                                       guidance=[gt, scale])
            ref = g[f'{tag}_{j}_{gt}_pos']
            shift = float(np.abs(ref - g[f'{tag}_{j}_none_pos']).max()) if j == 0 else 0.0
            assert U.maxdiff(new['pos'], ref) <= 2.0 ** -21, (tag, j)   # a few fp32 ulps of an O(1) position (CPU autograd order)
            if j == 0:
                assert np.array_equal(new['halfedge_type'].numpy(), g[f'{tag}_0_none_halfedge_type'])
                if not (tag == 'n12' and gt == 'logit_bond'):   # no real bond among 31 half-edges: that shift is exactly zero
                    assert shift > 1e-3
    with pytest.raises(NotImplementedError):
        O.guidance_objective('nope', torch.zeros(1, 5))


@pytest.mark.parametrize('tag', ['simple', 'guided'])
def test_step_replay(tag, P_full, P_bond):
    g = U.gold('step_replay.npz')
    P = oracle_params('simple') if tag == 'simple' else P_full
    tabs = U.tables(P)
    bn, hei, bh, ei, be = U.graph_from_sizes(g['sizes'])
    graph = {'batch_node': bn, 'halfedge_index': hei, 'batch_halfedge': bh, 'n_graphs': 4}
    for window in ('hi', 'lo'):
        pre = f'{tag}_{window}'
        st = {'h_node': F.one_hot(torch.from_numpy(g[pre + '_init_node_type']), 8).float(), 'pos': U.t32(g[pre + '_init_pos']),
              'h_halfedge': F.one_hot(torch.from_numpy(g[pre + '_init_halfedge_type']), 6).float(),
              'log_node': U.t32(g[pre + '_init_log_node']), 'log_halfedge': U.t32(g[pre + '_init_log_halfedge'])}
        for j, s in enumerate(g[pre + '_steps'][:2]):
            noise = {k: U.t32(g[f'{pre}_{j}_{k}']) for k in ('eps_pos', 'u_node', 'u_halfedge')}
            with torch.no_grad():
                new, preds = O.sample_step(P, U.CFG, tabs, st, graph, int(s), noise, Pb=P_bond, cfgb=U.CFGB,
                                           guidance=['uncertainty', 1e-4] if tag == 'guided' else None)
            assert U.maxdiff(new['pos'], g[f'{pre}_{j}_pos']) < 1e-4
            assert U.maxdiff(new['log_node'], g[f'{pre}_{j}_log_node']) < 1e-4
            assert U.maxdiff(new['log_halfedge'], g[f'{pre}_{j}_log_halfedge']) < 1e-4
            assert np.array_equal(new['node_type'].numpy(), g[f'{pre}_{j}_node_type'])
            assert np.array_equal(new['halfedge_type'].numpy(), g[f'{pre}_{j}_halfedge_type'])
            st = {'h_node': F.one_hot(torch.from_numpy(g[f'{pre}_{j}_node_type']), 8).float(), 'pos': U.t32(g[f'{pre}_{j}_pos']),
                  'h_halfedge': F.one_hot(torch.from_numpy(g[f'{pre}_{j}_halfedge_type']), 6).float(),
                  'log_node': U.t32(g[f'{pre}_{j}_log_node']), 'log_halfedge': U.t32(g[f'{pre}_{j}_log_halfedge'])}


def test_init_and_placeholder():
    g = U.gold('init.npz')
    for part, K, kind in (('node', 8, 'tomask'), ('edge', 6, 'absorb')):
        tab = O.cat_tables(O.beta_schedule(FULL[part], 1000), K, kind)
        c, oh, lv = O.cat_init(tab, 64, torch.from_numpy(g[f'{part}_u']))
        assert np.array_equal(c.numpy(), g[f'{part}_class']) and U.maxdiff(lv, g[f'{part}_log']) == 0.0
    p = U.gold('placeholder.npz')
    for B in (8, 256, 2048):
        np.random.seed(2920)
        ph = O.placeholder(B)
        assert np.array_equal(ph['n_nodes_list'], p[f'B{B}_sizes'])
        assert len(ph['batch_node']) == int(p[f'B{B}_N']) and len(ph['batch_halfedge']) == int(p[f'B{B}_Eh'])
        assert np.array_equal(ph['halfedge_index'][:, :16].numpy(), p[f'B{B}_he_first16'])
        assert np.array_equal(ph['halfedge_index'][:, -16:].numpy(), p[f'B{B}_he_last16'])
    assert (int(p['B8_N']), int(p['B8_Eh'])) == (204, 2552) and (int(p['B256_N']), int(p['B256_Eh'])) == (6279, 77333)


def test_pinning_record_says_bit_exact():
    """oracle/make_goldens.py's record of max |oracle - reference| per function: EXACTLY 0.0 on every forward quantity; the two
    quantities that pass through autograd (the guidance increment and the guided position it is added to) are pinned to one fp32
    ulp of a unit-scale position (1.2e-7), not to 0 -- CPU autograd accumulates in thread-dependent order, so regenerating the
    record gives between 9e-10 and 6e-8 there."""
    full = json.load(open(os.path.join(U.GOLD, 'PINNING.json')))
    # all eight guidance objectives against the reference's own sample() (oracle/make_goldens_guidance.py): positions after a
    # guided step, i.e. through CPU autograd -> a few fp32 ulps of an O(1) position
    assert full['guidance_types_pos_max'] <= 2.0 ** -21 and len(full['guidance_types_detail']) == 32
    rec = full['max_abs_diff_oracle_vs_reference']
    autograd = {'guidance_delta', 'sample_step_pos'}
    assert autograd <= set(rec)
    for name, d in rec.items():
        if name in autograd:
            assert d <= 2.0 ** -23, (name, d)
        else:
            assert d == 0.0, (name, d)


def test_oracle_matches_reference_on_stress_weights():
    """tests/golden/stress.npz (oracle/make_goldens_stress.py: the REAL reference with heavy-tailed weights -- LayerNorm gains up to
    30, biases x 8, residual streams 10^2 - 10^3): the oracle's forward, bond logits, guidance increment and one guided loop
    iteration, N = 12 and N = 101.  Outputs are O(10): tolerances relative to each quantity's scale."""
    g = U.gold('stress.npz')
    P, Pb = U.params(U.moldiff_stress()), U.params(U.bondpred_stress())
    for tag in ('n12', 'n101'):
        bn, hei, bh, ei, be = U.graph_from_sizes(g[f'{tag}_sizes'])
        xn = F.one_hot(torch.from_numpy(g[f'{tag}_node_type']), 8).float()
        xh = F.one_hot(torch.from_numpy(g[f'{tag}_halfedge_type']), 6).float()
        pos, t = U.t32(g[f'{tag}_pos']), torch.from_numpy(g[f'{tag}_t'])
        with torch.no_grad():
            o = O.moldiff_forward(P, U.CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
        for k in o:
            assert U.maxdiff(o[k], g[f'{tag}_{k}']) <= 1e-5 * max(1.0, float(np.abs(g[f'{tag}_{k}']).max())), (tag, k)
        d, lg = O.guidance_delta(Pb, U.CFGB, xn, pos, bn, ei, be, t, 1e-4)
        assert U.maxdiff(lg, g[f'{tag}_bond_logits']) <= 1e-5 * float(np.abs(g[f'{tag}_bond_logits']).max())
        assert U.maxdiff(d, g[f'{tag}_delta']) <= 1e-8 + 1e-3 * float(np.abs(g[f'{tag}_delta']).max())
        st = {'h_node': xn, 'pos': pos, 'h_halfedge': xh, 'log_node': U.t32(g[f'{tag}_step_log_node_in']),
              'log_halfedge': U.t32(g[f'{tag}_step_log_halfedge_in'])}
        noise = {k: U.t32(g[f'{tag}_step_{k}']) for k in ('eps_pos', 'u_node', 'u_halfedge')}
        with torch.no_grad():
            new, preds = O.sample_step(P, U.CFG, U.tables(P), st, {'batch_node': bn, 'halfedge_index': hei, 'batch_halfedge': bh,
                                                                    'n_graphs': len(g[f'{tag}_sizes'])}, int(g[f'{tag}_step']), noise,
                                       Pb=Pb, cfgb=U.CFGB, guidance=['uncertainty', 1e-4])
        assert U.maxdiff(new['pos'], g[f'{tag}_step_pos']) <= 1e-5 * max(1.0, float(np.abs(g[f'{tag}_step_pos']).max()))
        assert U.maxdiff(new['log_node'], g[f'{tag}_step_log_node']) < 1e-4 and U.maxdiff(new['log_halfedge'], g[f'{tag}_step_log_halfedge']) < 1e-4
        assert np.array_equal(new['node_type'].numpy(), g[f'{tag}_step_node_type'])
        assert np.array_equal(new['halfedge_type'].numpy(), g[f'{tag}_step_halfedge_type'])
    full = json.load(open(os.path.join(U.GOLD, 'PINNING.json')))
    assert all(full[k] == 0.0 for k in ('stress_moldiff_forward', 'stress_guidance_logits', 'stress_sample_step_preds', 'stress_sample_step_pos'))
    assert full['stress_guidance_delta'] <= 1e-8
