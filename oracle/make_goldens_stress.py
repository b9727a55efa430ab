"""Stress-weight goldens written by the REAL reference on CPU.  BUILD-CONTAINER ONLY (needs /root/reference).

    python oracle/make_goldens_stress.py        ->  tests/golden/stress.npz  (+ `stress_*` keys in tests/golden/PINNING.json)

Every other fixture uses the benign recipe weights (N(0,1)/sqrt(fan_in), LayerNorm gains near 1).  No trained checkpoint exists
offline (ckpt/README.md), so this is the next best probe of what trained weights may hold: `stress_state_dict` (LayerNorm gains
log-uniform in [0.1, 30], biases x 8, block 2's out_transform x 16 -> residual streams and un-normalised pairwise products reach
10^3 - 4 10^4, i.e. the upper half of float16's range that the split matrix path's operands live in).  Stored, for a 4-molecule
graph (N = 101) and the 2-molecule graph (N = 12):
  * `MolDiff.forward` (models/model.py:204-234) at mixed time steps: the three outputs;
  * one iteration of the guided sampling loop body (models/model.py:272-372) run through the reference's own modules with the
    random draws injected (make_goldens.reference_step): new state, predictions, Gumbel margins;
  * `BondPredictor.forward` logits and the 'uncertainty' guidance increment -1e-4 d/dpos (models/model.py:312-325) by the
    reference's autograd.
The oracle is compared with each (recorded in PINNING.json; forward quantities are expected to be exactly 0).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import moldiff_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle.make_goldens import SEED_BONDPRED, SEED_MOLDIFF, graph, reference_step, rng_inputs, two_mol_graph  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
CFG = dict(num_timesteps=1000, num_blocks=6, cutoff=15)
CFGB = dict(num_timesteps=1000, num_blocks=8, cutoff=20)


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    pin = {}

    def rec(name, a, b):
        d = float((a - b).abs().max()) if a.numel() else 0.0
        pin[name] = max(pin.get(name, 0.0), d)

    def build(cls, cfg, kn, ke, seed):
        m = cls(cfg.model, kn, ke).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd2 = dict(sd)
        sd2.update(O.stress_state_dict(shapes, seed))
        m.load_state_dict(sd2, strict=True)
        return m, {k: v.detach().clone() for k, v in m.state_dict().items()}

    m_full, P = build(MolDiff, ref_shim.load_yaml_cfg('configs/train/train_MolDiff.yml'), 8, 6, SEED_MOLDIFF)
    m_bond, Pb = build(BondPredictor, ref_shim.load_yaml_cfg('configs/train/train_bondpred.yml'), 8, 5, SEED_BONDPRED)
    tabs = {'pos': {k: P['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std')},
            'node': {k: P['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
            'edge': {k: P['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}
    out = {}
    for tag, gr in {'n12': two_mol_graph(), 'n101': graph(4)[1:]}.items():
        bn, hei, bh, ei, be = gr
        N, Eh, B = len(bn), len(bh), int(bn.max()) + 1
        g = rng_inputs(41 + (tag == 'n101'))
        nt, ht = torch.from_numpy(g.integers(0, 8, N)), torch.from_numpy(g.integers(0, 6, Eh))
        xn, xh = F.one_hot(nt, 8).float(), F.one_hot(ht, 6).float()
        pos = torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * 2.0)
        t = torch.from_numpy(g.integers(0, 1000, B))
        out[f'{tag}_sizes'] = np.bincount(bn.numpy())
        out[f'{tag}_node_type'], out[f'{tag}_halfedge_type'], out[f'{tag}_pos'], out[f'{tag}_t'] = nt.numpy(), ht.numpy(), pos.numpy(), t.numpy()
        # --- MolDiff.forward
        with torch.no_grad():
            r = m_full(xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
            o = O.moldiff_forward(P, CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
        for k in r:
            rec('stress_moldiff_forward', o[k], r[k])
            out[f'{tag}_{k}'] = r[k].numpy()
        # --- bond predictor logits + guidance increment (reference autograd)
        with torch.enable_grad():
            p = pos.clone().requires_grad_(True)
            logits = m_bond(xn, p, bn, ei, be, t)
            u = torch.sigmoid(-torch.logsumexp(logits, -1)).log().sum()
            delta = -torch.autograd.grad(u, p)[0] * 1e-4
        od, ol = O.guidance_delta(Pb, CFGB, xn, pos, bn, ei, be, t, 1e-4)
        rec('stress_guidance_logits', ol, logits.detach())
        rec('stress_guidance_delta', od, delta)
        out[f'{tag}_bond_logits'], out[f'{tag}_delta'] = logits.detach().numpy(), delta.numpy()
        # --- one guided iteration of the loop body at a common step (the loop has one t per iteration)
        step = 400
        st = {'h_node': xn, 'pos': pos, 'h_halfedge': xh,
              'log_node': F.log_softmax(torch.from_numpy(g.standard_normal((N, 8), dtype=np.float32)), -1),
              'log_halfedge': F.log_softmax(torch.from_numpy(g.standard_normal((Eh, 6), dtype=np.float32)), -1)}
        noise = {'eps_pos': torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32)),
                 'u_node': torch.from_numpy(g.random((N, 8), dtype=np.float32)),
                 'u_halfedge': torch.from_numpy(g.random((Eh, 6), dtype=np.float32))}
        guid = ['uncertainty', 1e-4]
        ref_state, ref_preds = reference_step(m_full, DF, st, bn, hei, bh, step, noise, m_bond, guid, B=B)
        o_state, o_preds = O.sample_step(P, CFG, tabs, st, {'batch_node': bn, 'halfedge_index': hei, 'batch_halfedge': bh, 'n_graphs': B},
                                         step, noise, Pb=Pb, cfgb=CFGB, guidance=guid)
        for k in ('pos', 'log_node', 'log_halfedge'):
            rec(f'stress_sample_step_{k}', o_state[k], ref_state[k])
        assert torch.equal(o_state['node_type'], ref_state['node_type']) and torch.equal(o_state['halfedge_type'], ref_state['halfedge_type'])
        for k in ref_preds:
            rec('stress_sample_step_preds', o_preds[k], ref_preds[k])
        out[f'{tag}_step'] = np.array(step)
        out[f'{tag}_step_log_node_in'], out[f'{tag}_step_log_halfedge_in'] = st['log_node'].numpy(), st['log_halfedge'].numpy()
        for k in noise:
            out[f'{tag}_step_{k}'] = noise[k].numpy()
        for k in ('pos', 'log_node', 'log_halfedge', 'node_type', 'halfedge_type'):
            out[f'{tag}_step_{k}'] = ref_state[k].numpy()
        out[f'{tag}_step_pred_pos'], out[f'{tag}_step_pred_node'] = ref_preds['pred_pos'].numpy(), ref_preds['pred_node'].numpy()
        for nm, lg, uu in (('node', ref_state['log_node'], noise['u_node']), ('halfedge', ref_state['log_halfedge'], noise['u_halfedge'])):
            z = lg - torch.log(-torch.log(uu + 1e-30) + 1e-30)
            top = z.topk(2, -1).values
            out[f'{tag}_step_{nm}_margin_min'] = np.array(float((top[:, 0] - top[:, 1]).min()))
    np.savez_compressed(os.path.join(OUT, 'stress.npz'), **out)
    pj = os.path.join(OUT, 'PINNING.json')
    full = json.load(open(pj))
    full.update(pin)
    with open(pj, 'w') as f:
        json.dump(full, f, indent=0, sort_keys=True)
    print('wrote stress.npz;', {k: f'{v:.3e}' for k, v in pin.items()})
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if v.dtype.kind == 'f' and v.ndim > 0 and k.startswith('n101')})


if __name__ == '__main__':
    main()
