"""Golden vectors for ALL EIGHT guidance objectives (models/model.py:317-359), written by the REAL reference's own
`MolDiff.sample` on CPU.  BUILD-CONTAINER ONLY (needs /root/reference, read-only).

    python oracle/make_goldens_guidance.py        ->  tests/golden/guidance_types.npz

How: the reference's `sample()` (models/model.py:236-378) is called UNMODIFIED with `bond_predictor` and
`guidance=[type, scale]` for each of the eight types.  Two things around it are replaced, neither of them reference arithmetic:
  * `tqdm` in `models.model`'s namespace becomes an `itertools.islice` over the loop's (i, step) pairs, so that only the
    iterations i = FIRST .. FIRST + NSTEPS - 1 (time steps 999 - i) of the 1000 run -- the loop body itself decides nothing
    about which iterations exist; skipped iterations leave the state at the prior;
  * the torch RNG entry points the method draws from (`torch.randn`, `torch.randn_like`, `torch.rand_like`) return arrays of a
    seeded numpy generator, re-seeded per run: all nine runs (no guidance + eight types) see the same prior and the same noise,
    so the runs differ by the guidance shift alone, and the arrays are stored for the teacher-forced GPU test.
What is stored per graph (n = (5,7): N = 12; the 4-molecule batch of step_replay.npz: N = 101): prior state, the noise of each
executed iteration, the unguided state after each iteration (positions, log-posteriors, sampled classes) and, per objective,
the positions after each iteration (iteration 2 starts from that objective's own guided state, so the second frame also pins
the carried `log_halfedge_type`).  The same run pins oracle.guidance_delta for every type (recorded in PINNING.json).
"""
import itertools
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
from oracle.make_goldens import SEED_BONDPRED, SEED_MOLDIFF, graph, two_mol_graph  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'guidance_types.npz')
FIRST, NSTEPS = 497, 2      # iterations 497, 498 = time steps 502, 501 (mid-chain: the posterior mixes prior and prediction)
SCALE = 1e-2                # 100x the shipped 1e-4 (configs/sample/sample_MolDiff.yml): the shift dominates fp32 position rounding
NOISE_SEED = 4242


class Draws:
    """Replacement for the RNG entry points: every call returns (and records) the next array of one seeded generator."""

    def __init__(self, seed):
        self.g = np.random.Generator(np.random.PCG64(seed))
        self.log = []

    def randn(self, *shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)) else shape
        a = torch.from_numpy(self.g.standard_normal(tuple(shape), dtype=np.float32))
        self.log.append(('randn', a))
        return a

    def randn_like(self, x, *a, **k):
        r = torch.from_numpy(self.g.standard_normal(tuple(x.shape), dtype=np.float32)).to(x.dtype)
        self.log.append(('randn_like', r))
        return r

    def rand_like(self, x, *a, **k):
        r = torch.from_numpy(self.g.random(tuple(x.shape), dtype=np.float64)).to(x.dtype)
        if x.dtype == torch.float32:   # keep u < 1 after the cast
            r = r.clamp(max=float(np.nextafter(np.float32(1), np.float32(0))))
        self.log.append(('rand_like', r))
        return r


def run_reference(model, bond, gr, guidance, mm):
    bn, hei, bh = gr
    d = Draws(NOISE_SEED)
    saved = (torch.randn, torch.randn_like, torch.rand_like, mm.tqdm)
    torch.randn, torch.randn_like, torch.rand_like = d.randn, d.randn_like, d.rand_like
    mm.tqdm = lambda it, total=None: itertools.islice(it, FIRST, FIRST + NSTEPS)
    try:
        out = model.sample(n_graphs=int(bn.max()) + 1, batch_node=bn, halfedge_index=hei, batch_halfedge=bh,
                           bond_predictor=bond, guidance=guidance)
    finally:
        torch.randn, torch.randn_like, torch.rand_like, mm.tqdm = saved
    return out, d.log


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    import models.model as mm   # the reference module whose namespace holds `tqdm`
    cfg_full = ref_shim.load_yaml_cfg('configs/train/train_MolDiff.yml')
    cfg_bond = ref_shim.load_yaml_cfg('configs/train/train_bondpred.yml')

    def build(cls, cfg, kn, ke, seed):
        m = cls(cfg.model, kn, ke).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd2 = dict(sd)
        sd2.update(O.recipe_state_dict(shapes, seed))
        m.load_state_dict(sd2, strict=True)
        return m, {k: v.detach().clone() for k, v in m.state_dict().items()}

    m_full, P_full = build(MolDiff, cfg_full, 8, 6, SEED_MOLDIFF)
    m_bond, P_bond = build(BondPredictor, cfg_bond, 8, 5, SEED_BONDPRED)
    CFG = dict(num_timesteps=1000, num_blocks=6, cutoff=15)
    CFGB = dict(num_timesteps=1000, num_blocks=8, cutoff=20)
    tabs = {'pos': {k: P_full['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std')},
            'node': {k: P_full['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
            'edge': {k: P_full['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}

    out = {'first': FIRST, 'nsteps': NSTEPS, 'scale': SCALE, 'types': np.array(O.GUIDANCE_TYPES)}
    pinned = {}
    ph = graph(4)
    for tag, gr, sizes in (('n12', two_mol_graph()[:3], np.array([5, 7])), ('n101', ph[1:4], ph[0]['n_nodes_list'])):
        bn, hei, bh = gr
        N, Eh = len(bn), len(bh)
        out[f'{tag}_sizes'] = np.asarray(sizes)
        base, log0 = run_reference(m_full, m_bond, gr, None, mm)
        # draw order of sample(): node prior (rand_like f64), pos prior (randn), halfedge prior (rand_like f64), then per
        # executed iteration: eps_pos (randn_like), u_node (rand_like), u_halfedge (rand_like)
        kinds = [k for k, _ in log0]
        assert kinds == ['rand_like', 'randn', 'rand_like'] + ['randn_like', 'rand_like', 'rand_like'] * NSTEPS, kinds
        traj = base['traj']
        out[f'{tag}_init_node_type'] = traj[0][0].argmax(-1).numpy().astype(np.uint8)
        out[f'{tag}_init_pos'] = traj[1][0].numpy()
        out[f'{tag}_init_halfedge_type'] = traj[2][0].argmax(-1).numpy().astype(np.uint8)
        for j in range(NSTEPS):
            out[f'{tag}_{j}_eps_pos'] = log0[3 + 3 * j][1].numpy()
            out[f'{tag}_{j}_u_node'] = log0[4 + 3 * j][1].numpy()
            out[f'{tag}_{j}_u_halfedge'] = log0[5 + 3 * j][1].numpy()
            out[f'{tag}_{j}_none_pos'] = traj[1][FIRST + 1 + j].numpy()
            out[f'{tag}_{j}_none_node_type'] = traj[0][FIRST + 1 + j].argmax(-1).numpy().astype(np.uint8)
            out[f'{tag}_{j}_none_halfedge_type'] = traj[2][FIRST + 1 + j].argmax(-1).numpy().astype(np.uint8)
        graph_d = {'batch_node': bn, 'halfedge_index': hei, 'batch_halfedge': bh, 'n_graphs': int(bn.max()) + 1}
        for gt in O.GUIDANCE_TYPES:
            res, log = run_reference(m_full, m_bond, gr, [gt, SCALE], mm)
            assert all(torch.equal(a[1], b[1]) for a, b in zip(log, log0)) and len(log) == len(log0)
            # the oracle, free-running from the same prior with the same noise, must land on the reference's frames
            st = {'h_node': traj[0][0].clone(), 'pos': traj[1][0].clone(), 'h_halfedge': traj[2][0].clone(),
                  'log_node': torch.log(traj[0][0].clamp(min=1e-30)), 'log_halfedge': torch.log(traj[2][0].clamp(min=1e-30))}
            for j in range(NSTEPS):
                step = 999 - (FIRST + j)
                noise = {'eps_pos': log0[3 + 3 * j][1], 'u_node': log0[4 + 3 * j][1], 'u_halfedge': log0[5 + 3 * j][1]}
                with torch.no_grad():
                    new, _ = O.sample_step(P_full, CFG, tabs, st, graph_d, step, noise, Pb=P_bond, cfgb=CFGB, guidance=[gt, SCALE])
                ref_pos = res['traj'][1][FIRST + 1 + j]
                d = float((new['pos'] - ref_pos).abs().max())
                pinned[f'{tag}_{gt}_{j}'] = d
                assert torch.equal(new['h_halfedge'], res['traj'][2][FIRST + 1 + j]) and torch.equal(new['h_node'], res['traj'][0][FIRST + 1 + j])
                out[f'{tag}_{j}_{gt}_pos'] = ref_pos.numpy()
                if j == 0:   # the unguided quantities of iteration 0 are common to all runs; keep the posteriors once
                    out[f'{tag}_0_log_node'] = new['log_node'].numpy()
                    out[f'{tag}_0_log_halfedge'] = new['log_halfedge'].numpy()
                st = {'h_node': res['traj'][0][FIRST + 1 + j].clone(), 'pos': ref_pos.clone(), 'h_halfedge': res['traj'][2][FIRST + 1 + j].clone(),
                      'log_node': new['log_node'], 'log_halfedge': new['log_halfedge']}
            shift = float((res['traj'][1][FIRST + 1] - traj[1][FIRST + 1]).abs().max())
            print(f'{tag} {gt:17s} |shift| {shift:.3e}   oracle-vs-reference {pinned[f"{tag}_{gt}_0"]:.2e} {pinned[f"{tag}_{gt}_1"]:.2e}')
    np.savez_compressed(OUT, **out)
    pj = os.path.join(ROOT, 'tests', 'golden', 'PINNING.json')
    rec = json.load(open(pj))
    rec['guidance_types_pos_max'] = max(pinned.values())
    rec['guidance_types_detail'] = pinned
    json.dump(rec, open(pj, 'w'), indent=1, sort_keys=True)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes; worst oracle-vs-reference position difference', max(pinned.values()))


if __name__ == '__main__':
    main()
