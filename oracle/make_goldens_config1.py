"""Golden vectors for BASELINE.json config #1 (sample_MolDiff_simple, batch_size = 8, 100 diffusion steps), written by the REAL
reference on CPU.  BUILD-CONTAINER ONLY (needs /root/reference, read-only).

    python oracle/make_goldens_config1.py        ->  tests/golden/config1_T100.npz

The model is the reference's MolDiff built from configs/train/train_MolDiff_simple.yml with diff.num_timesteps = 100 and
recipe weights (seed in tests/golden/state_dict_keys.json); the batch is the reference's size recipe
(utils/transforms.py:128-131) with numpy seed 2920 for 8 molecules.  The reference's own modules run the whole 100-step
chain (loop body models/model.py:272-308) with the random draws replaced by the Philox stream of tests/philox_ref.py
(seed 2023, draw = loop index + 1, molecule ids 0..7) -- the generator the product uses on the device, restated on the host.
Stored: the T = 100 schedule tables, and at six checkpoints of the chain the state a step starts from and everything the
step produces, so the GPU test can run that one step teacher-forced.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import moldiff_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle.make_goldens import reference_step, SEED_MOLDIFF  # noqa: E402
from tests.philox_ref import noise_ref  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'config1_T100.npz')
T, B, SEED = 100, 8, 2023
CHECK = [99, 80, 60, 40, 20, 0]   # time steps whose single step is stored


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    cfg = ref_shim.load_yaml_cfg('configs/train/train_MolDiff_simple.yml')
    cfg.model.diff.num_timesteps = T
    m = MolDiff(cfg.model, 8, 6).eval()
    sd = m.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
    sd2 = dict(sd)
    sd2.update(O.recipe_state_dict(shapes, SEED_MOLDIFF))
    m.load_state_dict(sd2, strict=True)

    out = {'T': T, 'seed': SEED, 'check': np.array(CHECK)}
    for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar'):
        out['pos_' + k] = getattr(m.pos_transition, k).numpy()
    for part, tr in (('node', m.node_transition), ('edge', m.edge_transition)):
        out[part + '_q_mats'] = tr.q_mats.numpy()
        out[part + '_transpopse_q_onestep_mats'] = tr.transpopse_q_onestep_mats.numpy()

    # the reference's size recipe (utils/transforms.py:128-131): numpy legacy RNG, normal(mean, std) truncated to int
    np.random.seed(2920)
    sizes = np.random.normal(24.923464980477522, 5.516291901819105, size=B).astype('int64')
    out['sizes'] = sizes
    bn, hei, bh = [], [], []
    off = 0
    for i, n in enumerate(sizes):
        iu, ju = np.triu_indices(int(n), k=1)
        bn += [i] * int(n)
        hei.append(np.stack([iu + off, ju + off]))
        bh += [i] * len(iu)
        off += int(n)
    bn, hei, bh = torch.tensor(bn), torch.from_numpy(np.concatenate(hei, 1)), torch.tensor(bh)
    N, Eh = len(bn), len(bh)
    ids = np.arange(B)

    # prior (draw 0): positions = eps; classes = Gumbel-max over log(init_prob) like sample_init (transition.py:331-338)
    e0, un0, uh0 = noise_ref(SEED, 0, sizes, ids, 8, 6)
    st = {'pos': torch.from_numpy(e0)}
    for part, tr, u, K in (('node', m.node_transition, un0, 8), ('halfedge', m.edge_transition, uh0, 6)):
        logp = torch.log(torch.from_numpy(np.asarray(tr.init_prob, dtype=np.float64)).float()).expand(len(u), K)
        g = -torch.log(-torch.log(torch.from_numpy(u) + 1e-30) + 1e-30)
        cls = (g + logp).argmax(-1)
        st['h_' + part] = tr.onehot_encode(cls)
        st['log_' + part] = DF.index_to_log_onehot(cls, K)
    for i in range(T):
        step = T - 1 - i
        e, un, uh = noise_ref(SEED, i + 1, sizes, ids, 8, 6)
        noise = {'eps_pos': torch.from_numpy(e), 'u_node': torch.from_numpy(un), 'u_halfedge': torch.from_numpy(uh)}
        new, preds = reference_step(m, DF, st, bn, hei, bh, step, noise, None, None, B=B)
        if step in CHECK:
            p = f't{step}_'
            out[p + 'in_node_type'] = st['h_node'].argmax(-1).numpy().astype(np.uint8)
            out[p + 'in_halfedge_type'] = st['h_halfedge'].argmax(-1).numpy().astype(np.uint8)
            out[p + 'in_pos'] = st['pos'].numpy()
            out[p + 'in_log_node'] = st['log_node'].numpy()
            out[p + 'in_log_halfedge'] = st['log_halfedge'].numpy()
            out[p + 'pred_pos'] = preds['pred_pos'].numpy()
            out[p + 'pred_node'] = preds['pred_node'].numpy()
            out[p + 'pos'] = new['pos'].numpy()
            out[p + 'log_node'] = new['log_node'].numpy()
            out[p + 'log_halfedge'] = new['log_halfedge'].numpy()
            out[p + 'node_type'] = new['node_type'].numpy().astype(np.uint8)
            out[p + 'halfedge_type'] = new['halfedge_type'].numpy().astype(np.uint8)
            for part, lg, u in (('node', new['log_node'], noise['u_node']), ('halfedge', new['log_halfedge'], noise['u_halfedge'])):
                gz = lg - torch.log(-torch.log(u + 1e-30) + 1e-30)
                top = gz.topk(2, -1).values
                out[p + part + '_margin_min'] = float((top[:, 0] - top[:, 1]).min())
        st = {k: new[k] for k in ('h_node', 'pos', 'h_halfedge', 'log_node', 'log_halfedge')}
    out['final_pos'] = st['pos'].numpy()
    out['final_node_type'] = st['h_node'].argmax(-1).numpy().astype(np.uint8)
    out['final_halfedge_type'] = st['h_halfedge'].argmax(-1).numpy().astype(np.uint8)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes; N, Eh =', N, Eh, '; margins',
          {k: round(float(v), 6) for k, v in out.items() if k.endswith('margin_min')})


if __name__ == '__main__':
    main()
