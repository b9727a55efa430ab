"""Golden vectors for the stand-alone forwards of the small modules, written by the REAL reference on CPU.
BUILD-CONTAINER ONLY (needs /root/reference, read-only).

    python oracle/make_goldens_dropin.py        ->  tests/golden/dropin.npz

  * `MLP.forward` (models/common.py:200-201): node_net of block 0 (256 -> 256 -> 256, 2 layers) on the 12 node rows of
    blocks_full.npz, and the bond predictor's 3-layer edge decoder (320 -> 64 -> 64 -> 5, models/bond_predictor.py:34);
  * stand-alone `BondFFN.forward` with out_dim = 1 (models/graph.py:133-141 as PosUpdate.edge_lin, :389): block 0's edge_lin on
    (bond (62,64), node (62,64), time (62,1)).
(`GaussianSmearing.forward` is already in smearing.npz.)  Weights: the recipe weights of make_goldens.py.
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
from oracle.make_goldens import SEED_BONDPRED, SEED_MOLDIFF, rng_inputs  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'dropin.npz')


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()

    def build(cls, cfg, kn, ke, seed):
        m = cls(cfg.model, kn, ke).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd2 = dict(sd)
        sd2.update(O.recipe_state_dict(shapes, seed))
        m.load_state_dict(sd2, strict=True)
        return m

    m_full = build(MolDiff, ref_shim.load_yaml_cfg('configs/train/train_MolDiff.yml'), 8, 6, SEED_MOLDIFF)
    m_bond = build(BondPredictor, ref_shim.load_yaml_cfg('configs/train/train_bondpred.yml'), 8, 5, SEED_BONDPRED)
    b = np.load(os.path.join(ROOT, 'tests', 'golden', 'blocks_full.npz'))
    g = rng_inputs(31)
    out = {}
    with torch.no_grad():
        x = torch.from_numpy(b['x'])
        out['mlp_node_net0_out'] = m_full.denoiser.node_blocks_with_edge[0].node_net(x).numpy()
        x320 = torch.from_numpy(g.standard_normal((31, 320), dtype=np.float32))
        out['x320'] = x320.numpy()
        out['mlp_bond_decoder_out'] = m_bond.edge_decoder(x320).numpy()
        bond = torch.from_numpy(b['edge_attr'])
        node = torch.from_numpy(g.standard_normal((62, 64), dtype=np.float32))
        tt = torch.from_numpy(g.random((62, 1), dtype=np.float32))
        out['edge_lin_node'], out['edge_lin_time'] = node.numpy(), tt.numpy()
        out['edge_lin0_out'] = m_full.denoiser.pos_blocks[0].edge_lin(bond, node, tt).numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
