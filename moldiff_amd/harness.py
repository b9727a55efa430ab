"""Host-side harness pieces either side of the hot path.

make_data_placeholder  utils/transforms.py:125-156  (defines the workload: sizes ~ N(24.92, 5.52), fully
                       connected half-edges per molecule); same return keys, same global-RNG consumption.
load_config, seed_all  utils/misc.py:22-24, :68-71
recipe_state_dict      deterministic synthetic weights (no checkpoint is available offline); see DESIGN.md.
"""
import math
import random

import numpy as np
import torch
import yaml

from .common import AttrDict

GEOM_DRUGS_MEAN_ATOMS = 24.923464980477522
GEOM_DRUGS_STD_ATOMS = 5.516291901819105


def load_config(path):
    with open(path, 'r') as f:
        return AttrDict(yaml.safe_load(f))


def seed_all(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def placeholder_from_sizes(n_nodes_list, device=None):
    """Pack molecules of the given sizes: batch_node (N), halfedge_index (2,Eh) = per-molecule row-major
    upper triangle offset by the molecule's first node, batch_halfedge (Eh)."""
    sizes = np.asarray(n_nodes_list, dtype=np.int64)
    sizes_pos = np.maximum(sizes, 0)
    offsets = np.concatenate([[0], np.cumsum(sizes_pos)[:-1]]) if len(sizes) else np.zeros(0, dtype=np.int64)
    batch_node = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes_pos)
    rows, cols, owner = [], [], []
    for i, (n, off) in enumerate(zip(sizes_pos, offsets)):
        iu, ju = np.triu_indices(int(n), k=1)
        rows.append(iu + off)
        cols.append(ju + off)
        owner.append(np.full(iu.shape[0], i, dtype=np.int64))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
    out = {
        'batch_node': torch.from_numpy(batch_node),
        'halfedge_index': torch.from_numpy(np.stack([cat(rows), cat(cols)]).astype(np.int64)),
        'batch_halfedge': torch.from_numpy(cat(owner)),
    }
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out


def make_data_placeholder(n_graphs, device=None, max_size=None):
    if max_size is None:  # GEOM-Drugs atom-count statistics, numpy's legacy global RNG like the reference
        sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=n_graphs)
    else:
        sizes = np.array([max_size] * n_graphs)
    return placeholder_from_sizes(sizes.astype('int64'), device)


FROZEN_MARKERS = ('_transition.', '.coeff', '.offset', 'ce_loss.weight')


def is_frozen_key(k):
    return any(m in k for m in FROZEN_MARKERS)


def recipe_state_dict(module, seed):
    """Fill every learnable tensor of `module` deterministically: keys in SORTED order, one PCG64 stream;
    2-D weights ~ N(0,1)/sqrt(fan_in), LayerNorm gains 1 + 0.1 z, biases 0.1 z.  Returns the new state_dict."""
    sd = module.state_dict()
    g = np.random.Generator(np.random.PCG64(seed))
    for k in sorted(sd):
        if is_frozen_key(k):
            continue
        shp = tuple(sd[k].shape)
        z = g.standard_normal(shp, dtype=np.float32)
        if len(shp) == 2:
            w = z * np.float32(1.0 / math.sqrt(shp[1]))
        elif k.endswith('.weight'):
            w = np.float32(1.0) + np.float32(0.1) * z
        else:
            w = np.float32(0.1) * z
        sd[k] = torch.from_numpy(w.astype(np.float32)).to(sd[k].device)
    return sd


def stress_state_dict(module, seed):
    """Heavy-tailed "stress" weights on top of the base recipe (trained-weight parity cannot be pinned offline,
    this probes what a trained checkpoint may hold that N(0, small) weights do not): every LayerNorm gain log-uniform in
    [0.1, 30], every bias x 8, and block 2's two `out_transform` matrices x 16 so the residual streams reach 10^2 - 10^3.
    Deterministic: its own PCG64 stream (seed + 1), keys in sorted order."""
    sd = recipe_state_dict(module, seed)
    g = np.random.Generator(np.random.PCG64(seed + 1))
    for k in sorted(sd):
        if is_frozen_key(k):
            continue
        w = sd[k]
        if w.dim() == 1 and k.endswith('.weight'):      # LayerNorm gain
            u = g.random(tuple(w.shape), dtype=np.float32)
            sd[k] = torch.from_numpy(np.exp(np.float32(math.log(0.1)) + u * np.float32(math.log(30.0) - math.log(0.1))).astype(np.float32)).to(w.device)
        elif w.dim() == 1:                              # bias
            sd[k] = w * 8.0
        elif '_blocks.2.out_transform.weight' in k:
            sd[k] = w * 16.0
    return sd


def default_config(name):
    """Model hyper-parameters of the three shipped training configs (the `model:` section that a checkpoint
    carries as ckpt['config'].model): 'MolDiff' (configs/train/train_MolDiff.yml:1-36), 'MolDiff_simple'
    (train_MolDiff_simple.yml:1-30; differs only in the bond schedule) and 'bondpred' (train_bondpred.yml:1-27)."""
    adv = lambda: dict(beta_schedule='advance', scale_start=0.9999, scale_end=0.0001, width=3)
    if name in ('MolDiff', 'MolDiff_simple'):
        bond = dict(init_prob='absorb', **adv())
        if name == 'MolDiff':
            bond = dict(init_prob='absorb', beta_schedule='segment', time_segment=[600, 400],
                        segment_diff=[dict(scale_start=0.9999, scale_end=0.001, width=3),
                                      dict(scale_start=0.001, scale_end=0.0001, width=2)])
        return AttrDict(name='diffusion', node_dim=256, edge_dim=64,
                        denoiser=dict(backbone='NodeEdgeNet', num_blocks=6, cutoff=15, use_gate=True),
                        diff=dict(num_timesteps=1000, time_dim=10, categorical_space='discrete', diff_pos=adv(),
                                  diff_atom=dict(init_prob='tomask', **adv()), diff_bond=bond))
    if name == 'bondpred':
        return AttrDict(name='bond_predictor', node_dim=256, edge_dim=64,
                        encoder=dict(backbone='NodeEdgeNet', num_blocks=8, cutoff=20, use_gate=True, update_edge=True,
                                     update_pos=False),
                        diff=dict(num_timesteps=1000, time_dim=20, categorical_space='discrete', diff_pos=adv(),
                                  diff_atom=dict(init_prob='tomask', **adv())))
    raise KeyError(name)
