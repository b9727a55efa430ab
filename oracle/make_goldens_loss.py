"""Golden vectors for `MolDiff.get_loss` (models/model.py:128-201), the quantity the reference's validation loop
evaluates under no_grad (scripts/train_drug3d.py:121-164).  BUILD-CONTAINER ONLY (needs /root/reference).

The REAL reference model is run on CPU with recipe weights; its three sources of randomness are pinned for the
duration of the call by temporarily replacing the torch entry points it draws from:
  torch.randint        (sample_time, model.py:99-100)         -> the committed half-list of time steps
  Tensor.normal_       (ContigousTransition.add_noise :35-36) -> committed eps_pos
  torch.rand_like      (log_sample_categorical, diffusion.py:80) -> committed u_node, then u_halfedge
Only inputs and the resulting losses are saved (tests/golden/loss.npz); the oracle restatement
(`moldiff_oracle.moldiff_loss`) is compared in the same run and the difference recorded in PINNING.json.
The same call is then differentiated with the reference's own autograd: the gradient of `loss` wrt every trainable
parameter is the golden for the training path (tests/golden/loss_grads.npz: the L2 norm of every gradient tensor and
the full tensor where it has <= 256 elements), and autograd through the oracle is pinned against it.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import moldiff_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
SEED_MOLDIFF = 20230807


class pinned_randomness:
    def __init__(self, t_half, eps_pos, us):
        self.t_half, self.eps_pos, self.us = t_half, eps_pos, list(us)

    def __enter__(self):
        self.saved = (torch.randint, torch.Tensor.normal_, torch.rand_like)
        t_half, eps_pos, us = self.t_half, self.eps_pos, self.us

        def randint(lo, hi, size, device=None, **kw):
            assert tuple(size) == tuple(t_half.shape), (size, t_half.shape)
            return t_half.clone()

        def normal_(self_, *a, **kw):
            assert self_.shape == eps_pos.shape
            return self_.copy_(eps_pos)

        def rand_like(x, **kw):
            u = us.pop(0)
            assert u.shape == x.shape
            return u.to(x.dtype)

        torch.randint, torch.Tensor.normal_, torch.rand_like = randint, normal_, rand_like
        return self

    def __exit__(self, *exc):
        torch.randint, torch.Tensor.normal_, torch.rand_like = self.saved
        assert not self.us, 'a pinned uniform draw was not consumed'


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    out, pins, grads_out = {}, {}, {}
    for nm, yml in (('full', 'configs/train/train_MolDiff.yml'), ('simple', 'configs/train/train_MolDiff_simple.yml')):
        cfg = ref_shim.load_yaml_cfg(yml)
        m = MolDiff(cfg.model, 8, 6).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd.update(O.recipe_state_dict(shapes, SEED_MOLDIFF))
        m.load_state_dict(sd, strict=True)
        P = {k: v.detach().clone() for k, v in m.state_dict().items()}

        g = np.random.Generator(np.random.PCG64(77 if nm == 'full' else 78))
        sizes = [7, 5, 12, 9, 3]
        bn, hei, bh, off = [], [], [], 0
        for i, n in enumerate(sizes):
            bn += [i] * n
            tri = torch.triu_indices(n, n, 1) + off
            hei.append(tri)
            bh += [i] * tri.shape[1]
            off += n
        bn, hei, bh = torch.tensor(bn), torch.cat(hei, 1), torch.tensor(bh)
        N, Eh, B = len(bn), len(bh), len(sizes)
        node_type = torch.from_numpy(g.integers(0, 7, N))
        node_pos = torch.from_numpy(g.standard_normal((N, 3)).astype(np.float32) * 2.0)
        for i in range(B):   # centre per molecule like the training transforms do
            node_pos[bn == i] -= node_pos[bn == i].mean(0, keepdim=True)
        half_type = torch.from_numpy((g.random(Eh) < 0.25) * g.integers(1, 5, Eh))
        # sample_time draws B//2+1 steps and mirrors them: [0, 437, 850] -> t = [0, 437, 850, 999, 562]
        t_half = torch.tensor([0, 437, 850])
        eps_pos = torch.from_numpy(g.standard_normal((N, 3)).astype(np.float32))
        u_node = torch.from_numpy(g.random((N, 8)).astype(np.float32))
        u_half = torch.from_numpy(g.random((Eh, 6)).astype(np.float32))
        with pinned_randomness(t_half, eps_pos, [u_node, u_half]):
            ref = m.get_loss(node_type, node_pos, bn, half_type, hei, bh, B)
        m.zero_grad()
        ref['loss'].backward()          # the reference's own autograd: golden parameter gradients
        ref_grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        no_grad = [k for k, v in m.named_parameters() if v.requires_grad and v.grad is None]
        print('trainable parameters the loss does not reach:', no_grad)
        ref = {k: v.detach() for k, v in ref.items()}
        t = torch.cat([t_half, 1000 - t_half - 1])[:B]
        tabs = {'pos': {k: P['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar')},
                'node': {k: P['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
                'edge': {k: P['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}
        cfgd = dict(num_timesteps=1000, num_blocks=6, cutoff=15)
        Pg = {k: (v.clone().requires_grad_(True) if k in ref_grads else v) for k, v in P.items()}
        orc = O.moldiff_loss(Pg, cfgd, tabs, node_type, node_pos, bn, half_type, hei, bh, B, t,
                             dict(eps_pos=eps_pos, u_node=u_node, u_halfedge=u_half))
        orc['loss'].backward()
        gd = max(float((Pg[k].grad - g).abs().max()) for k, g in ref_grads.items())
        gs = max(float(g.abs().max()) for g in ref_grads.values())
        pins[f'get_loss_{nm}_param_grads'] = gd
        print(nm, 'param grads: oracle-vs-reference max abs diff', gd, 'of max |g|', gs, 'over', len(ref_grads), 'tensors')
        orc = {k: v.detach() for k, v in orc.items()}
        for k, g in ref_grads.items():
            grads_out[f'{nm}/norm/{k}'] = np.float64(g.double().norm())
            if g.numel() <= 256:
                grads_out[f'{nm}/full/{k}'] = g.numpy()
        for k in ('loss', 'loss_pos', 'loss_node', 'loss_edge'):
            d = abs(float(ref[k]) - float(orc[k]))
            pins[f'get_loss_{nm}_{k}'] = d
            print(nm, k, float(ref[k]), float(orc[k]), d)
            out[f'{nm}_{k}'] = np.float32(float(ref[k]))
        out.update({f'{nm}_sizes': np.array(sizes), f'{nm}_node_type': node_type.numpy(), f'{nm}_node_pos': node_pos.numpy(),
                    f'{nm}_halfedge_type': half_type.numpy(), f'{nm}_t_half': t_half.numpy(), f'{nm}_t': t.numpy(),
                    f'{nm}_eps_pos': eps_pos.numpy(), f'{nm}_u_node': u_node.numpy(), f'{nm}_u_halfedge': u_half.numpy()})
    # ---- BondPredictor.get_loss (models/bond_predictor.py:84-124): draws = randint, normal_ (pos), rand_like (node) ----
    cfg = ref_shim.load_yaml_cfg('configs/train/train_bondpred.yml')
    m = BondPredictor(cfg.model, 8, 5).eval()
    sd = m.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
    sd.update(O.recipe_state_dict(shapes, 20230808))
    m.load_state_dict(sd, strict=True)
    Pb = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = np.random.Generator(np.random.PCG64(79))
    sizes = [7, 5, 12, 9, 3]
    bn, hei, bh, off = [], [], [], 0
    for i, n in enumerate(sizes):
        bn += [i] * n
        tri = torch.triu_indices(n, n, 1) + off
        hei.append(tri)
        bh += [i] * tri.shape[1]
        off += n
    bn, hei, bh = torch.tensor(bn), torch.cat(hei, 1), torch.tensor(bh)
    N, Eh, B = len(bn), len(bh), len(sizes)
    node_type = torch.from_numpy(g.integers(0, 7, N))
    node_pos = torch.from_numpy(g.standard_normal((N, 3)).astype(np.float32) * 2.0)
    half_type = torch.from_numpy((g.random(Eh) < 0.25) * g.integers(1, 5, Eh))
    t_half = torch.tensor([0, 311, 742])
    eps_pos = torch.from_numpy(g.standard_normal((N, 3)).astype(np.float32))
    u_node = torch.from_numpy(g.random((N, 8)).astype(np.float32))
    with pinned_randomness(t_half, eps_pos, [u_node]):
        ref = m.get_loss(node_type, node_pos, bn, half_type, hei, bh, B)
    m.zero_grad()
    ref['loss'].backward()
    ref_grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
    print('trainable parameters the loss does not reach:', [k for k, v in m.named_parameters() if v.requires_grad and v.grad is None])
    ref = {k: v.detach() for k, v in ref.items()}
    t = torch.cat([t_half, 1000 - t_half - 1])[:B]
    tabs = {'pos': {'alphas_bar': Pb['pos_transition.alphas_bar']}, 'node': {'q_mats': Pb['node_transition.q_mats']}}
    Pg = {k: (v.clone().requires_grad_(True) if k in ref_grads else v) for k, v in Pb.items()}
    orc = O.bondpred_loss(Pg, dict(num_timesteps=1000, num_blocks=8, cutoff=20), tabs, node_type, node_pos, bn, half_type,
                          hei, bh, B, t, dict(eps_pos=eps_pos, u_node=u_node))
    orc['loss'].backward()
    gd = max(float((Pg[k].grad - g).abs().max()) for k, g in ref_grads.items())
    pins['get_loss_bondpred_param_grads'] = gd
    print('bondpred param grads: oracle-vs-reference max abs diff', gd, 'over', len(ref_grads), 'tensors')
    orc = {k: v.detach() for k, v in orc.items()}
    for k, g in ref_grads.items():
        grads_out[f'bond/norm/{k}'] = np.float64(g.double().norm())
        if g.numel() <= 256:
            grads_out[f'bond/full/{k}'] = g.numpy()
    np.savez_compressed(os.path.join(OUT, 'loss_grads.npz'), **grads_out)
    for k in ('loss', 'loss_edge'):
        pins[f'get_loss_bondpred_{k}'] = abs(float(ref[k]) - float(orc[k]))
        print('bondpred', k, float(ref[k]), float(orc[k]))
        out[f'bond_{k}'] = np.float32(float(ref[k]))
    out.update({'bond_sizes': np.array(sizes), 'bond_node_type': node_type.numpy(), 'bond_node_pos': node_pos.numpy(),
                'bond_halfedge_type': half_type.numpy(), 'bond_t_half': t_half.numpy(), 'bond_t': t.numpy(),
                'bond_eps_pos': eps_pos.numpy(), 'bond_u_node': u_node.numpy()})
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **out)
    pf = os.path.join(OUT, 'PINNING.json')
    allp = json.load(open(pf))
    allp.update(pins)
    with open(pf, 'w') as f:
        json.dump(allp, f, indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
