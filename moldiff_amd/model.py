"""MolDiff: joint diffusion over atom types, positions and bond types -- MI355X-native sampling path.

Drop-in for the reference's ``models/model.py`` ``MolDiff`` on the path BASELINE.json names:
``__init__`` (:13-46), ``define_betas_alphas`` (:49-95), ``forward`` (:204-234), ``sample`` (:236-378).
Same constructor arguments, same ``state_dict`` keys/shapes (strict checkpoint load), same call
signatures and return layout.  ``get_loss`` (training, :128-201) is outside this round's scope and raises.

All arithmetic runs in ``libmoldiff_hip.so``; torch is used for device memory, streams and the output
containers only.  Differences a caller can see, all opt-in keyword arguments with reference defaults:
``sample(..., seed=, mol_ids=, noise=, return_traj=)`` (per-molecule Philox noise instead of torch's
global generator, see DESIGN.md "noise").
"""
import ctypes

import torch
import torch.nn as nn
from torch.nn import Module

from . import _lib
from .common import MLP, GaussianSmearing
from .diffusion import get_beta_schedule
from .graph import NodeEdgeNet, _sig
from .transition import ContigousTransition, GeneralCategoricalTransition


class MolDiff(Module):
    def __init__(self, config, num_node_types, num_edge_types, **kwargs):
        super().__init__()
        self.config = config
        self.num_node_types = num_node_types
        self.num_edge_types = num_edge_types
        self.bond_len_loss = getattr(config, 'bond_len_loss', False)
        self.define_betas_alphas(config.diff)
        node_dim, edge_dim, time_dim = config.node_dim, config.edge_dim, config.diff.time_dim
        self.node_embedder = nn.Linear(num_node_types, node_dim - time_dim, bias=False)
        self.edge_embedder = nn.Linear(num_edge_types, edge_dim - time_dim, bias=False)
        self.time_emb = nn.Sequential(
            GaussianSmearing(stop=self.num_timesteps, num_gaussians=time_dim, type_='linear'))
        if config.denoiser.backbone == 'NodeEdgeNet':
            self.denoiser = NodeEdgeNet(node_dim, edge_dim, **config.denoiser)
        else:
            raise NotImplementedError(config.denoiser.backbone)
        self.node_decoder = MLP(node_dim, num_node_types, node_dim)
        self.edge_decoder = MLP(edge_dim, num_edge_types, edge_dim)
        self._eng = None
        self._eng_sig = None

    def define_betas_alphas(self, config):
        self.num_timesteps = config.num_timesteps
        self.categorical_space = getattr(config, 'categorical_space', 'discrete')
        if self.categorical_space != 'discrete':
            raise NotImplementedError("categorical_space='continuous' is not built (no shipped config uses it)")
        self.scaling = [1., 1., 1.]
        T = self.num_timesteps
        self.pos_transition = ContigousTransition(get_beta_schedule(num_timesteps=T, **config.diff_pos))
        self.node_transition = GeneralCategoricalTransition(
            get_beta_schedule(num_timesteps=T, **config.diff_atom), self.num_node_types,
            init_prob=config.diff_atom.init_prob)
        self.edge_transition = GeneralCategoricalTransition(
            get_beta_schedule(num_timesteps=T, **config.diff_bond), self.num_edge_types,
            init_prob=config.diff_bond.init_prob)

    # ---- engine ---------------------------------------------------------------------------------
    def _engine(self):
        sig = _sig(self)
        if self._eng is None or sig != self._eng_sig:
            d = self.denoiser
            eng = _lib.Model(_lib.MDX_KIND_MOLDIFF, num_blocks=d.num_blocks, cutoff=d.cutoff, update_pos=d.update_pos,
                             time_dim=self.config.diff.time_dim, num_timesteps=self.num_timesteps,
                             num_node_types=self.num_node_types, num_edge_types=self.num_edge_types,
                             node_dim=d.node_dim, edge_dim=d.edge_dim,
                             num_gaussians=d.distance_expansion.offset.numel())
            eng.upload(self.state_dict())
            self._eng, self._eng_sig = eng, sig
        return self._eng

    def get_loss(self, *args, **kwargs):
        raise NotImplementedError('training loss is outside the sampling hot path built so far (SURVEY.md section 8(f))')

    def _forward_raw(self, eng, g, h_node_pert, pos_pert, h_edge_pert, h_halfedge_pert, t, out=None):
        dev = pos_pert.device
        N, Eh = g.N, g.Eh
        if out is None:
            out = (torch.empty(N, self.num_node_types, dtype=torch.float32, device=dev),
                   torch.empty(N, 3, dtype=torch.float32, device=dev),
                   torch.empty(Eh, self.num_edge_types, dtype=torch.float32, device=dev))
        ws, nb = g.workspace(dev)
        _lib.check(_lib.lib().mdx_moldiff_forward(
            eng.h, g.h, _lib.ptr(h_node_pert), _lib.ptr(pos_pert), _lib.ptr(h_edge_pert), _lib.ptr(h_halfedge_pert),
            _lib.ptr(t), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), ws, nb, _lib.stream()))
        return out

    def forward(self, h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, batch_edge, t):
        """Predict the clean molecule from the perturbed one at per-graph step `t`."""
        _lib._need_gpu(h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, t)
        eng = self._engine()
        g = _lib.graph_for(edge_index, batch_node, int(t.numel()))
        pn, pp, ph = self._forward_raw(eng, g, _lib.f32c(h_node_pert), _lib.f32c(pos_pert), _lib.f32c(h_edge_pert), None,
                                       _lib.i64c(t))
        return {'pred_node': pn, 'pred_pos': pp, 'pred_halfedge': ph}

    @torch.no_grad()
    def sample(self, n_graphs, batch_node, halfedge_index, batch_halfedge, bond_predictor=None, guidance=None, *,
               seed=None, mol_ids=None, noise=None, return_traj=True):
        """Run the T-step reverse chain for a packed batch of fully-connected molecule graphs.

        Returns {'pred': [node logits (N,Kn), pos (N,3), halfedge logits (Eh,Ke)] of the last step,
                 'traj': [(T+1,N,Kn), (T+1,N,3), (T+1,Eh,Ke)]} exactly like the reference.
        seed: noise key (default: drawn from torch's global generator, so torch.manual_seed governs it);
        mol_ids: global molecule ids (noise is keyed per molecule => results do not depend on sharding);
        noise: optional callable draw -> (eps_pos, u_node, u_halfedge) to inject explicit noise (tests);
        return_traj=False skips the (large) trajectory buffers.
        """
        _lib._need_gpu(batch_node, halfedge_index, batch_halfedge)
        if guidance is not None and guidance[1] > 0:
            raise NotImplementedError('bond-predictor guidance is not built yet (SURVEY.md section 8 rows a14/a15)')
        dev = batch_node.device
        T, Kn, Ke = self.num_timesteps, self.num_node_types, self.num_edge_types
        N, Eh = int(batch_node.numel()), int(batch_halfedge.numel())
        eng = self._engine()
        edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
        g = _lib.Graph(edge_index, batch_node, n_graphs, mol_ids)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        eps, u_n, u_h = torch.empty(N, 3, **f32), torch.empty(N, Kn, **f32), torch.empty(Eh, Ke, **f32)

        def draw(i):
            if noise is not None:
                e, a, b = noise(i)
                eps.copy_(e); u_n.copy_(a); u_h.copy_(b)
            else:
                _lib.check(L.mdx_noise(g.h, ctypes.c_uint64(seed), i, Kn, Ke, _lib.ptr(eps), _lib.ptr(u_n), _lib.ptr(u_h),
                                       _lib.stream()))

        nT = T + 1 if return_traj else 1
        node_traj = torch.zeros(nT, N, Kn, **f32)
        pos_traj = torch.zeros(nT, N, 3, **f32)
        halfedge_traj = torch.zeros(nT, Eh, Ke, **f32)

        # prior
        draw(0)
        tn, te = self.node_transition, self.edge_transition
        ln0 = torch.log(torch.from_numpy(tn.init_prob).float() + 1e-30).clamp_min(-32.).to(dev).unsqueeze(0).repeat(N, 1)
        lh0 = torch.log(torch.from_numpy(te.init_prob).float() + 1e-30).clamp_min(-32.).to(dev).unsqueeze(0).repeat(Eh, 1)
        h_node = node_traj[0]
        h_half = halfedge_traj[0]
        _lib.check(L.mdx_gumbel_argmax(_lib.ptr(ln0), _lib.ptr(u_n), Kn, N, None, _lib.ptr(h_node), _lib.stream()))
        _lib.check(L.mdx_gumbel_argmax(_lib.ptr(lh0), _lib.ptr(u_h), Ke, Eh, None, _lib.ptr(h_half), _lib.stream()))
        log_node = torch.log(h_node.clamp(min=1e-30))
        log_half = torch.log(h_half.clamp(min=1e-30))
        pos_traj[0].copy_(eps)
        pos = pos_traj[0]

        t = torch.empty(n_graphs, dtype=torch.int64, device=dev)
        bn, bh = _lib.i64c(batch_node), _lib.i64c(batch_halfedge)
        preds = (torch.empty(N, Kn, **f32), torch.empty(N, 3, **f32), torch.empty(Eh, Ke, **f32))
        log_node_new, log_half_new = torch.empty_like(log_node), torch.empty_like(log_half)
        pt, ntr, etr = self.pos_transition, self.node_transition, self.edge_transition
        for i, step in enumerate(range(T)[::-1]):
            t.fill_(step)
            draw(i + 1)
            self._forward_raw(eng, g, h_node, pos, None, h_half, t, out=preds)
            j = i + 1 if return_traj else 0
            pos_new, h_node_new, h_half_new = pos_traj[j], node_traj[j], halfedge_traj[j]
            if not return_traj:  # single frame: ping-pong through temporaries
                pos_new, h_node_new, h_half_new = torch.empty_like(pos), torch.empty_like(h_node), torch.empty_like(h_half)
            _lib.check(L.mdx_pos_posterior(_lib.ptr(pt.coef_x0), _lib.ptr(pt.coef_xt), _lib.ptr(pt.std), _lib.ptr(pos),
                                           _lib.ptr(preds[1]), _lib.ptr(eps), _lib.ptr(t), _lib.ptr(bn), N,
                                           _lib.ptr(pos_new), _lib.stream()))
            _lib.check(L.mdx_cat_posterior(_lib.ptr(ntr.q_mats), _lib.ptr(ntr.transpopse_q_onestep_mats), Kn, T,
                                           _lib.ptr(preds[0]), 1, _lib.ptr(log_node), _lib.ptr(t), _lib.ptr(bn), N,
                                           _lib.ptr(log_node_new), _lib.stream()))
            _lib.check(L.mdx_gumbel_argmax(_lib.ptr(log_node_new), _lib.ptr(u_n), Kn, N, None, _lib.ptr(h_node_new),
                                           _lib.stream()))
            _lib.check(L.mdx_cat_posterior(_lib.ptr(etr.q_mats), _lib.ptr(etr.transpopse_q_onestep_mats), Ke, T,
                                           _lib.ptr(preds[2]), 1, _lib.ptr(log_half), _lib.ptr(t), _lib.ptr(bh), Eh,
                                           _lib.ptr(log_half_new), _lib.stream()))
            _lib.check(L.mdx_gumbel_argmax(_lib.ptr(log_half_new), _lib.ptr(u_h), Ke, Eh, None, _lib.ptr(h_half_new),
                                           _lib.stream()))
            pos, h_node, h_half = pos_new, h_node_new, h_half_new
            log_node, log_node_new = log_node_new, log_node
            log_half, log_half_new = log_half_new, log_half
        if not return_traj:
            node_traj, pos_traj, halfedge_traj = h_node[None], pos[None], h_half[None]
        return {'pred': [preds[0], preds[1], preds[2]],
                'traj': [node_traj, pos_traj, halfedge_traj]}
