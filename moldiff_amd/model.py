"""MolDiff: joint diffusion over atom types, positions and bond types -- MI355X-native sampling path.

Drop-in for the reference's ``models/model.py`` ``MolDiff`` on the path BASELINE.json names:
``__init__`` (:13-46), ``define_betas_alphas`` (:49-95), ``forward`` (:204-234), ``sample`` (:236-378).
Same constructor arguments, same ``state_dict`` keys/shapes (strict checkpoint load), same call
signatures and return layout.  ``get_loss`` (:128-201): under ``no_grad`` (the reference's validation loop) it runs on
the fused sampling kernels; with grad enabled it runs layer by layer on the differentiable HIP operators of
``train_ops`` / ``train_graph`` so that ``loss.backward()`` yields every parameter gradient.

All arithmetic runs in ``libmoldiff_hip.so``; torch is used for device memory, streams and the output
containers only.  Differences a caller can see, all opt-in keyword arguments with reference defaults:
``sample(..., seed=, mol_ids=, noise=, return_traj=)`` (per-molecule Philox noise instead of torch's
global generator, see DESIGN.md "noise").
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Module

from . import _lib

# guidance objectives of models/model.py:317-359
GUIDANCE_TYPES = ('entropy', 'uncertainty', 'uncertainty_bond', 'entropy_bond', 'logit_bond', 'logit', 'crossent',
                  'crossent_bond')
from .common import MLP, GaussianSmearing
from .diffusion import get_beta_schedule
from .graph import NodeEdgeNet, _sig, synth_gates
from .transition import ContigousTransition, GeneralCategoricalTransition

_FUSED_LOSS = os.environ.get('MDX_TRAIN_FUSED_LOSS', '1') != '0'   # training: categorical loss tail as one launch (train_ops.cat_loss)


class MolDiff(Module):
    def __getstate__(self):
        # the packed-weight engine is a device handle: never copied or pickled (deepcopy / torch.save of the module
        # rebuild it lazily from the state_dict on first use)
        d = self.__dict__.copy()
        d['_eng'], d['_eng_sig'] = None, None
        return d

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
        if self.categorical_space not in ('discrete', 'continuous'):
            raise ValueError(self.categorical_space)
        # continuous: classes are real vectors (one-hot / scaling) under Gaussian diffusion, models/model.py:54-56,76-78,91-93
        self.scaling = list(getattr(config, 'scaling', [1., 1., 1.])) if self.categorical_space == 'continuous' else [1., 1., 1.]
        assert self.scaling[0] == 1, 'scaling for pos should be 1'
        T = self.num_timesteps
        self.pos_transition = ContigousTransition(get_beta_schedule(num_timesteps=T, **config.diff_pos))
        node_betas = get_beta_schedule(num_timesteps=T, **config.diff_atom)
        edge_betas = get_beta_schedule(num_timesteps=T, **config.diff_bond)
        if self.categorical_space == 'discrete':
            self.node_transition = GeneralCategoricalTransition(node_betas, self.num_node_types, init_prob=config.diff_atom.init_prob)
            self.edge_transition = GeneralCategoricalTransition(edge_betas, self.num_edge_types, init_prob=config.diff_bond.init_prob)
        else:
            self.node_transition = ContigousTransition(node_betas, self.num_node_types, self.scaling[1])
            self.edge_transition = ContigousTransition(edge_betas, self.num_edge_types, self.scaling[2])

    # ---- engine ---------------------------------------------------------------------------------
    # None = follow _lib.default_matrix_path (exact fp32 unless MOLDIFF_MATRIX_PATH says otherwise); or 'exact_f32' / 'split_f16'
    matrix_path = None

    def _engine(self):
        sig = _sig(self)
        if self._eng is None or sig != self._eng_sig:
            d = self.denoiser
            eng = _lib.Model(_lib.MDX_KIND_MOLDIFF, num_blocks=d.num_blocks, cutoff=d.cutoff, update_pos=d.update_pos,
                             time_dim=self.config.diff.time_dim, num_timesteps=self.num_timesteps,
                             num_node_types=self.num_node_types, num_edge_types=self.num_edge_types,
                             node_dim=d.node_dim, edge_dim=d.edge_dim,
                             num_gaussians=16, smear_start=d.distance_expansion.start)
            eng.upload({**self.state_dict(), **synth_gates(d, 'denoiser.')})
            self._eng, self._eng_sig = eng, sig
        return self._eng.use_matrix_path(self.matrix_path)

    def sample_time(self, num_graphs, device, **kwargs):
        """Antithetic time-step draw: half the batch uniform in [0, T), the other half mirrored (T-1-t)."""
        T = self.num_timesteps
        ts = torch.randint(0, T, size=(num_graphs // 2 + 1,), device=device)
        ts = torch.cat([ts, T - ts - 1], dim=0)[:num_graphs]
        return ts, torch.ones_like(ts).float() / T

    @torch.no_grad()
    def add_noise(self, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge, num_mol, t,
                  bond_predictor=None, **kwargs):
        """Perturb a clean batch to step `t` (the same step for every molecule): [h_node_pert, pos_pert, h_halfedge_pert]
        (models/model.py:106-126)."""
        time_step = t * torch.ones(num_mol, device=node_pos.device).long()
        pos_pert = self.pos_transition.add_noise(node_pos, time_step, batch_node)
        h_node_pert = self.node_transition.add_noise(node_type, time_step, batch_node)[0]
        h_halfedge_pert = self.edge_transition.add_noise(halfedge_type, time_step, batch_halfedge)[0]
        return [h_node_pert, pos_pert, h_halfedge_pert]

    def get_loss(self, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge, num_mol, *,
                 time_step=None, noise=None):
        """Diffusion loss of a clean batch (models/model.py:128-201).

        Perturb (pos, node types, bond types) to a random step, denoise, and score: MSE on positions, 100 x mean per-row
        {KL of the categorical posteriors | decoder NLL at t == 0} on node and bond types.  Returns {'loss', 'loss_pos',
        'loss_node', 'loss_edge'} (0-d device tensors).
        * Under ``torch.no_grad()`` (the reference's validation loop, scripts/train_drug3d.py:121-164) the denoiser and
          the posteriors run in the fused sampling kernels and the result carries no graph.
        * With grad enabled the denoiser is evaluated layer by layer with the differentiable HIP operators of
          ``train_ops`` (``train_graph.moldiff_forward``) and ``loss.backward()`` fills every parameter's ``.grad``.
        In both modes the O(rows x classes) loss algebra on the logits is torch tensor ops.
        time_step (num_mol,) / noise = dict(eps_pos, u_node, u_halfedge) may be injected (parity tests); by default they
        are drawn from torch's generator in the reference's order.
        """
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        with torch.enable_grad() if train else torch.no_grad():
            return self._get_loss(train, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge,
                                  num_mol, time_step, noise)

    def _get_loss_continuous(self, train, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge, num_mol,
                             time_step, noise):
        """models/model.py:144-148,185-187: Gaussian perturbation of the scaled one-hot classes, 30 x MSE on the decoders' outputs.
        noise = dict(eps_pos, eps_node, eps_halfedge) may be injected."""
        noise = noise or {}
        with torch.no_grad():
            t = self.sample_time(num_mol, node_pos.device)[0] if time_step is None else time_step
            pos_pert = self.pos_transition.add_noise(node_pos, t, batch_node, noise.get('eps_pos'))
            h_node, h_node_0 = self.node_transition.add_noise(node_type, t, batch_node, noise.get('eps_node'))
            h_half, h_half_0 = self.edge_transition.add_noise(halfedge_type, t, batch_halfedge, noise.get('eps_halfedge'))
            edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
            batch_edge = torch.cat([batch_halfedge, batch_halfedge], dim=0)
            h_edge = torch.cat([h_half, h_half], dim=0)
        if train:
            from . import train_graph
            preds = train_graph.moldiff_forward(self, h_node, pos_pert, batch_node, h_edge, edge_index, batch_edge, t, flipped_halves=True)
        else:
            preds = self.forward(h_node, pos_pert, batch_node, h_edge, edge_index, batch_edge, t,
                                 _graph=_lib.graph_for_halfedges(halfedge_index, batch_node, int(t.numel())))
        loss_pos = F.mse_loss(preds['pred_pos'], node_pos)
        out = {'loss_node': F.mse_loss(preds['pred_node'], h_node_0) * 30, 'loss_edge': F.mse_loss(preds['pred_halfedge'], h_half_0) * 30}
        if self.bond_len_loss:
            bond_index = halfedge_index[:, halfedge_type > 0]
            true_len = torch.norm(node_pos[bond_index[0]] - node_pos[bond_index[1]], dim=-1)
            pred_len = torch.norm(preds['pred_pos'][bond_index[0]] - preds['pred_pos'][bond_index[1]], dim=-1)
            out['loss_len'] = F.mse_loss(pred_len, true_len)
        total = loss_pos + out['loss_node'] + out['loss_edge'] + out.get('loss_len', 0)
        return {'loss': total, 'loss_pos': loss_pos, **out}

    def _get_loss(self, train, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge, num_mol,
                  time_step, noise):
        if self.categorical_space == 'continuous':
            return self._get_loss_continuous(train, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge,
                                             num_mol, time_step, noise)
        dev = node_pos.device
        noise = noise or {}
        with torch.no_grad():
            t = self.sample_time(num_mol, dev)[0] if time_step is None else time_step
            pos_pert = self.pos_transition.add_noise(node_pos, t, batch_node, noise.get('eps_pos'))
            h_node, log_node_t, log_node_0 = self.node_transition.add_noise(node_type, t, batch_node, noise.get('u_node'))
            h_half, log_half_t, log_half_0 = self.edge_transition.add_noise(halfedge_type, t, batch_halfedge,
                                                                            noise.get('u_halfedge'))
            edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
            batch_edge = torch.cat([batch_halfedge, batch_halfedge], dim=0)
            h_edge = torch.cat([h_half, h_half], dim=0)
        if train:
            from . import train_graph
            preds = train_graph.moldiff_forward(self, h_node, pos_pert, batch_node, h_edge, edge_index, batch_edge, t, flipped_halves=True)
        else:
            # (plan keyed on the half-edge tensor: `edge_index` is a fresh torch.cat on every call)
            preds = self.forward(h_node, pos_pert, batch_node, h_edge, edge_index, batch_edge, t,
                                 _graph=_lib.graph_for_halfedges(halfedge_index, batch_node, int(t.numel())))

        loss_pos = F.mse_loss(preds['pred_pos'], node_pos)
        out = {}
        for name, tr, logits, log_t, log_0, batch in (
                ('loss_node', self.node_transition, preds['pred_node'], log_node_t, log_node_0, batch_node),
                ('loss_edge', self.edge_transition, preds['pred_halfedge'], log_half_t, log_half_0, batch_halfedge)):
            if train and _FUSED_LOSS and logits.is_cuda and 2 <= logits.shape[-1] <= 8 and logits.shape[0] > 0:
                from . import train_ops      # round 6: the whole tail and its backward as one launch (csrc cat_loss_kernel)
                out[name] = train_ops.cat_loss(tr, logits, log_t, log_0, t, batch)
                continue
            log_recon = F.log_softmax(logits, dim=-1)
            post_true = tr.q_v_posterior(log_0, log_t, t, batch, v0_prob=True)
            post_pred = (tr.q_v_posterior_autograd(log_recon, log_t, t, batch) if train
                         else tr.q_v_posterior(log_recon, log_t, t, batch, v0_prob=True))
            out[name] = torch.mean(tr.compute_v_Lt(post_true, post_pred, log_0, t=t, batch=batch)) * 100
        if self.bond_len_loss:
            bond_index = halfedge_index[:, halfedge_type > 0]
            true_len = torch.norm(node_pos[bond_index[0]] - node_pos[bond_index[1]], dim=-1)
            pred_len = torch.norm(preds['pred_pos'][bond_index[0]] - preds['pred_pos'][bond_index[1]], dim=-1)
            out['loss_len'] = F.mse_loss(pred_len, true_len)
        total = loss_pos + out['loss_node'] + out['loss_edge'] + out.get('loss_len', 0)
        return {'loss': total, 'loss_pos': loss_pos, **out}

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

    def forward(self, h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, batch_edge, t, _graph=None):
        """Predict the clean molecule from the perturbed one at per-graph step `t`."""
        _lib._need_gpu(h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, t)
        eng = self._engine()
        g = _graph if _graph is not None else _lib.graph_for(edge_index, batch_node, int(t.numel()))
        pn, pp, ph = self._forward_raw(eng, g, _lib.f32c(h_node_pert), _lib.f32c(pos_pert), _lib.f32c(h_edge_pert), None,
                                       _lib.i64c(t))
        return {'pred_node': pn, 'pred_pos': pp, 'pred_halfedge': ph}

    def sampler(self, n_graphs, batch_node, halfedge_index, batch_halfedge, *, seed=None, mol_ids=None, noise=None,
                return_traj=True, bond_predictor=None, guidance=None, overlap_guidance=False):
        """Stateful driver of the reverse chain (``init()`` then ``step(i)`` for i = 0..T-1); ``sample`` wraps it.
        overlap_guidance=True runs the guidance chain on a side stream concurrently with the denoiser forward of the same step
        (same results).  It paid in round 1 (0.7 ms per step, the kernels left tails for each other); with the round-2 kernels
        filling every CU by themselves it costs 0.5 ms (27.6 vs 28.1 ms per step), so in line is the default."""
        if self.categorical_space == 'continuous':
            if guidance is not None and guidance[1] > 0:
                raise NotImplementedError('guidance in the continuous categorical space: the reference objectives that read the sampled '
                                          'bond classes do not exist there (models/model.py:340-359); not built')
            return _ContinuousSampler(self, n_graphs, batch_node, halfedge_index, batch_halfedge, seed, mol_ids, noise, return_traj)
        return _Sampler(self, n_graphs, batch_node, halfedge_index, batch_halfedge, seed, mol_ids, noise, return_traj,
                        bond_predictor, guidance, overlap_guidance)

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
        sm = self.sampler(n_graphs, batch_node, halfedge_index, batch_halfedge, seed=seed, mol_ids=mol_ids, noise=noise,
                          return_traj=return_traj, bond_predictor=bond_predictor, guidance=guidance)
        sm.init()
        for i in range(self.num_timesteps):
            sm.step(i)
        return sm.result()


class _Sampler:
    """One packed batch moving through models/model.py:244-378.  All buffers are allocated once in __init__; a step is ONE
    library call (``mdx_sample_step_full``: time tensor, noise draw, denoiser forward, 3 posteriors, 2 Gumbel-max draws and,
    with the default 'uncertainty' objective, the bond-predictor guidance on a concurrent stream).

    State layout: the one-hot inputs of the denoiser ping-pong between two frames; the trajectory the reference returns is
    kept compact -- class ids as one byte per atom / half-edge and frame, positions fp32 -- and expanded lazily at the API
    edge (``traj.LazyOneHot``): 0.16 GB instead of 2.1 GB at 256 molecules."""

    def __init__(self, model, n_graphs, batch_node, halfedge_index, batch_halfedge, seed, mol_ids, noise, return_traj,
                 bond_predictor, guidance, overlap_guidance=False):
        _lib._need_gpu(batch_node, halfedge_index, batch_halfedge)
        self.guidance = None
        if guidance is not None:
            gui_type, gui_scale = guidance
            if gui_scale > 0:
                if bond_predictor is None:
                    raise ValueError('guidance needs a bond_predictor')
                if gui_type not in GUIDANCE_TYPES:
                    raise NotImplementedError(f'Guidance type {gui_type} is not implemented')
                self.guidance = (gui_type, float(gui_scale))
                self.bp = bond_predictor
                self.bp_eng = bond_predictor._engine()
        self.m = m = model
        self.dev = dev = batch_node.device
        self.T, self.Kn, self.Ke = m.num_timesteps, m.num_node_types, m.num_edge_types
        self.N, self.Eh = int(batch_node.numel()), int(batch_halfedge.numel())
        self.n_graphs = n_graphs
        self.eng = m._engine()
        # The matrix path is state of the (shared) engine handle.  A sampler resolves it ONCE, here -- module attribute, else the
        # process default in force at construction -- and re-applies it before every launch, so a sampler built inside
        # `with default_matrix_path(...)` keeps its path after the context exits and another sampler / forward / get_loss on the
        # same model cannot flip it under a live chain.
        self._path = self.eng._path
        self._bp_path = self.bp_eng._path if self.guidance is not None else None
        edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
        self.g = _lib.Graph(edge_index, batch_node, n_graphs, mol_ids)
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        self.noise = noise
        self.return_traj = return_traj
        f32 = dict(dtype=torch.float32, device=dev)
        N, Eh, Kn, Ke, T = self.N, self.Eh, self.Kn, self.Ke, self.T
        self.eps, self.u_n, self.u_h = torch.empty(N, 3, **f32), torch.empty(N, Kn, **f32), torch.empty(Eh, Ke, **f32)
        nT = T + 1 if return_traj else 2
        self.h_node = torch.zeros(2, N, Kn, **f32)        # one-hot state, frames ping-pong
        self.h_half = torch.zeros(2, Eh, Ke, **f32)
        self.pos_traj = torch.zeros(nT, N, 3, **f32)
        self.node_ids = torch.zeros(nT, N, dtype=torch.uint8, device=dev)     # compact trajectory
        self.half_ids = torch.zeros(nT, Eh, dtype=torch.uint8, device=dev)
        self.t = torch.empty(max(n_graphs, 1), dtype=torch.int64, device=dev)
        self.bn, self.bh = _lib.i64c(batch_node), _lib.i64c(batch_halfedge)
        self.preds = (torch.empty(N, Kn, **f32), torch.empty(N, 3, **f32), torch.empty(Eh, Ke, **f32))
        self.log_node = [torch.empty(N, Kn, **f32), torch.empty(N, Kn, **f32)]
        self.log_half = [torch.empty(Eh, Ke, **f32), torch.empty(Eh, Ke, **f32)]
        self.gd = None
        if self.guidance is not None:
            Kb = self.bp.num_edge_types
            self.bp_logits, self.bp_glogits = torch.empty(Eh, Kb, **f32), torch.empty(Eh, Kb, **f32)
            self.delta = torch.empty(N, 3, **f32)
            self.edge_index = edge_index
            self.batch_edge = torch.cat([self.bh, self.bh], dim=0)
            if self.guidance[0] == 'uncertainty':
                # the default objective is part of the library call, with its own workspace; on request (overlap_guidance) it runs
                # on a side stream concurrently with the denoiser's forward of the same step (both only read the step's input state)
                self.side = torch.cuda.Stream(device=dev, priority=int(os.environ.get('MDX_SIDE_PRIORITY', '0')))
                nbytes = _lib.lib().mdx_workspace_bytes(self.N, 2 * self.Eh)
                self._ws2 = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
                off = (-self._ws2.data_ptr()) % 256
                # the sampler OWNS its tape: Graph.tape() hands out a per-graph buffer that is re-allocated when another predictor
                # (other device / block count) asks for it, and the library keeps using the pointer captured here on every step
                nb_t = _lib.lib().mdx_bondpred_tape_bytes(self.N, 2 * self.Eh, self.bp.encoder.num_blocks)
                self._tape = torch.empty(nb_t + 256, dtype=torch.uint8, device=dev)
                toff = (-self._tape.data_ptr()) % 256
                tptr, tbytes = ctypes.c_void_p(self._tape.data_ptr() + toff), ctypes.c_size_t(self._tape.numel() - toff)
                self.gd = _lib.MdxGuidance(self.bp_eng.h, self.guidance[1], tptr, tbytes.value,
                                           ctypes.c_void_p(self._ws2.data_ptr() + off), self._ws2.numel() - off,
                                           _lib.ptr(self.bp_logits), _lib.ptr(self.bp_glogits), _lib.ptr(self.delta),
                                           ctypes.c_void_p(self.side.cuda_stream if overlap_guidance else 0))
        pt, ntr, etr = m.pos_transition, m.node_transition, m.edge_transition
        self.tables = _lib.MdxTables(*(_lib.ptr(x) for x in (pt.coef_x0, pt.coef_xt, pt.std, ntr.q_mats, ntr.transpopse_q_onestep_mats,
                                                             etr.q_mats, etr.transpopse_q_onestep_mats)))
        self.cur, self.lcur, self.pcur = 0, 0, 0  # one-hot frame / log-prob frame / position (and id) frame of the current state

    def _pframe(self, j):
        return j if self.return_traj else j % 2

    @torch.no_grad()
    def init(self):
        """Prior draw (models/model.py:244-263): classes ~ init_prob by Gumbel-max, positions ~ N(0, I)."""
        m, L, dev = self.m, _lib.lib(), self.dev
        u_n, u_h = self.u_n, self.u_h
        if self.noise is not None:
            e, a, b = self.noise(0)
            self.eps.copy_(e)
            u_n = a.to(dev).contiguous() if a.dtype == torch.float64 else self.u_n.copy_(a)
            u_h = b.to(dev).contiguous() if b.dtype == torch.float64 else self.u_h.copy_(b)
        else:
            _lib.check(L.mdx_noise(self.g.h, ctypes.c_uint64(self.seed), 0, self.Kn, self.Ke, _lib.ptr(self.eps),
                                   _lib.ptr(self.u_n), _lib.ptr(self.u_h), _lib.stream()))
        # the prior draw is the one place where the reference computes in float64 (models/transition.py:331-339: its logits are a
        # float64 numpy array moved to the device): mdx_prior_draw evaluates the Gumbel-max there in float64 too.  Explicit noise
        # may be float64 (the reference's rand_like dtype at this point) or float32.
        for tr, n, u, oh, ids, logs in ((m.node_transition, self.N, u_n, self.h_node, self.node_ids, self.log_node),
                                        (m.edge_transition, self.Eh, u_h, self.h_half, self.half_ids, self.log_half)):
            _lib.prior_draw(tr.init_prob, u, n, onehot=oh[0], log_onehot=logs[0], cls8=ids[0])
        self.pos_traj[0].copy_(self.eps)
        self.cur, self.lcur, self.pcur = 0, 0, 0

    @torch.no_grad()
    def step(self, i):
        """Loop iteration i (diffusion step T-1-i), models/model.py:272-372: one library call."""
        L, T = _lib.lib(), self.T
        draw = i + 1
        if self.noise is not None:
            e, a, b = self.noise(draw)
            self.eps.copy_(e); self.u_n.copy_(a); self.u_h.copy_(b)
            draw = -1
        c, n = self.cur, 1 - self.cur
        lc, ln = self.lcur, 1 - self.lcur
        pc, pn = self.pcur, self._pframe(i + 1)
        P = _lib.ptr
        cur = _lib.MdxState(P(self.h_node[c]), P(self.pos_traj[pc]), P(self.h_half[c]), P(self.log_node[lc]), P(self.log_half[lc]))
        nxt = _lib.MdxState(P(self.h_node[n]), P(self.pos_traj[pn]), P(self.h_half[n]), P(self.log_node[ln]), P(self.log_half[ln]))
        nz = _lib.MdxStepNoise(self.seed, draw, P(self.eps), P(self.u_n), P(self.u_h))
        ws, nb = self.g.workspace(self.dev)
        self.eng.use_matrix_path(self._path)
        if self._bp_path is not None:
            self.bp_eng.use_matrix_path(self._bp_path)
        _lib.check(L.mdx_sample_step_full(self.eng.h, self.g.h, ctypes.byref(self.tables), T - 1 - i, P(self.bn), P(self.bh),
                                          ctypes.byref(cur), ctypes.byref(nxt), P(self.preds[0]), P(self.preds[1]), P(self.preds[2]),
                                          ctypes.byref(nz), P(self.t), P(self.node_ids[pn]), P(self.half_ids[pn]),
                                          ctypes.byref(self.gd) if self.gd is not None else None, ws, nb, _lib.stream()))
        if self.guidance is not None and self.gd is None:  # the seven objectives that are torch expressions on the logits
            self._guide(self.h_node[c], self.pos_traj[pc], self.pos_traj[pn], self.h_half[n], self.log_half[ln])
        self.cur, self.lcur, self.pcur = n, ln, pn

    def _guide(self, h_node, pos, pos_prev, h_half_prev, log_half):
        """models/model.py:309-362 for the non-default objectives: pos_prev += delta, delta = -+scale * d f(bond logits) / d pos
        evaluated at the step's INPUT state; the reference's torch expression on the (Eh,5) logits, differentiated through
        the HIP backward of the predictor.  (The default 'uncertainty' objective lives inside mdx_sample_step_full.)"""
        g = self.g
        gui_type, scale = self.guidance
        with torch.enable_grad(), _lib.pinned_matrix_path(self.bp, self._bp_path):
            pos_in = pos.detach().clone().requires_grad_(True)
            pred = self.bp(h_node.detach(), pos_in, self.bn, self.edge_index, self.batch_edge, self.t[:self.n_graphs], _graph=g)
            sign = -1.0
            if gui_type == 'entropy':
                p = torch.softmax(pred, dim=-1)
                obj = (-torch.sum(p * torch.log(p + 1e-12), dim=-1)).log().sum()
            elif gui_type == 'uncertainty':
                obj = torch.sigmoid(-torch.logsumexp(pred, dim=-1)).log().sum()
            elif gui_type == 'uncertainty_bond':
                p = torch.softmax(pred, dim=-1)
                u = torch.sigmoid(-torch.logsumexp(pred, dim=-1)).log()
                obj = (u * p[:, 1:].detach().sum(dim=-1)).sum()
            elif gui_type == 'entropy_bond':
                p = torch.softmax(pred, dim=-1)
                ent = (-torch.sum(p * torch.log(p + 1e-12), dim=-1)).log()
                obj = (ent * p[:, 1:].detach().sum(dim=-1)).sum()
            elif gui_type in ('logit_bond', 'logit'):
                cls = h_half_prev.argmax(dim=-1)
                keep = ((cls >= 1) & (cls <= 4)) if gui_type == 'logit_bond' else (cls <= 4)
                idx = keep.nonzero().squeeze(-1)
                obj = pred[idx, cls[idx]].sum()
                sign = 1.0
            elif gui_type == 'crossent':
                obj = F.cross_entropy(pred, log_half.exp()[:, :-1], reduction='none').log().sum()
            else:  # 'crossent_bond'
                obj = F.cross_entropy(pred[:, 1:], log_half.exp()[:, 1:-1], reduction='none').log().sum()
            delta = sign * torch.autograd.grad(obj, pos_in)[0] * scale
        pos_prev.add_(delta)

    def state(self):
        return {'h_node': self.h_node[self.cur], 'pos': self.pos_traj[self.pcur], 'h_halfedge': self.h_half[self.cur],
                'log_node': self.log_node[self.lcur], 'log_halfedge': self.log_half[self.lcur]}

    def set_state(self, h_node, pos, h_halfedge, log_node, log_halfedge, frame=0):
        """Teacher-forcing hook for the parity tests: the state becomes trajectory frame `frame`."""
        pf = self._pframe(frame)
        self.h_node[0].copy_(h_node); self.pos_traj[pf].copy_(pos); self.h_half[0].copy_(h_halfedge)
        self.node_ids[pf].copy_(h_node.argmax(-1)); self.half_ids[pf].copy_(h_halfedge.argmax(-1))
        self.log_node[0].copy_(log_node); self.log_half[0].copy_(log_halfedge)
        self.cur, self.lcur, self.pcur = 0, 0, pf

    def result(self):
        from .traj import LazyOneHot
        if self.return_traj:
            node_ids, pos, half_ids = self.node_ids, self.pos_traj, self.half_ids
        else:
            p = self.pcur
            node_ids, pos, half_ids = self.node_ids[p:p + 1], self.pos_traj[p:p + 1], self.half_ids[p:p + 1]
        traj = [LazyOneHot(node_ids, self.Kn), pos, LazyOneHot(half_ids, self.Ke)]
        return {'pred': [self.preds[0], self.preds[1], self.preds[2]], 'traj': traj}


class _ContinuousSampler:
    """The reverse chain for categorical_space == 'continuous' (models/model.py:244-308 with the else-branches :249-251, :301-304):
    atom and bond features are real vectors that start from N(0, I) and follow the same Gaussian posterior as the positions, each
    with its own schedule.  Same interface as ``_Sampler``.  A step is the fused denoiser forward plus three posterior launches
    (``mdx_gauss_posterior``); the noise comes from the library's per-molecule Philox streams like in the discrete chain -- the
    normals of the class features by the inverse normal CDF of its uniforms -- so results do not depend on sharding.
    No shipped config uses this space: it is built for parity, not tuned."""

    def __init__(self, model, n_graphs, batch_node, halfedge_index, batch_halfedge, seed, mol_ids, noise, return_traj):
        _lib._need_gpu(batch_node, halfedge_index, batch_halfedge)
        self.m = m = model
        self.dev = dev = batch_node.device
        self.T, self.Kn, self.Ke = m.num_timesteps, m.num_node_types, m.num_edge_types
        self.N, self.Eh = int(batch_node.numel()), int(batch_halfedge.numel())
        self.n_graphs = n_graphs
        self.eng = m._engine()
        self.edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
        self.batch_edge = torch.cat([batch_halfedge, batch_halfedge], dim=0)
        self.g = _lib.Graph(self.edge_index, batch_node, n_graphs, mol_ids)
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        self.noise = noise
        self.return_traj = return_traj
        f32 = dict(dtype=torch.float32, device=dev)
        nT = self.T + 1 if return_traj else 2
        self.node_traj = torch.zeros(nT, self.N, self.Kn, **f32)
        self.pos_traj = torch.zeros(nT, self.N, 3, **f32)
        self.half_traj = torch.zeros(nT, self.Eh, self.Ke, **f32)
        self.bn, self.bh = _lib.i64c(batch_node), _lib.i64c(batch_halfedge)
        self.t = torch.empty(max(n_graphs, 1), dtype=torch.int64, device=dev)
        self.eps, self.u_n, self.u_h = torch.empty(self.N, 3, **f32), torch.empty(self.N, self.Kn, **f32), torch.empty(self.Eh, self.Ke, **f32)
        self.preds = None
        self.cur = 0
        self._path = self.eng._path   # resolved once per sampler, like _Sampler

    def _frame(self, j):
        return j if self.return_traj else j % 2

    def _draw(self, draw):
        """(eps_pos, eps_node, eps_halfedge) ~ N(0, 1) of draw index `draw` (0 = prior, i + 1 = iteration i)."""
        if self.noise is not None:
            return tuple(x.to(self.dev, torch.float32) for x in self.noise(draw))
        _lib.check(_lib.lib().mdx_noise(self.g.h, ctypes.c_uint64(self.seed), draw, self.Kn, self.Ke, _lib.ptr(self.eps),
                                        _lib.ptr(self.u_n), _lib.ptr(self.u_h), _lib.stream()))
        lo, hi = 2.0 ** -24, 1.0 - 2.0 ** -24
        z = lambda u: torch.special.ndtri(u.clamp(lo, hi).double()).float()
        return self.eps.clone(), z(self.u_n), z(self.u_h)

    @torch.no_grad()
    def init(self):
        e, a, b = self._draw(0)
        self.node_traj[0].copy_(a); self.pos_traj[0].copy_(e); self.half_traj[0].copy_(b)
        self.cur = 0

    @torch.no_grad()
    def step(self, i):
        m, c, n = self.m, self.cur, self._frame(i + 1)
        self.t.fill_(self.T - 1 - i)
        t = self.t[:self.n_graphs]
        e, a, b = self._draw(i + 1)
        h_node, pos, h_half = self.node_traj[c], self.pos_traj[c], self.half_traj[c]
        with _lib.pinned_matrix_path(m, self._path):
            preds = m.forward(h_node, pos, self.bn, torch.cat([h_half, h_half], dim=0), self.edge_index, self.batch_edge, t, _graph=self.g)
        self.pos_traj[n].copy_(m.pos_transition.get_prev_from_recon(pos, preds['pred_pos'], t, self.bn, eps=e))
        self.node_traj[n].copy_(m.node_transition.get_prev_from_recon(h_node, preds['pred_node'], t, self.bn, eps=a))
        self.half_traj[n].copy_(m.edge_transition.get_prev_from_recon(h_half, preds['pred_halfedge'], t, self.bh, eps=b))
        self.preds = (preds['pred_node'], preds['pred_pos'], preds['pred_halfedge'])
        self.cur = n

    def state(self):
        return {'h_node': self.node_traj[self.cur], 'pos': self.pos_traj[self.cur], 'h_halfedge': self.half_traj[self.cur]}

    def set_state(self, h_node, pos, h_halfedge, frame=0):
        f = self._frame(frame)
        self.node_traj[f].copy_(h_node); self.pos_traj[f].copy_(pos); self.half_traj[f].copy_(h_halfedge)
        self.cur = f

    def result(self):
        if self.return_traj:
            traj = [self.node_traj, self.pos_traj, self.half_traj]
        else:
            p = self.cur
            traj = [self.node_traj[p:p + 1], self.pos_traj[p:p + 1], self.half_traj[p:p + 1]]
        return {'pred': list(self.preds), 'traj': traj}
