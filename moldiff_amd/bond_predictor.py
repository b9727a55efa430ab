"""BondPredictor: noisy (atom types, positions) -> bond-type logits, differentiable w.r.t. positions.

Constructor, ``state_dict`` keys and ``forward`` signature follow the reference's
``models/bond_predictor.py`` (:12-37, :128-162).  ``forward`` runs the HIP encoder (8 NodeEdgeNet blocks,
``update_pos=False``) + 3-layer decoder; when ``pos_node.requires_grad`` it records a per-block tape and the
returned logits carry a ``grad_fn`` whose backward is the hand-written data-gradient pass of
``csrc/mdx_bondpred.hip`` -- so the reference's guidance code
``torch.autograd.grad(f(logits), pos_in)`` (models/model.py:312-325) works unchanged for every guidance type
that is a function of the logits.  Only d/d pos is provided (what guidance needs); there is no weight gradient.
"""
import ctypes

import torch
import torch.nn as nn
from torch.nn import Module

from . import _lib
from .common import MLP, GaussianSmearing
from .diffusion import get_beta_schedule
from .graph import NodeEdgeNet, _sig, synth_gates
from .transition import ContigousTransition, GeneralCategoricalTransition


class _BondLogits(torch.autograd.Function):
    """logits = BondPredictor(h_node, pos, t); backward returns dL/dpos only."""

    @staticmethod
    def forward(ctx, pos, h_node, t, eng, g, num_blocks, num_edge_types):
        dev = pos.device
        pos_c, h_c, t_c = _lib.f32c(pos), _lib.f32c(h_node), _lib.i64c(t)
        logits = torch.empty(g.Eh, num_edge_types, dtype=torch.float32, device=dev)
        ws, nb = g.workspace(dev)
        tape, tptr, tbytes = None, ctypes.c_void_p(0), ctypes.c_size_t(0)
        need_grad = pos.requires_grad
        if need_grad:
            tape, tptr, tbytes = g.tape(dev, num_blocks)
            # one tape per graph: a second forward before this one's backward overwrites it -> detected in backward
            g._tape_version = getattr(g, '_tape_version', 0) + 1
            ctx.tape_version = g._tape_version
        _lib.check(_lib.lib().mdx_bondpred_forward(eng.h, g.h, _lib.ptr(h_c), _lib.ptr(pos_c), _lib.ptr(t_c), _lib.ptr(logits),
                                                   ws, nb, tptr, tbytes, _lib.stream()))
        ctx.eng, ctx.g, ctx.num_blocks = eng, g, num_blocks
        ctx.save_for_backward(pos_c)
        return logits

    @staticmethod
    def backward(ctx, glogits):
        (pos_c,) = ctx.saved_tensors
        g, eng = ctx.g, ctx.eng
        dev = pos_c.device
        gpos = torch.empty_like(pos_c)
        ws, nb = g.workspace(dev)
        if getattr(ctx, 'tape_version', None) != getattr(g, '_tape_version', None):
            raise RuntimeError('BondPredictor backward: the per-graph tape was overwritten by a later forward on the same graph '
                               '(run forward -> backward pairs one at a time, or use separate graphs)')
        _, tptr, tbytes = g.tape(dev, ctx.num_blocks)
        gl = _lib.f32c(glogits)
        _lib.check(_lib.lib().mdx_bondpred_backward(eng.h, g.h, _lib.ptr(pos_c), _lib.ptr(gl), 1.0,
                                                    _lib.ptr(gpos), ws, nb, tptr, tbytes, _lib.stream()))
        return gpos, None, None, None, None, None, None


class BondPredictor(Module):
    def __getstate__(self):
        # the packed-weight engine is a device handle: never copied or pickled (deepcopy / torch.save of the module
        # rebuild it lazily from the state_dict on first use)
        d = self.__dict__.copy()
        d['_eng'], d['_eng_sig'] = None, None
        return d

    def __init__(self, config, num_node_types, num_edge_types, **kwargs):
        # variants the kernels are not built for are rejected BEFORE any sub-module exists (INTEGRATION.md "Constructor variants beyond the shipped configs")
        if config.encoder.get('update_pos', True) if hasattr(config.encoder, 'get') else getattr(config.encoder, 'update_pos', True):
            raise NotImplementedError('the bond predictor kernels assume encoder.update_pos=False (the shipped config)')
        super().__init__()
        self.config = config
        self.num_node_types = num_node_types
        self.num_edge_types = num_edge_types
        self.define_betas_alphas(config.diff)
        node_dim, edge_dim = config.node_dim, config.edge_dim
        # num_timesteps == 0: the time-free predictor (models/bond_predictor.py:27-31) -- full-width embedders, no time embedding,
        # clean inputs in get_loss and t = 0 for the encoder's time columns (:97-102, :141-144)
        time_dim = config.diff.time_dim if self.num_timesteps > 0 else 0
        self.time_dim = time_dim
        self.node_embedder = nn.Linear(num_node_types, node_dim - time_dim, bias=False)
        self.edge_embedder = nn.Linear(num_node_types * 2, edge_dim - time_dim, bias=False)
        if self.num_timesteps != 0:
            self.time_emb = GaussianSmearing(stop=self.num_timesteps, num_gaussians=time_dim, type_='linear')
        self.encoder = NodeEdgeNet(node_dim, edge_dim, **config.encoder)
        self.edge_decoder = MLP(edge_dim + node_dim, num_edge_types, edge_dim, num_layer=3)
        self.edge_weight = torch.tensor([0.1] + [1.] * (self.num_edge_types - 1), dtype=torch.float32)
        self.ce_loss = torch.nn.CrossEntropyLoss(self.edge_weight)
        self._eng = None
        self._eng_sig = None

    def define_betas_alphas(self, config):
        self.num_timesteps = T = config.num_timesteps
        if T == 0:
            return
        self.categorical_space = getattr(config, 'categorical_space', 'discrete')
        if self.categorical_space not in ('discrete', 'continuous'):
            raise ValueError(self.categorical_space)
        self.scaling = list(getattr(config, 'scaling', [1., 1., 1.])) if self.categorical_space == 'continuous' else [1., 1., 1.]
        assert self.scaling[0] == 1, 'scaling for pos should be 1'
        self.pos_transition = ContigousTransition(get_beta_schedule(num_timesteps=T, **config.diff_pos))
        node_betas = get_beta_schedule(num_timesteps=T, **config.diff_atom)
        if self.categorical_space == 'discrete':
            self.node_transition = GeneralCategoricalTransition(node_betas, self.num_node_types, init_prob=config.diff_atom.init_prob)
        else:  # models/bond_predictor.py:69-71: noisy real-valued atom features
            self.node_transition = ContigousTransition(node_betas, self.num_node_types, self.scaling[1])

    # None = follow _lib.default_matrix_path (exact fp32 unless MOLDIFF_MATRIX_PATH says otherwise); or 'exact_f32' / 'split_f16'
    matrix_path = None

    def _engine(self):
        sig = _sig(self)
        if self._eng is None or sig != self._eng_sig:
            e = self.encoder
            eng = _lib.Model(_lib.MDX_KIND_BONDPRED, num_blocks=e.num_blocks, cutoff=e.cutoff, update_pos=False,
                             time_dim=self.time_dim, num_timesteps=self.num_timesteps,
                             num_node_types=self.num_node_types, num_edge_types=self.num_edge_types,
                             node_dim=e.node_dim, edge_dim=e.edge_dim, num_gaussians=16, smear_start=e.distance_expansion.start)
            eng.upload({**self.state_dict(), **synth_gates(e, 'encoder.')})
            self._eng, self._eng_sig = eng, sig
        return self._eng.use_matrix_path(self.matrix_path)

    def sample_time(self, num_graphs, device, **kwargs):
        T = self.num_timesteps
        ts = torch.randint(0, T, size=(num_graphs // 2 + 1,), device=device)
        ts = torch.cat([ts, T - ts - 1], dim=0)[:num_graphs]
        return ts, torch.ones_like(ts).float() / T

    def get_loss(self, node_type, node_pos, batch_node, halfedge_type, halfedge_index, batch_halfedge, num_mol, *,
                 time_step=None, noise=None):
        """Class-weighted cross-entropy of the predicted bond types on a noised batch (models/bond_predictor.py:84-124).
        Under ``no_grad`` it runs on the fused kernels; with grad enabled on the differentiable layer operators
        (``train_graph.bondpred_forward``) and ``loss.backward()`` fills every parameter's ``.grad``.
        time_step (num_mol,) / noise = dict(eps_pos, u_node) may be injected; default: torch's generator."""
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        noise = noise or {}
        with torch.no_grad():
            if self.num_timesteps != 0:
                t = self.sample_time(num_mol, node_pos.device)[0] if time_step is None else time_step
                pos = self.pos_transition.add_noise(node_pos, t, batch_node, noise.get('eps_pos'))
                h_node = self.node_transition.add_noise(
                    node_type, t, batch_node, noise.get('u_node' if self.categorical_space == 'discrete' else 'eps_node'))[0]
            else:  # time-free: clean one-hot types and positions (models/bond_predictor.py:100-102)
                t = torch.zeros(num_mol, dtype=torch.long, device=node_pos.device)
                pos = node_pos
                h_node = torch.nn.functional.one_hot(node_type, self.num_node_types).float()
            edge_index = torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1)
            batch_edge = torch.cat([batch_halfedge, batch_halfedge], dim=0)
        with torch.enable_grad() if train else torch.no_grad():
            if train:
                from . import train_graph
                pred_halfedge = train_graph.bondpred_forward(self, h_node, pos, batch_node, edge_index, batch_edge, t, flipped_halves=True)
            else:
                pred_halfedge = self(h_node, pos, batch_node, edge_index, batch_edge, t,
                                     _graph=_lib.graph_for_halfedges(halfedge_index, batch_node, int(t.numel())))
            loss_edge = self.ce_loss(pred_halfedge, halfedge_type)
        return {'loss': loss_edge, 'loss_edge': loss_edge}

    def forward(self, h_node, pos_node, batch_node, edge_index, batch_edge, t, _graph=None):
        """Predict the bond type of every half-edge (first half of `edge_index`) -> (Eh, num_edge_types)."""
        if self.num_timesteps == 0:  # the reference ignores `t` here and feeds zeros to the encoder (:143)
            t = torch.zeros(int(batch_node.max()) + 1 if t is None else int(t.numel()), dtype=torch.long, device=pos_node.device)
        _lib._need_gpu(h_node, pos_node, batch_node, edge_index, t)
        eng = self._engine()
        g = _graph if _graph is not None else _lib.graph_for(edge_index, batch_node, int(t.numel()))
        return _BondLogits.apply(pos_node, h_node, t, eng, g, self.encoder.num_blocks, self.num_edge_types)
