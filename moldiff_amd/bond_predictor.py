"""BondPredictor: noisy (atom types, positions) -> bond-type logits; differentiated w.r.t. positions for
guidance.  Constructor, state_dict keys and forward signature follow the reference's
models/bond_predictor.py (:12-37, :128-162).  The HIP forward/backward
is the next row to be built; until then forward raises instead of silently falling back.
"""
import torch
import torch.nn as nn
from torch.nn import Module

from . import _lib
from .common import MLP, GaussianSmearing
from .diffusion import get_beta_schedule
from .graph import NodeEdgeNet
from .transition import ContigousTransition, GeneralCategoricalTransition


class BondPredictor(Module):
    def __init__(self, config, num_node_types, num_edge_types, **kwargs):
        super().__init__()
        self.config = config
        self.num_node_types = num_node_types
        self.num_edge_types = num_edge_types
        self.define_betas_alphas(config.diff)
        node_dim, edge_dim = config.node_dim, config.edge_dim
        time_dim = config.diff.time_dim if self.num_timesteps > 0 else 0
        self.node_embedder = nn.Linear(num_node_types, node_dim - time_dim, bias=False)
        self.edge_embedder = nn.Linear(num_node_types * 2, edge_dim - time_dim, bias=False)
        if self.num_timesteps != 0:
            self.time_emb = GaussianSmearing(stop=self.num_timesteps, num_gaussians=time_dim, type_='linear')
        self.encoder = NodeEdgeNet(node_dim, edge_dim, **config.encoder)
        self.edge_decoder = MLP(edge_dim + node_dim, num_edge_types, edge_dim, num_layer=3)
        self.edge_weight = torch.tensor([0.1] + [1.] * (self.num_edge_types - 1), dtype=torch.float32)
        self.ce_loss = torch.nn.CrossEntropyLoss(self.edge_weight)

    def define_betas_alphas(self, config):
        self.num_timesteps = T = config.num_timesteps
        if T == 0:
            return
        self.categorical_space = getattr(config, 'categorical_space', 'discrete')
        if self.categorical_space != 'discrete':
            raise NotImplementedError("categorical_space='continuous' is not built")
        self.scaling = [1., 1., 1.]
        self.pos_transition = ContigousTransition(get_beta_schedule(num_timesteps=T, **config.diff_pos))
        self.node_transition = GeneralCategoricalTransition(
            get_beta_schedule(num_timesteps=T, **config.diff_atom), self.num_node_types,
            init_prob=config.diff_atom.init_prob)

    def get_loss(self, *args, **kwargs):
        raise NotImplementedError('training loss is outside the sampling hot path (SURVEY.md section 8(f))')

    def forward(self, h_node, pos_node, batch_node, edge_index, batch_edge, t):
        _lib._need_gpu(h_node, pos_node, batch_node, edge_index, t)
        raise NotImplementedError('BondPredictor HIP forward/backward is the next scope row (SURVEY.md 8 a14/a15)')
