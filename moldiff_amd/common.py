"""Parameter containers shared by the network modules.

``MLP`` and ``GaussianSmearing`` keep the constructor signatures and ``state_dict`` key grammar of the
reference (models/common.py:181-201, :216-237) so checkpoints load strictly.  Inside the networks their arithmetic is
fused into the HIP kernels (csrc/mdx_edge2.hip, mdx_node.hip); called on their own (``forward``) they run the
library's layer operators (csrc/mdx_train.hip: Linear / LayerNorm+ReLU / smearing kernels), differentiable like the reference's.
"""
import numpy as np
import torch
import torch.nn as nn


class AttrDict(dict):
    """Attribute-style nested dict for configs (stands in for EasyDict, which this image lacks)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


class MLP(nn.Module):
    """Linear -> [LayerNorm -> ReLU -> Linear]*  (models/common.py:181-201)."""

    def __init__(self, in_dim, out_dim, hidden_dim, num_layer=2, norm=True, act_fn='relu', act_last=False):
        super().__init__()
        if act_fn != 'relu' or not norm or act_last:
            raise NotImplementedError('the HIP kernels implement the LayerNorm+ReLU MLP the live path uses')
        mods = []
        for i in range(num_layer):
            fan_in = in_dim if i == 0 else hidden_dim
            fan_out = out_dim if i == num_layer - 1 else hidden_dim
            mods.append(nn.Linear(fan_in, fan_out))
            if i < num_layer - 1:
                mods += [nn.LayerNorm(hidden_dim), nn.ReLU()]
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        """models/common.py:200-201 on device rows (..., in_dim) -> (..., out_dim): Linear and LayerNorm+ReLU operator launches."""
        from . import train_graph
        lead = x.shape[:-1]
        y = train_graph.mlp(self, x.reshape(-1, x.shape[-1]))
        return y.reshape(*lead, y.shape[-1])


class GaussianSmearing(nn.Module):
    """Radial basis table: exp(coeff_k (clamp(x, start, stop) - offset_k)^2)."""

    def __init__(self, start=0.0, stop=10.0, num_gaussians=50, type_='exp'):
        super().__init__()
        self.start, self.stop = start, stop
        if type_ == 'exp':
            offset = torch.exp(torch.linspace(start=np.log(start + 1), end=np.log(stop + 1), steps=num_gaussians)) - 1
        elif type_ == 'linear':
            offset = torch.linspace(start=start, end=stop, steps=num_gaussians)
        else:
            raise NotImplementedError('type_ must be either exp or linear')
        gap = torch.diff(offset)
        gap = torch.cat([gap[:1], gap])
        self.register_buffer('coeff', -0.5 / gap ** 2)
        self.register_buffer('offset', offset)

    def forward(self, dist):
        """models/common.py:233-237: clamp to [start, stop], (n,) -> (n, num_gaussians); integer inputs (time steps) are
        promoted to float like the reference's subtraction does."""
        from . import train_graph
        d = dist.reshape(-1)
        if not d.is_floating_point():
            d = d.float()
        return train_graph.smear(self, d)
