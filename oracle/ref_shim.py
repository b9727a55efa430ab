"""Import the *real* reference (read-only at /root/reference) on CPU.  BUILD-CONTAINER ONLY.

Used by ``oracle/make_goldens.py`` to pin the oracle and to generate ``tests/golden/*.npz``.
The reference needs three third-party modules this image lacks; tiny stand-ins for THOSE
(not for reference code) are registered before import:
  * torch_scatter.scatter_sum  -> zero-initialised index_add (its documented semantics)
  * torch_geometric(.nn/.nn.pool) -> only imported for dead code (radius_graph, knn_graph, knn)
  * easydict.EasyDict -> attribute-style dict
Nothing from this file is used at test time on the GPU box.
"""
import sys
import types

import torch

REF = '/root/reference'


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return EasyDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(EasyDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, EasyDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _install():
    if 'torch_scatter' in sys.modules:
        return
    ts = types.ModuleType('torch_scatter')

    def scatter_sum(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0
        n = int(dim_size) if dim_size is not None else int(index.max()) + 1
        res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        return res.index_add_(0, index, src)

    def _na(*a, **k):
        raise NotImplementedError('dead-code stub')

    ts.scatter_sum = scatter_sum
    ts.scatter_add = scatter_sum
    for n in ('scatter_softmax', 'scatter_mean', 'scatter_max'):
        setattr(ts, n, _na)
    sys.modules['torch_scatter'] = ts
    tg = types.ModuleType('torch_geometric')
    tgn = types.ModuleType('torch_geometric.nn')
    tgp = types.ModuleType('torch_geometric.nn.pool')
    for m in (tgn, tgp):
        for n in ('radius_graph', 'knn_graph', 'knn'):
            setattr(m, n, _na)
    tg.nn = tgn
    tgn.pool = tgp
    sys.modules.update({'torch_geometric': tg, 'torch_geometric.nn': tgn, 'torch_geometric.nn.pool': tgp})
    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    sys.modules['easydict'] = ed
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load():
    """Returns (MolDiff, BondPredictor, graph_module, transition_module, diffusion_module, common_module)."""
    _install()
    from models.model import MolDiff
    from models.bond_predictor import BondPredictor
    import models.graph as G
    import models.transition as TR
    import models.diffusion as DF
    import models.common as CM
    return MolDiff, BondPredictor, G, TR, DF, CM


def load_yaml_cfg(rel):
    import yaml
    with open(f'{REF}/{rel}') as f:
        return EasyDict(yaml.safe_load(f))
