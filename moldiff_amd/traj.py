"""Compact sampling trajectory.

The reference keeps every frame of the reverse chain as one-hot fp32 rows -- ``node_traj (T+1,N,8)`` and
``halfedge_traj (T+1,Eh,6)`` (models/model.py:256-263,365-367,377) -- 2.1 GB at 256 molecules, of which its caller decodes
about 2 % (scripts/sample_drug3d.py:155-168).  The sampling kernels write one BYTE per atom / half-edge and frame instead
(``mdx_sample_step_full``'s ``node_cls`` / ``halfedge_cls``: 0.08 GB), and ``MolDiff.sample`` hands the caller a
``LazyOneHot``: a tensor subclass with the reference's shape / dtype that expands to one-hot rows only where it is looked at.

    traj = out['traj'][0]            # LazyOneHot, traj.shape == (T+1, N, 8), traj.dtype == float32
    traj[t], traj[:, mask]           # still compact (indexing the frame / row dimensions acts on the class ids)
    traj.cpu()                       # still compact, on the host
    traj.numpy(), traj.dense()       # the reference's array / tensor: materialised here, at the API edge
    traj.ids                         # (T+1, N) uint8 class ids
    any other torch op               # runs on the dense expansion (correct, costs the memory the reference always pays)
"""
import numpy as np
import torch
from torch.utils._pytree import tree_map

_aten = torch.ops.aten


class LazyOneHot(torch.Tensor):
    @staticmethod
    def __new__(cls, ids, num_classes):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(ids.shape) + (int(num_classes),), dtype=torch.float32,
                                                device=ids.device, requires_grad=False)
        r.ids = ids
        r.K = int(num_classes)
        return r

    def __repr__(self):
        return f'LazyOneHot(shape={tuple(self.shape)}, device={self.ids.device}, ids=uint8{tuple(self.ids.shape)})'

    def dense(self):
        """The reference's tensor: one-hot fp32 rows."""
        out = torch.zeros(tuple(self.ids.shape) + (self.K,), dtype=torch.float32, device=self.ids.device)
        out.scatter_(-1, self.ids.long().unsqueeze(-1), 1.0)
        return out

    def numpy(self):
        ids = self.ids.detach().cpu().numpy()
        return np.eye(self.K, dtype=np.float32)[ids]

    def argmax(self, dim=-1, keepdim=False):  # the only reduction the path's consumers take of a one-hot row
        if dim in (-1, self.dim() - 1):
            r = self.ids.long()
            return r.unsqueeze(-1) if keepdim else r
        return self.dense().argmax(dim, keepdim)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        self = next(a for a in args if isinstance(a, LazyOneHot))
        nd = self.ids.dim()

        def rewrap(ids):
            return LazyOneHot(ids, self.K)

        # indexing / moving that leaves the class dimension alone acts on the ids
        if func in (_aten.select.int, _aten.slice.Tensor) and args[0] is self:
            dim = args[1] % (nd + 1)
            if dim < nd:
                return rewrap(func(self.ids, dim, *args[2:], **kwargs))
        if func is _aten.index.Tensor and args[0] is self:
            idx = list(args[1])
            # dimensions the indices consume: a bool mask covers as many dimensions as it has, everything else (incl. None) one.
            # Only when they all fall on the frame / row dimensions is this an operation on the ids.
            used = sum(i.dim() if (torch.is_tensor(i) and i.dtype in (torch.bool, torch.uint8)) else 1 for i in idx)
            if used <= nd:
                return rewrap(func(self.ids, idx))
        if func in (_aten.clone.default, _aten.detach.default, _aten.alias.default):
            return rewrap(func(self.ids, **kwargs))
        if func is _aten._to_copy.default and kwargs.get('dtype') in (None, torch.float32):
            # device / layout moves keep the ids; a dtype change (.double(), .half(), .long()) is an operation on the VALUES and
            # takes the dense path below so that the result really has the requested dtype
            kw = {k: v for k, v in kwargs.items() if k != 'dtype'}
            return rewrap(func(self.ids, **kw))
        dense = lambda x: x.dense() if isinstance(x, LazyOneHot) else x
        return func(*tree_map(dense, args), **tree_map(dense, kwargs))
