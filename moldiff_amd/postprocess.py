"""Harness consumers of the sampling path's outputs.

* ``FeaturizeMol`` -- constructor + ``decode_output`` with the reference's signature (utils/transforms.py:13-31,
  :65-122).  The reference runs this per molecule on the host in numpy after a full D2H copy; the same host form
  is kept for drop-in use, and ``decode_batch`` is the accelerated path: one pair of HIP kernels
  (``mdx_decode_output``) arg-maxes, drops mask-type atoms, re-indexes and compacts atoms/bonds for the whole
  packed batch on the device, so only the compact arrays travel.
* ``seperate_outputs`` / ``seperate_outputs_no_traj`` -- utils/sample.py:4-55 (same spelling as the reference).
``FeaturizeMol.__call__`` (training-side featurisation of a processed record) lives in ``moldiff_amd/data.py``.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


class FeaturizeMol(object):
    def __init__(self, atomic_numbers, mol_bond_types, use_mask_node, use_mask_edge):
        self.atomic_numbers = torch.LongTensor(atomic_numbers)
        self.mol_bond_types = torch.LongTensor(mol_bond_types)
        self.num_element = self.atomic_numbers.size(0)
        self.num_bond_types = self.mol_bond_types.size(0)
        self.num_node_types = self.num_element + int(use_mask_node)
        self.num_edge_types = self.num_bond_types + 1 + int(use_mask_edge)  # +1: the non-bonded class
        self.use_mask_node, self.use_mask_edge = use_mask_node, use_mask_edge
        self.ele_to_nodetype = {ele: i for i, ele in enumerate(atomic_numbers)}
        self.nodetype_to_ele = {i: ele for i, ele in enumerate(atomic_numbers)}

    follow_batch = ['node_type', 'halfedge_type']

    def __call__(self, data):
        """Training-side featurisation of one processed record (utils/transforms.py:35-62): see ``moldiff_amd.data.featurize``."""
        from .data import featurize
        return featurize(data, self)

    def decode_output(self, pred_node, pred_pos, pred_halfedge, halfedge_index):
        """One molecule, numpy arrays in, dict out (element, atom_pos, bond_type, bond_index, atom_prob, bond_prob)."""
        pa = _softmax(pred_node)
        atom_type, atom_prob = np.argmax(pa, axis=-1), np.max(pa, axis=-1)
        keep = atom_type < self.num_element
        renum = -np.ones(len(keep), dtype=np.int64)
        renum[keep] = np.arange(keep.sum())
        out = {'element': np.array([self.nodetype_to_ele[i] for i in atom_type[keep]]),
               'atom_pos': pred_pos[keep], 'atom_prob': atom_prob[keep]}
        if self.num_edge_types == 1:
            return out
        ph = _softmax(pred_halfedge)
        edge_type, edge_prob = np.argmax(ph, axis=-1), np.max(ph, axis=-1)
        is_bond = (edge_type > 0) & (edge_type <= self.num_bond_types)
        bond_type, bond_prob, bond_index = edge_type[is_bond], edge_prob[is_bond], halfedge_index[:, is_bond]
        if not keep.all():
            bond_index = renum[bond_index]
            ok = ~(bond_index < 0).any(axis=0)
            bond_index, bond_type, bond_prob = bond_index[:, ok], bond_type[ok], bond_prob[ok]
        out.update(bond_type=np.concatenate([bond_type, bond_type]), bond_prob=np.concatenate([bond_prob, bond_prob]),
                   bond_index=np.concatenate([bond_index, bond_index[::-1]], axis=1))
        return out

    def decode_batch(self, pred, batch_node, halfedge_index, batch_halfedge, n_graphs, graph=None):
        """Whole packed batch on the device -> list of per-molecule dicts identical to
        [decode_output(*seperate_outputs(...)[i]) for i in range(n_graphs)].
        pred = [pred_node (N,Kn), pred_pos (N,3), pred_halfedge (Eh,Ke)] device tensors (model.sample()['pred'])."""
        pn, pp, ph = (_lib.f32c(t) for t in pred)
        _lib._need_gpu(pn, pp, ph, batch_node, halfedge_index)
        dev = pn.device
        if graph is None:
            graph = _lib.graph_for_halfedges(halfedge_index, batch_node, n_graphs)
        N, Eh, B = graph.N, graph.Eh, graph.B
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        atom_type, atom_prob, atom_pos = torch.empty(N, **i32), torch.empty(N, **f32), torch.empty(N, 3, **f32)
        bond_type, bond_prob, bond_index = torch.empty(Eh, **i32), torch.empty(Eh, **f32), torch.empty(2, max(Eh, 1), **i32)
        n_atoms, n_bonds = torch.empty(max(B, 1), **i32), torch.empty(max(B, 1), **i32)
        ws, nb = graph.workspace(dev)
        _lib.check(_lib.lib().mdx_decode_output(
            graph.h, _lib.ptr(pn), pn.shape[1], _lib.ptr(pp), _lib.ptr(ph), ph.shape[1], self.num_element,
            self.num_bond_types, _lib.ptr(atom_type), _lib.ptr(atom_prob), _lib.ptr(atom_pos), _lib.ptr(n_atoms),
            _lib.ptr(bond_type), _lib.ptr(bond_prob), _lib.ptr(bond_index), _lib.ptr(n_bonds), ws, nb, _lib.stream()))
        at, ap, apos = atom_type.cpu().numpy(), atom_prob.cpu().numpy(), atom_pos.cpu().numpy()
        bt, bp, bi = bond_type.cpu().numpy(), bond_prob.cpu().numpy(), bond_index.cpu().numpy()
        na, nbd = n_atoms.cpu().numpy(), n_bonds.cpu().numpy()
        node_ptr = np.concatenate([[0], np.cumsum(np.bincount(batch_node.cpu().numpy(), minlength=B))])
        he_ptr = np.concatenate([[0], np.cumsum(np.bincount(batch_halfedge.cpu().numpy(), minlength=B))])
        ele = np.asarray(self.atomic_numbers.numpy())
        out = []
        for m in range(B):
            a0, a1 = node_ptr[m], node_ptr[m] + na[m]
            b0, b1 = he_ptr[m], he_ptr[m] + nbd[m]
            idx = bi[:, b0:b1].astype(np.int64)
            out.append({'element': ele[at[a0:a1]], 'atom_pos': apos[a0:a1], 'atom_prob': ap[a0:a1],
                        'bond_type': np.concatenate([bt[b0:b1], bt[b0:b1]]).astype(np.int64),
                        'bond_prob': np.concatenate([bp[b0:b1], bp[b0:b1]]),
                        'bond_index': np.concatenate([idx, idx[::-1]], axis=1)})
        return out


def seperate_outputs(outputs, n_graphs, batch_node, halfedge_index, batch_halfedge):
    """Split packed numpy outputs {'pred': [...], 'traj': [...]} per molecule (host, numpy -- like the reference)."""
    pred, traj = outputs['pred'], outputs['traj']
    res = []
    for i in range(n_graphs):
        mn, mh = (batch_node == i), (batch_halfedge == i)
        assert mn.sum() * (mn.sum() - 1) == mh.sum() * 2
        he = halfedge_index[:, mh]
        first = mn.nonzero()[0].min()
        assert first == he.min()
        res.append({'pred': [pred[0][mn], pred[1][mn], pred[2][mh]],
                    'traj': [traj[0][:, mn], traj[1][:, mn], traj[2][:, mh]],
                    'halfedge_index': he - first})
    return res


def seperate_outputs_no_traj(outputs, n_graphs, batch_node, halfedge_index, batch_halfedge):
    res = []
    for i in range(n_graphs):
        mn, mh = (batch_node == i), (batch_halfedge == i)
        assert mn.sum() * (mn.sum() - 1) == mh.sum() * 2
        he = halfedge_index[:, mh]
        first = mn.nonzero()[0].min()
        assert first == he.min()
        res.append({'node': outputs[0][mn], 'pos': outputs[1][mn], 'halfedge': outputs[2][mh],
                    'halfedge_index': he - first})
    return res
