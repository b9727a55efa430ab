"""Data side of the training loop without torch_geometric: records -> featurised molecules -> collated batches.

The reference trains on ``Drug3DData`` objects (a PyG ``Data`` subclass, utils/data.py:8-33) that its ``DataLoader`` collates
with ``follow_batch=['node_type', 'halfedge_type']`` (utils/transforms.py:31, scripts/train_drug3d.py:59-69).  What reaches
``model.get_loss`` is seven fields; this module produces exactly those from the reference's *processed records* -- the dicts
its LMDB stores (utils/dataset.py / utils/parser.py: ``element (n)``, ``pos_all_confs (C,n,3)``, ``i_conf_list (C)``,
``bond_index (2,2b)``, ``bond_type (2b)``, ``num_atoms``, ``num_bonds``, ``smiles``) -- with three pieces:

* ``featurize(record, featurizer, rng)``   = ``FeaturizeMol.__call__`` (utils/transforms.py:35-62): node types from the element
  table, one random conformer moved to its centroid, the upper-triangle half-edge list with its bond types;
* ``collate(mols)``                        = PyG's ``Batch.from_data_list`` for these keys with ``Drug3DData.__inc__``
  (utils/data.py:25-33): tensors concatenated along dim 0, ``halfedge_index`` along dim 1 with every molecule's indices
  shifted by the number of nodes before it, ``node_type_batch`` / ``halfedge_type_batch`` from ``follow_batch``;
* ``RecordLoader``                         = the ``DataLoader`` + ``inf_iterator`` pair (utils/misc.py): shuffled, endless,
  per-rank seeded.  SDF parsing / RDKit / LMDB stay outside (CPU chemistry): export the records once with the reference's
  own ``Drug3DDataset`` (``torch.save([dataset[i] ... as dicts])``) and point ``dataset.records`` of the training config at it.
"""
import numpy as np
import torch


class MolBatch:
    """The collated batch: attribute names follow the reference's ``batch.<key>`` uses (scripts/train_drug3d.py:93-104)."""

    def __init__(self, node_type, node_pos, node_type_batch, halfedge_type, halfedge_index, halfedge_type_batch, num_graphs):
        self.node_type, self.node_pos, self.node_type_batch = node_type, node_pos, node_type_batch
        self.halfedge_type, self.halfedge_index, self.halfedge_type_batch = halfedge_type, halfedge_index, halfedge_type_batch
        self.num_graphs = num_graphs

    def to(self, device):
        return MolBatch(*(t.to(device) for t in (self.node_type, self.node_pos, self.node_type_batch, self.halfedge_type,
                                                 self.halfedge_index, self.halfedge_type_batch)), self.num_graphs)

    def loss_args(self):
        """Positional arguments of ``get_loss`` (models/model.py:128-131)."""
        return (self.node_type, self.node_pos, self.node_type_batch, self.halfedge_type, self.halfedge_index,
                self.halfedge_type_batch, self.num_graphs)


def featurize(record, featurizer, rng=None):
    """One processed record -> dict(node_type (n) int64, node_pos (n,3) f32, halfedge_index (2,h) int64, halfedge_type (h) int64,
    i_conf).  `rng`: numpy Generator for the conformer draw (the reference uses np.random.randint of the global stream)."""
    element = np.asarray(record['element']).reshape(-1)
    n = int(record.get('num_atoms', len(element)))
    table = featurizer.ele_to_nodetype
    if any(int(e) not in table for e in element):
        raise AssertionError('unknown element')
    node_type = torch.tensor([table[int(e)] for e in element], dtype=torch.int64)
    confs = torch.as_tensor(np.asarray(record['pos_all_confs']), dtype=torch.float32).reshape(-1, n, 3)
    idx = int(rng.integers(confs.shape[0])) if rng is not None else int(np.random.randint(confs.shape[0]))
    pos = confs[idx]
    pos = pos - pos.mean(dim=0)
    bond_index = torch.as_tensor(np.asarray(record['bond_index']), dtype=torch.int64).reshape(2, -1)
    bond_type = torch.as_tensor(np.asarray(record['bond_type']), dtype=torch.int64).reshape(-1)
    num_bonds = int(record.get('num_bonds', bond_index.shape[1] // 2))
    mat = torch.zeros(n, n, dtype=torch.int64)
    mat[bond_index[0, :2 * num_bonds], bond_index[1, :2 * num_bonds]] = bond_type[:2 * num_bonds]
    halfedge_index = torch.triu_indices(n, n, offset=1)
    halfedge_type = mat[halfedge_index[0], halfedge_index[1]]
    if int((halfedge_type > 0).sum()) != num_bonds:
        raise AssertionError('bond list is not symmetric')
    i_conf = record['i_conf_list'][idx] if 'i_conf_list' in record else idx
    return {'node_type': node_type, 'node_pos': pos, 'halfedge_index': halfedge_index, 'halfedge_type': halfedge_type,
            'i_conf': i_conf}


def collate(mols):
    """list of featurised molecules -> MolBatch (``__inc__`` of halfedge_index = number of nodes, follow_batch vectors)."""
    sizes = [int(m['node_type'].shape[0]) for m in mols]
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    node_type = torch.cat([m['node_type'] for m in mols]) if mols else torch.zeros(0, dtype=torch.int64)
    node_pos = torch.cat([m['node_pos'] for m in mols]) if mols else torch.zeros(0, 3)
    halfedge_type = torch.cat([m['halfedge_type'] for m in mols]) if mols else torch.zeros(0, dtype=torch.int64)
    halfedge_index = (torch.cat([m['halfedge_index'] + int(o) for m, o in zip(mols, offs)], dim=1) if mols
                      else torch.zeros(2, 0, dtype=torch.int64))
    node_batch = torch.cat([torch.full((s,), i, dtype=torch.int64) for i, s in enumerate(sizes)]) if mols else torch.zeros(0, dtype=torch.int64)
    half_batch = (torch.cat([torch.full((int(m['halfedge_type'].shape[0]),), i, dtype=torch.int64) for i, m in enumerate(mols)])
                  if mols else torch.zeros(0, dtype=torch.int64))
    return MolBatch(node_type, node_pos, node_batch, halfedge_type, halfedge_index, half_batch, len(mols))


class RecordLoader:
    """Endless shuffled batches over a list of processed records (DataLoader(shuffle=True) + inf_iterator, utils/misc.py)."""

    def __init__(self, records, featurizer, batch_size, seed=0, shuffle=True, device=None):
        if len(records) == 0:
            raise ValueError('no records')
        self.records, self.featurizer, self.batch_size, self.shuffle, self.device = records, featurizer, batch_size, shuffle, device
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self._order, self._pos = None, 0

    def _next_index(self):
        if self._order is None or self._pos >= len(self._order):
            self._order = self.rng.permutation(len(self.records)) if self.shuffle else np.arange(len(self.records))
            self._pos = 0
        i = int(self._order[self._pos])
        self._pos += 1
        return i

    def epoch_batches(self):
        """One pass in order, last batch short (the reference's validation loader)."""
        for lo in range(0, len(self.records), self.batch_size):
            b = collate([featurize(r, self.featurizer, self.rng) for r in self.records[lo:lo + self.batch_size]])
            yield b.to(self.device) if self.device is not None else b

    def __call__(self, it=None):
        n = min(self.batch_size, len(self.records))
        b = collate([featurize(self.records[self._next_index()], self.featurizer, self.rng) for _ in range(n)])
        return b.to(self.device) if self.device is not None else b


def synthetic_records(n_mols, seed, mean_atoms=24.92, std_atoms=5.52, n_confs=3):
    """Chemically meaningless but structurally valid processed records (a spanning tree of single bonds plus a few extra
    bonds of random order): exercises featurize/collate/training end to end where no GEOM-Drugs export is available."""
    g = np.random.Generator(np.random.PCG64(seed))
    elements = np.array([6, 7, 8, 9, 15, 16, 17])
    out = []
    for _ in range(n_mols):
        n = int(max(2, g.normal(mean_atoms, std_atoms)))
        pairs = {(int(g.integers(0, j)), j) for j in range(1, n)}
        for _ in range(n // 4):
            i, j = sorted(int(v) for v in g.choice(n, 2, replace=False))
            pairs.add((i, j))
        pairs = sorted(pairs)
        bt = g.integers(1, 5, len(pairs))
        bi = np.array([[p[0] for p in pairs] + [p[1] for p in pairs], [p[1] for p in pairs] + [p[0] for p in pairs]])
        out.append({'element': elements[g.integers(0, 7, n)], 'pos_all_confs': (g.standard_normal((n_confs, n, 3)) * 2).astype(np.float32),
                    'i_conf_list': list(range(n_confs)), 'bond_index': bi, 'bond_type': np.concatenate([bt, bt]), 'num_atoms': n,
                    'num_bonds': len(pairs), 'smiles': ''})
    return out
