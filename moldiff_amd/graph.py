"""E(3)-equivariant node/edge/position message-passing network, MI355X-native.

Class names, constructor arguments, parameter names (strict ``load_state_dict``) and ``forward``
signatures follow the reference's live classes in ``models/graph.py``:
    NodeBlock :10-55, BondFFN :122-141, EdgeBlock :251-295, NodeEdgeNet :298-374, PosUpdate :377-396.
The arithmetic does not live here: ``forward`` hands device pointers to ``libmoldiff_hip.so``
(fused MFMA edge/node kernels + deterministic CSR segment sums).  The graph is planned once per
``edge_index`` (edges stably sorted by (left, right); see csrc/mdx_api.hip) and cached.
"""
import ctypes
import weakref

import torch
import torch.nn as nn
from torch.nn import Linear, Module, ModuleList

from . import _lib
from .common import MLP, GaussianSmearing


def _sig(module):
    return tuple((p.data_ptr(), p._version) for p in module.state_dict(keep_vars=True).values())


NG_KERNEL = 16     # distance gaussians the fused kernels evaluate; a net with fewer is packed with dead ones (synth_gates)
GATE_OPEN = 32.0   # sigmoid(32) == 1.0f exactly (the division form and the exp2 / rcp form of the kernels alike)


def synth_gates(net, prefix=''):
    """Parameters the fused kernels need and a net built with `use_gate=False` or `update_edge=False` does not have, under the keys
    a full net would carry; merged into the state_dict handed to the engine ({} for a full net).

    use_gate=False (models/graph.py:21-22,123-124): the reference multiplies by no gate at all.  The kernels always evaluate one, so
    they get PASS-THROUGH gates: all weights zero, LayerNorm (1, 0), last bias GATE_OPEN -> sigmoid(gate) == 1.0f and message * 1.0f
    is the message, bit for bit; in the guidance backward sigmoid' == 0 exactly, so nothing flows through the synthetic path.

    update_edge=False (models/graph.py:317-320,352-361): every block re-derives the edge features from the distances alone
    (edge_embs is num_gaussians -> edge_dim and ignores the incoming features) and there is no EdgeBlock.  The kernels get edge_embs
    weights with a ZERO block in front of the distance columns (0 * h_edge adds nothing) and all-zero EdgeBlocks, whose update
    out_transform(relu(LayerNorm(0))) = 0 leaves the edge features as edge_embs made them.  Both cost the unused GEMMs; no shipped
    config pays it."""
    nd, ed = net.node_dim, net.edge_dim
    out = {}
    # num_gaussians < 16 (models/graph.py:309-312): the kernels always evaluate 16 -- the missing ones get offset 0 / coeff 0 (value
    # exp(0) = 1) and ZERO columns in every edge_embs weight, so they add exactly nothing
    ng = net.distance_expansion.offset.numel()
    if ng < NG_KERNEL:
        for key in ('offset', 'coeff'):
            v = getattr(net.distance_expansion, key).detach().cpu().float()
            out[f'{prefix}distance_expansion.{key}'] = torch.cat([v, torch.zeros(NG_KERNEL - ng)])

    def mlp(pre, din, dhid, dout, last_bias=0.0):
        out[pre + '.net.0.weight'] = torch.zeros(dhid, din)
        out[pre + '.net.0.bias'] = torch.zeros(dhid)
        out[pre + '.net.1.weight'] = torch.ones(dhid)
        out[pre + '.net.1.bias'] = torch.zeros(dhid)
        out[pre + '.net.3.weight'] = torch.zeros(dout, dhid)
        out[pre + '.net.3.bias'] = torch.full((dout,), last_bias)

    def lin(pre, dout, din, bias=True):
        out[pre + '.weight'] = torch.zeros(dout, din)
        if bias:
            out[pre + '.bias'] = torch.zeros(dout)

    for i in range(net.num_blocks):
        if not net.use_gate:
            mlp(f'{prefix}node_blocks_with_edge.{i}.gate', ed + nd + 1, nd, nd, GATE_OPEN)
            if net.update_pos:
                mlp(f'{prefix}pos_blocks.{i}.edge_lin.gate', 2 * ed + 1, 32, 1, GATE_OPEN)
        if net.update_edge:
            if not net.use_gate:
                for side in ('bond_ffn_left', 'bond_ffn_right'):
                    mlp(f'{prefix}edge_blocks.{i}.{side}.gate', ed + nd + 1, 32, ed, GATE_OPEN)
            if ng < NG_KERNEL:
                w = net.edge_embs[i].weight.detach().cpu().float()
                out[f'{prefix}edge_embs.{i}.weight'] = torch.cat([w, torch.zeros(ed, NG_KERNEL - ng)], dim=1)
        else:
            w = net.edge_embs[i].weight.detach().cpu().float()
            out[f'{prefix}edge_embs.{i}.weight'] = torch.cat([torch.zeros(ed, ed), w, torch.zeros(ed, NG_KERNEL - ng)], dim=1)
            eb = f'{prefix}edge_blocks.{i}'
            for side in ('bond_ffn_left', 'bond_ffn_right'):
                lin(f'{eb}.{side}.bond_linear', 2 * ed, ed, bias=False)
                lin(f'{eb}.{side}.node_linear', 2 * ed, nd, bias=False)
                mlp(f'{eb}.{side}.inter_module', 2 * ed, 2 * ed, ed)
                mlp(f'{eb}.{side}.gate', ed + nd + 1, 32, ed)
            lin(f'{eb}.node_ffn_left', ed, nd)
            lin(f'{eb}.node_ffn_right', ed, nd)
            lin(f'{eb}.self_ffn', ed, ed)
            out[f'{eb}.layer_norm.weight'], out[f'{eb}.layer_norm.bias'] = torch.ones(ed), torch.zeros(ed)
            lin(f'{eb}.out_transform', ed, ed)
    return out


class _Block(Module):
    """Common plumbing: a block reaches the HIP engine through the NodeEdgeNet that owns it."""
    _owner = None
    _index = -1

    def _net(self):
        net = self._owner() if self._owner is not None else None
        if net is None:
            raise RuntimeError(f'{type(self).__name__} must be a child of a NodeEdgeNet to run: its kernels are fused '
                               f'with the neighbouring blocks and share the network-level weight pack')
        return net


class NodeBlock(_Block):
    def __init__(self, node_dim, edge_dim, hidden_dim, use_gate):
        super().__init__()
        self.use_gate, self.node_dim = use_gate, node_dim
        self.node_net = MLP(node_dim, hidden_dim, hidden_dim)
        self.edge_net = MLP(edge_dim, hidden_dim, hidden_dim)
        self.msg_net = Linear(hidden_dim, hidden_dim)
        if use_gate:  # use_gate=False (models/graph.py:21-22,46-48): no gate module; the fused kernels get a pass-through one (synth_gates)
            self.gate = MLP(edge_dim + node_dim + 1, hidden_dim, hidden_dim)  # +1: time
        self.centroid_lin = Linear(node_dim, hidden_dim)
        self.layer_norm = nn.LayerNorm(hidden_dim)
        self.act = nn.ReLU()
        self.out_transform = Linear(hidden_dim, node_dim)

    def forward(self, x, edge_index, edge_attr, node_time):
        """x (N,H), edge_index (2,E), edge_attr (E,He), node_time (N,1) -> node update (N,H), no residual."""
        net = self._net()
        _lib._need_gpu(x, edge_attr, node_time, edge_index)
        eng = net._engine()
        g = net._graph(edge_index, x.shape[0])
        out = torch.empty(x.shape[0], self.node_dim, dtype=torch.float32, device=x.device)
        ws, nb = g.workspace(x.device)
        xc, ec, tc = _lib.f32c(x), _lib.f32c(edge_attr), _lib.f32c(node_time).view(-1)   # alive until the launch
        _lib.check(_lib.lib().mdx_node_block(eng.h, g.h, self._index, _lib.ptr(xc), _lib.ptr(ec), _lib.ptr(tc), _lib.ptr(out),
                                             ws, nb, _lib.stream()))
        return out


class BondFFN(_Block):
    def __init__(self, bond_dim, node_dim, inter_dim, use_gate, out_dim=None):
        super().__init__()
        out_dim = bond_dim if out_dim is None else out_dim
        self.use_gate = use_gate
        self.bond_linear = Linear(bond_dim, inter_dim, bias=False)
        self.node_linear = Linear(node_dim, inter_dim, bias=False)
        self.inter_module = MLP(inter_dim, out_dim, inter_dim)
        if use_gate:  # models/graph.py:123-124,138-140
            self.gate = MLP(bond_dim + node_dim + 1, out_dim, 32)  # +1: time

    _side = -1  # 0 / 1 = bond_ffn_left / bond_ffn_right of an EdgeBlock (set by the owning NodeEdgeNet)

    def forward(self, bond_feat_input, node_feat_input, time):
        """bond (E,He), node (E,H) -- one gathered node row per edge --, time (E,1) -> (E,He)  (models/graph.py:133-141)."""
        if self._side < 0:
            # not one of the EdgeBlock's fused FFNs (e.g. PosUpdate.edge_lin, out_dim = 1, models/graph.py:389): the layer operators
            from . import train_graph
            _lib._need_gpu(bond_feat_input, node_feat_input, time)
            return train_graph.bond_ffn(self, bond_feat_input, time.reshape(-1, 1), node_edges=node_feat_input)
        net = self._net()
        _lib._need_gpu(bond_feat_input, node_feat_input, time)
        eng = net._engine()
        E, dev = bond_feat_input.shape[0], bond_feat_input.device
        ident = torch.arange(E, dtype=torch.int64).unsqueeze(0).repeat(2, 1)
        g = _lib.Graph(ident, torch.zeros(E, dtype=torch.int64), 1)
        bond, node, tt = _lib.f32c(bond_feat_input), _lib.f32c(node_feat_input), _lib.f32c(time).view(-1)
        out = torch.empty(E, bond.shape[1], dtype=torch.float32, device=dev)
        ws, nb = g.workspace(dev)
        _lib.check(_lib.lib().mdx_bond_ffn(eng.h, g.h, self._index, self._side, _lib.ptr(bond), _lib.ptr(node), _lib.ptr(tt),
                                           _lib.ptr(out), ws, nb, _lib.stream()))
        torch.cuda.current_stream().synchronize()  # `g` and its workspace die with this frame
        return out


class EdgeBlock(_Block):
    def __init__(self, edge_dim, node_dim, hidden_dim=None, use_gate=True):
        super().__init__()
        self.use_gate = use_gate
        self.edge_dim = edge_dim
        inter_dim = edge_dim * 2 if hidden_dim is None else hidden_dim
        self.bond_ffn_left = BondFFN(edge_dim, node_dim, inter_dim=inter_dim, use_gate=use_gate)
        self.bond_ffn_right = BondFFN(edge_dim, node_dim, inter_dim=inter_dim, use_gate=use_gate)
        self.node_ffn_left = Linear(node_dim, edge_dim)
        self.node_ffn_right = Linear(node_dim, edge_dim)
        self.self_ffn = Linear(edge_dim, edge_dim)
        self.layer_norm = nn.LayerNorm(edge_dim)
        self.out_transform = Linear(edge_dim, edge_dim)
        self.act = nn.ReLU()

    def forward(self, h_bond, bond_index, h_node, bond_time):
        """h_bond (E,He), bond_index (2,E), h_node (N,H), bond_time (E,1) -> edge update (E,He), no residual."""
        net = self._net()
        _lib._need_gpu(h_bond, h_node, bond_time, bond_index)
        eng = net._engine()
        g = net._graph(bond_index, h_node.shape[0])
        out = torch.empty(h_bond.shape[0], self.edge_dim, dtype=torch.float32, device=h_bond.device)
        ws, nb = g.workspace(h_bond.device)
        bc, nc, tc = _lib.f32c(h_bond), _lib.f32c(h_node), _lib.f32c(bond_time).view(-1)   # alive until the launch
        _lib.check(_lib.lib().mdx_edge_block(eng.h, g.h, self._index, _lib.ptr(bc), _lib.ptr(nc), _lib.ptr(tc), _lib.ptr(out), ws,
                                             nb, _lib.stream()))
        return out


class PosUpdate(_Block):
    def __init__(self, node_dim, edge_dim, hidden_dim, use_gate):
        super().__init__()
        self.left_lin_edge = MLP(node_dim, edge_dim, hidden_dim)
        self.right_lin_edge = MLP(node_dim, edge_dim, hidden_dim)
        self.edge_lin = BondFFN(edge_dim, edge_dim, node_dim, use_gate, out_dim=1)

    def forward(self, h_node, h_edge, edge_index, relative_vec, distance, edge_time):
        """-> delta_pos (N,3) = sum_left  w_e * rel / d / (d + 1)."""
        net = self._net()
        _lib._need_gpu(h_node, h_edge, relative_vec, distance, edge_time, edge_index)
        eng = net._engine()
        g = net._graph(edge_index, h_node.shape[0])
        out = torch.empty(h_node.shape[0], 3, dtype=torch.float32, device=h_node.device)
        ws, nb = g.workspace(h_node.device)
        nc, ec, rc = _lib.f32c(h_node), _lib.f32c(h_edge), _lib.f32c(relative_vec)   # alive until the launch
        dc, tc = _lib.f32c(distance).view(-1), _lib.f32c(edge_time).view(-1)
        _lib.check(_lib.lib().mdx_pos_update(eng.h, g.h, self._index, _lib.ptr(nc), _lib.ptr(ec), _lib.ptr(rc), _lib.ptr(dc),
                                             _lib.ptr(tc), _lib.ptr(out), ws, nb, _lib.stream()))
        return out


class NodeEdgeNet(Module):
    def __getstate__(self):
        # the packed-weight engine is a device handle: never copied or pickled (deepcopy / torch.save of the module
        # rebuild it lazily from the state_dict on first use)
        d = self.__dict__.copy()
        d['_eng'], d['_eng_sig'] = None, None
        return d

    def __init__(self, node_dim, edge_dim, num_blocks, cutoff, use_gate, **kwargs):
        super().__init__()
        self.node_dim, self.edge_dim, self.num_blocks = node_dim, edge_dim, num_blocks
        self.cutoff, self.use_gate, self.kwargs = cutoff, use_gate, kwargs
        num_gaussians = kwargs.get('num_gaussians', 16)
        if num_gaussians > NG_KERNEL:
            raise NotImplementedError(f'num_gaussians > {NG_KERNEL} is not built (the kernels smear into one 16-wide feature tile)')
        start = kwargs.get('start', 0)
        self.distance_expansion = GaussianSmearing(start=start, stop=cutoff, num_gaussians=num_gaussians)
        self.update_edge = not ('update_edge' in kwargs and not kwargs['update_edge'])
        self.update_pos = not ('update_pos' in kwargs and not kwargs['update_pos'])
        # update_edge=False (models/graph.py:317-320,352-361): edge features are re-derived from the distances in every block, there
        # are no EdgeBlocks; the fused kernels get zero-update EdgeBlocks and a zero He-part of edge_embs (synth_params)
        input_edge_dim = (edge_dim if self.update_edge else 0) + num_gaussians
        self.node_blocks_with_edge = ModuleList()
        self.edge_embs = ModuleList()
        self.edge_blocks = ModuleList()
        self.pos_blocks = ModuleList()
        for _ in range(num_blocks):
            self.node_blocks_with_edge.append(NodeBlock(node_dim=node_dim, edge_dim=edge_dim, hidden_dim=node_dim,
                                                        use_gate=use_gate))
            self.edge_embs.append(Linear(input_edge_dim, edge_dim))
            if self.update_edge:
                self.edge_blocks.append(EdgeBlock(edge_dim=edge_dim, node_dim=node_dim, use_gate=use_gate))
            if self.update_pos:
                self.pos_blocks.append(PosUpdate(node_dim=node_dim, edge_dim=edge_dim, hidden_dim=edge_dim,
                                                 use_gate=use_gate))
        ref = weakref.ref(self)
        for lst in (self.node_blocks_with_edge, self.edge_blocks, self.pos_blocks):
            for i, blk in enumerate(lst):
                blk._owner, blk._index = ref, i
        for i, blk in enumerate(self.edge_blocks):
            for side, ffn in enumerate((blk.bond_ffn_left, blk.bond_ffn_right)):
                ffn._owner, ffn._index, ffn._side = ref, i, side
        self._eng = None
        self._eng_sig = None

    # ---- engine plumbing --------------------------------------------------------------------
    # None = follow _lib.default_matrix_path (exact fp32 unless MOLDIFF_MATRIX_PATH says otherwise); or 'exact_f32' / 'split_f16'
    matrix_path = None

    def _engine(self):
        sig = _sig(self)
        if self._eng is None or sig != self._eng_sig:
            eng = _lib.Model(_lib.MDX_KIND_NET, num_blocks=self.num_blocks, cutoff=self.cutoff, update_pos=self.update_pos,
                             node_dim=self.node_dim, edge_dim=self.edge_dim,
                             num_gaussians=NG_KERNEL, smear_start=self.distance_expansion.start)
            eng.upload({**self.state_dict(), **synth_gates(self)})
            self._eng, self._eng_sig = eng, sig
        return self._eng.use_matrix_path(self.matrix_path)

    @staticmethod
    def _graph(edge_index, n_nodes):
        bn = torch.zeros(n_nodes, dtype=torch.int64)
        key = ('net', edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, n_nodes)
        g = _lib._cache_get(key)
        if g is None:
            g = _lib.Graph(edge_index, bn, 1)
            g._keepalive = edge_index
            _lib._cache_put(key, g)
        return g

    def forward(self, h_node, pos_node, h_edge, edge_index, node_time, edge_time):
        _lib._need_gpu(h_node, pos_node, h_edge, edge_index, node_time, edge_time)
        eng = self._engine()
        g = self._graph(edge_index, h_node.shape[0])
        dev = h_node.device
        hn = torch.empty(h_node.shape[0], self.node_dim, dtype=torch.float32, device=dev)
        po = torch.empty(h_node.shape[0], 3, dtype=torch.float32, device=dev)
        he = torch.empty(h_edge.shape[0], self.edge_dim, dtype=torch.float32, device=dev)
        ws, nb = g.workspace(dev)
        a, b, c = _lib.f32c(h_node), _lib.f32c(pos_node), _lib.f32c(h_edge)   # alive until the launch
        tn, te = _lib.f32c(node_time).view(-1), _lib.f32c(edge_time).view(-1)
        _lib.check(_lib.lib().mdx_net_forward(eng.h, g.h, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(tn), _lib.ptr(te),
                                              _lib.ptr(hn), _lib.ptr(po), _lib.ptr(he), ws, nb, _lib.stream()))
        return hn, po, he
