"""Fused row-owner training operators (round 6, csrc/mdx_train_fused.hip): the EdgeBlock's BondFFN + scatter_sum as one forward and
one backward launch, against (a) the per-operator composition of the same module in the same float16 autocast arithmetic (same
rounding points: agreement to a few float16 ulp) and (b) the fp64 evaluation of the reference formula (models/graph.py:133-141,
:272-279 through torch autograd)."""
import numpy as np
import pytest
import torch

from moldiff_amd import graph as G
from moldiff_amd import train_graph as TG
from moldiff_amd import train_ops as T
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(sizes, seed, scale=1.0):
    g = U.rng(seed)
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, E = len(bn), ei.shape[1]
    m = G.BondFFN(64, 256, 128, True).to(DEV)
    with torch.no_grad():
        for k, p in sorted(m.named_parameters()):
            if p.dim() == 2:
                p.copy_(torch.from_numpy((g.standard_normal(tuple(p.shape)) * (1.2 / np.sqrt(p.shape[1]))).astype(np.float32)))
            elif 'net.1.' in k and k.endswith('weight'):
                p.copy_(torch.from_numpy((1.0 + 0.4 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
            else:
                p.copy_(torch.from_numpy((0.3 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
    x = torch.from_numpy((g.standard_normal((E, 64)) * scale).astype(np.float32)).to(DEV).half()
    h = torch.from_numpy(g.standard_normal((N, 256)).astype(np.float32)).to(DEV).half()
    te = torch.from_numpy(g.random((E, 1)).astype(np.float32)).to(DEV)
    gS = torch.from_numpy(g.standard_normal((N, 64)).astype(np.float32)).to(DEV)
    tg = TG.TrainGraph(ei.to(DEV), N)
    return m, x, h, te, gS, tg


def _run(m, x, h, te, gS, tg, fused):
    m.zero_grad(set_to_none=True)
    x = x.clone().requires_grad_(True)
    h = h.clone().requires_grad_(True)
    old, old_rows = T._FUSED, T.FUSED_MIN_ROWS
    T._FUSED, T.FUSED_MIN_ROWS = fused, 1
    try:
        with T.precision('fp16'):
            out = TG.bond_ffn_scatter(m, x, te, h, tg.left, tg.right)
            out.backward(gS)
    finally:
        T._FUSED, T.FUSED_MIN_ROWS = old, old_rows
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    return out.detach(), x.grad.detach(), h.grad.detach(), grads


def _rel2(a, b):
    """relative L2 distance"""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


@pytest.mark.parametrize('sizes', [[5, 7, 4], [24, 31, 18, 27, 22, 25, 30, 19, 26, 23], [2, 2, 3]])
def test_fused_bondffn_equals_the_per_operator_composition(sizes):
    """Same arithmetic, same rounding points, another summation order inside the LayerNorm statistics and the MFMA chains: outputs and
    every gradient within 2e-3 of the per-operator path relative to the tensor's largest entry (a float16 ulp is 1e-3)."""
    args = _setup(sizes, 11)
    o1, gx1, gh1, gr1 = _run(*args, fused=True)
    o0, gx0, gh0, gr0 = _run(*args, fused=False)
    assert o1.dtype == o0.dtype and gx1.dtype == gx0.dtype and gh1.dtype == gh0.dtype
    assert _rel(o1, o0) < 2e-3, _rel(o1, o0)
    assert _rel(gx1, gx0) < 4e-3, _rel(gx1, gx0)
    assert _rel(gh1, gh0) < 4e-3, _rel(gh1, gh0)
    assert set(gr1) == set(gr0)
    for k in gr0:
        assert torch.isfinite(gr1[k]).all(), k
        assert _rel(gr1[k], gr0[k]) < 6e-3, (k, _rel(gr1[k], gr0[k]))


def test_fused_bondffn_against_fp64_autograd_of_the_reference_formula():
    """The fused path vs torch autograd in float64 on the reference's formula: forward within 3e-3 of the output scale.  Gradients: in
    float16 arithmetic ~1e-3 of the 800 k LayerNorm->ReLU pre-activations of this batch land on the other side of zero than in float64,
    and each such kink event changes its row's gradient discretely -- the MAXIMUM over rows is therefore O(10 %) for ANY float16
    evaluation (measured 18 % on dL/dX for both paths, to four digits the same number: the fused kernels take the per-operator path's
    rounding points, so they flip the same units).  Asserted: relative L2 distance to float64 below 6 % and equal to the per-operator
    path's within 2 % of it, i.e. the fused node is as far from exact arithmetic as autocast's own arithmetic, not further."""
    m, x, h, te, gS, tg = _setup([24, 31, 18, 27, 22, 25, 30, 19], 13)
    o1, gx1, gh1, gr1 = _run(m, x, h, te, gS, tg, fused=True)
    o0, gx0, gh0, gr0 = _run(m, x, h, te, gS, tg, fused=False)
    P = {k: p.detach().double().clone().requires_grad_(True) for k, p in m.named_parameters()}
    xd, hd = x.double().clone().requires_grad_(True), h.double().clone().requires_grad_(True)
    li, ri = tg.left.index, tg.right.index
    F = torch.nn.functional

    def mlp(pre, v):
        v = F.linear(v, P[pre + '.net.0.weight'], P[pre + '.net.0.bias'])
        v = torch.relu(F.layer_norm(v, (v.shape[1],), P[pre + '.net.1.weight'], P[pre + '.net.1.bias'], 1e-5))
        return F.linear(v, P[pre + '.net.3.weight'], P[pre + '.net.3.bias'])
    node = hd[li]
    inter = mlp('inter_module', F.linear(xd, P['bond_linear.weight']) * F.linear(node, P['node_linear.weight']))
    gate = mlp('gate', torch.cat([xd, node, te.double()], -1))
    out = torch.zeros(hd.shape[0], 64, dtype=torch.float64, device=DEV).index_add_(0, ri, inter * torch.sigmoid(gate))
    out.backward(gS.double())
    assert _rel(o1, out.detach()) < 3e-3
    rows = [('dX', gx1, gx0, xd.grad), ('dh', gh1, gh0, hd.grad)] + [(k, gr1[k], gr0[k], P[k].grad) for k in gr1]
    errs = [(name, _rel2(got, ref), _rel2(base, ref), _rel(got, ref), _rel(base, ref)) for name, got, base, ref in rows]
    print('\n[fused BondFFN vs fp64]  tensor: relative L2 fused, per-operator | max-norm fused, per-operator')
    for name, e1, e0, m1, m0 in errs:
        print(f'    {name:28s} {e1:.3e}  {e0:.3e} | {m1:.3e}  {m0:.3e}')
    for name, e1, e0, m1, m0 in errs:
        assert e1 < 6e-2, (name, e1)
        assert e1 <= 1.02 * e0 + 1e-4, (name, e1, e0)


def test_fused_bondffn_with_the_gradient_sink_equals_autograd_accumulation():
    """Inside Trainer.step the weight gradients bypass autograd (train_ops.grad_sink): the fused node's six weight-gradient contractions
    and its LayerNorm-parameter partial rows must land in the flat gradient buffer exactly as the returned tensors would."""
    from moldiff_amd.trainer import FlatParams
    m, x, h, te, gS, tg = _setup([24, 31, 18, 27, 22, 25, 30, 19], 17)
    _, _, _, ref = _run(m, x, h, te, gS, tg, fused=True)
    flat = FlatParams(m)
    flat.grad.zero_()
    old, old_rows = T._FUSED, T.FUSED_MIN_ROWS
    T._FUSED, T.FUSED_MIN_ROWS = True, 1
    try:
        with T.grad_sink(flat), T.precision('fp16'):
            out = TG.bond_ffn_scatter(m, x.clone().requires_grad_(True), te, h.clone().requires_grad_(True), tg.left, tg.right)
            out.backward(gS)
    finally:
        T._FUSED, T.FUSED_MIN_ROWS = old, old_rows
    for k, p in m.named_parameters():
        got = p.grad if p.grad is not None else None
        assert got is not None, k
        assert _rel(got, ref[k]) < 2e-3, (k, _rel(got, ref[k]))


# ---------------------------------------------------------------------------------------------------------------------------
# EdgeBlock (two fused BondFFNs + the fused tail with the residual) against the per-operator composition
# ---------------------------------------------------------------------------------------------------------------------------
def _edge_block_run(m, x, h, te, g_out, tg, fused):
    m.zero_grad(set_to_none=True)
    x = x.clone().requires_grad_(True)
    h = h.clone().requires_grad_(True)
    old, old_rows = T._FUSED, T.FUSED_MIN_ROWS
    T._FUSED, T.FUSED_MIN_ROWS = fused, 1
    try:
        with T.precision('fp16'):
            out = TG.edge_block(m, x, tg, h, te, residual=True)
            out.backward(g_out)
    finally:
        T._FUSED, T.FUSED_MIN_ROWS = old, old_rows
    return out.detach(), x.grad.detach(), h.grad.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize('sizes', [[5, 7, 4], [24, 31, 18, 27, 22, 25, 30, 19, 26, 23]])
def test_fused_edge_block_with_residual_equals_the_per_operator_composition(sizes):
    """h + EdgeBlock(h) (models/graph.py:268-294, :360) with both BondFFNs and the tail fused vs the per-operator path in the same float16
    arithmetic: output within 2 float16 ulp of its scale; gradients within 1 % in L2 (individual rows may sit on different sides of a ReLU kink once the
    two paths' LayerNorm statistics differ in the last bit)."""
    g = U.rng(23)
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, E = len(bn), ei.shape[1]
    m = G.EdgeBlock(64, 256).to(DEV)
    with torch.no_grad():
        for k, p in sorted(m.named_parameters()):
            if p.dim() == 2:
                p.copy_(torch.from_numpy((g.standard_normal(tuple(p.shape)) * (1.0 / np.sqrt(p.shape[1]))).astype(np.float32)))
            elif ('net.1.' in k or 'layer_norm' in k) and k.endswith('weight'):
                p.copy_(torch.from_numpy((1.0 + 0.3 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
            else:
                p.copy_(torch.from_numpy((0.2 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
    x = torch.from_numpy(g.standard_normal((E, 64)).astype(np.float32)).to(DEV).half()
    h = torch.from_numpy(g.standard_normal((N, 256)).astype(np.float32)).to(DEV).half()
    te = torch.from_numpy(g.random((E, 1)).astype(np.float32)).to(DEV)
    g_out = torch.from_numpy(g.standard_normal((E, 64)).astype(np.float32)).to(DEV).half()
    tg = TG.TrainGraph(ei.to(DEV), N)
    o1, gx1, gh1, gr1 = _edge_block_run(m, x, h, te, g_out, tg, True)
    o0, gx0, gh0, gr0 = _edge_block_run(m, x, h, te, g_out, tg, False)
    assert o1.dtype == o0.dtype == torch.float16 and gx1.dtype == gx0.dtype
    print(f'\n[fused EdgeBlock vs per-operator] out max {_rel(o1, o0):.2e}  dX L2 {_rel2(gx1, gx0):.2e}  dh L2 {_rel2(gh1, gh0):.2e}')
    for k in gr0:
        print(f'    {k:40s} L2 {_rel2(gr1[k], gr0[k]):.2e}')
    assert _rel(o1, o0) < 3e-3, _rel(o1, o0)
    assert _rel2(gx1, gx0) < 1e-2 and _rel2(gh1, gh0) < 1e-2, (_rel2(gx1, gx0), _rel2(gh1, gh0))
    for k in gr0:
        assert torch.isfinite(gr1[k]).all(), k
        assert _rel2(gr1[k], gr0[k]) < 1e-2, (k, _rel2(gr1[k], gr0[k]))


# ---------------------------------------------------------------------------------------------------------------------------
# PosUpdate with the fused front of its BondFFN against the per-operator composition
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('sizes', [[5, 7, 4], [24, 31, 18, 27, 22, 25, 30, 19, 26, 23]])
def test_fused_pos_update_equals_the_per_operator_composition(sizes):
    """PosUpdate.forward (models/graph.py:384-396) with gather / product / bond_linear / node_linear / gate MLP fused into one launch each
    way: the position increment and every gradient against the per-operator path in the same float16 arithmetic."""
    g = U.rng(29)
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, E = len(bn), ei.shape[1]
    m = G.PosUpdate(256, 64, 64, True).to(DEV)
    with torch.no_grad():
        for k, p in sorted(m.named_parameters()):
            if p.dim() == 2:
                p.copy_(torch.from_numpy((g.standard_normal(tuple(p.shape)) * (1.0 / np.sqrt(p.shape[1]))).astype(np.float32)))
            elif 'net.1.' in k and k.endswith('weight'):
                p.copy_(torch.from_numpy((1.0 + 0.3 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
            else:
                p.copy_(torch.from_numpy((0.2 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
    he = torch.from_numpy(g.standard_normal((E, 64)).astype(np.float32)).to(DEV).half()
    hn = torch.from_numpy(g.standard_normal((N, 256)).astype(np.float32)).to(DEV).half()
    pos = torch.from_numpy((g.standard_normal((N, 3)) * 2.0).astype(np.float32)).to(DEV)
    te = torch.from_numpy(g.random((E, 1)).astype(np.float32)).to(DEV)
    g_out = torch.from_numpy(g.standard_normal((N, 3)).astype(np.float32)).to(DEV)
    tg = TG.TrainGraph(ei.to(DEV), N)

    def run(fused):
        m.zero_grad(set_to_none=True)
        x, h, p = he.clone().requires_grad_(True), hn.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        old, old_rows = T._FUSED, T.FUSED_MIN_ROWS
        T._FUSED, T.FUSED_MIN_ROWS = fused, 1
        try:
            with T.precision('fp16'):
                rel, dist = T.edge_geom(p, tg.left, tg.right)
                out = TG.pos_update(m, h, x, tg, rel, dist, te)
                out.backward(g_out)
        finally:
            T._FUSED, T.FUSED_MIN_ROWS = old, old_rows
        return out.detach(), x.grad.detach(), h.grad.detach(), p.grad.detach(), {k: q.grad.detach().clone() for k, q in m.named_parameters()}

    o1, gx1, gh1, gp1, gr1 = run(True)
    o0, gx0, gh0, gp0, gr0 = run(False)
    print(f'\n[fused PosUpdate vs per-operator] out max {_rel(o1, o0):.2e}  dX L2 {_rel2(gx1, gx0):.2e}  dh L2 {_rel2(gh1, gh0):.2e}  dpos L2 {_rel2(gp1, gp0):.2e}')
    for k in gr0:
        print(f'    {k:40s} L2 {_rel2(gr1[k], gr0[k]):.2e}')
    assert _rel(o1, o0) < 3e-3, _rel(o1, o0)
    assert _rel2(gx1, gx0) < 1e-2 and _rel2(gh1, gh0) < 1e-2 and _rel2(gp1, gp0) < 1e-2
    for k in gr0:
        assert torch.isfinite(gr1[k]).all(), k
        assert _rel2(gr1[k], gr0[k]) < 1e-2, (k, _rel2(gr1[k], gr0[k]))


# ---------------------------------------------------------------------------------------------------------------------------
# NodeBlock with the fused message path (streamed weights) against the per-operator composition
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('sizes', [[5, 7, 4], [24, 31, 18, 27, 22, 25, 30, 19, 26, 23]])
def test_fused_node_block_equals_the_per_operator_composition(sizes):
    """NodeBlock.forward (models/graph.py:29-55) with edge_net, the product with node_net(x)[col], msg_net, the gate MLP and the sigmoid
    product in one launch each way (weights streamed as float16 A-operand packs): output and every gradient against the per-operator
    path in the same float16 arithmetic."""
    g = U.rng(31)
    bn, hei, bh, ei, be = U.graph_from_sizes(sizes)
    N, E = len(bn), ei.shape[1]
    m = G.NodeBlock(256, 64, 256, True).to(DEV)
    with torch.no_grad():
        for k, p in sorted(m.named_parameters()):
            if p.dim() == 2:
                p.copy_(torch.from_numpy((g.standard_normal(tuple(p.shape)) * (1.0 / np.sqrt(p.shape[1]))).astype(np.float32)))
            elif ('net.1.' in k or 'layer_norm' in k) and k.endswith('weight'):
                p.copy_(torch.from_numpy((1.0 + 0.3 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
            else:
                p.copy_(torch.from_numpy((0.2 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
    he = torch.from_numpy(g.standard_normal((E, 64)).astype(np.float32)).to(DEV).half()
    hn = torch.from_numpy(g.standard_normal((N, 256)).astype(np.float32)).to(DEV).half()
    tn = torch.from_numpy(g.random((N, 1)).astype(np.float32)).to(DEV)
    g_out = torch.from_numpy(g.standard_normal((N, 256)).astype(np.float32)).to(DEV).half()
    tg = TG.TrainGraph(ei.to(DEV), N)

    def run(fused):
        m.zero_grad(set_to_none=True)
        x, h = he.clone().requires_grad_(True), hn.clone().requires_grad_(True)
        old, old_rows = T._FUSED, T.FUSED_MIN_ROWS
        T._FUSED, T.FUSED_MIN_ROWS = fused, 1
        try:
            with T.precision('fp16'):
                out = TG.node_block(m, h, tg, x, tn)
                out.backward(g_out)
        finally:
            T._FUSED, T.FUSED_MIN_ROWS = old, old_rows
        return out.detach(), x.grad.detach(), h.grad.detach(), {k: q.grad.detach().clone() for k, q in m.named_parameters()}

    o1, gx1, gh1, gr1 = run(True)
    o0, gx0, gh0, gr0 = run(False)
    print(f'\n[fused NodeBlock vs per-operator] out max {_rel(o1, o0):.2e}  dX L2 {_rel2(gx1, gx0):.2e}  dh L2 {_rel2(gh1, gh0):.2e}')
    for k in gr0:
        print(f'    {k:40s} L2 {_rel2(gr1[k], gr0[k]):.2e}')
    assert _rel(o1, o0) < 3e-3, _rel(o1, o0)
    assert _rel2(gx1, gx0) < 1e-2 and _rel2(gh1, gh0) < 1e-2
    for k in gr0:
        assert torch.isfinite(gr1[k]).all(), k
        assert _rel2(gr1[k], gr0[k]) < 1e-2, (k, _rel2(gr1[k], gr0[k]))
