"""Each layer operator of the training path (csrc/mdx_train.hip via moldiff_amd/train_ops.py) against the torch op it
replaces: forward values and every gradient, on awkward shapes (K, N not multiples of 4/16, ragged segments, empty
segments).  fp32; tolerances relative to the magnitude of the quantity compared."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util as U
from moldiff_amd import train_ops as T

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _leaf(g, *shape, scale=1.0):
    return (U.t32(g.standard_normal(shape)) * scale).to(DEV).requires_grad_(True)


@pytest.mark.parametrize('M,K,N', [(1, 8, 246), (37, 321, 256), (1000, 64, 64), (5003, 129, 32), (4097, 80, 64), (300, 256, 1),
                                   (513, 6, 54), (20000, 256, 256)])
def test_linear_forward_and_all_gradients(M, K, N):
    g = U.rng(M + K + N)
    x, w, b = _leaf(g, M, K), _leaf(g, N, K, scale=K ** -0.5), _leaf(g, N)
    gy = U.t32(g.standard_normal((M, N))).to(DEV)
    y = T.linear(x, w, b)
    y.backward(gy)
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    y2 = F.linear(x2.double(), w2.double(), b2.double())
    y2.backward(gy.double())
    assert _rel(y, y2) < 2e-6
    assert _rel(x.grad, x2.grad) < 2e-6 and _rel(w.grad, w2.grad) < 5e-6 and _rel(b.grad, b2.grad) < 5e-6


@pytest.mark.parametrize('K,N', [(256, 256), (64, 256), (256, 64), (80, 64), (128, 128), (64, 128), (128, 64), (32, 64), (64, 32),
                                 (16, 64), (64, 20), (256, 252)])
@pytest.mark.parametrize('with_addend', [False, True])
def test_linear_on_many_rows_takes_the_row_owner_kernel(K, N, with_addend):
    """From train_ops.ROWS_MIN rows on, Linear forward and grad_input run on csrc/mdx_linear_rows.hip (weight packed per call,
    read transposed for grad_input).  Ragged row count, every built (K, N) class incl. N not a multiple of 16 / 32; the input
    is a column slice of a wider tensor (row stride != K)."""
    M = T.ROWS_MIN + 123
    assert T._L().mdx_op_linear_rows_supported(N, K) == 1
    g = U.rng(K * 1000 + N)
    wide = _leaf(g, M, K + 16)
    w, b = _leaf(g, N, K, scale=K ** -0.5), _leaf(g, N)
    add = _leaf(g, M, N) if with_addend else None
    gy = U.t32(g.standard_normal((M, N))).to(DEV)
    x = wide[:, 8:8 + K]
    assert T.linear_rows_ok(T._rows(x), N, K, add)
    y = T.linear(x, w, b, add)
    y.backward(gy)
    w2, b2, wide2 = (t.detach().clone().requires_grad_(True) for t in (w, b, wide))
    y2 = F.linear(wide2[:, 8:8 + K].double(), w2.double(), b2.double())
    if with_addend:
        a2 = add.detach().clone().requires_grad_(True)
        y2 = y2 + a2.double()
    y2.backward(gy.double())
    assert _rel(y, y2) < 2e-6
    assert _rel(wide.grad, wide2.grad) < 2e-6 and _rel(w.grad, w2.grad) < 5e-6 and _rel(b.grad, b2.grad) < 5e-6
    if with_addend:
        assert _rel(add.grad, a2.grad) < 1e-6


def test_linear_without_bias_and_noncontiguous_input():
    g = U.rng(5)
    big = _leaf(g, 700, 100)
    w = _leaf(g, 30, 50, scale=0.1)
    y = T.linear(big[:, 25:75], w)
    y.sum().backward()
    b2, w2 = big.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    F.linear(b2[:, 25:75], w2).sum().backward()
    assert _rel(big.grad, b2.grad) < 2e-6 and _rel(w.grad, w2.grad) < 5e-6


@pytest.mark.parametrize('M,Fdim,relu', [(1, 32, True), (777, 64, True), (4099, 128, True), (5000, 256, True), (333, 256, False),
                                         (100, 246, True)])
def test_layernorm_relu(M, Fdim, relu):
    g = U.rng(M + Fdim)
    x, ga, be = _leaf(g, M, Fdim, scale=2.0), _leaf(g, Fdim), _leaf(g, Fdim, scale=0.3)
    gy = U.t32(g.standard_normal((M, Fdim))).to(DEV)
    y = T.ln_relu(x, ga, be, relu)
    y.backward(gy)
    x2, g2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, ga, be))
    y2 = F.layer_norm(x2, (Fdim,), g2, b2, 1e-5)
    y2 = F.relu(y2) if relu else y2
    y2.backward(gy)
    assert _rel(y, y2) < 5e-6
    assert _rel(x.grad, x2.grad) < 2e-5 and _rel(ga.grad, g2.grad) < 2e-5 and _rel(be.grad, b2.grad) < 2e-5


@pytest.mark.parametrize('op,fn', [(T.add, lambda a, b: a + b), (T.sub, lambda a, b: a - b), (T.mul, lambda a, b: a * b),
                                   (T.gate, lambda a, b: a * torch.sigmoid(b))])
@pytest.mark.parametrize('shape', [(1234, 7), (1024, 8)])      # scalar path / 16-byte vector path
def test_elementwise_pairs(op, fn, shape):
    g = U.rng(9)
    a, b = _leaf(g, *shape), _leaf(g, *shape)
    gy = U.t32(g.standard_normal(shape)).to(DEV)
    op(a, b).backward(gy)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    fn(a2, b2).backward(gy)
    assert _rel(op(a, b), fn(a2, b2)) < 1e-6 and _rel(a.grad, a2.grad) < 2e-6 and _rel(b.grad, b2.grad) < 2e-6


def test_gather_and_scatter_sum_are_adjoint_and_match_torch():
    g = U.rng(11)
    n, M, Fd = 50, 2000, 24
    idx = torch.from_numpy(g.integers(0, n - 5, M)).to(DEV)      # targets n-5..n-1 stay empty
    plan = T.IndexPlan(idx, n)
    x, src = _leaf(g, n, Fd), _leaf(g, M, Fd)
    gy, gs = U.t32(g.standard_normal((M, Fd))).to(DEV), U.t32(g.standard_normal((n, Fd))).to(DEV)
    T.gather(x, plan).backward(gy)
    T.scatter_sum(src, plan).backward(gs)
    x2, s2 = x.detach().clone().requires_grad_(True), src.detach().clone().requires_grad_(True)
    x2[idx].backward(gy)
    torch.zeros(n, Fd, device=DEV).index_add(0, idx, s2).backward(gs)
    assert torch.equal(T.gather(x, plan), x2[idx])
    assert _rel(T.scatter_sum(src, plan), torch.zeros(n, Fd, device=DEV).index_add(0, idx, s2)) < 2e-6
    assert _rel(x.grad, x2.grad) < 2e-6 and torch.equal(src.grad, s2.grad)
    assert torch.equal(T.scatter_sum(src, plan), T.scatter_sum(src, plan))          # deterministic


def test_edge_geometry_smearing_and_force():
    g = U.rng(13)
    n, E = 40, 900
    l = torch.from_numpy(g.integers(0, n, E)).to(DEV)
    r = (l + 1 + torch.from_numpy(g.integers(0, n - 1, E)).to(DEV)) % n          # never equal to l
    pl, pr = T.IndexPlan(l, n), T.IndexPlan(r, n)
    pos, w = _leaf(g, n, 3, scale=3.0), _leaf(g, E, 1)
    off = torch.linspace(0, 15, 16, device=DEV)
    coef = -0.5 / (off[1] - off[0]) ** 2 * torch.linspace(1.0, 0.3, 16, device=DEV)
    gs, gf = U.t32(g.standard_normal((E, 16))).to(DEV), U.t32(g.standard_normal((E, 3))).to(DEV)

    def ours(pos, w):
        rel, d = T.edge_geom(pos, pl, pr)
        return T.smear(d, off, coef, 0.0, 5.0), T.force(w, rel, d)

    def ref(pos, w):
        rel = pos[l] - pos[r]
        d = torch.norm(rel, dim=-1)
        sm = torch.exp(coef * (d.clamp(0.0, 5.0).unsqueeze(-1) - off) ** 2)
        return sm, w * rel / d.unsqueeze(-1) / (d.unsqueeze(-1) + 1)

    a = ours(pos, w)
    (a[0] * gs).sum().backward(retain_graph=True)
    (a[1] * gf).sum().backward()
    p2, w2 = pos.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    b = ref(p2, w2)
    ((b[0] * gs).sum() + (b[1] * gf).sum()).backward()
    assert _rel(a[0], b[0]) < 2e-6 and _rel(a[1], b[1]) < 2e-6
    assert _rel(pos.grad, p2.grad) < 2e-5 and _rel(w.grad, w2.grad) < 2e-6


def test_cpu_tensors_raise():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        T.linear(torch.zeros(4, 4), torch.zeros(4, 4))


def _bf(x):
    return x.detach().bfloat16().float()


@pytest.mark.parametrize('M,K,N', [(37, 321, 256), (5003, 129, 32), (4097, 80, 64), (20000, 256, 256), (513, 6, 54)])
def test_bf16_operand_linear_equals_fp32_gemm_of_bf16_rounded_operands(M, K, N):
    """precision('bf16'): products of bf16-rounded operands are exact in fp32, so forward and all three gradients must equal
    an fp32 (here fp64) GEMM of the rounded tensors up to fp32 summation order."""
    g = U.rng(M + K + N + 1)
    x, w, b = _leaf(g, M, K), _leaf(g, N, K, scale=K ** -0.5), _leaf(g, N)
    gy = U.t32(g.standard_normal((M, N))).to(DEV)
    with T.precision('bf16'):
        y = T.linear(x, w, b)
    y.backward(gy)                       # outside the context: the layer remembers its precision
    xr, wr, gr = _bf(x).double(), _bf(w).double(), _bf(gy).double()
    assert _rel(y, xr @ wr.T + b.detach().double()) < 3e-6
    assert _rel(x.grad, gr @ wr) < 3e-6
    assert _rel(w.grad, gr.T @ xr) < 1e-5
    assert _rel(b.grad, gr.sum(0)) < 1e-5
    # and it is close to, but not the same as, the fp32 result
    y32 = T.linear(x, w, b)
    assert 1e-5 < _rel(y, y32) < 2e-2


def test_mul_gather_matches_mul_of_gather():
    g = U.rng(21)
    n, M, Fd = 60, 3000, 64
    idx = torch.from_numpy(g.integers(0, n - 3, M)).to(DEV)
    plan = T.IndexPlan(idx, n)
    a, t = _leaf(g, M, Fd), _leaf(g, n, Fd)
    gy = U.t32(g.standard_normal((M, Fd))).to(DEV)
    y = T.mul_gather(a, t, plan)
    y.backward(gy)
    a2, t2 = a.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
    y2 = a2 * t2[idx]
    y2.backward(gy)
    assert torch.equal(y, y2) and torch.equal(a.grad, a2.grad)
    assert _rel(t.grad, t2.grad) < 2e-6


def test_half_storage_operators_equal_their_fp32_container_twins():
    """Half storage (round 3): in the mixed-precision mode every Linear / LayerNorm / product result is a float16 VALUE; `precision('fp16')`
    keeps it in a float16 container, `precision('fp16_f32store')` in an fp32 one.  On float16-representable inputs each typed operator
    (`mdx_op_*_t`) must return what its fp32-container twin returns, rounded to float16 where the container is float16 -- forward and
    backward, on ragged row counts."""
    from moldiff_amd import train_ops as T
    g = U.rng(31)
    dev = 'cuda:0'
    M, Nn = 1003, 77
    h = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).to(dev).half()
    idx = torch.from_numpy(g.integers(0, Nn, M)).to(dev)
    plan = T.IndexPlan(idx, Nn)
    x, w, b = h(M, 64), h(128, 64).float(), h(128).float()
    gam, bet = (1 + 0.1 * h(128)).float(), (0.1 * h(128)).float()
    tab, a2 = h(Nn, 128), h(M, 128)
    gout = h(M, 128)

    def run(mode, half):
        cast = (lambda t: t) if half else (lambda t: t.float())
        xs = [cast(t).clone().requires_grad_(True) for t in (x, tab, a2)]
        ws = [t.clone().requires_grad_(True) for t in (w, b, gam, bet)]
        with T.precision(mode):
            y = T.linear(xs[0], ws[0], ws[1])                       # (M,128) Linear: half operands, rounded result
            y = T.ln_relu(y, ws[2], ws[3], True)
            y = T.mul_gather(y, xs[1], plan)                        # product with a gathered per-node row
            y = T.gate(y, xs[2])
            y = T.add(y, T.gather(T.scatter_sum(y, plan).to(y.dtype), plan))
            y.backward(cast(gout))
        return [y.detach().float()] + [t.grad.float() for t in xs + ws]

    half, full = run('fp16', True), run('fp16_f32store', False)
    names = ['out', 'dx', 'dtab', 'da2', 'dW', 'db', 'dgamma', 'dbeta']
    for n, p, q in zip(names, half, full):
        scale = float(q.abs().max())
        # identical arithmetic; the half container rounds the LayerNorm output and the residual sum one operator earlier
        assert float((p - q).abs().max()) <= 4e-3 * scale, (n, float((p - q).abs().max()), scale)
    assert half[0].shape == (M, 128)


@pytest.mark.parametrize('M,N,K,splits', [(64, 64, 64, 1), (1003, 64, 128, 5), (20011, 128, 64, 37), (50021, 256, 256, 192),
                                          (7001, 128, 256, 40), (333, 960, 256, 2)])
def test_float16_weight_gradient_by_lds_transpose_reads(M, N, K, splits):
    """dW = dY^T X on float16 containers (csrc hgemm_tn_tr_kernel: tiles copied to LDS as they are, MFMA operands fetched with the
    LDS transpose read ds_read_b64_tr_b16) against the fp64 product of the same float16 values: the kernel multiplies exactly and
    accumulates in fp32, so with the output rounding switched off (round_out = 0 through the C ABI) the error is fp32 summation
    error.  Asymmetric random operands (a transposed or permuted operand would show), ragged row counts (the last 64-row step is
    padded with zeros), the bias gradient from the same pass.  Through train_ops (autocast semantics: the gradient of a weight that
    autocast cast to half is itself rounded to half) the result is that product rounded to float16, and equals the fp32-container
    twin (the converting kernel) up to one rounding boundary."""
    from moldiff_amd import _lib
    g = U.rng(41)
    dy = torch.from_numpy(g.standard_normal((M, N)).astype(np.float32)).to(DEV).half()
    x = torch.from_numpy((g.standard_normal((M, K)) * np.linspace(0.5, 2.0, K)).astype(np.float32)).to(DEV).half()
    ref = dy.double().t() @ x.double()
    refb = dy.double().sum(0)
    tol = 2e-6 * float((dy.double().abs().t() @ x.double().abs()).max())
    # exact-product check: C ABI, float16 containers on both sides (dt = 3), float16 MFMA (half_kind = 2), no output rounding
    L = _lib.lib()
    part = torch.empty((splits + (splits + 255) // 256) * (N * K + N), dtype=torch.float32, device=DEV)
    dw = torch.empty(N, K, dtype=torch.float32, device=DEV)
    db = torch.empty(N, dtype=torch.float32, device=DEV)
    _lib.check(L.mdx_op_xgemm_tn_t(_lib.ptr(dy), dy.stride(0), _lib.ptr(x), x.stride(0), _lib.ptr(dw), K, _lib.ptr(db), M, N, K, splits,
                                   _lib.ptr(part), 2, 0, 3, _lib.stream()))
    assert float((dw.double() - ref).abs().max()) <= tol
    assert float((db.double() - refb).abs().max()) <= 2e-6 * float(dy.double().abs().sum(0).max())
    # operator level: rounded to float16 like autocast's weight gradient; twin = fp32 containers (converting kernel)
    with T.precision('fp16'):
        dwh, dbh = T.sgemm_tn(dy, x, splits, want_bias=True)
    with T.precision('fp16_f32store'):
        dw2, db2 = T.sgemm_tn(dy.float(), x.float(), splits, want_bias=True)
    half_ulp = ref.abs() * 2.0 ** -10 + 2.0 ** -24
    assert bool(((dwh.double() - ref).abs() <= half_ulp + tol).all())
    assert bool(((dwh.double() - dw2.double()).abs() <= half_ulp + tol).all())
    assert torch.equal(dbh, db2)                       # same values, same row order


@pytest.mark.parametrize('K,N', [(64, 64), (64, 256), (256, 256), (256, 64), (128, 128), (32, 64), (64, 32), (128, 256)])
def test_float16_linear_on_many_rows_is_the_row_owner_kernel_and_equals_the_tile_kernel(K, N):
    """Y = X W^T + b + addend on float16 rows: from 1,024 rows on (and K, N among the instantiated widths) the operator runs
    csrc hgemm_nt_rows_kernel -- whole weight resident in LDS, every wave streams its own 16-row tiles straight from HBM into MFMA
    operand registers -- below that hgemm_nt_kernel (LDS tiles, K loop).  Same products, same accumulation order: the rows of a big
    call must equal, bit for bit, the same rows computed by a small call; and both the fp64 result rounded to float16.  Ragged row
    count, float16 and fp32 addends, fp32 output container (keep32), and the grad_input form (W^T view, float16 result)."""
    g = U.rng(43)
    M = 5003
    h = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).to(DEV).half()
    x, w, b = h(M, K), (h(N, K) * 0.2).float(), h(N).float()
    lo, hi = 2000, 2900
    for addend, keep32 in ((None, False), (h(M, N), False), (h(M, N).float(), True)):
        with T.precision('fp16'):
            big = T.sgemm_nt(x, w, b, addend=addend, keep32=keep32)
            small = T.sgemm_nt(x[lo:hi], w, b, addend=None if addend is None else addend[lo:hi], keep32=keep32)
            tail = T.sgemm_nt(x[M - 7:], w, b, addend=None if addend is None else addend[M - 7:], keep32=keep32)
        assert big.dtype == (torch.float32 if keep32 else torch.float16)
        assert torch.equal(big[lo:hi], small) and torch.equal(big[M - 7:], tail)
        ref = x.double() @ w.half().double().t() + b.double() + (0 if addend is None else addend.double())
        err = (big.double() - ref).abs()
        bound = (ref.abs() * 2.0 ** -10 + 1e-3) if not keep32 else (2e-6 * (x.double().abs() @ w.half().double().abs().t() + 1))
        assert bool((err <= bound).all()), float((err - bound).max())
    gy = h(M, N)
    with T.precision('fp16'):
        gx = T.sgemm_nt(gy, w.t().contiguous(), out_dtype=torch.float16)          # grad_input: (M,N) @ (N,K)
        gxs = T.sgemm_nt(gy[lo:hi], w.t().contiguous(), out_dtype=torch.float16)
    assert gx.dtype == torch.float16 and torch.equal(gx[lo:hi], gxs)
    ref = gy.double() @ w.half().double()
    assert bool(((gx.double() - ref).abs() <= ref.abs() * 2.0 ** -10 + 1e-3).all())


@pytest.mark.parametrize('K,N', [(64, 256), (256, 256), (64, 32), (64, 64), (128, 128), (256, 64)])
@pytest.mark.parametrize('addend_kind', [None, 'half', 'fp32'])
def test_linear_with_layernorm_epilogue_equals_the_two_operators(K, N, addend_kind):
    """relu(LayerNorm(Linear(x) + addend)) in ONE launch (csrc hgemm_nt_rows_kernel<.., LN>, train_ops.linear_ln_relu) against the
    Linear launch followed by the LayerNorm launch: the Linear's stored result bit for bit (same products, same order), the row
    statistics to fp32 rounding (another summation order over the row), the activated rows to one float16 ulp on a handful of
    elements, and every gradient -- the backward is the two operators' own -- to the same."""
    g = U.rng(K + 3 * N)
    M = 5003                                                             # ragged: the last tile has 11 rows
    h = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).to(DEV).half()
    x0, w0, b0 = h(M, K), (h(N, K) * K ** -0.5).float(), h(N).float()
    g0, be0 = (1 + 0.3 * h(N)).float(), (0.2 * h(N)).float()
    a0 = None if addend_kind is None else (h(M, N) if addend_kind == 'half' else h(M, N).float())
    gy = h(M, N)
    res = []
    for fused in (True, False):
        x, w, b, gm, be = (t.detach().clone().requires_grad_(True) for t in (x0, w0, b0, g0, be0))
        a = None if a0 is None else a0.detach().clone().requires_grad_(True)
        with T.precision('fp16'):
            assert T.linear_ln_ok(x, w, a)
            if fused:
                y = T.linear_ln_relu(x, w, b, gm, be, addend=a)
                assert y.grad_fn.name().startswith('_LinearLnRelu')
                pre, stats = y.grad_fn.ln.saved_tensors[0], y.grad_fn.ln.saved_tensors[3]
            else:
                pre_t = T.linear(x, w, b, addend=a)
                y = T.ln_relu(pre_t, gm, be, True)
                pre, stats = pre_t.detach(), y.grad_fn.saved_tensors[3]
            y.backward(gy)
        res.append(dict(y=y.detach(), pre=pre, stats=stats, gx=x.grad, gw=w.grad, gb=b.grad, gg=gm.grad, gbe=be.grad,
                        ga=None if a is None else a.grad))
    f, u = res
    assert f['y'].dtype == torch.float16 and f['pre'].dtype == torch.float16
    assert torch.equal(f['pre'], u['pre'])
    assert _rel(f['stats'][:, 0], u['stats'][:, 0]) < 1e-5 and _rel(f['stats'][:, 1], u['stats'][:, 1]) < 1e-5
    dy = (f['y'].float() - u['y'].float()).abs()
    assert float(dy.max()) <= 2.0 ** -10 * float(u['y'].float().abs().max()) and float((dy > 0).float().mean()) < 1e-3
    for k in ('gx', 'gw', 'gb', 'gg', 'gbe', 'ga'):
        if f[k] is not None:
            assert f[k].dtype == u[k].dtype and _rel(f[k], u[k]) < 2e-3, k
    # and against float64 on the float16 operands
    ref = F.relu(F.layer_norm(x0.double() @ w0.half().double().t() + b0.double() + (0 if a0 is None else a0.double()), (N,), g0.double(), be0.double()))
    assert _rel(f['y'], ref) < 4e-3


def test_linear_with_layernorm_epilogue_falls_back_where_the_fused_kernel_is_not_built():
    """fp32 mode, few rows, odd widths: linear_ln_relu runs the two operators (same values as before the fused kernel existed)."""
    g = U.rng(77)
    x, w, b, gm, be = _leaf(g, 300, 80), _leaf(g, 48, 80, scale=0.1), _leaf(g, 48), _leaf(g, 48), _leaf(g, 48)
    assert not T.linear_ln_ok(x, w, None)
    y = T.linear_ln_relu(x, w, b, gm, be)
    ref = F.relu(F.layer_norm(F.linear(x.double(), w.double(), b.double()), (48,), gm.double(), be.double()))
    assert _rel(y, ref) < 5e-6
    with T.precision('fp16'):
        xs = x.detach().half()
        assert not T.linear_ln_ok(xs, w, None)           # 300 rows, K = 80
        y16 = T.linear_ln_relu(xs, w, b, gm, be)
    assert _rel(y16, ref) < 2e-2


def test_flipped_plan_equals_a_second_stable_sort():
    """Round 6: for the directed edge list [half-edges ; flipped half-edges] the right end points' CSR plan is derived from the left
    one's by one launch (csrc plan_flip_kernel) -- order and segment starts must equal what the stable sort of the right index gives,
    on a ragged list with isolated nodes and repeated pairs."""
    g = U.rng(77)
    n, Eh = 211, 3001
    hl = torch.from_numpy(g.integers(0, n - 7, Eh))      # (nodes n-7.. have no edge at all)
    hr = torch.from_numpy(g.integers(0, n - 7, Eh))
    left, right = torch.cat([hl, hr]).to(DEV), torch.cat([hr, hl]).to(DEV)
    pl = T.IndexPlan(left, n)
    ref = T.IndexPlan(right, n)
    got = T.FlippedPlan(pl, right, Eh)
    assert torch.equal(got.ptr, ref.ptr) and torch.equal(got.order, ref.order) and torch.equal(got.index, ref.index) and got.n == ref.n


@pytest.mark.parametrize('half', [False, True])
def test_fanout_sums_the_consumers_gradients_in_one_launch(half):
    """Round 6: T.fanout hands a tensor to k consumers as aliases of one node whose backward is a single k-input sum (csrc
    sum_n4_kernel: fp32 accumulation, one rounding) -- equal to autograd's pairwise accumulation up to the rounding of the container."""
    g = U.rng(5)
    x = _leaf(g, 515, 64)
    ws = [U.t32(g.standard_normal((515, 64))).to(DEV) for _ in range(7)]
    xin = x.half() if half else x
    if half:
        xin.retain_grad()
    parts = T.fanout(xin, 7)
    assert len(parts) == 7 and all(p.data_ptr() == xin.data_ptr() for p in parts)
    loss = sum((p.float() * w).sum() for p, w in zip(parts[:6], ws))      # the 7th alias is never used: its gradient is skipped
    loss.backward()
    ref = sum(ws[:6])
    tol = 2e-3 if half else 2e-6
    assert _rel(x.grad, ref) < tol
    assert T.fanout(x, 2) == (x, x)                                       # nothing to gain below three consumers
