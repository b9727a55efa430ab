"""Differentiable layer operators of the training path: ``torch.autograd.Function`` shells around the HIP forward /
backward kernels of ``csrc/mdx_train.hip`` (C-ABI ``mdx_op_*``).

torch is the graph engine and the allocator here (it records which operator produced which tensor and calls the
backward entry points in reverse order); every O(rows x features) computation -- Linear forward, data and weight
gradients (MFMA SGEMM), LayerNorm(+ReLU), gates, gathers / segmented sums, edge geometry -- runs in the library.  There
is no CPU fallback: CPU tensors raise.  Replaces, for training, ``torch.nn.functional.linear / layer_norm / relu``,
``torch_scatter.scatter_sum`` and index_select as used by the reference's models/graph.py and models/common.py.
"""
import torch

from . import _lib
from ._lib import ptr, stream, check

ADD, SUB, MUL, GATE = 0, 1, 2, 3

# ---- C++ fast path (round 6; csrc/mdx_fast.cpp -> moldiff_amd/_mdx_fast.so) ------------------------------------------------------------
# The operator bodies below cost the host ~20 ms per training step in Python (a Linear node 32 us, a fused BondFFN node 110 us) -- as
# long as the GPU needs for the step.  `_mdx_fast` holds the same logic in C++ against the same C ABI: the gradient sink and the
# weight-gradient queue as C++ state, Linear / Linear+LayerNorm+ReLU / element-wise nodes as C++ autograd functions, the fused operators'
# forward / backward bodies.  With it loaded, EVERY sink record and queue entry goes through the C++ state (any precision mode); the fast
# nodes themselves cover the float16 autocast mode with float16 containers inside a sink (Trainer.step, precision='fp16').  MDX_TRAIN_FAST=0
# keeps the Python bodies (the two are bit-identical: tests/test_gpu_trainer.py).  A missing extension is an error, not a silent slow path.
_FAST_ON = __import__('os').environ.get('MDX_TRAIN_FAST', '1') != '0'
_FASTMOD = None


def _fast():
    """the C++ extension, or None when MDX_TRAIN_FAST=0"""
    global _FASTMOD
    if not _FAST_ON:
        return None
    if _FASTMOD is None:
        _lib.lib()      # libmoldiff_hip.so first (the extension links it)
        try:
            from . import _mdx_fast
        except ImportError as e:
            raise RuntimeError('moldiff_amd/_mdx_fast.so (the C++ fast path of the training operators) is not built or does not load: run '
                               '`python -c "import __graft_entry__ as g; g.build()"` (make -C moldiff_amd/csrc), or set MDX_TRAIN_FAST=0 for '
                               f'the Python operator bodies.  [{e}]') from e
        _FASTMOD = _mdx_fast
        _FASTMOD.set_options(_WG_ON, WGRAD_ROWS, WGRAD_QUEUE_BYTES)
    return _FASTMOD


def _fast_sync_precision():
    if _FASTMOD is not None:
        a = _AMP
        _FASTMOD.set_precision(*((int(a[0]), int(a[1]), int(a[2])) if a is not None else (0, 0, 0)))


def _L():
    return _lib.lib()


def _c(t):
    _lib._need_gpu(t)
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# Mixed precision of the Linear layers: None = exact fp32, else (half_kind, autocast) with half_kind 1 = bfloat16, 2 = float16.
#   autocast = False ('bf16'): operands rounded, everything else fp32 (round 2's mode).
#   autocast = True  ('fp16', 'bf16_autocast'): the arithmetic of the reference's training loop, which wraps get_loss in
#     torch.autocast(dtype=float16) (scripts/train_drug3d.py:93): a Linear takes operands AND returns its result in the half type
#     (fp32 accumulation), so do its data / weight / bias gradients; products of two such results (gates, pairwise products) are
#     half too; LayerNorm, softmax, the posteriors, the loss and every segment sum stay fp32.  Tensors live in fp32 containers
#     holding half-representable values.
#   store = True ('fp16'): those half VALUES also live in float16 CONTAINERS (every Linear / LayerNorm / product result and its
#     gradient) -- the operators are memory-bound, so this halves what limits them; 'fp16_f32store' keeps fp32 containers (round 3's
#     first cut; the two differ only where the half container rounds a LayerNorm output one operator earlier -- the node residual stream
#     stays fp32 in both, as under the reference's autocast: train_graph.res_add).
_AMP = None
KINDS = {'f32': None, 'bf16': (1, False, False), 'fp16': (2, True, True), 'fp16_f32store': (2, True, False),
         'bf16_autocast': (1, True, False)}


class precision:
    """Context manager selecting the precision of `linear` (forward, data and weight gradients) and of the element-wise products:
    'f32' (default: exact fp32 MFMA), 'bf16' (GEMM operands rounded to bf16, fp32 accumulate and results), 'fp16' / 'bf16_autocast'
    (autocast-equivalent: see _AMP above)."""

    def __init__(self, kind):
        if isinstance(kind, str):
            if kind not in KINDS:
                raise ValueError(kind)
            kind = KINDS[kind]
        self.kind = kind

    def __enter__(self):
        global _AMP
        self.prev, _AMP = _AMP, self.kind
        _fast_sync_precision()
        return self

    def __exit__(self, *exc):
        global _AMP
        _AMP = self.prev
        _fast_sync_precision()


def _round_kind():
    return _AMP[0] if (_AMP is not None and _AMP[1]) else 0


def _store_dtype():
    """container of a Linear / LayerNorm / product result in the current mode"""
    return torch.float16 if (_AMP is not None and _AMP[2]) else torch.float32


def _t(t):
    """device tensor, contiguous, in its own container type (fp32 or float16); anything else is converted to fp32"""
    _lib._need_gpu(t)
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float16):
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _h(t):
    return 1 if (t is not None and t.dtype == torch.float16) else 0


def _same(a, b):
    """two operands of an element-wise operator in one container type (mixed -> fp32; any length: the library falls back to its scalar
    kernels when the 4-wide float16 ones do not apply)"""
    a, b = _t(a), _t(b)
    if a.dtype != b.dtype:
        a, b = a.float(), b.float()
    return a, b


def _t4(t):
    """like _t, but a float16 tensor whose rows are not a multiple of 4 wide goes through fp32 (the half kernels are 4-wide)"""
    t = _t(t)
    return t.float() if (t.dtype == torch.float16 and t.shape[-1] % 4) else t


# ---- deferred parameter gradients (round 3) ----------------------------------------------------------------------------------
# While a `grad_sink` is active (Trainer.step), a Linear / LayerNorm whose parameters live in the sink's flat parameter buffer does
# not hand its weight / bias / affine gradients to autograd: it leaves the split partials of its weight-gradient GEMM (LayerNorm: its
# per-wave partial rows) where they are, records {partials, destination slot in the flat GRADIENT buffer, sizes} and returns None.
# `flush_grad_sink()` -- called once after backward -- sums every record in ONE launch (`mdx_op_reduce_deferred`, same fixed order as
# the per-layer reduction kernels) straight into the flat gradient buffer.  That replaces ~660 reduction launches and ~540 of torch's
# accumulation / slice-gradient kernels per step.  Without a sink everything goes through autograd as before.
_SINK = None


class grad_sink:
    def __init__(self, flat):
        self.flat = flat

    def __enter__(self):
        global _SINK
        f = self.flat
        self.prev = _SINK
        _SINK = {'data': f.data.data_ptr(), 'grad': f.grad.data_ptr(), 'nbytes': f.data.numel() * 4, 'recs': [], 'recs2': [], 'keep': [], 'blocks': 0, 'blocks2': 0,
                 'device': f.data.device, 'seen': set(), 'wq': [], 'wq_bytes': 0, 'fast': None}
        F = _fast() if (f.data.is_cuda and self.prev is None) else None     # (a nested sink keeps the Python bookkeeping)
        if F is not None and not F.sink_active():
            _fast_sync_precision()
            F.set_options(_WG_ON, WGRAD_ROWS, WGRAD_QUEUE_BYTES)
            F.sink_begin(f.data, f.grad)
            _SINK['fast'] = F
        return self

    def __exit__(self, *exc):
        global _SINK
        if _SINK is not None and _SINK['fast'] is not None:
            _SINK['fast'].sink_end()          # (flushes what is left)
        elif _SINK is not None and (_SINK['recs'] or _SINK['wq']):
            flush_grad_sink()
        _SINK = self.prev


def _sink_dst(t):
    """address of `t`'s slot in the flat gradient buffer when t (a parameter or a view of one) lives in the sink's parameter buffer"""
    if _SINK is None or t is None:
        return None
    off = t.data_ptr() - _SINK['data']
    return _SINK['grad'] + off if 0 <= off < _SINK['nbytes'] else None


def _sink_record(part_ptr, dst, S, rows, cols, ld, pstride, rkind, keep):
    sk = _SINK
    if sk['fast'] is not None:
        sk['fast'].sink_record(part_ptr, dst, S, rows, cols, ld, pstride, rkind, keep)
        return
    # The reduction launch sums all records of a flush with plain (non-atomic) read-modify-writes, one half wave per 128-element
    # block of a record: two records for the SAME slot (a layer applied twice, tied weights, a second backward() inside one sink)
    # would race and lose a contribution.  A repeated destination therefore flushes what has been recorded first -- launches are
    # stream-ordered, so the second gradient is added on top of the first like autograd's AccumulateGrad would.
    if dst in sk['seen']:
        flush_grad_sink()
    sk['seen'].add(dst)
    if S > _RED_CHUNK:
        # more partials than one chunk (a LayerNorm over the E edge rows leaves E / 64 partial rows): the two fixed-order stages of the
        # per-layer reduction (csrc launch_reduce_partials) -- a `chunked` record (bit 6) makes the first launch store the sum of each
        # chunk of 256 in the scratch planes the workspace / wgrad layouts keep right behind the partials; their sum is a second launch
        nc = (S + _RED_CHUNK - 1) // _RED_CHUNK
        _sink_add(sk['recs'], 'blocks', part_ptr, dst, S, rows, cols, ld, pstride, 64, nc)
        _sink_add(sk['recs2'], 'blocks2', part_ptr + 4 * S * pstride, dst, nc, rows, cols, ld, pstride, rkind)
    else:
        _sink_add(sk['recs'], 'blocks', part_ptr, dst, S, rows, cols, ld, pstride, rkind)
    sk['keep'].append(keep)


_RED_CHUNK = 256     # = RED_CHUNK of csrc/mdx_train.hip


def _sink_add(recs, counter, P, dst, S, rows, cols, ld, pstride, rkind, nchunks=1):
    sk = _SINK
    recs.append((P, dst, S, rows, cols, ld, pstride, rkind | (sk[counter] << 8)))
    sk[counter] += nchunks * ((rows * cols + 127) // 128)


def flush_grad_sink():
    """every recorded gradient summed into its slot of the flat gradient buffer: one launch, plus one over the chunk sums of the
    records that had more than 256 partials"""
    sk = _SINK
    if sk is None:
        return
    if sk['fast'] is not None:
        sk['fast'].flush()
        return
    _flush_wgrads()
    if not sk['recs']:
        return
    for recs, counter in ((sk['recs'], 'blocks'), (sk['recs2'], 'blocks2')):
        if recs:
            desc = torch.tensor(recs, dtype=torch.int64).to(sk['device'], non_blocking=True)
            check(_L().mdx_op_reduce_deferred(ptr(desc), len(recs), sk[counter], stream()))
            sk['keep'].append(desc)
    # the partial buffers may be reused once the launches above have been enqueued (stream order); drop the references
    sk['recs'], sk['recs2'], sk['keep'], sk['blocks'], sk['blocks2'] = [], [], [], 0, 0
    sk['seen'] = set()


def sgemm_nt(a, b, bias=None, splits=1, addend=None, keep32=False, out_dtype=None):
    """a (M,K) @ b (N,K)^T + bias + addend -> (M,N).  Rows of a / b may be strided (column slices of a wider matrix).
    keep32: in an autocast mode, do not round the result (it is a partial sum that enters another Linear as its addend).
    a / addend may be fp32 or float16 containers (mixed precision only); out_dtype: container of the result (default: the mode's)."""
    M, K = a.shape
    N = b.shape[0]
    assert a.stride(1) == 1 and b.stride(1) == 1 and b.dtype == torch.float32
    if _AMP is not None and splits <= 1:
        if out_dtype is None:
            out_dtype = torch.float32 if keep32 else _store_dtype()
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
        rnd = int((_AMP[1] and not keep32) or out_dtype == torch.float16)
        check(_L().mdx_op_xgemm_nt_t(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), ptr(addend),
                                     addend.stride(0) if addend is not None else 0, ptr(out), N, M, N, K, _AMP[0], rnd,
                                     _h(a) | (_h(addend) << 1) | (_h(out) << 2), stream()))
        return out
    assert a.dtype == torch.float32 and (addend is None or addend.dtype == torch.float32)
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    part = torch.empty(splits * M * N, dtype=torch.float32, device=a.device) if splits > 1 else None
    check(_L().mdx_op_sgemm_nt(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), ptr(addend), N if addend is not None else 0, ptr(out),
                               N, M, N, K, splits, ptr(part), stream()))
    return out


ROWS_MIN = 16384   # from this many rows on a Linear runs on the row-owner kernel (one wave per 16 rows; the weight is re-packed per call)


def linear_rows_ok(a, n, k, addend=None):
    """Can `a` (M,k) @ W -> (M,n) take the row-owner kernel?  Shape built, enough rows, 16-byte aligned rows, fp32 mode."""
    return (_AMP is None and a.dtype == torch.float32 and a.shape[0] >= ROWS_MIN and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
            and (addend is None or (addend.stride(1) == 1 and addend.stride(0) % 4 == 0 and addend.data_ptr() % 16 == 0))
            and _L().mdx_op_linear_rows_supported(n, k) == 1)


def linear_rows(a, w, trans_w, bias=None, addend=None):
    """a (M,K) @ W^T (W (N,K), trans_w = False) or a @ W (W (K,N), trans_w = True) + bias + addend -> (M,N) on the row-owner
    kernel (csrc/mdx_linear_rows.hip)."""
    M, K = a.shape
    N = w.shape[1] if trans_w else w.shape[0]
    assert w.stride(1) == 1 and (w.shape[0] if trans_w else w.shape[1]) == K
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ws = torch.empty(_L().mdx_op_linear_rows_ws(N, K) // 4, dtype=torch.float32, device=a.device)
    check(_L().mdx_op_linear_rows(ptr(a), a.stride(0), ptr(w), w.stride(0), 1 if trans_w else 0, ptr(bias), ptr(addend),
                                  addend.stride(0) if addend is not None else 0, ptr(out), N, M, N, K, ptr(ws), stream()))
    return out


_LAYOUTS = {}


def _wgrad_layout(M, N, K, splits, half):
    """(row ranges actually used, offset of the bias partials) of a weight-gradient call: a pure function of its arguments, asked of the
    library once per shape (it used to be a call per weight gradient per step)"""
    key = (M, N, K, splits, half)
    r = _LAYOUTS.get(key)
    if r is None:
        import ctypes
        S, boff = ctypes.c_int64(), ctypes.c_int64()
        check(_L().mdx_op_wgrad_layout(M, N, K, splits, half, ctypes.byref(S), ctypes.byref(boff)))
        r = _LAYOUTS[key] = (S.value, boff.value)
    return r


# ---- queued weight gradients (round 6) -----------------------------------------------------------------------------------------
# Inside a gradient sink, in the float16 autocast mode, a deferred weight gradient is not launched where autograd reaches it: the
# contraction is QUEUED (operands kept alive) and `_flush_wgrads` -- at the end of backward, or when the queue holds WGRAD_QUEUE_BYTES
# of operands -- runs the whole queue as one launch per tile class (csrc wgrad_grouped_kernel) with the partials in ONE buffer; the
# sink records are made then.  ~260 launches (and as many partial-buffer allocations) per step become <= 5; the row ranges are chosen
# for the queue as a whole (WGRAD_ROWS rows per block; a stand-alone launch needs ~512 blocks of its own to fill the chip, i.e. up to
# 128 row ranges = 33 MB of partials for one 256 x 256 weight).  MDX_WGRAD_GROUPED=0 keeps the per-call launches.
_WG_ON = __import__('os').environ.get('MDX_WGRAD_GROUPED', '1') != '0'
WGRAD_ROWS = int(__import__('os').environ.get('MDX_WGRAD_ROWS', '2048'))
WGRAD_QUEUE_BYTES = int(float(__import__('os').environ.get('MDX_WGRAD_QUEUE_GB', '24')) * 2 ** 30)
_PLANS = {}


def _wgrad_plan(M, N, K, dt, ldg, ldx, aligned):
    """(kind, gx, gy, S, mper, bias offset, partial floats, blocks) of a queued weight gradient (csrc mdx_op_wgrad_plan), cached per shape"""
    key = (M, N, K, dt, ldg, ldx, aligned, WGRAD_ROWS)
    r = _PLANS.get(key)
    if r is None:
        import ctypes
        out = (ctypes.c_int64 * 8)()
        check(_L().mdx_op_wgrad_plan(M, N, K, max(1, (M + WGRAD_ROWS - 1) // WGRAD_ROWS), dt, ldg, ldx, aligned, out))
        r = _PLANS[key] = tuple(out)
    return r


def _flush_wgrads():
    """run the queued weight gradients (one launch per tile class) and hand their partials to the gradient sink"""
    sk = _SINK
    if sk is None or not sk['wq']:
        return
    jobs, sk['wq'], sk['wq_bytes'] = sk['wq'], [], 0
    total = sum((j[2][6] + 3) // 4 * 4 for j in jobs)
    part = torch.empty(total, dtype=torch.float32, device=sk['device'])
    base, off = part.data_ptr(), 0
    by_kind, placed = {}, []
    for g, x, plan, dims, dst_w, ldw, dst_b, rk in jobs:
        kind, gx, gy, S, mper, boff, psize, blocks = plan
        M, N, K, dt = dims
        P = base + 4 * off
        by_kind.setdefault(kind, []).append([g.data_ptr(), x.data_ptr(), P, (P + 4 * boff) if dst_b is not None else 0, g.stride(0), x.stride(0),
                                             M, N, K, mper, gx, gy, S, dt, 0, blocks])
        placed.append((P, S, N, K, boff, dst_w, ldw, dst_b, rk))
        off += (psize + 3) // 4 * 4
    rows, launches = [], []
    for kind in sorted(by_kind):
        recs = by_kind[kind]
        recs.sort(key=lambda r: -r[9])       # the long blocks first (no tail of E-row blocks behind the N-row jobs)
        fb = 0
        for r in recs:
            r[14], fb = fb, fb + r[15]
        launches.append((kind, len(rows), len(recs), fb))
        rows += recs
    desc = torch.tensor(rows, dtype=torch.int64).to(sk['device'], non_blocking=True)
    st = stream()
    for kind, start, n, tb in launches:
        check(_L().mdx_op_wgrad_grouped(desc.data_ptr() + 128 * start, n, tb, kind, st))
    sk['keep'] += [part, desc]
    for P, S, N, K, boff, dst_w, ldw, dst_b, rk in placed:
        _sink_record(P, dst_w, S, N, K, ldw, N * K, rk, None)
        if dst_b is not None:
            _sink_record(P + 4 * boff, dst_b, S, 1, N, N, N, rk, None)
    sk['keep'] += [part, desc]       # (again: a repeated destination above flushes the sink, which drops its references)


def sgemm_tn(g, x, splits, want_bias=False, defer=None):
    """g (M,N)^T @ x (M,K) -> (N,K): the weight gradient, straight from the row-major tensors; with want_bias also the
    column sums of g (the bias gradient) from the same pass -> (dW, db).
    defer = (dst_w, ldw, dst_b): leave the split partials for `flush_grad_sink` (destinations in the flat gradient buffer) -> None."""
    M, N = g.shape
    K = x.shape[1]
    splits = max(1, int(splits))
    if defer is not None and _WG_ON and _SINK is not None and _AMP is not None and _AMP[0] == 2 and g.stride(1) == 1 and x.stride(1) == 1:
        dst_w, ldw, dst_b = defer
        if _SINK['fast'] is not None:
            _SINK['fast'].wq_append(g, x, dst_w, ldw, (dst_b or 0) if want_bias else 0, _AMP[0] if _AMP[1] else 0)
            return None
        dt = _h(g) | (_h(x) << 1)
        plan = _wgrad_plan(M, N, K, dt, g.stride(0), x.stride(0), int(g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0))
        sk = _SINK
        if sk['fast'] is not None:
            sk['fast'].wq_append(g, x, dst_w, ldw, (dst_b or 0) if want_bias else 0, _AMP[0] if _AMP[1] else 0)
            return None
        sk['wq'].append((g, x, plan, (M, N, K, dt), dst_w, ldw, dst_b if want_bias else None, _AMP[0] if _AMP[1] else 0))
        sk['wq_bytes'] += g.numel() * g.element_size() + x.numel() * x.element_size()
        if sk['wq_bytes'] > WGRAD_QUEUE_BYTES:
            _flush_wgrads()
        return None
    part = torch.empty((splits + (splits + 255) // 256) * (N * K + N), dtype=torch.float32, device=g.device)
    if defer is not None:
        dst_w, ldw, dst_b = defer
        S, boff = _wgrad_layout(M, N, K, splits, 1 if _AMP is not None else 0)
        wb = ptr(part) if want_bias else None        # any non-null pointer = "also write the bias partials"
        if _AMP is not None:
            check(_L().mdx_op_xgemm_tn_t(ptr(g), g.stride(0), ptr(x), x.stride(0), None, K, wb, M, N, K, splits, ptr(part), _AMP[0],
                                         int(_AMP[1]), _h(g) | (_h(x) << 1), stream()))
            rk = _AMP[0] if _AMP[1] else 0
        else:
            check(_L().mdx_op_sgemm_tn(ptr(g), g.stride(0), ptr(x), x.stride(0), None, K, wb, M, N, K, splits, ptr(part), stream()))
            rk = 0
        _sink_record(part.data_ptr(), dst_w, S, N, K, ldw, N * K, rk, part)
        if want_bias:
            _sink_record(part.data_ptr() + 4 * boff, dst_b, S, 1, N, N, N, rk, part)
        return None
    out = torch.empty(N, K, dtype=torch.float32, device=g.device)
    db = torch.empty(N, dtype=torch.float32, device=g.device) if want_bias else None
    if _AMP is not None:
        check(_L().mdx_op_xgemm_tn_t(ptr(g), g.stride(0), ptr(x), x.stride(0), ptr(out), K, ptr(db), M, N, K, splits, ptr(part), _AMP[0],
                                     int(_AMP[1]), _h(g) | (_h(x) << 1), stream()))
    else:
        assert g.dtype == torch.float32 and x.dtype == torch.float32
        check(_L().mdx_op_sgemm_tn(ptr(g), g.stride(0), ptr(x), x.stride(0), ptr(out), K, ptr(db), M, N, K, splits, ptr(part), stream()))
    return (out, db) if want_bias else out


# Transposed weights of the step (Trainer.refresh_transposed): the flat parameter buffer's 2-D tensors, each transposed at its own
# offset of a second flat buffer by ONE launch per step.  A weight W (N,K) or a column slice W[:, a:b] of it (the hoisted layers
# split their weights by columns, train_graph.py) then has its transpose as a VIEW: rows a..b of W^T (K,N).
_WT = None


class TransposedParams:
    def __init__(self, flat):
        self.base, self.nbytes = flat.data.data_ptr(), 4 * flat.numel
        # every W^T starts on 16 bytes in its own buffer (round 6: with the parameter buffer's offsets a third of the weights' transposes
        # were unaligned and fell back to a transpose launch per data gradient)
        self.buf = torch.empty(flat.numel + 4 * len(flat.params) + 4, dtype=torch.float32, device=flat.data.device)
        pad = (-self.buf.data_ptr() // 4) % 4
        self.entries, desc, off, doff, tiles = {}, [], 0, pad, 1
        for p in flat.params:
            n = p.numel()
            if p.dim() == 2 and p.shape[0] % 4 == 0:     # W^T rows are R = shape[0] floats: 16-byte aligned rows for the GEMM loads
                R, C = p.shape
                self.entries[4 * off] = (off, R, C, doff)
                desc += [self.base + 4 * off, self.buf.data_ptr() + 4 * doff, R, C]
                tiles = max(tiles, ((R + 31) // 32) * ((C + 31) // 32))
                doff += (n + 3) // 4 * 4
            off += n
        self.offsets = sorted(self.entries)
        self.n, self.tiles = len(desc) // 4, tiles
        self.desc = torch.tensor(desc, dtype=torch.int64, device=flat.data.device)

    def refresh(self):
        check(_L().mdx_op_transpose_batch(ptr(self.desc), self.n, self.tiles, stream()))

    def view(self, w):
        """W^T of a parameter matrix or of a column slice of one, as a view into the transposed buffer; None if `w` is neither."""
        import bisect
        rel = w.data_ptr() - self.base
        if rel < 0 or rel >= self.nbytes or w.dim() != 2 or w.dtype != torch.float32 or w.stride(1) != 1:
            return None
        i = bisect.bisect_right(self.offsets, rel) - 1
        if i < 0:
            return None
        off, R, C, doff = self.entries[self.offsets[i]]
        col = rel // 4 - off
        if col < 0 or col >= C or w.stride(0) != C or w.shape[0] != R or col + w.shape[1] > C:
            return None
        if (self.buf.data_ptr() + 4 * (doff + col * R)) % 16:
            return None                                 # the GEMM's 16-byte operand loads want aligned rows
        return self.buf[doff:doff + R * C].view(C, R)[col:col + w.shape[1]]


class transposed_params:
    """context: the dgrad GEMMs inside take W^T from `tp` (a TransposedParams refreshed for the current weights)"""

    def __init__(self, tp):
        self.tp = tp

    def __enter__(self):
        global _WT
        self.prev, _WT = _WT, self.tp
        _fast_sync_wt()

    def __exit__(self, *a):
        global _WT
        _WT = self.prev
        _fast_sync_wt()


def _fast_sync_wt():
    F = _fast() if (_WT is None or _WT.buf.is_cuda) else None
    if F is not None:
        tp = _WT
        if tp is None:
            F.set_wt(0, 0, 0, [], [])
        else:
            F.set_wt(tp.base, tp.nbytes, tp.buf.data_ptr(), tp.offsets, [list(tp.entries[o]) for o in tp.offsets])   # (off, R, C, dst off)


def transpose(x, pad=4):
    """(R,C) -> contiguous-rows (C,R) view whose leading dimension is padded to a multiple of `pad` floats (so that the
    SGEMM's 16-byte loads stay aligned for any R)."""
    if _WT is not None:
        v = _WT.view(x)
        if v is not None:
            return v
    R, C = x.shape
    ld = (R + pad - 1) // pad * pad
    buf = torch.empty(C, ld, dtype=torch.float32, device=x.device)
    check(_L().mdx_op_transpose(ptr(x), x.stride(0), R, C, ptr(buf), ld, stream()))
    return buf[:, :R]


def colreduce(x, y=None):
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = torch.empty(((M + 511) // 512) * N, dtype=torch.float32, device=x.device) if M > 512 else None
    check(_L().mdx_op_colreduce(ptr(x), ptr(y), x.stride(0), M, N, ptr(out), ptr(ws), stream()))
    return out


_ENV = None


def _env():
    """constants of the weight-gradient split (round 3's A/B environment knobs are gone: transpose-read kernel on, 512 workgroups
    targeted, 64-wide tiles for the conversion kernel)"""
    global _ENV
    if _ENV is None:
        _ENV = (True, 512, False)
    return _ENV


def _splits_for(rows, n, k, half=False):
    """row ranges of a weight gradient (the contraction runs over `rows`): enough of them to fill the chip with workgroups, each >= 128
    rows.  fp32 / converting kernels: 64 x 64 tiles, ~1024 workgroups (flat between 512 and 4096; each loops over its rows in short
    steps, many short loops hide the load latency better than few long ones).  float16 containers with tile-aligned widths take the
    transpose-read kernel (csrc hgemm_tn_tr_kernel): 128-wide tiles where the layer allows, ~512 workgroups = one resident set
    (two per CU); more ranges mean more split partials to write and reduce (measured per step: 384 -> 34.2, 512 -> 33.5, 768 -> 34.1, 1024 -> 34.8 ms)."""
    use_tr, target, wide = _env()
    if half and use_tr and n % 64 == 0 and k % 64 == 0:
        tiles = (n // (128 if n % 128 == 0 else 64)) * (k // (128 if k % 128 == 0 else 64))
        return max(1, min(rows // 128, (target + tiles - 1) // tiles))
    tn = 128 if (wide and _AMP is not None and n >= 128) else 64
    tk = 128 if (wide and _AMP is not None and k >= 128) else 64
    tiles = ((n + tn - 1) // tn) * ((k + tk - 1) // tk)
    return max(1, min(rows // 128, (1024 + tiles - 1) // tiles))


def _rows(t):
    """device matrix (fp32 or float16 container) whose rows are contiguous (row stride arbitrary: column slices of a weight stay views)."""
    _lib._need_gpu(t)
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float16):
        t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, addend, keep32):
        xc, wc = _rows(x), _rows(w)
        if _AMP is None and xc.dtype != torch.float32:
            xc = xc.float()
        ctx.save_for_backward(xc, wc)
        ctx.has_bias, ctx.has_addend = b is not None, addend is not None
        ctx.b_ref = b.detach() if b is not None else None      # (a view of the parameter: only its address is used, see grad_sink)
        ctx.x_dtype = x.dtype
        ctx.addend_dtype = addend.dtype if addend is not None else None
        ctx.prec = _AMP     # the backward of this layer runs in the forward's precision
        bc = _c(b) if b is not None else None
        ac = None
        if addend is not None:
            ac = _rows(addend) if _AMP is not None else _c(addend)
        if linear_rows_ok(xc, wc.shape[0], wc.shape[1], ac):
            return linear_rows(xc, wc, False, bc, ac)
        return sgemm_nt(xc, wc, bc, addend=ac, keep32=keep32)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _t(gy) if ctx.prec is not None else _c(gy)
        gx = gw = gb = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        with precision(ctx.prec):
            if ctx.needs_input_grad[0]:
                if linear_rows_ok(gy, w.shape[1], w.shape[0]):
                    gx = linear_rows(gy, w, True)                     # (M,N) @ (N,K), the weight read transposed by the pack
                else:                                                 # (M,N) @ (K,N)^T; the gradient lives in x's container type
                    gx = sgemm_nt(gy, transpose(w), out_dtype=x.dtype if ctx.prec is not None else None)
                if gx.dtype != ctx.x_dtype:
                    gx = gx.to(ctx.x_dtype)
            if ctx.needs_input_grad[1]:
                dst_w = _sink_dst(w)
                dst_b = _sink_dst(ctx.b_ref) if want_b else None
                if dst_w is not None and (not want_b or dst_b is not None):
                    sgemm_tn(gy, x, _splits_for(x.shape[0], gy.shape[1], x.shape[1], _h(gy) and _h(x) and _AMP is not None and _AMP[0] == 2), want_bias=want_b,
                             defer=(dst_w, w.stride(0), dst_b))              # summed into the flat gradient buffer at flush time
                else:
                    r = sgemm_tn(gy, x, _splits_for(x.shape[0], gy.shape[1], x.shape[1], _h(gy) and _h(x) and _AMP is not None and _AMP[0] == 2), want_bias=want_b)
                    gw, gb = r if want_b else (r, None)
            elif want_b:
                gb = colreduce(gy.float() if gy.dtype != torch.float32 else gy)
        ga = None
        if ctx.has_addend and ctx.needs_input_grad[3]:
            ga = gy if gy.dtype == ctx.addend_dtype else gy.to(ctx.addend_dtype)
        return gx, gw, gb, ga, None


def linear(x, w, b=None, addend=None, keep32=False):
    """y = x @ w.T + b + addend for 2-D x (rows, in_features); `addend` (rows, out_features) is added in the GEMM epilogue
    (used for the per-node part of a layer whose reference input is a concatenation [edge part | node part | time]).
    keep32: the result is itself such a partial sum -- an autocast mode must not round it (the reference rounds the WHOLE layer's
    result once)."""
    sk = _SINK
    if sk is not None and sk['fast'] is not None and sk['fast'].linear_fast_ok(x, w, b):
        return sk['fast'].linear(x, w, b, addend, bool(keep32))
    return _Linear.apply(x, w, b, addend, keep32)


class _LnRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, relu):
        xc, g, b = _t(x), _c(gamma), _c(beta)
        M, F = xc.shape
        ctx.x_dtype = x.dtype
        if F not in (32, 64, 128, 256):     # the half kernels cover the widths the networks use
            xc = xc.float()
        # the only consumers of a LayerNorm result are Linears, which round their operand to half anyway: a half container is exact
        y = torch.empty(M, F, dtype=(_store_dtype() if (_AMP is not None and F in (32, 64, 128, 256)) else torch.float32), device=xc.device)
        stats = torch.empty(M, 2, dtype=torch.float32, device=xc.device)
        check(_L().mdx_op_ln_relu_fwd_t(ptr(xc), ptr(g), ptr(b), M, F, int(relu), ptr(y), ptr(stats), _h(xc) | (_h(y) << 1), stream()))
        ctx.save_for_backward(xc, g, b, stats)
        ctx.relu = int(relu)
        ctx.gb_ref = (gamma.detach(), beta.detach())
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g, b, stats = ctx.saved_tensors
        gy = _t(gy)
        M, F = x.shape
        if F not in (32, 64, 128, 256):
            gy = gy.float()
        dx = torch.empty_like(x)
        ws = torch.empty(_L().mdx_op_ln_relu_bwd_ws(M, F) // 4 + 1, dtype=torch.float32, device=x.device)
        dst_g, dst_b = _sink_dst(ctx.gb_ref[0]), _sink_dst(ctx.gb_ref[1])
        defer = dst_g is not None and dst_b is not None and M > 0 and ws.data_ptr() % 16 == 0
        dgb = None if defer else torch.empty(2 * F, dtype=torch.float32, device=x.device)
        check(_L().mdx_op_ln_relu_bwd_t(ptr(gy), ptr(x), ptr(stats), ptr(g), ptr(b), M, F, ctx.relu, ptr(dx), ptr(dgb), ptr(ws),
                                        _h(gy) | (_h(x) << 1) | (_h(dx) << 2), stream()))
        dx = dx if dx.dtype == ctx.x_dtype else dx.to(ctx.x_dtype)
        if defer:       # [dgamma | dbeta] partial rows stay in ws: summed into the flat gradient buffer at flush time
            rows = int(_L().mdx_op_ln_relu_bwd_rows(M))
            _sink_record(ws.data_ptr(), dst_g, rows, 1, F, F, 2 * F, 0, ws)
            _sink_record(ws.data_ptr() + 4 * F, dst_b, rows, 1, F, F, 2 * F, 0, ws)
            return dx, None, None, None
        return dx, dgb[:F], dgb[F:], None


def ln_relu(x, gamma, beta, relu=True):
    return _LnRelu.apply(x, gamma, beta, relu)


class _StageCtx:
    """What _Linear / _LnRelu use of an autograd context, so that their backward bodies can run as the two halves of _LinearLnRelu's."""

    def save_for_backward(self, *ts):
        self.saved_tensors = ts


_LN_OK = {}


def linear_ln_ok(x, w, addend):
    """Can Linear(x) -> LayerNorm -> ReLU take the fused launch (csrc mdx_op_xgemm_nt_ln_t)?  float16 mode, float16 rows of x, widths
    the row-owner kernel is built for."""
    if _AMP is None or _AMP[0] != 2 or x.dtype != torch.float16 or x.dim() != 2 or _store_dtype() != torch.float16:
        return False
    M, K = x.shape
    N = w.shape[0]
    key = (M >= 1024, N, K)
    r = _LN_OK.get(key)
    if r is None:
        r = _LN_OK[key] = _L().mdx_op_xgemm_nt_ln_supported(M, N, K) == 1
    return (r and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and w.stride(1) == 1
            and (addend is None or (addend.dim() == 2 and addend.stride(1) == 1)))


class _LinearLnRelu(torch.autograd.Function):
    """relu(LayerNorm(Linear(x) + addend)) as one launch forward (the LayerNorm runs in the GEMM's epilogue, csrc hgemm_nt_rows_kernel
    <.., LN>): the first two layers of every common.MLP (reference models/common.py:191-196).  The Linear's result and the row
    statistics are stored for the backward, which is the LayerNorm's backward followed by the Linear's -- the bodies of _LnRelu and
    _Linear, unchanged."""

    @staticmethod
    def forward(ctx, x, w, b, addend, gamma, beta):
        xc, wc = _rows(x), _rows(w)
        M, K = xc.shape
        N = wc.shape[0]
        bc = _c(b) if b is not None else None
        ac = _rows(addend) if addend is not None else None
        g, bt = _c(gamma), _c(beta)
        pre = torch.empty(M, N, dtype=torch.float16, device=xc.device)
        post = torch.empty(M, N, dtype=torch.float16, device=xc.device)
        stats = torch.empty(M, 2, dtype=torch.float32, device=xc.device)
        check(_L().mdx_op_xgemm_nt_ln_t(ptr(xc), xc.stride(0), ptr(wc), wc.stride(0), ptr(bc), ptr(ac), ac.stride(0) if ac is not None else 0,
                                        ptr(pre), N, ptr(g), ptr(bt), ptr(post), N, ptr(stats), 1, M, N, K, _AMP[0], 1,
                                        1 | (_h(ac) << 1) | 4 | 8, stream()))
        lin, ln = _StageCtx(), _StageCtx()
        lin.save_for_backward(xc, wc)
        lin.has_bias, lin.has_addend = b is not None, addend is not None
        lin.b_ref = b.detach() if b is not None else None
        lin.x_dtype, lin.addend_dtype, lin.prec = x.dtype, (addend.dtype if addend is not None else None), _AMP
        ln.save_for_backward(pre, g, bt, stats)
        ln.relu, ln.x_dtype, ln.gb_ref = 1, torch.float16, (gamma.detach(), beta.detach())
        ctx.lin, ctx.ln = lin, ln
        return post

    @staticmethod
    def backward(ctx, gy):
        need = ctx.needs_input_grad                       # x, w, b, addend, gamma, beta
        lin, ln = ctx.lin, ctx.ln
        ln.needs_input_grad = (True, need[4], need[5], False)
        gpre, gg, gb_, _ = _LnRelu.backward(ln, gy)
        lin.needs_input_grad = (need[0], need[1], need[2], need[3], False)
        gx, gw, gb, ga, _ = _Linear.backward(lin, gpre)
        ctx.lin = ctx.ln = None
        return gx, gw, gb, ga, gg, gb_


def linear_ln_relu(x, w, b, gamma, beta, addend=None):
    """relu(LayerNorm(x @ w.T + b + addend)): one launch where the fused kernel is built, the two operators otherwise"""
    if linear_ln_ok(x, w, addend):
        sk = _SINK
        if sk is not None and sk['fast'] is not None and sk['fast'].linear_ln_fast_ok(x, w, b, gamma, beta):
            return sk['fast'].linear_ln_relu(x, w, b, addend, gamma, beta)
        return _LinearLnRelu.apply(x, w, b, addend, gamma, beta)
    return ln_relu(linear(x, w, b, addend=addend), gamma, beta, True)


def linear_ln_relu_dot(x, w, b, gamma, beta, w2, b2):
    """linear(relu(LayerNorm(x @ w.T + b)), w2, b2) with w2 of ONE row -- common.MLP(…, 1) as PosUpdate's inter module uses it.  On the
    C++ fast path (float16 mode inside a sink) one node whose backward forms the second Linear's rank-1 data gradient inside the LayerNorm
    backward (csrc mdx_op_ln_relu_bwd_r1_t); otherwise the two operators."""
    sk = _SINK
    if (sk is not None and sk['fast'] is not None and b is not None and b2 is not None and linear_ln_ok(x, w, None)
            and sk['fast'].linear_ln_dot_fast_ok(x, w, b, gamma, beta, w2, b2)):
        return sk['fast'].linear_ln_relu_dot(x, w, b, gamma, beta, w2, b2)
    return linear(linear_ln_relu(x, w, b, gamma, beta), w2, b2)


class _Ew(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, a, b):
        ctx.dtypes = (a.dtype, b.dtype)
        ac, bc = _same(a, b)
        assert ac.shape == bc.shape, (ac.shape, bc.shape)
        out = torch.empty_like(ac)
        rk = _round_kind() if op in (MUL, GATE) else 0     # autocast: a product of two half tensors is a half tensor
        check(_L().mdx_op_ew_fwd_t(op | (rk << 8), ptr(ac), ptr(bc), ptr(out), ac.numel(), _h(ac) | (_h(bc) << 1) | (_h(out) << 2), stream()))
        ctx.op = op
        ctx.save_for_backward(ac, bc)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _t(g)
        da = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        check(_L().mdx_op_ew_bwd_t(ctx.op, ptr(a), ptr(b), ptr(g), ptr(da), ptr(db), a.numel(),
                                   _h(a) | (_h(b) << 1) | (_h(g) << 2) | (_h(da) << 3) | (_h(db) << 4), stream()))
        if da is not None and da.dtype != ctx.dtypes[0]:
            da = da.to(ctx.dtypes[0])
        if db is not None and db.dtype != ctx.dtypes[1]:
            db = db.to(ctx.dtypes[1])
        return None, da, db


def _ew(op, a, b):
    sk = _SINK
    if sk is not None and sk['fast'] is not None and a.is_cuda and sk['fast'].fast_mode():
        return sk['fast'].ew(op, a, b)
    return _Ew.apply(op, a, b)


def add(a, b):
    return _ew(ADD, a, b)


def sub(a, b):
    return _ew(SUB, a, b)


def mul(a, b):
    return _ew(MUL, a, b)


def gate(a, b):
    """a * sigmoid(b)"""
    return _ew(GATE, a, b)


_FANOUT = __import__('os').environ.get('MDX_TRAIN_FANOUT', '1') != '0'


class _Fanout(torch.autograd.Function):
    """k aliases of x for its k consumers; the backward sums their gradients in ONE launch (csrc sum_n4_kernel: fp32 sum in consumer
    order, one rounding to x's container) where autograd would run k - 1 element-wise adds."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.dtype, ctx.shape = x.dtype, x.shape
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        import ctypes
        gs = [_t(g) for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            g = gs[0]
            return (g if g.dtype == ctx.dtype else g.to(ctx.dtype)), None
        odt = ctx.dtype if ctx.dtype in (torch.float16, torch.float32) else torch.float32
        out = torch.empty(ctx.shape, dtype=odt, device=gs[0].device)
        k = len(gs)
        ptrs = (ctypes.c_void_p * k)(*[g.data_ptr() for g in gs])
        halfs = (ctypes.c_int32 * k)(*[_h(g) for g in gs])
        check(_L().mdx_op_sum_n(ptrs, halfs, k, out.numel(), ptr(out), _h(out), stream()))
        return (out if odt == ctx.dtype else out.to(ctx.dtype)), None


def fanout(x, k):
    """k aliases of x, one per consumer (see _Fanout); plain repetition when there is nothing to gain"""
    if k < 3 or k > 12 or not _FANOUT or not x.is_cuda or not x.requires_grad or not torch.is_grad_enabled():
        return (x,) * k
    return _Fanout.apply(x, k)


class IndexPlan:
    """An index vector (rows -> targets in [0, n)) with the CSR needed to sum rows per target without atomics:
    `order` = stable argsort, `ptr` = segment starts.  Built once per batch (index bookkeeping, not arithmetic)."""

    def __init__(self, index, n):
        _lib._need_gpu(index)
        self.index = index.detach().to(torch.int64).contiguous()
        self.n = int(n)
        srt = torch.sort(self.index, stable=True)
        self.order = srt.indices.contiguous()
        # segment starts from the sorted values (torch.bincount would read the maximum back to the host: a device sync per plan)
        self.ptr = torch.searchsorted(srt.values, torch.arange(self.n + 1, dtype=torch.int64, device=index.device))


class FlippedPlan:
    """IndexPlan of the RIGHT end points of a directed edge list [half-edges ; flipped half-edges] derived from the plan of the left ones
    (csrc plan_flip_kernel): same segment starts, the order by one launch instead of a second stable sort."""

    def __init__(self, left_plan, index, n_half):
        self.index = index.detach().to(torch.int64).contiguous()
        self.n, self.ptr = left_plan.n, left_plan.ptr
        self.order = torch.empty_like(left_plan.order)
        check(_L().mdx_op_plan_flip(ptr(left_plan.order), ptr(left_plan.ptr), self.n, int(n_half), ptr(self.order), stream()))


def _gather_raw(x, plan, out_dtype=None):
    M, F = plan.index.numel(), x.shape[1]
    y = torch.empty(M, F, dtype=out_dtype or x.dtype, device=x.device)
    check(_L().mdx_op_gather_rows_t(ptr(x), ptr(plan.index), M, F, ptr(y), _h(x) | (_h(y) << 1), stream()))
    return y


def _segsum_raw(src, plan, out_dtype=torch.float32):
    F = src.shape[1]
    out = torch.empty(plan.n, F, dtype=out_dtype, device=src.device)
    # bit 2: in an autocast mode fp32 rows may be summed in the dealt order too (the fp32 mode keeps the reference's sequential order)
    check(_L().mdx_op_segsum_rows_t(ptr(src), ptr(plan.order), ptr(plan.ptr), plan.n, F, ptr(out),
                                    _h(src) | (_h(out) << 1) | (4 if (_AMP is not None and _AMP[1]) else 0), stream()))
    return out


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan, ctx.dtype = plan, x.dtype
        return _gather_raw(_t4(x), plan)

    @staticmethod
    def backward(ctx, g):
        g = _t4(g)
        half = ctx.dtype == torch.float16 and g.shape[-1] % 4 == 0
        r = _segsum_raw(g, ctx.plan, torch.float16 if half else torch.float32)
        return (r if r.dtype == ctx.dtype else r.to(ctx.dtype)), None


class _SegSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan):
        ctx.plan, ctx.dtype = plan, src.dtype
        return _segsum_raw(_t4(src), plan)            # sums are fp32 whatever the rows' container

    @staticmethod
    def backward(ctx, g):
        g = _t4(g)
        half = ctx.dtype == torch.float16 and g.shape[-1] % 4 == 0
        r = _gather_raw(g, ctx.plan, torch.float16 if half else torch.float32)
        return (r if r.dtype == ctx.dtype else r.to(ctx.dtype)), None


def gather(x, plan):
    """x[plan.index] for 2-D x."""
    return _Gather.apply(x, plan)


def scatter_sum(src, plan):
    """torch_scatter.scatter_sum(src, plan.index, dim=0, dim_size=plan.n)."""
    return _SegSum.apply(src, plan)


class _MulGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, table, plan):
        ctx.dtypes = (a.dtype, table.dtype)
        ac, tc = _same(a, table)
        M, F = ac.shape
        y = torch.empty_like(ac)
        check(_L().mdx_op_mul_gather_fwd_t(ptr(ac), ptr(tc), ptr(plan.index), M, F | (_round_kind() << 16), ptr(y),
                                           _h(ac) | (_h(tc) << 1) | (_h(y) << 2), stream()))
        ctx.plan = plan
        ctx.save_for_backward(ac, tc)
        return y

    @staticmethod
    def backward(ctx, g):
        a, t = ctx.saved_tensors
        g, plan = _t(g), ctx.plan
        M, F = a.shape
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        dt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        check(_L().mdx_op_mul_gather_bwd_t(ptr(g), ptr(a), ptr(t), ptr(plan.index), ptr(plan.order), ptr(plan.ptr), M, plan.n, F, ptr(da),
                                           ptr(dt), _h(g) | (_h(a) << 1) | (_h(t) << 2) | (_h(da) << 3) | (_h(dt) << 4), stream()))
        if da is not None and da.dtype != ctx.dtypes[0]:
            da = da.to(ctx.dtypes[0])
        if dt is not None and dt.dtype != ctx.dtypes[1]:
            dt = dt.to(ctx.dtypes[1])
        return da, dt, None


def mul_gather(a, table, plan):
    """a * table[plan.index] for 2-D a (rows, F), F % 4 == 0, without materialising the gathered rows."""
    if a.shape[1] % 4:
        return mul(a, gather(table, plan))
    return _MulGather.apply(a, table, plan)


class _EdgeGeom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, pl, pr):
        p = _c(pos)
        E = pl.index.numel()
        rel = torch.empty(E, 3, dtype=torch.float32, device=p.device)
        dist = torch.empty(E, dtype=torch.float32, device=p.device)
        check(_L().mdx_op_edge_geom_fwd(ptr(p), ptr(pl.index), ptr(pr.index), E, ptr(rel), ptr(dist), stream()))
        ctx.pl, ctx.pr, ctx.prec = pl, pr, _AMP
        ctx.save_for_backward(rel, dist)
        return rel, dist

    @staticmethod
    def backward(ctx, grel, gdist):
        rel, dist = ctx.saved_tensors
        E = rel.shape[0]
        g = torch.empty_like(rel)
        check(_L().mdx_op_edge_geom_bwd(ptr(rel), ptr(dist), ptr(_c(grel) if grel is not None else None),
                                        ptr(_c(gdist) if gdist is not None else None), E, ptr(g), stream()))
        with precision(ctx.prec):
            gl, gr = _segsum_raw(g, ctx.pl), _segsum_raw(g, ctx.pr)
        out = torch.empty_like(gl)
        check(_L().mdx_op_ew_fwd(SUB, ptr(gl), ptr(gr), ptr(out), gl.numel(), stream()))
        return out, None, None


def edge_geom(pos, plan_left, plan_right):
    """rel = pos[left] - pos[right] (E,3), dist = |rel| (E,)."""
    return _EdgeGeom.apply(pos, plan_left, plan_right)


class _Smear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dist, offset, coeff, lo, hi):
        d, off, co = _c(dist), _c(offset), _c(coeff)
        E, G = d.numel(), off.numel()
        out = torch.empty(E, G, dtype=torch.float32, device=d.device)
        check(_L().mdx_op_smear_fwd(ptr(d), ptr(off), ptr(co), G, float(lo), float(hi), E, ptr(out), stream()))
        ctx.save_for_backward(d, off, co)
        ctx.lohi = (float(lo), float(hi))
        return out

    @staticmethod
    def backward(ctx, g):
        d, off, co = ctx.saved_tensors
        gd = torch.empty_like(d)
        check(_L().mdx_op_smear_bwd(ptr(d), ptr(off), ptr(co), off.numel(), ctx.lohi[0], ctx.lohi[1], d.numel(), ptr(_c(g)), ptr(gd),
                                    stream()))
        return gd, None, None, None, None


def smear(dist, offset, coeff, lo, hi):
    """GaussianSmearing with per-gaussian coefficients: exp(coeff_k (clamp(d, lo, hi) - offset_k)^2)."""
    return _Smear.apply(dist, offset, coeff, lo, hi)


class _Force(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, rel, dist):
        wc, rc, dc = _c(w).reshape(-1), _c(rel), _c(dist)
        out = torch.empty_like(rc)
        check(_L().mdx_op_force_fwd(ptr(wc), ptr(rc), ptr(dc), dc.numel(), ptr(out), stream()))
        ctx.save_for_backward(wc, rc, dc)
        ctx.wshape, ctx.wdtype = w.shape, w.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        w, rel, d = ctx.saved_tensors
        gw, grel, gd = torch.empty_like(w), torch.empty_like(rel), torch.empty_like(d)
        check(_L().mdx_op_force_bwd(ptr(w), ptr(rel), ptr(d), ptr(_c(g)), d.numel(), ptr(gw), ptr(grel), ptr(gd), stream()))
        return gw.reshape(ctx.wshape).to(ctx.wdtype), grel, gd


def force(w, rel, dist):
    """w * rel / dist / (dist + 1) per edge (PosUpdate, models/graph.py:393)."""
    return _Force.apply(w, rel, dist)


# ---- fused row-owner operators (round 6; csrc/mdx_train_fused.hip) -------------------------------------------------------------------
# A whole BondFFN of the EdgeBlock + the scatter_sum that follows it as ONE autograd node: 2 launches forward (fused chain, segment sum)
# and 9 backward (fused data-gradient chain, 6 weight-gradient contractions, 2 segment sums) where the per-operator composition issues
# 12 and ~28.  float16 autocast mode with float16 containers only; every other mode keeps the per-operator path.
FUSED_MIN_ROWS = int(__import__('os').environ.get('MDX_TRAIN_FUSED_MIN_ROWS', '1024'))   # below this the per-operator path runs (tests lower it to cover the fused path on small graphs)
_FUSED = __import__('os').environ.get('MDX_TRAIN_FUSED', '1') != '0'


def bondffn_fused_ok(bond_in, node_lin, gate_node, dims):
    """dims = (bond, inter, out, gate hidden, node columns of the gate): the kernel is built for (64, 128, 64, 32, any)."""
    return (_FUSED and _AMP is not None and _AMP[0] == 2 and _AMP[1] and _AMP[2] and bond_in.dtype == torch.float16 and bond_in.dim() == 2
            and bond_in.shape[0] >= FUSED_MIN_ROWS and dims[:4] == (64, 128, 64, 32) and node_lin.dtype == torch.float16
            and gate_node.dtype == torch.float32)


def _wslice(w):
    """a weight or a column slice of one: fp32, unit column stride (row stride arbitrary)"""
    w = w.detach()
    assert w.dtype == torch.float32 and w.stride(-1) == 1
    return w


class _BondFfnScatter(torch.autograd.Function):
    """scatter_sum(BondFFN(bond_in, h_node[idx], time), oidx) with the node-side Linears hoisted by the caller:
    args = bond_in (E,64) f16, NL = node_linear(h) (N,128) f16, GN = gate.net.0[:, node columns](h) (N,32) fp32, time (E,1) fp32,
    plan_in (idx), plan_out (oidx), then the parameters in the order of PARAMS."""
    PARAMS = ('Wb', 'Wi1', 'bi1', 'g1', 'be1', 'Wi2', 'bi2', 'Wg1', 'bg1', 'gg', 'gbe', 'Wt', 'Wg2', 'bg2')

    @staticmethod
    def _args(x, NL, GN, time, plan_in, P, bufs, E):
        a = _lib.MdxBondFfnArgs()
        a.X, a.ldx = x.data_ptr(), x.stride(0)
        for nm, ld in (('Wb', 'ldwb'), ('Wi1', 'ldwi1'), ('Wi2', 'ldwi2'), ('Wg1', 'ldwg1'), ('Wg2', 'ldwg2')):
            setattr(a, nm, P[nm].data_ptr())
            setattr(a, ld, P[nm].stride(0))
        a.Wt, a.ldwt = P['Wt'].data_ptr(), P['Wt'].stride(0)
        for nm in ('bi1', 'g1', 'be1', 'bi2', 'bg1', 'gg', 'gbe', 'bg2'):
            setattr(a, nm, P[nm].data_ptr())
        a.NL, a.ldnl, a.GN, a.ldgn = NL.data_ptr(), NL.stride(0), GN.data_ptr(), GN.stride(0)
        a.idx, a.te = plan_in.index.data_ptr(), time.data_ptr()
        for nm, t in bufs.items():
            setattr(a, nm, t.data_ptr())
        a.E = E
        return a

    @staticmethod
    def forward(ctx, bond_in, NL, GN, time, plan_in, plan_out, *params):
        import ctypes
        x = _rows(bond_in)
        if x.stride(0) % 8 or x.data_ptr() % 16:
            x = x.contiguous()
        NLc, GNc = _rows(NL), _rows(GN)
        tc = _c(time).reshape(-1)
        P = {k: _wslice(v) if v.dim() == 2 else _c(v) for k, v in zip(_BondFfnScatter.PARAMS, params)}
        E, dev = x.shape[0], x.device
        h = lambda f: torch.empty(E, f, dtype=torch.float16, device=dev)
        bufs = {'prod': h(128), 'pre1': h(128), 'post1': h(128), 'inter': h(64), 'gpre': h(32), 'gpost': h(32), 'gate': h(64), 'out': h(64)}
        a = _BondFfnScatter._args(x, NLc, GNc, tc, plan_in, P, bufs, E)
        check(_L().mdx_op_bondffn_fwd(ctypes.byref(a), stream()))
        ctx.x, ctx.NL, ctx.GN, ctx.time, ctx.P, ctx.bufs = x, NLc, GNc, tc, P, bufs
        ctx.plan_in, ctx.plan_out = plan_in, plan_out
        ctx.refs = {k: v.detach() for k, v in zip(_BondFfnScatter.PARAMS, params)}     # (views of the parameters: addresses for the gradient sink)
        ctx.time2d = time.detach()
        ctx.prec = _AMP
        ctx.x_dtype = bond_in.dtype
        return _segsum_raw(bufs['out'], plan_out)

    @staticmethod
    def backward(ctx, gS):
        import ctypes
        x, P, bufs, E, dev = ctx.x, ctx.P, ctx.bufs, ctx.x.shape[0], ctx.x.device
        gS = _c(gS)
        h = lambda f: torch.empty(E, f, dtype=torch.float16, device=dev)
        g = {'g_inter': h(64), 'g_gate': h(64), 'g_pre1': h(128), 'g_bf': h(128), 'g_nl': h(128), 'g_gpre': h(32), 'g_x': h(64)}
        nwg, lnf = int(_L().mdx_op_bondffn_workgroups()), int(_L().mdx_op_bondffn_lnp_floats())
        lnp = torch.empty(nwg, lnf, dtype=torch.float32, device=dev)
        b = _lib.MdxBondFfnBwdArgs()
        b.f = _BondFfnScatter._args(x, ctx.NL, ctx.GN, ctx.time, ctx.plan_in, P, bufs, E)
        b.gS, b.ldgs, b.oidx = gS.data_ptr(), gS.stride(0), ctx.plan_out.index.data_ptr()
        for nm, t in g.items():
            setattr(b, nm, t.data_ptr())
        b.lnp = lnp.data_ptr()
        check(_L().mdx_op_bondffn_bwd(ctypes.byref(b), stream()))
        need = dict(zip(_BondFfnScatter.PARAMS, ctx.needs_input_grad[6:]))
        grads = {k: None for k in _BondFfnScatter.PARAMS}
        with precision(ctx.prec):
            wgrad = lambda gy, xin, wname, bname: _wgrad_into(grads, need, ctx.refs, E, gy, xin, wname, bname)
            wgrad(g['g_inter'], bufs['post1'], 'Wi2', 'bi2')
            wgrad(g['g_gate'], bufs['gpost'], 'Wg2', 'bg2')
            wgrad(g['g_pre1'], bufs['prod'], 'Wi1', 'bi1')
            wgrad(g['g_bf'], x, 'Wb', None)
            wgrad(g['g_gpre'], x, 'Wg1', 'bg1')
            wgrad(g['g_gpre'], ctx.time2d if ctx.time2d.dim() == 2 else ctx.time2d.reshape(-1, 1), 'Wt', None)
        # LayerNorm-parameter gradients: one partial row per workgroup
        for nm, off, F in (('g1', 0, 128), ('be1', 128, 128), ('gg', 256, 32), ('gbe', 288, 32)):
            if not need[nm]:
                continue
            dst = _sink_dst(ctx.refs[nm])
            if dst is not None:
                _sink_record(lnp.data_ptr() + 4 * off, dst, nwg, 1, F, F, lnf, 0, lnp)
            else:
                grads[nm] = lnp[:, off:off + F].sum(0)
        ni = ctx.needs_input_grad
        g_x = g['g_x'] if ni[0] else None
        if g_x is not None and g_x.dtype != ctx.x_dtype:
            g_x = g_x.to(ctx.x_dtype)
        g_NL = _segsum_raw(g['g_nl'], ctx.plan_in, torch.float16) if ni[1] else None
        g_GN = _segsum_raw(g['g_gpre'], ctx.plan_in, torch.float32) if ni[2] else None
        ctx.bufs = ctx.P = None
        return (g_x, g_NL, g_GN, None, None, None) + tuple(grads[k] for k in _BondFfnScatter.PARAMS)


def bondffn_scatter(bond_in, node_lin, gate_node, time, plan_in, plan_out, params):
    """params: dict with the keys of _BondFfnScatter.PARAMS (parameters or column slices of them)"""
    ps = [params[k] for k in _BondFfnScatter.PARAMS]
    F = _fast_for(ps)
    if F is not None:
        return _BondFfnScatterF.apply(F, bond_in, node_lin, gate_node, time, plan_in, plan_out, *ps)
    return _BondFfnScatter.apply(bond_in, node_lin, gate_node, time, plan_in, plan_out, *ps)


def _wgrad_into(grads, need, refs, E, gy, xin, wname, bname):
    """weight (+ bias) gradient of one Linear inside a fused node: deferred into the flat gradient buffer when a sink holds the parameter,
    else returned through `grads`"""
    if not need[wname]:
        return
    w = refs[wname]
    want_b = bname is not None and need[bname]
    dst_w = _sink_dst(w)
    dst_b = _sink_dst(refs[bname]) if want_b else None
    sp = _splits_for(E, gy.shape[1], xin.shape[1], _h(gy) and _h(xin))
    if dst_w is not None and (not want_b or dst_b is not None):
        sgemm_tn(gy, xin, sp, want_bias=want_b, defer=(dst_w, w.stride(0), dst_b))
    else:
        r = sgemm_tn(gy, xin, sp, want_bias=want_b)
        grads[wname], gb_ = r if want_b else (r, None)
        if want_b:
            grads[bname] = gb_


_FUSED_TAIL = __import__('os').environ.get('MDX_TRAIN_FUSED_TAIL', '1') != '0'


def edge_tail_fused_ok(h_bond, by_left, by_right):
    return (_FUSED and _FUSED_TAIL and _AMP is not None and _AMP[0] == 2 and _AMP[1] and _AMP[2] and h_bond.dtype == torch.float16 and h_bond.dim() == 2
            and h_bond.shape[1] == 64 and h_bond.shape[0] >= FUSED_MIN_ROWS and by_left.dtype == torch.float16 and by_right.dtype == torch.float16
            and by_left.shape[1] == 64 and by_right.shape[1] == 64)


class _EdgeTail(torch.autograd.Function):
    """h + out_transform(relu(LN(self_ffn(h) + BL[left] + BR[right]))) (models/graph.py:281-294 + the residual of :360) as one node:
    args = h (E,64) f16, BL, BR (N,64) f16, plan_left, plan_right, then PARAMS."""
    PARAMS = ('Ws', 'bs', 'lng', 'lnb', 'Wo', 'bo')

    @staticmethod
    def _args(x, BL, BR, pl, pr, P, pre, post, out, E):
        a = _lib.MdxEdgeTailArgs()
        a.H, a.ldh, a.BL, a.ldbl, a.BR, a.ldbr = x.data_ptr(), x.stride(0), BL.data_ptr(), BL.stride(0), BR.data_ptr(), BR.stride(0)
        a.il, a.ir = pl.index.data_ptr(), pr.index.data_ptr()
        a.Ws, a.ldws, a.Wo, a.ldwo = P['Ws'].data_ptr(), P['Ws'].stride(0), P['Wo'].data_ptr(), P['Wo'].stride(0)
        for nm in ('bs', 'lng', 'lnb', 'bo'):
            setattr(a, nm, P[nm].data_ptr())
        a.pre, a.post, a.out = pre.data_ptr(), post.data_ptr(), (out.data_ptr() if out is not None else None)
        a.E = E
        return a

    @staticmethod
    def forward(ctx, h, BL, BR, plan_l, plan_r, *params):
        import ctypes
        x = _rows(h)
        if x.stride(0) % 8 or x.data_ptr() % 16:
            x = x.contiguous()
        BLc, BRc = _rows(BL), _rows(BR)
        P = {k: _wslice(v) if v.dim() == 2 else _c(v) for k, v in zip(_EdgeTail.PARAMS, params)}
        E, dev = x.shape[0], x.device
        pre, post, out = (torch.empty(E, 64, dtype=torch.float16, device=dev) for _ in range(3))
        a = _EdgeTail._args(x, BLc, BRc, plan_l, plan_r, P, pre, post, out, E)
        check(_L().mdx_op_edge_tail_fwd(ctypes.byref(a), stream()))
        ctx.x, ctx.BL, ctx.BR, ctx.P, ctx.pre, ctx.post = x, BLc, BRc, P, pre, post
        ctx.plan_l, ctx.plan_r, ctx.prec, ctx.x_dtype = plan_l, plan_r, _AMP, h.dtype
        ctx.refs = {k: v.detach() for k, v in zip(_EdgeTail.PARAMS, params)}
        return out

    @staticmethod
    def backward(ctx, g_out):
        import ctypes
        x, P, E, dev = ctx.x, ctx.P, ctx.x.shape[0], ctx.x.device
        g_out = _rows(g_out)
        if g_out.dtype != torch.float16:
            g_out = g_out.to(torch.float16)
        if g_out.stride(0) % 4 or g_out.data_ptr() % 8:
            g_out = g_out.contiguous()
        g_pre, g_h = torch.empty(E, 64, dtype=torch.float16, device=dev), torch.empty(E, 64, dtype=torch.float16, device=dev)
        nwg, lnf = int(_L().mdx_op_bondffn_workgroups()), int(_L().mdx_op_edge_tail_lnp_floats())
        lnp = torch.empty(nwg, lnf, dtype=torch.float32, device=dev)
        b = _lib.MdxEdgeTailBwdArgs()
        b.f = _EdgeTail._args(x, ctx.BL, ctx.BR, ctx.plan_l, ctx.plan_r, P, ctx.pre, ctx.post, None, E)
        b.g_out, b.ldg, b.g_pre, b.g_h, b.lnp = g_out.data_ptr(), g_out.stride(0), g_pre.data_ptr(), g_h.data_ptr(), lnp.data_ptr()
        check(_L().mdx_op_edge_tail_bwd(ctypes.byref(b), stream()))
        need = dict(zip(_EdgeTail.PARAMS, ctx.needs_input_grad[5:]))
        grads = {k: None for k in _EdgeTail.PARAMS}
        with precision(ctx.prec):
            _wgrad_into(grads, need, ctx.refs, E, g_out, ctx.post, 'Wo', 'bo')
            _wgrad_into(grads, need, ctx.refs, E, g_pre, x, 'Ws', 'bs')
        for nm, off in (('lng', 0), ('lnb', 64)):
            if not need[nm]:
                continue
            dst = _sink_dst(ctx.refs[nm])
            if dst is not None:
                _sink_record(lnp.data_ptr() + 4 * off, dst, nwg, 1, 64, 64, lnf, 0, lnp)
            else:
                grads[nm] = lnp[:, off:off + 64].sum(0)
        ni = ctx.needs_input_grad
        gh = g_h if ni[0] else None
        if gh is not None and gh.dtype != ctx.x_dtype:
            gh = gh.to(ctx.x_dtype)
        gBL = _segsum_raw(g_pre, ctx.plan_l, torch.float16) if ni[1] else None
        gBR = _segsum_raw(g_pre, ctx.plan_r, torch.float16) if ni[2] else None
        ctx.pre = ctx.post = ctx.P = None
        return (gh, gBL, gBR, None, None) + tuple(grads[k] for k in _EdgeTail.PARAMS)


def edge_tail(h, by_left, by_right, plan_l, plan_r, params):
    ps = [params[k] for k in _EdgeTail.PARAMS]
    F = _fast_for(ps)
    if F is not None:
        return _EdgeTailF.apply(F, h, by_left, by_right, plan_l, plan_r, *ps)
    return _EdgeTail.apply(h, by_left, by_right, plan_l, plan_r, *ps)


_FUSED_POS = __import__('os').environ.get('MDX_TRAIN_FUSED_POS', '1') != '0'


def posffn_fused_ok(h_edge, lf, rf, dims):
    """dims = (bond, node, inter, gate hidden, out): the kernel is built for PosUpdate's BondFFN (64, 64, 256, 32, 1)"""
    return (_FUSED and _FUSED_POS and _AMP is not None and _AMP[0] == 2 and _AMP[1] and _AMP[2] and h_edge.dtype == torch.float16 and h_edge.dim() == 2
            and h_edge.shape[0] >= FUSED_MIN_ROWS and dims == (64, 64, 256, 32, 1) and lf.dtype == torch.float16 and rf.dtype == torch.float16
            and lf.shape[1] == 64 and rf.shape[1] == 64)


class _PosFfnFront(torch.autograd.Function):
    """(prod, gate) of PosUpdate's BondFFN (models/graph.py:388-390 with :133-141): prod = bond_linear(h_edge) * node_linear(a) (E,256),
    gate = gate MLP([h_edge | a | t]) (E,1), a = LF[left] * RF[right].  args = h_edge (E,64) f16, LF, RF (N,64) f16, time (E,1) fp32,
    plan_left, plan_right, then PARAMS."""
    PARAMS = ('Wb', 'Wn', 'Wg1x', 'Wg1a', 'Wt', 'bg1', 'gg', 'gbe', 'Wg2', 'bg2')

    @staticmethod
    def _args(x, LF, RF, tc, pl, pr, P, bufs, E):
        a = _lib.MdxPosFfnArgs()
        a.X, a.ldx, a.LF, a.ldlf, a.RF, a.ldrf = x.data_ptr(), x.stride(0), LF.data_ptr(), LF.stride(0), RF.data_ptr(), RF.stride(0)
        a.il, a.ir, a.te = pl.index.data_ptr(), pr.index.data_ptr(), tc.data_ptr()
        for nm, ld in (('Wb', 'ldwb'), ('Wn', 'ldwn'), ('Wg1x', 'ldwg1x'), ('Wg1a', 'ldwg1a'), ('Wt', 'ldwt')):
            setattr(a, nm, P[nm].data_ptr())
            setattr(a, ld, P[nm].stride(0))
        for nm in ('bg1', 'gg', 'gbe', 'Wg2', 'bg2'):
            setattr(a, nm, P[nm].data_ptr())
        for nm, t in bufs.items():
            setattr(a, nm, t.data_ptr())
        a.E = E
        return a

    @staticmethod
    def forward(ctx, h_edge, LF, RF, time, plan_l, plan_r, *params):
        import ctypes
        al = lambda t: t if (t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0) else t.contiguous()
        x, LFc, RFc = al(_rows(h_edge)), al(_rows(LF)), al(_rows(RF))
        tc = _c(time).reshape(-1)
        P = {}
        for k, v in zip(_PosFfnFront.PARAMS, params):
            P[k] = _wslice(v) if (v.dim() == 2 and k != 'Wg2') else _c(v).reshape(-1) if k in ('Wg2', 'bg2') else _c(v)
        E, dev = x.shape[0], x.device
        h = lambda *f: torch.empty(E, *f, dtype=torch.float16, device=dev)
        bufs = {'a': h(64), 'prod': h(256), 'gpre': h(32), 'gpost': h(32), 'gate': h(1)}
        a = _PosFfnFront._args(x, LFc, RFc, tc, plan_l, plan_r, P, bufs, E)
        check(_L().mdx_op_posffn_fwd(ctypes.byref(a), stream()))
        ctx.x, ctx.LF, ctx.RF, ctx.tc, ctx.P, ctx.bufs = x, LFc, RFc, tc, P, bufs
        ctx.plan_l, ctx.plan_r, ctx.prec, ctx.x_dtype = plan_l, plan_r, _AMP, h_edge.dtype
        ctx.refs = {k: v.detach() for k, v in zip(_PosFfnFront.PARAMS, params)}
        ctx.time2d = time.detach().reshape(-1, 1)
        return bufs['prod'], bufs['gate']

    @staticmethod
    def backward(ctx, g_prod, g_gate):
        import ctypes
        x, P, bufs, E, dev = ctx.x, ctx.P, ctx.bufs, ctx.x.shape[0], ctx.x.device
        f16 = lambda t, f: (torch.zeros(E, f, dtype=torch.float16, device=dev) if t is None else
                            (t if t.dtype == torch.float16 else t.to(torch.float16)).contiguous())
        g_prod, g_gate = f16(g_prod, 256), f16(g_gate, 1)
        h = lambda f: torch.empty(E, f, dtype=torch.float16, device=dev)
        g = {'g_bf': h(256), 'g_nf': h(256), 'g_gpre': h(32), 'g_x': h(64), 'g_lf': h(64), 'g_rf': h(64)}
        nwg, lnf = int(_L().mdx_op_bondffn_workgroups()), int(_L().mdx_op_posffn_lnp_floats())
        lnp = torch.empty(nwg, lnf, dtype=torch.float32, device=dev)
        b = _lib.MdxPosFfnBwdArgs()
        b.f = _PosFfnFront._args(x, ctx.LF, ctx.RF, ctx.tc, ctx.plan_l, ctx.plan_r, P, bufs, E)
        b.g_prod, b.ldgp, b.g_gate, b.lnp = g_prod.data_ptr(), g_prod.stride(0), g_gate.data_ptr(), lnp.data_ptr()
        for nm, t in g.items():
            setattr(b, nm, t.data_ptr())
        check(_L().mdx_op_posffn_bwd(ctypes.byref(b), stream()))
        need = dict(zip(_PosFfnFront.PARAMS, ctx.needs_input_grad[6:]))
        grads = {k: None for k in _PosFfnFront.PARAMS}
        with precision(ctx.prec):
            wg = lambda gy, xin, wname, bname: _wgrad_into(grads, need, ctx.refs, E, gy, xin, wname, bname)
            wg(g['g_bf'], x, 'Wb', None)
            wg(g['g_nf'], bufs['a'], 'Wn', None)
            wg(g['g_gpre'], x, 'Wg1x', 'bg1')
            wg(g['g_gpre'], bufs['a'], 'Wg1a', None)
            wg(g['g_gpre'], ctx.time2d, 'Wt', None)
            wg(g_gate, bufs['gpost'], 'Wg2', 'bg2')
        for nm, off in (('gg', 0), ('gbe', 32)):
            if not need[nm]:
                continue
            dst = _sink_dst(ctx.refs[nm])
            if dst is not None:
                _sink_record(lnp.data_ptr() + 4 * off, dst, nwg, 1, 32, 32, lnf, 0, lnp)
            else:
                grads[nm] = lnp[:, off:off + 32].sum(0)
        ni = ctx.needs_input_grad
        g_x = g['g_x'] if ni[0] else None
        if g_x is not None and g_x.dtype != ctx.x_dtype:
            g_x = g_x.to(ctx.x_dtype)
        g_LF = _segsum_raw(g['g_lf'], ctx.plan_l, torch.float16) if ni[1] else None
        g_RF = _segsum_raw(g['g_rf'], ctx.plan_r, torch.float16) if ni[2] else None
        ctx.bufs = ctx.P = None
        return (g_x, g_LF, g_RF, None, None, None) + tuple(grads[k] for k in _PosFfnFront.PARAMS)


def posffn_front(h_edge, lf, rf, time, plan_l, plan_r, params):
    ps = [params[k] for k in _PosFfnFront.PARAMS]
    F = _fast_for(ps)
    if F is not None and ps[8].is_contiguous() and ps[9].is_contiguous():
        return _PosFfnFrontF.apply(F, h_edge, lf, rf, time, plan_l, plan_r, *ps)
    return _PosFfnFront.apply(h_edge, lf, rf, time, plan_l, plan_r, *ps)


_FUSED_NODE = __import__('os').environ.get('MDX_TRAIN_FUSED_NODE', '1') != '0'


def nodemsg_fused_ok(edge_attr, hn, pn, shapes):
    """shapes = (edge_net.0, edge_net.3, msg_net, gate.0 edge columns, gate.3 weight shapes): built for 64 -> 256 -> 256"""
    return (_FUSED and _FUSED_NODE and _AMP is not None and _AMP[0] == 2 and _AMP[1] and _AMP[2] and edge_attr.dtype == torch.float16
            and edge_attr.dim() == 2 and edge_attr.shape[1] == 64 and edge_attr.shape[0] >= FUSED_MIN_ROWS and hn.dtype == torch.float16
            and hn.shape[1] == 256 and pn.dtype == torch.float32 and pn.shape[1] == 256
            and shapes == ((256, 64), (256, 256), (256, 256), (256, 64), (256, 256)))


class _NodeMsg(torch.autograd.Function):
    """scatter_sum(msg, left) of the NodeBlock message path (models/graph.py:40-50) as one node:
    args = edge_attr (E,64) f16, HN = node_net(x) (N,256) f16, PN = gate.net.0[:, node + time columns](cat(x, t)) (N,256) fp32,
    plan_col (right), plan_row (left), then PARAMS."""
    PARAMS = ('W1e', 'b1e', 'lng_e', 'lnb_e', 'W2e', 'b2e', 'Wm', 'bm', 'Wg1', 'bg1', 'lng_g', 'lnb_g', 'Wg2', 'bg2')
    # (name, weight, n_out, n_in, perm, trans) of the ten A-operand packs: five forward, five backward (transposes)
    PACKS = (('pk_w1e', 'W1e', 256, 64, 0, 0), ('pk_w2e', 'W2e', 256, 256, 1, 0), ('pk_wm', 'Wm', 256, 256, 1, 0), ('pk_wg1', 'Wg1', 256, 64, 0, 0),
             ('pk_wg2', 'Wg2', 256, 256, 1, 0), ('pk_wg2t', 'Wg2', 256, 256, 1, 1), ('pk_wg1t', 'Wg1', 64, 256, 1, 1), ('pk_wmt', 'Wm', 256, 256, 1, 1),
             ('pk_w2et', 'W2e', 256, 256, 1, 1), ('pk_w1et', 'W1e', 64, 256, 1, 1))

    @staticmethod
    def _args(x, HN, PN, plan_col, P, packs, bufs, E):
        a = _lib.MdxNodeMsgArgs()
        a.X, a.ldx, a.HN, a.ldhn, a.PN, a.ldpn, a.col = x.data_ptr(), x.stride(0), HN.data_ptr(), HN.stride(0), PN.data_ptr(), PN.stride(0), plan_col.index.data_ptr()
        for nm in ('pk_w1e', 'pk_w2e', 'pk_wm', 'pk_wg1', 'pk_wg2'):
            setattr(a, nm, packs[nm])
        for nm in ('b1e', 'lng_e', 'lnb_e', 'b2e', 'bm', 'bg1', 'lng_g', 'lnb_g', 'bg2'):
            setattr(a, nm, P[nm].data_ptr())
        for nm, t in bufs.items():
            setattr(a, nm, t.data_ptr())
        a.E = E
        return a

    @staticmethod
    def forward(ctx, edge_attr, HN, PN, plan_col, plan_row, *params):
        import ctypes
        x = _rows(edge_attr)
        if x.stride(0) % 8 or x.data_ptr() % 16:
            x = x.contiguous()
        HNc, PNc = _rows(HN), _rows(PN)
        P = {k: _wslice(v) if v.dim() == 2 else _c(v) for k, v in zip(_NodeMsg.PARAMS, params)}
        E, dev = x.shape[0], x.device
        # the ten weight packs of this call: one buffer, one launch
        sizes = [no * ni for _, _, no, ni, _, _ in _NodeMsg.PACKS]
        buf = torch.empty(sum(sizes), dtype=torch.float16, device=dev)
        jobs, packs, off = _lib.MdxPackJobs(), {}, 0
        for i, (nm, wn, no, ni, perm, trans) in enumerate(_NodeMsg.PACKS):
            j = jobs.job[i]
            j.W, j.ld, j.n_out, j.n_in, j.perm, j.trans = P[wn].data_ptr(), P[wn].stride(0), no, ni, perm, trans
            j.out = packs[nm] = buf.data_ptr() + 2 * off
            off += sizes[i]
        jobs.n = len(_NodeMsg.PACKS)
        check(_L().mdx_op_pack_a(ctypes.byref(jobs), stream()))
        h = lambda: torch.empty(E, 256, dtype=torch.float16, device=dev)
        bufs = {k: h() for k in ('he_pre', 'he_post', 'he', 'p', 'm0', 'g_pre', 'g_post', 'gt', 'msg')}
        a = _NodeMsg._args(x, HNc, PNc, plan_col, P, packs, bufs, E)
        check(_L().mdx_op_nodemsg_fwd(ctypes.byref(a), stream()))
        out = _segsum_raw(bufs['msg'], plan_row)
        del bufs['msg']
        ctx.x, ctx.HN, ctx.PN, ctx.P, ctx.bufs, ctx.packs, ctx.packbuf = x, HNc, PNc, P, bufs, packs, buf
        ctx.plan_col, ctx.plan_row, ctx.prec, ctx.x_dtype = plan_col, plan_row, _AMP, edge_attr.dtype
        ctx.refs = {k: v.detach() for k, v in zip(_NodeMsg.PARAMS, params)}
        return out

    @staticmethod
    def backward(ctx, gA):
        import ctypes
        x, P, bufs, E, dev = ctx.x, ctx.P, ctx.bufs, ctx.x.shape[0], ctx.x.device
        gA = _c(gA)
        h = lambda f=256: torch.empty(E, f, dtype=torch.float16, device=dev)
        g = {'g_m0': h(), 'g_gt': h(), 'g_gpre': h(), 'g_hne': h(), 'g_he': h(), 'g_pre': h(), 'g_x': h(64)}
        nwg, lnf = int(_L().mdx_op_bondffn_workgroups()), int(_L().mdx_op_nodemsg_lnp_floats())
        lnp = torch.empty(nwg, lnf, dtype=torch.float32, device=dev)
        b = _lib.MdxNodeMsgBwdArgs()
        fb = dict(bufs, msg=bufs['m0'])      # (the forward's msg buffer is gone; the backward does not read it)
        b.f = _NodeMsg._args(x, ctx.HN, ctx.PN, ctx.plan_col, P, ctx.packs, fb, E)
        b.gA, b.ldga, b.row = gA.data_ptr(), gA.stride(0), ctx.plan_row.index.data_ptr()
        for nm in ('pk_wg2t', 'pk_wg1t', 'pk_wmt', 'pk_w2et', 'pk_w1et'):
            setattr(b, nm, ctx.packs[nm])
        for nm, t in g.items():
            setattr(b, nm, t.data_ptr())
        b.lnp = lnp.data_ptr()
        check(_L().mdx_op_nodemsg_bwd(ctypes.byref(b), stream()))
        need = dict(zip(_NodeMsg.PARAMS, ctx.needs_input_grad[5:]))
        grads = {k: None for k in _NodeMsg.PARAMS}
        with precision(ctx.prec):
            wg = lambda gy, xin, wname, bname: _wgrad_into(grads, need, ctx.refs, E, gy, xin, wname, bname)
            wg(g['g_m0'], bufs['p'], 'Wm', 'bm')
            wg(g['g_gt'], bufs['g_post'], 'Wg2', 'bg2')
            wg(g['g_gpre'], x, 'Wg1', 'bg1')
            wg(g['g_he'], bufs['he_post'], 'W2e', 'b2e')
            wg(g['g_pre'], x, 'W1e', 'b1e')
        for nm, off in (('lng_e', 0), ('lnb_e', 256), ('lng_g', 512), ('lnb_g', 768)):
            if not need[nm]:
                continue
            dst = _sink_dst(ctx.refs[nm])
            if dst is not None:
                _sink_record(lnp.data_ptr() + 4 * off, dst, nwg, 1, 256, 256, lnf, 0, lnp)
            else:
                grads[nm] = lnp[:, off:off + 256].sum(0)
        ni = ctx.needs_input_grad
        g_x = g['g_x'] if ni[0] else None
        if g_x is not None and g_x.dtype != ctx.x_dtype:
            g_x = g_x.to(ctx.x_dtype)
        g_HN = _segsum_raw(g['g_hne'], ctx.plan_col, torch.float16) if ni[1] else None
        g_PN = _segsum_raw(g['g_gpre'], ctx.plan_col, torch.float32) if ni[2] else None
        ctx.bufs = ctx.P = ctx.packbuf = None
        return (g_x, g_HN, g_PN, None, None) + tuple(grads[k] for k in _NodeMsg.PARAMS)


def nodemsg(edge_attr, hn, pn, plan_col, plan_row, params):
    ps = [params[k] for k in _NodeMsg.PARAMS]
    F = _fast_for(ps)
    if F is not None:
        return _NodeMsgF.apply(F, edge_attr, hn, pn, plan_col, plan_row, *ps)
    return _NodeMsg.apply(edge_attr, hn, pn, plan_col, plan_row, *ps)


# ---- the categorical loss tail as one node (round 6; csrc/mdx_transition.hip cat_loss_kernel) -----------------------------------------
class _CatLoss(torch.autograd.Function):
    """100 x mean over rows of {KL(q(v_{t-1} | v_t, v_0) || p_theta) for t > 0 | decoder NLL at t == 0} (models/model.py:170-189) as ONE
    launch that also leaves d row / d logits; the backward is a scaling.  Replaces ~45 torch launches forward and ~50 backward per loss
    term (log_softmax, the posterior algebra of transition.q_v_posterior_autograd, compute_v_Lt)."""

    @staticmethod
    def forward(ctx, logits, q_mats, qT, log_vt, log_v0, t, batch):
        lg, lvt, lv0 = _c(logits), _c(log_vt), _c(log_v0)
        n, K = lg.shape
        tt = t.detach().to(torch.int64).contiguous()
        bb = batch.detach().to(torch.int64).contiguous()
        row = torch.empty(n, dtype=torch.float32, device=lg.device)
        dl = torch.empty(n, K, dtype=torch.float32, device=lg.device)
        check(_L().mdx_op_cat_loss(ptr(q_mats), ptr(qT), K, q_mats.shape[0], ptr(lg), ptr(lvt), ptr(lv0), ptr(tt), ptr(bb), n, ptr(row), ptr(dl),
                                   stream()))
        ctx.save_for_backward(dl)
        ctx.scale, ctx.dtype = 100.0 / max(n, 1), logits.dtype
        return torch.mean(row) * 100

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        gl = dl * (g * ctx.scale)
        return (gl if gl.dtype == ctx.dtype else gl.to(ctx.dtype)), None, None, None, None, None, None


def cat_loss(transition, logits, log_vt, log_v0, t, batch):
    """torch.mean(transition.compute_v_Lt(q_v_posterior(log_v0, log_vt), q_v_posterior(log_softmax(logits), log_vt), log_v0)) * 100"""
    return _CatLoss.apply(logits, transition.q_mats, transition.transpopse_q_onestep_mats, log_vt, log_v0, t, batch)


# ---- fused operators on the C++ fast path (csrc/mdx_fast.cpp): thin autograd shells --------------------------------------------------------
# Same kernels, same buffers, same queue entries and sink records as the Python bodies above; the C++ side fills the argument structs,
# allocates, launches, queues the weight gradients and runs the segment sums.  Parameter gradients never pass through autograd here (every
# parameter lives in the gradient sink: `all_in_sink`), so the shells return None for them.
def _fast_for(params):
    sk = _SINK
    F = sk['fast'] if sk is not None else None
    return F if (F is not None and F.all_in_sink(params)) else None


class _BondFfnScatterF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, bond_in, NL, GN, time, plan_in, plan_out, *params):
        ps = list(params)
        ctx.prec = _AMP
        r = F.bondffn_fwd(bond_in, NL, GN, time, plan_in.index, plan_out.order, plan_out.ptr, plan_out.n, ps)
        ctx.F, ctx.saved, ctx.ps, ctx.plan_in, ctx.plan_out = F, r[1:], ps, plan_in, plan_out
        t2 = time.detach()
        ctx.time2d = t2 if t2.dim() == 2 else t2.reshape(-1, 1)
        ctx.x_dtype = bond_in.dtype
        return r[0]

    @staticmethod
    def backward(ctx, gS):
        ni, pi = ctx.needs_input_grad, ctx.plan_in
        with precision(ctx.prec):
            g = ctx.F.bondffn_bwd(gS, ctx.saved, pi.index, pi.order, pi.ptr, pi.n, ctx.plan_out.index, ctx.time2d, ctx.ps, ni[1], ni[2], ni[3])
        gx = g[0]
        if gx is not None and gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        ctx.saved = None
        return (None, gx, g[1], g[2], None, None, None) + (None,) * len(ctx.ps)


class _EdgeTailF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, h, BL, BR, plan_l, plan_r, *params):
        ps = list(params)
        ctx.prec = _AMP
        r = F.edge_tail_fwd(h, BL, BR, plan_l.index, plan_r.index, ps)
        ctx.F, ctx.saved, ctx.ps, ctx.plan_l, ctx.plan_r, ctx.x_dtype = F, r[1:], ps, plan_l, plan_r, h.dtype
        return r[0]

    @staticmethod
    def backward(ctx, g_out):
        ni, pl, pr = ctx.needs_input_grad, ctx.plan_l, ctx.plan_r
        with precision(ctx.prec):
            g = ctx.F.edge_tail_bwd(g_out, ctx.saved, pl.index, pr.index, pl.order, pl.ptr, pr.order, pr.ptr, pl.n, ctx.ps, ni[1], ni[2], ni[3])
        gh = g[0]
        if gh is not None and gh.dtype != ctx.x_dtype:
            gh = gh.to(ctx.x_dtype)
        ctx.saved = None
        return (None, gh, g[1], g[2], None, None) + (None,) * len(ctx.ps)


class _PosFfnFrontF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, h_edge, LF, RF, time, plan_l, plan_r, *params):
        ps = list(params)
        ctx.prec = _AMP
        r = F.posffn_fwd(h_edge, LF, RF, time, plan_l.index, plan_r.index, ps)
        # saved for the backward: [x, LF, RF, te, a, gpre, gpost, prod, gate]
        ctx.F, ctx.saved, ctx.ps, ctx.plan_l, ctx.plan_r, ctx.x_dtype = F, r[2:] + [r[0], r[1]], ps, plan_l, plan_r, h_edge.dtype
        ctx.time2d = time.detach().reshape(-1, 1)
        return r[0], r[1]

    @staticmethod
    def backward(ctx, g_prod, g_gate):
        ni, pl, pr = ctx.needs_input_grad, ctx.plan_l, ctx.plan_r
        with precision(ctx.prec):
            g = ctx.F.posffn_bwd(g_prod, g_gate, ctx.saved, pl.index, pr.index, pl.order, pl.ptr, pr.order, pr.ptr, pl.n, ctx.time2d, ctx.ps,
                                 ni[1], ni[2], ni[3])
        gx = g[0]
        if gx is not None and gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        ctx.saved = None
        return (None, gx, g[1], g[2], None, None, None) + (None,) * len(ctx.ps)


class _NodeMsgF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, edge_attr, HN, PN, plan_col, plan_row, *params):
        ps = list(params)
        ctx.prec = _AMP
        r = F.nodemsg_fwd(edge_attr, HN, PN, plan_col.index, plan_row.order, plan_row.ptr, plan_row.n, ps)
        ctx.F, ctx.saved, ctx.ps, ctx.plan_col, ctx.plan_row, ctx.x_dtype = F, r[1:], ps, plan_col, plan_row, edge_attr.dtype
        return r[0]

    @staticmethod
    def backward(ctx, gA):
        ni, pc = ctx.needs_input_grad, ctx.plan_col
        with precision(ctx.prec):
            g = ctx.F.nodemsg_bwd(gA, ctx.saved, pc.index, pc.order, pc.ptr, pc.n, ctx.plan_row.index, ctx.ps, ni[1], ni[2], ni[3])
        gx = g[0]
        if gx is not None and gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        ctx.saved = None
        return (None, gx, g[1], g[2], None, None) + (None,) * len(ctx.ps)
