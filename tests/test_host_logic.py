"""CPU: host-side logic of the product package (no GPU compute calls) and the C-ABI surface."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import moldiff_amd as M
from moldiff_amd import _lib
from moldiff_amd import diffusion as D
from moldiff_amd.harness import default_config, placeholder_from_sizes
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- state_dict contract ---------------------------------------------------------------------------

def test_state_dict_keys_match_reference_contract():
    for kind, which, kn, ke in (('MolDiff', 'MolDiff', 8, 6), ('MolDiff_simple', 'MolDiff', 8, 6)):
        sd = M.MolDiff(default_config(kind), kn, ke).state_dict()
        ref = U.KEYS[which]
        assert set(sd) == set(ref)
        assert all(list(sd[k].shape) == ref[k] for k in ref)
    sd = M.BondPredictor(default_config('bondpred'), 8, 5).state_dict()
    ref = U.KEYS['BondPredictor']
    assert set(sd) == set(ref) and all(list(sd[k].shape) == ref[k] for k in ref)


def test_recipe_weights_hash():
    m = U.moldiff('MolDiff')
    h = hashlib.sha256()
    sd = m.state_dict()
    for k in sorted(sd):
        if not M.is_frozen_key(k):
            h.update(sd[k].numpy().tobytes())
    assert h.hexdigest() == U.KEYS['recipe_sha256_MolDiff']


def test_strict_load_rejects_missing_key():
    m = M.MolDiff(default_config('MolDiff'), 8, 6)
    sd = m.state_dict()
    sd.pop('denoiser.edge_embs.0.weight')
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd, strict=True)


def test_schedule_tables_match_reference_golden():
    g = U.gold('schedules.npz')
    for nm, kind in (('full', 'MolDiff'), ('simple', 'MolDiff_simple')):
        cfg = default_config(kind).diff
        for part, key in (('pos', 'diff_pos'), ('node', 'diff_atom'), ('edge', 'diff_bond')):
            kw = {k: v for k, v in cfg[key].items() if k != 'init_prob'}
            assert np.array_equal(D.get_beta_schedule(num_timesteps=1000, **kw), g[f'{nm}_{part}_betas'])
        m = M.MolDiff(default_config(kind), 8, 6)
        for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar'):
            assert np.array_equal(getattr(m.pos_transition, k).numpy(), g[f'{nm}_pos_{k}'])
        for part, tr in (('node', m.node_transition), ('edge', m.edge_transition)):
            for k in ('q_mats', 'transpopse_q_onestep_mats'):
                assert np.array_equal(getattr(tr, k)[g['probe_t']].numpy(), g[f'{nm}_{part}_{k}_probe'])
            assert np.array_equal(tr.init_prob, g[f'{nm}_{part}_init_prob'])


def test_smearing_buffers_match_reference_golden():
    g = U.gold('smearing.npz')
    m = M.MolDiff(default_config('MolDiff'), 8, 6)
    assert np.array_equal(m.denoiser.distance_expansion.offset.numpy(), g['d15_offset'])
    assert np.array_equal(m.denoiser.distance_expansion.coeff.numpy(), g['d15_coeff'])
    assert np.array_equal(m.time_emb[0].offset.numpy(), g['t10_offset'])
    b = M.BondPredictor(default_config('bondpred'), 8, 5)
    assert np.array_equal(b.encoder.distance_expansion.coeff.numpy(), g['d20_coeff'])
    assert np.array_equal(b.time_emb.coeff.numpy(), g['t20_coeff'])


def test_unknown_schedule_and_backbone_raise_like_reference():
    with pytest.raises(NotImplementedError):
        D.get_beta_schedule('nope', 10)
    cfg = default_config('MolDiff')
    cfg.denoiser.backbone = 'Other'
    with pytest.raises(NotImplementedError):
        M.MolDiff(cfg, 8, 6)


@pytest.mark.skipif(not os.path.exists('/root/reference/configs'), reason='reference tree not present (GPU box)')
def test_default_configs_equal_shipped_yaml():
    for name, path in (('MolDiff', 'train_MolDiff.yml'), ('MolDiff_simple', 'train_MolDiff_simple.yml'),
                       ('bondpred', 'train_bondpred.yml')):
        ref = M.load_config(f'/root/reference/configs/train/{path}').model
        assert dict(default_config(name)) == dict(ref)


# ---- placeholder -----------------------------------------------------------------------------------

def test_make_data_placeholder_matches_reference_golden():
    p = U.gold('placeholder.npz')
    for B in (8, 256, 2048):
        np.random.seed(2920)
        ph = M.make_data_placeholder(B)
        assert len(ph['batch_node']) == int(p[f'B{B}_N']) and len(ph['batch_halfedge']) == int(p[f'B{B}_Eh'])
        assert np.array_equal(ph['halfedge_index'][:, :16].numpy(), p[f'B{B}_he_first16'])
        assert np.array_equal(ph['halfedge_index'][:, -16:].numpy(), p[f'B{B}_he_last16'])
        assert np.array_equal(torch.bincount(ph['batch_node'], minlength=B).numpy(), np.maximum(p[f'B{B}_sizes'], 0))
    ph = M.make_data_placeholder(3, max_size=4)
    assert ph['halfedge_index'].shape == (2, 18) and ph['batch_halfedge'].tolist() == [0] * 6 + [1] * 6 + [2] * 6


def test_placeholder_edge_cases():
    ph = placeholder_from_sizes([0, 1, 2, -3])
    assert ph['batch_node'].tolist() == [1, 2, 2] and ph['halfedge_index'].tolist() == [[1], [2]]
    ph = placeholder_from_sizes([])
    assert ph['batch_node'].numel() == 0 and ph['halfedge_index'].shape == (2, 0)


# ---- C ABI surface ---------------------------------------------------------------------------------

def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'moldiff_hip.h')).read()
    declared = set(re.findall(r'\b(mdx_[a-z_0-9]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    L = _lib.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)
    assert L.mdx_version() >= 100


def test_model_create_rejects_unsupported_dims():
    cfg = _lib.MdxConfig(_lib.MDX_KIND_NET, 128, 64, 6, 15.0, 16, 1, 0, 1, 1, 1)
    h = ctypes.c_void_p()
    rc = _lib.lib().mdx_model_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == 4 and b'node_dim' in _lib.lib().mdx_last_error()


def test_finalize_reports_missing_parameter():
    cfg = _lib.MdxConfig(_lib.MDX_KIND_NET, 256, 64, 1, 15.0, 16, 1, 0, 1, 1, 1)
    h = ctypes.c_void_p()
    assert _lib.lib().mdx_model_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    rc = _lib.lib().mdx_model_finalize(h)
    assert rc == 3 and b'distance_expansion.offset' in _lib.lib().mdx_last_error()
    _lib.lib().mdx_model_destroy(h)


def _plan(ei, N):
    E = ei.shape[1]
    i32 = lambda n: np.zeros(n, dtype=np.int32)
    left, right, i2r, rp, cp, ce = i32(E), i32(E), i32(E), i32(N + 1), i32(N + 1), i32(E)
    ei = np.ascontiguousarray(ei, dtype=np.int64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = _lib.lib().mdx_graph_plan_host(N, E, P(ei), P(left), P(right), P(i2r), P(rp), P(cp), P(ce))
    return rc, left, right, i2r, rp, cp, ce


def test_graph_plan_host_invariants():
    bn, hei, bh, ei, be = U.graph_from_sizes([5, 0, 1, 7, 3])
    N, E = len(bn), ei.shape[1]
    rc, left, right, i2r, rp, cp, ce = _plan(ei.numpy(), N)
    assert rc == 0
    ein = ei.numpy()
    # permutation, sorted by (left, right), index arrays consistent with the reference order
    assert sorted(i2r.tolist()) == list(range(E))
    assert np.array_equal(left, ein[0][i2r]) and np.array_equal(right, ein[1][i2r])
    key = left.astype(np.int64) * N + right
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.diff(rp), np.bincount(ein[0], minlength=N))
    assert np.array_equal(np.diff(cp), np.bincount(ein[1], minlength=N))
    for v in range(N):
        assert np.all(left[rp[v]:rp[v + 1]] == v)
        ids = ce[cp[v]:cp[v + 1]]
        assert np.all(right[ids] == v) and np.all(np.diff(ids) > 0)


def test_graph_plan_rejects_out_of_range_and_handles_empty():
    rc, *_ = _plan(np.array([[0, 5], [1, 0]]), 3)
    assert rc == 1 and b'out of range' in _lib.lib().mdx_last_error()
    rc, left, right, i2r, rp, cp, ce = _plan(np.zeros((2, 0), dtype=np.int64), 4)
    assert rc == 0 and rp.tolist() == [0] * 5


def test_analytic_halfedge_index_formula():
    """pair (i<j) of a molecule of n atoms sits at i*n - i(i+1)/2 + (j-i-1)."""
    n = 9
    hei = placeholder_from_sizes([n])['halfedge_index'].numpy()
    for h, (i, j) in enumerate(hei.T):
        assert h == i * n - i * (i + 1) // 2 + (j - i - 1)


def test_cpu_tensor_calls_fail_loudly_without_gpu_compute():
    m = U.moldiff('MolDiff_simple')
    bn, hei, bh, ei, be = U.graph_from_sizes([4, 6])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.sample(2, bn, hei, bh)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.denoiser(torch.zeros(10, 256), torch.zeros(10, 3), torch.zeros(ei.shape[1], 64), ei, torch.zeros(10, 1),
                   torch.zeros(ei.shape[1], 1))


def test_product_package_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import moldiff_amd, moldiff_amd.distributed; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; "
            "assert not bad, bad" % ROOT)
    subprocess.run([sys.executable, '-c', code], check=True)
    for fn in os.listdir(os.path.join(ROOT, 'moldiff_amd')):
        if fn.endswith('.py'):
            src = open(os.path.join(ROOT, 'moldiff_amd', fn)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_plateau_scheduler_follows_torch_reduce_on_plateau():
    """trainer.PlateauScheduler vs torch.optim.lr_scheduler.ReduceLROnPlateau (the reference's 'plateau' scheduler,
    utils/train.py:75-82) on a noisy, flattening loss curve."""
    import types
    from moldiff_amd.trainer import PlateauScheduler
    g = np.random.Generator(np.random.PCG64(4))
    losses = [2.0 * np.exp(-i / 15.0) + 0.5 + 0.01 * g.standard_normal() for i in range(120)]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.6, patience=3, min_lr=1e-5)
    tr = types.SimpleNamespace(lr=1e-3)
    mine = PlateauScheduler(tr, factor=0.6, patience=3, min_lr=1e-5)
    for x in losses:
        ref.step(x)
        mine.step(x)
        assert abs(tr.lr - opt.param_groups[0]['lr']) < 1e-12
    assert tr.lr < 1e-3


def test_bench_workload_is_the_reference_recipe():
    """bench.py's batch = the reference's size recipe with seed 2920: first sizes, N, Eh at B = 8 / 256."""
    import bench
    _, ph8, s8 = bench.build_workload(8, 0, None)
    assert list(s8) == [24, 30, 28, 19, 21, 29, 26, 27]
    assert int(ph8['batch_node'].numel()) == 204 and int(ph8['batch_halfedge'].numel()) == 2552
    _, ph, s = bench.build_workload(256, 0, None)
    assert int(ph['batch_node'].numel()) == 6279 and int(ph['batch_halfedge'].numel()) == 77333
    assert int(s.min()) == 12 and int(s.max()) == 39
    _, ph1, s1 = bench.build_workload(256, 1, None)            # rank 1 gets the next 256 draws of the same stream
    np.random.seed(2920)
    allsz = np.random.normal(24.923464980477522, 5.516291901819105, size=512).astype('int64')
    assert np.array_equal(s, allsz[:256]) and np.array_equal(s1, allsz[256:])


def test_bench_multi_rank_workload_is_the_balanced_shard_of_one_job():
    """bench.py --gpus N: the job is 256 x N molecules (the first 256 x N draws of the recipe); the ranks take contiguous slices of the
    serpentine-by-size order like moldiff_amd.sample_drug3d does, so they partition the job, hold 256 molecules each and differ by
    < 0.5 % in directed edges (consecutive blocks of draws: 4.8 % at 8 ranks); one rank keeps the draws in order."""
    import bench
    s1, ids1 = bench.rank_molecules(256, 0, 1)
    assert np.array_equal(ids1, np.arange(256)) and int(s1.sum()) == 6279
    for world in (2, 8):
        np.random.seed(2920)
        allsz = np.random.normal(24.923464980477522, 5.516291901819105, size=256 * world).astype('int64')
        parts = [bench.rank_molecules(256, r, world) for r in range(world)]
        ids = np.concatenate([p[1] for p in parts])
        assert sorted(ids.tolist()) == list(range(256 * world))                       # a partition of the job
        edges = []
        for sz, idx in parts:
            assert len(sz) == 256 and np.array_equal(sz, allsz[idx])
            edges.append(int((sz * (sz - 1)).sum()))
        assert max(edges) <= 1.005 * (sum(edges) / world)
        _, ph, sz = bench.build_workload(256, 1, None, 'MolDiff_simple', world)
        assert np.array_equal(sz, parts[1][0]) and int(ph['batch_node'].numel()) == int(sz.sum())


def test_featurize_and_collate_match_the_restated_reference_pipeline():
    """moldiff_amd/data.py (PyG-free featurise + collate with __inc__ offsets + follow_batch vectors) against the oracle's
    restatement of utils/transforms.py:35-62 and of Batch.from_data_list with utils/data.py:25-33."""
    import numpy as np
    import torch
    from moldiff_amd.data import RecordLoader, collate, featurize, synthetic_records
    from moldiff_amd.postprocess import FeaturizeMol
    from oracle import moldiff_oracle as O
    feat = FeaturizeMol([6, 7, 8, 9, 15, 16, 17], [1, 2, 3, 4], use_mask_node=True, use_mask_edge=True)
    recs = synthetic_records(7, seed=3)

    class FixedRng:                      # the conformer draw, made explicit for both sides
        def __init__(self, v): self.v = v
        def integers(self, n): return self.v % n
    mols = [featurize(r, feat, FixedRng(i)) for i, r in enumerate(recs)]
    refs = [O.featurize_ref(r, feat.ele_to_nodetype, i % 3) for i, r in enumerate(recs)]
    for a, b in zip(mols, refs):
        for k in b:
            assert torch.equal(a[k], b[k]), k
        assert abs(float(a['node_pos'].mean())) < 1e-6
    got, want = collate(mols), O.collate_ref(refs)
    for k in ('node_type', 'node_pos', 'halfedge_type', 'halfedge_index', 'node_type_batch', 'halfedge_type_batch'):
        assert torch.equal(getattr(got, k), want[k]), k
    assert got.num_graphs == want['num_graphs'] == 7
    # the collated list is the packed fully-connected layout the kernels expect: same as the sampling placeholder
    from moldiff_amd.harness import placeholder_from_sizes
    ph = placeholder_from_sizes([int(m['node_type'].shape[0]) for m in mols])
    assert torch.equal(got.halfedge_index, ph['halfedge_index']) and torch.equal(got.node_type_batch, ph['batch_node'])
    # loader: endless, shuffled, every record once per epoch
    ld = RecordLoader(recs, feat, batch_size=3, seed=1)
    seen = [ld().num_graphs for _ in range(5)]
    assert seen == [3] * 5
    sizes = sorted(int(b.node_type.shape[0]) for b in RecordLoader(recs, feat, 4, shuffle=False).epoch_batches())
    assert sum(sizes) == sum(int(r['num_atoms']) for r in recs)
    with __import__('pytest').raises(AssertionError):
        featurize(dict(recs[0], element=np.array([5] * recs[0]['num_atoms'])), feat, FixedRng(0))


def test_config1_T100_schedule_tables_match_reference_golden():
    """BASELINE config #1 runs the simple model with 100 diffusion steps: every table the chain reads, for every t, equals
    what the real reference builds for num_timesteps = 100 (tests/golden/config1_T100.npz, oracle/make_goldens_config1.py)."""
    g = U.gold('config1_T100.npz')
    cfg = default_config('MolDiff_simple')
    cfg.diff['num_timesteps'] = int(g['T'])
    m = M.MolDiff(cfg, 8, 6)
    assert m.num_timesteps == 100
    for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar'):
        assert np.array_equal(getattr(m.pos_transition, k).numpy(), g['pos_' + k])
    for part, tr in (('node', m.node_transition), ('edge', m.edge_transition)):
        for k in ('q_mats', 'transpopse_q_onestep_mats'):
            assert np.array_equal(getattr(tr, k).numpy(), g[f'{part}_{k}'])


def test_lazy_one_hot_dtype_changes_and_masks_take_the_dense_path():
    """`.double()/.half()/.long()` must return a tensor of THAT dtype (values of the one-hot rows), and a bool mask that
    also covers the class dimension must not be applied to the ids."""
    from moldiff_amd.traj import LazyOneHot
    ids = torch.tensor([[0, 2, 1], [3, 3, 0]], dtype=torch.uint8)
    t = LazyOneHot(ids, 4)
    dense = F.one_hot(ids.long(), 4).float()
    assert isinstance(t[1], LazyOneHot) and isinstance(t[:, torch.tensor([True, False, True])], LazyOneHot)
    assert isinstance(t.to('cpu'), LazyOneHot) and isinstance(t.float(), LazyOneHot)
    for conv, dt in ((t.double(), torch.float64), (t.half(), torch.float16), (t.long(), torch.int64)):
        assert not isinstance(conv, LazyOneHot) and conv.dtype == dt and torch.equal(conv, dense.to(dt))
    full_mask = dense > 0.5                                  # (2,3,4): covers the class dimension too
    picked = t[full_mask]
    assert not isinstance(picked, LazyOneHot) and torch.equal(picked, dense[full_mask])
    row_mask = torch.tensor([[True, False, True], [False, True, False]])   # (2,3): frame x row -> still compact
    assert isinstance(t[row_mask], LazyOneHot) and torch.equal(t[row_mask].dense(), dense[row_mask])


def test_transposed_parameter_views_cover_weights_and_their_column_slices():
    """train_ops.TransposedParams (the grad_input GEMMs' W^T, transposed once per step on the device): pure bookkeeping over the flat
    parameter buffer -- a weight or a column slice of one maps to a row range of its transpose, everything else to None (the
    per-call transpose then serves it).  The device kernel is emulated with torch here."""
    import torch
    from moldiff_amd import train_ops as T
    from moldiff_amd.trainer import FlatParams
    net = torch.nn.Sequential(torch.nn.Linear(80, 64), torch.nn.LayerNorm(64), torch.nn.Linear(64, 256), torch.nn.Linear(256, 5))
    flat = FlatParams(net)
    tp = T.TransposedParams(flat)
    assert tp.n == 2                                              # the (5, 256) weight has rows of 5 floats: not 16-byte aligned
    for off, (o, R, C, do) in tp.entries.items():              # (round 6: every W^T starts on 16 bytes at its own offset `do`)
        tp.buf[do:do + R * C] = flat.data[o:o + R * C].view(R, C).t().contiguous().view(-1)
        assert (tp.buf.data_ptr() + 4 * do) % 16 == 0
    w = net[0].weight
    assert torch.equal(tp.view(w), w.t()) and tp.view(w).stride() == (64, 1)
    v = tp.view(w[:, 16:48])
    assert v is not None and torch.equal(v, w[:, 16:48].t()) and v.data_ptr() == tp.view(w).data_ptr() + 4 * 16 * 64
    assert tp.view(w[8:16]) is None                               # row slices are not views of the transpose
    assert tp.view(net[3].weight) is None and tp.view(net[1].weight) is None and tp.view(torch.zeros(64, 80)) is None
    w2 = net[2].weight
    v2 = tp.view(w2)
    assert v2 is not None and torch.equal(v2, w2.t())             # (aligned whatever its offset in the parameter buffer)


def test_class_range_assert_is_synchronous_on_the_api_and_recorded_inside_the_deferred_context():
    import pytest
    import torch
    from moldiff_amd import diffusion as D
    x = torch.tensor([0, 3, 7])
    assert D.index_to_log_onehot(x, 8).shape == (3, 8)
    with pytest.raises(AssertionError, match='8 >= 8'):
        D.index_to_log_onehot(torch.tensor([1, 8]), 9 - 1)
    with D.deferred_class_checks() as chk:      # CPU tensors are checked on the spot even inside the context (nothing to wait for)
        with pytest.raises(AssertionError):
            D.index_to_log_onehot(torch.tensor([9]), 8)
        assert chk.items == [] and chk.finish() is None
    assert D.index_to_log_onehot(torch.tensor([2]), 3, checked=False).argmax() == 2


def test_weight_gradient_row_ranges():
    from moldiff_amd import train_ops as T
    assert T._splits_for(154666, 256, 256, True) == 128           # 4 tiles of 128 x 128 -> ~512 workgroups (two per CU)
    assert T._splits_for(154666, 64, 64, True) == 512
    assert T._splits_for(154666, 64, 80, True) == T._splits_for(154666, 64, 80, False)   # 80 is not tile-aligned: converting kernel
    assert T._splits_for(300, 256, 256, True) == 2                # never fewer than 128 rows per range
    assert T._splits_for(100, 64, 64) == 1


def test_cpp_fast_path_extension_loads_and_keeps_its_state_machine_on_the_cpu():
    """csrc/mdx_fast.cpp (moldiff_amd/_mdx_fast.so): the extension imports without a GPU, exposes every entry point train_ops calls, and
    its sink bookkeeping (destination lookup, nesting guard, precision state) works on plain CPU tensors -- no launch is made here."""
    import torch
    from moldiff_amd import train_ops as T
    F = T._fast()
    assert F is not None
    for name in ('set_precision', 'set_options', 'sink_begin', 'sink_end', 'sink_active', 'fast_mode', 'sink_dst', 'sink_record', 'wq_append', 'flush',
                 'set_wt', 'launches', 'linear_fast_ok', 'linear', 'linear_ln_fast_ok', 'linear_ln_relu', 'linear_ln_dot_fast_ok',
                 'linear_ln_relu_dot', 'ew', 'all_in_sink', 'bondffn_fwd', 'bondffn_bwd', 'edge_tail_fwd', 'edge_tail_bwd', 'posffn_fwd',
                 'posffn_bwd', 'nodemsg_fwd', 'nodemsg_bwd'):
        assert hasattr(F, name), name
    data, grad = torch.zeros(64), torch.zeros(64)
    assert not F.sink_active() and F.sink_dst(data[8:16]) == 0
    F.sink_begin(data, grad)
    try:
        assert F.sink_active() and not F.fast_mode()                   # no float16 mode set: the fast nodes stay off
        assert F.sink_dst(data[8:16]) == grad.data_ptr() + 32 and F.sink_dst(torch.zeros(4)) == 0
        F.set_precision(2, 1, 1)
        assert F.fast_mode()
        import pytest
        with pytest.raises(RuntimeError, match='do not nest'):
            F.sink_begin(data, grad)
        w = torch.nn.Parameter(torch.zeros(4, 4))
        assert not F.linear_fast_ok(torch.zeros(3, 4), w, None)       # CPU tensors / parameters outside the sink never take the fast node
    finally:
        F.set_precision(0, 0, 0)
        F.sink_end()
    assert not F.sink_active()
