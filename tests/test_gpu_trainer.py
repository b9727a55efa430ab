"""Optimizer step and whole training steps on the GPU (reference: scripts/train_drug3d.py:88-109, utils/train.py:64-70)."""
import numpy as np
import pytest
import torch

from tests import util as U
from moldiff_amd import _lib
from moldiff_amd.trainer import Trainer, FlatParams

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_clip_and_adamw_kernels_match_torch_optim():
    g = U.rng(3)
    shapes = [(33, 7), (256,), (64, 129), (1,), (300, 300)]
    ps = [torch.nn.Parameter(U.t32(g.standard_normal(s)).to(DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mod = torch.nn.Module()
    mod.ps = torch.nn.ParameterList(ps)
    tr = Trainer(mod, lr=1e-2, betas=(0.99, 0.999), eps=1e-8, weight_decay=1e-3, max_grad_norm=5.0)
    opt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.99, 0.999), eps=1e-8, weight_decay=1e-3)
    for it in range(4):
        grads = [U.t32(g.standard_normal(s) * (30.0 if it % 2 == 0 else 0.01)).to(DEV) for s in shapes]   # clipped / not clipped
        tr.zero_grad()
        loss = sum((p * gr).sum() for p, gr in zip(ps, grads))
        gn = tr.backward_and_step(loss)
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        gn_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        opt.step()
        assert abs(float(gn) - float(gn_ref)) <= 1e-5 * float(gn_ref)
        for p, r in zip(ps, ref):
            assert U.maxdiff(p, r) <= 2e-6 * max(1.0, float(r.abs().max()))


def test_flat_views_alias_the_parameters():
    m = torch.nn.Linear(5, 3).to(DEV)
    f = FlatParams(m)
    assert f.numel == 18 and m.weight.data_ptr() == f.data.data_ptr()
    m.weight.grad.fill_(2.0)
    assert float(f.grad[:15].sum()) == 30.0
    f.zero_grad()
    assert float(m.weight.grad.abs().sum()) == 0.0


def _tiny_batch(seed, sizes=(6, 9, 5, 11)):
    g = U.rng(seed)
    bn, hei, bh, _, _ = U.graph_from_sizes(list(sizes), DEV)
    N, Eh = len(bn), len(bh)
    pos = U.t32(g.standard_normal((N, 3)) * 1.5).to(DEV)
    return (torch.from_numpy(g.integers(0, 7, N)).to(DEV), pos, bn,
            torch.from_numpy((g.random(Eh) < 0.3) * g.integers(1, 5, Eh)).to(DEV), hei, bh, len(sizes))


@pytest.mark.parametrize('which', ['moldiff', 'bondpred'])
def test_training_steps_reduce_the_loss_and_refresh_the_fused_engine(which):
    import copy
    base = U.moldiff('MolDiff_simple', DEV) if which == 'moldiff' else U.bondpred(DEV)
    m = copy.deepcopy(base)
    for mod in m.modules():                 # deepcopy must not share the cached engine with the fixture model
        if hasattr(mod, '_eng'):
            mod._eng, mod._eng_sig = None, None
    tr = Trainer(m, lr=2e-4, max_grad_norm=50.0)
    batch = _tiny_batch(7)
    t = torch.tensor([120, 480, 700, 930], device=DEV)
    g = U.rng(8)
    N, Eh = batch[1].shape[0], batch[3].shape[0]
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV),
                 u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
    if which == 'bondpred':
        noise.pop('u_halfedge')
    with torch.no_grad():
        before = float(m.get_loss(*batch, time_step=t, noise=noise)['loss'])
    first = None
    for it in range(12):
        out = tr.step(*batch, time_step=t, noise=noise)
        first = float(out['loss']) if first is None else first
        assert torch.isfinite(out['loss']) and torch.isfinite(out['grad_norm'])
    assert abs(first - before) <= 2e-5 * max(1.0, abs(before))          # train-mode forward == fused forward
    with torch.no_grad():
        after_fused = float(m.get_loss(*batch, time_step=t, noise=noise)['loss'])    # fused engine sees the NEW weights
    after_train = float(m.get_loss(*batch, time_step=t, noise=noise)['loss'])
    assert after_fused < before * 0.98
    assert abs(after_fused - after_train) <= 2e-5 * max(1.0, abs(after_train))


@pytest.mark.parametrize('fused_rows', [1024, 1])
def test_fp16_autocast_training_steps_reduce_the_loss(fused_rows):
    """the float16 autocast mode (the reference's use_amp: True) is shown to DESCEND, not only to match one step's
    gradients: fixed time steps, fixed noise, 12 optimisation steps with the loss scaler; the loss -- evaluated before and after by the
    fp32 sampling kernels under no_grad -- falls by more than 2 %, and every step's loss in the float16 arithmetic is within 1 % of the
    fp32 evaluation of the same weights.  fused_rows = 1: with the fused EdgeBlock kernels of round 6 forced on at this size."""
    import copy
    from moldiff_amd import train_ops
    m = copy.deepcopy(U.moldiff('MolDiff_simple', DEV))
    for mod in m.modules():
        if hasattr(mod, '_eng'):
            mod._eng, mod._eng_sig = None, None
    # lr = the shipped configs' 1e-4 (configs/train_MolDiff*.yml).  At 2e-4 this 4-molecule batch is not a descent any more but an
    # oscillation (AdamW's first steps are sign-like): 12 steps end above or below the start depending on float16 rounding of single
    # gradients (tools/probe_catloss_descent.py: both loss-tail variants oscillate at 2e-4, both fall 1.66 -> 1.10-1.13 at 1e-4)
    tr = Trainer(m, lr=1e-4, max_grad_norm=50.0, precision='fp16', init_scale=1024.0)
    batch = _tiny_batch(7)
    t = torch.tensor([120, 480, 700, 930], device=DEV)
    g = U.rng(8)
    N, Eh = batch[1].shape[0], batch[3].shape[0]
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV),
                 u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
    with torch.no_grad():
        before = float(m.get_loss(*batch, time_step=t, noise=noise)['loss'])
    old = train_ops.FUSED_MIN_ROWS
    train_ops.FUSED_MIN_ROWS = fused_rows
    try:
        losses = []
        for it in range(12):
            out = tr.step(*batch, time_step=t, noise=noise)
            assert torch.isfinite(out['loss']) and torch.isfinite(out['grad_norm'])
            losses.append(float(out['loss']))
    finally:
        train_ops.FUSED_MIN_ROWS = old
    tr.check_deferred()
    assert abs(losses[0] - before) <= 1e-2 * max(1.0, abs(before)), (losses[0], before)
    with torch.no_grad():
        after = float(m.get_loss(*batch, time_step=t, noise=noise)['loss'])
    assert int(tr.state[3].item()) == 0                     # no step was skipped by the loss scaler
    assert after < before * 0.98, (before, after, losses)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize('use_amp', [False, True])
def test_train_entry_point_end_to_end(tmp_path, use_amp):
    """python -m moldiff_amd.train_drug3d on a cut-down config: trains, validates, writes a checkpoint that loads strictly.
    use_amp=False (fp32): every iteration is an optimizer step.  use_amp=True (the shipped configs, like the reference's): float16
    arithmetic with GradScaler semantics -- the scale starts at 65536 like torch's, so with recipe weights (gradient norms of O(10))
    the first iterations overflow float16, are SKIPPED and halve the scale, exactly what the reference's scaler does in its first
    iterations; steps + skipped = iterations, and the scale ends at 65536 / 2^skipped."""
    import yaml
    from moldiff_amd import train_drug3d, MolDiff
    cfg = yaml.safe_load(open('configs/train_MolDiff_simple.yml'))
    iters = 12 if use_amp else 4
    cfg['train'].update(batch_size=6, max_iters=iters, val_freq=iters // 2, use_amp=use_amp)
    p = tmp_path / 'cfg.yml'
    p.write_text(yaml.safe_dump(cfg))
    assert train_drug3d.main(['--config', str(p), '--device', DEV, '--logdir', str(tmp_path / 'logs'), '--val_batches', '1',
                              '--recipe-weights']) == 0
    ck = torch.load(tmp_path / 'logs' / 'checkpoints' / f'{iters}.pt', map_location='cpu', weights_only=False)
    st = ck['optimizer']['amp_state']
    steps, skipped, scale = int(st[2]), int(st[3]), float(st[0])
    assert ck['iteration'] == iters and ck['optimizer']['steps'] == steps and steps + skipped == iters
    if use_amp:
        assert 0 < skipped < iters and steps > 0 and scale == 65536.0 / 2 ** skipped
    else:
        assert skipped == 0 and scale == 1.0
    m = MolDiff(ck['config'].model, 8, 6)
    m.load_state_dict(ck['model'], strict=True)
    assert all(torch.isfinite(v).all() for v in ck['model'].values() if v.is_floating_point())
    ref = U.moldiff('MolDiff_simple')
    assert any(not torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))    # weights moved


def test_train_entry_point_on_processed_records(tmp_path):
    """The PyG-free data side (moldiff_amd/data.py): records -> FeaturizeMol.__call__ -> collate with __inc__ offsets ->
    get_loss/backward/AdamW, from a torch.save'd record file like an export of the reference's LMDB."""
    import yaml
    from moldiff_amd import train_drug3d
    from moldiff_amd.data import synthetic_records
    recs = synthetic_records(24, seed=9)
    torch.save({'train': recs[:20], 'val': recs[20:]}, tmp_path / 'records.pt')
    cfg = yaml.safe_load(open('configs/train_MolDiff_simple.yml'))
    cfg['train'].update(batch_size=5, max_iters=3, val_freq=3)
    cfg['dataset'] = {'name': 'records', 'path': str(tmp_path / 'records.pt')}
    p = tmp_path / 'cfg.yml'
    p.write_text(yaml.safe_dump(cfg))
    assert train_drug3d.main(['--config', str(p), '--device', DEV, '--logdir', str(tmp_path / 'logs'), '--val_batches', '2',
                              '--recipe-weights']) == 0
    ck = torch.load(tmp_path / 'logs' / 'checkpoints' / '3.pt', map_location='cpu', weights_only=False)
    assert ck['iteration'] == 3 and all(torch.isfinite(v).all() for v in ck['model'].values() if v.is_floating_point())


def _dp_rank(rank, world, port, q):
    """Two data-parallel ranks sharing the one GPU of the test box (gloo carries the all-reduce; on a real node it is
    RCCL, same code path): different batches per rank, identical parameters on every rank after every step."""
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = U.moldiff('MolDiff_simple', DEV).train()
    tr = Trainer(m, lr=3e-4)
    norms = []
    for it in range(3):
        batch = _tiny_batch(100 + 10 * it + rank, sizes=(5 + rank, 8, 6))
        torch.manual_seed(50 + it + 7 * rank)
        out = tr.step(*batch)
        norms.append(float(out['grad_norm']))
    flat = tr.flat.data.detach().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, norms, bool(all(torch.equal(gathered[0], g) for g in gathered)), float(flat.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_data_parallel_ranks_stay_in_lockstep():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] and res[1][2]                       # bit-identical parameters across ranks
    assert res[0][1] == res[1][1]                        # same (averaged) gradient norm seen by both
    assert res[0][3] == res[1][3]


def test_checkpoint_written_by_training_is_sampled_by_the_sampling_entry_point(tmp_path):
    """train_drug3d -> checkpoints/<it>.pt -> sample_drug3d --config (model.checkpoint = that file): the loop closes."""
    import yaml
    from moldiff_amd import sample_drug3d, train_drug3d
    cfg = yaml.safe_load(open('configs/train_MolDiff_simple.yml'))
    cfg['train'].update(batch_size=4, max_iters=2, val_freq=2)
    tp = tmp_path / 'train.yml'
    tp.write_text(yaml.safe_dump(cfg))
    assert train_drug3d.main(['--config', str(tp), '--device', DEV, '--logdir', str(tmp_path / 'logs'), '--val_batches', '1',
                              '--recipe-weights']) == 0
    scfg = yaml.safe_load(open('configs/sample_MolDiff_simple.yml'))
    scfg['model']['checkpoint'] = str(tmp_path / 'logs' / 'checkpoints' / '2.pt')
    scfg['sample'].update(num_mols=3, batch_size=3)
    sp = tmp_path / 'sample.yml'
    sp.write_text(yaml.safe_dump(scfg))
    log_dir = sample_drug3d.main(['--config', str(sp), '--outdir', str(tmp_path / 'out'), '--device', DEV, '--batch_size', '3'])
    pool = torch.load(str(log_dir) + '/samples_all.pt', weights_only=False)
    assert len(pool['finished']) + len(pool['failed']) >= 3


def _toy(shapes, seed=11):
    g = U.rng(seed)
    ps = [torch.nn.Parameter(U.t32(g.standard_normal(s)).to(DEV)) for s in shapes]
    mod = torch.nn.Module()
    mod.ps = torch.nn.ParameterList(ps)
    return g, ps, mod


def test_non_finite_gradient_skips_the_step_and_does_not_advance_adam(  ):
    """a skipped update must not advance the Adam bias-correction step (torch's GradScaler does not call optimizer.step
    on found_inf).  Step 1 finite, step 2 carries a NaN gradient (skipped: parameters, moments and the step count untouched, the norm
    reported as non-finite), step 3 finite: identical to torch.optim.AdamW that only saw steps 1 and 3."""
    shapes = [(17, 5), (64,), (33, 33)]
    g, ps, mod = _toy(shapes)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    tr = Trainer(mod, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, max_grad_norm=5.0)
    opt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    for it in range(3):
        grads = [U.t32(g.standard_normal(s)).to(DEV) for s in shapes]
        if it == 1:
            grads[1][7] = float('nan')
        tr.zero_grad()
        before = [p.detach().clone() for p in ps]
        gn = tr.backward_and_step(sum((p * gr).sum() for p, gr in zip(ps, grads)))
        if it == 1:
            assert not torch.isfinite(gn)
            assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before))
            continue
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 5.0)
        opt.step()
    assert tr.steps == 2 and tr.skipped == 1
    for p, r in zip(ps, ref):
        assert U.maxdiff(p, r) <= 2e-6 * max(1.0, float(r.abs().max()))


def test_fp16_loss_scale_follows_grad_scaler():
    """precision='fp16' carries torch.cuda.amp.GradScaler's state machine on the device: the gradient the optimizer sees is
    unscaled (the update equals the fp32 update), `growth_interval` consecutive finite steps double the scale, an overflow halves it,
    skips the step and restarts the count (scripts/train_drug3d.py:105-109)."""
    shapes = [(9, 4), (30,)]
    g, ps, mod = _toy(shapes, 12)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    tr = Trainer(mod, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0, max_grad_norm=None, precision='fp16', init_scale=1024.0,
                 growth_interval=2)
    opt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0)
    scales = [tr.loss_scale]
    for it in range(5):
        grads = [U.t32(g.standard_normal(s)).to(DEV) for s in shapes]
        if it == 3:
            grads[0][0, 0] = float('inf')
        tr.zero_grad()
        gn = tr.backward_and_step(sum((p * gr).sum() for p, gr in zip(ps, grads)))
        scales.append(tr.loss_scale)
        if it == 3:
            assert not torch.isfinite(gn)
            continue
        want = float(torch.cat([x.flatten() for x in grads]).double().norm())
        assert abs(float(gn) - want) <= 1e-5 * want                    # the norm of the UNSCALED gradient
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
    #            init    it0     it1 (2 clean: x2)  it2    it3 (overflow: /2)  it4
    assert scales == [1024.0, 1024.0, 2048.0, 2048.0, 1024.0, 1024.0], scales
    assert tr.steps == 4 and tr.skipped == 1
    for p, r in zip(ps, ref):
        assert U.maxdiff(p, r) <= 2e-6 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize('sizes', [(7, 12, 5, 9, 16), (30, 40, 35, 45, 22)], ids=['small', 'two_stage'])
@pytest.mark.parametrize('precision', ['f32', 'fp16'])
def test_deferred_gradient_sink_equals_autograd_accumulation(precision, sizes):
    """Trainer.step routes every weight / bias / LayerNorm gradient of the HIP operators around autograd: the split partials are
    summed into the flat gradient buffer by ONE launch (train_ops.grad_sink / mdx_op_reduce_deferred).  Same fixed summation order as
    the per-layer reductions, so the flat gradient must equal what plain `loss.backward()` accumulates -- bit for bit in fp32.  'two_stage': 6,000 edge rows, so
    the edge LayerNorms leave more than 256 partial rows and both routes take their two-stage form (chunk sums, then their sum)."""
    import copy
    from moldiff_amd import train_ops
    base = U.moldiff('MolDiff', DEV)
    m = copy.deepcopy(base)
    for mod in m.modules():
        if hasattr(mod, '_eng'):
            mod._eng, mod._eng_sig = None, None
    tr = Trainer(m, lr=0.0, max_grad_norm=None, precision=precision, init_scale=1.0)   # lr 0: the weights stay put
    batch = _tiny_batch(21, sizes=sizes)
    t = torch.tensor([5, 310, 640, 880, 999], device=DEV)
    g = U.rng(22)
    N, Eh = batch[1].shape[0], batch[3].shape[0]
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV),
                 u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
    sunk = {}
    old = train_ops._WG_ON
    try:
        for queued in (False, True):                  # round 6: with and without the weight-gradient queue (float16 mode only)
            train_ops._WG_ON = queued
            tr.step(*batch, time_step=t, noise=noise)
            sunk[queued] = tr.flat.grad.clone()
    finally:
        train_ops._WG_ON = old
    tr.zero_grad()
    with train_ops.precision(precision):
        loss = m.get_loss(*batch, time_step=t, noise=noise)['loss']
    loss.backward()                                   # no sink: autograd's own accumulation
    plain = tr.flat.grad.clone()
    assert float(plain.abs().max()) > 0
    if precision == 'f32':
        assert torch.equal(sunk[False], plain) and torch.equal(sunk[True], plain)
    else:
        assert float((sunk[False] - plain).abs().max()) <= 1e-6 * float(plain.abs().max())
        # the queue cuts the rows into its own ranges: fp32 partial sums in another order, then the SAME float16 rounding of each
        # element's sum -- at most one float16 ulp (2^-10 relative; absolute 2^-24 x scale for elements near zero) apart
        d = (sunk[True] - plain).abs()
        assert bool((d <= 2.0 ** -10 * plain.abs() + 2.0 ** -20 * float(plain.abs().max())).all())
        assert float((d > 0).float().mean()) < 0.05


def test_batched_weight_transposes_are_exact_views():
    """Trainer.step transposes every weight matrix once per step with ONE launch (mdx_op_transpose_batch) and the grad_input GEMMs
    take W^T -- of a whole weight or of a column slice of one (the hoisted layers split their weights by columns) -- as a view into
    that buffer.  Every view must equal the transpose torch computes, after a step that changed the weights."""
    import copy
    m = copy.deepcopy(U.moldiff('MolDiff', DEV))
    tr = Trainer(m, lr=1e-3, precision='fp16', init_scale=1.0)
    batch = _tiny_batch(31, sizes=(6, 11, 8))
    tr.step(*batch)
    tr.step(*batch)
    tr.wt.refresh()
    torch.cuda.synchronize()
    assert tr.wt.n > 100
    seen = 0
    for p in tr.flat.params:
        if p.dim() != 2:
            continue
        v = tr.wt.view(p.data)
        if p.shape[0] % 4:
            assert v is None
            continue
        if v is None:      # unaligned offset inside the flat buffer: the per-call transpose serves it
            continue
        assert torch.equal(v, p.data.t())
        if p.shape[1] >= 16:
            s = p.data[:, 4:12]
            vs = tr.wt.view(s)
            assert vs is not None and torch.equal(vs, s.t())
        seen += 1
    assert seen > 100


def test_class_range_assert_is_postponed_not_dropped():
    """models/diffusion.py:54 asserts the class range of the clean batch (a host-device synchronisation).  model.get_loss keeps it
    where the reference has it; Trainer.step keeps the maxima on the device, copies them asynchronously and raises the same
    AssertionError one call later (next step or check_deferred) -- the host no longer waits for the GPU at the top of every step.
    (The out-of-range id itself is never fed to F.one_hot on the device here: that would be a device-side fault.)"""
    import copy
    from moldiff_amd import diffusion as D
    m = copy.deepcopy(U.moldiff('MolDiff', DEV))
    batch = list(_tiny_batch(33, sizes=(6, 9)))
    bad = list(batch)
    bad[0] = batch[0].clone()
    bad[0][3] = 8                               # node classes are 0..7
    with pytest.raises(AssertionError, match='8 >= 8'):
        m.get_loss(*bad)                         # synchronous, before anything touches the id
    tr = Trainer(m, lr=0.0, precision='f32')
    tr.step(*batch)
    assert tr._deferred is not None              # node and bond classes of the batch were recorded ...
    tr.check_deferred()                          # ... and are in range
    assert tr._deferred is None
    with D.deferred_class_checks() as chk:       # the postponed form of the assert, on values that are out of range
        ok = D.index_to_log_onehot(batch[0], 8)
        chk.items.append((bad[0].max(), 8))
    assert len(chk.items) == 2 and torch.equal(ok, D.index_to_log_onehot(batch[0], 8))
    verify = chk.finish()
    with pytest.raises(AssertionError, match='8 >= 8'):
        verify()


@pytest.mark.parametrize('rows', [64, 2048])
def test_queued_weight_gradients_equal_the_per_call_launches(rows):
    """Round 6: inside a gradient sink the float16 mode queues every weight-gradient contraction and runs the queue as one launch per
    tile class (train_ops._flush_wgrads, csrc wgrad_grouped_kernel).  The flat gradient buffer must be what the per-call launches
    produce: same contraction bodies, fp32 partials; only the row ranges differ (rows per block instead of ~512 blocks per call), so
    the two agree to fp32 summation-order rounding before the final float16 rounding of each sum (<= 1 float16 ulp on an element)."""
    import copy
    from moldiff_amd import train_ops
    base = U.moldiff('MolDiff_simple', DEV)
    batch = _tiny_batch(11, sizes=(9, 14, 7, 12, 10))
    t = torch.tensor([100, 300, 500, 700, 900], device=DEV)
    g = U.rng(12)
    N, Eh = batch[1].shape[0], batch[3].shape[0]
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV),
                 u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
    grads = {}
    old = (train_ops._WG_ON, train_ops.WGRAD_ROWS, train_ops.FUSED_MIN_ROWS)
    try:
        train_ops.FUSED_MIN_ROWS = 1
        for on in (False, True):
            train_ops._WG_ON, train_ops.WGRAD_ROWS = on, rows
            m = copy.deepcopy(base)
            for mod in m.modules():
                if hasattr(mod, '_eng'):
                    mod._eng, mod._eng_sig = None, None
            tr = Trainer(m, lr=0.0, max_grad_norm=None, precision='fp16', init_scale=256.0)
            tr.step(*batch, time_step=t, noise=noise)
            grads[on] = tr.flat.grad.clone()
    finally:
        train_ops._WG_ON, train_ops.WGRAD_ROWS, train_ops.FUSED_MIN_ROWS = old
    a, b = grads[False], grads[True]
    assert torch.isfinite(a).all() and torch.isfinite(b).all() and float(a.abs().max()) > 0
    # float16 rounding of each sum: one ulp = 2^-10 relative; elements near zero compare absolutely against the buffer's scale
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 2.0 ** -9 * scale
    assert float((a - b).norm()) <= 2e-4 * float(a.norm())
    assert float((a != b).float().mean()) < 0.05       # almost every element is bit-identical


def test_fused_categorical_loss_equals_the_torch_tail():
    """Round 6: train_ops.cat_loss (csrc cat_loss_kernel: log_softmax, both posteriors, KL / NLL rows and the hand-written backward in
    one launch) against the layer-by-layer torch evaluation it replaces (models/model.py:170-189 through
    transition.q_v_posterior_autograd + compute_v_Lt and torch.autograd): value and d/d logits, with rows at t == 0, at t == 1 and
    with confident logits (the clamp at -32 / eps = 1e-30 branches)."""
    import torch.nn.functional as F
    from moldiff_amd import train_ops
    from moldiff_amd.diffusion import index_to_log_onehot
    m = U.moldiff('MolDiff_simple', DEV)
    g = U.rng(5)
    B = 40
    t = torch.from_numpy(g.integers(0, 1000, B)).to(DEV)
    t[:6] = torch.tensor([0, 0, 1, 1, 999, 2], device=DEV)
    for tr, n in ((m.node_transition, 1500), (m.edge_transition, 4000)):
        K = tr.num_classes
        batch = torch.from_numpy(np.sort(g.integers(0, B, n))).to(DEV)
        v = torch.from_numpy(g.integers(0, K, n)).to(DEV)
        scale = torch.from_numpy(g.choice([0.3, 3.0, 40.0], n)).float().to(DEV).unsqueeze(-1)
        logits = (U.t32(g.standard_normal((n, K))).to(DEV) * scale).requires_grad_(True)
        with torch.no_grad():
            log_v0 = index_to_log_onehot(v, K, checked=False)
            _, log_vt = tr.q_vt_sample(log_v0, t, batch, U.t32(g.random((n, K))).to(DEV))
        log_recon = F.log_softmax(logits, dim=-1)
        post_true = tr.q_v_posterior(log_v0, log_vt, t, batch, v0_prob=True)
        post_pred = tr.q_v_posterior_autograd(log_recon, log_vt, t, batch)
        ref = torch.mean(tr.compute_v_Lt(post_true, post_pred, log_v0, t=t, batch=batch)) * 100
        (g_ref,) = torch.autograd.grad(ref * 3.0, logits)
        logits2 = logits.detach().clone().requires_grad_(True)
        got = train_ops.cat_loss(tr, logits2, log_vt, log_v0, t, batch)
        (g_got,) = torch.autograd.grad(got * 3.0, logits2)
        assert torch.isfinite(got) and torch.isfinite(g_got).all()
        assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (float(got), float(ref))
        gs = float(g_ref.abs().max())
        assert gs > 0 and float((g_got - g_ref).abs().max()) <= 2e-5 * gs, (float((g_got - g_ref).abs().max()), gs)
        assert float((g_got - g_ref).norm()) <= 2e-6 * float(g_ref.norm())


@pytest.mark.parametrize('fused_rows', [1024, 1])
def test_cpp_fast_path_is_bit_identical_to_the_python_bodies(fused_rows):
    """Round 6: csrc/mdx_fast.cpp (moldiff_amd/_mdx_fast.so) holds the training operators' host logic in C++ -- gradient sink, weight-
    gradient queue, Linear / Linear+LayerNorm / element-wise autograd nodes, the fused operators' bodies.  Same kernels, same buffers,
    same row ranges: the loss, the flat gradient buffer and the updated weights of two optimisation steps must be BIT-identical with
    MDX_TRAIN_FAST on and off (fused_rows = 1: with the fused row-owner kernels forced on at this size, so their C++ bodies run)."""
    import copy
    from moldiff_amd import train_ops
    base = U.moldiff('MolDiff', DEV)
    batch = _tiny_batch(41, sizes=(9, 14, 7, 12, 10, 6))
    t = torch.tensor([0, 100, 300, 500, 700, 999], device=DEV)
    g = U.rng(42)
    N, Eh = batch[1].shape[0], batch[3].shape[0]
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV),
                 u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
    res = {}
    old = (train_ops._FAST_ON, train_ops.FUSED_MIN_ROWS)
    try:
        train_ops.FUSED_MIN_ROWS = fused_rows
        for fast in (False, True):
            train_ops._FAST_ON = fast
            m = copy.deepcopy(base)
            for mod in m.modules():
                if hasattr(mod, '_eng'):
                    mod._eng, mod._eng_sig = None, None
            tr = Trainer(m, lr=1e-4, max_grad_norm=50.0, precision='fp16', init_scale=256.0)
            n0 = train_ops._fast().launches() if fast else 0
            o1 = tr.step(*batch, time_step=t, noise=noise)
            g1 = tr.flat.grad.clone()
            o2 = tr.step(*batch, time_step=t, noise=noise)
            res[fast] = (float(o1['loss']), float(o2['loss']), g1, tr.flat.grad.clone(), tr.flat.data.clone(), float(o2['grad_norm']))
            if fast:
                assert train_ops._fast().launches() - n0 > 100        # the C++ bodies did run
    finally:
        train_ops._FAST_ON, train_ops.FUSED_MIN_ROWS = old
    a, b = res[False], res[True]
    assert a[0] == b[0] and a[1] == b[1] and a[5] == b[5], (a[0], b[0], a[1], b[1])
    assert float(a[2].abs().max()) > 0
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_fused_categorical_add_noise_equals_the_torch_composition():
    """Round 6: GeneralCategoricalTransition.add_noise as one launch (csrc cat_add_noise_kernel) against the operator-by-operator
    evaluation it replaces (index_to_log_onehot, q_vt_pred, Gumbel-max, one-hot; models/transition.py:266-283): the drawn classes are
    identical for injected uniforms (apart from draws whose two best Gumbel scores are closer than 1e-5: none expected), and
    log_vt / log_v0 / the one-hot tensors are bit-identical given the classes."""
    from moldiff_amd import transition as TR
    m = U.moldiff('MolDiff_simple', DEV)
    g = U.rng(9)
    B = 37
    t = torch.from_numpy(g.integers(0, 1000, B)).to(DEV)
    t[:4] = torch.tensor([0, 1, 999, 500], device=DEV)
    for tr, n in ((m.node_transition, 3000), (m.edge_transition, 9000)):
        K = tr.num_classes
        batch = torch.from_numpy(np.sort(g.integers(0, B, n))).to(DEV)
        v = torch.from_numpy(g.integers(0, K, n)).to(DEV)
        u = U.t32(g.random((n, K))).to(DEV)
        old = TR._FUSED_NOISE
        try:
            TR._FUSED_NOISE = False
            oh0, lvt0, lv00 = tr.add_noise(v, t, batch, u)
            TR._FUSED_NOISE = True
            oh1, lvt1, lv01 = tr.add_noise(v, t, batch, u)
        finally:
            TR._FUSED_NOISE = old
        assert torch.equal(lv00, lv01)
        same = (oh0.argmax(-1) == oh1.argmax(-1))
        assert float((~same).float().mean()) <= 1e-3, float((~same).float().mean())
        assert torch.equal(oh0[same], oh1[same]) and torch.equal(lvt0[same], lvt1[same])
        assert bool((oh1.sum(-1) == 1).all())
