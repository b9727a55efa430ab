"""CPU: the N>1 path (shard function + end-of-run gather) with world_size-2 gloo processes."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from moldiff_amd import distributed as DD
from moldiff_amd.harness import placeholder_from_sizes


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 256, 2049):
        for w in (1, 2, 3, 8):
            spans = [DD.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_balanced_order_is_a_permutation_and_balances_cost():
    np.random.seed(2920)
    sizes = np.random.normal(24.92, 5.52, 2048).astype('int64')
    order = DD.balanced_order(sizes, 8)
    assert sorted(order.tolist()) == list(range(2048))
    cost = [(sizes[order][slice(*DD.shard_bounds(2048, 8, r))] ** 2).sum() for r in range(8)]
    naive = [(sizes[slice(*DD.shard_bounds(2048, 8, r))] ** 2).sum() for r in range(8)]
    assert max(cost) / min(cost) < 1.01 <= max(naive) / min(naive) + 1.0


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    my_sizes, ids = DD.shard_molecules(sizes, world, rank)
    ph = placeholder_from_sizes(my_sizes)
    N, Eh = len(ph['batch_node']), len(ph['batch_halfedge'])
    # stand-in for the per-rank sampler output: values that encode the GLOBAL molecule id, so the test can check
    # that the gathered tensors are in global molecule order
    gid = torch.from_numpy(ids)[ph['batch_node']].float()
    ghe = torch.from_numpy(ids)[ph['batch_halfedge']].float()
    pred = [gid[:, None].repeat(1, 8), gid[:, None].repeat(1, 3) + 0.5, ghe[:, None].repeat(1, 6)]
    out = DD.gather_pred(pred, dst=0)
    if rank == 0:
        q.put([o.numpy() for o in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_global_order():
    sizes = [5, 9, 3, 0, 12, 7, 4]
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = placeholder_from_sizes(sizes)
    ids = np.arange(len(sizes))
    assert out[0].shape == (len(full['batch_node']), 8)
    assert np.array_equal(out[0][:, 0], ids[full['batch_node'].numpy()].astype(np.float32))
    assert np.array_equal(out[1][:, 0], ids[full['batch_node'].numpy()].astype(np.float32) + 0.5)
    assert out[2].shape == (len(full['batch_halfedge']), 6)
    assert np.array_equal(out[2][:, 0], ids[full['batch_halfedge'].numpy()].astype(np.float32))


# ---- data-parallel training: flat gradient buffer averaged over ranks (moldiff_amd/trainer.py) -----------------------
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from moldiff_amd.trainer import FlatParams, allreduce_mean_
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)                       # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 2))
    flat = FlatParams(net)
    x = torch.arange(12, dtype=torch.float32).reshape(2, 6) * (rank + 1)       # a different shard per rank
    flat.zero_grad()
    net(x).square().sum().backward()           # autograd accumulates straight into the flat buffer
    local = flat.grad.clone()
    w = allreduce_mean_(flat.grad)
    q.put((rank, w, local.numpy(), flat.grad.numpy().copy(), [p.grad.data_ptr() - flat.grad.data_ptr() for p in flat.params]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_over_the_flat_buffer():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, l0, a0, off0), (_, w1, l1, a1, off1) = res
    assert w0 == w1 == 2
    assert np.abs(l0 - l1).max() > 0                                   # the shards really differ
    assert np.array_equal(a0, a1)                                      # every rank ends with the same gradient
    assert np.allclose(a0, 0.5 * (l0 + l1), rtol=1e-6, atol=1e-6)      # ... the mean of the per-rank gradients
    assert off0 == off1 and off0[0] == 0 and all(b > a for a, b in zip(off0, off0[1:]))   # .grad are views, in order


def _init_worker(rank, world, port, q):
    import torch.distributed as dist
    from moldiff_amd.trainer import FlatParams, broadcast_replicas_
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)              # every rank initialises its nn.Linear / LayerNorm from a DIFFERENT stream
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 2))
    flat = FlatParams(net)
    before = flat.data.clone()
    w = broadcast_replicas_((flat.data,), 0)
    q.put((rank, w, before.numpy(), flat.data.numpy().copy(), net[0].weight.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_from_rank0_weights_even_with_different_init_streams():
    """without a broadcast, default-initialised replicas differ and the averaged gradient is applied to N
    different models.  Trainer.sync_replicas = broadcast_replicas_ over the flat parameter buffer."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, b0, a0, lin0), (_, w1, b1, a1, lin1) = res
    assert w0 == w1 == 2
    assert np.abs(b0 - b1).max() > 0            # the default initialisations really differed
    assert np.array_equal(a0, b0)               # rank 0 keeps its weights
    assert np.array_equal(a1, a0)               # rank 1 now holds them too ...
    assert np.array_equal(lin1, lin0)           # ... and the module's parameters are views of the synced buffer


# ---- deferred class-range check: every rank fails in the same step ----------------------------------------------
def _defer_worker(rank, world, port, q):
    import torch.distributed as dist
    from moldiff_amd.diffusion import deferred_class_checks
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    outcome = []
    for bad_rank in (None, 1):              # a clean step, then a step in which ONLY rank 1 holds an out-of-range class id
        with deferred_class_checks() as chk:
            chk.items.append((torch.tensor(7 if rank != bad_rank else 9), 8))     # (max id of the batch, num_classes)
            chk.items.append((torch.tensor(5), 6))
        verify = chk.finish()
        try:
            verify()
            outcome.append('ok')
        except AssertionError as e:
            outcome.append(str(e))
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_deferred_class_check_raises_on_every_rank_in_the_same_step():
    """The rank with the bad batch raises the reference's AssertionError (models/diffusion.py:54); its peer raises too instead of
    waiting in the next gradient all-reduce for a rank that has left."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_defer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == 'ok'
    assert res[1][1] == 'Error: 9 >= 8'
    assert 'another rank' in res[0][1]


# ---- the same check on a SUB-group: ranks outside the trainer's group never reach it ------------------------------
def _defer_subgroup_worker(rank, world, port, q):
    import torch.distributed as dist
    from moldiff_amd.diffusion import deferred_class_checks
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    grp = dist.new_group([0, 1])            # (every rank creates it; only 0 and 1 are members)
    outcome = []
    if rank in (0, 1):
        for bad_rank in (None, 0):
            with deferred_class_checks() as chk:
                chk.items.append((torch.tensor(3 if rank != bad_rank else 8), 8))
            verify = chk.finish(group=grp)  # the flag all-reduce pairs with the GROUP's ranks; on WORLD it would wait for rank 2 for ever
            try:
                verify()
                outcome.append('ok')
            except AssertionError as e:
                outcome.append(str(e))
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_deferred_class_check_on_a_sub_group_does_not_wait_for_ranks_outside_it():
    """Trainer(group=...) hands its group to deferred_class_checks.finish: with three ranks and a trainer group of two, the third rank
    issues no collective at that point -- the check must complete on the two (a WORLD all-reduce would hang until the timeout below)
    and a bad batch on rank 0 must still fail rank 1 in the same step."""
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_defer_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[2] == []
    assert res[0][0] == res[1][0] == 'ok'
    assert res[0][1] == 'Error: 8 >= 8'
    assert 'another rank' in res[1][1]
