"""Multi-GPU sampling: independent per-GPU streams + one end-of-run gather (RCCL over xGMI).

The reference is single-device.  Molecules are independent (block-diagonal graph,
per-graph time step, models/model.py:272), so the N-GPU path is: every rank draws the SAME global size list,
takes a contiguous slice of molecules, samples it with noise keyed by global molecule id (so the result of a
molecule does not depend on the number of GPUs), and only the final `pred` tensors travel -- one
variable-length gather to rank 0 (about 2 MB per rank at 256 molecules: latency-bound, no bucketing needed).
`traj` stays rank-local (the reference only looks at ~2% of trajectories, scripts/sample_drug3d.py:155).
Works with backend "nccl" (= RCCL on ROCm, device tensors) and "gloo" (CPU tensors, used by the CPU tests).
"""
import numpy as np
import torch


def shard_bounds(n_items, world_size, rank):
    """Contiguous near-equal slices: the first (n % world) ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_molecules(n_nodes_list, world_size, rank):
    """-> (sizes of this rank's molecules, their global molecule ids)."""
    sizes = np.asarray(n_nodes_list, dtype=np.int64)
    lo, hi = shard_bounds(len(sizes), world_size, rank)
    return sizes[lo:hi], np.arange(lo, hi, dtype=np.int64)


def balanced_order(n_nodes_list, world_size):
    """Optional load balancing: permutation of molecule ids such that contiguous sharding of the permuted list
    gives every rank a similar sum of n^2 (cost is proportional to directed edges): serpentine deal by size."""
    sizes = np.asarray(n_nodes_list, dtype=np.int64)
    order = np.argsort(-sizes, kind='stable')
    buckets = [[] for _ in range(world_size)]
    for j, mol in enumerate(order):
        r = j % (2 * world_size)
        buckets[r if r < world_size else 2 * world_size - 1 - r].append(int(mol))
    return np.array([m for b in buckets for m in b], dtype=np.int64)


def _gather_rows(flat, dst, group, extra=()):
    """(n_r,) payloads of different lengths -> list of (n_r,) tensors on `dst`, None elsewhere.  The element counts go to everybody
    (8 bytes per rank: every rank needs the common padded length), the payload travels to `dst` ONLY (`dist.gather`: point-to-point
    sends over xGMI with RCCL) -- an all_gather would move world_size times the bytes to ranks that drop them.
    extra: a few integers per rank that ride on the int64 count exchange (exact whatever the payload's dtype); with it the result
    is (parts, [extra of rank 0, extra of rank 1, ...])."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([flat.numel()] + [int(x) for x in extra], dtype=torch.int64, device=flat.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    extras = [[int(v) for v in c[1:].tolist()] for c in counts]
    counts = [int(c[0].item()) for c in counts]
    pad = max(counts + [1])
    buf = torch.zeros(pad, dtype=flat.dtype, device=flat.device)
    buf[:flat.numel()] = flat
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gather_list=outs, dst=dst_global, group=group)
    if rank != dst:
        return None
    parts = [o[:c] for o, c in zip(outs, counts)]
    return (parts, extras) if len(extra) else parts


def gather_variable(t, dst=0, group=None):
    """Gather row-variable tensors (n_r, C) from every rank to `dst` (group rank); returns the list on dst, None elsewhere."""
    tail = tuple(t.shape[1:])
    parts = _gather_rows(t.contiguous().reshape(-1), dst, group)
    if parts is None:
        return None
    return [p.reshape((-1,) + tail) for p in parts]


def gather_pred(pred, dst=0, group=None):
    """pred = [pred_node (N_r,Kn), pred_pos (N_r,3), pred_halfedge (Eh_r,Ke)] -> concatenated over ranks on dst
    (rank order == global molecule order for contiguous shards).  The three tensors of a rank travel as ONE flat buffer headed by
    row counts: one count exchange + one gather-to-dst per batch (latency-bound at ~2 MB per rank: fewer, larger messages).  The row
    counts travel as int64 with the count exchange, not in the payload, so they are exact for any prediction dtype; the three tensors
    must share one dtype (they share the buffer)."""
    dt = pred[0].dtype
    if any(p.dtype != dt for p in pred):
        raise TypeError('gather_pred: the three prediction tensors must have one dtype, got %s' % [p.dtype for p in pred])
    got = _gather_rows(torch.cat([p.contiguous().reshape(-1) for p in pred]), dst, group, extra=[p.shape[0] for p in pred])
    if got is None:
        return None
    parts, all_rows = got
    widths = [int(p.shape[1]) for p in pred]
    cols = [[] for _ in pred]
    for flat, rows in zip(parts, all_rows):
        off = 0
        for j, (r, w) in enumerate(zip(rows, widths)):
            cols[j].append(flat[off:off + r * w].reshape(r, w))
            off += r * w
    return [torch.cat(c, 0) for c in cols]
