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


def gather_variable(t, dst=0, group=None):
    """Gather row-variable tensors (n_r, C) from every rank to `dst`; returns the list on dst, None elsewhere.
    One all_gather of the row counts + one padded all_gather of the payload."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    pad = max(counts + [1])
    buf = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    buf[:t.shape[0]] = t
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    if dist.get_rank(group) != dst:
        return None
    return [o[:c] for o, c in zip(outs, counts)]


def gather_pred(pred, dst=0, group=None):
    """pred = [pred_node (N_r,Kn), pred_pos (N_r,3), pred_halfedge (Eh_r,Ke)] -> concatenated over ranks on dst
    (rank order == global molecule order for contiguous shards)."""
    parts = [gather_variable(p.contiguous(), dst, group) for p in pred]
    if parts[0] is None:
        return None
    return [torch.cat(p, 0) for p in parts]
