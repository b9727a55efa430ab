"""Sampling entry point -- the MI355X counterpart of the reference's ``scripts/sample_drug3d.py``.

    python -m moldiff_amd.sample_drug3d --config configs/sample_MolDiff_simple.yml --outdir ./outputs \
        --device cuda:0 [--batch_size N] [--recipe-weights]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m moldiff_amd.sample_drug3d ...

Same flags (--config --outdir --device --batch_size), same YAML keys (model.checkpoint, bond_predictor,
sample.{seed,batch_size,num_mols,save_traj_prob,guidance}), same seeding rule (seed + sum(ord(outdir)),
scripts/sample_drug3d.py:47), same batch-size rule ``min(batch_size, 2*remaining)`` (:112) and the same
give-up rule (:106-108).  What differs, deliberately:
  * the hot path runs in the HIP library, and decode/compaction of the predictions runs on the device
    (``FeaturizeMol.decode_batch``) instead of a per-molecule numpy loop over a full D2H copy;
  * RDKit reconstruction (utils/reconstruct.py, CPU chemistry) is out of scope: a molecule counts as
    finished when its decoded bond graph is connected (the reference's test is "no '.' in the SMILES");
    molecules are written as V2000 mol blocks (bond order 4 = aromatic) and collected in ``samples_all.pt``;
  * ``sample.save_traj_prob``: the trajectory is kept compact on the device (class ids, one byte per atom / half-edge and
    frame) and only the drawn molecules' frames are decoded (``traj_mol<id>.sdf``, one mol block per frame); the draw is a
    function of (seed, global molecule id) instead of the position in numpy's global stream, so it does not depend on the
    number of GPUs;
  * with WORLD_SIZE > 1 every rank samples a contiguous slice of each batch's cost-balanced molecule order (noise keyed by
    global molecule id); per batch the last-step predictions travel to rank 0 as tensors (``distributed.gather_pred``:
    one count all_gather + ONE padded gather-to-rank-0 of a flat buffer holding all three) and one 2-element all-reduce carries the loop condition.
No pretrained checkpoint ships with the reference (Google-Drive download); ``--recipe-weights`` substitutes the
deterministic synthetic weights used by the tests so the entry point can be exercised end to end.
"""
import argparse
import os
import shutil
import time

import numpy as np
import torch

from . import BondPredictor, MolDiff, _lib
from .distributed import balanced_order, gather_pred, shard_bounds
from .harness import default_config, load_config, placeholder_from_sizes, recipe_state_dict, seed_all
from .harness import GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS
from .postprocess import FeaturizeMol

ELEMENT_SYMBOL = {6: 'C', 7: 'N', 8: 'O', 9: 'F', 15: 'P', 16: 'S', 17: 'Cl'}


def is_connected(n_atoms, bond_index):
    """True when the undirected bond graph over n_atoms has one component (= no '.' in a SMILES)."""
    if n_atoms == 0:
        return False
    parent = list(range(n_atoms))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for i, j in bond_index.T:
        ri, rj = find(int(i)), find(int(j))
        if ri != rj:
            parent[ri] = rj
    return len({find(i) for i in range(n_atoms)}) == 1


def mol_block(info, name='moldiff_amd'):
    """V2000 mol block from a decode_output dict (one line per directed-pair's first half)."""
    ele, pos = info['element'], info['atom_pos']
    nb = info['bond_index'].shape[1] // 2
    lines = [name, '  moldiff_amd', '', '%3d%3d  0  0  0  0  0  0  0  0999 V2000' % (len(ele), nb)]
    for e, p in zip(ele, pos):
        lines.append('%10.4f%10.4f%10.4f %-3s 0  0  0  0  0  0  0  0  0  0  0  0' % (p[0], p[1], p[2], ELEMENT_SYMBOL.get(int(e), 'X')))
    for k in range(nb):
        lines.append('%3d%3d%3d  0' % (info['bond_index'][0, k] + 1, info['bond_index'][1, k] + 1, info['bond_type'][k]))
    lines.append('M  END')
    return '\n'.join(lines) + '\n'


def build_models(config, device, recipe):
    ckpt_path = config.model.checkpoint
    if os.path.exists(ckpt_path):
        ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
        train_config = ckpt['config']
        model = MolDiff(train_config.model, 8, 6)
        model.load_state_dict(ckpt['model'])
    elif recipe:
        kind = 'MolDiff' if 'bond_predictor' in config else 'MolDiff_simple'
        model = MolDiff(default_config(kind), 8, 6)
        model.load_state_dict(recipe_state_dict(model, 20230807))
        train_config = None
    else:
        raise FileNotFoundError(f'{ckpt_path} not found (the reference distributes checkpoints via Google Drive); '
                                f'pass --recipe-weights to run with synthetic weights')
    model = model.to(device).eval()
    bond_predictor, guidance = None, None
    if 'bond_predictor' in config:
        bp_path = config.bond_predictor
        if os.path.exists(bp_path):
            ck = torch.load(bp_path, map_location='cpu', weights_only=False)
            bond_predictor = BondPredictor(ck['config']['model'], 8, 5)
            bond_predictor.load_state_dict(ck['model'])
        elif recipe:
            bond_predictor = BondPredictor(default_config('bondpred'), 8, 5)
            bond_predictor.load_state_dict(recipe_state_dict(bond_predictor, 20230808))
        else:
            raise FileNotFoundError(bp_path)
        bond_predictor = bond_predictor.to(device).eval()
        # MOLDIFF_GUIDANCE_MATRIX_PATH=split_f16: only the guidance predictor on the split float16 path, the denoiser stays where
        # MOLDIFF_MATRIX_PATH puts it (exact fp32 by default) -- the 'mixed' configuration of DESIGN.md section 3.4
        gpath = os.environ.get('MOLDIFF_GUIDANCE_MATRIX_PATH')
        if gpath:
            bond_predictor.matrix_path = _lib.resolve_matrix_path(gpath)
    if 'guidance' in config.sample:
        guidance = config.sample.guidance
    return model, bond_predictor, guidance


def traj_blocks(featurizer, traj, sel, sizes, device):
    """Trajectories of the molecules `sel` (batch-local indices) as lists of decode_output dicts, one per frame
    (scripts/sample_drug3d.py:157-163).  The frames are cut out of the compact trajectory (class ids, one byte each),
    re-packed as a small batch of their own and decoded on the device frame by frame with ``decode_batch``."""
    node_traj, pos_traj, half_traj = traj
    sizes = np.asarray(sizes, dtype=np.int64)
    node_off = np.concatenate([[0], np.cumsum(sizes)])
    half_off = np.concatenate([[0], np.cumsum(sizes * (sizes - 1) // 2)])
    nidx = torch.as_tensor(np.concatenate([np.arange(node_off[m], node_off[m + 1]) for m in sel]), device=device)
    hidx = torch.as_tensor(np.concatenate([np.arange(half_off[m], half_off[m + 1]) for m in sel]), device=device)
    sub = placeholder_from_sizes(sizes[list(sel)], device)
    graph = _lib.Graph(torch.cat([sub['halfedge_index'], sub['halfedge_index'].flip(0)], dim=1), sub['batch_node'], len(sel))
    nt, pt, ht = node_traj[:, nidx], pos_traj[:, nidx], half_traj[:, hidx]   # still compact
    frames = []
    for t in range(pt.shape[0]):
        frames.append(featurizer.decode_batch([nt[t].dense(), pt[t].contiguous(), ht[t].dense()], sub['batch_node'],
                                              sub['halfedge_index'], sub['batch_halfedge'], len(sel), graph=graph))
    return {m: [frames[t][j] for t in range(len(frames))] for j, m in enumerate(sel)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=str, required=True)
    ap.add_argument('--outdir', type=str, default='./outputs')
    ap.add_argument('--device', type=str, default='cuda:0')
    ap.add_argument('--batch_size', type=int, default=0)
    ap.add_argument('--recipe-weights', action='store_true')
    ap.add_argument('--num_mols', type=int, default=0, help='override sample.num_mols')
    args = ap.parse_args(argv)

    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # one process per GPU over RCCL; MDX_DIST_BACKEND=gloo (ranks may then share a GPU) exists only so that the
        # multi-rank control flow can be exercised on a single-GPU test box
        backend = os.environ.get('MDX_DIST_BACKEND', 'nccl')
        lr = int(os.environ.get('LOCAL_RANK', '0'))
        args.device = f"cuda:{lr if backend == 'nccl' else lr % torch.cuda.device_count()}"
    device = torch.device(args.device)
    # the library allocates and launches on the CURRENT HIP device: make it the one the tensors live on (the reference's
    # default is --device cuda:7) before any handle is created
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    comm_dev = device if (world > 1 and backend == 'nccl') else torch.device('cpu')

    config = load_config(args.config)
    config_name = os.path.basename(args.config).rsplit('.', 1)[0]
    seed = int(config.sample.seed + np.sum([ord(s) for s in args.outdir]))
    seed_all(seed)
    log_dir = os.path.join(args.outdir, config_name + '_' + time.strftime('%Y%m%d_%H%M%S'))
    if world > 1:  # every rank writes trajectories of its own molecules: agree on rank 0's directory name
        name = [log_dir]
        dist.broadcast_object_list(name, src=0)   # once per run, not per batch
        log_dir = name[0]
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
        shutil.copyfile(args.config, os.path.join(log_dir, os.path.basename(args.config)))
    os.makedirs(log_dir + '_SDF', exist_ok=True)
    featurizer = FeaturizeMol([6, 7, 8, 9, 15, 16, 17], [1, 2, 3, 4], use_mask_node=True, use_mask_edge=True)
    model, bond_predictor, guidance = build_models(config, device, args.recipe_weights)
    num_mols = args.num_mols or config.sample.num_mols
    batch_size = args.batch_size if args.batch_size > 0 else config.sample.batch_size
    save_traj_prob = float(getattr(config.sample, 'save_traj_prob', 0.0) or 0.0)
    pool = {'finished': [], 'failed': []}
    n_finished, n_failed = 0, 0
    next_id, i_batch = 0, 0
    while n_finished < num_mols:
        if n_failed > 3 * num_mols:
            if rank == 0:
                print('Too many failed molecules. Stop sampling.')
            break
        n_graphs = min(batch_size, (num_mols - n_finished) * 2)
        # every rank draws the same sizes (same numpy stream) and takes a contiguous slice of the cost-balanced order
        sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=n_graphs).astype('int64')
        sizes = np.maximum(sizes, 2)  # the reference's harness cannot handle molecules without half-edges
        order = balanced_order(sizes, world) if world > 1 else np.arange(n_graphs)
        lo, hi = shard_bounds(n_graphs, world, rank)
        mine = order[lo:hi]
        ph = placeholder_from_sizes(sizes[mine], device)
        ids = next_id + mine.astype(np.int64)
        out = model.sample(hi - lo, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], bond_predictor, guidance,
                           seed=seed + i_batch, mol_ids=ids, return_traj=save_traj_prob > 0)
        # trajectories stay rank-local (scripts/sample_drug3d.py:155 looks at ~2 % of them): the owner decodes and writes
        # them, named by global molecule id; whether a molecule is drawn depends only on (seed, id), not on the sharding
        if save_traj_prob > 0:
            local = featurizer.decode_batch(out['pred'], ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], hi - lo)
            sel = [j for j, info in enumerate(local)
                   if is_connected(len(info['element']), info['bond_index'])
                   and np.random.default_rng([seed, int(ids[j])]).random() < save_traj_prob]
            if sel:
                for j, frames in traj_blocks(featurizer, out['traj'], sel, sizes[mine], device).items():
                    path = os.path.join(log_dir + '_SDF', 'traj_mol%d.sdf' % int(ids[j]))
                    with open(path, 'w') as f:
                        for info in frames:
                            f.write(mol_block(info) + '$$$$\n')
        # the only data-path collective: this batch's last-step predictions to rank 0 (one count all_gather + one padded
        # gather-to-rank-0 of a flat buffer, RCCL over xGMI; about 2 MB per rank at 256 molecules)
        pred = out['pred']
        if dist is not None:
            pred = gather_pred([p.to(comm_dev) for p in pred], dst=0)
        counts = torch.zeros(2, dtype=torch.int64, device=comm_dev)
        if rank == 0:
            full = placeholder_from_sizes(sizes[order], device)
            mols = featurizer.decode_batch([p.to(device) for p in pred], full['batch_node'], full['halfedge_index'],
                                           full['batch_halfedge'], n_graphs)
            inv = np.argsort(order)                       # back to the batch's original molecule order
            mols = [mols[inv[k]] for k in range(n_graphs)]
            gen = []
            for k, info in enumerate(mols):
                info['mol_id'] = next_id + k
                if is_connected(len(info['element']), info['bond_index']):
                    gen.append(info)
                else:
                    pool['failed'].append(info)
            for i, info in enumerate(gen):
                with open(os.path.join(log_dir + '_SDF', '%d.sdf' % (i + len(pool['finished']))), 'w') as f:
                    f.write(mol_block(info) + '$$$$\n')
                # whether a finished molecule's trajectory was written is a function of (seed, global id) and of its connectivity
                # (tested above) -- not of the rank that sampled it: rank 0 re-derives the file name its owner used, so every
                # selected molecule carries its trajectory like in the reference (scripts/sample_drug3d.py:155-190)
                if save_traj_prob > 0 and np.random.default_rng([seed, int(info['mol_id'])]).random() < save_traj_prob:
                    info['traj_file'] = 'traj_mol%d.sdf' % info['mol_id']
            pool['finished'].extend(gen)
            print('[Pool] Finished %d | Failed %d' % (len(pool['finished']), len(pool['failed'])))
            counts[0], counts[1] = len(pool['finished']), len(pool['failed'])
        if dist is not None:  # one small all-reduce keeps the loop condition identical on every rank
            dist.all_reduce(counts)
        n_finished, n_failed = int(counts[0]), int(counts[1])
        next_id += n_graphs
        i_batch += 1
    if rank == 0:
        torch.save(pool, os.path.join(log_dir, 'samples_all.pt'))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return log_dir


if __name__ == '__main__':
    main()
