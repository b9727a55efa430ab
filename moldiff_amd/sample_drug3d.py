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
  * with WORLD_SIZE > 1 every rank samples a contiguous slice of each batch (noise keyed by global molecule
    id) and rank 0 gathers the decoded molecules -- the only collective of the run.
No pretrained checkpoint ships with the reference (Google-Drive download); ``--recipe-weights`` substitutes the
deterministic synthetic weights used by the tests so the entry point can be exercised end to end.
"""
import argparse
import os
import shutil
import time

import numpy as np
import torch

from . import BondPredictor, MolDiff
from .distributed import shard_bounds
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
    if 'guidance' in config.sample:
        guidance = config.sample.guidance
    return model, bond_predictor, guidance


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
        torch.cuda.set_device(torch.device(args.device))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(args.device))
        else:
            dist.init_process_group(backend)
    device = torch.device(args.device)

    config = load_config(args.config)
    config_name = os.path.basename(args.config).rsplit('.', 1)[0]
    seed = int(config.sample.seed + np.sum([ord(s) for s in args.outdir]))
    seed_all(seed)
    log_dir = os.path.join(args.outdir, config_name + '_' + time.strftime('%Y%m%d_%H%M%S'))
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
        os.makedirs(log_dir + '_SDF', exist_ok=True)
        shutil.copyfile(args.config, os.path.join(log_dir, os.path.basename(args.config)))
    featurizer = FeaturizeMol([6, 7, 8, 9, 15, 16, 17], [1, 2, 3, 4], use_mask_node=True, use_mask_edge=True)
    model, bond_predictor, guidance = build_models(config, device, args.recipe_weights)
    num_mols = args.num_mols or config.sample.num_mols
    batch_size = args.batch_size if args.batch_size > 0 else config.sample.batch_size
    pool = {'finished': [], 'failed': []}
    next_id, i_batch = 0, 0
    while len(pool['finished']) < num_mols:
        if len(pool['failed']) > 3 * num_mols:
            print('Too many failed molecules. Stop sampling.')
            break
        n_graphs = min(batch_size, (num_mols - len(pool['finished'])) * 2)
        # every rank draws the same sizes (same numpy stream), then takes its slice
        sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=n_graphs).astype('int64')
        sizes = np.maximum(sizes, 2)  # the reference's harness cannot handle molecules without half-edges
        lo, hi = shard_bounds(n_graphs, world, rank)
        ph = placeholder_from_sizes(sizes[lo:hi], device)
        ids = np.arange(next_id + lo, next_id + hi, dtype=np.int64)
        next_id += n_graphs
        out = model.sample(hi - lo, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], bond_predictor, guidance,
                           seed=seed + i_batch, mol_ids=ids, return_traj=False)
        mols = featurizer.decode_batch(out['pred'], ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], hi - lo)
        if dist is not None:
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(mols, gathered, dst=0)
            mols = [m for part in gathered for m in part] if rank == 0 else []
        if rank == 0:
            gen = []
            for info in mols:
                if is_connected(len(info['element']), info['bond_index']):
                    gen.append(info)
                else:
                    pool['failed'].append(info)
            for i, info in enumerate(gen):
                with open(os.path.join(log_dir + '_SDF', '%d.sdf' % (i + len(pool['finished']))), 'w') as f:
                    f.write(mol_block(info) + '$$$$\n')
            pool['finished'].extend(gen)
            print('[Pool] Finished %d | Failed %d' % (len(pool['finished']), len(pool['failed'])))
        if dist is not None:  # keep the loop condition identical on every rank
            counts = [len(pool['finished']), len(pool['failed'])]
            dist.broadcast_object_list(counts, src=0)
            if rank != 0:
                pool['finished'], pool['failed'] = [None] * counts[0], [None] * counts[1]
        i_batch += 1
    if rank == 0:
        torch.save(pool, os.path.join(log_dir, 'samples_all.pt'))
    if dist is not None:
        dist.destroy_process_group()
    return log_dir


if __name__ == '__main__':
    main()
